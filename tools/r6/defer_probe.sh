#!/bin/bash
# GPU box: GPU tests that exercise the forms without output waves, then the pipelined loop and a kernel trace.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6c}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_ahead.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/${TAG}_tests.txt
{ for r in 1 2; do echo -n "pipelined loop: "; python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1; done; } | tee $OUT/${TAG}_loop.txt
rm -rf $OUT/${TAG}_trace
rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o t --output-format csv -- python3 tools/loop_trace.py pipe_dev 40 > $OUT/${TAG}_trace.log 2>&1
f=$(find $OUT/${TAG}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $f 30 --timeline 30 > $OUT/${TAG}_trace_summary.txt 2>&1
rm -rf $OUT/${TAG}_trace
head -45 $OUT/${TAG}_trace_summary.txt
