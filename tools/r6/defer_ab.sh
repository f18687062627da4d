#!/bin/bash
# GPU box: deferred output -- where its kernel and the tension kernel run (tuning library switches), two rounds; a trace of one.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6d}
OUT=$PWD/gpurun_out; mkdir -p $OUT
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
python3 -m pytest tests/test_gpu_ahead.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/${TAG}_tests.txt
run() { echo -n "$1: "; shift; env "$@" python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1; }
{ for r in 1 2; do
  run "shipped (output stream, tension behind the analysis)" X=1
  run "output on the walk stream" SPEEDY_HIP_LIB=$T SPX_NO_OUT_STREAM=1
  run "output stream + tension stream" SPEEDY_HIP_LIB=$T SPX_TENSION_STREAM=1
  run "output on the walk stream + tension stream" SPEEDY_HIP_LIB=$T SPX_NO_OUT_STREAM=1 SPX_TENSION_STREAM=1
  run "output stream + tension stream, three walk streams" SPEEDY_HIP_LIB=$T SPX_TENSION_STREAM=1 SPX_WALK_STREAMS=3
done; } | tee $OUT/${TAG}_ab.txt
for v in a b; do
  rm -rf $OUT/${TAG}_trace
  if [ $v = a ]; then E="X=1"; else E="SPX_TENSION_STREAM=1"; fi
  env SPEEDY_HIP_LIB=$T $E rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o t --output-format csv -- python3 tools/loop_trace.py pipe_dev 40 > $OUT/${TAG}_trace.log 2>&1
  f=$(find $OUT/${TAG}_trace -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_summary.py $f 30 --timeline 36 > $OUT/${TAG}_trace_${v}_summary.txt 2>&1
  rm -rf $OUT/${TAG}_trace
  head -48 $OUT/${TAG}_trace_${v}_summary.txt | tail -43
done
