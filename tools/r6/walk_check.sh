#!/bin/bash
# GPU box: a walk-kernel change -- the tests that compare every walk form with the oracle, then the pipelined loop and the plain call.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6e}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_ahead.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/${TAG}_tests.txt
{ for r in 1 2 3; do echo -n "pipelined loop: "; python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1; done
  bash tools/variant_times.sh 2>&1 | tail -4; } | tee $OUT/${TAG}_loop.txt
