#!/bin/bash
# GPU box: the whole GPU suite, then the API thread figures and the pipelined loop.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6h}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/${TAG}_tests.txt
B=speedy_amd/lib/stream_bench
{
  for spec in "256 threads:16" "256 threads:16" "256 threads:64" "256 threads:256" "256 percall"; do
    set -- $spec
    echo -n "$spec: "; timeout 300 $B $1 8 1000 3.5 1 $2 16000 2>&1 | tail -1
  done
} | tee $OUT/${TAG}_api.txt
{ for r in 1 2; do echo -n "pipelined loop: "; python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1; done; } | tee $OUT/${TAG}_loop.txt
