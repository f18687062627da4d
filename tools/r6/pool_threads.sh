#!/bin/bash
# GPU box: the drop-in API in the reference's call order -- one thread (percall), then T threads x handles (ORDER=threads:T).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6g}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_pool.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/${TAG}_tests.txt
B=speedy_amd/lib/stream_bench
{
  for spec in "256 rounds" "256 percall" "256 threads:16" "256 threads:16" "256 threads:16" "256 threads:32" "256 threads:64" "64 threads:16" "16 threads:16"; do
    set -- $spec
    echo -n "$spec: "; timeout 300 $B $1 8 1000 3.5 1 $2 16000 2>&1 | tail -1
  done
  for g in 0 8 48; do echo -n "256 threads:16, SPX_POOL_GATHER_US=$g: "; SPX_POOL_GATHER_US=$g timeout 300 $B 256 8 1000 3.5 1 threads:16 16000 2>&1 | tail -1; done
  echo -n "256 threads:16, SPX_POOL_SPIN_US=0 (waiters block at once): "; SPX_POOL_SPIN_US=0 timeout 300 $B 256 8 1000 3.5 1 threads:16 16000 2>&1 | tail -1
  echo "256 threads:16 with SPX_POOL_TIMES=1:"; SPX_POOL_TIMES=1 timeout 300 $B 256 8 1000 3.5 1 threads:16 16000 2>&1 | tail -3
  echo "16 rounds with SPX_POOL_TIMES=1:"; SPX_POOL_TIMES=1 timeout 300 $B 16 8 1000 3.5 1 rounds 16000 2>&1 | tail -3
} | tee $OUT/${TAG}_api.txt
