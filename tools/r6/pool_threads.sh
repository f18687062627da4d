#!/bin/bash
# GPU box: the drop-in API in the reference's call order -- one thread (percall), then T threads x handles (ORDER=threads:T);
# the walk A/B (ragged-round skip against -DSPX_NO_RAGGED_SKIP) interleaved on the same box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6f}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_pool.py tests/test_gpu_sonic2.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/${TAG}_tests.txt
B=speedy_amd/lib/stream_bench
{
  for spec in "256 rounds" "256 percall" "16 percall" "256 threads:16" "256 threads:16" "256 threads:64" "256 threads:256" "1024 threads:16" "64 threads:16" "16 threads:16"; do
    set -- $spec
    echo -n "$spec: "; timeout 300 $B $1 8 1000 3.5 1 $2 16000 2>&1 | tail -1
  done
  for g in 0 8 48; do echo -n "256 threads:16, SPX_POOL_GATHER_US=$g: "; SPX_POOL_GATHER_US=$g timeout 300 $B 256 8 1000 3.5 1 threads:16 16000 2>&1 | tail -1; done
} | tee $OUT/${TAG}_api.txt
bash tools/variant_times.sh noskip 2>&1 | tee $OUT/${TAG}_walk_ab.txt
{ for r in 1 2; do echo -n "shipped: "; python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1
  echo -n "noskip: "; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_noskip.so python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1; done; } | tee -a $OUT/${TAG}_walk_ab.txt
