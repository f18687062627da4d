#!/bin/bash
# GPU box: tools/r6/grouped_probe.py over group sizes and walk forms (tuning library).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6j}
OUT=$PWD/gpurun_out; mkdir -p $OUT
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
run() { echo "== $1"; shift; env "$@" 2>&1 | tail -2; }
{
  for g in 8 4; do
    run "throughput form 2+0, G=$g" SPEEDY_HIP_LIB=$T SPX_NO_EXCLUSIVE_CU=1 SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 timeout 300 python3 tools/r6/grouped_probe.py $g 2
  done
  run "throughput form 2+0, G=8, three halves" SPEEDY_HIP_LIB=$T SPX_NO_EXCLUSIVE_CU=1 SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 timeout 300 python3 tools/r6/grouped_probe.py 8 3
  run "lean form 4+0, G=4" SPEEDY_HIP_LIB=$T SPX_NO_EXCLUSIVE_CU=1 SPX_WALK_NWM=4 SPX_WALK_NWC=0 timeout 300 python3 tools/r6/grouped_probe.py 4 2
  run "library's choice, no exclusive CU, G=4" SPEEDY_HIP_LIB=$T SPX_NO_EXCLUSIVE_CU=1 timeout 300 python3 tools/r6/grouped_probe.py 4 2
} | tee $OUT/${TAG}_grouped_probe.txt
