#!/bin/bash
# GPU box: A/B of two builds of the library on the bench workload, interleaved so that box-to-box and time drift cancel.
#   bash tools/ab_lib.sh TAG [BASE_SO] [REPS]      BASE_SO defaults to speedy_amd/lib/ab/libspeedy_hip_base.so
TAG=${1:-ab}
BASE=${2:-speedy_amd/lib/ab/libspeedy_hip_base.so}
REPS=${3:-2}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
for r in $(seq $REPS); do
  for which in new base; do
    for m in serial conc; do
      if [ $m = serial ]; then S="SPX_SERIAL=1"; else S="SPX_NOOP=1"; fi
      if [ $which = base ]; then LIB="SPEEDY_HIP_LIB=$PWD/$BASE"; else LIB="SPX_NOOP2=1"; fi
      line=$(env $S $LIB timeout 300 python3 bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1)
      echo "$line" | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
    print('%-5s %-6s ms/step=%.3f walk=%.3f analysis=%.3f tension=%.3f' % ('$which', '$m', d['ms_per_step'], k['spx_walk_kernel'], k['spx_analysis_kernel'], k['spx_tension_kernel']))
except Exception as e:
    print('$which $m FAILED', e)
" | tee -a "$OUT/${TAG}_ab.txt"
    done
  done
done
