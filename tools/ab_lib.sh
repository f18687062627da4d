#!/bin/bash
# GPU box: A/B of library builds / tuning variables on the bench workload, interleaved so that drift cancels.
# (tuning variables need the tuning build: make -C speedy_amd/csrc tuning; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so)
#   bash tools/ab_lib.sh TAG REPS "label|ENV=.. ENV=.." "label2|..." ...
# e.g. bash tools/ab_lib.sh r02x 2 "new|" "base|SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_base.so"
TAG=${1:-ab}; REPS=${2:-2}; shift; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
if [ $# -eq 0 ]; then set -- "new|" "base|SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_base.so"; fi
for r in $(seq $REPS); do
  for v in "$@"; do
    label=${v%%|*}; envs=${v#*|}
    for m in ${AB_MODES:-serial conc}; do
      if [ $m = serial ]; then S="--serial"; else S=""; fi
      line=$(env SPX_NOOP=1 $envs timeout 300 python3 bench.py $S --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates 2>/dev/null | tail -1)
      echo "$line" | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
    g=lambda w: [v for n, v in k.items() if w in n][0]   # keys are the real kernel names since round 3
    print('%-8s %-6s ms/step=%.3f walk=%.3f analysis=%.3f tension=%.3f' % ('$label', '$m', d['ms_per_step'], g('walk'), g('analysis'), g('tension')))
except Exception as e:
    print('$label $m FAILED', e)
" | tee -a "$OUT/${TAG}_ab.txt"
    done
  done
done
