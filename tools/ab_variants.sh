#!/bin/bash
# GPU box: walk-kernel time of the bench batch for the shipped library and for every variant named (speedy_amd/lib/ab/
# libspeedy_hip_<NAME>.so, tools/build_variant.sh), REPS rounds interleaved.   bash tools/ab_variants.sh REPS NAME...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPS=$1; shift
B="python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --steps 12 --warmup 3"
P='import json,sys; d=json.loads(sys.stdin.read()); print("step %.3f ms  walk %.3f ms" % (d["ms_per_step"], [v for k,v in d["roofline"]["kernel_ms_per_step"].items() if "walk" in k][0]))'
for r in $(seq $REPS); do
  echo -n "shipped: "; $B 2>/dev/null | python3 -c "$P"
  for n in "$@"; do
    L=$PWD/speedy_amd/lib/ab/libspeedy_hip_$n.so
    [ -f $L ] || { echo "$n: not built"; continue; }
    echo -n "$n: "; SPEEDY_HIP_LIB=$L $B 2>/dev/null | python3 -c "$P"
  done
done
