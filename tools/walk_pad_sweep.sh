#!/bin/bash
# Walk kernel time of the bench batch for every code-placement variant built as speedy_amd/lib/ab/libspeedy_hip_pad<N>.so
# (SPX_WALK_PAD = N s_nop's in the kernel's prologue: the step loop shifted by 4 N bytes), the shipped library first and last.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --steps 12 --warmup 3"
P='import json,sys; d=json.loads(sys.stdin.read()); print("step %.3f ms  walk %.3f ms" % (d["ms_per_step"], [v for k,v in d["roofline"]["kernel_ms_per_step"].items() if "walk" in k][0]))'
echo -n "shipped: "; $B 2>/dev/null | python3 -c "$P"
for n in $(seq 0 ${1:-16}); do
  L=$PWD/speedy_amd/lib/ab/libspeedy_hip_pad$n.so
  [ -f $L ] || continue
  echo -n "pad $n ($((4*n)) bytes): "; SPEEDY_HIP_LIB=$L $B 2>/dev/null | python3 -c "$P"
done
echo -n "shipped: "; $B 2>/dev/null | python3 -c "$P"
