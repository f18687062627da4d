#!/bin/bash
# pipelined calls with the analysis workgroups padded in LDS: fewer of them beside a live walk workgroup, more on a freed CU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
B="python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined --steps 20 --warmup 4"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("step %.3f ms  kernels %s" % (d["ms_per_step"], {k.split("<")[0][4:12]: round(v,3) for k,v in d["roofline"]["kernel_ms_per_step"].items()}))'
for rep in 1 2; do
for pad in 0 4096 7168 9216 16384 44000; do
  echo -n "pad $pad: "; SPX_AHEAD_ANY=1 SPX_ANALYSIS_LDS_PAD=$pad $B 2>/dev/null | python3 -c "$P"
done
done
