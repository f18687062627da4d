#!/bin/bash
# SQ instruction / wait counters of the kernels (two PMC passes, kernel-trace only).  Output: gpurun_out/TAG_sq{1,2}/
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
export SPX_SERIAL=1   # kernels one after the other: counters per kernel are not blurred by sharing
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/${TAG}_sq1" -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
  -d "$OUT/${TAG}_sq2" -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_sq2.log" 2>&1
ls "$OUT/${TAG}_sq1" "$OUT/${TAG}_sq2"
