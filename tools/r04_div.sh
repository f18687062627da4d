#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
import torch
from speedy_amd._lib import lib
L=lib()
for lo,hi in ((-17,8),(-17,40),(40,80),(80,120),(100,127),(-60,-17),(-126,-60)):
    print(lo,hi, L.spx_debug_fdiv_check(1, 1<<20, lo, hi), flush=True)
PY
bash tools/ab_variants.sh 3 ieeediv
