#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
{ bash tools/walk_pad_sweep.sh 16; bash tools/walk_pad_sweep.sh 16; } > "$OUT/r04n_walk_pad.txt" 2>&1
cat "$OUT/r04n_walk_pad.txt"
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kernel_resources or register_budgets or walk_form" 2>&1 | tail -5
