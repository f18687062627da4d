#!/bin/bash
# A/B of library variants on the bench batch + correctness of the shipped library first
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
TAG=$1; shift
{
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not soak" 2>&1 | tail -3
bash tools/ab_variants.sh 3 "$@"
} > "$OUT/${TAG}_ab.log" 2>&1
cat "$OUT/${TAG}_ab.log"
