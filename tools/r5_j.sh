#!/bin/bash
TAG=${1:-r5j}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 2400 python3 -m pytest tests -m gpu -q > "$OUT/${TAG}_pytest_full.log" 2>&1; tail -6 "$OUT/${TAG}_pytest_full.log"
bash tools/sq_counters.sh $TAG 2>&1 | tail -5
