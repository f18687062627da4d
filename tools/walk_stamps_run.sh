#!/bin/bash
TAG=${1:-r04x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python3 tools/walk_stamps.py > "$OUT/${TAG}_walk_stamps.txt" 2>&1; cat "$OUT/${TAG}_walk_stamps.txt"
bash tools/sq_counters.sh ${TAG} 2>&1 | tail -5
python3 tools/perf_reference.py
