#!/bin/bash
# the default bench line (+ summary)
TAG=${1:-r05}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
t0=$(date +%s)
timeout 900 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; tail -c 400 "$OUT/${TAG}_bench.err"
echo "bench wall $(( $(date +%s) - t0 )) s"
python3 tools/bench_summary.py "$OUT/${TAG}_bench.json"
