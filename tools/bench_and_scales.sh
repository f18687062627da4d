#!/bin/bash
TAG=${1:-r04x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 900 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; tail -c 300 "$OUT/${TAG}_bench.err"; python3 tools/bench_summary.py "$OUT/${TAG}_bench.json"
python3 tools/scale_configs.py > "$OUT/${TAG}_scale_configs.txt" 2>&1; cat "$OUT/${TAG}_scale_configs.txt"
python3 tools/scale_streams.py > "$OUT/${TAG}_scale_streams.txt" 2>&1; cat "$OUT/${TAG}_scale_streams.txt"
