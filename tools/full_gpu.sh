#!/bin/bash
# full GPU suite + the default bench line (+ optional tag):  gpurun -- bash tools/full_gpu.sh TAG
TAG=${1:-r04x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1; tail -4 "$OUT/${TAG}_pytest.log"
timeout 900 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; tail -c 300 "$OUT/${TAG}_bench.err"; python3 tools/bench_summary.py "$OUT/${TAG}_bench.json"
