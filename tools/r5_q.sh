#!/bin/bash
TAG=${1:-r5q}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 2400 python3 -m pytest tests -m gpu -q -rs > "$OUT/${TAG}_pytest_full.log" 2>&1; tail -8 "$OUT/${TAG}_pytest_full.log"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; python3 tools/bench_summary.py "$OUT/${TAG}_bench.json" | head -4
timeout 600 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-api --no-config4 --steps 20 --warmup 3 > "$OUT/${TAG}_bench_2rank_1gpu.json" 2> "$OUT/${TAG}_bench_2rank.err"; python3 tools/bench_summary.py "$OUT/${TAG}_bench_2rank_1gpu.json" | head -2
