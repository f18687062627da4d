#!/bin/bash
TAG=${1:-r5d}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python3 tools/numa_probe.py > "$OUT/${TAG}_numa_probe.txt" 2>&1; cat "$OUT/${TAG}_numa_probe.txt"
python3 tools/kernel_resources.py "$OUT/kernel_resources.json"
{ for o in lib_first pipe_first; do python3 tools/order_probe.py $o 4 2>/dev/null; done
  python3 tools/order_probe.py lib_first 6 2>/dev/null; } > "$OUT/${TAG}_order_probe.txt"; cat "$OUT/${TAG}_order_probe.txt"
timeout 300 python3 tools/scale_streams.py 256 512 > "$OUT/${TAG}_scale.txt" 2>&1; cat "$OUT/${TAG}_scale.txt"
timeout 600 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; python3 tools/bench_summary.py "$OUT/${TAG}_bench.json" | head -12
