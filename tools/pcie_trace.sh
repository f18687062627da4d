#!/bin/bash
# which engine moves the PCIe leg's data: rocprofv3 kernel trace + memory-copy trace of bench.py's pcie leg only
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/pcie_trace" -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/pcie_trace.log" 2>&1
ls "$OUT/pcie_trace"
for f in "$OUT"/pcie_trace/*stats*.csv; do echo "== $f"; head -12 "$f" | cut -c1-200; done
