#!/bin/bash
TAG=${1:-r5e}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python3 tools/numa_probe.py > "$OUT/${TAG}_numa_probe.txt" 2>&1; cat "$OUT/${TAG}_numa_probe.txt"
for k in turns3 pipe_dev; do
  python3 tools/loop_trace.py $k 2>/dev/null
  rocprofv3 --kernel-trace -d "$OUT/${TAG}_trace_$k" -o t --output-format csv -- python3 tools/loop_trace.py $k > "$OUT/${TAG}_trace_$k.log" 2>&1
  tail -1 "$OUT/${TAG}_trace_$k.log"
  python3 tools/trace_summary.py $(find "$OUT/${TAG}_trace_$k" -name "*kernel_trace.csv" | head -1) 30 | tee "$OUT/${TAG}_trace_${k}_summary.txt"
  python3 tools/trace_step.py $(find "$OUT/${TAG}_trace_$k" -name "*kernel_trace.csv" | head -1) 24 > "$OUT/${TAG}_trace_${k}_last.txt"
done
