#!/bin/bash
TAG=${1:-r5g}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
{ for i in 1 2; do python3 tools/loop_trace.py turns3 2>/dev/null; python3 tools/loop_trace.py pipe_dev 2>/dev/null; python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done
  SPX_PROBE_DEPTH=6 python3 tools/loop_trace.py pipe_host 200 2>/dev/null; } | tee "$OUT/${TAG}_loops.txt"
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ahead.py -m gpu -x -q 2>&1 | tail -3
rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/${TAG}_trace_host" -o t --output-format csv -- python3 tools/loop_trace.py pipe_host 120 > "$OUT/${TAG}_trace_host.log" 2>&1
tail -1 "$OUT/${TAG}_trace_host.log"
python3 tools/copy_trace_summary.py "$OUT/${TAG}_trace_host" 70 > "$OUT/${TAG}_trace_host_timeline.txt" 2>&1; head -3 "$OUT/${TAG}_trace_host_timeline.txt"
rm -rf "$OUT/${TAG}_trace_host"
