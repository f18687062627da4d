#!/bin/bash
TAG=${1:-r5o}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
{ for r in 1 2 3; do
    echo -n "two copy streams: "; python3 tools/loop_trace.py pipe_host 200 2>/dev/null
    echo -n "one copy stream:  "; SPX_PIPE_ONE_COPY_STREAM=1 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
  done
  echo -n "two copy streams, depth 6: "; SPX_PROBE_DEPTH=6 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
  echo -n "two copy streams, no gather: "; SPEEDY_HIP_LIB=$T SPX_PIPE_NO_GATHER=1 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
} 2>&1 | tee "$OUT/${TAG}_copy_streams.txt"
