#!/bin/bash
TAG=${1:-r5n}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
{ for r in 1 2; do
    echo -n "two walk streams (shipped):  "; python3 tools/loop_trace.py pipe_dev 100 2>/dev/null
    echo -n "three walk streams (tuning): "; SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS3=1 python3 tools/loop_trace.py pipe_dev 100 2>/dev/null
  done
  echo -n "host pipeline (shipped):            "; python3 tools/loop_trace.py pipe_host 200 2>/dev/null
  echo -n "host pipeline, NO gather kernel:    "; SPEEDY_HIP_LIB=$T SPX_PIPE_NO_GATHER=1 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
  echo -n "host pipeline, three walk streams:  "; SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS3=1 python3 tools/loop_trace.py pipe_host 200 2>/dev/null
} 2>&1 | tee "$OUT/${TAG}_walk3.txt"
