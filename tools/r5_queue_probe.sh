#!/bin/bash
TAG=${1:-r5f}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
{ echo -n "turns3 shipped: "; python3 tools/loop_trace.py turns3 2>/dev/null
  echo -n "pipe_dev shipped: "; python3 tools/loop_trace.py pipe_dev 2>/dev/null
  for d in 0 1 2 3 4; do echo -n "dummy $d: "; SPEEDY_HIP_LIB=$T SPX_PIPE_DUMMY_STREAMS=$d python3 tools/loop_trace.py pipe_dev 2>/dev/null; done
  for pr in -1 1; do echo -n "prio $pr: "; SPEEDY_HIP_LIB=$T SPX_PIPE_PRIO=$pr python3 tools/loop_trace.py pipe_dev 2>/dev/null; done
  echo -n "null stream: "; SPEEDY_HIP_LIB=$T SPX_PIPE_NULL_STREAM=1 python3 tools/loop_trace.py pipe_dev 2>/dev/null
  echo -n "pipe_dev shipped again: "; python3 tools/loop_trace.py pipe_dev 2>/dev/null
} > "$OUT/${TAG}_queue_probe.txt" 2>&1
cat "$OUT/${TAG}_queue_probe.txt"
