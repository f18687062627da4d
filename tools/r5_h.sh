#!/bin/bash
TAG=${1:-r5h}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
{ for r in 1 2; do for d in 4 5 6 8; do echo -n "depth $d: "; SPX_PROBE_DEPTH=$d python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done; done
  for w in 8 16 32 64 128 256; do echo -n "pack wgs $w depth 6: "; SPEEDY_HIP_LIB=$T SPX_PIPE_PACK_WGS=$w SPX_PROBE_DEPTH=6 python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done
  for w in 16 64; do echo -n "pack wgs $w depth 4: "; SPEEDY_HIP_LIB=$T SPX_PIPE_PACK_WGS=$w SPX_PROBE_DEPTH=4 python3 tools/loop_trace.py pipe_host 200 2>/dev/null; done
} 2>&1 | tee "$OUT/${TAG}_host_sweep.txt"
