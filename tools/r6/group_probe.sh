#!/bin/bash
# GPU box: many walk launches in flight in the throughput form, no gate -- does the pipelined loop of 256-stream batches approach the
# large-batch regime (2 048 streams in one call: 0.79 ms per 256 streams) when the hardware is left to mix the kernels?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6i}
OUT=$PWD/gpurun_out; mkdir -p $OUT
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
export GPU_MAX_HW_QUEUES=16
run() { echo -n "$1: "; shift; env "$@" python3 tools/loop_trace.py pipe_dev 200 2>&1 | tail -1; }
{
  run "shipped" X=1
  for ws in 4 6 8; do for d in 8 12 16; do
    run "throughput form, $ws walk streams, depth $d, gate" SPEEDY_HIP_LIB=$T SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 SPX_WALK_STREAMS=$ws SPX_PROBE_DEPTH=$d
    run "throughput form, $ws walk streams, depth $d, NO gate" SPEEDY_HIP_LIB=$T SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 SPX_WALK_STREAMS=$ws SPX_PROBE_DEPTH=$d SPX_NO_GATE=1
  done; done
  run "lean form, 4 walk streams, depth 8, NO gate" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=4 SPX_PROBE_DEPTH=8 SPX_NO_GATE=1
  run "lean form, 2 walk streams, depth 4, NO gate" SPEEDY_HIP_LIB=$T SPX_NO_GATE=1
} | tee $OUT/${TAG}_group_probe.txt
