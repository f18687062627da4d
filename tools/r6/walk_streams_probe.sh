#!/bin/bash
# GPU box: the pipelined loop of the bench batch (tools/loop_trace.py pipe_dev) with the walk kernel's form and the number of walk
# streams varied through the tuning library's switches.   bash tools/r6/walk_streams_probe.sh TAG [STEPS]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6a}; STEPS=${2:-100}
OUT=gpurun_out; mkdir -p $OUT
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
run() { echo -n "$1: "; shift; env "$@" python3 tools/loop_trace.py pipe_dev $STEPS 2>&1 | tail -1; }
{
for r in 1 2; do
  run "shipped (lean 4+0, two walk streams)" X=1
  run "tuning, lean 4+0, three walk streams, depth 6" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=3 SPX_PROBE_DEPTH=6
  for ws in 2 3 4; do
    run "tuning, throughput form 2+0 (1536-frame window), $ws walk streams, depth 6" SPEEDY_HIP_LIB=$T SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 SPX_WALK_STREAMS=$ws SPX_PROBE_DEPTH=6
  done
  run "tuning, throughput form, 4 walk streams, depth 8" SPEEDY_HIP_LIB=$T SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 SPX_WALK_STREAMS=4 SPX_PROBE_DEPTH=8
  run "tuning, throughput form, 3 walk streams, depth 4" SPEEDY_HIP_LIB=$T SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 SPX_WALK_STREAMS=3 SPX_PROBE_DEPTH=4
done
} | tee $OUT/${TAG}_walk_streams.txt
