#!/bin/bash
# GPU box: the driver's window (bench.py --steps 20 --warmup 5) against the number of walk streams and buffer sets (tuning library) --
# the 20-step window pays fill and drain (0.97 ms per step where 100 steps read 0.91): does a third walk launch in flight shorten them?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=${1:-r6m}
OUT=$PWD/gpurun_out; mkdir -p $OUT
T=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so
Q="--no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined"
run() { echo -n "$1: "; shift; env "$@" python3 bench.py --steps ${STEPS:-20} --warmup 5 $Q 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms per step, %.0f Msamples/s' % (d['ms_per_step'], d['value']))"; }
{
  for rep in 1 2; do
    run "shipped library" X=1
    run "tuning, 2 walk streams, 4 buffer sets" SPEEDY_HIP_LIB=$T
    run "tuning, 3 walk streams, 4 buffer sets" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=3
    run "tuning, 3 walk streams, 5 buffer sets" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=3 SPX_BENCH_DEPTH=5
    run "tuning, 4 walk streams, 6 buffer sets" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=4 SPX_BENCH_DEPTH=6
    run "tuning, 2 walk streams, 6 buffer sets" SPEEDY_HIP_LIB=$T SPX_BENCH_DEPTH=6
  done
  STEPS=100 run "100 steps: shipped" X=1
  STEPS=100 run "100 steps: tuning, 3 walk streams, 5 buffer sets" SPEEDY_HIP_LIB=$T SPX_WALK_STREAMS=3 SPX_BENCH_DEPTH=5
} | tee $OUT/${TAG}_window20.txt
