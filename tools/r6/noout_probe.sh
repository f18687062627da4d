#!/bin/bash
# GPU box: the pipelined loop with a variant library against the shipped one (two rounds interleaved), then a kernel trace of each.
#   bash tools/r6/noout_probe.sh TAG VARIANT...      (speedy_amd/lib/ab/libspeedy_hip_<VARIANT>.so)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
for r in 1 2; do
  echo -n "shipped: "; python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1
  for v in "$@"; do
    L=$PWD/speedy_amd/lib/ab/libspeedy_hip_$v.so
    echo -n "$v: "; SPEEDY_HIP_LIB=$L python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1
    echo -n "$v, three walk streams (depth 4): "; SPEEDY_HIP_LIB=$L SPX_WALK_STREAMS=3 python3 tools/loop_trace.py pipe_dev 100 2>&1 | tail -1
  done
done
} | tee $OUT/${TAG}_variants.txt
for v in shipped "$@"; do
  if [ $v = shipped ]; then L=""; else L=$PWD/speedy_amd/lib/ab/libspeedy_hip_$v.so; fi
  rm -rf $OUT/${TAG}_trace_$v
  SPEEDY_HIP_LIB=$L rocprofv3 --kernel-trace -d $OUT/${TAG}_trace_$v -o t --output-format csv -- python3 tools/loop_trace.py pipe_dev 40 > $OUT/${TAG}_trace_$v.log 2>&1
  f=$(find $OUT/${TAG}_trace_$v -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_summary.py $f 30 --timeline 40 > $OUT/${TAG}_trace_${v}_summary.txt 2>&1
  rm -rf $OUT/${TAG}_trace_$v
  head -12 $OUT/${TAG}_trace_${v}_summary.txt
done
