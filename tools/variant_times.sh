#!/bin/bash
# GPU box: the plain call's walk kernel (concurrent mode, full form) and the pipelined loop (lean form) for the shipped library and
# every variant library named on the command line (speedy_amd/lib/ab/libspeedy_hip_<NAME>.so), two rounds interleaved.
#   bash tools/variant_times.sh xf1 pad0 pad1 ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
CODE='import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_perf_guard import measure_walk_ms, measure_pipelined
(a, b), k = measure_walk_ms(); p, kp = measure_pipelined()
print("plain-call walk %.3f / %.3f ms   pipelined step %.3f / %.3f ms (lean walk %.3f)" % (a, b, p[0][0], p[1][0], min(p[0][1], p[1][1])))'
for r in 1 2; do
  for v in shipped "$@"; do
    if [ $v = shipped ]; then L=""; else L=$PWD/speedy_amd/lib/ab/libspeedy_hip_$v.so; fi
    echo -n "$v: "; SPEEDY_HIP_LIB=$L python3 -c "$CODE" 2>/dev/null | tail -1
  done
done
