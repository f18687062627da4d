#!/bin/bash
# A/B two builds of the library on the same GPU box: tools/ab.sh lib_a.so lib_b.so  (serial mode, walk kernel ms)
for rep in 1 2; do for l in "$@"; do
  SPX_SERIAL=1 SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/$l python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('$l walk=%.3f analysis=%.3f' % (k['spx_walk_kernel'], k['spx_analysis_kernel']))"
done; done
