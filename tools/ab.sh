#!/bin/bash
# A/B builds of the library on the same GPU box: tools/ab.sh [-c] lib_a.so lib_b.so ...
# default: sequential launches, prints the walk kernel's ms; -c: concurrent mode, prints ms per step too.
MODE=serial
if [ "$1" = "-c" ]; then MODE=conc; shift; fi
for rep in 1 2; do for l in "$@"; do
  if [ $MODE = serial ]; then export SPX_SERIAL=1; else unset SPX_SERIAL; fi
  SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/$l python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('$l $MODE ms/step=%.3f walk=%.3f analysis=%.3f tension=%.3f' % (d['ms_per_step'], k['spx_walk_kernel'], k['spx_analysis_kernel'], k['spx_tension_kernel']))"
done; done
