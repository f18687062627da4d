#!/bin/bash
# Tuning sweep on the GPU box: waves per stream in the walk kernel x (concurrent | serial) execution.
for nw in 2 4 8 16; do
  for mode in conc serial; do
    if [ $mode = serial ]; then export SPX_SERIAL=1; else unset SPX_SERIAL; fi
    SPX_WALK_NW=$nw python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('nw=$nw $mode ms/step=%.3f walk=%.3f analysis=%.3f' % (d['ms_per_step'], k['spx_walk_kernel'], k['spx_analysis_kernel']))"
  done
done
