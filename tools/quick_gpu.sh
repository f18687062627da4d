#!/bin/bash
# Developer loop on the GPU box: parity + fuzz, then the bench in serial and concurrent mode.
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sonic2.py -m gpu -q -x 2>&1 | tail -4
for m in serial conc; do
  if [ $m = serial ]; then export SPX_SERIAL=1; else unset SPX_SERIAL; fi
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('$m ms/step=%.3f walk=%.3f analysis=%.3f' % (d['ms_per_step'], k['spx_walk_kernel'], k['spx_analysis_kernel']))"
done
