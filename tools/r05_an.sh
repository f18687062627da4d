#!/bin/bash
tag=${1:-r05}
mkdir -p gpurun_out
out=gpurun_out/${tag}_an.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_speedy_unit.py -m gpu -x -q 2>&1 | tail -6 > $out
python tools/analysis_time.py 8000 16000 24000 32000 48000 >> $out 2>&1
cat $out
