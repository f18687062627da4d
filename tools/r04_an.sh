#!/bin/bash
# round 4: the compiled-in analysis kernels of 44.1 / 48 kHz -- parity tests touching those rates, then per-kind timings
tag=${1:-r04u}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_speedy_unit.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_an.log
python tools/scale_configs.py >> gpurun_out/${tag}_an.log 2>&1
tail -22 gpurun_out/${tag}_an.log
