#!/bin/bash
# round 4: phase stamps of the analysis kernel at three rates, parity of the analysis half, per-kind timings
tag=${1:-r04v}
mkdir -p gpurun_out
out=gpurun_out/${tag}_an.log
: > $out
for r in 16000 48000 44100; do
  SPEEDY_HIP_LIB=speedy_amd/lib/ab/libspeedy_hip_astamps.so python tools/analysis_stamps.py $r >> $out 2>&1
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_speedy_unit.py -m gpu -x -q 2>&1 | tail -5 >> $out
python tools/analysis_time.py >> $out 2>&1
python tools/scale_configs.py >> $out 2>&1
cat $out
