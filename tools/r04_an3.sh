#!/bin/bash
tag=${1:-r04y}
mkdir -p gpurun_out
out=gpurun_out/${tag}_an.log
: > $out
for r in 16000 48000 44100; do
  SPEEDY_HIP_LIB=speedy_amd/lib/ab/libspeedy_hip_astamps.so python tools/analysis_stamps.py $r >> $out 2>&1
done
cat $out
