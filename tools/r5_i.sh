#!/bin/bash
TAG=${1:-r5i}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python3 tools/kernel_resources.py "$OUT/kernel_resources.json" > /dev/null; cp "$OUT/kernel_resources.json" profiles/kernel_resources.json
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_speedy_unit.py tests/test_gpu_fuzz.py tests/test_gpu_sonic2.py -m gpu -x -q > "$OUT/${TAG}_pytest_dftv2.log" 2>&1; tail -4 "$OUT/${TAG}_pytest_dftv2.log"
{ for r in 1 2 3; do
    echo -n "dft v2 (shipped): "; python3 tools/analysis_time.py 16000 22050 48000 2>/dev/null | tr '\n' ' '; echo
    echo -n "dft v1 (dftv1):   "; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_dftv1.so python3 tools/analysis_time.py 16000 22050 48000 2>/dev/null | tr '\n' ' '; echo
  done
} > "$OUT/${TAG}_dft_ab.txt" 2>&1
cat "$OUT/${TAG}_dft_ab.txt"
for i in 1 2; do python3 tools/loop_trace.py turns3 2>/dev/null; python3 tools/loop_trace.py pipe_dev 2>/dev/null; done | tee "$OUT/${TAG}_loops.txt"
