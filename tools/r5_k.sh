#!/bin/bash
TAG=${1:-r5k}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 2400 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_committed_kernel_resources_are_the_librarys > "$OUT/${TAG}_pytest_full.log" 2>&1; tail -6 "$OUT/${TAG}_pytest_full.log"
python3 tools/scale_configs.py > "$OUT/${TAG}_scale_configs.txt" 2>&1; cat "$OUT/${TAG}_scale_configs.txt"
{ echo "# pipeline (resident input, device out), 512 streams per batch: split into two overlapping sub-batches (shipped)"; SPX_PROBE_STREAMS=512 python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
  echo "# the same as ONE call per batch (tuning build, SPX_SPLIT_MAX=1)"; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so SPX_SPLIT_MAX=1 SPX_PROBE_STREAMS=512 python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
  echo "# 384 streams"; SPX_PROBE_STREAMS=384 python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
  SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_tuning.so SPX_SPLIT_MAX=1 SPX_PROBE_STREAMS=384 python3 tools/loop_trace.py pipe_dev 30 2>/dev/null
} | tee "$OUT/${TAG}_split_overlapped.txt"
python3 tools/kernel_resources.py "$OUT/kernel_resources.json" > /dev/null
