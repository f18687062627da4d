#!/bin/bash
# Round-5 probes (gpurun -- bash tools/r5_probe.sh TAG): stream creation order, host topology, log spec v1 / v2 A/B.
TAG=${1:-r5c}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
{ lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
  numactl -H 2>/dev/null | head -12; rocm-smi --showtoponuma 2>/dev/null | head -20; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4
  grep -E "Mems_allowed_list|Cpus_allowed_list" /proc/self/status; } > "$OUT/${TAG}_topology.txt" 2>&1
cat "$OUT/${TAG}_topology.txt"
{ for o in lib_first pipe_first torch_first; do timeout 300 python3 tools/order_probe.py $o 4 2>/dev/null; done
  timeout 300 python3 tools/order_probe.py lib_first 3 2>/dev/null
  SPX_PROBE_QUEUES=4 timeout 300 python3 tools/order_probe.py lib_first 4 2>/dev/null
  SPX_PROBE_QUEUES=4 timeout 300 python3 tools/order_probe.py pipe_first 4 2>/dev/null
} > "$OUT/${TAG}_order_probe.txt" 2>&1
cat "$OUT/${TAG}_order_probe.txt"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_speedy_unit.py -m gpu -x -q > "$OUT/${TAG}_pytest_logv2.log" 2>&1; tail -3 "$OUT/${TAG}_pytest_logv2.log"
{ for r in 1 2 3; do
    echo -n "v2 (shipped): "; python3 tools/analysis_time.py 16000 22050 2>/dev/null | tr '\n' ' '; echo
    echo -n "v1 (logv1):   "; SPEEDY_HIP_LIB=$PWD/speedy_amd/lib/ab/libspeedy_hip_logv1.so python3 tools/analysis_time.py 16000 22050 2>/dev/null | tr '\n' ' '; echo
  done
  bash tools/ab_variants.sh 2 logv1
} > "$OUT/${TAG}_log_ab.txt" 2>&1
cat "$OUT/${TAG}_log_ab.txt"
