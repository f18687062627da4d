#!/bin/bash
# GPU box: correctness of the walk kernels, then an A/B of kernel variants on the bench workload.
#   bash tools/ab_walk.sh TAG "VARIANT1;VARIANT2;..."     VARIANT = space-separated env assignments (may be empty)
TAG=${1:-ab}
VARIANTS=${2:-";SPX_WALK_OLD=1"}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sonic2.py -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1
tail -5 "$OUT/${TAG}_pytest.log"
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  for m in serial conc; do
    if [ $m = serial ]; then S="SPX_SERIAL=1"; else S=""; fi
    line=$(env $v $S timeout 300 python3 bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
    print('[%s] %-6s ms/step=%.3f walk=%.3f analysis=%.3f tension=%.3f' % ('$v', '$m', d['ms_per_step'], k['spx_walk_kernel'], k['spx_analysis_kernel'], k['spx_tension_kernel']))
except Exception as e:
    print('[%s] %s FAILED %s' % ('$v', '$m', e))
" | tee -a "$OUT/${TAG}_ab.txt"
  done
done
