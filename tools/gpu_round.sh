#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/gpu_round.sh TAG [quick]): parity tests, the bench line at N = 1 and with two
# self-spawned ranks, kernel stats, and the serial-mode PMC passes behind roofline.traffic.
TAG=${1:-r03}
MODE=${2:-full}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
if [ "$MODE" = full ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1; tail -3 "$OUT/${TAG}_pytest.log"
else
  timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1; tail -3 "$OUT/${TAG}_pytest.log"
fi
timeout 600 python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; cat "$OUT/${TAG}_bench.json"
timeout 600 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-api --no-config4 > "$OUT/${TAG}_bench_2rank_1gpu.json" 2> "$OUT/${TAG}_bench_2rank.err"
cat "$OUT/${TAG}_bench_2rank_1gpu.json"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined \
  > "$OUT/${TAG}_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$OUT/${TAG}_pmc_$c" -o pmc --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 \
    --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
ls "$OUT" | grep "${TAG}" | head -40
