#!/bin/bash
# The round's final collection (gpurun -- bash tools/r5_final.sh TAG): bench line, PCIe leg x10, references, counters, kernel stats, PMC traffic.
TAG=${1:-r5m}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
bash tools/r5_round.sh $TAG bench,pcie,refs,sq 2>&1 | tail -60
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined > "$OUT/${TAG}_stats.log" 2>&1
cp $(find "$OUT/${TAG}_stats" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats_pipelined_loop.csv" 2>/dev/null; head -8 "$OUT/${TAG}_kernel_stats_pipelined_loop.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$OUT/${TAG}_pmc_$c" -o pmc --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 \
    --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
bash tools/pmc_pipelined.sh ${TAG}p 2>&1 | tail -8
timeout 600 python3 tools/scale_streams.py 256 512 1024 2048 > "$OUT/${TAG}_scale_streams.txt" 2>&1; cat "$OUT/${TAG}_scale_streams.txt"
