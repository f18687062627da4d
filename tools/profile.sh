#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile.sh TAG): the evidence behind bench.py's roofline numbers.
#   1. rocprofv3 --kernel-trace --stats  of the default bench command  -> gpurun_out/TAG_stats/
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE (separate passes, no other trace domain)
#   3. the bench line itself
# Copy the summaries into profiles/ afterwards with tools/profile_collect.py TAG.
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline \
  > "$OUT/${TAG}_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/${TAG}_pmc_$c" -o pmc --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 \
    --no-cpu-baseline > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
ls -R "$OUT" | grep -i "${TAG}" | head -40
