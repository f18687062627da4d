#!/bin/bash
# kernel stats of the timed loop only (what the bench line's kernel_avg_launch_ms must agree with)
TAG=${1:-r05s}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined > "$OUT/${TAG}_stats.log" 2>&1
tail -1 "$OUT/${TAG}_stats.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_avg_launch_ms'])"
head -4 "$OUT"/${TAG}_stats/*kernel_stats.csv | cut -c1-60,150-260
