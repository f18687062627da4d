#!/bin/bash
# HBM traffic counters of the kernels of the PIPELINED timed loop (the lean walk form among them), two PMC passes.  The
# pipelined order has no polling kernels (its gate kernel is bounded), so it is safe under the profiler's kernel serialisation;
# --no-unpipelined keeps the concurrent mode's polling kernels out of the run.
TAG=${1:-pmcp}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d "$OUT/${TAG}_pmc_$c" -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 \
    --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined > "$OUT/${TAG}_pmc_$c.log" 2>&1
  echo "$c rc $?"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = {}
    for f in glob.glob(os.path.join(out, "%s_pmc_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "spx_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, "launches", len(v), "avg KB %.0f" % (sum(v) / len(v)))
PY
