#!/bin/bash
# which engine moves the PCIe leg's data and how the leg's kernels line up: rocprofv3 kernel trace + memory-copy trace of
# bench.py's pcie leg only.   SPX_BENCH_PCIE=ahead SPX_BENCH_PCIE_NBUF=4 SPX_BENCH_PCIE_LAG=2 bash tools/pcie_trace.sh TAG
TAG=${1:-pcie_trace}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/$TAG" -o t --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined > "$OUT/$TAG.log" 2>&1
python3 - "$OUT/$TAG" <<'PY'
import csv, sys, os
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(os.path.join(d, "t_kernel_trace.csv"))):
    n = r["Kernel_Name"]
    short = "walk" if "walk_fast" in n else "analysis" if "analysis" in n else "tension" if "tension" in n else "blit" if "copyBuffer" in n else \
            "gate" if "gate" in n else "pack" if "pack" in n else "stage" if "stage" in n else None
    if short:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
for r in csv.DictReader(open(os.path.join(d, "t_memory_copy_trace.csv"))):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "H2D" if "HOST_TO_DEVICE" in r["Direction"] else "D2H"))
ev.sort()
t0 = ev[0][0]
tail = [e for e in ev if e[1] - e[0] > 30000][-60:]
for s, e, n in tail:
    print("%9.3f ms  +%7.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
PY
