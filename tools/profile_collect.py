"""Copy the rocprofv3 summaries of a tools/profile.sh run from gpurun_out/ into profiles/ and derive
profiles/pmc_traffic.json (HBM bytes per launch of each kernel, corrected as MI355X_MICROARCH.md prescribes).
Usage: python tools/profile_collect.py TAG [ROUND_DIR]     e.g.  python tools/profile_collect.py v6 r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


stats = find("%s_stats/**/*kernel_stats.csv" % tag)
if stats:
    shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
bench = os.path.join(src, "%s_bench.json" % tag)
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, "%s_bench.json" % tag))
traffic = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("%s_pmc_%s/**/*counter_collection.csv" % (tag, counter))
    if not f:
        continue
    shutil.copy(f, os.path.join(dst, "%s_pmc_%s.csv" % (tag, counter)))
    acc = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        # the kernel's own name with its template arguments, as spx_batch_kernel_names() spells it: no "void", no argument list
        key = name[5:] if name.startswith("void ") else name
        depth = 0
        for pos, c in enumerate(key):
            depth += (c == "<") - (c == ">")
            if c == "(" and depth == 0:
                key = key[:pos]
                break
        if not key.startswith(("spx_analysis_kernel", "spx_tension_kernel", "spx_walk")):
            key = None
        if key and row["Counter_Name"] == counter:
            acc.setdefault(key, []).append(float(row["Counter_Value"]))
    for key, vals in acc.items():
        traffic.setdefault(key, {})[counter + "_KB"] = sum(vals) / len(vals)
for key, t in traffic.items():
    # gfx950: FETCH_SIZE reports half the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section)
    t["hbm_bytes_per_launch"] = int(2 * t.get("FETCH_SIZE_KB", 0) * 1024 + t.get("WRITE_SIZE_KB", 0) * 1024)
    t["correction"] = ("FETCH_SIZE doubled (gfx950 reports half the bytes of a coalesced streaming read, "
                       "MI355X_MICROARCH.md HBM section); WRITE_SIZE used as is.")
    t["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                   "--serial --steps 3 --warmup 1 --no-cpu-baseline --no-pcie (one kernel in flight at a time); "
                   "profiles/%s/%s_pmc_*.csv" % (rnd, tag))
if traffic:
    traffic["_note"] = ("HBM bytes per launch from rocprofv3 PMC passes in serial mode (bench.py --serial); tools/profile_collect.py %s %s" % (tag, rnd))
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("collected into", dst, "kernels:", sorted(traffic))
