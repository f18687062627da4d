#!/bin/bash
# The evidence of a round's final build, ONE GPU call (gpurun -- bash tools/collect_round.sh TAG): everything lands in
# gpurun_out/TAG/ ready to be copied into profiles/<round>/ (and pmc_traffic.json / sq_counters.json / perf_reference.json /
# kernel_resources.json into profiles/):
#   bench.json, bench_steps20.json          the default bench command and the driver's (--steps 20 --warmup 5)
#   kernel_stats_pipelined_loop.csv         rocprofv3 --kernel-trace --stats of the timed loop (its average launch of the walk
#                                           kernel must agree with roofline.kernel_ms_per_launch of the line)
#   pmc_traffic.json                        HBM bytes per launch: --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernels in sequence
#                                           (--serial) and the pipelined loop's lean walk form; FETCH_SIZE doubled (gfx950)
#   sq_counters.json                        SQ instruction / wait / LDS counters per kernel (tools/sq_counters.sh)
#   perf_reference.json, kernel_resources.json, scale_streams.txt, scale_configs.txt
TAG=${1:-r6z}
PARTS=${2:-bench,stats,pmc,sq,refs,scale}     # second argument: only these parts (comma-separated)
has() { case ",$PARTS," in *",$1,"*) return 0;; esac; return 1; }
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out; D=$OUT/$TAG; mkdir -p "$D"
Q="--no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates"
if has bench; then
python3 bench.py > "$D/bench.json" 2> "$D/bench.err"; echo "bench rc $?"; head -c 600 "$D/bench.json"; echo
python3 bench.py --steps 20 --warmup 5 > "$D/bench_steps20.json" 2>> "$D/bench.err"; echo "bench steps20 rc $?"
fi
if has stats; then
timeout 600 rocprofv3 --kernel-trace --stats -d "$D/stats" -o stats --output-format csv -- python3 bench.py $Q --no-unpipelined > "$D/stats.log" 2>&1
cp $(find "$D/stats" -name "*kernel_stats.csv" | head -1) "$D/kernel_stats_pipelined_loop.csv" 2>/dev/null; head -6 "$D/kernel_stats_pipelined_loop.csv"
fi
if has pmc; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$D/dir_pmc_$c" -o pmc --output-format csv -- python3 bench.py --serial --steps 3 --warmup 1 $Q > "$D/pmc_$c.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$D/dir_pmcp_$c" -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 $Q --no-unpipelined > "$D/pmcp_$c.log" 2>&1
done
fi
has sq && { bash tools/sq_counters.sh $TAG/sq > "$D/sq.log" 2>&1; cp "$OUT/$TAG/sq_sq_counters.json" "$D/sq_counters.json" 2>/dev/null; rm -f "$OUT/$TAG/sq_sq_counters.json"; }
if has refs; then
python3 tools/perf_reference.py > "$D/perf_reference.log" 2>&1; cp "$OUT/perf_reference.json" "$D/perf_reference.json"
python3 tools/kernel_resources.py > "$D/kernel_resources.log" 2>&1; cp profiles/kernel_resources.json "$D/kernel_resources.json"
fi
if has scale; then
timeout 600 python3 tools/scale_streams.py 256 512 1024 2048 > "$D/scale_streams.txt" 2>&1
timeout 900 python3 tools/scale_configs.py > "$D/scale_configs.txt" 2>&1
fi
has pmc && python3 - "$D" <<'PY'
import csv, glob, json, os, sys
D = sys.argv[1]
traffic = {}
def key_of(name):
    key = name[5:] if name.startswith("void ") else name
    depth = 0
    for pos, c in enumerate(key):
        depth += (c == "<") - (c == ">")
        if c == "(" and depth == 0:
            return key[:pos]
    return key
for sub, what in (("pmc", "kernels in sequence (bench.py --serial --steps 3 --warmup 1): one kernel in flight at a time"),
                  ("pmcp", "the pipelined loop through the pipeline object (--no-unpipelined: its walk kernel in the lean form; no polling kernels)")):
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(D, "dir_%s_%s" % (sub, counter), "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for row in csv.DictReader(open(f)):
                k = key_of(row["Kernel_Name"])
                if k.startswith(("spx_analysis_kernel", "spx_tension_kernel", "spx_walk")) and row["Counter_Name"] == counter:
                    acc.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                if sub == "pmcp" and k in traffic and "source" in traffic[k] and "sequence" in traffic[k]["source"] and counter + "_KB" in traffic[k]:
                    continue   # (the serial pass is the reference for kernels both runs launch)
                t = traffic.setdefault(k, {})
                t[counter + "_KB"] = sum(v) / len(v)
                t["source"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), " + what
for k, t in traffic.items():
    t["hbm_bytes_per_launch"] = int(2 * t.get("FETCH_SIZE_KB", 0) * 1024 + t.get("WRITE_SIZE_KB", 0) * 1024)
    t["correction"] = "FETCH_SIZE doubled (gfx950 reports half the bytes of a coalesced streaming read, MI355X_MICROARCH.md HBM section); WRITE_SIZE used as is."
traffic["_note"] = "HBM bytes per launch, tools/collect_round.sh (one build, one GPU call)"
json.dump(traffic, open(os.path.join(D, "pmc_traffic.json"), "w"), indent=1)
print({k: v.get("hbm_bytes_per_launch") for k, v in traffic.items() if k != "_note"})
PY
rm -rf "$D"/dir_pmc* "$D/stats" "$OUT/$TAG"/sq_sq[0-9]*
ls -la "$D"
