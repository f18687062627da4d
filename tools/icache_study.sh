#!/bin/bash
# Instruction-cache evidence for the link-order effect (DESIGN.md 2 "one device code object"): the bench step with the
# shipped library (one code object for the three step kernels) and with separate code objects in two link orders --
# round 2's order and the order that cost 26 % then -- timed, then with SQC instruction-cache counters (their own pass).
#   gpurun -- bash tools/icache_study.sh      (the variant libraries are built by the commands in profiles/r03/README)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r03t_icache; mkdir -p $OUT
B="bench.py --no-cpu-baseline --no-pcie --no-api --no-config4"
P='import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step %.3f" % d["ms_per_step"], {k.split("<")[0]: round(v,3) for k,v in d["roofline"]["kernel_ms_per_step"].items()})'
rocprofv3 --list-avail 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/counters_available.txt
for v in shipped sep_r02order sep_slow; do
  L=""; [ $v != shipped ] && L=$PWD/speedy_amd/lib/ab/libspeedy_hip_$v.so
  for rep in 1 2; do echo -n "$v: "; SPEEDY_HIP_LIB=$L python3 $B 2>/dev/null | python3 -c "$P"; done
  SPEEDY_HIP_LIB=$L rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $OUT/$v -o pmc --output-format csv -- python3 $B --steps 6 --warmup 2 > $OUT/$v.log 2>&1
  python3 - "$v" $(find $OUT/$v -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
v, f = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for key in ("spx_walk_fast_kernel", "spx_analysis_kernel", "spx_tension_kernel"):
        if key in k:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in acc.items():
    m = {n: sum(x) / len(x) for n, x in c.items()}
    req = m.get("SQC_ICACHE_REQ", 0.0)
    print("  %-13s %-22s per launch: req %.3e  hits %.3e  misses %.3e  miss rate %.4f" % (v, key, req, m.get("SQC_ICACHE_HITS", 0), m.get("SQC_ICACHE_MISSES", 0), m.get("SQC_ICACHE_MISSES", 0) / req if req else 0))
PY
done 2>&1 | tee $OUT/summary.txt
cat $OUT/counters_available.txt
