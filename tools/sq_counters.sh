#!/bin/bash
# SQ instruction / wait counters of the kernels (two PMC passes, kernel-trace only).  Output: gpurun_out/TAG_sq{1,2}/
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
# --serial: kernels one after the other, so that the counters per kernel are not blurred by sharing
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/${TAG}_sq1" -o pmc --output-format csv -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/${TAG}_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
  -d "$OUT/${TAG}_sq2" -o pmc --output-format csv -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/${TAG}_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SENDMSG \
  -d "$OUT/${TAG}_sq3" -o pmc --output-format csv -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates > "$OUT/${TAG}_sq3.log" 2>&1
# ... and the kernels of the PIPELINED timed loop (the lean walk form among them; no polling kernels there, so the profiler's
# serialisation is harmless): the instruction counts behind roofline.valu_issue
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/${TAG}_sq4" -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-api --no-config4 --no-large-batch --no-other-rates --no-unpipelined > "$OUT/${TAG}_sq4.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, tag + "_sq*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        # the kernel's own name with its template arguments (no "void", no argument list): instantiations are kept apart
        key = name[5:] if name.startswith("void ") else name
        depth = 0
        for pos, c in enumerate(key):
            depth += (c == "<") - (c == ">")
            if c == "(" and depth == 0:
                key = key[:pos]
                break
        if key.startswith(("spx_analysis_kernel", "spx_tension_kernel", "spx_walk")):
            acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump({"note": "rocprofv3 --kernel-trace --pmc (three passes of SQ counters with bench.py --serial --steps 2 --warmup 1, one of the "
                   "pipelined loop with --no-unpipelined: the lean walk form); averages per launch; 256 streams x 10 s; tools/sq_counters.sh",
           "counters": res},
          open(os.path.join(out, tag + "_sq_counters.json"), "w"), indent=1)
for k, d in res.items():
    print(k, {c: int(v) for c, v in sorted(d.items())})
PY
