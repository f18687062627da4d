"""Per-kernel durations and the period of a steady loop from a rocprofv3 --kernel-trace CSV (the last `n` launches of each kernel):
python tools/trace_summary.py kernel_trace.csv [n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
by = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    name = name[5:] if name.startswith("void ") else name
    name = name.split("(")[0]
    by[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
for name, v in sorted(by.items()):
    v.sort()
    v = v[-n:]
    dur = [(e - s) / 1e3 for s, e, _ in v]
    per = [(v[i + 1][0] - v[i][0]) / 1e3 for i in range(len(v) - 1)]
    q = sorted({x[2] for x in v})
    print("%-52s n %3d  dur avg %8.1f us (min %8.1f max %8.1f)  start-to-start avg %8.1f us  queues %s" % (
        name[:52], len(v), sum(dur) / len(dur), min(dur), max(dur), (sum(per) / len(per)) if per else 0.0, ",".join(q)))
# the last launches in start order (a call's timeline):  python tools/trace_summary.py kernel_trace.csv 30 --timeline 24
if "--timeline" in sys.argv:
    k = int(sys.argv[sys.argv.index("--timeline") + 1])
    allk = []
    for name, v in by.items():
        allk += [(s, e, q, name) for s, e, q in v]
    allk.sort()
    allk = allk[-k:]
    t0 = allk[0][0]
    for s, e, q, name in allk:
        print("%10.1f %10.1f %9.1f us  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:70]))
