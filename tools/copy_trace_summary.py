"""Kernel + memory-copy timeline of the last `n` events of a rocprofv3 --kernel-trace --memory-copy-trace run.
python tools/copy_trace_summary.py DIR [n]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), name[:44])))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s bytes" % (r.get("Direction", r.get("Kind", "?")), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
copies = [e for e in ev if e[2].startswith("COPY") and (e[1] - e[0]) > 500000]
if len(copies) > 4:
    c = copies[-40:]
    dur = [(b - a) / 1e3 for a, b, _ in c]
    gap = [(c[i + 1][0] - c[i][1]) / 1e3 for i in range(len(c) - 1)]
    print("big copies: n %d  duration avg %.1f us (min %.1f max %.1f)  gap between copies avg %.1f us (min %.1f max %.1f)  start-to-start %.1f us"
          % (len(c), sum(dur) / len(dur), min(dur), max(dur), sum(gap) / len(gap), min(gap), max(gap), (c[-1][0] - c[0][0]) / 1e3 / (len(c) - 1)))
last = ev[-n:]
t0 = last[0][0]
for a, b, name in last:
    print("%9.1f %9.1f %8.1f us  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, name))
