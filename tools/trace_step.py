"""Timeline of the last step in a rocprofv3 --kernel-trace CSV: kernel, start and end in microseconds relative to the
step's first kernel.  Usage: python tools/trace_step.py kernel_trace.csv [n_last_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"]
    name = name[5:] if name.startswith("void ") else name
    print("%9.1f %9.1f  %7.1f us  grid %-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                 (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                 r.get("Grid_Size_X", r.get("Grid_Size", "?")), name[:70]))
