"""Diagnostic only: per-phase shader-cycle shares of the walk kernel (workgroup 0), from the -DSPX_STAMPS build.
Usage on the GPU box:  SPEEDY_HIP_LIB=speedy_amd/lib/libspeedy_hip_stamps.so python tools/walk_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

NAMES = {0: "event loop (between process calls)", 1: "pre-step (copy steps, loop ctl)", 2: "ensure_window",
         3: "phase B build signals", 4: "sync c", 5: "coarse accumulate", 6: "sync e", 7: "coarse select",
         8: "refine accumulate", 9: "sync h", 10: "refine select", 11: "decision", 12: "after OLA -> end of process",
         13: "overlap-add"}
rate, n, nstreams = 16000, 160000, int(os.environ.get("NSTREAMS", "256"))
plan = Plan(rate, False)
base = [speech_like(n, rate, seed=i) for i in range(8)]
b = Batch(plan, [n] * nstreams, 1, 3.5, 1.0, 0.0)
b.upload([base[i % 8] for i in range(nstreams)])
b.run()
torch.cuda.synchronize()
L = plan.L
L.spx_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
L.spx_debug_stamps(buf, 1)
b.run()
torch.cuda.synchronize()
L.spx_debug_stamps(buf, 1)
tot = sum(buf)
print("stream 0: total stamped cycles %d" % tot)
for i in range(14):
    print("  %2d %-40s %10d  %5.1f %%" % (i, NAMES.get(i, ""), buf[i], 100.0 * buf[i] / max(1, tot)))

# analysis kernel, workgroup 7, lane 0
try:
    L.spx_debug_astamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    ab = (C.c_ulonglong * 16)()
    L.spx_debug_astamps(ab, 1)
    b.run()
    torch.cuda.synchronize()
    L.spx_debug_astamps(ab, 1)
    an = ["frame load+window", "DFT stages", "untangle+mag", "loop exit", "sync", "phase2 energy (+sync)",
          "phase3 log terms (+sync)", "phase4 accumulate"]
    tot = sum(ab[:8])
    print("analysis tile (wave 0 of workgroup 7): total %d cycles" % tot)
    for i in range(8):
        print("  %d %-28s %9d  %5.1f %%" % (i, an[i], ab[i], 100.0 * ab[i] / max(1, tot)))
except AttributeError:
    pass
