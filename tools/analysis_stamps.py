"""Diagnostic only: cycle shares of the analysis kernel's phases (wave 0 of workgroup 7) from a -DSPX_STAMPS build
(speedy_amd/lib/stamps/libspeedy_hip_astamps.so), kernels run one at a time (spx_set_concurrent(0))."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SPEEDY_HIP_LIB", os.path.join(ROOT, "speedy_amd", "lib", "stamps", "libspeedy_hip_astamps.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

NAMES = ["phase 1: stage 1 (pre-emphasis, window, first radix)", "phase 1: remaining DFT stages", "phase 1: untangle + magnitudes",
         "phase 1: wave 0 done, (loop exit)", "barrier after phase 1", "phase 2: energy / max / inverse norm (one lane per frame)",
         "phases 3 + 4: gated log terms + float sums", "records stored (+ flag)"]
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n = 10 * rate
plan = Plan(rate, False)
plan.L.spx_set_concurrent(0)
base = [speech_like(n, rate, seed=i) for i in range(8)]
b = Batch(plan, [n] * 256, 1, 3.5, 1.0, 0.0)
b.upload([base[i % 8] for i in range(256)])
b.run()
torch.cuda.synchronize()
dbg = plan.L.spx_debug_astamps
dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
dbg(buf, 1)
b.run()
torch.cuda.synchronize()
dbg(buf, 1)
tot = sum(buf[i] for i in range(8))
print("rate %d: workgroup 7, wave 0: %d cycles for one tile" % (rate, tot))
for i in range(8):
    print("  %d %-58s %8d  %5.1f %%" % (i, NAMES[i], buf[i], 100.0 * buf[i] / max(1, tot)))
