"""The analysis kernel on its own (kernels in sequence), 256 streams x 10 s, per sample rate: ms per launch from the engine's
own HIP events.  python tools/analysis_time.py [rates...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

rates = [int(v) for v in sys.argv[1:]] or [16000, 22050, 44100, 48000]
for rate in rates:
    n = 10 * rate
    plan = Plan(rate, False)
    plan.L.spx_set_concurrent(0)
    base = [speech_like(n, rate, seed=i) for i in range(4)]
    b = Batch(plan, [n] * 256, 1, 3.5, 1.0, 0.0)
    b.upload([base[i % 4] for i in range(256)])
    for _ in range(3):
        b.run()
    torch.cuda.synchronize()
    plan.L.spx_set_timing(1)
    reps = 5
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    plan.L.spx_set_timing(0)
    sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
    plan.L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
    print("rate %5d: analysis %.3f ms  walk %.3f ms  (kernels in sequence, 256 x 10 s mono)" % (rate, sa.value / reps, sw.value / reps))
    del b
    torch.cuda.empty_cache()
