"""Throughput against the number of streams in one batch call on one GPU (16 kHz mono, 10 s, 3.5x nonlinear)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

rate, n = 16000, 160000
plan = Plan(rate, False)
chunks = int(os.environ.get("SPX_CHUNKS", "0"))  # 0 = leave the engine's own choice
if chunks:
    plan.L.spx_set_pipeline_chunks(chunks)
base = [speech_like(n, rate, seed=i) for i in range(32)]
for ns in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048]:
    b = Batch(plan, [n] * ns, 1, 3.5, 1.0, 0.0)
    b.upload([base[i % 32] for i in range(ns)])
    for _ in range(2):
        b.run()
    torch.cuda.synchronize()
    plan.L.spx_set_timing(1)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    plan.L.spx_set_timing(0)
    sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
    plan.L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
    k = max(1, nc.value)
    import zlib
    import numpy as np
    outs = b.results()
    crc = zlib.crc32(np.concatenate(outs[:40]).tobytes())   # the same for every kernel variant / mode
    print("chunks=%d streams=%5d  %.3f ms/call  %.0f Msamples/s   (kernel sums per call: analysis %.2f ms, walk %.2f ms)  crc40=%08x"
          % (chunks, ns, dt * 1e3, ns * n / dt / 1e6, sa.value / k, sw.value / k, crc))
    del b
    torch.cuda.empty_cache()
