"""Throughput against the number of streams in one batch call on one GPU (16 kHz mono, 10 s, 3.5x nonlinear)."""
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
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("chunks=%d streams=%5d  %.3f ms/call  %.0f Msamples/s" % (chunks, ns, dt * 1e3, ns * n / dt / 1e6))
    del b
    torch.cuda.empty_cache()
