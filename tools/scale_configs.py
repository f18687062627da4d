"""Throughput of one 256-stream batch call for the stream kinds of BASELINE configs[4] and beyond (10 s each)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, Pipeline, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

ns = 256
chunks = int(os.environ.get("SPX_CHUNKS", "0"))  # 0 = the engine's own choice
only = [int(v) for v in os.environ.get("SPX_ONLY_RATES", "").split(",") if v]
for rate, ch, speed, nl in [(16000, 1, 3.5, 1.0), (16000, 1, 1.5, 1.0), (22050, 1, 1.5, 1.0), (22050, 1, 3.5, 1.0),
                            (16000, 2, 3.5, 1.0), (22050, 2, 1.5, 1.0), (44100, 1, 3.5, 1.0), (48000, 2, 3.5, 1.0),
                            (16000, 1, 2.0, 0.0), (16000, 1, 0.5, 1.0), (16000, 1, 0.5, 0.0), (16000, 1, 0.25, 0.0), (22050, 2, 0.8, 1.0), (11025, 1, 3.5, 1.0), (12000, 1, 3.5, 1.0)]:
    if only and rate not in only:
        continue
    n = 10 * rate
    plan = Plan(rate, False)
    if chunks:
        plan.L.spx_set_pipeline_chunks(chunks)
    base = [speech_like(n, rate, seed=i, channels=ch) for i in range(4)]
    b = Batch(plan, [n] * ns, ch, speed, nl, 0.0)
    b.upload([base[i % 4] for i in range(ns)])
    for _ in range(4):   # the engine's mode trial (spx_engine.hip) needs three calls of a shape
        b.run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    plan.L.spx_set_timing(1)
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    plan.L.spx_set_timing(0)
    import ctypes as C
    sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
    plan.L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
    # the same kind with consecutive calls pipelined (spx_batch_run_ahead, two batches taking turns); the library takes the mode
    # for the shapes of its concurrent mode and runs every other call as the plain one
    b2 = Batch(plan, [n] * ns, ch, speed, nl, 0.0)
    b2.d_in.copy_(b.d_in)
    turn = [b, b2]
    for k in range(4):
        turn[k % 2].run_ahead()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8):
        turn[k % 2].run_ahead()
    torch.cuda.synchronize()
    dta = (time.perf_counter() - t0) / 8
    mode = plan.L.spx_debug_last_call_concurrent()
    # ... and batch after batch through the owning pipeline object (four buffer sets, outputs left on the device: the walk kernels of
    # consecutive batches overlap where the library's co-residency arithmetic allows it -- what bench.py's headline runs through)
    pipe = Pipeline(plan, [n] * ns, ch, speed, nl, 0.0, depth=4, device_out=True)
    for k in range(6):
        pipe.submit(b.d_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(12):
        pipe.submit(b.d_in)
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / 12
    pipe.close()
    print("rate=%5d ch=%d speed=%.1f nl=%.0f  %.3f ms/call  %.0f Msamples/s (frames)   analysis %.2f walk %.2f ms   | calls pipelined: %.3f ms  %.0f Msamples/s%s"
          "   | pipeline object: %.3f ms  %.0f Msamples/s" %
          (rate, ch, speed, nl, dt * 1e3, ns * n / dt / 1e6, sa.value / reps, sw.value / reps, dta * 1e3, ns * n / dta / 1e6,
           "" if mode == 2 else "  (mode not taken)", dtp * 1e3, ns * n / dtp / 1e6))
    del b, b2, pipe
    torch.cuda.empty_cache()
