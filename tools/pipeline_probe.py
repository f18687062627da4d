"""Experiment: consecutive batches software-pipelined through the split-phase entry points -- spx_batch_analyze of batch k + 1 on
one HIP stream beside spx_batch_walk (tension + walk) of batch k on another, two workspaces.  ms per batch in steady state,
against spx_batch_run (the three kernels of ONE batch side by side)."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

rate, n, ns = 16000, 160000, 256
plan = Plan(rate, False)
L = plan.L
streams = [speech_like(n, rate, seed=1234 + i) for i in range(ns)]
bs = [Batch(plan, [n] * ns, 1, 3.5, 1.0, 0.0) for _ in range(2)]
for b in bs:
    b.upload(streams)
print("spx_batch_run, one batch repeated      %.3f ms" % (bench.time_window(bs[0].run, 20, 4) * 1e3))
ref = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in bs[0].results()]


def analyze(b, s):
    rc = L.spx_batch_analyze(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_ws.data_ptr(), b.d_ws.numel(), None, s.cuda_stream)
    assert rc == 0, L.spx_last_error().decode()


def walk(b, s):
    rc = L.spx_batch_walk(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_out.data_ptr(), b.d_nout.data_ptr(), b.d_ws.data_ptr(),
                          b.d_ws.numel(), None, s.cuda_stream)
    assert rc == 0, L.spx_last_error().decode()


sA, sW = torch.cuda.Stream(), torch.cuda.Stream()
ev_an = [torch.cuda.Event() for _ in bs]
ev_wk = [torch.cuda.Event() for _ in bs]
for label, reps in (("warm", 6), ("timed", 30)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        i = k % 2
        if k >= 2:
            sA.wait_event(ev_wk[i])           # the walk of batch k-2 has finished with this workspace
        analyze(bs[i], sA)
        ev_an[i].record(sA)
        sW.wait_event(ev_an[i])
        walk(bs[i], sW)
        ev_wk[i].record(sW)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print("analyze(k+1) beside walk(k), two streams %.3f ms per batch" % (dt * 1e3))
for b in bs:
    got = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in b.results()]
    print("outputs equal spx_batch_run's:", got == ref)

# the library's own form of it: spx_batch_run_ahead on two alternating batches, ONE stream
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    for label, reps in (("warm", 6), ("timed", 30)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(reps):
            bs[k % 2].run_ahead(stream=s1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
print("spx_batch_run_ahead, two batches alternating  %.3f ms per batch" % (dt * 1e3))
for b in bs:
    got = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in b.results()]
    print("outputs equal spx_batch_run's:", got == ref)

# overlapped walk kernels, two and three batches taking turns
bs.append(Batch(plan, [n] * ns, 1, 3.5, 1.0, 0.0))
bs[2].d_in.copy_(bs[0].d_in)
for turns in (2, 3, 2, 3):
    for label, reps in (("warm", 6), ("timed", 30)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(reps):
            bs[k % turns].run_ahead(overlap=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print("spx_batch_run_overlapped, %d batches taking turns  %.3f ms per batch" % (turns, dt * 1e3))
for b in bs:
    got = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in b.results()]
    print("outputs equal spx_batch_run's:", got == ref)
