"""Experiment: the walk kernel writing its int16 output straight into pinned host memory (zero-copy over PCIe) instead of HBM +
gather + D2H copy.  Prints ms per batch for: resident, HBM output + D2H pipeline (bench.py's pcie_pipeline), zero-copy output."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

rate, n, ns = 16000, 160000, 256
plan = Plan(rate, False)
streams = [speech_like(n, rate, seed=1234 + i) for i in range(ns)]
b = Batch(plan, [n] * ns, 1, 3.5, 1.0, 0.0)
b.upload(streams)
print("resident           %.3f ms" % (bench.time_window(b.run, 20, 4) * 1e3))
ref = b.results()
dt, _ = bench.pcie_pipeline(plan, streams, n, reps=20)
print("HBM out + D2H      %.3f ms" % (dt * 1e3))

# zero-copy: output buffer in pinned host memory
bs = [Batch(plan, [n] * ns, 1, 3.5, 1.0, 0.0) for _ in range(2)]
h_in = torch.empty(bs[0].d_in.numel(), dtype=torch.int16).pin_memory()
h_in.zero_()
off = 0
for x in streams:
    h_in[off:off + x.size] = torch.from_numpy(x)
    off += x.size
for q in bs:
    q.d_out_dev = q.d_out
    q.d_out = torch.zeros(q.d_out.numel(), dtype=torch.int16).pin_memory()
h_nout = [torch.zeros(ns, dtype=torch.int64).pin_memory() for _ in bs]
s_h2d, s_run = torch.cuda.Stream(), torch.cuda.Stream()
ev_in = [torch.cuda.Event() for _ in bs]
ev_done = [torch.cuda.Event() for _ in bs]
warm, reps = 3, 20
for k in range(warm + reps):
    if k == warm:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    i = k % 2
    with torch.cuda.stream(s_h2d):
        if k >= 2:
            s_h2d.wait_event(ev_done[i])
        bs[i].d_in.copy_(h_in, non_blocking=True)
        ev_in[i].record(s_h2d)
    with torch.cuda.stream(s_run):
        s_run.wait_event(ev_in[i])
        bs[i].run(stream=s_run)
        h_nout[i].copy_(bs[i].d_nout, non_blocking=True)
        ev_done[i].record(s_run)
    if k >= 1:
        ev_done[(k - 1) % 2].synchronize()     # batch k-1's output is in host memory: "drained"
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("zero-copy output   %.3f ms" % (dt * 1e3))
out = bs[0].d_out.numpy()
ok = all(np.array_equal(out[bs[0].out_offs[i]:bs[0].out_offs[i] + ref[i].size], ref[i]) for i in range(ns))
print("zero-copy output equals the resident run's:", ok)

# the transfers alone
for name, src, dst in (("H2D of a batch's input (%.1f MB)" % (h_in.numel() * 2 / 1e6), h_in, bs[0].d_in),
                       ("D2H of a batch's output region (%.1f MB)" % (bs[0].d_out.numel() * 2 / 1e6), bs[0].d_out_dev, bs[0].d_out)):
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-46s %.3f ms  %.1f GB/s" % (name, dt * 1e3, src.numel() * 2 / dt / 1e9))

# what slows the step: the H2D alone, the D2H alone (steps back to back on one stream, the copy on another)
bb = b
s_c = torch.cuda.Stream()
h_big = torch.zeros(bb.d_out.numel(), dtype=torch.int16).pin_memory()
for name, fn in (("steps with an H2D of the other buffer running", lambda: bs[1].d_in.copy_(h_in, non_blocking=True)),
                 ("steps with a D2H of an output buffer running", lambda: h_big[:13_400_000].copy_(bs[1].d_out_dev[:13_400_000], non_blocking=True)),
                 ("steps alone", lambda: None)):
    for _ in range(3):
        bb.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        with torch.cuda.stream(s_c):
            fn()
        bb.run()
    torch.cuda.synchronize()
    print("%-48s %.3f ms per step" % (name, (time.perf_counter() - t0) / 20 * 1e3))
