"""Which step of a window stalls?  REPS times [synchronise; 6 steps with an event after each; synchronise]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from speedy_amd.batch import Batch, Plan
n = bench.RATE * bench.SECONDS
streams = bench.make_streams(bench.STREAMS_PER_GPU, n, 0)
plan = Plan(bench.RATE, False)
b = Batch(plan, [n] * bench.STREAMS_PER_GPU, 1, bench.SPEED, 1.0, 0.0)
b.upload(streams)
for _ in range(30):
    b.run()
torch.cuda.synchronize()
R, K = 60, 6
rows = []
for rep in range(R):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        b.run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(K)] + [wall])
rows = np.array(rows)
print("mode", "serial" if os.environ.get("SPX_SERIAL") else "concurrent")
print("per-step median:", np.round(np.median(rows, axis=0), 3).tolist())
print("per-step mean  :", np.round(rows.mean(axis=0), 3).tolist())
slow = rows[rows[:, -1] > np.median(rows[:, -1]) + 0.3]
print("windows with a stall: %d of %d; their per-step mean:" % (len(slow), R), np.round(slow.mean(axis=0), 3).tolist() if len(slow) else None)
fast = rows[rows[:, -1] <= np.median(rows[:, -1]) + 0.3]
print("windows without; per-step mean:", np.round(fast.mean(axis=0), 3).tolist())
