"""configs[4] shard: spx_batch_run_mixed call after call against spx_batch_run_mixed_ahead on two alternating batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from speedy_amd import config4 as C4
from speedy_amd.batch import Plan
ids = list(range(256))
streams = C4.make_streams(ids, threads=8)
plans = [Plan(r, False) for r in C4.RATES]
bs = [C4.mixed_batch(plans, ids, streams) for _ in range(2)]
print("spx_batch_run_mixed, one batch repeated       %.3f ms" % (bench.time_window(bs[0].run, 12, 4) * 1e3))
want = bs[0].crcs()
for b in bs:
    b.d_out.zero_()
for label, reps in (("warm", 6), ("timed", 20)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        bs[k % 2].run_ahead()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print("spx_batch_run_mixed_ahead, two batches in turn  %.3f ms" % (dt * 1e3))
print("same bytes:", bs[0].crcs() == want, bs[1].crcs() == want)
