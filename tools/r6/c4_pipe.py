"""BASELINE configs[4], one GPU's shard (256 mixed streams: 16 / 22.05 kHz, mono / stereo, 1.5x / 3.5x), step after step: the plain mixed
call, two MixedBatch objects taking turns under spx_batch_run_mixed_ahead (bench.py's config4_shard until round 6), and the owning
pipeline object created with spx_pipeline_create_mixed (outputs left on the device) at several depths.  Outputs compared with the plain
call's.    python3 tools/r6/c4_pipe.py"""
import os
import sys
import time
import zlib

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd import config4 as C4  # noqa: E402
from speedy_amd.batch import Pipeline, Plan  # noqa: E402

ids = list(range(256))
streams = C4.make_streams(ids, threads=8)
plans = [Plan(r, False) for r in C4.RATES]
b = C4.mixed_batch(plans, ids, streams)
dt = bench.time_window(b.run, 10, 4)
want = b.crcs()
frames = C4.input_frames(ids)
print("plain spx_batch_run_mixed: %.3f ms per step, %.0f Msamples/s" % (dt * 1e3, frames / dt / 1e6))
b2 = C4.mixed_batch(plans, ids, streams)
turn = [b, b2]
for k in range(4):
    turn[k % 2].run_ahead()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(40):
    turn[k % 2].run_ahead()
torch.cuda.synchronize()
dta = (time.perf_counter() - t0) / 40
print("two batches under spx_batch_run_mixed_ahead: %.3f ms per step, %.0f Msamples/s, same output: %s"
      % (dta * 1e3, frames / dta / 1e6, b.crcs() == want and b2.crcs() == want))
pidx = [C4.RATES.index(C4.cfg(i)[0]) for i in ids]
for depth in (2, 3, 4, 6):
    pipe = Pipeline(plans, [C4.SECONDS * C4.cfg(i)[0] for i in ids], [C4.cfg(i)[1] for i in ids], [C4.cfg(i)[2] for i in ids], 1.0, 0.0,
                    depth=depth, device_out=True, plan_index=pidx)
    ts = [pipe.submit(b.d_in, device=True) for _ in range(6)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts += [pipe.submit(b.d_in, device=True) for _ in range(40)]
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / 40
    ok = all([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in pipe.results(t)] == want for t in ts[-depth:])
    print("pipeline object (mixed), depth %d: %.3f ms per step, %.0f Msamples/s, same output: %s" % (depth, dtp * 1e3, frames / dtp / 1e6, ok))
    pipe.close()
