"""2 048 streams x 10 s per call (the bench's `large_batch` kind, distinct signals), call after call: plain spx_batch_run against the
owning pipeline object (round 6: the producers of a large pipelined call start at once -- the next call's first analysis chunk runs
beside this call's last walk chunk).  python3 tools/r6/large_pipe.py [streams]"""
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
from speedy_amd.batch import Batch, Pipeline, Plan  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = bench.RATE * bench.SECONDS
plan = Plan(bench.RATE, False)
b = Batch(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0)
for blk in range((S + 255) // 256):
    xs = bench.make_streams(256, n, blk)[: S - 256 * blk]
    b.d_in[blk * 256 * n: blk * 256 * n + len(xs) * n].copy_(torch.from_numpy(np.concatenate(xs)))
dt = bench.time_window(b.run, reps=6, warm=3)
want = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in b.results()]
print("plain spx_batch_run, %d streams: %.3f ms per call, %.0f Msamples/s" % (S, dt * 1e3, S * n / dt / 1e6))
for depth in (2, 3, 4):
    pipe = Pipeline(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0, depth=depth, device_out=True)
    ts = [pipe.submit(b.d_in) for _ in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts += [pipe.submit(b.d_in) for _ in range(10)]
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / 10
    ok = all([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in pipe.results(t)] == want for t in ts[-depth:])
    print("pipeline object, depth %d: %.3f ms per call, %.0f Msamples/s, outputs equal the plain call's: %s" % (depth, dtp * 1e3, S * n / dtp / 1e6, ok))
    pipe.close()
