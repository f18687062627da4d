"""ONE spx_batch_run of N streams x 10 s at 16 kHz mono (the bench's `large_batch` shape), a few steps -- for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o big -- python3 tools/big_trace.py [N] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = bench.RATE * bench.SECONDS
plan = Plan(bench.RATE, False)
streams = (bench.make_streams(256, n, 0) * ((S + 255) // 256))[:S]
b = Batch(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0)
b.upload(streams)
for _ in range(4):
    b.run()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    b.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%d streams, one call: %.3f ms per step, %.0f Msamples/s" % (S, dt * 1e3, S * n / dt / 1e6))
