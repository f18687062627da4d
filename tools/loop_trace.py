"""One loop kind only, for a kernel trace:  rocprofv3 --kernel-trace -- python3 tools/loop_trace.py {turns3|pipe_dev|pipe_dev_null} [steps]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Pipeline, Plan  # noqa: E402

kind = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n, S = bench.RATE * bench.SECONDS, int(os.environ.get("SPX_PROBE_STREAMS", bench.STREAMS_PER_GPU))
plan = Plan(bench.RATE, False)
streams = (bench.make_streams(bench.STREAMS_PER_GPU, n, 0) * ((S + 255) // 256))[:S]
if kind == "turns3":
    bs = [Batch(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0) for _ in range(3)]
    for b in bs:
        b.upload(streams)
    fn = lambda k: bs[k % 3].run_ahead(overlap=True)  # noqa: E731
elif kind == "pipe_host":
    depth = int(os.environ.get("SPX_PROBE_DEPTH", "4"))
    pipe = Pipeline(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0, depth=depth)
    h_in = torch.from_numpy(pipe.pack(streams)).pin_memory()
    tick = []

    def fn(k):
        tick.append(pipe.submit(h_in))
        if len(tick) >= depth:
            pipe.wait(tick[-depth])
else:
    b = Batch(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0)
    b.upload(streams)
    pipe = Pipeline(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0, depth=int(os.environ.get("SPX_PROBE_DEPTH", "4")), device_out=True)
    fn = lambda k: pipe.submit(b.d_in)  # noqa: E731
for k in range(12):
    fn(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    fn(12 + k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%s (%d streams): %.3f ms per step, %.0f Msamples/s" % (kind, S, dt * 1e3, S * n / dt / 1e6))
