"""Does the ORDER in which a process creates its HIP streams move the rates?  One process per order (HIP maps streams onto hardware
queues / pipes in creation order).  python tools/order_probe.py {lib_first|pipe_first|torch_first} [depth]
Prints: plain spx_batch_run (three kernels side by side), three Batch objects taking turns (spx_batch_run_overlapped), the owning
pipeline with resident input (device out), the owning pipeline host to host -- ms per step, each one window of 40 after 12."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("SPX_PROBE_QUEUES", "8"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Pipeline, Plan  # noqa: E402

order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
flags_nowait = len(sys.argv) > 3 and sys.argv[3] == "nowait"
n = bench.RATE * bench.SECONDS
S = bench.STREAMS_PER_GPU
plan = Plan(bench.RATE, False)
streams = bench.make_streams(S, n, 0)
extra = []
if order == "torch_first":
    extra = [torch.cuda.Stream() for _ in range(3)]
bs = [Batch(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0) for _ in range(3)]
for b in bs:
    b.upload(streams)
pipe_d = pipe_h = None


def make_pipes():
    global pipe_d, pipe_h
    pipe_d = Pipeline(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0, depth=depth, device_out=True)
    pipe_h = Pipeline(plan, [n] * S, 1, bench.SPEED, 1.0, 0.0, depth=depth)


def touch_lib():
    bs[0].run()
    for k in range(4):
        bs[k % 3].run_ahead(overlap=True)
    torch.cuda.synchronize()


if order == "pipe_first":
    make_pipes()
    touch_lib()
else:
    touch_lib()
    make_pipes()
h_in = torch.from_numpy(pipe_h.pack(streams)).pin_memory()


def window(fn, reps=40, warm=12):
    for k in range(warm):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        fn(warm + k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def host_loop(k, tick=[]):
    tick.append(pipe_h.submit(h_in))
    if len(tick) >= depth:
        pipe_h.wait(tick[-depth])


for rnd in range(2):
    r = {"plain": window(lambda k: bs[0].run()),
         "turns3": window(lambda k: bs[k % 3].run_ahead(overlap=True)),
         "pipe_dev": window(lambda k: pipe_d.submit(bs[0].d_in)),
         "pipe_host": window(host_loop, reps=100, warm=30)}
    print("%s depth %d queues %s round %d: " % (order, depth, os.environ["GPU_MAX_HW_QUEUES"], rnd) + "  ".join("%s %.3f" % kv for kv in r.items()), flush=True)
