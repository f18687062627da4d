"""Does a pipelined loop of 256-stream batches reach the large call's regime (2 048 streams per call: 0.73 ms per 256 streams) when
its launches are GROUPED the way the large call's are -- the analysis kernels of G batches back to back on one stream, then the G walk
launches side by side on G streams, the next group's analysis kernels behind them?  Built from the two stage calls
(spx_batch_analyze / spx_batch_walk), so the library's own mode decision is out of the picture; walk form by the tuning library's
switches.  Outputs compared with the plain call's.
    SPEEDY_HIP_LIB=speedy_amd/lib/ab/libspeedy_hip_tuning.so SPX_NO_EXCLUSIVE_CU=1 SPX_WALK_NWM=2 SPX_WALK_NWC=0 SPX_WALK_WCAP=1536 \
        python3 tools/r6/grouped_probe.py G [halves] [groups]"""
import ctypes as C
import os
import sys
import time
import zlib

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = bench.RATE * bench.SECONDS
plan = Plan(bench.RATE, False)
L = plan.L
bs = []
for h in range(H):
    for i in range(G):
        b = Batch(plan, [n] * 256, 1, bench.SPEED, 1.0, 0.0)
        if h == 0:
            b.upload(bench.make_streams(256, n, i))
        else:
            b.d_in = bs[i].d_in      # the halves share their inputs
        bs.append(b)
want = []
for i in range(G):
    bs[i].run()
    torch.cuda.synchronize()
    want.append([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in bs[i].results()])
dt1 = bench.time_window(bs[0].run, reps=10, warm=3)
print("plain spx_batch_run, one batch: %.3f ms" % (dt1 * 1e3))

sp = torch.cuda.Stream()
ws = [torch.cuda.Stream() for _ in range(G)]
done = [[None] * G for _ in range(H)]


def analyze(b, s):
    rc = L.spx_batch_analyze(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_ws.data_ptr(), b.d_ws.numel(), None, s.cuda_stream)
    assert rc == 0, L.spx_last_error()


def walk(b, s):
    rc = L.spx_batch_walk(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_out.data_ptr(), b.d_nout.data_ptr(), b.d_ws.data_ptr(),
                          b.d_ws.numel(), None, s.cuda_stream)
    assert rc == 0, L.spx_last_error()


def group(g):
    h = g % H
    for i in range(G):
        if done[h][i] is not None:
            sp.wait_event(done[h][i])      # the workspace's previous walk kernel
        analyze(bs[h * G + i], sp)
    ev = torch.cuda.Event()
    ev.record(sp)
    for i in range(G):
        ws[i].wait_event(ev)
        walk(bs[h * G + i], ws[i])
        e = torch.cuda.Event()
        e.record(ws[i])
        done[h][i] = e


for g in range(H + 1):
    group(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for g in range(REPS):
    group(H + 1 + g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (REPS * G)
ok = all([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in bs[h * G + i].results()] == want[i] for h in range(H) for i in range(G))
print("grouped, G = %d, %d halves: %.3f ms per batch of 256, %.0f Msamples/s, outputs equal the plain call's: %s"
      % (G, H, dt * 1e3, 256 * n / dt / 1e6, ok))
