"""configs[4] as ONE spx_batch_run_mixed call of N streams (default 2048: `config4_full`), a few steps -- for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o c4 -- python3 tools/c4_trace.py [N] [steps] [ahead]
   (ahead: two batches taking turns through spx_batch_run_mixed_ahead, the bench's `config4_shard` loop at N = 256)
   python3 tools/trace_summary.py DIR/.../c4_kernel_trace.csv"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd import config4 as C4  # noqa: E402
from speedy_amd.batch import Plan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ids = list(range(n))
streams = C4.make_streams(ids, threads=8)
plans = [Plan(r, False) for r in C4.RATES]
if int(os.environ.get("SPX_CHUNKS", "0")) > 0:     # A/B: the number of time chunks of a large call (0 / unset = the engine's own choice: 2)
    plans[0].L.spx_set_pipeline_chunks(int(os.environ["SPX_CHUNKS"]))
b = C4.mixed_batch(plans, ids, streams)
ahead = len(sys.argv) > 3 and sys.argv[3] == "ahead"
turn = [b, C4.mixed_batch(plans, ids, streams)] if ahead else [b]
run = (lambda k: turn[k % 2].run_ahead()) if ahead else (lambda k: b.run())
for k in range(4):
    run(k)
    if not ahead:
        torch.cuda.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    run(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("config4 x %d streams, one mixed call%s: %.3f ms per step, %.0f Msamples/s" % (n, " (pipelined)" if ahead else "", dt * 1e3, C4.input_frames(ids) / dt / 1e6))
