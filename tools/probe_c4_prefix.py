"""Diagnostic: what in front of the configs[4] shard (in the same process) changes its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speedy_amd import config4 as C4
from speedy_amd.batch import Batch, Plan
from speedy_amd.synth import speech_like

what = sys.argv[1]
ids = list(range(256))
streams = C4.make_streams(ids)

def shard():
    plans = [Plan(r, False) for r in C4.RATES]
    b = C4.mixed_batch(plans, ids, streams)
    for _ in range(4):
        b.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3

if what == "alone":
    pass
elif what == "dummy_streams":
    keep = [torch.cuda.Stream() for _ in range(int(sys.argv[2]))]
elif what in ("main_conc", "main_serial"):
    if what == "main_serial":
        from speedy_amd._lib import lib
        lib().spx_set_concurrent(0)
    n = 160000
    x = [speech_like(n, 16000, seed=i) for i in range(16)]
    p = Plan(16000, False)
    b = Batch(p, [n] * 256, 1, 3.5, 1.0, 0.0)
    b.upload([x[i % 16] for i in range(256)])
    for _ in range(5):
        b.run()
    torch.cuda.synchronize()
    if what == "main_serial":
        lib().spx_set_concurrent(1)
elif what == "chunks_set":
    from speedy_amd._lib import lib
    lib().spx_set_pipeline_chunks(1)
print(what, sys.argv[2:], "shard %.3f ms" % shard(), flush=True)
