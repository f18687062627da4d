"""One GPU's shard of BASELINE configs[4]: 256 streams x 10 s, stream i at 16 kHz if i is even else
22.05 kHz, mono if (i/2) is even else stereo, speed 1.5 if (i/4) is even else 3.5, nonlinear 1.  One spx_batch_run per
sample rate (a plan is per rate); a "step" is both calls.  Prints ms per step and Msamples/s of input frames."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402


def cfg(i):
    return (16000 if i % 2 == 0 else 22050, 1 if (i // 2) % 2 == 0 else 2, 1.5 if (i // 4) % 2 == 0 else 3.5)


two_streams = "--two-streams" in sys.argv
batches, frames = [], 0
for rate in (16000, 22050):
    idx = [i for i in range(256) if cfg(i)[0] == rate]
    plan = Plan(rate, False)
    n = 10 * rate
    base = {}
    streams = []
    for i in idx:
        ch = cfg(i)[1]
        key = (ch, i % 16)
        if key not in base:
            base[key] = speech_like(n, rate, seed=4000 + i % 16, channels=ch)
        streams.append(base[key])
    b = Batch(plan, [n] * len(idx), [cfg(i)[1] for i in idx], [cfg(i)[2] for i in idx], 1.0, 0.0)
    b.upload(streams)
    batches.append(b)
    frames += n * len(idx)
ss = [torch.cuda.Stream(), torch.cuda.Stream()] if two_streams else [None, None]
for _ in range(4):   # the engine's mode trial needs three calls of a shape
    for b, s in zip(batches, ss):
        b.run(stream=s)
    torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    for b, s in zip(batches, ss):
        b.run(stream=s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("configs[4] shard (256 mixed streams x 10 s)%s: %.3f ms per step, %.0f Msamples/s of input frames" %
      (" on two HIP streams" if two_streams else "", dt * 1e3, frames / dt / 1e6))
