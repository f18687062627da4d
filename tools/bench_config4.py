"""One GPU's shard of BASELINE configs[4]: 256 streams x 10 s, stream i at 16 kHz if i is even else
22.05 kHz, mono if (i/2) is even else stereo, speed 1.5 if (i/4) is even else 3.5, nonlinear 1.  A "step" is the whole
shard: ONE spx_batch_run_mixed call (default), or -- for comparison -- one spx_batch_run per sample rate in sequence
(--two-calls) or on two HIP streams (--two-calls --two-streams).  Prints ms per step and Msamples/s of input frames."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from speedy_amd.batch import Batch, MixedBatch, Plan  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

from speedy_amd import config4 as C4  # noqa: E402

RATES = C4.RATES
cfg = C4.cfg


def shard_streams(n=256, seed0=4000, first=0):
    """Distinct signals of the global streams first .. first + n - 1 (seed = 4000 + global index, speedy_amd/config4.py)."""
    assert seed0 == C4.SEED0
    return C4.make_streams(range(first, first + n))


def mixed_batch(plans, streams, first=0):
    return C4.mixed_batch(plans, list(range(first, first + len(streams))), streams)


def time_steps(run, reps=10, warm=4):
    for _ in range(warm):   # (the engine's mode trial of a per-rate call needs three calls of a shape)
        run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if __name__ == "__main__":
    two_calls, two_streams = "--two-calls" in sys.argv, "--two-streams" in sys.argv
    streams = shard_streams()
    frames = sum(10 * cfg(i)[0] for i in range(256))
    plans = [Plan(r, False) for r in RATES]
    if not two_calls:
        b = mixed_batch(plans, streams)
        dt = time_steps(b.run)
        label = "one spx_batch_run_mixed call"
    else:
        batches = []
        for k, rate in enumerate(RATES):
            idx = [i for i in range(256) if cfg(i)[0] == rate]
            bb = Batch(plans[k], [10 * rate] * len(idx), [cfg(i)[1] for i in idx], [cfg(i)[2] for i in idx], 1.0, 0.0)
            bb.upload([streams[i] for i in idx])
            batches.append(bb)
        ss = [torch.cuda.Stream(), torch.cuda.Stream()] if two_streams else [None, None]
        dt = time_steps(lambda: [bb.run(stream=s) for bb, s in zip(batches, ss)])
        label = "one call per rate" + (" on two HIP streams" if two_streams else ", in sequence")
    print("configs[4] shard (256 mixed streams x 10 s), %s: %.3f ms per step, %.0f Msamples/s of input frames" %
          (label, dt * 1e3, frames / dt / 1e6))
