"""BASELINE configs[4] ("config 5"): ONE batch of 2 048 streams x 10 s -- stream i at 16 kHz if i is even else
22.05 kHz, mono if (i/2) is even else stereo, speed 1.5 if (i/4) is even else 3.5, nonlinear 1, feedback 0 -- sharded 256 per
GPU in contiguous blocks (stream i -> GPU i / 256).  The streams are one global sequence of distinct synthetic signals
(seed = SEED0 + global index), so any partition of the batch can be compared with any other stream by stream."""
import numpy as np

from .synth import speech_like

RATES = (16000, 22050)
TOTAL_STREAMS = 2048
SECONDS = 10
SEED0 = 4000


def cfg(i):
    """(sample rate, channels, speed) of global stream i."""
    return (16000 if i % 2 == 0 else 22050, 1 if (i // 2) % 2 == 0 else 2, 1.5 if (i // 4) % 2 == 0 else 3.5)


def kind(i):
    return i % 8   # cfg depends on i mod 8 only


def make_streams(ids, threads=8):
    """int16 signals of the global streams `ids` (distinct: seed = SEED0 + global index)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        rate, ch, _ = cfg(i)
        return speech_like(SECONDS * rate, rate, seed=SEED0 + i, channels=ch)
    with ThreadPoolExecutor(max(1, threads)) as ex:
        return list(ex.map(one, ids))


def rank_ids(rank, world, total=TOTAL_STREAMS):
    """Strong scaling of the fixed batch: rank r of `world` takes the contiguous block [r * total / world, (r+1) * total / world)."""
    per = total // world
    return list(range(rank * per, (rank + 1) * per if rank < world - 1 else total))


def mixed_batch(plans, ids, streams=None, taps=False):
    """A MixedBatch (one spx_batch_run_mixed call) over the global streams `ids`; plans = [Plan(16000), Plan(22050)]."""
    from .batch import MixedBatch
    b = MixedBatch(plans, [RATES.index(cfg(i)[0]) for i in ids], [SECONDS * cfg(i)[0] for i in ids],
                   [cfg(i)[1] for i in ids], [cfg(i)[2] for i in ids], 1.0, 0.0, taps=taps)
    if streams is not None:
        b.upload(streams)
    return b


def input_frames(ids):
    return int(sum(SECONDS * cfg(i)[0] for i in ids))


def algorithmic_bytes(ids, out_frames):
    """2 * C * (n_in + n_out) per stream."""
    return int(sum(2 * cfg(i)[1] * (SECONDS * cfg(i)[0] + int(o)) for i, o in zip(ids, out_frames)))
