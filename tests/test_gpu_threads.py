"""spx_batch_run from several host threads (INTEGRATION.md "Concurrency"): one plan per thread and one plan shared by
two threads, each thread on its own HIP stream.  Every run must produce the bytes of a solo run -- whichever of the calls
the engine lets take the concurrent three-kernel mode (only one per device at a time; the others run their kernels in
sequence, spx_engine.hip SpxDevGuard)."""
import threading
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _crcs(outs):
    return [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs]


def _make(plan, seed0, n_streams=48, seconds=3):
    from speedy_amd.batch import Batch
    from speedy_amd.synth import speech_like
    n = 16000 * seconds
    streams = [speech_like(n, 16000, seed=seed0 + i) for i in range(n_streams)]
    b = Batch(plan, [n] * n_streams, 1, 3.5, 1.0, 0.0)
    b.upload(streams)
    return b


@pytest.mark.parametrize("shared_plan", [False, True])
def test_two_threads_same_bytes_as_solo(shared_plan):
    import torch
    from speedy_amd.batch import Plan
    plans = [Plan(16000, False)]
    plans.append(plans[0] if shared_plan else Plan(16000, False))
    batches = [_make(plans[t], 100 * (t + 1)) for t in range(2)]
    solo = []
    for b in batches:
        b.run()
        solo.append(_crcs(b.results()))
    errors = []
    got = [None, None]

    def worker(t):
        try:
            s = torch.cuda.Stream()
            for _ in range(6):
                batches[t].run(stream=s)
            s.synchronize()
            got[t] = _crcs(batches[t].results())
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert got[0] == solo[0] and got[1] == solo[1]
