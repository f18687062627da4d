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


def test_streaming_api_from_four_threads(orc):
    """Four host threads, each with its own sonicStream (different rates, channels and modes), writing and reading at
    the same time: the streams share nothing but the cached plans, so each must deliver exactly what it delivers alone."""
    import threading
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    cfgs = [(16000, 1, 3.5, 1.0), (22050, 2, 2.0, 1.0), (16000, 1, 0.7, 0.0), (44100, 1, 1.5, 1.0)]
    xs = [speech_like(3 * r, r, seed=300 + i, channels=c) for i, (r, c, _, _) in enumerate(cfgs)]
    refs = [orc.compress_sound(xs[i], r, c, sp, nl, 0.0, False, chunk=1000, taps=False)["out"]
            for i, (r, c, sp, nl) in enumerate(cfgs)]
    outs, errs = [None] * len(cfgs), []
    start = threading.Barrier(len(cfgs))

    def work(i):
        try:
            r, c, sp, nl = cfgs[i]
            s = SonicStream(r, c, False)
            s.set_speed(sp); s.enable_nonlinear(nl); s.set_feedback(0.0)
            got = []
            start.wait()
            x = xs[i]
            for pos in range(0, x.size // c, 1000):
                assert s.write_short(np.ascontiguousarray(x[pos * c:(pos + 1000) * c])) == 1
                got.append(s.read_short(4096))
            assert s.flush() == 1
            while True:
                g = s.read_short(4096)
                if g.size == 0:
                    break
                got.append(g)
            s.close()
            outs[i] = np.concatenate(got)
        except Exception as e:   # noqa: BLE001 -- reported in the main thread
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(cfgs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(len(cfgs)):
        assert np.array_equal(outs[i], refs[i]), cfgs[i]
