"""GPU differential fuzz: seeded random configurations (rate, channels, speed, nonlinear factor, feedback,
hysteresis shape) and adversarial signals through both entry points -- the batch C-ABI and the streaming sonic2 API
with random write sizes -- against the oracle.  Everything must be bit-identical (int16 output, float taps).

The signals are the ones a pitch search and an int16 cross-fade are most likely to get wrong: full-scale square
waves (the AMDF sums at their maximum), int16 extremes, DC, silence with a click, pure tones at the edges of the
65-400 Hz search range, white noise (no clear pitch: the previous-period rule fires), and speech-like synthetic.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SOAK = int(os.environ.get("SPX_FUZZ_SOAK", "0"))   # extra seeds for a one-off soak run
RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000]


def _signal(kind, n, rate, ch, rng):
    t = np.arange(n)
    if kind == "square":
        per = int(rng.integers(rate // 400, rate // 65 + 1))
        x = np.where((t // max(1, per // 2)) % 2 == 0, 32767, -32768)
    elif kind == "extremes":
        x = rng.choice(np.array([-32768, 32767, 0, -1, 1]), size=n)
    elif kind == "dc":
        x = np.full(n, int(rng.integers(-32768, 32768)))
    elif kind == "click":
        x = np.zeros(n, np.int64)
        if n:
            x[rng.integers(0, n, size=max(1, n // 4000))] = 30000
    elif kind == "tone":
        f = float(rng.choice([50.0, 65.0, 66.0, 120.0, 399.0, 400.0, 420.0, 1000.0]))
        x = np.round(20000 * np.sin(2 * np.pi * f * t / rate))
    elif kind == "noise":
        x = rng.integers(-20000, 20000, size=n)
    else:
        from speedy_amd.synth import speech_like
        return speech_like(n, rate, seed=int(rng.integers(1 << 30)), channels=ch)
    x = np.asarray(x, np.int64)
    if ch == 1:
        return x.astype(np.int16)
    cols = [np.roll(x, 3 * c) if c % 2 else x // (c + 1) for c in range(ch)]   # channels differ
    return np.stack(cols, axis=1).reshape(-1).astype(np.int16)


KINDS = ["square", "extremes", "dc", "click", "tone", "noise", "speech"]


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    for i in range(count):
        rate = int(rng.choice(RATES))
        ch = int(rng.choice([1, 1, 2, 3]))
        kind = KINDS[i % len(KINDS)]
        n = int(rng.integers(0, int(2.0 * rate)))
        speed = float(np.round(rng.choice([rng.uniform(0.3, 0.95), rng.uniform(1.05, 6.0), 1.0, 2.0, 0.5, rng.uniform(1.05, 6.0),
                                           rng.uniform(1.0, 1.00002), rng.uniform(25.0, 90.0)]), 6))   # the last: steps that fail (n == 0)
        nl = float(rng.choice([0.0, 1.0, 1.0, 0.5]))
        fb = float(rng.choice([0.0, 0.1, 0.5]))
        mm = bool(rng.integers(0, 2))
        yield i, rate, ch, kind, n, speed, nl, fb, mm, rng


@pytest.mark.parametrize("seed", list(range(1, 7)) + list(range(100, 100 + SOAK)))
def test_batch_fuzz(orc, seed):
    from speedy_amd.batch import compress_batch
    for i, rate, ch, kind, n, speed, nl, fb, mm, rng in _cases(seed, 21):
        x = _signal(kind, n, rate, ch, rng)
        # A batch job is ONE write of the whole stream.  That matters in exactly one regime: a linear job so fast
        # that a pitch step yields no output (speed > period + 1): the dependency then returns without consuming its
        # input and retries on the caller's next write, so its output depends on the caller's chunking.  The
        # streaming test below covers that regime call for call; here the oracle gets the same single write.
        ref = orc.compress_sound(x, rate, ch, speed, nl, fb, mm, chunk=1000 if nl != 0 else max(n, 1))
        outs, b = compress_batch([x], rate, ch, speed, nl, fb, mm, taps=(nl != 0))
        tag = (seed, i, rate, ch, kind, n, speed, nl, fb, mm)
        assert np.array_equal(outs[0], ref["out"]), tag
        if nl != 0:
            taps = b.tap_arrays(0)
            for key in ("tension", "speed", "features"):
                assert np.array_equal(taps[key], ref[key]), tag + (key,)


@pytest.mark.parametrize("seed", list(range(21, 25)) + list(range(2000, 2000 + SOAK)))
def test_batch_fuzz_rates_above_32_khz(orc, seed):
    """Round 4: the speed-up kernels at 24 ... 63 kHz -- the eight-search-wave form of the walk kernel with two lags per lane in
    its refine select (up to 121 lags; 44.1 kHz: 89, 48 kHz: 97), the compiled-in analysis kernels of 44.1 and 48 kHz, and the
    channel counts either side of skip x channels = 56, where the engine goes back to the general walk kernel.  Speed-up jobs
    of up to three seconds, several streams per call (the batch shape decides the form), every stream against the oracle."""
    from speedy_amd.batch import compress_batch
    rng = np.random.default_rng(seed)
    for i in range(6):
        rate = int(rng.choice([24000, 32000, 44100, 44100, 48000, 48000, 60000, 63999]))
        ch = int(rng.choice([1, 1, 2, 4, 5]))
        speed = float(np.round(rng.choice([rng.uniform(1.05, 1.99), rng.uniform(2.0, 6.0), 2.0, rng.uniform(25.0, 90.0)]), 6))
        nl = float(rng.choice([0.0, 1.0, 1.0, 0.5]))
        fb = float(rng.choice([0.0, 0.1]))
        mm = bool(rng.integers(0, 2))
        xs = []
        for k in range(int(rng.integers(1, 4))):
            n = int(rng.integers(0, int(3.0 * rate)))
            xs.append(_signal(KINDS[(i + k) % len(KINDS)], n, rate, ch, rng))
        outs, b = compress_batch(xs, rate, ch, speed, nl, fb, mm, taps=(nl != 0))
        for k, x in enumerate(xs):
            n = x.size // ch
            ref = orc.compress_sound(x, rate, ch, speed, nl, fb, mm, chunk=1000 if nl != 0 else max(n, 1))
            tag = (seed, i, k, rate, ch, n, speed, nl, fb, mm)
            assert np.array_equal(outs[k], ref["out"]), tag
            if nl != 0:
                taps = b.tap_arrays(k)
                for key in ("tension", "speed", "features"):
                    assert np.array_equal(taps[key], ref[key]), tag + (key,)


@pytest.mark.parametrize("seed", list(range(11, 15)) + list(range(1000, 1000 + SOAK)))
def test_streaming_fuzz(orc, seed):
    """Same idea through sonicWriteShortToStream / sonicReadShortFromStream with random write and read sizes: the
    frames available after every call must equal the oracle shim's."""
    from speedy_amd.sonic2 import SonicStream
    L = orc.lib()
    for i, rate, ch, kind, n, speed, nl, fb, mm, rng in _cases(seed, 14):
        x = _signal(kind, n, rate, ch, rng)
        tag = (seed, i, rate, ch, kind, n, speed, nl, fb, mm)
        h = L.orc_sonicCreateStream(rate, ch, int(mm))
        L.orc_sonicSetSpeed(h, speed)
        L.orc_sonicEnableNonlinearSpeedup(h, nl)
        L.orc_sonicSetDurationFeedbackStrength(h, fb)
        s = SonicStream(rate, ch, mm)
        s.set_speed(speed)
        s.enable_nonlinear(nl)
        s.set_feedback(fb)
        pos = 0
        buf = np.zeros(4096 * ch, np.int16)
        while pos < n:
            w = int(rng.integers(1, 3000))
            seg = np.ascontiguousarray(x[pos * ch:(pos + w) * ch])
            pos += w
            assert L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch) == 1
            s.write_short(seg)
            r = int(rng.integers(1, 4097))
            k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), r)
            got = s.read_short(r)
            assert got.size == k * ch and np.array_equal(got, buf[:k * ch]), tag + ("read", pos)
        L.orc_sonicFlushStream(h)
        s.flush()
        while True:
            k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 4096)
            got = s.read_short(4096)
            assert got.size == k * ch and np.array_equal(got, buf[:k * ch]), tag + ("drain",)
            if k == 0:
                break
        L.orc_sonicDestroyStream(h)
        s.close()


@pytest.mark.parametrize("seed", list(range(21, 27)) + list(range(2000, 2000 + SOAK)))
def test_batch_mixture_fuzz(orc, seed):
    """Random BATCHES: 3 .. 12 streams of one rate with per-stream channels, length, speed (some below 1), nonlinear
    factor and feedback in one call -- which kernel variant the batch gets depends on the mix -- each stream against
    the oracle."""
    from speedy_amd.batch import compress_batch
    rng = np.random.default_rng(seed)
    for rep in range(4):
        rate = int(rng.choice(RATES))
        k = int(rng.integers(3, 13))
        slow = bool(rng.integers(0, 3) == 0)          # one batch in three contains slow-down jobs
        multi = bool(rng.integers(0, 2))
        chs, speeds, nls, fbs, xs = [], [], [], [], []
        for i in range(k):
            ch = int(rng.choice([1, 2, 3])) if multi else 1
            n = int(rng.integers(0, int(1.0 * rate)))
            sp_ = float(np.round(rng.uniform(0.4, 0.95) if (slow and rng.integers(0, 2)) else rng.uniform(1.05, 5.0), 3))
            chs.append(ch); speeds.append(sp_)
            nls.append(float(rng.choice([0.0, 1.0, 1.0, 0.5]))); fbs.append(float(rng.choice([0.0, 0.1])))
            xs.append(_signal(KINDS[int(rng.integers(0, len(KINDS)))], n, rate, ch, rng))
        mm = bool(rng.integers(0, 2))
        from speedy_amd._lib import lib
        chunks, conc = int(rng.choice([1, 1, 2, 3])), int(rng.integers(0, 2))   # engine modes: same bytes in every one
        try:
            lib().spx_set_pipeline_chunks(chunks)
            lib().spx_set_concurrent(conc)
            outs, b = compress_batch(xs, rate, chs, speeds, nls, fbs, mm, taps=False)
        finally:
            lib().spx_set_pipeline_chunks(1)
            lib().spx_set_concurrent(1)
        for i in range(k):
            ref = orc.compress_sound(xs[i], rate, chs[i], speeds[i], nls[i], fbs[i], mm,
                                     chunk=1000 if nls[i] != 0 else max(xs[i].size // chs[i], 1))
            assert np.array_equal(outs[i], ref["out"]), (seed, rep, i, rate, chs[i], speeds[i], nls[i], fbs[i], mm, chunks, conc)


_case_log = []   # debugging: the case in progress, written out by the fixture below when a test fails under SPX_LC_DUMP


@pytest.fixture(autouse=True)
def _dump_failing_case(request):
    yield
    if os.environ.get("SPX_LC_DUMP") and _case_log and request.node.name.startswith("test_life_cycle_fuzz"):
        import pickle
        tag, x, log = _case_log
        with open("%s_%d_%d.pkl" % (os.environ["SPX_LC_DUMP"], tag[0], tag[1]), "wb") as f:   # the last case run = the failing one
            pickle.dump({"tag": tag, "x": x, "ops": list(log)}, f)


def _life_cycle_case(orc, rng, seed, i, ops, trace, coalesce=None, callbacks=None):
    """One random stream life (a generator: it yields after every call sequence step, so that several lives can be
    interleaved on one device -- the coalesced path then runs their staged work together)."""
    from speedy_amd.sonic2 import SonicStream
    L = orc.lib()
    rate = int(rng.choice([8000, 16000, 16000, 22050, 22050, 44100]))
    ch = int(rng.choice([1, 1, 1, 2, 3]))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.integers(rate // 10, int(2.5 * rate)))
    if i == 5:                                    # one long stream per seed: the device buffers slide several times
        n = int(rng.integers(6 * rate, 25 * rate))
    nl = float(rng.choice([0.0, 1.0, 1.0, 0.6]))
    speed = float(np.round(rng.choice([rng.uniform(0.4, 0.95), rng.uniform(1.05, 5.0), 2.0, 3.5]), 5))
    fb = float(rng.choice([0.0, 0.0, 0.1]))
    mm = bool(rng.integers(0, 2))
    small = bool(rng.integers(0, 2)) and i != 5
    with_cb = bool(rng.integers(0, 2))
    if callbacks is not None:
        with_cb = callbacks
    x = _signal(kind, n, rate, ch, rng)
    tag = (seed, i, rate, ch, kind, n, speed, nl, fb, mm, small)
    h = L.orc_sonicCreateStream(rate, ch, int(mm))
    s = SonicStream(rate, ch, mm, coalesce)
    ref_cb, got_cb = [], []
    if with_cb:                                   # tension and speed callbacks: same times, same values, same order
        keep = [orc.TENSION_FN(lambda _s, t, v: ref_cb.append(("t", t, np.float32(v)))),
                orc.TENSION_FN(lambda _s, t, v: ref_cb.append(("s", t, np.float32(v))))]
        L.orc_sonicTensionCallback(h, keep[0]); L.orc_sonicSpeedCallback(h, keep[1])
        s.on_tension(lambda t, v: got_cb.append(("t", t, np.float32(v))))
        s.on_speed(lambda t, v: got_cb.append(("s", t, np.float32(v))))
    L.orc_sonicSetSpeed(h, speed); s.set_speed(speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl); s.enable_nonlinear(nl)
    L.orc_sonicSetDurationFeedbackStrength(h, fb); s.set_feedback(fb)
    buf = np.zeros(8192 * ch, np.int16)
    pos, log, cur_nl = 0, [], nl
    _case_log[:] = [tag, x, log]
    if trace:
        print("CASE", tag, flush=True)
    while pos < n:
        if trace and log:
            print(" ", log[-1], flush=True)
        op = rng.random()
        if op < 0.06 and "flush" in ops:
            log.append("flush")
            L.orc_sonicFlushStream(h)
            assert s.flush() == 1, tag
        elif 0.06 <= op < 0.10 and "speed" in ops:
            v = float(np.round(rng.choice([rng.uniform(0.4, 0.95), rng.uniform(1.05, 5.0), 1.0]), 5))
            log.append(("speed", v))
            L.orc_sonicSetSpeed(h, v); s.set_speed(v)
        elif 0.10 <= op < 0.14 and "rate" in ops:
            v = float(rng.choice([0.5, 0.8, 1.0, 1.0, 1.25, 2.0]))
            log.append(("rate", v))
            L.orc_sonicSetRate(h, v); s.set_rate(v)
        elif 0.14 <= op < 0.17 and cur_nl != 0.0 and "nl" in ops:
            v = float(rng.choice([0.3, 0.7, 1.0]))
            cur_nl = v
            log.append(("nl", v))
            L.orc_sonicEnableNonlinearSpeedup(h, v); s.enable_nonlinear(v)
        elif 0.22 <= op < 0.26 and "direct" in ops:   # sonicInt*: the TSM stage alone, whatever the stream is doing
            u = rng.random()
            if u < 0.6:
                w = int(rng.integers(1, 1500))
                seg = np.ascontiguousarray(x[pos * ch:(pos + w) * ch])
                pos += w
                log.append(("iw", seg.size // ch))
                assert L.orc_sonicIntWriteShortToStream(h, orc.sptr(seg), seg.size // ch) == 1
                assert s.int_write_short(seg) == 1, tag + (s.L.speedyHipLastError(),)
            elif u < 0.8:
                log.append("iflush")
                L.orc_sonicIntFlushStream(h)
                assert s.int_flush() == 1, tag + (s.L.speedyHipLastError(),)
            else:
                v = float(np.round(rng.choice([rng.uniform(0.4, 0.95), rng.uniform(1.05, 4.0)]), 5))
                log.append(("ispeed", v))
                L.orc_sonicIntSetSpeed(h, v); s.int_set_speed(v)
        elif 0.19 <= op < 0.22 and "mode" in ops:   # linear <-> nonlinear inside one stream (soniclib.c:397-399)
            v = 0.0 if cur_nl != 0.0 else float(rng.choice([0.5, 1.0]))
            cur_nl = v
            log.append(("mode", v))
            L.orc_sonicEnableNonlinearSpeedup(h, v); s.enable_nonlinear(v)
        elif 0.17 <= op < 0.19 and "fb" in ops:
            v = float(rng.choice([0.0, 0.1, 0.3]))
            log.append(("fb", v))
            L.orc_sonicSetDurationFeedbackStrength(h, v); s.set_feedback(v)
        else:
            w = int(rng.integers(1, 400)) if (small or rng.random() < 0.3) else int(rng.integers(400, 4000))
            seg = np.ascontiguousarray(x[pos * ch:(pos + w) * ch])
            pos += w
            log.append(("w", seg.size // ch))
            assert L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch) == 1
            assert s.write_short(seg) == 1, tag + (s.L.speedyHipLastError(),)
            assert got_cb == ref_cb, tag + ("callbacks", pos, log[-12:])
        if trace:   # debugging: the frames readable after every call, not only the ones a read happens to ask for
            a, b = L.orc_sonicIntSamplesAvailable(h), s.L.sonicSamplesAvailable(s.h)
            assert a == b, tag + ("available", a, b, pos, log[-6:])
        if rng.random() < 0.1:   # the TSM stage's current speed (libsonic's sonicGetSpeed): last setter or last tension frame
            a, b = L.orc_sonicIntGetSpeed(h), s.L.sonicIntGetSpeed(s.h)
            assert np.float32(a) == np.float32(b), tag + ("speed now", a, b, pos, log[-6:])
        if rng.random() < 0.7:
            r = int(rng.integers(1, 8193))
            log.append(("r", r))
            k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), r)
            got = s.read_short(r)
            assert got.size == k * ch and np.array_equal(got, buf[:k * ch]), tag + ("read", pos, log[-12:])
        yield
    L.orc_sonicFlushStream(h)
    assert s.flush() == 1, tag + (s.L.speedyHipLastError(),)
    while True:
        k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 8192)
        got = s.read_short(8192)
        assert got.size == k * ch and np.array_equal(got, buf[:k * ch]), tag + ("drain", s.L.speedyHipLastError(), log[-12:])
        if k == 0:
            break
    L.orc_sonicDestroyStream(h)
    s.close()


@pytest.mark.parametrize("seed", list(range(31, 39)) + list(range(3000, 3000 + SOAK)))
def test_life_cycle_fuzz(orc, seed):
    """Random call sequences through the streaming API: writes from one frame to a few thousand (many shorter than an
    analysis hop), reads of random size, flushes anywhere (also before the first tension frame, twice in a row, with
    nothing written), and between writes new values for speed, rate (sonicSetRate), nonlinear factor -- including
    switches between 0 and non-zero, which make the reference interleave ring buffers and direct writes -- and
    feedback strength.  After every call the frames delivered must equal the oracle shim's."""
    rng = np.random.default_rng(seed)
    ops = os.environ.get("SPX_LC_OPS", "flush,speed,rate,nl,fb,mode,direct").split(",")   # debugging: leave op classes out
    trace = os.environ.get("SPX_LC_TRACE")                                     # debugging: print every call
    for i in range(6):
        for _ in _life_cycle_case(orc, rng, seed, i, ops, trace):
            pass


@pytest.mark.parametrize("seed", list(range(51, 55)) + list(range(5000, 5000 + SOAK // 4)))
def test_life_cycle_fuzz_eager(orc, seed):
    """The same lives with coalescing switched off: every handle runs its own launch sequence per write (the path handles
    with callbacks, a rate stage or mode switches always take)."""
    rng = np.random.default_rng(seed)
    ops = "flush,speed,rate,nl,fb,mode,direct".split(",")
    for i in range(4):
        for _ in _life_cycle_case(orc, rng, seed, i, ops, None, coalesce=False):
            pass


@pytest.mark.parametrize("seed", list(range(61, 69)) + list(range(6000, 6000 + SOAK)))
def test_life_cycle_fuzz_interleaved(orc, seed):
    """Six to twelve lives at once, their calls interleaved at random: writes and flushes of plain handles wait together
    and run as one launch sequence when any of them is asked for a result, handles drop out of the pool (callbacks, rate,
    sonicInt*, mode switches) while others have work staged, handles are destroyed with work staged.  Every call of every
    life still equals the oracle shim's."""
    pick = np.random.default_rng(seed)
    ops = "flush,speed,rate,nl,fb,mode,direct".split(",")
    cbs = None
    if pick.random() < 0.5:
        ops, cbs = ["flush", "speed", "fb", "nl"], False   # half of the seeds: nothing that takes a handle out of the pool
    k = int(pick.integers(6, 13))
    lives = [_life_cycle_case(orc, np.random.default_rng([seed, i]), seed, i % 5, ops, None, callbacks=cbs) for i in range(k)]
    while lives:
        j = int(pick.integers(0, len(lives)))
        try:
            next(lives[j])
        except StopIteration:
            lives.pop(j)


@pytest.mark.parametrize("seed", list(range(81, 85)) + list(range(8000, 8000 + SOAK // 4)))
def test_life_cycle_fuzz_threads(orc, seed):
    """Round 6 (flat combining, sonic2_pool.hip): the same random lives on SIX HOST THREADS at once, three lives after one another
    per thread -- every thread in the reference's call order on its own handles, while the other threads' staged writes travel in
    whatever launch sequence comes next, handles leave the pool (callbacks, rate, sonicInt*, mode switches) beside runs in flight
    and handles are destroyed while other threads wait for a run.  Every call of every life equals the oracle shim's."""
    import threading
    ops = "flush,speed,rate,nl,fb,mode,direct".split(",")
    pick = np.random.default_rng(seed)
    cbs = None
    if pick.random() < 0.5:
        ops, cbs = ["flush", "speed", "fb", "nl"], False   # half of the seeds: nothing that takes a handle out of the pool
    errors = []

    def worker(t):
        try:
            for i in range(3):
                for _ in _life_cycle_case(orc, np.random.default_rng([seed, t, i]), seed, (t + i) % 5, ops, None, callbacks=cbs):
                    pass
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)[:800]))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:2]


@pytest.mark.parametrize("seed", list(range(71, 74)) + list(range(7000, 7000 + SOAK // 10)))
def test_throughput_batch_fuzz(orc, seed):
    """Batches of 577 .. 1 100 streams: the walk kernel's throughput form (two search waves, no output waves, 1536-frame
    window, eight streams per CU) and two pipelined time chunks -- every stream against the oracle."""
    test_big_batch_fuzz(orc, seed, lo=577, hi=1101)


@pytest.mark.parametrize("seed", list(range(41, 44)) + list(range(4000, 4000 + SOAK // 10)))
def test_big_batch_fuzz(orc, seed, lo=257, hi=601):
    """Batches of 257 .. 600 streams (the large-batch path: speed-up kernels without output waves, pipelined time
    chunks, the sequential fallback) with per-stream channels, lengths 0 .. 1.5 s, speeds including exactly 1 and just
    above it, nonlinear factors 0 / 0.5 / 1, feedback, and the taps -- every stream against the oracle."""
    from speedy_amd.batch import compress_batch
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([8000, 11025, 16000, 16000, 22050, 22050, 44100]))
    k = int(rng.integers(lo, hi))
    slow = bool(rng.integers(0, 4) == 0)              # one batch in four contains slow-down jobs (general kernel)
    multi = bool(rng.integers(0, 2))
    with_taps = bool(rng.integers(0, 2))
    chs, speeds, nls, fbs, xs = [], [], [], [], []
    for i in range(k):
        ch = int(rng.choice([1, 1, 2, 3, 5])) if multi else 1
        n = int(rng.integers(0, int(1.5 * rate))) if rng.random() < 0.9 else int(rng.integers(0, 200))
        u = rng.random()
        if u < 0.05:
            sp_ = 1.0
        elif u < 0.10:
            sp_ = float(np.float32(1.0 + rng.uniform(0, 3e-5)))
        elif slow and u < 0.3:
            sp_ = float(np.round(rng.uniform(0.4, 0.95), 3))
        else:
            sp_ = float(np.round(rng.uniform(1.05, 5.0), 3))
        chs.append(ch); speeds.append(sp_)
        nls.append(float(rng.choice([0.0, 1.0, 1.0, 0.5]))); fbs.append(float(rng.choice([0.0, 0.0, 0.1])))
        xs.append(_signal(KINDS[int(rng.integers(0, len(KINDS)))], n, rate, ch, rng))
    mm = bool(rng.integers(0, 2))
    outs, b = compress_batch(xs, rate, chs, speeds, nls, fbs, mm, taps=with_taps)
    for i in range(k):
        ref = orc.compress_sound(xs[i], rate, chs[i], speeds[i], nls[i], fbs[i], mm,
                                 chunk=1000 if nls[i] != 0 else max(xs[i].size // chs[i], 1), taps=with_taps)
        tag = (seed, i, rate, chs[i], xs[i].size // chs[i], speeds[i], nls[i], fbs[i], mm)
        assert np.array_equal(outs[i], ref["out"]), tag
        if with_taps and nls[i] != 0:
            taps = b.tap_arrays(i)
            for key in ("tension", "speed", "features"):
                assert np.array_equal(taps[key], ref[key]), tag + (key,)


@pytest.mark.parametrize("seed", list(range(81, 87)) + list(range(8000, 8000 + SOAK // 10)))
def test_mixed_rate_batch_fuzz(orc, seed):
    """spx_batch_run_mixed, the call BASELINE configs[4] rides on: 2 .. 4 plans -- always one rate whose streams run on the
    general walk kernel or the plan-driven analysis (8 kHz, 44.1 kHz ...), hysteresis shape per plan -- 3 .. 900 streams dealt
    to them at random (a plan may get none), ragged lengths including 0, linear / nonlinear / slow-down jobs, per-stream
    channels and feedback, the taps in half of the cases -- every stream against the oracle."""
    from speedy_amd.batch import MixedBatch, Plan
    rng = np.random.default_rng(seed)
    n_plans = int(rng.integers(2, 5))
    rates = [int(rng.choice([8000, 11025, 32000, 44100, 48000]))]
    while len(rates) < n_plans:
        rates.append(int(rng.choice([16000, 22050, 16000, 22050, 8000, 44100, 24000])))
    order = rng.permutation(n_plans)
    rates = [rates[j] for j in order]
    mms = [bool(rng.integers(0, 2)) for _ in rates]
    plans = [Plan(r, m) for r, m in zip(rates, mms)]
    k = int(rng.choice([rng.integers(3, 40), rng.integers(40, 300), rng.integers(300, 901)]))
    empty = int(rng.integers(0, n_plans)) if rng.random() < 0.3 else -1          # a plan without streams
    slow = bool(rng.integers(0, 3) == 0)
    multi = bool(rng.integers(0, 2))
    with_taps = bool(rng.integers(0, 2))
    long_ones = max(1, 3000 // k)              # seconds of audio per stream shrink as the batch grows
    pidx, chs, speeds, nls, fbs, xs, lens = [], [], [], [], [], [], []
    for i in range(k):
        g = int(rng.integers(0, n_plans))
        if g == empty:
            g = (g + 1) % n_plans
        rate = rates[g]
        ch = int(rng.choice([1, 1, 2, 3])) if multi else 1
        u = rng.random()
        n = 0 if u < 0.04 else (int(rng.integers(1, 300)) if u < 0.10 else int(rng.integers(300, int(min(1.5, 0.3 * long_ones + 0.2) * rate))))
        u = rng.random()
        if u < 0.05:
            sp_ = 1.0
        elif slow and u < 0.3:
            sp_ = float(np.round(rng.uniform(0.4, 0.95), 3))
        else:
            sp_ = float(np.round(rng.uniform(1.05, 5.0), 3))
        pidx.append(g); chs.append(ch); speeds.append(sp_); lens.append(n)
        nls.append(float(rng.choice([0.0, 1.0, 1.0, 0.5]))); fbs.append(float(rng.choice([0.0, 0.0, 0.1])))
        xs.append(_signal(KINDS[int(rng.integers(0, len(KINDS)))], n, rate, ch, rng))
    b = MixedBatch(plans, pidx, lens, chs, speeds, nls, fbs, taps=with_taps)
    b.upload(xs)
    b.run()
    outs = b.results()
    for i in range(k):
        rate, mm = rates[pidx[i]], mms[pidx[i]]
        ref = orc.compress_sound(xs[i], rate, chs[i], speeds[i], nls[i], fbs[i], mm,
                                 chunk=1000 if nls[i] != 0 else max(lens[i], 1), taps=with_taps)
        tag = (seed, i, k, rates, pidx[i], chs[i], lens[i], speeds[i], nls[i], fbs[i], mm)
        assert np.array_equal(outs[i], ref["out"]), tag
        if with_taps and nls[i] != 0:
            taps = b.tap_arrays(i)
            for key in ("tension", "speed", "features"):
                assert np.array_equal(taps[key], ref[key]), tag + (key,)
    # ... and the same mix batch after batch through the owning pipeline object with its outputs left on the device (round 6: a
    # detached mixed call -- the walk kernels of consecutive batches overlap on the library's walk streams, a third and fourth group
    # on their plans' own): every batch of every buffer set equals the plain call's streams
    if sum(lens) > 0:
        import torch
        from speedy_amd.batch import Pipeline
        depth = int(rng.integers(2, 6))
        pipe = Pipeline(plans, lens, chs, speeds, nls, fbs, depth=depth, device_out=True, plan_index=pidx)
        d = torch.zeros(pipe.total_in + 64, dtype=torch.int16, device="cuda")
        d[: pipe.total_in].copy_(torch.from_numpy(pipe.pack(xs)))
        ts = [pipe.submit(d) for _ in range(2 * depth + 1)]
        for t in ts[-depth:]:
            got = pipe.results(t)
            for i in range(k):
                assert np.array_equal(got[i], outs[i]), (seed, "pipeline object", depth, t, i, rates, pidx[i], chs[i], lens[i], speeds[i], nls[i])
        pipe.close()
    for p in plans:
        p.close()
