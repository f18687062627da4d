"""GPU: coalesced execution of the drop-in API (include/sonic2.h, speedy_amd/csrc/sonic2_pool.hip).  Many sonicStream
handles are fed in turn the way a server would -- write to all, read from all (the per-handle loop is the reference
CLI's, speedy_wave.cc:199-231) -- and every handle must deliver what the oracle's restatement of the reference shim
(soniclib.c:391-452,519-552) delivers for the same calls: the same frames AND the same count readable after every write."""
import threading

import numpy as np
import pytest

from util import read_wav

pytestmark = pytest.mark.gpu


class _Ref:
    """One oracle-shim stream."""

    def __init__(self, orc, rate, ch, speed, nl, fb, mm):
        self.orc, self.L, self.ch = orc, orc.lib(), ch
        self.h = self.L.orc_sonicCreateStream(rate, ch, int(mm))
        self.L.orc_sonicSetSpeed(self.h, speed)
        self.L.orc_sonicEnableNonlinearSpeedup(self.h, nl)
        self.L.orc_sonicSetDurationFeedbackStrength(self.h, fb)
        self.buf = np.zeros(16384 * ch, np.int16)

    def write(self, seg):
        seg = np.ascontiguousarray(seg)
        assert self.L.orc_sonicWriteShortToStream(self.h, self.orc.sptr(seg), seg.size // self.ch) == 1

    def read(self, n):
        k = self.L.orc_sonicReadShortFromStream(self.h, self.orc.sptr(self.buf), n)
        return self.buf[:k * self.ch].copy()

    def flush(self):
        self.L.orc_sonicFlushStream(self.h)

    def close(self):
        self.L.orc_sonicDestroyStream(self.h)


def _configs(n, seed):
    rng = np.random.default_rng(seed)
    from speedy_amd.synth import speech_like
    tap, _, _ = read_wav("tapestry.wav")
    out = []
    for i in range(n):
        rate = 16000 if i % 4 != 3 else 22050
        ch = 2 if i % 8 == 5 else 1
        nl = 0.0 if i % 5 == 4 else 1.0
        speed = float(rng.choice([1.5, 2.0, 3.5, 3.5]))
        fb = 0.1 if i % 7 == 6 else 0.0
        mm = bool(i % 2)
        secs = float(rng.uniform(0.8, 2.0))
        if i % 6 == 0:
            x = np.roll(tap, 997 * i)[: int(secs * rate)]
        else:
            x = speech_like(int(secs * rate), rate, seed=100 + i)
        if ch == 2:
            x = np.stack([x, np.roll(x, 3)], axis=1).reshape(-1)
        out.append(dict(rate=rate, ch=ch, nl=nl, speed=speed, fb=fb, mm=mm, x=np.ascontiguousarray(x, np.int16)))
    return out


def test_64_interleaved_handles_equal_oracle_call_for_call(orc):
    """64 handles of mixed kinds (16 / 22.05 kHz, mono / stereo, linear / nonlinear, feedback, both hysteresis shapes),
    1000-frame writes to all of them, then a read from each: counts and bytes per call as the oracle's, and the pool
    really ran them together (tens of handles per launch sequence)."""
    from speedy_amd.sonic2 import SonicStream, pool_stats
    cfg = _configs(64, 1)
    refs = [_Ref(orc, c["rate"], c["ch"], c["speed"], c["nl"], c["fb"], c["mm"]) for c in cfg]
    hs = []
    for c in cfg:
        s = SonicStream(c["rate"], c["ch"], c["mm"])
        s.set_speed(c["speed"]); s.enable_nonlinear(c["nl"]); s.set_feedback(c["fb"])
        hs.append(s)
    runs0, jobs0 = pool_stats()
    chunk = 1000
    pos = 0
    longest = max(c["x"].size // c["ch"] for c in cfg)
    while pos < longest:
        live = [i for i, c in enumerate(cfg) if pos < c["x"].size // c["ch"]]
        for i in live:
            c = cfg[i]
            seg = c["x"][pos * c["ch"]:(pos + chunk) * c["ch"]]
            refs[i].write(seg)
            assert hs[i].write_short(seg) == 1
        for i in live:
            want = refs[i].read(chunk)
            got = hs[i].read_short(chunk)
            assert got.size == want.size and np.array_equal(got, want), (i, pos, got.size, want.size)
        pos += chunk
    for i in range(64):
        refs[i].flush()
        assert hs[i].flush() == 1
    for i in range(64):
        while True:
            want = refs[i].read(4096)
            got = hs[i].read_short(4096)
            assert np.array_equal(got, want), (i, "drain")
            if want.size == 0:
                break
        refs[i].close()
        hs[i].close()
    runs, jobs = pool_stats()
    assert runs > runs0 and (jobs - jobs0) / (runs - runs0) > 20, (runs - runs0, jobs - jobs0)


@pytest.mark.parametrize("name,ch,speed,nl,fb,mm,chunk", [
    ("tapestry.wav", 1, 3.5, 1.0, 0.0, False, 1000),
    ("tapestry.wav", 1, 2.0, 0.0, 0.0, False, 1024),
    ("tapestry22050.wav", 1, 1.5, 1.0, 0.1, True, 333),
    ("tapestry.wav", 2, 3.0, 1.0, 0.0, True, 128),
    ("tapestry.wav", 1, 0.6, 1.0, 0.0, False, 700),     # slow-down: the general walk kernel
])
def test_coalesced_and_eager_paths_agree(name, ch, speed, nl, fb, mm, chunk):
    """The same stream through both execution paths: identical bytes."""
    from speedy_amd.sonic2 import time_compress
    x, rate, _ = read_wav(name)
    if ch == 2:
        x = np.repeat(x, 2)
    a = time_compress(x, rate, ch, speed, nl, feedback=fb, chunk=chunk, match_matlab=mm, coalesce=True)
    b = time_compress(x, rate, ch, speed, nl, feedback=fb, chunk=chunk, match_matlab=mm, coalesce=False)
    assert a.size > 0 and np.array_equal(a, b)


def test_many_writes_before_a_read(orc):
    """Writes that wait together become one job: 37 writes of 1 .. 900 frames, one read; flush, write, flush without reads."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    rng = np.random.default_rng(5)
    for nl, speed in ((1.0, 3.5), (0.0, 2.0)):
        r = _Ref(orc, rate, ch, speed, nl, 0.0, False)
        s = SonicStream(rate, ch, False)
        s.set_speed(speed); s.enable_nonlinear(nl); s.set_feedback(0.0)
        pos = 0
        for rnd in range(6):
            for _ in range(37):
                w = int(rng.integers(1, 900))
                seg = x[pos:pos + w]
                pos += w
                r.write(seg)
                assert s.write_short(seg) == 1
            if rnd == 3:
                r.flush(); assert s.flush() == 1
                seg = x[pos:pos + 2500]; pos += 2500
                r.write(seg); assert s.write_short(seg) == 1
                r.flush(); assert s.flush() == 1
            assert s.available() == r.L.orc_sonicIntSamplesAvailable(r.h)
            want, got = r.read(16384), s.read_short(16384)
            assert np.array_equal(got, want), (nl, rnd)
        r.flush(); assert s.flush() == 1
        want, got = r.read(16384), s.read_short(16384)
        assert np.array_equal(got, want)
        r.close(); s.close()


def test_leaving_the_pool_mid_stream(orc):
    """Callbacks, a rate stage, a mode switch or a sonicInt* call arrive while the stream (and its neighbours) have work
    staged: the stream continues on a launch sequence of its own, bit-exact; the neighbours are not disturbed."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    kinds = ["callbacks", "rate", "mode", "direct", "none", "none"]
    refs, hs, seen = [], [], []
    for k in kinds:
        refs.append(_Ref(orc, rate, ch, 3.0, 1.0, 0.0, False))
        s = SonicStream(rate, ch, False)
        s.set_speed(3.0); s.enable_nonlinear(1.0); s.set_feedback(0.0)
        hs.append(s)
        seen.append(([], []))
    keep = []
    pos = 0
    step = 0
    while pos < x.size:
        seg = x[pos:pos + 800]
        pos += 800
        step += 1
        for i, k in enumerate(kinds):
            if step == 7:
                L = refs[i].L
                if k == "callbacks":
                    cb = orc.TENSION_FN(lambda _s, t, v, i=i: seen[i][0].append((t, np.float32(v))))
                    keep.append(cb)
                    L.orc_sonicTensionCallback(refs[i].h, cb)
                    hs[i].on_tension(lambda t, v, i=i: seen[i][1].append((t, np.float32(v))))
                elif k == "rate":
                    L.orc_sonicSetRate(refs[i].h, 1.25); hs[i].set_rate(1.25)
                elif k == "mode":
                    L.orc_sonicEnableNonlinearSpeedup(refs[i].h, 0.0); hs[i].enable_nonlinear(0.0)
                elif k == "direct":
                    d = np.ascontiguousarray(x[:500])
                    assert L.orc_sonicIntWriteShortToStream(refs[i].h, orc.sptr(d), 500) == 1
                    assert hs[i].int_write_short(d) == 1
            refs[i].write(seg)
            assert hs[i].write_short(seg) == 1
        if step % 3 == 0:
            for i in range(len(kinds)):
                want, got = refs[i].read(8192), hs[i].read_short(8192)
                assert np.array_equal(got, want), (kinds[i], step)
    for i in range(len(kinds)):
        refs[i].flush(); assert hs[i].flush() == 1
        want, got = refs[i].read(16384), hs[i].read_short(16384)
        assert np.array_equal(got, want), (kinds[i], "drain")
        assert seen[i][0] == seen[i][1]
        refs[i].close(); hs[i].close()
    assert len(seen[0][0]) > 100


def test_destroy_with_work_staged_and_reuse(orc):
    """A handle destroyed with staged writes takes them with it; its neighbours' staged work still runs and is right."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    r = _Ref(orc, rate, ch, 3.5, 1.0, 0.0, False)
    a, b = SonicStream(rate, ch, False), SonicStream(rate, ch, False)
    for s in (a, b):
        s.set_speed(3.5); s.enable_nonlinear(1.0); s.set_feedback(0.0)
    for pos in range(0, 20000, 1000):
        a.write_short(x[pos:pos + 1000]); b.write_short(x[pos:pos + 1000]); r.write(x[pos:pos + 1000])
        if pos == 9000:
            b.close()
            b = SonicStream(rate, ch, False)   # a fresh handle (very likely the same address, new state)
            b.set_speed(2.0)
    assert np.array_equal(a.read_short(16384), r.read(16384))
    a.close(); b.close(); r.close()


def test_long_stream_bounded_memory():
    """Ten minutes through a pooled handle in 1000-frame writes: the device buffers and the arena slots slide."""
    import torch
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    x = speech_like(16000 * 20, 16000, seed=3)
    s = SonicStream(16000, 1, False)
    s.set_speed(3.5); s.enable_nonlinear(1.0)
    free0 = None
    total = 0
    for rep in range(30):
        for pos in range(0, x.size, 1000):
            s.write_short(x[pos:pos + 1000])
            total += s.read_short(1000).size
        if rep == 2:
            free0 = torch.cuda.mem_get_info()[0]
    free1 = torch.cuda.mem_get_info()[0]
    s.close()
    assert total > 0.2 * 30 * x.size
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_handles_on_two_threads(orc):
    """Streams of one device driven from two host threads at once (the pool is locked): both equal the oracle."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    want = {}
    for speed in (2.5, 3.5):
        r = _Ref(orc, rate, ch, speed, 1.0, 0.0, False)
        outs = []
        for pos in range(0, x.size, 1000):
            r.write(x[pos:pos + 1000]); outs.append(r.read(1000))
        r.flush(); outs.append(r.read(16384)); r.close()
        want[speed] = np.concatenate(outs)
    got, errors = {}, []

    def worker(speed):
        try:
            for rep in range(3):
                s = SonicStream(rate, ch, False)
                s.set_speed(speed); s.enable_nonlinear(1.0); s.set_feedback(0.0)
                outs = []
                for pos in range(0, x.size, 1000):
                    s.write_short(x[pos:pos + 1000]); outs.append(s.read_short(1000))
                s.flush(); outs.append(s.read_short(16384)); s.close()
                got[(speed, rep)] = np.concatenate(outs)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(sp,)) for sp in (2.5, 3.5)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for (speed, rep), v in got.items():
        assert np.array_equal(v, want[speed]), (speed, rep)


def test_sixteen_threads_in_the_references_call_order_equal_oracle_call_for_call(orc):
    """Round 6 (flat combining, sonic2_pool.hip): sixteen host threads, each with four handles of mixed kinds, each running the
    REFERENCE's loop on its own handles -- write a chunk, read what is ready, next handle (speedy_wave.cc:199-220) -- at the same
    time.  Every call of every handle must return the oracle's count and bytes (the oracle is driven by the same thread right
    beside the handle), and the pool must have combined the threads' writes into common launch sequences."""
    from speedy_amd.sonic2 import SonicStream, pool_stats
    T, M = 16, 4
    cfg = _configs(T * M, 3)
    runs0, jobs0 = pool_stats()
    errors = []
    start = threading.Barrier(T)

    def worker(t):
        try:
            mine = cfg[t * M:(t + 1) * M]
            refs = [_Ref(orc, c["rate"], c["ch"], c["speed"], c["nl"], c["fb"], c["mm"]) for c in mine]
            hs = []
            for c in mine:
                s = SonicStream(c["rate"], c["ch"], c["mm"])
                s.set_speed(c["speed"]); s.enable_nonlinear(c["nl"]); s.set_feedback(c["fb"])
                hs.append(s)
            start.wait()
            chunk = 1000
            pos = 0
            longest = max(c["x"].size // c["ch"] for c in mine)
            while pos < longest:
                for i, c in enumerate(mine):
                    if pos >= c["x"].size // c["ch"]:
                        continue
                    seg = c["x"][pos * c["ch"]:(pos + chunk) * c["ch"]]
                    refs[i].write(seg)
                    assert hs[i].write_short(seg) == 1
                    want = refs[i].read(chunk)
                    got = hs[i].read_short(chunk)
                    assert got.size == want.size and np.array_equal(got, want), (t, i, pos, got.size, want.size)
                pos += chunk
            for i in range(M):
                refs[i].flush()
                assert hs[i].flush() == 1
                while True:
                    want = refs[i].read(4096)
                    got = hs[i].read_short(4096)
                    assert np.array_equal(got, want), (t, i, "drain")
                    if want.size == 0:
                        break
                refs[i].close()
                hs[i].close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e)[:500])
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    runs, jobs = pool_stats()
    assert runs > runs0 and (jobs - jobs0) / (runs - runs0) > 1.5, (runs - runs0, jobs - jobs0)
