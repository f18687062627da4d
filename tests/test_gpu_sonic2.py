"""GPU: the reference-compatible streaming API (include/sonic2.h) of the HIP library, driven exactly like the
reference's own tests drive soniclib.c, compared with the oracle's restatement of that shim call for call."""
import numpy as np
import pytest

import sonic_props as sp
from util import read_wav

pytestmark = pytest.mark.gpu


def _gpu_compress(x, rate, ch, speed, nonlinear, chunk=1024, mm=True, feedback=0.0):
    from speedy_amd.sonic2 import time_compress
    return time_compress(x, rate, ch, speed, nonlinear, feedback=feedback, chunk=chunk, match_matlab=mm)


def _oracle_stream(orc, x, rate, ch, speed, nl, fb, mm, chunk):
    """Write/read/flush/drain through the oracle shim, recording the read count after every write."""
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, int(mm))
    L.orc_sonicSetSpeed(h, speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl)
    L.orc_sonicSetDurationFeedbackStrength(h, fb)
    outs, counts = [], []
    buf = np.zeros(chunk * ch, np.int16)
    n = x.size // ch
    for pos in range(0, n, chunk):
        seg = np.ascontiguousarray(x[pos * ch:(pos + chunk) * ch])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch)
        got = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), chunk)
        counts.append(got)
        outs.append(buf[: got * ch].copy())
    L.orc_sonicFlushStream(h)
    while True:
        got = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), chunk)
        if got == 0:
            break
        outs.append(buf[: got * ch].copy())
    L.orc_sonicDestroyStream(h)
    return np.concatenate(outs), counts


@pytest.mark.parametrize("name,ch,speed,nl,fb,mm,chunk", [
    ("tapestry.wav", 1, 3.5, 1.0, 0.0, False, 1000),   # compress_sound, speedy_wave.cc:199-231
    ("tapestry.wav", 1, 3.0, 1.0, 0.1, True, 128),     # MeasureExcessDuration, speedy_test.cc:663-684
    ("tapestry.wav", 1, 2.0, 0.0, 0.0, False, 1024),   # CompressSound, sonic_classic_test.cc:463-498
    ("tapestry22050.wav", 1, 1.5, 1.0, 0.0, True, 333),
    ("tapestry.wav", 2, 3.0, 1.0, 0.0, True, 128),     # duplicated stereo
])
def test_stream_equals_oracle_call_for_call(orc, name, ch, speed, nl, fb, mm, chunk):
    """Same bytes AND the same number of frames readable after every single write."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, _ = read_wav(name)
    if ch == 2:
        x = np.repeat(x, 2)
    ref, ref_counts = _oracle_stream(orc, x, rate, ch, speed, nl, fb, mm, chunk)
    s = SonicStream(rate, ch, mm)
    s.set_speed(speed)
    s.enable_nonlinear(nl)
    s.set_feedback(fb)
    assert s.buffer_size() == 0  # sonic_test.cc:496
    outs, counts = [], []
    n = x.size // ch
    for pos in range(0, n, chunk):
        assert s.write_short(x[pos * ch:(pos + chunk) * ch]) == 1
        got = s.read_short(chunk)
        counts.append(got.size // ch)
        outs.append(got)
    if nl != 0:
        assert s.buffer_size() == rate // 100  # sonic_test.cc:500
    assert s.flush() == 1
    while True:
        got = s.read_short(chunk)
        if got.size == 0:
            break
        outs.append(got)
    s.close()
    assert counts == ref_counts
    assert np.array_equal(np.concatenate(outs), ref)


def test_callbacks_match_oracle(orc):
    """The five monitoring callbacks (sonic2.h:104-125), in value and in call order."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    L = orc.lib()
    ref = {"tension": [], "speed": [], "features": [], "spec": [], "norm": []}
    h = L.orc_sonicCreateStream(rate, ch, 0)
    n = L.orc_sonicSpectrogramSize(h)
    cbs = [orc.TENSION_FN(lambda s, t, v: ref["tension"].append((t, v))),
           orc.TENSION_FN(lambda s, t, v: ref["speed"].append((t, v))),
           orc.FEATURES_FN(lambda s, t, p: ref["features"].append((t, np.ctypeslib.as_array(p, shape=(15,)).copy()))),
           orc.FEATURES_FN(lambda s, t, p: ref["spec"].append((t, np.ctypeslib.as_array(p, shape=(n,)).copy()))),
           orc.FEATURES_FN(lambda s, t, p: ref["norm"].append((t, np.ctypeslib.as_array(p, shape=(n,)).copy())))]
    L.orc_sonicTensionCallback(h, cbs[0])
    L.orc_sonicSpeedCallback(h, cbs[1])
    L.orc_sonicFeaturesCallback(h, cbs[2])
    L.orc_sonicSpectrogramCallback(h, cbs[3])
    L.orc_sonicNormalizedSpectrogramCallback(h, cbs[4])
    L.orc_sonicSetSpeed(h, 3.5)
    L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
    L.orc_sonicSetDurationFeedbackStrength(h, 0.0)
    for pos in range(0, x.size, 1000):
        seg = np.ascontiguousarray(x[pos:pos + 1000])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size)
    L.orc_sonicDestroyStream(h)

    got = {"tension": [], "speed": [], "features": [], "spec": [], "norm": []}
    s = SonicStream(rate, ch, False)
    s.set_speed(3.5)
    s.enable_nonlinear(1.0)
    s.set_feedback(0.0)
    s.on_tension(lambda t, v: got["tension"].append((t, v)))
    s.on_speed(lambda t, v: got["speed"].append((t, v)))
    s.on_features(lambda t, f: got["features"].append((t, f)))
    s.on_spectrogram(lambda t, f: got["spec"].append((t, f)))
    s.on_normalized(lambda t, f: got["norm"].append((t, f)))
    assert s.spectrogram_size() == n == 480
    for pos in range(0, x.size, 1000):
        s.write_short(x[pos:pos + 1000])
    s.close()
    assert len(got["tension"]) == len(ref["tension"]) == 303  # 314 analysis calls - F + 1
    assert len(got["spec"]) == len(ref["spec"]) == 314
    for key in ("tension", "speed"):
        assert [t for t, _ in got[key]] == [t for t, _ in ref[key]]
        assert np.array_equal(np.float32([v for _, v in got[key]]), np.float32([v for _, v in ref[key]]))
    for key in ("features", "spec", "norm"):
        assert [t for t, _ in got[key]] == [t for t, _ in ref[key]]
        assert np.array_equal(np.array([v for _, v in got[key]]), np.array([v for _, v in ref[key]]))
    assert [t for t, _ in got["tension"]][:3] == [0, 1, 2] and [t for t, _ in got["spec"]][:3] == [1, 2, 3]
    assert len(got["norm"]) == 314


def test_float_api(orc):
    """sonicWriteFloatToStream / sonicReadFloatFromStream, nonlinear and linear scaling (soniclib.c:496)."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    xf = (x.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    L = orc.lib()
    for nl in (1.0, 0.0):
        h = L.orc_sonicCreateStream(rate, ch, 0)
        L.orc_sonicSetSpeed(h, 2.5)
        L.orc_sonicEnableNonlinearSpeedup(h, nl)
        L.orc_sonicWriteFloatToStream(h, orc.fptr(xf), xf.size)
        L.orc_sonicFlushStream(h)
        buf = np.zeros(xf.size, np.float32)
        nref = L.orc_sonicReadFloatFromStream(h, orc.fptr(buf), xf.size)
        L.orc_sonicDestroyStream(h)
        s = SonicStream(rate, ch, False)
        s.set_speed(2.5)
        s.enable_nonlinear(nl)
        assert s.write_float(xf) == 1
        s.flush()
        out = s.read_float(xf.size)
        s.close()
        assert out.size == nref and np.array_equal(out, buf[:nref])


def test_reference_property_tests_through_the_hip_api():
    """sonic_classic_test.cc / sonic_test.cc properties on the product itself."""
    comp = lambda x, rate, ch, speed, nl: _gpu_compress(x, rate, ch, speed, nl)  # noqa: E731
    sp.check_sine_speed(comp, 3.0)
    sp.check_sine_speed(comp, 0.5)
    sp.check_mono_stereo_identity(comp, sp.sine_440(), 16000)
    x, rate, _ = read_wav("tapestry.wav")
    sp.check_mono_stereo_identity(comp, x, rate)
    sp.check_nonlinear_sine(comp, 3.0)


def test_chirp_and_dtw_properties_through_the_hip_api(orc):
    """sonic_classic_test.cc:303-394 (speed changes between writes) and sonic_test.cc:639-724 (DTW slopes of the
    linear and the nonlinear 3x tapestry) on the product; the oracle's speedySpectrogram is only the measuring
    device the reference test uses too."""
    from speedy_amd.sonic2 import SonicStream
    from test_oracle_sonic_properties import _spectrogram
    sp.check_chirp_speedup(lambda rate, ch: SonicStream(rate, ch, True))
    comp = lambda x, rate, ch, speed, nl: _gpu_compress(x, rate, ch, speed, nl)  # noqa: E731
    sp.check_speech_dtw(comp, _spectrogram(orc))


def test_remaining_reference_constraints_through_the_hip_api(orc):
    """Round 4: sonic_test.cc:965-1039 (varying speed, all ten cases with the upstream pass / fail pattern and the oracle's
    fingerprint), :759-862 (stereo sinusoids; the silent channel stays exactly 0), :597-637 (float API) through
    include/sonic2.h of the HIP library -- coalesced and eager handles -- and the outputs of the first two equal to the
    oracle's, sample for sample."""
    from speedy_amd.sonic2 import SonicStream
    from test_oracle_sonic_properties import _OrcStream
    for coalesce in (None, False):
        mk = lambda rate, ch: SonicStream(rate, ch, True, coalesce)  # noqa: E731
        sp.check_stereo_sinusoid(mk)
        sp.check_float_sinusoids(mk)
    mk = lambda rate, ch: SonicStream(rate, ch, True)  # noqa: E731
    got = sp.check_varying_speed(mk, sp.VARYING_SPEED_DELTAS)
    ref = [sp.varying_speed_delta(lambda rate, ch: _OrcStream(orc, rate, ch), s1, s2) for s1, s2, _ in sp.VARYING_SPEED_SPECS]
    assert got == ref, (got, ref)      # the same output lengths as the oracle, case for case
    for ch, matching in ((1, 1), (2, 1), (2, 0)):
        x = sp.create_sinusoid_test(22050, ch, matching, 1.0)
        a = sp.time_compress_vector(mk, x, 22050, ch, 3.0, 1e-5)
        b = sp.time_compress_vector(lambda rate, c: _OrcStream(orc, rate, c), x, 22050, ch, 3.0, 1e-5)
        assert np.array_equal(a, b), (ch, matching)
    xf = sp.create_sinusoid_float_test(22050, 1, 1)
    a = sp.time_compress_float_vector(mk, xf, 22050, 1, 3.0, 1e-5)
    b = sp.time_compress_float_vector(lambda rate, c: _OrcStream(orc, rate, c), xf, 22050, 1, 3.0, 1e-5)
    assert np.array_equal(a, b)


def test_negative_speed_input_does_not_crash():
    """speedy_test.cc:1059-1076: 24 kHz file, speed 0.25, nonlinear, one big write."""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("negative_speed.wav")
    s = SonicStream(rate, ch, True)
    s.set_speed(0.25)
    s.enable_nonlinear(1.0)
    assert s.write_short(x) == 1
    s.close()


def test_feedback_reduces_excess_duration(orc):
    """speedy_test.cc:653-711: 100 concatenations of tapestry at 3x; a stronger duration feedback leaves less
    excess duration.  (2048-frame writes instead of 128 keep the call count reasonable; the stream is the same.)"""
    from speedy_amd.sonic2 import SonicStream
    x, rate, ch = read_wav("tapestry.wav")
    long_x = np.tile(x, 100)

    def excess(fb):
        s = SonicStream(rate, ch, True)
        s.set_speed(3.0)
        s.enable_nonlinear(1.0)
        s.set_feedback(fb)
        got = 0
        for pos in range(0, long_x.size, 2048):
            s.write_short(long_x[pos:pos + 2048])
            got += s.read_short(4096).size
        s.close()
        return got

    def oracle_excess(fb):
        L = orc.lib()
        h = L.orc_sonicCreateStream(rate, ch, 1)
        L.orc_sonicSetSpeed(h, 3.0)
        L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
        L.orc_sonicSetDurationFeedbackStrength(h, fb)
        buf = np.zeros(4096, np.int16)
        got = 0
        for pos in range(0, long_x.size, 2048):
            seg = np.ascontiguousarray(long_x[pos:pos + 2048])
            L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size)
            got += L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 4096)
        L.orc_sonicDestroyStream(h)
        return got

    gots = [excess(fb) for fb in (0.0, 0.1, 0.2, 0.4)]
    assert gots == [oracle_excess(fb) for fb in (0.0, 0.1, 0.2, 0.4)]
    e = [abs(long_x.size / 3.0 - g) / rate for g in gots]
    assert e[1] < e[0] and e[2] < e[1] and e[3] < e[2]


def test_stereo_original_sonic_entry_points():
    """sonic_test.cc:728-754 (TestStereoOriginalSonic): the raw sonicInt* calls, stereo sine at 3x, repeated flushes;
    length within 1 %."""
    from speedy_amd._lib import c_short_p, lib
    L = lib()
    rate, ch, speed = 22050, 2, 3.0
    n = rate
    i = np.arange(n)
    mono = (16000 * np.sin(2 * np.pi * 237.0 * i / np.float32(rate))).astype(np.int16)
    x = np.repeat(mono, 2)
    h = L.sonicIntCreateStream(rate, ch)
    assert h and L.sonicIntGetNumChannels(h) == 2
    L.sonicIntSetSpeed(h, speed)
    assert L.sonicIntWriteShortToStream(h, x.ctypes.data_as(c_short_p), n) == 1
    buf = np.zeros(1024 * ch, np.int16)
    total = 0
    while True:
        got = L.sonicIntReadShortFromStream(h, buf.ctypes.data_as(c_short_p), 1024)
        total += got * ch
        L.sonicIntFlushStream(h)
        if got <= 0:
            break
    L.sonicIntDestroyStream(h)
    assert abs(total - x.size / speed) <= x.size / speed * 0.01


@pytest.mark.parametrize("nl", [1.0, 0.0])
def test_speed_changes_between_writes(orc, nl):
    """sonicSetSpeed between writes, across 1 (3.0 -> 0.7 -> 1.0 -> 2.2 -> 0.5 -> 4.0): the stream moves from the
    speed-up kernel to the general one and carries its state across; every read must equal the oracle shim's."""
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate, ch = 16000, 1
    x = speech_like(6 * rate, rate, seed=91)
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, 0)
    L.orc_sonicEnableNonlinearSpeedup(h, nl)
    s = SonicStream(rate, ch, False)
    s.enable_nonlinear(nl)
    speeds = [3.0, 0.7, 1.0, 2.2, 0.5, 4.0]
    buf = np.zeros(200000, np.int16)
    seg_len = x.size // len(speeds)
    for k, sp_ in enumerate(speeds):
        L.orc_sonicSetSpeed(h, sp_)
        s.set_speed(sp_)
        seg = x[k * seg_len:(k + 1) * seg_len]
        for pos in range(0, seg.size, 1500):
            part = np.ascontiguousarray(seg[pos:pos + 1500])
            assert L.orc_sonicWriteShortToStream(h, orc.sptr(part), part.size) == 1
            s.write_short(part)
            n = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 100000)
            got = s.read_short(100000)
            assert got.size == n and np.array_equal(got, buf[:n]), (nl, k, pos)
    L.orc_sonicFlushStream(h)
    s.flush()
    n = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 200000)
    got = s.read_short(200000)
    assert got.size == n and np.array_equal(got, buf[:n])
    L.orc_sonicDestroyStream(h)
    s.close()


def _drive_both(orc, L, h, s, ops, buf):
    """Run the same op list through the oracle shim (handle h) and the HIP stream s; every read must agree."""
    for i, op in enumerate(ops):
        kind = op[0]
        if kind == "w":
            part = np.ascontiguousarray(op[1])
            assert L.orc_sonicWriteShortToStream(h, orc.sptr(part), part.size // s.channels) == 1
            assert s.write_short(part) == 1, (i, s.L.speedyHipLastError())
        elif kind == "f":
            L.orc_sonicFlushStream(h)
            assert s.flush() == 1
        elif kind == "speed":
            L.orc_sonicSetSpeed(h, op[1]); s.set_speed(op[1])
        elif kind == "nl":
            L.orc_sonicEnableNonlinearSpeedup(h, op[1]); s.enable_nonlinear(op[1])
        elif kind == "r":
            n = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), op[1])
            got = s.read_short(op[1])
            assert got.size == n * s.channels, (i, op, got.size, n)
            assert np.array_equal(got, buf[: n * s.channels]), (i, op)


@pytest.mark.parametrize("nl,ch,speed,fb", [(1.0, 1, 3.5, 0.0), (1.0, 2, 1.6, 0.1), (0.0, 1, 2.0, 0.0), (0.5, 1, 0.7, 0.0)])
def test_write_after_flush_matches_the_reference_life_cycle(orc, nl, ch, speed, fb):
    """soniclib.c:529-552 leaves a stream usable after sonicFlushStream: the shim's read index jumps to its write
    index (tension frames in between are never computed), the TSM stage pads, truncates and empties its input, and later
    writes continue -- including the partial ring buffer that was pending at the flush.  Three flushes, ragged chunks,
    reads in between; every read equals the oracle shim's."""
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate = 16000
    x = speech_like(9 * rate, rate, seed=321, channels=ch)
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, 0)
    s = SonicStream(rate, ch, False)
    L.orc_sonicSetSpeed(h, speed); s.set_speed(speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl); s.enable_nonlinear(nl)
    L.orc_sonicSetDurationFeedbackStrength(h, fb); s.set_feedback(fb)
    buf = np.zeros(400000 * ch, np.int16)
    ops, pos, n = [], 0, x.size // ch
    rng = np.random.default_rng(7)
    cuts = [int(2.3 * rate) + 77, int(5.1 * rate) + 3, int(5.1 * rate) + 3 + 1000]   # flush points (one right after another write)
    while pos < n:
        k = int(rng.choice([1000, 1537, 160, 4000]))
        end = min(n, pos + k)
        for c in cuts:
            if pos < c <= end:
                end = c
        ops.append(("w", x[pos * ch:end * ch]))
        ops.append(("r", int(rng.integers(1, 5000))))
        if end in cuts:
            ops.append(("f",))
            ops.append(("r", 400000))
        pos = end
    ops += [("f",), ("r", 400000), ("r", 10)]
    _drive_both(orc, L, h, s, ops, buf)
    L.orc_sonicDestroyStream(h)
    s.close()


@pytest.mark.parametrize("rate_hz,chunk,flush_at", [(22050, 160, 17), (22050, 160, 10), (16000, 100, 12), (22050, 500, 3)])
def test_flush_before_the_first_tension_frame(orc, rate_hz, chunk, flush_at):
    """A flush that arrives before any tension frame exists (fewer than F analysis calls so far): the ring buffers go to
    the TSM stage at the global speed, and the first tension the shim computes afterwards is for a time > 0 -- that call
    is the one speedy.c:691 skips."""
    from speedy_amd.synth import speech_like
    x = speech_like(36 * chunk, rate_hz, seed=71)
    ro, rg, co, cg = _rate_streams(orc, x, rate_hz, 1, 2.0, 1.0, False, chunk, {"flush_at": flush_at})
    assert co == cg
    assert np.array_equal(ro, rg)
    assert ro.size > 0


def test_nonlinear_factor_changes_between_writes(orc):
    """The shim re-reads the nonlinear factor on every write (soniclib.c:397,343-345): 1.0 -> 0.3 -> 0.8 mid-stream."""
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate, ch = 22050, 1
    x = speech_like(6 * rate, rate, seed=55)
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, 1)
    s = SonicStream(rate, ch, True)
    L.orc_sonicSetSpeed(h, 3.0); s.set_speed(3.0)
    buf = np.zeros(300000, np.int16)
    ops = []
    third = x.size // 3
    for k, f in enumerate([1.0, 0.3, 0.8]):
        ops.append(("nl", f))
        seg = x[k * third:(k + 1) * third]
        for p in range(0, seg.size, 2048):
            ops.append(("w", seg[p:p + 2048]))
            ops.append(("r", 100000))
    ops += [("f",), ("r", 300000)]
    _drive_both(orc, L, h, s, ops, buf)
    L.orc_sonicDestroyStream(h)
    s.close()


def _rate_streams(orc, x, rate_hz, ch, speed, nl, mm, chunk, plan):
    """Drive the oracle shim and the HIP API with the same call sequence: `plan` maps a write index to a sonicSetRate
    value issued before that write; a flush + further writes if plan has key "flush_at".  Returns both outputs and the
    per-write read counts."""
    from speedy_amd.sonic2 import SonicStream
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate_hz, ch, int(mm))
    s = SonicStream(rate_hz, ch, mm)
    for f in (lambda v: L.orc_sonicSetSpeed(h, v), s.set_speed):
        f(speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl)
    s.enable_nonlinear(nl)
    L.orc_sonicSetDurationFeedbackStrength(h, 0.0)
    s.set_feedback(0.0)
    buf = np.zeros(4 * chunk * ch, np.int16)
    ro, rg, co, cg = [], [], [], []
    n = x.size // ch
    for w, pos in enumerate(range(0, n, chunk)):
        if w in plan:
            L.orc_sonicSetRate(h, plan[w])
            s.set_rate(plan[w])
        if plan.get("flush_at") == w:
            L.orc_sonicFlushStream(h)
            assert s.flush() == 1
        seg = np.ascontiguousarray(x[pos * ch:(pos + chunk) * ch])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch)
        assert s.write_short(seg) == 1
        got = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 4 * chunk)
        co.append(got)
        ro.append(buf[: got * ch].copy())
        g = s.read_short(4 * chunk)
        cg.append(g.size // ch)
        rg.append(g)
    L.orc_sonicFlushStream(h)
    assert s.flush() == 1
    while True:
        got = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 4 * chunk)
        if got == 0:
            break
        ro.append(buf[: got * ch].copy())
    while True:
        g = s.read_short(4 * chunk)
        if g.size == 0:
            break
        rg.append(g)
    L.orc_sonicDestroyStream(h)
    s.close()
    return np.concatenate(ro), np.concatenate(rg), co, cg


@pytest.mark.parametrize("name,ch,speed,nl,chunk,plan", [
    ("tapestry.wav", 1, 2.0, 0.0, 1000, {0: 1.25}),              # linear mode, faster playback rate
    ("tapestry.wav", 1, 1.0, 0.0, 777, {0: 0.8}),                # rate stage alone (speed 1 copies through)
    ("tapestry.wav", 1, 3.5, 1.0, 1000, {0: 0.5}),               # nonlinear speed-up, an octave down
    ("tapestry22050.wav", 1, 1.5, 1.0, 500, {0: 2.0}),
    ("tapestry.wav", 2, 3.0, 1.0, 640, {0: 1.1}),                # stereo
    ("tapestry.wav", 1, 2.5, 1.0, 1000, {5: 1.3, 20: 1.0, 30: 0.7}),           # set, back to 1, set again mid-stream
    ("tapestry.wav", 1, 2.0, 0.0, 1000, {0: 1.5, "flush_at": 12}),            # flush in the middle, then more input
    ("tapestry.wav", 1, 0.7, 0.0, 1000, {3: 1.2}),               # slow-down + rate
])
def test_set_rate_matches_the_oracle(orc, name, ch, speed, nl, chunk, plan):
    """sonicSetRate != 1 (soniclib.c:169-175 forwards it to the TSM dependency): the rate stage after the HIP walk
    kernel against the oracle's restatement of the dependency's adjustRate -- same bytes, same readable counts after
    every write (PARITY UNPINNED like the whole TSM stage: the dependency's source is not in the reference tree)."""
    x, rate_hz, _ = read_wav(name)
    x = x[: 40 * chunk]
    if ch == 2:
        x = np.repeat(x, 2)
    ro, rg, co, cg = _rate_streams(orc, x, rate_hz, ch, speed, nl, True, chunk, plan)
    assert co == cg
    assert np.array_equal(ro, rg)
    assert ro.size > 0


@pytest.mark.parametrize("seed", range(6))
def test_set_rate_fuzz(orc, seed):
    """Random rates (changed a few times mid-stream, sometimes back to 1), speeds, channel counts, chunk sizes and an
    optional flush in the middle: the HIP streaming API against the oracle shim, bytes and per-write counts."""
    rng = np.random.default_rng(900 + seed)
    rate_hz = int(rng.choice([16000, 22050, 8000]))
    ch = int(rng.choice([1, 1, 2, 3]))
    nl = float(rng.choice([0.0, 1.0]))
    speed = float(rng.choice([0.6, 1.0, 1.3, 2.0, 3.5]))
    if speed == 1.0:
        nl = 0.0
    chunk = int(rng.choice([160, 500, 1000, 1777]))
    writes = 36
    from speedy_amd.synth import speech_like
    x = speech_like(writes * chunk, rate_hz, seed=70 + seed, channels=ch)
    plan = {}
    for w in sorted(rng.choice(np.arange(writes), size=4, replace=False)):
        plan[int(w)] = float(rng.choice([0.5, 0.75, 1.0, 1.2, 1.5, 2.0]))
    if rng.random() < 0.5:
        plan["flush_at"] = int(rng.integers(5, writes - 5))
    ro, rg, co, cg = _rate_streams(orc, x, rate_hz, ch, speed, nl, bool(rng.integers(0, 2)), chunk, plan)
    assert co == cg, (rate_hz, ch, speed, nl, chunk, plan)
    assert np.array_equal(ro, rg), (rate_hz, ch, speed, nl, chunk, plan)


@pytest.mark.parametrize("start_nl,ch,speed", [(0.0, 1, 2.0), (1.0, 1, 3.5), (1.0, 2, 0.7), (0.0, 3, 1.4)])
def test_switching_between_linear_and_nonlinear_inside_one_stream(orc, start_nl, ch, speed):
    """soniclib.c:397-399 looks at the nonlinear factor on every write: with 0 the samples bypass the shim's ring and
    reach the TSM stage at once -- ahead of ring buffers written earlier that still wait for their tension -- and a
    flush hands the waiting buffers over at the last speed.  Same interleaving here, call for call."""
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate = 16000
    x = speech_like(6 * rate, rate, seed=31, channels=ch)
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, 0)
    s = SonicStream(rate, ch, False)
    L.orc_sonicSetSpeed(h, speed); s.set_speed(speed)
    L.orc_sonicSetDurationFeedbackStrength(h, 0.0); s.set_feedback(0.0)
    buf = np.zeros(8192 * ch, np.int16)
    nl, total = start_nl, 0
    for w, pos in enumerate(range(0, x.size // ch, 700)):
        if w % 9 == 4:
            nl = 0.0 if nl != 0.0 else 1.0                 # the other mode for the next nine writes
        if w == 50:
            L.orc_sonicFlushStream(h)
            assert s.flush() == 1
        L.orc_sonicEnableNonlinearSpeedup(h, nl); s.enable_nonlinear(nl)
        seg = np.ascontiguousarray(x[pos * ch:(pos + 700) * ch])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch)
        assert s.write_short(seg) == 1, s.L.speedyHipLastError()
        k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 8192)
        got = s.read_short(8192)
        assert got.size == k * ch and np.array_equal(got, buf[:k * ch]), w
        total += k
    L.orc_sonicFlushStream(h)
    assert s.flush() == 1
    while True:
        k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 8192)
        got = s.read_short(8192)
        assert got.size == k * ch and np.array_equal(got, buf[:k * ch])
        total += k
        if k == 0:
            break
    assert total > 0
    L.orc_sonicDestroyStream(h)
    s.close()


def test_ten_minute_soak_memory_stays_flat(orc):
    """10 minutes of 16 kHz audio in 1000-frame writes, each followed by a read (the reference CLI's loop): the device
    buffers slide, so free device memory after minute 2 and after minute 10 agree, and the bytes equal the oracle's."""
    import torch
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate = 16000
    piece = speech_like(60 * rate, rate, seed=808)
    x = np.tile(piece, 10)
    ref = orc.compress_sound(x, rate, 1, 3.5, 1.0, 0.1, False, chunk=1000, taps=False)["out"]
    s = SonicStream(rate, 1, False)
    s.set_speed(3.5); s.enable_nonlinear(1.0); s.set_feedback(0.1)
    outs, free_at = [], {}
    for pos in range(0, x.size, 1000):
        assert s.write_short(x[pos:pos + 1000]) == 1
        outs.append(s.read_short(1000))
        if pos in (120 * rate, 600 * rate - 1000):
            torch.cuda.synchronize()
            free_at[pos] = torch.cuda.mem_get_info()[0]
    s.flush()
    while True:
        got = s.read_short(4096)
        if got.size == 0:
            break
        outs.append(got)
    s.close()
    assert np.array_equal(np.concatenate(outs), ref)
    a, b = free_at[120 * rate], free_at[600 * rate - 1000]
    assert abs(a - b) <= 8 << 20, (a, b)   # the whole 10-minute input alone would be 19 MB, its taps far more


def test_settings_outside_the_defined_ranges_are_refused():
    """sonicSetSpeed / sonicSetRate / sonicEnableNonlinearSpeedup return nothing (sonic2.h:70-84): the next write or
    flush refuses a value the TSM stage has no defined behaviour for, with a message, and works again once it is fixed."""
    from speedy_amd.sonic2 import SonicStream
    x = np.zeros(2000, np.int16)
    nan = float("nan")
    for setter, bad, good in [("set_speed", [0.0, -2.0, nan, float("inf")], 2.0),
                              ("enable_nonlinear", [-0.5, 1.01, nan], 1.0),
                              ("set_rate", [0.0, -1.0, nan], 1.0),
                              ("set_feedback", [nan], 0.0)]:
        s = SonicStream(16000, 1, False)
        s.set_speed(2.0); s.enable_nonlinear(1.0)
        assert s.write_short(x) == 1
        for v in bad:
            getattr(s, setter)(v)
            assert s.write_short(x) == 0 and s.L.speedyHipLastError() != b"", (setter, v)
            assert s.flush() == 0, (setter, v)
        getattr(s, setter)(good)
        assert s.write_short(x) == 1 and s.flush() == 1
        s.close()
    with pytest.raises(RuntimeError):                       # more channels than one CU's LDS window can hold
        SonicStream(16000, 400, False)


@pytest.mark.parametrize("rate_hz,ch", [(50000, 1), (60000, 2), (3999, 1)])
def test_streaming_at_rates_with_the_small_analysis_tile(orc, rate_hz, ch):
    """Above 49 kHz the plan's analysis tile is the 8-frame one (LDS); the streaming API sizes its launches from the plan."""
    from speedy_amd.synth import speech_like
    x = speech_like(int(1.2 * rate_hz), rate_hz, seed=5, channels=ch)
    ro, rg, co, cg = _rate_streams(orc, x, rate_hz, ch, 3.0, 1.0, False, 1777, {"flush_at": 9})
    assert co == cg
    assert np.array_equal(ro, rg) and ro.size > 0


def test_create_destroy_cycles_leak_nothing():
    """1500 streams created, written, read, flushed and destroyed (both modes, with and without a rate stage), and 60
    plans created and destroyed around a batch call: free device memory and the process's resident set end where they
    were after a warm-up round."""
    import resource
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    x = speech_like(4000, 16000, seed=2)

    def round_(n_streams, n_plans):
        for i in range(n_streams):
            s = SonicStream(16000, 1 + (i % 2), False)
            s.set_speed(2.0 if i % 3 else 0.8)
            s.enable_nonlinear(1.0 if i % 2 else 0.0)
            if i % 5 == 0:
                s.set_rate(1.25)
            seg = x if s.channels == 1 else np.repeat(x, 2)
            assert s.write_short(seg) == 1
            s.read_short(4096)
            assert s.flush() == 1
            s.read_short(8192)
            s.close()
        for i in range(n_plans):
            p = Plan(16000 if i % 2 else 22050, False)
            b = Batch(p, [x.size] * 4, 1, 2.0, 1.0, 0.0)
            b.upload([x] * 4)
            b.run()
            b.results()
            p.L.spx_plan_destroy(p.h)
            p.h = None

    round_(100, 6)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    round_(1500, 60)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    # measured: 8 MiB of device memory (what the stream-ordered allocator's pool keeps) and 76 KiB of host memory
    assert free0 - free1 < 32 << 20, (free0, free1)
    assert rss1 - rss0 < 50 << 10, (rss0, rss1)              # ru_maxrss is in KiB
