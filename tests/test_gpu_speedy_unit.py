"""GPU: the unit-level API (include/speedy.h = reference speedy.h:61-100) of the HIP library, put through the
reference's own unit tests (speedy_test.cc) and its Matlab fixture comparison (speedy_test.cc:859-1057), and compared
bit for bit with the CPU oracle's restatement of speedy.c driven the same way."""
import math

import numpy as np
import pytest

from test_oracle_kat import _decaying_sine, cround
from test_oracle_matlab_fixture import _snr, _xcorr
from util import matlab_fixture, read_wav

pytestmark = pytest.mark.gpu


def _hip(rate, mm=True):
    from speedy_amd.speedy import Speedy
    return Speedy(rate, mm)


def test_spectrogram_known_answers():
    """speedy_test.cc:197-254: sin(10 pi i / N) -> bin 10 = 88.8677; a 2200 Hz sine peaks at speedyFreqToBin."""
    rate = 22050
    s = _hip(rate)
    W, N = s.frame_size, s.fft_size
    assert (W, N) == (330, 660) and s.frame_step == 220            # speedy_test.cc:228-230
    i = np.arange(W, dtype=np.float32)
    spec = s.spectrogram(np.sin(10 * 2 * np.pi * i / np.float32(N)).astype(np.float32))
    assert int(np.argmax(spec[: N // 2])) == 10 and abs(spec[10] - 88.8677) < 1e-3
    x = np.sin(2 * np.pi * 2200.0 * i / np.float32(rate)).astype(np.float32)
    spec = s.spectrogram(x)
    pos = int(np.argmax(spec[: N // 2]))
    assert pos == s.FreqToBin(2200.0) and abs(spec[pos] - 88.48474) < 1e-3
    assert abs(spec[pos - 1] - 76.94) < 0.1 and abs(spec[pos + 1] - 68.02) < 0.1
    s.close()


@pytest.mark.parametrize("mm", [True, False])
def test_tension_known_answer_and_oracle_equality(orc, mm):
    """speedy_test.cc:457-530 (decaying 220 Hz sine: tension min -0.6, max 0.14273257, last -0.31351471 with the
    MATCH_MATLAB shape) -- and every tension, feature row and spectrum equal to the oracle's, bit for bit."""
    rate, x = 22050, _decaying_sine()
    g, o = _hip(rate, mm), orc.Speedy(rate, mm)
    W = g.frame_size
    step = np.float32(rate / np.float32(100))
    frame_count = int((x.size - W) / step + 1)
    tg, out_t = [], 0
    for t in range(frame_count):
        start = cround(np.float32(t) * step)
        g.add_data(x[start:start + W], t)
        o.add_data(x[start:start + W], t)
        assert g.GetCurrentTime() == t
        assert np.array_equal(g.spectrogram(), o.spectrogram()), t
        okg, vg = g.compute_tension(out_t)
        oko, vo = o.compute_tension(out_t)
        assert okg == oko
        if okg:
            assert np.float32(vg) == np.float32(vo), (t, vg, vo)
            assert np.array_equal(g.features(), o.features()), (t, g.features(), o.features())
            assert np.array_equal(g.normalized(), o.normalized()), t
            tg.append(vg)
            out_t += 1
    assert frame_count == out_t + g.HysteresisFuture() and len(tg) == out_t     # speedy_test.cc:714-757
    if mm:
        assert abs(min(tg) + 0.6) < 1e-5 and abs(max(tg) - 0.14273257) < 1e-6 and abs(tg[-1] + 0.31351471) < 1e-5
    g.close()


def test_real_speech_normalized(orc):
    """speedy_test.cc:598-651 on the HIP library's unit-level API; tensions and speeds bit-equal to the oracle's."""
    import sonic_props as sp

    class _O(orc.Speedy):
        def speed_from_tension(self, t, rg, fb):
            return self.L.orc_speedyComputeSpeedFromTension(float(t), float(rg), float(fb), self.h)
    tg, sg = sp.check_real_speech_normalized(lambda rate: _hip(rate, True), cround)
    to, so = sp.check_real_speech_normalized(lambda rate: _O(rate, True), cround)
    assert np.array_equal(tg, to) and np.array_equal(sg, so)


def test_speed_from_tension_and_feedback(orc):
    """speedy.c:768-788 through the device state record: same speeds as the oracle for a tension sequence, with feedback."""
    g, o = _hip(16000), orc.Speedy(16000, True)
    rng = np.random.default_rng(1)
    for fb in (0.0, 0.3):
        for ten in rng.uniform(-0.6, 0.9, 50):
            a = g.speed_from_tension(ten, 3.0, fb)
            b = o.L.orc_speedyComputeSpeedFromTension(float(ten), 3.0, fb, o.h)
            assert np.float32(a) == np.float32(b), (fb, ten, a, b)
    g.close()


def test_tapestry_feature_computations_against_the_matlab_fixture():
    """speedy_test.cc:859-1057 with the HIP unit-level API in place of speedy.c: 314 frames in, 306 tension frames out,
    spectrogram / normalised spectrogram SNR > 27 dB at delay 0 and best there, every feature's best delay and SNR
    threshold as the reference's test demands."""
    fx = matlab_fixture()
    exp_spec, exp_norm, exp_feat = fx["spectrogram"], fx["normalized"], fx["features"]
    data, rate, ch = read_wav("tapestry22050.wav")
    x = (data.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    s = _hip(rate, True)
    W = s.frame_size
    step = np.float32(rate / np.float32(100))
    frame_count = int((x.size - W) / step + 1)
    spec, norm, feat, out_t, half = [], [], [], 0, s.fft_size // 2
    for t in range(frame_count):
        start = cround(np.float32(t) * step)
        s.add_data(x[start:start + W], t)
        spec.append(s.spectrogram()[:half])
        ok, v = s.compute_tension(out_t)
        if ok:
            norm.append(s.normalized())
            feat.append(s.features())
            out_t += 1
    s.close()
    spec, norm, feat = np.array(spec), np.array(norm), np.array(feat)
    assert spec.shape[0] == 314 and norm.shape[0] == 306 and feat.shape[0] == 306
    col, max_delay = 150, 20
    snrs = [10 * math.log10(_snr(exp_spec[col], spec[col + d])) for d in range(-max_delay, max_delay)]
    assert snrs[max_delay] > 27 and all(snrs[max_delay] > v for i, v in enumerate(snrs) if i != max_delay)
    for fr in range(norm.shape[0]):
        assert abs(float(np.sum(norm[fr] * norm[fr], dtype=np.float32)) - 1) < 4e-3
    nsnrs = [10 * math.log10(_snr(exp_norm[col], norm[col + d])) for d in range(-max_delay, max_delay)]
    assert nsnrs[max_delay] > 27 and all(nsnrs[max_delay] > v for i, v in enumerate(nsnrs) if i != max_delay)
    feature_list = [("Spectrogram energy", 0, 2e5), ("Energy Lowpass", 8, 7e5), ("Energy Local", 8, 4e4),
                    ("Energy Compressed", 8, 9e5), ("Energy Hysteresis", 0, 320), ("Low Energy Frame", 0, 1e8),
                    ("Local Spectral Difference", 0, 19), ("Emphasis Weighted Local Difference", 0, 29),
                    ("Emphasis Weighted Lowpass Filter", -1, 2300), ("Relative Spectral Difference", 0, 28),
                    ("Speech Changes", 0, 7), ("Audio Tension", 0, 8)]
    for k, (name, best_delay, thr) in enumerate(feature_list):
        with np.errstate(divide="ignore", invalid="ignore"):
            r = _xcorr(list(feat[:, k]), list(exp_feat[:, k]), 10)
        r = [(-1 if (v != v) else v) for v in r]
        best = int(np.argmax(r))
        assert best - 10 == best_delay, (name, best - 10, r[best])
        assert r[best] > thr, (name, r[best])


def test_shim_time_base_and_short_input(orc):
    """speedyAddDataShort with at_time starting at 1, as the shim drives it (soniclib.c:288-296)."""
    x, rate, _ = read_wav("tapestry.wav")
    g, o = _hip(rate, False), orc.Speedy(rate, False)
    W, B = g.frame_size, g.frame_step
    out_t = 0
    for j in range(60):
        fr = x[j * B:j * B + W]
        g.add_data_short(fr, j + 1)
        o.add_data_short(fr, j + 1)
        okg, vg = g.compute_tension(out_t)
        oko, vo = o.compute_tension(out_t)
        assert okg == oko
        if okg:
            assert np.float32(vg) == np.float32(vo), (j, vg, vo)
            out_t += 1
    assert out_t > 30
    g.close()


def test_first_tension_call_is_the_skipped_one_whatever_its_time(orc):
    """speedy.c:293,691: skip_frame_count starts at 1, so the FIRST speedyComputeTension that succeeds is treated as a
    low-energy frame -- also when that call is not for time 0 (the shim after an early flush, soniclib.c:538-550) --
    and times left out later are simply never computed."""
    x, rate, _ = read_wav("tapestry.wav")
    g, o = _hip(rate, False), orc.Speedy(rate, False)
    W, B = g.frame_size, g.frame_step
    base = 100 * B                                     # speech, not the leading silence
    asked = [5, 6, 7, 10, 11, 15] + list(range(16, 40))
    got = 0
    for j in range(60):
        fr = x[base + j * B:base + j * B + W]
        g.add_data_short(fr, j + 1)
        o.add_data_short(fr, j + 1)
        while asked:
            okg, vg = g.compute_tension(asked[0])
            oko, vo = o.compute_tension(asked[0])
            assert okg == oko
            if not okg:
                break
            assert np.float32(vg) == np.float32(vo), (asked[0], vg, vo)
            assert np.array_equal(g.features(), o.features()), asked[0]
            asked.pop(0)
            got += 1
    assert got == 30 and not asked
    g.close()


def test_unit_api_at_a_rate_with_the_small_analysis_tile(orc):
    """50 kHz: the float-frame analysis launch takes the plan's 8-frame tile (the 16-frame one does not fit the LDS)."""
    from speedy_amd.synth import speech_like
    rate = 50000
    x = speech_like(rate, rate, seed=9)
    g, o = _hip(rate, False), orc.Speedy(rate, False)
    W, B = g.frame_size, g.frame_step
    out_t = 0
    for j in range(40):
        fr = x[20 * B + j * B:20 * B + j * B + W]
        g.add_data_short(fr, j + 1)
        o.add_data_short(fr, j + 1)
        assert np.array_equal(g.spectrogram(), o.spectrogram()), j
        okg, vg = g.compute_tension(out_t)
        oko, vo = o.compute_tension(out_t)
        assert okg == oko
        if okg:
            assert np.float32(vg) == np.float32(vo), (j, vg, vo)
            out_t += 1
    assert out_t > 20
    g.close()


# ---- the reference's hooks between stages (speedy.h:102-133) through the HIP library: its own unit tests
# (speedy_test.cc:135-453, the same restatements tests/test_oracle_kat.py runs on the oracle) and oracle equality ----
K_RATE = 22050


def _fptr(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_hook_first_order_filter():
    """speedy_test.cc:135-156"""
    from speedy_amd._lib import lib
    L = lib()
    tc = 10
    f = L.CreateFirstOrderFilter(float(tc))
    assert f
    out = first = L.IterateFirstOrderFilter(f, 1.0)
    for _ in range(tc):
        out = L.IterateFirstOrderFilter(f, 0.0)
    assert abs(first * math.exp(-1) - out) < 1e-7
    L.ResetFirstOrderFilter(f)
    assert abs(L.IterateFirstOrderFilter(f, 0.0)) < 1e-7
    L.DesignFirstOrderLowpassFilter(f, 3.0)
    assert abs(L.IterateFirstOrderFilter(f, 1.0) - (1 - np.float32(math.exp(-1.0 / 3.0)))) < 1e-7
    L.DeleteFirstOrderFilter(f)


def test_hook_preemphasis(orc):
    """speedy_test.cc:259-284, and the state after speedyAddData as the reference leaves it (speedy.c:545)."""
    s = _hip(K_RATE)
    x = np.array([1, 0, 0, 0], np.float32)
    s.PreemphasisFilter(_fptr(x), 4)
    assert np.allclose(x, [1.0, -0.97, 0, 0], atol=1e-7)
    s2 = _hip(K_RATE)
    outs = []
    for v in (1.0, 0.0, 0.0, 0.0):
        y = np.array([v], np.float32)
        s2.PreemphasisFilter(_fptr(y), 1)
        outs.append(float(y[0]))
    assert np.allclose(outs, [1.0, -0.97, 0.0, 0.0], atol=1e-7)
    g, o = _hip(K_RATE), orc.Speedy(K_RATE)
    rng = np.random.default_rng(4)
    fr = rng.uniform(-1, 1, g.frame_size).astype(np.float32)
    g.add_data(fr, 0); o.add_data(fr, 0)
    a = rng.uniform(-1, 1, 50).astype(np.float32)
    b = a.copy()
    g.PreemphasisFilter(_fptr(a), 50); o.PreemphasisFilter(orc.fptr(b), 50)
    assert np.array_equal(a, b)
    for z in (s, s2, g):
        z.close()


@pytest.mark.parametrize("match_matlab", [True, False])
def test_hook_hysteresis_triangle(orc, match_matlab):
    """speedy_test.cc:288-313, both #ifdef branches, tolerance 1e-8; equal to the oracle's values bit for bit."""
    if match_matlab:
        correct = [0] * 9 + [k / 16. for k in range(1, 8)] + [1] + [k / 24. for k in range(11, 0, -1)] + [0] * 4
    else:
        correct = [0] * 5 + [k / 24. for k in range(1, 12)] + [1.] + [k / 16. for k in range(7, 0, -1)] + [0] * 8
    g, o = _hip(K_RATE, match_matlab), orc.Speedy(K_RATE, match_matlab)
    for i in range(32):
        g.AddToHysteresisBuffer(float(i == 16), i)
        o.AddToHysteresisBuffer(float(i == 16), i)
    for i in range(32):
        v = g.EvaluateHysteresis(i)
        assert abs(v - correct[i]) < 1e-8, i
        assert np.float32(v) == np.float32(o.EvaluateHysteresis(i)), i
    g.close()


def test_hook_normalize_by_energy(orc):
    """speedy_test.cc:317-328, and a random spectrum against the oracle."""
    from speedy_amd._lib import lib
    L = lib()
    x = np.array([0, 0, 1, 0, 1], np.float32)
    y = np.zeros(5, np.float32)
    e = L.speedyNormalizeByEnergy(_fptr(x), _fptr(y), 5)
    assert abs(e - 2.0) < 1e-7
    assert np.allclose(y, [0, 0, math.sqrt(0.5), 0, math.sqrt(0.5)], atol=1e-7)
    x = np.abs(np.random.default_rng(1).normal(size=330)).astype(np.float32)
    y, y2 = np.zeros(330, np.float32), np.zeros(330, np.float32)
    e = L.speedyNormalizeByEnergy(_fptr(x), _fptr(y), 330)
    e2 = orc.lib().orc_speedyNormalizeByEnergy(orc.fptr(x), orc.fptr(y2), 330)
    assert np.float32(e) == np.float32(e2) and np.array_equal(y, y2)


def test_hook_local_energy(orc):
    """speedy_test.cc:380-412: speedyAddData and then speedyComputeLocalEnergy on the same frame -- the energy filter runs
    TWICE per frame; pinned at sqrt(2) for 6 frames, ends at 1.7745e-4 +- 1e-8; every value equal to the oracle's."""
    g, o = _hip(K_RATE), orc.Speedy(K_RATE)
    N = g.frame_size
    i = np.arange(N)
    amp = np.float32(1.0)
    at_max = 0
    for t in range(100):
        x = (np.sin(2 * np.pi * i / np.float32(N)) * amp).astype(np.float32)
        g.add_data(x, t); o.add_data(x, t)
        assert g.GetCurrentTime() == t
        spec = g.spectrogram_at(t)
        g.ComputeLocalEnergy(_fptr(spec), t)
        o.ComputeLocalEnergy(orc.fptr(o.spectrogram_at(t)), t)
        assert np.float32(g.GetEnergyCompressed()) == np.float32(o.GetEnergyCompressed()), t
        if g.GetEnergyCompressed() > 1.414:
            at_max += 1
        amp = np.float32(amp * np.float32(0.9))
    assert at_max == 6
    assert abs(g.GetEnergyCompressed() - 1.7745e-04) < 1e-8
    g.close()


def test_hook_spectral_difference(orc):
    """speedy_test.cc:418-453: last speech_changes == 0 +- 1e-6; every feature row equal to the oracle's where both define it."""
    g, o = _hip(K_RATE), orc.Speedy(K_RATE)
    N = g.frame_size
    i = np.arange(N)
    amp = np.float32(1.0)
    last = None
    for t in range(100):
        freq = t / 2.0
        x = (np.sin(2 * np.pi * freq * i / np.float32(N)) * amp).astype(np.float32)
        g.add_data(x, t); o.add_data(x, t)
        ct = g.GetCurrentTime()
        cur, prev = g.spectrogram_at(ct), g.spectrogram_at(ct - 1)
        assert np.array_equal(cur, o.spectrogram_at(ct)) and np.array_equal(prev, o.spectrogram_at(ct - 1)), t
        g.ComputeSpectralDifference(_fptr(cur), _fptr(prev), t)
        o.ComputeSpectralDifference(orc.fptr(cur), orc.fptr(prev), t)
        last = g.GetSpeechChanges()
        assert np.float32(last) == np.float32(o.GetSpeechChanges()), t
        fg, fo = g.features(), o.features()
        for idx in (0, 4, 5, 6, 7, 8, 9, 10, 13):       # what speedyComputeSpectralDifference writes (speedy.c:664-729)
            assert np.float32(fg[idx]) == np.float32(fo[idx]), (t, idx, fg[idx], fo[idx])
        amp = np.float32(amp * np.float32(0.9))
    assert abs(last) < 1e-6
    g.close()


def test_history_is_bounded_when_tensions_lag(orc):
    """A caller that only adds frames (or lags far behind with speedyComputeTension) does not make the stream keep every
    row: device memory stays flat over 20 000 frames.  A tension far behind the newest frame is answered as the reference
    answers it -- from whatever its 21-entry spectrum ring and 42-entry hysteresis ring hold NOW (speedy.c:198-200,484-487,
    594-608; round 4: the same value as the oracle's, bit for bit) -- and recent ones still come out right."""
    import ctypes as C
    import torch
    from speedy_amd._lib import c_float_p, lib
    L = lib()
    L.speedyHipSetMatchMatlab(0)
    s = L.speedyCreateStream(16000)
    o = orc.Speedy(16000, False)
    W = L.speedyInputFrameSize(s)
    rng = np.random.default_rng(3)
    frame = (rng.standard_normal(W) * 0.1).astype(np.float32)
    for t in range(600):
        L.speedyAddData(s, frame.ctypes.data_as(c_float_p), t)
        o.add_data(frame, t)
    free0 = torch.cuda.mem_get_info()[0]
    for t in range(600, 20000):
        x = np.roll(frame, t)
        L.speedyAddData(s, x.ctypes.data_as(c_float_p), t)
        o.add_data(x, t)
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)
    v = C.c_float(0)
    for t in (100, 19999 - 12, 7, 19999 - 12, 19999 - 30):      # stale, recent, stale, the same again, between the two rings
        ok, ref = o.compute_tension(t)
        assert L.speedyComputeTension(s, t, C.byref(v)) == int(ok) == 1
        assert np.float32(v.value) == np.float32(ref), (t, v.value, ref)
        got = np.ctypeslib.as_array(L.speedyGetInternalState(s), shape=(15,)).copy()
        want = o.features()
        for i in (0, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14):
            assert got[i] == want[i], (t, i, got[i], want[i])
    L.speedyDestroyStream(s)
