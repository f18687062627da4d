"""Pins the CPU oracle's analysis path against the reference's own known-answer tests.

Each test restates one TEST_F of reference speedy_test.cc (cited per test) with the same inputs, the
same call sequence and the same tolerances.  The reference builds its tests with -DMATCH_MATLAB
(reference Makefile:57-67), so match_matlab=True unless a test covers both branches.
"""
import math

import numpy as np
import pytest

from util import read_wav

K_RATE = 22050  # speedy_test.cc:192


def cround(v):
    """std::round (half away from zero) for non-negative values; Python's round() is half-to-even."""
    return int(math.floor(float(v) + 0.5))


def test_first_order_filter(orc):
    """speedy_test.cc:135-156"""
    import ctypes as C
    L = orc.lib()

    class Fof(C.Structure):
        _fields_ = [("state", C.c_float), ("alpha", C.c_float)]

    L.orc_fof_design.argtypes = [C.POINTER(Fof), C.c_float]
    L.orc_fof_iterate.argtypes = [C.POINTER(Fof), C.c_float]
    L.orc_fof_iterate.restype = C.c_float
    L.orc_fof_reset.argtypes = [C.POINTER(Fof)]
    f = Fof()
    tc = 10
    L.orc_fof_design(C.byref(f), tc)
    out = L.orc_fof_iterate(C.byref(f), 1.0)
    first = out
    for _ in range(tc):
        out = L.orc_fof_iterate(C.byref(f), 0.0)
    assert abs(first * math.exp(-1) - out) < 1e-7
    L.orc_fof_reset(C.byref(f))
    assert abs(L.orc_fof_iterate(C.byref(f), 0.0)) < 1e-7


def test_spectrogram_calculation(orc):
    """speedy_test.cc:197-218: sin(10*pi*i/N) -> bin 10 = 88.8677 +- 1e-3, sidelobes below bin 1."""
    s = orc.Speedy(K_RATE)
    N = s.fft_size // 2
    i = np.arange(N)
    x = np.zeros(2 * N, np.float32)
    x[:N] = np.sin((10 * i / np.float32(N)).astype(np.float32) * np.pi)
    spec = s.spectrogram(x)
    freq = 10
    assert abs(spec[freq] - 88.8677) < 1e-3
    for k in range(N):
        if k != freq:
            assert spec[freq] > spec[k]
        if abs(k - freq) > 3:
            assert 20 * math.log10(spec[k]) <= 20 * math.log10(spec[1])


def test_spectrogram_sine_2200(orc):
    """speedy_test.cc:222-254: sizes 330/660, 2200 Hz peak bin, 88.48474 +- 1e-3, neighbours +-0.1."""
    s = orc.Speedy(K_RATE)
    assert s.frame_size == 330 and s.fft_size == 660
    i = np.arange(330)
    x = np.sin(2 * np.pi * i / np.float32(K_RATE) * np.float32(2200.0)).astype(np.float32)
    spec = s.spectrogram(x)
    half = spec[: s.fft_size // 2]
    pos = int(np.argmax(half))
    assert pos == s.FreqToBin(2200.0)
    assert abs(half[pos] - 88.4847412109375) < 1e-3
    assert abs(half[pos - 1] - 76.9396) < 1e-1
    assert abs(half[pos + 1] - 68.0196) < 1e-1


def test_preemphasis(orc):
    """speedy_test.cc:259-284: impulse response {1,-0.97,0,0} and state carried across calls."""
    s = orc.Speedy(K_RATE)
    x = np.array([1, 0, 0, 0], np.float32)
    s.PreemphasisFilter(orc.fptr(x), 4)
    assert np.allclose(x, [1.0, -0.97, 0, 0], atol=1e-7)
    s2 = orc.Speedy(K_RATE)
    outs = []
    for v in (1.0, 0.0, 0.0, 0.0):
        y = np.array([v], np.float32)
        s2.PreemphasisFilter(orc.fptr(y), 1)
        outs.append(float(y[0]))
    assert np.allclose(outs, [1.0, -0.97, 0.0, 0.0], atol=1e-7)


@pytest.mark.parametrize("match_matlab", [True, False])
def test_hysteresis_triangle(orc, match_matlab):
    """speedy_test.cc:288-313, both #ifdef branches, tolerance 1e-8."""
    if match_matlab:
        correct = [0] * 9 + [k / 16. for k in range(1, 8)] + [1] + [k / 24. for k in range(11, 0, -1)] + [0] * 4
    else:
        correct = [0] * 5 + [k / 24. for k in range(1, 12)] + [1.] + [k / 16. for k in range(7, 0, -1)] + [0] * 8
    assert len(correct) == 32
    s = orc.Speedy(K_RATE, match_matlab)
    for i in range(32):
        s.AddToHysteresisBuffer(float(i == 16), i)
    for i in range(32):
        assert abs(s.EvaluateHysteresis(i) - correct[i]) < 1e-8, i


def test_normalize_by_energy(orc):
    """speedy_test.cc:317-328"""
    x = np.array([0, 0, 1, 0, 1], np.float32)
    y = np.zeros(5, np.float32)
    e = orc.lib().orc_speedyNormalizeByEnergy(orc.fptr(x), orc.fptr(y), 5)
    assert abs(e - 2.0) < 1e-7
    assert np.allclose(y, [0, 0, math.sqrt(0.5), 0, math.sqrt(0.5)], atol=1e-7)


def test_add_data_peak_bins(orc):
    """speedy_test.cc:331-373"""
    s = orc.Speedy(K_RATE)
    N = s.frame_size
    i = np.arange(N)
    s.add_data(np.sin(2 * np.pi * i / np.float32(N)), 0)
    assert s.GetCurrentTime() == 0
    s.add_data(np.sin(2 * 2 * np.pi * i / np.float32(N)), 1)
    assert s.GetCurrentTime() == 1
    for t, b in ((0, 2), (1, 4)):
        spec = s.spectrogram_at(t)[: s.fft_size // 2]
        assert int(np.argmax(spec)) == b
        assert spec[b] > spec[b - 1] and spec[b] > spec[b + 1]


def test_local_energy(orc):
    """speedy_test.cc:380-412: pinned at sqrt(2) for 6 frames, ends at 1.7745e-4 +- 1e-8."""
    s = orc.Speedy(K_RATE)
    N = s.frame_size
    i = np.arange(N)
    amp = np.float32(1.0)
    at_max = 0
    for t in range(100):
        x = (np.sin(2 * np.pi * i / np.float32(N)) * amp).astype(np.float32)
        s.add_data(x, t)
        assert s.GetCurrentTime() == t
        spec = s.spectrogram_at(t)
        s.ComputeLocalEnergy(orc.fptr(spec), t)
        if s.GetEnergyCompressed() > 1.414:
            at_max += 1
        amp = np.float32(amp * np.float32(0.9))
    assert at_max == 6
    assert abs(s.GetEnergyCompressed() - 1.7745e-04) < 1e-8


def test_spectral_difference(orc):
    """speedy_test.cc:418-453: last speech_changes == 0 +- 1e-6."""
    s = orc.Speedy(K_RATE)
    N = s.frame_size
    i = np.arange(N)
    amp = np.float32(1.0)
    last = None
    for t in range(100):
        freq = t / 2.0
        x = (np.sin(2 * np.pi * freq * i / np.float32(N)) * amp).astype(np.float32)
        s.add_data(x, t)
        ct = s.GetCurrentTime()
        cur = s.spectrogram_at(ct)
        prev = s.spectrogram_at(ct - 1)
        s.ComputeSpectralDifference(orc.fptr(cur), orc.fptr(prev), t)
        last = s.GetSpeechChanges()
        amp = np.float32(amp * np.float32(0.9))
    assert abs(last) < 1e-6


def _decaying_sine():
    """Input of speedy_test.cc:457-478, with the C++ expression's types: the envelope is float
    (std::exp(float)), the sine is double, the product is rounded to float on store; the sound starts
    at int(kSilentStart * kSampleRate) = 3307."""
    rate = 22050
    n = rate
    start_f = np.float32(0.15) * np.float32(rate)  # float * int -> float
    x = np.zeros(n, np.float32)
    i = np.arange(n)
    e = np.exp(-(i.astype(np.float32) - start_f) / np.float32(rate * np.float32(0.5)))  # float32
    sn = np.sin(2 * np.pi * 220.0 * i / float(np.float32(rate)))  # double
    first = int(start_f)
    x[first:] = (e.astype(np.float64) * sn)[first:].astype(np.float32)
    return x


def test_tension_kat(orc):
    """speedy_test.cc:457-530: 99 frames in / 91 out, min -0.6, max 0.14273257, last -0.31351471."""
    x = _decaying_sine()
    s = orc.Speedy(22050, True)
    step = np.float32(22050 / np.float32(100))
    W = s.frame_size
    frame_count = int((x.size - W) / step + 1)
    tension = []
    out_t = 0
    for t in range(frame_count):
        start = cround(np.float32(t) * step)
        s.add_data(x[start:start + W], t)
        ok, v = s.compute_tension(out_t)
        if ok:
            tension.append(v)
            out_t += 1
    assert frame_count == 99 and out_t == 91
    tension = np.array(tension)
    assert abs(tension.min() - (-0.6)) < 1e-5
    assert abs(tension.max() - 0.14273257553577423) < 1e-6
    assert abs(tension[-1] - (-0.31351470947265625)) < 1e-5


def _tapestry_tensions(orc, always_zero_time):
    data, rate, ch = read_wav("tapestry.wav")
    assert data.size == 50381 and data[0] == 15 and ch == 1
    x = data.astype(np.float32)  # the reference feeds raw int16 magnitudes here (speedy_test.cc:541)
    s = orc.Speedy(rate, True)
    step = np.float32(rate / np.float32(100))
    W = s.frame_size
    frame_count = int((x.size - W) / step + 1)
    tension = []
    out_t = 0
    for t in range(frame_count):
        start = cround(np.float32(t) * step)
        s.add_data(x[start:start + W], t)
        ok, v = s.compute_tension(out_t)
        if ok:
            tension.append(v)
            out_t = 0 if always_zero_time else out_t + 1
    return s, np.array(tension, np.float32)


def test_real_speech_statistics(orc):
    """speedy_test.cc:534-594 (including its `output_time = 0` quirk at :564)."""
    s, tension = _tapestry_tensions(orc, True)
    assert tension.min() < -0.4 and tension.max() > 0.75
    assert abs(tension.astype(np.float64).mean()) < tension.max() / 6.0
    Rg = 2.1
    L = orc.lib()
    speed = np.array([L.orc_speedyComputeSpeedFromTension(float(t), Rg, 0.0, s.h) for t in tension])
    avg = speed.mean()
    assert abs(avg - Rg) < Rg / 10.0
    assert avg <= Rg - Rg / 20.0


def test_feature_return_latency(orc):
    """speedy_test.cc:714-757: frames in == tensions out + kTemporalHysteresisFuture; features[11]==tension."""
    for mm in (True, False):
        rate, n, f0 = 16000, 8000, 440.0
        i = np.arange(n)
        x = np.cos(2 * np.pi * f0 * i / np.float32(rate)).astype(np.float32)
        s = orc.Speedy(rate, mm)
        W = s.frame_size
        step = np.float32(rate / np.float32(100))
        frame_count = int((n - W) / step + 1)
        peak = int(f0 / (rate / s.fft_size))
        out_t = 0
        for t in range(frame_count):
            start = cround(np.float32(t) * step)
            s.add_data(x[start:start + W], t)
            ok, v = s.compute_tension(out_t)
            if ok:
                out_t += 1
                assert s.features()[11] == np.float32(v)
                spec = s.spectrogram()
                assert spec[peak] > spec[peak - 1] and spec[peak] > spec[peak + 1]
        assert out_t > 0
        assert frame_count == out_t + s.HysteresisFuture()
