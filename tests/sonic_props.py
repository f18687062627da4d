"""Property checks restated from the reference's sonic_classic_test.cc and sonic_test.cc.  Each takes a
`compress(x, rate, channels, speed, nonlinear) -> int16 array` callable, so the same assertions run against
the CPU oracle (tests/test_oracle_sonic_properties.py) and against the HIP streaming API (tests/test_gpu_sonic2.py).
"""
import math

import numpy as np

from util import read_wav


def teager_variance(data):
    """sonic_classic_test.cc:106-123 (online mean / variance of x[n]^2 - x[n-1]x[n+1], float arithmetic)."""
    d = np.asarray(data, np.float32)
    t = d[1:-1] * d[1:-1] - d[:-2] * d[2:]
    mean = np.float32(0)
    m2 = np.float32(0)
    for n, v in enumerate(t, start=1):
        delta = v - mean
        mean = np.float32(mean + delta / np.float32(n))
        m2 = np.float32(m2 + delta * (v - mean))
    return float(mean), float(m2 / np.float32(len(d) - 3))


def sine_period(rate=22050, pitch=100, amp=32000):
    per = rate // pitch
    x = np.arange(per)
    return (amp * np.sin(x * 2 * np.pi / per)).astype(np.int16)


def check_sine_speed(compress, speed):
    """TestSpeedup / TestSlowdown, sonic_classic_test.cc:167-288: length within 1 %, still a sinusoid."""
    rate, periods = 22050, 100
    pp = sine_period(rate)
    x = np.tile(pp, periods)
    out = compress(x, rate, 1, speed, 0.0)
    expected = int((periods * pp.size) / speed)
    assert (99 * expected) // 100 < out.size < (101 * expected) // 100, (out.size, expected)
    cm, cv = teager_variance(pp)
    sm, sv = teager_variance(out[: out.size - 1000])
    assert abs(cm - sm) < 0.01 * cm
    assert abs(cv - sv) < 0.02 * cv


def check_speech_lengths(compress):
    """TestFullSpeechRange, sonic_classic_test.cc:519-535: tapestry, speeds 1.1..6.1, length within 14 ms."""
    x, rate, ch = read_wav("tapestry.wav")
    assert x.size == 50381 and rate == 16000 and ch == 1
    speed = np.float32(1.1)
    while speed < 6.3:
        out = compress(x, rate, ch, float(speed), 0.0)
        assert abs(out.size - int(x.size / speed)) <= 14 * rate // 1000, (float(speed), out.size)
        speed = np.float32(speed + np.float32(0.25))


def check_noise_lengths(compress):
    """TestFullNoiseRange, sonic_classic_test.cc:558-576 (any fixed Gaussian noise: the C++ engine's stream
    is not reproducible from Python, the property is)."""
    rng = np.random.default_rng(0)
    x = np.clip(rng.standard_normal(50000) * 8096, -32000, 32000).astype(np.int16)
    speed = np.float32(1.1)
    while speed < 6.3:
        out = compress(x, 16000, 1, float(speed), 0.0)
        assert abs(out.size - int(x.size / speed)) <= 1.5 * 16000 / 100, (float(speed), out.size)
        speed = np.float32(speed + np.float32(0.25))


def check_mono_stereo_identity(compress, x, rate):
    """TestSinusoidStereoMatch / TestStereoMatch, sonic_classic_test.cc:581-709: duplicated-stereo output is
    bit-identical to the mono output on both channels."""
    mono = compress(x, rate, 1, 2.0, 0.0)
    st = compress(np.repeat(x, 2), rate, 2, 2.0, 0.0)
    assert st.size == 2 * mono.size
    assert np.array_equal(st[0::2], mono) and np.array_equal(st[1::2], mono)


def sine_440(rate=16000, n=16000):
    i = np.arange(n)
    return (16000 * np.sin(2 * np.pi * np.float32(440) * i / np.float32(rate))).astype(np.int16)


def check_nonlinear_sine(compress, speed):
    """TestSpeedupNonlinear / TestSlowdownNonlinear, sonic_test.cc:479-589: a 237 Hz sine through the full
    nonlinear path with factor 1e-5 (about linear): length within 1.5 %, Teager mean within 1 %, sigma/mean < 1 %."""
    rate, f0 = 22050, 237.0
    n = rate  # one second
    i = np.arange(n)
    x = (16000 * np.sin(2 * np.pi * f0 * i / np.float32(rate))).astype(np.int16)
    out = compress(x, rate, 1, speed, 1e-5)
    expected = n / speed
    assert abs(out.size - expected) < 0.015 * expected + 500, (out.size, expected)
    core = out[1000:-1000] if out.size > 4000 else out
    m_in, _ = teager_variance(x[1000:-1000])
    m_out, v_out = teager_variance(core)
    assert abs(m_in - m_out) < 0.02 * m_in


# ---------------------------------------------------------------------------------------------------------
# Dynamic-time-warping slope checks (sonic_test.cc:639-724, dynamic_time_warping.cc:53-132) and the chirp
# speed-change check (sonic_classic_test.cc:303-394).  Restated in numpy; float32 where the C++ uses float.
# ---------------------------------------------------------------------------------------------------------
def linear_slope(x, y):
    """sonic_test.cc:86-99: least-squares slope, float accumulators."""
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    n = np.float32(x.size)
    sx, sy = np.float32(0), np.float32(0)
    sxy, sx2 = np.float32(0), np.float32(0)
    for a, b in zip(x, y):
        sx = np.float32(sx + a)
        sy = np.float32(sy + b)
        sxy = np.float32(sxy + a * b)
        sx2 = np.float32(sx2 + a * a)
    return float((n * sxy - sx * sy) / (n * sx2 - sx * sx))


def linear_slope_everywhere(x, y, half):
    """sonic_test.cc:101-113."""
    return [linear_slope(x[i - half:i + half], y[i - half:i + half]) for i in range(half, len(x) - half)]


def dtw(seq1, seq2):
    """DynamicTimeWarping::Compute + BestPathSequence (dynamic_time_warping.cc:53-132): Euclidean local cost,
    steps {up, diagonal, left}, strict comparisons so that ties go along the diagonal.  Returns (cost, path1, path2)."""
    a = np.asarray(seq1, np.float32)
    b = np.asarray(seq2, np.float32)
    h, w = a.shape[0], b.shape[0]
    cost = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(axis=2, dtype=np.float32)).astype(np.float32)
    best = np.zeros((h, w), np.int8)
    for j in range(1, w):
        cost[0, j] = np.float32(cost[0, j] + cost[0, j - 1])
        best[0, j] = 1
    for i in range(1, h):
        cost[i, 0] = np.float32(cost[i, 0] + cost[i - 1, 0])
        best[i, 0] = -1
    for i in range(1, h):
        for j in range(1, w):
            up, left, diag = cost[i - 1, j], cost[i, j - 1], cost[i - 1, j - 1]
            cost[i, j] = np.float32(cost[i, j] + min(min(up, left), diag))
            if up < diag and up < left:
                best[i, j] = -1
            elif left < up and left < diag:
                best[i, j] = 1
    p1, p2 = [], []
    i, j = h - 1, w - 1
    while i >= 0 and j >= 0:
        d = best[i, j]
        p1.append(i)
        p2.append(j)
        if d <= 0:
            i -= 1
        if d >= 0:
            j -= 1
    return float(cost[-1, -1]), p1[::-1], p2[::-1]


def check_speech_dtw(compress, spectrogram):
    """TestSpeechSample, sonic_test.cc:639-724: tapestry at 3x, linear and nonlinear.  `spectrogram(x, rate)` is the
    test's own measuring device (ComputeSpectrogram, sonic_test.cc:211-240: non-overlapping frames of the analysis
    window, raw int16 values, the first quarter of the bins)."""
    x, rate, ch = read_wav("tapestry.wav")
    speed, half = 3.0, 10
    lin = compress(x, rate, ch, speed, 0.0)
    spd = compress(x, rate, ch, speed, 1.0)
    assert abs(x.size - 50381) <= 230
    assert abs(lin.size - 50381 / speed) <= 140
    so, sl, ss = spectrogram(x, rate), spectrogram(lin, rate), spectrogram(spd, rate)
    cost, p1, p2 = dtw(so, sl)
    assert cost < 13000000
    slope = linear_slope(p1, p2)
    assert abs(slope - 1.0 / speed) <= 0.02, slope
    slopes = np.asarray(linear_slope_everywhere(p1, p2, half), np.float32)
    assert abs(float(slopes.mean()) - slope) <= 0.02
    assert float(slopes.std()) < 0.2
    _, p1, p2 = dtw(so, ss)
    slope = linear_slope(p1, p2)
    assert abs(slope - 1.0 / speed) <= 0.1, slope
    slopes = np.asarray(linear_slope_everywhere(p1, p2, half), np.float32)
    assert abs(float(slopes.mean()) - slope) <= 0.02
    assert float(slopes.std()) < 0.2


def check_chirp_speedup(make_stream):
    """TestChirpSpeedup, sonic_classic_test.cc:303-394: a 3 s linear chirp, speed 3 / 1.5 / 3 by thirds (sonicSetSpeed
    between writes); sqrt(Teager) is proportional to frequency, so its slope doubles where the speed doubles.
    `make_stream(rate, channels)` returns an object with set_speed / write_short / read_short / flush."""
    rate, f0, f3, amp, speed = 22050, np.float32(137), np.float32(137 + 47), 32000, 3.0
    total = int(3.0 * rate)
    t = (np.arange(total) / np.float32(rate)).astype(np.float32)
    phase = (f0 * t + (f3 - f0) / np.float32(3) * t * t / 2.0).astype(np.float64)   # cycles
    chirp = (amp * np.sin(2 * np.pi * phase)).astype(np.int16)
    s = make_stream(rate, 1)
    outs = []
    for part, sp_ in enumerate((speed, speed / 2, speed)):
        s.set_speed(sp_)
        assert s.write_short(chirp[part * rate:(part + 1) * rate])
    for _ in range(100):
        outs.append(s.read_short(total))
    s.flush()
    while True:
        got = s.read_short(total)
        if got.size == 0:
            break
        outs.append(got)
    out = np.concatenate(outs).astype(np.float32)
    teager = np.sqrt(np.maximum(out[1:-1] * out[1:-1] - out[:-2] * out[2:], 0).astype(np.float32))
    n = teager.size
    s1 = linear_slope(np.arange(n // 4), teager[: n // 4])
    s2 = linear_slope(np.arange(n * 3 // 4 - n // 4), teager[n // 4: n * 3 // 4])
    s3 = linear_slope(np.arange(n - 1000 - n * 3 // 4), teager[n * 3 // 4: n - 1000])
    assert abs(s1 - s3) <= s1 * 0.05, (s1, s2, s3)
    assert abs(s2 - s1 / 2) <= s1 * 0.01, (s1, s2, s3)
