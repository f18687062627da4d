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
