"""The oracle's own building blocks: the mixed-radix DFT against the O(n^2) definition (the role of the
reference's kiss_fft_test.cc:50-85) and orc_log against libm."""
import math

import numpy as np
import pytest


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 11, 15, 16, 120, 165, 240, 330, 360, 480, 661, 720])
def test_dft_matches_definition(orc, n):
    L = orc.lib()
    rng = np.random.default_rng(n)
    x = rng.standard_normal(2 * n)
    a = np.zeros(2 * n)
    b = np.zeros(2 * n)
    L.orc_dft_forward(n, orc.dptr(x), orc.dptr(a))
    L.orc_dft_naive(n, orc.dptr(x), orc.dptr(b))
    scale = np.abs(b).max() + 1.0
    assert np.abs(a - b).max() / scale < 2e-14 * max(1, math.log2(n + 1))


def test_kiss_fft_cosine_8pt(orc):
    """kiss_fft_test.cc:50-85: 8-point cosine -> bins 1 and 7 hold N/2, the rest 0."""
    L = orc.lib()
    n = 8
    x = np.zeros(2 * n)
    x[0::2] = np.cos(2 * np.pi * np.arange(n) / n)
    y = np.zeros(2 * n)
    L.orc_dft_forward(n, orc.dptr(x), orc.dptr(y))
    mag = np.hypot(y[0::2], y[1::2])
    assert abs(mag[1] - 4) < 1e-12 and abs(mag[7] - 4) < 1e-12
    assert np.abs(np.delete(mag, [1, 7])).max() < 1e-12


@pytest.mark.parametrize("W", [120, 165, 240, 330, 360, 661])
def test_packed_real_spectrum(orc, W):
    """|DFT_2W| of the zero-padded real frame via the packed W-point transform == numpy's rfft (double)."""
    L = orc.lib()
    rng = np.random.default_rng(W)
    x = rng.standard_normal(W).astype(np.float32)
    mags = np.zeros(2 * W, np.float32)
    L.orc_spectrum_magnitudes(W, orc.fptr(x), orc.fptr(mags))
    ref = np.abs(np.fft.fft(np.concatenate([x.astype(np.float64), np.zeros(W)])))
    assert np.abs(mags - ref).max() <= 2e-7 * ref.max() + 1e-7
    assert np.array_equal(mags[W + 1:], mags[1:W][::-1])


def test_log_within_one_ulp_of_libm(orc):
    L = orc.lib()
    rng = np.random.default_rng(7)
    xs = np.concatenate([np.exp(rng.uniform(-40, 40, 20000)), 1 + rng.uniform(-1e-6, 1e-6, 2000),
                         rng.uniform(0.5, 2.0, 20000), [1.0, 2.0, 0.5, 1e-300, 1e300]])
    worst = 0.0
    for v in xs:
        a, b = L.orc_log(float(v)), math.log(float(v))
        ulp = math.ulp(b) if b != 0 else 5e-324
        worst = max(worst, abs(a - b) / ulp)
    assert worst <= 1.0, worst


def test_log_spec_v2_against_libm_on_every_positive_normal_float(orc):
    """Log spec v2 (oracle/orc_speedy.c orc_log_v2_f32: what the analysis computes for its float quotients) over the
    WHOLE domain, 2 130 706 432 arguments, against glibc's log: at most 1 ulp apart everywhere -- and equal for all but a few ten
    thousand (v1, the fdlibm sequence, differs from glibc for about a quarter of them).  oracle/orc_logcheck.c, threaded."""
    import ctypes as C
    import os
    L = orc.lib()
    hist = (C.c_uint64 * 4)()
    worst = (C.c_uint64 * 2)()
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    assert L.orc_logcheck_run(8, 2040, threads, 1, None, hist, worst) == 0
    assert sum(hist) == 2130706432
    assert hist[2] == 0 and hist[3] == 0, (list(hist), [hex(v) for v in worst])
    assert hist[1] < 100000, list(hist)


def test_log_spec_dispatch(orc):
    """orc_log_spec: v2 for positive normal floats, v1 (orc_log) for every other double; exact values at the table's fixed points."""
    L = orc.lib()
    assert L.orc_get_log_spec() == 2
    assert L.orc_log_spec(1.0) == 0.0 and L.orc_log_v2_f32(1.0) == 0.0
    for v in (0.5, 2.0, 1.5, 0.7, 1e-19, 3e19, 1.0000001192092896, 0.9999999403953552):
        a, b = L.orc_log_spec(float(np.float32(v))), math.log(float(np.float32(v)))
        assert abs(a - b) <= math.ulp(b), v
    for v in (1e-300, 1e300, 1.0 + 2.0 ** -40, 5e-324, float(np.float32(1e-40))):     # not floats / not normal floats: the fdlibm sequence
        assert L.orc_log_spec(v) == L.orc_log(v), v
    assert L.orc_log_spec(0.0) == -math.inf and math.isnan(L.orc_log_spec(-1.0))
    L.orc_set_log_spec(1)
    try:
        assert L.orc_log_spec(1.5) == L.orc_log(1.5)
    finally:
        L.orc_set_log_spec(2)


@pytest.fixture
def libm_twiddles(orc):
    L = orc.lib()
    L.orc_set_twiddle_spec(2)
    yield
    L.orc_set_twiddle_spec(3)


@pytest.mark.parametrize("n", [7, 11, 13, 31])
def test_twiddles_are_one_sincos_call(orc, n, libm_twiddles):
    """(Twiddle spec 2, the tables of round 5 -- still selectable: orc_set_twiddle_spec(2); spec 3, the default since round 6, takes no
    libm call at all: tests/test_oracle_twiddle.py.)  DFT spec: a twiddle (cos, -sin)(2 pi t / n) is ONE glibc sincos call -- sincos and the separate sin / cos
    round a few entries differently in the last bit, and a compiler may or may not merge the pair (gcc does, clang does not: the GPU's
    tables and the oracle's once differed that way).  An odd-prime transform of the unit impulse at index 1 hands the table back
    unchanged -- u_1 = 1, P_j = fma(c, 1, 0), Q_j = s * 1 -- so the oracle's entries can be compared with libm's sincos bit for bit
    (outputs j and n - j are the conjugate pair built from entry min(j, n - j): the butterfly reads half the table)."""
    import ctypes as C
    import ctypes.util
    import math
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    libm.sincos.restype = None
    x = np.zeros(2 * n)
    x[2] = 1.0                                   # (re, im) interleaved: element 1 = 1 + 0i
    out = np.zeros(2 * n)
    orc.lib().orc_dft_forward(n, x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    for j in range(n):
        s, c = C.c_double(), C.c_double()
        t = min(j, n - j)
        libm.sincos(2.0 * math.pi * t / n, C.byref(s), C.byref(c))
        im = -s.value if j == t else s.value
        assert out[2 * j] == c.value and out[2 * j + 1] == im, (n, j, out[2 * j], c.value, out[2 * j + 1], im)


@pytest.mark.parametrize("n,t", [(240, 32), (360, 48), (480, 64), (720, 96)])
def test_twiddle_entries_that_tell_sincos_from_sin(orc, n, t, libm_twiddles):
    """(Twiddle spec 2.)  The same pin on entries where this glibc's sincos and sin disagree in the last bit (the angle 2 pi 2/15).  In a composite
    transform whose first radix leaves more than t outputs per branch, output t of the unit impulse at index 1 is table entry t
    times 1 + 0i through a butterfly of zeros: exact.  Where the two libm routes do disagree the oracle must side with sincos."""
    import ctypes as C
    import ctypes.util
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    libm.sincos.restype = None
    x = np.zeros(2 * n)
    x[2] = 1.0
    out = np.zeros(2 * n)
    orc.lib().orc_dft_forward(n, orc.dptr(x), orc.dptr(out))
    a = 2.0 * math.pi * t / n
    s, c = C.c_double(), C.c_double()
    libm.sincos(a, C.byref(s), C.byref(c))
    assert out[2 * t] == c.value and out[2 * t + 1] == -s.value
    if s.value != math.sin(a):
        assert out[2 * t + 1] != -math.sin(a)
