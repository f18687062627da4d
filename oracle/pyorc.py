"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from speedy_amd/ (the product fails loudly without its HIP library instead of falling back).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_float_p = C.POINTER(C.c_float)
c_short_p = C.POINTER(C.c_short)
c_double_p = C.POINTER(C.c_double)

TENSION_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float)
FEATURES_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, c_float_p)
HANDOFF_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float, c_short_p, C.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # ORC_LIB: another build of the same sources, e.g. oracle/liborc_asan.so (make -C oracle asan-test)
    path = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborc.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, i, f, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
    sig = {
        "orc_speedyCreateStream": (vp, [i, i]),
        "orc_speedyDestroyStream": (None, [vp]),
        "orc_speedyInputFrameSize": (i, [vp]),
        "orc_speedyInputFrameStep": (i, [vp]),
        "orc_speedyFFTSize": (i, [vp]),
        "orc_speedyHysteresisFuture": (i, [vp]),
        "orc_speedyHysteresisPast": (i, [vp]),
        "orc_speedyBinToFreq": (f, [vp, i]),
        "orc_speedyFreqToBin": (i, [vp, f]),
        "orc_speedyAddData": (None, [vp, c_float_p, i64]),
        "orc_speedyAddDataShort": (None, [vp, c_short_p, i64]),
        "orc_speedyComputeTension": (i, [vp, i64, c_float_p]),
        "orc_speedyComputeSpeedFromTension": (f, [f, f, f, vp]),
        "orc_speedyGetCurrentTime": (i64, [vp]),
        "orc_speedySpectrogram": (c_float_p, [vp, c_float_p]),
        "orc_speedyEvaluateHysteresis": (f, [vp, i64]),
        "orc_speedyAddToHysteresisBuffer": (None, [vp, f, i64]),
        "orc_speedyComputeSpectralDifference": (None, [vp, c_float_p, c_float_p, i64]),
        "orc_speedyComputeLocalEnergy": (None, [vp, c_float_p, i64]),
        "orc_speedyGetSpectrogramAtTime": (c_float_p, [vp, i64]),
        "orc_speedyPreemphasisFilter": (None, [vp, c_float_p, i]),
        "orc_speedyGetNormalizedSpectrogram": (c_float_p, [vp]),
        "orc_speedyGetSpectrogram": (c_float_p, [vp]),
        "orc_speedyGetInternalState": (c_float_p, [vp]),
        "orc_speedyGetEnergyCompressed": (f, [vp]),
        "orc_speedyGetSpeechChanges": (f, [vp]),
        "orc_speedyNormalizeByEnergy": (f, [c_float_p, c_float_p, i]),
        "orc_log": (C.c_double, [C.c_double]),
        "orc_log_v2_f32": (C.c_double, [C.c_float]),
        "orc_logcheck_run": (C.c_int, [C.c_uint, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "orc_log_spec": (C.c_double, [C.c_double]),
        "orc_set_log_spec": (None, [C.c_int]),
        "orc_get_log_spec": (C.c_int, []),
        "orc_set_dft_spec": (None, [C.c_int]),
        "orc_get_dft_spec": (C.c_int, []),
        "orc_set_twiddle_spec": (None, [C.c_int]),
        "orc_get_twiddle_spec": (C.c_int, []),
        "orc_twiddle_entry": (None, [C.c_long, C.c_long, c_double_p, c_double_p]),
        "orc_twiddle_hash": (C.c_ulonglong, [C.c_long, C.c_long]),
        "orc_dft_forward": (None, [i, c_double_p, c_double_p]),
        "orc_dft_naive": (None, [i, c_double_p, c_double_p]),
        "orc_spectrum_magnitudes": (None, [i, c_float_p, c_float_p]),
        "orc_sonicIntCreateStream": (vp, [i, i]),
        "orc_sonicIntDestroyStream": (None, [vp]),
        "orc_sonicIntGetNumChannels": (i, [vp]),
        "orc_sonicIntGetSpeed": (f, [vp]),
        "orc_sonicIntSetSpeed": (None, [vp, f]),
        "orc_sonicIntSetRate": (None, [vp, f]),
        "orc_sonicIntWriteShortToStream": (i, [vp, c_short_p, i]),
        "orc_sonicIntWriteFloatToStream": (i, [vp, c_float_p, i]),
        "orc_sonicIntReadShortFromStream": (i, [vp, c_short_p, i]),
        "orc_sonicIntReadFloatFromStream": (i, [vp, c_float_p, i]),
        "orc_sonicIntFlushStream": (i, [vp]),
        "orc_sonicIntSamplesAvailable": (i, [vp]),
        "orc_sonicIntStepCount": (C.c_long, [vp]),
        "orc_sonicIntSetPeriodLog": (None, [vp, C.POINTER(C.c_int), C.c_long]),
        "orc_sonicCreateStream": (vp, [i, i, i]),
        "orc_sonicDestroyStream": (None, [vp]),
        "orc_sonicWriteShortToStream": (i, [vp, c_short_p, i]),
        "orc_sonicReadShortFromStream": (i, [vp, c_short_p, i]),
        "orc_sonicWriteFloatToStream": (i, [vp, c_float_p, i]),
        "orc_sonicReadFloatFromStream": (i, [vp, c_float_p, i]),
        "orc_sonicSetRate": (None, [vp, f]),
        "orc_sonicSetSpeed": (None, [vp, f]),
        "orc_sonicFlushStream": (i, [vp]),
        "orc_sonicEnableNonlinearSpeedup": (None, [vp, f]),
        "orc_sonicSetDurationFeedbackStrength": (None, [vp, f]),
        "orc_getSonicBufferSize": (i, [vp]),
        "orc_sonicSpectrogramSize": (i, [vp]),
        "orc_sonicTensionCallback": (None, [vp, TENSION_FN]),
        "orc_sonicSpeedCallback": (None, [vp, TENSION_FN]),
        "orc_sonicFeaturesCallback": (None, [vp, FEATURES_FN]),
        "orc_sonicSpectrogramCallback": (None, [vp, FEATURES_FN]),
        "orc_sonicNormalizedSpectrogramCallback": (None, [vp, FEATURES_FN]),
        "orc_sonicHandoffCallback": (None, [vp, HANDOFF_FN]),
        "orc_compress_sound": (C.c_long, [c_short_p, C.c_long, i, i, f, f, f, i, i, c_short_p, C.c_long,
                                          c_float_p, c_float_p, c_float_p, C.c_long, C.POINTER(C.c_long)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def sptr(a):
    return a.ctypes.data_as(c_short_p)


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def np_from(ptr, n):
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class Speedy:
    """Thin OO view of the orc_speedy* unit-level API (mirrors reference speedy.h)."""

    def __init__(self, sample_rate, match_matlab=True):
        self.L = lib()
        self.h = self.L.orc_speedyCreateStream(sample_rate, int(match_matlab))
        assert self.h

    def close(self):
        if self.h:
            self.L.orc_speedyDestroyStream(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_speedy" + name)
        return lambda *a: fn(self.h, *a)

    @property
    def frame_size(self):
        return self.L.orc_speedyInputFrameSize(self.h)

    @property
    def frame_step(self):
        return self.L.orc_speedyInputFrameStep(self.h)

    @property
    def fft_size(self):
        return self.L.orc_speedyFFTSize(self.h)

    def add_data(self, x, t):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.size >= self.frame_size
        self.L.orc_speedyAddData(self.h, fptr(x), t)

    def add_data_short(self, x, t):
        x = np.ascontiguousarray(x, dtype=np.int16)
        assert x.size >= self.frame_size
        self.L.orc_speedyAddDataShort(self.h, sptr(x), t)

    def compute_tension(self, t):
        out = C.c_float(0)
        ok = self.L.orc_speedyComputeTension(self.h, t, C.byref(out))
        return bool(ok), out.value

    def spectrogram(self, x=None):
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            self.L.orc_speedySpectrogram(self.h, fptr(x))
        return np_from(self.L.orc_speedyGetSpectrogram(self.h), self.fft_size)

    def spectrogram_at(self, t):
        return np_from(self.L.orc_speedyGetSpectrogramAtTime(self.h, t), self.fft_size)

    def normalized(self):
        return np_from(self.L.orc_speedyGetNormalizedSpectrogram(self.h), self.fft_size // 2)

    def features(self):
        return np_from(self.L.orc_speedyGetInternalState(self.h), 15)


def compress_sound(x, sample_rate, channels, speed, nonlinear=1.0, feedback=0.0, match_matlab=False,
                   chunk=1000, taps=True):
    """speedy_wave.cc:154-242 through the oracle.  x: int16 [n*channels] interleaved.
    Returns dict(out=int16[...], tension, speed, features)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.int16)
    n_in = x.size // channels
    cap = n_in + 8 * sample_rate
    if speed < 1:   # a nonlinear slow-down may run at the speed floor 0.01 for whole passages (speedy.c:776), and
        # one pitch step at speed s emits up to 2/s frames per frame consumed; the flush padding is stretched too
        s_min = 0.01 if nonlinear > 0 else max(float(speed), 1e-4)
        cap = int((n_in + 4 * (sample_rate // 65) + 64) * 2.0 / s_min) + 8 * sample_rate
    out = np.zeros(cap * channels, dtype=np.int16)
    ncap = n_in // max(1, sample_rate // 100) + 16
    ten = np.zeros(ncap, np.float32)
    spd = np.zeros(ncap, np.float32)
    fea = np.zeros((ncap, 15), np.float32)
    nt = C.c_long(0)
    n = L.orc_compress_sound(sptr(x), n_in, sample_rate, channels, speed, nonlinear, feedback,
                             int(match_matlab), chunk, sptr(out), cap,
                             fptr(ten) if taps else None, fptr(spd) if taps else None,
                             fptr(fea) if taps else None, ncap, C.byref(nt))
    if n < 0:
        raise RuntimeError("oracle output overflow")
    k = nt.value
    return dict(out=out[: n * channels].copy(), tension=ten[:k].copy(), speed=spd[:k].copy(),
                features=fea[:k].copy())


_BENCH = None


def bench_lib():
    """oracle/liborc_bench.so: the same sources plus the POSIX-thread runner (oracle/orc_bench.c), built for this machine."""
    global _BENCH
    if _BENCH is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc_bench.so"])
        L = C.CDLL(os.path.join(_HERE, "liborc_bench.so"))
        L.orc_bench_run.restype = C.c_double
        L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _BENCH = L
    return _BENCH


def crc_streams(streams, sample_rate, channels, speed, nonlinear=1.0, feedback=0.0, match_matlab=False, chunk=1000,
                threads=None):
    """The speedy_wave.cc loop of compress_sound over equally long streams of one kind on `threads` POSIX threads (one
    stream per task): returns (wall seconds, frames produced per stream, CRC-32 of each stream's int16 output bytes)."""
    L = bench_lib()
    n = np.asarray(streams[0]).size // channels
    assert all(np.asarray(x).size == n * channels for x in streams)
    buf = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int16).ravel() for x in streams]), np.int16)
    frames = (C.c_long * len(streams))()
    crcs = (C.c_uint32 * len(streams))()
    if threads is None:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(threads, 16, len(streams)))
    dt = L.orc_bench_run(buf.ctypes.data, n, len(streams), int(sample_rate), int(channels), float(speed), float(nonlinear),
                         float(feedback), int(bool(match_matlab)), int(chunk), int(threads), frames, crcs)
    return dt, list(frames), list(crcs)
