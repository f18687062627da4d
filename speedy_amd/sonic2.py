"""Python view of the reference-compatible streaming API (include/sonic2.h), for tests that read like the
reference's own (sonic_test.cc, speedy_test.cc:653-692)."""
import ctypes as C

import numpy as np

from ._lib import FEATURES_FN, TENSION_FN, c_float_p, c_short_p, lib


class SonicStream:
    def __init__(self, sample_rate, channels, match_matlab=False, coalesce=None):
        """coalesce: None = the process-wide default (speedyHipSetCoalescing / SPX_NO_POOL), False = this handle runs its
        own launch sequence per write, True = coalesced.  The choice is per handle; no process-wide switch is touched."""
        self.L = lib()
        self.h = self.L.speedyHipCreateSonicStreamEx(int(sample_rate), int(channels), int(bool(match_matlab)),
                                                     -1 if coalesce is None else int(bool(coalesce)))
        if not self.h:
            raise RuntimeError("sonicCreateStream: " + self.L.speedyHipLastError().decode())
        self.channels = channels
        self._keep = []

    def close(self):
        if self.h:
            self.L.sonicDestroyStream(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_speed(self, v):
        self.L.sonicSetSpeed(self.h, float(v))

    def set_rate(self, v):
        self.L.sonicSetRate(self.h, float(v))

    def enable_nonlinear(self, v):
        self.L.sonicEnableNonlinearSpeedup(self.h, float(v))

    def set_feedback(self, v):
        self.L.sonicSetDurationFeedbackStrength(self.h, float(v))

    def buffer_size(self):
        return self.L.getSonicBufferSize(self.h)

    def spectrogram_size(self):
        return self.L.sonicSpectrogramSize(self.h)

    def write_short(self, x):
        x = np.ascontiguousarray(x, np.int16)
        return self.L.sonicWriteShortToStream(self.h, x.ctypes.data_as(c_short_p), x.size // self.channels)

    def write_float(self, x):
        x = np.ascontiguousarray(x, np.float32)
        return self.L.sonicWriteFloatToStream(self.h, x.ctypes.data_as(c_float_p), x.size // self.channels)

    def read_short(self, max_frames):
        buf = np.zeros(max_frames * self.channels, np.int16)
        n = self.L.sonicReadShortFromStream(self.h, buf.ctypes.data_as(c_short_p), max_frames)
        return buf[: n * self.channels]

    def read_float(self, max_frames):
        buf = np.zeros(max_frames * self.channels, np.float32)
        n = self.L.sonicReadFloatFromStream(self.h, buf.ctypes.data_as(c_float_p), max_frames)
        return buf[: n * self.channels]

    def flush(self):
        return self.L.sonicFlushStream(self.h)

    def available(self):
        return self.L.sonicSamplesAvailable(self.h)

    # the TSM stage alone (sonicInt*, sonic_test.cc:735-750): no ring, no analysis, whatever the nonlinear factor
    def int_write_short(self, x):
        x = np.ascontiguousarray(x, np.int16)
        return self.L.sonicIntWriteShortToStream(self.h, x.ctypes.data_as(c_short_p), x.size // self.channels)

    def int_flush(self):
        return self.L.sonicIntFlushStream(self.h)

    def int_set_speed(self, v):
        self.L.sonicIntSetSpeed(self.h, float(v))

    def on_tension(self, fn):
        cb = TENSION_FN(lambda s, t, v: fn(t, v))
        self._keep.append(cb)
        self.L.sonicTensionCallback(self.h, cb)

    def on_speed(self, fn):
        cb = TENSION_FN(lambda s, t, v: fn(t, v))
        self._keep.append(cb)
        self.L.sonicSpeedCallback(self.h, cb)

    def on_features(self, fn):
        cb = FEATURES_FN(lambda s, t, p: fn(t, np.ctypeslib.as_array(p, shape=(15,)).copy()))
        self._keep.append(cb)
        self.L.sonicFeaturesCallback(self.h, cb)

    def on_spectrogram(self, fn):
        n = self.spectrogram_size()
        cb = FEATURES_FN(lambda s, t, p: fn(t, np.ctypeslib.as_array(p, shape=(n,)).copy()))
        self._keep.append(cb)
        self.L.sonicSpectrogramCallback(self.h, cb)

    def on_normalized(self, fn):
        n = self.spectrogram_size()   # fft_size floats like the reference's buffer; bins >= fft_size/2 are zero
        cb = FEATURES_FN(lambda s, t, p: fn(t, np.ctypeslib.as_array(p, shape=(n,)).copy()))
        self._keep.append(cb)
        self.L.sonicNormalizedSpectrogramCallback(self.h, cb)


def pool_stats():
    """(launch sequences run, stream jobs served) by the current device's pool."""
    L = lib()
    r, j = C.c_ulonglong(0), C.c_ulonglong(0)
    L.speedyHipPoolStats(C.byref(r), C.byref(j))
    return r.value, j.value


def time_compress(x, sample_rate, channels, speed, nonlinear=0.0, feedback=None, chunk=1000, match_matlab=False,
                  taps=None, coalesce=None):
    """The write/read/flush/drain loop of compress_sound (speedy_wave.cc:199-231) and TimeCompressVector
    (sonic_test.cc:364-403).  Returns the concatenated int16 output."""
    s = SonicStream(sample_rate, channels, match_matlab, coalesce)
    s.set_speed(speed)
    s.enable_nonlinear(nonlinear)
    if feedback is not None:
        s.set_feedback(feedback)
    if taps is not None:
        taps(s)
    x = np.ascontiguousarray(x, np.int16)
    n = x.size // channels
    out = []
    for pos in range(0, n, chunk):
        seg = x[pos * channels:(pos + chunk) * channels]
        assert s.write_short(seg) == 1
        out.append(s.read_short(chunk))
    s.flush()
    while True:
        got = s.read_short(chunk)
        if got.size == 0:
            break
        out.append(got)
    s.close()
    return np.concatenate(out) if out else np.zeros(0, np.int16)
