"""Python view of the unit-level API (include/speedy.h = reference speedy.h:61-100) of the HIP library, with the same
method names as the oracle's view (oracle/pyorc.py::Speedy) so that the reference's unit tests, restated once, run
against either."""
import ctypes as C

import numpy as np

from ._lib import c_float_p, c_short_p, lib


class Speedy:
    def __init__(self, sample_rate, match_matlab=True):
        self.L = lib()
        self.L.speedyHipSetMatchMatlab(int(bool(match_matlab)))
        self.h = self.L.speedyCreateStream(int(sample_rate))
        if not self.h:
            raise RuntimeError("speedyCreateStream: " + self.L.speedyHipLastError().decode())

    def close(self):
        if self.h:
            self.L.speedyDestroyStream(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def frame_size(self):
        return self.L.speedyInputFrameSize(self.h)

    @property
    def frame_step(self):
        return self.L.speedyInputFrameStep(self.h)

    @property
    def fft_size(self):
        return self.L.speedyFFTSize(self.h)

    def HysteresisFuture(self):
        return self.L.speedyHipHysteresisFuture(self.h)

    def HysteresisPast(self):
        return self.L.speedyHipHysteresisPast(self.h)

    def GetCurrentTime(self):
        return self.L.speedyGetCurrentTime(self.h)

    def FreqToBin(self, f):
        return self.L.speedyFreqToBin(self.h, float(f))

    def BinToFreq(self, b):
        return self.L.speedyBinToFreq(self.h, int(b))

    def GetEnergyCompressed(self):
        return self.L.speedyGetEnergyCompressed(self.h)

    def GetSpeechChanges(self):
        return self.L.speedyGetSpeechChanges(self.h)

    def _arr(self, ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def add_data(self, x, t):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.size >= self.frame_size
        self.L.speedyAddData(self.h, x.ctypes.data_as(c_float_p), int(t))

    def add_data_short(self, x, t):
        x = np.ascontiguousarray(x, dtype=np.int16)
        assert x.size >= self.frame_size
        self.L.speedyAddDataShort(self.h, x.ctypes.data_as(c_short_p), int(t))

    def compute_tension(self, t):
        out = C.c_float(0)
        ok = self.L.speedyComputeTension(self.h, int(t), C.byref(out))
        return bool(ok), out.value

    def speed_from_tension(self, tension, rg, fb):
        return self.L.speedyComputeSpeedFromTension(float(tension), float(rg), float(fb), self.h)

    def spectrogram(self, x=None):
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            return self._arr(self.L.speedySpectrogram(self.h, x.ctypes.data_as(c_float_p)), self.fft_size)
        return self._arr(self.L.speedyGetSpectrogram(self.h), self.fft_size)

    def spectrogram_at(self, t):
        return self._arr(self.L.speedyGetSpectrogramAtTime(self.h, int(t)), self.fft_size)

    def normalized(self):
        return self._arr(self.L.speedyGetNormalizedSpectrogram(self.h), self.fft_size // 2)

    def features(self):
        return self._arr(self.L.speedyGetInternalState(self.h), 15)

    # ---- the hooks between stages (speedy.h:102-133), named as the oracle's Python view names them ----
    def PreemphasisFilter(self, ptr, n):
        self.L.speedyPreemphasisFilter(self.h, ptr, int(n))

    def AddToHysteresisBuffer(self, v, t):
        self.L.speedyAddToHysteresisBuffer(self.h, float(v), int(t))

    def EvaluateHysteresis(self, t):
        return self.L.speedyEvaluateHysteresis(self.h, int(t))

    def ComputeLocalEnergy(self, ptr, t):
        self.L.speedyComputeLocalEnergy(self.h, ptr, int(t))

    def ComputeSpectralDifference(self, cur, last, t):
        self.L.speedyComputeSpectralDifference(self.h, cur, last, int(t))

    def SaveSpectrogramData(self, ptr, t):
        self.L.speedySaveSpectrogramData(self.h, ptr, int(t))
