"""speedy_test.cc:859-1057 (TestTapestryFeatureComputations) restated against the oracle: the reference's
Matlab matrices for tapestry22050.wav pin shape and alignment at SNR level (the reference C itself only
reaches ~27.6 dB against them), with per-feature best delays and SNR thresholds."""
import math

import numpy as np

from util import matlab_fixture, read_wav
from test_oracle_kat import cround


def _snr(signal, estimate):
    signal = np.asarray(signal, np.float32)
    estimate = np.asarray(estimate, np.float32)
    err = np.float32(0)
    d = signal - estimate
    return float(np.sum(signal * signal, dtype=np.float32) / np.sum(d * d, dtype=np.float32))


def _portion(a, start, count):
    end = min(start + count, len(a))
    return a[start:end - 1]  # speedy_test.cc:831-838 drops the last element


def _xcorr(a, b, num_delays):
    out = []
    for delay in range(-num_delays, num_delays + 1):
        if delay < 0:
            n = len(a) + delay
            out.append(_snr(_portion(a, -delay, n), _portion(b, 0, n)))
        else:
            n = len(a) - delay
            out.append(_snr(_portion(a, 0, n), _portion(b, delay, n)))
    return out


def run_unit_level(orc, x, rate, match_matlab=True):
    """The speedyAddData / speedyComputeTension loop of speedy_test.cc:911-935."""
    s = orc.Speedy(rate, match_matlab)
    W = s.frame_size
    step = np.float32(rate / np.float32(100))
    frame_count = int((x.size - W) / step + 1)
    spec, norm, feat, tension = [], [], [], []
    out_t = 0
    half = s.fft_size // 2
    for t in range(frame_count):
        start = cround(np.float32(t) * step)
        s.add_data(x[start:start + W], t)
        spec.append(s.spectrogram()[:half])
        ok, v = s.compute_tension(out_t)
        if ok:
            tension.append(v)
            norm.append(s.normalized())
            feat.append(s.features())
            out_t += 1
    return np.array(spec), np.array(norm), np.array(feat), np.array(tension, np.float32)


def test_tapestry_feature_computations(orc):
    fx = matlab_fixture()
    exp_spec, exp_norm, exp_feat = fx["spectrogram"], fx["normalized"], fx["features"]
    assert exp_spec.shape == (314, 330) and exp_norm.shape == (314, 330) and exp_feat.shape == (314, 12)
    data, rate, ch = read_wav("tapestry22050.wav")
    assert data.size == 69431 and ch == 1 and rate == 22050
    x = (data.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    assert abs(x.max() - 0.41369) < 0.001
    spec, norm, feat, tension = run_unit_level(orc, x, rate, True)
    assert spec.shape[0] == 314 and norm.shape[0] == 306 and feat.shape[0] == 306

    col, max_delay = 150, 20
    snrs = [10 * math.log10(_snr(exp_spec[col], spec[col + d])) for d in range(-max_delay, max_delay)]
    assert snrs[max_delay] > 27
    assert all(snrs[max_delay] > v for i, v in enumerate(snrs) if i != max_delay)

    for fr in range(norm.shape[0]):
        assert abs(float(np.sum(norm[fr] * norm[fr], dtype=np.float32)) - 1) < 4e-3
    nsnrs = [10 * math.log10(_snr(exp_norm[col], norm[col + d])) for d in range(-max_delay, max_delay)]
    assert nsnrs[max_delay] > 27
    assert all(nsnrs[max_delay] > v for i, v in enumerate(nsnrs) if i != max_delay)

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
