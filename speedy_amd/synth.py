"""Deterministic synthetic "speech-like" int16 signals: a pitch-modulated harmonic
source under a syllable envelope with silent gaps, plus noise at -30 dB; peak about 0.4 FS.  Gates, both
pitch-search stages and both speed branches get exercised.  numpy only, same on every host."""
import numpy as np


def speech_like(n, sample_rate, seed, channels=1):
    if n <= 0:
        return np.zeros(0, np.int16)
    rng = np.random.default_rng(1234 + seed)
    t = np.arange(n) / float(sample_rate)
    # f0 random walk 90..250 Hz, updated every 20 ms
    hop = max(1, sample_rate // 50)
    steps = rng.normal(0, 6.0, n // hop + 2)
    f0c = np.clip(150 + np.cumsum(steps), 90, 250)
    f0 = np.interp(np.arange(n), np.arange(f0c.size) * hop, f0c)
    phase = 2 * np.pi * np.cumsum(f0) / sample_rate
    sig = np.zeros(n)
    for h in range(1, 11):
        sig += np.sin(h * phase + rng.uniform(0, 2 * np.pi)) / h
    # syllable envelope 3..6 Hz with about 20 % silent gaps
    syl = rng.uniform(3, 6)
    env = 0.5 * (1 - np.cos(2 * np.pi * syl * t + rng.uniform(0, 2 * np.pi)))
    gate_len = max(1, sample_rate // 4)
    gates = (rng.uniform(0, 1, n // gate_len + 2) > 0.2).astype(float)
    gate = np.interp(np.arange(n), np.arange(gates.size) * gate_len, gates)
    sig = sig * env * gate
    sig = sig / (np.abs(sig).max() + 1e-9) * 0.4
    sig += rng.normal(0, 0.4 * 10 ** (-30 / 20), n)
    mono = np.clip(np.round(sig * 32768.0), -32768, 32767).astype(np.int16)
    if channels == 1:
        return mono
    out = np.empty((n, channels), np.int16)
    for c in range(channels):
        off = rng.integers(-40, 40)
        out[:, c] = np.clip(mono.astype(np.int32) + off, -32768, 32767).astype(np.int16)
    return out.reshape(-1)
