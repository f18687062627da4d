"""Shared test helpers: PCM16 WAV fixture reader and the synthetic speech-like generator."""
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_wav(name):
    """Minimal RIFF/PCM16 reader for the reference's fixtures (replaces libsonic's wave.c, which is not
    in the reference tree).  Returns (int16 samples interleaved, sample_rate, channels)."""
    with open(os.path.join(GOLDEN, name), "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, rate, ch, data = 12, None, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            assert fmt == 1 and bits == 16
        elif cid == b"data":
            data = np.frombuffer(body[: len(body) // 2 * 2], dtype="<i2").astype(np.int16)
        pos += 8 + size + (size & 1)
    return data, rate, ch


def matlab_fixture():
    return np.load(os.path.join(GOLDEN, "tapestry22050_matlab.npz"))
