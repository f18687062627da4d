"""GPU: the speedy_wave-style C++ CLI (tools/speedy_wave_hip.cpp, plain g++ against include/sonic2.h) on the
reference's tapestry.wav = BASELINE configs[0], compared with the oracle's compress_sound."""
import os
import struct
import subprocess

import numpy as np
import pytest

from util import GOLDEN, read_wav

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "speedy_amd", "lib", "speedy_wave_hip")


def _read_out(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[36:40] == b"data"
    n = struct.unpack("<I", b[40:44])[0]
    return np.frombuffer(b[44:44 + n], dtype="<i2")


@pytest.mark.parametrize("args,nl,fb,mm", [
    (["--speed", "3.5"], 1.0, 0.0, False),                                   # speedy_wave defaults
    (["--speed", "2.0", "--linear"], 0.0, 0.0, False),
    (["--speed", "3.0", "--duration_feedback_strength", "0.1", "--match_matlab"], 1.0, 0.1, True),
])
def test_cli_matches_oracle(orc, tmp_path, args, nl, fb, mm):
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "cli"])
    out = str(tmp_path / "out.wav")
    tf, sf = str(tmp_path / "t.txt"), str(tmp_path / "s.txt")
    r = subprocess.run([CLI, "--input", os.path.join(GOLDEN, "tapestry.wav"), "--output", out, "--tension_file", tf,
                        "--speed_file", sf] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Read 1 channel data at a sample rate of 16000." in r.stdout
    x, rate, ch = read_wav("tapestry.wav")
    speed = float(args[1])
    ref = orc.compress_sound(x, rate, ch, speed, nl, fb, mm, chunk=1000)
    got = _read_out(out)
    assert np.array_equal(got, ref["out"])
    if nl:
        t = np.loadtxt(tf, dtype=np.float64)
        s = np.loadtxt(sf, dtype=np.float64)
        assert t.size == ref["tension"].size and s.size == ref["speed"].size
        assert np.allclose(t, ref["tension"], rtol=1e-5, atol=1e-6)  # %g keeps 6 significant digits
        assert np.allclose(s, ref["speed"], rtol=1e-5, atol=1e-6)
