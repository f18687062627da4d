"""CPU: the oracle's run-time FFTW probe.  The image has no libfftw3, so the plumbing is driven with a
TEST DOUBLE (tests/fftw_double.c: the three entry points over a naive double-precision DFT) in a child process whose
LD_LIBRARY_PATH holds it: the library is found, the backend switches, spectra equal the port's to float precision, a whole stream's
tension stays within north_star's 1e-4 and the frame count is the same.  Without it: "absent", backend 0."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import pyorc
L = pyorc.lib()
L.orc_fftw_available.restype = int
print("AVAILABLE", L.orc_fftw_available())
print("BACKEND", L.orc_set_fft_backend(1))
from speedy_amd.synth import speech_like
x = speech_like(16000, 16000, seed=3)
a = pyorc.compress_sound(x, 16000, 1, 3.5, 1.0, 0.0, False)
W = 240
rng = np.random.default_rng(0)
fr = (rng.standard_normal(W) * 0.1).astype(np.float32)
ma = np.zeros(2 * W, np.float32); mb = np.zeros(2 * W, np.float32)
L.orc_spectrum_magnitudes(W, pyorc.fptr(fr), pyorc.fptr(ma))
print("BACKEND0", L.orc_set_fft_backend(0))
L.orc_spectrum_magnitudes(W, pyorc.fptr(fr), pyorc.fptr(mb))
b = pyorc.compress_sound(x, 16000, 1, 3.5, 1.0, 0.0, False)
print("SPEC", float(np.abs(ma - mb).max() / mb.max()))
print("TENSION", float(np.abs(a["tension"] - b["tension"]).max()) if a["tension"].size == b["tension"].size else -1.0)
print("FRAMES", a["out"].size, b["out"].size)
"""


def _child(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln and ln.split()[0].isupper()}


def test_absent_on_this_image(orc):
    got = _child({})
    if got["AVAILABLE"] == ["1"]:
        import pytest
        pytest.skip("this box has a real libfftw3")
    assert got["BACKEND"] == ["0"] and got["SPEC"] == ["0.0"]


def test_backend_with_a_test_double(orc):
    d = os.path.join(ROOT, "tests", "_fftw_double")
    os.makedirs(d, exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", os.path.join(d, "libfftw3.so.3"), os.path.join(ROOT, "tests", "fftw_double.c"), "-lm"])
    got = _child({"LD_LIBRARY_PATH": d + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")})
    assert got["AVAILABLE"] == ["1"] and got["BACKEND"] == ["1"] and got["BACKEND0"] == ["0"]
    assert float(got["SPEC"][0]) < 5e-7                     # float magnitudes of two double-precision transforms
    assert 0.0 <= float(got["TENSION"][0]) <= 1e-4          # north_star's tolerance for the float taps
    assert got["FRAMES"][0] == got["FRAMES"][1]
