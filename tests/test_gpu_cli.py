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


def _run(args, timeout=180):
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "cli"])
    r = subprocess.run([CLI, "--input", os.path.join(GOLDEN, "tapestry.wav")] + args, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_cli_match_nonlinear_two_pass(orc, tmp_path):
    """speedy_wave.cc:424-427 with the flags of its header example (:62): probe pass nonlinear without output,
    final pass linear at the achieved rate."""
    out = str(tmp_path / "matched.wav")
    _run(["--output", out, "--nonlinear", "0.0", "--speed", "3", "--match_nonlinear"])
    x, rate, ch = read_wav("tapestry.wav")
    probe = orc.compress_sound(x, rate, ch, 3.0, 1.0, 0.0, False, chunk=1000)
    achieved = float(x.size // ch) / float(probe["out"].size // ch)
    ref = orc.compress_sound(x, rate, ch, achieved, 0.0, 0.0, False, chunk=1000)
    assert np.array_equal(_read_out(out), ref["out"])


def test_cli_length_two_pass(orc, tmp_path):
    """speedy_wave.cc:428-461: --length rescales the request by (wanted / achieved) of a probing pass."""
    out = str(tmp_path / "len.wav")
    _run(["--output", out, "--length", "1.5"])
    x, rate, ch = read_wav("tapestry.wav")
    total = x.size // ch
    want = float(np.float32(total) / np.float32(rate)) / 1.5
    probe = orc.compress_sound(x, rate, ch, want, 1.0, 0.0, False, chunk=1000)
    got_speed = float(total) / float(probe["out"].size // ch)
    ref = orc.compress_sound(x, rate, ch, want * (want / got_speed), 1.0, 0.0, False, chunk=1000)
    got = _read_out(out)
    assert np.array_equal(got, ref["out"])
    assert abs(got.size / rate - 1.5) < 0.15    # what the flag is for


def test_cli_feature_and_spectrogram_files(orc, tmp_path):
    """--features_file / --spectrogram_file / --normalized_spectrogram_file (speedy_wave.cc:86-124): one text
    row per callback, compared with the oracle shim's callbacks fed the same 1000-frame writes."""
    out = str(tmp_path / "o.wav")
    ff, gf, nf = (str(tmp_path / n) for n in ("f.txt", "g.txt", "n.txt"))
    _run(["--output", out, "--features_file", ff, "--spectrogram_file", gf, "--normalized_spectrogram_file", nf])
    x, rate, ch = read_wav("tapestry.wav")
    L = orc.lib()
    rows = {"f": [], "g": [], "n": []}
    h = L.orc_sonicCreateStream(rate, ch, 0)
    n = L.orc_sonicSpectrogramSize(h)
    cbs = [orc.FEATURES_FN(lambda s, t, p: rows["f"].append(np.ctypeslib.as_array(p, shape=(15,)).copy())),
           orc.FEATURES_FN(lambda s, t, p: rows["g"].append(np.ctypeslib.as_array(p, shape=(n,)).copy())),
           orc.FEATURES_FN(lambda s, t, p: rows["n"].append(np.ctypeslib.as_array(p, shape=(n,)).copy()))]
    L.orc_sonicFeaturesCallback(h, cbs[0])
    L.orc_sonicSpectrogramCallback(h, cbs[1])
    L.orc_sonicNormalizedSpectrogramCallback(h, cbs[2])
    L.orc_sonicSetSpeed(h, 3.5)
    L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
    for pos in range(0, x.size, 1000):
        seg = np.ascontiguousarray(x[pos:pos + 1000])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size)
    L.orc_sonicDestroyStream(h)
    for path, key in ((ff, "f"), (gf, "g"), (nf, "n")):
        got, ref = np.loadtxt(path), np.array(rows[key], dtype=np.float64)
        assert got.shape == ref.shape, key
        bad = ~np.isclose(got, ref, rtol=1e-5, atol=1e-6, equal_nan=True)   # %g keeps 6 significant digits
        assert not bad.any(), (key, int(bad.sum()), got[bad][:4], ref[bad][:4])


EXAMPLE = os.path.join(ROOT, "speedy_amd", "lib", "batch_example")


@pytest.mark.parametrize("name,speed,nl,copies,split", [("tapestry.wav", 3.5, 1.0, 5, 0), ("tapestry.wav", 3.5, 1.0, 3, 1),
                                                        ("tapestry22050.wav", 0.8, 0.0, 2, 0), ("tapestry.wav", 2.0, 1.0, 300, 1)])
def test_c_batch_example(orc, tmp_path, name, speed, nl, copies, split):
    """INTEGRATION.md section 2 as a plain C99 program (tools/batch_example.c: gcc -std=c99 -pedantic -Werror over
    include/speedy_hip.h, no HIP headers): device memory through spx_device_alloc / spx_copy_*, one spx_batch_run or
    spx_batch_analyze + spx_batch_walk, the device-side gather -- output equal to the oracle's."""
    if not os.path.exists(EXAMPLE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "example"])
    x, rate, ch = read_wav(name)
    x = x[: 3 * rate * ch]
    raw, out = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    x.astype("<i2").tofile(raw)
    r = subprocess.run([EXAMPLE, raw, str(rate), str(ch), str(speed), str(nl), str(copies), str(split), out],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    ref = orc.compress_sound(x, rate, ch, speed, nl, 0.0, False, chunk=1000 if nl else x.size // ch, taps=False)["out"]
    assert np.array_equal(np.fromfile(out, dtype="<i2"), ref)


@pytest.mark.parametrize("name,speed,nl,copies,batches,depth", [("tapestry.wav", 3.5, 1.0, 64, 9, 4), ("tapestry22050.wav", 1.5, 1.0, 7, 5, 2),
                                                                ("tapestry.wav", 2.0, 0.0, 300, 4, 3)])
def test_c_pipeline_example(orc, tmp_path, name, speed, nl, copies, batches, depth):
    """INTEGRATION.md section 2 "batch after batch" as a plain C99 program (tools/pipeline_example.c over include/speedy_hip.h, no
    HIP headers): batches of COPIES streams through one spx_pipeline, host memory to host memory, inputs produced in the
    pipeline's staging buffer and in a pinned buffer of the caller's by turns -- every stream of every batch equal to the oracle's."""
    exe = os.path.join(ROOT, "speedy_amd", "lib", "pipeline_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "pipeexample"])
    x, rate, ch = read_wav(name)
    x = x[: 3 * rate * ch]
    raw, out = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    x.astype("<i2").tofile(raw)
    r = subprocess.run([exe, raw, str(rate), str(ch), str(speed), str(nl), str(copies), str(batches), str(depth), out],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr + r.stdout
    ref = orc.compress_sound(x, rate, ch, speed, nl, 0.0, False, chunk=1000 if nl else x.size // ch, taps=False)["out"]
    assert np.array_equal(np.fromfile(out, dtype="<i2"), ref)


@pytest.mark.parametrize("streams,batches,depth", [(96, 9, 4), (256, 7, 3), (300, 3, 2)])
def test_c_mixed_pipeline_example(orc, tmp_path, streams, batches, depth):
    """tools/mixed_pipeline_example.c (plain C99 over include/speedy_hip.h): one GPU's kind of BASELINE configs[4] shard through
    spx_pipeline_create_mixed with SPX_PIPELINE_DEVICE_OUT -- input and outputs resident on the device, the walk kernels of
    consecutive batches overlapping (round 6) -- every stream of every batch equal to ONE spx_batch_run_mixed call's (the program
    checks that itself); here four of its streams, one per kind, against the oracle."""
    exe = os.path.join(ROOT, "speedy_amd", "lib", "mixed_pipeline_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "pipeexample"])
    x, rate, ch = read_wav("tapestry.wav")
    x = x[: 2 * rate]
    raw = str(tmp_path / "in.raw")
    x.astype("<i2").tofile(raw)
    r = subprocess.run([exe, raw, str(streams), str(batches), str(depth)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "every stream equals spx_batch_run_mixed's" in r.stdout
    # the plain mixed call the program compares with is the oracle's: the same streams through the Python mirror
    from speedy_amd.batch import MixedBatch, Plan
    plans = [Plan(16000, False), Plan(22050, False)]
    n = 8
    pidx = [i % 2 for i in range(n)]
    chs = [1 if (i // 2) % 2 == 0 else 2 for i in range(n)]
    speeds = [1.5 if (i // 4) % 2 == 0 else 3.5 for i in range(n)]
    mb = MixedBatch(plans, pidx, [x.size // c for c in chs], chs, speeds, 1.0, 0.0)
    mb.upload([x] * n)
    mb.run()
    outs = mb.results()
    for i in range(n):
        ref = orc.compress_sound(x, [16000, 22050][pidx[i]], chs[i], speeds[i], 1.0, 0.0, False, chunk=1000, taps=False)["out"]
        assert np.array_equal(outs[i], ref), i


@pytest.mark.parametrize("seed,handles,env", [(1, 24, {}), (2, 40, {}), (3, 16, {}),
                                              # a frame arena that must grow several times, a staging area that forces runs
                                              (4, 48, {"SPX_POOL_FRAMES": "1024"}),
                                              (5, 32, {"SPX_POOL_FRAMES": "1024", "SPX_POOL_STAGE_BYTES": "30000"})])
def test_c_api_fuzz_two_execution_paths_agree(seed, handles, env):
    """tools/api_fuzz.c, a plain C99 program over include/sonic2.h: a random schedule of writes (short and float), reads,
    flushes, setters, callbacks switched on mid-stream, sonicInt* calls and destroy / re-create over many handles, run once
    with coalesced execution and once with every handle on its own launch sequences -- every handle delivers the same bytes
    at every read.  (The same program runs under ASan + UBSan: tools/asan_host.sh.)"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "apifuzz"])
    r = subprocess.run([os.path.join(ROOT, "speedy_amd", "lib", "api_fuzz"), str(seed), str(handles), "2500"], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "0 handles differ" in r.stdout


@pytest.mark.parametrize("speed,nonlinear,match", [(3.5, 1.0, False), (2.0, 0.0, False), (3.0, 0.0, True)])
def test_reference_cli_binary_runs_on_this_library(orc, tmp_path, speed, nonlinear, match):
    """tests/_refcli/speedy_wave_ref is the REFERENCE's own speedy_wave.cc -- compiled where it lies in the build container,
    unmodified, against include/compat + its own headers (tests/Makefile `refcli`) -- linked with libspeedy_hip.so.  Run on
    tapestry.wav as the reference's header comment runs it (speedy_wave.cc:50-66); the WAV it writes must hold exactly the
    oracle's samples, and the tension and speed files its callbacks write the oracle's values.  --match_nonlinear is the
    two-pass use (speedy_wave.cc:424-427): a nonlinear pass without callbacks or output (this library's coalesced path),
    whose achieved speed-up then drives a linear pass.  Skipped where the binary was not built."""
    exe = os.path.join(ROOT, "tests", "_refcli", "speedy_wave_ref")
    if not os.path.exists(exe):
        pytest.skip("tests/_refcli/speedy_wave_ref not built (needs /root/reference at build time)")
    from util import read_wav
    x, rate, ch = read_wav("tapestry.wav")
    out = tmp_path / "out.wav"
    ten, spd = tmp_path / "tension.txt", tmp_path / "speed.txt"
    cmd = [exe, "--input", os.path.join(ROOT, "tests", "golden", "tapestry.wav"), "--output", str(out), "--speed", str(speed),
           "--nonlinear", str(nonlinear), "--tension_file", str(ten), "--speed_file", str(spd)]
    if match:
        cmd.append("--match_nonlinear")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    if match:
        first = orc.compress_sound(x, rate, ch, speed, 1.0, 0.0, False, chunk=1000, taps=False)
        assert "output %d frames with nonlinear=1." % first["out"].size in r.stdout, r.stdout[-600:]
        speed = float(x.size) / first["out"].size        # speedy_wave.cc:240-241, in double
    ref = orc.compress_sound(x, rate, ch, speed, nonlinear, 0.0, False, chunk=1000, taps=(nonlinear != 0))   # the CLI's defaults
    raw = open(out, "rb").read()
    got = np.frombuffer(raw[44:], np.int16)
    assert got.size == ref["out"].size and np.array_equal(got, ref["out"]), (got.size, ref["out"].size)
    assert "Compress_sound read %d frames, and output %d frames" % (x.size, ref["out"].size) in r.stdout
    if nonlinear:
        t = np.array([float(v) for v in open(ten).read().split()], np.float32)
        s_ = np.array([float(v) for v in open(spd).read().split()], np.float32)
        assert t.size == ref["tension"].size and t.size > 250
        assert np.allclose(t, ref["tension"], rtol=1e-5, atol=1e-6)         # "%g" keeps 6 significant digits
        assert np.allclose(s_, ref["speed"], rtol=1e-5, atol=1e-6)
