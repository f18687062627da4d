"""GPU: the headline kernel's run time is watched.  The walk kernel of the bench batch (BASELINE configs[3]: 256 streams x
10 s, the bench's own streams) is compared with the committed reference (profiles/perf_reference.json, written by
`python tools/perf_reference.py` after a deliberate change) -- in round 2 a change of the link order alone cost 26 %
unnoticed.  More than 10 % above it fails; 5 .. 10 % is reported as a skip (the pool's boxes differ by up to 3 % among
themselves); skipped with a message too when the box is too noisy to tell (two measurements more than 3 % apart)."""
import ctypes as C
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(test, guarded, detail):
    """A skipped guard is green: what each guard actually did goes into gpurun_out/perf_guard.json, and bench.py copies the file
    into its line (config.perf_guard) -- a record can then tell a guarded run from one whose box was too noisy to judge."""
    path = os.path.join(ROOT, "gpurun_out", "perf_guard.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[test] = {"guarded": bool(guarded), "detail": detail}
        json.dump(cur, open(path, "w"), indent=1)
    except Exception:  # noqa: BLE001
        pass


def measure_walk_ms(steps=12, warm=4):
    import torch
    import bench
    from speedy_amd.batch import Batch, Plan
    n = bench.RATE * bench.SECONDS
    plan = Plan(bench.RATE, False)
    b = Batch(plan, [n] * bench.STREAMS_PER_GPU, 1, bench.SPEED, 1.0, 0.0)
    b.upload(bench.make_streams(bench.STREAMS_PER_GPU, n, 0))
    L = plan.L
    out = []
    for _ in range(2):
        for _ in range(warm):
            b.run()
        torch.cuda.synchronize()
        L.spx_set_timing(1)
        for _ in range(steps):
            b.run()
        torch.cuda.synchronize()
        L.spx_set_timing(0)
        sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
        L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
        out.append(sw.value / max(1, nc.value))
    names = L.spx_batch_kernel_names(plan.h, bench.STREAMS_PER_GPU, 1, 1).decode().split(";")
    return out, names[2]


def measure_pipelined(steps=40, warm=12):
    """The headline's loop: the bench batch through the owning pipeline object (resident input, outputs left on the device).
    Two windows; per window (ms per step, average launch of the walk kernel -- its LEAN form, two launches in flight -- in ms)."""
    import time
    import torch
    import bench
    from speedy_amd.batch import Batch, Pipeline, Plan
    n = bench.RATE * bench.SECONDS
    plan = Plan(bench.RATE, False)
    b = Batch(plan, [n] * bench.STREAMS_PER_GPU, 1, bench.SPEED, 1.0, 0.0)
    b.upload(bench.make_streams(bench.STREAMS_PER_GPU, n, 0))
    pipe = Pipeline(plan, [n] * bench.STREAMS_PER_GPU, 1, bench.SPEED, 1.0, 0.0, depth=4, device_out=True)
    L = plan.L
    out = []
    for _ in range(2):
        for _ in range(warm):
            pipe.submit(b.d_in)
        torch.cuda.synchronize()
        L.spx_set_timing(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit(b.d_in)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        L.spx_set_timing(0)
        sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
        L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
        out.append((dt * 1e3, sw.value / max(1, nc.value)))
    form = L.spx_debug_last_walk_form()
    names = (L.spx_batch_kernel_names_lean if form == 64 else L.spx_batch_kernel_names)(plan.h, bench.STREAMS_PER_GPU, 1, 1).decode().split(";")
    pipe.close()
    return out, names[2]


def test_pipelined_step_and_lean_walk_kernel_within_10_percent_of_reference():
    """Round 5: the guard also watches the kernel the HEADLINE runs on -- the lean walk form inside the pipelined loop -- and the
    step time of that loop (profiles/perf_reference.json "pipelined", written by tools/perf_reference.py)."""
    ref = json.load(open(os.path.join(ROOT, "profiles", "perf_reference.json"))).get("pipelined")
    if not ref:
        pytest.skip("profiles/perf_reference.json has no 'pipelined' entry yet (python tools/perf_reference.py on the GPU box)")
    (a, b), kernel = measure_pipelined()
    if abs(a[0] - b[0]) > 0.04 * min(a[0], b[0]):
        _record("pipelined", False, "noisy box: two windows %.3f / %.3f ms per step" % (a[0], b[0]))
        pytest.skip("noisy box: two windows of the pipelined loop %.3f / %.3f ms per step" % (a[0], b[0]))
    assert kernel == ref["kernel"], (kernel, ref["kernel"])
    step, walk = min(a[0], b[0]), min(a[1], b[1])
    msg = "pipelined loop %.3f ms per step (reference %.3f), lean walk kernel %.3f ms per launch (reference %.3f)" % (
        step, ref["ms_per_step"], walk, ref["walk_ms_per_launch"])
    _record("pipelined", True, msg)
    assert step <= 1.10 * ref["ms_per_step"] and walk <= 1.10 * ref["walk_ms_per_launch"], msg
    if step > 1.05 * ref["ms_per_step"]:
        pytest.skip(msg + " -- between 5 and 10 %: check on another box")


def test_walk_kernel_within_5_percent_of_reference():
    ref = json.load(open(os.path.join(ROOT, "profiles", "perf_reference.json")))
    (a, b), kernel = measure_walk_ms()
    if abs(a - b) > 0.03 * min(a, b):
        _record("walk_kernel", False, "noisy box: two measurements %.3f / %.3f ms" % (a, b))
        pytest.skip("noisy box: two measurements of the walk kernel %.3f / %.3f ms" % (a, b))
    got = min(a, b)
    assert kernel == ref["kernel"], (kernel, ref["kernel"])
    excess = got / ref["walk_ms_per_step"] - 1.0
    msg = "walk kernel %.3f ms per step, reference %.3f (profiles/perf_reference.json): %.1f %% above it" % (
        got, ref["walk_ms_per_step"], 100.0 * excess)
    # boxes of this pool differ by up to 3 % among themselves (same library, same streams: 2.28 ... 2.35 ms seen in round 3):
    # 5 .. 10 % above the reference is reported, not failed -- it needs a look on a second box (tools/perf_reference.py);
    # beyond 10 % it is a regression on any box
    _record("walk_kernel", True, msg)
    assert excess <= 0.10, msg
    if excess > 0.05:
        pytest.skip(msg + " -- between 5 and 10 %: check on another box")
    if got < 0.93 * ref["walk_ms_per_step"]:
        print("walk kernel %.3f ms against a reference of %.3f: refresh profiles/perf_reference.json" % (got, ref["walk_ms_per_step"]))


def test_shipped_code_placement_is_within_1_percent_of_its_neighbours():
    """The step loop's speed depends on where it lies relative to the 64-byte instruction fetch lines; the kernel
    ships with SPX_WALK_PAD s_nop's in its prologue that fix the offset.  __graft_entry__.build() also builds the library with
    the offset 8 bytes lower and 8 bytes higher (speedy_amd/lib/ab/libspeedy_hip_pad{lo,hi}.so): each of the three is timed in
    a process of its own, two rounds interleaved, and the shipped offset must be within 1 % of the best -- the sweep that used
    to be "re-run by hand after any change to the kernel" runs by itself.  Skipped where the variants were not built or the
    box is too noisy to tell."""
    import subprocess
    import sys
    libs = {"shipped": None}
    for name in ("padlo", "padhi"):
        path = os.path.join(ROOT, "speedy_amd", "lib", "ab", "libspeedy_hip_%s.so" % name)
        if not os.path.exists(path):
            pytest.skip("speedy_amd/lib/ab/libspeedy_hip_%s.so not built (python -c 'import __graft_entry__ as g; g.build()')" % name)
        libs[name] = path
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from test_gpu_perf_guard import measure_walk_ms; "
            "(a, b), k = measure_walk_ms(); print('WALK', min(a, b), abs(a - b) / min(a, b))" % (ROOT, os.path.join(ROOT, "tests")))
    got = {k: [] for k in libs}
    noisy = []
    for _ in range(2):
        for name, path in libs.items():
            env = dict(os.environ)
            env.pop("SPEEDY_HIP_LIB", None)
            if path:
                env["SPEEDY_HIP_LIB"] = path
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-1500:]
            ms, spread = [float(v) for v in [ln for ln in r.stdout.splitlines() if ln.startswith("WALK")][-1].split()[1:]]
            if spread > 0.01:   # (judged at the end: a verdict that does not depend on it stands)
                noisy.append("%s measured twice %.1f %% apart" % (name, 100 * spread))
            got[name].append(ms)
    best = {k: min(v) for k, v in got.items()}
    # (a verdict that holds even with the shipped library's SLOWER round against every variant's faster one does not depend on the
    # rounds agreeing: the last collection of round 6 skipped with shipped 1.4943 / 1.4949, padlo 1.4956 / 1.5115, padhi 1.5096 / 1.5066)
    if max(got["shipped"]) <= 1.01 * min(best.values()):
        _record("code_placement", True, "%r (both rounds of the shipped library within 1 %% of the best)" % got)
        return
    if noisy:
        _record("code_placement", False, "noisy box: " + "; ".join(noisy))
        pytest.skip("noisy box: " + "; ".join(noisy))
    if max(abs(a - b) / min(a, b) for a, b in got.values()) > 0.007:
        _record("code_placement", False, "noisy box: rounds disagree %r" % got)
        pytest.skip("noisy box: rounds disagree %r" % got)
    _record("code_placement", True, "%r" % best)
    assert best["shipped"] <= 1.01 * min(best.values()), "SPX_WALK_PAD is no longer the best offset: %r -- sweep -DSPX_WALK_PAD=n variants (tools/build_variant.sh, tools/variant_times.sh)" % best
