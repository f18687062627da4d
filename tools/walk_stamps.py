"""Diagnostic only: per-phase shader-cycle shares of the walk kernel (workgroup 0 = stream 0), from the
`make -C speedy_amd/csrc stamps` builds -- one library per stamped region so that the single accumulator does not
disturb the kernel's register allocation.  Usage on the GPU box:  python tools/walk_stamps.py   (the children call
spx_set_concurrent(0) so that the walk kernel is measured alone)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "event loop (between process calls)", 1: "pre-step (copy steps, loop ctl)", 2: "ensure_window",
         3: "phase B build signals", 4: "sync c", 5: "coarse accumulate", 6: "sync e", 7: "coarse select",
         8: "refine accumulate", 9: "sync h", 10: "refine select", 11: "decision", 12: "after OLA -> end of process",
         13: "overlap-add", 14: "frame-rate passes (prologue)"}


KIND = os.environ.get("STAMP_KIND", "fast")   # "fast": spx_walk_fast.hip (make fstamps); "old": spx_walk.hip (make stamps)
if KIND == "fast":
    NAMES = {0: "event loop (between events)", 1: "step entry / loop control", 2: "window check + refill",
             3: "publish + coarse accumulate", 4: "barrier A wait", 5: "coarse sums read + select",
             6: "refine: candidate n / rem divisions", 7: "barrier B wait", 8: "refine sums read + select",
             9: "previous-period rule", 10: "n, state update", 11: "after the steps of an event",
             12: "refine: setup + ragged loads issued", 13: "refine: common share summed + added",
             14: "refine: ragged tasks summed + added", 15: "window refill (whole)", 16: "window check before a refill"}
FN = "spx_debug_fstamps" if KIND == "fast" else "spx_debug_stamps"
PREFIX = "libspeedy_hip_fstamps_%d.so" if KIND == "fast" else "libspeedy_hip_stamps_%d.so"
NSEL = 17 if KIND == "fast" else 15


def child(sel):
    sys.path.insert(0, ROOT)
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate, n, nstreams = 16000, 160000, int(os.environ.get("NSTREAMS", "256"))
    plan = Plan(rate, False)
    plan.L.spx_set_concurrent(0)
    base = [speech_like(n, rate, seed=i, channels=int(os.environ.get("CHANNELS", "1"))) for i in range(8)]
    ch = int(os.environ.get("CHANNELS", "1"))
    b = Batch(plan, [n] * nstreams, ch, 3.5, 1.0, 0.0)
    b.upload([base[i % 8] for i in range(nstreams)])
    b.run()
    torch.cuda.synchronize()
    L = plan.L
    dbg = getattr(L, FN)
    dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 32)()
    dbg(buf, 1)
    b.run()
    torch.cuda.synchronize()
    dbg(buf, 1)
    print("STAMP %d %d %d %d" % (sel, buf[sel], buf[30], buf[31]))


if len(sys.argv) > 1:
    child(int(sys.argv[1]))
    sys.exit(0)

rows = []
for sel in [int(v) for v in os.environ["STAMP_SELS"].split()] if os.environ.get("STAMP_SELS") else range(NSEL):
    env = dict(os.environ,
               SPEEDY_HIP_LIB=os.path.join(ROOT, "speedy_amd", "lib", "stamps", PREFIX % sel))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(sel)], env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("STAMP"):
            rows.append([int(v) for v in line.split()[1:]])
    if out.returncode:
        print("region", sel, "failed:", out.stderr[-400:])
if rows:
    steps = rows[0][3]
    total = sum(r[1] for r in rows)
    print("stream 0: %d pitch steps, kernel %d cycles (%.0f per step); stamped regions sum to %d" %
          (steps, rows[0][2], rows[0][2] / max(1, steps), total))
    for sel, cyc, kern, _ in rows:
        print("  %2d %-40s %10d  %5.1f %%  %7.0f cycles/step" % (sel, NAMES.get(sel, ""), cyc, 100.0 * cyc / max(1, total),
                                                                  cyc / max(1, steps)))
