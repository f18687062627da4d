"""First-contact diagnostics on the GPU box: prints where (if anywhere) the HIP path and the oracle part."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyorc  # noqa: E402
from util import read_wav  # noqa: E402
from speedy_amd.batch import compress_batch  # noqa: E402

pyorc.build()
x, rate, ch = read_wav("tapestry.wav")
for mm, speed, nl, fb in ((False, 3.5, 1.0, 0.0), (True, 3.0, 1.0, 0.1), (False, 2.0, 0.0, 0.0)):
    t0 = time.time()
    outs, b = compress_batch([x], rate, ch, speed, nl, fb, mm, taps=(nl != 0), spectrogram_taps=(nl != 0))
    t1 = time.time()
    ref = pyorc.compress_sound(x, rate, ch, speed, nl, fb, mm)
    print(f"mm={mm} speed={speed} nl={nl} fb={fb}: gpu out {outs[0].size} oracle out {ref['out'].size}  ({t1-t0:.2f}s)")
    if nl != 0:
        taps = b.tap_arrays(0)
        for key in ("tension", "speed", "features"):
            a, r = taps[key], ref[key]
            if a.shape != r.shape:
                print("  ", key, "shape", a.shape, r.shape)
                continue
            d = np.abs(a - r)
            bad = np.argwhere(a != r)
            print(f"   {key}: max|d|={d.max():.3e} mismatching={len(bad)} first={bad[:3].tolist()}")
            if key == "features" and len(bad):
                cols = sorted(set(bad[:, 1].tolist()))
                print("     feature columns differing:", cols)
                for c in cols[:6]:
                    rows = bad[bad[:, 1] == c][:, 0]
                    r0 = rows[0]
                    print(f"     col {c}: first row {r0}: gpu {a[r0, c]!r} oracle {r[r0, c]!r}")
    n = min(outs[0].size, ref["out"].size)
    neq = np.nonzero(outs[0][:n] != ref["out"][:n])[0]
    print(f"   out: equal={np.array_equal(outs[0], ref['out'])} first mismatch={neq[:3].tolist()}")
