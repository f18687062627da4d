"""Delta-debugging of a failing life-cycle fuzz case: replays an op list (from the test's trace) against oracle and HIP
and shrinks it while the final outputs still differ.  usage: lc_ddmin.py CASE.pkl (written by the test under SPX_LC_DUMP=prefix)"""
import sys, os, re, ast, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch  # noqa
from oracle import pyorc as orc
from speedy_amd.sonic2 import SonicStream
import test_gpu_fuzz as T

import pickle
d = pickle.load(open(sys.argv[1], "rb"))
tag, x, ops = d["tag"], d["x"], d["ops"]
_, _, rate, ch, kind, n, speed, nl, fb, mm, small = tag
print("case", tag, "ops", len(ops), flush=True)

def differs(ops):
    L = orc.lib()
    h = L.orc_sonicCreateStream(rate, ch, int(mm))
    s = SonicStream(rate, ch, mm)
    L.orc_sonicSetSpeed(h, speed); s.set_speed(speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl); s.enable_nonlinear(nl)
    L.orc_sonicSetDurationFeedbackStrength(h, fb); s.set_feedback(fb)
    pos = 0
    for op in ops:
        if op == "flush":
            L.orc_sonicFlushStream(h); s.flush()
        elif op[0] == "speed":
            L.orc_sonicSetSpeed(h, op[1]); s.set_speed(op[1])
        elif op[0] == "rate":
            L.orc_sonicSetRate(h, op[1]); s.set_rate(op[1])
        elif op[0] in ("nl", "mode"):
            L.orc_sonicEnableNonlinearSpeedup(h, op[1]); s.enable_nonlinear(op[1])
        elif op[0] == "fb":
            L.orc_sonicSetDurationFeedbackStrength(h, op[1]); s.set_feedback(op[1])
        elif op == "iflush":
            L.orc_sonicIntFlushStream(h); s.int_flush()
        elif op[0] == "ispeed":
            L.orc_sonicIntSetSpeed(h, op[1]); s.int_set_speed(op[1])
        elif op[0] == "iw":
            seg = np.ascontiguousarray(x[pos * ch:(pos + op[1]) * ch]); pos += op[1]
            L.orc_sonicIntWriteShortToStream(h, orc.sptr(seg), op[1]); s.int_write_short(seg)
        elif op[0] == "r":
            buf0 = np.zeros(op[1] * ch, np.int16)
            k0 = L.orc_sonicReadShortFromStream(h, orc.sptr(buf0), op[1]); g0 = s.read_short(op[1])
            if not (g0.size == k0 * ch and np.array_equal(g0, buf0[:k0 * ch])):
                L.orc_sonicDestroyStream(h); s.close()
                return True, k0, g0.size // ch
        elif op[0] == "w":
            seg = np.ascontiguousarray(x[pos * ch:(pos + op[1]) * ch]); pos += op[1]
            L.orc_sonicWriteShortToStream(h, orc.sptr(seg), op[1]); s.write_short(seg)
    L.orc_sonicFlushStream(h); s.flush()
    buf = np.zeros(200000 * ch, np.int16)
    k = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 200000)
    got = s.read_short(200000)
    L.orc_sonicDestroyStream(h); s.close()
    return not (got.size == k * ch and np.array_equal(got, buf[:k * ch])), k, got.size // ch

d, k, g = differs(ops)
print("full replay differs:", d, k, g, flush=True)
if d:
    n_chunks = 2
    while len(ops) >= 2:
        size = max(1, len(ops) // n_chunks)
        reduced = False
        for i in range(0, len(ops), size):
            cand = ops[:i] + ops[i + size:]
            if cand and differs(cand)[0]:
                ops = cand; n_chunks = max(2, n_chunks - 1); reduced = True
                break
        if not reduced:
            if size == 1:
                break
            n_chunks = min(len(ops), n_chunks * 2)
    print("minimal", len(ops), ops, differs(ops), flush=True)
