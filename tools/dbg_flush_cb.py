import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch  # noqa
from oracle import pyorc as orc
from speedy_amd.synth import speech_like
from speedy_amd.sonic2 import SonicStream

def run(rate_hz, ch, speed, nl, chunk, flush_at, writes=36, seed=71, cb=True):
    x = speech_like(writes * chunk, rate_hz, seed=seed, channels=ch)
    L = orc.lib()
    ref = {"tension": [], "speed": []}
    got = {"tension": [], "speed": []}
    h = L.orc_sonicCreateStream(rate_hz, ch, 0)
    cbs = [orc.TENSION_FN(lambda s, t, v: ref["tension"].append((t, v))),
           orc.TENSION_FN(lambda s, t, v: ref["speed"].append((t, v)))]
    L.orc_sonicTensionCallback(h, cbs[0]); L.orc_sonicSpeedCallback(h, cbs[1])
    s = SonicStream(rate_hz, ch, False)
    if cb:
        s.on_tension(lambda t, v: got["tension"].append((t, v)))
        s.on_speed(lambda t, v: got["speed"].append((t, v)))
    L.orc_sonicSetSpeed(h, speed); s.set_speed(speed)
    L.orc_sonicEnableNonlinearSpeedup(h, nl); s.enable_nonlinear(nl)
    L.orc_sonicSetDurationFeedbackStrength(h, 0.0); s.set_feedback(0.0)
    buf = np.zeros(8 * chunk * ch, np.int16)
    ro, rg = [], []
    n = x.size // ch
    for w, pos in enumerate(range(0, n, chunk)):
        if flush_at == w:
            L.orc_sonicFlushStream(h); s.flush()
        seg = np.ascontiguousarray(x[pos * ch:(pos + chunk) * ch])
        L.orc_sonicWriteShortToStream(h, orc.sptr(seg), chunk)
        s.write_short(seg)
        g = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 8 * chunk); ro.append(buf[:g * ch].copy())
        rg.append(s.read_short(8 * chunk))
        if ro[-1].size != rg[-1].size or not np.array_equal(ro[-1], rg[-1]):
            print("  write", w, "read differs", ro[-1].size, rg[-1].size)
    L.orc_sonicFlushStream(h); s.flush()
    g = L.orc_sonicReadShortFromStream(h, orc.sptr(buf), 8 * chunk); a = buf[:g * ch].copy()
    b = s.read_short(8 * chunk)
    print("final drain", a.size, b.size, "equal" if np.array_equal(a, b) else "DIFF")
    if cb:
        print(" tension times ref", [t for t, _ in ref["tension"]])
        print(" tension times got", [t for t, _ in got["tension"]])
        for key in ("tension", "speed"):
            for (t0, v0), (t1, v1) in zip(ref[key], got[key]):
                if t0 != t1 or np.float32(v0) != np.float32(v1):
                    print("  first", key, "diff", t0, v0, t1, v1); break
    else:
        print(" ref speeds", [(t, round(v, 3)) for t, v in ref["speed"]])
    L.orc_sonicDestroyStream(h); s.close()

print("== 22050 chunk 160 flush17"); run(22050, 1, 2.0, 1.0, 160, 17)
print("== same without callbacks"); run(22050, 1, 2.0, 1.0, 160, 17, cb=False)
print("== 22050 chunk 160 flush 30"); run(22050, 1, 2.0, 1.0, 160, 30)
print("== 22050 chunk 160 flush 10"); run(22050, 1, 2.0, 1.0, 160, 10)
