import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch  # noqa
from oracle import pyorc as orc
import test_gpu_sonic2 as T
from speedy_amd.synth import speech_like

def run(tag, rate_hz, ch, speed, nl, chunk, plan, mm=False, writes=36, seed=71):
    x = speech_like(writes * chunk, rate_hz, seed=seed, channels=ch)
    ro, rg, co, cg = T._rate_streams(orc, x, rate_hz, ch, speed, nl, mm, chunk, plan)
    m = min(ro.size, rg.size)
    d = np.nonzero(ro[:m] != rg[:m])[0]
    cum = np.cumsum(co)
    w = int(np.searchsorted(cum, d[0] // ch, side="right")) if d.size else -1
    print(tag, "sizes", ro.size // ch, rg.size // ch, "counts eq", co == cg, "first mismatch frame",
          (d[0] // ch if d.size else None), "in write", w, "nmis", d.size)
    if d.size: print("   cum", list(cum))

run("no rate, flush17", 22050, 2, 2.0, 1.0, 160, {'flush_at': 17})
run("no rate, flush17 mono", 22050, 1, 2.0, 1.0, 160, {'flush_at': 17})
run("no rate, flush17 mono 3.5", 22050, 1, 3.5, 1.0, 160, {'flush_at': 17})
run("rate, flush17 mono 2.0", 22050, 1, 2.0, 1.0, 160, {0: 1.5, 'flush_at': 17})
run("rate, flush17 mono 2.0 chunk 200", 22050, 1, 2.0, 1.0, 200, {0: 1.5, 'flush_at': 17})
run("rate, flush17 mono 2.0 chunk 230", 22050, 1, 2.0, 1.0, 230, {0: 1.5, 'flush_at': 17})
run("16k rate, flush17 mono 2.0 chunk 100", 16000, 1, 2.0, 1.0, 100, {0: 1.5, 'flush_at': 17})
run("16k rate, flush30 mono 2.0 chunk 100", 16000, 1, 2.0, 1.0, 100, {0: 1.5, 'flush_at': 30})
run("16k no rate, flush17 mono 2.0 chunk 100", 16000, 1, 2.0, 1.0, 100, {'flush_at': 17})
run("lin rate, flush17 mono 2.0", 22050, 1, 2.0, 0.0, 160, {0: 1.5, 'flush_at': 17})
