"""Latency of the drop-in streaming API (sonic2.h): one stream, 1000-frame writes each followed by a read, as the
reference CLI does (speedy_wave.cc:199-231)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from speedy_amd.sonic2 import SonicStream  # noqa: E402
from speedy_amd.synth import speech_like  # noqa: E402

rate = 16000
x = speech_like(60 * rate, rate, seed=1)
for nl, speed in ((1.0, 3.5), (0.0, 2.0)):
    for chunk in (1000, 16000):
        s = SonicStream(rate, 1, False)
        s.set_speed(speed)
        s.enable_nonlinear(nl)
        t0 = time.perf_counter()
        n_out = 0
        calls = 0
        for pos in range(0, x.size, chunk):
            s.write_short(x[pos:pos + chunk])
            n_out += s.read_short(chunk).size
            calls += 1
        s.flush()
        while True:
            got = s.read_short(4096).size
            if not got:
                break
            n_out += got
        dt = time.perf_counter() - t0
        s.close()
        print("nonlinear=%.0f speed=%.1f chunk=%5d: %6.1f us per write+read, %.0fx real time, %d frames out" %
              (nl, speed, chunk, dt / calls * 1e6, 60.0 / dt, n_out))
