"""Latency and many-handle throughput of the drop-in streaming API (sonic2.h).

Part 1 (Python, one stream): 1000-frame writes each followed by a read, as the reference CLI does
(speedy_wave.cc:199-231) -- the per-call latency of the API.
Part 2 (--streams N,N,...; C, one host thread, tools/stream_bench.c): N live handles, every round writes 1000 frames to
each handle and then reads from each.  With coalescing (the default) a round is ONE launch sequence for all handles;
`percall` (read right after each write) and SPX_NO_POOL=1 (every handle its own launch sequence) are printed beside it."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_stream():
    import numpy as np  # noqa: F401
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate = 16000
    x = speech_like(60 * rate, rate, seed=1)
    for coalesce in (True, False):
        for nl, speed in ((1.0, 3.5), (0.0, 2.0)):
            for chunk in (1000, 16000):
                s = SonicStream(rate, 1, False, coalesce)
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
                print("one stream, %s: nonlinear=%.0f speed=%.1f chunk=%5d: %6.1f us per write+read, %.0fx real time, %d frames out" %
                      ("coalesced path" if coalesce else "eager path    ", nl, speed, chunk, dt / calls * 1e6, 60.0 / dt, n_out),
                      flush=True)


def many(streams, seconds):
    exe = os.path.join(ROOT, "speedy_amd", "lib", "stream_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "streambench"])
    for n in streams:
        for label, order, env in (("coalesced, write all / read all", "rounds", {}),
                                  ("coalesced, read after each write", "percall", {}),
                                  ("SPX_NO_POOL=1, write all / read all", "rounds", {"SPX_NO_POOL": "1"})):
            secs = seconds if order == "rounds" and not env else min(seconds, max(1.0, 64.0 / n))
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([exe, str(n), str(secs), "1000", "3.5", "1", order], env=e, capture_output=True, text=True)
            if out.returncode != 0:
                print("streams=%d %s: FAILED %s" % (n, label, out.stderr.strip()), flush=True)
                continue
            r = json.loads(out.stdout.strip().splitlines()[-1])
            print("streams=%4d  %-36s %9.1f Msamples/s  %8.1fx real time per stream  %8.1f us per round  (%.1f handles per "
                  "launch sequence)" % (n, label, r["msamples_per_s"], r["x_realtime_per_stream"], r["us_per_round"],
                                        r["handles_per_sequence"]), flush=True)
            print("   " + json.dumps(r), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="", help="comma-separated handle counts for the many-handle part, e.g. 1,16,64,256")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per handle")
    ap.add_argument("--skip-one", action="store_true")
    a = ap.parse_args()
    if not a.skip_one:
        one_stream()
    if a.streams:
        many([int(v) for v in a.streams.split(",")], a.seconds)
