"""GPU-box diagnostic for bench.py's cpu_baseline: host topology (CPUs visible, affinity, cgroup quota) and the thread
scaling curve of oracle/orc_bench.c on the bench streams.  Usage: python tools/cpu_scaling.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speedy_amd.synth import speech_like  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'", shell=True, capture_output=True,
                     text=True).stdout)
subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
L.orc_bench_run.restype = C.c_double
L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
n = 160000
base = [speech_like(n, 16000, seed=1000 + i) for i in range(8)]
for threads in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    k = max(8, 8 * threads)
    buf = np.ascontiguousarray(np.concatenate([base[i % 8] for i in range(k)]), np.int16)
    frames = (C.c_long * k)()
    crcs = (C.c_uint32 * k)()
    dt = L.orc_bench_run(buf.ctypes.data, n, k, 16000, 1, 3.5, 1.0, 0.0, 0, 1000, threads, frames, crcs)
    print("threads %4d  streams %5d  %.3f s  %8.1f Msamples/s  %.2f per thread" %
          (threads, k, dt, k * n / dt / 1e6, k * n / dt / 1e6 / threads))
