"""bench.py's pcie_pipeline several times in one process: is the figure stable?  SPX_DEBUG_MODE=1 prints the mode per call."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from speedy_amd.batch import Plan
from speedy_amd.synth import speech_like
rate, n, ns = 16000, 160000, 256
plan = Plan(rate, False)
streams = [speech_like(n, rate, seed=1234 + i) for i in range(ns)]
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    dt, _ = bench.pcie_pipeline(plan, streams, n, reps=20)
    print("pass %d: %.3f ms per batch" % (r, dt * 1e3), flush=True)
