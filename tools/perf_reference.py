"""Write profiles/perf_reference.json: the walk kernel's time per step on the bench batch, as measured now on this box
(tests/test_gpu_perf_guard.py compares later builds with it).  Run on the GPU box after a deliberate change of the kernel;
gpurun brings the file back under gpurun_out/ -- copy it to profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_perf_guard import measure_pipelined, measure_walk_ms  # noqa: E402

(a, b), kernel = measure_walk_ms()
(pa, pb), pkernel = measure_pipelined()
ref = {"kernel": kernel, "walk_ms_per_step": round(min(a, b), 4), "measurements_ms": [round(a, 4), round(b, 4)],
       "workload": "bench.py batch: 256 streams x 10 s, 16 kHz mono, 3.5x nonlinear, seeds 1234 + stream index",
       "how": "HIP events around the walk kernel inside spx_batch_run (spx_set_timing), 12 steps after 4 warm-up steps, the "
              "smaller of two passes; python tools/perf_reference.py",
       # the headline's loop: the same batch through the owning pipeline object (resident input, four buffer sets), its walk kernel in
       # the lean form with two launches in flight
       "pipelined": {"kernel": pkernel, "ms_per_step": round(min(pa[0], pb[0]), 4), "walk_ms_per_launch": round(min(pa[1], pb[1]), 4),
                     "windows": [[round(v, 4) for v in pa], [round(v, 4) for v in pb]],
                     "how": "spx_pipeline_submit x 40 after 12 warm-up submits, wall clock per step and HIP events around the walk kernel "
                            "(spx_set_timing); the smaller of two windows"}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(ref, open(os.path.join(ROOT, "gpurun_out", "perf_reference.json"), "w"), indent=1)
print(json.dumps(ref))
