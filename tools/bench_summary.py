"""Print the figures of a bench.py JSON line one per row (developer convenience)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f %s, %.3f ms per step, n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
r = d["roofline"]
print("hbm frac %.5f  kernels %s" % (r["frac"], {k.split("<")[0][4:]: round(v, 3) for k, v in r["kernel_ms_per_step"].items()}))
if r.get("latency"):
    la = r["latency"]
    print("latency: steps max %d mean %.0f, %.0f cycles/step achieved, floor %.0f, frac %.3f" % (
        la["steps_per_stream"]["max"], la["steps_per_stream"]["mean"], la["achieved_cycles_per_step"],
        la["model_floor_cycles_per_step"], la["frac"]))
if r.get("valu_fp64"):
    v = r["valu_fp64"]
    print("valu_fp64: analysis alone %.3f ms, %.2f TFLOP/s of %.1f, frac %.3f" % (v["standalone_ms"], v["achieved"], v["peak"], v["frac"]))
for k in ("large_batch", "other_rates", "pcie_inclusive", "config4_shard", "config4_full", "api_256_handles"):
    v = d.get(k)
    if v:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("note", "kernels", "hbm")},
              ("hbm frac %.4f" % v["hbm"]["frac"]) if "hbm" in v else "")
c = d.get("cpu_baseline")
if c:
    print("cpu", {a: (round(b, 3) if isinstance(b, float) else b) for a, b in c.items() if a not in ("sample",)})
