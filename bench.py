"""bench.py — the headline metric of BASELINE.json on MI355X.

Workload (config.workload): BASELINE.json configs[3] per GPU — 256 concurrent 16 kHz mono streams x 10 s of
synthetic speech-like int16, speed 3.5, nonlinear on, duration feedback 0 (speedy_wave.cc:33 default) —
the configuration the metric "Msamples/s processed (16 kHz mono, 3.5x nonlinear)" is quoted on.  One "step" =
one pass of the whole hot path (analysis kernel + walk kernel) over the batch, inputs already resident in HBM.
N > 1: one process per GPU, each with its own 256 streams (weak scaling, streams share nothing; the only
collective is the barrier / MAX-reduce of the timing, over RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RATE, SECONDS, STREAMS_PER_GPU, SPEED = 16000, 10, 256, 3.5
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_streams(n_streams, n, rank):
    """Distinct speech-like streams: 32 generated bases, the rest are rotations by a stream-specific offset."""
    from speedy_amd.synth import speech_like
    bases = [speech_like(n, RATE, seed=1000 * rank + i) for i in range(min(32, n_streams))]
    out = []
    for i in range(n_streams):
        b = bases[i % len(bases)]
        out.append(np.roll(b, (i // len(bases)) * 7919) if i >= len(bases) else b)
    return out


def cpu_baseline(streams, budget_s=12.0):
    """The CPU oracle (kind "port": C restatement of the reference path, the reference itself is unbuildable
    here) over a bounded sample of the SAME streams, one stream per task over all host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyorc
    pyorc.build()
    cores = os.cpu_count() or 1
    # probe one stream to size the sample
    t0 = time.perf_counter()
    pyorc.compress_sound(streams[0], RATE, 1, SPEED, 1.0, 0.0, False, taps=False)
    one = time.perf_counter() - t0
    n = int(max(cores, min(len(streams), budget_s * cores / max(one, 1e-4))))
    sample = streams[:n]

    def run(x):
        return pyorc.compress_sound(x, RATE, 1, SPEED, 1.0, 0.0, False, taps=False)["out"].size

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(run, sample))
    dt = time.perf_counter() - t0
    total = sum(x.size for x in sample)
    return {"value": total / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d bench streams (%d s each), one stream per task on %d threads; "
                      "single-thread rate %.2f Msamples/s" % (n, len(streams), SECONDS, cores,
                                                              streams[0].size / one / 1e6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("SPX_CHUNKS", "1")),
                    help="time chunks per stream inside one spx_batch_run (analysis of chunk c+1 overlaps the walk of c)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for "
                    "a functional check of the N > 1 path on a single GPU)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (the product has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dist = None
    red_dev = "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
            red_dev = "cpu"
        # work-partition handshake: every rank announces its shard (stream count, input frames)
        from speedy_amd.dist import handshake
        layout = handshake(dist, STREAMS_PER_GPU, STREAMS_PER_GPU * RATE * SECONDS, device=red_dev)
        assert layout.shape == (world, 2)

    from speedy_amd.batch import Batch, Plan
    n = RATE * SECONDS
    streams = make_streams(STREAMS_PER_GPU, n, rank)
    plan = Plan(RATE, False)
    b = Batch(plan, [n] * STREAMS_PER_GPU, 1, SPEED, 1.0, 0.0)
    b.upload(streams)
    L = plan.L
    L.spx_set_pipeline_chunks(args.chunks)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        b.run()
    barrier()
    L.spx_set_timing(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.run()
    barrier()
    dt = time.perf_counter() - t0
    L.spx_set_timing(0)
    sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
    L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
    ms_tension = float(L.spx_timing_last_tension_ms()) / max(1, nc.value)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_in_local = n * STREAMS_PER_GPU
    outs = b.results()
    n_out = int(sum(o.size for o in outs))

    # PCIe-inclusive rate (reported beside, never as `value`): pinned host input -> HBM, the same step, produced
    # output -> pinned host memory.
    pcie = None
    if rank == 0:
        h_in = torch.empty(b.d_in.numel(), dtype=torch.int16).pin_memory()
        h_in.copy_(b.d_in.cpu())
        h_out = torch.zeros(b.d_out.numel(), dtype=torch.int16).pin_memory()  # zeros: every page touched before timing
        reps = 5
        h_offs = torch.empty(b.n + 1, dtype=torch.int64).pin_memory()
        t1 = 0.0
        for rep in range(reps + 2):
            if rep == 2:  # two warm-up reps: first touch of the pinned buffers, allocator pools, host page cache
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            b.d_in.copy_(h_in, non_blocking=True)
            b.run()
            d_packed, d_offs = b.pack_outputs()          # gather on the device: one copy instead of one per stream
            h_offs.copy_(d_offs, non_blocking=True)
            torch.cuda.synchronize()
            total = int(h_offs[-1])
            h_out[:total].copy_(d_packed[:total], non_blocking=True)
            torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / reps
        pcie = {"value": n_in_local / dt1 / 1e6, "unit": "Msamples/s", "ms_per_step": dt1 * 1e3,
                "note": "rank 0 only: H2D of the int16 input + the step + device-side gather + one D2H of the produced "
                        "int16 output, pinned host buffers; not the headline value"}
    n_in = n * STREAMS_PER_GPU

    if rank == 0:
        total_in = n_in * world * args.steps
        ms_step = dt / args.steps * 1e3
        ms_analyze = sa.value / max(1, nc.value)
        ms_walk = sw.value / max(1, nc.value)
        algo_bytes = 2 * 1 * (n_in + n_out)  # int16 read once + int16 written once, per launch
        dom, dom_ms = ("spx_walk_kernel", ms_walk) if ms_walk >= ms_analyze else ("spx_analysis_kernel", ms_analyze)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s processed (16 kHz mono, 3.5x nonlinear)",
            "value": total_in / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 samples; f64 DFT, f32 features, int32 AMDF/OLA",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d streams/GPU x %d s, 16 kHz mono int16, speed 3.5, "
                                   "nonlinear 1.0, feedback 0" % (STREAMS_PER_GPU, SECONDS),
                       "streams_per_gpu": STREAMS_PER_GPU, "samples_per_stream": n,
                       "parallelism": "streams sharded %d/GPU, no data-path collective" % STREAMS_PER_GPU,
                       "realtime_factor_per_stream": SECONDS / (ms_step * 1e-3),
                       "out_samples_per_gpu": n_out, "pipeline_chunks": args.chunks,
                       "kernel_launches_per_step": {"spx_analysis_kernel": args.chunks,
                                                    "spx_tension_kernel": args.chunks,
                                                    "spx_walk_kernel": args.chunks}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms_per_step": {"spx_analysis_kernel": ms_analyze, "spx_tension_kernel": ms_tension,
                                                "spx_walk_kernel": ms_walk},
                         "kernel_avg_launch_ms": {"spx_analysis_kernel": ms_analyze / args.chunks,
                                                  "spx_tension_kernel": ms_tension / args.chunks,
                                                  "spx_walk_kernel": ms_walk / args.chunks},
                         "note": "latency-bound at this size: 256 sequential per-stream walks, one workgroup "
                                 "each (DESIGN.md)"},
        }
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if not args.no_cpu_baseline and world >= 1:
            line["cpu_baseline"] = cpu_baseline(streams)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
