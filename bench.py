"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (config.workload): BASELINE.json configs[3] per GPU -- 256 concurrent 16 kHz mono streams x 10 s of
synthetic speech-like int16, speed 3.5, nonlinear on, duration feedback 0 (speedy_wave.cc:33 default) --
the configuration the metric "Msamples/s processed (16 kHz mono, 3.5x nonlinear)" is quoted on.  One "step" =
one pass of the whole hot path (analysis, tension and walk kernels) over the batch, inputs already resident in HBM.

N > 1: one process per GPU, each with its own 256 streams (weak scaling, streams share nothing; the only
collectives are the work-partition handshake and the barrier / MAX-reduce of the timing, over RCCL).  Launched either
by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly: `python bench.py --gpus N`
starts the N rank processes itself, before anything touches the GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RATE, SECONDS, STREAMS_PER_GPU, SPEED = 16000, 10, 256, 3.5
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_streams(n_streams, n, rank):
    """Distinct speech-like streams, one seed per stream: seed = 1234 + global stream index, global index =
    rank * streams per GPU + i -- the streams of an N-rank run are the first 256 N of one global sequence, so partitions
    can be compared stream by stream (tools/check_scale.py)."""
    from concurrent.futures import ThreadPoolExecutor
    from speedy_amd.synth import speech_like   # its generator is seeded with 1234 + seed
    if os.environ.get("SPX_BENCH_R02_STREAMS"):   # A/B against round-2 numbers only: that round's 32 bases + rotations
        bases = [speech_like(n, RATE, seed=1000 * rank + i) for i in range(min(32, n_streams))]
        return [np.roll(bases[i % 32], (i // 32) * 7919) if i >= 32 else bases[i] for i in range(n_streams)]
    with ThreadPoolExecutor(max(1, min(8, usable_cpus()[0]))) as ex:
        return list(ex.map(lambda i: speech_like(n, RATE, seed=rank * n_streams + i), range(n_streams)))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256
    hardware threads but run the job under cpu.max = 16 CPUs; 256 runnable threads are then throttled to 9x one thread,
    16 threads scale 15x -- tools/cpu_scaling.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(streams, gpu_outputs, budget_s=12.0):
    """The CPU oracle (kind "port": C restatement of the reference path; the reference itself is unbuildable here,
    DESIGN.md "Oracle") on the host cores of this machine: oracle/orc_bench.c -- POSIX threads, one stream per task,
    the speedy_wave.cc write-1000/read loop per stream -- built here with -O3 -march=native -ffp-contract=off.
    A bounded sample of the SAME streams; the output CRC of each sampled stream is compared with the GPU's."""
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cores, quota = usable_cpus()
    n = streams[0].size

    def run(sample, threads):
        buf = np.ascontiguousarray(np.concatenate(sample), np.int16)
        frames = (C.c_long * len(sample))()
        crcs = (C.c_uint32 * len(sample))()
        dt = L.orc_bench_run(buf.ctypes.data, n, len(sample), RATE, 1, SPEED, 1.0, 0.0, 0, 1000, threads, frames, crcs)
        return dt, list(crcs)

    one, _ = run(streams[:1], 1)                       # single-thread rate, also sizes the sample
    k = int(max(cores, min(4 * len(streams), budget_s * cores / max(one, 1e-4))))
    sample = [streams[i % len(streams)] for i in range(k)]
    dt, crcs = run(sample, cores)
    mismatched = sum(1 for i, c in enumerate(crcs)
                     if c != zlib.crc32(np.ascontiguousarray(gpu_outputs[i % len(streams)]).tobytes()))
    return {"value": k * n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "host_hw_threads": os.cpu_count(), "cgroup_cpu_quota": quota,
            "single_thread_msamples_s": n / one / 1e6,
            "thread_scaling": (k * n / dt) / (n / one),
            "output_crc_mismatches_vs_gpu": mismatched,
            "sample": "%d streams (the %d bench streams%s, %d s each), one stream per task on %d POSIX threads, "
                      "oracle built -O3 -march=native -ffp-contract=off (oracle/orc_bench.c)"
                      % (k, len(streams), ", cycled" if k > len(streams) else "", SECONDS, cores)}


def api_many_handles(streams=256, seconds=10.0):
    """The same configuration through the reference's own API (include/sonic2.h): 256 live sonicStream handles on ONE host
    thread, every round writes 1000 frames to each handle and then reads from each (speedy_wave.cc:199-220 per handle).
    A C program (tools/stream_bench.c) in a child process; host-to-device and device-to-host transfers included."""
    exe = os.path.join(ROOT, "speedy_amd", "lib", "stream_bench")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "streambench"])
        best = None
        for _ in range(3):   # a run is ~30 ms long: the fastest of three
            out = subprocess.run([exe, str(streams), str(seconds), "1000", str(SPEED), "1", "rounds", str(RATE)],
                                 capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                return {"error": out.stderr.strip()[-300:]}
            r = json.loads(out.stdout.strip().splitlines()[-1])
            if best is None or r["msamples_per_s"] > best["msamples_per_s"]:
                best = r
        return {"value": best["msamples_per_s"], "unit": "Msamples/s", "streams": streams, "chunk_frames": 1000,
                "x_realtime_per_stream": best["x_realtime_per_stream"], "us_per_round": best["us_per_round"],
                "handles_per_launch_sequence": best["handles_per_sequence"],
                "note": "sonicWriteShortToStream x %d handles, then sonicReadShortFromStream x %d handles, per round; one host "
                        "thread; staged writes of all handles run as one launch sequence (sonic2_pool.hip); synthetic "
                        "speech-like input generated in C" % (streams, streams)}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def config4_shard(rank, reps=10):
    """One GPU's shard of BASELINE configs[4]: 256 streams x 10 s, 16 kHz / 22.05 kHz, mono / stereo, 1.5x / 3.5x, all in ONE
    spx_batch_run_mixed call per step, inputs resident in HBM.  Returns (seconds per step, input frames per step)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_config4 as c4
    from speedy_amd.batch import Plan
    streams = c4.shard_streams(256, seed0=4000 + 256 * rank)
    plans = [Plan(r, False) for r in c4.RATES]
    b = c4.mixed_batch(plans, streams)
    return c4.time_steps(b.run, reps=reps), sum(10 * c4.cfg(i)[0] for i in range(256))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (this process never touches the GPU),
    wait for them, exit non-zero if any failed.  Rank 0 prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 3600
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:          # one rank failed: the others would wait in a collective for ever
                    q.terminate()
        if time.time() > deadline:
            for q in procs:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    sys.exit(rc)


def pcie_pipeline(plan, streams, n, reps, warm=2):
    """PCIe-inclusive steady state (SURVEY.md 8d "first write to last drained read"): pinned host input -> HBM, the step,
    a device-side gather of the produced frames, one copy to pinned host memory -- double-buffered on three HIP
    streams, so that the H2D of batch k+1 and the D2H of batch k-1 overlap the step of batch k.
    Returns seconds per batch."""
    import torch
    from speedy_amd.batch import Batch
    bs = [Batch(plan, [n] * len(streams), 1, SPEED, 1.0, 0.0) for _ in range(2)]
    h_in = torch.empty(bs[0].d_in.numel(), dtype=torch.int16).pin_memory()
    h_in.zero_()
    off = 0
    for x in streams:
        h_in[off:off + x.size] = torch.from_numpy(x)
        off += x.size
    h_out = [torch.zeros(b.d_out.numel(), dtype=torch.int16).pin_memory() for b in bs]
    h_offs = [torch.zeros(b.n + 1, dtype=torch.int64).pin_memory() for b in bs]
    s_h2d, s_run, s_d2h = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in bs]
    ev_done = [torch.cuda.Event() for _ in bs]
    ev_out = [torch.cuda.Event() for _ in bs]
    packed = [None, None]
    totals = []
    t0 = 0.0

    def drain(k):  # output of batch k: wait for its step (long finished in steady state), then one D2H of the exact size
        i = k % 2
        ev_done[i].synchronize()
        total = int(h_offs[i][-1])
        totals.append(total)
        with torch.cuda.stream(s_d2h):
            h_out[i][:total].copy_(packed[i][:total], non_blocking=True)
            ev_out[i].record(s_d2h)

    for k in range(warm + reps):
        if k == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        i = k % 2
        with torch.cuda.stream(s_h2d):
            if k >= 2:
                s_h2d.wait_event(ev_done[i])      # the step of batch k-2 has finished reading this input buffer
            bs[i].d_in.copy_(h_in, non_blocking=True)
            ev_in[i].record(s_h2d)
        with torch.cuda.stream(s_run):
            s_run.wait_event(ev_in[i])
            if k >= 2:
                s_run.wait_event(ev_out[i])       # the output of batch k-2 has left this buffer
            bs[i].run(stream=s_run)
            packed[i], d_offs = bs[i].pack_outputs(stream=s_run)
            h_offs[i].copy_(d_offs, non_blocking=True)
            ev_done[i].record(s_run)
        if k >= 1:
            drain(k - 1)
    drain(warm + reps - 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    last = (warm + reps - 1) % 2      # what arrived in host memory is what the device packed (outside the timed region)
    assert torch.equal(h_out[last][:totals[-1]], packed[last][:totals[-1]].cpu()), "PCIe pipeline: host copy differs"
    return dt, totals[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] shard (256 mixed-rate streams in one call)")
    ap.add_argument("--no-api", action="store_true", help="skip the many-handle run of the drop-in API (tools/stream_bench.c)")
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("SPX_CHUNKS", "1")),
                    help="time chunks per stream inside one spx_batch_run (analysis of chunk c+1 overlaps the walk of c)")
    ap.add_argument("--crc-out", default=None, help="write this rank's per-stream output CRC-32s to CRC_OUT.rank<r>.json "
                    "(tests: N-rank runs must produce the same bytes per stream as solo runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for "
                    "a functional check of the N > 1 path when several ranks share one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # never returns

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (the product has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dist = None
    red_dev = "cuda"
    backend = args.backend
    rank_info, handshake_ms = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and world > ndev:
            # RCCL needs one device per rank.  No silent downgrade: a scaling run on a node that shows fewer GPUs than ranks
            # is a mis-provisioned run, not a slower one (several ranks sharing a GPU: ask for it with --backend gloo)
            sys.exit("bench.py: WORLD_SIZE=%d but only %d GPU(s) visible; RCCL (--backend nccl) needs one device per rank. "
                     "Use --backend gloo for a functional check with ranks sharing a GPU." % (world, ndev))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
            red_dev = "cpu"
        if world > ndev:
            # ranks share a GPU: the batch engine's concurrent mode counts the polling workgroups of ONE process
            # (spx_engine.hip SpxDevGuard); the engine also detects this itself through a per-device lock file
            os.environ["SPX_SHARED_GPU"] = "1"
        # work-partition handshake: every rank announces its shard (stream count, input frames)
        from speedy_amd.dist import handshake
        t_h = time.perf_counter()
        layout = handshake(dist, STREAMS_PER_GPU, STREAMS_PER_GPU * RATE * SECONDS, device=red_dev)
        if red_dev == "cuda":
            torch.cuda.synchronize()
        handshake_ms = (time.perf_counter() - t_h) * 1e3
        assert layout.shape == (world, 2) and int(layout[:, 0].sum()) == world * STREAMS_PER_GPU
        info = {"rank": rank, "local_rank": local_rank, "device_id": dev_index, "device": torch.cuda.get_device_name(dev_index),
                "host": socket.gethostname(), "pid": os.getpid()}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, info)

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

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

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
    dt = max_over_ranks(dt)
    n_in = n * STREAMS_PER_GPU
    outs = b.results()
    n_out = int(sum(o.size for o in outs))
    if args.crc_out:
        with open("%s.rank%d.json" % (args.crc_out, rank), "w") as f:
            json.dump([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs], f)

    # PCIe-inclusive steady state on every rank, MAX over ranks (reported beside `value`, which by the bench contract is
    # the rate with inputs already resident in HBM)
    pcie = None
    if not args.no_pcie:
        barrier()
        # three passes, the fastest counts: a pass is only ~25 ms long and shares the box's PCIe and host with whatever else runs
        dt1, total = min(pcie_pipeline(plan, streams, n, reps=max(4, min(args.steps, 10))) for _ in range(3))
        assert total == n_out, (total, n_out)
        dt1 = max_over_ranks(dt1)
        pcie = {"value": n_in * world / dt1 / 1e6, "unit": "Msamples/s", "ms_per_step": dt1 * 1e3,
                "vs_resident_step": dt1 / (dt / args.steps),
                "note": "every rank, MAX over ranks: pinned host int16 input -> HBM, the step, device-side gather, one D2H "
                        "of the produced int16 output; double-buffered on three HIP streams (H2D of batch k+1 and D2H "
                        "of batch k-1 overlap the step of batch k)"}

    # BASELINE configs[4], one GPU's shard per rank (256 mixed-rate streams in one call), MAX over ranks
    c4 = None
    if not args.no_config4:
        barrier()
        dt4, frames4 = config4_shard(rank)
        dt4 = max_over_ranks(dt4)
        c4 = {"value": frames4 * world / dt4 / 1e6, "unit": "Msamples/s", "ms_per_step": dt4 * 1e3, "streams_per_gpu": 256,
              "note": "BASELINE configs[4], every rank its shard of 256 streams x 10 s (stream i: 16 kHz if i even else 22.05 kHz; "
                      "mono if (i/2) even else stereo; speed 1.5 if (i/4) even else 3.5; nonlinear 1), ONE spx_batch_run_mixed "
                      "call per step, inputs resident in HBM, MAX over ranks; input sample frames of all ranks / that time"}

    if rank == 0:
        total_in = n_in * world * args.steps
        ms_step = dt / args.steps * 1e3
        ms_analyze = sa.value / max(1, nc.value)
        ms_walk = sw.value / max(1, nc.value)
        algo_bytes = 2 * 1 * (n_in + n_out)  # int16 read once + int16 written once, per launch
        # the kernels that served the batch, by the names a profiler prints (template arguments included)
        k_analysis, k_tension, k_walk = L.spx_batch_kernel_names(plan.h, STREAMS_PER_GPU, 1, 1).decode().split(";")
        dom, dom_ms = (k_walk, ms_walk) if ms_walk >= ms_analyze else (k_analysis, ms_analyze)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        traffic_note = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                pt = json.load(open(pj))
                traffic = pt.get(dom, {}).get("hbm_bytes_per_launch")
                traffic_note = pt.get("_note")
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s processed (16 kHz mono, 3.5x nonlinear)",
            "value": total_in / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "value_definition": "input sample frames of all ranks x steps / MAX-over-ranks wall time of the timed steps, inputs "
                                "resident in HBM when the timed region starts (the bench contract's definition); SURVEY 8(d)'s "
                                "first-write-to-last-drained-read rate is `pcie_inclusive`, the drop-in API's rate with 256 "
                                "live sonicStream handles is `api_256_handles`",
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 samples; f64 DFT, f32 features, int32 AMDF/OLA",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d streams/GPU x %d s, 16 kHz mono int16, speed 3.5, "
                                   "nonlinear 1.0, feedback 0" % (STREAMS_PER_GPU, SECONDS),
                       "streams_per_gpu": STREAMS_PER_GPU, "samples_per_stream": n,
                       "parallelism": "streams sharded %d/GPU, no data-path collective" % STREAMS_PER_GPU,
                       "launcher": "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else
                                   ("bench.py --gpus (self-spawned ranks)" if world > 1 else "single process"),
                       "backend": (backend if world > 1 else None),
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None),
                       "n_ranks_seen": (int(layout.shape[0]) if world > 1 else 1),
                       "handshake_ms": handshake_ms,
                       "ranks": rank_info if rank_info is not None else
                                [{"rank": 0, "device_id": dev_index, "device": torch.cuda.get_device_name(dev_index)}],
                       "stream_seeds": "1234 + global stream index (rank * %d + i), all distinct" % STREAMS_PER_GPU,
                       "realtime_factor_per_stream": SECONDS / (ms_step * 1e-3),
                       "out_samples_per_gpu": n_out, "pipeline_chunks": args.chunks,
                       "kernel_launches_per_step": {k_analysis: args.chunks, k_tension: args.chunks, k_walk: args.chunks}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms_per_step": {k_analysis: ms_analyze, k_tension: ms_tension, k_walk: ms_walk},
                         "kernel_avg_launch_ms": {k_analysis: ms_analyze / args.chunks, k_tension: ms_tension / args.chunks,
                                                  k_walk: ms_walk / args.chunks},
                         "limiter": "latency, not HBM: 256 per-stream chains of ~1300 dependent pitch steps, one "
                                    "workgroup per CU (DESIGN.md 5.3, 6); `bound` names the roofline the contract asks "
                                    "to be priced against"},
        }
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if c4 is not None:
            line["config4_shard"] = c4
        if not args.no_api and world == 1:
            line["api_256_handles"] = api_many_handles()
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (bench contract)
            line["cpu_baseline"] = cpu_baseline(streams, outs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
