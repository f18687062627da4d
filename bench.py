"""bench.py -- the headline metric of BASELINE.json on MI355X.

Workload (config.workload): BASELINE.json configs[3] per GPU -- 256 concurrent 16 kHz mono streams x 10 s of
synthetic speech-like int16, speed 3.5, nonlinear on, duration feedback 0 (speedy_wave.cc:33 default) --
the configuration the metric "Msamples/s processed (16 kHz mono, 3.5x nonlinear)" is quoted on.  One "step" =
one pass of the whole hot path (analysis, tension and walk kernels) over the batch, inputs already resident in HBM.

N > 1: one process per GPU, each with its own 256 streams (weak scaling, streams share nothing; the only
collectives are the work-partition handshake and the barrier / MAX-reduce of the timing, over RCCL).  Launched either
by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly: `python bench.py --gpus N`
starts the N rank processes itself, before anything touches the GPU.

Beside `value` the line carries (every figure ONE timed window of consecutive steps, no best-of-N anywhere):
  roofline            HBM (the contract's), plus `latency` (walk kernel: cycles per dependent pitch step against the modelled floor)
                      and `valu_fp64` (analysis kernel alone: fp64 operations of the DFT spec per second against the vector peak)
  large_batch         2 048 streams x 10 s in ONE call: the throughput regime, HBM fraction
  other_rates         256 x 10 s at 44.1 kHz mono and at 48 kHz stereo (beyond BASELINE's rates), CRC-checked against the CPU port
  pcie_inclusive      pinned host -> HBM -> step -> gather -> host, double-buffered
  config4_shard       one GPU's 256-stream shard of BASELINE configs[4] (weak: every rank its shard)
  config4_full        ALL of configs[4]: the fixed batch of --total-streams (2 048) mixed streams, rank r of N takes block r
                      (STRONG scaling; N = 1 runs all of it in one call)
  api_256_handles     the drop-in API with 256 live sonicStream handles on one host thread (write all, then read all)
  api_percall_256     ... in the reference's own call order (write, read, next handle) on one thread
  api_threads         ... and on 16 host threads x 16 handles (64 x 4 and 256 x 1 beside it)
  cpu_baseline        the CPU port on the host cores, with the output CRCs of the GPU legs checked against it
  cpu_baseline_fftw   the same with the reference's FFT library when the box has libfftw3 (dlopen); {"fftw": "absent"} otherwise

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

# HIP maps streams onto four hardware queues by default, and a queue runs its kernels in order.  The PCIe leg drives three
# streams of its own beside the library's two side streams: with four queues two of the five share one, and which two depends on
# the order of creation -- 2 of 8 runs of this file read 3.4-3.8 ms per batch there instead of 2.05 (profiles/r04/r05u_hw_queues.txt).
# Eight queues give every stream its own; nothing else in the line moves.  (Read by the HIP runtime at its first call: set here,
# before torch is imported; an application with copy streams of its own beside this library should do the same, INTEGRATION.md.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

RATE, SECONDS, STREAMS_PER_GPU, SPEED = 16000, 10, 256, 3.5
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # fp64 vector instructions x lanes per second: 39.3 T (see `peak_source` in the line)


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


def bind_to_gpu_numa(dev_index):
    """Pin this process's host threads to the CPUs of its GPU's NUMA node (with the host-to-device link as the
    end-to-end bound and eight ranks on two sockets, an unbound rank measures placement luck).  The device's PCI address comes from
    torch's device properties (no GPU work), the node and its CPUs from sysfs; the affinity is only ever NARROWED (the CPUs the
    process may already use, intersected with the node's).  Returns what was found and done -- it goes into config.ranks."""
    info = {"numa_node": None, "cpus_allowed": len(os.sched_getaffinity(0)), "bound": False, "pci": None}
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev_index)
        pci = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        info["pci"] = pci
        base = "/sys/bus/pci/devices/" + pci
        node = int(open(base + "/numa_node").read().strip())
        info["numa_node"] = node
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        mine = os.sched_getaffinity(0)
        want = mine & cpus
        if node >= 0 and want and want != mine and not os.environ.get("SPX_BENCH_NO_BIND"):
            os.sched_setaffinity(0, want)
            info["bound"] = True
        info["cpus_allowed"] = len(os.sched_getaffinity(0))
        info["node_cpus"] = len(cpus)
    except Exception as e:  # noqa: BLE001  (no sysfs entry, no such property: recorded, never fatal)
        info["error"] = repr(e)[:120]
    return info


def cpu_baseline(streams, gpu_outputs, budget_s=12.0, c4_checks=None, rate_checks=None):
    """The CPU oracle (kind "port": C restatement of the reference path; the reference itself is unbuildable here,
    DESIGN.md "Oracle") on the host cores of this machine: oracle/orc_bench.c -- POSIX threads, one stream per task,
    the speedy_wave.cc write-1000/read loop per stream -- built here with -O3 -march=native -ffp-contract=off.
    A bounded sample of the SAME streams; the output CRC of each sampled stream is compared with the GPU's.
    c4_checks: {name: (streams, global ids, gpu crcs)} of the configs[4] legs -- a sample of every kind of every shard is run
    through the port as well and its CRCs compared (the checker's only other use in this file)."""
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    cores, quota = usable_cpus()
    n = streams[0].size

    def run(sample, threads, rate=RATE, ch=1, speed=SPEED):
        buf = np.ascontiguousarray(np.concatenate(sample), np.int16)
        frames = (C.c_long * len(sample))()
        crcs = (C.c_uint32 * len(sample))()
        dt = L.orc_bench_run(buf.ctypes.data, sample[0].size // ch, len(sample), rate, ch, speed, 1.0, 0.0, 0, 1000, threads,
                             frames, crcs)
        return dt, list(crcs)

    one, _ = run(streams[:1], 1)                       # single-thread rate, also sizes the sample
    # (at most eight cycles of the bench streams: 2 048 x 10 s = 17 s of CPU work at the port's 19 Msamples/s per thread)
    k = int(max(cores, min(8 * len(streams), budget_s * cores / max(one, 1e-4))))
    sample = [streams[i % len(streams)] for i in range(k)]
    dt, crcs = run(sample, cores)
    mismatched = sum(1 for i, c in enumerate(crcs)
                     if c != zlib.crc32(np.ascontiguousarray(gpu_outputs[i % len(streams)]).tobytes()))
    res = {"value": k * n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model(), "host_hw_threads": os.cpu_count(), "cgroup_cpu_quota": quota,
           "single_thread_msamples_s": n / one / 1e6,
           "thread_scaling": (k * n / dt) / (n / one),
           "output_crc_mismatches_vs_gpu": mismatched,
           "sample": "%d streams (the %d bench streams%s, %d s each), one stream per task on %d POSIX threads, "
                     "oracle built -O3 -march=native -ffp-contract=off (oracle/orc_bench.c)"
                     % (k, len(streams), ", cycled" if k > len(streams) else "", SECONDS, cores)}
    # "if libfftw3 happens to exist on the box, add an FFTW-backed run as the reference FFTW/CPU path".  Looked up at
    # run time (dlopen, oracle/orc_speedy.c orc_fftw_available); the call shape is the reference's (speedy.c:228-231, 458-473:
    # fftw_plan_dft_1d, N-point complex, FFTW_ESTIMATE, cabs).  Same sample, same threads; its output against the port's.
    L.orc_fftw_available.restype = C.c_int
    L.orc_set_fft_backend.restype = C.c_int
    L.orc_set_fft_backend.argtypes = [C.c_int]
    if L.orc_fftw_available() and L.orc_set_fft_backend(1) == 1:
        try:
            fdt, fcrcs = run(sample, cores)
            W = int(1.5 * RATE / 100.0)
            L.orc_spectrum_magnitudes.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
            rng = np.random.default_rng(1)
            num = den = 0.0
            for _ in range(32):
                x = (rng.standard_normal(W) * 0.1).astype(np.float32)
                a, b = np.zeros(2 * W, np.float32), np.zeros(2 * W, np.float32)
                L.orc_set_fft_backend(1)
                L.orc_spectrum_magnitudes(W, x.ctypes.data, a.ctypes.data)
                L.orc_set_fft_backend(0)
                L.orc_spectrum_magnitudes(W, x.ctypes.data, b.ctypes.data)
                num += float(np.sum(b.astype(np.float64) ** 2))
                den += float(np.sum((a.astype(np.float64) - b.astype(np.float64)) ** 2))
            res["cpu_baseline_fftw"] = {
                "value": k * n / fdt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port with the reference's FFT library",
                "fftw": "libfftw3 (dlopen), fftw_plan_dft_1d N-point complex, FFTW_ESTIMATE (speedy.c:228-231,458-473)",
                "streams_with_output_equal_to_the_port": sum(1 for a, b in zip(fcrcs, crcs) if a == b), "streams": len(crcs),
                "spectrogram_snr_db_vs_port": (10.0 * float(np.log10(num / den)) if den > 0 else float("inf"))}
        finally:
            L.orc_set_fft_backend(0)
    else:
        res["cpu_baseline_fftw"] = {"fftw": "absent", "note": "no libfftw3.so.3 on this box (dlopen); the baseline above is the port's own "
                                    "double-precision mixed-radix DFT (SURVEY 8d)"}
    if c4_checks:
        from speedy_amd import config4 as C4
        chk = {}
        for name, (c4_streams, ids, gpu_crcs) in c4_checks.items():
            # per kind (rate, channels, speed): the first stream of that kind in every block of 256 (every GPU's shard), at most 16
            bad = total = 0
            for kind in range(8):
                pick = [j for j, i in enumerate(ids) if C4.kind(i) == kind and (i % 256) < 8][:16]
                if not pick:
                    continue
                rate, ch, speed = C4.cfg(ids[pick[0]])
                _, cr = run([c4_streams[j] for j in pick], min(cores, len(pick)), rate, ch, speed)
                bad += sum(1 for j, c in zip(pick, cr) if c != gpu_crcs[j])
                total += len(pick)
            chk[name] = {"streams_checked": total, "output_crc_mismatches_vs_gpu": bad}
        res["config4_crc_check"] = chk
    if rate_checks:
        # the other_rates legs: every distinct signal of each through the port, output CRCs compared
        chk = {}
        for name, (base, rate, ch, gpu_crcs) in rate_checks.items():
            _, cr = run(base, min(cores, len(base)), rate, ch, SPEED)
            chk[name] = {"streams_checked": len(base), "output_crc_mismatches_vs_gpu": sum(1 for a, c in zip(cr, gpu_crcs) if a != c)}
        res["other_rates_crc_check"] = chk
    return res


def api_many_handles(streams=256, seconds=40.0, order="rounds"):
    """The same configuration through the reference's own API (include/sonic2.h) with 256 live sonicStream handles, a C program
    (tools/stream_bench.c) in a child process; host-to-device and device-to-host transfers included.  ONE run, like every other
    figure of the line.  `order`:
      rounds      one host thread; every round writes 1000 frames to each handle and THEN reads from each (api_256_handles)
      percall     one host thread in the reference's own call order -- write a chunk, read right away, next handle
                  (speedy_wave.cc:199-220, sonic_test.cc:384-392): one launch sequence per handle per write (api_percall_256)
      threads:T   T host threads, each running that write -> read loop over its own 256 / T handles: the library combines what
                  the threads stage into common launch sequences (sonic2_pool.hip, flat combining; api_threads)"""
    exe = os.path.join(ROOT, "speedy_amd", "lib", "stream_bench")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "streambench"])
        out = subprocess.run([exe, str(streams), str(seconds), "1000", str(SPEED), "1", order, str(RATE)],
                             capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            return {"error": out.stderr.strip()[-300:]}
        best = json.loads(out.stdout.strip().splitlines()[-1])
        res = {"value": best["msamples_per_s"], "unit": "Msamples/s", "streams": streams, "chunk_frames": 1000, "order": order,
               "x_realtime_per_stream": best["x_realtime_per_stream"], "us_per_round": best["us_per_round"],
               "handles_per_launch_sequence": best["handles_per_sequence"], "seconds_per_handle": seconds}
        if order == "rounds":
            res["note"] = ("sonicWriteShortToStream x %d handles, then sonicReadShortFromStream x %d handles, per round; one host "
                           "thread; staged writes of all handles run as one launch sequence (sonic2_pool.hip); synthetic "
                           "speech-like input generated in C; one run" % (streams, streams))
        elif order == "percall":
            res["note"] = ("the reference's call order on one host thread: write 1000 frames to a handle, read, next handle -- every read "
                           "needs its write's result, so every write is a launch sequence of its own")
        else:
            res["threads"] = best.get("threads")
            res["note"] = ("%s host threads x %d handles each, every thread in the reference's call order (write, read, next handle); a "
                           "thread cannot have more than one write outstanding, so a launch sequence carries at most one handle per "
                           "thread: the ceiling is threads x 1000 frames per sequence latency (~0.1 ms), whatever the handle count"
                           % (best.get("threads"), streams // max(1, int(best.get("threads") or 1))))
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def time_window(run, reps, warm):
    """ONE timed window of `reps` consecutive steps after `warm` untimed ones (each followed by a synchronisation: the
    engine's launch-mode trial of a batch shape needs three completed calls).  Seconds per step."""
    import torch
    for _ in range(warm):
        run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def config4_leg(ids, reps, warm=4, pipeline=True):
    """The global configs[4] streams `ids` in ONE spx_batch_run_mixed call per step, inputs resident in HBM.
    Returns dict(seconds per step, input frames, streams, crcs, algorithmic bytes, kernel names)."""
    from speedy_amd import config4 as C4
    from speedy_amd.batch import Plan
    import torch
    streams = C4.make_streams(ids, threads=max(1, min(16, usable_cpus()[0])))
    plans = [Plan(r, False) for r in C4.RATES]
    b = C4.mixed_batch(plans, ids, streams)
    dt_plain = time_window(b.run, reps, warm)
    counts = b.counts()
    crcs = b.crcs()
    steps = b.step_counts()
    dt = dt_plain
    pipelined = False
    if pipeline and len(ids) <= torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count:
        # consecutive steps software-pipelined (spx_batch_run_mixed_ahead): two batches with the same resident input take turns
        b2 = C4.mixed_batch(plans, ids, streams)
        turn = [b, b2]
        b.d_out.zero_()
        for k in range(warm):
            turn[k % 2].run_ahead()
        torch.cuda.synchronize()
        # (a window of 40 steps at least: the loop starts empty -- the first call has no predecessor to run beside -- and that one
        # ramp is 6 % of a 10-step window, 1.5 % of this one; the headline's window is 100 steps for the same reason)
        reps_p = max(reps, 40)
        t0 = time.perf_counter()
        for k in range(reps_p):
            turn[(warm + k) % 2].run_ahead()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps_p
        assert b.crcs() == crcs and b2.crcs() == crcs, "the pipelined mixed calls changed the output"
        pipelined = True
        del b2
        dt_pair = dt
        # ... and through the owning pipeline object (spx_pipeline_create_mixed, four buffer sets, outputs left on the device -- the
        # headline's method; round 6: the groups' walk kernels of consecutive batches overlap on the library's walk streams)
        from speedy_amd.batch import Pipeline
        pipe = Pipeline(plans, [C4.SECONDS * C4.cfg(i)[0] for i in ids], [C4.cfg(i)[1] for i in ids], [C4.cfg(i)[2] for i in ids], 1.0, 0.0,
                        depth=4, device_out=True, plan_index=[C4.RATES.index(C4.cfg(i)[0]) for i in ids])
        ts = [pipe.submit(b.d_in) for _ in range(max(warm, 6))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts += [pipe.submit(b.d_in) for _ in range(reps_p)]
        torch.cuda.synchronize()
        dt_pipe = (time.perf_counter() - t0) / reps_p
        for t in ts[-pipe.depth:]:   # every buffer set holds the plain call's bytes
            assert [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in pipe.results(t)] == crcs, "the pipeline object's mixed batches differ"
        pipe.close()
        del pipe
        dt = dt_pipe
        through = "spx_pipeline (mixed), 4 buffer sets, outputs on the device"
    else:
        dt_pair = dt_pipe = None
        through = "spx_batch_run_mixed, call after call"
    return {"dt": dt, "dt_plain": dt_plain, "dt_pair": dt_pair, "dt_pipe": dt_pipe, "through": through,
            "pipelined": pipelined, "frames": C4.input_frames(ids), "streams": streams, "crcs": crcs,
            "algo_bytes": C4.algorithmic_bytes(ids, counts), "steps": steps, "out_frames": int(counts.sum())}


def other_rate_leg(rate, ch, reps=5, warm=4, distinct=8):
    """Widening beyond BASELINE's rates (round 4): 256 streams x 10 s at `rate`, `ch` channels, the headline's speed and
    nonlinear factor, ONE spx_batch_run per step, inputs resident in HBM.  `distinct` different signals (seed 7000 + i), cycled
    over the 256 streams; their output CRCs go to the CPU port for checking (cpu_baseline)."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    n = SECONDS * rate
    plan = Plan(rate, False)
    base = [speech_like(n, rate, seed=7000 + i, channels=ch) for i in range(distinct)]
    b = Batch(plan, [n] * STREAMS_PER_GPU, ch, SPEED, 1.0, 0.0)
    b.upload([base[i % distinct] for i in range(STREAMS_PER_GPU)])
    dt = time_window(b.run, reps, warm)
    outs = b.results()
    crcs = [zlib.crc32(np.ascontiguousarray(outs[i]).tobytes()) for i in range(distinct)]
    ka, kt, kw = plan.L.spx_batch_kernel_names(plan.h, STREAMS_PER_GPU, ch, 1).decode().split(";")
    n_out = int(sum(o.size for o in outs)) // ch
    res = {"rate": rate, "channels": ch, "ms_per_step": dt * 1e3, "value": n * STREAMS_PER_GPU / dt / 1e6, "unit": "Msamples/s",
           "hbm_frac": 2 * ch * (n * STREAMS_PER_GPU + n_out) / dt / 1e9 / HBM_PEAK_GBS,
           "kernels": {"analysis": ka, "tension": kt, "walk": kw}}
    del b
    torch.cuda.empty_cache()
    return res, (base, rate, ch, crcs)


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


PCIE_LINK_PEAK_GBS = 63.0   # PCIe 5.0 x16, one direction: 32 GT/s x 16 lanes x 128/130 / 8 (the link the MI355X OAM sits on)


def pcie_pipeline(plan, streams, n, reps=200, warm=50, depth=4):
    """PCIe-inclusive STEADY STATE (SURVEY.md 8d "first write to last drained read"; the reference's caller loop,
    speedy_wave.cc:199-220, with a batch of streams as its unit) on the library's owning pipeline object (spx_pipeline,
    include/speedy_hip.h): pinned host input -> spx_pipeline_submit -> spx_pipeline_wait -> the produced frames of every stream,
    densely packed, in pinned host memory.  The library issues the H2D copy, the overlapped batch call and the gather into host
    memory itself; this loop only submits and, `depth - 1` tickets later, waits -- so up to `depth` batches are in flight.
    Timed over `reps` batches after `warm` untimed ones WITHOUT a pause in between: the figure is the interval between the
    completions of ticket warm - 1 and ticket warm + reps - 1, i.e. the pipeline's output rate in its steady phase.
    Returns a dict (seconds per batch, per-batch completion intervals, the H2D link rate measured alone, CRCs of the last batch)."""
    import torch
    from speedy_amd.batch import Pipeline
    pipe = Pipeline(plan, [n] * len(streams), 1, SPEED, 1.0, 0.0, depth=depth)
    h_in = torch.from_numpy(pipe.pack(streams)).pin_memory()
    # the link alone: the same 82 MB, host to device, copy after copy on one stream (the floor of this leg)
    d_probe = torch.empty(h_in.numel(), dtype=torch.int16, device="cuda")
    s_probe = torch.cuda.Stream()
    with torch.cuda.stream(s_probe):
        for _ in range(5):
            d_probe.copy_(h_in, non_blocking=True)
        s_probe.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            d_probe.copy_(h_in, non_blocking=True)
        s_probe.synchronize()
    h2d_alone_s = (time.perf_counter() - t0) / 20
    del d_probe
    lag = pipe.depth - 1
    tickets, done = [], []
    total = warm + reps
    for k in range(total):
        tickets.append(pipe.submit(h_in))
        if k >= lag:
            pipe.wait(tickets[k - lag])        # the drained read: batch k - lag is in host memory
            done.append(time.perf_counter())
    for t in tickets[total - lag:]:
        pipe.wait(t)
        done.append(time.perf_counter())
    dt = (done[total - 1] - done[warm - 1]) / reps
    iv = np.diff(np.asarray(done[warm - 1:total])) * 1e3
    outs = pipe.results(tickets[-1])
    res = {"dt": dt, "total": int(sum(o.size for o in outs)), "crcs": [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs],
           "h2d_alone_s": h2d_alone_s, "depth": pipe.depth, "reps": reps, "warm": warm,
           "interval_ms": {"p10": float(np.percentile(iv, 10)), "p50": float(np.percentile(iv, 50)), "p90": float(np.percentile(iv, 90)),
                           "max": float(iv.max())},
           "walk_form": int(plan.L.spx_debug_last_walk_form()), "last_mode": int(plan.L.spx_debug_last_call_concurrent())}
    pipe.close()
    return res


def standalone_analysis_ms(plan, batch, reps=10, warm=3):
    """The analysis kernel ALONE on the batch (spx_batch_analyze on the current stream: nothing beside it), HIP events around
    one window of `reps` launches.  Milliseconds per launch."""
    import torch
    L = plan.L
    hs = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = L.spx_batch_analyze(plan.h, batch.jobs, batch.n, batch.d_in.data_ptr(), batch.d_ws.data_ptr(), batch.d_ws.numel(), None, hs)
        if rc != 0:
            raise RuntimeError("spx_batch_analyze: " + L.spx_last_error().decode())
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def load_scratch_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "gpurun_out", name)))
    except Exception:  # noqa: BLE001
        return None


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: a window long enough that the drain of the pipelined loop -- the last walk kernel runs 2 ms behind the last submit,
    # inside the timed region -- is 2 % of it and not 5: 100 steps take a tenth of a second)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--pcie-child", type=str, default=None, metavar="DEV,RANK,REPS",
                    help="internal: run ONLY the PCIe-inclusive pipeline on device DEV with rank RANK's streams and print one JSON object")
    ap.add_argument("--no-config4", action="store_true", help="skip both configs[4] legs (the 256-stream shard and the full batch)")
    ap.add_argument("--no-config4-full", action="store_true", help="skip the full configs[4] batch (strong scaling leg)")
    ap.add_argument("--total-streams", type=int, default=2048,
                    help="configs[4] as a whole: the fixed batch of this many mixed streams, N ranks take 1/N each (strong "
                         "scaling; N = 1 runs all of it in one call); reported as config4_full")
    ap.add_argument("--no-pipeline", action="store_true", help="time spx_batch_run on ONE batch, call after call (no software pipelining of consecutive steps)")
    ap.add_argument("--no-unpipelined", action="store_true", help="skip the window of plain spx_batch_run calls behind `unpipelined` (profiling runs)")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the 2 048-stream call of the headline kind")
    ap.add_argument("--no-other-rates", action="store_true", help="skip the 44.1 kHz mono / 48 kHz stereo calls (widening row; N = 1 only)")
    ap.add_argument("--no-api", action="store_true", help="skip the many-handle run of the drop-in API (tools/stream_bench.c)")
    ap.add_argument("--serial", action="store_true", help="spx_set_concurrent(0): the three kernels of a step back to back on one "
                    "stream (what the per-kernel PMC passes behind roofline.traffic need: one kernel in flight at a time)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="force this many time chunks per stream inside every spx_batch_run (analysis of chunk c+1 overlaps the walk "
                    "of c); default: the library's own choice -- one for the headline batch, two above two streams per CU")
    ap.add_argument("--crc-out", default=None, help="write this rank's per-stream output CRC-32s to CRC_OUT.rank<r>.json "
                    "(tests: N-rank runs must produce the same bytes per stream as solo runs); the configs[4] legs' CRCs "
                    "go to CRC_OUT.c4.rank<r>.json keyed by global stream index")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for "
                    "a functional check of the N > 1 path when several ranks share one GPU)")
    args = ap.parse_args()
    if args.pcie_child:
        # The PCIe leg in a process of its own, started before the parent touches the GPU (see there).
        import torch
        from speedy_amd.batch import Plan
        dev, rk, reps = (int(v) for v in args.pcie_child.split(","))
        torch.cuda.set_device(dev)
        numa_ = bind_to_gpu_numa(dev)
        n_ = RATE * SECONDS
        plan_ = Plan(RATE, False)
        streams_ = make_streams(STREAMS_PER_GPU, n_, rk)
        if os.environ.get("SPX_SHARED_GPU"):
            pass   # (ranks sharing a GPU: the engine runs its kernels in sequence, as in the parent)
        # (round 5: the steady phase -- >= 200 timed batches behind >= 50 untimed ones, no pause between them)
        res_ = pcie_pipeline(plan_, streams_, n_, reps=max(200, reps), warm=int(os.environ.get("SPX_BENCH_PCIE_WARM", "50")),
                             depth=int(os.environ.get("SPX_BENCH_PCIE_DEPTH", "4")))
        res_["numa"] = numa_
        print(json.dumps(res_), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)   # never returns

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()     # (does not initialise the GPU)
    if ndev < 1:
        sys.exit("bench.py needs an MI355X (the product has no CPU path)")
    dev_index = local_rank % max(1, ndev)
    # The PCIe-inclusive leg runs FIRST, in a child process, before this process has touched the GPU: how HIP streams fall onto
    # hardware queues (and how those are shared between processes) depends on what else holds queues on the device, and behind
    # the other legs -- or beside an initialised parent -- this leg read 2.3, 2.4 or 4 ms per batch where it reads 2.0 alone
    # (profiles/r04/r05u_hw_queues.txt).  Every rank runs its own (MAX over ranks below).
    pcie_early = None
    if not args.no_pcie:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--pcie-child",
                                "%d,%d,%d" % (dev_index, rank, max(10, args.steps))], capture_output=True, text=True, timeout=900)
        if child.returncode != 0:
            sys.exit("bench.py: the PCIe leg failed: " + child.stderr[-1500:])
        pcie_early = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith("{")][-1])
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (the product has no CPU path)")
    torch.cuda.set_device(dev_index)
    numa = bind_to_gpu_numa(dev_index)
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
                "host": socket.gethostname(), "pid": os.getpid(), "numa_node": numa["numa_node"], "cpus_allowed": numa["cpus_allowed"],
                "bound_to_numa_node": numa["bound"], "pci": numa["pci"],
                "h2d_alone_gbs": (2 * STREAMS_PER_GPU * RATE * SECONDS / pcie_early["h2d_alone_s"] / 1e9) if pcie_early else None}
        rank_info = [None] * world
        dist.all_gather_object(rank_info, info)

    from speedy_amd.batch import Batch, Plan
    n = RATE * SECONDS
    streams = make_streams(STREAMS_PER_GPU, n, rank)
    plan = Plan(RATE, False)
    b = Batch(plan, [n] * STREAMS_PER_GPU, 1, SPEED, 1.0, 0.0)
    b.upload(streams)
    L = plan.L
    # (until round 6 this line ran unconditionally with 1, and a chunk count SET by the caller is binding: the `large_batch` and
    # `config4_full` calls then ran as ONE time chunk where every other caller of the library gets two)
    if args.chunks is not None:
        L.spx_set_pipeline_chunks(args.chunks)
    args.chunks = args.chunks or 1    # what the headline batch (one stream per CU) runs with: the per-kernel accounting below
    if args.serial:
        L.spx_set_concurrent(0)

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

    def sum_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # Consecutive steps are software-pipelined by the library (spx_batch_run_overlapped, include/speedy_hip.h): three Batch objects
    # with the SAME resident input take turns, step k + 1's analysis and tension kernels run beside step k's walk kernel, and
    # the walk kernels of consecutive steps overlap too (two streams of the library's taking turns).
    # Every step is the whole hot path over one batch of 256 streams; nothing is cached or skipped.  (--no-pipeline: one
    # Batch, spx_batch_run call after call, as rounds 1-4 timed it; reported as `unpipelined` in every line.)
    # Round 5: through the library's OWNING pipeline object (spx_pipeline, include/speedy_hip.h) -- the same object the
    # PCIe-inclusive leg runs on, here with the input resident in HBM (spx_pipeline_submit(.., in_is_device = 1)) and the outputs
    # left in device memory (SPX_PIPELINE_DEVICE_OUT): the buffer sets, the streams and the relaxed stream order are the
    # library's business, not this file's.
    pipe = None
    tickets = []
    if not args.no_pipeline:
        from speedy_amd.batch import Pipeline
        pipe = Pipeline(plan, [n] * STREAMS_PER_GPU, 1, SPEED, 1.0, 0.0, depth=int(os.environ.get("SPX_BENCH_DEPTH", "4")), device_out=True)

    pipe_depth = pipe.depth if pipe is not None else 1

    def step(k):
        if pipe is not None:
            tickets.append(pipe.submit(b.d_in))
        else:
            b.run()

    # the unpipelined figure (also warms up); --no-unpipelined: profiling runs, whose per-kernel averages should be the timed loop's
    dt_single = None if args.no_unpipelined else time_window(b.run, reps=max(5, args.steps), warm=max(3, args.warmup))
    if dt_single is None:
        # (the batch whose outputs and chain lengths the line reports; its kernels in sequence: a profiler that serialises kernels
        # must not meet the concurrent mode's polling kernels)
        L.spx_set_concurrent(0)
        b.run()
        torch.cuda.synchronize()
        L.spx_set_concurrent(0 if args.serial else 1)
    for k in range(args.warmup):
        step(k)
    barrier()
    L.spx_set_timing(1)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    L.spx_set_timing(0)
    # what the library did with the last timed call: 2 = pipelined with its predecessor, 1 = its three kernels side by side, 0 = in
    # sequence (ranks sharing a GPU, --serial ...)
    last_mode = int(L.spx_debug_last_call_concurrent())
    sa, sw, nc = C.c_double(0), C.c_double(0), C.c_int(0)
    L.spx_timing_collect(C.byref(sa), C.byref(sw), C.byref(nc))
    ms_tension = float(L.spx_timing_last_tension_ms()) / max(1, nc.value)
    dt = max_over_ranks(dt)
    n_in = n * STREAMS_PER_GPU
    walk_form = int(L.spx_debug_last_walk_form())
    outs = b.results()
    if pipe is not None:   # every buffer set of the pipeline holds the plain call's bytes
        for t in tickets[-pipe.depth:]:
            outs2 = pipe.results(t)
            assert len(outs2) == len(outs) and all(np.array_equal(x, y) for x, y in zip(outs, outs2)), "the batches of the pipelined loop differ"
            del outs2
        pipe.close()   # (its streams and buffers go before the other legs run)
    n_out = int(sum(o.size for o in outs))
    chain_steps = b.step_counts()         # pitch searches per stream: the length of every stream's dependent chain
    if dt_single is not None:
        dt_single = max_over_ranks(dt_single)
    if args.crc_out:
        with open("%s.rank%d.json" % (args.crc_out, rank), "w") as f:
            json.dump([zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs], f)

    # the analysis kernel alone (its fp64 vector roofline), rank 0's device
    ms_analysis_alone = standalone_analysis_ms(plan, b) if rank == 0 else None

    # the throughput regime: 2 048 streams of the headline kind in ONE call (2 048 distinct signals), rank 0's device
    large = None
    if not args.no_large_batch and rank == 0:
        nl = 8 * STREAMS_PER_GPU
        bl = Batch(plan, [n] * nl, 1, SPEED, 1.0, 0.0)
        # 2 048 DISTINCT signals (seeds 1234 + 0 .. 2047; the first 256 are the bench's own): eight copies of 256 streams would make
        # all eight copies of a chain end together, which flatters the walk kernel's tail
        bl.d_in[: n * STREAMS_PER_GPU].copy_(b.d_in[: n * STREAMS_PER_GPU])
        for blk in range(1, 8):
            extra = make_streams(STREAMS_PER_GPU, n, blk)   # seeds blk * 256 + i
            bl.d_in[blk * n * STREAMS_PER_GPU:(blk + 1) * n * STREAMS_PER_GPU].copy_(torch.from_numpy(np.concatenate(extra)))
            del extra
        dtl = time_window(bl.run, reps=5, warm=3)
        nout_l = bl.d_nout.cpu().numpy()
        assert (nout_l > 0).all()
        algo_l = 2 * (n * nl + int(nout_l.sum()))
        ka, kt, kw = L.spx_batch_kernel_names(plan.h, nl, 1, 1).decode().split(";")
        large = {"streams": nl, "ms_per_step": dtl * 1e3, "value": n * nl / dtl / 1e6, "unit": "Msamples/s",
                 "hbm": {"achieved": algo_l / dtl / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo_l / dtl / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": algo_l, "of": "the whole call (analysis, tension and walk kernels of two pipelined time chunks)"},
                 "kernels": {"analysis": ka, "tension": kt, "walk": kw},
                 "note": "BASELINE configs[3]'s kind at eight times its batch: %d x %d s, 16 kHz mono, 3.5x nonlinear, ONE spx_batch_run "
                         "per step, %d distinct signals (seeds 1234 + 0 .. %d); the walk kernel in its throughput form" % (nl, SECONDS, nl, nl - 1)}
        # ... and call after call through the owning pipeline object (round 6: a large pipelined call starts its producers at once --
        # the next call's first analysis chunk runs beside this call's last walk chunk; three buffer sets, outputs left on the device)
        try:
            from speedy_amd.batch import Pipeline
            pl = Pipeline(plan, [n] * nl, 1, SPEED, 1.0, 0.0, depth=3, device_out=True)
            tl = [pl.submit(bl.d_in) for _ in range(4)]
            torch.cuda.synchronize()
            t0l = time.perf_counter()
            tl += [pl.submit(bl.d_in) for _ in range(10)]
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0l) / 10
            _, _, c_ptr = pl.wait(tl[-1])
            cnt_p = torch.empty(nl, dtype=torch.int64)
            pl.L.spx_copy_to_host(cnt_p.data_ptr(), c_ptr, nl * 8, None)
            pl.L.spx_stream_synchronize(None)
            same = bool(np.array_equal(cnt_p.numpy(), np.asarray(nout_l)))
            large["pipelined"] = {"ms_per_step": dtp * 1e3, "value": n * nl / dtp / 1e6, "unit": "Msamples/s", "buffer_sets": 3,
                                  "hbm_frac": algo_l / dtp / 1e9 / HBM_PEAK_GBS, "produced_counts_equal_the_plain_calls": same,
                                  "note": "spx_pipeline_submit x 10 after 4 warm-up submits, input resident, outputs on the device"}
            pl.close()
            del pl
        except Exception as e:  # noqa: BLE001
            large["pipelined"] = {"error": repr(e)[:200]}
        del bl
        torch.cuda.empty_cache()

    # widening row: the speed-up kernels at 44.1 kHz mono and 48 kHz stereo (rank 0, N = 1 only)
    other = None
    rate_checks = {}
    if not args.no_other_rates and world == 1:
        other = {}
        for rate_o, ch_o in ((44100, 1), (48000, 2)):
            name = "%d_hz_%s" % (rate_o, "mono" if ch_o == 1 else "%dch" % ch_o)
            other[name], rate_checks[name] = other_rate_leg(rate_o, ch_o)

    # PCIe-inclusive steady state on every rank, MAX over ranks (reported beside `value`, which by the bench contract is
    # the rate with inputs already resident in HBM)
    pcie = None
    if not args.no_pcie:
        barrier()
        dt1, total = pcie_early["dt"], pcie_early["total"]
        assert total == n_out, (total, n_out)
        # what arrived in host memory is what the resident steps produce, stream by stream
        assert pcie_early["crcs"] == [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs], "PCIe leg: outputs differ from the resident steps'"
        dt1 = max_over_ranks(dt1)
        in_bytes, out_bytes = 2 * n_in, 2 * n_out
        pcie = {"value": n_in * world / dt1 / 1e6, "unit": "Msamples/s", "ms_per_step": dt1 * 1e3,
                "vs_resident_step": dt1 / (dt / args.steps),
                "batches_timed": pcie_early["reps"], "batches_warmup": pcie_early["warm"], "buffer_sets": pcie_early["depth"],
                "completion_interval_ms": pcie_early["interval_ms"],
                # the fourth roofline of the line: the host-to-device link (the input is 3x the output: H2D is the busier direction)
                "link": {"bound": "pcie_h2d", "h2d_gbs": in_bytes / dt1 / 1e9, "d2h_gbs": out_bytes / dt1 / 1e9,
                         "link_peak_gbs": PCIE_LINK_PEAK_GBS, "frac": in_bytes / dt1 / 1e9 / PCIE_LINK_PEAK_GBS,
                         "h2d_alone_gbs": in_bytes / pcie_early["h2d_alone_s"] / 1e9,
                         "h2d_alone_ms": pcie_early["h2d_alone_s"] * 1e3,
                         "frac_of_h2d_alone": pcie_early["h2d_alone_s"] / dt1,
                         "peak_source": "PCIe 5.0 x16, one direction: 32 GT/s x 16 lanes x 128/130 / 8 = 63 GB/s; h2d_alone = the same "
                                        "82 MB copied host -> device back to back with nothing else running (the floor of this leg)"},
                "note": "every rank, MAX over ranks: pinned host int16 input -> spx_pipeline_submit -> spx_pipeline_wait -> the produced "
                        "int16 frames of all streams in pinned host memory (include/speedy_hip.h: the library's owning pipeline object "
                        "issues the H2D copy, the overlapped batch call and the gather into host memory itself; %d buffer sets); the "
                        "steady phase: the interval between the completions of batch %d and batch %d, no pause before the timed window; "
                        "the leg runs first, in a child process, before this process touches the GPU; every stream's output CRC equals "
                        "the resident steps'" % (pcie_early["depth"], pcie_early["warm"], pcie_early["warm"] + pcie_early["reps"])}

    # BASELINE configs[4].  (1) WEAK: every rank one GPU's shard of 256 mixed streams (global streams 256 r .. 256 r + 255) in
    # one call.  (2) STRONG: the fixed batch of --total-streams streams, rank r of N its contiguous block, one call per rank.
    c4 = c4f = None
    c4_checks = {}
    c4_crc_dump = {}
    if not args.no_config4:
        from speedy_amd import config4 as C4
        barrier()
        ids = list(range(256 * rank, 256 * (rank + 1)))
        leg = config4_leg(ids, reps=10, pipeline=not args.no_pipeline)
        dt4 = max_over_ranks(leg["dt"])
        frames4 = sum_over_ranks(leg["frames"])
        c4 = {"value": frames4 / dt4 / 1e6, "unit": "Msamples/s", "ms_per_step": dt4 * 1e3, "streams_per_gpu": 256,
              "hbm_frac": sum_over_ranks(leg["algo_bytes"]) / dt4 / 1e9 / (HBM_PEAK_GBS * world),
              "chain_steps_max": int(leg["steps"].max()),
              "pipelined": leg["pipelined"], "timed_through": leg["through"],
              "unpipelined_ms_per_step": max_over_ranks(leg["dt_plain"]) * 1e3,
              "two_batches_ahead_ms_per_step": max_over_ranks(leg["dt_pair"]) * 1e3 if leg["dt_pair"] is not None else None,
              "pipeline_object_ms_per_step": max_over_ranks(leg["dt_pipe"]) * 1e3 if leg["dt_pipe"] is not None else None,
              "note": "BASELINE configs[4], WEAK: every rank its shard of 256 streams x 10 s (global stream i: 16 kHz if i even else "
                      "22.05 kHz; mono if (i/2) even else stereo; speed 1.5 if (i/4) even else 3.5; nonlinear 1; 2 048 distinct signals, "
                      "seed 4000 + i), ONE mixed-rate batch per step, inputs resident in HBM, MAX over ranks; input sample frames of all "
                      "ranks / that time.  Pipelined like `value`, one window of 40 steps through the owning pipeline object created with "
                      "spx_pipeline_create_mixed (four buffer sets, outputs left on the device; a step's producers beside the previous "
                      "steps' walk kernels, the walk kernels of consecutive steps overlapping); `two_batches_ahead_ms_per_step`: two "
                      "batches taking turns under spx_batch_run_mixed_ahead (rounds 4 - 5's figure); every buffer set's output CRCs equal "
                      "the plain call's.  `unpipelined_ms_per_step` is spx_batch_run_mixed on one batch, call after call (10 steps)"}
        c4_checks["config4_shard"] = (leg["streams"], ids, leg["crcs"])
        c4_crc_dump.update({str(i): c for i, c in zip(ids, leg["crcs"])})
        if not args.no_config4_full:
            total = args.total_streams
            barrier()
            ids_f = C4.rank_ids(rank, world, total)
            if ids_f == ids:       # (N = 8: a rank's block of the full batch IS its shard)
                legf = leg
            else:
                del leg
                legf = config4_leg(ids_f, reps=6, warm=4, pipeline=not args.no_pipeline)
            dtf = max_over_ranks(legf["dt"])
            frames_f = sum_over_ranks(legf["frames"])
            c4f = {"value": frames_f / dtf / 1e6, "unit": "Msamples/s", "ms_per_step": dtf * 1e3, "total_streams": total,
                   "streams_per_gpu": len(ids_f), "scaling": "strong", "pipelined": legf["pipelined"],
                   "unpipelined_ms_per_step": max_over_ranks(legf["dt_plain"]) * 1e3,
                   "hbm_frac": sum_over_ranks(legf["algo_bytes"]) / dtf / 1e9 / (HBM_PEAK_GBS * world),
                   "note": "BASELINE configs[4] as a whole, STRONG scaling: the fixed batch of %d mixed streams x 10 s (same global "
                           "sequence as config4_shard), rank r of N takes the contiguous block r (stream i -> GPU i / (%d / N)), ONE "
                           "spx_batch_run_mixed call per rank and step, inputs resident in HBM, MAX over ranks; N = 1 runs the whole "
                           "batch in one call (two groups of %d streams: throughput-form walk kernels, pipelined time chunks)"
                           % (total, total, total // 2)}
            c4_checks["config4_full"] = (legf["streams"], ids_f, legf["crcs"])
            c4_crc_dump.update({str(i): c for i, c in zip(ids_f, legf["crcs"])})
        if args.crc_out:
            with open("%s.c4.rank%d.json" % (args.crc_out, rank), "w") as f:
                json.dump(c4_crc_dump, f)

    if rank == 0:
        total_in = n_in * world * args.steps
        ms_step = dt / args.steps * 1e3
        ms_analyze = sa.value / max(1, nc.value)
        ms_walk = sw.value / max(1, nc.value)
        algo_bytes = 2 * 1 * (n_in + n_out)  # int16 read once + int16 written once, per launch
        # the kernels that served the batch, by the names a profiler prints (template arguments included)
        # (the pipelined loop on three workspaces launches the walk kernel in its lean form)
        names_fn = L.spx_batch_kernel_names_lean if (pipe is not None and last_mode == 2 and walk_form == 16 * 4) else L.spx_batch_kernel_names
        k_analysis, k_tension, k_walk = names_fn(plan.h, STREAMS_PER_GPU, 1, 1).decode().split(";")
        dom, dom_ms = (k_walk, ms_walk) if ms_walk >= ms_analyze else (k_analysis, ms_analyze)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        traffic_note = None
        pt = load_json("pmc_traffic.json")
        if pt:
            traffic = pt.get(dom, {}).get("hbm_bytes_per_launch")
            # (how THIS kernel's counters were taken: the lean walk form inside the pipelined loop, where the profiler may serialise
            # the launches -- fine for byte counts -- or the serial-mode passes of the other kernels)
            traffic_note = pt.get(dom, {}).get("source") or pt.get("_note")
        # ---- the two rooflines that actually bound the path (SURVEY 8d "report all three") ----
        clock_mhz = float(getattr(torch.cuda.get_device_properties(dev_index), "clock_rate", 2400000)) / 1e3   # nominal shader clock (2 400 MHz)
        lm = load_json("latency_model.json")
        fm = load_json("flop_model.json")
        smax, smean = int(chain_steps.max()), float(chain_steps.mean())
        latency = None
        if lm and smax > 0 and ms_walk > 0:
            cyc = ms_walk * 1e-3 * clock_mhz * 1e6 / smax
            latency = {"bound": "latency of the per-stream chain of dependent pitch steps (one stream per CU)", "kernel": k_walk,
                       "steps_per_stream": {"max": smax, "mean": smean, "min": int(chain_steps.min()),
                                            "source": "counted by the walk kernel itself (spx_batch_read_steps)"},
                       "clock_mhz": clock_mhz,
                       "achieved_cycles_per_step": cyc, "model_floor_cycles_per_step": lm["floor_cycles_per_step"],
                       "frac": lm["floor_cycles_per_step"] / cyc,
                       "definition": "achieved = walk kernel ms x nominal clock / steps of the LONGEST chain (the kernel ends with its "
                                     "slowest stream; its first ~85 us wait for speeds are inside); floor = tools/latency_model.py "
                                     "(profiles/latency_model.json: dependent latencies of one step priced with "
                                     "tools/ubench/issue_costs.hip); frac = floor / achieved"}
        valu = None
        if fm and ms_analysis_alone:
            frames = int(sum(b.frames))
            flop = fm[str(RATE)]["flop_per_frame"]
            tf = flop * frames / (ms_analysis_alone * 1e-3) / 1e12
            peak = FP64_VALU_PEAK_TFLOPS
            valu = {"bound": "fp64 vector ALU", "kernel": k_analysis, "flop_per_frame": flop, "frames_per_launch": frames,
                    "standalone_ms": ms_analysis_alone, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                    "fma_per_frame": fm[str(RATE)].get("of_which_fma"),
                    "peak_source": "MI355X fp64 vector INSTRUCTION rate: 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz = 39.3 T operations/s "
                                   "(the 78.6 TFLOP/s FMA figure counts every fma twice; here an fma is ONE operation, as are a division and a "
                                   "square root).  Since round 5 the DFT and log specs fuse their multiply-add pairs explicitly (DESIGN.md 4, 4a: "
                                   "12 646 operations per frame at 16 kHz, 5 033 of them fma; 19 355 unfused in rounds 1-4), so `achieved` fell "
                                   "with the operation count while the kernel got faster",
                    "definition": "fp64 operations of the DFT and log specs per frame (tools/flop_count.py, profiles/flop_model.json) x frames / "
                                  "the analysis kernel ALONE (spx_batch_analyze, one window of 10 launches, HIP events)"}
        # The roofline that BINDS the pipelined loop (round-4 review): every wave64 VALU instruction occupies its SIMD for 4 cycles,
        # so the three kernels' VALU wave-instructions per batch (SQ_INSTS_VALU, profiles/sq_counters.json, tools/sq_counters.sh)
        # x 4 / (SIMDs x clock) is the least time a batch can take whatever overlaps with whatever.
        valu_issue = None
        sq = load_json("sq_counters.json")
        if sq:
            cnt = sq.get("counters", {})
            per = {k: cnt.get(k, {}).get("SQ_INSTS_VALU") for k in (k_analysis, k_tension, k_walk)}
            if all(v is not None for v in per.values()):
                cus = int(torch.cuda.get_device_properties(dev_index).multi_processor_count)
                simds = 4 * cus
                floor_ms = sum(per.values()) * 4.0 / simds / (clock_mhz * 1e6) * 1e3
                valu_issue = {"bound": "VALU issue (wave64 instructions x 4 cycles per SIMD)", "valu_wave_insts_per_batch": per,
                              "simds": simds, "clock_mhz": clock_mhz, "floor_ms_per_step": floor_ms, "ms_per_step": ms_step,
                              "frac": floor_ms / ms_step, "hbm_frac_at_floor": algo_bytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "source": "profiles/sq_counters.json (" + str(sq.get("note", ""))[:160] + ")",
                              "definition": "sum of SQ_INSTS_VALU of the step's three kernels x 4 cycles / (SIMDs x clock) / ms_per_step: the "
                                            "share of the step during which every SIMD of the chip would have to issue a vector instruction; "
                                            "hbm_frac_at_floor = the HBM fraction the path would reach AT this bound"}
        line = {
            "metric": "Msamples/s processed (16 kHz mono, 3.5x nonlinear)",
            "value": total_in / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "value_definition": "input sample frames of all ranks x steps / MAX-over-ranks wall time of the timed steps, inputs "
                                "resident in HBM when the timed region starts (the bench contract's definition); SURVEY 8(d)'s "
                                "first-write-to-last-drained-read rate is `pcie_inclusive`, the drop-in API's rate with 256 "
                                "live sonicStream handles is `api_256_handles`.  ONE rule for every figure in this line: a single "
                                "timed window of consecutive steps after untimed warm-up steps -- no best-of-N anywhere.  "
                                + ("The timed steps go through the library's owning pipeline object (spx_pipeline: submit after submit "
                                   "of the resident input, outputs left in device memory; config.buffer_sets buffer sets owned by the "
                                   "library): step k+1's analysis and tension kernels run beside step k's walk kernel and the walk "
                                   "kernels of consecutive steps overlap (so `roofline.kernel_avg_launch_ms` of the walk kernel is LONGER "
                                   "than `ms_per_step`); every step is the whole hot path over one batch.  config.unpipelined_ms_per_step "
                                   "is spx_batch_run on one batch, call after call (what rounds 1-4 reported as `value`)" if pipe is not None else
                                   "spx_batch_run on one batch, call after call (--no-pipeline)"),
            "pipelined": pipe is not None and last_mode == 2,
            "launch_mode_of_the_timed_calls": {2: "pipelined with the previous call (spx_batch_run_overlapped)", 1: "three kernels side by side",
                                               0: "kernels in sequence"}.get(last_mode, str(last_mode)),
            "walk_form": {68: "4 search + 4 output waves", 64: "lean: 4 search waves that also do the output work", 32: "throughput form"}.get(walk_form, str(walk_form)),
            "unpipelined": None if dt_single is None else
                           {"ms_per_step": dt_single * 1e3, "value": n_in * world / dt_single / 1e6, "unit": "Msamples/s",
                            "note": "spx_batch_run on ONE batch, call after call (its three kernels side by side, the walk "
                                    "waiting for its first speeds at the start of every call); MAX over ranks"},
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16 samples; f64 DFT, f32 features, int32 AMDF/OLA",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d streams/GPU x %d s, 16 kHz mono int16, speed 3.5, "
                                   "nonlinear 1.0, feedback 0" % (STREAMS_PER_GPU, SECONDS),
                       "streams_per_gpu": STREAMS_PER_GPU, "samples_per_stream": n,
                       # how `value` was produced, where a parsed record finds it (round-4 review): the like-for-like figure of rounds
                       # 1-4, the buffer sets the library owns for the pipelined steps, the walk kernels in flight at a time
                       "unpipelined_ms_per_step": None if dt_single is None else dt_single * 1e3,
                       "buffer_sets": (pipe_depth if pipe is not None else 1),
                       "walk_launches_in_flight": (2 if (pipe is not None and last_mode == 2) else 1),
                       "timed_through": ("spx_pipeline_submit (device-resident input, SPX_PIPELINE_DEVICE_OUT)" if pipe is not None else "spx_batch_run"),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       # what tests/test_gpu_perf_guard.py did on this box, if it ran here (guarded: false = the box was too noisy to judge)
                       "perf_guard": load_scratch_json("perf_guard.json"),
                       "parallelism": "streams sharded %d/GPU, no data-path collective" % STREAMS_PER_GPU,
                       "launcher": "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else
                                   ("bench.py --gpus (self-spawned ranks)" if world > 1 else "single process"),
                       "backend": (backend if world > 1 else None),
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None),
                       "n_ranks_seen": (int(layout.shape[0]) if world > 1 else 1),
                       "handshake_ms": handshake_ms,
                       "ranks": rank_info if rank_info is not None else
                                [{"rank": 0, "device_id": dev_index, "device": torch.cuda.get_device_name(dev_index),
                                  "numa_node": numa["numa_node"], "cpus_allowed": numa["cpus_allowed"], "bound_to_numa_node": numa["bound"],
                                  "pci": numa["pci"],
                                  "h2d_alone_gbs": (2 * n_in / pcie_early["h2d_alone_s"] / 1e9) if pcie_early else None}],
                       "stream_seeds": "1234 + global stream index (rank * %d + i), all distinct" % STREAMS_PER_GPU,
                       "realtime_factor_per_stream": SECONDS / (ms_step * 1e-3),
                       "out_samples_per_gpu": n_out, "pipeline_chunks": args.chunks,
                       "kernel_launches_per_step": {k_analysis: args.chunks, k_tension: args.chunks, k_walk: args.chunks}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # two launches of the dominant kernel are in flight at a time in the pipelined loop, so a launch lasts longer
                         # than a step: `achieved` / `frac` above are per LAUNCH (the contract's definition, what rocprofv3's average
                         # duration reproduces); these two are the same bytes over the time between two steps
                         "achieved_per_step_period": algo_bytes / (ms_step * 1e-3) / 1e9,
                         "frac_per_step_period": algo_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_per_step": {k_analysis: ms_analyze, k_tension: ms_tension, k_walk: ms_walk},
                         "kernel_avg_launch_ms": {k_analysis: ms_analyze / args.chunks, k_tension: ms_tension / args.chunks,
                                                  k_walk: ms_walk / args.chunks},
                         "limiter": "latency, not HBM: 256 per-stream chains of ~1300 dependent pitch steps, one "
                                    "workgroup per CU (DESIGN.md 5.3, 6); `bound` names the roofline the contract asks "
                                    "to be priced against; `latency` and `valu_fp64` are the rooflines that bound the two "
                                    "big kernels, `large_batch.hbm` the HBM fraction where the path is throughput-bound",
                         "latency": latency, "valu_fp64": valu, "valu_issue": valu_issue},
        }
        if large is not None:
            line["large_batch"] = large
        if other:
            other["note"] = ("beyond BASELINE's rates: 256 streams x %d s, speed %.1f nonlinear, ONE spx_batch_run per step, inputs "
                             "resident in HBM, input sample frames / time; 8 distinct signals per kind (seed 7000 + i)" % (SECONDS, SPEED))
            line["other_rates"] = other
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if c4 is not None:
            line["config4_shard"] = c4
        if c4f is not None:
            line["config4_full"] = c4f
        if not args.no_api and world == 1:
            line["api_256_handles"] = api_many_handles()
            # the reference's OWN call order: one thread, then 16 threads x 16 handles, then a thread per handle
            line["api_percall_256"] = api_many_handles(seconds=4.0, order="percall")
            line["api_threads"] = api_many_handles(seconds=16.0, order="threads:16")
            line["api_threads"]["more_threads"] = {
                "64x4": {k: v for k, v in api_many_handles(seconds=16.0, order="threads:64").items() if k in ("value", "handles_per_launch_sequence", "x_realtime_per_stream", "error")},
                "256x1": {k: v for k, v in api_many_handles(seconds=16.0, order="threads:256").items() if k in ("value", "handles_per_launch_sequence", "x_realtime_per_stream", "error")}}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (bench contract)
            line["cpu_baseline"] = cpu_baseline(streams, outs, c4_checks=c4_checks, rate_checks=rate_checks)
            line["cpu_baseline_fftw"] = line["cpu_baseline"].pop("cpu_baseline_fftw", {"fftw": "absent"})
            for name, r in line["cpu_baseline"].get("other_rates_crc_check", {}).items():
                line["other_rates"][name]["output_crc_mismatches_vs_cpu_port"] = r["output_crc_mismatches_vs_gpu"]
                line["other_rates"][name]["streams_checked_against_cpu_port"] = r["streams_checked"]
            for name, r in line["cpu_baseline"].get("config4_crc_check", {}).items():
                if name in line:
                    line[name]["output_crc_mismatches_vs_cpu_port"] = r["output_crc_mismatches_vs_gpu"]
                    line[name]["streams_checked_against_cpu_port"] = r["streams_checked"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
