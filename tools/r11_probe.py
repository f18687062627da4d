"""Does the 22.05 kHz analysis kernel's transform equal the oracle's at the fp64 level?  A float magnitude shows an fp64 last-bit
difference about once in 2^29 values, so: MANY frames of noise through the GPU's spectrogram tap, one 64-bit hash per frame.
  python3 tools/r11_probe.py run OUT.npy BATCHES          (SPEEDY_HIP_LIB selects the library build)
  python3 tools/r11_probe.py compare A.npy B.npy          every frame whose hash differs between two builds, and which of the two the
                                                          ORACLE's spectrogram of that frame agrees with
  python3 tools/r11_probe.py oracle BATCHES               the library against the ORACLE ITSELF, every frame: the CPU port with hashing
                                                          callbacks (oracle/orc_bench.c orc_bench_run_hashed, all host cores) -- spectrogram
                                                          rows and features / tension / speed, frame for frame
  python3 tools/r11_probe.py audio BATCHES | pipeline BATCHES    the OUTPUT AUDIO of every stream against the CPU port's (CRC-32), plain
                                                          calls with the settings cycled | through the owning pipeline object
(round 5: the hand-written radix-11 stage had kept the unfused sums of DFT spec v1; profiles/r05/r5ah_r11_probe.txt)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RATE, NS, SECS = int(os.environ.get("SPX_PROBE_RATE", "22050")), 256, 1   # (SPX_PROBE_RATE: any rate; two builds / two settings of one build)


def stream(seed, i):
    return np.random.default_rng([seed, i]).integers(-20000, 20000, size=RATE * SECS).astype(np.int16)


def hashes(rows):   # rows: (frames, N) float32 -> int64 per frame
    return rows.view(np.int32).astype(np.int64).sum(axis=1)


def against_the_oracle(rate, batches, verbose=True, channels=1, match_matlab=False, feedback=0.0, speed=3.0):
    """(spectrogram rows that differ, rows compared, tension frames that differ, tension frames compared) -- the library against the CPU port."""
    global RATE
    RATE = rate
    import ctypes as C
    import subprocess
    import torch
    from speedy_amd.batch import Batch, Plan
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run_hashed.restype = C.c_double
    L.orc_bench_run_hashed.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    plan = Plan(RATE, match_matlab)
    n = RATE * SECS
    b = Batch(plan, [n] * NS, channels, speed, 1.0, feedback, taps=True, spectrogram_taps=True)
    T = int(b.frames[0])
    K = max(0, T - plan.F + 1)
    MAXF = T + 8
    threads = len(os.sched_getaffinity(0))
    bad_spec = bad_tap = 0
    w = None
    t_cpu = 0.0
    for seed in range(batches):
        xs = [np.random.default_rng([seed, i, channels]).integers(-20000, 20000, size=n * channels).astype(np.int16) if channels > 1
              else stream(seed, i) for i in range(NS)]
        b.upload(xs)
        b.run()
        torch.cuda.synchronize()
        hs = b.t_spec.view(torch.int32).view(-1, plan.N).to(torch.int64).sum(dim=1).cpu().numpy().reshape(NS, T)
        if w is None:
            w = torch.arange(1, 16, device=b.t_features.device, dtype=torch.int64)
        ht = ((b.t_features.view(torch.int32).view(-1, 15).to(torch.int64) * w).sum(dim=1) + 31 * b.t_tension.view(torch.int32).to(torch.int64)
              + 37 * b.t_speed.view(torch.int32).to(torch.int64)).cpu().numpy().reshape(NS, -1)
        buf = np.ascontiguousarray(np.concatenate(xs), np.int16)
        osp = np.zeros((NS, MAXF), np.int64)
        otp = np.zeros((NS, MAXF), np.int64)
        nsp = np.zeros(NS, np.int32)
        ntp = np.zeros(NS, np.int32)
        t_cpu += L.orc_bench_run_hashed(buf.ctypes.data, n, NS, RATE, channels, speed, 1.0, feedback, 1 if match_matlab else 0, 1000, threads, MAXF, osp.ctypes.data,
                                        otp.ctypes.data, nsp.ctypes.data, ntp.ctypes.data)
        assert int(nsp.min()) >= T and int(ntp.min()) >= K, (int(nsp.min()), T, int(ntp.min()), K)
        ds = np.argwhere(hs != osp[:, :T])
        dt = np.argwhere(ht[:, :K] != otp[:, :K])
        bad_spec += len(ds)
        bad_tap += len(dt)
        for i, f in ds[:5]:
            print("batch %d stream %d frame %d: spectrogram row differs from the oracle's" % (seed, i, f))
        for i, f in dt[:5]:
            print("batch %d stream %d tension frame %d: features / tension / speed differ from the oracle's" % (seed, i, f))
    if verbose:
        print("rate %d ch %d mm %d fb %.2f speed %.1f: %d batches x %d streams: %d of %d spectrogram rows and %d of %d tension frames differ from the ORACLE's (CPU port %.1f s on %d threads)"
              % (RATE, channels, int(match_matlab), feedback, speed, batches, NS, bad_spec, batches * NS * T, bad_tap, batches * NS * K, t_cpu, threads))
    return bad_spec, batches * NS * T, bad_tap, batches * NS * K


def audio_against_the_oracle(rate, batches, verbose=True):
    """(streams whose OUTPUT AUDIO differs from the CPU port's, streams compared): batch k takes the k-th of a cycle of (channels,
    speed, nonlinear, feedback) settings on both sides of 1; two-second noise streams of ragged lengths; CRC-32 per stream."""
    import ctypes as C
    import subprocess
    import zlib
    from speedy_amd.batch import Batch, Plan
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    settings = [(1, 3.5, 1.0, 0.0), (1, 2.0, 0.0, 0.0), (2, 1.5, 1.0, 0.1), (1, 1.2, 1.0, 0.0), (1, 0.7, 0.0, 0.0), (2, 3.0, 1.0, 0.0),
                (1, 0.4, 1.0, 0.0), (1, 5.5, 0.3, 0.2)]
    plan = Plan(rate, False)
    threads = len(os.sched_getaffinity(0))
    bad = total = 0
    for k in range(batches):
        ch, speed, nl, fb = settings[k % len(settings)]
        n = 2 * rate - 97 * (k % 11)
        xs = [np.random.default_rng([k, i, 7]).integers(-20000, 20000, size=n * ch).astype(np.int16) for i in range(NS)]
        b = Batch(plan, [n] * NS, ch, speed, nl, fb)
        b.upload(xs)
        b.run()
        outs = b.results()
        buf = np.ascontiguousarray(np.concatenate(xs), np.int16)
        frames = (C.c_long * NS)()
        crcs = (C.c_uint32 * NS)()
        L.orc_bench_run(buf.ctypes.data, n, NS, rate, ch, speed, nl, fb, 0, 1000, threads, frames, crcs)
        for i in range(NS):
            if zlib.crc32(np.ascontiguousarray(outs[i]).tobytes()) != crcs[i] or outs[i].size != frames[i] * ch:
                bad += 1
                if verbose and bad <= 5:
                    print("batch %d stream %d (ch %d speed %.1f nl %.1f fb %.1f): the output differs from the oracle's" % (k, i, ch, speed, nl, fb))
        total += NS
    if verbose:
        print("rate %d: the OUTPUT AUDIO of %d of %d streams differs from the ORACLE's (%d batches, settings cycled)" % (rate, bad, total, batches))
    return bad, total


def pipeline_audio_against_the_oracle(rate, batches, verbose=True, seconds=2):
    """The same comparison through the OWNING PIPELINE OBJECT the bench's headline runs on (spx_pipeline: device-resident input, outputs
    left on the device, four buffer sets, the walk kernels of consecutive batches overlapping, each batch's producers beside the
    walk kernels before it): one pipeline per setting, `batches` batches through each with three in flight; every stream's CRC-32
    against the CPU port's.  (streams that differ, streams compared)"""
    import ctypes as C
    import subprocess
    import zlib
    import torch
    from speedy_amd.batch import Pipeline, Plan
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    settings = [(1, 3.5, 1.0, 0.0), (1, 2.0, 0.0, 0.0), (2, 1.5, 1.0, 0.1), (1, 1.2, 1.0, 0.0), (2, 3.0, 1.0, 0.0), (1, 5.5, 0.3, 0.2),
                (1, 0.7, 0.0, 0.0)]
    plan = Plan(rate, False)
    threads = len(os.sched_getaffinity(0))
    bad = total = 0
    for s, (ch, speed, nl, fb) in enumerate(settings):
        n = seconds * rate - 131 * s
        pipe = Pipeline(plan, [n] * NS, ch, speed, nl, fb, depth=4, device_out=True)
        flight = []

        def check(k, t, buf):
            nonlocal bad, total
            outs = pipe.results(t)
            frames = (C.c_long * NS)()
            crcs = (C.c_uint32 * NS)()
            L.orc_bench_run(buf.ctypes.data, n, NS, rate, ch, speed, nl, fb, 0, 1000, threads, frames, crcs)
            for i in range(NS):
                if zlib.crc32(np.ascontiguousarray(outs[i]).tobytes()) != crcs[i] or outs[i].size != frames[i] * ch:
                    bad += 1
                    if verbose and bad <= 5:
                        print("setting %d batch %d stream %d (ch %d speed %.1f nl %.1f fb %.1f): the output differs from the oracle's"
                              % (s, k, i, ch, speed, nl, fb))
            total += NS

        for k in range(batches):
            buf = np.random.default_rng([s, k, 11]).integers(-20000, 20000, size=NS * n * ch).astype(np.int16)
            dev = torch.zeros(buf.size + 64, dtype=torch.int16, device="cuda")   # (+ 64: include/speedy_hip.h spx_pipeline_submit, device input)
            dev[: buf.size].copy_(torch.from_numpy(buf))
            flight.append((k, pipe.submit(dev), buf, dev))
            if len(flight) > 3:
                k0, t0, b0, _ = flight.pop(0)
                check(k0, t0, b0)
        for k0, t0, b0, _ in flight:
            check(k0, t0, b0)
        pipe.close()
    if verbose:
        print("rate %d: through the pipeline object, the OUTPUT AUDIO of %d of %d streams differs from the ORACLE's (%d settings x %d batches)"
              % (rate, bad, total, len(settings), batches))
    return bad, total


def mixed_audio_against_the_oracle(batches, verbose=True, seconds=2):
    """BASELINE configs[4]'s shard shape -- 256 streams, stream i at 16 kHz if i is even else 22.05 kHz, mono if (i / 2) is even else
    stereo, speed 1.5 if (i / 4) is even else 3.5, nonlinear -- as ONE spx_batch_run_mixed call per batch: noise streams of ragged
    lengths, even batches as plain calls, odd ones pipelined (run_ahead, two batch objects taking turns, the way bench.py's
    config4 leg is timed); every stream's CRC-32 against the CPU port's, kind by kind.  (streams that differ, streams compared)"""
    import ctypes as C
    import subprocess
    import zlib
    from speedy_amd import config4 as C4
    from speedy_amd.batch import MixedBatch, Plan
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    plans = [Plan(r, False) for r in C4.RATES]
    threads = len(os.sched_getaffinity(0))
    ids = list(range(NS))
    bad = total = 0
    turn = {}
    pending = []

    def check(k, b, xs, lens):
        nonlocal bad, total
        outs = b.results()
        for kind in range(8):
            members = [i for i in ids if C4.kind(i) == kind]
            rate, ch, speed = C4.cfg(kind)
            n = lens[kind]
            buf = np.ascontiguousarray(np.concatenate([xs[i] for i in members]), np.int16)
            frames = (C.c_long * len(members))()
            crcs = (C.c_uint32 * len(members))()
            L.orc_bench_run(buf.ctypes.data, n, len(members), rate, ch, speed, 1.0, 0.0, 0, 1000, threads, frames, crcs)
            for j, i in enumerate(members):
                if zlib.crc32(np.ascontiguousarray(outs[i]).tobytes()) != crcs[j] or outs[i].size != frames[j] * ch:
                    bad += 1
                    if verbose and bad <= 5:
                        print("batch %d stream %d (rate %d ch %d speed %.1f): the output differs from the oracle's" % (k, i, rate, ch, speed))
            total += len(members)

    for k in range(batches):
        lens = [seconds * C4.cfg(kind)[0] - 89 * ((k + kind) % 7) for kind in range(8)]   # per kind: the oracle call takes one length
        xs = [np.random.default_rng([k, i, 13]).integers(-20000, 20000, size=lens[C4.kind(i)] * C4.cfg(i)[1]).astype(np.int16) for i in ids]
        key = (tuple(lens), k % 4)
        b = turn.get(key)
        if b is None:
            b = MixedBatch(plans, [C4.RATES.index(C4.cfg(i)[0]) for i in ids], [lens[C4.kind(i)] for i in ids],
                           [C4.cfg(i)[1] for i in ids], [C4.cfg(i)[2] for i in ids], 1.0, 0.0)
            if len(turn) >= 8:
                turn.pop(next(iter(turn)))
            turn[key] = b
        b.upload(xs)
        if k % 2 == 0:
            b.run()
        else:
            b.run_ahead()
        pending.append((k, b, xs, lens))
        if len(pending) > 1:          # the previous batch is checked while this one runs
            check(*pending.pop(0))
    for q in pending:
        check(*q)
    if verbose:
        print("configs[4] shard shape: the OUTPUT AUDIO of %d of %d streams differs from the ORACLE's (%d mixed batches, plain and pipelined calls taking turns)"
              % (bad, total, batches))
    return bad, total


def mixed_pipeline_audio_against_the_oracle(batches, verbose=True, seconds=2):
    """The same shard shape through the OWNING PIPELINE OBJECT created with spx_pipeline_create_mixed, outputs left on the device (round 6:
    the groups' walk kernels of consecutive batches overlap on the library's walk streams): one pipeline of four buffer sets per
    length setting, noise streams, every batch different; every stream's CRC-32 against the CPU port's, kind by kind."""
    import ctypes as C
    import subprocess
    import zlib
    import torch
    from speedy_amd import config4 as C4
    from speedy_amd.batch import Pipeline, Plan
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_bench.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "liborc_bench.so"))
    L.orc_bench_run.restype = C.c_double
    L.orc_bench_run.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    plans = [Plan(r, False) for r in C4.RATES]
    threads = len(os.sched_getaffinity(0))
    ids = list(range(NS))
    bad = total = 0
    pipes = {}
    pending = []

    def check(k, pipe, t, xs, lens):
        nonlocal bad, total
        outs = pipe.results(t)
        for kind in range(8):
            members = [i for i in ids if C4.kind(i) == kind]
            rate, ch, speed = C4.cfg(kind)
            buf = np.ascontiguousarray(np.concatenate([xs[i] for i in members]), np.int16)
            frames = (C.c_long * len(members))()
            crcs = (C.c_uint32 * len(members))()
            L.orc_bench_run(buf.ctypes.data, lens[kind], len(members), rate, ch, speed, 1.0, 0.0, 0, 1000, threads, frames, crcs)
            for j, i in enumerate(members):
                if zlib.crc32(np.ascontiguousarray(outs[i]).tobytes()) != crcs[j] or outs[i].size != frames[j] * ch:
                    bad += 1
                    if verbose and bad <= 5:
                        print("batch %d stream %d (rate %d ch %d speed %.1f): the output differs from the oracle's" % (k, i, rate, ch, speed))
            total += len(members)

    for k in range(batches):
        lens = [seconds * C4.cfg(kind)[0] - 89 * ((k // 3 + kind) % 7) for kind in range(8)]   # three consecutive batches share a pipeline
        xs = [np.random.default_rng([k, i, 17]).integers(-20000, 20000, size=lens[C4.kind(i)] * C4.cfg(i)[1]).astype(np.int16) for i in ids]
        key = tuple(lens)
        ent = pipes.get(key)
        if ent is None:
            pipe = Pipeline(plans, [lens[C4.kind(i)] for i in ids], [C4.cfg(i)[1] for i in ids], [C4.cfg(i)[2] for i in ids], 1.0, 0.0,
                            depth=4, device_out=True, plan_index=[C4.RATES.index(C4.cfg(i)[0]) for i in ids])
            ent = pipes[key] = (pipe, [torch.zeros(pipe.total_in + 64, dtype=torch.int16, device="cuda") for _ in range(4)])
        pipe, bufs = ent
        d = bufs[k % 4]
        d[: pipe.total_in].copy_(torch.from_numpy(pipe.pack(xs)))
        torch.cuda.synchronize()
        t = pipe.submit(d)
        pending.append((k, pipe, t, xs, lens))
        if len(pending) > 2:          # two batches stay in flight while an earlier one is checked
            check(*pending.pop(0))
    for q in pending:
        check(*q)
    for pipe, _ in pipes.values():
        pipe.close()
    if verbose:
        print("configs[4] shard shape through the mixed PIPELINE OBJECT (outputs on the device, walk kernels of consecutive batches overlapping): "
              "the OUTPUT AUDIO of %d of %d streams differs from the ORACLE's (%d batches)" % (bad, total, batches))
    return bad, total


if __name__ != "__main__":
    pass
elif sys.argv[1] == "audio":
    audio_against_the_oracle(RATE, int(sys.argv[2]))
elif sys.argv[1] == "mixed":
    mixed_audio_against_the_oracle(int(sys.argv[2]))
elif sys.argv[1] == "mixedpipe":
    mixed_pipeline_audio_against_the_oracle(int(sys.argv[2]))
elif sys.argv[1] == "pipeline":
    pipeline_audio_against_the_oracle(RATE, int(sys.argv[2]))
elif sys.argv[1] == "oracle":   # [channels match_matlab feedback speed]
    against_the_oracle(RATE, int(sys.argv[2]), True, *([int(sys.argv[3]), bool(int(sys.argv[4])), float(sys.argv[5]), float(sys.argv[6])] if len(sys.argv) > 6 else []))
elif sys.argv[1] == "run":
    import torch
    from speedy_amd.batch import Batch, Plan
    out, batches = sys.argv[2], int(sys.argv[3])
    plan = Plan(RATE, False)
    b = Batch(plan, [RATE * SECS] * NS, 1, 3.0, 1.0, 0.0, taps=True, spectrogram_taps=True)
    res, res2, res3 = [], [], []
    for seed in range(batches):
        b.upload([stream(seed, i) for i in range(NS)])
        b.run()
        torch.cuda.synchronize()
        h = b.t_spec.view(torch.int32).view(-1, plan.N).to(torch.int64).sum(dim=1)
        res.append(h.cpu().numpy())
        # ... and the frame-rate stage's taps (15 features, tension, speed per tension frame): everything behind the magnitudes
        w = torch.arange(1, 16, device=b.t_features.device, dtype=torch.int64)
        hf = (b.t_features.view(torch.int32).view(-1, 15).to(torch.int64) * w).sum(dim=1) + 31 * b.t_tension.view(torch.int32).to(torch.int64) \
            + 37 * b.t_speed.view(torch.int32).to(torch.int64)
        res2.append(hf.cpu().numpy())
        # ... and the audio itself: one weighted sum of the whole output buffer and the produced counts per batch
        o = b.d_out.to(torch.int64)
        res3.append(int((o * (torch.arange(o.numel(), device=o.device, dtype=torch.int64) % 65521 + 1)).sum().item()) ^ int(b.d_nout.sum().item()))
    np.save(out, np.stack(res))
    np.save(out.replace(".npy", "_audio.npy"), np.array(res3, np.int64))
    np.save(out.replace(".npy", "_features.npy"), np.stack(res2))
    print("frames per batch", res[0].size, "batches", batches)
else:
    A, B = np.load(sys.argv[2]), np.load(sys.argv[3])
    assert A.shape == B.shape
    fa, fb = sys.argv[2].replace(".npy", "_features.npy"), sys.argv[3].replace(".npy", "_features.npy")
    if os.path.exists(fa) and os.path.exists(fb):
        FA, FB = np.load(fa), np.load(fb)
        print("rate %d: %d of %d tension frames differ in features / tension / speed" % (RATE, int((FA != FB).sum()), FA.size))
    aa, ab = sys.argv[2].replace(".npy", "_audio.npy"), sys.argv[3].replace(".npy", "_audio.npy")
    if os.path.exists(aa) and os.path.exists(ab):
        print("rate %d: %d of %d batches differ in the output audio" % (RATE, int((np.load(aa) != np.load(ab)).sum()), np.load(aa).size))
    diff = np.argwhere(A != B)
    print("rate %d: %d of %d frames differ between the two builds" % (RATE, len(diff), A.size))
    from oracle import pyorc as orc
    L = orc.lib()
    if os.environ.get("SPX_PROBE_ORACLE_SPEC"):   # the oracle on DFT spec 1 (the unfused transform of rounds 1 - 4) instead of 2
        L.orc_set_dft_spec(int(os.environ["SPX_PROBE_ORACLE_SPEC"]))
        print("oracle DFT spec", L.orc_get_dft_spec())
    T = A.shape[1] // NS
    agree = {"first": 0, "second": 0, "neither": 0}
    for seed, idx in diff[:40]:
        i, f = idx // T, idx % T
        x = stream(int(seed), int(i))
        rows = []
        h = L.orc_sonicCreateStream(RATE, 1, 0)
        nb = L.orc_sonicSpectrogramSize(h)
        cb = orc.FEATURES_FN(lambda s, t, p: rows.append(np.ctypeslib.as_array(p, shape=(nb,)).copy()))
        L.orc_sonicSpectrogramCallback(h, cb)
        L.orc_sonicSetSpeed(h, 3.0)
        L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
        L.orc_sonicWriteShortToStream(h, orc.sptr(x), x.size)
        L.orc_sonicDestroyStream(h)
        ho = int(hashes(np.array(rows[int(f)], np.float32)[None, :])[0])
        k = "first" if ho == A[seed, idx] else ("second" if ho == B[seed, idx] else "neither")
        agree[k] += 1
        print("batch %d stream %d frame %d: the oracle agrees with the %s build" % (seed, i, f, k))
        if os.environ.get("SPX_PROBE_DETAIL"):   # this process's library on that stream: which bins differ from the oracle's, by how much
            from speedy_amd.batch import Batch, Plan
            plan = Plan(RATE, False)
            bb = Batch(plan, [x.size], 1, 3.0, 1.0, 0.0, taps=True, spectrogram_taps=True)
            bb.upload([x])
            bb.run()
            g = bb.tap_arrays(0)["spectrogram"][int(f)]
            o = np.array(rows[int(f)], np.float32)
            for kk in np.nonzero(g != o)[0]:
                print("      bin %d: this library %.9g  oracle %.9g  (%d float ulps)" % (kk, g[kk], o[kk], int(g[kk:kk + 1].view(np.int32)[0]) - int(o[kk:kk + 1].view(np.int32)[0])))
    print(agree)
