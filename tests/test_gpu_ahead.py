"""spx_batch_run_ahead (include/speedy_hip.h): consecutive batch calls software-pipelined -- a call's analysis and tension kernels
run beside the previous call's walk kernel.  The results must be spx_batch_run's (and the oracle's) whatever the calls overlap
with: batches of DIFFERENT content taking turns on two and three workspaces, one workspace handed over again and again, shapes
that do not fit the mode, other calls in between."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _crc(outs):
    return [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs]


def _make(plan, rate, ch, n_streams, seed, seconds, speed=3.5, nl=1.0, fb=0.0, ragged=True):
    from speedy_amd.batch import Batch
    from speedy_amd.synth import speech_like
    rng = np.random.default_rng(seed)
    lens = [int(rate * seconds * (rng.uniform(0.3, 1.0) if ragged else 1.0)) for _ in range(n_streams)]
    base = [speech_like(max(lens), rate, seed=1000 * seed + i, channels=ch) for i in range(min(n_streams, 12))]
    streams = [base[i % len(base)][: lens[i] * ch] for i in range(n_streams)]
    b = Batch(plan, lens, ch, speed, nl, fb)
    b.upload(streams)
    return b, streams


@pytest.mark.parametrize("rate,ch,n_streams", [(16000, 1, 256), (16000, 1, 97), (22050, 1, 256), (16000, 2, 128), (48000, 2, 64)])
def test_alternating_batches_of_different_content(orc, rate, ch, n_streams):
    """Three batches with different signals and lengths take turns (a ring of three workspaces, then of two): every pass of
    every batch must reproduce what spx_batch_run gives for it; a sample of the streams is checked against the oracle."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(rate, False)
    bs = []
    for seed in (1, 2, 3):
        b, streams = _make(plan, rate, ch, n_streams, seed, seconds=1.5)
        b.run()
        want = _crc(b.results())
        for i in (0, n_streams // 2, n_streams - 1):
            ref = orc.compress_sound(streams[i], rate, ch, 3.5, 1.0, 0.0, False, chunk=1000)
            assert want[i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, i)
        b.d_out.zero_()
        bs.append((b, want))
    torch.cuda.synchronize()
    for ring in (3, 2):
        order = [k % ring for k in range(9)]
        for k in order:
            bs[k][0].run_ahead()
        if (rate, ch) == (16000, 1):     # the shapes of the headline take the mode (the others decide by their co-residency arithmetic)
            assert plan.L.spx_debug_last_call_concurrent() == 2
        torch.cuda.synchronize()
        for k in set(order):
            assert _crc(bs[k][0].results()) == bs[k][1], (ring, k)
            bs[k][0].d_out.zero_()


def test_one_workspace_again_and_again(orc):
    """The caller breaks the contract's advice (no second workspace): every call then waits for the previous one -- same bytes."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(16000, False)
    import time
    b, _ = _make(plan, 16000, 1, 200, 7, seconds=1.0)
    b.run()
    want = _crc(b.results())
    for _ in range(6):
        b.run_ahead()
    torch.cuda.synchronize()
    assert _crc(b.results()) == want
    # ... and at a plain call's price: in round 4 the gate kernel in front of every such call spun its full bound for a counter
    # that the call's own staging kernel had just cleared (2 ms of idle device per call: 3.6 ms where a plain call takes 1.6)
    def window(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    plain, again = window(b.run), window(b.run_ahead)
    assert again < 1.35 * plain + 0.2e-3, (again, plain)
    assert _crc(b.results()) == want


def test_shapes_outside_the_mode_and_calls_in_between(orc):
    """More streams than CUs, slow-down jobs, linear jobs, a plain spx_batch_run between two pipelined calls, two plans of
    different rates taking turns on the same HIP stream: spx_batch_run_ahead decides per call and the bytes never change."""
    import torch
    from speedy_amd.batch import Plan
    p16, p22 = Plan(16000, False), Plan(22050, False)
    cases = [(p16, 16000, 1, 600, 3.5, 1.0), (p16, 16000, 1, 64, 0.7, 1.0), (p16, 16000, 1, 64, 2.0, 0.0),
             (p22, 22050, 2, 100, 1.5, 1.0), (p16, 16000, 1, 256, 3.5, 1.0), (p22, 22050, 1, 256, 3.5, 1.0)]
    bs = []
    for j, (plan, rate, ch, n, speed, nl) in enumerate(cases):
        b, _ = _make(plan, rate, ch, n, 20 + j, seconds=0.8, speed=speed, nl=nl)
        b.run()
        bs.append((b, _crc(b.results())))
        b.d_out.zero_()
    torch.cuda.synchronize()
    seq = [4, 5, 4, 0, 5, 1, 4, 2, 5, 3, 4, 5]
    for t, k in enumerate(seq):
        if t == 6:
            bs[k][0].run()          # a plain call in between
        else:
            bs[k][0].run_ahead()
    torch.cuda.synchronize()
    for k in set(seq):
        assert _crc(bs[k][0].results()) == bs[k][1], k


def test_input_that_arrives_on_another_stream(orc):
    """spx_batch_run_ahead_when: the input of every call is still being copied (pinned host -> device, on a stream of the
    caller's) when the call is made; the call's producers wait for the caller's event, not for the caller's stream."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(16000, False)
    bs, want, hosts = [], [], []
    for seed in (31, 32):
        b, streams = _make(plan, 16000, 1, 256, seed, seconds=1.0)
        b.run()
        want.append(_crc(b.results()))
        hosts.append(b.d_in.cpu().pin_memory())
        bs.append(b)
    s_copy = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in bs]
    done = [torch.cuda.Event() for _ in bs]
    torch.cuda.synchronize()
    for k in range(8):
        i = k % 2
        with torch.cuda.stream(s_copy):
            if k >= 2:
                s_copy.wait_event(done[i])          # the walk of the call that last read this input has finished
            bs[i].d_in.zero_()                      # whoever reads too early reads silence
            bs[i].d_in.copy_(hosts[i], non_blocking=True)
            evs[i].record(s_copy)
        bs[i].d_out.zero_()
        bs[i].run_ahead(in_ready=evs[i])
        done[i].record()
    torch.cuda.synchronize()
    for i in range(2):
        assert _crc(bs[i].results()) == want[i], i


def test_two_host_threads_each_with_a_plan_and_a_stream(orc):
    """Two threads pipeline their own batches on their own plans and HIP streams; the library's side stream is the device's, so
    their producers (and gate kernels) share it."""
    import threading
    import torch
    from speedy_amd.batch import Plan
    work, errors = [], []
    for t, rate in enumerate((16000, 22050)):
        plan = Plan(rate, False)
        pair = []
        for seed in (41 + 2 * t, 42 + 2 * t):
            b, _ = _make(plan, rate, 1, 256, seed, seconds=1.0)
            b.run()
            pair.append((b, _crc(b.results())))
            b.d_out.zero_()
        work.append(pair)
    torch.cuda.synchronize()

    def body(pair):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            for k in range(10):
                pair[k % 2][0].run_ahead(stream=s)
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=body, args=(pair,)) for pair in work]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for pair in work:
        for b, want in pair:
            assert _crc(b.results()) == want


def test_pipelined_calls_are_faster_than_plain_ones():
    """What the mode is for (bench.py's `value` against its `unpipelined`): the bench batch, two workspaces."""
    import time
    import torch
    import bench
    from speedy_amd.batch import Batch, Plan
    n = bench.RATE * bench.SECONDS
    plan = Plan(bench.RATE, False)
    streams = bench.make_streams(bench.STREAMS_PER_GPU, n, 0)
    bs = [Batch(plan, [n] * len(streams), 1, bench.SPEED, 1.0, 0.0) for _ in range(2)]
    for b in bs:
        b.upload(streams)
    plain = bench.time_window(bs[0].run, 12, 4)
    for k in range(4):
        bs[k % 2].run_ahead()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(12):
        bs[k % 2].run_ahead()
    torch.cuda.synchronize()
    ahead = (time.perf_counter() - t0) / 12
    assert ahead < 0.97 * plain, (ahead, plain)


def test_mixed_rate_calls_pipelined(orc):
    """spx_batch_run_mixed_ahead: three mixed-rate batches of different content (16 / 22.05 / 8 kHz, mono and stereo, two speeds,
    ragged lengths) take turns on rings of three and two workspaces, with a plain mixed call and a call of more streams than the
    device has CUs in between; every pass must give what spx_batch_run_mixed gives for the batch, and a sample of the streams is
    checked against the oracle."""
    import torch
    from speedy_amd.batch import MixedBatch, Plan
    from speedy_amd.synth import speech_like
    rates = [16000, 22050, 8000]
    plans = [Plan(r, False) for r in rates]

    def make(seed, n):
        rng = np.random.default_rng(seed)
        pidx = [int(rng.integers(0, 3)) for _ in range(n)]
        chs = [int(rng.choice([1, 1, 2])) for _ in range(n)]
        speeds = [float(rng.choice([1.5, 3.5])) for _ in range(n)]
        lens = [int(rates[pidx[i]] * rng.uniform(0.2, 0.9)) for i in range(n)]
        xs = [speech_like(lens[i], rates[pidx[i]], seed=seed * 1000 + (i % 16), channels=chs[i]) for i in range(n)]
        b = MixedBatch(plans, pidx, lens, chs, speeds, 1.0, 0.0)
        b.upload(xs)
        b.run()
        want = b.crcs()
        res = b.results()
        for i in (0, n // 3, n - 1):
            ref = orc.compress_sound(xs[i], rates[pidx[i]], chs[i], speeds[i], 1.0, 0.0, False, chunk=1000)
            assert np.array_equal(res[i], ref["out"]), (seed, i)
        b.d_out.zero_()
        return b, want

    bs = [make(51, 200), make(52, 256), make(53, 77)]
    big = make(54, 700)
    torch.cuda.synchronize()
    for ring in (3, 2):
        for t in range(10):
            k = t % ring
            if t == 4:
                bs[k][0].run()
            elif t == 7:
                big[0].run_ahead()          # more streams than CUs: runs as a plain call would, between two pipelined ones
                bs[k][0].run_ahead()
            else:
                bs[k][0].run_ahead()
        torch.cuda.synchronize()
        for k in range(ring):
            assert bs[k][0].crcs() == bs[k][1], (ring, k)
            bs[k][0].d_out.zero_()
        assert big[0].crcs() == big[1]
        big[0].d_out.zero_()


@pytest.mark.parametrize("rate,ch,n_streams", [(16000, 1, 256), (22050, 1, 256), (16000, 2, 200), (22050, 2, 128), (48000, 1, 64),
                                               (16000, 1, 390), (22050, 1, 512)])    # (more streams than CUs: two overlapping sub-batches per call)
def test_overlapped_walks_of_batches_of_different_content(orc, rate, ch, n_streams):
    """spx_batch_run_overlapped: as the first test, with the walk kernels of consecutive calls overlapping (in their lean form
    where three workspaces take turns and the streams are mono); every batch is consumed (copied) right behind its own call, as
    the call's contract asks.  Shapes outside the mode (48 kHz) run as plain calls."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(rate, False)
    bs = []
    for seed in (71, 72, 73):
        b, streams = _make(plan, rate, ch, n_streams, seed, seconds=1.2)
        b.run()
        bs.append((b, _crc(b.results())))
        for i in (0, n_streams // 2, n_streams - 1):     # what the overlapped calls are compared with is the oracle's
            ref = orc.compress_sound(streams[i], rate, ch, 3.5, 1.0, 0.0, False, chunk=1000)
            assert bs[-1][1][i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, i)
        b.d_out.zero_()
    torch.cuda.synchronize()
    for ring in (3, 2):
        copies = []
        for t in range(9):
            b = bs[t % ring][0]
            b.run_ahead(overlap=True)
            copies.append((t % ring, b.d_out.clone(), b.d_nout.clone()))     # the consumer of THIS call's output
        if (rate, ch) == (16000, 1):
            assert plan.L.spx_debug_last_call_concurrent() == 2
        torch.cuda.synchronize()
        for k, o, c in copies:
            b = bs[k][0]
            keep_o, keep_c = b.d_out, b.d_nout
            b.d_out, b.d_nout = o, c
            assert _crc(b.results()) == bs[k][1], (ring, k)
            b.d_out, b.d_nout = keep_o, keep_c


def test_one_output_buffer_for_every_call(orc):
    """Two workspaces take turns but the caller hands over the SAME out / n_out buffers every time and consumes them on the
    stream between the calls (here: a device copy): the next call's walk kernel -- which runs on a stream of the library's and
    would otherwise overlap the previous one -- must wait for that consumer."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(16000, False)
    bs, want = [], []
    for seed in (61, 62):
        b, _ = _make(plan, 16000, 1, 256, seed, seconds=1.0, ragged=False)
        b.run()
        torch.cuda.synchronize()
        want.append((b.d_out.clone(), b.d_nout.clone()))
        bs.append(b)
    bs[1].d_out, bs[1].d_nout = bs[0].d_out, bs[0].d_nout      # one output buffer for both
    torch.cuda.synchronize()
    stash = []
    for k in range(8):
        bs[k % 2].run_ahead(overlap=True)
        stash.append((bs[0].d_out.clone(), bs[0].d_nout.clone()))   # the consumer, on the caller's stream
    torch.cuda.synchronize()
    for k, (o, c) in enumerate(stash):
        assert torch.equal(c, want[k % 2][1]), k
        b = bs[k % 2]
        got, ref, cnt = o.cpu().numpy(), want[k % 2][0].cpu().numpy(), c.cpu().numpy()
        for i in range(b.n):          # the produced frames of every stream (what lies behind them in the shared buffer is the other batch's)
            a, e = b.out_offs[i], b.out_offs[i] + int(cnt[i])
            assert np.array_equal(got[a:e], ref[a:e]), (k, i)


def test_plain_calls_between_overlapped_ones_with_consumers(orc):
    """Two buffer sets, every output consumed (copied) right behind its call, and every third call a plain spx_batch_run: the
    overlapped call behind a plain one must still wait for the consumer of the buffer it overwrites (the note of where the
    caller's stream stood is left by every call, not only by overlapped ones)."""
    import torch
    from speedy_amd.batch import Plan
    plan = Plan(16000, False)
    bs = []
    for seed in (81, 82):
        b, _ = _make(plan, 16000, 1, 256, seed, seconds=1.0)
        b.run()
        bs.append((b, _crc(b.results())))
        b.d_out.zero_()
    torch.cuda.synchronize()
    copies = []
    for t in range(12):
        b = bs[t % 2][0]
        if t % 3 == 2:
            b.run()
        else:
            b.run_ahead(overlap=True)
        # a slow consumer: several passes over the output before the copy that is kept
        scratch = b.d_out.clone()
        for _ in range(4):
            scratch = scratch + b.d_out
        copies.append((t % 2, b.d_out.clone(), b.d_nout.clone(), scratch))
    torch.cuda.synchronize()
    for k, o, c, scratch in copies:
        b = bs[k][0]
        assert torch.equal(scratch, (o.to(torch.int32) * 5).to(torch.int16)), k
        keep_o, keep_c = b.d_out, b.d_nout
        b.d_out, b.d_nout = o, c
        assert _crc(b.results()) == bs[k][1], k
        b.d_out, b.d_nout = keep_o, keep_c


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_call_sequences(orc, seed):
    """Four batches of different content, sixty calls in a random order (a batch may come twice in a row, or after one, two or
    three others), each call plain, pipelined or overlapped at random, every output copied right behind its own call: every copy
    must hold what spx_batch_run gives for that batch (itself checked against the oracle on three streams per batch).  Round 5:
    MIXED-rate calls with the same lead plan (plain and pipelined) come in between too -- they go through the same ring and must
    leave their note of the caller's stream in it -- and some consumers are slow (several passes over the output)."""
    import torch
    from speedy_amd.batch import MixedBatch, Plan
    from speedy_amd.synth import speech_like
    rng = np.random.default_rng(1000 + seed)
    rate = int(rng.choice([16000, 16000, 22050]))
    plan = Plan(rate, False)
    bs = []
    for j in range(4):
        nst = int(rng.choice([256, 256, 131]))
        b, streams = _make(plan, rate, 1, nst, 90 + 10 * seed + j, seconds=float(rng.uniform(0.5, 1.2)))
        b.run()
        bs.append((b, _crc(b.results())))
        for i in (0, nst // 2, nst - 1):
            ref = orc.compress_sound(streams[i], rate, 1, 3.5, 1.0, 0.0, False, chunk=1000)
            assert bs[-1][1][i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, j, i)
        b.d_out.zero_()
    # a mixed-rate batch whose LEAD plan is the same plan (its ring is the one the calls above go through)
    p8 = Plan(8000, False)
    pidx = [int(v) for v in rng.integers(0, 2, 96)]
    mlens = [int((rate if g == 0 else 8000) * rng.uniform(0.3, 0.8)) for g in pidx]
    mx = [speech_like(mlens[i], rate if pidx[i] == 0 else 8000, seed=500 + seed * 100 + (i % 8)) for i in range(96)]
    mb = MixedBatch([plan, p8], pidx, mlens, 1, 3.5, 1.0, 0.0)
    mb.upload(mx)
    mb.run()
    m_want = mb.crcs()
    for i in (0, 47, 95):
        ref = orc.compress_sound(mx[i], rate if pidx[i] == 0 else 8000, 1, 3.5, 1.0, 0.0, False, chunk=1000)
        assert m_want[i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, "mixed", i)
    mb.d_out.zero_()
    torch.cuda.synchronize()
    copies, m_copies = [], []
    for t in range(60):
        kind = int(rng.integers(0, 5))
        if kind == 4:
            mb.run() if rng.integers(0, 2) else mb.run_ahead()
            m_copies.append((mb.d_out.clone(), mb.d_nout.clone()))
            continue
        k = int(rng.integers(0, 4))
        b = bs[k][0]
        if kind == 0:
            b.run()
        elif kind == 1:
            b.run_ahead()
        else:
            b.run_ahead(overlap=True)
        if rng.integers(0, 4) == 0:        # a slow consumer in front of the copy that is kept
            scratch = b.d_out.clone()
            for _ in range(3):
                scratch = scratch + b.d_out
        copies.append((k, b.d_out.clone(), b.d_nout.clone()))
        if rng.integers(0, 8) == 0:
            torch.cuda.synchronize()       # a pause now and then: the pipeline drains and fills again
    torch.cuda.synchronize()
    keep_o, keep_c = mb.d_out, mb.d_nout
    for t, (o, c) in enumerate(m_copies):
        mb.d_out, mb.d_nout = o, c
        assert mb.crcs() == m_want, (seed, "mixed", t)
    mb.d_out, mb.d_nout = keep_o, keep_c
    for t, (k, o, c) in enumerate(copies):
        b = bs[k][0]
        keep_o, keep_c = b.d_out, b.d_nout
        b.d_out, b.d_nout = o, c
        assert _crc(b.results()) == bs[k][1], (seed, t, k)
        b.d_out, b.d_nout = keep_o, keep_c


def test_sub_batches_of_an_overlapped_call_with_taps(orc):
    """An overlapped call of 257 .. 512 streams is cut into two overlapping sub-batches (run_split): every tap row -- tension, speed,
    features, spectrogram, normalised spectrum -- must land where the one-call layout puts it (rows of stream s at the sum of the
    frame counts of the streams before it, whatever sub-batch it fell into), and hold the oracle's values."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate, n_streams = 16000, 300
    plan = Plan(rate, False)
    rng = np.random.default_rng(5)
    lens = [int(rate * rng.uniform(0.3, 0.7)) for _ in range(n_streams)]
    base = [speech_like(max(lens), rate, seed=700 + i) for i in range(10)]
    xs = [base[i % 10][: lens[i]] for i in range(n_streams)]
    plain = Batch(plan, lens, 1, 3.5, 1.0, 0.0, taps=True, spectrogram_taps=True)
    plain.upload(xs)
    plain.run()
    want = _crc(plain.results())
    bs = []
    for _ in range(2):
        b = Batch(plan, lens, 1, 3.5, 1.0, 0.0, taps=True, spectrogram_taps=True)
        b.upload(xs)
        bs.append(b)
    torch.cuda.synchronize()
    for k in range(4):
        bs[k % 2].run_ahead(overlap=True)
    torch.cuda.synchronize()
    for b in bs:
        assert _crc(b.results()) == want
        for i in (0, 149, 150, 151, 299):       # both sides of the cut
            got, ref = b.tap_arrays(i), plain.tap_arrays(i)
            for key in ("tension", "speed", "features", "spectrogram", "normalized"):
                assert np.array_equal(got[key], ref[key]), (i, key)
    o = orc.compress_sound(xs[151], rate, 1, 3.5, 1.0, 0.0, False, chunk=1000)
    assert np.array_equal(bs[0].tap_arrays(151)["speed"], o["speed"]) and np.array_equal(bs[0].tap_arrays(151)["tension"], o["tension"])


def test_step_counts_after_a_split_call_then_the_two_halves_on_the_same_workspace(orc):
    """an overlapped call of 300 streams is cut into two sub-batches, and the plan remembers that for the workspace
    (spx_batch_read_steps looks for the state records in the sub-batches' slices).  spx_batch_analyze + spx_batch_walk on the SAME
    workspace afterwards run as one batch -- the record must go with them, or the step counts come from the wrong places."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate, n_streams = 16000, 300
    plan = Plan(rate, False)
    L = plan.L
    lens = [int(rate * (0.4 + 0.001 * i)) for i in range(n_streams)]
    xs = [speech_like(lens[i], rate, seed=40 + i % 12) for i in range(n_streams)]
    b = Batch(plan, lens, 1, 3.5, 1.0, 0.0)
    b.upload(xs)
    b.run()
    torch.cuda.synchronize()
    want_crc, want_steps = _crc(b.results()), list(b.step_counts())
    for _ in range(2):
        b.run_ahead(overlap=True)          # split in two: the record says k = 2
    torch.cuda.synchronize()
    assert list(b.step_counts()) == want_steps and _crc(b.results()) == want_crc
    hs = torch.cuda.current_stream().cuda_stream
    assert L.spx_batch_analyze(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_ws.data_ptr(), b.d_ws.numel(), None, hs) == 0
    assert L.spx_batch_walk(plan.h, b.jobs, b.n, b.d_in.data_ptr(), b.d_out.data_ptr(), b.d_nout.data_ptr(), b.d_ws.data_ptr(),
                            b.d_ws.numel(), None, hs) == 0
    torch.cuda.synchronize()
    assert list(b.step_counts()) == want_steps and _crc(b.results()) == want_crc


def test_large_calls_pipelined_producers_ahead(orc):
    """Round 6: a pipelined call of more than two streams per CU (time chunks, throughput-form walk kernels) starts its producers at
    once on the side stream -- the next call's first analysis chunk beside this call's last walk chunk.  600 ragged streams per call,
    three batches of different content taking turns, plain calls in between and one workspace handed over twice in a row: every
    call's outputs are the plain call's, and one stream per batch is the oracle's."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate, n_streams = 16000, 600
    plan = Plan(rate, False)
    L = plan.L
    rng = np.random.default_rng(17)
    base = [speech_like(rate, rate, seed=300 + i) for i in range(12)]
    bs, want = [], []
    for k in range(3):
        lens = [int(rate * rng.uniform(0.35, 1.0)) for _ in range(n_streams)]
        xs = [np.roll(base[(i + 5 * k) % 12], 31 * i)[: lens[i]] for i in range(n_streams)]
        b = Batch(plan, lens, 1, 3.5 if k != 1 else 1.5, 1.0, 0.0)
        b.upload(xs)
        b.run()
        torch.cuda.synchronize()
        want.append(_crc(b.results()))
        ref = orc.compress_sound(xs[577], rate, 1, 3.5 if k != 1 else 1.5, 1.0, 0.0, False, chunk=1000, taps=False)["out"]
        assert np.array_equal(b.results()[577], ref)
        bs.append(b)
    order = [0, 1, 2, 0, 1, 2, 2, 0, 1]          # (2, 2: the same workspace again -- the producers must wait for its own walk kernel)
    for j, k in enumerate(order):
        if j == 4:
            bs[0].run()                            # a plain call in between (its own workspace is not in flight: batch 0 ran at j = 3)
        bs[k].run_ahead(overlap=(j % 2 == 0))
        assert L.spx_debug_last_call_concurrent() == 0   # kernels in sequence (the producers' early start is not a mode of its own here)
    torch.cuda.synchronize()
    for k in range(3):
        assert _crc(bs[k].results()) == want[k], k


def test_mixed_batches_against_the_oracle_at_scale():
    """BASELINE configs[4]'s shard shape (256 streams: two rates, mono and stereo, speeds 1.5 and 3.5 in one spx_batch_run_mixed call),
    eight batches of ragged two-second noise streams, plain and pipelined calls taking turns: every stream's CRC-32 against the CPU
    port's (tools/r11_probe.py mixed 80: 20 480 streams, none differs -- profiles/r05/r5zh_mixed_probe.txt)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import r11_probe
    bad, total = r11_probe.mixed_audio_against_the_oracle(8, verbose=False)
    assert total == 8 * 256 and bad == 0, (bad, total)
