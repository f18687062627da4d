"""spx_pipeline (include/speedy_hip.h): the owning pipeline -- batch after batch of one shape, host memory to host memory, with
the library issuing the copy in, the overlapped batch call and the gather into pinned host memory itself.  Whatever is in flight
beside a batch, its output must be the oracle's (and spx_batch_run's)."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _crc(outs):
    return [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in outs]


def _signals(rate, ch, n_streams, seed, lens):
    from speedy_amd.synth import speech_like
    base = [speech_like(max(lens), rate, seed=1000 * seed + i, channels=ch) for i in range(min(n_streams, 10))]
    return [base[i % len(base)][: lens[i] * ch] for i in range(n_streams)]


@pytest.mark.parametrize("rate,ch,n_streams,depth", [(16000, 1, 256, 4), (16000, 1, 256, 3), (16000, 1, 61, 2), (22050, 1, 256, 4),
                                                     (16000, 2, 128, 3), (48000, 2, 40, 4), (16000, 1, 600, 3),
                                                     # 257 .. 512 streams: every batch call is cut into two overlapping sub-batches (run_split)
                                                     (16000, 1, 400, 3), (16000, 1, 512, 4), (16000, 2, 300, 3)])
def test_batches_of_different_content_through_the_pipeline(orc, rate, ch, n_streams, depth):
    """Three batches of one shape and DIFFERENT content go through the pipeline again and again (14 submits, tickets waited for
    with a lag, as a caller that keeps the device busy would): every ticket's output must be what spx_batch_run gives for that
    content, and that is checked against the oracle on three streams per content."""
    import torch
    from speedy_amd.batch import Batch, Pipeline, Plan
    plan = Plan(rate, False)
    rng = np.random.default_rng(n_streams * 7 + depth)
    lens = [int(rate * rng.uniform(0.4, 1.1)) for _ in range(n_streams)]
    b = Batch(plan, lens, ch, 3.5, 1.0, 0.0)
    contents, want = [], []
    for seed in (11, 12, 13):
        xs = _signals(rate, ch, n_streams, seed, lens)
        b.upload(xs)
        b.run()
        crcs = _crc(b.results())
        for i in (0, n_streams // 2, n_streams - 1):
            ref = orc.compress_sound(xs[i], rate, ch, 3.5, 1.0, 0.0, False, chunk=1000)
            assert crcs[i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, i)
        want.append(crcs)
    pipe = Pipeline(plan, lens, ch, 3.5, 1.0, 0.0, depth=depth)
    assert pipe.depth == depth
    for xs_seed in (11, 12, 13):
        contents.append(torch.from_numpy(pipe.pack(_signals(rate, ch, n_streams, xs_seed, lens))).pin_memory())
    tickets = []
    lag = depth - 1
    for k in range(14):
        tickets.append((pipe.submit(contents[k % 3]), k % 3))
        if k >= lag:
            t, c = tickets[k - lag]
            assert _crc(pipe.results(t)) == want[c], (k, t, c)
    for t, c in tickets[len(tickets) - lag:]:
        assert _crc(pipe.results(t)) == want[c], (t, c)
    if (rate, ch, n_streams) == (16000, 1, 256):
        assert plan.L.spx_debug_last_call_concurrent() == 2          # the headline's shape takes the pipelined order
        assert plan.L.spx_debug_last_walk_form() == (16 * 4 + 0 if depth >= 3 else 16 * 4 + 4)   # ... lean walk kernels from three buffer sets on
    # an expired ticket is refused, a future one too
    with pytest.raises(RuntimeError):
        pipe.wait(tickets[0][0])
    with pytest.raises(RuntimeError):
        pipe.wait(tickets[-1][0] + 1)
    pipe.close()


def test_offsets_are_aligned_and_counts_are_the_batch_calls(orc):
    import torch
    from speedy_amd.batch import Batch, Pipeline, Plan
    plan = Plan(16000, False)
    lens = [16000 + 137 * i for i in range(33)]
    xs = _signals(16000, 1, 33, 5, lens)
    b = Batch(plan, lens, 1, 2.0, 0.0, 0.0)       # linear jobs: the TSM stage alone
    b.upload(xs)
    b.run()
    res = b.results()
    pipe = Pipeline(plan, lens, 1, 2.0, 0.0, 0.0, depth=2)
    t = pipe.submit(pipe.pack(xs))                 # pageable host memory is accepted as well
    out, offsets, counts = pipe.wait(t)
    assert all(int(o) % 32 == 0 for o in offsets)
    assert [int(c) for c in counts] == [r.size for r in res]
    for i in range(33):
        assert np.array_equal(out[int(offsets[i]):int(offsets[i]) + int(counts[i])], res[i]), i
    pipe.close()


def test_device_input_and_device_output(orc):
    """The resident form (bench.py's `value`): the input is a device tensor, the outputs stay in device memory."""
    import torch
    from speedy_amd.batch import Batch, Pipeline, Plan
    plan = Plan(16000, False)
    lens = [24000] * 256
    xs = _signals(16000, 1, 256, 21, lens)
    b = Batch(plan, lens, 1, 3.5, 1.0, 0.0)
    b.upload(xs)
    b.run()
    want = _crc(b.results())
    ref = orc.compress_sound(xs[3], 16000, 1, 3.5, 1.0, 0.0, False, chunk=1000)
    assert want[3] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes())
    pipe = Pipeline(plan, lens, 1, 3.5, 1.0, 0.0, depth=4, device_out=True)
    d_in = b.d_in
    ts = [pipe.submit(d_in) for _ in range(9)]
    for t in ts[-4:]:
        assert _crc(pipe.results(t)) == want, t
    pipe.close()


def test_staging_buffer_of_the_pipeline(orc):
    """spx_pipeline_host_input: the caller produces each batch's input in the pipeline's own pinned staging buffer."""
    from speedy_amd.batch import Batch, Pipeline, Plan
    plan = Plan(22050, False)
    lens = [22050] * 40
    b = Batch(plan, lens, 2, 1.5, 1.0, 0.0)
    want = []
    packed = []
    pipe = Pipeline(plan, lens, 2, 1.5, 1.0, 0.0, depth=3)
    for seed in (31, 32):
        xs = _signals(22050, 2, 40, seed, lens)
        b.upload(xs)
        b.run()
        want.append(_crc(b.results()))
        packed.append(pipe.pack(xs))
    ref = orc.compress_sound(_signals(22050, 2, 40, 32, lens)[7], 22050, 2, 1.5, 1.0, 0.0, False, chunk=1000)
    assert want[1][7] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes())
    ts = []
    for k in range(8):
        h = pipe.host_input()
        h[:] = packed[k % 2]
        ts.append((pipe.submit(h), k % 2))
    for t, c in ts[-3:]:
        assert _crc(pipe.results(t)) == want[c], (t, c)
    pipe.close()


def test_mixed_rate_pipeline(orc):
    """spx_pipeline_create_mixed: one GPU's kind of BASELINE configs[4] shard (16 / 22.05 kHz, mono / stereo, 1.5x / 3.5x) in small."""
    import torch
    from speedy_amd.batch import MixedBatch, Pipeline, Plan
    from speedy_amd.synth import speech_like
    rates = [16000, 22050]
    plans = [Plan(r, False) for r in rates]
    n = 120
    pidx = [i % 2 for i in range(n)]
    chs = [1 if (i // 2) % 2 == 0 else 2 for i in range(n)]
    speeds = [1.5 if (i // 4) % 2 == 0 else 3.5 for i in range(n)]
    lens = [int(rates[pidx[i]] * (0.5 + 0.004 * i)) for i in range(n)]
    want, packed = [], []
    mb = MixedBatch(plans, pidx, lens, chs, speeds, 1.0, 0.0)
    pipe = Pipeline(plans, lens, chs, speeds, 1.0, 0.0, depth=3, plan_index=pidx)
    for seed in (41, 42, 43):
        xs = [speech_like(lens[i], rates[pidx[i]], seed=seed * 100 + (i % 12), channels=chs[i]) for i in range(n)]
        mb.upload(xs)
        mb.run()
        want.append(mb.crcs())
        for i in (0, 61, n - 1):
            ref = orc.compress_sound(xs[i], rates[pidx[i]], chs[i], speeds[i], 1.0, 0.0, False, chunk=1000)
            assert want[-1][i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, i)
        packed.append(torch.from_numpy(pipe.pack(xs)).pin_memory())
    ts = [(pipe.submit(packed[k % 3]), k % 3) for k in range(3)]
    for k in range(3, 12):
        t, c = ts[k - 2]
        assert _crc(pipe.results(t)) == want[c], (t, c)
        ts.append((pipe.submit(packed[k % 3]), k % 3))
    for t, c in ts[-2:]:
        assert _crc(pipe.results(t)) == want[c], (t, c)
    pipe.close()


@pytest.mark.parametrize("rates,n,depth", [((16000, 22050), 256, 4), ((16000, 22050), 256, 2), ((16000, 22050), 90, 3),
                                           ((16000, 22050, 11025), 150, 4), ((22050,), 64, 4)])
def test_mixed_rate_pipeline_outputs_on_the_device_walk_kernels_overlapping(orc, rates, n, depth):
    """Round 6: a mixed pipeline that leaves its outputs on the device is DETACHED -- the first two groups' walk kernels of consecutive
    batches overlap on the library's walk streams (spx_mixed_walk2, spx_mode.h), a third group runs on its plan's own stream.  Batches
    of different content through every buffer set, many more batches than sets: every batch equals the plain mixed call's bytes, samples
    of it the oracle's."""
    import torch
    from speedy_amd.batch import MixedBatch, Pipeline, Plan
    from speedy_amd.synth import speech_like
    plans = [Plan(r, False) for r in rates]
    pidx = [i % len(rates) for i in range(n)]
    chs = [1 if (i // 2) % 2 == 0 else 2 for i in range(n)]
    speeds = [1.5 if (i // 4) % 2 == 0 else 3.5 for i in range(n)]
    lens = [int(rates[pidx[i]] * (1.0 + 0.006 * (i % 97))) for i in range(n)]
    mb = MixedBatch(plans, pidx, lens, chs, speeds, 1.0, 0.0)
    pipe = Pipeline(plans, lens, chs, speeds, 1.0, 0.0, depth=depth, device_out=True, plan_index=pidx)
    want, dev_in = [], []
    for seed in (51, 52, 53):
        xs = [speech_like(lens[i], rates[pidx[i]], seed=seed * 100 + (i % 9), channels=chs[i]) for i in range(n)]
        mb.upload(xs)
        mb.run()
        want.append(mb.crcs())
        for i in (0, n // 2 + 1, n - 1):
            ref = orc.compress_sound(xs[i], rates[pidx[i]], chs[i], speeds[i], 1.0, 0.0, False, chunk=1000)
            assert want[-1][i] == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), (seed, i)
        d = torch.zeros(pipe.total_in + 64, dtype=torch.int16, device="cuda")
        d[: pipe.total_in].copy_(torch.from_numpy(pipe.pack(xs)))
        dev_in.append(d)
    torch.cuda.synchronize()
    ts = []
    for k in range(5 * depth + 3):
        ts.append((pipe.submit(dev_in[k % 3]), k % 3))
        if len(ts) > depth:          # the oldest batch still held by a buffer set
            t, c = ts[-depth]
            assert _crc(pipe.results(t)) == want[c], (k, t, c)
    for t, c in ts[-depth:]:
        assert _crc(pipe.results(t)) == want[c], (t, c)
    pipe.close()


def test_two_pipelines_and_plain_calls_side_by_side(orc):
    """Two pipelines of different plans alive at once, with plain spx_batch_run calls of a third shape in between."""
    from speedy_amd.batch import Batch, Pipeline, Plan
    p16, p22 = Plan(16000, False), Plan(22050, False)
    l16, l22 = [16000] * 256, [11025] * 100
    x16, x22 = _signals(16000, 1, 256, 51, l16), _signals(22050, 1, 100, 52, l22)
    b16, b22 = Batch(p16, l16, 1, 3.5, 1.0, 0.0), Batch(p22, l22, 1, 3.5, 1.0, 0.0)
    b16.upload(x16); b16.run(); w16 = _crc(b16.results())
    b22.upload(x22); b22.run(); w22 = _crc(b22.results())
    for i, (x, rate, w) in enumerate(((x16[5], 16000, w16[5]), (x22[9], 22050, w22[9]))):
        ref = orc.compress_sound(x, rate, 1, 3.5, 1.0, 0.0, False, chunk=1000)
        assert w == zlib.crc32(np.ascontiguousarray(ref["out"]).tobytes()), i
    pa, pb = Pipeline(p16, l16, 1, 3.5, 1.0, 0.0, depth=3), Pipeline(p22, l22, 1, 3.5, 1.0, 0.0, depth=2)
    ia, ib = pa.pack(x16), pb.pack(x22)
    ta, tb = [], []
    for k in range(6):
        ta.append(pa.submit(ia))
        tb.append(pb.submit(ib))
        if k % 2:
            b16.run()
    for t in ta[-3:]:
        assert _crc(pa.results(t)) == w16
    for t in tb[-2:]:
        assert _crc(pb.results(t)) == w22
    assert _crc(b16.results()) == w16
    # ... and the caller's OWN overlapped calls on the plan a detached pipeline is working on (outputs left on the device: its calls
    # leave no note of any stream in the plan's ring): every copy taken right behind a call holds that call's output
    import torch
    pd = Pipeline(p16, l16, 1, 3.5, 1.0, 0.0, depth=4, device_out=True)
    b16b = Batch(p16, l16, 1, 3.5, 1.0, 0.0)
    b16b.upload(x16)
    b16.d_out.zero_(); b16b.d_out.zero_()
    torch.cuda.synchronize()
    copies, td = [], []
    for k in range(8):
        td.append(pd.submit(b16.d_in))
        q = (b16, b16b)[k % 2]
        q.run_ahead(overlap=True)
        copies.append((q, q.d_out.clone(), q.d_nout.clone()))
    torch.cuda.synchronize()
    for q, o, c in copies:
        keep_o, keep_c = q.d_out, q.d_nout
        q.d_out, q.d_nout = o, c
        assert _crc(q.results()) == w16
        q.d_out, q.d_nout = keep_o, keep_c
    for t in td[-4:]:
        assert _crc(pd.results(t)) == w16
    pa.close(); pb.close(); pd.close()


def test_input_may_be_overwritten_once_consumed(orc):
    """spx_pipeline_input_consumed: a HOST input may be overwritten once its copy in has been made, a DEVICE input only once the
    batch is done (the walk kernel reads it to the end) -- overwriting right behind the call never changes a batch's output, for
    either kind, with several batches in flight."""
    import torch
    from speedy_amd.batch import Pipeline, Plan
    from speedy_amd.synth import speech_like
    rate, n = 16000, 24000
    plan = Plan(rate, False)
    lens = [n - 100 * i for i in range(40)]
    xa = [speech_like(l, rate, seed=900 + i) for i, l in enumerate(lens)]
    xb = [speech_like(l, rate, seed=950 + i) for i, l in enumerate(lens)]
    pipe = Pipeline(plan, lens, 1, 3.0, 1.0, 0.0, depth=3)
    ia, ib = pipe.pack(xa), pipe.pack(xb)
    want = {}
    for key, inp in (("a", ia), ("b", ib)):
        want[key] = _crc(pipe.results(pipe.submit(inp)))
    assert want["a"] != want["b"]
    # host memory: one buffer, refilled as soon as each batch's input has been consumed
    buf = ia.copy()
    seq, tickets = "abbaabab", []
    for k, key in enumerate(seq):
        buf[:] = ia if key == "a" else ib
        t = pipe.submit(buf)
        pipe.input_consumed(t)
        buf[:] = 0x5555                       # garbage behind the copy
        tickets.append(t)
        if k >= 2:
            assert _crc(pipe.results(tickets[k - 2])) == want[seq[k - 2]], k
    # device memory: the same with a device buffer; consumed = the batch is done
    dbuf = torch.zeros(pipe.total_in + 64, dtype=torch.int16, device="cuda")   # (+ 64: include/speedy_hip.h spx_pipeline_submit, device input)
    da, db = torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()
    for key in "abba":
        dbuf[: pipe.total_in].copy_((da if key == "a" else db)[: pipe.total_in])
        torch.cuda.synchronize()
        t = pipe.submit(dbuf)
        pipe.input_consumed(t)
        dbuf.fill_(0x5555)
        torch.cuda.synchronize()
        assert _crc(pipe.results(t)) == want[key], key
    with pytest.raises(RuntimeError):
        pipe.input_consumed(10 ** 6)
    pipe.close()


def test_refused_arguments_edge_shapes_and_destroy_in_flight(orc):
    """What the pipeline refuses (depth 1 or above 8, no streams, a plan index out of range, a bad job), the shapes at the edge --
    ONE stream, streams too short for a single analysis frame or EMPTY beside ordinary ones, linear jobs (no analysis at all),
    a slow-down job -- against the oracle, and a pipeline destroyed with batches still in flight."""
    import ctypes as C
    from speedy_amd.batch import Pipeline, Plan, StreamJob
    from speedy_amd.synth import speech_like
    rate = 16000
    plan = Plan(rate, False)
    L = plan.L
    for depth in (1, 9, -1):
        with pytest.raises(RuntimeError):
            Pipeline(plan, [8000] * 4, 1, 3.0, 1.0, 0.0, depth=depth)
    jobs = (StreamJob * 1)()
    assert not L.spx_pipeline_create(plan.h, jobs, 0, 0, 0) and b"bad arguments" in L.spx_last_error()
    with pytest.raises(RuntimeError):
        Pipeline([plan], [8000] * 2, 1, 3.0, 1.0, 0.0, plan_index=[0, 1])
    with pytest.raises(RuntimeError):
        Pipeline(plan, [8000, 8000], [1, 0], 3.0, 1.0, 0.0)       # channels < 1
    # edge shapes
    cases = [([24000], [3.0], [1.0]),                                  # one stream
             ([24000, 0, 100, 239, 16000, 1], [3.0, 2.0, 3.5, 2.5, 1.5, 2.0], [1.0, 1.0, 1.0, 1.0, 1.0, 1.0]),   # empty / shorter than a window
             ([20000, 12000, 16000], [2.0, 0.5, 1.0], [0.0, 0.0, 0.0])]    # linear only: speed-up, slow-down, unity
    for lens, speeds, nls in cases:
        xs = [speech_like(n, rate, seed=1100 + i) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)]
        pipe = Pipeline(plan, lens, 1, speeds, nls, 0.0, depth=2)
        inp = pipe.pack(xs)
        ts = [pipe.submit(inp) for _ in range(3)]
        outs = pipe.results(ts[-1])
        assert _crc(pipe.results(ts[-2])) == _crc(outs)
        for i, (x, got) in enumerate(zip(xs, outs)):
            ref = orc.compress_sound(x, rate, 1, speeds[i], nls[i], 0.0, False, chunk=max(1, x.size), taps=False)["out"] if x.size else np.zeros(0, np.int16)
            assert np.array_equal(got, ref), (lens, i, got.size, ref.size)
        pipe.close()
    # destroyed with batches in flight (and never waited for): nothing hangs, the plan serves the next pipeline
    n = 10 * rate
    big = [speech_like(n, rate, seed=1200 + i) for i in range(8)] * 32
    pipe = Pipeline(plan, [n] * 256, 1, 3.5, 1.0, 0.0, depth=4)
    inp = pipe.pack(big)
    for _ in range(6):
        pipe.submit(inp)
    pipe.close()
    pipe = Pipeline(plan, [n] * 256, 1, 3.5, 1.0, 0.0, depth=3)
    t = pipe.submit(inp)
    outs = pipe.results(t)
    ref = orc.compress_sound(big[3], rate, 1, 3.5, 1.0, 0.0, False, chunk=1000, taps=False)["out"]
    assert np.array_equal(outs[3], ref) and np.array_equal(outs[3 + 8 * 31], ref)
    pipe.close()


def test_two_host_threads_each_with_a_pipeline_on_one_plan(orc):
    """Two host threads, each submitting to a pipeline of its own, both on ONE plan (the plan's ring of calls, its staging slots and
    the library's side and walk streams are shared; a pipeline itself is not thread-safe and is not shared): every batch of either
    thread holds its own content's output."""
    import threading
    from speedy_amd.batch import Batch, Pipeline, Plan
    from speedy_amd.synth import speech_like
    rate, n = 16000, 40000
    plan = Plan(rate, False)
    shapes = {"a": ([n] * 200, 3.5), "b": ([n - 333 * (i % 7) for i in range(150)], 2.0)}
    xs = {k: [speech_like(l, rate, seed=(1300 if k == "a" else 1500) + i) for i, l in enumerate(v[0])] for k, v in shapes.items()}
    want = {}
    for k, (lens, speed) in shapes.items():
        b = Batch(plan, lens, 1, speed, 1.0, 0.0)
        b.upload(xs[k])
        b.run()
        outs = b.results()
        want[k] = _crc(outs)
        ref = orc.compress_sound(xs[k][5], rate, 1, speed, 1.0, 0.0, False, chunk=1000, taps=False)["out"]
        assert np.array_equal(outs[5], ref)
    errors = []

    def work(k):
        try:
            lens, speed = shapes[k]
            pipe = Pipeline(plan, lens, 1, speed, 1.0, 0.0, depth=3)
            inp = pipe.pack(xs[k])
            ts = []
            for j in range(10):
                ts.append(pipe.submit(inp))
                if j >= 2:
                    assert _crc(pipe.results(ts[j - 2])) == want[k], (k, j)
            for t in ts[-2:]:
                assert _crc(pipe.results(t)) == want[k], (k, t)
            pipe.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in shapes]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors


@pytest.mark.parametrize("rate", [16000, 22050])
def test_pipelined_batches_against_the_oracle_at_scale(rate):
    """The path the bench's `value` is measured on -- device-resident input, outputs left on the device, four buffer sets, walk
    kernels of consecutive batches overlapping -- against the CPU port at scale: seven settings (mono / stereo, speeds 0.7 ... 5.5,
    linear / nonlinear, duration feedback) x 4 batches x 256 two-second noise streams, three batches in flight, CRC-32 per stream
    (tools/r11_probe.py pipeline 40: 71 680 streams per rate, profiles/r05/r5zg_pipeline_probe.txt)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import r11_probe
    bad, total = r11_probe.pipeline_audio_against_the_oracle(rate, 4, verbose=False)
    assert total == 7 * 4 * 256 and bad == 0, (rate, bad, total)
