"""GPU parity (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the CPU oracle
on the same inputs.

Bars (north_star / task rules): int16 output and sample indexing bit-exact; float features / tension /
speed within 1e-4 (TOL below) -- and in fact bit-identical, because the kernels follow the oracle's
operation order (DESIGN.md "Why the floats are bit-exact"); that stronger property is asserted too.
"""
import os

import numpy as np
import pytest

from util import read_wav

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4  # north_star: "float spectrogram/tension within 1e-4"


def _batch(streams, rate, ch, speed, nl, fb, mm, taps=True, spec=False):
    from speedy_amd.batch import compress_batch
    return compress_batch(streams, rate, ch, speed, nl, fb, mm, taps=taps, spectrogram_taps=spec)


def _oracle(orc, x, rate, ch, speed, nl, fb, mm):
    return orc.compress_sound(x, rate, ch, speed, nl, fb, mm, chunk=1000)


@pytest.mark.parametrize("name,mm,speed,fb", [
    ("tapestry.wav", False, 3.5, 0.0),      # BASELINE configs[0]: speedy_wave defaults on tapestry.wav
    ("tapestry.wav", True, 3.5, 0.1),
    ("tapestry22050.wav", True, 3.0, 0.0),  # the file of the Matlab fixtures
    ("tapestry22050.wav", False, 1.5, 0.1),
    ("negative_speed.wav", True, 0.25, 0.1),  # speedy_test.cc:1059-1076
])
def test_reference_wavs_match_oracle(orc, name, mm, speed, fb):
    x, rate, ch = read_wav(name)
    outs, b = _batch([x], rate, ch, speed, 1.0, fb, mm)
    ref = _oracle(orc, x, rate, ch, speed, 1.0, fb, mm)
    taps = b.tap_arrays(0)
    assert taps["tension"].shape == ref["tension"].shape
    for key in ("tension", "speed", "features"):
        assert np.abs(taps[key] - ref[key]).max() <= TOL, key
        assert np.array_equal(taps[key], ref[key]), key + " not bit-identical"
    assert outs[0].size == ref["out"].size
    assert np.array_equal(outs[0], ref["out"])


def test_spectrogram_tap_matches_oracle(orc):
    """The spectrogram / normalised-spectrogram taps against the oracle's callbacks, every frame."""
    x, rate, ch = read_wav("tapestry22050.wav")
    L = orc.lib()
    spec_rows, norm_rows = [], []
    h = L.orc_sonicCreateStream(rate, ch, 1)
    n = L.orc_sonicSpectrogramSize(h)
    cb1 = orc.FEATURES_FN(lambda s, t, p: spec_rows.append(np.ctypeslib.as_array(p, shape=(n,)).copy()))
    cb2 = orc.FEATURES_FN(lambda s, t, p: norm_rows.append(np.ctypeslib.as_array(p, shape=(n,)).copy()))
    L.orc_sonicSpectrogramCallback(h, cb1)
    L.orc_sonicNormalizedSpectrogramCallback(h, cb2)
    L.orc_sonicSetSpeed(h, 3.0)
    L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
    L.orc_sonicWriteShortToStream(h, orc.sptr(x), x.size)
    L.orc_sonicDestroyStream(h)
    outs, b = _batch([x], rate, ch, 3.0, 1.0, 0.0, True, spec=True)
    taps = b.tap_arrays(0)
    ref = np.array(spec_rows)
    assert taps["spectrogram"].shape == ref.shape
    assert np.abs(taps["spectrogram"] - ref).max() <= TOL
    assert np.array_equal(taps["spectrogram"], ref)
    # the reference's callback at analysis call j hands out the buffer the tension computation k = j-F filled
    # (soniclib.c:303-310); the batch tap is indexed by k and holds the W computed bins
    F, W = b.plan.F, n // 2
    nref = np.array(norm_rows)[F:, :W]
    assert nref.shape[0] == len(norm_rows) - F and np.array_equal(taps["normalized"][:nref.shape[0]], nref)
    assert abs(float((taps["normalized"][150] ** 2).sum()) - 1.0) < 4e-3  # speedy_test.cc:975-978


@pytest.mark.parametrize("rate", [8000, 12000, 16000, 24000, 32000, 44100, 48000, 11025])
def test_spectrogram_taps_of_every_compiled_in_window(orc, rate):
    """The spectrogram and normalised-spectrogram taps (every bin of every frame, the mirrored upper half and bin W included)
    at the rates that have analysis kernels of their own -- 8 / 12 / 24 / 32 / 48 kHz over compiled-in plans, 44.1 kHz by Rader's
    algorithm over the 660-point plan, 16 kHz hand-written -- and at one that takes the plan-driven kernel (11.025 kHz);
    stream lengths that leave a partial last tile."""
    from speedy_amd.synth import speech_like
    L = orc.lib()
    for seed, n in ((3, int(0.83 * rate) + 17), (4, int(0.29 * rate))):
        x = speech_like(n, rate, seed=seed)
        spec_rows, norm_rows = [], []
        h = L.orc_sonicCreateStream(rate, 1, 0)
        nb = L.orc_sonicSpectrogramSize(h)
        cb1 = orc.FEATURES_FN(lambda s, t, p: spec_rows.append(np.ctypeslib.as_array(p, shape=(nb,)).copy()))
        cb2 = orc.FEATURES_FN(lambda s, t, p: norm_rows.append(np.ctypeslib.as_array(p, shape=(nb,)).copy()))
        L.orc_sonicSpectrogramCallback(h, cb1)
        L.orc_sonicNormalizedSpectrogramCallback(h, cb2)
        L.orc_sonicSetSpeed(h, 2.5)
        L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
        L.orc_sonicWriteShortToStream(h, orc.sptr(x), x.size)
        L.orc_sonicDestroyStream(h)
        outs, b = _batch([x], rate, 1, 2.5, 1.0, 0.0, False, spec=True)
        taps = b.tap_arrays(0)
        ref = np.array(spec_rows)
        assert taps["spectrogram"].shape == ref.shape, (rate, n)
        assert np.array_equal(taps["spectrogram"], ref), (rate, n)
        F, W = b.plan.F, nb // 2
        nref = np.array(norm_rows)[F:, :W]
        assert np.array_equal(taps["normalized"][:nref.shape[0]], nref), (rate, n)


@pytest.mark.parametrize("rate,ch,speed,nl,fb,mm", [
    (16000, 1, 3.5, 1.0, 0.0, False),
    (16000, 2, 3.5, 1.0, 0.1, False),
    (22050, 1, 1.5, 1.0, 0.0, False),
    (22050, 2, 1.5, 1.0, 0.1, True),
    (16000, 1, 2.0, 0.0, 0.0, False),   # BASELINE configs[1]: linear, TSM only
    (16000, 2, 3.0, 0.0, 0.0, False),
    (16000, 1, 0.4, 0.0, 0.0, False),   # slow-down: insertPitchPeriod (sonic_test.cc:536-589)
    (16000, 1, 0.7, 0.0, 0.0, False),
    (16000, 1, 1.0, 0.0, 0.0, False),   # copy-through
    (24000, 1, 3.5, 1.0, 0.0, False),
    (48000, 2, 2.5, 1.0, 0.0, False),
    (44100, 1, 3.5, 1.0, 0.0, False),   # W = 661, prime: the generic-radix DFT stage
    (8000, 1, 3.5, 1.0, 0.0, False),
    (16000, 1, 3.5, 1e-5, 0.0, True),   # sonic_test.cc's "nonlinear = 1e-5" (full path, ~linear)
])
def test_synthetic_streams_match_oracle(orc, rate, ch, speed, nl, fb, mm):
    from speedy_amd.synth import speech_like
    n = int(2.5 * rate)
    streams = [speech_like(n + 37 * i, rate, seed=i, channels=ch) for i in range(3)]
    outs, b = _batch(streams, rate, ch, speed, nl, fb, mm, taps=(nl != 0))
    for i, x in enumerate(streams):
        ref = _oracle(orc, x, rate, ch, speed, nl, fb, mm)
        if nl != 0:
            taps = b.tap_arrays(i)
            for key in ("tension", "speed", "features"):
                assert taps[key].shape == ref[key].shape
                assert np.abs(taps[key] - ref[key]).max() <= TOL, (i, key)
                assert np.array_equal(taps[key], ref[key]), (i, key, "not bit-identical")
        assert outs[i].size == ref["out"].size, (i, outs[i].size, ref["out"].size)
        assert np.array_equal(outs[i], ref["out"]), i


def test_edge_inputs(orc):
    """Empty, shorter-than-a-window, shorter-than-the-look-ahead and ragged batches in one launch."""
    from speedy_amd.synth import speech_like
    rate = 16000
    lens = [0, 1, 100, 241, 400, 2001, 2161, 5000, 16000]
    streams = [speech_like(n, rate, seed=n) for n in lens]
    for nl in (1.0, 0.0):
        outs, b = _batch(streams, rate, 1, 3.5, nl, 0.0, False, taps=False)
        for i, x in enumerate(streams):
            ref = _oracle(orc, x, rate, 1, 3.5, nl, 0.0, False)
            assert np.array_equal(outs[i], ref["out"]), (nl, lens[i])


def test_mixed_speeds_in_one_batch(orc):
    """Per-stream speed / feedback / channel count inside one launch (BASELINE configs[4] shape)."""
    from speedy_amd.synth import speech_like
    rate = 16000
    chs = [1, 2, 1, 2]
    speeds = [1.5, 3.5, 3.5, 1.5]
    fbs = [0.0, 0.1, 0.0, 0.1]
    streams = [speech_like(20000, rate, seed=10 + i, channels=chs[i]) for i in range(4)]
    outs, b = _batch(streams, rate, chs, speeds, 1.0, fbs, False, taps=False)
    for i, x in enumerate(streams):
        ref = _oracle(orc, x, rate, chs[i], speeds[i], 1.0, fbs[i], False)
        assert np.array_equal(outs[i], ref["out"]), i


def test_full_size_round_trip_properties():
    """BASELINE configs[3] shape at full size (256 x 10 s): properties that need no oracle.
    Length within 1.5 % of n/speed-ish bounds, determinism across two runs, partition independence."""
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate, n, B = 16000, 160000, 256
    plan = Plan(rate, False)
    base = [speech_like(n, rate, seed=i) for i in range(8)]
    streams = [base[i % 8] for i in range(B)]
    b = Batch(plan, [n] * B, 1, 3.5, 1.0, 0.0)
    b.upload(streams)
    b.run()
    r1 = b.results()
    b.run()
    r2 = b.results()
    for i in range(B):
        assert np.array_equal(r1[i], r2[i])
        assert np.array_equal(r1[i], r1[i % 8])  # same input, any slot of the batch -> same bytes
        assert 2.5 < n / r1[i].size < 4.5
    # the bench workload itself: three kernels on three HIP streams (default) against back-to-back launches,
    # and one of the streams against the oracle
    from speedy_amd._lib import lib
    lib().spx_set_concurrent(0)
    try:
        b.run()
        r3 = b.results()
    finally:
        lib().spx_set_concurrent(1)
    for i in range(B):
        assert np.array_equal(r1[i], r3[i]), i
    from oracle import pyorc
    for k in range(8):   # every distinct input of the batch against the oracle
        ref = pyorc.compress_sound(base[k], rate, 1, 3.5, 1.0, 0.0, False, chunk=1000)["out"]
        assert np.array_equal(r1[k], ref), k


def test_bench_streams_match_oracle(orc):
    """The bench workload's own inputs (bench.make_streams: 256 distinct streams x 10 s, seed = 1234 + global stream index):
    a sample of them against the oracle, so the number bench.py reports is for bytes the oracle agrees with (bench.py itself
    compares the CRC of every sampled stream of its cpu_baseline leg with the GPU's)."""
    import bench
    from speedy_amd.batch import Batch, Plan
    n = bench.RATE * bench.SECONDS
    streams = bench.make_streams(bench.STREAMS_PER_GPU, n, 0)
    b = Batch(Plan(bench.RATE, False), [n] * len(streams), 1, bench.SPEED, 1.0, 0.0)
    b.upload(streams)
    b.run()
    outs = b.results()
    for i in list(range(0, 32, 3)) + [32, 77, 130, 201, 255]:
        ref = orc.compress_sound(streams[i], bench.RATE, 1, bench.SPEED, 1.0, 0.0, False, taps=False)["out"]
        assert np.array_equal(outs[i], ref), i


def _config4_stream_cfg(i):
    """SURVEY.md 8(d), config 5 (= BASELINE configs[4]): rate by i%2, channels by (i/2)%2, speed by (i/4)%2."""
    return (16000 if i % 2 == 0 else 22050, 1 if (i // 2) % 2 == 0 else 2, 1.5 if (i // 4) % 2 == 0 else 3.5)


def _run_config4_shard(ids, cache, plans=None):
    """One rank's shard of configs[4]: the streams `ids` (global indices), 10 s each, both sample rates in ONE call
    (spx_batch_run_mixed: a plan per rate, every group of the batch launched together).  Returns {global index: int16 output}."""
    from speedy_amd.batch import MixedBatch, Plan
    from speedy_amd.synth import speech_like
    rates = (16000, 22050)
    plans = plans or [Plan(r, False) for r in rates]
    streams = []
    for i in ids:
        rate, ch, _ = _config4_stream_cfg(i)
        key = (rate, ch, i % 24)      # 24 distinct signals per kind; slots differ in the mix they sit in
        if key not in cache:
            cache[key] = speech_like(10 * rate, rate, seed=5000 + i % 24, channels=ch)
        streams.append(cache[key])
    b = MixedBatch(plans, [rates.index(_config4_stream_cfg(i)[0]) for i in ids], [10 * _config4_stream_cfg(i)[0] for i in ids],
                   [_config4_stream_cfg(i)[1] for i in ids], [_config4_stream_cfg(i)[2] for i in ids], 1.0, 0.0)
    b.upload(streams)
    b.run()
    first = dict(zip(ids, b.results()))
    b.run()                            # the same call again (warm: the other launch mode may be taken): same bytes
    for k, o in zip(ids, b.results()):
        assert np.array_equal(o, first[k]), k
    return first


def test_config4_one_gpu_shard_full_size(orc):
    """BASELINE configs[4], one GPU's shard at FULL size: 256 mixed streams x 10 s (16 k / 22.05 k, mono / stereo,
    1.5x / 3.5x, nonlinear).  Per-stream bytes identical for the partitions {256} and {128, 128}; at least 8 streams of
    each of the 8 kinds against the oracle."""
    import zlib
    cache = {}
    whole = _run_config4_shard(list(range(256)), cache)
    halves = {}
    halves.update(_run_config4_shard(list(range(0, 128)), cache))
    halves.update(_run_config4_shard(list(range(128, 256)), cache))
    assert sorted(whole) == list(range(256)) == sorted(halves)
    for i in range(256):
        assert zlib.crc32(whole[i].tobytes()) == zlib.crc32(halves[i].tobytes()), i
    checked = {}
    for i in range(256):
        rate, ch, speed = _config4_stream_cfg(i)
        kind = (rate, ch, speed)
        if checked.get(kind, 0) >= 8:
            continue
        x = cache[(rate, ch, i % 24)]
        ref = orc.compress_sound(x, rate, ch, speed, 1.0, 0.0, False, taps=False)["out"]
        assert np.array_equal(whole[i], ref), (i, kind)
        checked[kind] = checked.get(kind, 0) + 1
    assert len(checked) == 8 and all(v >= 8 for v in checked.values()), checked


def test_step_counts_equal_the_oracles(orc):
    """The walk kernels count their pitch searches (SpxWalkState::steps, the denominator of bench.py's latency roofline):
    the same number libsonic's findPitchPeriod runs in the oracle, on every kernel form -- 4 + 4 waves (one stream per CU),
    the throughput form (more than two streams per CU), multi-channel, the general kernel (slow-down)."""
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    L = orc.lib()

    def oracle_steps(x, rate, ch, speed, nl):
        h = L.orc_sonicCreateStream(rate, ch, 0)
        L.orc_sonicSetSpeed(h, speed); L.orc_sonicEnableNonlinearSpeedup(h, nl); L.orc_sonicSetDurationFeedbackStrength(h, 0.0)
        x = np.ascontiguousarray(x, np.int16)
        n = x.size // ch
        for pos in range(0, n, 1000):
            seg = np.ascontiguousarray(x[pos * ch:(pos + 1000) * ch])
            L.orc_sonicWriteShortToStream(h, orc.sptr(seg), seg.size // ch)
        L.orc_sonicFlushStream(h)
        k = int(L.orc_sonicIntStepCount(h))
        L.orc_sonicDestroyStream(h)
        return k
    for rate, ch, speed, nl, count in ((16000, 1, 3.5, 1.0, 12), (16000, 1, 3.5, 1.0, 600), (22050, 2, 1.5, 1.0, 9),
                                       (16000, 1, 0.6, 0.0, 5), (44100, 1, 2.0, 1.0, 4)):
        base = [speech_like(rate * 2, rate, seed=900 + i, channels=ch) for i in range(6)]
        b = Batch(Plan(rate, False), [rate * 2] * count, ch, speed, nl, 0.0)
        b.upload([base[i % 6] for i in range(count)])
        b.run()
        steps = b.step_counts()
        ref = [oracle_steps(x, rate, ch, speed, nl) for x in base]
        assert list(steps) == [ref[i % 6] for i in range(count)], (rate, ch, speed, nl, count, list(steps[:6]), ref)


def test_config4_whole_batch_on_one_gpu(orc):
    """ALL of BASELINE configs[4] on one GPU: the 2 048 mixed streams x 10 s (rate by i%2, channels by (i/2)%2,
    speed by (i/4)%2; 2 048 distinct signals, seed = 4000 + global index) run two ways -- ONE 2 048-stream
    spx_batch_run_mixed call (what N = 1 of the strong-scaling curve runs: two groups of 1 024 streams, the walk kernel's
    throughput form, pipelined time chunks) and the eight 256-stream calls the eight ranks of the 8-GPU layout make (one
    stream per CU, 4 + 4 waves) -- per-stream CRCs equal between the two, and equal to the oracle's on 8 streams of each of
    the 8 kinds IN EVERY SHARD (512 oracle runs on the host's threads)."""
    from speedy_amd import config4 as C4
    from speedy_amd.batch import Plan
    ids = list(range(C4.TOTAL_STREAMS))
    streams = C4.make_streams(ids)
    assert len({x.tobytes()[:4096] for x in streams}) == len(streams)        # distinct signals
    plans = [Plan(r, False) for r in C4.RATES]
    b = C4.mixed_batch(plans, ids, streams)
    b.run()
    whole = b.crcs()
    b.run()                       # again (warm): the same bytes
    assert b.crcs() == whole
    counts = b.counts()
    del b
    for r in range(8):
        sl = slice(256 * r, 256 * (r + 1))
        bs = C4.mixed_batch(plans, ids[sl], streams[sl])
        bs.run()
        assert bs.crcs() == whole[sl], r
        assert list(bs.counts()) == list(counts[sl]), r
        del bs
    checked = 0
    for k in range(8):            # one oracle pass per kind: 8 streams from every shard
        rate, ch, speed = C4.cfg(k)
        pick = [256 * r + k + 8 * j for r in range(8) for j in range(8)]
        assert all(C4.cfg(i) == (rate, ch, speed) for i in pick)
        _, frames, crcs = orc.crc_streams([streams[i] for i in pick], rate, ch, speed, 1.0, 0.0, False)
        for i, f, c in zip(pick, frames, crcs):
            assert counts[i] == f and whole[i] == c, (i, rate, ch, speed, int(counts[i]), f)
            checked += 1
    assert checked == 512


@pytest.mark.parametrize("chunks", [2, 5])
def test_time_chunk_pipelining_speed_up_batch(orc, chunks):
    """The same with every stream speeding up (the speed-up walk kernels, mono and multi-channel, carry their state from
    one time range to the next)."""
    from speedy_amd._lib import lib
    from speedy_amd.synth import speech_like
    for rate in (16000, 22050):
        chs = [1 + (i % 3) for i in range(6)]
        streams = [speech_like(30000 + 777 * i, rate, seed=40 + i, channels=chs[i]) for i in range(6)]
        speeds, nls = [3.5, 1.5, 2.0, 2.6, 3.5, 1.2], [1.0, 1.0, 0.0, 1.0, 1.0, 1.0]
        try:
            lib().spx_set_pipeline_chunks(chunks)
            outs, b = _batch(streams, rate, chs, speeds, nls, 0.0, False, taps=False)
        finally:
            lib().spx_set_pipeline_chunks(1)
        for i, x in enumerate(streams):
            ref = _oracle(orc, x, rate, chs[i], speeds[i], nls[i], 0.0, False)
            assert np.array_equal(outs[i], ref["out"]), (rate, i)


@pytest.mark.parametrize("chunks", [2, 5])
def test_time_chunk_pipelining_is_bit_exact(orc, chunks):
    """spx_set_pipeline_chunks: the batch call split into time ranges (analysis of range c+1 overlapping the walk of
    range c on a second HIP stream) must give the same bytes as the single-range call."""
    from speedy_amd._lib import lib
    from speedy_amd.synth import speech_like
    rate = 16000
    streams = [speech_like(30000 + 777 * i, rate, seed=40 + i, channels=1 + (i & 1)) for i in range(5)]
    chs = [1 + (i & 1) for i in range(5)]
    nls = [1.0, 1.0, 0.0, 1.0, 1.0]
    try:
        lib().spx_set_pipeline_chunks(chunks)
        outs, b = _batch(streams, rate, chs, [3.5, 1.5, 2.0, 0.6, 3.5], nls, 0.1, False, taps=False)
    finally:
        lib().spx_set_pipeline_chunks(1)
    for i, x in enumerate(streams):
        ref = _oracle(orc, x, rate, chs[i], [3.5, 1.5, 2.0, 0.6, 3.5][i], nls[i], 0.1, False)
        assert np.array_equal(outs[i], ref["out"]), i


@pytest.mark.parametrize("speed,nl", [(2.0, 0.0), (3.5, 1.0)])
def test_baseline_single_stream_60s(orc, speed, nl):
    """BASELINE configs[1] (60 s, 2.0x linear: TSM only) and configs[2] (60 s, 3.5x nonlinear: full path), one stream."""
    from speedy_amd.synth import speech_like
    rate, n = 16000, 960000
    x = speech_like(n, rate, seed=77)
    outs, b = _batch([x], rate, 1, speed, nl, 0.0, False, taps=(nl != 0))
    ref = _oracle(orc, x, rate, 1, speed, nl, 0.0, False)
    if nl:
        taps = b.tap_arrays(0)
        assert taps["tension"].shape == ref["tension"].shape
        assert np.array_equal(taps["tension"], ref["tension"])
        assert np.array_equal(taps["speed"], ref["speed"])
    assert np.array_equal(outs[0], ref["out"])
    assert abs(n / outs[0].size - speed) < 0.15 * speed


def test_baseline_mixed_batch_shape(orc):
    """BASELINE configs[4] in miniature: stream i has rate 16000 if i even else 22050, channels 1 if (i/2) even
    else 2, speed 1.5 if (i/4) even else 3.5, nonlinear 1.  One plan (one launch pair) per rate."""
    from speedy_amd.synth import speech_like
    n_streams = 16
    cfg = [(16000 if i % 2 == 0 else 22050, 1 if (i // 2) % 2 == 0 else 2, 1.5 if (i // 4) % 2 == 0 else 3.5)
           for i in range(n_streams)]
    for rate in (16000, 22050):
        idx = [i for i in range(n_streams) if cfg[i][0] == rate]
        streams = [speech_like(3 * rate, rate, seed=200 + i, channels=cfg[i][1]) for i in idx]
        outs, b = _batch(streams, rate, [cfg[i][1] for i in idx], [cfg[i][2] for i in idx], 1.0, 0.0, False, taps=False)
        for k, i in enumerate(idx):
            ref = _oracle(orc, streams[k], rate, cfg[i][1], cfg[i][2], 1.0, 0.0, False)
            assert np.array_equal(outs[k], ref["out"]), (rate, i)


def test_streaming_random_chunking(orc):
    """Ragged write sizes (1 .. 4000 frames, some empty) through the streaming API: same bytes as one big write."""
    from speedy_amd.sonic2 import SonicStream
    from speedy_amd.synth import speech_like
    rate = 16000
    x = speech_like(60000, rate, seed=5, channels=2)
    ref = _oracle(orc, x, rate, 2, 3.5, 1.0, 0.1, False)["out"]
    rng = np.random.default_rng(3)
    s = SonicStream(rate, 2, False)
    s.set_speed(3.5)
    s.enable_nonlinear(1.0)
    s.set_feedback(0.1)
    pos, n, outs = 0, x.size // 2, []
    while pos < n:
        k = int(rng.choice([0, 1, 7, 160, 161, 999, 4000]))
        seg = x[pos * 2:(pos + k) * 2]
        assert s.write_short(seg) == 1
        pos += seg.size // 2
        if rng.random() < 0.5:
            outs.append(s.read_short(int(rng.integers(1, 3000))))
    s.flush()
    while True:
        got = s.read_short(1000)
        if got.size == 0:
            break
        outs.append(got)
    s.close()
    assert np.array_equal(np.concatenate(outs), ref)


def test_concurrent_handoff_under_uneven_load():
    """spx_set_concurrent: the analysis kernel publishes per-tile flags (agent-scope release) while the walk kernel
    consumes them on another HIP stream (relaxed poll + agent-scope acquire).  Uneven load on purpose: 200 streams of
    0.1 .. 6 s, repeated launches into the same buffers (consumer caches warm); every output sample and every
    tension / speed value must equal the sequential two-launch mode."""
    from speedy_amd._lib import lib
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate = 16000
    rng = np.random.default_rng(11)
    lens = [int(v) for v in rng.integers(1600, 96000, 200)]
    base = [speech_like(96000, rate, seed=300 + i) for i in range(10)]
    streams = [base[i % 10][: lens[i]] for i in range(200)]
    plan = Plan(rate, False)
    L = lib()

    def run(concurrent, reps):
        L.spx_set_concurrent(int(concurrent))
        try:
            b = Batch(plan, lens, 1, 3.5, 1.0, 0.1, taps=True)
            b.upload(streams)
            outs = None
            for _ in range(reps):
                b.run()
                outs = b.results()
            taps = [b.tap_arrays(i) for i in range(0, 200, 7)]
            return outs, taps
        finally:
            L.spx_set_concurrent(1)

    ref_out, ref_taps = run(False, 1)
    for reps in (1, 3):
        out, taps = run(True, reps)
        for i in range(200):
            assert np.array_equal(out[i], ref_out[i]), (reps, i)
        for a, r in zip(taps, ref_taps):
            assert np.array_equal(a["tension"], r["tension"]) and np.array_equal(a["speed"], r["speed"])


def test_many_streams_fall_back_to_sequential_launches(orc):
    """More streams than the chip can keep resident next to the producer kernels (600 short streams, default =
    concurrent mode requested): the engine must run the three kernels back to back instead of letting spinning consumer
    workgroups starve the producers.  Checked against the oracle on a sample of the streams."""
    from speedy_amd.synth import speech_like
    rate = 16000
    base = [speech_like(12000 + 500 * i, rate, seed=700 + i) for i in range(12)]
    streams = [base[i % 12] for i in range(600)]
    outs, b = _batch(streams, rate, 1, 3.5, 1.0, 0.0, False, taps=False)
    refs = [_oracle(orc, x, rate, 1, 3.5, 1.0, 0.0, False)["out"] for x in base]
    for i in range(600):
        assert np.array_equal(outs[i], refs[i % 12]), i


def test_pack_outputs_matches_per_stream_results():
    """spx_batch_pack_outputs: the device-side gather of all produced frames (one D2H for a whole batch)."""
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate = 16000
    chs = [1, 2, 1, 3, 1]
    lens = [0, 5000, 20000, 7001, 300]
    streams = [speech_like(n, rate, seed=40 + i, channels=c) for i, (n, c) in enumerate(zip(lens, chs))]
    b = Batch(Plan(rate, False), lens, chs, [3.5, 1.5, 2.0, 3.5, 3.5], 1.0, 0.0)
    b.upload(streams)
    b.run()
    packed, offs = b.pack_outputs()
    outs = b.results()
    offs = offs.cpu().numpy()
    packed = packed.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == sum(o.size for o in outs)
    for i, o in enumerate(outs):
        assert offs[i + 1] - offs[i] == o.size
        assert np.array_equal(packed[offs[i]:offs[i + 1]], o), i


def test_ten_minute_stream(orc):
    """One 10-minute stream (9.6 M frames, 60 000 analysis frames): large positions, many window refills, the
    hand-off between the three kernels over a long run.  Bit-exact output and speed taps against the oracle."""
    from speedy_amd.synth import speech_like
    rate = 16000
    piece = speech_like(60 * rate, rate, seed=77)
    x = np.tile(piece, 10)
    outs, b = _batch([x], rate, 1, 3.5, 1.0, 0.1, False, taps=True)
    ref = _oracle(orc, x, rate, 1, 3.5, 1.0, 0.1, False)
    assert np.array_equal(b.tap_arrays(0)["speed"], ref["speed"])
    assert np.array_equal(outs[0], ref["out"])


def test_thousand_tiny_streams(orc):
    """1000 streams of 0 .. 0.25 s in one call (the large-batch path: four pipelined time chunks, most of them empty
    for streams this short), every stream against the oracle."""
    from speedy_amd.synth import speech_like
    rate = 16000
    rng = np.random.default_rng(5)
    lens = [int(v) for v in rng.integers(0, 4000, 1000)]
    base = speech_like(4000, rate, seed=123)
    streams = [base[:n] for n in lens]
    outs, b = _batch(streams, rate, 1, 3.5, 1.0, 0.0, False, taps=False)
    cache = {}
    for i, n in enumerate(lens):
        if n not in cache:
            cache[n] = _oracle(orc, streams[i], rate, 1, 3.5, 1.0, 0.0, False)["out"]
        assert np.array_equal(outs[i], cache[n]), (i, n)


def test_jobs_outside_the_defined_ranges_are_refused():
    """The reference stores any float (soniclib.c:177-182,555-570) and has no defined behaviour for most of them -- a speed
    <= 0 gives the TSM stage negative step counts.  spx_batch_run refuses such jobs before anything is launched."""
    from speedy_amd.batch import Batch, Plan
    plan = Plan(16000, False)
    x = np.zeros(4000, np.int16)
    nan, inf = float("nan"), float("inf")
    bad = [("speed", 0.0), ("speed", -1.0), ("speed", nan), ("speed", inf), ("nonlinear", -0.1), ("nonlinear", 1.5),
           ("nonlinear", nan), ("feedback", nan), ("feedback", inf), ("channels", 0), ("n_in", -1), ("in_off", -8),
           ("out_off", -8), ("out_cap", -1)]
    for field, value in bad:
        b = Batch(plan, [x.size], 1, 2.0, 1.0, 0.0)
        b.upload([x])
        setattr(b.jobs[0], field, value)
        with pytest.raises(RuntimeError):
            b.run()
    b = Batch(plan, [x.size], 1, 2.0, 1.0, 0.0)      # and the batch object is still good for a valid job afterwards
    b.upload([x])
    b.run()
    assert b.results()[0].size > 0
    # too many channels for one CU's LDS window (the general kernel keeps every channel of the window in LDS)
    b = Batch(plan, [64], 1, 2.0, 0.0, 0.0)
    b.jobs[0].channels = 400
    with pytest.raises(RuntimeError):
        b.run()


@pytest.mark.parametrize("rate,ch", [(16000, 64), (22050, 40), (22050, 64), (44100, 40), (8000, 100), (48000, 24)])
def test_many_channels(orc, rate, ch):
    """The general walk kernel keeps every channel of its window in LDS and shortens the window as channels grow -- never
    below what one pitch search needs (a 22.05 kHz stream with 40 channels once got a window shorter than that)."""
    from speedy_amd.batch import compress_batch
    from speedy_amd.synth import speech_like
    x = speech_like(int(0.6 * rate), rate, seed=ch, channels=ch)
    for speed, nl in ((2.0, 1.0), (0.7, 0.0), (3.5, 1.0)):
        ref = orc.compress_sound(x, rate, ch, speed, nl, 0.0, False, chunk=1000 if nl else x.size // ch)
        outs, _ = compress_batch([x], rate, ch, speed, nl, 0.0, False)
        assert np.array_equal(outs[0], ref["out"]), (rate, ch, speed, nl)


@pytest.mark.parametrize("rate", [1000, 3999, 4001, 7919, 12345, 24000, 37800, 50000, 60000, 62000, 88200, 96000, 127999])
def test_unusual_sample_rates(orc, rate):
    """Rates nobody tunes for: below the 4 kHz decimation threshold (skip = 1), prime window lengths (generic and Rader
    DFT stages), the range above 49 kHz where the plan falls back to the 8-frame analysis tile, and (round 3) the range
    above 61 kHz -- the reference takes any rate, speedy.c:213 -- where two waves, then one, transform 4-frame tiles.
    Mono and stereo, linear and nonlinear, taps included."""
    from speedy_amd.batch import compress_batch
    from speedy_amd.synth import speech_like
    for ch in (1, 2):
        x = speech_like(int(1.0 * rate), rate, seed=rate % 97, channels=ch)
        for speed, nl in ((2.0, 1.0), (0.7, 0.0), (3.5, 1.0), (1.3, 0.0)):
            ref = orc.compress_sound(x, rate, ch, speed, nl, 0.0, False, chunk=1000 if nl else x.size // ch)
            outs, b = compress_batch([x], rate, ch, speed, nl, 0.0, False, taps=(nl != 0))
            assert np.array_equal(outs[0], ref["out"]), (rate, ch, speed, nl)
            if nl:
                t = b.tap_arrays(0)
                for key in ("tension", "speed", "features"):
                    assert np.array_equal(t[key], ref[key]), (rate, ch, speed, nl, key)


def test_sample_rates_outside_the_supported_range_fail_loudly(orc):
    """Below 1 kHz and from 128 kHz on there is no plan (the walk kernels hold at most 256 lags per search).  Everything
    in between runs in both modes, through the streaming API too (96 kHz here: the one-stream analysis tile of four frames)."""
    from speedy_amd.batch import Plan
    from speedy_amd.sonic2 import time_compress
    from speedy_amd.synth import speech_like
    for rate in (999, 128000):
        with pytest.raises(RuntimeError):
            Plan(rate, False)
    rate = 96000
    x = speech_like(rate // 2, rate, seed=3)
    for speed, nl in ((2.0, 1.0), (2.0, 0.0)):
        ref = orc.compress_sound(x, rate, 1, speed, nl, 0.0, False, chunk=1000 if nl else x.size, taps=False)["out"]
        for coalesce in (True, False):
            got = time_compress(x, rate, 1, speed, nl, feedback=0.0, chunk=3000, coalesce=coalesce)
            assert np.array_equal(got, ref), (speed, nl, coalesce)


@pytest.mark.parametrize("rate,n_streams,multi,slow", [(16000, 700, False, False), (22050, 400, True, False),
                                                       (16000, 350, True, True)])
def test_large_ragged_batches(orc, rate, n_streams, multi, slow):
    """Hundreds of streams of 0 .. 2.5 s in one call -- the large-batch path (pipelined time chunks, streams that end
    in different chunks, the sequential fallback when the co-residency bound says so) -- every stream against the oracle."""
    from speedy_amd.batch import compress_batch
    from speedy_amd.synth import speech_like
    rng = np.random.default_rng(rate + n_streams)
    chs, speeds, nls, xs = [], [], [], []
    for i in range(n_streams):
        ch = int(rng.choice([1, 1, 2, 3])) if multi else 1
        n = int(rng.integers(0, int(2.5 * rate))) if i % 7 else int(rng.integers(0, 300))
        sp_ = float(np.round(rng.uniform(0.5, 0.95) if (slow and i % 5 == 0) else rng.uniform(1.05, 4.5), 3))
        chs.append(ch); speeds.append(sp_)
        nls.append(float(rng.choice([0.0, 1.0, 1.0, 1.0])))
        xs.append(speech_like(n, rate, seed=1000 + i, channels=ch))
    outs, _ = compress_batch(xs, rate, chs, speeds, nls, 0.0, False)
    for i in range(n_streams):
        ref = orc.compress_sound(xs[i], rate, chs[i], speeds[i], nls[i], 0.0, False,
                                 chunk=1000 if nls[i] != 0 else max(xs[i].size // chs[i], 1), taps=False)
        assert np.array_equal(outs[i], ref["out"]), (i, chs[i], xs[i].size // chs[i], speeds[i], nls[i])


@pytest.mark.parametrize("ch,speed,nl,n_streams", [(1, 3.5, 1.0, 3), (2, 1.5, 1.0, 3), (1, 0.6, 0.0, 3), (1, 2.0, 0.0, 300)])
def test_output_capacity_too_small_is_reported_and_contained(orc, ch, speed, nl, n_streams):
    """A job whose out_cap is smaller than what the stream produces: its n_out comes back NEGATIVE (minus the frames it
    would have produced), the frames that fitted are the oracle's first out_cap frames, nothing is written past its
    region (the neighbour's output is intact) and spx_batch_pack_outputs gathers no more than the capacity."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    rate = 16000
    plan = Plan(rate, False)
    lens = [12000 + 500 * (i % 5) for i in range(n_streams)]
    xs = [speech_like(n, rate, seed=90 + i, channels=ch) for i, n in enumerate(lens)]
    refs = [orc.compress_sound(x, rate, ch, speed, nl, 0.0, False, chunk=1000 if nl else x.size // ch, taps=False)["out"]
            for x in xs]
    b = Batch(plan, lens, ch, speed, nl, 0.0)
    b.upload(xs)
    victim = 1
    cap = refs[victim].size // ch // 3
    b.jobs[victim].out_cap = cap                      # the region stays as large as before; only the declared capacity shrinks
    b.d_out.fill_(12345)
    b.run()
    with pytest.raises(RuntimeError):
        b.results()
    nout = b.d_nout.cpu().numpy()
    assert nout[victim] == -(refs[victim].size // ch)
    out = b.d_out.cpu().numpy()
    o0 = b.out_offs[victim]
    assert np.array_equal(out[o0:o0 + cap * ch], refs[victim][:cap * ch])
    assert (out[o0 + cap * ch:b.out_offs[victim + 1]] == 12345).all()            # nothing past the declared capacity
    for i in range(n_streams):
        if i != victim:
            assert nout[i] == refs[i].size // ch
            assert np.array_equal(out[b.out_offs[i]:b.out_offs[i] + refs[i].size], refs[i]), i
    packed, offs = b.pack_outputs()
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    assert offs[victim + 1] - offs[victim] == cap * ch
    assert np.array_equal(packed.cpu().numpy()[offs[victim]:offs[victim + 1]], refs[victim][:cap * ch])


def test_walk_form_follows_the_co_residency_arithmetic(orc):
    """Which form of the walk kernel a batch gets (spx_engine.hip): 16 kHz mono keeps its 4 search + 4 output
    waves in the concurrent mode; 22.05 kHz mono gives up the output waves there (the lean form: 128 + 48 + 2 x 168 registers
    fit a SIMD, 2 x 112 + 48 + 2 x 168 do not) -- and still matches the oracle; 22.05 kHz stereo keeps them (its cross-fades
    read the input from HBM) and runs in sequence."""
    import torch
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    L = Plan(16000, False).L
    forms = {}
    for rate, ch in ((16000, 1), (22050, 1), (22050, 2)):
        n = rate * 2
        plan = Plan(rate, False)
        xs = [speech_like(n, rate, seed=900 + i, channels=ch) for i in range(6)]
        b = Batch(plan, [n] * 6, ch, 3.5, 1.0, 0.0)
        b.upload(xs)
        b.run()
        outs = b.results()
        forms[(rate, ch)] = L.spx_debug_last_walk_form()
        if (rate, ch) == (16000, 1) and not L.spx_debug_last_call_concurrent():
            pytest.skip("the concurrent mode is not available to this process (another process holds the device's lock?)")
        for x, got in zip(xs, outs):
            ref = orc.compress_sound(x, rate, ch, 3.5, 1.0, 0.0, False, chunk=n, taps=False)["out"]
            assert np.array_equal(got, ref)
    if os.environ.get("SPX_NO_LEAN_WALK") or os.environ.get("SPX_SERIAL") or os.environ.get("SPX_SHARED_GPU"):
        pytest.skip("a tuning variable overrides the launch mode: forms %r" % forms)
    assert forms[(16000, 1)] == 16 * 4 + 4, forms
    assert forms[(22050, 1)] == 16 * 4 + 0, forms
    assert forms[(22050, 2)] == 16 * 4 + 4, forms


@pytest.mark.parametrize("rate,ch", [(16000, 1), (22050, 2), (44100, 1)])
def test_slow_down_batches_run_on_the_speed_up_kernel(orc, rate, ch):
    """Round 5: a batch with slow-down jobs (libsonic's insertPitchPeriod) no longer falls back to the general walk kernel -- the
    speed-up kernel's instantiations with the slow-down event serve it (spx_walk_fast.hip, MC + 2).  Speeds below 0.5, between 0.5
    and 1, unity, above 1 and above 2 in ONE batch, linear and nonlinear, a speed so low that steps fail: bit-equal to the oracle,
    and the form of the kernel that ran is the fast one."""
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    plan = Plan(rate, False)
    speeds = [0.3, 0.5, 0.75, 0.99, 1.0, 1.3, 2.0, 3.5, 0.02, 0.45, 0.6, 0.9]
    nls = [1.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.3, 0.0]
    n = int(rate * 1.2)
    xs = [speech_like(n, rate, seed=300 + i, channels=ch) for i in range(len(speeds))]
    b = Batch(plan, [n] * len(speeds), ch, speeds, nls, 0.0)
    b.upload(xs)
    b.run()
    outs = b.results()
    assert plan.L.spx_debug_last_walk_form() != 0, "a slow-down batch ran on the general kernel"
    for i, (x, got) in enumerate(zip(xs, outs)):
        ref = orc.compress_sound(x, rate, ch, speeds[i], nls[i], 0.0, False, chunk=n, taps=False)["out"]
        assert np.array_equal(got, ref), (i, speeds[i], nls[i], got.size, ref.size)


@pytest.mark.parametrize("rate,ch", [(11025, 1), (11025, 2), (9800, 1), (11900, 1), (15000, 1), (15999, 2)])
def test_rates_with_more_than_64_coarse_lags_run_on_the_speed_up_kernel(orc, rate, ch):
    """Round 5: the rates whose coarse pitch search has more than 64 lags (about 9.8 - 12 kHz at skip 2 -- 11.025 kHz has 72 -- and
    14.7 - 16 kHz at skip 3) leave the general walk kernel: two coarse lags per lane in the coarse select (spx_walk_fast.hip WIDEC,
    SPEC = 2).  Speed-up and slow-down jobs, linear and nonlinear, bit-equal to the oracle; the form that ran is the fast one
    wherever the coarse triangle fits the search lanes (11.9 kHz: 558 groups of pairs for 512 lane slots -- the general kernel)."""
    from speedy_amd.batch import Batch, Plan
    from speedy_amd.synth import speech_like
    plan = Plan(rate, False)
    n = int(rate * 1.5)
    for speeds, nls in (([3.5, 2.0, 1.3, 2.7, 3.5], [1.0, 0.0, 1.0, 0.3, 0.0]), ([0.4, 3.5, 0.8, 1.0, 1.7], [1.0, 1.0, 0.0, 0.0, 1.0])):
        xs = [speech_like(n + 53 * i, rate, seed=700 + i, channels=ch) for i in range(len(speeds))]
        b = Batch(plan, [x.size // ch for x in xs], ch, speeds, nls, 0.0)
        b.upload(xs)
        b.run()
        outs = b.results()
        form = plan.L.spx_debug_last_walk_form()
        if rate == 11900:
            assert form == 0
        else:
            assert form in (16 * 4 + 4, 16 * 4 + 0), (rate, ch, form)
        for i, (x, got) in enumerate(zip(xs, outs)):
            ref = orc.compress_sound(x, rate, ch, speeds[i], nls[i], 0.0, False, chunk=x.size // ch, taps=False)["out"]
            assert np.array_equal(got, ref), (rate, ch, i, speeds[i], nls[i], got.size, ref.size)


def test_dft_spec_v2_on_frames_sensitive_to_the_radix_11_fusion(orc):
    """The 22.05 kHz window is 330 = 2 x 3 x 5 x 11 points; DFT spec v2 fuses the multiply-adds of the odd-prime butterfly (oracle
    orc_butterfly_v2, kernel spx_acc).  Until the end of round 5 the kernel's hand-written radix-11 stage had kept the unfused sums:
    an fp64 last-bit difference that a float magnitude shows about once in 1e9 values -- tools/r11_probe.py found six such frames
    in ten million (profiles/r05/r5ah_r11_probe.txt), and these are they: the spectrogram tap of each against the oracle's."""
    _sensitive_frames(orc, 22050, [(116, 202, 96), (175, 40, 76), (188, 224, 41), (339, 70, 3), (374, 1, 62), (384, 228, 40)])


def test_twiddle_tables_are_the_oracles_on_frames_that_told_them_apart(orc):
    """The same probe, one level deeper (frames sensitive to the spec VERSION, the oracle on both specs), showed the GPU's transform
    and the oracle's differing at the fp64 level at 32 - 48 kHz in both specs: the twiddle tables.  Both sides wrote cos(a), -sin(a);
    gcc merges the pair into glibc's sincos, clang -- the library's host compiler -- does not, and sincos rounds a few entries
    differently in the last bit.  A twiddle is now one explicit sincos call on both sides (DFT spec).  These are the two
    48 kHz frames of that probe whose magnitudes were one float ulp off the oracle's."""
    _sensitive_frames(orc, 48000, [(13, 255, 23), (30, 25, 42)])


@pytest.mark.parametrize("rate,batches", [(16000, 40), (22050, 40), (44100, 12), (48000, 12), (11025, 12)])
def test_every_frame_against_the_oracle_at_scale(rate, batches):
    """The library against the ORACLE ITSELF on every frame of batches x 256 one-second noise streams: the CPU port with hashing
    callbacks (oracle/orc_bench.c orc_bench_run_hashed, every host core) hands back one hash per spectrogram row and one per tension
    frame (15 features + tension + speed), and so does the GPU from its taps -- a million frames per rate here (tools/r11_probe.py
    oracle 400: ten million per rate, profiles/r05/r5ah_r11_probe.txt).  What a float hides from a test of a few thousand frames --
    an fp64 last-bit difference shows once in 1e9 magnitudes -- it does not hide from this one: the same comparison found the two
    defects of round 5 (an unfused stage, the twiddle tables) at six and fifteen frames in ten million."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import r11_probe
    bad_spec, n_spec, bad_tap, n_tap = r11_probe.against_the_oracle(rate, batches, verbose=False)
    assert n_spec >= batches * 256 * 90 and n_tap >= batches * 256 * 70, (n_spec, n_tap)
    assert bad_spec == 0 and bad_tap == 0, (rate, bad_spec, n_spec, bad_tap, n_tap)


@pytest.mark.parametrize("rate", [16000, 22050])
def test_output_audio_against_the_oracle_at_scale(rate):
    """... and the audio: 2 048 two-second noise streams per rate (eight batches cycling through mono / stereo, speeds 0.4 ... 5.5,
    linear / nonlinear, duration feedback), CRC-32 per stream against the CPU port's (tools/r11_probe.py audio 80: 20 480 streams at
    each of four rates, none differs)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import r11_probe
    bad, total = r11_probe.audio_against_the_oracle(rate, 8, verbose=False)
    assert total == 2048 and bad == 0, (rate, bad, total)


def _sensitive_frames(orc, rate, cases):
    from speedy_amd.batch import Batch, Plan
    xs = [np.random.default_rng([seed, i]).integers(-20000, 20000, size=rate).astype(np.int16) for seed, i, _ in cases]
    plan = Plan(rate, False)
    b = Batch(plan, [rate] * len(xs), 1, 3.0, 1.0, 0.0, taps=True, spectrogram_taps=True)
    b.upload(xs)
    b.run()
    L = orc.lib()
    for k, ((seed, i, f), x) in enumerate(zip(cases, xs)):
        rows = []
        h = L.orc_sonicCreateStream(rate, 1, 0)
        nb = L.orc_sonicSpectrogramSize(h)
        cb = orc.FEATURES_FN(lambda s, t, p: rows.append(np.ctypeslib.as_array(p, shape=(nb,)).copy()))
        L.orc_sonicSpectrogramCallback(h, cb)
        L.orc_sonicSetSpeed(h, 3.0)
        L.orc_sonicEnableNonlinearSpeedup(h, 1.0)
        L.orc_sonicWriteShortToStream(h, orc.sptr(x), x.size)
        L.orc_sonicDestroyStream(h)
        got = b.tap_arrays(k)["spectrogram"]
        ref = np.array(rows, np.float32)
        assert got.shape == ref.shape
        assert np.array_equal(got[f], ref[f]), (seed, i, f, int((got[f] != ref[f]).sum()))
        assert np.array_equal(got, ref), (seed, i)


def test_register_budgets_of_the_concurrent_mode():
    """The concurrent mode needs two analysis waves beside a stream's walk and tension waves on a SIMD's 512 registers
    (DESIGN.md 2).  Both cases that matter are tight: 16 kHz mono 2 x 96 + tension + 2 x 128, and 22.05 kHz mono with the lean
    walk form 128 + 48 + 2 x 168 = 512 exactly -- one register more in the tension kernel (allocated in eights) silently
    sends 22.05 kHz batches back to running their kernels in sequence (it happened in round 3)."""
    from speedy_amd._lib import lib
    L = lib()
    v = {k: L.spx_debug_kernel_vgprs(i) for i, k in enumerate(["tension", "walk16", "lean22", "analysis16", "analysis22", "walk16mc"])}
    assert all(x > 0 for x in v.values()), v
    assert v["tension"] <= 48, v
    assert 2 * v["walk16"] + v["tension"] + 2 * v["analysis16"] <= 512, v
    assert v["lean22"] + v["tension"] + 2 * v["analysis22"] <= 512, v
    assert 2 * v["walk16mc"] + v["tension"] + 2 * v["analysis16"] <= 512, v


def test_kernel_resources_of_every_form_the_engine_selects():
    """Round 4: the register allocation, the spilled bytes and the form of the walk kernel for EVERY batch shape spx_walk_config
    can pick at the rates of the BASELINE configs (and one rate of each other kernel family), and of the analysis kernels --
    pinned.  The engine's choice between the concurrent and the sequential launch order is arithmetic over these numbers
    (DESIGN.md 2), a SIMD's 512 registers leave no slack, and a change that costs a kernel a register or makes it spill has so
    far only ever been noticed by accident.  Allocated VGPRs and scratch bytes may go DOWN without touching this table."""
    import ctypes as C
    from speedy_amd._lib import lib
    L = lib()
    # (rate, channels, streams, short_jobs, lean) -> (form, allocated VGPRs <=, scratch bytes <=)
    table = {
        (16000, 1, 256, 0, 0): (16 * 4 + 4, 96, 60),      # spx_walk_fast_kernel<4, 4, 16000, 1, 0>: the bench's
        (16000, 2, 256, 0, 0): (16 * 4 + 4, 96, 136),     # <4, 4, 16000, 1, 1>
        (22050, 1, 256, 0, 0): (16 * 4 + 4, 120, 0),      # <4, 4, 22050, 1, 0>
        (22050, 1, 256, 0, 1): (16 * 4 + 0, 128, 12),      # <4, 0, 22050, 0, 0>: the lean form of the concurrent mode
        (22050, 2, 256, 0, 0): (16 * 4 + 4, 128, 36),     # <4, 4, 22050, 1, 1> (36 under the max-memory-clause scheduling of round 5, 28 before)
        (16000, 1, 2048, 0, 0): (16 * 2 + 0, 128, 28),    # <2, 0, 16000, 0, 0>: the throughput form
        (16000, 2, 2048, 0, 0): (16 * 2 + 0, 128, 108),
        (22050, 1, 2048, 0, 0): (16 * 2 + 0, 128, 80),
        (22050, 2, 2048, 0, 0): (16 * 2 + 0, 128, 152),
        (16000, 1, 1024, 1, 0): (16 * 4 + 0, 128, 0),     # <4, 0, 16000, 0, 0>: short (coalesced sonic2.h) jobs beyond one stream per CU
        (8000, 1, 256, 0, 0): (16 * 4 + 4, 128, 28),      # <4, 4, 0, 0, 0>: the plan-driven instantiation
        (24000, 1, 256, 0, 0): (16 * 8 + 4, 168, 0),      # <8, 4, 0, 0, 0>: rates whose ragged tasks need eight search waves (three waves per SIMD: 168)
        (44100, 1, 256, 0, 0): (16 * 8 + 4, 168, 0),      # ... and, since round 4, the rates with more than 64 refine lags
        (48000, 2, 256, 0, 0): (16 * 8 + 4, 168, 0),      # <8, 4, 0, 0, 1>
        (96000, 1, 256, 0, 0): (0, 128, 0),               # spx_walk_kernel<8, 0>: the general kernel
    }
    out = (C.c_int * 5)()
    for (rate, ch, n, short, lean), (form, vg, sc) in table.items():
        assert L.spx_debug_walk_info(rate, ch, n, 1, short, lean, out) == 0
        got = list(out)
        assert got[3] == form, ((rate, ch, n, short, lean), got)
        assert 0 < got[0] <= vg and 0 <= got[1] <= sc, ((rate, ch, n, short, lean), got, (vg, sc))
    a = (C.c_int * 3)()
    # (22.05 kHz: since the fused transform of round 5 the 16-frame instantiation keeps seven dwords in scratch at its 168 registers --
    # three waves per SIMD -- and is 5 % faster all the same, profiles/r05/r5i_dft_ab.txt)
    for rate, vg, sc in ((16000, 128, 0), (22050, 168, 28), (44100, 256, 0), (48000, 256, 0), (32000, 160, 0), (24000, 136, 0), (8000, 88, 0),
                         (11025, 120, 0)):   # <= 256: two waves per SIMD
        assert L.spx_debug_analysis_info(rate, a) == 0
        assert 0 < a[0] <= vg and 0 <= a[1] <= sc, (rate, list(a))
    # the budgets of the concurrent mode, from the same source
    L.spx_debug_walk_info(16000, 1, 256, 1, 0, 0, out); w16 = out[0]
    L.spx_debug_walk_info(22050, 1, 256, 1, 0, 1, out); lean22 = out[0]
    L.spx_debug_analysis_info(16000, a); a16 = a[0]
    L.spx_debug_analysis_info(22050, a); a22 = a[0]
    ten = L.spx_debug_kernel_vgprs(0)
    assert 2 * w16 + ten + 2 * a16 <= 512 and lean22 + ten + 2 * a22 <= 512, (w16, lean22, a16, a22, ten)


def test_committed_kernel_resources_are_the_librarys():
    """profiles/kernel_resources.json -- what tests/test_mode_table.py replays the engine's launch-mode decision from on the CPU --
    against the library on this box (spx_debug_mode_resources: spx_walk_config, the LDS layouts, hipFuncGetAttributes).  A kernel
    change that moves a number must come with a refreshed file (python tools/kernel_resources.py), and the CPU table then shows
    whether a mode moved with it."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    want = json.load(open(os.path.join(ROOT, "profiles", "kernel_resources.json")))
    got = kernel_resources.collect()
    assert got["fields"] == want["fields"]
    assert got["shapes"] == want["shapes"], {k: (dict(zip(got["fields"], v)), dict(zip(got["fields"], want["shapes"].get(k, []))))
                                             for k, v in got["shapes"].items() if v != want["shapes"].get(k)}


def test_cross_fade_quotient_equals_the_integer_division():
    """The walk kernels' cross-fade divides d (n - t) + u t by n, truncating toward zero (libsonic overlapAdd).  Round 5 forms the
    quotient with a Newton reciprocal, one fused multiply-add with a signed 2^-20 and the truncating conversion -- in the lean walk
    form these instructions are on the chain.  Against the integer division for EVERY n up to 4096 and every numerator k n - 1,
    k n, k n + 1 with |k| <= 32768 (the cases a rounding error could flip), and the 24-bit numerator against the 32-bit one."""
    from speedy_amd._lib import lib
    assert lib().spx_debug_xfade_check(1, 4096) == 0
    assert lib().spx_debug_xfade_check(0, 10) == -1


def test_fast_division_equals_the_ieee_quotient():
    """The walk kernel forms the candidate step lengths n = (int)(period / (speed - 1)) with a per-event reciprocal and two
    correction rounds -- the IEEE division sequence without its scaling and fix-up halves, which are no-ops for these operands
    (spx_walk_fast.hip fast_div).  Bit-equality with the `/` operator over 4 M random speeds x every count up to 4096 x both
    numerator forms of a step (3.4e10 divisions)."""
    from speedy_amd._lib import lib
    assert lib().spx_debug_fdiv_check(20261003, 1 << 22, -17, 8) == 0      # speeds 1.00001 .. 257: everything a test uses
    assert lib().spx_debug_fdiv_check(7, 1 << 20, -60, 80) == 0             # ... and the whole range the sequence is exact on
    # (speeds of 1e18 and more -- SPX_FAST_MAX_SPEED, far inside it -- run on the general kernel, which divides the IEEE way)
    from speedy_amd.batch import compress_batch
    from speedy_amd.synth import speech_like
    from oracle import pyorc
    x = speech_like(16000, 16000, seed=77)
    for speed in (9.9e17, 3.0e18, 1.0e30):
        outs, _ = compress_batch([x], 16000, 1, speed, 0.0, 0.0, False)
        assert np.array_equal(outs[0], pyorc.compress_sound(x, 16000, 1, speed, 0.0, 0.0, False, chunk=x.size, taps=False)["out"]), speed


def test_log_spec_v2_gpu_equals_oracle_on_every_positive_normal_float(orc):
    """Log spec v2 on the GPU (spx_log.h, the table read from LDS as the analysis kernel reads it) against the oracle's
    orc_log_v2_f32 over ALL 2 130 706 432 positive normal floats -- every argument the kernel can ever meet: checksums of the
    results' bit patterns per block of 2^20 float patterns, 2 032 blocks, equal one by one (spx_debug_log_check against
    oracle/orc_logcheck.c)."""
    import ctypes as C
    from speedy_amd._lib import lib
    L = lib()
    gpu = (C.c_ulonglong * 2040)()
    assert L.spx_debug_log_check(8, 2040, gpu) == 0
    cpu = (C.c_uint64 * 2040)()
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    assert orc.lib().orc_logcheck_run(8, 2040, threads, 0, cpu, None, None) == 0
    bad = [b for b in range(8, 2040) if gpu[b] != cpu[b]]
    assert not bad, (len(bad), bad[:8])
    assert len({int(v) for v in gpu[8:2040]}) > 2000     # (the sums are not trivially equal)


def test_scale_free_division_and_square_root_equal_the_ieee_sequences():
    """The analysis kernel's fp32 ratio, fp64 log quotient and fp64 square root run the compiler's IEEE sequences without their
    scaling / fix-up halves (spx_log.h): 3 x 2^26 pseudo-random operands of the ranges the kernel feeds them, bit for bit."""
    from speedy_amd._lib import lib
    L = lib()
    for seed in (1, 2):
        assert L.spx_debug_arith_check(seed, 1 << 18, 256) == 0
