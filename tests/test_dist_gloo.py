"""N > 1 path on CPU: world_size-2 gloo.  Streams are sharded with no data-path collective; the handshake and the
result reduction are the only communication.  The per-rank compute stand-in is the CPU oracle (allowed in
tests); the property checked is the one the 1/2/4/8-GPU run relies on: per-stream output bytes do not depend on
the partition."""
import os
import sys
import zlib

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import pyorc
    from speedy_amd import dist as sd
    from speedy_amd.synth import speech_like
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sd.partition(n_streams, world, rank)
    crcs, frames = {}, 0
    for i in range(lo, hi):
        x = speech_like(8000 + 100 * i, 16000, seed=i)
        frames += x.size
        out = pyorc.compress_sound(x, 16000, 1, 3.5, 1.0, 0.0, False, taps=False)["out"]
        crcs[i] = zlib.crc32(out.tobytes())
    layout = sd.handshake(dist, hi - lo, frames)
    total, tmax = sd.reduce_totals(dist, frames, 0.001 * (rank + 1))
    q.put((rank, crcs, layout.tolist(), total, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_everything():
    from speedy_amd.dist import partition, partition_by_cost
    for n in (0, 1, 7, 256, 2048):
        for w in (1, 2, 4, 8):
            spans = [partition(n, w, r) for r in range(w)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))
    parts = partition_by_cost([5, 1, 9, 3, 7, 2], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 1, 9, 3, 7, 2][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 3


def test_two_rank_shard_matches_single_process():
    from oracle import pyorc
    from speedy_amd.synth import speech_like
    pyorc.build()
    n_streams, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = {}
    for rank, crcs, layout, total, tmax in res:
        got.update(crcs)
        assert layout == [[3, layout[0][1]], [3, layout[1][1]]]
        assert abs(tmax - 0.002) < 1e-9  # MAX over ranks
    expected_total = 0
    for i in range(n_streams):
        x = speech_like(8000 + 100 * i, 16000, seed=i)
        expected_total += x.size
        ref = pyorc.compress_sound(x, 16000, 1, 3.5, 1.0, 0.0, False, taps=False)["out"]
        assert got[i] == zlib.crc32(ref.tobytes())
    assert all(r[3] == expected_total for r in res)
