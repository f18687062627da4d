"""N > 1 on real hardware with the HIP library as the compute: `python bench.py --gpus 2` starts two rank processes
itself (before anything touches the GPU); on a 1-GPU box both ranks share the device and the collectives run over gloo
(RCCL needs one device per rank).  Each rank drives libspeedy_hip.so on its own 256-stream shard at the same time as the
other; the per-stream output CRCs must equal those of solo runs of the same streams."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_same_bytes_as_solo(tmp_path):
    import bench
    from speedy_amd.batch import Batch, Plan
    crc = str(tmp_path / "crc")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-pcie", "--no-cpu-baseline", "--no-api", "--no-config4", "--no-large-batch", "--no-other-rates", "--crc-out", crc, "--backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["n_ranks_seen"] == 2 and line["config"]["backend"] == "gloo" and line["config"]["handshake_ms"] > 0
    assert [r["rank"] for r in line["config"]["ranks"]] == [0, 1] and all("device" in r for r in line["config"]["ranks"])
    # round 6: every rank says where it ran -- the NUMA node of its GPU, the CPUs it was left with, whether it was bound there
    for rk in line["config"]["ranks"]:
        assert {"numa_node", "cpus_allowed", "bound_to_numa_node", "pci", "h2d_alone_gbs"} <= set(rk), rk
        assert rk["cpus_allowed"] >= 1
    assert line["config"]["streams_per_gpu"] == bench.STREAMS_PER_GPU
    n = bench.RATE * bench.SECONDS
    plan = Plan(bench.RATE, False)
    for rank in (0, 1):
        got = json.load(open("%s.rank%d.json" % (crc, rank)))
        streams = bench.make_streams(bench.STREAMS_PER_GPU, n, rank)
        b = Batch(plan, [n] * len(streams), 1, bench.SPEED, 1.0, 0.0)
        b.upload(streams)
        b.run()
        solo = [zlib.crc32(np.ascontiguousarray(o).tobytes()) for o in b.results()]
        assert got == solo, rank


def test_rccl_collectives_of_the_n_rank_path_with_one_rank():
    """The lease has one GPU, so RCCL cannot carry two ranks here -- but the calls bench.py and speedy_amd/dist.py make
    at N > 1 (init with a bound device, device-side all_gather / all_reduce, barrier) do run on the nccl backend with a
    world of one (tools/rccl_sanity.py, its own process: a process group is per process)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_sanity.py")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and "rccl sanity ok" in r.stdout, (r.stdout[-1000:], r.stderr[-1000:])
    assert "[[256, 40960000]]" in r.stdout


def test_no_silent_downgrade_of_the_backend():
    """More ranks than GPUs with the RCCL backend is a mis-provisioned scaling run: bench.py refuses it (non-zero exit, a
    message) instead of quietly switching to gloo."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-pcie", "--no-cpu-baseline", "--no-api", "--no-config4", "--no-large-batch", "--no-other-rates"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "one device per rank" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_check_scale_partitions_agree():
    """tools/check_scale.py (the per-stream CRC comparison of 1 / 2 / 4 / 8-rank partitions) on what a 1-GPU box can run:
    1 and 2 ranks, the two ranks sharing the GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scale.py"), "--gpus", "1,2", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["streams_checked"] == 512 and out["runs"]["2"]["n_ranks_seen"] == 2


def test_config4_strong_scaling_partitions_agree():
    """BASELINE configs[4] as a fixed batch (bench.py --total-streams, the strong-scaling leg `config4_full`) on what a 1-GPU box
    can run: 512 mixed streams as ONE rank's single call and as two ranks' 256-stream calls (the ranks share the GPU, gloo) --
    every global stream the same output CRC in both runs, the weak leg's shards included."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scale.py"), "--gpus", "1,2", "--backend", "gloo",
                        "--config4", "--total-streams", "512"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["config4_streams_checked"] == 512 and out["runs"]["2"]["n_ranks_seen"] == 2


def test_eight_ranks_functionally_on_one_gpu():
    """The world size the target names, before a node exists (round-4 review): `bench.py --gpus 8` with eight self-spawned ranks
    SHARING this box's GPU over gloo -- port allocation, the per-device lock file, the handshake, the time-outs and the
    strong-scaling partition of configs[4] at N = 8 -- and tools/check_scale.py's CRC agreement between the 1-rank and the 8-rank
    runs: 2 048 headline streams and all 2 048 configs[4] streams (one call at N = 1, eight 256-stream shards at N = 8).  No
    scaling number is asked for: the ranks take turns on one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scale.py"), "--gpus", "1,8", "--backend", "gloo",
                        "--config4", "--total-streams", "2048", "--steps", "1"],
                       capture_output=True, text=True, timeout=3000, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["config4_streams_checked"] == 2048, out
    run8 = out["runs"]["8"]
    assert run8["n_ranks_seen"] == 8 and run8["backend"] == "gloo" and run8["ranks"] == list(range(8)), run8
    assert len(run8["rank_numa"]) == 8 and all(len(v) == 3 and v[1] >= 1 for v in run8["rank_numa"]), run8   # (round 6: where each rank ran)
