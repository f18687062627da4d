"""configs[4]'s correctness check for the N-rank path: per-stream output bytes must not depend on how
the streams are partitioned over ranks.  Runs `bench.py --gpus N --crc-out ...` for every N asked for (the streams of an
N-rank run are the first 256 N of ONE global sequence, seed = 1234 + global index) and compares the CRC-32 of every
global stream across the runs that contain it.

  python tools/check_scale.py --gpus 1,2,4,8                 # an 8-GPU node, RCCL
  python tools/check_scale.py --gpus 1,2 --backend gloo      # one GPU: the ranks share it, collectives over gloo

Exit code 0 and one JSON line when every stream agrees everywhere."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--config4", action="store_true", help="also run the configs[4] legs (weak shard + the fixed batch of "
                    "--total-streams mixed streams, strong scaling) and compare THEIR per-stream CRCs, keyed by global stream index")
    ap.add_argument("--total-streams", type=int, default=2048)
    a = ap.parse_args()
    counts = [int(v) for v in a.gpus.split(",")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    seen = {}      # global stream index -> (crc, first run that produced it)
    seen4 = {}     # the same for the configs[4] streams (their own global sequence)
    lines = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n in counts:
            crc = os.path.join(tmp, "crc%d" % n)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", "1",
                   "--no-pcie", "--no-cpu-baseline", "--no-api", "--no-large-batch", "--no-other-rates", "--crc-out", crc, "--backend", a.backend]
            cmd += ["--total-streams", str(a.total_streams)] if a.config4 else ["--no-config4"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600, env=env)
            if r.returncode != 0:
                sys.exit("bench.py --gpus %d failed (rc %d): %s" % (n, r.returncode, r.stderr[-1500:]))
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            lines[n] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "n_ranks_seen": line["config"]["n_ranks_seen"],
                        "backend": line["config"]["backend"],
                        "ranks": [r["rank"] for r in line["config"]["ranks"]],
                        "rank_numa": [[r.get("numa_node"), r.get("cpus_allowed"), r.get("bound_to_numa_node")] for r in line["config"]["ranks"]]}
            per = line["config"]["streams_per_gpu"]
            for rank in range(n):
                got = json.load(open("%s.rank%d.json" % (crc, rank)))
                assert len(got) == per, (n, rank, len(got))
                for i, c in enumerate(got):
                    g = rank * per + i
                    if g in seen and seen[g][0] != c:
                        sys.exit("stream %d: CRC %08x in the %d-rank run, %08x in the %d-rank run" % (g, c, n, seen[g][0], seen[g][1]))
                    seen.setdefault(g, (c, n))
                if a.config4:
                    for g, c in json.load(open("%s.c4.rank%d.json" % (crc, rank))).items():
                        if g in seen4 and seen4[g][0] != c:
                            sys.exit("configs[4] stream %s: CRC %08x in the %d-rank run, %08x in the %d-rank run" % (g, c, n, seen4[g][0], seen4[g][1]))
                        seen4.setdefault(g, (c, n))
            if a.config4:
                lines[n]["config4_full_ms"] = line["config4_full"]["ms_per_step"]
                lines[n]["config4_shard_ms"] = line["config4_shard"]["ms_per_step"]
    print(json.dumps({"ok": True, "runs": lines, "streams_checked": len(seen), "config4_streams_checked": len(seen4),
                      "note": "every global stream has the same output CRC-32 in every run that contains it"}))


if __name__ == "__main__":
    main()
