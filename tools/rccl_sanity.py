"""One-rank RCCL sanity on the GPU box: the collectives bench.py and speedy_amd/dist.py issue at N > 1 (init with a bound
device, all_gather of int64 on the device, float64 MAX all_reduce, barrier) run on the nccl (= RCCL) backend."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from speedy_amd.dist import handshake, reduce_totals
print("handshake", handshake(dist, 256, 256 * 160000, device="cuda").tolist())
print("totals", reduce_totals(dist, 40960000, 0.00223, device="cuda"))
t = torch.tensor([2.23], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
print("max", float(t.item()), "backend", dist.get_backend())
dist.destroy_process_group()
print("rccl sanity ok")
