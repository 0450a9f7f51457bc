"""One-rank RCCL sanity on the box: the same init call bench.py makes for N > 1."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
t = torch.arange(8, dtype=torch.float32, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
print("rccl all_reduce ok:", t.tolist())
dist.destroy_process_group()
