"""Does RCCL accept two ranks on ONE GPU (the test box has one)?  torchrun --nproc-per-node 2 this file."""
import os
import torch
import torch.distributed as dist

dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda") * (dist.get_rank() + 1)
try:
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ok", x.tolist(), flush=True)
except Exception as e:      # noqa: BLE001
    print("rank", dist.get_rank(), "all_reduce FAILED:", repr(e)[:300], flush=True)
dist.destroy_process_group()
