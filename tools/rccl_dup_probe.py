"""Can two RCCL ranks share one GPU on this image?  (diagnostic: torchrun --nproc-per-node 2 tools/rccl_dup_probe.py)"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
x = torch.ones(4, device="cuda:0") * (dist.get_rank() + 1)
try:
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank %d: all_reduce over RCCL with both ranks on cuda:0 -> %s" % (dist.get_rank(), x.tolist()), flush=True)
except Exception as exc:        # noqa: BLE001
    print("rank %d: RCCL refuses: %s" % (dist.get_rank(), str(exc).splitlines()[0][:200]), flush=True)
