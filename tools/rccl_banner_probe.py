import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("PROBE-STDOUT-LINE")
dist.destroy_process_group()
