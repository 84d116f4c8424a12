import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda") * (dist.get_rank() + 1)
dist.all_reduce(t)
print("rank", dist.get_rank(), t.tolist(), flush=True)
dist.destroy_process_group()
