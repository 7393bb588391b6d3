import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
t = torch.ones(1, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print("nccl world", dist.get_world_size(), "ok", float(t))
dist.destroy_process_group()
