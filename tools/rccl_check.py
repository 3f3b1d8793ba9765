import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(8, device=dev, dtype=torch.int32)
out = [torch.empty_like(t)]
dist.all_gather(out, t); dist.barrier(); torch.cuda.synchronize()
x = torch.ones(4, device=dev, dtype=torch.float64); dist.all_reduce(x, op=dist.ReduceOp.MAX)
print("rccl ok", out[0].tolist(), x.tolist())
dist.destroy_process_group()
