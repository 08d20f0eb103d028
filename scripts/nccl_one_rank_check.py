import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # keep the RCCL banner (NCCL_DEBUG=VERSION on the pool) off stdout
dev = torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(2*5*3, dtype=torch.int64, device=dev).reshape(2,5,3)
out = torch.empty((1,2,5,3), dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(out, x); dist.barrier(); torch.cuda.synchronize()
assert torch.equal(out[0], x)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.ones((4,3), device=dev); dist.broadcast(c, src=0)
print("nccl 1-rank ok", float(t.item()))
dist.destroy_process_group()
