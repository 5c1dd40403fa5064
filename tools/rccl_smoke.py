#!/usr/bin/env python
"""RCCL sanity check on a 1-GPU box: process group "nccl" (= RCCL on ROCm) with one rank, SUM all-reduce of a tensor the
size of the flat gradient bucket (47 M floats) and the MAX all-reduce of the abnormal-loss flag -- the two collectives of
gtos_amd.train.Trainer.  Multi-rank behaviour is covered on CPU by tests/test_dp_gloo.py."""
import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(47_000_000, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
t=time.time()
for _ in range(5): dist.all_reduce(x)
torch.cuda.synchronize()
print("rccl all_reduce ok", float(x[0]), (time.time()-t)/5*1e3, "ms")
flag = torch.tensor([1.0], device="cuda"); dist.all_reduce(flag, op=dist.ReduceOp.MAX); print(float(flag))
dist.barrier(); dist.destroy_process_group()
