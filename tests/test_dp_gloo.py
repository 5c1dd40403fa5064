"""Data-parallel path on CPU: world_size 2, gloo.  The flat gradient bucket reduced with ONE all-reduce must equal
the reference's per-parameter all_reduce(SUM)/world_size (generator/train.py:74-79), the ranks must draw different
synthetic graphs, and the collective abnormal-loss decision must agree on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gtos_amd.flat import FlatParams
    from gtos_amd import synth
    torch.manual_seed(19940117)                       # identical init on every rank (train.py:98-100)
    model = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.LayerNorm(7), torch.nn.Linear(7, 3))
    ref = [p.detach().clone() for p in model.parameters()]
    flat = FlatParams(model)                          # CPU: views only, no kernels
    assert all(torch.equal(p.detach(), r) for p, r in zip(model.parameters(), ref))
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 13)
    model(x).pow(2).sum().backward()                  # autograd accumulates into the flat bucket views
    per_param = [p.grad.clone() for p in model.parameters()]
    assert float(flat.grad.abs().sum()) > 0
    for g in per_param:                               # the reference way
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g /= world
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)  # ours: one bucket
    ok = all(torch.allclose(p.grad / world, g, rtol=1e-6, atol=1e-7) for p, g in zip(model.parameters(), per_param))
    b, _ = synth.make_batch(5, 2, 6, 5, first_graph=rank * 2)
    sig = torch.tensor([float(b["concept"].sum())])
    sigs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(sigs, sig)
    flag = torch.tensor([1.0 if rank == 1 else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    q.put((rank, ok, [float(s) for s in sigs], float(flag)))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_per_parameter():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sigs, flag in res:
        assert ok, "flat all-reduce differs from per-parameter all-reduce on rank %d" % rank
        assert sigs[0] != sigs[1], "ranks drew identical synthetic graphs"
        assert flag == 1.0
