#!/usr/bin/env python
"""The RCCL code path of Trainer.step on ONE GPU: process group "nccl" (= RCCL on ROCm) with a single rank, collectives forced
on -- the segment-wise async all-reduce of flat-bucket views launched from inside backward, the waits, the collective
abnormal-loss flag -- and the result compared with the same steps without any collective (one rank: the same up to the run-to-run noise of the
fp32-atomic reductions).
Multi-rank behaviour is covered on CPU (tests/test_dp_gloo.py) and, with gloo, by bench.py --gpus 2 on one GPU.

    python tools/rccl_step_check.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops, synth  # noqa: E402
from gtos_amd.config import build_generator  # noqa: E402
from gtos_amd.generator import Generator  # noqa: E402
from gtos_amd.pathtrie import attach_path_trie  # noqa: E402
from gtos_amd.relindex import attach_relation_index  # noqa: E402
from gtos_amd.train import Trainer  # noqa: E402


def run(force):
    dev = torch.device("cuda:0")
    model = build_generator(Generator, "C1", dev, dropout=0.1).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    # embed_dim only sets the learning rate (d^-0.5 ...): a small one keeps the two runs comparable -- the step is not bitwise
    # reproducible run to run (fp32 atomics in a few reductions), and warm-up 2 lets the collective skip flag come into play
    tr = Trainer(model, 250000, warmup_steps=2, compute_dtype=torch.bfloat16, world_size=1, force_collectives=force)
    batch, _ = synth.make_config_batch("C1")
    batch = {k: v.to(dev) for k, v in attach_relation_index(attach_path_trie(batch)).items()}
    ops.set_seed(7)
    loss = model(batch)                       # one backward + gradient collectives, no optimizer: the reduced bucket itself
    loss.backward()
    tr.all_reduce_grads()
    torch.cuda.synchronize()
    grad = tr.flat.grad.clone()
    tr.flat.zero_grad()
    losses = [tr.step(batch) for _ in range(5)]
    torch.cuda.synchronize()
    return losses, grad, tr.overlap, tr.comm_exposed_ms()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    l1, p1, ov1, ms = run(True)
    l0, p0, ov0, _ = run(False)
    assert ov1 and not ov0
    assert float((p1 - p0).norm() / p0.norm()) < 1e-3, float((p1 - p0).norm() / p0.norm())       # the all-reduced gradient bucket
    assert all(abs(a - b) < 5e-3 * max(1.0, abs(b)) for a, b in zip(l1, l0)), (l1, l0)
    print("rccl step check ok: 5 steps, losses %s, 4 segments all-reduced per step, compute stream stalled %.3f ms on them" % (
        [round(v, 4) for v in l1], ms))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
