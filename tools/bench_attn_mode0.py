#!/usr/bin/env python
"""Micro-benchmark of the relation-free attention launches of the decoder / sentence encoder (mode 0 of the fused attention kernels):
T x S x B of C2's decoder side, forward and backward, through the C ABI with HIP events.  GTOS_ATTN_M0=0/1 switches the round-4
specialisation (four keys per register set in flight).   python tools/bench_attn_mode0.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops  # noqa: E402

CASES = [  # name, T, S, B, d, H, causal mask, key padding, weights wanted
    ("decoder self   50x50  H=8", 50, 50, 64, 512, 8, True, True, False),
    ("decoder cross  50x100 H=8", 50, 100, 64, 512, 8, False, True, False),
    ("alignment      50x100 H=1", 50, 100, 64, 512, 1, False, True, True),
    ("snt encoder    50x50  H=8", 50, 50, 64, 512, 8, True, True, False),
    ("C3 cross       70x60  H=8", 70, 60, 64, 512, 8, False, True, False),
]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    for name, T, S, B, d, H, causal, kp, need_w in CASES:
        q = torch.randn(T, B, d, generator=g).to(dev, torch.bfloat16).requires_grad_()
        kv = torch.randn(S, B, 2 * d, generator=g).to(dev, torch.bfloat16).requires_grad_()
        mask = torch.ones(T, S, dtype=torch.bool).triu_(1).to(dev) if causal and T == S else None
        pad = torch.zeros(S, B, dtype=torch.bool)
        pad[S - 3:, ::4] = kp
        pad = pad.to(dev)
        go = torch.randn(T, B, d, generator=g).to(dev, torch.bfloat16)

        def fwd():
            return ops.attention_core(q, kv, (0, 0, d), d, H, (d // H) ** -0.5, key_pad=pad, attn_mask=mask, p_drop=0.2, need_weights=need_w)
        o, w = fwd()
        o.backward(go)
        torch.cuda.synchronize()
        reps = 50
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        outs = [fwd() for _ in range(reps)]
        e[1].record()
        for o, w in outs:
            o.backward(go)
        e[2].record()
        torch.cuda.synchronize()
        print("%-28s fwd %7.1f us   bwd (q + kv kernels) %7.1f us" % (name, 1e3 * e[0].elapsed_time(e[1]) / reps, 1e3 * e[1].elapsed_time(e[2]) / reps),
              flush=True)


if __name__ == "__main__":
    main()
