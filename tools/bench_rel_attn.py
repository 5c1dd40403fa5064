#!/usr/bin/env python
"""Relation-attention kernel micro-benchmark (HIP events on the launch stream): dense-signature and factored operand,
forward and backward, against the HBM roofline.  Algorithmic bytes per SURVEY.md section 8d:
   forward  P*2d*s + 4*n*B*d*s + n*B            fwd+bwd (dense)  3*P*2d*s + 10*n*B*d*s

    python tools/bench_rel_attn.py [--cfg C2] [--reps 10] [--mode dense|factored|both]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops, synth  # noqa: E402
from gtos_amd.relindex import build_relation_index  # noqa: E402

HBM = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C2")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--mode", default="both")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.cfg]
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    s_el = 2 if dt == torch.bfloat16 else 4
    n, B, d, H = cfg["N"] + 1, cfg["B"], cfg["d"], cfg["H"]
    P = n * n * B
    batch, stats = synth.make_config_batch(a.cfg)
    idx = batch["relation"].to(dev)
    R = stats["R"]
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(n, B, 3 * d, generator=g).to(dev, dt)
    bankp = (0.3 * torch.randn(R, 2 * d, generator=g)).to(dev, dt)
    pad = torch.zeros(n, B, dtype=torch.bool, device=dev)
    index = build_relation_index(batch["relation"], R).to(dev)      # the loader's host-built index, as in the training step
    fact = ops.FactoredRelation(torch.zeros(R, d, device=dev, dtype=dt), idx, index=index)
    out = []
    fwd_bytes = P * 2 * d * s_el + 4 * n * B * d * s_el + n * B
    bwd_dense_bytes = 2 * P * 2 * d * s_el + P * d * s_el + 10 * n * B * d * s_el     # re-read rarb, write d_rarb, re-read ra
    for mode in (["dense", "factored"] if a.mode == "both" else [a.mode]):
        if mode == "dense":
            rel = bankp[idx.reshape(-1)].view(n, n, B, 2 * d).contiguous()
            f = None
        else:
            rel, f = bankp, fact
        rel = rel.detach().requires_grad_()
        q = qkv.detach().requires_grad_()
        fwd = lambda: ops.attention_core(q.detach(), None, (0, d, 2 * d), d, H, 0.125, rel=rel.detach(), fact=f, key_pad=pad)
        ms_f = timed(fwd, a.reps)
        o, _ = ops.attention_core(q, None, (0, d, 2 * d), d, H, 0.125, rel=rel, fact=f, key_pad=pad)
        do = torch.randn_like(o)
        bwd = lambda: torch.autograd.grad(o, (q, rel), do, retain_graph=True)
        ms_b = timed(bwd, a.reps)
        rec = {"cfg": a.cfg, "mode": mode, "dtype": a.dtype, "n": n, "B": B, "P": P, "R": R,
               "fwd_us": round(ms_f * 1e3, 1), "fwd_alg_GB": round(fwd_bytes / 1e9, 3),
               "fwd_GBps": round(fwd_bytes / ms_f / 1e6, 1), "fwd_frac_hbm": round(fwd_bytes / ms_f / 1e6 / HBM, 4),
               "bwd_us": round(ms_b * 1e3, 1)}
        if mode == "dense":
            rec["bwd_alg_GB"] = round(bwd_dense_bytes / 1e9, 3)
            rec["bwd_GBps"] = round(bwd_dense_bytes / ms_b / 1e6, 1)
            rec["bwd_frac_hbm"] = round(bwd_dense_bytes / ms_b / 1e6 / HBM, 4)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del rel, o
    return out


if __name__ == "__main__":
    main()
