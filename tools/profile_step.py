#!/usr/bin/env python
"""Per-shape GEMM time of one training step (HIP events around every gtos_gemm launch).

    python tools/profile_step.py [--config C2]
Prints, per (layout, fixed dims, dtypes): launches, summed long dimension, total ms, TF/s -- the table DESIGN.md quotes."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops, synth  # noqa: E402
from gtos_amd.config import build_generator  # noqa: E402
from gtos_amd.generator import Generator  # noqa: E402
from gtos_amd.train import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = synth.CONFIGS[a.config]
    model = build_generator(Generator, a.config, dev, factored_relation=True).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    trainer = Trainer(model, cfg["d"], warmup_steps=2000, compute_dtype=torch.bfloat16, world_size=1)
    batch, stats = synth.make_config_batch(a.config)
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    batch = {k: v.to(dev) for k, v in attach_relation_index(attach_path_trie(batch)).items()}
    for _ in range(2):
        trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.step(batch)
    torch.cuda.synchronize()
    plain = time.perf_counter() - t0
    # RelationEncoder alone (generator/encoder.py:66-119): forward and backward, HIP events, dropout as in training
    enc = model.relation_encoder
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf, tb = [], []
    for it in range(4):
        ev[0].record()
        out = enc(batch['relation_bank'], batch['relation_length'], trie=batch.get('relation_trie'))
        ev[1].record()
        out.backward(torch.ones_like(out))
        ops.join_side()
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            tf.append(ev[0].elapsed_time(ev[1]))
            tb.append(ev[1].elapsed_time(ev[2]))
    trainer.flat.zero_grad()
    print("RelationEncoder alone (R=%d, sum(len)=%d): forward %.2f ms, backward %.2f ms" % (
        stats["R"], int(batch['relation_length'].sum()), sum(tf) / len(tf), sum(tb) / len(tb)))
    ops.GEMM_PROFILE = {}
    t0 = time.perf_counter()
    trainer.step(batch)
    torch.cuda.synchronize()
    timed = time.perf_counter() - t0
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    rows = []
    for key, evs in prof.items():
        lay, n, k, dts, sk, big = key
        ms = sum(s.elapsed_time(e) for _, s, e in evs)
        long_sum = sum(m for m, _, _ in evs)
        rows.append((ms, lay, n, k, dts, sk, len(evs), long_sum, 2.0 * long_sum * n * k / ms / 1e9))
    rows.sort(reverse=True)
    print("step %.1f ms (%.1f ms with per-GEMM events); GEMM total %.1f ms in %d launches" % (
        plain * 1e3, timed * 1e3, sum(r[0] for r in rows), sum(r[6] for r in rows)))
    print("%-3s %6s %6s %-18s %3s %6s %10s %9s %8s" % ("lay", "N", "K|M", "dtypes", "sk", "calls", "sum(long)", "ms", "TF/s"))
    for ms, lay, n, k, dts, sk, calls, ls, tf in rows[:40]:
        print("%-3s %6d %6d %-18s %3d %6d %10d %9.3f %8.1f" % (lay, n, k, dts, sk, calls, ls, ms, tf))


if __name__ == "__main__":
    main()
