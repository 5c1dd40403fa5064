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
    batch = {k: v.to(dev) for k, v in batch.items()}
    for _ in range(2):
        trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.step(batch)
    torch.cuda.synchronize()
    plain = time.perf_counter() - t0
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
