"""The launch plan of one training step of a BASELINE configuration, derived on the CPU with tests/dryrun.py (no GPU): which
libgtos_hip.so entry points a step calls and how often.  (ATen kernels and the extra launches an entry point makes inside the
library -- split-K reductions, multi-pass sums -- are not in this count; profiles/r3z_bench_c2_kernel_stats.csv has those.)
python tests/launch_plan.py C1 C2 > profiles/<round>_launch_plan.json"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from dryrun import DryRun                                                 # noqa: E402
from gtos_amd import synth                                                # noqa: E402


def plan(cfg):
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    from gtos_amd.train import Trainer
    dev = torch.device("cpu")
    with DryRun() as rec:
        model = build_generator(Generator, cfg, dev, factored_relation=True).to(dev)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        trainer = Trainer(model, synth.CONFIGS[cfg]["d"], warmup_steps=2000, compute_dtype=torch.bfloat16, world_size=1, rank=0)
        batch, st = synth.make_config_batch(cfg, rank=0)
        attach_relation_index(attach_path_trie(batch))
        trainer.step(batch, sync=False)
        n0, t0 = len(rec.calls), time.perf_counter()
        trainer.step(batch, sync=False)
        hist, gemms = {}, {}
        for name, args in rec.calls[n0:]:
            hist[name] = hist.get(name, 0) + 1
            if name == "gtos_gemm":                        # (dtype_a, dtype_out, trans_a, trans_b, M, N, K, ..., splitk at index 18)
                key = ("T" if args[2] else "N") + ("T" if args[3] else "N"), args[4], args[5], args[6], args[18]
                gemms[key] = gemms.get(key, 0) + 1
    shapes = sorted(({"op": k[0], "M": k[1], "N": k[2], "K": k[3], "splitk": k[4], "calls": v, "gflop": round(2e-9 * k[1] * k[2] * k[3] * v, 1)}
                     for k, v in gemms.items()), key=lambda r: -r["gflop"])
    return {"config": cfg, "batch": {k: st[k] for k in ("n", "B", "T", "R")}, "entry_point_calls_per_step": sum(hist.values()),
            "host_seconds_of_the_dry_step": round(time.perf_counter() - t0, 2), "by_entry": dict(sorted(hist.items(), key=lambda kv: -kv[1])),
            "gemm_tflop_per_step": round(sum(r["gflop"] for r in shapes) / 1e3, 2), "gemm_shapes": shapes[:24]}


if __name__ == "__main__":
    print(json.dumps({"tool": "tests/launch_plan.py (CPU dry run: launches recorded, not executed)",
                      "plans": [plan(c) for c in (sys.argv[1:] or ["C1"])]}, indent=1))
