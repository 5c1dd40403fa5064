#!/usr/bin/env python
"""Inference benchmark: beam search over the K/V-cached incremental decoder.

    python tools/bench_decode.py [--config C2] [--graphs 64] [--beam 8] [--max-steps 50] [--cpu-graphs 2]
Synthetic AMR-shaped graphs of the named BASELINE config (eval-mode batch: all shortest paths, K alternatives per pair),
random-weight model, bf16.  Reports sentences/s, decoder steps/s and hypothesis-steps/s; with --cpu-graphs N the pinned
oracle (full-prefix recompute on the host cores) decodes N of the graphs for the CPU column."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtos_amd import synth  # noqa: E402
from gtos_amd.config import generator_args  # noqa: E402


def local_vocabs(batch, pv):
    """cp_seq ids beyond the predictable vocabulary are per-graph copy ids: give them strings."""
    out = []
    cp = batch['cp_seq']
    for b in range(cp.shape[1]):
        out.append({int(i): "copy%d" % int(i) for i in cp[:, b].tolist() if i >= pv.size})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--graphs", type=int, default=64)
    ap.add_argument("--beam", type=int, default=8)
    ap.add_argument("--max-steps", type=int, default=50)
    ap.add_argument("--cpu-graphs", type=int, default=0)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    from gtos_amd.generator import Generator
    dev = torch.device("cuda:0")
    cfg = synth.CONFIGS[a.config]
    vocabs = synth.synth_vocabs()
    torch.manual_seed(19940117)
    model = Generator(vocabs, device=dev, depth_size=256 if cfg["kind"] == "dep" else 32, **generator_args(cfg)).to(dev)
    model.set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    model.eval()
    batch, stats = synth.make_config_batch(a.config, train=False, B=a.graphs)
    batch_dev = {k: v.to(dev) for k, v in batch.items()}
    batch_dev['local_idx2token'] = batch['local_idx2token'] = local_vocabs(batch, vocabs['predictable_token'])
    model.work(batch_dev, a.beam, 3)                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        enc = model.encode_step(batch_dev, train=False)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    beams = model.work(batch_dev, a.beam, a.max_steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = max(b.steps for b in beams)
    hyp_steps = sum(len(h.seq) - 1 for b in beams for h in (b.completed_hypotheses + b.hypotheses))
    out = {"config": a.config, "graphs": a.graphs, "n": stats["n"], "beam": a.beam, "max_steps": a.max_steps,
           "dtype": a.dtype, "seconds": dt, "encode_seconds": t_enc, "sentences_per_s": a.graphs / dt,
           "decoder_steps": steps, "ms_per_decoder_step": 1e3 * (dt - t_enc) / max(1, steps),
           "final_hypothesis_tokens": hyp_steps}
    if a.cpu_graphs:
        from oracle import gtos_oracle as O
        from bench import host_cores
        torch.set_num_threads(host_cores())
        torch.manual_seed(19940117)
        ref = O.Generator(vocabs, depth_size=256 if cfg["kind"] == "dep" else 32, **generator_args(cfg))
        ref.eval()
        small, _ = synth.make_config_batch(a.config, train=False, B=a.cpu_graphs)
        small['local_idx2token'] = local_vocabs(small, vocabs['predictable_token'])
        t0 = time.perf_counter()
        O.generator_work(ref, small, vocabs, a.beam, a.max_steps)
        ct = time.perf_counter() - t0
        out["cpu_oracle"] = {"graphs": a.cpu_graphs, "seconds": ct, "sentences_per_s": a.cpu_graphs / ct,
                             "threads": torch.get_num_threads()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
