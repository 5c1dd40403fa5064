#!/usr/bin/env python
"""hipGraph replay of a training step at one configuration, outside bench.py:  python -X faulthandler tools/graph_probe.py C2 [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops, synth  # noqa: E402
from gtos_amd.config import build_generator  # noqa: E402
from gtos_amd.encoder import set_relation_mask_sharing  # noqa: E402
from gtos_amd.generator import Generator  # noqa: E402
from gtos_amd.pathtrie import attach_path_trie  # noqa: E402
from gtos_amd.relindex import attach_relation_index  # noqa: E402
from gtos_amd.train import GraphedStep, Trainer  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C1"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    cfg = synth.CONFIGS[name]
    m = build_generator(Generator, name, dev).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    set_relation_mask_sharing(m, "node")
    m.train()
    batch, _ = synth.make_config_batch(name, rank=0)
    attach_relation_index(attach_path_trie(batch))
    batch = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in batch.items()}
    tr = Trainer(m, cfg["d"], warmup_steps=2000, compute_dtype=torch.bfloat16)
    for _ in range(5):
        tr.step(batch, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batch, sync=False)
    torch.cuda.synchronize()
    print("eager %.2f ms/step" % (1e3 * (time.perf_counter() - t0) / steps), flush=True)
    t0 = time.perf_counter()
    gs = GraphedStep(tr, batch)
    print("capture %.2f s" % (time.perf_counter() - t0), flush=True)
    for _ in range(3):
        gs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = [gs() for _ in range(steps)]
    torch.cuda.synchronize()
    print("graph %.2f ms/step" % (1e3 * (time.perf_counter() - t0) / steps), [round(p.value(), 4) for p in pend[:4]], flush=True)
    gs.close()


if __name__ == "__main__":
    main()
