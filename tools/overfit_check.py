#!/usr/bin/env python
"""Training sanity check: 300 steps on ONE fixed C1 batch (dropout 0.1, warm-up 400) in bf16 and in fp32 through the
whole product path (bf16: trie-evaluated GRU with per-node dropout masks, host relation index, fused copy/NLL kernel; fp32:
per-row GRU; factored attention, flat segmented Adam).  The two precisions should track each other and
the loss should fall from ~7.5 to ~1 within 150 steps (it rises again as the learning rate approaches its peak on this
single tiny batch).  Usage on the GPU box: python tools/overfit_check.py"""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from gtos_amd import synth, ops
from gtos_amd.config import build_generator
from gtos_amd.generator import Generator
from gtos_amd.train import Trainer
dev = torch.device("cuda:0")
for cfgname, dtype in (("C1", torch.bfloat16), ("C1", torch.float32)):
    cfg = synth.CONFIGS[cfgname]
    model = build_generator(Generator, cfgname, dev, dropout=0.1).to(dev)
    model.set_compute_dtype(dtype); model.train()
    tr = Trainer(model, cfg["d"], warmup_steps=400, compute_dtype=dtype, world_size=1)
    batch, _ = synth.make_config_batch(cfgname)
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    batch = {k: v.to(dev) for k, v in attach_relation_index(attach_path_trie(batch)).items()}
    ops.set_seed(1)
    losses = [tr.step(batch) for _ in range(300)]
    print(cfgname, dtype, [round(l, 3) for l in losses[::30]], "final", round(losses[-1], 3))
