#!/usr/bin/env python
"""Dropout-semantics A/B for the trie-evaluated RelationEncoder: does drawing the embedding / inter-layer
dropout masks per TRIE NODE (paths that share a prefix share the mask on the shared part) regularise like the reference's
independent masks per (path, position) (generator/encoder.py:99-100, nn.GRU(dropout=...), :76-82)?

Data: the reference's shipped development set, 2,169 dependency-parsed sentence pairs (tests/golden/dep_dev_trees.json.gz,
made by tests/golden/make_golden_trees.py), translator flavour.  The first 1,900 trees train, the last 269 are held out.
Three arms, identical in everything else (initial weights, batch order, <UNK> noise, dropout hash seeds):
  node   GTOS_GRU_TRIE=1  production path: masks per trie node
  row    GTOS_GRU_TRIE=0  per-row path: masks per (path, position), the reference's semantics
  none   RelationEncoder dropout 0 (everything else keeps dropout): how much this dropout matters at all
Every --every steps: mean training loss since the last record and the held-out loss (eval mode).  Usage on the GPU box:
    python tools/dropout_ab.py --steps 2000 --out gpurun_out/dropout_ab.json                       (all 1,900 training trees)
    python tools/dropout_ab.py --steps 3000 --train-trees 256 --out gpurun_out/dropout_ab_256.json  (over-fitting regime)"""
import argparse
import gzip
import json
import os
import random
import sys
import tempfile
import time
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtos_amd import gru, ops  # noqa: E402
from gtos_amd.data import DependencyLoader, Prefetcher  # noqa: E402
from gtos_amd.generator import Generator  # noqa: E402
from gtos_amd.train import Trainer  # noqa: E402
from gtos_amd.vocab import Vocab, CLS, rCLS, SEL, TL, STR, END  # noqa: E402


def build_vocabs(trees, tmp):
    """The six vocabularies of translator/extract.py from the training trees (every token kept: min count 1)."""
    cnt = {k: Counter() for k in ("concept", "token", "predictable_token", "relation", "concept_char", "token_char")}
    for dep, head, tok, tgt in trees:
        cnt["concept"].update(tok)
        cnt["token"].update(tgt)
        cnt["predictable_token"].update(tgt)
        cnt["relation"].update(dep)
        cnt["relation"].update(r + "_r_" for r in dep)
        for w in tok:
            cnt["concept_char"].update(w)
        for w in tgt:
            cnt["token_char"].update(w)
    specials = {"concept": [CLS], "token": [STR, END], "predictable_token": [END], "relation": [CLS, rCLS, SEL, TL],
                "concept_char": [STR, END], "token_char": [STR, END]}
    out = {}
    for k, c in cnt.items():
        path = os.path.join(tmp, k)
        with open(path, "w", encoding="utf8") as f:
            for w, n in c.most_common():
                if "\t" not in w and w.strip():
                    f.write("%s\t%d\n" % (w, n))
        out[k] = Vocab(path, 2 if k in ("concept", "token", "predictable_token") else 1, specials[k])
    return out


def run_arm(arm, a, vocabs, train_trees, held_trees, dev, seed=0):
    """``seed``: 0 = the round-3 run (initial weights 19940117, hash seeds 4242, batch order 7); s > 0 offsets all three."""
    from gtos_amd.encoder import set_relation_mask_sharing
    gru.TRIE = arm != "row"
    torch.manual_seed(19940117 + seed)
    model = Generator(vocabs, 32, 300, 32, 300, [(3, 256)], 128, 128, 100, 256, 2, a.d, 2 * a.d, 8, a.dropout, 1, a.layers, 2, None, dev,
                      depth_size=256).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    if arm == "none":
        model.relation_encoder.dropout = 0.0
    # round 4: the mask semantics are a property of the module: "node" = masks per trie node (opt-in), "path" = per (path, position),
    # the reference's and the library default ("row" = the same through the GTOS_GRU_TRIE=0 switch of round 3)
    set_relation_mask_sharing(model, "node" if arm in ("node", "none") else "path")
    tr = Trainer(model, a.d, warmup_steps=a.warmup, compute_dtype=torch.bfloat16)
    ops.set_seed(4242 + 1000003 * seed)
    held = [{k: (v.to(dev) if hasattr(v, "to") else v) for k, v in b.items()}
            for b in DependencyLoader(vocabs, held_trees, a.batch_size, for_train=False)]

    def held_out_loss():
        model.eval()
        tot = 0.0
        with torch.no_grad():
            for b in held:
                tot += float(model(b)) * b["token_in"].shape[1]           # forward returns the batch mean of per-sentence losses
        model.train()
        return tot / len(held_trees)

    loader = DependencyLoader(vocabs, train_trees, a.batch_size, for_train=True, rng=random.Random(7 + seed))
    loader.set_unk_rate(a.unk_rate)

    def epochs():
        while True:
            yield from loader.thunks()
    feed = Prefetcher(epochs(), depth=4, workers=2, device=dev)
    model.train()
    curve, pend, t0 = [], [], time.time()
    for step in range(1, a.steps + 1):
        pend.append(tr.step(next(feed), sync=False))
        if step % a.every == 0 or step == a.steps:
            vals = [v for v in (p.value() for p in pend) if v is not None]
            pend = []
            rec = {"step": step, "train_loss": sum(vals) / max(1, len(vals)), "held_out_loss": held_out_loss()}
            curve.append(rec)
            print("%-5s step %5d  train %.4f  held-out %.4f  (%.0f s)" % (arm, step, rec["train_loss"], rec["held_out_loss"], time.time() - t0),
                  flush=True)
    feed.close()
    return {"arm": arm, "seed": seed, "curve": curve, "seconds": round(time.time() - t0, 1), "discarded": tr.discarded}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=8000, help="size units (source tokens^2 + target tokens), translator/data.py")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--unk-rate", type=float, default=0.33)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--arms", default="node,row,none")
    ap.add_argument("--train-trees", type=int, default=1900,
                    help="size of the training subset (the held-out trees are always the last 269); a few hundred trees put the run in "
                         "the over-fitting regime, where regularisation differences show in the held-out loss")
    ap.add_argument("--seeds", default="0", help="comma-separated: one run of every arm per seed (initial weights, hash seeds, batch order)")
    ap.add_argument("--out", default="gpurun_out/dropout_ab.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    trees = [tuple(t) for t in json.load(gzip.open(os.path.join(ROOT, "tests", "golden", "dep_dev_trees.json.gz"), "rt", encoding="utf8"))]
    train_trees, held_trees = trees[:min(1900, a.train_trees)], trees[1900:]
    with tempfile.TemporaryDirectory() as tmp:
        vocabs = build_vocabs(train_trees, tmp)
    res = {"config": vars(a), "train_trees": len(train_trees), "held_out_trees": len(held_trees),
           "vocab_sizes": {k: v.size for k, v in vocabs.items()}, "arms": []}
    seeds = [int(x) for x in a.seeds.split(",")]
    for seed in seeds:
        for arm in a.arms.split(","):
            res["arms"].append(run_arm(arm, a, vocabs, train_trees, held_trees, dev, seed))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    # side-by-side table
    print("\nstep   " + "   ".join("%-22s" % ("%s/s%d train / held-out" % (r["arm"], r["seed"])) for r in res["arms"]))
    for i in range(len(res["arms"][0]["curve"])):
        print("%5d  " % res["arms"][0]["curve"][i]["step"] + "   ".join(
            "%8.4f / %8.4f     " % (r["curve"][i]["train_loss"], r["curve"][i]["held_out_loss"]) for r in res["arms"]))
    if len(seeds) > 1:                                  # per arm: mean and range over the seeds, at every record
        print("\nover seeds %s: mean [min .. max] of the held-out loss" % seeds)
        arms = a.arms.split(",")
        for i in range(len(res["arms"][0]["curve"])):
            cells = []
            for arm in arms:
                v = [r["curve"][i]["held_out_loss"] for r in res["arms"] if r["arm"] == arm]
                cells.append("%s %.4f [%.4f .. %.4f]" % (arm, sum(v) / len(v), min(v), max(v)))
            print("%5d  " % res["arms"][0]["curve"][i]["step"] + "    ".join(cells))


if __name__ == "__main__":
    main()
