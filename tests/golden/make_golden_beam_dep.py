"""Golden vectors for the TRANSLATOR flavour of the inference path, on the real data the reference ships.

  beam_dep_dev.npz / beam_dep_dev.json
      translator/extract.py builds the six vocabulary files from translator_data/dev.txt; the first 6 dependency trees
      go through translator/dependencyGraph.py + translator/data.py:batchify; translator/generator.py:Generator.work
      (single shortest path per pair, depth table of 256) decodes them with translator/search.py on a small
      random-weight model.  The .json holds the vocabulary files (only lines that matter: every token that reaches its
      threshold, plus the total count mass so that `coverage` can be checked), what the reference's Vocab makes of
      them, the copy vocabularies and the beams.  The .npz also carries the TRAINING forward/backward of the same real
      batch (loss + every parameter gradient, dropout 0).

Run in the build container only:  python tests/golden/make_golden_beam_dep.py
Harness shims (monkey-patches, the reference files are not edited): np.int; Tensor.cuda -> identity; bool causal mask.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
np.int = int
torch.Tensor.cuda = lambda self, *a, **k: self     # translator/search.py moves index tensors with .cuda(device)
sys.path.insert(0, os.path.join(REF, "translator"))

import data as rdata                          # noqa: E402
import generator as rgen                      # noqa: E402
import transformer as rtf                     # noqa: E402
from extract import IO, LexicalMap            # noqa: E402
from dependencyGraph import dependencyGraph   # noqa: E402

rtf.SelfAttentionMask.get_mask = staticmethod(lambda size: torch.ones((size, size), dtype=torch.bool).triu_(1))
# q *= scaling on a chunk view trips autograd on current torch (training forward only) -> hand out clones
_orig_qkv = rtf.MultiheadAttention.in_proj_qkv
rtf.MultiheadAttention.in_proj_qkv = lambda self, q: tuple(t.clone() for t in _orig_qkv(self, q))

GEN_ARGS = (8, 12, 8, 12, [(3, 16)], 10, 10, 6, 8, 2)      # tests/tests_support.py SMALL_GEN_ARGS
D, FF, H, SNT_L, GRAPH_L, INF_L = 32, 64, 4, 1, 2, 3
ALPHA = 0.6
RUNS = [(3, 8, 1), (2, 6, 2)]                              # (beam size, max_time_step, min_time_step)
SPECS = {'concept': ('concept_vocab', 5, [rdata.CLS]), 'token': ('token_vocab', 5, [rdata.STR, rdata.END]),
         'predictable_token': ('predictable_token_vocab', 5, [rdata.END]),
         'token_char': ('token_char_vocab', 100, [rdata.STR, rdata.END]),
         'concept_char': ('concept_char_vocab', 100, [rdata.STR, rdata.END]),
         'relation': ('relation_vocab', 5, [rdata.CLS, rdata.rCLS, rdata.SEL, rdata.TL])}


def main():
    tdir = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(REF, "translator", "extract.py"), "--train_data",
                           os.path.join(REF, "translator_data", "dev.txt")], cwd=tdir, stdout=subprocess.DEVNULL)
    # keep the fixture small: drop the below-threshold tail of every file but keep its count mass in ONE synthetic line
    # (a token that stays below the threshold), so sizes, ids and coverage are what the full file gives
    files = {}
    for name, (fn, thr, _) in SPECS.items():
        kept, tail = [], 0
        for line in open(os.path.join(tdir, fn)):
            tok, cnt = line.rstrip("\n").split("\t")
            if int(cnt) >= thr:
                kept.append(line.rstrip("\n"))
            else:
                tail += int(cnt)
        text = "\n".join(kept) + "\n"
        # the tail mass is spread over lines that each stay below the threshold
        k = 0
        while tail > 0:
            c = min(tail, thr - 1)
            text += "<tail%d>\t%d\n" % (k, c)
            tail -= c
            k += 1
        files[fn] = text
        with open(os.path.join(tdir, fn + ".small"), "w") as fo:
            fo.write(text)
    V = rdata.Vocab
    vocabs = {name: V(os.path.join(tdir, fn + ".small"), thr, sp) for name, (fn, thr, sp) in SPECS.items()}
    full = {name: V(os.path.join(tdir, fn), thr, sp) for name, (fn, thr, sp) in SPECS.items()}
    for name in vocabs:                                    # the reduced files are equivalent to the full ones
        assert vocabs[name]._idx2token == full[name]._idx2token and abs(vocabs[name].coverage - full[name].coverage) < 1e-12
    vocab_truth = {k: {"size": v.size, "coverage": v.coverage, "idx2token": list(v._idx2token)} for k, v in vocabs.items()}

    lex = LexicalMap()
    items, trees = [], []
    for k, (dep, head, tok, tgt) in enumerate(IO.read1(os.path.join(REF, "translator_data", "dev.txt"))):
        if k >= 6:
            break
        trees.append([dep, head, tok, tgt])
        g = dependencyGraph(dep, head, tok, tgt)
        concept, depth, relation, ok = g.collect_concepts_and_relations()
        assert ok
        cp_seq, t2i, i2t = lex.get(concept, vocabs['predictable_token'])
        order = [c for j, c in enumerate(cp_seq) if c in t2i and c not in cp_seq[:j]]     # hash-seed independent copy ids
        t2i = {c: vocabs['predictable_token'].size + j for j, c in enumerate(order)}
        i2t = {v: c for c, v in t2i.items()}
        items.append({'concept': concept, 'depth': depth, 'relation': relation, 'token': tgt, 'cp_seq': cp_seq,
                      'token2idx': t2i, 'idx2token': i2t})
    from utils import move_to_device                      # translator/utils.py: numpy -> tensors, as work.py does
    batch = move_to_device(rdata.batchify(items, vocabs), torch.device('cpu'))

    torch.manual_seed(20240118)
    model = rgen.Generator(vocabs, *GEN_ARGS, D, FF, H, 0.0, SNT_L, GRAPH_L, INF_L, None, torch.device('cpu'))
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            elif "token_generator.generator" in n_ or "token_generator.transfer" in n_ or "token_generator.diverter" in n_:
                p.copy_(0.3 * torch.randn_like(p))
            elif "embed" in n_ or "probe_generator" in n_ or "concept_depth" in n_:
                p.copy_(0.3 * torch.randn_like(p))
            else:
                p.mul_(4.0)
        pv = vocabs['predictable_token']
        gw, gb = model.decoder.token_generator.generator.weight, model.decoder.token_generator.generator.bias
        gw[pv.token2idx(rdata.END)] *= 0.25
        gb[pv.token2idx(rdata.END)] = 4.2
        gb[pv.unk_idx] += 2.0
        model.decoder.token_generator.diverter.bias[0] += 2.0
    model.eval()
    runs = []
    for (beam_size, max_step, min_step) in RUNS:
        beams = model.work(batch, beam_size, max_step, min_step)
        expect = []
        for beam in beams:
            fin = [(h.seq, h.score) for h in beam.completed_hypotheses]
            alive = [(h.seq, h.score) for h in beam.hypotheses]
            steps = beam.steps
            best = [(h.seq, h.score) for h in beam.get_k_best(beam_size, ALPHA)]
            expect.append({"finished": fin, "alive": alive, "steps": steps, "k_best": best})
            print(len(fin), len(alive), steps, " ".join(best[0][0]), "%.4f" % best[0][1])
        runs.append({"beam": beam_size, "max_step": max_step, "min_step": min_step, "expect": expect})
    arrs = {"sd/" + k: v.numpy() for k, v in model.state_dict().items()}
    for k in ("concept", "concept_char", "concept_depth", "relation", "relation_bank", "relation_length", "cp_seq",
              "token_in", "token_char_in", "token_out"):
        arrs["batch/" + k] = np.asarray(batch[k])
    # the same real batch through the TRAINING forward/backward (dropout 0): loss and every parameter gradient
    model.train()
    loss = model(batch)
    loss.backward()
    arrs["train/loss"] = loss.detach().numpy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            arrs["grad/" + k] = p.grad.numpy()
    print("training loss on the real batch: %.6f" % float(loss))
    np.savez_compressed(os.path.join(HERE, "beam_dep_dev.npz"), **arrs)
    meta = {"files": files, "vocab_truth": vocab_truth, "trees": trees,
            "local_token2idx": batch['local_token2idx'],
            "local_idx2token": [{str(k): v for k, v in d.items()} for d in batch['local_idx2token']],
            "cfg": {"gen_args": [list(a) if isinstance(a, tuple) else a for a in GEN_ARGS], "d": D, "ff": FF, "H": H,
                    "snt_layers": SNT_L, "graph_layers": GRAPH_L, "inference_layers": INF_L, "alpha": ALPHA, "depth_size": 256},
            "runs": runs}
    with open(os.path.join(HERE, "beam_dep_dev.json"), "w") as fo:
        json.dump(meta, fo, indent=0, ensure_ascii=False)
    print("beam_dep_dev.npz %.1f KB, beam_dep_dev.json %.1f KB" % (os.path.getsize(os.path.join(HERE, "beam_dep_dev.npz")) / 1024,
                                                                  os.path.getsize(os.path.join(HERE, "beam_dep_dev.json")) / 1024))


if __name__ == "__main__":
    main()
