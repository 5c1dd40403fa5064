"""Golden vectors for the inference path (beam search) and the vocabulary files, produced by the REFERENCE's own code.

  beam_smatch.npz / beam_smatch.json
      The six AMRs of generator/smatch/test_input{1,2}.txt -> generator/AMRGraph.py -> generator/data.py:batchify
      (train=False) -> generator/generator.py:Generator.work (beam search of generator/search.py) on a small
      random-weight model.  The .npz holds the state_dict and the batch tensors, the .json the vocabulary files (text),
      what the reference's Vocab makes of them, the per-graph copy vocabularies and the beams (finished / alive
      hypotheses with scores, k-best).

Run in the build container only:  python tests/golden/make_golden_beam.py
Harness shims (monkey-patches, the reference files are not edited): np.int; Tensor.cuda -> identity (the reference
moves step inputs with .cuda(device), there is no GPU here); bool causal mask.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/generator"
sys.path.insert(0, REF)
np.int = int
torch.Tensor.cuda = lambda self, *a, **k: self

import data as rdata                          # noqa: E402
import generator as rgen                      # noqa: E402
import transformer as rtf                     # noqa: E402
from smatch import AMR                        # noqa: E402
from AMRGraph import AMRGraph                 # noqa: E402
from extract import LexicalMap                # noqa: E402

rtf.SelfAttentionMask.get_mask = staticmethod(lambda size: torch.ones((size, size), dtype=torch.bool).triu_(1))

GEN_ARGS = (8, 12, 8, 12, [(3, 16)], 10, 10, 6, 8, 2)      # tests/tests_support.py SMALL_GEN_ARGS
D, FF, H, SNT_L, GRAPH_L, INF_L = 32, 64, 4, 1, 2, 3
ALPHA = 0.6
RUNS = [(4, 9, 1), (3, 7, 3), (1, 5, 1)]           # (beam size, max_time_step, min_time_step)


def main():
    graphs = []
    for fn in ("test_input1.txt", "test_input2.txt"):
        with open(os.path.join(REF, "smatch", fn), encoding="utf8") as f:
            while True:
                line = AMR.get_amr_line(f)
                if not line:
                    break
                graphs.append(AMRGraph(AMR.parse_AMR_line(line)))
    labels = sorted({d['label'] for g in graphs for _, _, d in g.graph.edges(data=True)})
    concepts = sorted({c for g in graphs for c in g.name2concept.values()})
    words = ["the", "boy", "girl", "wants", "to", "go", "believe", "a", "and", "is", "not", "he", "she", "it", "that",
             "dog", "cat", "see", "say", "will", "can", "do", "of", "in", "on", "with", "for", "very", "good", "bad"]
    rng = np.random.RandomState(7)
    files = {}

    def vocab_text(tokens, counts, blank_after=None):
        lines = []
        for i, (t, c) in enumerate(zip(tokens, counts)):
            lines.append("%s\t%d" % (t, c))
            if blank_after is not None and i == blank_after:
                lines.append("")                        # malformed line: pins the reference's fall-through behaviour
        return "\n".join(lines) + "\n"

    # counts straddle the thresholds (5 for words/concepts/relations, 100 for characters)
    files["relation_vocab"] = vocab_text(labels, [1000] * len(labels))
    files["concept_vocab"] = vocab_text(concepts, [int(c) for c in rng.choice([3, 4, 5, 9, 50], len(concepts))])
    # a few concepts are predictable tokens as well (copy and generate mass meet), three words fall below the threshold
    tok = words + concepts[:4]
    files["token_vocab"] = vocab_text(tok, [int(c) for c in rng.choice([2, 6, 40, 700], len(tok), p=[.1, .3, .3, .3])], blank_after=5)
    files["predictable_token_vocab"] = vocab_text(tok, [int(c) for c in rng.choice([4, 6, 40, 700], len(tok), p=[.1, .3, .3, .3])])
    chars = sorted({ch for w in tok + concepts for ch in w})
    files["concept_char_vocab"] = vocab_text(chars, [int(c) for c in rng.choice([50, 99, 100, 5000], len(chars))])
    files["token_char_vocab"] = vocab_text(chars, [int(c) for c in rng.choice([99, 100, 5000], len(chars))])
    tdir = tempfile.mkdtemp()
    for k, v in files.items():
        with open(os.path.join(tdir, k), "w") as fo:
            fo.write(v)
    V = rdata.Vocab
    vocabs = {'concept': V(tdir + '/concept_vocab', 5, [rdata.CLS]), 'token': V(tdir + '/token_vocab', 5, [rdata.STR, rdata.END]),
              'predictable_token': V(tdir + '/predictable_token_vocab', 5, [rdata.END]),
              'token_char': V(tdir + '/token_char_vocab', 100, [rdata.STR, rdata.END]),
              'concept_char': V(tdir + '/concept_char_vocab', 100, [rdata.STR, rdata.END]),
              'relation': V(tdir + '/relation_vocab', 5, [rdata.CLS, rdata.rCLS, rdata.SEL, rdata.TL])}
    vocab_truth = {k: {"size": v.size, "coverage": v.coverage, "idx2token": list(v._idx2token),
                       "token2idx": {t: v.token2idx(t) for t in sorted(set(v._idx2token)) + ["never-seen"]},
                       "priority": {t: v.priority(t) for t in list(v._priority)[:10] + ["never-seen"]}}
                   for k, v in vocabs.items()}

    lex = LexicalMap()
    items = []
    for g in graphs:
        concept, depth, relation, ok = g.collect_concepts_and_relations()
        assert ok
        cp_seq, t2i, i2t = lex.get(concept, vocabs['predictable_token'])
        # the copy ids come from iterating a set (extract.py:56-62): re-number them in first-occurrence order so that the
        # fixture does not depend on the hash seed of this process
        order = [c for k, c in enumerate(cp_seq) if c in t2i and c not in cp_seq[:k]]
        t2i = {c: vocabs['predictable_token'].size + k for k, c in enumerate(order)}
        i2t = {v: k for k, v in t2i.items()}
        item = {'concept': concept, 'depth': depth, 'relation': relation, 'token': ["the"], 'cp_seq': cp_seq,
                'token2idx': t2i, 'idx2token': i2t, 'abstract': {}}
        rel = json.loads(json.dumps(relation))                       # the loader indexes relation[str(i)][str(j)]
        item['relation'] = rel
        items.append(item)
    batch = rdata.batchify(items, vocabs, train=False)

    torch.manual_seed(20240117)
    model = rgen.Generator(vocabs, *GEN_ARGS, D, FF, H, 0.0, SNT_L, GRAPH_L, INF_L, None, torch.device('cpu'))
    with torch.no_grad():
        for n_, p in model.named_parameters():                       # spread the random model's predictions out
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            elif "token_generator.generator" in n_ or "token_generator.transfer" in n_ or "token_generator.diverter" in n_:
                p.copy_(0.6 * torch.randn_like(p))
            elif "embed" in n_ or "probe_generator" in n_ or "concept_depth" in n_:
                p.copy_(0.3 * torch.randn_like(p))
            else:
                p.mul_(4.0)
        pv = vocabs['predictable_token']
        gb = model.decoder.token_generator.generator.bias            # make <END> and <UNK> regular top-k members
        model.decoder.token_generator.generator.weight[pv.token2idx(rdata.END)] *= 0.25
        gb[pv.token2idx(rdata.END)] = 5.0
        gb[pv.unk_idx] += 2.5
        model.decoder.token_generator.diverter.bias[0] += 3.0        # generate vs copy: keep both in play
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
    for k in ("concept", "concept_char", "concept_depth", "relation", "relation_bank", "relation_length", "cp_seq"):
        arrs["batch/" + k] = batch[k].numpy()
    np.savez_compressed(os.path.join(HERE, "beam_smatch.npz"), **arrs)
    meta = {"files": files, "vocab_truth": vocab_truth,
            "local_idx2token": [{str(k): v for k, v in d.items()} for d in batch['local_idx2token']],
            "cfg": {"gen_args": [list(a) if isinstance(a, tuple) else a for a in GEN_ARGS], "d": D, "ff": FF, "H": H,
                    "snt_layers": SNT_L, "graph_layers": GRAPH_L, "inference_layers": INF_L, "alpha": ALPHA},
            "runs": runs}
    with open(os.path.join(HERE, "beam_smatch.json"), "w") as fo:
        json.dump(meta, fo, indent=0)
    print("beam_smatch.npz %.1f KB, beam_smatch.json %.1f KB" % (os.path.getsize(os.path.join(HERE, "beam_smatch.npz")) / 1024,
                                                                os.path.getsize(os.path.join(HERE, "beam_smatch.json")) / 1024))


if __name__ == "__main__":
    main()
