"""Golden vectors for the training-time UNK noise of the reference loader (translator/data.py:77-99,127-129,185; the
reference trains with --unk_rate 0.33, translator/train.sh / generator/train.sh:27).

  unk_dep_dev.npz   the 6 real dependency trees and vocabulary files of beam_dep_dev.json through the reference's
                    batchify(items, vocabs, unk_rate) after random.seed(seed), for a few (unk_rate, seed) settings:
                    `concept` and `token_in` (the two tensors the noise touches) plus `token_out` / `cp_seq` (which it must
                    not touch).

Run in the build container only:  python tests/golden/make_golden_unk.py
"""
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
np.int = int
sys.path.insert(0, os.path.join(REF, "translator"))

import data as rdata                          # noqa: E402
from extract import LexicalMap                # noqa: E402
from dependencyGraph import dependencyGraph   # noqa: E402

SETTINGS = [(0.33, 11), (0.33, 12), (0.9, 5), (0.0, 3)]
SPECS = {'concept': ('concept_vocab', 5, [rdata.CLS]), 'token': ('token_vocab', 5, [rdata.STR, rdata.END]),
         'predictable_token': ('predictable_token_vocab', 5, [rdata.END]),
         'token_char': ('token_char_vocab', 100, [rdata.STR, rdata.END]),
         'concept_char': ('concept_char_vocab', 100, [rdata.STR, rdata.END]),
         'relation': ('relation_vocab', 5, [rdata.CLS, rdata.rCLS, rdata.SEL, rdata.TL])}


def main():
    meta = json.load(open(os.path.join(HERE, "beam_dep_dev.json")))
    tdir = tempfile.mkdtemp()
    for fn, text in meta["files"].items():
        with open(os.path.join(tdir, fn), "w") as fo:
            fo.write(text)
    vocabs = {name: rdata.Vocab(os.path.join(tdir, fn), thr, sp) for name, (fn, thr, sp) in SPECS.items()}
    lex = LexicalMap()
    items = []
    for k, (dep, head, tok, tgt) in enumerate(meta["trees"]):
        g = dependencyGraph(dep, head, tok, tgt)
        concept, depth, relation, ok = g.collect_concepts_and_relations()
        assert ok
        cp_seq, t2i, i2t = lex.get(concept, vocabs['predictable_token'])
        t2i = {c: int(v) for c, v in meta["local_token2idx"][k].items()}       # the fixture's (hash-seed independent) copy ids
        items.append({'concept': concept, 'depth': depth, 'relation': relation, 'token': tgt, 'cp_seq': cp_seq,
                      'token2idx': t2i, 'idx2token': {v: c for c, v in t2i.items()}})
    out = {"settings": np.array(SETTINGS)}
    for k, (rate, seed) in enumerate(SETTINGS):
        random.seed(seed)
        b = rdata.batchify(items, vocabs, unk_rate=rate)
        for key in ("concept", "token_in", "token_out", "cp_seq"):
            out["%d/%s" % (k, key)] = np.asarray(b[key])
        print(rate, seed, "unk concepts", int((np.asarray(b['concept']) == vocabs['concept'].unk_idx).sum()),
              "unk tokens", int((np.asarray(b['token_in']) == vocabs['token'].unk_idx).sum()))
    np.savez_compressed(os.path.join(HERE, "unk_dep_dev.npz"), **out)
    print("unk_dep_dev.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "unk_dep_dev.npz")) / 1024))


if __name__ == "__main__":
    main()
