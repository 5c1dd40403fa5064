"""Golden batch compositions of the reference's GENERATOR DataLoader (generator/data.py:269-316).

  loader_amr_synth.json   sizes of 1500 synthetic items (number of concepts, number of tokens: all the batching policy looks
                          at) and, for four (batch_size, for_train, seed) settings, the list of batches, each a list of item
                          indices, exactly as DataLoader.__iter__ forms and orders them after ``random.seed(seed)``.

The reference loader runs unmodified on a JSON file of stub items written to a temporary directory; ``batchify`` is
replaced by a function that returns the items' indices (the tensor work is irrelevant to the policy).

Run in the build container only:  python tests/golden/make_golden_loader_amr.py
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
sys.path.insert(0, os.path.join(REF, "generator"))
import data as rdata  # noqa: E402

SETTINGS = [(20000, True, 1234), (66666, True, 7), (30000, False, 0), (2000000, True, 3)]


class _Lex(object):
    def get(self, concept, vocab):
        return concept, {}, {}


def main():
    rng = random.Random(20260927)
    sizes = []
    for _ in range(1500):
        n = max(1, min(120, int(rng.lognormvariate(3.0, 0.6))))       # concepts per graph: a few .. ~100
        m = max(1, int(n * rng.uniform(0.8, 2.5)))                    # target tokens
        sizes.append([n, m])
    items = [{"id": i, "concept": ["c"] * n, "token": ["t"] * m} for i, (n, m) in enumerate(sizes)]
    rdata.batchify = lambda data, vocabs, unk_rate, train: [d["id"] for d in data]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "items.json")
        with open(path, "w", encoding="utf8") as fo:
            json.dump(items, fo)
        for batch_size, train, seed in SETTINGS:
            dl = rdata.DataLoader({'predictable_token': None}, _Lex(), path, batch_size, train)
            random.seed(seed)
            batches = list(iter(dl))
            out.append({"batch_size": batch_size, "train": train, "seed": seed, "n_examples": len(dl.data), "batches": batches})
            print(batch_size, train, seed, len(batches), [len(b) for b in batches[:6]])
    with open(os.path.join(HERE, "loader_amr_synth.json"), "w") as fo:
        json.dump({"sizes": sizes, "runs": out}, fo)
    print("loader_amr_synth.json %.1f KB" % (os.path.getsize(os.path.join(HERE, "loader_amr_synth.json")) / 1024))


if __name__ == "__main__":
    main()
