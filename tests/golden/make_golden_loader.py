"""Golden batch compositions of the reference's translator DataLoader on its own dev.txt.

  loader_dep_dev.json   for (batch_size, for_train, seed) settings: the list of batches, each a list of example indices
                        (position in dev.txt), exactly as translator/data.py:DataLoader.__iter__ forms and orders them
                        after ``random.seed(seed)``.

The reference loader runs unmodified; each graph's collect_concepts_and_relations is replaced by a stub that returns the
example's index (the graph work is irrelevant to the batching policy and slow), and ``batchify`` by a function that
returns those indices.

Run in the build container only:  python tests/golden/make_golden_loader.py
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
np.int = int
sys.path.insert(0, os.path.join(REF, "translator"))
import data as rdata  # noqa: E402

SETTINGS = [(2000, True, 1234), (6000, True, 7), (3000, False, 0), (100000, True, 3)]


class _Lex(object):
    def get(self, concept, vocab):
        return concept, {}, {}


def main():
    path = os.path.join(REF, "translator_data", "dev.txt")
    rdata.batchify = lambda res, vocabs, unk_rate: [r['concept'][0] for r in res]
    out = []
    for batch_size, train, seed in SETTINGS:
        dl = rdata.DataLoader({'predictable_token': None}, _Lex(), path, batch_size, train)
        for i, g in enumerate(dl.data):
            g.collect_concepts_and_relations = (lambda i=i: ([i], [], {}, True))
        random.seed(seed)
        batches = list(iter(dl))
        out.append({"batch_size": batch_size, "train": train, "seed": seed, "n_examples": len(dl.data), "batches": batches})
        print(batch_size, train, seed, len(batches), [len(b) for b in batches[:6]])
    sizes = [[len(g.name2concept), len(g.target)] for g in dl.data]        # all the policy looks at (dependencyGraph.__len__)
    with open(os.path.join(HERE, "loader_dep_dev.json"), "w") as fo:
        json.dump({"sizes": sizes, "runs": out}, fo)
    print("loader_dep_dev.json %.1f KB" % (os.path.getsize(os.path.join(HERE, "loader_dep_dev.json")) / 1024))


if __name__ == "__main__":
    main()
