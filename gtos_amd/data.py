"""Batch assembly for the translator flavour: dependency trees -> the batch dict the model consumes.

Counterpart of /root/reference/translator/data.py:126-201 (``batchify``) with the graph work of
translator/dependencyGraph.py:30-74 (BFS node order, all-pairs shortest label paths) done by the C++ host library
(gtos_amd.relbatch, include/gtos_host.h) and the token / character tensors by gtos_amd.vocab.  Bit-exact with the
reference on its own data (tests/test_host_relbatch.py, tests/test_beam_and_vocab.py); the only freedom is the numbering
of the per-graph copy ids, which the reference takes from a ``set`` iteration (extract.py:56-62) and this module assigns
in first-occurrence order.
"""
import torch

from . import relbatch
from .vocab import CLS, rCLS, SEL, TL, STR, END, lists_to_tensor, strings_to_char_tensor, copy_vocab


def relation_special_ids(rel_vocab):
    return (rel_vocab.padding_idx, rel_vocab.token2idx(CLS), rel_vocab.token2idx(rCLS), rel_vocab.token2idx(SEL),
            rel_vocab.token2idx(TL))


def batchify_dependency(trees, vocabs, n_threads=0):
    """trees: list of (dep labels, heads (1-based, 0 = root), source tokens, target tokens) as translator/extract.py's
    IO.read1 yields them.  Returns the dict of translator/data.py:185-201 (int64 tensors, time-major) including
    ``local_idx2token`` / ``local_token2idx``; ``concept_depth`` is the node's position in the source sentence
    (dependencyGraph.py:72-73), concepts are in BFS order from the root."""
    rv = vocabs['relation']
    graphs = []
    for dep, head, tok, tgt in trees:
        ids = rv.token2idx(list(dep))
        rev = rv.token2idx([r + '_r_' for r in dep])
        graphs.append(relbatch.dependency_edges(list(head), ids, rev))
    rel = relbatch.build_relation_batch(graphs, relation_special_ids(rv), path_mode=relbatch.PATH_FIRST, n_threads=n_threads)
    concepts, depths, cps, t2is, i2ts = [], [], [], [], []
    for b, (dep, head, tok, tgt) in enumerate(trees):
        order = rel['order'][b, :len(tok)].tolist()
        conc = [tok[n] for n in order]
        concepts.append(conc)
        depths.append(order)                              # "we just use the sequential order" (dependencyGraph.py:72)
        cp_seq, t2i, i2t = copy_vocab(conc, vocabs['predictable_token'])
        cps.append(cp_seq); t2is.append(t2i); i2ts.append(i2t)
    aug = [[STR] + list(tgt) + [END] for _, _, _, tgt in trees]
    with_cls = [[CLS] + c for c in concepts]
    return {
        'concept': lists_to_tensor(with_cls, vocabs['concept']),
        'concept_char': strings_to_char_tensor(with_cls, vocabs['concept_char']),
        'concept_depth': lists_to_tensor([[0] + d for d in depths]),
        'relation': rel['relation'], 'relation_bank': rel['relation_bank'], 'relation_length': rel['relation_length'],
        'local_idx2token': i2ts, 'local_token2idx': t2is,
        'token_in': lists_to_tensor(aug, vocabs['token'])[:-1],
        'token_char_in': strings_to_char_tensor(aug, vocabs['token_char'])[:-1],
        'token_out': lists_to_tensor(aug, vocabs['predictable_token'], t2is)[1:],
        'cp_seq': lists_to_tensor(cps, vocabs['predictable_token'], t2is),
    }


def read_dependency_file(path):
    """The reference's 4-lines-per-example text format (translator/extract.py:11-45): dependency labels, heads, source
    tokens, target tokens, separated by single spaces."""
    out = []
    with open(path, encoding='utf8') as f:
        lines = [l.rstrip('\n') for l in f]
    for k in range(0, len(lines) - 3, 4):
        dep, head = lines[k].split(' '), [int(x) for x in lines[k + 1].split(' ')]
        tok, tgt = lines[k + 2].split(' '), lines[k + 3].split(' ')
        assert len(dep) == len(head) == len(tok)
        out.append((dep, head, tok, tgt))
    return out


class DependencyLoader(object):
    """Batching policy of translator/data.py:207-264 (``DataLoader``): examples are shuffled and (stably) sorted by size
    ``n_source_tokens**2 + n_target_tokens`` when training, packed greedily until a batch holds ``batch_size`` size units
    (or 257 examples), a trailing batch is kept if it is at least half full (always in evaluation), and the batch order is
    shuffled.  Both shuffles draw from ``rng`` (default: the ``random`` module, like the reference) BEFORE the first batch is
    assembled, so under the same seed the batches and their order are the reference's.  Yields batchify_dependency dicts."""

    def __init__(self, vocabs, filename, batch_size, for_train, rng=None, n_threads=0):
        import random
        self.data = read_dependency_file(filename) if isinstance(filename, str) else list(filename)
        self.vocabs, self.batch_size, self.train = vocabs, batch_size, for_train
        self.rng = rng if rng is not None else random
        self.n_threads = n_threads

    @staticmethod
    def size_of(tree):
        dep, head, tok, tgt = tree
        return len(tok) ** 2 + len(tgt)

    def batch_indices(self):
        idx = list(range(len(self.data)))
        if self.train:
            self.rng.shuffle(idx)
            idx.sort(key=lambda i: self.size_of(self.data[i]))
        batches, units, cur = [], 0, []
        for i in idx:
            units += self.size_of(self.data[i])
            cur.append(i)
            if units >= self.batch_size or len(cur) > 256:
                batches.append(cur)
                units, cur = 0, []
        if not self.train or units > self.batch_size / 2:
            batches.append(cur)
        if self.train:
            self.rng.shuffle(batches)
        return batches

    def __iter__(self):
        for b in self.batch_indices():
            yield batchify_dependency([self.data[i] for i in b], self.vocabs, n_threads=self.n_threads)
