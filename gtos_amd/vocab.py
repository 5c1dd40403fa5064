"""Vocabulary files and batch tensorisation (used by the inference path, rank 2).

File format and semantics of /root/reference/generator/data.py:12-110 (the translator's copy is identical):
``token<TAB>count`` per line; ids are [<PAD>, <UNK>] + specials + every token whose count reaches the threshold, in
file order; ``priority`` keeps the raw count of every token; ``coverage`` is the kept fraction of the token mass.
"""
import threading

import numpy as np
import torch

PAD, UNK = '<PAD>', '<UNK>'
CLS = '<CLS>'
STR, END = '<STR>', '<END>'
SEL, rCLS, TL = '<SELF>', '<rCLS>', '<TL>'

# (file-name attribute of the checkpoint's args, min count, specials): generator/work.py:78-84, generator/train.py:95-101
VOCAB_SPECS = {
    'concept': ('concept_vocab', 5, [CLS]),
    'token': ('token_vocab', 5, [STR, END]),
    'predictable_token': ('predictable_token_vocab', 5, [END]),
    'token_char': ('token_char_vocab', 100, [STR, END]),
    'concept_char': ('concept_char_vocab', 100, [STR, END]),
    'relation': ('relation_vocab', 5, [CLS, rCLS, SEL, TL]),
}


class Vocab(object):
    """Same public surface as the reference class: size, padding_idx, unk_idx, token2idx, idx2token, priority, coverage."""

    def __init__(self, filename, min_occur_cnt, specials=None):
        tokens = [PAD, UNK] + list(specials or [])
        self._priority = {}
        total = kept = 0
        token = cnt = None
        with open(filename) as f:
            for raw in f:
                fields = raw.strip().split('\t')
                try:
                    if len(fields) != 2:
                        raise ValueError(raw)
                    token, cnt = fields[0], int(fields[1])
                    total += cnt
                except ValueError:
                    # The reference prints a malformed line and then falls through with the PREVIOUS line's token and
                    # count (data.py:19-28), i.e. that token is entered once more.  Vocabulary sizes fix the embedding
                    # shapes of existing checkpoints, so the quirk is kept.
                    if cnt is None:
                        raise ValueError("%s: malformed first line %r" % (filename, raw))
                if cnt >= min_occur_cnt:
                    tokens.append(token)
                    kept += cnt
                self._priority[token] = cnt
        self.coverage = kept / total if total else 0.0
        self._idx2token = tokens
        self._token2idx = {}
        for i, t in enumerate(tokens):                        # a repeated token keeps its LAST id, like dict(zip(...))
            self._token2idx[t] = i
        self._padding_idx = self._token2idx[PAD]
        self._unk_idx = self._token2idx[UNK]

    def priority(self, x):
        return self._priority.get(x, 0)

    @property
    def size(self):
        return len(self._idx2token)

    @property
    def unk_idx(self):
        return self._unk_idx

    @property
    def padding_idx(self):
        return self._padding_idx

    def idx2token(self, x):
        if isinstance(x, list):
            return [self.idx2token(i) for i in x]
        return self._idx2token[x]

    def token2idx(self, x):
        if isinstance(x, list):
            return [self.token2idx(i) for i in x]
        return self._token2idx.get(x, self._unk_idx)


def load_vocabs(args_or_dir):
    """The six vocabularies of a run: from a checkpoint's ``args`` namespace (attributes ``concept_vocab`` ...) or from
    a directory holding files with those names."""
    import os
    out = {}
    for name, (attr, min_cnt, specials) in VOCAB_SPECS.items():
        path = os.path.join(args_or_dir, attr) if isinstance(args_or_dir, str) else getattr(args_or_dir, attr)
        out[name] = Vocab(path, min_cnt, specials)
    return out


def lists_to_tensor(xs, vocab=None, local_vocabs=None, unk_rate=0., rng=None):
    """Ragged token lists -> int64 [max_len, batch], padded (data.py:76-98).  ``local_vocabs[i]`` (token -> id, the
    per-graph copy vocabulary) takes precedence over ``vocab``.  ``unk_rate`` > 0 is the reference's training-time noise
    (data.py:84-85; train.sh --unk_rate 0.33): every token, in sequence-major order, becomes <UNK> when
    ``rng.random() < unk_rate`` -- drawn BEFORE the local-vocabulary lookup, from ``rng`` (default: the ``random`` module,
    like the reference), so under the same generator state the noise pattern is the reference's.  (The reference draws a
    number per token even at rate 0; this function only draws when the rate is positive.)"""
    pad = vocab.padding_idx if vocab is not None else 0
    width = max(len(x) for x in xs)
    noisy = vocab is not None and unk_rate > 0.
    if noisy and rng is None:
        import random as rng
    # the lookups go straight to the vocabulary's dict when it has one (Vocab, synth.SynthVocab): a method call per token is what
    # this function costs (6.4 k tokens per call at C2, 10 calls per batch)
    table = getattr(vocab, "_token2idx", None) if vocab is not None else None
    unk = getattr(vocab, "_unk_idx", None) if table is not None else None
    if table is not None and unk is None:
        unk = vocab.token2idx(object())           # whatever this vocabulary answers for a token it has never seen
    out = np.full((len(xs), width), pad, dtype=np.int64)
    for i, x in enumerate(xs):
        if vocab is None:
            ids = list(x)
        else:
            local = local_vocabs[i] if local_vocabs is not None else None
            if noisy:
                ids = []
                for w in x:
                    if rng.random() < unk_rate:
                        ids.append(vocab.unk_idx)
                    elif local is not None and w in local:
                        ids.append(local[w])
                    else:
                        ids.append(vocab.token2idx(w))
            elif table is not None:
                if local:
                    ids = [local[w] if w in local else table.get(w, unk) for w in x]
                else:
                    ids = [table.get(w, unk) for w in x]
            else:
                ids = [local[w] if (local is not None and w in local) else vocab.token2idx(w) for w in x]
        if ids:
            out[i, :len(ids)] = ids
    return torch.from_numpy(np.ascontiguousarray(out.T))


_CHAR_ROWS = {}       # (id(vocab), max_string_len) -> [vocab, {string: id row}, rows, rows as one array, lock]
_CHAR_ROWS_LOCK = threading.Lock()


def strings_to_char_tensor(xs, vocab, max_string_len=20):
    """Ragged lists of strings -> int64 [max_len, batch, max_string_len + 2] of <STR> chars <END> ids (data.py:100-112).
    The id row of a string is computed once per (vocabulary, width) and kept (bounded): batch assembly runs on loader threads
    beside the training loop, and every Python-level loop it avoids is GIL time the launch thread gets back.

    Thread safety (loader THREADS share this cache: ``Prefetcher(loader.thunks(), workers > 1)``, ``bench.py --loader threads``):
    hits are plain dict reads; a miss builds its row outside the lock, then appends it and only THEN publishes its index under
    the entry's lock, so no reader can see an index whose row does not exist yet and two strings can never take the same index;
    the bounded-size reset and the growth of the gather table happen under the same lock, and a call works on the (index, rows)
    objects it fetched at its start -- a reset by another thread swaps in new objects instead of clearing the ones in use."""
    width = max(len(x) for x in xs)
    key = (id(vocab), max_string_len)
    with _CHAR_ROWS_LOCK:
        ent = _CHAR_ROWS.get(key)
        if ent is None or ent[0] is not vocab or len(ent[2]) > 2000000:
            ent = _CHAR_ROWS[key] = [vocab, {}, [], None, threading.Lock()]
    _, index, rows, _, lock = ent

    def ix(z):
        k = index.get(z)
        if k is None:
            chars = list(z[:max_string_len])
            row = vocab.token2idx([STR] + chars + [END]) + [vocab.padding_idx] * (max_string_len - len(chars))
            with lock:
                k = index.get(z)
                if k is None:
                    rows.append(row)
                    k = index[z] = len(rows) - 1
        return k
    pad_ix = ix(PAD)
    # one small index matrix from the Python side (a dict lookup per string), the [B, width, chars] tensor by ONE gather from the
    # table of id rows: the nested-list conversion of the full tensor was 22 x the elements
    idx = np.full((len(xs), width), pad_ix, dtype=np.int64)
    for i, x in enumerate(xs):
        if x:
            idx[i, :len(x)] = [ix(z) for z in x]
    need = int(idx.max()) + 1
    table = ent[3]
    if table is None or table.shape[0] < need:
        with lock:
            table = ent[3]
            if table is None or table.shape[0] < need:
                have = 0 if table is None else table.shape[0]
                fresh = np.asarray(rows[have:len(rows)], dtype=np.int64).reshape(-1, max_string_len + 2)
                table = ent[3] = fresh if table is None else np.concatenate([table, fresh])
    return torch.from_numpy(np.ascontiguousarray(table[idx].transpose(1, 0, 2)))


def copy_vocab(concepts, vocab):
    """Per-graph copy vocabulary (extract.py:47-63): every concept that the predictable-token vocabulary does not know
    gets a fresh id after its end.  Returns (cp_seq, token2idx, idx2token).  The reference iterates a ``set``; ids are
    assigned here in first-occurrence order, which is deterministic (any assignment is valid: the ids only link
    cp_seq, token_out and local_idx2token inside one batch)."""
    token2idx, idx2token = {}, {}
    nxt = vocab.size
    for c in concepts:
        if c not in token2idx and vocab.token2idx(c) == vocab.unk_idx:
            token2idx[c] = nxt
            idx2token[nxt] = c
            nxt += 1
    return list(concepts), token2idx, idx2token
