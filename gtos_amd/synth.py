"""Synthetic AMR-shaped / dependency-tree batches in the reference's batch-dict layout.

Host-side only (numpy).  The layout follows what the reference's ``batchify`` emits
(/root/reference/generator/data.py:126-267): ``concept[n,B]`` with a ``<CLS>`` row prepended,
``concept_char[n,B,C]``, ``concept_depth[n,B]``, ``relation[n,n,B]`` (type ids, ``relation[a][c][b]`` =
label path from node c to node a, data.py:164-165), ``relation_bank[L,R]`` / ``relation_length[R]``
(distinct label paths, <=8 labels, longer ones collapse to ``<TL>``, data.py:151-154,166-176),
``token_in/token_out[T,B]``, ``token_char_in[T,B,C]``, ``cp_seq[n-1,B]``.

Everything is drawn from a counter-based splitmix64 stream seeded with ``config*10**6 + graph`` so the
GPU box regenerates bit-identical inputs without torch RNG.
"""
import numpy as np
import torch

PAD, UNK = 0, 1
# relation vocab = [PAD, UNK, CLS, rCLS, SELF, TL] + labels (generator/train.py:92)
REL_CLS, REL_RCLS, REL_SELF, REL_TL = 2, 3, 4, 5
REL_FIRST_LABEL = 6
CONCEPT_CLS = 2            # concept vocab = [PAD, UNK, CLS] + ...
TOK_STR, TOK_END = 2, 3    # token vocab   = [PAD, UNK, STR, END] + ...
PRED_END = 2               # predictable   = [PAD, UNK, END] + ...
CHAR_STR, CHAR_END = 2, 3
MAX_PATH = 8

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    """Vectorised splitmix64: value_i = mix(seed + (i+1)*GOLD)."""

    def __init__(self, seed):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, n):
        with np.errstate(over="ignore"):
            z = self.state + _GOLD * np.arange(1, n + 1, dtype=np.uint64)
            self.state = self.state + _GOLD * np.uint64(n)
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def randint(self, hi, n=None):
        """ints in [0, hi); hi may be an array (then n is ignored)."""
        if n is None:
            hi = np.asarray(hi)
            return (self.uniform(hi.size).reshape(hi.shape) * hi).astype(np.int64)
        return (self.uniform(n) * hi).astype(np.int64)

    def one(self, hi):
        return int(self.randint(hi, 1)[0])


def _zipf_table(n):
    w = 1.0 / np.arange(1, n + 1)
    return np.cumsum(w / w.sum())


def _zipf(rng, cdf, n):
    return np.minimum(np.searchsorted(cdf, rng.uniform(n)), len(cdf) - 1)


def _amr_graph(rng, N, extra_frac, n_labels, lab_cdf):
    """Rooted random tree (+ re-entrancies), every edge doubled with its reverse label
    (generator/AMRGraph.py:76-80).  Returns adjacency [(nbr,label)] in creation order."""
    adj = [[] for _ in range(N)]
    depth = [0] * N
    und = set()

    def add(u, v):
        lab = int(_zipf(rng, lab_cdf, 1)[0])
        adj[u].append((v, REL_FIRST_LABEL + lab))
        adj[v].append((u, REL_FIRST_LABEL + n_labels + lab))
        und.add((min(u, v), max(u, v)))

    for v in range(1, N):
        ok = [u for u in range(v) if depth[u] <= 30]
        u = ok[rng.one(len(ok))]
        depth[v] = depth[u] + 1
        add(u, v)
    want, tries = int(extra_frac * N), 0
    while want > 0 and tries < 50 * N:
        tries += 1
        u, v = rng.one(N), rng.one(N)
        if u == v or (min(u, v), max(u, v)) in und:
            continue
        add(u, v)
        want -= 1
    return adj


def _dep_tree(rng, N, n_labels, lab_cdf):
    """Random projective dependency tree: each word attaches to a node on the right frontier
    (translator/dependencyGraph.py:30-34 doubles each arc with an ``_r_`` reverse label)."""
    adj = [[] for _ in range(N)]
    frontier = [0]
    for v in range(1, N):
        k = rng.one(len(frontier))
        u = frontier[k]
        lab = int(_zipf(rng, lab_cdf, 1)[0])
        adj[u].append((v, REL_FIRST_LABEL + lab))
        adj[v].append((u, REL_FIRST_LABEL + n_labels + lab))
        frontier = frontier[:k + 1] + [v]
    return adj


def _bfs_order(adj, root=0):
    order, depth, seen = [root], [0], {root}
    i = 0
    while i < len(order):
        u = order[i]
        for v, _ in adj[u]:
            if v not in seen:
                seen.add(v)
                order.append(v)
                depth.append(depth[i] + 1)
        i += 1
    assert len(order) == len(adj), "not connected"
    return order, depth


def _all_pairs_label_paths(rng, adj, order, n_alt=1):
    """paths[i][j] = list (len n_alt, possibly repeated) of label tuples of a shortest path i -> j, where
    i,j index ``order``.  Ties are broken by a per-source random neighbour permutation."""
    N = len(adj)
    pos = {v: i for i, v in enumerate(order)}
    out = [[[] for _ in range(N)] for _ in range(N)]
    for _ in range(n_alt):
        for si, s in enumerate(order):
            perm_key = rng.uniform(N)
            path = {s: ()}
            queue = [s]
            qi = 0
            while qi < len(queue):
                u = queue[qi]
                qi += 1
                pu = path[u]
                nb = adj[u]
                if len(nb) > 1:
                    nb = sorted(nb, key=lambda e: perm_key[e[0]])
                for v, lab in nb:
                    if v not in path:
                        path[v] = pu + (lab,)
                        queue.append(v)
            row = out[si]
            for v, p in path.items():
                row[pos[v]].append(p)
    return out


def _canon(p):
    if len(p) == 0:
        return (REL_SELF,)
    if len(p) > MAX_PATH:
        return (REL_TL,)
    return p


DEFAULT_VOCAB = dict(concept=8000, token=12000, predictable_token=10000, relation=86,
                     concept_char=75, token_char=80)


def make_batch(config_id, B, N, T, kind="amr", extra_frac=0.1, train=True, padded=False,
               vocab=None, n_alt=3, first_graph=0):
    """Build one batch.  Returns (batch dict of int64 tensors, stats dict)."""
    vocab = dict(DEFAULT_VOCAB, **(vocab or {}))
    n_labels = (vocab["relation"] - REL_FIRST_LABEL) // 2
    lab_cdf = _zipf_table(n_labels)
    c_cdf = _zipf_table(vocab["concept"] - 3)
    t_cdf = _zipf_table(vocab["token"] - 4)
    p_cdf = _zipf_table(vocab["predictable_token"] - 3)
    CH = 14   # <STR> + up to 12 chars + <END>

    graphs = []
    for g in range(B):
        rng = SplitMix64(config_id * 10 ** 6 + first_graph + g)
        Ng = N if not padded else max(2, N // 2 + rng.one(N - N // 2 + 1))
        Tg = T if not padded else max(2, T // 2 + rng.one(T - T // 2 + 1))
        if kind == "amr":
            adj = _amr_graph(rng, Ng, extra_frac, n_labels, lab_cdf)
            order, depth = _bfs_order(adj)
            depth = [min(d, 31) for d in depth]
        else:
            adj = _dep_tree(rng, Ng, n_labels, lab_cdf)
            order = list(range(Ng))
            depth = list(range(Ng))        # translator: depth := word index (dependencyGraph.py:72-73)
        paths = _all_pairs_label_paths(rng, adj, order, 1 if train else n_alt)
        concept = 3 + _zipf(rng, c_cdf, Ng)
        # copy ids: 80% map to a predictable-vocab id, the rest get graph-local ids >= V_pred
        cp = 3 + _zipf(rng, p_cdf, Ng)
        local = rng.uniform(Ng) < 0.2
        cp[local] = vocab["predictable_token"] + np.arange(int(local.sum()))
        ntok = Tg - 1
        tok = 4 + _zipf(rng, t_cdf, ntok)
        tgt = 3 + _zipf(rng, p_cdf, ntok)
        copy_here = rng.uniform(ntok) < 0.3
        tgt[copy_here] = cp[rng.randint(Ng, ntok)][copy_here]

        def words(count, nchar):
            ln = 3 + rng.randint(10, count)
            ch = 4 + rng.randint(nchar - 4, count * 12).reshape(count, 12)
            w = np.zeros((count, CH), dtype=np.int64)
            w[:, 0] = CHAR_STR
            for r in range(count):
                w[r, 1:1 + ln[r]] = ch[r, :ln[r]]
                w[r, 1 + ln[r]] = CHAR_END
            return w
        graphs.append(dict(N=Ng, T=Tg, paths=paths, depth=depth, concept=concept, cp=cp, tok=tok, tgt=tgt,
                           cchar=words(Ng + 1, vocab["concept_char"]), tchar=words(Tg, vocab["token_char"])))

    n = 1 + max(g["N"] for g in graphs)
    Tm = max(g["T"] for g in graphs)
    concept = np.zeros((n, B), np.int64)
    cchar = np.zeros((n, B, CH), np.int64)
    cdepth = np.zeros((n, B), np.int64)
    tok_in = np.zeros((Tm, B), np.int64)
    tok_out = np.zeros((Tm, B), np.int64)
    tchar = np.zeros((Tm, B, CH), np.int64)
    cp_seq = np.zeros((n - 1, B), np.int64)

    if train:       # type ids 0/1/2 = <CLS>/<rCLS>/<SELF>  (data.py:134-147)
        types = {(REL_CLS,): 0, (REL_RCLS,): 1, (REL_SELF,): 2}
        t_cls, t_rcls, t_self = 0, 1, 2
        rel = np.zeros((n, n, B), np.int64)
    else:           # eval: 0 = <PAD> row, then <CLS>/<rCLS>/<SELF>  (data.py:178-189)
        types = {(PAD,): 0, (REL_CLS,): 1, (REL_RCLS,): 2, (REL_SELF,): 3}
        t_cls, t_rcls, t_self = 1, 2, 3
        rel_lists = []
    tot_len = 0
    for b, g in enumerate(graphs):
        Ng, Tg = g["N"], g["T"]
        concept[0, b] = CONCEPT_CLS
        concept[1:1 + Ng, b] = g["concept"]
        cchar[:1 + Ng, b] = g["cchar"]
        cdepth[1:1 + Ng, b] = g["depth"]
        tok_in[0, b] = TOK_STR
        tok_in[1:Tg, b] = g["tok"]
        tok_out[:Tg - 1, b] = g["tgt"]
        tok_out[Tg - 1, b] = PRED_END
        tchar[:Tg, b] = g["tchar"]
        cp_seq[:Ng, b] = g["cp"]
        # brs[c][a] = path from node c to node a (row 0 / col 0 = <CLS> node); tensor index [a][c][b]
        if train:
            rel[0, 0, b] = t_self
            rel[1:1 + Ng, 0, b] = t_cls
            rel[0, 1:1 + Ng, b] = t_rcls
            for c in range(Ng):
                row = g["paths"][c]
                for a in range(Ng):
                    p = _canon(row[a][0])
                    t = types.get(p)
                    if t is None:
                        t = types[p] = len(types)
                    rel[1 + a, 1 + c, b] = t
        else:
            cells = {}
            cells[(0, 0)] = [t_self]
            for a in range(Ng):
                cells[(1 + a, 0)] = [t_cls]
                cells[(0, 1 + a)] = [t_rcls]
            for c in range(Ng):
                row = g["paths"][c]
                for a in range(Ng):
                    alts = row[a]
                    if len(alts[0]) == 0 or len(alts[0]) > MAX_PATH:
                        alts = alts[:1]
                    ids = []
                    for p in dict.fromkeys(_canon(p) for p in alts):
                        t = types.get(p)
                        if t is None:
                            t = types[p] = len(types)
                        ids.append(t)
                    cells[(1 + a, 1 + c)] = ids
            rel_lists.append(cells)
    if not train:
        K = max(len(v) for cells in rel_lists for v in cells.values())
        rel = np.zeros((n, n, B, K), np.int64)
        for b, cells in enumerate(rel_lists):
            for (a, c), ids in cells.items():
                rel[a, c, b, :len(ids)] = ids
    R = len(types)
    L = max(len(p) for p in types)
    bank = np.zeros((L, R), np.int64)
    blen = np.zeros((R,), np.int64)
    for p, t in types.items():
        bank[:len(p), t] = p
        blen[t] = len(p)
        tot_len += len(p)
    batch = dict(concept=concept, concept_char=cchar, concept_depth=cdepth, relation=rel,
                 relation_bank=bank, relation_length=blen, token_in=tok_in, token_char_in=tchar,
                 token_out=tok_out, cp_seq=cp_seq)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    stats = dict(n=n, B=B, T=Tm, P=n * n * B, R=R, mean_path_len=tot_len / R,
                 tl_frac=float((rel == types.get((REL_TL,), -1)).mean()) if train else 0.0)
    return batch, stats


# BASELINE.json configs.  ``id`` seeds the PRNG.
CONFIGS = {
    "C1": dict(id=1, kind="amr", B=8, N=20, T=16, layers=4, d=256, ff=512, H=8, extra_frac=0.1),
    "C2": dict(id=2, kind="amr", B=64, N=100, T=50, layers=8, d=512, ff=1024, H=8, extra_frac=0.1),
    "C3": dict(id=3, kind="dep", B=64, N=60, T=70, layers=8, d=512, ff=1024, H=8, extra_frac=0.0),
    "C4": dict(id=2, kind="amr", B=64, N=100, T=50, layers=8, d=512, ff=1024, H=8, extra_frac=0.1),
    "C5": dict(id=5, kind="amr", B=32, N=300, T=100, layers=8, d=512, ff=1024, H=8, extra_frac=1.0),
}


def make_config_batch(name, rank=0, train=True, padded=False, B=None):
    c = CONFIGS[name]
    B = B or c["B"]
    return make_batch(c["id"], B, c["N"], c["T"], kind=c["kind"], extra_frac=c["extra_frac"],
                      train=train, padded=padded, first_graph=rank * B)


class SynthVocab(object):
    """String vocabulary of a given size for the inference benchmark / tests: ids 0,1 = <PAD>,<UNK>, then the specials,
    then made-up words (``prefix`` + number) or, for character vocabularies, single characters.  Same surface as
    gtos_amd.vocab.Vocab."""

    def __init__(self, size, specials, prefix="w", chars=False):
        toks = ['<PAD>', '<UNK>'] + list(specials)
        if chars:
            pool = list("abcdefghijklmnopqrstuvwxyz0123456789-_.") + [chr(0x100 + i) for i in range(size)]
            toks += pool[:size - len(toks)]
        else:
            toks += ["%s%d" % (prefix, i) for i in range(size - len(toks))]
        self._idx2token = toks
        self._token2idx = {t: i for i, t in enumerate(toks)}

    size = property(lambda self: len(self._idx2token))
    padding_idx = property(lambda self: 0)
    unk_idx = property(lambda self: 1)

    def idx2token(self, x):
        return [self.idx2token(i) for i in x] if isinstance(x, list) else self._idx2token[x]

    def token2idx(self, x):
        return [self.token2idx(i) for i in x] if isinstance(x, list) else self._token2idx.get(x, 1)


def synth_vocabs(vocab=None):
    """The six vocabularies at DEFAULT_VOCAB sizes with the reference's specials (generator/work.py:78-84).  Words of
    the token and predictable-token vocabularies coincide, so a generated id maps to an in-vocabulary input token."""
    v = dict(DEFAULT_VOCAB, **(vocab or {}))
    return {'concept': SynthVocab(v['concept'], ['<CLS>'], "c"), 'token': SynthVocab(v['token'], ['<STR>', '<END>']),
            'predictable_token': SynthVocab(v['predictable_token'], ['<END>']),
            'token_char': SynthVocab(v['token_char'], ['<STR>', '<END>'], chars=True),
            'concept_char': SynthVocab(v['concept_char'], ['<STR>', '<END>'], chars=True),
            'relation': SynthVocab(v['relation'], ['<CLS>', '<rCLS>', '<SELF>', '<TL>'], "r")}


def make_amr_items(name, count, first_graph=0, vocabs=None):
    """A pool of ``count`` synthetic AMR items of a BASELINE config for the LOADER path (gtos_amd.data.AMRLoader ->
    batchify_amr -> C++ relation batch / tries / index): the same graph family as make_batch (same seeds, same generator),
    but as what the reference's preprocessing leaves in memory -- concept strings in BFS order, depths, token strings -- plus
    the graph itself as (n, root, edges[E,3]) in BFS positions, edges sorted by (source, target), which is what
    data._edges_from_paths recovers from a preprocessed item's length-1 paths.  Returns (items, graphs)."""
    c = CONFIGS[name]
    vocabs = vocabs or synth_vocabs()
    sizes = {k: v.size for k, v in vocabs.items()}
    n_labels = (sizes["relation"] - REL_FIRST_LABEL) // 2
    lab_cdf = _zipf_table(n_labels)
    c_cdf, t_cdf = _zipf_table(sizes["concept"] - 3), _zipf_table(sizes["token"] - 4)
    items, graphs = [], []
    for g in range(count):
        rng = SplitMix64(c["id"] * 10 ** 6 + first_graph + g)
        N, T = c["N"], c["T"]
        if c["kind"] == "amr":
            adj = _amr_graph(rng, N, c["extra_frac"], n_labels, lab_cdf)
            order, depth = _bfs_order(adj)
            depth = [min(d, 31) for d in depth]
        else:
            adj = _dep_tree(rng, N, n_labels, lab_cdf)
            order, depth = _bfs_order(adj)
        pos = {v: i for i, v in enumerate(order)}
        edges = sorted((pos[u], pos[v], lab) for u in range(N) for v, lab in adj[u])
        concept = vocabs["concept"].idx2token((3 + _zipf(rng, c_cdf, N)).tolist())
        token = vocabs["token"].idx2token((4 + _zipf(rng, t_cdf, T - 1)).tolist())
        items.append({"concept": concept, "depth": depth, "token": token})
        graphs.append((N, 0, np.array(edges, dtype=np.int32).reshape(-1, 3)))
    return items, graphs


def dep_vocabs(vocab=None):
    """synth_vocabs() with a relation vocabulary whose strings pair up like the translator's: the arc label of a word (child -> head) is
    ``d<k>`` and its reverse ``d<k>_r_`` (translator/dependencyGraph.py:30-34), with the ids ``_dep_tree`` gives the two directions."""
    vs = synth_vocabs(vocab)
    rv = vs['relation']
    n_labels = (rv.size - REL_FIRST_LABEL) // 2
    toks = list(rv._idx2token)
    for k in range(n_labels):
        toks[REL_FIRST_LABEL + n_labels + k] = "d%d" % k          # child -> head: what extract.py reads from the treebank
        toks[REL_FIRST_LABEL + k] = "d%d_r_" % k                  # head -> child
    rv._idx2token = toks
    rv._token2idx = {t: i for i, t in enumerate(toks)}
    return vs


def make_dep_trees(name, count, first_graph=0, vocabs=None):
    """A pool of ``count`` synthetic dependency trees of a BASELINE config ("C3") for the LOADER path (gtos_amd.data.DependencyLoader ->
    batchify_dependency): the same tree family as make_batch(kind="dep") (same seeds, same generator), as the 4-tuples translator/
    extract.py's reader yields -- (arc labels, 1-based heads with 0 = root, source tokens, target tokens), all strings of ``vocabs``
    (``dep_vocabs()``)."""
    c = CONFIGS[name]
    assert c["kind"] == "dep", name
    vocabs = vocabs or dep_vocabs()
    sizes = {k: v.size for k, v in vocabs.items()}
    n_labels = (sizes["relation"] - REL_FIRST_LABEL) // 2
    lab_cdf = _zipf_table(n_labels)
    c_cdf, p_cdf = _zipf_table(sizes["concept"] - 3), _zipf_table(sizes["predictable_token"] - 3)
    rv = vocabs["relation"]
    trees = []
    for g in range(count):
        rng = SplitMix64(c["id"] * 10 ** 6 + first_graph + g)
        N, T = c["N"], c["T"]
        adj = _dep_tree(rng, N, n_labels, lab_cdf)
        head, dep = [0] * N, [rv.idx2token(REL_FIRST_LABEL + n_labels)] * N        # the root's label is never used (head 0: no arc)
        for u in range(N):
            for v, lab in adj[u]:
                if v > u:                                  # u is v's head (a word attaches to an EARLIER node of the right frontier)
                    head[v] = u + 1
                    dep[v] = rv.idx2token(lab + n_labels)  # the child -> head label; adj[u] holds the head -> child one
        tok = vocabs["concept"].idx2token((3 + _zipf(rng, c_cdf, N)).tolist())
        tgt = vocabs["predictable_token"].idx2token((3 + _zipf(rng, p_cdf, T - 1)).tolist())
        trees.append((dep, head, tok, tgt))
    return trees
