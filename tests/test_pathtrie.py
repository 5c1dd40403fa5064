"""Host trie builder (libgtos_host.so, include/gtos_host.h) against a direct Python construction of the same tries."""
import numpy as np
import pytest
import torch

from gtos_amd.pathtrie import build_path_trie, CHUNK


def _random_bank(seed, R, L, V, closed=True):
    """Label paths with heavy prefix sharing: paths out of a random tree (prefix-closed) plus a few random ones."""
    rng = np.random.RandomState(seed)
    seqs = {(int(rng.randint(1, V)),)}
    while len(seqs) < R:
        base = list(seqs)[rng.randint(len(seqs))]
        if len(base) < L and rng.rand() < 0.8:
            seqs.add(base + (int(rng.randint(1, V)),))
        else:
            seqs.add(tuple(int(v) for v in rng.randint(1, V, size=rng.randint(1, L + 1))))
    seqs = list(seqs)
    rng.shuffle(seqs)
    Lm = max(len(s) for s in seqs)
    bank = np.zeros((Lm, R), np.int64)
    for r, s in enumerate(seqs):
        bank[:len(s), r] = s
    return seqs, torch.from_numpy(bank), torch.tensor([len(s) for s in seqs])


def _check(seqs, trie, chunk=CHUNK):
    R = len(seqs)
    N = sum(len(s) for s in seqs)
    assert (trie.R, trie.N, trie.L) == (R, N, max(len(s) for s in seqs))
    assert trie.batch_sizes == [sum(len(s) > t for s in seqs) for t in range(trie.L)]
    order = trie.seq_order.tolist()
    assert sorted(order) == list(range(R))
    # packed order: length descending, then lexicographic
    keys = [(-len(seqs[s]), seqs[s]) for s in order]
    assert keys == sorted(keys)
    assert [order[p] for p in trie.seq_pos.tolist()] == list(range(R))
    offs = np.concatenate([[0], np.cumsum(trie.batch_sizes)])
    for side, rows_node, rev in ((trie.pf, trie.row_pf.tolist(), False), (trie.sf, trie.row_sf.tolist(), True)):
        strings = [tuple(reversed(s)) for s in seqs] if rev else seqs
        want_nodes = sorted({s[:k + 1] for s in strings for k in range(len(s))}, key=lambda p: (len(p), p))
        assert side.n_nodes == len(want_nodes)
        node_id = {p: i for i, p in enumerate(want_nodes)}
        lo = side.level_off
        assert lo == [sum(len(p) <= k for p in want_nodes) for k in range(trie.L + 1)]
        assert side.tok.tolist() == [p[-1] for p in want_nodes]
        assert side.par.tolist() == [node_id[p[:-1]] if len(p) > 1 else side.n_nodes for p in want_nodes]
        co = side.child_off.view(-1, 2).tolist()
        par = side.par.tolist()
        for u, (a, b) in enumerate(co):
            kids = [v for v in range(side.n_nodes) if par[v] == u] if side.n_nodes < 3000 else None
            if kids is not None:
                assert kids == list(range(a, b)), (u, a, b, kids)
        # node of every packed row
        for m, s in enumerate(order):
            st = strings[s]
            for t in range(len(st)):
                p = offs[t] + m
                key = st[:len(st) - t] if rev else st[:t + 1]      # suffix trie: tokens t..len-1 reversed = first len-t reversed tokens
                assert rows_node[p] == node_id[key]
        # CSR rows + chunks
        rows = side.rows.tolist()
        cn, cs, cc, sl = side.chunk_node.tolist(), side.chunk_start.tolist(), side.chunk_cnt.tolist(), side.chunk_slot.tolist()
        seen = [[] for _ in range(side.n_nodes)]
        nch = [0] * side.n_nodes
        for c in range(len(cn)):
            assert 0 < cc[c] <= chunk
            seen[cn[c]] += rows[cs[c]:cs[c] + cc[c]]
            nch[cn[c]] += 1
        heavy = side.heavy_node.tolist()
        for u in range(side.n_nodes):
            assert sorted(seen[u]) == [p for p in range(N) if rows_node[p] == u] if N < 4000 else len(seen[u]) > 0
        for c in range(len(cn)):
            assert (sl[c] >= 0) == (nch[cn[c]] > 1)
            if sl[c] >= 0:
                assert heavy[sl[c]] == cn[c]
        assert sorted(r for u in seen for r in u) == list(range(N))


@pytest.mark.parametrize("seed,R,L,V", [(1, 1, 1, 5), (2, 40, 4, 6), (3, 300, 8, 5), (4, 500, 8, 300), (5, 700, 11, 9)])
def test_pathtrie_matches_python_construction(seed, R, L, V):
    seqs, bank, length = _random_bank(seed, R, L, V)
    _check(seqs, build_path_trie(bank, length, chunk=8), chunk=8)


def test_pathtrie_on_synthetic_amr_bank_is_prefix_closed():
    from gtos_amd import synth
    batch, stats = synth.make_batch(2, 6, 30, 8)
    trie = build_path_trie(batch["relation_bank"], batch["relation_length"])
    assert trie.pf.n_nodes == stats["R"]                      # paths out of BFS trees: every prefix is itself a path
    assert stats["R"] <= trie.sf.n_nodes <= 1.2 * stats["R"]
    assert trie.N == int(batch["relation_length"].sum())
    moved = trie.to("cpu")
    assert moved.pf.level_off == trie.pf.level_off and torch.equal(moved.row_sf, trie.row_sf)


def test_pathtrie_rejects_bad_lengths():
    with pytest.raises(ValueError):
        build_path_trie(torch.ones(3, 4, dtype=torch.int64), torch.tensor([1, 0, 2, 3]))
    with pytest.raises(ValueError):
        build_path_trie(torch.ones(3, 4, dtype=torch.int64), torch.tensor([1, 4, 2, 3]))


# ------------------------------------------------------------------------------------------------ device builder (torch ops)
def _same_object(a, b, path=""):
    """every public attribute of two index objects: tensors equal in dtype, shape and value, everything else ==."""
    bad = []
    for k in sorted(set(vars(a)) | set(vars(b))):
        if k.startswith("_"):
            continue
        x, y = vars(a).get(k), vars(b).get(k)
        if isinstance(x, torch.Tensor):
            if not (isinstance(y, torch.Tensor) and x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)):
                bad.append(path + k)
        elif hasattr(x, "__dict__") and not isinstance(x, (int, float, list, tuple)):
            bad += _same_object(x, y, path + k + ".")
        elif x != y:
            bad.append(path + k)
    return bad


@pytest.mark.parametrize("seed,R,L,V", [(1, 1, 1, 5), (2, 40, 4, 6), (3, 300, 8, 5), (4, 500, 8, 250), (5, 700, 7, 9)])
@pytest.mark.parametrize("chunk", [8, 64])
def test_device_trie_builder_equals_the_host_builder(seed, R, L, V, chunk):
    """gtos_amd.pathtrie_device (torch ops, here on CPU tensors) against csrc_host/pathtrie.cpp: every array of both tries, the
    packed order, the row -> node maps, the chunk lists and the derived backward indices (sum_idx, multi_ranges, wave_off)."""
    from gtos_amd.pathtrie_device import build_path_trie_device
    seqs, bank, length = _random_bank(seed, R, L, V)
    host, dev_ = build_path_trie(bank, length, chunk=chunk), build_path_trie_device(bank, length, chunk=chunk)
    assert _same_object(host, dev_) == []
    _check(seqs, dev_, chunk=chunk)


def test_device_trie_builder_on_an_amr_bank_and_its_limits():
    from gtos_amd import synth
    from gtos_amd.pathtrie_device import build_path_trie_device
    batch, _ = synth.make_batch(3, 6, 40, 8)
    assert _same_object(build_path_trie(batch["relation_bank"], batch["relation_length"]),
                        build_path_trie_device(batch["relation_bank"], batch["relation_length"])) == []
    wide = torch.cat([batch["relation_bank"], torch.zeros(3, batch["relation_bank"].shape[1], dtype=torch.int64)])   # L > 8 rows, paths <= 8
    assert _same_object(build_path_trie(wide, batch["relation_length"]), build_path_trie_device(wide, batch["relation_length"])) == []
    with pytest.raises(ValueError):                                       # label ids the one-byte keys cannot hold
        build_path_trie_device(torch.full((2, 3), 300, dtype=torch.int64), torch.tensor([1, 2, 2]))
    with pytest.raises(ValueError):                                       # a path of 9 labels
        build_path_trie_device(torch.ones(9, 2, dtype=torch.int64), torch.tensor([9, 1]))
    with pytest.raises(ValueError):
        build_path_trie_device(torch.ones(3, 4, dtype=torch.int64), torch.tensor([1, 0, 2, 3]))


def test_device_trie_builder_equals_the_host_builder_at_c2_size():
    """The bank bench.py trains on (64 graphs, 434,624 paths, 2.5 M rows): both builders, every array."""
    from gtos_amd import synth
    from gtos_amd.pathtrie_device import build_path_trie_device
    batch, st = synth.make_config_batch("C2", rank=0, B=64)
    host = build_path_trie(batch["relation_bank"], batch["relation_length"])
    assert host.R == st["R"] and host.pf.n_heavy > 0 and host.sf.n_multi > 0
    assert _same_object(host, build_path_trie_device(batch["relation_bank"], batch["relation_length"])) == []


# ------------------------------------------------------------------------------------------------ staged GPU builder, emulated
class _EmulBackend(object):
    """oracle/trie_emul.cpp: the per-thread stages of csrc/trie_kernels.h (the code the HIP kernels wrap) as serial host loops."""

    def __init__(self, order=0):
        import ctypes
        from oracle.build_emul import build
        out = build()
        self.lib = ctypes.CDLL(out)
        P, I, L_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        self.lib.gtos_trie_emul_phase_a.argtypes = [I, L_, P, P, P, P, P, P]
        self.lib.gtos_trie_emul_phase_b.argtypes = [L_, L_, I, I, I, I, P, P, P, P]
        self.lib.gtos_trie_emul_set_order.argtypes = [ctypes.c_uint64]
        self.lib.gtos_trie_emul_set_order.restype = None
        self.lib.gtos_trie_emul_set_order(order)        # 0: ascending thread order, 1: descending, > 1: a seeded permutation

    def phase_a(self, L, R, N, bank, length, common, pf, sf, sizes):
        from gtos_amd.pathtrie_hip import _table, _COMMON, _SIDE
        assert self.lib.gtos_trie_emul_phase_a(L, R, bank.data_ptr(), length.data_ptr(), _table(_COMMON, common), _table(_SIDE, pf),
                                               _table(_SIDE, sf), sizes.data_ptr()) == 0

    def phase_b(self, R, N, n_pf, n_sf, chunk, rows_per_wave, common, pf, sf, sizes):
        from gtos_amd.pathtrie_hip import _table, _COMMON, _SIDE
        assert self.lib.gtos_trie_emul_phase_b(R, N, n_pf, n_sf, chunk, rows_per_wave, _table(_COMMON, common), _table(_SIDE, pf), _table(_SIDE, sf),
                                               sizes.data_ptr()) == 0


@pytest.mark.parametrize("seed,R,L,V", [(1, 1, 1, 5), (2, 40, 4, 6), (3, 300, 8, 5), (4, 500, 8, 250), (6, 900, 6, 12)])
@pytest.mark.parametrize("chunk", [8, 64])
def test_staged_gpu_trie_builder_stages_equal_the_host_builder(seed, R, L, V, chunk):
    """The stage code of the HIP trie builder (csrc/trie_kernels.h) driven by the product's Python glue (gtos_amd/pathtrie_hip.py)
    with the sorts / scans emulated on the host: every array equals the host builder's."""
    from gtos_amd.pathtrie_hip import build_path_trie_staged
    seqs, bank, length = _random_bank(seed, R, L, V)
    host = build_path_trie(bank, length, chunk=chunk)
    staged = build_path_trie_staged(bank, length, _EmulBackend(), chunk=chunk)
    assert _same_object(host, staged) == []
    _check(seqs, staged, chunk=chunk)


def test_staged_gpu_trie_builder_stages_at_c2_size_and_limits():
    from gtos_amd import synth
    from gtos_amd.pathtrie_hip import build_path_trie_staged
    batch, st = synth.make_config_batch("C2", rank=0, B=64)
    host = build_path_trie(batch["relation_bank"], batch["relation_length"])
    staged = build_path_trie_staged(batch["relation_bank"], batch["relation_length"], _EmulBackend(), n_rows=host.N)
    assert _same_object(host, staged) == []
    with pytest.raises(ValueError):                                       # label ids the one-byte keys cannot hold
        build_path_trie_staged(torch.full((2, 3), 300, dtype=torch.int64), torch.tensor([1, 2, 2]), _EmulBackend())
    with pytest.raises(ValueError):                                       # a path of 9 labels
        build_path_trie_staged(torch.ones(9, 2, dtype=torch.int64), torch.tensor([9, 1]), _EmulBackend())


# ------------------------------------------------------------------------------------------------ relation index (host)
@pytest.mark.parametrize("B", [5, 8])
def test_relation_index_groups_pairs_by_type(B):
    from gtos_amd import synth
    from gtos_amd.relindex import build_relation_index
    batch, st = synth.make_batch(3, B, 12, 6)
    rel, R = batch['relation'], batch['relation_bank'].shape[1]
    ix = build_relation_index(rel, R, chunk=4)
    n = rel.shape[0]
    flat = rel.reshape(-1)
    occ = torch.bincount(flat, minlength=R)
    single = occ[rel] == 1                                               # bit 31 flags the ids of types that occur once
    for got, want, flag in ((ix.idx_q, rel.permute(1, 2, 0), single.permute(1, 2, 0)), (ix.idx_k, rel.permute(0, 2, 1), single.permute(0, 2, 1))):
        assert torch.equal(got.long() & 0x7fffffff, want) and torch.equal(got < 0, flag)
    ps = ix.pair_sorted.long()
    assert sorted(ps.tolist()) == list(range(flat.numel()))
    xcd_of = lambda p_: ((p_ % B) // (B // 8)) if B % 8 == 0 else (p_ % B) % 8
    xo = ix.xcd_off.tolist()
    assert xo[0] == 0 and xo[-1] == ix.nchunks and xo == sorted(xo)
    home_of = lambda c_: max(x for x in range(8) if xo[x] <= c_)
    seen, keys, loads, roaming = {}, [], [0] * 8, 0
    for c in range(ix.nchunks):
        t, s, k, sl = int(ix.chunk_type[c]), int(ix.chunk_start[c]), int(ix.chunk_count[c]), int(ix.chunk_slot[c])
        assert 0 <= k <= (16 if int(occ[t]) > 4 else 4) and all(int(flat[p]) == t for p in ps[s:s + k])   # heavy: chunks of 4 x 4
        seen.setdefault(t, []).extend(ps[s:s + k].tolist())
        assert int(occ[t]) != 1                                          # singleton types have no chunk
        assert (sl >= 0) == (int((flat == t).sum()) > 4)
        if sl >= 0:
            assert int(ix.heavy_types[sl]) == t
        first = int(ps[min(s, flat.numel() - 1)])
        gb = first % B
        x = home_of(c)
        owners = {xcd_of(int(p_)) for p_ in ps[s:s + max(k, 1)]}
        if len(owners) == 1:
            assert owners == {x}                                          # a chunk whose pairs live on one XCD is walked there
        else:
            roaming += 1
        loads[x] += (k + 3) // 4 + 1
        # inside an XCD: the long chunks (> 8 pairs) first, longest first, then the short ones in (graph, key row) order
        keys.append((x, 0, -k, 0) if k > 8 else (x, 1, gb, first // (n * B)))
    assert set(seen) == {t for t in range(R) if int(occ[t]) != 1}       # every other type has a chunk, even one without pairs
    for t, got in seen.items():
        assert sorted(got) == torch.nonzero(flat == t).flatten().tolist()
    assert keys == sorted(keys)
    if roaming:        # chunks that span XCDs go where the fewest load rounds are: greedy, longest first -> no XCD far above the mean
        assert max(loads) <= sum(loads) / 8.0 + 2 * (16 // 4 + 1) or roaming < 8
    with pytest.raises(ValueError):
        build_relation_index(rel, R - 1)


# ------------------------------------------------------------------------------------------------ the trie algebra, on the CPU
def _trie_relation_encoder(ref, trie, double=True):
    """The evaluation order of gtos_amd.gru.TrieBiGRUFn restated with plain torch ops on the oracle's parameters: layer 0 once
    per prefix / suffix trie node, layer-1 input gates as Gf[prefix node] + Gb[suffix node], recurrences over the packed
    rows, final states un-permuted, out_proj.  Independent of the HIP kernels: this is the identity the kernels implement."""
    from oracle.gtos_oracle import gru_cell
    import torch.nn.functional as F
    P = lambda n, l, s="": getattr(ref.rnn, "%s_l%d%s" % (n, l, s))
    hs = ref.hidden_size
    sides = (trie.pf, trie.sf)
    Y0 = []
    for d, side in enumerate(sides):
        suf = "_reverse" if d else ""
        x = ref.rel_embed(side.tok)                                        # [nodes, rel_dim]
        xg = F.linear(x, P("weight_ih", 0, suf), P("bias_ih", 0, suf))
        H = torch.zeros(side.n_nodes + 1, hs, dtype=x.dtype)
        for k in range(trie.L):
            lo, hi = side.level_off[k], side.level_off[k + 1]
            if hi > lo:
                hnew = gru_cell(xg[lo:hi], H[side.par_long[lo:hi]], P("weight_hh", 0, suf), P("bias_hh", 0, suf))
                H = torch.cat([H[:lo], hnew, H[hi:]])                      # functional update (autograd-friendly)
        Y0.append(H[:side.n_nodes])
    bs, offs = trie.batch_sizes, [0]
    for a in bs:
        offs.append(offs[-1] + a)
    R = trie.R
    fin = []
    for d in (0, 1):
        suf = "_reverse" if d else ""
        w_ih = P("weight_ih", 1, suf)
        Gf = F.linear(Y0[0], w_ih[:, :hs])                                  # per prefix node
        Gb = F.linear(Y0[1], w_ih[:, hs:])                                  # per suffix node
        xg = Gf[trie.row_pf.long()] + Gb[trie.row_sf.long()] + P("bias_ih", 1, suf)     # [N, 3h] packed rows
        h = torch.zeros(R, hs, dtype=xg.dtype)
        for t in (range(trie.L) if d == 0 else range(trie.L - 1, -1, -1)):
            A, off = bs[t], offs[t]
            hn = gru_cell(xg[off:off + A], h[:A], P("weight_hh", 1, suf), P("bias_hh", 1, suf))
            h = torch.cat([hn, h[A:]])
        fin.append(h)
    fin = torch.cat(fin, 1)[trie.seq_pos]                                   # packed order -> bank order
    return ref.out_proj(fin)


@pytest.mark.parametrize("kind", ["amr", "random"])
def test_trie_evaluation_is_the_same_function_as_the_reference_gru(kind):
    """RelationEncoder (generator/encoder.py:66-119, via the pinned oracle) == the trie evaluation, outputs and every
    parameter gradient, in fp64 on the CPU (dropout 0)."""
    from oracle import gtos_oracle as O
    if kind == "amr":
        from gtos_amd import synth
        batch, _ = synth.make_batch(5, 4, 14, 6)
        bank, length = batch["relation_bank"], batch["relation_length"]
        V = 90
    else:
        seqs, bank, length = _random_bank(11, 120, 7, 9)
        V = 12
    torch.manual_seed(3)
    ref = O.RelationEncoder(O.VocabSpec(V, 0), 10, 24, 16, 2, 0.0).double()
    trie = build_path_trie(bank, length, chunk=8)
    want = ref(bank, length)
    wout = torch.randn_like(want)
    (want * wout).sum().backward()
    g_want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    ref.zero_grad()
    got = _trie_relation_encoder(ref, trie)
    torch.testing.assert_close(got, want, rtol=1e-10, atol=1e-12)
    (got * wout).sum().backward()
    for k, p in ref.named_parameters():
        torch.testing.assert_close(p.grad, g_want[k], rtol=1e-8, atol=1e-11, msg=lambda m, k=k: "%s: %s" % (k, m))


# ------------------------------------------------------------------------------------------------ property-based (hypothesis)
def test_pathtrie_and_relation_index_invariants_property_based():
    from hypothesis import given, settings, strategies as st
    from gtos_amd.relindex import build_relation_index

    seqs_st = st.lists(st.lists(st.integers(1, 6), min_size=1, max_size=5).map(tuple), min_size=1, max_size=40, unique=True)

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(seqs_st, st.integers(1, 5))
    def check_trie(seqs, chunk):
        Lm = max(len(s) for s in seqs)
        bank = torch.zeros(Lm, len(seqs), dtype=torch.int64)
        for r, s in enumerate(seqs):
            bank[:len(s), r] = torch.tensor(s)
        _check(list(seqs), build_path_trie(bank, torch.tensor([len(s) for s in seqs]), chunk=chunk), chunk=chunk)

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(st.integers(1, 5), st.integers(1, 9), st.integers(1, 12), st.integers(0, 10 ** 6), st.integers(1, 4))
    def check_index(n, B, R, seed, chunk):
        g = torch.Generator().manual_seed(seed)
        rel = torch.randint(0, R, (n, n, B), generator=g)
        ix = build_relation_index(rel, R, chunk=chunk)
        flat = rel.reshape(-1)
        occ = torch.bincount(flat, minlength=R)
        assert torch.equal(ix.idx_q.long() & 0x7fffffff, rel.permute(1, 2, 0)) and torch.equal(ix.idx_q < 0, (occ[rel] == 1).permute(1, 2, 0))
        ps = ix.pair_sorted.long()
        got = {}
        for c in range(ix.nchunks):
            t, s, k = int(ix.chunk_type[c]), int(ix.chunk_start[c]), int(ix.chunk_count[c])
            assert k <= (4 * chunk if int(occ[t]) > chunk else chunk)
            got.setdefault(t, []).extend(ps[s:s + k].tolist())
            assert (int(ix.chunk_slot[c]) >= 0) == (int(occ[t]) > chunk)
        assert set(got) == {t for t in range(R) if int(occ[t]) != 1}
        for t, pairs in got.items():
            assert sorted(pairs) == torch.nonzero(flat == t).flatten().tolist()
        assert int(ix.xcd_off[-1]) == ix.nchunks

    check_trie()
    check_index()
