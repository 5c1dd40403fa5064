"""Host graph->relation-tensor path (libgtos_host.so) against golden vectors produced by the reference's own pipeline
(tests/golden/make_golden_host.py) and against networkx on random multi-path graphs.  Integer work: bit-exact."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def rb():
    from gtos_amd import build, relbatch
    build.build_host(verbose=False)
    relbatch.load()
    return relbatch


@pytest.mark.parametrize("bi", [0, 1])
def test_translator_flavour_matches_reference_on_real_dev_data(rb, bi):
    g = load_golden("host_dep_dev")
    p = "b%d/" % bi
    off = g[p + "off"]
    graphs = []
    for k in range(len(off) - 1):
        sl = slice(off[k], off[k + 1])
        graphs.append(rb.dependency_edges(g[p + "heads"][sl].tolist(), g[p + "dep_ids"][sl].tolist(), g[p + "rev_ids"][sl].tolist()))
    out = rb.build_relation_batch(graphs, g["special_ids"].tolist(), path_mode=rb.PATH_FIRST)
    assert torch.equal(out["relation"], torch.from_numpy(g[p + "relation"]))
    assert torch.equal(out["relation_bank"], torch.from_numpy(g[p + "relation_bank"]))
    assert torch.equal(out["relation_length"], torch.from_numpy(g[p + "relation_length"]))
    # translator: concept_depth = [0] + BFS node order (dependencyGraph.py:72-73), 0-padded
    want = torch.from_numpy(g[p + "concept_depth"])[1:].t()
    got = out["order"].clamp(min=0).long()
    assert torch.equal(got, want)
    # uniform choice among alternatives must agree too: shortest paths in a tree are unique
    out2 = rb.build_relation_batch(graphs, g["special_ids"].tolist(), path_mode=rb.PATH_UNIFORM, seed=7)
    assert torch.equal(out2["relation"], out["relation"]) and torch.equal(out2["relation_bank"], out["relation_bank"])


def test_generator_flavour_eval_matches_reference_on_smatch_amrs(rb):
    g = load_golden("host_amr_smatch")
    eo = g["edge_off"]
    graphs = [(int(g["n_nodes"][k]), int(g["roots"][k]), g["edges"][eo[k]:eo[k + 1]]) for k in range(len(g["n_nodes"]))]
    out = rb.build_relation_batch(graphs, g["special_ids"].tolist(), path_mode=rb.PATH_ALL)
    assert torch.equal(out["relation"], torch.from_numpy(g["relation"]))
    assert torch.equal(out["relation_bank"], torch.from_numpy(g["relation_bank"]))
    assert torch.equal(out["relation_length"], torch.from_numpy(g["relation_length"]))
    want_depth = torch.from_numpy(g["concept_depth"])[1:].t()
    assert torch.equal(out["depth"].long(), want_depth)


def _random_graph(rng, n, extra):
    """connected labelled graph with reverse twins, as (n, root, edges) plus the networkx DiGraph built the reference way"""
    import networkx as nx
    G = nx.DiGraph()
    edges = []
    for v in range(n):
        G.add_node(v)

    def add(u, v):
        lab = int(rng.integers(6, 20))
        for (a, b, l) in ((u, v, lab), (v, u, lab + 20)):
            G.add_edge(a, b, label=l)
            edges.append((a, b, l))
    for v in range(1, n):
        add(int(rng.integers(0, v)), v)
    for _ in range(extra):
        u, v = int(rng.integers(0, n)), int(rng.integers(0, n))
        if u != v:
            add(u, v)
    return G, (n, 0, np.array(edges, dtype=np.int32))


def test_all_paths_and_first_path_orders_match_networkx(rb):
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(5)
    special = [0, 2, 3, 4, 5]
    for trial in range(6):
        n = int(rng.integers(5, 14))
        G, graph = _random_graph(rng, n, extra=int(rng.integers(2, 3 * n)))
        out_all = rb.build_relation_batch([graph], special, path_mode=rb.PATH_ALL)
        out_first = rb.build_relation_batch([graph], special, path_mode=rb.PATH_FIRST)
        order = out_all["order"][0].tolist()
        bank_all, len_all = out_all["relation_bank"], out_all["relation_length"]
        bank_f, len_f = out_first["relation_bank"], out_first["relation_length"]

        def labels(bank, length, t):
            return bank[:int(length[t]), t].tolist()
        for i, s in enumerate(order):
            sp = nx.single_source_shortest_path(G, s)
            for j, t in enumerate(order):
                want = [[G[p[k]][p[k + 1]]["label"] for k in range(len(p) - 1)] for p in nx.all_shortest_paths(G, s, t)]
                ids = [int(x) for x in out_all["relation"][j + 1, i + 1, 0] if int(x) != 0]
                got = [labels(bank_all, len_all, x) for x in ids]
                if len(want[0]) == 0:
                    assert got == [[4]]
                else:
                    assert got == want, (trial, s, t)
                first = [G[sp[t][k]][sp[t][k + 1]]["label"] for k in range(len(sp[t]) - 1)] or [4]
                assert labels(bank_f, len_f, int(out_first["relation"][j + 1, i + 1, 0])) == first


def test_uniform_choice_is_uniform_over_alternatives(rb):
    # diamond s->{a,b,c}->t: three shortest paths 0->4; the choice must be ~uniform over seeds and always a shortest path
    edges = []
    for mid, lab in ((1, 6), (2, 7), (3, 8)):
        edges += [(0, mid, lab), (mid, 0, lab + 20), (mid, 4, lab + 3), (4, mid, lab + 23)]
    graph = (5, 0, np.array(edges, dtype=np.int32))
    counts = {}
    for seed in range(600):
        out = rb.build_relation_batch([graph], [0, 2, 3, 4, 5], path_mode=rb.PATH_UNIFORM, seed=seed)
        pos = out["order"][0].tolist().index(4)
        t = int(out["relation"][pos + 1, 1, 0])                  # path from the root (position 0) to node 4
        key = tuple(out["relation_bank"][:2, t].tolist())
        counts[key] = counts.get(key, 0) + 1
    assert set(counts) == {(6, 9), (7, 10), (8, 11)}
    assert min(counts.values()) > 150 and max(counts.values()) < 250


def test_synthetic_c2_batch_speed_and_invariants(rb):
    """64 AMR-shaped graphs of 100 nodes (the C2 batch): the native path must be >10x the Python generator's 2.3 s and agree
    with it on everything that does not depend on tie-breaking."""
    from gtos_amd import synth
    rng_graphs = []
    cfg = synth.CONFIGS["C2"]
    lab_cdf = synth._zipf_table(40)
    for gidx in range(cfg["B"]):
        r = synth.SplitMix64(2 * 10 ** 6 + gidx)
        adj = synth._amr_graph(r, cfg["N"], cfg["extra_frac"], 40, lab_cdf)
        edges = [(u, v, l) for u in range(len(adj)) for (v, l) in adj[u]]
        rng_graphs.append((cfg["N"], 0, np.array(edges, dtype=np.int32)))
    t0 = time.time()
    out = rb.build_relation_batch(rng_graphs, [synth.PAD, synth.REL_CLS, synth.REL_RCLS, synth.REL_SELF, synth.REL_TL],
                                  path_mode=rb.PATH_UNIFORM, seed=1)
    dt = time.time() - t0
    rel = out["relation"]
    n = cfg["N"] + 1
    assert rel.shape == (n, n, cfg["B"])
    assert int(rel.max()) == out["relation_bank"].shape[1] - 1
    assert bool((torch.diagonal(rel[1:, 1:, 0]) == 2).all()) and int(rel[0, 0, 0]) == 2
    assert bool((rel[1:, 0, :] == 0).all()) and bool((rel[0, 1:, :] == 1).all())
    assert int(out["relation_length"].max()) <= 8
    # path lengths are tie-break independent: compare the length multiset of graph 0 with the Python BFS
    order, depth = synth._bfs_order([[(int(v), int(l)) for (u2, v, l) in rng_graphs[0][2] if u2 == u] for u in range(cfg["N"])])
    assert out["order"][0].tolist() == order and out["depth"][0].tolist() == depth
    print("native relation batch for C2: %.3f s, R=%d" % (dt, out["relation_bank"].shape[1]))
    assert dt < 1.0


def _amr_vocabs(tmp_path, text):
    """Only the relation vocabulary matters for the relation tensors; the others get a minimal file."""
    from gtos_amd.vocab import Vocab, CLS, rCLS, SEL, TL, STR, END
    (tmp_path / "relation_vocab").write_text(text)
    (tmp_path / "mini").write_text("a\t1000\nb\t1000\n")
    m = str(tmp_path / "mini")
    return {'relation': Vocab(str(tmp_path / "relation_vocab"), 5, [CLS, rCLS, SEL, TL]), 'concept': Vocab(m, 5, [CLS]),
            'token': Vocab(m, 5, [STR, END]), 'predictable_token': Vocab(m, 5, [END]),
            'token_char': Vocab(m, 100, [STR, END]), 'concept_char': Vocab(m, 100, [STR, END])}


def _paths_of(rel, bank, length):
    """{(a, c, b): sorted list of label-path tuples} from relation [n,n,B,K] / [n,n,B] + bank."""
    rel = rel.numpy() if rel.dim() == 4 else rel.unsqueeze(-1).numpy()
    bank, length = bank.numpy(), length.numpy()
    out = {}
    n, _, B, K = rel.shape
    for a in range(n):
        for c in range(n):
            for b in range(B):
                ids = [int(t) for t in rel[a, c, b] if t != 0]
                out[(a, c, b)] = sorted(tuple(int(v) for v in bank[:length[t], t]) for t in ids)
    return out


def test_batchify_amr_from_preprocessed_items(tmp_path):
    """gtos_amd.data.batchify_amr rebuilds each graph from the length-1 paths of the preprocessed item and must give every
    pair the reference's SET of shortest label paths (eval: all of them, K=3 here; train: one of them)."""
    import json
    from conftest import GOLDEN
    from gtos_amd.data import batchify_amr
    g = np.load(os.path.join(GOLDEN, "host_amr_smatch.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "host_amr_smatch_items.json")))
    vocabs = _amr_vocabs(tmp_path, meta["relation_vocab"])
    items = [dict(it, token=["a"]) for it in meta["items"]]
    want = _paths_of(torch.from_numpy(g["relation"]), torch.from_numpy(g["relation_bank"]), torch.from_numpy(g["relation_length"]))
    ev = batchify_amr(items, vocabs, train=False)
    assert tuple(ev["relation"].shape) == tuple(g["relation"].shape)
    assert _paths_of(ev["relation"], ev["relation_bank"], ev["relation_length"]) == want
    assert torch.equal(ev["concept_depth"], torch.from_numpy(g["concept_depth"]))
    assert max(len(v) for v in want.values()) == 3                      # the re-entrant graphs have 3-way ties
    # the eval bank starts with <PAD>, <CLS>, <rCLS>, <SELF> (data.py:186-189), the train bank with <CLS>, <rCLS>, <SELF>
    rv = vocabs['relation']
    assert ev["relation_bank"][0, :4].tolist() == [rv.padding_idx, rv.token2idx('<CLS>'), rv.token2idx('<rCLS>'), rv.token2idx('<SELF>')]
    seen = {}
    for seed in range(12):
        tr = batchify_amr(items, vocabs, train=True, seed=seed)
        assert tr["relation"].dim() == 3
        assert tr["relation_bank"][0, :3].tolist() == [rv.token2idx('<CLS>'), rv.token2idx('<rCLS>'), rv.token2idx('<SELF>')]
        got = _paths_of(tr["relation"], tr["relation_bank"], tr["relation_length"])
        for key, alts in want.items():
            a, c, b = key
            if a == 0 or c == 0:
                continue                                                 # <CLS> row / column: eval and train banks differ by <PAD>
            if not alts:
                assert got[key] == []                                    # padding beyond this graph's nodes
                continue
            assert len(got[key]) == 1 and got[key][0] in alts, key
            seen.setdefault(key, set()).add(got[key][0])
    multi = [k for k, v in want.items() if len(v) > 1 and k[0] and k[1]]
    assert multi and all(len(seen[k]) > 1 for k in multi)                # over 12 seeds every tie is broken both ways


def test_amr_loader_batches_like_the_reference():
    """Batch composition and order of generator/data.py:DataLoader for four (batch_size, train, seed) settings over 1500
    synthetic items (the fixture carries their sizes, which is all the policy looks at; tests/golden/make_golden_loader_amr.py
    ran the reference's loader on them)."""
    import json
    import random
    from conftest import GOLDEN
    from gtos_amd.data import AMRLoader
    g = json.load(open(os.path.join(GOLDEN, "loader_amr_synth.json")))
    items = [{"concept": ["c"] * n, "token": ["t"] * m} for n, m in g["sizes"]]
    for run in g["runs"]:
        assert run["n_examples"] == len(items)
        random.seed(run["seed"])
        dl = AMRLoader(None, items, run["batch_size"], run["train"])
        assert dl.batch_indices() == run["batches"], (run["batch_size"], run["train"], run["seed"])
        dl2 = AMRLoader(None, items, run["batch_size"], run["train"], rng=random.Random(run["seed"]))
        assert dl2.batch_indices() == run["batches"]


def test_amr_loader_yields_training_batches_from_a_json_file(tmp_path):
    """File -> AMRLoader -> batchify_amr: every item lands in exactly one batch, the graph of an item is recovered once (the
    cached edges are reused by later epochs), training batches carry the trie and the relation index, record() adds the items."""
    import json
    import random
    from conftest import GOLDEN
    from gtos_amd import data
    meta = json.load(open(os.path.join(GOLDEN, "host_amr_smatch_items.json")))
    vocabs = _amr_vocabs(tmp_path, meta["relation_vocab"])
    items = [dict(it, token=["a"] * (1 + k % 3)) for k, it in enumerate(meta["items"] * 3)]
    path = os.path.join(str(tmp_path), "items.json")
    with open(path, "w", encoding="utf8") as fo:
        json.dump(items, fo)
    calls = []
    orig = data._edges_from_paths
    data._edges_from_paths = lambda item, rv: (calls.append(1), orig(item, rv))[1]
    try:
        dl = data.AMRLoader(vocabs, path, batch_size=60, for_train=True, rng=random.Random(5))
        assert len(calls) == len(items)                                  # recovered at load time ...
        dl.set_unk_rate(0.1)
        dl.record()
        seen = 0
        for epoch in range(2):
            for batch, its in dl:
                B = len(its)
                seen += B
                assert batch["relation"].dim() == 3 and batch["relation"].shape[2] == B
                assert batch["relation_trie"].R == batch["relation_bank"].shape[1]
                assert "relation_index" in batch
                assert batch["concept"].shape[1] == B and batch["token_in"].shape[1] == B
        assert seen == 2 * len(items)
        assert len(calls) == len(items)                                  # ... and never again
        ev = data.AMRLoader(vocabs, items, batch_size=10 ** 9, for_train=False)
        (b,) = list(ev)
        assert b["relation"].dim() == 4 and b["relation"].shape[2] == len(items)
    finally:
        data._edges_from_paths = orig


# ------------------------------------------------------------------------------------------------ loader overlap
def test_prefetcher_keeps_order_defers_assembly_and_propagates_errors():
    import threading
    import time
    from gtos_amd.data import Prefetcher
    built = []

    def make(i):
        def f():
            time.sleep(0.01 * ((7 - i) % 3))          # uneven assembly times: order must still be the source order
            built.append((i, threading.current_thread().name))
            return {"i": torch.tensor([i])}
        return f
    got = [int(b["i"]) for b in Prefetcher((make(i) for i in range(9)), depth=3, workers=3)]
    assert got == list(range(9))
    assert all(name != threading.current_thread().name for _, name in built)      # assembled off the consumer thread
    # stays at most `depth` ahead
    built.clear()
    pf = Prefetcher((make(i) for i in range(9)), depth=2, workers=2)
    first = next(pf)
    time.sleep(0.2)
    assert int(first["i"]) == 0 and len(built) <= 1 + 2 + 2
    assert [int(b["i"]) for b in pf] == list(range(1, 9))

    def boom():
        raise ValueError("bad batch")
    with pytest.raises(ValueError):
        list(Prefetcher([make(0), boom, make(2)], depth=2))


def test_prefetcher_hands_out_finished_batches_while_a_slow_source_assembles_the_next():
    """A source that assembles inside its own __next__ (iter(AMRLoader)) must not hold the hand-over lock: with batches ready, the
    consumer's next() returns at once although a worker sits inside the source (round-3 advisor finding: it waited a whole assembly)."""
    import time
    from gtos_amd.data import Prefetcher

    def slow_source():
        for i in range(8):
            time.sleep(0.25)
            yield {"i": torch.tensor([i])}
    pf = Prefetcher(slow_source(), depth=3, workers=1)
    time.sleep(0.9)                                   # three batches finished (depth), the worker is blocked on the depth bound
    waits = []
    got = []
    for _ in range(3):
        t0 = time.perf_counter()
        got.append(int(next(pf)["i"]))                # each take frees a slot: the worker re-enters the slow source right away
        waits.append(time.perf_counter() - t0)
    assert got == [0, 1, 2]
    assert max(waits) < 0.1, waits                    # ready batches are handed over without waiting for the source's 0.25 s
    assert [int(b["i"]) for b in pf] == [3, 4, 5, 6, 7]


def test_prefetcher_over_the_real_loader_matches_direct_iteration():
    from gtos_amd import synth
    from gtos_amd.data import Prefetcher
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index

    def job(k):
        return lambda: attach_relation_index(attach_path_trie(synth.make_batch(9, 3, 10, 5, first_graph=3 * k)[0]))
    direct = [job(k)() for k in range(4)]
    via = list(Prefetcher((job(k) for k in range(4)), depth=2, workers=2))
    for a, b in zip(direct, via):
        assert torch.equal(a["relation"], b["relation"]) and torch.equal(a["relation_trie"].row_sf, b["relation_trie"].row_sf)
        assert torch.equal(a["relation_index"].pair_sorted, b["relation_index"].pair_sorted)


def test_prefetcher_never_runs_more_than_depth_ahead_and_close_releases_the_workers():
    import threading
    import time
    from gtos_amd.data import Prefetcher
    started = []

    def make(i):
        def f():
            started.append(i)
            return {"i": torch.tensor([i])}
        return f
    pf = Prefetcher((make(i) for i in range(50)), depth=3, workers=6)
    time.sleep(0.3)
    assert len(started) == 3                      # reservation and take are one critical section: exactly `depth` ahead
    assert int(next(pf)["i"]) == 0
    time.sleep(0.2)
    assert len(started) == 4
    pf.close()
    for t in pf._threads:
        t.join(2.0)
    assert not any(t.is_alive() for t in pf._threads)
    with pytest.raises(StopIteration):
        next(pf)
    with Prefetcher((make(i) for i in range(5)), depth=2) as pf2:
        assert int(next(pf2)["i"]) == 0
    for t in pf2._threads:
        t.join(2.0)
    assert not any(t.is_alive() for t in pf2._threads)


def test_device_tensor_walk_reaches_the_index_objects():
    from gtos_amd import synth
    from gtos_amd.data import _device_tensors
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    b = attach_relation_index(attach_path_trie(synth.make_batch(9, 3, 10, 5)[0]))
    b["local_idx2token"] = [{5: "x"}]
    got = {id(t) for t in _device_tensors(b)}
    tr, ix = b["relation_trie"], b["relation_index"]
    for t in (b["relation"], tr.row_pf, tr.seq_order, tr.pf.rows, tr.sf.par_long, tr.sf.tok, ix.idx_q, ix.chunk_slot, ix.heavy_types):
        assert id(t) in got
    assert all(isinstance(t, torch.Tensor) for t in _device_tensors((b, ["a", "b"])))


def test_loaders_drop_an_empty_trailing_batch_keep_items_clean_and_offer_thunks(tmp_path):
    import json
    import random
    from conftest import GOLDEN
    from gtos_amd import data
    meta = json.load(open(os.path.join(GOLDEN, "host_amr_smatch_items.json")))
    vocabs = _amr_vocabs(tmp_path, meta["relation_vocab"])
    items = [dict(it, token=["a"] * (1 + k % 3)) for k, it in enumerate(meta["items"] * 2)]
    before = json.dumps(items, sort_keys=True)
    unit = data.AMRLoader.size_of(items[0])
    ev = data.AMRLoader(vocabs, items[:1], batch_size=unit, for_train=False)        # the only item closes a batch exactly
    assert ev.batch_indices() == [[0]]
    assert len(list(ev)) == 1
    dep = data.DependencyLoader(None, [(["a"], [0], ["x"], ["y"])], batch_size=2, for_train=False)
    assert dep.batch_indices() == [[0]]

    def loader():
        return data.AMRLoader(vocabs, items, batch_size=2 * unit, for_train=True, rng=random.Random(3))
    first = [f() for f in loader().thunks()]
    again = [f() for f in reversed(list(loader().thunks()))][::-1]                    # run order does not matter
    assert len(first) == len(again) >= 2
    for a, b in zip(first, again):
        assert torch.equal(a["relation"], b["relation"]) and torch.equal(a["concept"], b["concept"])
    via = list(data.Prefetcher(loader().thunks(), depth=2, workers=3))
    assert len(via) == len(first)
    for a, b in zip(first, via):
        assert torch.equal(a["relation"], b["relation"]) and torch.equal(a["token_in"], b["token_in"])
    assert json.dumps(items, sort_keys=True) == before                               # graphs cached beside the items, not on them
    plain = data.batchify_amr(items[:2], vocabs, index_prep=False)
    assert "relation_trie" not in plain and "relation_index" not in plain


def test_trie_builder_rejection_falls_back_to_the_per_row_encoder(monkeypatch):
    from gtos_amd import data
    from gtos_amd.pathtrie import build_path_trie
    with pytest.raises(ValueError):
        build_path_trie(torch.ones(3, 4, dtype=torch.int64), torch.tensor([1, 2, 2, 3]), chunk=65)
    with pytest.raises(ValueError):
        build_path_trie(torch.ones(65, 2, dtype=torch.int64), torch.tensor([65, 1]))      # a 65-label path (translator flavour)
    batch = {"relation": torch.zeros(2, 2, 1, dtype=torch.int64), "relation_bank": torch.ones(65, 2, dtype=torch.int64),
             "relation_length": torch.tensor([65, 1])}
    out = data._index_prep(dict(batch), True)
    assert "relation_trie" not in out and "relation_index" in out


def test_prefetcher_process_workers_give_the_same_batches_in_order(tmp_path):
    """processes=True: jobs -> forked worker processes (loader.run_job) -> shared memory -> consumer, same batches as the
    thunks of the same rng state, in order; an exception in a worker surfaces in the consumer; close() reaps the workers."""
    import json
    import random
    from conftest import GOLDEN
    from gtos_amd import data
    meta = json.load(open(os.path.join(GOLDEN, "host_amr_smatch_items.json")))
    vocabs = _amr_vocabs(tmp_path, meta["relation_vocab"])
    items = [dict(it, token=["a"] * (1 + k % 3)) for k, it in enumerate(meta["items"] * 3)]
    unit = data.AMRLoader.size_of(items[0])

    def loader():
        return data.AMRLoader(vocabs, items, batch_size=2 * unit, for_train=True, rng=random.Random(11))
    want = [f() for f in loader().thunks()]
    ld = loader()
    pf = data.Prefetcher(ld.jobs(), depth=3, workers=3, processes=True, runner=ld.run_job)
    got = list(pf)
    procs = list(pf._procs)
    pf.close()
    assert len(got) == len(want) >= 3
    for a, b in zip(want, got):
        assert torch.equal(a["relation"], b["relation"]) and torch.equal(a["token_in"], b["token_in"])
        assert torch.equal(a["relation_trie"].row_sf, b["relation_trie"].row_sf) and a["relation_trie"].batch_sizes == b["relation_trie"].batch_sizes
        assert torch.equal(a["relation_index"].pair_sorted, b["relation_index"].pair_sorted)
        assert a["local_idx2token"] == b["local_idx2token"]
    assert all(not p.is_alive() for p in procs)
    dl = data.DependencyLoader(None, [(["a"], [0], ["x"], ["y"])] * 4, batch_size=2, for_train=False)
    with pytest.raises(RuntimeError, match="loader worker failed"):          # vocabs=None: the worker raises, the consumer sees it
        list(data.Prefetcher(dl.jobs(), depth=2, workers=1, processes=True, runner=dl.run_job))


def test_prefetcher_builds_the_tries_itself_when_the_loader_leaves_them_out(tmp_path):
    """index_prep="device": the loader (threads or worker processes) ships the relation index only, Prefetcher(device_tries=True)
    adds the tries with the torch-op builder (on the consumer's device; here the CPU) -- the same tries the host builder gives."""
    import json
    import random
    from conftest import GOLDEN
    from gtos_amd import data
    from test_pathtrie import _same_object
    meta = json.load(open(os.path.join(GOLDEN, "host_amr_smatch_items.json")))
    vocabs = _amr_vocabs(tmp_path, meta["relation_vocab"])
    items = [dict(it, token=["a"] * (1 + k % 3)) for k, it in enumerate(meta["items"] * 3)]
    unit = data.AMRLoader.size_of(items[0])

    def loader(prep):
        return data.AMRLoader(vocabs, items, batch_size=2 * unit, for_train=True, rng=random.Random(11), index_prep=prep)
    want = [f() for f in loader(True).thunks()]
    bare = [f() for f in loader("device").thunks()]
    assert all("relation_trie" not in b and "relation_index" in b for b in bare)
    ld = loader("device")
    with data.Prefetcher(ld.jobs(), depth=3, workers=2, processes=True, runner=ld.run_job, device_tries=True) as pf:
        got_p = list(pf)
    ld = loader("device")
    with data.Prefetcher(ld.thunks(), depth=2, workers=2, device_tries=True) as pf:
        got_t = list(pf)
    assert len(got_p) == len(got_t) == len(want) >= 3
    for a, b, c in zip(want, got_p, got_t):
        assert torch.equal(a["relation"], b["relation"]) and torch.equal(a["relation"], c["relation"])
        assert _same_object(a["relation_trie"], b["relation_trie"]) == [] and _same_object(a["relation_trie"], c["relation_trie"]) == []
    # a batch that already has its tries is left alone
    ld = loader(True)
    with data.Prefetcher(ld.thunks(), depth=2, workers=1, device_tries=True) as pf:
        first = next(pf)
    assert _same_object(first["relation_trie"], want[0]["relation_trie"]) == []


def test_index_prep_auto_resolution(monkeypatch):
    """The loaders' default: "device_all" only where a GPU (and libgtos_hip.so) is visible, the host builders elsewhere; the
    environment override; an explicit argument wins."""
    from gtos_amd import data
    monkeypatch.delenv("GTOS_INDEX_PREP", raising=False)
    want = "device_all" if torch.cuda.is_available() else True
    assert data.resolve_index_prep("auto") == want
    for env, val in (("host", True), ("device", "device"), ("device_all", "device_all"), ("off", False)):
        monkeypatch.setenv("GTOS_INDEX_PREP", env)
        assert data.resolve_index_prep("auto") == val
        assert data.resolve_index_prep(True) is True and data.resolve_index_prep("device") == "device"
    monkeypatch.setenv("GTOS_INDEX_PREP", "host")
    dl = data.DependencyLoader(None, [(["a"], [0], ["x"], ["y"])] * 4, batch_size=2, for_train=False)
    assert dl.index_prep is True
