"""The staged GPU relation-batch builder (gtos_amd/relbatch_hip.py, csrc/relbatch_kernels.h) on the CPU: its per-thread stage code run
as serial host loops (oracle/relbatch_emul.cpp) through the product's Python glue == the host builder (csrc_host/relbatch.cpp, itself
bit-exact with the reference's batchify: tests/test_host_relbatch.py), array for array."""
import ctypes

import numpy as np
import pytest
import torch

from gtos_amd import relbatch, synth
from gtos_amd.relbatch_hip import PATH_FIRST, PATH_UNIFORM, _geom, _table, build_relation_batch_staged, graphs_csr

IDS = (0, 2, 3, 4, 5)           # pad, cls, rcls, self, tl


class EmulBackend(object):
    def __init__(self, order=0):
        from oracle.build_emul import build
        self.lib = ctypes.CDLL(build("relbatch"))
        P = ctypes.c_void_p
        self.lib.gtos_relbatch_emul_phase_a.argtypes = [P, P]
        self.lib.gtos_relbatch_emul_phase_b.argtypes = [P, ctypes.c_int64, P]
        self.lib.gtos_relbatch_emul_set_order.argtypes = [ctypes.c_uint64]
        self.lib.gtos_relbatch_emul_set_order.restype = None
        self.lib.gtos_relbatch_emul_set_order(order)                   # 0: ascending thread order, 1: descending, > 1: a seeded permutation

    def phase_a(self, geom, bufs, total):
        assert self.lib.gtos_relbatch_emul_phase_a(_geom(geom), _table(bufs)) == 0

    def phase_b(self, geom, R, bufs, total):
        assert self.lib.gtos_relbatch_emul_phase_b(_geom(geom), R, _table(bufs)) == 0

    def all_phase(self, which, geom, bufs, n, R=0):
        fn = getattr(self.lib, "gtos_relbatch_emul_all_" + which)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p] if which == "fill" else [ctypes.c_void_p, ctypes.c_void_p]
        assert (fn(_geom(geom), R, _table(bufs)) if which == "fill" else fn(_geom(geom), _table(bufs))) == 0


def _random_graphs(seed, B, nlo, nhi, extra, labels=40, tree_only=False):
    """connected labelled graphs in the builder's input form: a random rooted tree plus ``extra`` re-entrancy edges per node, every edge
    doubled with a reverse label (generator/AMRGraph.py:76-80); a repeated (src, dst) now and then (networkx overwrites the label)."""
    rng = np.random.RandomState(seed)
    graphs = []
    for _ in range(B):
        n = int(rng.randint(nlo, nhi + 1))
        edges = []
        for v in range(1, n):
            u = int(rng.randint(0, v))
            l = int(rng.randint(6, 6 + labels))
            edges += [(v, u, l), (u, v, l + labels)]
        if not tree_only:
            for _ in range(int(extra * n)):
                u, v = int(rng.randint(0, n)), int(rng.randint(0, n))
                if u != v:
                    l = int(rng.randint(6, 6 + labels))
                    edges += [(u, v, l), (v, u, l + labels)]
        perm = rng.permutation(n)                      # node ids are not in BFS order
        edges = [(int(perm[a]), int(perm[b]), l) for a, b, l in edges]
        graphs.append((n, int(perm[0]), np.array(edges, dtype=np.int32).reshape(-1, 3)))
    return graphs


def _same(a, b):
    bad = []
    for k in ("relation", "relation_bank", "relation_length", "order", "depth"):
        x, y = a[k], b[k]
        if not (x.dtype == y.dtype and x.shape == y.shape and torch.equal(x.cpu(), y.cpu())):
            bad.append((k, tuple(x.shape), tuple(y.shape)))
    if "relation_rows" in b and b["relation_rows"] != int(a["relation_length"].sum()):
        bad.append("relation_rows")
    return bad


@pytest.mark.parametrize("mode", [PATH_FIRST, PATH_UNIFORM])
@pytest.mark.parametrize("seed,B,nlo,nhi,extra", [(1, 1, 1, 1, 0.0), (2, 3, 2, 9, 0.3), (3, 5, 10, 30, 0.1), (4, 4, 20, 40, 1.0), (5, 8, 30, 30, 0.1)])
def test_staged_relation_batch_stages_equal_the_host_builder(mode, seed, B, nlo, nhi, extra):
    graphs = _random_graphs(seed, B, nlo, nhi, extra)
    host = relbatch.build_relation_batch(graphs, IDS, path_mode=mode, seed=1234 + seed, n_threads=1)
    staged = build_relation_batch_staged(graphs, IDS, EmulBackend(), path_mode=mode, seed=1234 + seed)
    assert _same(host, staged) == []


def test_staged_relation_batch_long_paths_high_seed_and_limits():
    graphs = _random_graphs(7, 3, 40, 60, 0.0, tree_only=True)               # deep trees: distances beyond 8 collapse to <TL>
    for max_len in (8, 3, 1):
        for seed in (0, (1 << 64) - 3):
            host = relbatch.build_relation_batch(graphs, IDS, path_mode=PATH_UNIFORM, seed=seed, max_len=max_len, n_threads=1)
            staged = build_relation_batch_staged(graphs, IDS, EmulBackend(), path_mode=PATH_UNIFORM, seed=seed, max_len=max_len)
            assert _same(host, staged) == []
    with pytest.raises(ValueError):
        build_relation_batch_staged(graphs, IDS, EmulBackend(), path_mode=relbatch.PATH_ALL)
    with pytest.raises(ValueError):                                          # a disconnected graph
        build_relation_batch_staged([(3, 0, np.array([[0, 1, 7], [1, 0, 8]], np.int32))], IDS, EmulBackend())
    with pytest.raises(ValueError):
        build_relation_batch_staged(graphs, (0, 2, 2, 4, 5), EmulBackend())


def test_staged_relation_batch_on_the_synthetic_amr_batches():
    """the loader's own graphs (synth items of a BASELINE config, through data._item_graph) in training mode"""
    from gtos_amd import data
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 8, first_graph=0, vocabs=vocabs)
    rv = vocabs['relation']
    assert [g[0] for g in graphs] == [len(x['concept']) for x in items]
    ids = data.relation_special_ids(rv)
    host = relbatch.build_relation_batch(graphs, ids, path_mode=PATH_UNIFORM, seed=99, n_threads=2)
    staged = build_relation_batch_staged(graphs, ids, EmulBackend(), path_mode=PATH_UNIFORM, seed=99)
    assert _same(host, staged) == []
    c = graphs_csr(graphs)
    assert c["S"] == sum(g[0] for g in graphs) and c["P"] == sum(g[0] ** 2 for g in graphs)


def test_device_all_loader_path_equals_the_host_loader(monkeypatch):
    """index_prep="device_all": the loader ships the flattened graphs only; attach_device_relations / attach_device_relation_index /
    attach_device_tries (here with the emulation backends on CPU tensors) rebuild relation, bank, length, index and tries equal to what
    the host loader ships for the same job."""
    import random
    from gtos_amd import data, pathtrie_hip, relbatch_hip, relindex_hip
    from test_pathtrie import _EmulBackend as TrieEmul, _same_object
    from test_relindex_dev import EmulBackend as IndexEmul
    monkeypatch.setattr(relbatch_hip.HipBackend, "shared", classmethod(lambda cls: EmulBackend()))
    monkeypatch.setattr(relindex_hip.HipBackend, "shared", classmethod(lambda cls: IndexEmul()))
    monkeypatch.setattr(pathtrie_hip.HipBackend, "shared", classmethod(lambda cls: TrieEmul()))
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 16, first_graph=0, vocabs=vocabs)
    unit = data.AMRLoader.size_of(items[0])

    def loader(prep):
        return data.AMRLoader(vocabs, items, batch_size=8 * unit - unit // 2, for_train=True, rng=random.Random(5), n_threads=1, graphs=graphs,
                              index_prep=prep)
    host_ld, dev_ld = loader(True), loader("device_all")
    for jh, jd in zip(list(host_ld.jobs())[:2], list(dev_ld.jobs())[:2]):
        assert jh == jd
        want, got = host_ld.run_job(jh), dev_ld.run_job(jd)
        assert 'relation' not in got and 'relation_index' not in got and 'relation_trie' not in got
        got = {k: (v.to("cpu") if hasattr(v, "to") else v) for k, v in got.items()}          # RelationGraphs passes through
        data.attach_device_relations(got, "cpu")
        data.attach_device_relation_index(got)
        data.attach_device_tries(got, "hip")
        assert 'relation_graphs' not in got
        for k in ("relation", "relation_bank", "relation_length", "concept", "token_in", "token_out", "cp_seq", "concept_depth"):
            assert torch.equal(want[k], got[k]), k
        assert _same_object(want["relation_index"], got["relation_index"]) == []
        assert _same_object(want["relation_trie"], got["relation_trie"]) == []
    # the same through the Prefetcher's consumer-side preparation (no device here: CPU tensors, emulation backends)
    host_ld, dev_ld = loader(True), loader("device_all")
    want = [host_ld.run_job(j) for j in host_ld.jobs()]
    with data.Prefetcher(dev_ld.thunks(), depth=2, workers=1, device_tries="hip") as pf:
        got = list(pf)
    dev_ld2 = loader("device_all")
    with data.Prefetcher(dev_ld2.jobs(), depth=2, workers=2, processes=True, runner=dev_ld2.run_job, device_tries="hip") as pf:
        got2 = list(pf)                                    # worker processes: the flattened graphs travel pickled beside the packed tensors
    dev_ld3, dev_ld4 = loader("device_all"), loader("device_all")
    with data.Prefetcher(dev_ld3.thunks(), depth=2, workers=2, device_tries="hip", prep_in_worker=True) as pf:
        got3 = list(pf)                                    # the preparation on the loader threads instead of the consumer
        assert pf.stats["batches"] == 2
    with data.Prefetcher(dev_ld4.jobs(), depth=2, workers=2, processes=True, runner=dev_ld4.run_job, device_tries="hip", prep_in_worker=True) as pf:
        got4 = list(pf)                                    # ... and on the receiver thread of the worker-process mode
    assert len(got) == len(got2) == len(got3) == len(got4) == len(want) == 2
    for w, g in list(zip(want, got)) + list(zip(want, got2)) + list(zip(want, got3)) + list(zip(want, got4)):
        assert 'relation_graphs' not in g and g['relation_rows'] == int(w['relation_length'].sum())
        assert torch.equal(w["relation"], g["relation"]) and torch.equal(w["relation_bank"], g["relation_bank"])
        assert _same_object(w["relation_index"], g["relation_index"]) == [] and _same_object(w["relation_trie"], g["relation_trie"]) == []


def test_device_all_dependency_flavour_reproduces_the_reference_batch(monkeypatch, tmp_path):
    """translator flavour (one shortest path per pair by first discovery): batchify_dependency(index_prep="device_all") + the device-side
    attachments (emulated) reproduce the batch the REFERENCE's batchify made from the same dev.txt trees (tests/golden/beam_dep_dev),
    and the staged builder alone reproduces the reference's relation tensors of the host_dep_dev fixture."""
    from conftest import load_golden
    from gtos_amd import data, pathtrie_hip, relbatch_hip, relindex_hip
    from test_beam_and_vocab import T, load_case, make_vocabs
    from test_pathtrie import _EmulBackend as TrieEmul, _same_object
    from test_relindex_dev import EmulBackend as IndexEmul
    monkeypatch.setattr(relbatch_hip.HipBackend, "shared", classmethod(lambda cls: EmulBackend()))
    monkeypatch.setattr(relindex_hip.HipBackend, "shared", classmethod(lambda cls: IndexEmul()))
    monkeypatch.setattr(pathtrie_hip.HipBackend, "shared", classmethod(lambda cls: TrieEmul()))
    meta, arrs = load_case("beam_dep_dev")
    vocabs = make_vocabs(meta, tmp_path)
    trees = [(d, h, t, g) for d, h, t, g in meta["trees"]]
    host = data.batchify_dependency(trees, vocabs, n_threads=1)
    got = data.batchify_dependency(trees, vocabs, n_threads=1, index_prep="device_all")
    assert 'relation' not in got
    data.attach_device_relations(got, "cpu")
    data.attach_device_relation_index(got)
    data.attach_device_tries(got, "hip")
    for k in ("concept", "concept_char", "concept_depth", "relation", "relation_bank", "relation_length", "cp_seq", "token_in", "token_char_in",
              "token_out"):
        assert torch.equal(got[k], T(arrs["batch/" + k])), k
    assert _same_object(host["relation_index"], got["relation_index"]) == []
    assert ("relation_trie" in host) == ("relation_trie" in got)
    if "relation_trie" in host:
        assert _same_object(host["relation_trie"], got["relation_trie"]) == []
    g = load_golden("host_dep_dev")
    for bi in (0, 1):
        p = "b%d/" % bi
        off = g[p + "off"]
        graphs = [relbatch.dependency_edges(g[p + "heads"][off[k]:off[k + 1]].tolist(), g[p + "dep_ids"][off[k]:off[k + 1]].tolist(),
                                            g[p + "rev_ids"][off[k]:off[k + 1]].tolist()) for k in range(len(off) - 1)]
        for mode in (PATH_FIRST, PATH_UNIFORM):                                     # shortest paths in a tree are unique
            out = build_relation_batch_staged(graphs, g["special_ids"].tolist(), EmulBackend(), path_mode=mode, seed=7)
            for k in ("relation", "relation_bank", "relation_length"):
                assert torch.equal(out[k], torch.from_numpy(g[p + k])), (bi, mode, k)


@pytest.mark.parametrize("order", [0, 1, 4242])
@pytest.mark.parametrize("seed,B,nlo,nhi,extra", [(1, 1, 1, 1, 0.0), (2, 3, 2, 9, 0.3), (3, 5, 10, 30, 0.1), (4, 4, 12, 24, 1.0), (5, 6, 20, 20, 0.3)])
def test_staged_relation_batch_every_shortest_path_mode_equals_the_host_builder(order, seed, B, nlo, nhi, extra):
    """GTOS_PATH_ALL (the eval-mode batches: relation [n,n,B,K], every alternative in networkx's enumeration order, <PAD> = type 0)"""
    from gtos_amd.relbatch_hip import build_relation_batch_all_staged
    graphs = _random_graphs(seed, B, nlo, nhi, extra, labels=3)            # few labels: many alternative shortest paths
    try:
        for max_len in (8, 2):
            host = relbatch.build_relation_batch(graphs, IDS, path_mode=relbatch.PATH_ALL, max_len=max_len, n_threads=1)
            staged = build_relation_batch_all_staged(graphs, IDS, EmulBackend(order), max_len=max_len)
            assert host["relation"].dim() == 4 and _same(host, staged) == []
    finally:
        EmulBackend(0)
    with pytest.raises(ValueError):
        build_relation_batch_all_staged(graphs, (2, 2, 3, 4, 5), EmulBackend())


def test_staged_every_shortest_path_mode_reproduces_the_reference_eval_batch():
    """the reference's own eval-mode batchify on real AMRs (tests/golden/host_amr_smatch, made by generator/data.py:178-232 under
    networkx): relation [n,n,B,K], bank and lengths from the staged builder's stage code, bit for bit"""
    from conftest import load_golden
    from gtos_amd.relbatch_hip import build_relation_batch_all_staged
    g = load_golden("host_amr_smatch")
    eo = g["edge_off"]
    graphs = [(int(g["n_nodes"][k]), int(g["roots"][k]), g["edges"][eo[k]:eo[k + 1]]) for k in range(len(g["n_nodes"]))]
    out = build_relation_batch_all_staged(graphs, g["special_ids"].tolist(), EmulBackend())
    assert torch.equal(out["relation"], torch.from_numpy(g["relation"]))
    assert torch.equal(out["relation_bank"], torch.from_numpy(g["relation_bank"]))
    assert torch.equal(out["relation_length"], torch.from_numpy(g["relation_length"]))
    assert torch.equal(out["depth"].long(), torch.from_numpy(g["concept_depth"])[1:].t())


def test_device_all_eval_batch_equals_the_host_eval_batch(monkeypatch):
    """batchify_amr(train=False, index_prep="device_all"): the eval batch ([n,n,B,K], every alternative) rebuilt by the device-side
    attachments (emulated) == the host eval batch"""
    from gtos_amd import data, pathtrie_hip, relbatch_hip
    from test_pathtrie import _EmulBackend as TrieEmul, _same_object
    monkeypatch.setattr(relbatch_hip.HipBackend, "shared", classmethod(lambda cls: EmulBackend()))
    monkeypatch.setattr(pathtrie_hip.HipBackend, "shared", classmethod(lambda cls: TrieEmul()))
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 6, first_graph=0, vocabs=vocabs)
    cache = {id(d): (vocabs['relation'], d, g) for d, g in zip(items, graphs)}
    want = data.batchify_amr(items, vocabs, train=False, n_threads=1, graph_cache=cache)
    got = data.batchify_amr(items, vocabs, train=False, n_threads=1, graph_cache=cache, index_prep="device_all")
    assert 'relation' not in got and want['relation'].dim() == 4
    data.attach_device_relations(got, "cpu")
    data.attach_device_relation_index(got)                 # (no index for eval batches)
    data.attach_device_tries(got, "hip")
    assert 'relation_index' not in got
    for k in ("relation", "relation_bank", "relation_length", "concept", "token_in"):
        assert torch.equal(want[k], got[k]), k
    assert _same_object(want["relation_trie"], got["relation_trie"]) == []


def test_flattened_graphs_refuse_invalid_node_counts_before_writing():
    """gtos_relbatch_csr sizes nothing from unvalidated counts (found by the sanitizer fuzz of the host ABI: a batch whose second graph has
    a non-positive node count used to overrun the outputs sized from the sum)"""
    good = (3, 0, np.array([[0, 1, 7], [1, 0, 8], [1, 2, 7], [2, 1, 8]], np.int32))
    for bad_n in (0, -1):
        with pytest.raises(ValueError):
            graphs_csr([good, (bad_n, 0, np.zeros((0, 3), np.int32))])
    with pytest.raises(ValueError):
        graphs_csr([])
    c = graphs_csr([good, (1, 0, np.zeros((0, 3), np.int32))])          # a single node: no adjacency at all
    assert c["S"] == 4 and c["P"] == 10 and c["adj_base"].tolist() == [0, 4, 4]


def test_device_all_batch_completed_on_the_host_when_the_consumer_is_not_a_gpu():
    """the loaders' index_prep="auto" resolves to "device_all" wherever a GPU is visible, so a CPU / fp32 parity model fed
    from such a loader meets a batch that carries ``relation_graphs`` instead of relation / bank / length.  complete_on_device() on a
    non-GPU target must not hand host pointers to the HIP stage kernels: it falls back to the C++ host builder (the real HipBackend says
    ``needs_device``; no emulation here) and yields the reference-shaped batch the host loader ships."""
    import random
    from gtos_amd import data, relbatch_hip
    assert relbatch_hip.HipBackend.needs_device
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 16, first_graph=0, vocabs=vocabs)
    unit = data.AMRLoader.size_of(items[0])

    def loader(prep):
        return data.AMRLoader(vocabs, items, batch_size=8 * unit - unit // 2, for_train=True, rng=random.Random(5), n_threads=1, graphs=graphs,
                              index_prep=prep)
    host_ld, dev_ld = loader(True), loader("device_all")
    for jh, jd in zip(list(host_ld.jobs())[:2], list(dev_ld.jobs())[:2]):
        want, got = host_ld.run_job(jh), dev_ld.run_job(jd)
        assert 'relation_graphs' in got and 'relation' not in got
        got = data.complete_on_device(got, torch.device("cpu"))
        assert 'relation_graphs' not in got
        for k in ("relation", "relation_bank", "relation_length", "concept", "token_in"):
            assert torch.equal(want[k], got[k]) and not got[k].is_cuda, k
        assert int(got['relation_rows']) == int(want['relation_length'].sum())
