"""The staged GPU relation-index builder (gtos_amd/relindex_hip.py, csrc/relindex_kernels.h) on the CPU: its per-thread stage code run as
serial host loops (oracle/relindex_emul.cpp) through the product's Python glue == the host builder (csrc_host/relindex.cpp), array for
array."""
import ctypes

import pytest
import torch

from gtos_amd import synth
from gtos_amd.relindex import build_relation_index
from gtos_amd.relindex_hip import _geom, _table, build_relation_index_staged
from test_pathtrie import _same_object


class EmulBackend(object):
    def __init__(self, order=0):
        from oracle.build_emul import build
        self.lib = ctypes.CDLL(build("relindex"))
        P = ctypes.c_void_p
        self.lib.gtos_relindex_emul_phase_a.argtypes = [P, P]
        self.lib.gtos_relindex_emul_phase_b.argtypes = [P, ctypes.c_int64, P]
        self.lib.gtos_relindex_emul_set_order.argtypes = [ctypes.c_uint64]
        self.lib.gtos_relindex_emul_set_order.restype = None
        self.lib.gtos_relindex_emul_set_order(order)                   # 0: ascending thread order, 1: descending, > 1: a seeded permutation

    def phase_a(self, geom, bufs):
        assert self.lib.gtos_relindex_emul_phase_a(_geom(geom), _table(bufs)) == 0

    def phase_b(self, geom, nchunks, bufs):
        assert self.lib.gtos_relindex_emul_phase_b(_geom(geom), nchunks, _table(bufs)) == 0


def _random_relation(seed, n, B, R, heavy=0.3):
    """type ids with a few very frequent types (the <CLS> / <SELF> / <TL> pattern), many rare ones and some that never occur"""
    g = torch.Generator().manual_seed(seed)
    rel = torch.randint(0, R, (n, n, B), generator=g)
    hot = torch.rand((n, n, B), generator=g) < heavy
    rel[hot] = torch.randint(0, min(R, 4), (int(hot.sum()),), generator=g)
    return rel


@pytest.mark.parametrize("seed,n,B,R", [(1, 1, 1, 1), (2, 5, 3, 40), (3, 9, 8, 300), (4, 13, 16, 2000), (5, 21, 7, 50), (6, 30, 64, 20000)])
@pytest.mark.parametrize("chunk", [32, 4])
def test_staged_relation_index_stages_equal_the_host_builder(seed, n, B, R, chunk):
    rel = _random_relation(seed, n, B, R)
    host = build_relation_index(rel, R, chunk=chunk)
    staged = build_relation_index_staged(rel, R, EmulBackend(), chunk=chunk)
    assert _same_object(host, staged) == []


def test_staged_relation_index_on_amr_batches_and_limits(monkeypatch):
    for name, B in (("C1", 8), ("C2", 16)):
        batch, _ = synth.make_config_batch(name, rank=0, B=B)
        R = batch["relation_bank"].shape[1]
        host = build_relation_index(batch["relation"], R)
        assert _same_object(host, build_relation_index_staged(batch["relation"], R, EmulBackend())) == []
    with pytest.raises(ValueError):
        build_relation_index_staged(torch.full((2, 2, 2), 7), 5, EmulBackend())


def test_staged_builders_equal_the_host_builders_property_based():
    """hypothesis: arbitrary small connected graphs (self-loops, repeated edges, every node order), relation tensors and banks through the
    three staged builders' stage code (emulated) == the host builders."""
    import numpy as np
    from hypothesis import HealthCheck, given, settings, strategies as st
    from gtos_amd import relbatch
    from gtos_amd.pathtrie import build_path_trie
    from gtos_amd.pathtrie_hip import build_path_trie_staged
    from gtos_amd.relbatch_hip import build_relation_batch_staged
    from test_pathtrie import _EmulBackend as TrieEmul
    from test_relbatch_dev import EmulBackend as RelEmul, IDS, _same
    trie_emul, rel_emul, idx_emul = TrieEmul(), RelEmul(), EmulBackend()

    @st.composite
    def graph(draw):
        n = draw(st.integers(1, 7))
        parents = [draw(st.integers(0, v - 1)) for v in range(1, n)]
        edges = []
        for v, u in enumerate(parents, start=1):                       # a spanning tree keeps the graph connected
            l = draw(st.integers(6, 12))
            edges += [(v, u, l), (u, v, l + 20)]
        for _ in range(draw(st.integers(0, 6))):                       # anything on top: re-entrancies, repeats, self-loops
            u, v, l = draw(st.integers(0, n - 1)), draw(st.integers(0, n - 1)), draw(st.integers(6, 12))
            edges += [(u, v, l), (v, u, l + 20)]
        perm = draw(st.permutations(list(range(n))))
        return n, perm[0], np.array([(perm[a], perm[b], l) for a, b, l in edges], dtype=np.int32).reshape(-1, 3)

    @settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(st.lists(graph(), min_size=1, max_size=4), st.sampled_from([relbatch.PATH_FIRST, relbatch.PATH_UNIFORM]), st.integers(0, 2 ** 64 - 1),
           st.integers(1, 8), st.integers(1, 5))
    def check(graphs, mode, seed, max_len, chunk):
        host = relbatch.build_relation_batch(graphs, IDS, path_mode=mode, seed=seed, max_len=max_len, n_threads=1)
        staged = build_relation_batch_staged(graphs, IDS, rel_emul, path_mode=mode, seed=seed, max_len=max_len)
        assert _same(host, staged) == []
        R = host["relation_bank"].shape[1]
        assert _same_object(build_relation_index(host["relation"], R, chunk=chunk),
                            build_relation_index_staged(staged["relation"], R, idx_emul, chunk=chunk)) == []
        assert _same_object(build_path_trie(host["relation_bank"], host["relation_length"], chunk=chunk),
                            build_path_trie_staged(staged["relation_bank"], staged["relation_length"], trie_emul, chunk=chunk,
                                                   n_rows=staged["relation_rows"])) == []

    check()


@pytest.mark.parametrize("order", [1, 2, 77, 12345])
def test_staged_builders_do_not_depend_on_the_thread_order(order):
    """A GPU runs the threads of a stage in no particular order.  The emulation libraries visit them descending / in seeded random
    permutations (oracle/emul_order.h): every array of all three builders must come out the same as under the ascending loop, i.e. no
    stage has two threads writing different values to one location or reading what another thread of the same stage writes."""
    from gtos_amd import data, relbatch
    from gtos_amd.pathtrie import build_path_trie
    from gtos_amd.pathtrie_hip import build_path_trie_staged
    from gtos_amd.relbatch_hip import build_relation_batch_staged
    from test_pathtrie import _EmulBackend as TrieEmul
    from test_relbatch_dev import EmulBackend as RelEmul, _random_graphs, _same, IDS
    try:
        graphs = _random_graphs(11, 6, 15, 40, 0.4)
        host = relbatch.build_relation_batch(graphs, IDS, path_mode=relbatch.PATH_UNIFORM, seed=3, n_threads=1)
        staged = build_relation_batch_staged(graphs, IDS, RelEmul(order), path_mode=relbatch.PATH_UNIFORM, seed=3)
        assert _same(host, staged) == []
        R = host["relation_bank"].shape[1]
        for chunk in (32, 3):
            assert _same_object(build_relation_index(host["relation"], R, chunk=chunk),
                                build_relation_index_staged(host["relation"], R, EmulBackend(order), chunk=chunk)) == []
            assert _same_object(build_path_trie(host["relation_bank"], host["relation_length"], chunk=chunk),
                                build_path_trie_staged(host["relation_bank"], host["relation_length"], TrieEmul(order), chunk=chunk)) == []
        batch, _ = synth.make_config_batch("C2", rank=0, B=8)
        R = batch["relation_bank"].shape[1]
        assert _same_object(build_relation_index(batch["relation"], R), build_relation_index_staged(batch["relation"], R, EmulBackend(order))) == []
        assert _same_object(build_path_trie(batch["relation_bank"], batch["relation_length"]),
                            build_path_trie_staged(batch["relation_bank"], batch["relation_length"], TrieEmul(order))) == []
    finally:
        RelEmul(0), EmulBackend(0), TrieEmul(0)          # back to the ascending order for the tests that follow
