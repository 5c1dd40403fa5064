"""GPU: the staged HIP builders of a batch's relation tensors (gtos_amd.relbatch_hip -> gtos_relbatch_dev_phase_a / _b) and of its relation
index (gtos_amd.relindex_hip -> gtos_relindex_dev_phase_a / _b) against the host builders (csrc_host/relbatch.cpp, relindex.cpp), array
for array, and the loader mode that leaves all of it to the device (index_prep="device_all" through Prefetcher).  The stage code itself
is proven equal on the CPU (tests/test_relbatch_dev.py, tests/test_relindex_dev.py through the emulation libraries); this file covers
what only the GPU can: the launch glue, rocPRIM, double arithmetic on the device taking the host's branches, and the one-wave greedy
placement.

Round 4: first executed on an MI355X at the end of round 3 (all cases equal), collected as ordinary ``-m gpu`` tests since -- the
child-process / xfail indirection of round 3 is gone, a regression in a device builder turns the GPU suite red."""
import random

import pytest
import torch

from gtos_amd import relbatch, synth
from test_pathtrie import _same_object
from test_relbatch_dev import IDS, _random_graphs, _same
from test_relindex_dev import _random_relation

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("mode", [relbatch.PATH_FIRST, relbatch.PATH_UNIFORM])
@pytest.mark.parametrize("seed,B,nlo,nhi,extra", [(1, 1, 1, 1, 0.0), (2, 3, 2, 9, 0.3), (3, 5, 10, 30, 0.1), (4, 4, 20, 40, 1.0), (5, 8, 30, 30, 0.1)])
def test_hip_relation_batch_equals_the_host_builder(mode, seed, B, nlo, nhi, extra):
    from gtos_amd.relbatch_hip import HipBackend, build_relation_batch_staged
    graphs = _random_graphs(seed, B, nlo, nhi, extra)
    host = relbatch.build_relation_batch(graphs, IDS, path_mode=mode, seed=1234 + seed, n_threads=1)
    hip = build_relation_batch_staged(graphs, IDS, HipBackend.shared(), path_mode=mode, seed=1234 + seed, device=dev())
    assert hip["relation"].is_cuda and _same(host, hip) == []


def test_hip_relation_batch_at_c2_size_long_paths_and_seeds():
    from gtos_amd import data
    from gtos_amd.relbatch_hip import HipBackend, build_relation_batch_staged
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C2", 64, first_graph=0, vocabs=vocabs)
    ids = data.relation_special_ids(vocabs['relation'])
    for seed in (99, (1 << 64) - 3):
        host = relbatch.build_relation_batch(graphs, ids, path_mode=relbatch.PATH_UNIFORM, seed=seed, n_threads=2)
        assert _same(host, build_relation_batch_staged(graphs, ids, HipBackend.shared(), path_mode=relbatch.PATH_UNIFORM, seed=seed, device=dev())) == []
    deep = _random_graphs(7, 3, 40, 60, 0.0, tree_only=True)              # distances beyond max_len collapse to <TL>
    for max_len in (8, 3, 1):
        host = relbatch.build_relation_batch(deep, IDS, path_mode=relbatch.PATH_UNIFORM, seed=5, max_len=max_len, n_threads=1)
        assert _same(host, build_relation_batch_staged(deep, IDS, HipBackend.shared(), path_mode=relbatch.PATH_UNIFORM, seed=5, max_len=max_len,
                                                       device=dev())) == []


@pytest.mark.parametrize("seed,B,nlo,nhi,extra", [(1, 1, 1, 1, 0.0), (2, 3, 2, 9, 0.3), (3, 5, 10, 30, 0.1), (4, 4, 12, 24, 1.0), (5, 6, 20, 20, 0.3)])
def test_hip_relation_batch_every_shortest_path_mode_equals_the_host_builder(seed, B, nlo, nhi, extra):
    """GTOS_PATH_ALL (eval-mode batches, relation [n,n,B,K]) incl. the reference's own eval batch of real AMRs"""
    from conftest import load_golden
    from gtos_amd.relbatch_hip import HipBackend, build_relation_batch_all_staged
    graphs = _random_graphs(seed, B, nlo, nhi, extra, labels=3)
    for max_len in (8, 2):
        host = relbatch.build_relation_batch(graphs, IDS, path_mode=relbatch.PATH_ALL, max_len=max_len, n_threads=1)
        assert _same(host, build_relation_batch_all_staged(graphs, IDS, HipBackend.shared(), max_len=max_len, device=dev())) == []
    if seed == 1:
        g = load_golden("host_amr_smatch")
        eo = g["edge_off"]
        graphs = [(int(g["n_nodes"][k]), int(g["roots"][k]), g["edges"][eo[k]:eo[k + 1]]) for k in range(len(g["n_nodes"]))]
        out = build_relation_batch_all_staged(graphs, g["special_ids"].tolist(), HipBackend.shared(), device=dev())
        for k in ("relation", "relation_bank", "relation_length"):
            assert torch.equal(out[k].cpu(), torch.from_numpy(g[k])), k


@pytest.mark.parametrize("seed,n,B,R", [(1, 1, 1, 1), (2, 5, 3, 40), (3, 9, 8, 300), (4, 13, 16, 2000), (5, 21, 7, 50), (6, 30, 64, 20000)])
@pytest.mark.parametrize("chunk", [32, 4])
def test_hip_relation_index_equals_the_host_builder(seed, n, B, R, chunk):
    from gtos_amd.relindex import build_relation_index
    from gtos_amd.relindex_hip import HipBackend, build_relation_index_staged
    rel = _random_relation(seed, n, B, R)
    host = build_relation_index(rel, R, chunk=chunk)
    assert _same_object(host, build_relation_index_staged(rel.to(dev()), R, HipBackend.shared(), chunk=chunk).cpu()) == []


def test_hip_relation_index_at_c2_size_and_limits():
    from gtos_amd.relindex import build_relation_index
    from gtos_amd.relindex_hip import HipBackend, build_relation_index_staged
    batch, _ = synth.make_config_batch("C2", rank=0, B=64)
    R = batch["relation_bank"].shape[1]
    host = build_relation_index(batch["relation"], R)
    assert _same_object(host, build_relation_index_staged(batch["relation"].to(dev()), R, HipBackend.shared()).cpu()) == []
    with pytest.raises(ValueError):
        build_relation_index_staged(torch.full((2, 2, 2), 7, device=dev()), 5, HipBackend.shared())


def test_device_all_loader_through_the_prefetcher_equals_the_host_loader():
    """AMRLoader(index_prep="device_all") -> Prefetcher(device=cuda): the batch the consumer gets holds relation / bank / length / index /
    tries built on the GPU, equal to what the host loader ships for the same jobs."""
    from gtos_amd import data
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 32, first_graph=0, vocabs=vocabs)
    unit = data.AMRLoader.size_of(items[0])

    def loader(prep):
        return data.AMRLoader(vocabs, items, batch_size=8 * unit - unit // 2, for_train=True, rng=random.Random(5), n_threads=1, graphs=graphs,
                              index_prep=prep)
    host_ld, dev_ld = loader(True), loader("device_all")
    want = [host_ld.run_job(j) for j in host_ld.jobs()]
    with data.Prefetcher(dev_ld.thunks(), depth=2, workers=1, device=dev(), device_tries="hip") as pf:
        got = list(pf)
    torch.cuda.synchronize()
    assert len(got) == len(want) > 1
    for w, g in zip(want, got):
        assert 'relation_graphs' not in g and g["relation"].is_cuda
        for k in ("relation", "relation_bank", "relation_length", "concept", "token_in", "token_out", "cp_seq"):
            assert torch.equal(w[k], g[k].cpu()), k
        assert _same_object(w["relation_index"], g["relation_index"].cpu()) == []
        assert _same_object(w["relation_trie"], g["relation_trie"].to(torch.device("cpu"))) == []


def test_loader_default_leaves_the_relation_section_to_the_device_and_the_model_completes_the_batch(monkeypatch):
    """Round 4: ``AMRLoader(...)`` / ``DependencyLoader(...)`` without an ``index_prep`` argument resolve to "device_all" on a GPU box
    (the suite pins GTOS_INDEX_PREP=host for the host-side tests; cleared here).  A batch taken straight from the loader -- no
    Prefetcher -- carries ``relation_graphs``; ``Generator.forward`` builds relation / bank / index / tries on its device and gives the
    loss of the host-prepared batch (same graphs, same path draws)."""
    from gtos_amd import data
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    monkeypatch.delenv("GTOS_INDEX_PREP", raising=False)
    assert data.resolve_index_prep("auto") == "device_all"
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C1", 16, first_graph=0, vocabs=vocabs)
    unit = data.AMRLoader.size_of(items[0])
    dev_ld = data.AMRLoader(vocabs, items, batch_size=8 * unit - unit // 2, for_train=True, rng=random.Random(5), n_threads=1, graphs=graphs)
    host_ld = data.AMRLoader(vocabs, items, batch_size=8 * unit - unit // 2, for_train=True, rng=random.Random(5), n_threads=1, graphs=graphs,
                             index_prep=True)
    assert dev_ld.index_prep == "device_all" and host_ld.index_prep is True
    m = build_generator(Generator, "C1", dev(), dropout=0.0).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    for bd, bh in zip(dev_ld, host_ld):
        assert 'relation_graphs' in bd and 'relation' not in bd
        bd = {k: (v.to(dev()) if hasattr(v, "to") else v) for k, v in bd.items()}
        loss_d = m(bd)
        assert 'relation_graphs' not in bd and bd['relation'].is_cuda and 'relation_trie' in bd and 'relation_index' in bd
        assert torch.equal(bd['relation'].cpu(), bh['relation']) and torch.equal(bd['relation_bank'].cpu(), bh['relation_bank'])
        loss_h = m({k: (v.to(dev()) if hasattr(v, "to") else v) for k, v in bh.items()})
        assert abs(float(loss_d) - float(loss_h)) <= 1e-5 * max(1.0, abs(float(loss_h))), (float(loss_d), float(loss_h))
