"""vocabulary files, checkpoint format, beam search.

Fixtures: tests/golden/beam_smatch.{npz,json}, produced by the reference itself (tests/golden/make_golden_beam.py):
its Vocab on six small vocabulary files (thresholds, a malformed line), and Generator.work on the six AMRs of
generator/smatch/test_input*.txt with three (beam size, max steps, min steps) settings."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR, T

SCORE_TOL = 2e-4          # accumulated fp32 log-likelihoods, CPU oracle / GPU kernels vs the reference's CPU run


CASES = ["beam_smatch", "beam_dep_dev"]     # generator flavour (AMR, K-path eval batch) / translator flavour (real dev.txt)


def load_case(name="beam_smatch"):
    meta = json.load(open(os.path.join(GOLDEN_DIR, name + ".json"), encoding="utf8"))
    arrs = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return meta, arrs


def write_vocab_files(meta, tmp_path):
    for name, text in meta["files"].items():
        (tmp_path / name).write_text(text)
    return str(tmp_path)


def make_vocabs(meta, tmp_path):
    from gtos_amd.vocab import load_vocabs
    return load_vocabs(write_vocab_files(meta, tmp_path))


def batch_of(meta, arrs, device=None):
    b = {k[len("batch/"):]: T(arrs[k]) for k in arrs.files if k.startswith("batch/")}
    if device is not None:
        b = {k: v.to(device) for k, v in b.items()}
    b["local_idx2token"] = [{int(k): v for k, v in d.items()} for d in meta["local_idx2token"]]
    return b


def state_dict_of(arrs):
    return {k[len("sd/"):]: T(arrs[k]) for k in arrs.files if k.startswith("sd/")}


def check_hyps(got, want, what):
    assert [list(s) for s, _ in got] == [list(s) for s, _ in want], what
    for (_, a), (_, b) in zip(got, want):
        if b == float("-inf") or a == float("-inf"):
            assert a == b, what
        else:
            assert abs(a - b) < SCORE_TOL, (what, a, b)


# ------------------------------------------------------------------------------------------------ vocabulary files
@pytest.mark.parametrize("case", CASES)
def test_vocab_matches_reference(case, tmp_path):
    meta, _ = load_case(case)
    vocabs = make_vocabs(meta, tmp_path)
    for name, truth in meta["vocab_truth"].items():
        v = vocabs[name]
        assert v.size == truth["size"], name
        assert abs(v.coverage - truth["coverage"]) < 1e-12
        assert v.idx2token(list(range(v.size))) == truth["idx2token"]
        for tok, idx in truth.get("token2idx", {}).items():
            assert v.token2idx(tok) == idx, (name, tok)
        for tok, pr in truth.get("priority", {}).items():
            assert v.priority(tok) == pr
        assert v.padding_idx == 0 and v.unk_idx == 1
    if case == "beam_smatch":
        # the malformed (blank) line re-enters the previous token, as the reference does: sizes decide embedding shapes
        assert meta["vocab_truth"]["token"]["idx2token"].count("go") == 2


def test_tensorisers_shapes_and_padding(tmp_path):
    from gtos_amd.vocab import lists_to_tensor, strings_to_char_tensor, copy_vocab, STR, END
    meta, _ = load_case()
    vocabs = make_vocabs(meta, tmp_path)
    tv, cv = vocabs["token"], vocabs["token_char"]
    x = lists_to_tensor([["the", "boy"], ["girl"]], tv)
    assert x.shape == (2, 2) and x[1, 1].item() == tv.padding_idx and x[0, 0].item() == tv.token2idx("the")
    c = strings_to_char_tensor([["the", "boy"], ["girl"]], cv)
    assert c.shape == (2, 2, 22)
    assert c[0, 0, 0].item() == cv.token2idx(STR) and c[0, 0, 4].item() == cv.token2idx(END)
    assert c[1, 1].tolist() == strings_to_char_tensor([["<PAD>"]], cv)[0, 0].tolist()       # padded slot = the string <PAD>
    pv = vocabs["predictable_token"]
    cp, t2i, i2t = copy_vocab(["zzz-01", "the", "zzz-01", "qqq"], pv)
    assert cp == ["zzz-01", "the", "zzz-01", "qqq"] and t2i == {"zzz-01": pv.size, "qqq": pv.size + 1}
    assert i2t == {pv.size: "zzz-01", pv.size + 1: "qqq"}
    loc = lists_to_tensor([["zzz-01", "the"]], pv, [t2i])
    assert loc[:, 0].tolist() == [pv.size, pv.token2idx("the")]


# ------------------------------------------------------------------------------------------------ checkpoints
def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    from gtos_amd.checkpoint import save_checkpoint, load_checkpoint, build_from_checkpoint
    meta, arrs = load_case()
    vdir = write_vocab_files(meta, tmp_path)
    cfg = meta["cfg"]
    ga = cfg["gen_args"]
    args = argparse.Namespace(
        token_char_dim=ga[0], token_dim=ga[1], concept_char_dim=ga[2], concept_dim=ga[3],
        cnn_filters=[tuple(f) for f in ga[4]], char2word_dim=ga[5], char2concept_dim=ga[6], rel_dim=ga[7],
        rnn_hidden_size=ga[8], rnn_num_layers=ga[9], embed_dim=cfg["d"], ff_embed_dim=cfg["ff"], num_heads=cfg["H"],
        dropout=0.2, snt_layers=cfg["snt_layers"], graph_layers=cfg["graph_layers"],
        inference_layers=cfg["inference_layers"], pretrained_file=None,
        **{n: os.path.join(vdir, n) for n in meta["files"]})
    sd = state_dict_of(arrs)
    # exactly what generator/train.py:164 writes
    path = str(tmp_path / "epoch1_batch2")
    torch.save({"args": args, "model": sd}, path)
    model, args2, vocabs = build_from_checkpoint(path, torch.device("cpu"))
    assert vars(args2) == vars(args)
    got = model.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert not model.training
    # and back: a checkpoint written by this stack has the reference's layout
    path2 = str(tmp_path / "resaved")
    save_checkpoint(path2, args, model)
    raw = torch.load(path2, map_location="cpu", weights_only=False)
    assert set(raw) == {"args", "model"} and isinstance(raw["args"], argparse.Namespace)
    a3, sd3 = load_checkpoint(path2)
    assert all(torch.equal(sd3[k], sd[k]) for k in sd)


# ------------------------------------------------------------------------------------------------ beam search
def test_beam_bookkeeping_rules():
    """Beam.advance on hand-made steps: <UNK> -> -inf, budget shrinks with finished hypotheses, min_time_step drops
    early <END>, stable tie order."""
    from gtos_amd.search import Beam
    from gtos_amd.vocab import END, UNK
    b = Beam(3, 2, 10)
    parents = b.advance([[("a", -1.0), (END, -0.5), (UNK, -0.1)]])
    # <END> after 0 tokens is dropped (min_time_step 2), <UNK> kept alive with -inf (the reference keeps it too)
    assert parents == [0, 0] and [h.seq[-1] for h in b.hypotheses] == ["a", UNK] and b.completed_hypotheses == []
    assert b.hypotheses[1].score == float("-inf")
    parents = b.advance([[("b", -1.0), ("c", -1.0), (END, -3.0)], [("x", -0.1), ("y", -0.2), ("z", -0.3)]])
    assert [h.seq[-1] for h in b.hypotheses] == ["b", "c"] and parents == [0, 0]      # tie keeps rank order; 3rd is <END>
    assert b.completed_hypotheses == []                                                  # 1 token < min_time_step 2
    parents = b.advance([[(END, -0.1), ("d", -5.0)], [("e", -0.2), (END, -9.0)]])
    assert len(b.completed_hypotheses) == 1 and [h.seq[-1] for h in b.hypotheses] == ["e", "d"] and parents == [1, 0]
    b.advance([[("f", -1.0), ("g", -2.0), ("h", -3.0)], [("i", -1.0), ("j", -2.0), ("k", -3.0)]])
    assert len(b.hypotheses) == 2                                                        # budget = 3 - 1 finished
    assert not b.completed()
    best = b.get_k_best(1, 0.6)[0]
    assert best.seq[-1] == END


@pytest.mark.parametrize("case", CASES)
def test_oracle_beam_search_matches_reference(case, tmp_path):
    from oracle import gtos_oracle as O
    meta, arrs = load_case(case)
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    model = O.Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"],
                        cfg["inference_layers"], depth_size=cfg.get("depth_size", 32))
    model.load_state_dict(state_dict_of(arrs))
    model.eval()
    batch = batch_of(meta, arrs)
    for run in meta["runs"]:
        out = O.generator_work(model, batch, vocabs, run["beam"], run["max_step"], run["min_step"])
        for b, ((fin, alive), want) in enumerate(zip(out, run["expect"])):
            tag = "run %s sentence %d" % ((run["beam"], run["max_step"], run["min_step"]), b)
            check_hyps(fin, want["finished"], tag + " finished")
            if len(fin) < run["beam"]:                 # the reference stops updating a full beam; its leftovers differ
                check_hyps(alive, want["alive"], tag + " alive")
            check_hyps(O.k_best(fin, alive, run["beam"], cfg["alpha"]), want["k_best"], tag + " k-best")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_beam_search_matches_reference(case, tmp_path):
    from gtos_amd.generator import Generator
    meta, arrs = load_case(case)
    dev = torch.device("cuda:0")
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    model = Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"],
                      cfg["inference_layers"], None, dev, depth_size=cfg.get("depth_size", 32)).to(dev)
    model.load_state_dict(state_dict_of(arrs))
    model.eval()
    batch = batch_of(meta, arrs, dev)
    for run in meta["runs"]:
        beams = model.work(batch, run["beam"], run["max_step"], run["min_step"])
        for b, (beam, want) in enumerate(zip(beams, run["expect"])):
            tag = "run %s sentence %d" % ((run["beam"], run["max_step"], run["min_step"]), b)
            fin = [(h.seq, h.score) for h in beam.completed_hypotheses]
            alive = [(h.seq, h.score) for h in beam.hypotheses]
            assert beam.steps == want["steps"], tag
            check_hyps(fin, want["finished"], tag + " finished")
            check_hyps(alive, want["alive"], tag + " alive")
            best = [(h.seq, h.score) for h in beam.get_k_best(run["beam"], cfg["alpha"])]
            check_hyps(best, want["k_best"], tag + " k-best")


def _search_by_batch_reference_style(model, beams, mem_dict):
    """The data flow of the reference's search loop (generator/search.py:113-166) around ``model.decode_step(inp, state_dict,
    mem_dict, offset, topk)``: every hypothesis carries its own state dict (slices along dim 1), the live ones are joined with
    torch.cat before a step, the memory is gathered per hypothesis (tensors: index_select along dim 1; lists: indexed), and the
    returned state is index_select-ed by the surviving parents and split again -- all generic over the dictionary keys."""
    dev = mem_dict['probe'].device
    states = [[{}] for _ in beams]                     # per beam, per live hypothesis
    while True:
        owners, hyp_states, last, offset = [], [], [], -1
        for bi, beam in enumerate(beams):
            if not beam.completed():
                for h, st in zip(beam.hypotheses, states[bi]):
                    owners.append(bi)
                    hyp_states.append(st)
                    last.append(h.seq[-1:])
                    offset = len(h.seq) - 1
        if not owners:
            break
        inp = model.prepare_incremental_input(last)
        joined = {k: torch.cat([st[k] for st in hyp_states], 1 if hyp_states[0][k].dim() >= 3 else 0) for k in hyp_states[0]}
        idx = torch.tensor(owners, device=dev)
        cur_mem = {k: ([v[i] for i in owners] if isinstance(v, list) else v.index_select(1, idx)) for k, v in mem_dict.items()}
        new_state, results = model.decode_step(inp, joined, cur_mem, offset, beams[0].beam_size)
        pos = 0
        for bi, beam in enumerate(beams):
            if beam.completed():
                continue
            n = len(beam.hypotheses)
            parents = beam.advance(results[pos:pos + n])
            par = torch.tensor([pos + p for p in parents], dtype=torch.int64, device=dev)
            states[bi] = []
            if len(parents):
                cut = {k: v.index_select(1 if v.dim() >= 3 else 0, par).split(1, dim=1 if v.dim() >= 3 else 0) for k, v in new_state.items()}
                states[bi] = [{k: cut[k][j] for k in cut} for j in range(len(parents))]
            pos += n
    return beams


@pytest.mark.gpu
@pytest.mark.parametrize("with_caches", [True, False])
def test_decode_step_with_the_reference_signature_under_a_reference_style_search(with_caches, tmp_path):
    """Generator.decode_step(inp, state_dict, mem_dict, offset, topk) -- the reference's signature (generator.py:119) -- driven
    by a restatement of the reference's own search loop reproduces the reference's beams (beam_smatch goldens), with the
    memory of ``reference_memory`` (projected graph K/V travel with it) and with only the reference's five memory keys."""
    from gtos_amd.generator import Generator
    from gtos_amd.search import Beam
    meta, arrs = load_case("beam_smatch")
    dev = torch.device("cuda:0")
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    model = Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"],
                      cfg["inference_layers"], None, dev, depth_size=cfg.get("depth_size", 32)).to(dev)
    model.load_state_dict(state_dict_of(arrs))
    model.eval()
    batch = batch_of(meta, arrs, dev)
    for run in meta["runs"]:
        mem = model.reference_memory(batch)
        if not with_caches:
            mem = {k: mem[k] for k in ('graph_state', 'graph_padding_mask', 'probe', 'local_idx2token', 'cp_seq')}
        beams = [Beam(run["beam"], run["min_step"], run["max_step"]) for _ in range(mem['probe'].shape[1])]
        _search_by_batch_reference_style(model, beams, mem)
        for b, (beam, want) in enumerate(zip(beams, run["expect"])):
            tag = "run %s sentence %d" % ((run["beam"], run["max_step"], run["min_step"]), b)
            assert beam.steps == want["steps"], tag
            check_hyps([(h.seq, h.score) for h in beam.completed_hypotheses], want["finished"], tag + " finished")
            check_hyps([(h.seq, h.score) for h in beam.hypotheses], want["alive"], tag + " alive")


@pytest.mark.gpu
def test_hip_incremental_step_equals_full_prefix_recompute(tmp_path):
    """The K/V-cache decode step against the pinned oracle recomputing whole prefixes (teacher-forced random prefixes,
    several hypotheses per graph): next-token log-likelihoods agree to fp32 tolerance at every step."""
    from gtos_amd.generator import Generator
    from oracle import gtos_oracle as O
    meta, arrs = load_case()
    dev = torch.device("cuda:0")
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    args = (cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"], cfg["inference_layers"])
    model = Generator(vocabs, *ga, *args, None, dev).to(dev)
    ref = O.Generator(vocabs, *ga, *args)
    sd = state_dict_of(arrs)
    model.load_state_dict(sd)
    ref.load_state_dict(sd)
    model.eval()
    ref.eval()
    batch_cpu, batch = batch_of(meta, arrs), batch_of(meta, arrs, dev)
    words = [w for w in vocabs["token"].idx2token(list(range(4, vocabs["token"].size)))]
    rng = np.random.RandomState(3)
    owners = [0, 0, 1, 3, 3, 3, 5]                       # hypotheses per graph: ragged on purpose
    prefixes = [["<STR>"] for _ in owners]
    with torch.no_grad():
        g, gm, pr = model.encode_step(batch, train=False)
        g = g.contiguous()
        dec = model.decoder
        memory = {'probe': pr, 'graph_padding_mask': gm, 'cp_seq': batch['cp_seq'],
                  'tot_ext': 1 + int(batch['cp_seq'].max().item()), 'local_idx2token': batch['local_idx2token'],
                  'snt_ext_kv': [l.external_attn.project_kv(g) for l in model.snt_encoder.layers],
                  'inf_ext_kv': [l.external_attn.project_kv(g) for l in dec.inference_core.layers],
                  'align_kv': dec.token_generator.alignment_layer.project_kv(g)}
        rg, rgm, rpr = ref.encode_step(batch_cpu, train=False)
        state = None
        own_t = torch.tensor(owners, device=dev)
        V = vocabs["predictable_token"].size
        for step in range(5):
            state, results = model.decode_step_batched([p[-1] for p in prefixes], state, memory, own_t, step, 6)
            for h, b in enumerate(owners):
                want = O.next_token_ll(ref, rg[:, b:b + 1], rgm[:, b:b + 1], rpr[:, b:b + 1],
                                       batch_cpu['cp_seq'][:, b:b + 1], [prefixes[h]], vocabs)[0]
                top_s, top_i = torch.topk(want, 6)
                got_s = [s for _, s in results[h]]
                assert np.allclose(got_s, top_s.tolist(), atol=2e-4), (step, h)
            for p in prefixes:
                p.append(words[rng.randint(len(words))])


# ------------------------------------------------------------------------------------------------ real-data train step
def test_oracle_training_step_on_real_translator_batch(tmp_path):
    """Loss and every parameter gradient of translator/generator.py on a real dev.txt batch (ragged lengths, copy ids,
    real vocabularies) -- pins the oracle on data the synthetic fixtures do not resemble."""
    from oracle import gtos_oracle as O
    meta, arrs = load_case("beam_dep_dev")
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    model = O.Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"],
                        cfg["inference_layers"], depth_size=256)
    model.load_state_dict(state_dict_of(arrs))
    model.train()
    loss = model(batch_of(meta, arrs))
    loss.backward()
    assert abs(float(loss.detach()) - float(arrs["train/loss"])) < 1e-4
    want = {k[len("grad/"):]: T(arrs[k]) for k in arrs.files if k.startswith("grad/")}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=2e-3, atol=2e-5, msg=lambda m, k=k: "%s: %s" % (k, m))


@pytest.mark.gpu
@pytest.mark.parametrize("factored", [True, False])
def test_hip_training_step_on_real_translator_batch(factored, tmp_path):
    from gtos_amd.generator import Generator
    meta, arrs = load_case("beam_dep_dev")
    dev = torch.device("cuda:0")
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    model = Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, cfg["snt_layers"], cfg["graph_layers"],
                      cfg["inference_layers"], None, dev, depth_size=256, factored_relation=factored).to(dev)
    model.load_state_dict(state_dict_of(arrs))
    model.train()
    loss = model(batch_of(meta, arrs, dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(arrs["train/loss"])) < 1e-3
    want = {k[len("grad/"):]: T(arrs[k]) for k in arrs.files if k.startswith("grad/")}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    for k in want:
        torch.testing.assert_close(got[k].cpu(), want[k], rtol=2e-3, atol=1e-3, msg=lambda m, k=k: "%s: %s" % (k, m))


# ------------------------------------------------------------------------------------------------ batch assembly
def test_batchify_dependency_matches_reference_batch(tmp_path):
    """Raw dev.txt trees -> gtos_amd.data.batchify_dependency (C++ relation path + vocab tensorisers) must reproduce
    translator/data.py:batchify bit for bit: every tensor of the batch and the per-graph copy vocabularies."""
    from gtos_amd.data import batchify_dependency, read_dependency_file
    meta, arrs = load_case("beam_dep_dev")
    vocabs = make_vocabs(meta, tmp_path)
    trees = [(d, h, t, g) for d, h, t, g in meta["trees"]]
    got = batchify_dependency(trees, vocabs, n_threads=2)
    for k in ("concept", "concept_char", "concept_depth", "relation", "relation_bank", "relation_length", "cp_seq",
              "token_in", "token_char_in", "token_out"):
        want = T(arrs["batch/" + k])
        assert got[k].dtype == torch.int64 and tuple(got[k].shape) == tuple(want.shape), k
        assert torch.equal(got[k], want), k
    assert got["local_idx2token"] == [{int(k): v for k, v in d.items()} for d in meta["local_idx2token"]]
    assert got["local_token2idx"] == meta["local_token2idx"]
    # the text reader: the reference's 4-line format
    path = tmp_path / "mini.txt"
    path.write_text("".join(" ".join(str(x) for x in part) + "\n" for tr in trees[:2] for part in tr), encoding="utf8")
    back = read_dependency_file(str(path))
    assert [list(map(list, tr)) for tr in back] == [[list(p) for p in tr] for tr in trees[:2]]


def test_dependency_loader_batches_like_the_reference():
    """Batch composition and order of translator/data.py:DataLoader on the reference's dev.txt (2169 examples; the
    fixture carries their sizes, which is all the policy looks at) for four (batch_size, train, seed) settings."""
    import random
    from gtos_amd.data import DependencyLoader
    g = json.load(open(os.path.join(GOLDEN_DIR, "loader_dep_dev.json")))
    trees = [(["d"] * n, [0] * n, ["w"] * n, ["t"] * m) for n, m in g["sizes"]]
    for run in g["runs"]:
        assert run["n_examples"] == len(trees)
        random.seed(run["seed"])
        dl = DependencyLoader(None, trees, run["batch_size"], run["train"])
        assert dl.batch_indices() == run["batches"], (run["batch_size"], run["train"], run["seed"])
        # an explicit generator gives the same result as the module-level one
        dl2 = DependencyLoader(None, trees, run["batch_size"], run["train"], rng=random.Random(run["seed"]))
        assert dl2.batch_indices() == run["batches"]


@pytest.mark.gpu
def test_end_to_end_translator_flow_on_gpu(tmp_path):
    """The pieces compose like translator/train.py + work.py: raw trees -> DependencyLoader/batchify -> Trainer steps
    (fp32, dropout on) -> checkpoint in the reference's format -> build_from_checkpoint -> beam search."""
    import argparse
    import random
    from gtos_amd.checkpoint import save_checkpoint, build_from_checkpoint
    from gtos_amd.data import DependencyLoader
    from gtos_amd.generator import Generator
    from gtos_amd.train import Trainer
    meta, arrs = load_case("beam_dep_dev")
    dev = torch.device("cuda:0")
    vdir = write_vocab_files(meta, tmp_path)
    from gtos_amd.vocab import load_vocabs
    vocabs = load_vocabs(vdir)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    torch.manual_seed(3)
    model = Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.1, cfg["snt_layers"], cfg["graph_layers"],
                      cfg["inference_layers"], None, dev, depth_size=256).to(dev)
    model.train()
    trainer = Trainer(model, cfg["d"], warmup_steps=200, world_size=1)
    trees = [tuple(t) for t in meta["trees"]]
    random.seed(5)
    loader = DependencyLoader(vocabs, trees, 1200, for_train=True, n_threads=2)   # size units: n_src**2 + n_tgt per tree
    losses = []
    for epoch in range(40):
        for batch in loader:
            batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
            losses.append(trainer.step(batch))
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    args = argparse.Namespace(
        token_char_dim=ga[0], token_dim=ga[1], concept_char_dim=ga[2], concept_dim=ga[3], cnn_filters=ga[4],
        char2word_dim=ga[5], char2concept_dim=ga[6], rel_dim=ga[7], rnn_hidden_size=ga[8], rnn_num_layers=ga[9],
        embed_dim=cfg["d"], ff_embed_dim=cfg["ff"], num_heads=cfg["H"], dropout=0.1, snt_layers=cfg["snt_layers"],
        graph_layers=cfg["graph_layers"], inference_layers=cfg["inference_layers"], pretrained_file=None,
        **{n: os.path.join(vdir, n) for n in meta["files"]})
    path = str(tmp_path / "ckpt")
    save_checkpoint(path, args, model)
    served, _, _ = build_from_checkpoint(path, dev)            # depth table size (256) is read off the state_dict
    assert served.concept_depth.weight.shape[0] == 256 and not served.training
    batch = next(iter(DependencyLoader(vocabs, trees, 10 ** 9, for_train=False)))
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    beams = served.work(batch, 3, 12)
    assert len(beams) == len(trees)
    for beam in beams:
        best = beam.get_k_best(1, 0.6)[0]
        assert best.seq[0] == "<STR>" and len(best.seq) >= 2 and best.score == best.score


def test_cached_step_inputs_equal_the_tensorisers(tmp_path):
    """Generator.prepare_incremental_input caches (token id, character row) per string: same tensors as the generic path."""
    from gtos_amd.generator import Generator
    from gtos_amd.vocab import lists_to_tensor, strings_to_char_tensor
    meta, _ = load_case()
    vocabs = make_vocabs(meta, tmp_path)
    cfg = meta["cfg"]
    ga = [[tuple(f) for f in a] if isinstance(a, list) else a for a in cfg["gen_args"]]
    m = Generator(vocabs, *ga, cfg["d"], cfg["ff"], cfg["H"], 0.0, 1, 2, 3, None, torch.device("cpu"))
    seq = [["the"], ["boy"], ["never-seen-token"], ["the"], ["a-token-longer-than-twenty-characters"], ["<STR>"]]
    for _ in range(2):                                     # second pass is served from the cache
        t, c = m.prepare_incremental_input(seq)
        assert torch.equal(t, lists_to_tensor(seq, vocabs['token']))
        assert torch.equal(c, strings_to_char_tensor(seq, vocabs['token_char']))


# ------------------------------------------------------------------------------------------------ training-time UNK noise
def test_unk_rate_noise_matches_the_reference_loader(tmp_path):
    """translator/data.py:77-99 under --unk_rate: with the same `random` state, batchify_dependency replaces exactly the
    concept / token_in entries the reference replaces, and never touches token_out / cp_seq (tests/golden/make_golden_unk.py)."""
    import random
    from gtos_amd.data import batchify_dependency, DependencyLoader
    meta = json.load(open(os.path.join(GOLDEN_DIR, "beam_dep_dev.json")))
    vocabs = make_vocabs(meta, tmp_path)
    g = np.load(os.path.join(GOLDEN_DIR, "unk_dep_dev.npz"))
    trees = [tuple(t) for t in meta["trees"]]
    clean = batchify_dependency(trees, vocabs)
    for k, (rate, seed) in enumerate(g["settings"]):
        rng = random.Random(int(seed))
        b = batchify_dependency(trees, vocabs, unk_rate=float(rate), rng=rng, replay_reference_draws=True)
        for key in ("concept", "token_in"):
            assert torch.equal(b[key], T(g["%d/%s" % (k, key)])), (rate, seed, key)
        for key in ("token_out", "cp_seq"):
            assert torch.equal(b[key], clean[key]) and torch.equal(b[key], T(g["%d/%s" % (k, key)]))
        if rate > 0:
            assert not torch.equal(b["concept"], clean["concept"])
    # the module-level default generator is `random`, like the reference
    random.seed(int(g["settings"][0][1]))
    b = batchify_dependency(trees, vocabs, unk_rate=float(g["settings"][0][0]))
    assert torch.equal(b["concept"], T(g["0/concept"]))
    # the loader threads its rate and generator through (translator/data.py:218-219,265)
    dl = DependencyLoader(vocabs, trees, 10 ** 9, for_train=False, rng=random.Random(int(g["settings"][1][1])))
    dl.set_unk_rate(float(g["settings"][1][0]))
    b = next(iter(dl))
    assert torch.equal(b["concept"], T(g["1/concept"]))
    assert abs(float((b["token_in"] == vocabs['token'].unk_idx).float().mean()) - float((T(g["1/token_in"]) == vocabs['token'].unk_idx).float().mean())) < 0.1


def test_char_row_cache_is_thread_safe(tmp_path):
    """Loader THREADS share the string -> id-row cache of strings_to_char_tensor (round-3 advisor finding: an index could be published
    before its row existed, or taken twice): eight threads on a cold cache must give the single-thread tensors."""
    import random
    import sys
    import threading
    from gtos_amd import vocab as V
    from gtos_amd.vocab import Vocab, strings_to_char_tensor
    chars = [chr(ord('a') + i) for i in range(26)]
    f = tmp_path / "char_vocab"
    f.write_text("".join("%s\t10\n" % c for c in chars))
    cv = Vocab(str(f), 1, [V.STR, V.END])
    rng = random.Random(3)
    words = ["".join(rng.choice(chars) for _ in range(rng.randint(1, 9))) for _ in range(3000)]
    batches = [[[rng.choice(words) for _ in range(rng.randint(1, 12))] for _ in range(6)] for _ in range(160)]
    V._CHAR_ROWS.clear()
    want = [strings_to_char_tensor(b, cv) for b in batches]
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-6)
    try:
        for _ in range(3):
            V._CHAR_ROWS.clear()
            got = [None] * len(batches)

            def work(t):
                for i in range(t, len(batches), 8):
                    got[i] = strings_to_char_tensor(batches[i], cv)
            ths = [threading.Thread(target=work, args=(t,)) for t in range(8)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            assert all(torch.equal(a, b) for a, b in zip(want, got))
    finally:
        sys.setswitchinterval(old)
