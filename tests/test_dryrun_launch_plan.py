"""The GPU path's Python glue exercised on the CPU (tests/dryrun.py): a full training step of a BASELINE configuration with every
libgtos_hip.so launch recorded instead of executed.  No numbers are checked (kernel outputs are uninitialised memory) -- what is:
every ``call()`` site the step reaches passes the binding's signature (argument count and kind), the production switches select the
entry points DESIGN.md says they do, and the launch plan of a step is a pure function of the batch (two steps, same plan)."""
import torch

from dryrun import DryRun
from gtos_amd import synth


def _trainer(cfg, dtype=torch.bfloat16, factored=True):
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.train import Trainer
    dev = torch.device("cpu")
    model = build_generator(Generator, cfg, dev, factored_relation=factored).to(dev)
    model.set_compute_dtype(dtype)
    model.train()
    return Trainer(model, synth.CONFIGS[cfg]["d"], warmup_steps=2000, compute_dtype=dtype, world_size=1, rank=0)


def _plan(rec, n0):
    """the recorded calls from index n0 on as (name, scalar arguments): pointers dropped (addresses change from step to step)"""
    import ctypes
    from gtos_amd import _lib
    out = []
    for name, args in rec.calls[n0:]:
        sig = _lib.SIGNATURES[name]
        out.append((name,) + tuple(a for c, a in zip(sig, args) if c is not ctypes.c_void_p and c is not ctypes.c_uint64))   # (uint64: dropout seeds)
    return out


def test_training_step_launch_plan_c1_bf16_trie_factored():
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    with DryRun() as rec:
        trainer = _trainer("C1")
        batch, _ = synth.make_config_batch("C1", rank=0)
        attach_relation_index(attach_path_trie(batch))
        trainer.step(batch, sync=False)
        n1 = len(rec.calls)
        trainer.step(batch, sync=False)
        n2 = len(rec.calls)
        trainer.step(batch, sync=False)
        plan_a, plan_b = _plan(rec, n1)[:n2 - n1], _plan(rec, n2)
        hist = {}
        for name, _ in rec.calls[n2:]:
            hist[name] = hist.get(name, 0) + 1
    assert plan_a == plan_b                                       # same batch, same plan (shapes, flags, split-K factors, launch order)
    # every operand of the GEMM / attention (incl. the bank-gradient chunk lists) / LayerNorm / column-sum / GRU-step / segment-sum
    # launches of the three steps lay inside its tensor's storage, every gathered row index inside its table (real index arrays)
    assert rec.extent_checks > 1000 and rec.unknown_ptrs == 0     # (1,074 since round 6: a small layer's weight AND bias gradient are one checked job of gtos_gemm_tn_batch)
    L = synth.CONFIGS["C1"]["layers"]
    # the production path of DESIGN.md: factored attention (one bank-gradient launch per graph layer), the RelationEncoder with the
    # reference's dropout semantics on the packed-path kernels (round 5: one embedding launch off the batch's own sort order, fused steps
    # -- never the per-row cell kernels --, the layers' input gradients inside the backward step launches, ONE grouped weight-gradient
    # product per layer and direction), fused copy / NLL, device-side step control and ONE fused optimizer sweep per segment
    assert hist["gtos_rel_attn_bwd_bank"] == L
    assert hist["gtos_rel_attn_fwd"] == hist["gtos_rel_attn_bwd"] >= L
    assert hist.get("gtos_gru_step_fwd", 0) > 0 and hist.get("gtos_gru_step_bwd_fused", 0) == hist["gtos_gru_step_fwd"] + 4
    assert hist["gtos_gru_weight_grads"] == 4 and hist["gtos_embed_packed_paths"] == 1 and "gtos_gru_step_bwd" not in hist
    assert "gtos_gru_cell_fwd" not in hist and "gtos_relation_gather_mean" not in hist
    assert hist["gtos_copy_nll_fwd"] == hist["gtos_copy_nll_bwd"] == 1
    assert hist["gtos_step_control"] == 2 and "gtos_adam_step" not in hist and hist["gtos_adam_step_ctl"] >= 1   # (flag phase + apply phase)
    assert 300 < sum(hist.values()) < 800, sum(hist.values())     # 507 at the end of round 3


def test_training_step_launch_plan_fp32_per_row_dense():
    """the fp32 parity mode takes the other branches: dense relation operand, per-row GRU (cell kernels), no trie / index needed"""
    with DryRun() as rec:
        trainer = _trainer("C1", dtype=torch.float32, factored=False)
        batch, _ = synth.make_config_batch("C1", rank=0)
        trainer.step(batch, sync=False)
        n1 = len(rec.calls)
        trainer.step(batch, sync=False)
        hist = {}
        for name, _ in rec.calls[n1:]:
            hist[name] = hist.get(name, 0) + 1
    assert hist.get("gtos_gru_cell_fwd", 0) > 0 and "gtos_gru_step_fwd" not in hist
    assert "gtos_rel_attn_bwd_bank" not in hist and hist["gtos_rel_attn_fwd"] == hist["gtos_rel_attn_bwd"]


def test_dryrun_restores_what_it_patched():
    before = (torch.cuda.current_stream, torch.cuda.Event, torch.Tensor.is_cuda)
    with DryRun():
        assert torch.zeros(1).is_cuda
    assert (torch.cuda.current_stream, torch.cuda.Event, torch.Tensor.is_cuda) == before
    assert not torch.zeros(1).is_cuda


def test_training_step_launch_plan_dependency_flavour_and_padded_batch():
    """C3 (translator flavour: dependency trees, depth ids up to the sentence length) and a padded AMR batch (graphs of different sizes:
    masks, ragged tries) reach their kernels with well-formed arguments too."""
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    with DryRun() as rec:
        for cfg, kw in (("C3", dict(B=4)), ("C1", dict(padded=True))):
            trainer = _trainer(cfg)
            batch, _ = synth.make_config_batch(cfg, rank=0, **kw)
            attach_relation_index(attach_path_trie(batch))
            n0 = len(rec.calls)
            trainer.step(batch, sync=False)
            names = {n for n, _ in rec.calls[n0:]}
            assert {"gtos_rel_attn_fwd", "gtos_rel_attn_bwd_bank", "gtos_gru_step_fwd", "gtos_copy_nll_fwd", "gtos_adam_step_ctl"} <= names, cfg


def test_training_step_glue_over_random_model_shapes():
    """Random small model / batch geometries (1-8 heads of width 8-64 incl. 24 and 40, GRU widths with and without the trie path, one node,
    one token, padded batches, both flavours, both precisions, dense and factored relation operand) through two training steps of the
    dry run: no branch of the glue trips over a shape, and every launch it makes is well-formed.  (210 such draws ran clean when this
    test was written; it keeps a dozen.)"""
    import numpy as np
    from gtos_amd.config import default_vocabs
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    from gtos_amd.train import Trainer
    rng = np.random.RandomState(7)
    for it in range(12):
        H, hd = int(rng.choice([1, 2, 4, 6, 8])), int(rng.choice([8, 16, 24, 32, 40, 64]))
        d = H * hd
        dtype = [torch.bfloat16, torch.float32][it % 2]
        kind = ["amr", "dep"][(it // 2) % 2]
        args = dict(word_char_dim=8, word_dim=int(rng.choice([12, 16, 300])), concept_char_dim=8, concept_dim=int(rng.choice([12, 16])),
                    cnn_filters=[(3, 16)], char2word_dim=int(rng.choice([8, 16])), char2concept_dim=8, rel_dim=int(rng.choice([10, 12, 100])),
                    rnn_hidden_size=int(rng.choice([64, 128, 192, 256])), rnn_num_layers=2, embed_dim=d, ff_embed_dim=int(rng.choice([d, 2 * d, 72])),
                    num_heads=H, dropout=float(rng.choice([0.0, 0.2])), snt_layers=int(rng.choice([1, 2])), graph_layers=int(rng.choice([1, 2, 3])),
                    inference_layers=int(rng.choice([1, 3])), pretrained_file=None)
        with DryRun() as rec:
            torch.manual_seed(1)
            model = Generator(default_vocabs(), device=torch.device("cpu"), depth_size=256 if kind == "dep" else 32,
                              factored_relation=bool(rng.randint(0, 2)), **args)
            model.set_compute_dtype(dtype)
            model.train()
            trainer = Trainer(model, d, warmup_steps=10, compute_dtype=dtype, world_size=1, rank=0)
            batch, _ = synth.make_batch(1000 + it, int(rng.randint(1, 6)), int(rng.randint(1, 12)), int(rng.randint(1, 9)), kind=kind,
                                        padded=bool(rng.randint(0, 2)))
            if dtype == torch.bfloat16:
                attach_relation_index(attach_path_trie(batch))
            trainer.step(batch, sync=False)
            trainer.step(batch, sync=False)
            assert len(rec.calls) > 100, (it, len(rec.calls))


def test_eval_mode_encoder_glue_on_an_every_alternative_batch():
    """eval mode (no grad) on an eval batch (relation [n,n,B,K]: the K-path mean of generator/generator.py:76-78 as a derived bank +
    the factored kernels): the encoder half of ``Generator.work`` under the dry run, bf16 and fp32"""
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    for dtype in (torch.bfloat16, torch.float32):
        with DryRun() as rec:
            model = build_generator(Generator, "C1", torch.device("cpu"))
            model.set_compute_dtype(dtype)
            model.eval()
            batch, _ = synth.make_config_batch("C1", train=False)
            assert batch["relation"].dim() == 4
            with torch.no_grad():
                model.encode_step(attach_path_trie(batch), train=False)
            names = {n for n, _ in rec.calls}
        assert "gtos_rel_attn_fwd" in names and "gtos_relation_gather_mean" in names and "gtos_rel_attn_bwd" not in names


def test_dryrun_checks_gemm_operand_extents():
    """the recorder knows the storage behind every pointer that went through ptr(): a gtos_gemm operand whose rows x leading dimension
    run past its storage is refused (the real launches of the plans above all pass this check: column blocks of gradient slabs, row blocks
    of packed projections, transposed weight views)"""
    import pytest
    from gtos_amd import ops
    from gtos_amd._lib import call
    with DryRun() as rec:
        a, w = torch.zeros(6, 16, dtype=torch.bfloat16), torch.zeros(8, 16, dtype=torch.bfloat16)
        ops.gemm(a, w, trans_b=True)
        view = torch.zeros(6, 16, dtype=torch.bfloat16)[:, :8]                  # a column block: [6, 8] with row stride 16
        ops.gemm(view, torch.zeros(4, 8, dtype=torch.bfloat16), trans_b=True)
        assert rec.extent_checks == 2
        b4, c7 = torch.zeros(4, 8, dtype=torch.bfloat16), torch.zeros(7, 4, dtype=torch.bfloat16)
        with pytest.raises(AssertionError, match="past its storage"):           # 7 rows claimed of a 6-row operand
            call("gtos_gemm", 1, 1, 0, 1, 7, 4, 8, ops.ptr(view), 16, ops.ptr(b4), 8, ops.ptr(c7), 4, None, 0, 0.0, 0, 0, 1, None, 0, 0)
        with pytest.raises(AssertionError, match="leading dimension"):
            call("gtos_gemm", 1, 1, 0, 1, 6, 4, 8, ops.ptr(view), 4, ops.ptr(b4), 8, ops.ptr(c7), 4, None, 0, 0.0, 0, 0, 1, None, 0, 0)


def test_beam_search_glue_under_the_dry_run():
    """``Generator.work`` (encode once, incremental decoding with K/V caches, beam bookkeeping of gtos_amd/search.py) on an eval batch:
    the hypotheses are meaningless (kernel outputs are uninitialised memory) but every launch of the decode loop is well-formed and in
    bounds -- cached-key attention with growing S, the copy / generate likelihood rows, the beam reorders."""
    from gtos_amd.config import generator_args
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    with DryRun() as rec:
        vocabs = synth.synth_vocabs()
        torch.manual_seed(1)
        model = Generator(vocabs, device=torch.device("cpu"), depth_size=32, **generator_args(synth.CONFIGS["C1"]))
        model.set_compute_dtype(torch.bfloat16)
        model.eval()
        batch, _ = synth.make_config_batch("C1", train=False)
        batch = attach_path_trie(batch)
        pv, cp = vocabs['predictable_token'], batch['cp_seq']
        batch['local_idx2token'] = [{int(i): "copy%d" % int(i) for i in cp[:, b].tolist() if i >= pv.size} for b in range(cp.shape[1])]
        beams = model.work(batch, 4, 5)
        hist = rec.histogram()
    assert len(beams) == batch['concept'].shape[1] and max(b.steps for b in beams) == 5
    assert hist["gtos_copy_ll_fwd"] == 5 and "gtos_rel_attn_bwd" not in hist and rec.extent_checks > 300 and rec.unknown_ptrs == 0
