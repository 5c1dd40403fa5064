"""Pins oracle/gtos_oracle.py to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub, T, opt_mask
from oracle import gtos_oracle as O

TOL = dict(rtol=2e-5, atol=2e-5)


def check_param_grads(module, g):
    want = sub(g, "grad/")
    got = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    assert set(want) == set(got)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-4, atol=2e-5, msg=lambda m, k=k: "%s: %s" % (k, m))


@pytest.mark.parametrize("name", ["gt_tiny", "gt_pad", "gt_mask", "gt_hd64", "gt_h8"])
def test_graph_transformer(name):
    g = load_golden(name)
    L, d, ff, H, n, B = [int(v) for v in g["cfg"]]
    m = O.GraphTransformer(L, d, ff, H, 0.0)
    m.load_state_dict(sub(g, "sd/"))
    x, rel = T(g["x"]).requires_grad_(), T(g["relation"]).requires_grad_()
    pad, am = opt_mask(g["pad"]), opt_mask(g["attn_mask"])
    out = m(x, rel, self_padding_mask=pad, self_attn_mask=am)
    torch.testing.assert_close(out, T(g["out"]), **TOL)
    attn = m.get_attn_weights(x, rel, self_padding_mask=pad, self_attn_mask=am)
    torch.testing.assert_close(attn, T(g["attn"]), **TOL)
    (out * T(g["wout"])).sum().backward()
    torch.testing.assert_close(x.grad, T(g["dx"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(rel.grad, T(g["drelation"]), rtol=1e-4, atol=2e-5)
    check_param_grads(m, g)


@pytest.mark.parametrize("name", ["relenc_small", "relenc_wide"])
def test_relation_encoder(name):
    g = load_golden(name)
    V, rel_dim, d, hid, R, Lmax = [int(v) for v in g["cfg"]]
    m = O.RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, 0.0)
    m.load_state_dict(sub(g, "sd/"))
    out = m(T(g["tokens"]), T(g["lengths"]))
    torch.testing.assert_close(out, T(g["out"]), **TOL)
    (out * T(g["wout"])).sum().backward()
    check_param_grads(m, g)


@pytest.mark.parametrize("name", ["tl_self", "tl_kv"])
def test_transformer_layer(name):
    g = load_golden(name)
    d, ff, H, Tq, S, B, with_kv = [int(v) for v in g["cfg"]]
    m = O.TransformerLayer(d, ff, H, 0.0, with_external=True)
    m.load_state_dict(sub(g, "sd/"))
    x, ext = T(g["x"]).requires_grad_(), T(g["ext"]).requires_grad_()
    kv = T(g["kv"]).requires_grad_() if with_kv else None
    out, sw, ew = m(x, kv, T(g["self_pad"]), T(g["attn_mask"]), ext, T(g["ext_pad"]), need_weights=True)
    torch.testing.assert_close(out, T(g["out"]), **TOL)
    torch.testing.assert_close(sw, T(g["self_w"]), **TOL)
    torch.testing.assert_close(ew, T(g["ext_w"]), **TOL)
    (out * T(g["wout"])).sum().backward()
    torch.testing.assert_close(x.grad, T(g["dx"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(ext.grad, T(g["dext"]), rtol=1e-4, atol=2e-5)
    if with_kv:
        torch.testing.assert_close(kv.grad, T(g["dkv"]), rtol=1e-4, atol=2e-5)
    check_param_grads(m, g)


def build_small_generator(g, cls=None):
    from tests_support import SMALL_VOCAB
    d, ff, H, gl = [int(v) for v in g["cfg"]]
    vocabs = {k: O.VocabSpec(v, 0) for k, v in SMALL_VOCAB.items()}
    m = (cls or O.Generator)(vocabs, 8, 12, 8, 12, [(3, 16)], 10, 10, 6, 8, 2, d, ff, H, 0.0, 1, gl, 2)
    sd = sub(g, "sd/")
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("name", ["gen_small", "gen_padded"])
def test_generator(name):
    g = load_golden(name)
    m = build_small_generator(g)
    batch, ebatch = sub(g, "batch/"), sub(g, "ebatch/")
    m.train()
    graph, gmask, probe = m.encode_step(batch)
    torch.testing.assert_close(graph, T(g["graph"]), **TOL)
    torch.testing.assert_close(probe, T(g["probe"]), **TOL)
    assert torch.equal(gmask, T(g["gmask"]))
    loss = m(batch)
    torch.testing.assert_close(loss, T(g["loss"]), **TOL)
    loss.backward()
    check_param_grads(m, g)
    m.eval()
    with torch.no_grad():
        egraph, _, eprobe = m.encode_step(ebatch, train=False)
    torch.testing.assert_close(egraph, T(g["egraph"]), **TOL)
    torch.testing.assert_close(eprobe, T(g["eprobe"]), **TOL)


def test_adam_and_lr():
    g = load_golden("adam_steps")
    warmup, d = [int(v) for v in g["cfg"]]
    p = [T(g["p0_init"]), T(g["p1_init"])]
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    wd = [1e-4, 0.0]
    for step in range(1, 4):
        lr = O.inverse_sqrt_lr(d, step, warmup)
        assert abs(lr - float(g["lrs"][step - 1])) < 1e-12
        gs = [T(g["g0_%d" % step]), T(g["g1_%d" % step])]
        coef, norm = O.clip_coef(gs, 1.0)
        torch.testing.assert_close(norm, T(g["norm_%d" % step]), rtol=1e-5, atol=1e-6)
        for i in range(2):
            p[i], m[i], v[i] = O.adam_step(p[i], gs[i] * coef, m[i], v[i], lr, wd[i])
        torch.testing.assert_close(p[0], T(g["p0_%d" % step]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(p[1], T(g["p1_%d" % step]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m[0], T(g["m0_%d" % step]), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(v[0], T(g["v0_%d" % step]), rtol=1e-5, atol=1e-7)


def test_sinusoid_and_causal_mask():
    tab = O.sinusoid_table(7, 8)
    assert tab.shape == (7, 8)
    assert torch.allclose(tab[0], torch.tensor([0., 0, 0, 0, 1, 1, 1, 1]))
    assert abs(float(tab[3, 0]) - np.sin(3.0)) < 1e-6 and abs(float(tab[3, 4]) - np.cos(3.0)) < 1e-6
    cm = O.causal_mask(3)
    assert cm.tolist() == [[False, True, True], [False, False, True], [False, False, False]]
