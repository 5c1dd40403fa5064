"""Generate the golden vectors in this directory by running the REFERENCE itself.

Run in the build container only (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's hot-path modules unmodified from /root/reference/generator, applies the
harness-side shims SURVEY.md section 8c lists (monkey-patches; the reference files are not edited),
feeds fixed-seed inputs and stores inputs / state_dicts / outputs / gradients as float32 ``.npz``
fixtures.  The fixtures are data only.  tests/test_oracle_golden.py pins ``oracle/gtos_oracle.py`` to
them; the GPU parity tests compare the HIP path against them and against the pinned oracle.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/generator"
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

np.int = int                                  # shim (1): numpy >= 1.24 dropped np.int

import graph_transformer as ref_gt            # noqa: E402
import transformer as ref_tf                  # noqa: E402
import encoder as ref_enc                     # noqa: E402
import decoder as ref_dec                     # noqa: E402
import generator as ref_gen                   # noqa: E402
import adam as ref_adam                       # noqa: E402

from gtos_amd.synth import make_batch         # noqa: E402
from oracle.gtos_oracle import VocabSpec      # noqa: E402  (a namedtuple: .size, .padding_idx)

# shim (2): q *= scaling on a chunk view trips autograd on current torch -> hand out clones
_orig_qkv = ref_tf.MultiheadAttention.in_proj_qkv
ref_tf.MultiheadAttention.in_proj_qkv = lambda self, q: tuple(t.clone() for t in _orig_qkv(self, q))
# shim (3): masked_fill_ needs bool masks now
ref_tf.SelfAttentionMask.get_mask = staticmethod(
    lambda size: torch.ones((size, size), dtype=torch.bool).triu_(1))


def to_np(d, prefix):
    return {prefix + k: v.detach().cpu().numpy() for k, v in d.items()}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-34s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def grads_of(module, loss, extra):
    loss.backward()
    g = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    return to_np(g, "grad/"), [e.grad for e in extra]


# ---------------------------------------------------------------- GraphTransformer cases
def graph_transformer_case(name, seed, L, d, ff, H, n, B, pad_lens=None, use_attn_mask=False):
    torch.manual_seed(seed)
    m = ref_gt.GraphTransformer(L, d, ff, H, dropout=0.0)
    for p in m.parameters():                      # biases/LN away from 0/1 so they are exercised
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    x = torch.randn(n, B, d, requires_grad=True)
    rel = (0.5 * torch.randn(n, n, B, d)).requires_grad_()
    pad = None
    if pad_lens is not None:
        pad = torch.zeros(n, B, dtype=torch.bool)
        for b, ln in enumerate(pad_lens):
            pad[ln:, b] = True
    amask = torch.ones(n, n, dtype=torch.bool).triu_(1) if use_attn_mask else None
    out = m(x, rel, self_padding_mask=pad, self_attn_mask=amask)
    attn = m.get_attn_weights(x, rel, self_padding_mask=pad, self_attn_mask=amask)
    wout = torch.randn_like(out)
    pg, (dx, drel) = grads_of(m, (out * wout).sum(), [x, rel])
    save(name, cfg=np.array([L, d, ff, H, n, B]), x=x, relation=rel,
         pad=pad if pad is not None else np.zeros((0,)), attn_mask=amask if amask is not None else np.zeros((0,)),
         out=out, attn=attn, wout=wout, dx=dx, drelation=drel, **to_np(m.state_dict(), "sd/"), **pg)


# ---------------------------------------------------------------- RelationEncoder
def relation_encoder_case(name, seed, V, rel_dim, d, hid, R, Lmax):
    torch.manual_seed(seed)
    m = ref_enc.RelationEncoder(VocabSpec(V, 0), rel_dim, d, hid, 2, dropout=0.0)
    lengths = torch.randint(1, Lmax + 1, (R,))
    lengths[0] = Lmax
    toks = torch.randint(2, V, (Lmax, R))
    for r in range(R):
        toks[lengths[r]:, r] = 0
    out = m(toks, lengths)
    wout = torch.randn_like(out)
    pg, _ = grads_of(m, (out * wout).sum(), [])
    save(name, cfg=np.array([V, rel_dim, d, hid, R, Lmax]), tokens=toks, lengths=lengths, out=out,
         wout=wout, **to_np(m.state_dict(), "sd/"), **pg)


# ---------------------------------------------------------------- TransformerLayer (decoder blocks)
def transformer_layer_case(name, seed, d, ff, H, T, S, B, with_kv):
    torch.manual_seed(seed)
    m = ref_tf.TransformerLayer(d, ff, H, dropout=0.0, with_external=True)
    for p in m.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    x = torch.randn(T, B, d, requires_grad=True)
    Tk = T + 2 if with_kv else T
    kv = torch.randn(Tk, B, d, requires_grad=True) if with_kv else None
    ext = torch.randn(S, B, d, requires_grad=True)
    self_pad = torch.zeros(Tk, B, dtype=torch.bool)
    self_pad[Tk - 1:, 0] = True
    ext_pad = torch.zeros(S, B, dtype=torch.bool)
    ext_pad[S - 2:, B - 1] = True
    amask = torch.ones(T, Tk, dtype=torch.bool).triu_(1)
    out, sw, ew = m(x, kv, self_pad, amask, ext, ext_pad, need_weights=True)
    wout = torch.randn_like(out)
    pg, gs = grads_of(m, (out * wout).sum(), [x, ext] + ([kv] if with_kv else []))
    extra = dict(kv=kv, dkv=gs[2]) if with_kv else {}
    save(name, cfg=np.array([d, ff, H, T, S, B, int(with_kv)]), x=x, ext=ext, self_pad=self_pad,
         ext_pad=ext_pad, attn_mask=amask, out=out, self_w=sw, ext_w=ew, wout=wout, dx=gs[0],
         dext=gs[1], **extra, **to_np(m.state_dict(), "sd/"), **pg)


# ---------------------------------------------------------------- full Generator (train loss + eval encode)
SMALL_VOCAB = dict(concept=60, token=70, predictable_token=50, relation=26, concept_char=20, token_char=22)


def build_ref_generator(vocab, d, ff, H, gl, depth_seed):
    vocabs = {k: VocabSpec(v, 0) for k, v in vocab.items()}
    torch.manual_seed(depth_seed)
    m = ref_gen.Generator(vocabs, 8, 12, 8, 12, [(3, 16)], 10, 10, 6, 8, 2, d, ff, H, 0.0, 1, gl, 2,
                          None, torch.device("cpu"))
    # the reference zero-inits concept_depth and most biases; perturb so every term matters
    for p in m.parameters():
        if p.dim() == 1 or p.abs().sum() == 0:
            p.data.add_(0.05 * torch.randn_like(p))
    return m


def generator_case(name, seed, padded):
    d, ff, H, gl = 32, 48, 4, 2
    m = build_ref_generator(SMALL_VOCAB, d, ff, H, gl, seed)
    batch, _ = make_batch(900 + seed, 3, 6, 5, kind="amr", extra_frac=0.5, train=True, padded=padded,
                          vocab=SMALL_VOCAB)
    batch["relation"] = batch["relation"].contiguous()          # shim (4)
    m.train()                                                   # dropout p = 0: deterministic
    graph, gmask, probe = m.encode_step(batch)
    loss = m(batch)
    pg, _ = grads_of(m, loss, [])
    # eval-mode relation aggregation over alternative shortest paths (generator.py:83-88)
    ebatch, _ = make_batch(900 + seed, 3, 6, 5, kind="amr", extra_frac=0.5, train=False, padded=padded,
                           vocab=SMALL_VOCAB)
    m.eval()
    with torch.no_grad():
        egraph, egmask, eprobe = m.encode_step(ebatch, train=False)
        eattn = m.encoder_attn(ebatch)
    save(name, cfg=np.array([d, ff, H, gl]), loss=loss, graph=graph, probe=probe, gmask=gmask,
         egraph=egraph, eprobe=eprobe, eattn=eattn,
         **to_np(batch, "batch/"), **to_np(ebatch, "ebatch/"), **to_np(m.state_dict(), "sd/"), **pg)


# ---------------------------------------------------------------- optimizer step
def adam_case(name, seed):
    torch.manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    opt = ref_adam.AdamWeightDecayOptimizer([{"params": [ps[0]], "weight_decay": 1e-4},
                                             {"params": [ps[1]], "weight_decay": 0.0}],
                                            lr=1e-3, betas=(0.9, 0.999), eps=1e-6)
    rec = dict(p0_init=ps[0].detach().clone(), p1_init=ps[1].detach().clone())
    warmup, d = 4, 32
    lrs = []
    for step in range(1, 4):
        gs = [torch.randn_like(p) * 3 for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        lr = d ** -0.5 * min(step ** -0.5, step * (warmup ** -1.5))    # update_lr, train.py:81-83
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        lrs.append(lr)
        rec.update({"g0_%d" % step: gs[0], "g1_%d" % step: gs[1], "norm_%d" % step: norm,
                    "p0_%d" % step: ps[0].detach().clone(), "p1_%d" % step: ps[1].detach().clone(),
                    "m0_%d" % step: opt.state[ps[0]]["exp_avg"].clone(),
                    "v0_%d" % step: opt.state[ps[0]]["exp_avg_sq"].clone()})
    save(name, cfg=np.array([warmup, d]), lrs=np.array(lrs), **rec)


if __name__ == "__main__":
    torch.set_num_threads(4)
    graph_transformer_case("gt_tiny", 1, L=2, d=16, ff=32, H=1, n=2, B=1)
    graph_transformer_case("gt_pad", 2, L=2, d=32, ff=48, H=4, n=5, B=3, pad_lens=[5, 3, 4])
    graph_transformer_case("gt_mask", 3, L=1, d=32, ff=64, H=4, n=7, B=3, pad_lens=[7, 7, 2], use_attn_mask=True)
    graph_transformer_case("gt_hd64", 4, L=2, d=128, ff=256, H=2, n=9, B=2, pad_lens=[9, 6])
    graph_transformer_case("gt_h8", 5, L=1, d=256, ff=64, H=8, n=21, B=2, pad_lens=[21, 13])
    relation_encoder_case("relenc_small", 6, V=26, rel_dim=6, d=32, hid=8, R=19, Lmax=8)
    relation_encoder_case("relenc_wide", 7, V=86, rel_dim=100, d=64, hid=64, R=37, Lmax=5)
    transformer_layer_case("tl_self", 8, d=32, ff=48, H=4, T=6, S=5, B=3, with_kv=False)
    transformer_layer_case("tl_kv", 9, d=32, ff=48, H=4, T=4, S=7, B=2, with_kv=True)
    generator_case("gen_small", 10, padded=False)
    generator_case("gen_padded", 11, padded=True)
    adam_case("adam_steps", 12)
