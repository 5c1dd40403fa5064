"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors generated from the reference
and against the pinned CPU oracle on the same seeded inputs.  Tolerances: 1e-3 fp32, 1e-2 bf16 (BASELINE.json)."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, sub, T, opt_mask
from tests_support import SMALL_VOCAB, SMALL_GEN_ARGS

pytestmark = pytest.mark.gpu

FP32 = dict(rtol=1e-3, atol=1e-3)
GRAD = dict(rtol=2e-3, atol=1e-3)
BF16_ATTN_ABS = 1e-3      # attention weights (probabilities), absolute; measured 1.2e-4 / 1.2e-4 / 1.6e-4 on gt_pad / gt_hd64 / gt_h8
BF16_OUT_REL = 4e-3       # post-LN outputs, relative to max(1, |gold|) (north_star: 1e-2).  Round 4, fp32 residual stream (ops.FP32_STREAM):
#                           measured 1.1e-3 / 1.8e-3 / 1.5e-3 on gt_pad / gt_hd64 / gt_h8 (absolute 1.1e-3 / 1.9e-3 / 1.6e-3 on outputs up to |4.8|);
#                           with the bf16 stream of rounds 1-3 it was 8.0e-3 / 1.33e-2 / 9.6e-3


def measured(name, got, want):
    """Print the measured maxima behind a bf16 bar (pytest -s shows them; profiles/README.md lists them per round)."""
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    e = (got - want).abs()
    print("MEASURED %s: max abs %.3e, max relative to max(1,|want|) %.3e, |want| max %.3g" % (
        name, float(e.max()), float((e / want.abs().clamp_min(1.0)).max()), float(want.abs().max())))


def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda:0")


def to_dev(d):
    return {k: v.to(dev()) for k, v in d.items()}


def cmp_param_grads(module, g, tol=GRAD):
    want = sub(g, "grad/")
    got = {k: p.grad for k, p in module.named_parameters() if p.grad is not None}
    assert set(want) == set(got)
    for k in want:
        torch.testing.assert_close(got[k].cpu().float(), want[k], msg=lambda m, k=k: "%s: %s" % (k, m), **tol)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (130, 260, 100), (257, 129, 512), (64, 48, 1030)])
def test_gemm(dtype, ta, tb, M, N, K):
    from gtos_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    a_d, b_d = a.to(dev(), dtype), b.to(dev(), dtype)
    ref = (a_d.float().t() if ta else a_d.float()) @ (b_d.float().t() if tb else b_d.float())
    tol = dict(rtol=1e-4, atol=1e-3) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * math.sqrt(K))
    out = ops.gemm(a_d, b_d, trans_a=ta, trans_b=tb)
    torch.testing.assert_close(out.float(), ref, **tol)
    out = ops.gemm(a_d, b_d, trans_a=ta, trans_b=tb, bias=bias.to(dev()), relu=True)
    torch.testing.assert_close(out.float(), torch.relu(ref + bias.to(dev())), **tol)
    acc = torch.ones((M, N), device=dev(), dtype=torch.float32)
    ops.gemm(a_d, b_d, trans_a=ta, trans_b=tb, out=acc, accumulate=True, splitk=4)
    torch.testing.assert_close(acc, ref + 1, **tol)


@pytest.mark.parametrize("M,N,K", [(40003, 2056, 1096), (262144, 256, 1024), (70000, 1024, 2048), (70000, 1024, 2056)])
def test_gemm_256_macro_tile(M, N, K):
    """Shapes that take the 256x256 NT kernel (bf16 in/out, >= 1024 macro tiles): ragged M/N/K tails, fused bias+ReLU,
    bf16 accumulate; against fp32 matmul of the same bf16 operands."""
    from gtos_amd import ops
    torch.manual_seed(M % 97)
    a = (torch.randn(M, K, device=dev()) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev()) * 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev())
    want = a.float() @ b.float().t()
    tol = dict(rtol=2e-2, atol=0.01 * K ** 0.5)
    got = ops.gemm(a, b, trans_b=True)
    torch.testing.assert_close(got.float(), want, **tol)
    got = ops.gemm(a, b, trans_b=True, bias=bias, relu=True)
    torch.testing.assert_close(got.float(), torch.relu(want + bias), **tol)
    base = torch.randn(M, N, device=dev()).to(torch.bfloat16)
    out = base.clone()
    ops.gemm(a, b, trans_b=True, out=out, accumulate=True)
    torch.testing.assert_close(out.float(), want + base.float(), rtol=3e-2, atol=0.02 * K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(140037, 1000, 1056), (66000, 520, 1024), (70003, 264, 1088), (66000, 512, 1120), (70000, 1024, 4096),
                                   (8300, 4200, 1152), (33000, 1000, 1216)])
def test_gemm_pipelined_256_tile(M, N, K):
    """Forward-shaped products with K >= 1024, K % 32 == 0 and >= 512 macro tiles take the 8-phase half-tile kernel (gemm256q_nt_kernel, round 6):
    K/64 = 16, 17, 18, 19, 64 -- both parities of its two-k-tile loop --, K % 64 == 32 (K = 1056, 1120: the upper half of the last k tile comes
    from a block of zeros), any N including N > 2048.  Ragged M / N tails (rows past the end are re-read, never stored), strided A, fused bias +
    ReLU, bf16 accumulate; against fp32 matmul of the same bf16 operands."""
    from gtos_amd import ops
    torch.manual_seed(M % 89)
    wide = (torch.randn(M, K + 64, device=dev()) * 0.5).to(torch.bfloat16)
    a = wide[:, 64:]                                        # row stride K + 64
    b = (torch.randn(N, K, device=dev()) * 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev())
    want = a.float() @ b.float().t()
    tol = dict(rtol=2e-2, atol=0.01 * K ** 0.5)
    got = ops.gemm(a, b, trans_b=True)
    torch.testing.assert_close(got.float(), want, **tol)
    got = ops.gemm(a.contiguous(), b, trans_b=True, bias=bias, relu=True)
    torch.testing.assert_close(got.float(), torch.relu(want + bias), **tol)
    base = torch.randn(M, N, device=dev()).to(torch.bfloat16)
    out = base.clone()
    ops.gemm(a, b, trans_b=True, out=out, accumulate=True)
    torch.testing.assert_close(out.float(), want + base.float(), rtol=3e-2, atol=0.02 * K ** 0.5)
    # deterministic dropout epilogue: the same mask as the 128x128 kernel's (a function of seed and element index)
    small = ops.gemm(a[:300], b, trans_b=True, p_drop=0.3, seed=77)
    big = ops.gemm(a, b, trans_b=True, p_drop=0.3, seed=77)
    assert torch.equal(small == 0, big[:300] == 0)


def test_gemm_tn_batch_small_weight_gradients_vs_torch():
    """gtos_gemm_tn_batch (round 6): the weight and bias gradients of many small linear layers in one launch -- dW_j += dY_j^T X_j, db_j += column
    sums of dY_j -- against fp32 matmul of the same bf16 operands: the shapes of a C2 step's graph and decoder layers, ragged M / N (multiples of 8),
    K that is no multiple of the 32-row stage, strided operands (column blocks of a wider buffer), a job without a bias, targets that already hold
    values (the kernel accumulates), more jobs than one launch's table holds (48), and a second call on top (gradient accumulation)."""
    import ctypes
    from gtos_amd._lib import call, stream
    torch.manual_seed(5)
    shapes = [(6464, 1536, 512), (6464, 512, 512), (6464, 1024, 512), (6464, 512, 1024), (3200, 512, 512), (3200, 1024, 512), (3200, 264, 520),
              (1000, 8, 8), (77, 520, 72), (6464, 1536, 512)] + [(3200 - 8 * i, 512, 256 + 8 * i) for i in range(45)]
    jobs, keep = [], []
    for q, (K, M, N) in enumerate(shapes):
        wide = (torch.randn(K, M + 16, device=dev()) * 0.5).to(torch.bfloat16)
        dy = wide[:, 8:8 + M] if q % 3 == 0 else wide[:, :M].contiguous()
        x = (torch.randn(K, N, device=dev()) * 0.5).to(torch.bfloat16)
        base = torch.randn(M, N, device=dev())
        bias0 = torch.randn(M, device=dev()) if q % 4 != 1 else None
        jobs.append((dy, x, base.clone(), None if bias0 is None else bias0.clone(), base, bias0))
    n = len(jobs)
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    A = vp(*[j[0].data_ptr() for j in jobs]); B = vp(*[j[1].data_ptr() for j in jobs]); C = vp(*[j[2].data_ptr() for j in jobs])
    bias = vp(*[(j[3].data_ptr() if j[3] is not None else None) for j in jobs])
    lda = i64(*[j[0].stride(0) for j in jobs]); ldb = i64(*[j[1].stride(0) for j in jobs]); ldc = i64(*[j[2].stride(0) for j in jobs])
    M_ = i32(*[j[0].shape[1] for j in jobs]); N_ = i32(*[j[1].shape[1] for j in jobs]); K_ = i32(*[j[0].shape[0] for j in jobs])
    args = (n, ctypes.addressof(A), ctypes.addressof(lda), ctypes.addressof(M_), ctypes.addressof(B), ctypes.addressof(ldb), ctypes.addressof(N_),
            ctypes.addressof(K_), ctypes.addressof(C), ctypes.addressof(ldc), ctypes.addressof(bias), stream())
    for rep in (1, 2):
        call("gtos_gemm_tn_batch", *args)
        torch.cuda.synchronize()
        for q, (dy, x, out, b_out, base, bias0) in enumerate(jobs):
            want = base + rep * (dy.float().t() @ x.float())
            torch.testing.assert_close(out, want, rtol=2e-3, atol=2e-3 * dy.shape[0] ** 0.5, msg=lambda m: "job %d %s: %s" % (q, tuple(dy.shape), m))
            if b_out is not None:
                torch.testing.assert_close(b_out, bias0 + rep * dy.float().sum(0), rtol=2e-3, atol=2e-3 * dy.shape[0] ** 0.5)
    # a job the kernel cannot take launches nothing and says so
    bad = i32(*([7] + [8] * (n - 1)))
    with pytest.raises(Exception):
        call("gtos_gemm_tn_batch", n, ctypes.addressof(A), ctypes.addressof(lda), ctypes.addressof(bad), ctypes.addressof(B), ctypes.addressof(ldb),
             ctypes.addressof(N_), ctypes.addressof(K_), ctypes.addressof(C), ctypes.addressof(ldc), ctypes.addressof(bias), stream())


def test_small_weight_gradients_batched_equal_launched_in_place(monkeypatch):
    """ops.DW_BATCH: a model's small weight / bias gradients noted during backward and launched together at ops.join_side() are the gradients the
    per-layer split-K products + column sums give (fp32 accumulation in another order: 1e-5 relative)."""
    from gtos_amd import ops
    torch.manual_seed(3)
    lin = [torch.nn.Linear(512, 1536), torch.nn.Linear(1536, 512), torch.nn.Linear(512, 264)]
    x0 = (torch.randn(3200, 512) * 0.5)
    res = []
    for mode in (True, False):
        monkeypatch.setattr(ops, "DW_BATCH", mode)
        ws = [(l.weight.detach().clone().to(dev()).requires_grad_(), l.bias.detach().clone().to(dev()).requires_grad_()) for l in lin]
        flat = torch.zeros(sum(w.numel() + b.numel() for w, b in ws), device=dev())
        off = 0
        for w, b in ws:                                   # gradients land in views of a flat fp32 bucket, like FlatParams
            for t in (w, b):
                t.grad = flat[off:off + t.numel()].view(t.shape)
                off += t.numel()
        h = x0.to(dev()).to(torch.bfloat16)
        for w, b in ws:
            h = ops.linear(h, w, b)
        h.float().square().sum().backward()
        ops.join_side()
        torch.cuda.synchronize()
        assert float(flat.abs().max()) > 0
        res.append(flat.clone())
    torch.testing.assert_close(res[0], res[1], rtol=1e-4, atol=1e-2)
    assert not ops._DW_PENDING


@pytest.mark.parametrize("M,N,K,sk", [(768, 512, 300040, 43), (256, 256, 100000, 64), (1024, 512, 6464, 12), (512, 264, 70008, 16)])
def test_gemm_weight_gradient_long_k_splitk(M, N, K, sk):
    """dW = A^T B with K in the hundreds of thousands: transpose-read operand path + split-K partial tiles reduced
    into the fp32 accumulator, contiguous and strided operands (the GRU backward passes column blocks of d4)."""
    from gtos_amd import ops
    torch.manual_seed(K % 89)
    a = (torch.randn(K, M, device=dev()) * 0.5).to(torch.bfloat16)
    b = (torch.randn(K, N, device=dev()) * 0.5).to(torch.bfloat16)
    base = torch.randn(M, N, device=dev())
    out = base.clone()
    ops.gemm(a, b, trans_a=True, out=out, accumulate=True, splitk=sk)
    want = base + a.float().t() @ b.float()
    torch.testing.assert_close(out, want, rtol=2e-3, atol=2e-3 * K ** 0.5)
    # strided operands (column blocks of a wider matrix, as the GRU backward passes them)
    wide = (torch.randn(K, M + 256, device=dev()) * 0.5).to(torch.bfloat16)
    out2 = torch.zeros(M, N, device=dev())
    ops.gemm(wide[:, 256:], b, trans_a=True, out=out2, accumulate=True, splitk=sk)
    torch.testing.assert_close(out2, wide[:, 256:].float().t() @ b.float(), rtol=2e-3, atol=2e-3 * K ** 0.5)


def test_gemm_strided_views_and_dropout():
    from gtos_amd import ops
    x = torch.randn(50, 96, device=dev())
    w = torch.randn(64, 32, device=dev())
    out = ops.gemm(x[:, 32:64], w, trans_b=True)          # lda = 96
    torch.testing.assert_close(out, x[:, 32:64] @ w.t(), rtol=1e-4, atol=1e-3)
    big = torch.randn(512, 256, device=dev())
    wb = torch.randn(256, 256, device=dev())
    y0 = ops.gemm(big, wb, trans_b=True)
    y1 = ops.gemm(big, wb, trans_b=True, p_drop=0.25, seed=123)
    y2 = ops.gemm(big, wb, trans_b=True, p_drop=0.25, seed=123)
    assert torch.equal(y1, y2)
    kept = y1 != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    torch.testing.assert_close(y1[kept], (y0 / 0.75)[kept], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ graph encoder
def run_graph_transformer(g, dtype, factored_from=None):
    from gtos_amd.graph_transformer import GraphTransformer, set_compute_dtype
    L, d, ff, H, n, B = [int(v) for v in g["cfg"]]
    m = GraphTransformer(L, d, ff, H, 0.0).to(dev())
    m.load_state_dict(sub(g, "sd/"))
    set_compute_dtype(m, dtype)
    x = T(g["x"]).to(dev()).requires_grad_()
    rel = T(g["relation"]).to(dev()).requires_grad_()
    pad, am = opt_mask(g["pad"]), opt_mask(g["attn_mask"])
    pad = pad.to(dev()) if pad is not None else None
    am = am.to(dev()) if am is not None else None
    out = m(x, rel, self_padding_mask=pad, self_attn_mask=am)
    with torch.no_grad():
        attn = m.get_attn_weights(x, rel, self_padding_mask=pad, self_attn_mask=am)
    (out.float() * T(g["wout"]).to(dev())).sum().backward()
    return m, out, attn, x.grad, rel.grad


@pytest.mark.parametrize("name", ["gt_tiny", "gt_pad", "gt_mask", "gt_hd64", "gt_h8"])
def test_graph_transformer_fp32_vs_golden(name):
    g = load_golden(name)
    m, out, attn, dx, drel = run_graph_transformer(g, torch.float32)
    torch.testing.assert_close(out.cpu(), T(g["out"]), **FP32)
    torch.testing.assert_close(attn.cpu(), T(g["attn"]), **FP32)
    torch.testing.assert_close(dx.cpu(), T(g["dx"]), **GRAD)
    torch.testing.assert_close(drel.cpu(), T(g["drelation"]), **GRAD)
    cmp_param_grads(m, g)


@pytest.mark.parametrize("name", ["gt_pad", "gt_hd64", "gt_h8"])
def test_graph_transformer_bf16_vs_golden(name):
    g = load_golden(name)
    m, out, attn, dx, drel = run_graph_transformer(g, torch.bfloat16)
    gold, gattn = T(g["out"]), T(g["attn"])
    e_out = float((out.float().cpu() - gold).abs().max())
    e_rel = float(((out.float().cpu() - gold).abs() / gold.abs().clamp_min(1.0)).max())
    e_attn = float((attn.cpu() - gattn).abs().max())
    rel_err = float((dx.float().cpu() - T(g["dx"])).norm() / T(g["dx"]).norm())
    print("bf16 vs golden %s: max |out err| %.3e (|out| max %.2f, relative to max(1,|gold|) %.3e), max |attn err| %.3e, dx rel %.3e" % (
        name, e_out, float(gold.abs().max()), e_rel, e_attn, rel_err))
    # north_star: 1e-2 in bf16.  Attention weights are probabilities and sit an order of magnitude inside it.  Post-LN outputs reach
    # |3-5|, where ONE bf16 rounding of the stored output is already up to 1.6e-2 absolute, so the output bar is relative to
    # max(1, |gold|): 1e-2 holds on two of the three fixtures, the 4-layer hd=64 one measures 1.33e-2 (bar 1.5e-2).
    assert e_attn < BF16_ATTN_ABS, e_attn
    assert e_rel < BF16_OUT_REL, (e_out, e_rel)
    assert rel_err < 3e-2, rel_err


def make_factored_case(seed, n, B, d, R):
    g = torch.Generator().manual_seed(seed)
    bank = 0.5 * torch.randn(R, d, generator=g)
    idx = torch.randint(0, R, (n, n, B), generator=g)
    idx[:, :, 0] = torch.randint(0, 3, (n, n), generator=g)      # a few very frequent types -> multi-chunk path
    x = torch.randn(n, B, d, generator=g)
    pad = torch.zeros(n, B, dtype=torch.bool)
    pad[n - 2:, B - 1] = True
    return bank, idx, x, pad


@pytest.mark.parametrize("n,B,d,H,R", [(9, 3, 32, 4, 40), (40, 4, 128, 2, 700), (33, 8, 512, 8, 3000)])
def test_factored_relation_matches_oracle(n, B, d, H, R):
    """FactoredRelation(bank, idx) must equal the reference's dense index_select path, forward and backward."""
    from gtos_amd.graph_transformer import GraphTransformer
    from gtos_amd.ops import FactoredRelation
    from oracle import gtos_oracle as O
    bank, idx, x, pad = make_factored_case(n * 100 + d, n, B, d, R)
    torch.manual_seed(5)
    ref = O.GraphTransformer(2, d, 2 * d, H, 0.0)
    for p in ref.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    m = GraphTransformer(2, d, 2 * d, H, 0.0).to(dev())
    m.load_state_dict(ref.state_dict())
    wout = torch.randn(n, B, d)
    bank_r = bank.clone().requires_grad_()
    x_r = x.clone().requires_grad_()
    out_r = ref(x_r, O.relation_lookup_train(bank_r, idx), self_padding_mask=pad)
    (out_r * wout).sum().backward()
    bank_d = bank.to(dev()).requires_grad_()
    x_d = x.to(dev()).requires_grad_()
    out_d = m(x_d, FactoredRelation(bank_d, idx.to(dev())), self_padding_mask=pad.to(dev()))
    (out_d * wout.to(dev())).sum().backward()
    torch.testing.assert_close(out_d.cpu(), out_r, **FP32)
    torch.testing.assert_close(x_d.grad.cpu(), x_r.grad, **GRAD)
    torch.testing.assert_close(bank_d.grad.cpu(), bank_r.grad, **GRAD)
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, msg=lambda s, k=k: "%s: %s" % (k, s), **GRAD)


@pytest.mark.parametrize("n,B,d,H,R,npad", [(2, 1, 16, 1, 3, 0), (3, 1, 64, 8, 1, 1), (1100, 1, 32, 4, 50, 37), (5, 11, 32, 4, 7, 2)])
def test_graph_transformer_edge_shapes(n, B, d, H, R, npad):
    """Smallest graphs (<CLS> + one node), one relation type for every pair, head_dim 8 (one lane per head), B not a
    multiple of 8 (no XCD graph map), and more keys than the LDS mask buffer holds (1100 > 1024: masks read from global
    memory) -- factored and dense operands against the pinned oracle, forward and backward."""
    from gtos_amd.graph_transformer import GraphTransformer
    from gtos_amd.ops import FactoredRelation
    from oracle import gtos_oracle as O
    g = torch.Generator().manual_seed(n * 7 + d)
    bank = 0.5 * torch.randn(R, d, generator=g)
    idx = torch.randint(0, R, (n, n, B), generator=g)
    x = torch.randn(n, B, d, generator=g)
    pad = torch.zeros(n, B, dtype=torch.bool)
    if npad:
        pad[n - npad:, B - 1] = True
    torch.manual_seed(11)
    ref = O.GraphTransformer(1, d, 2 * d, H, 0.0)
    m = GraphTransformer(1, d, 2 * d, H, 0.0).to(dev())
    m.load_state_dict(ref.state_dict())
    wout = torch.randn(n, B, d, generator=g)
    bank_r, x_r = bank.clone().requires_grad_(), x.clone().requires_grad_()
    out_r = ref(x_r, O.relation_lookup_train(bank_r, idx), self_padding_mask=pad)
    (out_r * wout).sum().backward()
    for factored in (True, False):
        bank_d, x_d = bank.to(dev()).requires_grad_(), x.to(dev()).requires_grad_()
        rel = FactoredRelation(bank_d, idx.to(dev())) if factored else bank_d.index_select(0, idx.to(dev()).reshape(-1)).view(n, n, B, d)
        out_d = m(x_d, rel, self_padding_mask=pad.to(dev()))
        (out_d * wout.to(dev())).sum().backward()
        torch.testing.assert_close(out_d.cpu(), out_r, **FP32)
        torch.testing.assert_close(x_d.grad.cpu(), x_r.grad, **GRAD)
        torch.testing.assert_close(bank_d.grad.cpu(), bank_r.grad, rtol=2e-3, atol=2e-3 if n > 1000 else 1e-3)
        m.zero_grad()


def test_relation_encoder_edge_shapes():
    """A single path of length 1; all paths of length 1; one long path among short ones -- against the pinned oracle."""
    from gtos_amd.encoder import RelationEncoder
    from oracle import gtos_oracle as O
    from oracle.gtos_oracle import VocabSpec
    cases = [torch.tensor([1]), torch.ones(17, dtype=torch.int64), torch.tensor([1, 8, 1, 2, 1, 1])]
    for lengths in cases:
        R, L = lengths.numel(), int(lengths.max())
        torch.manual_seed(3 + R)
        toks = torch.randint(1, 30, (L, R))
        for r in range(R):
            toks[int(lengths[r]):, r] = 0
        ref = O.RelationEncoder(VocabSpec(30, 0), 12, 32, 16, 2, 0.0)
        m = RelationEncoder(VocabSpec(30, 0), 12, 32, 16, 2, 0.0).to(dev())
        m.load_state_dict(ref.state_dict())
        out_r = ref(toks, lengths)
        out_r.square().sum().backward()
        out_d = m(toks.to(dev()), lengths.to(dev()))
        out_d.square().sum().backward()
        torch.testing.assert_close(out_d.cpu(), out_r, **FP32)
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            torch.testing.assert_close(p.grad.cpu(), q.grad, msg=lambda s_, k=k: "%s: %s" % (k, s_), **GRAD)


def test_gemm_degenerate_sizes():
    from gtos_amd import ops
    a = torch.randn(0, 16, device=dev())
    b = torch.randn(8, 16, device=dev())
    assert ops.gemm(a, b, trans_b=True).shape == (0, 8)                 # no rows: nothing launched
    a1 = torch.randn(1, 8, device=dev()).to(torch.bfloat16)
    b1 = torch.randn(1, 8, device=dev()).to(torch.bfloat16)
    torch.testing.assert_close(ops.gemm(a1, b1, trans_b=True).float(), a1.float() @ b1.float().t(), rtol=2e-2, atol=2e-2)


def test_relation_gather_mean_eval_lookup():
    from gtos_amd import ops
    from oracle import gtos_oracle as O
    g = torch.Generator().manual_seed(3)
    bank = torch.randn(50, 64, generator=g)
    idx = torch.randint(0, 50, (7, 7, 3, 4), generator=g)
    idx[..., 2:] = torch.where(torch.rand(7, 7, 3, 2, generator=g) < 0.5, 0, idx[..., 2:])
    idx[0, 0, 0] = 0
    got = ops.relation_gather_mean(bank.to(dev()), idx.to(dev()), zero_row0=True)
    torch.testing.assert_close(got.cpu(), O.relation_lookup_eval(bank, idx), rtol=1e-5, atol=1e-6)
    got = ops.relation_gather_mean(bank.to(dev()), idx[..., 0].contiguous().to(dev()), zero_row0=False)
    torch.testing.assert_close(got.cpu(), O.relation_lookup_train(bank, idx[..., 0]), rtol=0, atol=0)


# ------------------------------------------------------------------------------------------------ relation encoder
@pytest.mark.parametrize("name", ["relenc_small", "relenc_wide"])
def test_relation_encoder_vs_golden(name):
    from gtos_amd.encoder import RelationEncoder
    from oracle.gtos_oracle import VocabSpec
    g = load_golden(name)
    V, rel_dim, d, hid, R, Lmax = [int(v) for v in g["cfg"]]
    m = RelationEncoder(VocabSpec(V, 0), rel_dim, d, hid, 2, 0.0).to(dev())
    m.load_state_dict(sub(g, "sd/"))
    out = m(T(g["tokens"]).to(dev()), T(g["lengths"]).to(dev()))
    torch.testing.assert_close(out.cpu(), T(g["out"]), **FP32)
    (out * T(g["wout"]).to(dev())).sum().backward()
    cmp_param_grads(m, g)


def _rel_frob(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", ["packed", "x", "h", "off"])
def test_relation_encoder_bf16_fused_step_vs_golden(fuse, monkeypatch):
    """bf16 GRU: the packed-path evaluation (round 5's default: gtos_amd.gru.PackedPathGRUFn), round 4's BiGRUFinalFn on the fused MFMA
    step kernel (gtos_gru_step_fwd, input product fused or not) and its GEMM + cell path against the reference vectors; bf16 bar = 2e-3
    abs on outputs (north_star 1e-2; measured 8.1e-4), 3e-2 relative Frobenius on gradients."""
    from gtos_amd import encoder, gru
    from gtos_amd.encoder import RelationEncoder
    from oracle.gtos_oracle import VocabSpec
    monkeypatch.setattr(encoder, "PACKED", fuse == "packed")
    if fuse != "packed":
        monkeypatch.setattr(gru, "FUSE", fuse)
    g = load_golden("relenc_wide")
    V, rel_dim, d, hid, R, Lmax = [int(v) for v in g["cfg"]]
    m = RelationEncoder(VocabSpec(V, 0), rel_dim, d, hid, 2, 0.0).to(dev())
    m.load_state_dict(sub(g, "sd/"))
    m.compute_dtype = torch.bfloat16
    out = m(T(g["tokens"]).to(dev()), T(g["lengths"]).to(dev()))
    measured("relation_encoder bf16 fuse=%s vs golden" % fuse, out, T(g["out"]))
    torch.testing.assert_close(out.float().cpu(), T(g["out"]), rtol=0, atol=2e-3)          # measured 8.1e-4 on outputs up to |0.17|
    (out.float() * T(g["wout"]).to(dev())).sum().backward()
    want = sub(g, "grad/")
    for k, p in m.named_parameters():
        assert _rel_frob(p.grad.cpu(), want[k]) < 3e-2, k


@pytest.mark.gpu
def test_gru_fused_step_matches_unfused_many_tiles(monkeypatch):
    """hs=256 (the C2 width), ragged lengths, >1 row panel with a partial last tile, layer dropout on: the fused step
    kernels must reproduce the GEMM + cell path (same dropout counters) and the pinned oracle (p=0)."""
    from gtos_amd import gru, ops
    from oracle import gtos_oracle as O
    torch.manual_seed(5)
    R, L, hs, ind = 333, 6, 256, 104
    lengths = torch.randint(1, L + 1, (R,))
    lengths[0] = L
    sl, _ = torch.sort(lengths, descending=True, stable=True)
    bs = [int((sl > t).sum()) for t in range(L)]
    N = sum(bs)
    x32 = 0.5 * torch.randn(N, ind)
    ws32 = []
    for l in range(2):
        for _ in range(2):
            i = ind if l == 0 else 2 * hs
            ws32 += [0.08 * torch.randn(3 * hs, i), 0.08 * torch.randn(3 * hs, hs), 0.1 * torch.randn(3 * hs), 0.1 * torch.randn(3 * hs)]
    wout = torch.randn(R, 2 * hs)

    def run(fuse, p):
        monkeypatch.setattr(gru, "FUSE", fuse)
        ops.set_seed(77)
        x = x32.to(dev(), torch.bfloat16).requires_grad_()
        ws = [w.to(dev()).requires_grad_() for w in ws32]
        out = gru.bigru_final(x, bs, hs, 2, p, ws)
        (out.float() * wout.to(dev())).sum().backward()
        return out.float().cpu(), x.grad.float().cpu(), [w.grad.cpu() for w in ws]

    ref = run("off", 0.3)
    for fuse in ("h", "x"):
        got = run(fuse, 0.3)
        measured("gru fused %s vs unfused (p=0.3)" % fuse, got[0], ref[0])
        torch.testing.assert_close(got[0], ref[0], rtol=0, atol=1e-2)                     # measured 5.9e-3 on states up to |0.95|
        assert _rel_frob(got[1], ref[1]) < 3e-2
        for a, b in zip(got[2], ref[2]):
            assert _rel_frob(a, b) < 3e-2
    # p = 0 against the oracle GRU (fp32, CPU)
    xs = x32.clone().requires_grad_()
    wo = [w.clone().requires_grad_() for w in ws32]
    inp = xs
    fin = None
    offs = [0]
    for a in bs:
        offs.append(offs[-1] + a)
    for l in range(2):
        outs, fin = [], []
        for d_ in range(2):
            w_ih, w_hh, b_ih, b_hh = wo[l * 8 + d_ * 4: l * 8 + d_ * 4 + 4]
            h = torch.zeros(R, hs)
            ys = [None] * L
            for t in (range(L) if d_ == 0 else range(L - 1, -1, -1)):
                A = bs[t]
                hn = O.gru_cell(torch.nn.functional.linear(inp[offs[t]:offs[t] + A], w_ih, b_ih), h[:A], w_hh, b_hh)
                h = torch.cat([hn, h[A:]], 0)
                ys[t] = hn
            outs.append(torch.cat(ys, 0))
            fin.append(h)
        inp = torch.cat(outs, 1)
    want = torch.cat(fin, 1)
    (want * wout).sum().backward()
    for fuse in ("x", "h"):
        got = run(fuse, 0.0)
        measured("gru fused %s vs oracle (hs=256, outputs O(1))" % fuse, got[0], want)
        torch.testing.assert_close(got[0], want.detach(), rtol=0, atol=1e-2)              # measured 6.1e-3 / 6.8e-3 on states up to |0.93|
        assert _rel_frob(got[1], xs.grad) < 4e-2
        for a, b in zip(got[2], wo):
            assert _rel_frob(a, b.grad) < 4e-2


# ------------------------------------------------------------------------------------------------ decoder blocks
@pytest.mark.parametrize("name", ["tl_self", "tl_kv"])
def test_transformer_layer_vs_golden(name):
    from gtos_amd.transformer import TransformerLayer
    g = load_golden(name)
    d, ff, H, Tq, S, B, with_kv = [int(v) for v in g["cfg"]]
    m = TransformerLayer(d, ff, H, 0.0, with_external=True).to(dev())
    m.load_state_dict(sub(g, "sd/"))
    x = T(g["x"]).to(dev()).requires_grad_()
    ext = T(g["ext"]).to(dev()).requires_grad_()
    kv = T(g["kv"]).to(dev()).requires_grad_() if with_kv else None
    out, sw, ew = m(x, kv, T(g["self_pad"]).to(dev()), T(g["attn_mask"]).to(dev()), ext, T(g["ext_pad"]).to(dev()),
                    need_weights=True)
    torch.testing.assert_close(out.cpu(), T(g["out"]), **FP32)
    torch.testing.assert_close(sw.cpu(), T(g["self_w"]), **FP32)
    torch.testing.assert_close(ew.cpu(), T(g["ext_w"]), **FP32)
    (out * T(g["wout"]).to(dev())).sum().backward()
    torch.testing.assert_close(x.grad.cpu(), T(g["dx"]), **GRAD)
    torch.testing.assert_close(ext.grad.cpu(), T(g["dext"]), **GRAD)
    if with_kv:
        torch.testing.assert_close(kv.grad.cpu(), T(g["dkv"]), **GRAD)
    cmp_param_grads(m, g)


# ------------------------------------------------------------------------------------------------ whole model
def build_generator(g, factored):
    from gtos_amd.generator import Generator
    from oracle.gtos_oracle import VocabSpec
    d, ff, H, gl = [int(v) for v in g["cfg"]]
    vocabs = {k: VocabSpec(v, 0) for k, v in SMALL_VOCAB.items()}
    m = Generator(vocabs, *SMALL_GEN_ARGS, d, ff, H, 0.0, 1, gl, 2, None, dev(), factored_relation=factored).to(dev())
    m.load_state_dict(sub(g, "sd/"))
    return m


@pytest.mark.parametrize("name", ["gen_small", "gen_padded"])
@pytest.mark.parametrize("factored", [True, False])
def test_generator_vs_golden(name, factored):
    g = load_golden(name)
    m = build_generator(g, factored)
    batch, ebatch = to_dev(sub(g, "batch/")), to_dev(sub(g, "ebatch/"))
    m.train()
    graph, gmask, probe = m.encode_step(batch)
    torch.testing.assert_close(graph.cpu(), T(g["graph"]), **FP32)
    torch.testing.assert_close(probe.cpu(), T(g["probe"]), **FP32)
    assert torch.equal(gmask.cpu(), T(g["gmask"]))
    loss = m(batch)
    torch.testing.assert_close(loss.cpu(), T(g["loss"]), **FP32)
    loss.backward()
    cmp_param_grads(m, g)
    m.eval()
    with torch.no_grad():
        egraph, _, eprobe = m.encode_step(ebatch, train=False)
        eattn = m.encoder_attn(ebatch)
    torch.testing.assert_close(egraph.cpu(), T(g["egraph"]), **FP32)
    torch.testing.assert_close(eprobe.cpu(), T(g["eprobe"]), **FP32)
    torch.testing.assert_close(eattn.cpu(), T(g["eattn"]), **FP32)


def test_generator_bf16_loss_close():
    g = load_golden("gen_small")
    m = build_generator(g, True)
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    loss = m(to_dev(sub(g, "batch/")))
    assert abs(loss.item() - float(g["loss"])) < 1e-2 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    want = sub(g, "grad/")
    for k, p in m.named_parameters():
        if p.grad is not None and want[k].norm() > 1e-3:
            rel = (p.grad.cpu().float() - want[k]).norm() / want[k].norm()
            assert rel < 0.1, (k, rel)


# ------------------------------------------------------------------------------------------------ optimizer
def test_flat_adam_vs_golden():
    from gtos_amd.flat import FlatParams, inverse_sqrt_lr
    g = load_golden("adam_steps")
    warmup, d = [int(v) for v in g["cfg"]]

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight = torch.nn.Parameter(T(g["p0_init"]).clone())
            self.bias = torch.nn.Parameter(T(g["p1_init"]).clone())
    m = M().to(dev())
    flat = FlatParams(m)
    for step in range(1, 4):
        m.weight.grad.copy_(T(g["g0_%d" % step]))
        m.bias.grad.copy_(T(g["g1_%d" % step]))
        norm = flat.grad_norm()
        torch.testing.assert_close(norm.cpu().squeeze(), T(g["norm_%d" % step]), rtol=1e-5, atol=1e-6)
        lr = inverse_sqrt_lr(d, step, warmup)
        assert abs(lr - float(g["lrs"][step - 1])) < 1e-12
        flat.step(lr, gscale=1.0, max_norm=1.0)
        flat.zero_grad()
        # fp32 bar of BASELINE.json (1e-3); the kernel uses fast reciprocal/sqrt and FMA contraction
        torch.testing.assert_close(m.weight.detach().cpu(), T(g["p0_%d" % step]), rtol=1e-3, atol=2e-5)
        torch.testing.assert_close(m.bias.detach().cpu(), T(g["p1_%d" % step]), rtol=1e-3, atol=2e-5)


# ------------------------------------------------------------------------------------------------ dropout behaviour
def test_attention_dropout_statistics_and_determinism():
    from gtos_amd import ops
    n, B, d, H = 64, 8, 512, 8
    qkv = torch.randn(n, B, 3 * d, device=dev())
    ops.set_seed(77)
    o1, w1 = ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 0.125, p_drop=0.2, need_weights=True)
    ops.set_seed(77)
    o2, w2 = ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 0.125, p_drop=0.2, need_weights=True)
    assert torch.equal(o1, o2) and torch.equal(w1, w2)
    o0, w0 = ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 0.125, p_drop=0.0, need_weights=True)
    torch.testing.assert_close(w0.sum(1), torch.ones_like(w0.sum(1)), rtol=1e-4, atol=1e-4)
    kept = w1 > 0
    assert abs(kept.float().mean().item() - 0.8) < 0.01
    torch.testing.assert_close(w1[kept], (w0 / 0.8)[kept], rtol=1e-4, atol=1e-6)
    # o must be the dropped weights applied to v
    v = qkv[:, :, 2 * d:].reshape(n, B, H, d // H)
    torch.testing.assert_close(o1.view(n, B, H, d // H), torch.einsum("ijbh,jbhe->ibhe", w1, v), rtol=1e-3, atol=1e-3)


def test_training_mode_dropout_backward_consistent():
    """With dropout on, the analytic backward must match finite differences of the same (seeded) forward."""
    from gtos_amd.graph_transformer import GraphTransformerLayer
    from gtos_amd import ops
    torch.manual_seed(0)
    n, B, d = 6, 2, 32
    m = GraphTransformerLayer(d, 64, 4, 0.3).to(dev()).double().float()
    m.train()
    x = torch.randn(n, B, d, device=dev())
    rel = 0.5 * torch.randn(n, n, B, d, device=dev())
    wout = torch.randn(n, B, d, device=dev())

    def f(xx):
        ops.set_seed(1234)
        return (m(xx, rel)[0] * wout).sum()
    xg = x.clone().requires_grad_()
    f(xg).backward()
    eps = 1e-2
    for _ in range(5):
        dirn = torch.randn_like(x)
        dirn /= dirn.norm()
        num = (f(x + eps * dirn) - f(x - eps * dirn)).item() / (2 * eps)
        ana = (xg.grad * dirn).sum().item()
        assert abs(num - ana) < 5e-2 * max(1.0, abs(ana)), (num, ana)


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_c2_shape():
    """BASELINE config-2 shape (n=101, B=64, d=512, H=8), bf16: softmax rows sum to one, padded keys get zero
    weight, output is linear in V, factored == dense on identical inputs."""
    from gtos_amd import ops
    n, B, d, H = 101, 64, 512, 8
    g = torch.Generator().manual_seed(11)
    R = 5000
    qkv = torch.randn(n, B, 3 * d, generator=g).to(dev(), torch.bfloat16)
    bankp = (0.3 * torch.randn(R, 2 * d, generator=g)).to(dev(), torch.bfloat16)
    idx = torch.randint(0, R, (n, n, B), generator=g).to(dev())
    pad = torch.zeros(n, B, dtype=torch.bool)
    pad[90:, ::3] = True
    pad = pad.to(dev())
    fact = ops.FactoredRelation(torch.zeros(R, d, device=dev()), idx)
    o_f, w_f = ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 0.125, rel=bankp, fact=fact, key_pad=pad, need_weights=True)
    dense = bankp[idx.reshape(-1)].view(n, n, B, 2 * d)
    o_d, w_d = ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 0.125, rel=dense, key_pad=pad, need_weights=True)
    assert torch.equal(o_f, o_d) and torch.equal(w_f, w_d)
    torch.testing.assert_close(w_f.sum(1), torch.ones_like(w_f.sum(1)), rtol=1e-3, atol=1e-3)
    assert float(w_f[:, 90:, ::3].abs().max()) == 0.0
    qkv2 = qkv.clone()
    qkv2[:, :, 2 * d:] *= 2
    o2, _ = ops.attention_core(qkv2, None, (0, d, 2 * d), d, H, 0.125, rel=bankp, fact=fact, key_pad=pad)
    torch.testing.assert_close(o2.float(), 2 * o_f.float(), rtol=2e-2, atol=2e-2)


# ------------------------------------------------------------------------------------------------ BASELINE configs vs the oracle
def _full_model_pair(cfg_name, B, layers=None):
    """Product Generator on the GPU and the pinned oracle on the CPU with identical weights (train.sh dims, dropout 0)."""
    import copy
    from gtos_amd import synth
    from gtos_amd.config import default_vocabs, generator_args
    from gtos_amd.generator import Generator
    from oracle import gtos_oracle as O
    cfg = dict(synth.CONFIGS[cfg_name])
    if layers:
        cfg["layers"] = layers
    args = generator_args(cfg)
    args["dropout"] = 0.0
    depth = 256 if cfg["kind"] == "dep" else 32
    torch.manual_seed(11)
    ref = O.Generator({k: O.VocabSpec(v.size, 0) for k, v in default_vocabs().items()}, depth_size=depth, **args)
    for p in ref.parameters():
        if p.dim() == 1 or float(p.detach().abs().sum()) == 0:
            p.data.add_(0.02 * torch.randn_like(p))
    m = Generator(default_vocabs(), device=dev(), depth_size=depth, **args).to(dev())
    m.load_state_dict(ref.state_dict())
    batch, stats = synth.make_config_batch(cfg_name, B=B, padded=True)
    return ref, m, batch, stats


@pytest.mark.parametrize("cfg_name,B", [("C1", 8), ("C3", 3)])
def test_baseline_config_full_step_vs_oracle(cfg_name, B):
    """C1 (the reference's own CPU-runnable case, full size) and a translator-flavour C3 slice: loss and every parameter
    gradient of the full model against the oracle, fp32."""
    ref, m, batch, stats = _full_model_pair(cfg_name, B, layers=2 if cfg_name == "C3" else None)
    ref.train()
    m.train()
    loss_r = ref(batch)
    loss_r.backward()
    loss = m({k: v.to(dev()) for k, v in batch.items()})
    loss.backward()
    assert abs(loss.item() - loss_r.item()) < 1e-3 * max(1.0, abs(loss_r.item())), (loss.item(), loss_r.item())
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        err = (p.grad.cpu() - q.grad).abs().max().item()
        assert err < 1e-3 + 2e-3 * q.grad.abs().max().item(), (k, err, q.grad.abs().max().item())


def test_baseline_config_c1_bf16_loss():
    ref, m, batch, stats = _full_model_pair("C1", 8)
    m.set_compute_dtype(torch.bfloat16)
    ref.train()
    m.train()
    loss_r = ref(batch).item()
    loss = m({k: v.to(dev()) for k, v in batch.items()}).item()
    assert abs(loss - loss_r) < 1e-2 * max(1.0, abs(loss_r)), (loss, loss_r)


def test_trainer_two_steps_reduce_loss_and_keep_mirror_in_sync():
    """Flat-bucket trainer on C1 in bf16: loss is finite, parameters move, the bf16 mirror tracks the fp32 masters."""
    from gtos_amd import synth
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.train import Trainer
    m = build_generator(Generator, "C1", dev()).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    tr = Trainer(m, 256, warmup_steps=100, compute_dtype=torch.bfloat16)      # lr(step 8) = 5e-4
    batch, _ = synth.make_config_batch("C1")
    batch = {k: v.to(dev()) for k, v in batch.items()}
    before = tr.flat.param.clone()
    losses = [tr.step(batch) for _ in range(8)]
    assert all(l is not None and np.isfinite(l) for l in losses)
    assert losses[-1] < losses[0]
    assert float((tr.flat.param - before).abs().max()) > 0
    # the bf16 mirror and its transposed twin (one batched launch per optimizer step) track the fp32 masters
    n2 = 0
    for name, p, o in tr.flat.entries:
        assert torch.equal(p._gtos_mirror, p.detach().to(torch.bfloat16)), name
        if p.dim() == 2:
            assert torch.equal(p._gtos_mirror_t, p._gtos_mirror.t()), name
            n2 += 1
    assert n2 > 50
    from gtos_amd import ops
    lin = m.graph_encoder.layers[0].self_attn
    wbf = ops.compute_weight(lin.in_proj_weight, torch.bfloat16)
    d_ = lin.embed_dim
    assert ops.weight_t(lin.in_proj_weight, wbf).data_ptr() == lin.in_proj_weight._gtos_mirror_t.data_ptr()
    assert torch.equal(ops.weight_t(lin.in_proj_weight, wbf[d_:3 * d_], (d_, 3 * d_)), wbf[d_:3 * d_].t())
    torch.testing.assert_close(tr.flat.mirror.float(), tr.flat.param, rtol=1e-2, atol=1e-3)
    assert float(tr.flat.grad.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_side_stream_paths_give_the_same_gradients(dtype, monkeypatch):
    """The flat gradient bucket after one backward with the auxiliary-stream paths on (GRU weight gradients and the
    relation projections' backward GEMMs beside the main chain) equals the single-stream result: same kernels, same
    reduction order, only the stream differs -- and against the per-parameter oracle gradients in fp32."""
    from gtos_amd import synth, ops, gru
    from gtos_amd.config import build_generator
    from gtos_amd.flat import FlatParams
    from gtos_amd.generator import Generator
    batch, _ = synth.make_config_batch("C1")
    batch = {k: v.to(dev()) for k, v in batch.items()}

    def run(side):
        monkeypatch.setattr(ops, "BWD_SIDE", side)
        monkeypatch.setattr(gru, "SIDE_STREAM", side)
        monkeypatch.setattr(gru, "TRIE_SIDE", side)
        monkeypatch.setattr(ops, "BWD_SIDE_MIN_ROWS", 0)          # C1 is below the size thresholds: force the side paths
        monkeypatch.setattr(gru, "SIDE_MIN_ROWS", 0)
        m = build_generator(Generator, "C1", dev(), dropout=0.0).to(dev())
        m.set_compute_dtype(dtype)
        m.train()
        flat = FlatParams(m, mirror_dtype=dtype)
        loss = m(batch)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), flat.grad.clone(), m, flat

    l1, g1, m1, f1 = run(True)
    l0, g0, _, _ = run(False)
    assert abs(l1 - l0) < 1e-6
    assert float(g1.abs().max()) > 0
    # not bitwise: a few reductions use fp32 atomics (embedding scatter, bias partial sums, multi-chunk relation types)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-6 if dtype == torch.float32 else 1e-4)
    if dtype == torch.float32:
        from oracle import gtos_oracle as O
        from gtos_amd.config import generator_args
        cfg = synth.CONFIGS["C1"]
        torch.manual_seed(19940117)
        ref = O.Generator({k: O.VocabSpec(v, 0) for k, v in synth.DEFAULT_VOCAB.items()}, depth_size=32,
                          **dict(generator_args(cfg), dropout=0.0))
        ref.load_state_dict({k: v.detach().cpu() for k, v in m1.state_dict().items()})
        ref.train()
        cpu_batch = {k: v.cpu() for k, v in batch.items()}
        ref(cpu_batch).backward()
        for (k, p), (_, q) in zip(m1.named_parameters(), ref.named_parameters()):
            torch.testing.assert_close(p.grad.cpu(), q.grad, msg=lambda s_, k=k: "%s: %s" % (k, s_), **GRAD)


# ------------------------------------------------------------------------------------------------ round 2: C2 / C5 slices
_SLICE_CACHE = {}


def _oracle_slice(cfg_name, B):
    """Oracle loss + gradients of a full-depth slice of a BASELINE config (computed once per session: C5 takes ~1 min)."""
    key = (cfg_name, B)
    if key not in _SLICE_CACHE:
        ref, m, batch, stats = _full_model_pair(cfg_name, B)
        ref.train()
        loss_r = ref(batch)
        loss_r.backward()
        grads = {k: p.grad.clone() for k, p in ref.named_parameters()}
        _SLICE_CACHE[key] = (ref.state_dict(), batch, stats, float(loss_r), grads)
        del m
    return _SLICE_CACHE[key]


# bf16 bars on a full-depth (8-layer, d=512) model.  Loss: 1e-2 relative (north_star; measured 3e-5).  Gradients: the
# whole flat gradient within BF16_GRAD_GLOBAL relative Frobenius error of the fp32 oracle's, every parameter tensor within
# BF16_GRAD_REL.  The per-tensor bar is loose for a stated reason: the worst tensors (relation_in_proj.weight, the relation
# encoder's out_proj / GRU biases, the character embedding / convolution) are sums over 10^4..10^6 per-pair / per-row
# gradient rows that are each stored ONCE in bf16, and whose large parts cancel exactly in exact arithmetic
# (sum_j dS_ij = 0 makes the q_i . dS_ij part of sum_j d rb_ij vanish; what is left is |ra|/|q| ~ 1/30 of the terms at the
# reference's N(0, 0.02) initialisation).  Rounding each term to 2^-9 leaves an error of 2^-9 * |terms| * sqrt(count)
# against a true sum that is that much smaller than the terms -- 10-20 % here, with every kernel bit-exact in structure
# (the fp32 run of the same code agrees with the oracle to 4e-4).
# Measured in round 4 (fp32 residual stream; profiles/r4c_bf16_grad_table.tsv): global 2.67e-2 / 1.87e-2 / 2.04e-2 and worst tensor
# 0.169 / 0.227 / 0.184 (a relation_in_proj.weight each time, or the character embedding) on C2 B=3 / C5 B=1 / C3 B=3: the bars
# keep 1.5x / 1.1x of headroom over the largest value.
# Round 5, by tensor size (the same table): the tensors at 0.17-0.23 have gradient norms of 1e-5 .. 3e-5 against 0.5-1.3 for the largest
# ones -- five orders below what moves the model -- and the largest tensors are where they should be: decoder projections 0.5-0.7 %,
# concept_encoder.out_proj 2.6-3.8 %, the graph layers' fc1.weight (norm 0.15) 7.6 % at worst.  Hence two tiers: every tensor that
# carries at least 5 % of the largest tensor's norm within BF16_GRAD_REL_MAJOR = 0.1 (VERDICT round 4's figure; measured 0.076), the
# small ones (down to 1e-4 of the largest norm) within BF16_GRAD_REL = 0.25 (measured 0.227).  What the small tensors' error is NOT: a
# storage choice that fp32 rows would fix -- the fc1 / character-CNN tensors at 8-18 % go through no bank-gradient row at all; the bf16
# FUNCTION (activations rounded layer by layer, attention patterns shifted by 1e-3) has a different gradient than the fp32 one there.
BF16_GRAD_GLOBAL = 4e-2
BF16_GRAD_REL = 0.25
BF16_GRAD_REL_MAJOR = 0.1


def _product_on(cfg_name, sd, dtype):
    from gtos_amd import synth
    from gtos_amd.config import default_vocabs, generator_args
    from gtos_amd.generator import Generator
    cfg = dict(synth.CONFIGS[cfg_name])
    args = generator_args(cfg)
    args["dropout"] = 0.0
    m = Generator(default_vocabs(), device=dev(), depth_size=256 if cfg["kind"] == "dep" else 32, **args).to(dev())
    m.load_state_dict(sd)
    m.set_compute_dtype(dtype)
    m.train()
    return m


def _check_slice(cfg_name, B, dtype):
    sd, batch, stats, loss_r, grads = _oracle_slice(cfg_name, B)
    m = _product_on(cfg_name, sd, dtype)
    loss = m({k: v.to(dev()) for k, v in batch.items()})
    loss.backward()
    torch.cuda.synchronize()
    table = []
    for k, p in m.named_parameters():
        q = grads[k]
        table.append((k, _rel_frob(p.grad.cpu(), q), float(q.norm())))
    worst = sorted(table, key=lambda r: -r[1])[:8]
    print("%s B=%d %s: loss %.6f vs %.6f; worst relative gradient errors: %s" % (
        cfg_name, B, dtype, loss.item(), loss_r, ", ".join("%s %.3g" % (k, e) for k, e, _ in worst)))
    if dtype == torch.float32:
        assert abs(loss.item() - loss_r) < 1e-3 * max(1.0, abs(loss_r)), (loss.item(), loss_r)
        for k, p in m.named_parameters():
            q = grads[k]
            err = (p.grad.cpu() - q).abs().max().item()
            assert err < 1e-3 + 2e-3 * q.abs().max().item(), (k, err, q.abs().max().item())
    else:
        assert abs(loss.item() - loss_r) < 1e-2 * max(1.0, abs(loss_r)), (loss.item(), loss_r)
        num = sum(float((p.grad.cpu() - grads[k]).double().pow(2).sum()) for k, p in m.named_parameters())
        den = sum(float(grads[k].double().pow(2).sum()) for k, _ in m.named_parameters())
        glob = (num / den) ** 0.5
        print("global relative gradient error %.4f; tensors above 6e-2: %d of %d" % (
            glob, sum(e > 6e-2 for _, e, _ in table), len(table)))
        if os.environ.get("GTOS_GRAD_TABLE"):
            with open(os.environ["GTOS_GRAD_TABLE"], "a") as fo:
                for k, e, nrm in sorted(table, key=lambda r: -r[1]):
                    fo.write("%s B=%d\t%s\t%.4g\t%.4g\n" % (cfg_name, B, k, e, nrm))
                fo.write("%s B=%d\tGLOBAL\t%.4g\t%.4g\n" % (cfg_name, B, glob, den ** 0.5))
        assert glob < BF16_GRAD_GLOBAL, glob
        gmax = max(nrm for _, _, nrm in table)
        major = [(k, e, nrm) for k, e, nrm in table if nrm >= 0.05 * gmax]
        print("largest tensors (norm >= 5 %% of the largest, %d of them): worst %s" % (
            len(major), ", ".join("%s %.3g" % (k, e) for k, e, _ in sorted(major, key=lambda r: -r[1])[:3])))
        for k, e, nrm in table:
            if nrm >= 0.05 * gmax:
                assert e < BF16_GRAD_REL_MAJOR, (k, e, nrm)
            elif nrm > 1e-4 * gmax:                   # gradients that are numerically zero carry no relative information
                assert e < BF16_GRAD_REL, (k, e, nrm)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c2_slice_full_depth_vs_oracle(dtype):
    """BASELINE config C2 at full depth and width (n~100, L=8, d=512, H=8), 3 graphs: loss and EVERY parameter gradient
    against the pinned oracle, fp32 and bf16."""
    _check_slice("C2", 3, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c5_slice_full_depth_vs_oracle(dtype):
    """BASELINE config C5 (300-node dense graphs, n~300: the long-key case) at full depth, one graph."""
    _check_slice("C5", 1, dtype)


def test_c3_slice_bf16_translator_flavour():
    """Translator flavour (dependency trees, depth table 256, single path per pair) in bf16, full depth."""
    _check_slice("C3", 3, torch.bfloat16)


def _attention_fact_vs_dense(qkv, bankp, idx, pad, H, wout, p_drop=0.0, dense=True):
    """One relation-attention forward + backward on the factored and on the dense operand; returns (o, dqkv, d_bankp) of
    both.  The dense operand's bank gradient is the index_add of its per-pair gradient accumulated in fp32 (autograd's own
    index_select backward would add thousands of bf16 rows of a frequent type in bf16)."""
    from gtos_amd import ops
    n, B, d3 = qkv.shape
    d = d3 // 3
    res = []
    for factored in ((True, False) if dense else (True,)):
        q = qkv.clone().requires_grad_()
        if factored:
            rel = bankp.clone().requires_grad_()
            fact = ops.FactoredRelation(torch.zeros(bankp.shape[0], 8, device=qkv.device), idx)
        else:
            fact = None
            rel = bankp.index_select(0, idx.reshape(-1)).view(n, n, B, 2 * d).requires_grad_()
        ops.set_seed(99)
        o, _ = ops.attention_core(q, None, (0, d, 2 * d), d, H, (d // H) ** -0.5, rel=rel, fact=fact, key_pad=pad, p_drop=p_drop)
        (o.float() * wout).sum().backward()
        if factored:
            db = rel.grad.float()
        else:
            db = torch.zeros(bankp.shape, dtype=torch.float32, device=qkv.device)
            db.index_add_(0, idx.reshape(-1), rel.grad.reshape(-1, 2 * d).float())
        res.append((o.detach(), q.grad, db))
    return res


def test_c5_shape_attention_factored_equals_dense_fwd_bwd():
    """n=301 keys per query (C5), d=512, H=8, bf16, with key padding and weight dropout: the factored operand must give
    the dense operand's output and gradients (q/k/v bitwise-identical math; the bank gradient is the index_add of the
    dense per-pair gradient, accumulated in fp32 instead of through bf16 pair rows)."""
    n, B, d, H, R = 301, 4, 512, 8, 60000
    g = torch.Generator().manual_seed(301)
    qkv = torch.randn(n, B, 3 * d, generator=g).to(dev(), torch.bfloat16)
    bankp = (0.3 * torch.randn(R, 2 * d, generator=g)).to(dev(), torch.bfloat16)
    idx = torch.randint(0, R, (n, n, B), generator=g)
    idx[0, :, :] = 1
    idx[:, 0, :] = 0
    idx = idx.to(dev())
    pad = torch.zeros(n, B, dtype=torch.bool)
    pad[250:, 1] = True
    pad = pad.to(dev())
    wout = torch.randn(n, B, d, generator=g).to(dev())
    for p_drop in (0.0, 0.2):
        (o_f, dq_f, db_f), (o_d, dq_d, db_d) = _attention_fact_vs_dense(qkv, bankp, idx, pad, H, wout, p_drop)
        assert torch.equal(o_f, o_d)
        torch.testing.assert_close(dq_f.float(), dq_d.float(), rtol=2e-2, atol=2e-3)
        assert _rel_frob(db_f, db_d) < 1e-2
        assert float(o_f[:, 1].float().abs().max()) > 0


def test_c2_full_size_backward_properties_with_the_real_bank():
    """The full C2 batch (n=101, B=64, d=512, H=8, bf16) with the type ids of the REAL synthetic batch (R = 434,624 types,
    a handful of which -- <CLS>, <rCLS>, <SELF>, <TL> -- span thousands of chunks, the rest 1-2 pairs): the bank-gradient
    kernel's heavy/light split at scale.  Properties: factored == dense for the output and dq/dk/dv; d_bank equals the
    fp32 index_add of the dense operand's per-pair gradient; rows of types that never occur are exactly zero; the
    gradient is linear in the upstream gradient."""
    from gtos_amd import synth
    batch, stats = synth.make_config_batch("C2")
    idx = batch["relation"].to(dev())
    n, _, B = idx.shape
    d, H, R = 512, 8, stats["R"]
    assert (n, B, R) == (101, 64, 434624)
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(n, B, 3 * d, generator=g).to(dev(), torch.bfloat16)
    bankp = (0.3 * torch.randn(R + 7, 2 * d, generator=g)).to(dev(), torch.bfloat16)     # 7 trailing types never occur
    wout = torch.randn(n, B, d, generator=g).to(dev())
    (o_f, dq_f, db_f), (o_d, dq_d, db_d) = _attention_fact_vs_dense(qkv, bankp, idx, None, H, wout)
    assert torch.equal(o_f, o_d)
    torch.testing.assert_close(dq_f.float(), dq_d.float(), rtol=2e-2, atol=2e-3)
    assert _rel_frob(db_f, db_d) < 1e-2
    counts = torch.bincount(idx.reshape(-1), minlength=R + 7)
    assert float(db_f[counts == 0].float().abs().max()) == 0.0
    heavy = torch.nonzero(counts > 1000).flatten()
    assert heavy.numel() >= 3                                   # <CLS>, <rCLS>, <SELF> (and <TL>)
    assert _rel_frob(db_f[heavy], db_d[heavy]) < 1e-2
    _, dq2, db2 = _attention_fact_vs_dense(qkv, bankp, idx, None, H, 2 * wout, dense=False)[0]
    torch.testing.assert_close(db2.float(), 2 * db_f.float(), rtol=2e-2, atol=1e-3)


def test_relation_encoder_large_vocabulary_backward():
    """A relation vocabulary whose embedding table does not fit the LDS scatter buffer (V*dim*4 > 60 KB): real AMR role
    inventories with their `_reverse_` twins exceed the synthetic V=86.  Forward + every gradient against the oracle."""
    from gtos_amd.encoder import RelationEncoder
    from oracle import gtos_oracle as O
    V, rel_dim = 260, 100
    torch.manual_seed(8)
    R, L = 300, 5
    lengths = torch.randint(1, L + 1, (R,))
    toks = torch.randint(1, V, (L, R))
    for r in range(R):
        toks[int(lengths[r]):, r] = 0
    ref = O.RelationEncoder(O.VocabSpec(V, 0), rel_dim, 64, 32, 2, 0.0)
    m = RelationEncoder(O.VocabSpec(V, 0), rel_dim, 64, 32, 2, 0.0).to(dev())
    m.load_state_dict(ref.state_dict())
    out_r = ref(toks, lengths)
    out_r.square().sum().backward()
    out_d = m(toks.to(dev()), lengths.to(dev()))
    out_d.square().sum().backward()
    torch.testing.assert_close(out_d.cpu(), out_r, **FP32)
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, msg=lambda s_, k=k: "%s: %s" % (k, s_), **GRAD)


# ------------------------------------------------------------------------------------------------ trie-evaluated GRU
def _relenc_pair(bank, length, d=64, hid=64, rel_dim=20, V=90, seed=4):
    from gtos_amd.encoder import RelationEncoder
    from oracle import gtos_oracle as O
    torch.manual_seed(seed)
    ref = O.RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, 0.0)
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(3.0)                       # stronger recurrences than the default init: differences would show
    m = RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, 0.0).to(dev())
    m.load_state_dict(ref.state_dict())
    return ref, m


def _grads_of(m):
    return {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize("case", ["amr", "random", "single", "all_len1", "duplicates"])
def test_trie_gru_equals_flat_gru_and_oracle(case, monkeypatch):
    """RelationEncoder in bf16: the trie evaluation (layer 0 per prefix / suffix node, layer-1 input gates from per-node
    tables, segmented-sum backward) against the per-row evaluation of the same module (GTOS_GRU_TRIE=0) and against the
    pinned fp32 oracle, forward and every parameter gradient."""
    from gtos_amd import gru, synth
    from gtos_amd.pathtrie import build_path_trie
    if case == "amr":           # paths out of BFS trees: prefix-closed, heavy nodes near the root (multi-chunk reductions)
        batch, _ = synth.make_batch(3, 6, 40, 8)
        bank, length = batch["relation_bank"], batch["relation_length"]
    elif case == "random":      # no sharing at all beyond chance, ragged lengths, more labels
        g = torch.Generator().manual_seed(9)
        length = torch.randint(1, 9, (700,), generator=g)
        bank = torch.randint(1, 90, (8, 700), generator=g)
        for r in range(700):
            bank[int(length[r]):, r] = 0
    elif case == "all_len1":    # one trie level only: no parents, no children
        bank, length = torch.arange(1, 18).view(1, 17), torch.ones(17, dtype=torch.int64)
    elif case == "duplicates":  # the module API does not forbid repeated paths: they share every node, rows stay distinct
        base = torch.tensor([[5, 5, 9, 9, 9, 3], [6, 6, 2, 2, 2, 0], [7, 7, 0, 0, 0, 0]])
        bank, length = base, torch.tensor([3, 3, 2, 2, 2, 1])
    else:
        bank, length = torch.tensor([[7]]), torch.tensor([1])
    ref, m = _relenc_pair(bank, length)
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1))
    out_r = ref(bank, length)
    (out_r * wout).sum().backward()
    m.compute_dtype = torch.bfloat16
    res = {}
    for trie_on in (True, False):
        monkeypatch.setattr(gru, "TRIE", trie_on)
        m.zero_grad()
        trie = build_path_trie(bank, length).to(dev()) if trie_on else None
        out = m(bank.to(dev()), length.to(dev()), trie=trie)
        (out.float() * wout.to(dev())).sum().backward()
        res[trie_on] = (out.detach().float().cpu(), _grads_of(m))
    torch.testing.assert_close(res[True][0], res[False][0], rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(res[True][0], out_r.detach(), rtol=2e-2, atol=2e-2)
    for k, q in ref.named_parameters():
        e_flat = _rel_frob(res[False][1][k], q.grad)
        e_trie = _rel_frob(res[True][1][k], q.grad)
        assert e_trie < max(3e-2, 1.5 * e_flat), (k, e_trie, e_flat)


def test_trie_gru_dropout_is_deterministic_and_backward_matches_forward():
    """Training mode (embedding + inter-layer dropout per trie node): same seed -> same output; a different seed -> a
    different one; and the analytic gradient is the gradient of THAT forward: the directional derivative along the
    normalised gradient, measured by central differences of the seeded forward, equals the gradient norm."""
    from gtos_amd import ops, synth
    from gtos_amd.pathtrie import build_path_trie
    batch, _ = synth.make_batch(3, 4, 30, 8)
    bank, length = batch["relation_bank"], batch["relation_length"]
    ref, m = _relenc_pair(bank, length)
    m.compute_dtype = torch.bfloat16
    m.dropout = 0.3
    m.mask_sharing = "node"                                # the opt-in this test is about (default: masks per (path, position), one row each)
    m.train()
    trie = build_path_trie(bank, length).to(dev())
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1)).to(dev())
    bank_d, len_d = bank.to(dev()), length.to(dev())

    def f(seed=1234):
        ops.set_seed(seed)
        return (m(bank_d, len_d, trie=trie).float() * wout).sum()
    a, b, c = f(), f(), f(99)
    assert float(a.detach()) == float(b.detach()) and float(a.detach()) != float(c.detach())
    m.zero_grad()
    f().backward()
    for name in ("rnn.weight_hh_l0", "rnn.weight_ih_l1_reverse", "rel_embed.weight"):
        p = dict(m.named_parameters())[name]
        g = p.grad.clone()
        dirn = g / g.norm()
        eps = 0.05 * float(p.detach().norm()) / 10
        with torch.no_grad():
            p.add_(eps * dirn)
            up = float(f())
            p.sub_(2 * eps * dirn)
            dn = float(f())
            p.add_(eps * dirn)
        num = (up - dn) / (2 * eps)
        assert abs(num - float(g.norm())) < 0.15 * float(g.norm()), (name, num, float(g.norm()))


def test_segment_sum_kernels():
    from gtos_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(3)
    N, W, ld = 5000, 768, 1024
    src = torch.randn(N, ld, generator=g).to(dev(), torch.bfloat16)
    # ranges
    cuts = torch.sort(torch.randint(0, N, (300,), generator=g))[0].tolist()
    ranges = []
    for a_, b_ in zip([0] + cuts, cuts + [N]):
        ranges += [a_, b_]
    ranges += [17, 17]                                     # an empty segment -> zeros
    nseg = len(ranges) // 2
    rt = torch.tensor(ranges, dtype=torch.int32, device=dev())
    dst = torch.full((nseg, W), 7.0, dtype=torch.bfloat16, device=dev())
    call("gtos_segment_sum_ranges", nseg, ptr(rt), src.data_ptr() + 2 * 128, ld, W, ptr(dst), W, stream())
    want = torch.stack([src[ranges[2 * s]:ranges[2 * s + 1], 128:128 + W].float().sum(0) for s in range(nseg)])
    torch.testing.assert_close(dst.float(), want, rtol=1e-2, atol=1e-2 * want.abs().max().item())
    assert float(dst[-1].float().abs().max()) == 0.0
    # row lists with heavy nodes (host trie of a real bank provides them)
    from gtos_amd import synth
    from gtos_amd.pathtrie import build_path_trie
    batch, _ = synth.make_batch(3, 6, 40, 8)
    trie = build_path_trie(batch["relation_bank"], batch["relation_length"], chunk=16).to(dev())
    assert trie.pf.n_heavy > 0
    from gtos_amd import gru as gru_mod
    from gtos_amd.gru import _seg_rows
    for side, row_node in ((trie.pf, trie.row_pf), (trie.sf, trie.row_sf)):
        d4 = torch.randn(trie.N, 1024, generator=g).to(dev(), torch.bfloat16)
        want = torch.zeros(side.n_nodes, 768, device=dev()).index_add_(0, row_node.long(), d4[:, :768].float())
        outs = []
        for stream_kernel in (True, False):                # a wave per range of rows (default) / a wave per chunk
            out = torch.empty(side.n_nodes, 768, dtype=torch.bfloat16, device=dev())
            keep, gru_mod.SEG_STREAM = gru_mod.SEG_STREAM, stream_kernel
            try:
                _seg_rows(side, d4, 768, out)
            finally:
                gru_mod.SEG_STREAM = keep
            torch.testing.assert_close(out.float(), want, rtol=1e-2, atol=1e-2 * want.abs().max().item())
            outs.append(out)
        light = torch.ones(side.n_nodes, dtype=torch.bool)
        light[side.heavy_node.cpu().long()] = False        # single-chunk nodes: the same fp32 sum in the same order, one rounding
        assert torch.equal(outs[0][light.to(dev())], outs[1][light.to(dev())])
        # two sources over the same row lists in one pass
        d4b = torch.randn(trie.N, 1024, generator=g).to(dev(), torch.bfloat16)
        out_a, out_b = torch.empty_like(out), torch.empty_like(out)
        _seg_rows(side, d4, 768, out_a, d4b, out_b)
        want_b = torch.zeros(side.n_nodes, 768, device=dev()).index_add_(0, row_node.long(), d4b[:, :768].float())
        torch.testing.assert_close(out_a.float(), want, rtol=1e-2, atol=1e-2 * want.abs().max().item())
        torch.testing.assert_close(out_b.float(), want_b, rtol=1e-2, atol=1e-2 * want_b.abs().max().item())


@pytest.mark.parametrize("W", [256, 512, 768, 1024, 1280, 1536])
@pytest.mark.parametrize("rows_per_wave", [8, 100, 256])
def test_segment_sum_stream_kernel_on_a_synthetic_csr(W, rows_per_wave):
    """gtos_segment_sum_stream against index_add: chunks without rows (in front, in the middle, at the end), one-row chunks, full
    64-row chunks, multi-chunk (heavy) nodes, wave ranges that start and end anywhere, every supported width."""
    import numpy as np
    from gtos_amd._lib import call, ptr, stream
    rng = np.random.RandomState(W + rows_per_wave)
    counts = [0, 0] + rng.choice([0, 1, 1, 1, 2, 3, 5, 17, 64, 150, 700], size=400).tolist() + [0, 3, 0, 0]
    chunk_node, chunk_start, chunk_cnt, chunk_slot, heavy_node = [], [], [], [], []
    pos = 0
    for u, n in enumerate(counts):
        nch = max(1, -(-n // 64))
        slot = -1
        if nch > 1:
            slot = len(heavy_node)
            heavy_node.append(u)
        for k in range(nch):
            chunk_node.append(u); chunk_start.append(pos + 64 * k); chunk_cnt.append(max(0, min(64, n - 64 * k))); chunk_slot.append(slot)
        pos += n
    total, n_nodes, n_src = pos, len(counts), pos + 37
    rows = rng.permutation(n_src)[:total].astype(np.int32)                 # a gather: every source row at most once
    n_waves = max(1, -(-total // rows_per_wave))
    off = np.searchsorted(np.asarray(chunk_start), np.arange(n_waves) * rows_per_wave, side="left")
    off[0] = 0
    wave_off = np.concatenate([off, [len(chunk_node)]]).astype(np.int32)
    t = lambda a_: torch.tensor(np.asarray(a_), dtype=torch.int32, device=dev())
    ld = W + 64
    src = torch.randn(n_src, ld, generator=torch.Generator().manual_seed(W)).to(dev(), torch.bfloat16)
    dst = torch.full((n_nodes, W), 3.0, dtype=torch.bfloat16, device=dev())
    heavy = torch.zeros(max(1, len(heavy_node)), W, dtype=torch.float32, device=dev())
    rows_t, cn, cs, cc, sl, wo, hn = t(rows), t(chunk_node), t(chunk_start), t(chunk_cnt), t(chunk_slot), t(wave_off), t(heavy_node)
    call("gtos_segment_sum_stream", len(chunk_node), total, ptr(rows_t), ptr(cn), ptr(cs), ptr(cc), ptr(sl), ptr(wo), n_waves,
         src.data_ptr() + 2 * 32, ld, W, ptr(dst), W, ptr(heavy), stream())
    call("gtos_segment_sum_finish", len(heavy_node), ptr(hn), ptr(heavy), W, ptr(dst), W, stream())
    node_of_pos = np.repeat(np.arange(n_nodes), counts)
    want = torch.zeros(n_nodes, W, device=dev()).index_add_(0, torch.tensor(node_of_pos, device=dev()), src[rows_t.long(), 32:32 + W].float())
    torch.testing.assert_close(dst.float(), want, rtol=1e-2, atol=1e-2 * want.abs().max().item())
    empty = torch.tensor([n == 0 for n in counts], device=dev())
    assert float(dst[empty].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ fused copy/generate mixture
def _mixture_reference(logits, div, align, cp_seq, tot):
    """The op sequence of the oracle's TokenGenerator (pinned to generator/decoder.py:40-63 by the golden vectors)."""
    T_, B, V = logits.shape
    gate = torch.softmax(div, -1)
    probs = gate[..., :1] * torch.softmax(logits, -1)
    if tot > V:
        probs = torch.cat([probs, probs.new_zeros(T_, B, tot - V)], -1)
    index = cp_seq.transpose(0, 1).reshape(1, B, -1).expand(T_, -1, -1)
    probs = probs.scatter_add(-1, index, (gate[..., 1:] * align).reshape(T_, B, -1))
    return torch.log(probs + 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T_,B,V,S", [(5, 3, 64, 7), (9, 4, 203, 12), (50, 8, 10000, 100)])
def test_fused_copy_mixture_nll_and_loglikelihood(T_, B, V, S, dtype):
    from gtos_amd import ops
    g = torch.Generator().manual_seed(T_ * 31 + V)
    logits = (2.0 * torch.randn(T_, B, V, generator=g)).to(dtype).float()       # both sides see the rounded values
    div = torch.randn(T_, B, 2, generator=g).to(dtype).float()
    align = torch.softmax(torch.randn(T_, B, S, generator=g), -1)
    cp_seq = torch.randint(3, V + 6, (S, B), generator=g)                        # some ids beyond the vocabulary (copy-only)
    cp_seq[1] = cp_seq[0]                                                        # several concepts sharing a copy id
    cp_seq[S - 1] = 0                                                            # padded source position
    tot = 1 + int(cp_seq.max())
    target = torch.randint(1, V, (T_, B), generator=g)
    target[0] = cp_seq[0]                                                        # targets reachable by copying
    target[1, 0] = int(cp_seq.max())
    target[T_ - 1] = 0                                                           # padding
    lr, dr, ar = logits.clone().requires_grad_(), div.clone().requires_grad_(), align.clone().requires_grad_()
    ll_r = _mixture_reference(lr, dr, ar, cp_seq, tot)
    nll_r = (-ll_r.gather(-1, target.unsqueeze(-1)).squeeze(-1)).masked_fill(target.eq(0), 0.)
    wt = torch.rand(T_, B, generator=g)
    (nll_r * wt).sum().backward()
    ld = logits.to(dev(), dtype).requires_grad_()
    dd = div.to(dev(), dtype).requires_grad_()
    ad = align.to(dev()).requires_grad_()
    nll = ops.copy_nll(ld, dd, ad, cp_seq.to(dev()), target.to(dev()), 0)
    (nll * wt.to(dev())).sum().backward()
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(nll.cpu(), nll_r.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ld.grad.float().cpu(), lr.grad, **tol)
    torch.testing.assert_close(dd.grad.float().cpu(), dr.grad, **tol)
    torch.testing.assert_close(ad.grad.cpu(), ar.grad, rtol=1e-4, atol=1e-5)
    ll = ops.copy_log_likelihood(ld.detach(), dd.detach(), ad.detach(), cp_seq.to(dev()), tot)
    torch.testing.assert_close(ll.cpu(), ll_r.detach(), rtol=1e-4, atol=1e-4)


def test_factored_relation_host_index_equals_device_index():
    """ops.FactoredRelation with the loader's host-built RelationIndex must give the results of the device-built index:
    same output, same dq/dk/dv, same bank gradient (chunk order and pair order inside a type differ, sums do not)."""
    from gtos_amd import ops, synth
    from gtos_amd.relindex import build_relation_index
    batch, st = synth.make_batch(4, 8, 40, 8)
    idx = batch["relation"].to(dev())
    n, _, B = idx.shape
    d, H, R = 256, 4, st["R"]
    g = torch.Generator().manual_seed(8)
    qkv = torch.randn(n, B, 3 * d, generator=g).to(dev(), torch.bfloat16)
    bankp = (0.3 * torch.randn(R, 2 * d, generator=g)).to(dev(), torch.bfloat16)
    wout = torch.randn(n, B, d, generator=g).to(dev())
    index = build_relation_index(batch["relation"], R).to(dev())
    assert index.n_heavy >= 3
    res = []
    for ix in (index, None):
        q = qkv.clone().requires_grad_()
        rel = bankp.clone().requires_grad_()
        fact = ops.FactoredRelation(torch.zeros(R, 8, device=dev()), idx, index=ix)
        o, _ = ops.attention_core(q, None, (0, d, 2 * d), d, H, (d // H) ** -0.5, rel=rel, fact=fact)
        (o.float() * wout).sum().backward()
        res.append((o.detach(), q.grad, rel.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][2].float(), res[1][2].float(), rtol=2e-2, atol=1e-3)


def test_attention_kernel_rejects_shapes_outside_the_boundary():
    """The C ABI returns an error code (no launch, no fallback) for shapes outside the documented boundary: a head width
    that is not a multiple of 8 (the kernels read 8 channels per lane) or above 512, d not a multiple of H."""
    from gtos_amd import ops
    from gtos_amd._lib import GtosHipError
    from gtos_amd.graph_transformer import RelationMultiheadAttention
    for d, H in ((36, 3), (40, 10), (1040, 1), (100, 8)):
        qkv = torch.randn(5, 2, 3 * d, device=dev())
        with pytest.raises(GtosHipError):
            ops.attention_core(qkv, None, (0, d, 2 * d), d, H, 1.0)
    with pytest.raises(GtosHipError):
        RelationMultiheadAttention(40, 10)                   # refused at construction


@pytest.mark.parametrize("n,B,d,H,R", [(7, 3, 48, 6, 30), (11, 2, 96, 4, 50), (9, 8, 640, 5, 60), (6, 2, 1024, 8, 40), (13, 5, 768, 12, 80),
                                       (5, 1, 24, 1, 9)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_transformer_any_head_geometry_vs_oracle(n, B, d, H, R, dtype):
    """Shapes off the power-of-two fast path -- the reference only asks d % H == 0 (graph_transformer.py:75): head widths
    24 / 40 / 96 (idle lanes inside a head), 6 / 5 / 12 heads (idle heads in a slice), d = 640 / 768 / 1024 (two head slices,
    blockIdx.y) -- factored and dense relation operands and the plain (mode 0) attention of the decoder blocks, forward and
    every gradient against the pinned oracle."""
    from gtos_amd.graph_transformer import GraphTransformer
    from gtos_amd.transformer import TransformerLayer
    from gtos_amd.ops import FactoredRelation
    from oracle import gtos_oracle as O
    bank, idx, x, pad = make_factored_case(n * 100 + d, n, B, d, R)
    torch.manual_seed(5)
    ref = O.GraphTransformer(2, d, 2 * d, H, 0.0)
    for p in ref.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    m = GraphTransformer(2, d, 2 * d, H, 0.0).to(dev())
    m.load_state_dict(ref.state_dict())
    if dtype == torch.bfloat16:
        from gtos_amd.graph_transformer import set_compute_dtype
        set_compute_dtype(m, dtype)
    wout = torch.randn(n, B, d)
    bank_r, x_r = bank.clone().requires_grad_(), x.clone().requires_grad_()
    out_r = ref(x_r, O.relation_lookup_train(bank_r, idx), self_padding_mask=pad)
    (out_r * wout).sum().backward()
    tol = FP32 if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    gtol = GRAD if dtype == torch.float32 else None
    for factored in (True, False):
        m.zero_grad()
        bank_d, x_d = bank.to(dev()).requires_grad_(), x.to(dev()).requires_grad_()
        rel = FactoredRelation(bank_d, idx.to(dev())) if factored else bank_d.index_select(0, idx.to(dev()).reshape(-1)).view(n, n, B, d)
        out_d = m(x_d, rel, self_padding_mask=pad.to(dev()))
        (out_d.float() * wout.to(dev())).sum().backward()
        torch.testing.assert_close(out_d.float().cpu(), out_r, **tol)
        if gtol:
            torch.testing.assert_close(x_d.grad.cpu(), x_r.grad, **gtol)
            torch.testing.assert_close(bank_d.grad.cpu(), bank_r.grad, **gtol)
            for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
                torch.testing.assert_close(p.grad.cpu(), q.grad, msg=lambda s_, k=k: "%s: %s" % (k, s_), **gtol)
        else:
            assert _rel_frob(x_d.grad.float().cpu(), x_r.grad) < 5e-2 and _rel_frob(bank_d.grad.float().cpu(), bank_r.grad) < 5e-2
    if dtype == torch.float32:
        # mode 0 with a causal mask + cross attention over the graph states (the decoder's blocks), returned head-max weights
        torch.manual_seed(6)
        rl = O.TransformerLayer(d, 2 * d, H, 0.0, with_external=True)
        tl = TransformerLayer(d, 2 * d, H, 0.0, with_external=True).to(dev())
        tl.load_state_dict(rl.state_dict())
        T_ = 6
        y = torch.randn(T_, B, d)
        causal = torch.ones(T_, T_, dtype=torch.bool).triu_(1)
        y_r, mem_r = y.clone().requires_grad_(), x.clone().requires_grad_()
        o_r, _, ext_r = rl(y_r, self_attn_mask=causal, external_memories=mem_r, external_padding_mask=pad, need_weights=True)
        (o_r.sum() + ext_r.sum()).backward()
        y_d, mem_d = y.to(dev()).requires_grad_(), x.to(dev()).requires_grad_()
        o_d, _, ext_d = tl(y_d, self_attn_mask=causal.to(dev()), external_memories=mem_d, external_padding_mask=pad.to(dev()),
                           need_weights=True)
        (o_d.sum() + ext_d.sum()).backward()
        torch.testing.assert_close(o_d.cpu(), o_r, **FP32)
        torch.testing.assert_close(ext_d.cpu(), ext_r, **FP32)
        torch.testing.assert_close(y_d.grad.cpu(), y_r.grad, **GRAD)
        torch.testing.assert_close(mem_d.grad.cpu(), mem_r.grad, **GRAD)


def test_batched_bank_gradient_equals_per_layer_products(monkeypatch):
    """The layers' d(rel) written side by side into one slab + ONE deep-K product for the bank gradient (ops.GradAccumGroup)
    against the per-layer products: same flat gradient bucket (bf16: the batched form rounds once instead of L times)."""
    from gtos_amd import synth, ops
    from gtos_amd.config import build_generator
    from gtos_amd.flat import FlatParams
    from gtos_amd.generator import Generator
    batch, _ = synth.make_config_batch("C1")
    batch = {k: v.to(dev()) for k, v in batch.items()}
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "BATCH_DX", on)
        m = build_generator(Generator, "C1", dev(), dropout=0.0).to(dev())
        m.set_compute_dtype(torch.bfloat16)
        m.train()
        flat = FlatParams(m, mirror_dtype=torch.bfloat16)
        loss = m(batch)
        loss.backward()
        ops.join_side()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), flat.grad.clone(), [n for n, _, _ in flat.entries], flat))
    assert res[0][0] == res[1][0]
    g1, g0 = res[0][1], res[1][1]
    assert float((g1 - g0).norm() / g0.norm()) < 5e-3
    # the relation encoder's parameters are the ones downstream of the bank gradient
    for n_, p_, off in res[0][3].entries:
        if n_.startswith("relation_encoder."):
            k = p_.numel()
            a_, b_ = g1[off:off + k], g0[off:off + k]
            assert float((a_ - b_).norm() / b_.norm().clamp_min(1e-12)) < 3e-2, n_


def test_bench_two_ranks_on_one_gpu_functional():
    """The N>1 path of bench.py end to end on the 1-GPU box: `python bench.py --gpus 2` self-launches two ranks, both on
    cuda:0, process group gloo (RCCL refuses two ranks on one device): sharded synthetic batches, the segment-wise async
    all-reduce launched from the backward boundary markers, the 1/W fold, max-over-ranks timing, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GTOS_ONE_DEVICE="1", GTOS_DIST_BACKEND="gloo", GTOS_BENCH_ALL_LEGS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--prewarm-seconds", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * d["config"]["B_per_gpu"] and d["scaling"] == "weak"
    assert d["value"] > 0 and np.isfinite(d["config"]["loss_last"])
    assert "allreduce_exposed_ms_per_step" in d["config"]
    # round 4: one command, one line, both scaling modes + what the collective layer saw + the secondary legs
    assert d["collectives"]["ranks_seen_by_allreduce"] == 2 and d["collectives"]["backend"] == "gloo"
    o = d["other_scaling"]
    assert o["scaling"] == "strong" and o["global_batch"] == d["config"]["B_per_gpu"] and o["value"] > 0
    assert "per (path, position)" in d["config"]["relation_gru"]        # the headline runs the reference's dropout semantics ...
    assert d["node_masks"]["relation_masks"] == "node" and d["node_masks"]["ms_per_step"] > 0      # ... and measures the opt-in beside it
    assert d["loader_in_loop"]["workers_per_gpu"] == 1 and d["loader_in_loop"]["ms_per_step"] > 0, d["loader_in_loop"]


def _bench_line(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_strong_scaling_two_ranks_on_one_gpu_functional():
    """`--scaling strong`: the batch is B graphs in TOTAL (generator/train.py:183 divides the batch budget by the world
    size), rank r holds graphs [r*B/N, (r+1)*B/N); per-rank timing and exposed all-reduce time are reported."""
    d = _bench_line(["--gpus", "2", "--config", "C1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--scaling", "strong",
                     "--prewarm-seconds", "1"],
                    dict(GTOS_ONE_DEVICE="1", GTOS_DIST_BACKEND="gloo"))
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and c["global_batch"] == 8 and c["B_per_gpu"] == 4
    assert len(c["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in c["per_rank_ms_per_step"])
    assert abs(d["ms_per_step"] - max(c["per_rank_ms_per_step"])) < 1e-3 and c["rank_spread_ms_per_step"] >= 0    # (rounded to 3 decimals)
    assert d["value"] > 0 and np.isfinite(c["loss_last"])


def test_bench_fresh_batches_loader_in_the_loop():
    """`--fresh-batches`: AMRLoader thunks -> Prefetcher workers (C++ relation batch, tries, index; upload on a copy stream)
    -> Trainer.step, a new batch every step."""
    d = _bench_line(["--config", "C1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--fresh-batches", "--workers", "2",
                     "--prewarm-seconds", "1"],
                    dict(GTOS_BENCH_NO_DETAIL="1"))
    ld = d["config"]["loader"]
    assert ld["workers"] == 2 and ld["host_assembly_s_per_batch"] > 0 and ld["consumer_wait_ms_per_step"] >= 0
    assert d["config"]["B_per_gpu"] == 8 and d["value"] > 0 and np.isfinite(d["config"]["loss_last"])
    assert "NEW loader-built batch" in d["config"]["workload"]


@pytest.mark.parametrize("rows", [300, 40000 + 77])
def test_gru_layer1_step_kernel_vs_torch(rows):
    """One forward step of the second GRU layer (gate tables gathered by node id) against plain torch fp32 on the same bf16
    operands: small `rows` runs the tile-per-workgroup kernel, large `rows` the persistent one (W_hh slice resident in LDS,
    decoupled software-pipelined waves); ragged tail, rows that finish (n_out < rows) and rows that continue."""
    from gtos_amd.gru import _step_fwd
    hs, nf, nb = 256, 5000, 7000
    g = torch.Generator().manual_seed(rows)
    bf = torch.bfloat16
    gf = (0.5 * torch.randn(nf, 3 * hs, generator=g)).to(dev(), bf)
    gb = (0.5 * torch.randn(nb, 3 * hs, generator=g)).to(dev(), bf)
    fi = torch.randint(0, nf, (rows,), generator=g).to(dev(), torch.int32)
    bi = torch.randint(0, nb, (rows,), generator=g).to(dev(), torch.int32)
    h = torch.randn(rows, hs, generator=g).to(dev(), bf)
    wh = (0.1 * torch.randn(3 * hs, hs, generator=g)).to(dev(), bf)
    b_ih = (0.1 * torch.randn(3 * hs, generator=g)).to(dev())
    b_hh = (0.1 * torch.randn(3 * hs, generator=g)).to(dev())
    n_out = rows - rows // 3
    h_out = torch.zeros(rows, hs, device=dev(), dtype=bf)
    h_fin = torch.zeros(rows, hs, device=dev(), dtype=bf)
    gates = torch.zeros(rows, 4 * hs, device=dev(), dtype=bf)
    _step_fwd(rows, hs, None, None, h, None, b_ih, wh, b_hh, h_out, n_out, h_fin, gates, None, 0, hs, 0.0, 0, 0,
              gf=gf, gf_idx=fi, gb=gb, gb_idx=bi)
    xg = gf.float()[fi.long()] + gb.float()[bi.long()] + b_ih
    hg = h.float() @ wh.float().t() + b_hh
    r = torch.sigmoid(xg[:, :hs] + hg[:, :hs])
    z = torch.sigmoid(xg[:, hs:2 * hs] + hg[:, hs:2 * hs])
    hn = hg[:, 2 * hs:]
    n = torch.tanh(xg[:, 2 * hs:] + r * hn)
    o = (1 - z) * n + z * h.float()
    want = torch.cat([r, z, n, hn], 1)
    torch.testing.assert_close(gates.float(), want, rtol=2e-2, atol=2e-2)
    got_h = torch.cat([h_out[:n_out], h_fin[n_out:]]).float()
    torch.testing.assert_close(got_h, o, rtol=2e-2, atol=2e-2)
    assert float(h_fin[:n_out].float().abs().max()) == 0.0 and float(h_out[n_out:].float().abs().max()) == 0.0


def test_graph_layer_dropout_backward_matches_forward_bf16_d512():
    """Training mode at the C2 width (d=512, H=8, ff=1024) in bf16 with all four dropout sites of a graph-encoder layer on
    (graph_transformer.py:57,62,64,155): the analytic input gradient is the gradient of THAT seeded forward -- its
    directional derivative along the normalised gradient, by central differences of the seeded forward, equals its norm."""
    from gtos_amd.graph_transformer import GraphTransformerLayer, set_compute_dtype
    from gtos_amd import ops
    torch.manual_seed(0)
    n, B, d = 24, 4, 512
    m = GraphTransformerLayer(d, 1024, 8, 0.3).to(dev())
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 2:
                p.mul_(2.5)                       # N(0, 0.05): attention and FFN actually bend the output
    set_compute_dtype(m, torch.bfloat16)
    m.train()
    x = torch.randn(n, B, d, device=dev())
    R = 300
    bank = (0.5 * torch.randn(R, d, device=dev())).requires_grad_()
    idx = torch.randint(0, R, (n, n, B), device=dev())
    wout = torch.randn(n, B, d, device=dev())

    def f(xx, seed=4321):
        ops.set_seed(seed)
        return (m(xx, ops.FactoredRelation(bank, idx))[0].float() * wout).sum()
    assert float(f(x).detach()) == float(f(x).detach()) and float(f(x).detach()) != float(f(x, 7).detach())
    xg = x.clone().requires_grad_()
    f(xg).backward()
    g = xg.grad.float()
    dirn = g / g.norm()
    eps = 0.05 * float(x.norm()) / 10
    num = (float(f(x + eps * dirn).detach()) - float(f(x - eps * dirn).detach())) / (2 * eps)
    assert abs(num - float(g.norm())) < 0.15 * float(g.norm()), (num, float(g.norm()))
    gb = bank.grad.float()
    dirb = gb / gb.norm()
    epsb = 0.05 * float(bank.detach().norm()) / 10
    with torch.no_grad():
        bank.add_(epsb * dirb)
        up = float(f(x).detach())
        bank.sub_(2 * epsb * dirb)
        dn = float(f(x).detach())
        bank.add_(epsb * dirb)
    numb = (up - dn) / (2 * epsb)
    assert abs(numb - float(gb.norm())) < 0.2 * float(gb.norm()), (numb, float(gb.norm()))


def test_prefetcher_uploads_batches_on_a_copy_stream():
    """data.Prefetcher(device=...): batches assembled on a loader thread arrive on the GPU (tensors, path tries, relation
    index) and a bf16 model step runs on them."""
    from gtos_amd import synth
    from gtos_amd.config import build_generator
    from gtos_amd.data import Prefetcher
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    m = build_generator(Generator, "C1", dev(), dropout=0.0).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    m.train()

    def job(k):
        return lambda: attach_relation_index(attach_path_trie(synth.make_config_batch("C1", rank=k)[0]))
    losses = []
    for batch in Prefetcher((job(k) for k in range(3)), depth=2, workers=2, device=dev()):
        assert batch["concept"].is_cuda and batch["relation_trie"].device.type == "cuda" and batch["relation_index"].device.type == "cuda"
        losses.append(float(m(batch).detach()))
    assert len(losses) == 3 and all(np.isfinite(l) for l in losses) and len(set(losses)) == 3


def test_trainer_step_on_rccl_single_rank():
    """tools/rccl_step_check.py in a subprocess: Trainer.step with its collectives forced on over a 1-rank "nccl" (RCCL) group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_step_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl step check ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_device_side_step_control_matches_the_reference_loop_bookkeeping():
    """gtos_step_control + gtos_adam_step_ctl against the oracle's restatement of generator/train.py:136-148: running mean,
    the 5x abnormal-loss rule after the warm-up, the lr schedule -- and a discarded batch leaves parameters, moments and the
    bf16 mirror untouched while the counters advance exactly like the reference's."""
    from gtos_amd.train import Trainer
    from oracle import gtos_oracle as O
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1)).to(dev())

    class Wrap(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, batch):
            x, scale = batch
            return (self.net(x).pow(2).mean() + 1.0) * scale
    m = Wrap(model)
    tr = Trainer(m, 64, warmup_steps=3, compute_dtype=torch.bfloat16, segment_of=None)
    ref = O.LoopCounters(64, 3)
    x = torch.randn(32, 16, device=dev())
    scales = [1.0, 1.1, 0.9, 1.0, 1.05, 40.0, 1.0, 30.0, 0.95]          # steps 5 and 7 exceed 5x the running mean after warm-up
    pending = []
    for k, sc in enumerate(scales):
        before = (tr.flat.param.clone(), tr.flat.m.clone(), tr.flat.v.clone(), tr.flat.mirror.clone())
        res = tr.step((x, torch.tensor(sc, device=dev())), sync=(k % 2 == 0))
        val = res if (k % 2 == 0) else res.value()
        # the oracle's view of the same step (the loss value is the device's own: this checks the bookkeeping, not the model)
        lossv = val if val is not None else None
        if lossv is None:
            assert k in (5, 7), k
            assert ref.abnormal(1e9)                                      # past the warm-up: a huge loss is abnormal for the oracle too
            ref.advance(0.0, True)
            for a, b in zip(before, (tr.flat.param, tr.flat.m, tr.flat.v, tr.flat.mirror)):
                assert torch.equal(a, b)                                  # nothing moved
        else:
            assert not ref.abnormal(lossv), (k, lossv)
            lr = ref.advance(lossv, False)
            assert abs(float(tr._ctl[0]) - lr) < 1e-6 * lr
            assert not torch.equal(before[0], tr.flat.param)
        assert tr.batches_acm == ref.batches_acm and tr.discarded == ref.discarded
        assert abs(tr.loss_acm - ref.loss_acm) < 1e-5 * max(1.0, abs(ref.loss_acm))
        assert float(tr.flat.grad.abs().max()) == 0.0
    assert tr.discarded == 2 and tr.batches_acm == 7 and tr.steps_issued == 9
    tr.set_counters(100, 250.0, 3)
    assert tr.batches_acm == 100 and tr.discarded == 3 and tr.steps_issued == 103


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_token_encoder_elementwise_kernels_vs_torch(dtype):
    """gtos_highway_*, gtos_max_relu_*, gtos_token_row_* against the ATen op sequences of generator/encoder.py:141-149,169-172,
    196-199 on the same operands (fp32: exact up to rounding; bf16: the kernel computes in fp32 from bf16 operands)."""
    import torch.nn.functional as F
    from gtos_amd import ops
    g = torch.Generator().manual_seed(3)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    # highway gate
    N, D = 77, 40
    y = torch.randn(N, 2 * D, generator=g).to(dev(), dtype).requires_grad_()
    x = torch.randn(N, D, generator=g).to(dev(), dtype).requires_grad_()
    w = torch.randn(N, D, generator=g).to(dev())
    out = ops.highway_gate(y, x)
    (out.float() * w).sum().backward()
    y2, x2 = y.detach().float().requires_grad_(), x.detach().float().requires_grad_()
    nx, gate = y2.chunk(2, -1)
    s = torch.sigmoid(gate)
    ref = s * x2 + (1 - s) * F.relu(nx)
    (ref * w).sum().backward()
    torch.testing.assert_close(out.float(), ref, **tol)
    torch.testing.assert_close(y.grad.float(), y2.grad, **tol)
    torch.testing.assert_close(x.grad.float(), x2.grad, **tol)
    # max over time + relu (ties: the first maximum, like torch.max)
    N, L, Fc = 33, 12, 24
    yy = torch.randn(N, L, Fc, generator=g)
    yy[0, 3] = yy[0, 7] = 5.0                                   # a tie
    yy[1] = -yy[1].abs()                                        # all negative: relu kills output and gradient
    yd = yy.to(dev(), dtype).requires_grad_()
    wo = torch.randn(N, Fc, generator=g).to(dev())
    o = ops.max_relu(yd)
    (o.float() * wo).sum().backward()
    y3 = yd.detach().float().requires_grad_()
    r = F.relu(y3.max(1)[0])
    (r * wo).sum().backward()
    torch.testing.assert_close(o.float(), r, **tol)
    torch.testing.assert_close(yd.grad.float(), y3.grad, **tol)
    # token row: cat + embedding + dropout + zero pad; p = 0 against torch, p > 0: mask statistics and fwd/bwd mask agreement
    N, Cc, Ct, V = 50, 16, 30, 23
    feat = torch.randn(N, Cc, generator=g).to(dev(), dtype).requires_grad_()
    table = torch.randn(V, Ct, generator=g).to(dev()).requires_grad_()
    tok = torch.randint(0, V, (N,), generator=g).to(dev())
    tok[:5] = 0                                                 # padding rows
    wt = torch.randn(N, 48, generator=g).to(dev())
    row = ops.token_row(feat, tok, table, 0.0, pad_idx=0)
    assert row.shape == (N, 48) and float(row[:, 46:].abs().max()) == 0.0
    (row.float() * wt).sum().backward()
    f2, t2 = feat.detach().float().requires_grad_(), table.detach().clone().requires_grad_()
    emb = F.embedding(tok, t2, padding_idx=0).to(dtype).float()
    ref = F.pad(torch.cat([f2, emb], -1), (0, 2))
    (ref * wt).sum().backward()
    torch.testing.assert_close(row.float(), ref, **tol)
    torch.testing.assert_close(feat.grad.float(), f2.grad, **tol)
    torch.testing.assert_close(table.grad, t2.grad, **tol)
    assert float(table.grad[0].abs().max()) == 0.0
    ops.set_seed(5)
    big = torch.ones(4000, Cc, device=dev(), dtype=dtype).requires_grad_()
    tb = torch.ones(V, Ct, device=dev()).requires_grad_()
    tk = torch.randint(1, V, (4000,), generator=g).to(dev())
    dr = ops.token_row(big, tk, tb, 0.25, pad_idx=None)
    kept = (dr[:, :46] != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.01 and abs(float(dr[:, :46].float().max()) - 1 / 0.75) < 1e-2
    dr.float().sum().backward()
    torch.testing.assert_close(big.grad.float(), dr[:, :Cc].detach().float(), rtol=1e-2, atol=1e-2)   # same mask backward


# ------------------------------------------------------------------------------------------------ RelationEncoder, TRAINING mode (dropout > 0)
def _hash_keep(seed, idx, p):
    """numpy restatement of csrc/common.h drop_keep(seed, idx, p): the counter-based mask every kernel regenerates."""
    M = np.uint64(0xFFFFFFFF)

    def mix32(h):
        h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x7feb352d)) & M
        h = h ^ (h >> np.uint64(15)); h = (h * np.uint64(0x846ca68b)) & M
        return h ^ (h >> np.uint64(16))
    idx = np.asarray(idx).astype(np.uint64)
    seed = np.uint64(seed)
    h = mix32((idx & M) ^ (seed & M))
    h = mix32((h + (idx >> np.uint64(32)) * np.uint64(0x9E3779B9) + (seed >> np.uint64(32))) & M)
    r = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy(r >= np.float32(p))


def _relenc_case(seed=3, R=150, L=6, V=40, hid=64):
    g = torch.Generator().manual_seed(seed)
    length = torch.randint(1, L + 1, (R,), generator=g)
    length[0] = L
    bank = torch.randint(6, V, (L, R), generator=g)
    bank[:, 1] = bank[:, 0]                                 # paths 0 and 1: the SAME label sequence (same length), path 2 shares a prefix
    length[1] = length[0]
    bank[:3, 2] = bank[:3, 0]
    for r in range(R):
        bank[int(length[r]):, r] = 0
    return bank, length


@pytest.mark.parametrize("dtype,with_trie,hid", [(torch.float32, False, 64), (torch.bfloat16, False, 64), (torch.bfloat16, True, 64),
                                                  (torch.bfloat16, True, 256)])
def test_relation_encoder_training_mode_reference_masks_vs_oracle(dtype, with_trie, hid):
    """The RelationEncoder's TRAINING-mode function with the reference's dropout semantics (masks per (path, position, channel) on the
    label embeddings and between the GRU layers, generator/encoder.py:91-92,105) -- the library default, ``mask_sharing="path"``: the HIP
    path under its counter-based masks against the pinned oracle given EXACTLY those masks (oracle.MASK_HOOK), outputs and every
    parameter gradient.  p = 0.3; fp32 1e-3, bf16 1e-2 of the output scale.  bf16 is the production evaluation (PackedPathGRUFn:
    labels padded to the 64-wide k tile, the sort order and step sizes taken from the batch's trie when it has one, input gradients
    inside the backward step launches, grouped weight gradients, one-hot embedding gradient); hid = 256 is the C2 width."""
    from gtos_amd import ops
    from gtos_amd.encoder import RelationEncoder
    from gtos_amd.pathtrie import build_path_trie
    from oracle import gtos_oracle as O
    bank, length = _relenc_case()
    L, R = bank.shape
    V, rel_dim, d, p = 40, 100, 64, 0.3
    dim_pad = rel_dim + ((-rel_dim) % 8 if dtype == torch.float32 else (-rel_dim) % 64)
    torch.manual_seed(5)
    ref = O.RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, p)
    with torch.no_grad():
        for n_, q in ref.named_parameters():
            if n_.startswith("rnn.weight"):
                q.mul_(2.0 if hid == 64 else 1.0)
    m = RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, p).to(dev())
    m.load_state_dict(ref.state_dict())
    m.compute_dtype = dtype
    assert m.mask_sharing == "path"                         # the default IS the reference's function
    m.train(); ref.train()
    # the seeds the product will draw: embedding dropout first, then the inter-layer dropout of layer 0
    ops.set_seed(99)
    s_e, s_y = ops.next_seed(), ops.next_seed()
    ops.set_seed(99)
    # packed row of (position t, path r): paths sorted by length (descending; the module's own stable sort, or the order the batch's
    # trie carries), packed time-major
    trie = build_path_trie(bank, length) if with_trie else None
    if with_trie:
        order, bs = trie.seq_order.long(), list(trie.batch_sizes)
    else:
        sl, order = torch.sort(length, descending=True, stable=True)
        bs = [int((sl > t).sum()) for t in range(L)]
    rank = torch.empty(R, dtype=torch.long); rank[order] = torch.arange(R)
    offs = np.concatenate([[0], np.cumsum(bs)])
    row = torch.from_numpy(offs[:L]).view(L, 1) + rank.view(1, R)                     # [L,R] (meaningless past a path's end: masked by lengths)

    def hook(tag, x):
        if tag == "relenc.embed":
            idx = row.unsqueeze(-1) * dim_pad + torch.arange(rel_dim)
            return _hash_keep(s_e, idx.numpy(), p)
        if tag == "relenc.layer0":
            idx = row.unsqueeze(-1) * (2 * hid) + torch.arange(2 * hid)
            return _hash_keep(s_y, idx.numpy(), p)
        return None
    wout = torch.randn(R, d, generator=torch.Generator().manual_seed(1))
    O.MASK_HOOK = hook
    try:
        want = ref(bank, length)
        (want * wout).sum().backward()
    finally:
        O.MASK_HOOK = None
    out = m(bank.to(dev()), length.to(dev()), trie=trie.to(dev()) if with_trie else None)
    (out.float() * wout.to(dev())).sum().backward()
    ops.join_side()
    torch.cuda.synchronize()
    measured("relation_encoder TRAIN p=0.3 reference masks %s hid=%d trie=%s vs oracle" % (dtype, hid, with_trie), out, want)
    scale = float(want.abs().max())
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    assert float((out.float().cpu() - want).abs().max()) < tol * max(scale, 0.1), (float((out.float().cpu() - want).abs().max()), scale)
    wg = dict(ref.named_parameters())
    for k, q in m.named_parameters():
        e = _rel_frob(q.grad.cpu(), wg[k].grad)
        assert e < (2e-3 if dtype == torch.float32 else 4e-2), (k, e)
    # and the masks really are per (path, position): paths 0 and 1 are the same label sequence, their vectors differ under dropout
    assert float((out[0] - out[1]).abs().max()) > 1e-3


def test_relation_encoder_training_mode_trie_shared_masks_vs_oracle():
    """The opt-in ``mask_sharing="node"`` (masks per trie node, shared by the paths through it; separate embedding masks for the two
    directions): the trie evaluation under its counter-based masks against the oracle given the SAME masks expanded to (path, position) --
    so the opt-in is a fully specified function too, not only a statistical claim.  bf16, p = 0.3."""
    from gtos_amd import ops
    from gtos_amd.encoder import RelationEncoder
    from gtos_amd.pathtrie import build_path_trie
    from oracle import gtos_oracle as O
    bank, length = _relenc_case()
    L, R = bank.shape
    V, rel_dim, d, hid, p = 40, 100, 64, 64, 0.3
    dim_pad = rel_dim + (-rel_dim) % 8
    torch.manual_seed(5)
    ref = O.RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, p)
    with torch.no_grad():
        for n_, q in ref.named_parameters():
            if n_.startswith("rnn.weight"):
                q.mul_(2.0)
    m = RelationEncoder(O.VocabSpec(V, 0), rel_dim, d, hid, 2, p).to(dev())
    m.load_state_dict(ref.state_dict())
    m.compute_dtype = torch.bfloat16
    m.mask_sharing = "node"
    m.train(); ref.train()
    trie = build_path_trie(bank, length)
    ops.set_seed(123)
    se_pf, sy_pf, se_sf, sy_sf = ops.next_seed(), ops.next_seed(), ops.next_seed(), ops.next_seed()
    ops.set_seed(123)
    bs = trie.batch_sizes
    offs = np.concatenate([[0], np.cumsum(bs)])
    rank = torch.empty(R, dtype=torch.long); rank[trie.seq_order.long()] = torch.arange(R)
    row = (torch.from_numpy(offs[:len(bs)]).view(-1, 1) + rank.view(1, R)).clamp(max=trie.N - 1)       # [L,R]; rows past a path's end are never used
    live = torch.arange(len(bs)).view(-1, 1) < length.view(1, R)
    pn = torch.where(live, trie.row_pf.long()[row], torch.zeros_like(row))
    sn = torch.where(live, trie.row_sf.long()[row], torch.zeros_like(row))

    def hook(tag, x):
        if tag == "relenc.embed":
            return _hash_keep(se_pf, (pn.unsqueeze(-1) * dim_pad + torch.arange(rel_dim)).numpy(), p)
        if tag == "relenc.embed.reverse":
            return _hash_keep(se_sf, (sn.unsqueeze(-1) * dim_pad + torch.arange(rel_dim)).numpy(), p)
        if tag == "relenc.layer0":
            kf = _hash_keep(sy_pf, (pn.unsqueeze(-1) * hid + torch.arange(hid)).numpy(), p)
            kb = _hash_keep(sy_sf, (sn.unsqueeze(-1) * hid + torch.arange(hid)).numpy(), p)
            return torch.cat([kf, kb], -1)
        return None
    wout = torch.randn(R, d, generator=torch.Generator().manual_seed(1))
    O.MASK_HOOK = hook
    try:
        want = ref(bank[:len(bs)], length)
        (want * wout).sum().backward()
    finally:
        O.MASK_HOOK = None
    out = m(bank.to(dev()), length.to(dev()), trie=trie.to(dev()))
    (out.float() * wout.to(dev())).sum().backward()
    measured("relation_encoder TRAIN p=0.3 trie-shared masks bf16 vs oracle", out, want)
    scale = float(want.abs().max())
    assert float((out.float().cpu() - want).abs().max()) < 1e-2 * max(scale, 0.1)
    wg = dict(ref.named_parameters())
    for k, q in m.named_parameters():
        e = _rel_frob(q.grad.cpu(), wg[k].grad)
        assert e < 4e-2, (k, e)
    # the sharing itself: paths 0 and 1 are the same label sequence -> the same nodes -> the same masks -> the same vector
    assert float((out[0] - out[1]).abs().max()) == 0.0


def test_batched_relation_projection_weight_gradient_opt_in(monkeypatch):
    """GTOS_BATCH_DW (opt-in): the eight relation projections' weight gradients as ONE [L*2d, R] x [R, d] product over the
    gradient slab, launched on the auxiliary stream behind the bank's input gradient -- same gradients as the per-layer products.  A
    16-graph C2 batch (R > 100,000: the side-stream branch of LinearFn.backward), full model, bf16, dropout 0."""
    from gtos_amd import ops, synth
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    batch, stats = synth.make_config_batch("C2", B=16)
    assert stats["R"] > ops.BWD_SIDE_MIN_ROWS
    db = {k: v.to(dev()) for k, v in attach_relation_index(attach_path_trie(batch)).items()}
    m = build_generator(Generator, "C2", dev(), dropout=0.0).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(ops, "BATCH_DW", flag)
        m.zero_grad()
        loss = m(db)
        loss.backward()
        ops.join_side()
        torch.cuda.synchronize()
        res[flag] = (float(loss), {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
    assert res[True][0] == res[False][0]
    for k in res[False][1]:
        a, b = res[True][1][k], res[False][1][k]
        assert _rel_frob(a, b) < (2e-3 if "relation_in_proj" in k else 1e-2), (k, _rel_frob(a, b))     # (split-K order / atomics differ run to run)


def test_relation_projection_recomputed_in_backward_gives_the_same_gradients(monkeypatch):
    """GTOS_PROJ_RECOMPUTE=1 (opt-in memory plan, C5): the attention core drops a layer's projected bank after its forward and makes it
    again in its backward from the bank and the layer's weight.  Same loss, same gradients (the same GEMM on the same operands), with
    the projection prefetch out of the way so that every layer takes the recompute route."""
    from gtos_amd import ops, synth
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    m = build_generator(Generator, "C1", dev(), dropout=0.0).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    batch, _ = synth.make_config_batch("C1")
    attach_relation_index(attach_path_trie(batch))
    batch = {k: (v.to(dev()) if hasattr(v, "to") else v) for k, v in batch.items()}
    monkeypatch.setattr(ops, "PROJ_SIDE", False)

    def run(flag):
        monkeypatch.setattr(ops, "PROJ_RECOMPUTE", flag)
        m.zero_grad(set_to_none=True)
        loss = m(dict(batch))
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert l0 == l1 and set(g0) == set(g1) and len(g0) > 50
    worst = max(float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-6)) for n in g0)
    print("MEASURED projection recompute: worst relative gradient difference %.2e" % worst)
    assert worst <= 2e-3, worst            # fp32 atomic orders in the bank-gradient kernel differ run to run; everything else is bit-equal


# ------------------------------------------------------------------------------------------------ packed-path GRU (round 5)
@pytest.mark.parametrize("case", ["amr", "random", "single", "all_len1", "duplicates"])
@pytest.mark.parametrize("hid", [64, 256])
def test_packed_path_gru_equals_per_row_gru_and_oracle(case, hid, monkeypatch):
    """RelationEncoder in bf16 at dropout 0: round 5's packed-path evaluation (PackedPathGRUFn -- embedding straight off the bank through
    the sort order, fused steps, the layers' input gradients as role-B workgroups of the backward step launches, grouped weight-gradient
    products, one-hot embedding gradient; with the batch's trie as the source of the order, and with a bare bank) against round 4's
    BiGRUFinalFn behind sort / cat / index_select and against the pinned fp32 oracle: forward and every parameter gradient."""
    from gtos_amd import encoder, gru, ops, synth
    from gtos_amd.pathtrie import build_path_trie
    if case == "amr":
        batch, _ = synth.make_batch(3, 6, 40, 8)
        bank, length = batch["relation_bank"], batch["relation_length"]
    elif case == "random":
        g = torch.Generator().manual_seed(9)
        length = torch.randint(1, 9, (700,), generator=g)
        bank = torch.randint(1, 90, (8, 700), generator=g)
        for r in range(700):
            bank[int(length[r]):, r] = 0
    elif case == "all_len1":
        bank, length = torch.arange(1, 18).view(1, 17), torch.ones(17, dtype=torch.int64)
    elif case == "duplicates":
        base = torch.tensor([[5, 5, 9, 9, 9, 3], [6, 6, 2, 2, 2, 0], [7, 7, 0, 0, 0, 0]])
        bank, length = base, torch.tensor([3, 3, 2, 2, 2, 1])
    else:
        bank, length = torch.tensor([[7]]), torch.tensor([1])
    ref, m = _relenc_pair(bank, length, hid=hid)
    if hid == 256:
        with torch.no_grad():                 # (_relenc_pair triples the default init: at 256 channels that saturates every cell)
            rp = dict(ref.named_parameters())
            for k_, q_ in m.named_parameters():
                rp[k_].div_(3.0)
                q_.copy_(rp[k_])
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1))
    out_r = ref(bank, length)
    (out_r * wout).sum().backward()
    m.compute_dtype = torch.bfloat16
    monkeypatch.setattr(gru, "TRIE", False)                 # (at dropout 0 the trie evaluation would take over)
    res = {}
    for name in ("packed+trie", "packed", "per-row"):
        monkeypatch.setattr(encoder, "PACKED", name != "per-row")
        m.zero_grad()
        trie = build_path_trie(bank, length).to(dev()) if name == "packed+trie" else None
        out = m(bank.to(dev()), length.to(dev()), trie=trie)
        (out.float() * wout.to(dev())).sum().backward()
        ops.join_side()
        torch.cuda.synchronize()
        res[name] = (out.detach().float().cpu(), _grads_of(m))
    for name in ("packed+trie", "packed"):
        measured("packed-path GRU (%s, %s, hid %d) vs oracle" % (name, case, hid), res[name][0], out_r.detach())
        torch.testing.assert_close(res[name][0], res["per-row"][0], rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(res[name][0], out_r.detach(), rtol=2e-2, atol=2e-2)
        for k, q in ref.named_parameters():
            e_old, e_new = _rel_frob(res["per-row"][1][k], q.grad), _rel_frob(res[name][1][k], q.grad)
            assert e_new < max(3e-2, 1.5 * e_old), (name, k, e_new, e_old)
    # the order is the only thing the trie contributes: the two packed runs are the same function
    torch.testing.assert_close(res["packed+trie"][0], res["packed"][0], rtol=0, atol=2e-2)


@pytest.mark.parametrize("rows,hs,in_dim,in_valid,ldx", [(1000, 64, 128, 100, 128), (300, 128, 64, 64, 64), (5000, 256, 128, 100, 128),
                                                         (70001, 256, 512, 512, 512), (200003, 256, 512, 512, 640), (64, 64, 64, 8, 64)])
def test_gru_grouped_weight_gradients_vs_torch(rows, hs, in_dim, in_valid, ldx):
    """gtos_gru_weight_grads: dW_ih += d4[:, 0:3hs]^T x, dW_hh += d4[:, {0:2hs, 3hs:4hs}]^T h_prev as ONE grouped product over
    d4^T [x | h_prev] (256 x 256 tiles of the needed blocks only, split-K partial tiles reduced in a fixed order) against fp32 matmuls
    of the same bf16 operands; accumulates into what the targets held; deterministic."""
    from gtos_amd import ops
    from gtos_amd._lib import call, ptr, stream
    torch.manual_seed(rows % 97)
    d4 = (torch.randn(rows, 4 * hs, device=dev()) * 0.5).to(torch.bfloat16)
    xw = (torch.randn(rows, ldx, device=dev()) * 0.5).to(torch.bfloat16)
    x = xw[:, :in_dim]
    hp = (torch.randn(rows, hs, device=dev()) * 0.5).to(torch.bfloat16)
    base_ih, base_hh = torch.randn(3 * hs, in_valid, device=dev()), torch.randn(3 * hs, hs, device=dev())
    want_ih = base_ih + d4[:, :3 * hs].float().t() @ x[:, :in_valid].float()
    want_hh = base_hh + torch.cat([d4[:, :2 * hs], d4[:, 3 * hs:]], 1).float().t() @ hp.float()
    outs = []
    for _ in range(2):
        g_ih, g_hh = base_ih.clone(), base_hh.clone()
        ws = ops._workspace(dev())
        call("gtos_gru_weight_grads", rows, hs, in_dim, in_valid, ptr(d4), ptr(x), ldx, ptr(hp), hs, ptr(g_ih), in_valid, ptr(g_hh), hs,
             ptr(ws), ws.numel() * 4, stream())
        torch.cuda.synchronize()
        outs.append((g_ih, g_hh))
    tol = dict(rtol=2e-3, atol=2e-3 * rows ** 0.5)
    torch.testing.assert_close(outs[0][0], want_ih, **tol)
    torch.testing.assert_close(outs[0][1], want_hh, **tol)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows,rows_prev,hs,n_in", [(300, 200, 64, 128), (200, 300, 64, 64), (1000, 1000, 256, 512), (0, 777, 256, 512),
                                                    (40000, 50001, 256, 128)])
def test_gru_backward_step_input_gradient_role_vs_torch(rows, rows_prev, hs, n_in):
    """gtos_gru_step_bwd_fused: with ``dinp`` the launch also computes dinp = d4_prev[:, 0:3hs] x W_ih for the step processed just before
    (role B workgroups), written, accumulated, and masked with the layer-input dropout; the cell part (role A) is bit-identical to the
    launch without role B; rows == 0 runs role B alone."""
    from gtos_amd.gru import _step_bwd_fused, N_BIAS_PARTIALS
    torch.manual_seed(rows + rows_prev)
    bf = torch.bfloat16
    d4_prev = (torch.randn(rows_prev, 4 * hs, device=dev()) * 0.3).to(bf)
    w_ih = (torch.randn(3 * hs, n_in, device=dev()) * 0.1).to(bf)
    wi_t = w_ih.t().contiguous()
    wh_t = (torch.randn(hs, 3 * hs, device=dev()) * 0.1).to(bf)
    want = d4_prev[:, :3 * hs].float() @ w_ih.float()
    gates = torch.rand(max(rows, 1), 4 * hs, device=dev()).to(bf)
    hprev = torch.randn(max(rows, 1), hs, device=dev()).to(bf)
    res = {}
    for fused in (False, True):
        dh = torch.randn(max(rows, 1), hs, generator=torch.Generator(device=dev()).manual_seed(3), device=dev()).to(bf)
        d4 = torch.zeros(max(rows, 1), 4 * hs, device=dev(), dtype=bf)
        bpart = torch.zeros(N_BIAS_PARTIALS, 4 * hs, device=dev())
        wide = torch.full((rows_prev, n_in + 64), 7.0, device=dev(), dtype=bf)
        dinp = wide[:, :n_in]                                                                      # a column block: row stride n_in + 64
        kw = dict(wi_t=wi_t, dinp=dinp, n_in=n_in) if fused else {}
        if rows > 0 or fused:
            _step_bwd_fused(rows, hs, d4_prev, rows_prev, wh_t, gates if rows else None, hprev if rows else None, None, hs,
                            dh if rows else None, d4 if rows else None, 0.0, 0, 0, bpart if rows else None, **kw)
        torch.cuda.synchronize()
        res[fused] = (dh.clone(), d4.clone(), bpart.clone(), dinp, wide)
    for a_, b_ in zip(res[False][:2], res[True][:2]):
        assert torch.equal(a_, b_)
    # (the bias partials land in slot blockIdx % n_partials: another grid, other slots -- their column sums are the bias gradients)
    torch.testing.assert_close(res[False][2].sum(0), res[True][2].sum(0), rtol=1e-4, atol=1e-3)
    dinp = res[True][3]
    torch.testing.assert_close(dinp.float(), want, rtol=2e-2, atol=0.02 * (3 * hs) ** 0.5 * 0.03)
    assert bool((res[True][4][:, n_in:] == 7).all())                     # nothing beyond the block is touched
    # accumulate on top, then the masked form: dinp = mask * product, counter in_drop_base + m * n_in + column
    before = dinp.clone()
    _step_bwd_fused(0, hs, d4_prev, rows_prev, wh_t, None, None, None, hs, None, None, 0.0, 0, 0, None, wi_t=wi_t, dinp=dinp, n_in=n_in, dinp_acc=True)
    torch.testing.assert_close(dinp.float(), before.float() + want, rtol=3e-2, atol=0.02)
    masked = torch.empty(rows_prev, n_in, device=dev(), dtype=bf)
    _step_bwd_fused(0, hs, d4_prev, rows_prev, wh_t, None, None, None, hs, None, None, 0.0, 0, 0, None, wi_t=wi_t, dinp=masked, n_in=n_in,
                    p_in=0.25, seed_in=4242, in_drop_base=5 * n_in)
    torch.cuda.synchronize()
    idx = (torch.arange(rows_prev).view(-1, 1) + 5) * n_in + torch.arange(n_in).view(1, -1)
    keep = _hash_keep(4242, idx.numpy(), 0.25).to(dev())
    torch.testing.assert_close(masked.float(), torch.where(keep, want / 0.75, torch.zeros_like(want)), rtol=2e-2, atol=0.02)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_embed_packed_paths_equals_sort_pack_embed(p):
    """gtos_embed_packed_paths (the RelationEncoder's packed input straight off the bank through the sort order, no host read) == the
    reference's sort -> pack -> embed -> dropout sequence as round 4 ran it (torch.sort / cat + gtos_embed_rows_fwd): bit-identical
    rows under the same seed, plus the one-hot operand and the packed tokens."""
    from gtos_amd import ops
    from gtos_amd._lib import call, dt, ptr, stream
    from gtos_amd.gru import PackPlan
    bank, length = _relenc_case(R=1500, L=7, V=86)
    L, R = bank.shape
    table = torch.randn(86, 100, device=dev())
    plan = PackPlan.of_lengths(length.to(dev()), L)
    sl, order = torch.sort(length, descending=True, stable=True)
    toks = bank.index_select(1, order)
    packed = torch.cat([toks[t, :a] for t, a in enumerate(plan.batch_sizes)]).to(dev())
    assert plan.N == packed.numel() == int(length.sum())
    for dtype in (torch.bfloat16, torch.float32):
        ops.set_seed(11)
        want = ops.embed_rows(packed, table, 128, p, dtype)
        X = torch.empty(plan.N, 128, device=dev(), dtype=dtype)
        onehot = torch.empty(plan.N, 88, device=dev(), dtype=torch.bfloat16)
        tokens = torch.empty(plan.N, device=dev(), dtype=torch.int64)
        ops.set_seed(11)
        call("gtos_embed_packed_paths", dt(X), L, R, plan.N, ptr(bank.to(dev())), ptr(plan.order32), ptr(plan.offs_dev), ptr(table), 100, 128, ptr(X),
             float(p), ops.next_seed() if p > 0 else 0, ptr(onehot), 88, ptr(tokens), stream())
        torch.cuda.synchronize()
        assert torch.equal(X, want) and torch.equal(tokens, packed)
        assert torch.equal(onehot.float(), torch.nn.functional.one_hot(packed, 88).float())


_RING_CHILD = r'''
import sys, torch
from gtos_amd import gru, ops
torch.manual_seed(7)
dev = torch.device("cuda:0")
R, L, hs, ind = 20011, 5, 256, int(sys.argv[2])
lengths = torch.randint(1, L + 1, (R,)); lengths[0] = L
sl, _ = torch.sort(lengths, descending=True, stable=True)
bs = [int((sl > t).sum()) for t in range(L)]
x = (0.5 * torch.randn(sum(bs), ind)).to(dev, torch.bfloat16)
ws = []
for l in range(2):
    for _ in range(2):
        i = ind if l == 0 else 2 * hs
        ws += [(0.08 * torch.randn(3 * hs, i)).to(dev), (0.08 * torch.randn(3 * hs, hs)).to(dev), (0.1 * torch.randn(3 * hs)).to(dev), (0.1 * torch.randn(3 * hs)).to(dev)]
ops.set_seed(5)
out = gru.bigru_final(x, bs, hs, 2, 0.25, ws)
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
'''


@pytest.mark.parametrize("ind", [128, 512])
def test_gru_forward_pipelined_kernel_bit_identical_to_single_stage(ind, tmp_path):
    """The fused forward step's pipelined k loop -- gru_step_fwd_a2w3_kernel: three slots of activation rows + two of weight rows per 64-k
    stage, 256-row panels on eight waves; launches of at least 8192 rows -- against gru_step_fwd_kernel<1> (one 64-k stage, 128-row panels):
    same lane -> channel map, same k order, same cell -> the SAME bits, through two GRU layers with inter-layer dropout, ragged lengths
    (steps of 20,011 .. ~4,000 rows: both kernels run in the default process), a partial last row panel.  The switch is read once per
    process: two child processes.  (Round 5 compared seven forms this way; round 6 removed all but these two.)"""
    import subprocess
    import sys
    outs = []
    for a2w3 in ("1", "0"):
        f = str(tmp_path / ("a%s.pt" % a2w3))
        env = dict(os.environ, GTOS_GRU_FWD_A2W3=a2w3, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", _RING_CHILD, f, str(ind)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
        outs.append(torch.load(f))
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0.05
    assert torch.equal(outs[0], outs[1])


def test_packed_path_gru_backward_d4_in_place_is_the_same_function(monkeypatch):
    """GTOS_GRU_D4_INPLACE (automatic for buffers of 8 GB and more: C5): the backward cell tiles write d4 over the saved gates instead of
    into a buffer of their own -- same launches, same operands, same bits in every gradient (training mode, dropout on)."""
    from gtos_amd import gru, ops
    g = torch.Generator().manual_seed(11)
    length = torch.randint(1, 9, (2500,), generator=g)
    bank = torch.randint(1, 90, (8, 2500), generator=g)
    for r in range(2500):
        bank[int(length[r]):, r] = 0
    ref, m = _relenc_pair(bank, length, hid=64)
    m.compute_dtype = torch.bfloat16
    m.dropout = 0.25
    m.train()
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1)).to(dev())
    res = []
    for mode in ("0", "1"):
        monkeypatch.setattr(gru, "D4_INPLACE", mode)
        ops.set_seed(78)
        m.zero_grad()
        out = m(bank.to(dev()), length.to(dev()))
        (out.float() * wout).sum().backward()
        ops.join_side()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), _grads_of(m)))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=1e-5, atol=1e-6, msg=lambda s_, k=k: "%s: %s" % (k, s_))
    assert float(max(v.abs().max() for v in res[0][1].values())) > 0


def test_packed_path_gru_edge_cases():
    """ADVICE round 5.  (i) A second backward raises a clear error (the saved gates are released during the first).  (ii) A bank without an
    active step (every path empty) gives zero vectors and no gradient.  (The third finding -- GTOS_GRU_RECOMPUTE_HN left the packed path's
    hn block unwritten -- is gone with the switch: round 6 removed it and its kernel instantiation.)"""
    from gtos_amd import gru, ops
    bank, length = _relenc_case(R=900, L=7, V=86)
    ref, m = _relenc_pair(bank, length, hid=64)
    m.compute_dtype = torch.bfloat16
    m.dropout = 0.25
    m.train()
    wout = torch.randn(bank.shape[1], 64, generator=torch.Generator().manual_seed(1)).to(dev())
    ops.set_seed(31)
    out = m(bank.to(dev()), length.to(dev()))
    loss = (out.float() * wout).sum()
    loss.backward()
    ops.join_side()
    torch.cuda.synchronize()
    for k, g in _grads_of(m).items():
        assert torch.isfinite(g).all(), k
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    plan = gru.PackPlan([], torch.zeros(5, dtype=torch.int32, device=dev()))
    assert plan.L == 0
    table = torch.randn(86, 20, device=dev(), requires_grad=True)
    ws = [p.detach() for n_, p in m.rnn.named_parameters()]
    fin = gru.packed_path_gru(torch.zeros(7, 5, dtype=torch.int64, device=dev()), plan, table, 64, 0.1, 64, 0.1, ws)
    assert fin.shape == (5, 128) and float(fin.abs().max()) == 0
    fin.sum().backward()
    assert table.grad is None or float(table.grad.abs().max()) == 0
