"""Repeated-run bitwise soak of the LDS-DMA pipelined kernels at production launch sizes.

Round 5 shipped -- for at least a round -- pipelined k loops whose slot-freeing barriers were not preceded by ``lgkmcnt(0)``: hipcc sinks a
stage's last MFMAs, and the wait of the fragment reads that feed them, below the stage's closing ``s_barrier``, so a wave could be past the
barrier with ``ds_read``s of a slot still queued while another wave's LDS-DMA refilled it (gru_step.hip, GTOS_VMCNT_LDS).  It was found by
luck (one parametrisation of one test at 40,000 rows).  This file is the systematic version: every kernel that refills LDS slots by DMA while
other waves may still read them is launched >= 30 times on the same inputs at the sizes the C2 / C5 steps launch it, beside a second stream
that keeps HBM busy (latency under load is what moves such races), and every output is compared bit for bit with the first launch.

  forward GRU step   gru_step_fwd_a2w3_kernel (three slots of activation rows + two of weight rows per 64-k stage)
  backward GRU step  gru_step_bwd_kernel with role B (kloop_a2: the row panel one stage ahead)
  GEMM               gemm256q_nt_kernel, gemm256p_tn_kernel (split-K), the grouped TN product of gtos_gru_weight_grads, the batched one of gtos_gemm_tn_batch
  whole function     the packed-path RelationEncoder over the whole C2 bank, training mode, DENSE upstream gradient, run twice

`tools/race_demo.sh` rebuilds the library with -DGTOS_RACE_DEMO (the waits as they were before commit 9d39564) and shows this file failing."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

REPS = int(os.environ.get("GTOS_SOAK_REPS", "30"))
C2_ROWS, MID_ROWS, C5_ROWS = 434624, 40003, 1840000       # first steps of a C2 direction, a late step, C5's widest step


def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda:0")


class Hammer(object):
    """A second stream that streams two 1 GiB buffers back and forth while the soaked launches run."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.a = torch.empty(1 << 28, dtype=torch.float32, device=dev())
        self.b = torch.empty_like(self.a)
        self.a.zero_()

    def push(self, n=3):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.b.copy_(self.a, non_blocking=True)

    def drain(self):
        self.stream.synchronize()


_HAMMER = {}


def hammer():
    if "h" not in _HAMMER:
        _HAMMER["h"] = Hammer()
    return _HAMMER["h"]


def soak(name, launch, outputs, reset=None, reps=None):
    """launch() writes ``outputs`` (tensors); reset() restores in-place operands.  First launch = reference; every later one must match."""
    reps = reps or REPS
    h = hammer()
    if reset:
        reset()
    launch()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outputs]
    assert all(bool(torch.isfinite(r.float()).all()) for r in ref), name
    assert any(float(r.float().abs().max()) > 0 for r in ref), name
    bad = []
    for i in range(reps):
        for o in outputs:
            o.fill_(0)                                      # a launch that skips rows shows up as zeros, not as the previous launch's values
        if reset:
            reset()
        h.push()
        launch()
        torch.cuda.synchronize()
        for k, (o, r) in enumerate(zip(outputs, ref)):
            if not torch.equal(o, r):
                rows = (o.view(o.shape[0], -1) != r.view(r.shape[0], -1)).any(1)
                bad.append((i, k, int(rows.sum()), int(rows.nonzero()[0])))
    h.drain()
    assert not bad, "%s: %d of %d repeated launches differ from the first (launch, output, differing rows, first row): %s" % (
        name, len(set(b[0] for b in bad)), reps, bad[:6])


def r_(*shape, scale=0.3):
    return (torch.randn(*shape, device=dev()) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("rows", [C2_ROWS, MID_ROWS, C5_ROWS])
@pytest.mark.parametrize("layer", [0, 1])
def test_soak_forward_gru_step(rows, layer):
    from gtos_amd._lib import call, ptr, stream
    torch.manual_seed(rows % 1000 + layer)
    hs, ind = 256, (128, 512)[layer]
    x, h_in, wi, wh = r_(rows, ind), r_(rows, hs), r_(3 * hs, ind, scale=0.1), r_(3 * hs, hs, scale=0.1)
    bi, bh = torch.randn(3 * hs, device=dev()) * 0.1, torch.randn(3 * hs, device=dev()) * 0.1
    n_out = rows - rows // 7                                # some rows finish at this step
    # (zeros: a row goes EITHER to h_out (m < n_out) or to h_fin, and y is a column block -- the untouched parts must compare equal too)
    h_out = torch.zeros(rows, hs, device=dev(), dtype=torch.bfloat16)
    h_fin = torch.zeros(rows, hs, device=dev(), dtype=torch.bfloat16)
    gates = torch.zeros(rows, 4 * hs, device=dev(), dtype=torch.bfloat16)
    y = torch.zeros(rows, 2 * hs, device=dev(), dtype=torch.bfloat16) if layer == 0 else None

    def launch():
        call("gtos_gru_step_fwd", rows, hs, ptr(x), ind, ind, ptr(wi), ptr(bi), None, None, None, None, None, ptr(h_in), None, ptr(wh), ptr(bh),
             ptr(h_out), n_out, ptr(h_fin), hs, None, ptr(gates), ptr(y), 2 * hs, 0.2 if y is not None else 0.0, 99, 0, stream())
    soak("fwd step L%d rows %d" % (layer, rows), launch, [h_out, h_fin, gates] + ([y] if y is not None else []))


@pytest.mark.parametrize("rows,rows_prev", [(C2_ROWS, C2_ROWS), (MID_ROWS, 50001), (50001, MID_ROWS), (C5_ROWS, C5_ROWS - 12345)])
@pytest.mark.parametrize("layer", [0, 1])
def test_soak_backward_gru_step_with_input_gradient_role(rows, rows_prev, layer):
    from gtos_amd.gru import _step_bwd_fused, N_BIAS_PARTIALS
    torch.manual_seed(rows % 1000 + layer)
    hs, n_in = 256, (128, 512)[layer]
    d4_prev, wh_t, wi_t = r_(rows_prev, 4 * hs), r_(hs, 3 * hs, scale=0.1), r_(n_in, 3 * hs, scale=0.1)
    gates = torch.rand(rows, 4 * hs, device=dev()).to(torch.bfloat16)
    hprev, dh0 = r_(rows, hs), r_(rows, hs)
    dy = r_(rows, 2 * hs) if layer == 0 else None
    dh, d4 = torch.empty_like(dh0), torch.empty(rows, 4 * hs, device=dev(), dtype=torch.bfloat16)
    dinp = torch.empty(rows_prev, n_in, device=dev(), dtype=torch.bfloat16)
    bpart = torch.zeros(N_BIAS_PARTIALS, 4 * hs, device=dev())

    def launch():
        _step_bwd_fused(rows, hs, d4_prev, rows_prev, wh_t, gates, hprev, None if dy is None else dy.data_ptr(), 2 * hs, dh, d4,
                        0.2 if dy is not None else 0.0, 17, 0, bpart, wi_t=wi_t, dinp=dinp, n_in=n_in, p_in=0.2 if layer == 0 else 0.0, seed_in=5)
    soak("bwd step L%d rows %d/%d" % (layer, rows, rows_prev), launch, [dh, d4, dinp], reset=lambda: dh.copy_(dh0))


@pytest.mark.parametrize("M,N,K", [(C2_ROWS, 1024, 4096), (C2_ROWS, 512, 8192), (C2_ROWS, 1024, 1024), (MID_ROWS, 512, 2016)])
def test_soak_gemm_nt_pingpong(M, N, K):
    """gemm256q_nt_kernel (K % 64 == 0: eight half-tile buffers refilled by DMA two phases after their last read) at the bank-gradient
    slab shape, the relation-projection backward, a deep K and a K % 64 == 32 (the zero-filled half of the last k tile)."""
    from gtos_amd import ops
    torch.manual_seed(K)
    a, b = r_(M, K), r_(N, K, scale=0.1)
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    soak("gemm NT %dx%dx%d" % (M, N, K), lambda: ops.gemm(a, b, trans_b=True, out=out), [out], reps=max(10, REPS // 2) if K >= 4096 else None)


@pytest.mark.parametrize("rows,M,N", [(2497192, 768, 256), (C2_ROWS, 1024, 512)])
def test_soak_gemm_tn_splitk(rows, M, N):
    """gemm256p_tn_kernel + the fixed-order split-K reduction: weight-gradient shapes (GRU recurrent weights over all packed rows; a
    relation projection)."""
    from gtos_amd import ops
    torch.manual_seed(M)
    dy, x = r_(rows, M), r_(rows, N)
    out = torch.zeros(M, N, device=dev())
    soak("gemm TN %d: %dx%d" % (rows, M, N), lambda: ops.gemm(dy, x, trans_a=True, out=out, splitk=ops._splitk(M, N, rows)), [out], reps=max(10, REPS // 2))


@pytest.mark.parametrize("layer", [0, 1])
def test_soak_grouped_weight_gradients(layer):
    from gtos_amd import ops
    from gtos_amd._lib import call, ptr, stream
    torch.manual_seed(layer)
    N, hs, ind = 2497192, 256, (128, 512)[layer]
    valid = 100 if layer == 0 else ind
    d4, x, hp = r_(N, 4 * hs), r_(N, ind), r_(N, hs)
    gi, gh = torch.zeros(3 * hs, valid, device=dev()), torch.zeros(3 * hs, hs, device=dev())
    ws = ops._workspace(dev())

    def launch():
        call("gtos_gru_weight_grads", N, hs, ind, valid, ptr(d4), ptr(x), ind, ptr(hp), hs, ptr(gi), valid, ptr(gh), hs, ptr(ws), ws.numel() * 4, stream())
    soak("grouped dW L%d" % layer, launch, [gi, gh], reps=max(10, REPS // 3))


def test_soak_batched_small_weight_gradients():
    """gtos_gemm_tn_batch (round 6: gemm256p_tn_kernel in its job-table mode): the weight gradients of a C2 step's graph and decoder layers -- 36
    jobs, the row counts and shapes ops.flush_dw() hands it -- launched repeatedly on the same operands into zeroed targets: no split-K, one
    writer per tile, so every target must be the same bits every time (the bias column sums go through fp32 atomics and are left out)."""
    import ctypes
    from gtos_amd._lib import call, stream
    torch.manual_seed(9)
    shapes = [(6464, 1536, 512), (6464, 512, 512), (6464, 1024, 512), (6464, 512, 1024)] * 6 + [(3200, 512, 512), (3200, 1024, 512), (3200, 1536, 512)] * 4
    jobs = [(r_(K, M), r_(K, N), torch.zeros(M, N, device=dev())) for K, M, N in shapes]
    n = len(jobs)
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    A = vp(*[j[0].data_ptr() for j in jobs]); B = vp(*[j[1].data_ptr() for j in jobs]); C = vp(*[j[2].data_ptr() for j in jobs])
    nob = vp(*[None] * n)
    lda = i64(*[j[0].stride(0) for j in jobs]); ldb = i64(*[j[1].stride(0) for j in jobs]); ldc = i64(*[j[2].stride(0) for j in jobs])
    M_ = i32(*[j[0].shape[1] for j in jobs]); N_ = i32(*[j[1].shape[1] for j in jobs]); K_ = i32(*[j[0].shape[0] for j in jobs])

    def launch():
        call("gtos_gemm_tn_batch", n, ctypes.addressof(A), ctypes.addressof(lda), ctypes.addressof(M_), ctypes.addressof(B), ctypes.addressof(ldb),
             ctypes.addressof(N_), ctypes.addressof(K_), ctypes.addressof(C), ctypes.addressof(ldc), ctypes.addressof(nob), stream())
    soak("batched small dW", launch, [j[2] for j in jobs])


def test_soak_c2_full_size_packed_path_backward_dense_upstream_run_to_run():
    """The whole C2 bank through the packed-path RelationEncoder in training mode (p = 0.2), upstream gradient DENSE (every one of the
    434,624 paths weighted -- test_full_size_c2.py's oracle leg weights every 400th), three times with the same seeds beside the
    hammer stream: outputs and every weight / embedding gradient bit-identical run to run (the bias gradients are sums of fp32 atomics
    in arrival order: compared to 1e-5 of their norm)."""
    from gtos_amd import ops
    from test_full_size_c2 import c2_batch, _encoder_pair
    batch, _ = c2_batch()
    bank, length, trie = batch["relation_bank"].to(dev()), batch["relation_length"].to(dev()), batch["relation_trie"].to(dev())
    _, m = _encoder_pair(scale_rnn=1.5, dropout=0.2)
    m.compute_dtype = torch.bfloat16
    m.train()
    wout = torch.randn(bank.shape[1], 512, generator=torch.Generator().manual_seed(3)).to(dev())
    h = hammer()
    runs = []
    for _ in range(3):
        ops.set_seed(2468)
        m.zero_grad()
        h.push(40)
        out = m(bank, length, trie=trie)
        (out.float() * wout).sum().backward()
        ops.join_side()
        torch.cuda.synchronize()
        runs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    h.drain()
    assert float(runs[0][0].float().abs().max()) > 0.01
    for out, grads in runs[1:]:
        assert torch.equal(out, runs[0][0])
        for k, g in grads.items():
            g0 = runs[0][1][k]
            if "bias" in k:
                assert float((g - g0).norm()) <= 1e-5 * float(g0.norm()) + 1e-12, k
            else:
                assert torch.equal(g, g0), (k, int((g != g0).sum()))
