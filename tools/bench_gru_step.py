"""Isolated timing of the packed-path RelationEncoder kernels at C2 size (one launch at a time, HIP events): forward step of layer 0 / 1,
backward step with and without the input-gradient tiles, the grouped weight-gradient product.  Random operands (timing only).
    python tools/bench_gru_step.py [--rows 434624] [--reps 10] [--only fwd,bwd,dinp,dw]
(The measuring switches of round 5 -- k loop alone, cell alone, DMA alone ... -- left the kernels in round 6; what they showed is in DESIGN.md section 5.)
Prints per kernel: us per launch, algorithmic GB/s, TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import gru, ops                                                    # noqa: E402
from gtos_amd._lib import call, ptr, stream                                      # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=434624)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--p-drop", type=float, default=0.0, help="dropout probability of the backward legs (layer 0: output mask and embedding mask)")
    a = ap.parse_args()
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    A, hs = a.rows, 256
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(bf)                     # noqa: E731
    for layer, ind in ((0, 128), (1, 512)):
        x, h_in = r(A, ind), r(A, hs)
        wi, wh = r(3 * hs, ind), r(3 * hs, hs)
        bi, bh = torch.zeros(3 * hs, device=dev), torch.zeros(3 * hs, device=dev)
        h_out, gates = torch.empty(A, hs, device=dev, dtype=bf), torch.empty(A, 4 * hs, device=dev, dtype=bf)
        Y = torch.empty(A, 2 * hs, device=dev, dtype=bf) if layer == 0 else None
        nb = A * 2 * (ind + hs + 4 * hs + hs + (hs if Y is not None else 0))
        fl = 2.0 * A * 3 * hs * (ind + hs)
        if not a.only or "fwd" in a.only:
            us = timed(lambda: gru._step_fwd_call(A, hs, x, None, h_in, wi, bi, wh, bh, h_out, A, None, gates, None if Y is None else Y.data_ptr(),
                                                  2 * hs, 0.2 if Y is not None else 0.0, 77, 0, None, None, None, None, None, True), a.reps)
            print("fwd  L%d  rows %d: %8.1f us  %7.1f GB/s  %7.1f TF/s" % (layer, A, us, nb / us / 1e3, fl / us / 1e6), flush=True)
        # backward: d4_prev of a same-size previous step
        d4p, d4 = r(A, 4 * hs), torch.empty(A, 4 * hs, device=dev, dtype=bf)
        wh_t, wi_t = r(hs, 3 * hs), r(ind, 3 * hs)
        gts = torch.rand(A, 4 * hs, device=dev).to(bf)
        dh = r(A, hs)
        dy = r(A, 2 * hs) if layer == 0 else None
        dinp = torch.empty(A, ind, device=dev, dtype=bf)
        bpart = torch.zeros(gru.N_BIAS_PARTIALS, 4 * hs, device=dev)
        for fused in (False, True):
            if a.only and "bwd" not in a.only:
                continue
            for acc in ((False, True) if fused else (False,)):
                kw = dict(wi_t=wi_t, dinp=dinp, n_in=ind, dinp_acc=acc) if fused else {}
                pdrop = a.p_drop if dy is not None else 0.0            # (the layer-output dropout mask is re-drawn per element in the cell)
                if fused and layer == 0:
                    kw.update(p_in=a.p_drop, seed_in=5)
                us = timed(lambda: gru._step_bwd_fused(A, hs, d4p, A, wh_t, gts, h_in, None if dy is None else dy.data_ptr(), 2 * hs, dh, d4, pdrop, 17, 0, bpart, **kw), a.reps)
                nbb = 2 * (A * (11 * hs + (hs if dy is not None else 0)) + A * (4 if fused else 3) * hs + (A * ind * (2 if acc else 1) if fused else 0))
                flb = 2.0 * A * 3 * hs * (hs + (ind if fused else 0))
                print("bwd  L%d  rows %d %-9s %-10s: %8.1f us  %7.1f GB/s (%.3f of 8 TB/s)  %7.1f TF/s" % (
                    layer, A, "cell+dinp" if fused else "cell only", "accumulate" if acc else "", us, nbb / us / 1e3, nbb / us / 8e6, flb / us / 1e6), flush=True)
        if not a.only or "dinp" in a.only:
            us = timed(lambda: gru._step_bwd_fused(0, hs, d4p, A, wh_t, None, None, None, 2 * hs, None, None, 0.0, 0, 0, None, wi_t=wi_t, dinp=dinp, n_in=ind), a.reps)
            print("dinp L%d  rows %d role B alone: %8.1f us  %7.1f TF/s;" % (layer, A, us, 2.0 * A * 3 * hs * ind / us / 1e6), end=" ", flush=True)
            us = timed(lambda: ops.gemm(d4p[:, :3 * hs], wi_t, trans_b=True, out=dinp), a.reps)
            print("as a GEMM: %8.1f us  %7.1f TF/s" % (us, 2.0 * A * 3 * hs * ind / us / 1e6), flush=True)
        if not a.only or "dw" in a.only:
            N = 2497192
            d4n, xn, hn = r(N, 4 * hs), r(N, ind), r(N, hs)
            gi, gh = torch.zeros(3 * hs, ind, device=dev), torch.zeros(3 * hs, hs, device=dev)
            ws = ops._workspace(dev)
            us = timed(lambda: call("gtos_gru_weight_grads", N, hs, ind, ind, ptr(d4n), ptr(xn), ind, ptr(hn), hs, ptr(gi), ind, ptr(gh), hs, ptr(ws),
                                    ws.numel() * 4, stream()), max(3, a.reps // 2))
            print("dW   L%d  rows %d grouped: %8.1f us  %7.1f TF/s  %7.1f GB/s" % (layer, N, us, 2.0 * N * 3 * hs * (ind + hs) / us / 1e6,
                                                                                 N * 2 * (4 * hs + ind + hs) / us / 1e3), flush=True)

            def three():
                ops.gemm(d4n[:, :2 * hs], hn, trans_a=True, out=gh[:2 * hs], accumulate=True, splitk=ops._splitk(2 * hs, hs, N))
                ops.gemm(d4n[:, 3 * hs:], hn, trans_a=True, out=gh[2 * hs:], accumulate=True, splitk=ops._splitk(hs, hs, N))
                ops.gemm(d4n[:, :3 * hs], xn, trans_a=True, out=gi, accumulate=True, splitk=ops._splitk(3 * hs, ind, N))
            us = timed(three, max(3, a.reps // 2))
            print("dW   L%d  rows %d three GEMMs: %8.1f us  %7.1f TF/s" % (layer, N, us, 2.0 * N * 3 * hs * (ind + hs) / us / 1e6), flush=True)
            del d4n, xn, hn


if __name__ == "__main__":
    main()
