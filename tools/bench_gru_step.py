"""Isolated timing of the packed-path RelationEncoder kernels at C2 size (one launch at a time, HIP events): forward step of layer 0 / 1,
backward step with and without the input-gradient tiles, the grouped weight-gradient product.  Random operands (timing only).
    python tools/bench_gru_step.py [--rows 434624] [--reps 10]
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


def concurrent(a):
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    A, hs = a.rows, 256
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(bf)                     # noqa: E731
    for layer, ind in ((0, 128), (1, 512)):
        bufs = []
        for _ in range(2):                                                        # two independent problem instances
            bufs.append(dict(x=r(A, ind), h_in=r(A, hs), wi=r(3 * hs, ind), wh=r(3 * hs, hs), bi=torch.zeros(3 * hs, device=dev),
                             bh=torch.zeros(3 * hs, device=dev), h_out=torch.empty(A, hs, device=dev, dtype=bf),
                             gates=torch.empty(A, 4 * hs, device=dev, dtype=bf)))
        s2 = torch.cuda.Stream()

        def launch(b, dbg):
            call("gtos_gru_step_fwd", A, hs, ptr(b["x"]), ind, ind, ptr(b["wi"]), ptr(b["bi"]), None, None, None, None, None, ptr(b["h_in"]), None,
                 ptr(b["wh"]), ptr(b["bh"]), ptr(b["h_out"]), A, None, hs, None, ptr(b["gates"]), None, 2 * hs, 0.0, 0, 0, 1 | (dbg << 8), stream())

        def serial(d0, d1):
            launch(bufs[0], d0); launch(bufs[1], d1)

        def side_by_side(d0, d1):
            s2.wait_stream(torch.cuda.current_stream())
            launch(bufs[0], d0)
            with torch.cuda.stream(s2):
                launch(bufs[1], d1)
            torch.cuda.current_stream().wait_stream(s2)
        for name, d0, d1 in (("k loop alone + cell alone", 2, 1), ("DMA alone + cell alone", 3, 1), ("reads+MFMA alone + cell alone", 4, 1),
                             ("DMA alone + reads+MFMA alone", 3, 4), ("cell alone + cell alone", 1, 1), ("k loop alone + k loop alone", 2, 2)):
            t_ser = timed(lambda: serial(d0, d1), a.reps)
            t_par = timed(lambda: side_by_side(d0, d1), a.reps)
            print("L%d %-34s one after the other %7.1f us   side by side on two streams %7.1f us   (%.2f)" % (layer, name, t_ser, t_par, t_par / t_ser), flush=True)


def phase(a):
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    A, hs = a.rows, 256
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(bf)                     # noqa: E731
    for layer, ind in ((0, 128), (1, 512)):
        b = dict(x=r(A, ind), h_in=r(A, hs), wi=r(3 * hs, ind), wh=r(3 * hs, hs), bi=torch.zeros(3 * hs, device=dev),
                 bh=torch.zeros(3 * hs, device=dev), h_out=torch.empty(A, hs, device=dev, dtype=bf),
                 gates=torch.empty(A, 4 * hs, device=dev, dtype=bf))

        def launch(dbg, nw, delay_us):
            call("gtos_gru_step_fwd", A, hs, ptr(b["x"]), ind, ind, ptr(b["wi"]), ptr(b["bi"]), None, None, None, None, None, ptr(b["h_in"]), None,
                 ptr(b["wh"]), ptr(b["bh"]), ptr(b["h_out"]), A, None, hs, None, ptr(b["gates"]), None, 2 * hs, 0.0, 0, 0,
                 1 | (dbg << 8) | (nw << 16) | (delay_us << 20), stream())
        launch(0, 4, 0)
        ref = b["h_out"].clone()
        print("L%d  eight waves %7.1f us   four waves %7.1f us   (k loop alone %7.1f, cell alone %7.1f)" % (
            layer, timed(lambda: launch(0, 8, 0), a.reps), timed(lambda: launch(0, 4, 0), a.reps),
            timed(lambda: launch(2, 4, 0), a.reps), timed(lambda: launch(1, 4, 0), a.reps)), flush=True)
        for dbg, what in ((5, "workgroups 256..511 late"), (6, "every other of the first 512 late")):
            for d in (0, 20, 40, 60, 80, 120, 200):
                t = timed(lambda: launch(dbg, 4, d), a.reps)
                same = bool((b["h_out"] == ref).all())
                print("L%d  %-36s by %3d us: %7.1f us   %s" % (layer, what, d, t, "same result" if same else "RESULT DIFFERS"), flush=True)


def dbuf(a):
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    hs = 256
    torch.manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(bf)                     # noqa: E731
    for A in (a.rows, 100003):
        for layer, ind in ((0, 128), (1, 512)):
            b = dict(x=r(A, ind), h_in=r(A, hs), wi=r(3 * hs, ind), wh=r(3 * hs, hs), bi=torch.randn(3 * hs, device=dev) * 0.1,
                     bh=torch.randn(3 * hs, device=dev) * 0.1)

            def launch(dbg, nw, out):
                call("gtos_gru_step_fwd", A, hs, ptr(b["x"]), ind, ind, ptr(b["wi"]), ptr(b["bi"]), None, None, None, None, None, ptr(b["h_in"]), None,
                     ptr(b["wh"]), ptr(b["bh"]), ptr(out[0]), A, None, hs, None, ptr(out[1]), None, 2 * hs, 0.0, 0, 0,
                     1 | (dbg << 8) | (nw << 16), stream())
            outs = {}
            for name, nw in (("ring, eight waves", 8), ("two slots of 64-k, 8 waves", 2), ("two slots of 64-k, 4 waves", 3), ("A two slots + W three slots", 6), ("A three slots + W two slots", 7)):
                out = (torch.zeros(A, hs, device=dev, dtype=bf), torch.zeros(A, 4 * hs, device=dev, dtype=bf))
                launch(0, nw, out)
                torch.cuda.synchronize()
                outs[name] = (out[0].clone(), out[1].clone())       # (the timed launches below overwrite `out`)
                t = [timed(lambda d=d: launch(d, nw, out), a.reps) for d in (0, 2, 3, 1)]
                print("rows %7d L%d  %-26s full %7.1f us   k loop alone %7.1f   its DMA alone %7.1f   cell alone %7.1f" % ((A, layer, name) + tuple(t)), end="", flush=True)
                if nw == 2:
                    print("   DMA alone with two stages in flight %7.1f" % timed(lambda: launch(10, nw, out), a.reps), end="")
                if nw == 2:                    # the cell's loads and stores with whole 128-byte row segments per 8 consecutive lanes (timing only: wrong places)
                    print("   LINE-WISE cell accesses (timing only): cell alone %7.1f, full %7.1f" % (timed(lambda: launch(8, nw, out), a.reps), timed(lambda: launch(9, nw, out), a.reps)), end="")
                if nw in (2, 3):               # mixed roles: half the workgroups DMA alone, half cell alone -> (DMA alone + cell alone) / 2 if they queue on one resource
                    print("   mixed roles %7.1f (half the sum %7.1f, half the larger %7.1f)" % (timed(lambda: launch(7, nw, out), a.reps), (t[2] + t[3]) / 2, max(t[2], t[3]) / 2), end="")
                print(flush=True)
            vals = list(outs.values())
            print("rows %7d L%d  bit-identical to the ring: state %s, gates %s" % (A, layer, [bool((vals[0][0] == v[0]).all()) for v in vals[1:]],
                                                                                  [bool((vals[0][1] == v[1]).all()) for v in vals[1:]]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=434624)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--concurrent", action="store_true",
                    help="forward step of layer 1: its k loop alone and its cell alone (the kernel's measuring switches, per launch) one after the other "
                         "on one stream and SIDE BY SIDE on two streams -- do the two phases use independent resources?")
    ap.add_argument("--phase", action="store_true",
                    help="forward step, four waves (two workgroups per CU): some first-generation workgroups start late, so that the two "
                         "workgroups of a CU are in opposite phases (k loop / cell) -- measuring switches 5 / 6 of the kernel")
    ap.add_argument("--dbuf", action="store_true",
                    help="forward step: the three-slot ring of 32-k stages (64-byte rows) against two slots of 64-k stages (whole 128-byte lines "
                         "per row and DMA instruction), each whole and split by the measuring switches; results compared bit for bit")
    a = ap.parse_args()
    if a.dbuf:
        return dbuf(a)
    if a.concurrent:
        return concurrent(a)
    if a.phase:
        return phase(a)
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
                us = timed(lambda: gru._step_bwd_fused(A, hs, d4p, A, wh_t, gts, h_in, None if dy is None else dy.data_ptr(), 2 * hs, dh, d4, 0.0, 0, 0, bpart, **kw), a.reps)
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
