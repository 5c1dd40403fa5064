"""Bidirectional multi-layer GRU over packed, length-sorted relation label paths (final states only).

MI355X counterpart of ``nn.utils.rnn.pack_padded_sequence`` + ``nn.GRU`` as RelationEncoder uses them
(/root/reference/generator/encoder.py:93-111), with explicit BPTT.  bf16 (hidden size a multiple of 64): one fused MFMA
kernel per time step and direction, forward (gtos_gru_step_fwd: x W_ih^T and h W_hh^T gate products + cell; the state
is written straight into the next step's operand slot) and backward (gtos_gru_step_bwd: recurrent gradient product +
cell backward + bias sums, one d4 = [dr|dz|dn_x|dn_h] buffer).  fp32 / other sizes: the input-gate products of ALL
steps are one GEMM, each step one [active,h]x[h,3h] GEMM plus gtos_gru_cell_fwd / gtos_gru_cell_bwd.  Weight and
input gradients are GEMMs over all steps at once in both paths.
"""
import os

import torch

from . import _lib
from ._lib import call, dt, ptr, stream
from .ops import gemm, compute_weight, next_seed, weight_t, _grad_target, _splitk, side_stream as _side_stream, defer_side_join


def _cell_fwd(A, hs, xg, hg, h, y, y_off_elems, ldy, hprev, gates, p, seed, drop_base):
    yp = None if y is None else y.data_ptr() + y_off_elems * y.element_size()
    call("gtos_gru_cell_fwd", dt(xg), A, hs, ptr(xg), ptr(hg), ptr(h), yp, ldy, ptr(hprev), ptr(gates),
         float(p), seed, drop_base, stream())


N_BIAS_PARTIALS = 1024

# Forward time step on bf16: "x" = input AND recurrent gate products + cell in ONE kernel (gtos_gru_step_fwd),
# "h" = recurrent product + cell fused, input gates by one big GEMM, "off" = GEMM + cell kernel per step (the fp32 path).
FUSE = os.environ.get("GTOS_GRU_FUSE", "x")
# Backward: run the weight-gradient GEMMs of one direction on a second HIP stream while the (memory-bound) BPTT steps of
# the next direction occupy the main stream: the step kernels are limited to 2 waves per SIMD by registers, an MFMA GEMM
# wave fits beside them.
SIDE_STREAM = os.environ.get("GTOS_GRU_SIDE", "1") != "0"
SIDE_MIN_ROWS = 200000          # below this the GEMMs are launch-bound and the stream hand-over costs more than it hides


def _step_fwd(A, hs, x, xg, h_in, wi, b_ih, wh, b_hh, h_out, n_out, h_fin, gates, y, y_off_elems, ldy, p, seed, drop_base):
    yp = None if y is None else y.data_ptr() + y_off_elems * y.element_size()
    call("gtos_gru_step_fwd", A, hs, ptr(x), 0 if x is None else x.stride(0), 0 if x is None else x.shape[1],
         ptr(wi) if x is not None else None, ptr(b_ih) if x is not None else None, ptr(xg), ptr(h_in), ptr(wh), ptr(b_hh),
         ptr(h_out), n_out, ptr(h_fin), ptr(gates), yp, ldy, float(p), seed, drop_base, stream())


def _cell_bwd(A, hs, gates, hprev, dy, dy_off_elems, ldy, dh, dxg, dhg, p, seed, drop_base, bpart):
    dyp = None if dy is None else dy.data_ptr() + dy_off_elems * dy.element_size()
    call("gtos_gru_cell_bwd", dt(gates), A, hs, ptr(gates), ptr(hprev), dyp, ldy, ptr(dh), ptr(dxg), ptr(dhg),
         float(p), seed, drop_base, ptr(bpart), 0 if bpart is None else bpart.shape[0], stream())


class BiGRUFinalFn(torch.autograd.Function):
    """x [N,in] packed time-major rows (step t occupies rows offs[t]:offs[t]+batch_sizes[t], sequences sorted by
    decreasing length); returns the top layer's [fwd final ; bwd final] state per sequence, [R, 2*hs]."""

    @staticmethod
    def forward(ctx, x, batch_sizes, hs, num_layers, p_drop, *weights):
        # weights: per layer [w_ih, w_hh, b_ih, b_hh, w_ih_rev, w_hh_rev, b_ih_rev, b_hh_rev]
        L, R = len(batch_sizes), batch_sizes[0]
        offs = [0]
        for a in batch_sizes:
            offs.append(offs[-1] + a)
        N = offs[-1]
        assert x.shape[0] == N
        dev, dtp = x.device, x.dtype
        inp = x.contiguous()
        saved = []
        finals = None
        for l in range(num_layers):
            last = l == num_layers - 1
            Y = None if last else torch.empty((N, 2 * hs), dtype=dtp, device=dev)
            seed = next_seed() if (p_drop > 0 and not last) else 0
            pl = p_drop if not last else 0.0
            finals = []
            layer_saved = []
            for direction in (0, 1):
                w_ih, w_hh, b_ih, b_hh = weights[l * 8 + direction * 4: l * 8 + direction * 4 + 4]
                wi, wh = compute_weight(w_ih, dtp), compute_weight(w_hh, dtp)
                gates = torch.empty((N, 4 * hs), dtype=dtp, device=dev)
                hprev = torch.empty((N, hs), dtype=dtp, device=dev)
                steps = range(L) if direction == 0 else range(L - 1, -1, -1)
                fuse = FUSE if (dtp == torch.bfloat16 and hs % 64 == 0 and inp.shape[1] % 8 == 0) else "off"
                if fuse != "off":
                    # hprev[offs[t] + m] IS the state row m enters step t with: each step writes its result straight into
                    # the next step's slot (or into `h` once the sequence is finished), nothing is copied
                    h = torch.empty((R, hs), dtype=dtp, device=dev)
                    xg = None if fuse == "x" else gemm(inp, wi, trans_b=True, bias=b_ih.detach())
                    bi, bh = b_ih.detach(), b_hh.detach()
                    if direction == 0:
                        hprev[:R].zero_()
                    else:                               # rows that become active at step t start from h = 0
                        for t in range(L):
                            lo = batch_sizes[t + 1] if t + 1 < L else 0
                            if batch_sizes[t] > lo:
                                hprev[offs[t] + lo: offs[t] + batch_sizes[t]].zero_()
                    for t in steps:
                        A, off = batch_sizes[t], offs[t]
                        nxt = t + 1 if direction == 0 else t - 1
                        if 0 <= nxt < L:
                            h_out, n_out = hprev[offs[nxt]:], min(A, batch_sizes[nxt])
                        else:
                            h_out, n_out = h, A
                        _step_fwd(A, hs, inp[off:off + A] if fuse == "x" else None, None if fuse == "x" else xg[off:off + A],
                                  hprev[off:off + A], wi, bi, wh, bh, h_out, n_out, h, gates[off:off + A],
                                  Y, off * 2 * hs + direction * hs, 2 * hs, pl, seed, off * 2 * hs + direction * hs)
                else:
                    xg = gemm(inp, wi, trans_b=True, bias=b_ih.detach())
                    h = torch.zeros((R, hs), dtype=dtp, device=dev)
                    hg = torch.empty((R, 3 * hs), dtype=dtp, device=dev)
                    for t in steps:
                        A, off = batch_sizes[t], offs[t]
                        gemm(h[:A], wh, trans_b=True, bias=b_hh.detach(), out=hg[:A])
                        _cell_fwd(A, hs, xg[off:off + A], hg, h, Y, off * 2 * hs + direction * hs, 2 * hs,
                                  hprev[off:off + A], gates[off:off + A], pl, seed, off * 2 * hs + direction * hs)
                finals.append(h)
                layer_saved.append((weight_t(w_ih, wi), weight_t(w_hh, wh), gates, hprev))
            saved.append((inp, seed, pl, layer_saved))
            inp = Y
        ctx.cfg = (batch_sizes, offs, hs, num_layers, weights, saved)
        return torch.cat(finals, 1)

    @staticmethod
    def backward(ctx, d_out):
        batch_sizes, offs, hs, num_layers, weights, saved = ctx.cfg
        L, R, N = len(batch_sizes), batch_sizes[0], offs[-1]
        d_out = d_out.contiguous()
        dev = d_out.device
        grads = [None] * len(weights)
        used_side = False
        dY = None                                   # gradient w.r.t. this layer's (dropped) output [N, 2hs]
        for l in range(num_layers - 1, -1, -1):
            inp, seed, pl, layer_saved = saved[l]
            dtp = inp.dtype
            d_inp = None
            for direction in (0, 1):
                wi_t, wh_t, gates, hprev = layer_saved[direction]
                base = l * 8 + direction * 4
                w_ih, w_hh, b_ih, b_hh = weights[base:base + 4]
                steps = range(L - 1, -1, -1) if direction == 0 else range(L)
                want_bias = b_ih.requires_grad or b_hh.requires_grad
                fused = FUSE != "off" and dtp == torch.bfloat16 and hs % 64 == 0
                dh_dt = dtp if fused else torch.float32     # the fused step keeps the running state gradient in bf16
                if l == num_layers - 1:
                    dh = d_out[:, direction * hs:(direction + 1) * hs].to(dh_dt).contiguous()
                else:
                    dh = torch.zeros((R, hs), dtype=dh_dt, device=dev)
                if fused:
                    # d4 = [d r | d z | d n_x | d n_h]: d(xg) and d(hg) share their first two blocks, so ONE buffer serves both
                    d4 = torch.empty((N, 4 * hs), dtype=dtp, device=dev)
                    bpart = torch.zeros((N_BIAS_PARTIALS, 4 * hs), dtype=torch.float32, device=dev) if want_bias else None
                    prev = None
                    for t in steps:
                        A, off = batch_sizes[t], offs[t]
                        dyp = None if dY is None else dY.data_ptr() + (off * 2 * hs + direction * hs) * dY.element_size()
                        call("gtos_gru_step_bwd", A, hs, None if prev is None else ptr(d4[offs[prev]:]),
                             0 if prev is None else batch_sizes[prev], ptr(wh_t), ptr(gates[off:off + A]), ptr(hprev[off:off + A]),
                             dyp, 2 * hs, ptr(dh), dt(dh), ptr(d4[off:off + A]), float(pl), seed, off * 2 * hs + direction * hs,
                             ptr(bpart), N_BIAS_PARTIALS, stream())
                        prev = t
                    dxg = d4[:, :3 * hs]
                    w_jobs = ((w_hh, d4[:, :2 * hs], hprev, 1, slice(0, 2 * hs)), (w_hh, d4[:, 3 * hs:], hprev, 1, slice(2 * hs, 3 * hs)),
                              (w_ih, dxg, inp, 0, slice(0, 3 * hs)))
                else:
                    dxg = torch.empty((N, 3 * hs), dtype=dtp, device=dev)
                    dhg = torch.empty((N, 3 * hs), dtype=dtp, device=dev)
                    # bias gradients accumulate inside the cell kernel (per-block partial column sums) when the shape allows
                    fuse_bias = (256 % (hs // 8) == 0) and want_bias
                    bpart = torch.zeros((N_BIAS_PARTIALS, 4 * hs), dtype=torch.float32, device=dev) if fuse_bias else None
                    for t in steps:
                        A, off = batch_sizes[t], offs[t]
                        _cell_bwd(A, hs, gates[off:off + A], hprev[off:off + A], dY, off * 2 * hs + direction * hs, 2 * hs,
                                  dh, dxg[off:off + A], dhg[off:off + A], pl, seed, off * 2 * hs + direction * hs, bpart)
                        gemm(dhg[off:off + A], wh_t, trans_b=True, out=dh[:A], accumulate=True)   # dh += d(hg) W_hh
                    w_jobs = ((w_hh, dhg, hprev, 1, slice(0, 3 * hs)), (w_ih, dxg, inp, 0, slice(0, 3 * hs)))
                # parameter gradients over all steps at once -- on the side stream (see SIDE_STREAM), after this
                # direction's steps; gradient tensors that are not views of the flat bucket are allocated on the main
                # stream first so that the caching allocator ties them to the stream that will consume them
                for (wt, _, _, slot, _) in w_jobs:
                    if wt.requires_grad and _grad_target(wt) is None and grads[base + slot] is None:
                        grads[base + slot] = torch.zeros(wt.shape, dtype=torch.float32, device=dev)
                for (bt, slot) in ((b_hh, 3), (b_ih, 2)):
                    if bt.requires_grad and _grad_target(bt) is None:
                        grads[base + slot] = torch.zeros(bt.shape, dtype=torch.float32, device=dev)
                main = torch.cuda.current_stream(dev)
                side = _side_stream(dev) if (SIDE_STREAM and fused and N >= SIDE_MIN_ROWS) else main
                if side is not main:
                    side.wait_stream(main)
                    # locals that die (or are rebound) before the side stream is done with them: tell the allocator
                    for t_ in (d4, bpart, hprev, inp):
                        if t_ is not None:
                            t_.record_stream(side)
                with torch.cuda.stream(side):
                    for (wt, dyv, xin, slot, rows) in w_jobs:
                        if wt.requires_grad:
                            tgt = _grad_target(wt)
                            if tgt is None:
                                tgt = grads[base + slot]
                            nrow = rows.stop - rows.start
                            gemm(dyv, xin, trans_a=True, out=tgt[rows], accumulate=True, splitk=_splitk(nrow, wt.shape[1], N))
                    bsum = bpart.sum(0) if bpart is not None else None     # [4*hs]: d(r), d(z), d(n_x), d(n_h)
                    for (bt, slot) in ((b_hh, 3), (b_ih, 2)):
                        if bt.requires_grad:
                            tgt = _grad_target(bt)
                            if tgt is None:
                                tgt = grads[base + slot]
                            if bsum is not None:
                                tgt[:2 * hs] += bsum[:2 * hs]
                                tgt[2 * hs:] += bsum[2 * hs:3 * hs] if slot == 2 else bsum[3 * hs:]
                            else:
                                dyv = dhg if slot == 3 else dxg
                                call("gtos_colsum", dt(dyv), N, 3 * hs, 3 * hs, ptr(dyv), ptr(tgt), stream())
                used_side = used_side or side is not main
                if l > 0 or ctx.needs_input_grad[0]:
                    if d_inp is None:
                        d_inp = gemm(dxg, wi_t, trans_b=True)
                    else:
                        gemm(dxg, wi_t, trans_b=True, out=d_inp, accumulate=True)
            dY = d_inp
        if used_side:
            if all(gr is None for gr in grads):
                # every gradient of this function went into the flat bucket: nobody reads the side stream's results before
                # the optimizer / all-reduce, which join it (ops.join_side) -- the remaining GEMMs overlap what follows
                defer_side_join(dev)
            else:
                torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
        return (dY if ctx.needs_input_grad[0] else None, None, None, None, None) + tuple(grads)


def bigru_final(x_packed, batch_sizes, hs, num_layers, p_drop, weights):
    return BiGRUFinalFn.apply(x_packed, tuple(batch_sizes), hs, num_layers, float(p_drop), *weights)
