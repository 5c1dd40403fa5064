"""Bidirectional multi-layer GRU over packed, length-sorted relation label paths (final states only).

MI355X counterpart of ``nn.utils.rnn.pack_padded_sequence`` + ``nn.GRU`` as RelationEncoder uses them
(/root/reference/generator/encoder.py:93-111), with explicit BPTT.  bf16 (hidden size a multiple of 64): one fused MFMA
kernel per time step and direction, forward (gtos_gru_step_fwd: x W_ih^T and h W_hh^T gate products + cell; the state
is written straight into the next step's operand slot) and backward (gtos_gru_step_bwd: recurrent gradient product +
cell backward + bias sums, one d4 = [dr|dz|dn_x|dn_h] buffer).  fp32 / other sizes: the input-gate products of ALL
steps are one GEMM, each step one [active,h]x[h,3h] GEMM plus gtos_gru_cell_fwd / gtos_gru_cell_bwd.  Weight and
input gradients are GEMMs over all steps at once in both paths.

The production path (bf16, two layers) is TrieBiGRUFn at the end of this file: the same function evaluated on the prefix /
suffix tries of the relation bank (gtos_amd/pathtrie.py) -- layer 0 once per trie node, layer 1 with per-node input-gate tables
and a persistent step kernel, segmented-sum backward; BiGRUFinalFn (one row per path and position, like the reference's packed
sequence) remains the fp32 parity path and the GTOS_GRU_TRIE=0 fallback.
"""
import contextlib
import os

import torch

from . import _lib
from ._lib import call, dt, ptr, stream
from .ops import (gemm, compute_weight, next_seed, weight_t, _grad_target, _splitk, side_stream as _side_stream, side_ok, defer_side_join, _Timed,
                  embed_bwd_workspace, _workspace, WORKSPACE_BYTES, note_memory)


def _cell_fwd(A, hs, xg, hg, h, y, y_off_elems, ldy, hprev, gates, p, seed, drop_base):
    yp = None if y is None else y.data_ptr() + y_off_elems * y.element_size()
    call("gtos_gru_cell_fwd", dt(xg), A, hs, ptr(xg), ptr(hg), ptr(h), yp, ldy, ptr(hprev), ptr(gates),
         float(p), seed, drop_base, stream())


N_BIAS_PARTIALS = 1024

# Forward time step on bf16: "x" = input AND recurrent gate products + cell in ONE kernel (gtos_gru_step_fwd),
# "h" = recurrent product + cell fused, input gates by one big GEMM, "off" = GEMM + cell kernel per step (the fp32 path).
FUSE = "x"                       # (a module constant: the parity tests run the lower fusion levels through monkeypatch)
# Backward: run the weight-gradient GEMMs of one direction on a second HIP stream while the (memory-bound) BPTT steps of
# the next direction occupy the main stream: the step kernels are limited to 2 waves per SIMD by registers, an MFMA GEMM
# wave fits beside them.
SIDE_STREAM = os.environ.get("GTOS_GRU_SIDE", "1") != "0"
SIDE_MIN_ROWS = 200000          # below this the GEMMs are launch-bound and the stream hand-over costs more than it hides
# (Round 4 measured GTOS_GRU_RECOMPUTE_HN -- the forward not storing hn = W_hn h + b_hn, the backward step rebuilding it on the MFMA: layer-1
# forward step 419 -> 402 us per launch, backward step 487 -> 536 us, training step 61.3 -> 62.1 ms.  The saved bytes are worth less than four
# more k tiles in a kernel at the chip's read + write ceiling; the switch and its kernel instantiation were removed in round 6.)


def _step_fwd(A, hs, x, xg, h_in, wi, b_ih, wh, b_hh, h_out, n_out, h_fin, gates, y, y_off_elems, ldy, p, seed, drop_base,
              h_idx=None, gf=None, gf_idx=None, gb=None, gb_idx=None, fin_idx=None, tag=None):
    """``h_fin``: [*, hs] or a column block of a wider matrix (its row stride is passed on); ``fin_idx``: int32 row map of the
    finished rows (packed row m -> row fin_idx[m] of h_fin)."""
    yp = None if y is None else y.data_ptr() + y_off_elems * y.element_size()
    need_bi = x is not None or gf is not None
    # ``tag`` (bench.py's per-kernel rows): the span is recorded under that name with its ALGORITHMIC bytes as units -- per active row the
    # input row, the entering state, the four saved gate blocks, the new state and, for a layer with a successor, its dropped copy
    name = tag or "gru_step_fwd_%s" % ("x" if x is not None else ("tables" if gf is not None else "xg"))
    units = A if tag is None else A * 2 * (x.shape[1] + hs + 4 * hs + hs + (hs if y is not None else 0))
    with _Timed(name, detail=True, units=units):
        _step_fwd_call(A, hs, x, xg, h_in, wi, b_ih, wh, b_hh, h_out, n_out, h_fin, gates, yp, ldy, p, seed, drop_base,
                       h_idx, gf, gf_idx, gb, gb_idx, need_bi, fin_idx)


def _step_fwd_call(A, hs, x, xg, h_in, wi, b_ih, wh, b_hh, h_out, n_out, h_fin, gates, yp, ldy, p, seed, drop_base,
                   h_idx, gf, gf_idx, gb, gb_idx, need_bi, fin_idx=None):
    call("gtos_gru_step_fwd", A, hs, ptr(x), 0 if x is None else x.stride(0), 0 if x is None else x.shape[1],
         ptr(wi) if x is not None else None, ptr(b_ih) if need_bi else None, ptr(xg),
         ptr(gf), ptr(gf_idx), ptr(gb), ptr(gb_idx), ptr(h_in), ptr(h_idx), ptr(wh), ptr(b_hh),
         ptr(h_out), n_out, ptr(h_fin), hs if h_fin is None else h_fin.stride(0), ptr(fin_idx), ptr(gates), yp, ldy,
         float(p), seed, drop_base, stream())


def _step_bwd(A, hs, d4_prev, rows_prev, wh_t, gates, hprev, dy_ptr, ldy, dh, d4, p, seed, drop_base, bpart, hprev_idx=None, hp_out=None,
              sum_idx=None, dh_src=None, zero_row=-1):
    """``dh``: [rows, hs] or a column block of a wider matrix (row stride passed on); ``hp_out``: [rows, hs] receiving the
    (gathered) entering state rows; ``sum_idx`` / ``dh_src``: per-row source rows of the recurrent operand (in d4_prev) and of the
    incoming state gradient (in dh_src) -- the trie's children-sum indirection."""
    with _Timed("gru_step_bwd_%s" % ("trie" if hprev_idx is not None else "rows"), detail=True, units=A):
        call("gtos_gru_step_bwd", A, hs, ptr(d4_prev), rows_prev if d4_prev is not None else 0, ptr(wh_t), ptr(gates), ptr(hprev),
             ptr(hprev_idx), dy_ptr, ldy, ptr(dh), dt(dh), dh.stride(0), ptr(d4), float(p), seed, drop_base,
             ptr(bpart), N_BIAS_PARTIALS if bpart is not None else 0, ptr(hp_out), ptr(sum_idx), ptr(dh_src), int(zero_row), stream())


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
        with _Timed("relation_gru_bwd"):
            return BiGRUFinalFn._backward(ctx, d_out)

    @staticmethod
    def _backward(ctx, d_out):
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
                        _step_bwd(A, hs, None if prev is None else d4[offs[prev]:], 0 if prev is None else batch_sizes[prev], wh_t,
                                  gates[off:off + A], hprev[off:off + A], dyp, 2 * hs, dh, d4[off:off + A], pl, seed,
                                  off * 2 * hs + direction * hs, bpart)
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
                side = _side_stream(dev) if (SIDE_STREAM and side_ok(dev) and fused and N >= SIDE_MIN_ROWS) else main
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


# =====================================================================================================================
# The reference's TRAINING-mode function at production size (round 5): masks per (path, position, channel), hence one GRU row per
# (path, position) in both layers -- evaluated with no host read (the sorted order and the step sizes come with the batch), the
# layers' input gradients inside the backward step launches (gtos_gru_step_bwd_fused role B), one grouped product per layer and
# direction for both weight gradients (gtos_gru_weight_grads) and the label-embedding gradient as a one-hot product.
class PackPlan(object):
    """Packed time-major layout of a bank's paths sorted by decreasing length (generator/encoder.py:93-98 builds the same thing with
    sort + pack_padded_sequence): step t holds rows offs[t] .. offs[t] + batch_sizes[t], row m of a step is sorted path m, which is bank
    column order[m].  batch_sizes are host integers (launch geometry); order / offs live on the device."""

    def __init__(self, batch_sizes, order32, order64=None):
        self.batch_sizes = [int(a) for a in batch_sizes if a > 0]
        self.offs = [0]
        for a in self.batch_sizes:
            self.offs.append(self.offs[-1] + a)
        self.L, self.N = len(self.batch_sizes), self.offs[-1]
        self.order32 = order32
        self.order64 = order64 if order64 is not None else order32.to(torch.int64)
        self.offs_dev = torch.tensor(self.offs, dtype=torch.int32).to(order32.device, non_blocking=True)

    @staticmethod
    def of_trie(trie):
        """The plan a batch's PathTrie carries (built by the loader with the bank); cached on the trie."""
        plan = getattr(trie, "_pack_plan", None)
        if plan is None or plan.order32.device != trie.seq_order32.device:
            plan = trie._pack_plan = PackPlan(trie.batch_sizes, trie.seq_order32, trie.seq_order)
        return plan

    @staticmethod
    def of_lengths(lengths, max_len):
        """From the lengths alone (a caller feeding the reference's own batch): one sort and ONE host read of the step sizes."""
        sorted_len, order = torch.sort(lengths, descending=True, stable=True)
        bs = (sorted_len.unsqueeze(0) > torch.arange(max_len, device=lengths.device).unsqueeze(1)).sum(1).tolist()
        return PackPlan(bs, order.to(torch.int32), order)


def _step_bwd_fused(A, hs, d4_prev, rows_prev, wh_t, gates, hprev, dy_ptr, ldy, dh, d4, p, seed, drop_base, bpart,
                    wi_t=None, dinp=None, n_in=0, dinp_acc=False, p_in=0.0, seed_in=0, in_drop_base=0, tag="gru_step_bwd_packed"):
    """gtos_gru_step_bwd_fused: the backward step of ``A`` rows (A == 0: none) plus, with ``dinp``, the input gradient of the
    ``rows_prev`` rows of the step processed just before (role B workgroups of the same launch).  The span's units are the launch's
    ALGORITHMIC bytes: per active row gates 4h + entering state h read, state gradient h read + written, d4 4h written, dy h read when
    there is one; of the previous step's d4 the blocks its consumers need, each once (3h for the recurrent product alone, all 4h where
    the input-gradient tiles read the same rows); dinp written (and read first when it accumulates)."""
    rp = rows_prev if d4_prev is not None else 0
    both = min(A, rp) if dinp is not None else 0
    nbytes = 2 * (A * (4 * hs + hs + 2 * hs + 4 * hs + (hs if dy_ptr else 0)) + both * 4 * hs
                  + ((rp - both) * 3 * hs if dinp is not None else min(A, rp) * 3 * hs) + (rp * n_in * (2 if dinp_acc else 1) if dinp is not None else 0))
    with _Timed(tag, detail=True, units=nbytes):
        call("gtos_gru_step_bwd_fused", A, hs, ptr(d4_prev), rows_prev if d4_prev is not None else 0, ptr(wh_t), ptr(gates), ptr(hprev), None,
             dy_ptr, ldy, ptr(dh), 0 if dh is None else dt(dh), hs if dh is None else dh.stride(0), ptr(d4), float(p), seed, drop_base,
             ptr(bpart), N_BIAS_PARTIALS if bpart is not None else 0, None, None, None, -1,
             ptr(wi_t), ptr(dinp), 0 if dinp is None else dinp.stride(0), n_in, int(dinp_acc), float(p_in), seed_in, in_drop_base, stream())


# (Round 5 measured, same box each, and round 6 removed the switches: the layers' input gradients as one GEMM per direction instead of role-B
# tiles of the backward step launches 84.25 vs 81.70 ms per step; three weight-gradient GEMMs per (layer, direction) instead of the grouped
# product 82.84 vs 81.70; the forward's direction 1 on the auxiliary stream beside direction 0 84.9-86.7 vs 84.5-84.7: profiles/r5_ab_switches.txt.)
# packed path, backward: d4 written over the saved gates -- "1" always, "0" never, default: when the buffer is at least D4_INPLACE_MIN_BYTES
# (measured, call 26: C2 -- 5 GB per buffer -- 80.7 vs 80.3 ms per step in place, reserved 64 vs 71 GB; C5 -- 18 GB -- 275.7 vs 276.5-277.2 ms,
# reserved 145 vs 161-163 GB: worth it where memory is what is short)
D4_INPLACE = os.environ.get("GTOS_GRU_D4_INPLACE", "auto")
D4_INPLACE_MIN_BYTES = 8 << 30


class PackedPathGRUFn(torch.autograd.Function):
    """(bank [L,R] int64, PackPlan, label-embedding table [V,dim], weights of a 2-layer bidirectional GRU) -> [R, 2*hs] final states of
    the top layer in BANK order.  bf16, hs % 64 == 0, dim_pad % 64 == 0; weights as in BiGRUFinalFn (nn.GRU's own tensors: layer 0's
    w_ih is [3hs, dim], padded to dim_pad columns in here, so every gradient lands in its parameter's own ``.grad``).  Dropout p_embed on
    the embedded labels, p_layer between the layers, both per (path, position, channel)."""

    @staticmethod
    def forward(ctx, bank, plan, table, dim_pad, p_embed, hs, p_layer, *weights):
        dev, dtp = table.device, torch.bfloat16
        bs, offs, L, N = plan.batch_sizes, plan.offs, plan.L, plan.N
        R = bank.shape[1]
        if L == 0 or R == 0:                               # no active step (every path empty / no path): the final states are zero vectors
            ctx.cfg = False
            return torch.zeros((R, 2 * hs), dtype=dtp, device=dev)
        V, dim = table.shape
        tab = table.detach()
        # (a custom Function's forward runs under no_grad: whether anybody will call backward is what needs_input_grad says)
        want_table = table.requires_grad and ctx.needs_input_grad[2]
        Vp = (V + 7) // 8 * 8
        X = torch.empty((N, dim_pad), dtype=dtp, device=dev)
        onehot = torch.empty((N, Vp), dtype=dtp, device=dev) if (want_table and V <= 256) else None
        tokens = torch.empty((N,), dtype=torch.int64, device=dev) if (want_table and V > 256) else None
        seed_e = next_seed() if p_embed > 0 else 0            # the seeds in the order BiGRUFinalFn's caller draws them: embedding, layer 0
        call("gtos_embed_packed_paths", dt(X), L, R, N, ptr(bank), ptr(plan.order32), ptr(plan.offs_dev), ptr(tab), dim, dim_pad, ptr(X),
             float(p_embed), seed_e, ptr(onehot), Vp, ptr(tokens), stream())
        fin = (torch.empty if bs[0] == R else torch.zeros)((R, 2 * hs), dtype=dtp, device=dev)      # (an empty path keeps a zero vector)
        park = [torch.empty((bs[0], hs), dtype=dtp, device=dev) for _ in (0, 1)]   # where layer 0's finished rows land (nobody reads them)
        inp, saved = X, []
        # (The two directions of a layer are independent; running direction 1 on the auxiliary stream beside direction 0 was measured in
        # round 5 and bought nothing -- a launch costs its k loop PLUS its cell however the launches are interleaved)
        for l in range(2):
            last = l == 1
            Y = None if last else torch.empty((N, 2 * hs), dtype=dtp, device=dev)
            seed = next_seed() if (p_layer > 0 and not last) else 0
            pl = p_layer if not last else 0.0
            layer_saved = []
            bufs = [(torch.empty((N, 4 * hs), dtype=dtp, device=dev), torch.empty((N, hs), dtype=dtp, device=dev)) for _ in (0, 1)]
            wts = []
            for direction in (0, 1):
                w_ih, w_hh, b_ih, b_hh = weights[l * 8 + direction * 4: l * 8 + direction * 4 + 4]
                wi, wh = compute_weight(w_ih, dtp), compute_weight(w_hh, dtp)
                if wi.shape[1] != inp.shape[1]:             # layer 0: the label width (100) padded to the k tile
                    wi = torch.nn.functional.pad(wi.detach(), (0, inp.shape[1] - wi.shape[1]))
                    wi_t = wi.t().contiguous()
                else:
                    wi_t = weight_t(w_ih, wi)
                wts.append((wi, wh, wi_t, weight_t(w_hh, wh), b_ih.detach(), b_hh.detach()))
            for direction in (0, 1):
                wi, wh, wi_t, wh_t, bi, bh = wts[direction]
                gates, hprev = bufs[direction]
                if direction == 0:
                    hprev[:bs[0]].zero_()
                else:                                   # rows that become active at step t start from h = 0
                    for t in range(L):
                        lo = bs[t + 1] if t + 1 < L else 0
                        if bs[t] > lo:
                            hprev[offs[t] + lo: offs[t] + bs[t]].zero_()
                h_fin = fin[:, direction * hs:(direction + 1) * hs] if last else park[direction]
                for t in (range(L) if direction == 0 else range(L - 1, -1, -1)):
                    A, off = bs[t], offs[t]
                    nxt = t + 1 if direction == 0 else t - 1
                    if 0 <= nxt < L:
                        h_out, n_out = hprev[offs[nxt]:], min(A, bs[nxt])
                    else:
                        h_out, n_out = None, 0
                    _step_fwd(A, hs, inp[off:off + A], None, hprev[off:off + A], wi, bi, wh, bh, h_out, n_out, h_fin, gates[off:off + A],
                              Y, off * 2 * hs + direction * hs, 2 * hs, pl, seed, off * 2 * hs + direction * hs,
                              fin_idx=plan.order32 if last else None, tag="gru_step_fwd_packed_l%d" % l)
                layer_saved.append((wi_t, wh_t, gates, hprev))
            saved.append((inp, seed, pl, layer_saved))
            inp = Y
        ctx.cfg = (plan, table, dim_pad, p_embed, seed_e, hs, weights, saved, onehot, tokens)
        return fin

    @staticmethod
    def backward(ctx, d_out):
        with _Timed("relation_gru_bwd"):
            return PackedPathGRUFn._backward(ctx, d_out)

    @staticmethod
    def _backward(ctx, d_out):
        if ctx.cfg is False:                               # the empty plan: nothing depends on any input
            return (None,) * len(ctx.needs_input_grad)
        if ctx.cfg is None:
            raise RuntimeError("PackedPathGRUFn: backward called a second time -- the saved gates and states are released piece by piece "
                               "during the first (retain_graph is not supported on this path)")
        plan, table, dim_pad, p_embed, seed_e, hs, weights, saved, onehot, tokens = ctx.cfg
        bs, offs, L, N = plan.batch_sizes, plan.offs, plan.L, plan.N
        dev, dtp = d_out.device, torch.bfloat16
        # bank order -> packed order, one gather; the two column blocks are the running state gradients of the two directions
        dfin = d_out.to(dtp).index_select(0, plan.order64)
        grads = [None] * len(weights)
        for base in (8, 12, 0, 4):                    # gradient tensors that are not views of the flat bucket: created on the main stream
            for slot in range(4):
                wt_ = weights[base + slot]
                if wt_.requires_grad and _grad_target(wt_) is None:
                    grads[base + slot] = torch.zeros(wt_.shape, dtype=torch.float32, device=dev)
        want_table = onehot is not None or tokens is not None
        main = torch.cuda.current_stream(dev)
        used_side, held = False, None
        dY = None                                     # d(loss) / d(layer 0 output, after its dropout), [N, 2hs]
        dX = None
        saved = [list(e) for e in saved]              # (released piece by piece below: at C5 a direction's gates alone are 22 GB)
        ctx.cfg = None
        for l in (1, 0):
            inp, seed, pl, layer_saved = saved[l]
            saved[l] = None
            layer_saved = list(layer_saved)
            n_in = inp.shape[1]
            want_dinp = l == 1 or want_table
            d_in = torch.empty((N, n_in), dtype=dtp, device=dev) if want_dinp else None
            for direction in (0, 1):
                wi_t, wh_t, gates, hprev = layer_saved[direction]
                layer_saved[direction] = None
                base = l * 8 + direction * 4
                w_ih, w_hh, b_ih, b_hh = weights[base:base + 4]
                want_bias = b_ih.requires_grad or b_hh.requires_grad
                dh = dfin[:, direction * hs:(direction + 1) * hs] if l == 1 else torch.zeros((bs[0], hs), dtype=dtp, device=dev)
                # d4 = [d r | d z | d n_x | d n_h] of every packed row, written by the cell tiles: IN PLACE over the saved gates (same shape; a
                # lane reads the four gate values of its (row, channels) and writes the four gradients to the same addresses, the rows of other
                # steps are either still gates -- not yet processed -- or already gradients -- what the next launch's products read): one
                # [N, 4hs] buffer less while a direction runs (C5: 18 GB; reserved memory 145 instead of 161-163 GB).  See D4_INPLACE above.
                in_place = D4_INPLACE == "1" or (D4_INPLACE != "0" and gates.numel() * gates.element_size() >= D4_INPLACE_MIN_BYTES)
                d4 = gates if in_place else torch.empty((N, 4 * hs), dtype=dtp, device=dev)
                note_memory(dev)
                bpart = torch.zeros((N_BIAS_PARTIALS, 4 * hs), dtype=torch.float32, device=dev) if want_bias else None
                rb = dict(wi_t=wi_t, n_in=n_in, dinp_acc=direction == 1, p_in=p_embed if l == 0 else 0.0,
                          seed_in=seed_e if l == 0 else 0) if want_dinp else None
                tag = "gru_step_bwd_packed_l%d" % l
                prev = None
                for t in (range(L - 1, -1, -1) if direction == 0 else range(L)):
                    A, off = bs[t], offs[t]
                    dyp = None if dY is None else dY.data_ptr() + (off * 2 * hs + direction * hs) * dY.element_size()
                    kw = {} if (rb is None or prev is None) else dict(rb, dinp=d_in[offs[prev]:], in_drop_base=offs[prev] * n_in)
                    _step_bwd_fused(A, hs, None if prev is None else d4[offs[prev]:], 0 if prev is None else bs[prev], wh_t,
                                    gates[off:off + A], hprev[off:off + A], dyp, 2 * hs, dh, d4[off:off + A], pl, seed,
                                    off * 2 * hs + direction * hs, bpart, tag=tag, **kw)
                    prev = t
                if rb is not None:                    # the input gradient of the step processed last: role B workgroups only
                    _step_bwd_fused(0, hs, d4[offs[prev]:], bs[prev], wh_t, None, None, None, 2 * hs, None, None, 0.0, 0, 0, None, tag=tag,
                                    **dict(rb, dinp=d_in[offs[prev]:], in_drop_base=offs[prev] * n_in))
                del gates                              # the steps were their last reader
                # Parameter gradients over all steps at once, on the auxiliary stream beside the NEXT direction's steps.  No record_stream:
                # a block freed on one stream while another still reads it can only be recycled once the device has passed the free, and
                # with no host read left in a step the host runs steps ahead of the device -- every such block (5 GB of d4 per direction at
                # C2) then sits in limbo while the allocator mallocs new ones (237 GB reserved for 44 GB allocated, measured).  Instead the
                # operands of a direction's products stay referenced (``held``) until the main stream has waited for the auxiliary one,
                # one direction later.
                side = _side_stream(dev) if (SIDE_STREAM and side_ok(dev) and N >= SIDE_MIN_ROWS) else main
                if held is not None:
                    main.wait_stream(_side_stream(dev))          # the previous direction's products ran beside the steps just queued
                    held = None
                if side is not main:
                    side.wait_stream(main)
                    held = (d4, bpart, hprev, inp)
                with torch.cuda.stream(side):
                    tg_ih = _grad_target(w_ih) if w_ih.requires_grad else None
                    tg_hh = _grad_target(w_hh) if w_hh.requires_grad else None
                    tg_ih = grads[base] if (tg_ih is None and w_ih.requires_grad) else tg_ih
                    tg_hh = grads[base + 1] if (tg_hh is None and w_hh.requires_grad) else tg_hh
                    # both weight gradients of the (layer, direction) as ONE grouped product d4^T [x | h_prev]; a frozen weight gets a scratch target
                    scr_ih = torch.zeros(w_ih.shape, dtype=torch.float32, device=dev) if tg_ih is None else tg_ih
                    scr_hh = torch.zeros(w_hh.shape, dtype=torch.float32, device=dev) if tg_hh is None else tg_hh
                    if (tg_ih is not None or tg_hh is not None) and w_ih.shape[1] % 4 == 0:
                        with _Timed("gru_dw_grouped_l%d" % l, detail=True, units=2 * N * 3 * hs * (w_ih.shape[1] + hs)):      # (units: useful flops)
                            ws = _workspace(dev)
                            call("gtos_gru_weight_grads", N, hs, n_in, w_ih.shape[1], ptr(d4), ptr(inp), inp.stride(0), ptr(hprev), hprev.stride(0),
                                 ptr(scr_ih), scr_ih.stride(0), ptr(scr_hh), scr_hh.stride(0), ptr(ws), ws.numel() * 4, stream())
                    elif tg_ih is not None or tg_hh is not None:               # a label width that is no multiple of 4: per-matrix products
                        for (tg, dyv, xin, rows) in ((tg_hh, d4[:, :2 * hs], hprev, slice(0, 2 * hs)), (tg_hh, d4[:, 3 * hs:], hprev, slice(2 * hs, 3 * hs)),
                                                     (tg_ih, d4[:, :3 * hs], inp, slice(0, 3 * hs))):
                            if tg is None:
                                continue
                            sk = _splitk(rows.stop - rows.start, xin.shape[1], N)
                            if tg.shape[1] == xin.shape[1]:
                                gemm(dyv, xin, trans_a=True, out=tg[rows], accumulate=True, splitk=sk)
                            else:                      # layer 0's input is wider (zero-padded) than its weight
                                tg[rows] += gemm(dyv, xin, trans_a=True, out_dtype=torch.float32, splitk=sk)[:, :tg.shape[1]]
                    if want_bias:
                        _acc_bias_grads(grads, base, b_ih, b_hh, bpart.sum(0), hs)
                used_side = used_side or side is not main
                del d4, hprev, bpart                   # (``held`` keeps what the auxiliary stream still reads)
            del inp
            if l == 1:
                dY = d_in
            else:
                dX, dY = d_in, None
        dtab = None
        if want_table:
            tgt = _grad_target(table)
            if tgt is None:
                tgt = dtab = torch.zeros(table.shape, dtype=torch.float32, device=dev)
            V, dim = table.shape
            if onehot is not None:
                # label-embedding gradient = onehot(token)^T dX (dX already carries the forward's dropout mask): one product on the MFMA GEMM
                part = gemm(onehot, dX, trans_a=True, out_dtype=torch.float32, splitk=_splitk(onehot.shape[1], dim_pad, N))
                tgt += part[:V, :dim]
            else:
                ws = embed_bwd_workspace(N, V, dim_pad, dev)
                call("gtos_embed_rows_bwd", dt(dX), N, V, dim, dim_pad, ptr(tokens), ptr(dX), ptr(tgt), 0.0, 0, ptr(ws),
                     0 if ws is None else ws.numel() * 4, stream())
        if used_side:
            if dtab is None and all(gr is None for gr in grads):
                # every gradient went into the flat bucket: its readers join the side stream (ops.join_side), which also releases the
                # last direction's operands
                defer_side_join(dev, list(t_ for t_ in (held or ()) if t_ is not None))
            else:
                main.wait_stream(_side_stream(dev))
        return (None, None, dtab, None, None, None, None) + tuple(grads)


def packed_path_gru(bank, plan, table, dim_pad, p_embed, hs, p_layer, weights):
    return PackedPathGRUFn.apply(bank, plan, table, dim_pad, float(p_embed), hs, float(p_layer), *weights)


def bigru_final(x_packed, batch_sizes, hs, num_layers, p_drop, weights):
    return BiGRUFinalFn.apply(x_packed, tuple(batch_sizes), hs, num_layers, float(p_drop), *weights)


# =====================================================================================================================
# Trie-evaluated 2-layer bi-GRU (bf16).  Same function of its inputs as BiGRUFinalFn at dropout 0 / in eval mode:
#   * layer 0, forward direction: the state after tokens 0..t of a path depends only on that PREFIX, so it is computed once
#     per node of the prefix trie (level k = all distinct prefixes of k+1 tokens, one fused step launch per level, the parent
#     state gathered by index); reverse direction: once per node of the SUFFIX trie.  The relation bank of a batch is
#     (almost) prefix/suffix closed -- ~R nodes per trie instead of sum(len) = 5.7 R rows (csrc_host/pathtrie.cpp);
#   * layer 1 (needs every (path, position) row: its input mixes a prefix state and a suffix state): the input-gate product
#     x W_ih^T with x = [y0f(prefix node) ; y0b(suffix node)] splits into Gf[prefix node] + Gb[suffix node] with
#     Gf = Y0f W_ih[:, :h]^T, Gb = Y0b W_ih[:, h:]^T computed once per trie node; the step kernel gathers the two rows.
#   * backward mirrors it: the gradient of a shared node is the segmented sum over the rows / children that share it
#     (gtos_segment_sum_*), then ordinary GEMMs on ~R rows.
# With dropout > 0 (training) the embedding dropout and the inter-layer dropout masks are drawn per TRIE NODE and channel:
# every path still sees independent Bernoulli(1-p) masks at each of its positions, exactly the reference's per-path
# distribution; what changes is that two paths sharing a prefix (suffix) share the mask on the shared part, and that the two
# directions draw separate embedding masks.  GTOS_GRU_TRIE=0 selects the per-row path above.
TRIE = os.environ.get("GTOS_GRU_TRIE", "1") != "0"
# Trie backward: its GEMMs on the auxiliary stream beside the BPTT steps.  Measured at C2: no gain (73.6 vs 73.7 ms/step -- the
# step kernels slow down by what the GEMMs gain once the per-row work is gone), so it is off by default and the profile stays
# one kernel at a time.
TRIE_SIDE = False                # (module constant; tests/test_hip_parity.py runs both settings through monkeypatch)
# Layer 0 walks each trie level by level (<= 8 small launches per trie, the first levels far too small to fill the chip); the
# prefix and the suffix trie are independent, so the suffix side runs on the auxiliary stream beside the prefix side.
TRIE_L0_OVERLAP = True
# (Layer 1, forward: the input-gate table products of the reverse direction on the auxiliary stream beside the forward direction's steps were
# measured at C2 in round 3: 64.28 vs 64.29 ms per step; removed in round 6.)
# Label-embedding gradient of the trie path as two GEMMs (one-hot product) instead of the LDS-atomics scatter kernel.
EMBED_GRAD_GEMM = True           # (False: 63.29 vs 62.8 ms per step, round 3)
# Layer 0, backward: children -> parent sums through the row indirection of pathtrie.TrieSide.sum_idx (0: a summed row per node).
TRIE_SUM_INDEX = True            # (False: 64.65 vs 64.29 ms per step, round 3)


# Segmented sums of the gate gradients by the streaming kernel (a wave per range of ~256 rows) instead of a wave per chunk.
SEG_STREAM = True                # (False: 1,142-1,156 vs 1,077-1,094 us per launch, round 3; tests and tools/bench_segsum.py flip it)


def _seg_rows(side, src, width, dst, src2=None, dst2=None):
    """dst[node] = sum of src rows of the node (row lists of the trie side), fp32 accumulation; (src2, dst2): a second
    matrix reduced over the same row lists in the same pass."""
    heavy = torch.zeros((2 if src2 is not None else 1, max(1, side.n_heavy), width), dtype=torch.float32, device=src.device)
    if SEG_STREAM and src2 is None and width % 256 == 0:
        with _Timed("segment_sum_rows", detail=True, units=int(side.rows.numel())):
            call("gtos_segment_sum_stream", side.n_chunks, int(side.rows.numel()), ptr(side.rows), ptr(side.chunk_node), ptr(side.chunk_start),
                 ptr(side.chunk_cnt), ptr(side.chunk_slot), ptr(side.wave_off), side.n_waves, ptr(src), src.stride(0), width, ptr(dst),
                 dst.stride(0), ptr(heavy[0]), stream())
            call("gtos_segment_sum_finish", side.n_heavy, ptr(side.heavy_node), ptr(heavy[0]), width, ptr(dst), dst.stride(0), stream())
        return
    with _Timed("segment_sum_rows", detail=True, units=int(side.rows.numel()) * (2 if src2 is not None else 1)):
        call("gtos_segment_sum_rows", side.n_chunks, ptr(side.rows), ptr(side.chunk_node), ptr(side.chunk_start), ptr(side.chunk_cnt),
             ptr(side.chunk_slot), ptr(src), ptr(src2), src.stride(0), width, ptr(dst), ptr(dst2), dst.stride(0),
             ptr(heavy[0]), ptr(heavy[1]) if src2 is not None else None, stream())
        call("gtos_segment_sum_finish", side.n_heavy, ptr(side.heavy_node), ptr(heavy[0]), width, ptr(dst), dst.stride(0), stream())
        if src2 is not None:
            call("gtos_segment_sum_finish", side.n_heavy, ptr(side.heavy_node), ptr(heavy[1]), width, ptr(dst2), dst2.stride(0), stream())


def _seg_ranges(n_seg, ranges, src, width, dst):
    call("gtos_segment_sum_ranges", n_seg, ptr(ranges), ptr(src), src.stride(0), width, ptr(dst), dst.stride(0), stream())


def _acc_weight_grad(grads, slot, wt, dy, x, rows=None, cols=None):
    """(flat bucket view or grads[slot])[rows, cols] += dy^T x"""
    if not wt.requires_grad:
        return
    tgt = _grad_target(wt)
    if tgt is None:
        if grads[slot] is None:
            grads[slot] = torch.zeros(wt.shape, dtype=torch.float32, device=dy.device)
        tgt = grads[slot]
    if rows is not None:
        tgt = tgt[rows]
    if cols is not None:
        tgt = tgt[:, cols]
    gemm(dy, x, trans_a=True, out=tgt, accumulate=True, splitk=_splitk(dy.shape[1], x.shape[1], dy.shape[0]))


def _acc_bias_grads(grads, base, b_ih, b_hh, bsum, hs):
    for bt, slot in ((b_hh, 3), (b_ih, 2)):
        if not bt.requires_grad:
            continue
        tgt = _grad_target(bt)
        if tgt is None:
            if grads[base + slot] is None:
                grads[base + slot] = torch.zeros(bt.shape, dtype=torch.float32, device=bsum.device)
            tgt = grads[base + slot]
        tgt[:2 * hs] += bsum[:2 * hs]
        tgt[2 * hs:] += bsum[2 * hs:3 * hs] if slot == 2 else bsum[3 * hs:]


class TrieBiGRUFn(torch.autograd.Function):
    """(trie, embedding table [V,dim], weights of a 2-layer bidirectional GRU) -> [R, 2*hs] final states of the top layer
    in BANK order (row s = path s of relation_bank).  weights as in BiGRUFinalFn; layer 0's w_ih zero-padded to dim_pad columns."""

    @staticmethod
    def forward(ctx, trie, table, dim_pad, p_embed, hs, p_layer, *weights):
        dev, dtp = table.device, torch.bfloat16
        L, R, N = trie.L, trie.R, trie.N
        bs = trie.batch_sizes
        offs = [0]
        for a in bs:
            offs.append(offs[-1] + a)
        sides = (trie.pf, trie.sf)
        dim = table.shape[1]
        tab = table.detach()
        # ---- layer 0 on the tries (suffix side on the auxiliary stream, see TRIE_L0_OVERLAP)
        l0 = []
        main = torch.cuda.current_stream(dev)
        aux = _side_stream(dev) if (TRIE_L0_OVERLAP and table.is_cuda and side_ok(dev) and N >= SIDE_MIN_ROWS) else main
        seeds = [(next_seed() if p_embed > 0 else 0, next_seed() if p_layer > 0 else 0) for _ in sides]
        if aux is not main:
            aux.wait_stream(main)
        for d, side in enumerate(sides):
            with torch.cuda.stream(aux if d == 1 else main):
                n = side.n_nodes
                w_ih, w_hh, b_ih, b_hh = weights[d * 4: d * 4 + 4]
                wi, wh = compute_weight(w_ih, dtp), compute_weight(w_hh, dtp)
                seed_e, seed_y = seeds[d]
                X = torch.empty((n, dim_pad), dtype=dtp, device=dev)
                call("gtos_embed_rows_fwd", dt(X), n, dim, dim_pad, ptr(side.tok), ptr(tab), ptr(X), float(p_embed), seed_e, stream())
                H = torch.empty((n + 1, hs), dtype=dtp, device=dev)
                H[n].zero_()                                        # the state every level-0 node starts from
                gates = torch.empty((n, 4 * hs), dtype=dtp, device=dev)
                Y = torch.empty((n, hs), dtype=dtp, device=dev) if p_layer > 0 else None
                bi, bh = b_ih.detach(), b_hh.detach()
                for k in range(L):
                    lo, hi = side.level_off[k], side.level_off[k + 1]
                    if hi > lo:
                        _step_fwd(hi - lo, hs, X[lo:hi], None, H, wi, bi, wh, bh, H[lo:hi], hi - lo, None, gates[lo:hi],
                                  Y, lo * hs, hs, p_layer, seed_y, lo * hs, h_idx=side.par[lo:hi])
                l0.append((X, H, gates, Y, seed_e, seed_y, weight_t(w_ih, wi), weight_t(w_hh, wh)))
        if aux is not main:
            main.wait_stream(aux)      # layer 1 reads both sides; the suffix side's buffers live in the auxiliary stream's pool and
            #                            are only handed back after backward, when main has long passed this point
        src = [l0[d][3] if p_layer > 0 else l0[d][1][:sides[d].n_nodes] for d in (0, 1)]
        # ---- layer 1: per-node input-gate tables, then the recurrent steps over the packed rows.  A sequence's final state goes
        # straight to its BANK row of fin [R, 2hs] (column block d): no torch.cat of the directions, no unsort gather
        fin = torch.empty((R, 2 * hs), dtype=dtp, device=dev)
        l1 = []
        # input-gate tables of both directions first: direction 0's on the main stream, direction 1's on the auxiliary stream,
        # where the two MFMA-bound products run beside direction 0's (HBM-bound) recurrent steps.  Their buffers come from the
        # main stream's pool (the consumer's), the auxiliary stream only fills them.
        tables = []
        for d in (0, 1):
            wi = compute_weight(weights[8 + d * 4], dtp)
            Gf = torch.empty((sides[0].n_nodes, 3 * hs), dtype=dtp, device=dev)      # [nodes of the prefix trie, 3hs]
            Gb = torch.empty((sides[1].n_nodes, 3 * hs), dtype=dtp, device=dev)      # [nodes of the suffix trie, 3hs]
            gemm(src[0], wi[:, :hs], trans_b=True, out=Gf)
            gemm(src[1], wi[:, hs:], trans_b=True, out=Gb)
            tables.append((Gf, Gb))
        for d in (0, 1):
            w_ih, w_hh, b_ih, b_hh = weights[8 + d * 4: 8 + d * 4 + 4]
            wi, wh = compute_weight(w_ih, dtp), compute_weight(w_hh, dtp)
            Gf, Gb = tables[d]
            gates = torch.empty((N, 4 * hs), dtype=dtp, device=dev)
            hprev = torch.empty((N, hs), dtype=dtp, device=dev)
            h = fin[:, d * hs:(d + 1) * hs]
            bi, bh = b_ih.detach(), b_hh.detach()
            if d == 0:
                hprev[:R].zero_()
            else:                                               # rows that become active at step t start from h = 0
                for t in range(L):
                    lo = bs[t + 1] if t + 1 < L else 0
                    if bs[t] > lo:
                        hprev[offs[t] + lo: offs[t] + bs[t]].zero_()
            for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
                A, off = bs[t], offs[t]
                nxt = t + 1 if d == 0 else t - 1
                if 0 <= nxt < L:
                    h_out, n_out = hprev[offs[nxt]:], min(A, bs[nxt])
                else:
                    h_out, n_out = None, 0                      # the direction's last step: every row is finished
                _step_fwd(A, hs, None, None, hprev[off:off + A], None, bi, wh, bh, h_out, n_out, h, gates[off:off + A],
                          None, 0, hs, 0.0, 0, 0, gf=Gf, gf_idx=trie.row_pf[off:off + A], gb=Gb, gb_idx=trie.row_sf[off:off + A],
                          fin_idx=trie.seq_order32)
            l1.append((gates, hprev, wi, weight_t(w_hh, wh)))
        ctx.cfg = (trie, table, dim_pad, p_embed, hs, p_layer, weights, l0, l1, offs)
        return fin

    @staticmethod
    def backward(ctx, d_out):
        with _Timed("relation_gru_bwd"):
            return TrieBiGRUFn._backward(ctx, d_out)

    @staticmethod
    def _backward(ctx, d_out):
        trie, table, dim_pad, p_embed, hs, p_layer, weights, l0, l1, offs = ctx.cfg
        dev, dtp = d_out.device, torch.bfloat16
        L, R, N = trie.L, trie.R, trie.N
        bs = trie.batch_sizes
        sides = (trie.pf, trie.sf)
        # bank order -> packed order, one gather of the [R, 2hs] gradient; its two column blocks are the running state
        # gradients of the two directions (updated in place by the step kernels)
        d_out = d_out.to(dtp).index_select(0, trie.seq_order)
        grads = [None] * len(weights)
        src = [l0[d][3] if p_layer > 0 else l0[d][1][:sides[d].n_nodes] for d in (0, 1)]
        dsrc = [None, None]
        # ---- layer 1.  The BPTT steps (memory-bound, 2-3 workgroups per CU) stay on the main stream; every GEMM of this
        # function (weight gradients, gate-table input gradients) goes to the auxiliary stream and runs beside the steps /
        # segment sums that follow it.
        main = torch.cuda.current_stream(dev)
        aux = _side_stream(dev) if (TRIE_SIDE and SIDE_STREAM and side_ok(dev) and N >= SIDE_MIN_ROWS) else main
        use_side = aux is not main

        keep = []      # tensors the auxiliary stream reads: kept alive until main has waited for it (no record_stream: blocks
        #                recorded on a second stream are not reusable at the next step and the allocator falls back to hipMalloc)

        def on_side(*tensors):
            """Context for work that may run on the auxiliary stream once everything queued on main so far is done."""
            if use_side:
                aux.wait_stream(main)
                keep.extend(t_ for t_ in tensors if t_ is not None)
            return torch.cuda.stream(aux)
        # gradient tensors that are not views of the flat bucket are created on the main stream first
        for base in (8, 12, 0, 4):
            for slot in range(4):
                wt_ = weights[base + slot]
                if wt_.requires_grad and _grad_target(wt_) is None and grads[base + slot] is None:
                    grads[base + slot] = torch.zeros(wt_.shape, dtype=torch.float32, device=dev)
        # Per direction: BPTT steps -> recurrent weight gradient -> the gate-table gradients of BOTH tries (segmented sums over this
        # direction's d4), then d4, gates and the saved states of the direction are released before the other direction allocates its
        # own (round 4: at C5 a direction's d4 + gates + states are 49 GB; rounds 2-3 kept both directions' alive until the sums).
        dGs = [[None, None], [None, None]]                 # [trie side][direction]
        wis = [l1[0][2], l1[1][2]]
        for d in (0, 1):
            gates, hprev, wi, wh_t = l1[d]
            base = 8 + d * 4
            w_ih, w_hh, b_ih, b_hh = weights[base:base + 4]
            want_bias = b_ih.requires_grad or b_hh.requires_grad
            dh = d_out[:, d * hs:(d + 1) * hs]
            d4 = torch.empty((N, 4 * hs), dtype=dtp, device=dev)
            bpart = torch.zeros((N_BIAS_PARTIALS, 4 * hs), dtype=torch.float32, device=dev) if want_bias else None
            prev = None
            for t in (range(L - 1, -1, -1) if d == 0 else range(L)):
                A, off = bs[t], offs[t]
                _step_bwd(A, hs, None if prev is None else d4[offs[prev]:], 0 if prev is None else bs[prev], wh_t,
                          gates[off:off + A], hprev[off:off + A], None, hs, dh, d4[off:off + A], 0.0, 0, 0, bpart)
                prev = t
            with on_side(d4, hprev, bpart):
                _acc_weight_grad(grads, base + 1, w_hh, d4[:, :2 * hs], hprev, rows=slice(0, 2 * hs))
                _acc_weight_grad(grads, base + 1, w_hh, d4[:, 3 * hs:], hprev, rows=slice(2 * hs, 3 * hs))
                if want_bias:
                    _acc_bias_grads(grads, base, b_ih, b_hh, bpart.sum(0), hs)
            # gradient of the per-node input-gate tables: sum of d(xg) = d4[:, :3hs] over the rows of each node
            # (one pass over both directions -- _seg_rows(..., src2, dst2) -- measured slower: 3.09 vs 2 x 1.33 ms)
            for s_, side_t in enumerate(sides):
                dGs[s_][d] = torch.empty((side_t.n_nodes, 3 * hs), dtype=dtp, device=dev)
                _seg_rows(side_t, d4, 3 * hs, dGs[s_][d])
            if not use_side:                      # (with the auxiliary stream reading them they stay in `keep` until the join)
                l1[d] = None
            del d4, gates, hprev, bpart
        for s_, side_t in enumerate(sides):
            dG = dGs[s_]
            cols = slice(0, hs) if s_ == 0 else slice(hs, 2 * hs)
            with on_side(dG[0], dG[1], src[s_]):
                for d in (0, 1):
                    w_ih, wi = weights[8 + d * 4], wis[d]
                    _acc_weight_grad(grads, 8 + d * 4, w_ih, dG[d], src[s_], cols=cols)
                    wt = weight_t(w_ih, wi[:, cols], rows=("cols", s_))                 # [hs, 3hs]
                    if dsrc[s_] is None:
                        dsrc[s_] = gemm(dG[d], wt, trans_b=True)
                    else:
                        gemm(dG[d], wt, trans_b=True, out=dsrc[s_], accumulate=True)
        if use_side:
            main.wait_stream(aux)                  # layer 0 consumes dsrc; everything in `keep` is done with
        del dGs
        keep.clear()
        # ---- layer 0 on the tries, deepest level first
        dtab = None
        if table.requires_grad and _grad_target(table) is None:
            dtab = torch.zeros(table.shape, dtype=torch.float32, device=dev)
        # the two tries are independent: the suffix side runs on the auxiliary stream beside the prefix side (small launches per
        # level on both); with TRIE_SIDE the GEMMs go to the auxiliary stream instead and both sides stay on main
        l0_overlap = TRIE_L0_OVERLAP and side_ok(dev) and not use_side and N >= SIDE_MIN_ROWS     # small banks are launch-bound: no stream hand-overs
        emb_parts = []
        aux0 = _side_stream(dev) if l0_overlap else main
        if l0_overlap:
            aux0.wait_stream(main)
        for d, side in enumerate(sides):
          with torch.cuda.stream(aux0 if d == 1 else main):
            X, H, gates, Y, seed_e, seed_y, wi_t, wh_t = l0[d]
            n = side.n_nodes
            base = d * 4
            w_ih, w_hh, b_ih, b_hh = weights[base:base + 4]
            want_bias = b_ih.requires_grad or b_hh.requires_grad
            hp = torch.empty((n, hs), dtype=dtp, device=dev)          # the state each node started from (written by the steps)
            bpart = torch.zeros((N_BIAS_PARTIALS, 4 * hs), dtype=torch.float32, device=dev) if want_bias else None
            dy = dsrc[d]
            if TRIE_SUM_INDEX:
                # Children -> parent sums without a pass over every node: node u reads its recurrent operand (sum of its
                # children's d4 rows) and its incoming state gradient (sum of their dh * z) from row side.sum_idx[u] of the SAME
                # buffers -- its only child's row, nothing at all for a leaf (sum_idx = n: zeros), or one of the extra rows behind
                # row n that hold the sums of the nodes with several children, the only ones gtos_segment_sum_ranges still
                # visits (a quarter of the nodes at C2; half are leaves).
                nm = side.n_multi
                d4x = torch.empty((n + 1 + nm, 4 * hs), dtype=dtp, device=dev)
                dhx = torch.empty((n + 1 + nm, hs), dtype=dtp, device=dev)   # per node: (state gradient) * z, what its parent receives
                d4 = d4x[:n]
                for k in range(L - 1, -1, -1):
                    lo, hi = side.level_off[k], side.level_off[k + 1]
                    A = hi - lo
                    if A == 0:
                        continue
                    has_kids = k + 1 < L and side.level_off[k + 2] > side.level_off[k + 1]
                    mlo, mhi = side.multi_level_off[k], side.multi_level_off[k + 1]
                    if mhi > mlo:
                        rng_ = side.multi_ranges[2 * mlo:2 * mhi]
                        _seg_ranges(mhi - mlo, rng_, d4x, 4 * hs, d4x[n + 1 + mlo:])
                        _seg_ranges(mhi - mlo, rng_, dhx, hs, dhx[n + 1 + mlo:])
                    _step_bwd(A, hs, d4x if has_kids else None, n + 1 + nm, wh_t, gates[lo:hi], H, dy.data_ptr() + lo * hs * dy.element_size(),
                              hs, dhx[lo:hi], d4[lo:hi], p_layer, seed_y, lo * hs, bpart, hprev_idx=side.par[lo:hi], hp_out=hp[lo:hi],
                              sum_idx=side.sum_idx[lo:hi], dh_src=dhx, zero_row=n)
                held = (d4x, dhx, bpart, hp)
            else:                                      # GTOS_GRU_SUMIDX=0: a summed row per node of every level (round-2 form)
                d4 = torch.empty((n, 4 * hs), dtype=dtp, device=dev)
                dhz = torch.zeros((n, hs), dtype=dtp, device=dev)
                widest = max(side.level_off[k + 1] - side.level_off[k] for k in range(L))
                S = torch.empty((widest, 4 * hs), dtype=dtp, device=dev)
                for k in range(L - 1, -1, -1):
                    lo, hi = side.level_off[k], side.level_off[k + 1]
                    A = hi - lo
                    if A == 0:
                        continue
                    has_kids = k + 1 < L and side.level_off[k + 2] > side.level_off[k + 1]
                    if has_kids:
                        rng_ = side.child_off[2 * lo:2 * hi]
                        _seg_ranges(A, rng_, d4, 4 * hs, S)
                        _seg_ranges(A, rng_, dhz, hs, dhz[lo:hi])
                    _step_bwd(A, hs, S if has_kids else None, A, wh_t, gates[lo:hi], H, dy.data_ptr() + lo * hs * dy.element_size(), hs,
                              dhz[lo:hi], d4[lo:hi], p_layer, seed_y, lo * hs, bpart, hprev_idx=side.par[lo:hi], hp_out=hp[lo:hi])
                held = (d4, dhz, S, bpart, hp)
            with (on_side(d4, hp, X, bpart) if use_side else contextlib.nullcontext()):
                _acc_weight_grad(grads, base + 1, w_hh, d4[:, :2 * hs], hp, rows=slice(0, 2 * hs))
                _acc_weight_grad(grads, base + 1, w_hh, d4[:, 3 * hs:], hp, rows=slice(2 * hs, 3 * hs))
                _acc_weight_grad(grads, base, w_ih, d4[:, :3 * hs], X)
                if want_bias:
                    _acc_bias_grads(grads, base, b_ih, b_hh, bpart.sum(0), hs)
                if table.requires_grad:
                    tgt = _grad_target(table)
                    if tgt is None:
                        tgt = dtab
                    V = table.shape[0]
                    if EMBED_GRAD_GEMM and V <= 256:
                        # label-embedding gradient = onehot(token)^T (mask * dX): two products on the MFMA GEMM.  The dropout
                        # mask of the forward (counter row * dim_pad + column) is the GEMM epilogue's own counter; the scatter
                        # kernel it replaces spent 0.4 ms per trie in LDS float atomics for 95 MB of input.
                        dX = gemm(d4[:, :3 * hs], wi_t, trans_b=True, p_drop=p_embed, seed=seed_e)      # [n, dim_pad], masked
                        Vp = (V + 7) // 8 * 8
                        part = gemm(side.token_onehot(Vp, dtp), dX, trans_a=True, out_dtype=torch.float32, splitk=_splitk(Vp, dim_pad, n))
                        emb_parts.append(part)           # added to the table gradient after the two sides have joined (one writer)
                        held = held + (part,)
                    else:
                        dX = gemm(d4[:, :3 * hs], wi_t, trans_b=True)        # [n, dim_pad]
                        ws = embed_bwd_workspace(n, V, dim_pad, dev)
                        call("gtos_embed_rows_bwd", dt(dX), n, V, table.shape[1], dim_pad, ptr(side.tok), ptr(dX), ptr(tgt),
                             float(p_embed), seed_e, ptr(ws), 0 if ws is None else ws.numel() * 4, stream())
            if l0_overlap and d == 1:
                keep.extend(held)                      # allocated in the auxiliary stream's pool: released after the join below
        if l0_overlap:
            main.wait_stream(aux0)
        if emb_parts and use_side:
            main.wait_stream(aux)          # (GTOS_GRU_TRIE_SIDE=1: the products above ran on the auxiliary stream)
        for part in emb_parts:
            tgt = _grad_target(table)
            tgt = dtab if tgt is None else tgt
            tgt += part[:table.shape[0], :table.shape[1]]
        keep.clear()
        if use_side:
            if dtab is None and all(gr is None for gr in grads):
                defer_side_join(dev, keep)     # every gradient went into the flat bucket: its readers join the side stream
            else:
                main.wait_stream(aux)
        return (None, dtab, None, None, None, None) + tuple(grads)


def trie_bigru_final(trie, table, dim_pad, p_embed, hs, p_layer, weights):
    return TrieBiGRUFn.apply(trie, table, dim_pad, float(p_embed), hs, float(p_layer), *weights)
