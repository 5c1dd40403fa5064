"""TEST INFRASTRUCTURE: run the GPU path's Python glue on the CPU with every kernel launch RECORDED instead of executed.

``with DryRun() as rec:`` replaces libgtos_hip.so by a recorder (every gtos_* call returns 0 and is appended to ``rec.calls`` as
(name, args) after its arguments were checked against the binding's ctypes signature), fakes the torch.cuda stream / event API
and lets CPU tensors report ``is_cuda``.  Kernel outputs stay uninitialised memory, so numbers mean nothing; what the harness
gives without a GPU is (1) every ``call()`` site of the branches a training step takes exercised against ``_lib.SIGNATURES`` (argument
count and kind) and, for the GEMM, attention, bank-gradient, LayerNorm, column-sum, GRU-step and segment-sum launches (85 % of a step), a
BOUNDS CHECK of every operand: rows x leading dimension inside the storage of the tensor the pointer came from, every gathered row
index (read from the real index arrays of the batch: trie row lists, parent / children-sum indices, chunk lists, pair lists) inside
its table, (2) the launch plan of a step -- which entry points, how often, with which shapes -- as data, (3) the host time of
a step's Python glue, which is what bounds the launch-bound configurations.  Nothing under gtos_amd/ knows about this module."""
import contextlib
import ctypes

import torch

from gtos_amd import _lib


class FakeStream(object):
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        return e or FakeEvent()

    def synchronize(self):
        pass

    def query(self):
        return True


class FakeEvent(object):
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def wait(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0


def _kind_ok(ctype, v):
    if ctype in (ctypes.c_void_p,):
        return v is None or isinstance(v, int)
    if ctype in (ctypes.c_float,):
        return isinstance(v, (int, float)) and not isinstance(v, bool)
    return isinstance(v, int) and not isinstance(v, bool) or isinstance(v, bool)      # c_int / c_int64 / c_uint64 / c_size_t


class Recorder(object):
    def __init__(self):
        self.calls = []
        self.recent = {}         # storage base -> end, of every tensor whose data_ptr() was taken since the last recorded launch
        self.check_extents = True
        self.unknown_ptrs = 0
        self.extent_checks = 0

    def note_storage(self, t):
        """called for EVERY Tensor.data_ptr() under the dry run: the storage [base, end) a launch argument can have been computed from.
        The table is cleared after every recorded launch, so an argument only ever resolves to a storage whose tensor was alive when
        the arguments of THAT launch were evaluated (the CPU allocator reuses freed memory at once)."""
        st = t.untyped_storage()
        self.recent[st.data_ptr()] = st.data_ptr() + st.nbytes()

    def _end_of(self, p):
        for base, end in self.recent.items():
            if base <= p < end:
                return end
        return None

    def _check_gemm(self, args):
        """every operand of a gtos_gemm call lies inside the storage its pointer came from: rows x columns with the leading dimension the
        wrapper passed (A: [M,K] or, transposed, [K,M]; B: [K,N] or [N,K]; C: [M,N]; bias [N] fp32)"""
        in_dt, out_dt, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias = args[:14]
        if M <= 0 or N <= 0 or K <= 0:
            return
        es_in, es_out = (4, 2)[in_dt], (4, 2)[out_dt]
        for name, p, rows, cols, ld, es in (("A", A, K if ta else M, M if ta else K, lda, es_in), ("B", B, N if tb else K, K if tb else N, ldb, es_in),
                                            ("C", C, M, N, ldc, es_out)):
            self._rows("gtos_gemm: operand " + name, p, rows, cols, ld, es)
        if bias is not None:
            self._rows("gtos_gemm: bias", bias, 1, N, N, 4)
        # the product is never written over one of its own operands
        span = lambda p_, rows, cols, ld, es: (p_, p_ + ((rows - 1) * ld + cols) * es)      # noqa: E731
        c0, c1 = span(C, M, N, ldc, es_out)
        for name, (x0, x1) in (("A", span(A, K if ta else M, M if ta else K, lda, es_in)), ("B", span(B, N if tb else K, K if tb else N, ldb, es_in))):
            assert x1 <= c0 or c1 <= x0, "gtos_gemm: C overlaps operand %s" % name
        self.extent_checks += 1

    def _rows(self, what, p, rows, cols, ld, es):
        """a [rows, cols] operand with leading dimension ld (elements of es bytes) behind pointer p lies inside its storage"""
        if p is None or rows <= 0 or cols <= 0:
            return
        end = self._end_of(p)
        if end is None:                 # a buffer that never went through ptr() (an offset into a fresh torch.empty): nothing to check against
            self.unknown_ptrs += 1
            return
        assert ld >= cols or rows == 1, "%s: leading dimension %d below its %d columns" % (what, ld, cols)
        need = p + ((rows - 1) * ld + cols) * es
        assert need <= end, "%s [%d x %d, ld %d] runs %d bytes past its storage" % (what, rows, cols, ld, need - end)

    def _check_other(self, name, a):
        es = lambda dtc: (4, 2)[dtc]                                          # noqa: E731
        if name == "gtos_gemm_tn_batch":                                      # ten host arrays of n entries: read them back
            import ctypes as ct_
            n = a[0]
            vp = lambda addr: list((ct_.c_void_p * n).from_address(addr))      # noqa: E731
            i64 = lambda addr: list((ct_.c_int64 * n).from_address(addr))      # noqa: E731
            i32 = lambda addr: list((ct_.c_int * n).from_address(addr))        # noqa: E731
            A, lda, M, B, ldb, N, K, C, ldc, bias = vp(a[1]), i64(a[2]), i32(a[3]), vp(a[4]), i64(a[5]), i32(a[6]), i32(a[7]), vp(a[8]), i64(a[9]), vp(a[10])
            for j in range(n):
                assert M[j] % 8 == 0 and N[j] % 8 == 0 and K[j] > 0, (M[j], N[j], K[j])
                self._rows("gtos_gemm_tn_batch: dY of job %d" % j, A[j], K[j], M[j], lda[j], 2)
                self._rows("gtos_gemm_tn_batch: X of job %d" % j, B[j], K[j], N[j], ldb[j], 2)
                self._rows("gtos_gemm_tn_batch: dW of job %d" % j, C[j], M[j], N[j], ldc[j], 4)
                if bias[j]:
                    self._rows("gtos_gemm_tn_batch: db of job %d" % j, bias[j], 1, M[j], M[j], 4)
                self.extent_checks += 1                                        # (one checked product per job, like the gtos_gemm call it replaces)
        elif name == "gtos_rel_attn_fwd":
            dtc, mode, T, S, B, H, d = a[:7]
            for what, p, ld, rows in (("q", a[7], a[8], T * B), ("k", a[9], a[10], S * B), ("v", a[11], a[12], S * B), ("o", a[20], a[21], T * B)):
                self._rows("gtos_rel_attn_fwd: " + what, p, rows, d, ld, es(dtc))
            self._rows("gtos_rel_attn_fwd: lse", a[22], 1, T * B * H, T * B * H, 4)
            self._rows("gtos_rel_attn_fwd: key_pad", a[15], 1, S * B, S * B, 1)
            self._rows("gtos_rel_attn_fwd: attn_mask", a[16], 1, T * S, T * S, 1)
            if mode == 2:
                self._rows("gtos_rel_attn_fwd: idx_q", a[14], 1, T * B * S, T * B * S, 4)
        elif name == "gtos_rel_attn_bwd":
            dtc, mode, T, S, B, H, d = a[:7]
            for what, p, ld, rows in (("q", a[7], a[8], T * B), ("k", a[9], a[10], S * B), ("v", a[11], a[12], S * B), ("o", a[21], a[22], T * B),
                                      ("d_o", a[25], a[26], T * B), ("dq", a[28], a[29], T * B), ("dk", a[30], a[31], S * B), ("dv", a[32], a[33], S * B)):
                self._rows("gtos_rel_attn_bwd: " + what, p, rows, d, ld, es(dtc))
            self._rows("gtos_rel_attn_bwd: lse", a[23], 1, T * B * H, T * B * H, 4)
        elif name == "gtos_colsum":
            dtc, rows, N, ld, dy, out = a[:6]
            self._rows("gtos_colsum: dy", dy, rows, N, ld, es(dtc))
            self._rows("gtos_colsum: out", out, 1, N, N, 4)
        elif name == "gtos_ln_residual_fwd":
            dtc, rows, d = a[:3]
            for what, p in (("x", a[3]), ("r", a[4]), ("y", a[10])):
                self._rows("gtos_ln_residual_fwd: " + what, p, rows, d, d, es(dtc))
            for what, p, n in (("gamma", a[7], d), ("beta", a[8], d), ("mean", a[11], rows), ("rstd", a[12], rows)):
                self._rows("gtos_ln_residual_fwd: " + what, p, 1, n, n, 4)
        elif name == "gtos_ln_residual_fwd2":
            xd, rd, yd, rows, d = a[:5]
            for what, p, e in (("x", a[5], es(xd)), ("r", a[6], es(rd)), ("y", a[12], es(yd)), ("y2", a[13], 2)):
                self._rows("gtos_ln_residual_fwd2: " + what, p, rows, d, d, e)
            for what, p, n in (("gamma", a[9], d), ("beta", a[10], d), ("mean", a[14], rows), ("rstd", a[15], rows)):
                self._rows("gtos_ln_residual_fwd2: " + what, p, 1, n, n, 4)
        elif name == "gtos_ln_residual_bwd2":
            xd, rd, yd, rows, d = a[:5]
            assert a[5] is not None or a[6] is not None, "gtos_ln_residual_bwd2: no incoming gradient"
            for what, p, e in (("dy", a[5], es(yd)), ("dy2", a[6], 2), ("x", a[7], es(xd)), ("r", a[8], es(rd)), ("dx", a[14], es(xd)),
                               ("dr", a[15], es(rd))):
                self._rows("gtos_ln_residual_bwd2: " + what, p, rows, d, d, e)
            for what, p, n in (("gamma", a[11], d), ("mean", a[12], rows), ("rstd", a[13], rows), ("dgamma", a[16], d), ("dbeta", a[17], d)):
                self._rows("gtos_ln_residual_bwd2: " + what, p, 1, n, n, 4)
        elif name == "gtos_gru_step_fwd":
            (rows, hs, x, ldx, in_dim, w_ih, b_ih, xg, gf, gf_idx, gb, gb_idx, h_in, h_idx, w_hh, b_hh, h_out, n_out, h_fin, ld_fin, fin_idx,
             gates, y, ldy) = a[:24]
            W = "gtos_gru_step_fwd: "
            self._rows(W + "x", x, rows, in_dim, ldx, 2)
            self._rows(W + "w_ih", w_ih if x is not None else None, 3 * hs, in_dim, in_dim, 2)
            self._rows(W + "b_ih", b_ih, 1, 3 * hs, 3 * hs, 4)
            self._rows(W + "xg", xg, rows, 3 * hs, 3 * hs, 2)
            self._gather(W + "gf", gf, gf_idx, 0, rows, 3 * hs, 3 * hs, 2)
            self._gather(W + "gb", gb, gb_idx, 0, rows, 3 * hs, 3 * hs, 2)
            if h_idx is not None:
                self._gather(W + "h_in", h_in, h_idx, 0, rows, hs, hs, 2)
            else:
                self._rows(W + "h_in", h_in, rows, hs, hs, 2)
            self._rows(W + "w_hh", w_hh, 3 * hs, hs, hs, 2)
            self._rows(W + "b_hh", b_hh, 1, 3 * hs, 3 * hs, 4)
            self._rows(W + "h_out", h_out, min(rows, n_out), hs, hs, 2)
            if rows > n_out:
                if fin_idx is not None:
                    self._gather(W + "h_fin", h_fin, fin_idx, n_out, rows, hs, ld_fin, 2)
                else:
                    self._rows(W + "h_fin", h_fin, rows, hs, ld_fin, 2)
            self._rows(W + "gates", gates, rows, 4 * hs, 4 * hs, 2)
            self._rows(W + "y", y, rows, hs, ldy, 2)
        elif name in ("gtos_gru_step_bwd", "gtos_gru_step_bwd_fused"):
            (rows, hs, d4_prev, rows_prev, w_hh_t, gates, hprev, hprev_idx, dy, ldy, dh, dh_dtype, ld_dh, d4, p_drop, seed, drop_base,
             bias_partials, n_partials, hprev_out, sum_idx, dh_src, zero_row) = a[:23]
            W = name + ": "
            if name == "gtos_gru_step_bwd_fused":
                w_ih_t, dinp, ld_dinp, n_in, acc, p_in, seed_in, in_drop_base = a[23:31]
                if dinp is not None:
                    assert d4_prev is not None and sum_idx is None and n_in % 64 == 0 and rows_prev > 0, W + "role B arguments"
                    self._rows(W + "w_ih_t", w_ih_t, n_in, 3 * hs, 3 * hs, 2)
                    self._rows(W + "dinp", dinp, rows_prev, n_in, ld_dinp, 2)
                    self._rows(W + "d4_prev (role B)", d4_prev, rows_prev, 4 * hs, 4 * hs, 2)
                if rows <= 0:
                    self.extent_checks += 1
                    return
            es_dh = (4, 2)[dh_dtype]
            self._rows(W + "w_hh_t", w_hh_t if d4_prev is not None else None, hs, 3 * hs, 3 * hs, 2)
            self._rows(W + "gates", gates, rows, 4 * hs, 4 * hs, 2)
            if hprev_idx is not None:
                self._gather(W + "hprev", hprev, hprev_idx, 0, rows, hs, hs, 2)
            else:
                self._rows(W + "hprev", hprev, rows, hs, hs, 2)
            self._rows(W + "dy", dy, rows, hs, ldy, 2)
            self._rows(W + "dh", dh, rows, hs, ld_dh, es_dh)
            self._rows(W + "d4", d4, rows, 4 * hs, 4 * hs, 2)
            self._rows(W + "bias_partials", bias_partials, n_partials, 4 * hs, 4 * hs, 4)
            self._rows(W + "hprev_out", hprev_out, rows, hs, hs, 2)
            if sum_idx is not None:       # rows equal to zero_row take zeros without a fetch
                self._gather(W + "d4_prev", d4_prev, sum_idx, 0, rows, 4 * hs, 4 * hs, 2, skip=zero_row)
                self._gather(W + "dh_src", dh_src, sum_idx, 0, rows, hs, hs, es_dh, skip=zero_row)
            elif d4_prev is not None:
                self._rows(W + "d4_prev", d4_prev, min(rows, rows_prev), 4 * hs, 4 * hs, 2)
        elif name in ("gtos_segment_sum_stream", "gtos_segment_sum_rows"):
            import ctypes
            import numpy as np
            if name == "gtos_segment_sum_stream":
                (n_chunks, total_rows, rows_p, chunk_node, chunk_start, chunk_cnt, chunk_slot, wave_off, n_waves, src, ld_src, width, dst, ld_dst,
                 heavy) = a[:15]
                self._rows(name + ": wave_off", wave_off, 1, n_waves + 1, n_waves + 1, 4)
            else:
                (n_chunks, rows_p, chunk_node, chunk_start, chunk_cnt, chunk_slot, src, src2, ld_src, width, dst, dst2, ld_dst, heavy, heavy2) = a[:15]
                total_rows = None
            if n_chunks > 0:
                arr = lambda p_, n_: np.ctypeslib.as_array((ctypes.c_int32 * n_).from_address(p_))      # noqa: E731
                for what, p_ in (("chunk_node", chunk_node), ("chunk_start", chunk_start), ("chunk_cnt", chunk_cnt), ("chunk_slot", chunk_slot)):
                    self._rows(name + ": " + what, p_, 1, n_chunks, n_chunks, 4)
                cs, cc = arr(chunk_start, n_chunks), arr(chunk_cnt, n_chunks)
                n_rows = int((cs + cc).max())
                assert int(cs.min()) >= 0 and int(cc.min()) >= 0 and (total_rows is None or n_rows <= total_rows), name + ": chunk ranges"
                self._rows(name + ": rows", rows_p, 1, n_rows, n_rows, 4)
                if n_rows:
                    self._gather(name + ": src", src, rows_p, 0, n_rows, width, ld_src, 2)
                self._gather(name + ": dst", dst, chunk_node, 0, n_chunks, width, ld_dst, 2)
                slots = arr(chunk_slot, n_chunks)
                if heavy is not None and int(slots.max()) >= 0:
                    self._rows(name + ": heavy", heavy, int(slots.max()) + 1, width, width, 4)
        elif name == "gtos_rel_attn_bwd_bank":
            import ctypes
            import numpy as np
            (dtc, n, B, H, d, q, ldq, k, ldk, bank, gs, pair_sorted, chunk_type, chunk_start, chunk_count, chunk_slot, xcd_off, nchunks, d_bank,
             ld_dbank, heavy) = a[:21]
            W, P = name + ": ", n * n * B
            self._rows(W + "q", q, n * B, d, ldq, es(dtc))
            self._rows(W + "k", k, n * B, d, ldk, es(dtc))
            self._rows(W + "gs", gs, 1, P * H, P * H, 4)
            self._rows(W + "xcd_off", xcd_off, 1, 9, 9, 4)
            if nchunks > 0:
                arr = lambda p_, n_: np.ctypeslib.as_array((ctypes.c_int32 * n_).from_address(p_))      # noqa: E731
                for what, p_ in (("chunk_type", chunk_type), ("chunk_start", chunk_start), ("chunk_count", chunk_count), ("chunk_slot", chunk_slot)):
                    self._rows(W + what, p_, 1, nchunks, nchunks, 4)
                cs, cc = arr(chunk_start, nchunks), arr(chunk_count, nchunks)
                assert int(cs.min()) >= 0 and int(cc.min()) >= 0 and int((cs + cc).max()) <= P, W + "chunk ranges outside the pair list"
                self._rows(W + "pair_sorted", pair_sorted, 1, P, P, 4)
                ps = arr(pair_sorted, P)
                assert int(ps.min()) >= 0 and int(ps.max()) < P, W + "pair ids outside [0, n*n*B)"
                assert int(arr(xcd_off, 9)[8]) == nchunks, W + "xcd_off does not cover the chunk list"
                self._gather(W + "bank", bank, chunk_type, 0, nchunks, 2 * d, 2 * d, es(dtc))
                self._gather(W + "d_bank", d_bank, chunk_type, 0, nchunks, 2 * d, ld_dbank, es(dtc))
                slots = arr(chunk_slot, nchunks)
                if heavy is not None and int(slots.max()) >= 0:
                    self._rows(W + "heavy", heavy, int(slots.max()) + 1, 2 * d, 2 * d, 4)
        elif name == "gtos_gru_weight_grads":
            rows, hs, in_dim, in_valid, d4, x, ldx, hprev, ldh, dwih, ld_ih, dwhh, ld_hh, wsp, ws_bytes = a[:15]
            assert hs % 64 == 0 and in_dim % 8 == 0 and 0 < in_valid <= in_dim and in_valid % 4 == 0, name + ": shape"
            self._rows(name + ": d4", d4, rows, 4 * hs, 4 * hs, 2)
            self._rows(name + ": x", x, rows, in_dim, ldx, 2)
            self._rows(name + ": hprev", hprev, rows, hs, ldh, 2)
            self._rows(name + ": dwih", dwih, 3 * hs, in_valid, ld_ih, 4)
            self._rows(name + ": dwhh", dwhh, 3 * hs, hs, ld_hh, 4)
            self._rows(name + ": workspace", wsp, 1, ws_bytes, ws_bytes, 1)
        elif name == "gtos_embed_packed_paths":
            import ctypes
            import numpy as np
            dtc, L, R, n_rows, bank, order, offs, table, dim, dim_pad, x, p_drop, seed, onehot, vp, tokens = a[:16]
            self._rows(name + ": bank", bank, L, R, R, 8)
            self._rows(name + ": offs", offs, 1, L + 1, L + 1, 4)
            o = np.ctypeslib.as_array((ctypes.c_int32 * (L + 1)).from_address(offs))
            assert int(o[0]) == 0 and int(o[L]) == n_rows and bool((np.diff(o) >= 0).all()), name + ": step offsets"
            widest = int(np.diff(o).max())
            self._rows(name + ": order", order, 1, widest, widest, 4)
            od = np.ctypeslib.as_array((ctypes.c_int32 * widest).from_address(order))
            assert int(od.min()) >= 0 and int(od.max()) < R, name + ": sorted order outside the bank's columns"
            bk = np.ctypeslib.as_array((ctypes.c_int64 * (L * R)).from_address(bank))
            self._rows(name + ": table", table, int(bk.max()) + 1, dim, dim, 4)
            assert int(bk.min()) >= 0 and (onehot is None or int(bk.max()) < vp), name + ": label ids"
            self._rows(name + ": x", x, n_rows, dim_pad, dim_pad, es(dtc))
            self._rows(name + ": onehot", onehot, n_rows, vp, vp, 2)
            self._rows(name + ": tokens", tokens, 1, n_rows, n_rows, 8)
        elif name == "gtos_embed_rows_fwd":
            dtc, n, dim, dim_pad, tok, table, out = a[:7]
            self._gather(name + ": table", table, tok, 0, n, dim, dim, 4, itype=8)
            self._rows(name + ": out", out, n, dim_pad, dim_pad, es(dtc))
        elif name == "gtos_token_row_fwd":
            dtc, N, Cc, Ct, Cp, feat, tok, table = a[:8]
            self._rows(name + ": feat", feat, N, Cc, Cc, es(dtc))
            self._gather(name + ": table", table, tok, 0, N, Ct, Ct, 4, itype=8)
        elif name == "gtos_relation_gather_mean":
            dtc, P, K, d, bank, idx, zero_row0, out = a[:8]
            self._gather(name + ": bank", bank, idx, 0, P * K, d, d, es(dtc), itype=8)
            self._rows(name + ": out", out, P, d, d, es(dtc))
        elif name == "gtos_segment_sum_ranges":
            n_seg, ranges, src, ld_src, width, dst, ld_dst = a[:7]
            if n_seg > 0:
                import ctypes
                import numpy as np
                self._rows(name + ": ranges", ranges, 1, 2 * n_seg, 2 * n_seg, 4)
                r = np.ctypeslib.as_array((ctypes.c_int32 * (2 * n_seg)).from_address(ranges))
                assert int(r.min()) >= 0 and bool((r[1::2] >= r[0::2]).all()), name + ": ranges not ordered"
                self._rows(name + ": src", src, int(r[1::2].max()), width, ld_src, 2)
                self._rows(name + ": dst", dst, n_seg, width, ld_dst, 2)
        else:
            return
        self.extent_checks += 1

    def _gather(self, what, table, idx, lo, hi, cols, ld, es, skip=None, itype=4):
        """table rows idx[lo:hi] (int32 indices read from the real index array) of `cols` elements, row stride ld, lie inside the table"""
        if table is None or idx is None or hi <= lo:
            return
        import ctypes
        import numpy as np
        self._rows(what + " index", idx, 1, hi, hi, itype)
        end = self._end_of(table)
        if end is None:
            self.unknown_ptrs += 1
            return
        ix = np.ctypeslib.as_array(((ctypes.c_int32 if itype == 4 else ctypes.c_int64) * hi).from_address(idx))[lo:hi]
        if skip is not None:
            ix = ix[ix != skip]
            if ix.size == 0:
                return
        assert int(ix.min()) >= 0, "%s: negative row index %d" % (what, int(ix.min()))
        need = table + (int(ix.max()) * ld + cols) * es
        assert need <= end, "%s: row index %d reaches %d bytes past the table" % (what, int(ix.max()), need - end)

    def __getattr__(self, name):
        if name == "gtos_abi_version":
            return lambda: _lib.ABI_VERSION
        if name not in _lib.SIGNATURES:
            raise AttributeError("no such entry point in the binding: %s" % name)
        sig = _lib.SIGNATURES[name]

        def fn(*args):
            assert len(args) == len(sig), "%s: %d arguments for a %d-argument signature" % (name, len(args), len(sig))
            for k, (c, v) in enumerate(zip(sig, args)):
                assert _kind_ok(c, v), "%s: argument %d is %r, signature says %s" % (name, k, v, c.__name__)
            if self.check_extents:
                if name == "gtos_gemm":
                    self._check_gemm(args)
                else:
                    self._check_other(name, args)
            self.recent.clear()
            self.calls.append((name, args))
            return 0
        return fn

    def names(self):
        return [c[0] for c in self.calls]

    def histogram(self):
        h = {}
        for n, _ in self.calls:
            h[n] = h.get(n, 0) + 1
        return h


@contextlib.contextmanager
def DryRun():
    rec = Recorder()
    saved = {}

    def patch(obj, name, value):
        saved[(obj, name)] = (getattr(obj, name), name in vars(obj) if isinstance(obj, type) else True)
        setattr(obj, name, value)
    patch(_lib, "_lib", rec)                                   # load() hands out the recorder
    real_data_ptr = torch.Tensor.data_ptr

    def data_ptr(self):
        rec.note_storage(self)
        return real_data_ptr(self)
    patch(torch.Tensor, "data_ptr", data_ptr)                  # every launch argument starts as some tensor's data_ptr()
    patch(_lib, "_raw_stream", False)
    patch(torch.cuda, "current_stream", lambda device=None: FakeStream())
    patch(torch.cuda, "Stream", FakeStream)
    patch(torch.cuda, "Event", FakeEvent)
    patch(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    patch(torch.cuda, "synchronize", lambda device=None: None)
    patch(torch.cuda, "current_device", lambda: 0)
    patch(torch.Tensor, "is_cuda", property(lambda self: True))
    patch(torch.Tensor, "record_stream", lambda self, s: None)
    # no kernel runs, so what a kernel would have written stays what the allocator handed out: zeros instead of whatever the heap held, or the
    # glue downstream of a kernel (top-k token ids of the beam search into the vocabulary) depends on stale memory -- the beam-search dry run
    # failed one run in three with an IndexError from an out-of-vocabulary id picked out of garbage
    real_empty, real_empty_like, real_new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
    patch(torch, "empty", lambda *a, **k: real_empty(*a, **k).zero_())
    patch(torch, "empty_like", lambda *a, **k: real_empty_like(*a, **k).zero_())
    patch(torch.Tensor, "new_empty", lambda self, *a, **k: real_new_empty(self, *a, **k).zero_())
    try:
        yield rec
    finally:
        for (obj, name), (old, own) in saved.items():
            if isinstance(obj, type) and not own:
                delattr(obj, name)                             # the attribute lived on a base class: drop the override
            else:
                setattr(obj, name, old)
