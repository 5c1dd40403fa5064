"""TEST INFRASTRUCTURE: run the GPU path's Python glue on the CPU with every kernel launch RECORDED instead of executed.

``with DryRun() as rec:`` replaces libgtos_hip.so by a recorder (every gtos_* call returns 0 and is appended to ``rec.calls`` as
(name, args) after its arguments were checked against the binding's ctypes signature), fakes the torch.cuda stream / event API
and lets CPU tensors report ``is_cuda``.  Kernel outputs stay uninitialised memory, so numbers mean nothing; what the harness
gives without a GPU is (1) every ``call()`` site of the branches a training step takes exercised against ``_lib.SIGNATURES`` (argument
count and kind), (2) the launch plan of a step -- which entry points, how often, with which shapes -- as data, (3) the host time of
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
        if name == "gtos_rel_attn_fwd":
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
        else:
            return
        self.extent_checks += 1

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
    try:
        yield rec
    finally:
        for (obj, name), (old, own) in saved.items():
            if isinstance(obj, type) and not own:
                delattr(obj, name)                             # the attribute lived on a base class: drop the override
            else:
                setattr(obj, name, old)
