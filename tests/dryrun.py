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
        self.extents = {}        # data_ptr -> (end of the storage the pointer was taken from): filled by the patched _lib.ptr
        self.extent_checks = 0

    def note_ptr(self, t):
        if t is None:
            return None
        p = t.data_ptr()
        st = t.untyped_storage()
        end = st.data_ptr() + st.nbytes()
        self.extents[p] = max(self.extents.get(p, 0), end)
        return p

    def _check_gemm(self, args):
        """every operand of a gtos_gemm call lies inside the storage its pointer came from: rows x columns with the leading dimension the
        wrapper passed (A: [M,K] or, transposed, [K,M]; B: [K,N] or [N,K]; C: [M,N]; bias [N] fp32)"""
        in_dt, out_dt, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias = args[:14]
        if M <= 0 or N <= 0 or K <= 0:
            return
        es_in, es_out = (4, 2)[in_dt], (4, 2)[out_dt]
        for name, p, rows, cols, ld, es in (("A", A, K if ta else M, M if ta else K, lda, es_in), ("B", B, N if tb else K, K if tb else N, ldb, es_in),
                                            ("C", C, M, N, ldc, es_out)):
            assert p in self.extents, "gtos_gemm: operand %s did not come through ptr()" % name
            assert ld >= cols or rows == 1, "gtos_gemm: leading dimension %d of %s below its %d columns" % (ld, name, cols)
            need = p + ((rows - 1) * ld + cols) * es
            assert need <= self.extents[p], "gtos_gemm: operand %s [%d x %d, ld %d] runs %d bytes past its storage" % (
                name, rows, cols, ld, need - self.extents[p])
        if bias is not None:
            assert bias in self.extents and bias + 4 * N <= self.extents[bias], "gtos_gemm: bias shorter than N"
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
            if name == "gtos_gemm" and self.extents:
                self._check_gemm(args)
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
    import gtos_amd.flat
    import gtos_amd.gru
    import gtos_amd.ops
    import gtos_amd.train
    for mod in (_lib, gtos_amd.ops, gtos_amd.gru, gtos_amd.flat, gtos_amd.train):      # every module took its own reference to ptr()
        if hasattr(mod, "ptr"):
            patch(mod, "ptr", rec.note_ptr)
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
