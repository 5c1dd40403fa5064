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
