"""ctypes binding of libgtos_hip.so (the C ABI declared in include/gtos_hip.h).

PyTorch only supplies device memory and the current HIP stream; every call passes raw device pointers.
There is NO fallback: if the library is missing or a kernel rejects a shape, an exception is raised.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgtos_hip.so")
ABI_VERSION = 21

c_p, c_i, c_l, c_f, c_u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64

# name -> argument ctypes (return type is always int); mirrors include/gtos_hip.h one to one
SIGNATURES = {
    "gtos_abi_version": [],
    "gtos_pathtrie_dev_workspace": [c_l, c_l, c_p],
    "gtos_pathtrie_dev_phase_a": [c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_pathtrie_dev_phase_b": [c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relbatch_dev_workspace": [c_l, c_p],
    "gtos_relbatch_dev_phase_a": [c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relbatch_dev_phase_b": [c_p, c_l, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relbatch_dev_all_count": [c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relbatch_dev_all_keys": [c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relbatch_dev_all_fill": [c_p, c_l, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relindex_dev_workspace": [c_l, c_p],
    "gtos_relindex_dev_phase_a": [c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_relindex_dev_phase_b": [c_p, c_l, c_p, c_p, ctypes.c_size_t, c_p],
    "gtos_gemm": [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_f, c_u64, c_i, c_i, c_p, c_l, c_p],
    "gtos_rel_attn_fwd": [c_i] * 7 + [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_f, c_f, c_u64, c_p, c_l, c_p, c_p, c_p],
    "gtos_rel_attn_bwd": [c_i] * 7 + [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_u64,
                                      c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p],
    "gtos_rel_attn_bwd_bank": [c_i] * 5 + [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p, c_p],  # ... nchunks, d_bank, ld, heavy, stream
    "gtos_ln_residual_fwd": [c_i, c_i, c_i, c_p, c_p, c_f, c_u64, c_p, c_p, c_f, c_p, c_p, c_p, c_p],
    "gtos_ln_residual_bwd": [c_i, c_i, c_i, c_p, c_p, c_p, c_f, c_u64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gtos_ln_residual_fwd2": [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_u64, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p],
    "gtos_ln_residual_bwd2": [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_u64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gtos_relu_dropout_bwd": [c_i, c_l, c_p, c_p, c_f, c_p],
    "gtos_colsum": [c_i, c_i, c_i, c_l, c_p, c_p, c_p],
    "gtos_gru_cell_fwd": [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_f, c_u64, c_l, c_p],
    "gtos_gru_step_fwd": [c_i, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p, c_p, c_p, c_l,
                          c_f, c_u64, c_l, c_p],
    "gtos_gru_step_bwd": [c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_i, c_l, c_p, c_f, c_u64, c_l, c_p, c_i, c_p, c_p, c_p, c_i, c_p],
    "gtos_gru_step_bwd_fused": [c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_i, c_l, c_p, c_f, c_u64, c_l, c_p, c_i, c_p, c_p, c_p, c_i,
                                c_p, c_p, c_l, c_i, c_i, c_f, c_u64, c_l, c_p],
    "gtos_gru_weight_grads": [c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p],
    "gtos_gemm_tn_batch": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],      # n, 10 host arrays of n entries, stream
    "gtos_embed_packed_paths": [c_i, c_i, c_i, c_l, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_f, c_u64, c_p, c_i, c_p, c_p],
    "gtos_segment_sum_rows": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p],
    "gtos_segment_sum_stream": [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p, c_l, c_p, c_p],
    "gtos_segment_sum_finish": [c_i, c_p, c_p, c_i, c_p, c_l, c_p],
    "gtos_segment_sum_ranges": [c_i, c_p, c_p, c_l, c_i, c_p, c_l, c_p],
    "gtos_gru_cell_bwd": [c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_f, c_u64, c_l, c_p, c_i, c_p],
    "gtos_relation_gather_mean": [c_i, c_l, c_i, c_i, c_p, c_p, c_i, c_p, c_p],
    "gtos_embed_rows_fwd": [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_f, c_u64, c_p],
    "gtos_embed_rows_bwd": [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_f, c_u64, c_p, c_l, c_p],
    "gtos_copy_nll_fwd": [c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p],
    "gtos_copy_nll_bwd": [c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gtos_copy_ll_fwd": [c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p],
    "gtos_highway_fwd": [c_i, c_l, c_i, c_p, c_p, c_p, c_p],
    "gtos_highway_bwd": [c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "gtos_max_relu_fwd": [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    "gtos_max_relu_bwd": [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "gtos_token_row_fwd": [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_u64, c_p],
    "gtos_token_row_bwd": [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_u64, c_p],
    "gtos_sqnorm": [c_l, c_p, c_p, c_p],
    "gtos_adam_step": [c_l, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_p, c_p],
    "gtos_cast_f32_to_bf16": [c_l, c_p, c_p, c_p],
    "gtos_transpose_batch_bf16": [c_i, c_p, c_p, c_i, c_p, c_p, c_p],
    "gtos_step_control": [c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p],
    "gtos_set_seed_epoch": [c_p],
    "gtos_adam_step_ctl": [c_l, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_p, c_p],
}

_lib = None
_load_lock = threading.Lock()


class GtosHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises GtosHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GtosHipError("%s is missing: run `python -m gtos_amd.build` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the gtos hot path" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        if lib.gtos_abi_version() != ABI_VERSION:
            raise GtosHipError("libgtos_hip.so ABI %d != binding ABI %d: rebuild" % (lib.gtos_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise GtosHipError("%s failed with code %d (%s)" % (
            name, rc, "HIP launch error" if rc > 0 else "unsupported argument/shape"))


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = None     # torch._C._cuda_getCurrentRawStream once it has been checked against the public API; False: not usable


def stream():
    """The current HIP stream's handle, once per launch: the public route builds a torch.cuda.Stream object every time (~1.5 us, as
    much as the ctypes call it feeds); the raw getter returns the integer directly.  It is adopted only after it has agreed with the
    public API on this process's first launch, on the current stream and on a probe stream; any surprise keeps the public route."""
    global _raw_stream
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    pub = torch.cuda.current_stream().cuda_stream
    if _raw_stream is None:
        try:
            raw = torch._C._cuda_getCurrentRawStream
            ok = raw(torch.cuda.current_device()) == pub
            probe = torch.cuda.Stream()                      # ... and it must follow a stream switch (the default stream's handle is 0:
            with torch.cuda.stream(probe):                   # agreeing there proves little)
                ok = ok and raw(torch.cuda.current_device()) == probe.cuda_stream == torch.cuda.current_stream().cuda_stream
            ok = ok and raw(torch.cuda.current_device()) == pub
            _raw_stream = raw if ok else False
        except Exception:
            _raw_stream = False
    return pub


def dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise GtosHipError("unsupported dtype %s" % t.dtype)


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise GtosHipError("gtos_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)
