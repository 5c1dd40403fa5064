"""The relation index of the factored attention operand built by the staged GPU builder (csrc/relindex_dev.hip; per-thread stages in
csrc/relindex_kernels.h): the device-side counterpart of ``relindex.build_relation_index`` (csrc_host/relindex.cpp), same arrays.

Two phases with one host read between them (the chunk and heavy-type counts): phase A sorts the cells by type (graph-major inside a
type), writes the query- / key-major ids and the chunk records in type order; phase B places the chunks on the XCDs (a chunk whose
pairs live on one XCD stays there; the others go to the least loaded XCD, longest first -- a serial walk over a few thousand chunks,
one thread) and sorts them into their final order.  tests/test_relindex_dev.py runs the SAME stage code as serial host loops through the
test suite's emulation library and compares with the host builder array for array; tests/test_zzz_hip_relbatch.py runs the HIP library
on the GPU.

Status: written at the end of round 3 without GPU time left -- the HIP entry points compile for gfx950 and have not run yet.
Opt-in: nothing selects this module by default.
"""
import ctypes
import os

import torch

from .relindex import CHUNK, RelationIndex

# order of geom[] and of the pointer table: csrc/relindex_kernels.h (enum GE_* / T_*)
_GEOM = ("n", "B", "chunk", "mult", "R", "P")
_TABLE = ("relation", "key", "val", "skey", "pair_sorted", "idx_q", "idx_k", "cnt", "cum_cnt", "nch", "cum_nch", "heavy", "cum_heavy",
          "heavy_types", "c_type", "c_start", "c_cnt", "c_slot", "c_xf", "c_xl", "c_home", "c_key0", "load", "sizes",
          "rkey", "rval", "rkey_s", "roam_sorted", "rcost", "home_q", "fkey", "fval", "fkey_s", "perm", "v8", "v8_cum", "chunk_type", "chunk_start", "chunk_count", "chunk_slot",
          "xcd_off")


def _table(bufs):
    return (ctypes.c_void_p * len(_TABLE))(*[(bufs[n].data_ptr() if n in bufs else None) for n in _TABLE])


def _geom(g):
    return (ctypes.c_int64 * len(_GEOM))(*[int(g[k]) for k in _GEOM])


class HipBackend(object):
    """gtos_relindex_dev_phase_a / _b of libgtos_hip.so on the current stream."""
    _shared = None

    def __init__(self):
        from ._lib import load, stream
        self._lib, self._stream = load(), stream
        self._ws = None

    @classmethod
    def shared(cls):
        if cls._shared is None:
            cls._shared = cls()
        return cls._shared

    def _workspace(self, n, dev):
        out = ctypes.c_int64(0)
        if self._lib.gtos_relindex_dev_workspace(n, ctypes.byref(out)):
            raise RuntimeError("gtos_relindex_dev_workspace rejected %d elements" % n)
        need = int(out.value)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def phase_a(self, geom, bufs):
        ws = self._workspace(geom["P"] + geom["R"] + 1, bufs["relation"].device)
        rc = self._lib.gtos_relindex_dev_phase_a(_geom(geom), _table(bufs), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_relindex_dev_phase_a failed: %d" % rc)

    def phase_b(self, geom, nchunks, bufs):
        ws = self._workspace(geom["P"] + geom["R"] + 1, bufs["relation"].device)
        rc = self._lib.gtos_relindex_dev_phase_b(_geom(geom), nchunks, _table(bufs), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_relindex_dev_phase_b failed: %d" % rc)


def build_relation_index_staged(relation, R, backend, chunk=CHUNK):
    """relation: int64 [n,n,B] type ids in [0,R) on the backend's device.  ValueError outside the covered case (ids out of range)."""
    relation = relation.to(torch.int64).contiguous()
    n, n2, B = relation.shape
    assert n == n2
    R = int(R)
    P = n * n * B
    if P > 0x7fffffff or R > 0x7fffffff or R < 1 or n >= 1 << 20:
        raise ValueError("relation tensor outside the builder's range")
    dev = relation.device
    mult = 4                                        # heavy types: chunks of 4 * chunk pairs (csrc_host/relindex.cpp)
    geom = dict(n=n, B=B, chunk=chunk, mult=mult, R=R, P=P)
    i32, i64 = torch.int32, torch.int64
    NC = R + P // chunk + 1

    def E(k, dt=i32):
        return torch.empty(max(1, k), dtype=dt, device=dev)
    bufs = dict(relation=relation, key=E(P), val=E(P), skey=E(P), pair_sorted=E(P), idx_q=E(P), idx_k=E(P),
                cnt=E(R), cum_cnt=E(R), nch=E(R), cum_nch=E(R), heavy=E(R), cum_heavy=E(R),
                heavy_types=E(P // chunk + 1), c_type=E(NC), c_start=E(NC), c_cnt=E(NC), c_slot=E(NC), c_xf=E(NC), c_xl=E(NC), c_home=E(NC),
                c_key0=E(NC, i64), load=E(8, i64),
                sizes=torch.zeros(4, dtype=i32, device=dev))
    backend.phase_a(geom, bufs)
    err, nchunks, n_heavy, _ = bufs["sizes"].tolist()                     # the one host read
    if err:
        raise ValueError("relation type ids must lie in [0, R)")
    out = {}
    if nchunks > 0:
        bufs.update(rkey=E(nchunks), rval=E(nchunks), rkey_s=E(nchunks), roam_sorted=E(nchunks), rcost=E(nchunks), home_q=E(nchunks), fkey=E(nchunks, i64), fval=E(nchunks),
                    fkey_s=E(nchunks, i64), perm=E(nchunks), v8=E(8 * nchunks), v8_cum=E(8 * nchunks), chunk_type=E(nchunks), chunk_start=E(nchunks), chunk_count=E(nchunks),
                    chunk_slot=E(nchunks), xcd_off=E(9))
        backend.phase_b(geom, nchunks, bufs)
        out = {k: bufs[k][:nchunks] for k in ("chunk_type", "chunk_start", "chunk_count", "chunk_slot")}
        out["xcd_off"] = bufs["xcd_off"]
    else:                                                                 # every type occurs exactly once: no chunk at all
        z = torch.zeros(0, dtype=i32, device=dev)
        out = dict(chunk_type=z, chunk_start=z, chunk_count=z, chunk_slot=z, xcd_off=torch.zeros(9, dtype=i32, device=dev))
    out.update(idx_q=bufs["idx_q"][:P].view(n, B, n), idx_k=bufs["idx_k"][:P].view(n, B, n), pair_sorted=bufs["pair_sorted"][:P],
               heavy_types=bufs["heavy_types"][:n_heavy])
    return RelationIndex(n, B, R, out)
