"""Host-side index preparation of the factored relation operand (include/gtos_host.h, csrc_host/relindex.cpp).

``build_relation_index(relation, R)`` turns the batch's ``relation[n,n,B]`` type ids into what the attention kernels read
(query-/key-major int32 ids, pairs grouped by type in <=32-pair chunks ordered for L2 locality, heavy-type slots).  Built
with the batch on the host -- it is integer work on loader output, like the relation bank itself
(generator/data.py:134-176) -- and shipped as ``batch['relation_index']``; ``ops.FactoredRelation`` falls back to device
sort / search ops when a batch comes without it.
"""
import ctypes

import numpy as np
import torch

from . import relbatch

CHUNK = 32
_NAMES = ("idx_q", "idx_k", "pair_sorted", "chunk_type", "chunk_start", "chunk_count", "chunk_slot", "xcd_off", "heavy_types")


def _lib():
    return relbatch.load()          # every signature of libgtos_host.so is set there, once, under a lock


class RelationIndex(object):
    def __init__(self, n, B, R, arrays):
        self.n, self.B, self.R = n, B, R
        self.__dict__.update(arrays)
        self.nchunks = int(self.chunk_type.numel())
        self.n_heavy = int(self.heavy_types.numel())

    def to(self, device, *a, **k):
        return RelationIndex(self.n, self.B, self.R, {k_: getattr(self, k_).to(device) for k_ in _NAMES})

    def cpu(self):
        return self.to("cpu")

    @property
    def device(self):
        return self.idx_q.device

    def matches(self, bank, relation):
        return (tuple(relation.shape) == (self.n, self.n, self.B) and bank.shape[0] == self.R and self.device == relation.device)


def build_relation_index(relation, R, chunk=CHUNK):
    """relation: int64 [n,n,B] type ids in [0,R) (CPU tensor; a device tensor is copied to the host)."""
    rel = np.ascontiguousarray(relation.detach().cpu().numpy().astype(np.int64, copy=False))
    n, n2, B = rel.shape
    assert n == n2
    lib = _lib()
    h = lib.gtos_relindex_build(n, B, int(R), rel.ctypes.data, chunk)
    if not h:
        raise ValueError("gtos_relindex_build rejected the relation tensor (type ids must lie in [0, R))")
    try:
        sizes = np.zeros(3, dtype=np.int64)
        lib.gtos_relindex_sizes(h, sizes.ctypes.data)
        P, nc, nh = [int(v) for v in sizes]
        shapes = [P, P, P, nc, nc, nc, nc, 9, nh]
        arrs = [np.zeros(max(1, s), dtype=np.int32) for s in shapes]
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        assert lib.gtos_relindex_export(h, ptrs) == len(arrs)
    finally:
        lib.gtos_relindex_free(h)
    arrays = {k: torch.from_numpy(a[:s]) for k, a, s in zip(_NAMES, arrs, shapes)}
    arrays["idx_q"] = arrays["idx_q"].view(n, B, n)
    arrays["idx_k"] = arrays["idx_k"].view(n, B, n)
    return RelationIndex(n, B, int(R), arrays)


def attach_relation_index(batch):
    """Adds ``batch['relation_index']`` for train-mode batches (relation [n,n,B]); eval batches ([n,n,B,K]) are untouched."""
    if batch['relation'].dim() == 3:
        batch['relation_index'] = build_relation_index(batch['relation'], batch['relation_bank'].shape[1])
    return batch
