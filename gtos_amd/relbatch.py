"""Graph -> relation tensors through libgtos_host.so (include/gtos_host.h): the native counterpart of the reference's
networkx all-pairs shortest paths + ``batchify`` relation section."""
import ctypes
import os
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc_host", "libgtos_host.so")
PATH_FIRST, PATH_UNIFORM, PATH_ALL = 0, 1, 2
_lib = None
_load_lock = threading.Lock()


def load():
    """libgtos_host.so with EVERY entry's argument types set (relation batch, path tries, relation index), created once under a
    lock: loader threads call in concurrently, and a second CDLL object whose pointer arguments were still untyped (ctypes then
    passes them as 32-bit ints) segfaulted in gtos_pathtrie_build."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -m gtos_amd.build`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        P = ctypes.c_void_p
        lib.gtos_relbatch_build.restype = P
        lib.gtos_relbatch_build.argtypes = [ctypes.c_int, P, P, P, P, P, P, ctypes.c_int, ctypes.c_uint64, P, ctypes.c_int, ctypes.c_int]
        lib.gtos_relbatch_dims.restype = ctypes.c_int
        lib.gtos_relbatch_dims.argtypes = [P] + [ctypes.POINTER(ctypes.c_int)] * 4
        lib.gtos_relbatch_export.restype = ctypes.c_int
        lib.gtos_relbatch_export.argtypes = [P] * 6
        lib.gtos_relbatch_free.restype = None
        lib.gtos_relbatch_free.argtypes = [P]
        lib.gtos_relbatch_csr.restype = ctypes.c_int64
        lib.gtos_relbatch_csr.argtypes = [ctypes.c_int] + [P] * 13
        lib.gtos_pathtrie_build.restype = P
        lib.gtos_pathtrie_build.argtypes = [ctypes.c_int, ctypes.c_int64, P, P, ctypes.c_int]
        lib.gtos_pathtrie_sizes.restype = ctypes.c_int
        lib.gtos_pathtrie_sizes.argtypes = [P, P]
        lib.gtos_pathtrie_export.restype = ctypes.c_int
        lib.gtos_pathtrie_export.argtypes = [P, P]
        lib.gtos_pathtrie_derived_sizes.restype = ctypes.c_int
        lib.gtos_pathtrie_derived_sizes.argtypes = [P, ctypes.c_int, P]
        lib.gtos_pathtrie_export_derived.restype = ctypes.c_int
        lib.gtos_pathtrie_export_derived.argtypes = [P, ctypes.c_int, P]
        lib.gtos_pathtrie_free.restype = None
        lib.gtos_pathtrie_free.argtypes = [P]
        lib.gtos_relindex_build.restype = P
        lib.gtos_relindex_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, P, ctypes.c_int]
        lib.gtos_relindex_sizes.restype = ctypes.c_int
        lib.gtos_relindex_sizes.argtypes = [P, P]
        lib.gtos_relindex_export.restype = ctypes.c_int
        lib.gtos_relindex_export.argtypes = [P, P]
        lib.gtos_relindex_free.restype = None
        lib.gtos_relindex_free.argtypes = [P]
        _lib = lib
    return _lib


def build_relation_batch(graphs, special_ids, path_mode=PATH_FIRST, seed=0, max_len=8, n_threads=0):
    """graphs: list of (n_nodes, root, edges) with edges = int array [E,3] of (src, dst, label_id) in insertion order,
    reverse-labelled twins included.  special_ids = (pad, cls, rcls, self, tl) relation-vocabulary ids.
    Returns dict(relation, relation_bank, relation_length, order, depth) of torch tensors in the reference's layout."""
    lib = load()
    B = len(graphs)
    n_nodes = np.array([g[0] for g in graphs], dtype=np.int32)
    roots = np.array([g[1] for g in graphs], dtype=np.int32)
    edges = [np.asarray(g[2], dtype=np.int32).reshape(-1, 3) for g in graphs]
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(e) for e in edges])
    allE = np.concatenate(edges) if off[-1] else np.zeros((0, 3), np.int32)
    src, dst, lab = [np.ascontiguousarray(allE[:, k]) for k in range(3)]
    ids = np.array(special_ids, dtype=np.int32)
    h = lib.gtos_relbatch_build(B, n_nodes.ctypes.data, roots.ctypes.data, off.ctypes.data, src.ctypes.data, dst.ctypes.data,
                                lab.ctypes.data, path_mode, seed, ids.ctypes.data, max_len, n_threads)
    if not h:
        raise ValueError("gtos_relbatch_build rejected the batch (disconnected graph or label id outside [1,255])")
    try:
        dims = [ctypes.c_int() for _ in range(4)]
        lib.gtos_relbatch_dims(h, *[ctypes.byref(d) for d in dims])
        n, R, L, K = [d.value for d in dims]
        rel = np.empty((n, n, B) if path_mode != PATH_ALL else (n, n, B, K), dtype=np.int64)
        bank = np.empty((L, R), dtype=np.int64)
        length = np.empty((R,), dtype=np.int64)
        order = np.empty((B, n - 1), dtype=np.int32)
        depth = np.empty((B, n - 1), dtype=np.int32)
        lib.gtos_relbatch_export(h, rel.ctypes.data, bank.ctypes.data, length.ctypes.data, order.ctypes.data, depth.ctypes.data)
    finally:
        lib.gtos_relbatch_free(h)
    return dict(relation=torch.from_numpy(rel), relation_bank=torch.from_numpy(bank), relation_length=torch.from_numpy(length),
                order=torch.from_numpy(order), depth=torch.from_numpy(depth))


def dependency_edges(heads, dep_ids, dep_rev_ids):
    """Edge list of a dependency tree as translator/dependencyGraph.py:19-34 inserts it: for every word src with head
    des = head-1 >= 0: (src, des, rel) then (des, src, rel + '_r_').  Returns (n, root, edges[E,3])."""
    edges, root = [], None
    for s, hd in enumerate(heads):
        if hd == 0:
            root = s
    for s, hd in enumerate(heads):
        des = hd - 1
        if des < 0:
            continue
        edges.append((s, des, dep_ids[s]))
        edges.append((des, s, dep_rev_ids[s]))
    return len(heads), root, np.array(edges, dtype=np.int32).reshape(-1, 3)
