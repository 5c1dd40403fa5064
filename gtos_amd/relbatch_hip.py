"""The relation tensors of a batch built by the staged GPU builder (csrc/relbatch_dev.hip; per-thread stages in
csrc/relbatch_kernels.h): the device-side counterpart of ``relbatch.build_relation_batch`` -- the one-path-per-pair modes
(PATH_FIRST: translator flavour; PATH_UNIFORM: generator flavour in training) through ``build_relation_batch_staged``, the every-path
mode of the eval batches (PATH_ALL) through ``build_relation_batch_all_staged`` --, i.e. the relation section of the reference's batchify
(generator/data.py:134-176, translator/data.py:132-176) and the all-pairs shortest label paths under it.

The host flattens the graphs (ordered adjacency + BFS order, ``gtos_relbatch_csr`` of libgtos_host.so: a few thousand integers);
the device does the all-pairs work: a BFS per (graph, source), a key per pair, a key sort, the distinct keys numbered in first-seen
order by a second sort, the scatter into ``relation[n,n,B]`` and the bank.  One host read (R and L) between the two phases.
The result equals the host builder's array for array (tests/test_relbatch_dev.py runs the SAME stage code as serial host loops
through the test suite's emulation library; tests/test_zzz_hip_relbatch.py runs the HIP library on the GPU).

Status: written at the end of round 3 without GPU time left -- the HIP entry points compile for gfx950 and have not run yet.
Opt-in: nothing selects this module by default.
"""
import ctypes

import numpy as np
import torch

from . import relbatch

PATH_FIRST, PATH_UNIFORM, PATH_ALL = relbatch.PATH_FIRST, relbatch.PATH_UNIFORM, relbatch.PATH_ALL
# order of geom[] and of the pointer table: csrc/relbatch_kernels.h (enum GE_* / T_*)
_GEOM = ("B", "n", "nmax", "emax", "max_len", "mode", "S", "P", "seed", "cls", "rcls", "self", "tl", "T", "K", "pad")
_TABLE = ("ng", "node_off", "pair_off", "adj_base", "adj_off", "adj_dst", "adj_lab", "order",
          "level", "count", "head", "tail", "queue", "dpred", "dnext", "dlab",
          "key", "posn", "skey", "spos", "flag", "cum", "first_pos", "seg_id", "seg_key", "first_alt", "sorted_seg", "type_of_seg",
          "len_seen", "sizes", "relation", "bank", "length", "nalt", "cum_alt", "cmax_alt", "nalt64")


def _table(bufs):
    return (ctypes.c_void_p * len(_TABLE))(*[(bufs[n].data_ptr() if n in bufs else None) for n in _TABLE])


def _geom(g):
    vals = [int(g.get(k, 0)) for k in _GEOM]
    return (ctypes.c_int64 * len(_GEOM))(*[v - (1 << 64) if v >= 1 << 63 else v for v in vals])    # (the seed is a uint64 bit pattern)


def graphs_csr(graphs):
    """graphs: list of (n_nodes, root, edges[E,3]) as relbatch.build_relation_batch takes them.  Returns the flattened graphs (numpy
    arrays named as the pointer table) plus 'depth' and the integers B, S, P, nmax, emax."""
    lib = relbatch.load()
    B = len(graphs)
    n_nodes = np.array([g[0] for g in graphs], dtype=np.int32)
    roots = np.array([g[1] for g in graphs], dtype=np.int32)
    edges = [np.asarray(g[2], dtype=np.int32).reshape(-1, 3) for g in graphs]
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(e) for e in edges])
    allE = np.concatenate(edges) if off[-1] else np.zeros((0, 3), np.int32)
    src, dst, lab = [np.ascontiguousarray(allE[:, k]) for k in range(3)]
    if B == 0 or int(n_nodes.min()) <= 0:
        raise ValueError("every graph needs at least one node")
    S, E = int(n_nodes.sum()), int(off[-1])
    out = dict(ng=n_nodes, node_off=np.zeros(B + 1, np.int32), adj_base=np.zeros(B + 1, np.int32), adj_off=np.zeros(S + B, np.int32),
               adj_dst=np.zeros(max(1, E), np.int32), adj_lab=np.zeros(max(1, E), np.int32), order=np.zeros(S, np.int32),
               depth=np.zeros(S, np.int32))
    rc = lib.gtos_relbatch_csr(B, n_nodes.ctypes.data, roots.ctypes.data, off.ctypes.data, src.ctypes.data, dst.ctypes.data, lab.ctypes.data,
                               *[out[k].ctypes.data for k in ("node_off", "adj_base", "adj_off", "adj_dst", "adj_lab", "order", "depth")])
    if rc < 0:
        raise ValueError("gtos_relbatch_csr rejected the batch (disconnected graph, label id outside [1,255], or a graph too large)")
    pair_off = np.zeros(B + 1, np.int64)
    pair_off[1:] = np.cumsum(n_nodes.astype(np.int64) ** 2)
    out["pair_off"] = pair_off
    deg = np.diff(out["adj_base"])
    out.update(B=B, S=S, P=int(pair_off[-1]), nmax=int(n_nodes.max()), emax=max(1, int(deg.max())))
    return out


class HipBackend(object):
    """gtos_relbatch_dev_phase_a / _b of libgtos_hip.so on the current stream."""
    _shared = None
    needs_device = True            # its buffers must live on the GPU (data.attach_device_relations falls back to the host builder otherwise)

    def __init__(self):
        from ._lib import load, stream
        self._lib, self._stream = load(), stream
        self._ws = None

    @classmethod
    def shared(cls):
        if cls._shared is None:
            cls._shared = cls()
        return cls._shared

    @classmethod
    def backend_needs_device(cls):
        """Whether the backend in use takes device pointers -- WITHOUT loading libgtos_hip.so: a host-side consumer asks this to decide
        for the C++ host builder, and must get its answer on a box where the HIP library cannot be loaded.  (The tests' emulation
        backends are installed as ``_shared`` and do not set ``needs_device``.)"""
        return bool(getattr(cls._shared if cls._shared is not None else cls, "needs_device", False))

    def _workspace(self, total, dev):
        out = ctypes.c_int64(0)
        if self._lib.gtos_relbatch_dev_workspace(total, ctypes.byref(out)):
            raise RuntimeError("gtos_relbatch_dev_workspace rejected %d elements" % total)
        need = int(out.value)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def phase_a(self, geom, bufs, total):
        ws = self._workspace(total, bufs["key"].device)
        rc = self._lib.gtos_relbatch_dev_phase_a(_geom(geom), _table(bufs), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_relbatch_dev_phase_a failed: %d" % rc)

    def phase_b(self, geom, R, bufs, total):
        ws = self._workspace(total, bufs["key"].device)
        rc = self._lib.gtos_relbatch_dev_phase_b(_geom(geom), R, _table(bufs), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_relbatch_dev_phase_b failed: %d" % rc)

    def all_phase(self, which, geom, bufs, n, R=0):
        """the three phases of the every-shortest-path mode: 'count', 'keys', 'fill'"""
        ws = self._workspace(n, bufs["nalt"].device)
        fn = getattr(self._lib, "gtos_relbatch_dev_all_" + which)
        rc = (fn(_geom(geom), R, _table(bufs), ws.data_ptr(), ws.numel(), self._stream()) if which == "fill" else
              fn(_geom(geom), _table(bufs), ws.data_ptr(), ws.numel(), self._stream()))
        if rc:
            raise RuntimeError("gtos_relbatch_dev_all_%s failed: %d" % (which, rc))


def build_relation_batch_staged(graphs, special_ids, backend, path_mode=PATH_FIRST, seed=0, max_len=8, device="cpu", csr=None):
    """As ``relbatch.build_relation_batch`` (same arguments, same dict of tensors: relation [n,n,B], relation_bank [L,R],
    relation_length [R] on ``device``; order / depth [B, n-1] on the host), for PATH_FIRST / PATH_UNIFORM.  ``csr``: the result of
    ``graphs_csr(graphs)`` when the loader made it ahead (a worker, or the previous step)."""
    if path_mode not in (PATH_FIRST, PATH_UNIFORM):
        raise ValueError("the staged builder covers the one-path-per-pair modes; the eval-mode enumeration is the host builder's")
    if not 1 <= max_len <= 8:
        raise ValueError("max_len must be in 1..8")
    pad, cls, rcls, self_, tl = [int(v) for v in special_ids]
    if len({cls, rcls, self_}) != 3 or min(cls, rcls, self_, tl) < 0 or max(cls, rcls, self_, tl) > 255:
        raise ValueError("<CLS>, <rCLS>, <SELF> must be three different one-byte ids")
    c = graphs_csr(graphs) if csr is None else csr
    dev = torch.device(device)
    B, S, P, nmax, emax = c["B"], c["S"], c["P"], c["nmax"], c["emax"]
    n = nmax + 1
    total = P + 3
    if 8 * total >= 1 << 32:                    # 32-bit positions, and the bank's row count (<= 8 per pair) rides in half a 64-bit word
        raise ValueError("too many pairs for the builder's 32-bit counters")
    geom = dict(B=B, n=n, nmax=nmax, emax=emax, max_len=max_len, mode=1 if path_mode == PATH_UNIFORM else 0, S=S, P=P,
                seed=int(seed) & 0xffffffffffffffff, cls=cls, rcls=rcls, tl=tl)
    geom["self"] = self_
    i8, i16, i32, i64, f64 = torch.uint8, torch.int16, torch.int32, torch.int64, torch.float64

    def E(shape, dt):
        return torch.empty(shape, dtype=dt, device=dev)
    # the graphs: ONE upload of the concatenated int32 arrays (pair_off separately: int64)
    names = ("ng", "node_off", "adj_base", "adj_off", "adj_dst", "adj_lab", "order")
    flat_host = torch.from_numpy(np.concatenate([c[k] for k in names]))      # (kept alive until the host read below: the copies
    pair_host = torch.from_numpy(c["pair_off"])                               #  are queued, not necessarily done, when .to returns)
    flat = flat_host.to(dev, non_blocking=True)
    bufs, at = {}, 0
    for k in names:
        bufs[k] = flat[at:at + c[k].size]
        at += c[k].size
    bufs["pair_off"] = pair_host.to(dev, non_blocking=True)
    bufs.update(level=E((S, nmax), i16), count=E((S, nmax), f64), head=E((S, nmax), i16), tail=E((S, nmax), i16), queue=E((S, nmax), i16),
                dpred=E((S, emax), i16), dnext=E((S, emax), i16), dlab=E((S, emax), i8),
                key=E(total, i64), posn=E(total, i32), skey=E(total, i64), spos=E(total, i32), flag=E(total, i64), cum=E(total, i64),
                first_pos=E(total, i32), seg_id=E(total, i32), seg_key=E(total, i64), len_seen=E(8, i32), sizes=E(8, i32),
                relation=torch.zeros((n, n, B), dtype=i64, device=dev))
    backend.phase_a(geom, bufs, total)
    R, L, N = bufs["sizes"][:3].tolist()                                  # the one host read: distinct paths, longest, bank rows
    del flat_host, pair_host
    bufs.update(first_alt=E(R, i32), sorted_seg=E(R, i32), type_of_seg=E(R, i32), bank=torch.zeros((8, R), dtype=i64, device=dev),
                length=E(R, i64))
    backend.phase_b(geom, R, bufs, total)
    order = np.full((B, n - 1), -1, np.int32)
    depth = np.zeros((B, n - 1), np.int32)
    for b in range(B):
        lo, hi = int(c["node_off"][b]), int(c["node_off"][b + 1])
        order[b, :hi - lo] = c["order"][lo:hi]
        depth[b, :hi - lo] = c["depth"][lo:hi]
    return dict(relation=bufs["relation"], relation_bank=bufs["bank"][:L], relation_length=bufs["length"], relation_rows=N,
                order=torch.from_numpy(order), depth=torch.from_numpy(depth))


def build_relation_batch_all_staged(graphs, special_ids, backend, max_len=8, device="cpu", csr=None):
    """``relbatch.build_relation_batch(..., path_mode=PATH_ALL)`` on the device: EVERY shortest path of every pair in networkx's
    enumeration order (generator/data.py:178-232, the eval-mode batches): relation [n,n,B,K] with type 0 = <PAD> behind a pair's
    last alternative, relation_bank [L,R], relation_length [R].  Three phases, two host reads (paths in total and most per pair; then R,
    L, N)."""
    if not 1 <= max_len <= 8:
        raise ValueError("max_len must be in 1..8")
    pad, cls, rcls, self_, tl = [int(v) for v in special_ids]
    if len({pad, cls, rcls, self_}) != 4 or min(pad, cls, rcls, self_, tl) < 0 or max(pad, cls, rcls, self_, tl) > 255:
        raise ValueError("<PAD>, <CLS>, <rCLS>, <SELF> must be four different one-byte ids")
    c = graphs_csr(graphs) if csr is None else csr
    dev = torch.device(device)
    B, S, P, nmax, emax = c["B"], c["S"], c["P"], c["nmax"], c["emax"]
    n = nmax + 1
    geom = dict(B=B, n=n, nmax=nmax, emax=emax, max_len=max_len, mode=2, S=S, P=P, seed=0, cls=cls, rcls=rcls, tl=tl, T=0, K=0, pad=pad)
    geom["self"] = self_
    i8, i16, i32, i64, f64 = torch.uint8, torch.int16, torch.int32, torch.int64, torch.float64

    def E(shape, dt):
        return torch.empty(shape, dtype=dt, device=dev)
    names = ("ng", "node_off", "adj_base", "adj_off", "adj_dst", "adj_lab", "order")
    flat_host = torch.from_numpy(np.concatenate([c[k] for k in names]))
    pair_host = torch.from_numpy(c["pair_off"])
    flat = flat_host.to(dev, non_blocking=True)
    bufs, at = {}, 0
    for k in names:
        bufs[k] = flat[at:at + c[k].size]
        at += c[k].size
    bufs["pair_off"] = pair_host.to(dev, non_blocking=True)
    bufs.update(level=E((S, nmax), i16), count=E((S, nmax), f64), head=E((S, nmax), i16), tail=E((S, nmax), i16), queue=E((S, nmax), i16),
                dpred=E((S, emax), i16), dnext=E((S, emax), i16), dlab=E((S, emax), i8), nalt=E(P, i32), cum_alt=E(P, i64), cmax_alt=E(P, i32), nalt64=E(P, i64),
                len_seen=E(8, i32), sizes=torch.zeros(8, dtype=i32, device=dev))
    backend.all_phase("count", geom, bufs, P)
    T, K = bufs["sizes"][3:5].tolist()                                    # host read 1: paths in total, most of one pair
    del flat_host, pair_host
    if T < 0 or 8 * (T + 4) >= 1 << 32:
        raise ValueError("too many shortest paths for the builder's 32-bit counters")
    geom["T"], geom["K"] = T, K
    total = T + 4
    bufs.update(key=E(total, i64), posn=E(total, i32), skey=E(total, i64), spos=E(total, i32), flag=E(total, i64), cum=E(total, i64),
                first_pos=E(total, i32), seg_id=E(total, i32), seg_key=E(total, i64))
    backend.all_phase("keys", geom, bufs, total)
    R, L, N = bufs["sizes"][:3].tolist()                                  # host read 2: distinct paths, longest, bank rows
    bufs.update(first_alt=E(R, i32), sorted_seg=E(R, i32), type_of_seg=E(R, i32), bank=torch.zeros((8, R), dtype=i64, device=dev),
                length=E(R, i64), relation=torch.zeros((n, n, B, K), dtype=i64, device=dev))
    backend.all_phase("fill", geom, bufs, total, R)
    order = np.full((B, n - 1), -1, np.int32)
    depth = np.zeros((B, n - 1), np.int32)
    for b in range(B):
        lo, hi = int(c["node_off"][b]), int(c["node_off"][b + 1])
        order[b, :hi - lo] = c["order"][lo:hi]
        depth[b, :hi - lo] = c["depth"][lo:hi]
    return dict(relation=bufs["relation"], relation_bank=bufs["bank"][:L], relation_length=bufs["length"], relation_rows=N,
                order=torch.from_numpy(order), depth=torch.from_numpy(depth))
