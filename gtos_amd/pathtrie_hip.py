"""The path tries built by the staged GPU builder (csrc/pathtrie_dev.hip; per-thread stages in csrc/trie_kernels.h).

Two phases with one host read between them: phase A sorts the path keys and scans the "opens a node" bits (R-sized work), which
fixes the node counts; the host then allocates the node-sized arrays and phase B fills them (nodes, children, node of every packed
row, rows sorted by node, chunks, heavy nodes, children-sum indices, wave ranges).  The result equals gtos_amd.pathtrie.build_path_trie
array for array; tests/test_pathtrie.py proves that on the CPU by running the SAME stage code as serial host loops (the test suite's
emulation library), so what the GPU adds is rocPRIM's sort / scan and the launch glue.

Status at the end of round 3: opt-in (``Prefetcher(device_tries="hip")``, ``bench.py --device-tries hip``, ``GTOS_TRIE_DEVICE=hip``).
On an MI355X (tools/hip_trie_check.py, profiles/r3w_hip_trie_check.json): equal to the host builder on the C2 bank, 1.5 ms per build
(torch-op builder ~12 ms, host builder 56 ms); ``bench.py --fresh-batches --workers 2 --device-tries hip``: 65.8 ms per C2 step.
``build_path_trie_staged(bank, length, backend)`` takes the backend explicitly: ``HipBackend.shared()`` (libgtos_hip.so) or the
test's emulation backend.
"""
import ctypes

import torch

from .pathtrie import CHUNK, PathTrie, TrieSide

# order of the pointer tables: csrc/trie_kernels.h (enum C_* / T_*)
_COMMON = ("len8", "scratch", "cumlen", "start", "batch", "offs", "seq_order", "seq_pos", "seq_order64", "seq_pos64", "lexf", "lexb",
           "row_pf", "row_sf", "key_alt", "id_alt", "iota")
_SIDE = ("key", "order", "newmask", "cum", "node_tab", "lvl", "tok", "par", "par_long", "child_off", "row_key", "rows", "off", "aux",
         "aux_cum", "chunk_node", "chunk_start", "chunk_cnt", "chunk_slot", "heavy_node", "sum_idx", "multi_ranges", "wave_off")
_SZ_PF, _SZ_SF, _SZ_TOTAL = 1, 33, 65
_S_NODES, _S_CHUNKS, _S_HEAVY, _S_MULTI, _S_LEVEL, _S_MLEVEL = 0, 1, 2, 3, 4, 13


def _table(names, bufs):
    return (ctypes.c_void_p * len(names))(*[(bufs[n].data_ptr() if n in bufs else None) for n in names])


class HipBackend(object):
    """gtos_pathtrie_dev_phase_a / _b of libgtos_hip.so on the current stream."""

    _shared = None

    def __init__(self):
        from ._lib import load, stream
        self._lib, self._stream = load(), stream
        self._ws = None

    @classmethod
    def shared(cls):
        """one backend (and one rocPRIM workspace) per process: the builder runs on one stream at a time"""
        if cls._shared is None:
            cls._shared = cls()
        return cls._shared

    def _workspace(self, R, N, dev):
        out = ctypes.c_int64(0)
        if self._lib.gtos_pathtrie_dev_workspace(R, N, ctypes.byref(out)):
            raise RuntimeError("gtos_pathtrie_dev_workspace rejected R = %d, N = %d" % (R, N))
        need = int(out.value)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def phase_a(self, L, R, N, bank, length, common, pf, sf, sizes):
        ws = self._workspace(R, N, bank.device)
        rc = self._lib.gtos_pathtrie_dev_phase_a(L, R, bank.data_ptr(), length.data_ptr(), _table(_COMMON, common), _table(_SIDE, pf),
                                                 _table(_SIDE, sf), sizes.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_pathtrie_dev_phase_a failed: %d" % rc)

    def phase_b(self, R, N, n_pf, n_sf, chunk, rows_per_wave, common, pf, sf, sizes):
        ws = self._workspace(R, N, sizes.device)
        rc = self._lib.gtos_pathtrie_dev_phase_b(R, N, n_pf, n_sf, chunk, rows_per_wave, _table(_COMMON, common), _table(_SIDE, pf), _table(_SIDE, sf),
                                                 sizes.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        if rc:
            raise RuntimeError("gtos_pathtrie_dev_phase_b failed: %d" % rc)


def build_path_trie_staged(bank, length, backend, chunk=CHUNK, n_rows=None):
    """bank: int64 [L,R], length: int64 [R] on the backend's device; ``n_rows`` = sum(length) when the caller knows it (the loader
    does, on the host), else it is read from the device.  ValueError outside the covered case (as pathtrie_device)."""
    if not 1 <= chunk <= 64:
        raise ValueError("chunk must be in 1..64: gtos_segment_sum_rows reads one row id per lane of a 64-lane wave")
    dev = bank.device
    bank, length = bank.to(torch.int64).contiguous(), length.to(torch.int64).contiguous()
    L, R = bank.shape
    N = int(length.sum()) if n_rows is None else int(n_rows)
    if N > 0x7fffffff or N < R:
        raise ValueError("row count outside the builder's range")
    rpw = TrieSide.ROWS_PER_WAVE
    n_waves = max(1, -(-N // rpw))
    i8, i32, i64 = torch.uint8, torch.int32, torch.int64

    def E(n, dt, cols=None):
        return torch.empty((max(1, n),) if cols is None else (max(1, n), cols), dtype=dt, device=dev)
    sizes = torch.zeros(_SZ_TOTAL, dtype=i32, device=dev)
    common = dict(len8=E(R, i8), scratch=E(R, i32, 8), cumlen=E(R, i32, 8), start=E(8, i32), batch=E(8, i32), offs=E(9, i64),
                  seq_order=E(R, i32), seq_pos=E(R, i32), seq_order64=E(R, i64), seq_pos64=E(R, i64), lexf=E(R, i32), lexb=E(R, i32),
                  row_pf=E(N, i32), row_sf=E(N, i32), key_alt=E(R, i64), id_alt=E(R, i32))
    sides = [dict(key=E(R, i64), order=E(R, i32), newmask=E(R, i8), cum=E(R, i32, 8), node_tab=E(R, i32, 8), lvl=E(9, i32))
             for _ in range(2)]
    backend.phase_a(L, R, N, bank, length, common, sides[0], sides[1], sizes)
    sz = sizes.tolist()                                                   # host read 1: the node counts
    if sz[0]:
        raise ValueError("the staged trie builder covers paths of 1..8 labels with ids in [0, 255)")
    common["iota"] = torch.arange(N, dtype=i32, device=dev)
    for side, base in zip(sides, (_SZ_PF, _SZ_SF)):
        n = sz[base + _S_NODES]
        cub, hub, mub = n + N // chunk + 1, N // chunk + 1, n // 2 + 1
        side.update(tok=E(n, i64), par=E(n, i32), par_long=E(n, i64), child_off=torch.zeros(max(1, 2 * n), dtype=i32, device=dev),
                    row_key=E(N, i32), rows=E(N, i32), off=E(n + 1, i32), aux=E(n, i32, 8), aux_cum=E(n, i32, 8),
                    chunk_node=E(cub, i32), chunk_start=E(cub, i32), chunk_cnt=E(cub, i32), chunk_slot=E(cub, i32),
                    heavy_node=E(hub, i32), sum_idx=E(n, i32), multi_ranges=E(2 * mub, i32), wave_off=E(n_waves + 1, i32))
    backend.phase_b(R, N, sz[_SZ_PF + _S_NODES], sz[_SZ_SF + _S_NODES], chunk, rpw, common, sides[0], sides[1], sizes)
    sz = torch.cat([sizes, common["batch"]]).tolist()                     # host read 2: chunk / heavy / multi counts, batch sizes
    batch_sizes = [b for b in sz[_SZ_TOTAL:] if b > 0]
    lmax = len(batch_sizes)
    out = []
    for side, base in zip(sides, (_SZ_PF, _SZ_SF)):
        n, nc, nh, nm = (sz[base + k] for k in (_S_NODES, _S_CHUNKS, _S_HEAVY, _S_MULTI))
        arrays = dict(tok=side["tok"][:n], par=side["par"][:n], par_long=side["par_long"][:n], child_off=side["child_off"][:2 * n],
                      rows=side["rows"][:N], chunk_node=side["chunk_node"][:nc], chunk_start=side["chunk_start"][:nc],
                      chunk_cnt=side["chunk_cnt"][:nc], chunk_slot=side["chunk_slot"][:nc], heavy_node=side["heavy_node"][:nh],
                      sum_idx=side["sum_idx"][:n], multi_ranges=side["multi_ranges"][:2 * nm], wave_off=side["wave_off"][:n_waves + 1])
        level_off = sz[base + _S_LEVEL:base + _S_LEVEL + 9][:lmax + 1]
        mlo = sz[base + _S_MLEVEL:base + _S_MLEVEL + 9][:lmax + 1]
        out.append(TrieSide(arrays, level_off, mlo))
    return PathTrie(lmax, R, N, batch_sizes, (common["seq_order64"][:R], common["seq_pos64"][:R], common["row_pf"][:N],
                                            common["row_sf"][:N], common["seq_order"][:R]), out[0], out[1])
