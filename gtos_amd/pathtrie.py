"""Prefix / suffix tries of a batch's relation bank (include/gtos_host.h, csrc_host/pathtrie.cpp).

Index preparation for the trie-evaluated RelationEncoder (gtos_amd/gru.py): built on the HOST from the integer
``relation_bank`` / ``relation_length`` of a batch -- like the bank itself, which the reference's ``batchify`` builds on the
host (generator/data.py:134-176) -- and shipped with the batch as ``batch['relation_trie']``.  A ``PathTrie`` behaves like
a tensor for ``{k: v.to(device) for k, v in batch.items()}``.
"""
import ctypes

import numpy as np
import torch

from . import relbatch

CHUNK = 64
_COMMON = ("batch_sizes", "seq_order", "seq_pos", "row_pf", "row_sf")
_PER_TRIE = ("level_off", "tok", "par", "child_off", "rows", "chunk_node", "chunk_start", "chunk_cnt", "chunk_slot", "heavy_node")


def _lib():
    return relbatch.load()          # every signature of libgtos_host.so is set there, once, under a lock


class TrieSide(object):
    """One trie: int32 index tensors + the level offsets as Python ints (launch geometry, never read from the device).

    Derived at build time (numpy, on the host) for the backward walk of gtos_amd.gru: ``sum_idx`` [n] -- the row of the
    ``[n + 1 + n_multi, .]`` gradient buffers that holds the SUM over node u's children: the child itself when there is one, row
    n (all zero) for a leaf, row n + 1 + j for the j-th node with several children; ``multi_ranges`` [2 * n_multi] -- the child
    ranges of those nodes in node order (so level-major); ``multi_level_off`` [L + 1] -- how many of them precede each level.
    For the streaming segmented sum (gtos_segment_sum_stream): ``wave_off`` [n_waves + 1] -- chunk indices cutting the chunk list
    (a CSR over ``rows``) into ranges of about ``ROWS_PER_WAVE`` rows."""
    ROWS_PER_WAVE = 256

    def __init__(self, arrays, level_off, multi_level_off=None):
        self.__dict__.update(arrays)
        if self.tok.dtype != torch.int64:
            self.tok = self.tok.to(torch.int64)        # the embedding kernels take int64 token ids
        if "par_long" not in arrays:
            self.par_long = self.par.to(torch.int64)   # index_select operand (the parent-state gather of the weight gradient)
        self.level_off = list(level_off)
        self.n_nodes = self.level_off[-1]
        self.n_chunks = int(self.chunk_node.numel())
        self.n_heavy = int(self.heavy_node.numel())
        if "sum_idx" not in arrays:
            n = self.n_nodes
            co = self.child_off.cpu().numpy().reshape(-1, 2)[:n].astype(np.int64)
            nc = co[:, 1] - co[:, 0]
            multi = nc >= 2
            slot = np.cumsum(multi) - 1
            idx = np.where(nc == 1, co[:, 0], np.where(multi, n + 1 + slot, n))
            self.sum_idx = torch.from_numpy(idx.astype(np.int32))
            self.multi_ranges = torch.from_numpy(np.ascontiguousarray(co[multi].astype(np.int32).reshape(-1)))
            before = np.concatenate([[0], np.cumsum(multi)])
            multi_level_off = [int(before[o]) for o in self.level_off]
        self.multi_level_off = list(multi_level_off)
        self.n_multi = self.multi_level_off[-1]
        if "wave_off" not in arrays:
            starts = self.chunk_start.cpu().numpy().astype(np.int64)
            total = int(self.rows.numel())
            n_waves = max(1, -(-total // self.ROWS_PER_WAVE))
            off = np.searchsorted(starts, np.arange(n_waves, dtype=np.int64) * self.ROWS_PER_WAVE, side="left")
            off[0] = 0                   # (chunks without rows in front of the first row belong to the first wave)
            self.wave_off = torch.from_numpy(np.concatenate([off, [self.n_chunks]]).astype(np.int32))
        self.n_waves = int(self.wave_off.numel()) - 1

    def to(self, device):
        arrays = {k: v.to(device) for k, v in self.__dict__.items() if isinstance(v, torch.Tensor) and not k.startswith("_")}
        return TrieSide(arrays, self.level_off, self.multi_level_off)

    def token_onehot(self, width, dtype):
        """[n_nodes, width] one-hot rows of the node tokens (width >= vocabulary size, a multiple of 8), built once per batch:
        the left operand that turns the label-embedding gradient into a product on the MFMA GEMM (gtos_amd.gru)."""
        oh = getattr(self, "_onehot", None)
        if oh is None or oh.shape[1] != width or oh.dtype != dtype or oh.device != self.tok.device:
            oh = torch.zeros((self.n_nodes, width), dtype=dtype, device=self.tok.device)
            oh.scatter_(1, self.tok.view(-1, 1), 1.0)
            self._onehot = oh
        return oh


class PathTrie(object):
    def __init__(self, L, R, N, batch_sizes, common, pf, sf):
        self.L, self.R, self.N = L, R, N
        self.batch_sizes = list(batch_sizes)
        self.seq_order, self.seq_pos, self.row_pf, self.row_sf = common[:4]
        # int32 copy of seq_order for the step kernel's scatter of the final states (packed row m -> bank row seq_order[m])
        self.seq_order32 = common[4] if len(common) > 4 else self.seq_order.to(torch.int32)
        self.pf, self.sf = pf, sf

    def to(self, device, *a, **k):
        common = tuple(t.to(device) for t in (self.seq_order, self.seq_pos, self.row_pf, self.row_sf, self.seq_order32))
        return PathTrie(self.L, self.R, self.N, self.batch_sizes, common, self.pf.to(device), self.sf.to(device))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    @property
    def device(self):
        return self.row_pf.device

    def matches(self, src_tokens, src_lengths):
        return src_tokens.shape[1] == self.R and self.device == src_tokens.device


def build_path_trie(bank, length, chunk=CHUNK):
    """bank: int64 [L,R] (relation_bank), length: int64 [R]; CPU tensors (device tensors are copied to the host)."""
    bank = np.ascontiguousarray(bank.detach().cpu().numpy().astype(np.int64, copy=False))
    length = np.ascontiguousarray(length.detach().cpu().numpy().astype(np.int64, copy=False))
    L, R = bank.shape
    if not 1 <= chunk <= 64:
        raise ValueError("chunk must be in 1..64: gtos_segment_sum_rows reads one row id per lane of a 64-lane wave")
    lib = _lib()
    h = lib.gtos_pathtrie_build(L, R, bank.ctypes.data, length.ctypes.data, chunk)
    if not h:
        raise ValueError("gtos_pathtrie_build rejected the bank (at most 64 labels per path, lengths in 1..L, label ids non-negative)")
    try:
        sizes = np.zeros(9, dtype=np.int64)
        lib.gtos_pathtrie_sizes(h, sizes.ctypes.data)
        Lm, R_, N, nPF, nSF, cPF, hPF, cSF, hSF = [int(v) for v in sizes]
        shapes = [Lm, R, R, N, N]
        for n, c, hv in ((nPF, cPF, hPF), (nSF, cSF, hSF)):
            shapes += [Lm + 1, n, n, 2 * n, N, c, c, c, c, hv]
        arrs = [np.empty(max(1, s), dtype=np.int32) for s in shapes]       # the export fills every array completely
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        got = lib.gtos_pathtrie_export(h, ptrs)
        assert got == len(arrs)
        # the indices of the backward walk and of the streaming segmented sum (TrieSide derives them with numpy when they are missing)
        dsz = np.zeros(4, dtype=np.int64)
        lib.gtos_pathtrie_derived_sizes(h, TrieSide.ROWS_PER_WAVE, dsz.ctypes.data)
        dshapes = []
        for n, (nm, nw) in zip((nPF, nSF), ((int(dsz[0]), int(dsz[1])), (int(dsz[2]), int(dsz[3])))):
            dshapes += [n, 2 * nm, Lm + 1, nw + 1]
        darrs = [np.empty(max(1, s), dtype=np.int32) for s in dshapes]
        dptrs = (ctypes.c_void_p * len(darrs))(*[a.ctypes.data for a in darrs])
        assert lib.gtos_pathtrie_export_derived(h, TrieSide.ROWS_PER_WAVE, dptrs) == len(darrs)
    finally:
        lib.gtos_pathtrie_free(h)
    nps = [a[:s] for a, s in zip(arrs, shapes)]
    ts = [torch.from_numpy(a) for a in nps]

    def wide(i):                                   # int64 copy made by numpy (torch's CPU int32 -> int64 cast is ~50x slower here)
        return torch.from_numpy(nps[i].astype(np.int64))
    common = dict(zip(_COMMON, ts[:5]))
    sides = []
    for k in (0, 1):
        names = dict(zip(_PER_TRIE, range(5 + 10 * k, 15 + 10 * k)))
        d = {n: ts[i] for n, i in names.items()}
        d["tok"] = wide(names["tok"])              # the embedding kernels take int64 token ids
        d["par_long"] = wide(names["par"])         # index_select operand (the parent-state gather of the weight gradient)
        level_off = d.pop("level_off").tolist()
        dv = [torch.from_numpy(a[:s_]) for a, s_ in zip(darrs[4 * k:4 * k + 4], dshapes[4 * k:4 * k + 4])]
        d["sum_idx"], d["multi_ranges"], d["wave_off"] = dv[0], dv[1], dv[3]
        sides.append(TrieSide(d, level_off, dv[2].tolist()))
    # seq_order / seq_pos feed index_select (int64); the row -> node maps stay int32 for the kernels
    return PathTrie(Lm, R, N, common["batch_sizes"].tolist(), (wide(1), wide(2), common["row_pf"], common["row_sf"], ts[1]),
                    sides[0], sides[1])


def attach_path_trie(batch):
    """Adds ``batch['relation_trie']`` (host side, before the batch moves to the device) and returns the batch."""
    batch['relation_trie'] = build_path_trie(batch['relation_bank'], batch['relation_length'])
    return batch
