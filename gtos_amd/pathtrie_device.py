"""The prefix / suffix tries of a relation bank built with torch ops on the bank's own device.

Same result, array for array, as the host builder (csrc_host/pathtrie.cpp through gtos_amd.pathtrie.build_path_trie;
tests/test_pathtrie.py compares them on the CPU), for the case the host builder's fast path covers: at most 8 labels per path and
label ids below 255, i.e. every bank the reference's own data produces (generator/data.py:134-176 caps the paths at 8 labels).  What
the host does with one radix sort, an LCP pass and two counting sorts per trie is here a key sort, a cumulative sum over the
[R, L] "opens a new node" matrix and two more sorts -- all of it R- or N-sized device work (R = 434,624, N = 2.5 M at C2), so a batch
that arrives without ``relation_trie`` can get its tries on the GPU instead of 0.1-0.2 s of host time per batch and worker.

Opt-in (``GTOS_TRIE_DEVICE=1`` in gtos_amd.data.Prefetcher / gtos_amd.encoder): round 3 ended before this path could be timed on
the GPU; on the CPU it is slower than the C++ builder and only the equality with it is tested.  The calls that need a size on the
host (``nonzero``, ``repeat_interleave``, ``tolist``) synchronise the stream they run on -- run it on the copy stream of the loader.
"""
import torch

from .pathtrie import CHUNK, PathTrie, TrieSide

_SIGN = -2 ** 63


def _cumsum0(x):
    """exclusive-prefix form [0, x0, x0+x1, ...] (one element longer)"""
    return torch.cat([x.new_zeros(1), torch.cumsum(x, 0)])


def _one_side(tp1, length, valid, seq_pos, offs, reverse, N, chunk):
    """tp1 [R,L]: label + 1 at the valid positions, 0 behind the end (for the suffix trie: of the REVERSED paths)."""
    dev = tp1.device
    R, L = tp1.shape
    ar = torch.arange(L, device=dev)
    # lexicographic order, a proper prefix first: the bytes (label + 1), most significant first, as one unsigned 64-bit key;
    # torch sorts signed, so the sign bit is flipped.  Stable in the sequence id, like the host's LSD radix sort.
    key = (tp1 << (8 * (7 - ar))[None, :]).sum(1)
    order = torch.sort(key ^ _SIGN, stable=True).indices
    ts, ls = tp1[order], length[order]
    # sequence i shares lcp(i-1, i) nodes with its predecessor in the sorted order and opens a node at every later position
    eq = (ts[1:] == ts[:-1]).to(torch.int64)
    lcp = torch.cat([ls.new_zeros(1), torch.cumprod(eq, 1).sum(1)])
    new = (ar[None, :] >= lcp[:, None]) & (ar[None, :] < ls[:, None])                  # [R, L]
    level_off_t = _cumsum0(new.sum(0))                                                  # [L + 1]
    # (the running count down the R sorted sequences as an INNERMOST-dimension scan of the transposed matrix: ATen's outer-dimension
    #  scan of a tall [R, 8] tensor took 79 ms per trie on the MI355X, 93 % of the whole build)
    opened = torch.cumsum(new.t().contiguous().to(torch.int64), 1).t()
    nid = level_off_t[:-1][None, :] + opened - 1                                         # node of sorted sequence i at level k
    level_off = level_off_t.tolist()                                                    # (sync) launch geometry lives on the host
    n = level_off[-1]
    ids = nid[new]
    tok = torch.empty(n, dtype=torch.int64, device=dev)
    tok[ids] = ts[new] - 1
    prev = torch.cat([torch.full((R, 1), n, dtype=torch.int64, device=dev), nid[:, :-1]], 1)
    par = torch.empty(n, dtype=torch.int64, device=dev)
    par[ids] = prev[new]
    # children of a node: a contiguous range of the next level; (0, 0) for a leaf
    v = torch.arange(n, device=dev)
    inner = par < n
    big = torch.full((n,), n, dtype=torch.int64, device=dev)
    lo = big.scatter_reduce(0, par[inner], v[inner], "amin", include_self=True)
    hi = torch.zeros(n, dtype=torch.int64, device=dev).scatter_reduce(0, par[inner], v[inner] + 1, "amax", include_self=True)
    lo = torch.where(hi > 0, lo, torch.zeros_like(lo))
    child_off = torch.stack([lo, hi], 1).reshape(-1)
    # node of every packed row: (sequence s, level k) sits at offs[pos] + seq_pos[s], pos = k (prefix) or len - 1 - k (suffix)
    inside = ar[None, :] < ls[:, None]
    pos = (ls[:, None] - 1 - ar[None, :]) if reverse else ar[None, :].expand(R, L)
    row_idx = offs[pos.clamp(min=0)] + seq_pos[order][:, None]
    row_node = torch.empty(N, dtype=torch.int64, device=dev)
    row_node[row_idx[inside]] = nid[inside]
    # rows of every node (CSR, stable in the row id), cut into chunks of <= chunk rows; a node without rows keeps one empty chunk
    rows = torch.sort(row_node, stable=True).indices
    cnt = torch.bincount(row_node, minlength=n)
    start = torch.cumsum(cnt, 0) - cnt
    nch = torch.clamp((cnt + chunk - 1) // chunk, min=1)
    chunk_node = torch.repeat_interleave(v, nch)                                        # (sync)
    local = torch.arange(chunk_node.numel(), device=dev) - (torch.cumsum(nch, 0) - nch)[chunk_node]
    chunk_start = start[chunk_node] + local * chunk
    chunk_cnt = torch.clamp(cnt[chunk_node] - local * chunk, min=0, max=chunk)
    heavy_node = torch.nonzero(nch > 1).flatten()                                       # (sync)
    slot_of = torch.full((n,), -1, dtype=torch.int64, device=dev)
    slot_of[heavy_node] = torch.arange(heavy_node.numel(), device=dev)
    chunk_slot = slot_of[chunk_node]
    # what TrieSide derives with numpy for a host-built trie (gtos_amd.pathtrie)
    nc = hi - lo
    multi = nc >= 2
    before = _cumsum0(multi.to(torch.int64))
    sum_idx = torch.where(nc == 1, lo, torch.where(multi, n + 1 + before[:-1], torch.full_like(lo, n)))
    multi_ranges = torch.stack([lo[multi], hi[multi]], 1).reshape(-1)                   # (sync)
    multi_level_off = before[level_off_t].tolist()                                      # (sync)
    n_waves = max(1, -(-N // TrieSide.ROWS_PER_WAVE))
    wave_off = torch.searchsorted(chunk_start.contiguous(), torch.arange(n_waves, device=dev) * TrieSide.ROWS_PER_WAVE, right=False)
    wave_off[0] = 0
    wave_off = torch.cat([wave_off, wave_off.new_full((1,), chunk_node.numel())])
    i32 = lambda t: t.to(torch.int32)
    arrays = dict(tok=tok, par=i32(par), par_long=par, child_off=i32(child_off), rows=i32(rows), chunk_node=i32(chunk_node),
                  chunk_start=i32(chunk_start), chunk_cnt=i32(chunk_cnt), chunk_slot=i32(chunk_slot), heavy_node=i32(heavy_node),
                  sum_idx=i32(sum_idx), multi_ranges=i32(multi_ranges), wave_off=i32(wave_off))
    return TrieSide(arrays, level_off, multi_level_off), i32(row_node), order, ls


def build_path_trie_device(bank, length, chunk=CHUNK):
    """bank: int64 [L,R] (relation_bank), length: int64 [R], on any device.  ValueError outside the covered case (paths longer than
    8 labels, label ids >= 255 or negative, lengths outside 1..L): the caller falls back to the host builder."""
    if not 1 <= chunk <= 64:
        raise ValueError("chunk must be in 1..64: gtos_segment_sum_rows reads one row id per lane of a 64-lane wave")
    dev = bank.device
    bank, length = bank.to(torch.int64), length.to(torch.int64)
    L, R = bank.shape
    ar = torch.arange(L, device=dev)
    valid = ar[None, :] < length[:, None]                                               # [R, L]
    tok = torch.where(valid, bank.t(), torch.zeros((), dtype=torch.int64, device=dev))
    lmin, lmax, tmin, tmax, N = torch.stack([length.min(), length.max(), tok.min(), tok.max(), length.sum()]).tolist()   # (sync)
    if lmin < 1 or lmax > L or lmax > 8 or tmin < 0 or tmax >= 255 or N > 0x7fffffff:
        raise ValueError("build_path_trie_device covers paths of 1..8 labels with ids in [0, 255)")
    if L > 8:
        tok, valid, ar, L = tok[:, :8], valid[:, :8], ar[:8], 8
    one = valid.to(torch.int64)
    tp1_f = tok + one
    back = (length[:, None] - 1 - ar[None, :]).clamp(min=0)
    tp1_b = torch.gather(tp1_f, 1, back) * one
    # packed order of the second GRU layer: length descending, then lexicographic
    key_f = (tp1_f << (8 * (7 - ar))[None, :]).sum(1)
    ord_f = torch.sort(key_f ^ _SIGN, stable=True).indices
    seq_order = ord_f[torch.sort(-length[ord_f], stable=True).indices]
    seq_pos = torch.empty(R, dtype=torch.int64, device=dev)
    seq_pos[seq_order] = torch.arange(R, device=dev)
    longer = torch.flip(torch.cumsum(torch.flip(torch.bincount(length, minlength=L + 1), [0]), 0), [0])     # [l] = #sequences of length >= l
    batch_sizes_t = longer[1:lmax + 1]                                                  # [t] = #sequences longer than t
    offs = _cumsum0(longer[1:])                                                          # [L + 1]
    pf, row_pf, _, _ = _one_side(tp1_f, length, valid, seq_pos, offs, False, N, chunk)
    sf, row_sf, _, _ = _one_side(tp1_b, length, valid, seq_pos, offs, True, N, chunk)
    for side in (pf, sf):                                                                # level offsets beyond the longest path collapse
        side.level_off = side.level_off[:lmax + 1]
        side.multi_level_off = side.multi_level_off[:lmax + 1]
    return PathTrie(lmax, R, N, batch_sizes_t.tolist(), (seq_order, seq_pos, row_pf, row_sf, seq_order.to(torch.int32)), pf, sf)
