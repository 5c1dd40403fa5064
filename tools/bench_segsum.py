#!/usr/bin/env python
"""Segmented-sum micro-benchmark on the C2 relation tries: the two gate-table gradient reductions (rows -> prefix-trie nodes,
rows -> suffix-trie nodes) of the trie GRU backward, HIP events.   python tools/bench_segsum.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import synth  # noqa: E402
from gtos_amd.gru import _seg_rows  # noqa: E402
from gtos_amd.pathtrie import build_path_trie  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    batch, stats = synth.make_config_batch("C2")
    trie = build_path_trie(batch["relation_bank"], batch["relation_length"]).to(dev)
    hs = 256
    d4 = torch.randn(trie.N, 4 * hs, device=dev).to(torch.bfloat16)
    for name, side in (("prefix", trie.pf), ("suffix", trie.sf)):
        out = torch.empty(side.n_nodes, 3 * hs, dtype=torch.bfloat16, device=dev)
        _seg_rows(side, d4, 3 * hs, out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _seg_rows(side, d4, 3 * hs, out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        byt = trie.N * 3 * hs * 2 + side.n_nodes * 3 * hs * 2
        print("%s trie: %d nodes, %d chunks, %d heavy: %.3f ms  %.2f TB/s" % (name, side.n_nodes, side.n_chunks, side.n_heavy, ms, byt / ms / 1e9))


if __name__ == "__main__":
    main()
