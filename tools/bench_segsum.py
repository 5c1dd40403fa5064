"""Micro-benchmark of the segmented sums of the trie GRU backward at C2 size: row stride 1024 (the d4 layout: 3h of 4h read) against a
compact stride 768, streaming kernel against the wave-per-chunk kernel.  python tools/bench_segsum.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import synth                                  # noqa: E402
from gtos_amd import gru as gru_mod                         # noqa: E402
from gtos_amd.pathtrie import build_path_trie               # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    batch, stats = synth.make_config_batch("C2", rank=0, B=64)
    trie = build_path_trie(batch["relation_bank"], batch["relation_length"]).to(dev)
    N = trie.N
    print("rows", N, "pf nodes", trie.pf.n_nodes, "sf nodes", trie.sf.n_nodes, flush=True)
    for ld in (1024, 768, 1280):
        src = torch.randn(N, ld, device=dev).to(torch.bfloat16)
        for name, side in (("prefix", trie.pf), ("suffix", trie.sf)):
            out = torch.empty(side.n_nodes, 768, dtype=torch.bfloat16, device=dev)
            for stream_kernel in (True, False):
                gru_mod.SEG_STREAM = stream_kernel
                for _ in range(3):
                    gru_mod._seg_rows(side, src, 768, out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    gru_mod._seg_rows(side, src, 768, out)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100.0
                gb = (N * 768 * 2 + side.n_nodes * 768 * 2) / 1e9
                print("ld %4d %-6s %-7s %8.1f us  %6.2f TB/s (read + written)" % (ld, name, "stream" if stream_kernel else "chunk", us, gb / us * 1e3),
                      flush=True)
        del src


if __name__ == "__main__":
    main()
