#!/usr/bin/env python
"""GEMM micro-benchmark through the C ABI (HIP events on the launch stream).

    python tools/bench_gemm.py [--reps 5]
Shapes are the dominant GEMMs of one C2 training step (relation GRU and relation projection)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import ops  # noqa: E402

SHAPES = [  # name, ta, tb, M, N, K, out dtype
    ("gru_xg_l1   NT", False, True, 2497000, 768, 512, torch.bfloat16),
    ("gru_xg_l0   NT", False, True, 2497000, 768, 104, torch.bfloat16),
    ("gru_hg      NT", False, True, 400000, 768, 256, torch.bfloat16),
    ("rel_proj    NT", False, True, 434624, 1024, 512, torch.bfloat16),
    ("gru_dx_l1   NN", False, False, 2497000, 512, 768, torch.bfloat16),
    ("rel_dbank   NN", False, False, 434624, 512, 1024, torch.bfloat16),
    ("gru_dh      NNf", False, False, 400000, 256, 768, torch.float32),
    ("gru_dwih_l1 TN", True, False, 768, 512, 2497000, torch.float32),
    ("rel_dw      TN", True, False, 1024, 512, 434624, torch.float32),
    ("gru_tables  NT", False, True, 434624, 768, 256, torch.bfloat16),
    ("relenc_out  NT", False, True, 434624, 512, 512, torch.bfloat16),
    ("ffn_fc1     NT", False, True, 6464, 1024, 512, torch.bfloat16),
    # the few-thousand-row products of the graph layers (n*B = 6464 rows) and of the decoder (T*B = 3200 rows)
    ("small_out   NT", False, True, 6464, 512, 512, torch.bfloat16),
    ("small_qkv   NT", False, True, 6464, 1536, 512, torch.bfloat16),
    ("small_fc2   NT", False, True, 6464, 512, 1024, torch.bfloat16),
    ("small_dec   NT", False, True, 3200, 512, 512, torch.bfloat16),
    ("small_dec1k NT", False, True, 3200, 1024, 512, torch.bfloat16),
    ("tn_m1024_Kbig TN", True, False, 1024, 512, 2497000, torch.float32),
    ("tn_m768_Ksml  TN", True, False, 768, 512, 434624, torch.float32),
    ("deepK       NT", False, True, 434624, 1024, 4096, torch.bfloat16),     # same tile count as rel_proj, 8x the k tiles
    ("deepK_dbank NT", False, True, 434624, 512, 8192, torch.bfloat16),     # the bank-gradient slab product of a C2 step
    ("square8k    NT", False, True, 8192, 8192, 8192, torch.bfloat16),
    # t(K) at the relation-projection shape: intercept = output write + launch, slope = k-loop rate
    ("ksweep32    NT", False, True, 434624, 1024, 32, torch.bfloat16),
    ("ksweep128   NT", False, True, 434624, 1024, 128, torch.bfloat16),
    ("ksweep256   NT", False, True, 434624, 1024, 256, torch.bfloat16),
    ("ksweep512   NT", False, True, 434624, 1024, 512, torch.bfloat16),
    ("ksweep1024  NT", False, True, 434624, 1024, 1024, torch.bfloat16),
    ("ksweep2016  NT", False, True, 434624, 1024, 2016, torch.bfloat16),
    # the same 0.89 GB of output with narrower rows: a 128-wide tile writes whole rows at N = 128
    ("ksweep32n128 NT", False, True, 434624 * 8, 128, 32, torch.bfloat16),
    ("ksweep32n256 NT", False, True, 434624 * 4, 256, 32, torch.bfloat16),
    ("ksweep32n4096 NT", False, True, 434624 // 4, 4096, 32, torch.bfloat16),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--torch", action="store_true", help="also time torch.matmul (hipBLASLt) on the same operands")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, ta, tb, M, N, K, od in SHAPES:
        if a.only and a.only not in name:
            continue
        A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        B = torch.randn((N, K) if tb else (K, N), device=dev).to(torch.bfloat16)
        out = torch.zeros((M, N), device=dev, dtype=od)
        sk = ops._splitk(M, N, K) if ta else 1
        acc = od == torch.float32
        ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, accumulate=acc, splitk=sk)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = a.reps * (20 if M < 20000 else 1)            # small products: enough launches for the event pair to resolve them
        s.record()
        for _ in range(reps):
            ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, accumulate=acc, splitk=sk)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        flops = 2.0 * M * N * K
        byts = (M * K + N * K) * 2 + M * N * out.element_size()
        print("%-16s M=%8d N=%5d K=%8d splitk=%2d  %8.3f ms  %7.1f TF/s  min-traffic %6.2f GB -> %5.2f TB/s" % (
            name, M, N, K, sk, ms, flops / ms / 1e9, byts / 1e9, byts / ms / 1e9), flush=True)
        if a.torch:
            At, Bt = (A.t() if ta else A), (B.t() if tb else B)
            ref = torch.matmul(At, Bt)
            torch.cuda.synchronize()
            s.record()
            for _ in range(a.reps):
                ref = torch.matmul(At, Bt)
            e.record()
            torch.cuda.synchronize()
            mt = s.elapsed_time(e) / a.reps
            print("%-16s   torch.matmul (bf16 out)        %8.3f ms  %7.1f TF/s" % ("", mt, flops / mt / 1e9), flush=True)
            del ref
        del A, B, out


if __name__ == "__main__":
    main()
