"""gemm256q_nt_kernel vs torch.matmul (fp32 accumulate of the same bf16 operands) at ragged M / N and deep K; prints max relative error."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gtos_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
for M, N, K in [(434624, 512, 8192), (131072 + 77, 1024, 1024), (140000, 1000, 4096), (8192, 8192, 8192), (133000, 256, 1088)]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(A, B, trans_b=True, out=out)
    torch.cuda.synchronize()
    worst = 0.0
    for lo in range(0, M, 65536):
        ref = (A[lo:lo + 65536].float() @ B.float().t())
        err = (out[lo:lo + 65536].float() - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, err)
    print("M=%d N=%d K=%d: max |err| / max |ref| = %.3e %s" % (M, N, K, worst, "OK" if worst < 1e-2 else "FAIL"), flush=True)
