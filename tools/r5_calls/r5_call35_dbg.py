import torch, os
from gtos_amd.gru import _step_bwd_fused, N_BIAS_PARTIALS
dev = torch.device("cuda:0"); bf = torch.bfloat16
rows, rows_prev, hs, n_in = 40000, 50001, 256, 128
torch.manual_seed(rows + rows_prev)
d4_prev = (torch.randn(rows_prev, 4 * hs, device=dev) * 0.3).to(bf)
w_ih = (torch.randn(3 * hs, n_in, device=dev) * 0.1).to(bf); wi_t = w_ih.t().contiguous()
wh_t = (torch.randn(hs, 3 * hs, device=dev) * 0.1).to(bf)
gates = torch.rand(rows, 4 * hs, device=dev).to(bf); hprev = torch.randn(rows, hs, device=dev).to(bf)
def run(fused):
    dh = torch.randn(rows, hs, generator=torch.Generator(device=dev).manual_seed(3), device=dev).to(bf)
    d4 = torch.zeros(rows, 4 * hs, device=dev, dtype=bf)
    bpart = torch.zeros(N_BIAS_PARTIALS, 4 * hs, device=dev)
    wide = torch.full((rows_prev, n_in + 64), 7.0, device=dev, dtype=bf)
    kw = dict(wi_t=wi_t, dinp=wide[:, :n_in], n_in=n_in) if fused else {}
    _step_bwd_fused(rows, hs, d4_prev, rows_prev, wh_t, gates, hprev, None, hs, dh, d4, 0.0, 0, 0, bpart, **kw)
    torch.cuda.synchronize()
    return dh.float()
ref = run(False)
# the exact value: dh_in + d4_prev[:, r|z|hn] @ wh_t^T ... compare the recurrent part only through differences between runs
for name, f in (("plain again", False), ("fused", True), ("fused again", True)):
    o = run(f)
    bad = (o != ref)
    r = bad.any(1).nonzero().flatten()
    print(name, "differing rows:", r.numel(), "first", r[:8].tolist(), "last", r[-8:].tolist(), "cols of first bad row", bad[r[0]].nonzero().flatten()[:10].tolist() if r.numel() else None,
          "max abs diff", float((o - ref).abs().max()))
    if r.numel():
        import collections
        print("   panels (row // 128) histogram head:", collections.Counter((r // 128).tolist()).most_common(6), "rows mod 128 of bad:", sorted(set((r % 128).tolist()))[:20])
