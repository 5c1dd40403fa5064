"""gru_step_bwd8_kernel (round 6: the backward GRU step on 256-row panels, eight waves, the recurrent product and the previous step's input
gradient fed from ONE walk over d4_prev) against gru_step_bwd_kernel (128-row panels, one workgroup per role and tile): every accumulator
sees the same MFMAs on the same operands in the same order, so dh, d4 and dinp are the SAME BITS; the bias partial sums land in other slots
(another grid) and are sums of fp32 atomics, so their column sums agree to rounding.  Reference function: the autograd of nn.GRU on the packed
sequence, /root/reference/generator/encoder.py:101-105."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda:0")


CASES = [
    # rows, rows_prev, hs, n_in, layer-0 style (dy + dropout masks on both sides), accumulate
    (50001, 40000, 256, 512, False, False),      # direction 0: fewer rows in the step processed before; a partial last panel
    (40000, 50001, 256, 512, False, True),       # direction 1: more rows before (panels with role B only), dinp accumulates
    (20011, 20011, 256, 128, True, False),       # layer 0: 32 input columns per workgroup, output-gradient + dropout, embedding mask on dinp
    (9000, 12000, 256, 128, True, True),
    (30000, 30000, 256, 0, False, False),        # no input-gradient role at all
    (16000, 16000, 128, 256, False, False),      # hs = 128: two channel tiles, 8 stages
    (16000, 15000, 128, 128, True, False),       # ... 64 input columns per workgroup (NBT = 4)
    (10000, 10000, 64, 128, False, True),        # hs = 64: one channel tile, 4 stages (the loop's remainder only)
    (10000, 9000, 192, 192, False, False),       # hs = 192: 12 stages = two whole rounds of the six-stage pattern (n_in = 3 * 64: NBT = 4)
    (70001, 66000, 256, 512, False, False),      # several panels per persistent worker; direction 0
    (66000, 70001, 256, 128, True, True),        # ... direction 1 at layer 0
    (300, 290, 256, 512, False, True),           # a launch smaller than one panel per workgroup
]


@pytest.mark.parametrize("rows,rows_prev,hs,n_in,l0,acc", CASES)
def test_gru_backward_step_wide_kernel_bit_identical(rows, rows_prev, hs, n_in, l0, acc):
    from gtos_amd._lib import call
    from gtos_amd.gru import _step_bwd_fused, N_BIAS_PARTIALS
    torch.manual_seed(rows + rows_prev + hs)
    bf = torch.bfloat16
    rnd = lambda *s, sc=0.3: (torch.randn(*s, device=dev()) * sc).to(bf)          # noqa: E731
    d4_prev, wh_t = rnd(rows_prev, 4 * hs), rnd(hs, 3 * hs, sc=0.1)
    wi_t = rnd(n_in, 3 * hs, sc=0.1) if n_in else None
    gates, hprev, dh0 = torch.rand(rows, 4 * hs, device=dev()).to(bf), rnd(rows, hs), rnd(rows, hs)
    dy = rnd(rows, 2 * hs) if l0 else None
    dinp0 = rnd(rows_prev, n_in + 64) if n_in else None                           # a column block of a wider matrix
    res = []
    try:
        for kernel in (0, 1, 2):                          # 128-row tiles, 256-row tiles, persistent (where it applies: hs 256, n_in 128 / 512)
            call("gtos_gru_bwd_config", kernel, 0)
            dh, d4 = dh0.clone(), torch.zeros(rows, 4 * hs, device=dev(), dtype=bf)
            bpart = torch.zeros(N_BIAS_PARTIALS, 4 * hs, device=dev())
            wide_m = dinp0.clone() if n_in else None
            kw = dict(wi_t=wi_t, dinp=wide_m[:, :n_in], n_in=n_in, dinp_acc=acc, p_in=0.25 if l0 else 0.0, seed_in=77, in_drop_base=3 * n_in) if n_in else {}
            _step_bwd_fused(rows, hs, d4_prev, rows_prev, wh_t, gates, hprev, None if dy is None else dy.data_ptr() + hs * 2, 2 * hs, dh, d4,
                            0.2 if l0 else 0.0, 4711, hs, bpart, **kw)
            torch.cuda.synchronize()
            res.append((dh, d4, wide_m, bpart))
    finally:
        call("gtos_gru_bwd_config", 2, -2)
    assert float(res[0][1].float().abs().max()) > 0
    for k in (1, 2):
        name = ("256-row", "persistent")[k - 1]
        assert torch.equal(res[0][0], res[k][0]), "%s dh: %d rows differ" % (name, int((res[0][0] != res[k][0]).any(1).sum()))
        assert torch.equal(res[0][1], res[k][1]), "%s d4: %d rows differ" % (name, int((res[0][1] != res[k][1]).any(1).sum()))
        if n_in:
            assert torch.equal(res[0][2], res[k][2]), "%s dinp: %d rows differ" % (name, int((res[0][2] != res[k][2]).any(1).sum()))
            assert torch.equal(res[k][2][:, n_in:], dinp0[:, n_in:])               # nothing beyond the block is touched
            if not acc:
                assert not torch.equal(res[k][2][:, :n_in], dinp0[:, :n_in])
        torch.testing.assert_close(res[0][3].sum(0), res[k][3].sum(0), rtol=1e-4, atol=1e-3)


def test_wide_kernel_is_what_production_sized_launches_run():
    """The dispatch rule of gtos_gru_step_bwd_fused (include/gtos_hip.h): >= 8192 covered rows with a recurrent product -> the wide kernel.
    Checked through its side effect on the bias partial slots: the 128-row grid of this launch has 8x the workgroups of the 256-row one,
    so it reaches slots the wide grid never touches."""
    from gtos_amd._lib import call
    from gtos_amd.gru import _step_bwd_fused, N_BIAS_PARTIALS
    torch.manual_seed(1)
    bf, hs, rows = torch.bfloat16, 256, 8192
    rnd = lambda *s: (torch.randn(*s, device=dev()) * 0.3).to(bf)                  # noqa: E731
    d4_prev, wh_t, wi_t = rnd(rows, 4 * hs), rnd(hs, 3 * hs), rnd(512, 3 * hs)
    gates, hprev = torch.rand(rows, 4 * hs, device=dev()).to(bf), rnd(rows, hs)
    used = []
    for kernel, min_rows in ((1, 8192), (1, 8193), (0, 0)):
        call("gtos_gru_bwd_config", kernel, min_rows)
        dh, d4, dinp = rnd(rows, hs), torch.empty(rows, 4 * hs, device=dev(), dtype=bf), torch.empty(rows, 512, device=dev(), dtype=bf)
        bpart = torch.zeros(N_BIAS_PARTIALS, 4 * hs, device=dev())
        _step_bwd_fused(rows, hs, d4_prev, rows, wh_t, gates, hprev, None, 2 * hs, dh, d4, 0.0, 0, 0, bpart, wi_t=wi_t, dinp=dinp, n_in=512)
        torch.cuda.synchronize()
        used.append(int((bpart.abs().sum(1) > 0).sum()))
    call("gtos_gru_bwd_config", 2, -2)
    assert used[0] < used[1] == used[2], used
