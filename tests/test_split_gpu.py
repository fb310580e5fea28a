"""GPU parity of the split 16-bit paths ("bf16x3" / "fp16x3", PS_BF16X3 / PS_F16X3): a value is hi + lo, tensors and weights store blocks of 32
logical channels as [hi(32) | lo(32)] (one 128-byte K-line), every product is x_hi w_hi + x_hi w_lo + x_lo w_hi on the 16-bit MFMA kernels with
f32 accumulation.  It exists
to meet the reference's fp32 arithmetic (models/resnet38d.py:156-188) -- north_star: fp32 logits within 1e-4 relative, mask indices
bit-exact -- at 16-bit MFMA speed / 3 instead of the exact-f32 MFMA's 1/16.  Checked here against torch-CPU fp32 ops on the SAME
split-representable operands (so only the dropped lo*lo term, accumulation order and the output's hi+lo rounding differ): 1e-4 relative
everywhere, observed ~1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
TOL = 1e-4


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


BF, HF = torch.bfloat16, torch.float16  # plane types of the two split formats (PS_BF16X3, PS_F16X3)


def split_repr(t, dt=BF):
    """The value a split tensor can hold: hi + lo with hi = dt(t), lo = dt(t - hi)."""
    hi = t.to(dt).float()
    return hi + (t - hi).to(dt).float()


def planes(t, weights=False, dt=BF):
    """f32 [..., C] -> 16-bit [..., 2C] in the split layout: blocks of 32 logical channels as [hi(32) | lo(32)] (activations and weights alike)."""
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    sh, c = t.shape[:-1], t.shape[-1]
    return torch.stack([hi.reshape(*sh, c // 32, 32), lo.reshape(*sh, c // 32, 32)], dim=-2).reshape(*sh, 2 * c).contiguous()


def merge(p, weights=False):
    sh, c2 = p.shape[:-1], p.shape[-1]
    b = p.reshape(*sh, c2 // 64, 2, 32).float()
    return (b[..., 0, :] + b[..., 1, :]).reshape(*sh, c2 // 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", [BF, HF])
@pytest.mark.parametrize("weights", [False, True])
def test_convert_rows_split_planes_and_casts(weights, dt):
    """ps_convert_rows: f32 -> split planes (both layouts) bit for bit as torch's RNE casts give them, back to f32 = hi + lo exactly; strided
    rows (channel slices of wider buffers); the plain 16-bit casts."""
    from pistoseg_amd import ops

    g = torch.Generator().manual_seed(11)
    rows, c = 37, 96
    src = torch.randn(rows, c, generator=g) * torch.logspace(-6, 6 if dt == BF else 3, rows).view(-1, 1)  # (fp16 planes: |v| <= 65504)
    wide = torch.full((rows, 2 * c + 24), 7.0, dtype=dt, device=D)
    ops.convert_rows(src.to(D), wide[:, 8:8 + 2 * c], c, dst_split=True, weights=weights)
    assert torch.equal(wide[:, 8:8 + 2 * c].cpu(), planes(src, weights, dt))
    assert bool((wide[:, :8] == 7.0).all()) and bool((wide[:, 8 + 2 * c:] == 7.0).all())  # nothing outside the slice is touched
    if not weights:
        back = torch.empty(rows, c, device=D)
        ops.convert_rows(wide[:, 8:8 + 2 * c], back, c, src_split=True)
        assert torch.equal(back.cpu(), split_repr(src, dt))
        assert rel_err(back.cpu(), src) < 2.0 ** -16
        if dt == HF:  # 22 significant bits while lo is a normal fp16 number, 2^-25 absolute below
            assert float((back.cpu() - src).abs().div(src.abs().clamp_min(2.0 ** -3)).max()) < 2.0 ** -21
    for dt2 in (torch.bfloat16, torch.float16):
        small = (torch.randn(rows, c, generator=g)).to(D)
        low = torch.empty(rows, c, device=D, dtype=dt2)
        ops.convert_rows(small, low, c)
        assert torch.equal(low.cpu(), small.cpu().to(dt2))
        up = torch.empty(rows, c, device=D)
        ops.convert_rows(low, up, c)
        assert torch.equal(up.cpu(), low.cpu().float())


# (n, h, w, cin, cout, k, stride, dilation), which kernel family the geometry selects for the FORWARD
SPLIT_CASES = [
    ((2, 13, 10, 64, 128, 3, 2, 1), None),      # small / ragged: the 4-wave kernels
    ((2, 13, 10, 128, 128, 3, 1, 1), None),
    ((2, 13, 10, 512, 64, 1, 1, 1), None),
    ((2, 9, 11, 256, 512, 1, 2, 1), None),
    ((24, 28, 28, 256, 512, 3, 1, 2), "halo"),  # the persistent kernels the bench shapes select, reached by geometry
    ((10, 56, 56, 128, 256, 3, 1, 1), "halo"),
    ((17, 32, 32, 64, 512, 3, 1, 4), "halo"),   # 256-pixel tiles (maps of 256 x 256 inputs)
    ((37, 28, 28, 256, 512, 1, 1, 1), "ws2"),
    ((42, 56, 56, 128, 256, 3, 2, 1), "ws2"),   # stride-2 3x3: the data gradient runs as four parity-class launches
    ((17, 28, 28, 1024, 1024, 1, 1, 1), "ws2"),  # a plain GEMM below conv_gemm256_kernel's range (K < 2048, 212 tiles)
    ((32, 28, 28, 2048, 1024, 1, 1, 1), "gemm256"),  # round 5: the 256 x 256 tile GEMM kernel's split instantiation (third MFMA group per quadrant): 392 tiles, K = 2048
]


@pytest.mark.parametrize("dt", [BF, HF])
@pytest.mark.parametrize("case,family", SPLIT_CASES)
def test_split_conv_fwd_dgrad_wgrad_match_fp32_cpu(case, family, dt):
    """Forward with the full epilogue (residual add, raw output, BN + ReLU + dropout output), data gradient with the ReLU-mask epilogue and a
    second addend, weight gradient (three launches of the 16-bit kernels on the hi / lo halves): all within 1e-4 of torch-CPU fp32 on the same split-representable
    operands (resnet38d.py:16-21,38-41,64,86)."""
    import ctypes as C

    from pistoseg_amd import _lib, ops

    n, h, w, cin, cout, k, s_, d = case
    spec = ops.ConvSpec(cin, cout, k, s_, d)
    if family is not None:
        g_ = ops._geom(spec, _lib.PS_BF16X3 if dt == BF else _lib.PS_F16X3, n, h, w, 2 * cin, 2 * cout)
        want = {"halo": (7,), "ws2": (4, 5), "gemm256": (8,)}[family]
        assert int(_lib.load().ps_conv_variant(C.byref(g_), 0)) in want
    SR = lambda t: split_repr(t, dt)
    PL = lambda t, weights=False: planes(t, weights, dt)
    g = torch.Generator().manual_seed(sum(case))
    x = SR(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = SR(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    pad = d if k == 3 else 0
    y = F.conv2d(x, wt, stride=s_, padding=pad, dilation=d)
    res = SR(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    drop = (torch.rand(n, cout, generator=g) > 0.3).float() / 0.7
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) * drop.view(n, cout, 1, 1)
    gy = SR(torch.randn(y.shape, generator=g))
    y.backward(gy)
    mask_src = SR(F.relu(torch.randn(n, cin, h, w, generator=g)))
    add1 = SR(torch.randn(n, cin, h, w, generator=g))
    sc2 = torch.rand(cin, generator=g) + 0.5
    gx_ref = torch.where(mask_src > 0, x.grad * sc2.view(1, -1, 1, 1), torch.zeros(())) + add1

    ho, wo = spec.out_hw(h, w)
    xd = PL(nhwc(x.detach())).to(D)
    wf = PL(wt.detach().permute(0, 2, 3, 1).contiguous(), weights=True).to(D)  # [cout][kh][kw][cin, split]
    wd = PL(wt.detach().permute(1, 2, 3, 0).contiguous(), weights=True).to(D)  # [cin][kh][kw][cout, split]
    nan = float("nan")
    out_raw = torch.full((n, ho, wo, 2 * cout), nan, device=D, dtype=dt)
    # the activated output as a channel slice of a wider buffer (the fused bottleneck's [a | a3] layout)
    wide = torch.full((n, ho, wo, 2 * cout + 2 * 64), nan, device=D, dtype=dt)
    out_act = wide[..., 2 * 64:]
    ops.conv2d_fwd(spec, xd, wf, add0=PL(nhwc(res)).to(D), out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), drop=drop.to(D),
                   out_act=out_act, split=True)
    assert rel_err(merge(out_raw.cpu()), nhwc((y + res).detach())) < TOL
    assert rel_err(merge(out_act.cpu()), nhwc(act.detach())) < TOL
    assert bool(torch.isnan(wide[..., :2 * 64].float()).all())         # nothing outside the slice touched
    gyd = PL(nhwc(gy)).to(D)
    gx = torch.full((n, h, w, 2 * cin), nan, device=D, dtype=dt)
    ops.conv2d_dgrad(spec, gyd, wd, (h, w), mask_src=PL(nhwc(mask_src)).to(D), bn_scale=sc2.to(D), add1=PL(nhwc(add1)).to(D), out=gx, split=True)
    assert rel_err(merge(gx.cpu()), nhwc(gx_ref)) < TOL
    dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
    ops.conv2d_wgrad(spec, xd, gyd, dw, split=True, opts=ops.LaunchOpts(wgrad_terms=3))
    assert rel_err(dw.cpu(), wt.grad.permute(0, 2, 3, 1)) < TOL
    # the default: hi halves only -- exactly the gradient of the hi-rounded operands, i.e. 2^-8 (bf16) / 2^-11 (fp16) per product away from the full one
    dw1 = torch.zeros_like(dw)
    ops.conv2d_wgrad(spec, xd, gyd, dw1, split=True)
    xh, gh = x.detach().to(dt).float(), gy.to(dt).float()
    ref1 = torch.nn.grad.conv2d_weight(xh, wt.shape, gh, stride=s_, padding=pad, dilation=d)
    assert rel_err(dw1.cpu(), ref1.permute(0, 2, 3, 1)) < TOL
    assert rel_err(dw1.cpu(), wt.grad.permute(0, 2, 3, 1)) < (2e-2 if dt == BF else 2e-3)
    print(f"[split conv {str(dt)[6:]} {case}] raw {rel_err(merge(out_raw.cpu()), nhwc((y + res).detach())):.2e} act {rel_err(merge(out_act.cpu()), nhwc(act.detach())):.2e} "
          f"dgrad {rel_err(merge(gx.cpu()), nhwc(gx_ref)):.2e} wgrad {rel_err(dw.cpu(), wt.grad.permute(0, 2, 3, 1)):.2e}")


def test_split_conv_rejects_narrow_channel_strides():
    from pistoseg_amd import _lib, ops

    spec = ops.ConvSpec(64, 64, 1)
    x = torch.zeros((1, 4, 4, 128), device=D, dtype=torch.bfloat16)
    wf = torch.zeros((64, 1, 1, 128), device=D, dtype=torch.bfloat16)
    y = torch.zeros((1, 4, 4, 64), device=D, dtype=torch.bfloat16)  # room for the hi halves only: must be refused, not overrun
    with pytest.raises(_lib.PsError):
        ops.conv2d_fwd(spec, x, wf, out_raw=y, split=True)


@pytest.mark.parametrize("dt", [BF, HF])
@pytest.mark.parametrize("GEOM", [(24, 28, 28, 256, 512, 3, 2),    # halo kernel: 84 pixel tiles x 4 cout tiles = 336 tiles
                                  (37, 28, 28, 256, 512, 1, 1)])   # a GEMM: the split types' conv_igemm_ws2_kernel, 130 x 4 = 520 tiles
def test_split_convs_through_the_tile_queue_are_bit_identical(dt, GEOM):
    """ps_conv_geom.tile_queue = 1 on the split types (halo kernel with three MFMA groups per K-line, weight gradient on the gathered hi halves):
    same bits as the static schedule for forward and data gradient, f32 atomic ordering for the weight gradient."""
    from pistoseg_amd import ops

    n, h, w, cin, cout, k, d = GEOM
    spec = ops.ConvSpec(cin, cout, k, 1, d)
    g = torch.Generator().manual_seed(77)
    PL = lambda t: planes(t, False, dt)
    x = PL(torch.randn(n, h, w, cin, generator=g)).to(D)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.02
    wf, wd = PL(wt.permute(0, 2, 3, 1).contiguous()).to(D), PL(wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()).to(D)
    gy = PL(torch.randn(n, h, w, cout, generator=g)).to(D)

    def run():
        y = torch.empty((n, h, w, 2 * cout), device=D, dtype=dt)
        ops.conv2d_fwd(spec, x, wf, out_raw=y, split=True)
        gx = torch.empty((n, h, w, 2 * cin), device=D, dtype=dt)
        ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx, split=True)
        dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
        ops.conv2d_wgrad(spec, x, gy, dw, split=True)
        return y, gx, dw

    stream_k_before, ops.STREAM_K = ops.STREAM_K, False  # like for like: the queue hands out whole tiles (stream-K re-associates the static schedule's sums)
    try:
        ops.TILE_QUEUE = 0
        ref = run()
        ops.TILE_QUEUE = 1
        got = run()
    finally:
        ops.TILE_QUEUE = 0
        ops.STREAM_K = stream_k_before
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert rel_err(got[2].cpu(), ref[2].cpu()) < 1e-5


@pytest.mark.parametrize("dt", [BF, HF])
@pytest.mark.parametrize("case", [
    (24, 28, 28, 256, 512, 3, 1, 2),   # halo kernel
    (3, 13, 10, 128, 128, 3, 1, 1),    # 4-wave kernel, ragged
    (37, 28, 28, 256, 512, 1, 1, 1),   # ws2 GEMM
    (42, 56, 56, 128, 256, 3, 2, 1),   # stride 2: the data gradient runs as four parity-class launches (the epilogue's row map)
    (3, 28, 28, 512, 1024, 1, 1, 1),   # three images per tile: dropout multipliers per row (the epilogue's row-by-row path)
])
def test_split_epilogue_hi_copy_and_weight_gradient_on_it(case, dt):
    """ps_epilogue.out_hi / ops.attach_hi: the epilogue that writes a split tensor as its `out` also writes the hi halves as a plain 16-bit
    tensor -- bit for bit the hi halves of what it stored, for the forward's activated output (BN + ReLU + dropout, as a channel slice) and for
    the data gradient's ReLU-masked output -- and a split weight gradient whose operands both carry such a copy runs on the copies: equal to
    the gathered-hi-halves launch up to f32 summation order."""
    from pistoseg_amd import ops

    n, h, w, cin, cout, k, s_, d = case
    spec = ops.ConvSpec(cin, cout, k, s_, d)
    ho, wo = spec.out_hw(h, w)
    g = torch.Generator().manual_seed(sum(case) + 1)
    PL = lambda t: planes(t, False, dt)
    hi_of = lambda p: p.reshape(*p.shape[:-1], p.shape[-1] // 64, 2, 32)[..., 0, :].reshape(*p.shape[:-1], p.shape[-1] // 2)
    x = PL(torch.randn(n, h, w, cin, generator=g)).to(D)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    wf, wd = PL(wt.permute(0, 2, 3, 1).contiguous()).to(D), PL(wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()).to(D)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(D), (torch.randn(cout, generator=g) * 0.1).to(D)
    drop = ((torch.rand(n, cout, generator=g) > 0.3).float() / 0.7).to(D)
    wide = torch.zeros((n, ho, wo, 2 * cout + 128), device=D, dtype=dt)
    wide_hi = torch.full((n, ho, wo, cout + 64), 7.0, device=D, dtype=dt)
    act = ops.attach_hi(wide[..., 128:], wide_hi[..., 64:])
    ops.conv2d_fwd(spec, x, wf, bn_scale=scale, bn_shift=shift, drop=drop, out_act=act, split=True)
    assert torch.equal(wide_hi[..., 64:], hi_of(act)) and bool((wide_hi[..., :64] == 7.0).all())
    gy = PL(torch.randn(n, ho, wo, cout, generator=g)).to(D)
    mask = PL(torch.relu(torch.randn(n, h, w, cin, generator=g))).to(D)
    sc2, drop2 = (torch.rand(cin, generator=g) + 0.5).to(D), ((torch.rand(n, cin, generator=g) > 0.3).float() / 0.7).to(D)
    gx = ops.attach_hi(torch.zeros((n, h, w, 2 * cin), device=D, dtype=dt), torch.full((n, h, w, cin), 7.0, device=D, dtype=dt))
    ops.conv2d_dgrad(spec, gy, wd, (h, w), mask_src=mask, bn_scale=sc2, drop=drop2 if s_ == 1 else None, out=gx, split=True)
    assert torch.equal(gx._ps_hi, hi_of(gx))
    # weight gradient: operands with companions -> the plain 16-bit kernel on them
    dw_ref = torch.zeros((cout, k, k, cin), device=D)
    ops.conv2d_wgrad(spec, x, gy, dw_ref, split=True)
    ops.attach_hi(x, hi_of(x).contiguous())
    ops.attach_hi(gy, hi_of(gy).contiguous())
    dw = torch.zeros((cout, k, k, cin), device=D)
    ops.conv2d_wgrad(spec, x, gy, dw, split=True)
    assert rel_err(dw.cpu(), dw_ref.cpu()) < 1e-5
