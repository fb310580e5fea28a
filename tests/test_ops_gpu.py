"""GPU parity of the individual HIP kernels (through the C-ABI) against the torch-CPU primitives the oracle
is made of.  f32 path: 1e-4 relative (north_star); bf16 path: compared against the same CPU op on
bf16-rounded operands (only accumulation order and output rounding differ)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True, params=["product"])
def library(request):
    """Which binary a test of this module launches through:

    * "product" (the default) = libpistoseg_hip.so, what `bench.py` and the model-level tests run (tunables `constexpr`, no `ps_debug_*`
      symbols, only reachable kernels instantiated).  Kernel selection there is a pure function of the geometry, so the cases below include
      shapes that SELECT the persistent kernels (halo / ws2 / gemm256 / wgrad_ws2) by themselves -- asserted through `ps_conv_variant` /
      `ps_conv_wgrad_variant`;
    * "debug" = libpistoseg_hip_debug.so (same sources, -DPS_DEBUG_HOOKS): only for the tests that drive `ps_debug_set_*` -- the staging /
      tiling variant sweeps (`@debug_only`) and the tests that compare a forced variant with the default (`@both_libraries`: they run their
      CPU comparison on the product library as well).  Same sources is not same code (the hand-scheduled loops are sensitive to any
      codegen change), which is why the parity claim rests on the product runs."""
    from pistoseg_amd import _lib

    _lib.use_debug_library(request.param == "debug")
    yield request.param
    _lib.use_debug_library(False)


both_libraries = pytest.mark.parametrize("library", ["product", "debug"], indirect=True)
debug_only = pytest.mark.parametrize("library", ["debug"], indirect=True)  # tests that drive `ps_debug_set_*`
# conv_igemm.hip's variant codes (include/pistoseg_hip.h)
V_4WAVE, V_WS_128, V_WS_112, V_WS2_256, V_WS2_224, V_OTHER, V_HALO, V_GEMM256 = 1, 2, 3, 4, 5, 6, 7, 8


def conv_variant(spec, dtype, n, h, w, kind):
    """Which kernel family the loaded library selects for this launch (`ps_conv_variant` / `ps_conv_wgrad_variant`)."""
    import ctypes as C

    from pistoseg_amd import _lib, ops

    g = ops._geom(spec, ops._dt(dtype), n, h, w, spec.cin, spec.cout)
    if kind == "wgrad":
        return int(_lib.load().ps_conv_wgrad_variant(C.byref(g)))
    return int(_lib.load().ps_conv_variant(C.byref(g), 1 if kind == "dgrad" else 0))


HALO_RING_DEFAULT = 3  # library default of ps_debug_set_halo_ring
WS2_DEFAULT = 1  # library default of ps_debug_set_ws2 (restored after tests that force a variant)
F32_TOL = 1e-4
BF16_TOL = 1.2e-2
F16_TOL = 2e-3  # fp16 storage (11 significant bits), f32 accumulate
TOL = {torch.float32: F32_TOL, torch.bfloat16: BF16_TOL, torch.float16: F16_TOL}


def quant(dtype):
    """Round operands to the storage dtype on the CPU side, so only accumulation order / output rounding differ."""
    return (lambda t: t) if dtype == torch.float32 else (lambda t: t.to(dtype).float())


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def nhwc(t):  # NCHW -> channels-last [N,H,W,C]
    return t.permute(0, 2, 3, 1).contiguous()


def w_fwd_layout(w):  # OIHW -> [cout][kh][kw][cin]
    return w.permute(0, 2, 3, 1).contiguous()


def w_dgrad_layout(w):  # OIHW -> [cin][kh][kw][cout]
    return w.permute(1, 2, 3, 0).contiguous()


# the distinct (cin, cout, k, s, d) classes of SURVEY 8a plus head shapes
CONV_CASES = [
    (64, 128, 3, 2, 1), (64, 128, 1, 2, 1), (128, 128, 3, 1, 1), (128, 256, 3, 2, 1), (256, 256, 3, 1, 1),
    (256, 512, 1, 2, 1), (512, 512, 3, 1, 1), (512, 1024, 1, 1, 1), (512, 1024, 3, 1, 2), (1024, 512, 3, 1, 2),
    (512, 1024, 3, 1, 4), (1024, 2048, 1, 1, 1), (512, 64, 1, 1, 1), (256, 192, 1, 1, 1),
]


# operand staging: buffer LDS-DMA (the default: both libraries), flat LDS-DMA and registers (debug switches)
@pytest.mark.parametrize("library,glds", [("product", 2), ("debug", 2), ("debug", 1), ("debug", 0)], indirect=["library"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype, glds, library):
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    if library == "debug":
        lib.ps_debug_set_glds(glds)
    try:
        cin, cout, k, s, d = case
        if glds != 2 and cin > 256:
            pytest.skip("the alternative staging variants are covered on the small cases")
        n, h, w = 2, 13, 10  # odd/even sizes, M not a multiple of 128
        g = torch.Generator().manual_seed(sum(case) * 7 + k)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        x, wt = quant(dtype)(x), quant(dtype)(wt)
        x.requires_grad_(True)
        wt.requires_grad_(True)
        y = F.conv2d(x, wt, stride=s, padding=d if k == 3 else 0, dilation=d)
        gy = torch.randn(y.shape, generator=g)
        gy = quant(dtype)(gy)
        y.backward(gy)
        tol = TOL[dtype]

        spec = ops.ConvSpec(cin, cout, k, s, d)
        xd = nhwc(x.detach()).to(dev(), dtype)
        wf = w_fwd_layout(wt.detach()).to(dev(), dtype)
        ho, wo = spec.out_hw(h, w)
        yd = torch.full((n, ho, wo, cout), float("nan"), device=dev(), dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, out_raw=yd)
        assert rel_err(yd.float().cpu(), nhwc(y.detach())) < tol

        # dgrad
        gyd = nhwc(gy).to(dev(), dtype)
        wd = w_dgrad_layout(wt.detach()).to(dev(), dtype)
        gxd = torch.full((n, h, w, cin), float("nan"), device=dev(), dtype=dtype)
        ops.conv2d_dgrad(spec, gyd, wd, (h, w), out_raw=gxd)
        assert rel_err(gxd.float().cpu(), nhwc(x.grad)) < tol

        # weight transpose kernel produces the dgrad layout
        wd2 = torch.empty_like(wd)
        ops.weight_transpose(wf, wd2, cout, k * k, cin)
        assert torch.equal(wd2, wd)

        # wgrad (f32 accumulate, atomics)
        dw = torch.zeros((cout, k, k, cin), device=dev(), dtype=torch.float32)
        ops.conv2d_wgrad(spec, xd, gyd, dw)
        assert rel_err(dw.cpu(), w_fwd_layout(wt.grad)) < tol
    finally:
        if library == "debug":
            lib.ps_debug_set_glds(2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv_epilogues(dtype):
    from pistoseg_amd import ops

    tol = TOL[dtype]
    n, h, w, cin, cout = 3, 9, 11, 128, 256
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g)
    q = quant(dtype)
    x, wt, res = q(rnd(n, cin, h, w)), q(rnd(cout, cin, 3, 3) * 0.03), q(rnd(n, cout, h, w))
    scale, shift = torch.rand(cout, generator=g) + 0.5, rnd(cout) * 0.2
    drop = (torch.rand(n, cout, generator=g) > 0.3).float() / 0.7
    spec = ops.ConvSpec(cin, cout, 3, 1, 2)
    y = F.conv2d(x, wt, padding=2, dilation=2) + res
    act = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) * drop.view(n, cout, 1, 1)
    D = dev()
    xd, wf, resd = nhwc(x).to(D, dtype), w_fwd_layout(wt).to(D, dtype), nhwc(res).to(D, dtype)
    # outputs are channel slices of wider buffers (ldc > C)
    wide_raw = torch.zeros((n, h, w, cout + 64), device=D, dtype=dtype)
    wide_act = torch.zeros((n, h, w, cout + 128), device=D, dtype=dtype)
    out_raw, out_act = wide_raw[..., 64:], wide_act[..., :cout]
    ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), drop=drop.to(D), out_act=out_act)
    assert rel_err(out_raw.float().cpu(), nhwc(y)) < tol
    assert rel_err(out_act.float().cpu(), nhwc(act)) < tol
    assert float(wide_raw[..., :64].abs().max()) == 0 and float(wide_act[..., cout:].abs().max()) == 0

    # backward epilogue: dx = ((conv^T(dy) + add0) * [mask>0] * scale * drop) + add1
    dy = q(rnd(n, cout, h, w))
    mask_src = q(F.relu(rnd(n, cin, h, w)))
    add0, add1 = q(rnd(n, cin, h, w)), q(rnd(n, cin, h, w))
    sc2 = torch.rand(cin, generator=g) + 0.5
    drop2 = (torch.rand(n, cin, generator=g) > 0.3).float() / 0.7
    dx = F.conv_transpose2d(dy, wt, padding=2, dilation=2) + add0
    ref = torch.where(mask_src > 0, dx * sc2.view(1, -1, 1, 1) * drop2.view(n, cin, 1, 1), torch.zeros(())) + add1
    wd = w_dgrad_layout(wt).to(D, dtype)
    out = torch.empty((n, h, w, cin), device=D, dtype=dtype)
    ops.conv2d_dgrad(spec, nhwc(dy).to(D, dtype), wd, (h, w), add0=nhwc(add0).to(D, dtype), mask_src=nhwc(mask_src).to(D, dtype),
                     bn_scale=sc2.to(D), drop=drop2.to(D), add1=nhwc(add1).to(D, dtype), out=out)
    assert rel_err(out.float().cpu(), nhwc(ref)) < tol


def test_conv_full_size_tile_properties():
    """BASELINE-size layer (b7 3x3 d4 at 28x28, bs=4) checked through linearity and a sampled oracle."""
    from pistoseg_amd import ops

    D = dev()
    n, h, w, cin, cout = 4, 28, 28, 1024, 2048
    g = torch.Generator().manual_seed(11)
    x1, x2 = torch.randn(n, h, w, cin, generator=g), torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.01
    spec = ops.ConvSpec(cin, cout, 3, 1, 4)
    wf = w_fwd_layout(wt).to(D)

    def run(x):
        y = torch.empty((n, h, w, cout), device=D)
        ops.conv2d_fwd(spec, x.to(D), wf, out_raw=y)
        return y

    y1, y2, y12 = run(x1), run(x2), run(x1 + 2 * x2)
    assert rel_err(y12.cpu(), (y1 + 2 * y2).cpu()) < 1e-5  # linearity
    ref = F.conv2d(x1[:1].permute(0, 3, 1, 2), wt, padding=4, dilation=4)  # one image on the CPU
    assert rel_err(y1[:1].cpu(), nhwc(ref)) < F32_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv1a_and_fc8(dtype):
    from pistoseg_amd import ops

    D = dev()
    tol = TOL[dtype]
    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 17, 20
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    y = F.conv2d(x, wt, padding=1)
    act = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    out_act = torch.empty((n, h, w, 64), device=D, dtype=dtype)
    out_raw = torch.empty((n, h, w, 64), device=D, dtype=dtype)
    ops.conv1a_fwd(x.to(D), wt.to(D), scale.to(D), shift.to(D), out_act, out_raw)
    if dtype != torch.float32:  # the 16-bit paths round image and weights to the storage type: compare on rounded operands
        qq = quant(dtype)
        y = F.conv2d(qq(x), qq(wt), padding=1)
        act = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    assert rel_err(out_raw.float().cpu(), nhwc(y)) < tol
    assert rel_err(out_act.float().cpu(), nhwc(act)) < tol

    # fc8 forward / backward
    k, c, gsz = 4096, 4, 5
    q = quant(dtype)
    feat = q(F.relu(torch.randn(n, k, gsz, gsz, generator=g)))
    w8 = torch.randn(c, k, generator=g) * 0.02
    drop = (torch.rand(n, k, generator=g) > 0.5).float() * 2
    s7 = torch.rand(k, generator=g) + 0.5
    feat.requires_grad_(True)
    w8.requires_grad_(True)
    cam = F.conv2d(feat * drop.view(n, k, 1, 1), w8.view(c, k, 1, 1))
    dcam = torch.randn(cam.shape, generator=g)
    cam.backward(dcam)
    fd = nhwc(feat.detach()).to(D, dtype)
    camd = torch.empty((n, gsz, gsz, c), device=D)
    ops.fc8_fwd(fd, w8.detach().to(D), drop.to(D), camd)
    assert rel_err(camd.cpu(), nhwc(cam.detach())) < tol
    dx = torch.empty_like(fd)
    dw = torch.zeros((c, k), device=D)
    ops.fc8_bwd(fd, w8.detach().to(D), drop.to(D), s7.to(D), nhwc(dcam).to(D), dx, dw)
    ref_dx = feat.grad * (feat.detach() > 0) * s7.view(1, -1, 1, 1)
    assert rel_err(dx.float().cpu(), nhwc(ref_dx)) < tol
    assert rel_err(dw.cpu(), w8.grad) < tol


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("sizes", [((28, 28), (224, 224)), ((32, 32), (28, 28)), ((7, 9), (40, 33)), ((224, 224), (32, 32)), ((256, 256), (32, 32)), ((5, 5), (5, 5)),
                                   ((3, 5), (4, 1100))])  # (destination rows above 1024 pixels: the wave-per-source-pixel backward)
def test_bilinear_fwd_bwd(sizes, align):
    from pistoseg_amd import ops

    D = dev()
    (hi, wi), (ho, wo) = sizes
    g = torch.Generator().manual_seed(hi * 131 + ho)
    x = torch.randn(2, 5, hi, wi, generator=g, requires_grad=True)
    y = F.interpolate(x, (ho, wo), mode="bilinear", align_corners=align)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    # channels-last f32 source -> NCHW destination (the cam upsample) ...
    src = nhwc(x.detach()).to(D)
    dst = torch.empty((2, 5, ho, wo), device=D)
    ops.bilinear_fwd(src, "nhwc", dst, "nchw", align)
    assert rel_err(dst.cpu(), y.detach()) < 1e-5
    # ... and NCHW -> channels-last bf16 slice (the x_s feature of the affinity head)
    wide = torch.zeros((2, ho, wo, 16), device=D, dtype=torch.bfloat16)
    ops.bilinear_fwd(x.detach().to(D), "nchw", wide[..., 8:13], "nhwc", align)
    assert rel_err(wide[..., 8:13].float().cpu(), nhwc(y.detach())) < 1e-2
    dsrc = torch.full((2, hi, wi, 5), float("nan"), device=D)
    ops.bilinear_bwd(gy.to(D), "nchw", dsrc, "nhwc", align)
    assert rel_err(dsrc.cpu(), nhwc(x.grad)) < 1e-5


@pytest.mark.parametrize("ignore", [3, None])
def test_softmax_ce(ignore):
    from pistoseg_amd import ops

    D = dev()
    g = torch.Generator().manual_seed(9)
    c = 3 if ignore == 3 else 4
    logits = (torch.randn(3, c, 37, 41, generator=g) * 3).requires_grad_(True)
    tgt = torch.randint(0, 4, (3, 37, 41), generator=g)
    ce = F.cross_entropy(logits, tgt, reduction="none", **({"ignore_index": ignore} if ignore is not None else {}))
    loss = ce.mean()
    loss.backward()
    l, dl = ops.softmax_ce(logits.detach().to(D), tgt.to(D), ignore, want_grad=True)
    assert abs(float(l) - float(loss)) <= 1e-5 * abs(float(loss))
    assert rel_err(dl.cpu(), logits.grad) < 1e-5
    if ignore is not None:
        assert float(dl.cpu().permute(0, 2, 3, 1)[tgt == ignore].abs().max()) == 0.0


def test_ce_and_argmax_run_time_channel_count():
    """C = 2..5 are compile-time instantiations of the CE / argmax kernels (a pixel's values loaded once); any other count takes the
    run-time loops: same results."""
    from pistoseg_amd import _lib, ops

    D = dev()
    g = torch.Generator().manual_seed(77)
    for c in (7, 1, 5):
        logits = (torch.randn(2, c, 19, 23, generator=g) * 3).requires_grad_(True)
        tgt = torch.randint(0, c + 1, (2, 19, 23), generator=g)
        loss = F.cross_entropy(logits, tgt, ignore_index=c)  # mean over kept pixels ...
        kept = (tgt != c).sum()
        (loss * kept / tgt.numel()).backward()               # ... the kernel's mean is over ALL pixels (segmentation_module.py:97-98)
        l, dl = ops.softmax_ce(logits.detach().to(D), tgt.to(D), c, want_grad=True)
        assert abs(float(l) - float(loss * kept / tgt.numel())) <= 1e-5 * abs(float(loss)) + 1e-7
        assert rel_err(dl.cpu(), logits.grad) < 1e-5 or float(logits.grad.abs().max()) == 0.0
        x = logits.detach()
        assert torch.equal(torch.argmax(x, dim=1).byte(), ops.argmax_mask(x.to(D), mode=_lib.PS_MASK_PLAIN).cpu())
        assert torch.equal(torch.argmax(torch.softmax(x, 1), dim=1).byte(), ops.argmax_mask(x.to(D), mode=_lib.PS_MASK_PLAIN, softmax_first=True).cpu())


def test_argmax_modes_match_oracle(golden_dir):
    """Mask indices are bit-exact against the oracle / reference goldens on identical inputs."""
    from oracle import ref_cpu
    from pistoseg_amd import _lib, ops

    D = dev()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 5, 64, 48, generator=g)
    x[0, 1] = x[0, 2]  # ties: first maximum must win
    # loss.py:57-60
    for probs in (False, True):
        ref = ref_cpu.logits_to_mask(x, probs=probs)
        got = ops.argmax_mask(x.to(D), mode=_lib.PS_MASK_PLAIN, softmax_first=not probs).cpu()
        assert torch.equal(ref, got)
    xn = x.clone()
    xn[1, 3, 5, 5] = float("nan")  # torch.argmax treats NaN as the maximum
    assert torch.equal(torch.argmax(xn, dim=1).byte(), ops.argmax_mask(xn.to(D), mode=_lib.PS_MASK_PLAIN).cpu())
    # infer_revise_masks.py:137-143
    label = torch.tensor([[1, 1, 0, 1, 1], [1, 0, 1, 1, 0], [1, 1, 1, 1, 1]], dtype=torch.float32)
    ref = torch.argmax((x * label.view(3, 5, 1, 1))[:, 1:], dim=1)
    got = ops.argmax_mask(x.to(D), mode=_lib.PS_MASK_MUL, first_ch=1, label=label.to(D)).cpu()
    assert torch.equal(ref.to(torch.uint8), got)
    # infer_pseudo_masks.py:69-87 against the reference's own goldens
    gm = np.load(os.path.join(golden_dir, "mask_reduce.npz"))
    for i in range(5):
        logit = torch.from_numpy(gm[f"c{i}.logit"])[None]
        lab = torch.from_numpy(gm[f"c{i}.label"]).float()[None]
        tissue = torch.from_numpy((gm[f"c{i}.tissue"] != 0).astype(np.uint8))[None]
        m, e = ops.argmax_mask(logit.to(D), mode=_lib.PS_MASK_FILL, label=lab.to(D), tissue=tissue.to(D), want_entropy=True)
        assert np.array_equal(m.cpu().numpy()[0].astype(np.int64), gm[f"c{i}.mask"])
        np.testing.assert_allclose(e.cpu().numpy()[0], gm[f"c{i}.entropy"], rtol=1e-4, atol=1e-6)


def test_confusion_matches_golden(golden_dir):
    from pistoseg_amd import _lib, ops

    D = dev()
    g = np.load(os.path.join(golden_dir, "miou.npz"))
    logits = torch.from_numpy(g["logits"]).to(D)
    pred = ops.argmax_mask(logits, mode=_lib.PS_MASK_PLAIN, softmax_first=True)
    cm = torch.zeros(9, dtype=torch.int64, device=D)
    ops.confusion_accum(pred, torch.from_numpy(g["gt"]).to(D), cm, 3)
    ops.confusion_accum(pred, torch.from_numpy(g["gt"]).to(D), cm, 3)
    assert np.array_equal(cm.cpu().numpy().reshape(3, 3), 2 * g["cm"].astype(np.int64))
    # the IoU formulas on the device (ps_iou_from_confusion) against the values the REFERENCE's mIoUMask returned for this matrix (loss.py:28-53):
    # bit for bit -- a matrix counted twice has the same ratios
    vals = ops.iou_from_confusion(cm, 3).cpu().numpy()
    assert vals[0] == float(g["miou"]) and vals[1] == float(g["fwiou"]) and np.array_equal(vals[2:], g["tissue_iou"])
    # ... and through the mirror's forward: lazily fetched values, the reference's return contract (Mean_IoU, FW_IoU)
    from pistoseg_amd.metrics import mIoUMask

    meter = mIoUMask(num_classes=3)
    miou, fwiou = meter(logits, torch.from_numpy(g["gt"]).to(D))
    assert float(miou) == float(g["miou"]) and float(fwiou) == float(g["fwiou"]) and f"{miou:.6f}" == f"{float(g['miou']):.6f}"


def test_optimizers_match_torch():
    from pistoseg_amd import ops

    D = dev()
    g = torch.Generator().manual_seed(2)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    # AdamW (segmentation_module.py:86-90)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=0.05)
    p, m, v = p0.clone().to(D), torch.zeros(n, device=D), torch.zeros(n, device=D)
    pb = torch.empty(n, device=D, dtype=torch.bfloat16)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pt.grad = gr.clone()
        opt.step()
        ops.adamw_step(p, gr.to(D), m, v, pb, 1e-3, (0.9, 0.999), 1e-8, 0.05, step)
    assert rel_err(p.cpu(), pt.detach()) < 1e-6
    assert torch.equal(pb.cpu(), p.cpu().bfloat16())
    # SGD with momentum + L2 (utils.PolyOptimizer's effective update)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([pt], lr=0.01, momentum=5e-4, weight_decay=5e-4)
    p, buf = p0.clone().to(D), torch.zeros(n, device=D)
    for step in range(3):
        gr = torch.randn(n, generator=g)
        pt.grad = gr.clone()
        opt.step()
        ops.sgd_step(p, gr.to(D), buf, None, 0.01, 5e-4, 5e-4, step == 0)
    assert rel_err(p.cpu(), pt.detach()) < 1e-6


def test_scaled_optimizers_fp16_shadow_and_nonfinite():
    """fp16 loss-scaling plumbing: scaled gradients + grad_inv_scale == the unscaled update; fp16 shadow weights;
    inf/nan detection over the gradient arena."""
    from pistoseg_amd import ops

    D = dev()
    g = torch.Generator().manual_seed(4)
    n = 70_001
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    scale = 4096.0
    pa, ma, va = p0.clone().to(D), torch.zeros(n, device=D), torch.zeros(n, device=D)
    pbm, mb, vb = p0.clone().to(D), torch.zeros(n, device=D), torch.zeros(n, device=D)
    sh = torch.empty(n, device=D, dtype=torch.float16)
    ops.adamw_step(pa, gr.to(D), ma, va, None, 1e-3, (0.9, 0.999), 1e-8, 0.05, 1)
    ops.adamw_step(pbm, (gr * scale).to(D), mb, vb, sh, 1e-3, (0.9, 0.999), 1e-8, 0.05, 1, grad_inv_scale=1.0 / scale)
    assert torch.equal(pa, pbm)  # power-of-two scale: exact
    assert torch.equal(sh.cpu(), pbm.cpu().half())
    pa, ba = p0.clone().to(D), torch.zeros(n, device=D)
    pbm, bb = p0.clone().to(D), torch.zeros(n, device=D)
    ops.sgd_step(pa, gr.to(D), ba, None, 0.01, 0.9, 5e-4, True)
    ops.sgd_step(pbm, (gr * scale).to(D), bb, sh, 0.01, 0.9, 5e-4, True, grad_inv_scale=1.0 / scale)
    assert torch.equal(pa, pbm) and torch.equal(sh.cpu(), pbm.cpu().half())
    # cast kernel
    lo = torch.empty(n, device=D, dtype=torch.float16)
    ops.cast_f32_lowp(p0.to(D), lo)
    assert torch.equal(lo.cpu(), p0.half())
    # non-finite count
    t = torch.randn(n, generator=g)
    assert int(ops.nonfinite_count(t.to(D))) == 0
    t[5], t[n - 1], t[4097] = float("inf"), float("nan"), float("-inf")
    assert int(ops.nonfinite_count(t.to(D))) == 3
    # the device-guarded AdamW (ps_adamw_step_guarded): equal to torch's AdamW over the APPLIED steps -- bias corrections from its own step
    # count --, and a step whose gradient overflowed changes nothing (GradScaler.step's `if not found_inf`)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=0.05)
    pg, mg, vg = p0.clone().to(D), torch.zeros(n, device=D), torch.zeros(n, device=D)
    shg = torch.empty(n, device=D, dtype=torch.float16)
    state = torch.zeros(2, device=D, dtype=torch.int32)
    for step in range(5):
        gr2 = torch.randn(n, generator=g)
        overflow = step in (1, 3)
        gdev = (gr2 * scale).to(D)
        if overflow:
            gdev[17] = float("inf")
        else:
            pt.grad = gr2.clone()
            opt.step()
        before = (pg.clone(), mg.clone(), vg.clone(), shg.clone())
        state[1:].zero_()
        ops.nonfinite_count(gdev, out=state[1:])
        ops.adamw_step_guarded(pg, gdev, mg, vg, shg, 1e-3, (0.9, 0.999), 1e-8, 0.05, state, grad_inv_scale=1.0 / scale)
        if overflow:
            assert all(torch.equal(a, b) for a, b in zip(before, (pg, mg, vg, shg))) and state.tolist()[1] == 1
    assert state.tolist()[0] == 3 and rel_err(pg.cpu(), pt.detach()) < 1e-6 and torch.equal(shg.cpu(), pg.cpu().half())


@pytest.mark.selfcheck
@debug_only
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, s, d): large enough for the 256x128 three-stage kernel (>= 256 tiles)
    (4, 28, 28, 2048, 4096, 1, 1, 1), (16, 28, 28, 512, 1024, 3, 1, 4), (12, 56, 56, 256, 512, 3, 2, 1), (3, 112, 112, 128, 128, 3, 1, 1),
])
def test_conv_pipelined_kernels_are_bit_identical_to_two_stage(case, dtype):
    """The 8-wave ping-pong kernel and the 3-stage LDS-DMA ring change the pipeline and the tiling, not the arithmetic: every output element is the same MFMA
    chain over the same K order, so fwd and dgrad must agree BITWISE with the 2-stage kernel (race screen: repeated),
    and one image is checked against the CPU."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, k, s, d = case
    g = torch.Generator().manual_seed(n * 1000 + cin)
    q = quant(dtype)
    x = q(torch.randn(n, h, w, cin, generator=g))
    wt = q(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    spec = ops.ConvSpec(cin, cout, k, s, d)
    ho, wo = spec.out_hw(h, w)
    D = dev()
    xd, wf, wd = x.to(D, dtype), w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
    res = q(torch.randn(n, ho, wo, cout, generator=g)).to(D, dtype)
    scale = (torch.rand(cout, generator=g) + 0.5).to(D)
    shift = (torch.randn(cout, generator=g) * 0.1).to(D)
    gy = q(torch.randn(n, ho, wo, cout, generator=g)).to(D, dtype)
    mask = q(torch.randn(n, h, w, cin, generator=g)).to(D, dtype)

    def run():
        y_raw = torch.empty((n, ho, wo, cout), device=D, dtype=dtype)
        y_act = torch.empty_like(y_raw)
        ops.conv2d_fwd(spec, xd, wf, add0=res, out_raw=y_raw, bn_scale=scale, bn_shift=shift, out_act=y_act)
        gx = torch.empty((n, h, w, cin), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, gy, wd, (h, w), mask_src=mask, out=gx)
        return y_raw, y_act, gx

    try:
        lib.ps_debug_set_3stage(0)
        lib.ps_debug_set_pp(0)
        lib.ps_debug_set_ws(0)
        lib.ps_debug_set_ws2(0)
        lib.ps_debug_set_halo(0)  # the halo kernel re-associates the K sum: covered by test_conv_halo_window_kernel
        ref = run()
        # (3stage, ping-pong, forced group height, wave-specialised, large-tile wave-specialised)
        for setup in ((1, 0, 0, 0, 0), (0, 2, 0, 0, 0), (0, 2, 112, 0, 0), (0, 2, 128, 0, 0), (0, 0, 0, 2, 0), (0, 0, 112, 2, 0), (0, 0, 128, 2, 0),
                      (0, 0, 0, 0, 256), (0, 0, 0, 0, 224)):
            lib.ps_debug_set_3stage(setup[0])
            lib.ps_debug_set_pp(setup[1])
            lib.ps_debug_set_bm(setup[2])
            lib.ps_debug_set_ws(setup[3])
            lib.ps_debug_set_ws2(setup[4])
            for _ in range(3):
                got = run()
                for a_, b_ in zip(got, ref):
                    assert torch.equal(a_, b_), setup
    finally:
        lib.ps_debug_set_3stage(0)
        lib.ps_debug_set_pp(0)
        lib.ps_debug_set_bm(0)
        lib.ps_debug_set_ws(1)
        lib.ps_debug_set_ws2(WS2_DEFAULT)
        lib.ps_debug_set_halo(1)
    cpu = F.conv2d(x[:1].permute(0, 3, 1, 2), wt, stride=s, padding=d if k == 3 else 0, dilation=d) + res[:1].float().cpu().permute(0, 3, 1, 2)
    tol = TOL[dtype]
    assert rel_err(ref[0][:1].float().cpu(), nhwc(cpu)) < tol


@debug_only
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, d): 3x3 stride 1, width a multiple of 28.  Tiles are 8 rows x 28 columns: on 28x28 maps they straddle
    # images (784 = 3.5 tiles) and the last tile is ragged for odd n; 56 / 84 / 112-wide maps have 2-4 column blocks whose halos
    # are real neighbour columns; h = 12 is not a multiple of 8; every dilation of the net; 1 .. 4 K-lines; 1 .. 3 cout tiles
    (3, 28, 28, 128, 128, 1), (2, 28, 28, 64, 256, 2), (3, 28, 28, 256, 128, 4), (1, 28, 28, 128, 256, 4), (5, 28, 28, 64, 128, 1),
    (7, 28, 28, 192, 384, 2), (2, 56, 56, 64, 128, 1), (1, 112, 112, 64, 128, 1), (1, 56, 84, 128, 128, 2), (1, 12, 56, 64, 256, 4),
    # width a multiple of 32 (the maps of 256 x 256 inputs, stages 2 and 4): tiles of 8 rows x 32 columns = 256 pixels, windows of up
    # to 40 columns (10 DMAs per loader wave); 32 x 32 maps = 4 whole tiles per image, 64 / 96 / 128-wide maps have 2-4 column blocks;
    # h = 20 is not a multiple of 8 (tiles straddle images); every dilation; odd n
    (3, 32, 32, 128, 128, 1), (2, 32, 32, 64, 256, 2), (3, 32, 32, 256, 128, 4), (5, 32, 32, 192, 384, 4), (2, 64, 64, 64, 128, 1),
    (1, 128, 128, 64, 128, 1), (1, 40, 96, 128, 128, 2), (3, 20, 64, 64, 256, 4), (1, 32, 256, 64, 128, 1),
])
def test_conv_halo_window_kernel(case, dtype):
    _halo_case(case, dtype, 3)


@debug_only
@pytest.mark.parametrize("ring", [4, 5])
@pytest.mark.parametrize("case", [(3, 28, 28, 256, 128, 4), (7, 28, 28, 192, 384, 2), (2, 56, 56, 64, 128, 1), (1, 12, 56, 64, 256, 4),
                                  (5, 32, 32, 192, 384, 4), (3, 20, 64, 64, 256, 4)])
def test_conv_halo_weight_ring_depths(case, ring):
    """The halo kernel with its weights 3 / 4 K-steps ahead (ring of 4 / 5 stages; 256-pixel tiles cap at 4): same results."""
    _halo_case(case, torch.bfloat16, ring)


@debug_only
def test_conv_halo_staggered_start_is_bit_identical():
    """The diagnostic staggered start of the halo kernel's persistent blocks (ps_debug_set_halo_stagger: blocks of an XCD begin up to 3 x 2048 cycles apart,
    NOTES R5.7) only delays blocks: same bytes out, forward with the full epilogue and data gradient."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, d = 7, 28, 28, 192, 384, 2
    g = torch.Generator().manual_seed(11)
    D, dtype = dev(), torch.bfloat16
    xd = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
    wf = (torch.randn(cout, 3, 3, cin, generator=g) * 0.03).to(D, dtype)
    wd = (torch.randn(cin, 3, 3, cout, generator=g) * 0.03).to(D, dtype)
    resd, gyd = torch.randn(n, h, w, cout, generator=g).to(D, dtype), torch.randn(n, h, w, cout, generator=g).to(D, dtype)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(D), (torch.randn(cout, generator=g) * 0.1).to(D)
    spec = ops.ConvSpec(cin, cout, 3, 1, d)

    def run():
        out_raw, out_act = torch.empty(n, h, w, cout, device=D, dtype=dtype), torch.empty(n, h, w, cout, device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale, bn_shift=shift, out_act=out_act)
        gx = torch.empty(n, h, w, cin, device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, gyd, wd, (h, w), out_raw=gx)
        return out_raw, out_act, gx

    try:
        lib.ps_debug_set_halo(2)
        plain = run()
        lib.ps_debug_set_halo_stagger((4 << 8) | 3)  # 4 phases per XCD, 3 x 2048 cycles apart
        late = run()
    finally:
        lib.ps_debug_set_halo_stagger(0)
        lib.ps_debug_set_halo(1)
    assert all(torch.isfinite(t.float()).all() for t in plain)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(plain, late))


def _halo_case(case, dtype, ring):
    """conv_igemm_halo_kernel (pixel window + halo staged once per tap row, K order (K-line, ty, tx)) forced on small problems:
    forward with the full epilogue and the data gradient against the CPU, plus agreement with the gathered-tile kernels."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, d = case
    g = torch.Generator().manual_seed(h + cin + cout + d)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = q(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
    y = F.conv2d(x, wt, padding=d, dilation=d)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    tol = TOL[dtype]
    spec = ops.ConvSpec(cin, cout, 3, 1, d)
    D = dev()
    xd, wf, wd = nhwc(x.detach()).to(D, dtype), w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
    resd, gyd = nhwc(res).to(D, dtype), nhwc(gy).to(D, dtype)

    def run():
        out_raw = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        out_act = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), out_act=out_act)
        gx = torch.full((n, h, w, cin), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, gyd, wd, (h, w), out_raw=gx)
        return out_raw, out_act, gx

    try:
        lib.ps_debug_set_halo(2)
        lib.ps_debug_set_halo_ring(ring)
        got = [run() for _ in range(2)]
        lib.ps_debug_set_halo(0)
        other = run()
    finally:
        lib.ps_debug_set_halo(1)
        lib.ps_debug_set_halo_ring(HALO_RING_DEFAULT)
    refs = (nhwc((y + res).detach()), nhwc(act.detach()), nhwc(x.grad))
    for trial in got:
        for a_, r_, o_ in zip(trial, refs, other):
            assert rel_err(a_.float().cpu(), r_) < tol
            assert rel_err(a_.float().cpu(), o_.float().cpu()) < max(tol, 1e-5)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(got[0], got[1]))  # deterministic (race screen)


@both_libraries
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, s, d): ragged pixel counts (zero-filled K tail), 1 .. several work items per persistent block
    (2, 13, 10, 128, 256, 3, 2, 1), (2, 13, 10, 256, 256, 3, 1, 1), (3, 9, 11, 256, 512, 1, 2, 1), (2, 13, 10, 512, 1024, 3, 1, 2),
    (4, 14, 14, 1024, 2048, 3, 1, 4), (16, 28, 28, 512, 512, 3, 1, 1), (8, 28, 28, 2048, 4096, 1, 1, 1),
    (1, 7, 9, 256, 256, 3, 1, 2), (3, 28, 28, 256, 512, 1, 1, 1), (5, 12, 28, 512, 256, 3, 1, 4),  # 1 / 37 / 27 K-steps (odd counts, one step)
])
def test_wgrad_large_tile_persistent_kernel(case, dtype, library):
    """conv_wgrad_ws2_kernel (256x128 tile, persistent, 3-stage ring) against the CPU autograd weight gradient on identically rounded
    operands.  Product library: only the cases whose geometry SELECTS the kernel (asserted); debug library: forced on every case, and
    also compared with the 128x128 kernel (same products, f32 atomics in a different order)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, k, s, d = case
    spec = ops.ConvSpec(cin, cout, k, s, d)
    if library == "product" and conv_variant(spec, dtype, n, h, w, "wgrad") not in (1, 2):
        pytest.skip("the product library serves this small problem with the 128x128 kernel (covered by test_conv_fwd_dgrad_wgrad)")
    if library == "debug" and cin * cout * k * k >= 8 * 1024 * 1024 and conv_variant(spec, dtype, n, h, w, "wgrad") in (1, 2):
        pytest.skip("a large case that the product library already runs on this kernel by geometry (the CPU reference is the slow part)")
    g = torch.Generator().manual_seed(cin + cout + k)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g))
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, stride=s, padding=d if k == 3 else 0, dilation=d)
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    D = dev()
    xd, gyd = nhwc(x).to(D, dtype), nhwc(gy).to(D, dtype)
    old = None
    got = []

    def launches(k_):  # race screen: repeated launches agree up to f32 atomic ordering
        for _ in range(k_):
            dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
            ops.conv2d_wgrad(spec, xd, gyd, dw)
            got.append(dw)

    try:
        if library == "debug":
            lib.ps_debug_set_wgrad256(0)
            lib.ps_debug_set_wgrad_ws2(2)   # conv_wgrad_ws2_kernel forced
            assert conv_variant(spec, dtype, n, h, w, "wgrad") == 1
            launches(3)
            if s == 1 and cin % 256 == 0:
                lib.ps_debug_set_wgrad256(2)  # conv_wgrad256_kernel forced (stride 1, 256-channel tiles both ways)
                assert conv_variant(spec, dtype, n, h, w, "wgrad") == 2
                launches(3)
            lib.ps_debug_set_wgrad256(0)
            lib.ps_debug_set_wgrad_ws2(0)
            old = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
            ops.conv2d_wgrad(spec, xd, gyd, old)
        else:
            launches(3)
    finally:
        if library == "debug":
            lib.ps_debug_set_wgrad_ws2(1)
            lib.ps_debug_set_wgrad256(0)
    ref = w_fwd_layout(wt.grad)
    for dw in got:
        assert rel_err(dw.cpu(), ref) < 1e-4   # exact products, f32 accumulation
        assert old is None or rel_err(dw.cpu(), old.cpu()) < 1e-5
        assert rel_err(dw.cpu(), got[0].cpu()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, s, d)
    (2, 13, 10, 128, 256, 3, 2, 1), (3, 9, 11, 256, 512, 1, 2, 1), (2, 13, 10, 512, 64, 1, 1, 1), (6, 28, 28, 256, 192, 1, 1, 1),  # 128x128 .. 64x64 tiles, split over pixel ranges
    (16, 28, 28, 512, 512, 3, 1, 1), (8, 28, 28, 2048, 4096, 1, 1, 1), (21, 56, 56, 256, 512, 3, 2, 1),  # 16-bit: the persistent 256x128 kernel by geometry
])
def test_wgrad_deterministic_mode(case, dtype, library):
    """ps_conv2d_wgrad_det (pixel ranges store partial sums into a workspace, a second kernel adds them in range order): ACCUMULATES
    into dw like the atomic entry point, matches CPU autograd, and is BIT-IDENTICAL from launch to launch -- the reference's
    torch.use_deterministic_algorithms(True) / Trainer(deterministic=True) (revise_pseudo_labels.py:140-146, segmentation_train.py:153-160)."""
    import ctypes as C

    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, k, s_, d = case
    spec = ops.ConvSpec(cin, cout, k, s_, d)
    g = torch.Generator().manual_seed(cin + cout + k + n)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g))
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, stride=s_, padding=d if k == 3 else 0, dilation=d)
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    D = dev()
    xd, gyd = nhwc(x).to(D, dtype), nhwc(gy).to(D, dtype)
    init = torch.randn(cout, k, k, cin, generator=g)
    geom = ops._geom(spec, ops._dt(dtype), n, h, w, cin, cout)
    need = int(lib.ps_conv2d_wgrad_det_workspace_bytes(C.byref(geom)))
    assert need >= 0 and need % (cout * k * k * cin * 4) == 0
    outs = []
    for rep in range(3):
        dw = init.clone().to(D)
        if rep == 2:  # garbage in the workspace must not matter
            ops._wgrad_workspace(max(need, 16), D, ops._stream()).fill_(0x7F)
        ops.conv2d_wgrad(spec, xd, gyd, dw, deterministic=True)
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = w_fwd_layout(wt.grad)
    tol = F32_TOL if dtype == torch.float32 else 1e-4  # 16-bit: exact products of identically rounded operands, f32 accumulation
    assert rel_err((outs[0].cpu() - init), ref) < tol
    atomic = init.clone().to(D)
    ops.conv2d_wgrad(spec, xd, gyd, atomic, deterministic=False)
    assert rel_err(outs[0].cpu() - init, atomic.cpu() - init) < 1e-5
    if need > 0:  # a split problem without (enough) workspace is refused, not silently run on atomics
        dw = init.clone().to(D)
        rc = lib.ps_conv2d_wgrad_det(C.byref(geom), xd.data_ptr(), gyd.data_ptr(), dw.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream)
        assert rc != 0 and b"workspace" in lib.ps_last_error()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case,family", [
    # (n, h, w, cin, cout, k, s, d): the smallest batches at which the GEOMETRY selects each persistent kernel (>= 256 tiles), ragged:
    ((19, 28, 28, 128, 512, 3, 1, 1), "halo"),    # 66.5 pixel tiles of 8 x 28: straddles images, ragged last tile; tail half tiles
    ((10, 56, 56, 64, 256, 3, 1, 2), "halo"),     # two column blocks per row, dilation 2
    ((17, 32, 32, 64, 512, 3, 1, 4), "halo"),     # 256-pixel tiles (the maps of 256 x 256 inputs), dilation 4
    ((11, 28, 28, 2048, 4096, 1, 1, 1), "gemm256"), # b7's 1x1 shape: the 256 x 256 tile GEMM kernel, 33.7 pixel tiles (ragged)
    ((21, 28, 28, 2048, 2048, 1, 1, 1), "gemm256"), # odd image count: ragged last pixel tile (64.3 tiles of 256 pixels)
    ((32, 28, 28, 2048, 1024, 1, 1, 1), "gemm256"), # 1024 produced channels, K = 2048 at the stage-3 batch: 98 x 4 = 392 tiles, the rule's lower edge (384)
    ((31, 28, 28, 2048, 1024, 1, 1, 1), "ws2"),     # ... and one image fewer: 95 x 4 = 380 tiles -> the large-tile persistent kernel
    ((37, 28, 28, 512, 1024, 1, 1, 1), "ws2"),      # K = 512: below the 256 x 256 kernel's range -> ws2, ragged last pixel tile
    ((21, 56, 56, 256, 512, 3, 2, 1), "ws2"),       # stride-2 3x3 (b4's first conv)
])
def test_persistent_kernels_selected_by_geometry_match_cpu(case, family, dtype, library):
    """The kernels the benchmark runs (conv_igemm_halo_kernel, conv_igemm_ws2_kernel, conv_wgrad_ws2_kernel), reached WITHOUT any debug
    switch -- in the product library that is the only way to reach them -- against CPU autograd on identically rounded operands:
    forward with the full epilogue (residual add, raw output, BN + ReLU + dropout output), data gradient with the ReLU-mask epilogue,
    weight gradient (resnet38d.py:16-21,38-41,64,86)."""
    from pistoseg_amd import ops

    n, h, w, cin, cout, k, s_, d = case
    spec = ops.ConvSpec(cin, cout, k, s_, d)
    want = {"halo": (V_HALO,), "ws2": (V_WS2_256, V_WS2_224), "gemm256": (V_GEMM256,)}[family]
    assert conv_variant(spec, dtype, n, h, w, "fwd") in want
    wg_ws2 = conv_variant(spec, dtype, n, h, w, "wgrad") in (1, 2)
    assert wg_ws2 or cout % 256 or cin % 128  # every eligible shape here is big enough for the persistent weight-gradient kernel
    g = torch.Generator().manual_seed(sum(case))
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = q(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    pad = d if k == 3 else 0
    y = F.conv2d(x, wt, stride=s_, padding=pad, dilation=d)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    drop = (torch.rand(n, cout, generator=g) > 0.3).float() / 0.7
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) * drop.view(n, cout, 1, 1)
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    mask_src = q(F.relu(torch.randn(n, cin, h, w, generator=g)))
    sc2 = torch.rand(cin, generator=g) + 0.5
    gx_ref = torch.where(mask_src > 0, x.grad * sc2.view(1, -1, 1, 1), torch.zeros(()))
    tol = TOL[dtype]
    D = dev()
    ho, wo = spec.out_hw(h, w)
    xd, wf, wd = nhwc(x.detach()).to(D, dtype), w_fwd_layout(wt.detach()).to(D, dtype), w_dgrad_layout(wt.detach()).to(D, dtype)
    gyd = nhwc(gy).to(D, dtype)
    out_raw = torch.full((n, ho, wo, cout), float("nan"), device=D, dtype=dtype)
    out_act = torch.full((n, ho, wo, cout), float("nan"), device=D, dtype=dtype)
    ops.conv2d_fwd(spec, xd, wf, add0=nhwc(res).to(D, dtype), out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), drop=drop.to(D), out_act=out_act)
    assert rel_err(out_raw.float().cpu(), nhwc((y + res).detach())) < tol
    assert rel_err(out_act.float().cpu(), nhwc(act.detach())) < tol
    gx = torch.full((n, h, w, cin), float("nan"), device=D, dtype=dtype)
    ops.conv2d_dgrad(spec, gyd, wd, (h, w), mask_src=nhwc(mask_src).to(D, dtype), bn_scale=sc2.to(D), out=gx)
    assert rel_err(gx.float().cpu(), nhwc(gx_ref)) < tol
    dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
    ops.conv2d_wgrad(spec, xd, gyd, dw)
    assert rel_err(dw.cpu(), w_fwd_layout(wt.grad)) < 1e-4  # exact products of 16-bit operands, f32 accumulation


@both_libraries
@pytest.mark.parametrize("case", [(512, 512, 3, 1, 28), (1024, 2048, 3, 4, 28), (256, 256, 3, 1, 56), (2048, 4096, 1, 1, 28)])
def test_full_size_layers_kernel_families_agree(case, library):
    """BASELINE-size layers (bs = 64, bf16), too big for a full CPU reference inside the suite: size-independent checks instead, plus
    image 0's forward on the CPU.
    (1, debug library) the production kernel for the layer (halo / large-tile) against the small-tile two-blocks-per-CU family on the same data --
    different tilings, staging and (halo) K order, so agreement within bf16 output rounding is strong evidence for both;
    (2) linearity of the forward in its input; (3) <dy, conv(x)> == <dgrad(dy), x> == <wgrad(x, dy), w> (adjoint identities, f32 sums)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    cin, cout, k, d, hw = case
    n, dtype, D = 64, torch.bfloat16, dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + k)
    x = torch.randn(n, hw, hw, cin, generator=g).to(D, dtype)
    gy = torch.randn(n, hw, hw, cout, generator=g).to(D, dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k)) ** 0.5
    wf, wd = w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
    spec = ops.ConvSpec(cin, cout, k, 1, d)

    def run(xin):
        y = torch.empty((n, hw, hw, cout), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xin, wf, out_raw=y)
        return y

    def run_bwd():
        gx = torch.empty((n, hw, hw, cin), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, gy, wd, (hw, hw), out_raw=gx)
        dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
        ops.conv2d_wgrad(spec, x, gy, dw)
        return gx, dw

    # the persistent kernels are what the geometry selects, in either library
    want = (V_HALO,) if k == 3 else (V_GEMM256,)
    for kind in ("fwd", "dgrad"):
        assert conv_variant(spec, dtype, n, hw, hw, kind) in want, kind
    assert conv_variant(spec, dtype, n, hw, hw, "wgrad") == 1
    y, (gx, dw) = run(x), run_bwd()
    if library == "debug":
        try:
            lib.ps_debug_set_halo(0)
            lib.ps_debug_set_ws2(0)
            lib.ps_debug_set_wgrad_ws2(0)
            y_small, (gx_small, dw_small) = run(x), run_bwd()
        finally:
            lib.ps_debug_set_halo(1)
            lib.ps_debug_set_ws2(WS2_DEFAULT)
            lib.ps_debug_set_wgrad_ws2(1)
        assert rel_err(y.float(), y_small.float()) < 8e-3 and rel_err(gx.float(), gx_small.float()) < 8e-3  # one bf16 ulp of the largest output
        assert float((y.float() - y_small.float()).abs().mean() / y_small.float().abs().mean()) < 1e-3
        assert rel_err(dw, dw_small) < 1e-4
    # images 0 and 63 of the forward and of the data gradient on the CPU (same bf16-rounded operands)
    wq = wt.to(dtype).float()
    for i in (0, n - 1):
        ref = F.conv2d(x[i:i + 1].float().cpu().permute(0, 3, 1, 2), wq, padding=d if k == 3 else 0, dilation=d)
        assert rel_err(y[i:i + 1].float().cpu(), nhwc(ref)) < BF16_TOL, i
        ref = torch.nn.grad.conv2d_input((1, cin, hw, hw), wq, gy[i:i + 1].float().cpu().permute(0, 3, 1, 2), padding=d if k == 3 else 0, dilation=d)
        assert rel_err(gx[i:i + 1].float().cpu(), nhwc(ref)) < BF16_TOL, i
    # linearity: conv(x + 2 x2) == conv(x) + 2 conv(x2) up to output rounding
    x2 = torch.randn(n, hw, hw, cin, generator=g).to(D, dtype)
    lhs = run((x.float() + 2 * x2.float()).to(dtype)).float()
    rhs = y.float() + 2 * run(x2).float()
    assert float((lhs - rhs).abs().mean() / rhs.abs().mean()) < 2e-2
    # adjoint identities <dy, conv(x)> == <dgrad(dy), x> == <wgrad(x, dy), w>, f64 sums of the stored results.  y and gx carry an
    # independent bf16 rounding error per element (relative 2^-9 uniform): 6 sigma of the resulting error of each inner product
    def ip(a_, b_):
        p_ = a_.double() * b_.double()
        return float(p_.sum()), float((p_ * p_).sum()) ** 0.5 * 2.0 ** -9
    (a1, s1), (a2, s2) = ip(gy, y), ip(gx, x)
    a3 = float((dw.double() * wf.double()).sum())  # dw is f32: no storage rounding
    assert abs(a1 - a2) < 6 * (s1 + s2), (a1, a2, s1, s2)
    assert abs(a1 - a3) < 6 * s1 + 1e-4 * abs(a3), (a1, a3, s1)


@pytest.mark.parametrize("tpb", [1, 2, 5, -16, -100, -4000])  # negative: ps_conv_geom.cus_reserved = -tpb instead
@pytest.mark.selfcheck
def test_persistent_kernels_batched_work_split_is_exact(tpb, library):
    """ps_conv_geom.tiles_per_block = n: the persistent kernels' blocks are dispatched in batches and take n work items each;
    ps_conv_geom.cus_reserved = r: their grid and static schedule are sized for #CUs - r compute units (both used while an all-reduce
    shares the GPU; 4000 reserved: clamped to a quarter of the device).  Only the item -> block assignment changes: forward / data
    gradient are bit-identical to the default schedule, the weight gradient up to f32 atomic ordering (its split-K plan follows the CU count)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    D = dev()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(31 + tpb)
    # > 256 tiles / items each: halo (3x3 on 28x28), ws2 (1x1), wgrad ws2
    cases = [(40, 28, 28, 256, 256, 3, 2), (36, 28, 28, 512, 512, 1, 1)]
    for n, h, w, cin, cout, k, d in cases:
        x = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
        wt = (torch.randn(cout, cin, k, k, generator=g) * 0.03)
        wf, wd = w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
        gy = torch.randn(n, h, w, cout, generator=g).to(D, dtype)
        spec = ops.ConvSpec(cin, cout, k, 1, d)

        def run():
            y = torch.empty((n, h, w, cout), device=D, dtype=dtype)
            ops.conv2d_fwd(spec, x, wf, out_raw=y)
            gx = torch.empty((n, h, w, cin), device=D, dtype=dtype)
            ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx)
            dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
            ops.conv2d_wgrad(spec, x, gy, dw)
            return y, gx, dw

        # every launch here is served by a persistent kernel by the geometry alone (no switch needed in either library)
        assert conv_variant(spec, dtype, n, h, w, "fwd") in (V_HALO, V_WS2_224, V_WS2_256)
        assert conv_variant(spec, dtype, n, h, w, "wgrad") in (1, 2)
        try:
            ops.TILES_PER_BLOCK, ops.CUS_RESERVED = 0, 0
            ref = run()
            ops.TILES_PER_BLOCK, ops.CUS_RESERVED = (tpb, 0) if tpb > 0 else (0, -tpb)
            got = run()
        finally:
            ops.TILES_PER_BLOCK, ops.CUS_RESERVED = 0, 0
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        assert rel_err(got[2].cpu(), ref[2].cpu()) < 1e-5


@debug_only
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout): 1x1 stride 1.  K-tiles (cin / 64): 4 (the minimum), 5 (odd: the unpaired tail K-tile), 8, 17; pixel counts that
    # are not multiples of 256 (zero-filled tail rows, rows dropped by the epilogue), down to a tile with 2 live rows; 1..3 cout tiles
    (3, 15, 14, 256, 256), (1, 16, 16, 320, 512), (2, 13, 10, 512, 768), (1, 2, 129, 1088, 256), (5, 28, 28, 512, 512),
])
def test_gemm256_kernel_forced_on_small_problems(case, dtype):
    """conv_gemm256_kernel (256 x 256 tile, eight MFMA waves in two alternating groups) forced on small ragged problems: forward with
    the full epilogue, data gradient with the ReLU-mask epilogue -- against the CPU, against the ws2 / 4-wave kernels (same MFMA chain
    per output element: BIT-IDENTICAL), and repeated (race screen: every LDS hand-off in this kernel is ordered by counted waits)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case))
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = q(torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5)
    y = F.conv2d(x, wt)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    drop = (torch.rand(n, cout, generator=g) > 0.3).float() / 0.7
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) * drop.view(n, cout, 1, 1)
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    # data gradient of a layer with `cout` inputs: produces cout channels from cin
    wt_t = q(torch.randn(cin, cout, 1, 1, generator=g) * 0.05)
    gy_t = q(torch.randn(n, cin, h, w, generator=g))
    mask_src = q(F.relu(torch.randn(n, cout, h, w, generator=g)))
    sc2 = torch.rand(cout, generator=g) + 0.5
    gx_ref = torch.where(mask_src > 0, torch.nn.grad.conv2d_input((n, cout, h, w), wt_t, gy_t) * sc2.view(1, -1, 1, 1), torch.zeros(()))
    tol = TOL[dtype]
    spec, spec_t = ops.ConvSpec(cin, cout, 1, 1, 1), ops.ConvSpec(cout, cin, 1, 1, 1)
    D = dev()
    xd, wf, resd = nhwc(x.detach()).to(D, dtype), w_fwd_layout(wt).to(D, dtype), nhwc(res).to(D, dtype)
    gyd, wdt, maskd = nhwc(gy_t).to(D, dtype), w_dgrad_layout(wt_t).to(D, dtype), nhwc(mask_src).to(D, dtype)

    def run():
        out_raw = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        out_act = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), drop=drop.to(D), out_act=out_act)
        gx = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec_t, gyd, wdt, (h, w), mask_src=maskd, bn_scale=sc2.to(D), out=gx)
        return out_raw, out_act, gx

    try:
        lib.ps_debug_set_gemm256(2)
        assert conv_variant(spec, dtype, n, h, w, "fwd") == V_GEMM256 and conv_variant(spec_t, dtype, n, h, w, "dgrad") == V_GEMM256
        got = [run() for _ in range(4)]
        lib.ps_debug_set_gemm256(0)
        assert conv_variant(spec, dtype, n, h, w, "fwd") != V_GEMM256
        other = run()
    finally:
        lib.ps_debug_set_gemm256(1)
    refs = (nhwc((y + res).detach()), nhwc(act.detach()), nhwc(gx_ref))
    for idx, (a_, r_, o_) in enumerate(zip(got[0], refs, other)):
        assert rel_err(a_.float().cpu(), r_) < tol
        if idx != 1:  # K order and MFMA chain per output element are the other kernels': raw output and masked gradient bit for bit
            assert torch.equal(a_, o_)
        else:  # dropout output: a wave whose rows lie in one image folds the multiplier into the BN affine (one f32 rounding less), and
            # which waves do depends on the kernel's wave tile -- equal up to that rounding
            assert rel_err(a_.float().cpu(), o_.float().cpu()) < 2.0 ** -9
    for trial in got[1:]:
        assert all(torch.equal(a_, b_) for a_, b_ in zip(trial, got[0]))


@debug_only
@pytest.mark.selfcheck
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(21, 28, 28, 2048, 2048), (11, 28, 28, 2048, 4096)])  # 520 = 2 x 256 + 8 tiles; 544 = 2 x 256 + 32 tiles
def test_gemm256_tail_rows_on_the_gathered_tile_kernels(case, dtype):
    """A conv_gemm256_kernel launch whose tile count leaves a partial last round of at most half the CUs hands those pixel rows to a second
    launch on the gathered-tile kernels (every tensor pointer advanced, `m_off` for the dropout image index): same MFMA chain per output
    element, so forward (raw) and masked data gradient are BIT-IDENTICAL to the single launch; the dropout output up to the fold rounding."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case))
    D = dev()
    x = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
    wf = (torch.randn(cout, 1, 1, cin, generator=g) * (2.0 / cin) ** 0.5).to(D, dtype)
    res = torch.randn(n, h, w, cout, generator=g).to(D, dtype)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(D), (torch.randn(cout, generator=g) * 0.1).to(D)
    drop = ((torch.rand(n, cout, generator=g) > 0.3).float() / 0.7).to(D)
    gy = torch.randn(n, h, w, cin, generator=g).to(D, dtype)       # data gradient of a layer with `cout` inputs: produces cout channels
    wd = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).to(D, dtype)
    mask = torch.relu(torch.randn(n, h, w, cout, generator=g)).to(D, dtype)
    sc2 = (torch.rand(cout, generator=g) + 0.5).to(D)
    spec, spec_t = ops.ConvSpec(cin, cout, 1, 1, 1), ops.ConvSpec(cout, cin, 1, 1, 1)
    assert conv_variant(spec, dtype, n, h, w, "fwd") == V_GEMM256 and conv_variant(spec_t, dtype, n, h, w, "dgrad") == V_GEMM256

    def run():
        raw = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        act = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, x, wf, add0=res, out_raw=raw, bn_scale=scale, bn_shift=shift, drop=drop, out_act=act)
        gx = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec_t, gy, wd, (h, w), mask_src=mask, bn_scale=sc2, out=gx)
        return raw, act, gx

    try:
        lib.ps_debug_set_gemm256_tail(1)
        split = run()
        lib.ps_debug_set_gemm256_tail(0)
        single = run()
    finally:
        lib.ps_debug_set_gemm256_tail(1)
    assert torch.equal(split[0], single[0]) and torch.equal(split[2], single[2])
    assert not torch.isnan(split[1].float()).any() and rel_err(split[1].float(), single[1].float()) < 2.0 ** -9
    try:  # ps_conv_geom.gpu_shared = 1: the single-launch schedule through the launch option (what the two-stream backward asks for)
        ops.GPU_SHARED = 1
        shared = run()
    finally:
        ops.GPU_SHARED = 0
    assert all(torch.equal(a_, b_) for a_, b_ in zip(shared, single))


@debug_only
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bm", [112, 128, 224, 256])
@pytest.mark.parametrize("case", [(128, 256, 3, 2, 1), (256, 256, 3, 1, 2), (512, 128, 1, 1, 1)])
def test_conv_pixel_tile_variants(case, bm, dtype):
    """The 112-pixel (7-fragment, waves 1x4) and 128-pixel (waves 2x2) tilings of the 128-cout kernel and the 224- /
    256-pixel tilings of the large-tile wave-specialised kernel, forced on a small problem with ragged M (tiles with
    zero-filled tail rows, 1-2 K-steps up to 36), against the CPU (fwd with the full epilogue, and dgrad)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    cin, cout, k, s, d = case
    n, h, w = 3, 15, 14
    g = torch.Generator().manual_seed(cin + bm)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = q(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    y = F.conv2d(x, wt, stride=s, padding=d if k == 3 else 0, dilation=d)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    gy = q(torch.randn(y.shape, generator=g))
    y.backward(gy)
    tol = TOL[dtype]
    spec = ops.ConvSpec(cin, cout, k, s, d)
    D = dev()
    try:
        if bm >= 224:
            lib.ps_debug_set_ws2(bm)
        else:
            lib.ps_debug_set_bn(128)
            lib.ps_debug_set_bm(bm)
        ho, wo = spec.out_hw(h, w)
        out_raw = torch.full((n, ho, wo, cout), float("nan"), device=D, dtype=dtype)
        out_act = torch.full((n, ho, wo, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, nhwc(x.detach()).to(D, dtype), w_fwd_layout(wt).to(D, dtype), add0=nhwc(res).to(D, dtype), out_raw=out_raw,
                       bn_scale=scale.to(D), bn_shift=shift.to(D), out_act=out_act)
        assert rel_err(out_raw.float().cpu(), nhwc((y + res).detach())) < tol
        assert rel_err(out_act.float().cpu(), nhwc(act.detach())) < tol
        gx = torch.full((n, h, w, cin), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, nhwc(gy).to(D, dtype), w_dgrad_layout(wt).to(D, dtype), (h, w), out_raw=gx)
        assert rel_err(gx.float().cpu(), nhwc(x.grad)) < tol
    finally:
        lib.ps_debug_set_bn(0)
        lib.ps_debug_set_bm(0)
        lib.ps_debug_set_ws2(WS2_DEFAULT)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_weight_transpose_batched_matches_permute(dtype):
    """ps_weight_transpose_batched: many tensors in one launch, bit-exact against torch's permute -- shapes on the 64 x 64
    16-byte path (multiples of 64), on the element-wise path (ragged), and two 1x1 tensors side by side in one K-concatenated
    destination (row stride > cout), the layout of the fused bottleneck units."""
    from pistoseg_amd import ops

    d = dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(128, 9, 64), (64, 1, 192), (50, 9, 20), (3, 1, 4096), (256, 9, 128)]
    items, want = [], []
    for cout, taps, cin in shapes:
        src = torch.randn(cout, taps, cin, generator=g).to(dtype).to(d)
        dst = torch.full((cin * taps, cout), 7.0, device=d, dtype=dtype)
        items.append((src.reshape(cout, taps * cin), dst, cout, taps, cin))
        want.append(src.permute(2, 1, 0).reshape(cin * taps, cout))
    cin, c1, c2 = 128, 192, 64  # K-concatenated pair
    cat = torch.full((cin, c1 + c2), 7.0, device=d, dtype=dtype)
    s1 = torch.randn(c1, 1, cin, generator=g).to(dtype).to(d)
    s2 = torch.randn(c2, 1, cin, generator=g).to(dtype).to(d)
    items += [(s1.reshape(c1, cin), cat[:, :c1], c1, 1, cin), (s2.reshape(c2, cin), cat[:, c1:], c2, 1, cin)]
    ops.weight_transpose_batched(items)
    torch.cuda.synchronize()
    for (src, dst, *_), w in zip(items, want):
        assert torch.equal(dst, w)
    assert torch.equal(cat, torch.cat([s1.reshape(c1, cin).t(), s2.reshape(c2, cin).t()], dim=1))
    # more items than one launch carries
    many = []
    for i in range(60):
        src = torch.randn(64, 1, 64, generator=g).to(dtype).to(d)
        many.append((src.reshape(64, 64), torch.empty(64, 64, device=d, dtype=dtype), 64, 1, 64))
    ops.weight_transpose_batched(many)
    torch.cuda.synchronize()
    for src, dst, *_ in many:
        assert torch.equal(dst, src.t())


@both_libraries
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
# cin, cout, H, W, N (odd sizes: ragged classes); the last case is b4's first conv at a batch where the geometry itself selects the split
@pytest.mark.parametrize("case", [(128, 256, 56, 56, 2), (256, 512, 28, 28, 3), (128, 128, 23, 31, 2), (256, 512, 56, 56, 37)])
def test_stride2_dgrad_parity_class_launches_are_bit_identical(case, dtype, library):
    """Stride-2 3x3 data gradient as four parity-class launches (only the live taps of each class) vs the one-launch path that gathers
    all nine taps per pixel: skipping zero contributions does not change any f32 sum, so the outputs must be bit-identical -- with the
    ReLU-mask + add1 epilogue and with a plain store (debug library: both schedules forced); and against the CPU primitive (both
    libraries; the product library splits exactly when the geometry says so)."""
    from pistoseg_amd import _lib, ops

    cin, cout, h, w, n = case
    lib = _lib.load()
    d = dev()
    g = torch.Generator().manual_seed(11)
    spec = ops.ConvSpec(cin, cout, 3, 2, 1)
    ho, wo = spec.out_hw(h, w)
    gy = torch.randn(n, ho, wo, cout, generator=g).to(dtype).to(d)
    wd = (torch.randn(cin, 3, 3, cout, generator=g) * 0.05).to(dtype).to(d)
    act = torch.randn(n, h, w, cin, generator=g).to(dtype).to(d)
    add = torch.randn(n, h, w, cin, generator=g).to(dtype).to(d)
    sc = (torch.rand(cin, generator=g) + 0.5).to(d)
    outs = []
    try:
        for mode in ((0, 2) if library == "debug" else (1, 1)):
            if library == "debug":
                lib.ps_debug_set_s2split(mode)
            a = torch.full((n, h, w, cin), 7.0, device=d, dtype=dtype)
            b = torch.full((n, h, w, cin), 7.0, device=d, dtype=dtype)
            ops.conv2d_dgrad(spec, gy, wd, (h, w), mask_src=act, bn_scale=sc, add1=add, out=a)
            ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=b)
            torch.cuda.synchronize()
            outs.append((a, b))
    finally:
        if library == "debug":
            lib.ps_debug_set_s2split(1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and against the CPU primitive (same check as test_conv_fwd_dgrad_wgrad)
    q = quant(dtype)
    wt = wd.float().cpu().permute(3, 0, 1, 2).contiguous()  # [cin][kh][kw][cout] -> OIHW of the forward conv
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt, q(gy.float().cpu()).permute(0, 3, 1, 2), stride=2, padding=1)
    assert rel_err(outs[1][1].float().cpu().permute(0, 3, 1, 2), ref) < TOL[dtype]
    full = torch.where(act.float().cpu().permute(0, 3, 1, 2) > 0, ref * sc.cpu().view(1, -1, 1, 1), torch.zeros(())) + add.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(outs[1][0].float().cpu().permute(0, 3, 1, 2), full) < TOL[dtype]


def test_dropout2d_masks_kernel():
    """One launch draws every Dropout2d mask of a step: values in {0, 1/(1-p)}, keep rate 1-p per segment, repeatable per (seed, offset),
    fresh per offset; the model's sample_dropout hands out views of that one buffer under the names the plans use."""
    from pistoseg_amd import ops
    from pistoseg_amd.seg_model import ResNet38dSeg

    D = dev()
    segs = [("a", 512, 0.3), ("b", 1024, 0.3), ("c", 4096, 0.5), ("d", 7, 0.0)]
    n = 64
    m1 = ops.dropout2d_masks(segs, n, D, seed=1234, offset=1)
    m1b = ops.dropout2d_masks(segs, n, D, seed=1234, offset=1)
    m2 = ops.dropout2d_masks(segs, n, D, seed=1234, offset=2)
    m3 = ops.dropout2d_masks(segs, n, D, seed=1235, offset=1)
    for name, c, p in segs:
        t = m1[name]
        assert tuple(t.shape) == (n, c) and t.dtype == torch.float32
        vals = torch.unique(t).cpu().tolist()
        assert all(abs(v) < 1e-12 or abs(v - 1.0 / (1.0 - p)) < 1e-6 for v in vals), (name, vals)
        keep = float((t > 0).float().mean())
        assert abs(keep - (1.0 - p)) < 4.0 * (p * (1 - p) / (n * c)) ** 0.5 + 1e-9, (name, keep)
        assert torch.equal(t, m1b[name])
        if p > 0:
            assert not torch.equal(t, m2[name]) and not torch.equal(t, m3[name])
    # rows (samples) are not copies of each other, channels neither
    assert not torch.equal(m1["c"][0], m1["c"][1]) and not torch.equal(m1["c"][:, 0], m1["c"][:, 1])
    model = ResNet38dSeg(3, "bf16")
    torch.manual_seed(7)
    d1 = model.sample_dropout(4, D)
    assert sorted(d1) == ["b6.dropout_2b1", "b6.dropout_2b2", "b7.dropout_2b1", "b7.dropout_2b2", "dropout7"]
    assert tuple(d1["b7.dropout_2b2"].shape) == (4, 2048) and tuple(d1["dropout7"].shape) == (4, 4096)
    d2 = model.sample_dropout(4, D)
    assert not torch.equal(d1["dropout7"], d2["dropout7"])


@both_libraries
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, d): tile counts that leave a partial last round of <= half the 256 CUs, in whole pixel tiles:
    # 24 x 784 / 224 = 84 pixel tiles x 4 cout tiles = 336 = 256 + 80;  20 x 1024 / 256 = 80 x 4 = 320 = 256 + 64 (256-pixel tiles);
    # 9 x 56 x 56 / 224 = 126 x 3 = 378 = 256 + 122 with 122 % 3 != 0 -> NOT split (the tail must be whole pixel tiles)
    (24, 28, 28, 64, 512, 1), (24, 28, 28, 128, 512, 4), (20, 32, 32, 64, 512, 2), (9, 56, 56, 64, 384, 1),
])
def test_conv_halo_tail_as_half_tiles(case, dtype, library):
    """The partial last round of a halo launch goes to a second launch of 64-cout half tiles: same MFMA chain per output element, so
    forward (full epilogue) and data gradient are BIT-IDENTICAL to the single-launch schedule (debug library: both forced), and match
    the CPU (both libraries: these geometries select the halo kernel and its tail split by themselves)."""
    from pistoseg_amd import _lib, ops

    lib = _lib.load()
    n, h, w, cin, cout, d = case
    g = torch.Generator().manual_seed(h + cin + cout + d)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = q(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
    y = F.conv2d(x, wt, padding=d, dilation=d)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    spec = ops.ConvSpec(cin, cout, 3, 1, d)
    D = dev()
    xd, wf = nhwc(x.detach()).to(D, dtype), w_fwd_layout(wt).to(D, dtype)
    resd = nhwc(res).to(D, dtype)
    # the data gradient of a layer with `cout` INPUT channels produces `cout` channels: use the transposed role of the same weights
    gy = q(torch.randn(n, cin, h, w, generator=g))
    wt_t = q(torch.randn(cin, cout, 3, 3, generator=g) * 0.05)  # conv cout -> cin; its dgrad produces cout channels
    gx_ref = torch.nn.grad.conv2d_input((n, cout, h, w), wt_t, gy, padding=d, dilation=d)
    spec_t = ops.ConvSpec(cout, cin, 3, 1, d)
    gyd, wdt = nhwc(gy).to(D, dtype), w_dgrad_layout(wt_t).to(D, dtype)

    def run():
        out_raw = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        out_act = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), out_act=out_act)
        gx = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec_t, gyd, wdt, (h, w), out_raw=gx)
        return out_raw, out_act, gx

    assert conv_variant(spec, dtype, n, h, w, "fwd") == V_HALO and conv_variant(spec_t, dtype, n, h, w, "dgrad") == V_HALO
    stream_k_before, ops.STREAM_K = ops.STREAM_K, False  # (this test is about the static schedule's tail launch; stream-K: test_conv_halo_stream_k)
    request_finalizer = lambda: setattr(ops, "STREAM_K", stream_k_before)  # noqa: E731
    try:
        _halo_tail_body(lib, ops, library, run, y, res, act, gx_ref, dtype)
    finally:
        request_finalizer()


def _halo_tail_body(lib, ops, library, run, y, res, act, gx_ref, dtype):
    if library == "debug":
        try:
            lib.ps_debug_set_halo_tail(1)
            split = [run() for _ in range(2)]
            lib.ps_debug_set_halo_tail(0)
            single = run()
        finally:
            lib.ps_debug_set_halo_tail(1)
        for a_, b_ in zip(split[0], single):
            assert torch.equal(a_, b_)
    else:
        split = [run() for _ in range(2)]
    assert all(torch.equal(a_, b_) for a_, b_ in zip(split[0], split[1]))
    # the launch option ps_conv_geom.gpu_shared (another stream fills the partial last round: no tail launch) -- on the PRODUCT library this
    # is the single-launch schedule, so the product's own split is checked bit for bit as well
    try:
        ops.GPU_SHARED = 1
        shared = run()
    finally:
        ops.GPU_SHARED = 0
    assert all(torch.equal(a_, b_) for a_, b_ in zip(split[0], shared))
    tol = TOL[dtype]
    for a_, r_ in zip(split[0], (nhwc((y + res).detach()), nhwc(act.detach()), nhwc(gx_ref))):
        assert rel_err(a_.float().cpu(), r_) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, d, reserved CUs, stream-K expected): T tiles on nb CUs -> whole rounds + a partial round of R tiles cut along K into nb shares
    # (shares of >= 4 K-lines only: the library's rule, measured in profiles/r05c_convbench_stream_k.txt)
    (32, 28, 28, 512, 512, 1, 0, True),    # BASELINE configs[3]'s per-GPU layer: 448 tiles = 1 round + 192; 8 K-lines -> shares of 6 lines, two parts per tile
    (24, 28, 28, 1024, 512, 2, 0, True),   # 336 = 256 + 80; 16 K-lines -> shares of 5 lines, tiles in 3-5 parts (the ordered re-sum path)
    (20, 32, 32, 1024, 512, 4, 0, False),  # 256-pixel tiles: 320 = 256 + 64 -- the static schedule (+ its half-tile tail): the 256-pixel stream-K instance spills, measured slower
    (32, 28, 28, 128, 512, 1, 0, False),   # 448 tiles of 2 K-lines: shares of 1.5 lines -- not worth the exchange, the static schedule stays
    (32, 28, 28, 512, 512, 1, 32, False),  # 448 tiles beside a collective that holds 32 CUs (ps_conv_geom.cus_reserved): 2 x 224, whole rounds, nothing to cut
    (30, 28, 28, 512, 512, 2, 32, True),   # 420 = 224 + 196 on 224 CUs: shares of 7 lines
    (16, 28, 28, 512, 512, 1, 0, False),   # run.sh's own batch size: 224 tiles < 256 CUs -- the halo kernel, one tile per CU on 224 of them (cutting EVERY tile loses: measured)
    (10, 28, 28, 512, 512, 2, 0, False),   # 140 tiles: the smallest launches the halo kernel serves (half a round)
    (45, 28, 28, 1024, 512, 4, 0, True),   # odd image count: 157.5 pixel tiles -> the RAGGED last tile lies in the stream-K region (632 = 2 x 256 + 120; shares of 7.5 lines)
])
def test_conv_halo_stream_k(case, dtype):
    """Stream-K finish of the halo kernel's partial last round (ps_epilogue.sk_ws): forward with the full epilogue (residual, raw + BN/ReLU outputs)
    and data gradient against the CPU, bit-identical from run to run (fixed part order), and equal to the static schedule up to the f32
    re-association at the split points (the partial sums are f32; the stored 16-bit values almost always round the same)."""
    import ctypes

    from pistoseg_amd import _lib, ops

    n, h, w, cin, cout, d, reserved, expect_sk = case
    g = torch.Generator().manual_seed(h + cin + cout + d + n)
    q = quant(dtype)
    x = q(torch.randn(n, cin, h, w, generator=g))
    wt = q(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
    y = F.conv2d(x, wt, padding=d, dilation=d)
    res = q(torch.randn(y.shape, generator=g))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    act = F.relu((y + res) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    spec = ops.ConvSpec(cin, cout, 3, 1, d)
    D = dev()
    xd, wf = nhwc(x).to(D, dtype), w_fwd_layout(wt).to(D, dtype)
    resd = nhwc(res).to(D, dtype)
    gy = q(torch.randn(n, cin, h, w, generator=g))
    wt_t = q(torch.randn(cin, cout, 3, 3, generator=g) * 0.05)  # conv cout -> cin; its dgrad produces cout channels
    gx_ref = torch.nn.grad.conv2d_input((n, cout, h, w), wt_t, gy, padding=d, dilation=d)
    spec_t = ops.ConvSpec(cout, cin, 3, 1, d)
    gyd, wdt = nhwc(gy).to(D, dtype), w_dgrad_layout(wt_t).to(D, dtype)

    def run():
        out_raw = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        out_act = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, xd, wf, add0=resd, out_raw=out_raw, bn_scale=scale.to(D), bn_shift=shift.to(D), out_act=out_act)
        gx = torch.full((n, h, w, cout), float("nan"), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec_t, gyd, wdt, (h, w), out_raw=gx)
        return out_raw, out_act, gx

    assert conv_variant(spec, dtype, n, h, w, "fwd") == V_HALO and conv_variant(spec_t, dtype, n, h, w, "dgrad") == V_HALO
    before = (ops.STREAM_K, ops.CUS_RESERVED)
    try:
        ops.CUS_RESERVED = reserved
        geom = ops._geom(spec, ops._dt(xd), n, h, w, cin, cout)
        need = int(_lib.load().ps_conv_sk_workspace_bytes(ctypes.byref(geom), 0))
        assert (need > 0) == expect_sk, need  # (the library's own plan for this geometry)
        ops.STREAM_K = True
        sk = [run() for _ in range(3)]
        ops.STREAM_K = False
        static = run()
    finally:
        ops.STREAM_K, ops.CUS_RESERVED = before
    for other in sk[1:]:
        assert all(torch.equal(a_, b_) for a_, b_ in zip(sk[0], other))  # deterministic: parts are added in part order, whoever arrives last
    tol = TOL[dtype]
    for a_, s_, r_ in zip(sk[0], static, (nhwc((y + res).detach()), nhwc(act.detach()), nhwc(gx_ref))):
        assert rel_err(a_.float().cpu(), r_) < tol
        diff = (a_.float() - s_.float()).abs()
        # the same products, summed in f32 with different association at <= 7 split points: a few 16-bit roundings flip by one ulp
        assert float(diff.max()) <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * float(s_.float().abs().max()) and float((diff > 0).float().mean()) < 2e-2
        if need == 0:
            assert torch.equal(a_, s_)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    # (n, h, w, cin, cout, k, d): halo (3x3) + the weight gradient's persistent kernel, more work items than CUs
    (40, 28, 28, 256, 256, 3, 2),    # 280 tiles: almost everything is a block's first draw; 4 K-lines
    (64, 28, 28, 128, 512, 3, 1),    # 896 tiles = 3.5 rounds (main launch from the queue + a static tail launch of half tiles); 2 K-lines: the shortest tile the queue serves
    (23, 56, 56, 128, 256, 3, 1),    # two column blocks per row, ragged last tile
    (17, 32, 32, 64, 512, 3, 4),     # 256-pixel tiles, ONE K-line: the dispatcher keeps the static kernel (nothing to compare, must still be identical)
    (36, 28, 28, 512, 512, 1, 1),    # 1x1: conv_igemm_ws2_kernel's queue (tiles of eight K-steps, drawn two ahead), the weight gradient's XM = 2 instantiation
    (37, 28, 28, 128, 512, 1, 1),    # 1x1 with TWO K-steps per tile: too short for the ws2 queue's two-tile lead (static kernel; must still be identical)
    (37, 28, 28, 192, 512, 1, 1),    # ... and THREE: the shortest tile the ws2 queue serves; ragged last pixel tile
])
def test_tile_queue_launch_option_is_exact(case, dtype, library):
    """ps_conv_geom.tile_queue = 1: every tile (halo kernel) / work item (weight gradient) is drawn from per-XCD ticket counters by the blocks
    themselves instead of a static schedule.  Only the block that computes a tile changes: forward and data gradient are BIT-IDENTICAL to the
    static schedule, the atomic weight gradient up to f32 ordering and the deterministic one bit for bit; the counters re-arm themselves
    (many launches through one stream's ring of counter blocks, and two streams at once)."""
    from pistoseg_amd import ops

    D = dev()
    n, h, w, cin, cout, k, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.03
    wf, wd = w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
    gy = torch.randn(n, h, w, cout, generator=g).to(D, dtype)
    res = torch.randn(n, h, w, cout, generator=g).to(D, dtype)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(D), (torch.randn(cout, generator=g) * 0.1).to(D)
    spec = ops.ConvSpec(cin, cout, k, 1, d)

    def run(det):
        y, a = torch.empty((n, h, w, cout), device=D, dtype=dtype), torch.empty((n, h, w, cout), device=D, dtype=dtype)
        ops.conv2d_fwd(spec, x, wf, add0=res, out_raw=y, bn_scale=scale, bn_shift=shift, out_act=a)
        gx = torch.empty((n, h, w, cin), device=D, dtype=dtype)
        ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx)
        dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
        ops.conv2d_wgrad(spec, x, gy, dw, deterministic=det)
        return y, a, gx, dw

    stream_k_before, ops.STREAM_K = ops.STREAM_K, False  # like for like: the queue hands out whole tiles; the static schedule's stream-K finish re-associates sums
    try:
        ops.TILE_QUEUE = 0
        ref, ref_det = run(False), run(True)
        ops.TILE_QUEUE = 1
        got, got_det = run(False), run(True)
        assert all(torch.equal(p, q) for p, q in zip(got[:3], ref[:3]))
        assert rel_err(got[3].cpu(), ref[3].cpu()) < 1e-5
        assert torch.equal(got_det[3], ref_det[3])  # a pixel range's partial sums do not depend on the block that made them
        # counters re-arm: > 256 queue launches on this stream (its ring of counter blocks wraps), checked at intervals and at the end
        bad = 0
        for i in range(300):
            y = torch.empty_like(ref[0])
            ops.conv2d_fwd(spec, x, wf, add0=res, out_raw=y)
            if i % 37 == 0 or i >= 297:
                bad += int(not torch.equal(y, ref[0]))
        assert bad == 0
        # two streams draw from their own rings at the same time
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for s in (s1, s2, s1, s2):
            with torch.cuda.stream(s):
                gx = torch.empty((n, h, w, cin), device=D, dtype=dtype)
                ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx)
                outs.append(gx)
        torch.cuda.synchronize()
        assert all(torch.equal(o, ref[2]) for o in outs)
    finally:
        ops.TILE_QUEUE = 0
        ops.STREAM_K = stream_k_before


@debug_only
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,h,w", [(2, 224, 224), (3, 34, 224), (1, 256, 256), (2, 6, 256)])
def test_conv_front_fused_conv1a_and_stride2_convs(n, h, w, dtype, library):
    """EXPERIMENT kept in the debug library (correct, not faster: profiles/r04_front_fusion.txt).  ps_debug_conv_front_s2: conv1a + BN + ReLU + the first ResBlock's 1x1 and 3x3 stride-2 convs in one launch (conv1a's activation never
    leaves LDS) against (i) torch-CPU on identically rounded operands -- image, conv1a weights, the activation `a` and both weight sets
    rounded to the storage type, f32 accumulation, as the unfused kernels do -- and (ii) the unfused launches themselves (resnet38d.py:123,161-162,
    ResBlock.forward :28-41): image borders (conv1a's zero padding, the stride-2 conv's zero padding of `a`), ragged heights, both tile widths."""
    from pistoseg_amd import ops

    D = dev()
    q = quant(dtype)
    g = torch.Generator().manual_seed(n * 1000 + h + w)
    x = torch.randn(n, 3, h, w, generator=g)
    w1a = torch.randn(64, 3, 3, 3, generator=g) * 0.3
    sc0, sh0 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    wb1 = q(torch.randn(128, 64, 1, 1, generator=g) * 0.15)
    w2a = q(torch.randn(128, 64, 3, 3, generator=g) * 0.06)
    sc1, sh1 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.2
    a = q(F.relu(F.conv2d(q(x), q(w1a), padding=1) * sc0.view(1, -1, 1, 1) + sh0.view(1, -1, 1, 1)))
    ref_b1 = F.conv2d(a, wb1, stride=2)
    ref_2a = F.relu(F.conv2d(a, w2a, stride=2, padding=1) * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1))
    xd = x.to(D)
    wide = torch.full((n, h // 2, w // 2, 128 + 64), float("nan"), device=D, dtype=dtype)  # out_2a as a channel slice of a wider buffer
    out_2a, out_b1 = wide[..., 64:], torch.full((n, h // 2, w // 2, 128), float("nan"), device=D, dtype=dtype)
    ops.conv_front_s2(xd, w1a.to(D), sc0.to(D), sh0.to(D), w_fwd_layout(wb1).to(D, dtype), w_fwd_layout(w2a).to(D, dtype), out_b1, sc1.to(D), sh1.to(D), out_2a)
    torch.cuda.synchronize()
    assert bool(torch.isnan(wide[..., :64]).all())  # nothing outside the slice is touched
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3  # one storage rounding of the outputs + the few `a` values that round the other way
    assert rel_err(out_b1.float().cpu(), nhwc(ref_b1)) < tol and rel_err(out_2a.float().cpu(), nhwc(ref_2a)) < tol
    # the unfused launches
    ad = torch.empty((n, h, w, 64), device=D, dtype=dtype)
    ops.conv1a_fwd(xd, w1a.to(D), sc0.to(D), sh0.to(D), ad)
    u_b1, u_2a = torch.empty_like(out_b1), torch.empty((n, h // 2, w // 2, 128), device=D, dtype=dtype)
    ops.conv2d_fwd(ops.ConvSpec(64, 128, 1, 2, 1), ad, w_fwd_layout(wb1).to(D, dtype), out_raw=u_b1)
    ops.conv2d_fwd(ops.ConvSpec(64, 128, 3, 2, 1), ad, w_fwd_layout(w2a).to(D, dtype), bn_scale=sc1.to(D), bn_shift=sh1.to(D), out_act=u_2a)
    assert rel_err(out_b1.float().cpu(), u_b1.float().cpu()) < tol and rel_err(out_2a.float().cpu(), u_2a.float().cpu()) < tol
    # most elements agree bit for bit (the rest are storage-rounding neighbours: conv1a's K = 27 sum runs in a different order)
    same = float((out_2a == u_2a).float().mean())
    print(f"[front {dtype} {n}x{h}x{w}] identical to the unfused launches: {same:.4f} of a2, max rel err vs CPU {rel_err(out_2a.float().cpu(), nhwc(ref_2a)):.2e}")
    assert same > 0.9
