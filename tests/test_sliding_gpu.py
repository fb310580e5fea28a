"""GPU parity of the sliding-window evaluation kernels against the CPU oracle's statement-by-statement
restatement of the reference's numpy / torch-builtin code.  f64 canvases: values within 1e-12 relative (summation order of
overlapping tiles and 1-ulp softmax differences aside), mask indices bit-exact except where the oracle's own top-2 gap is
below the probability error."""
import numpy as np
import pytest
import torch

from _parity import assert_tie_excused
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def make_tiles(seed=0, c=3):
    """Two images, two scales each, overlapping 64x64 tiles on a stride-40 grid, ragged right/bottom tiles (original_h/w < 64)."""
    g = torch.Generator().manual_seed(seed)
    sizes = {"1001": (150, 117), "1002": (96, 64)}  # (w, h)
    names, oh, ow = [], [], []
    s = 64
    for idx, (w, h) in sizes.items():
        for scale in (1.0, 0.75):
            w_, h_ = int(w * scale), int(h * scale)
            ys = sorted(set(list(range(0, max(h_ - s, 0) + 1, 40)) + [max(h_ - s, 0)]))
            xs = sorted(set(list(range(0, max(w_ - s, 0) + 1, 40)) + [max(w_ - s, 0)]))
            for y in ys:
                for x in xs:
                    names.append(f"{idx}_{scale}_{y}_{x}-[1, 0, 1].png")
                    oh.append(min(s, h_ - y))
                    ow.append(min(s, w_ - x))
    logits = torch.randn(len(names), c, s, s, generator=g) * 3
    return sizes, names, oh, ow, logits


def test_sliding_window_accumulation_matches_oracle():
    from pistoseg_amd.sliding import SlidingWindowAccumulator

    c = 3
    sizes, names, oh, ow, logits = make_tiles(1, c)
    bs = 7
    batches = [(logits[i:i + bs], names[i:i + bs], oh[i:i + bs], ow[i:i + bs]) for i in range(0, len(names), bs)]
    ref_big = ref_cpu.sliding_window_big_masks(batches, sizes, c)

    acc = SlidingWindowAccumulator(c, D, lambda idx: sizes[idx])
    for lg, nm, h_, w_ in batches:
        acc.add_batch(lg.to(D), nm, h_, w_)
    full = acc.merge_scales()
    assert set(full) == set(ref_big)
    for k, (canvas, cnt) in full.items():
        ref_sum, ref_cnt = ref_big[k]
        assert np.array_equal(cnt.cpu().numpy(), ref_cnt[..., 0])
        got = canvas.cpu().numpy()
        assert np.allclose(got, ref_sum, rtol=0, atol=3e-7)  # f32 softmax (1 ulp) feeding f64 sums
    # masks: bit-exact except where the oracle's top-2 probabilities are closer than the softmax rounding error
    gt = {k: torch.randint(0, 4, (sizes[k][1], sizes[k][0]), generator=torch.Generator().manual_seed(5), dtype=torch.uint8) for k in sizes}
    ref_masks = ref_cpu.big_mask_predictions(ref_big, {k: v.numpy() for k, v in gt.items()}, bg_value=3)
    got_masks = acc.predictions(gt, bg_value=3)
    for k in ref_masks:
        diff = ref_masks[k] != got_masks[k].cpu().numpy()
        if diff.any():
            p = ref_big[k][0] / ref_big[k][1]
            top2 = np.sort(p, axis=2)[..., -2:]
            assert_tie_excused(f"big mask {k}", int(diff.sum()), diff.size, bool(((top2[..., 1] - top2[..., 0])[diff] < 1e-6).all()))
    # confusion matrix through the device-resident mIoUMask
    miou = acc.big_mask_iou(lambda idx: gt[idx])
    cm_ref = sum(ref_cpu.confusion_matrix(ref_cpu.big_mask_predictions(ref_big)[k], gt[k].numpy(), c) for k in sizes)
    assert np.abs(miou.confusion_matrix - cm_ref).sum() <= 2  # tie pixels only


def test_uncovered_pixels_and_identical_logits_are_exact():
    """Integer-valued logits with exactly representable softmax: sums are exact in f64 whatever the order, so the device canvases
    must equal the oracle's bit for bit; uncovered pixels are NaN (0/0) and argmax there is 0 (torch.argmax treats NaN as max)."""
    from pistoseg_amd.sliding import SlidingWindowAccumulator

    sizes = {"7": (100, 80)}
    names = ["7_1.0_0_0-x.png", "7_1.0_10_20-x.png"]  # leave the right / bottom uncovered
    logits = torch.zeros(2, 3, 64, 64)                  # softmax = 1/3 everywhere
    logits[1, 2] = 200.0                                # ... and exactly (0, 0, 1) for the second tile (exp(-200) == 0 in f32)
    batches = [(logits, names, [64, 64], [64, 64])]
    ref_big = ref_cpu.sliding_window_big_masks(batches, sizes, 3)
    acc = SlidingWindowAccumulator(3, D, lambda idx: sizes[idx])
    acc.add_batch(logits.to(D), names, [64, 64], [64, 64])
    canvas, cnt = acc.merge_scales()["7"]
    got, ref = canvas.cpu().numpy(), ref_big["7"][0]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(ref, nan=-1.0))
    m = acc.predictions()["7"].cpu().numpy()
    assert np.array_equal(m, ref_cpu.big_mask_predictions(ref_big)["7"])
    assert m[79, 99] == 0 and np.isnan(got[79, 99]).all()


def test_multi_scale_cam_matches_oracle():
    """OEEM stage 0 accumulation: [C, w_, h_] canvases, counter clamped to >= 1, three bilinear resizes, mean over scales."""
    from pistoseg_amd.sliding import MultiScaleCamAccumulator

    c, side = 3, 56
    w, h = 90, 120  # reference naming: orig_img.shape[:2]
    g = torch.Generator().manual_seed(3)
    crops, poss, sizes = [], [], []
    for scale in (1.0, 1.25, 0.8):
        w_, h_ = int(w * scale), int(h * scale)
        ys = sorted(set(list(range(0, w_ - side + 1, 28)) + [w_ - side]))[:-1]  # drop the last row: leaves uncovered pixels
        xs = sorted(set(list(range(0, h_ - side + 1, 28)) + [h_ - side]))
        pos = [(y, x) for y in ys for x in xs]
        crops.append(torch.randn(len(pos), c, side, side, generator=g))
        poss.append(pos)
        sizes.append((w_, h_))
    ref = ref_cpu.multi_scale_cam(crops, poss, sizes, (w, h), c, side)
    acc = MultiScaleCamAccumulator(c, (w, h), D)
    for cr, pos, sz in zip(crops, poss, sizes):
        acc.add_scale(cr.to(D), pos, sz)
    got = acc.result((32, 32)).cpu().numpy()
    assert got.shape == ref.shape == (c, 32, 32)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("s", [8, 33, 224, 70])  # (odd rotations go through 32 x 32 LDS tiles: whole and ragged tile grids)
def test_d4_views_match_torch_rot90_flip(s):
    from pistoseg_amd import ops
    from pistoseg_amd.tta import D4_VIEWS

    x = torch.randn(2, 3, s, s, generator=torch.Generator().manual_seed(s))
    xd = x.to(D)
    for hflip, k in D4_VIEWS:
        ref = torch.rot90(x.flip(3) if hflip else x, k, (2, 3))
        v = torch.empty_like(xd)
        ops.d4_view(xd, v, hflip, k, inverse=False, accumulate=False)
        assert torch.equal(v.cpu(), ref)
        back = torch.empty_like(xd)
        ops.d4_view(v, back, hflip, k, inverse=True, accumulate=False)
        assert torch.equal(back.cpu(), x)  # inverse(view(x)) == x


def test_d4_tta_wrapper_matches_oracle_definition():
    """Third-party ttach is absent: the oracle restates its public definition (parity unpinned).  With a model that is NOT
    equivariant (a fixed random conv), the merged output must match the oracle's eight-pass mean; batched == sequential."""
    from pistoseg_amd.tta import SegmentationTTAWrapper

    g = torch.Generator().manual_seed(9)
    w = torch.randn(4, 3, 3, 3, generator=g)
    x = torch.randn(2, 3, 32, 32, generator=g)

    class Toy(torch.nn.Module):
        def forward(self, t):
            return torch.nn.functional.conv2d(t, w.to(t.device), padding=1)

    ref = ref_cpu.d4_tta(lambda t: torch.nn.functional.conv2d(t, w, padding=1), x)
    for batched in (True, False):
        got = SegmentationTTAWrapper(Toy(), batched=batched)(x.to(D)).cpu()
        assert float((got - ref).abs().max()) < 1e-5 * float(ref.abs().max())


def test_d4_tta_on_the_segmentation_model():
    """End to end on the fp32 ResNet38-d seg model against the oracle's eight CPU passes (tolerance: fp32 logits 1e-4)."""
    from oracle.make_golden import make_inputs
    from pistoseg_amd.seg_model import ResNet38dSeg
    from pistoseg_amd.tta import SegmentationTTAWrapper

    c = 3
    sd = ref_cpu.make_state_dict(c, False, seed=42)
    model = ResNet38dSeg(classes=c, precision="fp32")
    model.load_state_dict(sd, strict=True)
    model = model.to(D)
    model.eval()
    x, *_ = make_inputs(1, 64, 4, 121)
    got = SegmentationTTAWrapper(model)(x.to(D)).cpu()
    with torch.no_grad():
        ref = ref_cpu.d4_tta(lambda t: ref_cpu.seg_forward(sd, t), x)
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())


def test_module_validation_step_and_epoch_end_match_oracle():
    """SegmentationModule.validation_step / validation_epoch_end (segmentation_module.py:127-214) end to end on the fp32 model:
    patch-level and big-mask mIoU against the oracle's forward + numpy accumulation."""
    import argparse

    from oracle.make_golden import make_inputs
    from pistoseg_amd.segmentation_module import SegmentationModule

    c, s = 3, 64
    args = argparse.Namespace(patch_size=s, num_classes=c, dataset="wsss4luad", model="ResNet38d", encoder="resnet38d", lr=1e-3, weight_decay=0.05,
                              tta=True, log_path="/tmp", precision="fp32", val_data="/nonexistent/val/img")
    mod = SegmentationModule(args).to(D)
    assert hasattr(mod, "tta_wrapper")  # args.tta keeps the factory, as the reference (mosaic_module.py:75-76)
    sd = ref_cpu.make_state_dict(c, False, seed=42)
    mod.model.load_state_dict(sd, strict=True)
    mod.model.eval()
    sizes = {"31": (100, 72)}
    names = ["31_1.0_0_0-[1, 1, 0].png", "31_1.0_8_36-[1, 1, 0].png", "31_0.75_0_0-[1, 1, 0].png", "31_0.75_0_11-[1, 1, 0].png"]
    oh, ow = [64, 64, 54, 54], [64, 64, 64, 64]
    x, *_ = make_inputs(4, s, 4, 131)
    g = torch.Generator().manual_seed(8)
    tile_gt = torch.randint(0, 4, (4, s, s), generator=g)
    big_gt = torch.randint(0, 4, (72, 100), generator=g, dtype=torch.uint8)
    mod.image_size_fn = lambda idx: sizes[idx]
    mod.gt_mask_fn = lambda idx: big_gt
    mod.on_validation_epoch_start()
    for i in (0, 2):  # two batches of two tiles
        mod.validation_step((x[i:i + 2].to(D), tile_gt[i:i + 2].to(D), names[i:i + 2], oh[i:i + 2], ow[i:i + 2]), i // 2)
    got = mod.validation_epoch_end()
    with torch.no_grad():
        ref_logits = ref_cpu.seg_forward(sd, x)
    cm_patch = ref_cpu.confusion_matrix(ref_cpu.logits_to_mask(ref_logits).numpy(), tile_gt.numpy().astype(np.uint8), c)
    big = ref_cpu.sliding_window_big_masks([(ref_logits, names, oh, ow)], sizes, c)
    cm_big = ref_cpu.confusion_matrix(ref_cpu.big_mask_predictions(big)["31"], big_gt.numpy(), c)
    miou_patch, fw_patch = ref_cpu.miou_from_confusion(cm_patch)
    miou_big, fw_big = ref_cpu.miou_from_confusion(cm_big)
    # a handful of near-tie pixels may flip between the device's and the CPU's fp32 logits: IoUs agree to 1e-3
    assert abs(got["validation_miou_patch_epoch"] - miou_patch) < 1e-3 and abs(got["validation_fwiou_patch_epoch"] - fw_patch) < 1e-3
    assert abs(got["validation_miou_mask_epoch"] - miou_big) < 1e-3 and abs(got["validation_fwiou_mask_epoch"] - fw_big) < 1e-3
    assert mod.sliding is None and float(mod.valid_iou.confusion_matrix.sum()) == 0  # reset for the next epoch


def test_multi_scale_cam_matches_reference_golden(golden_dir):
    """OEEM stage 0 accumulation against the fixture minted by exec'ing the reference's own statements
    (OEEM/classification/prepare_seg_inputs.py:96-138; oracle/make_golden_eval.py)."""
    import os

    from oracle.make_golden_eval import oeem_case
    from pistoseg_amd.sliding import MultiScaleCamAccumulator

    g = np.load(os.path.join(golden_dir, "oeem_ms_cam.npz"))["ensemble_cam"]
    w, h, scales, crops, poss = oeem_case(3)
    acc = MultiScaleCamAccumulator(3, (w, h), D)
    for s, cr, pos in zip(scales, crops, poss):
        acc.add_scale(cr.to(D), pos, (int(w * s), int(h * s)))
    got = acc.result((32, 32)).cpu().numpy()
    assert got.shape == g.shape and np.allclose(got, g, rtol=1e-12, atol=1e-13)
