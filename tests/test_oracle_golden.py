"""Pins the CPU oracle (oracle/ref_cpu.py) to golden vectors minted from the reference itself
(oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.make_golden import make_inputs, with_bg

RTOL = 2e-5  # oracle vs reference: same torch kernels, differences are summation order only


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def check_summary(g, prefix, t, rtol=RTOL):
    t = t.detach()
    assert tuple(g[f"{prefix}.shape"]) == tuple(t.shape)
    flat = t.reshape(-1).numpy()
    ref = g[f"{prefix}.val"]
    scale = max(float(np.abs(ref).max()), 1e-6)
    np.testing.assert_allclose(flat[g[f"{prefix}.idx"]], ref, rtol=0, atol=rtol * scale)
    abssum = float(g[f"{prefix}.abssum"])
    assert abs(float(np.abs(flat.astype(np.float64)).sum()) - abssum) <= 1e-5 * abssum + 1e-6


def test_state_dict_key_sets():
    assert len(ref_cpu.state_dict_spec(None, False)) == 228  # resnet38d.Net
    assert len(ref_cpu.state_dict_spec(4, True)) == 233  # revise_net.Net
    sd = ref_cpu.make_state_dict(4, True)
    n_params = sum(v.numel() for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 105_326_016
    tk = ref_cpu.trainable_keys(sd)
    assert len(tk) == 40 and sum(sd[k].numel() for k in tk) == 104_457_344


def test_backbone_forward_as_dict(golden_dir):
    g = load(golden_dir, "backbone_s32.npz")
    sd = ref_cpu.make_state_dict(None, False, seed=42)
    x, *_ = make_inputs(2, 32, 4, seed=100)
    with torch.no_grad():
        d = ref_cpu.forward_as_dict(sd, x)
    for k in ("conv3", "conv4", "conv5", "conv6"):
        check_summary(g, k, d[k])


@pytest.mark.parametrize("tag,n,s,c,seed", [("s64_c4", 2, 64, 4, 101), ("s224_c4", 1, 224, 4, 102), ("s256_c5", 1, 256, 5, 103)])
def test_revise_forward(golden_dir, tag, n, s, c, seed):
    g = load(golden_dir, f"revise_{tag}.npz")
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    x, pmask, pcam, lab = make_inputs(n, s, c, seed)
    pm, pc, label = with_bg(pmask, pcam, lab)
    with torch.no_grad():
        outs = ref_cpu.revise_forward(sd, x, pm, pc)
        masks = ref_cpu.revise_infer_masks(outs, label)
        seg = ref_cpu.seg_forward(sd, x)
    for name, t in zip(("cam", "cam_rv", "pmask_rv", "pcam_rv"), outs):
        check_summary(g, name, t)
    # the seg model of this build == outputs[0] of the RFM net
    assert torch.equal(seg, outs[0])
    # mask indices are bit-exact
    for name, m in zip(("pmask_rv_mask", "pcam_rv_mask", "cam_rv_mask"), masks):
        assert np.array_equal(g[name], m.numpy().astype(np.uint8)), name
    assert np.array_equal(g["cam_mask"], ref_cpu.logits_to_mask(outs[0]).numpy())


def test_helpers(golden_dir):
    g = load(golden_dir, "helpers.npz")
    cam = torch.from_numpy(g["in"])
    np.testing.assert_array_equal(ref_cpu.get_norm_cam_d(cam.clone()).numpy(), g["get_norm_cam_d"])
    np.testing.assert_array_equal(ref_cpu.max_norm(cam).numpy(), g["max_norm"])
    np.testing.assert_array_equal(ref_cpu.max_onehot(cam).numpy(), g["max_onehot"])
    np.testing.assert_allclose(ref_cpu.adaptive_min_pooling_loss(cam[:, 1:]).numpy(), g["adaptive_min_pooling_loss"], rtol=1e-6)
    np.testing.assert_allclose(ref_cpu.rfm(cam, torch.from_numpy(g["rfm_A"]), 7, 7).numpy(), g["rfm_out"], rtol=1e-6, atol=1e-7)
    for s in (224, 256):
        out = ref_cpu.interpolate_tensor(torch.from_numpy(g[f"interp_in_{s}"]), (32, 32)).numpy()
        np.testing.assert_array_equal(out, g[f"interp_out_{s}"])
    # K9: at S=224 the align_corners=False resize to 32x32 is exactly x[..., 3::7, 3::7]
    np.testing.assert_array_equal(g["interp_out_224"], g["interp_in_224"][..., 3::7, 3::7])


def test_mask_reduce(golden_dir):
    g = load(golden_dir, "mask_reduce.npz")
    for i in range(5):
        m, e = ref_cpu.get_mask_pred_and_entropy(torch.from_numpy(g[f"c{i}.logit"]), g[f"c{i}.tissue"], g[f"c{i}.label"].tolist())
        assert np.array_equal(np.asarray(m).astype(np.int64), g[f"c{i}.mask"])
        np.testing.assert_allclose(np.asarray(e, dtype=np.float32), g[f"c{i}.entropy"], rtol=1e-6, atol=1e-7)


def test_miou(golden_dir):
    g = load(golden_dir, "miou.npz")
    pred = ref_cpu.logits_to_mask(torch.from_numpy(g["logits"])).numpy()
    cm = ref_cpu.confusion_matrix(pred, g["gt"].astype(np.uint8), 3)
    np.testing.assert_array_equal(cm, g["cm"])
    miou, fw = ref_cpu.miou_from_confusion(cm.astype(np.float64))
    assert abs(miou - float(g["miou"])) < 1e-12 and abs(fw - float(g["fwiou"])) < 1e-12


def test_rfm_loss_and_gradients(golden_dir):
    g = load(golden_dir, "rfm_loss_grad_s64.npz")
    n, s, c = 2, 64, 4
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    tk = ref_cpu.trainable_keys(sd)
    assert sorted(tk) == sorted(g["trainable"].tolist())
    for k in tk:
        sd[k].requires_grad_(True)
    x, pmask, pcam, lab = make_inputs(n, s, c, seed=104)
    pm, pc, label = with_bg(pmask, pcam, lab)
    outs = ref_cpu.revise_forward(sd, x, pm, pc)
    loss, lcls, lrfm, lecr = ref_cpu.rfm_losses(outs, pm, pc, label, (s, s))
    for name, v in (("loss", loss), ("loss_cls", lcls), ("loss_rfm", lrfm), ("loss_ecr", lecr)):
        np.testing.assert_allclose(v.detach().numpy(), g[name], rtol=2e-5)
    loss.backward()
    has = sorted(k for k in tk if sd[k].grad is not None and sd[k].grad.abs().sum() > 0)
    assert has == sorted(g["has_grad"].tolist())
    for key in [k[5:-6] for k in g.files if k.startswith("grad.") and k.endswith(".shape")]:
        check_summary(g, f"grad.{key}", sd[key].grad, rtol=2e-4)
    assert g["param_group_sizes"].tolist() == [35, 0, 5, 0]


def test_poly_optimizer(golden_dir):
    g = load(golden_dir, "poly_optimizer.npz")
    p0 = torch.nn.Parameter(torch.from_numpy(g["p0_init"].copy()))
    p1 = torch.nn.Parameter(torch.from_numpy(g["p1_init"].copy()))
    opt = ref_cpu.PolyOptimizerOracle([{"params": [p0], "lr": 0.01, "weight_decay": 5e-4}, {"params": [p1], "lr": 0.1, "weight_decay": 0}],
                                      lr=0.01, weight_decay=5e-4, max_step=4)
    assert [opt.param_groups[0]["momentum"], opt.param_groups[0]["weight_decay"]] == g["group0"].tolist()
    for step in range(6):
        p0.grad = torch.from_numpy(g[f"g0_step{step}"].copy())
        p1.grad = torch.from_numpy(g[f"g1_step{step}"].copy())
        opt.step()
        np.testing.assert_array_equal(p0.detach().numpy(), g[f"p0_step{step}"])
        np.testing.assert_array_equal(p1.detach().numpy(), g[f"p1_step{step}"])


def test_seg_ce_mean_counts_ignored_pixels():
    torch.manual_seed(0)
    logits = torch.randn(2, 3, 8, 8)
    tgt = torch.randint(0, 4, (2, 8, 8))
    ce = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=3)(logits, tgt)
    assert torch.equal(ref_cpu.seg_ce_loss(logits, tgt, 3), ce.mean())
    assert (ce[tgt == 3] == 0).all() and ce.numel() == tgt.numel()


def test_oeem_wide_resnet_forward_cam_and_cls(golden_dir):
    """Oracle restatement of OEEM/classification/network/wide_resnet.py (b7 dilated by 2, 5632-channel heads) against the
    reference's own outputs (oracle/make_golden_oeem.py)."""
    g = np.load(os.path.join(golden_dir, "oeem_cam.npz"))
    from oracle.make_golden import make_inputs

    c, n, s, seed = 3, 2, 64, 301
    sd = ref_cpu.wide_state_dict(c, seed=42)
    x, *_ = make_inputs(n, s, 4, seed)
    with torch.no_grad():
        cam = ref_cpu.wide_forward_cam(sd, x)
        cls = ref_cpu.wide_forward_cls(sd, x)
    assert tuple(cam.shape) == tuple(g[f"cam_c{c}_s{s}.shape"])
    got = cam.reshape(-1)[torch.from_numpy(g[f"cam_c{c}_s{s}.idx"])]
    assert torch.allclose(got, torch.from_numpy(g[f"cam_c{c}_s{s}.val"]), rtol=1e-5, atol=1e-6)
    assert np.allclose(cls.numpy(), g[f"cls_c{c}_s{s}"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------------------
# Evaluation restatements vs fixtures minted by executing the reference's own method bodies (oracle/make_golden_eval.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_sliding_window_restatement_matches_reference_method_bodies(golden_dir):
    """ref_cpu.sliding_window_big_masks / big_mask_predictions / confusion_matrix against what SegmentationModule.validation_step +
    validation_epoch_end (models/segmentation_module.py:127-251) themselves produced on the same canned logits."""
    from oracle.make_golden_eval import eval_case

    g = np.load(os.path.join(golden_dir, "seg_eval.npz"))
    pre = "SegmentationModule.wsss4luad."
    sizes, batches, gt = eval_case(3)
    big = ref_cpu.sliding_window_big_masks([(lg, nm, oh, ow) for lg, _, nm, oh, ow in batches], sizes, 3)
    cm = np.zeros((3, 3))
    for k, (pred, cnt) in big.items():
        assert np.array_equal(pred / cnt, g[pre + "big." + k])  # same statements, same builtins: bit-identical
        cm += ref_cpu.confusion_matrix(ref_cpu.big_mask_predictions(big)[k], gt[k], 3)
    miou, fw = ref_cpu.miou_from_confusion(cm)
    logged = g[pre + "val_logged"]  # [t, s, n, miou, fwiou]_patch, [t, s, n, miou, fwiou]_mask
    assert abs(miou - logged[8]) < 1e-15 and abs(fw - logged[9]) < 1e-15
    # patch-level meter
    cmp_ = np.zeros((3, 3))
    for lg, mk, *_ in batches:
        cmp_ += ref_cpu.confusion_matrix(ref_cpu.logits_to_mask(lg).numpy(), mk.numpy().astype(np.uint8), 3)
    miou_p, fw_p = ref_cpu.miou_from_confusion(cmp_)
    assert abs(miou_p - logged[3]) < 1e-15 and abs(fw_p - logged[4]) < 1e-15
    # CE of training_step (mean over all pixels, ignore_index=3)
    loss = ref_cpu.seg_ce_loss(batches[0][0], batches[0][1], 3)
    assert abs(float(loss) - float(g[pre + "train_loss"])) < 1e-6


def test_multi_scale_cam_restatement_matches_reference_statements(golden_dir):
    """ref_cpu.multi_scale_cam against the exec'ed statements of OEEM/classification/prepare_seg_inputs.py:96-138."""
    from oracle.make_golden_eval import oeem_case

    g = np.load(os.path.join(golden_dir, "oeem_ms_cam.npz"))
    w, h, scales, crops, poss = oeem_case(3)
    sizes = [(int(w * s), int(h * s)) for s in scales]
    got = ref_cpu.multi_scale_cam(crops, poss, sizes, (w, h), 3, 56)
    assert np.array_equal(got, g["ensemble_cam"])


def test_logged_key_fixture_names_the_checkpoint_monitor(golden_dir):
    import json

    keys = json.load(open(os.path.join(golden_dir, "logged_keys.json")))
    for k, v in keys.items():
        assert "validation_miou_mask_epoch" in v["validation_epoch_end"], k
