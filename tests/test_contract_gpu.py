"""The Lightning-facing contract of the module shells, against fixtures minted by executing the REFERENCE's own method bodies
(oracle/make_golden_eval.py -> tests/golden/logged_keys.json, seg_eval.npz):

  * the metric keys `training_step / training_epoch_end / validation_epoch_end` log, in order, per module and dataset branch --
    `validation_miou_mask_epoch` is what `ModelCheckpoint(monitor=...)` watches (segmentation_train.py:108-117);
  * the values: CE loss and its gradient, patch-level and big-mask IoUs of the sliding-window evaluation
    (models/segmentation_module.py:127-251) on canned logits -- f64 canvases within 3e-7 absolute (f32 softmax, 1 ulp), metrics equal;
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_eval import eval_case

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def train_args(dataset, num_classes, **kw):
    """An `args` Namespace with the fields of segmentation_train.py:31-67 / mosaic_train.py:45-82 (+ log_path, set by the scripts)."""
    base = dict(model="ResNet38d", encoder="resnet38d", num_classes=num_classes, dataset=dataset, log_dir="mosaic_logs", cutmix_pseudo=False,
                cutmix_prob=0.8, patch_size=32, pseudo_mask_dir="-", mosaic_data="-", train_data="./data/training", val_data="/nonexistent/val/patches",
                test_data="./data/testing", gpus=[1], epochs=1, batch_size=5, num_workers=0, pin_memory=False, momentum=0.9, weight_decay=0.05,
                lr=5e-4, tta=False, log_path="/tmp", precision="fp32")
    base.update(kw)
    return argparse.Namespace(**base)


class Canned(torch.nn.Module):
    """Stands where `self.model` is: returns the next canned logits (the fixtures were minted on canned logits, too)."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.next = None

    def forward(self, x):
        return self.next


def shell(cls_name, dataset, c):
    import pistoseg_amd.segmentation_module as M

    mod = getattr(M, cls_name)(train_args(dataset, c)).to(D)
    mod.model = Canned().to(D)
    return mod


@pytest.mark.parametrize("cls_name", ["SegmentationModule", "MosaicModule"])
@pytest.mark.parametrize("dataset,c", [("wsss4luad", 3), ("bcss", 4)])
def test_logged_keys_and_values_match_the_reference(golden_dir, cls_name, dataset, c):
    keys = json.load(open(os.path.join(golden_dir, "logged_keys.json")))[f"{cls_name}.{dataset}"]
    g = np.load(os.path.join(golden_dir, "seg_eval.npz"))
    pre = f"{cls_name}.{dataset}."
    sizes, batches, gt = eval_case(c, ignore=dataset == "wsss4luad")
    mod = shell(cls_name, dataset, c)
    mod.image_size_fn = lambda idx: sizes[idx]
    mod.gt_mask_fn = lambda idx: torch.from_numpy(gt[idx])

    # ---- training_step / training_epoch_end: key order; CE value + gradient where the reference defines the loss itself
    lg = batches[0][0].to(D).requires_grad_(True)
    mod.model.next = lg
    mod.logged.clear()
    loss = mod.training_step({"image": None, "mask": batches[0][1].to(D), "label": None}, 0)
    assert list(mod.logged) == keys["training_step"]
    if cls_name == "SegmentationModule":
        loss.backward()
        ref = float(g[pre + "train_loss"])
        assert abs(float(loss) - ref) < 2e-6 * abs(ref)
        dref = g[pre + "train_dlogits"]
        assert np.abs(lg.grad.cpu().numpy() - dref).max() < 1e-6 * np.abs(dref).max()
        assert abs(float(mod.logged["train_miou"]) - g[pre + "train_logged"][1]) < 1e-12
    mod.logged.clear()
    mod.training_epoch_end(None)
    assert list(mod.logged) == keys["training_epoch_end"]

    # ---- a validation epoch on the canned logits
    mod.logged.clear()
    mod.on_validation_epoch_start()
    canv = None
    for i, (lgb, mk, names, oh, ow) in enumerate(batches):
        mod.model.next = lgb.to(D)
        mod.validation_step((lgb.to(D), mk.to(D), names, torch.tensor(oh), torch.tensor(ow)), i)
    if dataset == "wsss4luad":
        acc = mod.sliding
        if cls_name == "SegmentationModule":
            for key, (canvas, cnt) in acc.ms.items():  # per-(image, scale) SUM canvases: channel totals
                assert np.allclose(canvas.sum(dim=(0, 1)).cpu().numpy(), g[pre + "ms_sum_total." + key], rtol=0, atol=1e-6)
            canv = {k: (cv / ct.unsqueeze(-1)).cpu().numpy() for k, (cv, ct) in acc.merge_scales().items()}
    out = mod.validation_epoch_end(None)
    assert list(mod.logged) == keys["validation_epoch_end"]         # same keys, same order
    assert "validation_miou_mask_epoch" in mod.logged                 # ModelCheckpoint's monitor
    got = np.array([float(mod.logged[k]) for k in keys["validation_epoch_end"]])
    assert np.abs(got - g[pre + "val_logged"]).max() < 1e-12, (got, g[pre + "val_logged"])
    assert list(out) == keys["validation_epoch_end"]
    if canv is not None:
        for k, v in canv.items():
            assert np.abs(v - g[pre + "big." + k]).max() < 3e-7       # scale-averaged probabilities [h, w, C]


def test_checkpoint_name_carries_the_monitored_metric(tmp_path):
    """`filename='{epoch:02d}-{validation_miou_mask_epoch:.4f}'` (segmentation_train.py:110-111): consumers look for 'epoch='."""
    mod = shell("SegmentationModule", "bcss", 4)
    path = mod.save_checkpoint(str(tmp_path), epoch=2, metric=0.61803)
    assert os.path.basename(path) == "epoch=02-validation_miou_mask_epoch=0.6180.ckpt"
