"""`SegmentationModule` / `MosaicModule` -- mirrors of the reference's Lightning shells
(models/segmentation_module.py:53-127, models/mosaic_module.py:59-141) around the MI355X ResNet38-d model.

Same constructor (`args` Namespace with `patch_size, num_classes, dataset, model, encoder, lr, weight_decay, tta, ...`),
`forward`, `training_step(batch, idx) -> loss`, `configure_optimizers() -> ([AdamW], [ExponentialLR(0.9)])`,
`load_from_checkpoint`, and checkpoint layout (`state_dict` keys `model.*`, `hyper_parameters['args']`).  The model
plug point is `create_model(args.model, encoder_name=args.encoder, in_channels=3, classes=args.num_classes)`
(the `smp.create_model` call site, segmentation_module.py:72-81) with `--model ResNet38d`.
Losses run as fused HIP kernels attached to autograd by a one-node Function, so Lightning's `loss.backward()` works.
"""
from __future__ import annotations

import torch
from torch.optim import AdamW
from torch.optim.lr_scheduler import ExponentialLR

import os
from functools import partial

from . import ops
from .arena import ArenaAdamW
from .lightning_shim import LightningModule
from .metrics import mIoUMask
from .seg_model import ResNet38dSeg, create_model
from .sliding import SlidingWindowAccumulator
from .tta import SegmentationTTAWrapper


class _PixelLoss(torch.autograd.Function):
    """loss = kernel(logits, target); the fused kernel already produced d loss / d logits."""

    @staticmethod
    def forward(ctx, logits, target, kind, ignore_index):
        fn = ops.softmax_ce if kind == "ce" else ops.dice_loss
        loss, dlogits = fn(logits.contiguous().float(), target.contiguous(), ignore_index, want_grad=True)
        ctx.save_for_backward(dlogits)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None, None


def pixel_ce_mean(logits, target, ignore_index):
    """torch.mean(nn.CrossEntropyLoss(reduction='none'[, ignore_index])(logits, target)) -- segmentation_module.py:63-66,101-102."""
    return _PixelLoss.apply(logits, target, "ce", ignore_index)


def dice_multiclass(logits, target, ignore_index):
    """smp DiceLoss(mode='multiclass'[, ignore_index]) -- mosaic_module.py:65-68 (third-party definition, parity unpinned)."""
    return _PixelLoss.apply(logits, target, "dice", ignore_index)


class _Shell(LightningModule):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.patch_size = getattr(args, "patch_size", 224)
        self.train_iou = mIoUMask(num_classes=args.num_classes)
        self.valid_iou = mIoUMask(num_classes=args.num_classes)
        self.test_iou = mIoUMask(num_classes=args.num_classes)
        extra = {"decoder_attention_type": "scse"} if "Unet" in args.model else {}  # segmentation_module.py:72-77 (smp models only)
        self.model = create_model(args.model, encoder_name=getattr(args, "encoder", None), in_channels=3, classes=args.num_classes,
                                  precision=getattr(args, "precision", "bf16"), **extra)
        if getattr(args, "tta", False):  # mosaic_module.py:75-76 / segmentation_module.py keep a factory, applied by the scripts
            self.tta_wrapper = partial(SegmentationTTAWrapper, merge_mode="mean")
        self.save_hyperparameters()
        self.sliding = None
        # how the reference finds an image's size / ground truth: files next to args.val_data (segmentation_module.py:151,183);
        # both are overridable so that no PIL / filesystem access is needed (tests, packed datasets)
        self.image_size_fn = self._image_size_from_disk
        self.gt_mask_fn = self._gt_mask_from_disk

    # ------------------------------------------------------------------ validation (segmentation_module.py:118-214, mosaic_module.py:118-214)
    def _val_root(self):
        return "/".join(self.args.val_data.split("/")[:-1])

    def _image_size_from_disk(self, image_idx):
        from PIL import Image

        return Image.open(os.path.join(self._val_root(), "img", image_idx + ".png")).size

    def _gt_mask_from_disk(self, image_idx):
        import numpy as np
        from PIL import Image

        return torch.from_numpy(np.asarray(Image.open(os.path.join(self._val_root(), "mask", image_idx + ".png"))).copy())

    def on_validation_epoch_start(self):
        if self.args.dataset == "wsss4luad":
            dev = next(self.model.parameters()).device
            self.sliding = SlidingWindowAccumulator(self.args.num_classes, dev, lambda idx: self.image_size_fn(idx))

    def validation_step(self, batch, batch_idx):
        image_batch, mask_batch, name_batch, original_h_batch, original_w_batch = batch
        with torch.no_grad():
            output = self(image_batch)
        self.valid_iou(output, mask_batch)
        if self.args.dataset == "wsss4luad":
            if self.sliding is None:
                self.on_validation_epoch_start()
            # the reference's per-sample softmax -> .cpu().numpy() -> canvas += probs loop, as one scatter-add launch per batch
            self.sliding.add_batch(output, list(name_batch), [int(v) for v in original_h_batch], [int(v) for v in original_w_batch])

    def validation_epoch_end(self, validation_step_outputs=None):
        """Logs exactly the reference's keys (models/segmentation_module.py:200-204,218-222,244-249; mosaic_module.py likewise):
        `validation_miou_mask_epoch` is what `ModelCheckpoint(monitor=...)` watches and formats into the checkpoint file name
        (segmentation_train.py:108-117, mosaic_train.py:121-130).  Key lists pinned by tests/golden/logged_keys.json."""
        out = {}
        tissue_iou = self.valid_iou.Tissue_Intersection_over_Union()
        miou, fwiou = self.valid_iou.Mean_Intersection_over_Union(), self.valid_iou.Frequency_Weighted_Intersection_over_Union()
        if self.args.dataset == "wsss4luad":
            if self.sliding is None:  # an epoch without a validation batch: empty canvases, as the reference's empty dicts
                self.on_validation_epoch_start()
            big_mask_iou = self.sliding.big_mask_iou(lambda idx: self.gt_mask_fn(idx))
            big_tissue = big_mask_iou.Tissue_Intersection_over_Union()
            out = {
                "validation_tiou_patch_epoch": (tissue_iou[0], False), "validation_siou_patch_epoch": (tissue_iou[1], False),
                "validation_niou_patch_epoch": (tissue_iou[2], False),
                "validation_miou_patch_epoch": (miou, True), "validation_fwiou_patch_epoch": (fwiou, True),
                "validation_tiou_mask_epoch": (big_tissue[0], False), "validation_siou_mask_epoch": (big_tissue[1], False),
                "validation_niou_mask_epoch": (big_tissue[2], False),
                "validation_miou_mask_epoch": (big_mask_iou.Mean_Intersection_over_Union(), True),
                "validation_fwiou_mask_epoch": (big_mask_iou.Frequency_Weighted_Intersection_over_Union(), True),
            }
        else:  # bcss: per-tissue IoUs of the patch meter under the *_mask_epoch names (four tissues)
            out = {
                "validation_tmr_mask_epoch": (tissue_iou[0], False), "validation_str_mask_epoch": (tissue_iou[1], False),
                "validation_lym_mask_epoch": (tissue_iou[2], False), "validation_nec_mask_epoch": (tissue_iou[3], False),
                "validation_miou_mask_epoch": (miou, True), "validation_fwiou_mask_epoch": (fwiou, True),
            }
        for k, (v, bar) in out.items():
            self.log(k, v, prog_bar=bar)
        self.valid_iou.reset()
        self.sliding = None
        return {k: v for k, (v, _) in out.items()}

    def configure_optimizers(self):
        """`[AdamW(params, lr, weight_decay)], [ExponentialLR(gamma=0.9)]` (segmentation_module.py:86-90, mosaic_module.py:92-96).  For the in-tree
        ResNet38-d model the AdamW is `arena.ArenaAdamW`, a torch.optim.AdamW subclass with the same arithmetic whose `step()` is one fused launch
        over the model's flat parameter arena and whose `zero_grad()` is one memset (the autograd node writes the gradients into that arena)."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        opt_cls = ArenaAdamW if isinstance(self.model, ResNet38dSeg) else AdamW
        optimizer = opt_cls(params, self.args.lr, weight_decay=self.args.weight_decay)
        scheduler = ExponentialLR(optimizer, gamma=0.9)
        return [optimizer], [scheduler]

    def _train_miou(self):
        """What `self.log("train_miou...", self.train_iou.Mean_Intersection_over_Union())` logs, without the device->host copy when the meter has
        just been updated on the device (metrics.mIoUMask.forward keeps the values it computed): a 0-d device tensor, same f64 value."""
        vals = getattr(self.train_iou, "last_iou", None)
        return vals[0] if vals is not None else self.train_iou.Mean_Intersection_over_Union()

    def forward(self, x):
        return self.model(x)

    def training_epoch_end(self, training_step_outputs=None):
        self.log("train_miou_epoch", self.train_iou.Mean_Intersection_over_Union())
        self.log("train_fwiou_epoch", self.train_iou.Frequency_Weighted_Intersection_over_Union())
        self.train_iou.reset()


class SegmentationModule(_Shell):
    def __init__(self, args):
        super().__init__(args)
        self.retrain_iteration = 0
        self.ignore_index = 3 if args.dataset == "wsss4luad" else None  # segmentation_module.py:63-66

    def training_step(self, batch, batch_idx):
        mask_pred = self.model(batch["image"])
        loss = pixel_ce_mean(mask_pred, batch["mask"], self.ignore_index)
        self.log("train_loss", loss, prog_bar=True)
        self.train_iou(mask_pred, batch["mask"])
        self.log("train_miou", self._train_miou(), prog_bar=True)
        return loss


class MosaicModule(_Shell):
    def __init__(self, args):
        super().__init__(args)
        self.ignore_index = args.num_classes if args.dataset == "wsss4luad" else None  # mosaic_module.py:65-68
        self.parameter = None

    def training_step(self, batch, batch_idx):
        mask_pred = self(batch["image"])
        loss = dice_multiclass(mask_pred, batch["mask"], self.ignore_index)
        self.log("train_loss", loss, prog_bar=True)
        self.train_iou(mask_pred, batch["mask"])
        self.log("train_miou_epoch", self._train_miou(), prog_bar=True)
        return loss
