"""mIoU bookkeeping -- mirror of the reference's `loss.mIoUMask` (loss.py:8-67) with the confusion matrix kept
on the device: `forward(logits, mask)` does (softmax ->) argmax -> uint8 and a bincount-style accumulate in HIP
kernels; the C x C matrix only crosses to the host when an IoU is queried (the reference does a device->host
copy of the full prediction every training step)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops


class mIoUMask(torch.nn.Module):
    def __init__(self, num_classes=3, ignore_class=None, eps=1e-7):
        super().__init__()
        self.eps = eps
        self.num_class = num_classes + (1 if ignore_class is not None else 0)
        self.ignore_class = ignore_class
        self._cm = None

    def _device_cm(self, device):
        if self._cm is None or self._cm.device != device:
            self._cm = torch.zeros(self.num_class * self.num_class, dtype=torch.int64, device=device)
        return self._cm

    @property
    def confusion_matrix(self) -> np.ndarray:
        if self._cm is None:
            return np.zeros((self.num_class,) * 2)
        return self._cm.cpu().numpy().reshape(self.num_class, self.num_class).astype(np.float64)

    def reset(self):
        if self._cm is not None:
            self._cm.zero_()

    def Tissue_Intersection_over_Union(self):
        cm = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
        iou[np.isnan(iou)] = 0
        return iou

    def Mean_Intersection_over_Union(self):
        return np.mean(self.Tissue_Intersection_over_Union())

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            freq = cm.sum(1) / cm.sum()
            iu = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
        return (freq[freq > 0] * iu[freq > 0]).sum()

    def forward(self, logits, mask, probs=False):
        assert self.ignore_class is None, "ignore_class is never used by the reference's call sites"
        logits = logits.detach().float().contiguous()
        pred = ops.argmax_mask(logits, mode=_lib.PS_MASK_PLAIN, softmax_first=not probs)
        ops.confusion_accum(pred, mask.to(torch.int64).contiguous(), self._device_cm(logits.device), self.num_class)
        return self.Mean_Intersection_over_Union(), self.Frequency_Weighted_Intersection_over_Union()
