"""mIoU bookkeeping -- mirror of the reference's `loss.mIoUMask` (loss.py:8-67) with the confusion matrix kept
on the device: `forward(logits, mask)` does (softmax ->) argmax -> uint8, a bincount-style accumulate and the IoU
formulas in HIP kernels and returns its two values WITHOUT a device->host copy (the reference copies the full
prediction to the host every training step): they are `LazyScalar`s, numbers that fetch themselves
when first formatted, compared or used in arithmetic.  The explicit queries (`Mean_Intersection_over_Union()` ...)
return host floats / arrays as in the reference; `*_device()` variants return 0-d device tensors for `self.log`."""
from __future__ import annotations

import numbers

import numpy as np
import torch

from . import _lib, ops


class LazyScalar(numbers.Real):
    """A float64 that still lives on the device: one element of a device tensor, copied to the host on first use."""

    __slots__ = ("_t", "_i", "_v")

    def __init__(self, tensor: torch.Tensor, index: int):
        self._t, self._i, self._v = tensor, index, None

    def tensor(self) -> torch.Tensor:
        """0-d device view (no synchronisation): what to hand to `self.log`."""
        return self._t[self._i]

    def __float__(self) -> float:
        if self._v is None:
            self._v = float(self._t[self._i].item())
            self._t = None
        return self._v

    def item(self) -> float:
        return float(self)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(float(self), dtype=dtype or np.float64)

    def __repr__(self):
        return repr(np.float64(float(self)))

    __str__ = lambda self: str(np.float64(float(self)))  # noqa: E731

    def __format__(self, spec):
        return format(float(self), spec)

    def __hash__(self):
        return hash(float(self))

    def __bool__(self):
        return bool(float(self))

    # numbers.Real's abstract arithmetic: all of it on the host value
    def _bin(op):  # noqa: N805
        def f(self, other):
            return op(float(self), float(other) if isinstance(other, LazyScalar) else other)

        return f

    def _rbin(op):  # noqa: N805
        def f(self, other):
            return op(other, float(self))

        return f

    import operator as _o

    __add__, __radd__ = _bin(_o.add), _rbin(_o.add)
    __sub__, __rsub__ = _bin(_o.sub), _rbin(_o.sub)
    __mul__, __rmul__ = _bin(_o.mul), _rbin(_o.mul)
    __truediv__, __rtruediv__ = _bin(_o.truediv), _rbin(_o.truediv)
    __floordiv__, __rfloordiv__ = _bin(_o.floordiv), _rbin(_o.floordiv)
    __mod__, __rmod__ = _bin(_o.mod), _rbin(_o.mod)
    __pow__, __rpow__ = _bin(_o.pow), _rbin(_o.pow)
    __lt__, __le__, __eq__ = _bin(_o.lt), _bin(_o.le), _bin(_o.eq)
    del _bin, _rbin, _o

    def __neg__(self):
        return -float(self)

    def __pos__(self):
        return float(self)

    def __abs__(self):
        return abs(float(self))

    def __trunc__(self):
        return int(float(self))

    def __floor__(self):
        return int(np.floor(float(self)))

    def __ceil__(self):
        return int(np.ceil(float(self)))

    def __round__(self, ndigits=None):
        return round(float(self), ndigits)


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

    def iou_device(self) -> torch.Tensor:
        """f64 [2 + num_class] on the device: (mIoU, fwIoU, per-tissue IoU ...) of the matrix as it stands, bit-identical to the three host
        queries above (ps_iou_from_confusion); enqueued on the current stream, no synchronisation."""
        assert self._cm is not None and self._cm.is_cuda, "nothing accumulated on a device yet"
        return ops.iou_from_confusion(self._cm, self.num_class)

    def Mean_Intersection_over_Union_device(self) -> torch.Tensor:
        """0-d device tensor, for `self.log(...)` inside a training step (the host value appears only when the logger formats it)."""
        return self.iou_device()[0]

    def forward(self, logits, mask, probs=False):
        assert self.ignore_class is None, "ignore_class is never used by the reference's call sites"
        logits = logits.detach().float().contiguous()
        pred = ops.argmax_mask(logits, mode=_lib.PS_MASK_PLAIN, softmax_first=not probs)
        ops.confusion_accum(pred, mask.to(torch.int64).contiguous(), self._device_cm(logits.device), self.num_class)
        vals = self.iou_device()
        self.last_iou = vals  # (training_step logs vals[0] without asking again)
        return LazyScalar(vals, 0), LazyScalar(vals, 1)
