"""Sliding-window evaluation on the device: what `SegmentationModule.validation_step` /
`validation_epoch_end` (models/segmentation_module.py:127-185) and `segmentation_test.py:141-207` do per tile on the host in
numpy float64 -- softmax -> add into a per-(image, scale) canvas -> divide by the coverage count -> bilinear resize to the
image size -> average over scales -> argmax -> confusion matrix -- with the canvases resident in HBM (f64, as the reference's
`np.zeros` canvases) and no device->host copy per tile.

Tile names follow the reference's dataset convention `"{image_idx}_{scale}_{y}_{x}-....png"` (segmentation_module.py:146-149).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .metrics import mIoUMask

Tensor = torch.Tensor


def parse_tile_name(name: str) -> Tuple[str, float, Tuple[int, int]]:
    """(image_idx, scale, (y, x)) exactly as segmentation_module.py:146-149 parses a patch file name."""
    parts = name.split("_")
    return parts[0], float(parts[1]), (int(parts[2]), int(parts[3].split("-")[0]))


class SlidingWindowAccumulator:
    """Per-(image, scale) probability canvases + coverage counts on the device.

    image_size_fn(image_idx) -> (w, h) replaces the reference's `Image.open(<val_data>/../img/<idx>.png).size`."""

    def __init__(self, num_classes: int, device, image_size_fn: Callable[[str], Tuple[int, int]], channels_last: bool = True,
                 apply_softmax: bool = True):
        self.c, self.device, self.image_size_fn = num_classes, torch.device(device), image_size_fn
        self.channels_last, self.apply_softmax = channels_last, apply_softmax
        self.ms: Dict[str, Tuple[Tensor, Tensor]] = {}      # f"{image_idx}_{scale}" -> (sum canvas, count)   [pred_big_mask_dict_ms]
        self.sizes: Dict[str, Tuple[int, int]] = {}          # image_idx -> (w, h)
        self.full: Optional[Dict[str, Tuple[Tensor, Tensor]]] = None  # image_idx -> (sum over scales, count)  [pred_big_mask_dict]

    def _canvas(self, h: int, w: int):
        shape = (h, w, self.c) if self.channels_last else (self.c, h, w)
        return torch.zeros(shape, device=self.device, dtype=torch.float64), torch.zeros((h, w), device=self.device, dtype=torch.float64)

    def add_batch(self, scores: Tensor, names: Sequence[str], original_h: Sequence[int], original_w: Sequence[int]) -> None:
        """scores: [N, C, S, S] f32 logits (or CAM scores when apply_softmax=False) of one batch of tiles."""
        n, c, sh, sw = scores.shape
        assert c == self.c and len(names) == n
        recs = (_lib.TileDst * n)()
        for j, name in enumerate(names):
            image_idx, scale, (y0, x0) = parse_tile_name(name)
            key = f"{image_idx}_{scale}"
            if key not in self.ms:
                if image_idx not in self.sizes:
                    self.sizes[image_idx] = tuple(int(v) for v in self.image_size_fn(image_idx))
                w, h = self.sizes[image_idx]
                self.ms[key] = self._canvas(int(h * scale), int(w * scale))  # h_ = int(h * scale), w_ = int(w * scale)
            canvas, count = self.ms[key]
            ch, cw = count.shape
            vh, vw = min(int(original_h[j]), sh), min(int(original_w[j]), sw)
            if y0 < 0 or x0 < 0 or y0 + vh > ch or x0 + vw > cw:
                raise ValueError(f"tile {name!r} ({vh}x{vw} at {y0},{x0}) does not fit its {ch}x{cw} canvas")  # numpy would raise a broadcast error
            r = recs[j]
            r.canvas, r.count, r.canvas_h, r.canvas_w = canvas.data_ptr(), count.data_ptr(), ch, cw
            r.y0, r.x0, r.vh, r.vw, r.channels_last = y0, x0, vh, vw, int(self.channels_last)
        raw = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(self.device)
        ops.softmax_scatter_accum(scores.detach().float().contiguous(), raw, self.apply_softmax)
        self.full = None

    def merge_scales(self) -> Dict[str, Tuple[Tensor, Tensor]]:
        """segmentation_module.py:163-178: every (image, scale) canvas / count, resized to the image size, summed over scales."""
        if self.full is None:
            full: Dict[str, Tuple[Tensor, Tensor]] = {}
            for key, (canvas, count) in self.ms.items():
                image_idx = key.split("_")[0]
                w, h = self.sizes[image_idx]
                if image_idx not in full:
                    full[image_idx] = self._canvas(h, w)
                dst, dcnt = full[image_idx]
                ops.canvas_resize_accum(canvas, count, 1.0, dst, dcnt, self.channels_last, zero_uncovered=False, accumulate=True)
            self.full = full
        return self.full

    def predictions(self, gt: Optional[Dict[str, Tensor]] = None, bg_value: int = -1) -> Dict[str, Tensor]:
        """image_idx -> uint8 mask [h, w]: argmax of the scale-averaged probabilities; with gt and bg_value the known background is
        written back (segmentation_test.py:199-201)."""
        out = {}
        for image_idx, (canvas, count) in self.merge_scales().items():
            g = None if gt is None else gt[image_idx].to(self.device, torch.uint8).contiguous()
            out[image_idx] = ops.canvas_argmax(canvas, count, self.channels_last, g, bg_value)
        return out

    def big_mask_iou(self, gt_fn: Callable[[str], Tensor]) -> mIoUMask:
        """`big_mask_iou(torch.from_numpy(mask_pred...), mask, probs=True)` over every image (segmentation_module.py:180-185)."""
        miou = mIoUMask(num_classes=self.c)
        for image_idx, pred in self.predictions().items():
            gt = gt_fn(image_idx).to(self.device)
            ops.confusion_accum(pred.reshape(-1), gt.to(torch.int64).reshape(-1).contiguous(), miou._device_cm(self.device), miou.num_class)
        return miou


class MultiScaleCamAccumulator:
    """OEEM stage 0 (OEEM/classification/prepare_seg_inputs.py:96-138) for ONE image: per scale, overlapping
    side x side CAM crops are summed into a [C, w_, h_] f64 canvas with a coverage counter (clamped to >= 1), normalised, resized to
    the image size, averaged over the scales and reduced to 32 x 32."""

    def __init__(self, num_classes: int, image_hw: Tuple[int, int], device):
        self.c, self.hw, self.device = num_classes, (int(image_hw[0]), int(image_hw[1])), torch.device(device)
        self.ensemble = torch.zeros((num_classes,) + self.hw, device=self.device, dtype=torch.float64)
        self.n_scales = 0

    def add_scale(self, cam_crops: Tensor, positions: Sequence[Tuple[int, int]], scaled_hw: Tuple[int, int]) -> None:
        """cam_crops: [K, C, sy, sx] f32 (already interpolated to the crop size, prepare_seg_inputs.py:117);
        positions[k] = (y, x); scaled_hw = (w_, h_) of the scaled image in the reference's (first, second) axis order."""
        k, c, sy, sx = cam_crops.shape
        hs, ws = int(scaled_hw[0]), int(scaled_hw[1])
        canvas = torch.zeros((c, hs, ws), device=self.device, dtype=torch.float64)
        count = torch.zeros((hs, ws), device=self.device, dtype=torch.float64)
        recs = (_lib.TileDst * k)()
        for j, (y, x) in enumerate(positions):
            if y < 0 or x < 0 or y + sy > hs or x + sx > ws:
                raise ValueError(f"crop {j} ({sy}x{sx} at {y},{x}) does not fit the {hs}x{ws} canvas")
            r = recs[j]
            r.canvas, r.count, r.canvas_h, r.canvas_w = canvas.data_ptr(), count.data_ptr(), hs, ws
            r.y0, r.x0, r.vh, r.vw, r.channels_last = int(y), int(x), sy, sx, 0
        raw = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(self.device)
        ops.softmax_scatter_accum(cam_crops.detach().float().contiguous(), raw, apply_softmax=False)
        # norm_cam = sum_cam / max(sum_counter, 1) -> F.interpolate(..., (w, h)) ; ensemble_cam += norm_cam
        ops.canvas_resize_accum(canvas, count, 1.0, self.ensemble, None, channels_last=False, zero_uncovered=True, accumulate=True)
        self.n_scales += 1

    def result(self, out_hw: Tuple[int, int] = (32, 32)) -> Tensor:
        """ensemble_cam /= len(scales); F.interpolate(..., (32, 32)) -> [C, 32, 32] f64 (what the reference np.save()s)."""
        out = torch.empty((self.c,) + tuple(out_hw), device=self.device, dtype=torch.float64)
        ops.canvas_resize_accum(self.ensemble, None, float(self.n_scales), out, None, channels_last=False, zero_uncovered=False, accumulate=False)
        return out
