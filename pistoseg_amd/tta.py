"""d4 test-time augmentation: mirror of `tta.SegmentationTTAWrapper(model, tta.aliases.d4_transform(),
merge_mode='mean')` (infer_pseudo_masks.py:96, mosaic_module.py:76).

ttach==0.0.3 is a third-party dependency that is not vendored in the reference (environment.yaml:204): its public definition is
restated here -- d4_transform = Compose([HorizontalFlip(), Rotate90([0, 90, 180, 270])]), i.e. eight views in
itertools.product order (flip slowest); a view is flip-then-rotate, its inverse on the output rotate-back-then-flip; the
'mean' merger sums the de-augmented outputs in view order and divides by 8 -- and is therefore *parity unpinned*.

The eight views are pure index permutations (HIP kernel `ps_d4_view`); they are batched into ONE forward of 8N tiles, which
is per-sample identical to eight forwards because every BatchNorm of the net is frozen in eval mode.
"""
from __future__ import annotations

import itertools

import torch

from . import ops

D4_VIEWS = list(itertools.product([False, True], [0, 1, 2, 3]))  # (hflip, k): ttach's product order


class SegmentationTTAWrapper(torch.nn.Module):
    def __init__(self, model: torch.nn.Module, merge_mode: str = "mean", batched: bool = True):
        super().__init__()
        assert merge_mode == "mean", "the reference only uses merge_mode='mean'"
        self.model, self.batched = model, batched

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        x = image.contiguous().float()
        n = x.shape[0]
        stacked = torch.empty((len(D4_VIEWS) * n,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)  # the views land in ONE batch: no torch.cat
        views = [stacked[i * n:(i + 1) * n] for i in range(len(D4_VIEWS))]
        for (hflip, k), v in zip(D4_VIEWS, views):
            ops.d4_view(x, v, hflip, k, inverse=False, accumulate=False)
        if self.batched:
            outs = self.model(stacked).float().contiguous()
            outs = [outs[i * n:(i + 1) * n] for i in range(len(D4_VIEWS))]
        else:
            outs = [self.model(v).float().contiguous() for v in views]
        merged = torch.empty_like(outs[0])
        for i, ((hflip, k), o) in enumerate(zip(D4_VIEWS, outs)):
            ops.d4_view(o.contiguous(), merged, hflip, k, inverse=True, accumulate=i > 0)
        ops.scale_inplace_(merged, float(len(D4_VIEWS)))
        return merged
