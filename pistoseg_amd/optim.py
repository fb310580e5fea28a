"""Optimisers of the reference, stepping through the fused HIP kernels.

`PolyOptimizer` mirrors utils.PolyOptimizer (utils.py:166-187) and `ArenaAdamW` is what `configure_optimizers()` returns in place of
`torch.optim.AdamW` (models/segmentation_module.py:86-90); both live in `pistoseg_amd/arena.py` with the flat parameter arena they step.
"""
from __future__ import annotations

from .arena import ArenaAdamW, PolyOptimizer, _dense_pair, _touched  # noqa: F401

FusedAdamW = ArenaAdamW  # earlier name
