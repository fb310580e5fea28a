"""Optimisers of the reference, stepping through the fused HIP kernels.

`PolyOptimizer` mirrors utils.PolyOptimizer (utils.py:166-187) including its quirk: the reference calls
`torch.optim.SGD.__init__(params, lr, weight_decay)`, so its `weight_decay` argument lands in SGD's third
positional slot -- *momentum* -- and only the per-group `weight_decay` given in the param-group dicts decays
weights.  LR follows (1 - step/max_step) ** 0.9 (the class's own `momentum` attribute is the exponent).
"""
from __future__ import annotations

import torch

from . import ops


def _touched(p: torch.Tensor) -> None:
    """The kernels rewrite parameter memory through raw pointers; tell torch (the models cache bf16 / transposed weight
    views keyed on the parameter's version counter)."""
    torch._C._increment_version(p)


def _dense_pair(p: torch.Tensor, g: torch.Tensor):
    """The kernels treat a parameter as flat memory: p must be dense and g laid out identically."""
    if g.stride() != p.stride():
        g2 = torch.empty_like(p)  # preserves p's (dense) strides
        g2.copy_(g)
        g = g2
    return p, g


class PolyOptimizer(torch.optim.Optimizer):
    def __init__(self, params, lr, weight_decay, max_step, momentum=0.9):
        defaults = dict(lr=lr, momentum=weight_decay, weight_decay=0.0)  # sic: see module docstring
        super().__init__(params, defaults)
        self.global_step = 0
        self.max_step = max_step
        self.momentum = momentum  # the poly exponent
        self._initial_lr = [group["lr"] for group in self.param_groups]

    @torch.no_grad()
    def step(self, closure=None):
        if self.global_step < self.max_step:
            lr_mult = (1 - self.global_step / self.max_step) ** self.momentum
            for g, lr0 in zip(self.param_groups, self._initial_lr):
                g["lr"] = lr0 * lr_mult
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first and group["momentum"] != 0:
                    st["momentum_buffer"] = torch.empty_like(p)
                pd, gd = _dense_pair(p.data, p.grad)
                ops.sgd_step(pd, gd, st.get("momentum_buffer"), None, group["lr"], group["momentum"], group["weight_decay"], first)
                _touched(p)
        self.global_step += 1


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (models/segmentation_module.py:86-90), one HIP launch per parameter tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                pd, gd = _dense_pair(p.data, p.grad)
                ops.adamw_step(pd, gd, st["exp_avg"], st["exp_avg_sq"], None, group["lr"], group["betas"], group["eps"], group["weight_decay"], st["step"])
                _touched(p)
