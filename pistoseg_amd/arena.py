"""Flat parameter / gradient arena of a ResNet38-d model, and the optimisers that step it in one launch.

The reference steps its models through stock torch optimisers: `configure_optimizers()` returns `[AdamW(params, lr, weight_decay)]`
(models/segmentation_module.py:86-90, models/mosaic_module.py:92-96) and stage 3 builds `PolyOptimizer(param_groups, lr, weight_decay, max_step)`
(revise_pseudo_labels.py:171-177, utils.py:166-187); Lightning / `train_epoch` then run `zero_grad(); loss.backward(); step()`.  Here the same
three calls cost three launches:

  * `ParamArena` lays the model's trainable conv weights out in ONE f32 buffer, in the order the reverse plan finalises their gradients
    (heads, b7, b6, ...: all-reduce buckets are contiguous slices), with a gradient buffer of the same layout and, for the 16-bit
    precisions, a bf16 / fp16 shadow.  The `nn.Parameter`s keep the reference's names and OIHW shapes (checkpoint layout); their storage
    is re-pointed at slices of the arena (channels-last strides = the kernels' W_fwd layout), and `p.grad` at slices of the gradient buffer;
  * the models' autograd nodes (`seg_model._SegFunction`, `revise_net._RFMFunction`) write their weight gradients straight into that
    buffer (weight gradients on a side stream, as the native trainers do) and hand autograd nothing to accumulate;
  * `ArenaAdamW` / `PolyOptimizer` are `torch.optim.Optimizer`s (schedulers, `state_dict`, `param_groups` work as usual) whose `step()` is one
    fused launch per parameter group over the arena -- it also refreshes the 16-bit shadow -- and whose `zero_grad()` is one memset.

The native trainers (`trainer.SegTrainer` / `RFMTrainer`) use the same arena.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def arena_order(model) -> List[Tuple[str, torch.nn.Parameter]]:
    """Trainable conv weights in the order their gradients become final during the reverse plan
    (fc8, b7, b6, ... ) so that all-reduce buckets are contiguous arena slices."""
    from .resnet38d import UNITS

    named = dict(model.trainable_conv_params())
    out: List[Tuple[str, torch.nn.Parameter]] = []
    units = getattr(model, "units", UNITS)
    unit_names = {u[0] for u in units}
    scratch = {id(m.weight) for m in getattr(model, "from_scratch_layers", [])}
    heads = [k for k in named if k.split(".")[0] not in unit_names]
    # heads finish first; the from-scratch ones (revise_net.py:22: 10 x lr in stage 3) lead, so that group is one contiguous slice
    for k in sorted(heads, key=lambda k: id(named[k]) not in scratch):
        out.append((k, named.pop(k)))
    for u in reversed(units):
        for k in list(named):
            if k.split(".")[0] == u[0]:
                out.append((k, named.pop(k)))
    assert not named
    return out


class ParamArena:
    """p_flat / g_flat (/ pb_flat): f32 masters, f32 gradients, 16-bit shadow of `model`'s trainable conv weights.  One per model
    (`ParamArena.of(model)`); optimiser state (moments, momentum buffers) belongs to the optimiser."""

    def __init__(self, model):
        params = list(model.parameters())
        assert params and params[0].is_cuda, "move the model to the GPU first"
        dev = params[0].device
        self.model_ref = weakref.ref(model)
        self.device = dev
        self.precision = model.precision
        self.entries = arena_order(model)
        self.names = [n for n, _ in self.entries]
        total = sum(p.numel() for _, p in self.entries)
        self.p_flat = torch.empty(total, device=dev, dtype=torch.float32)
        self.g_flat = torch.zeros(total, device=dev, dtype=torch.float32)
        # 16-bit shadow of the weights, refreshed by the fused optimisers (the split path re-derives its [hi(32) | lo(32)] blocks from the master instead)
        self.pb_flat = torch.empty(total, device=dev, dtype=model.compute_dtype) if model.precision in ("bf16", "fp16") else None
        self.grads: Dict[str, Tensor] = {}          # name -> [cout][kh][kw][cin] view of g_flat: what the weight-gradient kernels accumulate into
        self.grad_views: Dict[str, Tensor] = {}     # name -> the same memory shaped like the parameter (OIHW, channels-last strides): `p.grad`
        self.offsets: Dict[str, Tuple[int, int]] = {}
        scratch = {id(m.weight) for m in getattr(model, "from_scratch_layers", [])}
        self.packs_f9 = hasattr(model, "_unpack_w9_grad") and any(n.startswith("f9_") for n in self.names)
        off, self.n_scratch = 0, 0
        with torch.no_grad():
            for name, p in self.entries:
                cout, cin, kh, kw = p.shape
                n = p.numel()
                view = self.p_flat[off:off + n].view(cout, kh, kw, cin)
                view.copy_(p.detach().permute(0, 2, 3, 1))
                p.data = view.permute(0, 3, 1, 2)  # OIHW shape, channels-last strides, arena storage
                gv = self.g_flat[off:off + n].view(cout, kh, kw, cin)
                self.grad_views[name] = gv.permute(0, 3, 1, 2)
                if not (self.packs_f9 and name.startswith("f9_")):
                    self.grads[name] = gv
                    if self.pb_flat is not None:
                        model.register_shadow(name, p, self.pb_flat[off:off + n].view(cout, kh, kw, cin))
                self.offsets[name] = (off, n)
                off += n
                if id(p) in scratch:
                    assert off - n == self.n_scratch, "scratch heads must be contiguous at the start of the arena"
                    self.n_scratch = off
        if self.packs_f9:  # f9_1 / f9_2 run as ONE stacked 1x1 conv (revise_net.Net._w9): its packed gradient, unpacked by `heads_done`
            from .revise_net import FCAT

            self.f9_packed = torch.zeros((384, 1, 1, FCAT), device=dev, dtype=torch.float32)
            self.grads["f9"] = self.f9_packed
        else:
            self.f9_packed = None
        self.wgrad_stream: Optional[torch.cuda.Stream] = None  # side stream of the autograd path's weight gradients (made on first use)
        self.reducer = None  # dist.BucketedAllReduce of the autograd path (attach_reducer)
        self._checked_epoch = getattr(model, "_freeze_epoch", 0)
        self.sync_shadow()

    # ------------------------------------------------------------------ lookup
    @classmethod
    def of(cls, model, create: bool = True) -> Optional["ParamArena"]:
        """The model's arena; (re)built when the model has none, moved to another device, changed precision or its set of trainable
        weights changed.  create=False: None instead of building one."""
        a = getattr(model, "_param_arena", None)
        if a is not None and a.valid_for(model):
            return a
        if not create:
            return None
        a = cls(model)
        model._param_arena = a
        return a

    def valid_for(self, model) -> bool:
        if self.precision != model.precision:
            return False
        name, p = self.entries[0]
        o, _ = self.offsets[name]
        if p.data_ptr() != self.p_flat.data_ptr() + 4 * o:  # (`.to(device)` / a torch-side `p.data = ...` moves the parameter out of the arena)
            return False
        epoch = getattr(model, "_freeze_epoch", 0)  # bumped by Net.train(), the only place the reference flips requires_grad (resnet38d.py:191-213)
        if epoch != self._checked_epoch or not all(q.requires_grad for _, q in self.entries):
            cur = model.trainable_conv_params()
            if len(cur) != len(self.entries) or {n for n, _ in cur} != set(self.names):
                return False
            self._checked_epoch = epoch
        return True

    def owns(self, p: Tensor) -> Optional[Tuple[int, int]]:
        """(offset, numel) of parameter p in the arena, or None."""
        lo = self.p_flat.data_ptr()
        d = p.data_ptr() - lo
        if p.dtype != torch.float32 or d < 0 or d >= 4 * self.p_flat.numel():
            return None
        return d // 4, p.numel()

    # ------------------------------------------------------------------ weights
    def sync_shadow(self) -> None:
        """Re-derive the whole 16-bit weight arena from the f32 master in one launch (after construction, or after the masters were
        written through torch: `load_state_dict` on resume).  Per-parameter staleness is also caught lazily by `Net.w_fwd`."""
        model = self.model_ref()
        if self.pb_flat is not None:
            ops.cast_f32_lowp(self.p_flat, self.pb_flat)
            for name, p in self.entries:
                if name in model._bf16_shadow:
                    model._shadow_version[name] = p._version
        model.invalidate_weight_cache()

    def weights_stepped(self) -> None:
        """After a fused optimiser launch rewrote p_flat (and pb_flat) through raw pointers."""
        self.model_ref().invalidate_weight_cache()

    # ------------------------------------------------------------------ gradients
    def zero_grads(self) -> None:
        self.g_flat.zero_()

    def bind_param_grads(self) -> None:
        """Start of an autograd backward: every parameter's `.grad` IS its slice of g_flat.  A parameter whose `.grad` is None (first step,
        `zero_grad(set_to_none=True)`, `model.zero_grad()`) gets its slice zeroed and bound; one that already holds its slice keeps what is in
        it (autograd's accumulate-into-.grad semantics: the kernels add); a foreign tensor assigned by the caller is copied in."""
        fresh = []
        for name, p in self.entries:
            g = p.grad
            view = self.grad_views[name]
            if g is None:
                fresh.append(name)
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
            else:
                continue
            p.grad = view
        if len(fresh) == len(self.entries):
            self.g_flat.zero_()
        else:
            for name in fresh:
                o, n = self.offsets[name]
                self.g_flat[o:o + n].zero_()
        if self.f9_packed is not None:
            self.f9_packed.zero_()

    def heads_done(self) -> None:
        """RFM net: the packed q | k weight gradient is final -> add it into f9_1 / f9_2's slots (they hold zeros, or what earlier backwards left)."""
        if self.f9_packed is None:
            return
        from .revise_net import FCAT

        g1, g2 = self.model_ref()._unpack_w9_grad(self.f9_packed.view(384, FCAT))
        for nm, g in (("f9_1.weight", g1), ("f9_2.weight", g2)):
            if nm in self.offsets:
                self.grad_views[nm].add_(g)

    def side_stream(self) -> torch.cuda.Stream:
        if self.wgrad_stream is None:
            self.wgrad_stream = torch.cuda.Stream(device=self.device)
        return self.wgrad_stream

    def attach_reducer(self, process_group=None, bucket_mb: float = 48.0, grad_payload: str = "fp32", share: str = "reserve+queue",
                       reserved_cus: Optional[int] = None):
        """N > 1 through the reference's own training loop (one process per GPU, no torch DDP wrapper): the autograd nodes all-reduce the
        arena in buckets behind their reverse plan, exactly as the native trainers do (dist.BucketedAllReduce).  The loss must then be the
        local mean: the node divides the incoming gradient by the world size."""
        from .dist import BucketedAllReduce, default_reserved_cus, plan_buckets

        buckets = plan_buckets([(name, p.numel()) for name, p in self.entries], int(bucket_mb * (1 << 20) / 4))
        self.reducer = BucketedAllReduce(self.g_flat, buckets, process_group, launch_opts=self.model_ref().launch, payload=grad_payload, share=share,
                                         reserved_cus=default_reserved_cus() if reserved_cus is None else reserved_cus)
        return self.reducer

    # ------------------------------------------------------------------ ranges
    def ranges_of(self, params) -> Tuple[List[Tuple[int, int]], List[torch.nn.Parameter]]:
        """Merged contiguous [lo, hi) arena ranges covering those of `params` that live in the arena, and the ones that do not."""
        spans, outside = [], []
        for p in params:
            hit = self.owns(p)
            if hit is None:
                outside.append(p)
            else:
                spans.append((hit[0], hit[0] + hit[1]))
        spans.sort()
        merged: List[Tuple[int, int]] = []
        for lo, hi in spans:
            if merged and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        return merged, outside


def model_of(params) -> Optional[object]:
    """The pistoseg_amd model that owns these parameters (every `resnet38d.Net` tags its parameters in `train()`)."""
    from .resnet38d import owner_of

    for p in params:
        m = owner_of(p)
        if m is not None:
            return m
    return None


class _ArenaOptMixin:
    """Shared plumbing, mixed into a torch optimiser class: finds the owning model's arena, maps param groups to arena ranges, keeps
    per-parameter state entries as views of flat state buffers (so `state_dict()` / `load_state_dict()` have torch's layout), zeroes
    gradients with one memset."""

    STATE_BUFFERS: Tuple[str, ...] = ()

    def _arena_init(self) -> None:
        self._arena: Optional[ParamArena] = None
        self._flat_state: Dict[str, Tensor] = {}
        self._plan = None

    # -- arena -------------------------------------------------------------------------------------
    def arena(self) -> Optional[ParamArena]:
        """The arena of the model these parameters belong to (None: not a pistoseg_amd model, or not on the GPU yet -- the optimiser then
        steps parameter by parameter)."""
        a = self._arena
        model = a.model_ref() if a is not None else None
        if a is not None and model is not None and a.valid_for(model):
            return a
        allp = [p for g in self.param_groups for p in g["params"]]
        model = model_of(allp)
        if model is None or not allp[0].is_cuda:
            return None
        old = self._flat_state
        a = ParamArena.of(model)
        self._arena, self._plan = a, None
        self._flat_state = {k: torch.zeros_like(a.p_flat) for k in self.STATE_BUFFERS}
        for p in allp:  # carry over state made before (per-parameter steps, a loaded state_dict)
            hit = a.owns(p)
            st = self.state.get(p)
            if hit is None or not st:
                continue
            for k in self.STATE_BUFFERS:
                if k in st and torch.is_tensor(st[k]):
                    self._state_view(k, p, hit).copy_(st[k])
        del old
        self._bind_state()
        return a

    def _state_view(self, key: str, p: Tensor, hit: Tuple[int, int]) -> Tensor:
        o, n = hit
        cout, cin, kh, kw = p.shape
        return self._flat_state[key][o:o + n].view(cout, kh, kw, cin).permute(0, 3, 1, 2)

    def _bind_state(self) -> None:
        a = self._arena
        for g in self.param_groups:
            for p in g["params"]:
                hit = a.owns(p)
                if hit is None:
                    continue
                st = self.state[p]
                for k in self.STATE_BUFFERS:
                    if k in st or self._state_is_live():
                        st[k] = self._state_view(k, p, hit)

    def _state_is_live(self) -> bool:
        return False

    @staticmethod
    def _has_grads(group) -> bool:
        """torch's optimisers skip parameters whose `.grad` is None; the fused launch covers a group's whole arena range, so it runs when ANY of the
        group's parameters holds a gradient (the autograd nodes bind all of them at once) and is skipped when none does (a `step()` before any backward)."""
        return any(p.grad is not None for p in group["params"])

    def group_plan(self):
        """[(group, [(lo, hi)], [parameters outside the arena])] -- recomputed when the groups change."""
        a = self.arena()
        sig = tuple(tuple(id(p) for p in g["params"]) for g in self.param_groups)
        if self._plan is None or self._plan[0] != sig:
            plan = []
            for g in self.param_groups:
                if a is None:
                    plan.append((g, [], list(g["params"])))
                else:
                    ranges, outside = a.ranges_of(g["params"])
                    plan.append((g, ranges, outside))
            self._plan = (sig, plan)
        return self._plan[1]

    # -- torch.optim.Optimizer API -------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        """One memset over the gradient arena; `p.grad` stays bound to its slice (the next backward adds into zeros).  Parameters outside
        the arena are handled as torch does."""
        a = self.arena()
        if a is None:
            return super().zero_grad(set_to_none=set_to_none)
        a.zero_grads()
        for _, _, outside in self.group_plan():
            for p in outside:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_().zero_()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)  # (replaces the per-parameter state tensors by copies)
        a = self.arena()
        if a is None:
            return
        for g in self.param_groups:
            for p in g["params"]:
                hit = a.owns(p)
                st = self.state.get(p)
                if hit is None or not st:
                    continue
                for k in self.STATE_BUFFERS:
                    if k in st and torch.is_tensor(st[k]):
                        view = self._state_view(k, p, hit)
                        if st[k].data_ptr() != view.data_ptr():
                            view.copy_(st[k])
                        st[k] = view
        self._after_load()

    def _after_load(self) -> None:
        pass


def _dense_pair(p: Tensor, g: Tensor):
    """The kernels treat a parameter as flat memory: p must be dense and g laid out identically."""
    if g.stride() != p.stride():
        g2 = torch.empty_like(p)  # preserves p's (dense) strides
        g2.copy_(g)
        g = g2
    return p, g


def _touched(p: Tensor) -> None:
    """The kernels rewrite parameter memory through raw pointers; tell torch (the models cache bf16 / transposed weight
    views keyed on the parameter's version counter)."""
    torch._C._increment_version(p)


class ArenaAdamW(_ArenaOptMixin, torch.optim.AdamW):
    """`torch.optim.AdamW(params, lr, weight_decay=...)` (models/segmentation_module.py:86-90) as ONE fused launch per parameter group over
    the model's arena (`ps_adamw_step`: decoupled weight decay, bias correction, 16-bit shadow refresh).  A subclass of torch's AdamW: same
    constructor, `param_groups`, `state` keys (`step`, `exp_avg`, `exp_avg_sq`) and `state_dict()` layout; parameters that do not live in a
    pistoseg_amd arena are stepped one launch each.  (amsgrad / maximize are not implemented by the kernel and are refused.)"""

    STATE_BUFFERS = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, maximize=False):
        if amsgrad or maximize:
            raise ValueError("ArenaAdamW: amsgrad / maximize are not implemented by ps_adamw_step")
        torch.optim.AdamW.__init__(self, params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._arena_init()
        self._steps_applied: Dict[int, int] = {}  # group index -> fused steps (all arena parameters of a group share it)

    def _state_is_live(self) -> bool:
        return True  # moments exist (as zeros) from the start, like the arena itself

    def _bind_state(self) -> None:
        super()._bind_state()
        for g in self.param_groups:
            for p in g["params"]:
                if "exp_avg" in self.state[p]:
                    self.state[p].setdefault("step", 0)

    def _after_load(self) -> None:
        self._steps_applied = {}
        for gi, g in enumerate(self.param_groups):
            steps = [int(self.state[p]["step"]) for p in g["params"] if p in self.state and "step" in self.state[p]]
            if steps:
                self._steps_applied[gi] = max(steps)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a = self.arena()
        for gi, (group, ranges, outside) in enumerate(self.group_plan()):
            lr, betas, eps, wd = float(group["lr"]), group["betas"], group["eps"], group["weight_decay"]
            if ranges and self._has_grads(group):
                t = self._steps_applied.get(gi, 0) + 1
                self._steps_applied[gi] = t
                m, v = self._flat_state["exp_avg"], self._flat_state["exp_avg_sq"]
                for lo, hi in ranges:
                    ops.adamw_step(a.p_flat[lo:hi], a.g_flat[lo:hi], m[lo:hi], v[lo:hi], None if a.pb_flat is None else a.pb_flat[lo:hi],
                                   lr, betas, eps, wd, t)
                for p in group["params"]:
                    st = self.state[p]
                    if "exp_avg" in st:
                        st["step"] = t  # (python int: torch's AdamW accepts and saves either)
            for p in outside:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1
                pd, gd = _dense_pair(p.data, p.grad)
                ops.adamw_step(pd, gd, st["exp_avg"], st["exp_avg_sq"], None, lr, betas, eps, wd, st["step"])
                _touched(p)
        if a is not None:
            a.weights_stepped()
        return loss


class PolyOptimizer(_ArenaOptMixin, torch.optim.SGD):
    """Mirror of utils.PolyOptimizer (utils.py:166-187), a `torch.optim.SGD` subclass as there, including its quirk: the reference calls
    `torch.optim.SGD.__init__(params, lr, weight_decay)`, so its `weight_decay` argument lands in SGD's third positional slot --
    *momentum* -- and only the per-group `weight_decay` given in the param-group dicts decays weights.  LR follows
    (1 - step/max_step) ** 0.9 (the class's own `momentum` attribute is the exponent).  One fused SGD launch per parameter group over
    the model's arena (stage 3: the scratch heads at 10 x lr, the backbone at lr: two launches)."""

    STATE_BUFFERS = ("momentum_buffer",)

    def __init__(self, params, lr, weight_decay, max_step, momentum=0.9):
        torch.optim.SGD.__init__(self, params, lr, weight_decay)  # sic: SGD's third positional is `momentum` -- see the class docstring
        self._arena_init()
        self.global_step = 0
        self.max_step = max_step
        self.momentum = momentum  # the poly exponent
        self._initial_lr = [group["lr"] for group in self.param_groups]
        self._arena_first = True  # the fused launches have not initialised the momentum buffers yet (torch SGD: buf = grad on the first step)

    def _after_load(self) -> None:
        self._arena_first = not any("momentum_buffer" in st for st in self.state.values())

    @torch.no_grad()
    def step(self, closure=None):
        if self.global_step < self.max_step:
            lr_mult = (1 - self.global_step / self.max_step) ** self.momentum
            for g, lr0 in zip(self.param_groups, self._initial_lr):
                g["lr"] = lr0 * lr_mult
        a = self.arena()
        stepped = False
        for group, ranges, outside in self.group_plan():
            lr, mom, wd = float(group["lr"]), group["momentum"], group["weight_decay"]
            if ranges and self._has_grads(group):
                buf = self._flat_state["momentum_buffer"]
                for lo, hi in ranges:
                    ops.sgd_step(a.p_flat[lo:hi], a.g_flat[lo:hi], buf[lo:hi] if mom != 0 else None, None if a.pb_flat is None else a.pb_flat[lo:hi],
                                 lr, mom, wd, self._arena_first)
                stepped = True
                if mom != 0 and self._arena_first:
                    for p in group["params"]:
                        hit = a.owns(p)
                        if hit is not None:
                            self.state[p]["momentum_buffer"] = self._state_view("momentum_buffer", p, hit)
            for p in outside:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first and mom != 0:
                    st["momentum_buffer"] = torch.empty_like(p)
                pd, gd = _dense_pair(p.data, p.grad)
                ops.sgd_step(pd, gd, st.get("momentum_buffer"), None, lr, mom, wd, first)
                _touched(p)
        if stepped:
            self._arena_first = False
            a.weights_stepped()
        self.global_step += 1
