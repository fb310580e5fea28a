"""Native training step for the ResNet38-d segmentation model (stage 5, `segmentation_train.py`).

What `pl.Trainer.fit` does per batch in the reference -- `SegmentationModule.training_step`
(models/segmentation_module.py:96-111: forward, per-pixel CE, mean over all pixels, mIoU bookkeeping), then
Lightning's backward and `AdamW.step` (:86-90) -- is one call here, with no autograd graph:

  forward plan -> fused CE fwd+bwd -> reverse plan (wgrad straight into a flat f32 gradient arena)
  -> [N > 1: bucketed RCCL all-reduce of arena slices on a side stream, overlapped with the rest of the
     backward] -> one fused AdamW launch over the flat parameter arena (also refreshes the bf16 weights).

Parameters stay `nn.Parameter`s with the reference's names and OIHW shapes (checkpoint layout); their
storage is re-pointed at slices of the arena (channels-last strides = the kernels' W_fwd layout).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from .arena import ParamArena, arena_order  # noqa: F401  (arena_order: re-exported, earlier rounds imported it from here)
from .dist import BucketedAllReduce, default_reserved_cus, plan_buckets
from .seg_model import ResNet38dSeg

Tensor = torch.Tensor


def init_weights_he(model: torch.nn.Module, seed: int = 42) -> None:
    """Random-init weights of the architecture (no checkpoint is available offline): He-normal convs with a
    damped last conv per residual branch, mildly randomised BatchNorm statistics."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 4:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                std = math.sqrt(2.0 / fan_in)
                if name.endswith(("conv_branch2b1.weight", "conv_branch2b2.weight")) and not (
                    name.startswith(("b6", "b7")) and "2b1" in name):
                    std *= 0.5
                if name.startswith("fc8"):
                    std = math.sqrt(1.0 / fan_in)
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.8 + 0.6)
            elif name.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.4 - 0.2)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.6 - 0.3)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.6)


class _ArenaMixin:
    """The trainers keep their parameters, gradients and 16-bit shadow in the model's `arena.ParamArena` (shared with the reference-API
    optimisers `arena.ArenaAdamW` / `arena.PolyOptimizer`); the attributes below are views of it."""

    def _use_arena(self, model) -> None:
        self.arena = ParamArena.of(model)
        a = self.arena
        self.entries, self.offsets, self.grads = a.entries, a.offsets, a.grads
        self.p_flat, self.g_flat, self.pb_flat = a.p_flat, a.g_flat, a.pb_flat
        self.n_scratch, self.f9_packed = a.n_scratch, a.f9_packed

    # -- dynamic loss scaling without a host round trip per step (fp16 / fp16x3) ----------------------------------------------------------
    # The overflow check gates the optimiser ON THE DEVICE (ps_adamw_step_guarded / ps_sgd_step_guarded: opt_state = [steps applied, non-finite
    # elements of this step]); the host learns each step's flag through an asynchronous 4-byte copy and adapts the scale `scale_lag` steps later
    # (`settle()` brings the bookkeeping up to date; it is exact once called).  Policy: torch.cuda.amp.GradScaler's (x 0.5 and skip on overflow,
    # x 2 every 200 clean steps).
    def _init_dynamic_scale(self, dev, default_scale: float, loss_scale: Optional[float]) -> None:
        fp16 = self.model.precision in ("fp16", "fp16x3")
        self.dynamic_scale = fp16 and loss_scale is None
        self.loss_scale = float(loss_scale) if loss_scale is not None else (default_scale if fp16 else 1.0)
        self.clean_steps, self.skipped_steps = 0, 0
        self.opt_state = torch.zeros(2, device=dev, dtype=torch.int32)
        self.scale_lag = 2
        self._pending_flags = []  # (event, pinned flag, loss scale the step ran with), oldest first
        # (pinned slots and events are made once: allocating pinned memory synchronises the device)
        self._flag_slots = [torch.empty(1, dtype=torch.int32, pin_memory=True) for _ in range(self.scale_lag + 2)] if self.dynamic_scale else []
        self._flag_events = [torch.cuda.Event() for _ in self._flag_slots]
        self._flag_next = 0

    def _count_nonfinite(self) -> None:
        """opt_state[1] <- non-finite elements of the gradient arena (after the all-reduce: every rank sees the same count and takes the same decisions)."""
        flag = self.opt_state[1:]
        flag.zero_()
        ops.nonfinite_count(self.g_flat, out=flag)

    def _enqueue_flag(self) -> None:
        self.settle(keep=len(self._flag_slots) - 1)  # (a free slot in the ring)
        host, ev = self._flag_slots[self._flag_next], self._flag_events[self._flag_next]
        self._flag_next = (self._flag_next + 1) % len(self._flag_slots)
        host.copy_(self.opt_state[1:], non_blocking=True)
        ev.record()
        self._pending_flags.append((ev, host, self.loss_scale))
        self.settle(keep=self.scale_lag)

    def settle(self, keep: int = 0) -> None:
        """Bring the dynamic-loss-scale bookkeeping (`loss_scale`, `skipped_steps`, the applied-step count, `clean_steps`) up to date with all
        but the newest `keep` steps: waits for their overflow flags.  A flag only ever lowers the scale below the scale ITS step ran with, so two
        overflowing steps enqueued with the same scale halve it once."""
        while len(self._pending_flags) > keep:
            ev, host, used = self._pending_flags.pop(0)
            ev.synchronize()
            if int(host[0]) > 0:
                self.loss_scale = max(min(self.loss_scale, used * 0.5), 1.0)
                self.clean_steps = 0
                self.skipped_steps += 1
            else:
                self._step_applied()
                self.clean_steps += 1
                if self.clean_steps >= 200:
                    self.loss_scale, self.clean_steps = min(self.loss_scale * 2.0, 2.0 ** 24), 0

    # -- optimiser on a side stream (defer_optimizer=True) ------------------------------------------------------------------------------------
    # The update -- one HBM-bound launch over the arenas (3.1 GB at bs-independent 0.5 ms) plus the zero fill of the gradient arena -- depends on
    # the whole backward, but nothing needs its result before the next step's first TRAINABLE layer: conv1a and the frozen b2 units
    # (models/resnet38d.py:191-205; 1.7 ms of a bs = 64 forward) run beside it.  The launch goes to `opt_stream`; the model's forward waits for its
    # event in front of unit b3 (`Net.wait_weights`), `state_dict()` and `load_state_dict()` wait too.  Code that reads parameters or arenas through
    # torch right after `train_step` must call `wait_weights()` first -- which is why the trainers default to defer_optimizer=False.
    def _init_deferred(self, defer: bool, dev) -> None:
        self.defer_optimizer = bool(defer)
        self.opt_stream = torch.cuda.Stream(device=dev) if defer else None
        self._zeroed_by_optimizer = False
        if defer:
            model = self.model
            model.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m.wait_weights() if prefix == "" else None)

    def _optimizer_scope(self):
        """Context in which the optimiser launches of this step run: the side stream (behind everything the launch stream holds), or nothing."""
        import contextlib

        if self.opt_stream is None:
            return contextlib.nullcontext()
        self.opt_stream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.opt_stream)

    def _optimizer_done(self) -> None:
        """Inside _optimizer_scope, behind the update: zero the gradient arena for the next step there too and leave the event for the forward."""
        if self.opt_stream is None:
            return
        self.g_flat.zero_()
        if self.f9_packed is not None:
            self.f9_packed.zero_()
        self._zeroed_by_optimizer = True
        ev = torch.cuda.Event()
        ev.record()
        self.model._weights_event = ev

    def _zero_grads(self) -> None:
        """Start of a step: the gradient arena must be zero (the weight gradients accumulate)."""
        if self._zeroed_by_optimizer:  # the deferred optimiser of the previous step did it, ordered in front of this step's weight gradients by the forward's wait
            self._zeroed_by_optimizer = False
            return
        self.g_flat.zero_()
        if self.f9_packed is not None:
            self.f9_packed.zero_()

    def wait_weights(self) -> None:
        self.model.wait_weights()

    def sync_shadow(self) -> None:
        """Re-derive the whole 16-bit weight arena from the f32 master in one launch (after construction, or after the masters were
        written through torch: `load_state_dict` on resume).  Per-parameter staleness is also caught lazily by `Net.w_fwd`."""
        self.arena.sync_shadow()

    def load_state_dict(self, state_dict, strict: bool = True):
        """Checkpoint resume through the trainer: the parameters live in the f32 arena (their `.data` are views of it), so the
        module's own load writes the masters in place; then the 16-bit shadows and every derived weight layout are refreshed."""
        self.model.wait_weights()
        out = self.model.load_state_dict(state_dict, strict=strict)
        self.sync_shadow()
        return out


class SegTrainer(_ArenaMixin):
    def __init__(self, model: ResNet38dSeg, lr: float = 1e-3, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8,
                 ignore_index: Optional[int] = 3, process_group=None, bucket_mb: float = 48.0, track_iou: bool = True,
                 loss_scale: Optional[float] = None, overlap_wgrad: bool = True, deterministic: Optional[bool] = None, grad_payload: str = "fp32",
                 share: str = "reserve+queue", reserved_cus: Optional[int] = None, defer_optimizer: bool = False):
        """defer_optimizer: the AdamW launch (and the next step's gradient zero fill) on a side stream, overlapped with the next forward's frozen
        layers (see _ArenaMixin._init_deferred).  grad_payload: what the N > 1 gradient exchange puts on the wire -- "fp32" (SUM all-reduce of the f32 arena slices) or "bf16" (each
        bucket cast to bf16, all-reduced, widened back: half the xGMI bytes; see dist.BucketedAllReduce).  share / reserved_cus: how this model's
        conv launches make room for the collectives while buckets are in flight ("batch": tiles_per_block = 1; "reserve": cus_reserved; "queue": tile_queue;
        "reserve+queue", the default: both -- beside a kernel that holds 16 / 48 CUs a step costs + 9 % / + 25 % with it, + 14 % / + 29 % in batch mode, + 8 % /
        + 50 % under the bare reservation: profiles/r04_hog_step_probe_queue.txt)."""
        assert next(model.parameters()).is_cuda, "move the model to the GPU first"
        self.model = model
        # `pl.Trainer(deterministic=True)` (segmentation_train.py:153-160): weight gradients without atomics (ps_conv2d_wgrad_det), so two
        # identical runs are bit-identical.  None = follow torch.are_deterministic_algorithms_enabled(), the switch that call site sets.
        # Per-trainer state: it lives in this model's launch options (ops.LaunchOpts), not in a process-wide switch.
        self.deterministic = deterministic
        model.launch.deterministic = deterministic
        # weight gradients on a second stream (see Net.backward_backbone)
        self.wgrad_stream = torch.cuda.Stream(device=next(model.parameters()).device) if overlap_wgrad else None
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.ignore_index = ignore_index
        self.step_count = 0
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.track_iou = track_iou
        dev = next(model.parameters()).device
        self.device = dev
        model.train()
        self._use_arena(model)
        self._init_deferred(defer_optimizer, dev)
        self.m_flat = torch.zeros_like(self.p_flat)
        self.v_flat = torch.zeros_like(self.p_flat)
        # fp16 activations gradients underflow without a loss scale (a CE gradient is ~1/(N*H*W) = 3e-7 per pixel)
        self._init_dynamic_scale(dev, 65536.0, loss_scale)
        # all-reduce buckets: (unit after which the bucket is final, start, end) over the arena
        self.reducer: Optional[BucketedAllReduce] = None
        if self.world > 1:
            buckets = plan_buckets([(name, p.numel()) for name, p in self.entries], int(bucket_mb * (1 << 20) / 4))
            self.reducer = BucketedAllReduce(self.g_flat, buckets, process_group, launch_opts=model.launch, payload=grad_payload, share=share,
                                             reserved_cus=default_reserved_cus() if reserved_cus is None else reserved_cus)
        self.cm = torch.zeros(model.classes * model.classes, device=dev, dtype=torch.int64)  # train_iou confusion

    # ------------------------------------------------------------------
    def train_step(self, image: Tensor, mask: Tensor) -> Tensor:
        """One optimisation step on a batch {'image': [N,3,H,W] f32, 'mask': [N,H,W] int64}; returns the loss (1-element device tensor)."""
        model = self.model
        n = image.shape[0]
        self._zero_grads()
        drop = model.sample_dropout(n, image.device)
        feats, saved = model.run_backbone(image, save=True, drop=drop)
        logits, _ = model.head_forward(feats["conv6"], drop.get("dropout7"), image.shape[-2:])
        # mean over the GLOBAL batch: every rank scales its gradient by 1/world, the all-reduce sums
        loss, dlogits = ops.softmax_ce(logits, mask, self.ignore_index, want_grad=True, grad_scale=self.loss_scale / self.world)
        if self.track_iou:  # self.train_iou(mask_pred, mask), segmentation_module.py:108 -- kept on the device
            pred = ops.argmax_mask(logits, mode=_lib.PS_MASK_PLAIN, softmax_first=True)
            ops.confusion_accum(pred, mask, self.cm, model.classes)
        dw8 = self.grads["fc8.weight"].view(model.classes, 4096)
        g_x7 = model.head_backward(saved.conv6, drop.get("dropout7"), dlogits, dw8)
        if self.reducer is not None:
            self.reducer.begin_step()
            self.reducer.on_unit_done("fc8")
        model.backward_backbone(saved, g_x7, self.grads, after_unit=self.reducer.on_unit_done if self.reducer is not None else None,
                                wgrad_stream=self.wgrad_stream)
        if self.reducer is not None:
            self.reducer.finish()
        with self._optimizer_scope():
            if self.dynamic_scale:
                self._count_nonfinite()
                ops.adamw_step_guarded(self.p_flat, self.g_flat, self.m_flat, self.v_flat, self.pb_flat, self.lr, self.betas, self.eps,
                                       self.weight_decay, self.opt_state, grad_inv_scale=1.0 / self.loss_scale)
                self._enqueue_flag()
            else:
                self.step_count += 1
                ops.adamw_step(self.p_flat, self.g_flat, self.m_flat, self.v_flat, self.pb_flat, self.lr, self.betas, self.eps,
                               self.weight_decay, self.step_count, grad_inv_scale=1.0 / self.loss_scale)
            self._optimizer_done()
        model.invalidate_weight_cache()
        return loss

    def _step_applied(self) -> None:
        self.step_count += 1

    def lr_scheduler_step(self, gamma: float = 0.9) -> None:
        """ExponentialLR(gamma=0.9) once per epoch (segmentation_module.py:88)."""
        self.lr *= gamma


class RFMTrainer(_ArenaMixin):
    """Native stage-3 step (`revise_pseudo_labels.py:232-301`): RFM forward plan -> fused cls/rfm/ecr losses with their
    gradients (rfm_loss.py) -> reverse plan into a flat f32 gradient arena -> [N > 1: bucketed RCCL all-reduce overlapped
    with the backward] -> utils.PolyOptimizer's update (SGD, momentum = the reference's mis-placed weight_decay argument,
    per-group L2 decay, poly LR) as two fused launches over the arena: scratch heads (10 x lr) and pretrained backbone (lr).
    """

    def __init__(self, model, lr: float = 0.01, wt_dec: float = 5e-4, max_step: int = 1000, power: float = 0.9, process_group=None,
                 bucket_mb: float = 48.0, loss_scale: Optional[float] = None, overlap_wgrad: bool = True, deterministic: Optional[bool] = None,
                 grad_payload: str = "fp32", share: str = "reserve+queue", reserved_cus: Optional[int] = None, defer_optimizer: bool = False):
        from .revise_net import FCAT, Net

        assert isinstance(model, Net) and next(model.parameters()).is_cuda
        self.model, self.FCAT = model, FCAT
        self.deterministic = deterministic  # torch.use_deterministic_algorithms(True) of revise_pseudo_labels.py:140-146; None = follow that switch
        model.launch.deterministic = deterministic  # per-trainer state, carried by this model's launch options
        self.wgrad_stream = torch.cuda.Stream(device=next(model.parameters()).device) if overlap_wgrad else None  # see backward_backbone
        self.lr0, self.wt_dec, self.max_step, self.power = lr, wt_dec, max_step, power
        self.global_step = 0
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        dev = next(model.parameters()).device
        model.train()
        self._use_arena(model)  # heads (from_scratch_layers) first, then b7 ... b3
        self._init_deferred(defer_optimizer, dev)
        self.buf_flat = torch.zeros_like(self.p_flat)
        self._init_dynamic_scale(dev, 1024.0, loss_scale)
        self.reducer: Optional[BucketedAllReduce] = None
        if self.world > 1:
            buckets = plan_buckets([(name, p.numel()) for name, p in self.entries], int(bucket_mb * (1 << 20) / 4))
            self.reducer = BucketedAllReduce(self.g_flat, buckets, process_group, launch_opts=model.launch, payload=grad_payload, share=share,
                                             reserved_cus=default_reserved_cus() if reserved_cus is None else reserved_cus)

    def train_step(self, x: Tensor, pmask: Tensor, pcam: Tensor, label: Tensor):
        """x [N,3,H,W]; pmask/pcam [N,C,32,32] with the zero background channel; label [N,C] with label[:,0] = 1.
        Returns (loss, loss_cls, loss_rfm, loss_ecr) as 1-element device tensors."""
        from .rfm_loss import rfm_losses

        model = self.model
        self._zero_grads()
        drop = model.sample_dropout(x.shape[0], x.device)
        outs, ctx = model.rfm_forward(x, pmask, pcam, save=True, drop=drop)
        # (the loss block's top-k backward chooses among exact ties: deterministic mode takes them in index order)
        losses, d_outs = rfm_losses(outs, pmask, pcam, label, want_grad=True, grad_scale=self.loss_scale / self.world, deterministic=self.deterministic)
        if self.reducer is not None:
            self.reducer.begin_step()

        def after(name):
            if name == "heads":  # fc8 / f8_3 / f8_4 / f9 gradients are final: unpack f9 into its arena slots
                self.arena.heads_done()
                if self.reducer is not None:
                    for nm in ("fc8", "f8_3", "f8_4", "f9_1", "f9_2"):
                        self.reducer.on_unit_done(nm)
            elif self.reducer is not None:
                self.reducer.on_unit_done(name)

        model.rfm_backward(ctx, list(d_outs), self.grads, after_unit=after, wgrad_stream=self.wgrad_stream)
        if self.reducer is not None:
            self.reducer.finish()
        inv = 1.0 / self.loss_scale  # the scale these gradients were produced with
        ns, tot = self.n_scratch, self.p_flat.numel()
        shadow = lambda lo, hi: None if self.pb_flat is None else self.pb_flat[lo:hi]  # noqa: E731
        groups = [(lo, hi, lr) for lo, hi, lr in ((0, ns, 10 * self.lr0), (ns, tot, self.lr0)) if hi > lo]  # scratch heads at 10 x lr (revise_pseudo_labels.py:173-176)
        with self._optimizer_scope():
            if self.dynamic_scale:
                # overflow check, `first step` and the poly schedule on the device (steps applied so far = opt_state[0]): no host read per step
                self._count_nonfinite()
                for k, (lo, hi, lr) in enumerate(groups):
                    ops.sgd_step_guarded(self.p_flat[lo:hi], self.g_flat[lo:hi], self.buf_flat[lo:hi], shadow(lo, hi), lr, self.wt_dec, self.wt_dec,
                                         self.opt_state, advance=k == len(groups) - 1, poly_max_step=self.max_step, poly_power=self.power, grad_inv_scale=inv)
                self._enqueue_flag()
            else:
                # utils.PolyOptimizer.step
                mult = (1 - self.global_step / self.max_step) ** self.power if self.global_step < self.max_step else None
                if mult is not None:
                    self._lr_mult = mult
                mult = getattr(self, "_lr_mult", 1.0)
                first = self.global_step == 0
                for lo, hi, lr in groups:
                    ops.sgd_step(self.p_flat[lo:hi], self.g_flat[lo:hi], self.buf_flat[lo:hi], shadow(lo, hi), lr * mult, self.wt_dec, self.wt_dec, first,
                                 grad_inv_scale=inv)
                self.global_step += 1
            self._optimizer_done()
        model.invalidate_weight_cache()
        return losses

    def _step_applied(self) -> None:
        self.global_step += 1
