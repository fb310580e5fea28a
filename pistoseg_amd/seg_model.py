"""The ResNet38-d segmentation model: backbone -> fc8 (4096 -> classes, 1x1) -> bilinear upsample
(align_corners=True) to the input size.

This is the `cam` branch of the reference's `revise_net.Net.forward` (models/revise_net.py:50,86), the only
reference-defined "ResNet38-d -> per-pixel class logits" head; it plugs into
`SegmentationModule` / `MosaicModule` at the `smp.create_model(args.model, ...)` call site
(models/segmentation_module.py:72-81) through `create_model('ResNet38d', ...)`.  State-dict keys are the
backbone's plus `fc8.weight`, i.e. a subset of the RFM checkpoint (revise_pseudo_labels.py:214).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import ops, resnet38d
from .arena import ParamArena

Tensor = torch.Tensor


class ResNet38dSeg(resnet38d.Net):
    def __init__(self, classes: int = 3, precision: str = "bf16"):
        super().__init__(precision=precision)
        self.dropout7 = nn.Dropout2d(0.5)
        self.fc8 = nn.Conv2d(4096, classes, 1, bias=False)
        nn.init.xavier_uniform_(self.fc8.weight)
        self.from_scratch_layers = [self.fc8]
        self.not_training = [self.conv1a, self.b2, self.b2_1, self.b2_2]  # revise_net.py:27
        self.classes = classes
        self.fuse_head = True  # inference: fc8 fused into b7's last conv launch (ps_conv1x1_head_fwd) where the geometry allows
        self.train(True)

    def dropout_segments(self):
        return super().dropout_segments() + [("dropout7", 4096, 0.5)]  # revise_net.py:11,50

    # ------------------------------------------------------------------ head
    def head_forward(self, conv6: Tensor, drop7: Optional[Tensor], out_hw) -> (Tensor, Tensor):
        if self.split:
            conv6 = self.act_to_f32(conv6)  # the head reads conv6 = hi + lo in f32 (the narrow-head kernels take f32 / 16-bit tensors)
        n, g1, g2, _ = conv6.shape
        cam = torch.empty((n, g1, g2, self.classes), device=conv6.device, dtype=torch.float32)
        ops.fc8_fwd(conv6, self.fc8.weight.detach().reshape(self.classes, 4096), drop7, cam)
        logits = torch.empty((n, self.classes, out_hw[0], out_hw[1]), device=conv6.device, dtype=torch.float32)
        ops.bilinear_fwd(cam, "nhwc", logits, "nchw", True)
        return logits, cam

    def head_backward(self, conv6: Tensor, drop7: Optional[Tensor], dlogits: Tensor, dw8: Tensor) -> Tensor:
        """Returns the gradient w.r.t. b7's raw output; accumulates fc8's weight gradient into dw8 [C,4096]."""
        n, g1, g2, _ = conv6.shape
        dcam = torch.empty((n, g1, g2, self.classes), device=conv6.device, dtype=torch.float32)
        ops.bilinear_bwd(dlogits.contiguous(), "nchw", dcam, "nhwc", True)
        scale7, _ = self.bn_affine(self.bn7, "bn7")
        g_x7 = self.alloc_unit_out_grad(self.units[-1][0], n, g1, g2, conv6.device, conv6.dtype)  # [G | g2] buffer of the last unit
        if self.split:  # f32 head backward on conv6 = hi + lo, its f32 result cut into planes
            g32 = torch.empty((n, g1, g2, 4096), device=conv6.device, dtype=torch.float32)
            ops.fc8_bwd(self.act_to_f32(conv6), self.fc8.weight.detach().reshape(self.classes, 4096), drop7, scale7, dcam, g32, dw8)
            self.act_from_f32(g32, g_x7)
            return g_x7
        ops.fc8_bwd(conv6, self.fc8.weight.detach().reshape(self.classes, 4096), drop7, scale7, dcam, g_x7, dw8)
        return g_x7

    # ------------------------------------------------------------------ reference-style call
    def forward(self, x: Tensor) -> Tensor:
        """x: [N,3,H,W] f32 -> logits [N,classes,H,W] f32."""
        params = [p for _, p in self.trainable_conv_params()]
        if torch.is_grad_enabled() and params:
            return _SegFunction.apply(self, x, *params)
        n, _, h, w = x.shape
        limit = self.max_tiles_per_launch(h, w)
        if n > limit:  # the kernels address a tensor through 32-bit buffer offsets (< 2 GiB each): big inference batches
            return torch.cat([self.forward(x[i:i + limit]) for i in range(0, n, limit)], 0)  # (d4 TTA: 8 views x N) go in slices
        drop = self.sample_dropout(n, x.device) if self.training else {}
        # inference without dropout: fc8 folded into b7's last launch where the fused kernel serves it (16-bit paths); conv6 is then never written
        head = self.fc8.weight.detach().reshape(self.classes, 4096) if (self.fuse_head and not drop) else None
        feats, _ = self.run_backbone(x, save=False, drop=drop, head=head)
        if "cam" in feats:
            logits = torch.empty((n, self.classes, h, w), device=x.device, dtype=torch.float32)
            ops.bilinear_fwd(feats["cam"], "nhwc", logits, "nchw", True)
            return logits
        logits, _ = self.head_forward(feats["conv6"], drop.get("dropout7"), x.shape[-2:])
        return logits

    def max_tiles_per_launch(self, h: int, w: int) -> int:
        """Largest batch one forward plan can take: its biggest tensor (conv1a's 64-channel output at full resolution) must stay
        below 2 GiB, the range of the kernels' buffer descriptors (ps_conv2d_fwd rejects larger problems)."""
        esize = {"fp32": 4, "bf16x3": 4, "fp16x3": 4}.get(self.precision, 2)
        return max(1, ((1 << 31) - 1) // (h * w * 64 * esize))

    def new_grad_buffers(self, device) -> Dict[str, Tensor]:
        """Zeroed f32 gradient buffers in the kernels' [cout][kh][kw][cin] layout, one per trainable conv."""
        out = {}
        for name, p in self.trainable_conv_params():
            cout, cin, kh, kw = p.shape
            out[name] = torch.zeros((cout, kh, kw, cin), device=device, dtype=torch.float32)
        return out


class _SegFunction(torch.autograd.Function):
    """Whole-model autograd node: forward = the fused forward plan, backward = the explicit reverse plan."""

    @staticmethod
    def forward(ctx, model: ResNet38dSeg, x: Tensor, *params: Tensor):
        drop = model.sample_dropout(x.shape[0], x.device) if model.training else {}
        feats, saved = model.run_backbone(x, save=True, drop=drop)
        logits, _ = model.head_forward(feats["conv6"], drop.get("dropout7"), x.shape[-2:])
        ctx.model, ctx.saved_acts, ctx.drop7 = model, saved, drop.get("dropout7")
        if getattr(model, "debug_keep_saved", False):
            model._last_saved = saved  # tests compare ReLU patterns with the oracle
        ctx.names = [n for n, _ in model.trainable_conv_params()]
        return logits

    @staticmethod
    def backward(ctx, dlogits: Tensor):
        model, saved = ctx.model, ctx.saved_acts
        ctx.saved_acts = None
        if model.grad_sink != "arena":  # gradients handed to autograd (torch DDP's hooks, torch.autograd.grad): fresh buffers every backward
            grads = model.new_grad_buffers(dlogits.device)
            dw8 = grads["fc8.weight"].view(model.classes, 4096) if "fc8.weight" in grads else torch.zeros(
                (model.classes, 4096), device=dlogits.device)
            g_x7 = model.head_backward(saved.conv6, ctx.drop7, dlogits, dw8)
            model.backward_backbone(saved, g_x7, grads)
            return (None, None) + tuple(grads[n].permute(0, 3, 1, 2) for n in ctx.names)
        # the weight gradients go straight into the model's gradient arena, which the parameters' `.grad` are views of (arena.ParamArena):
        # nothing to allocate, zero or accumulate per step, weight gradients on the side stream, [N > 1: buckets all-reduced behind the plan]
        arena = ParamArena.of(model)
        arena.bind_param_grads()
        grads, red = arena.grads, arena.reducer
        if red is not None:  # SUM all-reduce of local-mean losses: the global mean's gradient is 1/world of each
            dlogits = dlogits * (1.0 / torch.distributed.get_world_size(red.group))
        dw8 = grads["fc8.weight"].view(model.classes, 4096) if "fc8.weight" in grads else torch.zeros(
            (model.classes, 4096), device=dlogits.device)
        g_x7 = model.head_backward(saved.conv6, ctx.drop7, dlogits, dw8)
        if red is not None:
            red.begin_step()
            red.on_unit_done("fc8")
        model.backward_backbone(saved, g_x7, grads, after_unit=red.on_unit_done if red is not None else None,
                                wgrad_stream=arena.side_stream() if model.overlap_wgrad else None)
        if red is not None:
            red.finish()
        return (None, None) + (None,) * len(ctx.names)


def create_model(arch: str, encoder_name: Optional[str] = None, in_channels: int = 3, classes: int = 3, precision: str = "bf16", **kw):
    """Plug point mirroring `smp.create_model(args.model, encoder_name=args.encoder, in_channels=3,
    classes=args.num_classes, ...)` (models/segmentation_module.py:72-81).

    `--model ResNet38d` gives the in-tree MI355X model.  Any other name is an smp architecture (run.sh's defaults are
    UnetPlusPlus / efficientnet-b0 / -b3: third-party bodies, out of this hot path's scope): it is handed to
    `segmentation_models_pytorch.create_model` unchanged when that package is installed, so an unmodified run.sh keeps working on
    PyTorch-ROCm eager kernels; without smp it is an error -- or, with PISTOSEG_SUBSTITUTE_MODEL=1, the ResNet38-d model stands in."""
    import os

    if arch.lower() in ("resnet38d", "resnet38-d", "resnet38d_seg"):
        assert in_channels == 3
        return ResNet38dSeg(classes=classes, precision=precision)
    try:
        import segmentation_models_pytorch as smp
    except ImportError:
        smp = None
    if smp is not None:
        return smp.create_model(arch, encoder_name=encoder_name, in_channels=in_channels, classes=classes, **kw)
    if os.environ.get("PISTOSEG_SUBSTITUTE_MODEL") == "1":
        import warnings

        warnings.warn(f"segmentation_models_pytorch is not installed: --model {arch} / --encoder {encoder_name} replaced by the ResNet38-d model")
        return ResNet38dSeg(classes=classes, precision=precision)
    raise ValueError(f"--model {arch!r} is an smp architecture and segmentation_models_pytorch is not installed; pistoseg_amd itself provides "
                     f"--model ResNet38d (set PISTOSEG_SUBSTITUTE_MODEL=1 to let it stand in)")
