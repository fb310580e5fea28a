"""ResNet38-d backbone on MI355X -- host-side mirror of the reference's `models/resnet38d.py`.

Same module tree / state-dict keys (`b{2..7}[_k].{bn_branch2a,conv_branch2a,bn_branch2b1,conv_branch2b1,
[bn_branch2b2,conv_branch2b2],[conv_branch1]}.*`, `conv1a.weight`, `bn7.*`), same `forward` /
`forward_as_dict` / `train()` behaviour (reference: models/resnet38d.py:119-213), but the nn.Conv2d /
nn.BatchNorm2d children are *parameter holders only*: the arithmetic runs as a fixed plan of fused HIP
launches (pistoseg_amd/ops.py -> libpistoseg_hip.so):

  * activations are channels-last [N,H,W,C] in the compute dtype (bf16 / fp16, or f32 for the exact parity path) -- or, with
    precision="bf16x3", SPLIT bf16: value = hi + lo (16 significant bits), stored as 2 C bf16 channels per pixel in blocks of 32 logical
    channels [hi(32) | lo(32)], weights alike; the 16-bit MFMA kernels stage them like plain tensors and multiply every K-line three
    times (x_hi w_hi + x_hi w_lo + x_lo w_hi, f32 accumulation): the reference's fp32 results (resnet38d.py:156-188) to 1e-4 at a
    third of the bf16 MFMA work instead of sixteen times (exact-f32 MFMA);
    precision="fp16x3" is the same on fp16 planes (22 instead of 16 significant bits per value, fp16's range, loss-scaled gradients);
  * every conv writes, from its epilogue, the NEXT BatchNorm+ReLU(+Dropout2d) already applied, plus the raw
    residual stream only where an identity shortcut will read it (pre-activation net: BN/ReLU are
    frozen per-channel affines, resnet38d.py:206-211);
  * backward is an explicit reverse plan (dgrad with the ReLU/BN mask fused, wgrad into f32) -- see
    `backward_backbone`.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .ops import ConvSpec

Tensor = torch.Tensor
BN_EPS = 1e-5

# name, kind, cin, cmid, cout, stride, first_dilation, dilation, dropout  (models/resnet38d.py:125-146)
UNITS = [
    ("b2", "res", 64, 128, 128, 2, 1, 1, 0.0),
    ("b2_1", "res", 128, 128, 128, 1, 1, 1, 0.0),
    ("b2_2", "res", 128, 128, 128, 1, 1, 1, 0.0),
    ("b3", "res", 128, 256, 256, 2, 1, 1, 0.0),
    ("b3_1", "res", 256, 256, 256, 1, 1, 1, 0.0),
    ("b3_2", "res", 256, 256, 256, 1, 1, 1, 0.0),
    ("b4", "res", 256, 512, 512, 2, 1, 1, 0.0),
    ("b4_1", "res", 512, 512, 512, 1, 1, 1, 0.0),
    ("b4_2", "res", 512, 512, 512, 1, 1, 1, 0.0),
    ("b4_3", "res", 512, 512, 512, 1, 1, 1, 0.0),
    ("b4_4", "res", 512, 512, 512, 1, 1, 1, 0.0),
    ("b4_5", "res", 512, 512, 512, 1, 1, 1, 0.0),
    ("b5", "res", 512, 512, 1024, 1, 1, 2, 0.0),
    ("b5_1", "res", 1024, 512, 1024, 1, 2, 2, 0.0),
    ("b5_2", "res", 1024, 512, 1024, 1, 2, 2, 0.0),
    ("b6", "bot", 1024, 512, 2048, 1, 4, 4, 0.3),
    ("b7", "bot", 2048, 1024, 4096, 1, 4, 4, 0.5),
]
TAP_OF_UNIT = {"b4": "conv3", "b5": "conv4", "b6": "conv5"}  # x_bn_relu taps, resnet38d.py:172,180,184


_OWNERS: "weakref.WeakValueDictionary[int, Net]" = weakref.WeakValueDictionary()  # id(model) -> model, see Net.train


def owner_of(p):
    """The `Net` whose `train()` tagged parameter p (None for foreign parameters, copies, or a model that is gone)."""
    m = _OWNERS.get(getattr(p, "_ps_owner", None))
    return m if m is not None and any(q is p for q in m.parameters()) else None


def _channels_last_(conv: nn.Conv2d) -> None:
    """Store an OIHW weight as [cout][kh][kw][cin] (torch.channels_last) -- the kernels' W_fwd layout."""
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)


class ResBlock(nn.Module):
    """Parameter holder for models/resnet38d.py:6-51."""

    def __init__(self, cin, cmid, cout, stride=1, first_dilation=None, dilation=1):
        super().__init__()
        self.same_shape = cin == cout and stride == 1
        first_dilation = dilation if first_dilation is None else first_dilation
        self.bn_branch2a = nn.BatchNorm2d(cin)
        self.conv_branch2a = nn.Conv2d(cin, cmid, 3, stride, padding=first_dilation, dilation=first_dilation, bias=False)
        self.bn_branch2b1 = nn.BatchNorm2d(cmid)
        self.conv_branch2b1 = nn.Conv2d(cmid, cout, 3, padding=dilation, dilation=dilation, bias=False)
        if not self.same_shape:
            self.conv_branch1 = nn.Conv2d(cin, cout, 1, stride, bias=False)


class ResBlock_bot(nn.Module):
    """Parameter holder for models/resnet38d.py:53-101."""

    def __init__(self, cin, cout, stride=1, dilation=1, dropout=0.0):
        super().__init__()
        self.same_shape = cin == cout and stride == 1
        self.bn_branch2a = nn.BatchNorm2d(cin)
        self.conv_branch2a = nn.Conv2d(cin, cout // 4, 1, stride, bias=False)
        self.bn_branch2b1 = nn.BatchNorm2d(cout // 4)
        self.dropout_2b1 = nn.Dropout2d(dropout)
        self.conv_branch2b1 = nn.Conv2d(cout // 4, cout // 2, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn_branch2b2 = nn.BatchNorm2d(cout // 2)
        self.dropout_2b2 = nn.Dropout2d(dropout)
        self.conv_branch2b2 = nn.Conv2d(cout // 2, cout, 1, bias=False)
        self.conv_branch1 = nn.Conv2d(cin, cout, 1, stride, bias=False)


class _Saved:
    """Activations kept by a training forward for `backward_backbone` (channels-last, compute dtype)."""

    def __init__(self):
        self.unit_in: Dict[str, Tensor] = {}   # activated input a of each unit
        self.mid: Dict[str, Tuple[Tensor, ...]] = {}  # a2 (, a3)
        self.drop: Dict[str, Optional[Tensor]] = {}
        self.hw: Dict[str, Tuple[int, int]] = {}  # input spatial size of each unit
        self.conv6: Optional[Tensor] = None
        self.n = 0


class Net(nn.Module):
    """Drop-in for the reference's `models.resnet38d.Net` (same constructor, keys and methods)."""

    def __init__(self, precision: str = "bf16", units=None):
        super().__init__()
        assert precision in ("bf16", "fp16", "fp32", "bf16x3", "fp16x3")
        self.precision = precision
        self.split = precision in ("bf16x3", "fp16x3")  # activations / gradients in the split layout (ops.conv2d_fwd(split=True), PS_BF16X3 / PS_F16X3)
        self.cm = ops.SPLIT_WIDTH if self.split else 1  # stored channels per logical channel
        self.launch = ops.LaunchOpts()      # this model's launch options (tiles_per_block / gpu_shared / deterministic), passed to every conv launch
        # measurement hook (bench.py's instrumented step): run a ONE-stream backward with the launch schedule of the two-stream one (gpu_shared set
        # for the backward's duration), so that per-kernel times are exclusive AND describe the dispatches of the timed step
        self.shared_backward_schedule = False
        self.units = list(UNITS if units is None else units)  # the OEEM stage-0 net differs in b7's dilation only (oeem.py)
        self.conv1a = nn.Conv2d(3, 64, 3, padding=1, bias=False)
        for name, kind, cin, cmid, cout, stride, fdil, dil, p in self.units:
            if kind == "res":
                unit = ResBlock(cin, cmid, cout, stride=stride, first_dilation=fdil, dilation=dil)
            else:
                unit = ResBlock_bot(cin, cout, stride=stride, dilation=dil, dropout=p)
            self.add_module(name, unit)
        self.bn7 = nn.BatchNorm2d(4096)
        self.not_training = [self.conv1a]
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m is not self.conv1a:
                _channels_last_(m)
        self.fuse_bottleneck = True       # shortcut conv + last 1x1 of a bottleneck unit (and their dgrads) as one K-concatenated GEMM
        self.hi_copies = True             # split precisions, training: activations / gradients also keep a plain 16-bit copy of their hi halves for the weight gradients (ops.attach_hi)
        self._out_grad_buf: Dict[str, Tensor] = {}  # unit name -> [N,H,W, cout + cout/4] buffer whose first cout channels hold dL/d(unit output)
        self._cache: Dict[str, Tuple] = {}
        self._weights_epoch = 0           # bumped by code that rewrites parameter memory behind torch's back
        self._bf16_shadow: Dict[str, Tensor] = {}  # param name -> bf16 W_fwd view kept fresh by the fused optimiser
        self._shadow_version: Dict[str, int] = {}  # param name -> the parameter's torch version the shadow was last derived at
        self._drop_seed, self._drop_calls = None, 0  # Philox key / per-call counter of the Dropout2d masks (sample_dropout)
        self._wd_plan: Dict[str, Tuple] = {}       # cache key -> (deps, [(src fn, dst view, cout, taps, cin)]): see refresh_dgrad_weights
        # Where the autograd nodes of the reference-style calls (`model(x)` under grad mode, then `loss.backward()`) put the weight gradients:
        #   "arena"    (default) straight into the model's flat gradient arena, which every parameter's `.grad` is a view of (arena.ParamArena):
        #              no per-step allocation / zero fill / accumulate pass; `arena.ArenaAdamW` / `arena.PolyOptimizer` then step in one launch;
        #   "autograd" returned to autograd as fresh tensors (needed under a torch DistributedDataParallel wrapper, whose hooks fire on
        #              accumulation, and for torch.autograd.grad).
        self.grad_sink = "arena"
        self.overlap_wgrad = True  # autograd path: weight gradients on a side stream (see backward_backbone), as the native trainers run them
        # A trainer that runs its optimiser on a side stream (trainer.SegTrainer(defer_optimizer=True)) leaves here the event behind that launch: the
        # forward waits for it in front of the first TRAINABLE unit -- conv1a and the frozen b2 units (models/resnet38d.py:191-205) run beside the update
        self._weights_event = None
        self.train(True)  # apply the freezing rules from the start (the reference's scripts always call train())

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        return self.forward_as_dict(x)["conv6"]

    def forward_as_dict(self, x):
        """NCHW f32 feature dict {conv3, conv4, conv5, conv6} (resnet38d.py:159-188)."""
        feats, _ = self.run_backbone(x, save=False, drop=self.sample_dropout(x.shape[0], x.device) if self.training else None)
        return {k: self.act_to_f32(v).permute(0, 3, 1, 2) for k, v in feats.items()}

    def train(self, mode=True):
        """Quirk kept from resnet38d.py:191-213: freezes `not_training` layers and every BatchNorm (which
        always runs in eval mode), and returns None."""
        super().train(mode)
        for layer in self.not_training:
            if isinstance(layer, nn.Conv2d):
                layer.weight.requires_grad = False
            elif isinstance(layer, nn.Module):
                for c in layer.children():
                    if getattr(c, "weight", None) is not None:
                        c.weight.requires_grad = False
                    if getattr(c, "bias", None) is not None:
                        c.bias.requires_grad = False
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()
                layer.bias.requires_grad = False
                layer.weight.requires_grad = False
        self._freeze_epoch = getattr(self, "_freeze_epoch", 0) + 1  # (arena.ParamArena re-checks the trainable set when this moves)
        _OWNERS[id(self)] = self
        for p in self.parameters():  # lets an optimiser built from bare parameter lists find the model's arena (arena.model_of)
            p._ps_owner = id(self)   # (a plain key into a weak registry: parameter objects stay picklable)
        return

    # ------------------------------------------------------------------ device-side views of the parameters
    @property
    def compute_dtype(self):
        return {"bf16": torch.bfloat16, "fp16": torch.float16, "bf16x3": torch.bfloat16, "fp16x3": torch.float16}.get(self.precision, torch.float32)

    def act_to_f32(self, t: Tensor) -> Tensor:
        """An activation / gradient tensor of the plans ([N,H,W,cm*C], possibly a channel slice) as a contiguous f32 [N,H,W,C]: the edge
        between the conv stack's storage format and the f32 head / loss kernels (f32 tensors pass through)."""
        if t.dtype == torch.float32:
            return t  # (possibly a channel slice: every kernel takes the channel stride)
        c = t.shape[3] // self.cm
        out = torch.empty(tuple(t.shape[:3]) + (c,), device=t.device, dtype=torch.float32)
        ops.convert_rows(t, out, c, src_split=self.split)
        return out

    def act_from_f32(self, t: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """The inverse: f32 [N,H,W,C] -> the plans' storage format (into `out`, which may be a channel slice, or a new tensor)."""
        c = t.shape[3]
        if out is None:
            if self.precision == "fp32":
                return t
            out = torch.empty(tuple(t.shape[:3]) + (self.cm * c,), device=t.device, dtype=self.compute_dtype)
        if out.dtype == torch.float32:
            out.copy_(t)
        else:
            ops.convert_rows(t, out, c, dst_split=self.split)
            if getattr(out, "_ps_hi", None) is not None:  # the plain copy of the hi halves (ops.attach_hi): round16(t) is the split tensor's hi
                ops.convert_rows(t, out._ps_hi, c)
        return out

    def _cached(self, key: str, deps: Tuple[Tensor, ...], make, raw_pointer_updates: bool = True):
        # raw_pointer_updates: the tensors may be rewritten behind torch's back by the fused optimiser (conv weights in the
        # training arena) -> also key on the weights epoch.  BatchNorm tensors are frozen (resnet38d.py:206-211) and only
        # ever change through torch (load_state_dict bumps _version), so their affine maps survive optimiser steps.
        sig = tuple((t.data_ptr(), t._version) for t in deps) + (self.precision, self._weights_epoch if raw_pointer_updates else 0)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = make()
        self._cache[key] = (sig, val)
        return val

    def w_fwd(self, conv: nn.Conv2d, key: str, f32: bool = False) -> Tensor:
        """[cout][kh][kw][cin] in the compute dtype (the f32 parameter storage itself on the fp32 path, or with f32=True: the heads that
        compute in f32 inside the 16-bit models)."""
        w = conv.weight
        if not w.is_contiguous(memory_format=torch.channels_last) and w.shape[2] > 1:
            _channels_last_(conv)
            w = conv.weight
        flat = w.detach().permute(0, 2, 3, 1)
        assert flat.is_contiguous()
        if self.precision == "fp32" or f32:
            return flat
        if self.split:  # [cout][kh][kw][cin in the split layout], re-derived from the f32 master whenever it changed

            def make_split():
                cout, kh, kw, cin = flat.shape
                out = torch.empty((cout, kh, kw, self.cm * cin), device=w.device, dtype=self.compute_dtype)
                ops.convert_rows(flat.reshape(cout * kh * kw, cin), out.view(cout * kh * kw, self.cm * cin), cin, dst_split=True, weights=True)
                return out

            return self._cached("wf:" + key, (w,), make_split)
        shadow = self._bf16_shadow.get(key + ".weight")
        if shadow is not None:
            # The fused optimiser refreshes the shadow in the launch that updates the f32 master (raw pointers: no version bump).
            # Any torch-side write since -- load_state_dict on resume, re-initialisation, a manual copy_ -- bumps the parameter's
            # version: re-cast from the master before the stale 16-bit weights can be used.
            if w._version != self._shadow_version.get(key + ".weight"):
                ops.cast_f32_lowp(flat, shadow)
                self._shadow_version[key + ".weight"] = w._version
            return shadow

        def make():
            out = torch.empty(flat.shape, device=w.device, dtype=self.compute_dtype)
            ops.cast_f32_lowp(flat, out)
            return out

        return self._cached("wf:" + key, (w,), make)

    def w_dgrad(self, conv: nn.Conv2d, key: str, f32: bool = False) -> Tensor:
        """[cin][kh][kw][cout] in the compute dtype (f32=True: in f32, see w_fwd)."""
        w = conv.weight
        cout, cin, k, _ = w.shape
        if (f32 and self.precision != "fp32") or self.split:
            # transposed in f32, then (split path) split along cout; re-derived lazily when the master changes

            def make_f32():
                t = torch.empty((cin, k, k, cout), device=w.device, dtype=torch.float32)
                ops.weight_transpose(self.w_fwd(conv, key, f32=True), t, cout, k * k, cin)
                if f32:
                    return t
                out = torch.empty((cin, k, k, self.cm * cout), device=w.device, dtype=self.compute_dtype)
                ops.convert_rows(t.view(cin * k * k, cout), out.view(cin * k * k, self.cm * cout), cout, dst_split=True, weights=True)
                return out

            return self._cached(("wd32:" if f32 else "wd:") + key, (w,), make_f32)

        def make():
            src = self.w_fwd(conv, key)
            out = torch.empty((cin, k, k, cout), device=w.device, dtype=self.compute_dtype)
            ops.weight_transpose(src, out, cout, k * k, cin)
            self._wd_plan["wd:" + key] = ((w,), [(lambda: self.w_fwd(conv, key), out.view(cin * k * k, cout), cout, k * k, cin)])
            return out

        return self._cached("wd:" + key, (w,), make)

    def refresh_dgrad_weights(self) -> None:
        """Re-derives every data-gradient weight layout the reverse plan has asked for before (w_dgrad / w_dgrad_cat) whose source
        changed, in ONE launch into the existing buffers -- instead of ~33 lazily triggered transposes per step."""
        items, fresh = [], []
        for ckey, (deps, parts) in self._wd_plan.items():
            hit = self._cache.get(ckey)
            sig = tuple((t.data_ptr(), t._version) for t in deps) + (self.precision, self._weights_epoch)
            if hit is None or hit[0] == sig:
                continue
            if hit[1].dtype != self.compute_dtype or hit[1].device != deps[0].device:  # precision / device changed: rebuild lazily
                del self._cache[ckey]
                continue
            for src_fn, dst, cout, taps, cin in parts:
                items.append((src_fn().reshape(cout, taps * cin), dst, cout, taps, cin))
            fresh.append((ckey, sig, hit[1]))
        if items:
            ops.weight_transpose_batched(items)
        for ckey, sig, val in fresh:
            self._cache[ckey] = (sig, val)

    def w_fwd_cat(self, unit: "ResBlock_bot", name: str) -> Tensor:
        """[cout][cin + cout/2]: conv_branch1 and conv_branch2b2 side by side along K, so that
        `branch1(a) + branch2b2(a3)` (resnet38d.py:76-97) is one 1x1 conv over the concatenated activation [a | a3]."""
        w1, w2 = unit.conv_branch1.weight, unit.conv_branch2b2.weight
        cout, cin, c2 = w1.shape[0], w1.shape[1], w2.shape[1]

        def make():
            m = self.cm  # (split path: conv_branch1's split channels, then conv_branch2b2's -- the order of the activation buffer [a | a3])
            out = torch.empty((cout, m * (cin + c2)), device=w1.device, dtype=self.compute_dtype)
            ops.copy_rows(self.w_fwd(unit.conv_branch1, name + ".conv_branch1").reshape(cout, m * cin), out[:, :m * cin])
            ops.copy_rows(self.w_fwd(unit.conv_branch2b2, name + ".conv_branch2b2").reshape(cout, m * c2), out[:, m * cin:])
            return out

        return self._cached("wfc:" + name, (w1, w2), make)

    def w_dgrad_cat(self, unit: "ResBlock_bot", name: str) -> Tensor:
        """[cin][cout + cout/4]: the transposed conv_branch1 and conv_branch2a side by side along K (both read the unit's activated
        input): d/da = W1^T G + W2a^T g2 is one 1x1 data gradient over the concatenated gradient [G | g2]."""
        w1, w2 = unit.conv_branch1.weight, unit.conv_branch2a.weight
        cout, cin, c4 = w1.shape[0], w1.shape[1], w2.shape[0]

        def make_split():
            m = self.cm
            out = torch.empty((cin, m * (cout + c4)), device=w1.device, dtype=self.compute_dtype)
            for conv, cname, lo, co in ((unit.conv_branch1, ".conv_branch1", 0, cout), (unit.conv_branch2a, ".conv_branch2a", m * cout, c4)):
                t = torch.empty((cin, co), device=w1.device, dtype=torch.float32)
                ops.weight_transpose(self.w_fwd(conv, name + cname, f32=True), t, co, 1, cin)
                ops.convert_rows(t, out[:, lo:lo + m * co], co, dst_split=True, weights=True)
            return out

        if self.split:
            return self._cached("wdc:" + name, (w1, w2), make_split)

        def make():
            out = torch.empty((cin, cout + c4), device=w1.device, dtype=self.compute_dtype)
            parts = [(lambda: self.w_fwd(unit.conv_branch1, name + ".conv_branch1"), out[:, :cout], cout, 1, cin),
                     (lambda: self.w_fwd(unit.conv_branch2a, name + ".conv_branch2a"), out[:, cout:], c4, 1, cin)]
            ops.weight_transpose_batched([(fn().reshape(co, cin), dst, co, 1, cin) for fn, dst, co, _t, _c in parts])
            self._wd_plan["wdc:" + name] = ((w1, w2), parts)
            return out

        return self._cached("wdc:" + name, (w1, w2), make)

    def _want_hi(self) -> bool:
        """Split precisions: whether training tensors get plain 16-bit companions for the weight gradients (ops.attach_hi) -- not when all three
        product terms are asked for (`launch.wgrad_terms = 3`: those launches read the split tensors themselves)."""
        return bool(self.split and self.hi_copies and self.launch.wgrad_terms in (None, 0, 1))

    def alloc_unit_out_grad(self, name: str, n: int, h: int, w: int, device, dtype) -> Tensor:
        """Buffer for dL/d(output of unit `name`) [N,h,w,cout].  For a bottleneck unit it is the first cout channels of a wider
        buffer that the reverse plan completes with the branch gradient g2, so the two data gradients into the unit's input run
        as one GEMM (w_dgrad_cat).  Callers that produce the gradient of the LAST unit's output (head backward) should use this."""
        u = next(x for x in self.units if x[0] == name)
        cout = u[4]
        m = self.cm
        want_hi = self._want_hi()
        if u[1] == "bot" and self.fuse_bottleneck:
            buf = torch.empty((n, h, w, m * (cout + cout // 4)), device=device, dtype=dtype)
            self._out_grad_buf[name] = buf
            out = buf[..., :m * cout]
            if want_hi:  # the plain 16-bit companions of [G | g2] and of G (ops.attach_hi)
                ops.attach_hi(buf, torch.empty((n, h, w, cout + cout // 4), device=device, dtype=dtype))
                ops.attach_hi(out, buf._ps_hi[..., :cout])
            return out
        t = torch.empty((n, h, w, m * cout), device=device, dtype=dtype)
        return ops.attach_hi(t, torch.empty((n, h, w, cout), device=device, dtype=dtype)) if want_hi else t

    def bn_affine(self, bn: nn.BatchNorm2d, key: str) -> Tuple[Tensor, Tensor]:
        """Eval-mode BN as y = x*scale + shift (f32 per-channel vectors; BN is frozen on this path)."""

        def make():
            with torch.no_grad():
                scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                shift = bn.bias.float() - bn.running_mean.float() * scale
            return scale.contiguous(), shift.contiguous()

        return self._cached("bn:" + key, (bn.weight, bn.bias, bn.running_mean, bn.running_var), make, raw_pointer_updates=False)

    def dropout_segments(self):
        """(name, channels, p) of every Dropout2d of the net (resnet38d.py:63,67,85,90), in a fixed order."""
        segs = []
        for name, kind, cin, cmid, cout, stride, fdil, dil, p in self.units:
            if kind == "bot" and p > 0:
                segs += [(f"{name}.dropout_2b1", cout // 4, p), (f"{name}.dropout_2b2", cout // 2, p)]
        return segs

    def sample_dropout(self, n: int, device) -> Dict[str, Tensor]:
        """Per-(sample, channel) Dropout2d multipliers for a training forward: one HIP launch for all of them (Philox keyed by torch's
        seed at first use + a per-call counter, so `torch.manual_seed` makes training runs repeatable)."""
        segs = self.dropout_segments()
        if not segs:
            return {}
        if self._drop_seed is None:
            self._drop_seed = int(torch.initial_seed())
        self._drop_calls += 1
        return ops.dropout2d_masks(segs, n, device, self._drop_seed, self._drop_calls)

    def unit_specs(self, name, kind, cin, cmid, cout, stride, fdil, dil):
        if kind == "res":
            return {
                "conv_branch2a": ConvSpec(cin, cmid, 3, stride, fdil),
                "conv_branch2b1": ConvSpec(cmid, cout, 3, 1, dil),
                "conv_branch1": ConvSpec(cin, cout, 1, stride, 1),
            }
        return {
            "conv_branch2a": ConvSpec(cin, cout // 4, 1, stride, 1),
            "conv_branch2b1": ConvSpec(cout // 4, cout // 2, 3, 1, dil),
            "conv_branch2b2": ConvSpec(cout // 2, cout, 1, 1, 1),
            "conv_branch1": ConvSpec(cin, cout, 1, stride, 1),
        }

    # ------------------------------------------------------------------ forward plan
    def run_backbone(self, x: Tensor, save: bool, drop: Optional[Dict[str, Tensor]] = None, head: Optional[Tensor] = None):
        """x: NCHW f32 on the GPU.  Returns ({conv3,conv4,conv5,conv6} channels-last, _Saved or None).
        head (inference only: save=False, no dropout): the narrow 1x1 head's weight [C, 4096] f32 on conv6.  Where the fused kernel serves
        the last unit's final conv (16-bit paths), that launch reduces relu(bn7(.)) straight to `feats["cam"]` [n,h,w,C] f32 and conv6 is
        never materialised (`feats` then has no "conv6")."""
        if not x.is_cuda:
            raise RuntimeError("pistoseg_amd.resnet38d.Net runs on the GPU only (no CPU fallback); move the module and input to cuda")
        x = x.contiguous().float()
        n, _, h, w = x.shape
        dt, dev = self.compute_dtype, x.device
        drop = drop or {}
        saved = _Saved() if save else None
        feats: Dict[str, Tensor] = {}

        cm, kw = self.cm, dict(split=self.split, opts=self.launch)

        # training on the split types: an activation that a TRAINABLE conv reads also keeps a plain 16-bit copy of its hi halves, written by the
        # same epilogue, for that conv's weight gradient (ops.attach_hi).  Only those: the extra store is not free (the frozen front units hold the
        # largest tensors), and raw / shortcut tensors are never weight-gradient operands
        first_tr = self.first_trainable_unit() if (self._want_hi() and saved is not None) else len(self.units)

        def new(hh, ww, c, hi=False):  # c LOGICAL channels (split path: 2 stored channels each)
            t = torch.empty((n, hh, ww, cm * c), device=dev, dtype=dt)
            return ops.attach_hi(t, torch.empty((n, hh, ww, c), device=dev, dtype=dt)) if hi else t

        def cslice(t, lo, hi_):  # logical channels [lo, hi_) of a (split) tensor, with its companion's slice
            v = t[..., cm * lo:cm * hi_]
            return ops.attach_hi(v, t._ps_hi[..., lo:hi_]) if getattr(t, "_ps_hi", None) is not None else v

        def new_unit_input(hh, ww, idx):
            """Activated input of unit idx; for a fused bottleneck unit it is the first cin channels of [a | a3]."""
            u = self.units[idx]
            if u[1] == "bot" and self.fuse_bottleneck:
                wide = new(hh, ww, u[2] + u[4] // 2, hi=idx >= first_tr)
                return cslice(wide, 0, u[2]), wide
            return new(hh, ww, u[2], hi=idx >= first_tr), None

        first = getattr(self, self.units[0][0])
        sc0, sh0 = self.bn_affine(first.bn_branch2a, self.units[0][0] + ".bn_branch2a")
        a = new(h, w, 64, hi=0 >= first_tr)
        a_wide = None
        if self.split:  # conv1a (3 -> 64) on the exact-f32 kernel, then cut into planes
            a32 = torch.empty((n, h, w, 64), device=dev, dtype=torch.float32)
            ops.conv1a_fwd(x, self.conv1a.weight.detach().contiguous(), sc0, sh0, a32)
            self.act_from_f32(a32, a)  # (fills the plain copy of the hi halves as well, where `a` carries one)
            del a32
        else:
            ops.conv1a_fwd(x, self.conv1a.weight.detach().contiguous(), sc0, sh0, a)
        xraw = None
        first_trainable = self.first_trainable_unit() if self._weights_event is not None else -1
        for i, (name, kind, cin, cmid, cout, stride, fdil, dil, _p) in enumerate(self.units):
            unit = getattr(self, name)
            if i == first_trainable:
                self.wait_weights()  # (a deferred optimiser launch: everything up to here read frozen weights only)
            specs = self.unit_specs(name, kind, cin, cmid, cout, stride, fdil, dil)
            if i + 1 < len(self.units):
                nxt_name, nxt = self.units[i + 1][0], getattr(self, self.units[i + 1][0])
                nscale, nshift = self.bn_affine(nxt.bn_branch2a, nxt_name + ".bn_branch2a")
                need_raw = self.units[i + 1][1] == "res" and nxt.same_shape
            else:
                nscale, nshift = self.bn_affine(self.bn7, "bn7")
                need_raw = False
            if name in TAP_OF_UNIT:
                feats[TAP_OF_UNIT[name]] = a
            ho, wo = specs["conv_branch2a"].out_hw(h, w)
            if saved is not None:
                saved.unit_in[name] = a
                saved.hw[name] = (h, w)
            same = kind == "res" and unit.same_shape
            fused = kind == "bot" and a_wide is not None and stride == 1
            if same:
                shortcut = xraw
            elif not fused:
                shortcut = new(ho, wo, cout)
                ops.conv2d_fwd(specs["conv_branch1"], a, self.w_fwd(unit.conv_branch1, name + ".conv_branch1"), out_raw=shortcut, **kw)
            last = i + 1 == len(self.units)
            # inference with the head folded into the last unit's final launch: conv6 is never written, so it is not even allocated
            head_fused = (last and fused and head is not None and not save and head.shape[1] == cout and not self.split
                          and ops.conv1x1_head_supported(ConvSpec(cin + cout // 2, cout, 1), a_wide, head.shape[0]) > 0)
            if not last:
                a_next, a_wide_next = new_unit_input(ho, wo, i + 1)
            else:
                a_next, a_wide_next = (None if head_fused else new(ho, wo, cout)), None
            xraw_next = new(ho, wo, cout) if need_raw else None
            if kind == "res":
                s1, b1 = self.bn_affine(unit.bn_branch2b1, name + ".bn_branch2b1")
                a2 = new(ho, wo, cmid, hi=i >= first_tr)
                ops.conv2d_fwd(specs["conv_branch2a"], a, self.w_fwd(unit.conv_branch2a, name + ".conv_branch2a"), bn_scale=s1, bn_shift=b1, out_act=a2, **kw)
                ops.conv2d_fwd(specs["conv_branch2b1"], a2, self.w_fwd(unit.conv_branch2b1, name + ".conv_branch2b1"), add0=shortcut,
                               out_raw=xraw_next, bn_scale=nscale, bn_shift=nshift, out_act=a_next, **kw)
                if saved is not None:
                    saved.mid[name] = (a2,)
            else:
                d1, d2 = drop.get(f"{name}.dropout_2b1"), drop.get(f"{name}.dropout_2b2")
                s1, b1 = self.bn_affine(unit.bn_branch2b1, name + ".bn_branch2b1")
                s2, b2 = self.bn_affine(unit.bn_branch2b2, name + ".bn_branch2b2")
                a2 = new(ho, wo, cout // 4, hi=i >= first_tr)
                ops.conv2d_fwd(specs["conv_branch2a"], a, self.w_fwd(unit.conv_branch2a, name + ".conv_branch2a"), bn_scale=s1, bn_shift=b1, drop=d1, out_act=a2, **kw)
                a3 = cslice(a_wide, cin, cin + cout // 2) if fused else new(ho, wo, cout // 2, hi=i >= first_tr)
                ops.conv2d_fwd(specs["conv_branch2b1"], a2, self.w_fwd(unit.conv_branch2b1, name + ".conv_branch2b1"), bn_scale=s2, bn_shift=b2, drop=d2, out_act=a3, **kw)
                if fused:  # branch1(a) + branch2b2(a3) = one 1x1 conv over [a | a3]: no shortcut tensor, no residual read
                    cat_spec = ConvSpec(cin + cout // 2, cout, 1)
                    if head_fused:
                        cam = torch.empty((n, ho, wo, head.shape[0]), device=dev, dtype=torch.float32)
                        ops.conv1x1_head_fwd(cat_spec, a_wide, self.w_fwd_cat(unit, name), nscale, nshift, head, cam)
                        feats["cam"] = cam
                    else:
                        ops.conv2d_fwd(cat_spec, a_wide, self.w_fwd_cat(unit, name), out_raw=xraw_next, bn_scale=nscale, bn_shift=nshift, out_act=a_next, **kw)
                else:
                    ops.conv2d_fwd(specs["conv_branch2b2"], a3, self.w_fwd(unit.conv_branch2b2, name + ".conv_branch2b2"), add0=shortcut,
                                   out_raw=xraw_next, bn_scale=nscale, bn_shift=nshift, out_act=a_next, **kw)
                if saved is not None:
                    saved.mid[name] = (a2, a3)
                    saved.drop[name + ".dropout_2b1"], saved.drop[name + ".dropout_2b2"] = d1, d2
            a, a_wide, xraw, h, w = a_next, a_wide_next, xraw_next, ho, wo
        if a is not None:
            feats["conv6"] = a
        if saved is not None:
            saved.conv6 = a
            saved.n = n
        self.wait_weights()  # (a net whose only trainable weights are the heads behind the backbone)
        return feats, saved

    # ------------------------------------------------------------------ backward plan
    def first_trainable_unit(self) -> int:
        for i, u in enumerate(self.units):
            unit = getattr(self, u[0])
            if any(p.requires_grad for p in unit.parameters()):
                return i
        return len(self.units)

    def backward_backbone(self, saved: _Saved, g_x7: Tensor, grads: Dict[str, Tensor], g_taps: Optional[Dict[str, Tensor]] = None,
                          after_unit=None, wgrad_stream=None) -> None:
        """Reverse plan.  g_x7: gradient w.r.t. b7's raw output (the ReLU(bn7) mask already applied),
        channels-last compute dtype.  grads: parameter name -> f32 buffer [cout][kh][kw][cin] that wgrad
        ACCUMULATES into (caller zeroes).  g_taps: gradients w.r.t. the conv4 / conv5 taps.
        after_unit(name): optional callback after a unit's weight gradients are complete (DDP buckets).
        wgrad_stream: optional side stream for the weight gradients.  They depend on the data-gradient chain but nothing in the
        backward depends on them, so on a second HIP stream their blocks fill the CUs that a data-gradient launch leaves idle in
        its last partial round (and vice versa); after_unit then runs with that stream current (the all-reduce waits on it), and
        the caller's stream waits for it at the end."""
        g_taps = g_taps or {}
        first = self.first_trainable_unit()
        G = g_x7
        self.refresh_dgrad_weights()
        self._out_grad_buf = {k: v for k, v in self._out_grad_buf.items() if k == self.units[-1][0]}  # drop stale buffers of aborted steps
        dt, dev, n = G.dtype, G.device, saved.n
        shared_before = self.launch.gpu_shared
        if wgrad_stream is not None or self.shared_backward_schedule:
            self.launch.gpu_shared = 1  # the data gradients' partial last rounds are filled by the side stream's weight-gradient blocks (ps_conv_geom.gpu_shared)
        try:
            self._backward_units(saved, G, grads, g_taps, after_unit, wgrad_stream, first, dt, dev, n)
        finally:
            self.launch.gpu_shared = shared_before
        if wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(wgrad_stream)

    def _backward_units(self, saved, G, grads, g_taps, after_unit, wgrad_stream, first, dt, dev, n) -> None:
        cm, kw = self.cm, dict(split=self.split, opts=self.launch)
        for i in range(len(self.units) - 1, -1, -1):
            name, kind, cin, cmid, cout, stride, fdil, dil, _p = self.units[i]
            if i < first:
                break
            unit = getattr(self, name)
            specs = self.unit_specs(name, kind, cin, cmid, cout, stride, fdil, dil)
            a = saved.unit_in[name]
            h, w = saved.hw[name]
            ho, wo = specs["conv_branch2a"].out_hw(h, w)
            need_dx = i > first
            tap = g_taps.get(TAP_OF_UNIT.get(name, ""))
            s_in, _ = self.bn_affine(unit.bn_branch2a, name + ".bn_branch2a")
            same = kind == "res" and unit.same_shape

            def wgrad(cname, x_act, dy):
                p = getattr(unit, cname).weight
                if not p.requires_grad:
                    return
                if wgrad_stream is None:
                    ops.conv2d_wgrad(specs[cname], x_act, dy, grads[f"{name}.{cname}.weight"], **kw)
                    return
                wgrad_stream.wait_stream(torch.cuda.current_stream())  # dy (and the zeroed gradient arena) are ready
                with torch.cuda.stream(wgrad_stream):
                    ops.conv2d_wgrad(specs[cname], x_act, dy, grads[f"{name}.{cname}.weight"], **kw)
                # keep the caching allocator from recycling them under the side stream -- the plain 16-bit companions too (ops.attach_hi):
                # in the split precisions they are separate allocations and the ones the weight gradient actually reads
                for t in (x_act, dy, ops._hi_of(x_act), ops._hi_of(dy)):
                    if t is not None:
                        t.record_stream(wgrad_stream)

            want_hi = self._want_hi()

            def new(hh, ww, c, hi=True):  # (hi: the tensor is some weight gradient's dY -- every `out` of a data gradient here is)
                t = torch.empty((n, hh, ww, cm * c), device=dev, dtype=dt)
                return ops.attach_hi(t, torch.empty((n, hh, ww, c), device=dev, dtype=dt)) if want_hi and hi else t

            if kind == "res":
                (a2,) = saved.mid[name]
                s1, _ = self.bn_affine(unit.bn_branch2b1, name + ".bn_branch2b1")
                wgrad("conv_branch2b1", a2, G)
                gh = new(ho, wo, cmid)
                ops.conv2d_dgrad(specs["conv_branch2b1"], G, self.w_dgrad(unit.conv_branch2b1, name + ".conv_branch2b1"), (ho, wo),
                                 mask_src=a2, bn_scale=s1, out=gh, **kw)
                wgrad("conv_branch2a", a, gh)
                if not same:
                    wgrad("conv_branch1", a, G)
                if need_dx:
                    Gp = new(h, w, cin)
                    if same:
                        assert tap is None
                        ops.conv2d_dgrad(specs["conv_branch2a"], gh, self.w_dgrad(unit.conv_branch2a, name + ".conv_branch2a"), (h, w),
                                         mask_src=a, bn_scale=s_in, add1=G, out=Gp, **kw)
                    else:
                        t = new(h, w, cin, hi=False)
                        ops.conv2d_dgrad(specs["conv_branch1"], G, self.w_dgrad(unit.conv_branch1, name + ".conv_branch1"), (h, w), add0=tap, out_raw=t, **kw)
                        ops.conv2d_dgrad(specs["conv_branch2a"], gh, self.w_dgrad(unit.conv_branch2a, name + ".conv_branch2a"), (h, w),
                                         add0=t, mask_src=a, bn_scale=s_in, out=Gp, **kw)
                    G = Gp
            else:
                a2, a3 = saved.mid[name]
                d1, d2 = saved.drop[name + ".dropout_2b1"], saved.drop[name + ".dropout_2b2"]
                s1, _ = self.bn_affine(unit.bn_branch2b1, name + ".bn_branch2b1")
                s2, _ = self.bn_affine(unit.bn_branch2b2, name + ".bn_branch2b2")
                # G may be the first cout channels of a [G | g2] buffer (alloc_unit_out_grad): then both data gradients into the
                # unit's input run as one GEMM over the concatenated channels
                GG = self._out_grad_buf.pop(name, None)
                fused = GG is not None and GG.data_ptr() == G.data_ptr() and stride == 1 and tuple(GG.shape[:3]) == (n, ho, wo)
                wgrad("conv_branch2b2", a3, G)
                g3 = new(ho, wo, cout // 2)
                ops.conv2d_dgrad(specs["conv_branch2b2"], G, self.w_dgrad(unit.conv_branch2b2, name + ".conv_branch2b2"), (ho, wo),
                                 mask_src=a3, bn_scale=s2, drop=d2, out=g3, **kw)
                wgrad("conv_branch2b1", a2, g3)
                if fused:
                    g2 = GG[..., cm * cout:]
                    if getattr(GG, "_ps_hi", None) is not None:
                        ops.attach_hi(g2, GG._ps_hi[..., cout:])
                else:
                    g2 = new(ho, wo, cout // 4)
                ops.conv2d_dgrad(specs["conv_branch2b1"], g3, self.w_dgrad(unit.conv_branch2b1, name + ".conv_branch2b1"), (ho, wo),
                                 mask_src=a2, bn_scale=s1, drop=d1, out=g2, **kw)
                wgrad("conv_branch2a", a, g2)
                wgrad("conv_branch1", a, G)
                if need_dx:
                    prev = self.units[i - 1][0]
                    Gp = self.alloc_unit_out_grad(prev, n, h, w, dev, dt)
                    if fused:
                        ops.conv2d_dgrad(ConvSpec(cin, cout + cout // 4, 1), GG, self.w_dgrad_cat(unit, name), (h, w),
                                         add0=tap, mask_src=a, bn_scale=s_in, out=Gp, **kw)
                    else:
                        t = new(h, w, cin, hi=False)
                        ops.conv2d_dgrad(specs["conv_branch1"], G, self.w_dgrad(unit.conv_branch1, name + ".conv_branch1"), (h, w), add0=tap, out_raw=t, **kw)
                        ops.conv2d_dgrad(specs["conv_branch2a"], g2, self.w_dgrad(unit.conv_branch2a, name + ".conv_branch2a"), (h, w),
                                         add0=t, mask_src=a, bn_scale=s_in, out=Gp, **kw)
                    G = Gp
            if after_unit is not None:
                if wgrad_stream is None:
                    after_unit(name)
                else:
                    with torch.cuda.stream(wgrad_stream):
                        after_unit(name)

    def wait_weights(self) -> None:
        """Make the current stream wait for a deferred optimiser launch (no-op otherwise): called by the forward plan in front of the first
        trainable unit, by `state_dict()` (hook below) and by anything else that reads trainable weights outside the plans."""
        ev = self._weights_event
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._weights_event = None

    def register_shadow(self, name: str, param: nn.Parameter, view: Tensor) -> None:
        """`view`: 16-bit [cout][kh][kw][cin] storage that a trainer keeps equal to `param` (cast by its fused optimiser)."""
        self._bf16_shadow[name] = view
        self._shadow_version[name] = -1  # force one cast from the master on first use

    def invalidate_weight_cache(self) -> None:
        """Call after parameter memory was rewritten by a raw-pointer kernel (fused optimiser step)."""
        self._weights_epoch += 1

    def trainable_conv_params(self) -> List[Tuple[str, nn.Parameter]]:
        """(name, parameter) of every conv weight that currently requires grad, in state-dict order."""
        return [(k, p) for k, p in self.named_parameters() if p.dim() == 4 and p.requires_grad]


_MX_BN = {"beta": "bias", "gamma": "weight", "mean": "running_mean", "var": "running_var"}


def mxnet_key_to_torch(key: str) -> Optional[str]:
    """One parameter name of the ImageNet ResNet38 MXNet checkpoint (`ilsvrc-cls_rna-a1_cls1000_ep-0001.params`: `arg:res3b1_branch2a_weight`,
    `aux:bn5a_branch2b1_moving_var`, `arg:bn7_gamma`, `arg:conv1a_weight`, `arg:linear1000_*`) -> this net's state-dict key, or None for
    the 1000-way classifier.  Unit `<stage>a` is `b<stage>`, `<stage>b<k>` is `b<stage>_<k>` (models/resnet38d.py:215-263)."""
    toks = key.split("_")
    head = toks[0]
    if "conv1a" in head:
        return "conv1a.weight"
    if "linear1000" in head:
        return None
    if len(toks) > 1 and "branch" in toks[1]:
        unit = f"b{head[-2]}" if head[-1] == "a" else f"b{head[-3]}_{head[-1]}"
        if "res" in head:
            return f"{unit}.conv_{toks[1]}.weight"
        return f"{unit}.bn_{toks[1]}.{_MX_BN[toks[-1]]}"
    return "bn7." + _MX_BN[toks[-1]]


def convert_mxnet_to_torch(source) -> Dict[str, Tensor]:
    """Mirror of the reference's `convert_mxnet_to_torch(filename)` (models/resnet38d.py:215-263; used by revise_pseudo_labels.py:179-181 for
    `.params` checkpoints).  `source`: a path (needs the `mxnet` package, as the reference does) or an already loaded {name: array} mapping
    (arrays with `.asnumpy()` or anything `numpy.asarray` takes)."""
    import numpy as np

    if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__"):
        import mxnet  # third-party, absent from the MI355X image: the reference imports it lazily in the same place

        source = mxnet.nd.load(source)
    out: Dict[str, Tensor] = {}
    for k, v in source.items():
        name = mxnet_key_to_torch(k)
        if name is not None:
            out[name] = torch.from_numpy(np.asarray(v.asnumpy() if hasattr(v, "asnumpy") else v))
    return out
