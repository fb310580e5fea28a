"""RFM (revise) network on MI355X -- host-side mirror of the reference's `models/revise_net.py`.

`Net(num_classes=n_class+1)`, `forward(x, pmask, pcam) -> (cam, cam_rv, pmask_rv, pcam_rv)`,
`get_parameter_groups()`, `from_scratch_layers`, `not_training` and the state-dict keys (233 of them,
`fc8/f8_3/f8_4/f9_1/f9_2.weight` after the backbone's) are the reference's (revise_net.py:8-117); the
arithmetic is a fixed plan of HIP launches:

  backbone plan -> fc8 -> [f8_3 | f8_4 | x resized] written straight into one 256-channel concat buffer
  (conv epilogues + strided bilinear) -> q|k by ONE 1x1 conv (f9_1 and f9_2 stacked) -> batched q.k GEMM kept
  transposed (P[n][j][i] = A[n][i][j]) so the reference's dim=1 softmax is a row softmax -> the three
  normalised maps [cam | pmask | pcam] multiplied by A in one pass -> 4 bilinear upsamples.

`get_norm_cam_d` is no-grad in the reference (revise_net.py:32): cam_rv / pmask_rv / pcam_rv receive gradient
only through A (-> f9 -> f8_3/f8_4 -> conv4/conv5 taps -> backbone); fc8 trains only through `cam`.

In the split precisions (bf16x3 / fp16x3) and in fp32 the RFM heads compute in f32: the concat feature F, q | k, the affinity matrix and
their gradients are f32 tensors and `f8_3` / `f8_4` / `f9_1` / `f9_2` run on the exact-f32 MFMA kernels with the f32 master weights
(0.4 GF per tile of a 556 GF step); the split taps conv4 / conv5 are widened on the way in, the tap gradients narrowed on the way out.
In the bf16 / fp16 models the heads stay in the storage type (`heads_f32 = False`).  Round 3's review suspected the 16-bit rounding of
F / q / k behind the bf16 stage-3 gradients being 21 % off the CPU oracle (`f8_4.weight`, BASELINE configs[3]) and asked for f32 heads
there too; built and measured (tools/scratch/rfm_heads_ab.py, profiles/r04_rfm_heads_f32_ab.txt): the per-tensor errors do not move
(bf16 worst 0.215 vs 0.196, median 0.087 vs 0.088; fp16 0.240 / 0.111 either way) and the step gets 3.5 % slower.  The error is made
upstream and amplified by the loss itself: `max_onehot` and the two top-k selections (revise_pseudo_labels.py:115-130,268-282) are
discontinuous in the `*_rv` maps, so any perturbation of the taps above ~1e-5 changes WHICH elements carry gradient -- bf16x3 (taps
3e-5 off) is at 5.8 %, fp16x3 (5e-6) at 1.1e-3, the exact-f32 path at 9.8e-4.  The parity-grade stage-3 path is therefore fp16x3 / fp32.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from .arena import ParamArena
from .ops import ConvSpec
from .seg_model import ResNet38dSeg

Tensor = torch.Tensor
FCAT = 256  # concat buffer channels: [f8_3 (64) | f8_4 (128) | x_s (3) | zero pad (61)]


class Net(ResNet38dSeg):
    def __init__(self, num_classes: int = 4, precision: str = "bf16"):
        super().__init__(classes=num_classes, precision=precision)
        self.f8_3 = nn.Conv2d(512, 64, 1, bias=False)
        self.f8_4 = nn.Conv2d(1024, 128, 1, bias=False)
        self.f9_1 = nn.Conv2d(192 + 3, 192, 1, bias=False)
        self.f9_2 = nn.Conv2d(192 + 3, 192, 1, bias=False)
        nn.init.kaiming_normal_(self.f8_3.weight)
        nn.init.kaiming_normal_(self.f8_4.weight)
        nn.init.xavier_uniform_(self.f9_1.weight, gain=4)
        nn.init.xavier_uniform_(self.f9_2.weight, gain=4)
        # RFM heads (F, q | k, affinity, their gradients) in f32 inside the bf16 / fp16 models (module docstring: measured, no parity gain,
        # +3.5 % step time -> off).  The split precisions and fp32 always run them in f32.
        self.heads_f32 = False
        self.from_scratch_layers = [self.f8_3, self.f8_4, self.f9_1, self.f9_2, self.fc8]
        self.not_training = [self.conv1a, self.b2, self.b2_1, self.b2_2]
        self.train(True)

    # ------------------------------------------------------------------ reference API
    def get_parameter_groups(self):
        """revise_net.py:98-117: (pretrained W, pretrained b, scratch W, scratch b)."""
        groups = ([], [], [], [])
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.GroupNorm)):
                if m.weight.requires_grad:
                    (groups[2] if m in self.from_scratch_layers else groups[0]).append(m.weight)
                if m.bias is not None and m.bias.requires_grad:
                    (groups[3] if m in self.from_scratch_layers else groups[1]).append(m.bias)
        return groups

    def get_norm_cam_d(self, cam: Tensor) -> Tensor:
        """NCHW f32 -> NCHW f32 (revise_net.py:29-41)."""
        n, c, h, w = cam.shape
        out = torch.empty((n, c, h, w), device=cam.device, dtype=torch.float32)
        ops.norm_cam(cam.float().contiguous(), "nchw", out, (c * h * w, h * w, 1), 0)
        return out

    def forward(self, x: Tensor, pmask: Tensor, pcam: Tensor):
        params = [p for _, p in self.trainable_conv_params()]
        if torch.is_grad_enabled() and params:
            return _RFMFunction.apply(self, x, pmask, pcam, *params)
        drop = self.sample_dropout(x.shape[0], x.device) if self.training else {}
        outs, _ = self.rfm_forward(x, pmask, pcam, save=False, drop=drop)
        return outs

    # ------------------------------------------------------------------ packed f9 weights
    def _w9(self, transposed: bool) -> Tensor:
        """f9_1 and f9_2 stacked as one [384][256] W_fwd (columns re-ordered to the concat buffer's channel
        order and zero padded), or its [256][384] dgrad layout."""

        def make_fwd():
            with torch.no_grad():
                rows = []
                for conv in (self.f9_1, self.f9_2):
                    w = conv.weight.detach().reshape(192, 195).float()
                    rows.append(torch.cat([w[:, 3:67], w[:, 67:195], w[:, 0:3], w.new_zeros(192, FCAT - 195)], dim=1))
                w9 = torch.cat(rows, dim=0)
                return (w9 if self._heads32() else w9.to(self.compute_dtype)).contiguous()

        wf = self._cached("w9f" + ("32" if self._heads32() else ""), (self.f9_1.weight, self.f9_2.weight), make_fwd)
        if not transposed:
            return wf

        def make_t():
            out = torch.empty((FCAT, 384), device=wf.device, dtype=wf.dtype)
            ops.weight_transpose(wf, out, 384, 1, FCAT)
            return out

        return self._cached("w9d" + ("32" if self._heads32() else ""), (self.f9_1.weight, self.f9_2.weight), make_t)

    def _heads32(self) -> bool:
        return self.heads_f32 or self.split or self.precision == "fp32"

    @staticmethod
    def _unpack_w9_grad(dw9: Tensor):
        """[384, 256] f32 gradient of the packed weight -> gradients of f9_1.weight / f9_2.weight [192,195,1,1]."""
        outs = []
        for r in (dw9[:192], dw9[192:]):
            outs.append(torch.cat([r[:, 192:195], r[:, 0:64], r[:, 64:192]], dim=1).reshape(192, 195, 1, 1))
        return outs

    # ------------------------------------------------------------------ forward plan
    def rfm_forward(self, x: Tensor, pmask: Tensor, pcam: Tensor, save: bool, drop: Optional[Dict[str, Tensor]] = None):
        drop = drop or {}
        x = x.contiguous().float()
        n, _, H, W = x.shape
        C = self.classes
        head = self.fc8.weight.detach().reshape(C, 4096) if (self.fuse_head and not save and not drop) else None  # stage-4 inference: see run_backbone
        feats, saved = self.run_backbone(x, save=save, drop=drop, head=head)
        conv4, conv5 = feats["conv4"], feats["conv5"]
        g1, g2 = conv5.shape[1:3]
        P = g1 * g2
        h32 = self._heads32()
        dev, dt, kw = x.device, (torch.float32 if h32 else self.compute_dtype), dict(opts=self.launch)  # dt: the heads' dtype (module docstring)
        if "cam" in feats:
            cam_lr = feats["cam"]
        else:  # fc8 on dropout7(conv6)
            cam_lr = torch.empty((n, g1, g2, C), device=dev, dtype=torch.float32)
            conv6 = self.act_to_f32(feats["conv6"]) if self.split else feats["conv6"]
            ops.fc8_fwd(conv6, self.fc8.weight.detach().reshape(C, 4096), drop.get("dropout7"), cam_lr)
            del conv6
        # concat feature (revise_net.py:61-66), in f32 from the widened taps
        c4f, c5f = (self.act_to_f32(conv4), self.act_to_f32(conv5)) if h32 else (conv4, conv5)
        F = torch.zeros((n, g1, g2, FCAT), device=dev, dtype=dt)
        ops.conv2d_fwd(ConvSpec(512, 64, 1), c4f, self.w_fwd(self.f8_3, "f8_3", f32=h32), out_act=F[..., 0:64], **kw)
        ops.conv2d_fwd(ConvSpec(1024, 128, 1), c5f, self.w_fwd(self.f8_4, "f8_4", f32=h32), out_act=F[..., 64:192], **kw)
        ops.bilinear_fwd(x, "nchw", F[..., 192:195], "nhwc", True)
        # q | k (revise_net.py:69-71)
        QK = torch.empty((n, g1, g2, 384), device=dev, dtype=dt)
        ops.conv2d_fwd(ConvSpec(FCAT, 384, 1), F, self._w9(False), out_raw=QK, **kw)
        # transposed affinity: S[b][j][i] = sum_c k[b,j,c] * q[b,i,c]; softmax over i (= dim 1 of A)
        q, k = QK[..., :192], QK[..., 192:]
        Pm = torch.empty((n, P, P), device=dev, dtype=torch.float32)
        ops.bgemm(k, q, Pm, n, P, P, 192, (P * 384, 384, 1), (P * 384, 1, 384), (P * P, P, 1))
        ops.softmax_rows_(Pm, n * P, P)
        # normalised maps, pixel-major [cam | pmask | pcam]
        V = torch.empty((n, P, 3 * C), device=dev, dtype=torch.float32)
        ops.norm_cam(cam_lr, "nhwc", V, (P * 3 * C, 1, 3 * C), 0)
        for j, src in enumerate((pmask, pcam)):
            src = src.to(dev).float().contiguous()
            hs, ws = src.shape[-2:]
            tmp = torch.empty((n, hs, ws, C), device=dev, dtype=torch.float32)
            ops.norm_cam(src, "nchw", tmp, (hs * ws * C, 1, C), 0)
            ops.bilinear_fwd(tmp, "nhwc", V.view(n, g1, g2, 3 * C)[..., (j + 1) * C:(j + 2) * C], "nhwc", True)
        R = torch.empty((n, P, 3 * C), device=dev, dtype=torch.float32)
        ops.rfm_apply(Pm, V, R)
        # 4 upsamples (revise_net.py:78-86)
        outs = []
        Rg = R.view(n, g1, g2, 3 * C)
        for src in (cam_lr, Rg[..., 0:C], Rg[..., C:2 * C], Rg[..., 2 * C:3 * C]):
            o = torch.empty((n, C, H, W), device=dev, dtype=torch.float32)
            ops.bilinear_fwd(src, "nhwc", o, "nchw", True)
            outs.append(o)
        ctx = None
        if save:
            ctx = dict(saved=saved, F=F, QK=QK, Pm=Pm, V=V, R=R, conv4=c4f, conv5=c5f, drop7=drop.get("dropout7"), hw=(H, W), g=(g1, g2))
        return tuple(outs), ctx

    # ------------------------------------------------------------------ reverse plan
    def rfm_backward(self, ctx, d_outs, grads: Dict[str, Tensor], after_unit=None, wgrad_stream=None) -> None:
        """d_outs = gradients of (cam, cam_rv, pmask_rv, pcam_rv) (NCHW f32, or None).  grads: name -> f32 buffer
        [cout][kh][kw][cin] that the weight gradients are ACCUMULATED into (caller zeroes); 'f9' maps to the packed
        [384,1,1,256] buffer (see _unpack_w9_grad)."""
        saved, F, QK, Pm, V, R = ctx["saved"], ctx["F"], ctx["QK"], ctx["Pm"], ctx["V"], ctx["R"]
        self.refresh_dgrad_weights()  # one launch for every data-gradient weight layout, the heads' included
        n, C = saved.n, self.classes
        g1, g2 = ctx["g"]
        P = g1 * g2
        dev, dt, kw, h32 = F.device, F.dtype, dict(opts=self.launch), F.dtype == torch.float32  # (f32 heads: see the module docstring)
        d_cam, d_rvs = d_outs[0], d_outs[1:]
        g_taps = {}
        if any(d is not None for d in d_rvs):
            dR = torch.zeros((n, P, 3 * C), device=dev, dtype=torch.float32)
            dRg = dR.view(n, g1, g2, 3 * C)
            for j, d in enumerate(d_rvs):
                if d is not None:
                    ops.bilinear_bwd(d.contiguous(), "nchw", dRg[..., j * C:(j + 1) * C], "nhwc", True)
            ops.affinity_softmax_bwd_(Pm, dR, V, R)  # Pm now holds dS[b][j][i]
            dS = Pm
            q, k = QK[..., :192], QK[..., 192:]
            dQK = torch.empty_like(QK)
            # dq[b,i,c] = sum_j dS[b,j,i] k[b,j,c] ; dk[b,j,c] = sum_i dS[b,j,i] q[b,i,c]
            ops.bgemm(dS, k, dQK[..., :192], n, P, 192, P, (P * P, 1, P), (P * 384, 384, 1), (P * 384, 384, 1))
            ops.bgemm(dS, q, dQK[..., 192:], n, P, 192, P, (P * P, P, 1), (P * 384, 384, 1), (P * 384, 384, 1))
            spec9 = ConvSpec(FCAT, 384, 1)
            if "f9" in grads:
                ops.conv2d_wgrad(spec9, F, dQK, grads["f9"], **kw)
            dFm = torch.empty_like(F)
            ops.conv2d_dgrad(spec9, dQK, self._w9(True), (g1, g2), mask_src=F, out=dFm, **kw)
            for name, conv, lo, hi, tapname, cin in (("f8_3", self.f8_3, 0, 64, "conv4", 512), ("f8_4", self.f8_4, 64, 192, "conv5", 1024)):
                spec = ConvSpec(cin, hi - lo, 1)
                dy = dFm[..., lo:hi]
                if f"{name}.weight" in grads:
                    ops.conv2d_wgrad(spec, ctx[tapname], dy, grads[f"{name}.weight"], **kw)
                gt = torch.empty((n, g1, g2, cin), device=dev, dtype=dt)
                ops.conv2d_dgrad(spec, dy, self.w_dgrad(conv, name, f32=h32), (g1, g2), out_raw=gt, **kw)
                g_taps[tapname] = self.act_from_f32(gt) if h32 else gt  # narrowed to the backbone's storage format: the reverse plan adds it to the unit's input gradient
        dw8 = grads["fc8.weight"].view(C, 4096) if "fc8.weight" in grads else torch.zeros((C, 4096), device=dev)
        if d_cam is None:
            d_cam = torch.zeros((n, C) + tuple(ctx["hw"]), device=dev)
        g_x7 = self.head_backward(saved.conv6, ctx["drop7"], d_cam, dw8)
        if after_unit is not None:
            after_unit("heads")
        self.backward_backbone(saved, g_x7, grads, g_taps=g_taps, after_unit=after_unit, wgrad_stream=wgrad_stream)

    def new_grad_buffers(self, device) -> Dict[str, Tensor]:
        out = {}
        for name, p in self.trainable_conv_params():
            if name.startswith("f9_"):
                continue
            cout, cin, kh, kw = p.shape
            out[name] = torch.zeros((cout, kh, kw, cin), device=device, dtype=torch.float32)
        if self.f9_1.weight.requires_grad or self.f9_2.weight.requires_grad:
            out["f9"] = torch.zeros((384, 1, 1, FCAT), device=device, dtype=torch.float32)
        return out


class _RFMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: Net, x, pmask, pcam, *params):
        drop = model.sample_dropout(x.shape[0], x.device) if model.training else {}
        outs, saved = model.rfm_forward(x, pmask, pcam, save=True, drop=drop)
        ctx.model, ctx.saved_ctx = model, saved
        ctx.names = [n for n, _ in model.trainable_conv_params()]
        if getattr(model, "debug_keep_saved", False):
            model._last_saved = saved["saved"]
        return outs

    @staticmethod
    def backward(ctx, *d_outs):
        model = ctx.model
        saved_ctx, ctx.saved_ctx = ctx.saved_ctx, None
        dev = saved_ctx["F"].device
        d_outs = [d if d is None else d.float() for d in d_outs]
        if model.grad_sink != "arena":  # gradients handed to autograd: fresh buffers every backward (see seg_model._SegFunction)
            grads = model.new_grad_buffers(dev)
            model.rfm_backward(saved_ctx, d_outs, grads)
            res = []
            f9 = model._unpack_w9_grad(grads["f9"].view(384, FCAT)) if "f9" in grads else (None, None)
            for name in ctx.names:
                if name == "f9_1.weight":
                    res.append(f9[0])
                elif name == "f9_2.weight":
                    res.append(f9[1])
                else:
                    res.append(grads[name].permute(0, 3, 1, 2))
            return (None, None, None, None) + tuple(res)
        arena = ParamArena.of(model)
        arena.bind_param_grads()
        red = arena.reducer
        if red is not None:
            inv = 1.0 / torch.distributed.get_world_size(red.group)
            d_outs = [d if d is None else d * inv for d in d_outs]
            red.begin_step()

        def after(name):
            if name == "heads":  # fc8 / f8_3 / f8_4 / f9 gradients are final: unpack f9 into its arena slots
                arena.heads_done()
                if red is not None:
                    for nm in ("fc8", "f8_3", "f8_4", "f9_1", "f9_2"):
                        red.on_unit_done(nm)
            elif red is not None:
                red.on_unit_done(name)

        model.rfm_backward(saved_ctx, d_outs, arena.grads, after_unit=after, wgrad_stream=arena.side_stream() if model.overlap_wgrad else None)
        if red is not None:
            red.finish()
        return (None, None, None, None) + (None,) * len(ctx.names)
