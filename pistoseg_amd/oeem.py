"""OEEM stage 0 -- multi-scale sliding-window CAM: mirror of
`OEEM/classification/network/wide_resnet.py` (the ResNet38-d of stage 0: b7 dilated by 2 instead of 4, a 5632-channel concat
of the conv4 / conv5 / conv6 taps feeding `fc_cls` and `fc_cam`) and of the per-image loop of
`OEEM/classification/prepare_seg_inputs.py:96-138`.

Same state-dict keys as the reference (`fc_cls.weight [C,5632]`, `fc_cls.bias`, `fc_cam.weight [C,5632,1,1]`, `fc_cam.bias` +
the backbone's).  The concat is never materialised: `fc_cam(cat[c4, c5, c6])` is three calls of the narrow-head kernel on
column slices of the weight, accumulating into one CAM buffer.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
from torch import nn

from . import ops, resnet38d
from .sliding import MultiScaleCamAccumulator

Tensor = torch.Tensor
WIDE_UNITS = [u if u[0] != "b7" else ("b7", "bot", 2048, 1024, 4096, 1, 2, 2, 0.5) for u in resnet38d.UNITS]  # wide_resnet.py:129
TAP_CHANNELS = (("conv4", 512), ("conv5", 1024), ("conv6", 4096))  # wide_resnet.py:166-172: 5632 = 512 + 1024 + 4096


class wideResNet(resnet38d.Net):
    def __init__(self, num_class: int = 3, precision: str = "bf16"):
        super().__init__(precision=precision, units=WIDE_UNITS)
        self.num_class = num_class
        self.fc_cls = nn.Linear(5632, num_class)
        self.fc_cam = nn.Conv2d(5632, num_class, 1, stride=1, padding=0, bias=True)
        self.train(True)

    def _head(self, feats, weight: Tensor, bias: Tensor) -> Tensor:
        """[N,g,g,C] f32 = 1x1 head over cat[conv4, conv5, conv6] (channels-last features), without the concat."""
        c6 = feats["conv6"]
        n, g1, g2, _ = c6.shape
        w = weight.detach().reshape(self.num_class, 5632)
        cam = torch.empty((n, g1, g2, self.num_class), device=c6.device, dtype=torch.float32)
        off = 0
        for i, (name, ch) in enumerate(TAP_CHANNELS):
            ops.fc_head_fwd(feats[name], w[:, off:off + ch], 5632, bias.detach() if i == 0 else None, None, cam, accumulate=i > 0)
            off += ch
        return cam

    @torch.no_grad()
    def forward_cam(self, x: Tensor) -> Tensor:
        """wide_resnet.py:182-186: [N,3,H,W] -> CAM scores [N,C,H/8,W/8] f32 (NCHW like the reference)."""
        feats, _ = self.run_backbone(x, save=False)
        return self._head(feats, self.fc_cam.weight, self.fc_cam.bias).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """wide_resnet.py:174-180: fc_cls(flatten(avgpool(features))).  The pool commutes with the linear head, so this is the
        spatial mean of the 1x1 head's map (inference only: stage-0 training is not part of the hot path)."""
        feats, _ = self.run_backbone(x, save=False)
        m = self._head(feats, self.fc_cls.weight, self.fc_cls.bias).permute(0, 3, 1, 2).contiguous()
        return ops.gap(m)


Net = wideResNet  # alias in the style of the repo's other mirrors


@torch.no_grad()
def image_cam_32x32(net: wideResNet, scaled_im_list: Sequence[Tensor], scaled_position_list: Sequence[Sequence[Tuple[int, int]]], scales: Sequence[float],
                    image_wh: Tuple[int, int], side_length: int, batch_size: int = 64) -> Tensor:
    """prepare_seg_inputs.py:96-138 for one image: per scale the crops [K,3,side,side] go through `forward_cam`, the scores are
    resized to the crop size (bilinear, align_corners=False), summed into the scale's canvas with a coverage counter, normalised,
    resized to the image size and averaged over the scales; the result is reduced to 32 x 32 (f64, what the reference np.save()s).
    image_wh = (w, h) = orig_img.shape[:2] in the reference's naming."""
    dev = next(net.parameters()).device
    w, h = int(image_wh[0]), int(image_wh[1])
    acc = MultiScaleCamAccumulator(net.num_class, (w, h), dev)
    for s, scale in enumerate(scales):
        w_, h_ = int(w * scale), int(h * scale)
        ix, iy = min(side_length, w_), min(side_length, h_)  # interpolatex / interpolatey (:103-109)
        crops = scaled_im_list[s]
        cams: List[Tensor] = []
        for b in range(0, crops.shape[0], batch_size):
            scores = net.forward_cam(crops[b:b + batch_size].to(dev))
            out = torch.empty((scores.shape[0], net.num_class, ix, iy), device=dev, dtype=torch.float32)
            ops.bilinear_fwd(scores, "nchw", out, "nchw", False)
            cams.append(out)
        acc.add_scale(torch.cat(cams, 0), scaled_position_list[s], (w_, h_))
    return acc.result((32, 32))
