"""`from models.net_cls import NetCLS` (segmentation_test.py:10 -- imported there, never instantiated by any stage script).
models/net_cls.py:8-25: ResNet38-d + dropout + fc8(4096 -> 4) and the spatial mean of the foreground CAM channels."""
import _pistoseg_compat  # noqa: F401
import torch

from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg


class NetCLS(ResNet38dSeg):
    def __init__(self, precision: str = "bf16"):
        super().__init__(classes=4, precision=precision)

    @torch.no_grad()
    def forward(self, x):
        drop = self.sample_dropout(x.shape[0], x.device) if self.training else {}
        feats, _ = self.run_backbone(x, save=False, drop=drop)
        n, g1, g2, _ = feats["conv6"].shape
        cam = torch.empty((n, g1, g2, self.classes), device=x.device, dtype=torch.float32)
        ops.fc8_fwd(feats["conv6"], self.fc8.weight.detach().reshape(self.classes, 4096), drop.get("dropout7"), cam)
        return ops.gap(cam.permute(0, 3, 1, 2).contiguous())[:, 1:].squeeze()
