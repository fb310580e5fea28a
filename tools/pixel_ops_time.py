"""HIP-event times of the per-pixel kernels at the training / stage-2 shapes (bs = 64, 224 x 224, 3 classes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops
if os.environ.get("PISTOSEG_HIP_DEBUG_LIB"): _lib.use_debug_library(True)
D = torch.device("cuda:0")
def t(name, fn, mb):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:34s} {us:7.1f} us  {mb / us:5.2f} TB/s ({mb:.0f} MB)")
n, c, s = 64, 3, 224
logits = torch.randn(n, c, s, s, device=D); tgt = torch.randint(0, c + 1, (n, s, s), device=D)
t("softmax_ce fwd+bwd", lambda: ops.softmax_ce(logits, tgt, c, want_grad=True), (2 * n * c * s * s * 4 + n * s * s * 8) / 1e6)
t("argmax_mask plain softmax", lambda: ops.argmax_mask(logits, mode=_lib.PS_MASK_PLAIN, softmax_first=True), (n * c * s * s * 4 + n * s * s) / 1e6)
lab = torch.tensor([[1., 1., 0.]] * n, device=D); tissue = torch.full((n, s, s), 255, dtype=torch.uint8, device=D)
t("argmax_mask fill + entropy", lambda: ops.argmax_mask(logits, mode=_lib.PS_MASK_FILL, label=lab, tissue=tissue, want_entropy=True), (n * c * s * s * 4 + n * s * s * 6) / 1e6)
cam = torch.randn(n, 28, 28, c, device=D); up = torch.empty(n, c, s, s, device=D)
t("bilinear_fwd 28->224 (ac)", lambda: ops.bilinear_fwd(cam, "nhwc", up, "nchw", True), n * c * s * s * 4 / 1e6)
small = torch.empty(n, c, 32, 32, device=D)
t("bilinear_fwd 224->32", lambda: ops.bilinear_fwd(logits, "nchw", small, "nchw", False), n * c * s * s * 4 / 1e6)
dc = torch.empty(n, 28, 28, c, device=D)
t("bilinear_bwd 224->28 (ac)", lambda: ops.bilinear_bwd(logits, "nchw", dc, "nhwc", True), n * c * s * s * 4 / 1e6)
img = torch.randn(64, 3, 224, 224, device=D); view = torch.empty_like(img)
for hflip, k in ((False, 0), (True, 2), (False, 1), (True, 3)):
    t(f"d4_view hflip={int(hflip)} k={k}", lambda: ops.d4_view(img, view, hflip, k, inverse=False, accumulate=False), 2 * img.numel() * 4 / 1e6)
