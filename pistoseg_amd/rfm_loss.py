"""The stage-3 loss block on MI355X: `loss_cls + loss_rfm + loss_ecr` of the reference's `train_epoch`
(revise_pseudo_labels.py:253-282) with its gradient w.r.t. the four network outputs, as HIP reductions
(csrc/rfm_ops.hip): global-average-pool + multilabel soft margin, adaptive min pooling (channel max ->
k-smallest by radix select), masked L1, and the two ECR terms (max_norm*label -> bilinear -> max_onehot ->
|.| -> top-k largest by radix select).  Nothing here builds an autograd graph; `rfm_losses` returns the
loss scalars (1-element device tensors) and, if asked, d loss / d (cam, cam_rv, pmask_rv, pcam_rv).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def _ecr_reference(p: Tensor, label: Tensor, hw) -> Tensor:
    """max_norm(p) * label, channel 0 <- 1 - max fg, bilinear (align_corners=True) to the image size
    (revise_pseudo_labels.py:268-274).  p: [N,C,h,w] f32 -> [N,C,H,W] f32."""
    n, c, h, w = p.shape
    low = torch.empty((n, c, h, w), device=p.device, dtype=torch.float32)
    ops.norm_cam(p, "nchw", low, (c * h * w, h * w, 1), 1, label)
    up = torch.empty((n, c, hw[0], hw[1]), device=p.device, dtype=torch.float32)
    ops.bilinear_fwd(low, "nchw", up, "nchw", True)
    return up


def rfm_losses(outputs, pmask: Tensor, pcam: Tensor, label: Tensor, want_grad: bool, grad_scale: float = 1.0, deterministic: Optional[bool] = None):
    """outputs = (cam, cam_rv, pmask_rv, pcam_rv), NCHW f32 on the device; pmask / pcam carry the zero background
    channel; label: [N, C] (or [N,C,1,1]) with label[:, 0] = 1.
    Returns ((loss, loss_cls, loss_rfm, loss_ecr), grads or None) with grads = (d_cam, d_cam_rv, d_pmask_rv, d_pcam_rv).
    deterministic: the top-k backwards take elements tied at the threshold in index order (the caller's setting; None = ops' default,
    i.e. torch.use_deterministic_algorithms, revise_pseudo_labels.py:140-146)."""
    cam, cam_rv, pmask_rv, pcam_rv = [t.contiguous() for t in outputs]
    n, c, H, W = cam.shape
    dev = cam.device
    label = label.reshape(n, c).to(dev, torch.float32).contiguous()
    pmask = pmask.to(dev, torch.float32).contiguous()
    pcam = pcam.to(dev, torch.float32).contiguous()
    z = lambda: torch.zeros(1, device=dev, dtype=torch.float32)
    l_cls, l_rfm, l_ecr = z(), z(), z()
    d_cam = d_cam_rv = d_pm = d_pc = None
    if want_grad:
        d_cam, d_cam_rv, d_pm, d_pc = (torch.zeros_like(t) for t in (cam, cam_rv, pmask_rv, pcam_rv))

    # ---- loss_cls = multilabel_soft_margin(GAP(cam)[:,1:], label[:,1:]) + adaptive_min_pooling((cam_rv*label)[:,1:])
    gp = ops.gap(cam)
    dgap = ops.softmargin(gp, label, l_cls, accumulate=False, want_grad=want_grad, grad_scale=grad_scale)
    if want_grad:
        ops.gap_bwd(dgap, d_cam)
    m, arg = ops.chmax(cam_rv, label)
    k = (H * W) // 4
    thr, take, sums = ops.topk_select(m, k, largest=False, relu=True)
    ops.sum_scaled(sums, 1.0 / (k * n), l_cls, accumulate=True)
    if want_grad:
        ops.minpool_bwd(m, arg, label, thr, take, d_cam_rv, grad_scale / (k * n), deterministic=deterministic)

    # ---- loss_rfm = mean |pmask_rv*label - pcam_rv*label| over the foreground channels
    ops.l1_masked(pmask_rv, pcam_rv, label, l_rfm, accumulate=False, da=d_pm, db=d_pc, grad_scale=grad_scale)

    # ---- loss_ecr: top-k (k = int(4*H*W*0.2), the 4 is hard-coded in the reference) of |max_onehot(ref) - rv*label|
    k2 = int(4 * H * W * 0.2)
    assert k2 <= c * H * W
    for ref_src, rv, drv in ((pmask, pcam_rv, d_pc), (pcam, pmask_rv, d_pm)):
        ref = _ecr_reference(ref_src, label, (H, W))
        t = torch.empty_like(rv)
        ops.ecr_tensor(ref, rv, label, t)
        thr, take, sums = ops.topk_select(t.view(n, -1), k2, largest=True)
        ops.sum_scaled(sums, 1.0 / (k2 * n), l_ecr, accumulate=True)
        if want_grad:
            ops.ecr_bwd(ref, rv, label, t, thr, take, drv, grad_scale / (k2 * n), deterministic=deterministic)
    total = l_cls + l_rfm + l_ecr
    grads = (d_cam, d_cam_rv, d_pm, d_pc) if want_grad else None
    return (total, l_cls, l_rfm, l_ecr), grads


class _RFMLossBlock(torch.autograd.Function):
    """The whole loss block as one autograd node: forward = the fused reductions with their gradients, backward hands them out."""

    @staticmethod
    def forward(ctx, cam, cam_rv, pmask_rv, pcam_rv, pmask, pcam, label, deterministic):
        (total, l_cls, l_rfm, l_ecr), grads = rfm_losses((cam, cam_rv, pmask_rv, pcam_rv), pmask, pcam, label, want_grad=True,
                                                          deterministic=deterministic)
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(l_cls, l_rfm, l_ecr)
        return total.reshape(()), l_cls.reshape(()), l_rfm.reshape(()), l_ecr.reshape(())

    @staticmethod
    def backward(ctx, g_total, *_unused):
        # (g_total is the 1.0 of `l.backward()`, or a loss scale: applied to the four maps in place -- they are this node's own buffers)
        grads = [g.mul_(g_total) for g in ctx.saved_tensors]
        return (*grads, None, None, None, None)


def rfm_loss_block(cam: Tensor, cam_rv: Tensor, pmask_rv: Tensor, pcam_rv: Tensor, pmask: Tensor, pcam: Tensor, label: Tensor,
                   deterministic: Optional[bool] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """`l, loss_cls, loss_rfm, loss_ecr` of the reference's `train_epoch` (revise_pseudo_labels.py:253-282) from the network's four outputs,
    attached to autograd: `l.backward()` then `optimizer.step()` work as in the script.  The one-call replacement for the script's ~25 eager
    torch statements (two `topk` over 200 k elements per sample among them); pmask / pcam carry the zero background channel and label is
    [N, C] or [N, C, 1, 1] with label[:, 0] = 1, as the script builds them (:238-248).  The three partial losses are for logging (no gradient)."""
    return _RFMLossBlock.apply(cam, cam_rv, pmask_rv, pcam_rv, pmask, pcam, label, deterministic)
