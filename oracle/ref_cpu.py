"""CPU oracle for the PistoSeg segmentation hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state-dict in, tensors out) restatement in plain
torch-CPU fp32 of the arithmetic on the reference's hot path.  It is the *checker*: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The
product path (`pistoseg_amd/`) never routes through it and fails loudly when the HIP library
is missing.

Parity pin: every function below is checked against golden vectors minted by importing the
reference's own modules in the build container (`oracle/make_golden.py` ->
`tests/golden/*.npz`, test: `tests/test_oracle_golden.py`).  Third-party arithmetic that is
not under /root/reference (smp DiceLoss, ttach d4 TTA) is restated from its public definition
and is labelled "parity unpinned" where it appears.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5  # nn.BatchNorm2d default, models/resnet38d.py:14,18

# --------------------------------------------------------------------------------------
# Architecture table (models/resnet38d.py:119-146).  One row per residual unit:
#   name, kind, cin, cmid, cout, stride, first_dilation, dilation, dropout_p
# kind "res": ResBlock (resnet38d.py:6-51); kind "bot": ResBlock_bot (resnet38d.py:53-101).
# --------------------------------------------------------------------------------------
BLOCKS: List[Tuple[str, str, int, int, int, int, int, int, float]] = [
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
# Feature taps returned by forward_as_dict (resnet38d.py:172,180,184): x_bn_relu of these units.
TAPS = {"b4": "conv3", "b5": "conv4", "b6": "conv5"}
# revise_net.py:27 -- frozen units of the RFM net (plus every BatchNorm, resnet38d.py:206-211).
RFM_NOT_TRAINING = ("conv1a", "b2", "b2_1", "b2_2")


def _block_same_shape(cin: int, cout: int, stride: int) -> bool:
    return cin == cout and stride == 1


def state_dict_spec(num_classes: Optional[int] = None, rfm_heads: bool = False):
    """Ordered (key, shape) list in the reference's state-dict key order.

    resnet38d.Net has 228 keys; revise_net.Net adds fc8, f8_3, f8_4, f9_1, f9_2
    (revise_net.py:13-19) -> 233.  `num_classes` adds fc8 only (the seg head of this build).
    """
    spec: List[Tuple[str, Tuple[int, ...]]] = [("conv1a.weight", (64, 3, 3, 3))]

    def bn(prefix: str, c: int):
        spec.extend(
            [
                (f"{prefix}.weight", (c,)),
                (f"{prefix}.bias", (c,)),
                (f"{prefix}.running_mean", (c,)),
                (f"{prefix}.running_var", (c,)),
                (f"{prefix}.num_batches_tracked", ()),
            ]
        )

    for name, kind, cin, cmid, cout, stride, fdil, dil, _p in BLOCKS:
        if kind == "res":
            bn(f"{name}.bn_branch2a", cin)
            spec.append((f"{name}.conv_branch2a.weight", (cmid, cin, 3, 3)))
            bn(f"{name}.bn_branch2b1", cmid)
            spec.append((f"{name}.conv_branch2b1.weight", (cout, cmid, 3, 3)))
            if not _block_same_shape(cin, cout, stride):
                spec.append((f"{name}.conv_branch1.weight", (cout, cin, 1, 1)))
        else:
            bn(f"{name}.bn_branch2a", cin)
            spec.append((f"{name}.conv_branch2a.weight", (cout // 4, cin, 1, 1)))
            bn(f"{name}.bn_branch2b1", cout // 4)
            spec.append((f"{name}.conv_branch2b1.weight", (cout // 2, cout // 4, 3, 3)))
            bn(f"{name}.bn_branch2b2", cout // 2)
            spec.append((f"{name}.conv_branch2b2.weight", (cout, cout // 2, 1, 1)))
            spec.append((f"{name}.conv_branch1.weight", (cout, cin, 1, 1)))
    bn("bn7", 4096)
    if num_classes is not None:
        spec.append(("fc8.weight", (num_classes, 4096, 1, 1)))
    if rfm_heads:
        spec.append(("f8_3.weight", (64, 512, 1, 1)))
        spec.append(("f8_4.weight", (128, 1024, 1, 1)))
        spec.append(("f9_1.weight", (192, 195, 1, 1)))
        spec.append(("f9_2.weight", (192, 195, 1, 1)))
    return spec


def make_state_dict(num_classes: Optional[int] = 4, rfm_heads: bool = True, seed: int = 42) -> Dict[str, Tensor]:
    """Deterministic synthetic weights (no checkpoint is available offline).

    Every tensor is drawn from its own frozen legacy stream `RandomState(crc32(key) ^ seed)` so
    that the GPU box regenerates bit-identical weights without the reference being present.
    Conv weights are He-scaled; BN statistics/affine are randomised so that eval-mode BN is a
    non-trivial per-channel affine map.  The residual-branch output convs are damped so that the
    38-layer residual stream stays O(1) (finite in bf16).
    """
    sd: Dict[str, Tensor] = {}
    for key, shape in state_dict_spec(num_classes, rfm_heads):
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.int64)
            continue
        leaf = key.rsplit(".", 1)[1]
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt(2.0 / fan_in)
            if ".conv_branch2b1." in key and shape[2] == 3 and not key.startswith(("b6", "b7")):
                std *= 0.5  # last conv of a ResBlock branch
            if ".conv_branch2b2." in key:
                std *= 0.5  # last conv of a bottleneck branch
            if key.startswith("fc8"):
                std = math.sqrt(1.0 / fan_in)
            if key.startswith("f9_"):
                std = 0.35 * math.sqrt(1.0 / fan_in)  # keeps q.k affinity logits O(1)
            arr = rs.standard_normal(shape).astype(np.float32) * np.float32(std)
        elif leaf == "weight":
            arr = rs.uniform(0.6, 1.4, shape).astype(np.float32)
        elif leaf == "bias":
            arr = rs.uniform(-0.2, 0.2, shape).astype(np.float32)
        elif leaf == "running_mean":
            arr = rs.uniform(-0.3, 0.3, shape).astype(np.float32)
        elif leaf == "running_var":
            arr = rs.uniform(0.6, 1.6, shape).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(key)
        sd[key] = torch.from_numpy(arr)
    return sd


# --------------------------------------------------------------------------------------
# Backbone (models/resnet38d.py)
# --------------------------------------------------------------------------------------
def _bn_relu(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """Eval-mode BatchNorm2d followed by ReLU.  BN is *always* eval on this path, even in
    training: Net.train() forces every BatchNorm2d to .eval() (resnet38d.py:206-211)."""
    y = F.batch_norm(
        x,
        sd[f"{prefix}.running_mean"],
        sd[f"{prefix}.running_var"],
        sd[f"{prefix}.weight"],
        sd[f"{prefix}.bias"],
        training=False,
        eps=BN_EPS,
    )
    return F.relu(y)


def _drop(x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """Dropout2d with an injected per-(n, c) multiplier (already scaled by 1/(1-p)); None = eval.
    Reference: Dropout2d is active in training (resnet38d.py:63,67,85,90; revise_net.py:11,50)."""
    if mask is None:
        return x
    return x * mask.view(mask.shape[0], mask.shape[1], 1, 1)


def _res_unit(sd, name, cin, cmid, cout, stride, fdil, dil, x, collect=None):
    """ResBlock.forward, resnet38d.py:26-48.  The 1x1 shortcut is applied to the *activated*
    input when the shape changes; otherwise the raw input is the identity shortcut."""
    a = _bn_relu(sd, f"{name}.bn_branch2a", x)
    if _block_same_shape(cin, cout, stride):
        shortcut = x
    else:
        shortcut = F.conv2d(a, sd[f"{name}.conv_branch1.weight"], stride=stride)
    h = F.conv2d(a, sd[f"{name}.conv_branch2a.weight"], stride=stride, padding=fdil, dilation=fdil)
    h = _bn_relu(sd, f"{name}.bn_branch2b1", h)
    if collect is not None:
        collect[name] = (a, h)
    h = F.conv2d(h, sd[f"{name}.conv_branch2b1.weight"], padding=dil, dilation=dil)
    return shortcut + h, a


def _bot_unit(sd, name, cin, cout, stride, dil, x, drop1, drop2, collect=None):
    """ResBlock_bot.forward, resnet38d.py:73-98 (shortcut conv always present)."""
    a = _bn_relu(sd, f"{name}.bn_branch2a", x)
    shortcut = F.conv2d(a, sd[f"{name}.conv_branch1.weight"], stride=stride)
    h = F.conv2d(a, sd[f"{name}.conv_branch2a.weight"], stride=stride)
    h1 = _drop(_bn_relu(sd, f"{name}.bn_branch2b1", h), drop1)
    h = F.conv2d(h1, sd[f"{name}.conv_branch2b1.weight"], padding=dil, dilation=dil)
    h2 = _drop(_bn_relu(sd, f"{name}.bn_branch2b2", h), drop2)
    if collect is not None:
        collect[name] = (a, h1, h2)
    h = F.conv2d(h2, sd[f"{name}.conv_branch2b2.weight"])
    return shortcut + h, a


def forward_as_dict(sd: Dict[str, Tensor], x: Tensor, drop: Optional[Dict[str, Tensor]] = None, collect=None, blocks=None) -> Dict[str, Tensor]:
    """Net.forward_as_dict, resnet38d.py:159-188.  `drop` maps
    {'b6.dropout_2b1','b6.dropout_2b2','b7.dropout_2b1','b7.dropout_2b2'} -> [N, C] multipliers.
    `collect` (optional dict) receives every unit's post-ReLU activations (for ReLU-pattern checks in tests)."""
    drop = drop or {}
    out: Dict[str, Tensor] = {}
    x = F.conv2d(x, sd["conv1a.weight"], padding=1)
    for name, kind, cin, cmid, cout, stride, fdil, dil, _p in (BLOCKS if blocks is None else blocks):
        if kind == "res":
            x, a = _res_unit(sd, name, cin, cmid, cout, stride, fdil, dil, x, collect)
        else:
            x, a = _bot_unit(
                sd, name, cin, cout, stride, dil, x, drop.get(f"{name}.dropout_2b1"), drop.get(f"{name}.dropout_2b2"), collect
            )
        if name in TAPS:
            out[TAPS[name]] = a
    out["conv6"] = _bn_relu(sd, "bn7", x)
    return out


def bilinear(x: Tensor, size: Tuple[int, int], align_corners: bool) -> Tensor:
    return F.interpolate(x, size, mode="bilinear", align_corners=align_corners)


def seg_forward(sd: Dict[str, Tensor], x: Tensor, drop: Optional[Dict[str, Tensor]] = None, collect=None) -> Tensor:
    """The build's "ResNet38-d segmentation model" (SURVEY 0.2): the `cam` branch of
    revise_net.Net.forward -- fc8 1x1 conv on dropout7(conv6), bilinear align_corners=True
    upsample to the input size (revise_net.py:50,86).  Equals outputs[0] of revise_forward."""
    drop = drop or {}
    H, W = x.shape[-2:]
    conv6 = forward_as_dict(sd, x, drop, collect)["conv6"]
    if collect is not None:
        collect["conv6"] = (conv6,)
    cam = F.conv2d(_drop(conv6, drop.get("dropout7")), sd["fc8.weight"])
    return bilinear(cam, (H, W), True)


# --------------------------------------------------------------------------------------
# RFM head (models/revise_net.py)
# --------------------------------------------------------------------------------------
def get_norm_cam_d(cam: Tensor) -> Tensor:
    """revise_net.py:29-41, entirely under no_grad: per-(n,c) min/max normalisation, channel 0
    becomes 1 - max over foreground, foreground entries strictly below the per-pixel foreground
    max are zeroed (ties are kept)."""
    with torch.no_grad():
        n, c, h, w = cam.shape
        flat = cam.detach().reshape(n, c, -1)
        lo = flat.min(dim=-1)[0].view(n, c, 1, 1)
        hi = flat.max(dim=-1)[0].view(n, c, 1, 1) + 1e-5
        nrm = (cam.detach() - lo) / (hi - lo)
        fg = nrm[:, 1:]
        nrm[:, 0] = 1 - fg.max(dim=1)[0]
        fg_max = fg.max(dim=1, keepdim=True)[0]
        nrm[:, 1:] = torch.where(fg < fg_max, torch.zeros_like(fg), fg)
    return nrm


def rfm(cam: Tensor, A: Tensor, h: int, w: int) -> Tensor:
    """Net.RFM, revise_net.py:90-96: resize to the feature grid, multiply by the affinity."""
    n = A.shape[0]
    flat = bilinear(cam, (h, w), True).reshape(n, -1, h * w)
    return torch.matmul(flat, A).view(n, -1, h, w)


def affinity(sd: Dict[str, Tensor], x: Tensor, conv4: Tensor, conv5: Tensor) -> Tensor:
    """revise_net.py:61-74: f = cat[x resized, relu(f8_3 conv4), relu(f8_4 conv5)] (195 ch);
    A = softmax(q^T k, dim=1) -- the softmax runs over the ROW index (columns sum to one)."""
    n, _, h, w = conv4.shape
    f3 = F.relu(F.conv2d(conv4, sd["f8_3.weight"]))
    f4 = F.relu(F.conv2d(conv5, sd["f8_4.weight"]))
    xs = bilinear(x, (h, w), True)
    f = torch.cat([xs, f3, f4], dim=1)
    q = F.conv2d(f, sd["f9_1.weight"]).view(n, -1, h * w)
    k = F.conv2d(f, sd["f9_2.weight"]).view(n, -1, h * w)
    return F.softmax(torch.matmul(q.transpose(1, 2), k), dim=1)


def revise_forward(sd, x, pmask, pcam, drop: Optional[Dict[str, Tensor]] = None):
    """revise_net.Net.forward(x, pmask, pcam) -> (cam, cam_rv, pmask_rv, pcam_rv), revise_net.py:43-88."""
    drop = drop or {}
    H, W = x.shape[-2:]
    d = forward_as_dict(sd, x, drop)
    cam = F.conv2d(_drop(d["conv6"], drop.get("dropout7")), sd["fc8.weight"])
    h, w = cam.shape[-2:]
    cam_n, pmask_n, pcam_n = get_norm_cam_d(cam), get_norm_cam_d(pmask), get_norm_cam_d(pcam)
    A = affinity(sd, x, d["conv4"], d["conv5"])
    pmask_rv = bilinear(rfm(pmask_n, A, h, w), (H, W), True)
    pcam_rv = bilinear(rfm(pcam_n, A, h, w), (H, W), True)
    cam_rv = bilinear(rfm(cam_n, A, h, w), (H, W), True)
    return bilinear(cam, (H, W), True), cam_rv, pmask_rv, pcam_rv


# --------------------------------------------------------------------------------------
# Losses (revise_pseudo_labels.py, models/segmentation_module.py)
# --------------------------------------------------------------------------------------
def adaptive_min_pooling_loss(x: Tensor) -> Tensor:
    """revise_pseudo_labels.py:115-123: channel max, mean of relu over the k = h*w//4 smallest."""
    n, c, h, w = x.shape
    k = h * w // 4
    m = x.max(dim=1)[0].reshape(n, -1)
    small = torch.topk(m, k=k, dim=-1, largest=False)[0]
    return F.relu(small).sum() / (k * n)


def max_onehot(x: Tensor) -> Tensor:
    """revise_pseudo_labels.py:125-130: zero every foreground entry that differs from the
    per-pixel foreground max (returns a new tensor; the reference edits in place)."""
    x = x.clone()
    fg = x[:, 1:]
    fg_max = fg.max(dim=1, keepdim=True)[0]
    x[:, 1:] = torch.where(fg != fg_max, torch.zeros_like(fg), fg)
    return x


def max_norm(p: Tensor, e: float = 1e-5) -> Tensor:
    """revise_pseudo_labels.py:132-138 (dup infer_revise_masks.py:72-78)."""
    n, c, h, w = p.shape
    flat = p.reshape(n, c, -1)
    hi = flat.max(dim=-1)[0].view(n, c, 1, 1)
    lo = flat.min(dim=-1)[0].view(n, c, 1, 1)
    return (p - lo) / (hi - lo + e)


def rfm_losses(outputs, pmask, pcam, label, image_hw):
    """The loss arithmetic of train_epoch, revise_pseudo_labels.py:253-282.

    outputs = (cam, cam_rv, pmask_rv, pcam_rv); pmask/pcam already carry the zero background
    channel (:239-240); label is [N, C, 1, 1] with label[:, 0] = 1 (:242-245).
    Returns (loss, loss_cls, loss_rfm, loss_ecr)."""
    cam, cam_rv, pmask_rv, pcam_rv = outputs
    H, W = image_hw
    label_cam = F.adaptive_avg_pool2d(cam, (1, 1))
    loss_rvmin = adaptive_min_pooling_loss((cam_rv * label)[:, 1:])
    loss_cls = F.multilabel_soft_margin_loss(label_cam[:, 1:], label[:, 1:]) + loss_rvmin

    pmask_rv = pmask_rv * label
    pcam_rv = pcam_rv * label
    loss_rfm = torch.mean(torch.abs(pmask_rv[:, 1:] - pcam_rv[:, 1:]))

    ns, cs, hs, ws = cam.shape
    pm = max_norm(pmask) * label
    pc = max_norm(pcam) * label
    pm = torch.cat([1 - pm[:, 1:].max(dim=1, keepdim=True)[0], pm[:, 1:]], dim=1)
    pc = torch.cat([1 - pc[:, 1:].max(dim=1, keepdim=True)[0], pc[:, 1:]], dim=1)
    pm = bilinear(pm, (H, W), True)
    pc = bilinear(pc, (H, W), True)
    t1 = torch.abs(max_onehot(pm.detach()) - pcam_rv)
    t2 = torch.abs(max_onehot(pc.detach()) - pmask_rv)
    k = int(4 * hs * ws * 0.2)  # hard-coded 4, not cs (:277)
    loss_ecr = torch.mean(torch.topk(t1.reshape(ns, -1), k=k, dim=-1)[0]) + torch.mean(
        torch.topk(t2.reshape(ns, -1), k=k, dim=-1)[0]
    )
    return loss_cls + loss_rfm + loss_ecr, loss_cls, loss_rfm, loss_ecr


def seg_ce_loss(logits: Tensor, target: Tensor, ignore_index: Optional[int]) -> Tensor:
    """SegmentationModule.training_step, models/segmentation_module.py:63-66,101-102:
    CrossEntropyLoss(reduction='none'[, ignore_index=3]) then torch.mean over ALL N*H*W pixels
    (ignored pixels contribute 0 to the sum but still count in the denominator)."""
    if ignore_index is None:
        ce = F.cross_entropy(logits, target, reduction="none")
    else:
        ce = F.cross_entropy(logits, target, reduction="none", ignore_index=ignore_index)
    return ce.mean()


def dice_loss_multiclass(logits: Tensor, target: Tensor, ignore_index: Optional[int], eps: float = 1e-7) -> Tensor:
    """smp.losses.DiceLoss(mode='multiclass', ignore_index=...) as used by
    models/mosaic_module.py:65-68,108 -- PARITY UNPINNED: segmentation-models-pytorch==0.3.0
    (environment.yaml:182) is not vendored, so this follows its published definition
    (from_logits=True, smooth=0, log_loss=False, dims=(0,2)): softmax probabilities (log_softmax
    then exp), one-hot target with ignored pixels masked from both, per-class
    1 - 2*sum(p*t)/max(sum(p+t), eps), classes absent from the target zeroed, mean over classes."""
    n, c = logits.shape[:2]
    p = logits.log_softmax(dim=1).exp().reshape(n, c, -1)
    t = target.reshape(n, -1)
    if ignore_index is not None:
        keep = t != ignore_index
        p = p * keep.unsqueeze(1)
        onehot = F.one_hot((t * keep).long(), c).permute(0, 2, 1) * keep.unsqueeze(1)
    else:
        onehot = F.one_hot(t.long(), c).permute(0, 2, 1)
    onehot = onehot.to(p.dtype)
    inter = (p * onehot).sum(dim=(0, 2))
    card = (p + onehot).sum(dim=(0, 2))
    score = 2.0 * inter / card.clamp_min(eps)
    loss = (1.0 - score) * (onehot.sum(dim=(0, 2)) > 0).to(p.dtype)
    return loss.mean()


# --------------------------------------------------------------------------------------
# CAM / logit -> mask reductions (infer_pseudo_masks.py, infer_revise_masks.py, loss.py)
# --------------------------------------------------------------------------------------
def get_mask_pred_and_entropy(logit: Tensor, tissue: np.ndarray, patch_label: Sequence[int]):
    """infer_pseudo_masks.py:69-87.  Single-label tiles get a constant mask and zero entropy;
    otherwise absent classes are filled with -1e10, softmax over channels, entropy
    -sum p*log(p+1e-10), argmax (first max wins); non-tissue pixels get index len(label)."""
    patch_label = list(patch_label)
    h, w = logit.shape[-2:]
    if sum(patch_label) == 1:
        mask = np.full((h, w), patch_label.index(1))
        entropy = np.zeros_like(mask)
    else:
        z = logit.clone()
        for i, present in enumerate(patch_label):
            if present == 0:
                z[i] = -1e10
        p = torch.softmax(z, dim=0)
        entropy = (-(p * torch.log(p + 1e-10)).sum(dim=0)).numpy()
        mask = torch.argmax(p, dim=0).numpy()
    mask[tissue == 0] = len(patch_label)
    return mask, entropy


def interpolate_tensor(t: Tensor, shape: Tuple[int, int]) -> Tensor:
    """infer_pseudo_masks.py:89-90: bilinear, align_corners=False."""
    return F.interpolate(t.unsqueeze(0), shape, mode="bilinear")[0]


def revise_infer_masks(outputs, label):
    """infer_revise_masks.py:137-143: (X_rv * label)[:, 1:] -> argmax over channels, for
    pmask_rv, pcam_rv, cam_rv (in that order)."""
    _, cam_rv, pmask_rv, pcam_rv = outputs
    return tuple(torch.argmax((t * label)[:, 1:], dim=1) for t in (pmask_rv, pcam_rv, cam_rv))


def confusion_matrix(pred: np.ndarray, gt: np.ndarray, num_class: int) -> np.ndarray:
    """loss.mIoUMask._generate_matrix as *called* (loss.py:16-26,31-33,65): add_batch receives
    (pred, mask) into parameters named (gt_image, pre_image) and forwards them swapped again,
    so rows end up indexed by the ground truth and columns by the prediction; ground-truth
    values >= num_class are dropped."""
    keep = (gt >= 0) & (gt < num_class)
    idx = num_class * gt[keep].astype("int") + pred[keep]
    return np.bincount(idx, minlength=num_class**2).reshape(num_class, num_class)


def miou_from_confusion(cm: np.ndarray) -> Tuple[float, float]:
    """loss.py:35-53: (mIoU with NaN -> 0, frequency-weighted IoU)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        iu = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
        freq = cm.sum(1) / cm.sum()
    miou = np.where(np.isnan(iu), 0, iu).mean()
    fw = (freq[freq > 0] * iu[freq > 0]).sum()
    return float(miou), float(fw)


def logits_to_mask(logits: Tensor, probs: bool = False) -> Tensor:
    """loss.py:55-60: (softmax ->) argmax over channels -> uint8."""
    if not probs:
        logits = F.softmax(logits, dim=1)
    return torch.argmax(logits, dim=1).byte()


# --------------------------------------------------------------------------------------
# Optimisers (utils.py:166-187; models/segmentation_module.py:86-90)
# --------------------------------------------------------------------------------------
class PolyOptimizerOracle(torch.optim.SGD):
    """utils.PolyOptimizer: `SGD.__init__(params, lr, weight_decay)` passes weight_decay as SGD's
    third positional argument, which is *momentum*; the per-group weight_decay given in the
    param-group dicts still applies.  LR is scaled by (1 - step/max_step) ** 0.9 before each step."""

    def __init__(self, params, lr, weight_decay, max_step, power=0.9):
        super().__init__(params, lr, momentum=weight_decay)
        self.global_step = 0
        self.max_step = max_step
        self.power = power
        self._initial_lr = [g["lr"] for g in self.param_groups]

    def step(self, closure=None):
        if self.global_step < self.max_step:
            mult = (1 - self.global_step / self.max_step) ** self.power
            for g, lr0 in zip(self.param_groups, self._initial_lr):
                g["lr"] = lr0 * mult
        super().step(closure)
        self.global_step += 1


def trainable_keys(sd: Dict[str, Tensor], not_training: Sequence[str] = RFM_NOT_TRAINING) -> List[str]:
    """Conv weights that receive gradients after Net.train() (resnet38d.py:191-213 with
    revise_net.py:27): every conv outside the frozen units; no BatchNorm parameter."""
    keys = []
    for k, v in sd.items():
        if v.dim() != 4:
            continue
        unit = k.split(".")[0]
        if unit in not_training:
            continue
        keys.append(k)
    return keys


# --------------------------------------------------------------------------------------
# Sliding-window evaluation.  These restate host-side numpy / torch-builtin code of the
# reference statement by statement.  PINNED: tests/test_oracle_golden.py checks `sliding_window_big_masks` and `multi_scale_cam`
# bit for bit against tests/golden/seg_eval.npz / oeem_ms_cam.npz, which oracle/make_golden_eval.py minted by executing the
# reference's own method bodies (validation_step / validation_epoch_end; the per-image statements of prepare_seg_inputs.py).
# --------------------------------------------------------------------------------------
def sliding_window_big_masks(batches, image_sizes: Dict[str, Tuple[int, int]], num_classes: int = 3):
    """models/segmentation_module.py:127-178 (== segmentation_test.py:141-196): `batches` yields
    (logits [N,C,S,S] f32, names, original_h, original_w); image_sizes[image_idx] = (w, h) (PIL's Image.size).
    Returns {image_idx: (pred_big_mask [h,w,C] f64 -- the SUM over scales --, cnt_big_mask [h,w,1])}."""
    pred_ms: Dict[str, np.ndarray] = {}
    cnt_ms: Dict[str, np.ndarray] = {}
    for output, name_batch, original_h_batch, original_w_batch in batches:
        for j in range(output.shape[0]):
            original_w, original_h = int(original_w_batch[j]), int(original_h_batch[j])
            output_ = output[j][:, :original_h, :original_w]
            probs = torch.softmax(output_, dim=0).numpy().transpose(1, 2, 0)
            name = name_batch[j]
            image_idx = name.split("_")[0]
            scale = float(name.split("_")[1])
            position = (int(name.split("_")[2]), int(name.split("_")[3].split("-")[0]))
            dict_key = f"{image_idx}_{scale}"
            if dict_key not in pred_ms:
                w, h = image_sizes[image_idx]
                w_, h_ = int(w * scale), int(h * scale)
                pred_ms[dict_key] = np.zeros((h_, w_, num_classes))
                cnt_ms[dict_key] = np.zeros((h_, w_, 1))
            pred_ms[dict_key][position[0]:position[0] + output_.shape[1], position[1]:position[1] + output_.shape[2], :] += probs
            cnt_ms[dict_key][position[0]:position[0] + output_.shape[1], position[1]:position[1] + output_.shape[2], :] += 1
    pred_big: Dict[str, np.ndarray] = {}
    cnt_big: Dict[str, np.ndarray] = {}
    for k, mask in pred_ms.items():
        with np.errstate(divide="ignore", invalid="ignore"):
            mask = mask / cnt_ms[k]
        image_idx = k.split("_")[0]
        w, h = image_sizes[image_idx]
        if image_idx not in pred_big:
            pred_big[image_idx] = np.zeros((h, w, num_classes))
            cnt_big[image_idx] = np.zeros((h, w, 1))
        mask = F.interpolate(torch.from_numpy(mask.transpose(2, 0, 1)).unsqueeze(0), (h, w), mode="bilinear")[0].numpy().transpose(1, 2, 0)
        pred_big[image_idx][:, :, :] += mask
        cnt_big[image_idx][:, :, :] += 1
    return {k: (pred_big[k], cnt_big[k]) for k in pred_big}


def big_mask_predictions(big: Dict[str, Tuple[np.ndarray, np.ndarray]], gt: Optional[Dict[str, np.ndarray]] = None, bg_value: int = -1):
    """segmentation_module.py:180-185 + loss.py:55-57 (probs=True): mask_pred /= cnt; argmax over classes (uint8);
    with gt: `mask_pred[mask == 3] = 3` (segmentation_test.py:199-201)."""
    out = {}
    for k, (pred, cnt) in big.items():
        with np.errstate(divide="ignore", invalid="ignore"):
            p = pred / cnt
        m = torch.argmax(torch.from_numpy(p.transpose(2, 0, 1)).unsqueeze(0), dim=1).byte()[0].numpy()
        if gt is not None and bg_value >= 0:
            m = m.copy()
            m[gt[k] == bg_value] = bg_value
        out[k] = m
    return out


def multi_scale_cam(cam_crops_per_scale, positions_per_scale, scaled_sizes, image_wh: Tuple[int, int], num_of_class: int, side_length: int):
    """OEEM/classification/prepare_seg_inputs.py:96-138 for one image, starting from the per-crop CAM scores that
    `F.interpolate(cam_scores, (interpolatex, interpolatey))` produced (:117): sum_cam / sum_counter per scale, resize to (w, h),
    mean over scales, resize to 32 x 32.  image_wh = (w, h) in the reference's naming (orig_img.shape[:2])."""
    w, h = image_wh
    ensemble_cam = np.zeros((num_of_class, w, h))
    for cam_list, position_list, (w_, h_) in zip(cam_crops_per_scale, positions_per_scale, scaled_sizes):
        cam_list = cam_list.numpy()
        sum_cam = np.zeros((num_of_class, w_, h_))
        sum_counter = np.zeros_like(sum_cam)
        for k in range(cam_list.shape[0]):
            y, x = position_list[k][0], position_list[k][1]
            crop = cam_list[k]
            sum_cam[:, y:y + side_length, x:x + side_length] += crop
            sum_counter[:, y:y + side_length, x:x + side_length] += 1
        sum_counter[sum_counter < 1] = 1
        norm_cam = sum_cam / sum_counter
        norm_cam = F.interpolate(torch.unsqueeze(torch.tensor(norm_cam), 0), (w, h), mode="bilinear", align_corners=False).numpy()[0]
        ensemble_cam += norm_cam
    ensemble_cam /= len(scaled_sizes)
    return F.interpolate(torch.unsqueeze(torch.tensor(ensemble_cam), 0), (32, 32), mode="bilinear", align_corners=False).numpy()[0]


def d4_tta(model_fn, image: Tensor) -> Tensor:
    """ttach.SegmentationTTAWrapper(model, d4_transform(), merge_mode='mean') (infer_pseudo_masks.py:96) restated from the
    package's public definition -- third-party ttach==0.0.3, absent here: PARITY UNPINNED.
    d4_transform = Compose([HorizontalFlip(), Rotate90(angles=[0, 90, 180, 270])]); views in itertools.product order."""
    import itertools

    total = None
    for hflip, angle in itertools.product([False, True], [0, 90, 180, 270]):
        x = image.flip(3) if hflip else image                 # HorizontalFlip.apply_aug_image
        x = torch.rot90(x, angle // 90, (2, 3))               # Rotate90.apply_aug_image
        y = model_fn(x)
        y = torch.rot90(y, ((-angle) % 360) // 90, (2, 3))    # deaugment in reverse order: Rotate90.apply_deaug_mask(-angle) ...
        y = y.flip(3) if hflip else y                         # ... then HorizontalFlip.apply_deaug_mask
        total = y if total is None else total + y             # Merger('mean').append
    return total / 8                                          # Merger.result


# --------------------------------------------------------------------------------------
# OEEM stage 0: OEEM/classification/network/wide_resnet.py -- the same ResNet38-d with b7 dilated by 2
# (:129) and two heads over cat[conv4, conv5, conv6] (5632 channels, :166-186).  Pinned by tests/golden/oeem_cam.npz, minted
# by oracle/make_golden_oeem.py from the reference's own Net.forward_cam.
# --------------------------------------------------------------------------------------
WIDE_BLOCKS = [b if b[0] != "b7" else ("b7", "bot", 2048, 1024, 4096, 1, 2, 2, 0.5) for b in BLOCKS]


def wide_state_dict(num_class: int = 3, seed: int = 42) -> Dict[str, Tensor]:
    """Backbone weights of make_state_dict + fc_cls / fc_cam, in the reference's key order (wide_resnet.py:131-139)."""
    sd = make_state_dict(num_classes=None, rfm_heads=False, seed=seed)
    for key, shape in (("fc_cls.weight", (num_class, 5632)), ("fc_cls.bias", (num_class,)), ("fc_cam.weight", (num_class, 5632, 1, 1)),
                       ("fc_cam.bias", (num_class,))):
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        scale = math.sqrt(1.0 / 5632) if key.endswith("weight") else 0.1
        sd[key] = torch.from_numpy(rs.standard_normal(shape).astype(np.float32) * np.float32(scale))
    return sd


def wide_features(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """Net._shared_forward, wide_resnet.py:143-172."""
    d = forward_as_dict(sd, x, blocks=WIDE_BLOCKS)
    return torch.cat([d["conv4"], d["conv5"], d["conv6"]], dim=1)


def wide_forward_cam(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """Net.forward_cam, wide_resnet.py:182-186."""
    return F.conv2d(wide_features(sd, x), sd["fc_cam.weight"], sd["fc_cam.bias"])


def wide_forward_cls(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """Net.forward, wide_resnet.py:174-180."""
    f = F.adaptive_avg_pool2d(wide_features(sd, x), (1, 1)).flatten(1)
    return F.linear(f, sd["fc_cls.weight"], sd["fc_cls.bias"])


def image_cam_32x32(sd, scaled_im_list, scaled_position_list, scales, image_wh, side_length: int, num_of_class: int):
    """prepare_seg_inputs.py:96-138 for one image, model included (forward_cam -> F.interpolate to the crop size -> multi_scale_cam)."""
    w, h = image_wh
    crops, sizes = [], []
    for s in range(len(scales)):
        w_, h_ = int(w * scales[s]), int(h * scales[s])
        ix, iy = min(side_length, w_), min(side_length, h_)
        cam_scores = wide_forward_cam(sd, scaled_im_list[s])
        crops.append(F.interpolate(cam_scores, (ix, iy), mode="bilinear", align_corners=False))
        sizes.append((w_, h_))
    return multi_scale_cam(crops, scaled_position_list, sizes, (w, h), num_of_class, side_length)
