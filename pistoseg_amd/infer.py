"""Inference reductions of stages 2 and 4 (CAM / logit -> mask), sharded over ranks by contiguous tile ranges.

stage 2 (infer_pseudo_masks.py:116-154): logits -> bilinear(align_corners=False) to 32x32 (`logits_32x32/*.pt`) and
`get_mask_pred_and_entropy` (absent classes filled with -1e10, softmax, entropy, argmax, tissue==0 -> index C).
stage 4 (infer_revise_masks.py:115-143): (X_rv * label)[:, 1:] -> argmax for pmask_rv, pcam_rv, cam_rv.
PNG / palette / resize post-processing stays host-side I/O (out of scope).
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch

from . import _lib, ops
from .dist import shard_range

Tensor = torch.Tensor


def interpolate_tensor(logits: Tensor, target_shape=(32, 32)) -> Tensor:
    """Batched form of infer_pseudo_masks.py:89-90: [N,C,H,W] f32 -> [N,C,h,w] f32, bilinear align_corners=False."""
    n, c = logits.shape[:2]
    out = torch.empty((n, c) + tuple(target_shape), device=logits.device, dtype=torch.float32)
    ops.bilinear_fwd(logits.contiguous(), "nchw", out, "nchw", False)
    return out


def get_mask_pred_and_entropy(logits: Tensor, tissue: Optional[Tensor], patch_label: Tensor):
    """Batched form of infer_pseudo_masks.py:69-87.  logits [N,C,H,W] f32, tissue uint8 [N,H,W] (0 = background),
    patch_label [N,C] in {0,1}.  Returns (mask uint8 [N,H,W], entropy f32 [N,H,W])."""
    return ops.argmax_mask(logits.contiguous(), mode=_lib.PS_MASK_FILL, label=patch_label, tissue=tissue, want_entropy=True)


@torch.no_grad()
def infer_pseudo_masks(model, images: Tensor, patch_label: Tensor, tissue: Optional[Tensor] = None, batch_size: int = 64,
                       rank: int = 0, world: int = 1, tta: bool = False, writer=None, streams: int = 1):
    """Stage 2 (infer_pseudo_masks.py:116-154) over this rank's contiguous shard of `images` ([T,3,H,W], host or device): per batch
    forward (x8 d4 views when tta, wrapped exactly where the reference wraps: :96) -> `interpolate_tensor` to 32x32 (:126) ->
    `get_mask_pred_and_entropy` (:137).  Returns (lo, hi, logits_32x32 [t,C,32,32], mask uint8 [t,H,W], entropy [t,H,W]); no
    collective on the data path.  `writer` (packed.PackedTilesWriter over ALL T tiles) receives this rank's rows [lo, hi) of the
    32x32 logits -- the reference's one `logits_32x32/<name>.pt` per tile (:127) as rows of one file -- in a single device->host
    copy after the loop, so the launch stream never waits for the host inside the loop.
    streams=2: consecutive (independent) batches alternate between two HIP streams, so that another batch's launch can take the CUs
    a persistent launch leaves idle in its partial last round (512-channel layers at bs=64: 3.5 rounds).  Measured r02 (bs=64, 224x224):
    5637 tiles/s on one stream, 5480 on two, 5504 on three -- the second kernel's blocks start late on the CUs the first still holds
    and then serialise their static tile share (the CU-hog effect of DESIGN 6); the default stays 1."""
    dev = next(model.parameters()).device
    if tta:
        from .tta import SegmentationTTAWrapper

        model = SegmentationTTAWrapper(model, merge_mode="mean")
    lo, hi = shard_range(images.shape[0], rank, world)
    small, masks, ents = [], [], []
    model.eval()
    main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
    side = [main] + [torch.cuda.Stream(device=dev) for _ in range(max(1, streams) - 1)] if main is not None else [None]
    for st in side[1:]:
        st.wait_stream(main)  # inputs produced on the caller's stream
    for bi, s in enumerate(range(lo, hi, batch_size)):
        e = min(hi, s + batch_size)
        st = side[bi % len(side)]
        ctx = torch.cuda.stream(st) if st is not None and st is not main else contextlib.nullcontext()
        with ctx:
            x = images[s:e].to(dev, non_blocking=True)
            logits = model(x)
            small.append(interpolate_tensor(logits))
            m, en = get_mask_pred_and_entropy(logits, None if tissue is None else tissue[s:e].to(dev), patch_label[s:e].to(dev))
            masks.append(m)
            ents.append(en)
            if st is not None and st is not main:
                x.record_stream(st)                  # possibly a view of the caller's tensor (allocated on the caller's stream)
                for t in (small[-1], m, en):
                    t.record_stream(main)            # allocated on `st`, consumed by the concatenation on the caller's stream
    for st in side[1:]:
        main.wait_stream(st)
    cat = lambda xs: torch.cat(xs, 0) if xs else None
    small = cat(small)
    if writer is not None and small is not None:
        writer.write_rows(lo, small)
    return lo, hi, small, cat(masks), cat(ents)


@torch.no_grad()
def infer_revise_masks(model, x: Tensor, pmask: Tensor, pcam: Tensor, label: Tensor):
    """Stage 4 for one batch.  label: [N, C] with the background score prepended.  Returns uint8 masks
    (pmask_rv_mask, pcam_rv_mask, cam_rv_mask), each [N,H,W] with values in 0..C-2."""
    _, cam_rv, pmask_rv, pcam_rv = model(x, pmask, pcam)
    lab = label.reshape(x.shape[0], -1).to(cam_rv.device)
    return tuple(ops.argmax_mask(t, mode=_lib.PS_MASK_MUL, first_ch=1, label=lab) for t in (pmask_rv, pcam_rv, cam_rv))


def _with_background(t: Tensor, dev) -> Tensor:
    """infer_revise_masks.py:125-128: `torch.concat([zeros(n,1,h,w), t], dim=1).cuda()` -- a zero background channel in front."""
    n, c, h, w = t.shape
    out = torch.zeros((n, c + 1, h, w), device=dev, dtype=torch.float32)
    out[:, 1:].copy_(t, non_blocking=True)
    return out


@torch.no_grad()
def infer_revise_masks_sharded(model, images, pmask, cam, label, batch_size: int = 64, rank: int = 0, world: int = 1):
    """Stage 4 as the reference runs it (infer_revise_masks.py:115-143, `infer(dataloader, model, args)`) over this rank's contiguous
    shard of the dataset tensors -- what `RefineDataset` (:28-69) yields, stacked: images [T,3,S,S] (S = 256, :46), pmask [T,C-1,32,32]
    (`logits_32x32/*.pt`), cam [T,C-1,32,32] (`*.npy`), label [T,C-1] -- host or device resident.  Per batch: zero background channel
    in front of pmask / cam (:125-128), background score 1 in front of the label (:130-133), `model(x, pmask, pcam)` (:135),
    `(X_rv * label)[:, 1:]` -> `argmax(dim=1)` for pmask_rv, pcam_rv, cam_rv (:137-143).  `model` is the RFM net or an
    `nn.DataParallel` wrapper around it (:108-111: the checkpoint keys carry `module.`).
    Returns (lo, hi, pmask_rv_mask, pcam_rv_mask, cam_rv_mask): uint8 [hi-lo, S, S] on the device, values 0..C-2.  No collective on
    the data path (tiles are independent); `dist.gather_masks` collects them on rank 0 when one host writes the PNGs (:145-210,
    host-side I/O, out of scope)."""
    net = model.module if isinstance(model, torch.nn.DataParallel) else model
    dev = next(net.parameters()).device
    model.eval()
    lo, hi = shard_range(images.shape[0], rank, world)
    outs = ([], [], [])
    for s in range(lo, hi, batch_size):
        e = min(hi, s + batch_size)
        x = images[s:e].to(dev, non_blocking=True)
        pm = _with_background(pmask[s:e], dev)
        pc = _with_background(cam[s:e], dev)
        lab = torch.ones((e - s, label.shape[1] + 1), device=dev, dtype=torch.float32)
        lab[:, 1:].copy_(label[s:e], non_blocking=True)
        _, cam_rv, pmask_rv, pcam_rv = model(x, pm, pc)
        for acc, t in zip(outs, (pmask_rv, pcam_rv, cam_rv)):
            acc.append(ops.argmax_mask(t, mode=_lib.PS_MASK_MUL, first_ch=1, label=lab))
    size = tuple(images.shape[-2:])
    cat = lambda xs: torch.cat(xs, 0) if xs else torch.empty((0,) + size, device=dev, dtype=torch.uint8)
    return (lo, hi) + tuple(cat(a) for a in outs)
