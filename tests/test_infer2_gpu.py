"""BASELINE configs[2] end to end: `infer.infer_pseudo_masks` -- the stage-2 loop of infer_pseudo_masks.py:116-154 (forward, optional
d4 TTA, 32x32 downsample, label-masked softmax / entropy / argmax / background fill) -- over a tile set cut into contiguous shards,
against the oracle's composition of the same statements on the CPU (fp32 parity path: logits_32x32 / entropy within 1e-4, masks
bit-exact up to ties below the logit error), plus the packed writer the shards share."""
import numpy as np
import pytest
import torch

from _parity import assert_tie_excused
from oracle import ref_cpu
from oracle.make_golden import make_inputs

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def oracle_stage2(sd, x, tissue, labels, tta):
    """The reference loop body on CPU: model(image_batch) [wrapped in the d4 TTA when asked], then per tile interpolate_tensor and
    get_mask_pred_and_entropy."""
    fwd = lambda t: ref_cpu.seg_forward(sd, t)
    with torch.no_grad():
        logits = ref_cpu.d4_tta(fwd, x) if tta else fwd(x)
    small, masks, ents = [], [], []
    for lg, ts, lab in zip(logits, tissue, labels):
        small.append(ref_cpu.interpolate_tensor(lg, (32, 32)))
        m, e = ref_cpu.get_mask_pred_and_entropy(lg, ts.numpy(), [int(v) for v in lab])
        masks.append(np.asarray(m))
        ents.append(np.asarray(e, dtype=np.float32))
    return logits, torch.stack(small), np.stack(masks), np.stack(ents)


@pytest.mark.parametrize("tta,world,streams", [(False, 2, 1), (False, 3, 2), (True, 2, 2), (False, 7, 1)])  # 7 ranks over 5 tiles: two EMPTY shards
def test_infer_pseudo_masks_sharded_matches_oracle_composition(tmp_path, tta, world, streams):
    from pistoseg_amd import infer
    from pistoseg_amd.packed import PackedTiles, PackedTilesWriter
    from pistoseg_amd.seg_model import ResNet38dSeg

    c, s, T = 3, 64, 5  # 5 tiles over 2 or 3 shards: ragged last shard, batch_size 2 -> ragged last batch
    sd = ref_cpu.make_state_dict(c, False, seed=42)
    model = ResNet38dSeg(classes=c, precision="fp32")
    model.load_state_dict(sd, strict=True)
    model = model.to(D)
    x, *_ = make_inputs(T, s, 4, seed=160)
    rs = np.random.RandomState(161)
    labels = torch.tensor([[1, 1, 0], [0, 1, 0], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=torch.float32)  # incl. a single-label tile
    tissue = torch.from_numpy((rs.uniform(size=(T, s, s)) > 0.2).astype(np.uint8) * 255)
    ref_logits, ref_small, ref_mask, ref_ent = oracle_stage2(sd, x, tissue, labels, tta)

    names = [f"tile{i}" for i in range(T)]
    pack = str(tmp_path / "logits_32x32.pack")
    got_small, got_mask, got_ent = [None] * T, [None] * T, [None] * T
    covered = []
    for rank in range(world):  # every rank's call, one after the other on the test GPU; nothing is exchanged between them
        writer = PackedTilesWriter(pack, names, (c, 32, 32), shared=True)
        lo, hi, small, masks, ents = infer.infer_pseudo_masks(model, x, labels, tissue, batch_size=2, rank=rank, world=world, tta=tta, writer=writer, streams=streams)
        writer.close()
        covered.append((lo, hi))
        if hi == lo:
            assert small is None and masks is None and ents is None  # an empty shard: nothing launched, nothing written
        if hi > lo:
            assert small.shape == (hi - lo, c, 32, 32) and masks.dtype == torch.uint8 and masks.shape == (hi - lo, s, s)
            for i in range(lo, hi):
                got_small[i], got_mask[i], got_ent[i] = small[i - lo].cpu(), masks[i - lo].cpu().numpy(), ents[i - lo].cpu().numpy()
    # contiguous shards tile [0, T) exactly, in rank order (dist.shard_range)
    assert covered[0][0] == 0 and covered[-1][1] == T and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    scale = float(ref_logits.abs().max())
    # logits_32x32 (what the reference torch.save()s per tile, :126-127): within 1e-4 of the logit range
    got_small = torch.stack(got_small)
    err = float((got_small - ref_small).abs().max())
    assert err < (1e-4 if not tta else 2e-4) * scale, err
    # ... and the packed file holds exactly what the ranks returned
    assert torch.equal(PackedTiles(pack).get(names), got_small)
    # masks: bit-exact except where the oracle's top-2 (label-masked) logits are closer than the logit error; tissue==0 -> index C
    ndiff, within = 0, True
    for i in range(T):
        lab = labels[i].tolist()
        z = ref_logits[i].clone()
        for k_, present in enumerate(lab):
            if not present:
                z[k_] = -1e10
        top2 = torch.topk(z, 2, dim=0)[0]
        gap = (top2[0] - top2[1]).abs().numpy()
        diff = got_mask[i] != ref_mask[i]
        ndiff += int(diff.sum())
        within = within and bool((gap[diff] <= 4 * max(err, 1e-4 * scale)).all())
        assert (got_mask[i][tissue[i].numpy() == 0] == c).all()
        if sum(lab) == 1:  # single-label shortcut (:70-73): constant mask, zero entropy
            assert (got_mask[i][tissue[i].numpy() != 0] == lab.index(1)).all() and float(np.abs(got_ent[i]).max()) == 0.0
    assert_tie_excused(f"stage-2 masks (tta={tta}, {world} shards)", ndiff, T * s * s, within)
    # entropy: -sum p log(p + 1e-10), f32; its sensitivity to a logit error e is O(e * log C)
    ent_err = max(float(np.abs(got_ent[i] - ref_ent[i]).max()) for i in range(T))
    assert ent_err < 2e-3, ent_err
    print(f"[parity] stage 2 (tta={tta}, {world} shards): logits_32 err {err:.2e} (range {scale:.2f}), entropy err {ent_err:.2e}")


def test_packed_tiles_to_device_and_stage3_indexing(tmp_path):
    """SURVEY 8f row 3 on the device: a pack written from device tensors (stage-2 logits, f32) and one converted from float64 `.npy` CAMs
    are brought into HBM whole (`to_device`) and indexed with a batch's tile indices, as stage 3 consumes them -- values bit-identical
    to what RefineDataset's per-file loaders return (revise_pseudo_labels.py:57-60: torch.load(...), torch.from_numpy(cam).float())."""
    from pistoseg_amd.packed import PackedTiles, PackedTilesWriter, pack_cam_dir

    g = torch.Generator().manual_seed(9)
    names = [f"img{i:03d}-[1, 0, 1]" for i in range(13)]
    logits = torch.randn(13, 3, 32, 32, generator=g)
    w = PackedTilesWriter(str(tmp_path / "logits.pack"), names, (3, 32, 32))
    w.write_rows(0, logits[:7].to(D))       # device tensors, as infer_pseudo_masks hands them over
    w.write_rows(7, logits[7:].to(D))
    w.close()
    cam_dir = tmp_path / "cam"
    cam_dir.mkdir()
    cams = torch.randn(13, 3, 32, 32, generator=g, dtype=torch.float64)
    for n_, c_ in zip(names, cams):
        np.save(cam_dir / (n_ + ".npy"), c_.numpy())
    pc = pack_cam_dir(str(cam_dir), str(tmp_path / "cam.pack"))
    pl = PackedTiles(str(tmp_path / "logits.pack"))
    dl, dc = pl.to_device(D), pc.to_device(D)
    assert dl.is_cuda and dl.dtype == torch.float32 and tuple(dl.shape) == (13, 3, 32, 32)
    idx = torch.tensor([pl.index[n_] for n_ in (names[5], names[0], names[12])], device=D)
    assert torch.equal(dl[idx].cpu(), logits[[5, 0, 12]])
    order = [pc.index[n_] for n_ in names]
    assert torch.equal(dc.cpu()[order], cams.to(torch.float32))  # the float64 -> float32 cast of the reference's loader, bit for bit
