"""Packed on-disk intermediates: bit-exact round trips against the reference's per-file formats
(torch.save'd [C,32,32] f32 logits, np.save'd float64 CAMs) and what RefineDataset.__getitem__ returns for them
(revise_pseudo_labels.py:57-60)."""
import os

import numpy as np
import torch

from pistoseg_amd.packed import PackedTiles, PackedTilesWriter, pack_cam_dir, pack_logits_dir


def test_pack_round_trip_matches_per_file_loaders(tmp_path):
    g = torch.Generator().manual_seed(0)
    names = [f"{1000 + i}_1.0_{i * 3}_{i * 7}-[1, 0, 1]" for i in range(17)]
    ld, cd = tmp_path / "logits_32x32", tmp_path / "cam"
    ld.mkdir()
    cd.mkdir()
    logits, cams = {}, {}
    for n in names:
        logits[n] = torch.randn(3, 32, 32, generator=g)
        cams[n] = torch.randn(3, 32, 32, generator=g, dtype=torch.float64).numpy() * 1e-3
        torch.save(logits[n], ld / (n + ".pt"))           # infer_pseudo_masks.py:127
        np.save(cd / (n + ".npy"), cams[n])                # prepare_seg_inputs.py:138
    pl = pack_logits_dir(str(ld), str(tmp_path / "logits.pack"))
    pc = pack_cam_dir(str(cd), str(tmp_path / "cam.pack"))
    assert len(pl) == len(pc) == 17 and pl.names == sorted(names)
    for n in names:
        ref_pmask = torch.load(ld / (n + ".pt"), map_location="cpu")                      # RefineDataset: pmask
        ref_cam = torch.from_numpy(np.load(cd / (n + ".npy"))).to(torch.float32)          # RefineDataset: cam
        assert torch.equal(pl[n], ref_pmask) and torch.equal(pc[n], ref_cam)
    batch = [names[5], names[0], names[16]]
    assert torch.equal(pl.get(batch), torch.stack([logits[n] for n in batch]))
    assert pc.array.dtype == np.float64 and n in pc and "nope" not in pc
    # one file each instead of 17 + 17
    assert os.path.getsize(tmp_path / "logits.pack") < 4096 * 2 + 17 * 3 * 32 * 32 * 4 + 1


def test_sharded_writers_fill_disjoint_rows(tmp_path):
    """Two ranks (contiguous shard ranges, as infer_pseudo_masks shards its tiles) write one pack."""
    from pistoseg_amd.dist import shard_range

    names = [f"t{i}" for i in range(11)]
    path = str(tmp_path / "x.pack")
    data = torch.arange(11 * 2 * 4 * 4, dtype=torch.float32).reshape(11, 2, 4, 4)
    PackedTilesWriter(path, names, (2, 4, 4)).close()
    for rank in (1, 0):
        lo, hi = shard_range(11, rank, 2)
        w = PackedTilesWriter(path, names, (2, 4, 4), create=False)
        w.write_rows(lo, data[lo:hi])
        w.close()
    assert torch.equal(PackedTiles(path).get(names), data)


def _shared_writer(rank, path, n, q):
    from pistoseg_amd.dist import shard_range

    names = [f"t{i}" for i in range(n)]
    lo, hi = shard_range(n, rank, 4)
    w = PackedTilesWriter(path, names, (3, 8, 8), shared=True)  # every rank constructs it the same way, concurrently
    w.write_rows(lo, torch.full((hi - lo, 3, 8, 8), float(rank + 1)))
    w.close()
    q.put(rank)


def test_shared_writers_race_without_truncating_each_other(tmp_path):
    """Four processes construct a `shared=True` writer over the same path at once: the file is published atomically by one of them,
    nobody truncates rows another rank already wrote, and a writer with a different index is refused."""
    import multiprocessing as mp

    from pistoseg_amd.dist import shard_range

    path, n = str(tmp_path / "shared.pack"), 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shared_writer, args=(r, path, n, q)) for r in range(4)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = PackedTiles(path).get([f"t{i}" for i in range(n)])
    for r in range(4):
        lo, hi = shard_range(n, r, 4)
        assert bool((got[lo:hi] == r + 1).all())
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    import pytest

    with pytest.raises(ValueError):
        PackedTilesWriter(path, [f"u{i}" for i in range(n)], (3, 8, 8), shared=True)
    with pytest.raises(ValueError):
        PackedTilesWriter(path, [f"t{i}" for i in range(n)], (3, 8, 9), create=False)
