"""N > 1 logic on CPU: world_size-2 gloo processes exercise the bucketed gradient all-reduce, the inference
index sharding and the mask gather exactly as the GPU ranks drive them."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pistoseg_amd.dist import BucketedAllReduce, allreduce_confusion, gather_masks, plan_buckets, shard_range

        # ---- gradient arena: entries in the order the reverse plan finalises them
        entries = [("fc8.weight", 7), ("b7.conv_branch2a.weight", 100), ("b7.conv_branch1.weight", 50), ("b6.conv_branch2a.weight", 30),
                   ("b5.conv_branch2a.weight", 200), ("b4.conv_branch2a.weight", 10), ("b3.conv_branch2a.weight", 5)]
        total = sum(n for _, n in entries)
        buckets = plan_buckets(entries, limit_elems=120)
        assert buckets[0][1] == 0 and buckets[-1][2] == total
        assert all(b[2] == nb[1] for b, nb in zip(buckets, buckets[1:]))  # tile the arena exactly
        flat = torch.zeros(total)
        red = BucketedAllReduce(flat, buckets, None)
        for step in range(2):  # two steps: state resets between steps
            flat.zero_()
            red.begin_step()
            off = 0
            last_unit = None
            for name, n in entries:  # "backward": fill unit by unit, notify at unit boundaries
                unit = name.split(".")[0]
                if last_unit is not None and unit != last_unit:
                    red.on_unit_done(last_unit)
                flat[off:off + n] = torch.arange(n, dtype=torch.float32) * (rank + 1) + step
                off += n
                last_unit = unit
            # the last unit (b3) is never reported, as for a frozen tail: finish() must flush it
            red.finish()
            off = 0
            for name, n in entries:
                expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)) + step * world
                assert torch.equal(flat[off:off + n], expect), (name, step)
                off += n
        # ---- the bf16 wire format: every bucket is cast to bf16, summed, widened back -- equal to the f32 exchange of the bf16-ROUNDED
        # per-rank gradients up to one bf16 rounding of the sum (2^-9 relative), and the launch options it toggles are its own model's
        class _Opts:
            tiles_per_block = None
        g = torch.Generator().manual_seed(5 + rank)
        mine = torch.randn(total, generator=g)
        flat16 = mine.clone()
        red16 = BucketedAllReduce(flat16, buckets, None, launch_opts=_Opts(), payload="bf16")
        red16.begin_step()
        for unit in ("fc8", "b7", "b6", "b5", "b4"):
            red16.on_unit_done(unit)
        red16.finish()
        parts = [torch.randn(total, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        exact = sum(parts)
        rounded = sum(p_.to(torch.bfloat16).float() for p_ in parts)
        assert flat16.dtype == torch.float32
        # (a bf16 sum over p ranks rounds p - 1 times, 2^-9 relative each)
        assert float((flat16 - rounded).abs().max()) <= (world - 1) * 2.0 ** -8 * float(rounded.abs().max()) + 1e-6
        assert float((flat16 - exact).abs().max()) <= (world - 1) * 2.0 ** -6 * float(exact.abs().max())
        assert red16.launch_opts.tiles_per_block is None  # (CPU tensors: no comm stream, nothing to share)
        # ---- inference sharding + gather (BASELINE config 3: 10k tiles over p ranks)
        n_items = 10_001
        lo, hi = shard_range(n_items, rank, world)
        covered = torch.zeros(n_items, dtype=torch.int64)
        covered[lo:hi] = 1
        dist.all_reduce(covered)
        assert int(covered.min()) == 1 and int(covered.max()) == 1
        small = 7
        lo, hi = shard_range(small, rank, world)
        local = torch.arange(lo, hi, dtype=torch.uint8).view(-1, 1, 1).expand(-1, 2, 3).contiguous()
        allm = gather_masks(local, small)
        if rank == 0:
            assert torch.equal(allm[:, 0, 0], torch.arange(small, dtype=torch.uint8))
        cm = torch.full((9,), rank + 1, dtype=torch.int64)
        assert int(allreduce_confusion(cm)[0]) == sum(r + 1 for r in range(world))
        if world == 8:
            # BASELINE configs[2] / [3] at the node's rank count: 10 000 tiles -> 1250 per rank, and the REAL model's bucket plan (48 MB buckets over the
            # 104 M-element arena in reverse-plan order) is the same list on every rank -- a rank that cut its buckets differently would deadlock RCCL
            assert shard_range(10_000, rank, world) == (1250 * rank, 1250 * (rank + 1))
            from pistoseg_amd.arena import arena_order
            from pistoseg_amd.revise_net import Net
            from pistoseg_amd.seg_model import ResNet38dSeg

            for model in (ResNet38dSeg(3), Net(4)):
                ent = [(name, p.numel()) for name, p in arena_order(model)]
                plan = plan_buckets(ent, int(48 * (1 << 20) / 4))
                assert plan[0][1] == 0 and plan[-1][2] == sum(n for _, n in ent) and all(b[2] == nb[1] for b, nb in zip(plan, plan[1:]))
                assert all(b % 8 == 0 and e % 8 == 0 for _, b, e in plan)  # the bf16 wire format's alignment (dist.BucketedAllReduce)
                plans = [None] * world
                dist.all_gather_object(plans, plan)
                assert all(pl == plan for pl in plans)
                if rank == 0:
                    print(f"{type(model).__module__}: {len(plan)} buckets, MB {[round((e - b) * 4 / 2**20, 1) for _, b, e in plan]}")
        out.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        out.put((rank, traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_world8_gloo():
    """The node's rank count on CPU: same worker with 8 ranks (sums over 8 ranks, 8 shards, the real models' bucket plans compared across ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(8)], res


def test_reserved_cus_follow_rccl_channel_settings(monkeypatch):
    sys.path.insert(0, ROOT)
    from pistoseg_amd.dist import RCCL_DEFAULT_CHANNELS, default_reserved_cus

    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert default_reserved_cus() == RCCL_DEFAULT_CHANNELS == 32
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "16")
    assert default_reserved_cus() == 16
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "24")  # a floor above the cap wins, as in RCCL
    assert default_reserved_cus() == 24
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "junk")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "64")
    assert default_reserved_cus() == 64


def test_plan_buckets_edges():
    sys.path.insert(0, ROOT)
    from pistoseg_amd.dist import plan_buckets, shard_range

    assert plan_buckets([], 10) == []
    assert plan_buckets([("b7.x.weight", 5)], 10) == [("b7", 0, 5)]
    # a unit is never split across buckets
    b = plan_buckets([("b7.a.weight", 8), ("b7.b.weight", 8), ("b6.a.weight", 1)], 10)
    assert b == [("b7", 0, 16), ("b6", 16, 17)]
    assert shard_range(0, 0, 8) == (0, 0) and shard_range(3, 7, 8) == (3, 3) and shard_range(10, 1, 4) == (3, 6)
