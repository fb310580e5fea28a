"""Data-parallel steps of both native trainers across two ranks.

Backend: `nccl` (= RCCL) with one GPU per rank whenever the box has two GPUs.  RCCL refuses two ranks of one communicator on the same
device ("Duplicate GPU detected", profiles/r02_rccl_two_ranks_one_gpu_probe.txt), so on the one-GPU test box the two ranks share
cuda:0 over `gloo` instead -- that still drives the whole bucket path (reverse-order arena buckets launched from the backward,
1/world gradient scaling, the f9 unpack of RFMTrainer, the tiles-per-block switch while buckets are in flight); RCCL itself is then
exercised by the world-size-1 test at the bottom (same code path, real RCCL communicator and kernels on the side stream).

Checked: every rank ends with identical parameters, equal to a single-process step on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

DROP_SPECS = (("b6.dropout_2b1", 512, 0.3), ("b6.dropout_2b2", 1024, 0.3), ("b7.dropout_2b1", 1024, 0.5),
              ("b7.dropout_2b2", 2048, 0.5), ("dropout7", 4096, 0.5))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(kind):
    g = torch.Generator().manual_seed(5)
    n = 4
    drops = []
    for _ in range(2):
        drops.append({name: (torch.rand(n, c, generator=g) >= p).float() / (1 - p) for name, c, p in DROP_SPECS})
    if kind.startswith("seg"):
        x = torch.randn(n, 3, 64, 64, generator=g)
        y = torch.randint(0, 4, (n, 64, 64), generator=g)
        return (x, y), drops
    from oracle.make_golden import make_inputs, with_bg

    x, pmask, pcam, lab = make_inputs(n, 64, 4, seed=140)
    pm, pc, label = with_bg(pmask, pcam, lab)
    return (x, pm, pc, label.reshape(n, 4)), drops


def _run(rank, world, port, backend, kind, out):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    from oracle import ref_cpu

    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    D = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        kw = {"device_id": D} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        inputs, drops = _data(kind)
        per = inputs[0].shape[0] // world
        sl = slice(rank * per, (rank + 1) * per)
        pg = dist.group.WORLD if world > 1 else None
        it = iter(drops)
        sample = lambda n_, dev_: {k: v[sl].to(dev_) for k, v in next(it).items()}
        if kind.startswith("seg"):
            from pistoseg_amd.seg_model import ResNet38dSeg
            from pistoseg_amd.trainer import SegTrainer

            model = ResNet38dSeg(3, "fp32")
            model.load_state_dict(ref_cpu.make_state_dict(3, False, seed=42))
            model = model.to(D)
            model.sample_dropout = sample
            # "seg_bf16": the second wire format (buckets cast to bf16, summed, widened back) with CUs reserved for the collectives and the
            # persistent kernels on their ticket queues while buckets are in flight
            extra = dict(grad_payload="bf16", share="reserve+queue", reserved_cus=32) if kind == "seg_bf16" else {}
            tr = SegTrainer(model, lr=1e-3, weight_decay=0.05, ignore_index=3, process_group=pg, bucket_mb=64.0, track_iou=False, **extra)
            step = lambda: float(tr.train_step(inputs[0][sl].to(D), inputs[1][sl].to(D)))
        else:
            from pistoseg_amd.revise_net import Net
            from pistoseg_amd.trainer import RFMTrainer

            model = Net(4, "fp32")
            model.load_state_dict(ref_cpu.make_state_dict(4, True, seed=42))
            model = model.to(D)
            model.sample_dropout = sample
            tr = RFMTrainer(model, lr=0.01, wt_dec=5e-4, max_step=10, process_group=pg, bucket_mb=64.0)
            step = lambda: float(tr.train_step(*(t[sl].to(D) for t in inputs))[0])
        if world > 1:
            assert len(tr.reducer.buckets) >= 3  # several buckets so the overlap path is exercised
            assert dist.get_backend(pg) == backend
        init = tr.p_flat[::499].cpu().numpy()
        losses = [step() for _ in range(2)]
        torch.cuda.synchronize()
        f9 = {k: tr.p_flat[o:o + n].cpu().numpy() for k, (o, n) in tr.offsets.items() if k.startswith("f9_")}
        out.put((world, rank, losses, tr.p_flat[::499].cpu().numpy(), f9, init))  # every 499th weight (numpy: plain pickle)
    except Exception as e:  # pragma: no cover
        import traceback

        out.put((world, rank, "ERR " + repr(e) + traceback.format_exc(), None, None, None))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["seg", "rfm", "seg_bf16"])
def test_two_rank_step_equals_single_process_step(kind):
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(0, 1, 0, backend, kind, q))]
    port = _free_port()
    procs += [ctx.Process(target=_run, args=(r, 2, port, backend, kind, q)) for r in range(2)]
    procs[0].start()
    single = q.get(timeout=300)
    procs[0].join(timeout=60)
    for p in procs[1:]:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs[1:]:
        p.join(timeout=60)
    assert not isinstance(single[2], str), single[2]
    for r in res:
        assert not isinstance(r[2], str), r[2]
    r0, r1 = sorted(res, key=lambda t: t[1])
    assert (r0[3] == r1[3]).all(), "ranks diverged"
    for k in r0[4]:
        assert (r0[4][k] == r1[4][k]).all(), f"ranks diverged on {k}"
    # global loss = mean of the per-rank means (equal shard sizes; every loss of the path is a mean over samples of per-sample terms)
    for s in range(2):  # (seg_bf16: step 2 starts from weights updated with bf16-rounded gradients)
        assert abs(0.5 * (r0[2][s] + r1[2][s]) - single[2][s]) < (2e-3 if kind == "seg_bf16" and s > 0 else 2e-4) * abs(single[2][s])
    d = abs(r0[3] - single[3])
    if kind == "seg":
        assert float(d.mean()) < 1e-6 and float(d.max()) <= 2 * 2 * 1e-3 * 1.1  # Adam sign noise bound, see test_modules_gpu
    elif kind == "seg_bf16":
        # gradients rounded to bf16 on the wire (2^-9 relative): Adam's normalised update moves by that fraction of lr per step on average
        print(f"seg_bf16: mean |dp| {float(d.mean()):.3e}, max |dp| {float(d.max()):.3e}")
        assert float(d.mean()) < 2e-5 and float(d.max()) <= 2 * 2 * 1e-3 * 1.1
    else:
        # plain SGD: the update is linear in the gradient.  Between a 4-tile step and two 2-tile steps the weight gradients are summed
        # in a different order (f32 atomics) and step 2 starts from parameters that already differ in the last bits, so the
        # parameters agree to a small fraction of the distance they moved -- not bit for bit.
        upd = float(abs(single[3] - single[5]).max())
        print(f"rfm: max |update| {upd:.3e}, max |dp| {float(d.max()):.3e}, mean |dp| {float(d.mean()):.3e}")
        assert float(d.max()) <= 5e-3 * upd and float(d.mean()) <= 1e-4 * upd, (float(d.max()), float(d.mean()), upd)
        for k in r0[4]:  # the packed f9 gradient was unpacked into both arena slots before their bucket was reduced
            assert float(abs(r0[4][k] - single[4][k]).max()) <= 5e-3 * upd, k
    print(f"{kind}: backend {backend}, losses {r0[2]} / {r1[2]} vs single {single[2]}, max |dp| {float(d.max()):.2e}")


def _rccl_world1(port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from pistoseg_amd.dist import BucketedAllReduce, plan_buckets

        assert dist.get_backend() == "nccl"
        n = 1 << 22
        flat = torch.zeros(3 * n, device="cuda")
        red = BucketedAllReduce(flat, plan_buckets([("fc8.w", n), ("b7.w", n), ("b6.w", n)], n // 2), dist.group.WORLD)
        assert len(red.buckets) == 3
        ok = True
        for step in range(3):
            red.begin_step()
            for i, unit in enumerate(("fc8", "b7", "b6")):
                # the producer of a bucket is still running on the launch stream when the bucket is handed to RCCL on the side stream
                a = torch.randn(2048, 2048, device="cuda")
                for _ in range(4):
                    a = a @ a * 1e-3
                flat[i * n:(i + 1) * n] = float(step * 3 + i + 1) + 0.0 * a.flatten()[0]
                red.on_unit_done(unit)
            red.finish()
            want = torch.cat([torch.full((n,), float(step * 3 + i + 1)) for i in range(3)])
            ok = ok and bool(torch.equal(flat.cpu(), want))
        # the bf16 wire format and the CU reservation through the same communicator: cast -> RCCL all-reduce of the bf16 staging slice -> widen,
        # all on the communication stream; the launch option toggled is the given model's
        class _Opts:
            tiles_per_block = None
            cus_reserved = None
            tile_queue = None
        opts = _Opts()
        red16 = BucketedAllReduce(flat, red.buckets, dist.group.WORLD, launch_opts=opts, payload="bf16", share="reserve+queue", reserved_cus=32)
        red16.begin_step()
        vals = torch.randn(3 * n, device="cuda")
        flat.copy_(vals)
        seen = []
        share = red16._share_gpu
        red16._share_gpu = lambda on: (share(on), seen.append((on, opts.cus_reserved, opts.tiles_per_block, opts.tile_queue)))[0]
        red16.on_unit_done("fc8")
        red16.on_unit_done("b7")
        red16.finish()
        # (a finished bucket may hand the CUs back before the next one is launched: only the first and the last transition are fixed)
        ok = ok and seen[0] == (True, 32, None, 1) and seen[-1] == (False, None, None, None) and opts.cus_reserved is None and opts.tile_queue is None
        ok = ok and bool(torch.equal(flat, vals.to(torch.bfloat16).float()))
        # the reference-API path's own reducer (arena.ParamArena.attach_reducer) through the same RCCL communicator: stage 5 as Lightning drives it,
        # gradients all-reduced in buckets behind the autograd node's reverse plan -- one rank, so the weights must equal a run without a reducer
        import argparse

        from oracle import ref_cpu
        from oracle.make_golden import make_inputs
        from pistoseg_amd.arena import ParamArena
        from pistoseg_amd.segmentation_module import SegmentationModule

        sd = ref_cpu.make_state_dict(3, False, seed=42)
        x, *_ = make_inputs(2, 64, 4, 91)
        target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(4)).cuda()
        states, drops = [], None
        for with_reducer in (False, True):
            args = argparse.Namespace(patch_size=64, num_classes=3, dataset="wsss4luad", model="ResNet38d", encoder="resnet38d", lr=1e-3, weight_decay=0.05,
                                      tta=False, log_path="/tmp", precision="bf16")
            mod = SegmentationModule(args).cuda()
            mod.model.load_state_dict(sd)
            mod.model.launch.deterministic = True
            if drops is None:
                drops = [mod.model.sample_dropout(2, torch.device("cuda", 0)) for _ in range(2)]
            it = iter(drops)
            mod.model.sample_dropout = lambda n_, dev_: next(it)
            (opt,), _ = mod.configure_optimizers()
            if with_reducer:
                red_api = ParamArena.of(mod.model).attach_reducer(dist.group.WORLD, bucket_mb=16.0)
                assert len(red_api.buckets) >= 3 and mod.model.launch.stream_k is False
            for i in range(2):
                loss = mod.training_step({"image": x.cuda(), "mask": target, "label": None}, i)
                opt.zero_grad()
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            states.append({k: v.detach().cpu().clone() for k, v in mod.model.state_dict().items()})
        ok = ok and all(torch.equal(states[0][k], states[1][k]) for k in states[0])
        out.put(("ok" if ok else "values wrong (side stream ran ahead of the producer?)"))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        out.put("ERR " + repr(e) + traceback.format_exc())


def test_bucketed_allreduce_through_rccl_world1():
    """backend 'nccl' IS RCCL: a one-rank communicator on the test GPU runs the reducer's real code path (RCCL all-reduce launched on the
    comm stream behind an event on the producing stream, tiles-per-block switch, finish() join)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=300)
    p.join(timeout=60)
    assert msg == "ok", msg
