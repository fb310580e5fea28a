"""The reference's own training loops driven against the mirrors: `SegmentationModule.training_step` + `configure_optimizers()`'s optimiser +
`loss.backward()` (models/segmentation_module.py:86-111) and the stage-3 `train_epoch` body (revise_pseudo_labels.py:250-301) -- arena-backed
optimisers against the torch optimisers they stand in for, gradient accumulation semantics, no host synchronisation inside a step."""
import argparse

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.make_golden import make_inputs

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def make_args(**kw):
    base = dict(patch_size=64, num_classes=3, dataset="wsss4luad", model="ResNet38d", encoder="resnet38d", lr=1e-3, weight_decay=0.05,
                tta=False, log_path="/tmp", precision="fp32")
    base.update(kw)
    return argparse.Namespace(**base)


def _seg_model(sd, precision="fp32", deterministic=True):
    from pistoseg_amd.seg_model import ResNet38dSeg

    m = ResNet38dSeg(3, precision)
    m.load_state_dict(sd)
    m = m.to(D)
    m.train()
    m.launch.deterministic = deterministic  # weight gradients without atomics: the two paths compared below see identical gradients
    return m


def test_arena_adamw_is_torch_adamw_over_the_arena():
    """configure_optimizers()'s optimiser: same arithmetic as torch.optim.AdamW on the same gradients (3 steps, ExponentialLR between), state_dict in
    torch's layout (loads into a stock AdamW and back), `p.grad` = slices of the gradient arena, one memset zero_grad."""
    from pistoseg_amd import ops
    from pistoseg_amd.arena import ArenaAdamW, ParamArena

    sd = ref_cpu.make_state_dict(3, False, seed=42)
    x, *_ = make_inputs(2, 64, 4, 109)
    target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(2)).to(D)
    drops, results, opts = None, [], []
    for arena in (True, False):
        model = _seg_model(sd)
        if drops is None:
            drops = [model.sample_dropout(2, D) for _ in range(3)]
        it = iter(drops)
        model.sample_dropout = lambda n_, dev_: next(it)
        params = [p for p in model.parameters() if p.requires_grad]
        if arena:
            opt = ArenaAdamW(params, 1e-3, weight_decay=0.05)
            assert isinstance(opt, torch.optim.AdamW)
        else:
            model.grad_sink = "autograd"  # fresh gradient tensors handed to autograd, accumulated into p.grad by torch
            opt = torch.optim.AdamW(params, 1e-3, weight_decay=0.05)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
        losses = []
        for _ in range(3):
            logits = model(x.to(D))
            loss, dl = ops.softmax_ce(logits.detach(), target, 3, want_grad=True)
            opt.zero_grad()
            logits.backward(dl)
            opt.step()
            sched.step()
            losses.append(float(loss))
        if arena:
            a = ParamArena.of(model, create=False)
            assert a is not None
            for name, p in a.entries:
                o, n = a.offsets[name]
                assert p.grad is not None and p.grad.data_ptr() == a.g_flat.data_ptr() + 4 * o and p.data_ptr() == a.p_flat.data_ptr() + 4 * o
                assert p.grad.shape == p.shape and p.grad.stride() == p.stride()
        results.append((losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
        opts.append((opt, params))
    (l1, s1), (l2, s2) = results
    assert abs(l1[0] - l2[0]) < 1e-6 and abs(l1[2] - l2[2]) < 2e-4 * abs(l2[2]), (l1, l2)
    for k in s1:
        if s1[k].is_floating_point():
            # identical gradients (deterministic weight gradients), so what is left is the fused kernel's rounding against torch's foreach ops; an
            # element whose moments sit at the noise floor can still flip the sign of an early Adam step (2 lr per step) -- bound the maximum by
            # that, require the mean to vanish
            d = (s1[k] - s2[k]).abs()
            assert float(d.max()) <= 3 * 2 * 1e-3 * 1.1 and float(d.mean()) < 1e-6, (k, float(d.max()), float(d.mean()))
    # optimiser state: torch's keys and layout, interchangeable with the stock optimiser's
    (oa, pa), (ot, pt) = opts
    sda, sdt = oa.state_dict(), ot.state_dict()
    assert sda["param_groups"][0]["lr"] == pytest.approx(1e-3 * 0.9 ** 3) and sda["param_groups"][0]["lr"] == pytest.approx(sdt["param_groups"][0]["lr"])
    assert set(sda["state"]) == set(sdt["state"])
    for i in sda["state"]:
        assert set(sda["state"][i]) == {"step", "exp_avg", "exp_avg_sq"} and int(sda["state"][i]["step"]) == 3 == int(sdt["state"][i]["step"])
        for key in ("exp_avg", "exp_avg_sq"):
            a_, t_ = sda["state"][i][key].float().cpu(), sdt["state"][i][key].float().cpu()
            assert a_.shape == t_.shape and float((a_ - t_).abs().max()) <= 5e-3 * float(t_.abs().max()) + 1e-12, (i, key)  # (the weights differ by rounding after step 1, so do the later gradients)
    ot.load_state_dict(sda)  # arena state into the stock optimiser ...
    before = oa._flat_state["exp_avg"].clone()
    oa.load_state_dict(sdt)  # ... and torch's back into the arena optimiser: still views of the flat buffers
    st0 = oa.state[pa[0]]
    hit = oa.arena().owns(pa[0])
    assert st0["exp_avg"].data_ptr() == oa._flat_state["exp_avg"].data_ptr() + 4 * hit[0]
    assert float((oa._flat_state["exp_avg"] - before).abs().max()) <= 5e-3 * float(before.abs().max())
    oa.step()
    assert int(oa.state[pa[0]]["step"]) == 4


@pytest.mark.selfcheck
def test_autograd_sink_modes_and_accumulation_semantics():
    """`grad_sink="arena"` and `"autograd"` produce the same gradients; two backwards without zero_grad accumulate (autograd's contract); a foreign
    `zero_grad(set_to_none=True)` (stock optimiser, `model.zero_grad()`) is honoured: the next backward starts from zeros."""
    from pistoseg_amd import ops

    sd = ref_cpu.make_state_dict(3, False, seed=42)
    x, *_ = make_inputs(2, 64, 4, 31)
    target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(5)).to(D)
    model = _seg_model(sd)
    drop = model.sample_dropout(2, D)
    model.sample_dropout = lambda n_, dev_: drop

    def backward_once():
        logits = model(x.to(D))
        _, dl = ops.softmax_ce(logits.detach(), target, 3, want_grad=True)
        logits.backward(dl)

    named = dict(model.named_parameters())
    keys = [k for k, p in named.items() if p.requires_grad]
    backward_once()
    g1 = {k: named[k].grad.clone() for k in keys}
    backward_once()  # no zero_grad in between: .grad accumulates
    for k in keys:
        assert torch.equal(named[k].grad, 2 * g1[k]), k
    model.zero_grad(set_to_none=True)
    assert all(named[k].grad is None for k in keys)
    backward_once()
    for k in keys:
        assert torch.equal(named[k].grad, g1[k]), k
    model.zero_grad(set_to_none=True)
    model.grad_sink = "autograd"
    backward_once()
    for k in keys:
        assert torch.equal(named[k].grad, g1[k]), k
    assert named["conv1a.weight"].grad is None


def test_module_api_step_equals_native_trainer_and_never_synchronises():
    """Stage 5 as Lightning drives it -- training_step, optimizer.zero_grad(), loss.backward(), optimizer.step() -- against trainer.SegTrainer:
    same losses and weights after 3 steps; and a steady-state step issues no synchronising call (torch's sync debug mode raises on .item() /
    .cpu() / blocking copies), the logged values staying on the device."""
    from pistoseg_amd.arena import ArenaAdamW
    from pistoseg_amd.metrics import LazyScalar
    from pistoseg_amd.segmentation_module import SegmentationModule
    from pistoseg_amd.trainer import SegTrainer

    sd = ref_cpu.make_state_dict(3, False, seed=42)
    x, *_ = make_inputs(2, 64, 4, 77)
    target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(9)).to(D)
    xd = x.to(D)
    drops = None
    res = []
    for api in (True, False):
        if api:
            mod = SegmentationModule(make_args()).to(D)
            model = mod.model
            model.load_state_dict(sd)
            model.launch.deterministic = True
        else:
            model = _seg_model(sd)
        model.train()
        if drops is None:
            drops = [model.sample_dropout(2, D) for _ in range(3)]
        it = iter(drops)
        model.sample_dropout = lambda n_, dev_: next(it)
        losses = []
        if api:
            (opt,), (sched,) = mod.configure_optimizers()
            assert isinstance(opt, ArenaAdamW) and isinstance(opt, torch.optim.AdamW) and sched.gamma == 0.9
            batch = {"image": xd, "mask": target, "label": None}
            for i in range(3):
                if i == 2:
                    torch.cuda.synchronize()
                    torch.cuda.set_sync_debug_mode("error")
                try:
                    loss = mod.training_step(batch, i)
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
                finally:
                    torch.cuda.set_sync_debug_mode("default")
                losses.append(float(loss))
            assert torch.is_tensor(mod.logged["train_miou"]) and mod.logged["train_miou"].is_cuda and mod.logged["train_miou"].dtype == torch.float64
            assert float(mod.logged["train_miou"]) == mod.train_iou.Mean_Intersection_over_Union()
            model.sample_dropout = lambda n_, dev_: drops[0]
            with torch.no_grad():
                lazy = mod.train_iou(mod.model(xd), target)
            assert all(isinstance(v, LazyScalar) for v in lazy)
            assert float(lazy[0]) == mod.train_iou.Mean_Intersection_over_Union() and float(lazy[1]) == mod.train_iou.Frequency_Weighted_Intersection_over_Union()
        else:
            tr = SegTrainer(model, lr=1e-3, weight_decay=0.05, ignore_index=3, deterministic=True)
            losses = [float(tr.train_step(xd, target)) for _ in range(3)]
        res.append((losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    (l1, s1), (l2, s2) = res
    assert l1 == l2, (l1, l2)  # same kernels, same launch order, deterministic weight gradients: bit-identical
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_iou_on_device_is_bit_identical_to_the_host_formulas():
    """ps_iou_from_confusion against loss.py:28-53 evaluated by numpy on the copied matrix: every class count the C-ABI takes, an empty matrix,
    empty rows / columns (0/0 -> 0 in the tissue IoU, dropped from the frequency-weighted sum)."""
    from pistoseg_amd.metrics import mIoUMask

    rng = np.random.RandomState(11)
    for nc in (1, 2, 3, 4, 5, 7, 8, 9, 12, 16):
        for case in range(4):
            m = mIoUMask(num_classes=nc)
            cm = rng.randint(0, 10 ** (3 + 2 * case), size=(nc, nc)).astype(np.int64)
            if case == 1 and nc > 1:
                cm[rng.randint(nc)] = 0
                cm[:, rng.randint(nc)] = 0
            if case == 3:
                cm[:] = 0
            m._cm = torch.from_numpy(cm.reshape(-1).copy()).to(D)
            got = m.iou_device().cpu().numpy()
            with np.errstate(all="ignore"):
                want = np.concatenate([[m.Mean_Intersection_over_Union(), m.Frequency_Weighted_Intersection_over_Union()], m.Tissue_Intersection_over_Union()])
            assert np.array_equal(got, want), (nc, case, got, want)


def _rfm_inputs(n, c, s, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, s, s, generator=g)
    pm = torch.randn(n, c - 1, 32, 32, generator=g)
    pc = torch.randn(n, c - 1, 32, 32, generator=g)
    lab = (torch.rand(n, c - 1, generator=g) < 0.5).float()
    lab[torch.arange(n), torch.randint(0, c - 1, (n,), generator=g)] = 1.0
    return x, pm, pc, lab


def test_stage3_reference_loop_against_native_trainer():
    """The body of the reference's `train_epoch` (revise_pseudo_labels.py:232-301) against the mirrors -- DataParallel-wrapped `Net`, autograd,
    `PolyOptimizer` over `get_parameter_groups()` exactly as :169-177 builds it, fused loss block (rfm_loss.rfm_loss_block) -- equals
    trainer.RFMTrainer after 3 steps; the oracle's eager torch loss block (the script's statements, on the CPU) gives the same losses on the same outputs."""
    from pistoseg_amd.optim import PolyOptimizer
    from pistoseg_amd.revise_net import Net
    from pistoseg_amd.rfm_loss import rfm_loss_block
    from pistoseg_amd.trainer import RFMTrainer

    c, n, s = 4, 2, 64
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    x, pm_fg, pc_fg, lab = _rfm_inputs(n, c, s, 21)
    xd = x.to(D)
    pmask = torch.cat([torch.zeros(n, 1, 32, 32), pm_fg], 1).to(D)
    pcam = torch.cat([torch.zeros(n, 1, 32, 32), pc_fg], 1).to(D)
    label = torch.cat([torch.ones(n, 1), lab], 1).to(D)
    drops, res = None, []
    lr, wd, max_step = 1e-3, 5e-4, 10
    for api in (True, False):
        net = Net(c, precision="fp32")
        net.load_state_dict(sd)
        net = net.to(D)
        net.train()
        net.launch.deterministic = True
        if drops is None:
            drops = [net.sample_dropout(n, D) for _ in range(3)]
        it = iter(drops)
        net.sample_dropout = lambda n_, dev_: next(it)
        losses = []
        if api:
            groups = net.get_parameter_groups()
            assert [len(g) for g in groups] == [35, 0, 5, 0]
            optimizer = PolyOptimizer([{"params": groups[0], "lr": lr, "weight_decay": wd}, {"params": groups[1], "lr": 2 * lr, "weight_decay": 0},
                                       {"params": groups[2], "lr": 10 * lr, "weight_decay": wd}, {"params": groups[3], "lr": 20 * lr, "weight_decay": 0}],
                                      lr=lr, weight_decay=wd, max_step=max_step)
            assert isinstance(optimizer, torch.optim.SGD) and optimizer.param_groups[0]["momentum"] == wd  # the reference's misplaced argument
            model = torch.nn.DataParallel(net, device_ids=[0]).to(D)
            model.train()
            for i in range(3):
                cam, cam_rv, pmask_rv, pcam_rv = model(xd, pmask, pcam)
                l, l_cls, l_rfm, l_ecr = rfm_loss_block(cam, cam_rv, pmask_rv, pcam_rv, pmask, pcam, label.view(n, c, 1, 1), deterministic=True)
                if i == 0:  # the script's own statements (oracle restatement) on the same outputs
                    ref = ref_cpu.rfm_losses(tuple(t.detach().cpu() for t in (cam, cam_rv, pmask_rv, pcam_rv)), pmask.cpu(), pcam.cpu(),
                                             label.view(n, c, 1, 1).cpu(), (s, s))
                    for a_, b_ in zip((l, l_cls, l_rfm, l_ecr), ref):
                        assert abs(float(a_) - float(b_)) <= 1e-5 * abs(float(b_)) + 1e-7, (float(a_), float(b_))
                losses.append([v.item() for v in (l, l_cls, l_rfm, l_ecr)])
                optimizer.zero_grad()
                l.backward()
                optimizer.step()
            assert optimizer.global_step == 3
            plan = optimizer.group_plan()
            assert [len(r) for _, r, _ in plan] == [1, 0, 1, 0] and all(not outside for _, _, outside in plan)  # two fused launches per step
            st = optimizer.state_dict()["state"]
            assert len(st) == 40 and all(set(v) == {"momentum_buffer"} for v in st.values())
        else:
            tr = RFMTrainer(net, lr=lr, wt_dec=wd, max_step=max_step, deterministic=True)
            for i in range(3):
                losses.append([float(v) for v in tr.train_step(xd, pmask, pcam, label)])
        res.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    (l1, s1), (l2, s2) = res
    for a_, b_ in zip(l1, l2):
        assert a_ == pytest.approx(b_, rel=1e-6, abs=1e-8), (l1, l2)
    for k in s1:
        if s1[k].is_floating_point():
            d = float((s1[k] - s2[k]).abs().max())
            assert d <= 1e-6 * float(s2[k].abs().max()) + 1e-9, (k, d)


@pytest.mark.selfcheck
def test_side_stream_weight_gradients_in_split_precision_survive_a_slow_side_stream():
    """in the split precisions the weight gradients read plain 16-bit companions of x / dY that are separate allocations; the
    caching allocator must not recycle them under the side stream.  A step whose side stream is held back by a long sleep kernel before every
    weight gradient equals the one-stream step bit for bit (deterministic weight gradients)."""
    from pistoseg_amd import ops
    from pistoseg_amd.seg_model import ResNet38dSeg

    sd = ref_cpu.make_state_dict(3, False, seed=42)
    x, *_ = make_inputs(2, 64, 4, 88)
    target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(3)).to(D)
    grads = []
    real_wgrad = ops.conv2d_wgrad
    for overlap in (False, True):
        model = ResNet38dSeg(3, "fp16x3")
        model.load_state_dict(sd)
        model = model.to(D)
        model.train()
        model.launch.deterministic = True
        model.overlap_wgrad = overlap
        drop = model.sample_dropout(2, D)
        if grads:
            drop = grads[0][1]
        model.sample_dropout = lambda n_, dev_, d=drop: d
        if overlap:
            def slow_wgrad(*a, **kw):  # runs with the side stream current: delay it so that main-stream allocations race ahead
                torch.cuda._sleep(20_000_000)
                return real_wgrad(*a, **kw)

            ops.conv2d_wgrad = slow_wgrad
        try:
            logits = model(x.to(D))
            _, dl = ops.softmax_ce(logits.detach(), target, 3, want_grad=True, grad_scale=1024.0)
            logits.backward(dl)
            torch.cuda.synchronize()
        finally:
            ops.conv2d_wgrad = real_wgrad
        grads.append(({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, drop))
    a, b = grads[0][0], grads[1][0]
    assert set(a) == set(b) and len(a) == 36
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.selfcheck
@pytest.mark.parametrize("kind", ["seg", "seg_fp16x3", "rfm"])
def test_deferred_optimizer_is_the_same_step(kind):
    """`defer_optimizer=True` (the update and the gradient zero fill on a side stream, overlapped with the next forward's frozen layers) runs the
    same kernels on the same data: weights after 4 steps are bit-identical to the strict trainer's (deterministic weight gradients), `state_dict()`
    waits for the pending update, and a slowed side stream changes nothing (the forward waits in front of the first trainable unit)."""
    from pistoseg_amd.revise_net import Net
    from pistoseg_amd.seg_model import ResNet38dSeg
    from pistoseg_amd.trainer import RFMTrainer, SegTrainer

    n, s, c = 2, 64, 4
    rfm = kind == "rfm"
    precision = "fp16x3" if kind == "seg_fp16x3" else "bf16"
    sd = ref_cpu.make_state_dict(c if rfm else 3, rfm, seed=42)
    x, pm_fg, pc_fg, lab = _rfm_inputs(n, c, s, 5)
    target = torch.randint(0, 4, (n, s, s), generator=torch.Generator().manual_seed(6)).to(D)
    pmask = torch.cat([torch.zeros(n, 1, 32, 32), pm_fg], 1).to(D)
    pcam = torch.cat([torch.zeros(n, 1, 32, 32), pc_fg], 1).to(D)
    label = torch.cat([torch.ones(n, 1), lab], 1).to(D)
    xd = x.to(D)
    drops, res = None, []
    for defer in (False, True):
        model = Net(c, precision=precision) if rfm else ResNet38dSeg(3, precision)
        model.load_state_dict(sd)
        model = model.to(D)
        model.train()
        if drops is None:
            drops = [model.sample_dropout(n, D) for _ in range(4)]
        it = iter(drops)
        model.sample_dropout = lambda n_, dev_: next(it)
        if rfm:
            tr = RFMTrainer(model, lr=1e-3, wt_dec=5e-4, max_step=10, deterministic=True, defer_optimizer=defer)
            step = lambda: tr.train_step(xd, pmask, pcam, label)[0]  # noqa: E731
        else:
            tr = SegTrainer(model, lr=2e-4, weight_decay=0.05, ignore_index=3, deterministic=True, defer_optimizer=defer)
            step = lambda: tr.train_step(xd, target)  # noqa: E731
        losses = []
        for i in range(4):
            losses.append(step())
            if defer and i == 1:  # hold the side stream back: the next forward must still see the new weights
                with torch.cuda.stream(tr.opt_stream):
                    torch.cuda._sleep(50_000_000)
        state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}  # (the pre-hook waits for the pending update)
        assert model._weights_event is None
        tr.settle()
        res.append(([float(v) for v in losses], state, tr.skipped_steps))
    (l1, s1, k1), (l2, s2, k2) = res
    assert l1 == l2 and k1 == k2 == 0, (l1, l2, k1, k2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_arena_adamw_under_torch_grad_scaler():
    """Lightning's `precision=16` plugin wraps the step in `torch.cuda.amp.GradScaler`: `scaler.scale(loss).backward(); scaler.step(opt); scaler.update()`.
    The scaler unscales `p.grad` in place (= the gradient arena) and skips `opt.step()` on overflow; with the fp16 model and `ArenaAdamW` that gives
    the native trainer's loss-scaled step (same scale, no overflow): weights after 2 steps equal `SegTrainer(loss_scale=...)`'s to f32 rounding of
    the unscale (the trainer divides inside the AdamW kernel, the scaler in a foreach pass)."""
    from pistoseg_amd.segmentation_module import SegmentationModule
    from pistoseg_amd.trainer import SegTrainer

    sd = ref_cpu.make_state_dict(3, False, seed=42)
    x, *_ = make_inputs(2, 64, 4, 55)
    target = torch.randint(0, 4, (2, 64, 64), generator=torch.Generator().manual_seed(8)).to(D)
    xd = x.to(D)
    scale = 1024.0
    res, drops = [], None
    for api in (True, False):
        if api:
            mod = SegmentationModule(make_args(precision="fp16", lr=2e-4)).to(D)
            model = mod.model
            model.load_state_dict(sd)
        else:
            model = _seg_model(sd, precision="fp16")
        model.train()
        model.launch.deterministic = True
        if drops is None:
            drops = [model.sample_dropout(2, D) for _ in range(2)]
        it = iter(drops)
        model.sample_dropout = lambda n_, dev_: next(it)
        if api:
            (opt,), _ = mod.configure_optimizers()
            scaler = torch.cuda.amp.GradScaler(init_scale=scale, growth_interval=10 ** 6)
            for i in range(2):
                loss = mod.training_step({"image": xd, "mask": target, "label": None}, i)
                opt.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            assert scaler.get_scale() == scale  # no overflow, no growth
        else:
            tr = SegTrainer(model, lr=2e-4, weight_decay=0.05, ignore_index=3, deterministic=True, loss_scale=scale)
            for _ in range(2):
                tr.train_step(xd, target)
        res.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in res[0]:
        if res[0][k].is_floating_point():
            d = (res[0][k] - res[1][k]).abs()
            assert float(d.max()) <= 2 * 2 * 2e-4 * 1.1 and float(d.mean()) < 1e-6, (k, float(d.max()), float(d.mean()))
