"""Stage 4 end to end (`infer_revise_masks.py:93-143`): the RFM net behind the reference's `nn.DataParallel` wrapper, its `module.`-prefixed
flat checkpoint (`revise_pseudo_labels.py:186,214`; reloaded AFTER wrapping, `infer_revise_masks.py:108-111`), and the `infer` loop over a
tile set cut into contiguous shards -- against the oracle's composition of the same statements on the CPU (fp32 parity path: masks bit-exact
up to ties below the output error)."""
import numpy as np
import pytest
import torch

from _parity import assert_tie_excused
from oracle import ref_cpu
from oracle.make_golden import make_inputs, with_bg

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
NAMES = ("pmask_rv", "pcam_rv", "cam_rv")


def oracle_stage4(sd, x, pmask, pcam, lab):
    """infer_revise_masks.py:119-143 on the CPU: zero background channel, background score 1, forward, (X_rv * label)[:, 1:] -> argmax."""
    pm, pc, label = with_bg(pmask, pcam, lab)
    with torch.no_grad():
        outs = ref_cpu.revise_forward(sd, x, pm, pc)
    masks = ref_cpu.revise_infer_masks(outs, label)
    scores = {"pmask_rv": (outs[2] * label)[:, 1:], "pcam_rv": (outs[3] * label)[:, 1:], "cam_rv": (outs[1] * label)[:, 1:]}
    return dict(zip(NAMES, masks)), scores


def assert_masks(tag, got, ref_masks, scores, err=2e-4):
    """Bit-exact except where the oracle's own top-2 scores are closer than the f32 output error (err: absolute, of maps in [0, 1])."""
    for name in NAMES:
        g, r = got[name].cpu().long(), ref_masks[name]
        assert g.dtype == torch.int64 and tuple(g.shape) == tuple(r.shape)
        diff = g != r
        top2 = torch.topk(scores[name], 2, dim=1)[0]
        gap = (top2[:, 0] - top2[:, 1]).abs()
        assert_tie_excused(f"stage-4 {tag} {name}", int(diff.sum()), diff.numel(), bool((gap[diff] <= 2 * err).all()))


@pytest.mark.parametrize("world", [1, 2, 3, 7])  # 7 ranks over 5 tiles: two EMPTY shards
def test_infer_revise_masks_sharded_matches_oracle(world):
    from pistoseg_amd import infer
    from pistoseg_amd.revise_net import Net

    c, s, T = 4, 64, 5  # 5 tiles: ragged last shard; batch_size 2: ragged last batch
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    model = Net(c, "fp32")
    model.load_state_dict(sd, strict=True)
    model = model.to(D)
    x, pmask, pcam, lab = make_inputs(T, s, c, seed=170)
    ref_masks, scores = oracle_stage4(sd, x, pmask, pcam, lab)
    got = {k: torch.full((T, s, s), 255, dtype=torch.uint8) for k in NAMES}
    covered = []
    for rank in range(world):  # every rank's call, one after the other on the test GPU; nothing is exchanged between them
        lo, hi, m_pmask, m_pcam, m_cam = infer.infer_revise_masks_sharded(model, x, pmask, pcam, lab, batch_size=2, rank=rank, world=world)
        covered.append((lo, hi))
        for k, m in zip(NAMES, (m_pmask, m_pcam, m_cam)):
            assert m.dtype == torch.uint8 and tuple(m.shape) == (hi - lo, s, s) and m.is_cuda
            got[k][lo:hi] = m.cpu()
    assert covered[0][0] == 0 and covered[-1][1] == T and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert_masks(f"world={world}", got, ref_masks, scores)
    for k in NAMES:  # values are foreground indices 0..C-2 (infer_revise_masks.py:137-143 drop the background channel before argmax)
        assert int(got[k].max()) <= c - 2


def test_dataparallel_wrapper_and_module_prefixed_checkpoint(tmp_path):
    """revise_pseudo_labels.py:186,214: `model = torch.nn.DataParallel(model)` ... `torch.save(model.state_dict(), 'ResNet38-RFM.pth')`;
    infer_revise_masks.py:108-111: `Net(...).cuda()` -> `DataParallel(model).cuda()` -> `load_state_dict(torch.load(ckpt))`.  The mirror must
    live inside that wrapper unchanged: 233 `module.*` keys, a strict reload after wrapping, forward through the wrapper, same masks."""
    from pistoseg_amd import infer
    from pistoseg_amd.revise_net import Net
    from pistoseg_amd.trainer import RFMTrainer

    c, s, T = 4, 64, 3
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    x, pmask, pcam, lab = make_inputs(T, s, c, seed=171)
    # stage 3: train one step inside the wrapper's module (the native trainer re-points the parameters at its arena), save the WRAPPER's state
    net = Net(c, "fp32")
    net.load_state_dict(sd, strict=False)  # revise_pseudo_labels.py:185
    net = net.to(D)
    wrapped = torch.nn.DataParallel(net).to(D)
    tr = RFMTrainer(wrapped.module, lr=0.01, wt_dec=5e-4, max_step=10)
    pm, pc, label = with_bg(pmask, pcam, lab)
    tr.train_step(x.to(D), pm.to(D), pc.to(D), label.reshape(T, c).to(D))
    state = wrapped.state_dict()
    assert len(state) == 233 and all(k.startswith("module.") for k in state)
    assert sorted(k[len("module."):] for k in state) == sorted(sd)
    path = str(tmp_path / "ResNet38-RFM.pth")
    torch.save(state, path)
    trained = {k[len("module."):]: v.detach().cpu().clone().contiguous() for k, v in state.items()}
    assert not torch.equal(trained["fc8.weight"], sd["fc8.weight"])  # the step moved the weights

    # stage 4: fresh net, wrap, THEN load the module.-prefixed checkpoint (strict), run the loop through the wrapper
    model = Net(num_classes=c, precision="fp32").to(D)
    model = torch.nn.DataParallel(model).to(D)
    res = model.load_state_dict(torch.load(path))
    assert not res.missing_keys and not res.unexpected_keys
    lo, hi, *masks = infer.infer_revise_masks_sharded(model, x, pmask, pcam, lab, batch_size=2)
    assert (lo, hi) == (0, T)
    ref_masks, scores = oracle_stage4(trained, x, pmask, pcam, lab)
    assert_masks("module. checkpoint through DataParallel", dict(zip(NAMES, masks)), ref_masks, scores)
    # the wrapper's forward is the module's forward (one visible device: DataParallel degenerates)
    with torch.no_grad():
        a = model(x.to(D), pm.to(D), pc.to(D))
        b = model.module(x.to(D), pm.to(D), pc.to(D))
    assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_stage4_bf16_fused_head_equals_unfused():
    """Stage-4 inference on the bf16 path takes the fused b7 + bn7 + fc8 launch (revise_net.py:50 without dropout): the `cam` output equals the
    unfused path's up to f32 summation order, and the masks of the three revised maps agree except where the normalised CAM's non-maximum
    suppression (discontinuous in the CAM) sits on a tie."""
    from pistoseg_amd.revise_net import Net

    c, s, T = 4, 96, 3
    sd = ref_cpu.make_state_dict(c, True, seed=42)
    model = Net(c, "bf16")
    model.load_state_dict(sd, strict=True)
    model = model.to(D)
    model.eval()
    x, pmask, pcam, lab = make_inputs(T, s, c, seed=172)
    pm, pc, _ = with_bg(pmask, pcam, lab)
    with torch.no_grad():
        fused = [o.cpu() for o in model(x.to(D), pm.to(D), pc.to(D))]
        model.fuse_head = False
        unfused = [o.cpu() for o in model(x.to(D), pm.to(D), pc.to(D))]
    model.fuse_head = True
    assert float((fused[0] - unfused[0]).abs().max()) <= 1e-5 * float(unfused[0].abs().max())
    for a, b in zip(fused[1:], unfused[1:]):
        assert float((a - b).abs().mean()) <= 1e-4 * float(b.abs().max())
