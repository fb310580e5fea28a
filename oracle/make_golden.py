"""Mint golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

Usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]

The reference's Python modules are imported from --ref (never copied); seeded inputs and the
deterministic weights of `oracle.ref_cpu.make_state_dict` are fed through the reference's own
`models.resnet38d.Net`, `models.revise_net.Net`, `loss.mIoUMask`, `utils.PolyOptimizer` and the
helper functions / loss block of `revise_pseudo_labels.py` and `infer_pseudo_masks.py`; outputs
are written as small .npz fixtures (sampled values + checksums for big tensors, full arrays for
small ones).  Third-party packages that are absent from this image (cv2, albumentations,
pytorch_lightning, smp, ttach, ...) are stubbed with MagicMock so the pure-torch helpers import;
nothing from those packages is executed.

`/root/reference` does not exist on the GPU box: the fixtures plus this script are what travels.
"""
from __future__ import annotations

import argparse
import inspect
import os
import sys
import textwrap
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu  # noqa: E402

STUB_ROOTS = {
    "cv2", "albumentations", "skimage", "torchvision", "pytorch_lightning", "segmentation_models_pytorch",
    "ttach", "timm", "matplotlib", "tqdm", "png", "mxnet", "monai", "efficientnet_pytorch", "torchmetrics",
}


class _StubFinder:
    """Resolve any import under an absent third-party root to a MagicMock module."""

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery

        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock()
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def sample_idx(numel: int, k: int = 512, seed: int = 7) -> np.ndarray:
    rs = np.random.RandomState(seed)
    return rs.randint(0, numel, size=min(k, numel)).astype(np.int64)


def summarize(t: torch.Tensor, prefix: str, out: dict, k: int = 512):
    """Store shape, float64 sum / abs-sum and k sampled entries of a big tensor."""
    a = t.detach().double().reshape(-1).numpy()
    idx = sample_idx(a.size, k)
    out[f"{prefix}.shape"] = np.array(t.shape, dtype=np.int64)
    out[f"{prefix}.sum"] = np.array(a.sum())
    out[f"{prefix}.abssum"] = np.abs(a).sum()
    out[f"{prefix}.idx"] = idx
    out[f"{prefix}.val"] = t.detach().reshape(-1).numpy()[idx].astype(np.float32)


def make_inputs(n, s, c, seed):
    """Seeded synthetic inputs shared by the generator and the tests (frozen legacy streams)."""
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((n, 3, s, s)).astype(np.float32))
    pmask = torch.from_numpy(rs.standard_normal((n, c - 1, 32, 32)).astype(np.float32))
    pcam = torch.from_numpy(rs.standard_normal((n, c - 1, 32, 32)).astype(np.float32))
    lab = (rs.uniform(size=(n, c - 1)) < 0.5).astype(np.float32)
    lab[np.arange(n), rs.randint(0, c - 1, size=n)] = 1.0
    return x, pmask, pcam, torch.from_numpy(lab)


def with_bg(pmask, pcam, lab):
    """revise_pseudo_labels.py:236-245 input plumbing (zero bg channel, bg label = 1)."""
    n, _, h, w = pmask.shape
    pm = torch.cat([torch.zeros((n, 1, h, w)), pmask], dim=1)
    pc = torch.cat([torch.zeros((n, 1, h, w)), pcam], dim=1)
    label = torch.cat((torch.ones((n, 1)), lab), dim=1).unsqueeze(2).unsqueeze(3)
    return pm, pc, label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    sys.path.insert(0, args.ref)
    import importlib.util

    for root in sorted(STUB_ROOTS):
        if importlib.util.find_spec(root) is not None:
            STUB_ROOTS.discard(root)  # really installed: use it
    sys.meta_path.append(_StubFinder())
    torch.manual_seed(0)

    import models.resnet38d as R38
    import models.revise_net as RV
    import loss as RLOSS

    nthreads = torch.get_num_threads()
    import revise_pseudo_labels as RPL
    import infer_pseudo_masks as IPM  # side effect: torch.set_num_threads(2)
    import utils as RUTILS

    torch.set_num_threads(nthreads)

    # ---------------------------------------------------------------- backbone
    sd = ref_cpu.make_state_dict(num_classes=None, rfm_heads=False, seed=42)
    net = R38.Net()
    assert list(net.state_dict().keys()) == list(sd.keys()), "state-dict key order drifted"
    net.load_state_dict(sd)
    net.eval()  # NB returns None (resnet38d.py:191-213)
    g = {}
    x, *_ = make_inputs(2, 32, 4, seed=100)
    with torch.no_grad():
        d = net.forward_as_dict(x)
    for k, v in d.items():
        summarize(v, k, g)
    np.savez_compressed(os.path.join(args.out, "backbone_s32.npz"), **g)
    print("backbone_s32", {k: tuple(v.shape) for k, v in d.items()})

    # ---------------------------------------------------------------- revise net forward (eval)
    for tag, n, s, c, seed in [("s64_c4", 2, 64, 4, 101), ("s224_c4", 1, 224, 4, 102), ("s256_c5", 1, 256, 5, 103)]:
        sd = ref_cpu.make_state_dict(num_classes=c, rfm_heads=True, seed=42)
        net = RV.Net(num_classes=c)
        assert list(net.state_dict().keys()) == list(sd.keys())
        net.load_state_dict(sd)
        net.eval()
        x, pmask, pcam, lab = make_inputs(n, s, c, seed)
        pm, pc, label = with_bg(pmask, pcam, lab)
        with torch.no_grad():
            outs = net(x, pm, pc)
            # infer_revise_masks.py:137-143
            masks = [torch.argmax((t * label)[:, 1:], dim=1) for t in (outs[2], outs[3], outs[1])]
        g = {}
        for name, t in zip(("cam", "cam_rv", "pmask_rv", "pcam_rv"), outs):
            summarize(t, name, g)
        for name, m in zip(("pmask_rv_mask", "pcam_rv_mask", "cam_rv_mask"), masks):
            g[name] = m.numpy().astype(np.uint8)
        # seg-model mask: argmax of softmax(cam) as loss.mIoUMask.forward does (loss.py:55-60)
        g["cam_mask"] = torch.argmax(torch.softmax(outs[0], dim=1), dim=1).byte().numpy()
        np.savez_compressed(os.path.join(args.out, f"revise_{tag}.npz"), **g)
        print("revise", tag, [tuple(t.shape) for t in outs])

    # ---------------------------------------------------------------- helpers
    rs = np.random.RandomState(200)
    g = {}
    cam = torch.from_numpy(rs.standard_normal((2, 4, 32, 32)).astype(np.float32))
    cam[0, 1, 3, 4] = cam[0, 2, 3, 4]  # a foreground tie
    g["in"] = cam.numpy()
    g["get_norm_cam_d"] = RV.Net.get_norm_cam_d(None, cam.clone()).numpy()
    g["max_norm"] = RPL.max_norm(cam.clone()).numpy()
    g["max_onehot"] = RPL.max_onehot(cam.clone()).numpy()
    g["adaptive_min_pooling_loss"] = RPL.adaptive_min_pooling_loss(cam[:, 1:].clone()).numpy()
    A = torch.softmax(torch.from_numpy(rs.standard_normal((2, 49, 49)).astype(np.float32)), dim=1)
    g["rfm_A"] = A.numpy()
    g["rfm_out"] = RV.Net.RFM(None, cam.clone(), A, 7, 7).numpy()
    lg = torch.from_numpy(rs.standard_normal((3, 224, 224)).astype(np.float32))
    g["interp_in_224"] = lg.numpy()
    g["interp_out_224"] = IPM.interpolate_tensor(lg, (32, 32)).numpy()
    lg = torch.from_numpy(rs.standard_normal((3, 256, 256)).astype(np.float32))
    g["interp_in_256"] = lg.numpy()
    g["interp_out_256"] = IPM.interpolate_tensor(lg, (32, 32)).numpy()
    np.savez_compressed(os.path.join(args.out, "helpers.npz"), **g)

    # ---------------------------------------------------------------- stage-2 mask reduction
    g = {}
    rs = np.random.RandomState(201)
    for i, lab in enumerate([[0, 1, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0, 1]]):
        lg = torch.from_numpy(rs.standard_normal((len(lab), 48, 48)).astype(np.float32) * 3)
        tissue = (rs.uniform(size=(48, 48)) > 0.2).astype(np.uint8) * 255
        m, e = IPM.get_mask_pred_and_entropy(lg.clone(), tissue, list(lab))
        g[f"c{i}.logit"] = lg.numpy()
        g[f"c{i}.tissue"] = tissue
        g[f"c{i}.label"] = np.array(lab, dtype=np.int64)
        g[f"c{i}.mask"] = np.asarray(m).astype(np.int64)
        g[f"c{i}.entropy"] = np.asarray(e).astype(np.float32)
    np.savez_compressed(os.path.join(args.out, "mask_reduce.npz"), **g)

    # ---------------------------------------------------------------- mIoU
    g = {}
    rs = np.random.RandomState(202)
    logits = torch.from_numpy(rs.standard_normal((2, 3, 40, 40)).astype(np.float32))
    gt = torch.from_numpy(rs.randint(0, 4, size=(2, 40, 40)).astype(np.int64))  # 3 = ignore
    m = RLOSS.mIoUMask(num_classes=3)
    miou, fw = m(logits, gt)
    g.update(logits=logits.numpy(), gt=gt.numpy(), cm=m.confusion_matrix, miou=np.array(miou), fwiou=np.array(fw),
             tissue_iou=m.Tissue_Intersection_over_Union())
    np.savez_compressed(os.path.join(args.out, "miou.npz"), **g)

    # ---------------------------------------------------------------- RFM loss block + gradients
    # Runs the reference's own loss statements (revise_pseudo_labels.py:253-282) by exec'ing the
    # slice of train_epoch's source between two marker lines, on CPU tensors.
    src = inspect.getsource(RPL.train_epoch).splitlines()
    i0 = next(i for i, l in enumerate(src) if "label_cam = F.adaptive_avg_pool2d" in l)
    i1 = next(i for i, l in enumerate(src) if l.strip() == "l = loss_cls + loss_rfm + loss_ecr")
    block = textwrap.dedent("\n".join(l for l in src[i0 : i1 + 1]))
    n, s, c = 2, 64, 4
    sd = ref_cpu.make_state_dict(num_classes=c, rfm_heads=True, seed=42)
    net = RV.Net(num_classes=c)
    net.load_state_dict(sd)
    net.train()
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0  # RNG parity with the device is impossible; goldens pin the routing
    x, pmask, pcam, lab = make_inputs(n, s, c, seed=104)
    pm, pc, label = with_bg(pmask, pcam, lab)
    outs = net(x, pm, pc)
    ns = dict(F=torch.nn.functional, torch=torch, cam=outs[0], cam_rv=outs[1], pmask_rv=outs[2], pcam_rv=outs[3],
              label=label, pmask=pm.clone(), pcam=pc.clone(), H=s, W=s, max_norm=RPL.max_norm, max_onehot=RPL.max_onehot,
              adaptive_min_pooling_loss=RPL.adaptive_min_pooling_loss)
    exec(block, ns)
    ns["l"].backward()
    g = {"loss": ns["l"].detach().numpy(), "loss_cls": ns["loss_cls"].detach().numpy(),
         "loss_rfm": ns["loss_rfm"].detach().numpy(), "loss_ecr": ns["loss_ecr"].detach().numpy()}
    named = dict(net.named_parameters())
    g["trainable"] = np.array(sorted(k for k, p in named.items() if p.requires_grad))
    g["has_grad"] = np.array(sorted(k for k, p in named.items() if p.grad is not None and p.grad.abs().sum() > 0))
    for k in ["fc8.weight", "f9_1.weight", "f9_2.weight", "f8_3.weight", "f8_4.weight", "b7.conv_branch2b1.weight",
              "b7.conv_branch1.weight", "b5_1.conv_branch2a.weight", "b4.conv_branch2a.weight", "b4.conv_branch1.weight",
              "b3.conv_branch2a.weight", "b3_1.conv_branch2b1.weight"]:
        summarize(named[k].grad, f"grad.{k}", g, k=256)
    groups = net.get_parameter_groups()
    g["param_group_sizes"] = np.array([len(x_) for x_ in groups])
    np.savez_compressed(os.path.join(args.out, "rfm_loss_grad_s64.npz"), **g)
    print("rfm loss", float(ns["l"]), "groups", [len(x_) for x_ in groups])

    # ---------------------------------------------------------------- PolyOptimizer
    g = {}
    rs = np.random.RandomState(203)
    p0 = torch.nn.Parameter(torch.from_numpy(rs.standard_normal((5, 7)).astype(np.float32)))
    p1 = torch.nn.Parameter(torch.from_numpy(rs.standard_normal((11,)).astype(np.float32)))
    g["p0_init"], g["p1_init"] = p0.detach().numpy().copy(), p1.detach().numpy().copy()
    opt = RUTILS.PolyOptimizer([{"params": [p0], "lr": 0.01, "weight_decay": 5e-4},
                                {"params": [p1], "lr": 0.1, "weight_decay": 0}], lr=0.01, weight_decay=5e-4, max_step=4)
    g["group0"] = np.array([opt.param_groups[0]["momentum"], opt.param_groups[0]["weight_decay"]])
    grads = []
    for step in range(6):
        g0 = torch.from_numpy(rs.standard_normal((5, 7)).astype(np.float32))
        g1 = torch.from_numpy(rs.standard_normal((11,)).astype(np.float32))
        grads.append((g0.numpy(), g1.numpy()))
        p0.grad, p1.grad = g0.clone(), g1.clone()
        opt.step()
        g[f"p0_step{step}"], g[f"p1_step{step}"] = p0.detach().numpy().copy(), p1.detach().numpy().copy()
        g[f"g0_step{step}"], g[f"g1_step{step}"] = grads[-1]
    np.savez_compressed(os.path.join(args.out, "poly_optimizer.npz"), **g)
    print("done ->", args.out)


if __name__ == "__main__":
    main()
