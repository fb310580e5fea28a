"""Mint the evaluation goldens by RUNNING THE REFERENCE'S OWN METHOD BODIES (build container only; test infrastructure).

Usage:  python oracle/make_golden_eval.py [--ref /root/reference] [--out tests/golden]

`models/segmentation_module.py` / `models/mosaic_module.py` cannot be imported here (their base class comes from
pytorch_lightning, which is absent), so the Lightning shells cannot be instantiated.  Their methods are ordinary functions of
`self`, though: this script parses the reference file, compiles the `FunctionDef`s of
  on_validation_epoch_start / validation_step / validation_epoch_end / training_step / training_epoch_end
(models/segmentation_module.py:96-251, models/mosaic_module.py:102-258) as they stand, and calls them with a stub `self` that
carries `args`, the reference's own `loss.mIoUMask`, canned logits as the "model" and a recording `log`.  The image sizes / ground
truths the methods read with PIL come from small PNGs written to a temp directory.  What is written (all data, no source):

  tests/golden/logged_keys.json   the metric keys each method logs, per module and dataset branch (the Lightning contract:
                                  ModelCheckpoint(monitor='validation_miou_mask_epoch'), segmentation_train.py:108-117)
  tests/golden/seg_eval.npz       sliding-window evaluation: per-image averaged probability canvases (f64), logged metric values,
                                  CE / mIoU values of training_step
  tests/golden/mxnet_key_map.json MXNet -> torch parameter names: the reference's convert_mxnet_to_torch (models/resnet38d.py:215-263) run on a
                                  stand-in `mxnet.nd.load` that returns one tagged array per parameter name of the ResNet38 ImageNet checkpoint
  tests/golden/oeem_ms_cam.npz    OEEM stage 0 multi-scale CAM accumulation (OEEM/classification/prepare_seg_inputs.py:96-138):
                                  the statements of the per-image loop body exec'ed with a stub `net_cam` returning canned CAM crops

The seeded input generators (`eval_case`, `oeem_case`) are shared with the tests, which regenerate identical inputs.
"""
from __future__ import annotations

import argparse
import ast
import contextlib
import io
import json
import os
import sys
import tempfile
import textwrap
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------------------------------- seeded inputs (shared with tests)
def eval_case(num_classes: int = 3, seed: int = 11, tile: int = 32, stride: int = 20, ignore: bool = True):
    """Two images x two scales of overlapping `tile`-px tiles with ragged right / bottom tiles, named as the reference's
    ValidationDataset names them ("{idx}_{scale}_{y}_{x}-[labels].png").  Returns (sizes {idx: (w, h)}, batches, gt {idx: uint8 [h, w]})
    (patch masks include the value `num_classes` = ignored pixels iff `ignore`: the bcss CE has no ignore_index)
    where batches = list of (logits [n, C, tile, tile] f32, mask [n, tile, tile] int64, names, original_h, original_w)."""
    rs = np.random.RandomState(seed)
    sizes = {"1001": (90, 70), "1002": (64, 48)}
    names, oh, ow = [], [], []
    for idx, (w, h) in sizes.items():
        for scale in (1.0, 0.75):
            w_, h_ = int(w * scale), int(h * scale)
            ys = sorted(set(list(range(0, max(h_ - tile, 0) + 1, stride)) + [max(h_ - tile, 0)]))
            xs = sorted(set(list(range(0, max(w_ - tile, 0) + 1, stride)) + [max(w_ - tile, 0)]))
            for y in ys:
                for x in xs:
                    names.append(f"{idx}_{scale}_{y}_{x}-[1, 0, 1].png")
                    oh.append(min(tile, h_ - y))
                    ow.append(min(tile, w_ - x))
    logits = torch.from_numpy((rs.standard_normal((len(names), num_classes, tile, tile)) * 3).astype(np.float32))
    masks = torch.from_numpy(rs.randint(0, num_classes + (1 if ignore else 0), size=(len(names), tile, tile)).astype(np.int64))
    gt = {k: rs.randint(0, 4, size=(h, w)).astype(np.uint8) for k, (w, h) in sizes.items()}
    bs = 5
    batches = [(logits[i:i + bs], masks[i:i + bs], names[i:i + bs], oh[i:i + bs], ow[i:i + bs]) for i in range(0, len(names), bs)]
    return sizes, batches, gt


def oeem_case(num_of_class: int = 3, seed: int = 12, side_length: int = 56, stride: int = 28):
    """One image, three scales of overlapping CAM crops (the last row of crops dropped: uncovered pixels hit the counter clamp).
    Returns (w, h, scales, cam_crops per scale [K, C, side, side] f32, positions per scale)."""
    rs = np.random.RandomState(seed)
    w, h = 90, 120  # the reference's naming: w, h, _ = orig_img.shape
    scales = [1.0, 1.25, 0.8]
    crops, poss = [], []
    for s in scales:
        w_, h_ = int(w * s), int(h * s)
        ys = sorted(set(list(range(0, w_ - side_length + 1, stride)) + [w_ - side_length]))[:-1]
        xs = sorted(set(list(range(0, h_ - side_length + 1, stride)) + [h_ - side_length]))
        pos = [(y, x) for y in ys for x in xs]
        crops.append(torch.from_numpy(rs.standard_normal((len(pos), num_of_class, side_length, side_length)).astype(np.float32)))
        poss.append(pos)
    return w, h, scales, crops, poss


# --------------------------------------------------------------------------------------------------- reference method extraction
def method_functions(path: str, class_name: str, names, namespace: dict) -> dict:
    """Compile the named methods of `class_name` in the reference file at `path` as plain functions inside `namespace`."""
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == class_name)
    out = {}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
            out[node.name] = namespace[node.name]
    missing = set(names) - set(out)
    assert not missing, f"{class_name} lost methods {missing}"
    return out


class StubSelf:
    """What the method bodies touch on `self`: args, the three mIoUMask meters, the loss module, the model (canned logits), log."""

    def __init__(self, args, miou_cls, loss_mod, attr):
        self.args = args
        self.train_iou, self.valid_iou, self.test_iou = (miou_cls(num_classes=args.num_classes) for _ in range(3))
        setattr(self, attr, loss_mod)
        self.logged = []
        self._next = None
        self.model = self  # training_step of SegmentationModule calls self.model(...), MosaicModule calls self(...)

    def __call__(self, image):
        return self._next

    def log(self, name, value, prog_bar=False, **kw):
        self.logged.append((name, float(value), bool(prog_bar)))


def run_shell(ref: str, module_file: str, class_name: str, dataset: str, num_classes: int, workdir: str, loss_mod, loss_attr):
    from PIL import Image

    import loss as RLOSS  # the reference's loss.py (sys.path)

    ns = dict(torch=torch, np=np, F=torch.nn.functional, Image=Image, os=os, mIoUMask=RLOSS.mIoUMask)
    fns = method_functions(os.path.join(ref, "models", module_file), class_name,
                           ["on_validation_epoch_start", "validation_step", "validation_epoch_end", "training_step", "training_epoch_end"], ns)
    sizes, batches, gt = eval_case(num_classes, ignore=dataset == "wsss4luad")
    os.makedirs(os.path.join(workdir, "img"), exist_ok=True)
    os.makedirs(os.path.join(workdir, "mask"), exist_ok=True)
    for k, (w, h) in sizes.items():
        Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(os.path.join(workdir, "img", k + ".png"))
        Image.fromarray(gt[k]).save(os.path.join(workdir, "mask", k + ".png"))
    args = Namespace(dataset=dataset, num_classes=num_classes, val_data=os.path.join(workdir, "patches"), pseudo_mask_dir="-", mosaic_data="-",
                     log_path="-", patch_size=32)
    self = StubSelf(args, RLOSS.mIoUMask, loss_mod, loss_attr)
    keys, vals = {}, {}
    with contextlib.redirect_stdout(io.StringIO()):
        # ---- training_step on the first batch (CE path only has a reference-defined loss; Dice is third-party)
        if loss_mod is not None:
            lg, mk = batches[0][0].clone().requires_grad_(True), batches[0][1]
            self._next = lg
            loss = fns["training_step"](self, {"image": None, "mask": mk, "label": None}, 0)
            loss.backward()
            vals["train_loss"] = loss.detach().numpy()
            vals["train_dlogits"] = lg.grad.numpy()
            keys["training_step"] = [k for k, _, _ in self.logged]
            vals["train_logged"] = np.array([v for _, v, _ in self.logged])
            self.logged = []
            fns["training_epoch_end"](self, None)
            keys["training_epoch_end"] = [k for k, _, _ in self.logged]
            self.logged = []
        # ---- validation epoch
        fns["on_validation_epoch_start"](self)
        for i, (lg, mk, names, oh, ow) in enumerate(batches):
            self._next = lg
            fns["validation_step"](self, (lg, mk, names, torch.tensor(oh), torch.tensor(ow)), i)
        canv = {k: v.copy() for k, v in getattr(self, "pred_big_mask_dict_ms", {}).items()}
        fns["validation_epoch_end"](self, None)
    keys["validation_epoch_end"] = [k for k, _, _ in self.logged]
    vals["val_logged"] = np.array([v for _, v, _ in self.logged])
    vals["val_prog_bar"] = np.array([p for _, _, p in self.logged])
    if dataset == "wsss4luad" and class_name == "SegmentationModule":  # (the Mosaic shell runs the same statements: metric values only)
        for k, v in canv.items():
            vals[f"ms_sum_total.{k}"] = v.sum(axis=(0, 1))  # per-(image, scale) SUM canvases before the division: channel totals
        for k, v in self.pred_big_mask_dict.items():
            vals[f"big.{k}"] = v                         # scale-averaged probabilities [h, w, 3] after `/= cnt`
    return keys, vals


def run_oeem(ref: str):
    """Exec the per-image statements of prepare_seg_inputs.py (from `ensemble_cam = np.zeros(...)` to the final 32x32 interpolate) with
    a stub `net_cam` that returns canned CAM crops.  `.cuda()` is stripped from the one statement that has it (no GPU here)."""
    path = os.path.join(ref, "OEEM", "classification", "prepare_seg_inputs.py")
    src = open(path).read().splitlines()
    i0 = next(i for i, l in enumerate(src) if l.strip().startswith("ensemble_cam = np.zeros((num_of_class, w, h))"))
    i1 = next(i for i, l in enumerate(src) if "ensemble_cam = F.interpolate(torch.unsqueeze(torch.tensor(ensemble_cam),0), (32, 32)" in l)
    block = textwrap.dedent("\n".join(src[i0:i1 + 1])).replace("ims.cuda()", "ims")
    c, side = 3, 56
    w, h, scales, crops, poss = oeem_case(c, side_length=side)

    class _Net:  # net_cam.module.forward_cam(ims): `ims` carries the indices of the crops it stands for
        def __init__(self):
            self.module, self.s = self, 0

        def forward_cam(self, ims):
            return crops[self.s][ims.long().reshape(-1)]

    net = _Net()
    # scaled_im_list[s] = list of [1]-shaped index tensors so that torch.vstack / torch.split work as in the reference
    scaled_im_list = [[torch.tensor([[float(k)]]) for k in range(len(p))] for p in poss]

    class _Scales(list):  # the loop reads scales[s]; advance the stub's scale cursor with it
        def __getitem__(self, i):
            net.s = i
            return list.__getitem__(self, i)

    ns = dict(np=np, torch=torch, F=torch.nn.functional, num_of_class=c, w=w, h=h, scales=_Scales(scales), side_length=side,
              scaled_im_list=scaled_im_list, scaled_position_list=poss, batch_size=4, net_cam=net)
    exec(block, ns)
    return {"ensemble_cam": ns["ensemble_cam"]}


def mxnet_param_names():
    """The parameter names of the MXNet ResNet38 ImageNet checkpoint, written out from its architecture (unit `<stage>a`, `<stage>b<k>`;
    branch2a / branch2b1 [/ branch2b2] / branch1; BatchNorm gamma, beta + moving_mean, moving_var) -- shared with the tests."""
    names = ["arg:conv1a_weight", "arg:linear1000_weight", "arg:linear1000_bias"]
    units = {2: 2, 3: 2, 4: 5, 5: 2, 6: 0, 7: 0}  # stage -> number of b<k> units after the a unit
    for st, extra in units.items():
        for u in ["a"] + [f"b{k}" for k in range(1, extra + 1)]:
            branches = ["branch2a", "branch2b1"] + (["branch2b2"] if st >= 6 else [])
            for br in branches:
                names.append(f"arg:res{st}{u}_{br}_weight")
                names += [f"arg:bn{st}{u}_{br}_gamma", f"arg:bn{st}{u}_{br}_beta", f"aux:bn{st}{u}_{br}_moving_mean", f"aux:bn{st}{u}_{br}_moving_var"]
            if u == "a":
                names.append(f"arg:res{st}{u}_branch1_weight")
    names += ["arg:bn7_gamma", "arg:bn7_beta", "aux:bn7_moving_mean", "aux:bn7_moving_var"]
    return names


def run_mxnet_names(ref: str):
    """models/resnet38d.py:215-263 executed with a stand-in `mxnet` module whose nd.load returns one tagged array per name."""
    import types

    import models.resnet38d as R38

    class _Arr:
        def __init__(self, i):
            self.i = i

        def asnumpy(self):
            return np.array([float(self.i)], dtype=np.float32)

    names = mxnet_param_names()
    fake = types.ModuleType("mxnet")
    fake.nd = types.SimpleNamespace(load=lambda fn: {n: _Arr(i) for i, n in enumerate(names)})
    sys.modules["mxnet"] = fake
    try:
        out = R38.convert_mxnet_to_torch("unused.params")
    finally:
        del sys.modules["mxnet"]
    inv = {int(v.item()): k for k, v in out.items()}
    return {n: inv.get(i) for i, n in enumerate(names)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    sys.path.insert(0, a.ref)
    torch.manual_seed(0)
    all_keys, npz = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for module_file, cls, attr, lossf in (("segmentation_module.py", "SegmentationModule", "train_ce", True),
                                              ("mosaic_module.py", "MosaicModule", "train_dice", False)):
            for dataset, c in (("wsss4luad", 3), ("bcss", 4)):
                loss_mod = None
                if lossf:  # segmentation_module.py:63-66
                    loss_mod = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=3) if dataset == "wsss4luad" else torch.nn.CrossEntropyLoss(reduction="none")
                keys, vals = run_shell(a.ref, module_file, cls, dataset, c, os.path.join(tmp, f"{cls}_{dataset}"), loss_mod, attr)
                all_keys[f"{cls}.{dataset}"] = keys
                for k, v in vals.items():
                    npz[f"{cls}.{dataset}.{k}"] = v
                print(cls, dataset, {m: len(v) for m, v in keys.items()})
    # the Mosaic shell's training_step only differs by the third-party Dice loss: record its log keys from the source text
    tree = ast.parse(open(os.path.join(a.ref, "models", "mosaic_module.py")).read())
    mcls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MosaicModule")
    for fn in mcls.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("training_step", "training_epoch_end"):
            ks = [c.args[0].value if isinstance(c.args[0], ast.Constant) else c.args[0].values[0].value
                  for c in ast.walk(fn) if isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == "log"]
            for ds in ("wsss4luad", "bcss"):
                all_keys[f"MosaicModule.{ds}"][fn.name] = ks
    with open(os.path.join(a.out, "logged_keys.json"), "w") as f:
        json.dump(all_keys, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(a.out, "seg_eval.npz"), **npz)
    np.savez_compressed(os.path.join(a.out, "oeem_ms_cam.npz"), **run_oeem(a.ref))
    with open(os.path.join(a.out, "mxnet_key_map.json"), "w") as f:
        json.dump(run_mxnet_names(a.ref), f, indent=0, sort_keys=True)
    print("wrote logged_keys.json, seg_eval.npz, oeem_ms_cam.npz ->", a.out)


if __name__ == "__main__":
    main()
