"""Mint the OEEM stage-0 golden by RUNNING THE REFERENCE ITSELF (build container only):
`OEEM/classification/network/wide_resnet.py` is imported from --ref (never copied), loaded with the deterministic weights of
`oracle.ref_cpu.wide_state_dict`, and `wideResNet.forward_cam` / `wideResNet.forward` are evaluated on seeded inputs.

Usage:  python oracle/make_golden_oeem.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu  # noqa: E402
from oracle.make_golden import make_inputs, summarize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    path = os.path.join(args.ref, "OEEM", "classification", "network", "wide_resnet.py")
    spec = importlib.util.spec_from_file_location("ref_wide_resnet", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = {}
    for c, n, s, seed in ((3, 2, 64, 301), (4, 1, 224, 302)):
        sd = ref_cpu.wide_state_dict(c, seed=42)
        net = mod.wideResNet(num_class=c)
        assert list(net.state_dict().keys()) == list(sd.keys()), "state-dict key order drifted"
        net.load_state_dict(sd)
        net.eval()
        x, *_ = make_inputs(n, s, 4, seed)
        with torch.no_grad():
            cam = net.forward_cam(x)
            cls = net.forward(x)
        summarize(cam, f"cam_c{c}_s{s}", g)
        g[f"cls_c{c}_s{s}"] = cls.numpy()
        print(f"c={c} s={s}: cam {tuple(cam.shape)} cls {cls.numpy().round(4).tolist()}")
    np.savez_compressed(os.path.join(args.out, "oeem_cam.npz"), **g)


if __name__ == "__main__":
    main()
