"""OEEM stage 0 on the GPU: the wideResNet mirror against the reference golden (minted by
oracle/make_golden_oeem.py from OEEM/classification/network/wide_resnet.py) and the multi-scale sliding-window CAM loop
(prepare_seg_inputs.py:96-138) against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.make_golden import make_inputs

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def build(c, precision="fp32"):
    from pistoseg_amd.oeem import wideResNet

    sd = ref_cpu.wide_state_dict(c, seed=42)
    net = wideResNet(num_class=c, precision=precision)
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net = net.to(D)
    net.eval()
    return sd, net


@pytest.mark.parametrize("c,n,s,seed", [(3, 2, 64, 301), (4, 1, 224, 302)])
def test_forward_cam_and_cls_match_reference_golden(golden_dir, c, n, s, seed):
    g = np.load(os.path.join(golden_dir, "oeem_cam.npz"))
    sd, net = build(c)
    x, *_ = make_inputs(n, s, 4, seed)
    cam = net.forward_cam(x.to(D)).cpu()
    assert tuple(cam.shape) == tuple(g[f"cam_c{c}_s{s}.shape"])
    got = cam.reshape(-1)[torch.from_numpy(g[f"cam_c{c}_s{s}.idx"])]
    ref = torch.from_numpy(g[f"cam_c{c}_s{s}.val"])
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    assert abs(float(cam.double().abs().sum()) - float(g[f"cam_c{c}_s{s}.abssum"])) < 1e-4 * float(g[f"cam_c{c}_s{s}.abssum"])
    cls = net(x.to(D)).cpu().numpy()
    assert np.abs(cls - g[f"cls_c{c}_s{s}"]).max() < 1e-4 * np.abs(g[f"cls_c{c}_s{s}"]).max()


def test_multi_scale_image_cam_matches_oracle():
    """One 100 x 88 image, three scales, overlapping 64 x 64 crops on a stride-32 grid (the smallest scale is narrower than the
    crop, which exercises interpolatex < side_length): [C, 32, 32] f64 result within 1e-4 of the oracle's (fp32 net + f64 canvases)."""
    from pistoseg_amd.oeem import image_cam_32x32

    c, side = 3, 64
    sd, net = build(c)
    w, h = 100, 88
    scales = [1.0, 1.25, 0.6]
    rs = np.random.RandomState(17)
    im_lists, pos_lists = [], []
    for sc in scales:
        w_, h_ = int(w * sc), int(h * sc)
        ys = sorted(set(list(range(0, max(w_ - side, 0) + 1, 32)) + [max(w_ - side, 0)]))
        xs = sorted(set(list(range(0, max(h_ - side, 0) + 1, 32)) + [max(h_ - side, 0)]))
        pos = [(y, x) for y in ys for x in xs]
        im_lists.append(torch.from_numpy(rs.standard_normal((len(pos), 3, side, side)).astype(np.float32)))
        pos_lists.append(pos)
    with torch.no_grad():
        ref = ref_cpu.image_cam_32x32(sd, im_lists, pos_lists, scales, (w, h), side, c)
    got = image_cam_32x32(net, im_lists, pos_lists, scales, (w, h), side, batch_size=3).cpu().numpy()
    assert got.shape == ref.shape == (c, 32, 32) and got.dtype == np.float64
    assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()


def test_wide_net_bf16_runs_and_is_close():
    sd, net = build(3, "bf16")
    x, *_ = make_inputs(2, 64, 4, 301)
    cam = net.forward_cam(x.to(D)).cpu()
    with torch.no_grad():
        ref = ref_cpu.wide_forward_cam(sd, x)
    assert float((cam - ref).abs().max()) < 6e-2 * float(ref.abs().max())
