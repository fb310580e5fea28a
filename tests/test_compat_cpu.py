"""The drop-in import shim (pistoseg_amd/compat): the reference's stage scripts' own import lines resolve to the MI355X mirrors with
only PYTHONPATH changed; the oracle's evaluation restatements agree with the fixtures minted from the reference's method bodies."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_scripts_import_lines_resolve_to_the_mirrors():
    """segmentation_train.py:8, mosaic_train.py:14, infer_pseudo_masks.py:17, revise_pseudo_labels.py:4,25,28,
    infer_revise_masks.py:17,20, segmentation_test.py:10,13,14,29 -- verbatim, with PYTHONPATH=pistoseg_amd/compat only."""
    code = "\n".join([
        "from models.segmentation_module import SegmentationModule",
        "from models.mosaic_module import MosaicModule",
        "from models.revise_net import Net",
        "from models.net_cls import NetCLS",
        "import models.resnet38d",
        "from loss import mIoUMask",
        "import utils",
        "import pistoseg_amd",
        "assert SegmentationModule.__module__ == 'pistoseg_amd.segmentation_module' and MosaicModule.__module__ == SegmentationModule.__module__",
        "assert Net.__module__ == 'pistoseg_amd.revise_net' and mIoUMask.__module__ == 'pistoseg_amd.metrics'",
        "assert utils.PolyOptimizer.__module__ == 'pistoseg_amd.arena' and models.resnet38d.Net.__module__ == 'pistoseg_amd.resnet38d'",
        "net = Net(num_classes=4); assert len(net.state_dict()) == 233 and net.eval() is None",
        "print('ok')",
    ])
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "pistoseg_amd", "compat"))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_utils_shim_overlays_a_utils_module_further_down_the_path(tmp_path):
    """With the reference tree behind the shim on sys.path its host-side helpers stay visible; only PolyOptimizer is replaced."""
    (tmp_path / "utils.py").write_text("def get_background(region):\n    return 'host helper'\nclass PolyOptimizer: pass\n")
    (tmp_path / "loss.py").write_text("class DiceLoss: pass\nclass mIoUMask: pass\n")
    code = ("import utils, loss; assert utils.get_background(None) == 'host helper'; assert utils.PolyOptimizer.__module__ == 'pistoseg_amd.arena';"
            "assert loss.mIoUMask.__module__ == 'pistoseg_amd.metrics' and hasattr(loss, 'DiceLoss'); print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "pistoseg_amd", "compat"), str(tmp_path)]))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_create_model_plug_point_handles_smp_names():
    """run.sh's default model names are smp architectures: without smp they are refused with a pointer to --model ResNet38d, or replaced
    when PISTOSEG_SUBSTITUTE_MODEL=1; `ResNet38d` always gives the in-tree model."""
    import importlib.util

    import pytest

    from pistoseg_amd.seg_model import ResNet38dSeg, create_model

    assert isinstance(create_model("ResNet38d", encoder_name="resnet38d", classes=4), ResNet38dSeg)
    if importlib.util.find_spec("segmentation_models_pytorch") is not None:
        pytest.skip("smp installed: names are delegated to it")
    os.environ.pop("PISTOSEG_SUBSTITUTE_MODEL", None)
    with pytest.raises(ValueError, match="ResNet38d"):
        create_model("UnetPlusPlus", encoder_name="efficientnet-b0", classes=3, decoder_attention_type="scse")
    os.environ["PISTOSEG_SUBSTITUTE_MODEL"] = "1"
    try:
        with pytest.warns(UserWarning):
            m = create_model("UnetPlusPlus", encoder_name="efficientnet-b0", classes=3, decoder_attention_type="scse")
        assert isinstance(m, ResNet38dSeg) and m.classes == 3
    finally:
        os.environ.pop("PISTOSEG_SUBSTITUTE_MODEL", None)


def test_convert_mxnet_to_torch_matches_reference_key_map():
    """`convert_mxnet_to_torch` (models/resnet38d.py:215-263; ImageNet `.params` init of stage 3, revise_pseudo_labels.py:179-181) against the
    name map minted by running the reference's function on a stand-in mxnet loader (tests/golden/mxnet_key_map.json): every parameter of the
    MXNet checkpoint lands on the same state-dict key, the 1000-way classifier is dropped, and the result fills every backbone tensor."""
    import json

    import torch

    from oracle.make_golden_eval import mxnet_param_names
    from pistoseg_amd.resnet38d import Net, convert_mxnet_to_torch, mxnet_key_to_torch

    want = json.load(open(os.path.join(ROOT, "tests", "golden", "mxnet_key_map.json")))
    names = mxnet_param_names()
    assert sorted(names) == sorted(want)
    assert {n: mxnet_key_to_torch(n) for n in names} == want

    class Arr:  # what mxnet.nd.load hands back: objects with .asnumpy()
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a

    net = Net(precision="fp32")
    sd = net.state_dict()
    rs = np.random.RandomState(0)
    src = {n: Arr(rs.standard_normal(tuple(sd[t].shape)).astype(np.float32)) for n, t in want.items() if t is not None}
    src["arg:linear1000_weight"] = Arr(np.zeros((1000, 4096), np.float32))
    out = convert_mxnet_to_torch(src)
    assert set(out) == {t for t in want.values() if t is not None}
    res = net.load_state_dict(out, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    k = "b5_1.conv_branch2a.weight"
    assert torch.equal(net.state_dict()[k], torch.from_numpy(src["arg:res5b1_branch2a_weight"].a))


def test_rfm_net_checkpoint_contract_under_dataparallel_wrapper(tmp_path):
    """Stage 3 saves the state of a `nn.DataParallel` wrapper (`revise_pseudo_labels.py:186,214`): a FLAT dict of 233 `module.*` keys;
    stage 4 wraps a fresh net and only then loads it (`infer_revise_masks.py:108-111`).  Host-side contract of the mirror (no GPU needed):
    same key set as the oracle's state dict behind the prefix, strict reload after wrapping, values round-trip, `.eval()` of the wrapper
    reaches the module's quirky `train()` (returns None for the module, the wrapper itself for the wrapper)."""
    import torch

    from oracle import ref_cpu
    from pistoseg_amd.revise_net import Net

    sd = ref_cpu.make_state_dict(4, True, seed=42)
    net = Net(num_classes=4, precision="fp32")
    net.load_state_dict(sd, strict=False)
    wrapped = torch.nn.DataParallel(net)
    state = wrapped.state_dict()
    assert len(state) == 233 and sorted(state) == sorted("module." + k for k in sd)
    path = str(tmp_path / "ResNet38-RFM.pth")
    torch.save(state, path)
    fresh = torch.nn.DataParallel(Net(num_classes=4, precision="fp32"))
    res = fresh.load_state_dict(torch.load(path))
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd.items():
        assert torch.equal(fresh.module.state_dict()[k], v), k
    assert fresh.eval() is fresh and not fresh.module.training and fresh.module.eval() is None
    # a checkpoint saved WITHOUT the wrapper is refused by the wrapped net (and vice versa): the prefix is part of the layout
    import pytest

    with pytest.raises(RuntimeError):
        fresh.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        Net(num_classes=4, precision="fp32").load_state_dict(state)


def test_deterministic_switch_follows_torch_unless_forced():
    """`ops.DETERMINISTIC = None` means: do what the reference's own switches say -- `torch.use_deterministic_algorithms(True)`
    (revise_pseudo_labels.py:140-146) / `pl.Trainer(deterministic=True)` (segmentation_train.py:153-160) both set torch's flag; True / False
    force it, an explicit per-call override wins."""
    import torch

    from pistoseg_amd import ops

    prev_flag, prev = torch.are_deterministic_algorithms_enabled(), ops.DETERMINISTIC
    try:
        ops.DETERMINISTIC = None
        torch.use_deterministic_algorithms(False)
        assert ops.deterministic_enabled() is False
        torch.use_deterministic_algorithms(True, warn_only=True)
        assert ops.deterministic_enabled() is True and ops.deterministic_enabled(False) is False
        ops.DETERMINISTIC = False
        assert ops.deterministic_enabled() is False and ops.deterministic_enabled(True) is True
        ops.DETERMINISTIC = True
        torch.use_deterministic_algorithms(False)
        assert ops.deterministic_enabled() is True
    finally:
        ops.DETERMINISTIC = prev
        torch.use_deterministic_algorithms(prev_flag)


def test_launch_options_are_per_model_state_and_reach_the_geometry(monkeypatch):
    """`tiles_per_block` / `gpu_shared` are fields of every `ps_conv_geom` (include/pistoseg_hip.h).  On the host they are state of the CALLER:
    each model owns an `ops.LaunchOpts` that its plans pass to every launch (a field left at None falls back to the module default, which only
    direct users of `ops` -- op tests, probes -- change); the backbone's two-stream backward sets `gpu_shared` on ITS model for its own duration
    only (also when a launch raises), the gradient reducer `tiles_per_block` on ITS model -- a second model in the process sees neither."""
    import torch

    from pistoseg_amd import ops
    from pistoseg_amd.dist import BucketedAllReduce
    from pistoseg_amd.resnet38d import Net

    spec = ops.ConvSpec(512, 512, 3, 1, 1)
    assert ops._geom(spec, 1, 2, 28, 28, 512, 512).gpu_shared == 0
    monkeypatch.setattr(ops, "GPU_SHARED", 1)  # module default: what a None field means
    g = ops._geom(spec, 1, 2, 28, 28, 512, 512, ops.LaunchOpts())
    assert g.gpu_shared == 1 and g.tiles_per_block == ops.TILES_PER_BLOCK
    g = ops._geom(spec, 1, 2, 28, 28, 512, 512, ops.LaunchOpts(tiles_per_block=3, gpu_shared=0))
    assert g.gpu_shared == 0 and g.tiles_per_block == 3
    monkeypatch.setattr(ops, "GPU_SHARED", 0)

    net, other = Net(), Net()
    assert net.launch is not other.launch
    seen = []

    def fake_units(self, saved, G, grads, g_taps, after_unit, wgrad_stream, first, dt, dev, n):
        seen.append((self.launch.gpu_shared, other.launch.gpu_shared))
        raise RuntimeError("launch failed")

    monkeypatch.setattr(Net, "_backward_units", fake_units)
    monkeypatch.setattr(Net, "refresh_dgrad_weights", lambda self: None)

    class Saved:
        n = 1

    class FakeStream:  # stands for the side stream; never used because the unit loop is replaced
        pass

    G = torch.zeros(1, 4, 4, 8)
    for stream, expect in ((None, None), (FakeStream(), 1)):
        try:
            net.backward_backbone(Saved(), G, {}, wgrad_stream=stream)
        except RuntimeError as e:
            assert "launch failed" in str(e)
        assert seen[-1] == (expect, None) and net.launch.gpu_shared is None and ops.GPU_SHARED == 0

    # the reducer switches ITS model's tiles_per_block (device arenas only; emulated here by handing it a stand-in comm stream)
    red = BucketedAllReduce(torch.zeros(8), [("b7", 0, 8)], None, shared_tiles_per_block=2, launch_opts=net.launch)
    red.comm_stream = object()
    red._share_gpu(True)
    assert net.launch.tiles_per_block == 2 and other.launch.tiles_per_block is None and ops.TILES_PER_BLOCK == 0
    red._share_gpu(False)
    assert net.launch.tiles_per_block is None
    # ... or, share="reserve", its cus_reserved: grids sized for the CUs the collective leaves
    red = BucketedAllReduce(torch.zeros(8), [("b7", 0, 8)], None, launch_opts=net.launch, share="reserve", reserved_cus=32)
    red.comm_stream = object()
    red._share_gpu(True)
    assert net.launch.cus_reserved == 32 and net.launch.tiles_per_block is None and other.launch.cus_reserved is None
    assert ops._geom(spec, 1, 2, 28, 28, 512, 512, net.launch).cus_reserved == 32 and ops._geom(spec, 1, 2, 28, 28, 512, 512, other.launch).cus_reserved == 0
    red._share_gpu(False)
    assert net.launch.cus_reserved is None
    # ... or, share="queue" / "reserve+queue", its tile_queue: the persistent kernels' blocks draw their tiles from ticket counters
    for mode, res in (("queue", None), ("reserve+queue", 32)):
        red = BucketedAllReduce(torch.zeros(8), [("b7", 0, 8)], None, launch_opts=net.launch, share=mode, reserved_cus=32)
        red.comm_stream = object()
        red._share_gpu(True)
        assert net.launch.tile_queue == 1 and net.launch.cus_reserved == res and net.launch.tiles_per_block is None and other.launch.tile_queue is None
        gq = ops._geom(spec, 1, 2, 28, 28, 512, 512, net.launch)
        assert gq.tile_queue == 1 and gq.cus_reserved == (res or 0) and ops._geom(spec, 1, 2, 28, 28, 512, 512, other.launch).tile_queue == 0
        red._share_gpu(False)
        assert net.launch.tile_queue is None and net.launch.cus_reserved is None


def test_gpu_suite_order_puts_parity_before_selfchecks_before_control_flow():
    """conftest's collection order (the driver runs `pytest -x`): oracle / golden parity files first, `selfcheck` tests next, bench /
    launcher / DDP control flow last."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", "gpu"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    ids = [ln for ln in r.stdout.splitlines() if "::" in ln]
    assert len(ids) > 400, r.stdout[-2000:]
    files = [i.split("::")[0].split("/")[-1] for i in ids]
    first_control = min(k for k, f in enumerate(files) if f in ("test_bench_gpu.py", "test_bench_launch.py", "test_ddp_gpu.py"))
    assert all(f in ("test_bench_gpu.py", "test_bench_launch.py", "test_ddp_gpu.py") for f in files[first_control:])
    assert files[0] == "test_ops_gpu.py"
    # the known self-comparison tests sit between the parity block and the control-flow block
    k_side = next(k for k, i in enumerate(ids) if "side_stream_weight_gradients" in i)
    k_last_parity = max(k for k, i in enumerate(ids) if "test_contract_gpu.py" in i)
    assert k_last_parity < k_side < first_control


def test_bench_traffic_lookup_is_by_name_and_exact_kernel_family(tmp_path, monkeypatch):
    """bench.py quotes `roofline.traffic` from the committed PMC summaries: the one stamped with THIS build of the kernels that holds the
    EXACT kernel family (storage type included), chosen by file name -- a fresh checkout gives the files arbitrary mtimes (the fp16 bs = 128
    summary was once quoted for the bf16 headline on a GPU box)."""
    import json
    import os
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    mk = lambda sha, fams: dict({k: {"hbm_bytes_per_launch_corrected": v} for k, v in fams.items()}, _build={"csrc_sha16": sha, "git_head": "abc"})
    (prof / "r09a_pmc_hbm_traffic.json").write_text(json.dumps(mk("OLD", {"conv_igemm_halo_kernel<bf16>": 1.0})))
    (prof / "r09b_pmc_hbm_traffic.json").write_text(json.dumps(mk("NOW", {"conv_igemm_halo_kernel<bf16>": 474e6, "conv_wgrad_ws2_kernel": 600e6})))
    time.sleep(0.05)  # ... and the fp16 summary is the NEWER file
    (prof / "r09b_cfg5_fp16_bs128_pmc_hbm_traffic.json").write_text(json.dumps(mk("NOW", {"conv_igemm_halo_kernel<f16>": 931e6})))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "NOW")
    t = bench.pmc_traffic("conv_igemm_halo_kernel<bf16>")
    assert t["bytes_per_launch"] == 474000000 and t["source"] == "r09b_pmc_hbm_traffic.json"
    assert bench.pmc_traffic("conv_igemm_halo_kernel<f16>")["source"] == "r09b_cfg5_fp16_bs128_pmc_hbm_traffic.json"
    assert bench.pmc_traffic("conv_wgrad_ws2_kernel<bf16>")["bytes_per_launch"] == 600000000  # families named without a storage type: by their stem
    assert bench.pmc_traffic("conv_igemm_halo_kernel<fp16x3>")["bytes_per_launch"] is None
    monkeypatch.setattr(bench, "csrc_sha16", lambda: "OTHER")
    t = bench.pmc_traffic("conv_igemm_halo_kernel<bf16>")
    assert t["bytes_per_launch"] is None and t["stale"] is True
