"""Same-box A/B of debug-library switches on the whole training step:  python tools/train_ab.py name=value[,name=value] ...
Each argument is one variant (comma-separated ps_debug_set_<name>(value) calls; the word `base` = no switch); the variants are run
interleaved, 3 rounds x 10 steps each, and the best round per variant is printed.  bs = 64, 224 x 224, bf16, two-stream backward.
--infer as the first argument: the no-grad forward (eval mode) instead of the training step; --serial: weight gradients on the launch stream;
--rfm: the stage-3 step (RFMTrainer, bs = 32, C = 4: BASELINE configs[3]) instead of the segmentation step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he

lib = _lib.use_debug_library()
D = torch.device("cuda:0")
argv = sys.argv[1:]
INFER, SERIAL, RFM = "--infer" in argv, "--serial" in argv, "--rfm" in argv
argv = [a for a in argv if a not in ("--infer", "--serial", "--rfm")]
variants = argv or ["base"]
if RFM:
    from pistoseg_amd.revise_net import Net
    from pistoseg_amd.trainer import RFMTrainer
    B, c = 32, 4
    model = Net(c, precision="bf16"); init_weights_he(model, seed=42); model = model.to(D)
    tr = RFMTrainer(model, lr=0.01, wt_dec=5e-4, max_step=10 ** 6)
    g = torch.Generator(device="cpu").manual_seed(4321)
    x = torch.randn(B, 3, 224, 224, generator=g).to(D)
    pmask = torch.cat([torch.zeros(B, 1, 32, 32), torch.randn(B, c - 1, 32, 32, generator=g)], 1).to(D)
    pcam = torch.cat([torch.zeros(B, 1, 32, 32), torch.randn(B, c - 1, 32, 32, generator=g)], 1).to(D)
    lab = (torch.rand(B, c - 1, generator=g) < 0.5).float()
    lab[torch.arange(B), torch.randint(0, c - 1, (B,), generator=g)] = 1.0
    label = torch.cat([torch.ones(B, 1), lab], 1).to(D)
    def one(): tr.train_step(x, pmask, pcam, label)
else:
    B = 64
    model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=42); model = model.to(D)
    tr = SegTrainer(model)
    x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
    if INFER:
        model.eval()
        def one():
            with torch.no_grad(): model(x)
    else:
        def one(): tr.train_step(x, y)
if SERIAL:
    tr.wgrad_stream = None
from pistoseg_amd import ops
_geom = ops._geom
def _geom_unshared(*a):  # A/B of the gpu_shared launch option: the pseudo-switch `gpu_shared=0` hides the two-stream backward's hint from the library
    g = _geom(*a); g.gpu_shared = 0; return g
def apply(v):
    lib.ps_debug_reset()  # every tunable back to the library default (one list, in the library)
    ops._geom = _geom
    if hasattr(model, "heads_f32"): model.heads_f32 = False
    ops.TILE_QUEUE = 0
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=")
            if k == "gpu_shared":
                ops._geom = _geom if int(val) else _geom_unshared
            elif k == "tile_queue":  # the launch option (ps_conv_geom.tile_queue), through the module default
                ops.TILE_QUEUE = int(val)
            elif k == "heads_f32":  # --rfm: the RFM heads in f32 (1) or in the backbone's 16-bit type (0)
                model.heads_f32 = bool(int(val))
            else:
                getattr(lib, "ps_debug_set_" + k)(int(val))
for _ in range(5): one()
best = {v: 1e9 for v in variants}
for r in range(3):
    for v in variants:
        apply(v)
        one(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): one()
        torch.cuda.synchronize()
        best[v] = min(best[v], (time.perf_counter() - t0) / 10)
apply("base")
for v in variants: print(f"{v:32s} {1e3 * best[v]:7.3f} ms/step  {B / best[v]:7.1f} tiles/s")
