"""Soak of the tile_queue launch option: STEPS deterministic two-stream training steps (bs 64, 224 x 224, bf16) under the static schedule and
under `tile_queue = 1` [+ `cus_reserved = 32`] -- tens of thousands of queue launches through both streams' counter rings -- must end on
bit-identical weights:  python tools/queue_soak.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he

_lib.load()
D = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(7)
x = torch.randn(64, 3, 224, 224, generator=g).to(D)
y = torch.randint(0, 4, (64, 224, 224), generator=g).to(D)


def run(queue, reserved):
    torch.manual_seed(0)
    model = ResNet38dSeg(3, "bf16")
    init_weights_he(model, seed=42)
    model = model.to(D)
    tr = SegTrainer(model, lr=2e-5, track_iou=False, deterministic=True)
    model.launch.tile_queue, model.launch.cus_reserved = (1 if queue else None), (reserved or None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [tr.train_step(x, y) for _ in range(STEPS)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return [float(l) for l in losses], tr.p_flat.clone(), dt


for res in (0, 32):  # (the reservation changes the weight gradient's split-K plan, i.e. its f32 summation order: compare like with like)
    ref_l, ref_p, ref_t = run(False, res)
    print(f"static schedule reserved={res:2d}: {STEPS} steps in {ref_t:.2f} s, loss {ref_l[0]:.6f} -> {ref_l[-1]:.6f}", flush=True)
    l, p, t = run(True, res)
    same = l == ref_l and bool(torch.equal(p, ref_p))
    print(f"tile_queue=1    reserved={res:2d}: {STEPS} steps in {t:.2f} s, loss {l[0]:.6f} -> {l[-1]:.6f}, losses and weights bit-identical to the static run: {same}", flush=True)
    assert same
