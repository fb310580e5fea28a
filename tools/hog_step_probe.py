"""Whole training steps beside a kernel that holds `hog` CUs (an RCCL all-reduce beside the backward): static persistent schedule
(tiles_per_block 0) vs 1-2 tiles per block vs a reserved-CU grid vs the in-kernel ticket queue, with the weight gradients on the side
stream (two kernels share the GPU anyway)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he

lib = _lib.use_debug_library()  # the ps_debug_* switches live in libpistoseg_hip_debug.so only
D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=1); model = model.to(D)
tr = SegTrainer(model)
x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
hog_stream = torch.cuda.Stream()
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize()
def run(label):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        if hog:
            with torch.cuda.stream(hog_stream):
                _lib.check(lib.ps_debug_hog(hog, 400000, 96 * 1024, hog_stream.cuda_stream), "hog")  # 0.4 s, 96 KiB LDS each: those CUs are lost
            torch.cuda._sleep(2000000)
        t0 = time.perf_counter()
        for _ in range(8): tr.train_step(x, y)
        torch.cuda.current_stream().synchronize()
        tr.wgrad_stream.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8)
        torch.cuda.synchronize()
    print(f"hog={hog:2d} CUs {label}: {best*1e3:6.2f} ms/step  {64/best:6.0f} tiles/s", flush=True)


rows = [("static schedule          ", dict()),
        ("tiles_per_block=1        ", dict(TILES_PER_BLOCK=1)),   # small batches of tiles that the dispatcher re-balances
        ("cus_reserved=32          ", dict(CUS_RESERVED=32)),     # grid and static schedule sized for #CUs - reserved (ps_conv_geom.cus_reserved)
        ("tile_queue=1             ", dict(TILE_QUEUE=1)),        # every tile / work item drawn from per-XCD ticket counters (ps_conv_geom.tile_queue)
        ("cus_reserved=32 + queue  ", dict(CUS_RESERVED=32, TILE_QUEUE=1))]
for hog in (0, 16, 32, 48):
    for label, opts in rows:
        for k, v in opts.items(): setattr(ops, k, v)
        run(label)
        for k in opts: setattr(ops, k, 0)
