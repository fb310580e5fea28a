import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pistoseg_amd import ops, _lib
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import init_weights_he
_lib.load()
D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=42); model = model.to(D)
g = torch.Generator().manual_seed(5)
x = torch.randn(24, 3, 224, 224, generator=g).to(D)
rec = []
orig = ops.conv2d_fwd
def wrapped(spec, x_, w_, **kw):
    r = orig(spec, x_, w_, **kw)
    outs = [kw.get(k) for k in ("out_raw", "out_act") if kw.get(k) is not None]
    rec.append((f"{spec.cin}->{spec.cout} k{spec.ksize} s{spec.stride} d{spec.dilation} @{x_.shape[1]} epi={[k for k in ('add0','out_raw','out_act','drop') if kw.get(k) is not None]}", [o.clone() for o in outs]))
    return r
ops.conv2d_fwd = wrapped
import pistoseg_amd.resnet38d as R
for mode in ("eval", "train"):
    model.train(mode == "train")
    torch.manual_seed(0)
    drops = model.sample_dropout(24, D)
    fixed = {k: (torch.rand(v.shape, generator=torch.Generator().manual_seed(1)) >= 0.5).float().to(D) * 2 for k, v in drops.items()}
    model.sample_dropout = lambda n_, dev_: fixed
    runs = []
    for q in (None, 1, 1):
        model.launch.tile_queue = q
        rec.clear()
        with torch.no_grad():
            feats, _ = model.run_backbone(x, save=(mode == "train"), drop=(fixed if mode == "train" else None))
        torch.cuda.synchronize()
        runs.append(list(rec))
    for i, (name, outs) in enumerate(runs[0]):
        for r in (1, 2):
            bad = [int((a != b).sum()) for a, b in zip(outs, runs[r][i][1])]
            if any(bad):
                print(mode, "run", r, "launch", i, name, "differing elements", bad, "of", [o.numel() for o in outs], flush=True)
    print(mode, "compared", len(runs[0]), "launches")
