"""Per-launch table of one training step (HIP events around every conv launch):  python tools/step_profile.py [batch] [bf16|fp16] [classes]
(PS_TILE_QUEUE=1 in the environment: the tile_queue launch option on every launch)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device("cuda:0")
rec = []
def wrap(name):
    orig = getattr(ops, name)
    def f(spec, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(spec, *a, **kw); e1.record()
        t = a[0]
        n = t.shape[0]
        if name == "conv2d_dgrad":
            h, w = a[2]; ho, wo = spec.out_hw(h, w)
        else:
            h, w = t.shape[1:3]; ho, wo = spec.out_hw(h, w)
        flops = 2.0 * n * ho * wo * spec.cout * spec.cin * spec.ksize ** 2
        epi = "+".join(k for k in ("add0", "out_raw", "out_act", "out", "mask_src", "add1", "drop") if kw.get(k) is not None)
        rec.append((name[7:], f"{spec.cin}->{spec.cout} k{spec.ksize} s{spec.stride} d{spec.dilation} @{h}", epi, flops, e0, e1))
        return r
    setattr(ops, name, f)
for nm in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad"): wrap(nm)
import pistoseg_amd.resnet38d, pistoseg_amd.seg_model
ops.TILE_QUEUE = int(os.environ.get("PS_TILE_QUEUE", "0"))
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
CLASSES = int(sys.argv[3]) if len(sys.argv) > 3 else 3
model = ResNet38dSeg(CLASSES, PREC); init_weights_he(model); model = model.to(D)
tr = SegTrainer(model, overlap_wgrad=False)  # one stream: every launch alone on the GPU
x = torch.randn(BATCH, 3, 224, 224, device=D); y = torch.randint(0, CLASSES + 1, (BATCH, 224, 224), device=D)
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize(); rec.clear()
tr.train_step(x, y); torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, shape, epi, fl, e0, e1 in rec:
    k = (kind, shape, epi); d = agg.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
tot = collections.defaultdict(float)
for (kind, shape, epi), (n, ms, fl) in agg.items():
    print(f"{kind:6s} {shape:34s} {epi:28s} n={n:2d} avg={ms/n*1e3:7.1f}us {fl/ms/1e9:6.0f}TF total={ms:6.2f}ms")
    tot[kind] += ms
print(dict(tot), "sum", sum(tot.values()))
