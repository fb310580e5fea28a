import sys, time, torch
sys.path.insert(0, '/root/repo')
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device('cuda:0')
prec, B, c = "fp16", 128, 4
model = ResNet38dSeg(c, prec); init_weights_he(model, seed=42); model = model.to(D)
tr = SegTrainer(model, ignore_index=None)
x = torch.randn(B, 3, 224, 224, device=D); y = torch.randint(0, c, (B, 224, 224), device=D)
evs = []
nf, ad = ops.nonfinite_count, ops.adamw_step_guarded
def nf_timed(g, out=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = nf(g, out=out); e1.record(); evs.append(("nonfinite", e0, e1)); return r
def ad_timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ad(*a, **k); e1.record(); evs.append(("adamw_guarded", e0, e1)); return r
ops.nonfinite_count, ops.adamw_step_guarded = nf_timed, ad_timed
for _ in range(3): tr.train_step(x, y)
evs.clear()
for _ in range(6): tr.train_step(x, y)
torch.cuda.synchronize()
for name in ("nonfinite", "adamw_guarded"):
    ts = [a.elapsed_time(b) for n, a, b in evs if n == name]
    print(name, [round(t * 1e3) for t in ts], "us")
