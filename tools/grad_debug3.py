import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from oracle.make_golden import make_inputs
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
D = torch.device("cuda:0")
c, n, s = 3, 2, 64
sd = ref_cpu.make_state_dict(c, False, seed=42)
model = ResNet38dSeg(classes=c, precision="fp32"); model.load_state_dict(sd); model = model.to(D); model.train()
g = torch.Generator().manual_seed(77)
x, *_ = make_inputs(n, s, 4, 106)
target = torch.randint(0, 4, (n, s, s), generator=g)
drop = {k: v.cpu() for k, v in model.sample_dropout(n, D).items()}
model.sample_dropout = lambda n_, dev_: {k: v.to(dev_) for k, v in drop.items()}
logits = model(x.to(D))
loss, dlogits = ops.softmax_ce(logits.detach(), target.to(D), 3, want_grad=True)
logits.backward(dlogits)
def oracle(dtype):
    sdr = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tk = ref_cpu.trainable_keys(sdr)
    for k in tk: sdr[k].requires_grad_(True)
    dr = {k: v.to(dtype) for k, v in drop.items()}
    lg = ref_cpu.seg_forward(sdr, x.to(dtype), dr)
    l = ref_cpu.seg_ce_loss(lg, target, 3); l.backward()
    return {k: sdr[k].grad for k in tk}
g32, g64 = oracle(torch.float32), oracle(torch.float64)
named = dict(model.named_parameters())
print(torch.__config__.show()[:600])
for k in g32:
    a, b, r = named[k].grad.cpu().double(), g32[k].double(), g64[k]
    f = lambda u, v: float((u-v).norm()/v.norm())
    print(f"{k:30s} gpu-vs-f64 {f(a,r):.2e}  cpu32-vs-f64 {f(b,r):.2e}  gpu-vs-cpu32 {f(a,b):.2e}")
