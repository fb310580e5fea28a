import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
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
rec = []
orig = ops.conv2d_dgrad
def spy(spec, dy, w, hw, **kw):
    orig(spec, dy, w, hw, **kw)
    rec.append((spec, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}, dy.clone()))
ops.conv2d_dgrad = spy
import pistoseg_amd.resnet38d as R
logits = model(x.to(D))
loss, dlogits = ops.softmax_ce(logits.detach(), target.to(D), 3, want_grad=True)
logits.backward(dlogits)
# oracle: gradient wrt x6 (input of b7) and wrt a (activated)
with torch.no_grad():
    xx = F.conv2d(x, sd["conv1a.weight"], padding=1)
    for name, kind, cin, cmid, cout, stride, fdil, dil, _p in ref_cpu.BLOCKS[:-1]:
        if kind == "res": xx, a = ref_cpu._res_unit(sd, name, cin, cmid, cout, stride, fdil, dil, xx)
        else: xx, a = ref_cpu._bot_unit(sd, name, cin, cout, stride, dil, xx, drop.get(f"{name}.dropout_2b1"), drop.get(f"{name}.dropout_2b2"))
x6 = xx.clone().requires_grad_(True)
a7 = ref_cpu._bn_relu(sd, "b7.bn_branch2a", x6); a7.retain_grad()
sc = F.conv2d(a7, sd["b7.conv_branch1.weight"]); sc.retain_grad()
h = F.conv2d(a7, sd["b7.conv_branch2a.weight"]); h.retain_grad()
h1 = ref_cpu._drop(ref_cpu._bn_relu(sd, "b7.bn_branch2b1", h), drop["b7.dropout_2b1"])
h2 = F.conv2d(h1, sd["b7.conv_branch2b1.weight"], padding=4, dilation=4)
h3 = ref_cpu._drop(ref_cpu._bn_relu(sd, "b7.bn_branch2b2", h2), drop["b7.dropout_2b2"])
x7 = sc + F.conv2d(h3, sd["b7.conv_branch2b2.weight"]); x7.retain_grad()
conv6 = ref_cpu._bn_relu(sd, "bn7", x7)
cam = F.conv2d(ref_cpu._drop(conv6, drop["dropout7"]), sd["fc8.weight"])
lg = ref_cpu.bilinear(cam, (s, s), True)
l = ref_cpu.seg_ce_loss(lg, target, 3); l.backward()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def rel(a, b):
    a, b = a.double(), b.double(); return float((a-b).abs().max()/b.abs().max()), float((a-b).norm()/b.norm())
# recorded dgrads in order: b7: 2b2(g3), 2b1(g2), branch1 (t), 2a (Gp)
names = ["b7.2b2->g3", "b7.2b1->g2", "b7.b1->t", "b7.2a->Gp"]
for nm, (spec, kw, dy) in zip(names, rec[:4]):
    print(nm, spec, {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in kw.items()})
print("G(x7) vs oracle", rel(rec[0][2].cpu(), nhwc(x7.grad)))
print("g2 (dy of 2a dgrad) vs oracle h.grad", rel(rec[3][2].cpu(), nhwc(h.grad)))
t = rec[2][1]["out_raw"].cpu()
t_ref = F.conv_transpose2d(x7.grad, sd["b7.conv_branch1.weight"])
print("t vs oracle", rel(t, nhwc(t_ref)))
print("Gp vs oracle x6.grad", rel(rec[3][1]["out"].cpu(), nhwc(x6.grad)))
print("a7 (mask src) vs oracle", rel(rec[3][1]["mask_src"].cpu(), nhwc(a7.detach())))
da = nhwc(a7.grad)
pre = t + nhwc(F.conv_transpose2d(h.grad, sd["b7.conv_branch2a.weight"]))
print("t + dgrad2a(cpu) vs a7.grad", rel(pre, da))
# recompute epilogue on cpu from GPU pieces
scale = (sd["b7.bn_branch2a.weight"] / torch.sqrt(sd["b7.bn_branch2a.running_var"] + 1e-5))
gp_cpu = torch.where(nhwc(a7.detach()) > 0, da * scale.view(1,1,1,-1), torch.zeros(()))
print("cpu-composed Gp vs x6.grad", rel(gp_cpu, nhwc(x6.grad)))
