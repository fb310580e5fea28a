import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from oracle.make_golden import make_inputs
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
D = torch.device("cuda:0")
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
c, n, s = 3, 2, 64
sd = ref_cpu.make_state_dict(c, False, seed=42)
model = ResNet38dSeg(classes=c, precision=precision); model.load_state_dict(sd); model = model.to(D); model.train()
g = torch.Generator().manual_seed(77)
x, *_ = make_inputs(n, s, 4, 106)
target = torch.randint(0, 4, (n, s, s), generator=g)
drop = {k: v.cpu() for k, v in model.sample_dropout(n, D).items()}
model.sample_dropout = lambda n_, dev_: {k: v.to(dev_) for k, v in drop.items()}
logits = model(x.to(D))
loss, dlogits = ops.softmax_ce(logits.detach(), target.to(D), 3, want_grad=True)
logits.backward(dlogits)
sd_ref = {k: v.clone() for k, v in sd.items()}
tk = ref_cpu.trainable_keys(sd_ref)
for k in tk: sd_ref[k].requires_grad_(True)
ref_logits = ref_cpu.seg_forward(sd_ref, x, drop)
ref_loss = ref_cpu.seg_ce_loss(ref_logits, target, 3); ref_loss.backward()
print("loss", float(loss), float(ref_loss))
named = dict(model.named_parameters())
for k in tk:
    a, b = named[k].grad.cpu().double(), sd_ref[k].grad.double()
    print(f"{k:32s} maxrel {float((a-b).abs().max()/b.abs().max()):.3e}  l2rel {float((a-b).norm()/b.norm()):.3e}")
