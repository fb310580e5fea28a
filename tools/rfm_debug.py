import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from oracle.make_golden import make_inputs, with_bg
from pistoseg_amd.revise_net import Net
from pistoseg_amd.rfm_loss import rfm_losses
from pistoseg_amd.trainer import RFMTrainer
D = torch.device("cuda:0")
n, s, c = 2, 64, 4
sd = ref_cpu.make_state_dict(c, True, seed=42)
x, pmask, pcam, lab = make_inputs(n, s, c, seed=110)
pm, pc, label = with_bg(pmask, pcam, lab)
xd, pmd, pcd, lbd = x.to(D), pm.to(D), pc.to(D), label.reshape(n, c).to(D)
m1 = Net(c, "fp32"); m1.load_state_dict(sd); m1 = m1.to(D); m1.train()
drop = m1.sample_dropout(n, D)
m1.sample_dropout = lambda a, b: drop
m2 = Net(c, "fp32"); m2.load_state_dict(sd); m2 = m2.to(D); m2.train(); m2.sample_dropout = lambda a, b: drop
tr = RFMTrainer(m1, lr=0.0, wt_dec=0.0, max_step=10)   # lr 0: weights unchanged, gradients stay in the arena
l1 = tr.train_step(xd, pmd, pcd, lbd)
outs = m2(xd, pmd, pcd)
(l2, *_), grads = rfm_losses([o.detach() for o in outs], pmd, pcd, lbd, want_grad=True)
torch.autograd.backward(outs, grads)
print("loss", float(l1[0]), float(l2))
named = dict(m2.named_parameters())
for name, (o, nn) in tr.offsets.items():
    g1 = tr.g_flat[o:o+nn]
    p = named[name]
    g2 = p.grad.permute(0, 2, 3, 1).reshape(-1)
    print(f"{name:30s} {float((g1-g2).abs().max()/g2.abs().max().clamp_min(1e-20)):.3e}")
