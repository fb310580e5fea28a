"""bf16 stage-3 step, RFM heads in f32 vs in bf16: per-tensor gradient error against the CPU oracle (n = 8 tiles of 224 x 224, C = 4)."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import ref_cpu
from oracle.make_golden import make_inputs
from pistoseg_amd.revise_net import Net
from pistoseg_amd.trainer import RFMTrainer
D = torch.device('cuda:0')
n, s, c, chunk = 8, 224, 4, 4
sd = ref_cpu.make_state_dict(c, True, seed=42)
x, pmask, pcam, lab = make_inputs(n, s, c, seed=180)
pm = torch.cat([torch.zeros(n, 1, 32, 32), pmask], 1); pc = torch.cat([torch.zeros(n, 1, 32, 32), pcam], 1)
label = torch.cat([torch.ones(n, 1), lab], 1).view(n, c, 1, 1)
g = torch.Generator().manual_seed(181)
drop = {name: (torch.rand(n, ch, generator=g) >= p).float() / (1 - p) for name, ch, p in (("b6.dropout_2b1", 512, 0.3), ("b6.dropout_2b2", 1024, 0.3), ("b7.dropout_2b1", 1024, 0.5), ("b7.dropout_2b2", 2048, 0.5), ("dropout7", 4096, 0.5))}
sd_ref = {k: v.clone() for k, v in sd.items()}
tk = ref_cpu.trainable_keys(sd_ref)
for k in tk: sd_ref[k].requires_grad_(True)
for lo in range(0, n, chunk):
    sl = slice(lo, lo + chunk)
    outs = ref_cpu.revise_forward(sd_ref, x[sl], pm[sl], pc[sl], {k: v[sl] for k, v in drop.items()})
    (ref_cpu.rfm_losses(outs, pm[sl], pc[sl], label[sl], (s, s))[0] * (chunk / n)).backward()
for prec, h32, ls in (("fp16", False, 1024.0), ("fp16", False, 65536.0), ("fp16", False, 2.0 ** 20), ("fp16x3", True, 1024.0), ("fp16x3", True, 65536.0)):
    model = Net(num_classes=c, precision=prec); model.load_state_dict(sd, strict=True); model = model.to(D); model.train()
    model.heads_f32 = h32
    model.sample_dropout = lambda n_, dev_: {k: v.to(dev_) for k, v in drop.items()}
    tr = RFMTrainer(model, lr=0.0, wt_dec=0.0, max_step=10, loss_scale=ls)
    tr.train_step(x.to(D), pm.to(D), pc.to(D), label.reshape(n, c).to(D)); torch.cuda.synchronize()
    errs = {}
    for k in tk:
        o, cnt = tr.offsets[k]; co, ci, kh, kw = sd[k].shape
        a = tr.g_flat[o:o + cnt].view(co, kh, kw, ci).permute(0, 3, 1, 2).cpu().double() / tr.loss_scale
        b = sd_ref[k].grad.double()
        errs[k] = float((a - b).norm() / b.norm())
    worst = sorted(errs.items(), key=lambda t: -t[1])[:4]
    print(f"{prec} heads_f32={h32} loss_scale={ls} skipped={tr.skipped_steps}: " + ", ".join(f"{k} {v:.3e}" for k, v in worst) + f"; median {sorted(errs.values())[len(errs)//2]:.3e}", flush=True)
    del tr, model; torch.cuda.empty_cache()
