"""Race screen for the forward / data-gradient kernels: they contain no atomics, so repeated launches on the same input must agree BIT FOR BIT
(a wrong `vmcnt` allowance in a loader, or a slot refilled too early, shows up as an intermittent difference).
  python tools/determinism_check.py [repeats]   -- bs = 64 and bs = 37 (ragged tile counts), bf16 and fp16, 224 and 256 pixel tiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import init_weights_he

D = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for prec in ("bf16", "fp16"):
    model = ResNet38dSeg(3, prec); init_weights_he(model, 42); model = model.to(D); model.eval()
    for n, s in ((64, 224), (37, 224), (24, 256)):
        x = torch.randn(n, 3, s, s, generator=torch.Generator().manual_seed(n)).to(D)
        with torch.no_grad():
            ref = model(x).clone()
            bad = sum(0 if torch.equal(model(x), ref) else 1 for _ in range(reps))
        print(f"[{prec}] forward n={n} s={s}: {bad} of {reps} repeats differ", flush=True)
        assert bad == 0
# data gradients of the three kernel families, repeated
dt = torch.bfloat16
for (cin, cout, k, s_, d, H) in ((512, 512, 3, 1, 1, 28), (1024, 2048, 3, 1, 4, 28), (2048, 4096, 1, 1, 1, 28), (256, 512, 3, 2, 1, 56)):
    spec = ops.ConvSpec(cin, cout, k, s_, d)
    ho, wo = spec.out_hw(H, H)
    g = torch.Generator().manual_seed(cin + cout)
    gy = torch.randn(64, ho, wo, cout, generator=g).to(D, dt)
    wd = (torch.randn(cin, k, k, cout, generator=g) * 0.02).to(D, dt)
    outs = []
    for _ in range(reps):
        gx = torch.empty(64, H, H, cin, device=D, dtype=dt)
        ops.conv2d_dgrad(spec, gy, wd, (H, H), out_raw=gx)
        outs.append(gx)
    bad = sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:])
    print(f"dgrad {cin}->{cout} k{k} s{s_} d{d}: {bad} of {reps - 1} repeats differ", flush=True)
    assert bad == 0
print("determinism ok")
