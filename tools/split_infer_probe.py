"""Probe: one batch-64 forward vs the same batch split over 2 / 4 HIP streams (do the other streams' blocks fill the last partial
rounds of the persistent conv launches?)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import init_weights_he
D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=1); model = model.to(D); model.eval()  # (train()/eval() return None, a quirk kept from the reference)
x = torch.randn(64, 3, 224, 224, device=D)
streams = [torch.cuda.Stream() for _ in range(4)]
def run(parts):
    if parts == 1:
        with torch.no_grad(): return model(x)
    cur = torch.cuda.current_stream()
    outs = []
    per = 64 // parts
    for i in range(parts):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s), torch.no_grad():
            outs.append(model(x[i * per:(i + 1) * per]))
    for i in range(parts): cur.wait_stream(streams[i])
    return torch.cat(outs)
for parts in (1, 2, 4, 1, 2):
    for _ in range(3): run(parts)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): run(parts)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"parts={parts}: {dt*1e3:.2f} ms  {64/dt:.0f} tiles/s", flush=True)
