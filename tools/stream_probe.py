import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import init_weights_he
D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model); model = model.to(D); model.eval()
x = torch.randn(64, 3, 224, 224, device=D)
def run_single():
    with torch.no_grad(): model(x)
def run_split(k, streams):
    cur = torch.cuda.current_stream()
    chunks = x.chunk(k)
    with torch.no_grad():
        for s, c in zip(streams, chunks):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                model(c)
        for s in streams: cur.wait_stream(s)
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("single stream: %.2f ms" % timeit(run_single))
for k in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(k)]
    print("%d streams (batch split): %.2f ms" % (k, timeit(lambda: run_split(k, streams))))
# same split, one stream (to separate the effect of smaller launches)
def run_split_serial(k):
    with torch.no_grad():
        for c in x.chunk(k): model(c)
print("2 chunks, one stream: %.2f ms" % timeit(lambda: run_split_serial(2)))
