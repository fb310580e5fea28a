"""Peak device memory of the bs=64 bf16 training step with and without the side stream for the weight gradients."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device("cuda:0")
for overlap in (False, True):
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=1); model = model.to(D)
    tr = SegTrainer(model, overlap_wgrad=overlap)
    x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
    for _ in range(4): tr.train_step(x, y)
    torch.cuda.synchronize()
    print(f"overlap={overlap}: peak allocated {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.max_memory_reserved()/2**30:.1f} GiB", flush=True)
    del tr, model
