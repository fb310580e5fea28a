import sys, torch, traceback, collections
sys.path.insert(0, '.')
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device('cuda:0')
model = ResNet38dSeg(3, 'bf16'); init_weights_he(model, seed=1); model = model.to(D)
tr = SegTrainer(model)
x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('copy_', 'clone', '_to_copy', 'contiguous', 'cat', 'fill_', 'zero_', 'zeros', 'full')):
            st = [f for f in traceback.extract_stack() if 'pistoseg_amd' in f.filename or 'bench' in f.filename]
            where = ' <- '.join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in st[-3:])
            shape = tuple(args[0].shape) if args and hasattr(args[0], 'shape') else ()
            sites[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))
with M():
    tr.train_step(x, y)
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]): print(v, k)
print('--- inference')
sites.clear()
model.eval()
with M(), torch.no_grad():
    model(x)
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]): print(v, k)
