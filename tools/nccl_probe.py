import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from pistoseg_amd.dist import BucketedAllReduce, plan_buckets
flat = torch.arange(1000, device="cuda", dtype=torch.float32)
red = BucketedAllReduce(flat, plan_buckets([("fc8.weight", 10), ("b7.a.weight", 500), ("b6.a.weight", 490)], 100), dist.group.WORLD)
red.begin_step(); red.on_unit_done("fc8"); red.on_unit_done("b7"); red.finish()
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", float(flat.sum()), float(t))
dist.destroy_process_group()
