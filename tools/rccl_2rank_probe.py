"""Can RCCL run two ranks of one job on ONE GPU (the only kind of box the test tier has)?  Each rank tries an all-reduce
with backend 'nccl' on cuda:0 and prints what happened; run under a timeout."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((1 << 20,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: nccl all_reduce on a shared GPU ok, value {float(t[0])}", flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    mp.start_processes(run, args=(2, int(sys.argv[1]) if len(sys.argv) > 1 else 29533), nprocs=2, join=True, start_method="spawn")
