"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference has no distributed code (its `nn.DataParallel` wrappers degenerate to one device);
tile batches are independent (BN is frozen, resnet38d.py:206-211), so
  * training = batch-dim data parallel with ONE exchange per step: a SUM all-reduce of the flat f32 gradient
    arena, cut into buckets that are contiguous in the order the reverse plan finalises them and launched on
    a side stream as soon as their last unit is done (overlap with the remaining dgrad/wgrad).  Two wire
    formats, chosen per trainer: "fp32" (the arena slices themselves: 417.8 MB per step and GPU)
    and "bf16" (each bucket cast to bf16, all-reduced, widened back into the arena: 208.9 MB -- xGMI is
    point-to-point, a ring moves 2 (p-1)/p x payload per GPU at one link's ~153 GB/s, so halving the payload
    halves the exposed tail of the last bucket; the sum then carries bf16 rounding, 2^-9 relative per addition);
  * inference = contiguous index ranges per rank, no collective on the data path, optional gather of uint8
    masks / all-reduce of the confusion matrix.
Everything here is device-agnostic so the N > 1 logic is covered by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


RCCL_DEFAULT_CHANNELS = 32  # workgroups (one CU each) an RCCL ring collective occupies on an 8-GPU xGMI node when nothing overrides it


def default_reserved_cus() -> int:
    """CUs the conv launches leave to an in-flight collective (`share="reserve"`): RCCL runs one workgroup per channel, so this is its channel
    count -- `NCCL_MAX_NCHANNELS` caps it and `NCCL_MIN_NCHANNELS` floors it when the job sets them (the usual way to trade collective bandwidth
    against compute CUs), else RCCL's default.  Echoed in bench.py's `config` so that a scaling run states what it assumed."""
    import os

    def env_int(name):
        try:
            v = int(os.environ.get(name, ""))
            return v if v > 0 else None
        except ValueError:
            return None

    n = RCCL_DEFAULT_CHANNELS
    hi, lo = env_int("NCCL_MAX_NCHANNELS"), env_int("NCCL_MIN_NCHANNELS")
    if hi is not None:
        n = min(n, hi)
    if lo is not None:
        n = max(n, lo)
    return max(1, min(n, 128))


def plan_buckets(entries: Sequence[Tuple[str, int]], limit_elems: int) -> List[Tuple[str, int, int]]:
    """entries: (parameter name 'unit.conv.weight', numel) in arena order.  Returns (closing unit, start, end)
    slices that tile the arena exactly; a bucket closes at a unit boundary once it holds >= limit_elems."""
    buckets: List[Tuple[str, int, int]] = []
    start, off, last_unit = 0, 0, None
    for name, n in entries:
        unit = name.split(".")[0]
        if last_unit is not None and unit != last_unit and off - start >= limit_elems:
            buckets.append((last_unit, start, off))
            start = off
        last_unit = unit
        off += n
    if off > start:
        buckets.append((last_unit, start, off))
    return buckets


class BucketedAllReduce:
    """SUM all-reduce of a flat gradient arena in buckets, driven by 'unit finished' notifications.

    While buckets are in flight the communication kernels hold some CUs: the conv kernels' persistent blocks (one per CU, each
    with a static share of the tiles) would then serialise the share of every block that cannot start (measured with a CU hog,
    tools/hog_probe.py: +63 % on a launch that loses 16 CUs, +0..29 % with 1-2 tiles per block), so from the first bucket launch to
    `finish()` the conv launches are told to cut their work into `shared_tiles_per_block`-tile blocks that the hardware
    dispatcher re-balances (the `tiles_per_block` launch option of ps_conv_geom); forward passes and single-GPU runs keep the fully persistent schedule."""

    def __init__(self, flat: Tensor, buckets: List[Tuple[str, int, int]], group=None, shared_tiles_per_block: int = 1, launch_opts=None,
                 payload: str = "fp32", share: str = "batch", reserved_cus: int = 16):
        """launch_opts: the `ops.LaunchOpts` of the model whose backward runs beside the buckets (what `_share_gpu` switches is state of THAT
        model's launches, not of the process).  payload: "fp32" | "bf16" (see the module docstring).
        share: how the conv launches make room for the communication kernels while buckets are in flight --
          "batch"  : `tiles_per_block = shared_tiles_per_block` -- small blocks that the hardware dispatcher re-balances (+11 % on a step beside a
                     16-CU kernel, +1.5 % alone: profiles/r03_hog_step_probe.txt);
          "reserve": `cus_reserved = reserved_cus` -- the persistent grids and their static schedules are sized for the other CUs, every block
                     is resident at once and the launch costs the ideal #CUs / (#CUs - reserved) (profiles/r04_hog_step_probe.txt).  Set
                     reserved_cus to the CUs the collective really holds (RCCL: its channel count, NCCL_MAX_NCHANNELS);
          "queue"  : `tile_queue = 1` -- the persistent kernels' blocks draw every tile / work item from per-XCD ticket counters, so a block that
                     cannot become resident takes nothing instead of delaying the launch by its static share: no knowledge of the collective's
                     size needed (+16 % beside a 16-CU kernel, like "batch", +27 % beside a 48-CU one; nothing alone);
          "reserve+queue": both -- the reservation's +8-9 % while the collective fits into it, the queue's graceful +26 % (instead of the
                     bare reservation's +51 %) when it does not (48 CUs held, 32 reserved): profiles/r04_hog_step_probe_queue.txt."""
        assert payload in ("fp32", "bf16") and share in ("batch", "reserve", "queue", "reserve+queue")
        self.share, self.reserved_cus = share, reserved_cus
        self.flat, self.buckets, self.group = flat, buckets, group
        self.comm_stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
        self.shared_tiles_per_block = shared_tiles_per_block
        self.launch_opts = launch_opts
        if launch_opts is not None and flat.is_cuda and hasattr(launch_opts, "stream_k"):
            launch_opts.stream_k = False  # this reducer changes the model's launch options by bucket timing: see ops.LaunchOpts.stream_k
        self.payload = payload
        # bf16 wire format: one staging arena the size of the gradient arena (buckets are disjoint slices of it)
        self.stage = torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16) if payload == "bf16" else None
        if payload == "bf16" and flat.is_cuda:  # ps_cast_f32_lowp / ps_convert_rows move 8 elements (16 bytes of bf16) per lane: every bucket must start and end on one
            bad = [(u, b, e) for u, b, e in buckets if b % 8 or (e - b) % 8]
            if bad or flat.data_ptr() % 32:
                raise ValueError(f"bf16 gradient payload needs buckets aligned to 8 elements (and a 32-byte aligned arena); offending (unit, start, end): {bad[:3]}")
        self._next = 0
        self._pending = []
        self._sharing = False
        self._opts_before = (None, None, None)
        self.measure = False   # record launch-stream events around finish()'s wait (comm_report)
        self._exposed = []
        # what the exchange costs, for bench.py's N > 1 line: bytes each rank puts through the collectives per step
        self.bytes_per_step = sum(e - b for _, b, e in buckets) * (2 if payload == "bf16" else 4)

    def _share_gpu(self, on: bool) -> None:
        """Switches the model's launch options for the time buckets are in flight and puts back WHAT WAS THERE BEFORE when they are done (a
        user may have set `tile_queue` / `cus_reserved` on the model for good: ops.LaunchOpts)."""
        if self.comm_stream is None or on == self._sharing or self.launch_opts is None:
            return
        lo = self.launch_opts
        if on:
            self._opts_before = (lo.tiles_per_block, lo.cus_reserved, lo.tile_queue)
            if self.share == "batch":  # (all three are per-launch arguments of the C-ABI: ps_conv_geom)
                lo.tiles_per_block = self.shared_tiles_per_block
            if "reserve" in self.share:
                lo.cus_reserved = max(self.reserved_cus, lo.cus_reserved or 0)
            if "queue" in self.share:
                lo.tile_queue = 1
        else:
            lo.tiles_per_block, lo.cus_reserved, lo.tile_queue = self._opts_before
        self._sharing = on

    def begin_step(self) -> None:
        self._next, self._pending = 0, []

    def _exchange(self, start: int, end: int):
        """Enqueue one bucket's exchange on the current stream; returns the collective's work handle."""
        if self.stage is None:
            return dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        src, st = self.flat[start:end], self.stage[start:end]
        if src.is_cuda:
            from . import ops  # device path only: the CPU (gloo) tests never load the HIP library

            ops.cast_f32_lowp(src, st)
        else:
            st.copy_(src)
        w = dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w.wait()  # RCCL: a stream dependency (the host does not block); gloo: the host waits, the tests' ranks are CPU processes
        if src.is_cuda:
            ops.convert_rows(st, src, end - start)
        else:
            src.copy_(st)
        return w

    def _launch(self, start: int, end: int) -> None:
        self._share_gpu(True)
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                w = self._exchange(start, end)
        else:
            w = self._exchange(start, end)
        self._pending.append(w)

    def on_unit_done(self, unit: str) -> None:
        while self._next < len(self.buckets) and self.buckets[self._next][0] == unit:
            _, s, e = self.buckets[self._next]
            self._launch(s, e)
            self._next += 1
        # every earlier bucket already reduced (non-blocking query)?  then the conv kernels have the GPU to themselves again
        if self._sharing and all(w.is_completed() for w in self._pending):
            self._share_gpu(False)

    def finish(self) -> None:
        """Flush buckets whose closing unit was never reported (frozen units), then wait for everything."""
        while self._next < len(self.buckets):
            _, s, e = self.buckets[self._next]
            self._launch(s, e)
            self._next += 1
        ev = None
        if self.measure and self.comm_stream is not None:  # how long the launch stream stands still for the collectives: the EXPOSED part of the exchange
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        self._pending = []
        self._share_gpu(False)

    def comm_report(self) -> dict:
        """bench.py's N > 1 line: what one step's exchange moves and how much of it the backward did not hide (mean over the steps since
        `measure` was switched on; synchronises)."""
        out = {"bytes_per_step": self.bytes_per_step, "buckets": len(self.buckets), "payload": self.payload, "share": self.share,
               "reserved_cus": self.reserved_cus if "reserve" in self.share else 0, "exposed_ms": None}
        if self._exposed:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._exposed]
            out["exposed_ms"] = round(sum(ms) / len(ms), 3)
            out["exposed_ms_max"] = round(max(ms), 3)
            out["steps_measured"] = len(ms)
        return out


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous index range of this rank: [rank*ceil(n/p), min(n, (rank+1)*ceil(n/p)))  (SURVEY.md 8e)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def gather_masks(local: Tensor, n_items: int, group=None) -> Optional[Tensor]:
    """Gather per-rank uint8 masks [n_local, H, W] to rank 0 in shard order (C2 of SURVEY 2.1)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_items + world - 1) // world
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, out, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat(out, dim=0)[:n_items]


def allreduce_confusion(cm: Tensor, group=None) -> Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cm, op=dist.ReduceOp.SUM, group=group)
    return cm
