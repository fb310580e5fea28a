"""Packed on-disk intermediates between the pipeline's stages.

The reference hands the stage-2 logits and the stage-0 CAMs to stage 3 as one tiny file per tile --
`logits_32x32/<name>.pt` (a pickled *CUDA* tensor [C,32,32] f32, infer_pseudo_masks.py:127) and `<name>.npy` (float64
[C,32,32], OEEM/classification/prepare_seg_inputs.py:136-138) -- and `RefineDataset.__getitem__` re-opens both for every sample
of every epoch (revise_pseudo_labels.py:49-67): 2 x 10k+ file opens, an unpickle and a float64 -> float32 cast per epoch.

A pack is ONE file: a fixed-size little-endian header, a JSON index (names, shape, dtype) and the raw [n, C, 32, 32]
array, read through `numpy.memmap` (zero copies until a sample is touched; the whole array of a 10k-tile stage is 123 MB
at C = 3, so it can also simply be kept in HBM).  Values round-trip bit-exactly; `get()` returns what the reference's
loaders return (f32 tensors, CAMs cast from f64 exactly as `torch.from_numpy(cam).to(torch.float32)`).
"""
from __future__ import annotations

import json
import os
import struct
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

MAGIC = b"PSPACK01"
_HEADER = struct.Struct("<8sQQ")  # magic, json bytes, data offset


class PackedTiles:
    """Read side: `pack[name]` / `pack.get(names)` -> f32 tensors; `pack.array` is the [n, ...] memmap."""

    def __init__(self, path: str, mode: str = "r"):
        self.path = str(path)
        with open(self.path, "rb") as f:
            magic, jlen, off = _HEADER.unpack(f.read(_HEADER.size))
            if magic != MAGIC:
                raise ValueError(f"{path}: not a pistoseg pack")
            meta = json.loads(f.read(jlen).decode())
        self.names: List[str] = meta["names"]
        self.item_shape = tuple(meta["item_shape"])
        self.dtype = np.dtype(meta["dtype"])
        self.index: Dict[str, int] = {n: i for i, n in enumerate(self.names)}
        self.array = np.memmap(self.path, dtype=self.dtype, mode=mode, offset=off, shape=(len(self.names),) + self.item_shape)

    def __len__(self):
        return len(self.names)

    def __contains__(self, name: str):
        return name in self.index

    def __getitem__(self, name: str) -> torch.Tensor:
        return torch.from_numpy(np.array(self.array[self.index[name]])).to(torch.float32)

    def get(self, names: Sequence[str]) -> torch.Tensor:
        idx = np.fromiter((self.index[n] for n in names), dtype=np.int64, count=len(names))
        return torch.from_numpy(np.ascontiguousarray(self.array[idx])).to(torch.float32)

    def to_device(self, device) -> torch.Tensor:
        """The whole pack as one device tensor (f32): stage 3 then indexes it with the batch's tile indices."""
        return torch.from_numpy(np.ascontiguousarray(self.array)).to(torch.float32).to(device)


class PackedTilesWriter:
    """Write side: preallocates the file; rows are written in any order.

    create=True  (default) one writer owns the file: it is created / truncated.
    create=False           the file must exist with exactly this index (names, shape, dtype) -- checked.
    shared=True            several ranks construct a writer over the SAME path concurrently and write disjoint rows (their
                           `shard_range`): whoever gets there first publishes a fully initialised file atomically (`os.link` of a
                           private temp file: it either appears complete or not at all), everybody else opens and checks it.  Nobody
                           truncates, so rows already written by a faster rank survive (a stale file with a different index raises)."""

    def __init__(self, path: str, names: Sequence[str], item_shape: Sequence[int], dtype="float32", create: bool = True, shared: bool = False):
        self.path = str(path)
        self.names = list(names)
        if len(set(self.names)) != len(self.names):
            raise ValueError("duplicate tile names")
        self.item_shape = tuple(int(v) for v in item_shape)
        self.dtype = np.dtype(dtype)
        meta = json.dumps({"names": self.names, "item_shape": self.item_shape, "dtype": self.dtype.str}).encode()
        off = (_HEADER.size + len(meta) + 4095) // 4096 * 4096  # page-aligned data
        nbytes = int(np.prod((len(self.names),) + self.item_shape)) * self.dtype.itemsize

        def init_file(fname):
            with open(fname, "wb") as f:
                f.write(_HEADER.pack(MAGIC, len(meta), off))
                f.write(meta)
                f.truncate(off + nbytes)

        if shared:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
            if not os.path.exists(self.path):
                tmp = f"{self.path}.{os.getpid()}.{id(self):x}.tmp"
                init_file(tmp)
                try:
                    os.link(tmp, self.path)  # atomic publish; FileExistsError = another rank won
                except FileExistsError:
                    pass
                finally:
                    os.unlink(tmp)
            self._check_existing(meta, off)
        elif create:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
            init_file(self.path)
        else:
            self._check_existing(meta, off)
        self.array = np.memmap(self.path, dtype=self.dtype, mode="r+", offset=off, shape=(len(self.names),) + self.item_shape)
        self.index = {n: i for i, n in enumerate(self.names)}

    def _check_existing(self, meta: bytes, off: int) -> None:
        with open(self.path, "rb") as f:
            magic, jlen, off2 = _HEADER.unpack(f.read(_HEADER.size))
            if magic != MAGIC or jlen != len(meta) or off2 != off or f.read(jlen) != meta:
                raise ValueError(f"{self.path}: existing pack has a different index (names / shape / dtype) than this writer's")

    def write(self, name: str, value) -> None:
        self.write_rows(self.index[name], torch.as_tensor(value).unsqueeze(0))

    def write_rows(self, start: int, values) -> None:
        """rows [start, start + k) <- values [k, ...] (tensor on any device, or array)."""
        v = values.detach().cpu().numpy() if isinstance(values, torch.Tensor) else np.asarray(values)
        if tuple(v.shape[1:]) != self.item_shape:
            raise ValueError(f"rows of shape {v.shape[1:]} do not match the pack's {self.item_shape}")
        self.array[start:start + v.shape[0]] = v.astype(self.dtype, copy=False)

    def close(self) -> None:
        self.array.flush()


def pack_logits_dir(logits_dir: str, out_path: str, names: Optional[Iterable[str]] = None) -> PackedTiles:
    """Convert a reference `logits_32x32/` directory of `<name>.pt` files (torch.save of a [C,32,32] tensor, possibly a CUDA
    tensor: loaded with map_location='cpu' as RefineDataset does) into one pack."""
    d = Path(logits_dir)
    names = sorted(p.stem for p in d.glob("*.pt")) if names is None else list(names)
    first = torch.load(d / (names[0] + ".pt"), map_location="cpu")
    w = PackedTilesWriter(out_path, names, first.shape, "float32")
    for i, n in enumerate(names):
        w.write_rows(i, torch.load(d / (n + ".pt"), map_location="cpu").to(torch.float32).unsqueeze(0))
    w.close()
    return PackedTiles(out_path)


def pack_cam_dir(cam_dir: str, out_path: str, names: Optional[Iterable[str]] = None, dtype="float64") -> PackedTiles:
    """Convert a stage-0 CAM directory of `<name>.npy` files (float64 [C,32,32]) into one pack; float64 is kept by default so
    that `get()`'s cast reproduces `torch.from_numpy(cam).to(torch.float32)` bit for bit."""
    d = Path(cam_dir)
    names = sorted(p.stem for p in d.glob("*.npy")) if names is None else list(names)
    first = np.load(d / (names[0] + ".npy"))
    w = PackedTilesWriter(out_path, names, first.shape, dtype)
    for i, n in enumerate(names):
        w.write_rows(i, np.load(d / (n + ".npy"))[None])
    w.close()
    return PackedTiles(out_path)
