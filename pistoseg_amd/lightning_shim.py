"""`pl.LightningModule` stand-in used when pytorch_lightning is not installed (it is not in this image).

Only what the reference's module shells and entry scripts touch is provided: `self.log`, `save_hyperparameters`
(captures the constructor's `args` Namespace under `hparams['args']`), `load_from_checkpoint(path, **overrides)`
reading the Lightning `.ckpt` dict layout (`state_dict` with `model.*` keys, `hyper_parameters={'args': Namespace}`;
segmentation_test.py:95-105, infer_pseudo_masks.py:95), `save_checkpoint`, `.cuda(i)`, `.eval()`.
If pytorch_lightning IS importable the shells subclass the real thing instead.
"""
from __future__ import annotations

import inspect
import os
from typing import Any, Dict

import torch

try:  # pragma: no cover - not installed in this image
    import pytorch_lightning as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:
    HAVE_LIGHTNING = False

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams: Dict[str, Any] = {}
            self.logged: Dict[str, Any] = {}
            self.current_epoch = 0

        def log(self, name, value, **_):
            self.logged[name] = value

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            args = frame.f_locals.get("args")
            self.hparams = {"args": args}

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
            ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            module = cls(**hp)
            module.load_state_dict(ckpt["state_dict"], strict=True)
            return module

        def save_checkpoint(self, dirpath: str, epoch: int, metric: float = 0.0, metric_name: str = "validation_miou_mask_epoch") -> str:
            """Writes `<dirpath>/epoch=EE-<metric_name>=M.MMMM.ckpt` -- consumers find it by the substring 'epoch='
            (infer_pseudo_masks.py:166-173, segmentation_test.py:268-275)."""
            os.makedirs(dirpath, exist_ok=True)
            path = os.path.join(dirpath, f"epoch={epoch:02d}-{metric_name}={metric:.4f}.ckpt")
            torch.save({"epoch": epoch, "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                        "hyper_parameters": dict(self.hparams), "pytorch-lightning_version": "shim"}, path)
            return path
