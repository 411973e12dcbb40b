"""A predict-only ``Trainer`` with the call surface eval.py uses from ``lightning.Trainer``
(reference src/eval.py:129,154): device placement, eval mode, no-grad, ``predict`` over a dataloader.
One process per GPU: under ``torch.distributed.run`` each rank binds to LOCAL_RANK and the model
shards replicas over ranks (DiffusionLitModule.predict_step)."""
from __future__ import annotations

import os
from typing import Any, List, Optional

import torch

from .. import ops


class Trainer:
    def __init__(self, accelerator: str = "gpu", devices: Any = 1, default_root_dir: Optional[str] = None,
                 deterministic: bool = False, logger: Any = None, **_):
        self.accelerator, self.devices, self.default_root_dir, self.logger = accelerator, devices, default_root_dir, logger

    def _device(self) -> torch.device:
        if self.accelerator in ("cpu",) or not torch.cuda.is_available():
            raise ops.HipLibraryError(
                "trainer.accelerator=cpu / no HIP device visible: the sampling path runs on MI355X only "
                "(there is deliberately no CPU fallback; use the reference itself for CPU runs)")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("S2S_DIST_BACKEND", "nccl") == "gloo":
            local %= torch.cuda.device_count()
        if local >= torch.cuda.device_count():
            raise ops.HipLibraryError(f"rank with LOCAL_RANK={local} but {torch.cuda.device_count()} GPU(s) are visible (one device per rank; "
                                      "trainer.devices must not exceed the node's GPUs)")
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def predict(self, model, dataloaders, ckpt_path: Optional[str] = None) -> List[Any]:
        import torch.distributed as dist

        dev = self._device()
        from .launch import in_distributed_job

        # (a 1-rank torch.distributed.run job initialises RCCL too: the collective path is the same code at every world size)
        if in_distributed_job() and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if os.environ.get("S2S_DIST_BACKEND", "nccl") == "gloo":   # TEST hook: several ranks sharing one GPU (RCCL wants one each)
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        if ckpt_path is not None:  # Lightning-style .ckpt: {'state_dict': {'net.…': tensor}}
            sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
            # the reference's Lightning Trainer restores strictly; here only `net.*` is part of the sampling path, so
            # non-network entries (loss / metric buffers of a training checkpoint) may be absent from this module, but
            # every network parameter must be found -- a silently unloaded key would sample from init weights
            res = model.load_state_dict(sd, strict=False)
            missing = [k for k in res.missing_keys if k.startswith("net.")]
            unexpected = [k for k in res.unexpected_keys if k.startswith("net.")]
            if missing or unexpected:
                raise RuntimeError(f"checkpoint {ckpt_path} does not match the network: missing {missing[:5]} "
                                   f"(+{max(0, len(missing) - 5)}), unexpected {unexpected[:5]} (+{max(0, len(unexpected) - 5)})")
        ops.load_library()
        model = model.to(dev).eval()
        out = []
        inf = getattr(getattr(model, "hparams", None), "inference", None)
        if inf is None and isinstance(getattr(model, "hparams", None), dict):
            inf = model.hparams.get("inference")
        mixed = bool((inf.get("mixed_batch", False) if hasattr(inf, "get") else getattr(inf, "mixed_batch", False)) if inf is not None else False)
        with torch.no_grad():
            if mixed:   # all targets in length-bucketed padded batches (BASELINE configs[4]; model.inference.mixed_batch=true)
                return [model.predict_mixed(list(dataloaders))]
            for i, batch in enumerate(dataloaders):
                batch = {k: (v.to(dev) if torch.is_tensor(v) and k != "residue_idx" else v) for k, v in batch.items()}
                out.append(model.predict_step(batch, i))
        return out
