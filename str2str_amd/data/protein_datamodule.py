"""Minimal ``ProteinDataModule`` for prediction (reference: src/data/protein_datamodule.py:9-57, 60-241):
batch-of-one pad-collated feature dicts in dataset order.  No Lightning dependency."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch


class BatchTensorConverter:
    def __init__(self, target_keys: Optional[List] = None):
        self.target_keys = target_keys

    def __call__(self, raw: Sequence[Dict[str, object]]):
        keys = self.target_keys if self.target_keys is not None else [k for k, v in raw[0].items() if torch.is_tensor(v)]
        out = {k: self.collate_dense_tensors([d[k] for d in raw], pad_v=0.0) for k in keys}
        out.update({k: [d[k] for d in raw] for k in raw[0] if k not in keys})
        return out

    @staticmethod
    def collate_dense_tensors(samples: Sequence[torch.Tensor], pad_v: float = 0.0) -> torch.Tensor:
        if len(samples) == 0:
            return torch.Tensor()
        if len({x.dim() for x in samples}) != 1:
            raise RuntimeError(f"Samples has varying dimensions: {[x.dim() for x in samples]}")
        shape = [max(s) for s in zip(*[x.shape for x in samples])]
        res = torch.full((len(samples), *shape), pad_v, dtype=samples[0].dtype, device=samples[0].device)
        for i, t in enumerate(samples):
            res[i][tuple(slice(0, k) for k in t.shape)] = t
        return res


class ProteinDataModule:
    def __init__(self, dataset, batch_size: int = 1, generator_seed: int = 42, train_val_split=(0.95, 0.05),
                 num_workers: int = 0, pin_memory: bool = False, shuffle: bool = False, **_):
        self.dataset, self.batch_size = dataset, batch_size
        self._collate = BatchTensorConverter()

    def setup(self, stage: Optional[str] = None) -> None:
        pass

    def test_dataloader(self):
        n = len(self.dataset)
        return [self._collate([self.dataset[j] for j in range(i, min(n, i + self.batch_size))])
                for i in range(0, n, self.batch_size)]

    predict_dataloader = test_dataloader
