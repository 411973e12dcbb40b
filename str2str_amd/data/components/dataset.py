"""Sampling-time dataset + featurisation (PDB file -> the feature dict ``predict_step`` reads).

Interface of the reference's ``src/data/components/dataset.py`` (ProteinFeatureTransform :26-143,
RandomAccessProteinDataset :201-287, SamplingPDBDataset :305-320).  ``predict_step`` consumes only
aatype, residue_mask, fixed_mask, residue_idx, torsion_angles_sin_cos[..., 2, :], the BACKBONE frame
rigidgroups_gt_frames[..., 0, :, :], chain_index, residue_index and accession_code, so the
featuriser computes exactly those (a ~40-line restatement of the slice of OpenFold's
atom37_to_frames / atom37_to_torsion_angles that reaches the sampler, data_transforms.py:758-894,
924-1090): group-0 frame = from_3_points(C, CA, N) o diag(-1, 1, -1) as float32, psi = dihedral frame
of (N, CA, C, O) with the reference's sign flip.  Other rigid groups / torsions are left zero.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Optional, Sequence

import numpy as np
import torch

from ...common import protein
from ...common.rigid_utils import Rigid

CA_IDX = 1


class ProteinFeatureTransform:
    def __init__(self, unit: Optional[str] = "angstrom", truncate_length: Optional[int] = None,
                 strip_missing_residues: bool = True, recenter_and_scale: bool = True, eps: float = 1e-8):
        if unit != "angstrom":
            raise ValueError(f"Invalid unit: {unit}")
        if truncate_length is not None:
            raise NotImplementedError("random truncation is a training-time augmentation")
        self.strip_missing_residues, self.recenter_and_scale, self.eps = strip_missing_residues, recenter_and_scale, eps

    def __call__(self, feats: dict) -> dict:
        feats = dict(feats)
        seq_mask = feats["atom_mask"][:, CA_IDX]
        feats.update(seq_mask=seq_mask, residue_mask=seq_mask,
                     residue_idx=feats["residue_index"] - np.min(feats["residue_index"]),
                     fixed_mask=np.zeros_like(seq_mask), sc_ca_t=np.zeros(seq_mask.shape + (3,)))
        if self.strip_missing_residues:
            known = np.where(feats["aatype"] != 20)[0]
            lo, hi = int(known.min()), int(known.max()) + 1
            feats = {k: v[lo:hi] for k, v in feats.items()}
        if self.recenter_and_scale:
            ca = feats["atom_positions"][:, CA_IDX]
            center = np.sum(ca, axis=0) / (np.sum(feats["seq_mask"]) + self.eps)
            feats["atom_positions"] = (feats["atom_positions"] - center[None, None, :]) * feats["atom_mask"][..., None]
        out = {k: torch.as_tensor(v) for k, v in feats.items()}
        out["aatype"] = out["aatype"].long()
        out["atom_positions"] = out["atom_positions"].double()
        out["atom_mask"] = out["atom_mask"].double()
        out.update(self.backbone_geometry(out["atom_positions"], out["atom_mask"]))
        return out

    @staticmethod
    def backbone_geometry(pos: torch.Tensor, mask: torch.Tensor) -> dict:
        n, ca, c, o = pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 4]
        L = pos.shape[0]
        bb = Rigid.from_3_points(p_neg_x_axis=c, origin=ca, p_xy_plane=n, eps=1e-8)
        rot = bb.get_rots().get_rot_mats() * torch.tensor([-1.0, 1.0, -1.0])  # compose with diag(-1, 1, -1)
        frames = torch.zeros(L, 8, 4, 4)
        frames[:, 0, :3, :3] = rot
        frames[:, 0, :3, 3] = bb.get_trans()
        frames[:, 0, 3, 3] = 1.0
        tf = Rigid.from_3_points(ca, c, n, eps=1e-8)  # psi: atoms (N, CA, C, O)
        rt = tf.get_rots().get_rot_mats().transpose(-1, -2)
        rel = torch.einsum("nij,nj->ni", rt.double(), o) - torch.einsum("nij,nj->ni", rt, tf.get_trans()).double()
        sc = torch.stack([rel[:, 2], rel[:, 1]], dim=-1)
        sc = sc / torch.sqrt(torch.sum(sc**2, dim=-1, keepdim=True) + 1e-8)
        tors = torch.zeros(L, 7, 2, dtype=torch.float64)
        tors[:, 2] = -sc
        tmask = torch.zeros(L, 7, dtype=torch.float64)
        tmask[:, 2] = mask[:, 0] * mask[:, 1] * mask[:, 2] * mask[:, 4]
        return {"rigidgroups_gt_frames": frames, "torsion_angles_sin_cos": tors, "torsion_angles_mask": tmask}


class RandomAccessProteinDataset(torch.utils.data.Dataset):
    def __init__(self, path_to_dataset: str, transform=None, suffix: str = ".pdb",
                 accession_code_fillter: Optional[Sequence[str]] = None, training: bool = False, **_):
        path_to_dataset = os.path.expanduser(path_to_dataset)
        suffix = suffix if suffix.startswith(".") else "." + suffix
        if suffix != ".pdb":
            raise NotImplementedError("only .pdb inputs are on the sampling path")
        pattern = os.path.join(path_to_dataset, "*" + suffix) if os.path.isdir(path_to_dataset) else path_to_dataset
        self._data = sorted(glob(pattern))
        assert len(self._data) > 0, f"No {suffix} file found in '{path_to_dataset}'"
        if accession_code_fillter:
            keep = set(accession_code_fillter)
            self._data = [p for p in self._data if os.path.splitext(os.path.basename(p))[0] in keep]
        self.data = np.asarray(self._data)
        self.transform, self.suffix, self.training = transform, suffix, training

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        path = self.data[idx]
        with open(path, "r") as f:
            obj = protein.from_pdb_string(f.read()).to_dict()
        if self.transform is not None:
            obj = self.transform(obj)
        obj["accession_code"] = os.path.splitext(os.path.basename(path))[0]
        return obj


class SamplingPDBDataset(RandomAccessProteinDataset):
    def __init__(self, path_to_dataset: str, training: bool = False, suffix: str = ".pdb",
                 transform: Optional[ProteinFeatureTransform] = None,
                 accession_code_fillter: Optional[Sequence[str]] = None):
        assert os.path.isdir(path_to_dataset), f"Invalid path (expected to be directory): {path_to_dataset}"
        super().__init__(path_to_dataset, transform=transform, suffix=suffix,
                         accession_code_fillter=accession_code_fillter, training=training)
