"""I/O fixtures from the reference (authoring container only): feature dicts of three bundled PDBs
through the reference's ProteinFeatureTransform, and the text its PDB writers emit.

The reference's reader needs Bio.PDB (not installed), so the parsed Protein dict comes from this
repo's reader and is cross-checked by a to_pdb round trip; everything downstream (featurisation,
to_pdb, atom37_to_pdb, merge_pdbfiles) is the reference's own code, imported with empty stubs for the
packages it imports but does not use on this path (Bio.PDB, biotite, lightning, hydra.utils)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()
for name in ["Bio", "Bio.PDB", "biotite", "biotite.structure", "biotite.structure.io", "biotite.structure.io.pdb",
             "lightning", "hydra", "hydra.utils", "pandas_stub"]:
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
sys.modules["Bio.PDB"].PDBParser = object
sys.modules["biotite.structure.io.pdb"].PDBFile = object
sys.modules["lightning"].LightningDataModule = object
sys.modules["hydra.utils"].instantiate = lambda *a, **k: None
sys.modules["biotite.structure"].io = sys.modules["biotite.structure.io"]
sys.modules["biotite"].structure = sys.modules["biotite.structure"]
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
from src.common import pdb_utils as ref_pdb_utils  # noqa: E402
from src.common import protein as ref_protein  # noqa: E402
from src.data.components.dataset import ProteinFeatureTransform as RefTransform  # noqa: E402

from str2str_amd.common import protein as my_protein  # noqa: E402

KEYS = ["aatype", "residue_mask", "fixed_mask", "residue_idx", "residue_index", "chain_index", "atom_positions", "atom_mask"]
out = {}
for code in ["CLN025", "NuG2", "lambda"]:
    with open(os.path.join(HERE, "pdb", f"{code}.pdb")) as f:
        txt = f.read()
    prot = my_protein.from_pdb_string(txt)
    feats = RefTransform(strip_missing_residues=False, recenter_and_scale=False)(prot.to_dict())
    for k in KEYS:
        out[f"{code}/{k}"] = feats[k].numpy()
    out[f"{code}/bb_frame"] = feats["rigidgroups_gt_frames"][:, 0].numpy()
    out[f"{code}/psi"] = feats["torsion_angles_sin_cos"][:, 2].numpy()
    out[f"{code}/psi_mask"] = feats["torsion_angles_mask"][:, 2].numpy()
    # reference writer on the parsed protein (round trip of the reader)
    rp = ref_protein.Protein(**prot.to_dict())
    with open(os.path.join(HERE, f"io_{code}_to_pdb.txt"), "w") as f:
        f.write(ref_protein.to_pdb(rp))
np.savez_compressed(os.path.join(HERE, "io_features.npz"), **out)

# multi-MODEL writer + merge on synthetic coordinates (2 models, GLY + unknown residue included)
g = np.random.default_rng(0)
N = 7
pos = np.zeros((2, N, 37, 3), dtype=np.float32)
pos[:, :, :5] = g.normal(size=(2, N, 5, 3)).astype(np.float32) * 10
aatype = np.array([[0, 7, 19, 20, 4, 7, 13]])
pos[:, 1, 3] = 0  # GLY: CB slot zero as compute_backbone leaves it
pos[:, 5, 3] = 0
chain = np.array([[0, 0, 0, 0, 1, 1, 1]])
resi = np.array([[3, 4, 5, 6, 10, 11, 12]])
d = os.path.join(HERE, "_tmp_io")
os.makedirs(os.path.join(d, "0.25"), exist_ok=True)
os.makedirs(os.path.join(d, "0.3"), exist_ok=True)
p1 = ref_pdb_utils.atom37_to_pdb(save_to=os.path.join(d, "0.25", "x.pdb"), atom_positions=pos, aatype=aatype, chain_index=chain, residue_index=resi)
p2 = ref_pdb_utils.atom37_to_pdb(save_to=os.path.join(d, "0.3", "x.pdb"), atom_positions=pos[:1] + 1.0, aatype=aatype, chain_index=chain, residue_index=resi)
ref_pdb_utils.merge_pdbfiles([p1, p2], os.path.join(d, "all_delta", "x.pdb"), verbose=False)
np.savez_compressed(os.path.join(HERE, "io_writer_inputs.npz"), pos=pos, aatype=aatype, chain_index=chain, residue_index=resi)
import shutil  # noqa: E402

shutil.copy(p1, os.path.join(HERE, "io_atom37_two_models.pdb.txt"))
shutil.copy(os.path.join(d, "all_delta", "x.pdb"), os.path.join(HERE, "io_merged.pdb.txt"))
shutil.rmtree(d)
print("ok", {k: v.shape for k, v in out.items() if k.startswith("CLN025")})
