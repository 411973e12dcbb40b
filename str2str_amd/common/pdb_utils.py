"""Sample writers: atom37 coordinates -> multi-MODEL PDB, and ordered merging of PDB files.
File layout and text identical to the reference's ``src/common/pdb_utils.py`` (merge_pdbfiles :31-83,
protein_with_default_params :175-203, atom37_to_pdb :205-252).  The text is produced by the native formatter behind
the C ABI (``s2s_write_pdb_models`` / ``s2s_merge_pdb_files``, csrc/pdb_format.cpp: ~50x the f-string writer, streaming);
``S2S_PDB_WRITER=python`` selects the pure-Python restatement below (both are pinned byte for byte by the golden
texts the reference's own writers produced)."""
from __future__ import annotations

import os
import re
from typing import Optional

import numpy as np

from . import protein


def protein_with_default_params(atom_positions, atom_mask, aatype=None, b_factors=None, chain_index=None,
                                residue_index=None) -> protein.Protein:
    assert atom_positions.ndim == 3 and atom_positions.shape[-2:] == (37, 3)
    n = atom_positions.shape[0]
    sqz = lambda x: np.squeeze(x) if x.shape[0] == 1 and len(x.shape) > 1 else x  # noqa: E731
    return protein.Protein(
        atom_positions=atom_positions, atom_mask=atom_mask,
        aatype=np.zeros(n, dtype=int) if aatype is None else sqz(aatype),
        residue_index=np.arange(n) + 1 if residue_index is None else sqz(residue_index),
        chain_index=np.zeros(n) if chain_index is None else sqz(chain_index),
        b_factors=np.zeros([n, 37]) if b_factors is None else sqz(b_factors),
    )


def atom37_to_pdb(save_to: str, atom_positions: np.ndarray, aatype: Optional[np.ndarray] = None,
                  b_factors: Optional[np.ndarray] = None, chain_index: Optional[np.ndarray] = None,
                  residue_index: Optional[np.ndarray] = None, overwrite: bool = False, no_indexing: bool = True) -> str:
    if not no_indexing:
        idx = 0
        if not overwrite:
            d, stem = os.path.dirname(save_to), os.path.basename(save_to).strip(".pdb")
            found = [re.findall(r"_(\d+).pdb", x) for x in os.listdir(d) if stem in x]
            idx = max([int(f[0]) for f in found if f] + [0])
        save_to = save_to.replace(".pdb", "") + f"_{idx + 1}.pdb"
    if atom_positions.ndim == 3:
        atom_positions = atom_positions[None]
    elif atom_positions.ndim != 4:
        raise ValueError(f"Invalid positions shape {atom_positions.shape}")
    if os.environ.get("S2S_PDB_WRITER", "native") != "python":
        from .. import ops

        ops.write_pdb_models(save_to, atom_positions, aatype=aatype, residue_index=residue_index, chain_index=chain_index,
                             b_factors=b_factors, first_model=1, add_end=2)
        return save_to
    with open(save_to, "w") as f:
        for mi, pos37 in enumerate(atom_positions):
            mask = np.sum(np.abs(pos37), axis=-1) > 1e-7
            prot = protein_with_default_params(pos37, mask, aatype=aatype, b_factors=b_factors, chain_index=chain_index,
                                               residue_index=residue_index)
            f.write(protein.to_pdb(prot, model=mi + 1, add_end=False))
        f.write("END")
    return save_to


class AsyncPdbWriter:
    """Multi-MODEL PDB files written BEHIND the sampler: ``submit`` starts the device -> pinned-host copy of a coordinate tensor on the
    current stream and hands copy + ``atom37_to_pdb`` to one background thread, so the caller goes straight on to the next trajectory
    (the native writer releases the GIL; the reference writes synchronously, src/models/diffusion_module.py:353-360, and its GPU idles
    meanwhile).  ``results()`` waits for every file, in submission order, and re-raises a writer's exception."""

    def __init__(self):
        from concurrent.futures import ThreadPoolExecutor

        self._pool = ThreadPoolExecutor(max_workers=1)
        self._futs = []
        self._pinned = {}

    def submit(self, atom_positions, save_to: str, **kw):
        import torch

        if os.environ.get("S2S_ASYNC_PDB", "1") == "0":      # the reference's order of events: copy, write, then go on
            arr = atom_positions.cpu().numpy() if torch.is_tensor(atom_positions) else atom_positions
            res = atom37_to_pdb(atom_positions=arr, save_to=save_to, **kw)
            self._futs.append(self._pool.submit(lambda: res))
            return
        if not (torch.is_tensor(atom_positions) and atom_positions.is_cuda):
            arr = atom_positions.cpu().numpy() if torch.is_tensor(atom_positions) else atom_positions
            self._futs.append(self._pool.submit(atom37_to_pdb, atom_positions=arr, save_to=save_to, **kw))
            return
        key = (tuple(atom_positions.shape), atom_positions.dtype, len(self._futs) & 1)    # two buffers per shape: copy k + 1 while file k is written
        host = self._pinned.get(key)
        if host is None:
            try:
                host = torch.empty(atom_positions.shape, dtype=atom_positions.dtype, pin_memory=True)
            except RuntimeError:     # no page-locked memory to be had: the reference's order for this file
                arr = atom_positions.cpu().numpy()
                self._futs.append(self._pool.submit(atom37_to_pdb, atom_positions=arr, save_to=save_to, **kw))
                return
            self._pinned[key] = host
        if len(self._futs) >= 2:
            self._futs[-2].result()          # the writer is done with this buffer
        host.copy_(atom_positions, non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def job():
            done.synchronize()
            return atom37_to_pdb(atom_positions=host.numpy(), save_to=save_to, **kw)

        self._futs.append(self._pool.submit(job))

    def results(self):
        out = [f.result() for f in self._futs]
        self._futs = []
        return out

    def close(self, cancel_pending: bool = False):
        """Stop the worker thread and drop the page-locked buffers.  ``cancel_pending``: files still queued are not written (the
        error path: a sampler failure must not leave a worker writing behind the exception)."""
        self._pool.shutdown(wait=True, cancel_futures=cancel_pending)
        self._futs = []
        self._pinned = {}

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close(cancel_pending=exc_type is not None)
        return False


def merge_pdbfiles(input, output_file: str, verbose: bool = True) -> None:
    files = [os.path.join(input, f) for f in os.listdir(input) if f.endswith(".pdb")] if isinstance(input, str) else list(input)
    os.makedirs(os.path.dirname(output_file), exist_ok=True)
    if os.environ.get("S2S_PDB_WRITER", "native") != "python":
        from .. import ops

        ops.merge_pdb_files(files, output_file)
        if verbose:
            print(f"Merged {len(files)} PDB files into {output_file}.")
        return
    model_number = 0
    out = []
    for path in files:
        with open(path, "r") as fh:
            lines = fh.readlines()
        if not any(ln.startswith("MODEL") or ln.startswith("ENDMDL") for ln in lines):
            model_number += 1
            out.append(f"MODEL     {model_number}")
            out += [ln.strip() for ln in lines if ln.startswith("TER") or ln.startswith("ATOM")]
            out.append("ENDMDL")
        else:
            for ln in lines:
                if ln.startswith("MODEL"):
                    model_number += 1
                    if model_number > 1:
                        out.append("ENDMDL")
                    out.append(f"MODEL     {model_number}")
                elif ln.startswith("END"):
                    continue
                elif ln.startswith("TER") or ln.startswith("ATOM"):
                    out.append(ln.strip())
    out += ["ENDMDL", "END"]
    with open(output_file, "w") as fo:
        fo.write("\n".join(x.ljust(80) for x in out) + "\n")
    if verbose:
        print(f"Merged {len(files)} PDB files into {output_file} with {model_number} models.")


_AMINO_ACIDS = frozenset("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split())
# Peptide-linking components of the CCD that occur in experimental / reference structures (biotite's filter_amino_acids, which the
# reference's reader applies, accepts every such component, as ATOM or HETATM records): the common modified residues with the
# standard residue each derives from.  Only membership matters for the C-alpha trace; a residue in neither table is an error.
_MODIFIED_AMINO_ACIDS = {
    "MSE": "MET", "SEC": "CYS", "PYL": "LYS", "HYP": "PRO", "SEP": "SER", "TPO": "THR", "PTR": "TYR", "TYS": "TYR", "CSO": "CYS",
    "CSD": "CYS", "CME": "CYS", "OCS": "CYS", "CSS": "CYS", "CSX": "CYS", "SMC": "CYS", "CAS": "CYS", "KCX": "LYS", "MLY": "LYS",
    "MLZ": "LYS", "M3L": "LYS", "ALY": "LYS", "LLP": "LYS", "PCA": "GLU", "CGU": "GLU", "HIC": "HIS", "NEP": "HIS", "MHS": "HIS",
    "NLE": "LEU", "MLE": "LEU", "ABA": "ALA", "AIB": "ALA", "DAL": "ALA", "ORN": "LYS", "FME": "MET", "MHO": "MET", "SAC": "SER",
    "DSN": "SER", "DTH": "THR", "DVA": "VAL", "DLE": "LEU", "DPR": "PRO", "DPN": "PHE", "DTR": "TRP", "DTY": "TYR", "DAR": "ARG",
    "DAS": "ASP", "DGL": "GLU", "DGN": "GLN", "DLY": "LYS", "DCY": "CYS", "DHI": "HIS", "DIL": "ILE", "MED": "MET", "DSG": "ASN",
    "TRO": "TRP", "HTR": "TRP", "PHI": "PHE", "YCM": "CYS", "AGM": "ARG", "ASX": "ASP", "GLX": "GLU", "UNK": "ALA",
}


def extract_backbone_coords(input_path: str, max_n_model: Optional[int] = None) -> np.ndarray:
    """CA coordinates [n_models, L, 3] float32 of a (multi-MODEL) PDB file, a .npy file or a directory of PDB files
    (reference pdb_utils.py:255-317, which goes through biotite; here a fixed-column scan of the ATOM records)."""
    assert os.path.exists(input_path), f"File {input_path} does not exist."
    if input_path.endswith(".npy"):
        coords = np.load(input_path)
    elif os.path.isdir(input_path):
        coords = np.concatenate([extract_backbone_coords(os.path.join(input_path, f)) for f in os.listdir(input_path)
                                 if f.endswith(".pdb")], axis=0)
    elif input_path.endswith(".pdb"):
        # biotite's reader as the reference calls it (altloc="first", then filter_backbone: amino-acid residues only): the first
        # alternate location of every (chain, residue number, insertion code), standard residue names
        models, cur, seen = [], [], set()
        with open(input_path) as fh:
            for ln in fh:
                if ln.startswith(("ATOM", "HETATM")) and ln[12:16] == " CA ":      # a C-alpha (a calcium ion is "CA  ")
                    if ln[17:20] not in _AMINO_ACIDS and ln[17:20] not in _MODIFIED_AMINO_ACIDS:
                        # biotite's filter_amino_acids accepts every peptide-linking component of the CCD; this reader carries the
                        # standard twenty + the common modified residues (selenomethionine, phosphoserine, ...), as ATOM or HETATM
                        # records like biotite (which does not filter on the record type).  A residue in neither table is an error
                        # here, not a silently shorter chain
                        raise ValueError(f"{input_path}: C-alpha of unknown residue {ln[17:20]!r} {ln[21]}{ln[22:27].strip()}: "
                                         "not one of the standard or common modified amino acids (extend _MODIFIED_AMINO_ACIDS)")
                    key = (ln[21], ln[22:27])          # chain id, resSeq + iCode
                    if key not in seen:                 # later altlocs of a residue already taken are skipped
                        seen.add(key)
                        cur.append((float(ln[30:38]), float(ln[38:46]), float(ln[46:54])))
                elif ln.startswith("ENDMDL"):
                    models.append(cur); cur, seen = [], set()
        if cur:
            models.append(cur)
        lens = {len(m) for m in models}
        if len(lens) > 1:
            raise ValueError(f"{input_path}: models with different numbers of CA atoms {sorted(lens)} (a truncated last model?)")
        coords = np.asarray(models, dtype=np.float32)
    else:
        raise ValueError(f"Unrecognized input path {input_path}.")
    if max_n_model is not None and len(coords) > max_n_model > 0:
        coords = coords[:max_n_model]
    return coords
