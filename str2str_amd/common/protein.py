"""``Protein`` record + PDB text I/O of the sampling boundary.

Same fields and the same fixed-column text as the reference's ``src/common/protein.py`` (Protein :34-69,
from_pdb_string :72-143, to_pdb :152-234): output files must stay byte-compatible.  The reader is a
self-contained fixed-column ATOM/HETATM parser (the reference delegates to Bio.PDB, which is not a
dependency here); it keeps the reference's rules: single MODEL only, no insertion codes, unknown
residue names -> 'X' (index 20), atoms outside the 37 standard names ignored, residues without any
known atom skipped, chain ids mapped to integers in sorted order.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

from . import residue_constants as rc

PDB_CHAIN_IDS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"
PDB_MAX_CHAINS = len(PDB_CHAIN_IDS)  # 62


@dataclasses.dataclass(frozen=True)
class Protein:
    atom_positions: np.ndarray  # [num_res, 37, 3] Angstrom
    aatype: np.ndarray  # [num_res]
    atom_mask: np.ndarray  # [num_res, 37]
    residue_index: np.ndarray  # [num_res] as in the PDB
    chain_index: np.ndarray  # [num_res]
    b_factors: np.ndarray  # [num_res, 37]

    def __post_init__(self):
        if len(np.unique(self.chain_index)) > PDB_MAX_CHAINS:
            raise ValueError(f"Cannot build an instance with more than {PDB_MAX_CHAINS} chains "
                             "because these cannot be written to PDB format.")

    def to_dict(self):
        return dataclasses.asdict(self)


def from_pdb_string(pdb_str: str, chain_id: Optional[str] = None) -> Protein:
    n_models = 0
    chains = {}  # chain id -> {(resseq, icode, hetflag): dict(resname, atoms{name: (xyz, b, occ)})} in file order
    order = []
    for line in pdb_str.splitlines():
        rec = line[:6]
        if rec.startswith("MODEL"):
            n_models += 1
            if n_models > 1:
                raise ValueError("Only single model PDBs are supported. Found more than 1 model.")
            continue
        if rec not in ("ATOM  ", "HETATM"):
            continue
        name = line[12:16].strip()
        altloc = line[16]
        resname = line[17:20].strip()
        ch = line[21]
        resseq = int(line[22:26])
        icode = line[26]
        xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
        occ = float(line[54:60]) if line[54:60].strip() else 1.0
        bf = float(line[60:66]) if line[60:66].strip() else 0.0
        if chain_id is not None and ch != chain_id:
            continue
        if icode != " ":
            raise ValueError(f"PDB contains an insertion code at chain {ch} and residue index {resseq}. "
                             "These are not supported.")
        if ch not in chains:
            chains[ch] = {}
            order.append(ch)
        key = (resseq, "H_" + resname if rec == "HETATM" else " ")  # Bio.PDB residue id = (hetfield, resseq, icode)
        res = chains[ch].setdefault(key, {"resname": resname, "resseq": resseq, "atoms": {}})
        prev = res["atoms"].get(name)
        if prev is None or (altloc != " " and occ > prev[2]):  # keep the best-occupied alternate location
            res["atoms"][name] = (xyz, bf, occ)

    pos_l, aa_l, mask_l, idx_l, ch_l, b_l = [], [], [], [], [], []
    for ch in order:
        for res in chains[ch].values():
            short = rc.restype_3to1.get(res["resname"], "X")
            pos = np.zeros((rc.atom_type_num, 3))
            mask = np.zeros((rc.atom_type_num,))
            bfs = np.zeros((rc.atom_type_num,))
            for name, (xyz, bf, _) in res["atoms"].items():
                k = rc.atom_order.get(name)
                if k is None:
                    continue
                pos[k], mask[k], bfs[k] = xyz, 1.0, bf
            if np.sum(mask) < 0.5:
                continue
            aa_l.append(rc.restype_order.get(short, rc.restype_num))
            pos_l.append(pos); mask_l.append(mask); idx_l.append(res["resseq"]); ch_l.append(ch); b_l.append(bfs)
    uniq = np.unique(ch_l)
    cmap = {c: n for n, c in enumerate(uniq)}
    return Protein(atom_positions=np.array(pos_l), atom_mask=np.array(mask_l), aatype=np.array(aa_l),
                   residue_index=np.array(idx_l), chain_index=np.array([cmap[c] for c in ch_l]), b_factors=np.array(b_l))


def _chain_end(atom_index, end_resname, chain_name, residue_index) -> str:
    return f"{'TER':<6}{atom_index:>5}      {end_resname:>3} {chain_name:>1}{residue_index:>4}"


def to_pdb(prot: Protein, model: int = 1, add_end: bool = True) -> str:
    restypes = rc.restypes + ["X"]
    res3 = lambda r: rc.restype_1to3.get(restypes[r], "UNK")  # noqa: E731
    aatype = prot.aatype
    residue_index = prot.residue_index.astype(int)
    chain_index = prot.chain_index.astype(int)
    if np.any(aatype > rc.restype_num):
        raise ValueError("Invalid aatypes.")
    chain_ids = {}
    for i in np.unique(chain_index):
        if i >= PDB_MAX_CHAINS:
            raise ValueError(f"The PDB format supports at most {PDB_MAX_CHAINS} chains.")
        chain_ids[i] = PDB_CHAIN_IDS[i]
    lines = [f"MODEL     {model}"]
    atom_index = 1
    last_chain = chain_index[0]
    for i in range(aatype.shape[0]):
        if last_chain != chain_index[i]:
            lines.append(_chain_end(atom_index, res3(aatype[i - 1]), chain_ids[chain_index[i - 1]], residue_index[i - 1]))
            last_chain = chain_index[i]
            atom_index += 1
        name3 = res3(aatype[i])
        for atom_name, pos, mask, b in zip(rc.atom_types, prot.atom_positions[i], prot.atom_mask[i], prot.b_factors[i]):
            if mask < 0.5 or (name3 == "GLY" and atom_name == "CB"):
                continue
            name = atom_name if len(atom_name) == 4 else f" {atom_name}"
            lines.append(f"{'ATOM':<6}{atom_index:>5} {name:<4}{'':>1}{name3:>3} {chain_ids[chain_index[i]]:>1}"
                         f"{residue_index[i]:>4}{'':>1}   {pos[0]:>8.3f}{pos[1]:>8.3f}{pos[2]:>8.3f}"
                         f"{1.0:>6.2f}{b:>6.2f}          {atom_name[0]:>2}{'':>2}")
            atom_index += 1
    lines.append(_chain_end(atom_index, res3(aatype[-1]), chain_ids[chain_index[-1]], residue_index[-1]))
    lines.append("ENDMDL")
    if add_end:
        lines.append("END")
    return "\n".join(line.ljust(80) for line in lines) + "\n"
