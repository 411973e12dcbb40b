"""The handful of chemistry tables the I/O boundary needs (atom37 naming, residue codes).

Standard amino-acid nomenclature as used by the reference's ``src/common/residue_constants.py``
(:492-499 atom_types, :534-540 restypes, :589-617 3-letter codes); the idealised backbone geometry
lives in ``str2str_amd/data/backbone_tables.py``.
"""
atom_types = [
    "N", "CA", "C", "CB", "O", "CG", "CG1", "CG2", "OG", "OG1", "SG", "CD", "CD1", "CD2", "ND1", "ND2", "OD1", "OD2",
    "SD", "CE", "CE1", "CE2", "CE3", "NE", "NE1", "NE2", "OE1", "OE2", "CH2", "NH1", "NH2", "OH", "CZ", "CZ2", "CZ3",
    "NZ", "OXT",
]
atom_order = {name: i for i, name in enumerate(atom_types)}
atom_type_num = len(atom_types)  # 37

restypes = list("ARNDCQEGHILKMFPSTWYV")
restype_order = {r: i for i, r in enumerate(restypes)}
restype_num = len(restypes)  # 20; index 20 = unknown 'X'

restype_1to3 = {
    "A": "ALA", "R": "ARG", "N": "ASN", "D": "ASP", "C": "CYS", "Q": "GLN", "E": "GLU", "G": "GLY", "H": "HIS",
    "I": "ILE", "L": "LEU", "K": "LYS", "M": "MET", "F": "PHE", "P": "PRO", "S": "SER", "T": "THR", "W": "TRP",
    "Y": "TYR", "V": "VAL",
}
restype_3to1 = {v: k for k, v in restype_1to3.items()}
