"""PDB I/O + featurisation + config/CLI plumbing of the drop-in boundary (CPU only).  Expected values
come from the reference's own writers / featuriser (tests/golden/make_golden_io.py)."""
import os
import subprocess
import sys

import numpy as np
import torch

from conftest import GOLDEN, ROOT, golden, maxdiff
from str2str_amd.common import pdb_utils, protein
from str2str_amd.data.components.dataset import ProteinFeatureTransform, SamplingPDBDataset
from str2str_amd.data.protein_datamodule import ProteinDataModule

CODES = ["CLN025", "NuG2", "lambda"]


def _read(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return f.read()


def test_reader_round_trip_is_byte_identical_to_reference_writer():
    for code in CODES:
        src = _read(f"pdb/{code}.pdb")
        prot = protein.from_pdb_string(src)
        assert protein.to_pdb(prot) == _read(f"io_{code}_to_pdb.txt")
        # coordinates survive the trip: every ATOM line's xyz columns equal the source file's
        src_atoms = [l[30:54] for l in src.splitlines() if l.startswith("ATOM") and l[12:16].strip() in protein.rc.atom_order]
        out_atoms = [l[30:54] for l in protein.to_pdb(prot).splitlines() if l.startswith("ATOM")]
        assert src_atoms == out_atoms and len(out_atoms) > 50


def test_reader_rules():
    base = _read("pdb/CLN025.pdb")
    two = "MODEL 1\n" + base + "ENDMDL\nMODEL 2\n" + base + "ENDMDL\n"
    try:
        protein.from_pdb_string(two)
        assert False
    except ValueError as e:
        assert "single model" in str(e).lower()
    line = [l for l in base.splitlines() if l.startswith("ATOM")][0]
    bad = line[:26] + "A" + line[27:]
    try:
        protein.from_pdb_string(bad)
        assert False
    except ValueError as e:
        assert "insertion code" in str(e)
    # unknown residue name -> index 20, hydrogens / unknown atom names ignored
    odd = "\n".join(l[:17] + "XYZ" + l[20:] if l.startswith("ATOM") else l for l in base.splitlines())
    assert (protein.from_pdb_string(odd).aatype == 20).all()


def test_featuriser_matches_reference_pipeline():
    g = golden("io_features.npz")
    tf = ProteinFeatureTransform(strip_missing_residues=False, recenter_and_scale=False)
    for code in CODES:
        feats = tf(protein.from_pdb_string(_read(f"pdb/{code}.pdb")).to_dict())
        for k in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "residue_index", "chain_index"):
            assert (feats[k].numpy() == g[f"{code}/{k}"]).all(), (code, k)
            assert feats[k].numpy().dtype == g[f"{code}/{k}"].dtype, (code, k)
        fr = feats["rigidgroups_gt_frames"]
        assert fr.dtype == torch.float32 and fr.shape[1:] == (8, 4, 4)
        assert maxdiff(fr[:, 0], g[f"{code}/bb_frame"]) < 1e-6, code
        assert maxdiff(feats["torsion_angles_sin_cos"][:, 2], g[f"{code}/psi"]) < 1e-6, code
        assert (feats["torsion_angles_mask"][:, 2].numpy() == g[f"{code}/psi_mask"]).all()


def test_writers_byte_identical(tmp_path):
    g = golden("io_writer_inputs.npz")
    kw = dict(aatype=g["aatype"], chain_index=g["chain_index"], residue_index=g["residue_index"])
    os.makedirs(tmp_path / "0.25")
    os.makedirs(tmp_path / "0.3")
    p1 = pdb_utils.atom37_to_pdb(save_to=str(tmp_path / "0.25" / "x.pdb"), atom_positions=g["pos"], **kw)
    p2 = pdb_utils.atom37_to_pdb(save_to=str(tmp_path / "0.3" / "x.pdb"), atom_positions=g["pos"][:1] + 1.0, **kw)
    assert open(p1).read() == _read("io_atom37_two_models.pdb.txt")
    assert not open(p1).read().endswith("\n") and open(p1).read().endswith("END")  # reference quirk: no final newline
    pdb_utils.merge_pdbfiles([p1, p2], str(tmp_path / "all_delta" / "x.pdb"), verbose=False)
    assert open(tmp_path / "all_delta" / "x.pdb").read() == _read("io_merged.pdb.txt")


def test_dataset_and_collate():
    ds = SamplingPDBDataset(os.path.join(GOLDEN, "pdb"), accession_code_fillter=["CLN025", "NuG2", "lambda"],
                            transform=ProteinFeatureTransform(strip_missing_residues=False, recenter_and_scale=False))
    assert len(ds) == 3 and ds[0]["accession_code"] == "CLN025"
    # the whole Science2011 fast-folder set (BASELINE configs[2]): 12 targets, lengths as SURVEY section 8 counted them
    full = SamplingPDBDataset(os.path.join(GOLDEN, "pdb"), transform=ProteinFeatureTransform())
    assert sorted(int(full[i]["aatype"].shape[0]) for i in range(len(full))) == [10, 20, 28, 35, 35, 39, 47, 47, 52, 56, 73, 80]
    dm = ProteinDataModule(ds, batch_size=1)
    batches = dm.test_dataloader()
    assert len(batches) == 3 and batches[1]["aatype"].shape == (1, 56) and batches[1]["accession_code"] == ["NuG2"]
    assert batches[0]["atom_positions"].dtype == torch.float64 and batches[0]["aatype"].dtype == torch.int64
    two = ProteinDataModule(ds, batch_size=2).test_dataloader()[0]  # pad-collate to the longer chain
    assert two["residue_mask"].shape == (2, 56) and float(two["residue_mask"][0, 10:].sum()) == 0


def test_eval_cli_plumbing(tmp_path):
    """configs/eval.yaml composes (defaults list, env + node interpolation, CLI overrides), objects instantiate from
    their _target_s, the checkpoint contract loads, and a CPU trainer fails loudly instead of falling back."""
    from str2str_amd.utils import config as C

    env = dict(os.environ, TEST_DATA=os.path.join(GOLDEN, "pdb"), CACHE_DIR=str(tmp_path / "cache"), PROJECT_ROOT=str(tmp_path))
    os.environ.update({k: env[k] for k in ("TEST_DATA", "CACHE_DIR", "PROJECT_ROOT")})
    cfg = C.compose(os.path.join(ROOT, "configs"), "eval.yaml",
                    ["task_name=inference", "ckpt_path=null", "trainer=cpu", "model.inference.n_replica=3",
                     "data.dataset.accession_code_fillter=[NuG2]"])
    assert cfg.task_name == "inference" and cfg.trainer.accelerator == "cpu" and cfg.model.inference.n_replica == 3
    assert cfg.model.inference.min_t == 1e-2 and cfg.data.dataset.transform.eps == 1e-8
    assert cfg.model.inference.output_dir.endswith("/samples") and "/inference/runs/" in cfg.paths.output_dir
    assert cfg.model.diffuser.rot_diffuser.cache_dir == str(tmp_path / "cache")
    dm = C.instantiate(cfg.data)
    assert len(dm.dataset) == 1
    model = C.instantiate(cfg.model)
    assert model.hparams.inference.replica_per_batch == 64 and len(model.net.state_dict()) == 274
    # checkpoint contract: {'state_dict': {'net.<key>': tensor}} strict-loads into model.net
    sys.path.insert(0, ROOT)
    import eval as entry

    sd = {"net." + k: torch.randn_like(v) for k, v in model.net.state_dict().items()}
    torch.save({"state_dict": sd}, tmp_path / "w.pth")
    model, rest = entry.load_model_checkpoint(model, str(tmp_path / "w.pth"))
    assert rest is None and torch.equal(model.net.state_dict()["embedder.node_embed.0.weight"], sd["net.embedder.node_embed.0.weight"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "task_name=inference", "ckpt_path=null", "trainer=cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_native_writer_byte_identical_to_reference_text(tmp_path, monkeypatch):
    """The native formatter behind the C ABI (csrc/pdb_format.cpp; host code, no GPU needed) against the texts the
    REFERENCE's writers produced: to_pdb of the three bundled targets, the two-model atom37 file, the merged file — and
    against the Python restatement on awkward values (ties at the third decimal, -0.0, |x| >= 1000, 5-digit atom serials,
    multiple chains, unknown residues, non-zero B factors)."""
    from str2str_amd import ops

    for code in CODES:
        prot = protein.from_pdb_string(_read(f"pdb/{code}.pdb"))
        pos = (prot.atom_positions * prot.atom_mask[..., None]).astype(np.float32)  # masked atoms are exactly zero
        txt = ops.format_pdb_models(pos, aatype=prot.aatype, residue_index=prot.residue_index, chain_index=prot.chain_index,
                                    b_factors=prot.b_factors, add_end=1)
        assert txt == _read(f"io_{code}_to_pdb.txt"), code
    g = golden("io_writer_inputs.npz")
    kw = dict(aatype=g["aatype"], chain_index=g["chain_index"], residue_index=g["residue_index"])
    for writer in ("native", "python"):
        monkeypatch.setenv("S2S_PDB_WRITER", writer)
        d = tmp_path / writer
        os.makedirs(d / "0.25"); os.makedirs(d / "0.3")
        p1 = pdb_utils.atom37_to_pdb(save_to=str(d / "0.25" / "x.pdb"), atom_positions=g["pos"], **kw)
        p2 = pdb_utils.atom37_to_pdb(save_to=str(d / "0.3" / "x.pdb"), atom_positions=g["pos"][:1] + 1.0, **kw)
        assert open(p1).read() == _read("io_atom37_two_models.pdb.txt"), writer
        pdb_utils.merge_pdbfiles([p1, p2], str(d / "all_delta" / "x.pdb"), verbose=False)
        assert open(d / "all_delta" / "x.pdb").read() == _read("io_merged.pdb.txt"), writer
    # awkward values: native == Python restatement (which is pinned to the reference above)
    rng = np.random.default_rng(0)
    n = 2300  # > 9999 atoms in one model -> 5-digit serials; residue numbers > 999
    pos = np.zeros((2, n, 37, 3), dtype=np.float32)
    pos[:, :, :5] = rng.normal(0, 40, size=(2, n, 5, 3)).astype(np.float32)
    pos[0, 0, 0] = [0.0625, -0.0625, 0.1875]      # exact ties at the third decimal (round half to even)
    pos[0, 1, 1] = [-0.0004, 1234.5675, -999.9995]
    pos[0, 2, 2] = [2.5e-8, 2.5e-8, 2.5e-8]       # below the 1e-7 mask threshold -> no line
    pos[0, 3, 4] = [-0.0, 0.0, 1e-6]
    aat = rng.integers(0, 21, size=n)
    chain = np.repeat(np.arange(4), n // 4 + 1)[:n]
    resi = np.arange(n) * 3 + 5
    bf = np.zeros((n, 37)); bf[:, 1] = rng.uniform(0, 99, size=n)
    kw = dict(aatype=aat, chain_index=chain, residue_index=resi, b_factors=bf)
    monkeypatch.setenv("S2S_PDB_WRITER", "python")
    want = open(pdb_utils.atom37_to_pdb(save_to=str(tmp_path / "py.pdb"), atom_positions=pos, **kw)).read()
    monkeypatch.setenv("S2S_PDB_WRITER", "native")
    got = open(pdb_utils.atom37_to_pdb(save_to=str(tmp_path / "na.pdb"), atom_positions=pos, **kw)).read()
    assert got == want and "  0.062  -0.062   0.188" in got and got.count("MODEL") == 2
    assert ops.format_pdb_models(pos, **kw) == want
    try:
        ops.format_pdb_models(pos, aatype=aat + 5)
        assert False
    except ValueError as e:
        assert "Invalid aatypes" in str(e)


def test_extract_backbone_coords_altloc_and_nonstandard(tmp_path):
    """Reference reader semantics (biotite, altloc='first' + filter_backbone, pdb_utils.py:255-317): the first alternate location of
    a residue, amino-acid residues only (standard + common modified ones; an unknown one is an error here), equal model lengths (a ragged file is an error, not a
    silent ragged array)."""
    import pytest

    from str2str_amd.common.pdb_utils import extract_backbone_coords

    def atom(serial, name, alt, res, chain, seq, x):
        return f"ATOM  {serial:5d} {name:^4s}{alt}{res:>3s} {chain}{seq:4d}    {x:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00           C  "

    model = [atom(1, "CA", "A", "ALA", "A", 1, 1.0), atom(2, "CA", "B", "ALA", "A", 1, 9.0),      # altloc B of residue 1: skipped
             atom(3, "CA", " ", "GLY", "A", 2, 2.0),
             "HETATM    9 CA    CA A 900       5.000   0.000   0.000  1.00  0.00          CA  ",          # a calcium ion is not a C-alpha
             atom(5, "CA", " ", "SER", "A", 4, 3.0)]
    txt = "\n".join(["MODEL        1"] + model + ["ENDMDL", "MODEL        2"] + model + ["ENDMDL", "END"])
    p = tmp_path / "alt.pdb"
    p.write_text(txt)
    ca = extract_backbone_coords(str(p))
    assert ca.shape == (2, 3, 3) and ca[0, :, 0].tolist() == [1.0, 2.0, 3.0]
    # a modified residue (selenomethionine as a HETATM record, as experimental files carry it) is part of the chain, as in biotite's
    # amino-acid filter (CCD peptide-linking components, any record type); a residue in neither table is an error, not a silently
    # shorter chain
    (tmp_path / "mse.pdb").write_text("\n".join(model[:3] + [atom(4, "CA", " ", "MSE", "A", 3, 7.0).replace("ATOM  ", "HETATM")] + ["END"]))
    ca = extract_backbone_coords(str(tmp_path / "mse.pdb"))
    assert ca.shape == (1, 3, 3) and ca[0, :, 0].tolist() == [1.0, 2.0, 7.0]
    (tmp_path / "xyz.pdb").write_text("\n".join(model[:3] + [atom(4, "CA", " ", "XYZ", "A", 3, 7.0)] + ["END"]))
    with pytest.raises(ValueError, match="XYZ"):
        extract_backbone_coords(str(tmp_path / "xyz.pdb"))
    (tmp_path / "ragged.pdb").write_text("\n".join(["MODEL        1"] + model + ["ENDMDL", "MODEL        2"] + model[:3]))
    with pytest.raises(ValueError):
        extract_backbone_coords(str(tmp_path / "ragged.pdb"))


def test_async_writer_gives_the_same_files_in_submission_order(tmp_path, monkeypatch):
    """AsyncPdbWriter (the writer predict_step hands its ensembles to): same bytes as atom37_to_pdb called directly, results in
    submission order, a failing write surfaces in results(); host arrays and host tensors take the same path as device tensors
    minus the pinned copy (no GPU here)."""
    import torch

    g = golden("io_writer_inputs.npz")
    kw = dict(aatype=g["aatype"], chain_index=g["chain_index"], residue_index=g["residue_index"])
    for mode in ("1", "0"):
        monkeypatch.setenv("S2S_ASYNC_PDB", mode)
        d = tmp_path / mode
        os.makedirs(d)
        w = pdb_utils.AsyncPdbWriter()
        w.submit(g["pos"], str(d / "a.pdb"), **kw)
        w.submit(torch.as_tensor(g["pos"][:1] + 1.0), str(d / "b.pdb"), **kw)
        w.submit(g["pos"], str(d / "c.pdb"), **kw)
        assert w.results() == [str(d / "a.pdb"), str(d / "b.pdb"), str(d / "c.pdb")]
        assert open(d / "a.pdb").read() == _read("io_atom37_two_models.pdb.txt") == open(d / "c.pdb").read()
        want = pdb_utils.atom37_to_pdb(save_to=str(d / "b_direct.pdb"), atom_positions=g["pos"][:1] + 1.0, **kw)
        assert open(d / "b.pdb").read() == open(want).read()
        w.close()
    monkeypatch.setenv("S2S_ASYNC_PDB", "1")
    w = pdb_utils.AsyncPdbWriter()
    w.submit(g["pos"], str(tmp_path / "no_such_dir" / "x.pdb"), **kw)
    import pytest

    with pytest.raises(Exception):
        w.results()
    w.close()
