"""Does a checkpoint fit the default (split-f16, "f16x3") arithmetic of the sampling path?  The first thing to run on a trained
`pretrain.pth` (reference README.md:112, configs/eval.yaml:18, src/utils/checkpoint_utils.py:3-27):

    python tools/check_checkpoint.py <pretrain.pth | synthetic | trained_like> [<target.pdb> | --n-res N] [--replicas B]

One network evaluation per t in {0.01, 0.5, 1.0} on the target (noised to t by the forward marginal, device noise, fixed seed), on
the f16x3 kernels with the range buffer cleared before each (csrc/range_flag.h) and once more on the exact fp32 kernels.  Prints
  * per kernel family (node stream, edge transition, edge embedding, IPA): the upper bound of max |x| the kernels saw, as a fraction
    of the guard's limit 2^15 (f16 tops out at 65504) -- "headroom" -- and which families would be demoted to fp32 by the sampler;
  * whether every weight fits the f16x3 packing (|32 w| < 65504);
  * the difference between the two arithmetics' predicted frames (backbone RMSD in Angstrom; the parity bar of the path is 1e-4);
  * a verdict: "f16x3 holds" | "holds with <families> on fp32 (automatic)" | "run with S2S_ARITH=f32".
"synthetic" / "trained_like" stand in for a checkpoint (seeded weights of fresh-initialisation / trained-like magnitudes, synth.py).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("pdb", nargs="?", default=None)
    ap.add_argument("--n-res", type=int, default=64, help="synthetic chain length when no PDB is given")
    ap.add_argument("--replicas", type=int, default=2)
    ap.add_argument("--json", action="store_true", help="one JSON object instead of the table")
    a = ap.parse_args()

    from str2str_amd import ops
    from str2str_amd.arith import FAMILIES, use_arith
    from str2str_amd.common.all_atom import compute_backbone
    from str2str_amd.factory import build_diffuser, build_net
    from str2str_amd.synth import synth_chain, synth_state_dict

    dev = torch.device("cuda")
    net = build_net()
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    if a.checkpoint in ("synthetic", "trained_like"):
        sd = synth_state_dict(manifest, seed=0, sigma_final=0.02 if a.checkpoint == "trained_like" else 0.002,
                              style="trained_like" if a.checkpoint == "trained_like" else "init")
    else:
        params = torch.load(a.checkpoint, map_location="cpu")["state_dict"]      # the reference's container (checkpoint_utils.py:16-20)
        sd = {k.replace("net.", "", 1) if k.startswith("net.") else k: v for k, v in params.items()}
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()

    if a.pdb:
        from str2str_amd.common import protein
        from str2str_amd.data.components.dataset import ProteinFeatureTransform

        f = ProteinFeatureTransform()(protein.from_pdb_string(open(a.pdb).read()).to_dict())
        feats = {k: (v[None] if torch.is_tensor(v) else v) for k, v in f.items()}
    else:
        feats = synth_chain(a.n_res)
    B, N = a.replicas, int(feats["aatype"].shape[1])
    diff = build_diffuser(os.path.join("/tmp", "str2str_cache_check"))
    rep = lambda v: v.to(dev).repeat(B, *(1,) * (v.ndim - 1))  # noqa: E731
    gt4 = rep(feats["rigidgroups_gt_frames"][..., 0, :, :].float())
    mask = rep(feats["residue_mask"].float())

    # weights against the f16x3 packing
    big = {k: float(v.abs().max()) for k, v in sd.items() if v.ndim == 2 and float(v.abs().max()) * 32 >= 65504}
    rows, demote = [], set()
    for t in (0.01, 0.5, 1.0):
        torch.cuda.manual_seed(1234)
        r_t = diff.forward_marginal_device(gt4, t, mask)
        batch = {k: rep(feats[k]) for k in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")}
        batch.update(rigids_t=r_t, sc_ca_t=torch.zeros(B, N, 3, device=dev), t=torch.full((B,), t))
        out = {}
        for mode in ("f16x3", "f32"):
            ops.range_flag_reset()
            try:
                with torch.no_grad(), use_arith(net, mode):
                    o = net(batch)
                bb = compute_backbone(o["rigids"], o["psi"], aatype=batch["aatype"], _rigids7=o["rigids7"])[0][..., :5, :]
                out[mode] = bb.double().cpu().numpy()
            except ops.WeightRangeError as e:
                out[mode] = None
                big.setdefault("(packing)", str(e))
            if mode == "f16x3":
                bits, head = ops.range_flag_read(), ops.range_headroom()
        fams = ops.range_families(bits) if out["f16x3"] is not None else list(FAMILIES)
        demote |= set(fams)
        rmsd = None
        if out["f16x3"] is not None and out["f32"] is not None:
            d = out["f16x3"] - out["f32"]
            rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
        rows.append({"t": t, "headroom": head, "flagged": fams, "finite": bool(out["f16x3"] is not None and np.isfinite(out["f16x3"]).all()),
                     "backbone_rmsd_f16x3_vs_f32_A": rmsd})
    if big:
        verdict = "run with S2S_ARITH=f32 (weights beyond the f16x3 packing: the sampler falls back by itself, every family)"
    elif not demote:
        verdict = "f16x3 holds (no family reached 2^15 at t = 0.01 / 0.5 / 1.0)"
    elif len(demote) < len(FAMILIES):
        verdict = f"holds with {' + '.join(sorted(demote))} on exact fp32 (the sampler demotes those families automatically)"
    else:
        verdict = "run with S2S_ARITH=f32 (every kernel family leaves f16's range)"
    res = {"checkpoint": a.checkpoint, "target": a.pdb or f"synthetic N={N}", "n_res": N, "replicas": B, "evaluations": rows,
           "weights_beyond_packing": big, "verdict": verdict}
    if a.json:
        print(json.dumps(res))
        return
    print(f"checkpoint {a.checkpoint}   target {res['target']}   {B} replicas")
    print(f"{'t':>5} | " + " | ".join(f"{f:>16}" for f in FAMILIES) + " | flagged -> fp32        | f16x3 vs f32 backbone RMSD")
    for r in rows:
        print(f"{r['t']:5.2f} | " + " | ".join(f"{'< ' if r['headroom'][f] < 1 else ''}{r['headroom'][f]:>13.4g}x" for f in FAMILIES)
              + f" | {', '.join(r['flagged']) or '-':22s} | " + ("n/a" if r["backbone_rmsd_f16x3_vs_f32_A"] is None else f"{r['backbone_rmsd_f16x3_vs_f32_A']:.2e} A"))
    print("(columns: upper bound of max |x| seen by that kernel family / 2^15; below 1 = inside the guard's limit)")
    if big:
        print("weights beyond |32 w| < 65504:", big)
    print("verdict:", verdict)


if __name__ == "__main__":
    main()
