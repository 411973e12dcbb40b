"""One network evaluation of the REFERENCE at the bench shape (BASELINE configs[1]: N = 256), B = 1, with the same
seeded synthetic weights as make_golden.py -> tests/golden/net_b1n256.npz (inputs + frames / psi / backbone).
Run in the authoring container (needs /root/reference):  python tests/golden/make_golden_n256.py  (~1 min on CPU)."""
import torch

import make_golden as G  # noqa: E402  (sets up the reference import shim)


def main():
    net, _ = G.build_net(seed=0, sigma_final=0.02)
    g = torch.Generator().manual_seed(140)
    batch = G.make_batch(g, 1, 256, False)
    with torch.no_grad():
        out = net(batch)
    G.npz("net_b1n256.npz", **{f"in_{k}": v for k, v in batch.items()}, rigids7=out["rigids"].to_tensor_7(), psi=out["psi"],
          atom37=out["atom37"][..., :5, :], atom14=out["atom14"][..., :5, :])


if __name__ == "__main__":
    main()
