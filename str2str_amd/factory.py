"""Construct the sampling objects with the reference's default hyper-parameters
(configs/model/diffusion.yaml:17-55 of the reference) — used by bench.py, smoke() and tests; the Hydra
entry instantiates the same classes from the YAML instead."""
from __future__ import annotations

import torch

from .models.net.denoising_ipa import DenoisingNet, EmbeddingModule
from .models.net.ipa import TranslationIPA
from .models.score.frame import FrameDiffuser
from .models.score.r3 import R3Diffuser
from .models.score.so3 import SO3Diffuser
from .synth import synth_state_dict


def build_net() -> DenoisingNet:
    emb = EmbeddingModule(init_embed_size=32, node_embed_size=256, edge_embed_size=128, num_bins=22, min_bin=1e-5,
                          max_bin=20.0, self_conditioning=True)
    tr = TranslationIPA(c_s=256, c_z=128, coordinate_scaling=0.1, no_ipa_blocks=4, skip_embed_size=64,
                        transformer_num_heads=4, transformer_num_layers=2, c_hidden=256, no_heads=8, no_qk_points=8,
                        no_v_points=12, dropout=0.0)
    return DenoisingNet(emb, tr).eval()


def build_diffuser(cache_dir: str = "./cache") -> FrameDiffuser:
    return FrameDiffuser(
        trans_diffuser=R3Diffuser(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
        rot_diffuser=SO3Diffuser(cache_dir=cache_dir, num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5,
                                 schedule="logarithmic", use_cached_score=False),
        min_t=1e-2,
    )


def build_synthetic_net(seed: int = 0, sigma_final: float = 0.002, device="cuda") -> DenoisingNet:
    net = build_net()
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(synth_state_dict(manifest, seed=seed, sigma_final=sigma_final), strict=True)
    return net.to(device).eval()
