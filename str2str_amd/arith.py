"""One switch for the arithmetic of every matrix product of the sampling path.

  "f16x3" (default)  operands split into f16 pairs, three products per block on the 16-bit matrix cores, fp32 accumulation:
                     fp32-equivalent precision, f16 RANGE (an activation must stay below 65504) -- guarded: the kernels raise a
                     device flag when a value they split reaches 2^15, and the sampler re-runs that chunk in "f32".
  "f32"              exact fp32 MFMA everywhere (pair MLPs csrc/pair_mlp.hip, attention csrc/ipa_attention.hip, node layers
                     s2s_node_linear_f32, encoder attention): the reference arithmetic, no range limit, ~2.5x slower.

Every module that launches matrix kernels (EmbeddingModule, EdgeTransition, InvariantPointAttention, TranslationIPA) carries an
``arith`` attribute, set at construction from ``S2S_ARITH``; ``use_arith(net, mode)`` switches a whole network for a block of code.
"""
from __future__ import annotations

import os

ARITH_MODES = ("f16x3", "f32")


def default_arith() -> str:
    mode = os.environ.get("S2S_ARITH", "f16x3")
    if mode not in ARITH_MODES:
        raise ValueError(f"S2S_ARITH={mode!r}: expected one of {ARITH_MODES}")
    return mode


def arith_modules(net):
    return [m for m in net.modules() if hasattr(m, "arith")]


def net_arith(net) -> str:
    """The arithmetic of a network ("mixed" if its modules disagree)."""
    modes = {m.arith for m in arith_modules(net)}
    return modes.pop() if len(modes) == 1 else ("mixed" if modes else default_arith())


class use_arith:
    """``with use_arith(net, "f32"): ...`` -- every module of ``net`` runs in that arithmetic inside the block."""

    def __init__(self, net, mode: str):
        if mode not in ARITH_MODES:
            raise ValueError(f"arith {mode!r}: expected one of {ARITH_MODES}")
        self.mods, self.mode = arith_modules(net), mode

    def __enter__(self):
        self.prev = [m.arith for m in self.mods]
        for m in self.mods:
            m.arith = self.mode
        return self

    def __exit__(self, *exc):
        for m, v in zip(self.mods, self.prev):
            m.arith = v
