"""One switch for the arithmetic of every matrix product of the sampling path.

  "f16x3" (default)  operands split into f16 pairs, three products per block on the 16-bit matrix cores, fp32 accumulation:
                     fp32-equivalent precision, f16 RANGE (an activation must stay below 65504) -- guarded: the kernels raise a
                     device flag when a value they split reaches 2^15, and the sampler re-runs that chunk in "f32".
  "f32"              exact fp32 MFMA everywhere (pair MLPs csrc/pair_mlp.hip, attention csrc/ipa_attention.hip, node layers
                     s2s_node_linear_f32, encoder attention): the reference arithmetic, no range limit, ~2.5x slower.

Every module that launches matrix kernels (EmbeddingModule, EdgeTransition, InvariantPointAttention, TranslationIPA) carries an
``arith`` attribute, set at construction from ``S2S_ARITH``; ``use_arith(net, mode)`` switches a whole network for a block of code,
``use_arith(net, mode, families=(...))`` only the named kernel families (below).
"""
from __future__ import annotations

import os

ARITH_MODES = ("f16x3", "f32")


def default_arith() -> str:
    mode = os.environ.get("S2S_ARITH", "f16x3")
    if mode not in ARITH_MODES:
        raise ValueError(f"S2S_ARITH={mode!r}: expected one of {ARITH_MODES}")
    return mode


# Kernel FAMILIES (what the range guard names, csrc/range_flag.h bits; ops.RANGE_FAMILIES) and the modules whose ``arith`` selects
# their kernels.  The families are independent: any subset may run on the exact kernels while the rest stays on f16x3 (the pair tensor
# changes layout at the boundary, the node stream's format follows the trunk) -- which is how the sampler answers a raised flag: only
# the family that overflowed pays the fp32 price.
#   "edge_transition"  EdgeTransition.arith                  (csrc/pair_mlp_f16.hip | pair_mlp.hip; 73 % of the cfg2 step)
#   "edge_embed"       EmbeddingModule.arith                 (edge embedding kernels)
#   "ipa"              InvariantPointAttention.arith         (point preparation + attention core)
#   "node"             TranslationIPA.arith, EmbeddingModule.node_arith   (node GEMMs, pack_planes, encoder attention: the activation
#                                                              format of the node stream, packed f16 planes | fp32 row-major)
FAMILIES = ("node", "edge_transition", "edge_embed", "ipa")
_FAMILY_OF_CLASS = {"EdgeTransition": "edge_transition", "EmbeddingModule": "edge_embed", "InvariantPointAttention": "ipa",
                    "TranslationIPA": "node"}


def arith_modules(net):
    return [m for m in net.modules() if hasattr(m, "arith")]


def family_slots(net, families=None):
    """[(module, attribute)] of the arithmetic switches of ``families`` (all when None)."""
    out = []
    for m in arith_modules(net):
        fam = _FAMILY_OF_CLASS.get(type(m).__name__)
        if families is None or fam in families:
            out.append((m, "arith"))
        if hasattr(m, "node_arith") and (families is None or "node" in families):
            out.append((m, "node_arith"))
    return out


def net_arith(net) -> str:
    """The arithmetic of a network ("mixed" if its modules disagree)."""
    modes = {getattr(m, a) for m, a in family_slots(net)}
    return modes.pop() if len(modes) == 1 else ("mixed" if modes else default_arith())


def family_arith(net) -> dict:
    """{family: "f16x3" | "f32" | "mixed"} as the network is configured now."""
    out = {}
    for fam in FAMILIES:
        modes = {getattr(m, a) for m, a in family_slots(net, (fam,))}
        out[fam] = modes.pop() if len(modes) == 1 else ("mixed" if modes else default_arith())
    return out


class use_arith:
    """``with use_arith(net, "f32"): ...`` -- every module of ``net`` (or only the kernel ``families`` named) runs in that arithmetic
    inside the block."""

    def __init__(self, net, mode: str, families=None):
        if mode not in ARITH_MODES:
            raise ValueError(f"arith {mode!r}: expected one of {ARITH_MODES}")
        if families is not None and not set(families) <= set(FAMILIES):
            raise ValueError(f"families {families!r}: expected a subset of {FAMILIES}")
        self.slots, self.mode = family_slots(net, None if families is None else tuple(families)), mode

    def __enter__(self):
        self.prev = [getattr(m, a) for m, a in self.slots]
        for m, a in self.slots:
            setattr(m, a, self.mode)
        return self

    def __exit__(self, *exc):
        for (m, a), v in zip(self.slots, self.prev):
            setattr(m, a, v)
