"""Per-node layers and the pair-stream ``EdgeTransition`` of the score network.

Same constructor arguments, parameter names and shapes as the reference's
``src/models/net/layers.py`` (Linear :64-124, NodeTransition :128-145, EdgeTransition :148-185,
TorsionAngleHead :188-213, BackboneUpdate :216-241) so a reference checkpoint loads unchanged.
Per-node (N-linear) layers are dense projections on the GPU BLAS (fp32 MFMA GEMMs); the N x N
``EdgeTransition`` runs the fused fp32-MFMA kernel ``s2s_edge_transition`` (csrc/pair_mlp.hip).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


def _trunc_normal_(w: torch.Tensor, scale: float):
    """AF2-style fan-in truncated normal (reference :31-41); the 0.8796... constant is the std of a
    unit normal truncated to [-2, 2]."""
    fan_in = w.shape[1]
    std = math.sqrt(scale / max(1, fan_in)) / 0.87962566103423978
    with torch.no_grad():
        nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)


class Linear(nn.Linear):
    def __init__(self, in_dim: int, out_dim: int, bias: bool = True, init: str = "default", init_fn=None):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.fill_(0)
            if init_fn is not None:
                init_fn(self.weight, self.bias)
            elif init == "default":
                _trunc_normal_(self.weight, 1.0)
            elif init == "relu":
                _trunc_normal_(self.weight, 2.0)
            elif init == "glorot":
                nn.init.xavier_uniform_(self.weight, gain=1)
            elif init == "gating":
                self.weight.fill_(0.0)
                if bias:
                    self.bias.fill_(1.0)
            elif init == "normal":
                nn.init.kaiming_normal_(self.weight, nonlinearity="linear")
            elif init == "final":
                self.weight.fill_(0.0)
            else:
                raise ValueError("Invalid init string.")


class ParamCache:
    """Derived device tensors (packed / concatenated weights) keyed on the source parameters'
    storage and version counters, so load_state_dict / .to() invalidate them."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, params, build):
        key = tuple((p.data_ptr(), p._version, p.device) for p in params)
        if key != self._key:
            with torch.no_grad():
                self._val = build()
            self._key = key
        return self._val

    def pinned(self):
        """The cached value (a captured HIP graph keeps it alive beyond the next rebuild; sampler._GraphedNet)."""
        return self._val


class NodeTransition(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.linear_1 = Linear(dim, dim, init="relu")
        self.linear_2 = Linear(dim, dim, init="relu")
        self.linear_3 = Linear(dim, dim, init="final")
        self.relu = nn.ReLU()
        self.ln = nn.LayerNorm(dim)

    def forward(self, s: torch.Tensor) -> torch.Tensor:
        x = F.relu(self.linear_1(s))
        x = F.relu(self.linear_2(x))
        return self.ln(self.linear_3(x) + s)


class EdgeTransition(nn.Module):
    def __init__(self, node_embed_size: int, edge_embed_in: int, edge_embed_out: int, num_layers: int = 2,
                 node_dilation: int = 2):
        super().__init__()
        bias_embed_size = node_embed_size // node_dilation
        self.initial_embed = Linear(node_embed_size, bias_embed_size, init="relu")
        hidden = bias_embed_size * 2 + edge_embed_in
        layers = []
        for _ in range(num_layers):
            layers += [Linear(hidden, hidden, init="relu"), nn.ReLU()]
        self.trunk = nn.Sequential(*layers)
        self.final_layer = Linear(hidden, edge_embed_out, init="final")
        self.layer_norm = nn.LayerNorm(edge_embed_out)
        self._shape = (edge_embed_in, bias_embed_size, hidden, edge_embed_out, num_layers)
        self._cache = ParamCache()
        self._proj_cache = ParamCache()
        self._proj_cache_f16 = ParamCache()   # one slot per arithmetic mode: a captured HIP graph keeps pointing at its stream
        # "f16x3" (default): two-way f16 split of both operands (11 + 11 bits + sign = fp32's 24), three products per block with
        # exact 2^+-5 scalings of the small factors, fp32 accumulation (csrc/pair_mlp_f16.hip): 1.65x faster than
        # "bf16x6": exact 3-way bf16 split, six plane-pair products (csrc/pair_mlp_bf16.hip), which is 1.6x faster than
        # "f32": v_mfma_f32_32x32x2_f32 (csrc/pair_mlp.hip).  All three pass the same parity suite.
        self.mfma_mode = os.environ.get("S2S_EDGE_MFMA", "f16x3")
        # "f16x3": two-way f16 split, three products per block (csrc/pair_mlp_f16.hip); the embedder treats it as "bf16x6".
        if self.mfma_mode not in ("bf16x6", "f16x3", "f32"):
            raise ValueError(f"S2S_EDGE_MFMA={self.mfma_mode!r}: expected 'bf16x6', 'f16x3' or 'f32'")

    def _packed(self):
        w1, w2, wf = self.trunk[0], self.trunk[2], self.final_layer

        def build():
            ce = self._shape[0]
            return {
                "w1p": ops.pack_weight(w1.weight[:, :ce].float(), tile_major=True),
                "w2p": ops.pack_weight(w2.weight.float(), tile_major=True),
                "wfp": ops.pack_weight(wf.weight.float(), tile_major=True),
                # node halves of layer 1: [W1[:, ce:ce+cb] ; W1[:, ce+cb:]] applied to n' (+ b1 on the row part)
                "wstream": ops.pack_bf16x3_stream(w1.weight[:, :ce].float(), w2.weight.float(), wf.weight.float()),
                "wstream_f16": ops.pack_f16x3_stream(w1.weight[:, :ce].float(), w2.weight.float(), wf.weight.float()),
                "w_ab": torch.cat([w1.weight[:, ce:ce + self._shape[1]], w1.weight[:, ce + self._shape[1]:]], dim=0).float().contiguous(),
                "b_ab": torch.cat([w1.bias, torch.zeros_like(w1.bias)]).float().contiguous(),
            }

        return self._cache.get([w1.weight, w1.bias, w2.weight, wf.weight], build)

    def forward(self, node_embed: torch.Tensor, edge_embed: torch.Tensor, edge_mask_1d: Optional[torch.Tensor] = None,
                next_proj=None):
        """edge_embed [B,N,N,c_z] -> [B,N,N,c_z].  ``edge_mask_1d`` (node mask [B,N]) optionally fuses
        the caller's ``* edge_mask[..., None]`` (reference ipa.py:372) into the kernel epilogue; ``next_proj``
        = (packed [linear_b; down_z], bias64) of the NEXT IPA block additionally returns its (attn_bias, pair_z)."""
        n_p = self.initial_embed(node_embed).contiguous()
        node_ab = F.linear(n_p, self._packed()["w_ab"], self._packed()["b_ab"]).contiguous()
        return self.pair_mlp(edge_embed, node_ab, n_p, edge_mask_1d, next_proj)

    def pair_mlp(self, edge_embed, node_ab, n_p, edge_mask_1d=None, next_proj=None):
        """The N x N part given the per-node vectors n' = initial_embed(node) [B,N,128] and
        node_ab = [W1[:,128:256] n' + b1 | W1[:,256:] n'] [B,N,768] (computed by the fused node path or by ``forward``)."""
        if self._shape != (128, 128, 384, 128, 2):
            raise ops.HipLibraryError(f"EdgeTransition kernel is built for c_z=128, c_s=256 (got {self._shape})")
        pk = self._packed()
        mask = None if edge_mask_1d is None else edge_mask_1d.type(torch.float32).contiguous()
        if self.mfma_mode == "f16x3":
            proj = None
            if next_proj is not None:
                stream = self._proj_cache_f16.get([pk["wstream_f16"], next_proj[3]], lambda: torch.cat([pk["wstream_f16"], next_proj[3]]))
                proj = (stream, next_proj[1])
            return ops.edge_transition_f16x3(edge_embed.contiguous(), node_ab, n_p, pk["wstream_f16"], self.trunk[2].bias,
                                             self.final_layer.bias, self.layer_norm.weight, self.layer_norm.bias, mask,
                                             self.layer_norm.eps, proj=proj)
        if self.mfma_mode == "bf16x6":
            proj = None
            if next_proj is not None:  # 31-stage stream: this layer's 30 stages + the next block's projection stage
                stream = self._proj_cache.get([pk["wstream"], next_proj[2]], lambda: torch.cat([pk["wstream"], next_proj[2]]))
                proj = (stream, next_proj[1])
            return ops.edge_transition_bf16x6(edge_embed.contiguous(), node_ab, n_p, pk["wstream"], self.trunk[2].bias,
                                              self.final_layer.bias, self.layer_norm.weight, self.layer_norm.bias, mask,
                                              self.layer_norm.eps, proj=proj)
        return ops.edge_transition(edge_embed.contiguous(), node_ab, n_p, pk["w1p"], pk["w2p"], pk["wfp"],
                                   self.trunk[2].bias, self.final_layer.bias, self.layer_norm.weight,
                                   self.layer_norm.bias, mask, self.layer_norm.eps,
                                   proj=None if next_proj is None else next_proj[:2])


class TorsionAngleHead(nn.Module):
    def __init__(self, in_dim: int, n_torsion_angles: int, eps: float = 1e-8):
        super().__init__()
        self.linear_1 = Linear(in_dim, in_dim, init="relu")
        self.linear_2 = Linear(in_dim, in_dim, init="relu")
        self.linear_3 = Linear(in_dim, in_dim, init="final")  # present in checkpoints, unused (reference :194 vs :199-213)
        self.linear_final = Linear(in_dim, n_torsion_angles * 2, init="final")
        self.relu = nn.ReLU()
        self.eps = eps

    def forward(self, s: torch.Tensor) -> torch.Tensor:
        x = self.linear_2(F.relu(self.linear_1(s))) + s
        u = self.linear_final(x)
        return u / torch.sqrt(torch.clamp(torch.sum(u**2, dim=-1, keepdim=True), min=self.eps))


class BackboneUpdate(nn.Module):
    def __init__(self, c_s: int):
        super().__init__()
        self.c_s = c_s
        self.linear = Linear(c_s, 6, init="final")

    def forward(self, s: torch.Tensor) -> torch.Tensor:
        return self.linear(s)
