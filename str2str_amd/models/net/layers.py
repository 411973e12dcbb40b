"""Per-node layers and the pair-stream ``EdgeTransition`` of the score network.

Same constructor arguments, parameter names and shapes as the reference's
``src/models/net/layers.py`` (Linear :64-124, NodeTransition :128-145, EdgeTransition :148-185,
TorsionAngleHead :188-213, BackboneUpdate :216-241) so a reference checkpoint loads unchanged.
In the sampling path every layer here is evaluated by the fused node kernels (``ops.node_apply``, called from
``TranslationIPA.forward`` on the packed parameters) and the N x N ``EdgeTransition`` by ``s2s_edge_transition_f16x3`` (or the exact
fp32 ``s2s_edge_transition``, see str2str_amd/arith.py); the plain ``forward`` methods of the small per-node modules are the
reference's module API (torch ops), kept for callers outside the sampler.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...arith import default_arith


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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """y = x W^T + b on the node kernel (s2s_node_linear / s2s_node_linear_f32 in the configured arithmetic) for callers outside
        the sampler, which packs and fuses whole chains of these layers itself (TranslationIPA._node_weights).  No library GEMM and
        no CPU path: every matrix product of the package runs on the package's own kernels."""
        if not x.is_cuda:
            raise ops.HipLibraryError("Linear.forward runs on the HIP device only (the package has no CPU / BLAS matrix path; "
                                      "CPU restatements live under oracle/ for the tests)")
        if not hasattr(self, "_pack"):
            self._pack = ParamCache()
        bias = self.bias if self.bias is not None else torch.zeros(self.out_features, device=x.device)
        layer = self._pack.get([self.weight, bias], lambda: ops.pack_node_layer(self.weight, bias))
        if self.in_features % 32:
            raise ops.HipLibraryError(f"Linear.forward: the node kernel wants in_features in multiples of 32 (got {self.in_features})")
        lead, M = x.shape[:-1], x.numel() // x.shape[-1]
        y, _ = ops.node_apply(ops.to_act(x.reshape(M, x.shape[-1]).float().contiguous(), default_arith()), layer, M)
        return y[:, : self.out_features].reshape(*lead, self.out_features)


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
        self._cache32 = ParamCache()
        self._proj_cache = ParamCache()   # (a captured HIP graph keeps pointing at its stream: sampler._GraphedNet pins it)
        self._node_cache = ParamCache()
        self.arith = default_arith()      # "f16x3" (csrc/pair_mlp_f16.hip) | "f32" (csrc/pair_mlp.hip): see str2str_amd/arith.py
        self.prescale_exp = 0             # f16x3: block exponent of the hidden activations (planes hold 2^-e x the value; the sampler raises
        #                                   it when the range guard reports activations of 2^15 and beyond: sampler.denoise_loop)

    def _packed(self):
        """Weight stream of the split-f16 kernel."""
        w1, w2, wf = self.trunk[0], self.trunk[2], self.final_layer
        ce = self._shape[0]
        return self._cache.get([w1.weight, w2.weight, wf.weight], lambda: {
            "wstream_f16": ops.pack_f16x3_stream(w1.weight[:, :ce].float(), w2.weight.float(), wf.weight.float())})

    def _packed_f32(self):
        w1, w2, wf = self.trunk[0], self.trunk[2], self.final_layer
        ce = self._shape[0]
        return self._cache32.get([w1.weight, w2.weight, wf.weight], lambda: {
            "w1p": ops.pack_weight(w1.weight[:, :ce].float(), tile_major=True),
            "w2p": ops.pack_weight(w2.weight.float(), tile_major=True),
            "wfp": ops.pack_weight(wf.weight.float(), tile_major=True)})

    def node_layers(self):
        """The per-node parts as node-stream layers: n' = initial_embed(node) and everything a pair (i, j) takes from ONE of its nodes
        through a linear layer, [W1[:, ce:ce+cb] n' + b1 | W1[:, ce+cb:] n' | Wf[:, ce+cb:] n' + bf]  (384 + 384 + 128 columns): the row
        half A_i and the column half B_j of the first hidden layer, and G_j = the j-side residual of the final layer's input
        x = h2 + [e | n'_i | n'_j] (reference layers.py:181) taken through that layer, its bias included."""
        w1, ie, wf = self.trunk[0], self.initial_embed, self.final_layer

        def build():
            ce, cb = self._shape[0], self._shape[1]
            w_ab = torch.cat([w1.weight[:, ce:ce + cb], w1.weight[:, ce + cb:], wf.weight[:, ce + cb:]], dim=0).float().contiguous()
            b_ab = torch.cat([w1.bias, torch.zeros_like(w1.bias), wf.bias]).float().contiguous()
            # "ab_s": the same vectors straight from s -- W_ab (W_ie s + b_ie) + b_ab = (W_ab W_ie) s + (W_ab b_ie + b_ab), folded in float64 on
            # the host -- so that both per-node parts read the block's node activations and run in ONE launch with the backbone update
            w64, ie64 = w_ab.detach().double().cpu(), ie.weight.detach().double().cpu()
            w_abs = (w64 @ ie64).float().to(w_ab.device)
            b_abs = (w64 @ ie.bias.detach().double().cpu() + b_ab.detach().double().cpu()).float().to(w_ab.device)
            # "abA" + "abB" / "ab_sA" + "ab_sB": the same layers in the FORM the f16x3 pair kernel reads (include/str2str_hip.h) -- the column
            # half B_j starts the layer-1 accumulators, which carry 2^5 x the layer output.  The factor does not go into the weights (it
            # would cost the f16x3 packing five bits of its weight range): the column half is its own layer whose epilogue scales the row
            # by 32 (``pre_scale``: acc (32 / 32) + 32 b, exact) and writes columns 384.. of the same [M, 768] buffer (``node_ab16``)
            h2 = self._shape[2]     # 384: A | B | G(128); B and G are accumulator start values (x 2^5): ONE layer of 512 columns
            return {"init": ops.pack_node_layer(ie.weight, ie.bias), "ab": ops.pack_node_layer(w_ab, b_ab),
                    "ab_s": ops.pack_node_layer(w_abs, b_abs),
                    "abA": ops.pack_node_layer(w_ab[:h2].contiguous(), b_ab[:h2].contiguous()),
                    "abB": ops.pack_node_layer(w_ab[h2:].contiguous(), 32.0 * b_ab[h2:]),
                    "ab_sA": ops.pack_node_layer(w_abs[:h2].contiguous(), b_abs[:h2].contiguous()),
                    "ab_sB": ops.pack_node_layer(w_abs[h2:].contiguous(), 32.0 * b_abs[h2:])}

        return self._node_cache.get([w1.weight, w1.bias, ie.weight, ie.bias, wf.weight, wf.bias], build)

    @staticmethod
    def ab16_specs(nl: dict, from_s: bool, n_rows: int, device):
        """node_ab in the f16x3 pair kernel's form [A_i + b1 | 32 B_j | 32 G_j] as two layers into one buffer -> (buffer [M, 896], specs
        for ``ops.node_apply_multi`` / ``ops.node_apply``)."""
        ab = torch.empty(n_rows, 896, device=device, dtype=torch.float32)
        c32 = ops.const_rows(n_rows, 32.0, device)
        ka, kb = ("ab_sA", "ab_sB") if from_s else ("abA", "abB")
        return ab, [(nl[ka], dict(out_f32=ab, out_col0=0)), (nl[kb], dict(out_f32=ab, out_col0=384, pre_scale=c32))]

    def node_parts(self, s_act, n_rows: int, kernel_form: bool = False):
        """-> (n' [M,128] fp32, node_ab [M,896] fp32) from the node activations (packed planes or fp32, see ops.node_apply).
        ``kernel_form``: node_ab as the f16x3 pair kernel reads it (column half and G x 2^5; ``pair_mlp(..., ab_kernel_form=True)``)."""
        nl = self.node_layers()
        n_p, n_pa = ops.node_apply(s_act, nl["init"], n_rows, want_xp=True)
        if kernel_form:
            node_ab, specs = self.ab16_specs(nl, False, n_rows, n_p.device)
            for layer, kw in specs:
                ops.node_apply(n_pa, layer, n_rows, **kw)
        else:
            node_ab, _ = ops.node_apply(n_pa, nl["ab"], n_rows)
        return n_p, node_ab

    def forward(self, node_embed: torch.Tensor, edge_embed: torch.Tensor, edge_mask_1d: Optional[torch.Tensor] = None,
                next_proj=None):
        """edge_embed [B,N,N,c_z] -> [B,N,N,c_z].  ``edge_mask_1d`` (node mask [B,N]) optionally fuses
        the caller's ``* edge_mask[..., None]`` (reference ipa.py:372) into the kernel epilogue; ``next_proj``
        = InvariantPointAttention.pair_proj_weights() of the NEXT IPA block additionally returns its (attn_bias, pair_z)."""
        B, N = node_embed.shape[:2]
        n_p, node_ab = self.node_parts(ops.to_act(node_embed.reshape(B * N, -1).float().contiguous(), self.arith), B * N)
        return self.pair_mlp(edge_embed, node_ab.view(B, N, -1), n_p.view(B, N, -1), edge_mask_1d, next_proj)

    def pair_mlp(self, edge_embed, node_ab, n_p, edge_mask_1d=None, next_proj=None, out_layout: str = "rowmajor", ab_kernel_form: bool = False):
        """The N x N part given the per-node vectors n' = initial_embed(node) [B,N,128] and
        node_ab = [W1[:,128:256] n' + b1 | W1[:,256:] n' | Wf[:,256:] n' + bf] [B,N,896] (``node_parts``).  Arithmetic "f16x3" only: ``edge_embed`` may be
        an ``ops.PairTiled`` and ``out_layout`` "tiled" / "none" (ops.edge_transition_f16x3) -- how the trunk chains its pair kernels."""
        if self._shape != (128, 128, 384, 128, 2):
            raise ops.HipLibraryError(f"EdgeTransition kernel is built for c_z=128, c_s=256 (got {self._shape})")
        mask = None if edge_mask_1d is None else edge_mask_1d.type(torch.float32).contiguous()
        if self.arith == "f16x3":
            pk = self._packed()
            proj = None
            if next_proj is not None:  # 31-stage stream: this layer's 30 stages + the next block's projection stage
                stream = self._proj_cache.get([pk["wstream_f16"], next_proj["wp_f16x2"]],
                                              lambda: torch.cat([pk["wstream_f16"], next_proj["wp_f16x2"]]))
                proj = (stream, next_proj["b64"])
            tiled_in = isinstance(edge_embed, ops.PairTiled)
            B, N = edge_embed.shape[0], edge_embed.shape[1]
            z, bias, pz = torch.ops.str2str_amd.edge_transition_f16x3_chain(
                edge_embed.buf if tiled_in else edge_embed.contiguous(), tiled_in, B, N, node_ab, n_p,
                pk["wstream_f16"] if proj is None else proj[0], self.trunk[2].bias, self.layer_norm.weight,
                self.layer_norm.bias, mask, self.layer_norm.eps, None if proj is None else proj[1], out_layout, int(self.prescale_exp),
                bool(ab_kernel_form))
            if out_layout == "tiled":
                z = ops.PairTiled(B, N, buf=z)
            return z if proj is None else (z, bias, pz)
        if out_layout != "rowmajor" or isinstance(edge_embed, ops.PairTiled) or ab_kernel_form:
            raise ops.HipLibraryError("EdgeTransition: the tiled pair layout and the scaled node_ab belong to the f16x3 kernels")
        pk = self._packed_f32()
        return ops.edge_transition(edge_embed.contiguous(), node_ab[..., :768].contiguous(), n_p, pk["w1p"], pk["w2p"], pk["wfp"],
                                   self.trunk[2].bias, self.final_layer.bias, self.layer_norm.weight,
                                   self.layer_norm.bias, mask, self.layer_norm.eps,
                                   proj=None if next_proj is None else (next_proj["wp"], next_proj["b64"]))


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
