"""Invariant Point Attention and the ``TranslationIPA`` trunk on the HIP kernels.

Interface, constructor arguments and ``state_dict`` keys follow the reference's
``src/models/net/ipa.py`` (InvariantPointAttention :34-268, TranslationIPA :271-387).  Point frames,
pair projections (linear_b / down_z), logits, softmax, o / o_pt / o_pair are the HIP launches
``s2s_ipa_prep_points_f16``, ``s2s_pair_project`` (normally fused into the producer of z), ``s2s_ipa_attention_f16w`` and
``s2s_ipa_opair`` -- for every chain length (operands padded per sample to the kernel's 32-residue tiles).
Every dense layer of the node stream (q / kv / point projections, linear_out, skip_embed, the transformer's
projections and feed-forward, trunk.linear, NodeTransition, BackboneUpdate, the per-node parts of EdgeTransition, the
torsion head) runs on ``s2s_node_linear`` (csrc/node_gemm.hip: split-f16 MFMA, activations travelling as packed f16 pair
planes) with bias / ReLU / mask / residual / LayerNorm fused into its epilogue -- ``TranslationIPA.forward``.
``arith = "f32"`` (str2str_amd/arith.py; ``S2S_ARITH`` at construction, or the sampler's fallback when the range guard of the f16
kernels fires) runs the same graph on the exact fp32 kernels: ``s2s_ipa_attention`` on fp32 projections, ``s2s_node_linear_f32``
with fp32 row-major activations.  No BLAS / SDPA call is left in this module.
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
from ...common.rigid_utils import Rigid, Rotation
_CHAIN4 = os.environ.get("S2S_CHAIN4", "1") != "0"   # trunk.linear + NodeTransition as one launch (s2s_node_chain)
_ENC_CHAIN3 = os.environ.get("S2S_ENC_CHAIN3", "1") != "0"   # an encoder layer's out_proj + norm1 + feed-forward + norm2 as one launch
from .layers import BackboneUpdate, EdgeTransition, Linear, NodeTransition, ParamCache, TorsionAngleHead


class _LazyPack(dict):
    """dict whose named entries are built when first read."""

    def __init__(self, eager: dict, **lazy):
        super().__init__(eager)
        self._lazy = lazy

    def __missing__(self, key):
        if key not in self._lazy:
            raise KeyError(key)
        self[key] = self._lazy[key]()
        return self[key]


class InvariantPointAttention(nn.Module):
    def __init__(self, c_s: int, c_z: int, c_hidden: int, no_heads: int, no_qk_points: int, no_v_points: int,
                 inf: float = 1e5, eps: float = 1e-8):
        super().__init__()
        self.c_s, self.c_z, self.c_hidden = c_s, c_z, c_hidden
        self.no_heads, self.no_qk_points, self.no_v_points = no_heads, no_qk_points, no_v_points
        self.inf, self.eps = inf, eps
        hc = c_hidden * no_heads
        self.linear_q = Linear(c_s, hc)
        self.linear_kv = Linear(c_s, 2 * hc)
        self.linear_q_points = Linear(c_s, no_heads * no_qk_points * 3)
        self.linear_kv_points = Linear(c_s, no_heads * (no_qk_points + no_v_points) * 3)
        self.linear_b = Linear(c_z, no_heads)
        self.down_z = Linear(c_z, c_z // 4)
        self.head_weights = nn.Parameter(torch.full((no_heads,), 0.541324854612918))  # softplus^-1(1)
        self.linear_out = Linear(no_heads * (c_z // 4 + c_hidden + no_v_points * 4), c_s, init="final")
        self.softmax = nn.Softmax(dim=-1)
        self.softplus = nn.Softplus()
        self.arith = default_arith()   # "f16x3" (s2s_ipa_attention_f16w) | "f32" (s2s_ipa_attention), see str2str_amd/arith.py
        self.fold = os.environ.get("S2S_IPA_FOLD", "1") != "0"   # f16 path: K = V = s, W_k / W_v folded into q' and linear_out (_folded_packs)
        self._last_folded = False
        self._cache = ParamCache()
        self._packs = ParamCache()

    def _derived(self):
        def build():
            wcat = torch.cat([self.linear_b.weight, self.down_z.weight], dim=0).float()
            bcat = torch.cat([self.linear_b.bias, self.down_z.bias]).float()
            b64 = bcat.new_zeros(64)
            b64[: bcat.numel()] = bcat
            hw = F.softplus(self.head_weights.float()) * math.sqrt(1.0 / (3 * (self.no_qk_points * 9.0 / 2)))
            wcat64 = wcat.new_zeros(64, wcat.shape[1])
            wcat64[: wcat.shape[0]] = wcat
            # "wp_f16x2" (the projection as one f16x3 weight stage) is packed on first use: the packing refuses weights beyond f16's
            # range (ops.WeightRangeError), and the exact kernels -- the fallback for exactly that case -- must not trip over it
            return _LazyPack({"wp": ops.pack_weight(wcat), "b64": b64.contiguous(), "hw": hw.contiguous()},
                             wp_f16x2=lambda: ops.pack_f16x2_layer(wcat64, "chain").reshape(-1).view(torch.int16).contiguous())

        return self._cache.get([self.linear_b.weight, self.linear_b.bias, self.down_z.weight, self.down_z.bias,
                                self.head_weights], build)

    def node_packs(self):
        """Node-stream layers of this block's projections (ops.pack_node_layer; k and v rows of linear_kv also separately: the
        f16 attention kernel consumes k as packed planes and v as A fragments)."""
        def build():
            H, C = self.no_heads, self.c_hidden
            pk = ops.pack_node_layer
            wkv = self.linear_kv.weight.view(H, 2, C, -1)
            bkv = self.linear_kv.bias.view(H, 2, C)
            d = {"q": pk(self.linear_q.weight, self.linear_q.bias), "kv": pk(self.linear_kv.weight, self.linear_kv.bias),
                 "k": pk(wkv[:, 0].reshape(H * C, -1), bkv[:, 0].reshape(-1)),
                 "v": pk(wkv[:, 1].reshape(H * C, -1), bkv[:, 1].reshape(-1)),
                 "qp": pk(self.linear_q_points.weight, self.linear_q_points.bias),
                 "kvp": pk(self.linear_kv_points.weight, self.linear_kv_points.bias),
                 "out": pk(self.linear_out.weight, self.linear_out.bias, True)}
            if self.c_s == C:
                d.update(self._folded_packs(wkv, bkv))
            return d

        return self._packs.get([p for lin in (self.linear_q, self.linear_kv, self.linear_q_points, self.linear_kv_points,
                                              self.linear_out) for p in (lin.weight, lin.bias)], build)

    def _folded_packs(self, wkv, bkv):
        """FOLDED projections of the f16 attention path (include/str2str_hip.h, s2s_ipa_attention_f16w): per head
            q_i . k_j = s_j . (W_k^T W_q s_i + W_k^T b_q) + (terms of i alone: they cancel in the softmax over j)
            sum_j a_ij v_j = W_v (sum_j a_ij s_j) + b_v                                   (the probabilities sum to one)
        so the attention reads the block's input s as K and V operand of every head (c_s = c_hidden), linear_k and linear_v are never
        evaluated -- 2/3 of the projections' arithmetic and output bytes, and 16 of the 17 K / V images the attention streamed -- and
        W_v / b_v move into linear_out.  "qf": s -> q' = (W_k^T W_q) s + W_k^T b_q [H c_s];  "outf": linear_out on [sum a s | o_pt ...].
        The products are formed in float64 and rounded once to fp32 (reference weights: ipa.py:131-143,166-171,259-266)."""
        H, C, cs = self.no_heads, self.c_hidden, self.c_s
        dev = self.linear_q.weight.device
        c64 = lambda t: t.detach().double().cpu()  # noqa: E731  (weight preparation on the host, like every packer: once per load_state_dict)
        wq, bq = c64(self.linear_q.weight).view(H, C, cs), c64(self.linear_q.bias).view(H, C)
        wk, wv, bv = c64(wkv[:, 0]), c64(wkv[:, 1]), c64(bkv[:, 1])
        w_qf = torch.einsum("hca,hcb->hab", wk, wq).reshape(H * cs, cs).float().to(dev)
        b_qf = torch.einsum("hca,hc->ha", wk, bq).reshape(-1).float().to(dev)
        wo = c64(self.linear_out.weight)
        wo_o = wo[:, : H * C].view(-1, H, C)
        w_of = torch.cat([torch.einsum("ohc,hca->oha", wo_o, wv).reshape(-1, H * cs), wo[:, H * C:]], dim=1).float().to(dev)
        b_of = (c64(self.linear_out.bias) + torch.einsum("ohc,hc->o", wo_o, bv)).float().to(dev)
        return {"qf": ops.pack_node_layer(w_qf, b_qf), "outf": ops.pack_node_layer(w_of, b_of, True)}

    def use_f16(self, n_res: int, n_rows: int = 0) -> bool:
        """Does the pre-split f16 operand kernel serve this call?  It is built for the reference configuration (c_hidden 256,
        8 / 12 points, c_z/4 = 32, 8 heads), takes any length, and addresses its fragment arrays (and, for ragged lengths, the
        bias array) through 32-bit buffer offsets; anything else runs on the exact fp32-operand kernel."""
        if self.arith != "f16x3":
            return False
        shape_ok = (self.c_hidden == 256 and self.no_qk_points == 8 and self.no_v_points == 12 and self.c_z // 4 == 32
                    and self.no_heads == 8)
        rows_pad = (n_rows // max(n_res, 1)) * ops.padded_len(n_res)
        return shape_ok and rows_pad * 12288 < (1 << 32) and (n_res % 32 == 0 or n_rows * n_res * 32 < (1 << 31))

    def attention_f16(self, s_xp, B: int, N: int, r7, mask, pair_proj):
        """Projections -> points -> attention core on pre-split f16 operands.  s_xp: packed planes of s [B*N, c_s].
        -> packed planes of linear_out's input [B*N, H*(c_hidden + 4 Pv + c_z/4)] -- of the FOLDED linear_out (node_packs()["outf"])
        when ``self.folded`` (the default: _folded_packs), of linear_out itself otherwise."""
        w, d, M, H = self.node_packs(), self._derived(), B * N, self.no_heads
        if self.folded:
            return self._attention_f16_folded(s_xp, B, N, r7, mask, pair_proj, w, d)
        if w["v"]["tg"] != 8:
            raise ops.HipLibraryError("s2s_node_linear_vfrag is instantiated for 8 tiles per column block")
        NP = ops.padded_len(N)
        # a ragged length: the q / k / v operands are produced straight into the per-sample padded row layout of the kernel
        rmap, Mo = ((NP, N), B * NP) if NP != N else (None, M)
        K = torch.ops.str2str_amd     # the kernels' operator surface (ops.register_torch_ops)
        # the five projections read the same planes and nothing of each other: ONE launch (s2s_node_linear_multi)
        names = ("q", "k", "v", "qp", "kvp")
        var = {n: (("w", w[n]["tg"]) if n == "v" else ops.small_rows_variant(w[n], M)) for n in names}   # (v: A fragments, 8 tiles per head)
        dims = [x for n in names for x in (w[n]["k"], w[n]["n"], var[n][1])]
        q_xp, k_xp, v_vf, qp, kvp = K.ipa_projections(s_xp, *[[w[n][var[n][0]], w[n]["b"]] for n in names], dims, M, Mo, *(rmap or (0, 0)))
        pts = K.ipa_prep_points_f16(r7.view(B, N, 7), qp, kvp, d["hw"], H, self.no_qk_points, self.no_v_points, self.c_hidden)
        attn_bias, pair_z = pair_proj
        feats, feats_xp = K.ipa_attention_f16w(q_xp, k_xp, v_vf, *pts, attn_bias, pair_z, mask, r7, H, self.c_hidden,
                                               self.no_qk_points, self.no_v_points, self.c_z // 4, self.inf, self.eps, True)
        c0 = H * self.c_hidden
        f2 = feats.view(M, -1)
        K.pack_planes(f2, c0, f2.shape[1] - c0, feats_xp, f2.shape[1], c0)
        return feats_xp

    def _attention_f16_folded(self, s_xp, B, N, r7, mask, pair_proj, w, d):
        M, H = B * N, self.no_heads
        NP = ops.padded_len(N)
        rmap, Mo = ((NP, N), B * NP) if NP != N else (None, M)
        K = torch.ops.str2str_amd
        names = ("qf", "qp", "kvp")
        var = {n: ops.small_rows_variant(w[n], M) for n in names}
        dims = [x for n in ("qf", None, None, "qp", "kvp") for x in ((w[n]["k"], w[n]["n"], var[n][1]) if n else (0, 0, 0))]
        q_xp, _, _, qp, kvp = K.ipa_projections(s_xp, [w["qf"][var["qf"][0]], w["qf"]["b"]], [], [], [w["qp"][var["qp"][0]], w["qp"]["b"]],
                                                [w["kvp"][var["kvp"][0]], w["kvp"]["b"]], dims, M, Mo, *(rmap or (0, 0)))
        *pts, k_sh, v_sh = K.ipa_prep_points_shared_kv(r7.view(B, N, 7), qp, kvp, d["hw"], s_xp, H, self.no_qk_points, self.no_v_points,
                                                       self.c_hidden)
        attn_bias, pair_z = pair_proj
        feats, feats_xp = K.ipa_attention_f16w(q_xp, s_xp if k_sh is None else k_sh, v_sh, *pts, attn_bias, pair_z, mask, r7, H,
                                               self.c_hidden, self.no_qk_points, self.no_v_points, self.c_z // 4, self.inf, self.eps, True)
        c0 = H * self.c_hidden
        f2 = feats.view(M, -1)
        K.pack_planes(f2, c0, f2.shape[1] - c0, feats_xp, f2.shape[1], c0)
        return feats_xp

    @property
    def folded(self) -> bool:
        """Is the f16 attention path the folded one (K = V = s)?  Default on (S2S_IPA_FOLD=0 or ``fold = False`` restore the per-head
        k / v projections: same kernels, the reference's grouping of the weights)."""
        return self.fold and self.c_s == self.c_hidden and self.no_heads == 8

    def out_pack(self, feats_a) -> dict:
        """linear_out's packed layer for the features ``attention`` returned (folded for the folded f16 path)."""
        w = self.node_packs()
        return w["outf"] if (self.folded and self._last_folded) else w["out"]

    def attention_f32(self, s_act, B: int, N: int, r7, mask, pair_proj):
        """The same on the exact fp32-operand kernel (s2s_ipa_attention: any length, any magnitude, any input arithmetic)
        -> linear_out's input in the node stream's activation format (packed planes for a planes input, fp32 otherwise)."""
        w, d, M, H = self.node_packs(), self._derived(), B * N, self.no_heads
        lin = lambda x: ops.node_apply(s_act, x, M)[0]  # noqa: E731
        q, kv, qp, kvp = lin(w["q"]), lin(w["kv"]), lin(w["qp"]), lin(w["kvp"])
        q_pts, k_pts, v_pts = torch.ops.str2str_amd.ipa_prep_points(r7.view(B, N, 7), qp.view(B, N, -1), kvp.view(B, N, -1), H,
                                                                    self.no_qk_points, self.no_v_points)
        attn_bias, pair_z = pair_proj
        feats = ops.ipa_attention(q.view(B, N, H, -1), kv.view(B, N, H, -1), q_pts, k_pts, v_pts, attn_bias, pair_z, mask, r7,
                                  d["hw"], H, self.c_hidden, self.no_qk_points, self.no_v_points, self.c_z // 4, self.inf,
                                  self.eps, logits_inplace=True)
        return ops.pack_planes(feats.view(M, -1)) if s_act.dtype == torch.int16 else feats.view(M, -1)

    def attention(self, s_act, B: int, N: int, r7, mask, pair_proj):
        """Attention core of the block on the kernel of the configured arithmetic (see the module docstring).  The features go to
        ``out_pack()`` -- the folded linear_out after the folded f16 path."""
        self._last_folded = False
        if s_act.dtype == torch.int16 and self.use_f16(N, B * N):
            self._last_folded = self.folded
            return self.attention_f16(s_act, B, N, r7, mask, pair_proj)
        return self.attention_f32(s_act, B, N, r7, mask, pair_proj)

    def pair_proj_weights(self):
        """What a pair-stream producer needs to emit this block's attention bias / pair_z in its own epilogue: the packed
        [linear_b; down_z] matrix as fp32 ("wp") and as one f16x3 weight stage ("wp_f16x2"), and its bias ("b64")."""
        return self._derived()

    def forward(self, s: torch.Tensor, z: torch.Tensor, r, mask: torch.Tensor, _rigids7: Optional[torch.Tensor] = None,
                _pair_proj=None):
        """s [B,N,c_s], z [B,N,N,c_z], r Rigid [B,N] (translations already scaled), mask [B,N]
        -> [B,N,c_s] (reference :100-268).  ``_rigids7`` lets the trunk pass its frame tensor directly;
        ``_pair_proj`` = (attn_bias, pair_z) already produced by the kernel that wrote z."""
        if not s.is_cuda:
            raise ops.HipLibraryError("InvariantPointAttention runs on the HIP device only (no CPU fallback)")
        if self.no_heads != 8 or self.c_z != 128:
            raise ops.HipLibraryError("IPA kernels are built for the reference configuration (H=8, c_z=128)")
        d = self._derived()
        r7 = (_rigids7 if _rigids7 is not None else r.to_tensor_7()).type(torch.float32).contiguous()
        mask = mask.type(torch.float32).contiguous()
        B, N = s.shape[:2]
        pp = _pair_proj if _pair_proj is not None else ops.pair_project(z.contiguous(), d["wp"], d["b64"])
        feats_a = self.attention(ops.to_act(s.reshape(B * N, -1).float().contiguous(), self.arith), B, N, r7, mask, pp)
        out, _ = ops.node_apply(feats_a, self.out_pack(feats_a), B * N)
        return out.view(B, N, -1)


class TranslationIPA(nn.Module):
    def __init__(self, c_s: int, c_z: int, coordinate_scaling: float, no_ipa_blocks: int, skip_embed_size: int,
                 transformer_num_heads: int = 4, transformer_num_layers: int = 2, c_hidden: int = 256, no_heads: int = 8,
                 no_qk_points: int = 8, no_v_points: int = 12, dropout: float = 0.0):
        super().__init__()
        self.coordinate_scaling = coordinate_scaling
        self.scale_pos = lambda x: x * coordinate_scaling
        self.scale_rigids = lambda x: x.apply_trans_fn(self.scale_pos)
        self.unscale_pos = lambda x: x / coordinate_scaling
        self.unscale_rigids = lambda x: x.apply_trans_fn(self.unscale_pos)
        self.trunk = nn.ModuleDict()
        self.num_blocks = no_ipa_blocks
        for b in range(no_ipa_blocks):
            self.trunk[f"ipa_{b}"] = InvariantPointAttention(c_s=c_s, c_z=c_z, c_hidden=c_hidden, no_heads=no_heads,
                                                            no_qk_points=no_qk_points, no_v_points=no_v_points)
            self.trunk[f"ipa_ln_{b}"] = nn.LayerNorm(c_s)
            self.trunk[f"skip_embed_{b}"] = Linear(c_s, skip_embed_size, init="final")
            d = c_s + skip_embed_size
            layer = nn.TransformerEncoderLayer(d_model=d, nhead=transformer_num_heads, dim_feedforward=d)
            self.trunk[f"transformer_{b}"] = nn.TransformerEncoder(layer, transformer_num_layers, enable_nested_tensor=False)
            self.trunk[f"linear_{b}"] = Linear(d, c_s, init="final")
            self.trunk[f"node_transition_{b}"] = NodeTransition(c_s)
            self.trunk[f"bb_update_{b}"] = BackboneUpdate(c_s)
            if b < no_ipa_blocks - 1:
                self.trunk[f"edge_transition_{b}"] = EdgeTransition(node_embed_size=c_s, edge_embed_in=c_z,
                                                                    edge_embed_out=c_z)
        self.torsion_pred = TorsionAngleHead(c_s, 1)
        self._wcache = ParamCache()
        # Padding semantics of the encoder layers.  False = the reference's: the FLOAT key-padding mask (1 - node_mask) is ADDED to
        # the logits, as PyTorch does for float masks -- a no-op for the all-ones masks of every reference
        # run.  True (set by the mixed-length sampler) removes padded keys (-inf), which is what a padded batch needs to
        # reproduce each chain's un-padded run.
        self.exact_padding = False
        self.fuse_pair_projection = True  # producers of z also emit the next IPA block's linear_b / down_z
        self.arith = default_arith()      # node stream: "f16x3" (packed f16 planes) | "f32" (fp32 row-major), str2str_amd/arith.py

    # ------------------------------------------------------------------ packed weights of the fused node path
    def _node_weights(self):
        def build():
            T = self.trunk
            pk = lambda lin, whole=False: ops.pack_node_layer(lin.weight, lin.bias, whole)  # noqa: E731
            out = {}
            for b in range(self.num_blocks):
                d = {"skip": pk(T[f"skip_embed_{b}"]), "lin": pk(T[f"linear_{b}"]), "bb": pk(T[f"bb_update_{b}"].linear)}
                nt = T[f"node_transition_{b}"]
                # (whole = the layer's epilogue normalises over the row: only those are tied to one column block at small row counts)
                d["nt1"], d["nt2"], d["nt3"] = pk(nt.linear_1), pk(nt.linear_2), pk(nt.linear_3, True)
                d["layers"] = []
                for layer in T[f"transformer_{b}"].layers:
                    att = layer.self_attn
                    d["layers"].append({"in": ops.pack_node_layer(att.in_proj_weight, att.in_proj_bias), "o": pk(att.out_proj, True),
                                        "l1": pk(layer.linear1), "l2": pk(layer.linear2, True)})
                out[b] = d
            tp = self.torsion_pred
            out["tor"] = {"l1": pk(tp.linear_1), "l2": pk(tp.linear_2), "fin": pk(tp.linear_final)}
            return out

        return self._wcache.get(list(self.parameters()), build)

    def _mask_terms(self, residue_mask: torch.Tensor, fixed_mask: torch.Tensor):
        """(node mask, diffuse mask = (1 - fixed) * node mask, the encoder layers' key bias, fixed mask flat), all float32: constants of a
        chunk, cached on the two mask tensors (five tiny launches per evaluation otherwise)."""
        key = (residue_mask.data_ptr(), residue_mask._version, tuple(residue_mask.shape), residue_mask.dtype, fixed_mask.data_ptr(),
               fixed_mask._version, fixed_mask.dtype, bool(self.exact_padding))
        if key != getattr(self, "_mk_key", None) or self._mk_src[0] is not residue_mask or self._mk_src[1] is not fixed_mask:
            node_mask = residue_mask.type(torch.float).contiguous()
            fixed = fixed_mask.type(torch.float).contiguous()
            diffuse_mask = ((1 - fixed) * node_mask).contiguous()
            pad = 1.0 - node_mask
            # float key-padding mask of the encoder layers: ADDED to the logits (PyTorch semantics, a no-op for all-ones masks);
            # exact-padding mode removes padded keys instead
            key_bias = (torch.where(pad > 0, float("-inf"), 0.0) if self.exact_padding else pad).float().contiguous()
            self._mk_val = (node_mask, diffuse_mask, key_bias, fixed.reshape(-1))
            self._mk_key, self._mk_src = key, (residue_mask, fixed_mask)
        return self._mk_val

    def forward(self, node_embed: torch.Tensor, edge_embed: torch.Tensor, batch: dict, _first_proj=None) -> dict:
        """reference :331-387.  Frames travel as one [B,N,7] tensor between the fused kernels; node activations as packed f16
        planes (arith "f16x3") or fp32 row-major (arith "f32") -- ``ops.node_apply`` runs a layer in the arithmetic of its input."""
        if not node_embed.is_cuda:
            raise ops.HipLibraryError("TranslationIPA runs on the HIP device only (no CPU fallback)")
        T, W = self.trunk, self._node_weights()
        f16 = self.arith == "f16x3"
        B, N, C = node_embed.shape
        M = B * N
        dev = node_embed.device
        node_mask, diffuse_mask, key_bias, fixed_flat = self._mask_terms(batch["residue_mask"], batch["fixed_mask"])
        nm, dm = node_mask.reshape(M), diffuse_mask.reshape(M)
        init7 = batch["rigids_t"].type(torch.float).contiguous()
        curr7 = torch.ops.str2str_amd.rigid_scale_trans(init7, self.coordinate_scaling, False)
        proj = _first_proj

        def lin(x, w, **kw):
            return ops.node_apply(x, w, M, **kw)

        s_f32 = node_embed.reshape(M, C).float().contiguous()
        init_a = batch.get("_node_embed_act")     # skip_embed reads the embedder's output in every block
        if init_a is None or (init_a.dtype == torch.int16) != f16:
            init_a = ops.to_act(s_f32, self.arith)
        s_a = init_a
        D = C + T["skip_embed_0"].out_features     # transformer width (320)
        # every block's skip_embed reads the embedder's output: the four layers are ONE launch, each into its block's buffer
        xbuf = []
        for b in range(self.num_blocks):
            xf = torch.empty(M, D, device=dev, dtype=torch.float32)
            xbuf.append((xf, ops.xp_alloc(M, D, dev) if f16 else xf))
        skip_done = None
        if f16 and 2 <= self.num_blocks <= 6:
            skip_done = ops.node_apply_multi(init_a, [(W[b]["skip"], dict(out_f32=xbuf[b][0], out_col0=C, out_xp=xbuf[b][1], out_xp_k=D, out_xp_k0=C))
                                                      for b in range(self.num_blocks)], M)
        for b in range(self.num_blocks):
            w, ipa = W[b], T[f"ipa_{b}"]
            # ---- InvariantPointAttention (:100-268): projections -> points -> attention core -> linear_out (+mask, +residual, LN)
            if proj is None:
                d = ipa._derived()
                if isinstance(edge_embed, ops.PairTiled):
                    edge_embed = ops.pair_untiled(edge_embed)
                proj = torch.ops.str2str_amd.pair_project(edge_embed.contiguous(), d["wp"], d["b64"])
            feats_a = ipa.attention(s_a, B, N, curr7, node_mask, tuple(proj))
            proj = None
            ln = T[f"ipa_ln_{b}"]
            x_f32, x_a = xbuf[b]                                           # [node_embed | skip_embed(init)] (:356)
            lin(feats_a, ipa.out_pack(feats_a), pre_mask=nm, residual=s_f32, ln=(ln.weight, ln.bias, ln.eps), out_f32=x_f32,
                out_xp=x_a, out_xp_k=D)
            if skip_done is None:
                lin(init_a, w["skip"], out_f32=x_f32, out_col0=C, out_xp=x_a, out_xp_k=D, out_xp_k0=C)
            # ---- 2 x post-norm TransformerEncoderLayer (:312-317,357)
            xf, xx = x_f32, x_a
            for layer, lw in zip(T[f"transformer_{b}"].layers, w["layers"]):
                qkv, _ = lin(xx, lw["in"])
                sa_f32, sa_xp = torch.ops.str2str_amd.encoder_attention(qkv, key_bias, B, N, layer.self_attn.num_heads, not f16, f16, self.arith)
                # the post-attention half of the layer as ONE launch (s2s_node_chain): out_proj + residual + norm1 (its fp32 result is
                # stored: the residual of norm2), linear1 -> relu -> linear2 + that residual + norm2; the hidden activations stay in
                # registers.  (S2S_ENC_CHAIN3=0: out_proj on its own, then the two feed-forward layers as a chain -- the same bits)
                if _ENC_CHAIN3:
                    x1 = torch.empty(M, D, device=dev, dtype=torch.float32)
                    xf, xx = ops.node_apply_chain(sa_xp if f16 else sa_f32, [lw["o"], lw["l1"], lw["l2"]], M, (False, True, False),
                                                  first_residual=xf, first_out_f32=x1, first_ln=(layer.norm1.weight, layer.norm1.bias, layer.norm1.eps),
                                                  residual=x1, ln=(layer.norm2.weight, layer.norm2.bias, layer.norm2.eps), want_xp=True)
                else:
                    x1, x1a = lin(sa_xp if f16 else sa_f32, lw["o"], residual=xf, ln=(layer.norm1.weight, layer.norm1.bias, layer.norm1.eps),
                                  want_xp=True)
                    xf, xx = ops.node_apply_chain(x1a, [lw["l1"], lw["l2"]], M, (True, False), residual=x1,
                                                  ln=(layer.norm2.weight, layer.norm2.bias, layer.norm2.eps), want_xp=True)
            # ---- node_embed + linear(tr) (:358), NodeTransition (:359, layers.py:128-145), mask (:360)
            # ... as ONE launch: trunk.linear's fp32 result is stored and read back as NodeTransition's residual (s2s_node_chain)
            nt = T[f"node_transition_{b}"]
            if _CHAIN4:
                n_f32 = torch.empty(M, C, device=dev, dtype=torch.float32)
                s_f32, s_a = ops.node_apply_chain(xx, [w["lin"], w["nt1"], w["nt2"], w["nt3"]], M, (False, True, True, False),
                                                  first_residual=x_f32, first_out_f32=n_f32, residual=n_f32,
                                                  ln=(nt.ln.weight, nt.ln.bias, nt.ln.eps), post_mask=nm, want_xp=True)
            else:
                n_f32, n_a = lin(xx, w["lin"], residual=x_f32, want_xp=True)
                s_f32, s_a = ops.node_apply_chain(n_a, [w["nt1"], w["nt2"], w["nt3"]], M, (True, True, False), residual=n_f32,
                                                  ln=(nt.ln.weight, nt.ln.bias, nt.ln.eps), post_mask=nm, want_xp=True)
            # ---- backbone update (:361-365) and the layers that read the same s: the EdgeTransition's per-node parts (:367-372; their
            #      pair MLP runs in its own kernel below), after the last block the torsion head's first layer -- ONE launch
            has_et = b < self.num_blocks - 1
            et = T[f"edge_transition_{b}"] if has_et else None
            specs = [(w["bb"], dict(pre_scale=dm))]
            if has_et:
                nl = et.node_layers()
                if et.arith == "f16x3":   # node_ab in the pair kernel's form: the column half x 2^5, as a layer of its own into the same buffer
                    node_ab16, ab_specs = et.ab16_specs(nl, True, M, s_a.device)
                    specs += [(nl["init"], {})] + ab_specs
                else:
                    specs += [(nl["init"], {}), (nl["ab_s"], {})]
            else:
                specs += [(W["tor"]["l1"], dict(relu=True, want_f32=False, want_xp=True))]
            outs = ops.node_apply_multi(s_a, specs, M)
            upd = outs[0][0]
            curr7 = torch.ops.str2str_amd.rigid_compose_update(curr7, upd, diffuse_mask)   # (the padded [M, 32] output in place: no copy)
            if has_et:
                n_p, node_ab = outs[1][0], (node_ab16 if et.arith == "f16x3" else outs[2][0])
                nxt = T[f"ipa_{b + 1}"].pair_proj_weights() if self.fuse_pair_projection else None
                # f16x3 with fused projections: the pair tensor stays in the kernels' tiled layout, and the last EdgeTransition's
                # output (read by nothing but the projections it already carries) is not written
                # (each EdgeTransition in ITS arithmetic: the layouts change where an f16x3 kernel meets an exact one)
                et16 = et.arith == "f16x3"
                last = b == self.num_blocks - 2
                nxt16 = (not last) and T[f"edge_transition_{b + 1}"].arith == "f16x3"
                lay = "rowmajor" if not (et16 and nxt is not None) else ("none" if last else ("tiled" if nxt16 else "rowmajor"))
                if not et16 and isinstance(edge_embed, ops.PairTiled):
                    edge_embed = ops.pair_untiled(edge_embed)
                res = et.pair_mlp(edge_embed, node_ab.view(B, N, -1), n_p.view(B, N, -1), node_mask, nxt,
                                  **({"out_layout": lay} if lay != "rowmajor" else {}), **({"ab_kernel_form": True} if et16 else {}))
                if nxt is not None:
                    edge_embed, *proj = res
                else:
                    edge_embed = res
        wt = W["tor"]
        t1 = outs[1][1]
        _, t2 = lin(t1, wt["l2"], residual=s_f32, want_f32=False, want_xp=True)
        # u / sqrt(max(sum u^2, eps)) (layers.py:199-213) straight from the head's padded output: one launch instead of six tiny ones
        # ... and DenoisingNet's blend with the input torsion under the fixed mask (denoising_ipa.py:192-193) in the same launch
        gt = batch.get("torsion_angles_sin_cos")
        blend = gt is not None and gt.is_cuda and gt.dtype == torch.float32 and gt.ndim == 4
        psi = torch.ops.str2str_amd.torsion_head(lin(t2, wt["fin"])[0], M, True, self.torsion_pred.eps,
                                                 gt[..., 2, :] if blend else None, fixed_flat if blend else None).view(B, N, 2)
        out7 = torch.ops.str2str_amd.rigid_scale_trans(curr7, self.coordinate_scaling, True)
        return {
            "psi_blended": blend,
            "in_rigids": Rigid.from_tensor_7(init7),
            "out_rigids": Rigid(Rotation(quats=out7[..., :4], normalize_quats=False), out7[..., 4:]),
            "out_rigids7": out7,
            "psi": psi,
        }
