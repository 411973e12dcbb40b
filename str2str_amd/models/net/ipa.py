"""Invariant Point Attention and the ``TranslationIPA`` trunk on the HIP kernels.

Interface, constructor arguments and ``state_dict`` keys follow the reference's
``src/models/net/ipa.py`` (InvariantPointAttention :34-268, TranslationIPA :271-387).  Point frames,
pair projections (linear_b / down_z), logits, softmax, o / o_pt / o_pair are the HIP launches
``s2s_ipa_prep_points_f16``, ``s2s_pair_project`` (normally fused into the producer of z), ``s2s_ipa_attention_f16w`` and
``s2s_ipa_opair`` -- for every chain length (operands padded per sample to the kernel's 32-residue tiles).  ``S2S_IPA_PATH``
(read once, at construction) selects the alternatives: ``planes`` = the range-safe three-way bf16 operand kernel (lengths that are
multiples of 32; others fall through to ``f32``), ``f32`` = the exact fp32-operand kernel (``s2s_ipa_attention``, any length).
Every dense layer of the node stream (q / kv / point projections, linear_out, skip_embed, the transformer's
projections and feed-forward, trunk.linear, NodeTransition, BackboneUpdate, the per-node parts of EdgeTransition, the
torsion head) runs on ``s2s_node_linear`` (csrc/node_gemm.hip: split-f16 MFMA, activations travelling as packed f16 pair
planes) with bias / ReLU / mask / residual / LayerNorm fused into its epilogue -- ``TranslationIPA.forward``.
No BLAS / SDPA call is left in this module (the round-2 layer-by-layer A/B path was removed; tools/node_gemm_bench.py compares
the GEMM kernel with rocBLAS directly).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...common.rigid_utils import Rigid, Rotation
from .layers import BackboneUpdate, EdgeTransition, Linear, NodeTransition, ParamCache, TorsionAngleHead


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
        self.ipa_path = os.environ.get("S2S_IPA_PATH", "f16")   # f16 (default) | planes | f32, fixed at construction
        if self.ipa_path not in ("f16", "planes", "f32"):
            raise ValueError(f"S2S_IPA_PATH={self.ipa_path!r}: expected f16, planes or f32")
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
            return {"wp": ops.pack_weight(wcat), "b64": b64.contiguous(), "hw": hw.contiguous(),
                    "wp_bf16x3": ops.pack_bf16x3_layer(wcat64, "chain").reshape(-1).view(torch.int16).contiguous(),
                    "wp_f16x2": ops.pack_f16x2_layer(wcat64, "chain").reshape(-1).view(torch.int16).contiguous()}

        return self._cache.get([self.linear_b.weight, self.linear_b.bias, self.down_z.weight, self.down_z.bias,
                                self.head_weights], build)

    def node_packs(self):
        """Packed bf16x3 weights of this block's node projections for s2s_node_linear (k and v rows of linear_kv separately:
        the planes kernel consumes k as packed planes and v as A fragments)."""
        def build():
            H, C = self.no_heads, self.c_hidden
            pk = TranslationIPA._pack
            wkv = self.linear_kv.weight.view(H, 2, C, -1)
            bkv = self.linear_kv.bias.view(H, 2, C)
            d = {"q": pk(self.linear_q.weight, self.linear_q.bias), "kv": pk(self.linear_kv.weight, self.linear_kv.bias),
                 "k": pk(wkv[:, 0].reshape(H * C, -1), bkv[:, 0].reshape(-1)),
                 "v": pk(wkv[:, 1].reshape(H * C, -1), bkv[:, 1].reshape(-1)),
                 "qp": pk(self.linear_q_points.weight, self.linear_q_points.bias),
                 "kvp": pk(self.linear_kv_points.weight, self.linear_kv_points.bias),
                 "out": pk(self.linear_out.weight, self.linear_out.bias, True)}
            if d["v"]["tg"] != 8:
                raise ops.HipLibraryError("s2s_node_linear_vfrag is instantiated for 8 tiles per column block")
            return d

        return self._packs.get([p for lin in (self.linear_q, self.linear_kv, self.linear_q_points, self.linear_kv_points,
                                              self.linear_out) for p in (lin.weight, lin.bias)], build)

    def use_planes(self, n_res: int, n_rows: int = 0) -> bool:
        """Does the pre-split operand path serve this call?  The kernels are built for the reference configuration (c_hidden 256,
        8 / 12 points, c_z/4 = 32, 8 heads) and address their fragment arrays through 32-bit buffer offsets (< 4 GiB); the f16
        kernel takes any length, the bf16 planes kernel multiples of 32."""
        if self.ipa_path == "f32" or (self.ipa_path == "planes" and n_res % 32):
            return False
        shape_ok = (self.c_hidden == 256 and self.no_qk_points == 8 and self.no_v_points == 12 and self.c_z // 4 == 32
                    and self.no_heads == 8)
        rows_pad = (n_rows // max(n_res, 1)) * ops.padded_len(n_res)
        return shape_ok and rows_pad * 12288 < (1 << 32) and (n_res % 32 == 0 or n_rows * n_res * 32 < (1 << 31))

    def attention_planes(self, s_xp, B: int, N: int, r7, mask, pair_proj):
        """Projections -> points -> attention core on pre-split operands.  s_xp: packed planes of s [B*N, c_s].
        -> packed planes of linear_out's input [B*N, H*(c_hidden + 4 Pv + c_z/4)]"""
        w, d, M, H = self.node_packs(), self._derived(), B * N, self.no_heads
        f16 = self.ipa_path == "f16"
        NP = ops.padded_len(N)
        # a ragged length: the q / k / v operands are produced straight into the per-sample padded row layout of the kernel
        rmap, Mo = ((NP, N), B * NP) if NP != N else (None, M)
        lin = lambda x, **kw: ops.node_linear(s_xp, x["w"], x["b"], Mo, x["k"], x["n"], x["tg"], row_map=rmap, **kw)  # noqa: E731
        # attention operands: f16 pair planes (s2s_ipa_attention_f16w, three products per block) or exact three-way bf16 planes
        # (S2S_IPA_PATH=planes: s2s_ipa_attention_planes, six products)
        fmt = 0 if f16 else 1
        _, q_xp = lin(w["q"], want_f32=False, want_xp=True, xp_format=fmt)
        _, k_xp = lin(w["k"], want_f32=False, want_xp=True, xp_format=fmt)
        v_vf = ops.node_linear_vfrag(s_xp, w["v"]["w"], w["v"]["b"], Mo, w["v"]["k"], w["v"]["n"], self.c_hidden // 32, f16=f16,
                                     row_map=rmap)
        qp, _ = ops.node_linear(s_xp, w["qp"]["w"], w["qp"]["b"], M, w["qp"]["k"], w["qp"]["n"], w["qp"]["tg"])
        kvp, _ = ops.node_linear(s_xp, w["kvp"]["w"], w["kvp"]["b"], M, w["kvp"]["k"], w["kvp"]["n"], w["kvp"]["tg"])
        pts = ops.ipa_prep_points_planes(r7.view(B, N, 7), qp, kvp, d["hw"], H, self.no_qk_points, self.no_v_points, self.c_hidden, f16=f16)
        attn_bias, pair_z = pair_proj
        feats, feats_xp = ops.ipa_attention_planes(q_xp, k_xp, v_vf, pts, attn_bias, pair_z, mask, r7, H, self.c_hidden,
                                                   self.no_qk_points, self.no_v_points, self.c_z // 4, self.inf, self.eps,
                                                   logits_inplace=True, f16=f16)
        c0 = H * self.c_hidden
        f2 = feats.view(M, -1)
        ops.pack_planes(f2, col0=c0, n_cols=f2.shape[1] - c0, out=feats_xp, out_k=f2.shape[1], k0=c0)
        return feats_xp

    def attention_f32(self, s_xp, B: int, N: int, r7, mask, pair_proj):
        """The same on the exact fp32-operand kernel (s2s_ipa_attention: any length, any magnitude) -> packed planes of
        linear_out's input."""
        w, d, M, H = self.node_packs(), self._derived(), B * N, self.no_heads
        lin = lambda x: ops.node_linear(s_xp, x["w"], x["b"], M, x["k"], x["n"], x["tg"])[0]  # noqa: E731
        q, kv, qp, kvp = lin(w["q"]), lin(w["kv"]), lin(w["qp"]), lin(w["kvp"])
        q_pts, k_pts, v_pts = ops.ipa_prep_points(r7.view(B, N, 7), qp.view(B, N, -1), kvp.view(B, N, -1), H, self.no_qk_points,
                                                  self.no_v_points)
        attn_bias, pair_z = pair_proj
        feats = ops.ipa_attention(q.view(B, N, H, -1), kv.view(B, N, H, -1), q_pts, k_pts, v_pts, attn_bias, pair_z, mask, r7,
                                  d["hw"], H, self.c_hidden, self.no_qk_points, self.no_v_points, self.c_z // 4, self.inf,
                                  self.eps, logits_inplace=True)
        return ops.pack_planes(feats.view(M, -1))

    def attention(self, s_xp, B: int, N: int, r7, mask, pair_proj):
        """Attention core of the block on the configured kernel (see the module docstring)."""
        if self.use_planes(N, B * N):
            return self.attention_planes(s_xp, B, N, r7, mask, pair_proj)
        return self.attention_f32(s_xp, B, N, r7, mask, pair_proj)

    def pair_proj_weights(self):
        """(packed [linear_b; down_z] weight, bias64, the same matrix as one bf16x3 weight stage): what a pair-stream
        producer needs to emit this block's attention bias / pair_z in its own epilogue."""
        d = self._derived()
        return d["wp"], d["b64"], d["wp_bf16x3"], d["wp_f16x2"]

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
        feats_xp = self.attention(ops.pack_planes(s.reshape(B * N, -1).float().contiguous()), B, N, r7, mask, pp)
        w = self.node_packs()["out"]
        out, _ = ops.node_linear(feats_xp, w["w"], w["b"], B * N, w["k"], w["n"], w["tg"])
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

    # ------------------------------------------------------------------ packed weights of the fused node path
    def _node_weights(self):
        def build():
            T = self.trunk
            pk = lambda lin, whole=False: self._pack(lin.weight, lin.bias, whole)  # noqa: E731
            out = {}
            for b in range(self.num_blocks):
                ipa = T[f"ipa_{b}"]
                d = dict(ipa.node_packs())
                d.update({"skip": pk(T[f"skip_embed_{b}"]), "lin": pk(T[f"linear_{b}"], True),
                          "bb": pk(T[f"bb_update_{b}"].linear)})
                nt = T[f"node_transition_{b}"]
                d["nt1"], d["nt2"], d["nt3"] = pk(nt.linear_1, True), pk(nt.linear_2, True), pk(nt.linear_3, True)
                d["layers"] = []
                for layer in T[f"transformer_{b}"].layers:
                    att = layer.self_attn
                    d["layers"].append({"in": self._pack(att.in_proj_weight, att.in_proj_bias), "o": pk(att.out_proj, True),
                                        "l1": pk(layer.linear1, True), "l2": pk(layer.linear2, True)})
                if b < self.num_blocks - 1:
                    et = T[f"edge_transition_{b}"]
                    ep = et._packed()
                    d["et_init"] = pk(et.initial_embed)
                    d["et_ab"] = self._pack(ep["w_ab"], ep["b_ab"])
                out[b] = d
            tp = self.torsion_pred
            out["tor"] = {"l1": pk(tp.linear_1, True), "l2": pk(tp.linear_2, True), "fin": pk(tp.linear_final)}
            return out

        return self._wcache.get(list(self.parameters()), build)

    @staticmethod
    def _pack(w, bias, whole_row=False):
        n_out, k = w.shape
        n_pad = -(-n_out // 32) * 32
        tg = ops.node_tiles(n_pad, whole_row=whole_row)
        b = w.new_zeros(n_pad, dtype=torch.float32)
        if bias is not None:
            b[:n_out] = bias.float()
        return {"w": ops.pack_node_weight(w.float(), tg), "b": b.contiguous(), "n": n_pad, "k": k, "tg": tg}

    def forward(self, node_embed: torch.Tensor, edge_embed: torch.Tensor, batch: dict, _first_proj=None) -> dict:
        """reference :331-387.  Frames travel as one [B,N,7] tensor between the fused kernels."""
        if not node_embed.is_cuda:
            raise ops.HipLibraryError("TranslationIPA runs on the HIP device only (no CPU fallback)")
        T, W = self.trunk, self._node_weights()
        B, N, C = node_embed.shape
        M = B * N
        dev = node_embed.device
        node_mask = batch["residue_mask"].type(torch.float).contiguous()
        diffuse_mask = ((1 - batch["fixed_mask"].type(torch.float)) * node_mask).contiguous()
        nm, dm = node_mask.reshape(M), diffuse_mask.reshape(M)
        init7 = batch["rigids_t"].type(torch.float).contiguous()
        curr7 = ops.rigid_scale_trans(init7, self.coordinate_scaling, divide=False)
        pad = 1.0 - node_mask
        # float key-padding mask of the encoder layers: ADDED to the logits (PyTorch semantics, a no-op for all-ones masks);
        # exact-padding mode removes padded keys instead
        key_bias = (torch.where(pad > 0, float("-inf"), 0.0) if self.exact_padding else pad).float().contiguous()
        proj = _first_proj

        def lin(xp, w, **kw):
            return ops.node_linear(xp, w["w"], w["b"], M, w["k"], w["n"], w["tg"], **kw)

        s_f32 = node_embed.reshape(M, C).float().contiguous()
        init_xp = batch.get("_node_embed_xp")     # skip_embed reads the embedder's output in every block
        if init_xp is None:
            init_xp = ops.pack_planes(s_f32)
        s_xp = init_xp
        D = C + T["skip_embed_0"].out_features     # transformer width (320)
        for b in range(self.num_blocks):
            w, ipa = W[b], T[f"ipa_{b}"]
            d = ipa._derived()
            # ---- InvariantPointAttention (:100-268): projections -> points -> attention core -> linear_out (+mask, +residual, LN)
            attn_bias, pair_z = proj if proj is not None else ops.pair_project(edge_embed.contiguous(), d["wp"], d["b64"])
            proj = None
            feats_xp = ipa.attention(s_xp, B, N, curr7, node_mask, (attn_bias, pair_z))
            ln = T[f"ipa_ln_{b}"]
            x_f32 = torch.empty(M, D, device=dev, dtype=torch.float32)     # [node_embed | skip_embed(init)] (:356)
            x_xp = ops.xp_alloc(M, D, dev)
            lin(feats_xp, w["out"], pre_mask=nm, residual=s_f32, ln=(ln.weight, ln.bias, ln.eps), out_f32=x_f32, out_xp=x_xp, out_xp_k=D)
            lin(init_xp, w["skip"], out_f32=x_f32, out_col0=C, out_xp=x_xp, out_xp_k=D, out_xp_k0=C)
            # ---- 2 x post-norm TransformerEncoderLayer (:312-317,357)
            xf, xx = x_f32, x_xp
            for layer, lw in zip(T[f"transformer_{b}"].layers, w["layers"]):
                qkv, _ = lin(xx, lw["in"])
                _, sa_xp = ops.encoder_attention(qkv, key_bias, B, N, layer.self_attn.num_heads)
                x1, x1x = lin(sa_xp, lw["o"], residual=xf, ln=(layer.norm1.weight, layer.norm1.bias, layer.norm1.eps), want_xp=True)
                _, hx = lin(x1x, lw["l1"], relu=True, want_f32=False, want_xp=True)
                xf, xx = lin(hx, lw["l2"], residual=x1, ln=(layer.norm2.weight, layer.norm2.bias, layer.norm2.eps), want_xp=True)
            # ---- node_embed + linear(tr) (:358), NodeTransition (:359, layers.py:128-145), mask (:360)
            n_f32, n_xp = lin(xx, w["lin"], residual=x_f32, want_xp=True)
            _, h1 = lin(n_xp, w["nt1"], relu=True, want_f32=False, want_xp=True)
            _, h2 = lin(h1, w["nt2"], relu=True, want_f32=False, want_xp=True)
            nt = T[f"node_transition_{b}"]
            s_f32, s_xp = lin(h2, w["nt3"], residual=n_f32, ln=(nt.ln.weight, nt.ln.bias, nt.ln.eps), post_mask=nm, want_xp=True)
            # ---- backbone update (:361-365)
            upd, _ = lin(s_xp, w["bb"], pre_scale=dm)
            curr7 = ops.rigid_compose_update(curr7, upd[:, :6].contiguous().view(B, N, 6), diffuse_mask)
            # ---- EdgeTransition (:367-372): per-node parts here, the pair MLP in its own kernel
            if b < self.num_blocks - 1:
                et = T[f"edge_transition_{b}"]
                n_p, n_pxp = lin(s_xp, w["et_init"], want_xp=True)
                node_ab, _ = lin(n_pxp, w["et_ab"])
                nxt = T[f"ipa_{b + 1}"].pair_proj_weights() if self.fuse_pair_projection else None
                res = et.pair_mlp(edge_embed, node_ab.view(B, N, -1), n_p.view(B, N, -1), node_mask, nxt)
                if nxt is not None:
                    edge_embed, *proj = res
                else:
                    edge_embed = res
        wt = W["tor"]
        _, t1 = lin(s_xp, wt["l1"], relu=True, want_f32=False, want_xp=True)
        _, t2 = lin(t1, wt["l2"], residual=s_f32, want_f32=False, want_xp=True)
        u = lin(t2, wt["fin"])[0][:, :2].reshape(B, N, 2)
        psi = u / torch.sqrt(torch.clamp(torch.sum(u**2, dim=-1, keepdim=True), min=self.torsion_pred.eps))
        out7 = ops.rigid_scale_trans(curr7, self.coordinate_scaling, divide=True)
        return {
            "in_rigids": Rigid.from_tensor_7(init7),
            "out_rigids": Rigid(Rotation(quats=out7[..., :4], normalize_quats=False), out7[..., 4:]),
            "out_rigids7": out7,
            "psi": psi,
        }
