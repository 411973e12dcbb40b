"""``EmbeddingModule`` and ``DenoisingNet`` (the SE(3) score network entry) on the HIP kernels.

Interface and parameter names follow the reference's ``src/models/net/denoising_ipa.py``
(get_positional_embedding :13-31, get_timestep_embedding :34-46, EmbeddingModule :49-159,
DenoisingNet :162-211).  The N x N edge embedding never builds the [B, N^2, 120] feature tensor:
its first Linear is a sum of four table rows (row part, column part, relative-position table,
distogram-bin table) gathered inside ``s2s_edge_embed`` (csrc/pair_mlp.hip), which then runs the two
128x128 layers on fp32 MFMA, LayerNorm and the edge mask and writes z once.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...arith import default_arith
from ...common.all_atom import compute_backbone
from ...common.rigid_utils import Rigid
from .ipa import TranslationIPA  # noqa: F401  (re-exported like the reference module)
from .layers import ParamCache


def get_positional_embedding(indices: torch.Tensor, embedding_dim: int, max_len: int = 2056) -> torch.Tensor:
    K = torch.arange(embedding_dim // 2, device=indices.device)
    arg = indices[..., None] * math.pi / (max_len ** (2 * K[None] / embedding_dim))
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, max_len: int = 10000) -> torch.Tensor:
    assert timesteps.ndim == 1
    timesteps = timesteps * max_len
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float, device=timesteps.device) * -(math.log(max_len) / (half - 1)))
    emb = timesteps.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1), mode="constant")
    return emb


class EmbeddingModule(nn.Module):
    def __init__(self, init_embed_size: int, node_embed_size: int, edge_embed_size: int, num_bins: int = 22,
                 min_bin: float = 1e-5, max_bin: float = 20.0, self_conditioning: bool = True):
        super().__init__()
        pos_embed_size = t_embed_size = init_embed_size
        node_in = t_embed_size + 1 + pos_embed_size
        edge_in = (t_embed_size + 1) * 2 + pos_embed_size
        self.node_embed = nn.Sequential(
            nn.Linear(node_in, node_embed_size), nn.ReLU(), nn.Linear(node_embed_size, node_embed_size), nn.ReLU(),
            nn.Linear(node_embed_size, node_embed_size), nn.LayerNorm(node_embed_size),
        )
        self.self_conditioning = self_conditioning
        if self_conditioning:
            edge_in += num_bins
        self.edge_embed = nn.Sequential(
            nn.Linear(edge_in, edge_embed_size), nn.ReLU(), nn.Linear(edge_embed_size, edge_embed_size), nn.ReLU(),
            nn.Linear(edge_embed_size, edge_embed_size), nn.LayerNorm(edge_embed_size),
        )
        self.time_embed = partial(get_timestep_embedding, embedding_dim=t_embed_size)
        self.position_embed = partial(get_positional_embedding, embedding_dim=pos_embed_size)
        self._dims = (init_embed_size, num_bins, float(min_bin), float(max_bin), edge_embed_size)
        self._wcache = ParamCache()
        self._w16cache = ParamCache()
        self._proj_cache = ParamCache()
        self.arith = default_arith()   # edge embedding kernels: "f16x3" (split-f16 MFMA, default) | "f32" (exact fp32 MFMA), str2str_amd/arith.py
        self.node_arith = default_arith()   # the node MLP (node-stream family: follows the trunk)
        self._idx_key = None
        self._idx_val = None
        self._idx_src = None
        self.node_embed_act = None  # the last node embedding in the node stream's activation format (read by the trunk)

    # ---- derived tensors
    def _weights(self):
        e0, e2, e4 = self.edge_embed[0], self.edge_embed[2], self.edge_embed[4]
        ie, nb = self._dims[0], self._dims[1]
        t1 = ie + 1

        def build():
            w0 = e0.weight.float()
            n0 = self.node_embed[0]
            wn = n0.weight.float()
            out = {
                # first Linear of both MLPs split by feature block: the timestep embedding is one vector per
                # SAMPLE, so its image is a [B, width] GEMM; the fixed-mask column and the positional block are
                # added per residue.  (Also avoids K = 33 / 65 GEMMs, which hit slow BLAS paths at some row counts.)
                "w_row_t": w0[:, :ie].contiguous(), "w_row_f": w0[:, ie].contiguous(),
                "w_col_t": w0[:, t1:t1 + ie].contiguous(), "w_col_f": w0[:, t1 + ie].contiguous(),
                "w_rel": w0[:, 2 * t1:2 * t1 + ie].contiguous(), "b0": e0.bias.float().contiguous(),
                "wn_t": wn[:, :ie].contiguous(), "wn_f": wn[:, ie].contiguous(), "wn_pos": wn[:, t1:t1 + ie].contiguous(),
                "bn0": n0.bias.float().contiguous(),
                # the three timestep blocks as ONE [512, ie] matrix (+ biases): a chunk shares one t, so their images are three slices
                # of one elementwise product + row sum instead of three of each
                "w_t_cat": torch.cat([wn[:, :ie], w0[:, :ie], w0[:, t1:t1 + ie]], dim=0).contiguous(),
                "b_t_cat": torch.cat([n0.bias.float(), e0.bias.float(), torch.zeros_like(e0.bias.float())]).contiguous(),
                "w2p": ops.pack_weight(e2.weight.float()), "w3p": ops.pack_weight(e4.weight.float()),
                "node_mlp": [ops.pack_node_layer(self.node_embed[2].weight, self.node_embed[2].bias),
                             ops.pack_node_layer(self.node_embed[4].weight, self.node_embed[4].bias, True)],
            }
            if self.self_conditioning:
                out["bin_tab"] = w0[:, 2 * t1 + ie:2 * t1 + ie + nb].t().contiguous()
            else:  # no distogram columns: a single zero row that is never selected (ca = 0 -> no bin)
                out["bin_tab"] = w0.new_zeros(1, w0.shape[0])
            out["bin_tab_cb"] = ops.column_blocked(out["bin_tab"])  # gather layout of the split-f16 kernel
            out["bin_lower"] = torch.linspace(self._dims[2], self._dims[3], nb).to(w0.device)
            return out

        ne = self.node_embed
        return self._wcache.get([e0.weight, e0.bias, e2.weight, e4.weight, ne[0].weight, ne[0].bias, ne[2].weight, ne[2].bias,
                                 ne[4].weight, ne[4].bias], build)

    def _fixed_terms(self, fixed_mask: torch.Tensor, w: dict, dev, node_pos: torch.Tensor):
        """Per-target constants of the first layers that do not depend on t: (node MLP: fixed-mask column + positional image [B,L,256],
        edge row term [B,L,128], (edge column term in the f16x3 kernel's gather layout [B,32,L,4], the same row-major)).  Cached on
        the mask tensor, the positional image and the weights."""
        key = (fixed_mask.data_ptr(), tuple(fixed_mask.shape), fixed_mask._version, str(fixed_mask.device), w["wn_f"].data_ptr(), w["wn_f"]._version,
               node_pos.data_ptr(), node_pos._version)
        if key != getattr(self, "_fx_key", None) or getattr(self, "_fx_src", None) is not fixed_mask:
            fixed = fixed_mask.to(dev)[..., None].float()
            B, L = fixed.shape[:2]
            self._fx_val = ((fixed * w["wn_f"] + node_pos).contiguous(), (fixed * w["w_row_f"]).contiguous(),
                            ((fixed[:, None] * w["w_col_f"].view(1, 32, 1, 4)).expand(B, 32, L, 4).contiguous(), (fixed * w["w_col_f"]).contiguous()))
            self._fx_key, self._fx_src = key, fixed_mask
        return self._fx_val

    def time_images(self, t_emb: torch.Tensor) -> torch.Tensor:
        """[n, 32] timestep embeddings -> [n, 512] first-layer images + biases [node MLP | edge row | edge column] (one row per
        step of a schedule: the sampler evaluates this once per trajectory and hands ``forward`` a row as ``t_img``).  Elementwise
        product + sum over the 32 embedding channels (no library GEMM), the same expression ``forward`` evaluates for a single t."""
        w = self._weights()
        t_emb = t_emb.to(w["b0"].device).reshape(-1, t_emb.shape[-1]).float()
        return ((w["w_t_cat"][None] * t_emb[:, None, :]).sum(-1) + w["b_t_cat"]).contiguous()

    def _index_tables(self, residue_idx: torch.Tensor, w_rel: torch.Tensor, wn_pos: torch.Tensor):
        """Per-target constants: first-layer image of the node positional features and the relative-position
        table rel_tab[d + off] = W_rel . posemb(d).  Cached on the index tensor (one host sync per target)."""
        key = (residue_idx.data_ptr(), tuple(residue_idx.shape), residue_idx._version, w_rel.data_ptr(), w_rel._version,
               wn_pos.data_ptr(), wn_pos._version)
        # the key is the tensor's identity, so the tensor itself is kept alive with the cached tables: a later target
        # can then never be allocated at the same (recycled) address and hit tables built for other residue numbering
        if key != self._idx_key or self._idx_src is not residue_idx:
            self._idx_src = residue_idx
            idx_cpu = residue_idx.detach().cpu()
            span = int(idx_cpu.max() - idx_cpu.min())
            d = torch.arange(-span, span + 1)
            dev = w_rel.device
            rel = _table_linear(self.position_embed(d).float().to(dev), w_rel)
            node_pos = _table_linear(self.position_embed(idx_cpu).float().to(dev), wn_pos)  # [B, L, node width]
            self._idx_val = (rel, span, node_pos, residue_idx.to(dev).contiguous())
            self._rel_cb = ops.column_blocked(rel)
            self._idx_key = key
        return self._idx_val

    def forward(self, residue_idx, t, fixed_mask, self_conditioning_ca, node_mask: Optional[torch.Tensor] = None,
                next_proj=None, t_emb: Optional[torch.Tensor] = None, edge_layout: str = "rowmajor", t_img: Optional[torch.Tensor] = None):
        """-> node_embed [B,N,D_node], edge_embed [B,N,N,D_edge] (reference :107-159; ``edge_layout`` "tiled": as an ``ops.PairTiled``
        for the trunk's f16x3 pair kernels, arithmetic "f16x3" only).  ``t`` may live on
        the host (the sampler knows it there): its embedding is then computed on the host and uploaded.
        ``node_mask`` optionally fuses DenoisingNet's mask multiplies (reference :186-187); ``next_proj`` (packed
        pair-projection weights of the first IPA block) adds (attn_bias, pair_z) as a third return value."""
        w = self._weights()
        dev = w["b0"].device
        if dev.type != "cuda":
            raise ops.HipLibraryError("EmbeddingModule runs on the HIP device only (no CPU fallback)")
        if self._dims[4] != 128:
            raise ops.HipLibraryError("edge_embed kernel is built for edge_embed_size=128")
        B, L = residue_idx.shape
        rel_tab, span, node_pos, idx_dev = self._index_tables(residue_idx, w["w_rel"], w["wn_pos"])
        # [B, 32]; evaluated once per DISTINCT t (a sampler chunk shares one t, and the host sin/cos of arguments up to 1e4 rad
        # costs ~40 us per element): same values, row for row
        if t_img is not None:   # the sampler's form: the timestep block's first-layer image of this step, computed with the schedule
            t_emb = None
        elif t_emb is not None:   # the sampler uploads the embeddings of the whole schedule once: no per-step H2D copy
            t_emb = t_emb.to(dev).reshape(-1, t_emb.shape[-1])                 # (a host->device copy here would make the
        else:                                                                  #  host wait for the GPU every evaluation)
            t_u, t_inv = torch.unique(t.detach().reshape(-1), return_inverse=True)
            t_emb = self.time_embed(t_u)[t_inv].to(dev)
        ne = self.node_embed
        mask = None if node_mask is None else node_mask.to(dev).float().contiguous()
        nn_, ne_ = w["wn_t"].shape[0], w["w_row_t"].shape[0]
        single_t = t_img is not None or t_emb.shape[0] == 1
        M = B * L
        f16 = self.arith == "f16x3"
        if single_t:
            # one timestep for the whole chunk (every sampler call): the three [*, 32] first-layer images are one row each -- a 32-term dot
            # product per output channel, evaluated elementwise for all three blocks at once (or handed in as t_img); the fixed-mask and
            # positional terms do not depend on t and are cached.  What is left per evaluation -- relu(t + const) into the node
            # stream's format and the two operand arrays of the edge embedding -- is ONE launch (s2s_embed_assemble; a network
            # evaluation of a small chunk is launch-latency bound: 22 tiny launches here in round 3)
            # (t_img [B, 512]: one image per sample -- trajectories of different t_delta in one batch, sampler.forward_backward_deltas;
            #  the kernel adds row sample of it instead of the one shared row: the same sums per element)
            per_sample = t_img is not None and t_img.ndim == 2 and t_img.shape[0] == B and B > 1
            img = (t_img if per_sample else t_img.reshape(-1)) if t_img is not None else (w["w_t_cat"] * t_emb).sum(-1) + w["b_t_cat"]
            Fn, Fa, Fb = self._fixed_terms(fixed_mask, w, dev, node_pos)
            h_act, node_a, node_b = torch.ops.str2str_amd.embed_assemble(img.contiguous(), Fn, Fa, Fb[0] if f16 else Fb[1], B, L,
                                                                         self.node_arith == "f16x3", f16)
        else:
            fixed = fixed_mask.to(dev)[..., None].float()
            tl = lambda wt, b=None: _table_linear(t_emb, wt, b)  # noqa: E731
            h = F.relu(tl(w["wn_t"], w["bn0"])[:, None, :] + fixed * w["wn_f"] + node_pos)
            h_act = ops.to_act(h.reshape(M, -1).contiguous(), self.node_arith)
        # layers 2, 3 + LayerNorm (+ DenoisingNet's node mask) on the fused node kernels; the packed planes of the result are
        # what the trunk's first projections and every skip_embed read
        nw = w["node_mlp"]
        node_embed, self.node_embed_act = ops.node_apply_chain(h_act, nw, M, (True, False), ln=(ne[5].weight, ne[5].bias, ne[5].eps),
                                                               post_mask=None if mask is None else mask.reshape(M), want_xp=True)
        node_embed = node_embed.view(B, L, -1)
        if not single_t:
            node_a = (tl(w["w_row_t"], w["b0"])[:, None, :] + fixed * w["w_row_f"]).expand(B, L, -1).contiguous()
            if f16:
                node_b = (tl(w["w_col_t"]).view(-1, 32, 1, 4) + fixed[:, None] * w["w_col_f"].view(1, 32, 1, 4)).expand(B, 32, L, 4).contiguous()
            else:
                node_b = (tl(w["w_col_t"])[:, None, :] + fixed * w["w_col_f"]).expand(B, L, -1).contiguous()
        ca = self_conditioning_ca.to(dev).float().contiguous() if self.self_conditioning else node_a.new_zeros(B, L, 3)
        e2, e4, ln = self.edge_embed[2], self.edge_embed[4], self.edge_embed[5]
        if f16:
            e2w, e4w = e2.weight, e4.weight
            ws = self._w16cache.get([e2w, e4w], lambda: ops.pack_f16x3_embed_stream(e2w.float(), e4w.float()))
            proj = None
            if next_proj is not None:  # 5-stage stream: W2 | W3 | the first IPA block's projection stage
                stream = self._proj_cache.get([ws, next_proj["wp_f16x2"]], lambda: torch.cat([ws, next_proj["wp_f16x2"]]))
                proj = (stream, next_proj["b64"])
            edge_embed = ops.edge_embed_f16x3(node_a, node_b, self._rel_cb, w["bin_tab_cb"], w["bin_lower"], idx_dev, ca, ws, e2.bias,
                                              e4.bias, ln.weight, ln.bias, mask, span, ln.eps, proj=proj, column_blocked_tables=True,
                                              out_layout=edge_layout)
        else:
            if edge_layout != "rowmajor":
                raise ops.HipLibraryError("EmbeddingModule: the tiled pair layout belongs to the f16x3 kernels")
            edge_embed = ops.edge_embed(node_a, node_b, rel_tab, w["bin_tab"], w["bin_lower"], idx_dev, ca, w["w2p"],
                                        w["w3p"], e2.bias, e4.bias, ln.weight, ln.bias, mask, span, ln.eps,
                                        proj=None if next_proj is None else (next_proj["wp"], next_proj["b64"]))
        if next_proj is not None:
            edge_embed, *proj = edge_embed
            return node_embed, edge_embed, tuple(proj)
        return node_embed, edge_embed


def _table_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., 32] W^T (+ b) for the per-target / per-timestep tables of the embedder, on the exact fp32 node kernel
    (s2s_node_linear_f32): the package has no library GEMM path."""
    layer = ops.pack_node_layer(w, b if b is not None else torch.zeros(w.shape[0], device=w.device))
    lead, n = x.shape[:-1], x.numel() // x.shape[-1]
    y, _ = ops.node_apply(x.reshape(n, x.shape[-1]).float().contiguous(), layer, n)
    return y[:, : w.shape[0]].reshape(*lead, w.shape[0]).contiguous()


class DenoisingNet(nn.Module):
    def __init__(self, embedder: nn.Module, translator: nn.Module):
        super().__init__()
        self.embedder = embedder
        self.translator = translator
        self.backbone_in_forward = True  # the sampler turns this off inside its loop (result unused there)

    @staticmethod
    def blend_psi(psi: torch.Tensor, torsion_angles_sin_cos: torch.Tensor, fixed_mask: torch.Tensor) -> torch.Tensor:
        """gt_psi * fixed + psi * (1 - fixed) (reference :192-193; the result takes the features' dtype, as there)."""
        gt_psi = torsion_angles_sin_cos.to(psi.device)[..., 2, :]
        fixed_mask = fixed_mask.to(psi.device).type(torch.float)
        return gt_psi * fixed_mask[..., None] + psi * (1 - fixed_mask[..., None])

    def forward(self, batch: dict, as_tensor_7: bool = False, defer_psi_blend: bool = False) -> dict:
        """reference :171-211: {'rigids', 'psi', 'atom37', 'atom14'} (+ 'rigids7', the frames as one tensor).
        ``defer_psi_blend`` (the sampler's loop): 'psi' is the torsion head's own output and 'psi_deferred' is set -- the blend with the
        input torsion under the fixed mask (:192-193; float64 for float64 features) feeds nothing inside the loop, the sampler
        applies it once to the last evaluation (``blend_psi``)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise ops.HipLibraryError(
                "DenoisingNet.forward needs the HIP device (MI355X): the sampling path has no CPU fallback")
        node_mask = batch["residue_mask"].to(dev).type(torch.float)
        fixed_mask = batch["fixed_mask"].to(dev).type(torch.float)
        fuse = getattr(self.translator, "fuse_pair_projection", False)
        # between the f16x3 pair kernels (embedding -> EdgeTransition 0 -> 1 -> ..) the pair tensor travels in their tiled layout
        # (producer and consumer both on their f16x3 kernels; the consumer of the embedding's pair tensor is EdgeTransition 0)
        et0 = self.translator.trunk["edge_transition_0"] if "edge_transition_0" in getattr(self.translator, "trunk", {}) else None
        tiled = fuse and getattr(self.embedder, "arith", None) == "f16x3" and getattr(et0, "arith", None) == "f16x3"
        emb = self.embedder(residue_idx=batch["residue_idx"], t=batch["t"], fixed_mask=fixed_mask,
                            self_conditioning_ca=batch["sc_ca_t"], node_mask=node_mask, t_emb=batch.get("t_emb"), t_img=batch.get("t_img"),
                            next_proj=self.translator.trunk["ipa_0"].pair_proj_weights() if fuse else None,
                            **({"edge_layout": "tiled"} if tiled else {}))
        node_embed, edge_embed = emb[0], emb[1]
        tb = dict(batch)
        tb["residue_mask"], tb["fixed_mask"] = node_mask, fixed_mask
        tb["rigids_t"] = batch["rigids_t"].to(dev)
        tb["_node_embed_act"] = getattr(self.embedder, "node_embed_act", None)
        model_out = self.translator(node_embed, edge_embed, tb, **({"_first_proj": emb[2]} if fuse else {}))
        if model_out.get("psi_blended") or defer_psi_blend:     # (the torsion head's kernel blends float32 features in its own launch)
            psi_pred = model_out["psi"]
        else:
            psi_pred = self.blend_psi(model_out["psi"], batch["torsion_angles_sin_cos"], fixed_mask)
        rigids_pred = model_out["out_rigids"]
        out = {"rigids": rigids_pred, "psi": psi_pred, "rigids7": model_out["out_rigids7"]}
        if defer_psi_blend and not model_out.get("psi_blended"):
            out["psi_deferred"] = True
        if self.backbone_in_forward:
            aatype = batch["aatype"].to(dev) if "aatype" in batch else None
            bb = compute_backbone(rigids_pred, psi_pred, aatype=aatype, _rigids7=model_out["out_rigids7"])
            out["atom37"], out["atom14"] = bb[0], bb[-1]
        if as_tensor_7:
            out["rigids"] = model_out["out_rigids7"]
        return out
