"""Oracle: the SE(3) score network forward (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional restatement over a reference-keyed ``state_dict`` (the 274 tensors of
``DenoisingNet.state_dict()``, SURVEY.md §5 checkpoint row).  Dimensions are read from the
tensors themselves, so any width of the reference architecture works.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .geometry import Frames, compute_backbone

SD = Dict[str, torch.Tensor]


def _lin(x, sd: SD, p: str):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(x, sd: SD, p: str, eps: float = 1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


# ------------------------------------------------------------------ denoising_ipa.py:13-46
def positional_embedding(indices: torch.Tensor, dim: int, max_len: int = 2056) -> torch.Tensor:
    """denoising_ipa.py:13-31: sin/cos of idx*pi / max_len^(2k/dim), k = 0..dim/2-1."""
    K = torch.arange(dim // 2)
    arg = indices[..., None] * math.pi / (max_len ** (2 * K[None] / dim))
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def timestep_embedding(t: torch.Tensor, dim: int, max_len: int = 10000) -> torch.Tensor:
    """denoising_ipa.py:34-46."""
    t = t * max_len
    half = dim // 2
    emb = math.log(max_len) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float) * -emb)
    emb = t.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def calc_distogram(pos: torch.Tensor, min_bin: float, max_bin: float, num_bins: int) -> torch.Tensor:
    """src/common/geo_utils.py:44-56: strict > lower and < upper (last upper = 1e8)."""
    d = torch.linalg.norm(pos[..., :, None, :] - pos[..., None, :, :], dim=-1)[..., None]
    lower = torch.linspace(min_bin, max_bin, num_bins)
    upper = torch.cat([lower[1:], lower.new_tensor([1e8])], dim=-1)
    return ((d > lower) * (d < upper)).type(pos.dtype)


def embedding(sd: SD, residue_idx, t, fixed_mask, sc_ca, *, num_bins=22, min_bin=1e-5, max_bin=20.0,
              prefix="embedder."):
    """EmbeddingModule.forward, denoising_ipa.py:107-159 (self_conditioning=True)."""
    B, L = residue_idx.shape
    init = (sd[prefix + "node_embed.0.weight"].shape[1] - 1) // 2  # t_embed == pos_embed size
    fixed = fixed_mask[..., None].float()
    t_embed = torch.tile(timestep_embedding(t, init)[:, None, :], (1, L, 1))
    t_embed = torch.cat([t_embed, fixed], dim=-1)
    pair = [
        torch.cat(
            [torch.tile(t_embed[:, :, None, :], (1, 1, L, 1)), torch.tile(t_embed[:, None, :, :], (1, L, 1, 1))],
            dim=-1,
        ).float().reshape(B, L * L, -1)
    ]
    node = [t_embed, positional_embedding(residue_idx, init)]
    rel = (residue_idx[:, :, None] - residue_idx[:, None, :]).reshape(B, L * L)
    pair.append(positional_embedding(rel, init))
    pair.append(calc_distogram(sc_ca, min_bin, max_bin, num_bins).reshape(B, L * L, -1))

    def mlp(x, p):
        x = F.relu(_lin(x, sd, p + ".0"))
        x = F.relu(_lin(x, sd, p + ".2"))
        x = _lin(x, sd, p + ".4")
        return _ln(x, sd, p + ".5")

    node_embed = mlp(torch.cat(node, dim=-1).float(), prefix + "node_embed")
    edge_embed = mlp(torch.cat(pair, dim=-1).float(), prefix + "edge_embed").reshape(B, L, L, -1)
    return node_embed, edge_embed


# ------------------------------------------------------------------ ipa.py:100-268
def ipa(sd: SD, p: str, s, z, r: Frames, mask, *, no_heads=8, no_qk_points=8, no_v_points=12,
        inf=1e5, eps=1e-8):
    """InvariantPointAttention.forward, ipa.py:100-268.  ``r`` translations are already x0.1."""
    H, Pq, Pv = no_heads, no_qk_points, no_v_points
    C = sd[p + ".linear_q.weight"].shape[0] // H
    q = _lin(s, sd, p + ".linear_q").view(s.shape[:-1] + (H, -1))
    kv = _lin(s, sd, p + ".linear_kv").view(s.shape[:-1] + (H, -1))
    k, v = torch.split(kv, C, dim=-1)

    R = r.get_rot_mats()[..., None, :, :]  # [B,N,1,3,3]
    T = r.trans[..., None, :]

    def pts(name):
        x = _lin(s, sd, p + name)
        x = torch.stack(torch.split(x, x.shape[-1] // 3, dim=-1), dim=-1)  # coordinate-major
        xr = torch.stack(
            [
                R[..., i, 0] * x[..., 0] + R[..., i, 1] * x[..., 1] + R[..., i, 2] * x[..., 2]
                for i in range(3)
            ],
            dim=-1,
        )
        return xr + T

    q_pts = pts(".linear_q_points").view(s.shape[:-1] + (H, Pq, 3))
    kv_pts = pts(".linear_kv_points").view(s.shape[:-1] + (H, -1, 3))
    k_pts, v_pts = torch.split(kv_pts, [Pq, Pv], dim=-2)

    b = _lin(z, sd, p + ".linear_b")  # [B,N,N,H]
    a = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1))  # [B,H,N,N]
    a = a * math.sqrt(1.0 / (3 * C))
    a = a + math.sqrt(1.0 / 3) * b.permute(0, 3, 1, 2)
    disp = q_pts.unsqueeze(-4) - k_pts.unsqueeze(-5)  # [B,N,N,H,Pq,3]
    pt_att = disp**2
    pt_att = sum(torch.unbind(pt_att, dim=-1))
    hw = F.softplus(sd[p + ".head_weights"]).view(1, 1, 1, -1, 1)
    hw = hw * math.sqrt(1.0 / (3 * (Pq * 9.0 / 2)))
    pt_att = torch.sum(pt_att * hw, dim=-1) * (-0.5)
    sq = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    sq = inf * (sq - 1)
    a = a + pt_att.permute(0, 3, 1, 2)
    a = a + sq.unsqueeze(-3)
    a = torch.softmax(a, dim=-1)

    o = torch.matmul(a, v.transpose(-2, -3)).transpose(-2, -3)
    o = o.reshape(o.shape[:-2] + (-1,))
    # [B,H,3,N,Pv]
    o_pt = torch.sum(a[..., None, :, :, None] * v_pts.permute(0, 2, 4, 1, 3)[..., None, :, :], dim=-2)
    o_pt = o_pt.permute(0, 3, 1, 4, 2)  # [B,N,H,Pv,3]
    Rt = r.get_rot_mats()[..., None, None, :, :].transpose(-1, -2)
    d = o_pt - r.trans[..., None, None, :]
    o_pt = torch.stack(
        [Rt[..., i, 0] * d[..., 0] + Rt[..., i, 1] * d[..., 1] + Rt[..., i, 2] * d[..., 2] for i in range(3)],
        dim=-1,
    )
    o_norm = torch.sqrt(torch.sum(o_pt**2, dim=-1) + eps)
    o_norm = o_norm.reshape(o_norm.shape[:-2] + (-1,))
    o_pt = o_pt.reshape(o_pt.shape[:-3] + (-1, 3))
    pair_z = _lin(z, sd, p + ".down_z")
    o_pair = torch.matmul(a.transpose(-2, -3), pair_z)
    o_pair = o_pair.reshape(o_pair.shape[:-2] + (-1,))
    feats = torch.cat([o, *torch.unbind(o_pt, dim=-1), o_norm, o_pair], dim=-1)
    return _lin(feats, sd, p + ".linear_out")


# ------------------------------------------------------------------ layers.py
def node_transition(sd: SD, p: str, s):
    """layers.py:128-145."""
    s0 = s
    s = F.relu(_lin(s, sd, p + ".linear_1"))
    s = F.relu(_lin(s, sd, p + ".linear_2"))
    s = _lin(s, sd, p + ".linear_3")
    return _ln(s + s0, sd, p + ".ln")


def edge_transition(sd: SD, p: str, node, edge):
    """layers.py:170-185 (num_layers=2 trunk: Linear,ReLU,Linear,ReLU)."""
    node = _lin(node, sd, p + ".initial_embed")
    B, N, _ = node.shape
    bias = torch.cat(
        [torch.tile(node[:, :, None, :], (1, 1, N, 1)), torch.tile(node[:, None, :, :], (1, N, 1, 1))], dim=-1
    )
    x = torch.cat([edge, bias], dim=-1).reshape(B * N * N, -1)
    h = F.relu(_lin(x, sd, p + ".trunk.0"))
    h = F.relu(_lin(h, sd, p + ".trunk.2"))
    y = _lin(h + x, sd, p + ".final_layer")
    y = _ln(y, sd, p + ".layer_norm")
    return y.reshape(B, N, N, -1)


def torsion_head(sd: SD, p: str, s, eps=1e-8):
    """layers.py:199-213 (linear_3 is dead weight)."""
    s0 = s
    s = F.relu(_lin(s, sd, p + ".linear_1"))
    s = _lin(s, sd, p + ".linear_2")
    s = s + s0
    u = _lin(s, sd, p + ".linear_final")
    return u / torch.sqrt(torch.clamp(torch.sum(u**2, dim=-1, keepdim=True), min=eps))


def transformer_encoder(sd: SD, p: str, x, key_padding_float, nhead: int):
    """nn.TransformerEncoder as built at ipa.py:312-317 and called at :357: post-norm layers,
    relu, seq-first [N,B,D]; the FLOAT key-padding mask is ADDED to the logits."""
    L, B, D = x.shape
    dh = D // nhead
    n_layers = 0
    while f"{p}.layers.{n_layers}.self_attn.in_proj_weight" in sd:
        n_layers += 1
    for l in range(n_layers):
        q = f"{p}.layers.{l}"
        qkv = F.linear(x, sd[q + ".self_attn.in_proj_weight"], sd[q + ".self_attn.in_proj_bias"])
        qh, kh, vh = qkv.chunk(3, dim=-1)

        def heads(t):
            return t.reshape(L, B * nhead, dh).transpose(0, 1)  # [B*h, L, dh]

        qh, kh, vh = heads(qh), heads(kh), heads(vh)
        att = torch.bmm(qh * (1.0 / math.sqrt(dh)), kh.transpose(1, 2))
        if key_padding_float is not None:
            att = att + key_padding_float[:, None, None, :].expand(B, nhead, 1, L).reshape(B * nhead, 1, L)
        att = torch.softmax(att, dim=-1)
        sa = torch.bmm(att, vh).transpose(0, 1).reshape(L, B, D)
        sa = _lin(sa, sd, q + ".self_attn.out_proj")
        x = _ln(x + sa, sd, q + ".norm1")
        ff = _lin(F.relu(_lin(x, sd, q + ".linear1")), sd, q + ".linear2")
        x = _ln(x + ff, sd, q + ".norm2")
    return x


# ------------------------------------------------------------------ ipa.py:331-387
def translation_ipa(sd: SD, node, edge, batch, *, coordinate_scaling=0.1, transformer_num_heads=4,
                    no_heads=8, no_qk_points=8, no_v_points=12, prefix="translator."):
    node_mask = batch["residue_mask"].type(torch.float)
    diffuse_mask = (1 - batch["fixed_mask"].type(torch.float)) * node_mask
    edge_mask = node_mask[..., None] * node_mask[..., None, :]
    init_frames = batch["rigids_t"].type(torch.float)
    curr = Frames.from_tensor_7(torch.clone(init_frames)).scale_translation(coordinate_scaling)
    init_node = node
    nb = 0
    while f"{prefix}trunk.ipa_{nb}.linear_q.weight" in sd:
        nb += 1
    T = prefix + "trunk."
    for b in range(nb):
        ipa_embed = ipa(sd, f"{T}ipa_{b}", node, edge, curr, node_mask, no_heads=no_heads,
                        no_qk_points=no_qk_points, no_v_points=no_v_points)
        ipa_embed = ipa_embed * node_mask[..., None]
        node = _ln(node + ipa_embed, sd, f"{T}ipa_ln_{b}")
        cat = torch.cat([node, _lin(init_node, sd, f"{T}skip_embed_{b}")], dim=-1).transpose(0, 1)
        tr = transformer_encoder(sd, f"{T}transformer_{b}", cat, 1.0 - node_mask, transformer_num_heads)
        node = node + _lin(tr.transpose(0, 1), sd, f"{T}linear_{b}")
        node = node_transition(sd, f"{T}node_transition_{b}", node)
        node = node * node_mask[..., None]
        upd = _lin(node * diffuse_mask[..., None], sd, f"{T}bb_update_{b}.linear")
        curr = curr.compose_q_update_vec(upd, diffuse_mask[..., None])
        if b < nb - 1:
            edge = edge_transition(sd, f"{T}edge_transition_{b}", node, edge) * edge_mask[..., None]
    psi = torsion_head(sd, prefix + "torsion_pred", node)
    curr = curr.unscale_translation(coordinate_scaling)
    return curr, psi, node, edge


def denoising_net(sd: SD, batch, **kw):
    """DenoisingNet.forward, denoising_ipa.py:171-211.  Returns dict(rigids=Frames(quat), psi,
    atom37, atom14)."""
    node_mask = batch["residue_mask"].type(torch.float)
    fixed_mask = batch["fixed_mask"].type(torch.float)
    edge_mask = node_mask[..., None] * node_mask[..., None, :]
    node, edge = embedding(sd, batch["residue_idx"], batch["t"], fixed_mask, batch["sc_ca_t"])
    node = node * node_mask[..., None]
    edge = edge * edge_mask[..., None]
    rigids, psi, _, _ = translation_ipa(sd, node, edge, batch, **kw)
    gt_psi = batch["torsion_angles_sin_cos"][..., 2, :]
    psi_pred = gt_psi * fixed_mask[..., None] + psi * (1 - fixed_mask[..., None])
    atom37, _, _, atom14 = compute_backbone(rigids, psi_pred, batch.get("aatype"))
    return {"rigids": rigids, "psi": psi_pred, "atom37": atom37, "atom14": atom14}
