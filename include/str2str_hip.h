/* str2str_hip.h — C ABI of libstr2str_hip.so (MI355X / gfx950 kernels for the Str2Str sampling path).
 *
 * Drop-in boundary: the reference has no FFI layer — its seams are Python call
 * sites.  Each entry point below replaces the chain of eager PyTorch ops behind ONE such call site
 * (cited per function as reference file:line) and is what a binding for that call site would load:
 * plain device pointers, sizes and a HIP stream; no torch types.  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (in this repo: torch tensors); kernels never
 *     allocate, free or synchronise; outputs must be pre-allocated, contiguous, 16-byte aligned;
 *   - `stream` is a hipStream_t (0 = default stream); work is only enqueued;
 *   - float = IEEE binary32; frames are "tensor_7": quaternion (w,x,y,z) + translation (x,y,z)
 *     (Rigid.to_tensor_7, src/common/rigid_utils.py:1203-1215);
 *   - return value: 0 on success, otherwise a hipError_t (argument errors = hipErrorInvalidValue);
 *   - re-entrant across streams; the only global state is the immutable backbone table.
 */
#ifndef STR2STR_HIP_H
#define STR2STR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Library ABI version (bumped on any signature change). */
int s2s_abi_version(void);

/* Arithmetic.  Every matrix product of the path exists in two forms behind the same operator contract:
 *   "f16x3" (default): operands split into f16 pairs, three products per block on the 16-bit matrix cores, fp32 accumulation --
 *           fp32-equivalent in precision, but an activation must stay below f16's 65504;
 *   "f32":  exact fp32 MFMA, no range limit (s2s_edge_transition, s2s_edge_embed, s2s_ipa_attention, s2s_node_linear_f32,
 *           s2s_encoder_attention): the reference arithmetic and the automatic fallback.
 * Range guard: every f16x3 kernel keeps a running maximum of the values it splits and ORs a bit into word 0 of the 8-int device
 * buffer registered here when that maximum reaches 2^15 or is not finite (bits: 1 node GEMM, 2 pack_planes, 4 edge transition,
 * 8 edge embedding, 16 IPA points, 32 encoder attention, 64 IPA attention).  Words 1..7 (1 + log2(bit)) collect magnitude buckets
 * of the same families: bit e set = a launch saw a maximum in [2^(8+e), 2^(9+e)) (e = 8: 2^16 or more); nothing is written below
 * 2^8.  The caller clears / reads the buffer (no kernel waits for it); NULL disables the reports.  The caller decides what a raised
 * bit means: str2str_amd/sampler.py re-runs the chunk with ONLY the flagged kernel families on their exact fp32 kernels. */
int s2s_set_range_flag(int* device_words);

/* ---- Pair-stream MLPs (fp32 MFMA).  Weight blobs are "packed" for the kernels' lane order:
 *      packed[((s4*T + t)*64 + lane)*4 + q] = W[32*t + (lane & 31)][8*s4 + 4*(lane >> 5) + q]
 *      for W [32*T, 8*S4] row-major (str2str_amd.ops.pack_weight does this).  s2s_edge_transition takes the
 *      TILE-MAJOR order instead: packed[((t*S4 + s4)*64 + lane)*4 + q] (pack_weight(..., tile_major=True)). ---- */

/* EdgeTransition.forward (src/models/net/layers.py:170-185) followed by the edge-mask multiply of
 * TranslationIPA.forward (src/models/net/ipa.py:371-372).
 *   edge     [B,N,N,128]  in        node_ab [B,N,768] = [W1[:,128:256].n'+b1 | W1[:,256:384].n']
 *   node_p   [B,N,128]    n' = initial_embed(node)
 *   w1_packed: W1[:, :128] (384x128); w2_packed: W2 (384x384); wf_packed: final_layer.weight (128x384)
 *   b2 [384], bf [128], ln_gamma/ln_beta [128], mask [B,N] or NULL, out [B,N,N,128] (may not alias edge).
 *   Optional fused epilogue (proj_w_packed != NULL): the NEXT IPA block's linear_b/down_z (see s2s_pair_project)
 *   applied to the freshly produced pair vectors while they are still in registers -> proj_attn_bias, proj_pair_z. */
int s2s_edge_transition(const float* edge, const float* node_ab, const float* node_p, const float* w1_packed,
                        const float* w2_packed, const float* wf_packed, const float* b2, const float* bf,
                        const float* ln_gamma, const float* ln_beta, const float* mask, float* out, int n_samples,
                        int n_res, float ln_eps, const float* proj_w_packed, const float* proj_bias_cat64,
                        float* proj_attn_bias, float* proj_pair_z, void* stream);

/* The same operator on split-f16 MFMA ("f16x3", csrc/pair_mlp_f16.hip; the default): every fp32 operand as two f16 numbers (11 + 11
 * bits + the residue's sign = fp32's 24), products w_h x_h + w_h x_l + w_l x_h with the weights stored as the split of 2^5 w (2^-5
 * back in the epilogues), fp32 accumulation -- three matrix instructions per block, the dropped w_l x_l below one fp32 rounding.
 * Activations must stay below f16's 65504 (range guard above).  weight_stream: 30 (+1 with the fused projection) stages x 32 KiB
 * of f16 A fragments (W_h, W_l) in slot order (ops.pack_f16x3_stream, ops.pack_f16x2_layer for the projection stage; the order is
 * part of the ABI version).
 *   Optional fused epilogue (proj_attn_bias != NULL): the NEXT IPA block's linear_b / down_z (see s2s_pair_project);
 *   the stream then carries a 31st stage with the chain-packed 64x128 [linear_b; down_z; 0] matrix, proj_bias_cat64 [64];
 *   outputs proj_attn_bias [B,8,N,N] (head-major) and proj_pair_z [B,N,N,32].
 *   Pair-tensor layouts (io_layout; 0 = the reference's [B,N,N,128] on both sides).  Between the library's own pair kernels the tensor
 *   may travel TILED: the flat pair sequence in blocks of 32 pairs (16 KiB), block b = pairs 32 b .. 32 b + 31, inside a block
 *   [16 groups g][2 halves h][32 pairs n][4 floats] = channels 8 g + 4 h .. + 3 of pair 32 b + n at float offset 4096 b + 256 g + 128 h + 4 n
 *   -- the order in which a wavefront's lanes hold a 32-pair tile, so that one load / store instruction covers 8 whole cache lines
 *   instead of 32 B in each of 32 (ops.pair_tiled / ops.pair_untiled convert).  The buffer of a tiled tensor holds whole blocks
 *   (B N N rounded up to 32 pairs; the padding is never read as data).
 *     io_layout bit 0: edge is tiled;  bit 1: out is written tiled;  bit 2: out is not written at all (out may be NULL; needs the
 *     fused projection -- the last EdgeTransition of the trunk, whose pair vectors only feed the next block's projections).
 *   node_ab HERE is [B,N,896] = [ 2^-e (W1[:,128:256].n' + b1) | 2^5 W1[:,256:384].n' | 2^(5-e) (Wf[:,256:384].n' + bf) ]  (e = prescale_exp
 *   below; W1 = the first hidden layer, Wf / bf = final_layer): everything the pair (i, j) takes from its two NODES through a linear
 *   layer, computed once per node.  The column half B_j (384) is the START VALUE of the layer-1 accumulators, which carry 2^5 x the
 *   layer output, and is loaded straight into them; the row half A_i (384) is added in the epilogue at the planes' scale; the third
 *   group G_j (128) is the start value of the FINAL layer's accumulators: the layer reads x = h2 + [e | n'_i | n'_j]
 *   (layers.py:181), and the part of it that depends on node j alone, Wf[:,256:384].n'_j + bf, does not have to be added to the
 *   hidden values of every pair.  (The i-side residual n'_i is still added in the kernel: node_p [B,N,128] = n'.)  All factors are
 *   powers of two: the caller folds them into the per-node layers that produce node_ab (EdgeTransition.ab16_specs) or scales a plain
 *   node_ab (ops.edge_transition_f16x3 does, unless told that node_ab has this form already). */
int s2s_edge_transition_f16x3(const float* edge, const float* node_ab, const float* node_p, const void* weight_stream,
                              const float* b2, const float* ln_gamma, const float* ln_beta,
                              const float* mask, float* out, int n_samples, int n_res, float ln_eps, int io_layout,
                              const float* proj_bias_cat64, float* proj_attn_bias, float* proj_pair_z, int prescale_exp,
                              void* stream);
/*   prescale_exp = e in 0 .. 15 (0: none): BLOCK EXPONENT of the hidden activations.  The two hidden layers' outputs (relu(layer 1),
 *   relu(layer 2) + x) are kept as f16 planes of 2^-e x their value: relu is positively homogeneous, so the factor rides in
 *   constants the epilogues apply anyway and LayerNorm removes it -- exact for a power of two, no extra instruction, e = 0 is bit
 *   for bit the unscaled kernel.  The row half and the third group of node_ab must then be handed in multiplied by 2^-e (see above).
 *   It moves the kernel's usable range from 2^15 to 2^(15+e) at the price of f16's subnormal spacing on activations below
 *   2^(e-3) (absolute error 2^(e-25): far below the large activations' own rounding); the range guard sees the SCALED values. */

/* EmbeddingModule.forward, edge branch (src/models/net/denoising_ipa.py:137-158, calc_distogram
 * src/common/geo_utils.py:44-56) + edge-mask multiply (denoising_ipa.py:187).
 *   node_a/node_b [B,N,128]: row / column parts of the first Linear (incl. bias in node_a)
 *   rel_table [n_rel,128]: first-layer image of posemb(d), d = idx_i - idx_j, row d + rel_offset
 *   bin_table [n_bins,128]: first-layer columns of the distogram one-hot; bin_lower [n_bins]
 *   residue_idx [B,N] int64; ca_xyz [B,N,3] (self-conditioning CA, Angstrom)
 *   w2/w3 packed 128x128; b2,b3,ln_gamma,ln_beta [128]; out [B,N,N,128]; proj_*: optional fused pair projection as above. */
int s2s_edge_embed(const float* node_a, const float* node_b, const float* rel_table, const float* bin_table,
                   const float* bin_lower, const long long* residue_idx, const float* ca_xyz, const float* w2_packed,
                   const float* w3_packed, const float* b2, const float* b3, const float* ln_gamma, const float* ln_beta,
                   const float* mask, float* out, int n_samples, int n_res, int rel_offset, int n_rel, int n_bins,
                   float ln_eps, const float* proj_w_packed, const float* proj_bias_cat64, float* proj_attn_bias,
                   float* proj_pair_z, void* stream);

/* The edge embedding on split-f16 MFMA (see s2s_edge_transition_f16x3; the default).  weight_stream: 4 stages x 32 KiB (W2 | W3,
 * chain-packed (W_h, W_l) fragments in slot order: ops.pack_f16x3_embed_stream) + a 5th stage with the [linear_b; down_z; 0]
 * matrix when the fused projection is requested (proj_attn_bias != NULL).
 *   Layout difference to s2s_edge_embed: node_b, rel_table and bin_table are COLUMN-BLOCKED --
 *   node_b [B][32][N][4], rel_table [32][n_rel][4], bin_table [32][n_bins][4], element [c][row][q] = channel 4c + q of that row
 *   (ops.column_blocked) -- so that the gathers of neighbouring pairs share cache lines; node_a stays [B,N,128];
 *   bin_lower must ascend (torch.linspace), n_bins <= 32.  Pair indices are 32-bit inside a launch: the entry point splits the
 *   samples over several launches when B*N*N >= 2^31 (N*N itself and n_rel*512 must stay below 2^31 / 2^32).
 *   out_tiled = 1: out is written in the tiled pair layout (s2s_edge_transition_f16x3), its buffer holds whole 32-pair blocks. */
int s2s_edge_embed_f16x3(const float* node_a, const float* node_b, const float* rel_table, const float* bin_table,
                         const float* bin_lower, const long long* residue_idx, const float* ca_xyz,
                         const void* weight_stream, const float* b2, const float* b3, const float* ln_gamma,
                         const float* ln_beta, const float* mask, float* out, int n_samples, int n_res, int rel_offset,
                         int n_rel, int n_bins, float ln_eps, int out_tiled, const float* proj_bias_cat64,
                         float* proj_attn_bias, float* proj_pair_z, void* stream);

/* linear_b and down_z of InvariantPointAttention (src/models/net/ipa.py:177, :253) in one pass over z.
 *   w_packed: [linear_b.weight (8 rows); down_z.weight (32 rows); 24 zero rows] (64x128) packed
 *   bias_cat64 [64]; attn_bias [B,8,N,N] (head-major: what s2s_ipa_attention streams per head); pair_z [B,N,N,32]. */
int s2s_pair_project(const float* edge, const float* w_packed, const float* bias_cat64, float* attn_bias, float* pair_z,
                     int n_samples, int n_res, void* stream);

/* ---- Invariant point attention ---- */

/* Point generation: split/stack of the coordinate-major linear outputs and Rigid.apply
 * (src/models/net/ipa.py:144-171; src/common/rigid_utils.py:1107-1120).
 *   rigids7 [M,7] (translation already x coordinate_scaling), q_pts_lin [M,3*H*Pq], kv_pts_lin [M,3*H*(Pq+Pv)]
 *   q_pts,k_pts [M,H,Pq*3]; v_pts [M,H,v_pts_stride] as (x,y,z,0) per point, zero padded (stride 64). */
int s2s_ipa_prep_points(const float* rigids7, const float* q_pts_lin, const float* kv_pts_lin, float* q_pts, float* k_pts,
                        float* v_pts, long long n_frames, int n_heads, int n_qk_points, int n_v_points, int v_pts_stride,
                        void* stream);

/* Attention core of InvariantPointAttention.forward (src/models/net/ipa.py:183-252): logits
 * (scalar + pair bias + point distances + mask), softmax over keys, o / o_pt (inverse-transformed,
 * with norms), written in linear_out's concat order (ipa.py:259-266):
 *   out [B,N, H*C | H*Pv (x) | H*Pv (y) | H*Pv (z) | H*Pv (norm) | H*c_pair_z]   (the o_pair columns are left to
 *   s2s_ipa_opair).
 *   q [B,N,H,C]; kv [B,N,H,2C]; attn_bias [B,H,N,N]; head_w_scaled [H] = softplus(head_weights)*sqrt(1/(3*(Pq*9/2))).
 *   logits_out [B,H,N,N] (masked logits; may alias attn_bias), stats_out [B,H,N,2] (row maximum, sum of exp).
 * Supported shape: C=256, Pq=8, Pv=12, c_pair_z=32, H multiple of 4 (configs/model/diffusion.yaml:29-40). */
int s2s_ipa_attention(const float* q, const float* kv, const float* q_pts, const float* k_pts, const float* v_pts64,
                      const float* attn_bias, float* logits_out, float* stats_out, const float* mask, const float* rigids7,
                      const float* head_w_scaled, float* out, int n_samples, int n_res, int n_heads, int c_hidden,
                      int n_qk_points, int n_v_points, int c_pair_z, float inf, float eps, void* stream);

/* The DEFAULT attention core (csrc/ipa_attention_f16w.hip), on operands that are ALREADY split into f16 pairs (x_h, x_l) in MFMA
 * fragment order: q_xp / k_xp = packed planes [rows/32][16 H][2][64][8] of the q and k projections (out_xp of s2s_node_linear with
 * linear_q and the k rows of linear_kv), v_vf = s2s_node_linear_vfrag of the v rows of linear_kv.
 * s2s_ipa_prep_points_f16 (ipa.py:144-171) writes the global-frame points as fragments: qp_xp [rows/32][H][2][2][64][8] (query
 * points x head_w_scaled[h] / sqrt(1/(3 c_hidden))), kp_xp (key points), vp_vf [rows/32][H][2][2][2][64][8] (value points as
 * (x,y,z,0) groups), and q2 / k2 [rows/32][H][32] = -1/2 head_w_scaled[h] |points|^2: the point term of the logits (ipa.py:191-205)
 * is evaluated as  w q.k - w/2 |q|^2 - w/2 |k|^2  with the cross term on the matrix cores.
 * s2s_ipa_attention_f16w: three products per block (a_h b_h + a_h b_l + a_l b_h), one wave per query tile (a workgroup is four
 * query tiles; a wave runs the whole contraction and owns all ten output tiles).  out [B,N,feat] receives only the o_pt columns
 * (H*c_hidden ..); the o columns are written as packed planes (k-steps 16 h .. 16 h + 15 of an activation with out_xp_ksteps
 * k-steps per row: the input of linear_out); logits_out / stats_out as s2s_ipa_attention (consumed by s2s_ipa_opair).
 * ANY n_res (ipa.py:183-257 has one code path for every length): with n_pad = n_res rounded up to 32, the fragment arrays, q2 and k2
 * hold n_pad rows PER SAMPLE (row tile = sample * n_pad/32 + tile):
 *   q_xp / k_xp from s2s_node_linear and v_vf from s2s_node_linear_vfrag, both with n_rows = n_samples * n_pad and the row map
 *   (map_pad = n_pad, map_src = n_res) when n_pad != n_res; points from s2s_ipa_prep_points_f16 (padded rows: zero points,
 *   k2 = -1e9, so a padded key never carries probability -- exactly the un-padded softmax).
 * attn_bias stays [B,H,n_res,n_res]; when n_pad != n_res logits_out must be a SEPARATE [B,H,n_pad,n_pad] buffer (s2s_ipa_opair:
 * logits_ld = n_pad); stats_out [B,H,n_res,2], out [B,n_res,feat] and out_xp (row tiles of the flat [B n_res] rows) are not padded.
 *
 * FOLDED PROJECTIONS (the default host path, str2str_amd/models/net/ipa.py fold_ipa_weights; ipa.py:131-143,183-190,229-252 of the
 * reference with the weights regrouped).  With q = W_q s_i + b_q, k = W_k s_j + b_k per head, q.k = s_j . (W_k^T W_q s_i + W_k^T b_q)
 * + terms that depend on i only and cancel in the softmax over j; and sum_j a_ij v_j = W_v (sum_j a_ij s_j) + b_v as the
 * probabilities sum to one.  So with c_s = c_hidden = 256 the attention can read the block's input s as the K AND the V operand of
 * every head: q' = (W_k^T W_q) s + W_k^T b_q is the only scalar projection left (linear_k / linear_v are never evaluated; W_v and
 * b_v move into linear_out's weight and bias), and the kernel aggregates s instead of v.  n_kv_heads = 1 selects that operand
 * layout: k_xp = packed planes of s [rows/32][16][2][64][8] (s itself when n_res % 32 == 0, else k_shared below), v_vf =
 * [rows/32][8][2][2][64][8] (v_shared below); n_kv_heads = n_heads is the per-head layout above.
 * Short chains with the shared operands (n_pad <= 64: one or two key tiles per sample) run on a second kernel behind the same entry
 * point: one wave per (sample, head, query tile), no LDS and no barriers -- the streaming kernel's workgroup of four query tiles of
 * one (sample, head) repeats work on three / two of its waves there; S2S_IPA_SHORT=0 keeps the streaming kernel for every length.
 * s2s_ipa_prep_points_f16 with s_xp != NULL (n_heads = 8) writes those shared operands in the same launch: v_shared always,
 * k_shared (rows gathered into the padded per-sample layout) when n_res % 32 != 0; NULL s_xp skips the step. */
int s2s_ipa_prep_points_f16(const float* rigids7, const float* q_pts_lin, const float* kv_pts_lin, const float* head_w_scaled,
                            void* qp_xp, void* kp_xp, void* vp_vf, float* q2, float* k2, int n_samples, int n_res, int n_heads,
                            int n_qk_points, int n_v_points, int c_hidden, const void* s_xp, void* k_shared, void* v_shared,
                            void* stream);
int s2s_ipa_attention_f16w(const void* q_xp, const void* k_xp, const void* v_vf, const void* qp_xp, const void* kp_xp,
                             const void* vp_vf, const float* q2, const float* k2, const float* attn_bias, float* logits_out,
                             float* stats_out, const float* mask, const float* rigids7, float* out, void* out_xp,
                             int out_xp_ksteps, int n_samples, int n_res, int n_heads, int c_hidden, int n_qk_points,
                             int n_v_points, int c_pair_z, float inf, float eps, int n_kv_heads, void* stream);

/* The pair term of InvariantPointAttention.forward (src/models/net/ipa.py:253-257):
 *   o_pair[b,i,h,:] = sum_j softmax_j(logits[b,h,i,:])[j] * pair_z[b,i,j,:]
 * from the logits / statistics s2s_ipa_attention stored, streaming pair_z [B,N,N,c_pair_z] once for all heads;
 * written to out[(b*N+i)*out_row_stride + out_col_offset + h*c_pair_z + c]  (H = 8, c_pair_z = 32).
 * logits_ld: rows per (sample, head) slab and row stride of ``logits`` ([B,H,ld,ld]; 0 = n_res; the padded length when
 * s2s_ipa_attention_f16w wrote them for a ragged n_res). */
int s2s_ipa_opair(const float* logits, const float* stats, const float* pair_z, float* out, int n_samples, int n_res,
                  int n_heads, int c_pair_z, int out_row_stride, int out_col_offset, int logits_ld, void* stream);

/* ---- Rigid frames ---- */

/* Rigid.compose_q_update_vec (src/common/rigid_utils.py:1042-1066, :590-619, :268-277).
 *   rigids7 [M,7], update6 [M,6], mask [M], out7 [M,7] (may alias rigids7). */
int s2s_rigid_compose_update(const float* rigids7, const float* update6, const float* mask, float* out7,
                             long long n_frames, int update_ld, void* stream);   /* update_ld: floats between consecutive update rows (>= 6): the
                                                                                    BackboneUpdate layer's padded [n, 32] output is read in place */

/* TorsionAngleHead's normalisation (src/models/net/layers.py:199-213: u / sqrt(max(u0^2 + u1^2, eps)), normalize != 0) and / or
 * DenoisingNet's blend with the input torsion under the fixed mask (denoising_ipa.py:193-195: gt * fixed + pred * (1 - fixed),
 * gt_sin_cos != NULL: row r at gt_sin_cos + r * gt_row_stride, fixed_mask [n_rows]).  u: rows of u_ld >= 2 floats (the head's padded
 * output is read in place); out2 [n_rows, 2]. */
int s2s_torsion_head(const float* u, int u_ld, int normalize, const float* gt_sin_cos, long long gt_row_stride, const float* fixed_mask,
                     float eps, float* out2, long long n_rows, void* stream);

/* TranslationIPA scale_rigids / unscale_rigids (src/models/net/ipa.py:288-292): translation * scale,
 * or translation / scale (true division) when divide != 0. */
int s2s_rigid_scale_trans(const float* rigids7, float* out7, long long n_frames, float scale, int divide, void* stream);

/* Upload the idealised-geometry tables used by s2s_frames_to_backbone (host pointers; synchronous;
 * call once per process).  Values: src/common/residue_constants.py:775-852 via all_atom.py:13-18. */
int s2s_set_backbone_tables(const float* pos_21x5x3, const float* mask_21x5, const int* is_psi_group_21x5,
                            const float* default_frames_21x2x4x4);

/* compute_backbone (src/common/all_atom.py:141-173).  rigids7 [M,7] (Angstrom), psi_sincos [M,2],
 * aatype [M] int64 or NULL; atom14_bb5 [M,5,3] (N,CA,C,O,CB) or NULL; atom37 [M,37,3] or NULL. */
int s2s_frames_to_backbone(const float* rigids7, const float* psi_sincos, const long long* aatype, float* atom14_bb5,
                           float* atom37, long long n_frames, void* stream);

/* ---- One reverse-diffusion geometry step ---- */

/* FrameDiffuser.score + FrameDiffuser.reverse + Rigid.to_tensor_7
 * (src/models/score/frame.py:109-143, :153-210; so3.py:274-309, :333-370; r3.py:79-137).
 *   x0_7 [B,N,7] predicted frames, xt_7 [B,N,7] current frames, mask/diffuse_mask [B,N],
 *   params8 [B,8] float: sigma(bin), g_rot^2, exp(-beta/2), 1-exp(-beta), b(t), g_trans^2, g_rot, g_trans
 *   z_rot,z_trans [B,N,3] double noise (only read when probability_flow == 0),
 *   rot_score_in/trans_score_in [B,N,3] double or NULL: when given, the score stage is skipped and
 *   these are used (FrameDiffuser.reverse called with caller-provided scores; x0_7 may be NULL),
 *   next7 [B,N,7] or NULL (score only); rot_score_out/trans_score_out [B,N,3] double or NULL.
 *   center_trans: 0 = off, 1 = centre of mass over all N residues (reference behaviour), 2 = over the
 *   residues with mask > 0 only (padded mixed-length batches).
 *   dt = 1 / int(num_timesteps * T) of the trajectory (diffusion_module.py:267); dt_per_sample [B] double or NULL: one step size
 *   per sample instead, for batches that hold trajectories of different t_delta (sampler.forward_backward_deltas). */
int s2s_se3_step(const float* x0_7, const float* xt_7, const float* mask, const float* diffuse_mask,
                 const float* params8, const double* z_rot, const double* z_trans,
                 const double* rot_score_in, const double* trans_score_in, float* next7,
                 double* rot_score_out, double* trans_score_out, int n_samples, int n_res, double dt,
                 const double* dt_per_sample, double coordinate_scaling, int probability_flow, int center_trans,
                 double noise_scale, void* stream);

/* The embedder's per-evaluation assembly in one launch (EmbeddingModule.forward, src/models/net/denoising_ipa.py:107-136: the first
 * Linear of the node MLP and of the edge MLP on [timestep embedding | fixed-mask column | positional block]).
 *   t_img [t_img_rows, 512] = first-layer image of the timestep embedding + bias: [node MLP 256 | edge row part 128 | edge column part 128];
 *          t_img_rows = 1: one timestep for the whole chunk, = n_rows / n_res: one row per sample (trajectories of different t in a batch)
 *   node_const [node_const_rows, 256] (rows = n_rows, or n_res when every sample shares it): fixed-mask + positional terms of the node MLP
 *   fa [n_rows,128], fb (column-blocked [B,32,n_res,4] when b_col_blocked, else [n_rows,128]): fixed-mask terms of the edge MLP
 *   -> h = relu(t_img[0:256] + node_const) as packed planes (h_xp) or fp32 [n_rows,256] (h_f32; exactly one of the two),
 *      node_a = t_img[256:384] + fa, node_b = t_img[384:512] + fb: the operands of s2s_node_linear / s2s_edge_embed(_f16x3). */
int s2s_embed_assemble(const float* t_img, long long t_img_rows, const float* node_const, long long node_const_rows, const float* fa, const float* fb,
                       long long n_rows, int n_res, void* h_xp, float* h_f32, float* node_a, float* node_b, int b_col_blocked,
                       void* stream);

/* ---- Per-node dense layers (split-f16 MFMA "f16x3", fp32-equivalent; see s2s_edge_transition_f16x3) ----
 * Activations travel between these layers as PACKED PLANES ("XP"): for X [M, K],
 *   XP[rt = row/32][ks = K/16][plane 2][lane 64][8] f16, lane = 32 g + (row & 31),
 *   element j = plane of X[row][32 (ks>>1) + (r&3) + 8 (r>>2) + 4 g], r = 8 (ks&1) + j;
 * planes = the f16 pair (x_h = rn16(x), x_l = rn16(x - x_h)); the attention kernel takes the same planes.  Rows past M inside the
 * last row tile are zero. */

/* fp32 row-major x [n_rows, ld], columns col0 .. col0 + n_cols (n_cols % 32 == 0), optionally scaled per row, -> k-steps
 * xp_kstep0 .. of an XP buffer holding xp_ksteps k-steps (concatenation along K = k-step ranges). */
int s2s_pack_planes(const float* x, long long n_rows, int ld, int col0, int n_cols, void* xp, int xp_ksteps, int xp_kstep0,
                    const float* row_scale, void* stream);

/* One nn.Linear of the node stream with its surrounding elementwise ops (reference: Linear src/models/net/layers.py:64-124;
 * call sites ipa.py:131-171 (q/kv/points), :259-266 (linear_out), :343-366 (LayerNorm, skip, transformer, linear, transitions),
 * layers.py:128-145,176,188-241; torch.nn.TransformerEncoderLayer's projections and feed-forward):
 *     v = acc * pre_scale[row] + bias;  relu;  v *= pre_mask[row];  v += residual[row, col];  LayerNorm(v) over the n_out
 *     columns (gamma/beta given; needs n_out == 32 * tiles_per_block);  v *= post_mask[row]
 *   xp: packed planes of the input [n_rows, k_in]; w_packed: ops.pack_node_weight(W [n_out, k_in], tiles_per_block);
 *   outputs: out_f32[row * out_ld + out_col0 + col] and/or the packed planes of the result as k-steps out_xp_kstep0 .. of an XP
 *   buffer with out_xp_ksteps k-steps (the input format of the next layer and of s2s_ipa_attention_f16w).
 *   Any pointer may be NULL to skip that step.
 *   Row map (map_pad > 0; only with bias / relu / LayerNorm epilogues): n_rows counts OUTPUT rows = n_samples * map_pad, and output
 *   row (sample, n) reads input row sample * map_src + min(n, map_src - 1) -- the per-sample padding to whole 32-row tiles that
 *   s2s_ipa_attention_f16w wants of its q / k / v operands for a ragged n_res (map_src = n_res, map_pad = n_res rounded up to 32). */
int s2s_node_linear(const void* xp, const void* w_packed, const float* bias, long long n_rows, int k_in, int n_out,
                    int tiles_per_block, const float* pre_scale, int relu, const float* pre_mask, const float* residual,
                    int residual_ld, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* post_mask,
                    float* out_f32, int out_ld, int out_col0, void* out_xp, int out_xp_ksteps, int out_xp_kstep0,
                    int map_pad, int map_src, void* stream);

/* The same layer, same epilogue, on EXACT fp32 MFMA: x fp32 row-major [n_rows, x_ld] (its first k_in columns, k_in % 8 == 0),
 * w_packed = ops.pack_node_weight_f32(W, tiles_per_block): [n_out/(32 TG)][k_in/8][TG][64][4] fp32 in the pack_weight lane order,
 * out_f32 required.  No planes, no range limit: the node stream of the "f32" arithmetic. */
int s2s_node_linear_f32(const float* x, int x_ld, const float* w_packed, const float* bias, long long n_rows, int k_in, int n_out,
                        int tiles_per_block, const float* pre_scale, int relu, const float* pre_mask, const float* residual,
                        int residual_ld, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* post_mask,
                        float* out_f32, int out_ld, int out_col0, void* stream);

/* The same GEMM with the operands swapped, for a projection whose output is consumed as the A operand of a later product over
 * its ROWS (the value projection of InvariantPointAttention, ipa.py:132-141, consumed by the PV step): the result (+ bias) is
 * stored as MFMA A fragments of f16 pairs  out_vf[row tile (32 rows)][head][column tile (32 cols) in head][k-step u (16 rows)]
 * [plane 2][lane 64][8], element j of lane (column c, half h) = row (r&3) + 8 (r>>2) + 4 h, r = 8 u + j, of the tile.  w_packed as
 * for s2s_node_linear with tiles_per_block = 8.  map_pad / map_src: the row map of s2s_node_linear. */
int s2s_node_linear_vfrag(const void* xp, const void* w_packed, const float* bias, long long n_rows, int k_in, int n_out,
                          int tiles_per_head, void* out_vf, int map_pad, int map_src, void* stream);

/* Up to six INDEPENDENT node layers (bias / ReLU epilogues, no residual / LayerNorm / masks) in one launch -- the five projections of an
 * IPA block (linear_q, the k and v halves of linear_kv, linear_q_points, linear_kv_points: ipa.py:131-171) read the same activations
 * and nothing of each other.  Each problem is the argument list of s2s_node_linear (bias / pre_scale / relu epilogue; tiles_per_block 1, 2,
 * 4, 5, 6, 8 or 10) or, with vfrag_tiles_per_head > 0, of s2s_node_linear_vfrag.  Besides the IPA projections the trunk runs this way
 * the four skip_embed layers (they all read the embedder's output) and, per block, BackboneUpdate + the EdgeTransition's per-node
 * parts (+ the torsion head's first layer after the last block): layers of one input, one launch. */
typedef struct s2s_node_problem {
    const void* xp; const void* w_packed; const float* bias;
    long long n_rows; int k_in, n_out, tiles_per_block;
    int vfrag_tiles_per_head;
    float* out_f32; int out_ld, out_col0;
    void* out_xp; int out_xp_ksteps, out_xp_kstep0;
    void* out_vf;
    int map_pad, map_src, relu;
    const float* pre_scale;   /* [n_rows] or NULL: row scale applied to the accumulator before the bias (s2s_node_linear) */
} s2s_node_problem;
int s2s_node_linear_multi(const s2s_node_problem* problems, int n_problems, void* stream);

/* A chain of 2 .. 4 layers of one output width (256 or 320) in one launch: every layer but the last is relu?(W x + b), the last one
 * has the epilogue and outputs of s2s_node_linear.  The hidden activations stay in registers (the accumulator layout of a layer is the
 * operand layout of the next); bit for bit the separate launches.  w_packed: ops.pack_node_weight(W, width / 32).
 * The FIRST layer may contract over k_in0 != width columns (320 -> 256) and may add a residual (mid_residual [n_rows, ld]) and store its
 * fp32 result (mid_out_f32 [n_rows, ld]) -- which may then be the last layer's ``residual``: trunk.linear + NodeTransition
 * (src/models/net/ipa.py:358-359, layers.py:128-145) as one launch.  The first layer may also carry a LayerNorm of its own behind that
 * residual (mid_ln_gamma / mid_ln_beta [width], mid_ln_eps; NULL = none): an nn.TransformerEncoderLayer's post-attention half --
 * out_proj + residual + norm1, linear1, relu, linear2 + residual (= the stored norm1 output) + norm2 (ipa.py:312-317) -- as one launch.
 * Also: the embedder's node MLP (denoising_ipa.py:113-120). */
typedef struct s2s_chain_layer { const void* w_packed; const float* bias; int relu; } s2s_chain_layer;
int s2s_node_chain(const void* xp, const s2s_chain_layer* layers, int n_layers, long long n_rows, int width, int k_in0,
                   const float* mid_residual, int mid_residual_ld, float* mid_out_f32, int mid_out_ld, const float* mid_ln_gamma,
                   const float* mid_ln_beta, float mid_ln_eps, const float* pre_mask, const float* residual, int residual_ld, const float* ln_gamma, const float* ln_beta, float ln_eps,
                   const float* post_mask, float* out_f32, int out_ld, int out_col0, void* out_xp, int out_xp_ksteps,
                   int out_xp_kstep0, void* stream);

/* The LayerNorm (+ post mask) half of a node layer on its own: fp32 rows x [n_rows, x_ld] (n_cols = 256 or 320) -> out_f32 and / or packed
 * planes, with the epilogue code of s2s_node_linear -- a layer run as s2s_node_linear(ln = NULL, out_f32 = x) followed by this call
 * equals the fused layer bit for bit.  For long contractions on few rows (linear_out, K = 2688, ipa.py:259-266): the GEMM can then run
 * in narrow column blocks instead of one block per row tile. */
int s2s_row_layernorm(const float* x, int x_ld, long long n_rows, int n_cols, const float* ln_gamma, const float* ln_beta, float ln_eps,
                      const float* post_mask, float* out_f32, int out_ld, int out_col0, void* out_xp, int out_xp_ksteps,
                      int out_xp_kstep0, void* stream);

/* Self-attention core of the trunk's TransformerEncoderLayer (src/models/net/ipa.py:312-317,357; torch.nn.MultiheadAttention with
 * d_model = n_heads * head_dim, head_dim = 80): softmax(q k^T / sqrt(head_dim) + key_bias[j]) v per (sample, head), exact fp32 MFMA.
 *   qkv [B*N, 3*D] fp32 = in_proj output (q | k | v); key_bias [B,N] or NULL: added to the logits of key j (PyTorch's float
 *   key-padding-mask semantics; -inf removes a key); out_f32 [B*N, D] and/or out_xp = packed planes of it (see above). */
int s2s_encoder_attention(const float* qkv, const float* key_bias, float* out_f32, void* out_xp, int n_samples, int n_res,
                          int n_heads, int head_dim, void* stream);

/* The same operator on split-f16 MFMA ("f16x3", the default arithmetic): q, k, v and the probabilities as f16 pairs, three products
 * per block, fp32 softmax and accumulation; q, k, v feed the range guard.  Same arguments. */
int s2s_encoder_attention_f16x3(const float* qkv, const float* key_bias, float* out_f32, void* out_xp, int n_samples, int n_res,
                          int n_heads, int head_dim, void* stream);

/* ---- Forward process / prior, once per trajectory ---- */

/* FrameDiffuser.forward_marginal (src/models/score/frame.py:36-107; so3.py:244-272, :315-331, :13-19; r3.py:49-74) or, with
 * rigids0_4x4 == NULL, FrameDiffuser.sample_prior (frame.py:212-255), followed by Rigid.to_tensor_7.
 *   rigids0_4x4 [B,N,4,4] starting frames (Angstrom) or NULL; noise drawn by the caller: z_axis [B,N,3] ~ N(0,1) (rotation
 *   axis), u01 [B,N] ~ U[0,1) (inverse CDF of the IGSO(3) angle), z_trans [B,N,3] ~ N(0,1);
 *   cdf_rows [R,n_omega] double (so3.py:185-187) and cdf_row_of_sample [B] (row per sample = sigma bin of its t),
 *   omega_grid [n_omega] (SO3Diffuser.discrete_omega); params2 [B,2] = exp(-marginal_b_t/2), sqrt(1-exp(-marginal_b_t));
 *   diffuse_mask [B,N] or NULL (= all ones); out rigids_t7 [B,N,7]. */
int s2s_forward_marginal(const float* rigids0_4x4, const float* z_axis, const float* u01, const float* z_trans,
                         const double* cdf_rows, const int* cdf_row_of_sample, const float* omega_grid, int n_omega,
                         const float* params2, const float* diffuse_mask, float coordinate_scaling, float* rigids_t7,
                         int n_samples, int n_res, void* stream);

/* ---- Ensemble metrics on the device (src/metrics/metrics.py) over CA coordinates [n, L, 3] float32 ---- */

/* Per sample: CA pairs (|i-j| > k_exclusion) closer than clash_bar (metrics.py:80-105), the largest adjacent CA-CA distance
 * (:12-23; bonding_validity :124-137 compares it with the reference ensemble's), the radius of gyration (:53-77, float64). */
int s2s_ca_sample_stats(const float* ca, int n_samples, int n_res, float clash_bar, int k_exclusion, int* n_clash,
                        float* adjacent_max, double* radius_of_gyration, void* stream);

/* pairwise_distance_ca (metrics.py:38-50): out [n_samples, D] float32, D = (L-offset)(L-offset+1)/2 upper-triangular CA distances per sample
 * in np.triu_indices(L, k=offset) order, in numpy's float32 arithmetic (bit for bit): the features of js_tica (:166-200).  n_samples <= 65535. */
int s2s_ca_pairwise_distances(const float* ca, int n_samples, int n_res, int offset, float* out, void* stream);

/* js_pwd (metrics.py:140-166): per pair channel (i, j >= i + offset; np.triu_indices order) the Jensen-Shannon distance between
 * the n_bins-bin histograms (range = the reference ensemble's [min, max], numpy's float32 bin arithmetic, + pseudo_count) of the
 * predicted and the reference ensemble -> js_per_channel [(L-offset)(L-offset+1)/2] float64 (the metric is their mean).
 * ref_weights [n_ref] / pred_weights [n_pred]: per-sample float64 histogram weights (the reference's `weights=`, metrics.py:139-150;
 * NULL = all ones, both NULL = integer counts). */
int s2s_ca_pwd_js(const float* ref_ca, int n_ref, const float* pred_ca, int n_pred, int n_res, int offset, int n_bins,
                  double pseudo_count, double* js_per_channel, const double* ref_weights, const double* pred_weights, void* stream);

/* ---- PDB text at the exit of the path (HOST pointers, host code; byte-identical to the reference's writers) ---- */

/* protein.to_pdb per model (src/common/protein.py:152-234) over atom37 [n_models, n_res, 37, 3] float32 HOST coordinates with
 * the atom mask of pdb_utils.atom37_to_pdb (src/common/pdb_utils.py:233: sum |xyz| > 1e-7; GLY CB skipped).
 *   aatype / residue_index / chain_index [n_res] int64 or NULL (defaults of protein_with_default_params, pdb_utils.py:175-203:
 *   ALA, 1..n_res, chain 0); b_factors [n_res,37] double or NULL (zeros); MODEL numbers first_model_number + m;
 *   add_end: 0 = none, 1 = an "END" line after every model (to_pdb(add_end=True)), 2 = one bare "END" without newline after
 *   the last model (atom37_to_pdb).  Returns the number of bytes of text; writes them when out != NULL and they fit in
 *   out_capacity (call with out = NULL to size the buffer).  < 0: -1 bad argument / aatype > 20, -2 more than 62 chains. */
long long s2s_format_pdb_models(const float* atom37, int n_models, int n_res, const long long* aatype,
                                const long long* residue_index, const long long* chain_index, const double* b_factors,
                                int first_model_number, int add_end, char* out, long long out_capacity);

/* The same text streamed to a file in bounded blocks of models (append != 0: append to an existing file).
 * Returns bytes written, -3 cannot open, -4 write error. */
long long s2s_write_pdb_models(const char* path, int append, const float* atom37, int n_models, int n_res,
                               const long long* aatype, const long long* residue_index, const long long* chain_index,
                               const double* b_factors, int first_model_number, int add_end);

/* pdb_utils.merge_pdbfiles (src/common/pdb_utils.py:31-83): MODELs of the inputs, in order, renumbered from 1. */
long long s2s_merge_pdb_files(const char* const* paths, int n_paths, const char* out_path);

/* ---- Host noise stream (parity mode) ----
 * Discard n_outputs 32-bit outputs of the Mersenne twister behind torch's CPU generator: state624 = the engine's 624 words in 64-bit
 * slots, *left / *next = its counters, as torch.get_rng_state() serialises them.  The reference draws -- and, under the probability-flow
 * ODE, never uses -- two float64 normal tensors of the whole chunk per denoise step (src/models/score/so3.py:360, src/models/score/r3.py:109);
 * a float64 normal tensor of n >= 16 elements costs 2 (n + (n % 16 ? 16 : 0)) engine outputs, and this call leaves the generator where
 * those draws leave it without computing them (host code; returns 0, or 1 for fields outside the engine's ranges). */
int s2s_mt19937_discard(unsigned long long* state624, int* left, unsigned long long* next, unsigned long long n_outputs);

#ifdef __cplusplus
}
#endif
#endif /* STR2STR_HIP_H */
