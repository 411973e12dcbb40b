// Invariant Point Attention core on f16 pair operands, ONE WAVE PER QUERY TILE (gfx950): the default attention kernel, any length.
// Reference: InvariantPointAttention.forward, src/models/net/ipa.py:183-257 (same operator as csrc/ipa_attention.hip, the exact
// fp32-operand kernel, which documents the MFMA orientation and the pair-term split).
//
// Every matrix operand arrives ALREADY split into f16 pairs (x_h, x_l) in MFMA fragment order, written by the epilogues of the
// producing GEMMs (csrc/node_gemm.hip) and by the point kernel below, so this kernel contains no operand split except the 16
// probabilities of a key tile, and all products run on v_mfma_f32_32x32x16_f16 (a_h b_h + a_h b_l + a_l b_h, fp32 accumulate =
// fp32-equivalent):
//   S^T[j, i]  = K[j, :] . Q[i, :]  + K'pts[j, :] . Q'pts[i, :]     18 k-steps: 16 of the head's channels + 2 of point coordinates
//   O^T[c, i] += V^T[c, j] P^T[j, i]                                10 output tiles: 8 of channels + 2 of (x, y, z, 0) value points
// The point term  -1/2 w_h sum_p |q_ip - k_jp|^2  =  w_h q.k  - 1/2 w_h |q_i|^2 - 1/2 w_h |k_j|^2 : the cross term rides in the
// QK^T accumulator (the query points are pre-scaled by w_h / c1, c1 = sqrt(1/(3C)) the scalar-logit scale), the two squared
// norms are per-residue fp32 scalars from the point kernel.  (ipa_attention.hip forms explicit differences on the VALU; the
// expansion costs ~1e-6 absolute in a logit at protein coordinates / 10 -- see DESIGN.md -- and removes 900 VALU
// instructions per key tile.)
//   q_xp, k_xp : packed planes [row tile][16 H k-steps][2][64][8] of the q / k projections (s2s_node_linear out_xp)
//   v_vf       : [row tile][H][8 col tiles][2][2][64][8] A fragments of the v projection (s2s_node_linear_vfrag)
//   qp_xp, kp_xp [row tile][H][2][2][64][8], vp_vf [row tile][H][2][2][2][64][8], q2 / k2 [row tile][H][32]: the point kernel
// With two planes per operand the 18 query fragments of a tile are 144 VGPRs, so a wave owns a whole 32-residue query tile -- all
// 18 k-steps of S^T and all 10 output tiles -- and a workgroup is four query tiles of one (sample, head): no partial-sum exchange,
// and a K / V image (36 / 40 KiB, contiguous in HBM) serves four query tiles.  The head output leaves as packed planes of
// linear_out's input (the accumulator registers ARE its fragments).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "geom.h"
#include "range_flag.h"
#include "str2str_hip.h"

namespace {

using namespace s2s;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // a 16 B fragment

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ void load7(const float* __restrict__ p, Quat<float>& q, Vec3<float>& t) {
    q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3];
    t.x = p[4]; t.y = p[5]; t.z = p[6];
}

// e^x for x <= ~0 in 6 VALU instructions: v_exp_f32 (2^t, 1 ulp) on t = fl(x log2 e), corrected to first order for the rounding
// of the product and of the constant (exact residual by FMA), so the argument error does not grow with |x|.  Relative error
// ~2 ulp; expf() expands to ~12 instructions with many temporaries (the softmax section was the register-pressure peak).
__device__ __forceinline__ float exp_neg(float x) {
    const float L2E = 1.44269504088896341f, L2E_LO = 1.92596299112661746e-8f, LN2 = 0.693147180559945309f;
    const float t = x * L2E;
    float e = __builtin_fmaf(x, L2E, -t);
    e = __builtin_fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * LN2, r);
}

// two values -> element pair `at / 2` of the planes (x_h = rn16(x), x_l = rn16(x - x_h)) in 3 instructions: v_cvt_pk_f16_f32 for both
// x_h, then ONE fused multiply-add per value that reads x_h as f16 and rounds to f16 -- (-x_h) * 1.0 + x, whose exact fp32 result is
// the difference -- the bits of convert back, subtract, convert (pair_mlp_f16.hip split2_f16; hipcc's expansion of the C form: 5 per value)
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2(float x0, float x1, f16x8& ph, f16x8& pl, int at) {
    unsigned hh, ll;
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, -%0, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, -%0, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hh), "=&v"(ll)
        : "v"(x0), "v"(x1));
    u32x4p hv = __builtin_bit_cast(u32x4p, ph), lv = __builtin_bit_cast(u32x4p, pl);
    hv[at / 2] = hh;
    lv[at / 2] = ll;
    ph = __builtin_bit_cast(f16x8, hv);
    pl = __builtin_bit_cast(f16x8, lv);
}
__device__ __forceinline__ void split8(const float* v, f16x8& ph, f16x8& pl) {   // x_h, x_l
#pragma unroll
    for (int j = 0; j < 8; j += 2) split2(v[j], v[j + 1], ph, pl, j);
}

// ------------------------------------------------------------------------------------------------------------------------
// Point generation (ipa.py:144-171, rigid_utils.py:1107-1120) straight into MFMA fragments.  One workgroup per (row tile of 32
// residues, head): the 8 + 8 + 12 global-frame points of every residue go through LDS, then 512 (fragment, lane) items are
// split and stored.  Coordinate k of a point row = 3 p + d (24 of the 32 columns of two k-steps; the rest zero).
constexpr int PQ = 8, PV = 12;

__global__ void __launch_bounds__(256) ipa_prep_f16_kernel(const float* __restrict__ rig, const float* __restrict__ qp_lin,
                                                              const float* __restrict__ kvp_lin, const float* __restrict__ head_w,
                                                              float q_scale_c1, f16x8* __restrict__ qp_xp, f16x8* __restrict__ kp_xp,
                                                              f16x8* __restrict__ vp_vf, float* __restrict__ q2, float* __restrict__ k2,
                                                              int H, int n_res, int n_pad, int* range_flag,
                                                              const f16x8* __restrict__ s_xp, f16x8* __restrict__ k_sh, f16x8* __restrict__ v_sh) {
    // Row tiles are tiles of the PADDED residue range (n_pad = n_res rounded up to 32, per sample): tile rt = sample rt / (n_pad / 32);
    // residues >= n_res of the last tile are padding: zero points, q2 = 0, k2 = -1e9 (such a key can never carry probability).
    __shared__ float sq[32][33], sk[32][33], sv[32][65];
    __shared__ int s_pad[32];
    __shared__ _Float16 s_t[2][32][34];
    const int tid = threadIdx.x;
    const int head = blockIdx.x % H;
    const long long rt = blockIdx.x / H;
    const float hw = head_w[head];
    const int tps = n_pad / 32;
    // ---- folded projections (ops.fold_ipa_weights): the K and V operands of EVERY head are the block's input s itself (c_s = 256 = 16
    // k-steps = 8 column tiles; H = 8): K image = the packed planes of s -- s_xp as it is when n_res % 32 == 0, else gathered here into
    // the per-sample padded rows (k_sh; a padded row repeats the sample's last row: finite, and its k2 is -1e9) --, V image = the same
    // values as A fragments [row tile][8 column tiles][2][2][64][8] (v_sh; node_gemm.hip's VF layout with one head): a 32 x 32
    // transposition per (row tile, column tile), and workgroup (row tile, head) does column tile `head`.  Exact: f16 planes are moved.
    if (s_xp) {
        const int row = tid & 31, qd = tid >> 5, ksb = qd & 1, g = (qd >> 1) & 1, plane = qd >> 2;
        const long long smp = rt / tps;
        const int n = (int)(rt - smp * tps) * 32 + row;
        const long long srow = smp * n_res + (n < n_res ? n : n_res - 1);
        const f16x8 v = s_xp[(((srow >> 5) * 16 + 2 * head + ksb) * 2 + plane) * 64 + 32 * g + (int)(srow & 31)];
        if (k_sh) k_sh[((rt * 16 + 2 * head + ksb) * 2 + plane) * 64 + 32 * g + row] = v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rr = 8 * ksb + j;
            s_t[plane][row][(rr & 3) + 8 * (rr >> 2) + 4 * g] = v[j];
        }
        __syncthreads();
        const int u = tid >> 7, pl = (tid >> 6) & 1, lane = tid & 63;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = s_t[pl][rowmap(8 * u + j, lane >> 5)][lane & 31];
        v_sh[((((rt * 8) + head) * 2 + u) * 2 + pl) * 64 + lane] = o;
    }
    {
        const int row = tid >> 3, p = tid & 7;
        const long long smp = rt / tps;
        const int n = (int)(rt - smp * tps) * 32 + row;
        const bool pad = n >= n_res;
        if (p == 0) s_pad[row] = pad;
        const long long r = smp * n_res + (pad ? n_res - 1 : n);
        Quat<float> q; Vec3<float> t;
        load7(rig + r * 7, q, t);
        const Mat3<float> R = quat_to_rot<float>(q);
        const int HPq = H * PQ, HPkv = H * (PQ + PV);
        const float* ql = qp_lin + r * 3 * HPq;
        const float* kl = kvp_lin + r * 3 * HPkv;
        {
            const int w = head * PQ + p;
            const Vec3<float> g = rot_vec_mul<float>(R, Vec3<float>{ql[w], ql[HPq + w], ql[2 * HPq + w]});
            sq[row][3 * p] = pad ? 0.f : g.x + t.x; sq[row][3 * p + 1] = pad ? 0.f : g.y + t.y; sq[row][3 * p + 2] = pad ? 0.f : g.z + t.z;
            sq[row][24 + p] = 0.f;
        }
        {
            const int c = head * (PQ + PV) + p;
            const Vec3<float> g = rot_vec_mul<float>(R, Vec3<float>{kl[c], kl[HPkv + c], kl[2 * HPkv + c]});
            sk[row][3 * p] = pad ? 0.f : g.x + t.x; sk[row][3 * p + 1] = pad ? 0.f : g.y + t.y; sk[row][3 * p + 2] = pad ? 0.f : g.z + t.z;
            sk[row][24 + p] = 0.f;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int pv = p + 8 * x;  // value-point slot 0..15; slots >= PV are zero padding
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv < PV && !pad) {
                const int c = head * (PQ + PV) + PQ + pv;
                const Vec3<float> g = rot_vec_mul<float>(R, Vec3<float>{kl[c], kl[HPkv + c], kl[2 * HPkv + c]});
                o = make_float4(g.x + t.x, g.y + t.y, g.z + t.z, 0.f);
            }
            sv[row][4 * pv] = o.x; sv[row][4 * pv + 1] = o.y; sv[row][4 * pv + 2] = o.z; sv[row][4 * pv + 3] = o.w;
        }
    }
    __syncthreads();
    if (tid < 64) {  // squared norms, fixed summation order
        const int row = tid & 31;
        const float (*s)[33] = tid < 32 ? sq : sk;
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < 24; ++x) acc += s[row][x] * s[row][x];
        (tid < 32 ? q2 : k2)[(rt * H + head) * 32 + row] = s_pad[row] ? (tid < 32 ? 0.f : -1.0e9f) : -0.5f * hw * acc;
    }
    const float qs = hw / q_scale_c1;
    float amax = 0.f;   // range guard (range_flag.h): the point coordinates are split into f16 planes here
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + 256 * it;  // 0..127 q, 128..255 k, 256..511 value points
        const int lane = item & 63, g = lane >> 5, c = lane & 31;
        float v[8];
        f16x8* dst;
        if (item < 256) {
            const int ks = (item >> 6) & 1;
            const bool isq = item < 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = isq ? sq[c][16 * ks + 8 * g + j] * qs : sk[c][16 * ks + 8 * g + j];
            dst = (isq ? qp_xp : kp_xp) + (((rt * H + head) * 2 + ks) * 2) * 64 + lane;
        } else {
            const int f = (item - 256) >> 6, ct = f >> 1, u = f & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sv[rowmap(8 * u + j, g)][32 * ct + c];
            dst = vp_vf + ((((rt * H + head) * 2 + ct) * 2 + u) * 2) * 64 + lane;
        }
        f16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = s2s::range_max(amax, v[j]);
        split8(v, ph, pl);
        dst[0] = ph; dst[64] = pl;
    }
    s2s::range_report(range_flag, amax, s2s::kRangeIpaPoints);
}

// ------------------------------------------------------------------------------------------------------------------------
struct PlaneArgs {
    const f16x8* q_xp; const f16x8* k_xp; const f16x8* v_vf;
    const f16x8* qp_xp; const f16x8* kp_xp; const f16x8* vp_vf;
    const float* q2; const float* k2;
    const float* attn_bias;  // [B,H,N,N]
    float* logits;           // [B,H,NP,NP] (may alias attn_bias when NP == N)
    float* stats;            // [B,H,N,2]
    const float* mask;       // [B,N]
    const float* rigids7;    // [B,N,7]
    float* out;              // [B,N,feat] fp32: only the o_pt columns are written here
    f16x8* out_xp;          // packed planes of the [B*N, 16*xp_ksteps] linear_out input: the o columns (k-steps 16 head ..)
    int xp_ksteps;
    int B, N, H;
    int NP;                  // N rounded up to the 32-residue tiles: the fragment arrays, q2 / k2 and the logits use NP rows per sample
    float inf, eps;
    int xcd_remap;
    int HKV;                 // heads of the K / V fragment arrays: H, or 1 = one image serves every head (folded projections)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void dma_kib(const f16x8* src_piece, f16x8* lds_piece, int lane) {
    // one 1 KiB fragment: 16 B per lane, destination = wave-uniform base + lane * 16 (LDS-DMA semantics)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src_piece + lane), (lds_ptr_t)lds_piece, 16, 0, 0);
}

typedef __attribute__((address_space(3))) const f16x8 lds_frag;
__device__ __forceinline__ lds_frag* frag_pin(const f16x8* p) {
    lds_frag* q = (lds_frag*)p;
    asm volatile("" : "+v"(q));
    return q;
}

// Timeline probe (tools/ipa_f16w_probe.py; only in -DS2S_IPA_PROBE=<block> builds)
#ifdef S2S_IPA_PROBE
__device__ unsigned long long s2s_ipa8_probe[4][128];
#define IPROBE(idx) do { if (blockIdx.x == S2S_IPA_PROBE) s2s_ipa8_probe[threadIdx.x >> 6][idx] = __builtin_readcyclecounter(); } while (0)
#else
#define IPROBE(idx) do { } while (0)
#endif

constexpr int KQ = 18;   // k-steps of QK^T: 16 channels + 2 point coordinates
constexpr int KH = KQ;       // every wave runs the whole contraction
constexpr int OT = 10;   // output tiles of PV: 8 channels + 2 value points
constexpr int OH = OT;       // ... and owns all ten output tiles
constexpr int CT = 8;

// A workgroup = 4 waves = 4 query tiles (32 residues each) of one (sample, head); wave w holds the query fragments of its tile
// (B operands must sit in VGPRs: hipcc allocates MFMA sources there only) and the accumulators of its ten output tiles.
//
// Two phases per work item (sample, head, 128 query residues) instead of an online softmax:
//   phase 1, key tiles 0 .. NT-1:  S^T -> masked logits -> global (the [B,H,N,N] buffer s2s_ipa_opair reads anyway), row maximum
//   phase 2, key tiles 0 .. NT-1:  logits back from L2, p = exp(s - max), row sum, O^T += V^T P^T
// The accumulators are never rescaled (a VALU pass over 80 matrix-core registers per tile, which also dragged the whole
// register allocation into accvgpr copies), the query fragments are dead in phase 2, and the K images (phase 1) and V images
// (phase 2) do not coexist in LDS: one stream of images through two 60 KiB buffers, image g in buffer g & 1.  The logit
// arithmetic of tile t-1 rides between the MFMAs of tile t, the exp / split of tile t+1 between those of tile t (one wave per
// SIMD: nothing else fills the matrix pipe's shadow).
//
// Image g goes global -> VGPR (one step before it is written) -> LDS (one step before it is read), 9 / 10 pieces of 1 KiB per
// wave, one "copy slot" (ds_write of a piece + re-load of its register) per few MFMAs.  LDS-DMA (global_load_lds_dwordx4) would
// need no registers but costs the issuing wave ~150 cycles per piece on this part -- as much per key tile as the tile's MFMAs.
// Every VMEM operation of the loops is UNCONDITIONAL (a piece that does not exist re-loads / re-stores piece 8): vmcnt
// counts in order, and a load issued on one side of a branch makes hipcc fall back to s_waitcnt vmcnt(0) at every older use.
//
// Workgroups are persistent: the image stream runs across work items (images 2 NT, 2 NT + 1 of an item are images 0, 1 of the
// next one), so only the first item of a workgroup pays the cold start (14 k cycles = 13 % of an item before this).
struct PlaneStage {
    f16x8 img[2][OT * 2 * 2 * 64];   // 2 x 40 KiB: K image [k-step 18][plane 2][lane] (36 KiB) or V image [tile 10][u][plane 2][lane]
    __attribute__((aligned(16))) float k2[4][32];                  // per-key scalars of key tile t in slot t % 4 (written two tiles ahead, read one tile late)
    __attribute__((aligned(16))) float km[4][32];
};

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16 B load that bypasses the (non-coherent) vector L1 (cache policy sc0 | sc1): the logits were stored by the partner wave, to
// lines this CU read as attention bias a moment ago.
__device__ __forceinline__ f32x4v load_l2(__amdgpu_buffer_rsrc_t rsrc, int byte_offset) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_offset, 0, 17);
    return __builtin_bit_cast(f32x4v, r);
}

// RAGGED (n_res % 32 != 0): operands arrive in the padded layout (NP = n_res rounded up to 32 rows per sample; padded key rows are
// finite, their k2 is -1e9, see ipa_prep_f16_kernel), the bias keeps its [B,H,N,N] strides (rows of the last tile may be read past
// their end: finite neighbours or 0 outside the slab -- the -1e9 of the key decides), the logits go to a SEPARATE [B,H,NP,NP] buffer
// (a padded row's stores never touch a neighbour), and every result row is written in the real-row layout, padded rows skipped.
template <bool RAGGED>
__global__ void __launch_bounds__(256) ipa_attention_f16w_kernel(PlaneArgs a) {
    __shared__ __attribute__((aligned(16))) PlaneStage st;
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, H = a.H;
    const int HKV = a.HKV;
    const int NP = RAGGED ? a.NP : N;             // logits stride / padded rows per sample
    const int NT = NP / 32;                       // key tiles = row tiles per sample
    const int n_qb = (NT + 3) / 4;
    const int n_items = a.B * H * n_qb;
    const float c1 = sqrtf(1.0f / (3 * 256));
    const float c2 = sqrtf(1.0f / 3);

    struct Item { int b, head, qb; };
    auto decode = [&](int item) -> Item {
        // Workgroup w runs on XCD w % 8 (observed dispatch order; a speed assumption only) and gridDim.x is a multiple of 8: give
        // every XCD (private L2) a contiguous range of logical ids, so the query blocks of one (sample, head) -- which read the
        // same K / V images -- run side by side on one XCD.
        int bid = item;
        if (a.xcd_remap) bid = (bid & 7) * (n_items >> 3) + (bid >> 3);
        Item it;
        it.qb = bid % n_qb; bid /= n_qb;
        it.head = bid % H;
        it.b = bid / H;
        return it;
    };
    int item = blockIdx.x;
    Item cur = decode(item);
    Item nxt = item + (int)gridDim.x < n_items ? decode(item + (int)gridDim.x) : cur;

    // ---- the image stream of an item: g < NT: K image of key tile g (48 pieces of k_xp + 6 of kp_xp); g < 2 NT: V image of tile
    // g - NT (48 of v_vf + 12 of vp_vf); g = 2 NT, 2 NT + 1: images 0, 1 of the next item.  Piece p of this wave: p < 12: piece
    // wave + 4 p of the first array, else piece wave + 4 (p - 12) of the second.  All of this is wave-uniform (scalar).
    auto piece_ok = [&](int gg, int p) -> bool {   // K images have 36 pieces (9 per wave), V images 40 (10 per wave)
        return gg >= NT || p < 9;
    };
    // sources as buffer resources over the whole arrays (the host checks that they are < 4 GiB): the per-piece offset is a scalar,
    // the per-lane part (lane * 16) one constant VGPR -- no vector address arithmetic in a copy slot
    const long long n_rt = (long long)a.B * NT;
    const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc((void*)a.k_xp, 0, (int)(n_rt * 16 * HKV * 2048), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_kp = __builtin_amdgcn_make_buffer_rsrc((void*)a.kp_xp, 0, (int)(n_rt * H * 4096), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc((void*)a.v_vf, 0, (int)(n_rt * HKV * 32768), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_vp = __builtin_amdgcn_make_buffer_rsrc((void*)a.vp_vf, 0, (int)(n_rt * H * 8192), 0x00020000);
    const int lane16 = lane * 16;
    // piece p of this wave: p < 8: piece wave + 4 p of the first array (32 KiB), else piece wave + 4 (p - 8) of the second
    auto piece_load = [&](int g, int p) -> f16x8 {
        const bool nx = g >= 2 * NT;
        const int gg = nx ? g - 2 * NT : g;
        const int bb = nx ? nxt.b : cur.b, hh = nx ? nxt.head : cur.head;
        const bool isk = __builtin_amdgcn_readfirstlane(gg < NT);   // (everything here is wave-uniform; keep it on the SALU even
        const unsigned rt = (unsigned)(bb * NT + (isk ? gg : gg - NT));      //  when the allocator parked an input in a VGPR)
        const int pm = __builtin_amdgcn_readfirstlane(piece_ok(gg, p) ? p : 8);
        u32x4 r;
        if (p < 8) {
            const int hk = HKV == 1 ? 0 : hh;
            const unsigned off = (isk ? (rt * (16 * HKV) + 16 * hk) * 2 : (rt * HKV + hk) * 32) * 1024u + (wave + 4 * pm) * 1024u;
            r = __builtin_amdgcn_raw_buffer_load_b128(isk ? r_k : r_v, lane16, __builtin_amdgcn_readfirstlane((int)off), 0);
        } else {
            const unsigned off = (rt * H + hh) * (isk ? 4u : 8u) * 1024u + (wave + 4 * (pm - 8)) * 1024u;
            r = __builtin_amdgcn_raw_buffer_load_b128(isk ? r_kp : r_vp, lane16, __builtin_amdgcn_readfirstlane((int)off), 0);
        }
        return __builtin_bit_cast(f16x8, r);
    };
    // cold start only (LDS-DMA wants a flat address)
    auto piece_src = [&](int g, int p) -> const f16x8* {
        const bool isk = g < NT;
        const long long rt = (long long)cur.b * NT + (isk ? g : g - NT);
        const int pm = piece_ok(g, p) ? p : 8;
        const int hk = HKV == 1 ? 0 : cur.head;
        if (pm < 8)
            return (isk ? a.k_xp + ((rt * (16 * HKV) + 16 * hk) * 2) * 64 : a.v_vf + ((rt * HKV + hk) * 32) * 64) + (wave + 4 * pm) * 64;
        return (isk ? a.kp_xp + ((rt * H + cur.head) * 4) * 64 : a.vp_vf + ((rt * H + cur.head) * 8) * 64) + (wave + 4 * (pm - 8)) * 64;
    };
    auto piece_dst = [&](int g, int p) -> f16x8* {
        const int gg = g >= 2 * NT ? g - 2 * NT : g;
        const int pm = piece_ok(gg, p) ? p : 8;
        return st.img[g & 1] + (pm < 8 ? wave + 4 * pm : 32 + wave + 4 * (pm - 8)) * 64;
    };
    f16x8 stg[10];
    auto stage_load = [&](int g, int p) { stg[p] = piece_load(g, p); };
    // one copy slot: piece p of image g leaves its register for LDS, the register is refilled with piece p of image g + 1
    auto stage_slot = [&](int g, int p) {
        if (p < 10) {
            piece_dst(g, p)[lane] = stg[p];
            stage_load(g + 1, p);
        }
    };
    // per-key scalars of a key tile (k2 and the key mask, 32 floats each) reach LDS slot t & 3 two steps before tile t's logits
    // are formed; the value is loaded a step before it is stored (waves 2, 3 store; every wave loads: no conditional VMEM).
    auto small_load = [&](const Item& it, int t) -> float {
        const int tc = min(t, NT - 1);
        if constexpr (RAGGED) {
            const int kk = tc * 32 + c;
            const float v = (wave & 1) ? a.k2[(((long long)it.b * NT + tc) * H + it.head) * 32 + c] : a.mask[(long long)it.b * N + min(kk, N - 1)];
            return ((wave & 1) || kk < N) ? v : 0.f;
        }
        return (wave & 1) ? a.k2[(((long long)it.b * NT + tc) * H + it.head) * 32 + c] : a.mask[(long long)it.b * N + tc * 32 + c];
    };
    auto small_store = [&](int t, float v) {
        if (wave >= 2 && lane < 32 && t < NT) ((wave & 1) ? st.k2 : st.km)[t & 3][lane] = v;
    };
    float sm_val;

    // ---- cold start: image 0 straight into LDS, image 1 into the staging registers
    IPROBE(126);
#pragma unroll
    for (int p = 0; p < 9; ++p)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(piece_src(0, p) + lane), (lds_ptr_t)piece_dst(0, p), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < 10; ++p) stage_load(1, p);
    small_store(0, small_load(cur, 0));
    small_store(1, small_load(cur, 1));
    sm_val = small_load(cur, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    IPROBE(127);

    // ---- per-item state of a lane; the query fragments, scalars and the first bias tile of the NEXT item are fetched before the
    // epilogue of the current one (the registers are free there and the epilogue covers the HBM latency)
    struct LaneItem { long long rt_q, row_i, brow0; int i; float q2, mask; };
    auto lane_item = [&](const Item& it) -> LaneItem {
        const int qt = it.qb * 4 + wave;             // this wave's query tile within the sample
        // odd tile count: the second pair of the last workgroup has no tile of its own; it repeats the first pair's (identical
        // values to identical addresses) so that no memory operation is conditional
        const int qtc = qt < NT ? qt : NT - 1;
        LaneItem L;
        L.rt_q = (long long)it.b * NT + qtc;
        L.i = qtc * 32 + c;                          // (RAGGED: index in the padded range; >= N for a padded query row)
        const int ic = RAGGED ? min(L.i, N - 1) : L.i;
        L.row_i = (long long)it.b * N + ic;
        L.brow0 = (((long long)it.b * H + it.head) * N + ic) * N + 4 * h;
        L.q2 = a.q2[(L.rt_q * H + it.head) * 32 + c];
        L.mask = a.mask[L.row_i];
        if constexpr (RAGGED) L.mask = L.i < N ? L.mask : 0.f;
        return L;
    };
    f16x8 qf[KH][2];
    // RAGGED: the bias rows are read through a bounds-checked resource (a row of the last key tile runs past its end, the last
    // row of the array past the allocation) at dword alignment
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.attn_bias, 0, RAGGED ? (int)((long long)a.B * H * N * N * 4) : 0, 0x00020000);
    auto bias_load = [&](long long brow, int t, int g) -> float4 {
        if constexpr (RAGGED) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(r_bias, (int)(brow * 4) + (t * 32 + 8 * g) * 4, 0, 0);
            return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        } else {
            return *reinterpret_cast<const float4*>(a.attn_bias + brow + t * 32 + 8 * g);
        }
    };
    float4 bias_cur[4], bias_prev[4];
    float4 xkeep[4];   // S^T of the previous key tile (accumulator layout), consumed one step later
    // this wave's query fragments (B operands): the 18 k-steps of [16 of q_xp | 2 of qp_xp], two planes each
    auto load_queries = [&](const Item& it, const LaneItem& L) {
        const f16x8* qs = a.q_xp + ((L.rt_q * (16 * H) + 16 * it.head) * 2) * 64;
        const f16x8* ps = a.qp_xp + ((L.rt_q * H + it.head) * 4) * 64;
#pragma unroll
        for (int x = 0; x < KH; ++x) {
            const int ks = x;
#pragma unroll
            for (int p = 0; p < 2; ++p) qf[x][p] = (ks < 16 ? qs + (ks * 2 + p) * 64 : ps + ((ks - 16) * 2 + p) * 64)[lane];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_cur[g] = bias_load(L.brow0, 0, g);
    };
    LaneItem Lc = lane_item(cur);
    load_queries(cur, Lc);
    float sm_n0, sm_n1, sm_n2;   // the next item's per-key scalars of tiles 0 .. 2
#ifdef S2S_IPA_PROBE
    int probe_item = 0;
#endif
    for (;;) {
#ifdef S2S_IPA_PROBE
    if (probe_item < 16) IPROBE(100 + probe_item);
    ++probe_item;
#endif
    const int b = cur.b, head = cur.head;
    const long long rt_q = Lc.rt_q, row_i = Lc.row_i, brow0 = Lc.brow0;
    const int i = Lc.i;
    const float q2_i = Lc.q2, mask_i = Lc.mask;
    float m_run = -INFINITY;

    // =========================================================== phase 1: logits and row maxima
    // Software pipeline: the logit arithmetic of tile t-1 (VALU + LDS reads) is issued between the MFMAs of tile t, one element
    // per two MFMAs, so it runs in the shadow of the matrix pipe (one wave per SIMD: nothing else would fill it).
    // this (sample, head)'s [N, N] logit slab as a buffer resource: 32-bit offsets, cache policy on the instruction
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.logits + ((long long)b * H + head) * NP * NP), 0,
                                                                           NP * NP * 4, 0x00020000);
    const int loff0 = (i * NP + 4 * h) * 4;
    f32x4v lg[4], lg1[4];   // logits of the tile whose probabilities are formed next (and of tile 1 across the phase change)
    float tmax = -INFINITY;
    float4 k2g, kmg;   // per-key scalars of the 4 keys 8g + 4h .. of the element group being evaluated
    auto logit_elem = [&](int tp, int r, const float4 (&xa)[4], float (&sl)[16]) {
        const int g = r >> 2, e = r & 3;
        if (e == 0) {
            k2g = *reinterpret_cast<const float4*>(&st.k2[tp & 3][8 * g + 4 * h]);
            kmg = *reinterpret_cast<const float4*>(&st.km[tp & 3][8 * g + 4 * h]);
        }
        const float sa = e == 0 ? xa[g].x : (e == 1 ? xa[g].y : (e == 2 ? xa[g].z : xa[g].w));
        const float bv = e == 0 ? bias_prev[g].x : (e == 1 ? bias_prev[g].y : (e == 2 ? bias_prev[g].z : bias_prev[g].w));
        const float k2v = e == 0 ? k2g.x : (e == 1 ? k2g.y : (e == 2 ? k2g.z : k2g.w));
        const float kmv = e == 0 ? kmg.x : (e == 1 ? kmg.y : (e == 2 ? kmg.z : kmg.w));
        float x = sa * c1 + c2 * bv;
        x = x + (q2_i + k2v);
        x = x + a.inf * (mask_i * kmv - 1.0f);
        sl[r] = x;
        tmax = fmaxf(tmax, x);
        if (r == 15) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (RAGGED)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(sl[4 * k]), __float_as_uint(sl[4 * k + 1]), __float_as_uint(sl[4 * k + 2]),
                                                                 __float_as_uint(sl[4 * k + 3])}, lrsrc, loff0 + tp * 128 + 32 * k, 0, 0);
                else
                    *reinterpret_cast<float4*>(a.logits + brow0 + tp * 32 + 8 * k) = make_float4(sl[4 * k], sl[4 * k + 1], sl[4 * k + 2], sl[4 * k + 3]);
            }
        }
    };
    auto step1 = [&](int t, auto have_c, auto prev_c, auto flush_c) {
        constexpr bool have = decltype(have_c)::value, prev = decltype(prev_c)::value, flush = decltype(flush_c)::value;
        const int par = t & 1;
        IPROBE(6 * t + 0);
        // S^T of tile t-1
        float4 xa[4];
        if constexpr (prev) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xa[g] = xkeep[g];   // S^T of tile t-1 (this wave's own)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias_prev[g] = bias_cur[g];
        }
        // this lane's 16 bias values of tile t (keys 32 t + 8 g + 4 h + e): one 128 B line per (query, tile)
        if constexpr (have && prev) {   // (tile 0's came with the query fragments)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias_cur[g] = bias_load(brow0, t, g);
        }
        float sl[16];
        f32x16 S0, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) S0[r] = 0.f, S1[r] = 0.f;
        if constexpr (have) {
            // ---------------- S^T = K . Q^T (+ point cross term): 54 MFMAs on two accumulation chains
            const f16x8* k_half = st.img[par] + lane;
            f16x8 kf[2][2];
            lds_frag* kp = frag_pin(k_half);
#pragma unroll
            for (int p = 0; p < 2; ++p) kf[0][p] = kp[p * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < KH; ++x) {
                if (x + 1 < KH) {
                    lds_frag* kn = frag_pin(k_half + (x + 1) * 128);
#pragma unroll
                    for (int p = 0; p < 2; ++p) kf[(x + 1) & 1][p] = kn[p * 64];
                }
                const f16x8 (&k)[2] = kf[x & 1];
                const f16x8 (&q)[2] = qf[x];
                // copy slots 2x, 2x+1 (ten in all): image t + 1 -> LDS (its buffer was released by the barrier of tile t - 1),
                // image t + 2 -> registers; two logit elements of tile t - 1 per k-step
                S0 = mfma_f16(k[1], q[0], S0); S1 = mfma_f16(k[0], q[1], S1);   // k_l q_h, k_h q_l
                if (prev && 2 * x < 16) logit_elem(t - 1, 2 * x, xa, sl);   // (same scheduling region as the MFMA pair)
                stage_slot(t + 1, 2 * x);
                __builtin_amdgcn_sched_barrier(0);
                if (x & 1) S1 = mfma_f16(k[0], q[0], S1); else S0 = mfma_f16(k[0], q[0], S0);   // k_h q_h, chains alternate
                if (prev && 2 * x + 1 < 16) logit_elem(t - 1, 2 * x + 1, xa, sl);
                stage_slot(t + 1, 2 * x + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            IPROBE(6 * t + 1);
#pragma unroll
            for (int g = 0; g < 4; ++g)   // kept in registers for the next step's logit arithmetic (no exchange between waves here)
                xkeep[g] = make_float4(S0[4 * g] + S1[4 * g], S0[4 * g + 1] + S1[4 * g + 1],
                                       S0[4 * g + 2] + S1[4 * g + 2], S0[4 * g + 3] + S1[4 * g + 3]);
        } else {
            // The logits of tiles 0 and 1 were stored in steps 1 and 2 and flushed (vmcnt(0) of every wave + barrier) by the last
            // regular step when NT >= 3: fetch them now, under this step's work -- read after the phase change they cost two
            // serial HBM latencies (~10 k cycles per item).
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lg[g] = load_l2(lrsrc, loff0 + 32 * g);
                lg1[g] = load_l2(lrsrc, loff0 + min(1, NT - 1) * 128 + 32 * g);
            }
#pragma unroll
            for (int p = 0; p < 10; ++p) stage_slot(t + 1, p);
#pragma unroll
            for (int r = 0; r < 16; ++r) logit_elem(t - 1, r, xa, sl);   // the last tile's logits: nothing left to hide them under
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and they must have reached L2 before phase 2 reads them back
        }
        if constexpr (flush) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // logits stored so far have reached L2
        IPROBE(6 * t + 2);
        __syncthreads();                                   // partial sums and image t + 1 visible; buffer t & 1 released
        IPROBE(6 * t + 3);
        if constexpr (have) { small_store(t + 2, sm_val); sm_val = small_load(cur, t + 3); }
        IPROBE(6 * t + 4);
    };
    step1(0, std::true_type{}, std::false_type{}, std::false_type{});
    for (int t = 1; t + 1 < NT; ++t) step1(t, std::true_type{}, std::true_type{}, std::false_type{});
    if (NT > 1) step1(NT - 1, std::true_type{}, std::true_type{}, std::true_type{});
    step1(NT, std::false_type{}, std::true_type{}, std::false_type{});
    m_run = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    // images NT (= V(0)) and NT + 1 are in flight; the last tile's logits have reached L2 (vmcnt(0) + barrier above)
    IPROBE(120);

    // =========================================================== phase 2: probabilities and value aggregation
    // Software pipeline again: exp + split of tile t+1 between the MFMAs of tile t.
    f32x16 O[OH];
#pragma unroll
    for (int t = 0; t < OH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    float l_run = 0.f;
    if (NT < 3) {   // too few steps for the early fetch to be ordered behind the stores: read tiles 0, 1 again
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            lg[g] = load_l2(lrsrc, loff0 + 32 * g);
            lg1[g] = load_l2(lrsrc, loff0 + min(1, NT - 1) * 128 + 32 * g);
        }
    }
    f16x8 pc[2][2], pn[2][2];   // planes (h, l) of 2^10 P^T of the current / next tile: [k-step u][plane]
    float pe[16];
    auto p_elem = [&](int r, bool pin) {   // probability of element r of the tile whose logits sit in lg
        float x = lg[r >> 2][r & 3];
        // pin: the value becomes known HERE (an opaque asm with a memory clobber, sandwiched between this slot's LDS stores), so the
        // exp / split arithmetic stays between the MFMAs it is written next to -- hipcc otherwise collects it at the tail of the
        // previous step, in front of the barrier, where nothing hides it (~1 k cycles per step)
        if (pin) asm volatile("" : "+v"(x) :: "memory");
        pe[r] = exp_neg(x - m_run);
        l_run += pe[r];
        if (pin) asm volatile("" : "+v"(pe[r]), "+v"(l_run) :: "memory");   // ... and is complete here (no sinking to the loop tail)
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) p_elem(r, false);
    {
        float ps[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ps[r] = pe[r] * 1024.0f;
        split8(ps, pc[0][0], pc[0][1]);
        split8(ps + 8, pc[1][0], pc[1][1]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) lg[g] = lg1[g];
    auto step2 = [&](int t, auto more_c, auto first_c) {
        constexpr bool more = decltype(more_c)::value, first = decltype(first_c)::value;
        IPROBE(60 + 6 * t + 0);
        if constexpr (!first) __syncthreads();               // V(t) visible; every wave is done with V(t - 1)
        IPROBE(60 + 6 * t + 1);
        // ---------------- O^T += V^T . P^T for this wave's five output tiles
        const f16x8* v_half = st.img[(NT + t) & 1] + lane;
        auto load_v = [&](int x, f16x8 (&d)[2][2]) {
            lds_frag* pa = frag_pin(v_half + x * 256);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p = 0; p < 2; ++p) d[u][p] = pa[(u * 2 + p) * 64];
        };
        f16x8 va[3][2][2], vb[3][2][2];   // fragments of the tile groups (0,1) (5,6) / (2,3,4) (7,8,9), alternately
        load_v(0, va[0]);
        load_v(1, va[1]);
        __builtin_amdgcn_sched_barrier(0);
        // product order of a k-step u (small terms first): (V plane, P plane) = (l,h) (h,l) (h,h)
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
        // The VALU work for the NEXT tile's probabilities is spread over this tile's 24 MFMA groups (6 per tile group): exp of
        // element gi in groups 0 .. 15, the split of element pair j in group 9 + 2j, copy slots in groups 0 .. 9, the logits of
        // tile t + 2 requested in group 16 (lg is dead by then); the next tile group's fragments are requested in this one's first groups.
        auto split_pair = [&](auto jc) {   // elements 2j, 2j+1 of 2^10 pe -> element pair (j & 3) of the planes of k-step j >> 2
            constexpr int j = decltype(jc)::value;
            split2(pe[2 * j] * 1024.0f, pe[2 * j + 1] * 1024.0f, pn[j >> 2][0], pn[j >> 2][1], 2 * (j & 3));
            asm volatile("" : "+v"(pn[j >> 2][0]), "+v"(pn[j >> 2][1]) :: "memory");   // done here, not at the loop tail
        };
        auto ride = [&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            if constexpr (more) {
                if constexpr (gi < 16) p_elem(gi, true);
                if constexpr (gi >= 9 && (gi & 1) && gi <= 23) split_pair(std::integral_constant<int, (gi - 9) / 2>{});
                if constexpr (gi == 16) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) lg[g] = load_l2(lrsrc, loff0 + min(t + 2, NT - 1) * 128 + 32 * g);
                }
            }
            if constexpr (!first && gi < 10) stage_slot(NT + t + 1, gi);
            if constexpr (gi == 0) load_v(2, vb[0]);
            if constexpr (gi == 1) load_v(3, vb[1]);
            if constexpr (gi == 2) load_v(4, vb[2]);
            if constexpr (gi == 6) load_v(5, va[0]);      // (group 0's fragments are dead from group index 6 on)
            if constexpr (gi == 7) load_v(6, va[1]);
            if constexpr (gi == 12) load_v(7, vb[0]);
            if constexpr (gi == 13) load_v(8, vb[1]);
            if constexpr (gi == 14) load_v(9, vb[2]);
        };
        auto group = [&](auto g0c, auto n3c, f32x16& o0, f32x16& o1, f32x16& o2, const f16x8 (&v)[3][2][2]) {
            constexpr int g0 = decltype(g0c)::value;
            constexpr bool three = decltype(n3c)::value;
            f32x16 oa = o0, ob = o1, oc = o2;
            auto one = [&](auto ic) {
                constexpr int i = decltype(ic)::value, u = i / 3, k = i % 3;
                oa = mfma_f16(v[0][u][PA[k]], pc[u][PB[k]], oa); ob = mfma_f16(v[1][u][PA[k]], pc[u][PB[k]], ob);
                if constexpr (three) oc = mfma_f16(v[2][u][PA[k]], pc[u][PB[k]], oc);
                ride(std::integral_constant<int, g0 + i>{});   // (same scheduling region as the MFMAs: hipcc interleaves the VALU between them)
                __builtin_amdgcn_sched_barrier(0);
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
            one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
            o0 = oa; o1 = ob;
            if constexpr (three) o2 = oc;
        };
        f32x16 odummy = O[0];
        group(std::integral_constant<int, 0>{}, std::false_type{}, O[0], O[1], odummy, va);
        group(std::integral_constant<int, 6>{}, std::true_type{}, O[2], O[3], O[4], vb);
        group(std::integral_constant<int, 12>{}, std::false_type{}, O[5], O[6], odummy, va);
        group(std::integral_constant<int, 18>{}, std::true_type{}, O[7], O[8], O[9], vb);
        if constexpr (more) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p = 0; p < 2; ++p) pc[u][p] = pn[u][p];
        }
        __builtin_amdgcn_sched_barrier(0);
        IPROBE(60 + 6 * t + 3);
    };
    IPROBE(123);
    if (NT > 1) {
        step2(0, std::true_type{}, std::true_type{});
        for (int t = 1; t + 1 < NT; ++t) step2(t, std::true_type{}, std::false_type{});
        step2(NT - 1, std::false_type{}, std::false_type{});
    } else {
        step2(0, std::false_type{}, std::true_type{});
    }
    IPROBE(121);

    // ---------------- the next item's query side (unconditionally: after the last item nxt == cur and the values are unused)
    const bool last = item + (int)gridDim.x >= n_items;
    const LaneItem Ln = lane_item(nxt);
    load_queries(nxt, Ln);
    sm_n0 = small_load(nxt, 0); sm_n1 = small_load(nxt, 1); sm_n2 = small_load(nxt, 2);
    // ---------------- epilogue
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (1.0f / l_tot) * (1.0f / 1024.0f);   // (the probabilities went into the products scaled by 2^10)
    {
        // o: accumulator registers 8u .. 8u+7 of channel tile T = fragment k-step 16 head + 2T + u of this row tile (chain order)
        // (RAGGED: the output rows are NOT padded -- row_i sits at position row_i & 31 of row tile row_i >> 5 of the [B N] rows)
        f16x8* o = RAGGED ? a.out_xp + (((row_i >> 5) * a.xp_ksteps + 16 * head) * 2) * 64 + ((int)(row_i & 31) + 32 * h)
                           : a.out_xp + ((rt_q * a.xp_ksteps + 16 * head) * 2) * 64 + lane;
        const bool row_ok = !RAGGED || i < N;
#pragma unroll
        for (int x = 0; x < OH; ++x) {
            const int T = x;
            if (T >= CT) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = O[x][8 * u + j] * inv;
                // this output is an INPUT of the node stream (linear_out, s2s_node_linear): f16 pair planes (x_h, x_l)
                f16x8 ph, pl;
                split8(v, ph, pl);
                f16x8* q = o + ((2 * T + u) * 2) * 64;
                if (row_ok) { q[0] = ph; q[64] = pl; }
            }
        }
    }
    {
        // frame of residue i: R = quat_to_rot(q) (rigid_utils.py:187-207), o_pt = R^T (x - t) (:1122-1133)
        const int feat = H * (256 + 4 * PV + 32);
        const float* f = a.rigids7 + row_i * 7;
        const float qa = f[0], qb_ = f[1], qc = f[2], qd = f[3];
        const float tx = f[4], ty = f[5], tz = f[6];
        const float r00 = qa * qa + qb_ * qb_ - qc * qc - qd * qd, r01 = 2 * qb_ * qc - 2 * qa * qd, r02 = 2 * qb_ * qd + 2 * qa * qc;
        const float r10 = 2 * qb_ * qc + 2 * qa * qd, r11 = qa * qa - qb_ * qb_ + qc * qc - qd * qd, r12 = 2 * qc * qd - 2 * qa * qb_;
        const float r20 = 2 * qb_ * qd - 2 * qa * qc, r21 = 2 * qc * qd + 2 * qa * qb_, r22 = qa * qa - qb_ * qb_ - qc * qc + qd * qd;
        float* ox = a.out + row_i * feat + H * 256 + head * PV;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt_idx = 8 * t + 2 * rq + h;  // point whose (x,y,z,0) group this lane holds
                const f32x16& ov = O[CT + t];           // output tiles 8, 9: the value points
                const float dx = ov[4 * rq + 0] * inv - tx;
                const float dy = ov[4 * rq + 1] * inv - ty;
                const float dz = ov[4 * rq + 2] * inv - tz;
                const float lx = r00 * dx + r10 * dy + r20 * dz;
                const float ly = r01 * dx + r11 * dy + r21 * dz;
                const float lz = r02 * dx + r12 * dy + r22 * dz;
                const float nr = sqrtf(lx * lx + ly * ly + lz * lz + a.eps);
                if (pt_idx < PV && (!RAGGED || i < N)) {
                    ox[pt_idx] = lx;
                    ox[H * PV + pt_idx] = ly;
                    ox[2 * H * PV + pt_idx] = lz;
                    ox[3 * H * PV + pt_idx] = nr;
                }
            }
    }
    if (h == 0 && (!RAGGED || i < N)) {
        float* st2 = a.stats + ((((long long)b * H + head) * N) + i) * 2;
        st2[0] = m_run;
        st2[1] = l_tot;
    }
    IPROBE(122);

    // ---------------- next item: its image 0 is already in buffer 0, its image 1 in the staging registers
    item += (int)gridDim.x;
    if (last) break;
    cur = nxt;
    Lc = Ln;
    nxt = item + (int)gridDim.x < n_items ? decode(item + (int)gridDim.x) : cur;
    __syncthreads();   // every wave is done with the last V image (buffer 1) before image 1 of the next item is written there
    small_store(0, sm_n0);
    small_store(1, sm_n1);
    sm_val = sm_n2;
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// SHORT CHAINS (n_pad <= 64: one or two key tiles) with the shared K / V operands of the folded projections.  The streaming kernel above
// gives a workgroup four query tiles of one (sample, head): with one or two tiles per sample, three or two of its waves repeat work
// (the Science2011 targets, the reference's default inference block).  Here a wave is one (sample, head, query tile) on its own -- no
// LDS, no barriers, no image stream: the K / V fragments of the whole sample are 64 + 72 fragment loads that every head and query
// tile of the sample repeats (L1 / L2 hits), the logits of both key tiles stay in registers between the two phases (they are still
// written for s2s_ipa_opair), and four heads of one query tile share a workgroup.  Same operands, same products, same logit
// arithmetic as the kernel above; the sums differ in order only.
template <bool RAGGED, int NT>
__global__ void __launch_bounds__(256, 1) ipa_attention_short_kernel(PlaneArgs a) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, H = a.H;
    const int NP = RAGGED ? a.NP : N;
    const float c1 = sqrtf(1.0f / (3 * 256)), c2 = sqrtf(1.0f / 3);
    // workgroup = (sample, query tile, half of the heads); wave = head
    int bid = blockIdx.x;
    const int hg = bid % (H / 4); bid /= (H / 4);
    const int qt = bid % NT;
    const int b = bid / NT;
    const int head = hg * 4 + wave;
    const long long rt_q = (long long)b * NT + qt;
    const int i = qt * 32 + c;
    const int ic = RAGGED ? min(i, N - 1) : i;
    const long long row_i = (long long)b * N + ic;
    const long long brow0 = (((long long)b * H + head) * N + ic) * N + 4 * h;
    const float q2_i = a.q2[(rt_q * H + head) * 32 + c];
    float mask_i = a.mask[row_i];
    if constexpr (RAGGED) mask_i = i < N ? mask_i : 0.f;
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.attn_bias, 0, RAGGED ? (int)((long long)a.B * H * N * N * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.logits + ((long long)b * H + head) * NP * NP), 0, NP * NP * 4, 0x00020000);
    const int loff0 = (i * NP + 4 * h) * 4;

    // ---- phase 1: S^T = K . Q'^T (+ point cross term) for every key tile, logits in registers
    f16x8 qf[KQ][2];
    {
        const f16x8* qs = a.q_xp + ((rt_q * (16 * H) + 16 * head) * 2) * 64 + lane;
        const f16x8* ps = a.qp_xp + ((rt_q * H + head) * 4) * 64 + lane;
#pragma unroll
        for (int x = 0; x < KQ; ++x)
#pragma unroll
            for (int p = 0; p < 2; ++p) qf[x][p] = x < 16 ? qs[(x * 2 + p) * 64] : ps[((x - 16) * 2 + p) * 64];
    }
    float sl[NT][16];
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long rt_k = (long long)b * NT + t;
        const f16x8* ks_ = a.k_xp + ((rt_k * 16) * 2) * 64 + lane;
        const f16x8* kp_ = a.kp_xp + ((rt_k * H + head) * 4) * 64 + lane;
        f16x8 kf[KQ][2];
#pragma unroll
        for (int x = 0; x < KQ; ++x)
#pragma unroll
            for (int p = 0; p < 2; ++p) kf[x][p] = x < 16 ? ks_[(x * 2 + p) * 64] : kp_[((x - 16) * 2 + p) * 64];
        float4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (RAGGED) {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(r_bias, (int)(brow0 * 4) + (t * 32 + 8 * g) * 4, 0, 0);
                bias4[g] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            } else {
                bias4[g] = *reinterpret_cast<const float4*>(a.attn_bias + brow0 + t * 32 + 8 * g);
            }
        }
        f32x16 S0, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) S0[r] = 0.f, S1[r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < KQ; ++x) {
            S0 = mfma_f16(kf[x][1], qf[x][0], S0); S1 = mfma_f16(kf[x][0], qf[x][1], S1);   // k_l q_h, k_h q_l
            if (x & 1) S1 = mfma_f16(kf[x][0], qf[x][0], S1); else S0 = mfma_f16(kf[x][0], qf[x][0], S0);   // k_h q_h
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 k2g = *reinterpret_cast<const float4*>(a.k2 + (rt_k * H + head) * 32 + 8 * g + 4 * h);
            float kmv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = t * 32 + 8 * g + 4 * h + e;
                kmv[e] = (!RAGGED || j < N) ? a.mask[(long long)b * N + min(j, N - 1)] : 0.f;
            }
            const float k2v[4] = {k2g.x, k2g.y, k2g.z, k2g.w};
            const float bv[4] = {bias4[g].x, bias4[g].y, bias4[g].z, bias4[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float x = (S0[r] + S1[r]) * c1 + c2 * bv[e];
                x = x + (q2_i + k2v[e]);
                x = x + a.inf * (mask_i * kmv[e] - 1.0f);
                sl[t][r] = x;
                tmax = fmaxf(tmax, x);
            }
            if constexpr (RAGGED)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(sl[t][4 * g]), __float_as_uint(sl[t][4 * g + 1]), __float_as_uint(sl[t][4 * g + 2]),
                                                             __float_as_uint(sl[t][4 * g + 3])}, lrsrc, loff0 + t * 128 + 32 * g, 0, 0);
            else
                *reinterpret_cast<float4*>(a.logits + brow0 + t * 32 + 8 * g) = make_float4(sl[t][4 * g], sl[t][4 * g + 1], sl[t][4 * g + 2], sl[t][4 * g + 3]);
        }
    }
    const float m_run = fmaxf(tmax, __shfl_xor(tmax, 32, 64));

    // ---- phase 2: probabilities and value aggregation
    f32x16 O[OT];
#pragma unroll
    for (int x = 0; x < OT; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[x][r] = 0.f;
    float l_run = 0.f;
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // (V plane, P plane) = (l,h) (h,l) (h,h): small terms first
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long rt_k = (long long)b * NT + t;
        float ps[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = exp_neg(sl[t][r] - m_run);
            l_run += pe;
            ps[r] = pe * 1024.0f;
        }
        f16x8 pc[2][2];
        split8(ps, pc[0][0], pc[0][1]);
        split8(ps + 8, pc[1][0], pc[1][1]);
        const f16x8* vs_ = a.v_vf + ((rt_k * 8) * 4) * 64 + lane;                 // [8 tiles][u][plane][lane]
        const f16x8* vp_ = a.vp_vf + ((rt_k * H + head) * 8) * 64 + lane;        // [2 tiles][u][plane][lane]
        // every V fragment of the key tile is requested before the first product (the query fragments are dead: 160 registers)
        f16x8 vf[OT][2][2];
#pragma unroll
        for (int x = 0; x < OT; ++x)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p = 0; p < 2; ++p) vf[x][u][p] = x < CT ? vs_[((x * 2 + u) * 2 + p) * 64] : vp_[(((x - CT) * 2 + u) * 2 + p) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < OT; ++x)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) O[x] = mfma_f16(vf[x][u][PA[k]], pc[u][PB[k]], O[x]);
    }

    // ---- epilogue (as above)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (1.0f / l_tot) * (1.0f / 1024.0f);
    const bool row_ok = !RAGGED || i < N;
    {
        f16x8* o = RAGGED ? a.out_xp + (((row_i >> 5) * a.xp_ksteps + 16 * head) * 2) * 64 + ((int)(row_i & 31) + 32 * h)
                           : a.out_xp + ((rt_q * a.xp_ksteps + 16 * head) * 2) * 64 + lane;
#pragma unroll
        for (int T = 0; T < CT; ++T)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = O[T][8 * u + j] * inv;
                f16x8 ph, pl;
                split8(v, ph, pl);
                f16x8* q = o + ((2 * T + u) * 2) * 64;
                if (row_ok) { q[0] = ph; q[64] = pl; }
            }
    }
    {
        const int feat = H * (256 + 4 * PV + 32);
        const float* f = a.rigids7 + row_i * 7;
        const float qa = f[0], qb_ = f[1], qc = f[2], qd = f[3];
        const float tx = f[4], ty = f[5], tz = f[6];
        const float r00 = qa * qa + qb_ * qb_ - qc * qc - qd * qd, r01 = 2 * qb_ * qc - 2 * qa * qd, r02 = 2 * qb_ * qd + 2 * qa * qc;
        const float r10 = 2 * qb_ * qc + 2 * qa * qd, r11 = qa * qa - qb_ * qb_ + qc * qc - qd * qd, r12 = 2 * qc * qd - 2 * qa * qb_;
        const float r20 = 2 * qb_ * qd - 2 * qa * qc, r21 = 2 * qc * qd + 2 * qa * qb_, r22 = qa * qa - qb_ * qb_ - qc * qc + qd * qd;
        float* ox = a.out + row_i * feat + H * 256 + head * PV;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt_idx = 8 * t + 2 * rq + h;
                const f32x16& ov = O[CT + t];
                const float dx = ov[4 * rq + 0] * inv - tx;
                const float dy = ov[4 * rq + 1] * inv - ty;
                const float dz = ov[4 * rq + 2] * inv - tz;
                const float lx = r00 * dx + r10 * dy + r20 * dz;
                const float ly = r01 * dx + r11 * dy + r21 * dz;
                const float lz = r02 * dx + r12 * dy + r22 * dz;
                const float nr = sqrtf(lx * lx + ly * ly + lz * lz + a.eps);
                if (pt_idx < PV && row_ok) {
                    ox[pt_idx] = lx;
                    ox[H * PV + pt_idx] = ly;
                    ox[2 * H * PV + pt_idx] = lz;
                    ox[3 * H * PV + pt_idx] = nr;
                }
            }
    }
    if (h == 0 && row_ok) {
        float* st2 = a.stats + ((((long long)b * H + head) * N) + i) * 2;
        st2[0] = m_run;
        st2[1] = l_tot;
    }
}

}  // namespace

#ifdef S2S_IPA_PROBE
extern "C" int s2s_debug_read_ipa8_probe(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(s2s_ipa8_probe), sizeof(s2s_ipa8_probe));
}
#endif

extern "C" int s2s_ipa_prep_points_f16(const float* rigids7, const float* q_pts_lin, const float* kv_pts_lin,
                                          const float* head_w_scaled, void* qp_xp, void* kp_xp, void* vp_vf, float* q2, float* k2,
                                          int n_samples, int n_res, int n_heads, int n_qk_points, int n_v_points, int c_hidden,
                                          const void* s_xp, void* k_shared, void* v_shared, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (n_qk_points != PQ || n_v_points != PV || c_hidden != 256 || n_heads < 1) return (int)hipErrorInvalidValue;
    // shared K / V operands: 8 heads <-> 8 column tiles of a 256-wide s; padded lengths need the gathered K planes
    if (s_xp && (n_heads != 8 || !v_shared || (n_res % 32 != 0 && !k_shared))) return (int)hipErrorInvalidValue;
    const int n_pad = (n_res + 31) / 32 * 32;
    const long long blocks = (long long)n_samples * (n_pad / 32) * n_heads;
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ipa_prep_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rigids7, q_pts_lin, kv_pts_lin,
                       head_w_scaled, sqrtf(1.0f / (3 * c_hidden)), (f16x8*)qp_xp, (f16x8*)kp_xp, (f16x8*)vp_vf, q2, k2, n_heads, n_res,
                       n_pad, s2s::g_range_flag, (const f16x8*)s_xp, (f16x8*)(n_res % 32 != 0 ? k_shared : nullptr), (f16x8*)v_shared);
    return (int)hipGetLastError();
}

extern "C" int s2s_ipa_attention_f16w(const void* q_xp, const void* k_xp, const void* v_vf, const void* qp_xp, const void* kp_xp,
                                        const void* vp_vf, const float* q2, const float* k2, const float* attn_bias,
                                        float* logits_out, float* stats_out, const float* mask, const float* rigids7, float* out,
                                        void* out_xp, int out_xp_ksteps, int n_samples, int n_res, int n_heads, int c_hidden,
                                        int n_qk_points, int n_v_points, int c_pair_z, float inf, float eps, int n_kv_heads, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (c_hidden != 256 || n_qk_points != PQ || n_v_points != PV || c_pair_z != 32 || n_heads < 1 ||
        out_xp_ksteps < 16 * n_heads || !out_xp || (n_kv_heads != n_heads && n_kv_heads != 1))
        return (int)hipErrorInvalidValue;
    const int n_pad = (n_res + 31) / 32 * 32;
    const bool ragged = n_pad != n_res;
    // ragged lengths: the logits live in their own [B,H,n_pad,n_pad] buffer (a row padded in place would run into its neighbour)
    if (ragged && (!logits_out || logits_out == attn_bias)) return (int)hipErrorInvalidValue;
    const int n_qb = (n_pad + 127) / 128;
    const long long items = (long long)n_samples * n_heads * n_qb;
    // the kernel addresses its fragment arrays (and, for ragged lengths, the bias array) through 32-bit buffer offsets
    if ((long long)n_samples * (n_pad / 32) * 16 * n_heads * 2048 >= (1ll << 32)) return (int)hipErrorInvalidValue;
    if (ragged && (long long)n_samples * n_heads * n_res * n_res * 4 >= (1ll << 31)) return (int)hipErrorInvalidValue;
    static const int remap_env = getenv("S2S_IPA_XCD") ? atoi(getenv("S2S_IPA_XCD")) : 1;
    // persistent workgroups, one per CU (83 KiB of LDS each); a multiple of 8 so that workgroup w stays on XCD w % 8
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipErrorUnknown;
        n_cu = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : prop.multiProcessorCount;
    }
    const long long blocks = items < n_cu ? items : n_cu;
    PlaneArgs a{(const f16x8*)q_xp, (const f16x8*)k_xp, (const f16x8*)v_vf, (const f16x8*)qp_xp, (const f16x8*)kp_xp,
                (const f16x8*)vp_vf, q2, k2, attn_bias, logits_out, stats_out, mask, rigids7, out, (f16x8*)out_xp, out_xp_ksteps,
                n_samples, n_res, n_heads, n_pad, inf, eps, (remap_env && items % 8 == 0 && blocks % 8 == 0 && n_qb > 1) ? 1 : 0, n_kv_heads};
    static const int short_env = getenv("S2S_IPA_SHORT") ? atoi(getenv("S2S_IPA_SHORT")) : 1;
    if (short_env && n_kv_heads == 1 && n_heads % 4 == 0 && n_pad <= 64) {   // one wave per (sample, head, query tile): see the short kernel
        const int nt = n_pad / 32;
        const long long wgs = (long long)n_samples * nt * (n_heads / 4);
        if (wgs < (1ll << 31)) {
            hipStream_t st = (hipStream_t)stream;
            if (nt == 1) {
                if (ragged) hipLaunchKernelGGL((ipa_attention_short_kernel<true, 1>), dim3((unsigned)wgs), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((ipa_attention_short_kernel<false, 1>), dim3((unsigned)wgs), dim3(256), 0, st, a);
            } else {
                if (ragged) hipLaunchKernelGGL((ipa_attention_short_kernel<true, 2>), dim3((unsigned)wgs), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((ipa_attention_short_kernel<false, 2>), dim3((unsigned)wgs), dim3(256), 0, st, a);
            }
            return (int)hipGetLastError();
        }
    }
    if (ragged)
        hipLaunchKernelGGL(ipa_attention_f16w_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ipa_attention_f16w_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
