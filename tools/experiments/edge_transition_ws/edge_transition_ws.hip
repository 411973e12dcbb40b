// EdgeTransition on split-f16 MFMA, WIDTH-SPLIT form (round 4): same operator, contract, arithmetic ("f16x3": every fp32 operand as
// two f16 planes x_h + x_l, products W_h x_h + W_h x_l + W_l x_h with the weights packed as the split of 2^5 w, fp32 accumulation,
// range guard) and pair-tensor layouts as the pair-per-lane kernel in pair_mlp_f16.hip (reference EdgeTransition.forward,
// src/models/net/layers.py:170-185 + mask ipa.py:372) -- a different division of the work inside a workgroup.
//
// Why.  In the pair-per-lane kernel every wave owns 32 pairs x ALL 384 hidden channels, so every wave reads EVERY weight fragment
// from LDS (4 KiB per 6 MFMAs) and the workgroup copies the whole 0.94 MB weight stream global -> VGPR -> LDS once per 128 pairs:
// 5.5 LDS instructions per 6 MFMAs.  On this part a non-matrix instruction is not hidden by the wave's own (or a second wave's) MFMAs
// -- tools/ubench/mfma_valu_overlap.hip: ~2.2 matrix-pipe cycles per VALU instruction whatever its place -- so those instructions
// are the kernel's overhead (tools/ubench/et_roof4.hip: the skeleton of this form sustains +9 % over the other's).  Here:
//   workgroup tile = 128 consecutive pairs = 4 pair tiles p of 32; wave w OWNS hidden channels [96 w, 96 w + 96) = hidden tiles
//   3 w .. 3 w + 2 of layer 1 and layer 2, and output channels [32 w, 32 w + 32) of the final layer -- for all four pair tiles.
//   A weight fragment is therefore used by exactly ONE wave, for four pair tiles (2 fragments -> 12 MFMAs), and comes straight from
//   L2 into that wave's VGPRs (its own quarter of the stream, in its consumption order: ops.pack_f16x3_stream_ws); the ACTIVATIONS
//   travel through LDS instead, as MFMA B fragments (1 KiB = 64 lanes x 16 B, lane-linear, conflict-free):
//     XB    the edge rows of the tile as planes        [k-step 8][plane 2][pair tile 4]            64 KiB, written by wave p for tile p
//     RING  one ROUND of a layer's output as planes    [producer wave 4][u 2][plane 2][pair tile 4] 64 KiB
//   A layer runs in three rounds r: every wave produces its hidden tile 3 w + r for the four pair tiles (4 accumulators), writes the
//   planes of those 32 channels (= k-steps 2 (3 w + r) + u of the next layer: accumulator registers 8 u .. 8 u + 7 ARE the next
//   layer's B fragment elements, the "chain" order of ops.fragment_order), barrier, and all four waves consume the round's 8 k-steps.
//     layer 1   A_r: 8 k-steps x (2 weight fragments, 8 activation fragments from XB) x 12 MFMAs -> a1 tile; epilogue
//               relu(acc / 32 + A_i + b1 + B_j) -> planes -> RING
//     layer 2   B_r: 8 k-steps x (6 weight fragments, 8 activation fragments from RING) x 36 MFMAs -> 12 accumulators (3 tiles x 4)
//     final     epilogue of own layer-2 tile r: relu(acc / 32 + b2) + x, x = [e | n'_i | n'_j] (layers.py:181) -> planes -> RING;
//               F_r: 8 k-steps x (2, 8) x 12 MFMAs -> output tile w (32 channels) of the four pair tiles, started at 32 bf
//     LayerNorm over the 128 channels of a pair = four waves' 32: per-pair partial sums / squared deviations exchanged through LDS
//               (two small exchanges), normalise + mask + store the own 32 channels of the 128 pairs
//     PROJ      the next IPA block's linear_b / down_z (ipa.py:177,253): LayerNorm output -> planes -> RING, then wave p computes the
//               64 x 128 projection of pair tile p (48 MFMAs; its weight stage is read by all four waves) and stores it
//   per tile and wave: 1440 (+48) MFMAs as before; 592 instead of 992 ds_read_b128, 128 instead of 248 ds_write_b128, ~17 instead
//   of 31 barriers; the weight bytes a CU pulls from L2 are unchanged (0.94 MB per 128 pairs).
//   The next tile's edge rows are requested during the final layer and written to XB (free since layer 1 ended) before the tile ends.
// Summation order differs from the pair-per-lane kernel (k-steps of a layer arrive round-major), so results agree with it to fp32
// rounding, not bit for bit; block 0's residual is the fp32 edge row itself (re-read; the other kernel uses x_h + x_l).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "range_flag.h"
#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kWS = 32.0f, kInvWS = 1.0f / 32.0f;
constexpr int kFrag = 1024;
// per-wave weight stream (bytes): L1 [r 3][k-step 8][plane 2] | L2 [r 3][k-step in round 8][tile 3][plane 2] | LF [r 3][k-step 8][plane 2]
constexpr int kOffL1 = 0, kOffL2 = 3 * 8 * 2 * kFrag, kOffLF = kOffL2 + 3 * 8 * 3 * 2 * kFrag, kWaveStream = kOffLF + 3 * 8 * 2 * kFrag;
static_assert(4 * kWaveStream == 30 * 32 * 1024, "the four wave streams are the 30 stages of the pair-per-lane stream");

__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// four fp32 values -> two packed f16 pairs of each plane (x_h = rn16(x), x_l = rn16(x - x_h)) + range maximum; see pair_mlp_f16.hip
__device__ __forceinline__ void split4(const float (&x)[4], unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1, float& amax) {
    asm volatile(
        "v_max3_f32 %4, %4, |%5|, |%6|\n\t"
        "v_cvt_pk_f16_f32 %0, %5, %6\n\t"
        "v_max3_f32 %4, %4, |%7|, |%8|\n\t"
        "v_cvt_pk_f16_f32 %1, %7, %8\n\t"
        "v_fma_mixlo_f16 %2, -%0, 1.0, %5 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, -%1, 1.0, %7 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, -%0, 1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, -%1, 1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "+v"(amax)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}

__device__ float s2s_ws_one[1] = {1.0f};   // stands in for an absent node mask (read with stride 0)

#ifdef S2S_WS_PROBE
// phase probe (tools/ws_phase_probe.py): s_memtime at the phase boundaries of a tile, thread 0 of workgroup 0, eight tiles kept
__device__ unsigned long long g_ws_probe[8 * 64];
#define WS_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                         if (probe_on) g_ws_probe[(probe_it & 7) * 64 + (k)] = t_; } while (0)
#else
#define WS_STAMP(k)
#endif
#define WS_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool PROJ>
__global__ void __launch_bounds__(256) edge_transition_ws_kernel(
    const float* __restrict__ edge, const float* __restrict__ node_ab, const float* __restrict__ node_p,
    const char* __restrict__ wblob, const float* __restrict__ b2, const float* __restrict__ bf,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask,
    float* __restrict__ out, long long M, int N, float ln_eps, int io_layout, unsigned mask_stride, const float* __restrict__ proj_b,
    float* __restrict__ proj_bias_out, float* __restrict__ proj_pz_out, int* __restrict__ range_flag) {
    __shared__ __attribute__((aligned(16))) char s_x[64 * 1024];      // XB
    __shared__ __attribute__((aligned(16))) char s_r[64 * 1024];      // RING
    __shared__ __attribute__((aligned(16))) float s_vec[768 + 64];     // b2 | 32 bf | gamma | beta | projection bias
    __shared__ __attribute__((aligned(16))) float s_st[2][128][4];     // LayerNorm partials [sum | squared deviations][pair][wave]
    const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) u32x4 lds_frag;
    typedef __attribute__((address_space(3))) float lds_float;
    lds_char* xb = (lds_char*)&s_x[lane * 16];
    lds_char* ring = (lds_char*)&s_r[lane * 16];
    asm volatile("" : "+v"(xb), "+v"(ring));   // opaque: every fragment access = base register + immediate
    auto frag_ld = [&](lds_char* base, int idx) -> u32x4 { return *(const lds_frag*)(base + idx * kFrag); };
    auto frag_st = [&](lds_char* base, int idx, const u32x4& v) { *(lds_frag*)(base + idx * kFrag) = v; };

    // weights: this wave's quarter of the stream (wave-uniform base in SGPRs, lane offset in the VGPR, position as the immediate)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(wblob + (long long)wave * kWaveStream), 0, kWaveStream, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(wblob + 4ll * kWaveStream), 0, 32 * 1024, 0x00020000);
    const unsigned voff = lane * 16;
    auto wld = [&](int byte_off) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(wrs, voff, byte_off, 0); };

    for (int i = threadIdx.x; i < 768; i += 256)
        s_vec[i] = i < 384 ? b2[i] : (i < 512 ? kWS * bf[i - 384] : (i < 640 ? gamma[i - 512] : beta[i - 640]));
    if (PROJ && threadIdx.x < 64) s_vec[768 + threadIdx.x] = proj_b[threadIdx.x];

    // ---- per-tile context: this lane's pair (column `col`) in each of the four pair tiles
    const unsigned NNu = (unsigned)N * (unsigned)N;
    const long long NN = (long long)N * N;
    const unsigned n_magic = N >= 2 ? (unsigned)((1ull << 32) / (unsigned)N) : 0u;
    auto div_n = [&](unsigned x, unsigned& q, unsigned& r) {   // floor(2^32 / N) trick, exact for x < 2^31 (pair_mlp_f16.hip)
        q = N >= 2 ? __umulhi(x, n_magic) : x;
        r = x - q * (unsigned)N;
        const bool fix = r >= (unsigned)N;
        q = fix ? q + 1 : q;
        r = fix ? r - (unsigned)N : r;
    };
    struct Ctx {
        unsigned p[4], bi[4], bj[4];   // flat pair index, flat node rows of i and j
        float em_i[4], em_j[4];        // node masks of i and j: multiplied where the edge mask is used (a product formed in setup would
        bool valid[4];                 // wait for the loads right there)
    };
    auto setup = [&](long long wg_tile) -> Ctx {
        Ctx c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long long pl = wg_tile * 128 + q * 32 + col;
            c.valid[q] = pl < M;
            if (!c.valid[q]) pl = M - 1;   // lanes past the end run on the last pair and store nothing
            const unsigned p = (unsigned)pl;
            unsigned bi, j, bb, i;
            div_n(p, bi, j);
            div_n(bi, bb, i);
            c.p[q] = p;
            c.bi[q] = bi;
            c.bj[q] = bb * (unsigned)N + j;
            c.em_i[q] = mask[bi * mask_stride];
            c.em_j[q] = mask[c.bj[q] * mask_stride];
        }
        return c;
    };
    auto pick = [&](const unsigned (&a)[4]) -> unsigned { return wave == 0 ? a[0] : (wave == 1 ? a[1] : (wave == 2 ? a[2] : a[3])); };
    const bool in_tiled = io_layout & 1, out_tiled = io_layout & 2, no_out = io_layout & 4;
    // float offset of channel group (g = channel / 8, half h) of pair p in either layout (header: "Pair-tensor layouts")
    auto pair_off = [&](unsigned p, bool tiled) -> unsigned long long {
        return tiled ? (unsigned long long)(p >> 5) * 4096u + (unsigned)(h * 128 + (p & 31) * 4) : (unsigned long long)p * 128u + 4u * h;
    };
    const int in_step = in_tiled ? 256 : 8, out_step = out_tiled ? 256 : 8;   // floats between a lane's consecutive 16 B groups

    float amax = 0.f;
    const long long n_wt = (M + 127) / 128;
    long long wt = blockIdx.x;
    Ctx cur = setup(wt);

    // ---- edge rows of pair tile `wave` of a tile: 16 loads of 16 B per lane (chain channel order), split, 16 fragments into XB
    // in four quarters of two k-steps (16 registers in flight each): the kernel has no room for a whole row (64) beside a phase's operands
    float4 xv[2][4];
    auto x_load = [&](unsigned p_own, int qt, int slot = 0) {   // p_own: this lane's pair in pair tile `wave`
        const float* er = edge + pair_off(p_own, in_tiled);
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[slot][i] = *reinterpret_cast<const float4*>(er + (4 * qt + i) * in_step);
    };
    auto x_store = [&](int qt, int slot = 0) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int ks = 2 * qt + k2;
            u32x4 ph, pl;
            const float a[4] = {xv[slot][2 * k2].x, xv[slot][2 * k2].y, xv[slot][2 * k2].z, xv[slot][2 * k2].w};
            const float b[4] = {xv[slot][2 * k2 + 1].x, xv[slot][2 * k2 + 1].y, xv[slot][2 * k2 + 1].z, xv[slot][2 * k2 + 1].w};
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split4(a, h0, h1, l0, l1, amax);
            split4(b, h2, h3, l2, l3, amax);
            ph = u32x4{h0, h1, h2, h3};
            pl = u32x4{l0, l1, l2, l3};
            lds_char* d = xb + wave * kFrag;
            frag_st(d, (ks * 2 + 0) * 4, ph);
            frag_st(d, (ks * 2 + 1) * 4, pl);
        }
    };
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) { x_load(pick(cur.p), qt); x_store(qt); }

    f32x16 a2[12];   // layer-2 accumulators: own hidden tile t (0..2) x pair tile q: a2[4 t + q]
    f32x16 s4[4];    // layer-1 tile of the round / final-layer output tile, per pair tile
    u32x4 wa[16];    // the 16 weight fragments of a layer-1 / final-layer round (requested a phase ahead)
    auto wa_load = [&](int byte_off) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wa[i] = wld(byte_off + i * kFrag);
    };
    // 8 k-steps x 12 MFMAs: acc[q] += W[tile] . B[k-step][q];  B fragments from `src` (XB or RING: index (k-step 2 + plane) 4 + q)
    auto small_round = [&](lds_char* src, f32x16 (&acc)[4]) {
        // B fragments (k-step, plane, pair tile) come two pair tiles at a time, one half k-step ahead: 2 x 4 fragments in flight
        u32x4 fb[2][4];   // [buffer][plane 2 x pair tile 2]
        auto ldb = [&](int kk, int half, u32x4 (&f)[4]) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) f[2 * pl + j] = frag_ld(src, (kk * 2 + pl) * 4 + 2 * half + j);
        };
        ldb(0, 0, fb[0]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int step = 2 * kk + half;
                if (step + 1 < 16) ldb((step + 1) >> 1, (step + 1) & 1, fb[(step + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4& wh = wa[2 * kk];
                const u32x4& wl = wa[2 * kk + 1];
                const u32x4 (&b)[4] = fb[step & 1];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[2 * half + j] = mfma_f16(wl, b[j], acc[2 * half + j]);         // W_l x_h
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[2 * half + j] = mfma_f16(wh, b[2 + j], acc[2 * half + j]);     // W_h x_l
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[2 * half + j] = mfma_f16(wh, b[j], acc[2 * half + j]);         // W_h x_h
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // layer 2, one round: 8 k-steps x 36 MFMAs on a2; weight fragments [k-step][tile 3][plane 2] one k-step ahead, the first k-step's
    // requested by the caller before the round's barrier (fa0); `late` runs at k-step 5 (requests of the NEXT phase's operands)
    u32x4 fa0[6];
    auto fa0_load = [&](int r) {
#pragma unroll
        for (int i = 0; i < 6; ++i) fa0[i] = wld(kOffL2 + (r * 8 * 6 + i) * kFrag);
    };
    auto big_round = [&](int r, bool first, auto&& late) {
        // pair-tile major inside a k-step: the 6 weight fragments of the k-step against ONE pair tile's two planes (9 MFMAs on three
        // accumulators), the next pair tile's planes requested meanwhile -- 2 x 2 activation fragments in flight instead of 2 x 8
        u32x4 fa[2][6], fb[2][2];
        const int wbase = kOffL2 + r * 8 * 6 * kFrag;
#pragma unroll
        for (int i = 0; i < 6; ++i) fa[0][i] = fa0[i];
        fb[0][0] = frag_ld(ring, 0);
        fb[0][1] = frag_ld(ring, 4);
        const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk == 5) late();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int step = 4 * kk + q;
                if (step + 1 < 32) {
                    const int k2 = (step + 1) >> 2, q2 = (step + 1) & 3;
                    fb[(step + 1) & 1][0] = frag_ld(ring, (k2 * 2 + 0) * 4 + q2);
                    fb[(step + 1) & 1][1] = frag_ld(ring, (k2 * 2 + 1) * 4 + q2);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&a)[6] = fa[kk & 1];
                const u32x4& xh = fb[step & 1][0];
                const u32x4& xl = fb[step & 1][1];
                const bool zero = first && kk == 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) a2[4 * t + q] = mfma_f16(a[2 * t + 1], xh, zero ? z16 : a2[4 * t + q]);   // W_l x_h
#pragma unroll
                for (int t = 0; t < 3; ++t) a2[4 * t + q] = mfma_f16(a[2 * t], xl, a2[4 * t + q]);                   // W_h x_l
#pragma unroll
                for (int t = 0; t < 3; ++t) a2[4 * t + q] = mfma_f16(a[2 * t], xh, a2[4 * t + q]);                   // W_h x_h
                // the next k-step's six weight fragments, two behind each of the first three pair tiles' MFMAs (a block of six in
                // front of the k-step costs the issue time of six VMEM instructions in one place)
                if (kk + 1 < 8 && q < 3) {
                    fa[(kk + 1) & 1][2 * q] = wld(wbase + ((kk + 1) * 6 + 2 * q) * kFrag);
                    fa[(kk + 1) & 1][2 * q + 1] = wld(wbase + ((kk + 1) * 6 + 2 * q + 1) * kFrag);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // 16 accumulator values of a unit -> planes of the next layer's k-steps u = 0, 1: staged in registers (pln[q][2 u + plane]) so that the
    // arithmetic of a round's four units runs BEFORE the barrier that frees RING, and only the 16 stores after it
    u32x4 pln[4][4];
    auto make_planes = [&](const float (&v)[16], int q) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float a[4] = {v[8 * u], v[8 * u + 1], v[8 * u + 2], v[8 * u + 3]};
            const float b[4] = {v[8 * u + 4], v[8 * u + 5], v[8 * u + 6], v[8 * u + 7]};
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split4(a, h0, h1, l0, l1, amax);
            split4(b, h2, h3, l2, l3, amax);
            pln[q][2 * u] = u32x4{h0, h1, h2, h3};
            pln[q][2 * u + 1] = u32x4{l0, l1, l2, l3};
        }
    };
    auto ring_put = [&]() {   // RING[this wave][u][plane][pair tile q]
        lds_char* d = ring + wave * (16 * kFrag);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) frag_st(d, k * 4 + q, pln[q][k]);
    };
    auto ld4 = [&](const float* p) -> float4 { return *reinterpret_cast<const float4*>(p); };
    // per-node seeds A_i + b1, B_j of hidden tile T for pair tile q (accumulator layout), and the residual row of layer-2 tile T
    float4 sa[3][4], sb[3][4];      // [pair tile mod 3][quarter]: three units in flight (a fourth set of 32 registers does not fit)
    // (32-bit row offsets into a buffer resource: a 64-bit row pointer per pair tile and array, kept across the rounds by the
    //  compiler, is what spilled in the first version -- and a scratch reload is a VMEM operation the weight prefetch then waits behind)
    const __amdgpu_buffer_rsrc_t nab_rs = __builtin_amdgcn_make_buffer_rsrc((void*)node_ab, 0, (unsigned)((M / N) * 3072u), 0x00020000);
    auto seeds = [&](const Ctx& c, int T, int q) {
        const unsigned oa = c.bi[q] * 3072u + 16u * h, ob = c.bj[q] * 3072u + 1536u + 16u * h;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(nab_rs, oa, 128 * T + 32 * rq, 0);
            const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(nab_rs, ob, 128 * T + 32 * rq, 0);
            sa[q % 3][rq] = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
            sb[q % 3][rq] = make_float4(__uint_as_float(y.x), __uint_as_float(y.y), __uint_as_float(y.z), __uint_as_float(y.w));
        }
    };
    float4 rs[4][4];
    // residual row x = [e | n'_i | n'_j] (layers.py:181) of layer-2 tile T, accumulator layout: ptr + rq * step.  T is wave-uniform;
    // ONE unconditional form for the three sources (a conditional VMEM operation makes hipcc wait for everything older at every use)
    auto resid = [&](int T, int q) {
        const bool isE = T < 4, isI = T < 8;
        const float* base = isE ? edge : node_p;
        const unsigned p = cur.p[q];
        const unsigned idx = isE ? (in_tiled ? p >> 5 : p) : (isI ? cur.bi[q] : cur.bj[q]);
        const unsigned mul = (isE && in_tiled) ? 4096u : 128u;
        const unsigned add = isE ? (in_tiled ? (unsigned)(h * 128) + (p & 31) * 4 + 1024u * T : 4u * h + 32u * T) : 4u * h + 32u * (T - (isI ? 4 : 8));
        const float* ptr = base + ((unsigned long long)idx * mul + add);
        const int step = isE ? in_step : 8;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) rs[q][rq] = *reinterpret_cast<const float4*>(ptr + rq * step);
    };

    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    wa_load(kOffL1);
    seeds(cur, 3 * wave, 0);
    seeds(cur, 3 * wave, 1);
    WS_LDS_BARRIER();   // XB of the first tile, s_vec
#ifdef S2S_WS_PROBE
    const bool probe_on = blockIdx.x == 0 && threadIdx.x == 0;
    int probe_it = 0;
#endif
    for (;;) {
        WS_STAMP(0);
        const long long wt_next = wt + gridDim.x;
        const bool has_next = wt_next < n_wt;
        Ctx nxt = cur;
        unsigned p_nx;   // this lane's pair in pair tile `wave` of the next tile (this tile again when there is none: nothing reads the rows then)
        {
            long long pl = (has_next ? wt_next : wt) * 128 + wave * 32 + col;
            p_nx = (unsigned)(pl < M ? pl : M - 1);
        }
        // =================================================== layers 1 and 2, three rounds
        // entering a round: its 16 layer-1 weight fragments (wa) and the seeds of pair tiles 0, 1 are on their way
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int T = 3 * wave + r;   // hidden tile this wave produces in this round (wave-uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = z16;
            small_round(xb, s4);
            WS_STAMP(1 + 5 * r);
            seeds(cur, T, 2);
            fa0_load(r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[16];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 x = sa[q % 3][rq], y = sb[q % 3][rq];
                    v[4 * rq + 0] = fmaxf(__builtin_fmaf(s4[q][4 * rq + 0], kInvWS, __fadd_rn(x.x, y.x)), 0.f);
                    v[4 * rq + 1] = fmaxf(__builtin_fmaf(s4[q][4 * rq + 1], kInvWS, __fadd_rn(x.y, y.y)), 0.f);
                    v[4 * rq + 2] = fmaxf(__builtin_fmaf(s4[q][4 * rq + 2], kInvWS, __fadd_rn(x.z, y.z)), 0.f);
                    v[4 * rq + 3] = fmaxf(__builtin_fmaf(s4[q][4 * rq + 3], kInvWS, __fadd_rn(x.w, y.w)), 0.f);
                }
                make_planes(v, q);
                if (q == 0) seeds(cur, T, 3);   // into the registers pair tile 0 has just released
            }
            WS_STAMP(2 + 5 * r);
            WS_LDS_BARRIER();          // nobody reads RING any more (the previous round / the previous tile's projection)
            WS_STAMP(3 + 5 * r);
            ring_put();
            WS_LDS_BARRIER();          // the round's a1 planes are in RING
            WS_STAMP(4 + 5 * r);
            if (r < 2) {
                big_round(r, r == 0, [&]() {   // the next round's layer-1 weights and first seeds
                    wa_load(kOffL1 + (r + 1) * 16 * kFrag);
                    seeds(cur, T + 1, 0);
                    seeds(cur, T + 1, 1);
                });
            } else {
                big_round(r, false, [&]() {    // the residual rows of the first final-layer round
#pragma unroll
                    for (int q = 0; q < 4; ++q) resid(3 * wave, q);
                });
            }
            if (r < 2) WS_STAMP(5 + 5 * r);
        }
        WS_STAMP(15);
        // =================================================== final layer, three rounds
        wa_load(kOffLF);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[16];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 bq = *reinterpret_cast<const float4*>(&s_vec[96 * wave + 32 * r + 8 * rq + 4 * h]);
                    const float4 x = rs[q][rq];
                    const f32x16& a = a2[4 * r + q];
                    v[4 * rq + 0] = __fadd_rn(fmaxf(__builtin_fmaf(a[4 * rq + 0], kInvWS, bq.x), 0.f), x.x);
                    v[4 * rq + 1] = __fadd_rn(fmaxf(__builtin_fmaf(a[4 * rq + 1], kInvWS, bq.y), 0.f), x.y);
                    v[4 * rq + 2] = __fadd_rn(fmaxf(__builtin_fmaf(a[4 * rq + 2], kInvWS, bq.z), 0.f), x.z);
                    v[4 * rq + 3] = __fadd_rn(fmaxf(__builtin_fmaf(a[4 * rq + 3], kInvWS, bq.w), 0.f), x.w);
                }
                make_planes(v, q);
            }
            if (r == 0) {   // output tile w starts at 32 bf
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const float4 bq = *reinterpret_cast<const float4*>(&s_vec[384 + 32 * wave + 8 * rq + 4 * h]);
                        s4[q][4 * rq + 0] = bq.x; s4[q][4 * rq + 1] = bq.y; s4[q][4 * rq + 2] = bq.z; s4[q][4 * rq + 3] = bq.w;
                    }
            }
            WS_STAMP(16 + 4 * r);
            WS_LDS_BARRIER();          // RING free
            WS_STAMP(17 + 4 * r);
            ring_put();
            if (r < 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) resid(3 * wave + r + 1, q);   // the next round's residual rows land under this round's MFMAs
            }
            WS_LDS_BARRIER();          // the round's final-layer input planes are in RING
            WS_STAMP(18 + 4 * r);
            x_load(p_nx, r);           // the next tile's edge rows, a quarter per round, two in the last (XB has been free since layer 1 ended)
            if (r == 2) x_load(p_nx, 3, 1);
            small_round(ring, s4);
            WS_STAMP(19 + 4 * r);
            x_store(r);
            if (r == 2) x_store(3, 1);
            if (r < 2) wa_load(kOffLF + (r + 1) * 16 * kFrag);
        }
        // =================================================== LayerNorm over a pair's 128 channels (32 here), mask, store
        // Scale invariant: statistics on the 32 x scaled accumulators with 1024 eps.  The four waves' partial statistics are combined
        // in ONE exchange (Chan et al.): per wave the sum S_w and the squared deviations M2_w from ITS mean over its 32 channels;
        // mean = sum S_w / 128,  M2 = sum M2_w + 32 sum (S_w / 32 - mean)^2.
        // VMEM order of the tail (the counter is in order, and a global store is acknowledged a few thousand cycles late): nothing that
        // is waited for soon may be younger than a store.  With the projection, the LayerNorm output stays in registers, the
        // projection runs (its weight loads have no store in front of them), the next tile's first operands are requested, and only
        // then the pair vectors and the projection are stored; without it, the next tile's operands are requested before the stores.
        if (has_next) nxt = setup(wt_next);   // (index arithmetic + mask loads; nothing waits for the masks before the next LayerNorm)
        if constexpr (!PROJ) {
            wa_load(kOffL1);
            seeds(nxt, 3 * wave, 0);
            seeds(nxt, 3 * wave, 1);
        }
        float mean[4], rstd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sm = __fadd_rn(sm, s4[q][i]);
            sm = xhalf_sum(sm);
            const float mw = __fmul_rn(sm, 1.0f / 32);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {   // (explicit operations throughout the LayerNorm: the four unrolled copies -- one per pair tile --
                const float dd = __fsub_rn(s4[q][i], mw);   //  must round alike, or a pair's result depends on which tile slot it falls into)
                v = __builtin_fmaf(dd, dd, v);
            }
            v = xhalf_sum(v);
            if (h == 0) { s_st[0][32 * q + col][wave] = sm; s_st[1][32 * q + col][wave] = v; }
        }
        WS_STAMP(28);
        WS_LDS_BARRIER();              // (also: every wave is through its last final-layer round -- RING is free)
        WS_STAMP(29);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&s_st[0][32 * q + col][0]);
            const float4 m2 = *reinterpret_cast<const float4*>(&s_st[1][32 * q + col][0]);
            mean[q] = __fmul_rn(__fadd_rn(__fadd_rn(t.x, t.y), __fadd_rn(t.z, t.w)), 1.0f / 128);
            const float d0 = __builtin_fmaf(t.x, 1.0f / 32, -mean[q]), d1 = __builtin_fmaf(t.y, 1.0f / 32, -mean[q]),
                        d2 = __builtin_fmaf(t.z, 1.0f / 32, -mean[q]), d3 = __builtin_fmaf(t.w, 1.0f / 32, -mean[q]);
            const float sq = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, __fmul_rn(d0, d0))));
            const float M2 = __builtin_fmaf(32.0f, sq, __fadd_rn(__fadd_rn(m2.x, m2.y), __fadd_rn(m2.z, m2.w)));
            rstd[q] = 1.0f / sqrtf(__builtin_fmaf(M2, 1.0f / 128, ln_eps * (kWS * kWS)));
        }
        float ov[4][16];   // PROJ: the LayerNorm output of the own 32 channels, stored after the projection
        auto out_store = [&](int q, const float (&v)[16]) {
            float* orow = out + pair_off(cur.p[q], out_tiled) + (out_tiled ? 1024 * wave : 32 * wave);
            if (cur.valid[q] && !no_out) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<float4*>(orow + rq * out_step) = make_float4(v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]);
            }
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[16];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 ga = *reinterpret_cast<const float4*>(&s_vec[512 + 32 * wave + 8 * rq + 4 * h]);
                const float4 be = *reinterpret_cast<const float4*>(&s_vec[640 + 32 * wave + 8 * rq + 4 * h]);
                float4 o;
                const float em = __fmul_rn(cur.em_i[q], cur.em_j[q]);
                auto nrm = [&](float a, float g_, float b_) {
                    return __fmul_rn(__builtin_fmaf(__fmul_rn(__fsub_rn(a, mean[q]), rstd[q]), g_, b_), em);
                };
                o.x = nrm(s4[q][4 * rq + 0], ga.x, be.x);
                o.y = nrm(s4[q][4 * rq + 1], ga.y, be.y);
                o.z = nrm(s4[q][4 * rq + 2], ga.z, be.z);
                o.w = nrm(s4[q][4 * rq + 3], ga.w, be.w);
                v[4 * rq + 0] = o.x; v[4 * rq + 1] = o.y; v[4 * rq + 2] = o.z; v[4 * rq + 3] = o.w;
            }
            if constexpr (PROJ) {
                make_planes(v, q);
#pragma unroll
                for (int i = 0; i < 16; ++i) ov[q][i] = v[i];
            } else {
                out_store(q, v);
            }
        }
        WS_STAMP(30);
        if constexpr (PROJ) ring_put();   // the LayerNorm planes (RING has been free since the LayerNorm barrier)
        if constexpr (PROJ) {
            // =============================================== fused projection of pair tile `wave`: 64 x 128 [linear_b; down_z; 0] on the LayerNorm
            // output; k-step 2 v + u = channels of wave v (chain order), weight stage [k-step 8][tile 2][plane 2] shared by the waves
            u32x4 pw[4][4];            // four k-steps of weight fragments in flight (a k-step is only 6 MFMAs = 192 cycles, L2 is ~800 away)
            auto pwl = [&](int ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[ks % 4][i] = __builtin_amdgcn_raw_buffer_load_b128(prs, voff, (ks * 4 + i) * kFrag, 0);
            };
            pwl(0); pwl(1); pwl(2); pwl(3);
            WS_STAMP(31);
            WS_LDS_BARRIER();          // every wave's LayerNorm planes are in RING, and the next tile's rows in XB
            WS_STAMP(32);
            f32x16 pq[2];
            pq[0] = z16; pq[1] = z16;
            lds_char* src = ring + wave * kFrag;
            u32x4 px[2][2];
            px[0][0] = frag_ld(src, 0); px[0][1] = frag_ld(src, 4);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) { px[(ks + 1) & 1][0] = frag_ld(src, ((ks + 1) * 2 + 0) * 4); px[(ks + 1) & 1][1] = frag_ld(src, ((ks + 1) * 2 + 1) * 4); }
                const u32x4& xh = px[ks & 1][0];
                const u32x4& xl = px[ks & 1][1];
                const u32x4 (&w)[4] = pw[ks % 4];
#pragma unroll
                for (int t = 0; t < 2; ++t) pq[t] = mfma_f16(w[2 * t + 1], xh, pq[t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) pq[t] = mfma_f16(w[2 * t], xl, pq[t]);
#pragma unroll
                for (int t = 0; t < 2; ++t) pq[t] = mfma_f16(w[2 * t], xh, pq[t]);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 4 < 8) pwl(ks + 4);   // into the registers this k-step has just released
                __builtin_amdgcn_sched_barrier(0);
            }
            WS_STAMP(34);
            wa_load(kOffL1);           // the next tile's first layer-1 round: weights and first seeds
            seeds(nxt, 3 * wave, 0);
            seeds(nxt, 3 * wave, 1);
            WS_STAMP(35);
#pragma unroll
            for (int q = 0; q < 4; ++q) out_store(q, ov[q]);
            WS_STAMP(36);
            // rows 0..7 (+ bias) -> attention bias, head-major; rows 8..39 -> pair_z channel row - 8 (same map as pair_mlp.hip)
            const bool ok = wave == 0 ? cur.valid[0] : (wave == 1 ? cur.valid[1] : (wave == 2 ? cur.valid[2] : cur.valid[3]));
            if (ok) {
                const unsigned p = pick(cur.p);
                const unsigned bb = pick(cur.bi) / (unsigned)N;
                const float4 b0 = *reinterpret_cast<const float4*>(&s_vec[768 + 4 * h]);
                float* o = proj_bias_out + ((unsigned long long)p + 7ull * bb * NNu + 4 * h * NN);
                o[0] = __builtin_fmaf(pq[0][0], kInvWS, b0.x);
                o[NN] = __builtin_fmaf(pq[0][1], kInvWS, b0.y);
                o[2 * NN] = __builtin_fmaf(pq[0][2], kInvWS, b0.z);
                o[3 * NN] = __builtin_fmaf(pq[0][3], kInvWS, b0.w);
#pragma unroll
                for (int g = 1; g <= 4; ++g) {
                    const int t = g >> 2, rq = g & 3;
                    const float4 bq = *reinterpret_cast<const float4*>(&s_vec[768 + 8 * g + 4 * h]);
                    *reinterpret_cast<float4*>(proj_pz_out + (unsigned long long)p * 32u + 8 * (g - 1) + 4 * h) =
                        make_float4(__builtin_fmaf(pq[t][4 * rq + 0], kInvWS, bq.x), __builtin_fmaf(pq[t][4 * rq + 1], kInvWS, bq.y),
                                    __builtin_fmaf(pq[t][4 * rq + 2], kInvWS, bq.z), __builtin_fmaf(pq[t][4 * rq + 3], kInvWS, bq.w));
                }
            }
        } else {
            WS_LDS_BARRIER();          // the next tile's rows are in XB
        }
        WS_STAMP(33);
#ifdef S2S_WS_PROBE
        ++probe_it;
#endif
        if (!has_next) break;
        cur = nxt;
        wt = wt_next;
    }
    s2s::range_report(range_flag, amax, s2s::kRangeEdgeTransition);
}

}  // namespace

// Same contract as s2s_edge_transition_f16x3 (include/str2str_hip.h); weight_stream in the per-wave order (ops.pack_f16x3_stream_ws).
extern "C" int s2s_edge_transition_f16x3_ws(const float* edge, const float* node_ab, const float* node_p, const void* weight_stream,
                                            const float* b2, const float* bf, const float* ln_gamma, const float* ln_beta,
                                            const float* mask, float* out, int n_samples, int n_res, float ln_eps, int io_layout,
                                            const float* proj_bias_cat64, float* proj_attn_bias, float* proj_pair_z, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if ((io_layout & ~7) || ((io_layout & 4) && !proj_attn_bias) || (!(io_layout & 4) && !out)) return (int)hipErrorInvalidValue;
    const long long NN = (long long)n_res * n_res;
    // 32-bit pair / head-major indices inside a launch (8 M < 2^32): split the samples over several launches when needed
    const char* cap_env = getenv("S2S_ET_MAX_PAIRS");   // test hook: a smaller per-launch pair budget exercises the split
    long long cap = cap_env ? atoll(cap_env) : 0;
    if (cap <= 0 || cap > (1ll << 29) - 1) cap = (1ll << 29) - 1;
    long long chunk = cap / NN;
    if ((io_layout & 3) && chunk < n_samples) {   // launches of a tiled tensor start on a 32-pair block
        long long k = 1;
        while ((k * NN) % 32) ++k;
        chunk -= chunk % k;
    }
    if (chunk < 1) return (int)hipErrorInvalidValue;
    static const float* one_of[64] = {};   // per device: the address of s2s_ws_one
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return (int)hipErrorInvalidValue;
    if (!one_of[dev_id] && hipGetSymbolAddress((void**)&one_of[dev_id], HIP_SYMBOL(s2s_ws_one)) != hipSuccess) return (int)hipErrorInvalidValue;
    const float* one = one_of[dev_id];
    static int n_cu = 0;  // persistent workgroups, one per CU
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    for (long long b0 = 0; b0 < n_samples; b0 += chunk) {
        const long long nb = n_samples - b0 < chunk ? n_samples - b0 : chunk;
        const long long M = nb * NN, rows0 = b0 * n_res;
        const long long wg_tiles = (M + 127) / 128;
        const long long grid = wg_tiles < n_cu ? wg_tiles : n_cu;
        const float* e = edge + b0 * NN * 128;
        const float* nab = node_ab + rows0 * 768;
        const float* np = node_p + rows0 * 128;
        const float* mk = mask ? mask + rows0 : one;
        const unsigned mks = mask ? 1u : 0u;
        float* o = out ? out + b0 * NN * 128 : nullptr;
        if (proj_attn_bias)
            hipLaunchKernelGGL(edge_transition_ws_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, e, nab, np,
                               (const char*)weight_stream, b2, bf, ln_gamma, ln_beta, mk, o, M, n_res, ln_eps, io_layout, mks, proj_bias_cat64,
                               proj_attn_bias + b0 * 8 * NN, proj_pair_z + b0 * NN * 32, s2s::g_range_flag);
        else
            hipLaunchKernelGGL(edge_transition_ws_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, e, nab, np,
                               (const char*)weight_stream, b2, bf, ln_gamma, ln_beta, mk, o, M, n_res, ln_eps, io_layout, mks,
                               (const float*)nullptr, (float*)nullptr, (float*)nullptr, s2s::g_range_flag);
    }
    return (int)hipGetLastError();
}

#ifdef S2S_WS_PROBE
extern "C" int s2s_ws_probe_read(unsigned long long* host_out) {   // 8 x 64 stamps
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_probe), sizeof(unsigned long long) * 8 * 64);
}
#endif
