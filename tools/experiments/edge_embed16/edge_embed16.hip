// Edge embedding on split-f16 MFMA with 16-PAIR tiles and TWO waves per SIMD (round 5; the alternative form of s2s_edge_embed_f16x3).
// Same operator and contract as csrc/pair_mlp_f16.hip edge_embed_f16_kernel (reference EmbeddingModule.forward edge branch,
// src/models/net/denoising_ipa.py:137-158, calc_distogram src/common/geo_utils.py:44-56, edge mask :187; with the first IPA block's
// linear_b / down_z fused in, ipa.py:177,253).
//
// Why a second form.  The 32-pair kernel holds the 128-channel state of 32 pairs per wave (accumulators 2 x 64 + planes 64 + the next
// tile's gathered rows 128 registers): ~430 registers, ONE wave per SIMD -- and with 6 - 8 non-matrix instructions per MFMA that wave is
// bound by its own instruction issue and the latency of its LDS / memory waits (PMC: instruction issue active 49 % of the wave's
// cycles, matrix pipe busy 33 %).  Here a wave owns 16 pairs on v_mfma_f32_16x16x32_f16 (a 16-channel x 16-pair accumulator tile is 4
// registers): the whole state is ~100 registers, a workgroup is 8 waves = 128 pairs sharing one weight stream, and two waves per SIMD
// fill each other's waits.  The price: a weight fragment serves 16 pairs instead of 32 (twice the LDS fragment reads per pair).
//
// Layouts (v_mfma_f32_16x16x32_f16: A 16 x 32, lane (i = l & 15, q = l >> 4) holds k = 8 q .. 8 q + 7 of row i; B 32 x 16 likewise for
// column n = l & 15; C / D lane (n, q) holds rows 4 q .. 4 q + 3 of column n).  Every layer is transposed, H^T = W X^T: A = weights
// (rows = 16 output channels), B = activations (columns = the wave's 16 pairs), so lane (pair n, q) holds output channels
// 16 T + 4 q + j of tile T -- and element e of k-step s of the NEXT layer's B operand is taken to be input channel
//     c(s, q, e) = 32 s + 16 (e >> 2) + 4 q + (e & 3)          (tiles 2 s and 2 s + 1, this lane's four rows of each),
// i.e. the accumulator registers ARE the next operand; the host packs the A fragments in that K order (ops.pack_f16x3_embed16_stream).
// Arithmetic, range guard and scaling conventions are those of pair_mlp_f16.hip (weights as the f16 pair split of 2^5 w, accumulators
// carry 2^5 x the layer output, LayerNorm on the scaled values with 1024 eps, three products per block, smallest first).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "range_flag.h"
#include "str2str_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 32 * 1024;
constexpr float kWS = 32.0f, kInvWS = 1.0f / 32.0f;

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float relu1(float x) {   // one v_max_f32 (fmaxf of an opaque value is preceded by a canonicalising v_max x, x)
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// four values -> elements at .. at + 3 of the planes (x_h = rn16(x), x_l = rn16(x - x_h)) + the range maximum: pair_mlp_f16.hip split4_f16
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, f16x8& ph, f16x8& pl, int at, float& amax) {
    unsigned h0, h1, l0, l1;
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
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    u32x4 hv = __builtin_bit_cast(u32x4, ph), lv = __builtin_bit_cast(u32x4, pl);
    hv[at / 2] = h0; hv[at / 2 + 1] = h1;
    lv[at / 2] = l0; lv[at / 2 + 1] = l1;
    ph = __builtin_bit_cast(f16x8, hv);
    pl = __builtin_bit_cast(f16x8, lv);
}
__device__ __forceinline__ float quad_sum(float v) {   // over the four lanes (q = 0 .. 3) that hold one pair's channels
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Weight stream (ops.pack_f16x3_embed16_stream): stages of 32 KiB, double buffered in LDS, one barrier per stage; 8 waves copy 4 KiB each.
//   stages 0, 1: layer 2, k-steps (0, 1) / (2, 3): [k-step 2][tile 8][plane 2 (W_h, W_l)][lane 64][8]
//   stages 2, 3: layer 3 likewise;   stage 4 (PROJ): [k-step 4][tile 3][plane 2] (24 KiB + padding): rows 0 .. 47 of [linear_b; down_z; 0]
// WAVES = 8: one workgroup of 512 threads per CU (two waves per SIMD that share the stream -- and, through its barriers, their phases);
// WAVES = 4: two independent workgroups of 256 threads per CU (one wave per SIMD each: they drift apart, so one's matrix phase meets the
// other's VALU / memory phase), each with its own stream: twice the L2 -> LDS fill traffic per pair.
template <bool PROJ, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) edge_embed16_kernel(
    const float* __restrict__ node_a, const float* __restrict__ node_b, const float* __restrict__ rel_tab,
    const float* __restrict__ bin_tab, const float* __restrict__ bin_lower, const long long* __restrict__ residue_idx,
    const float* __restrict__ ca, const char* __restrict__ wblob, const float* __restrict__ b2, const float* __restrict__ b3,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, float* __restrict__ out,
    long long M, int N, int rel_off, int n_rel, int n_bins, float ln_eps, int out_tiled, const float* __restrict__ proj_b,
    float* __restrict__ proj_bias_out, float* __restrict__ proj_pz_out, int* __restrict__ range_flag) {
    constexpr int kStages = PROJ ? 5 : 4;
    __shared__ __attribute__((aligned(16))) char s_w[2][kStageBytes];
    __shared__ __attribute__((aligned(16))) float s_vec[512 + 64];   // 32 b2 | 32 b3 | gamma | beta | projection bias
    __shared__ __attribute__((aligned(16))) float s_bins[64 + 4];     // distogram lower edges (ascending) | 1e8 | 3e38 ...
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    typedef __attribute__((address_space(3))) const f16x8 lds_frag;

    for (int i = threadIdx.x; i < 512; i += 64 * WAVES)
        s_vec[i] = i < 128 ? kWS * b2[i] : (i < 256 ? kWS * b3[i - 128] : (i < 384 ? gamma[i - 256] : beta[i - 384]));
    if (PROJ && threadIdx.x < 64) s_vec[512 + threadIdx.x] = proj_b[threadIdx.x];
    if (threadIdx.x < 68) s_bins[threadIdx.x] = (int)threadIdx.x < n_bins ? bin_lower[threadIdx.x] : ((int)threadIdx.x == n_bins ? 1e8f : 3.0e38f);

    // ---- weight pipe: this thread's 4 x 16 B of a stage
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, kStages * kStageBytes, 0x00020000);
    const unsigned voff = threadIdx.x * 16;
    constexpr int kPieces = kStageBytes / (64 * WAVES * 16), kPieceStep = 64 * WAVES * 16;
    f32x4 wst[kPieces];
    auto w_load = [&](int stage) {
#pragma unroll
        for (int k = 0; k < kPieces; ++k) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff + k * kPieceStep, stage * kStageBytes, 0);
            wst[k] = __builtin_bit_cast(f32x4, r);
        }
    };
    auto w_store = [&](unsigned buf_off) {
        lds_char* d = (lds_char*)&s_w[0][0] + (buf_off + voff);
#pragma unroll
        for (int k = 0; k < kPieces; ++k) *(lds_f4*)(d + k * kPieceStep) = wst[k];
    };

    const long long NN = (long long)N * N;
    const unsigned n_magic = N >= 2 ? (unsigned)((1ull << 32) / (unsigned)N) : 0u;
    auto div_n = [&](unsigned x, unsigned& qq, unsigned& r) {
        qq = N >= 2 ? __umulhi(x, n_magic) : x;
        r = x - qq * (unsigned)N;
        const bool fix = r >= (unsigned)N;
        qq = fix ? qq + 1 : qq;
        r = fix ? r - (unsigned)N : r;
    };
    const float* mask_or_any = mask ? mask : ca;
    const long long n_wt = (M + 16 * WAVES - 1) / (16 * WAVES);
    float amax = 0.f;

    // Stage g of the stream (counted across tiles: the weights repeat every kStages) sits in buffer g & 1.  Per stage: request the next
    // one (global -> registers), compute from the current buffer, store the next one to the other buffer (last read one stage ago, released by the
    // barrier that ended that stage), barrier.  The two buffer offsets are SWAPPED per stage, not derived from a parity bit: with
    // `base + (par ^ 1) * 32768` hipcc (ROCm 7.2) folded the address into a v_bitop3_b32 (a | (b ^ c)) and the 5-stage variant --
    // the one whose parity is a run-time value -- read weights from the wrong buffer (deterministically wrong layer-3 results).
    w_load(0);
    w_store(0);
    __syncthreads();
    unsigned cur_off = 0, nxt_off = kStageBytes;   // byte offsets of the buffer being read / being filled (swapped per stage)
    auto flip = [&]() { const unsigned t = cur_off; cur_off = nxt_off; nxt_off = t; };

    for (long long wt = blockIdx.x; wt < n_wt; wt += gridDim.x) {
        // ---- this lane's pair
        long long p = (wt * WAVES + wave) * 16 + n;
        const bool valid = p < M;
        p = valid ? p : M - 1;
        unsigned bi, j, bb, ii;
        div_n((unsigned)p, bi, j);
        div_n(bi, bb, ii);
        const unsigned bj = bb * (unsigned)N + j;
        const float ax = ca[bi * 3 + 0], ay = ca[bi * 3 + 1], az = ca[bi * 3 + 2];
        const float bx = ca[bj * 3 + 0], by = ca[bj * 3 + 1], bz = ca[bj * 3 + 2];
        const long long ri = residue_idx[bi], rj = residue_idx[bj];
        const float em = mask ? mask_or_any[bi] * mask_or_any[bj] : 1.0f;
        // distogram bin (no FMA contraction: mirrors torch.linalg.norm of the difference; strict > / < of geo_utils.py:55)
        const float dx = ax - bx, dy = ay - by, dz = az - bz;
        const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 e = *reinterpret_cast<const float4*>(&s_bins[4 * u]);
            cnt += (e.x < dist) + (e.y < dist) + (e.z < dist) + (e.w < dist);
        }
        cnt = cnt > n_bins ? n_bins : cnt;
        const float up = s_bins[cnt];
        const int bin = (cnt >= 1 && dist < up) ? cnt - 1 : -1;
        const float kb = bin < 0 ? 0.f : 1.f;
        long long d = ri - rj + rel_off;
        d = d < 0 ? 0 : (d >= n_rel ? n_rel - 1 : d);

        // ---- first layer: sum of four gathered rows, ((a + b) + r) + kb k, ReLU, split -> B operands of layer 2.
        //      k-step s, elements 0..3 = channels 32 s + 4 q .., elements 4..7 = channels 32 s + 16 + 4 q ..: 16 B chunks 8 s + q / 8 s + 4 + q
        //      of the row (node_b / rel_tab / bin_tab are column-blocked [32 chunks][rows][4])
        f16x8 xp[4][2];
        const float* ra = node_a + (unsigned long long)bi * 128u;
        const float* rb = node_b + ((unsigned long long)bb * 32u * (unsigned)N + j) * 4u;
        const float* rr = rel_tab + (unsigned long long)d * 4u;
        const float* rk = bin_tab + (unsigned long long)(bin < 0 ? 0 : bin) * 4u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float g[8];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int ch = 8 * s + 4 * hf + q;   // chunk of 4 channels
                const float4 va = *reinterpret_cast<const float4*>(ra + 4 * ch);
                const float4 vb = *reinterpret_cast<const float4*>(rb + (unsigned long long)ch * (unsigned)N * 4u);
                const float4 vr = *reinterpret_cast<const float4*>(rr + (unsigned long long)ch * (unsigned)n_rel * 4u);
                const float4 vk = *reinterpret_cast<const float4*>(rk + (unsigned long long)ch * (unsigned)n_bins * 4u);
                const float a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w}, r4[4] = {vr.x, vr.y, vr.z, vr.w},
                            k4[4] = {vk.x, vk.y, vk.z, vk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = __fmaf_rn(1.0f, b4[e], a4[e]);     // (the 32-pair kernel's association and roundings: identical first-layer values)
                    v = __fmaf_rn(1.0f, r4[e], v);
                    v = __fmaf_rn(kb, k4[e], v);
                    g[4 * hf + e] = relu1(v);
                }
            }
            split4(g[0], g[1], g[2], g[3], xp[s][0], xp[s][1], 0, amax);
            split4(g[4], g[5], g[6], g[7], xp[s][0], xp[s][1], 4, amax);
        }

        // ---- one layer: acc[T] (8 output tiles of 16 channels) over 4 k-steps = 2 stages; the next stage travels meanwhile
        auto layer = [&](f32x4 (&acc)[8], int stage0) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int stage = stage0 + st;
                w_load((stage + 1) % kStages);
                lds_frag* wl = (lds_frag*)((lds_char*)&s_w[0][0] + cur_off) + lane;
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl) {
                    const f16x8 xh = xp[2 * st + ksl][0], xl = xp[2 * st + ksl][1];
#pragma unroll
                    for (int T = 0; T < 8; ++T) {
                        const f16x8 fh = wl[((ksl * 8 + T) * 2 + 0) * 64], fl = wl[((ksl * 8 + T) * 2 + 1) * 64];
                        acc[T] = mfma16(fl, xh, acc[T]);   // W_l x_h
                        acc[T] = mfma16(fh, xl, acc[T]);   // W_h x_l
                        acc[T] = mfma16(fh, xh, acc[T]);   // W_h x_h
                    }
                }
                // The accumulators are read by VALU code right after a layer.  hipcc (ROCm 7.2) interleaves that code with the layer's last
                // v_mfma_f32_16x16x32_f16 and leaves too few wait states between an MFMA and a VALU read of its result (found on the 4-wave
                // form: relu of a tile two instructions behind its last MFMA, stale values in the next layer's planes): nothing may cross
                // this point, and the copy + barrier behind it cover the matrix pipe's latency.
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7" ::: "memory");
                w_store(nxt_off);
                __syncthreads();
                flip();
            }
        };
        auto bias_start = [&](f32x4 (&acc)[8], const float* vec) {   // 32 x bias of channels 16 T + 4 q + j
#pragma unroll
            for (int T = 0; T < 8; ++T) {
                const float4 b = *reinterpret_cast<const float4*>(vec + 16 * T + 4 * q);
                acc[T] = f32x4{b.x, b.y, b.z, b.w};
            }
        };

        f32x4 a2[8];
        bias_start(a2, s_vec);
        layer(a2, 0);
        // layer-2 output: ReLU, 2^-5, split -> k-step s of layer 3 = tiles 2 s (elements 0..3) and 2 s + 1 (4..7)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            split4(relu1(a2[2 * s][0]) * kInvWS, relu1(a2[2 * s][1]) * kInvWS, relu1(a2[2 * s][2]) * kInvWS, relu1(a2[2 * s][3]) * kInvWS,
                   xp[s][0], xp[s][1], 0, amax);
            split4(relu1(a2[2 * s + 1][0]) * kInvWS, relu1(a2[2 * s + 1][1]) * kInvWS, relu1(a2[2 * s + 1][2]) * kInvWS,
                   relu1(a2[2 * s + 1][3]) * kInvWS, xp[s][0], xp[s][1], 4, amax);
        }
        f32x4 a3[8];
        bias_start(a3, s_vec + 128);
        layer(a3, 2);

        // ---- LayerNorm(128) over the pair's channels (32 here, the rest in the three other lanes of the pair), edge mask, store
        float sum = 0.f;
#pragma unroll
        for (int T = 0; T < 8; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += a3[T][e];
        const float mean = __fmul_rn(quad_sum(sum), 1.0f / 128);
        float var = 0.f;
#pragma unroll
        for (int T = 0; T < 8; ++T)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = a3[T][e] - mean;
                var = __fmaf_rn(dd, dd, var);
            }
        const float rstd = 1.0f / sqrtf(__fmaf_rn(quad_sum(var), 1.0f / 128, ln_eps * (kWS * kWS)));
        // output position of channels 16 T + 4 q .. + 3 of pair p: row-major p 128 + 16 T + 4 q; tiled (include/str2str_hip.h): block p >> 5,
        // group g = 2 T + (q >> 1), half h = q & 1: float offset 4096 (p >> 5) + 256 g + 128 h + 4 (p & 31)
        float* orow = out_tiled ? out + ((unsigned long long)(p >> 5) * 4096u + 256u * (unsigned)(q >> 1) + 128u * (unsigned)(q & 1) + 4u * (unsigned)(p & 31))
                                : out + ((unsigned long long)p * 128u + 4u * (unsigned)q);
        const int ostep = out_tiled ? 512 : 16;   // floats between consecutive tiles T
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const float4 ga = *reinterpret_cast<const float4*>(s_vec + 256 + 16 * T + 4 * q);
            const float4 be = *reinterpret_cast<const float4*>(s_vec + 384 + 16 * T + 4 * q);
            float4 o;
            o.x = __fmul_rn(__fmaf_rn(__fmul_rn(a3[T][0] - mean, rstd), ga.x, be.x), em);
            o.y = __fmul_rn(__fmaf_rn(__fmul_rn(a3[T][1] - mean, rstd), ga.y, be.y), em);
            o.z = __fmul_rn(__fmaf_rn(__fmul_rn(a3[T][2] - mean, rstd), ga.z, be.z), em);
            o.w = __fmul_rn(__fmaf_rn(__fmul_rn(a3[T][3] - mean, rstd), ga.w, be.w), em);
            if (valid) *reinterpret_cast<float4*>(orow + T * ostep) = o;
            if constexpr (PROJ) split4(o.x, o.y, o.z, o.w, xp[T >> 1][0], xp[T >> 1][1], 4 * (T & 1), amax);
        }

        if constexpr (PROJ) {
            // ---- the first IPA block's linear_b / down_z on the LayerNorm output: rows 0 .. 7 (+ bias) -> attention bias, head-major
            //      [B,8,N,N]; rows 8 .. 39 -> pair_z channel row - 8 ([B,N,N,32]); three tiles of 16 rows (the fourth is padding)
            f32x4 pq[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            {
                w_load(0);   // the next tile's first stage travels under the projection
                lds_frag* wl = (lds_frag*)((lds_char*)&s_w[0][0] + cur_off) + lane;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int T = 0; T < 3; ++T) {
                        const f16x8 fh = wl[((ks * 3 + T) * 2 + 0) * 64], fl = wl[((ks * 3 + T) * 2 + 1) * 64];
                        pq[T] = mfma16(fl, xp[ks][0], pq[T]);
                        pq[T] = mfma16(fh, xp[ks][1], pq[T]);
                        pq[T] = mfma16(fh, xp[ks][0], pq[T]);
                    }
                __builtin_amdgcn_sched_barrier(0);   // (as in layer(): no VALU read of pq between its MFMAs)
                asm volatile("s_nop 7" ::: "memory");
                w_store(nxt_off);
                __syncthreads();
                flip();
            }
            if (valid) {
                const unsigned long long boff = (unsigned long long)p + 7ull * bb * (unsigned long long)NN;
#pragma unroll
                for (int T = 0; T < 3; ++T) {
                    const int row0 = 16 * T + 4 * q;                 // this lane's four rows of [linear_b; down_z]
                    const float4 bq = *reinterpret_cast<const float4*>(s_vec + 512 + row0);
                    const float v0 = __builtin_fmaf(pq[T][0], kInvWS, bq.x), v1 = __builtin_fmaf(pq[T][1], kInvWS, bq.y),
                                v2 = __builtin_fmaf(pq[T][2], kInvWS, bq.z), v3 = __builtin_fmaf(pq[T][3], kInvWS, bq.w);
                    if (row0 < 8) {
                        float* o = proj_bias_out + (boff + (unsigned long long)row0 * (unsigned long long)NN);
                        o[0] = v0; o[NN] = v1; o[2 * NN] = v2; o[3 * NN] = v3;
                    } else if (row0 < 40) {
                        *reinterpret_cast<float4*>(proj_pz_out + (unsigned long long)p * 32u + (row0 - 8)) = make_float4(v0, v1, v2, v3);
                    }
                }
            }
        }
    }
    s2s::range_report(range_flag, amax, s2s::kRangeEdgeEmbed);
}

}  // namespace

namespace {
// smallest number of samples whose pairs fill whole 32-pair blocks (launch boundaries of a tiled pair tensor)
long long tile_aligned_samples16(long long NN) {
    long long q = 32;
    while (q > 1 && (NN * (32 / q)) % 32 != 0) q /= 2;
    return 32 / q;
}
}  // namespace

extern "C" int s2s_edge_embed_f16x3_w16(const float* node_a, const float* node_b, const float* rel_table, const float* bin_table,
                                         const float* bin_lower, const long long* residue_idx, const float* ca_xyz,
                                         const void* weight_stream, const float* b2, const float* b3, const float* ln_gamma,
                                         const float* ln_beta, const float* mask, float* out, int n_samples, int n_res,
                                         int rel_offset, int n_rel, int n_bins, float ln_eps, int out_tiled, const float* proj_bias_cat64,
                                         float* proj_attn_bias, float* proj_pair_z, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (n_bins > 32 || (out_tiled & ~1)) return (int)hipErrorInvalidValue;
    const long long NN = (long long)n_res * n_res;
    if (NN >= (1ll << 31) || (long long)n_rel * 512 >= (1ll << 32)) return (int)hipErrorInvalidValue;
    const char* cap_env = getenv("S2S_EE_MAX_PAIRS");   // test hook: a smaller per-launch pair budget exercises the split
    long long cap = cap_env ? atoll(cap_env) : 0;
    if (cap <= 0 || cap > (1ll << 31) - 1) cap = (1ll << 31) - 1;
    long long chunk = cap / NN;
    if (out_tiled && chunk < n_samples) chunk -= chunk % tile_aligned_samples16(NN);
    if (chunk < 1) return (int)hipErrorInvalidValue;
    int n_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    static const int waves = getenv("S2S_EE16_WAVES") ? atoi(getenv("S2S_EE16_WAVES")) : 4;   // 4: two 256-thread workgroups per CU; 8: one of 512
    for (long long b0 = 0; b0 < n_samples; b0 += chunk) {
        const long long nb = n_samples - b0 < chunk ? n_samples - b0 : chunk;
        const long long M = nb * NN, rows0 = b0 * n_res;
        const int tile_pairs = waves == 8 ? 128 : 64;
        const long long wg_tiles = (M + tile_pairs - 1) / tile_pairs;
        const long long slots = (long long)n_cu * (waves == 8 ? 1 : 2);
        const long long grid = wg_tiles < slots ? wg_tiles : slots;
        const float* na = node_a + rows0 * 128;
        const float* nbp = node_b + rows0 * 128;
        const long long* ridx = residue_idx + rows0;
        const float* cap_ = ca_xyz + rows0 * 3;
        const float* mk = mask ? mask + rows0 : nullptr;
        float* o = out + b0 * NN * 128;
#define S2S_EE16_LAUNCH(P, W, ...) hipLaunchKernelGGL((edge_embed16_kernel<P, W>), dim3((unsigned)grid), dim3(64 * W), 0, (hipStream_t)stream, __VA_ARGS__)
        if (proj_attn_bias) {
            if (waves == 8)
                S2S_EE16_LAUNCH(true, 8, na, nbp, rel_table, bin_table, bin_lower, ridx, cap_, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta, mk, o,
                                M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, proj_bias_cat64, proj_attn_bias + b0 * 8 * NN,
                                proj_pair_z + b0 * NN * 32, s2s::g_range_flag);
            else
                S2S_EE16_LAUNCH(true, 4, na, nbp, rel_table, bin_table, bin_lower, ridx, cap_, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta, mk, o,
                                M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, proj_bias_cat64, proj_attn_bias + b0 * 8 * NN,
                                proj_pair_z + b0 * NN * 32, s2s::g_range_flag);
        } else {
            if (waves == 8)
                S2S_EE16_LAUNCH(false, 8, na, nbp, rel_table, bin_table, bin_lower, ridx, cap_, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta, mk,
                                o, M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                                s2s::g_range_flag);
            else
                S2S_EE16_LAUNCH(false, 4, na, nbp, rel_table, bin_table, bin_lower, ridx, cap_, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta, mk,
                                o, M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                                s2s::g_range_flag);
        }
#undef S2S_EE16_LAUNCH
    }
    return (int)hipGetLastError();
}
