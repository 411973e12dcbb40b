// Pair-stream MLP kernels on fp32 MFMA (gfx950): the N x N part of the score network.
//
//   s2s_edge_transition   EdgeTransition.forward          src/models/net/layers.py:170-185 (+ mask, ipa.py:372)
//   s2s_edge_embed        EmbeddingModule edge branch     src/models/net/denoising_ipa.py:137-158 (+ mask :187)
//   s2s_pair_project      linear_b + down_z of IPA        src/models/net/ipa.py:177,253
//
// Design (one wave = one tile of 32 consecutive pairs of the flattened [B*N*N] pair stream):
//   * Every layer is evaluated TRANSPOSED:  H^T[out, pair] = W[out, in] . X^T[in, pair]  with
//     v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  In that orientation the C/D register
//     layout of one layer (lane&31 = pair, regs = 16 of the 32 output rows, lane>>5 picks which 16)
//     IS the B-operand layout of the next layer, because the k-order of a dot product is free:
//     step s of the next layer consumes register s&15 of accumulator tile s>>4 in both wave halves.
//     Activations therefore never leave registers between layers; the [B*N*N, 384] hidden tensor
//     that eager PyTorch materialises 4x (12.9 GB each at B=128, N=256) does not exist.
//   * Weights are pre-packed on the host in exactly the order lanes consume them
//     (pack index = ((s4*T + t)*64 + lane)*4 + q  ->  W[32t + (lane&31)][8*s4 + 4*(lane>>5) + q]),
//     so a wave fetches the A operands of 4 MFMAs with ONE coalesced 1 KiB dwordx4 load; all waves
//     stream the same ~1 MB of weights, which stays L2-resident.
//   * "B layout" of a per-pair vector x[K]: lane (pair, h) holds x[8*g + 4*h + q] for all g, q<4 —
//     loaded from / stored to HBM as float4 (pairs are 512 B rows, every byte of a row is used).
//   * EdgeTransition's first layer only multiplies the 128 edge channels: the node halves of
//     W1.[e|n_i|n_j] are per-node vectors precomputed once per call (A_i, B_j) and enter as the
//     accumulator's initial value (491,520 FLOP/pair instead of 688,128).
#include <hip/hip_runtime.h>

#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

#ifndef S2S_PREFETCH
#define S2S_PREFETCH 8
#endif
constexpr int kPrefetch = S2S_PREFETCH;  // weight fragments in flight per wave (x 1 KiB)

// acc[t] += Wpacked . B   for T output tiles and S4 step-groups (K = 8*S4 inputs).
// bop(s4, q) returns this lane's B operand for step 4*s4+q (must be compile-time selectable).
template <int T, int S4, typename BOp>
__device__ __forceinline__ void mlp_layer(f32x16 (&acc)[T], const float4* __restrict__ wp, int lane, BOp bop) {
    constexpr int NIT = S4 * T;
    constexpr int D = 4 < NIT ? 4 : NIT;  // the streaming kernels (edge_embed, pair_project) run 2+ waves/SIMD: short ring
    float4 w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = wp[d * 64 + lane];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float4 cur = w[it % D];
        if (it + D < NIT) w[it % D] = wp[(it + D) * 64 + lane];
        const int s4 = it / T, t = it % T;
        acc[t] = mfma32(cur.x, bop(s4, 0), acc[t]);
        acc[t] = mfma32(cur.y, bop(s4, 1), acc[t]);
        acc[t] = mfma32(cur.z, bop(s4, 2), acc[t]);
        acc[t] = mfma32(cur.w, bop(s4, 3), acc[t]);
    }
}

// Tile-outer variant: output tile t is finished (all S4 step groups) before tile t+1 starts, so the
// per-tile prologue (accumulator seed: bias / gathered node terms) and epilogue (ReLU, residual) run
// in the shadow of the neighbouring tiles' MFMAs instead of in MFMA-free phases between layers.
// Packed weight order for this routine: index = ((t*S4 + s4)*64 + lane)  (pack_weight(..., tile_major=True)).
//   fetch(t)  : issue the loads tile t's prologue needs (called one tile ahead)
//   begin(t)  : seed acc[t]
//   end(t)    : finish acc[t]
template <int T, int S4, typename BOp, typename Fetch, typename Begin, typename End>
__device__ __forceinline__ void mlp_layer_tiles(f32x16 (&acc)[T], const float4* __restrict__ wp, int lane, BOp bop,
                                                Fetch fetch, Begin begin, End end) {
    constexpr int NIT = S4 * T;
    constexpr int D = kPrefetch < NIT ? kPrefetch : NIT;
    float4 w[D];
#pragma unroll
    for (int d = 0; d < D; ++d) w[d] = wp[d * 64 + lane];
    fetch(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = it / S4, s4 = it % S4;
        if (s4 == 0) {
            begin(t);
            if (t + 1 < T) fetch(t + 1);
        }
        const float4 cur = w[it % D];
        if (it + D < NIT) w[it % D] = wp[(it + D) * 64 + lane];
        acc[t] = mfma32(cur.x, bop(s4, 0), acc[t]);
        acc[t] = mfma32(cur.y, bop(s4, 1), acc[t]);
        acc[t] = mfma32(cur.z, bop(s4, 2), acc[t]);
        acc[t] = mfma32(cur.w, bop(s4, 3), acc[t]);
        if (s4 == S4 - 1) end(t);
        // pin the software pipeline: without this fence hipcc sinks each ring refill down to its use and
        // waits vmcnt(0) per fragment (one load in flight, 2x slower)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// this lane's 4 consecutive elements of group g of a B-layout vector
__device__ __forceinline__ float4 ldg4(const float* __restrict__ base, int g, int h) {
    return *reinterpret_cast<const float4*>(base + 8 * g + 4 * h);
}

__device__ __forceinline__ float f4(const float4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// LayerNorm over the 32*T channels a pair owns (half in this lane, half in lane^32), gamma/beta,
// optional scale, then float4 stores in B layout.
template <int T>
__device__ __forceinline__ void ln_store(f32x16 (&acc)[T], const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float eps, float scale, float* __restrict__ out_row, int h, bool valid) {
    constexpr float inv_n = 1.0f / (32 * T);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    const float mean = xhalf_sum(s) * inv_n;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[t][r] - mean;
            v += d * d;
        }
    const float rstd = 1.0f / sqrtf(xhalf_sum(v) * inv_n + eps);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int g = 4 * t + rq;
            const float4 ga = ldg4(gamma, g, h), be = ldg4(beta, g, h);
            float4 o;
            o.x = ((acc[t][4 * rq + 0] - mean) * rstd * ga.x + be.x) * scale;
            o.y = ((acc[t][4 * rq + 1] - mean) * rstd * ga.y + be.y) * scale;
            o.z = ((acc[t][4 * rq + 2] - mean) * rstd * ga.z + be.z) * scale;
            o.w = ((acc[t][4 * rq + 3] - mean) * rstd * ga.w + be.w) * scale;
            if (valid) *reinterpret_cast<float4*>(out_row + 8 * g + 4 * h) = o;
            acc[t][4 * rq + 0] = o.x; acc[t][4 * rq + 1] = o.y; acc[t][4 * rq + 2] = o.z; acc[t][4 * rq + 3] = o.w;
        }
}

// Optional fused epilogue: the IPA pair projections of the NEXT block (linear_b + down_z, ipa.py:177,253) applied
// to the pair vector this kernel has just produced and still holds in registers in B layout (acc after ln_store).
// Same arithmetic as pair_project_kernel on the stored values (bit-identical), minus a 512 B/pair re-read of z.
// attention bias, head-major [B, 8, N, N] (what s2s_ipa_attention streams per head): rows 4h..4h+3 of the projection
// (boff = offset of head 0 of this pair = p + 7*b*NN)
__device__ __forceinline__ void store_bias_headmajor(float* __restrict__ bias_out, long long boff, long long NN, int h,
                                                     const f32x16& acc0) {
    float* o = bias_out + boff + 4 * h * NN;
    o[0] = acc0[0];
    o[NN] = acc0[1];
    o[2 * NN] = acc0[2];
    o[3 * NN] = acc0[3];
}

__device__ __forceinline__ void project_store(f32x16 (&zacc)[4], const float4* __restrict__ wp, const float* __restrict__ bcat,
                                              float* __restrict__ bias_out, float* __restrict__ pairz_out, long long p,
                                              long long boff, long long NN, int lane, int h, bool valid) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x = ldg4(bcat, 4 * t + rq, h);
            acc[t][4 * rq + 0] = x.x; acc[t][4 * rq + 1] = x.y; acc[t][4 * rq + 2] = x.z; acc[t][4 * rq + 3] = x.w;
        }
    mlp_layer<2, 16>(acc, wp, lane, [&](int s4, int q) { return zacc[s4 >> 2][(s4 & 3) * 4 + q]; });
    if (!valid) return;
    store_bias_headmajor(bias_out, boff, NN, h, acc[0]);
#pragma unroll
    for (int g = 1; g <= 4; ++g) {
        const int t = g >> 2, rq = g & 3;
        *reinterpret_cast<float4*>(pairz_out + p * 32 + 8 * (g - 1) + 4 * h) =
            make_float4(acc[t][4 * rq + 0], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]);
    }
}

template <int T>
__device__ __forceinline__ void relu_(f32x16 (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);
}

struct PairIdx {
    long long p;   // clamped flat pair index
    long long bi;  // b*N + i
    long long bj;  // b*N + j
    long long boff;  // offset of this pair's head-0 entry in a head-major [B,8,N,N] tensor
    bool valid;
};

__device__ __forceinline__ PairIdx pair_index(long long tile, int lane, long long M, int N) {
    PairIdx x;
    long long p = tile * 32 + (lane & 31);
    x.valid = p < M;
    if (!x.valid) p = M - 1;
    x.p = p;
    const long long NN = (long long)N * N;
    const long long b = p / NN;
    const long long rem = p - b * NN;
    const long long i = rem / N, j = rem - i * N;
    x.bi = b * N + i;
    x.bj = b * N + j;
    x.boff = p + 7 * b * NN;
    return x;
}

// ------------------------------------------------------------------------------------------
// EdgeTransition: c_z = 128 edge channels, hidden 384 = 128 + 2*128.
//   node_ab [B*N, 768] : [0,384)  = W1[:,128:256].n'_i + b1   (row-i contribution to layer 1)
//                        [384,768) = W1[:,256:384].n'_j        (column-j contribution)
//   node_p  [B*N, 128] : n' = initial_embed(node)
//   w1p : packed W1[:, 0:128]  (384 x 128), w2p : packed W2 (384 x 384), wfp : packed Wf (128 x 384)
__global__ void __launch_bounds__(256) edge_transition_kernel(
    const float* __restrict__ edge, const float* __restrict__ node_ab, const float* __restrict__ node_p,
    const float4* __restrict__ w1p, const float4* __restrict__ w2p, const float4* __restrict__ wfp,
    const float* __restrict__ b2, const float* __restrict__ bf, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mask, float* __restrict__ out, long long M, int N,
    float ln_eps, const float4* __restrict__ proj_wp, const float* __restrict__ proj_b, float* __restrict__ proj_bias_out,
    float* __restrict__ proj_pz_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= M) return;
    const PairIdx px = pair_index(tile, lane, M, N);
    const float* erow = edge + px.p * 128;
    const float* arow = node_ab + px.bi * 768;
    const float* brow = node_ab + px.bj * 768 + 384;

    // ---- layer 1: 384 <- 128.  Seed of tile t = A_i + B_j (row gathers, fetched one tile ahead), ReLU at tile end
    f32x16 a1[12];
    float4 e[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) e[g] = ldg4(erow, g, h);
    {
        float4 sa[4], sb[4];  // single staging set: begin(t) drains it before fetch(t+1) refills it
        mlp_layer_tiles<12, 16>(
            a1, w1p, lane, [&](int s4, int q) { return f4(e[s4], q); },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) { sa[rq] = ldg4(arow, 4 * t + rq, h); sb[rq] = ldg4(brow, 4 * t + rq, h); }
            },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 x = sa[rq], y = sb[rq];
                    a1[t][4 * rq + 0] = x.x + y.x; a1[t][4 * rq + 1] = x.y + y.y;
                    a1[t][4 * rq + 2] = x.z + y.z; a1[t][4 * rq + 3] = x.w + y.w;
                }
            },
            [&](int t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[t][r] = fmaxf(a1[t][r], 0.f);
            });
    }

    // ---- layer 2: 384 <- 384.  Tile end: ReLU, then the residual  h2 + x,  x = [e | n'_i | n'_j]  (layers.py:181)
    f32x16 a2[12];
    {
        const float* npi = node_p + px.bi * 128;
        const float* npj = node_p + px.bj * 128;
        // re-read the edge row instead of keeping the 64 layer-1 operand registers alive across layer 2:
        // the asm launders the pointer so the compiler cannot CSE these loads with the earlier ones
        const float* erow2 = erow;
        asm volatile("" : "+v"(erow2));
        float4 sx[4], sbias[4];  // residual of tile t is fetched at begin(t), used at end(t)
        mlp_layer_tiles<12, 48>(
            a2, w2p, lane, [&](int s4, int q) { return a1[s4 >> 2][(s4 & 3) * 4 + q]; },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) sbias[rq] = ldg4(b2, 4 * t + rq, h);
            },
            [&](int t) {
                const float* src = t < 4 ? erow2 : (t < 8 ? npi : npj);
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 x = sbias[rq];
                    a2[t][4 * rq + 0] = x.x; a2[t][4 * rq + 1] = x.y; a2[t][4 * rq + 2] = x.z; a2[t][4 * rq + 3] = x.w;
                    sx[rq] = ldg4(src, 4 * (t & 3) + rq, h);
                }
            },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 x = sx[rq];
                    a2[t][4 * rq + 0] = fmaxf(a2[t][4 * rq + 0], 0.f) + x.x; a2[t][4 * rq + 1] = fmaxf(a2[t][4 * rq + 1], 0.f) + x.y;
                    a2[t][4 * rq + 2] = fmaxf(a2[t][4 * rq + 2], 0.f) + x.z; a2[t][4 * rq + 3] = fmaxf(a2[t][4 * rq + 3], 0.f) + x.w;
                }
            });
    }

    // ---- final layer: 128 <- 384, LayerNorm, edge mask
    f32x16 a3[4];
    {
        float4 sbias[4];
        mlp_layer_tiles<4, 48>(
            a3, wfp, lane, [&](int s4, int q) { return a2[s4 >> 2][(s4 & 3) * 4 + q]; },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) sbias[rq] = ldg4(bf, 4 * t + rq, h);
            },
            [&](int t) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 x = sbias[rq];
                    a3[t][4 * rq + 0] = x.x; a3[t][4 * rq + 1] = x.y; a3[t][4 * rq + 2] = x.z; a3[t][4 * rq + 3] = x.w;
                }
            },
            [&](int) {});
    }
    const float em = mask ? mask[px.bi] * mask[px.bj] : 1.0f;
    ln_store<4>(a3, gamma, beta, ln_eps, em, out + px.p * 128, h, px.valid);
    if (proj_wp) project_store(a3, proj_wp, proj_b, proj_bias_out, proj_pz_out, px.p, px.boff, (long long)N * N, lane, h, px.valid);
}

// ------------------------------------------------------------------------------------------
// Edge embedding.  First Linear(120 -> 128) of edge_embed is a sum of four table rows:
//   node_a[b,i] = W[:, 0:33].[t_emb, fixed_i] + bias,   node_b[b,j] = W[:, 33:66].[t_emb, fixed_j]
//   rel_tab[d + rel_off] = W[:, 66:98].posemb(d),  d = idx_i - idx_j
//   bin_tab[k] = W[:, 98+k]  for the distogram bin k of |ca_i - ca_j| (strict > lower, < upper;
//   geo_utils.py:44-56), none when the distance sits exactly on an edge or is 0.
// then two 128x128 MFMA layers, LayerNorm, edge mask.
__global__ void __launch_bounds__(256, 2) edge_embed_kernel(
    const float* __restrict__ node_a, const float* __restrict__ node_b, const float* __restrict__ rel_tab,
    const float* __restrict__ bin_tab, const float* __restrict__ bin_lower, const long long* __restrict__ residue_idx,
    const float* __restrict__ ca, const float4* __restrict__ w2p, const float4* __restrict__ w3p,
    const float* __restrict__ b2, const float* __restrict__ b3, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mask, float* __restrict__ out, long long M, int N,
    int rel_off, int n_rel, int n_bins, float ln_eps, const float4* __restrict__ proj_wp, const float* __restrict__ proj_b,
    float* __restrict__ proj_bias_out, float* __restrict__ proj_pz_out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= M) return;
    const PairIdx px = pair_index(tile, lane, M, N);

    // distogram bin of this pair (no FMA contraction: mirrors torch.linalg.norm of the difference)
    const float dx = ca[px.bi * 3 + 0] - ca[px.bj * 3 + 0];
    const float dy = ca[px.bi * 3 + 1] - ca[px.bj * 3 + 1];
    const float dz = ca[px.bi * 3 + 2] - ca[px.bj * 3 + 2];
    const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    int bin = -1;
    for (int k = 0; k < n_bins; ++k) {
        const float lo = bin_lower[k];
        const float up = (k + 1 < n_bins) ? bin_lower[k + 1] : 1e8f;
        if (dist > lo && dist < up) bin = k;
    }
    long long d = residue_idx[px.bi] - residue_idx[px.bj] + rel_off;
    d = d < 0 ? 0 : (d >= n_rel ? n_rel - 1 : d);
    const float* ra = node_a + px.bi * 128;
    const float* rb = node_b + px.bj * 128;
    const float* rr = rel_tab + d * 128;
    const float* rk = bin_tab + (long long)(bin < 0 ? 0 : bin) * 128;
    const float kb = bin < 0 ? 0.f : 1.f;

    float4 h1[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float4 a = ldg4(ra, g, h), b = ldg4(rb, g, h), r = ldg4(rr, g, h), k = ldg4(rk, g, h);
        h1[g].x = fmaxf(a.x + b.x + r.x + kb * k.x, 0.f);
        h1[g].y = fmaxf(a.y + b.y + r.y + kb * k.y, 0.f);
        h1[g].z = fmaxf(a.z + b.z + r.z + kb * k.z, 0.f);
        h1[g].w = fmaxf(a.w + b.w + r.w + kb * k.w, 0.f);
    }
    f32x16 a2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x = ldg4(b2, 4 * t + rq, h);
            a2[t][4 * rq + 0] = x.x; a2[t][4 * rq + 1] = x.y; a2[t][4 * rq + 2] = x.z; a2[t][4 * rq + 3] = x.w;
        }
    mlp_layer<4, 16>(a2, w2p, lane, [&](int s4, int q) { return q == 0 ? h1[s4].x : q == 1 ? h1[s4].y : q == 2 ? h1[s4].z : h1[s4].w; });
    relu_<4>(a2);
    f32x16 a3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x = ldg4(b3, 4 * t + rq, h);
            a3[t][4 * rq + 0] = x.x; a3[t][4 * rq + 1] = x.y; a3[t][4 * rq + 2] = x.z; a3[t][4 * rq + 3] = x.w;
        }
    mlp_layer<4, 16>(a3, w3p, lane, [&](int s4, int q) { return a2[s4 >> 2][(s4 & 3) * 4 + q]; });
    const float em = mask ? mask[px.bi] * mask[px.bj] : 1.0f;
    ln_store<4>(a3, gamma, beta, ln_eps, em, out + px.p * 128, h, px.valid);
    if (proj_wp) project_store(a3, proj_wp, proj_b, proj_bias_out, proj_pz_out, px.p, px.boff, (long long)N * N, lane, h, px.valid);
}

// ------------------------------------------------------------------------------------------
// IPA pair projections: out = Wcat . z + bcat with Wcat = [linear_b (H rows) ; down_z (c_z/4 rows)]
// zero-padded to 64 rows.  Writes bias_out [B, H, N, N] (head-major) and pairz_out [M, PZ] (H = 8, PZ = 32).
__global__ void __launch_bounds__(256) pair_project_kernel(const float* __restrict__ edge, const float4* __restrict__ wp,
                                                           const float* __restrict__ bcat, float* __restrict__ bias_out,
                                                           float* __restrict__ pairz_out, long long M, long long NN) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tile * 32 >= M) return;
    long long p = tile * 32 + (lane & 31);
    const bool valid = p < M;
    if (!valid) p = M - 1;
    const float* erow = edge + p * 128;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x = ldg4(bcat, 4 * t + rq, h);
            acc[t][4 * rq + 0] = x.x; acc[t][4 * rq + 1] = x.y; acc[t][4 * rq + 2] = x.z; acc[t][4 * rq + 3] = x.w;
        }
    float4 e[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) e[g] = ldg4(erow, g, h);
    mlp_layer<2, 16>(acc, wp, lane, [&](int s4, int q) { return q == 0 ? e[s4].x : q == 1 ? e[s4].y : q == 2 ? e[s4].z : e[s4].w; });
    if (!valid) return;
    // rows 0..7 -> attention bias (rows 0..3 live in h=0 / rq=0, rows 4..7 in h=1 / rq=0)
    store_bias_headmajor(bias_out, p + 7 * (p / NN) * NN, NN, h, acc[0]);
    // rows 8..39 -> pair_z channel c = row - 8 : group g = row/8 in 1..4, i.e. (t, rq) = (0,1..3), (1,0)
#pragma unroll
    for (int g = 1; g <= 4; ++g) {
        const int t = g >> 2, rq = g & 3;
        *reinterpret_cast<float4*>(pairz_out + p * 32 + 8 * (g - 1) + 4 * h) =
            make_float4(acc[t][4 * rq + 0], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]);
    }
}

inline int tiles_grid(long long M, int waves_per_block) {
    const long long tiles = (M + 31) / 32;
    return (int)((tiles + waves_per_block - 1) / waves_per_block);
}

}  // namespace

extern "C" {

int s2s_edge_transition(const float* edge, const float* node_ab, const float* node_p, const float* w1_packed,
                        const float* w2_packed, const float* wf_packed, const float* b2, const float* bf,
                        const float* ln_gamma, const float* ln_beta, const float* mask, float* out, int n_samples,
                        int n_res, float ln_eps, const float* proj_w_packed, const float* proj_bias_cat64,
                        float* proj_attn_bias, float* proj_pair_z, void* stream) {
    const long long M = (long long)n_samples * n_res * n_res;
    if (M <= 0) return 0;
    hipLaunchKernelGGL(edge_transition_kernel, dim3(tiles_grid(M, 4)), dim3(256), 0, (hipStream_t)stream, edge, node_ab,
                       node_p, (const float4*)w1_packed, (const float4*)w2_packed, (const float4*)wf_packed, b2, bf,
                       ln_gamma, ln_beta, mask, out, M, n_res, ln_eps, (const float4*)proj_w_packed, proj_bias_cat64,
                       proj_attn_bias, proj_pair_z);
    return (int)hipGetLastError();
}

int s2s_edge_embed(const float* node_a, const float* node_b, const float* rel_table, const float* bin_table,
                   const float* bin_lower, const long long* residue_idx, const float* ca_xyz, const float* w2_packed,
                   const float* w3_packed, const float* b2, const float* b3, const float* ln_gamma, const float* ln_beta,
                   const float* mask, float* out, int n_samples, int n_res, int rel_offset, int n_rel, int n_bins,
                   float ln_eps, const float* proj_w_packed, const float* proj_bias_cat64, float* proj_attn_bias,
                   float* proj_pair_z, void* stream) {
    const long long M = (long long)n_samples * n_res * n_res;
    if (M <= 0) return 0;
    hipLaunchKernelGGL(edge_embed_kernel, dim3(tiles_grid(M, 4)), dim3(256), 0, (hipStream_t)stream, node_a, node_b,
                       rel_table, bin_table, bin_lower, residue_idx, ca_xyz, (const float4*)w2_packed,
                       (const float4*)w3_packed, b2, b3, ln_gamma, ln_beta, mask, out, M, n_res, rel_offset, n_rel, n_bins,
                       ln_eps, (const float4*)proj_w_packed, proj_bias_cat64, proj_attn_bias, proj_pair_z);
    return (int)hipGetLastError();
}

int s2s_pair_project(const float* edge, const float* w_packed, const float* bias_cat64, float* attn_bias, float* pair_z,
                     int n_samples, int n_res, void* stream) {
    const long long M = (long long)n_samples * n_res * n_res;
    if (M <= 0) return 0;
    hipLaunchKernelGGL(pair_project_kernel, dim3(tiles_grid(M, 4)), dim3(256), 0, (hipStream_t)stream, edge,
                       (const float4*)w_packed, bias_cat64, attn_bias, pair_z, M, (long long)n_res * n_res);
    return (int)hipGetLastError();
}

}  // extern "C"
