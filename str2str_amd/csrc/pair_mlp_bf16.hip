// EdgeTransition on split-bf16 MFMA ("bf16x6"): fp32-equivalent accuracy at 2.67x the fp32-MFMA rate.
//
// Same operator and contract as s2s_edge_transition (csrc/pair_mlp.hip; reference EdgeTransition.forward,
// src/models/net/layers.py:170-185 + mask ipa.py:372).  Every fp32 operand is split EXACTLY into three bf16
// planes  x = x_h + x_m + x_l  (round-to-nearest residues: |x - x_h - x_m - x_l| <= 2^-27 |x|), weights once on
// the host, activations on the fly; a product keeps the six plane pairs (h,h) (h,m) (m,h) (h,l) (l,h) (m,m) — the
// dropped ones are <= 2^-26 of the product, below one fp32 rounding — on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: 6 MFMAs x 32 cycles per 32x32x16 block instead of 8 x 64 cycles on the fp32 MFMA.
//
// Structure per wave (32 pairs).  Output tiles are processed in PARTS of 4 (128 channels): layer 1 part by part, then for
// each part p: layer-2 tiles 4p..4p+3 over all 384 inputs, ReLU + residual, and immediately the final layer's k-steps that
// consume those 128 hidden channels.  Only one part of the layer-2 activations is ever live (a1 192 + a3 64 + part 64
// registers instead of 384 + 64), which is what leaves room for the weight staging registers.  Inside a part the k-step is
// the outer loop, so an activation group is split once per part and feeds 4 tiles x 6 plane pairs = 24 MFMAs:
//   * the C->B register chaining of pair_mlp.hip carries over: element j of lane (pair, g) in k-step 2t'+u is
//     accumulator register 8u+j of tile t' (row 32t' + (r&3) + 8(r>>2) + 4g); the host packs A fragments in that
//     k order (pack_bf16x3_stream), so no data movement between layers;
//   * the weight stream (1.41 MB for the three layers, all stages 72 KiB) is shared by the 4 waves of a workgroup
//     through LDS, double buffered, one barrier per stage (144 MFMAs per wave): each wave copies a quarter of the next
//     stage global -> VGPR -> LDS in three batches of 6 KiB slotted between the tile groups of the current stage
//     (measured: the LDS-DMA form of the same copy costs ~150 issue cycles per 1 KiB piece and held the MFMA pipe
//     at 40 %); lanes read their fragments with conflict-free ds_read_b128;
//   * two output tiles advance together so consecutive MFMAs hit different accumulators.
#include <hip/hip_runtime.h>

#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int kStageBytes = 48 * 1024;  // 4 k-steps x 4 output tiles x 3 planes x 1 KiB
constexpr int kStages = 30;             // layer 1: 3 parts x 2; then per part: layer 2 x 6, final x 2

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 ldg4(const float* __restrict__ base, int g, int h) {
    return *reinterpret_cast<const float4*>(base + 8 * g + 4 * h);
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// exact 3-way bf16 split of 8 floats
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        const float r1 = x[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        h[j] = a; m[j] = b; l[j] = (__bf16)r2;
    }
}

// one tile group (2 output tiles x 6 plane pairs) of a k-step; lds_ks = this k-step's fragments [T][3 planes][64 lanes]
template <int T>
__device__ __forceinline__ void tile_group(f32x16 (&acc)[T], int tg, const bf16x8* lds_ks, int lane, const bf16x8& xh,
                                           const bf16x8& xm, const bf16x8& xl) {
    const bf16x8* p = lds_ks + (2 * tg) * 3 * 64 + lane;
    const bf16x8 w0h = p[0], w0m = p[64], w0l = p[128], w1h = p[192], w1m = p[256], w1l = p[320];
    f32x16 a0 = acc[2 * tg], a1 = acc[2 * tg + 1];
    a0 = mfma_bf16(w0l, xh, a0); a1 = mfma_bf16(w1l, xh, a1);   // small terms first
    a0 = mfma_bf16(w0h, xl, a0); a1 = mfma_bf16(w1h, xl, a1);
    a0 = mfma_bf16(w0m, xm, a0); a1 = mfma_bf16(w1m, xm, a1);
    a0 = mfma_bf16(w0m, xh, a0); a1 = mfma_bf16(w1m, xh, a1);
    a0 = mfma_bf16(w0h, xm, a0); a1 = mfma_bf16(w1h, xm, a1);
    a0 = mfma_bf16(w0h, xh, a0); a1 = mfma_bf16(w1h, xh, a1);
    acc[2 * tg] = a0; acc[2 * tg + 1] = a1;
}

__global__ void __launch_bounds__(256) edge_transition_bf16_kernel(
    const float* __restrict__ edge, const float* __restrict__ node_ab, const float* __restrict__ node_p,
    const char* __restrict__ wblob, const float* __restrict__ b2, const float* __restrict__ bf,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask,
    float* __restrict__ out, long long M, int N, float ln_eps) {
    __shared__ __attribute__((aligned(16))) char s_w[2][kStageBytes];
    const int lane = threadIdx.x & 63, h = lane >> 5;
    // ---- weight pipe: 12 slots per stage (one per tile group); the next stage is copied global -> VGPR -> LDS
    //      by this wave's quarter (12 KiB) in three batches of four 1 KiB pieces
    const int wave = threadIdx.x >> 6;
    float4 stg0, stg1, stg2, stg3;  // scalars on purpose: an array captured by the lambdas is demoted to LDS by hipcc
    const int piece0 = wave * 1024 + lane * 16;
    auto load_batch = [&](int s, int b) {
        const char* g = wblob + (long long)s * kStageBytes + piece0 + 16 * 1024 * b;
        stg0 = *reinterpret_cast<const float4*>(g);
        stg1 = *reinterpret_cast<const float4*>(g + 4096);
        stg2 = *reinterpret_cast<const float4*>(g + 8192);
        stg3 = *reinterpret_cast<const float4*>(g + 12288);
    };
    auto store_batch = [&](int par, int b) {  // par = parity of the stage being filled (compile-time after inlining)
        char* d = &s_w[par][piece0 + 16 * 1024 * b];
        *reinterpret_cast<float4*>(d) = stg0;
        *reinterpret_cast<float4*>(d + 4096) = stg1;
        *reinterpret_cast<float4*>(d + 8192) = stg2;
        *reinterpret_cast<float4*>(d + 12288) = stg3;
    };
    // slot i (0..7) while stage `cur` (buffer `par`) is being computed: pump the copy of stage cur + 1
    auto slot = [&](int cur, int par, int i) {
        const int nxt = cur + 1;
        if (nxt >= kStages) return;  // wave-uniform
        if (i == 0) load_batch(nxt, 0);
        else if (i == 2) { store_batch(par ^ 1, 0); load_batch(nxt, 1); }
        else if (i == 5) { store_batch(par ^ 1, 1); load_batch(nxt, 2); }
        else if (i == 7) store_batch(par ^ 1, 2);
    };
    auto stage_image = [&](int par) -> const bf16x8* {
        __syncthreads();  // this stage is complete in LDS for every wave; the other buffer is free to refill
        return reinterpret_cast<const bf16x8*>(s_w[par]);
    };
#pragma unroll
    for (int b = 0; b < 3; ++b) { load_batch(0, b); store_batch(0, b); }

    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long p = tile * 32 + (lane & 31);
    const bool valid = p < M;
    if (!valid) p = M - 1;  // waves / lanes past the end run on the last pair and store nothing
    const long long NN = (long long)N * N;
    const long long bb = p / NN;
    const long long rem = p - bb * NN;
    const long long bi = bb * N + rem / N, bj = bb * N + rem % N;
    const float* erow = edge + p * 128;
    const float* arow = node_ab + bi * 768;
    const float* brow = node_ab + bj * 768 + 384;

    // one stage = 4 k-steps x 2 tile groups = 8 slots; xsrc(kk, x) fills the 8 inputs of k-step kk of this stage.
    // Explicit software pipeline (hipcc issues each ds_read right before its MFMA otherwise, exposing the LDS latency
    // once per tile group): slot i+1's six A fragments are fetched and, at odd slots, the next k-step's activations
    // are split while slot i's 12 MFMAs run; the fence at the end of a slot keeps that order.
    auto run_stage = [&](int cur, int par, f32x16 (&acc)[4], auto xsrc) {
        const bf16x8* st = stage_image(par);
        bf16x8 fr[2][12];  // [buffer][tile*3 + plane]
        bf16x8 xp[2][3];
        auto fetch = [&](int kk, bf16x8 (&f)[12]) {
            const bf16x8* p = st + kk * 4 * 3 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 12; ++k) f[k] = p[64 * k];
        };
        fetch(0, fr[0]);
        {
            float x[8];
            xsrc(0, x);
            split8(x, xp[0][0], xp[0][1], xp[0][2]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) fetch(kk + 1, fr[(kk + 1) & 1]);
            slot(cur, par, 2 * kk);
            const bf16x8 (&f)[12] = fr[kk & 1];
            const bf16x8 &xh = xp[kk & 1][0], &xm = xp[kk & 1][1], &xl = xp[kk & 1][2];
            // 24 MFMAs round-robin over the 4 accumulators (each one is touched every 4th issue)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 2], xh, acc[t]);  // (l,h)   small terms first
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 0], xl, acc[t]);  // (h,l)
            if (kk + 1 < 4) {  // next k-step's split rides in the MFMA shadow
                float x[8];
                xsrc(kk + 1, x);
                split8(x, xp[(kk + 1) & 1][0], xp[(kk + 1) & 1][1], xp[(kk + 1) & 1][2]);
            }
            slot(cur, par, 2 * kk + 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 1], xm, acc[t]);  // (m,m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 1], xh, acc[t]);  // (m,h)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 0], xm, acc[t]);  // (h,m)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(f[3 * t + 0], xh, acc[t]);  // (h,h)
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // (The part loops are fully unrolled: rolled into real loops the code shrinks from 93 KB to 78 KB but the register
    //  shuffling hipcc adds around the loop-carried accumulators costs more than it saves: 3.6 vs 3.2 ms at B=16, N=256.)

    // ---- layer 1: 384 <- 128 (edge channels), part by part; accumulators seeded with the per-node terms A_i + B_j (+ b1)
    f32x16 a1[12];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 x = ldg4(arow + 128 * part, 4 * t + rq, h), y = ldg4(brow + 128 * part, 4 * t + rq, h);
                acc[t][4 * rq + 0] = x.x + y.x; acc[t][4 * rq + 1] = x.y + y.y;
                acc[t][4 * rq + 2] = x.z + y.z; acc[t][4 * rq + 3] = x.w + y.w;
            }
#pragma unroll
        for (int s = 0; s < 2; ++s)
            run_stage(2 * part + s, s, acc, [&](int kk, float (&x)[8]) {
                const int ks = 4 * s + kk;  // edge channels 16*ks + 8*h + j
                const float4 u = *reinterpret_cast<const float4*>(erow + 16 * ks + 8 * h);
                const float4 v = *reinterpret_cast<const float4*>(erow + 16 * ks + 8 * h + 4);
                x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
            });
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a1[4 * part + t][r] = fmaxf(acc[t][r], 0.f);
            }
    }

    // ---- layer 2 (384 <- 384) and final layer (128 <- 384), fused part by part
    f32x16 a3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 x = ldg4(bf, 4 * t + rq, h);
            a3[t][4 * rq + 0] = x.x; a3[t][4 * rq + 1] = x.y; a3[t][4 * rq + 2] = x.z; a3[t][4 * rq + 3] = x.w;
        }
    const float* npi = node_p + bi * 128;
    const float* npj = node_p + bj * 128;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int base = 6 + 8 * part;  // stage number of this part's first layer-2 stage (even: buffer parity = s & 1)
        f32x16 a2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 x = ldg4(b2 + 128 * part, 4 * t + rq, h);
                a2[t][4 * rq + 0] = x.x; a2[t][4 * rq + 1] = x.y; a2[t][4 * rq + 2] = x.z; a2[t][4 * rq + 3] = x.w;
            }
#pragma unroll
        for (int s = 0; s < 6; ++s)
            run_stage(base + s, s & 1, a2, [&](int kk, float (&x)[8]) {
                const int ks = 4 * s + kk, tp = ks >> 1, u = ks & 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = a1[tp][8 * u + j];
            });
        // ReLU, then the residual  h2 + x,  x = [e | n'_i | n'_j]: part p is exactly block p of x  (layers.py:181)
        const float* rs = part == 0 ? erow : (part == 1 ? npi : npj);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 x = ldg4(rs, 4 * t + rq, h);
                a2[t][4 * rq + 0] = fmaxf(a2[t][4 * rq + 0], 0.f) + x.x; a2[t][4 * rq + 1] = fmaxf(a2[t][4 * rq + 1], 0.f) + x.y;
                a2[t][4 * rq + 2] = fmaxf(a2[t][4 * rq + 2], 0.f) + x.z; a2[t][4 * rq + 3] = fmaxf(a2[t][4 * rq + 3], 0.f) + x.w;
            }
#pragma unroll
        for (int s = 0; s < 2; ++s)
            run_stage(base + 6 + s, s & 1, a3, [&](int kk, float (&x)[8]) {
                const int ksl = 4 * s + kk, tp = ksl >> 1, u = ksl & 1;  // hidden channels of this part
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = a2[tp][8 * u + j];
            });
    }

    // ---- LayerNorm(128) over the pair's channels (half here, half in lane^32), edge mask, store
    const float em = mask ? mask[bi] * mask[bj] : 1.0f;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += a3[t][r];
    const float mean = xhalf_sum(sum) * (1.0f / 128);
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = a3[t][r] - mean;
            var += d * d;
        }
    const float rstd = 1.0f / sqrtf(xhalf_sum(var) * (1.0f / 128) + ln_eps);
    float* orow = out + p * 128;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int g = 4 * t + rq;
            const float4 ga = ldg4(gamma, g, h), be = ldg4(beta, g, h);
            float4 o;
            o.x = ((a3[t][4 * rq + 0] - mean) * rstd * ga.x + be.x) * em;
            o.y = ((a3[t][4 * rq + 1] - mean) * rstd * ga.y + be.y) * em;
            o.z = ((a3[t][4 * rq + 2] - mean) * rstd * ga.z + be.z) * em;
            o.w = ((a3[t][4 * rq + 3] - mean) * rstd * ga.w + be.w) * em;
            if (valid) *reinterpret_cast<float4*>(orow + 8 * g + 4 * h) = o;
        }
}

}  // namespace

extern "C" int s2s_edge_transition_bf16x6(const float* edge, const float* node_ab, const float* node_p,
                                          const void* weight_stream, const float* b2, const float* bf,
                                          const float* ln_gamma, const float* ln_beta, const float* mask, float* out,
                                          int n_samples, int n_res, float ln_eps, void* stream) {
    const long long M = (long long)n_samples * n_res * n_res;
    if (M <= 0) return 0;
    const long long tiles = (M + 31) / 32;
    hipLaunchKernelGGL(edge_transition_bf16_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       edge, node_ab, node_p, (const char*)weight_stream, b2, bf, ln_gamma, ln_beta, mask, out, M, n_res,
                       ln_eps);
    return (int)hipGetLastError();
}
