// Self-attention core of the trunk's nn.TransformerEncoderLayer (reference src/models/net/ipa.py:312-317,357: d_model 320,
// 4 heads of 80 channels, sequence = the N residues of one sample) on exact fp32 MFMA, flash style: softmax(q k^T / sqrt(dh)
// + key bias) v  per (sample, head) without materialising the [N, N] matrix.
//
// Input: the in_proj output qkv [B*N, 3*D] fp32 (q | k | v, each [heads, dh]); key_bias [B, N] is ADDED to the logits of key j
// -- PyTorch's semantics for a FLOAT key-padding mask (1 - mask: a no-op for the all-ones masks of every reference run);
// exact-padding mode (mixed-length batches) passes -inf there.  Output: the head-concatenated attention result as PACKED
// PLANES (the input format of s2s_node_linear, csrc/node_gemm.hip) and/or fp32 [B*N, D].
//
// A workgroup = 4 waves x 32 queries of one (sample, head); 32-key tiles of K and V (32 x 80 floats each) are double buffered in
// LDS (global -> VGPR before the tile's MFMAs, VGPR -> LDS after them).  Orientation as in ipa_attention.hip:
//   S^T[j, i] = K[j, :] . Q[i, :]   (A = key rows from LDS, B = the lane's query row in 40 registers; v_mfma_f32_32x32x2_f32)
//   O^T[c, i] += V^T[c, j] P^T[j, i] (A = value columns from LDS, B = the S^T accumulator itself, C layout == B layout)
// so a lane owns one query: max / sum / rescale are per-lane scalars (+ one cross-half exchange), and the accumulator registers
// 8u .. 8u+7 of output tile t are exactly fragment k-step 5 head + 2t + u of the packed-plane row (dh = 80 = 5 k-steps).
#include <hip/hip_runtime.h>
#include <math.h>

#include "range_flag.h"
#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int DH>
__global__ void __launch_bounds__(256) enc_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias,
                                                            float* __restrict__ out_f32, bf16x8* __restrict__ out_xp, int* range_flag, int B, int N,
                                                            int heads, float scale) {
    static_assert(DH == 80, "built for the reference configuration (d_model 320, 4 heads)");
    constexpr int KSd = DH + 4;          // padded LDS row stride (floats): conflict-free ds_read_b128 of the key rows
    constexpr int CT = (DH + 31) / 32;   // output tiles of 32 channels (the last one is partly padding)
    constexpr int HALF = DH / 2;         // channels per k-group of the QK^T contraction
    __shared__ __attribute__((aligned(16))) float s_k[2][32 * KSd];
    __shared__ __attribute__((aligned(16))) float s_v[2][32 * 96];   // rows padded to 96 channels (3 tiles)
    __shared__ float s_b[2][32];
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31, wave = threadIdx.x >> 6;
    const int n_qb = (N + 127) / 128;
    int bid = blockIdx.x;
    const int qb = bid % n_qb; bid /= n_qb;
    const int head = bid % heads;
    const int b = bid / heads;
    const int D = heads * DH;
    const int i = qb * 128 + wave * 32 + c;
    const bool ivalid = i < N;
    const long long row_i = (long long)b * N + (ivalid ? i : N - 1);
    const float* base = qkv + (long long)b * N * 3 * D + head * DH;

    // ---- tile staging: 32 keys x (K 80 + V 80) floats = 1280 float4, 5 per thread
    float4 st[5];
    auto tile_load = [&](int j0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = threadIdx.x + 256 * k;           // 0 .. 1279
            const int which = idx / 640, rem = idx % 640, r = rem / 20, c4 = rem % 20;
            const int j = min(j0 + r, N - 1);
            st[k] = *reinterpret_cast<const float4*>(base + (long long)j * 3 * D + D * (1 + which) + 4 * c4);
        }
    };
    auto tile_store = [&](int par, int j0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int which = idx / 640, rem = idx % 640, r = rem / 20, c4 = rem % 20;
            float* dst = which ? &s_v[par][r * 96 + 4 * c4] : &s_k[par][r * KSd + 4 * c4];
            *reinterpret_cast<float4*>(dst) = st[k];
        }
        if (threadIdx.x < 32) {
            const int j = j0 + threadIdx.x;
            s_b[par][threadIdx.x] = j < N ? (key_bias ? key_bias[(long long)b * N + j] : 0.f) : -INFINITY;
        }
    };
    // the padding channels 80..95 of the value rows are multiplied into output columns that are never stored: keep them finite
    for (int idx = threadIdx.x; idx < 2 * 32 * 16; idx += 256) s_v[idx / 512][((idx % 512) / 16) * 96 + 80 + idx % 16] = 0.f;

    tile_load(0);
    // ---- this lane's query row: channels HALF h + s, s = 0 .. HALF-1 (B operand of k-step s), pre-scaled like PyTorch (q * dh^-1/2)
    float qreg[HALF];
    {
        const float* qrow = qkv + row_i * 3 * D + head * DH + HALF * h;
#pragma unroll
        for (int s4 = 0; s4 < HALF / 4; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(qrow + 4 * s4);
            qreg[4 * s4 + 0] = v.x * scale; qreg[4 * s4 + 1] = v.y * scale; qreg[4 * s4 + 2] = v.z * scale; qreg[4 * s4 + 3] = v.w * scale;
        }
    }
    f32x16 O[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    tile_store(0, 0);
    __syncthreads();

    int cur = 0;
    for (int j0 = 0; j0 < N; j0 += 32, cur ^= 1) {
        const bool more = j0 + 32 < N;
        if (more) tile_load(j0 + 32);
        // ---- S^T = K . Q^T : two accumulation chains (a dependent fp32 MFMA issues only every ~64 cycles)
        f32x16 S, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f, S1[r] = 0.f;
        const float* krow = &s_k[cur][c * KSd + HALF * h];
#pragma unroll
        for (int s4 = 0; s4 < HALF / 4; ++s4) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * s4);
            S = mfma32(kf.x, qreg[4 * s4 + 0], S);
            S1 = mfma32(kf.y, qreg[4 * s4 + 1], S1);
            S = mfma32(kf.z, qreg[4 * s4 + 2], S);
            S1 = mfma32(kf.w, qreg[4 * s4 + 3], S1);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = (S[r] + S1[r]) + s_b[cur][rowmap(r, h)];
            S[r] = s;
            tmax = fmaxf(tmax, s);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;   // a fully masked prefix: exp(-inf - 0) = 0, no NaN
        const float alpha = expf(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(S[r] - m_use);
            S[r] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            f32x16 o = O[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            const float* vcol = &s_v[cur][32 * t + c + 4 * h * 96];
#pragma unroll
            for (int s = 0; s < 16; ++s) o = mfma32(vcol[((s & 3) + 8 * (s >> 2)) * 96], S[s], o);
            O[t] = o;
        }
        if (more) tile_store(cur ^ 1, j0 + 32);
        __syncthreads();
    }

    // ---- epilogue: register r of tile t = channel 32 t + (r&3) + 8 (r>>2) + 4 h of this lane's query
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (!ivalid) return;
    const long long row = (long long)b * N + i;
    if (out_f32) {
        float* o = out_f32 + row * D + head * DH;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int c0 = 32 * t + 8 * rq + 4 * h;
                if (c0 < DH)
                    *reinterpret_cast<float4*>(o + c0) = make_float4(O[t][4 * rq] * inv, O[t][4 * rq + 1] * inv, O[t][4 * rq + 2] * inv, O[t][4 * rq + 3] * inv);
            }
    }
    if (out_xp) {
        // packed planes of the [B*N, D] result: fragment (row tile, k-step 5 head + 2t + u, plane, lane 32 h + row % 32)
        const int KS = D / 16;
        bf16x8* o = out_xp + (((row >> 5) * KS + (DH / 16) * head) * 2) * 64 + 32 * h + (int)(row & 31);
        float amax = 0.f;   // range guard (range_flag.h)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (2 * t + u >= DH / 16) continue;
                // f16 pair planes of the node stream (x_h, x_l), csrc/node_gemm.hip
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                f16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = O[t][8 * u + j] * inv;
                    amax = s2s::range_max(amax, v);
                    asm volatile("" : "+v"(v));   // one materialised fp32 value feeds both planes (see node_gemm.hip split8_f16)
                    const _Float16 a_ = (_Float16)v;
                    ph[j] = a_; pl[j] = (_Float16)(v - (float)a_);
                }
                bf16x8* q = o + ((2 * t + u) * 2) * 64;
                q[0] = __builtin_bit_cast(bf16x8, ph); q[64] = __builtin_bit_cast(bf16x8, pl);
            }
        s2s::range_report(range_flag, amax, s2s::kRangeEncoderAttention);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same operator on split-f16 MFMA ("f16x3", the default arithmetic; see csrc/pair_mlp_f16.hip): every operand as two f16 numbers
// x = x_h + x_l, three v_mfma_f32_32x32x16_f16 per (k-step, tile) instead of eight v_mfma_f32_32x32x2_f32 at 1/16 of their rate
// -- 33 matrix instructions of 32 cycles per key tile and wave instead of 88 of 64.  Same workgroup shape, same online softmax in
// fp32; what changes is the operand plumbing:
//   K tile   -> LDS as A fragments  kf[k-step 5][plane][lane (key c, half h)][8] = channels 16 ks + 8 h + j   (split while staging)
//   V tile   -> LDS as A fragments of V^T  vf[tile 3][k-step u][plane][lane (channel c, half h)][8] = keys (j&3) + 8 (2u + (j>>2)) + 4 h,
//               the key order of the S^T accumulator registers 8u .. 8u+7: P^T needs no data movement to become the B operand
//   Q row    -> registers, B fragments of the 5 k-steps (split once per workgroup)
//   P        -> 2^10 p split in registers (the factor keeps the small part of small probabilities in f16's normal range; it is divided
//               out with the row sum at the end)
// RANGE: q, k, v are split here and feed the range maximum (range_flag.h, bit kRangeEncoderAttention).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// four fp32 values -> elements at .. at+3 of the planes (x_h, x_l): the 1.5-instruction split of csrc/pair_mlp_f16.hip (split4_f16)
__device__ __forceinline__ void enc_split4(float x0, float x1, float x2, float x3, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1, float& amax) {
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
}

// eight values -> one fragment pair (x_h, x_l)
__device__ __forceinline__ void enc_split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7, f16x8& ph, f16x8& pl,
                                           float& amax) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    enc_split4(x0, x1, x2, x3, h0, h1, l0, l1, amax);
    enc_split4(x4, x5, x6, x7, h2, h3, l2, l3, amax);
    ph = __builtin_bit_cast(f16x8, u32x4e{h0, h1, h2, h3});
    pl = __builtin_bit_cast(f16x8, u32x4e{l0, l1, l2, l3});
}

template <int DH>
__global__ void __launch_bounds__(256, 2) enc_attention_f16_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias,
                                                                float* __restrict__ out_f32, bf16x8* __restrict__ out_xp, int* range_flag, int B,
                                                                int N, int heads, float scale) {
    static_assert(DH == 80, "built for the reference configuration (d_model 320, 4 heads)");
    constexpr int KSQ = DH / 16;         // k-steps of the QK^T contraction (5)
    constexpr int CT = (DH + 31) / 32;   // output tiles of 32 channels (the last one is half padding)
    constexpr float kPS = 1024.0f;       // probabilities travel as 2^10 p
    __shared__ __attribute__((aligned(16))) f16x8 s_kf[2][KSQ * 2 * 64];      // 10 KiB per buffer
    __shared__ __attribute__((aligned(16))) f16x8 s_vf[2][CT * 2 * 2 * 64];   // 12 KiB per buffer
    __shared__ float s_b[2][32];
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31, wave = threadIdx.x >> 6;
    const int n_qb = (N + 127) / 128;
    int bid = blockIdx.x;
    const int qb = bid % n_qb; bid /= n_qb;
    const int head = bid % heads;
    const int b = bid / heads;
    const int D = heads * DH;
    const int i = qb * 128 + wave * 32 + c;
    const bool ivalid = i < N;
    const long long row_i = (long long)b * N + (ivalid ? i : N - 1);
    const float* base = qkv + (long long)b * N * 3 * D + head * DH;
    float amax = 0.f;   // range guard: every value split into f16 planes

    // ---- tile staging: 32 keys x (K 80 + V 80) floats = 1280 float4, 5 per thread (as in the fp32 kernel), split on the way into LDS
    float4 st[5];
    auto tile_load = [&](int j0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = threadIdx.x + 256 * k;           // 0 .. 1279
            const int which = idx / 640, rem = idx % 640, r = rem / 20, c4 = rem % 20;
            const int j = min(j0 + r, N - 1);
            st[k] = *reinterpret_cast<const float4*>(base + (long long)j * 3 * D + D * (1 + which) + 4 * c4);
        }
    };
    auto tile_store = [&](int par, int j0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int which = idx / 640, rem = idx % 640, r = rem / 20, c4 = rem % 20;   // key r of the tile, channels 4 c4 .. + 3
            unsigned h0, h1, l0, l1;
            enc_split4(st[k].x, st[k].y, st[k].z, st[k].w, h0, h1, l0, l1, amax);
            if (!which) {
                // K: fragment (k-step ch / 16, lane (key r, half (ch % 16) / 8)), elements ch % 8 .. + 3: one 8-byte store per plane
                const int ch = 4 * c4, ks = ch >> 4, hh = (ch >> 3) & 1, e = ch & 7;
                _Float16* ph = reinterpret_cast<_Float16*>(&s_kf[par][(ks * 2 + 0) * 64 + 32 * hh + r]) + e;
                _Float16* pl = reinterpret_cast<_Float16*>(&s_kf[par][(ks * 2 + 1) * 64 + 32 * hh + r]) + e;
                *reinterpret_cast<uint2*>(ph) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(pl) = make_uint2(l0, l1);
            } else {
                // V^T: key r sits at (k-step u = r >> 4, half (r >> 2) & 1, element (r & 3) + 4 ((r >> 3) & 1)) of the fragments of its
                // four channels (tile ch / 32, lane ch % 32): four 2-byte stores per plane
                const int u = r >> 4, hh = (r >> 2) & 1, e = (r & 3) + 4 * ((r >> 3) & 1);
                const unsigned hv[2] = {h0, h1}, lv[2] = {l0, l1};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 4 * c4 + q, t = ch >> 5, cc = ch & 31;
                    const unsigned short hb = (unsigned short)(hv[q >> 1] >> (16 * (q & 1))), lb = (unsigned short)(lv[q >> 1] >> (16 * (q & 1)));
                    reinterpret_cast<unsigned short*>(&s_vf[par][((t * 2 + u) * 2 + 0) * 64 + 32 * hh + cc])[e] = hb;
                    reinterpret_cast<unsigned short*>(&s_vf[par][((t * 2 + u) * 2 + 1) * 64 + 32 * hh + cc])[e] = lb;
                }
            }
        }
        if (threadIdx.x < 32) {
            const int j = j0 + threadIdx.x;
            s_b[par][threadIdx.x] = j < N ? (key_bias ? key_bias[(long long)b * N + j] : 0.f) : -INFINITY;
        }
    };
    // channels 80..95 of the last value tile are padding (their output columns are never stored): zero, in both buffers and planes
    for (int idx = threadIdx.x; idx < 2 * 2 * 2 * 64; idx += 256) {   // (buffer, u, plane, lane) of tile CT - 1
        const int par = idx >> 8, rest = idx & 255, ln = rest & 63;
        if ((ln & 31) >= DH - 32 * (CT - 1)) s_vf[par][((CT - 1) * 2 * 2) * 64 + rest] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }

    tile_load(0);
    // ---- this lane's query row as the B fragments of the 5 k-steps (channels 16 ks + 8 h + j), pre-scaled like PyTorch (q * dh^-1/2)
    f16x8 qh[KSQ], ql[KSQ];
    {
        const float* qrow = qkv + row_i * 3 * D + head * DH + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
            const float4 v0 = *reinterpret_cast<const float4*>(qrow + 16 * ks), v1 = *reinterpret_cast<const float4*>(qrow + 16 * ks + 4);
            enc_split8(v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale, qh[ks], ql[ks],
                       amax);
        }
    }
    f32x16 O[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    tile_store(0, 0);
    __syncthreads();

    int cur = 0;
    for (int j0 = 0; j0 < N; j0 += 32, cur ^= 1) {
        const bool more = j0 + 32 < N;
        if (more) tile_load(j0 + 32);
        // ---- S^T = K . Q^T on two accumulators (even / odd k-steps), three products each
        f32x16 S, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f, S1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
            const f16x8 kh = s_kf[cur][(ks * 2 + 0) * 64 + lane], kl = s_kf[cur][(ks * 2 + 1) * 64 + lane];
            f32x16& acc = (ks & 1) ? S1 : S;
            acc = mfma16(kl, qh[ks], acc);
            acc = mfma16(kh, ql[ks], acc);
            acc = mfma16(kh, qh[ks], acc);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sv = (S[r] + S1[r]) + s_b[cur][rowmap(r, h)];
            S[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;   // a fully masked prefix: exp(-inf - 0) = 0, no NaN
        const float alpha = expf(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(S[r] - m_use);
            S[r] = pv * kPS;
            psum += pv;
        }
        l_run = l_run * alpha + psum;
        // P^T as B fragments: k-step u = accumulator registers 8u .. 8u+7 (the key order the V^T fragments were written in)
        f16x8 ph[2], pl[2];
        float pmax = 0.f;   // (<= 2^10: not a range concern)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            enc_split8(S[8 * u + 0], S[8 * u + 1], S[8 * u + 2], S[8 * u + 3], S[8 * u + 4], S[8 * u + 5], S[8 * u + 6], S[8 * u + 7], ph[u], pl[u], pmax);
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            f32x16 o = O[t];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f16x8 vh = s_vf[cur][((t * 2 + u) * 2 + 0) * 64 + lane], vl = s_vf[cur][((t * 2 + u) * 2 + 1) * 64 + lane];
                o = mfma16(vl, ph[u], o);
                o = mfma16(vh, pl[u], o);
                o = mfma16(vh, ph[u], o);
            }
            O[t] = o;
        }
        if (more) tile_store(cur ^ 1, j0 + 32);
        __syncthreads();
    }

    // ---- epilogue: register r of tile t = channel 32 t + (r&3) + 8 (r>>2) + 4 h of this lane's query
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / (l_tot * kPS);
    // (every lane reports: invalid query lanes hold a valid row's data)
    if (ivalid) {
        const long long row = (long long)b * N + i;
        if (out_f32) {
            float* o = out_f32 + row * D + head * DH;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int c0 = 32 * t + 8 * rq + 4 * h;
                    if (c0 < DH)
                        *reinterpret_cast<float4*>(o + c0) = make_float4(O[t][4 * rq] * inv, O[t][4 * rq + 1] * inv, O[t][4 * rq + 2] * inv, O[t][4 * rq + 3] * inv);
                }
        }
        if (out_xp) {
            const int KS = D / 16;
            bf16x8* o = out_xp + (((row >> 5) * KS + (DH / 16) * head) * 2) * 64 + 32 * h + (int)(row & 31);
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (2 * t + u >= DH / 16) continue;
                    f16x8 oh, ol;
                    enc_split8(O[t][8 * u + 0] * inv, O[t][8 * u + 1] * inv, O[t][8 * u + 2] * inv, O[t][8 * u + 3] * inv, O[t][8 * u + 4] * inv,
                               O[t][8 * u + 5] * inv, O[t][8 * u + 6] * inv, O[t][8 * u + 7] * inv, oh, ol, amax);
                    bf16x8* q = o + ((2 * t + u) * 2) * 64;
                    q[0] = __builtin_bit_cast(bf16x8, oh); q[64] = __builtin_bit_cast(bf16x8, ol);
                }
        }
    }
    s2s::range_report(range_flag, amax, s2s::kRangeEncoderAttention);
}

}  // namespace

extern "C" int s2s_encoder_attention_f16x3(const float* qkv, const float* key_bias, float* out_f32, void* out_xp, int n_samples, int n_res,
                                           int n_heads, int head_dim, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (!qkv || (!out_f32 && !out_xp) || head_dim != 80 || n_heads < 1 || (n_heads * head_dim) % 32) return (int)hipErrorInvalidValue;
    const long long blocks = (long long)n_samples * n_heads * ((n_res + 127) / 128);
    hipLaunchKernelGGL((enc_attention_f16_kernel<80>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv, key_bias, out_f32,
                       (bf16x8*)out_xp, s2s::g_range_flag, n_samples, n_res, n_heads, 1.0f / sqrtf((float)head_dim));
    return (int)hipGetLastError();
}

extern "C" int s2s_encoder_attention(const float* qkv, const float* key_bias, float* out_f32, void* out_xp, int n_samples, int n_res,
                                     int n_heads, int head_dim, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (!qkv || (!out_f32 && !out_xp) || head_dim != 80 || n_heads < 1 || (n_heads * head_dim) % 32) return (int)hipErrorInvalidValue;
    const long long blocks = (long long)n_samples * n_heads * ((n_res + 127) / 128);
    hipLaunchKernelGGL((enc_attention_kernel<80>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv, key_bias, out_f32,
                       (bf16x8*)out_xp, s2s::g_range_flag, n_samples, n_res, n_heads, 1.0f / sqrtf((float)head_dim));
    return (int)hipGetLastError();
}
