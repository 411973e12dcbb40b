// Invariant Point Attention core (gfx950): logits -> masked softmax -> value aggregation, fused.
// Reference: InvariantPointAttention.forward, src/models/net/ipa.py:183-257 (the part between the
// input projections and linear_out).  The [B,N,N,H,Pq,3] displacement tensor (6.4 GB at B=128,
// N=256), the [B,H,N,N] attention matrix and the [B,H,3,N,N,Pv] product (9.7 GB) that eager
// PyTorch materialises are never formed: a workgroup of 4 waves owns (sample, head, 128 query residues),
// one wave per 32 of them, and streams 32-residue key tiles with an online softmax (flash-attention
// schedule).  The key tile (K, V, value points, key points, key mask: 77 KB) is fetched ONCE per
// workgroup straight into LDS by the LDS-DMA (global_load_lds, no VGPR round trip), double buffered,
// one barrier per tile, so all four waves (and the other workgroup sharing the head through L2) reuse it.
//
// MFMA orientation (QK^T: v_mfma_f32_32x32x2_f32, exact fp32; PV: v_mfma_f32_32x32x16_bf16 on the exact 3-way bf16 split of
// both operands, six plane-pair products, fp32 accumulate = fp32-equivalent, see pair_mlp_bf16.hip):
//   S^T[j, i] = K[j, :] . Q[i, :]      A = key tile (row j per lane), B = Q held in registers
//   O^T[c, i] += V^T[c, j] . P^T[j, i] A = value columns (coalesced 128 B per half wave),
//                                       B = the S^T accumulator itself (C layout == B layout,
//                                       k-order of the dot product is free)
//   so a lane owns ONE query residue i: softmax statistics, the running rescale, the point term
//   and the o_pair accumulation are per-lane scalars; only max/sum need one cross-half exchange.
// LDS images: K rows padded to C+4 floats so the 16 lanes of a ds_read_b128 group hit 16 different bank
// quads; V / value points row-major (ds_read_b32, 32 consecutive floats per half wave); key points and
// mask are read as broadcasts.  The VALU work (point distances, o_pair) is written inside the MFMA loops so
// it issues in the shadow of the matrix pipe.
// The pair term  o_pair[i,h,:] = sum_j a_ij pair_z[i,j,:]  is NOT done here: its "value" depends on the query, so every
// head would gather the same 128 B rows of pair_z per lane (8x the traffic, exposed HBM latency).  This kernel stores
// the masked logits (same [B,H,N,N] layout as the bias it reads, may alias it) and the final softmax statistics;
// ipa_opair_kernel below streams pair_z once and applies all heads at a time.
// Point term  -1/2 * softplus(w_h)*c * sum_p |q_ip - k_jp|^2  is evaluated with explicit differences on
// the VALU exactly as the reference forms it — not through the |q|^2+|k|^2-2q.k expansion, which loses
// ~2 digits to cancellation.  Output is written directly in linear_out's concat order (ipa.py:259-266):
//   [ o (H*C) | o_pt.x (H*Pv) | o_pt.y | o_pt.z | |o_pt| (H*Pv) | o_pair (H*PZ) ]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_b16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LDS read pointers made opaque at the point of use: hipcc otherwise hoists every LDS read of a key tile (all 16 rows of
// key points, all K fragments) to the top of the loop body -- there is no store in between that it can see -- and the
// 400+ live values push the accumulators out to scratch.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
__device__ __forceinline__ lds_cf* lds_pin(const float* p) {
    lds_cf* q = (lds_cf*)p;
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ float4 lds_ld4(lds_cf* p) {
    const f32x4 v = *(lds_cf4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

struct IpaArgs {
    const float* q;         // [B,N,H,C]
    const float* kv;        // [B,N,H,2C]  (k = first C, v = last C of every head; ipa.py:132-141)
    const float* q_pts;     // [B,N,H,PQ*3] global frame
    const float* k_pts;     // [B,N,H,PQ*3]
    const float* v_pts;     // [B,N,H,64]  (x,y,z,0) per point, zero padded
    const float* attn_bias; // [B,H,N,N]   linear_b(z), head-major
    float* logits;          // [B,H,N,N]   out: masked logits (may alias attn_bias)
    float* stats;           // [B,H,N,2]   out: running max, sum of exp of every query row
    const float* mask;      // [B,N]
    const float* rigids7;   // [B,N,7]     frames (scaled translation) for the inverse transform
    const float* head_w;    // [H]         softplus(head_weights) * sqrt(1/(3*(PQ*9/2)))
    float* out;             // [B,N,H*(C+4*PV+PZ)]
    int B, N, H;
    float inf, eps;
    int xcd_remap;  // 1: give every XCD (private L2) a contiguous range of (sample, head, query block) ids
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    // 16 B per lane, destination = wave-uniform base + lane*16 (LDS-DMA semantics)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// Timeline probe (tools/ipa_probe.py; only in -DS2S_IPA_PROBE=<block> builds)
#ifdef S2S_IPA_PROBE
__device__ unsigned long long s2s_ipa_probe[4][128];
#define IPROBE(idx) do { if (blockIdx.x == S2S_IPA_PROBE) s2s_ipa_probe[threadIdx.x >> 6][idx] = __builtin_readcyclecounter(); } while (0)
#else
#define IPROBE(idx) do { } while (0)
#endif

template <int C, int PQ>
struct KeyStage {
    static constexpr int KS = C + 4;  // padded K row stride (floats)
    float k[32 * KS];
    float v[32 * C];
    float vp[32 * 64];
    float kp[32 * PQ * 3];
    float km[32];
};

// FULL: N is a multiple of 32 -> no ragged-tile code (and no control flow) inside the key loop
template <int C, int PQ, int PV, int PZ, bool FULL>
__global__ void __launch_bounds__(256) ipa_attention_kernel(IpaArgs a) {
    static_assert(C == 256 && PV <= 16 && PZ % 4 == 0 && (PQ * 3) % 4 == 0 && 32 * PQ * 3 <= 3 * 256, "shape");
    using Stage = KeyStage<C, PQ>;
    constexpr int KS = Stage::KS;
    constexpr int CT = C / 32;      // value tiles
    constexpr int OT = CT + 2;      // + two tiles of packed value points
    constexpr int QG = C / 8;       // float4 groups of Q per lane
    __shared__ __attribute__((aligned(16))) Stage stage[2];

    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = threadIdx.x >> 6;
    const int N = a.N, H = a.H;
    const int n_qb = (N + 127) / 128;
    int bid = blockIdx.x;
    // Workgroup b runs on XCD b % 8 (observed dispatch order; a speed assumption only).  The query blocks of one (sample, head)
    // read the same K / V / point tiles: with consecutive LOGICAL ids on one XCD they run side by side there and the second
    // reader hits that XCD's L2 instead of fetching the tiles from HBM again.
    if (a.xcd_remap) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qb = bid % n_qb; bid /= n_qb;
    const int head = bid % H;
    const int b = bid / H;
    const int i = qb * 128 + wave * 32 + c;
    const bool ivalid = i < N;
    const int ic = ivalid ? i : N - 1;
    const long long row_i = (long long)b * N + ic;
    const long long kvrow_stride = (long long)H * 2 * C;
    const float* kv_bh = a.kv + (long long)b * N * kvrow_stride + (long long)head * 2 * C;

    // cooperative LDS-DMA of the key tile starting at key j0 into stage st (19 x 1 KiB per wave)
    auto load_tile = [&](Stage& st, int j0) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr;
            const float* src = kv_bh + (long long)min(j0 + row, N - 1) * kvrow_stride + lane * 4;
            dma16(src, st.k + row * KS);
            dma16(src + C, st.v + row * C);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int row0 = wave * 8 + x * 4;
            const int row = row0 + (lane >> 4);
            dma16(a.v_pts + (((long long)b * N + min(j0 + row, N - 1)) * H + head) * 64 + (lane & 15) * 4, st.vp + row0 * 64);
        }
        if (wave < 3) {
            const int f = (wave * 64 + lane) * 4;
            if (f < 32 * PQ * 3) {
                const int row = f / (PQ * 3), col = f % (PQ * 3);
                dma16(a.k_pts + (((long long)b * N + min(j0 + row, N - 1)) * H + head) * (PQ * 3) + col, st.kp + wave * 256);
            }
        } else if (lane < 32) {
            st.km[lane] = a.mask[(long long)b * N + min(j0 + lane, N - 1)];
        }
    };

    load_tile(stage[0], 0);

    // ---- this lane's query row (B operand of QK^T), points and mask
    // half of the query row lives in registers; the other half is re-read (L1/L2 hits) through a small
    // ring every key tile: 64 fewer live VGPRs is what keeps this kernel out of scratch
#ifndef S2S_IPA_QR
#define S2S_IPA_QR 0
#endif
    constexpr int QR = S2S_IPA_QR == 0 ? 0 : QG / S2S_IPA_QR;
    const float* qrow = a.q + (row_i * H + head) * C + 4 * h;
    float4 qreg[QR > 0 ? QR : 1];
#pragma unroll
    for (int g = 0; g < QR; ++g) qreg[g] = *reinterpret_cast<const float4*>(qrow + 8 * g);
    float qpt[PQ * 3];
    {
        const float* p = a.q_pts + (row_i * H + head) * (PQ * 3);
#pragma unroll
        for (int x = 0; x < PQ * 3; ++x) qpt[x] = p[x];
    }
    const float mask_i = a.mask[row_i];
    const float hw = a.head_w[head];
    const float c1 = sqrtf(1.0f / (3 * C));
    const float c2 = sqrtf(1.0f / 3);

    f32x16 O[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int j0 = 0; j0 < N; j0 += 32, cur ^= 1) {
        const Stage& st = stage[cur];
        IPROBE(8 * (j0 >> 5) + 0);
        // this lane's 16 bias values (keys j0 + 8g + 4h + e): one 128 B line per (query, tile), issued a whole
        // QK^T loop ahead of their use
        float4 bias4[4];
        const long long brow = (((long long)b * H + head) * N + ic) * N + j0 + 4 * h;
        const bool full_tile = FULL || j0 + 32 <= N;  // wave-uniform
        if (full_tile) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const float4*>(a.attn_bias + brow + 8 * g);
        } else {  // ragged last tile: element by element, clamped
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jg = j0 + 8 * g + 4 * h;
                const float* bp = a.attn_bias + brow + 8 * g - jg;  // row start
                bias4[g].x = bp[min(jg, N - 1)];
                bias4[g].y = bp[min(jg + 1, N - 1)];
                bias4[g].z = bp[min(jg + 2, N - 1)];
                bias4[g].w = bp[min(jg + 3, N - 1)];
            }
        }

        // ---------------- S^T = K . Q^T, with the point distances issued between the MFMAs.
        // Explicit software pipeline, one scheduling region per 4-MFMA group (hipcc otherwise serialises the whole
        // VALU point term behind the MFMAs and waits on every operand right where it is loaded): the K fragment of
        // group g+1, the ring slot of Q four groups ahead and the key points of the next key row are fetched while
        // group g's MFMAs run.
        f32x16 S, S1;  // two accumulation chains (even / odd groups): a dependent fp32 MFMA issues only every ~83 cycles
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f, S1[r] = 0.f;
        float pt[16];
        constexpr int QD = 4;
        float4 qring[QD];
#pragma unroll
        for (int d = 0; d < QD; ++d) qring[d] = *reinterpret_cast<const float4*>(qrow + 8 * (QR + d));
        float4 kf_next = lds_ld4(lds_pin(st.k + c * KS + 4 * h));
        float4 kpv[PQ * 3 / 4];  // key points of the key row whose distance term is evaluated next
        {
            lds_cf* kp = lds_pin(st.kp + rowmap(0, h) * (PQ * 3));
#pragma unroll
            for (int x = 0; x < PQ * 3 / 4; ++x) kpv[x] = lds_ld4(kp + 4 * x);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float4 kf = kf_next;
            if (g + 1 < QG) kf_next = lds_ld4(lds_pin(st.k + c * KS + 8 * (g + 1) + 4 * h));
            float4 qf;
            if (g < QR) {
                qf = qreg[g];
            } else {
                qf = qring[(g - QR) % QD];
                if (g + QD < QG) qring[(g - QR) % QD] = *reinterpret_cast<const float4*>(qrow + 8 * (g + QD));
            }
            S = mfma32(kf.x, qf.x, S);
            S1 = mfma32(kf.y, qf.y, S1);
            S = mfma32(kf.z, qf.z, S);
            S1 = mfma32(kf.w, qf.w, S1);
            if ((g & 1) == 0) {
                const int r = g >> 1;
                const float kpf[PQ * 3] = {kpv[0].x, kpv[0].y, kpv[0].z, kpv[0].w, kpv[1].x, kpv[1].y, kpv[1].z, kpv[1].w,
                                           kpv[2].x, kpv[2].y, kpv[2].z, kpv[2].w, kpv[3].x, kpv[3].y, kpv[3].z, kpv[3].w,
                                           kpv[4].x, kpv[4].y, kpv[4].z, kpv[4].w, kpv[5].x, kpv[5].y, kpv[5].z, kpv[5].w};
                float acc = 0.f;
#pragma unroll
                for (int p = 0; p < PQ; ++p) {
                    const float dx = qpt[p * 3 + 0] - kpf[p * 3 + 0];
                    const float dy = qpt[p * 3 + 1] - kpf[p * 3 + 1];
                    const float dz = qpt[p * 3 + 2] - kpf[p * 3 + 2];
                    acc += (dx * dx + dy * dy + dz * dz) * hw;
                }
                asm volatile("" : "+v"(acc));  // evaluated HERE, under this group's MFMAs (not sunk behind the loop)
                pt[r] = acc;
            } else if (g + 1 < QG) {
                lds_cf* kp = lds_pin(st.kp + rowmap((g + 1) >> 1, h) * (PQ * 3));
#pragma unroll
                for (int x = 0; x < PQ * 3 / 4; ++x) kpv[x] = lds_ld4(kp + 4 * x);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        IPROBE(8 * (j0 >> 5) + 1);
        // VMEM loads return in order: the next tile's DMA is issued only now, behind the last load of this tile (bias, Q
        // ring), so that nothing consumed during this tile has to wait for 77 KB of DMA; it lands under softmax + PV,
        // which touch no global memory loads (and must stay free of scratch reloads for the same reason).
        if (j0 + 32 < N) load_tile(stage[cur ^ 1], j0 + 32);
        IPROBE(8 * (j0 >> 5) + 2);
        // ---------------- logits (ipa.py:183-214)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + rowmap(r, h);
            const float4 b4 = bias4[r >> 2];
            const float bias = (r & 3) == 0 ? b4.x : ((r & 3) == 1 ? b4.y : ((r & 3) == 2 ? b4.z : b4.w));
            const float sq = a.inf * (mask_i * st.km[rowmap(r, h)] - 1.0f);
            float s = (S[r] + S1[r]) * c1 + c2 * bias;
            s = s + pt[r] * (-0.5f);
            s = s + sq;
            s = (j < N) ? s : -INFINITY;
            S[r] = s;
            tmax = fmaxf(tmax, s);
        }
        __builtin_amdgcn_sched_barrier(0);
        IPROBE(64 + 4 * (j0 >> 5) + 0);
        if (ivalid) {
            if (full_tile) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(a.logits + brow + 8 * g) = make_float4(S[4 * g], S[4 * g + 1], S[4 * g + 2], S[4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j0 + rowmap(r, h) < N) a.logits[brow - 4 * h + rowmap(r, h)] = S[r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        IPROBE(64 + 4 * (j0 >> 5) + 1);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);  // 0 on the first tile, 1 once the maximum has settled
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(S[r] - m_new);
            S[r] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
        __builtin_amdgcn_sched_barrier(0);
        IPROBE(64 + 4 * (j0 >> 5) + 2);

        IPROBE(8 * (j0 >> 5) + 3);
        // ---------------- O^T += V^T . P^T (+ value points)  (ipa.py:221-252), one output tile at a time; the A operands of
        // tile t+1 are fetched from LDS and split, and its accumulator is rescaled by alpha (16 VALU multiplies,
        // unconditionally: a wave-uniform branch around them brought scratch spills back and ran 10 % slower) while the
        // MFMAs of tile t run.
        auto fetch_v = [&](int t, float (&dst)[16]) {
            lds_cf* base = lds_pin(t < CT ? st.v + 32 * t + c + 4 * h * C : st.vp + 32 * (t - CT) + c + 4 * h * 64);
            const int rs = t < CT ? C : 64;  // row stride; key row of register r = (r&3) + 8(r>>2) (+ 4h, in the base)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = base[((r & 3) + 8 * (r >> 2)) * rs];
        };
        // PV runs on the bf16 matrix cores with the exact 3-way split of pair_mlp_bf16.hip (fp32-equivalent: six plane-pair
        // products, fp32 accumulate): 12 x 32-cycle MFMAs per output tile instead of 16 x ~82-cycle fp32 MFMAs.  The
        // k order of a k-step u is the accumulator order, element j <-> key row of register 8u+j, for both operands; the
        // split of the next tile's V^T fragments (88 VALU) rides under the current tile's MFMAs.
        bf16x8 pp[2][3];   // P^T planes (B operand), k-steps u = 0, 1
        auto split8 = [&](const float* v, bf16x8 (&d)[3]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 a_ = (__bf16)v[j];
                const float r1 = v[j] - (float)a_;
                const __bf16 b_ = (__bf16)r1;
                const float r2 = r1 - (float)b_;
                d[0][j] = a_; d[1][j] = b_; d[2][j] = (__bf16)r2;
            }
        };
        {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = S[r];
            split8(pv, pp[0]);
            split8(pv + 8, pp[1]);
        }
        float va[16];
        bf16x8 av[2][2][3];  // [buffer][k-step][plane] of V^T fragments (A operand)
        fetch_v(0, va);
        split8(va, av[0][0]);
        split8(va + 8, av[0][1]);
        fetch_v(1, va);
#pragma unroll
        for (int r = 0; r < 16; ++r) O[0][r] *= alpha;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < OT; ++t) {
            const bf16x8 (&a0)[3] = av[t & 1][0], (&a1)[3] = av[t & 1][1];
            f32x16 o = O[t];
            o = mfma_b16(a0[2], pp[0][0], o); o = mfma_b16(a0[0], pp[0][2], o); o = mfma_b16(a0[1], pp[0][1], o);
            o = mfma_b16(a0[1], pp[0][0], o); o = mfma_b16(a0[0], pp[0][1], o); o = mfma_b16(a0[0], pp[0][0], o);
            o = mfma_b16(a1[2], pp[1][0], o); o = mfma_b16(a1[0], pp[1][2], o); o = mfma_b16(a1[1], pp[1][1], o);
            o = mfma_b16(a1[1], pp[1][0], o); o = mfma_b16(a1[0], pp[1][1], o); o = mfma_b16(a1[0], pp[1][0], o);
            O[t] = o;
            if (t + 1 < OT) {
                bf16x8 (&n0)[3] = av[(t + 1) & 1][0], (&n1)[3] = av[(t + 1) & 1][1];
                split8(va, n0);
                split8(va + 8, n1);
                asm volatile("" : "+v"(n0[0]), "+v"(n0[1]), "+v"(n0[2]), "+v"(n1[0]), "+v"(n1[1]), "+v"(n1[2]));
                if (t + 2 < OT) fetch_v(t + 2, va);
#pragma unroll
                for (int r = 0; r < 16; ++r) O[t + 1][r] *= alpha;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        IPROBE(8 * (j0 >> 5) + 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile's DMA has landed
        IPROBE(8 * (j0 >> 5) + 5);
        __syncthreads();                                   // ... and everyone is done reading this one
        IPROBE(8 * (j0 >> 5) + 6);
    }

    IPROBE(120);
    // ---------------- epilogue: normalise, inverse-transform points, write concat layout
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int feat = H * (C + 4 * PV + PZ);
    float* orow = a.out + row_i * feat;
    {
        // o: transpose through LDS (the key stages are dead; the last tile's barrier has passed) so that every query row
        // leaves as ONE coalesced 1 KiB store per wave instead of 32 scattered 16 B stores per lane (store-issue bound).
        // Each wave owns 32 rows x (C + 4) floats of the stage memory.
        float* tr = reinterpret_cast<float*>(&stage[0]) + wave * 32 * KS;
        static_assert(4 * 32 * (C + 4) * 4 <= (int)sizeof(stage), "transpose buffer fits in the stages");
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(tr + c * KS + 32 * t + 8 * rq + 4 * h) =
                    make_float4(O[t][4 * rq] * inv, O[t][4 * rq + 1] * inv, O[t][4 * rq + 2] * inv, O[t][4 * rq + 3] * inv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int i0 = qb * 128 + wave * 32;
        for (int qq = 0; qq < 32; ++qq) {
            if (i0 + qq >= N) break;  // wave-uniform
            const float4 v = *reinterpret_cast<const float4*>(tr + qq * KS + 4 * lane);
            *reinterpret_cast<float4*>(a.out + ((long long)b * N + i0 + qq) * feat + head * C + 4 * lane) = v;
        }
    }
    {
        // frame of residue i: R = quat_to_rot(q) (rigid_utils.py:187-207), o_pt = R^T (x - t) (:1122-1133)
        const float* f = a.rigids7 + row_i * 7;
        const float qa = f[0], qb_ = f[1], qc = f[2], qd = f[3];
        const float tx = f[4], ty = f[5], tz = f[6];
        const float r00 = qa * qa + qb_ * qb_ - qc * qc - qd * qd, r01 = 2 * qb_ * qc - 2 * qa * qd, r02 = 2 * qb_ * qd + 2 * qa * qc;
        const float r10 = 2 * qb_ * qc + 2 * qa * qd, r11 = qa * qa - qb_ * qb_ + qc * qc - qd * qd, r12 = 2 * qc * qd - 2 * qa * qb_;
        const float r20 = 2 * qb_ * qd - 2 * qa * qc, r21 = 2 * qc * qd + 2 * qa * qb_, r22 = qa * qa - qb_ * qb_ - qc * qc + qd * qd;
        float* ox = orow + H * C + head * PV;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt_idx = 8 * t + 2 * rq + h;  // point whose (x,y,z,0) group this lane holds
                const float dx = O[CT + t][4 * rq + 0] * inv - tx;
                const float dy = O[CT + t][4 * rq + 1] * inv - ty;
                const float dz = O[CT + t][4 * rq + 2] * inv - tz;
                const float lx = r00 * dx + r10 * dy + r20 * dz;
                const float ly = r01 * dx + r11 * dy + r21 * dz;
                const float lz = r02 * dx + r12 * dy + r22 * dz;
                const float nr = sqrtf(lx * lx + ly * ly + lz * lz + a.eps);
                if (ivalid && pt_idx < PV) {
                    ox[pt_idx] = lx;
                    ox[H * PV + pt_idx] = ly;
                    ox[2 * H * PV + pt_idx] = lz;
                    ox[3 * H * PV + pt_idx] = nr;
                }
            }
    }
    IPROBE(121);
    if (ivalid && h == 0) {
        float* st2 = a.stats + ((((long long)b * H + head) * N) + i) * 2;
        st2[0] = m_run;
        st2[1] = l_tot;
    }
}

// o_pair[b,i,h,:] = sum_j softmax_j(logits[b,h,i,:]) * pair_z[b,i,j,:]   (ipa.py:253-257), written into the concat row.
// One wave per (b,i): the 8 x N probabilities go through LDS once, pair_z rows (128 B) are streamed exactly once from HBM
// as float4 (8 lanes per row, 8 rows per instruction); lane (jsub = lane>>3, c4 = lane&7) accumulates 4 channels x 8 heads
// over the rows j = jsub (mod 8) and the 8 partial sums meet in a xor-shuffle tree.
template <int H, int PZ>
__global__ void __launch_bounds__(256) ipa_opair_kernel(const float* __restrict__ logits, const float* __restrict__ stats,
                                                        const float* __restrict__ pair_z, float* __restrict__ out, int B, int N,
                                                        int feat, int col0, int LD) {
    // LD: rows per (sample, head) slab and row stride of the logits ([B,H,LD,LD]; LD = N unless the attention kernel padded them)
    static_assert(H == 8 && PZ == 32, "shape");
    constexpr int CH = 256;  // keys per chunk
    __shared__ __attribute__((aligned(16))) float a_s[4][CH * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;  // b*N + i
    if (row >= (long long)B * N) return;                    // wave-uniform; no workgroup barriers below
    const long long b = row / N, i = row - b * N;
    float m[H], inv[H];
#pragma unroll
    for (int hd = 0; hd < H; ++hd) {
        const float* st = stats + (((b * H + hd) * N) + i) * 2;
        m[hd] = st[0];
        inv[hd] = 1.0f / st[1];
    }
    float acc[H][4];
#pragma unroll
    for (int hd = 0; hd < H; ++hd)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[hd][e] = 0.f;
    float* as = a_s[wave];
    const int jsub = lane >> 3, c4 = lane & 7;
    for (int j0 = 0; j0 < N; j0 += CH) {
        // probabilities of this chunk -> LDS as [key][head]
        const int jl = 4 * lane;  // this lane's 4 keys
        float pr[H][4];
#pragma unroll
        for (int hd = 0; hd < H; ++hd) {
            const float* lrow = logits + ((b * H + hd) * LD + i) * LD + j0 + jl;
            float4 s4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (j0 + jl + 3 < N) s4 = *reinterpret_cast<const float4*>(lrow);
            else {
                if (j0 + jl < N) s4.x = lrow[0];
                if (j0 + jl + 1 < N) s4.y = lrow[1];
                if (j0 + jl + 2 < N) s4.z = lrow[2];
            }
            pr[hd][0] = expf(s4.x - m[hd]) * inv[hd]; pr[hd][1] = expf(s4.y - m[hd]) * inv[hd];
            pr[hd][2] = expf(s4.z - m[hd]) * inv[hd]; pr[hd][3] = expf(s4.w - m[hd]) * inv[hd];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<float4*>(as + (jl + e) * H) = make_float4(pr[0][e], pr[1][e], pr[2][e], pr[3][e]);
            *reinterpret_cast<float4*>(as + (jl + e) * H + 4) = make_float4(pr[4][e], pr[5][e], pr[6][e], pr[7][e]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int jn = min(CH, N - j0);
        const float* pz = pair_z + (row * N + j0) * PZ + 4 * c4;
        for (int j = jsub; j < jn; j += 8) {
            const float4 z = *reinterpret_cast<const float4*>(pz + (long long)j * PZ);
            const float4 a0 = *reinterpret_cast<const float4*>(as + j * H), a1 = *reinterpret_cast<const float4*>(as + j * H + 4);
            const float av[H] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int hd = 0; hd < H; ++hd) {
                acc[hd][0] += av[hd] * z.x; acc[hd][1] += av[hd] * z.y; acc[hd][2] += av[hd] * z.z; acc[hd][3] += av[hd] * z.w;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the chunk image is reused
    }
#pragma unroll
    for (int hd = 0; hd < H; ++hd)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[hd][e];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[hd][e] = v;
        }
    if (jsub == 0) {
        float* o = out + row * feat + col0 + 4 * c4;
#pragma unroll
        for (int hd = 0; hd < H; ++hd)
            *reinterpret_cast<float4*>(o + hd * PZ) = make_float4(acc[hd][0], acc[hd][1], acc[hd][2], acc[hd][3]);
    }
}

}  // namespace

#ifdef S2S_IPA_PROBE
extern "C" int s2s_debug_read_ipa_probe(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(s2s_ipa_probe), sizeof(s2s_ipa_probe));
}
#endif

extern "C" int s2s_ipa_opair(const float* logits, const float* stats, const float* pair_z, float* out, int n_samples, int n_res,
                             int n_heads, int c_pair_z, int out_row_stride, int out_col_offset, int logits_ld, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (logits_ld <= 0) logits_ld = n_res;
    if (n_heads != 8 || c_pair_z != 32 || logits_ld < n_res) return (int)hipErrorInvalidValue;
    const long long rows = (long long)n_samples * n_res;
    hipLaunchKernelGGL((ipa_opair_kernel<8, 32>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, stats,
                       pair_z, out, n_samples, n_res, out_row_stride, out_col_offset, logits_ld);
    return (int)hipGetLastError();
}

extern "C" int s2s_ipa_attention(const float* q, const float* kv, const float* q_pts, const float* k_pts, const float* v_pts64,
                                 const float* attn_bias, float* logits_out, float* stats_out, const float* mask,
                                 const float* rigids7, const float* head_w_scaled, float* out, int n_samples, int n_res, int n_heads, int c_hidden,
                                 int n_qk_points, int n_v_points, int c_pair_z, float inf, float eps, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (c_hidden != 256 || n_qk_points != 8 || n_v_points != 12 || c_pair_z != 32 || n_heads < 1)
        return (int)hipErrorInvalidValue;  // the reference configuration (configs/model/diffusion.yaml:29-40)
    const int n_qb = (n_res + 127) / 128;
    const long long blocks = (long long)n_samples * n_heads * n_qb;
    static const int remap_env = getenv("S2S_IPA_XCD") ? atoi(getenv("S2S_IPA_XCD")) : 1;
    IpaArgs a{q, kv, q_pts, k_pts, v_pts64, attn_bias, logits_out, stats_out, mask, rigids7, head_w_scaled, out, n_samples, n_res, n_heads, inf, eps,
              (remap_env && blocks % 8 == 0 && n_qb > 1) ? 1 : 0};
    if (n_res % 32 == 0)
        hipLaunchKernelGGL((ipa_attention_kernel<256, 8, 12, 32, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((ipa_attention_kernel<256, 8, 12, 32, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
