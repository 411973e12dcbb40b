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
// MFMA orientation (v_mfma_f32_32x32x2_f32, exact fp32):
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
// Point term  -1/2 * softplus(w_h)*c * sum_p |q_ip - k_jp|^2  is evaluated with explicit differences on
// the VALU exactly as the reference forms it — not through the |q|^2+|k|^2-2q.k expansion, which loses
// ~2 digits to cancellation.  o_pair[i,h,:] = sum_j a_ij pair_z[i,j,:] is not a GEMM (the "value"
// depends on i): VALU too.  Output is written directly in linear_out's concat order (ipa.py:259-266):
//   [ o (H*C) | o_pt.x (H*Pv) | o_pt.y | o_pt.z | |o_pt| (H*Pv) | o_pair (H*PZ) ]
#include <hip/hip_runtime.h>
#include <math.h>

#include "str2str_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct IpaArgs {
    const float* q;         // [B,N,H,C]
    const float* kv;        // [B,N,H,2C]  (k = first C, v = last C of every head; ipa.py:132-141)
    const float* q_pts;     // [B,N,H,PQ*3] global frame
    const float* k_pts;     // [B,N,H,PQ*3]
    const float* v_pts;     // [B,N,H,64]  (x,y,z,0) per point, zero padded
    const float* attn_bias; // [B,N,N,H]   linear_b(z)
    const float* pair_z;    // [B,N,N,PZ]  down_z(z)
    const float* mask;      // [B,N]
    const float* rigids7;   // [B,N,7]     frames (scaled translation) for the inverse transform
    const float* head_w;    // [H]         softplus(head_weights) * sqrt(1/(3*(PQ*9/2)))
    float* out;             // [B,N,H*(C+4*PV+PZ)]
    int B, N, H;
    float inf, eps;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    // 16 B per lane, destination = wave-uniform base + lane*16 (LDS-DMA semantics)
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

template <int C, int PQ>
struct KeyStage {
    static constexpr int KS = C + 4;  // padded K row stride (floats)
    float k[32 * KS];
    float v[32 * C];
    float vp[32 * 64];
    float kp[32 * PQ * 3];
    float km[32];
};

template <int C, int PQ, int PV, int PZ>
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
    constexpr int QR = QG / 2;
    const float* qrow = a.q + (row_i * H + head) * C + 4 * h;
    float4 qreg[QR];
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
    float opair[PZ];
#pragma unroll
    for (int x = 0; x < PZ; ++x) opair[x] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int j0 = 0; j0 < N; j0 += 32, cur ^= 1) {
        const Stage& st = stage[cur];
        if (j0 + 32 < N) load_tile(stage[cur ^ 1], j0 + 32);  // in flight during this tile's compute

        // ---------------- S^T = K . Q^T, with the point distances issued between the MFMAs
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        float pt[16];
        constexpr int QD = 4;
        float4 qring[QD];
#pragma unroll
        for (int d = 0; d < QD; ++d) qring[d] = *reinterpret_cast<const float4*>(qrow + 8 * (QR + d));
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float4 kf = *reinterpret_cast<const float4*>(st.k + c * KS + 8 * g + 4 * h);
            float4 qf;
            if (g < QR) {
                qf = qreg[g];
            } else {
                qf = qring[(g - QR) % QD];
                if (g + QD < QG) qring[(g - QR) % QD] = *reinterpret_cast<const float4*>(qrow + 8 * (g + QD));
            }
            S = mfma32(kf.x, qf.x, S);
            S = mfma32(kf.y, qf.y, S);
            S = mfma32(kf.z, qf.z, S);
            S = mfma32(kf.w, qf.w, S);
            if ((g & 1) == 0) {
                const int r = g >> 1;
                const float* kp = st.kp + rowmap(r, h) * (PQ * 3);
                float acc = 0.f;
#pragma unroll
                for (int p = 0; p < PQ; ++p) {
                    const float dx = qpt[p * 3 + 0] - kp[p * 3 + 0];
                    const float dy = qpt[p * 3 + 1] - kp[p * 3 + 1];
                    const float dz = qpt[p * 3 + 2] - kp[p * 3 + 2];
                    acc += (dx * dx + dy * dy + dz * dz) * hw;
                }
                pt[r] = acc;
            }
        }
        // ---------------- logits (ipa.py:183-214)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + rowmap(r, h);
            const int jc = min(j, N - 1);
            const float bias = a.attn_bias[(row_i * N + jc) * H + head];
            const float sq = a.inf * (mask_i * st.km[rowmap(r, h)] - 1.0f);
            float s = S[r] * c1 + c2 * bias;
            s = s + pt[r] * (-0.5f);
            s = s + sq;
            s = (j < N) ? s : -INFINITY;
            S[r] = s;
            tmax = fmaxf(tmax, s);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(S[r] - m_new);
            S[r] = p;
            psum += p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {  // the running max settles after a few tiles: skip the rescale then
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
            for (int x = 0; x < PZ; ++x) opair[x] *= alpha;
        }

        // ---------------- O^T += V^T . P^T (+ value points), o_pair FMAs in the MFMA shadow (ipa.py:221-257)
        float4 z[PZ / 4];
        {
            const float* pz = a.pair_z + (row_i * N + min(j0 + rowmap(0, h), N - 1)) * PZ;
#pragma unroll
            for (int x = 0; x < PZ / 4; ++x) z[x] = *reinterpret_cast<const float4*>(pz + 4 * x);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = rowmap(r, h);
            const float p = S[r];
            float4 zc[PZ / 4];
#pragma unroll
            for (int x = 0; x < PZ / 4; ++x) zc[x] = z[x];
            if (r + 1 < 16) {
                const float* pz = a.pair_z + (row_i * N + min(j0 + rowmap(r + 1, h), N - 1)) * PZ;
#pragma unroll
                for (int x = 0; x < PZ / 4; ++x) z[x] = *reinterpret_cast<const float4*>(pz + 4 * x);
            }
            const float* vrow = st.v + kr * C + c;
#pragma unroll
            for (int t = 0; t < CT; ++t) O[t] = mfma32(vrow[32 * t], p, O[t]);
            O[CT] = mfma32(st.vp[kr * 64 + c], p, O[CT]);
            O[CT + 1] = mfma32(st.vp[kr * 64 + 32 + c], p, O[CT + 1]);
#pragma unroll
            for (int x = 0; x < PZ / 4; ++x) {
                opair[4 * x + 0] += p * zc[x].x; opair[4 * x + 1] += p * zc[x].y;
                opair[4 * x + 2] += p * zc[x].z; opair[4 * x + 3] += p * zc[x].w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile's DMA has landed
        __syncthreads();                                   // ... and everyone is done reading this one
    }

    // ---------------- epilogue: normalise, inverse-transform points, write concat layout
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int feat = H * (C + 4 * PV + PZ);
    float* orow = a.out + row_i * feat;
    if (ivalid) {
        float* oo = orow + head * C;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4*>(oo + 32 * t + 8 * rq + 4 * h) =
                    make_float4(O[t][4 * rq] * inv, O[t][4 * rq + 1] * inv, O[t][4 * rq + 2] * inv, O[t][4 * rq + 3] * inv);
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
    {
        float* op = orow + H * (C + 4 * PV) + head * PZ;
#pragma unroll
        for (int x = 0; x < PZ / 4; ++x) {
            float4 v;
            v.x = (opair[4 * x + 0] + __shfl_xor(opair[4 * x + 0], 32, 64)) * inv;
            v.y = (opair[4 * x + 1] + __shfl_xor(opair[4 * x + 1], 32, 64)) * inv;
            v.z = (opair[4 * x + 2] + __shfl_xor(opair[4 * x + 2], 32, 64)) * inv;
            v.w = (opair[4 * x + 3] + __shfl_xor(opair[4 * x + 3], 32, 64)) * inv;
            if (ivalid && h == 0) *reinterpret_cast<float4*>(op + 4 * x) = v;
        }
    }
}

}  // namespace

extern "C" int s2s_ipa_attention(const float* q, const float* kv, const float* q_pts, const float* k_pts, const float* v_pts64,
                                 const float* attn_bias, const float* pair_z, const float* mask, const float* rigids7,
                                 const float* head_w_scaled, float* out, int n_samples, int n_res, int n_heads, int c_hidden,
                                 int n_qk_points, int n_v_points, int c_pair_z, float inf, float eps, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (c_hidden != 256 || n_qk_points != 8 || n_v_points != 12 || c_pair_z != 32 || n_heads < 1)
        return (int)hipErrorInvalidValue;  // the reference configuration (configs/model/diffusion.yaml:29-40)
    IpaArgs a{q, kv, q_pts, k_pts, v_pts64, attn_bias, pair_z, mask, rigids7, head_w_scaled, out, n_samples, n_res, n_heads, inf, eps};
    const int n_qb = (n_res + 127) / 128;
    const long long blocks = (long long)n_samples * n_heads * n_qb;
    hipLaunchKernelGGL((ipa_attention_kernel<256, 8, 12, 32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
