// Invariant Point Attention core (gfx950): logits -> masked softmax -> value aggregation, fused.
// Reference: InvariantPointAttention.forward, src/models/net/ipa.py:183-257 (the part between the
// input projections and linear_out).  The [B,N,N,H,Pq,3] displacement tensor (6.4 GB at B=128,
// N=256), the [B,H,N,N] attention matrix and the [B,H,3,N,N,Pv] product (9.7 GB) that eager
// PyTorch materialises are never formed: one wave owns (sample, head, 32 query residues) and
// streams 32-residue key tiles with an online softmax (flash-attention schedule).
//
// MFMA orientation (v_mfma_f32_32x32x2_f32, exact fp32):
//   S^T[j, i] = K[j, :] . Q[i, :]      A = key tile (row j per lane), B = Q held in 128 VGPRs
//   O^T[c, i] += V^T[c, j] . P^T[j, i] A = value columns (coalesced 128 B per half wave),
//                                       B = the S^T accumulator itself (C layout == B layout,
//                                       k-order of the dot product is free)
//   so a lane owns ONE query residue i: softmax statistics, the running rescale, the point term
//   and the o_pair accumulation are per-lane scalars; only max/sum need one cross-half exchange.
// Point term  -1/2 * softplus(w_h)*c * sum_p |q_ip - k_jp|^2  is evaluated with explicit differences on
// the VALU (cheap: 9*Pq flops per (i,j,h)), exactly as the reference forms it — not through the
// |q|^2+|k|^2-2q.k expansion, which loses ~2 digits to cancellation.
// o_pair[i,h,:] = sum_j a_ij pair_z[i,j,:] is not a GEMM (the "value" depends on i): VALU too.
// Output is written directly in linear_out's concat order (ipa.py:259-266):
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
__device__ __forceinline__ float f4get(const float4& v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

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

template <int C, int PQ, int PV, int PZ>
__global__ void __launch_bounds__(256) ipa_attention_kernel(IpaArgs a) {
    static_assert(C % 32 == 0 && PV <= 16 && PZ % 4 == 0, "shape");
    constexpr int CT = C / 32;      // value tiles
    constexpr int OT = CT + 2;      // + two tiles of packed value points
    constexpr int QG = C / 8;       // float4 groups of Q per lane
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int wave = threadIdx.x >> 6;
    const int n_it = (a.N + 31) / 32;
    const int hgroups = a.H / 4;
    int bid = blockIdx.x;
    const int hg = bid % hgroups; bid /= hgroups;
    const int it = bid % n_it;
    const int b = bid / n_it;
    const int head = hg * 4 + wave;
    const int N = a.N, H = a.H;
    const int i = it * 32 + c;
    const bool ivalid = i < N;
    const int ic = ivalid ? i : N - 1;
    const long long row_i = (long long)b * N + ic;

    // ---- this lane's query row (B operand of QK^T), points and mask
    float4 qreg[QG];
    {
        const float* qp = a.q + (row_i * H + head) * C;
#pragma unroll
        for (int g = 0; g < QG; ++g) qreg[g] = *reinterpret_cast<const float4*>(qp + 8 * g + 4 * h);
    }
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

    const long long kvrow_stride = (long long)H * 2 * C;
    const float* kv_bh = a.kv + (long long)b * N * kvrow_stride + (long long)head * 2 * C;

    // wave-private LDS: the key tile's points (32 x PQ*3) and key mask (32); every lane of a half
    // wave reads the same key, so these are broadcast ds_reads instead of 64 redundant global loads
    __shared__ float s_kpts[4][32 * PQ * 3 + 32];
    float* kl = s_kpts[wave];
    constexpr int KPL = (32 * PQ * 3) / 64;  // floats of the key-point tile per lane
    static_assert(KPL % 4 == 0 && (PQ * 3) % KPL == 0, "key point tile split");

    for (int j0 = 0; j0 < N; j0 += 32) {
        // ---------------- stage key points / key mask of this tile
        {
            const int jj = (lane * KPL) / (PQ * 3), col = (lane * KPL) % (PQ * 3);
            const int jr = min(j0 + jj, N - 1);
            const float* src = a.k_pts + (((long long)b * N + jr) * H + head) * (PQ * 3) + col;
#pragma unroll
            for (int x = 0; x < KPL / 4; ++x)
                *reinterpret_cast<float4*>(kl + lane * KPL + 4 * x) = *reinterpret_cast<const float4*>(src + 4 * x);
            if (lane < 32) kl[32 * PQ * 3 + lane] = a.mask[(long long)b * N + min(j0 + lane, N - 1)];
        }
        // ---------------- S^T = K . Q^T   (K fragments through a 4-deep prefetch ring)
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {
            const int ja = min(j0 + c, N - 1);
            const float* kp = kv_bh + ja * kvrow_stride + 4 * h;
            constexpr int D = 4;
            float4 kf[D];
#pragma unroll
            for (int d = 0; d < D; ++d) kf[d] = *reinterpret_cast<const float4*>(kp + 8 * d);
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const float4 cur = kf[g % D];
                if (g + D < QG) kf[g % D] = *reinterpret_cast<const float4*>(kp + 8 * (g + D));
                S = mfma32(cur.x, qreg[g].x, S);
                S = mfma32(cur.y, qreg[g].y, S);
                S = mfma32(cur.z, qreg[g].z, S);
                S = mfma32(cur.w, qreg[g].w, S);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---------------- logits (ipa.py:183-214)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + rowmap(r, h);
            const int jc = min(j, N - 1);
            const float* kp = kl + rowmap(r, h) * (PQ * 3);
            float pt = 0.f;
#pragma unroll
            for (int p = 0; p < PQ; ++p) {
                const float dx = qpt[p * 3 + 0] - kp[p * 3 + 0];
                const float dy = qpt[p * 3 + 1] - kp[p * 3 + 1];
                const float dz = qpt[p * 3 + 2] - kp[p * 3 + 2];
                pt += (dx * dx + dy * dy + dz * dz) * hw;
            }
            const float bias = a.attn_bias[(row_i * N + jc) * H + head];
            const float sq = a.inf * (mask_i * kl[32 * PQ * 3 + rowmap(r, h)] - 1.0f);
            float s = S[r] * c1 + c2 * bias;
            s = s + pt * (-0.5f);
            s = s + sq;
            s = (j < N) ? s : -INFINITY;
            S[r] = s;
            tmax = fmaxf(tmax, s);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
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
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
        for (int x = 0; x < PZ; ++x) opair[x] *= alpha;

        // ---------------- O^T += V^T . P^T   (+ value points as two extra tiles)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jc = min(j0 + rowmap(r, h), N - 1);
            const float* vp = kv_bh + jc * kvrow_stride + C + c;
            const float p = S[r];
#pragma unroll
            for (int t = 0; t < CT; ++t) O[t] = mfma32(vp[32 * t], p, O[t]);
            const float* pp = a.v_pts + (((long long)b * N + jc) * H + head) * 64 + c;
            O[CT] = mfma32(pp[0], p, O[CT]);
            O[CT + 1] = mfma32(pp[32], p, O[CT + 1]);
            if (r & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------- o_pair partial sums over this lane's 16 keys (ipa.py:253-257)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jc = min(j0 + rowmap(r, h), N - 1);
            const float* pz = a.pair_z + (row_i * N + jc) * PZ;
            const float p = S[r];
#pragma unroll
            for (int x = 0; x < PZ / 4; ++x) {
                const float4 z = *reinterpret_cast<const float4*>(pz + 4 * x);
                opair[4 * x + 0] += p * z.x; opair[4 * x + 1] += p * z.y;
                opair[4 * x + 2] += p * z.z; opair[4 * x + 3] += p * z.w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();  // all lanes done with this tile's LDS before it is restaged
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
        const float qa = f[0], qb = f[1], qc = f[2], qd = f[3];
        const float tx = f[4], ty = f[5], tz = f[6];
        const float r00 = qa * qa + qb * qb - qc * qc - qd * qd, r01 = 2 * qb * qc - 2 * qa * qd, r02 = 2 * qb * qd + 2 * qa * qc;
        const float r10 = 2 * qb * qc + 2 * qa * qd, r11 = qa * qa - qb * qb + qc * qc - qd * qd, r12 = 2 * qc * qd - 2 * qa * qb;
        const float r20 = 2 * qb * qd - 2 * qa * qc, r21 = 2 * qc * qd + 2 * qa * qb, r22 = qa * qa - qb * qb - qc * qc + qd * qd;
        float* ox = orow + H * C + head * PV;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt = 8 * t + 2 * rq + h;  // point whose (x,y,z,0) group this lane holds
                const float dx = O[CT + t][4 * rq + 0] * inv - tx;
                const float dy = O[CT + t][4 * rq + 1] * inv - ty;
                const float dz = O[CT + t][4 * rq + 2] * inv - tz;
                const float lx = r00 * dx + r10 * dy + r20 * dz;
                const float ly = r01 * dx + r11 * dy + r21 * dz;
                const float lz = r02 * dx + r12 * dy + r22 * dz;
                const float nr = sqrtf(lx * lx + ly * ly + lz * lz + a.eps);
                if (ivalid && pt < PV) {
                    ox[pt] = lx;
                    ox[H * PV + pt] = ly;
                    ox[2 * H * PV + pt] = lz;
                    ox[3 * H * PV + pt] = nr;
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
    if (c_hidden != 256 || n_qk_points != 8 || n_v_points != 12 || c_pair_z != 32 || n_heads % 4 != 0)
        return (int)hipErrorInvalidValue;  // the reference configuration (configs/model/diffusion.yaml:29-40)
    IpaArgs a{q, kv, q_pts, k_pts, v_pts64, attn_bias, pair_z, mask, rigids7, head_w_scaled, out, n_samples, n_res, n_heads, inf, eps};
    const int n_it = (n_res + 31) / 32;
    const long long blocks = (long long)n_samples * n_it * (n_heads / 4);
    hipLaunchKernelGGL((ipa_attention_kernel<256, 8, 12, 32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
