// Per-residue rigid-frame kernels (gfx950): one thread per residue, frames read as 7 floats
// (quaternion w,x,y,z + translation) — 28 B/residue, so these are launch/latency-bound, not
// bandwidth-bound.  Built with -ffp-contract=off: each op mirrors one eager
// PyTorch op of the reference.
//
//   s2s_rigid_compose_update   Rigid.compose_q_update_vec      rigid_utils.py:1042-1066, :590-619
//   s2s_rigid_scale_trans      TranslationIPA scale/unscale    ipa.py:288-292,339,379
//   s2s_ipa_prep_points        q/k/v point generation + r.apply   ipa.py:144-171, rigid_utils.py:1107-1120
//   s2s_frames_to_backbone     compute_backbone                all_atom.py:141-173 (+ :21-83, :99-138)
#include <hip/hip_runtime.h>

#include "geom.h"
#include "str2str_hip.h"

using namespace s2s;

namespace {

__device__ __forceinline__ void load7(const float* __restrict__ p, Quat<float>& q, Vec3<float>& t) {
    q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3];
    t.x = p[4]; t.y = p[5]; t.z = p[6];
}

__global__ void __launch_bounds__(256) compose_update_kernel(const float* __restrict__ rig, const float* __restrict__ upd,
                                                             const float* __restrict__ mask, float* __restrict__ out,
                                                             long long M, int upd_ld) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    Quat<float> q; Vec3<float> t;
    load7(rig + r * 7, q, t);
    const float* u = upd + r * upd_ld;
    const float m = mask[r];
    const Vec3<float> qv{u[0], u[1], u[2]};
    const Vec3<float> tv{u[3], u[4], u[5]};
    const Quat<float> du = quat_multiply_by_vec<float>(q, qv);
    Quat<float> nq{q.w + du.w * m, q.x + du.x * m, q.y + du.y * m, q.z + du.z * m};
    const float nrm = sqrtf(nq.w * nq.w + nq.x * nq.x + nq.y * nq.y + nq.z * nq.z);
    nq.w /= nrm; nq.x /= nrm; nq.y /= nrm; nq.z /= nrm;
    const Mat3<float> R = quat_to_rot<float>(q);  // rotation of the OLD quaternion
    const Vec3<float> d = rot_vec_mul<float>(R, tv);
    float* o = out + r * 7;
    o[0] = nq.w; o[1] = nq.x; o[2] = nq.y; o[3] = nq.z;
    o[4] = t.x + d.x * m; o[5] = t.y + d.y * m; o[6] = t.z + d.z * m;
}

__global__ void __launch_bounds__(256) scale_trans_kernel(const float* __restrict__ rig, float* __restrict__ out,
                                                          long long M, float scale, int divide) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* p = rig + r * 7;
    float* o = out + r * 7;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
    if (divide) { o[4] = p[4] / scale; o[5] = p[5] / scale; o[6] = p[6] / scale; }
    else { o[4] = p[4] * scale; o[5] = p[5] * scale; o[6] = p[6] * scale; }
}

// One thread per (residue, head).  Inputs are the raw linear outputs, coordinate-major:
//   qp_lin [M, 3*H*Pq]   : x block, y block, z block; inside a block index = h*Pq + p
//   kvp_lin[M, 3*H*(Pq+Pv)]: same, index = h*(Pq+Pv) + p ; p < Pq -> key point, else value point
// Outputs (global frame, R p + t with the residue's frame):
//   q_pts [M, H, Pq*3], k_pts [M, H, Pq*3]  (point-major xyz)
//   v_pts [M, H, VP]  with VP = 4*ceil16(Pv) laid out (x,y,z,0) per point, zero padded: the
//   4-float groups keep one point inside one lane of the attention kernel's MFMA C-layout.
// One thread per output point (q point, k point, or one of the VP/4 value-point slots incl. padding): the coordinate-major
// linear outputs are read with unit stride across threads and the (x,y,z,0) value points leave as one float4 per thread.
__global__ void __launch_bounds__(256) ipa_prep_points_kernel(const float* __restrict__ rig, const float* __restrict__ qp_lin,
                                                              const float* __restrict__ kvp_lin, float* __restrict__ q_pts,
                                                              float* __restrict__ k_pts, float* __restrict__ v_pts,
                                                              long long M, int H, int Pq, int Pv, int VP) {
    const int per_frame = H * (2 * Pq + VP / 4);
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= M * per_frame) return;
    const long long r = id / per_frame;
    int w = (int)(id - r * per_frame);
    Quat<float> q; Vec3<float> t;
    load7(rig + r * 7, q, t);
    const Mat3<float> R = quat_to_rot<float>(q);
    const int HPq = H * Pq, HPkv = H * (Pq + Pv);
    if (w < HPq) {  // query point c = h*Pq + p
        const float* ql = qp_lin + r * 3 * HPq;
        const Vec3<float> a{ql[w], ql[HPq + w], ql[2 * HPq + w]};
        const Vec3<float> g = rot_vec_mul<float>(R, a);
        float* o = q_pts + (r * HPq + w) * 3;
        o[0] = g.x + t.x; o[1] = g.y + t.y; o[2] = g.z + t.z;
        return;
    }
    w -= HPq;
    const float* kl = kvp_lin + r * 3 * HPkv;
    if (w < HPq) {  // key point (h, p): column h*(Pq+Pv) + p of the kv-point projection
        const int hh = w / Pq, p = w - hh * Pq, c = hh * (Pq + Pv) + p;
        const Vec3<float> a{kl[c], kl[HPkv + c], kl[2 * HPkv + c]};
        const Vec3<float> g = rot_vec_mul<float>(R, a);
        float* o = k_pts + (r * HPq + w) * 3;
        o[0] = g.x + t.x; o[1] = g.y + t.y; o[2] = g.z + t.z;
        return;
    }
    w -= HPq;  // value-point slot (h, p) with p < VP/4; slots >= Pv are zero padding
    const int slots = VP / 4, hh = w / slots, p = w - hh * slots;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < Pv) {
        const int c = hh * (Pq + Pv) + Pq + p;
        const Vec3<float> a{kl[c], kl[HPkv + c], kl[2 * HPkv + c]};
        const Vec3<float> g = rot_vec_mul<float>(R, a);
        o = make_float4(g.x + t.x, g.y + t.y, g.z + t.z, 0.f);
    }
    *reinterpret_cast<float4*>(v_pts + ((r * H + hh) * slots + p) * 4) = o;
}

// Backbone projection.  tables: pos[21][5][3], amask[21][5], group3[21][5] (1 if atom sits in the psi
// frame), dflt[21][2][12] (rows of [R|t] of the default frame of group 0 and group 3).
struct BackboneTables {
    float pos[21 * 5 * 3];
    float amask[21 * 5];
    int group3[21 * 5];
    float dflt[21 * 2 * 12];
};
__constant__ BackboneTables c_bb;

__global__ void __launch_bounds__(256) frames_to_backbone_kernel(const float* __restrict__ rig, const float* __restrict__ psi,
                                                                 const long long* __restrict__ aatype,
                                                                 float* __restrict__ atom14_5, float* __restrict__ atom37,
                                                                 long long M) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    Quat<float> q; Vec3<float> t;
    load7(rig + r * 7, q, t);
    const Mat3<float> R = quat_to_rot<float>(q);
    int aa = aatype ? (int)aatype[r] : 0;
    aa = aa < 0 ? 0 : (aa > 20 ? 20 : aa);
    const float sn = psi[r * 2 + 0], cs = psi[r * 2 + 1];
    float out[5][3];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        // torsion rotation of this group: group 0 -> (sin,cos) = (0,1); group 3 -> psi
        Mat3<float> tor;
        const float s = g ? sn : 0.f, c = g ? cs : 1.f;
        tor.m[0][0] = 1.f; tor.m[0][1] = 0.f; tor.m[0][2] = 0.f;
        tor.m[1][0] = 0.f; tor.m[1][1] = c;   tor.m[1][2] = -s;
        tor.m[2][0] = 0.f; tor.m[2][1] = s;   tor.m[2][2] = c;
        const float* d = c_bb.dflt + (aa * 2 + g) * 12;
        Mat3<float> Rd;
        Vec3<float> td;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Rd.m[i][0] = d[i * 4 + 0]; Rd.m[i][1] = d[i * 4 + 1]; Rd.m[i][2] = d[i * 4 + 2];
        }
        td.x = d[3]; td.y = d[7]; td.z = d[11];
        // default_r.compose(all_rots): rot = Rd*tor ; trans = Rd*0 + td   (all_atom.py:59-61)
        const Mat3<float> gr = rot_matmul<float>(Rd, tor);
        const Vec3<float> z0 = rot_vec_mul<float>(Rd, Vec3<float>{0.f, 0.f, 0.f});
        const Vec3<float> gt{z0.x + td.x, z0.y + td.y, z0.z + td.z};
        // r[..., None].compose(all_frames_to_bb)                          (all_atom.py:81)
        const Mat3<float> Gr = rot_matmul<float>(R, gr);
        const Vec3<float> rg = rot_vec_mul<float>(R, gt);
        const Vec3<float> Gt{rg.x + t.x, rg.y + t.y, rg.z + t.z};
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            if (c_bb.group3[aa * 5 + a] == g) {
                const float* pp = c_bb.pos + (aa * 5 + a) * 3;
                const Vec3<float> p = rot_vec_mul<float>(Gr, Vec3<float>{pp[0], pp[1], pp[2]});
                const float mk = c_bb.amask[aa * 5 + a];
                out[a][0] = (p.x + Gt.x) * mk; out[a][1] = (p.y + Gt.y) * mk; out[a][2] = (p.z + Gt.z) * mk;
            }
        }
    }
    if (atom14_5) {
        float* o = atom14_5 + r * 15;  // atom14 order: N, CA, C, O, CB
#pragma unroll
        for (int a = 0; a < 5; ++a) { o[a * 3] = out[a][0]; o[a * 3 + 1] = out[a][1]; o[a * 3 + 2] = out[a][2]; }
    }
    if (atom37) {
        float* o = atom37 + r * 111;  // atom37 order: N, CA, C, CB, O, then 32 empty slots
        const int src[5] = {0, 1, 2, 4, 3};
#pragma unroll
        for (int a = 0; a < 5; ++a) { o[a * 3] = out[src[a]][0]; o[a * 3 + 1] = out[src[a]][1]; o[a * 3 + 2] = out[src[a]][2]; }
        for (int x = 15; x < 111; ++x) o[x] = 0.f;
    }
}

inline int grid_for(long long n, int block) { return (int)((n + block - 1) / block); }

}  // namespace

// TorsionAngleHead's normalisation (layers.py:199-213: u / sqrt(max(sum u^2, eps))) and DenoisingNet's blend with the input torsion
// under the fixed mask (denoising_ipa.py:193-195: gt * fixed + pred * (1 - fixed)) -- each a chain of five tiny elementwise launches in
// eager PyTorch, and a network evaluation of a small chunk is launch-latency bound.  Same operations in the same order (this file is
// built with -ffp-contract=off).
__global__ void __launch_bounds__(256) torsion_head_kernel(const float* __restrict__ u, int u_ld, int normalize, const float* __restrict__ gt,
                                                           long long gt_stride, const float* __restrict__ fixed, float eps,
                                                           float* __restrict__ out, long long M) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    float a = u[r * u_ld], b = u[r * u_ld + 1];
    if (normalize) {
        const float s = sqrtf(fmaxf(a * a + b * b, eps));
        a = a / s;
        b = b / s;
    }
    if (gt) {
        const float f = fixed[r], c = 1.0f - f;
        a = gt[r * gt_stride] * f + a * c;
        b = gt[r * gt_stride + 1] * f + b * c;
    }
    out[2 * r] = a;
    out[2 * r + 1] = b;
}

extern "C" {

int s2s_torsion_head(const float* u, int u_ld, int normalize, const float* gt_sin_cos, long long gt_row_stride, const float* fixed_mask,
                     float eps, float* out2, long long n_rows, void* stream) {
    if (n_rows <= 0) return 0;
    if (!u || u_ld < 2 || !out2 || ((gt_sin_cos != nullptr) != (fixed_mask != nullptr))) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(torsion_head_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, u, u_ld, normalize, gt_sin_cos,
                       gt_row_stride, fixed_mask, eps, out2, n_rows);
    return (int)hipGetLastError();
}

int s2s_rigid_compose_update(const float* rigids7, const float* update6, const float* mask, float* out7,
                             long long n_frames, int update_ld, void* stream) {
    if (n_frames <= 0) return 0;
    if (update_ld < 6) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(compose_update_kernel, dim3(grid_for(n_frames, 256)), dim3(256), 0, (hipStream_t)stream, rigids7,
                       update6, mask, out7, n_frames, update_ld);
    return (int)hipGetLastError();
}

int s2s_rigid_scale_trans(const float* rigids7, float* out7, long long n_frames, float scale, int divide, void* stream) {
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(scale_trans_kernel, dim3(grid_for(n_frames, 256)), dim3(256), 0, (hipStream_t)stream, rigids7, out7,
                       n_frames, scale, divide);
    return (int)hipGetLastError();
}

int s2s_ipa_prep_points(const float* rigids7, const float* q_pts_lin, const float* kv_pts_lin, float* q_pts, float* k_pts,
                        float* v_pts, long long n_frames, int n_heads, int n_qk_points, int n_v_points, int v_pts_stride,
                        void* stream) {
    if (n_frames <= 0) return 0;
    if (v_pts_stride < 4 * n_v_points) return (int)hipErrorInvalidValue;
    if (v_pts_stride % 4) return (int)hipErrorInvalidValue;
    const long long threads = n_frames * n_heads * (2 * n_qk_points + v_pts_stride / 4);
    hipLaunchKernelGGL(ipa_prep_points_kernel, dim3(grid_for(threads, 256)), dim3(256), 0, (hipStream_t)stream,
                       rigids7, q_pts_lin, kv_pts_lin, q_pts, k_pts, v_pts, n_frames, n_heads, n_qk_points, n_v_points,
                       v_pts_stride);
    return (int)hipGetLastError();
}

int s2s_set_backbone_tables(const float* pos_21x5x3, const float* mask_21x5, const int* is_psi_group_21x5,
                            const float* default_frames_21x2x4x4) {
    BackboneTables h;
    for (int i = 0; i < 21 * 5 * 3; ++i) h.pos[i] = pos_21x5x3[i];
    for (int i = 0; i < 21 * 5; ++i) { h.amask[i] = mask_21x5[i]; h.group3[i] = is_psi_group_21x5[i] ? 1 : 0; }
    for (int a = 0; a < 21 * 2; ++a)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) h.dflt[a * 12 + i * 4 + j] = default_frames_21x2x4x4[a * 16 + i * 4 + j];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c_bb), &h, sizeof(h));
}

int s2s_frames_to_backbone(const float* rigids7, const float* psi_sincos, const long long* aatype, float* atom14_bb5,
                           float* atom37, long long n_frames, void* stream) {
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(frames_to_backbone_kernel, dim3(grid_for(n_frames, 256)), dim3(256), 0, (hipStream_t)stream, rigids7,
                       psi_sincos, aatype, atom14_bb5, atom37, n_frames);
    return (int)hipGetLastError();
}

}  // extern "C"
