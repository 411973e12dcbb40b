// Forward process / prior on the device (gfx950): one thread per residue, once per trajectory.
//
// Mirrors, op for op and dtype for dtype, what the reference does on the host with a Python loop per replica:
//   FrameDiffuser.forward_marginal   src/models/score/frame.py:36-107
//   SO3Diffuser.forward_marginal / sample   so3.py:315-331, :244-272  (inverse CDF of the IGSO(3) angle: np.interp of a
//                                    uniform draw on the cdf row of t's sigma bin; uniform random axis), compose_rotvec :13-19
//   R3Diffuser.forward_marginal      r3.py:49-74        x_t = exp(-b/2) * 0.1 x0 + sqrt(1 - exp(-b)) * z, then / 0.1
//   FrameDiffuser.sample_prior       frame.py:212-255   (rotation: IGSO(3) at t = 1; translation: N(0,1) / 0.1)
//   assemble_rigid + Rigid.to_tensor_7   frame.py:9-15, rigid_utils.py:1203-1215 (get_quats = matrix_to_quaternion)
// The NOISE is an input (z_axis, u, z_trans drawn by the caller on the device generator: Philox), so the kernel is a pure
// function: given the host path's draws it reproduces the host path's frames (tests/test_hip_parity.py), and the
// throughput mode of the sampler has no host loop, no np.interp and no host->device copy of frames per chunk.
#include <hip/hip_runtime.h>

#include "geom.h"
#include "str2str_hip.h"

using namespace s2s;

namespace {

// np.interp(u, xp, fp) for increasing xp: fp[0] below xp[0], fp[n-1] at or above xp[n-1], else
// slope * (u - xp[j]) + fp[j] with xp[j] <= u < xp[j+1]  (numpy/core/src/multiarray/compiled_base.c), all in float64.
__device__ __forceinline__ double interp_cdf(double u, const double* __restrict__ xp, const float* __restrict__ fp, int n) {
    if (!(u >= xp[0])) return (double)fp[0];      // also NaN-safe
    if (u >= xp[n - 1]) return (double)fp[n - 1];
    int lo = 0, hi = n - 1;                       // invariant: xp[lo] <= u < xp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= u) lo = mid; else hi = mid;
    }
    const double slope = ((double)fp[lo + 1] - (double)fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (u - xp[lo]) + (double)fp[lo];
}

__global__ void __launch_bounds__(256) forward_marginal_kernel(
    const float* __restrict__ rig0_4x4, const float* __restrict__ z_axis, const float* __restrict__ u01,
    const float* __restrict__ z_trans, const double* __restrict__ cdf_rows, const int* __restrict__ row_of_sample,
    const float* __restrict__ omega_grid, int n_omega, const float* __restrict__ params2,
    const float* __restrict__ diffuse_mask, float coord_scale, float* __restrict__ out7, long long M, int N) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const int b = (int)(r / N);
    // ---- IGSO(3) sample (so3.py:262-272): x = z / |z|, angle by inverse CDF; rotvec = x * angle
    const float zx = z_axis[r * 3 + 0], zy = z_axis[r * 3 + 1], zz = z_axis[r * 3 + 2];
    const float zn = sqrtf(zx * zx + zy * zy + zz * zz);
    const double ang = interp_cdf((double)u01[r], cdf_rows + (long long)row_of_sample[b] * n_omega, omega_grid, n_omega);
    const float angf = (float)ang;
    const Vec3<float> rv0t{(zx / zn) * angf, (zy / zn) * angf, (zz / zn) * angf};

    Vec3<float> rot_t;
    float tr[3];
    if (rig0_4x4) {
        const float* g = rig0_4x4 + r * 16;
        Mat3<float> R0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) R0.m[i][j] = g[i * 4 + j];
        const float t0[3] = {g[3], g[7], g[11]};
        const Vec3<float> rot_0 = quaternion_to_axis_angle<float>(matrix_to_quaternion<float>(R0));
        // compose_rotvec (so3.py:13-19): float32 matrices, float64 product, float64 matrix -> axis-angle, back to float32
        const Mat3<double> R1 = mat_cast<double, float>(quaternion_to_matrix<float>(axis_angle_to_quaternion<float>(rot_0)));
        const Mat3<double> R2 = mat_cast<double, float>(quaternion_to_matrix<float>(axis_angle_to_quaternion<float>(rv0t)));
        const Vec3<double> c = quaternion_to_axis_angle<double>(matrix_to_quaternion<double>(rot_matmul<double>(R1, R2)));
        const float e_half = params2[b * 2 + 0], std = params2[b * 2 + 1];
        const float m = diffuse_mask ? diffuse_mask[r] : 1.0f;
        rot_t.x = m * (float)c.x + (1.0f - m) * rot_0.x;
        rot_t.y = m * (float)c.y + (1.0f - m) * rot_0.y;
        rot_t.z = m * (float)c.z + (1.0f - m) * rot_0.z;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x0 = t0[k] * coord_scale;
            const float xt = z_trans[r * 3 + k] * std + e_half * x0;
            tr[k] = m * (xt / coord_scale) + (1.0f - m) * t0[k];
        }
    } else {  // prior (frame.py:212-255 without reference rigids)
        rot_t = rv0t;
#pragma unroll
        for (int k = 0; k < 3; ++k) tr[k] = z_trans[r * 3 + k] / coord_scale;
    }
    const Quat<float> q = matrix_to_quaternion<float>(quaternion_to_matrix<float>(axis_angle_to_quaternion<float>(rot_t)));
    float* o = out7 + r * 7;
    o[0] = q.w; o[1] = q.x; o[2] = q.y; o[3] = q.z;
    o[4] = tr[0]; o[5] = tr[1]; o[6] = tr[2];
}

}  // namespace

extern "C" int s2s_forward_marginal(const float* rigids0_4x4, const float* z_axis, const float* u01, const float* z_trans,
                                    const double* cdf_rows, const int* cdf_row_of_sample, const float* omega_grid, int n_omega,
                                    const float* params2, const float* diffuse_mask, float coordinate_scaling,
                                    float* rigids_t7, int n_samples, int n_res, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (!z_axis || !u01 || !z_trans || !cdf_rows || !cdf_row_of_sample || !omega_grid || n_omega < 2 || !rigids_t7)
        return (int)hipErrorInvalidValue;
    if (rigids0_4x4 && !params2) return (int)hipErrorInvalidValue;
    const long long M = (long long)n_samples * n_res;
    hipLaunchKernelGGL(forward_marginal_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rigids0_4x4,
                       z_axis, u01, z_trans, cdf_rows, cdf_row_of_sample, omega_grid, n_omega, params2, diffuse_mask,
                       coordinate_scaling, rigids_t7, M, n_res);
    return (int)hipGetLastError();
}
