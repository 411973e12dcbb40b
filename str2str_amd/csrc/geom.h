// Device-side rigid-frame / rotation algebra shared by the per-residue kernels (gfx950).
//
// Every routine mirrors one reference routine, including its branch thresholds and the order of
// its floating-point operations, so that the HIP path stays inside rounding distance of the
// reference PyTorch path (this translation unit is built with -ffp-contract=off for that reason):
//   quat_to_rot             src/common/rigid_utils.py:163-207   (quadratic form, no renormalisation)
//   quaternion_to_matrix    src/common/rotation3d.py:41-70      (two_s = 2/|q|^2)
//   matrix_to_quaternion    src/common/rotation3d.py:102-161    (4 candidates, first arg-max, floor 0.1)
//   axis_angle_to_quaternion  rotation3d.py:493-522             (|angle| < 1e-6 -> 1/2 - a^2/48)
//   quaternion_to_axis_angle  rotation3d.py:525-553             (half = atan2(|xyz|, w); no sign fix)
//   quat_multiply / quat_multiply_by_vec  rigid_utils.py:256-277
// Templated on the scalar so the float64 island of so3.compose_rotvec (so3.py:13-19) uses the
// same code in double.
#pragma once
#include <hip/hip_runtime.h>

namespace s2s {

template <typename T> struct Quat { T w, x, y, z; };
template <typename T> struct Vec3 { T x, y, z; };
template <typename T> struct Mat3 { T m[3][3]; };

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_sin(T x);
template <> __device__ __forceinline__ float t_sin<float>(float x) { return sinf(x); }
template <> __device__ __forceinline__ double t_sin<double>(double x) { return sin(x); }
template <typename T> __device__ __forceinline__ T t_cos(T x);
template <> __device__ __forceinline__ float t_cos<float>(float x) { return cosf(x); }
template <> __device__ __forceinline__ double t_cos<double>(double x) { return cos(x); }
template <typename T> __device__ __forceinline__ T t_atan2(T y, T x);
template <> __device__ __forceinline__ float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> __device__ __forceinline__ double t_atan2<double>(double y, double x) { return atan2(y, x); }
template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }

template <typename T>
__device__ __forceinline__ Mat3<T> quat_to_rot(const Quat<T>& q) {
    const T a = q.w, b = q.x, c = q.y, d = q.z;
    const T aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    Mat3<T> r;
    r.m[0][0] = aa + bb - cc - dd;
    r.m[0][1] = T(2) * b * c - T(2) * a * d;
    r.m[0][2] = T(2) * b * d + T(2) * a * c;
    r.m[1][0] = T(2) * b * c + T(2) * a * d;
    r.m[1][1] = aa - bb + cc - dd;
    r.m[1][2] = T(2) * c * d - T(2) * a * b;
    r.m[2][0] = T(2) * b * d - T(2) * a * c;
    r.m[2][1] = T(2) * c * d + T(2) * a * b;
    r.m[2][2] = aa - bb - cc + dd;
    return r;
}

template <typename T>
__device__ __forceinline__ Mat3<T> quaternion_to_matrix(const Quat<T>& q) {
    const T r = q.w, i = q.x, j = q.y, k = q.z;
    const T two_s = T(2) / (r * r + i * i + j * j + k * k);
    Mat3<T> o;
    o.m[0][0] = T(1) - two_s * (j * j + k * k);
    o.m[0][1] = two_s * (i * j - k * r);
    o.m[0][2] = two_s * (i * k + j * r);
    o.m[1][0] = two_s * (i * j + k * r);
    o.m[1][1] = T(1) - two_s * (i * i + k * k);
    o.m[1][2] = two_s * (j * k - i * r);
    o.m[2][0] = two_s * (i * k - j * r);
    o.m[2][1] = two_s * (j * k + i * r);
    o.m[2][2] = T(1) - two_s * (i * i + j * j);
    return o;
}

template <typename T> __device__ __forceinline__ T sqrt_positive_part(T x) { return x > T(0) ? t_sqrt<T>(x) : T(0); }

template <typename T>
__device__ __forceinline__ Quat<T> matrix_to_quaternion(const Mat3<T>& M) {
    const T m00 = M.m[0][0], m01 = M.m[0][1], m02 = M.m[0][2];
    const T m10 = M.m[1][0], m11 = M.m[1][1], m12 = M.m[1][2];
    const T m20 = M.m[2][0], m21 = M.m[2][1], m22 = M.m[2][2];
    T qa[4];
    qa[0] = sqrt_positive_part<T>(T(1) + m00 + m11 + m22);
    qa[1] = sqrt_positive_part<T>(T(1) + m00 - m11 - m22);
    qa[2] = sqrt_positive_part<T>(T(1) - m00 + m11 - m22);
    qa[3] = sqrt_positive_part<T>(T(1) - m00 - m11 + m22);
    int best = 0;
    T bv = qa[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (qa[k] > bv) { bv = qa[k]; best = k; }
    T c0, c1, c2, c3;
    if (best == 0) { c0 = qa[0] * qa[0]; c1 = m21 - m12; c2 = m02 - m20; c3 = m10 - m01; }
    else if (best == 1) { c0 = m21 - m12; c1 = qa[1] * qa[1]; c2 = m10 + m01; c3 = m02 + m20; }
    else if (best == 2) { c0 = m02 - m20; c1 = m10 + m01; c2 = qa[2] * qa[2]; c3 = m12 + m21; }
    else { c0 = m10 - m01; c1 = m20 + m02; c2 = m21 + m12; c3 = qa[3] * qa[3]; }
    const T den = T(2) * (bv > T(0.1) ? bv : T(0.1));
    Quat<T> q;
    q.w = c0 / den; q.x = c1 / den; q.y = c2 / den; q.z = c3 / den;
    return q;
}

template <typename T>
__device__ __forceinline__ Quat<T> axis_angle_to_quaternion(const Vec3<T>& v) {
    const T ang = t_sqrt<T>(v.x * v.x + v.y * v.y + v.z * v.z);
    const T half = ang * T(0.5);
    T s;
    if (t_abs<T>(ang) < T(1e-6)) s = T(0.5) - (ang * ang) / T(48);
    else s = t_sin<T>(half) / ang;
    Quat<T> q;
    q.w = t_cos<T>(half); q.x = v.x * s; q.y = v.y * s; q.z = v.z * s;
    return q;
}

template <typename T>
__device__ __forceinline__ Vec3<T> quaternion_to_axis_angle(const Quat<T>& q) {
    const T nrm = t_sqrt<T>(q.x * q.x + q.y * q.y + q.z * q.z);
    const T half = t_atan2<T>(nrm, q.w);
    const T ang = T(2) * half;
    T s;
    if (t_abs<T>(ang) < T(1e-6)) s = T(0.5) - (ang * ang) / T(48);
    else s = t_sin<T>(half) / ang;
    Vec3<T> v;
    v.x = q.x / s; v.y = q.y / s; v.z = q.z / s;
    return v;
}

template <typename T>
__device__ __forceinline__ Quat<T> quat_multiply(const Quat<T>& p, const Quat<T>& q) {
    Quat<T> r;
    r.w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
    r.x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
    r.y = p.w * q.y - p.x * q.z + p.y * q.w + p.z * q.x;
    r.z = p.w * q.z + p.x * q.y - p.y * q.x + p.z * q.w;
    return r;
}

template <typename T>
__device__ __forceinline__ Quat<T> quat_multiply_by_vec(const Quat<T>& q, const Vec3<T>& v) {
    Quat<T> r;
    r.w = -q.x * v.x - q.y * v.y - q.z * v.z;
    r.x = q.w * v.x + q.y * v.z - q.z * v.y;
    r.y = q.w * v.y - q.x * v.z + q.z * v.x;
    r.z = q.w * v.z + q.x * v.y - q.y * v.x;
    return r;
}

template <typename T>
__device__ __forceinline__ Vec3<T> rot_vec_mul(const Mat3<T>& r, const Vec3<T>& p) {
    Vec3<T> o;
    o.x = r.m[0][0] * p.x + r.m[0][1] * p.y + r.m[0][2] * p.z;
    o.y = r.m[1][0] * p.x + r.m[1][1] * p.y + r.m[1][2] * p.z;
    o.z = r.m[2][0] * p.x + r.m[2][1] * p.y + r.m[2][2] * p.z;
    return o;
}

template <typename T>
__device__ __forceinline__ Mat3<T> rot_matmul(const Mat3<T>& a, const Mat3<T>& b) {
    Mat3<T> o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return o;
}

template <typename T, typename U>
__device__ __forceinline__ Mat3<T> mat_cast(const Mat3<U>& a) {
    Mat3<T> o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.m[i][j] = (T)a.m[i][j];
    return o;
}

}  // namespace s2s
