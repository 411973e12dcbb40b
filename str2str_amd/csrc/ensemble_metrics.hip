// Ensemble metrics on the device: the N^2 x R reductions of the reference's evaluation
// (src/metrics/metrics.py:12-50 distances, :53-77 radius of gyration, :80-121 steric clashes / validity, :124-137 bonding validity,
// :140-166 js_pwd, :203-224 js_rg) over CA coordinates [R, L, 3] float32 (what extract_backbone_coords returns,
// src/common/pdb_utils.py:255-317) -- without the PDB round trip and without numpy's per-channel Python loop
// (np.apply_along_axis over L(L-1)/2 channels).
//
//   s2s_ca_sample_stats   per sample: number of CA pairs (|i-j| >= 1 + k_exclusion) closer than the clash bar, the largest adjacent
//                         CA-CA distance, the radius of gyration.
//   s2s_ca_pwd_js         per pair channel (i, j >= i + offset): range = [min, max] of the REFERENCE ensemble's distance, 50-bin
//                         histograms of both ensembles in numpy's own float32 bin arithmetic (np.histogram with range=), + 1e-6,
//                         Jensen-Shannon distance (scipy.spatial.distance.jensenshannon, natural log) in float64.
// Distances are formed exactly as numpy does on float32 input (subtract, square, sum x+y+z left to right, sqrt, every step
// rounded to float32; this unit is built with -ffp-contract=off), so the histogram COUNTS are those of the reference.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "str2str_hip.h"

namespace {

__device__ __forceinline__ float dist_f32(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

__global__ void __launch_bounds__(256) ca_sample_stats_kernel(const float* __restrict__ ca, int R, int L, float clash_bar, int k_excl,
                                                              int* __restrict__ n_clash, float* __restrict__ adj_max,
                                                              double* __restrict__ rg) {
    const int s = blockIdx.x;
    const float* x = ca + (long long)s * L * 3;
    __shared__ int s_cnt[4];
    __shared__ float s_max[4];
    __shared__ double s_acc[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // centre of mass (float64 here; the reference's float32 mean differs in the last bits only)
    double cx = 0, cy = 0, cz = 0;
    for (int i = tid; i < L; i += 256) { cx += x[3 * i]; cy += x[3 * i + 1]; cz += x[3 * i + 2]; }
    for (int o = 32; o > 0; o >>= 1) { cx += __shfl_down(cx, o, 64); cy += __shfl_down(cy, o, 64); cz += __shfl_down(cz, o, 64); }
    if (lane == 0) { s_acc[wave][0] = cx; s_acc[wave][1] = cy; s_acc[wave][2] = cz; }
    __syncthreads();
    const double mx = (s_acc[0][0] + s_acc[1][0] + s_acc[2][0] + s_acc[3][0]) / L, my = (s_acc[0][1] + s_acc[1][1] + s_acc[2][1] + s_acc[3][1]) / L,
                 mz = (s_acc[0][2] + s_acc[1][2] + s_acc[2][2] + s_acc[3][2]) / L;
    __syncthreads();
    double sq = 0;
    int cnt = 0;
    float amax = 0.f;
    for (int i = tid; i < L; i += 256) {
        const double dx = x[3 * i] - mx, dy = x[3 * i + 1] - my, dz = x[3 * i + 2] - mz;
        sq += dx * dx + dy * dy + dz * dz;
        if (i + 1 < L) amax = fmaxf(amax, dist_f32(x + 3 * i, x + 3 * i + 3));
        for (int j = i + 1 + k_excl; j < L; ++j) cnt += dist_f32(x + 3 * j, x + 3 * i) < clash_bar;  // dX = coords[j] - coords[i] (metrics.py:34)
    }
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_down(sq, o, 64);
        cnt += __shfl_down(cnt, o, 64);
        amax = fmaxf(amax, __shfl_down(amax, o, 64));
    }
    if (lane == 0) { s_acc[wave][3] = sq; s_cnt[wave] = cnt; s_max[wave] = amax; }
    __syncthreads();
    if (tid == 0) {
        n_clash[s] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        adj_max[s] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        rg[s] = sqrt((s_acc[0][3] + s_acc[1][3] + s_acc[2][3] + s_acc[3][3]) / L);
    }
}

// pairwise_distance_ca (metrics.py:38-50): the upper-triangular CA distances of every sample, [R, D] float32 in np.triu_indices(L, k)
// order (row i holds the L - k - i pairs (i, j >= i + k)) -- the features of js_tica.  dist_f32 is numpy's arithmetic on float32 input, so the
// features equal the reference's bit for bit (the TICA whitening amplifies a last-bit difference of the features by 1 / (smallest kept
// eigenvalue of C00): scores moved in the third decimal with torch's own reduction).
__global__ void __launch_bounds__(256) ca_pairwise_kernel(const float* __restrict__ ca, int R, int L, int offset, long long D,
                                                          float* __restrict__ out) {
    const int s = blockIdx.y;
    const float* x = ca + (long long)s * L * 3;
    for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < D; ch += (long long)gridDim.x * 256) {
        long long rem = ch;
        int i = 0;
        while (rem >= (long long)(L - offset - i)) { rem -= L - offset - i; ++i; }
        const int j = i + offset + (int)rem;
        out[(long long)s * D + ch] = dist_f32(x + 3 * j, x + 3 * i);
    }
}

// numpy's uniform-bin index for x in [first, last] (numpy/lib/_histograms_impl.py, the `range=` fast path), float32 arithmetic:
//   f = (x - first) / (last - first) * bins;  idx = (int)f;  idx == bins -> bins - 1;  then the two edge corrections against
//   edges[k] = linspace(first, last, bins + 1) in float32 (start + k * step, last edge = stop).
__device__ __forceinline__ int np_bin(float x, float first, float last, int bins) {
    const float denom = last - first;
    const float f = ((x - first) / denom) * (float)bins;
    int idx = (int)f;
    if (idx == bins) idx -= 1;
    const float step = denom / (float)bins;
    auto edge = [&](int k) { return k == bins ? last : first + (float)k * step; };
    if (x < edge(idx)) idx -= 1;
    else if (x >= edge(idx + 1) && idx != bins - 1) idx += 1;
    return idx;
}

// WEIGHTED: per-sample float64 weights (the reference's `weights=`, metrics.py:139-150: np.histogram(..., weights=w) sums the weights of
// a bin's samples in float64; here in the order the lanes' atomics land -- the sums differ from numpy's in the last bits only).
template <int BINS, bool WEIGHTED>
__global__ void __launch_bounds__(256) ca_pwd_js_kernel(const float* __restrict__ ref, int Rt, const float* __restrict__ pred, int R, int L,
                                                        int offset, long long n_ch, double pseudo, double* __restrict__ js,
                                                        const double* __restrict__ w_ref, const double* __restrict__ w_pred) {
    typedef typename std::conditional<WEIGHTED, double, int>::type hist_t;
    __shared__ hist_t s_hist[4][2][BINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ch = (long long)blockIdx.x * 4 + wave;
    if (ch >= n_ch) return;  // wave-uniform; only wave-level synchronisation below
    // channel -> (i, j) in np.triu_indices(L, k=offset) order: row i has L - offset - i entries
    long long rem = ch;
    int i = 0;
    while (rem >= (long long)(L - offset - i)) { rem -= L - offset - i; ++i; }
    const int j = i + offset + (int)rem;
    for (int b = lane; b < 2 * BINS; b += 64) s_hist[wave][b / BINS][b % BINS] = 0;
    float dmin = INFINITY, dmax = -INFINITY;
    for (int s = lane; s < Rt; s += 64) {
        const float* x = ref + (long long)s * L * 3;
        const float d = dist_f32(x + 3 * j, x + 3 * i);
        dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
    }
    for (int o = 32; o > 0; o >>= 1) { dmin = fminf(dmin, __shfl_xor(dmin, o, 64)); dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); }
    if (dmin == dmax) { dmin -= 0.5f; dmax += 0.5f; }  // np.histogram's _get_outer_edges for a degenerate range
    __builtin_amdgcn_wave_barrier();
    for (int e = 0; e < 2; ++e) {
        const float* base = e ? pred : ref;
        const int n = e ? R : Rt;
        for (int s = lane; s < n; s += 64) {
            const float* x = base + (long long)s * L * 3;
            const float d = dist_f32(x + 3 * j, x + 3 * i);
            if (d >= dmin && d <= dmax) {
                if constexpr (WEIGHTED) {
                    const double* w = e ? w_pred : w_ref;
                    atomicAdd(&s_hist[wave][e][np_bin(d, dmin, dmax, BINS)], w ? w[s] : 1.0);
                } else {
                    atomicAdd(&s_hist[wave][e][np_bin(d, dmin, dmax, BINS)], 1);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Jensen-Shannon distance of the two (count + pseudo) vectors, float64 (scipy: normalise, m = (p+q)/2, sqrt((KL(p|m)+KL(q|m))/2))
    double p = 0, q = 0;
    if (lane < BINS) { p = s_hist[wave][1][lane] + pseudo; q = s_hist[wave][0][lane] + pseudo; }
    double sp = p, sq_ = q;
    for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sq_ += __shfl_xor(sq_, o, 64); }
    double t = 0;
    if (lane < BINS) {
        p /= sp; q /= sq_;
        const double m = 0.5 * (p + q);
        t = (p > 0 ? p * log(p / m) : 0.0) + (q > 0 ? q * log(q / m) : 0.0);
    }
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) js[ch] = sqrt(t / 2.0);
}

}  // namespace

extern "C" int s2s_ca_sample_stats(const float* ca, int n_samples, int n_res, float clash_bar, int k_exclusion, int* n_clash,
                                   float* adjacent_max, double* radius_of_gyration, void* stream) {
    if (n_samples <= 0) return 0;
    if (!ca || n_res < 2 || !n_clash || !adjacent_max || !radius_of_gyration || k_exclusion < 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ca_sample_stats_kernel, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, ca, n_samples, n_res, clash_bar,
                       k_exclusion, n_clash, adjacent_max, radius_of_gyration);
    return (int)hipGetLastError();
}

extern "C" int s2s_ca_pairwise_distances(const float* ca, int n_samples, int n_res, int offset, float* out, void* stream) {
    if (n_samples <= 0) return 0;
    if (!ca || !out || offset < 0 || n_res <= offset || n_samples > 65535) return (int)hipErrorInvalidValue;
    const long long D = (long long)(n_res - offset) * (n_res - offset + 1) / 2;
    const unsigned gx = (unsigned)((D + 255) / 256 < 4096 ? (D + 255) / 256 : 4096);
    hipLaunchKernelGGL(ca_pairwise_kernel, dim3(gx, (unsigned)n_samples), dim3(256), 0, (hipStream_t)stream, ca, n_samples, n_res, offset, D, out);
    return (int)hipGetLastError();
}

extern "C" int s2s_ca_pwd_js(const float* ref_ca, int n_ref, const float* pred_ca, int n_pred, int n_res, int offset, int n_bins,
                             double pseudo_count, double* js_per_channel, const double* ref_weights, const double* pred_weights,
                             void* stream) {
    if (!ref_ca || !pred_ca || n_ref <= 0 || n_pred <= 0 || offset < 1 || n_res <= offset || n_bins != 50 || !js_per_channel)
        return (int)hipErrorInvalidValue;
    const long long n_ch = (long long)(n_res - offset) * (n_res - offset + 1) / 2;
    if (ref_weights || pred_weights)
        hipLaunchKernelGGL((ca_pwd_js_kernel<50, true>), dim3((unsigned)((n_ch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref_ca, n_ref,
                           pred_ca, n_pred, n_res, offset, n_ch, pseudo_count, js_per_channel, ref_weights, pred_weights);
    else
        hipLaunchKernelGGL((ca_pwd_js_kernel<50, false>), dim3((unsigned)((n_ch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref_ca, n_ref,
                           pred_ca, n_pred, n_res, offset, n_ch, pseudo_count, js_per_channel, (const double*)nullptr, (const double*)nullptr);
    return (int)hipGetLastError();
}
