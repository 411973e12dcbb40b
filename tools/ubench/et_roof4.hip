// Would a 4-way hidden-width split of the edge transition beat the pair-per-lane kernel?  (round 4 question)
// Today every wave owns 32 pairs x all 384 hidden channels and reads EVERY weight fragment from LDS (4 KiB per 6 MFMAs), the
// workgroup copying the 0.94 MB weight stream global -> VGPR -> LDS once per 128 pairs.  The alternative: a workgroup tile of 128
// pairs (4 pair tiles), wave w owns hidden channels [96 w, 96 w + 96): each weight fragment is used by exactly one wave, for 4 pair
// tiles (2 fragments -> 12 MFMAs), and can come straight from L2 into VGPRs; the ACTIVATIONS travel through LDS instead
// (B fragments: 8 x 1 KiB per k-step for 36 MFMAs; each wave writes the planes of the 32-channel tile it produced: 16 KiB per 8
// k-steps), two barriers per round of 8 k-steps.  This file times the steady state of layer 2 of both forms, same random operand
// bits, same VALU filler per MFMA, long enough for the power cap to act:
//   kA  the slot skeleton of csrc/pair_mlp_f16.hip (et_roof.hip level 3)
//   kB  the width-split skeleton
//   hipcc --offload-arch=gfx950 -O3 et_roof4.hip -o et_roof4 && ./et_roof4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) u32x4 lds_frag;
typedef __attribute__((address_space(3))) f32x4 lds_f4;

__device__ __forceinline__ f32x16 mm(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned rnd16(unsigned& seed) {
    seed = seed * 1664525u + 1013904223u;
    return ((seed >> 4) & 0x03ff03ffu) | 0x34003400u;   // two f16 values in [0.25, 0.5)
}

// ---- kA: today's slot skeleton (6 MFMAs, 4 LDS weight fragments, weight stream copy, barrier per 8 slots, V VALU per slot)
template <int V>
__global__ void __launch_bounds__(256) kA(float* out, unsigned long long* cyc, const char* wblob, int iters) {
    constexpr int SPS = 8, kStage = SPS * 4096, kQ = SPS / 2;
    __shared__ __attribute__((aligned(16))) char s_w[2][kStage];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_char* img[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[1][lane * 16]};
    asm volatile("" : "+v"(img[0]), "+v"(img[1]));
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * kStage / 4; i += 256) ((unsigned*)s_w)[i] = rnd16(seed);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 30 * 32768, 0x00020000);
    const unsigned voff = wave * (kStage / 4) + lane * 16;
    u32x4 b[2];
    for (int pl = 0; pl < 2; ++pl) for (int j = 0; j < 4; ++j) b[pl][j] = rnd16(seed);
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 fr[2][4];
    for (int kq = 0; kq < 4; ++kq) fr[0][kq] = fr[1][kq] = ((const lds_frag*)img[0])[64 * kq];
    f32x4 c[2 * kQ];
    for (int q = 0; q < 2 * kQ; ++q) c[q] = f32x4{1.f, 2.f, 3.f, 4.f};
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = 1.0f + 1e-3f * (lane + i);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2 * SPS; ++s) {
            const int ss = s % SPS, par = (s / SPS) & 1;
            if (ss == SPS - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const lds_frag* src = (const lds_frag*)img[ss == SPS - 1 ? par ^ 1 : par] + ((ss + 1) % SPS) * 4 * 64;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) fr[(s + 1) & 1][kq] = src[64 * kq];
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&f)[4] = fr[s & 1];
            const int so = ((it * 2 + (s / SPS)) % 30) * kStage + (ss >= SPS / 2 ? kQ * 1024 : 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int fa = 2 * (i & 1) + (i < 2 ? 1 : 0), xa = (i == 2 || i == 3) ? 1 : 0;
                acc[i & 1] = mm(f[fa], b[xa], acc[i & 1]);
                if (i < 4) {   // one 1 KiB piece of the weight pipe behind each of the first four MFMAs
                    if (ss == 0 || ss == 4) {
                        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * i, so, 0);
                        c[(ss ? kQ : 0) + i] = f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
                    }
                    if (ss == 1 || ss == 5) *(lds_f4*)(img[par ^ 1] + (wave * (kStage / 4) + (ss == 1 ? 0 : kQ * 1024) + 1024 * i)) = c[(ss == 1 ? 0 : kQ) + i];
                }
#pragma unroll
                for (int v = i * V / 6; v < (i + 1) * V / 6; ++v) va[v & 7] = __builtin_fmaf(va[v & 7], 1.0000001f, 1e-7f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

// ---- kB: width-split skeleton.  Per k-step and wave: 6 weight fragments straight from global (the wave's own quarter of the
// stream), 8 activation fragments from the LDS ring, 36 MFMAs on 12 accumulators (3 hidden tiles x 4 pair tiles x 3 products),
// 2 ds_write_b128 of produced planes, V VALU per 6 MFMAs; two barriers per 8 k-steps.  WST = 0: the weights through LDS as well
// (control: isolates the effect of the direct loads).
template <int V>
__global__ void __launch_bounds__(256) kB(float* out, unsigned long long* cyc, const char* wblob, int iters) {
    __shared__ __attribute__((aligned(16))) char s_ring[2][64 * 1024];   // [0] the a1 ring being read, [1] the planes being produced / X
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_char* ring = (lds_char*)&s_ring[0][lane * 16];
    lds_char* prod = (lds_char*)&s_ring[1][wave * 16384 + lane * 16];
    asm volatile("" : "+v"(ring), "+v"(prod));
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * 64 * 1024 / 4; i += 256) ((unsigned*)s_ring)[i] = rnd16(seed);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 30 * 32768, 0x00020000);
    // the wave's quarter of the weight stream: 6 KiB per k-step, 240 KiB per tile; walk 48 KiB per round, wrap at 240 KiB
    const unsigned voff = wave * (240 * 1024) + lane * 16;
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 fa[2][6], fb[2][8], pw[2];
    for (int j = 0; j < 4; ++j) { pw[0][j] = rnd16(seed); pw[1][j] = rnd16(seed); }
    for (int q = 0; q < 6; ++q) fa[0][q] = fa[1][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * q, 0, 0);
    for (int q = 0; q < 8; ++q) fb[0][q] = fb[1][q] = ((const lds_frag*)ring)[64 * q];
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = 1.0f + 1e-3f * (lane + i);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int rbase = (it % 5) * 49152;   // 5 rounds of 8 k-steps = 240 KiB
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // top of the k-step: next k-step's operands
            const int nk = (ks + 1) & 7;
#pragma unroll
            for (int q = 0; q < 8; ++q) fb[(ks + 1) & 1][q] = ((const lds_frag*)ring)[(nk * 8 + q) * 64];
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&A)[6] = fa[ks & 1];
            const u32x4 (&B)[8] = fb[ks & 1];
            // 36 MFMAs: product-major (W_l x_h, W_h x_l, W_h x_h), tile, pair tile -- consecutive MFMAs hit different accumulators
#pragma unroll
            for (int m = 0; m < 36; ++m) {
                const int prod_ = m / 12, t = (m % 12) / 4, p = m % 4;
                const int wa = 2 * t + (prod_ == 0 ? 1 : 0), xb = 2 * p + (prod_ == 1 ? 1 : 0);
                acc[4 * t + p] = mm(A[wa], B[xb], acc[4 * t + p]);
                if (m < 6) fa[(ks + 1) & 1][m] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * m, rbase + nk * 6144, 0);
                if (m == 12 || m == 24) *(lds_frag*)(prod + (2 * ks + (m == 24)) * 1024) = pw[m == 24];
#pragma unroll
                for (int v = (m % 6) * V / 6; v < (m % 6 + 1) * V / 6; ++v) va[v & 7] = __builtin_fmaf(va[v & 7], 1.0000001f, 1e-7f);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks == 3 || ks == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

template <class K> void run(const char* name, K kern, double mfma_per_iter, float* out, unsigned long long* cyc, const char* w, int iters) {
    kern<<<256, 256>>>(out, cyc, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) kern<<<256, 256>>>(out, cyc, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = iters * mfma_per_iter;
    const double tf = mf * 32768 * 1024 / (ms * 1e-3) / 1e12;
    printf("%-44s %6.1f cycles / 6 MFMAs  clock %.2f GHz  %7.2f ns / 6 MFMAs  %6.0f TFLOP/s = %.3f of 2.5 PF\n", name, 6.0 * c / mf,
           c / (ms * 1e-3) / 1e9, 6.0 * ms * 1e6 / mf, tf, tf / 2500.0);
}
int main() {
    float* out; unsigned long long* cyc; char* w;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 30 * 32768);
    {   // random f16 bits in the weight blob as well
        unsigned* h = (unsigned*)malloc(30 * 32768); unsigned s = 12345u;
        for (int i = 0; i < 30 * 32768 / 4; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 4) & 0x03ff03ffu) | 0x34003400u; }
        hipMemcpy(w, h, 30 * 32768, hipMemcpyHostToDevice); free(h);
    }
    for (int rep = 0; rep < 3; ++rep) {
        run("A  pair-per-lane slots, 16 VALU / 6 MFMAs", kA<16>, 96, out, cyc, w, 4000);
        run("B  width split,         16 VALU / 6 MFMAs", kB<16>, 288, out, cyc, w, 1333);
        run("A  pair-per-lane slots,  8 VALU / 6 MFMAs", kA<8>, 96, out, cyc, w, 4000);
        run("B  width split,          8 VALU / 6 MFMAs", kB<8>, 288, out, cyc, w, 1333);
        run("A  pair-per-lane slots,  0 VALU", kA<0>, 96, out, cyc, w, 4000);
        run("B  width split,          0 VALU", kB<0>, 288, out, cyc, w, 1333);
    }
    return 0;
}
