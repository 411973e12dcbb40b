// Slot skeleton of an f16x3 formulation (two-way f16 split, products hh + h l' + l' h, 3 MFMAs per (k-step, tile)) next to the
// bf16x6 one of slot_bench.hip: 12 MFMAs per slot either way, but an f16x3 slot covers TWO k-steps and reads 12 A fragments
// (stage of 48 KiB = 4 slots, barrier + weight copy twice as often).  Same work unit = one bf16x6 slot == half an f16x3 slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, const char* wblob, int iters) {
    __shared__ __attribute__((aligned(16))) char s_w[2][49152];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) u32x4 lds_frag;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_char* img[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[1][lane * 16]};
    asm volatile("" : "+v"(img[0]), "+v"(img[1]));
    // pseudo-random operand bits (power depends on toggling): small normal numbers in either format
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * 49152 / 4; i += 256) {
        seed = seed * 1664525u + 1013904223u;
        const unsigned lo = (seed >> 4) & 0x03ff03ffu;   // mantissa bits
        ((unsigned*)s_w)[i] = F16 ? (lo | 0x34003400u) : (lo & 0x007f007fu) | 0x3e803e80u;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 30 * 49152, 0x00020000);
    const unsigned voff = wave * 12288 + lane * 16;
    u32x4 b[6];
    for (int pl = 0; pl < 6; ++pl) {
        seed = seed * 1664525u + 1013904223u;
        for (int j = 0; j < 4; ++j) { seed = seed * 1664525u + 1013904223u; b[pl][j] = F16 ? ((seed & 0x03ff03ffu) | 0x34003400u) : ((seed & 0x007f007fu) | 0x3e803e80u); }
    }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NF = F16 ? 12 : 6;          // fragments per slot
    constexpr int SPS = F16 ? 4 : 8;          // slots per 48 KiB stage
    u32x4 fr[2][NF];
    for (int kq = 0; kq < NF; ++kq) fr[0][kq] = ((const lds_frag*)img[0])[64 * kq];
    f32x4 c[6];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2 * SPS; ++s) {  // two stages
            const int ss = s % SPS, par = (s / SPS) & 1;
            if (ss == SPS - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const lds_frag* src = (const lds_frag*)img[ss == SPS - 1 ? par ^ 1 : par] + ((ss + 1) % SPS) * NF * 64;
#pragma unroll
                for (int kq = 0; kq < NF; ++kq) fr[(s + 1) & 1][kq] = src[64 * kq];
            }
            // weight copy: 12 KiB per wave per stage in two halves
            if (ss == 0 || ss == SPS / 2) {
                const int so = ((it * 2 + (s / SPS)) % 30) * 49152 + (ss ? 6144 : 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * q, so, 0);
                    c[q] = f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&f)[NF] = fr[s & 1];
            f32x16 &t0_ = acc[0], &t1_ = acc[1];
            if constexpr (F16) {
                auto mm = [&](const u32x4& a, const u32x4& bb, f32x16 cc) {
                    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), cc, 0, 0, 0);
                };
#pragma unroll
                for (int u = 0; u < 2; ++u) {  // k-step u: planes (h, h5, l5) of two tiles = f[6u .. 6u+5]
                    t0_ = mm(f[6 * u + 0], b[3 * u + 0], t0_); t1_ = mm(f[6 * u + 3], b[3 * u + 0], t1_);
                    t0_ = mm(f[6 * u + 1], b[3 * u + 1], t0_); t1_ = mm(f[6 * u + 4], b[3 * u + 1], t1_);
                    t0_ = mm(f[6 * u + 2], b[3 * u + 2], t0_); t1_ = mm(f[6 * u + 5], b[3 * u + 2], t1_);
                }
            } else {
                auto mm = [&](const u32x4& a, const u32x4& bb, f32x16 cc) {
                    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), cc, 0, 0, 0);
                };
                t0_ = mm(f[2], b[0], t0_); t1_ = mm(f[5], b[0], t1_);
                t0_ = mm(f[0], b[2], t0_); t1_ = mm(f[3], b[2], t1_);
                t0_ = mm(f[1], b[1], t0_); t1_ = mm(f[4], b[1], t1_);
                t0_ = mm(f[1], b[0], t0_); t1_ = mm(f[4], b[0], t1_);
                t0_ = mm(f[0], b[1], t0_); t1_ = mm(f[3], b[1], t1_);
                t0_ = mm(f[0], b[0], t0_); t1_ = mm(f[3], b[0], t1_);
            }
            if (ss == 1 || ss == SPS / 2 + 1) {
                lds_char* d = img[par ^ 1] + (wave * 12288 + (ss == 1 ? 0 : 6144));
#pragma unroll
                for (int q = 0; q < 6; ++q) *(lds_f4*)(d + 1024 * q) = c[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

template <bool F16> void run(const char* name, float* out, unsigned long long* cyc, const char* w, int iters) {
    constexpr int SPS = F16 ? 4 : 8;
    k<F16><<<256, 256>>>(out, cyc, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k<F16><<<256, 256>>>(out, cyc, w, iters);   // ~0.3 s: long enough for the power cap to act
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double slots = iters * 2.0 * SPS;
    const double units = slots * (F16 ? 2 : 1);   // bf16x6-slot equivalents of work (one k-step x two tiles)
    printf("%-28s %6.1f cycles/slot  clock %.2f GHz  %.0f TF executed  %.3f us per 1000 work units (k-step x 2 tiles) per wave\n", name,
           c / slots, c / (ms * 1e-3) / 1e9, slots * 12 * 32768 * 1024 / (ms * 1e-3) / 1e12, ms * 1e3 / units * 1000);
}
int main() {
    float* out; unsigned long long* cyc; char* w;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 30 * 49152); hipMemset(w, 0x35, 30 * 49152);
    for (int rep = 0; rep < 2; ++rep) {
        run<false>("bf16x6 slot (6 frags)", out, cyc, w, 1500);
        run<true>("f16x3 slot (12 frags)", out, cyc, w, 3000);
    }
    return 0;
}
