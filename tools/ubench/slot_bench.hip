// Microbenchmark of the bf16x6 slot structure: 12 MFMAs (2 accumulators) + 6 ds_read_b128 of the next slot's A
// fragments, optionally + 6 ds_write_b128 / 6 buffer loads every few slots and a workgroup barrier every 8 slots.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // bit0: ds_read frags, bit1: barrier per 8 slots, bit2: weight copy (buffer load + ds_write), bit3: 12 accumulators round robin
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, const char* wblob, int iters) {
    __shared__ __attribute__((aligned(16))) char s_w[2][49152];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) bf16x8 lds_frag;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_char* img[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[1][lane * 16]};
    asm volatile("" : "+v"(img[0]), "+v"(img[1]));
    for (int i = threadIdx.x; i < 2 * 49152 / 4; i += 256) ((float*)s_w)[i] = 0.001f * (i & 1023);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 30 * 49152, 0x00020000);
    const unsigned voff = wave * 12288 + lane * 16;
    bf16x8 b[3];
    for (int pl = 0; pl < 3; ++pl) for (int j = 0; j < 8; ++j) b[pl][j] = (__bf16)(0.5f - j * 0.01f * (pl + 1));
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fr[2][6];
    for (int kq = 0; kq < 6; ++kq) fr[0][kq] = ((const lds_frag*)img[0])[64 * kq];
    f32x4 c[6];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {  // two stages
            const int ss = s & 7, par = (s >> 3) & 1;
            if (MODE & 2) { if (ss == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
            if (MODE & 1) {
                const lds_frag* src = (const lds_frag*)img[ss == 7 ? par ^ 1 : par] + ((ss + 1) & 7) * 6 * 64;
#pragma unroll
                for (int kq = 0; kq < 6; ++kq) fr[(s + 1) & 1][kq] = src[64 * kq];
            } else {
#pragma unroll
                for (int kq = 0; kq < 6; ++kq) { fr[(s + 1) & 1][kq] = fr[s & 1][kq]; asm volatile("" : "+v"(fr[(s + 1) & 1][kq])); }
            }
            if (MODE & 4) {
                if (ss == 0 || ss == 4) {
                    const int so = ((it * 2 + (s >> 3)) % 30) * 49152 + (ss == 4 ? 6144 : 0);
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * q, so, 0);
                        c[q] = f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 (&f)[6] = fr[s & 1];
            f32x16& t0_ = (MODE & 8) ? acc[(2 * s) % 12] : acc[0];
            f32x16& t1_ = (MODE & 8) ? acc[(2 * s + 1) % 12] : acc[1];
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], b[0], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[5], b[0], t1_, 0, 0, 0);
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], b[2], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], b[2], t1_, 0, 0, 0);
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], b[1], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[4], b[1], t1_, 0, 0, 0);
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], b[0], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[4], b[0], t1_, 0, 0, 0);
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], b[1], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], b[1], t1_, 0, 0, 0);
            t0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], b[0], t0_, 0, 0, 0); t1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], b[0], t1_, 0, 0, 0);
            if (MODE & 4) {
                if (ss == 1 || ss == 5) {
                    lds_char* d = img[par ^ 1] + (wave * 12288 + (ss == 5 ? 6144 : 0));
#pragma unroll
                    for (int q = 0; q < 6; ++q) *(lds_f4*)(d + 1024 * q) = c[q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, unsigned long long* cyc, const char* w) {
    const int iters = 400;
    k<MODE><<<256, 256>>>(out, cyc, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, cyc, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %6.1f cycles/slot (384 = MFMA bound)   clock %.2f GHz   %.0f TF\n", name, c / (iters * 16.0), c / (ms * 1e-3) / 1e9,
           iters * 16.0 * 12 * 32768 * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* cyc; char* w;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 30 * 49152); hipMemset(w, 0x11, 30 * 49152);
    run<0>("MFMA only (2 accumulators)", out, cyc, w);
    run<8>("MFMA only (12 accumulators)", out, cyc, w);
    run<1>("+ ds_read fragments", out, cyc, w);
    run<9>("+ ds_read fragments (12 acc)", out, cyc, w);
    run<3>("+ ds_read + barrier/8 slots", out, cyc, w);
    run<5>("+ ds_read + weight copy", out, cyc, w);
    run<7>("+ ds_read + barrier + weight copy", out, cyc, w);
    run<15>("+ ds_read + barrier + weight copy (12 acc)", out, cyc, w);
    return 0;
}
