// Which bf16 MFMA shape gives more FLOP/s under the power cap?  Random operands (power depends on data toggling), 8 independent
// accumulators, one wave per SIMD, every CU busy.  32x32x16: C traffic 8 KB per 32.8 kFLOP; 16x16x32: 2 KB per 16.4 kFLOP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k(const bf16x8* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095]; }
    float s = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 12; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SHAPE> void run(const bf16x8* src, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    k<SHAPE><<<256, 256>>>(src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<SHAPE><<<256, 256>>>(src, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = (SHAPE == 32 ? 48.0 * 32768 : 96.0 * 16384) * iters * 1024.0;
    printf("%s: %.1f cycles per MFMA, clock %.2f GHz, %.0f TFLOP/s whole chip (%.1f ms)\n", SHAPE == 32 ? "32x32x16" : "16x16x32",
           c / ((SHAPE == 32 ? 48.0 : 96.0) * iters), c / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12, ms);
}
int main() {
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    bf16x8* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 4096 * 16); hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<32>(src, out, cyc); run<16>(src, out, cyc); run<32>(src, out, cyc); run<16>(src, out, cyc);
    return 0;
}
