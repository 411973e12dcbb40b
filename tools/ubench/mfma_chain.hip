// Microbenchmark: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of accumulators used round-robin
// (dependent-accumulate latency).  One wave per SIMD (256 threads, 1 block per CU).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(0.5f - j * 0.01f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC> void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    k<NACC><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NACC><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * (48 / NACC) * NACC;
    printf("accumulators=%d: %.1f cycles/MFMA (s_memtime), %.1f TFLOP/s whole chip, clock %.2f GHz\n", NACC, c / n,
           n * 1024 * 32768.0 / (ms * 1e-3) / 1e12, c / (ms * 1e-3) / 1e9);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc);
    return 0;
}
