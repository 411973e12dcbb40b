// Microbenchmark: how many VALU fillers of a given kind hide under one v_mfma_f32_32x32x16_bf16 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(float (&x)[4], float (&ag)[4], int j) {
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j & 3]) : "v"(x[(j + 1) & 3]));
    if constexpr (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[j & 3]) : "v"(x[(j + 1) & 3]));
    if constexpr (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[j & 3]) : "a"(ag[j & 3]));
    if constexpr (KIND == 3) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[j & 3]));
    if constexpr (KIND == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[0]) : "v"(x[1]));
    if constexpr (KIND == 5) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[j & 3]));
    if constexpr (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * (j & 1)]) : "v"(*(double*)&x[2 * ((j + 1) & 1)]));
}

template <int KIND, int NF>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(0.5f - j * 0.01f); }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[4] = {1.f + threadIdx.x, 2.f, 3.f, 4.f}, ag[4] = {5.f, 6.f, 7.f, 8.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NF; ++j) filler<KIND>(x, ag, j);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NF> void run(float* out, unsigned long long* cyc) {
    const int iters = 3000;
    k<KIND, NF><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    k<KIND, NF><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf(" %5.1f", c / (iters * 12.0));
}
template <int KIND> void row(const char* name, float* out, unsigned long long* cyc) {
    printf("%-22s", name);
    run<KIND, 0>(out, cyc); run<KIND, 1>(out, cyc); run<KIND, 2>(out, cyc); run<KIND, 3>(out, cyc); run<KIND, 4>(out, cyc);
    run<KIND, 5>(out, cyc); run<KIND, 6>(out, cyc); run<KIND, 7>(out, cyc); run<KIND, 8>(out, cyc);
    printf("\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    printf("cycles per MFMA with N fillers between MFMAs   N= 0     1     2     3     4     5     6     7     8\n");
    row<0>("v_add_f32", out, cyc); row<1>("v_cvt_pk_bf16_f32", out, cyc); row<2>("v_accvgpr_read_b32", out, cyc);
    row<3>("v_lshlrev_b32", out, cyc); row<4>("v_sub_f32 (dep chain)", out, cyc); row<5>("v_and_b32 literal", out, cyc);
    row<6>("v_pk_add_f32", out, cyc);
    return 0;
}
