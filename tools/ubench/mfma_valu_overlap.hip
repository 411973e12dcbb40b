// Does non-MFMA work of a wave hide under its own MFMAs, or only under ANOTHER wave's?  (round 4)
// Per wave and loop pass: 6 independent v_mfma_f32_32x32x16_f16 (six accumulators) and, behind each, V/6 independent VALU
// instructions and D ds_read_b128 + D/..; random operand bits; every CU busy.  Launched with 1 wave per SIMD (256-thread
// workgroups, one per CU) and with 2 waves per SIMD (512-thread workgroups).  Printed: cycles per MFMA seen by ONE SIMD
// (= wave cycles per MFMA / waves per SIMD; the matrix pipe's 32 is the floor) and the chip's f16 rate.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int THREADS, bool PK = false>
__global__ void __launch_bounds__(THREADS) k(float* out, unsigned long long* cyc, int iters) {
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    u32x4 a[2], b[2];
    for (int p = 0; p < 2; ++p) for (int j = 0; j < 4; ++j) {
        seed = seed * 1664525u + 1013904223u; a[p][j] = ((seed >> 4) & 0x03ff03ffu) | 0x34003400u;
        seed = seed * 1664525u + 1013904223u; b[p][j] = ((seed >> 4) & 0x03ff03ffu) | 0x34003400u;
    }
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 vp[8];
    for (int i = 0; i < 8; ++i) vp[i] = f32x2{va[i], va[i] + 0.5f};
    const f32x2 pm = {1.0000001f, 1.0000002f}, pa = {1e-7f, 2e-7f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 1]), __builtin_bit_cast(f16x8, b[(i >> 1) & 1]), acc[i], 0, 0, 0);
#pragma unroll
                for (int v = i * V / 6; v < (i + 1) * V / 6; ++v) {
                    if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(vp[v & 7]) : "v"(pm), "v"(pa));   // two fp32 FMAs per instruction
                    else va[v & 7] = __builtin_fmaf(va[v & 7], 1.0000001f, 1e-7f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += va[i] + vp[i].x + vp[i].y;
    out[blockIdx.x * THREADS + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}
template <int V, int THREADS, bool PK = false> void run(float* out, unsigned long long* cyc) {
    const int iters = 20000 / (THREADS / 256);
    k<V, THREADS, PK><<<256, THREADS>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k<V, THREADS, PK><<<256, THREADS>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf_wave = iters * 24.0, wps = THREADS / 256;
    const double tf = mf_wave * wps * 1024 * 32768 / (ms * 1e-3) / 1e12;
    printf("%d wave(s) / SIMD, %2d %s per 6 MFMAs: %6.1f wave cycles per MFMA = %5.1f per SIMD  clock %.2f GHz  %5.0f TFLOP/s\n", (int)wps, V, PK ? "v_pk_fma_f32" : "VALU",
           c / mf_wave, c / mf_wave / wps, c / (ms * 1e-3) / 1e9, tf);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 256>(out, cyc);  run<0, 512>(out, cyc);
        run<12, 256>(out, cyc); run<12, 512>(out, cyc);
        run<24, 256>(out, cyc); run<24, 512>(out, cyc);
        run<48, 256>(out, cyc); run<48, 512>(out, cyc);
        run<12, 256, true>(out, cyc); run<24, 256, true>(out, cyc); run<48, 256, true>(out, cyc);   // packed fp32: same cost per instruction?
    }
    return 0;
}
