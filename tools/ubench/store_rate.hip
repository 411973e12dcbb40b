// What does a 1 KiB global store cost the wave that issues it?  (round 4: 16 stores cost 3-3.5 k ticks in both edge kernels' probes and
// 250-470 per store in the node GEMM epilogue.)  256 workgroups (one per CU) of W waves; every wave issues S lane-linear 16 B-per-lane
// stores (1 KiB, 8 whole cache lines each) to its own region, back to back or with `gap` dependent VALU instructions in between,
// in four flavours: plain, nt (aux 2), sc1 (aux ... ), and as 2 x dwordx2.  Prints cycles per store seen by a wave (s_memtime around
// the issue loop, NOT waiting for completion), the same including vmcnt(0), and the chip's write rate.
//   hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int S>
__global__ void __launch_bounds__(512) k(char* out, unsigned long long* cyc, int gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* base = out + ((size_t)(blockIdx.x * nw + wave) * S) * 1024 + lane * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + ((size_t)(blockIdx.x * nw + wave) * S) * 1024), 0, S * 1024, 0x00020000);
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    float f = threadIdx.x;
    unsigned long long t0, t1, t2;
    asm volatile("s_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 4
    for (int i = 0; i < S; ++i) {
        if (MODE == 0) *reinterpret_cast<u32x4*>(base + (size_t)i * 1024) = v;
        if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + (size_t)i * 1024));
        if (MODE == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, i * 1024, 0);
        if (MODE == 3) __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, i * 1024, 2);   // nt
        for (int g = 0; g < gap; ++g) f = __builtin_fmaf(f, 1.0000001f, 1e-7f);
        v.x += (unsigned)f;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (threadIdx.x == 0 && blockIdx.x == 5) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
}
template <int MODE, int S> void run(const char* name, int waves, int gap, char* out, unsigned long long* cyc) {
    k<MODE, S><<<256, 64 * waves>>>(out, cyc, gap);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k<MODE, S><<<256, 64 * waves>>>(out, cyc, gap);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-22s %d wave(s)/CU, %3d stores, gap %2d VALU: issue %6.1f ticks/store, complete %6.1f ticks/store, kernel %7.1f us = %5.2f TB/s\n", name,
           waves, S, gap, (double)c[0] / S, (double)c[1] / S, ms * 1e3, 256.0 * waves * S * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    char* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)256 * 8 * 256 * 1024); hipMalloc(&cyc, 16);
    for (int waves : {1, 4, 8}) {
        run<0, 32>("global_store_dwordx4", waves, 0, out, cyc);
        run<1, 32>("  nontemporal", waves, 0, out, cyc);
        run<2, 32>("buffer_store_dwordx4", waves, 0, out, cyc);
        run<3, 32>("  nt", waves, 0, out, cyc);
        run<0, 32>("global_store_dwordx4", waves, 16, out, cyc);
        run<0, 256>("global_store_dwordx4", waves, 0, out, cyc);
        run<1, 256>("  nontemporal", waves, 0, out, cyc);
    }
    return 0;
}
