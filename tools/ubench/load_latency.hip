// How long does a wavefront wait for a global load that misses L2, as a function of how far the address is from the previous one?
// (The edge-transition kernel reads each 64 KiB pair tile once; its tiles are 16 MiB apart per workgroup.)  One wave per CU on every
// CU, a 4 GiB buffer, each wave issues ONE global_load_dwordx4 per step (coalesced 1 KiB per wave) and waits for it; s_memtime around.
//   hipcc --offload-arch=gfx950 -O3 load_latency.hip -o load_latency && ./load_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) lat(const float4* buf, size_t stride_f4, size_t wave_off_f4, int steps, unsigned long long* out, float* sink) {
    const float4* p = buf + blockIdx.x * wave_off_f4 + threadIdx.x;
    float acc = 0.f;
    unsigned long long total = 0, worst = 0;
    for (int s = 0; s < steps; ++s) {
        unsigned long long t0, t1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(v)::"memory");
        acc += v[0];
        total += t1 - t0;
        worst = t1 - t0 > worst ? t1 - t0 : worst;
        p += stride_f4;
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = total; out[2 * blockIdx.x + 1] = worst; }
    if (acc == 123.456f) *sink = acc;
}

int main() {
    const size_t bytes = 4ull << 30;
    float4* buf; unsigned long long* out; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&out, 512 * 8); hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    struct Case { const char* name; size_t stride, wave_off; int steps; } cases[] = {
        {"same 1 KiB again (L2 / TCP hit)", 0, 16 << 20, 64},
        {"next 1 KiB (stride 1 KiB, fresh lines, same page)", 1 << 10, 16 << 20, 64},
        {"stride 16 KiB", 16 << 10, 16 << 20, 64},
        {"stride 64 KiB (a workgroup walking consecutive tiles)", 64 << 10, 16 << 20, 64},
        {"stride 2 MiB", 2 << 20, 2 << 10, 64},
        {"stride 16 MiB (the kernel's tile order: 256 workgroups x 64 KiB)", 16 << 20, 64 << 10, 64},
    };
    for (auto& c : cases) {
        // flush L2 / MALL with a big memset between cases
        hipMemset(buf, 2, bytes);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(lat, dim3(256), dim3(64), 0, 0, buf, c.stride / 16, c.wave_off / 16, c.steps, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
        double tot = 0; unsigned long long worst = 0;
        for (int i = 0; i < 256; ++i) { tot += h[2 * i]; worst = h[2 * i + 1] > worst ? h[2 * i + 1] : worst; }
        printf("%-70s mean %7.0f ticks  worst %7llu ticks (s_memtime, 100 MHz => x10 ns)\n", c.name, tot / 256 / c.steps, worst);
    }
    return 0;
}
