// Does the wave's sticky exception status (TRAPSTS.EXCP, hwreg 3) record an f32 -> f16 conversion that overflows?
// (range guard of the f16x3 kernels: a free overflow detector if it does).  hipcc --offload-arch=gfx950 trapsts_probe.hip -o trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, unsigned* out, _Float16* sink) {
    const int lane = threadIdx.x;
    unsigned before = __builtin_amdgcn_s_getreg(3 | (0 << 6) | (8 << 11));
    float x = in[lane];
    float y = in[64 + lane];
    asm volatile("" : "+v"(x), "+v"(y));
    // case selected per block: 0 = scalar cvt, 1 = packed cvt (v_cvt_pk_f16_f32 / cvt_pkrtz), 2 = no overflow, 3 = fp32 multiply overflow
    _Float16 h = 0;
    if (blockIdx.x == 0) h = (_Float16)x;
    else if (blockIdx.x == 1) { auto p = __builtin_amdgcn_cvt_pkrtz(x, y); h = (_Float16)p[0] + (_Float16)p[1]; }
    else if (blockIdx.x == 2) h = (_Float16)y;
    else if (blockIdx.x == 3) { float z = x * 1e35f; asm volatile("" : "+v"(z)); h = (_Float16)(z > 1.f ? 1.f : 0.f); }
    else if (blockIdx.x == 4) { f16x2 p; p[0] = (_Float16)x; p[1] = (_Float16)y; asm volatile("" : "+v"(p)); h = p[0] + p[1]; }   // rn pair (what the kernels do)
    sink[blockIdx.x * 64 + lane] = h;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    unsigned after = __builtin_amdgcn_s_getreg(3 | (0 << 6) | (8 << 11));
    if (lane == 0) { out[2 * blockIdx.x] = before; out[2 * blockIdx.x + 1] = after; }
}

int main() {
    float h_in[128];
    for (int i = 0; i < 64; ++i) { h_in[i] = i == 17 ? 70000.0f : 1.5f; h_in[64 + i] = 2.25f; }   // ONE lane overflows
    float* d_in; unsigned* d_out; _Float16* d_sink;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 64); hipMalloc(&d_sink, 5 * 64 * 2);
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(5), dim3(64), 0, 0, d_in, d_out, d_sink);
    unsigned h_out[10];
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    const char* names[5] = {"scalar cvt of 70000 (one lane)", "cvt_pkrtz of 70000", "no overflow", "fp32 multiply overflow", "rn pair cvt of 70000"};
    for (int b = 0; b < 5; ++b) printf("%-34s TRAPSTS.EXCP before 0x%03x after 0x%03x  overflow bit %u\n", names[b], h_out[2 * b], h_out[2 * b + 1], (h_out[2 * b + 1] >> 3) & 1);
    return 0;
}
