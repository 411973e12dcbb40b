// Does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs (the x_l planes of small values are subnormal)?  A = 2^-20 (subnormal),
// B = 2^10: every accumulator element is 16 * 2^-10 = 2^-6 if subnormals are honoured, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8); float h[2];
    const float as[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24, smallest subnormal */, 6.103515625e-05f /* 2^-14, smallest normal */};
    for (int i = 0; i < 3; ++i) {
        k<<<1, 64>>>(d, as[i], 1024.0f); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.4e (as f16 %.4e)  x 1024 x 16 terms -> acc = %.6e  (expected %.6e)\n", as[i], h[1], h[0], as[i] * 1024.0f * 16);
    }
    return 0;
}
