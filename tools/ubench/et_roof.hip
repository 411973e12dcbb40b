// What can the edge-transition kernel's instruction mix sustain on this part?  One wave per SIMD, every CU busy, random operand bits
// (power depends on toggling), runs long enough for the power cap to act.  Per "slot" (the unit of csrc/pair_mlp_f16.hip): 6
// v_mfma_f32_32x32x16_f16 on two alternating accumulators.  Variants add, one at a time, what the real kernel needs around them:
//   0  MFMAs only, operands in registers                         -> the power-limited matrix rate of this shape
//   1  + the slot's 4 weight fragments from LDS (ds_read_b128, one slot ahead)
//   2  + the weight stream: 8 KiB per wave per 8 slots global -> VGPR -> LDS, one barrier per stage
//   3  + V independent VALU instructions per slot (the splits / ReLU / accumulator reads of the real kernel: ~15 per slot)
//   hipcc --offload-arch=gfx950 -O3 et_roof.hip -o et_roof && ./et_roof
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LEVEL, int V, int SPS = 8, int FEAT = 7>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, const char* wblob, int iters) {
    constexpr int kStage = SPS * 4096;   // SPS slots of 4 fragments per stage
    __shared__ __attribute__((aligned(16))) char s_w[2][kStage];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) u32x4 lds_frag;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    lds_char* img[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[1][lane * 16]};
    asm volatile("" : "+v"(img[0]), "+v"(img[1]));
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * kStage / 4; i += 256) {
        seed = seed * 1664525u + 1013904223u;
        ((unsigned*)s_w)[i] = ((seed >> 4) & 0x03ff03ffu) | 0x34003400u;   // f16 values in [0.25, 0.5)
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 30 * 32768, 0x00020000);
    constexpr int kQ = SPS / 2;          // 1 KiB pieces per wave per half stage
    const unsigned voff = wave * (kStage / 4) + lane * 16;
    u32x4 b[2];
    for (int pl = 0; pl < 2; ++pl)
        for (int j = 0; j < 4; ++j) { seed = seed * 1664525u + 1013904223u; b[pl][j] = (seed & 0x03ff03ffu) | 0x34003400u; }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 fr[2][4];
    for (int kq = 0; kq < 4; ++kq) fr[0][kq] = fr[1][kq] = ((const lds_frag*)img[0])[64 * kq];
    f32x4 c[2 * kQ];
    for (int q = 0; q < 2 * kQ; ++q) c[q] = f32x4{1.f, 2.f, 3.f, 4.f};
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = 1.0f + 1e-3f * (lane + i);
    auto mm = [&](const u32x4& a, const u32x4& bb, f32x16 cc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), cc, 0, 0, 0);
    };
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2 * SPS; ++s) {  // two stages
            const int ss = s % SPS, par = (s / SPS) & 1;
            if constexpr (LEVEL >= 2 && (FEAT & 4)) {
                if (ss == SPS - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if constexpr (LEVEL >= 1) {
                const lds_frag* src = (const lds_frag*)img[ss == SPS - 1 ? par ^ 1 : par] + ((ss + 1) % SPS) * 4 * 64;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) fr[(s + 1) & 1][kq] = src[64 * kq];
            }
            if constexpr (LEVEL >= 2 && (FEAT & 1)) {
                if (ss == 0 || ss == SPS / 2) {
                    const int so = ((it * 2 + (s / SPS)) % (30 * 8 / SPS)) * kStage + (ss ? kQ * 1024 : 0);
#pragma unroll
                    for (int q = 0; q < kQ; ++q) {
                        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024 * q, so, 0);
                        c[(ss ? kQ : 0) + q] = f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&f)[4] = fr[LEVEL >= 1 ? (s & 1) : 0];
            f32x16 &t0_ = acc[0], &t1_ = acc[1];
            t0_ = mm(f[1], b[0], t0_); t1_ = mm(f[3], b[0], t1_);
            t0_ = mm(f[0], b[1], t0_); t1_ = mm(f[2], b[1], t1_);
            t0_ = mm(f[0], b[0], t0_); t1_ = mm(f[2], b[0], t1_);
            if constexpr (LEVEL >= 3) {
#pragma unroll
                for (int v = 0; v < V; ++v) va[v & 7] = __builtin_fmaf(va[v & 7], 1.0000001f, 1e-7f);
            }
            if constexpr (LEVEL >= 2 && (FEAT & 2)) {
                if (ss == 1 || ss == SPS / 2 + 1) {
                    lds_char* d = img[par ^ 1] + (wave * (kStage / 4) + (ss == 1 ? 0 : kQ * 1024));
#pragma unroll
                    for (int q = 0; q < kQ; ++q) *(lds_f4*)(d + 1024 * q) = c[(ss == 1 ? 0 : kQ) + q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    for (int i = 0; i < 8; ++i) sum += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 7) cyc[0] = t1 - t0;
}

template <int LEVEL, int V, int SPS = 8, int FEAT = 7> void run(const char* name, float* out, unsigned long long* cyc, const char* w, int iters) {
    k<LEVEL, V, SPS, FEAT><<<256, 256>>>(out, cyc, w, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) k<LEVEL, V, SPS, FEAT><<<256, 256>>>(out, cyc, w, iters);   // ~0.5 s: long enough for the power cap to act
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double slots = iters * 2.0 * SPS;
    const double tf = slots * 6 * 32768 * 1024 / (ms * 1e-3) / 1e12;
    printf("%-58s %6.1f cycles/slot (192 = matrix pipe)  counter clock %.2f GHz  %6.0f TFLOP/s = %.3f of 2.5 PF\n", name, c / slots,
           c / (ms * 1e-3) / 1e9, tf, tf / 2500.0);
}
int main() {
    float* out; unsigned long long* cyc; char* w;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8); hipMalloc(&w, 30 * 32768); hipMemset(w, 0x35, 30 * 32768);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("0 MFMAs only (register operands)", out, cyc, w, 4000);
        run<1, 0>("1 + 4 LDS weight fragments per slot", out, cyc, w, 4000);
        run<2, 0>("2 + weight stream global->VGPR->LDS, barrier per stage", out, cyc, w, 4000);
        run<3, 8>("3 + 8 VALU per slot", out, cyc, w, 4000);
        run<3, 16>("3 + 16 VALU per slot", out, cyc, w, 4000);
        run<3, 24>("3 + 24 VALU per slot", out, cyc, w, 4000);
        run<3, 16, 16>("3 + 16 VALU per slot, 16-slot (64 KiB) stages", out, cyc, w, 2000);
        run<2, 0, 8, 1>("2 pieces: global loads only", out, cyc, w, 4000);
        run<2, 0, 8, 2>("2 pieces: LDS stores only", out, cyc, w, 4000);
        run<2, 0, 8, 4>("2 pieces: barrier only", out, cyc, w, 4000);
        run<2, 0, 8, 3>("2 pieces: loads + stores, no barrier", out, cyc, w, 4000);
        run<2, 0, 8, 6>("2 pieces: stores + barrier", out, cyc, w, 4000);
    }
    return 0;
}
