// Host-side fast-forward of the Mersenne twister behind torch's CPU generator (host code, no device work).
//
// The reference consumes two float64 normal draws of a whole replica chunk per denoise step even under the probability-flow ODE, where
// they are not used (torch.randn_like in src/models/score/so3.py:360 and src/models/score/r3.py:109).  Parity mode must leave the host
// generator where the reference leaves it, so the sampler consumed the same draws (sampler._burn_step_draws) -- ~20 ns per double in
// torch.randn, 4750 x 2 chunks x 2 draws per target of the reference's default inference block = seconds of host time in front of
// the GPU work.  What a discarded draw leaves behind is only the engine's position: ATen fills a float64 normal tensor of n >= 16
// elements from n (+ 16 more when n % 16 != 0: the tail block is drawn again) uniform doubles of two 32-bit engine outputs each
// (aten/src/ATen/native/cpu/DistributionTemplates.h normal_fill, ATen/core/DistributionsHelper.h uniform_real_distribution<double>),
// so the state after the draws is the state after discarding that many outputs of at::mt19937 (ATen/core/MT19937RNGEngine.h).
// s2s_mt19937_discard does exactly that on the engine fields of a generator state the caller parsed (ops.host_rng_discard, which
// checks the whole scheme against real draws once per process and falls back to them if the layout ever differs).
#include <cstdint>

#include "str2str_hip.h"

namespace {
constexpr int kN = 624, kM = 397, kD = kN - kM;
inline uint32_t mix(uint32_t u, uint32_t v) {
    uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
// n times at::mt19937::next_state, in place on 32-bit words.  Three runs of at most 227 words: inside a run word j needs the OLD words
// j and j + 1 and a word 227 behind (new, finished by the run before) or 397 ahead (old), never a word the same run writes -- every run
// vectorises, and the compiler emits one clone per instruction set for the loader to pick (43 M outputs = the step draws of one
// 64-replica chunk of an 80-residue target over 700 steps: 26-38 ms as one scalar pass per twist through the 64-bit slots, 8-14 ms so
// on the x86-64 baseline).
#if defined(__x86_64__) && defined(__ELF__) && defined(__gnu_linux__)
#define S2S_TWIST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else   // (function multi-versioning is x86 + ifunc only; elsewhere the plain function: an optional speed-up must not break the build)
#define S2S_TWIST_CLONES
#endif
S2S_TWIST_CLONES void twist_n(uint32_t* __restrict w, unsigned long long n) {
    for (unsigned long long t = 0; t < n; ++t) {
        for (int j = 0; j < kD; ++j) w[j] = w[j + kM] ^ mix(w[j], w[j + 1]);
        for (int j = kD; j < 2 * kD; ++j) w[j] = w[j - kD] ^ mix(w[j], w[j + 1]);
        for (int j = 2 * kD; j < kN - 1; ++j) w[j] = w[j - kD] ^ mix(w[j], w[j + 1]);
        w[kN - 1] = w[kM - 1] ^ mix(w[kN - 1], w[0]);
    }
}
}  // namespace

extern "C" int s2s_mt19937_discard(unsigned long long* state624, int* left, unsigned long long* next, unsigned long long n_outputs) {
    if (!state624 || !left || !next || *left < 1 || *left > kN || *next > (unsigned long long)kN) return 1;
    unsigned long long k = n_outputs;
    // one output:  if (--left == 0) { twist; left = 624; next = 0; }  y = state[next++]
    if (k >= (unsigned long long)*left) {      // the left-th call from here twists and consumes word 0; then 623 plain outputs, the 624th twists again
        k -= (unsigned long long)*left;
        alignas(64) uint32_t w[kN];            // the serialised state keeps its 32-bit words in 64-bit slots
        for (int i = 0; i < kN; ++i) w[i] = (uint32_t)state624[i];
        twist_n(w, 1 + k / kN);
        for (int i = 0; i < kN; ++i) state624[i] = w[i];
        k %= kN;
        *left = kN;
        *next = 1;
    }
    *left -= (int)k;
    *next += k;
    return 0;
}
