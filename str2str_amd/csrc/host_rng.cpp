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
constexpr int kN = 624, kM = 397;
inline uint32_t mix(uint32_t u, uint32_t v) { return ((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1 ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
// at::mt19937::next_state on 64-bit slots that hold 32-bit words (the layout of torch's serialised CPU generator state)
void twist(uint64_t* s) {
    uint32_t w[kN];
    for (int i = 0; i < kN; ++i) w[i] = (uint32_t)s[i];
    int j = 0;
    for (; j < kN - kM; ++j) w[j] = w[j + kM] ^ mix(w[j], w[j + 1]);
    for (; j < kN - 1; ++j) w[j] = w[j + kM - kN] ^ mix(w[j], w[j + 1]);
    w[kN - 1] = w[kM - 1] ^ mix(w[kN - 1], w[0]);
    for (int i = 0; i < kN; ++i) s[i] = w[i];
}
}  // namespace

extern "C" int s2s_mt19937_discard(unsigned long long* state624, int* left, unsigned long long* next, unsigned long long n_outputs) {
    if (!state624 || !left || !next || *left < 1 || *left > kN || *next > (unsigned long long)kN) return 1;
    uint64_t* s = reinterpret_cast<uint64_t*>(state624);
    unsigned long long k = n_outputs;
    // one output:  if (--left == 0) { twist; left = 624; next = 0; }  y = state[next++]
    if (k >= (unsigned long long)*left) {      // the left-th call from here twists and consumes word 0
        k -= (unsigned long long)*left;
        twist(s);
        while (k >= (unsigned long long)kN) {  // 623 plain outputs, the 624th twists again
            twist(s);
            k -= kN;
        }
        *left = kN;
        *next = 1;
    }
    *left -= (int)k;
    *next += k;
    return 0;
}
