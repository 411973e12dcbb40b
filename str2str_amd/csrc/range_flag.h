// Range guard of the split-f16 ("f16x3") kernels.
//
// An activation that is split into f16 planes must stay below f16's 65504 (fp32 has 8 exponent bits, the planes 5).  Every kernel
// that performs such a split keeps a running maximum of the values it splits and, when that maximum reaches 2^15 (or is not
// finite), ORs its bit into word 0 of the library's range buffer (kRangeWords ints).  Nobody waits for it: the sampler clears the word before a
// trajectory chunk, reads it at the chunk's end (where it synchronises anyway) and, if a bit is set, re-runs the chunk on the exact
// fp32 kernels (str2str_amd/sampler.py).  The word is owned by the caller (s2s_set_range_flag); NULL disables the reports.
#pragma once
#include <hip/hip_runtime.h>

namespace s2s {

extern int* g_range_flag;   // device pointer registered by s2s_set_range_flag (abi.hip), or nullptr

enum RangeBit : int {
    kRangeNodeGemm = 1, kRangePackPlanes = 2, kRangeEdgeTransition = 4, kRangeEdgeEmbed = 8, kRangeIpaPoints = 16,
    kRangeEncoderAttention = 32, kRangeIpaAttention = 64,
};
constexpr float kRangeLimit = 32768.0f;   // 2^15: a factor two below f16's largest finite value
constexpr int kRangeWords = 8;            // the registered buffer: word 0 = the bits above, word 1 + log2(bit) = magnitude buckets of that family

__device__ __forceinline__ float range_max(float amax, float v) { return fmaxf(amax, fabsf(v)); }
// (NaN inputs do not move a maximum: a NaN activation is the fp32 reference's result too; an infinity does)
// HEADROOM: besides the flag bit a kernel records HOW LARGE its values were, coarsely and for free in the normal case: a maximum of
// 2^8 or more sets bit (exponent - 8, capped at 8) of its family's bucket word -- nothing is written while activations stay below
// 256, so ordinary networks pay one compare per lane and launch.  The host turns the highest bucket into "max |x| < 2^(k+9)",
// i.e. a headroom figure against 2^15 per kernel family (ops.range_headroom; bench.py config.range_headroom; tools/check_checkpoint.py).
__device__ __forceinline__ void range_report(int* flag, float amax, int bit) {
    if (!flag) return;
    if (!(amax < 256.0f)) {
        int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127 - 8;   // amax in [2^(8+e), 2^(9+e))
        e = e > 8 ? 8 : e;
        atomicOr(flag + 1 + (__builtin_ctz((unsigned)bit) & 7), 1 << e);
        if (!(amax < kRangeLimit)) atomicOr(flag, bit);
    }
}

}  // namespace s2s
