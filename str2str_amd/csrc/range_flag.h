// Range guard of the split-f16 ("f16x3") kernels.
//
// An activation that is split into f16 planes must stay below f16's 65504 (fp32 has 8 exponent bits, the planes 5).  Every kernel
// that performs such a split keeps a running maximum of the values it splits and, when that maximum reaches 2^15 (or is not
// finite), ORs its bit into ONE device word, the library's range flag.  Nobody waits for it: the sampler clears the word before a
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

__device__ __forceinline__ float range_max(float amax, float v) { return fmaxf(amax, fabsf(v)); }
// (NaN inputs do not move a maximum: a NaN activation is the fp32 reference's result too; an infinity does)
__device__ __forceinline__ void range_report(int* flag, float amax, int bit) {
    if (flag && !(amax < kRangeLimit)) atomicOr(flag, bit);
}

}  // namespace s2s
