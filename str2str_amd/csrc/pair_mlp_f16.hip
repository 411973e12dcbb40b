// EdgeTransition / edge embedding on split-f16 MFMA ("f16x3"): fp32-equivalent accuracy on the 16-bit matrix cores, the DEFAULT
// arithmetic of the pair stream.  Same operators and contracts as s2s_edge_transition / s2s_edge_embed (csrc/pair_mlp.hip, the
// exact fp32-MFMA kernels; reference EdgeTransition.forward, src/models/net/layers.py:170-185 + mask ipa.py:372, and
// EmbeddingModule.forward, denoising_ipa.py:137-158).
//
// Arithmetic.  Every fp32 operand is split into two f16 numbers  x = x_h + x_l (+ <= 2^-24 |x|),  x_h = rn16(x), x_l = rn16(x - x_h)
// -- 11 + 11 significant bits plus the sign of the residue, i.e. fp32's 24 -- and a product keeps  w_h x_h + w_h x_l + w_l x_h  (the
// dropped w_l x_l is below one fp32 rounding of the product) on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 MFMAs per
// (k-step, tile).  f16's narrow exponent is handled by one exact power-of-two scaling: the weight side stores the split of 2^5 w,
// W_h = rn16(32 w), W_l = rn16(32 w - W_h)  (so that the small part stays in f16's normal range; 2 A fragments per (k-step, tile),
// 4 KiB per slot, 32 KiB stages, 0.94 MB stream), the activation side two plane registers x_h, x_l, and the accumulators carry
// 32 x the layer output: the factor 2^-5 rides in the multiply-add that adds the bias / seeds in every epilogue (LayerNorm, scale
// invariant, runs on the scaled values with 1024 eps).  A value of x_l below f16's normal range (|x| < 0.25) is rounded to 2^-25
// absolute -- 1.7e-8 rms on an O(1) output, under fp32's own rounding.
// RANGE: activations must stay below f16's 65504.  Every activation that is split here feeds a running maximum; a tile whose
// maximum reaches 2^15 (or is not finite) raises the library's range flag (range_flag.h), and the sampler re-runs that chunk on the
// exact fp32 kernels (str2str_amd/sampler.py) -- an overflow is neither silent nor an error.  Weights with |32 w| >= 65504 are
// refused at pack time (ops.pack_f16x2_layer).
//
// Schedule.  One wave owns 32 pairs; the 4 waves of a workgroup share the weight stream (30 stages of 32 KiB through a ring of
// LDS buffers, see s_w).  A stage is 8 SLOTS of 6 MFMAs / 4 A fragments.  Per pair tile (240 slots):
//   A_t  (4 slots)  layer-1 output tile t (32 of the 384 hidden channels) over the 8 k-steps of the 128 edge channels;
//                   the edge row is split once into 8 x 2 f16 plane registers.
//   B_t  (12 slots) layer-2 k-steps 2t, 2t+1 (= the 32 channels of a1 tile t) into all 12 output tiles; the 192
//                   layer-2 accumulators stay resident, a1 is never materialised beyond two tiles.  B_11 runs tile-pair major
//                   (its first output tiles are complete early), every other B_t k-step major.
//   order: A_0 A_1 | B_0 A_2 | B_1 A_3 | ... | B_9 A_11 | B_10 B_11 | final layer (48 slots, k-step major) [| projection, 8 slots].
//   A slot = six MFMAs, each followed by at most one 1 KiB piece of the weight pipe and one small VALU piece, pinned there by
//   sched_barrier (the slot body in the kernel has the table).  Found with the in-kernel probe (-DS2S_ET_PROBE, tools/
//   et_phase_probe.py): a slot of bare MFMAs runs at the matrix pipe's 192 cycles; the four loads / four LDS stores of the weight
//   pipe issued as a block cost 60 / 120 cycles, a 20-instruction VALU block behind one MFMA ~100.  So:
//     - the ReLU + per-node seeds + split of a1 tile t-1 runs in 2-value halves behind MFMAs 2 and 4 of the A_t slots;
//     - the layer-2 epilogue (ReLU + residual + split = the final layer's input) runs in 4-value pieces: block 0 under the second
//       half of B_11, block 1 under the final layer's k-steps 0..7 into a second plane buffer, block 2 under k-steps 8..15;
//     - the next tile's edge row is requested two loads per slot over two stages (a 16-load burst held its slot for 2.4 k cycles) and
//       split in halves under the last final-layer block; its per-node seeds a quarter behind each of four MFMAs;
//     - what stays exposed: tile 11 of layer 1 -> planes (0.4 k cycles) and LayerNorm + store (5.2 k of a tile's 78 k).
//   Every activation is split exactly once.  C->B chaining as in pair_mlp.hip: element j of lane (pair, g) in k-step 2t'+u is
//   accumulator register 8u+j of tile t' (row 32t' + (r&3) + 8(r>>2) + 4g); the host packs A fragments in that k order and in the
//   slot order above (ops.pack_f16x3_stream; part of the ABI version).
//   Weight pipe: this wave's quarter of the next stage travels global -> VGPR -> LDS in two halves, each loaded (buffer
//   loads: SGPR base + constant lane offset) five slots before it is stored; the workgroup barrier sits at the top
//   of the last slot of a stage, after which the next stage's first fragments are fetched one slot ahead.
//   Workgroups are persistent (one per CU).  With the fused projection (PROJ) the stream has a 31st stage and the two LDS buffers
//   swap roles after every tile (odd stage count).
// Pair-tensor layouts: row-major [B,N,N,128] (the reference's) or TILED on either side (include/str2str_hip.h): a lane owns one pair,
//   so on row-major rows every load / store instruction touches 32 B in each of 32 cache lines; tiled, 8 whole lines.  The last
//   EdgeTransition of a trunk writes no pair tensor at all (io_layout bit 2).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "range_flag.h"
#include "str2str_hip.h"

// Three translation units from this one source (the four edge-transition instantiations + the two of the edge embedding took 7.5 minutes
// of a forced build as one unit): S2S_PM_PART 1 (this file compiled directly) = the edge transition for chains of 32+ residues and the
// public entry point, 2 (pair_mlp_f16_b.hip) = its per-lane-seed form for shorter chains, 3 (pair_mlp_f16_c.hip) = the edge embedding.
#ifndef S2S_PM_PART
#define S2S_PM_PART 1
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 32 * 1024;  // 8 slots x 4 fragments x 1 KiB
constexpr int kStagesBase = 30;         // + 1 stage (8 slots) for the fused pair projection of the next IPA block
constexpr float kWS = 32.0f, kInvWS = 1.0f / 32.0f;   // weights are packed as 2^5 w (see the header): accumulators carry 32 x the layer output

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 ldg4(const float* __restrict__ base, int g, int h) {
    return *reinterpret_cast<const float4*>(base + 8 * g + 4 * h);
}
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// Four fp32 values -> elements at .. at + 3 of the plane fragments (x_h, x_l), and into the range maximum (range_flag.h).
//   x_h = rn16(x):            v_cvt_pk_f16_f32, two values per instruction
//   x_l = rn16(x - x_h):      the difference is exact in fp32, so ONE fused multiply-add that reads x_h as f16 and rounds to f16
//                             (v_fma_mixlo / mixhi_f16:  (-x_h) * 1.0 + x) gives the bits of convert-back + subtract + convert
// = 1.5 VALU instructions per value (hipcc's expansion of the C expression: 3), written as ONE opaque block: both planes come from
// the same materialised fp32 value (node_gemm.hip split8_f16), the block stays where it is written (the schedule pins VALU pieces
// under specific MFMAs), and the maximum does not enter the compiler's reasoning (as an fmaxf chain it cost 300 spilled registers).
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
// two values -> elements at, at + 1 (at even) of the plane fragments: the unit the edge transition places behind single MFMAs
__device__ __forceinline__ void split2_f16(float x0, float x1, f16x8& ph, f16x8& pl, int at, float& amax) {
    unsigned hh, ll;
    asm volatile(
        "v_max3_f32 %2, %2, |%3|, |%4|\n\t"
        "v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mixlo_f16 %1, -%0, 1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, -%0, 1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hh), "=&v"(ll), "+v"(amax)
        : "v"(x0), "v"(x1));
    u32x4p hv = __builtin_bit_cast(u32x4p, ph), lv = __builtin_bit_cast(u32x4p, pl);
    hv[at / 2] = hh;
    lv[at / 2] = ll;
    ph = __builtin_bit_cast(f16x8, hv);
    pl = __builtin_bit_cast(f16x8, lv);
}
__device__ __forceinline__ void split4_f16(const float (&x)[4], f16x8& ph, f16x8& pl, int at, float& amax) {
    unsigned h0, h1, l0, l1;
    asm volatile(
        "v_max3_f32 %4, %4, |%5|, |%6|\n\t"
        "v_cvt_pk_f16_f32 %0, %5, %6\n\t"
        "v_max3_f32 %4, %4, |%7|, |%8|\n\t"
        "v_cvt_pk_f16_f32 %1, %7, %8\n\t"
        "v_fma_mixlo_f16 %2, -%0, 1.0, %5 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, -%1, 1.0, %7 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, -%0, 1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, -%1, 1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "+v"(amax)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
    u32x4p hv = __builtin_bit_cast(u32x4p, ph), lv = __builtin_bit_cast(u32x4p, pl);
    hv[at / 2] = h0; hv[at / 2 + 1] = h1;
    lv[at / 2] = l0; lv[at / 2 + 1] = l1;
    ph = __builtin_bit_cast(f16x8, hv);
    pl = __builtin_bit_cast(f16x8, lv);
}

// max(x, 0) as ONE instruction.  fmaxf() of a value the compiler cannot see through (a pinned register, an MFMA result) is preceded by
// a canonicalising v_max_f32 x, x (IEEE quieting of a signalling NaN): 128 of the edge embedding's ~1500 VALU instructions per tile.
// v_max_f32 itself returns the other operand for any NaN input, which is what the two-instruction form computes as well.
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// slot s of the schedule: phase 0 = A (layer 1), 1 = B (layer 2), 2 = F (final layer)
struct SlotDesc { int phase, t, a, b; };  // A: tile t, a = k-step pair 0..3;  B: tile t, a = u (k-step 2t+u), b = tile pair 0..5 (B_11: pair-major);
                                          // F: a = k-step 0..23, b = tile pair 0..1
constexpr SlotDesc slot_desc(int s) {
    if (s < 8) return {0, s / 4, s % 4, 0};
    if (s < 168) {
        const int tau = s - 8, blk = tau / 16, o = tau % 16;
        if (o < 12) return {1, blk, o / 6, o % 6};
        return {0, blk + 2, o - 12, 0};
    }
    if (s < 180) return {1, 10, (s - 168) / 6, (s - 168) % 6};
    if (s < 192) return {1, 11, (s - 180) % 2, (s - 180) / 2};   // B_11 tile-pair major: output tiles 2b, 2b+1 are complete after slot 181 + 2b

    if (s < 240) return {2, 0, (s - 192) / 2, (s - 192) % 2};
    return {3, 0, s - 240, 0};  // P: fused projection k-step s - 240 (both output tiles)
}
// ROTATED TILE LOOP (round 4).  LayerNorm + store and the fused projection of tile n run under / after the first two layer-1 tiles
// of tile n + 1:  A_0 A_1 (+ LayerNorm of the previous tile in pieces behind their MFMAs) | P (projection of the previous tile) |
// B_0 A_2 | ... | F.  The kernel's slot position ns (weight-pipe stage ns / 8) maps to the schedule slot s above; the weight
// stream keeps the host's order (projection stage last): logical stage 1 of a PROJ stream is blob stage 30.
constexpr int sched_slot(bool proj, int ns) { return !proj ? ns : (ns < 8 ? ns : (ns < 16 ? 240 + (ns - 8) : ns - 8)); }
constexpr int blob_stage(bool proj, int st) { return !proj ? st : (st == 0 ? 0 : (st == 1 ? 30 : st - 1)); }

#define S2S_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ float s2s_one[1] = {1.0f};   // stands in for an absent node mask (read with stride 0)

#if defined(S2S_ET_PROBE) || defined(S2S_EE_PROBE)
// Phase probe (tools/et_phase_probe.py): s_memtime stamps at 16 points of a tile, wave 0 of every workgroup, differences summed
// per workgroup.  The stamps are SMEM results consumed only after the tile's last lgkmcnt(0) wait.
__device__ unsigned long long g_et_probe[512 * 17];
#endif
#ifdef S2S_ET_PROBE
#define ET_STAMP(k) asm volatile("s_memtime %0" : "=s"(st##k))
#else
#define ET_STAMP(k)
#endif

// PROJ: also emit the NEXT IPA block's linear_b / down_z (ipa.py:177,253) of the pair vector just produced -- one more
// weight stage (64 x 128 Wcat, chain-packed), 8 more slots on the LayerNorm output while it is still in registers,
// attention bias written head-major [B,8,N,N], pair_z [B,N,N,32].  Saves the 512 B/pair re-read of z by s2s_pair_project.
template <bool PROJ, bool STAGE>
__global__ void __launch_bounds__(256) edge_transition_f16_kernel(
    const float* __restrict__ edge, const float* __restrict__ node_ab, const float* __restrict__ node_p,
    const char* __restrict__ wblob, const float* __restrict__ b2,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask,
    float* __restrict__ out, long long M, int N, float ln_eps, int io_layout, unsigned mask_stride, const float* __restrict__ proj_b,
    float* __restrict__ proj_bias_out, float* __restrict__ proj_pz_out, int* __restrict__ range_flag, float sk) {
    constexpr int kStages = kStagesBase + (PROJ ? 1 : 0);
    constexpr int kSlots = 8 * kStages;
    // Weight stages in LDS.  PROJ (31 stages, every launch of the sampler): a RING of four 32 KiB buffers, stage x in buffer x & 3, the
    // weight pipe TWO stages ahead (stage x is stored during stage x - 2), and a workgroup barrier only in front of the even stages
    // (+ stage 29, where the odd stage count breaks the period: stages 29 30 | 0 1 of the next tile sit in buffers 1 2 | 0 1) --
    // 17 barriers per tile instead of 31.  A buffer is rewritten two stages after its last read and read two stages after its last
    // write, and with no two consecutive stage boundaries without a barrier there is one in between both times.  (Same-call A/Bs,
    // profiles/r05_et_barrier_ab.txt: every second barrier skipped, results aside, -1.45 % per launch; this ring -0.4 .. -0.8 %.)  Without the projection (30 stages;
    // stand-alone callers): two buffers, one stage ahead, a barrier per stage, as before.
    constexpr int kRing = PROJ ? 4 : 2, kAhead = PROJ ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char s_w[kRing][kStageBytes];
    __shared__ __attribute__((aligned(16))) float s_vec[768 + 64];  // b2 | bf | gamma | beta | projection bias
    // STAGE (chains of 32+ residues): the two rows A_i (+ b1) of node_ab a wave's tile can touch, staged per tile (see stage_rows below)
    __shared__ __attribute__((aligned(16))) float s_rows[STAGE ? 4 * 2 * 384 : 4];
    const int lane = threadIdx.x & 63, h = lane >> 5, wave = threadIdx.x >> 6;

    // ---- weight pipe (see header).  Each wave moves one contiguous 12 KiB of every stage.
    const unsigned voff = wave * 8192 + lane * 16;
    typedef __attribute__((address_space(3))) char lds_char;
    // two base registers (a DS instruction's immediate offset has 16 bits): buffers (0, 1) and (2, 3) of the ring, or the two buffers
    lds_char* lds_image[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[kRing / 2][lane * 16]};
    asm volatile("" : "+v"(lds_image[0]), "+v"(lds_image[1]));  // opaque: every LDS access = base register + immediate
    auto ring_buf = [&](int r) -> lds_char* { return PROJ ? lds_image[r >> 1] + (r & 1) * kStageBytes : lds_image[r]; };   // r: compile-time at every call site
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, kStages * kStageBytes, 0x00020000);
    auto ldw = [&](unsigned vo, int so) -> f32x4 {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, vo, so, 0);
        return f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    };
    // scalars on purpose: an array captured by the lambdas is demoted to memory by hipcc
    f32x4 c0, c1, c2, c3;  // staging group A (first half of the quarter)
    f32x4 e0, e1, e2, e3;  // staging group B (second half)
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    auto cp_load_a = [&](int stage) {
        const int so = blob_stage(PROJ, stage) * kStageBytes;  // compile-time at every call site
        c0 = ldw(voff, so); c1 = ldw(voff + 1024, so); c2 = ldw(voff + 2048, so); c3 = ldw(voff + 3072, so);
    };
    auto cp_load_b = [&](int stage) {
        const int so = blob_stage(PROJ, stage) * kStageBytes + 4096;
        e0 = ldw(voff, so); e1 = ldw(voff + 1024, so); e2 = ldw(voff + 2048, so); e3 = ldw(voff + 3072, so);
    };
    auto cp_store_a = [&](int par) {
        lds_char* d = ring_buf(par) + wave * 8192;
        *(lds_f4*)(d) = c0; *(lds_f4*)(d + 1024) = c1; *(lds_f4*)(d + 2048) = c2; *(lds_f4*)(d + 3072) = c3;
    };
    auto cp_store_b = [&](int par) {
        lds_char* d = ring_buf(par) + (wave * 8192 + 4096);
        *(lds_f4*)(d) = e0; *(lds_f4*)(d + 1024) = e1; *(lds_f4*)(d + 2048) = e2; *(lds_f4*)(d + 3072) = e3;
    };
    // the same, one 1 KiB piece at a time (k = 0..3): inside the tile loop a piece rides behind a single MFMA
    auto cp_load_piece = [&](auto grp, auto kc, int stage) {
        constexpr int k = decltype(kc)::value;
        const int so = blob_stage(PROJ, stage) * kStageBytes + (decltype(grp)::value ? 4096 : 0);
        const f32x4 v = ldw(voff + 1024 * k, so);
        if constexpr (decltype(grp)::value == 0) {
            if constexpr (k == 0) c0 = v; else if constexpr (k == 1) c1 = v; else if constexpr (k == 2) c2 = v; else c3 = v;
        } else {
            if constexpr (k == 0) e0 = v; else if constexpr (k == 1) e1 = v; else if constexpr (k == 2) e2 = v; else e3 = v;
        }
    };
    auto cp_store_piece = [&](auto grp, auto kc, int par) {
        constexpr int k = decltype(kc)::value;
        lds_char* d = ring_buf(par) + (wave * 8192 + (decltype(grp)::value ? 4096 : 0) + 1024 * k);
        if constexpr (decltype(grp)::value == 0)
            *(lds_f4*)(d) = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
        else
            *(lds_f4*)(d) = k == 0 ? e0 : (k == 1 ? e1 : (k == 2 ? e2 : e3));
    };
    cp_load_a(0);
    cp_load_b(0);

    // ---- persistent workgroup: tiles of 128 pairs (32 per wave) blockIdx.x, blockIdx.x + gridDim.x, ...
    // Per-pair context as four 32-bit indices (the launcher keeps 8 M below 2^32): the row pointers are re-formed where they are
    // used (one v_mad_u64_u32 each) instead of living in twelve address registers through the whole tile -- this kernel sits at the
    // 512-register limit, and one more long-lived VGPR costs hundreds of spills.
    struct PairCtx {
        unsigned p, bi, bj, boff;  // flat pair index; flat node rows of i and j; offset of head 0 of this pair in a head-major [B,8,N,N] tensor
        unsigned r0, aoff;         // STAGE: the tile's first node row (wave-uniform), this lane's byte offset of its row's 16 B groups in s_rows
        float em_i, em_j;          // the two node masks, multiplied where the edge mask is used: a product formed here would wait
        bool valid;                // for the loads (vmcnt(0): the whole weight pipe) in the middle of the final layer
    };
    const unsigned NNu = (unsigned)N * (unsigned)N;
    const long long NN = (long long)N * N;
    // division by N through floor(2^32 / N) (quotient short by at most one for x < 2^31; a 64-bit division is ~100 VALU instructions)
    const unsigned n_magic = N >= 2 ? (unsigned)((1ull << 32) / (unsigned)N) : 0u;
    auto div_n = [&](unsigned x, unsigned& q, unsigned& r) {
        q = N >= 2 ? __umulhi(x, n_magic) : x;
        r = x - q * (unsigned)N;
        const bool fix = r >= (unsigned)N;
        q = fix ? q + 1 : q;
        r = fix ? r - (unsigned)N : r;
    };
    auto setup = [&](long long wg_tile) -> PairCtx {
        long long pl = (wg_tile * 4 + wave) * 32 + (lane & 31);
        PairCtx c;
        c.valid = pl < M;
        if (!c.valid) pl = M - 1;  // waves / lanes past the end run on the last pair and store nothing
        const unsigned p = (unsigned)pl;
        // p = (bb N + i) N + j:  flat row bi = p / N,  j = p - bi N,  bb = bi / N
        unsigned bi, j, bb, i;
        div_n(p, bi, j);
        div_n(bi, bb, i);
        c.p = p;
        c.bi = bi;
        c.bj = bb * (unsigned)N + j;
        c.r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)bi);   // consecutive pairs of a chain of 32+ residues: rows r0 and r0 + 1 at most
        c.aoff = (unsigned)wave * 3072u + (bi != c.r0 ? 1536u : 0u) + 16u * (unsigned)h;
        c.boff = p + 7u * bb * NNu;
        // (no branch on the mask's presence: a value merged from two paths is materialised -- and waited for -- at the join;
        //  the launcher passes a one-element "1.0" and stride 0 for an absent mask)
        c.em_i = mask[bi * mask_stride];
        c.em_j = mask[c.bj * mask_stride];
        return c;
    };
    // The 32 pair rows of a wave are one 16 KiB block of z in either layout (header, "Pair-tensor layouts"): row-major, a lane's
    // 16 B group g sits at  n 512 + g 32 + h 16  of the block; tiled, at  g 1024 + lane 16  -- a load / store instruction of the wave
    // then covers 8 whole cache lines instead of 32 B of 32 lines.  (p & 31, not lane & 31: a clamped lane re-reads the last pair.)
    const bool in_tiled = io_layout & 1, out_tiled = io_layout & 2, no_out = io_layout & 4;
    const int in_step = in_tiled ? 256 : 8, out_step = out_tiled ? 256 : 8;   // floats between a lane's consecutive 16 B groups
    auto row_of = [&](const float* base, unsigned p, bool tiled) -> const float* {
        const unsigned blk = tiled ? p >> 5 : p, mul = tiled ? 4096u : 128u;
        const unsigned in_blk = tiled ? ((lane & 32) + (p & 31)) * 4 : 4 * h;
        return base + ((unsigned long long)blk * mul + in_blk);
    };
    auto erow_of = [&](const PairCtx& c) -> const float* { return row_of(edge, c.p, in_tiled); };
    // The pair stream is NON-TEMPORAL (round 6): every edge row is read once and every output row written once per launch (4.3 GB each way at
    // cfg2), and as ordinary accesses they push the 0.97 MB weight stream -- which all 32 workgroups of an XCD re-read every tile -- out of the
    // 4 MiB L2 between two uses (the re-fetches from the Infinity Cache showed in FETCH_SIZE: 378 instead of 301 B per pair).  With the nt bit
    // on the row loads and the output stores: -0.5 .. -1.2 % per launch at every shape measured (same-call A/B, profiles/r06_et_nt_ab.txt;
    // loads alone -0.4 %; nt on the fused projection's stores as well: no further gain, not kept).  -DS2S_ET_NT=0 restores plain accesses.
#ifndef S2S_ET_NT
#define S2S_ET_NT 2
#endif
    typedef float f32x4nt __attribute__((ext_vector_type(4)));
    auto ldrow = [&](const float* r, int g) -> float4 {
        if constexpr (S2S_ET_NT >= 1) {
            const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(r + g * in_step));
            return make_float4(v.x, v.y, v.z, v.w);
        } else {
            return *reinterpret_cast<const float4*>(r + g * in_step);
        }
    };
    const long long n_wt = (M + 127) / 128;
    long long wt = blockIdx.x;
#ifndef S2S_ET_PHASES
#define S2S_ET_PHASES 1
#endif
    // Phase stagger (OFF since the end of round 4; -DS2S_ET_PHASES=16 restores it).  Every workgroup runs the same schedule on equal
    // tiles: left alone they stay in lockstep and ask HBM for their next 64 KiB tile in the same microsecond.  Starting the workgroups
    // of one XCD S2S_ET_PHASES different fractions of a tile apart was meant to spread those requests over the tile time -- but since
    // the edge row is requested two loads per slot, 16+ slots ahead of its use (kXv0 below), lockstep costs nothing, while the start
    // delay (up to 15/16 of a tile: ~45 us) is added to every launch: same-call A/B (profiles/r04t_et_phase_stagger_ab.txt)
    // 11.290 -> 11.274 ms at cfg2, 0.245 -> 0.215 ms at 100 x 35^2 pairs (3.7 tiles per workgroup), 1.019 -> 1.002 at 100 x 80^2.
    // What lockstep does cost is fabric READ REQUESTS, not time: PMC FETCH_SIZE 378 instead of 301 B per pair (1256 / 1104 B per pair
    // of corrected traffic against 1013 algorithmic, profiles/r05d_pmc_hbm_traffic*.json, round 5).  The pair stream flows through
    // the XCD's 4 MiB L2 and evicts the 0.97 MB weight stream between two uses of a stage when all 32 workgroups use it at the same
    // moment once per tile (~43 us); staggered, some workgroup touches every stage every ~3 us and it stays resident.  The re-fetches
    // are served by the 256 MiB Infinity Cache (the counter sits on the L2's fabric side and includes its hits), two stages ahead of
    // their use.  A fine stagger (32 x 0.5 us, one L2 miss apart) changes neither time nor the counter (1245 -> 1225, r05e).
    if constexpr (S2S_ET_PHASES > 1) {
        const int phase = (blockIdx.x >> 3) % S2S_ET_PHASES;
        for (int i = 0; i < phase * (16 / (S2S_ET_PHASES > 16 ? 16 : S2S_ET_PHASES)); ++i) __builtin_amdgcn_s_sleep(98);   // 98 x 64 cycles = 1/16 tile
    }
    PairCtx cur = setup(wt);

    // ---- the pair's 128 edge channels, split once: xpl[ks][plane] = B operand of layer-1 k-step ks.  Element j of k-step
    //      2t+u is channel 32t + 8(2u + (j>>2)) + 4h + (j&3): the accumulator layout of a 128-channel block (register 8u+j of
    //      tile t), so the exact sum of the three planes later serves as the residual row of block 0 without a second read.
    f16x8 xpl[8][2];
    float amax = 0.f;   // range guard (range_flag.h): running maximum of every value that is split into f16 planes
    // Block exponent of the hidden activations (sk = 2^-prescale_exp, 1 by default): the planes of h1 = relu(layer 1) and of
    // relu(layer 2) + x hold sk x the value -- relu is positively homogeneous, so the factor rides in constants that exist anyway
    // (layer 1: (acc + 32 B_j) * (sk / 32) + sk A_i, the caller hands node_ab in as [sk A_i | 32 B_j | 32 sk G_j]; layer 2: relu(acc / 32 + sk b2) + sk x; final
    // layer: the per-node start value 32 sk G_j, which carries the bias b_f) and LayerNorm removes it (eps scaled alike).  Exact for powers of two; sk = 1 gives today's bits.
    const float inv1 = kInvWS * sk;
    // planes (x_h, x_l), see the header
    auto split4 = [&](const float (&x)[4], f16x8& ph, f16x8& pm, int at) {
        split4_f16(x, ph, pm, at, amax);
    };
    {
        float4 xv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xv[i] = ldrow(erow_of(cur), i);  // accumulator ("chain") channel order, see xpl
        for (int i = threadIdx.x; i < 768; i += 256)
            s_vec[i] = i < 384 ? sk * b2[i] : (i < 512 ? 0.f : (i < 640 ? gamma[i - 512] : beta[i - 640]));   // (384..511 unused: bf rides in the per-node start values G_j)
        if (PROJ && threadIdx.x < 64) s_vec[768 + threadIdx.x] = proj_b[threadIdx.x];
        cp_store_a(0);
        cp_load_a(1);
        cp_store_b(0);
        if constexpr (kAhead == 2) {   // the ring starts with stages 0 and 1 in LDS and the first half of stage 2 on its way
            cp_load_b(1);
            cp_store_a(1);
            cp_load_a(2);
            cp_store_b(1);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float x[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
            split4(x, xpl[i >> 1][0], xpl[i >> 1][1], 4 * (i & 1));
        }
    }

    f32x16 a1t[2];     // layer-1 tiles t (even / odd)
    f32x16 a2[12];     // layer-2 accumulators, then relu(.)+residual = the final layer's input
    f32x16 a3[4];
    f32x16 pq[2];      // fused projection accumulators (64 padded output rows)
    float sa[16];  // per-node seeds A_i(+b1) of the a1 tile that is split next (the j-side seeds B_j start the tile's accumulators: seedc_piece)
    // quarter rq of the seeds of tile t (C layout: register 4rq + e = channel 32t + 8rq + 4h + e): A_i + b1 (384 channels) and B_j of
    // the per-node first-layer halves [B,N,768].  (The j-side rows differ from lane to lane -- 16 B in each of 32 rows per load
    // instruction.  Reading them from a column-blocked copy [B][96][N][4], 8 cache lines per instruction as in the edge embedding,
    // was measured: 2 % fewer cycles in the layer-2 blocks, no change in launch time; not worth a second copy of the node vectors.)
    // The row seeds A_i are the same for every pair of a row: 32 consecutive pairs of a chain of 32+ residues touch two rows at most, and
    // a wave used to fetch them with 48 load instructions per tile whose 64 lanes ask for one or two addresses -- the kernel without those
    // loads ran 2.2 % faster (tile 0's seeds for all tiles; timing only).  STAGE: the tile's rows r0, r0 + 1 (2 x 1536 B) go through LDS,
    // three contiguous 1 KiB loads + three LDS stores per wave and tile (stage_load / stage_store: requested right after the next tile's
    // setup, stored five slots later, first read 20+ slots after that; this tile's last read is ~40 slots before the store -- one
    // private region per wave, LDS operations of a wave are in order: no barrier), and a quarter of seeds is one ds_read_b128.
    f32x4 g_st0, g_st1, g_st2;   // staging registers of the row copy
    const unsigned n_node_rows = (unsigned)(M / N);
    auto stage_load = [&](const PairCtx& c) {
        if constexpr (STAGE) {
            const unsigned r1 = c.r0 + 1 < n_node_rows ? c.r0 + 1 : c.r0;
            auto piece = [&](int k) -> f32x4 {
                const unsigned o = (unsigned)k * 1024u + 16u * (unsigned)lane;   // byte o of [row r0 | row r1]
                const bool second = o >= 1536u;
                const float* src = node_ab + (unsigned long long)(second ? r1 : c.r0) * 896u + ((second ? o - 1536u : o) >> 2);
                const float4 v = *reinterpret_cast<const float4*>(src);
                return f32x4{v.x, v.y, v.z, v.w};
            };
            g_st0 = piece(0); g_st1 = piece(1); g_st2 = piece(2);
        }
    };
    auto stage_store = [&]() {
        if constexpr (STAGE) {
            typedef __attribute__((address_space(3))) f32x4 lds_v4;
            lds_v4* d = (lds_v4*)((__attribute__((address_space(3))) char*)s_rows + wave * 3072 + lane * 16);
            d[0] = g_st0; d[64] = g_st1; d[128] = g_st2;
        }
    };
    auto seeds_piece = [&](const PairCtx& c, int t, int rq) {
        if constexpr (STAGE) {
            const f32x4 x = *(const lds_f4*)((__attribute__((address_space(3))) const char*)s_rows + c.aoff + (32 * t + 8 * rq) * 4);
            sa[4 * rq + 0] = x[0]; sa[4 * rq + 1] = x[1]; sa[4 * rq + 2] = x[2]; sa[4 * rq + 3] = x[3];
        } else {
            const float4 x = ldg4(node_ab + (unsigned long long)c.bi * 896u + 32 * t, rq, h);
            sa[4 * rq + 0] = x.x; sa[4 * rq + 1] = x.y; sa[4 * rq + 2] = x.z; sa[4 * rq + 3] = x.w;
        }
    };
    // The j-side seeds are the START VALUE of the layer-1 accumulators (round 5): the caller hands the second half of node_ab in as
    // 32 B_j -- the accumulators' scale -- and quarter rq of tile t is loaded straight into the accumulator registers of a1t[t & 1]
    // (dead from the split of tile t - 2 on; hipcc loads into the accumulator file directly), so that the tile's first product lands
    // on it and the epilogue is relu(acc (sk / 32) + sk A_i): one add per hidden value less (-208 VALU instructions per tile on the
    // ISA, -0.5 % per launch in a same-call A/B).
    auto seedc_piece = [&](const PairCtx& c, int t, int rq) {
        const float4 y = ldg4(node_ab + (unsigned long long)c.bj * 896u + 384 + 32 * t, rq, h);
        a1t[t & 1][4 * rq + 0] = y.x; a1t[t & 1][4 * rq + 1] = y.y; a1t[t & 1][4 * rq + 2] = y.z; a1t[t & 1][4 * rq + 3] = y.w;
    };
    auto seeds_load = [&](const PairCtx& c, int t) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) seeds_piece(c, t, rq);
    };
    stage_load(cur);
    stage_store();
    seeds_load(cur, 0);
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) { seedc_piece(cur, 0, rq); seedc_piece(cur, 1, rq); }

    f16x8 fr[2][4];  // A fragments of the current / next slot: (W_h, W_l) of two (k-step, tile) units
    f16x8 xp[2][2];  // layer 2: planes of k-steps 2t, 2t+1 of the current a1 tile; final layer: current / next k-step
    auto fetch = [&](int par, int slot_in_stage, f16x8 (&f)[4]) {
        typedef __attribute__((address_space(3))) f16x8 lds_frag;
        const lds_frag* s = (const lds_frag*)ring_buf(par) + slot_in_stage * 4 * 64;
#ifndef S2S_ET_FETCH_INORDER
        // fragment 1 (W_l of the first unit) is what a slot's FIRST MFMA reads: requested last, the wait in front of that MFMA covers all
        // four reads (LDS returns in order) and the slot's other MFMAs need none -- one s_waitcnt per slot instead of two or three
        f[0] = s[0]; f[2] = s[128]; f[3] = s[192]; f[1] = s[64];
#else
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = s[64 * k];
#endif
    };
    // quarter qd (registers 4qd..4qd+3) of  relu(a1 tile + seeds)  -> planes of k-step qd>>1, elements 4(qd&1)..
    auto s_quarter = [&](const f32x16& tile_acc, auto qc) {
        constexpr int qd = decltype(qc)::value;
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fmaxf(__builtin_fmaf(tile_acc[4 * qd + j], inv1, sa[4 * qd + j]), 0.f);
        split4(x, xp[qd >> 1][0], xp[qd >> 1][1], 4 * (qd & 1));
    };
    // the same in two halves (values 2hh, 2hh+1 of the quarter): 10 VALU instructions, small enough to sit behind one MFMA
    auto s_half = [&](const f32x16& tile_acc, auto qc, auto hc) {
        constexpr int qd = decltype(qc)::value, j0 = 4 * qd + 2 * decltype(hc)::value;
        const float x0 = fmaxf(__builtin_fmaf(tile_acc[j0], inv1, sa[j0]), 0.f);
        const float x1 = fmaxf(__builtin_fmaf(tile_acc[j0 + 1], inv1, sa[j0 + 1]), 0.f);
        split2_f16(x0, x1, xp[qd >> 1][0], xp[qd >> 1][1], 4 * (qd & 1) + 2 * decltype(hc)::value, amax);
    };
    // residual rows n'_i (block 1) and n'_j (block 2) of  x = [e | n'_i | n'_j]  in accumulator layout, two 16 B groups per call
    float rs[64];
    auto row_load2 = [&](const float* r, float (&dst)[64], int g0) {
#pragma unroll
        for (int g = g0; g < g0 + 2; ++g) {
            const float4 v = ldg4(r, g, h);
            dst[4 * g + 0] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
        }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    S2S_LDS_BARRIER();  // stage 0 and s_vec are in LDS
    fetch(0, 0, fr[0]);

    // ---- LayerNorm(128) over the pair's channels (half here, half in lane ^ 32), + bf, edge mask, store -- of the PREVIOUS tile,
    //      cut into pieces that ride behind the MFMAs of this tile's first eight slots (rotated tile loop: sched_slot above).  The
    //      arithmetic and its order are those of a plain epilogue: bias, running sum and running sum of squared deviations over the
    //      lane's 64 channels in accumulator order, ((a - mean) rstd) gamma + beta, mask.  (LayerNorm is scale invariant: the
    //      statistics run on the 32 x scaled accumulator, with 1024 eps.)
    PairCtx prv = cur;   // the tile whose final-layer accumulators a3 holds; none before the first tile (nothing is stored, and
    prv.valid = false;   // a zero mask keeps the idle pass out of the range maximum)
    prv.em_i = 0.f;
    float ln_sum = 0.f, ln_var = 0.f, ln_mean = 0.f, ln_rstd = 0.f, ln_em = 0.f;
    float* ln_orow = out;
    f16x8 xln[8][2];     // PROJ: planes of the LayerNorm output = B operands of the projection k-steps (chain order)
    auto ln_add = [&](auto qc) {   // piece q = (tile t, quarter rq): running sum (bf is the accumulators' start value, slots 192 / 193)
        constexpr int q = decltype(qc)::value, t = q / 4, rq = q % 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) ln_sum += a3[t][4 * rq + j];
    };
    auto ln_dev = [&](auto qc) {
        constexpr int q = decltype(qc)::value, t = q / 4, rq = q % 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dd = a3[t][4 * rq + j] - ln_mean;
            ln_var += dd * dd;
        }
    };
    // normalise + mask + store [+ PROJ: planes of projection k-step q / 2]; a3 is only read (a value written back into an
    // accumulator tuple drags the whole 16-register tuple into the VALU's half of the register file)
    auto ln_out = [&](auto qc, const float4& ga, const float4& be) {
        constexpr int q = decltype(qc)::value, t = q / 4, rq = q % 4;
        float4 o;
        o.x = ((a3[t][4 * rq + 0] - ln_mean) * ln_rstd * ga.x + be.x) * ln_em;
        o.y = ((a3[t][4 * rq + 1] - ln_mean) * ln_rstd * ga.y + be.y) * ln_em;
        o.z = ((a3[t][4 * rq + 2] - ln_mean) * ln_rstd * ga.z + be.z) * ln_em;
        o.w = ((a3[t][4 * rq + 3] - ln_mean) * ln_rstd * ga.w + be.w) * ln_em;
        if (prv.valid && !no_out) {
            if constexpr (S2S_ET_NT >= 2) __builtin_nontemporal_store(f32x4nt{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4nt*>(ln_orow + q * out_step));
            else *reinterpret_cast<float4*>(ln_orow + q * out_step) = o;
        }
        if constexpr (PROJ) {
            const float x[4] = {o.x, o.y, o.z, o.w};
            split4(x, xln[2 * t + (rq >> 1)][0], xln[2 * t + (rq >> 1)][1], 4 * (rq & 1));
        }
    };
    // PROJ: rows 0..7 (+ bias) -> attention bias, head-major; rows 8..39 -> pair_z channel row - 8 (same map as pair_mlp.hip); group g of 5
    auto proj_store = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (prv.valid) {
            const float4 bq = ldg4(s_vec + 768, g, h);
            if constexpr (g == 0) {
                float* o = proj_bias_out + ((unsigned long long)prv.boff + 4 * h * NN);
                o[0] = __builtin_fmaf(pq[0][0], kInvWS, bq.x);
                o[NN] = __builtin_fmaf(pq[0][1], kInvWS, bq.y);
                o[2 * NN] = __builtin_fmaf(pq[0][2], kInvWS, bq.z);
                o[3 * NN] = __builtin_fmaf(pq[0][3], kInvWS, bq.w);
            } else {
                constexpr int t = g >> 2, rq = g & 3;
                *reinterpret_cast<float4*>(proj_pz_out + (unsigned long long)prv.p * 32u + 8 * (g - 1) + 4 * h) =
                    make_float4(__builtin_fmaf(pq[t][4 * rq + 0], kInvWS, bq.x), __builtin_fmaf(pq[t][4 * rq + 1], kInvWS, bq.y),
                                __builtin_fmaf(pq[t][4 * rq + 2], kInvWS, bq.z), __builtin_fmaf(pq[t][4 * rq + 3], kInvWS, bq.w));
            }
        }
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) a3[t] = zero16;

#ifdef S2S_ET_PROBE
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0, st7 = 0, st8 = 0, st9 = 0, st10 = 0, st11 = 0, st12 = 0,
                       st13 = 0, st14 = 0, st15 = 0;
#endif
    constexpr int kHead = PROJ ? 16 : 15;  // slots that still work for the previous tile: A_0 A_1 + its projection (PROJ) or the first seven slots of B_0
    long long wt_next = wt;
    bool has_next = false;
    PairCtx nxt = cur;  // next tile's context, edge row and planes: produced under the last 16 slots of this tile
    float4 xv[16];
    f16x8 xpn[8][2];
    f16x8 xq[8][2];   // final-layer input planes of k-steps 8..15 (block 1 of the layer-2 epilogue)
    auto slot = [&](auto sc) {
        constexpr int ns = decltype(sc)::value;      // position in the tile loop: weight-pipe stage ns / 8
        constexpr int s = sched_slot(PROJ, ns);      // slot of the schedule (slot_desc)
        constexpr SlotDesc d = slot_desc(s);
        constexpr int stage = ns / 8, ss = ns % 8, par = stage & (kRing - 1);
        constexpr int st_next = (stage + 1) % kStages, st_fill = (stage + kAhead) % kStages;   // the stage read next; the stage whose weights are stored during this one
#ifdef S2S_ET_ALL_BARRIERS   // (debug builds: a barrier in front of every stage -- the reference the ring's barrier placement is stress-tested against, tools/et_ring_stress.py)
        constexpr bool barrier_here = true;
#else
        constexpr bool barrier_here = !PROJ || (st_next & 1) == 0 || st_next == kStages - 2;    // in front of st_next (ring comment at s_w)
#endif
#if defined(S2S_ET_PROBE) && S2S_ET_PROBE == 3   // fine view of one layer-2 block: slot tops 72 .. 87 (B_4 A_6), 88
        if constexpr (s >= 72 && s <= 87) { if constexpr (s == 72) ET_STAMP(0); if constexpr (s == 73) ET_STAMP(1); if constexpr (s == 74) ET_STAMP(2);
            if constexpr (s == 75) ET_STAMP(3); if constexpr (s == 76) ET_STAMP(4); if constexpr (s == 77) ET_STAMP(5); if constexpr (s == 78) ET_STAMP(6);
            if constexpr (s == 79) ET_STAMP(7); if constexpr (s == 80) ET_STAMP(8); if constexpr (s == 81) ET_STAMP(9); if constexpr (s == 82) ET_STAMP(10);
            if constexpr (s == 83) ET_STAMP(11); if constexpr (s == 84) ET_STAMP(12); if constexpr (s == 85) ET_STAMP(13); if constexpr (s == 86) ET_STAMP(14);
            if constexpr (s == 87) ET_STAMP(15); }
#elif defined(S2S_ET_PROBE) && S2S_ET_PROBE == 2   // fine view of the last final-layer block: slot tops 216 .. 239
        if constexpr (s == 216) ET_STAMP(0);
        if constexpr (s == 220) ET_STAMP(1);
        if constexpr (s == 224) ET_STAMP(2);
        if constexpr (s == 225) ET_STAMP(3);
        if constexpr (s == 226) ET_STAMP(4);
        if constexpr (s == 227) ET_STAMP(5);
        if constexpr (s == 228) ET_STAMP(6);
        if constexpr (s == 229) ET_STAMP(7);
        if constexpr (s == 230) ET_STAMP(8);
        if constexpr (s == 232) ET_STAMP(9);
        if constexpr (s == 234) ET_STAMP(10);
        if constexpr (s == 236) ET_STAMP(11);
        if constexpr (s == 237) ET_STAMP(12);
        if constexpr (s == 238) ET_STAMP(13);
        if constexpr (s == 239) ET_STAMP(14);
#else   // phases of one pass of the loop, in time order
        if constexpr (ns == 0) ET_STAMP(0);
        if constexpr (ns == 4) ET_STAMP(1);
        if constexpr (ns == 8) ET_STAMP(2);
        if constexpr (ns == kHead) ET_STAMP(3);
        if constexpr (s == 24) ET_STAMP(4);
        if constexpr (s == 168) ET_STAMP(5);
        if constexpr (s == 180) ET_STAMP(7);
        if constexpr (s == 192) ET_STAMP(9);
        if constexpr (s == 208) ET_STAMP(11);
        if constexpr (s == 224) ET_STAMP(13);
#endif

        // ---------------- top of the slot: next slot's fragments, loads that land under later slots
        if constexpr (ss < 7) {
            fetch(par, ss + 1, fr[(ns + 1) & 1]);
        } else {
            if constexpr (barrier_here) S2S_LDS_BARRIER();
            fetch(st_next & (kRing - 1), 0, fr[(ns + 1) & 1]);
        }
        // next tile: context + edge row under the middle final-layer block, seeds of its tile 0 near the end
        // (the edge row in pieces: a burst of 16 loads per wave holds the slot for ~2 k cycles -- the workgroup's 64 KiB through one
        //  address unit -- and whatever waits next on the in-order counter waits for all of it.  Two loads per slot, in slots whose
        //  next counter wait (the weight store 4+ slots later) is at least 6 slots away: HBM latency fits in between.)
        constexpr int kXv0 = 208;
        if constexpr (s == kXv0) nxt = setup(has_next ? wt_next : wt);
        if constexpr (s == kXv0 + 3) stage_load(nxt);     // (slots ss = 3, 0 of the next stage: none of the edge row's loads)
        if constexpr (s == kXv0 + 8) stage_store();
        if constexpr (s > kXv0 && s <= kXv0 + 16 && ((s & 3) == 1 || (s & 3) == 2)) {
            constexpr int i = 4 * ((s - kXv0) / 4) + 2 * ((s & 3) - 1);   // 2 loads in each of the slots ss = 1, 2, 5, 6 of two stages
            const float* er = erow_of(nxt);
            xv[i] = ldrow(er, i);
            xv[i + 1] = ldrow(er, i + 1);
        }
        if constexpr (s == 236) seeds_load(nxt, 0);
        // ... and the start values of its first two layer-1 tiles (both accumulator tiles are dead from slot 179 on)
        if constexpr (s == 237) { seedc_piece(nxt, 0, 0); seedc_piece(nxt, 0, 1); seedc_piece(nxt, 0, 2); seedc_piece(nxt, 0, 3); }
        if constexpr (s == 238) { seedc_piece(nxt, 1, 0); seedc_piece(nxt, 1, 1); seedc_piece(nxt, 1, 2); seedc_piece(nxt, 1, 3); }
        // seeds of a1 tile t+1 are fetched in the middle of B_t, in a slot without weight-pipe work (consumed under A_{t+2}, 6+ slots
        // later; a fetch 3 slots ahead of its use cost ~300 cycles at the fetch and ~300 at the use: tools/et_phase_probe.py --block)
        constexpr bool seeds_slot = d.phase == 1 && d.a == 1 && d.b == 0 && d.t + 1 < 12;   // a quarter behind each of its first 4 MFMAs
        // residual rows of the layer-2 epilogue blocks 1 (n'_i, under B_11) and 2 (n'_j, under the first final-layer block)
        if constexpr (s >= 184 && s < 192) row_load2(node_p + (unsigned long long)cur.bi * 128u, rs, 2 * (s - 184));
        // bias of this slot's layer-2 epilogue pieces (below): block 0 two pieces per slot under the second half of B_11, blocks 1 and 2 one
        constexpr int ep_blk = (s >= 184 && s < 192) ? 0 : ((s >= 192 && s < 224) ? 1 + (s - 192) / 16 : -1);
        constexpr int ep_q = ep_blk == 0 ? 2 * (s - 184) : (s - 192) % 16;
        float4 ep_b = {0.f, 0.f, 0.f, 0.f}, ep_b1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ep_blk >= 0) ep_b = ldg4(s_vec + 128 * ep_blk, ep_q, h);
        if constexpr (ep_blk == 0) ep_b1 = ldg4(s_vec, ep_q + 1, h);
        // previous tile's LayerNorm (table below): gamma / beta of the two pieces normalised in this slot
        constexpr int ln_k = ns == 7 ? 0 : ((PROJ && d.phase == 3 && d.a < 7) ? d.a + 1 : ((!PROJ && ns >= 8 && ns < 15) ? ns - 7 : -1));   // their projection k-step
        float4 lnv[4] = {};
        if constexpr (ln_k >= 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { lnv[2 * k] = ldg4(s_vec + 512, 2 * ln_k + k, h); lnv[2 * k + 1] = ldg4(s_vec + 640, 2 * ln_k + k, h); }
        }
        // final layer: its accumulators start at G_j = 32 sk (Wf[:, 256:] n'_j + bf), the third column group of node_ab -- the j-side residual
        // of the layer's input x = h2 + [e | n'_i | n'_j] (layers.py:181) taken through the layer per NODE instead of being added to 64
        // hidden values per lane and tile: -64 multiply-adds, -64 registers (the rows n'_j), the bias out of LDS.  Loaded straight into
        // the accumulator registers (dead since the previous tile's LayerNorm), two 16 B pieces per slot in slots 182 .. 189: tile t is
        // complete 4+ slots before its first product (slot 192 / 193) and not live while the layer-2 blocks need every register --
        // same-call A/B against the kernel before (profiles/r05p_et_gseed_ab.txt): four pieces per slot from 184 -0.35 %, from 176 +1.7 %
        // (spills), two per slot from 180 / 182 / 184: -0.67 / -0.70 / -0.45 %.  (Bound: the kernel with that residual simply left out, -1.3 %.)
#ifndef S2S_ET_GSLOT
#define S2S_ET_GSLOT 182
#endif
#ifndef S2S_ET_GPER
#define S2S_ET_GPER 2
#endif
        if constexpr (s >= S2S_ET_GSLOT && s < S2S_ET_GSLOT + 16 / S2S_ET_GPER) {
#pragma unroll
            for (int u = 0; u < S2S_ET_GPER; ++u) {
                constexpr int p0 = (s - S2S_ET_GSLOT) * S2S_ET_GPER;
                const int t3 = (p0 + u) / 4, rq = (p0 + u) % 4;
                const float4 v = ldg4(node_ab + (unsigned long long)cur.bj * 896u + 768 + 32 * t3, rq, h);
                a3[t3][4 * rq + 0] = v.x; a3[t3][4 * rq + 1] = v.y; a3[t3][4 * rq + 2] = v.z; a3[t3][4 * rq + 3] = v.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---------------- the 6 MFMAs, each followed by at most one piece of the weight pipe and one small VALU piece, pinned there
        // (slot-by-slot probe: a slot of bare MFMAs runs at the matrix pipe's 192 cycles; weight-pipe instructions issued as a block
        //  in front of / behind the MFMAs cost 60-120 cycles per slot, a 20-instruction VALU block behind the first MFMA ~100):
        //   weight pipe  group B of stage+1 loaded in slot 0 and stored in slot 5, group A of stage+2 loaded in slot 4 and stored in
        //                slot 1 of the next stage, one 1 KiB piece behind each of MFMAs 0..3; both land in the buffer this stage's
        //                predecessor used, free from that stage's barrier (top of its slot 7) on;
        //   A_t          relu + seeds + split of a1 tile t-1, half a quarter (2 values) behind MFMAs 2 and 4;
        //   layer-2 epilogue  relu(a2 + b2) + x,  x = [e | n'_i | n'_j]  (layers.py:181), in accumulator layout, in 4-value pieces of two
        //                halves: block 0 (its tiles are complete after slot 183: B_11 runs tile-pair major) two pieces per slot under
        //                slots 184..191, in place into the planes the edge row used during layers 1-2 -- whose exact sum x_h + x_l is the
        //                residual row e; block 1 one piece per slot under the final layer's k-steps 0..7 (slots 192..207) into the
        //                planes xq, block 2 under k-steps 8..15 (which read xq) back into xpl;
        //   slots 228..235  split of the next tile's edge row, four halves per slot;
        //   previous tile (positions ns 0..7 = A_0 A_1 of this one, 8..15 = its projection; MFMA i of the slot):
        //                ns 0..3   running sum of piece 4 ns + i behind MFMAs 0..3
        //                ns 4, 5   mean behind MFMA 0 of ns 4; squared deviations of pieces 6 (ns-4) + {0 1 | 2 3 | 4 5} behind MFMAs 1, 3, 5
        //                ns 6      pieces 12 13 | 14 15 behind MFMAs 0, 1; rstd / mask / row pointer behind MFMA 3
        //                normalise + mask + store + (PROJ) split of the two pieces of projection k-step k, behind MFMAs 1 and 3:
        //                k = 0 in ns 7, k = a + 1 in projection slot a (whose MFMAs read k-step a): two k-steps of planes alive at a time
        //                (!PROJ: k = ns - 7 in ns 8..14, the first slots of B_0);
        //                PROJ: the projection's five store groups behind the MFMAs of the first slot after it (ns 16).
        const f16x8 (&f)[4] = fr[ns & 1];
        auto mfma_i = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (d.phase == 0) {
                f32x16& acc = a1t[d.t & 1];
                const f16x8 (&x)[2] = xpl[2 * d.a + (i >= 3)];
                constexpr int fa = (i == 0 ? 1 : (i < 3 ? 0 : (i == 3 ? 3 : 2))), xa = (i == 1 || i == 4) ? 1 : 0;   // W_l x_h, W_h x_l, W_h x_h
                acc = mfma_f16(f[fa], x[xa], acc);   // (a tile's first product lands on its start value 32 B_j: seedc_piece)
            } else {
                constexpr bool fin = d.phase == 2, prj = d.phase == 3;
                constexpr bool first = i < 2 && (prj ? d.a == 0 : (fin ? false : (d.t == 0 && d.a == 0)));   // (final layer: onto 32 bf)
                f32x16& t = prj ? pq[i & 1] : (fin ? a3[2 * d.b + (i & 1)] : a2[2 * d.b + (i & 1)]);
                const f16x8 (&x)[2] = prj ? xln[d.a] : (fin ? ((d.a >> 3) == 1 ? xq[d.a & 7] : xpl[d.a & 7]) : xp[d.a]);
                constexpr int fa = 2 * (i & 1) + (i < 2 ? 1 : 0), xa = (i == 2 || i == 3) ? 1 : 0;                   // W_l x_h, W_h x_l, W_h x_h
                if constexpr (first) t = mfma_f16(f[fa], x[xa], zero16); else t = mfma_f16(f[fa], x[xa], t);
            }
        };
        // (cutting a half once more -- arithmetic behind one MFMA, split behind the next -- measured 0.7 % slower: a VALU instruction costs
        //  its ~5 cycles wherever it sits; what remains to gain is fewer of them)
        auto ep_half = [&](auto qc, auto hc, const float4& bq) {   // values 2hh, 2hh+1 of piece q = (tile t, quarter rq) of block ep_blk
            constexpr int q = decltype(qc)::value, hh = decltype(hc)::value, t = q / 4, rq = q % 4, j0 = 4 * rq + 2 * hh, e0 = 4 * (rq & 1) + 2 * hh;
            const f32x16& a = a2[4 * (ep_blk < 0 ? 0 : ep_blk) + t];
            f16x8 (&P)[2] = ep_blk == 1 ? xq[2 * t + (rq >> 1)] : xpl[2 * t + (rq >> 1)];
            float r0 = 0.f, r1 = 0.f;
            if constexpr (ep_blk == 0) {   // the residual of block 0 is the edge row = x_h + x_l of the planes this piece replaces (to 2^-24 |x|)
#ifndef S2S_ET_NO_MIXRES
                // (float) x_h + (float) x_l as ONE v_fma_mix_f32 per value (x_h * 1.0 + x_l with both read as f16: the same single rounding
                //  as convert, convert, add)
                const unsigned hw = __builtin_bit_cast(u32x4p, P[0])[e0 / 2], lw = __builtin_bit_cast(u32x4p, P[1])[e0 / 2];
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(hw), "v"(lw));
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(hw), "v"(lw));
#else
                r0 = (float)P[0][e0] + (float)P[1][e0];
                r1 = (float)P[0][e0 + 1] + (float)P[1][e0 + 1];
#endif
            } else if constexpr (ep_blk == 1) {
                r0 = rs[16 * t + j0];
                r1 = rs[16 * t + j0 + 1];
            }
            float x0 = fmaxf(__builtin_fmaf(a[j0], kInvWS, hh ? bq.z : bq.x), 0.f), x1 = fmaxf(__builtin_fmaf(a[j0 + 1], kInvWS, hh ? bq.w : bq.y), 0.f);
            if constexpr (ep_blk != 2) {   // block 2's residual n'_j went through the final layer on the node side: G_j, the accumulators' start value
                x0 = __builtin_fmaf(r0, sk, x0);   // (sk = 1: r0 + relu(.), exactly)
                x1 = __builtin_fmaf(r1, sk, x1);
            }
            split2_f16(x0, x1, P[0], P[1], e0, amax);
        };
        static_for<0, 6>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma_i(ic);
            if constexpr (i < 4) {
                if constexpr (ss == 0) cp_load_piece(IC<1>{}, ic, st_fill);   // the pipe wraps into the next tile's first stages
                if constexpr (ss == 4) cp_load_piece(IC<0>{}, ic, (stage + kAhead + 1) % kStages);
#ifndef S2S_ET_STORE_INORDER
                // (the piece loaded LAST is stored first: its wait on the in-order counter covers the group, three waits fewer per slot)
                if constexpr (ss == 1) cp_store_piece(IC<0>{}, IC<3 - i>{}, st_fill & (kRing - 1));
                if constexpr (ss == 5) cp_store_piece(IC<1>{}, IC<3 - i>{}, st_fill & (kRing - 1));
#else
                if constexpr (ss == 1) cp_store_piece(IC<0>{}, ic, st_fill & (kRing - 1));
                if constexpr (ss == 5) cp_store_piece(IC<1>{}, ic, st_fill & (kRing - 1));
#endif
            }
            if constexpr (seeds_slot && i < 4) seeds_piece(cur, d.t + 1, i);
            if constexpr (seeds_slot && i < 4 && d.t + 2 < 12) seedc_piece(cur, d.t + 2, i);   // start value of a1 tile t + 2 (A_{t+2} follows this block)
            if constexpr (d.phase == 0 && d.t >= 1 && (i == 2 || i == 4)) s_half(a1t[(d.t - 1) & 1], IC<d.a>{}, IC<(i - 2) / 2>{});
            if constexpr (ep_blk > 0 && (i == 2 || i == 4)) ep_half(IC<ep_q>{}, IC<(i - 2) / 2>{}, ep_b);
            if constexpr (ep_blk == 0 && i >= 1 && i <= 4) {   // pieces ep_q (halves behind MFMAs 1, 2) and ep_q + 1 (3, 4)
                if constexpr (i <= 2) ep_half(IC<ep_q>{}, IC<i - 1>{}, ep_b); else ep_half(IC<ep_q + 1>{}, IC<i - 3>{}, ep_b1);
            }
            if constexpr (s >= 228 && s < 236 && i >= 1 && i <= 4) {   // next tile's edge row: k-step s - 228, elements 2(i-1), 2(i-1)+1
                constexpr int k = s - 228, e = 2 * (i - 1);
                const float4 v = xv[2 * k + (e >> 2)];
                if constexpr ((e & 2) == 0) split2_f16(v.x, v.y, xpn[k][0], xpn[k][1], e, amax);
                else split2_f16(v.z, v.w, xpn[k][0], xpn[k][1], e, amax);
            }
            // ---- previous tile: LayerNorm pieces (table above)
            if constexpr (ns < 4 && i < 4) ln_add(IC<4 * ns + i>{});
            if constexpr (ns == 4 && i == 0) ln_mean = xhalf_sum(ln_sum) * (1.0f / 128);
            if constexpr ((ns == 4 || ns == 5) && (i == 1 || i == 3 || i == 5)) { ln_dev(IC<6 * (ns - 4) + i - 1>{}); ln_dev(IC<6 * (ns - 4) + i>{}); }
            if constexpr (ns == 6 && (i == 0 || i == 1)) { ln_dev(IC<12 + 2 * i>{}); ln_dev(IC<12 + 2 * i + 1>{}); }
            if constexpr (ns == 6 && i == 3) {
                ln_rstd = 1.0f / sqrtf(xhalf_sum(ln_var) * (1.0f / 128) + ln_eps * ((kWS * sk) * (kWS * sk)));
                ln_em = prv.em_i * prv.em_j;
                ln_orow = const_cast<float*>(row_of(out, prv.p, out_tiled));
                ln_sum = 0.f;
                ln_var = 0.f;
            }
            if constexpr (ln_k >= 0 && (i == 1 || i == 3)) ln_out(IC<2 * ln_k + (i - 1) / 2>{}, lnv[i - 1], lnv[i]);
            if constexpr (PROJ && ns == kHead && i < 5) proj_store(ic);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ---------------- exposed steps
#if defined(S2S_ET_PROBE) && S2S_ET_PROBE == 2
        if constexpr (s == 239) ET_STAMP(15);
#elif defined(S2S_ET_PROBE) && S2S_ET_PROBE == 3
#else
        if constexpr (s == 179) ET_STAMP(6);
        if constexpr (s == 191) ET_STAMP(8);
        if constexpr (s == 207) ET_STAMP(10);
        if constexpr (s == 223) ET_STAMP(12);
        if constexpr (s == 239) ET_STAMP(14);
#endif
        if constexpr (s == 179) {  // B_10 done: tile 11 -> planes (nothing left to hide it under)
            s_quarter(a1t[1], IC<0>{}); s_quarter(a1t[1], IC<1>{}); s_quarter(a1t[1], IC<2>{}); s_quarter(a1t[1], IC<3>{});
        }
    };
    for (;;) {
    wt_next = wt + gridDim.x;
    has_next = wt_next < n_wt;
    static_for<0, kSlots>(slot);

#ifdef S2S_ET_PROBE
#if S2S_ET_PROBE == 1
    ET_STAMP(15);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wave == 0 && lane == 0 && blockIdx.x < 512) {
        unsigned long long* pr = g_et_probe + blockIdx.x * 17;
        const unsigned long long stv[16] = {st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11, st12, st13, st14, st15};
#pragma unroll
        for (int k = 0; k < 15; ++k) atomicAdd(pr + k, stv[k + 1] - stv[k]);
        atomicAdd(pr + 16, 1ull);
    }
#endif
    prv = cur;   // a3 holds this tile's final-layer accumulators: LayerNorm, store and projection in the next pass
#pragma unroll
    for (int i = 0; i < 8; ++i) { xpl[i][0] = xpn[i][0]; xpl[i][1] = xpn[i][1]; }
    // (the next tile's stage 0 sits in buffer 0 again: the ring index is the tile-local stage number)
    if (!has_next) break;
    cur = nxt;
    wt = wt_next;
    }  // persistent tile loop
    // after the last tile: its LayerNorm, store and projection, i.e. the head of one more pass (a second copy of those slots: a branch
    // out of the middle of the loop body costs ~250 spilled registers).  Its layer-1 MFMAs run on the last tile's planes once more
    // (xpn is that tile's row again when there is no next tile) and their results are dropped.
    static_for<0, kHead>(slot);
    if constexpr (PROJ) { proj_store(IC<0>{}); proj_store(IC<1>{}); proj_store(IC<2>{}); proj_store(IC<3>{}); proj_store(IC<4>{}); }
    s2s::range_report(range_flag, amax, s2s::kRangeEdgeTransition);
}


// ------------------------------------------------------------------------------------------------------------------
// Edge embedding on split-f16 MFMA: same operator and contract as s2s_edge_embed (csrc/pair_mlp.hip; reference
// EmbeddingModule.forward edge branch, denoising_ipa.py:137-158).  The first Linear(120 -> 128) is a sum of four gathered
// rows (+ ReLU); the two 128 x 128 layers, LayerNorm and (PROJ) the first IPA block's linear_b / down_z run as f16x3 slots
// exactly like the edge transition above: 5 weight stages of 32 KiB (W2 | W3 | [linear_b; down_z]), 40 slots of 6 MFMAs per
// 32-pair tile, persistent workgroups; the NEXT tile's rows are gathered and split under the current tile's MFMAs.
template <bool PROJ>
__global__ void __launch_bounds__(256) edge_embed_f16_kernel(
    const float* __restrict__ node_a, const float* __restrict__ node_b, const float* __restrict__ rel_tab,
    const float* __restrict__ bin_tab, const float* __restrict__ bin_lower, const long long* __restrict__ residue_idx,
    const float* __restrict__ ca, const char* __restrict__ wblob, const float* __restrict__ b2, const float* __restrict__ b3,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, float* __restrict__ out,
    long long M, int N, int rel_off, int n_rel, int n_bins, float ln_eps, int out_tiled, const float* __restrict__ proj_b,
    float* __restrict__ proj_bias_out, float* __restrict__ proj_pz_out, int* __restrict__ range_flag) {
    constexpr int kStages = PROJ ? 5 : 4;
    constexpr int kSlots = 8 * kStages;
    __shared__ __attribute__((aligned(16))) char s_w[2][kStageBytes];
    __shared__ __attribute__((aligned(16))) float s_vec[512 + 64];  // b2 | b3 | gamma | beta | projection bias
    __shared__ __attribute__((aligned(16))) float s_bins[64 + 4];    // distogram bin lower edges (ascending), padded with 3e38
    const int lane = threadIdx.x & 63, h = lane >> 5, wave = threadIdx.x >> 6;

    // ---- weight pipe (identical to the edge transition's)
    const unsigned voff = wave * 8192 + lane * 16;
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* lds_image[2] = {(lds_char*)&s_w[0][lane * 16], (lds_char*)&s_w[1][lane * 16]};
    asm volatile("" : "+v"(lds_image[0]), "+v"(lds_image[1]));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, kStages * kStageBytes, 0x00020000);
    auto ldw = [&](unsigned vo, int so) -> f32x4 {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, vo, so, 0);
        return f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    };
    f32x4 c0, c1, c2, c3, e0, e1, e2, e3;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    auto cp_load_a = [&](int stage) {
        const int so = stage * kStageBytes;
        c0 = ldw(voff, so); c1 = ldw(voff + 1024, so); c2 = ldw(voff + 2048, so); c3 = ldw(voff + 3072, so);
    };
    auto cp_load_b = [&](int stage) {
        const int so = stage * kStageBytes + 4096;
        e0 = ldw(voff, so); e1 = ldw(voff + 1024, so); e2 = ldw(voff + 2048, so); e3 = ldw(voff + 3072, so);
    };
    auto cp_store_a = [&](int par) {
        lds_char* d = lds_image[par] + wave * 8192;   // (last-loaded piece first: its counter wait covers the group)
        *(lds_f4*)(d + 3072) = c3; *(lds_f4*)(d + 2048) = c2; *(lds_f4*)(d + 1024) = c1; *(lds_f4*)(d) = c0;
    };
    auto cp_store_b = [&](int par) {
        lds_char* d = lds_image[par] + (wave * 8192 + 4096);
        *(lds_f4*)(d + 3072) = e3; *(lds_f4*)(d + 2048) = e2; *(lds_f4*)(d + 1024) = e1; *(lds_f4*)(d) = e0;
    };
    cp_load_a(0);
    cp_load_b(0);

    // ---- per-tile context: the four first-layer rows of this lane's pair
    // node_b / rel_tab / bin_tab come COLUMN-BLOCKED: [32 chunks of 4 channels][rows][4], so that the 16 B a lane wants of
    // its pair's row sit next to the neighbouring pairs' (consecutive j -> consecutive rows): one load instruction touches
    // 8 cache lines instead of 32.  Lane part of the address in voffset, chunk pair (2G, 2G+1) in the scalar offset.
    struct Ctx {
        const float* ra;
        unsigned vb, vr, vk;   // byte offsets of (row, chunk h) in node_b / rel_tab / bin_tab
        float kb;
        long long p, boff;
        float em;
        bool valid;
    };
    const long long NN = (long long)N * N;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)node_b, 0, (unsigned)(M / N * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rel_tab, 0, n_rel * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)bin_tab, 0, n_bins * 512, 0x00020000);
    // setup in two halves so that the per-pair loads (CA coordinates, residue indices, masks) are in flight for a few slots
    // before they are consumed; the distogram edges sit in LDS (s_bins)
    struct Raw {
        long long p, bi, bj, bb;
        float ax, ay, az, bx, by, bz;
        long long ii, ij;
        float mi, mj;
        bool valid;
    };
    // The launcher keeps the pair count of one launch below 2^31: 32-bit index arithmetic, division by N through
    // floor(2^32 / N) (quotient short by at most one for x < 2^31; a 64-bit division is ~100 VALU instructions).
    const unsigned n_magic = N >= 2 ? (unsigned)((1ull << 32) / (unsigned)N) : 0u;
    auto div_n = [&](unsigned x, unsigned& q, unsigned& r) {
        q = N >= 2 ? __umulhi(x, n_magic) : x;
        r = x - q * (unsigned)N;
        const bool fix = r >= (unsigned)N;
        q = fix ? q + 1 : q;
        r = fix ? r - (unsigned)N : r;
    };
    const float* mask_or_any = mask ? mask : ca;  // always a readable [B N] float array: no branch around the mask loads
    auto setup_a = [&](long long wg_tile) -> Raw {
        long long p = (wg_tile * 4 + wave) * 32 + (lane & 31);
        Raw r;
        r.valid = p < M;
        p = r.valid ? p : M - 1;
        r.p = p;
        // p = (bb N + i) N + j:  global row bi = p / N,  j = p - bi N,  bb = bi / N
        unsigned bi, j, bb, i;
        div_n((unsigned)p, bi, j);
        div_n(bi, bb, i);
        r.bi = bi; r.bb = bb; r.bj = bb * (unsigned)N + j;
        r.ax = ca[r.bi * 3 + 0]; r.ay = ca[r.bi * 3 + 1]; r.az = ca[r.bi * 3 + 2];
        r.bx = ca[r.bj * 3 + 0]; r.by = ca[r.bj * 3 + 1]; r.bz = ca[r.bj * 3 + 2];
        r.ii = residue_idx[r.bi];
        r.ij = residue_idx[r.bj];
        r.mi = mask_or_any[r.bi];
        r.mj = mask_or_any[r.bj];
        return r;
    };
    auto setup_b = [&](const Raw& r) -> Ctx {
        Ctx c;
        c.valid = r.valid;
        // distogram bin of this pair (no FMA contraction: mirrors torch.linalg.norm of the difference; geo_utils.py:44-56)
        const float dx = r.ax - r.bx, dy = r.ay - r.by, dz = r.az - r.bz;
        const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        // the reference's one-hot (dist > lower[k]) * (dist < upper[k]), upper = lower[k+1] (1e8 for the last): with ascending
        // edges (torch.linspace) that is k = #{edges < dist} - 1 unless dist sits exactly on the next edge (then no bin at all)
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // s_bins: n_bins <= 32 edges | 1e8 (the last bin's upper edge) | 3e38 ...
            const float4 e = *reinterpret_cast<const float4*>(&s_bins[4 * u]);
            cnt += (e.x < dist) + (e.y < dist) + (e.z < dist) + (e.w < dist);
        }
        cnt = cnt > n_bins ? n_bins : cnt;                  // dist beyond 1e8 when n_bins < 32: no bin, as below
        const float up = s_bins[cnt];                       // upper edge of bin cnt - 1
        const int bin = (cnt >= 1 && dist < up) ? cnt - 1 : -1;
        long long d = r.ii - r.ij + rel_off;
        d = d < 0 ? 0 : (d >= n_rel ? n_rel - 1 : d);
        c.ra = node_a + r.bi * 128;
        const unsigned jj = (unsigned)(r.bj - r.bb * N);
        c.vb = ((unsigned)r.bb * 32u * (unsigned)N + (unsigned)h * (unsigned)N + jj) * 16u;
        c.vr = ((unsigned)h * (unsigned)n_rel + (unsigned)d) * 16u;
        c.vk = ((unsigned)h * (unsigned)n_bins + (unsigned)(bin < 0 ? 0 : bin)) * 16u;
        c.kb = bin < 0 ? 0.f : 1.f;
        c.p = r.p;
        c.boff = r.p + 7 * r.bb * NN;
        c.em = mask ? r.mi * r.mj : 1.0f;
        return c;
    };
    float amax = 0.f;   // range guard (range_flag.h)
    auto split4 = [&](const float (&x)[4], f16x8& ph, f16x8& pm, int at) {  // planes (x_h, x_l)
        split4_f16(x, ph, pm, at, amax);
    };
    // first-layer sum in accumulator layout: g1[4G + q] = channel 8G + 4h + q, built row by row (same association as the
    // fp32 kernel: ((a + b) + r) + kb*k)
    float g1[64];
    auto row_set = [&](const float* r) {
#pragma unroll
        for (int G = 0; G < 16; ++G) {
            const float4 v = ldg4(r, G, h);
            g1[4 * G + 0] = v.x; g1[4 * G + 1] = v.y; g1[4 * G + 2] = v.z; g1[4 * G + 3] = v.w;
        }
    };
    float tmp[64];
    auto row_cb = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, int rows, float (&dst)[64], int G0, int G1) {
#pragma unroll
        for (int G = G0; G < G1; ++G) {  // chunk 2G + h of the row: scalar offset 2G rows-blocks of 16 B
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, G * 32 * rows, 0);
            dst[4 * G + 0] = __uint_as_float(v.x); dst[4 * G + 1] = __uint_as_float(v.y);
            dst[4 * G + 2] = __uint_as_float(v.z); dst[4 * G + 3] = __uint_as_float(v.w);
        }
    };
    auto row_add = [&](float scale) {  // pinned: left alone, the compiler sinks these adds to the splits 20 slots later and keeps
#pragma unroll                         // (spills) all four loaded rows until then
        for (int i = 63; i >= 0; --i) {   // (last-loaded values first: one counter wait for the whole row instead of one per load)
            g1[i] = __fmaf_rn(scale, tmp[i], g1[i]);  // (explicit: both template variants must round alike)
            asm volatile("" : "+v"(g1[i]));
        }
    };
    // ReLU + split of first-layer k-step ks (registers 8ks .. 8ks+7 of g1: chain order) into planes
    auto g1_split = [&](f16x8 (&dst)[2], int ks) {
        const float x0[4] = {relu1(g1[8 * ks + 0]), relu1(g1[8 * ks + 1]), relu1(g1[8 * ks + 2]), relu1(g1[8 * ks + 3])};
        const float x1[4] = {relu1(g1[8 * ks + 4]), relu1(g1[8 * ks + 5]), relu1(g1[8 * ks + 6]), relu1(g1[8 * ks + 7])};
        split4(x0, dst[0], dst[1], 0);
        split4(x1, dst[0], dst[1], 4);
    };

    const long long n_wt = (M + 127) / 128;
    long long wt = blockIdx.x;
    if (threadIdx.x < 68) s_bins[threadIdx.x] = (int)threadIdx.x < n_bins ? bin_lower[threadIdx.x] : ((int)threadIdx.x == n_bins ? 1e8f : 3.0e38f);
    __syncthreads();
    Ctx cur = setup_b(setup_a(wt));
    f16x8 xp[8][2];  // planes of the current layer's input (8 k-steps of 16)
    {
        row_set(cur.ra);
        for (int i = threadIdx.x; i < 512; i += 256)
            s_vec[i] = i < 128 ? kWS * b2[i] : (i < 256 ? kWS * b3[i - 128] : (i < 384 ? gamma[i - 256] : beta[i - 384]));   // biases x 32 (accumulator scale)
        if (PROJ && threadIdx.x < 64) s_vec[512 + threadIdx.x] = proj_b[threadIdx.x];
        cp_store_a(0);
        cp_load_a(1);
        cp_store_b(0);
        row_cb(rs_b, cur.vb, N, tmp, 0, 16); row_add(1.0f);
        row_cb(rs_r, cur.vr, n_rel, tmp, 0, 16); row_add(1.0f);
        row_cb(rs_k, cur.vk, n_bins, tmp, 0, 16); row_add(cur.kb);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) g1_split(xp[ks], ks);
    }
    f32x16 a2[4], a3[4], pq[2];
    f16x8 fr[2][4];
    auto fetch = [&](int par, int slot_in_stage, f16x8 (&f)[4]) {
        typedef __attribute__((address_space(3))) f16x8 lds_frag;
        const lds_frag* s = (const lds_frag*)lds_image[par] + slot_in_stage * 4 * 64;
        f[0] = s[0]; f[2] = s[128]; f[3] = s[192]; f[1] = s[64];   // (the slot's first MFMA reads f[1]: one counter wait per slot, as in the edge transition)
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The VALU work of a tile is cut into per-k-step pieces, each pinned (empty asm on its inputs / outputs: otherwise the
    // compiler sinks a piece to its first use, i.e. in FRONT of the MFMAs that wait for it) into a slot whose MFMAs do not depend
    // on it, and interleaved with them by sched_group_barrier.
    auto pin_frag = [&](f16x8 (&x)[2]) { asm volatile("" : "+v"(x[0]), "+v"(x[1])); };
    // accumulator start = bias (registers 4 rq + q of tile t <-> channel 32 t + 8 rq + 4 h + q)
    auto bias16 = [&](const float* vec, int t) -> f32x16 {   // (s_vec holds 32 x bias: the accumulators carry 32 x the layer output)
        const float4 b0 = ldg4(vec, 4 * t, h), b1 = ldg4(vec, 4 * t + 1, h), b2v = ldg4(vec, 4 * t + 2, h), b3v = ldg4(vec, 4 * t + 3, h);
        return f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2v.x, b2v.y, b2v.z, b2v.w, b3v.x, b3v.y, b3v.z, b3v.w};
    };
    // layer-2 output (bias already in the accumulator): ReLU + split of k-step k -> layer-3 input planes xp[k]
    auto l2_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value, t = k / 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rq = 2 * (k & 1) + u;
            const float xx[4] = {relu1(a2[t][4 * rq + 0]) * kInvWS, relu1(a2[t][4 * rq + 1]) * kInvWS,
                                 relu1(a2[t][4 * rq + 2]) * kInvWS, relu1(a2[t][4 * rq + 3]) * kInvWS};
            split4(xx, xp[k][0], xp[k][1], 4 * u);
        }
        pin_frag(xp[k]);
    };
    float ln_mean = 0.f, ln_rstd = 0.f;
    // output block of the wave (16 KiB): row-major  n 512 + g 32 + h 16,  tiled  g 1024 + lane 16  (edge transition, "Pair-tensor layouts")
    const unsigned out_lane = out_tiled ? lane * 16u : (unsigned)((lane & 31) * 512 + h * 16), out_step = out_tiled ? 1024u : 32u;
    f16x8 xq[2][2];  // LayerNorm output planes of the projection's current / next k-step
    // LayerNorm output of k-step k (16 channels): scale, shift, edge mask, store (+ planes for the projection)
    __amdgpu_buffer_rsrc_t rs_out;  // this tile's 32 output rows; rows past M are outside num_records: their stores are dropped
    auto out_rsrc = [&](long long wg_tile) {
        const long long p0 = (wg_tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * 32;
        const long long left = M - p0;
        // (tiled layout: the pairs of a partial last tile are interleaved with its padding, which the buffer holds: whole block)
        rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (p0 < M ? p0 : 0) * 128), 0,
                                                   (unsigned)(left <= 0 ? 0 : (left < 32 && !out_tiled ? left : 32)) * 512u, 0x00020000);
    };
    float4 lga[2], lbe[2];  // gamma / beta of the next LayerNorm piece, read at the top of its slot
    auto ln_load = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = 4 * (k / 2) + 2 * (k & 1) + u;
            lga[u] = ldg4(s_vec + 256, g, h);
            lbe[u] = ldg4(s_vec + 384, g, h);
        }
    };
    auto ln_piece = [&](auto kc, const Ctx& c) {
        constexpr int k = decltype(kc)::value, t = k / 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rq = 2 * (k & 1) + u, g = 4 * t + rq;
            const float4 ga = lga[u], be = lbe[u];
            float4 o;
            // (the channel offset goes into the store's immediate field, not into an SGPR soffset: a 128-bit buffer store with an
            // SGPR offset followed by VALU writes of its data registers lost the last lanes' data in the plain variant)
            // explicit roundings: the fused-projection and the plain variant of this kernel must agree bit for bit
            o.x = __fmul_rn(__fmaf_rn(__fmul_rn(a3[t][4 * rq + 0] - ln_mean, ln_rstd), ga.x, be.x), c.em);
            o.y = __fmul_rn(__fmaf_rn(__fmul_rn(a3[t][4 * rq + 1] - ln_mean, ln_rstd), ga.y, be.y), c.em);
            o.z = __fmul_rn(__fmaf_rn(__fmul_rn(a3[t][4 * rq + 2] - ln_mean, ln_rstd), ga.z, be.z), c.em);
            o.w = __fmul_rn(__fmaf_rn(__fmul_rn(a3[t][4 * rq + 3] - ln_mean, ln_rstd), ga.w, be.w), c.em);
#ifndef S2S_EE_NT
#define S2S_EE_NT 0
#endif
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)},
                                                   rs_out, out_lane + (unsigned)g * out_step, 0, S2S_EE_NT ? 2 : 0);   // (aux bit 1 = nt)
            if constexpr (PROJ) {
                const float xx[4] = {o.x, o.y, o.z, o.w};
                split4(xx, xq[k & 1][0], xq[k & 1][1], 4 * u);
            }
        }
        if constexpr (PROJ) pin_frag(xq[k & 1]);
    };
    // rows of the next tile's first layer, eight 16 B loads per slot (a burst of 32 per wave backs up the vector memory
    // pipe and with it the wave's MFMA issue)
    auto row_load8 = [&](const float* r, float (&dst)[64], int half) {
#pragma unroll
        for (int G = 8 * half; G < 8 * half + 8; ++G) {
            const float4 v = ldg4(r, G, h);
            dst[4 * G + 0] = v.x; dst[4 * G + 1] = v.y; dst[4 * G + 2] = v.z; dst[4 * G + 3] = v.w;
        }
    };
    S2S_LDS_BARRIER();
    fetch(0, 0, fr[0]);
    f32x16 initA0 = bias16(s_vec, 0), initA1 = bias16(s_vec, 1), initB0, initB1;
    // per-pair scalars (CA coordinates, residue indices, masks) of the tile after this one: requested a whole tile ahead -- used four
    // slots after the request they cost the tile ~1.2 k cycles of waiting (tools/ee_phase_probe.py, slot 4)
    Raw nraw = setup_a(wt + gridDim.x < n_wt ? wt + gridDim.x : wt);

    for (;;) {
    const long long wt_next = wt + gridDim.x;
    const bool has_next = wt_next < n_wt;
    Ctx nxt = cur;
    f16x8 xl[2];  // the next tile's last k-step (xp[7] is read by the final layer's last slot)
#ifdef S2S_EE_PROBE
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0, st7 = 0, st8 = 0, st9 = 0, st10 = 0, st11 = 0, st12 = 0,
                       st13 = 0, st14 = 0, st15 = 0;
#define EE_STAMP(k) asm volatile("s_memtime %0" : "=s"(st##k))
#else
#define EE_STAMP(k)
#endif
    static_for<0, kSlots>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int stage = s / 8, ss = s % 8, par = stage & 1;
#if defined(S2S_EE_PROBE) && S2S_EE_PROBE == 2   // slot by slot: tops of slots BASE .. BASE + 14
#ifndef S2S_EE_PROBE_BASE
#define S2S_EE_PROBE_BASE 0
#endif
        if constexpr (s >= S2S_EE_PROBE_BASE && s < S2S_EE_PROBE_BASE + 15) {
            constexpr int k = s - S2S_EE_PROBE_BASE;
            if constexpr (k == 0) EE_STAMP(0); if constexpr (k == 1) EE_STAMP(1); if constexpr (k == 2) EE_STAMP(2); if constexpr (k == 3) EE_STAMP(3);
            if constexpr (k == 4) EE_STAMP(4); if constexpr (k == 5) EE_STAMP(5); if constexpr (k == 6) EE_STAMP(6); if constexpr (k == 7) EE_STAMP(7);
            if constexpr (k == 8) EE_STAMP(8); if constexpr (k == 9) EE_STAMP(9); if constexpr (k == 10) EE_STAMP(10); if constexpr (k == 11) EE_STAMP(11);
            if constexpr (k == 12) EE_STAMP(12); if constexpr (k == 13) EE_STAMP(13); if constexpr (k == 14) EE_STAMP(14);
        }
#else
        if constexpr (s == 0) EE_STAMP(0);
        if constexpr (s == 4) EE_STAMP(1);
        if constexpr (s == 8) EE_STAMP(2);
        if constexpr (s == 12) EE_STAMP(3);
        if constexpr (s == 16) EE_STAMP(4);
        if constexpr (s == 20) EE_STAMP(5);
        if constexpr (s == 24) EE_STAMP(6);
        if constexpr (s == 28) EE_STAMP(7);
        if constexpr (s == 31) EE_STAMP(8);
        if constexpr (s == 32) EE_STAMP(10);
        if constexpr (s == 36) EE_STAMP(11);
        if constexpr (s == 39) EE_STAMP(12);
#endif
        constexpr int layer = s / 16;             // 0: layer 2, 1: layer 3, 2: projection
        constexpr int ks = layer < 2 ? (s % 16) / 2 : s - 32;
        constexpr int pr = layer < 2 ? s % 2 : 0;
        // ---------------- top of the slot: LDS / global loads whose results are used under later MFMAs
        // the accumulators of a layer's first two slots start from the bias, read one slot ahead (A: slots 0 / 16, B: 1 / 17)
        if constexpr (s == kSlots - 1) { initA0 = bias16(s_vec, 0); initA1 = bias16(s_vec, 1); }
        if constexpr (s == 0) { initB0 = bias16(s_vec, 2); initB1 = bias16(s_vec, 3); }
        if constexpr (s == 15) { initA0 = bias16(s_vec + 128, 0); initA1 = bias16(s_vec + 128, 1); }
        if constexpr (s == 16) { initB0 = bias16(s_vec + 128, 2); initB1 = bias16(s_vec + 128, 3); }
        if constexpr (PROJ && s >= 31 && s < 39) ln_load(IC<s - 31>{});
        if constexpr (ss < 7) {
            fetch(par, ss + 1, fr[(s + 1) & 1]);
        } else {
            S2S_LDS_BARRIER();
            fetch(par ^ 1, 0, fr[(s + 1) & 1]);
        }
        // next tile: per-pair loads under slot 0, context under slot 4, then its four first-layer rows
        if constexpr (s == 30) out_rsrc(wt);
        if constexpr (s == 5) row_load8(nxt.ra, g1, 0);
        if constexpr (s == 6) row_load8(nxt.ra, g1, 1);
        if constexpr (s == 7) row_cb(rs_b, nxt.vb, N, tmp, 0, 8);
        if constexpr (s == 8) row_cb(rs_b, nxt.vb, N, tmp, 8, 16);
        if constexpr (s == 13) row_cb(rs_r, nxt.vr, n_rel, tmp, 0, 8);
        if constexpr (s == 14) row_cb(rs_r, nxt.vr, n_rel, tmp, 8, 16);
        if constexpr (s == 18) row_cb(rs_k, nxt.vk, n_bins, tmp, 0, 8);
        if constexpr (s == 19) row_cb(rs_k, nxt.vk, n_bins, tmp, 8, 16);
        __builtin_amdgcn_sched_barrier(0);

        // ---------------- the 12 MFMAs and the VALU pieces that run under them
        const f16x8 (&f)[4] = fr[s & 1];
        const f16x8 (&x)[2] = layer < 2 ? xp[ks] : xq[ks & 1];
        f32x16& t0 = layer == 0 ? a2[2 * pr] : (layer == 1 ? a3[2 * pr] : pq[0]);
        f32x16& t1 = layer == 0 ? a2[2 * pr + 1] : (layer == 1 ? a3[2 * pr + 1] : pq[1]);
        if constexpr (ks == 0) {
            if constexpr (layer < 2) {
                t0 = mfma_f16(f[1], x[0], pr == 0 ? initA0 : initB0); t1 = mfma_f16(f[3], x[0], pr == 0 ? initA1 : initB1);
            } else {
                t0 = mfma_f16(f[1], x[0], zero16); t1 = mfma_f16(f[3], x[0], zero16);
            }
        } else {
            t0 = mfma_f16(f[1], x[0], t0); t1 = mfma_f16(f[3], x[0], t1);  // W_l x_h
        }
        t0 = mfma_f16(f[0], x[1], t0); t1 = mfma_f16(f[2], x[1], t1);      // W_h x_l
        t0 = mfma_f16(f[0], x[0], t0); t1 = mfma_f16(f[2], x[0], t1);      // W_h x_h
        if constexpr (s == 1) nxt = setup_b(nraw);                                                        // the next tile's context
        if constexpr (s == 26) nraw = setup_a(wt_next + gridDim.x < n_wt ? wt_next + gridDim.x : wt_next);   // requests for the tile after it
        if constexpr (s == 12) row_add(1.0f);     // a + b   (same association as the fp32 kernel: ((a + b) + r) + kb k)
        if constexpr (s == 17) row_add(1.0f);     // + relative-position row
        if constexpr (s == 22) row_add(nxt.kb);   // + distogram row
        // layer-2 output, k-step k (read by slots 16 + 2k, 17 + 2k) under slot 14 + 2k, k-step 0 under slot 15
        if constexpr (s >= 16 && s <= 28 && s % 2 == 0) l2_piece(IC<(s - 14) / 2>{});
        if constexpr (s == 15) l2_piece(IC<0>{});   // tiles 0, 1 of the layer-2 output are complete after slot 14
        // next tile's first layer: k-step k of xp is last read by slot 17 + 2k, so k = 0..6 go under slots 24..30 and the
        // last one (under slot 23) into xl
        if constexpr (s >= 24 && s < 31) { g1_split(xp[s - 24], s - 24); pin_frag(xp[s - 24]); }
        if constexpr (s == 23) { g1_split(xl, 7); pin_frag(xl); }
        // LayerNorm output k-step k + 1 under projection slot 32 + k
        if constexpr (PROJ && s >= 32 && s < 39) ln_piece(IC<s - 31>{}, cur);
        // weight pipe: one load / LDS store behind each of the first four MFMAs (edge transition, same slots)
        if constexpr (ss == 0) cp_load_b((stage + 1) % kStages);
        if constexpr (ss == 4) cp_load_a((stage + 2) % kStages);
        if constexpr (ss == 1) cp_store_a(par ^ 1);
        if constexpr (ss == 5) cp_store_b(par ^ 1);
        constexpr int copy_kind = (ss == 0 || ss == 4) ? 0x020 : ((ss == 1 || ss == 5) ? 0x200 : 0);   // VMEM read | DS write | none
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA,
            if (copy_kind != 0 && i < 4) __builtin_amdgcn_sched_group_barrier(copy_kind, 1, 0);   // one piece of the weight pipe,
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // up to eight VALU instructions behind it
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---------------- exposed steps
#if !(defined(S2S_EE_PROBE) && S2S_EE_PROBE == 2)
        if constexpr (s == 31) EE_STAMP(9);
#endif
        if constexpr (s == 31) {  // layer-3 output (bias included): LayerNorm statistics
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += a3[t][r];
            ln_mean = __fmul_rn(xhalf_sum(sum), 1.0f / 128);
            float var = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dd = a3[t][r] - ln_mean;
                    var = __fmaf_rn(dd, dd, var);
                }
            ln_rstd = 1.0f / sqrtf(__fmaf_rn(xhalf_sum(var), 1.0f / 128, ln_eps * (kWS * kWS)));   // (scaled accumulator: 1024 eps)
            if constexpr (PROJ) {
                ln_piece(IC<0>{}, cur);
            } else {
                static_for<0, 8>([&](auto kc) { ln_load(kc); ln_piece(kc, cur); });
            }
        }
    });
#if !(defined(S2S_EE_PROBE) && S2S_EE_PROBE == 2)
    EE_STAMP(13);
#endif
    if constexpr (PROJ) {
        if (cur.valid) {
            const float4 b0 = ldg4(s_vec + 512, 0, h);
            float* o = proj_bias_out + cur.boff + 4 * h * NN;
            o[0] = __builtin_fmaf(pq[0][0], kInvWS, b0.x);
            o[NN] = __builtin_fmaf(pq[0][1], kInvWS, b0.y);
            o[2 * NN] = __builtin_fmaf(pq[0][2], kInvWS, b0.z);
            o[3 * NN] = __builtin_fmaf(pq[0][3], kInvWS, b0.w);
#pragma unroll
            for (int g = 1; g <= 4; ++g) {
                const int t = g >> 2, rq = g & 3;
                const float4 bq = ldg4(s_vec + 512, g, h);
                *reinterpret_cast<float4*>(proj_pz_out + cur.p * 32 + 8 * (g - 1) + 4 * h) =
                    make_float4(__builtin_fmaf(pq[t][4 * rq + 0], kInvWS, bq.x), __builtin_fmaf(pq[t][4 * rq + 1], kInvWS, bq.y),
                                __builtin_fmaf(pq[t][4 * rq + 2], kInvWS, bq.z), __builtin_fmaf(pq[t][4 * rq + 3], kInvWS, bq.w));
            }
        }
    }
#ifdef S2S_EE_PROBE
#if S2S_EE_PROBE != 2
    EE_STAMP(14);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PROJ && wave == 0 && lane == 0 && blockIdx.x < 512) {
        unsigned long long* pr = g_et_probe + blockIdx.x * 17;
        const unsigned long long stv[16] = {st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11, st12, st13, st14, st14};
#pragma unroll
        for (int k = 0; k < 14; ++k) atomicAdd(pr + k, stv[k + 1] - stv[k]);
        atomicAdd(pr + 16, 1ull);
    }
#endif
    if (!has_next) break;
    cur = nxt;
    wt = wt_next;
    xp[7][0] = xl[0]; xp[7][1] = xl[1];
    if constexpr (kStages % 2 == 1) {  // odd stage count: the next tile's stage 0 sits in the other buffer
        lds_char* sw = lds_image[0];
        lds_image[0] = lds_image[1];
        lds_image[1] = sw;
    }
    }  // persistent tile loop
    s2s::range_report(range_flag, amax, s2s::kRangeEdgeEmbed);
}


// smallest number of samples whose pairs fill whole 32-pair blocks (launch boundaries of a tiled pair tensor)
long long tile_aligned_samples(long long NN) {
    long long q = 32;
    while (q > 1 && (NN * (32 / q)) % 32 != 0) q /= 2;   // 32 / q samples suffice when NN (32 / q) is a multiple of 32
    return 32 / q;
}

template <bool STAGE>
int et_launch(const float* edge, const float* node_ab, const float* node_p, const void* weight_stream, const float* b2,
              const float* ln_gamma, const float* ln_beta, const float* mask, float* out, int n_samples, int n_res, float ln_eps,
              int io_layout, const float* proj_bias_cat64, float* proj_attn_bias, float* proj_pair_z, int prescale_exp, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if ((io_layout & ~7) || ((io_layout & 4) && !proj_attn_bias) || (!(io_layout & 4) && !out)) return (int)hipErrorInvalidValue;
    if (prescale_exp < 0 || prescale_exp > 15) return (int)hipErrorInvalidValue;
    const float sk = ldexpf(1.0f, -prescale_exp);
    const long long NN = (long long)n_res * n_res;
    // 32-bit pair / head-major indices inside a launch (8 M < 2^32): split the samples over several launches when needed
    const char* cap_env = getenv("S2S_ET_MAX_PAIRS");   // test hook: a smaller per-launch pair budget exercises the split
    long long cap = cap_env ? atoll(cap_env) : 0;
    if (cap <= 0 || cap > (1ll << 29) - 1) cap = (1ll << 29) - 1;
    long long chunk = cap / NN;
    if ((io_layout & 3) && chunk < n_samples) chunk -= chunk % tile_aligned_samples(NN);   // launches of a tiled tensor start on a 32-pair block
    if (chunk < 1) return (int)hipErrorInvalidValue;
    static const float* one_of[64] = {};   // per device: the address of s2s_one
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return (int)hipErrorInvalidValue;
    if (!one_of[dev_id] && hipGetSymbolAddress((void**)&one_of[dev_id], HIP_SYMBOL(s2s_one)) != hipSuccess) return (int)hipErrorInvalidValue;
    const float* one = one_of[dev_id];
    static int n_cu = 0;  // persistent workgroups, one per CU
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    for (long long b0 = 0; b0 < n_samples; b0 += chunk) {
        const long long nb = n_samples - b0 < chunk ? n_samples - b0 : chunk;
        const long long M = nb * NN, rows0 = b0 * n_res;
        const long long wg_tiles = (M + 127) / 128;
        const long long grid = wg_tiles < n_cu ? wg_tiles : n_cu;
        const float* e = edge + b0 * NN * 128;
        const float* nab = node_ab + rows0 * 896;
        const float* np = node_p + rows0 * 128;
        const float* mk = mask ? mask + rows0 : one;
        const unsigned mks = mask ? 1u : 0u;
        float* o = out ? out + b0 * NN * 128 : nullptr;
        auto k_proj = &edge_transition_f16_kernel<true, STAGE>;
        auto k_plain = &edge_transition_f16_kernel<false, STAGE>;
        if (proj_attn_bias)
            hipLaunchKernelGGL(k_proj, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, e, nab, np,
                               (const char*)weight_stream, b2, ln_gamma, ln_beta, mk, o, M, n_res, ln_eps, io_layout, mks, proj_bias_cat64,
                               proj_attn_bias + b0 * 8 * NN, proj_pair_z + b0 * NN * 32, s2s::g_range_flag, sk);
        else
            hipLaunchKernelGGL(k_plain, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, e, nab, np,
                               (const char*)weight_stream, b2, ln_gamma, ln_beta, mk, o, M, n_res, ln_eps, io_layout, mks,
                               (const float*)nullptr, (float*)nullptr, (float*)nullptr, s2s::g_range_flag, sk);
    }
    return (int)hipGetLastError();
}

}  // namespace

#define S2S_ET_ARGS const float* edge, const float* node_ab, const float* node_p, const void* weight_stream, const float* b2, \
                    const float* ln_gamma, const float* ln_beta, const float* mask, float* out, int n_samples, int n_res, float ln_eps, \
                    int io_layout, const float* proj_bias_cat64, float* proj_attn_bias, float* proj_pair_z, int prescale_exp, void* stream
#define S2S_ET_PASS edge, node_ab, node_p, weight_stream, b2, ln_gamma, ln_beta, mask, out, n_samples, n_res, ln_eps, io_layout, \
                    proj_bias_cat64, proj_attn_bias, proj_pair_z, prescale_exp, stream
// chains below 32 residues: a 32-pair tile spans more than two rows, the row seeds stay per-lane loads (its own translation unit)
extern "C" __attribute__((visibility("hidden"))) int s2s_et_f16x3_short_chains(S2S_ET_ARGS);   // (inside the library only)
#if S2S_PM_PART == 2
extern "C" int s2s_et_f16x3_short_chains(S2S_ET_ARGS) { return et_launch<false>(S2S_ET_PASS); }
#endif
#if S2S_PM_PART == 1
extern "C" int s2s_edge_transition_f16x3(S2S_ET_ARGS) {
    return n_res >= 32 ? et_launch<true>(S2S_ET_PASS) : s2s_et_f16x3_short_chains(S2S_ET_PASS);
}
#endif

#if (defined(S2S_ET_PROBE) && S2S_PM_PART == 1) || (defined(S2S_EE_PROBE) && S2S_PM_PART == 3)   // (a probe build replaces the one unit its kernel lives in)
extern "C" int s2s_et_probe_read(unsigned long long* host_out, int reset) {   // 512 x 17 counters
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_et_probe), sizeof(unsigned long long) * 512 * 17);
    if (e == hipSuccess && reset) {
        static unsigned long long z[512 * 17];
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_et_probe), z, sizeof(z));
    }
    return (int)e;
}
#endif

#if S2S_PM_PART == 3
extern "C" int s2s_edge_embed_f16x3(const float* node_a, const float* node_b, const float* rel_table, const float* bin_table,
                                     const float* bin_lower, const long long* residue_idx, const float* ca_xyz,
                                     const void* weight_stream, const float* b2, const float* b3, const float* ln_gamma,
                                     const float* ln_beta, const float* mask, float* out, int n_samples, int n_res,
                                     int rel_offset, int n_rel, int n_bins, float ln_eps, int out_tiled, const float* proj_bias_cat64,
                                     float* proj_attn_bias, float* proj_pair_z, void* stream) {
    if (n_samples <= 0 || n_res <= 0) return 0;
    if (n_bins > 32 || (out_tiled & ~1)) return (int)hipErrorInvalidValue;  // the distogram edges are counted from a 32-entry LDS table
    // 32-bit pair indices and buffer offsets inside a launch: split the samples over several launches when needed
    const long long NN = (long long)n_res * n_res;
    if (NN >= (1ll << 31) || (long long)n_rel * 512 >= (1ll << 32)) return (int)hipErrorInvalidValue;
    const char* cap_env = getenv("S2S_EE_MAX_PAIRS");   // test hook: a smaller per-launch pair budget exercises the split
    long long cap = cap_env ? atoll(cap_env) : 0;
    if (cap <= 0 || cap > (1ll << 31) - 1) cap = (1ll << 31) - 1;
    long long chunk = cap / NN;
    const long long rows_cap = ((1ll << 32) - 1) / ((long long)n_res * 512);  // node_b descriptor
    if (rows_cap < chunk) chunk = rows_cap;
    if (out_tiled && chunk < n_samples) chunk -= chunk % tile_aligned_samples(NN);
    if (chunk < 1) return (int)hipErrorInvalidValue;
    int n_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
        n_cu = 256;
    for (long long b0 = 0; b0 < n_samples; b0 += chunk) {
        const long long nb = n_samples - b0 < chunk ? n_samples - b0 : chunk;
        const long long M = nb * NN, rows0 = b0 * n_res;
        const long long wg_tiles = (M + 127) / 128;
        const long long grid = wg_tiles < n_cu ? wg_tiles : n_cu;
        const float* na = node_a + rows0 * 128;
        const float* nbp = node_b + rows0 * 128;
        const long long* ridx = residue_idx + rows0;
        const float* cap = ca_xyz + rows0 * 3;
        const float* mk = mask ? mask + rows0 : nullptr;
        float* o = out + b0 * NN * 128;
        if (proj_attn_bias)
            hipLaunchKernelGGL(edge_embed_f16_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, na, nbp,
                               rel_table, bin_table, bin_lower, ridx, cap, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta,
                               mk, o, M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, proj_bias_cat64,
                               proj_attn_bias + b0 * 8 * NN, proj_pair_z + b0 * NN * 32, s2s::g_range_flag);
        else
            hipLaunchKernelGGL(edge_embed_f16_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, na, nbp,
                               rel_table, bin_table, bin_lower, ridx, cap, (const char*)weight_stream, b2, b3, ln_gamma, ln_beta,
                               mk, o, M, n_res, rel_offset, n_rel, n_bins, ln_eps, out_tiled, (const float*)nullptr, (float*)nullptr,
                               (float*)nullptr, s2s::g_range_flag);
    }
    return (int)hipGetLastError();
}
#endif   // S2S_PM_PART == 3
