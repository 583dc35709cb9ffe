// Pieces shared by the attention kernels (attn.hip: first-generation forward / two-phase backward; attn2.hip: the small-workgroup
// forward of the benchmark head geometry).
#pragma once
#include "common.h"

size_t swv2_attn_bias_range_offset(int heads, int L);       // attn.hip: the (max, min) part of a packed CPB table

namespace {

template <int LT, int DK>
struct AttnCfg {
    static constexpr int Lp = 16 * LT;
    static constexpr int DP = 16 * DK;
    static constexpr int NT = 64 * LT;          // threads per workgroup
    static constexpr int SLAB = Lp * DP;        // elements of one [Lp][DP] slab
};

// scaled / biased / masked scores of one query column (swapped layout: lane = query, acc[t][r] = key 16t + 4g + r), in
// place, log2 domain; returns the lane's partial maximum.  MASKED is the (wave-uniform) shift-mask case, instantiated
// separately so the common unmasked windows carry no select instructions; LFIX > 0 is a compile-time window area so the
// padded-key test folds away everywhere except in the last tile (the first build spent ~40 % of the forward kernel's
// instructions on these two tests).
template <int LT, bool HAS_BIAS, bool MASKED, int LFIX>
__device__ __forceinline__ float score_pass(f32x4 (&acc)[LT], const uint32_t (&biasp)[LT][2], float sc2, int Lrt, int g,
                                            int mask_thr, bool qid) {
    const int L = LFIX > 0 ? LFIX : Lrt;
    float mx = SWV2_NEG_BIG;
#pragma unroll
    for (int t = 0; t < LT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + 4 * g + r;
            float s;
            if (HAS_BIAS) {
                const uint32_t w = biasp[t][r >> 1];             // padded keys carry -1e30 in the bias row
                s = fmaf(acc[t][r], sc2, __uint_as_float((r & 1) ? (w & 0xffff0000u) : (w << 16)));
            } else {
                s = acc[t][r] * sc2;
                if (LFIX > 0 ? (16 * t + 16 > LFIX) : (16 * t + 16 > L)) s = (key < L) ? s : SWV2_NEG_BIG;
            }
            if (MASKED) s += ((key >= mask_thr) != qid) ? (-100.f * SWV2_LOG2E) : 0.f;
            acc[t][r] = s;
            mx = fmaxf(mx, s);
        }
    }
    return mx;
}

}  // namespace
