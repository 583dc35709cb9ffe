// Continuous position bias of SwinV2 (reference swinv2_global.py:240-261, 274-287): the 2 -> Hd -> heads meta MLP over the
// L^2 log-spaced relative coordinates of a window, forward and backward, as two small kernels instead of ~10 PyTorch
// passes over [L^2, Hd] fp32 tensors (measured 341 us per block in torch, forward + backward).
//   R[p]        = sign(delta) * log(1 + |delta|),  delta = (r_q - r_k, c_q - c_k),  p = t_q * L + t_k      (:251-261)
//   hidden[p,j] = relu(w1[j,0] R0 + w1[j,1] R1 + b1[j]) * keep[p,j] / (1 - drop)     (Dropout(0.125), train only, :245)
//   bias[h,p]   = sum_j w2[h,j] hidden[p,j] + b2[h]                                   (:282-286: [L^2,h]^T -> [h,L,L])
// The dropout keep-mask is drawn by the
// caller with the torch RNG (shape [L^2, Hd], any non-zero = keep) so that the stochastic draw stays a host-side torch op.
#include "common.h"

namespace {

constexpr int CPB_MAX_HEADS = 32;
typedef __attribute__((address_space(1))) float gfloat;
// pairs per workgroup of the backward kernels: ONE constant for the workspace size and the launch (measured at the 9x18 window:
// 64 -> 34 us, 128 -> 33 us, 256 -> 46 us)
constexpr int CPB_PPB = 128;

__device__ __forceinline__ void rel_coord(int p, int L, int ww, float& r0, float& r1) {
    const int tq = p / L, tk = p - tq * L;
    const int dr = tq / ww - tk / ww, dc = tq % ww - tk % ww;
    const float a = (float)dr, b = (float)dc;
    r0 = copysignf(log1pf(fabsf(a)), a) * (dr != 0);
    r1 = copysignf(log1pf(fabsf(b)), b) * (dc != 0);
}

// Forward: lane = pair, the 4 waves of a workgroup each take a quarter of the hidden units (uniform index -> the weights
// come through the scalar cache as SGPR operands), partial sums are combined through LDS.  No barrier inside the loop.
template <int HEADS_MAX>
__global__ __launch_bounds__(256) void cpb_fwd_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      const uint16_t* __restrict__ keep, float* __restrict__ bias, int L,
                                                      int ww, int heads, int Hd, float scale) {
    __shared__ float red[4][HEADS_MAX][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L2 = L * L;
    const int p = min(blockIdx.x * 64 + lane, L2 - 1);
    float r0, r1;
    rel_coord(p, L, ww, r0, r1);
    const int jq = (Hd + 3) >> 2, j0 = wave * jq, j1 = min(Hd, j0 + jq);
    float acc[HEADS_MAX];
#pragma unroll
    for (int h = 0; h < HEADS_MAX; ++h) acc[h] = 0.f;
    const uint16_t* krow = keep ? keep + (size_t)p * Hd : nullptr;
    int j = j0;
    if ((Hd & 7) == 0 && (jq & 7) == 0) {        // 8 hidden units per trip: one 16-byte mask load, batched scalar loads
        for (; j + 8 <= j1; j += 8) {
            uint4 kq = make_uint4(0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u);
            if (krow) kq = *reinterpret_cast<const uint4*>(krow + j);
            const uint32_t kw[4] = {kq.x, kq.y, kq.z, kq.w};
            float hd8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float hv = fmaxf(fmaf(w1[2 * (j + u)], r0, fmaf(w1[2 * (j + u) + 1], r1, b1[j + u])), 0.f);
                const uint32_t kbits = (kw[u >> 1] >> (16 * (u & 1))) & 0xffffu;
                hd8[u] = krow ? (kbits != 0 ? hv * scale : 0.f) : hv;
            }
#pragma unroll
            for (int h = 0; h < HEADS_MAX; ++h) {      // rows past `heads` re-read the last head; their sums are dropped
                const float* w2h = w2 + min(h, heads - 1) * Hd + j;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[h] = fmaf(w2h[u], hd8[u], acc[h]);
            }
        }
    }
    for (; j < j1; ++j) {
        float hdn = fmaxf(fmaf(w1[2 * j], r0, fmaf(w1[2 * j + 1], r1, b1[j])), 0.f);
        if (krow) hdn = krow[j] != 0 ? hdn * scale : 0.f;
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) acc[h] = fmaf(w2[min(h, heads - 1) * Hd + j], hdn, acc[h]);
    }
#pragma unroll
    for (int h = 0; h < HEADS_MAX; ++h) red[wave][h][lane] = acc[h];
    __syncthreads();
    for (int i = threadIdx.x; i < heads * 64; i += 256) {
        const int h = i >> 6, l = i & 63, pp = blockIdx.x * 64 + l;
        if (pp < L2) bias[(size_t)h * L2 + pp] = red[0][h][l] + red[1][h][l] + red[2][h][l] + red[3][h][l] + b2[h];
    }
}

// Backward: thread = hidden unit, so every weight gradient is a private register sum over the workgroup's pairs; the
// pair-uniform data (d bias of the heads, the two coordinates) is staged in LDS 64 pairs at a time and the keep-mask of
// 8 pairs is fetched ahead of its use.
template <int HEADS_MAX>
__global__ __launch_bounds__(512) void cpb_bwd_kernel(const float* __restrict__ dbias, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const uint16_t* __restrict__ keep, float* __restrict__ dw1,
                                                      float* __restrict__ db1, float* __restrict__ dw2,
                                                      float* __restrict__ db2, int L, int ww, int heads, int Hd,
                                                      float scale, int pairs_per_block, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float dbs[64][HEADS_MAX + 4];   // [pair][d bias of head 0.., r0, r1]
    const int j = threadIdx.x;
    const bool act = j < Hd;
    const int jc = act ? j : 0;
    const float wa = act ? w1[2 * j] : 0.f, wb = act ? w1[2 * j + 1] : 0.f, bb = act ? b1[j] : 0.f;
    float w2r[HEADS_MAX], g2[HEADS_MAX];
#pragma unroll
    for (int h = 0; h < HEADS_MAX; ++h) { w2r[h] = (act && h < heads) ? w2[h * Hd + j] : 0.f; g2[h] = 0.f; }
    float ga = 0.f, gb = 0.f, gbias = 0.f, gb2 = 0.f;
    const int L2 = L * L;
    const int p0 = blockIdx.x * pairs_per_block, p1 = min(L2, p0 + pairs_per_block);
    for (int pc = p0; pc < p1; pc += 64) {
        uint32_t kbits[2] = {~0u, ~0u};        // the chunk's keep flags of this hidden unit, in flight during the staging
        if (keep) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                uint32_t b = 0;
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    b |= (uint32_t)(keep[(size_t)min(pc + 32 * w + u, L2 - 1) * Hd + jc] != 0) << u;
                kbits[w] = b;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * HEADS_MAX; i += blockDim.x) {
            const int h = i >> 6, pp = i & 63;
            dbs[pp][h] = (h < heads && pc + pp < p1) ? dbias[(size_t)h * L2 + pc + pp] : 0.f;
        }
        if (threadIdx.x < 64) {
            float r0, r1;
            rel_coord(min(pc + (int)threadIdx.x, L2 - 1), L, ww, r0, r1);
            dbs[threadIdx.x][HEADS_MAX] = r0;
            dbs[threadIdx.x][HEADS_MAX + 1] = r1;
        }
        __syncthreads();
#pragma unroll 4
        for (int pp = 0; pp < 64; ++pp) {          // rows past p1 hold d bias = 0 and contribute nothing
            const float r0 = dbs[pp][HEADS_MAX], r1 = dbs[pp][HEADS_MAX + 1];
            const float pre = fmaf(wa, r0, fmaf(wb, r1, bb));
            const float m = keep ? (((kbits[pp >> 5] >> (pp & 31)) & 1u) ? scale : 0.f) : 1.f;
            const float hdn = fmaxf(pre, 0.f) * m;
            float dh = 0.f;
#pragma unroll
            for (int h = 0; h < HEADS_MAX; ++h) {      // heads past `heads` carry zeros (no per-head branch)
                const float d = dbs[pp][h];
                g2[h] = fmaf(d, hdn, g2[h]);
                dh = fmaf(d, w2r[h], dh);
            }
            dh = (pre > 0.f) ? dh * m : 0.f;
            ga = fmaf(dh, r0, ga);
            gb = fmaf(dh, r1, gb);
            gbias += dh;
        }
        if (j < heads)
            for (int pp = 0; pp < 64; ++pp) gb2 += dbs[pp][j];
    }
    if (part) {
        // one partial row per workgroup, laid out like the four gradients one after the other: [dw1 (2 Hd) | db1 (Hd) | dw2 (heads Hd) |
        // db2 (heads)]; cpb_fold_kernel adds the rows in a fixed order (no atomics: the gradient is bit-reproducible, and 205 workgroups
        // no longer queue on the same 4 232 addresses)
        float* row = part + (size_t)blockIdx.x * (3 * Hd + heads * Hd + heads);
        if (act) {
            row[2 * j] = ga;
            row[2 * j + 1] = gb;
            row[2 * Hd + j] = gbias;
#pragma unroll
            for (int h = 0; h < HEADS_MAX; ++h)
                if (h < heads) row[3 * Hd + h * Hd + j] = g2[h];
        }
        if (j < heads) row[3 * Hd + heads * Hd + j] = gb2;
        return;
    }
    if (act) {
        atomicAdd(dw1 + 2 * j, ga);
        atomicAdd(dw1 + 2 * j + 1, gb);
        atomicAdd(db1 + j, gbias);
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h)
            if (h < heads) atomicAdd(dw2 + h * Hd + j, g2[h]);
    }
    if (j < heads) atomicAdd(db2 + j, gb2);
}

// out[i] += sum over the workgroups' partial rows (fixed order); 64 entries x 8 row groups per workgroup, four loads in flight per thread
__global__ __launch_bounds__(512) void cpb_fold_kernel(const float* __restrict__ part, int rows, int n, float* __restrict__ dw1,
                                                        float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                        int Hd, int heads) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int r = rg;
        for (; r + 24 < rows; r += 32) {
            a0 += part[(size_t)r * n + i];
            a1 += part[(size_t)(r + 8) * n + i];
            a2 += part[(size_t)(r + 16) * n + i];
            a3 += part[(size_t)(r + 24) * n + i];
        }
        for (; r < rows; r += 8) a0 += part[(size_t)r * n + i];
    }
    red[rg][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && i < n) {
        const float t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + ((red[4][lane] + red[5][lane]) + (red[6][lane] + red[7][lane]));
        float* dst = i < 2 * Hd ? dw1 + i : i < 3 * Hd ? db1 + (i - 2 * Hd) : i < 3 * Hd + heads * Hd ? dw2 + (i - 3 * Hd) : db2 + (i - 3 * Hd - heads * Hd);
        *dst += t;
    }
}


// ------------------------------------------------------------------------------------------------
// All blocks of a stage in ONE launch (round 5).  Nothing in the CPB pipeline depends on activations, so the model computes the
// `depth` tables before block 0 (swv2_cpb_fwd_multi) and their parameter gradients after block 0's backward
// (swv2_cpb_bwd_multi) instead of 2 x depth x (draw + kernel + pack / fold / reduce) launches inside the blocks.
//   ptab       : device table [nblk][4] of the blocks' parameter pointers (w1 [Hd][2], b1 [Hd], w2 [heads][Hd], b2 [heads])
//   keep_words : u32 [nblk][L^2][Hd / 8] of uniformly random bits drawn by the caller (ONE torch `random_()` launch), see below;
//                kept units are scaled by 1 / (1 - 0.125).  NULL: eval mode.
// ------------------------------------------------------------------------------------------------
// ---- keep words (both multi kernels): hidden unit j of a pair <-> bit (j & 7) of dec(W) = W | W >> 8 | W >> 16, W = word widx(j) of
// the pair's Hd / 8 words, widx(j) = ((j >> 3) & 3) * (Hd / 32) + (j >> 5): kept iff the bit is set, i.e. dropped iff the bit is clear in
// all three low BYTES of the random word -- probability (1/2)^3 = 1/8, the reference's hard-coded Dropout(0.125) (:245).  (torch's
// `random_()` leaves bit 31 of an int32 clear: only the three low bytes are used.)  The word order makes the 12 words a lane of the
// forward needs -- units 32 ks + 8 g .. + 7 for every k-step ks -- contiguous.
__device__ __forceinline__ uint32_t keep_dec(uint32_t w) { return w | (w >> 8) | (w >> 16); }
__device__ __forceinline__ int keep_widx(int j, int KS) { return ((j >> 3) & 3) * KS + (j >> 5); }

// x = hi + lo with hi = the bf16 truncation of x (exact), lo = bf16(x - hi): the pair carries x to ~2^-17 relative, so three bf16 MFMAs
// (hi hi + lo hi + hi lo) give the fp32 product sums of the VALU formulation to ~1e-5
__device__ __forceinline__ void split_hi_lo8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = __float_as_uint(x[2 * i]) & 0xffff0000u, b = __float_as_uint(x[2 * i + 1]) & 0xffff0000u;
        h[i] = (a >> 16) | b;
        l[i] = f2bf2(x[2 * i] - __uint_as_float(a), x[2 * i + 1] - __uint_as_float(b));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void split_hi_lo4(const float (&x)[4], bf16x4& hi, bf16x4& lo) {
    uint32_t h[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t a = __float_as_uint(x[2 * i]) & 0xffff0000u, b = __float_as_uint(x[2 * i + 1]) & 0xffff0000u;
        h[i] = (a >> 16) | b;
        l[i] = f2bf2(x[2 * i] - __uint_as_float(a), x[2 * i + 1] - __uint_as_float(b));
    }
    hi = __builtin_bit_cast(bf16x4, h);
    lo = __builtin_bit_cast(bf16x4, l);
}

constexpr int CPB_MAX_KS = 16;           // hidden <= 512

// Forward of all blocks on the matrix pipe (round 5, second form).  The VALU form above (lane = pair, uniform weights from the
// scalar cache) issued ~16 vector instructions per (pair, hidden unit) -- 8 of them the heads' FMAs, 5 the 3-bit keep test -- and
// took 88 - 109 us per step at depth 12 (52 us without the mask).  Here the heads' contraction is  bias^T = hidden . W2^T  on
// v_mfma_f32_16x16x32_bf16 with the hidden activation as A operand straight from the registers it is computed in (lane (pair, g) holds
// units 32 ks + 8 g .. + 7 of its pair: exactly A[i][k]), split hi + lo so the sums keep fp32 accuracy (three MFMAs per k-step), and
// the keep test is one bit-field extract + AND per value.  W2's B fragments (hi, lo) are built once per workgroup in LDS.
template <int TP, int KI>          // TP 16-pair tiles per wave; KI k-steps per keep-word load group (4: Hd % 128 == 0, else 2)
__global__ __launch_bounds__(256) void cpb_fwd_mfma_kernel(const float* const* __restrict__ ptab, const uint32_t* __restrict__ keep_words,
                                                           float* __restrict__ bias_all, int L, int ww, int heads, int Hd, float scale) {
    __shared__ __attribute__((aligned(16))) uint16_t w2f[CPB_MAX_KS * 2 * 64 * 8];      // [ks][hi | lo][lane][8]
    __shared__ __attribute__((aligned(16))) float w1t[CPB_MAX_KS * 32 * 4];             // [unit][scale w1[.,0], scale w1[.,1], scale b1, 0]
    const int blk = blockIdx.y, KS = Hd >> 5;
    const gfloat* __restrict__ w1 = (const gfloat*)ptab[blk * 4 + 0];
    const gfloat* __restrict__ b1 = (const gfloat*)ptab[blk * 4 + 1];
    const gfloat* __restrict__ w2 = (const gfloat*)ptab[blk * 4 + 2];
    const gfloat* __restrict__ b2 = (const gfloat*)ptab[blk * 4 + 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int L2 = L * L;
    // relu(s x) = s relu(x) for the scale s = 1 / (1 - p) > 0: folded into the first layer
    for (int j = tid; j < Hd; j += 256) {
        w1t[4 * j] = w1[2 * j] * scale; w1t[4 * j + 1] = w1[2 * j + 1] * scale; w1t[4 * j + 2] = b1[j] * scale; w1t[4 * j + 3] = 0.f;
    }
    for (int i = tid; i < KS * 64; i += 256) {
        const int ks = i >> 6, l_ = i & 63, fr_ = l_ & 15, g_ = l_ >> 4;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fr_ < heads ? w2[fr_ * Hd + 32 * ks + 8 * g_ + e] : 0.f;
        bf16x8 hi, lo;
        split_hi_lo8(x, hi, lo);
        *(bf16x8*)(w2f + ((size_t)(ks * 2 + 0) * 64 + l_) * 8) = hi;
        *(bf16x8*)(w2f + ((size_t)(ks * 2 + 1) * 64 + l_) * 8) = lo;
    }
    __syncthreads();
    const int pb = blockIdx.x * (4 * TP * 16) + wave * (TP * 16);
    float r0[TP], r1[TP];
    // the lane's keep words: words g KS .. g KS + KS - 1 of its pair are contiguous -- KI words per load, the next group's loads issued
    // before this group's arithmetic (a 4-byte load per (k-step, tile) waited for on the spot cost 56 of 98 us)
    typedef uint32_t kwv_t __attribute__((ext_vector_type(KI)));
    const uint32_t* ksrc[TP];
    kwv_t knext[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        const int p = min(pb + 16 * t + fr, L2 - 1);
        rel_coord(p, L, ww, r0[t], r1[t]);
        ksrc[t] = keep_words ? keep_words + ((size_t)blk * L2 + p) * (Hd >> 3) + g * KS : nullptr;
        if (keep_words) knext[t] = *(const kwv_t*)ksrc[t];
    }
    f32x4 acc[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k0 = 0; k0 < KS; k0 += KI) {
        kwv_t kcur[TP];
#pragma unroll
        for (int t = 0; t < TP; ++t) kcur[t] = knext[t];
        if (keep_words && k0 + KI < KS) {
#pragma unroll
            for (int t = 0; t < TP; ++t) knext[t] = *(const kwv_t*)(ksrc[t] + k0 + KI);
        }
#pragma unroll
        for (int kk = 0; kk < KI; ++kk) {
            const int ks = k0 + kk;
            f32x4 wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = *(const f32x4*)(w1t + 4 * (32 * ks + 8 * g + e));
            const bf16x8 Bhi = *(const bf16x8*)(w2f + ((size_t)(ks * 2 + 0) * 64 + lane) * 8);
            const bf16x8 Blo = *(const bf16x8*)(w2f + ((size_t)(ks * 2 + 1) * 64 + lane) * 8);
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(fmaf(wv[e][0], r0[t], fmaf(wv[e][1], r1[t], wv[e][2])), 0.f);
                if (keep_words) {
                    const int dec = (int)keep_dec(kcur[t][kk]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = __uint_as_float(__float_as_uint(x[e]) & (uint32_t)__builtin_amdgcn_sbfe(dec, e, 1));
                }
                bf16x8 Ahi, Alo;
                split_hi_lo8(x, Ahi, Alo);
                acc[t] = mfma32(Ahi, Bhi, acc[t]);
                acc[t] = mfma32(Alo, Bhi, acc[t]);
                acc[t] = mfma32(Ahi, Blo, acc[t]);
            }
        }
    }
    if (fr < heads) {
        const float bb = b2[fr];
        float* dst = bias_all + ((size_t)blk * heads + fr) * L2;
#pragma unroll
        for (int t = 0; t < TP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = pb + 16 * t + 4 * g + r;
                if (p < L2) dst[p] = acc[t][r] + bb;
            }
    }
}

// Backward of all blocks: d bias of block b = the sum of `nchunk` tables [heads][L^2] (the per-workgroup tables the attention
// backward leaves, summed here while they are staged -- no reduction launch, no zero fill; nchunk = 1: a plain d bias table);
// one partial row per workgroup + cpb_fold_multi_kernel (fixed order, bit-reproducible).
// Second form (round 5), on the matrix pipe like the forward.  Measured on the VALU form (thread = hidden unit, ~35 vector instructions
// per (pair, unit): 16 FMAs of the two head contractions, 5 for the 3-bit keep test, LDS broadcast reads of the pair-uniform values):
// 250 - 345 us per step at depth 12, 122 us without mask and staging.  Here, per 32-pair step and 16-unit tile of a wave:
//   d hidden[p][j] = sum_h d[h][p] W2[h][j]     two 16 x 16 x 16 products (A = d^T of a 16-pair tile, B = W2 columns: registers)
//   elementwise on the accumulators (lane = unit j, 8 pairs): pre-activation, ReLU gate, keep bit, the three first-layer sums
//   d W2[h][j]    += sum_p d[h][p] hidden[p][j]  one 16 x 16 x 32 product whose B operand is the hidden tile just computed: the k-slot
//                                                order of a lane -- pairs 4g .. 4g+3 of the first 16-pair tile, then of the second -- is
//                                                the accumulator layout of the two d hidden tiles, and the A operand (d rows) is read
//                                                from LDS in the same order
// every operand split hi + lo (three MFMAs per product): fp32 accuracy.  The keep bits arrive pair-major (a word = 8 units of one pair)
// and are transposed once per workgroup into LDS (unit-major words of 32 pairs) with wave ballots.
template <int JT, bool VEC4>
__global__ __launch_bounds__(256) void cpb_bwd_mfma_kernel(const float* __restrict__ dpart, int nchunk, const float* const* __restrict__ ptab,
                                                           const uint32_t* __restrict__ keep_words, int L, int ww, int heads, float scale,
                                                           float* __restrict__ part) {
    constexpr int Hd = 64 * JT, KS = Hd / 32;
    __shared__ __attribute__((aligned(16))) float dbs[16][CPB_PPB];          // d bias sums; rows >= heads are zero
    __shared__ __attribute__((aligned(16))) float rs[2][CPB_PPB];            // the pairs' two log-spaced coordinates
    __shared__ __attribute__((aligned(16))) uint32_t bitsT[Hd][CPB_PPB / 32]; // keep bits, unit-major
    const int blk = blockIdx.y;
    const gfloat* __restrict__ w1 = (const gfloat*)ptab[blk * 4 + 0];
    const gfloat* __restrict__ b1 = (const gfloat*)ptab[blk * 4 + 1];
    const gfloat* __restrict__ w2 = (const gfloat*)ptab[blk * 4 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int L2 = L * L;
    const int p0 = blockIdx.x * CPB_PPB, p1 = min(L2, p0 + CPB_PPB);
    const float* dsrc = dpart + (size_t)blk * nchunk * heads * L2;
    const size_t cstride = (size_t)heads * L2;
    // ---- stage: dbs[h][pp] = sum over the chunks' tables (16-byte loads, 8 chunk loads in flight per thread)
    if constexpr (VEC4) {                      // L2 % 4 == 0: every (chunk, head) row segment is 16-byte aligned
        for (int i = tid; i < 16 * (CPB_PPB / 4); i += 256) {
            const int h = i / (CPB_PPB / 4), q4 = i % (CPB_PPB / 4), pp = 4 * q4;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            if (h < heads && p0 + pp < p1) {                // (p1 - p0 is a multiple of 4)
                const float* src = dsrc + (size_t)h * L2 + p0 + pp;
                int c = 0;
                for (; c + 7 < nchunk; c += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = *(const f32x4*)(src + (size_t)(c + e) * cstride);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { s0 += v[e]; s1 += v[e + 1]; }
                }
                for (; c < nchunk; ++c) s0 += *(const f32x4*)(src + (size_t)c * cstride);
            }
            *(f32x4*)&dbs[h][pp] = s0 + s1;
        }
    } else {
        for (int i = tid; i < 16 * CPB_PPB; i += 256) {
            const int h = i / CPB_PPB, pp = i % CPB_PPB;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (h < heads && p0 + pp < p1) {
                const float* src = dsrc + (size_t)h * L2 + p0 + pp;
                int c = 0;
                for (; c + 3 < nchunk; c += 4) {
                    s0 += src[(size_t)c * cstride];
                    s1 += src[(size_t)(c + 1) * cstride];
                    s2 += src[(size_t)(c + 2) * cstride];
                    s3 += src[(size_t)(c + 3) * cstride];
                }
                for (; c < nchunk; ++c) s0 += src[(size_t)c * cstride];
            }
            dbs[h][pp] = (s0 + s1) + (s2 + s3);
        }
    }
    for (int i = tid; i < CPB_PPB; i += 256) {
        float r0, r1;
        rel_coord(min(p0 + i, L2 - 1), L, ww, r0, r1);
        rs[0][i] = r0;
        rs[1][i] = r1;
    }
    // keep bits: (64-pair half, word index) items over the waves; lane = pair; eight ballots per item = the eight units of the word.
    // All of a wave's words are requested before the first ballot (one load per item waited for on the spot: 24 dependent round trips)
    if (keep_words) {
        constexpr int NIT = 2 * (Hd / 8) / 4;          // items per wave
        uint32_t wd[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = wave + 4 * i, hp = it & 1, w = it >> 1;
            const int p = min(p0 + 64 * hp + lane, L2 - 1);
            wd[i] = keep_words[((size_t)blk * L2 + p) * (Hd / 8) + w];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = wave + 4 * i, hp = it & 1, w = it >> 1;
            const uint32_t dec = keep_dec(wd[i]);
            const int j0 = 32 * (w % KS) + 8 * (w / KS);            // inverse of keep_widx: units j0 .. j0 + 7
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned long long m = __ballot((dec >> e) & 1u);
                if (lane == 0) { bitsT[j0 + e][2 * hp] = (uint32_t)m; bitsT[j0 + e][2 * hp + 1] = (uint32_t)(m >> 32); }
            }
        }
    }
    __syncthreads();
    // pair-side MFMA operands of the four 32-pair steps, split hi + lo ONCE per workgroup (wave w builds step w) and kept in LDS as
    // ready-made fragments: they are the same for every unit tile
    //   fA[st][hi | lo][lane] : A[i = head fr][k-slot e] = d[fr][pa + e] (e < 4), d[fr][pc + e - 4]      (d W2 product, K = 32)
    //   fT[st][T][hi | lo][lane] : A[i = pair fr of tile T][k = head 4g + e] = d[4g + e][32 st + 16 T + fr]   (d hidden product, K = 16)
    __shared__ __attribute__((aligned(16))) uint16_t fA[CPB_PPB / 32][2][64][8];
    __shared__ __attribute__((aligned(16))) uint16_t fT[CPB_PPB / 32][2][2][64][4];
    static_assert(CPB_PPB / 32 == 4, "one step per wave in the fragment prologue");
    {
        const int st = wave, pa = 32 * st + 4 * g, pc = pa + 16;
        const f32x4 v0 = *(const f32x4*)&dbs[fr][pa], v1 = *(const f32x4*)&dbs[fr][pc];
        const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        bf16x8 h8, l8;
        split_hi_lo8(x, h8, l8);
        *(bf16x8*)&fA[st][0][lane][0] = h8;
        *(bf16x8*)&fA[st][1][lane][0] = l8;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = dbs[4 * g + e][32 * st + 16 * T + fr];
            bf16x4 h4, l4;
            split_hi_lo4(y, h4, l4);
            *(bf16x4*)&fT[st][T][0][lane][0] = h4;
            *(bf16x4*)&fT[st][T][1][lane][0] = l4;
        }
    }
    __syncthreads();
    // the wave's units j = Hd / 4 * wave + 16 jt + fr: one unit tile after the other (everything a tile accumulates lives only in its pass)
    const int jw = (Hd / 4) * wave;
    const int n = 3 * Hd + heads * Hd + heads;
    float* row = part + ((size_t)blk * gridDim.x + blockIdx.x) * n;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int jt = 0; jt < JT; ++jt) {
        const int j = jw + 16 * jt + fr;
        const float wa = w1[2 * j] * scale, wb = w1[2 * j + 1] * scale, bb = b1[j] * scale;      // relu(s x) = s relu(x)
        bf16x4 w2h, w2l;                       // B[k = head 4g + e][n = unit]: W2 columns, hi | lo
        {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (4 * g + e < heads) ? w2[(4 * g + e) * Hd + j] : 0.f;
            split_hi_lo4(x, w2h, w2l);
        }
        f32x4 g2 = zero4;
        float ga = 0.f, gb = 0.f, gbias = 0.f;
#pragma unroll 1
        for (int st = 0; st < CPB_PPB / 32; ++st) {
            const int pa = 32 * st + 4 * g, pc = pa + 16;          // the lane's pairs: pa .. pa + 3 (tile 0), pc .. pc + 3 (tile 1)
            const f32x4 r0a = *(const f32x4*)&rs[0][pa], r0c = *(const f32x4*)&rs[0][pc];
            const f32x4 r1a = *(const f32x4*)&rs[1][pa], r1c = *(const f32x4*)&rs[1][pc];
            const float r0v[8] = {r0a[0], r0a[1], r0a[2], r0a[3], r0c[0], r0c[1], r0c[2], r0c[3]};
            const float r1v[8] = {r1a[0], r1a[1], r1a[2], r1a[3], r1c[0], r1c[1], r1c[2], r1c[3]};
            const int bw = keep_words ? (int)(bitsT[j][st] >> (4 * g)) : -1;      // bit e: pair pa + e; bit 16 + e: pair pc + e
            f32x4 dh[2];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const bf16x4 ath = *(const bf16x4*)&fT[st][T][0][lane][0], atl = *(const bf16x4*)&fT[st][T][1][lane][0];
                dh[T] = mfma16(ath, w2h, zero4);
                dh[T] = mfma16(atl, w2h, dh[T]);
                dh[T] = mfma16(ath, w2l, dh[T]);
            }
            float hdn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = fmaf(wa, r0v[e], fmaf(wb, r1v[e], bb));
                const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(bw, (e < 4 ? e : 12 + e), 1);
                hdn[e] = __uint_as_float(__float_as_uint(fmaxf(pre, 0.f)) & m);
                const float d_ = (e < 4) ? dh[0][e] : dh[1][e - 4];
                const float dm = __uint_as_float(__float_as_uint(pre > 0.f ? d_ : 0.f) & m);
                ga = fmaf(dm, r0v[e], ga);
                gb = fmaf(dm, r1v[e], gb);
                gbias += dm;
            }
            bf16x8 Bh, Bl;
            split_hi_lo8(hdn, Bh, Bl);
            const bf16x8 adh = *(const bf16x8*)&fA[st][0][lane][0], adl = *(const bf16x8*)&fA[st][1][lane][0];
            g2 = mfma32(adh, Bh, g2);
            g2 = mfma32(adl, Bh, g2);
            g2 = mfma32(adh, Bl, g2);
        }
        ga += __shfl_xor(ga, 16); ga += __shfl_xor(ga, 32);
        gb += __shfl_xor(gb, 16); gb += __shfl_xor(gb, 32);
        gbias += __shfl_xor(gbias, 16); gbias += __shfl_xor(gbias, 32);
        if (g == 0) {                          // (the ReLU gate's scale: d hidden / d pre = scale m [pre > 0])
            row[2 * j] = ga * scale;
            row[2 * j + 1] = gb * scale;
            row[2 * Hd + j] = gbias * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < heads) row[3 * Hd + (4 * g + r) * Hd + j] = g2[r];
    }
    if (tid < heads) {
        float s_ = 0.f;
        for (int pp = 0; pp < CPB_PPB; ++pp) s_ += dbs[tid][pp];
        row[3 * Hd + heads * Hd + tid] = s_;
    }
}

// grads[blk][i] += the sum of the block's partial rows (fixed order); layout of a row = [dw1 (2 Hd) | db1 (Hd) | dw2 (heads Hd) | db2 (heads)]
__global__ __launch_bounds__(512) void cpb_fold_multi_kernel(const float* __restrict__ part, int rows, int n, float* __restrict__ grads) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, i = blockIdx.x * 64 + lane, blk = blockIdx.y;
    const float* src = part + (size_t)blk * rows * n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int r = rg;
        for (; r + 24 < rows; r += 32) {
            a0 += src[(size_t)r * n + i];
            a1 += src[(size_t)(r + 8) * n + i];
            a2 += src[(size_t)(r + 16) * n + i];
            a3 += src[(size_t)(r + 24) * n + i];
        }
        for (; r < rows; r += 8) a0 += src[(size_t)r * n + i];
    }
    red[rg][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && i < n)
        grads[(size_t)blk * n + i] += ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + ((red[4][lane] + red[5][lane]) + (red[6][lane] + red[7][lane]));
}

}  // namespace

extern "C" int swv2_cpb_fwd(const float* w1, const float* b1, const float* w2, const float* b2, const void* keep_bf16,
                            float* bias, int wh, int ww, int heads, int hidden, float drop_p, void* stream) {
    SWV2_CHECK_ARG(w1 && b1 && w2 && b2 && bias, "cpb_fwd: null pointer");
    SWV2_CHECK_ARG(heads > 0 && heads <= CPB_MAX_HEADS && hidden > 0 && hidden <= 512 && drop_p >= 0.f && drop_p < 1.f,
                   "cpb_fwd: heads <= %d, hidden <= 512 required (heads=%d hidden=%d)", CPB_MAX_HEADS, heads, hidden);
    const int L = wh * ww, L2 = L * L;
    const float scale = 1.f / (1.f - drop_p);
#define CPB_FWD(HM)                                                                                                  \
    hipLaunchKernelGGL((cpb_fwd_kernel<HM>), dim3(cdiv(L2, 64)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2,  \
                       (const uint16_t*)keep_bf16, bias, L, ww, heads, hidden, scale)
    if (heads <= 4) CPB_FWD(4);
    else if (heads <= 8) CPB_FWD(8);
    else if (heads <= 16) CPB_FWD(16);
    else CPB_FWD(CPB_MAX_HEADS);
#undef CPB_FWD
    SWV2_CHECK_LAUNCH("swv2_cpb_fwd");
    return SWV2_OK;
}

extern "C" int swv2_cpb_bwd_ws(const float* dbias, const float* w1, const float* b1, const float* w2, const void* keep_bf16,
                               float* dw1, float* db1, float* dw2, float* db2, int wh, int ww, int heads, int hidden,
                               float drop_p, void* ws, size_t ws_bytes, void* stream);

extern "C" size_t swv2_cpb_bwd_ws_bytes(int wh, int ww, int heads, int hidden) {
    if (wh <= 0 || ww <= 0 || heads <= 0 || hidden <= 0) return 0;
    const int L2 = wh * ww * wh * ww;
    return (size_t)cdiv(L2, CPB_PPB) * (3 * hidden + heads * hidden + heads) * sizeof(float);
}

extern "C" int swv2_cpb_bwd(const float* dbias, const float* w1, const float* b1, const float* w2, const void* keep_bf16,
                            float* dw1, float* db1, float* dw2, float* db2, int wh, int ww, int heads, int hidden,
                            float drop_p, void* stream) {
    return swv2_cpb_bwd_ws(dbias, w1, b1, w2, keep_bf16, dw1, db1, dw2, db2, wh, ww, heads, hidden, drop_p, nullptr, 0, stream);
}

extern "C" int swv2_cpb_bwd_ws(const float* dbias, const float* w1, const float* b1, const float* w2, const void* keep_bf16,
                               float* dw1, float* db1, float* dw2, float* db2, int wh, int ww, int heads, int hidden,
                               float drop_p, void* ws, size_t ws_bytes, void* stream) {
    SWV2_CHECK_ARG(dbias && w1 && b1 && w2 && dw1 && db1 && dw2 && db2, "cpb_bwd: null pointer");
    SWV2_CHECK_ARG(heads > 0 && heads <= CPB_MAX_HEADS && hidden > 0 && hidden <= 512 && drop_p >= 0.f && drop_p < 1.f,
                   "cpb_bwd: heads <= %d, hidden <= 512 required", CPB_MAX_HEADS);
    const int L = wh * ww, L2 = L * L, ppb = CPB_PPB;
    const int threads = cdiv(hidden, 64) * 64;
    const float scale = 1.f / (1.f - drop_p);
    // with a workspace (swv2_cpb_bwd_ws_bytes): partial rows + a fixed-order fold instead of float atomics.  A workspace that is
    // too small is an error, not a silent switch to the (not bit-reproducible) atomics path
    SWV2_CHECK_ARG(!ws || ws_bytes >= swv2_cpb_bwd_ws_bytes(wh, ww, heads, hidden), "cpb_bwd_ws: workspace of %zu bytes, %zu needed",
                   ws_bytes, swv2_cpb_bwd_ws_bytes(wh, ww, heads, hidden));
    float* part = ws ? (float*)ws : nullptr;
#define CPB_BWD(HM)                                                                                                  \
    hipLaunchKernelGGL((cpb_bwd_kernel<HM>), dim3(cdiv(L2, ppb)), dim3(threads), 0, (hipStream_t)stream, dbias, w1, b1, \
                       w2, (const uint16_t*)keep_bf16, dw1, db1, dw2, db2, L, ww, heads, hidden, scale, ppb, part)
    if (heads <= 4) CPB_BWD(4);
    else if (heads <= 8) CPB_BWD(8);
    else if (heads <= 16) CPB_BWD(16);
    else CPB_BWD(CPB_MAX_HEADS);
#undef CPB_BWD
    if (part) {
        const int n = 3 * hidden + heads * hidden + heads;
        hipLaunchKernelGGL(cpb_fold_kernel, dim3(cdiv(n, 64)), dim3(512), 0, (hipStream_t)stream, (const float*)part, cdiv(L2, ppb), n, dw1,
                           db1, dw2, db2, hidden, heads);
    }
    SWV2_CHECK_LAUNCH("swv2_cpb_bwd");
    return SWV2_OK;
}

// ---- all blocks of a stage in one launch (kernels above) --------------------------------------------------------
namespace {
int cpb_multi_check(const void* ptab, int nblk, int wh, int ww, int heads, int hidden, float drop_p) {
    SWV2_CHECK_ARG(ptab && nblk > 0 && wh > 0 && ww > 0, "cpb_multi: null pointer table or empty geometry");
    SWV2_CHECK_ARG(heads > 0 && heads <= 16 && (hidden == 64 || hidden == 128 || hidden == 256 || hidden == 384 || hidden == 512),
                   "cpb_multi: heads <= 16 and hidden in {64, 128, 256, 384, 512} (got heads=%d hidden=%d)", heads, hidden);
    SWV2_CHECK_ARG(fabsf(drop_p - 0.125f) < 1e-6f, "cpb_multi: the keep words express the reference's Dropout(0.125) (three random bytes per 8 units), got p = %g",
                   (double)drop_p);
    return SWV2_OK;
}
}  // namespace

extern "C" int swv2_cpb_fwd_multi(const float* const* params_dev, int nblk, const uint32_t* keep_bits, float* bias, int wh, int ww,
                                  int heads, int hidden, float drop_p, void* stream) {
    int rc = cpb_multi_check(params_dev, nblk, wh, ww, heads, hidden, drop_p);
    if (rc) return rc;
    SWV2_CHECK_ARG(bias, "cpb_fwd_multi: null output");
    const int L = wh * ww, L2 = L * L;
    const float scale = keep_bits ? 1.f / (1.f - drop_p) : 1.f;        // (eval mode: no dropout, no scale)
    if (hidden % 128 == 0)
        hipLaunchKernelGGL((cpb_fwd_mfma_kernel<4, 4>), dim3(cdiv(L2, 256), nblk), dim3(256), 0, (hipStream_t)stream, params_dev, keep_bits, bias, L,
                           ww, heads, hidden, scale);
    else
        hipLaunchKernelGGL((cpb_fwd_mfma_kernel<4, 2>), dim3(cdiv(L2, 256), nblk), dim3(256), 0, (hipStream_t)stream, params_dev, keep_bits, bias, L,
                           ww, heads, hidden, scale);
    SWV2_CHECK_LAUNCH("swv2_cpb_fwd_multi");
    return SWV2_OK;
}

extern "C" size_t swv2_cpb_bwd_multi_ws_bytes(int nblk, int wh, int ww, int heads, int hidden) {
    if (nblk <= 0) return 0;
    return (size_t)nblk * swv2_cpb_bwd_ws_bytes(wh, ww, heads, hidden);
}

extern "C" int swv2_cpb_bwd_multi(const float* dbias_tables, int nchunk, const float* const* params_dev, int nblk,
                                  const uint32_t* keep_bits, float* grads, int wh, int ww, int heads, int hidden, float drop_p, void* ws,
                                  size_t ws_bytes, void* stream) {
    int rc = cpb_multi_check(params_dev, nblk, wh, ww, heads, hidden, drop_p);
    if (rc) return rc;
    SWV2_CHECK_ARG(dbias_tables && grads && nchunk > 0, "cpb_bwd_multi: null pointer or nchunk = %d", nchunk);
    SWV2_CHECK_ARG(ws && ws_bytes >= swv2_cpb_bwd_multi_ws_bytes(nblk, wh, ww, heads, hidden), "cpb_bwd_multi: workspace of %zu bytes, %zu needed",
                   ws_bytes, swv2_cpb_bwd_multi_ws_bytes(nblk, wh, ww, heads, hidden));
    const int L = wh * ww, L2 = L * L, rows = cdiv(L2, CPB_PPB), n = 3 * hidden + heads * hidden + heads;
    const float scale = keep_bits ? 1.f / (1.f - drop_p) : 1.f;
#define CPB_BWDM(JT_)                                                                                                       \
    do {                                                                                                                    \
        if ((L2 & 3) == 0)                                                                                                  \
            hipLaunchKernelGGL((cpb_bwd_mfma_kernel<JT_, true>), dim3(rows, nblk), dim3(256), 0, (hipStream_t)stream, dbias_tables, nchunk, \
                               params_dev, keep_bits, L, ww, heads, scale, (float*)ws);                                     \
        else                                                                                                                \
            hipLaunchKernelGGL((cpb_bwd_mfma_kernel<JT_, false>), dim3(rows, nblk), dim3(256), 0, (hipStream_t)stream, dbias_tables, nchunk, \
                               params_dev, keep_bits, L, ww, heads, scale, (float*)ws);                                     \
    } while (0)
    switch (hidden) {
        case 64: CPB_BWDM(1); break;
        case 128: CPB_BWDM(2); break;
        case 256: CPB_BWDM(4); break;
        case 384: CPB_BWDM(6); break;
        default: CPB_BWDM(8); break;          // 512 (cpb_multi_check)
    }
#undef CPB_BWDM
    static_assert(CPB_PPB % 32 == 0 && CPB_PPB <= 512, "pairs per workgroup");
    hipLaunchKernelGGL(cpb_fold_multi_kernel, dim3(cdiv(n, 64), nblk), dim3(512), 0, (hipStream_t)stream, (const float*)ws, rows, n, grads);
    SWV2_CHECK_LAUNCH("swv2_cpb_bwd_multi");
    return SWV2_OK;
}
