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
//   ptab      : device table [nblk][4] of the blocks' parameter pointers (w1 [Hd][2], b1 [Hd], w2 [heads][Hd], b2 [heads])
//   keep_bits : u32 [nblk][L^2][Hd / 8], hidden unit j of a pair = the 3-bit field (word[j / 8] >> 3 (j % 8)) & 7 of uniformly
//               random bits drawn by the caller (ONE torch `random_()` launch); dropped iff field < thr, i.e. with probability
//               thr / 8 (Dropout(0.125): thr = 1); kept units are scaled by 1 / (1 - drop_p).  NULL: eval mode.
// ------------------------------------------------------------------------------------------------
// PPL pairs per lane: the uniform weight values (88 scalar loads per 8 hidden units) serve PPL x 64 pairs per wave, so their latency is
// amortised over PPL times the vector work (PPL = 1: 88 us per step at depth 12 -- every trip waited for its scalar loads)
template <int HEADS_MAX, int PPL>
__global__ __launch_bounds__(256) void cpb_fwd_multi_kernel(const float* const* __restrict__ ptab, const uint32_t* __restrict__ keep_bits,
                                                            float* __restrict__ bias_all, int L, int ww, int heads, int Hd, float scale,
                                                            uint32_t thr) {
    constexpr int PW = 64 * PPL;                       // pairs per workgroup
    __shared__ float red[4][HEADS_MAX][PW];
    const int blk = blockIdx.y;
    // (pointers loaded from memory are generic-address-space pointers: cast to global so that the uniform weight loads stay scalar)
    const gfloat* __restrict__ w1 = (const gfloat*)ptab[blk * 4 + 0];
    const gfloat* __restrict__ b1 = (const gfloat*)ptab[blk * 4 + 1];
    const gfloat* __restrict__ w2 = (const gfloat*)ptab[blk * 4 + 2];
    const gfloat* __restrict__ b2 = (const gfloat*)ptab[blk * 4 + 3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L2 = L * L;
    float r0[PPL], r1[PPL];
    const uint32_t* krow[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int p = min(blockIdx.x * PW + 64 * i + lane, L2 - 1);
        rel_coord(p, L, ww, r0[i], r1[i]);
        krow[i] = keep_bits ? keep_bits + ((size_t)blk * L2 + p) * (Hd >> 3) : nullptr;
    }
    const int jq = Hd >> 2, j0 = wave * jq, j1 = j0 + jq;          // Hd % 32 == 0 (checked by the host): 8 units per trip
    float acc[PPL][HEADS_MAX];
#pragma unroll
    for (int i = 0; i < PPL; ++i)
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) acc[i][h] = 0.f;
    for (int j = j0; j < j1; j += 8) {
        uint32_t kw[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) kw[i] = keep_bits ? krow[i][j >> 3] : 0xffffffffu;
        float wa[8], wb[8], bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { wa[u] = w1[2 * (j + u)]; wb[u] = w1[2 * (j + u) + 1]; bb[u] = b1[j + u]; }
        float hd8[PPL][8];
#pragma unroll
        for (int i = 0; i < PPL; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float hv = fmaxf(fmaf(wa[u], r0[i], fmaf(wb[u], r1[i], bb[u])), 0.f);
                hd8[i][u] = keep_bits ? ((((kw[i] >> (3 * u)) & 7u) >= thr) ? hv * scale : 0.f) : hv;
            }
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) {          // rows past `heads` re-read the last head; their sums are dropped
            const gfloat* w2h = w2 + min(h, heads - 1) * Hd + j;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wv = w2h[u];
#pragma unroll
                for (int i = 0; i < PPL; ++i) acc[i][h] = fmaf(wv, hd8[i][u], acc[i][h]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i)
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) red[wave][h][64 * i + lane] = acc[i][h];
    __syncthreads();
    float* bias = bias_all + (size_t)blk * heads * L2;
    for (int i = threadIdx.x; i < heads * PW; i += 256) {
        const int h = i / PW, l = i % PW, pp = blockIdx.x * PW + l;
        if (pp < L2) bias[(size_t)h * L2 + pp] = red[0][h][l] + red[1][h][l] + red[2][h][l] + red[3][h][l] + b2[h];
    }
}

// backward of all blocks: d bias of block b = the sum of `nchunk` tables [heads][L^2] (the per-workgroup tables the attention
// backward leaves, summed here while they are staged -- no reduction launch, no zero fill; nchunk = 1: a plain d bias table);
// one partial row per workgroup + cpb_fold_multi_kernel (fixed order, bit-reproducible).
// The staging is the kernel's cost (nchunk = 32 at the benchmark shape: 27 MB per block, 324 MB per step): all CPB_PPB = 128 pairs of the
// workgroup in ONE round, 16-byte loads (4 consecutive pairs of one (chunk, head) row), 8 chunk loads in flight per thread -- the first
// version (64 pairs per round, 4-byte loads, 4 in flight: 6 KB in flight per workgroup) ran at 1.16 TB/s, 281 us per step.
template <int HEADS_MAX, bool VEC4, int U>
__global__ __launch_bounds__(128) void cpb_bwd_multi_kernel(const float* __restrict__ dpart, int nchunk, const float* const* __restrict__ ptab,
                                                            const uint32_t* __restrict__ keep_bits, int L, int ww, int heads, int Hd,
                                                            float scale, uint32_t thr, float* __restrict__ part) {
    // U hidden units per thread (j = tid + 128 u): the pair-uniform values of the inner loop (d bias of the heads, the two coordinates)
    // are LDS broadcast reads, 10 per pair and wave -- with one unit per thread (6 waves at 384 units) those reads, not the arithmetic,
    // bounded the kernel (~250 us per step at depth 12); three units per thread share each read
    __shared__ __attribute__((aligned(16))) float dbs[HEADS_MAX + 2][CPB_PPB];   // [d bias of head 0.., r0, r1][pair]
    const int blk = blockIdx.y;
    const gfloat* __restrict__ w1 = (const gfloat*)ptab[blk * 4 + 0];
    const gfloat* __restrict__ b1 = (const gfloat*)ptab[blk * 4 + 1];
    const gfloat* __restrict__ w2 = (const gfloat*)ptab[blk * 4 + 2];
    const int tid = threadIdx.x;
    bool act[U];
    float wa[U], wb[U], bb[U], w2r[U][HEADS_MAX], g2[U][HEADS_MAX], ga[U], gb[U], gbias[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = tid + 128 * u;
        act[u] = j < Hd;
        wa[u] = act[u] ? w1[2 * j] : 0.f;
        wb[u] = act[u] ? w1[2 * j + 1] : 0.f;
        bb[u] = act[u] ? b1[j] : 0.f;
        ga[u] = gb[u] = gbias[u] = 0.f;
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) { w2r[u][h] = (act[u] && h < heads) ? w2[h * Hd + j] : 0.f; g2[u][h] = 0.f; }
    }
    float gb2 = 0.f;
    const int L2 = L * L;
    const int p0 = blockIdx.x * CPB_PPB, p1 = min(L2, p0 + CPB_PPB);
    const float* dsrc = dpart + (size_t)blk * nchunk * heads * L2;
    const size_t cstride = (size_t)heads * L2;
    // ---- stage: dbs[h][pp] = sum over the chunks' tables (16-byte loads, 8 chunk loads in flight per thread), + the pairs' coordinates
    if constexpr (VEC4) {                      // L2 % 4 == 0: every (chunk, head) row segment is 16-byte aligned
        for (int i = tid; i < heads * (CPB_PPB / 4); i += 128) {
            const int h = i / (CPB_PPB / 4), q4 = i % (CPB_PPB / 4), pp = 4 * q4;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            if (p0 + pp < p1) {                // (p1 - p0 is a multiple of 4)
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
        for (int i = tid; i < heads * CPB_PPB; i += 128) {
            const int h = i / CPB_PPB, pp = i % CPB_PPB;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (p0 + pp < p1) {
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
    for (int i = tid; i < CPB_PPB; i += 128) {
        float r0, r1;
        rel_coord(min(p0 + i, L2 - 1), L, ww, r0, r1);
        dbs[HEADS_MAX][i] = r0;
        dbs[HEADS_MAX + 1][i] = r1;
    }
    // the pairs' keep flags of this thread's hidden units
    uint32_t kbits[U][CPB_PPB / 32];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int jc = act[u] ? tid + 128 * u : 0;
#pragma unroll
        for (int w = 0; w < CPB_PPB / 32; ++w) kbits[u][w] = ~0u;
        if (keep_bits) {
            const uint32_t* kw = keep_bits + (size_t)blk * L2 * (Hd >> 3) + (jc >> 3);
            const int sh = 3 * (jc & 7);
#pragma unroll
            for (int w = 0; w < CPB_PPB / 32; ++w) {
                uint32_t b = 0;
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    b |= (uint32_t)(((kw[(size_t)min(p0 + 32 * w + e, L2 - 1) * (Hd >> 3)] >> sh) & 7u) >= thr) << e;
                kbits[u][w] = b;
            }
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int pp = 0; pp < CPB_PPB; ++pp) {          // pairs past p1 hold d bias = 0 and contribute nothing
        const float r0 = dbs[HEADS_MAX][pp], r1 = dbs[HEADS_MAX + 1][pp];
        float d[HEADS_MAX];
#pragma unroll
        for (int h = 0; h < HEADS_MAX; ++h) d[h] = (h < heads) ? dbs[h][pp] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pre = fmaf(wa[u], r0, fmaf(wb[u], r1, bb[u]));
            const float m = keep_bits ? (((kbits[u][pp >> 5] >> (pp & 31)) & 1u) ? scale : 0.f) : 1.f;
            const float hdn = fmaxf(pre, 0.f) * m;
            float dh = 0.f;
#pragma unroll
            for (int h = 0; h < HEADS_MAX; ++h) {
                g2[u][h] = fmaf(d[h], hdn, g2[u][h]);
                dh = fmaf(d[h], w2r[u][h], dh);
            }
            dh = (pre > 0.f) ? dh * m : 0.f;
            ga[u] = fmaf(dh, r0, ga[u]);
            gb[u] = fmaf(dh, r1, gb[u]);
            gbias[u] += dh;
        }
    }
    if (tid < heads)
        for (int pp = 0; pp < CPB_PPB; ++pp) gb2 += dbs[tid][pp];
    const int n = 3 * Hd + heads * Hd + heads;
    float* row = part + ((size_t)blk * gridDim.x + blockIdx.x) * n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = tid + 128 * u;
        if (act[u]) {
            row[2 * j] = ga[u];
            row[2 * j + 1] = gb[u];
            row[2 * Hd + j] = gbias[u];
#pragma unroll
            for (int h = 0; h < HEADS_MAX; ++h)
                if (h < heads) row[3 * Hd + h * Hd + j] = g2[u][h];
        }
    }
    if (tid < heads) row[3 * Hd + heads * Hd + tid] = gb2;
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
int cpb_multi_check(const void* ptab, int nblk, int wh, int ww, int heads, int hidden, float drop_p, uint32_t* thr) {
    SWV2_CHECK_ARG(ptab && nblk > 0 && wh > 0 && ww > 0, "cpb_multi: null pointer table or empty geometry");
    SWV2_CHECK_ARG(heads > 0 && heads <= CPB_MAX_HEADS && hidden > 0 && hidden <= 512 && (hidden % 32) == 0,
                   "cpb_multi: heads <= %d, hidden <= 512 and a multiple of 32 required (heads=%d hidden=%d)", CPB_MAX_HEADS, heads, hidden);
    const float t = drop_p * 8.f;
    const int ti = (int)(t + 0.5f);
    SWV2_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && ti >= 0 && ti < 8 && fabsf(t - (float)ti) < 1e-5f,
                   "cpb_multi: the 3-bit keep fields express drop probabilities k / 8 (got %g)", (double)drop_p);
    *thr = (uint32_t)ti;
    return SWV2_OK;
}
}  // namespace

extern "C" int swv2_cpb_fwd_multi(const float* const* params_dev, int nblk, const uint32_t* keep_bits, float* bias, int wh, int ww,
                                  int heads, int hidden, float drop_p, void* stream) {
    uint32_t thr = 0;
    int rc = cpb_multi_check(params_dev, nblk, wh, ww, heads, hidden, drop_p, &thr);
    if (rc) return rc;
    SWV2_CHECK_ARG(bias, "cpb_fwd_multi: null output");
    const int L = wh * ww, L2 = L * L;
    const float scale = 1.f / (1.f - drop_p);
#define CPB_FWDM(HM, PPL)                                                                                                       \
    hipLaunchKernelGGL((cpb_fwd_multi_kernel<HM, PPL>), dim3(cdiv(L2, 64 * PPL), nblk), dim3(256), 0, (hipStream_t)stream, params_dev, \
                       keep_bits, bias, L, ww, heads, hidden, scale, thr)
    if (heads <= 4) CPB_FWDM(4, 4);
    else if (heads <= 8) CPB_FWDM(8, 4);
    else if (heads <= 16) CPB_FWDM(16, 2);
    else CPB_FWDM(CPB_MAX_HEADS, 1);
#undef CPB_FWDM
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
    uint32_t thr = 0;
    int rc = cpb_multi_check(params_dev, nblk, wh, ww, heads, hidden, drop_p, &thr);
    if (rc) return rc;
    SWV2_CHECK_ARG(dbias_tables && grads && nchunk > 0, "cpb_bwd_multi: null pointer or nchunk = %d", nchunk);
    SWV2_CHECK_ARG(ws && ws_bytes >= swv2_cpb_bwd_multi_ws_bytes(nblk, wh, ww, heads, hidden), "cpb_bwd_multi: workspace of %zu bytes, %zu needed",
                   ws_bytes, swv2_cpb_bwd_multi_ws_bytes(nblk, wh, ww, heads, hidden));
    const int L = wh * ww, L2 = L * L, rows = cdiv(L2, CPB_PPB), n = 3 * hidden + heads * hidden + heads;
    const float scale = 1.f / (1.f - drop_p);
#define CPB_BWDM2(HM, V, UU)                                                                                                 \
    hipLaunchKernelGGL((cpb_bwd_multi_kernel<HM, V, UU>), dim3(rows, nblk), dim3(128), 0, (hipStream_t)stream, dbias_tables, nchunk, \
                       params_dev, keep_bits, L, ww, heads, hidden, scale, thr, (float*)ws)
#define CPB_BWDM(HM)                                                                                                        \
    do {                                                                                                                    \
        const bool v4 = (L2 & 3) == 0;                                                                                      \
        if (hidden <= 128) { if (v4) CPB_BWDM2(HM, true, 1); else CPB_BWDM2(HM, false, 1); }                                \
        else if (hidden <= 256) { if (v4) CPB_BWDM2(HM, true, 2); else CPB_BWDM2(HM, false, 2); }                           \
        else if (hidden <= 384) { if (v4) CPB_BWDM2(HM, true, 3); else CPB_BWDM2(HM, false, 3); }                           \
        else { if (v4) CPB_BWDM2(HM, true, 4); else CPB_BWDM2(HM, false, 4); }                                              \
    } while (0)
    if (heads <= 4) CPB_BWDM(4);
    else if (heads <= 8) CPB_BWDM(8);
    else if (heads <= 16) CPB_BWDM(16);
    else CPB_BWDM(CPB_MAX_HEADS);
#undef CPB_BWDM2
#undef CPB_BWDM
    static_assert(CPB_PPB % 32 == 0 && CPB_PPB <= 512, "pairs per workgroup");
    hipLaunchKernelGGL(cpb_fold_multi_kernel, dim3(cdiv(n, 64), nblk), dim3(512), 0, (hipStream_t)stream, (const float*)ws, rows, n, grads);
    SWV2_CHECK_LAUNCH("swv2_cpb_bwd_multi");
    return SWV2_OK;
}
