// Cosine window attention core for SwinV2 (forward + backward), gfx950 / CDNA4.
//
// Semantics: reference networks/swinv2_global.py:298-318 (WindowMultiHeadAttention.forward) and :170-198 (NoPos):
//   S = sigma_h * qn kn^T + Bias_h + Mask ; P = softmax(S) ; O = P v
// with qn, kn already L2-normalised (done by the qkv GEMM epilogue, see gemm.hip), sigma_h = exp(min(tau_h, ln 100)).
//
// Data layout (window-ordered, head-major, zero padded; produced by the qkv GEMM epilogue):
//   qkvh [Bw][h][3][Lp][DP] bf16   (s = 0 qn, 1 kn, 2 v),  Lp = 16*LT >= L,  DP = 16*DK >= d
//   oh   [Bw][h][Lp][DP]    bf16
//   lse  [Bw][h][Lp]        fp32   log2-domain log-sum-exp of the scaled scores
// One (window, head) block is one contiguous slab, so every global access of these kernels is a fully coalesced
// 512-byte wave access.  roll/partition never appear here: they are folded into the qkv GEMM's row gather and the
// proj GEMM's row scatter.
//
// Work decomposition: one workgroup = (head, chunk of windows); LT waves, wave w owns one 16-row tile:
//   forward : wave = query tile, "swapped" product S^T = K Q^T so a lane owns one query column: the softmax row
//             reductions are in-lane + 2 cross-lane steps, and S^T accumulators are directly the B operand of
//             O^T = V^T P^T (k16 MFMA, natural k order).  The CPB bias rows of (head, q-tile) live in registers for
//             the lifetime of the workgroup (loaded once), pre-multiplied by log2(e); padded keys carry -1e30.
//   backward: wave = key tile, product S = Q K^T so P / dS accumulators are directly the B operands of
//             dV^T = dO^T P and dK^T = Q^T dS; dK, dV need no cross-wave reduction, the bias gradient of
//             (head, key-tile) accumulates in registers across all windows of the workgroup, and only dQ crosses
//             waves: the bf16 dS tiles go to an LDS image [key][q] and a second, barrier-separated phase (wave = query
//             tile) reads them back transposed -- no atomics (see the phase description above attn_bwd_kernel).
// These first-generation kernels now serve the shapes / options attn2.hip does not cover (window areas <= 64, the
// backward with a CPB bias); attn2.hip holds the small-workgroup kernels used at the benchmark shape.
#include <type_traits>

#include "attn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// CPB bias table, pre-packed once per block call for both kernels (swv2_attn_pack_bias): log2 domain, bf16,
// -1e30 on padded keys, 0 on padded queries.
//   forward part : u32 [h][LT q-tiles][LT key-tiles][2][64 lanes]: the packed pair (r = 2j, 2j+1) of lane (fr = query,
//                  g) -- exactly the registers of attn_fwd_kernel, so its load is one coalesced dword per register
//                  (the fp32 [h][L][L] table read in register layout touches one cache line per lane: 31 K lines =
//                  ~15 us per workgroup, measured 26 us)
//   backward part: bf16 [h][Lp keys][Lp + 4]: the LDS image of attn_bwd_kernel, copied with 16-byte loads (the
//                  in-kernel transposing conversion cost ~30 us per workgroup)
// ------------------------------------------------------------------------------------------------
template <int LT>
struct BiasPack {
    static constexpr int Lp = 16 * LT, DSP = Lp + 4;
    static constexpr size_t FWD_U32 = (size_t)LT * LT * 2 * 64;       // per head
    static constexpr size_t BWD_BF16 = (size_t)Lp * DSP;              // per head
    // third part: fp32 [h][2] = (max, min) of the head's packed values over the real (query, key) pairs -- the bound behind the
    // "fixed maximum" softmax of the bias-capable forward in attn2.hip
    __host__ __device__ static size_t range_off(int heads) { return heads * (FWD_U32 * 4 + BWD_BF16 * 2); }
    static size_t bytes(int heads) { return (range_off(heads) + (size_t)heads * 8 + 15) / 16 * 16; }
};

// one workgroup per (head, table): `nblk` tables [heads][L][L] -> `nblk` packed buffers of `pack_bytes` each
template <int LT>
__global__ __launch_bounds__(1024) void attn_pack_bias_kernel(const float* __restrict__ bias_all, int heads, int L, unsigned char* __restrict__ out_all,
                                                              size_t pack_bytes) {
    using P = BiasPack<LT>;
    __shared__ float red[2][16];
    const int hd = blockIdx.y, blk = blockIdx.z;
    const float* bias = bias_all + (size_t)blk * heads * L * L;
    unsigned char* out = out_all + (size_t)blk * pack_bytes;
    uint32_t* fwd = (uint32_t*)out;
    uint16_t* bwd = (uint16_t*)(out + heads * P::FWD_U32 * 4);
    float* range = (float*)(out + P::range_off(heads));
    auto val = [&](int q, int key) -> float {
        if (key >= L) return SWV2_NEG_BIG;
        return (q < L) ? bias[((size_t)hd * L + q) * L + key] * SWV2_LOG2E : 0.f;
    };
    for (size_t i = threadIdx.x; i < P::FWD_U32; i += blockDim.x) {
        const int lane = i & 63, j = (i >> 6) & 1, t = (int)((i >> 7) % LT), qt = (int)((i >> 7) / LT);
        const int q = 16 * qt + (lane & 15), key = 16 * t + 4 * (lane >> 4) + 2 * j;
        fwd[(size_t)hd * P::FWD_U32 + i] = f2bf2(val(q, key), val(q, key + 1));
    }
    float mx = -3.0e38f, mn = 3.0e38f;
    for (size_t i = threadIdx.x; i < P::BWD_BF16; i += blockDim.x) {
        const int key = (int)(i / P::DSP), q = (int)(i % P::DSP);
        const uint16_t b = (q < P::Lp) ? f2bf(val(q, key)) : (uint16_t)0;
        bwd[(size_t)hd * P::BWD_BF16 + i] = b;
        if (q < L && key < L) { mx = fmaxf(mx, bf2f(b)); mn = fminf(mn, bf2f(b)); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx; red[1][threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mx = fmaxf(mx, red[0][w]); mn = fminf(mn, red[1][w]); }
        range[2 * hd] = mx;
        range[2 * hd + 1] = mn;
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int LT, int DK, bool HAS_BIAS, int LFIX>
__global__ __launch_bounds__(64 * LT) void attn_fwd_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const float* __restrict__ bias,
    const uint32_t* __restrict__ bpack,   // swv2_attn_pack_bias forward part or null
    uint16_t* __restrict__ oh, float* __restrict__ lse, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    using C = AttnCfg<LT, DK>;
    constexpr int Lp = C::Lp, DP = C::DP, SLAB = C::SLAB;
    // two K|V buffers (the next window's slabs land while this one is computed) unless they exceed the LDS: head dims
    // padded to 128 at the 176-token window keep ONE buffer and refill it between two barriers
    constexpr int NBUF = (2 * 2 * SLAB * 2 <= 160 * 1024) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) uint16_t smem[NBUF * 2 * SLAB];   // [buf][K|V][Lp][DP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int qt = tid >> 6;                 // this wave's query tile
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const int q = 16 * qt + fr;              // this lane's query (token index in the window)

    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;

    // CPB bias rows of this (head, q-tile), log2 domain, -1e30 on padded keys; bf16 like the backward's LDS image
    // (identical P in both passes), held as packed pairs: 2 x LT registers
    uint32_t biasp[LT][2];
    if (HAS_BIAS) {
        if (bpack) {
            const uint32_t* src = bpack + (((size_t)hd * LT + qt) * LT) * 128 + lane;
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                biasp[t][0] = src[t * 128];
                biasp[t][1] = src[t * 128 + 64];
            }
        } else {
            // unconditional (clamped) loads, selection afterwards (exec-masked loads are serialised by the compiler)
            float bv[LT][4];
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * t + 4 * g + r;
                    bv[t][r] = bias[((size_t)hd * L + min(q, L - 1)) * L + min(key, L - 1)];
                }
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * t + 4 * g + r;
                    v[r] = (key < L) ? ((q < L) ? bv[t][r] * SWV2_LOG2E : 0.f) : SWV2_NEG_BIG;
                }
                biasp[t][0] = f2bf2(v[0], v[1]);
                biasp[t][1] = f2bf2(v[2], v[3]);
            }
        }
    }

    constexpr int CHUNKS_PER_T = DK;         // K + V slabs = 64*LT*DK 16-byte chunks over 64*LT threads
    uint4 stage[CHUNKS_PER_T];
    // Q fragment (B operand: B[k = d 4g+j][n = query fr]) straight from global, 512 B contiguous per wave; the NEXT
    // window's fragment is fetched with its K / V slabs: a load issued after the prefetch and needed at once would make
    // its s_waitcnt drain the (in-order) prefetch too and expose one HBM round trip per window
    bf16x4 qf[DK], qn[DK];
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB + SLAB;   // K slab, V slab follows
#pragma unroll
        for (int j = 0; j < CHUNKS_PER_T; ++j) stage[j] = *(const uint4*)(base + (size_t)(tid + j * C::NT) * 8);
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) qn[kk] = *(const bf16x4*)(base - SLAB + (size_t)q * DP + 16 * kk + 4 * g);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * 2 * SLAB;
#pragma unroll
        for (int j = 0; j < CHUNKS_PER_T; ++j) *(uint4*)(dst + (size_t)(tid + j * C::NT) * 8) = stage[j];
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue_loads(bw);
    write_stage(0);
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) qf[kk] = qn[kk];
    __syncthreads();

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = NBUF == 2 ? (it & 1) : 0;
        const int bw_next = bw + gridDim.x;
        if (bw_next < Bw) issue_loads(bw_next);

        const uint16_t* Ks = smem + buf * 2 * SLAB;
        const uint16_t* Vs = Ks + SLAB;
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;

        // S^T tiles: rows = keys 16t + 4g + r, column = query fr
        f32x4 acc[LT];
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x4 kf = *(const bf16x4*)(Ks + (16 * t + fr) * DP + 16 * kk + 4 * g);
                acc[t] = mfma16(kf, qf[kk], acc[t]);
            }
        }

        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        float mx, sum = 0.f;
        if (!HAS_BIAS && !do_mask) {
            // no bias, no mask: sigma > 0 commutes with the maximum, so the row maximum is taken over the raw cosines and
            // the scale is folded into the exponent's fma -- one vector instruction per element less
            const int Lc = LFIX > 0 ? LFIX : L;
            mx = SWV2_NEG_BIG;
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * t + 16 > Lc) acc[t][r] = (16 * t + 4 * g + r < Lc) ? acc[t][r] : SWV2_NEG_BIG;
                    mx = fmaxf(mx, acc[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mx *= sc2;
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(acc[t][r], sc2, -mx));
                    acc[t][r] = p;
                    sum += p;
                }
        } else {
            if (do_mask) mx = score_pass<LT, HAS_BIAS, true, LFIX>(acc, biasp, sc2, L, g, mask_thr, q >= mask_thr);
            else         mx = score_pass<LT, HAS_BIAS, false, LFIX>(acc, biasp, sc2, L, g, mask_thr, false);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(acc[t][r] - mx);
                    acc[t][r] = p;
                    sum += p;
                }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

        // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
        f32x4 o[DK];
#pragma unroll
        for (int dt = 0; dt < DK; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            const bf16x4 pb = f2bf4(acc[t]);
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                const bf16x4 vf = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4);
                o[dt] = mfma16(vf, pb, o[dt]);
            }
        }
        // next window's K / V -> the other LDS buffer BEFORE this window's stores are issued: a wait for the prefetch
        // placed after (exec-masked) stores would also wait for their acknowledgements
        if (NBUF == 1) __syncthreads();                   // every wave is done with the only buffer
        if (bw_next < Bw) {
            write_stage(NBUF == 2 ? (buf ^ 1) : 0);
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) qf[kk] = qn[kk];
        }
        const float inv = (q < L) ? 1.f / sum : 0.f;
        uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
#pragma unroll
        for (int dt = 0; dt < DK; ++dt) {
            f32x4 v = o[dt];
            v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
            *(bf16x4*)(orow + 16 * dt + 4 * g) = f2bf4(v);
        }
        if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx + __log2f(sum) : 0.f;
        __syncthreads();
    }
}

// Issue priority that falls with a wave's progress through phase 1 (see attn_bwd_stream.hip: the SIMD arbiter serves its oldest wave first, the
// youngest then finishes the phase alone with nothing to cover its latencies); reset to 0 behind the phase.  -DSWV2_ATTN1_NO_PRIO: A/B builds.
#ifndef SWV2_ATTN1_NO_PRIO
#define SWV2_P1PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define SWV2_P1PRIO(n) do {} while (0)
#endif
#ifdef SWV2_ATTN1_STAMPS          // diagnostic build (tools/probe_attn1_stamps.py): per-phase s_memtime sums of every wave 0
__device__ unsigned long long attn1_stamps[512 * 8];
__device__ unsigned long long attn1_win[64 * 128];        // wave 8 of the first 64 workgroups of head 0: s_memtime at the end of every window
#define GSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define GSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define GSTAMP_DECL
#define GSTAMP_START() do {} while (0)
#define GSTAMP(k) do {} while (0)
#endif
// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Per window, three barrier-separated phases:
//   stage  : q, k, dO slabs -> LDS (16-byte chunks, prefetched into registers during the previous window),
//            delta[q] = sum_d dO O computed by the staging threads themselves (shuffle over the row's chunks)
//   phase 1: wave = key tile.  S = Q K^T, dP = dO V^T (MFMA), P = exp2(S' - LSE'), dS = P (dP - delta);
//            dV^T += dO^T P, dK^T += Q^T dcos with the accumulators used directly as B operands; the bf16 dcos tile is
//            written to an LDS image [key][q] (8-byte writes); dK, dV leave through the L2-normalisation backward.
//   phase 2: wave = query tile.  dQ^T = sum_t K_t^T dcos_t^T with both operands as transposed LDS reads -- no atomics,
//            no cross-wave reduction -- then the normalisation backward and a coalesced store.
// The CPB bias (log2 domain, bf16, [key][q]) of the workgroup's head sits in LDS when it fits (BIAS_LDS), its
// gradient accumulates in registers (wave = key tile owns 16 x Lp entries) across all windows of the workgroup.
// TPW = key / query tiles per wave.  TPW = 2 (used without bias): 6 waves own 11 tiles, the Q / dO / statistics
// fragments of a step are read once for both key tiles, and the dcos image is kept compact ([L rounded up to 4 (+ one
// zero block)][Lp + 4]) so that TWO workgroups fit one CU's LDS (2 x 77 KB): 12 waves = 3 per SIMD, and one workgroup's
// barriers / staging overlap the other's MFMA + softmax work.  TPW = 1 is the one-tile-per-wave layout (11 waves).
// AUGP (16- and 32-wide heads, no bias, one key tile per wave; described for 16, the 32-wide variant at `QP` below): the softmax statistics, the padded-key flag and the shift mask ride in
// the unused half of K = 32 MFMA operands, so phase 1 reads no lse / delta rows and has no per-element selects: two LDS reads and
// ~6 vector instructions less per 16 x 16 tile in the kernel's LDS-bound phase.  Construction (head dim 16 fills k = 0 .. 15 of
// the K = 32 operand, the same 16 issue cycles as K = 16):
//   A operand (query side, one 32-slot row per query in the Q / dO slabs, written by the staging threads):
//       Q slab : k 0..15 q^ | k 16..18 lse / (sigma log2 e) as three bf16 parts hi + lo + lolo (|error| <= 2^-24 |v|) | k 19: 1 |
//                k 20, 21: the query's mask-region flags (region 1, region 0) | zeros
//       dO slab: k 0..15 dO | k 16..18 delta = rowsum(dO O) in three parts | zeros
//   B operand (key side, built once per wave): k 0..15 k^ resp. v | k 16..18: -1, -1, -1 | k 19: -1e30 on padded keys (K only) |
//                k 20, 21: the key's mask terms (c if the key is in region 0 resp. 1, with c = -100 log2 e / (sigma log2 e)) | zeros
//   so  S' = Q_aug K_aug^T = cos - lse / (sigma log2 e) + [pad] + [mask]  and  dP' = dO_aug V_aug^T = dP - delta  come out of the
//   matrix pipe, and the softmax backward is  p = exp2(S' sigma log2 e),  dS = p dP'.  The reference's mask (-100 where the query's
//   and the key's regions differ, swinv2_global.py:403-424) is bilinear in the two region flags, hence two slots.
template <int LT, int DK, bool HAS_BIAS, int LFIX, int TPW, bool AUGP = false>
__global__ __launch_bounds__(64 * ((LT + TPW - 1) / TPW)) void attn_bwd_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const float* __restrict__ bias,
    const uint16_t* __restrict__ bimg,     // swv2_attn_pack_bias backward part ([h][Lp][Lp + 4] bf16) or null
    const uint16_t* __restrict__ oh, const uint16_t* __restrict__ doh, const float* __restrict__ lse,
    const float* __restrict__ rnorm,       // [Bw][h][2][Lp]  1/max(|q|,eps), 1/max(|k|,eps)
    uint16_t* __restrict__ dqkvh,          // [Bw][h][3][Lp][DP]  grads w.r.t. the UN-normalised q, k and v
    float* __restrict__ dlogit,            // [h]      (atomically accumulated)
    float* __restrict__ dbias,             // [h][L][L] (atomically accumulated) or null
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr,
    float* __restrict__ dbws) {          // [gridDim.x][h][L][L] partial d bias tables (summed by dbias_reduce_kernel) or null
    using C = AttnCfg<LT, DK>;
    static_assert(TPW == 1 || !HAS_BIAS, "the bias-gradient rows are sized for one key tile per wave");
    constexpr int Lp = C::Lp, DP = C::DP, SLAB = C::SLAB;
    constexpr int WAVES = (LT + TPW - 1) / TPW, NT = 64 * WAVES;
    constexpr int DSP = Lp + 4;                              // row pitch (elements) of the [key][q] bf16 images
    // rows of the dcos image: all Lp keys, or (compile-time L, no bias) the real keys rounded up to a 4-row block plus
    // one block of zeros that stands in for every padded-key block in phase 2
    constexpr bool COMPACT = (LFIX > 0) && !HAS_BIAS && TPW > 1;
    constexpr int IREAL = COMPACT ? (LFIX + 3) / 4 * 4 : Lp;
    constexpr int IROWS = COMPACT ? IREAL + 4 : Lp;
    constexpr bool BIAS_LDS = HAS_BIAS && (LT * DK <= 11);   // 145 KB at LT=11, DK=1
    constexpr int CH = SLAB / 8;                             // 16-byte chunks per slab
    constexpr int CPT = (CH + NT - 1) / NT;                  // chunks per thread
    constexpr int CPR = 2 * DK;                              // chunks per row
    // ONE LDS object with the per-step arrays first: their addresses are then (one running lane offset) + (a constant below
    // 64 KB that fits the ds_read / ds_write offset field).  As separate __shared__ arrays beyond the first 64 KB each
    // access needed its own v_add per step (9 address adds in a ~45-instruction loop).
    // QG: the q / dO slabs do not fit beside K, V and the dcos image (head dim padded to 128 at the 176-token window:
    // 4 x 45 KB + 63 KB) -- their fragments are then read from global memory / L2 (the slab layout in memory is the LDS
    // image), the transposed ones as four 2-byte loads.  A coverage path for wide heads, not a tuned one.
    constexpr bool QG = (4 * SLAB * 2 + IROWS * DSP * 2 + Lp * 8 + 256 > 160 * 1024);
    constexpr bool AUG = AUGP && DK <= 2 && TPW == 1 && !QG && (!HAS_BIAS || (DK == 1 && BIAS_LDS && LFIX > 0));
    // bias image rows (keys): with AUG the padded keys are switched off inside the operand, so the image ends at the last real key
    // (rounded up to 4) and the last wave reads a clamped row -- 5 KB that the 80-byte Q / dO rows need
    constexpr int BROWS = (HAS_BIAS && AUG) ? (LFIX + 3) / 4 * 4 : Lp;
    // row pitch of the q / dO slabs: AUG rows carry 16 more operand slots, padded to 80 bytes (at 64 bytes the 16-byte operand
    // reads of 16 consecutive rows fall on 4 bank groups: conflict cycles 39 % of the LDS index cycles).  32-wide heads (DK = 2):
    // the 32 channels fill the K = 32 operand, the 8 statistics slots follow them in the same 80-byte row and go through a second
    // MFMA (operand = the row's slots 32 .. 39 in every lane group, against a key-side operand that is zero outside k = 0 .. 7).
    constexpr int QP = AUG ? 40 : DP;
    constexpr int QSTAT = AUG ? (DK == 1 ? 16 : 32) : 0;       // first statistics slot of a row
    constexpr int OFF_Q = 0, OFF_DO = OFF_Q + (QG ? 0 : Lp * QP * 2), OFF_LSE = OFF_DO + (QG ? 0 : Lp * QP * 2), OFF_DL = OFF_LSE + Lp * 4,
                  OFF_K = OFF_DL + Lp * 4, OFF_V = OFF_K + SLAB * 2, OFF_DS = OFF_V + SLAB * 2,
                  OFF_BIAS = OFF_DS + IROWS * DSP * 2, OFF_RED = OFF_BIAS + (BIAS_LDS ? BROWS * DSP * 2 : 16),
                  OFF_DLP = OFF_RED + ((WAVES * 4 + 15) / 16) * 16,
                  // chunks per row not a power of two (96 columns = 12 chunks): the per-chunk parts of delta go through LDS
                  LDS_BYTES = OFF_DLP + (((CPR & (CPR - 1)) != 0) ? CH * 4 : 0);
    static_assert(OFF_DS % 16 == 0 && OFF_BIAS % 16 == 0 && OFF_RED % 16 == 0, "16-byte aligned sub-arrays");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const uint16_t* Qs = (const uint16_t*)(lds + OFF_Q);       // QG: re-pointed at the window's global slabs below
    const uint16_t* dOs = (const uint16_t*)(lds + OFF_DO);
    float* const LSEs = (float*)(lds + OFF_LSE);
    float* const DLs = (float*)(lds + OFF_DL);
    uint16_t* const Ks = (uint16_t*)(lds + OFF_K);
    uint16_t* const Vs = (uint16_t*)(lds + OFF_V);
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
    uint16_t* const biasS = (uint16_t*)(lds + OFF_BIAS);
    float* const red = (float*)(lds + OFF_RED);
    constexpr bool CPR_P2 = (CPR & (CPR - 1)) == 0;
    [[maybe_unused]] float* const DLp = (float*)(lds + OFF_DLP);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tw = tid >> 6;                 // this wave owns tiles tw * TPW + i (key tiles in phase 1, query tiles in phase 2)
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    // transposed 16 x 16 fragment: element r of lane (g, fr) = X[row0 + 4g + r][col0 + fr]
    auto tr_frag = [&](const uint16_t* base, int row0, int col0) -> bf16x4 {
        if constexpr (QG) {
            const uint16_t* p_ = base + (size_t)(row0 + 4 * g) * DP + col0 + fr;
            bf16x4 v;
            v[0] = ((const short*)p_)[0]; v[1] = ((const short*)p_)[DP]; v[2] = ((const short*)p_)[2 * DP]; v[3] = ((const short*)p_)[3 * DP];
            return v;
        } else {
            return lds_tr_read(base + (row0 + 4 * g + (fr >> 2)) * DP + col0 + (fr & 3) * 4);
        }
    };

    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E;
    const float inv_sc2 = 1.f / sc2;
    (void)inv_sc2;

    if (COMPACT)
        for (int i = tid; i < 4 * DSP; i += NT) dSb[IREAL * DSP + i] = 0;      // the zero block

    // bias rows: registers (fallback) or LDS image biasS[key][q]; gradient rows always in registers (TPW = 1)
    f32x4 biasr[BIAS_LDS ? 1 : LT];
    f32x4 dbr[HAS_BIAS ? LT : 1];
    if (HAS_BIAS) {
        const int key = 16 * tw + fr;
#pragma unroll
        for (int qt = 0; qt < LT; ++qt) dbr[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (BIAS_LDS && bimg) {
            const uint4* src = (const uint4*)(bimg + (size_t)hd * Lp * DSP);       // Lp * DSP * 2 bytes, multiple of 16
            for (int i = tid; i < BROWS * DSP / 8; i += NT) ((uint4*)biasS)[i] = src[i];
        } else if (BIAS_LDS) {
            // 8 loads in flight per thread (a load -> LDS-write loop exposes one memory round trip per element: 44 trips)
            for (int i0 = 0; i0 < Lp * Lp; i0 += 8 * NT) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * NT + tid, q = i / Lp, k = i - q * Lp;       // k fastest: coalesced global reads
                    const bool in = (i < Lp * Lp) && (k < L) && (q < L);
                    v[u] = bias[in ? ((size_t)hd * L + q) * L + k : (size_t)hd * L * L];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * NT + tid, q = i / Lp, k = i - q * Lp;
                    if (i < Lp * Lp && k < BROWS) biasS[k * DSP + q] = f2bf((k < L) ? ((q < L) ? v[u] * SWV2_LOG2E : 0.f) : SWV2_NEG_BIG);
                }
            }
        } else {
#pragma unroll
            for (int qt = 0; qt < (BIAS_LDS ? 1 : LT); ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    const float b = bias[((size_t)hd * L + min(q, L - 1)) * L + min(key, L - 1)];     // unconditional load
                    const float v = (key < L) ? ((q < L) ? b * SWV2_LOG2E : 0.f) : SWV2_NEG_BIG;
                    biasr[qt][r] = bf2f(f2bf(v));           // bf16-rounded like the LDS image and the forward
                }
        }
    }
    float dsig = 0.f;

    // ---- staging registers: chunk c = tid + j*NT of the q, k, dO, o slabs
    uint4 sq[CPT], sk[CPT], sv[CPT], sdo[CPT], so[CPT];
    float slse = 0.f, slse_row = 0.f;
    static_assert(!AUG || CPT == 1, "AUG: one chunk per thread");
    auto issue = [&](int bw) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB, oslab = ((size_t)bw * h + hd) * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            if (c < CH) {
                sq[j] = *(const uint4*)(qkvh + slab0 + (size_t)c * 8);
                sk[j] = *(const uint4*)(qkvh + slab0 + SLAB + (size_t)c * 8);
                sv[j] = *(const uint4*)(qkvh + slab0 + 2 * SLAB + (size_t)c * 8);
                sdo[j] = *(const uint4*)(doh + oslab + (size_t)c * 8);
                so[j] = *(const uint4*)(oh + oslab + (size_t)c * 8);
            }
        }
        // UNCONDITIONAL (clamped) load; the 1e30 of the padded rows is selected in commit().  As `if (tid < Lp) slse = (tid <
        // L) ? load : 1e30f` the destination register was written on two paths, and the compiler guarded it with
        // s_waitcnt vmcnt(0) RIGHT BEHIND the slab loads above: the waves that stage lse waited here for the whole prefetch
        // (and the previous window's stores) in every window -- 22 % of the kernel (tools/probe_attn1_stamps.py)
        slse = lse[((size_t)bw * h + hd) * Lp + min(tid, Lp - 1)];
        if constexpr (AUG) slse_row = lse[((size_t)bw * h + hd) * Lp + min(tid / CPR, Lp - 1)];      // lse of this thread's chunk row
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            if (c < CH) {
                if constexpr (AUG) {          // rows of 32 / 40 slots: the channels, then the statistics slots written below
                    *(uint4*)((uint16_t*)(lds + OFF_Q) + (c / CPR) * QP + (c % CPR) * 8) = sq[j];
                    *(uint4*)((uint16_t*)(lds + OFF_DO) + (c / CPR) * QP + (c % CPR) * 8) = sdo[j];
                } else {
                    if (!QG) *(uint4*)((uint16_t*)(lds + OFF_Q) + c * 8) = sq[j];
                    if (!QG) *(uint4*)((uint16_t*)(lds + OFF_DO) + c * 8) = sdo[j];
                }
                *(uint4*)(Ks + c * 8) = sk[j];
                *(uint4*)(Vs + c * 8) = sv[j];
            }
            // delta partial over this chunk's 8 channels, reduced over the CPR chunks of the row (adjacent lanes)
            float dl = 0.f;
            if (c < CH) {
                const uint32_t a[4] = {sdo[j].x, sdo[j].y, sdo[j].z, sdo[j].w}, b[4] = {so[j].x, so[j].y, so[j].z, so[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), dl);
                    dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), dl);
                }
            }
            if constexpr (CPR_P2) {
                dl = group_allsum<CPR>(dl);             // (vector ALU only: common.h)
            } else {
                if (c < CH) DLp[c] = dl;            // summed per row by finish_delta() behind the barrier (a row's chunks straddle lane groups)
            }
            if constexpr (AUG) {
                // slots 16..23 of the row (the even chunk's thread): A-operand side of the statistics -- lse / (sigma log2 e) in three
                // bf16 parts, a constant 1 (padded-key flag), the query's mask-region flags -- and delta in three parts for the dO
                // slab; slots 24..31 (the odd chunk's thread): zeros.  The B-operand side (-1, -1, -1, flags) is built per wave.
                if (c < CH) {
                    const int row = c / CPR;
                    uint4 aq = make_uint4(0, 0, 0, 0), ad = make_uint4(0, 0, 0, 0);
                    if ((c % CPR) == 0) {
                        const bool q_ok = row < L;
                        const float lq = q_ok ? slse_row * inv_sc2 : 1.0e30f;            // padded query rows: P = 0
                        uint16_t l0 = f2bf(lq);
                        const float r1 = lq - bf2f(l0);
                        uint16_t l1 = f2bf(r1), l2 = f2bf(r1 - bf2f(l1));
                        if (!q_ok) l1 = l2 = 0;
                        const uint16_t d0 = f2bf(dl);
                        const float e1 = dl - bf2f(d0);
                        const uint16_t d1 = f2bf(e1), d2 = f2bf(e1 - bf2f(d1));
                        const uint32_t one = 0x3f80u, rqf = (row >= mask_thr) ? 0x3f80u : 0u;
                        aq = make_uint4(l0 | ((uint32_t)l1 << 16), l2 | (one << 16), rqf | ((one - rqf) << 16), 0);
                        ad = make_uint4(d0 | ((uint32_t)d1 << 16), d2, 0, 0);
                    }
                    if (DK == 1 || (c % CPR) == 0) {          // (DK = 1: the odd chunk's thread writes the zero slots 24 .. 31)
                        *(uint4*)((uint16_t*)(lds + OFF_Q) + row * QP + QSTAT + (DK == 1 ? (c & 1) * 8 : 0)) = aq;
                        *(uint4*)((uint16_t*)(lds + OFF_DO) + row * QP + QSTAT + (DK == 1 ? (c & 1) * 8 : 0)) = ad;
                    }
                }
            } else if constexpr (CPR_P2) {
                if (c < CH && (c % CPR) == 0) DLs[c / CPR] = dl;
            }
        }
        if (!AUG && tid < Lp) LSEs[tid] = (tid < L) ? slse : 1.0e30f;
    };

    auto finish_delta = [&]() {              // (behind the barrier that follows commit())
        if constexpr (!CPR_P2) {
            if (tid < Lp) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < CPR; ++e) a += DLp[tid * CPR + e];
                DLs[tid] = a;
            }
            __syncthreads();
        }
    };
    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue(bw);
    commit();
    __syncthreads();
    finish_delta();

    const int Lc = LFIX > 0 ? LFIX : L;
    GSTAMP_DECL
    GSTAMP_START();
#ifdef SWV2_ATTN1_STAMPS
    const unsigned long long st_first = st_prev;
#endif
    for (; bw < Bw; bw += gridDim.x) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        const int bw_next = bw + gridDim.x;
        if (bw_next < Bw) issue(bw_next);
        if (QG) { Qs = qkvh + slab0; dOs = doh + ((size_t)bw * h + hd) * SLAB; }

        GSTAMP(0);                      // prefetch issue of the next window
        // ================= phase 1: wave = key tile(s) =================
        bf16x4 kf[TPW][DK], vf[TPW][DK];
        f32x4 dk[TPW][DK], dv[TPW][DK];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int key = 16 * min(tw * TPW + i, LT - 1) + fr;      // a wave's surplus tile repeats the last one (unused)
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                kf[i][kk] = *(const bf16x4*)(Ks + key * DP + 16 * kk + 4 * g);
                vf[i][kk] = *(const bf16x4*)(Vs + key * DP + 16 * kk + 4 * g);   // (a global load here would drain the prefetch)
                dk[i][kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dv[i][kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        // one q-tile step; `br` / `dbrow`: this lane's bias row / bias-gradient row of the tile (TPW = 1 only)
        // MASKED / PADT (shift-mask window / a key tile with padded keys) are wave-uniform and loop-invariant: separate
        // instantiations, so the common case carries no per-element selects
        auto step = [&](const int qt, const f32x4 br, f32x4& dbrow, auto masked_c, auto pad_c) {
            constexpr bool MASKED = decltype(masked_c)::value, PADT = decltype(pad_c)::value;
            bf16x4 qa[DK], da[DK], tq[DK], td[DK];
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                qa[kk] = *(const bf16x4*)(Qs + (16 * qt + fr) * DP + 16 * kk + 4 * g);
                da[kk] = *(const bf16x4*)(dOs + (16 * qt + fr) * DP + 16 * kk + 4 * g);
                td[kk] = tr_frag(dOs, 16 * qt, 16 * kk);
                tq[kk] = tr_frag(Qs, 16 * qt, 16 * kk);
            }
            const f32x4 l4 = *(const f32x4*)(LSEs + 16 * qt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(DLs + 16 * qt + 4 * g);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int kt = tw * TPW + i;
                if (TPW > 1 && kt >= LT) continue;                      // wave-uniform
                const int key = 16 * kt + fr;
                const bool kid = key >= mask_thr;
                const bool key_ok = key < Lc;
                // S = Q K^T and dP = dO V^T : rows q = 16qt + 4g + r, column = key fr
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < DK; ++kk) {
                    s = mfma16(qa[kk], kf[i][kk], s);
                    dp = mfma16(da[kk], vf[i][kk], dp);
                }
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    float x;
                    if (HAS_BIAS) {
                        x = fmaf(s[r], sc2, br[r]) - l4[r];
                    } else {
                        x = fmaf(s[r], sc2, -l4[r]);
                        if (PADT) x = key_ok ? x : SWV2_NEG_BIG;
                    }
                    if (MASKED) x += ((q >= mask_thr) != kid) ? (-100.f * SWV2_LOG2E) : 0.f;
                    const float pr = __builtin_amdgcn_exp2f(x);
                    const float dsr = pr * (dp[r] - d4[r]);
                    p[r] = pr;
                    ds[r] = dsr;                              // dS; d(cos) = sigma * dS is applied once per tile at the end
                    if (HAS_BIAS) dbrow[r] += dsr;
                }
                const uint32_t pk2[2] = {f2bf2(p[0], p[1]), f2bf2(p[2], p[3])};      // explicit pairs: one cvt_pk each
                const bf16x4 pb = __builtin_bit_cast(bf16x4, pk2);
                const bf16x4 dsb = f2bf4(ds);
                if (!COMPACT || key < IREAL) *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;   // image [key][q] for phase 2
                // dV^T += dO^T P ; dK^T += Q^T dcos     (A operands: transposed reads of the staged dO / Q tiles)
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    dv[i][dt] = mfma16(td[dt], pb, dv[i][dt]);
                    dk[i][dt] = mfma16(tq[dt], dsb, dk[i][dt]);
                }
            }
        };
        if constexpr (HAS_BIAS && AUG) {
            // CPB bias with the statistics / padded-key flag / shift mask inside the K = 32 operands (16-wide heads, bias image in LDS):
            // S' = Q_aug K_aug^T and dP' = dO_aug V_aug^T as in the kernel without bias, p = exp2(fma(S', sigma log2 e, b)), dS = p dP'.
            // Against the branch below: two 16-byte operand reads instead of four 8-byte + two statistics reads, no subtractions /
            // selects.  Rolled loop and the scalar switch for the d bias rows as below (44 registers of d bias leave no room for the
            // two-stage pipeline of the kernel without bias).
            const int key = 16 * tw + fr;
            const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;
            bf16x8 kf8, vf8;
            {
                const uint32_t m1 = 0xbf80u;                                       // -1
                const uint32_t padk = (key < Lc) ? 0u : (uint32_t)f2bf(-1.0e30f);
                const bool kreg = key >= mask_thr;
                const uint32_t mk0 = f2bf(kreg ? 0.f : cmask), mk1 = f2bf(kreg ? cmask : 0.f);
                const uint4 augk = make_uint4(m1 | (m1 << 16), m1 | (padk << 16), mk0 | (mk1 << 16), 0);
                const uint4 augv = make_uint4(m1 | (m1 << 16), m1, 0, 0);
                const uint4 z = make_uint4(0, 0, 0, 0);
                const uint4 rk = *(const uint4*)(Ks + key * DP + (g & 1) * 8), rv = *(const uint4*)(Vs + key * DP + (g & 1) * 8);
                kf8 = __builtin_bit_cast(bf16x8, g < 2 ? rk : (g == 2 ? augk : z));
                vf8 = __builtin_bit_cast(bf16x8, g < 2 ? rv : (g == 2 ? augv : z));
            }
            const uint16_t* const Qa = (const uint16_t*)(lds + OFF_Q);
            const uint16_t* const Da = (const uint16_t*)(lds + OFF_DO);
#ifndef SWV2_BIAS_BWD_ROLLED
            // The two-stage software pipeline of the kernel without bias, FULLY UNROLLED so that the d bias rows are statically indexed
            // registers, with a scheduling fence after every pair of steps: the fence keeps the scheduler from hoisting the LDS reads
            // of all eleven steps to the top (which is what made earlier unrolled forms spill), and inside a pair stage A of the next
            // step interleaves with stage B of the current one.  The rolled form below needs a scalar switch to pick the row; the
            // compiler turns it into ~24 64-bit register copies per step.
            struct St { f32x4 s, dp; bf16x4 tq, td, b4; };
            auto stageA = [&](const int qt, St& o) {
                const bf16x8 qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                const bf16x8 da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
                o.td = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.tq = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.b4 = *(const bf16x4*)(biasS + min(key, BROWS - 1) * DSP + 16 * qt + 4 * g);
                o.s = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                o.dp = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
            };
            auto stageB = [&](const int qt, const St& in, f32x4& dbrow) {
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(fmaf(in.s[r], sc2, bf2f(in.b4[r])));
                    ds[r] = p[r] * in.dp[r];
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;
                dv[0][0] = mfma16(in.td, pb, dv[0][0]);
                dk[0][0] = mfma16(in.tq, dsb, dk[0][0]);
                dbrow += ds;
                asm volatile("" : "+v"(dbrow));      // add HERE: otherwise all eleven dS rows are kept live and added behind the loop
            };
            // (q-tiles in pairs with one K = 32 product for dV / dK, as in the kernel without bias below, needs two more stage sets: 39
            // registers spilled around the loop per window, 240 us against 128.)
            St sa, sb;
            SWV2_P1PRIO(3);
            stageA(0, sa);
#pragma unroll
            for (int qt = 0; qt + 1 < LT; qt += 2) {
                stageA(qt + 1, sb);
                stageB(qt, sa, dbr[qt]);
                if (qt + 2 < LT) stageA(qt + 2, sa);
                stageB(qt + 1, sb, dbr[qt + 1]);
                if (qt == 2) SWV2_P1PRIO(2); else if (qt == 4) SWV2_P1PRIO(1); else if (qt == 8) SWV2_P1PRIO(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LT & 1) stageB(LT - 1, sa, dbr[LT - 1]);
            SWV2_P1PRIO(0);
#else
#pragma unroll 1
            for (int qt = 0; qt < LT; ++qt) {
                const bf16x8 qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                const bf16x8 da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
                const bf16x4 td = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                const bf16x4 tq = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                const bf16x4 b4 = *(const bf16x4*)(biasS + min(key, BROWS - 1) * DSP + 16 * qt + 4 * g);
                const f32x4 sv = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                const f32x4 dpv = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc2, bf2f(b4[r])));
                    ds[r] = p[r] * dpv[r];
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;
                dv[0][0] = mfma16(td, pb, dv[0][0]);
                dk[0][0] = mfma16(tq, dsb, dk[0][0]);
#define SWV2_CASE(I) case I: if (I < LT) dbr[I < LT ? I : 0] += ds; break;
                switch (qt) { SWV2_CASE(0) SWV2_CASE(1) SWV2_CASE(2) SWV2_CASE(3) SWV2_CASE(4) SWV2_CASE(5)
                              SWV2_CASE(6) SWV2_CASE(7) SWV2_CASE(8) SWV2_CASE(9) SWV2_CASE(10) }
#undef SWV2_CASE
            }
#endif
        } else if constexpr (HAS_BIAS) {
            const int key = 16 * tw + fr;
            // rolled loop; the bias-gradient rows stay statically indexed registers through a (scalar, wave-uniform)
            // switch on the tile index -- full unrolling costs > 100 extra VGPRs and spills
#pragma unroll 1
            for (int qt = 0; qt < LT; ++qt) {
                f32x4 br;
                if constexpr (BIAS_LDS) {
                    const bf16x4 b4 = *(const bf16x4*)(biasS + key * DSP + 16 * qt + 4 * g);
                    br = (f32x4){bf2f(b4[0]), bf2f(b4[1]), bf2f(b4[2]), bf2f(b4[3])};
                } else {
                    br = (f32x4){0.f, 0.f, 0.f, 0.f};
#define SWV2_CASE(I) case I: if (I < LT) br = biasr[BIAS_LDS ? 0 : (I < LT ? I : 0)]; break;
                    switch (qt) { SWV2_CASE(0) SWV2_CASE(1) SWV2_CASE(2) SWV2_CASE(3) SWV2_CASE(4) SWV2_CASE(5)
                                  SWV2_CASE(6) SWV2_CASE(7) SWV2_CASE(8) SWV2_CASE(9) SWV2_CASE(10) }
#undef SWV2_CASE
                }
                f32x4 dsrow = {0.f, 0.f, 0.f, 0.f};
                if (do_mask) step(qt, br, dsrow, std::true_type{}, std::false_type{});
                else step(qt, br, dsrow, std::false_type{}, std::false_type{});
#define SWV2_CASE(I) case I: if (I < LT) dbr[I < LT ? I : 0] += dsrow; break;
                switch (qt) { SWV2_CASE(0) SWV2_CASE(1) SWV2_CASE(2) SWV2_CASE(3) SWV2_CASE(4) SWV2_CASE(5)
                              SWV2_CASE(6) SWV2_CASE(7) SWV2_CASE(8) SWV2_CASE(9) SWV2_CASE(10) }
#undef SWV2_CASE
            }
        } else if constexpr (AUG) {
            // statistics, padded keys and the shift mask inside the K = 32 operands (construction: the comment above the kernel); the
            // same two-stage software pipeline as below, with two 16-byte operand reads per step instead of two 8-byte fragment
            // reads + two 16-byte statistics reads, and  p = exp2(s * sc2), ds = p * dp  as the whole softmax backward
            struct St { f32x4 s, dp; bf16x4 tq[DK], td[DK]; };
            const int key = 16 * tw + fr;
            const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;
            bf16x8 kf8, vf8;
            [[maybe_unused]] bf16x8 kx8, vx8;                                      // DK = 2: key side of the statistics product
            {
                const uint32_t m1 = 0xbf80u;                                       // -1
                const uint32_t padk = (key < Lc) ? 0u : (uint32_t)f2bf(-1.0e30f);
                const bool kreg = key >= mask_thr;
                const uint32_t mk0 = f2bf(kreg ? 0.f : cmask), mk1 = f2bf(kreg ? cmask : 0.f);
                const uint4 augk = make_uint4(m1 | (m1 << 16), m1 | (padk << 16), mk0 | (mk1 << 16), 0);
                const uint4 augv = make_uint4(m1 | (m1 << 16), m1, 0, 0);
                const uint4 z = make_uint4(0, 0, 0, 0);
                if constexpr (DK == 1) {
                    const uint4 rk = *(const uint4*)(Ks + key * DP + (g & 1) * 8), rv = *(const uint4*)(Vs + key * DP + (g & 1) * 8);
                    kf8 = __builtin_bit_cast(bf16x8, g < 2 ? rk : (g == 2 ? augk : z));
                    vf8 = __builtin_bit_cast(bf16x8, g < 2 ? rv : (g == 2 ? augv : z));
                } else {
                    kf8 = *(const bf16x8*)(Ks + key * DP + 8 * g);
                    vf8 = *(const bf16x8*)(Vs + key * DP + 8 * g);
                    kx8 = __builtin_bit_cast(bf16x8, g == 0 ? augk : z);
                    vx8 = __builtin_bit_cast(bf16x8, g == 0 ? augv : z);
                }
            }
            const uint16_t* const Qa = (const uint16_t*)(lds + OFF_Q);
            const uint16_t* const Da = (const uint16_t*)(lds + OFF_DO);
            auto stageA = [&](const int qt, St& o) {
                const bf16x8 qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                const bf16x8 da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    o.td[dt] = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                    o.tq[dt] = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                }
                if constexpr (DK == 1) {
                    o.s = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                    o.dp = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                } else {
                    // the row's statistics slots in every lane group (one address per row: no bank conflict); the key side is zero for k >= 8
                    const bf16x8 qx = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + QSTAT);
                    const bf16x8 dx = *(const bf16x8*)(Da + (16 * qt + fr) * QP + QSTAT);
                    o.s = mfma32(qa, kf8, mfma32(qx, kx8, (f32x4){0.f, 0.f, 0.f, 0.f}));
                    o.dp = mfma32(da, vf8, mfma32(dx, vx8, (f32x4){0.f, 0.f, 0.f, 0.f}));
                }
            };
            auto stageB = [&](const int qt, const St& in) {
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(in.s[r] * sc2);
                    ds[r] = p[r] * in.dp[r];
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    dv[0][dt] = mfma16(in.td[dt], pb, dv[0][dt]);
                    dk[0][dt] = mfma16(in.tq[dt], dsb, dk[0][dt]);
                }
            };
            if constexpr (DK == 1) {
                // q-tiles in PAIRS: the dV / dK products of two tiles are one K = 32 MFMA (k-slot (g, j) = row 4g + j of the first tile for
                // j < 4, of the second for j >= 4, on both operands) -- 6 MFMAs per pair instead of 8; the odd last tile has its own accumulators
                auto stageB2 = [&](const int qt, const St& i0, const St& i1) {
                    f32x4 p0, p1, ds0, ds1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p0[r] = __builtin_amdgcn_exp2f(i0.s[r] * sc2);
                        p1[r] = __builtin_amdgcn_exp2f(i1.s[r] * sc2);
                        ds0[r] = p0[r] * i0.dp[r];
                        ds1[r] = p1[r] * i1.dp[r];
                    }
                    const bf16x4 pb0 = f2bf4(p0), pb1 = f2bf4(p1), dsb0 = f2bf4(ds0), dsb1 = f2bf4(ds1);
                    *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb0;
                    *(bf16x4*)(dSb + key * DSP + 16 * qt + 16 + 4 * g) = dsb1;
                    const bf16x8 pb = __builtin_shufflevector(pb0, pb1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 dsb = __builtin_shufflevector(dsb0, dsb1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 td = __builtin_shufflevector(i0.td[0], i1.td[0], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 tq = __builtin_shufflevector(i0.tq[0], i1.tq[0], 0, 1, 2, 3, 4, 5, 6, 7);
                    dv[0][0] = mfma32(td, pb, dv[0][0]);
                    dk[0][0] = mfma32(tq, dsb, dk[0][0]);
                };
                St a0, a1, b0, b1;
                SWV2_P1PRIO(3);
                stageA(0, a0);
                stageA(1, a1);
#pragma unroll
                for (int qt = 0; qt + 1 < LT; qt += 4) {
                    if (qt + 2 < LT) stageA(qt + 2, b0);
                    if (qt + 3 < LT) stageA(qt + 3, b1);
                    stageB2(qt, a0, a1);
                    if (qt == 4) SWV2_P1PRIO(1); else if (qt == 8) SWV2_P1PRIO(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (qt + 3 < LT) {
                        if (qt + 4 < LT) stageA(qt + 4, a0);
                        if (qt + 5 < LT) stageA(qt + 5, a1);
                        stageB2(qt + 2, b0, b1);
                        if (qt == 0) SWV2_P1PRIO(2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (LT & 1) {
                    // the odd last tile: its stage A went into a0 (LT % 4 == 1) or b0 (LT % 4 == 3)
                    const St& in = ((LT & 3) == 1) ? a0 : b0;
                    f32x4 p, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[r] = __builtin_amdgcn_exp2f(in.s[r] * sc2);
                        ds[r] = p[r] * in.dp[r];
                    }
                    const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                    *(bf16x4*)(dSb + key * DSP + 16 * (LT - 1) + 4 * g) = dsb;
                    const f32x4 tv = mfma16(in.td[0], pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                    const f32x4 tk = mfma16(in.tq[0], dsb, (f32x4){0.f, 0.f, 0.f, 0.f});
                    dv[0][0] += tv;
                    dk[0][0] += tk;
                }
            } else {
                St sa, sb;
                SWV2_P1PRIO(3);
                stageA(0, sa);
#pragma unroll 1         // (32-wide head slots: unrolled with fences measured equal, 35.0 ms per configs[3] step either way)
                for (int qt = 0; qt + 1 < LT; qt += 2) {
                    stageA(qt + 1, sb);
                    stageB(qt, sa);
                    if (qt + 2 < LT) stageA(qt + 2, sa);
                    stageB(qt + 1, sb);
                    if (4 * qt >= LT) { if (4 * qt >= 3 * LT) SWV2_P1PRIO(0); else if (2 * qt >= LT) SWV2_P1PRIO(1); else SWV2_P1PRIO(2); }
                }
                if (LT & 1) stageB(LT - 1, sa);
                SWV2_P1PRIO(0);
            }
        } else if constexpr (TPW == 1) {
            // software-pipelined over the q tiles (two register sets used alternately): stage A of step qt + 1 (fragment
            // reads, S and dP MFMAs) is issued before stage B of step qt (softmax backward on the vector ALU, dV / dK
            // MFMAs), so a wave has independent work while its MFMA results and LDS reads are in flight.  A plain
            // `#pragma unroll 2` of the one-piece step made the compiler hoist far more and spill (164 -> 203 us).
            struct St { f32x4 s, dp, l4, d4; bf16x4 tq[DK], td[DK]; };
            const int key = 16 * tw + fr;
            const bool kid = key >= mask_thr, key_ok = key < Lc;
            auto stageA = [&](const int qt, St& o) {
                bf16x4 qa[DK], da[DK];
#pragma unroll
                for (int kk = 0; kk < DK; ++kk) {
                    qa[kk] = *(const bf16x4*)(Qs + (16 * qt + fr) * DP + 16 * kk + 4 * g);
                    da[kk] = *(const bf16x4*)(dOs + (16 * qt + fr) * DP + 16 * kk + 4 * g);
                    o.td[kk] = tr_frag(dOs, 16 * qt, 16 * kk);
                    o.tq[kk] = tr_frag(Qs, 16 * qt, 16 * kk);
                }
                o.l4 = *(const f32x4*)(LSEs + 16 * qt + 4 * g);
                o.d4 = *(const f32x4*)(DLs + 16 * qt + 4 * g);
                o.s = (f32x4){0.f, 0.f, 0.f, 0.f};
                o.dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < DK; ++kk) {
                    o.s = mfma16(qa[kk], kf[0][kk], o.s);
                    o.dp = mfma16(da[kk], vf[0][kk], o.dp);
                }
            };
            auto stageB = [&](const int qt, const St& in, auto masked_c, auto pad_c) {
                constexpr bool MASKED = decltype(masked_c)::value, PADT = decltype(pad_c)::value;
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    float x = fmaf(in.s[r], sc2, -in.l4[r]);
                    if (PADT) x = key_ok ? x : SWV2_NEG_BIG;
                    if (MASKED) x += ((q >= mask_thr) != kid) ? (-100.f * SWV2_LOG2E) : 0.f;
                    const float pr = __builtin_amdgcn_exp2f(x);
                    p[r] = pr;
                    ds[r] = pr * (in.dp[r] - in.d4[r]);
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    dv[0][dt] = mfma16(in.td[dt], pb, dv[0][dt]);
                    dk[0][dt] = mfma16(in.tq[dt], dsb, dk[0][dt]);
                }
            };
            auto run = [&](auto masked_c, auto pad_c) {
                St sa, sb;
                stageA(0, sa);
#pragma unroll 1
                for (int qt = 0; qt + 1 < LT; qt += 2) {
                    stageA(qt + 1, sb);
                    stageB(qt, sa, masked_c, pad_c);
                    if (qt + 2 < LT) stageA(qt + 2, sa);
                    stageB(qt + 1, sb, masked_c, pad_c);
                }
                if (LT & 1) stageB(LT - 1, sa, masked_c, pad_c);
            };
            const bool pad_wave = 16 * tw + 16 > Lc;            // padded keys only in the last tile
            if (do_mask) run(std::true_type{}, std::true_type{});
            else if (pad_wave) run(std::false_type{}, std::true_type{});
            else run(std::false_type{}, std::false_type{});
        } else {
            f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
            if (do_mask) {
#pragma unroll 1
                for (int qt = 0; qt < LT; ++qt) step(qt, dummy, dummy, std::true_type{}, std::true_type{});
            } else {
#pragma unroll 1
                for (int qt = 0; qt < LT; ++qt) step(qt, dummy, dummy, std::false_type{}, std::true_type{});
            }
        }
        GSTAMP(1);                      // phase 1 loop
        // ---- dK (through the L2-normalisation) and dV of this wave's key tile(s)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int kt = tw * TPW + i;
            if (TPW > 1 && kt >= LT) continue;
            const int key = 16 * kt + fr;
            const float rk = rnorm[(((size_t)bw * h + hd) * 2 + 1) * Lp + key];
            float dot = 0.f;
#pragma unroll
            for (int dt = 0; dt < DK; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dk[i][dt][r], bf2f(kf[i][dt][r]), dot);
            dot = xor32_allsum(xor16_allsum(dot));
            // d logit_scale: sigma sum_{q,k} dS cos = sigma sum_k (sum_q dS[q][k] q^[q]) . k^[k] = sigma sum_k dot_k -- the same dot
            // product the normalisation backward needs, so the per-element accumulation in the step loop is gone
            if (g == 0) dsig += dot;
            const float rks = rk * sigma;                     // the accumulators hold sum_q q^ dS: d(cos) = sigma dS
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rks * (dk[i][dt][r] - bf2f(kf[i][dt][r]) * dot);
                *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 16 * dt + 4 * g) = f2bf4(v);
                *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 16 * dt + 4 * g) = f2bf4(dv[i][dt]);
            }
        }
        GSTAMP(2);                      // dK / dV normalisation backward + stores
        __syncthreads();
        GSTAMP(3);                      // barrier 1

        // ================= phase 2: wave = query tile(s) =================
        {
            f32x4 dq[TPW][DK];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) dq[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // key tiles in pairs: the two tiles' transposed fragments concatenate to one K = 32 operand (same k order on
            // both sides), which halves the MFMA count and the length of the dependent accumulation chain
            auto frag = [&](int t, bf16x4 (&kt_)[DK], bf16x4 (&ds_)[TPW]) {
                const int row = 16 * t + 4 * g + (fr >> 2);
                // padded-key blocks of the compact image read the zero block
                const int irow = (COMPACT && 16 * t + 4 * g >= IREAL) ? IREAL + (fr >> 2) : row;
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) kt_[dt] = lds_tr_read(Ks + row * DP + 16 * dt + (fr & 3) * 4);   // rows d, col key
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int qt = min(tw * TPW + i, LT - 1);
                    ds_[i] = lds_tr_read(dSb + irow * DSP + 16 * qt + (fr & 3) * 4);                // B[k = key][n = q]
                }
            };
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                bf16x4 k0[DK], k1[DK], d0[TPW], d1[TPW];
                frag(t, k0, d0);
                frag(t + 1, k1, d1);
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int dt = 0; dt < DK; ++dt)
                        dq[i][dt] = mfma32(__builtin_shufflevector(k0[dt], k1[dt], 0, 1, 2, 3, 4, 5, 6, 7),
                                           __builtin_shufflevector(d0[i], d1[i], 0, 1, 2, 3, 4, 5, 6, 7), dq[i][dt]);
            }
            if (LT & 1) {
                bf16x4 k0[DK], d0[TPW];
                frag(LT - 1, k0, d0);
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int dt = 0; dt < DK; ++dt) {
                        // own accumulator: a K = 16 MFMA chained directly onto the K = 32 accumulator gave wrong sums
                        // (measured; the same pair as two K = 16 MFMAs, or with this separate accumulator, is exact)
                        const f32x4 tail = mfma16(k0[dt], d0[i], (f32x4){0.f, 0.f, 0.f, 0.f});
                        dq[i][dt] += tail;
                    }
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int qt = tw * TPW + i;
                if (TPW > 1 && qt >= LT) continue;
                const int q = 16 * qt + fr;
                const float rq = rnorm[(((size_t)bw * h + hd) * 2 + 0) * Lp + q] * sigma;
                bf16x4 qn[DK];
                float dot = 0.f;
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    qn[dt] = *(const bf16x4*)(Qs + q * QP + 16 * dt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][dt][r], bf2f(qn[dt][r]), dot);
                }
                dot = xor32_allsum(xor16_allsum(dot));
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rq * (dq[i][dt][r] - bf2f(qn[dt][r]) * dot);
                    *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 16 * dt + 4 * g) = f2bf4(v);
                }
            }
        }
        GSTAMP(4);                      // phase 2: dQ + normalisation backward + stores
        __syncthreads();
        GSTAMP(5);                      // barrier 2
        if (bw_next < Bw) commit();
        GSTAMP(6);                      // commit (wait for the prefetch + LDS writes + delta)
        __syncthreads();
        if (bw_next < Bw) finish_delta();
        GSTAMP(7);                      // barrier 3
#ifdef SWV2_ATTN1_STAMPS
        if (lane == 0 && tw == 8 && blockIdx.y == 0 && blockIdx.x < 64) {
            const int it_ = (bw - (int)blockIdx.x) / (int)gridDim.x;
            if (it_ < 128) attn1_win[blockIdx.x * 128 + it_] = st_prev - st_first;
        }
#endif
    }
#ifdef SWV2_ATTN1_STAMPS
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x * WAVES + tw < 512)        // every wave of the first workgroups of head 0
        for (int k = 0; k < 8; ++k) attn1_stamps[(blockIdx.x * WAVES + tw) * 8 + k] = st_acc[k];
#endif

    // ---- flush the per-workgroup reductions: one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[tw] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
    if (HAS_BIAS) {
        const int key = 16 * tw + fr;
        if (key < L) {
#pragma unroll
            for (int qt = 0; qt < LT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    if (q < L) {
                        if (dbws) dbws[(((size_t)blockIdx.x * h + hd) * L + q) * L + key] = dbr[qt][r];
                        else atomicAdd(dbias + ((size_t)hd * L + q) * L + key, dbr[qt][r]);
                    }
                }
        }
    }
}

template <int LT, int DK, int LFIX>
int launch_fwd(const swv2_attn_args* a, hipStream_t st) {
    const int nchunk = a->Bw < a->max_chunks ? a->Bw : a->max_chunks;
    dim3 grid(nchunk, a->heads), block(64 * LT);
    const int nW = a->nwh * a->nww;
    if (a->bias)
        hipLaunchKernelGGL((attn_fwd_kernel<LT, DK, true, LFIX>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                           a->bias, (const uint32_t*)a->bias_pack, (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, nW, a->nww,
                           a->nwh, a->mask_thr);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<LT, DK, false, LFIX>), grid, block, 0, st, (const uint16_t*)a->qkvh,
                           a->logit_scale, a->bias, (const uint32_t*)nullptr, (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, nW,
                           a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}

// dbias[i] += sum over the workgroups' partial tables (chunk-major: ws[chunk][heads * L * L]), fixed order
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dbias, int n, int chunks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < chunks; c += 4) {
        s0 += ws[(size_t)c * n + i];
        s1 += ws[(size_t)(c + 1) * n + i];
        s2 += ws[(size_t)(c + 2) * n + i];
        s3 += ws[(size_t)(c + 3) * n + i];
    }
    for (; c < chunks; ++c) s0 += ws[(size_t)c * n + i];
    dbias[i] += (s0 + s1) + (s2 + s3);
}


template <int LT, int DK, int LFIX>
int launch_bwd(const swv2_attn_args* a, hipStream_t st) {
    const int nchunk = a->Bw < a->max_chunks ? a->Bw : a->max_chunks;
    const int nW = a->nwh * a->nww;
    if (a->bias) {
        dim3 grid(nchunk, a->heads), block(64 * LT);
        const uint16_t* bimg = a->bias_pack ? (const uint16_t*)((const char*)a->bias_pack + a->heads * BiasPack<LT>::FWD_U32 * 4)
                                            : nullptr;
        // d bias: every workgroup holds a full [L][L] table of its head.  With a scratch buffer the tables are stored as they
        // are and summed by one more launch (fixed order); without, 31 K float atomics per workgroup (36 us of 214 at the
        // benchmark shape)
        const size_t need = (size_t)nchunk * a->heads * a->L * a->L * sizeof(float);
        float* dbws = (a->dbias_ws && a->dbias_ws_bytes >= need) ? (float*)a->dbias_ws : nullptr;
        // (in FRONT of the launch: with dbias == NULL and a too small workspace the kernel would add into address 0 -- ADVICE r5)
        if (a->dbias_partials) SWV2_CHECK_ARG(dbws != nullptr, "attn_bwd: dbias_partials needs a workspace of %zu bytes", need);
        // 16-wide heads with the bias image in LDS: statistics / mask inside the MFMA operands here too
        if (DK == 1 && !(a->dbg & SWV2_ATTN_PLAIN_STATS))
            hipLaunchKernelGGL((attn_bwd_kernel<LT, DK, true, LFIX, 1, true>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                               a->bias, bimg, (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm,
                               (uint16_t*)a->dqkvh, a->dlogit_scale, a->dbias, a->Bw, a->heads, a->L, nW, a->nww, a->nwh,
                               a->mask_thr, dbws);
        else
        hipLaunchKernelGGL((attn_bwd_kernel<LT, DK, true, LFIX, 1>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                           a->bias, bimg, (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm,
                           (uint16_t*)a->dqkvh, a->dlogit_scale, a->dbias, a->Bw, a->heads, a->L, nW, a->nww, a->nwh,
                           a->mask_thr, dbws);
        if (dbws && !a->dbias_partials) {
            const int n = a->heads * a->L * a->L;
            hipLaunchKernelGGL(dbias_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)dbws, a->dbias, n, nchunk);
        }
    } else {
        // TPW = 2 (6 waves x 2 tiles, two workgroups per CU) measured SLOWER than 11 waves x 1 tile at the benchmark shape
        // (224 us vs 178 us): kept as a template option, not used
        constexpr int TPW = 1;
        dim3 grid(nchunk, a->heads), block(64 * ((LT + TPW - 1) / TPW));
        // 16-wide heads: statistics / mask inside the MFMA operands (SWV2_ATTN_PLAIN_STATS keeps the kernel that reads them from LDS)
        if (DK <= 2 && !(a->dbg & SWV2_ATTN_PLAIN_STATS))
            hipLaunchKernelGGL((attn_bwd_kernel<LT, DK, false, LFIX, TPW, true>), grid, block, 0, st, (const uint16_t*)a->qkvh,
                               a->logit_scale, a->bias, (const uint16_t*)nullptr, (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse,
                               a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, a->dbias, a->Bw, a->heads, a->L, nW, a->nww, a->nwh,
                               a->mask_thr, (float*)nullptr);
        else
        hipLaunchKernelGGL((attn_bwd_kernel<LT, DK, false, LFIX, TPW>), grid, block, 0, st, (const uint16_t*)a->qkvh,
                           a->logit_scale, a->bias, (const uint16_t*)nullptr, (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm,
                           (uint16_t*)a->dqkvh, a->dlogit_scale, a->dbias, a->Bw, a->heads, a->L, nW, a->nww, a->nwh,
                           a->mask_thr, (float*)nullptr);
    }
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}

int check_args(const swv2_attn_args* a, bool bwd) {
    SWV2_CHECK_ARG(a != nullptr, "attn: null args");
    SWV2_CHECK_ARG(a->qkvh && a->oh && a->lse && a->logit_scale, "attn: null tensor pointer");
    SWV2_CHECK_ARG(a->Bw > 0 && a->heads > 0 && a->L > 0 && a->head_dim > 0, "attn: non-positive size");
    SWV2_CHECK_ARG(a->head_dim % 4 == 0, "attn: head_dim %d must be a multiple of 4", a->head_dim);
    SWV2_CHECK_ARG(a->nwh > 0 && a->nww > 0 && a->Bw % (a->nwh * a->nww) == 0,
                   "attn: Bw=%d is not a multiple of the %dx%d windows per sample", a->Bw, a->nwh, a->nww);
    SWV2_CHECK_ARG(a->mask_thr >= 0 && a->mask_thr < a->L, "attn: mask_thr out of range");
    SWV2_CHECK_ARG(a->max_chunks > 0, "attn: max_chunks must be positive");
    if (bwd) SWV2_CHECK_ARG(a->doh && a->rnorm && a->dqkvh && a->dlogit_scale && (!a->bias || a->dbias || (a->dbias_partials && a->dbias_ws)),
                            "attn_bwd: null gradient pointer");
    return SWV2_OK;
}

}  // namespace

// tile geometry shared with the host side (gemm epilogue, python wrappers)
extern "C" int swv2_attn_geometry(int L, int head_dim, int* Lp, int* DP) {
    int LT = 0, DK = 0;
    if (L <= 64) LT = 4; else if (L <= 176) LT = 11;
    // head dims are padded to 16, 32, 64, 96 or 128 columns (the yaml default 768 / 8 = 96 runs unpadded)
    if (head_dim <= 16) DK = 1; else if (head_dim <= 32) DK = 2; else if (head_dim <= 64) DK = 4; else if (head_dim <= 96) DK = 6;
    else if (head_dim <= 128) DK = 8;
    SWV2_CHECK_ARG(LT && DK, "attention: unsupported window area L=%d (<=176) or head_dim=%d (<=128)", L, head_dim);
    if (Lp) *Lp = 16 * LT;
    if (DP) *DP = 16 * DK;
    return SWV2_OK;
}

// kernels are specialised for the window areas of the reference configurations (9x18 = 162 at 720x1440 / ratio 80,
// 6x9 = 54 at 192x288 / ratio 32); any other area <= 176 runs the generic (runtime-L) instantiation
#define SWV2_ATTN_DISPATCH(FN)                                                         \
    int Lp, DP;                                                                        \
    int rc = swv2_attn_geometry(a->L, a->head_dim, &Lp, &DP);                          \
    if (rc) return rc;                                                                 \
    hipStream_t st = (hipStream_t)stream;                                              \
    if (Lp == 176 && DP == 16 && a->L == 162) return FN<11, 1, 162>(a, st);            \
    if (Lp == 64 && DP == 16 && a->L == 54) return FN<4, 1, 54>(a, st);                \
    if (Lp == 64 && DP == 16) return FN<4, 1, 0>(a, st);                               \
    if (Lp == 64 && DP == 32) return FN<4, 2, 0>(a, st);                               \
    if (Lp == 176 && DP == 16) return FN<11, 1, 0>(a, st);                             \
    if (Lp == 176 && DP == 32 && a->L == 162) return FN<11, 2, 162>(a, st);            \
    if (Lp == 176 && DP == 32) return FN<11, 2, 0>(a, st);                             \
    if (Lp == 64 && DP == 64) return FN<4, 4, 0>(a, st);                               \
    if (Lp == 64 && DP == 96) return FN<4, 6, 0>(a, st);                               \
    if (Lp == 176 && DP == 96) return FN<11, 6, 0>(a, st);                             \
    if (Lp == 64 && DP == 128) return FN<4, 8, 0>(a, st);                              \
    if (Lp == 176 && DP == 64) return FN<11, 4, 0>(a, st);                             \
    if (Lp == 176 && DP == 128) return FN<11, 8, 0>(a, st);                            \
    swv2_set_error("attention: no kernel for Lp=%d DP=%d", Lp, DP);                    \
    return SWV2_ERR_UNSUPPORTED;

#ifdef SWV2_ATTN1_STAMPS
extern "C" int swv2_debug_attn1_win(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attn1_win), sizeof(unsigned long long) * 64 * 128) == hipSuccess ? 0 : -3;
}
extern "C" int swv2_debug_attn1_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attn1_stamps), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -3;
}
#endif

extern "C" size_t swv2_attn_dbias_ws_bytes(int heads, int L, int max_chunks) {
    if (heads <= 0 || L <= 0 || max_chunks <= 0) return 0;
    return (size_t)max_chunks * heads * L * L * sizeof(float);
}

extern "C" size_t swv2_attn_pack_bias_bytes(int heads, int L) {
    if (heads <= 0 || L <= 0) return 0;
    return L <= 64 ? BiasPack<4>::bytes(heads) : (L <= 176 ? BiasPack<11>::bytes(heads) : 0);
}

// byte offset of the (max, min) part inside a packed table (attn2.hip's bias forward)
size_t swv2_attn_bias_range_offset(int heads, int L) { return L <= 64 ? BiasPack<4>::range_off(heads) : BiasPack<11>::range_off(heads); }

extern "C" int swv2_attn_pack_bias_multi(const float* bias, int ntab, int heads, int L, void* out, void* stream) {
    SWV2_CHECK_ARG(bias && out && heads > 0 && ntab > 0 && L > 0 && L <= 176, "swv2_attn_pack_bias: null pointer or L=%d out of range", L);
    hipStream_t st = (hipStream_t)stream;
    const size_t pb = swv2_attn_pack_bias_bytes(heads, L);
    if (L <= 64)
        hipLaunchKernelGGL((attn_pack_bias_kernel<4>), dim3(1, heads, ntab), dim3(1024), 0, st, bias, heads, L, (unsigned char*)out, pb);
    else
        hipLaunchKernelGGL((attn_pack_bias_kernel<11>), dim3(1, heads, ntab), dim3(1024), 0, st, bias, heads, L, (unsigned char*)out, pb);
    SWV2_CHECK_LAUNCH("swv2_attn_pack_bias");
    return SWV2_OK;
}

extern "C" int swv2_attn_pack_bias(const float* bias, int heads, int L, void* out, void* stream) {
    return swv2_attn_pack_bias_multi(bias, 1, heads, L, out, stream);
}

// workgroups per head of the attention backward with a CPB table (= d bias tables per head it leaves in swv2_attn_args.dbias_ws)
extern "C" int swv2_attn_bias_chunks(int Bw) { return Bw < 32 ? Bw : 32; }

// the small-workgroup forward of attn2.hip: 0 / negative = handled (ok / error), 1 = shape not covered
int swv2_attn2_fwd(const swv2_attn_args* a, int Lp, int DP, void* stream);

// the kernels for 65 .. 96-channel heads of attn_wide.hip: 0 / negative = handled (ok / error), 1 = shape not covered
int swv2_attn_fwd_wide(const swv2_attn_args* a, int Lp, int DP, void* stream);
int swv2_attn_bwd_wide(const swv2_attn_args* a, int Lp, int DP, void* stream);

// the streamed-dQ backward of attn_bwd_stream.hip (176-row layout, 16-wide head slots, no table): 0 / negative = handled, 1 = not covered
int swv2_attn_bwd_stream(const swv2_attn_args* a, int Lp, int DP, void* stream);

// which forward kernel family serves a geometry (pure host function, no launch): 1 = the operand-folded softmax of attn2.hip, 0 = row
// maximum + exact sum (attn.hip, attn_wide.hip); negative: geometry not covered.  The parity tests declare the regime their oracle
// emulates and check it against this.
extern "C" int swv2_attn_fwd_regime(int L, int head_dim, int has_bias, int dbg) {
    int Lp, DP;
    const int rc = swv2_attn_geometry(L, head_dim, &Lp, &DP);
    if (rc) return rc;
    // (has_bias: a table in its PACKED form, as the model always passes it; a raw table without swv2_attn_pack_bias runs attn.hip)
    static const int fwd3b = getenv("SWV2_ATTN_FWD3B") ? atoi(getenv("SWV2_ATTN_FWD3B")) : 1;
    const bool shape = Lp == 176 && L >= 160 && !(dbg & SWV2_ATTN_FIRST_GEN);
    return (shape && (has_bias ? (DP == 16 && fwd3b) : (DP == 16 || DP == 32))) ? 1 : 0;
}

extern "C" int swv2_attn_fwd(const swv2_attn_args* a, void* stream) {
    int rc0 = check_args(a, false);
    if (rc0) return rc0;
    if (!(a->dbg & SWV2_ATTN_FIRST_GEN)) {
        int Lp2, DP2;
        int rc2 = swv2_attn_geometry(a->L, a->head_dim, &Lp2, &DP2);
        if (rc2) return rc2;
        rc2 = swv2_attn2_fwd(a, Lp2, DP2, stream);
        if (rc2 <= 0) return rc2;
        rc2 = swv2_attn_fwd_wide(a, Lp2, DP2, stream);
        if (rc2 <= 0) return rc2;
    }
    SWV2_ATTN_DISPATCH(launch_fwd)
}

extern "C" int swv2_attn_bwd(const swv2_attn_args* a, void* stream) {
    int rc0 = check_args(a, true);
    if (rc0) return rc0;
    {
        int Lp2, DP2;
        int rc2 = swv2_attn_geometry(a->L, a->head_dim, &Lp2, &DP2);
        if (rc2) return rc2;
        rc2 = swv2_attn_bwd_wide(a, Lp2, DP2, stream);
        if (rc2 <= 0) return rc2;
        rc2 = swv2_attn_bwd_stream(a, Lp2, DP2, stream);
        if (rc2 <= 0) return rc2;
    }
    SWV2_ATTN_DISPATCH(launch_bwd)
}
