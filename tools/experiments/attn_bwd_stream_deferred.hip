// Cosine window attention backward at the benchmark head geometry (176-row layout, 16-wide heads, no CPB table): the dQ phase
// STREAMS behind the dK / dV phase through LDS counters -- one workgroup barrier per window instead of three (gfx950 / CDNA4).
//
// Same semantics, data layout and arithmetic as attn_bwd_kernel<11, 1, false, *, 1, true> of attn.hip (reference
// networks/swinv2_global.py:298-318 under autograd): per (window, head)
//   S' = Q_aug K_aug^T, dP' = dO_aug V_aug^T (statistics / padded keys / shift mask inside the K = 32 operands),
//   P = exp2(S' sigma log2 e), dS = P dP', dV^T += dO^T P, dK^T += Q^T dS, dQ^T = K^T dS^T, then the L2-normalisation backward.
//
// Why a second kernel.  The two-phase kernel is paced by barriers (tools/probe_attn1_stamps.py, LABNOTES round 4 / 5): 11 waves sit
// 3 + 3 + 3 + 2 on the four SIMDs, the SIMD serves its oldest wave first, so waves 0 - 3 end phase 1 a third earlier than waves 8 - 10
// and wait (28 % of their time); then every wave runs a short, latency-bound phase 2 (transposed reads -> a dependent MFMA chain) with
// nothing to overlap it (20 %), then the commit of the next window's slabs between two more barriers (7 %).  Here:
//   * phase 1 is unchanged (wave = key tile, q-tiles in pairs, software-pipelined, fully unrolled with fences); after the dS tiles
//     of a q-tile pair are written, lane 0 of the wave adds 1 to the pair's LDS counter (LDS executes a wave's instructions in order:
//     the add is behind the tile writes);
//   * phase 2 (all 11 q-tiles) belongs to waves 3 and 7 -- the two waves that share a SIMD with no third one (a workgroup's waves go
//     to the SIMDs cyclically: {w, w + 4, w + 8} share one), i.e. the SIMD that idles a third of phase 1 in the two-phase kernel: they
//     run their own phase 1 first (at 2 waves per SIMD they are done at ~60 % of the window), then wave 3 takes q-tile pairs 0, 1 and
//     wave 7 pairs 2, 3, spinning on the pair's counter until it shows all 11 waves, then reading the dS image transposed (K^T fragments
//     once per pair, two interleaved accumulation chains).  The last pair and the odd tile go to waves 1 and 0, the oldest waves of two
//     other SIMDs, which are done first and would otherwise wait at the barrier.  (All 11 tiles on waves 3 / 7, one tile at a time:
//     128 us against 112 -- a tile's chain of transposed reads -> 6 dependent MFMAs -> row sum -> store is ~400 cycles of latency;
//     one tile per wave inside every wave's own loop: 110, the old waves' dQ + commit then sit behind the last signal);
//   * the q | dO | k | v slabs (and the 1 / |q|, 1 / |k| rows) are double-buffered in LDS, so the next window's slabs are committed
//     from the prefetch registers before the window's ONE barrier; the dS image is single (every reader has passed the barrier
//     before the next window's first tile is written).
// LDS: 2 x 40 832 (slabs) + 63 360 (dS image) + counters = 145 KB, one persistent workgroup per CU as before.
#include <stdlib.h>

#include "attn_common.h"

namespace {

typedef __attribute__((address_space(3))) unsigned lds_u32;

#ifdef SWV2_ATTNS_STAMPS          // diagnostic build (tools/probe_attn_stream_stamps.py): per-phase s_memtime sums of every wave
__device__ unsigned long long attns_stamps[512 * 8];
__device__ unsigned long long attns_win[64 * 128];        // wave 8 of the first 64 workgroups of head 0: s_memtime at the end of every window
__device__ unsigned long long attns_clock[512 * 2];       // per wave: s_memtime span, s_memrealtime span (100 MHz) of the window loop
#define SSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define SSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define SSTAMP_DECL
#define SSTAMP_START() do {} while (0)
#define SSTAMP(k) do {} while (0)
#endif

// issue-priority scheme (A/B builds): 0 = none; 1 = phase-1 waves 3 -> 0 with their progress, helpers 0; 2 = phase-1 waves 2 -> 0, helpers 3
#ifdef SWV2_ATTNS_NO_PRIO
#define SWV2_ATTNS_PRIO_MODE 0
#endif
#ifndef SWV2_ATTNS_PRIO_MODE
#define SWV2_ATTNS_PRIO_MODE 2
#endif
#define SWV2_PRIO(n) __builtin_amdgcn_s_setprio(n)
// helpers from this index on commit BEFORE their dQ pair (their pair completes late in the window; the prefetch has landed by then)
#ifndef SWV2_ATTNS_COMMIT_FIRST
#define SWV2_ATTNS_COMMIT_FIRST 99
#endif

template <int LFIX>
__global__ __launch_bounds__(1024) void attn_bwd_stream_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint16_t* __restrict__ oh,
    const uint16_t* __restrict__ doh, const float* __restrict__ lse,
    const float* __restrict__ rnorm,       // [Bw][h][2][Lp]  1/max(|q|,eps), 1/max(|k|,eps)
    uint16_t* __restrict__ dqkvh,          // [Bw][h][3][Lp][DP]  grads w.r.t. the UN-normalised q, k and v
    float* __restrict__ dlogit,            // [h]      (atomically accumulated)
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int LT = 11, Lp = 16 * LT, DP = 16, SLAB = Lp * DP;
    constexpr int PW = LT, HW = 5, WAVES = PW + HW;          // 11 phase-1 waves (wave = key tile) + 5 helper waves (staging, commit, dQ): 4 per SIMD
    constexpr int DSP = Lp + 4;                              // row pitch (elements) of the [key][q] bf16 dS image
    constexpr int QP = 40, QSTAT = 16;                       // 80-byte q / dO rows: 16 channels, 8 statistics slots, 16 bytes of padding (bank spread)
    constexpr int CH = SLAB / 8, CPR = 2;                    // 16-byte chunks per slab / per row
    constexpr int NPAIR = (LT + 1) / 2;                      // q-tile pairs (the odd last tile is a "pair" of its own)
    constexpr int HT = 64 * HW;                              // helper threads (320): chunk c = their index, the first 32 of the last helper also chunk 320 + lane
    static_assert(CH > HT && CH <= HT + 32, "second chunks on the first half of one helper wave");
    // one buffer: q rows | dO rows | k | v | 1/|q|, 1/|k|
    constexpr int B_Q = 0, B_DO = B_Q + Lp * QP * 2, B_K = B_DO + Lp * QP * 2, B_V = B_K + SLAB * 2, B_RN = B_V + SLAB * 2,
                  BUFB = B_RN + 2 * Lp * 4;
    // dS image [key][q]: q-tiles 0 .. 7 (pairs 0 .. 3) always in the main image; the LAST three q-tiles (pair 4 and the odd tile) of even windows in
    // its columns 128 .. 175, of odd windows in a buffer of their own ([176][48 + 4]): their dQ is computed during the NEXT window
    constexpr int TQ0 = 8, TCOL = 16 * TQ0, XTP = 16 * (LT - TQ0) + 4;      // first deferred q-tile, its column, row pitch of the extra buffer
    constexpr int OFF_DS = 2 * BUFB, OFF_XT = OFF_DS + Lp * DSP * 2, OFF_CNT = OFF_XT + Lp * XTP * 2, OFF_RED = OFF_CNT + 32,
                  LDS_BYTES = OFF_RED + ((WAVES * 4 + 15) / 16) * 16;
    static_assert(BUFB % 16 == 0 && OFF_DS % 16 == 0 && OFF_CNT % 16 == 0, "16-byte aligned sub-arrays");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
    uint16_t* const dXt = (uint16_t*)(lds + OFF_XT);
    float* const red = (float*)(lds + OFF_RED);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);   // phase-1 waves: the key tile; helpers: PW + helper index
    const int hd = blockIdx.y;
    const bool helper = tw >= PW;                              // wave-uniform
    const int hw = tw - PW;

    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E;
    const float inv_sc2 = 1.f / sc2;
    const int Lc = LFIX > 0 ? LFIX : L;

    if (tid < 8) ((unsigned*)(lds + OFF_CNT))[tid] = 0u;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u32*)lds;
    const unsigned cnt_addr = lds_base + OFF_CNT;           // the counters' LDS address (for the inline ds_add / ds_read)

    // ---- staging (helper waves): one 16-byte chunk of the q, k, dO, o slabs; lse of the chunk's row; one 1/|.| value -- twice on the last helper
    // (v goes straight to LDS by DMA: 44 staging registers beside the dQ pair did not fit the 128 of a 16-wave workgroup.  k does not: the
    // deferred dQ of the previous window still reads that buffer's K when the prefetch is issued; it is written at the commit like q)
    // (plain variables captured by reference, one set per chunk, bodies from a macro: as members of a struct -- or as arrays -- handed to a shared
    // lambda, the dO / o registers were kept in scratch memory, i.e. stored by VMEM operations that wait for the loads on the spot)
    uint4 s0q, s0k, s0d, s0o, s1q, s1k, s1d, s1o;
    float s0l = 0.f, s0r = 0.f, s1l = 0.f, s1r = 0.f;
    const bool two = helper && hw == HW - 1;                   // wave-uniform: this wave stages a second chunk (its lanes 0 .. 31)
    // 32-bit, loop-invariant lane offsets against wave-uniform bases: the loads take the (SGPR base + VGPR offset) form.  With 64-bit
    // per-lane addresses the compiler builds them in the loads' own destination registers and guards that overwrite with s_waitcnt
    // vmcnt(..) -- which, the counter being in order, also waits for earlier d(qkv) STORES (ISA, attn2.hip round 2)
    // v: LDS-DMA (M0 = the LDS address of the wave's first chunk, lane l lands 16 l bytes behind it), issued BEFORE the register loads: the
    // counter is in order, so the compiler's own wait for a younger register load covers it
#define SWV2_STG_ISSUE(Q, K, D, O, LS, RNV, CRAW, LDSV)                                                                                   \
    do {                                                                                                                                  \
        const int c_ = min((CRAW), CH - 1);                                                                                               \
        unsigned o16 = (unsigned)c_ * 16u, ol = (unsigned)(c_ / CPR) * 4u, orn = (unsigned)c_ * 4u;                                       \
        asm volatile("" : "+v"(o16), "+v"(ol), "+v"(orn)); /* opaque per call: else LICM builds 64-bit per-lane pointers outside the loop */ \
        if ((CRAW) < CH) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(o16), "s"(vb_), "s"(LDSV) : "memory"); \
        Q = *(const uint4*)(qb + o16);                                                                                                    \
        K = *(const uint4*)(kb_ + o16);                                                                                                   \
        D = *(const uint4*)(dob + o16);                                                                                                   \
        O = *(const uint4*)(ob + o16);                                                                                                    \
        LS = *(const float*)((const char*)(lse + item * Lp) + ol);                                                                        \
        RNV = *(const float*)((const char*)(rnorm + item * 2 * Lp) + orn);                                                                \
    } while (0)
    auto issue = [&](int bw, int nbuf, int hidx, auto two_c) {      // nbuf: the LDS buffer the window will be committed to; hidx: the helper thread's index
        const size_t item = (size_t)__builtin_amdgcn_readfirstlane(bw) * h + hd;
        const char* const qb = (const char*)(qkvh + item * 3 * SLAB);
        // k / v bases of their own: 5 632 / 11 264 do not fit the loads' offset field, and as known constants they are split into a
        // per-lane 64-bit add + a small offset
        unsigned kofs = SLAB * 2, vofs = 2 * SLAB * 2;
        asm volatile("" : "+s"(kofs), "+s"(vofs));
        const char* const kb_ = qb + kofs;
        const char* const vb_ = qb + vofs;
        const char* const dob = (const char*)(doh + item * SLAB);
        const char* const ob = (const char*)(oh + item * SLAB);
        const unsigned lds_v0 = lds_base + (unsigned)(nbuf * BUFB + B_V);
        SWV2_STG_ISSUE(s0q, s0k, s0d, s0o, s0l, s0r, hidx, lds_v0 + (unsigned)hw * 1024u);
        if constexpr (decltype(two_c)::value) SWV2_STG_ISSUE(s1q, s1k, s1d, s1o, s1l, s1r, HT + (hidx & 63), lds_v0 + (unsigned)HT * 16u);
    };
#define SWV2_STG_COMMIT(Q, K, D, O, LS, RNV, CRAW)                                                                                        \
    do {                                                                                                                                  \
        const int c_ = min((CRAW), CH - 1);                                                                                               \
        const int row = c_ / CPR, half = c_ % CPR;                                                                                        \
        /* delta partial over this chunk's 8 channels, reduced over the 2 chunks of the row (adjacent lanes; vector ALU only) */          \
        float dl = 0.f;                                                                                                                   \
        dl = fmaf(__uint_as_float(D.x << 16), __uint_as_float(O.x << 16), dl);                                                            \
        dl = fmaf(__uint_as_float(D.x & 0xffff0000u), __uint_as_float(O.x & 0xffff0000u), dl);                                            \
        dl = fmaf(__uint_as_float(D.y << 16), __uint_as_float(O.y << 16), dl);                                                            \
        dl = fmaf(__uint_as_float(D.y & 0xffff0000u), __uint_as_float(O.y & 0xffff0000u), dl);                                            \
        dl = fmaf(__uint_as_float(D.z << 16), __uint_as_float(O.z << 16), dl);                                                            \
        dl = fmaf(__uint_as_float(D.z & 0xffff0000u), __uint_as_float(O.z & 0xffff0000u), dl);                                            \
        dl = fmaf(__uint_as_float(D.w << 16), __uint_as_float(O.w << 16), dl);                                                            \
        dl = fmaf(__uint_as_float(D.w & 0xffff0000u), __uint_as_float(O.w & 0xffff0000u), dl);                                            \
        dl = group_allsum<CPR>(dl);                                                                                                       \
        /* slots 16..23 of the row (the even chunk's thread): lse / (sigma log2 e) in three bf16 parts, a constant 1 (padded-key flag), */ \
        /* the query's mask-region flags -- and delta in three parts for the dO row; slots 24..31 (odd chunk): zeros */                   \
        uint4 aq = make_uint4(0, 0, 0, 0), ad = make_uint4(0, 0, 0, 0);                                                                   \
        if (half == 0) {                                                                                                                  \
            const bool q_ok = row < L;                                                                                                    \
            const float lq = q_ok ? LS * inv_sc2 : 1.0e30f; /* padded query rows: P = 0 */                                                \
            uint16_t l0 = f2bf(lq);                                                                                                       \
            const float r1 = lq - bf2f(l0);                                                                                               \
            uint16_t l1 = f2bf(r1), l2 = f2bf(r1 - bf2f(l1));                                                                             \
            if (!q_ok) l1 = l2 = 0;                                                                                                       \
            const uint16_t d0 = f2bf(dl);                                                                                                 \
            const float e1 = dl - bf2f(d0);                                                                                               \
            const uint16_t d1 = f2bf(e1), d2 = f2bf(e1 - bf2f(d1));                                                                       \
            const uint32_t one_ = 0x3f80u, rqf = (row >= mask_thr) ? 0x3f80u : 0u;                                                        \
            aq = make_uint4(l0 | ((uint32_t)l1 << 16), l2 | (one_ << 16), rqf | ((one_ - rqf) << 16), 0);                                 \
            ad = make_uint4(d0 | ((uint32_t)d1 << 16), d2, 0, 0);                                                                         \
        }                                                                                                                                 \
        if ((CRAW) < CH) {                                                                                                                \
            *(uint4*)((uint16_t*)(B + B_Q) + row * QP + half * 8) = Q;                                                                    \
            *(uint4*)((uint16_t*)(B + B_DO) + row * QP + half * 8) = D;                                                                   \
            *(uint4*)((uint16_t*)(B + B_K) + c_ * 8) = K;                                                                                 \
            *(uint4*)((uint16_t*)(B + B_Q) + row * QP + QSTAT + half * 8) = aq;                                                           \
            *(uint4*)((uint16_t*)(B + B_DO) + row * QP + QSTAT + half * 8) = ad;                                                          \
            ((float*)(B + B_RN))[c_] = RNV;                                                                                               \
        }                                                                                                                                 \
    } while (0)
    auto commit = [&](int buf, int hidx, auto two_c) {
        unsigned char* const B = lds + buf * BUFB;
        SWV2_STG_COMMIT(s0q, s0k, s0d, s0o, s0l, s0r, hidx);
        if constexpr (decltype(two_c)::value) SWV2_STG_COMMIT(s1q, s1k, s1d, s1o, s1l, s1r, HT + (hidx & 63));
    };
    using T1 = std::false_type;
    using T2 = std::true_type;

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    if (helper) {
        // (the second chunk's registers live only inside the last helper's own branch: as one `if (two)` inside shared code they were live across
        // every helper's dQ pair and two of them spilled)
        if (two) { issue(bw, 0, hw * 64 + lane, T2{}); commit(0, hw * 64 + lane, T2{}); }
        else { issue(bw, 0, hw * 64 + lane, T1{}); commit(0, hw * 64 + lane, T1{}); }
    }
    __syncthreads();

    // ================= phase 2 (helper waves): dQ of NQ q-tiles starting at token q0 =================
    // dQ^T = sum_t K_t^T dS_t^T, key tiles in pairs (one K = 32 product per pair), both operands as transposed reads -- the K^T fragments are
    // read once for both q-tiles, whose accumulation chains interleave; one chain per q-tile in key order + the odd key tile on its own
    // accumulator: the summation order of the two-phase kernel (bit-identical d q).  img: the tiles' first column in a dS image of row
    // pitch PITCH; Bf: the window's slab buffer; waits until counter `slot` shows `target` (all phase-1 waves have written the tiles).
    auto spin_until = [&](const unsigned slot, const unsigned target) {
        const unsigned a = cnt_addr + 4u * slot;
        while (true) {
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            if ((unsigned)__builtin_amdgcn_readfirstlane(v) >= target) break;
            __builtin_amdgcn_s_sleep(4);
        }
    };
    constexpr unsigned DONE_SLOT = 6;
    auto done_signal = [&](const int ln_) {
        if (ln_ == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_addr + 4u * DONE_SLOT), "v"(1u) : "memory");
    };
    auto done_wait = [&](const unsigned target) { spin_until(DONE_SLOT, target); };
    SSTAMP_DECL
    auto phase2 = [&](const unsigned slot, const unsigned target, auto nq_c, const int q0, const uint16_t* img, auto pitch_c,
                      const unsigned char* Bf, const size_t slab, const int fr, const int g) {
        constexpr int NQ = decltype(nq_c)::value, PITCH = decltype(pitch_c)::value;
        const uint16_t* const Ks = (const uint16_t*)(Bf + B_K);
        const uint16_t* const Qa = (const uint16_t*)(Bf + B_Q);
        const float* const RN = (const float*)(Bf + B_RN);
        spin_until(slot, target);
        SSTAMP(2);
        f32x4 dq[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint16_t* const kb = Ks + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4;
        const uint16_t* const db = img + (4 * g + (fr >> 2)) * PITCH + (fr & 3) * 4;
#pragma unroll
        for (int t = 0; t + 1 < LT; t += 2) {
            const bf16x4 k0 = lds_tr_read(kb + 16 * t * DP), k1 = lds_tr_read(kb + 16 * (t + 1) * DP);
            const bf16x8 ka = __builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const bf16x4 d0 = lds_tr_read(db + 16 * t * PITCH + 16 * i), d1 = lds_tr_read(db + 16 * (t + 1) * PITCH + 16 * i);
                dq[i] = mfma32(ka, __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7), dq[i]);
            }
            // (a fence every second step bounds the transposed reads in flight: unfenced, the pair's reads beside a chunk of staging registers
            // exceeded the 128 registers of the 16-wave workgroup by four)
            if (NQ == 2 && (t & 2)) __builtin_amdgcn_sched_barrier(0);
        }
        {
            // own accumulator: a K = 16 MFMA chained directly onto a K = 32 accumulator gave wrong sums (attn.hip)
            const bf16x4 k0 = lds_tr_read(kb + 16 * (LT - 1) * DP);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const bf16x4 d0 = lds_tr_read(db + 16 * (LT - 1) * PITCH + 16 * i);
                const f32x4 tail = mfma16(k0, d0, (f32x4){0.f, 0.f, 0.f, 0.f});
                dq[i] += tail;
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = q0 + 16 * i + fr;
            const float rq = RN[q] * sigma;
            const bf16x4 qn = *(const bf16x4*)(Qa + q * QP + 4 * g);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][r], bf2f(qn[r]), dot);
            dot = xor32_allsum(xor16_allsum(dot));
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rq * (dq[i][r] - bf2f(qn[r]) * dot);
            *(bf16x4*)(dqkvh + slab + (size_t)q * DP + 4 * g) = f2bf4(v);
        }
        SSTAMP(3);
    };

    float dsig = 0.f;
    SSTAMP_START();
#ifdef SWV2_ATTNS_STAMPS
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        const int bw_next = bw + gridDim.x;
        const unsigned char* const B = lds + buf * BUFB;
        const uint16_t* const Qa = (const uint16_t*)(B + B_Q);
        const uint16_t* const Da = (const uint16_t*)(B + B_DO);
        const uint16_t* const Ks = (const uint16_t*)(B + B_K);
        const uint16_t* const Vs = (const uint16_t*)(B + B_V);
        const float* const RN = (const float*)(B + B_RN);
        const unsigned target = (unsigned)(PW * (it + 1));

        // (the lane id is re-derived behind an opaque asm in each role's branch: otherwise loop-invariant code motion hoists every
        // lane-dependent address of BOTH roles in front of the window loop and spills them -- 53 registers, reloaded by VMEM operations)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, g = ln >> 4;
        if (!helper) {
            // ================= phase 1: wave = key tile =================
            const int key = 16 * tw + fr;
            const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
            const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;
            const bf16x4 kf = *(const bf16x4*)(Ks + key * DP + 4 * g);          // (the normalisation backward's copy of k^)
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
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            // where this window's deferred dS tiles (q-tiles 8 .. 10) of this lane's key go
            uint16_t* const tailp = (buf ? dXt + key * XTP : dSb + key * DSP + TCOL) + 4 * g;

            struct St { f32x4 s, dp; bf16x4 tq, td; };
            auto stageA = [&](const int qt, St& o) {
                const bf16x8 qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                const bf16x8 da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
                o.td = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.tq = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.s = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                o.dp = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
            };
            // the dS tiles of q-tile pair `p` of this wave are in the image: tell the dQ side (the LDS runs a wave's instructions in order)
            auto signal = [&](const int p) {
                if (ln == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_addr + 4u * (unsigned)p), "v"(1u) : "memory");
            };
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
                if (qt >= TQ0) {
                    *(bf16x4*)(tailp + 16 * (qt - TQ0)) = dsb0;
                    *(bf16x4*)(tailp + 16 * (qt - TQ0) + 16) = dsb1;
                } else {
                    *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb0;
                    *(bf16x4*)(dSb + key * DSP + 16 * qt + 16 + 4 * g) = dsb1;
                }
                const bf16x8 pb = __builtin_shufflevector(pb0, pb1, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 dsb = __builtin_shufflevector(dsb0, dsb1, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 td = __builtin_shufflevector(i0.td, i1.td, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 tq = __builtin_shufflevector(i0.tq, i1.tq, 0, 1, 2, 3, 4, 5, 6, 7);
                dv = mfma32(td, pb, dv);
                dk = mfma32(tq, dsb, dk);
            };
            // Issue priority falls with the wave's own progress (2 at the start of the window, 1 behind the second pair, 0 behind the fourth): the
            // SIMD arbiter serves the oldest wave of the highest priority first, so without this the oldest wave of a SIMD runs ahead, ends at
            // ~60 % of the window, and the youngest finishes alone with nothing to cover its latencies (same box: 106.0 / 103.4 us without,
            // 101.1 / 99.3 with).  A wave that is behind now outranks one that is ahead.  The helper waves run at priority 3: they issue
            // little and sleep most of the time, but what they issue gates the window's end (at priority 0 their ~40 prefetch instructions
            // took a third of the window and the phase-1 waves waited 18 - 27 % of their time at the barrier).
            constexpr int PM = SWV2_ATTNS_PRIO_MODE;
            if (PM == 1) SWV2_PRIO(3); else if (PM == 2) SWV2_PRIO(2);
            St a0, a1, b0, b1;
            stageA(0, a0);
            stageA(1, a1);
#pragma unroll
            for (int qt = 0; qt + 1 < LT; qt += 4) {
                if (qt + 2 < LT) stageA(qt + 2, b0);
                if (qt + 3 < LT) stageA(qt + 3, b1);
                stageB2(qt, a0, a1);
                signal(qt >> 1);
                if (PM == 1) { if (qt == 4) SWV2_PRIO(1); else if (qt == 8) SWV2_PRIO(0); }
                __builtin_amdgcn_sched_barrier(0);
                if (qt + 3 < LT) {
                    if (qt + 4 < LT) stageA(qt + 4, a0);
                    if (qt + 5 < LT) stageA(qt + 5, a1);
                    stageB2(qt + 2, b0, b1);
                    signal((qt >> 1) + 1);
                    if (PM == 1) { if (qt == 0) SWV2_PRIO(2); }
                    if (PM == 2) { if (qt == 0) SWV2_PRIO(1); else if (qt == 4) SWV2_PRIO(0); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            static_assert((LT & 3) == 3, "the odd last tile's stage A went into b0");
            {
                const St& in = b0;
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(in.s[r] * sc2);
                    ds[r] = p[r] * in.dp[r];
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(tailp + 16 * (LT - 1 - TQ0)) = dsb;
                const f32x4 tv = mfma16(in.td, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                const f32x4 tk = mfma16(in.tq, dsb, (f32x4){0.f, 0.f, 0.f, 0.f});
                dv += tv;
                dk += tk;
                signal(NPAIR - 1);
            }
            SSTAMP(1);                      // phase 1 loop
            // ---- dK (through the L2-normalisation) and dV of this wave's key tile
            {
                const float rk = RN[Lp + key];
                float dot = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dk[r], bf2f(kf[r]), dot);
                dot = xor32_allsum(xor16_allsum(dot));
                // d logit_scale: sigma sum_{q,k} dS cos = sigma sum_k (sum_q dS[q][k] q^[q]) . k^[k] = sigma sum_k dot_k
                if (g == 0) dsig += dot;
                const float rks = rk * sigma;                     // the accumulators hold sum_q q^ dS: d(cos) = sigma dS
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rks * (dk[r] - bf2f(kf[r]) * dot);
                *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 4 * g) = f2bf4(v);
                *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 4 * g) = f2bf4(dv);
            }
            SSTAMP(4);                      // dK / dV normalisation backward + stores
        } else {
            // ================= helper waves: the next window's prefetch, phase 2 (dQ), the commit =================
            if (SWV2_ATTNS_PRIO_MODE == 2) SWV2_PRIO(3);
            const int hidx = hw * 64 + ln;
            const unsigned char* const Bp = lds + (buf ^ 1) * BUFB;        // the previous window's buffer (its deferred dQ reads k, q, 1/|q| there)
            const size_t slab0p = slab0 - (size_t)gridDim.x * h * 3 * SLAB;
            // helper hw < 4: pair hw of this window, then the commit (the prefetch has landed by then).  Helpers 0 .. 2 first take one of the last
            // three q-tiles of the PREVIOUS window (10, 8, 9: their dS tiles were complete before the barrier).  The commit overwrites the buffer
            // those three read: every helper waits for their done counter in front of it.  Helper 4 stages two chunks and computes nothing (two
            // chunks of staging registers beside a dQ pass spilled, and a spilled staging register is waited for at the prefetch itself).
            constexpr unsigned NDEF = 3;
            if (hw == HW - 1) {
                if (bw_next < Bw) {
                    issue(bw_next, buf ^ 1, hidx, T2{});
                    SSTAMP(0);
                    done_wait(NDEF * (unsigned)(it + 1));
                    commit(buf ^ 1, hidx, T2{});
                }
            } else {
                if (bw_next < Bw) issue(bw_next, buf ^ 1, hidx, T1{});
                SSTAMP(0);
                if (hw < (int)NDEF) {
                    if (it > 0) {
                        const int dqt = (hw == 0) ? LT - 1 : TQ0 + hw - 1;        // helper 0: tile 10 (counter 5); helpers 1, 2: tiles 8, 9 (counter 4)
                        const unsigned dslot = (hw == 0) ? NPAIR - 1 : NPAIR - 2;
                        if (buf) phase2(dslot, PW * it, std::integral_constant<int, 1>{}, 16 * dqt, dSb + 16 * dqt, std::integral_constant<int, DSP>{}, Bp, slab0p, fr, g);
                        else phase2(dslot, PW * it, std::integral_constant<int, 1>{}, 16 * dqt, dXt + 16 * (dqt - TQ0), std::integral_constant<int, XTP>{}, Bp, slab0p, fr, g);
                    }
                    done_signal(ln);
                }
                const bool cfirst = hw >= SWV2_ATTNS_COMMIT_FIRST;
                if (cfirst && bw_next < Bw) {
                    done_wait(NDEF * (unsigned)(it + 1));
                    commit(buf ^ 1, hidx, T1{});
                }
                phase2(hw, target, std::integral_constant<int, 2>{}, 32 * hw, dSb + 32 * hw, std::integral_constant<int, DSP>{}, B, slab0, fr, g);
                if (!cfirst && bw_next < Bw) {
                    done_wait(NDEF * (unsigned)(it + 1));
                    commit(buf ^ 1, hidx, T1{});
                }
            }
            SSTAMP(5);
        }
        __syncthreads();
        SSTAMP(6);                      // the window's barrier
#ifdef SWV2_ATTNS_STAMPS
        if (lane == 0 && tw == 8 && blockIdx.y == 0 && blockIdx.x < 64 && it < 128) attns_win[blockIdx.x * 128 + it] = st_prev - ck0;
#endif
    }
    // ---- the last window's deferred q-tiles (behind its barrier: every tile is in place)
    if (helper && hw < 3) {
        const int nwin = (Bw - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // windows of this workgroup (>= 1)
        const int lb = (nwin - 1) & 1;
        const unsigned char* const Bl = lds + lb * BUFB;
        const size_t slabl = ((size_t)(bw - (int)gridDim.x) * h + hd) * 3 * SLAB;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, g = ln >> 4;
        const unsigned tl = (unsigned)(PW * nwin);
        const int dqt = (hw == 0) ? LT - 1 : TQ0 + hw - 1;
        const unsigned dslot = (hw == 0) ? NPAIR - 1 : NPAIR - 2;
        if (lb) phase2(dslot, tl, std::integral_constant<int, 1>{}, 16 * dqt, dXt + 16 * (dqt - TQ0), std::integral_constant<int, XTP>{}, Bl, slabl, fr, g);
        else phase2(dslot, tl, std::integral_constant<int, 1>{}, 16 * dqt, dSb + 16 * dqt, std::integral_constant<int, DSP>{}, Bl, slabl, fr, g);
    }
#ifdef SWV2_ATTNS_STAMPS
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x * WAVES + tw < 512) {      // every wave of the first workgroups of head 0
        for (int k = 0; k < 8; ++k) attns_stamps[(blockIdx.x * WAVES + tw) * 8 + k] = st_acc[k];
        attns_clock[(blockIdx.x * WAVES + tw) * 2] = __builtin_amdgcn_s_memtime() - ck0;
        attns_clock[(blockIdx.x * WAVES + tw) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif

    // ---- flush the per-workgroup reduction: one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[tw] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < PW; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
}

}  // namespace

#ifdef SWV2_ATTNS_STAMPS
extern "C" int swv2_debug_attns_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_stamps), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -3;
}
extern "C" int swv2_debug_attns_win(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_win), sizeof(unsigned long long) * 64 * 128) == hipSuccess ? 0 : -3;
}
extern "C" int swv2_debug_attns_clock(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_clock), sizeof(unsigned long long) * 512 * 2) == hipSuccess ? 0 : -3;
}
#endif

// called by swv2_attn_bwd (attn.hip); returns 1 when this kernel does not cover the shape or is switched off (the caller then runs
// the two-phase kernel): the 176-row layout with 16-wide head slots, no CPB table.  Window areas: any L <= 176 of the layout (padded
// keys are switched off inside the operand, padded query rows carry lse = 1e30).
int swv2_attn_bwd_stream(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    static const int on = getenv("SWV2_ATTN_BWD_STREAM") ? atoi(getenv("SWV2_ATTN_BWD_STREAM")) : 1;
    if (!on || Lp != 176 || DP != 16 || a->bias || (a->dbg & (SWV2_ATTN_PLAIN_STATS | SWV2_ATTN_BWD_TWO_PHASE))) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = a->Bw < a->max_chunks ? a->Bw : a->max_chunks;
    dim3 grid(nchunk, a->heads), block(1024);
    const int nW = a->nwh * a->nww;
    if (a->L == 162)
        hipLaunchKernelGGL((attn_bwd_stream_kernel<162>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale, (const uint16_t*)a->oh,
                           (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, a->Bw, a->heads, a->L, nW, a->nww,
                           a->nwh, a->mask_thr);
    else
        hipLaunchKernelGGL((attn_bwd_stream_kernel<0>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale, (const uint16_t*)a->oh,
                           (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, a->Bw, a->heads, a->L, nW, a->nww,
                           a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}
