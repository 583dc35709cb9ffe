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
// nothing to overlap it (20 %), then the commit of the next window's slabs between two more barriers (7 %).  Here a workgroup is 16
// waves (4 per SIMD, 128 registers):
//   * waves 0 .. 10, phase 1, unchanged arithmetic (wave = key tile, q-tiles in pairs, software-pipelined, fully unrolled with fences);
//     behind the dS tiles of a q-tile pair, lane 0 of the wave adds 1 to the pair's LDS counter (the LDS executes a wave's
//     instructions in order: the add is behind the tile writes).  No staging registers, no dQ work: ~100 registers;
//   * waves 11 .. 15, helpers: each prefetches a fifth of the next window's slabs (v by LDS-DMA, the rest through registers), computes
//     the dQ of one q-tile pair -- spinning on the pair's counter until it shows all 11 phase-1 waves, then reading the dS image
//     transposed (K^T fragments once per pair, two interleaved accumulation chains, the two-phase kernel's summation order) -- and
//     commits its slab chunks (with delta = rowsum(dO O) and the statistics slots) to the OTHER LDS buffer;
//   * the q | dO | k | v slabs (and the 1 / |q|, 1 / |k| rows) are double-buffered in LDS, so the commit needs no barrier of its own;
//     the dS image is single (every reader has passed the window's ONE barrier before the next window's first tile is written);
//   * issue priority (s_setprio) falls with a phase-1 wave's own progress, so the three phase-1 waves of a SIMD advance together
//     instead of oldest first.
// Measured with tools/probe_attn_bwd_windows.py (s_memtime at every window's end, same box): 8 740 cycles per window for the two-phase
// kernel, 7 480 - 7 520 here, the same at 25 and at 100 windows per workgroup; d(qkv) bit-identical.  What was built on the way and
// lost (LABNOTES round 6): 11 waves with the dQ tiles inside every wave's own loop (8 560) or on the two waves of the short SIMD (one
// tile at a time: slower than two-phase -- a tile's chain of transposed reads -> 6 dependent MFMAs -> row sum -> store is ~400 cycles of
// latency); the last three q-tiles' dQ deferred into the next window through a second tail image (tools/experiments/
// attn_bwd_stream_deferred.hip: 7 620, the commits then wait for those tiles); helpers at priority 2 / 3 (7 620 - 7 670).
// The phase-1 loop is ~270 issue cycles per q-tile pair by the instruction costs of MI355X_MICROARCH (8 exp, 8 + 4 multiplies, 8 packs,
// 6 MFMAs, 9 LDS instructions, the signal): ~5 000 per window and SIMD -- the kernel runs at two thirds of its own issue bound.
// LDS: 2 x 40 832 (slabs) + 63 360 (dS image) + counters = 145 KB, one persistent workgroup per CU as before.
#include <stdlib.h>

#include "attn_common.h"

namespace {

typedef __attribute__((address_space(3))) unsigned lds_u32;
#ifdef SWV2_ATTNS_ARRIVE          // diagnostic: when each wave of ONE workgroup reaches the window's barrier (8 windows), and when the barrier opens
__device__ unsigned long long attns_arr[17 * 8];
#endif
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile u32x4s lds_cv4;      // (a generic volatile pointer compiled to flat loads + vmcnt(0) waits)

#ifdef SWV2_ATTNS_STAMPS          // diagnostic build (tools/probe_attn_stream_stamps.py): per-phase s_memtime sums of every wave
__device__ unsigned long long attns_stamps[512 * 8];
__device__ unsigned long long attns_win[64 * 128];        // wave 8 of the first 64 workgroups of head 0: s_memtime at the end of every window
__device__ unsigned long long attns_clock[512 * 2];       // per wave: s_memtime span, s_memrealtime span (100 MHz) of the window loop
#define SSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
__device__ unsigned long long attns_tl[16 * 24];          // one window of one workgroup: per wave up to 24 events (tag << 56 | cycles since the loop's start)
#define STL(tag, t_) do { if (tl_on && tl_n < 24) { if (lane == 0) attns_tl[tw * 24 + tl_n] = ((unsigned long long)(tag) << 56) | ((t_) - ck0); ++tl_n; } } while (0)
#define SSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; STL(k, t_); } while (0)
#define STLNOW(tag) do { if (tl_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); STL(tag, t_); } } while (0)
#else
#define STLNOW(tag) do {} while (0)
#define SSTAMP_DECL
#define SSTAMP_START() do {} while (0)
#define SSTAMP(k) do {} while (0)
#endif

#ifndef SWV2_ATTNS_NO_PRIO          // (A/B builds)
#define SWV2_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define SWV2_PRIO(n) do {} while (0)
#endif
#ifndef SWV2_ATTNS_ABL              // timing ablations (WRONG results; diagnostic builds only): 1 no exp, 2 no scale multiply, 4 no transposed q / dO reads,
#define SWV2_ATTNS_ABL 0            // 8 no dS image writes, 16 helpers skip the dQ pass, 32 no row-major q / dO reads
#endif
#ifndef SWV2_ATTNS_TAIL             // 1: the last five q-tiles' dQ on phase-1 waves that are done; helpers commit first and do pairs 0 .. 2
#define SWV2_ATTNS_TAIL 3
#endif
#ifndef SWV2_ATTNS_ISSUE_PRIO       // 1: the helpers issue the next window's prefetch at priority 3
#define SWV2_ATTNS_ISSUE_PRIO 1
#endif
#ifndef SWV2_ATTNS_PREFETCH_ALL     // 1: a dQ pass reads all its fragments before its first product
#define SWV2_ATTNS_PREFETCH_ALL 2
#endif
#ifndef SWV2_ATTNS_HPRIO            // 0: phase-1 priorities 3 3 2 1 1 0; 1: 2 2 1 1 0 0 and 2: 2 2 2 1 1 0 with the helpers' prefetch issue and commit at 3
#define SWV2_ATTNS_HPRIO 0
#endif
#ifndef SWV2_ATTNS_YOUTH            // 1: the phase-1 priority also depends on the wave's age within its SIMD
#define SWV2_ATTNS_YOUTH 0
#endif
#ifndef SWV2_ATTNS_PF2              // 1: prefetch distance two windows -- the helpers commit the NEXT window's rows first thing in a window (loaded a whole window ago)
#define SWV2_ATTNS_PF2 0            // and then request the window after it; k / v land by DMA in one of THREE buffers
#endif
#ifndef SWV2_ATTNS_NOBAR            // 1 (needs PF2): no workgroup barrier between windows -- every hand-over goes through LDS counters, so a window's tail
#define SWV2_ATTNS_NOBAR 0          // (the last q-tiles' dQ behind the last signals) overlaps the next window's first pairs
#endif
#if SWV2_ATTNS_NOBAR && !SWV2_ATTNS_PF2
#error "SWV2_ATTNS_NOBAR needs SWV2_ATTNS_PF2"
#endif
#ifndef SWV2_ATTNS_NOBAR_TAILW      // (A/B) 1: NOBAR with the last three q-tiles on phase-1 waves
#define SWV2_ATTNS_NOBAR_TAILW 0
#endif
#ifndef SWV2_ATTNS_COMMIT_FIRST     // helpers from this index on commit before their dQ pair (A/B builds: 99 = none)
#define SWV2_ATTNS_COMMIT_FIRST 2
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
    constexpr int P1A = SWV2_ATTNS_HPRIO ? 2 : 3, P1B = SWV2_ATTNS_HPRIO == 1 ? 1 : 2, P1C = 1, P1D = SWV2_ATTNS_HPRIO == 1 ? 0 : 1;
    constexpr int PW = LT, HW = 5, WAVES = PW + HW;          // 11 phase-1 waves (wave = key tile) + 5 helper waves (staging, commit, dQ): 4 per SIMD
    constexpr int DSP = Lp + 4;                              // row pitch (elements) of the [key][q] bf16 dS image
    constexpr int QP = 40, QSTAT = 16;                       // 80-byte q / dO rows: 16 channels, 8 statistics slots, 16 bytes of padding (bank spread)
    constexpr int CH = SLAB / 8, CPR = 2;                    // 16-byte chunks per slab / per row
    constexpr int NPAIR = (LT + 1) / 2;                      // q-tile pairs (the odd last tile is a "pair" of its own)
    constexpr int HT = 64 * HW;                              // helper threads (320): chunk c = their index, the first 32 of the last helper also chunk 320 + lane
    static_assert(CH > HT && CH <= HT + 32, "second chunks on the first half of one helper wave");
    // one buffer: q rows | dO rows | k | v | 1/|q|, 1/|k|
#if SWV2_ATTNS_PF2
    // two row buffers: q rows | dO rows | 1/|q|, 1/|k| ; three k | v buffers (the DMA of window w + 2 is issued while window w is computed)
    constexpr int B_Q = 0, B_DO = B_Q + Lp * QP * 2, B_RN = B_DO + Lp * QP * 2, BUFB = B_RN + 2 * Lp * 4;
    constexpr int OFF_KV = 2 * BUFB, KVB = 2 * SLAB * 2, NKV = 3;
    constexpr int OFF_DS = OFF_KV + NKV * KVB, OFF_CNT = OFF_DS + Lp * DSP * 2, OFF_RED = OFF_CNT + 64, LDS_BYTES = OFF_RED + ((WAVES * 4 + 15) / 16) * 16;
#else
    constexpr int B_Q = 0, B_DO = B_Q + Lp * QP * 2, B_K = B_DO + Lp * QP * 2, B_V = B_K + SLAB * 2, B_RN = B_V + SLAB * 2,
                  BUFB = B_RN + 2 * Lp * 4;
    constexpr int OFF_DS = 2 * BUFB, OFF_CNT = OFF_DS + Lp * DSP * 2, OFF_RED = OFF_CNT + 64, LDS_BYTES = OFF_RED + ((WAVES * 4 + 15) / 16) * 16;
#endif
    static_assert(BUFB % 16 == 0 && OFF_DS % 16 == 0 && OFF_CNT % 16 == 0, "16-byte aligned sub-arrays");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
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

    // counters (u32, monotone): words 0 .. 5 sig[p], dS tiles of q-tile pair p written (+1 per phase-1 wave and window); NOBAR: word 8 cd, helper
    // commits of the NEXT window done (+1 per helper and window); 9 .. 14 cons[p], q-tiles of pair p whose dQ pass has read its LDS operands;
    // 15 pd, phase-1 waves done with a window's rows / k / v
    if (tid < 16) ((unsigned*)(lds + OFF_CNT))[tid] = 0u;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u32*)lds;
    const unsigned cnt_addr = lds_base + OFF_CNT;           // the counters' LDS address (for the inline ds_add / ds_read)
#if SWV2_ATTNS_NOBAR
    // block A = {cd, cons0, cons1, cons2}, block B = {cons3, cons4, cons5, pd}: one 16-byte read each
    lds_cv4* const cntA = (lds_cv4*)((lds_u32*)lds + (OFF_CNT + 32) / 4);
    lds_cv4* const cntB = (lds_cv4*)((lds_u32*)lds + (OFF_CNT + 48) / 4);
    auto rfl = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    // A: commits of window `w` done and pairs 0 .. 2 of window w - 1 consumed; B: pairs 3 .. 5 of window w - 1 consumed
    auto okA = [&](const u32x4s a, int w) { return rfl(a.x) >= 5u * (unsigned)w && rfl(a.y) >= 2u * (unsigned)w && rfl(a.z) >= 2u * (unsigned)w && rfl(a.w) >= 2u * (unsigned)w; };
    auto okB = [&](const u32x4s b, int w) { return rfl(b.x) >= 2u * (unsigned)w && rfl(b.y) >= 2u * (unsigned)w && rfl(b.z) >= 1u * (unsigned)w; };
#endif

    // ---- staging (helper waves): one 16-byte chunk of the q, k, v, dO, o slabs; lse of the chunk's row; one 1/|.| value -- twice on helper 0
    // (k and v go straight to LDS by DMA: 44 staging registers beside the dQ pair did not fit the 128 of a 16-wave workgroup)
#ifdef SWV2_ATTNS_STAMPS
    bool tl_on = false;
    int tl_n = 0;
    unsigned long long ck0 = 0;
#endif
    struct Stg { uint4 q, dO, o; float lse, rn; };
    Stg s0, s1;
    const bool two = helper && hw == HW - 1;                   // wave-uniform: this wave stages a second chunk (its lanes 0 .. 31); it commits BEFORE its dQ pair
    // 32-bit, loop-invariant lane offsets against wave-uniform bases: the loads take the (SGPR base + VGPR offset) form.  With 64-bit
    // per-lane addresses the compiler builds them in the loads' own destination registers and guards that overwrite with s_waitcnt
    // vmcnt(..) -- which, the counter being in order, also waits for earlier d(qkv) STORES (ISA, attn2.hip round 2)
    auto issue = [&](int bw, int nbuf, int hidx) {      // nbuf: the LDS buffer the window will be committed to; hidx: the helper thread's index
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
        auto one = [&](Stg& st, const int craw, const unsigned lds_k) {
            STLNOW(20);
            const int c = min(craw, CH - 1);
            // (opaque per call: otherwise loop-invariant code motion folds the lane offsets into 64-bit per-lane pointers outside the window loop)
            unsigned o16 = (unsigned)c * 16u, ol = (unsigned)(c / CPR) * 4u, orn = (unsigned)c * 4u;
            asm volatile("" : "+v"(o16), "+v"(ol), "+v"(orn));
            // k, v: LDS-DMA (M0 = the LDS address of the wave's first chunk, lane l lands 16 l bytes behind it), issued BEFORE the register
            // loads: the counter is in order, so the compiler's own wait for a younger register load covers them
            if (craw < CH) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(o16), "s"(kb_), "s"(lds_k) : "memory");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(o16), "s"(vb_), "s"(lds_k + SLAB * 2) : "memory");
            }
            STLNOW(21);
            st.q = *(const uint4*)(qb + o16);
            st.dO = *(const uint4*)(dob + o16);
            st.o = *(const uint4*)(ob + o16);
            st.lse = *(const float*)((const char*)(lse + item * Lp) + ol);
            st.rn = *(const float*)((const char*)(rnorm + item * 2 * Lp) + orn);
        };
#if SWV2_ATTNS_PF2
        const unsigned lds_k0 = lds_base + (unsigned)(OFF_KV + nbuf * KVB);        // nbuf: one of the three k | v buffers
#else
        const unsigned lds_k0 = lds_base + (unsigned)(nbuf * BUFB + B_K);
#endif
        one(s0, hidx, lds_k0 + (unsigned)hw * 1024u);
        if (two) one(s1, HT + (hidx & 63), lds_k0 + (unsigned)HT * 16u);
    };
    auto commit = [&](int buf, int hidx) {
        unsigned char* const B = lds + buf * BUFB;
        auto one = [&](const Stg& st, const int craw) {
            const int c = min(craw, CH - 1);
            const int row = c / CPR, half = c % CPR;
            // delta partial over this chunk's 8 channels, reduced over the 2 chunks of the row (adjacent lanes; vector ALU only)
            float dl = 0.f;
            {
                const uint32_t a[4] = {st.dO.x, st.dO.y, st.dO.z, st.dO.w}, b[4] = {st.o.x, st.o.y, st.o.z, st.o.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), dl);
                    dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), dl);
                }
            }
            dl = group_allsum<CPR>(dl);
            // slots 16..23 of the row (the even chunk's thread): lse / (sigma log2 e) in three bf16 parts, a constant 1 (padded-key
            // flag), the query's mask-region flags -- and delta in three parts for the dO row; slots 24..31 (odd chunk): zeros
            uint4 aq = make_uint4(0, 0, 0, 0), ad = make_uint4(0, 0, 0, 0);
            if (half == 0) {
                const bool q_ok = row < L;
                const float lq = q_ok ? st.lse * inv_sc2 : 1.0e30f;            // padded query rows: P = 0
                uint16_t l0 = f2bf(lq);
                const float r1 = lq - bf2f(l0);
                uint16_t l1 = f2bf(r1), l2 = f2bf(r1 - bf2f(l1));
                if (!q_ok) l1 = l2 = 0;
                const uint16_t d0 = f2bf(dl);
                const float e1 = dl - bf2f(d0);
                const uint16_t d1 = f2bf(e1), d2 = f2bf(e1 - bf2f(d1));
                const uint32_t one_ = 0x3f80u, rqf = (row >= mask_thr) ? 0x3f80u : 0u;
                aq = make_uint4(l0 | ((uint32_t)l1 << 16), l2 | (one_ << 16), rqf | ((one_ - rqf) << 16), 0);
                ad = make_uint4(d0 | ((uint32_t)d1 << 16), d2, 0, 0);
            }
            if (craw < CH) {
                *(uint4*)((uint16_t*)(B + B_Q) + row * QP + half * 8) = st.q;
                *(uint4*)((uint16_t*)(B + B_DO) + row * QP + half * 8) = st.dO;
                *(uint4*)((uint16_t*)(B + B_Q) + row * QP + QSTAT + half * 8) = aq;
                *(uint4*)((uint16_t*)(B + B_DO) + row * QP + QSTAT + half * 8) = ad;
                ((float*)(B + B_RN))[c] = st.rn;
            }
        };
        one(s0, hidx);
        if (two) one(s1, HT + (hidx & 63));
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    if (helper) {
        issue(bw, 0, hw * 64 + lane);
        commit(0, hw * 64 + lane);
#if SWV2_ATTNS_PF2 && !SWV2_ATTNS_NOBAR
        if (bw + (int)gridDim.x < Bw) issue(bw + gridDim.x, 1, hw * 64 + lane);
#endif
    }
    __syncthreads();
#if SWV2_ATTNS_PF2
    int kv = 0;                 // the k | v buffer of the current window (it % 3)
#endif

    float dsig = 0.f;
    SSTAMP_DECL
    SSTAMP_START();
#ifdef SWV2_ATTNS_STAMPS
    ck0 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // NOBAR: the two roles run two loops of their own (no barrier inside forces them into one): the helpers' staging registers, live across
    // the loop, then do not count against the phase-1 code's registers (in one loop they were spilled right behind their loads)
    auto windows = [&](auto role_c) {
    const bool is_helper = SWV2_ATTNS_NOBAR ? (bool)decltype(role_c)::value : helper;
    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        const int bw_next = bw + gridDim.x;
        const unsigned char* const B = lds + buf * BUFB;
        const uint16_t* const Qa = (const uint16_t*)(B + B_Q);
        const uint16_t* const Da = (const uint16_t*)(B + B_DO);
#if SWV2_ATTNS_PF2
        const uint16_t* const Ks = (const uint16_t*)(lds + OFF_KV + kv * KVB);
        const uint16_t* const Vs = Ks + SLAB;
#else
        const uint16_t* const Ks = (const uint16_t*)(B + B_K);
        const uint16_t* const Vs = (const uint16_t*)(B + B_V);
#endif
        const float* const RN = (const float*)(B + B_RN);
        const unsigned target = (unsigned)(PW * (it + 1));
#ifdef SWV2_ATTNS_STAMPS
        tl_on = it == 10 && blockIdx.x == 3 && blockIdx.y == 0;
        tl_n = 0;
        STLNOW(15);
#endif

        // (the lane id is re-derived behind an opaque asm in each role's branch: otherwise loop-invariant code motion hoists every
        // lane-dependent address of BOTH roles in front of the window loop and spills them -- 53 registers, reloaded by VMEM operations)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, g = ln >> 4;
#if SWV2_ATTNS_NOBAR
        auto ctr_add = [&](const int word, const unsigned val) {
            if (ln == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_addr + 4u * (unsigned)word), "v"(val) : "memory");
        };
#endif
            // dQ of the q-tiles of pair `pr` (NQ = 2) or of the odd last tile (NQ = 1): dQ^T = sum_t K_t^T dS_t^T, key tiles in pairs (one
            // K = 32 product per pair), both operands as transposed reads -- the K^T fragments are read once for both q-tiles, whose
            // accumulation chains interleave; one chain per q-tile in key order + the odd key tile on its own accumulator: the summation
            // order of the two-phase kernel (bit-identical d q)
            auto phase2 = [&](const int qt0, auto nq_c, auto sleep_c) {      // q-tiles qt0 .. qt0 + NQ - 1 (of one pair)
                constexpr int NQ = decltype(nq_c)::value;
                if constexpr (SWV2_ATTNS_ABL & 16) return;
                const uint16_t* const kb = Ks + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4;
                const uint16_t* const db = dSb + (4 * g + (fr >> 2)) * DSP + 16 * qt0 + (fr & 3) * 4;
#if SWV2_ATTNS_PREFETCH_ALL >= 2
                // what does not depend on the signal is read in FRONT of the poll: the K^T fragments, 1 / |q| and q^ of the normalisation backward
                bf16x4 kk[LT], qn[NQ];
                float rq[NQ];
#pragma unroll
                for (int t = 0; t < LT; ++t) kk[t] = lds_tr_read(kb + 16 * t * DP);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    rq[i] = RN[16 * qt0 + 16 * i + fr];
                    qn[i] = *(const bf16x4*)(Qa + (16 * qt0 + 16 * i + fr) * QP + 4 * g);
                }
#endif
                {
                    const unsigned a = cnt_addr + 4u * (unsigned)(qt0 >> 1);
                    while (true) {
                        unsigned v;
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
                        if ((unsigned)__builtin_amdgcn_readfirstlane(v) >= target) break;
                        __builtin_amdgcn_s_sleep(decltype(sleep_c)::value);
                    }
                }
                SSTAMP(2);
                f32x4 dq[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if SWV2_ATTNS_PREFETCH_ALL
                // every fragment read first, then the products: a single wave has nothing else to cover an LDS round trip per product
                // (a dQ pass of one q-tile behind the window's last signals: ~970 -> ~620 cycles)
                bf16x4 dd[NQ][LT];
#if SWV2_ATTNS_PREFETCH_ALL < 2
                bf16x4 kk[LT];
#pragma unroll
                for (int t = 0; t < LT; ++t) kk[t] = lds_tr_read(kb + 16 * t * DP);
#endif
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int i = 0; i < NQ; ++i) dd[i][t] = lds_tr_read(db + 16 * t * DSP + 16 * i);
#if SWV2_ATTNS_NOBAR
                ctr_add(9 + (qt0 >> 1), (unsigned)NQ);       // behind the pass's last LDS read (the LDS runs a wave's instructions in order)
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t + 1 < LT; t += 2) {
                    const bf16x8 ka = __builtin_shufflevector(kk[t], kk[t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        dq[i] = mfma32(ka, __builtin_shufflevector(dd[i][t], dd[i][t + 1], 0, 1, 2, 3, 4, 5, 6, 7), dq[i]);
                }
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    // own accumulator: a K = 16 MFMA chained directly onto a K = 32 accumulator gave wrong sums (attn.hip)
                    const f32x4 tail = mfma16(kk[LT - 1], dd[i][LT - 1], (f32x4){0.f, 0.f, 0.f, 0.f});
                    dq[i] += tail;
                }
#else
#pragma unroll
                for (int t = 0; t + 1 < LT; t += 2) {
                    const bf16x4 k0 = lds_tr_read(kb + 16 * t * DP), k1 = lds_tr_read(kb + 16 * (t + 1) * DP);
                    const bf16x8 ka = __builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const bf16x4 d0 = lds_tr_read(db + 16 * t * DSP + 16 * i), d1 = lds_tr_read(db + 16 * (t + 1) * DSP + 16 * i);
                        dq[i] = mfma32(ka, __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7), dq[i]);
                    }
                }
                {
                    // own accumulator: a K = 16 MFMA chained directly onto a K = 32 accumulator gave wrong sums (attn.hip)
                    const bf16x4 k0 = lds_tr_read(kb + 16 * (LT - 1) * DP);
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const bf16x4 d0 = lds_tr_read(db + 16 * (LT - 1) * DSP + 16 * i);
                        const f32x4 tail = mfma16(k0, d0, (f32x4){0.f, 0.f, 0.f, 0.f});
                        dq[i] += tail;
                    }
                }
#endif
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int q = 16 * qt0 + 16 * i + fr;
#if SWV2_ATTNS_PREFETCH_ALL >= 2
                    const float rqs = rq[i] * sigma;
                    const bf16x4 qv = qn[i];
#else
                    const float rqs = RN[q] * sigma;
                    const bf16x4 qv = *(const bf16x4*)(Qa + q * QP + 4 * g);
#endif
                    float dot = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][r], bf2f(qv[r]), dot);
                    dot = xor32_allsum(xor16_allsum(dot));
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rqs * (dq[i][r] - bf2f(qv[r]) * dot);
                    *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 4 * g) = f2bf4(v);
                }
                SSTAMP(3);
            };
        if (!is_helper) {
            // ================= phase 1: wave = key tile =================
            const int key = 16 * tw + fr;
#if SWV2_ATTNS_NOBAR
            // both counter blocks are read once, first thing in the window (one LDS round trip per window and wave); a pair's columns of the dS
            // image are written when the previous window's passes over them are done: normally what the early read already shows, else a poll
            u32x4s aq = *cntA;
            const u32x4s bq = *cntB;
            while (rfl(aq.x) < 5u * (unsigned)it) { __builtin_amdgcn_s_sleep(1); aq = *cntA; }      // this window's rows and k / v are committed
            auto need = [&](auto p_c) {
                constexpr int P = decltype(p_c)::value;
                const unsigned tgt = (P == 5 ? 1u : 2u) * (unsigned)it;
                auto pick = [&](const u32x4s t) { return P == 0 ? t.y : P == 1 ? t.z : P == 2 ? t.w : P == 3 ? t.x : P == 4 ? t.y : t.z; };
                if (rfl(pick(P < 3 ? aq : bq)) >= tgt) return;
                while (true) {
                    const u32x4s t = P < 3 ? *cntA : *cntB;
                    if (rfl(pick(t)) >= tgt) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            };
#endif
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

            struct St { f32x4 s, dp; bf16x4 tq, td; };
            auto stageA = [&](const int qt, St& o) {
                bf16x8 qa, da;
                if constexpr (SWV2_ATTNS_ABL & 32) { qa = kf8; da = vf8; asm volatile("" : "+v"(qa), "+v"(da)); }
                else {
                    qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                    da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
                }
                if constexpr (SWV2_ATTNS_ABL & 4) { o.td = kf; o.tq = kf; asm volatile("" : "+v"(o.td), "+v"(o.tq)); }
                else {
                o.td = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.tq = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                }
                o.s = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
                o.dp = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
            };
            // the dS tiles of q-tile pair `p` of this wave are in the image: tell the dQ side (the LDS runs a wave's instructions in order)
            auto signal = [&](const int p) {
                // (EXEC narrowed to lane 0 inside the asm -- three instructions instead of the compare / saveexec / branch / restore -- measured equal:
                // 7 555 - 7 575 against 7 529 - 7 614 cycles per window)
                if (ln == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt_addr + 4u * (unsigned)p), "v"(1u) : "memory");
            };
            // q-tiles in PAIRS: the dV / dK products of two tiles are one K = 32 MFMA (k-slot (g, j) = row 4g + j of the first tile for
            // j < 4, of the second for j >= 4, on both operands) -- 6 MFMAs per pair instead of 8; the odd last tile has its own accumulators
            auto stageB2 = [&](const int qt, const St& i0, const St& i1) {
                f32x4 p0, p1, ds0, ds1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (SWV2_ATTNS_ABL & 1) { p0[r] = i0.s[r] * sc2; p1[r] = i1.s[r] * sc2; }
                    else if constexpr (SWV2_ATTNS_ABL & 2) { p0[r] = __builtin_amdgcn_exp2f(i0.s[r]); p1[r] = __builtin_amdgcn_exp2f(i1.s[r]); }
                    else {
                    p0[r] = __builtin_amdgcn_exp2f(i0.s[r] * sc2);
                    p1[r] = __builtin_amdgcn_exp2f(i1.s[r] * sc2);
                    }
                    ds0[r] = p0[r] * i0.dp[r];
                    ds1[r] = p1[r] * i1.dp[r];
                }
                const bf16x4 pb0 = f2bf4(p0), pb1 = f2bf4(p1), dsb0 = f2bf4(ds0), dsb1 = f2bf4(ds1);
                if constexpr (!(SWV2_ATTNS_ABL & 8)) {
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
            // Issue priority falls with the wave's own progress (3 at the start of the window, 0 behind the fourth pair): the SIMD arbiter serves
            // the oldest wave of the highest priority first, so without this the oldest wave of a SIMD runs ahead, ends at ~60 % of the window,
            // and the youngest finishes alone with nothing to cover its latencies.  A wave that is behind now outranks one that is ahead
            // (same box, cycles per window: 8 099 without, 7 716 with; helpers at priority 2 / 3 on top: 7 620 - 7 670 against 7 473 - 7 516).
#if SWV2_ATTNS_YOUTH
            // stage s = pairs signalled so far; the older waves of a SIMD (served first at equal priority) step down earlier
            const int cls = tw >> 2;       // 0: waves 0 .. 3 (oldest of their SIMD), 1: 4 .. 7, 2: 8 .. 10
            auto setp = [&](auto s_c) {
                constexpr int sg = decltype(s_c)::value;
                constexpr int p2 = 3 - (2 * sg) / 3, p1 = 3 - (2 * sg + 1) / 3, p0 = (3 - (2 * sg + 2) / 3) < 0 ? 0 : 3 - (2 * sg + 2) / 3;
                if (cls == 2) SWV2_PRIO(p2); else if (cls == 1) SWV2_PRIO(p1); else SWV2_PRIO(p0);
            };
            setp(std::integral_constant<int, 0>{});
#else
            SWV2_PRIO(P1A);
#endif
            St a0, a1, b0, b1;
            stageA(0, a0);
            stageA(1, a1);
#pragma unroll
            for (int qt = 0; qt + 1 < LT; qt += 4) {
                if (qt + 2 < LT) stageA(qt + 2, b0);
                if (qt + 3 < LT) stageA(qt + 3, b1);
#if SWV2_ATTNS_NOBAR
                if (qt == 0) need(std::integral_constant<int, 0>{}); else if (qt == 4) need(std::integral_constant<int, 2>{}); else need(std::integral_constant<int, 4>{});
#endif
                stageB2(qt, a0, a1);
                signal(qt >> 1);
                STLNOW(8 + (qt >> 1));
#if SWV2_ATTNS_YOUTH
                if (qt == 0) setp(std::integral_constant<int, 1>{}); else if (qt == 4) setp(std::integral_constant<int, 3>{}); else setp(std::integral_constant<int, 5>{});
#else
                if (qt == 0) SWV2_PRIO(P1A); else if (qt == 4) SWV2_PRIO(P1D); else SWV2_PRIO(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (qt + 3 < LT) {
                    if (qt + 4 < LT) stageA(qt + 4, a0);
                    if (qt + 5 < LT) stageA(qt + 5, a1);
#if SWV2_ATTNS_NOBAR
                    if (qt == 0) need(std::integral_constant<int, 1>{}); else need(std::integral_constant<int, 3>{});
#endif
                    stageB2(qt + 2, b0, b1);
                    signal((qt >> 1) + 1);
                    STLNOW(8 + (qt >> 1) + 1);
#if SWV2_ATTNS_YOUTH
                    if (qt == 0) setp(std::integral_constant<int, 2>{}); else setp(std::integral_constant<int, 4>{});
#else
                    if (qt == 0) SWV2_PRIO(P1B); else SWV2_PRIO(P1C);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            static_assert((LT & 3) == 3, "the odd last tile's stage A went into b0");
            {
#if SWV2_ATTNS_NOBAR
                need(std::integral_constant<int, 5>{});
#endif
                const St& in = b0;
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(in.s[r] * sc2);
                    ds[r] = p[r] * in.dp[r];
                }
                const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                *(bf16x4*)(dSb + key * DSP + 16 * (LT - 1) + 4 * g) = dsb;
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
#if SWV2_ATTNS_NOBAR
            ctr_add(15, 1u);                // this wave's reads of the window's rows / k / v as a phase-1 wave are done
#endif
#if SWV2_ATTNS_TAIL
            // the last five q-tiles' dQ: one tile each on five phase-1 waves that are done with their key tile (the oldest of each SIMD end phase 1
            // at 75 - 85 % of it) -- behind the last signals the window then ends with ONE single-tile pass per wave instead of the helpers' pair passes
            {
#if SWV2_ATTNS_NOBAR && !SWV2_ATTNS_NOBAR_TAILW
                const int tt = -1;
#elif SWV2_ATTNS_PF2
                const int tt = tw == 3 ? 8 : tw == 7 ? 9 : tw == 2 ? 10 : -1;
#elif SWV2_ATTNS_TAIL == 4
                const int tt = tw == 3 ? 5 : tw == 7 ? 6 : tw == 2 ? 7 : tw == 0 ? 8 : tw == 1 ? 9 : tw == 5 ? 10 : -1;
#else
                const int tt = tw == 3 ? 6 : tw == 7 ? 7 : tw == 0 ? 8 : tw == 2 ? 9 : tw == 1 ? 10 : -1;
#endif
                if (tt >= 0) {
#ifdef SWV2_ATTNS_TAILPRIO
                    SWV2_PRIO(SWV2_ATTNS_TAILPRIO);
#else
                    SWV2_PRIO(2);
#endif
#if SWV2_ATTNS_TAIL == 5
                    if (tw == 3) phase2(5, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});     // (helper 4 commits two chunks: its tile)
#endif
                    phase2(tt, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                }
            }
#endif
        } else {
            // ================= helper waves: the next window's prefetch, phase 2 (dQ), the commit =================
            const int hidx = hw * 64 + ln;
#if SWV2_ATTNS_PF2
            // the next window's rows were requested a whole window ago: commit them first thing (no wait), then request the window after it
            // (k / v by DMA into the k | v buffer that the PREVIOUS window used), then the helper's q-tiles as their signals arrive
#if SWV2_ATTNS_NOBAR
#ifdef SWV2_ATTNS_NOBAR_HP
            SWV2_PRIO(SWV2_ATTNS_NOBAR_HP);
#endif
            // the row buffer and the k | v buffer written next were read by the PREVIOUS window: all its phase-1 waves and all its dQ passes are done
            while (true) {
                const u32x4s a = *cntA, b = *cntB;
                if (rfl(b.w) >= (unsigned)(PW * it) && rfl(a.y) >= 2u * (unsigned)it && rfl(a.z) >= 2u * (unsigned)it && rfl(a.w) >= 2u * (unsigned)it && okB(b, it)) break;
                __builtin_amdgcn_s_sleep(2);
            }
#endif
            if (bw_next < Bw) {
                commit(buf ^ 1, hidx);
#if SWV2_ATTNS_NOBAR
                ctr_add(8, 1u);
#endif
            }
            SSTAMP(5);
            if (bw_next + (int)gridDim.x < Bw) issue(bw_next + gridDim.x, kv == 0 ? 2 : kv - 1, hidx);
            SSTAMP(0);
#if SWV2_ATTNS_NOBAR
            // single-tile passes only (a pair pass holds 66 fragment registers: with the staging registers live across the loop it spilled them)
#if SWV2_ATTNS_NOBAR_TAILW
            phase2(hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            if (hw < 3) phase2(5 + hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
#else
            // ALL q-tiles on the helpers 0 .. 3 (h: tiles h, h + 4, h + 8): without a barrier there is no window tail to shorten, and a
            // phase-1 wave that also runs a dQ pass falls a window behind the others -- everything then waits for it
            // (helper 4 stages and commits two chunks per thread: no tile)
            if (hw < 4) {
                phase2(hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                phase2(4 + hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                if (hw < 3) phase2(8 + hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            }
#endif
#else
            if (hw == 0) phase2(0, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            else phase2(hw + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            if (hw < 2) phase2(6 + hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
#endif
        }
#else
#if SWV2_ATTNS_ISSUE_PRIO
            SWV2_PRIO(3);                   // the prefetch goes out in front of the phase-1 waves' first pairs (otherwise: at 35 - 45 % of the window)
#endif
            if (bw_next < Bw) issue(bw_next, buf ^ 1, hidx);
#if SWV2_ATTNS_ISSUE_PRIO
            if (!(SWV2_ATTNS_HPRIO && SWV2_ATTNS_TAIL >= 2)) SWV2_PRIO(0);
#endif
            SSTAMP(0);
#if SWV2_ATTNS_TAIL
            // helpers 0, 1: the dQ of pairs 0, 1; helpers 2, 3: of q-tiles 4, 5; then (helper 4: only) the commit of the chunks the helper staged
#if SWV2_ATTNS_TAIL == 4
            // helper 0: pair 0, helpers 1 .. 3: q-tiles 2 .. 4 (complete at 40 - 55 % of the window, before the prefetch has landed), then the commit;
            // helper 4 (two chunks): the commit only
            if (hw == 0) phase2(0, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            else if (hw < 4) phase2(hw + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            if (bw_next < Bw) commit(buf ^ 1, hidx);
            SSTAMP(5);
#elif SWV2_ATTNS_TAIL >= 2
            if (SWV2_ATTNS_HPRIO) SWV2_PRIO(3);
            if (bw_next < Bw) commit(buf ^ 1, hidx);
            if (SWV2_ATTNS_HPRIO) SWV2_PRIO(0);
            SSTAMP(5);
            if (SWV2_ATTNS_TAIL == 2) {
                if (hw < 3) phase2(2 * hw, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            } else {      // 3: pair 0 on helper 0, q-tiles 2 .. 5 on helpers 1 .. 4 (5: helper 4 none)
                if (hw == 0) phase2(0, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                else if (SWV2_ATTNS_TAIL != 5 || hw < 4) phase2(hw + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            }
#else
            if (hw < 2) phase2(2 * hw, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            else if (hw < 4) phase2(2 + hw, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            if (bw_next < Bw) commit(buf ^ 1, hidx);
            SSTAMP(5);
#endif
#else
            // helper hw: pair hw and the commit; helper 4 (the last pair) commits first; helper 0 ends with the odd last tile
            if (hw == HW - 1) {
                if (bw_next < Bw) commit(buf ^ 1, hidx);
                SSTAMP(5);
                phase2(2 * (NPAIR - 2), std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            } else {
                // helpers 2, 3 commit BEFORE their pair: it completes at 55 / 73 % of the window, the prefetch has landed by then, and behind the
                // pair's signal only the dQ pass is left (commit behind it: the window ended with that commit; 7 716 -> 7 473 cycles per window)
                const bool cf = hw >= SWV2_ATTNS_COMMIT_FIRST;
                if (cf && bw_next < Bw) commit(buf ^ 1, hidx);
                phase2(2 * hw, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                if (!cf && bw_next < Bw) commit(buf ^ 1, hidx);
                SSTAMP(5);
                if (hw == 0) phase2(2 * (NPAIR - 1), std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            }
#endif
        }
#endif
#ifdef SWV2_ATTNS_ARRIVE
        const bool arr_on = blockIdx.x == 3 && blockIdx.y == 0 && it >= 8 && it < 16;
        if (arr_on && lane == 0) attns_arr[tw * 8 + it - 8] = __builtin_amdgcn_s_memtime();
#endif
#if !SWV2_ATTNS_NOBAR
        __syncthreads();
#endif
#ifdef SWV2_ATTNS_ARRIVE
        if (arr_on && tid == 0) attns_arr[16 * 8 + it - 8] = __builtin_amdgcn_s_memtime();
#endif
#if SWV2_ATTNS_PF2
        kv = kv == 2 ? 0 : kv + 1;
#endif
        SSTAMP(6);                      // the window's barrier
#ifdef SWV2_ATTNS_STAMPS
        if (lane == 0 && tw == 8 && blockIdx.y == 0 && blockIdx.x < 64 && it < 128) attns_win[blockIdx.x * 128 + it] = st_prev - ck0;
#endif
    }
    };
#if SWV2_ATTNS_NOBAR
    if (helper) {
        // (behind the barrier, and the registers defined on both paths: their prologue values are then dead at the role split -- they were spilled there)
        if (bw + (int)gridDim.x < Bw) issue(bw + gridDim.x, 1, hw * 64 + lane);
        else { s0 = Stg{}; s1 = Stg{}; }
        windows(std::integral_constant<int, 1>{});
    } else windows(std::integral_constant<int, 0>{});
#else
    windows(std::integral_constant<int, 0>{});
#endif
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

#ifdef SWV2_ATTNS_ARRIVE
extern "C" int swv2_debug_attns_arrive(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_arr), sizeof(unsigned long long) * 17 * 8) == hipSuccess ? 0 : -3;
}
#endif
#ifdef SWV2_ATTNS_STAMPS
extern "C" int swv2_debug_attns_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_stamps), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -3;
}
extern "C" int swv2_debug_attns_win(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_win), sizeof(unsigned long long) * 64 * 128) == hipSuccess ? 0 : -3;
}
extern "C" int swv2_debug_attns_tl(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attns_tl), sizeof(unsigned long long) * 16 * 24) == hipSuccess ? 0 : -3;
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
