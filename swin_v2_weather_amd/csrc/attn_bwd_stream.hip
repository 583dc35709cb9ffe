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
//   * waves 11 .. 15, helpers: each prefetches a fifth of the next window's slabs (k, v by LDS-DMA, the rest through registers), commits
//     its chunks (with delta = rowsum(dO O) and the statistics slots) to the OTHER LDS buffer, and computes the dQ of the first q-tiles
//     (helper 0 the first pair, helpers 1 .. 3 tiles 2 .. 4) -- spinning on the pair's counter until it shows all 11 phase-1 waves,
//     then reading the dS image transposed, every fragment before the first product (the two-phase kernel's summation order);
//   * the dQ of the last six q-tiles is computed by the phase-1 waves that finish their key tile first (one single-tile pass each
//     behind their dK / dV stores): with every dQ pass on the helpers the window's barrier waited ~1 000 cycles for the last pair's
//     pass while the phase-1 waves idled (tools/probe_attn_bwd_arrive.py);
//   * the q | dO | k | v slabs (and the 1 / |q|, 1 / |k| rows) are double-buffered in LDS, so the commit needs no barrier of its own;
//     the dS image is single (every reader has passed the window's ONE barrier before the next window's first tile is written);
//   * issue priority (s_setprio) falls with a phase-1 wave's own progress (the older waves of a SIMD one stage earlier), so the three
//     phase-1 waves of a SIMD advance together instead of oldest first.
// Measured with tools/probe_attn_bwd_windows.py (s_memtime at every window's end, same box): 8 740 cycles per window for the two-phase
// kernel, 7 480 - 7 520 with all dQ passes on the helpers, the same at 25 and at 100 windows per workgroup; barrier to barrier in a
// production build (tools/probe_attn_bwd_arrive.py) 6 960 -> 6 620 with the last tiles on the phase-1 waves; d(qkv) bit-identical.  What was built on the way and
// lost (LABNOTES round 6): 11 waves with the dQ tiles inside every wave's own loop (8 560) or on the two waves of the short SIMD (one
// tile at a time: slower than two-phase -- a tile's chain of transposed reads -> 6 dependent MFMAs -> row sum -> store is ~400 cycles of
// latency); the last three q-tiles' dQ deferred into the next window through a second tail image (tools/experiments/
// attn_bwd_stream_deferred.hip: 7 620, the commits then wait for those tiles); helpers at priority 2 / 3 (7 620 - 7 670); scale and statistics
// through the MFMA's C operand, prefetch distance two, no barrier at all (tools/experiments/attn_bwd_stream_nobar.hip, LABNOTES "second pass").
// The phase-1 loop is ~270 issue cycles per q-tile pair by the instruction costs of MI355X_MICROARCH (8 exp, 8 + 4 multiplies, 8 packs,
// 6 MFMAs, 9 LDS instructions, the signal): ~5 000 per window and SIMD -- the kernel runs at two thirds of its own issue bound.
// LDS: 2 x 40 832 (slabs) + 63 360 (dS image) + counters = 145 KB, one persistent workgroup per CU as before.
#include <stdlib.h>

#include "attn_common.h"

namespace {

typedef __attribute__((address_space(3))) unsigned lds_u32;

#ifdef SWV2_ATTNS_ARRIVE          // diagnostic build (tools/probe_attn_bwd_arrive.py): when each wave of ONE workgroup reaches the window's barrier and when it opens
__device__ unsigned long long attns_arr[17 * 8];      // (8 windows; nothing else is instrumented: the kernel runs as in production)
#endif

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

#ifndef SWV2_ATTNS_NO_PRIO          // (A/B builds)
#define SWV2_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define SWV2_PRIO(n) do {} while (0)
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
    constexpr int OFF_DS = 2 * BUFB, OFF_CNT = OFF_DS + Lp * DSP * 2, OFF_RED = OFF_CNT + 32, LDS_BYTES = OFF_RED + ((WAVES * 4 + 15) / 16) * 16;
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

    if (tid < 8) ((unsigned*)(lds + OFF_CNT))[tid] = 0u;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_u32*)lds;
    const unsigned cnt_addr = lds_base + OFF_CNT;           // the counters' LDS address (for the inline ds_add / ds_read)

    // ---- staging (helper waves): one 16-byte chunk of the q, k, v, dO, o slabs; lse of the chunk's row; one 1/|.| value -- twice on helper 0
    // (k and v go straight to LDS by DMA: 44 staging registers beside the dQ pair did not fit the 128 of a 16-wave workgroup)
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
            st.q = *(const uint4*)(qb + o16);
            st.dO = *(const uint4*)(dob + o16);
            st.o = *(const uint4*)(ob + o16);
            st.lse = *(const float*)((const char*)(lse + item * Lp) + ol);
            st.rn = *(const float*)((const char*)(rnorm + item * 2 * Lp) + orn);
        };
        const unsigned lds_k0 = lds_base + (unsigned)(nbuf * BUFB + B_K);
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
    }
    __syncthreads();

    float dsig = 0.f;
    SSTAMP_DECL
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
        // dQ of q-tiles qt0 .. qt0 + NQ - 1 (NQ = 2: a pair, the K^T fragments are read once for both): dQ^T = sum_t K_t^T dS_t^T, key tiles in pairs (one
        // K = 32 product per pair), both operands as transposed reads; one accumulation chain per q-tile in key order + the odd key tile on its own
        // accumulator: the summation order of the two-phase kernel (bit-identical d q).  EVERY fragment is read before the first product: a pass is one
        // wave's chain, and with the reads interleaved it paid an LDS round trip per product (~970 -> ~620 cycles for one tile behind the last signals)
        auto phase2 = [&](const int qt0, auto nq_c, auto sleep_c) {
            constexpr int NQ = decltype(nq_c)::value;
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
            const uint16_t* const kb = Ks + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4;
            const uint16_t* const db = dSb + (4 * g + (fr >> 2)) * DSP + 16 * qt0 + (fr & 3) * 4;
            bf16x4 kk[LT], dd[NQ][LT];
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                kk[t] = lds_tr_read(kb + 16 * t * DP);
#pragma unroll
                for (int i = 0; i < NQ; ++i) dd[i][t] = lds_tr_read(db + 16 * t * DSP + 16 * i);
            }
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
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int q = 16 * qt0 + 16 * i + fr;
                const float rq = RN[q] * sigma;
                const bf16x4 qn = *(const bf16x4*)(Qa + q * QP + 4 * g);
                float dot = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][r], bf2f(qn[r]), dot);
                dot = xor32_allsum(xor16_allsum(dot));
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rq * (dq[i][r] - bf2f(qn[r]) * dot);
                *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 4 * g) = f2bf4(v);
            }
            SSTAMP(3);
        };
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
                const bf16x8 td = __builtin_shufflevector(i0.td, i1.td, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 tq = __builtin_shufflevector(i0.tq, i1.tq, 0, 1, 2, 3, 4, 5, 6, 7);
                dv = mfma32(td, pb, dv);
                dk = mfma32(tq, dsb, dk);
            };
            // Issue priority falls with the wave's own progress (3 at the start of the window, 0 behind the fourth pair): the SIMD arbiter serves
            // the oldest wave of the highest priority first, so without this the oldest wave of a SIMD runs ahead, ends at ~60 % of the window,
            // and the youngest finishes alone with nothing to cover its latencies.  A wave that is behind now outranks one that is ahead
            // (same box, cycles per window: 8 099 without, 7 716 with; helpers at priority 2 / 3 on top: 7 620 - 7 670 against 7 473 - 7 516).
            // The older waves of a SIMD -- served first at equal priority -- step down one stage earlier than the younger ones (stage = pairs signalled;
            // waves 8 .. 10: 3 3 2 1 1 0, waves 4 .. 7: 3 2 2 1 0 0, waves 0 .. 3: 3 2 1 1 0 0; in situ 86.3 / 86.7 -> 85.5 / 85.4 us)
            const int cls = tw >> 2;
            auto setp = [&](auto s_c) {
                constexpr int sg = decltype(s_c)::value;
                constexpr int p2 = 3 - (2 * sg) / 3, p1 = 3 - (2 * sg + 1) / 3, p0 = (3 - (2 * sg + 2) / 3) < 0 ? 0 : 3 - (2 * sg + 2) / 3;
                if (cls == 2) SWV2_PRIO(p2); else if (cls == 1) SWV2_PRIO(p1); else SWV2_PRIO(p0);
            };
            setp(std::integral_constant<int, 0>{});
            St a0, a1, b0, b1;
            stageA(0, a0);
            stageA(1, a1);
#pragma unroll
            for (int qt = 0; qt + 1 < LT; qt += 4) {
                if (qt + 2 < LT) stageA(qt + 2, b0);
                if (qt + 3 < LT) stageA(qt + 3, b1);
                stageB2(qt, a0, a1);
                signal(qt >> 1);
                if (qt == 0) setp(std::integral_constant<int, 1>{}); else if (qt == 4) setp(std::integral_constant<int, 3>{}); else setp(std::integral_constant<int, 5>{});
                __builtin_amdgcn_sched_barrier(0);
                if (qt + 3 < LT) {
                    if (qt + 4 < LT) stageA(qt + 4, a0);
                    if (qt + 5 < LT) stageA(qt + 5, a1);
                    stageB2(qt + 2, b0, b1);
                    signal((qt >> 1) + 1);
                    if (qt == 0) setp(std::integral_constant<int, 2>{}); else setp(std::integral_constant<int, 4>{});
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
            // The dQ of the last six q-tiles, one single-tile pass each on the phase-1 waves that finish their key tile first (the four oldest + wave 7; wave 3,
            // alone with wave 7 and two helpers on its SIMD, takes two).  Which waves the window's barrier waits for, production build with one s_memtime per
            // wave and window (-DSWV2_ATTNS_ARRIVE, tools/probe_attn_bwd_arrive.py): with all dQ on the helpers the phase-1 waves arrived 900 - 2 600 cycles
            // before it opened and helper 4 (the last pair, behind the last signals: a ~1 700-cycle pair pass) 114; now every wave arrives within ~900.
            // 6 960 -> 6 620 cycles per window, in situ 89.3 -> 86.5 us.
            {
                const int tt = tw == 3 ? 6 : tw == 7 ? 7 : tw == 0 ? 8 : tw == 2 ? 9 : tw == 1 ? 10 : -1;
                if (tt >= 0) {
                    SWV2_PRIO(2);
                    if (tw == 3) phase2(5, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                    phase2(tt, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                }
            }
        } else {
            // ================= helper waves: the next window's prefetch, phase 2 (dQ), the commit =================
            const int hidx = hw * 64 + ln;
            if (bw_next < Bw) issue(bw_next, buf ^ 1, hidx);
            SSTAMP(0);
            // every helper commits first (its loads were issued at the window's start), then helper 0: the first pair, helpers 1 .. 3: q-tiles 2 .. 4
            // (all complete by ~55 % of the window); helper 4 stages and commits two chunks per thread and takes no tile
            if (bw_next < Bw) commit(buf ^ 1, hidx);
            SSTAMP(5);
            if (hw == 0) phase2(0, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            else if (hw < HW - 1) phase2(hw + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
        }
#ifdef SWV2_ATTNS_ARRIVE
        const bool arr_on = blockIdx.x == 3 && blockIdx.y == 0 && it >= 8 && it < 16;
        if (arr_on && lane == 0) attns_arr[tw * 8 + it - 8] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef SWV2_ATTNS_ARRIVE
        if (arr_on && tid == 0) attns_arr[16 * 8 + it - 8] = __builtin_amdgcn_s_memtime();
#endif
        SSTAMP(6);                      // the window's barrier
#ifdef SWV2_ATTNS_STAMPS
        if (lane == 0 && tw == 8 && blockIdx.y == 0 && blockIdx.x < 64 && it < 128) attns_win[blockIdx.x * 128 + it] = st_prev - ck0;
#endif
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
