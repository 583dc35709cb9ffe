// Third build of the window-attention BACKWARD for head dims <= 16 without CPB bias: one WAVE owns one (window, head).
// Opt-in (`dbg` bit 8) and parity-tested; the two-phase kernel of attn.hip stays the default (see swv2_attn3_bwd below).
//
// Follows the autograd of /root/reference/networks/swinv2_global.py:298-321 (cosine attention: normalised q, k, clamped
// learnt logit scale, shift mask, softmax, P V) like attn.hip's two-phase kernel.  Why it was built: the two-phase kernel
// spreads a window's 11 key tiles over 11 waves of one workgroup and meets at three barriers per window; its phase stamps
// (profiles/r02_stamps_attn_bwd.txt) show the waves waiting for the youngest wave of the most loaded SIMD (11 waves on 4
// SIMDs), the LDS as the busiest unit (Q / dO fragments re-read by every wave) and the MFMA pipe at 22 %.  Here nothing is
// shared between waves, so there is no barrier at all:
//   * the wave keeps the row-form fragments of Q and dO of all query tiles (MFMA A operands) and the dQ accumulators of
//     all query tiles in registers for the life of the (window, head), loops over the key tiles, and inside that over the
//     query tiles (fully unrolled: the accumulators are statically indexed registers);
//   * the softmax statistics ride in the unused half of the K = 32 MFMA operands (head dim 16 fills k = 0..15): the A
//     operand carries lse / (sigma log2 e) and delta as three bf16 parts (exact to 2^-24), a constant 1 and the query's
//     mask-region flags in k = 16..21, the B operand carries -1, -1, -1, the padded-key flag (-1e30) and the key's mask
//     terms, so the MFMA result IS (S - lse') and (dP - delta), masked -- no LDS reads of statistics, no per-element
//     selects (the mask of the reference, -100 where the regions differ, is bilinear in the two region flags);
//   * two query tiles form one K = 32 operand for dV and dK (4 MFMAs per tile pair instead of 5); the query-tile pairs run
//     as a three-stage software pipeline (S / dP MFMAs | softmax backward + dV / dK | dQ);
//   * the only LDS traffic per tile pair is wave-private: the transposed fragments of Q / dO (`ds_read_b64_tr_b16` of the
//     wave's own slab image) and the dS tile's trip through a 640-byte scratch tile into the B operand of dQ;
//   * waves are persistent; K | V tiles arrive two tiles ahead by LDS-DMA into a three-slot ring, and the next unit's cache
//     lines are touched by LDS-DMA loads into a dummy tile (no registers: all 256 are in use).
// Per 16 x 16 score tile the vector ALU is left with one multiply, exp2, one multiply and the bf16 packs.
// Measured (tools/probe_attn3_stamps.py, profiles/r02_stamps_attn_bwd3.txt): the query-tile steps run within 1.5x of their
// vector-ALU bound, but they are only 43 % of a unit: the per-unit prologue (22 KB through three dependent L2 round trips,
// operand construction), key-tile set-up and epilogue are latency that two waves per SIMD cannot hide, and 6 400 units on
// 2 048 wave slots quantise to 4 rounds (the remainder is handed to the two-phase kernel instead).
#include "attn_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // native vector: arrays of HIP uint4 indexed in lambdas go to scratch

__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// LDS-DMA loads (destination: LDS at `lds_addr` + lane * size, no register).  Inline assembly on purpose: through the
// builtin the compiler orders every later LDS access behind the load with s_waitcnt vmcnt(0) (it cannot tell that the wave's
// other LDS tiles do not alias the destination).  The compiler does not count these in its own vmcnt bookkeeping; VMEM
// operations return in order, so an uncounted operation can only make a compiler-placed wait longer, never too short, and
// the waits for the DMA'd tiles themselves are placed by hand (`vm_wait`).
__device__ __forceinline__ void dma_x4(const void* base, uint32_t lane_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_x1(const void* base, uint32_t lane_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(lane_off), "s"(base), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// v = hi + lo + lolo with three bf16 parts (|error| <= 2^-24 |v|)
__device__ __forceinline__ void split3(float v, uint16_t& hi, uint16_t& lo, uint16_t& lolo) {
    hi = f2bf(v);
    const float r1 = v - bf2f(hi);
    lo = f2bf(r1);
    lolo = f2bf(r1 - bf2f(lo));
}

#ifndef SWV2_A3_ABL
#define SWV2_A3_ABL 0       // timing ablations of tools/probe_attn3_stamps.py (wrong results)
#endif
#ifdef SWV2_ATTN3_STAMPS
__device__ unsigned long long attn3_stamps[1024 * 8];       // tools/probe_attn3_stamps.py: per-wave cycle sums of the phases
#define A3ACC(v) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); v += n_ - a3t; a3t = n_; } while (0)
#define A3PUT(k, v) do { if (lane == 0 && blockIdx.x * 4 + wv < 1024) attn3_stamps[(blockIdx.x * 4 + wv) * 8 + (k)] = (v); } while (0)
#else
#define A3ACC(v) do {} while (0)
#define A3PUT(k, v) do {} while (0)
#endif

template <int LT>
__global__ __launch_bounds__(256, 2) void attn_bwd3_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint16_t* __restrict__ oh,
    const uint16_t* __restrict__ doh, const float* __restrict__ lse, const float* __restrict__ rnorm,
    uint16_t* __restrict__ dqkvh, float* __restrict__ dlogit, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int Lp = 16 * LT, DP = 16, SLAB = Lp * DP;
    constexpr int DSP = 20;                                   // row pitch (elements) of the dS scratch tiles
    constexpr int W_Q = 0, W_DO = SLAB * 2, W_K = 2 * SLAB * 2, W_DS = W_K + 3 * 1024, W_RN = W_DS + 2 * 16 * DSP * 2,
                  W_TOUCH = W_RN + 2 * Lp * 4, W_BYTES = W_TOUCH + 256;
    static_assert(W_BYTES % 16 == 0, "16-byte aligned wave regions");
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * W_BYTES + 16];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: the slab bases below live in SGPRs
    unsigned char* const my = lds + wv * W_BYTES;
    uint16_t* const Qs = (uint16_t*)(my + W_Q);
    uint16_t* const dOs = (uint16_t*)(my + W_DO);
    uint16_t* const KV = (uint16_t*)(my + W_K);          // ring of three [K tile 16 x 16 | V tile 16 x 16] pairs (1 KB each)
    float* const RN = (float*)(my + W_RN);               // rnorm of the unit: [2][Lp]
    const uint32_t lds_kv = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(my + W_K);
    const uint32_t lds_touch = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(my + W_TOUCH);
    uint16_t* const DS0 = (uint16_t*)(my + W_DS);
    uint16_t* const DS1 = DS0 + 16 * DSP;

    // persistent waves: unit = (window, head) = bw * h + hd, a wave takes units u, u + nwaves, ...  While it works on one unit
    // it touches the cache lines of its NEXT unit (six 4-byte loads per lane spread over the key-tile loop, results
    // discarded): without that every wave of a round loads its 22 KB at the same moment -- 46 MB per round that nothing
    // overlaps, 16 k - 40 k cycles of prologue per 60 k-cycle unit -- with it the prologue reads from L2 / MALL while HBM
    // streams during the loops.
    const int nwaves = gridDim.x * 4, units = Bw * h;
    float dsig_acc = 0.f;
#ifdef SWV2_ATTN3_STAMPS
    unsigned long long a3t = __builtin_amdgcn_s_memtime(), a3pro = 0, a3top = 0, a3steps = 0, a3fin = 0, a3epi = 0;
    const unsigned long long a3start = a3t;
#endif
#pragma unroll 1
    for (int u = blockIdx.x * 4 + wv; u < units; u += nwaves) {
        // lane-derived values are re-derived per unit behind an opaque copy: hoisted out of this loop, the loop-invariant
        // LDS / global lane offsets of the unrolled body cost 166 spilled registers
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const int fr = lane_ & 15, g = lane_ >> 4;
        const int bw = u / h, hd = u - bw * h;
        const float tau = logit_scale[hd];
        const float sigma = __expf(fminf(tau, SWV2_LN100));
        const float sc2 = sigma * SWV2_LOG2E;
        const float inv_sc2 = 1.f / sc2;
        float dsig = 0.f;
        const size_t slab0 = (size_t)u * 3 * SLAB, oslab = (size_t)u * SLAB;
        const size_t stat0 = (size_t)u * Lp, rn0 = (size_t)u * 2 * Lp;
        const int un = u + nwaves;
        const bool has_next = un < units;
        const size_t unn = has_next ? (size_t)un : (size_t)u;
        auto touch = [&](const int k) {
            // k = 0..2: the q | k | v slabs, 3: dO, 4: O, 5: lse of the next unit; one 128-byte line per lane, destination a
            // dummy LDS tile: no register, and nothing ever waits for it
            const char* base = k < 3 ? (const char*)(qkvh + unn * 3 * SLAB) : (k == 3 ? (const char*)(doh + unn * SLAB) :
                               (k == 4 ? (const char*)(oh + unn * SLAB) : (const char*)(lse + unn * Lp)));
            const uint32_t nb = k < 3 ? 3 * SLAB * 2 : (k == 5 ? Lp * 4 : SLAB * 2);
            const uint32_t off = min((uint32_t)((k < 3 ? k : 0) * 8192 + lane_ * 128), nb - 4);
            dma_x1(base, off, lds_touch);
        };
        // K | V rows of key tile t -> ring slot t % 3: lanes 0..31 fetch the K tile (row = lane / 2, 16-byte half = lane & 1),
        // lanes 32..63 the V tile; the DMA puts lane l's 16 bytes at slot + 16 l, i.e. the two tiles row-major back to back
        const char* const kvb = (const char*)(qkvh + slab0 + SLAB);
        const uint32_t kvoff = (uint32_t)(lane_ >> 5) * (SLAB * 2) + (uint32_t)(lane_ & 31) * 16;
        auto kv_dma = [&](const int t) { dma_x4(kvb, kvoff + (uint32_t)min(t, LT - 1) * 512, lds_kv + (uint32_t)(t % 3) * 1024); };
        kv_dma(0);
        kv_dma(1);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;
        const int gh = g & 1;                                  // lanes g = 2, 3 repeat the loads of g = 0, 1 (unconditional loads)
        const u32x4 zero4 = {0u, 0u, 0u, 0u};

        // ---- query side: row-form A operands with the statistics in k = 16..23, slab images for the transposed reads
        bf16x8 qa[LT], da[LT];
        f32x4 dq[LT];
        {
            // tiles in chunks of CHK with the next chunk's loads in flight (all LT tiles at once needs 13 registers per tile
            // beside the operands being built: scratch spills, measured 38 k cycles).  Wave-uniform bases + one 32-bit lane
            // offset + immediate offsets: no 64-bit address registers.
            constexpr int CHK = 4, NCH = (LT + CHK - 1) / CHK;
            const char* const qb = (const char*)(qkvh + slab0);
            const char* const db = (const char*)(doh + oslab);
            const char* const ob = (const char*)(oh + oslab);
            const float* const lb = lse + stat0;
            const uint32_t loff = (uint32_t)(fr * DP + gh * 8) * 2;
            u32x4 rq[2][CHK], rd[2][CHK], ro[2][CHK];
            float rl[2][CHK];
            auto issue = [&](const int c, const int b) {
#pragma unroll
                for (int v = 0; v < CHK; ++v) {
                    const int i = min(c * CHK + v, LT - 1);
                    rq[b][v] = *(const u32x4*)(qb + loff + i * 512);
                    rd[b][v] = *(const u32x4*)(db + loff + i * 512);
                    ro[b][v] = *(const u32x4*)(ob + loff + i * 512);
                    rl[b][v] = lb[fr + 16 * i];
                }
            };
            issue(0, 0);
            {   // 1 / |q|, 1 / |k| of the unit -> LDS (no loads in the key-tile loop or behind it: see dma_x4)
                const float* const rb = rnorm + rn0;
#pragma unroll
                for (int c = 0; c < (2 * Lp + 63) / 64; ++c) {
                    const int i = min(c * 64 + lane_, 2 * Lp - 1);
                    RN[i] = rb[i];
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int b = c & 1;
                if (c + 1 < NCH) issue(c + 1, b ^ 1);
#pragma unroll
                for (int v = 0; v < CHK; ++v) {
                    const int i = c * CHK + v;
                    if (i >= LT) continue;
                    const int q = 16 * i + fr;
                    if (g < 2) {
                        *(u32x4*)(Qs + q * DP + g * 8) = rq[b][v];
                        *(u32x4*)(dOs + q * DP + g * 8) = rd[b][v];
                    }
                    // delta = sum_d dO O: this lane's 8 channels + the other half of the row
                    const u32x4 a = rd[b][v];
                    const u32x4 o = ro[b][v];
                    float dl = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(o[e] << 16), dl);
                        dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(o[e] & 0xffff0000u), dl);
                    }
                    dl += __shfl_xor(dl, 16);
                    const bool q_ok = q < L;
                    const float lq = q_ok ? rl[b][v] * inv_sc2 : 1.0e30f;       // padded query rows: P = 0
                    uint16_t l0, l1, l2, d0, d1, d2;
                    split3(lq, l0, l1, l2);
                    if (!q_ok) l1 = l2 = 0;
                    split3(dl, d0, d1, d2);
                    const uint32_t one = 0x3f80u, rqf = (q >= mask_thr) ? 0x3f80u : 0u;
                    const u32x4 augq = {l0 | ((uint32_t)l1 << 16), l2 | (one << 16), rqf | ((one - rqf) << 16), 0u};
                    const u32x4 augd = {d0 | ((uint32_t)d1 << 16), (uint32_t)d2, 0u, 0u};
                    qa[i] = __builtin_bit_cast(bf16x8, g < 2 ? rq[b][v] : (g == 2 ? augq : zero4));
                    da[i] = __builtin_bit_cast(bf16x8, g < 2 ? rd[b][v] : (g == 2 ? augd : zero4));
                    dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }

        A3ACC(a3pro);
        if (LT < 6)
            for (int k = LT; k < 6; ++k) touch(k);
        // ---- key tiles.  The K | V rows of tile j + 2 are DMA'd into the LDS ring during tile j; operands and the transposed
        // K fragment of tile j + 1 are read from the ring behind tile j's steps.
        vm_wait<0>();
        u32x4 rk = *(const u32x4*)(KV + fr * DP + gh * 8), rv = *(const u32x4*)(KV + 256 + fr * DP + gh * 8);
        bf16x4 kt = lds_tr_read(KV + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4);        // K[key 4g + r][d = fr]
        bf16x4 kn = *(const bf16x4*)(KV + fr * DP + 4 * g);                           // k^[key fr][d 4g + r]
#pragma unroll 1
        for (int j = 0; j < LT; ++j) {
            const uint16_t* const Ktn = KV + ((j + 1) % 3) * 512;
            const int key = 16 * j + fr;
            const uint32_t m1 = 0xbf80u;                                       // -1
            const uint32_t padk = (key < L) ? 0u : (uint32_t)f2bf(-1.0e30f);
            const bool kreg = key >= mask_thr;
            const uint32_t mk0 = f2bf(kreg ? 0.f : cmask), mk1 = f2bf(kreg ? cmask : 0.f);
            const u32x4 augk = {m1 | (m1 << 16), m1 | (padk << 16), mk0 | (mk1 << 16), 0u};
            const u32x4 augv = {m1 | (m1 << 16), m1, 0u, 0u};
            const bf16x8 kf = __builtin_bit_cast(bf16x8, g < 2 ? rk : (g == 2 ? augk : zero4));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, g < 2 ? rv : (g == 2 ? augv : zero4));
            kv_dma(j + 2);                                  // 2 uncounted VMEM operations per tile (+ the 2 stores below)
            touch(min(j, 5));                               // tiles 6.. touch the lse line again (L2 hit)
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            A3ACC(a3top);

            auto softmax_bwd = [&](const f32x4& s, const f32x4& dp, bf16x4& pb, bf16x4& dsb) {
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if SWV2_A3_ABL == 1 || SWV2_A3_ABL == 5
                    p[r] = s[r] * sc2;
#else
                    p[r] = __builtin_amdgcn_exp2f(s[r] * sc2);
#endif
                    ds[r] = p[r] * dp[r];
                }
#if SWV2_A3_ABL == 5
                pb = __builtin_bit_cast(bf16x4, (f32x2){p[0], p[1]});
                dsb = __builtin_bit_cast(bf16x4, (f32x2){ds[0], ds[1]});
#else
                pb = f2bf4(p);
                dsb = f2bf4(ds);
#endif
            };
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            // Software pipeline over the query-tile pairs, three stages in flight so that no stage waits for the LDS round
            // trip or the MFMA latency of the one before it (as one piece per pair the wave ran MFMA -> exp -> LDS write ->
            // transposed read -> MFMA strictly in series: 3 LDS latencies per pair, 122 us):
            //   A(p): S and dP MFMAs of the pair, transposed Q / dO fragment reads issued
            //   B(p): softmax backward on the vector ALU, dS tiles -> scratch, dV / dK MFMAs, transposed dS reads issued
            //   C(p): dQ MFMAs
            // issued as A(p + 1), B(p), C(p - 1).  The single scratch tile pair is safe: LDS operations of one wave execute
            // in order, and the reads of C(p - 1) were issued in B(p - 1).
            f32x4 dvt = z4, dkt = z4;
            struct StA { f32x4 s0, s1, p0, p1; bf16x4 td0, td1, tq0, tq1; };
            struct StT { bf16x4 t0, t1; };
            auto stA = [&](const int i, const bool two, StA& o) {
                o.s0 = mfma32(qa[i], kf, z4);
                o.p0 = mfma32(da[i], vf, z4);
#if SWV2_A3_ABL == 3 || SWV2_A3_ABL == 4
                o.td0 = o.tq0 = o.td1 = o.tq1 = kt;
#else
                o.td0 = lds_tr_read(dOs + (16 * i + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                o.tq0 = lds_tr_read(Qs + (16 * i + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
#endif
                if (two) {
                    const int i1 = min(i + 1, LT - 1);
                    o.s1 = mfma32(qa[i1], kf, z4);
                    o.p1 = mfma32(da[i1], vf, z4);
#if !(SWV2_A3_ABL == 3 || SWV2_A3_ABL == 4)
                    o.td1 = lds_tr_read(dOs + (16 * i + 16 + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                    o.tq1 = lds_tr_read(Qs + (16 * i + 16 + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
#endif
                }
            };
            auto stB = [&](const bool two, const StA& in, StT& t) {
                bf16x4 pb0, pb1, db0, db1;
                softmax_bwd(in.s0, in.p0, pb0, db0);
#if !(SWV2_A3_ABL == 2 || SWV2_A3_ABL == 4)
                *(bf16x4*)(DS0 + fr * DSP + 4 * g) = db0;              // image [key][q] of the dS tile(s)
#endif
                if (two) {
                    softmax_bwd(in.s1, in.p1, pb1, db1);
#if !(SWV2_A3_ABL == 2 || SWV2_A3_ABL == 4)
                    *(bf16x4*)(DS1 + fr * DSP + 4 * g) = db1;
#endif
                    dv = mfma32(cat8(in.td0, in.td1), cat8(pb0, pb1), dv);       // dV^T += dO^T P over both query tiles
                    dk = mfma32(cat8(in.tq0, in.tq1), cat8(db0, db1), dk);       // dK^T += Q^T dS
                } else {
                    // own accumulators: a K = 16 MFMA chained directly onto a K = 32 accumulator gave wrong sums (attn.hip)
                    dvt = mfma16(in.td0, pb0, z4);
                    dkt = mfma16(in.tq0, db0, z4);
                }
#if SWV2_A3_ABL == 2 || SWV2_A3_ABL == 4
                t.t0 = db0;
                if (two) t.t1 = db1;
#else
                t.t0 = lds_tr_read(DS0 + (4 * g + (fr >> 2)) * DSP + (fr & 3) * 4);      // dS^T[key 4g + r][q fr]
                if (two) t.t1 = lds_tr_read(DS1 + (4 * g + (fr >> 2)) * DSP + (fr & 3) * 4);
#endif
            };
            auto stC = [&](const int i, const bool two, const StT& t) {
                dq[i] = mfma16(kt, t.t0, dq[i]);                       // dQ^T += K^T dS^T
                if (two) dq[min(i + 1, LT - 1)] = mfma16(kt, t.t1, dq[min(i + 1, LT - 1)]);
            };
            constexpr int NS = (LT + 1) / 2;                           // pipeline steps: LT / 2 pairs (+ one single tile)
            StA sa[2];
            StT st[2];
            stA(0, LT > 1, sa[0]);
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                if (p + 1 < NS) stA(2 * p + 2, 2 * p + 3 < LT, sa[(p + 1) & 1]);
                stB(2 * p + 1 < LT, sa[p & 1], st[p & 1]);
                if (p > 0) stC(2 * p - 2, true, st[(p - 1) & 1]);
            }
            stC(2 * NS - 2, 2 * NS - 1 < LT, st[(NS - 1) & 1]);
            if (LT & 1) {
                dv += dvt;
                dk += dkt;
            }

            A3ACC(a3steps);
            // the transposed K fragment of tile j + 1 through the other LDS tile: on its way during the finalisation below
            // tile j + 1 was requested during tile j - 1; behind it in the (in-order) queue: touch (j - 1), the two stores of
            // tile j - 1, tile j + 2, touch (j) -- checked against the ISA (two global stores per tile, no other VMEM)
            vm_wait<5>();
            rk = *(const u32x4*)(Ktn + fr * DP + gh * 8);
            rv = *(const u32x4*)(Ktn + 256 + fr * DP + gh * 8);
            kt = lds_tr_read(Ktn + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4);      // (read into the loop-carried register itself:
                                                                                  //  a copy at the loop end would wait for it there)
            // ---- dK through the L2-normalisation backward, dV; d logit_scale = sigma sum_k (sum_q dS q^) . k^
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dk[r], bf2f(kn[r]), dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (g == 0) dsig += dot;
            const float rks = RN[Lp + key] * sigma;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rks * (dk[r] - bf2f(kn[r]) * dot);
            *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 4 * g) = f2bf4(v);
            *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 4 * g) = f2bf4(dv);
            kn = *(const bf16x4*)(Ktn + fr * DP + 4 * g);
            A3ACC(a3fin);
        }

        // ---- dQ through the normalisation backward
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            const int q = 16 * i + fr;
            const bf16x4 qn = *(const bf16x4*)(Qs + q * DP + 4 * g);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][r], bf2f(qn[r]), dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            const float rqs = RN[q] * sigma;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rqs * (dq[i][r] - bf2f(qn[r]) * dot);
            *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 4 * g) = f2bf4(v);
        }

        A3ACC(a3epi);
        // ---- d logit_scale: one atomic per wave and head (a wave keeps its head while nwaves % h == 0)
        dsig_acc += dsig * sigma;
        const bool flush = !has_next || (un % h) != hd;
        if (flush) {
            const float t = wave_sum(dsig_acc);
            if (lane == 0 && tau <= SWV2_LN100) atomicAdd(dlogit + hd, t);
            dsig_acc = 0.f;
        }
    }
    A3PUT(0, a3pro); A3PUT(1, a3top); A3PUT(2, a3steps); A3PUT(3, a3fin); A3PUT(4, a3epi);
    A3PUT(5, __builtin_amdgcn_s_memtime() - a3start);
}

template <int LT>
int launch_bwd3(const swv2_attn_args* a, int Bw, hipStream_t st) {
    const int units = Bw * a->heads;
    // two workgroups (eight waves) per CU; not more workgroups than there are units
    dim3 grid(units >= 2048 ? 512 : (units + 3) / 4), block(256);
    hipLaunchKernelGGL((attn_bwd3_kernel<LT>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale, (const uint16_t*)a->oh,
                       (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, Bw, a->heads, a->L,
                       a->nwh * a->nww, a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_bwd (wave per head)");
    return SWV2_OK;
}

}  // namespace

#ifdef SWV2_ATTN3_STAMPS
extern "C" int swv2_debug_attn3_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attn3_stamps), sizeof(attn3_stamps)) == hipSuccess ? 0 : -3;
}
#endif

int swv2_attn1_bwd_range(const swv2_attn_args* a, int w0, void* stream);

// 0 / negative = handled (ok / error), 1 = shape not covered (CPB bias, head dim > 16)
int swv2_attn3_bwd(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    // Opt-in (dbg bit 8): measured 101.7 us for three full rounds + 11.6 us for the remainder on the two-phase kernel = 113 us at
    // the benchmark shape, against 115 us for the two-phase kernel alone -- not worth making a kernel with hand-placed vmcnt
    // waits the default ("wave per head").  Parity-tested in both modes.
    if (a->bias || DP != 16 || !(a->dbg & 256)) return 1;
    if (Lp != 176 && Lp != 64) return 1;
    // A unit (window, head) is one wave's work for ~22 us and the chip holds 2048 waves: 6400 units (the benchmark: 800
    // windows x 8 heads) are 3.125 rounds, and the last eighth of a round costs a whole one.  The units beyond the last full
    // round go to the two-phase kernel (fine-grained: 11 waves per unit) when they are less than half a round.
    constexpr int SLOTS = 2048;
    const int units = a->Bw * a->heads, rounds = units / SLOTS, rem = units - rounds * SLOTS;
    int Bw3 = a->Bw;
    if (rounds >= 1 && rem > 0 && rem <= SLOTS / 2 && SLOTS % a->heads == 0 && !(a->dbg & 512)) Bw3 = rounds * SLOTS / a->heads;
    int rc = Lp == 176 ? launch_bwd3<11>(a, Bw3, st) : launch_bwd3<4>(a, Bw3, st);
    if (rc || Bw3 == a->Bw) return rc;
    return swv2_attn1_bwd_range(a, Bw3, stream);
}
