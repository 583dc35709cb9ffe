// Cosine window attention core, forward at the benchmark head geometry (gfx950 / CDNA4): small out-of-phase workgroups.
//
// Same semantics and data layout as attn.hip (reference networks/swinv2_global.py:298-318 and :170-198):
//   S = sigma_h * qn kn^T + Mask ; P = softmax(S) ; O = P v        (16-wide heads, no CPB bias, windows of 65 .. 176 tokens)
//
// Why a second file.  The first kernels (attn.hip) run ONE workgroup of LT (= 11) waves per CU, wave = one 16-row tile, with
// workgroup barriers per window: all waves of a CU sit in the same phase (MFMA, then softmax on the vector ALU, then MFMA), so
// the matrix pipe and the vector pipe never overlap.  Here a workgroup is 4 waves (one per SIMD) and 3 workgroups share a CU
// (168 VGPRs, <= 48 KB LDS each), each on its own (window, head) item, so MFMA phases of one workgroup overlap the softmax of
// another; a wave owns several 16-row tiles of its item (q tiles w, w + 4, w + 8), which amortises the operand-fragment reads.
// Measured instruction costs behind the design (tools/ubench.hip, LABNOTES.md): v_exp_f32 8.3 cycles per wave-instruction at
// any occupancy; v_fma 2.0 at >= 3 waves per SIMD; K = 16 and K = 32 MFMAs both 16 cycles.
// The other members of this generation (a bias-capable forward and a backward in the same 4-wave shape) lost their A/B against
// the first-generation kernels and live in tools/experiments/attn2_experiments.hip, outside the shipped library.
#include <type_traits>

#include "attn_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// In-kernel phase timing (diagnostic builds only, -DSWV2_ATTN_STAMPS, tools/probe_attn_stamps.py): wave 0 of every workgroup
// accumulates s_memtime deltas per phase and leaves them in the padded tail of its first item's lse row.  `dep` orders the
// stamp behind the value that ends the phase.
#ifdef SWV2_FWD3_SPAN           // diagnostic build (tools/probe_attn_fwd.py): per workgroup (wave 0) the s_memtime / s_memrealtime span of the item loop
__device__ unsigned long long fwd3_span[2048 * 2];
#endif
#ifdef SWV2_ATTN_STAMPS
#define STAMP_DECL unsigned long long st_prev = 0, st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define STAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define STAMP(k, dep) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); \
                           st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMP_DECL
#define STAMP_START() do {} while (0)
#define STAMP(k, dep) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// forward, third form (head_dim <= 16, no CPB bias): the softmax's per-element vector work is cut from
// max + fma + exp + add + 1/2 cvt to exp + 1/2 cvt by letting the MFMAs do the rest:
//   * scale: sigma log2(e) q^ is split into two bf16 parts (hi + lo, exact to ~2^-17) that fill the K = 32 operand against
//     the key tile repeated twice -- the K = 32 MFMA costs the same 16 cycles as the K = 16 one, so S' = sigma' cos comes
//     out of the matrix pipe already scaled, at fp32-product accuracy;
//   * max: cosines are bounded, S' <= sigma', so the accumulator is started at -sigma' and exp2 is applied to the MFMA
//     result directly ("fixed maximum"); P then spans [2^(-2 sigma'), 1], representable in bf16 / fp32 while
//     sigma' <= 40 (sigma <= 27.7; tau starts at ln 10).  Larger scales and shift-masked windows take the general path
//     (row maximum on the vector ALU);
//   * row sum: one more K = 32 MFMA per key-tile pair with an all-ones A operand (the matrix pipe idles most of the time).
//     The normaliser is then the sum of the bf16-rounded P -- exactly the values that multiply V.
// Measured before (tools/probe_attn_stamps.py): ~259 vector instructions per 16-query row at ~5.7 SIMD cycles each.
// ------------------------------------------------------------------------------------------------
// KREG: the wave keeps the K (A operand of S^T) and V^T (A operand of O^T) fragments of the whole item in registers (66
// VGPRs), read from LDS once per item and pinned there (an empty asm makes the values opaque: the compiler otherwise sinks the
// loop-invariant reads back into the row loop, where -- double-buffered in 8 registers -- every QK MFMA waits a full LDS
// round trip: 11 x ~80 cycles per row against 11 x 16 for back-to-back MFMAs; measured on one wave per SIMD: ~2000 cycles
// per row of pure compute).
template <int LT, int LFIX, int WAVES, int OCC, bool KREG>
__global__ __launch_bounds__(64 * WAVES, OCC) void attn_fwd3_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, uint16_t* __restrict__ oh, float* __restrict__ lse,
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr, int dbg) {
    constexpr int DP = 16, Lp = 16 * LT, SLAB = Lp * DP;
    constexpr int NT = 64 * WAVES;
    constexpr int CH = 2 * SLAB / 8;                  // 16-byte chunks of the K | V slabs (a multiple of 64)
    constexpr int CPT = (CH + NT - 1) / NT;
    // LDS per buffer: K image with every 4-value group stored twice ([key][g][d 4g.. | d 4g..], 64 B per key: the A operand of
    // the K = 32 product against (hi | lo) of sigma' q^ is then ONE 16-byte read per lane), then the V slab as it is
    // ... and the Q slab: the row loop must not wait on any global load (a vmcnt wait inside it also waits for the row's own
    // O / lse stores and for the next item's prefetch -- one full memory round trip per row, measured)
    constexpr int KIMG = 2 * SLAB, BUF = KIMG + 2 * SLAB;
    constexpr int QCH = SLAB / 8, QPT = (QCH + NT - 1) / NT;   // 16-byte chunks of the Q slab / per thread
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const bool last_chunk_ok = (wave * 64 + (CPT - 1) * NT) < CH;          // wave-uniform
    const int Lc = LFIX > 0 ? LFIX : L;

    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;
    const bool bounded = sc2 <= 40.f;

    u32x4 stage[CPT], stageq[QPT];
    // 32-bit, loop-invariant lane offsets against a wave-uniform base: the loads then take the (SGPR base + VGPR offset)
    // form.  With 64-bit per-lane addresses the compiler built them in the loads' own destination registers, guarded that
    // overwrite with s_waitcnt vmcnt(..0) -- and with the in-order counter that wait also covered the previous item's
    // output STORES, once per (window, head) item in front of the prefetch (ISA, round 2).
    unsigned soff[CPT], qoff[QPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) soff[j] = (unsigned)(SLAB + min(tid + j * NT, CH - 1) * 8);
#pragma unroll
    for (int j = 0; j < QPT; ++j) qoff[j] = (unsigned)(min(tid + j * NT, QCH - 1) * 8);
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) stage[j] = *(const u32x4*)(base + soff[j]);
#pragma unroll
        for (int j = 0; j < QPT; ++j) stageq[j] = *(const u32x4*)(base + qoff[j]);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (j < CPT - 1 || last_chunk_ok) {
                const int c = tid + j * NT;                      // chunk c: 8 values, K chunks first (2 per key row), then V
                if (c < CH / 2) {
                    const u32x4 lo = {stage[j][0], stage[j][1], stage[j][0], stage[j][1]};
                    const u32x4 hi = {stage[j][2], stage[j][3], stage[j][2], stage[j][3]};
                    *(u32x4*)(dst + (size_t)c * 16) = lo;        // key c / 2, groups 2 (c & 1) and 2 (c & 1) + 1: 32 B per chunk
                    *(u32x4*)(dst + (size_t)c * 16 + 8) = hi;
                } else {
                    *(u32x4*)(dst + KIMG + (size_t)(c - CH / 2) * 8) = stage[j];
                }
            }
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (tid + j * NT < QCH) *(u32x4*)(dst + KIMG + SLAB + (size_t)(tid + j * NT) * 8) = stageq[j];
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    const int bw_first = bw;
    STAMP_DECL
    issue_loads(bw);
    write_stage(0);
    __syncthreads();

    const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
    STAMP_START();
#ifdef SWV2_ATTN_STAMPS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef SWV2_FWD3_SPAN
    const unsigned long long span_r0 = __builtin_amdgcn_s_memrealtime(), span_c0 = __builtin_amdgcn_s_memtime();
#endif

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const int bw_next = bw + gridDim.x;
        // the next item's K / V are requested AFTER this item's first row fragment has been waited for (a wait for a younger
        // load would otherwise drain this prefetch too: in-order vmcnt)
        const uint16_t* Ki = smem + buf * BUF;
        const uint16_t* Vs = Ki + KIMG;
        const uint16_t* Qs = Vs + SLAB;
        // (What the ISA does with this prefetch, rounds 2 - 6: the row loop below stores, does not load, and uses these registers in the wave's
        // last row, so the compiler's wait-count pass flushes vmcnt(0) in the loop's PREHEADER -- the workgroup waits for its own prefetch before
        // its first row, and it is the CU's other two workgroups that cover the latency.  Peeling the last row removes the flush and measured
        // slower, 9 843 against 9 310 cycles per item: LABNOTES round 6.)
        if (bw_next < Bw) issue_loads(bw_next);      // consumed by write_stage at the top of the wave's last row
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const bool fixed = bounded && !do_mask;                       // wave-uniform
        const float c0 = fixed ? -sc2 : 0.f;
        // accumulator start of the last key tile: padded keys (rows of S^T = registers) get -1e30, so P = 0 there
        f32x4 cpad;
#pragma unroll
        for (int r = 0; r < 4; ++r) cpad[r] = (16 * (LT - 1) + 4 * g + r < Lc) ? c0 : SWV2_NEG_BIG;

        bf16x8 kreg[KREG ? LT : 1], vreg[KREG ? LT / 2 : 1];
        bf16x4 vtail = {0, 0, 0, 0};
        if constexpr (KREG) {
#pragma unroll
            for (int t = 0; t < LT; ++t) kreg[t] = *(const bf16x8*)(Ki + (16 * t + fr) * 32 + 8 * g);
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                vreg[t / 2] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (LT & 1) vtail = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
#pragma unroll
            for (int t = 0; t < LT; ++t) asm volatile("" : "+v"(kreg[t]));
#pragma unroll
            for (int t = 0; t < LT / 2; ++t) asm volatile("" : "+v"(vreg[t]));
            asm volatile("" : "+v"(vtail));
        }

        STAMP(5, cpad[0]);
#pragma unroll 1
        for (int qt = wave; qt < LT; qt += WAVES) {                   // wave-uniform trip count
            const int q = 16 * qt + fr;
#ifndef SWV2_FWD3_NO_PRIO
            // Issue priority falls with the wave's progress through its item.  The three waves of a SIMD belong to the CU's three workgroups and
            // the arbiter serves the oldest first: spans of the item loop (s_memtime, -DSWV2_FWD3_SPAN, tools/probe_attn_fwd.py) mean 78 K cycles,
            // max 104 K without this, 80.5 K / 96 K with it -- the launch ends with the slowest workgroup.
            if (qt < WAVES) __builtin_amdgcn_s_setprio(2);
            else if (qt < 2 * WAVES) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            // the next item's slabs go to the other LDS buffer before the LAST row's stores are issued: its wait (in-order
            // vmcnt) then covers only loads issued rows ago and the earlier rows' stores
            if (qt + WAVES >= LT && bw_next < Bw) write_stage(buf ^ 1);
            const bf16x4 qraw = *(const bf16x4*)(Qs + (size_t)q * DP + 4 * g);

            // B operand of S^T = K Q^T over k = (part, d): (hi | lo) bf16 split of sigma' q^[4g .. 4g + 3] (exact to ~2^-17)
            bf16x8 qB;
            {
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = bf2f(qraw[j]) * sc2;
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    w[j] = f2bf2(x[2 * j], x[2 * j + 1]);
                    w[2 + j] = f2bf2(x[2 * j] - __uint_as_float(w[j] << 16), x[2 * j + 1] - __uint_as_float(w[j] & 0xffff0000u));
                }
                qB = __builtin_bit_cast(bf16x8, w);
            }
            STAMP(0, qB[0]);

            // S'^T tiles: rows = keys 16t + 4g + r, column = query fr; already scaled, and (fixed) already minus sigma'
            f32x4 acc[LT];
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                bf16x8 kA;
                if constexpr (KREG) kA = kreg[t];
                else kA = *(const bf16x8*)(Ki + (16 * t + fr) * 32 + 8 * g);
                const f32x4 c = (t == LT - 1) ? cpad : (f32x4){c0, c0, c0, c0};
                acc[t] = mfma32(kA, qB, c);
            }
            float mx = sc2;
            if (!fixed) {
                mx = SWV2_NEG_BIG;
                const bool qid = q >= mask_thr;
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (do_mask) acc[t][r] += (((16 * t + 4 * g + r) >= mask_thr) != qid) ? (-100.f * SWV2_LOG2E) : 0.f;
                        mx = fmaxf(mx, acc[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] -= mx;
            }
            STAMP(1, acc[LT - 1][0]);
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // registers that can only hold padded keys (compile-time window area) skip the exp
                    if (LFIX > 0 && 16 * t + 4 * 0 + r >= LFIX && t == LT - 1) acc[t][r] = 0.f;
                    else acc[t][r] = __builtin_amdgcn_exp2f(acc[t][r]);
                }
            STAMP(2, acc[LT - 1][1]);

            // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q] and the row sums (all-ones A operand), pairs of key tiles per K = 32
            f32x4 o = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 p0 = f2bf4(acc[t]), p1 = f2bf4(acc[t + 1]);
                const bf16x8 pb = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf16x8 vA;
                if constexpr (KREG) vA = vreg[t / 2];
                else {
                    const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                    const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                    vA = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                o = mfma32(vA, pb, o);
                rs = mfma32(ones8, pb, rs);
            }
            if (LT & 1) {
                const bf16x4 pb = f2bf4(acc[LT - 1]);
                bf16x4 vf;
                if constexpr (KREG) vf = vtail;
                else vf = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                // own accumulators for the K = 16 tail (see attn.hip: chaining it onto the K = 32 accumulator was wrong)
                const f32x4 to = mfma16(vf, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                const f32x4 ts = mfma16(ones4, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                o += to;
                rs += ts;
            }
            const float sum = rs[0];                                  // every row of the ones product holds the column sums
            STAMP(3, sum);
            const float inv = (q < L) ? __builtin_amdgcn_rcpf(sum) : 0.f;
            uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
            f32x4 v = o;
            v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
            {
                *(bf16x4*)(orow + 4 * g) = f2bf4(v);
                if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx + __log2f(sum) : 0.f;
            }
            STAMP(4, v[0]);
        }
        __syncthreads();
        STAMP(6, stage[0][0]);
    }
#ifdef SWV2_ATTN_STAMPS
    if (tid == 0 && Lp - L >= 14) {
        unsigned long long* dst = (unsigned long long*)(lse + ((size_t)bw_first * h + hd) * Lp + L);
#pragma unroll
        for (int k_ = 0; k_ < 7; ++k_) dst[k_] = st_acc[k_];
        (void)rt0; (void)ct0;
    }
#endif
#ifdef SWV2_FWD3_SPAN
    if (tid == 0) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 2048) { fwd3_span[2 * wg] = __builtin_amdgcn_s_memtime() - span_c0; fwd3_span[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - span_r0; }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// the same forward for 32-wide head slots (head_dim 17 .. 32; BASELINE configs[4]: 192 channels, 8 heads of 24): the K = 32
// operand is filled by the head itself, so the (hi | lo) split of sigma' q^ takes TWO MFMAs per key tile against the same K
// fragment (the matrix pipe has the room: ~40 MFMAs = 640 cycles per 16-query row against ~1500 cycles of vector work), the K
// image needs no duplication (one 16-byte read per lane from the slab as it lies in memory), and O^T has two 16-row tiles.
// The item's q | k | v slabs are one contiguous 3 x Lp x 32 block in memory and are staged as such.  NBUF = 2: the next item's
// slabs go to the other LDS buffer during the wave's last row (one barrier per item, 2 x 33 KB: two workgroups per CU);
// NBUF = 1: one buffer, rewritten between two barriers (three workgroups per CU).
// ------------------------------------------------------------------------------------------------
template <int LT, int LFIX, int WAVES, int OCC, int NBUF, bool KREG = false>
__global__ __launch_bounds__(64 * WAVES, OCC) void attn_fwd3w_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, uint16_t* __restrict__ oh, float* __restrict__ lse,
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int DP = 32, Lp = 16 * LT, SLAB = Lp * DP;
    constexpr int NT = 64 * WAVES;
    constexpr int CH = 3 * SLAB / 8;                  // 16-byte chunks of the q | k | v slabs
    constexpr int CPT = (CH + NT - 1) / NT;
    constexpr int BUF = 3 * SLAB;
    __shared__ __attribute__((aligned(16))) uint16_t smem[NBUF * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const bool last_chunk_ok = (wave * 64 + (CPT - 1) * NT) < CH;          // wave-uniform (CH is a multiple of 64)
    const int Lc = LFIX > 0 ? LFIX : L;

    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;
    const bool bounded = sc2 <= 40.f;

    u32x4 stage[CPT];
    unsigned soff[CPT];                               // 32-bit lane offsets against a wave-uniform base (see attn_fwd3_kernel)
#pragma unroll
    for (int j = 0; j < CPT; ++j) soff[j] = (unsigned)(min(tid + j * NT, CH - 1) * 8);
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) stage[j] = *(const u32x4*)(base + soff[j]);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (j < CPT - 1 || last_chunk_ok) *(u32x4*)(dst + (size_t)(tid + j * NT) * 8) = stage[j];
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue_loads(bw);
    write_stage(0);
    __syncthreads();

    const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = NBUF == 2 ? (it & 1) : 0;
        const int bw_next = bw + gridDim.x;
        const uint16_t* Qs = smem + buf * BUF;
        const uint16_t* Ks = Qs + SLAB;
        const uint16_t* Vs = Ks + SLAB;
        if (bw_next < Bw) issue_loads(bw_next);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const bool fixed = bounded && !do_mask;                       // wave-uniform
        const float c0 = fixed ? -sc2 : 0.f;
        f32x4 cpad;                                                   // padded keys start at -1e30: P = 0 there
#pragma unroll
        for (int r = 0; r < 4; ++r) cpad[r] = (16 * (LT - 1) + 4 * g + r < Lc) ? c0 : SWV2_NEG_BIG;
        auto vfrag = [&](int t, int dt) { return lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4); };
        // KREG: the item's K and V^T fragments in registers, read once per item and wave and pinned (see attn_fwd3_kernel)
        bf16x8 kreg[KREG ? LT : 1], vreg[KREG ? LT / 2 : 1][2];
        bf16x4 vtail[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (KREG) {
#pragma unroll
            for (int t = 0; t < LT; ++t) kreg[t] = *(const bf16x8*)(Ks + (16 * t + fr) * DP + 8 * g);
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vreg[t / 2][dt] = __builtin_shufflevector(vfrag(t, dt), vfrag(t + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
            if (LT & 1) { vtail[0] = vfrag(LT - 1, 0); vtail[1] = vfrag(LT - 1, 1); }
#pragma unroll
            for (int t = 0; t < LT; ++t) asm volatile("" : "+v"(kreg[t]));
#pragma unroll
            for (int t = 0; t < LT / 2; ++t) asm volatile("" : "+v"(vreg[t][0]), "+v"(vreg[t][1]));
            asm volatile("" : "+v"(vtail[0]), "+v"(vtail[1]));
        }

#pragma unroll 1
        for (int qt = wave; qt < LT; qt += WAVES) {                   // wave-uniform trip count
            const int q = 16 * qt + fr;
            // (no progress priority here: at two workgroups per CU it measured 34.1 / 34.6 against 34.0 / 33.9 ms per configs[3] step)
            if (NBUF == 2 && qt + WAVES >= LT && bw_next < Bw) write_stage(buf ^ 1);
            const bf16x8 qraw = *(const bf16x8*)(Qs + (size_t)q * DP + 8 * g);
            // B operands of S^T = K Q^T: hi and lo bf16 parts of sigma' q^[8g .. 8g + 7] (sum exact to ~2^-17)
            bf16x8 qhi, qlo;
            {
                uint32_t wh[4], wl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = bf2f(qraw[2 * j]) * sc2, x1 = bf2f(qraw[2 * j + 1]) * sc2;
                    wh[j] = f2bf2(x0, x1);
                    wl[j] = f2bf2(x0 - __uint_as_float(wh[j] << 16), x1 - __uint_as_float(wh[j] & 0xffff0000u));
                }
                qhi = __builtin_bit_cast(bf16x8, wh);
                qlo = __builtin_bit_cast(bf16x8, wl);
            }
            f32x4 acc[LT];
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                bf16x8 kA;
                if constexpr (KREG) kA = kreg[t];
                else kA = *(const bf16x8*)(Ks + (16 * t + fr) * DP + 8 * g);
                const f32x4 c = (t == LT - 1) ? cpad : (f32x4){c0, c0, c0, c0};
                acc[t] = mfma32(kA, qlo, c);
                acc[t] = mfma32(kA, qhi, acc[t]);
            }
            float mx = sc2;
            if (!fixed) {
                mx = SWV2_NEG_BIG;
                const bool qid = q >= mask_thr;
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (do_mask) acc[t][r] += (((16 * t + 4 * g + r) >= mask_thr) != qid) ? (-100.f * SWV2_LOG2E) : 0.f;
                        mx = fmaxf(mx, acc[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] -= mx;
            }
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (LFIX > 0 && 16 * t + r >= LFIX && t == LT - 1) acc[t][r] = 0.f;      // registers that only hold padded keys
                    else acc[t][r] = __builtin_amdgcn_exp2f(acc[t][r]);
                }
            // O^T[d][q] (two 16-row tiles of d) and the row sums, pairs of key tiles per K = 32
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0, rs = o0;
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 p0 = f2bf4(acc[t]), p1 = f2bf4(acc[t + 1]);
                const bf16x8 pb = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf16x8 vA0, vA1;
                if constexpr (KREG) { vA0 = vreg[t / 2][0]; vA1 = vreg[t / 2][1]; }
                else {
                    vA0 = __builtin_shufflevector(vfrag(t, 0), vfrag(t + 1, 0), 0, 1, 2, 3, 4, 5, 6, 7);
                    vA1 = __builtin_shufflevector(vfrag(t, 1), vfrag(t + 1, 1), 0, 1, 2, 3, 4, 5, 6, 7);
                }
                o0 = mfma32(vA0, pb, o0);
                o1 = mfma32(vA1, pb, o1);
                rs = mfma32(ones8, pb, rs);
            }
            if (LT & 1) {
                const bf16x4 pb = f2bf4(acc[LT - 1]);
                // own accumulators for the K = 16 tail (see attn.hip)
                o0 += mfma16(KREG ? vtail[0] : vfrag(LT - 1, 0), pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                o1 += mfma16(KREG ? vtail[1] : vfrag(LT - 1, 1), pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                rs += mfma16(ones4, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
            }
            const float sum = rs[0];                                  // every row of the ones product holds the column sums
            const float inv = (q < L) ? __builtin_amdgcn_rcpf(sum) : 0.f;
            uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
            o0[0] *= inv; o0[1] *= inv; o0[2] *= inv; o0[3] *= inv;
            o1[0] *= inv; o1[1] *= inv; o1[2] *= inv; o1[3] *= inv;
            *(bf16x4*)(orow + 4 * g) = f2bf4(o0);
            *(bf16x4*)(orow + 16 + 4 * g) = f2bf4(o1);
            if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx + __log2f(sum) : 0.f;
        }
        __syncthreads();
        if (NBUF == 1 && bw_next < Bw) {
            write_stage(0);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the same forward WITH the CPB bias table (round 5; the table reference swinv2_global.py:274-287 adds before the softmax, :307).
// The bias costs no vector instruction: the wave keeps the packed bf16 rows of its (head, q-tiles) in registers for the lifetime of the
// workgroup (swv2_attn_pack_bias's forward part: exactly the B-operand layout of a 16 x 16 x 16 MFMA, 2 registers per key tile), and
// every S'^T tile STARTS as  I . B_bias + c0  -- one more MFMA with an identity A operand (exact: 1.0 x bf16 into fp32) whose result
// is the accumulator of the K = 32 score product.  Padded keys carry -1e30 in the table.  The "fixed maximum" reference becomes
// sigma' + max(bias') of the head (the (max, min) part of the packed buffer), valid while 2 sigma' + (max - min) <= 80.
// 66 registers of bias per wave (3 q-tiles) on top of the 145 of the kernel without bias: two workgroups per CU instead of three.
// ------------------------------------------------------------------------------------------------
template <int LT, int LFIX, int WAVES, int OCC, bool KREG>
__global__ __launch_bounds__(64 * WAVES, OCC) void attn_fwd3b_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint32_t* __restrict__ bpack,
    const float* __restrict__ brange, uint16_t* __restrict__ oh, float* __restrict__ lse, int Bw, int h, int L, int nW, int nww, int nwh,
    int mask_thr) {
    constexpr int DP = 16, Lp = 16 * LT, SLAB = Lp * DP;
    constexpr int NT = 64 * WAVES;
    constexpr int CH = 2 * SLAB / 8;
    constexpr int CPT = (CH + NT - 1) / NT;
    constexpr int KIMG = 2 * SLAB, BUF = KIMG + 2 * SLAB;
    constexpr int QCH = SLAB / 8, QPT = (QCH + NT - 1) / NT;
    constexpr int NQ = (LT + WAVES - 1) / WAVES;               // q-tiles per wave (the last wave may own one less)
    static_assert(NQ <= 4, "the bias-set switch has four cases");
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const bool last_chunk_ok = (wave * 64 + (CPT - 1) * NT) < CH;          // wave-uniform

    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;
    const float bmax = brange[2 * hd], bmin = brange[2 * hd + 1];
    const bool bounded = 2.f * sc2 + (bmax - bmin) <= 80.f;

    // bias rows of this wave's q-tiles: bq[i][t] = keys 16t + 4g .. + 3 of query 16 (wave + i WAVES) + fr, packed bf16, log2 domain
    // STREAM: only the CURRENT q-tile's rows are held (22 registers instead of 66): they are re-requested from L2 for the next q-tile the
    // moment the identity products have consumed them and land while the rest of the tile runs -- three workgroups per CU
    constexpr bool STREAM = OCC >= 3;
    uint32_t bq[STREAM ? 1 : NQ][LT][2];
    auto load_rows = [&](int i, uint32_t (&dst)[LT][2]) {
        const int qt = min(wave + i * WAVES, LT - 1);
        const uint32_t* src = bpack + (((size_t)hd * LT + qt) * LT) * 128 + lane;
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            dst[t][0] = src[t * 128];
            dst[t][1] = src[t * 128 + 64];
        }
    };
    if constexpr (STREAM) load_rows(0, bq[0]);
    else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) load_rows(i, bq[i]);
    }
    // identity A operand of the 16 x 16 x 16 MFMA: A[i = fr][k = 4g + j]
    bf16x4 idA;
#pragma unroll
    for (int j = 0; j < 4; ++j) idA[j] = (fr == 4 * g + j) ? (short)0x3F80 : (short)0;

    u32x4 stage[CPT], stageq[QPT];
    unsigned soff[CPT], qoff[QPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) soff[j] = (unsigned)(SLAB + min(tid + j * NT, CH - 1) * 8);
#pragma unroll
    for (int j = 0; j < QPT; ++j) qoff[j] = (unsigned)(min(tid + j * NT, QCH - 1) * 8);
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) stage[j] = *(const u32x4*)(base + soff[j]);
#pragma unroll
        for (int j = 0; j < QPT; ++j) stageq[j] = *(const u32x4*)(base + qoff[j]);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (j < CPT - 1 || last_chunk_ok) {
                const int c = tid + j * NT;
                if (c < CH / 2) {
                    const u32x4 lo = {stage[j][0], stage[j][1], stage[j][0], stage[j][1]};
                    const u32x4 hi = {stage[j][2], stage[j][3], stage[j][2], stage[j][3]};
                    *(u32x4*)(dst + (size_t)c * 16) = lo;
                    *(u32x4*)(dst + (size_t)c * 16 + 8) = hi;
                } else {
                    *(u32x4*)(dst + KIMG + (size_t)(c - CH / 2) * 8) = stage[j];
                }
            }
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (tid + j * NT < QCH) *(u32x4*)(dst + KIMG + SLAB + (size_t)(tid + j * NT) * 8) = stageq[j];
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue_loads(bw);
    write_stage(0);
    __syncthreads();

    const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const int bw_next = bw + gridDim.x;
        const uint16_t* Ki = smem + buf * BUF;
        const uint16_t* Vs = Ki + KIMG;
        const uint16_t* Qs = Vs + SLAB;
        if (bw_next < Bw) issue_loads(bw_next);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const bool fixed = bounded && !do_mask;                       // wave-uniform
        const float c0 = fixed ? -(sc2 + bmax) : 0.f;
        const f32x4 cinit = {c0, c0, c0, c0};
        // KREG: the item's K fragments (A operands of S^T) in registers, read once per item and wave and pinned (see attn_fwd3_kernel)
        bf16x8 kreg[KREG ? LT : 1];
        if constexpr (KREG) {
#pragma unroll
            for (int t = 0; t < LT; ++t) kreg[t] = *(const bf16x8*)(Ki + (16 * t + fr) * 32 + 8 * g);
#pragma unroll
            for (int t = 0; t < LT; ++t) asm volatile("" : "+v"(kreg[t]));
        }

        // rolled loop (unrolled, the compiler interleaves the rows and spills: 256 registers + 94 in scratch); the wave's i-th bias
        // register set is picked by a scalar switch around the identity products only
#pragma unroll 1
        for (int i = 0; i < NQ; ++i) {
            const int qt = wave + i * WAVES;
            if (qt >= LT) break;                                      // wave-uniform
            const int q = 16 * qt + fr;
#ifndef SWV2_FWD3_NO_PRIO
            // (issue priority falls with the wave's progress through its item, see attn_fwd3_kernel: 62.4 -> 57.7 - 58.0 us in situ, same box)
            if (i == 0) __builtin_amdgcn_s_setprio(2);
            else if (i == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            if (qt + WAVES >= LT && bw_next < Bw) write_stage(buf ^ 1);
            const bf16x4 qraw = *(const bf16x4*)(Qs + (size_t)q * DP + 4 * g);
            bf16x8 qB;
            {
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = bf2f(qraw[j]) * sc2;
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    w[j] = f2bf2(x[2 * j], x[2 * j + 1]);
                    w[2 + j] = f2bf2(x[2 * j] - __uint_as_float(w[j] << 16), x[2 * j + 1] - __uint_as_float(w[j] & 0xffff0000u));
                }
                qB = __builtin_bit_cast(bf16x8, w);
            }
            // S'^T tiles: bias + c0 through the identity product, then the scaled scores on top
            f32x4 acc[LT];
#define SWV2_BIAS_TILES(I)                                                               \
    case I:                                                                              \
        if (I < NQ) {                                                                    \
            _Pragma("unroll") for (int t = 0; t < LT; ++t) {                             \
                const uint32_t bw2[2] = {bq[I < NQ ? I : 0][t][0], bq[I < NQ ? I : 0][t][1]}; \
                acc[t] = mfma16(idA, __builtin_bit_cast(bf16x4, bw2), cinit);            \
            }                                                                            \
        }                                                                                \
        break;
            if constexpr (STREAM) {
#pragma unroll
                for (int t = 0; t < LT; ++t) {
                    const uint32_t bw2[2] = {bq[0][t][0], bq[0][t][1]};
                    acc[t] = mfma16(idA, __builtin_bit_cast(bf16x4, bw2), cinit);
                }
                // the next q-tile of this wave (the first one of the next item behind the last): same registers, requested now
                const bool more = (i + 1 < NQ) && (wave + (i + 1) * WAVES < LT);
                load_rows(more ? i + 1 : 0, bq[0]);
            } else {
            switch (i) { SWV2_BIAS_TILES(0) SWV2_BIAS_TILES(1) SWV2_BIAS_TILES(2) SWV2_BIAS_TILES(3) default: break; }
            }
#undef SWV2_BIAS_TILES
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                bf16x8 kA;
                if constexpr (KREG) kA = kreg[t];
                else kA = *(const bf16x8*)(Ki + (16 * t + fr) * 32 + 8 * g);
                acc[t] = mfma32(kA, qB, acc[t]);
            }
            float mx = sc2 + bmax;
            if (!fixed) {
                mx = SWV2_NEG_BIG;
                const bool qid = q >= mask_thr;
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (do_mask) acc[t][r] += (((16 * t + 4 * g + r) >= mask_thr) != qid) ? (-100.f * SWV2_LOG2E) : 0.f;
                        mx = fmaxf(mx, acc[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] -= mx;
            }
#pragma unroll
            for (int t = 0; t < LT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (LFIX > 0 && 16 * t + 4 * 0 + r >= LFIX && t == LT - 1) acc[t][r] = 0.f;
                    else acc[t][r] = __builtin_amdgcn_exp2f(acc[t][r]);
                }
            f32x4 o = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 p0 = f2bf4(acc[t]), p1 = f2bf4(acc[t + 1]);
                const bf16x8 pb = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const bf16x8 vA = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma32(vA, pb, o);
                rs = mfma32(ones8, pb, rs);
            }
            if (LT & 1) {
                const bf16x4 pb = f2bf4(acc[LT - 1]);
                const bf16x4 vf = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const f32x4 to = mfma16(vf, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                const f32x4 ts = mfma16(ones4, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                o += to;
                rs += ts;
            }
            const float sum = rs[0];
            const float inv = (q < L) ? __builtin_amdgcn_rcpf(sum) : 0.f;
            uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
            f32x4 v = o;
            v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
            *(bf16x4*)(orow + 4 * g) = f2bf4(v);
            if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx + __log2f(sum) : 0.f;
        }
        __syncthreads();
    }
}

template <int LT, int LFIX, int WAVES, int OCC, bool KREG>
int launch_fwd3b(const swv2_attn_args* a, const uint32_t* bpack, const float* brange, hipStream_t st) {
    int nchunk = (OCC * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    hipLaunchKernelGGL((attn_fwd3b_kernel<LT, LFIX, WAVES, OCC, KREG>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale, bpack, brange,
                       (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}

template <int LT, int LFIX, int WAVES, int OCC, int NBUF, bool KREG = false>
int launch_fwd3w(const swv2_attn_args* a, hipStream_t st) {
    int nchunk = (OCC * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    hipLaunchKernelGGL((attn_fwd3w_kernel<LT, LFIX, WAVES, OCC, NBUF, KREG>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                       (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}

template <int LT, int LFIX, int WAVES, int OCC, bool KREG>
int launch_fwd3(const swv2_attn_args* a, hipStream_t st) {
    // OCC workgroups per CU on 256 CUs, every workgroup loops over windows
    int nchunk = (OCC * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    hipLaunchKernelGGL((attn_fwd3_kernel<LT, LFIX, WAVES, OCC, KREG>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                       (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr, a->dbg);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}

}  // namespace

#ifdef SWV2_FWD3_SPAN
extern "C" int swv2_debug_fwd3_span(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fwd3_span), sizeof(unsigned long long) * 2048 * 2) == hipSuccess ? 0 : -3;
}
#endif

// called by swv2_attn_fwd (attn.hip); returns 1 when this kernel does not cover the shape (the caller then runs the
// first-generation kernel).  Window areas: only the LAST key tile's accumulators start at -1e30 for padded keys, so every other key tile
// must hold real keys only -- L >= 16 (LT - 1) = 160 (at 65 .. 159 tokens the zero K rows of the padded keys in tiles 4 .. 9
// entered the row sum with weight exp2(-sigma')); smaller areas run the first-generation kernel.  With a CPB table: the packed form (swv2_attn_pack_bias)
// is required -- its forward part is the register image, its (max, min) part bounds the fixed-maximum softmax.
int swv2_attn2_fwd(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (Lp != 176 || a->L < 160 || (DP != 16 && DP != 32)) return 1;
    if (a->bias) {
        static const int fwd3b = getenv("SWV2_ATTN_FWD3B") ? atoi(getenv("SWV2_ATTN_FWD3B")) : 1;
        if (!fwd3b || !a->bias_pack || DP != 16 || a->heads <= 0) return 1;
        const uint32_t* bpack = (const uint32_t*)a->bias_pack;
        const float* brange = (const float*)((const char*)a->bias_pack + swv2_attn_bias_range_offset(a->heads, a->L));
        static const int kreg = getenv("SWV2_ATTN_FWD3B_KREG") ? atoi(getenv("SWV2_ATTN_FWD3B_KREG")) : 0;
        if (kreg) return a->L == 162 ? launch_fwd3b<11, 162, 4, 2, true>(a, bpack, brange, st) : launch_fwd3b<11, 0, 4, 2, true>(a, bpack, brange, st);
        static const int stream = getenv("SWV2_ATTN_FWD3B_STREAM") ? atoi(getenv("SWV2_ATTN_FWD3B_STREAM")) : 1;
        if (stream && a->L == 162) return launch_fwd3b<11, 162, 4, 3, false>(a, bpack, brange, st);
        return a->L == 162 ? launch_fwd3b<11, 162, 4, 2, false>(a, bpack, brange, st) : launch_fwd3b<11, 0, 4, 2, false>(a, bpack, brange, st);
    }
    if (DP == 32) {
        // measured at B = 2 (800 windows x 8 heads, 24-wide heads): 4 waves x 2 workgroups per CU, two LDS buffers: 90 us with the K / V^T
        // fragments read per row, 80 us with them in registers (227 VGPRs); one buffer at 3 workgroups per CU (spills): 99; 6 waves x 2
        // workgroups (SIMDs loaded 4 / 4 / 2 / 2): 112; first generation: 203
        static const int kreg = getenv("SWV2_ATTN_FWD3W_KREG") ? atoi(getenv("SWV2_ATTN_FWD3W_KREG")) : 1;
        if (!kreg) return a->L == 162 ? launch_fwd3w<11, 162, 4, 2, 2, false>(a, st) : launch_fwd3w<11, 0, 4, 2, 2, false>(a, st);
        return a->L == 162 ? launch_fwd3w<11, 162, 4, 2, 2, true>(a, st) : launch_fwd3w<11, 0, 4, 2, 2, true>(a, st);
    }
    if (a->L == 162) return launch_fwd3<11, 162, 4, 3, false>(a, st);          // measured best: 49 us at B = 2 (first generation: 72)
    return launch_fwd3<11, 0, 4, 3, false>(a, st);
}
