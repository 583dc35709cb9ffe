// NOT PART OF THE SHIPPED LIBRARY.  Kept for the record: the bias-capable 4-wave forward (attn_fwd2_kernel) and the 4-wave
// backward (attn_bwd2_kernel) of the second attention generation.  Both were parity-green and both lost their A/B against the
// first-generation kernels of csrc/attn.hip (forward with CPB bias 115 vs 90 us; backward 134 - 148 vs 111 us; LABNOTES.md),
// so round 3 took them out of libswv2.so.  To build them again: put this file back under swin_v2_weather_amd/csrc/, add it to
// _lib.SOURCES and call swv2_attn2x_fwd / swv2_attn2x_bwd from swv2_attn_fwd / swv2_attn_bwd.
// Cosine window attention core, second generation (gfx950 / CDNA4): small out-of-phase workgroups.
//
// Same semantics and data layout as attn.hip (reference networks/swinv2_global.py:298-318 and :170-198):
//   S = sigma_h * qn kn^T + Bias_h + Mask ; P = softmax(S) ; O = P v
//
// Why a second generation.  The first kernels ran ONE workgroup of LT (= 11) waves per CU, wave = one 16-row tile, with
// workgroup barriers per window: all waves of a CU sit in the same phase (MFMA, then softmax on the vector ALU, then
// MFMA), so the matrix pipe and the vector pipe never overlap, and the K = 16 MFMA (16 cycles, the same as the K = 32
// form -- measured, tools/ubench.hip) is used everywhere.  Measured instruction costs per SIMD (tools/ubench.hip):
// v_exp_f32 8.3 cycles per wave-instruction at any occupancy, but it co-issues with ordinary VALU work of OTHER waves
// (exp + fma alternating: 9.3 cycles per pair at 4 waves per SIMD, 13.3 at one); v_fma 2.0 at >= 3 waves per SIMD, 2.6
// at 2, 5.1 alone; packed f32 / f16 arithmetic is no faster per element.  So the softmax needs ~35-45 SIMD cycles per
// 16x16 tile when waves in different phases share a SIMD, against ~75 + the serialised MFMA time in lockstep.
//
// Here a workgroup is 4 waves (one per SIMD) and 3 workgroups share a CU (168 VGPRs, <= 48 KB LDS each), each working
// on its own (window, head) item, so MFMA phases of one workgroup overlap the softmax of another; a wave owns several
// 16-row tiles of its item (q tiles w, w + 4, w + 8 in the forward; key tiles in the backward), which amortises the
// operand-fragment reads; every product whose contraction runs over tokens uses the K = 32 MFMA on a pair of tiles.
#include <type_traits>

#include "attn_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// In-kernel phase timing (diagnostic builds only, -DSWV2_ATTN_STAMPS, tools/probe_attn_stamps.py): wave 0 of every workgroup
// accumulates s_memtime deltas per phase and leaves them in the padded tail of its first item's lse row.  `dep` orders the
// stamp behind the value that ends the phase.
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
// forward: wave = q tiles {w, w + WAVES, ...}; swapped product S^T = K Q^T (lane = query column), O^T = V^T P^T with
// pairs of key tiles as one K = 32 operand.  K / V slabs double-buffered in LDS (register-staged prefetch of the next
// item, one barrier per item).
// ------------------------------------------------------------------------------------------------
template <int LT, int DK, bool HAS_BIAS, int LFIX, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 3) void attn_fwd2_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale,
    const uint32_t* __restrict__ bpack,   // swv2_attn_pack_bias forward part (HAS_BIAS)
    uint16_t* __restrict__ oh, float* __restrict__ lse, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    using C = AttnCfg<LT, DK>;
    constexpr int Lp = C::Lp, DP = C::DP, SLAB = C::SLAB;
    constexpr int NT = 64 * WAVES;
    constexpr int CH = 2 * SLAB / 8;                  // 16-byte chunks of the K | V slabs (a multiple of 64)
    constexpr int CPT = (CH + NT - 1) / NT;           // chunks per thread; the last one is valid for whole waves only
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * 2 * SLAB];   // [buf][K|V][Lp][DP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const bool last_chunk_ok = (wave * 64 + (CPT - 1) * NT) < CH;          // wave-uniform

    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;

    // native vector type + unconditional (clamped) loads: an array of HIP uint4 written under a condition lives in scratch
    // (the first build stored every prefetched chunk to scratch right behind an s_waitcnt: no prefetch at all)
    u32x4 stage[CPT];
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB + SLAB;   // K slab, V slab follows
#pragma unroll
        for (int j = 0; j < CPT; ++j) stage[j] = *(const u32x4*)(base + (size_t)min(tid + j * NT, CH - 1) * 8);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * 2 * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (j < CPT - 1 || last_chunk_ok) *(u32x4*)(dst + (size_t)(tid + j * NT) * 8) = stage[j];
    };
    // Q fragment (B operand: B[k = d 4g+j][n = query fr]) straight from global, 512 B contiguous per wave, always fetched one
    // q tile ahead (rolled q-tile loop: a register array indexed by the loop counter would live in scratch)
    bf16x4 qf[DK], qn[DK];
    auto load_q = [&](int bw, int qt) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB + (size_t)(16 * qt + fr) * DP + 4 * g;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) qn[kk] = *(const bf16x4*)(base + 16 * kk);
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    const int bw_first = bw;
    STAMP_DECL
    issue_loads(bw);
    load_q(bw, wave);
    write_stage(0);
    __syncthreads();
    STAMP_START();

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const int bw_next = bw + gridDim.x;
        if (bw_next < Bw) issue_loads(bw_next);

        const uint16_t* Ks = smem + buf * 2 * SLAB;
        const uint16_t* Vs = Ks + SLAB;
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);

#pragma unroll 1
        for (int qt = wave; qt < LT; qt += WAVES) {       // wave-uniform trip count
            const int q = 16 * qt + fr;
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) qf[kk] = qn[kk];
            STAMP(0, qf[0]);
            {   // next q tile of this item, or this wave's first q tile of the next item
                const bool wrap = qt + WAVES >= LT;
                load_q(wrap ? (bw_next < Bw ? bw_next : bw) : bw, wrap ? wave : qt + WAVES);
            }

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

            float mx, sum = 0.f;
            if (!HAS_BIAS && !do_mask) {
                // sigma > 0 commutes with the maximum: row maximum over the raw cosines, the scale folded into the fma
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
                STAMP(1, mx);
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(acc[t][r], sc2, -mx));
                        acc[t][r] = p;
                        sum += p;
                    }
            } else {
                uint32_t biasp[LT][2];
                if (HAS_BIAS) {
                    const uint32_t* src = bpack + (((size_t)hd * LT + qt) * LT) * 128 + lane;
#pragma unroll
                    for (int t = 0; t < LT; ++t) {
                        biasp[t][0] = src[t * 128];
                        biasp[t][1] = src[t * 128 + 64];
                    }
                }
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
            STAMP(2, sum);

            // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]; two key tiles = one K = 32 operand (same k order on both sides)
            f32x4 o[DK];
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 p0 = f2bf4(acc[t]), p1 = f2bf4(acc[t + 1]);
                const bf16x8 pb = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4);
                    const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4);
                    o[dt] = mfma32(__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), pb, o[dt]);
                }
            }
            if (LT & 1) {
                const bf16x4 pb = f2bf4(acc[LT - 1]);
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    const bf16x4 vf = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4);
                    // own accumulator for the K = 16 tail (see attn.hip: chaining it onto the K = 32 accumulator was wrong)
                    const f32x4 tail = mfma16(vf, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                    o[dt] += tail;
                }
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
            STAMP(3, o[0][0]);
        }
        if (bw_next < Bw) write_stage(buf ^ 1);
        STAMP(4, stage[0][0]);
        __syncthreads();
        STAMP(5, stage[0][0]);
    }
#ifdef SWV2_ATTN_STAMPS
    if (tid == 0 && Lp - L >= 14) {
        unsigned long long* dst = (unsigned long long*)(lse + ((size_t)bw_first * h + hd) * Lp + L);
#pragma unroll
        for (int k = 0; k < 7; ++k) dst[k] = st_acc[k];
    }
#endif
}

template <int LT, int DK, int LFIX>
int launch_fwd2(const swv2_attn_args* a, hipStream_t st) {
    constexpr int WAVES = 4;
    // 3 workgroups per CU, 256 CUs: heads x chunks ~ 768 workgroups, every workgroup loops over windows
    int nchunk = (3 * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    const int nW = a->nwh * a->nww;
    if (a->bias)
        hipLaunchKernelGGL((attn_fwd2_kernel<LT, DK, true, LFIX, WAVES>), grid, block, 0, st, (const uint16_t*)a->qkvh,
                           a->logit_scale, (const uint32_t*)a->bias_pack, (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, nW,
                           a->nww, a->nwh, a->mask_thr);
    else
        hipLaunchKernelGGL((attn_fwd2_kernel<LT, DK, false, LFIX, WAVES>), grid, block, 0, st, (const uint16_t*)a->qkvh,
                           a->logit_scale, (const uint32_t*)nullptr, (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, nW, a->nww,
                           a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}


// ------------------------------------------------------------------------------------------------
// backward: wave = key tiles {w, w + WAVES, ...} of the item, loop over pairs of q tiles.
//   S = Q K^T, dP = dO V^T (rows q, column = key: the accumulators are directly the B operands of the products that
//   contract over q), P = exp2(S' - LSE'), dS = P (dP - delta);
//   dV^T += dO^T P, dK^T += Q^T dS   : K = 32 MFMAs over the q-tile pair (A: transposed LDS reads of the staged dO / Q);
//   dQ^T += K^T dS^T                 : contracts over keys, so the dS tile goes through a wave-private LDS scratch tile
//                                      (8-byte write, transposed read; same wave, no barrier) and the wave's key tiles
//                                      pair up into a K = 32 operand; the K^T fragments are made once per item by an MFMA
//                                      against the identity (K as the A operand -> K^T in the accumulator layout).
//   The four waves' partial dQ tiles of a step meet in LDS: every wave stores its fp32 tiles to its own slot (one 16-byte
//   store per tile), one workgroup barrier, then all 256 threads sum the four slots, apply the normalisation backward and
//   store dQ (slots double buffered by step parity: one barrier per step).  LDS float atomics are NOT an option: a
//   ds_add_f32 wave-instruction occupied the CU's LDS for ~250 cycles (measured: 467 us per launch against 121 us).
// K / V fragments of the wave's own key tiles come straight from global memory (prefetched one item ahead); only Q, dO,
// LSE, delta and 1/|q| are staged in LDS (double buffered): ~50 KB per workgroup, three workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int LT, int DK, int LFIX, int WAVES, int OCC>
__global__ __launch_bounds__(64 * WAVES, OCC) void attn_bwd2_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint16_t* __restrict__ oh,
    const uint16_t* __restrict__ doh, const float* __restrict__ lse, const float* __restrict__ rnorm,
    uint16_t* __restrict__ dqkvh, float* __restrict__ dlogit, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    using C = AttnCfg<LT, DK>;
    constexpr int Lp = C::Lp, DP = C::DP, SLAB = C::SLAB;
    constexpr int NT = 64 * WAVES;
    static_assert(WAVES == 4, "the dQ summation below maps 256 threads onto 32 rows x 8 column groups");
    constexpr int KPW = (LT + WAVES - 1) / WAVES;            // key tiles per wave
    static_assert(KPW == 3, "the dQ pairing below is written for three key tiles per wave");
    constexpr int CH = SLAB / 8;                             // 16-byte chunks per slab
    constexpr int CPT = (CH + NT - 1) / NT;                  // chunks per thread
    constexpr int CPR = 2 * DK;                              // chunks per row
    // one staging buffer: Q, dO (bf16), LSE, delta, sigma/|q|, sigma/|k| (fp32) and, for head_dim <= 16, the image of
    // sigma' q^ split into bf16 (hi | lo) parts, [q][g][hi d 4g.. | lo d 4g..] (64 B per q): the A operand of the K = 32 product
    // S' = (hi | lo) . (k^ | k^) -- scaled logits straight from the matrix pipe, accumulator started at -LSE
    constexpr bool FOLD = (DK == 1);
    constexpr int OFF_QHL = 4 * SLAB + 16 * Lp;
    constexpr int BUFB = OFF_QHL + (FOLD ? 4 * SLAB : 0);
    constexpr int PARTW = 2 * 16 * DP * 4;                   // bytes of one wave's partial dQ tiles of a step ([2][16 q][DP])
    constexpr int OFF_PART = 2 * BUFB, OFF_SCR = OFF_PART + 2 * WAVES * PARTW, SCRW = 2 * 2 * 512,
                  OFF_RED = OFF_SCR + WAVES * SCRW, LDS_BYTES = OFF_RED + 16 * ((WAVES + 3) / 4);
    static_assert(BUFB % 16 == 0 && OFF_PART % 16 == 0 && OFF_SCR % 16 == 0, "16-byte aligned sub-arrays");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    using frag_t = std::conditional_t<DK == 1, bf16x4, bf16x8>;   // one row piece of a K / V / Q / dO tile (d = 4g.. or 8g..)
    static_assert(DK == 1 || DK == 2, "head_dim <= 32");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const int Lc = LFIX > 0 ? LFIX : L;

    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E;

    uint16_t* const scr = (uint16_t*)(lds + OFF_SCR + wave * SCRW);
    float* const red = (float*)(lds + OFF_RED);

    // selector operands of the transposing MFMAs (B[k][n] = 1 iff k-th d of the lane's fragment == 16 dt + n)
    frag_t sel[DK];
#pragma unroll
    for (int dt = 0; dt < DK; ++dt)
#pragma unroll
        for (int j = 0; j < 4 * DK; ++j) sel[dt][j] = ((DK == 1 ? 4 * g + j : 8 * g + j) == 16 * dt + fr) ? (short)0x3F80 : (short)0;

    // ---- staging registers: chunk c = tid + j*NT of the q, dO, o slabs (clamped, unconditional loads; native vectors)
    u32x4 sq[CPT], sdo[CPT], so[CPT];
    float slse = 0.f, srq = 0.f, srk = 0.f;
    frag_t kf[KPW], vf[KPW], kfn[KPW], vfn[KPW];
    auto issue = [&](int bw) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB, oslab = ((size_t)bw * h + hd) * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = min(tid + j * NT, CH - 1);
            sq[j] = *(const u32x4*)(qkvh + slab0 + (size_t)c * 8);
            sdo[j] = *(const u32x4*)(doh + oslab + (size_t)c * 8);
            so[j] = *(const u32x4*)(oh + oslab + (size_t)c * 8);
        }
        slse = lse[((size_t)bw * h + hd) * Lp + min(tid, Lp - 1)];
        srq = rnorm[(((size_t)bw * h + hd) * 2 + 0) * Lp + min(tid, Lp - 1)];
        srk = rnorm[(((size_t)bw * h + hd) * 2 + 1) * Lp + min(tid, Lp - 1)];
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const int key = 16 * min(wave + WAVES * j, LT - 1) + fr;
            const uint16_t* pk = qkvh + slab0 + SLAB + (size_t)key * DP + (DK == 1 ? 4 : 8) * g;
            kfn[j] = *(const frag_t*)pk;
            vfn[j] = *(const frag_t*)(pk + SLAB);
        }
    };
    auto commit = [&](int buf) {
        uint16_t* Qs = (uint16_t*)(lds + buf * BUFB);
        uint16_t* dOs = Qs + SLAB;
        float* LSEs = (float*)(lds + buf * BUFB + 4 * SLAB);
        float* DLs = LSEs + Lp;
        float* RQs = DLs + Lp;
        float* RKs = RQs + Lp;
        uint16_t* QHL = (uint16_t*)(lds + buf * BUFB + OFF_QHL);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            const bool ok = c < CH;
            if (ok) {
                *(u32x4*)(Qs + c * 8) = sq[j];
                *(u32x4*)(dOs + c * 8) = sdo[j];
                if constexpr (FOLD) {
                    // chunk c = row c / 2, d = 8 (c & 1) .. + 7 -> groups g = 2 (c & 1), 2 (c & 1) + 1 of that row
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        uint32_t w[4];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const uint32_t pr = sq[j][2 * gg + e];
                            const float x0 = __uint_as_float(pr << 16) * sc2, x1 = __uint_as_float(pr & 0xffff0000u) * sc2;
                            w[e] = f2bf2(x0, x1);
                            w[2 + e] = f2bf2(x0 - __uint_as_float(w[e] << 16), x1 - __uint_as_float(w[e] & 0xffff0000u));
                        }
                        *(u32x4*)(QHL + (size_t)c * 16 + gg * 8) = (u32x4){w[0], w[1], w[2], w[3]};
                    }
                }
            }
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dl = fmaf(__uint_as_float(sdo[j][e] << 16), __uint_as_float(so[j][e] << 16), dl);
                dl = fmaf(__uint_as_float(sdo[j][e] & 0xffff0000u), __uint_as_float(so[j][e] & 0xffff0000u), dl);
            }
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) dl += __shfl_xor(dl, o);
            if (ok && (c % CPR) == 0) DLs[c / CPR] = dl;
        }
        if (tid < Lp) {
            LSEs[tid] = (tid < L) ? slse : 1.0e30f;
            RQs[tid] = srq * sigma;
            RKs[tid] = srk * sigma;
        }
    };

    // K^T fragments (A operand of dQ^T = K^T dS^T: rows d, k = key), pairs of key tiles as one K = 32 operand
    bf16x8 kT01[DK];
    bf16x4 kT2[DK];
    auto make_kT = [&]() {
        bf16x4 t[KPW][DK];
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if constexpr (DK == 1) z = mfma16(kf[j], sel[dt], z);
                else z = mfma32(kf[j], sel[dt], z);
                t[j][dt] = f2bf4(z);                         // exact: the products are the bf16 values themselves
            }
#pragma unroll
        for (int dt = 0; dt < DK; ++dt) {
            kT01[dt] = __builtin_shufflevector(t[0][dt], t[1][dt], 0, 1, 2, 3, 4, 5, 6, 7);
            kT2[dt] = t[2][dt];
        }
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue(bw);
    commit(0);
#pragma unroll
    for (int j = 0; j < KPW; ++j) { kf[j] = kfn[j]; vf[j] = vfn[j]; }
    make_kT();
    float dsig = 0.f;
    __syncthreads();

    // per-lane accumulator start of the S tiles: -1e30 on padded keys (and on a wave's surplus tile), so P = 0 there
    // without any per-element select
    float sinit[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int kt = wave + WAVES * j;
        sinit[j] = (kt < LT && 16 * kt + fr < Lc) ? 0.f : SWV2_NEG_BIG;
    }
    // only a wave's LAST key tile can hold padded keys (or be surplus): the others start their S' accumulators at -LSE alone
    // dQ summation: thread -> (row of the step's 32 q rows, group of DP/8 columns)
    const int srow = tid >> 3, scol = (tid & 7) * (DP / 8);

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        const int bw_next = bw + gridDim.x;

        const uint16_t* Qs = (const uint16_t*)(lds + buf * BUFB);
        const uint16_t* dOs = Qs + SLAB;
        const float* LSEs = (const float*)(lds + buf * BUFB + 4 * SLAB);
        const float* DLs = LSEs + Lp;
        const float* RQs = DLs + Lp;
        const float* RKs = RQs + Lp;
        const uint16_t* QHL = (const uint16_t*)(lds + buf * BUFB + OFF_QHL);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);

        f32x4 dk[KPW][DK], dv[KPW][DK];
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                dk[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dv[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }

        // one step = NQ (2 or 1) q tiles against the wave's key tiles
        auto step = [&](const int i0, auto nq_c, auto masked_c) {
            constexpr int NQ = decltype(nq_c)::value;
            constexpr bool MASKED = decltype(masked_c)::value;
            frag_t qa[NQ], da[NQ];
            bf16x8 qhl[NQ];
            bf16x4 tq[NQ][DK], td[NQ][DK];
            f32x4 l4[NQ], d4[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int i = i0 + u;
                if constexpr (FOLD) qhl[u] = *(const bf16x8*)(QHL + (16 * i + fr) * 32 + 8 * g);
                else qa[u] = *(const frag_t*)(Qs + (16 * i + fr) * DP + (DK == 1 ? 4 : 8) * g);
                da[u] = *(const frag_t*)(dOs + (16 * i + fr) * DP + (DK == 1 ? 4 : 8) * g);
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    const int off = (16 * i + 4 * g + (fr >> 2)) * DP + 16 * dt + (fr & 3) * 4;
                    td[u][dt] = lds_tr_read(dOs + off);
                    tq[u][dt] = lds_tr_read(Qs + off);
                }
                l4[u] = *(const f32x4*)(LSEs + 16 * i + 4 * g);
                d4[u] = *(const f32x4*)(DLs + 16 * i + 4 * g);
                if constexpr (FOLD) { l4[u] = -l4[u]; d4[u] = -d4[u]; }      // accumulator start values
            }
            const int toff = (4 * g + (fr >> 2)) * 16 + (fr & 3) * 4;
            f32x4 dq[NQ][DK];
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                // (a wave's surplus tile runs too -- its start value makes P = dS = 0 -- so the loop body stays branch-free)
                const int key = 16 * (wave + WAVES * j) + fr;
                const bool kid = key >= mask_thr;
                bf16x4 pb[NQ], dsb[NQ];
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    f32x4 s, dp, p, ds;
                    if constexpr (FOLD) {
                        // x = sigma' cos - LSE and dP - delta come out of the MFMAs (start values -LSE, -delta)
                        s = l4[u];
                        if (j == KPW - 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[r] += sinit[j];
                        }
                        s = mfma32(qhl[u], __builtin_shufflevector(kf[j], kf[j], 0, 1, 2, 3, 0, 1, 2, 3), s);
                        dp = mfma16(da[u], vf[j], d4[u]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = s[r];
                            if (MASKED) x += (((16 * (i0 + u) + 4 * g + r) >= mask_thr) != kid) ? (-100.f * SWV2_LOG2E) : 0.f;
                            const float pr = __builtin_amdgcn_exp2f(x);
                            p[r] = pr;
                            ds[r] = pr * dp[r];               // dS; d(cos) = sigma dS is applied once at the end
                        }
                    } else {
                        s = (f32x4){sinit[j], sinit[j], sinit[j], sinit[j]};
                        dp = (f32x4){0.f, 0.f, 0.f, 0.f};
                        s = mfma32(qa[u], kf[j], s);
                        dp = mfma32(da[u], vf[j], dp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = fmaf(s[r], sc2, -l4[u][r]);
                            if (MASKED) x += (((16 * (i0 + u) + 4 * g + r) >= mask_thr) != kid) ? (-100.f * SWV2_LOG2E) : 0.f;
                            const float pr = __builtin_amdgcn_exp2f(x);
                            p[r] = pr;
                            ds[r] = pr * (dp[r] - d4[u][r]);
                        }
                    }
                    pb[u] = f2bf4(p);
                    dsb[u] = f2bf4(ds);
                    // [key][q] tile for the dQ product; the third key tile reuses the first one's slot (LDS operations of one
                    // wave execute in order, and the first pair's transposed reads are issued before this write)
                    *(bf16x4*)(scr + ((j & 1) * 2 + u) * 256 + fr * 16 + 4 * g) = dsb[u];
                }
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    if constexpr (NQ == 2) {
                        dv[j][dt] = mfma32(__builtin_shufflevector(td[0][dt], td[1][dt], 0, 1, 2, 3, 4, 5, 6, 7),
                                           __builtin_shufflevector(pb[0], pb[1], 0, 1, 2, 3, 4, 5, 6, 7), dv[j][dt]);
                        dk[j][dt] = mfma32(__builtin_shufflevector(tq[0][dt], tq[1][dt], 0, 1, 2, 3, 4, 5, 6, 7),
                                           __builtin_shufflevector(dsb[0], dsb[1], 0, 1, 2, 3, 4, 5, 6, 7), dk[j][dt]);
                    } else {
                        // own accumulators for the K = 16 tail (chaining K = 16 onto a K = 32 accumulator was wrong, attn.hip)
                        const f32x4 tv = mfma16(td[0][dt], pb[0], (f32x4){0.f, 0.f, 0.f, 0.f});
                        const f32x4 tk = mfma16(tq[0][dt], dsb[0], (f32x4){0.f, 0.f, 0.f, 0.f});
                        dv[j][dt] += tv;
                        dk[j][dt] += tk;
                    }
                }
                // partial dQ^T of the step's q tiles: key tiles 0 and 1 as one K = 32 operand, tile 2 on its own
                if (j == 1) {
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        const bf16x4 b0 = lds_tr_read(scr + (0 * 2 + u) * 256 + toff);
                        const bf16x4 b1 = lds_tr_read(scr + (1 * 2 + u) * 256 + toff);
                        const bf16x8 b01 = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int dt = 0; dt < DK; ++dt) dq[u][dt] = mfma32(kT01[dt], b01, (f32x4){0.f, 0.f, 0.f, 0.f});
                    }
                }
            }
            // -> this wave's slot ([u][q][d] fp32)
            const int sbuf = (i0 >> 1) & 1;
            float* const part = (float*)(lds + OFF_PART + (sbuf * WAVES + wave) * PARTW);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const bf16x4 b2 = lds_tr_read(scr + (0 * 2 + u) * 256 + toff);
#pragma unroll
                for (int dt = 0; dt < DK; ++dt) {
                    const f32x4 t2 = mfma16(kT2[dt], b2, (f32x4){0.f, 0.f, 0.f, 0.f});
                    const f32x4 dqs = dq[u][dt] + t2;
                    *(f32x4*)(part + (u * 16 + fr) * DP + 16 * dt + 4 * g) = dqs;    // rows d = 16dt + 4g + r of column q = fr
                }
            }
            __syncthreads();
            // all threads: sum the four slots, normalisation backward (dot over the row's DP columns = 8 adjacent lanes), store
            if (NQ == 2 || tid < 128) {
                const float* p0 = (const float*)(lds + OFF_PART + sbuf * WAVES * PARTW) + srow * DP + scol;
                const int q = 16 * i0 + srow;
                float sq_[DP / 8], qh[DP / 8];
#pragma unroll
                for (int e = 0; e < DP / 8; ++e) sq_[e] = 0.f;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv)
#pragma unroll
                    for (int e = 0; e < DP / 8; ++e) sq_[e] += p0[wv * (PARTW / 4) + e];
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < DP / 8; ++e) {
                    qh[e] = bf2f(Qs[q * DP + scol + e]);
                    dot = fmaf(sq_[e], qh[e], dot);
                }
                dot += __shfl_xor(dot, 1);
                dot += __shfl_xor(dot, 2);
                dot += __shfl_xor(dot, 4);
                const float rq = RQs[q];
                uint16_t* dst = dqkvh + slab0 + (size_t)q * DP + scol;
                if constexpr (DP == 16) {
                    *(uint32_t*)dst = f2bf2(rq * (sq_[0] - qh[0] * dot), rq * (sq_[1] - qh[1] * dot));
                } else {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = rq * (sq_[e] - qh[e] * dot);
                    *(bf16x4*)dst = f2bf4(v);
                }
            }
        };
        auto run = [&](auto masked_c) {
#pragma unroll 1
            for (int i0 = 0; i0 + 1 < LT; i0 += 2) step(i0, std::integral_constant<int, 2>{}, masked_c);
            // the next item's loads are issued here, not at the top: their staging registers would otherwise be live
            // across the pair loop (spills); the single-tile step and the epilogue below cover the latency, the other
            // workgroups of the CU the rest
            if (bw_next < Bw) issue(bw_next);
            if (LT & 1) step(LT - 1, std::integral_constant<int, 1>{}, masked_c);
        };
        if (do_mask) run(std::true_type{});
        else run(std::false_type{});

        // the next item's slabs go to the other LDS buffer BEFORE this item's dK / dV stores are issued: the wait for the
        // prefetch (in-order vmcnt) then does not include them
        if (bw_next < Bw) commit(buf ^ 1);
        // ---- dK (through the L2-normalisation backward) and dV of this wave's key tiles
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            if (wave + WAVES * j >= LT) continue;             // wave-uniform
            const int key = 16 * (wave + WAVES * j) + fr;
            const float rks = RKs[key];
            // k^ in the accumulator layout (rows d = 16 dt + 4g + r, column key): the lane's K fragment for DK = 1; for the
            // 8-wide fragment (d = 8g + j) the matching values are re-read from global (L2 hit)
            float kh[DK][4];
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                if constexpr (DK == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) kh[dt][r] = bf2f(kf[j][r]);
                } else {
                    const bf16x4 k4 = *(const bf16x4*)(qkvh + slab0 + SLAB + (size_t)key * DP + 16 * dt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) kh[dt][r] = bf2f(k4[r]);
                }
            }
            float dot = 0.f;
#pragma unroll
            for (int dt = 0; dt < DK; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dk[j][dt][r], kh[dt][r], dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            // d logit_scale = sigma sum_{q,k} dS cos = sigma sum_k (sum_q dS q^) . k^ : the dot product above
            if (g == 0) dsig += dot;
#pragma unroll
            for (int dt = 0; dt < DK; ++dt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rks * (dk[j][dt][r] - kh[dt][r] * dot);
                *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 16 * dt + 4 * g) = f2bf4(v);
                *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 16 * dt + 4 * g) = f2bf4(dv[j][dt]);
            }
        }
        if (bw_next < Bw) {
#pragma unroll
            for (int j = 0; j < KPW; ++j) { kf[j] = kfn[j]; vf[j] = vfn[j]; }
            make_kT();
        }
        __syncthreads();
    }

    // ---- one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[wave] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
}

template <int LT, int DK, int LFIX, int OCC>
int launch_bwd2(const swv2_attn_args* a, hipStream_t st) {
    constexpr int WAVES = 4;
    int nchunk = (OCC * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    const int nW = a->nwh * a->nww;
    hipLaunchKernelGGL((attn_bwd2_kernel<LT, DK, LFIX, WAVES, OCC>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                       (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale,
                       a->Bw, a->heads, a->L, nW, a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}

}  // namespace

// second-generation dispatch, called by swv2_attn_fwd / swv2_attn_bwd (attn.hip); returns 1 when no second-generation
// kernel covers the shape (the caller then runs the first-generation kernel)
int swv2_attn2x_fwd(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->bias && !a->bias_pack) return 1;
    if (!a->bias && Lp == 176 && DP == 16 && a->L == 162 && !(a->dbg & 64)) {
        if (a->dbg & 128) return launch_fwd3<11, 162, 4, 2, true>(a, st);      // probe: fragments pinned in registers, 2 workgroups per CU
        return launch_fwd3<11, 162, 4, 3, false>(a, st);                       // measured best: 51 us at B = 2 (first generation: 72)
    }
    if (!a->bias && Lp == 176 && DP == 16 && !(a->dbg & 64)) return launch_fwd3<11, 0, 4, 3, false>(a, st);
    // fwd2 (4-wave workgroups, rolled q-tile loop) measured SLOWER than the first generation on its remaining cases (CPB
    // bias at the benchmark shape: 115 vs 90 us; head dim 32): selectable for the parity tests with dbg bit 6 only
    if (!(a->dbg & 64)) return 1;
    if (Lp == 176 && DP == 16 && a->L == 162) return launch_fwd2<11, 1, 162>(a, st);
    if (Lp == 176 && DP == 16) return launch_fwd2<11, 1, 0>(a, st);
    if (Lp == 176 && DP == 32) return launch_fwd2<11, 2, 0>(a, st);
    return 1;
}

int swv2_attn2x_bwd(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->bias) return 1;                         // the bias-gradient rows need the one-key-tile-per-wave layout (attn.hip)
    // Measured at the benchmark shape (B = 2): 135-148 us against 121-127 us for the first-generation two-phase kernel -- the
    // per-step LDS round trips (dS transpose, partial-dQ slots + barrier) are latency-bound at two waves per SIMD.  Kept
    // selectable (dbg bit 6) and parity-tested; not the default.
    if (!(a->dbg & 64)) return 1;
    if (Lp == 176 && DP == 16 && a->L == 162) return (a->dbg & 32) ? launch_bwd2<11, 1, 162, 3>(a, st) : launch_bwd2<11, 1, 162, 2>(a, st);
    if (Lp == 176 && DP == 16) return launch_bwd2<11, 1, 0, 2>(a, st);
    if (Lp == 176 && DP == 32) return launch_bwd2<11, 2, 0, 1>(a, st);
    return 1;
}
