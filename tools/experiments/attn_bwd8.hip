// Cosine window attention, backward at the benchmark head geometry (176-row window layout, 16-wide heads), third form (round 5):
// 8 waves per workgroup, 256 registers per wave, waves 0 - 2 own TWO key tiles, waves 3 - 7 one.
//
// Same semantics, operands and two-phase structure as attn_bwd_kernel<11, 1, *, *, 1, true> in attn.hip (reference
// networks/swinv2_global.py:298-318 and its autograd; statistics / padded-key flag / shift mask inside the K = 32 MFMA operands; dS
// through an LDS image, dQ in a second, barrier-separated phase).  Why another wave -> tile map:
//   * the 11-wave kernel is bound by the LDS pipe in phase 1 (tools/ubench_lds.hip: a 16-byte wave read every 5.9 ns per CU, an 8-byte
//     or transposed one every 3.9 ns): every wave re-reads the Q-side operands of all 11 query tiles -- two 16-byte operand reads and
//     two transposed fragment reads per (wave, q tile), 2.9 us per window.  A wave that owns two key tiles reads them once for both:
//     (3 x 27.4 + 5 x 23.6) ns x 11 = 2.2 us per window.  The load per SIMD stays 3-3-3-2 key tiles (wave w sits on SIMD w mod 4:
//     SIMD 0 - 2 hold a two-tile and a one-tile wave, SIMD 3 two one-tile waves) -- round 2's 6-wave form (two tiles per wave
//     everywhere) put 4-4-2-2 on the SIMDs and lost (132 vs 105 us).
//   * with a CPB table the wave also accumulates the table's gradient: 44 registers per key tile.  At 168 registers (11 waves: three
//     per SIMD) the software-pipelined q-tile loop does not fit beside them (134 + 44 + the table operand; the spilled build ran 303
//     us), so that kernel runs a rolled loop at 166 - 171 us against 105 without a table.  At two waves per SIMD there are 256
//     registers: the pipelined loop and 88 registers of d bias rows fit.
//
// RESULT (round 5, same box, bench.py in situ at local batch 2; parity-green on the 39 attention / block tests of tests/test_gpu_parity.py,
// d(qkv) bit-identical to the 11-wave kernel): without a table 111.1 vs 105.7 us per launch (step 9.374 vs 9.332 ms) -- the LDS-pipe
// arithmetic above is right about the reads, but a SIMD now runs a two-tile wave beside a one-tile wave: when the short wave has finished
// its 11 steps the long one is alone on the SIMD and cannot hide its own LDS / MFMA latencies (three one-tile waves overlap for the whole
// phase).  With a table 289.5 vs 166 us: the kernel needs ~293 registers (d bias rows 88, operands 32, pipeline / step temporaries ~80,
// prefetch 21, addresses ~30, allocator slack), the 37 - 44 that go to scratch are the NEXT window's prefetch registers -- stored
// right behind their loads, i.e. waited for on the spot -- so every window pays an exposed HBM round trip; a single register set for
// the two-tile waves did not remove the spills (152 bytes).  Retired to tools/experiments; to revive: move to csrc/, add to
// _lib.SOURCES and call swv2_attn_bwd8 from launch_bwd<11, 1, *> in attn.hip (the hook is in this file's last function).
#include <type_traits>

#include "../../swin_v2_weather_amd/csrc/attn_common.h"

int swv2_attn_bwd8(const swv2_attn_args* a, int Lp, int DP, const uint16_t* bimg, float* dbws, int nchunk, void* stream);

namespace {

constexpr int B8_LT = 11, B8_DP = 16, B8_LP = 16 * B8_LT, B8_SLAB = B8_LP * B8_DP, B8_WAVES = 8, B8_NT = 64 * B8_WAVES;
constexpr int B8_DSP = B8_LP + 4;            // row pitch (elements) of the [key][q] bf16 images
constexpr int B8_QP = 40;                    // 80-byte rows of the q / dO slabs: 16 channels + 8 statistics slots + pad (conflict-free 16-byte reads)

template <int LFIX, bool HAS_BIAS>
__global__ __launch_bounds__(B8_NT, 1) void attn_bwd8_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale,
    const uint16_t* __restrict__ bimg,     // swv2_attn_pack_bias backward part ([h][Lp][Lp + 4] bf16) -- required with HAS_BIAS
    const uint16_t* __restrict__ oh, const uint16_t* __restrict__ doh, const float* __restrict__ lse,
    const float* __restrict__ rnorm, uint16_t* __restrict__ dqkvh, float* __restrict__ dlogit,
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr,
    float* __restrict__ dbws) {            // [gridDim.x][h][L][L] per-workgroup d bias tables (HAS_BIAS)
    constexpr int LT = B8_LT, DP = B8_DP, Lp = B8_LP, SLAB = B8_SLAB, NT = B8_NT, DSP = B8_DSP, QP = B8_QP;
    constexpr int CH = SLAB / 8, CPR = 2;                      // 16-byte chunks per slab / per row
    constexpr int BROWS = HAS_BIAS ? ((LFIX > 0 ? LFIX : Lp) + 3) / 4 * 4 : 0;       // bias image rows (keys): the real keys, rounded up to 4
    constexpr int OFF_Q = 0, OFF_DO = OFF_Q + Lp * QP * 2, OFF_K = OFF_DO + Lp * QP * 2, OFF_V = OFF_K + SLAB * 2, OFF_DS = OFF_V + SLAB * 2,
                  OFF_BIAS = OFF_DS + Lp * DSP * 2, OFF_RED = OFF_BIAS + (HAS_BIAS ? BROWS * DSP * 2 : 0), LDS_BYTES = OFF_RED + 64;
    static_assert(OFF_DS % 16 == 0 && OFF_BIAS % 16 == 0 && OFF_RED % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS layout");
    static_assert(CH <= NT, "one staging chunk per thread");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    uint16_t* const Qa = (uint16_t*)(lds + OFF_Q);
    uint16_t* const Da = (uint16_t*)(lds + OFF_DO);
    uint16_t* const Ks = (uint16_t*)(lds + OFF_K);
    uint16_t* const Vs = (uint16_t*)(lds + OFF_V);
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
    [[maybe_unused]] uint16_t* const biasS = (uint16_t*)(lds + OFF_BIAS);
    float* const red = (float*)(lds + OFF_RED);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const int nt = tw < 3 ? 2 : 1;                   // key tiles (phase 1) / query tiles (phase 2) of this wave
    const int t0 = tw < 3 ? 2 * tw : tw + 3;         // the first of them
    const int Lc = LFIX > 0 ? LFIX : L;

    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E;
    const float inv_sc2 = 1.f / sc2;

    // bias image of the head -> LDS (16-byte copies); d bias rows of the wave's key tiles in registers
    f32x4 dbr[HAS_BIAS ? 2 : 1][HAS_BIAS ? LT : 1];
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int qt = 0; qt < LT; ++qt) dbr[i][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint4* src = (const uint4*)(bimg + (size_t)hd * Lp * DSP);
        for (int i = tid; i < BROWS * DSP / 8; i += NT) ((uint4*)biasS)[i] = src[i];
    }
    float dsig = 0.f;

    // ---- staging registers: chunk c = tid of the q, k, v, dO, o slabs (threads CH .. NT - 1 stage nothing)
    uint4 sq = {}, sk = {}, sv = {}, sdo = {}, so = {};
    float slse_row = 0.f;
    const int c = tid, crow = min(c / CPR, Lp - 1);
    auto issue = [&](int bw) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB, oslab = ((size_t)bw * h + hd) * SLAB;
        const size_t cc = (size_t)min(c, CH - 1) * 8;                       // unconditional (clamped) loads: see attn.hip
        sq = *(const uint4*)(qkvh + slab0 + cc);
        sk = *(const uint4*)(qkvh + slab0 + SLAB + cc);
        sv = *(const uint4*)(qkvh + slab0 + 2 * SLAB + cc);
        sdo = *(const uint4*)(doh + oslab + cc);
        so = *(const uint4*)(oh + oslab + cc);
        slse_row = lse[((size_t)bw * h + hd) * Lp + crow];
    };
    auto commit = [&]() {
        if (c < CH) {
            *(uint4*)(Qa + (c / CPR) * QP + (c % CPR) * 8) = sq;
            *(uint4*)(Da + (c / CPR) * QP + (c % CPR) * 8) = sdo;
            *(uint4*)(Ks + c * 8) = sk;
            *(uint4*)(Vs + c * 8) = sv;
        }
        // delta partial over this chunk's 8 channels, summed over the row's two chunks (adjacent lanes)
        float dl = 0.f;
        {
            const uint32_t a[4] = {sdo.x, sdo.y, sdo.z, sdo.w}, b[4] = {so.x, so.y, so.z, so.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), dl);
                dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), dl);
            }
        }
        dl += __shfl_xor(dl, 1);
        if (c < CH) {
            // slots 16 .. 23 of the row (the even chunk's thread): lse / (sigma log2 e) in three bf16 parts, a constant 1 (padded-key flag),
            // the query's mask-region flags; delta in three parts for the dO slab.  Slots 24 .. 31 (the odd chunk's thread): zeros.
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
            *(uint4*)(Qa + row * QP + 16 + (c & 1) * 8) = aq;
            *(uint4*)(Da + row * QP + 16 + (c & 1) * 8) = ad;
        }
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue(bw);
    commit();
    __syncthreads();

    for (; bw < Bw; bw += gridDim.x) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        const int bw_next = bw + gridDim.x;
        if (bw_next < Bw) issue(bw_next);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;

        // ================= phase 1: wave = key tile(s) =================
        bf16x8 kf8[2], vf8[2];
        f32x4 dk[2], dv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = 16 * min(t0 + i, LT - 1) + fr;               // (a one-tile wave's second slot repeats a valid tile; unused)
            const uint32_t m1 = 0xbf80u;                                   // -1
            const uint32_t padk = (key < Lc) ? 0u : (uint32_t)f2bf(-1.0e30f);
            const bool kreg = key >= mask_thr;
            const uint32_t mk0 = f2bf(kreg ? 0.f : cmask), mk1 = f2bf(kreg ? cmask : 0.f);
            const uint4 augk = make_uint4(m1 | (m1 << 16), m1 | (padk << 16), mk0 | (mk1 << 16), 0);
            const uint4 augv = make_uint4(m1 | (m1 << 16), m1, 0, 0);
            const uint4 z = make_uint4(0, 0, 0, 0);
            const uint4 rk = *(const uint4*)(Ks + key * DP + (g & 1) * 8), rv = *(const uint4*)(Vs + key * DP + (g & 1) * 8);
            kf8[i] = __builtin_bit_cast(bf16x8, g < 2 ? rk : (g == 2 ? augk : z));
            vf8[i] = __builtin_bit_cast(bf16x8, g < 2 ? rv : (g == 2 ? augv : z));
            dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // two-stage software pipeline over the q tiles (two register sets used alternately): stage A of step qt + 1 (operand reads, S' and
        // dP' MFMAs, bias rows) is issued before stage B of step qt (p = exp2(fma(S', sigma log2 e, b)), dS = p dP', dV / dK MFMAs)
        auto phase1 = [&](auto ntl_c) {
            constexpr int NTL = decltype(ntl_c)::value;
            struct St { f32x4 s[NTL], dp[NTL]; bf16x4 tq, td; };
            auto stageA = [&](const int qt, St& o) {
                const bf16x8 qa = *(const bf16x8*)(Qa + (16 * qt + fr) * QP + 8 * g);
                const bf16x8 da = *(const bf16x8*)(Da + (16 * qt + fr) * QP + 8 * g);
                o.td = lds_tr_read(Da + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
                o.tq = lds_tr_read(Qa + (16 * qt + 4 * g + (fr >> 2)) * QP + (fr & 3) * 4);
#pragma unroll
                for (int i = 0; i < NTL; ++i) {
                    o.s[i] = mfma32(qa, kf8[i], (f32x4){0.f, 0.f, 0.f, 0.f});
                    o.dp[i] = mfma32(da, vf8[i], (f32x4){0.f, 0.f, 0.f, 0.f});
                }
            };
            auto stageB = [&](const int qt, const St& in) {
                f32x4 dsv[NTL];
#pragma unroll
                for (int i = 0; i < NTL; ++i) {
                    const int key = 16 * (t0 + i) + fr;
                    [[maybe_unused]] bf16x4 b4;
                    if constexpr (HAS_BIAS) b4 = *(const bf16x4*)(biasS + min(key, BROWS - 1) * DSP + 16 * qt + 4 * g);
                    f32x4 p, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (HAS_BIAS) p[r] = __builtin_amdgcn_exp2f(fmaf(in.s[i][r], sc2, bf2f(b4[r])));
                        else p[r] = __builtin_amdgcn_exp2f(in.s[i][r] * sc2);
                        ds[r] = p[r] * in.dp[i][r];
                    }
                    const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
                    *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;
                    dv[i] = mfma16(in.td, pb, dv[i]);
                    dk[i] = mfma16(in.tq, dsb, dk[i]);
                    dsv[i] = ds;
                }
                if constexpr (HAS_BIAS) {
                    // the d bias rows stay statically indexed registers through a (scalar, wave-uniform) switch on the tile index
#define SWV2_CASE(I) case I: dbr[0][I] += dsv[0]; if constexpr (NTL == 2) dbr[1][I] += dsv[NTL - 1]; break;
                    switch (qt) { SWV2_CASE(0) SWV2_CASE(1) SWV2_CASE(2) SWV2_CASE(3) SWV2_CASE(4) SWV2_CASE(5)
                                  SWV2_CASE(6) SWV2_CASE(7) SWV2_CASE(8) SWV2_CASE(9) SWV2_CASE(10) }
#undef SWV2_CASE
                }
            };
            if constexpr (HAS_BIAS && NTL == 2) {
                // two key tiles AND their d bias rows: one register set (the two tiles of a step overlap each other's MFMA / vector work);
                // the second set of the two-stage pipeline is what does not fit 256 registers (44 dwords of scratch, reloaded ~15 times
                // per window outside this loop: 289 us per launch)
                St sa;
#pragma unroll 1
                for (int qt = 0; qt < LT; ++qt) {
                    stageA(qt, sa);
                    stageB(qt, sa);
                }
            } else {
                St sa, sb;
                stageA(0, sa);
#pragma unroll 1
                for (int qt = 0; qt + 1 < LT; qt += 2) {
                    stageA(qt + 1, sb);
                    stageB(qt, sa);
                    if (qt + 2 < LT) stageA(qt + 2, sa);
                    stageB(qt + 1, sb);
                }
                if (LT & 1) stageB(LT - 1, sa);
            }
        };
        if (nt == 2) phase1(std::integral_constant<int, 2>{});
        else phase1(std::integral_constant<int, 1>{});

        // ---- dK (through the L2-normalisation) and dV of this wave's key tile(s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= nt) continue;                                        // wave-uniform
            const int key = 16 * (t0 + i) + fr;
            const float rk = rnorm[(((size_t)bw * h + hd) * 2 + 1) * Lp + key];
            // the accumulators hold rows d = 4g + r of column key fr: channels 4g .. 4g + 3 of this lane's key (still in LDS)
            const bf16x4 kf4 = *(const bf16x4*)(Ks + key * DP + 4 * g);
            float kv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) kv[r] = bf2f(kf4[r]);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dk[i][r], kv[r], dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (g == 0) dsig += dot;
            const float rks = rk * sigma;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rks * (dk[i][r] - kv[r] * dot);
            *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 4 * g) = f2bf4(v);
            *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 4 * g) = f2bf4(dv[i]);
        }
        __syncthreads();

        // ================= phase 2: wave = query tile(s) =================
        {
            f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            auto frag = [&](int t, bf16x4& kt_, bf16x4 (&ds_)[2]) {
                const int row = 16 * t + 4 * g + (fr >> 2);
                kt_ = lds_tr_read(Ks + row * DP + (fr & 3) * 4);                                  // rows d, col key
#pragma unroll
                for (int i = 0; i < 2; ++i) ds_[i] = lds_tr_read(dSb + row * DSP + 16 * min(t0 + i, LT - 1) + (fr & 3) * 4);   // B[k = key][n = q]
            };
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                bf16x4 k0, k1, d0[2], d1[2];
                frag(t, k0, d0);
                frag(t + 1, k1, d1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    dq[i] = mfma32(__builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(d0[i], d1[i], 0, 1, 2, 3, 4, 5, 6, 7), dq[i]);
            }
            {
                bf16x4 k0, d0[2];
                frag(LT - 1, k0, d0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // own accumulator: a K = 16 MFMA chained directly onto the K = 32 accumulator gave wrong sums (attn.hip)
                    const f32x4 tail = mfma16(k0, d0[i], (f32x4){0.f, 0.f, 0.f, 0.f});
                    dq[i] += tail;
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i >= nt) continue;
                const int q = 16 * (t0 + i) + fr;
                const float rq = rnorm[(((size_t)bw * h + hd) * 2 + 0) * Lp + q] * sigma;
                const bf16x4 qn = *(const bf16x4*)(Qa + q * QP + 4 * g);
                float dot = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dq[i][r], bf2f(qn[r]), dot);
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rq * (dq[i][r] - bf2f(qn[r]) * dot);
                *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 4 * g) = f2bf4(v);
            }
        }
        __syncthreads();
        if (bw_next < Bw) commit();
        __syncthreads();
    }

    // ---- flush the per-workgroup reductions: one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[tw] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < B8_WAVES; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= nt) continue;
            const int key = 16 * (t0 + i) + fr;
            if (key < L) {
#pragma unroll
                for (int qt = 0; qt < LT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = 16 * qt + 4 * g + r;
                        if (q < L) dbws[(((size_t)blockIdx.x * h + hd) * L + q) * L + key] = dbr[i][qt][r];
                    }
            }
        }
    }
}

}  // namespace

// called by swv2_attn_bwd (attn.hip) before its own dispatch: 0 / negative = handled (ok / error), 1 = shape / options not covered.
// Covers the 176-row layout with 16-wide heads; with a CPB table it needs the packed table (its LDS image) and a workspace for the
// workgroups' d bias tables (summed by the caller: dbias_reduce in attn.hip, or swv2_cpb_bwd_multi with dbias_partials).
int swv2_attn_bwd8(const swv2_attn_args* a, int Lp, int DP, const uint16_t* bimg, float* dbws, int nchunk, void* stream) {
    static const int on = getenv("SWV2_ATTN_BWD8") ? atoi(getenv("SWV2_ATTN_BWD8")) : 1;
    if (!on || Lp != B8_LP || DP != B8_DP || (a->dbg & (SWV2_ATTN_FIRST_GEN | SWV2_ATTN_PLAIN_STATS))) return 1;
    if (a->bias && (!bimg || !dbws)) return 1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(nchunk, a->heads), block(B8_NT);
    const int nW = a->nwh * a->nww;
#define SWV2_B8(LF, HB)                                                                                                          \
    hipLaunchKernelGGL((attn_bwd8_kernel<LF, HB>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale, bimg, (const uint16_t*)a->oh, \
                       (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, a->Bw, a->heads, a->L, nW, a->nww, \
                       a->nwh, a->mask_thr, dbws)
    if (a->bias) {
        if (a->L == 162) SWV2_B8(162, true);
        else return 1;                                 // (the run-time-L table image of the 11-wave kernel keeps all 176 rows)
    } else {
        if (a->L == 162) SWV2_B8(162, false);
        else SWV2_B8(0, false);
    }
#undef SWV2_B8
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}
