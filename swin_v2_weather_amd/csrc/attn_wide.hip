// Window-attention backward for wide heads (65 .. 96 channels in the 96-column head layout -- or the 128-column one -- : the
// reference's own swin_73var is 768 / 8 = 96), no CPB bias.  Same algorithm, data layout and phase structure as attn_bwd_kernel (attn.hip): wave = key tile in
// phase 1 (S = Q K^T, dP = dO V^T, P = exp2(S' - LSE'), dS = P (dP - delta), dV^T += dO^T P, dK^T += Q^T dS, bf16 dS image
// [key][q] in LDS), wave = query tile in phase 2 (dQ^T = sum_t K_t^T dS_t^T), the L2-normalisation backward in both epilogues.
// What differs is where the operands live.  At 128 columns attn_bwd_kernel cannot keep Q and dO in LDS beside K, V and the dS
// image (4 x 45 KB + 63 KB), reads their fragments from global memory -- the transposed ones as four 2-byte loads -- and keeps
// the next window's five slabs in 80 prefetch registers: 218 spilled registers, 1 780 us per launch at 73 x 720 x 1440 (14 % of
// the embed-768 step).  Here
//   * only the REAL channels are staged: Q and dO as [Lp][16 DKR + 16] bf16 tiles (pitch 112 at 96 channels = 224 bytes: the 8-byte
//     transposed reads of 8 consecutive rows and the 16-byte row reads of a ds_read_b128 lane group both fall on distinct banks);
//   * K and V of a wave's key tile are its MFMA B operands for the whole phase: loaded from global memory straight into
//     registers (16-byte loads), never staged; the K slab that phase 2 reads transposed is loaded over the Q tile between the
//     phases (the normalisation backward of dQ re-reads its q^ rows from global memory / L2);
//   * no cross-window register prefetch: a workgroup's loads are issued at the start of the window and consumed at once.
// LDS: 2 x 38.5 KB + 62 KB + 1.4 KB = 141 KB, one workgroup of 11 waves per CU, no spills.
#include <cstdlib>
#include <type_traits>

#include "attn_common.h"

namespace {

template <int LT, int DKR, int LFIX, int DP>
__global__ __launch_bounds__(64 * LT) void attn_bwd_wide_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint16_t* __restrict__ oh,
    const uint16_t* __restrict__ doh, const float* __restrict__ lse, const float* __restrict__ rnorm,
    uint16_t* __restrict__ dqkvh, float* __restrict__ dlogit, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    static_assert(DKR % 2 == 0 && DKR <= 6, "pairs of 16-channel tiles (K = 32 MFMA), at most 96 channels (LDS)");
    constexpr int Lp = 16 * LT, SLAB = Lp * DP, NT = 64 * LT;      // DP = columns of the head layout in memory (96 or 128)
    static_assert(DP >= 16 * DKR && DP % 16 == 0, "layout width");
    constexpr int QP = 16 * DKR + 16;                        // row pitch (elements) of the staged tiles
    constexpr int DSP = Lp + 4;                              // row pitch of the [key][q] dS image
    constexpr int KS = DKR / 2;                              // K = 32 steps over the channels
    constexpr int OFF_Q = 0, OFF_DO = OFF_Q + Lp * QP * 2, OFF_LSE = OFF_DO + Lp * QP * 2, OFF_DL = OFF_LSE + Lp * 4,
                  OFF_DS = OFF_DL + Lp * 4, OFF_RED = OFF_DS + Lp * DSP * 2, LDS_BYTES = OFF_RED + ((LT * 4 + 15) / 16) * 16;
    static_assert(OFF_DS % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS layout");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    uint16_t* const Qs = (uint16_t*)(lds + OFF_Q);           // phase 1: q^ rows; phase 2: k^ rows
    uint16_t* const dOs = (uint16_t*)(lds + OFF_DO);
    float* const LSEs = (float*)(lds + OFF_LSE);
    float* const DLs = (float*)(lds + OFF_DL);
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
    float* const red = (float*)(lds + OFF_RED);

    const int tid = threadIdx.x, lane = tid & 63, tw = tid >> 6;      // wave tw owns key tile tw (phase 1) / query tile tw (phase 2)
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E;
    const int Lc = LFIX > 0 ? LFIX : L;
    float dsig = 0.f;
    // staging map: 16 threads per row (12 of them carry a 16-byte chunk at 96 channels), NT / 16 rows per pass
    constexpr int RPP = NT / 16, PASSES = (Lp + RPP - 1) / RPP;
    const int srow = tid >> 4, scc = tid & 15;
    const bool sact = scc < 2 * DKR;

    for (int bw = blockIdx.x; bw < Bw; bw += gridDim.x) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB, oslab = ((size_t)bw * h + hd) * SLAB;
        // ---- stage q^, dO (-> LDS) and delta = rowsum(dO O); k^, v of this wave's key tile -> registers
        const int key = 16 * tw + fr;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            kf[kk] = *(const bf16x8*)(qkvh + slab0 + SLAB + (size_t)key * DP + 32 * kk + 8 * g);
            vf[kk] = *(const bf16x8*)(qkvh + slab0 + 2 * SLAB + (size_t)key * DP + 32 * kk + 8 * g);
        }
        {
            uint4 sq[PASSES], sd[PASSES], so[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = min(srow + RPP * p, Lp - 1), cc = sact ? scc : 0;      // unconditional (clamped) loads
                sq[p] = *(const uint4*)(qkvh + slab0 + (size_t)row * DP + cc * 8);
                sd[p] = *(const uint4*)(doh + oslab + (size_t)row * DP + cc * 8);
                so[p] = *(const uint4*)(oh + oslab + (size_t)row * DP + cc * 8);
            }
            const float slse = lse[((size_t)bw * h + hd) * Lp + min(tid, Lp - 1)];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = srow + RPP * p;
                float dl = 0.f;
                if (sact) {
                    const uint32_t a[4] = {sd[p].x, sd[p].y, sd[p].z, sd[p].w}, b[4] = {so[p].x, so[p].y, so[p].z, so[p].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(b[e] << 16), dl);
                        dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(b[e] & 0xffff0000u), dl);
                    }
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) dl += __shfl_xor(dl, o);
                if (row < Lp) {
                    if (sact) {
                        *(uint4*)(Qs + row * QP + scc * 8) = sq[p];
                        *(uint4*)(dOs + row * QP + scc * 8) = sd[p];
                    }
                    if (scc == 0) DLs[row] = dl;
                }
            }
            if (tid < Lp) LSEs[tid] = (tid < L) ? slse : 1.0e30f;        // padded query rows: P = 0
        }
        __syncthreads();

        // ================= phase 1: wave = key tile =================
        f32x4 dk[DKR], dv[DKR];
#pragma unroll
        for (int dt = 0; dt < DKR; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const bool kid = key >= mask_thr, key_ok = key < Lc;
        auto step = [&](const int qt, auto masked_c, auto pad_c) {
            constexpr bool MASKED = decltype(masked_c)::value, PADT = decltype(pad_c)::value;
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8 qa = *(const bf16x8*)(Qs + (16 * qt + fr) * QP + 32 * kk + 8 * g);
                const bf16x8 da = *(const bf16x8*)(dOs + (16 * qt + fr) * QP + 32 * kk + 8 * g);
                s = mfma32(qa, kf[kk], s);            // rows q = 16 qt + 4 g + r, column = key fr
                dp = mfma32(da, vf[kk], dp);
            }
            const f32x4 l4 = *(const f32x4*)(LSEs + 16 * qt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(DLs + 16 * qt + 4 * g);
            f32x4 p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * g + r;
                float x = fmaf(s[r], sc2, -l4[r]);
                if (PADT) x = key_ok ? x : SWV2_NEG_BIG;
                if (MASKED) x += ((q >= mask_thr) != kid) ? (-100.f * SWV2_LOG2E) : 0.f;
                const float pr = __builtin_amdgcn_exp2f(x);
                p[r] = pr;
                ds[r] = pr * (dp[r] - d4[r]);          // dS; d(cos) = sigma * dS is applied once per tile at the end
            }
            const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
            *(bf16x4*)(dSb + key * DSP + 16 * qt + 4 * g) = dsb;       // image [key][q] for phase 2
            // dV^T += dO^T P ; dK^T += Q^T dS     (A operands: transposed reads of the staged dO / Q tiles)
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) {
                const bf16x4 td = lds_tr_read(dOs + (16 * qt + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                const bf16x4 tq = lds_tr_read(Qs + (16 * qt + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                dv[dt] = mfma16(td, pb, dv[dt]);
                dk[dt] = mfma16(tq, dsb, dk[dt]);
            }
        };
        auto run = [&](auto masked_c, auto pad_c) {
#pragma unroll 1
            for (int qt = 0; qt < LT; ++qt) step(qt, masked_c, pad_c);
        };
        const bool pad_wave = 16 * tw + 16 > Lc;                  // padded keys only in the last tile
        if (do_mask) run(std::true_type{}, std::true_type{});
        else if (pad_wave) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});

        // ---- dK (through the L2-normalisation) and dV of this wave's key tile; the k^ values in the accumulators' layout from L2
        {
            const float rk = rnorm[(((size_t)bw * h + hd) * 2 + 1) * Lp + key];
            bf16x4 kn[DKR];
            float dot = 0.f;
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) {
                kn[dt] = *(const bf16x4*)(qkvh + slab0 + SLAB + (size_t)key * DP + 16 * dt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dk[dt][r], bf2f(kn[dt][r]), dot);
            }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (g == 0) dsig += dot;                              // d logit_scale = sigma sum_k dot_k (see attn_bwd_kernel)
            const float rks = rk * sigma;
            const bf16x4 z4 = {0, 0, 0, 0};
#pragma unroll
            for (int dt = 0; dt < DP / 16; ++dt) {
                bf16x4 ok = z4, ov = z4;                          // the pad columns of the 128-wide layout are written as zeros
                if (dt < DKR) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rks * (dk[dt < DKR ? dt : 0][r] - bf2f(kn[dt < DKR ? dt : 0][r]) * dot);
                    ok = f2bf4(v);
                    ov = f2bf4(dv[dt < DKR ? dt : 0]);
                }
                *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 16 * dt + 4 * g) = ok;
                *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 16 * dt + 4 * g) = ov;
            }
        }
        __syncthreads();                                          // phase 1 done with the Q tile; the dS image is complete

        // ---- k^ rows over the Q tile (transposed reads in phase 2)
        {
            uint4 sk[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = min(srow + RPP * p, Lp - 1), cc = sact ? scc : 0;
                sk[p] = *(const uint4*)(qkvh + slab0 + SLAB + (size_t)row * DP + cc * 8);
            }
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = srow + RPP * p;
                if (row < Lp && sact) *(uint4*)(Qs + row * QP + scc * 8) = sk[p];
            }
        }
        __syncthreads();

        // ================= phase 2: wave = query tile =================
        {
            f32x4 dq[DKR];
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // key tiles in pairs: the two tiles' transposed fragments concatenate to one K = 32 operand (same k order on both sides)
            auto frag = [&](int t, bf16x4 (&kt_)[DKR], bf16x4& ds_) {
                const int row = 16 * t + 4 * g + (fr >> 2);
#pragma unroll
                for (int dt = 0; dt < DKR; ++dt) kt_[dt] = lds_tr_read(Qs + row * QP + 16 * dt + (fr & 3) * 4);     // rows d, col key
                ds_ = lds_tr_read(dSb + row * DSP + 16 * tw + (fr & 3) * 4);                                     // B[k = key][n = q]
            };
#pragma unroll 1
            for (int t = 0; t + 1 < LT; t += 2) {
                bf16x4 k0[DKR], k1[DKR], d0, d1;
                frag(t, k0, d0);
                frag(t + 1, k1, d1);
#pragma unroll
                for (int dt = 0; dt < DKR; ++dt)
                    dq[dt] = mfma32(__builtin_shufflevector(k0[dt], k1[dt], 0, 1, 2, 3, 4, 5, 6, 7),
                                    __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7), dq[dt]);
            }
            if (LT & 1) {
                bf16x4 k0[DKR], d0;
                frag(LT - 1, k0, d0);
#pragma unroll
                for (int dt = 0; dt < DKR; ++dt) {
                    // own accumulator for the K = 16 tail (see attn_bwd_kernel)
                    const f32x4 tail = mfma16(k0[dt], d0, (f32x4){0.f, 0.f, 0.f, 0.f});
                    dq[dt] += tail;
                }
            }
            const int q = 16 * tw + fr;
            const float rq = rnorm[(((size_t)bw * h + hd) * 2 + 0) * Lp + q] * sigma;
            bf16x4 qn[DKR];
            float dot = 0.f;
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) {
                qn[dt] = *(const bf16x4*)(qkvh + slab0 + (size_t)q * DP + 16 * dt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) dot = fmaf(dq[dt][r], bf2f(qn[dt][r]), dot);
            }
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            const bf16x4 z4 = {0, 0, 0, 0};
#pragma unroll
            for (int dt = 0; dt < DP / 16; ++dt) {
                bf16x4 o = z4;
                if (dt < DKR) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rq * (dq[dt < DKR ? dt : 0][r] - bf2f(qn[dt < DKR ? dt : 0][r]) * dot);
                    o = f2bf4(v);
                }
                *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 16 * dt + 4 * g) = o;
            }
        }
        __syncthreads();                                          // the next window's staging overwrites the tiles
    }

    // ---- one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[tw] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < LT; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
}

// ------------------------------------------------------------------------------------------------
// forward for the same head widths: wave = query tile, swapped product S^T = K Q^T (a lane owns one query column), exactly the
// arithmetic of attn_fwd_kernel; K and V tiles of the real channels in LDS (pitch 112), the Q fragments straight from global
// memory into the K = 32 B operands, O^T = V^T P^T over PAIRS of key tiles (K = 32).  attn_fwd_kernel at 128 columns keeps one
// 90 KB K | V buffer, refills it between two barriers from 8 prefetch registers per thread and spills 36: 670 us per launch.
// ------------------------------------------------------------------------------------------------
template <int LT, int DKR, int LFIX, int DP>
__global__ __launch_bounds__(64 * LT) void attn_fwd_wide_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, uint16_t* __restrict__ oh, float* __restrict__ lse,
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int Lp = 16 * LT, SLAB = Lp * DP, NT = 64 * LT;
    constexpr int QP = 16 * DKR + 16, KS = DKR / 2;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * Lp * QP];
    uint16_t* const Ks = smem;
    uint16_t* const Vs = smem + Lp * QP;
    const int tid = threadIdx.x, lane = tid & 63, qt = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const int q = 16 * qt + fr;
    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;
    constexpr int RPP = NT / 16, PASSES = (Lp + RPP - 1) / RPP;
    const int srow = tid >> 4, scc = tid & 15;
    const bool sact = scc < 2 * DKR;
    const uint32_t nobias[LT][2] = {};

    for (int bw = blockIdx.x; bw < Bw; bw += gridDim.x) {
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        bf16x8 qf[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[kk] = *(const bf16x8*)(qkvh + slab0 + (size_t)q * DP + 32 * kk + 8 * g);
        {
            uint4 sk[PASSES], sv[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = min(srow + RPP * p, Lp - 1), cc = sact ? scc : 0;      // unconditional (clamped) loads
                sk[p] = *(const uint4*)(qkvh + slab0 + SLAB + (size_t)row * DP + cc * 8);
                sv[p] = *(const uint4*)(qkvh + slab0 + 2 * SLAB + (size_t)row * DP + cc * 8);
            }
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = srow + RPP * p;
                if (row < Lp && sact) {
                    *(uint4*)(Ks + row * QP + scc * 8) = sk[p];
                    *(uint4*)(Vs + row * QP + scc * 8) = sv[p];
                }
            }
        }
        __syncthreads();

        // S^T tiles: rows = keys 16 t + 4 g + r, column = query fr
        f32x4 acc[LT];
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (16 * t + fr) * QP + 32 * kk + 8 * g);
                acc[t] = mfma32(kf, qf[kk], acc[t]);
            }
        }
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        float mx, sum = 0.f;
        if (!do_mask) {          // sigma > 0 commutes with the maximum: the scale is folded into the exponent's fma (attn_fwd_kernel)
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
            mx = score_pass<LT, false, true, LFIX>(acc, nobias, sc2, L, g, mask_thr, q >= mask_thr);
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

        // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]: key tiles in pairs (the two tiles' fragments concatenate to one K = 32
        // operand, same k order on both sides); the odd last tile as a K = 16 product into its own accumulator
        f32x4 o[DKR];
#pragma unroll
        for (int dt = 0; dt < DKR; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t + 1 < LT; t += 2) {
            const bf16x8 pb = __builtin_shufflevector(f2bf4(acc[t]), f2bf4(acc[t + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) {
                const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                o[dt] = mfma32(__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), pb, o[dt]);
            }
        }
        if (LT & 1) {
            const bf16x4 pb = f2bf4(acc[LT - 1]);
#pragma unroll
            for (int dt = 0; dt < DKR; ++dt) {
                const bf16x4 vf = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * QP + 16 * dt + (fr & 3) * 4);
                const f32x4 tail = mfma16(vf, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                o[dt] += tail;
            }
        }
        const float inv = (q < L) ? 1.f / sum : 0.f;
        uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
        const bf16x4 z4 = {0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < DP / 16; ++dt) {
            bf16x4 ov = z4;                                    // the pad columns of the 128-wide layout are written as zeros
            if (dt < DKR) {
                f32x4 v = o[dt < DKR ? dt : 0];
                v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                ov = f2bf4(v);
            }
            *(bf16x4*)(orow + 16 * dt + 4 * g) = ov;
        }
        if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx + __log2f(sum) : 0.f;
        __syncthreads();                                          // the next window's staging overwrites the tiles
    }
}

}  // namespace

// 0 = launched, 1 = shape not covered (the caller falls back to attn_bwd_kernel), negative = error
static bool wide_off() { const char* e = getenv("SWV2_ATTN_WIDE"); return e && atoi(e) == 0; }      // (read per call: A/B and tests)

static bool wide_shape(const swv2_attn_args* a, int Lp, int DP) {
    return !wide_off() && !a->bias && Lp == 176 && (DP == 96 || DP == 128) && a->head_dim > 64 && a->head_dim <= 96 &&
           !(a->dbg & SWV2_ATTN_FIRST_GEN);
}

int swv2_attn_bwd_wide(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    if (!wide_shape(a, Lp, DP)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int nW = a->nwh * a->nww;
    // one workgroup (11 waves, 141 KB of LDS) per CU: persistent over the windows of its head
    int chunks = 256 / a->heads;
    if (chunks < 1) chunks = 1;
    if (chunks > a->Bw) chunks = a->Bw;
    dim3 grid(chunks, a->heads), block(64 * 11);
#define SWV2_LAUNCH_WIDE(LFIX, DPL)                                                                                                    \
    hipLaunchKernelGGL((attn_bwd_wide_kernel<11, 6, LFIX, DPL>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,         \
                       (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm, (uint16_t*)a->dqkvh, a->dlogit_scale, a->Bw, \
                       a->heads, a->L, nW, a->nww, a->nwh, a->mask_thr)
    if (DP == 96) { if (a->L == 162) SWV2_LAUNCH_WIDE(162, 96); else SWV2_LAUNCH_WIDE(0, 96); }
    else { if (a->L == 162) SWV2_LAUNCH_WIDE(162, 128); else SWV2_LAUNCH_WIDE(0, 128); }
#undef SWV2_LAUNCH_WIDE
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}

// the matching forward: same return convention
int swv2_attn_fwd_wide(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    if (!wide_shape(a, Lp, DP)) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int nW = a->nwh * a->nww;
    int chunks = 256 / a->heads;            // 79 KB of LDS: one workgroup of 11 waves per CU (two would leave 85 registers per wave)
    if (chunks < 1) chunks = 1;
    if (chunks > a->Bw) chunks = a->Bw;
    dim3 grid(chunks, a->heads), block(64 * 11);
#define SWV2_LAUNCH_WIDE(LFIX, DPL)                                                                                                    \
    hipLaunchKernelGGL((attn_fwd_wide_kernel<11, 6, LFIX, DPL>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,         \
                       (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, nW, a->nww, a->nwh, a->mask_thr)
    if (DP == 96) { if (a->L == 162) SWV2_LAUNCH_WIDE(162, 96); else SWV2_LAUNCH_WIDE(0, 96); }
    else { if (a->L == 162) SWV2_LAUNCH_WIDE(162, 128); else SWV2_LAUNCH_WIDE(0, 128); }
#undef SWV2_LAUNCH_WIDE
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}
