// Output projection of the attention branch fused with LayerNorm1, forward and backward (reference
// swinv2_global.py:318-319 + 468-476 + 490):
//   forward : x1[dst] = x[dst] + scale * LN1( merge_heads(oh) Wp^T + b )          (dst = window-reverse / un-roll row table)
//   backward: da1 = LN1 backward of (scale * dx1[dst]) ; d(oh) = split_heads(da1 Wp)
// Each replaces a GEMM launch + a LayerNorm launch (forward 72 + 168 MB -> 204 MB, backward 140 + 72 MB -> 174 MB).  The
// products are single-pass (K, N <= 128): the whole weight sits in LDS, a wave owns 16 * MT rows, products are computed
// transposed (accumulator tile = rows of the next layout, see mlp.hip).  The forward's A rows come straight from the
// head-major attention output (16 tokens x 32 B = 512 B contiguous per head), its epilogue is the LayerNorm epilogue of
// mlp_fwd_kernel with the window-reverse / un-roll row table; the backward's prologue is the row-layout LayerNorm backward of mlp_bwd_kernel with a row gather,
// and its output tiles (rows = one head's 16 channels, column = token) store directly as 512-byte head-major runs.
#include <cstdlib>
#include "gemm_common.h"

namespace {

constexpr int PL_MAX_HDP = 128;

// The weight into LDS with a compile-time trip count and the loads in batches of up to 6 before their LDS writes.  (As a loop over
// the run-time chunk count, `dst[i] = src[i]`, every iteration was a memory round trip of its own: 8 in series per workgroup at 128
// channels, 12 at 192.)  ROWS x COLS = the largest shape; rows >= `rows` / columns >= `cols` are skipped; source rows are `cols` wide.
template <int ROWS, int COLS, int PITCH, int NT>
__device__ __forceinline__ void stage_weight(uint16_t* dst, const uint16_t* __restrict__ src, int rows, int cols, int tid) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CPRW = COLS / 8, CHUNKS = ROWS * CPRW, PER = (CHUNKS + NT - 1) / NT, B = PER < 6 ? PER : 6;
#pragma unroll
    for (int j0 = 0; j0 < PER; j0 += B) {
        u32x4 v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int i = tid + (j0 + u) * NT, r = i / CPRW, c8 = i % CPRW;
            v[u] = *(const u32x4*)(src + (size_t)min(r, rows - 1) * cols + min(8 * c8, cols - 8));        // unconditional (clamped)
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int i = tid + (j0 + u) * NT, r = i / CPRW, c8 = i % CPRW;
            if (j0 + u < PER && r < rows && 8 * c8 < cols) *(u32x4*)(dst + r * PITCH + 8 * c8) = v[u];
        }
    }
}

struct ProjLnFwd {
    const uint16_t* oh; const uint16_t* wp; const float* bp; const float* gamma; const float* beta; const float* scale;
    const int32_t* rowidx; const float* x; uint16_t* a1; float* mean; float* rstd; float* y;
    int Mw, Lp, h, rows_per_sample; float eps;
};

// HS = head slot width (16, or 32: BASELINE configs[4], 192 channels in 8 heads of 24), NW = waves per workgroup: with the 192 x 256
// weight (101 KB) one workgroup fits a CU, so it is 8 waves instead of 2 x 4.
template <int C, int MT, int HS = 16, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void proj_ln_fwd_kernel(const ProjLnFwd a) {
    constexpr int NTC = C / 16, NT = 64 * NW;
    constexpr int HDPM = 8 * HS;                             // at most 8 heads
    constexpr int PWP = HDPM + 8, PA = C + 8;
    constexpr int EWAVE = 16 * PA * 2 + 16 * 2 * 4;
    constexpr int KSM = HDPM / 32;
    static_assert(C * PWP * 2 + NW * EWAVE + 3 * C * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) uint16_t Wps[C * PWP];
    __shared__ __attribute__((aligned(16))) unsigned char epi[NW * EWAVE];
    __shared__ __attribute__((aligned(16))) float cs[3 * C];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int hdp = a.h * HS, ksn = hdp / 32;
    const int row0 = blockIdx.x * (16 * NW * MT) + wave * (16 * MT);

    // The scatter rows (window reverse + un-roll table) and the residual rows of the EPILOGUE are fetched first: as loads
    // inside the epilogue they were two dependent memory round trips per row tile with nothing to overlap them (a
    // workgroup lived through five round trips: A + weight, then table -> residual for each of its two tiles; 56 us).
    // Issued here they overlap the A / weight fetch and the MFMAs.  Unconditional (clamped) loads: the compiler can
    // count them, so the wait for the weight does not also wait for the residual rows.
    constexpr int UNITS = 16 * (C / 8), NP = (UNITS + 63) / 64;
    int dstv[MT][NP];
    f32x4 xr[MT][NP][2];
    float scv[MT][NP];                      // drop-path scale of the destination row's sample (unconditional load, see below)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int u = min(lane + 64 * p, UNITS - 1), row = u / (C / 8);
            const int m = min(row0 + 16 * mt + row, a.Mw - 1);
            dstv[mt][p] = a.rowidx ? a.rowidx[m] : m;
        }
    // A^T fragments straight from the head-major attention output: lane (m = fr, g) holds k = 32 ks + 8 g .. + 7,
    // i.e. 8 channels of head 2 ks + (g >> 1) (HS = 16) resp. of head ks (HS = 32)
    bf16x8 xf[MT][KSM];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = min(row0 + 16 * mt + fr, a.Mw - 1), w = m / a.Lp, t = m - w * a.Lp;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            const int k0 = 32 * ks + 8 * g, kc = min(k0, hdp - 8);
            xf[mt][ks] = *(const bf16x8*)(a.oh + (((size_t)w * a.h + kc / HS) * a.Lp + t) * HS + (kc & (HS - 1)));
        }
    }
    stage_weight<C, HDPM, PWP, NT>(Wps, a.wp, C, hdp, tid);
    for (int i = tid; i < C; i += NT) { cs[i] = a.bp[i]; cs[C + i] = a.gamma[i]; cs[2 * C + i] = a.beta[i]; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int u = min(lane + 64 * p, UNITS - 1), c8 = u % (C / 8);
            const size_t off = (size_t)max(dstv[mt][p], 0) * C + 8 * c8;
            xr[mt][p][0] = *(const f32x4*)(a.x + off);
            xr[mt][p][1] = *(const f32x4*)(a.x + off + 4);
            // (as `a.scale ? a.scale[...] : 1.f` inside the pass loop this was a conditional load with its own vmcnt(0))
            scv[mt][p] = (a.scale ? a.scale : a.gamma)[a.scale ? max(dstv[mt][p], 0) / a.rows_per_sample : 0];
        }
    __syncthreads();

    f32x4 yacc[MT][NTC];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tn = 0; tn < NTC; ++tn) yacc[mt][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        if (ks < ksn) {
#pragma unroll
            for (int tn = 0; tn < NTC; ++tn) {
                const bf16x8 wf = *(const bf16x8*)(Wps + (16 * tn + fr) * PWP + 32 * ks + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) yacc[mt][tn] = mfma32(wf, xf[mt][ks], yacc[mt][tn]);
            }
        }
    }

    // epilogue: + bias, bf16 round (saved), LayerNorm in the accumulator layout, row layout through a per-wave LDS tile
    uint16_t* As = (uint16_t*)(epi + wave * EWAVE);
    float* St = (float*)(As + 16 * PA);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int tn = 0; tn < NTC; ++tn) {
            const f32x4 bv = *(const f32x4*)(cs + 16 * tn + 4 * g);
            const bf16x4 ar = f2bf4(yacc[mt][tn] + bv);
            *(bf16x4*)(As + fr * PA + 16 * tn + 4 * g) = ar;
#pragma unroll
            for (int e = 0; e < 4; ++e) { yacc[mt][tn][e] = bf2f(ar[e]); s += yacc[mt][tn][e]; }
        }
        s = xor32_allsum(xor16_allsum(s));
        const float mu = s * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int tn = 0; tn < NTC; ++tn)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = yacc[mt][tn][e] - mu; q = fmaf(d, d, q); }
        q = xor32_allsum(xor16_allsum(q));
        const float rs = rsqrtf(q * (1.f / C) + a.eps);
        if (g == 0) {
            const int m = min(row0 + 16 * mt + fr, a.Mw - 1);
            a.mean[m] = mu;
            a.rstd[m] = rs;
            St[2 * fr] = mu;
            St[2 * fr + 1] = rs;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int u = lane + 64 * p, row = u / (C / 8), c8 = u % (C / 8);
            if (UNITS % 64 == 0 || u < UNITS) {
                const int m = min(row0 + 16 * mt + row, a.Mw - 1);
                const u32x4 av = *(const u32x4*)(As + row * PA + 8 * c8);
                *(u32x4*)(a.a1 + (size_t)m * C + 8 * c8) = av;
                const int dst = dstv[mt][p];
                if (dst >= 0) {
                    const float mu_r = St[2 * row], rs_r = St[2 * row + 1];
                    const float sc = a.scale ? scv[mt][p] : 1.f;
                    const size_t off = (size_t)dst * C + 8 * c8;
                    float v[8];
                    unpack8(__builtin_bit_cast(uint4, av), v);
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const f32x4 gm = *(const f32x4*)(cs + C + 8 * c8 + 4 * hlf), bt = *(const f32x4*)(cs + 2 * C + 8 * c8 + 4 * hlf);
                        f32x4 o = xr[mt][p][hlf];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += sc * ((v[4 * hlf + e] - mu_r) * rs_r * gm[e] + bt[e]);
                        *(f32x4*)(a.y + off + 4 * hlf) = o;
                    }
                }
            }
        }
        if (mt + 1 < MT) __syncthreads();
    }
}

struct ProjLnBwd {
    const float* dy; const uint16_t* a1; const float* mean; const float* rstd; const float* gamma; const float* scale;
    const int32_t* rowidx; const uint16_t* wpt; uint16_t* da1; uint16_t* doh; float* ws;
    int Mw, Lp, h, rows_per_sample;
};

template <int C, int MT, int HS = 16, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void proj_ln_bwd_kernel(const ProjLnBwd a) {
    constexpr int KS = C / 32, NT = 64 * NW, HDPM = 8 * HS;
    constexpr int ROWS = 16 * NW * MT, PX = C + 8, PW = C + 8;
    constexpr int LPR = C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64;
    constexpr int RPP = NT / LPR, NPASS = ROWS / RPP;
    constexpr int BATCH = NPASS < 4 ? NPASS : 4;
    static_assert(NPASS % BATCH == 0, "row passes must come in whole batches");
    // d gamma / d beta partials of the RPP row groups: summed through LDS -- in two rounds (upper half of the row groups onto the
    // lower half, then across the lower half) where the one-round table would not fit beside the 102 KB weight (NW = 8)
    constexpr bool TWO_ROUNDS = NW == 8;
    constexpr int GROWS = TWO_ROUNDS ? RPP / 2 : RPP;
    static_assert(HDPM * PW * 2 + ROWS * PX * 2 + GROWS * 2 * C * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) uint16_t Wts[HDPM * PW];              // Wp^T rows: [heads * HS][C]
    __shared__ __attribute__((aligned(16))) uint16_t Xs[ROWS * PX];               // da1 tile
    __shared__ __attribute__((aligned(16))) float gs[GROWS * 2 * C];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int hdp = a.h * HS;
    const int wg_row0 = blockIdx.x * ROWS;

    stage_weight<HDPM, C, PW, NT>(Wts, a.wpt, hdp, C, tid);
    // ---- LayerNorm backward in row layout (see mlp_bwd_kernel), dy rows gathered through the row table; padded rows
    // (table < 0) give da1 = 0 and count nothing
    {
        const int lr = tid % LPR, rg = tid / LPR;
        const bool act = 4 * lr < C;
        const int c0 = act ? 4 * lr : 0;
        const f32x4 gm = *(const f32x4*)(a.gamma + c0);
        f32x4 dgm = {0.f, 0.f, 0.f, 0.f}, dbt = {0.f, 0.f, 0.f, 0.f};
        int srcv[NPASS];                       // the row table of all passes first: one (L2) round trip instead of one per batch
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int m = min(wg_row0 + i * RPP + rg, a.Mw - 1);
            srcv[i] = a.rowidx ? a.rowidx[m] : m;
        }
        // two batches of row loads in flight: batch b + 1 is issued BEFORE batch b's results are stored (the compiler cannot move
        // the loads above those stores itself -- da1 may alias dy as far as it knows -- so each batch used to be a full memory
        // round trip behind the previous one: four in series per workgroup)
        f32x4 d4s[2][BATCH];
        u32x2 a4s[2][BATCH];
        float mus[2][BATCH], rss[2][BATCH], scs[2][BATCH];
        int oks[2][BATCH];
        const float* scp = a.scale ? a.scale : a.gamma;          // unconditional load; ignored without a scale
        auto load_batch = [&](int pb, auto set_c) {
            constexpr int S = decltype(set_c)::value;
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int m = min(wg_row0 + (pb + i) * RPP + rg, a.Mw - 1);
                const int src = srcv[(pb + i) % NPASS];
                oks[S][i] = src >= 0;
                const int sr = max(src, 0);
                d4s[S][i] = *(const f32x4*)(a.dy + (size_t)sr * C + c0);
                a4s[S][i] = *(const u32x2*)(a.a1 + (size_t)m * C + c0);
                mus[S][i] = a.mean[m];
                rss[S][i] = a.rstd[m];
                scs[S][i] = scp[a.scale ? sr / a.rows_per_sample : 0];
            }
        };
        load_batch(0, std::integral_constant<int, 0>{});
#pragma unroll
        for (int pb = 0; pb < NPASS; pb += BATCH) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = (pb / BATCH) & 1;
            if (pb + BATCH < NPASS) {
                if (cur == 0) load_batch(pb + BATCH, std::integral_constant<int, 1>{});
                else load_batch(pb + BATCH, std::integral_constant<int, 0>{});
            }
            f32x4 d4[BATCH];
            u32x2 a4[BATCH];
            float mu[BATCH], rs[BATCH], sc[BATCH];
            int ok[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                d4[i] = d4s[cur][i]; a4[i] = a4s[cur][i]; mu[i] = mus[cur][i]; rs[i] = rss[cur][i];
                sc[i] = a.scale ? scs[cur][i] : 1.f; ok[i] = oks[cur][i];
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int row = (pb + i) * RPP + rg;
                const bool live = act && ok[i];
                const float once = (live && wg_row0 + row < a.Mw) ? 1.f : 0.f;
                const float av[4] = {__uint_as_float(a4[i][0] << 16), __uint_as_float(a4[i][0] & 0xffff0000u),
                                     __uint_as_float(a4[i][1] << 16), __uint_as_float(a4[i][1] & 0xffff0000u)};
                float gg[4], xh[4], t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = live ? sc[i] * d4[i][e] : 0.f;
                    xh[e] = (av[e] - mu[i]) * rs[i];
                    gg[e] = d * gm[e];
                    dgm[e] = fmaf(once * d, xh[e], dgm[e]);
                    dbt[e] = fmaf(once, d, dbt[e]);
                    t1 += gg[e];
                    t2 = fmaf(gg[e], xh[e], t2);
                }
                t1 = group_allsum<LPR>(t1); t2 = group_allsum<LPR>(t2);        // (vector ALU only: common.h)
                t1 *= (1.f / C); t2 *= (1.f / C);
                f32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = ok[i] ? rs[i] * (gg[e] - t1 - xh[e] * t2) : 0.f;
                const bf16x4 ob = f2bf4(o4);
                if (act) {
                    *(bf16x4*)(a.da1 + (size_t)min(wg_row0 + row, a.Mw - 1) * C + c0) = ob;
                    *(bf16x4*)(Xs + row * PX + c0) = ob;
                }
            }
        }
        if constexpr (TWO_ROUNDS) {
            if (act && rg >= GROWS) {
                *(f32x4*)(gs + ((rg - GROWS) * 2 + 0) * C + c0) = dgm;
                *(f32x4*)(gs + ((rg - GROWS) * 2 + 1) * C + c0) = dbt;
            }
            __syncthreads();
            if (act && rg < GROWS) {
                dgm += *(const f32x4*)(gs + (rg * 2 + 0) * C + c0);
                dbt += *(const f32x4*)(gs + (rg * 2 + 1) * C + c0);
            }
            __syncthreads();
        }
        if (act && rg < GROWS) {
            *(f32x4*)(gs + (rg * 2 + 0) * C + c0) = dgm;
            *(f32x4*)(gs + (rg * 2 + 1) * C + c0) = dbt;
        }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < GROWS; ++r) t += gs[r * 2 * C + i];
            a.ws[(size_t)blockIdx.x * 2 * C + i] = t;
        }
    }
    // ---- d(oh)^T[k][m] = sum_c Wp^T[k][c] da1[m][c]: one 16 x 16 tile per (head, row tile), stored head-major
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bf16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(Xs + (wave * 16 * MT + 16 * mt + fr) * PX + 32 * ks + 8 * g);
        const int m = min(wg_row0 + wave * 16 * MT + 16 * mt + fr, a.Mw - 1), w = m / a.Lp, t = m - w * a.Lp;
        for (int kt = 0; kt < a.h * (HS / 16); ++kt) {              // 16-channel tile kt of the head-major columns
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(Wts + (16 * kt + fr) * PW + 32 * ks + 8 * g);
                acc = mfma32(wf, xf[ks], acc);
            }
            *(bf16x4*)(a.doh + (((size_t)w * a.h + kt / (HS / 16)) * a.Lp + t) * HS + 16 * (kt % (HS / 16)) + 4 * g) = f2bf4(acc);
        }
    }
}

}  // namespace

extern "C" int swv2_proj_ln_supported(int C, int heads, int head_pad) {
    if (C == 192) return head_pad == 32 && heads >= 1 && heads <= 8;             // (32-wide slots: a K = 32 step is one head)
    return (C == 32 || C == 64 || C == 96 || C == 128) && head_pad == 16 && heads >= 2 && heads % 2 == 0 && heads * 16 <= PL_MAX_HDP;
}

extern "C" size_t swv2_proj_ln_bwd_ws_floats(int Mw, int C) { return (Mw > 0 && C > 0) ? (size_t)cdiv(Mw, 64) * 2 * C : 0; }

extern "C" int swv2_proj_ln_fwd(const swv2_proj_ln_args* a, void* stream) {
    SWV2_CHECK_ARG(a && a->oh && a->wp && a->bp && a->gamma && a->beta && a->x && a->a1 && a->mean && a->rstd && a->y,
                   "swv2_proj_ln_fwd: null pointer");
    SWV2_CHECK_ARG(a->Bw > 0 && a->Lp > 0 && a->Lp % 16 == 0 && a->rows_per_sample > 0, "swv2_proj_ln_fwd: bad geometry");
    if (!swv2_proj_ln_supported(a->C, a->heads, a->C == 192 ? 32 : 16)) {
        swv2_set_error("swv2_proj_ln_fwd: C=%d heads=%d not instantiated; use swv2_linear + swv2_ln_residual_fwd", a->C, a->heads);
        return SWV2_ERR_UNSUPPORTED;
    }
    const int Mw = a->Bw * a->Lp;
    ProjLnFwd k = {(const uint16_t*)a->oh, (const uint16_t*)a->wp, a->bp, a->gamma, a->beta, a->scale, a->rowidx, a->x,
                   (uint16_t*)a->a1, a->mean, a->rstd, a->y, Mw, a->Lp, a->heads, a->rows_per_sample, a->eps};
    hipStream_t st = (hipStream_t)stream;
    static const int force_mt = getenv("SWV2_PL_MT") ? atoi(getenv("SWV2_PL_MT")) : 0;
    const bool mt2 = force_mt ? force_mt == 2 : Mw >= 128 * 256;
#define PL_CASE(CC)                                                                                                  \
    case CC:                                                                                                         \
        if (mt2) hipLaunchKernelGGL((proj_ln_fwd_kernel<CC, 2>), dim3(cdiv(Mw, 128)), dim3(256), 0, st, k);          \
        else hipLaunchKernelGGL((proj_ln_fwd_kernel<CC, 1>), dim3(cdiv(Mw, 64)), dim3(256), 0, st, k);               \
        break;
    switch (a->C) {
        PL_CASE(32) PL_CASE(64) PL_CASE(96) PL_CASE(128)
        case 192: hipLaunchKernelGGL((proj_ln_fwd_kernel<192, 1, 32, 8>), dim3(cdiv(Mw, 128)), dim3(512), 0, st, k); break;
    }
#undef PL_CASE
    SWV2_CHECK_LAUNCH("swv2_proj_ln_fwd");
    return SWV2_OK;
}

extern "C" int swv2_proj_ln_bwd(const swv2_proj_ln_bwd_args* a, void* stream) { return swv2_proj_ln_bwd_impl(a, stream, nullptr); }

int swv2_proj_ln_bwd_impl(const swv2_proj_ln_bwd_args* a, void* stream, int* deferred) {
    SWV2_CHECK_ARG(a && a->dy && a->a1 && a->mean && a->rstd && a->gamma && a->wpt && a->da1 && a->doh && a->dgamma && a->dbeta &&
                       a->ws, "swv2_proj_ln_bwd: null pointer");
    SWV2_CHECK_ARG(a->Bw > 0 && a->Lp > 0 && a->Lp % 16 == 0 && a->rows_per_sample > 0, "swv2_proj_ln_bwd: bad geometry");
    if (!swv2_proj_ln_supported(a->C, a->heads, a->C == 192 ? 32 : 16)) {
        swv2_set_error("swv2_proj_ln_bwd: C=%d heads=%d not instantiated; use swv2_ln_residual_bwd + swv2_linear", a->C, a->heads);
        return SWV2_ERR_UNSUPPORTED;
    }
    const int Mw = a->Bw * a->Lp;
    ProjLnBwd k = {a->dy, (const uint16_t*)a->a1, a->mean, a->rstd, a->gamma, a->scale, a->rowidx, (const uint16_t*)a->wpt,
                   (uint16_t*)a->da1, (uint16_t*)a->doh, a->ws, Mw, a->Lp, a->heads, a->rows_per_sample};
    hipStream_t st = (hipStream_t)stream;
    static const int force_mtb = getenv("SWV2_PL_MT_BWD") ? atoi(getenv("SWV2_PL_MT_BWD")) : 0;
    const bool mt2 = force_mtb ? force_mtb == 2 : Mw >= 128 * 256;
#define PL_CASE(CC)                                                                                                  \
    case CC:                                                                                                         \
        if (mt2) hipLaunchKernelGGL((proj_ln_bwd_kernel<CC, 2>), dim3(cdiv(Mw, 128)), dim3(256), 0, st, k);          \
        else hipLaunchKernelGGL((proj_ln_bwd_kernel<CC, 1>), dim3(cdiv(Mw, 64)), dim3(256), 0, st, k);               \
        break;
    int nblk = cdiv(Mw, mt2 ? 128 : 64);
    switch (a->C) {
        PL_CASE(32) PL_CASE(64) PL_CASE(96) PL_CASE(128)
        case 192: nblk = cdiv(Mw, 128); hipLaunchKernelGGL((proj_ln_bwd_kernel<192, 1, 32, 8>), dim3(nblk), dim3(512), 0, st, k); break;
    }
#undef PL_CASE
    if (deferred) *deferred = nblk;
    else swv2_launch_ln_partials_reduce(a->ws, a->dgamma, a->dbeta, nblk, a->C, st);
    SWV2_CHECK_LAUNCH("swv2_proj_ln_bwd");
    return SWV2_OK;
}
