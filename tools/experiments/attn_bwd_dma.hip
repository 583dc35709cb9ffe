// Cosine window attention, backward at the benchmark head geometry (16-wide heads, no CPB bias, 176-row windows): the two-phase
// kernel of attn.hip (statistics inside the K = 32 MFMA operands, "AUG") with its staging taken off the critical path.
//
// attn_bwd_kernel moves every window through registers: 5 x 16-byte global loads per thread one window ahead, then -- behind the
// window's last barrier -- a commit (wait for the loads, LDS writes, delta = rowsum(dO O), the operand-carried statistics) and two
// more barriers.  Stamps (profiles/r02_stamps_attn_bwd.txt): prefetch issue + commit + barriers 2 / 3 are ~20 % of a window, and no
// wave does arithmetic meanwhile.  Here
//   * q | k | v, dO, O, lse and rnorm of the NEXT window go to a second LDS buffer by LDS-DMA (global_load_lds_dwordx4: 32 wave
//     instructions of 1 KB per window, three per wave, issued at the top of the window; no staging registers, no LDS writes);
//   * the slabs stay in their memory layout (32-byte rows: the 16-byte operand reads and the transposed reads of 16 consecutive rows
//     are conflict-free); the statistics halves of the K = 32 operands live in two small arrays (augq / augd: 16 bytes per query:
//     lse / (sigma log2 e) resp. delta in three bf16 parts, the constant 1, the mask-region flags) that lanes 32 .. 47 of an operand
//     read address instead of the slab (lanes 48 .. 63: a zero chunk) -- still ONE ds_read_b128 per operand;
//   * delta and the statistics of the next window are computed in phase 2 by the wave that owns the query tile (16 lanes: two rows of
//     dO, O from the landed buffer), i.e. while the other waves still run their dQ products;
//   * no global load is left in the loop (rnorm comes with the DMA), so every vmcnt wait is a counted one placed by hand, and a
//     window has two workgroup barriers instead of four.
// Semantics, operand construction and output layout are those of attn_bwd_kernel<11, 1, false, L, 1, true> (attn.hip: the comment
// above that kernel); reference: networks/swinv2_global.py:298-321 (backward of the cosine attention core).
#include "attn_common.h"

namespace {

__device__ __forceinline__ void a2_dma(const void* base, uint32_t byte_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base), "s"(lds_addr) : "memory");
}

#ifdef SWV2_ATTN2_STAMPS          // diagnostic build (tools/probe_attn1_stamps.py dma): per-phase s_memtime sums of every wave of the first workgroups
__device__ unsigned long long attn2_stamps[512 * 8];
#define HSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define HSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define HSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define HSTAMP_DECL
#define HSTAMP_START() do {} while (0)
#define HSTAMP(k) do {} while (0)
#endif

template <int LFIX>
__global__ __launch_bounds__(704) void attn_bwd_dma_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, const uint16_t* __restrict__ oh,
    const uint16_t* __restrict__ doh, const float* __restrict__ lse, const float* __restrict__ rnorm, uint16_t* __restrict__ dqkvh,
    float* __restrict__ dlogit, int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int LT = 11, Lp = 176, DP = 16, SLAB = Lp * DP, WAVES = 11, DSP = Lp + 4;
    // one staging buffer (bytes): q | k | v (17 DMA instructions, the last one half used) | rnorm (2) | per query tile: 16 rows of dO, then
    // 16 rows of O (1 KB = ONE instruction per tile, issued by the wave that builds the tile's statistics: 64-bit per-lane addresses)
    constexpr int B_QKV = 0, B_RN = 17 * 1024, B_DOO = B_RN + 2048, BUFB = B_DOO + LT * 1024;
    constexpr int NDMA = 19;                                       // shared pieces (q | k | v, rnorm): instruction ii = tw + 11 j
    constexpr int OFF_AUG = 2 * BUFB, AUGB = Lp * 16;              // per buffer: augq [Lp][8] | augd [Lp][8]
    constexpr int OFF_ZERO = OFF_AUG + 4 * AUGB, OFF_DS = OFF_ZERO + 64, OFF_RED = OFF_DS + Lp * DSP * 2, LDS_BYTES = OFF_RED + 64;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    uint16_t* const dSb = (uint16_t*)(lds + OFF_DS);
    float* const red = (float*)(lds + OFF_RED);

    const int tid = threadIdx.x, lane = tid & 63, tw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const int Lc = LFIX > 0 ? LFIX : L;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;

    const float tau = logit_scale[hd];
    const float sigma = __expf(fminf(tau, SWV2_LN100));
    const float sc2 = sigma * SWV2_LOG2E, inv_sc2 = 1.f / sc2;
    if (tid < 16) ((uint32_t*)(lds + OFF_ZERO))[tid] = 0u;

    // ---- DMA of window bw into buffer b; returns this lane's lse value of the window (lanes 0 .. 15: row 16 tw + lane), a plain load
    auto issue = [&](int bw, int b) -> float {
        const size_t item = (size_t)bw * h + hd;
        const unsigned char* pq = (const unsigned char*)(qkvh + item * 3 * SLAB);
        const unsigned char* pr = (const unsigned char*)(rnorm + item * 2 * Lp);
        const uint32_t lb = lds0 + (uint32_t)(b * BUFB);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ii = tw + 11 * j;                           // wave-uniform
            if (ii >= NDMA) break;
            const unsigned char* base = ii < 17 ? pq : pr;
            const int k = ii < 17 ? ii : ii - 17, valid = ii < 17 ? 3 * SLAB * 2 : 2 * Lp * 4;
            const uint32_t off = (uint32_t)min(k * 1024 + lane * 16, valid - 16);      // lanes past the source repeat its last chunk (into padding)
            a2_dma(base, off, lb + (uint32_t)(ii * 1024));
        }
        {   // this wave's query tile: lanes 0 .. 31 <- 16 rows of dO, lanes 32 .. 63 <- the same rows of O
            const unsigned char* src = (const unsigned char*)((lane < 32 ? doh : oh) + item * SLAB) + tw * 512 + (lane & 31) * 16;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lb + (uint32_t)(B_DOO + tw * 1024)) : "memory");
        }
        return lse[item * Lp + 16 * tw + (lane & 15)];
    };
    // ---- delta and the statistics halves of the window in buffer b, rows 16 tw .. + 15 (lanes 0 .. 15); ls = the row's lse
    auto build_aug = [&](int b, float ls) {
        if (lane < 16) {
            const int row = 16 * tw + lane;
            const unsigned char* bb = lds + b * BUFB;
            const unsigned char* dd = bb + B_DOO + tw * 1024 + lane * 32;
            const uint4 d0 = *(const uint4*)(dd), d1 = *(const uint4*)(dd + 16);
            const uint4 o0 = *(const uint4*)(dd + 512), o1 = *(const uint4*)(dd + 512 + 16);
            const uint32_t a[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, c[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dl = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(c[e] << 16), dl);
                dl = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(c[e] & 0xffff0000u), dl);
            }
            const bool q_ok = row < L;
            const float lq = q_ok ? ls * inv_sc2 : 1.0e30f;                  // padded query rows: P = 0
            const uint16_t l0 = f2bf(lq);
            const float r1 = lq - bf2f(l0);
            uint16_t l1 = f2bf(r1), l2 = f2bf(r1 - bf2f(l1));
            if (!q_ok) l1 = l2 = 0;
            const uint16_t e0 = f2bf(dl);
            const float e1f = dl - bf2f(e0);
            const uint16_t e1 = f2bf(e1f), e2 = f2bf(e1f - bf2f(e1));
            const uint32_t one = 0x3f80u, rqf = (row >= mask_thr) ? 0x3f80u : 0u;
            unsigned char* ab = lds + OFF_AUG + b * 2 * AUGB;
            *(uint4*)(ab + row * 16) = make_uint4(l0 | ((uint32_t)l1 << 16), l2 | (one << 16), rqf | ((one - rqf) << 16), 0);
            *(uint4*)(ab + AUGB + row * 16) = make_uint4(e0 | ((uint32_t)e1 << 16), e2, 0, 0);
        }
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    {
        const float ls0 = issue(bw, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        build_aug(0, ls0);
    }
    __syncthreads();

    float dsig = 0.f;
    // per-lane byte addresses of the two 16-byte A operands of a step: slab rows for lanes 0 .. 31 (channels 8 g ..), the statistics
    // row for lanes 32 .. 47, the zero chunk for lanes 48 .. 63; step per query tile: 512 / 256 / 0 bytes
    const int qa_step = g < 2 ? 512 : (g == 2 ? 256 : 0), da_step = g < 2 ? 1024 : (g == 2 ? 256 : 0);
    HSTAMP_DECL
    HSTAMP_START();
    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        const int bw_next = bw + gridDim.x;
        const size_t slab0 = ((size_t)bw * h + hd) * 3 * SLAB;
        float ls_next = 0.f;
        if (bw_next < Bw) ls_next = issue(bw_next, nxt);
        HSTAMP(0);                      // DMA issue of the next window
        const unsigned char* bb = lds + cur * BUFB;
        const uint16_t* Qs = (const uint16_t*)(bb + B_QKV);
        const uint16_t* Ks = Qs + SLAB;
        const uint16_t* Vs = Ks + SLAB;
        const uint16_t* dOs = (const uint16_t*)(bb + B_DOO);         // 16 rows per 1 KB (the tile's O rows behind them)
        const float* RNs = (const float*)(bb + B_RN);
        const unsigned char* augq = lds + OFF_AUG + cur * 2 * AUGB;
        const unsigned char* augd = augq + AUGB;

        // ================= phase 1: wave = key tile =================
        const int key = 16 * tw + fr;
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const float cmask = do_mask ? fmaxf(-100.f * SWV2_LOG2E * inv_sc2, -1.0e30f) : 0.f;
        bf16x8 kf8, vf8;
        bf16x4 kf4;
        {
            const uint32_t m1 = 0xbf80u;                                       // -1
            const uint32_t padk = (key < Lc) ? 0u : (uint32_t)f2bf(-1.0e30f);
            const bool kreg = key >= mask_thr;
            const uint32_t mk0 = f2bf(kreg ? 0.f : cmask), mk1 = f2bf(kreg ? cmask : 0.f);
            const uint4 augk = make_uint4(m1 | (m1 << 16), m1 | (padk << 16), mk0 | (mk1 << 16), 0);
            const uint4 augv = make_uint4(m1 | (m1 << 16), m1, 0, 0);
            const uint4 rk = *(const uint4*)(Ks + key * DP + (g & 1) * 8), rv = *(const uint4*)(Vs + key * DP + (g & 1) * 8);
            const uint4 z = make_uint4(0, 0, 0, 0);
            kf8 = __builtin_bit_cast(bf16x8, g < 2 ? rk : (g == 2 ? augk : z));
            vf8 = __builtin_bit_cast(bf16x8, g < 2 ? rv : (g == 2 ? augv : z));
            kf4 = *(const bf16x4*)(Ks + key * DP + 4 * g);
        }
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* qa_p = g < 2 ? (const unsigned char*)Qs + fr * 32 + g * 16 : (g == 2 ? augq + fr * 16 : lds + OFF_ZERO);
        const unsigned char* da_p = g < 2 ? (const unsigned char*)dOs + fr * 32 + g * 16 : (g == 2 ? augd + fr * 16 : lds + OFF_ZERO);
        const uint16_t* tq_p = Qs + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4;
        const uint16_t* td_p = dOs + (4 * g + (fr >> 2)) * DP + (fr & 3) * 4;
        uint16_t* ds_p = dSb + key * DSP + 4 * g;
        struct St { f32x4 s, dp; bf16x4 tq, td; };
        auto stageA = [&](const int qt, St& o) {
            const bf16x8 qa = *(const bf16x8*)(qa_p + qt * qa_step);
            const bf16x8 da = *(const bf16x8*)(da_p + qt * da_step);
            o.td = lds_tr_read(td_p + qt * 512);
            o.tq = lds_tr_read(tq_p + qt * (16 * DP));
            o.s = mfma32(qa, kf8, (f32x4){0.f, 0.f, 0.f, 0.f});
            o.dp = mfma32(da, vf8, (f32x4){0.f, 0.f, 0.f, 0.f});
        };
        auto stageB = [&](const int qt, const St& in) {
            f32x4 p, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __builtin_amdgcn_exp2f(in.s[r] * sc2);
                ds[r] = p[r] * in.dp[r];
            }
            const bf16x4 pb = f2bf4(p), dsb = f2bf4(ds);
            *(bf16x4*)(ds_p + 16 * qt) = dsb;
            dv = mfma16(in.td, pb, dv);
            dk = mfma16(in.tq, dsb, dk);
        };
        {
            St sa, sb;
            stageA(0, sa);
#pragma unroll 1
            for (int qt = 0; qt + 1 < LT; qt += 2) {
                stageA(qt + 1, sb);
                stageB(qt, sa);
                if (qt + 2 < LT) stageA(qt + 2, sa);
                stageB(qt + 1, sb);
            }
            stageB(LT - 1, sa);
        }
        HSTAMP(1);                      // phase 1 loop
        // ---- dK (through the L2-normalisation) and dV of this wave's key tile
        {
            const float rk = RNs[Lp + key];
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dk[r], bf2f(kf4[r]), dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            if (g == 0) dsig += dot;          // d logit_scale = sigma sum_k (sum_q dS q^) . k^  (attn.hip)
            const float rks = rk * sigma;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rks * (dk[r] - bf2f(kf4[r]) * dot);
            *(bf16x4*)(dqkvh + slab0 + SLAB + (size_t)key * DP + 4 * g) = f2bf4(v);
            *(bf16x4*)(dqkvh + slab0 + 2 * SLAB + (size_t)key * DP + 4 * g) = f2bf4(dv);
        }
        HSTAMP(2);                      // dK / dV normalisation backward + stores
        __syncthreads();
        HSTAMP(3);                      // barrier 1

        // ================= phase 2: wave = query tile =================
        {
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
            auto frag = [&](int t, bf16x4& kt_, bf16x4& ds_) {
                const int row = 16 * t + 4 * g + (fr >> 2);
                kt_ = lds_tr_read(Ks + row * DP + (fr & 3) * 4);                          // rows d, col key
                ds_ = lds_tr_read(dSb + row * DSP + 16 * tw + (fr & 3) * 4);              // B[k = key][n = q]
            };
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                bf16x4 k0, k1, d0, d1;
                frag(t, k0, d0);
                frag(t + 1, k1, d1);
                dq = mfma32(__builtin_shufflevector(k0, k1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7), dq);
            }
            {
                bf16x4 k0, d0;
                frag(LT - 1, k0, d0);
                const f32x4 tail = mfma16(k0, d0, (f32x4){0.f, 0.f, 0.f, 0.f});     // own accumulator (attn.hip: K = 16 onto K = 32)
                dq += tail;
            }
            const int q = 16 * tw + fr;
            const float rq = RNs[q] * sigma;
            const bf16x4 qn = *(const bf16x4*)(Qs + q * DP + 4 * g);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(dq[r], bf2f(qn[r]), dot);
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rq * (dq[r] - bf2f(qn[r]) * dot);
            *(bf16x4*)(dqkvh + slab0 + (size_t)q * DP + 4 * g) = f2bf4(v);
        }
        // the next window's DMA (issued at the top of this window: older than the three stores of dK, dV, dQ) has landed for this wave:
        // its own query tile's dO / O rows for the statistics now, the shared slabs for everybody behind the barrier
        HSTAMP(4);                      // phase 2: dQ + normalisation backward + store
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        HSTAMP(6);                      // wait for the next window's DMA
        if (bw_next < Bw) build_aug(nxt, ls_next);
        HSTAMP(7);                      // statistics of the next window
        __syncthreads();
        HSTAMP(5);                      // barrier 2
    }
#ifdef SWV2_ATTN2_STAMPS
    if (lane == 0 && blockIdx.y == 0 && blockIdx.x * WAVES + tw < 512)
        for (int k = 0; k < 8; ++k) attn2_stamps[(blockIdx.x * WAVES + tw) * 8 + k] = st_acc[k];
#endif

    // ---- one atomic per workgroup for the logit scale
    dsig = wave_sum(dsig);
    if (lane == 0) red[tw] = dsig;
    __syncthreads();
    if (tid == 0 && tau <= SWV2_LN100) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += red[i];
        atomicAdd(dlogit + hd, t * sigma);
    }
}

}  // namespace

#ifdef SWV2_ATTN2_STAMPS
extern "C" int swv2_debug_attn2_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(attn2_stamps), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -3;
}
#endif

// called by swv2_attn_bwd (attn.hip): 0 / negative = handled (ok / error), 1 = shape not covered
int swv2_attn_bwd_dma(const swv2_attn_args* a, int Lp, int DP, void* stream) {
    if (a->bias || Lp != 176 || DP != 16 || (a->dbg & (SWV2_ATTN_PLAIN_STATS | SWV2_ATTN_FIRST_GEN))) return 1;
    static const int use = getenv("SWV2_ATTN_BWD_DMA") ? atoi(getenv("SWV2_ATTN_BWD_DMA")) : 1;
    if (!use) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = a->Bw < a->max_chunks ? a->Bw : a->max_chunks;
    dim3 grid(nchunk, a->heads), block(704);
#define SWV2_BWD2_ARGS (const uint16_t*)a->qkvh, a->logit_scale, (const uint16_t*)a->oh, (const uint16_t*)a->doh, a->lse, a->rnorm, \
                       (uint16_t*)a->dqkvh, a->dlogit_scale, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr
    if (a->L == 162) hipLaunchKernelGGL((attn_bwd_dma_kernel<162>), grid, block, 0, st, SWV2_BWD2_ARGS);
    else hipLaunchKernelGGL((attn_bwd_dma_kernel<0>), grid, block, 0, st, SWV2_BWD2_ARGS);
#undef SWV2_BWD2_ARGS
    SWV2_CHECK_LAUNCH("swv2_attn_bwd");
    return SWV2_OK;
}
