// Fused attention branch of a SwinV2 block, forward, for blocks WITHOUT the CPB bias (reference
// swinv2_global.py:446-478 + 170-198 + 490):
//     x1 = x + drop_path1( LayerNorm1( proj( W-MSA( roll/partition(x) ) ) ) )      scattered back through reverse / un-roll
// as ONE kernel instead of four (qkv GEMM, attention core, proj GEMM, LayerNorm kernel): 559 MB -> 313 MB of HBM traffic
// per block at the benchmark shape, and the q / k / v / o tiles of a window never make a round trip between kernels.
// What the backward needs is still written exactly as the unfused kernels write it (qkvh, rnorm, oh, lse, a1, mean,
// rstd), so the backward path is unchanged.
//
// One workgroup = one window at a time, LT waves, wave = 16 tokens (query tile) for the whole kernel:
//   x rows of the window (gathered through the roll/partition table) -> B fragments in registers, once per window
//   per head: (q|k|v)^T = W_h X^T (MFMA K = 32; W_h = the head's 48 weight rows, streamed through LDS, double
//             buffered) -> bias, L2-normalise q and k in the accumulator layout (a token's 16 channels sit in 4 lanes)
//             -> k, v to an LDS slab (all waves need all keys), q stays in registers: the accumulator tile
//             (row = channel 4g+r, column = token) IS the B operand of S^T = K Q^T
//             -> softmax exactly as attn_fwd_kernel -> O^T = V^T P^T -> the head's 16 output channels of the wave's
//             tokens go to a per-wave LDS tile (and to HBM for the backward)
//   after the last head: Y^T = Wp O^T from that tile (MFMA K = 32; keeping the Y^T accumulators live through the head
//   loop instead cost 32 VGPRs and spilled), + proj bias, bf16 round (saved), LayerNorm1, drop-path, residual,
//   scatter -- the epilogue of mlp_fwd_kernel with a row table.
#include "attn_common.h"
#include "gemm_common.h"

namespace {

constexpr int AF_MAX_HDP = 128;          // heads * 16 <= 128

struct AttnBranch {
    const float* x; const int32_t* rowidx; const uint16_t* wqkv; const float* bqkv; const uint16_t* wproj;
    const float* bproj; const float* logit_scale; const float* gamma; const float* beta; const float* scale;
    uint16_t* qkvh; float* rnorm; uint16_t* oh; float* lse; uint16_t* a1; float* mean; float* rstd; float* y;
    int Bw, h, L, nW, nww, nwh, mask_thr, rows_per_sample; float eps;
};

template <int LT, int C, int LFIX>
__global__ __launch_bounds__(64 * LT) void attn_branch_fwd_kernel(const AttnBranch a) {
    constexpr int Lp = 16 * LT, SLAB = Lp * 16, NT = 64 * LT;
    constexpr int KS = C / 32, NTC = C / 16;
    constexpr int PQ = C + 8;                       // pitch of the per-head qkv weight rows
    constexpr int PWP = AF_MAX_HDP + 8;             // pitch of the proj weight rows
    constexpr int PA = (C > AF_MAX_HDP ? C : AF_MAX_HDP) + 8;     // pitch of the per-wave tile (O of all heads, then a1)
    constexpr int EWAVE = 16 * PA * 2 + 16 * 2 * 4;
    constexpr int WQ_PIECES = 48 * (C / 8);         // 16-byte pieces of one head's qkv weight rows
    constexpr int SPT = (WQ_PIECES + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) uint16_t Wps[C * PWP];
    __shared__ __attribute__((aligned(16))) uint16_t Wqs[2][48 * PQ];
    __shared__ __attribute__((aligned(16))) uint16_t KVs[2][2 * SLAB];     // [buf][K | V][Lp][16]
    __shared__ __attribute__((aligned(16))) unsigned char epi[LT * EWAVE];
    __shared__ __attribute__((aligned(16))) float cs[3 * C];               // proj bias | gamma | beta
    __shared__ __attribute__((aligned(16))) float bq[3 * AF_MAX_HDP];      // qkv bias (a global load inside the head loop
    __shared__ float sc2s[AF_MAX_HDP / 16];                                // would drain the weight prefetch), sigma * log2 e
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63, tw = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int h = a.h, hdp = h * 16;
    const int t = 16 * tw + fr;                     // this lane's token (query) in the window
    const int Lc = LFIX > 0 ? LFIX : a.L;
    const bool valid = t < Lc;

    // ---- once per workgroup: proj weight + row constants -> LDS
    for (int i = tid; i < C * (hdp / 8); i += NT) {
        const int r = i / (hdp / 8), c8 = i % (hdp / 8);
        *(u32x4*)(Wps + r * PWP + 8 * c8) = *(const u32x4*)(a.wproj + (size_t)r * hdp + 8 * c8);
    }
    for (int i = tid; i < C; i += NT) { cs[i] = a.bproj[i]; cs[C + i] = a.gamma[i]; cs[2 * C + i] = a.beta[i]; }
    for (int i = tid; i < 3 * hdp; i += NT) bq[(i / hdp) * AF_MAX_HDP + i % hdp] = a.bqkv[i];
    if (tid < h) sc2s[tid] = __expf(fminf(a.logit_scale[tid], SWV2_LN100)) * SWV2_LOG2E;

    // ---- per-head qkv weight rows (48 x C), staged through registers one head ahead
    u32x4 sw[SPT];
    auto issue_w = [&](int hd) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = min(tid + NT * i, WQ_PIECES - 1), row = idx / (C / 8), c8 = idx % (C / 8);
            const int part = row >> 4, j = row & 15;
            sw[i] = *(const u32x4*)(a.wqkv + (size_t)(part * hdp + hd * 16 + j) * C + 8 * c8);
        }
    };
    auto commit_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx % (C / 8);
            if (idx < WQ_PIECES) *(u32x4*)(&Wqs[buf][row * PQ + 8 * c8]) = sw[i];
        }
    };
    // x^T fragments of this lane's token: lane (m = fr, g) holds c = 32 ks + 8 g .. + 7 (zeros for padded tokens)
    bf16x8 xf[KS];
    auto load_x = [&](int w, bf16x8 (&dst)[KS]) {
        const int src = a.rowidx[(size_t)w * Lp + t];
        const float* p = a.x + (size_t)max(src, 0) * C + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            RawF32 raw = {*(const f32x4*)(p + 32 * ks), *(const f32x4*)(p + 32 * ks + 4)};
            uint4 pk = cvt_f32x8(raw);
            if (src < 0) pk = make_uint4(0, 0, 0, 0);
            dst[ks] = __builtin_bit_cast(bf16x8, pk);
        }
    };

    int w = blockIdx.x;
    if (w >= a.Bw) return;
    issue_w(0);
    load_x(w, xf);
    commit_w(0);
    __syncthreads();

    int it = 0;                                     // running (window, head) counter: LDS buffer parity
    for (; w < a.Bw; w += gridDim.x) {
        const int w_next = w + gridDim.x;
        const bool do_mask = (a.mask_thr > 0) && (((w % a.nW) / a.nww) == a.nwh - 1);
        uint16_t* As = (uint16_t*)(epi + tw * EWAVE);      // per-wave tile: O of all heads [16 tokens][heads * 16], later a1
        float* St = (float*)(As + 16 * PA);
        for (int hd = 0; hd < h; ++hd, ++it) {
            const int buf = it & 1;
            issue_w((hd + 1) % h);                   // lands during the QKV MFMAs; staging registers die at the barrier
            const uint16_t* Wq = Wqs[buf];
            uint16_t* Ks = KVs[buf];
            uint16_t* Vs = Ks + SLAB;
            // ---- (q | k | v)^T of this wave's 16 tokens: rows = channel 4g + r, column = token fr
            f32x4 qkv[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                f32x4 acc = *(const f32x4*)(bq + s * AF_MAX_HDP + hd * 16 + 4 * g);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 wf = *(const bf16x8*)(Wq + (16 * s + fr) * PQ + 32 * ks + 8 * g);
                    acc = mfma32(wf, xf[ks], acc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = valid ? acc[e] : 0.f;           // padded tokens are zero rows
                qkv[s] = acc;
            }
            float rn[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(qkv[s][e], qkv[s][e], ss);
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                rn[s] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int e = 0; e < 4; ++e) qkv[s][e] *= rn[s];
            }
            const size_t wh = (size_t)w * h + hd;
            if (g == 0) {
                a.rnorm[(wh * 2 + 0) * Lp + t] = valid ? rn[0] : 0.f;
                a.rnorm[(wh * 2 + 1) * Lp + t] = valid ? rn[1] : 0.f;
            }
            const bf16x4 qb = f2bf4(qkv[0]), kb = f2bf4(qkv[1]), vb = f2bf4(qkv[2]);
            *(bf16x4*)(Ks + t * 16 + 4 * g) = kb;
            *(bf16x4*)(Vs + t * 16 + 4 * g) = vb;
            // next head's weights -> the other buffer (loaded during this head's QKV phase), visible after the barrier
            commit_w(buf ^ 1);
            __syncthreads();
            uint16_t* qdst = a.qkvh + wh * 3 * SLAB + (size_t)t * 16 + 4 * g;
            *(bf16x4*)qdst = qb;
            *(bf16x4*)(qdst + SLAB) = kb;
            *(bf16x4*)(qdst + 2 * SLAB) = vb;

            // ---- S^T = K Q^T (rows = keys 16 t2 + 4g + r, column = query fr), softmax, O^T = V^T P^T
            const float sc2 = sc2s[hd];
            f32x4 acc[LT];
#pragma unroll
            for (int t2 = 0; t2 < LT; ++t2) {
                const bf16x4 kf = *(const bf16x4*)(Ks + (16 * t2 + fr) * 16 + 4 * g);
                acc[t2] = mfma16(kf, qb, (f32x4){0.f, 0.f, 0.f, 0.f});
            }
            const uint32_t nob[LT][2] = {};
            float mx;
            if (do_mask) mx = score_pass<LT, false, true, LFIX>(acc, nob, sc2, a.L, g, a.mask_thr, t >= a.mask_thr);
            else         mx = score_pass<LT, false, false, LFIX>(acc, nob, sc2, a.L, g, a.mask_thr, false);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < LT; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(acc[t2][r] - mx);
                    acc[t2][r] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < LT; ++t2) {
                const bf16x4 pb = f2bf4(acc[t2]);
                const bf16x4 vf = lds_tr_read(Vs + (16 * t2 + 4 * g + (fr >> 2)) * 16 + (fr & 3) * 4);
                o = mfma16(vf, pb, o);
            }
            const float inv = valid ? 1.f / sum : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= inv;
            const bf16x4 ob = f2bf4(o);
            *(bf16x4*)(a.oh + wh * SLAB + (size_t)t * 16 + 4 * g) = ob;
            if (g == 0) a.lse[wh * Lp + t] = valid ? mx + __log2f(sum) : 0.f;
            *(bf16x4*)(As + fr * PA + 16 * hd + 4 * g) = ob;            // O tile of the wave's tokens, all heads
        }

        // ---- projection Y^T[n][token] = sum_k Wp[n][k] O[token][k], k over heads * 16 (both operands from LDS)
        f32x4 yacc[NTC];
#pragma unroll
        for (int tn = 0; tn < NTC; ++tn) yacc[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < hdp / 32; ++ks) {
            const bf16x8 of = *(const bf16x8*)(As + fr * PA + 32 * ks + 8 * g);
#pragma unroll
            for (int tn = 0; tn < NTC; ++tn) {
                const bf16x8 wf = *(const bf16x8*)(Wps + (16 * tn + fr) * PWP + 32 * ks + 8 * g);
                yacc[tn] = mfma32(wf, of, yacc[tn]);
            }
        }

        // ---- epilogue (see mlp_fwd_kernel): + proj bias, bf16 round (saved), LayerNorm1 in the accumulator layout,
        // then row layout through a per-wave LDS tile: drop-path scale, residual, scatter through the row table
        {
            float s = 0.f;
#pragma unroll
            for (int tn = 0; tn < NTC; ++tn) {
                const f32x4 bv = *(const f32x4*)(cs + 16 * tn + 4 * g);
                const bf16x4 ar = f2bf4(yacc[tn] + bv);
                *(bf16x4*)(As + fr * PA + 16 * tn + 4 * g) = ar;
#pragma unroll
                for (int e = 0; e < 4; ++e) { yacc[tn][e] = bf2f(ar[e]); s += yacc[tn][e]; }
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mu = s * (1.f / C);
            float q = 0.f;
#pragma unroll
            for (int tn = 0; tn < NTC; ++tn)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = yacc[tn][e] - mu; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            const float rs = rsqrtf(q * (1.f / C) + a.eps);
            if (g == 0) {
                a.mean[(size_t)w * Lp + t] = mu;
                a.rstd[(size_t)w * Lp + t] = rs;
                St[2 * fr] = mu;
                St[2 * fr + 1] = rs;
            }
        }
        __syncthreads();
        {
            constexpr int UNITS = 16 * (C / 8);
#pragma unroll
            for (int p = 0; p < (UNITS + 63) / 64; ++p) {
                const int u = lane + 64 * p, row = u / (C / 8), c8 = u % (C / 8);
                if (UNITS % 64 == 0 || u < UNITS) {
                    const size_t m = (size_t)w * Lp + 16 * tw + row;
                    const u32x4 av = *(const u32x4*)(As + row * PA + 8 * c8);
                    *(u32x4*)(a.a1 + m * C + 8 * c8) = av;
                    const int dst = a.rowidx[m];
                    if (dst >= 0) {
                        const float mu_r = St[2 * row], rs_r = St[2 * row + 1];
                        const float sc = a.scale ? a.scale[dst / a.rows_per_sample] : 1.f;
                        const size_t off = (size_t)dst * C + 8 * c8;
                        float v[8];
                        unpack8(__builtin_bit_cast(uint4, av), v);
#pragma unroll
                        for (int hlf = 0; hlf < 2; ++hlf) {
                            const f32x4 gm = *(const f32x4*)(cs + C + 8 * c8 + 4 * hlf), bt = *(const f32x4*)(cs + 2 * C + 8 * c8 + 4 * hlf);
                            f32x4 o = *(const f32x4*)(a.x + off + 4 * hlf);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += sc * ((v[4 * hlf + e] - mu_r) * rs_r * gm[e] + bt[e]);
                            *(f32x4*)(a.y + off + 4 * hlf) = o;
                        }
                    }
                }
            }
        }
        if (w_next < a.Bw) load_x(w_next, xf);
        __syncthreads();
    }
}

template <int LT, int C, int LFIX>
void launch_branch(const AttnBranch& k, int grid, hipStream_t st) {
    hipLaunchKernelGGL((attn_branch_fwd_kernel<LT, C, LFIX>), dim3(grid), dim3(64 * LT), 0, st, k);
}

}  // namespace

extern "C" int swv2_attn_branch_supported(int C, int heads, int L, int head_dim) {
    return (C == 32 || C == 64 || C == 96 || C == 128) && heads >= 2 && heads % 2 == 0 && heads * 16 <= AF_MAX_HDP && head_dim > 0 &&
           head_dim <= 16 && L > 0 && L <= 176;
}

extern "C" int swv2_attn_branch_fwd(const swv2_attn_branch_args* a, void* stream) {
    SWV2_CHECK_ARG(a && a->x && a->rowidx && a->wqkv && a->bqkv && a->wproj && a->bproj && a->logit_scale && a->gamma && a->beta &&
                       a->qkvh && a->rnorm && a->oh && a->lse && a->a1 && a->mean && a->rstd && a->y, "swv2_attn_branch_fwd: null pointer");
    SWV2_CHECK_ARG(a->Bw > 0 && a->nwh > 0 && a->nww > 0 && a->Bw % (a->nwh * a->nww) == 0 && a->rows_per_sample > 0 &&
                       a->mask_thr >= 0 && a->mask_thr < a->L, "swv2_attn_branch_fwd: bad geometry");
    if (!swv2_attn_branch_supported(a->C, a->heads, a->L, a->head_dim)) {
        swv2_set_error("swv2_attn_branch_fwd: C=%d heads=%d L=%d head_dim=%d not instantiated; use the unfused sequence", a->C,
                       a->heads, a->L, a->head_dim);
        return SWV2_ERR_UNSUPPORTED;
    }
    AttnBranch k = {a->x, a->rowidx, (const uint16_t*)a->wqkv, a->bqkv, (const uint16_t*)a->wproj, a->bproj, a->logit_scale,
                    a->gamma, a->beta, a->scale, (uint16_t*)a->qkvh, a->rnorm, (uint16_t*)a->oh, a->lse, (uint16_t*)a->a1,
                    a->mean, a->rstd, a->y, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr,
                    a->rows_per_sample, a->eps};
    hipStream_t st = (hipStream_t)stream;
    const int grid = a->Bw < 512 ? a->Bw : 512;          // one workgroup per CU resident, two rounds of windows interleave
    const bool big = a->L > 64;
#define AB_CASE(CC)                                                                                   \
    case CC:                                                                                          \
        if (big) { if (a->L == 162) launch_branch<11, CC, 162>(k, grid, st); else launch_branch<11, CC, 0>(k, grid, st); } \
        else launch_branch<4, CC, 0>(k, grid, st);                                                     \
        break;
    switch (a->C) { AB_CASE(32) AB_CASE(64) AB_CASE(96) AB_CASE(128) }
#undef AB_CASE
    SWV2_CHECK_LAUNCH("swv2_attn_branch_fwd");
    return SWV2_OK;
}
