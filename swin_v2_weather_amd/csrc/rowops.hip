// Row-wise (HBM-bound) kernels of the SwinV2 hot path: fused LayerNorm + drop-path + residual (+ window reverse /
// un-roll scatter), its backward, weight preparation, batch sums, the geometric l2 loss, fused Adam.
// All of them stream rows with 16-byte accesses; G lanes cooperate on one row (G = 16/32/64 by channel count).
#include "common.h"

namespace {

constexpr int LN_BLOCK = 256;

__device__ __forceinline__ void unpack8f(uint4 c, float* v) {
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8f(const float* v) {
    uint4 r;
    r.x = f2bf2(v[0], v[1]); r.y = f2bf2(v[2], v[3]); r.z = f2bf2(v[4], v[5]); r.w = f2bf2(v[6], v[7]);
    return r;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_allsum<G>(v); }       // (vector ALU only: common.h)

// ------------------------------------------------------------------------------------------------
// y[dst] = res[rsrc] + scale[b] * (LN(a[m]) * gamma + beta)
//   a     : [M][C] bf16 branch output (proj or fc2 or patch conv), logical rows m
//   dst   : rowidx ? rowidx[m] : m   (window reverse + un-roll folded into the scatter; <0 = padded row, skipped)
//   res   : fp32 rows; row = res_mod ? dst % res_mod : dst  (res_mod = T broadcasts pos_embed over the batch)
//   scale : per-sample drop-path factor, b = dst / rows_per_sample ; null = 1
// saves mean / rstd per logical row for the backward.  CH = chunks (of 8 channels) per lane.
// reference: x + drop_path(norm(branch(x)))  swinv2_global.py:490,496 ; PatchEmbed norm + pos_embed :545,780
// ------------------------------------------------------------------------------------------------
template <int G, int CH>
__global__ __launch_bounds__(LN_BLOCK) void ln_residual_fwd_kernel(
    const uint16_t* __restrict__ a, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const int32_t* __restrict__ rowidx,
    float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int C, int res_mod,
    int rows_per_sample, float eps) {
    const int gl = threadIdx.x % G;
    const int rows_per_block = LN_BLOCK / G;
    for (long m = (long)blockIdx.x * rows_per_block + threadIdx.x / G; m < M; m += (long)gridDim.x * rows_per_block) {
        long dst = m;
        if (rowidx) dst = rowidx[m];
        float v[CH][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c0 = (gl + i * G) * 8;
            if (c0 < C) {
                unpack8f(*(const uint4*)(a + m * C + c0), v[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[i][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
            }
        }
        const float mu = group_sum<G>(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if ((gl + i * G) * 8 < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q = fmaf(d, d, q); }
            }
        const float rs = rsqrtf(group_sum<G>(q) / C + eps);
        if (gl == 0) { mean[m] = mu; rstd[m] = rs; }
        if (dst < 0) continue;
        const float sc = scale ? scale[(int)dst / rows_per_sample] : 1.f;
        const long rrow = res_mod ? (long)((int)dst % res_mod) : dst;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c0 = (gl + i * G) * 8;
            if (c0 >= C) continue;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const f32x4 gm = *(const f32x4*)(gamma + c0 + 4 * hlf), bt = *(const f32x4*)(beta + c0 + 4 * hlf);
                f32x4 r = res ? *(const f32x4*)(res + rrow * C + c0 + 4 * hlf) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] += sc * ((v[i][4 * hlf + e] - mu) * rs * gm[e] + bt[e]);
                *(f32x4*)(y + dst * C + c0 + 4 * hlf) = r;
            }
        }
    }
}

// backward of the above w.r.t. the branch output a, gamma, beta.  dy is the gradient of y (fp32, destination rows);
// the gradient w.r.t. `res` is dy itself (the caller owns that accumulation).
//   da[m] = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = scale * dy[dst] * gamma   (zeros for padded rows)
//   dgamma += sum_rows scale * dy * xhat ; dbeta += sum_rows scale * dy      (fp32 atomics, one per column per block)
template <int G, int CH>
__global__ __launch_bounds__(LN_BLOCK) void ln_residual_bwd_kernel(
    const uint16_t* __restrict__ a, const float* __restrict__ dy, const float* __restrict__ gamma,
    const float* __restrict__ scale, const int32_t* __restrict__ rowidx, const float* __restrict__ mean,
    const float* __restrict__ rstd, uint16_t* __restrict__ da, float* __restrict__ ws, int M, int C,
    int rows_per_sample) {
    const int gl = threadIdx.x % G;
    const int rows_per_block = LN_BLOCK / G;
    float dg[CH][8], dbt[CH][8];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; dbt[i][e] = 0.f; }

    for (long m = (long)blockIdx.x * rows_per_block + threadIdx.x / G; m < M; m += (long)gridDim.x * rows_per_block) {
        long dst = m;
        if (rowidx) dst = rowidx[m];
        if (dst < 0) {                               // padded row: zero gradient
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c0 = (gl + i * G) * 8;
                if (c0 < C) *(uint4*)(da + m * C + c0) = make_uint4(0, 0, 0, 0);
            }
            continue;
        }
        const float mu = mean[m], rs = rstd[m];
        const float sc = scale ? scale[(int)dst / rows_per_sample] : 1.f;
        float xh[CH][8], gg[CH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c0 = (gl + i * G) * 8;
            if (c0 < C) {
                float av[8];
                unpack8f(*(const uint4*)(a + m * C + c0), av);
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const f32x4 d4 = *(const f32x4*)(dy + dst * C + c0 + 4 * hlf);
                    const f32x4 gm = *(const f32x4*)(gamma + c0 + 4 * hlf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * hlf + e;
                        const float d = sc * d4[e];
                        xh[i][k] = (av[k] - mu) * rs;
                        gg[i][k] = d * gm[e];
                        dg[i][k] = fmaf(d, xh[i][k], dg[i][k]);
                        dbt[i][k] += d;
                        s1 += gg[i][k];
                        s2 = fmaf(gg[i][k], xh[i][k], s2);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[i][e] = 0.f; gg[i][e] = 0.f; }
            }
        }
        s1 = group_sum<G>(s1) / C;
        s2 = group_sum<G>(s2) / C;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c0 = (gl + i * G) * 8;
            if (c0 >= C) continue;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (gg[i][e] - s1 - xh[i][e] * s2);
            *(uint4*)(da + m * C + c0) = pack8f(o);
        }
    }
    // block reduction of dgamma / dbeta: rows_per_block partial sums per column
    __shared__ float sg[LN_BLOCK * 8], sb[LN_BLOCK * 8];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c0 = (gl + i * G) * 8;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg[threadIdx.x * 8 + e] = dg[i][e]; sb[threadIdx.x * 8 + e] = dbt[i][e]; }
        __syncthreads();
        if (threadIdx.x < G && c0 < C) {             // first row-group's lanes sum over the row groups
            float tg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < rows_per_block; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    tg[e] += sg[(r * G + gl) * 8 + e];
                    tb[e] += sb[(r * G + gl) * 8 + e];
                }
            // per-block partial sums (plain stores); ln_partials_reduce_kernel folds them: thousands of same-address
            // float atomics serialise at the memory side (measured: 7x the kernel's streaming time)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ws[((size_t)blockIdx.x * 2 + 0) * C + c0 + e] = tg[e];
                ws[((size_t)blockIdx.x * 2 + 1) * C + c0 + e] = tb[e];
            }
        }
    }
}

// dgamma[j] += sum_b ws[b][0][j], dbeta[j] += sum_b ws[b][1][j] : 16 columns x 64 slices of blocks per workgroup (a
// launch of 2 C / 16 workgroups; the per-thread chain over ~1000 partial rows is what this tiny kernel's time is, so it is
// spread over 64 slices: 11 -> ~4 us, 25 launches per step)
constexpr int LNR_COLS = 16, LNR_SLICES = 64;
__global__ __launch_bounds__(LNR_COLS * LNR_SLICES) void ln_partials_reduce_kernel(const float* __restrict__ ws,
                                                                                   float* __restrict__ dgamma,
                                                                                   float* __restrict__ dbeta, int nblocks,
                                                                                   int C, const float* __restrict__ ws2,
                                                                                   float* __restrict__ dgamma2,
                                                                                   float* __restrict__ dbeta2, int nblocks2) {
    __shared__ float part[LNR_SLICES][LNR_COLS];
    if (blockIdx.y == 1) { ws = ws2; dgamma = dgamma2; dbeta = dbeta2; nblocks = nblocks2; }      // second set (grid.y = 2)
    const int col = threadIdx.x % LNR_COLS, sl = threadIdx.x / LNR_COLS;
    const int j = blockIdx.x * LNR_COLS + col;
    float s = 0.f;
    if (j < 2 * C) {
#pragma unroll 4
        for (int b = sl; b < nblocks; b += LNR_SLICES) s += ws[(size_t)b * 2 * C + j];
    }
    part[sl][col] = s;
    __syncthreads();
    for (int h = LNR_SLICES / 2; h > 0; h >>= 1) {            // fixed-order tree: deterministic
        if (sl < h) part[sl][col] += part[sl + h][col];
        __syncthreads();
    }
    if (sl == 0 && j < 2 * C) {
        if (j < C) dgamma[j] += part[0][col]; else dbeta[j - C] += part[0][col];
    }
}

// out_bf16[i][j] = W'[rmap ? rmap[i] : i][cmap ? cmap[j] : j]  (0 where a map entry is < 0), W' = transpose ? w^T : w
__global__ void prep_weight_kernel(const float* __restrict__ w, int rows, int cols, int transpose,
                                   const int32_t* __restrict__ rmap, int out_rows, const int32_t* __restrict__ cmap,
                                   int out_cols, uint16_t* __restrict__ out) {
    const long n = (long)out_rows * out_cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int i = idx / out_cols, j = idx - (long)i * out_cols;
        const int r = rmap ? rmap[i] : i, c = cmap ? cmap[j] : j;
        float v = 0.f;
        if (r >= 0 && c >= 0) v = transpose ? w[(long)c * cols + r] : w[(long)r * cols + c];
        out[idx] = f2bf(v);
    }
}

// The same for MANY parameters in one launch (every optimizer step changes every weight: ~100 prepared copies for the
// depth-12 model; one workgroup per chunk of PREP_CHUNK output elements, tables in device memory as for adam_multi_kernel).
// out_f32 items keep fp32 (the head-padded qkv bias).
constexpr int PREP_CHUNK = 4096;
// transposed copies of matrices with both dimensions >= 64 are cut into 64 x 64 output tiles instead of linear chunks: the source
// is read along its rows (= down the output's columns) and turned in LDS.  Read element by element in output order every lane
// touches its own line: the embed-768 model's 110 M transposed elements took 2.9 ms per step.
static inline bool prep_tiled(int out_rows, int out_cols, int transpose) { return transpose && out_rows >= 64 && out_cols >= 64; }
__global__ __launch_bounds__(256) void prep_multi_kernel(const swv2_prep_item* __restrict__ items, const int2* __restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const swv2_prep_item it = items[c.x];
    if (it.transpose && it.out_rows >= 64 && it.out_cols >= 64) {
        __shared__ float t[64][65];
        const int tj_n = (it.out_cols + 63) >> 6, i0 = (c.y / tj_n) * 64, j0 = (c.y - (c.y / tj_n) * tj_n) * 64;
        const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
        {
            const int i = i0 + lane, r = i < it.out_rows ? (it.row_map ? it.row_map[i] : i) : -1;       // source column
#pragma unroll 4
            for (int jj = grp; jj < 64; jj += 4) {
                const int j = j0 + jj, cc = j < it.out_cols ? (it.col_map ? it.col_map[j] : j) : -1;   // source row
                t[jj][lane] = (r >= 0 && cc >= 0) ? it.w[(long)cc * it.cols + r] : 0.f;
            }
        }
        __syncthreads();
        const int j = j0 + lane;
        if (j < it.out_cols) {
#pragma unroll 4
            for (int ii = grp; ii < 64; ii += 4) {
                const int i = i0 + ii;
                if (i >= it.out_rows) break;
                const float v = t[lane][ii];
                if (it.out_f32) ((float*)it.out)[(long)i * it.out_cols + j] = v; else ((uint16_t*)it.out)[(long)i * it.out_cols + j] = f2bf(v);
            }
        }
        return;
    }
    const long n = (long)it.out_rows * it.out_cols, lo = (long)c.y * PREP_CHUNK, hi = min(n, lo + PREP_CHUNK);
    for (long idx = lo + threadIdx.x; idx < hi; idx += 256) {
        const int i = idx / it.out_cols, j = idx - (long)i * it.out_cols;
        const int r = it.row_map ? it.row_map[i] : i, cc = it.col_map ? it.col_map[j] : j;
        float v = 0.f;
        if (r >= 0 && cc >= 0) v = it.transpose ? it.w[(long)cc * it.cols + r] : it.w[(long)r * it.cols + cc];
        if (it.out_f32) ((float*)it.out)[idx] = v; else ((uint16_t*)it.out)[idx] = f2bf(v);
    }
}

// out[i] (+)= sum_b in[b][i]
__global__ void batch_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int B, long n, int accumulate) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        f32x4 s = accumulate ? *(const f32x4*)(out + i) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) s += *(const f32x4*)(in + (long)b * n + i);
        *(f32x4*)(out + i) = s;
    }
}

// per-row mean / rstd of the 2x2-merged rows (PatchMerging LayerNorm(4C), swinv2_global.py:521)
__global__ void merge_stats_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd,
                                   int B, int H, int W, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int h2 = H >> 1, w2 = W >> 1;
    const long M = (long)B * h2 * w2;
    for (long m = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * (blockDim.x >> 6)) {
        const int b = m / (h2 * w2), ij = m - (long)b * h2 * w2, i = ij / w2, j = ij - i * w2;
        float s = 0.f, q = 0.f;
        for (int k = lane; k < 4 * C; k += 64) {
            const int blk = k / C, c = k - blk * C, wp = blk >> 1, hp = blk & 1;
            const float v = x[(((long)b * H + 2 * i + hp) * W + 2 * j + wp) * C + c];
            s += v;
            q = fmaf(v, v, q);
        }
        s = wave_sum(s);
        q = wave_sum(q);
        const float mu = s / (4 * C);
        const float var = fmaxf(q / (4 * C) - mu * mu, 0.f);
        if (lane == 0) { mean[m] = mu; rstd[m] = rsqrtf(var + eps); }
    }
}

// backward of the PatchMerging LayerNorm + gather: dn [M][4C] bf16 (grad of the normalised rows) -> dx [B][H][W][C]
// fp32 (every x element belongs to exactly one merged row), dgamma / dbeta [4C] via atomics.
__global__ void merge_ln_bwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ dn,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int B, int H, int W, int C) {
    const int lane = threadIdx.x & 63;
    const int h2 = H >> 1, w2 = W >> 1, K = 4 * C;
    const long M = (long)B * h2 * w2;
    for (long m = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * (blockDim.x >> 6)) {
        const int b = m / (h2 * w2), ij = m - (long)b * h2 * w2, i = ij / w2, j = ij - i * w2;
        const float mu = mean[m], rs = rstd[m];
        float s1 = 0.f, s2 = 0.f;
        for (int k = lane; k < K; k += 64) {
            const int blk = k / C, c = k - blk * C, wp = blk >> 1, hp = blk & 1;
            const float xh = (x[(((long)b * H + 2 * i + hp) * W + 2 * j + wp) * C + c] - mu) * rs;
            const float d = bf2f(dn[m * K + k]);
            const float gg = d * gamma[k];
            s1 += gg;
            s2 = fmaf(gg, xh, s2);
            atomicAdd(dgamma + k, d * xh);
            atomicAdd(dbeta + k, d);
        }
        s1 = wave_sum(s1) / K;
        s2 = wave_sum(s2) / K;
        for (int k = lane; k < K; k += 64) {
            const int blk = k / C, c = k - blk * C, wp = blk >> 1, hp = blk & 1;
            const long xi = (((long)b * H + 2 * i + hp) * W + 2 * j + wp) * C + c;
            const float xh = (x[xi] - mu) * rs;
            const float gg = bf2f(dn[m * K + k]) * gamma[k];
            dx[xi] = rs * (gg - s1 - xh * s2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// geometric l2 loss (losses.py:188-232, grids.py:115-117): per (b, c) quadrature sums in one pass
//   sums[(b*C + c)*2 + 0] = sum_{h,w} q[h] (prd - tar)^2 ; [..+1] = sum_{h,w} q[h] tar^2
// and the backward  dprd = coef[b][c] * q[h] * (prd - tar)   (coef from the host: chain rule of the chosen l2 variant)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ prd, const float* __restrict__ tar,
                                                        const float* __restrict__ qw, float* __restrict__ sums,
                                                        int H, int W, int slices) {
    const int bc = blockIdx.x / slices, sl = blockIdx.x - bc * slices;
    const long plane = (long)H * W;
    const long lo = plane * sl / slices / 4 * 4, hi = (sl + 1 == slices) ? plane : plane * (sl + 1) / slices / 4 * 4;
    const float* p = prd + bc * plane;
    const float* t = tar + bc * plane;
    float s0 = 0.f, s1 = 0.f;
    // four independent 16-byte loads of each operand in flight per thread (one pair per iteration ran at 4.2 TB/s)
    long i = lo + threadIdx.x * 4;
    for (; i + 3 * 1024 < hi; i += 4 * 1024) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(p + i + u * 1024); b[u] = *(const f32x4*)(t + i + u * 1024); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float q = qw[(i + u * 1024) / W];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = a[u][e] - b[u][e];
                s0 = fmaf(q * d, d, s0);
                s1 = fmaf(q * b[u][e], b[u][e], s1);
            }
        }
    }
    for (; i < hi; i += 256 * 4) {
        const f32x4 a = *(const f32x4*)(p + i), b = *(const f32x4*)(t + i);
        const float q = qw[i / W];                 // W % 4 == 0: the 4 elements share a latitude row
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = a[e] - b[e];
            s0 = fmaf(q * d, d, s0);
            s1 = fmaf(q * b[e], b[e], s1);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    __shared__ float r0[4], r1[4];
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = s0; r1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums + bc * 2, r0[0] + r0[1] + r0[2] + r0[3]);
        atomicAdd(sums + bc * 2 + 1, r1[0] + r1[1] + r1[2] + r1[3]);
    }
}

// loss = sum_{b,c} chw[c] f(S0 / S1)  (relative; S0 alone when absolute; f = sqrt unless squared) and the coefficient
// the backward kernel multiplies in: coef[bc] = 2 d loss / d S0[bc].  One workgroup; replaces the half dozen elementwise /
// reduction launches (and as many again in their autograd) that torch spent on this [B, C] tensor (losses.py:188-232).
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ sums, int layers, const float* __restrict__ chw,
                                                            int BC, int C, int absolute, int squared,
                                                            float* __restrict__ loss, float* __restrict__ coef) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < BC; i += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int l = 0; l < layers; ++l) { s0 += sums[((size_t)l * BC + i) * 2]; s1 += sums[((size_t)l * BC + i) * 2 + 1]; }
        const float w = chw[i % C];
        const float den = absolute ? 1.f : s1;
        const float r = s0 / den;
        float f, df;                                    // f(r), f'(r)
        if (squared) { f = r; df = 1.f; } else { f = sqrtf(r); df = 0.5f / f; }
        acc += w * f;
        coef[i] = 2.f * w * df / den;
    }
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (red[0] + red[1]) + (red[2] + red[3]);
}

// per-group partial sums of the head epilogue (gemm.hip, Epi<E_UNPATCH_LOSS>: part[g][slot][Cout][2]) -> sums[j][b][..].  grid =
// (B, SWV2_LOSS_PART_SLICES), 4 sub-slices of 256 threads; thread v < 2 Cout owns one (channel, sum) value and adds the groups
// of its sample, slice and sub-slice in ascending order (coalesced: the 2 Cout values of a group slot are contiguous); the
// sub-slices are combined through LDS in a fixed order.  Slot 1 of the group in front of the sample's first group holds the
// rows of this sample that sit in a group starting in the previous one.
__global__ __launch_bounds__(1024) void loss_part_reduce_kernel(const float* __restrict__ part, int M, int T, int Cout, int Ct,
                                                                int coff, int B, float* __restrict__ sums) {
    const int b = blockIdx.x, j = blockIdx.y, nv = 2 * Cout;
    constexpr int GR = SWV2_LOSS_GROUP_ROWS;
    const int ngroups = (M + GR - 1) / GR;
    const int g_lo = (b * T + GR - 1) / GR, g_hi = min(((b + 1) * T + GR - 1) / GR, ngroups);     // groups whose first row lies in sample b
    constexpr int S = SWV2_LOSS_PART_SLICES;
    const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
    __shared__ float red[3][256];
    const int g0 = g_lo + ((j - g_lo) % S + S) % S + sub * S;          // first g >= g_lo with g % S == j, then the sub's offset
    const size_t gs = (size_t)2 * nv;                                  // floats per group
    for (int v0 = 0; v0 < nv; v0 += 256) {
        const int v = v0 + t;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (v < nv) {
            int g = g0;
            for (; g + 12 * S < g_hi; g += 16 * S) {
                a0 += part[(size_t)g * gs + v];
                a1 += part[(size_t)(g + 4 * S) * gs + v];
                a2 += part[(size_t)(g + 8 * S) * gs + v];
                a3 += part[(size_t)(g + 12 * S) * gs + v];
            }
            for (; g < g_hi; g += 4 * S) a0 += part[(size_t)g * gs + v];
            if (sub == 0 && b > 0 && g_lo > 0 && (g_lo - 1) % S == j) a1 += part[(size_t)(g_lo - 1) * gs + nv + v];
        }
        const float tot = (a0 + a1) + (a2 + a3);
        if (sub) red[sub - 1][t] = tot;
        __syncthreads();
        if (!sub && v < nv) sums[(((size_t)j * B + b) * Ct + coff) * 2 + v] = ((tot + red[0][t]) + red[1][t]) + red[2][t];
        __syncthreads();
    }
}

// grid = (plane / 4096, B * C): a workgroup handles 4 x 256 float4 units of one (b, c) plane, all eight loads in flight
// before the first store; 32-bit index arithmetic (the first version did two 64-bit divisions per float4)
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ prd, const float* __restrict__ tar,
                                                        const float* __restrict__ qw, const float* __restrict__ coef,
                                                        float* __restrict__ dprd, int H, int W) {
    const int plane = H * W, bc = blockIdx.y;
    const size_t base = (size_t)bc * plane;
    const float cf = coef[bc];
    f32x4 a[4], b[4];
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        idx[u] = ((blockIdx.x * 4 + u) * 256 + threadIdx.x) * 4;
        const int j = min(idx[u], plane - 4);                        // clamped: unconditional loads
        a[u] = *(const f32x4*)(prd + base + j);
        b[u] = *(const f32x4*)(tar + base + j);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (idx[u] < plane) {
            const float c = cf * qw[idx[u] / W];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = c * (a[u][e] - b[u][e]);
            *(f32x4*)(dprd + base + idx[u]) = o;
        }
    }
}

// fused Adam over one flat fp32 buffer (torch.optim.Adam semantics, train.py:176: betas (0.9, 0.95), eps 1e-8,
// no weight decay, bias correction).  step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t); inv_scale un-scales grads.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float b1, float b2, float eps, float step_size,
                            float bc2_sqrt, float inv_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * inv_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// torch.optim.Adam over MANY tensors in one launch (torch's fused multi-tensor path needs 5 launches of ~45 us for the 160
// parameters of the depth-12 model, 1.3 TB/s; the update is 28 bytes per element of pure streaming).  items / chunks live
// in device memory: chunk c = elements [start, start + ADAM_CHUNK) of tensor chunks[c].item; one workgroup per chunk.
constexpr int ADAM_CHUNK = 4096;
__global__ __launch_bounds__(256) void adam_multi_kernel(const swv2_adam_item* __restrict__ items, const int2* __restrict__ chunks,
                                                         float b1, float b2, float eps, float step_size, float bc2_sqrt,
                                                         float inv_scale) {
    const int2 c = chunks[blockIdx.x];
    const swv2_adam_item it = items[c.x];
    const long lo = (long)c.y * ADAM_CHUNK, hi = min(it.n, lo + ADAM_CHUNK);
    float* __restrict__ p = it.p;
    const float* __restrict__ g = it.g;
    float* __restrict__ m = it.m;
    float* __restrict__ v = it.v;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (vec) {
        for (long i = lo + 4 * threadIdx.x; i + 3 < hi; i += 4 * 256) {
            const f32x4 gi = *(const f32x4*)(g + i) * inv_scale;
            f32x4 mi = *(const f32x4*)(m + i), vi = *(const f32x4*)(v + i), pi = *(const f32x4*)(p + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mi[e] = b1 * mi[e] + (1.f - b1) * gi[e];
                vi[e] = b2 * vi[e] + (1.f - b2) * gi[e] * gi[e];
                pi[e] -= step_size * mi[e] / (sqrtf(vi[e]) / bc2_sqrt + eps);
            }
            *(f32x4*)(m + i) = mi; *(f32x4*)(v + i) = vi; *(f32x4*)(p + i) = pi;
        }
    }
    // scalar path: everything (unaligned tensors) or the tail of < 4 elements
    const long s0 = vec ? lo + ((hi - lo) & ~3L) : lo;
    for (long i = s0 + threadIdx.x; i < hi; i += 256) {
        const float gi = g[i] * inv_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

template <int G, int CH>
void launch_ln_fwd(const swv2_ln_args* a, hipStream_t st) {
    const int rows_per_block = LN_BLOCK / G;
    const int grid = min(cdiv(a->M, rows_per_block), 256 * 16);
    hipLaunchKernelGGL((ln_residual_fwd_kernel<G, CH>), dim3(grid), dim3(LN_BLOCK), 0, st, (const uint16_t*)a->a, a->res,
                       a->gamma, a->beta, a->scale, a->rowidx, a->y, a->mean, a->rstd, a->M, a->C, a->res_mod,
                       a->rows_per_sample, a->eps);
}
template <int G, int CH>
void launch_ln_bwd(const swv2_ln_args* a, hipStream_t st) {
    const int rows_per_block = LN_BLOCK / G;
    const int grid = min(cdiv(a->M, rows_per_block), SWV2_LN_BWD_MAX_BLOCKS);
    hipLaunchKernelGGL((ln_residual_bwd_kernel<G, CH>), dim3(grid), dim3(LN_BLOCK), 0, st, (const uint16_t*)a->a, a->dy,
                       a->gamma, a->scale, a->rowidx, a->mean, a->rstd, (uint16_t*)a->da, a->ws, a->M, a->C,
                       a->rows_per_sample);
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3(cdiv(2 * a->C, LNR_COLS)), dim3(LNR_COLS * LNR_SLICES), 0, st, a->ws,
                       a->dgamma, a->dbeta, grid, a->C, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0);
}

int ln_check(const swv2_ln_args* a, bool bwd) {
    SWV2_CHECK_ARG(a && a->a && a->gamma && a->mean && a->rstd, "ln_residual: null pointer");
    SWV2_CHECK_ARG(a->M > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 1024, "ln_residual: C=%d must be a multiple of 8, <= 1024", a->C);
    SWV2_CHECK_ARG(a->rows_per_sample > 0, "ln_residual: rows_per_sample must be positive");
    if (bwd) SWV2_CHECK_ARG(a->dy && a->da && a->dgamma && a->dbeta && a->ws, "ln_residual_bwd: null gradient / workspace pointer");
    else SWV2_CHECK_ARG(a->y && a->beta, "ln_residual_fwd: null output pointer");
    return SWV2_OK;
}

}  // namespace

// shared with mlp.hip (fused MLP backward): dgamma[j] += sum_b ws[b][0][j], dbeta[j] += sum_b ws[b][1][j]
void swv2_launch_ln_partials_reduce2(const float* ws1, float* dg1, float* db1, int n1, const float* ws2, float* dg2, float* db2,
                                     int n2, int C, hipStream_t st) {
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3(cdiv(2 * C, LNR_COLS), 2), dim3(LNR_COLS * LNR_SLICES), 0, st, ws1, dg1, db1,
                       n1, C, ws2, dg2, db2, n2);
}

void swv2_launch_ln_partials_reduce(const float* ws, float* dgamma, float* dbeta, int nblocks, int C, hipStream_t st) {
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3(cdiv(2 * C, LNR_COLS)), dim3(LNR_COLS * LNR_SLICES), 0, st, ws, dgamma,
                       dbeta, nblocks, C, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0);
}

#define LN_DISPATCH(FN)                                              \
    const int chunks = a->C / 8;                                     \
    if (chunks <= 16) FN<16, 1>(a, st);                              \
    else if (chunks <= 32) FN<32, 1>(a, st);                         \
    else if (chunks <= 64) FN<64, 1>(a, st);                         \
    else FN<64, 2>(a, st);

extern "C" int swv2_ln_residual_fwd(const swv2_ln_args* a, void* stream) {
    int rc = ln_check(a, false);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(launch_ln_fwd)
    SWV2_CHECK_LAUNCH("swv2_ln_residual_fwd");
    return SWV2_OK;
}

extern "C" int swv2_ln_residual_bwd(const swv2_ln_args* a, void* stream) {
    int rc = ln_check(a, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(launch_ln_bwd)
    SWV2_CHECK_LAUNCH("swv2_ln_residual_bwd");
    return SWV2_OK;
}

// ------------------------------------------------------------------------------------------------
// second half of the q / k normalisation for 128-wide padded heads (the GEMM epilogue leaves the squared norms in rnorm
// and the un-normalised values in qkvh, see heads_item_wide in gemm.hip): rn = 1 / max(|.|, 1e-12) (F.normalize's eps,
// swinv2_global.py:300-304), q, k rows rescaled in place, rnorm <- rn (0 on padded rows).  16 lanes per 128-wide row.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void qk_normalize_kernel(uint16_t* __restrict__ qkvh, float* __restrict__ rnorm, long rows,
                                                           int h, int Lp, int L, int DP) {
    const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);        // row index over [Bw][h][2][Lp]
    const int c = threadIdx.x & 15;                                   // 16-byte chunk of the row (DP / 8 of them: 12 or 16)
    if (i >= rows) return;
    const int t = (int)(i % Lp);
    const long bh2 = i / Lp;
    const int part = (int)(bh2 & 1);
    const long bh = bh2 >> 1;
    const float ss = rnorm[i];
    const float rn = (t < L) ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
    if (c * 8 < DP) {
        uint16_t* row = qkvh + ((bh * 3 + part) * Lp + t) * DP + c * 8;
        float v[8];
        unpack8f(*(const uint4*)row, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= rn;
        *(uint4*)row = pack8f(v);
    }
    if (c == 0) rnorm[i] = rn;
}
}  // namespace

extern "C" int swv2_qk_normalize(void* qkvh, float* rnorm, int Bw, int heads, int Lp, int L, int DP, void* stream) {
    SWV2_CHECK_ARG(qkvh && rnorm && Bw > 0 && heads > 0 && Lp > 0 && L > 0 && L <= Lp, "swv2_qk_normalize: bad argument");
    SWV2_CHECK_ARG(DP == 96 || DP == 128, "swv2_qk_normalize: only the 96- and 128-wide head layouts need this pass (DP=%d)", DP);
    const long rows = (long)Bw * heads * 2 * Lp;
    hipLaunchKernelGGL(qk_normalize_kernel, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkvh,
                       rnorm, rows, heads, Lp, L, DP);
    SWV2_CHECK_LAUNCH("swv2_qk_normalize");
    return SWV2_OK;
}

extern "C" int swv2_prep_weight(const float* w, int rows, int cols, int transpose, const int32_t* row_map, int out_rows,
                                const int32_t* col_map, int out_cols, void* out_bf16, void* stream) {
    SWV2_CHECK_ARG(w && out_bf16 && rows > 0 && cols > 0 && out_rows > 0 && out_cols > 0, "prep_weight: bad argument");
    const long n = (long)out_rows * out_cols;
    hipLaunchKernelGGL(prep_weight_kernel, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, w, rows, cols,
                       transpose, row_map, out_rows, col_map, out_cols, (uint16_t*)out_bf16);
    SWV2_CHECK_LAUNCH("swv2_prep_weight");
    return SWV2_OK;
}

extern "C" int swv2_prep_chunk(void) { return PREP_CHUNK; }

extern "C" int swv2_prep_item_chunks(int out_rows, int out_cols, int transpose) {
    if (out_rows <= 0 || out_cols <= 0) return 0;
    if (prep_tiled(out_rows, out_cols, transpose)) return ((out_rows + 63) / 64) * ((out_cols + 63) / 64);
    return (int)(((long)out_rows * out_cols + PREP_CHUNK - 1) / PREP_CHUNK);
}

extern "C" int swv2_prep_multi(const swv2_prep_item* items_dev, const int* chunks_dev, int n_chunks, void* stream) {
    SWV2_CHECK_ARG(items_dev && chunks_dev && n_chunks > 0, "prep_multi: bad argument");
    hipLaunchKernelGGL(prep_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, items_dev, (const int2*)chunks_dev);
    SWV2_CHECK_LAUNCH("swv2_prep_multi");
    return SWV2_OK;
}

extern "C" int swv2_batch_sum(const float* in, float* out, int B, long n, int accumulate, void* stream) {
    SWV2_CHECK_ARG(in && out && B > 0 && n > 0 && n % 4 == 0, "batch_sum: bad argument (n must be a multiple of 4)");
    hipLaunchKernelGGL(batch_sum_kernel, dim3(min(cdiv(n / 4, 256), 8192)), dim3(256), 0, (hipStream_t)stream, in, out, B, n,
                       accumulate);
    SWV2_CHECK_LAUNCH("swv2_batch_sum");
    return SWV2_OK;
}

extern "C" int swv2_merge_stats(const float* x, float* mean, float* rstd, int B, int H, int W, int C, float eps,
                                void* stream) {
    SWV2_CHECK_ARG(x && mean && rstd && B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0, "merge_stats: bad argument");
    const long M = (long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(merge_stats_kernel, dim3(min(cdiv(M, 4), 4096)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, B, H,
                       W, C, eps);
    SWV2_CHECK_LAUNCH("swv2_merge_stats");
    return SWV2_OK;
}

extern "C" int swv2_merge_ln_bwd(const float* x, const void* dn_bf16, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, float* dgamma, float* dbeta, int B, int H, int W, int C,
                                 void* stream) {
    SWV2_CHECK_ARG(x && dn_bf16 && gamma && mean && rstd && dx && dgamma && dbeta, "merge_ln_bwd: null pointer");
    const long M = (long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(merge_ln_bwd_kernel, dim3(min(cdiv(M, 4), 4096)), dim3(256), 0, (hipStream_t)stream, x,
                       (const uint16_t*)dn_bf16, gamma, mean, rstd, dx, dgamma, dbeta, B, H, W, C);
    SWV2_CHECK_LAUNCH("swv2_merge_ln_bwd");
    return SWV2_OK;
}

extern "C" int swv2_loss_sums(const float* prd, const float* tar, const float* quad_w, float* sums, int BC, int H, int W,
                              void* stream) {
    SWV2_CHECK_ARG(prd && tar && quad_w && sums && BC > 0 && H > 0 && W > 0 && W % 4 == 0, "loss_sums: bad argument (W % 4)");
    // at most 2048 workgroups = one round at 8 per CU (7 .. 56 slices at B * C = 146 all run at 4.5 TB/s: the read stream is the bound)
    const int slices = BC >= 2048 ? 1 : 2048 / BC;
    hipLaunchKernelGGL(loss_sums_kernel, dim3(BC * slices), dim3(256), 0, (hipStream_t)stream, prd, tar, quad_w, sums, H, W,
                       slices);
    SWV2_CHECK_LAUNCH("swv2_loss_sums");
    return SWV2_OK;
}

extern "C" int swv2_loss_part_reduce(const float* part, int M, int T, int B, int Cout, int Ct, int coff, float* sums, void* stream) {
    SWV2_CHECK_ARG(part && sums && M > 0 && T >= SWV2_LOSS_GROUP_ROWS && B > 0 && M == B * T && Cout > 0 && coff >= 0 && coff + Cout <= Ct,
                   "loss_part_reduce: bad argument");
    hipLaunchKernelGGL(loss_part_reduce_kernel, dim3(B, SWV2_LOSS_PART_SLICES), dim3(1024), 0, (hipStream_t)stream, part, M, T, Cout,
                       Ct, coff, B, sums);
    SWV2_CHECK_LAUNCH("swv2_loss_part_reduce");
    return SWV2_OK;
}

extern "C" int swv2_loss_finalize(const float* sums, int layers, const float* chw, int BC, int C, int absolute, int squared, float* loss,
                                  float* coef, void* stream) {
    SWV2_CHECK_ARG(sums && chw && loss && coef && BC > 0 && C > 0 && BC % C == 0 && layers > 0, "loss_finalize: bad argument");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, layers, chw, BC, C, absolute, squared, loss,
                       coef);
    SWV2_CHECK_LAUNCH("swv2_loss_finalize");
    return SWV2_OK;
}

extern "C" int swv2_loss_grad(const float* prd, const float* tar, const float* quad_w, const float* coef, float* dprd,
                              int BC, int H, int W, void* stream) {
    SWV2_CHECK_ARG(prd && tar && quad_w && coef && dprd && BC > 0 && W % 4 == 0, "loss_grad: bad argument (W % 4)");
    SWV2_CHECK_ARG((long)H * W < (1L << 30) && BC <= 65535, "loss_grad: plane or B*C too large for the 32-bit index map");
    hipLaunchKernelGGL(loss_grad_kernel, dim3(cdiv((long)H * W, 4096), BC), dim3(256), 0, (hipStream_t)stream, prd, tar, quad_w,
                       coef, dprd, H, W);
    SWV2_CHECK_LAUNCH("swv2_loss_grad");
    return SWV2_OK;
}

// The loss epilogue's weighted residual back in IMAGE layout, scaled: out[b][c][4i+p][4j+q] = coef[b][c] * resid[(b,i,j)][c*16 + p*4 + q]
// (+ add[b][c][..]) -- d loss / d prediction of a rollout step whose head carries a skip connection from an input that needs a
// gradient (swinv2_global.py:799-801 inside helpers.py:26-41).  Workgroup = 16 consecutive patches of one patch row: their residual rows
// (16 x N bf16, contiguous rows) go through LDS, then lane -> (patch, image row p) per channel so that a store instruction writes
// 256 contiguous bytes.  Reads 2 N bytes per patch once; the two-pass loss_grad_kernel reads prediction and target (8 N bytes).
template <int TOK>
__global__ __launch_bounds__(256) void resid_to_image_kernel(const uint16_t* __restrict__ resid, const float* __restrict__ coef,
                                                             const float* __restrict__ add, float* __restrict__ out, int Cout, int H,
                                                             int W, int Cs, int Cadd, int gw) {
    extern __shared__ __attribute__((aligned(16))) uint16_t rs[];
    const int N = Cout * 16, P = N + 8;                    // LDS pitch
    const int RP = SWV2_LOSS_RESID_PITCH(N);               // row pitch of the residual in memory
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * TOK, i = blockIdx.y, b = blockIdx.z;
    const int gh = H >> 2;
    const long m0 = ((long)b * gh + i) * gw + j0;
    const int ntok = min(TOK, gw - j0);
    const int cpr = N / 8;                                 // 16-byte chunks per row
    for (int u = tid; u < ntok * cpr; u += 256) {
        const int t = u / cpr, ch = u - t * cpr;
        *(uint4*)(rs + t * P + 8 * ch) = *(const uint4*)(resid + (m0 + t) * RP + 8 * ch);
    }
    __syncthreads();
    const int t = tid & (TOK - 1);
    const long plane = (long)H * W;
    for (int cp = tid / TOK; cp < Cout * 4; cp += 256 / TOK) {
        const int c = cp >> 2, p = cp & 3;
        if (t >= ntok) continue;
        const uint2 r = *(const uint2*)(rs + t * P + c * 16 + p * 4);
        const float cf = coef[b * Cout + c];
        f32x4 v = {cf * __uint_as_float(r.x << 16), cf * __uint_as_float(r.x & 0xffff0000u), cf * __uint_as_float(r.y << 16),
                   cf * __uint_as_float(r.y & 0xffff0000u)};
        const long pix = (long)(4 * i + p) * W + 4 * (j0 + t);
        if (add) v += *(const f32x4*)(add + ((long)b * Cadd + c) * plane + pix);
        *(f32x4*)(out + ((long)b * Cs + c) * plane + pix) = v;
    }
}

extern "C" int swv2_loss_resid_to_image(const void* resid, const float* coef, const float* add, float* out, int B, int Cout, int H, int W,
                                        int Cs, int Cadd, void* stream) {
    SWV2_CHECK_ARG(resid && coef && out && B > 0 && Cout > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && Cs >= Cout && (!add || Cadd >= Cout),
                   "loss_resid_to_image: bad argument");
    SWV2_CHECK_ARG(B <= 65535 && H / 4 <= 65535, "loss_resid_to_image: grid too large");
    SWV2_CHECK_ARG((((uintptr_t)resid | (uintptr_t)out | (uintptr_t)add) & 15) == 0, "loss_resid_to_image: unaligned pointer");
    constexpr int TOK = 16;
    const int gw = W / 4, gh = H / 4;
    const size_t lds = (size_t)TOK * (Cout * 16 + 8) * 2;
    SWV2_CHECK_ARG(lds <= 64 * 1024, "loss_resid_to_image: Cout too large for the staging tile");
    hipLaunchKernelGGL((resid_to_image_kernel<TOK>), dim3(cdiv(gw, TOK), gh, B), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)resid,
                       coef, add, out, Cout, H, W, Cs, Cadd, gw);
    SWV2_CHECK_LAUNCH("swv2_loss_resid_to_image");
    return SWV2_OK;
}

extern "C" int swv2_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int swv2_adam_multi(const swv2_adam_item* items_dev, const int* chunks_dev, int n_chunks, float lr, float beta1,
                               float beta2, float eps, int step, float grad_inv_scale, void* stream) {
    SWV2_CHECK_ARG(items_dev && chunks_dev && n_chunks > 0 && step > 0, "adam_multi: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, items_dev, (const int2*)chunks_dev,
                       beta1, beta2, eps, lr / bc1, sqrtf(bc2), grad_inv_scale);
    SWV2_CHECK_LAUNCH("swv2_adam_multi");
    return SWV2_OK;
}

extern "C" int swv2_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, int step, float grad_inv_scale, void* stream) {
    SWV2_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(min(cdiv(n, 256), 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1,
                       beta2, eps, lr / bc1, sqrtf(bc2), grad_inv_scale);
    SWV2_CHECK_LAUNCH("swv2_adam_step");
    return SWV2_OK;
}
