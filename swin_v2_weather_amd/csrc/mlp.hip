// Fused MLP branch of a SwinV2 block, forward (reference swinv2_global.py:492-496 with timm Mlp :381-386):
//     x2 = x1 + drop_path2 * LayerNorm2( fc2( GELU( fc1(x1) ) ) )
// as ONE kernel.  The unfused sequence (fc1 GEMM, fc2 GEMM, LN kernel) moves 660 MB per block at the benchmark shape
// (the [M][hidden] activation is written twice and read once, the fc2 output and x1 are re-read by the LN kernel); here
// the hidden activation never leaves the registers and HBM sees x1 once in, and (x2, the saved pre-activation, the saved
// fc2 output, mean / rstd) out: 300 MB.
//
// Chained-MFMA layout trick (no LDS round trip for the hidden activation): both products are computed TRANSPOSED,
//     H^T[hid][m] = W1[hid][c] X^T[c][m]      A = W1 rows (LDS),  B = X rows (registers, loaded once per wave)
//     Y^T[n][m]   = W2[n][hid] H^T[hid][m]    A = W2 rows (LDS),  B = GELU(H^T) straight from the accumulators:
// the 16x16 C tile of a wave64 MFMA holds (row 4g+r, col l&15) in lane l, which is exactly the B-operand layout
// (k = 4g+j, n = l&15) of the next product; two C tiles (32 hidden units) make one K=32 operand, with the matching
// k-permutation applied to the W2 fragment reads.
// A wave owns 16*MT rows for the whole kernel (X fragments + Y^T accumulators stay in registers); the weights stream
// through LDS in chunks of 32 hidden units (W1[32][C] and W2[C][32], double buffered), shared by the 4 waves.
#include <cstdlib>
#include "gemm_common.h"

namespace {

constexpr int MLP_MAX_HIDDEN = 2048;
constexpr int MLP_RECOMP_MAX_HIDDEN = 1024;      // recompute mode of the backward: fc1 bias in the tail of the weight-chunk region

struct MlpFwd {
    const float* x; const uint16_t* w1; const float* b1; const uint16_t* w2; const float* b2;
    const float* gamma; const float* beta; const float* scale;
    uint16_t* hpre; uint16_t* a2; float* mean; float* rstd; float* y;
    int M, hidden, rows_per_sample; float eps;
};

template <int C, int MT>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(const MlpFwd a) {
    constexpr int KS = C / 32, NT = C / 16;
    constexpr int P1 = C + 8, P2 = 40;                 // LDS row pitches (elements) of the W1 / W2 chunks
    constexpr int W1E = 32 * P1, W2E = C * P2;
    constexpr int NCHUNK = 4 * C;                      // 16-byte pieces of either weight chunk
    constexpr int SPT = (NCHUNK + 255) / 256;
    constexpr int ROWS = 64 * MT;                      // rows per workgroup
    constexpr int PX = C + 8;                          // pitch (bf16) of the staged x tile
    constexpr int PA = C + 8;                          // pitch of the per-wave epilogue tile (bf16)
    constexpr int EWAVE = 16 * PA * 2 + 16 * 2 * 4;    // per wave: bf16 tile + (mean, rstd) of its 16 rows
    constexpr int WBYTES = 2 * (W1E + W2E) * 2, XBYTES = ROWS * PX * 2, EBYTES = 4 * EWAVE;
    constexpr int SBYTES = WBYTES > XBYTES ? (WBYTES > EBYTES ? WBYTES : EBYTES) : (XBYTES > EBYTES ? XBYTES : EBYTES);
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SBYTES];
    // fc1 bias in LDS: a global load inside the chunk loop would make its s_waitcnt drain the (older, in-order) weight
    // prefetch of the next chunk as well
    // (C = 192, one row tile per wave: sized for hidden <= 1024 so that TWO workgroups fit a CU -- 81 152 bytes each; the launch falls
    // back to two row tiles per wave for larger hidden sizes)
    constexpr int B1N = (C == 192 && MT == 1) ? 1024 : MLP_MAX_HIDDEN;
    __shared__ __attribute__((aligned(16))) float b1s[B1N];
    __shared__ __attribute__((aligned(16))) float cs[3 * C];          // fc2 bias | LayerNorm gamma | beta
    __shared__ __attribute__((aligned(16))) uint16_t gtab[GT_N];      // bf16(GELU(x)) for every bf16 x in the table range
    constexpr int PHS = 72;                                           // pitch (bf16) of the wave-private hpre staging tile
    __shared__ __attribute__((aligned(16))) uint16_t hst_all[4 * 16 * MT * PHS];
    uint16_t* smem = (uint16_t*)smem_raw;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int wg_row0 = blockIdx.x * ROWS;
    const int row0 = wg_row0 + wave * (16 * MT);
    const int hid = a.hidden, nch = hid / 32;
    uint16_t* const Hst = hst_all + (threadIdx.x >> 6) * (16 * MT * PHS);

    // weight-chunk staging (global/L2 -> registers -> LDS), one chunk ahead of the MFMAs
    u32x4 s1[SPT], s2[SPT];
    auto issue = [&](int ch) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = min(tid + 256 * i, NCHUNK - 1);      // unconditional loads keep the staging in registers
            s1[i] = *(const u32x4*)(a.w1 + (size_t)(32 * ch + idx / (C / 8)) * C + 8 * (idx % (C / 8)));
            s2[i] = *(const u32x4*)(a.w2 + (size_t)(idx >> 2) * hid + 32 * ch + 8 * (idx & 3));
        }
    };
    auto commit = [&](int buf) {
        uint16_t* W1s = smem + buf * (W1E + W2E);
        uint16_t* W2s = W1s + W1E;
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = tid + 256 * i;
            if (NCHUNK % 256 == 0 || idx < NCHUNK) {
                *(u32x4*)(W1s + (idx / (C / 8)) * P1 + 8 * (idx % (C / 8))) = s1[i];
                // the 32 hidden units of a W2 row are stored in the ORDER the K = 32 operand wants them -- position
                // 8 g + 4 half + j holds hidden 16 half + 4 g + j -- so a lane's whole A fragment is one 16-byte read (it was two
                // 8-byte reads per MFMA pair: 16 of the 48 LDS instructions per chunk and wave of a kernel bound by the LDS
                // instruction rate); the 16-byte source piece q (hidden 8 q .. 8 q + 7) lands as two 8-byte halves
                const int q = idx & 3;
                uint16_t* row = W2s + (idx >> 2) * P2 + 4 * (q >> 1);
                *(u32x2*)(row + 16 * (q & 1)) = (u32x2){s2[i][0], s2[i][1]};
                *(u32x2*)(row + 16 * (q & 1) + 8) = (u32x2){s2[i][2], s2[i][3]};
            }
        }
    };
#ifdef SWV2_MLP_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define STAMP(i)
#endif
    issue(0);
    for (int i = tid; i < hid; i += 256) b1s[i] = a.b1[i];
    for (int i = tid; i < C; i += 256) { cs[i] = a.b2[i]; cs[C + i] = a.gamma[i]; cs[2 * C + i] = a.beta[i]; }
    for (int i = tid; i < GT_N; i += 256) gtab[i] = f2bf(gelu_f(bf2f(gelu_tab_arg(i))));

    // x tile -> LDS as bf16 with coalesced 16-byte loads (a lane-per-row fragment load would touch 64 lines per
    // instruction), then each wave picks up its B fragments: lane (m = fr, g) holds c = 32 ks + 8 g .. + 7
    {
        constexpr int UNITS = ROWS * (C / 4);             // f32x4 units of the tile
#pragma unroll
        for (int i = 0; i < UNITS / 256; ++i) {
            const int u = tid + 256 * i, row = u / (C / 4), c4 = u % (C / 4);
            // rows past M repeat row M - 1: every later load / store of such a row is then an unconditional duplicate of
            // that row's (identical) values -- no exec-masked memory ops, so the compiler can count them (s_waitcnt)
            const f32x4 v = *(const f32x4*)(a.x + (size_t)min(wg_row0 + row, a.M - 1) * C + 4 * c4);
            *(bf16x4*)(smem + row * PX + 4 * c4) = f2bf4(v);
        }
    }
    __syncthreads();
    bf16x8 xf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[mt][ks] = *(const bf16x8*)(smem + (wave * 16 * MT + 16 * mt + fr) * PX + 32 * ks + 8 * g);
    __syncthreads();

    f32x4 yacc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) yacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    commit(0);
    __syncthreads();
    STAMP(0);
    auto chunk = [&](const int ch) {
        if (ch + 1 < nch) issue(ch + 1);
        const uint16_t* W1s = smem + (ch & 1) * (W1E + W2E);
        const uint16_t* W2s = W1s + W1E;
        // ---- H^T chunk (32 hidden x 16 MT rows), bias, bf16 round (saved), GELU -> B operand of the second product
        f32x4 hacc[2][MT];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const f32x4 bias = *(const f32x4*)(b1s + 32 * ch + 16 * ht + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = bias;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(W1s + (16 * ht + fr) * P1 + 32 * ks + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = mfma32(wf, xf[mt][ks], hacc[ht][mt]);
            }
        }
        STAMP(1);
        bf16x4 hb[MT][2], hrs[MT][2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x4 hr = f2bf4(hacc[ht][mt]);             // the pre-activation as the backward will read it
                hrs[mt][ht] = hr;
                // GELU of the stored bf16 value: table lookup (bit-identical to the formula), formula for the rare group
                // with an argument outside the table
                const uint32_t w0 = __builtin_bit_cast(u32x2, hr)[0], w1 = __builtin_bit_cast(u32x2, hr)[1];
                bool bad = false;
                const uint32_t o0 = gelu_tab_off2<2>(w0, bad), o1 = gelu_tab_off2<2>(w1, bad);
                if (__builtin_expect(__any((int)bad), 0)) {
                    f32x4 act;
#pragma unroll
                    for (int e = 0; e < 4; ++e) act[e] = gelu_f(bf2f(hr[e]));
                    hb[mt][ht] = f2bf4(act);
                } else {
                    const unsigned char* tb = (const unsigned char*)gtab;
                    const uint32_t r0 = (uint32_t)*(const uint16_t*)(tb + (o0 & 0xffffu)) | ((uint32_t)*(const uint16_t*)(tb + (o0 >> 16)) << 16);
                    const uint32_t r1 = (uint32_t)*(const uint16_t*)(tb + (o1 & 0xffffu)) | ((uint32_t)*(const uint16_t*)(tb + (o1 >> 16)) << 16);
                    const u32x2 rr = {r0, r1};
                    hb[mt][ht] = __builtin_bit_cast(bf16x4, rr);
                }
            }
        STAMP(2);
        // ---- Y^T += W2[:, chunk] GELU(H^T chunk)
        bf16x8 hop[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hop[mt] = __builtin_shufflevector(hb[mt][0], hb[mt][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 wf = *(const bf16x8*)(W2s + (16 * t + fr) * P2 + 8 * g);       // (hidden 4 g .. + 3, 16 + 4 g .. + 3: see commit)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) yacc[mt][t] = mfma32(wf, hop[mt], yacc[mt][t]);
        }
        // next chunk's weights -> the other LDS buffer, then this chunk's stores: a wait for those loads placed after
        // the stores would also wait for the stores' acknowledgements (one in-order vmcnt counter)
        STAMP(3);
        if (ch + 1 < nch) commit((ch + 1) & 1);
        STAMP(4);
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // the pre-activation leaves through a wave-private LDS tile, two chunks (64 hidden units = 128 bytes per
                // row) at a time: straight from the accumulator layout a store instruction covers 16 rows x 32 bytes --
                // quarter cache lines, measured 10 us of the 100 us kernel against the same bytes in 512-byte runs
                if (a.hpre) *(bf16x4*)(Hst + (16 * mt + fr) * PHS + 32 * (ch & 1) + 16 * ht + 4 * g) = hrs[mt][ht];
            }
        if (!a.hpre) {
            // recompute mode (swv2_mlp_bwd / the fc2 weight gradient rebuild the pre-activation from x): nothing is kept
        } else if (ch & 1) {                            // two chunks staged: 8 rows x 128 bytes per store instruction
#pragma unroll
            for (int p_ = 0; p_ < 2 * MT; ++p_) {
                const int u = lane + 64 * p_, row = u >> 3, c8 = u & 7;
                const int r = min(row0 + row, a.M - 1);
                *(u32x4*)(a.hpre + (size_t)r * hid + 64 * (ch >> 1) + 8 * c8) = *(const u32x4*)(Hst + row * PHS + 8 * c8);
            }
        } else if (ch + 1 == nch) {                     // odd number of chunks: the last one alone (64 bytes per row)
            for (int u = lane; u < 16 * MT * 4; u += 64) {
                const int row = u >> 2, c8 = u & 3;
                const int r = min(row0 + row, a.M - 1);
                *(u32x4*)(a.hpre + (size_t)r * hid + 64 * (ch >> 1) + 8 * c8) = *(const u32x4*)(Hst + row * PHS + 8 * c8);
            }
        }
        STAMP(5);
#ifndef SWV2_MLP_ABL_NO_BARRIER      // (timing ablation, wrong results: tools/build_variant.sh)
        __syncthreads();
#endif
        STAMP(6);
    };
    // residual rows and drop-path scales of ALL row passes, unconditionally: loaded inside the pass loop each one was a
    // memory round trip of its own (the ISA read  load, s_waitcnt vmcnt(0), store  per pass; 19 % of the kernel in the
    // epilogue).  The first row tile's are requested BEFORE the last weight chunk (they land while its MFMAs run), the others behind
    // it (in flight while the first tile's LayerNorm statistics are computed): requested at the top of the epilogue they were one
    // bare memory round trip per workgroup with nothing in front of them.
    constexpr int UNITS = 16 * (C / 8), NP = (UNITS + 63) / 64;
    f32x4 xres[MT][NP][2];
    float scv[MT][NP];
    auto load_res = [&](const int mt) {
        const float* scp = a.scale ? a.scale : a.gamma;         // any valid address: the value is ignored without a scale
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int u = min(lane + 64 * p, UNITS - 1), row = u / (C / 8), c8 = u % (C / 8);
            const int rc = min(row0 + 16 * mt + row, a.M - 1);
            const size_t off = (size_t)rc * C + 8 * c8;
            xres[mt][p][0] = *(const f32x4*)(a.x + off);
            xres[mt][p][1] = *(const f32x4*)(a.x + off + 4);
            scv[mt][p] = scp[a.scale ? rc / a.rows_per_sample : 0];
        }
    };
    for (int ch = 0; ch + 1 < nch; ++ch) chunk(ch);
    load_res(0);                      // (measured 91.8 - 96.0 us against 96.1 - 97.0 with the request behind the last chunk: ~1 %)
    chunk(nch - 1);
#pragma unroll
    for (int mt = 1; mt < MT; ++mt) load_res(mt);

    // ---- epilogue.  Accumulator layout (lane (m = fr, g) holds n = 16 t + 4 g + r; a row is spread over the 4 lanes
    // with equal fr): + b2, bf16 round (the saved fc2 output), row mean / rstd (2 shuffle steps each).  The bf16 tile
    // and the statistics go through a per-wave LDS tile; normalisation, drop-path scale and residual then run in ROW
    // layout (a lane owns 8 consecutive columns), so the residual read and both stores are whole rows per instruction.
    uint16_t* As = (uint16_t*)(smem_raw + wave * EWAVE);
    float* St = (float*)(As + 16 * PA);                       // [16 rows][mean, rstd]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 b2v = *(const f32x4*)(cs + 16 * t + 4 * g);
            const bf16x4 ar = f2bf4(yacc[mt][t] + b2v);
            *(bf16x4*)(As + fr * PA + 16 * t + 4 * g) = ar;
#pragma unroll
            for (int e = 0; e < 4; ++e) { yacc[mt][t][e] = bf2f(ar[e]); s += yacc[mt][t][e]; }
        }
        s = xor32_allsum(xor16_allsum(s));
        const float mu = s * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = yacc[mt][t][e] - mu; q = fmaf(d, d, q); }
        q = xor32_allsum(xor16_allsum(q));
        const float rs = rsqrtf(q * (1.f / C) + a.eps);
        if (g == 0) {
            const int rc = min(row0 + 16 * mt + fr, a.M - 1);
            a.mean[rc] = mu;
            a.rstd[rc] = rs;
            St[2 * fr] = mu;
            St[2 * fr + 1] = rs;
        }
        __syncthreads();
        const int rbase = row0 + 16 * mt;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int u = lane + 64 * p, row = u / (C / 8), c8 = u % (C / 8);
            if (UNITS % 64 == 0 || u < UNITS) {
                const int rc = min(rbase + row, a.M - 1);
                const u32x4 av = *(const u32x4*)(As + row * PA + 8 * c8);
                const float mu_r = St[2 * row], rs_r = St[2 * row + 1];
                const float sc = a.scale ? scv[mt][p] : 1.f;
                const size_t off = (size_t)rc * C + 8 * c8;
                *(u32x4*)(a.a2 + off) = av;
                float v[8];
                unpack8(__builtin_bit_cast(uint4, av), v);
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const f32x4 gm = *(const f32x4*)(cs + C + 8 * c8 + 4 * hlf), bt = *(const f32x4*)(cs + 2 * C + 8 * c8 + 4 * hlf);
                    f32x4 o = xres[mt][p][hlf];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += sc * ((v[4 * hlf + e] - mu_r) * rs_r * gm[e] + bt[e]);
                    *(f32x4*)(a.y + off + 4 * hlf) = o;
                }
            }
        }
        if (mt + 1 < MT) __syncthreads();
    }
#ifdef SWV2_MLP_STAMPS
    STAMP(7);
    __syncthreads();
    if (lane == 0 && (blockIdx.x % 97) == 0) {       // a sample of workgroups: their own `mean` rows are overwritten with the stamps (probe only)
        unsigned long long* o = (unsigned long long*)(a.mean + wg_row0) + wave * 8;
        for (int i = 0; i < 8; ++i) o[i] = tacc[i];
    }
#endif
}

template <int C, int MT>
void launch_mlp_fwd(const MlpFwd& k, hipStream_t st) {
    // (<192, 1> holds an fc1-bias table of 1 024 entries, mlp_fwd_kernel::B1N: swv2_mlp_fwd selects <192, 2> for wider hidden layers)
    hipLaunchKernelGGL((mlp_fwd_kernel<C, MT>), dim3(cdiv(k.M, 64 * MT)), dim3(256), 0, st, k);
}

// ------------------------------------------------------------------------------------------------------------------
// Fused MLP branch, backward (autograd of the above): per row tile
//     da2 = LayerNorm2 backward of (scale * dy)                     (row statistics from the saved mean / rstd)
//     dH  = (da2 W2) * GELU'(hpre)        dH^T[hid][m] = W2^T[hid][n] da2^T[n][m]     A = W2^T rows (LDS), B = da2 (regs)
//     dx  = dy + dH W1                    dx^T[c][m]   = W1^T[c][hid] dH^T[hid][m]    A = W1^T rows (LDS), B = dH (accs)
// The same chained-MFMA skeleton as the forward: da2 fragments and the dx^T accumulators stay in registers, the
// weights stream through LDS in chunks of 32 hidden units, dH is written once (bf16, for the fc1 weight gradient) and
// never re-read here.  Replaces LN backward + two GEMM launches (693 MB -> 462 MB at the benchmark shape).
// d gamma / d beta: per-lane column sums -> DPP reduction over the 16 rows of a tile -> LDS over the 4 waves -> one
// partial row per workgroup in `ws`, folded by swv2_launch_ln_partials_reduce.
// ------------------------------------------------------------------------------------------------------------------
struct MlpBwd {
    const float* dy; const uint16_t* a2; const float* mean; const float* rstd; const float* gamma; const float* scale;
    const uint16_t* hpre; const uint16_t* w2t; const uint16_t* w1t;
    uint16_t* da2; uint16_t* dh; float* dx; float* ws;
    int M, hidden, rows_per_sample;
    float* zero; long zero_n;      // optional: a buffer this launch zeroes on the side (the block's parameter-gradient carve)
    const float* x; const uint16_t* w1; const float* b1;      // RECOMP: the forward's input, fc1.weight [hid][C] bf16, fc1.bias
};

// RECOMP: the fc1 pre-activation is not read back (16 T C bytes per block written by the forward and read here) but rebuilt
// from the block's x1 rows exactly as the forward built it -- H^T chunk = W1 chunk . X^T + b1 on the same MFMA sequence, rounded
// to bf16 like the value the forward fed to GELU -- at the price of one more product per chunk (the kernel is HBM-bound).  The
// fc1 weight then serves BOTH products that need it from ONE row-major LDS chunk: as A operand of the recompute (row reads)
// and, through transposed LDS reads, as A operand of dx^T = W1^T dH^T; the separately staged W1^T chunk goes away.
template <int C, int MT, bool RECOMP>
__global__ __launch_bounds__(256, (C >= 192 && MT == 2) ? 1 : 2) void mlp_bwd_kernel(const MlpBwd a) {
    // swv2_block_bwd: this is the FIRST kernel of a block's backward, and everything that accumulates into the block's 13
    // parameter gradients runs behind it on the stream -- so it zeroes them (16 bytes per thread and pass, spread over all
    // workgroups) and the separate 3 us fill launch in front of every block's backward goes away.
    if (a.zero) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.zero_n; i += (long)gridDim.x * 256 * 4)
            *(f32x4*)(a.zero + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    constexpr int KS = C / 32, NT = C / 16;
    constexpr int P1 = C + 8, P2 = 40;
    constexpr int W1E = 32 * P1, W2E = C * P2;
    constexpr int NCHUNK = 4 * C;
    constexpr int SPT = (NCHUNK + 255) / 256;
    constexpr int PY = C + 4;
    constexpr int ROWS = 64 * MT, PX = C + 8;          // rows per workgroup, pitch (bf16) of the staged da2 tile
    constexpr int XBYTES = ROWS * PX * 2;
    constexpr int GBYTES = (256 / (C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64)) * 2 * C * 4;      // d gamma / d beta row groups
    // RECOMP: the fc1 bias sits behind the weight chunk buffers during the chunk loop (a global load inside the loop would drain
    // the weight prefetch, see the forward kernel); at C = 128 that tail of the prologue's region is free, so the LDS footprint
    // -- two workgroups per CU at 79.9 KB each -- does not grow
    constexpr int B1BYTES = RECOMP ? MLP_RECOMP_MAX_HIDDEN * 4 : 0;
    constexpr int WBYTES = 2 * (W1E + W2E) * 2, EBYTES = 4 * 16 * PY * 4, PBYTES = XBYTES + GBYTES, WB1 = WBYTES + B1BYTES;
    constexpr int SBYTES0 = WB1 > EBYTES ? (WB1 > PBYTES ? WB1 : PBYTES) : (EBYTES > PBYTES ? EBYTES : PBYTES);
    constexpr int SBYTES = (SBYTES0 + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SBYTES];
    uint16_t* smem = (uint16_t*)smem_raw;
    float* const b1s = (float*)(smem_raw + WBYTES);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int wg_row0 = blockIdx.x * ROWS;
    const int row0 = wg_row0 + wave * (16 * MT);
    const int hid = a.hidden, nch = hid / 32;
    constexpr int PHS = 72;                                           // pitch (bf16) of the wave-private dh staging tile
    __shared__ __attribute__((aligned(16))) uint16_t hst_all[4 * 16 * MT * PHS];
    uint16_t* const Hst = hst_all + wave * (16 * MT * PHS);
    // GELU'(x) for every bf16 x in the table range (18 KB) -- where two workgroups per CU still fit beside it, or where only one fits
    // a CU anyway (192 channels, two row tiles per wave)
    constexpr int LBASE = SBYTES + 4 * 16 * MT * PHS * 2;
    constexpr bool ONE_WG = C >= 192 && MT == 2;          // (the launch bounds of that shape: one wave per SIMD, > 256 registers)
    constexpr bool GTAB = LBASE + GT_N * 4 <= (ONE_WG ? 156 : 80) * 1024;
    // round 5: where the full table does not fit beside the second workgroup (192 channels: 65.5 + 18 KB) its POSITIVE half does (9 KB):
    // GELU'(-x) = 1 - GELU'(x)  (Phi(-x) = 1 - Phi(x), x phi(x) odd), one subtraction + select per value instead of the formula
    // (an exp, a reciprocal and a dozen fmas: 26 % of a wave's life at C = 128 before the table); |difference| to the formula <= 1 ulp of 1
    constexpr bool GHALF = !GTAB && !ONE_WG && (LBASE + GT_HALF * 4 <= 80 * 1024);
    __shared__ __attribute__((aligned(16))) float ggtab[GTAB ? GT_N : (GHALF ? GT_HALF : 4)];
    if (GTAB)
        for (int i = tid; i < GT_N; i += 256) ggtab[i] = gelu_grad_f(bf2f(gelu_tab_arg(i)));
    else if (GHALF)
        for (int i = tid; i < GT_HALF; i += 256) ggtab[i] = gelu_grad_f(bf2f(gelu_tab_arg(i)));

    static_assert(32 * P1 <= W2E, "the row-major fc1 chunk of the recompute mode fits the W1^T chunk's slot");
    u32x4 s1[SPT], s2[SPT];
    auto issue = [&](int ch) {          // W2^T chunk: rows 32 ch .. + 32 of [hid][C];  W1^T chunk: columns of [C][hid]
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = min(tid + 256 * i, NCHUNK - 1);
            s1[i] = *(const u32x4*)(a.w2t + (size_t)(32 * ch + idx / (C / 8)) * C + 8 * (idx % (C / 8)));
            if constexpr (RECOMP) s2[i] = *(const u32x4*)(a.w1 + (size_t)(32 * ch + idx / (C / 8)) * C + 8 * (idx % (C / 8)));     // W1 chunk, rows
            else s2[i] = *(const u32x4*)(a.w1t + (size_t)(idx >> 2) * hid + 32 * ch + 8 * (idx & 3));
        }
    };
    auto commit = [&](int buf) {
        uint16_t* W1s = smem + buf * (W1E + W2E);
        uint16_t* W2s = W1s + W1E;
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = tid + 256 * i;
            if (NCHUNK % 256 == 0 || idx < NCHUNK) {
                *(u32x4*)(W1s + (idx / (C / 8)) * P1 + 8 * (idx % (C / 8))) = s1[i];
                if constexpr (RECOMP) {
                    *(u32x4*)(W2s + (idx / (C / 8)) * P1 + 8 * (idx % (C / 8))) = s2[i];
                } else {        // K = 32 operand order within a row, as in the forward kernel's commit
                    const int q = idx & 3;
                    uint16_t* row = W2s + (idx >> 2) * P2 + 4 * (q >> 1);
                    *(u32x2*)(row + 16 * (q & 1)) = (u32x2){s2[i][0], s2[i][1]};
                    *(u32x2*)(row + 16 * (q & 1) + 8) = (u32x2){s2[i][2], s2[i][3]};
                }
            }
        }
    };
    // saved pre-activation: fetched for TWO chunks at a time as 128-byte row segments (8 rows per load instruction) into
    // registers, a pair ahead, and handed to the accumulator layout (lane (m = fr, g): hidden 16 ht + 4 g .. + 3) through the
    // wave-private tile Hst.  Loaded straight in the accumulator layout an instruction touched 16 rows x 32 bytes.
    // Hst is shared with the dh stores: within a chunk its hpre values are read before its dh values are written to the
    // same place, the pair's dh rows are flushed at the end of the odd chunk, and only then the next pair's hpre lands.
    u32x4 hq[RECOMP ? 1 : 2 * MT];
    auto issue_hq = [&](int pair) {
        if constexpr (!RECOMP)
#pragma unroll
        for (int p_ = 0; p_ < 2 * MT; ++p_) {
            const int u = lane + 64 * p_, row = u >> 3, c8 = u & 7;
            const int r = min(row0 + row, a.M - 1), col = min(64 * pair + 8 * c8, hid - 8);
            hq[p_] = *(const u32x4*)(a.hpre + (size_t)r * hid + col);
        }
    };
    auto commit_hq = [&]() {
        if constexpr (!RECOMP)
#pragma unroll
        for (int p_ = 0; p_ < 2 * MT; ++p_) {
            const int u = lane + 64 * p_, row = u >> 3, c8 = u & 7;
            *(u32x4*)(Hst + row * PHS + 8 * c8) = hq[p_];
        }
    };
    // RECOMP: the x1 tile's fragments (B operand of the recompute) through the same LDS region the da2 tile uses afterwards
    bf16x8 xf1[RECOMP ? MT : 1][KS];
#ifdef SWV2_MLP_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    issue(0);
    issue_hq(0);
    if constexpr (RECOMP) {
        constexpr int UNITS = ROWS * (C / 4);
#pragma unroll
        for (int i = 0; i < UNITS / 256; ++i) {
            const int u = tid + 256 * i, row = u / (C / 4), c4 = u % (C / 4);
            const f32x4 v = *(const f32x4*)(a.x + (size_t)min(wg_row0 + row, a.M - 1) * C + 4 * c4);
            *(bf16x4*)(smem + row * PX + 4 * c4) = f2bf4(v);
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xf1[mt][ks] = *(const bf16x8*)(smem + (wave * 16 * MT + 16 * mt + fr) * PX + 32 * ks + 8 * g);
        __syncthreads();
    }

    // ---- LayerNorm backward in ROW layout: LPR lanes per row (4 columns each), the workgroup's rows in passes of RPP
    // rows.  A thread keeps the same 4 columns in every pass, so d gamma / d beta are private register sums over the
    // rows (one LDS reduction over the RPP row groups at the end) and all global accesses are whole rows.  da2 goes to
    // global (bf16, for the fc2 weight gradient) and to an LDS tile from which the waves pick up their B fragments.
    bf16x8 xf[MT][KS];
    {
        constexpr int LPR = C <= 32 ? 8 : C <= 64 ? 16 : C <= 128 ? 32 : 64;      // lanes per row (power of two >= C / 4)
        constexpr int RPP = 256 / LPR, NPASS = ROWS / RPP;
        constexpr int BATCH = NPASS < 4 ? NPASS : 4;                               // passes whose loads are in flight together
        static_assert(NPASS % BATCH == 0, "row passes must come in whole batches");
        const int lr = tid % LPR, rg = tid / LPR;
        const bool act = 4 * lr < C;
        const int c0 = act ? 4 * lr : 0;
        const f32x4 gm = *(const f32x4*)(a.gamma + c0);
        f32x4 dgm = {0.f, 0.f, 0.f, 0.f}, dbt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int pb = 0; pb < NPASS; pb += BATCH) {
            f32x4 d4[BATCH];
            u32x2 a4[BATCH];
            float mu[BATCH], rs[BATCH], sc[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int rc = min(wg_row0 + (pb + i) * RPP + rg, a.M - 1);
                d4[i] = *(const f32x4*)(a.dy + (size_t)rc * C + c0);
                a4[i] = *(const u32x2*)(a.a2 + (size_t)rc * C + c0);
                mu[i] = a.mean[rc];
                rs[i] = a.rstd[rc];
                sc[i] = a.scale ? a.scale[rc / a.rows_per_sample] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int row = (pb + i) * RPP + rg;
                const float once = (act && wg_row0 + row < a.M) ? 1.f : 0.f;   // duplicate rows (past M) count once
                const float av[4] = {__uint_as_float(a4[i][0] << 16), __uint_as_float(a4[i][0] & 0xffff0000u),
                                     __uint_as_float(a4[i][1] << 16), __uint_as_float(a4[i][1] & 0xffff0000u)};
                float gg[4], xh[4], t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = act ? sc[i] * d4[i][e] : 0.f;
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
                for (int e = 0; e < 4; ++e) o4[e] = rs[i] * (gg[e] - t1 - xh[e] * t2);
                const bf16x4 ob = f2bf4(o4);
                if (act) {
                    *(bf16x4*)(a.da2 + (size_t)min(wg_row0 + row, a.M - 1) * C + c0) = ob;
                    *(bf16x4*)(smem + row * PX + c0) = ob;
                }
            }
        }
        float* gs = (float*)(smem_raw + XBYTES);                   // [RPP row groups][2][C]
        if (act) {
            *(f32x4*)(gs + (rg * 2 + 0) * C + c0) = dgm;
            *(f32x4*)(gs + (rg * 2 + 1) * C + c0) = dbt;
        }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += 256) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPP; ++r) t += gs[r * 2 * C + i];
            a.ws[(size_t)blockIdx.x * 2 * C + i] = t;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xf[mt][ks] = *(const bf16x8*)(smem + (wave * 16 * MT + 16 * mt + fr) * PX + 32 * ks + 8 * g);
        __syncthreads();
    }
    STAMP(0);
    STAMP(1);

    f32x4 yacc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) yacc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (RECOMP)
        for (int i = tid; i < hid; i += 256) b1s[i] = a.b1[i];        // (the prologue is done with this region)
    commit(0);
    __syncthreads();
    // one chunk; `cur` holds this chunk's saved pre-activation, `nxt` receives the next chunk's (two register sets used
    // alternately: a copy between them would wait for the loads right where it is written)
    auto chunk = [&](int ch) {
        if (ch + 1 < nch) issue(ch + 1);
        if (!(ch & 1) && ch + 2 < nch) issue_hq((ch >> 1) + 1);
        const uint16_t* W1s = smem + (ch & 1) * (W1E + W2E);
        const uint16_t* W2s = W1s + W1E;
        f32x4 hacc[2][MT];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = *(const bf16x8*)(W1s + (16 * ht + fr) * P1 + 32 * ks + 8 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hacc[ht][mt] = mfma32(wf, xf[mt][ks], hacc[ht][mt]);
            }
        }
        STAMP(2);
        // RECOMP: the pre-activation of this chunk, as the forward computed it (bias as the accumulator's start value, the
        // same k order, rounded to bf16): only its table offsets are kept while the dH accumulators are finished
        u32x2 hpw[RECOMP ? 2 : 1][RECOMP ? MT : 1];
        if constexpr (RECOMP) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                f32x4 hp[MT];
                const f32x4 bias = *(const f32x4*)(b1s + 32 * ch + 16 * ht + 4 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hp[mt] = bias;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 wf = *(const bf16x8*)(W2s + (16 * ht + fr) * P1 + 32 * ks + 8 * g);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) hp[mt] = mfma32(wf, xf1[mt][ks], hp[mt]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hpw[ht][mt] = __builtin_bit_cast(u32x2, f2bf4(hp[mt]));
            }
        }
        bf16x4 hb[MT][2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                u32x2 cw;
                if constexpr (RECOMP) cw = hpw[ht][mt];
                else cw = *(const u32x2*)(Hst + (16 * mt + fr) * PHS + 32 * (ch & 1) + 16 * ht + 4 * g);
                const uint32_t w0 = cw[0], w1 = cw[1];
                // GELU'(stored bf16 pre-activation): fp32 table in LDS (same entries as the forward's GELU table, filled
                // with gelu_grad_f, so bit-identical to the formula); the formula -- an exp, a reciprocal and a dozen
                // fmas per value, 26 % of a wave's life -- only for the rare group with an argument outside the table
                bool bad = false;
                const uint32_t o0 = GHALF ? gelu_tab_off2_abs<4>(w0, bad) : gelu_tab_off2<4>(w0, bad);
                const uint32_t o1 = GHALF ? gelu_tab_off2_abs<4>(w1, bad) : gelu_tab_off2<4>(w1, bad);
                f32x4 gg;
                if ((!GTAB && !GHALF) || __builtin_expect(__any((int)bad), 0)) {
                    gg[0] = gelu_grad_f(__uint_as_float(w0 << 16));
                    gg[1] = gelu_grad_f(__uint_as_float(w0 & 0xffff0000u));
                    gg[2] = gelu_grad_f(__uint_as_float(w1 << 16));
                    gg[3] = gelu_grad_f(__uint_as_float(w1 & 0xffff0000u));
                } else {
                    const unsigned char* tb = (const unsigned char*)ggtab;
                    gg[0] = *(const float*)(tb + (o0 & 0xffffu));
                    gg[1] = *(const float*)(tb + (o0 >> 16));
                    gg[2] = *(const float*)(tb + (o1 & 0xffffu));
                    gg[3] = *(const float*)(tb + (o1 >> 16));
                    if constexpr (GHALF) {
                        gg[0] = (w0 & 0x8000u) ? 1.f - gg[0] : gg[0];
                        gg[1] = (w0 & 0x80000000u) ? 1.f - gg[1] : gg[1];
                        gg[2] = (w1 & 0x8000u) ? 1.f - gg[2] : gg[2];
                        gg[3] = (w1 & 0x80000000u) ? 1.f - gg[3] : gg[3];
                    }
                }
                hb[mt][ht] = f2bf4(hacc[ht][mt] * gg);
            }
        bf16x8 hop[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hop[mt] = __builtin_shufflevector(hb[mt][0], hb[mt][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8 wf;
            if constexpr (RECOMP) {     // W1^T fragments as transposed reads of the row-major chunk: element q = W1[4 g + q (+ 16)][16 t + fr]
                const bf16x4 lo = lds_tr_read(W2s + (4 * g + (fr >> 2)) * P1 + 16 * t + (fr & 3) * 4);
                const bf16x4 hi = lds_tr_read(W2s + (16 + 4 * g + (fr >> 2)) * P1 + 16 * t + (fr & 3) * 4);
                wf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                wf = *(const bf16x8*)(W2s + (16 * t + fr) * P2 + 8 * g);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) yacc[mt][t] = mfma32(wf, hop[mt], yacc[mt][t]);
        }
        STAMP(3);
        if (ch + 1 < nch) commit((ch + 1) & 1);       // before this chunk's stores (see the forward kernel)
        STAMP(4);
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)              // through the wave-private tile, 128 bytes per row (see the forward kernel)
                *(bf16x4*)(Hst + (16 * mt + fr) * PHS + 32 * (ch & 1) + 16 * ht + 4 * g) = hb[mt][ht];
        if (ch & 1) {
#pragma unroll
            for (int p_ = 0; p_ < 2 * MT; ++p_) {
                const int u = lane + 64 * p_, row = u >> 3, c8 = u & 7;
                const int r = min(row0 + row, a.M - 1);
                *(u32x4*)(a.dh + (size_t)r * hid + 64 * (ch >> 1) + 8 * c8) = *(const u32x4*)(Hst + row * PHS + 8 * c8);
            }
            if (ch + 1 < nch) commit_hq();              // next pair's pre-activations (fetched two chunks ago)
        } else if (ch + 1 == nch) {
            for (int u = lane; u < 16 * MT * 4; u += 64) {
                const int row = u >> 2, c8 = u & 3;
                const int r = min(row0 + row, a.M - 1);
                *(u32x4*)(a.dh + (size_t)r * hid + 64 * (ch >> 1) + 8 * c8) = *(const u32x4*)(Hst + row * PHS + 8 * c8);
            }
        }
        STAMP(5);
        __syncthreads();
        STAMP(6);
    };
    commit_hq();
    for (int ch = 0; ch < nch; ++ch) chunk(ch);

    // ---- dx = dy + (dH W1): accumulators -> per-wave LDS tile -> whole rows
    float* Ys = (float*)smem_raw + wave * 16 * PY;
    // the dy rows of ALL passes first: written as  dx[off] = dy[off] + ...  per pass, every load had to wait behind the previous
    // pass's store (dx and dy may alias as far as the compiler knows) -- 16 serial memory round trips per workgroup
    // (192 channels x 2 row tiles: 96 registers of dy beside 96 of accumulators -- there the rows of one tile at a time)
    constexpr int EPASS = 16 * (C / 4) / 64;
    constexpr bool DY_ALL = MT * EPASS <= 16;
    f32x4 dyv[DY_ALL ? MT : 1][EPASS];
    auto load_dy = [&](int mt) {
#pragma unroll
        for (int p = 0; p < EPASS; ++p) {
            const int u = lane + 64 * p, row = u / (C / 4), c4 = u % (C / 4);
            dyv[DY_ALL ? mt : 0][p] = *(const f32x4*)(a.dy + (size_t)min(row0 + 16 * mt + row, a.M - 1) * C + 4 * c4);
        }
    };
    if constexpr (DY_ALL) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_dy(mt);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (!DY_ALL) load_dy(mt);
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4*)(Ys + fr * PY + 16 * t + 4 * g) = yacc[mt][t];
        __syncthreads();
        const int rbase = row0 + 16 * mt;
#pragma unroll
        for (int p = 0; p < EPASS; ++p) {
            const int u = lane + 64 * p, row = u / (C / 4), c4 = u % (C / 4);
            const size_t off = (size_t)min(rbase + row, a.M - 1) * C + 4 * c4;
            *(f32x4*)(a.dx + off) = dyv[DY_ALL ? mt : 0][p] + *(const f32x4*)(Ys + row * PY + 4 * c4);
        }
        if (mt + 1 < MT) __syncthreads();
    }
#ifdef SWV2_MLP_STAMPS
    STAMP(7);
    if (lane == 0 && (blockIdx.x % 97) == 0) {
        unsigned long long* o = (unsigned long long*)(a.ws + (size_t)gridDim.x * 2 * C) + ((blockIdx.x / 97) * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = tacc[i];
    }
#endif
}

template <int C, int MT>
void launch_mlp_bwd(const MlpBwd& k, hipStream_t st) {
    if (k.hpre) hipLaunchKernelGGL((mlp_bwd_kernel<C, MT, false>), dim3(cdiv(k.M, 64 * MT)), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((mlp_bwd_kernel<C, MT, true>), dim3(cdiv(k.M, 64 * MT)), dim3(256), 0, st, k);
}

}  // namespace

extern "C" int swv2_mlp_supported(int C, int hidden) {
    return (C == 32 || C == 64 || C == 96 || C == 128 || C == 192 || C == 256) && hidden > 0 && hidden % 32 == 0 && hidden <= MLP_MAX_HIDDEN;
}

extern "C" int swv2_mlp_recompute_supported(int C, int hidden) {
    return swv2_mlp_supported(C, hidden) && hidden <= MLP_RECOMP_MAX_HIDDEN;
}

extern "C" int swv2_mlp_fwd(const swv2_mlp_args* a, void* stream) {
    SWV2_CHECK_ARG(a && a->x && a->w1 && a->b1 && a->w2 && a->b2 && a->gamma && a->beta && a->a2 && a->mean &&
                       a->rstd && a->y, "swv2_mlp_fwd: null pointer");      // (hpre may be NULL: not kept, see swv2_mlp_bwd)
    SWV2_CHECK_ARG(a->M > 0 && a->rows_per_sample > 0, "swv2_mlp_fwd: M and rows_per_sample must be positive");
    if (!swv2_mlp_supported(a->C, a->hidden)) {
        swv2_set_error("swv2_mlp_fwd: C=%d hidden=%d not instantiated (C in {32,64,96,128,192,256}, hidden %% 32 == 0); "
                       "use swv2_linear + swv2_ln_residual_fwd", a->C, a->hidden);
        return SWV2_ERR_UNSUPPORTED;
    }
    MlpFwd k = {a->x, (const uint16_t*)a->w1, a->b1, (const uint16_t*)a->w2, a->b2, a->gamma, a->beta, a->scale,
                (uint16_t*)a->hpre, (uint16_t*)a->a2, a->mean, a->rstd, a->y, a->M, a->hidden, a->rows_per_sample, a->eps};
    hipStream_t st = (hipStream_t)stream;
    // rows per workgroup (64 * MT): two tiles per wave halve the LDS fragment reads per MFMA (measured at C = 128,
    // M = 129600: 113 us vs 138 us), one tile keeps small problems spread over the CUs
    static const int force_mt = getenv("SWV2_MLP_MT") ? atoi(getenv("SWV2_MLP_MT")) : 0;
    const bool mt2 = force_mt ? force_mt == 2 : a->M >= 128 * 256;
    switch (a->C) {
        case 32: mt2 ? launch_mlp_fwd<32, 2>(k, st) : launch_mlp_fwd<32, 1>(k, st); break;
        case 64: mt2 ? launch_mlp_fwd<64, 2>(k, st) : launch_mlp_fwd<64, 1>(k, st); break;
        case 96: mt2 ? launch_mlp_fwd<96, 2>(k, st) : launch_mlp_fwd<96, 1>(k, st); break;
        case 128: mt2 ? launch_mlp_fwd<128, 2>(k, st) : launch_mlp_fwd<128, 1>(k, st); break;
        case 192: {
            // <192, 2>: 94 KB of LDS and 359 registers, ONE 4-wave workgroup per CU (cfg 4: 47.8 -> 45.7 ms/step against <192, 1> at one
            // workgroup per CU); <192, 1> with its fc1-bias table trimmed fits twice (SWV2_MLP_FWD192=2 restores two row tiles)
            static const int v192 = getenv("SWV2_MLP_FWD192") ? atoi(getenv("SWV2_MLP_FWD192")) : 1;
            // <192, 1> holds an fc1-bias table of 1 024 entries (mlp_fwd_kernel::B1N): wider hidden layers run <192, 2> whatever the row
            // count (selected on `mt2 &&` the small-M case wrote past the table)
            (a->hidden > 1024 || (mt2 && v192 == 2)) ? launch_mlp_fwd<192, 2>(k, st) : launch_mlp_fwd<192, 1>(k, st);
            break;
        }
        case 256: launch_mlp_fwd<256, 1>(k, st); break;
    }
    SWV2_CHECK_LAUNCH("swv2_mlp_fwd");
    return SWV2_OK;
}

extern "C" size_t swv2_mlp_bwd_ws_floats(int M, int C) { return (M > 0 && C > 0) ? (size_t)cdiv(M, 64) * 2 * C : 0; }

extern "C" int swv2_mlp_bwd(const swv2_mlp_bwd_args* a, void* stream) { return swv2_mlp_bwd_impl(a, stream, nullptr, nullptr, 0); }

int swv2_mlp_bwd_impl(const swv2_mlp_bwd_args* a, void* stream, int* deferred, float* zero, long zero_n) {
    SWV2_CHECK_ARG(!zero || (zero_n % 4 == 0 && ((uintptr_t)zero & 15) == 0), "swv2_mlp_bwd: the zeroed buffer must be 16-byte aligned / sized");
    SWV2_CHECK_ARG(a && a->dy && a->a2 && a->mean && a->rstd && a->gamma && a->w2t && a->da2 && a->dh &&
                       a->dx && a->dgamma && a->dbeta && a->ws, "swv2_mlp_bwd: null pointer");
    SWV2_CHECK_ARG((a->hpre && a->w1t) || (a->x && a->w1 && a->b1),
                   "swv2_mlp_bwd: either the saved pre-activation (hpre, w1t) or the recompute inputs (x, w1, b1) are needed");
    SWV2_CHECK_ARG(a->hpre || swv2_mlp_recompute_supported(a->C, a->hidden), "swv2_mlp_bwd: recompute mode needs hidden <= %d", MLP_RECOMP_MAX_HIDDEN);
    SWV2_CHECK_ARG(a->M > 0 && a->rows_per_sample > 0, "swv2_mlp_bwd: M and rows_per_sample must be positive");
    if (!swv2_mlp_supported(a->C, a->hidden)) {
        swv2_set_error("swv2_mlp_bwd: C=%d hidden=%d not instantiated; use swv2_ln_residual_bwd + swv2_linear", a->C, a->hidden);
        return SWV2_ERR_UNSUPPORTED;
    }
    MlpBwd k = {a->dy, (const uint16_t*)a->a2, a->mean, a->rstd, a->gamma, a->scale, (const uint16_t*)a->hpre,
                (const uint16_t*)a->w2t, (const uint16_t*)a->w1t, (uint16_t*)a->da2, (uint16_t*)a->dh, a->dx, a->ws, a->M,
                a->hidden, a->rows_per_sample, zero, zero_n, a->x, (const uint16_t*)a->w1, a->b1};
    hipStream_t st = (hipStream_t)stream;
    static const int force_mt = getenv("SWV2_MLP_MT") ? atoi(getenv("SWV2_MLP_MT")) : 0;
    const bool mt2 = force_mt ? force_mt == 2 : a->M >= 128 * 256;      // measured at C = 128, M = 129600: 170 us vs 185 us
    bool two192 = false;
    switch (a->C) {
        case 32: mt2 ? launch_mlp_bwd<32, 2>(k, st) : launch_mlp_bwd<32, 1>(k, st); break;
        case 64: mt2 ? launch_mlp_bwd<64, 2>(k, st) : launch_mlp_bwd<64, 1>(k, st); break;
        case 96: mt2 ? launch_mlp_bwd<96, 2>(k, st) : launch_mlp_bwd<96, 1>(k, st); break;
        case 128: mt2 ? launch_mlp_bwd<128, 2>(k, st) : launch_mlp_bwd<128, 1>(k, st); break;
        case 192: {
            // measured on BASELINE configs[4] (B = 2, ms per step): one row tile per wave, two workgroups per CU, GELU' by formula (the
            // full table does not fit beside two workgroups; since round 5 its positive half does, GHALF): 35.5; the same with the table and one workgroup per CU: 38.3; two row tiles
            // per wave (302 registers, one workgroup per CU, table; SWV2_MLP_BWD192=2): 37.2
            static const int v192 = getenv("SWV2_MLP_BWD192") ? atoi(getenv("SWV2_MLP_BWD192")) : 0;
            two192 = k.hpre && mt2 && v192 == 2;
            if (two192) hipLaunchKernelGGL((mlp_bwd_kernel<192, 2, false>), dim3(cdiv(k.M, 128)), dim3(256), 0, st, k);
            else launch_mlp_bwd<192, 1>(k, st);
            break;
        }
        case 256: launch_mlp_bwd<256, 1>(k, st); break;
    }
    const bool two = ((a->C <= 128) && mt2) || two192;
    if (deferred) *deferred = cdiv(a->M, two ? 128 : 64);
    else swv2_launch_ln_partials_reduce(a->ws, a->dgamma, a->dbeta, cdiv(a->M, two ? 128 : 64), a->C, st);
    SWV2_CHECK_LAUNCH("swv2_mlp_bwd");
    return SWV2_OK;
}
