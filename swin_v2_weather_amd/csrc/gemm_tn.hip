// Weight-gradient products of the GEMM engine: dW[N][K] += sum_m dY[m][N]^T X[m][K] (+ bias gradient).  See gemm.hip for
// the engine overview and gemm_common.h for the operand loaders shared with the NT kernel.
#include "gemm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// TN kernel: dW[N][K] (+)= sum over a slice of rows of dY^T X ; optional db[N] = column sums of dY
// grid = (ntiles_n * ntiles_k, splits).  Output accumulated with fp32 atomics (caller zeroes dW / db).
// ------------------------------------------------------------------------------------------------
constexpr int TM = 128;                     // rows per step (4 MFMA k-steps of 32): ~100 KB in flight per workgroup
constexpr int TP = 136;                     // LDS row pitch (elements) of the [TM][128] tiles: 272 B
constexpr int TCH = TM * 16 / NTHREADS;     // 16-byte chunks per thread per operand per step (8)

struct TnOut {                 // one product's destination
    float* dW; float* db; const int32_t* nmap; const int32_t* kmap; int ldw, M, N, K, ntk, tiles; float* ws;
};
constexpr int TN_SMEM = 2 * TM * TP;

// In-kernel phase timing (diagnostic builds only, -DSWV2_TN_STAMPS, tools/probe_tn_stamps.py): wave 0 of every workgroup
// accumulates s_memtime deltas per phase and overwrites the head of its partial tile with them (results are garbage then).
#ifdef SWV2_TN_STAMPS
#define TSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define TSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define TSTAMP_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define TSTAMP_DECL
#define TSTAMP_START() do {} while (0)
#define TSTAMP(k) do {} while (0)
#define TSTAMP_WAITV() do {} while (0)
#endif        // [Y | X][TM][TP] bf16, single buffer

// one 128 x 128 output tile over the row chunks of one slice (body shared by the single-product and the grouped kernel)
// XG: the X rows are gathered through xl.d.rowidx (compile time, so the un-gathered products carry no index load at all)
template <int YK, int XK, bool XG = false, bool HINC = false>
__device__ __forceinline__ void tn_tile(const ALoad<YK>& yl, const ALoad<XK>& xl_arg, const TnOut& o, int tile, int slice,
                                        int chunk_stride, uint16_t* smem, float* dbs, uint16_t* gtab) {
    float* __restrict__ dW = o.dW;
    float* __restrict__ db = o.db;
    const int32_t* __restrict__ nmap = o.nmap;
    const int32_t* __restrict__ kmap = o.kmap;
    const int ldw = o.ldw, M = o.M, N = o.N, K = o.K, ntk = o.ntk, tiles = o.tiles;
    float* __restrict__ ws = o.ws;
    ALoad<XK> xl = xl_arg;
    if constexpr (XK == A_BF16_GELU) {          // GELU-on-load of the stored bf16 pre-activation through a lookup table
        for (int i = threadIdx.x; i < GT_N; i += NTHREADS) gtab[i] = f2bf(gelu_f(bf2f(gelu_tab_arg(i))));
        xl.tab = gtab;
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int tn = tile / ntk, tk = tile - tn * ntk;
    const int n_base = tn * BN, k_base = tk * BN;
    // rows: slice s owns the TM-row chunks s, s + S, s + 2S, ... (S = chunk_stride = number of slices).  At any moment the S
    // slices then stream ONE contiguous S x TM-row region of dY / X, spread over every HBM channel; contiguous row ranges
    // per slice (the first version) put all slices a multiple of 256 KB .. 1 MB apart, i.e. on the same channels at the
    // same time, and the kernels sat at 2.6 - 3.0 TB/s.
    const int nchunks = (M + TM - 1) / TM;
    const int steps = (slice < chunk_stride && slice < nchunks) ? (nchunks - slice + chunk_stride - 1) / chunk_stride : 0;
    const int m_hi = M;
    const bool want_db = (db != nullptr) && (tk == 0);
    if (steps <= 0) return;
    if (tid < BN) dbs[tid] = 0.f;

    // staging: each tile is TM x 16 chunks; a thread keeps a fixed chunk column and walks rows srow + 16 i.  Operands whose
    // adjacent ROWS are adjacent in memory (the 4 x 4 patch im2col: a row is 16 bytes of each of 16 x Cin image rows) use
    // the row-fastest map instead -- thread = (row tid & 127, chunk columns (tid >> 7) + 2 i) -- so a wave reads 64
    // consecutive 16-byte groups instead of 4 rows x 16 scattered ones (PatchEmbed / head gradients: 2.9 TB/s before).
    const int srow = tid >> 4, scol = tid & 15;
    constexpr bool YRF = ALoad<YK>::ROW_FASTEST, XRF = ALoad<XK>::ROW_FASTEST;
    auto yr_ = [&](int i) { return YRF ? (tid & (TM - 1)) : srow + 16 * i; };
    auto yc_ = [&](int i) { return YRF ? (tid >> 7) + 2 * i : scol; };
    auto xr_ = [&](int i) { return XRF ? (tid & (TM - 1)) : srow + 16 * i; };
    auto xc_ = [&](int i) { return XRF ? (tid >> 7) + 2 * i : scol; };
    static_assert(TM == 128 && NTHREADS == 256, "row-fastest map: 128 rows x 2 chunk columns per pass");
    typename ALoad<YK>::Raw ry[TCH];
    typename ALoad<XK>::Raw rx[TCH];
    float colsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Staging pipeline.  EVERY global load here is unconditional (clamped address) and the validity of a chunk travels in a
    // bit mask: rows past M and the padded rows of a window (table entry -1) are loaded from row 0 and zeroed when the
    // tile is written to LDS.  With loads under `if (row < 0) ...` -- the first version -- the compiler guarded their
    // destination registers with s_waitcnt vmcnt(0): one in the middle of issue() and one right behind the index loads of
    // resolve(), i.e. BEFORE the MFMAs, so the "prefetch" of the next step was waited for at once and the kernel ran
    // load -> wait -> compute in series (2.6 - 3.2 TB/s; found in the ISA, tools/probe_tn_stamps.py showed 40 % in issue).
    // Gathered X rows: the table entries of step s + 2 are fetched behind the data loads of step s + 1 and consumed one
    // iteration later, by which time they have long arrived.
    int yrow[TCH], xrow[TCH];          // source rows of the next issue (X gathered: raw table entries)
    uint32_t xin = 0;                  // X gathered: bit i = chunk row i of the next issue lies inside [0, M)
    uint32_t okmask = 0;               // chunks in flight: bit i = Y chunk i valid, bit 16 + i = X chunk i valid
    // LINEAR operands (plain row-major rows, not gathered): chunk i of a step sits at  (m0 + srow + 16 i) * ld + column  -- one
    // 32-bit multiply per operand and step, then adds.  Formed per chunk as  ptr + (long)row * ld + k  the staging loop carried
    // 64 v_mul_lo_u32 + 40 v_mad_u64_u32 + 8 v_mul_hi per step (quarter-rate instructions: about as many vector-issue cycles as the
    // step's 64 MFMAs, found in the ISA of the grouped kernel); chunks past M load from the operand's first rows (masked at commit).
    constexpr bool YLIN = ALoad<YK>::LINEAR && !YRF, XLIN = ALoad<XK>::LINEAR && !XRF && !XG;
    const uint32_t ycol = (uint32_t)min(n_base + scol * 8, max(N - 8, 0)), xcol = (uint32_t)min(k_base + scol * 8, max(K - 8, 0));
    const uint32_t ystep = 16u * (uint32_t)yl.d.ld, xstep = 16u * (uint32_t)xl.d.ld;
    uint32_t yoff0 = 0, xoff0 = 0;     // element offset of chunk 0 of the next issue
    // head-major operands ([Bw][h][S][Lp][DP]): the window / token split once per step, then adds (ALoad<A_HEADS>::Step); needs a
    // step (TM rows) to span at most one window boundary
    // (HINC: selected by the kernel when Lp >= TM; column offsets of masked chunks: the operand's first row)
    constexpr bool YHD = YK == A_HEADS && HINC, XHD = XK == A_HEADS && HINC;
    typename ALoad<A_HEADS>::Step yst = {0u, 0}, xst = {0u, 0};
    auto resolve = [&](int s) {
        xin = 0;
        if constexpr (YLIN || XLIN || YHD || XHD) {
            const int m0 = (slice + s * chunk_stride) * TM + srow;
            if constexpr (YLIN) yoff0 = (uint32_t)m0 * (uint32_t)yl.d.ld + ycol;
            if constexpr (XLIN) xoff0 = (uint32_t)m0 * (uint32_t)xl.d.ld + xcol;
            if constexpr (YHD) yst = yl.step_base(min(m0, M - 1), (int)ycol);
            if constexpr (XHD) xst = xl.step_base(min(m0, M - 1), (int)xcol);
        }
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
            const int m0 = (slice + s * chunk_stride) * TM, my = m0 + yr_(i), mx = m0 + xr_(i);
            const bool iny = (s < steps) && (my < m_hi), inx = (s < steps) && (mx < m_hi);
            yrow[i] = iny ? my : -1;
            if constexpr (XG) {
                xrow[i] = xl.d.rowidx[min(mx, M - 1)];
                xin |= (uint32_t)inx << i;
            } else {
                xrow[i] = inx ? mx : -1;
            }
        }
    };
    auto issue = [&](int s) {
        okmask = 0;
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
            const int yk = n_base + yc_(i) * 8, xk = k_base + xc_(i) * 8;
            const int xr = XG ? (((xin >> i) & 1) ? xrow[i] : -1) : xrow[i];
            const bool yok = yrow[i] >= 0 && yk < N, xok = xr >= 0 && xk < K;
            if constexpr (YLIN) ry[i] = yl.raw_lin(yrow[i] >= 0 ? yoff0 + i * ystep : ycol);
            else if constexpr (YHD) ry[i] = yl.raw_lin(yrow[i] >= 0 ? yl.step_off(yst, i) : 0u);
            else ry[i] = yl.raw_unc(max(yrow[i], 0), yk < N ? yk : 0);
            if constexpr (XLIN) rx[i] = xl.raw_lin(xr >= 0 ? xoff0 + i * xstep : xcol);
            else if constexpr (XHD) rx[i] = xl.raw_lin(xr >= 0 ? xl.step_off(xst, i) : 0u);
            else if constexpr (XG && XK == A_F32) rx[i] = xl.raw_lin((uint32_t)max(xr, 0) * (uint32_t)xl.d.ld + xcol);     // gathered rows: one 32-bit multiply
            else rx[i] = xl.raw_unc(max(xr, 0), xk < K ? xk : 0);
            okmask |= ((uint32_t)yok << i) | ((uint32_t)xok << (16 + i));
        }
    };
    auto commit = [&]() {
        uint16_t* Ys = smem;
        uint16_t* Xs = smem + TM * TP;
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
            const uint4 yv = ((okmask >> i) & 1) ? yl.cvt(ry[i]) : zero4;
            *(uint4*)(Ys + yr_(i) * TP + yc_(i) * 8) = yv;
            *(uint4*)(Xs + xr_(i) * TP + xc_(i) * 8) = ((okmask >> (16 + i)) & 1) ? xl.cvt(rx[i]) : zero4;
            if (want_db && !YRF) {                       // (a row-fastest dY has no fixed column per thread: rejected by the launcher)
                float v[8];
                unpack8(yv, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) colsum[e] += v[e];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TSTAMP_DECL
    TSTAMP_START();
    resolve(0);
    issue(0);
    resolve(1);
    TSTAMP(7);
    for (int s = 0; s < steps; ++s) {
        TSTAMP_WAITV();
        TSTAMP(0);
        commit();
        TSTAMP(1);
        __syncthreads();
        TSTAMP(2);
        if (s + 1 < steps) { issue(s + 1); resolve(s + 2); }
        TSTAMP(3);
        const uint16_t* Ys = smem;
        const uint16_t* Xs = smem + TM * TP;
        // A operand = dY^T (rows n, k = m), B operand = X (k = m, cols k'): both are transposed reads of row-major tiles
#pragma unroll
        for (int kk = 0; kk < TM / 32; ++kk) {
            bf16x8 af[4], bf[4];
            const int r0 = 32 * kk + 8 * g + (fr >> 2), cc = (fr & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x4 a0 = lds_tr_read(Ys + r0 * TP + wr * 64 + i * 16 + cc);
                const bf16x4 a1 = lds_tr_read(Ys + (r0 + 4) * TP + wr * 64 + i * 16 + cc);
                const bf16x4 b0 = lds_tr_read(Xs + r0 * TP + wc * 64 + i * 16 + cc);
                const bf16x4 b1 = lds_tr_read(Xs + (r0 + 4) * TP + wc * 64 + i * 16 + cc);
                af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[3][3][3]));
        TSTAMP(4);
        __syncthreads();                    // all waves done with the tile before the next commit overwrites it
        TSTAMP(5);
    }
    // accumulate the tile: rows n = n_base + wr*64 + 16i + 4g + r, cols k = k_base + wc*64 + 16j + fr
    if (ws) {                               // workspace path: plain partial tile, summed by tn_reduce_kernel
        float* pt = ws + ((size_t)slice * tiles + tile) * (BN * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pt[(wr * 64 + 16 * i + 4 * g + r) * BN + wc * 64 + 16 * j + fr] = acc[i][j][r];
#ifdef SWV2_TN_STAMPS
        TSTAMP_WAITV();
        TSTAMP(6);
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ((unsigned long long*)pt)[k] = st_acc[k];
        }
#endif
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = n_base + wr * 64 + 16 * i + 4 * g + r, k = k_base + wc * 64 + 16 * j + fr;
                if (n >= N || k >= K) continue;
                if (nmap) n = nmap[n];
                if (kmap) k = kmap[k];
                if (n >= 0 && k >= 0) atomicAdd(dW + (long)n * ldw + k, acc[i][j][r]);
            }
    if (want_db) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&dbs[scol * 8 + e], colsum[e]);
        __syncthreads();
        if (tid < BN && n_base + tid < N) {
            const int n = nmap ? nmap[n_base + tid] : n_base + tid;
            if (n >= 0) atomicAdd(db + n, dbs[tid]);
        }
    }
}

// single product: grid = tiles * slices (slices padded to a multiple of 8)
// XCD-aware block -> (tile, row slice) map.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), and
// the tiles of one row slice re-read the same dY / X rows: give all tiles of a slice consecutive slots on ONE XCD so
// the re-reads hit that XCD's L2 instead of HBM (PMC before: 415 MB vs 198 MB algorithmic for the fc1 gradient).
// Speed only, never correctness.
template <int YK, int XK, bool XG>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(ALoad<YK> yl, ALoad<XK> xl, TnOut o, int chunk_stride) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[TN_SMEM];
    __shared__ float dbs[BN];
    __shared__ __attribute__((aligned(16))) uint16_t gtab[XK == A_BF16_GELU ? GT_N : 8];
    const int lin = blockIdx.x;                       // 1-D launch: lin = xcd + 8 * (q * tiles + tile)
    const int x8 = lin & 7, rest = lin >> 3;
    const int tile = rest % o.tiles, slice = (rest / o.tiles) * 8 + x8;
    tn_tile<YK, XK, XG>(yl, xl, o, tile, slice, chunk_stride, smem, dbs, gtab);
}

// The four weight-gradient products of one transformer block in ONE launch (fc2, fc1, proj, qkv: 4 + 4 + 1 + 3 tiles at
// C = 128).  Measured on the separate launches (bench.py at local batch 1 / 2 / 4, tools/batch_fit.py): each costs
// ~13 - 16 us independent of the row count (ramp, first-load latency, partial-tile drain) on top of ~4.3 us per 128-row
// step, i.e. 2.6 - 3.2 TB/s at 8 steps per workgroup but 5.8 TB/s marginal.  Grouped, every workgroup walks ~25 steps and
// the fixed part is paid once; one tn_group_reduce launch replaces four tn_reduce launches.
struct TnGroup {
    ALoad<A_BF16> y0; ALoad<A_BF16_GELU> x0;      // fc2:  d(a2)^T GELU(hpre)
    ALoad<A_BF16> y1; ALoad<A_F32> x1;            // fc1:  d(h)^T x1
    ALoad<A_BF16> y2; ALoad<A_HEADS> x2;          // proj: d(a1)^T merge(oh)
    ALoad<A_HEADS> y3; ALoad<A_F32> x3;           // qkv:  d(qkv)^T gather(x)
    TnOut o[4];
    int first[5];                                 // first global tile index of each product; first[4] = total
    int slices;
};
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_group_kernel(TnGroup a) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[TN_SMEM];
    __shared__ float dbs[BN];
    __shared__ __attribute__((aligned(16))) uint16_t gtab[GT_N];
    const int lin = blockIdx.x, x8 = lin & 7, rest = lin >> 3, tt = a.first[4];
    const int t = rest % tt, slice = (rest / tt) * 8 + x8;
    if (t < a.first[1]) tn_tile<A_BF16, A_BF16_GELU>(a.y0, a.x0, a.o[0], t, slice, a.slices, smem, dbs, gtab);
    else if (t < a.first[2]) tn_tile<A_BF16, A_F32>(a.y1, a.x1, a.o[1], t - a.first[1], slice, a.slices, smem, dbs, gtab);
    else if (a.x2.d.p2 >= TM) {       // head-major operands walked incrementally (a step spans at most one window boundary)
        if (t < a.first[3]) tn_tile<A_BF16, A_HEADS, false, true>(a.y2, a.x2, a.o[2], t - a.first[2], slice, a.slices, smem, dbs, gtab);
        else tn_tile<A_HEADS, A_F32, true, true>(a.y3, a.x3, a.o[3], t - a.first[3], slice, a.slices, smem, dbs, gtab);
    } else {
        if (t < a.first[3]) tn_tile<A_BF16, A_HEADS>(a.y2, a.x2, a.o[2], t - a.first[2], slice, a.slices, smem, dbs, gtab);
        else tn_tile<A_HEADS, A_F32, true>(a.y3, a.x3, a.o[3], t - a.first[3], slice, a.slices, smem, dbs, gtab);
    }
}

// ------------------------------------------------------------------------------------------------
// Second stage of the workspace path: dW[nmap(n)][kmap(k)] += sum over row slices of the partial tiles
// ws[slice][tile][128][128].  Device-scope float atomics resolve beyond the per-XCD L2s and cost ~5 ns each when 64+
// workgroups hit the same 64 K addresses (measured: the fc1 gradient went 94 -> 177 us from 64 to 256 slices), so with
// a workspace the slices write plain coalesced partials and this kernel owns every output element exclusively.
// block = 256 threads = 64 tile entries x 4 slice groups.
// ------------------------------------------------------------------------------------------------
struct TnReduce { TnOut o[4]; int first[5]; int slices; };
__global__ __launch_bounds__(256) void tn_reduce_kernel(TnReduce a) {
    __shared__ float part[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6, gt = blockIdx.y;
    const int pi = (gt >= a.first[1]) + (gt >= a.first[2]) + (gt >= a.first[3]);
    const TnOut& o = a.o[pi];
    const int tile = gt - a.first[pi], tiles = o.tiles, ntk = o.ntk;
    const int slices = min(a.slices, (o.M + TM - 1) / TM);
    const float* src = o.ws + (size_t)tile * (BN * BN) + e;
    const size_t stride = (size_t)tiles * (BN * BN);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sl = sg;
    for (; sl + 12 < slices; sl += 16) {
        s0 += src[(size_t)sl * stride];
        s1 += src[(size_t)(sl + 4) * stride];
        s2 += src[(size_t)(sl + 8) * stride];
        s3 += src[(size_t)(sl + 12) * stride];
    }
    for (; sl < slices; sl += 4) s0 += src[(size_t)sl * stride];
    part[sg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0) {
        const int tn = tile / ntk, tk = tile - tn * ntk;
        int n = tn * BN + (e >> 7), k = tk * BN + (e & 127);
        if (n < o.N && k < o.K) {
            if (o.nmap) n = o.nmap[n];
            if (o.kmap) k = o.kmap[k];
            const int l = threadIdx.x;
            if (n >= 0 && k >= 0) o.dW[(long)n * o.ldw + k] += (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wide weight gradients (N, K multiples of 256, both >= 512: the reference's embed_dim 768 model): gemm.hip's gemm_tn_wide_kernel
// writes S partial matrices, tn_wide_reduce_kernel adds them into dW / db through the row / column maps.  An fp32 X operand
// (x, x1: the residual stream) is first cast -- and gathered -- into a bf16 matrix in the workspace (the wide kernel's operands
// are raw bf16 by LDS-DMA).
// ------------------------------------------------------------------------------------------------
struct TnWidePlan { bool ok; int S; size_t part_bytes, db_bytes, cast_bytes, total; };
TnWidePlan tn_wide_plan(int M, int N, int K) {
    TnWidePlan p = {};
    p.ok = (N % 256 == 0) && (K % 256 == 0) && N >= 512 && K >= 512 && (M % 32 == 0) && M >= 8192 &&
           (double)M * (N > K ? N : K) * 2 < 4.29e9 && (double)N * K < 2.0e9;
    if (!p.ok) return p;
    const int tiles = (N / 256) * (K / 256);
    p.S = std::max(1, 256 / tiles);
    if (p.S > M / 32) p.S = M / 32;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    p.part_bytes = up((size_t)p.S * N * K * 4);
    p.db_bytes = up((size_t)p.S * (K / 256) * N * 4);
    p.cast_bytes = up((size_t)M * K * 2);                    // (needed for an fp32 X only; always budgeted)
    p.total = p.part_bytes + p.db_bytes + p.cast_bytes;
    return p;
}
// bf16 copy of fp32 rows, optionally gathered (table entry < 0: a zero row); 8 elements per thread
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rowidx, uint16_t* __restrict__ dst,
                                                        int M, int C, long ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, per_row = C / 8;
    if (i >= (long)M * per_row) return;
    const int m = (int)(i / per_row), c = (int)(i - (long)m * per_row) * 8;
    const int r = rowidx ? rowidx[m] : m;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (r >= 0) {
        const float* p = src + (long)r * ld + c;
        RawF32 v = {*(const f32x4*)p, *(const f32x4*)(p + 4)};
        o = cvt_f32x8(v);
    }
    *(uint4*)(dst + (long)m * C + c) = o;
}
// dW[nmap(n)][kmap(k)] += sum_s part[s][n][k] ; db[nmap(n)] += sum_{s, tk} dbpart[s][tk][n]     (fixed order: deterministic)
__global__ __launch_bounds__(256) void tn_wide_reduce_kernel(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dW,
                                                             float* __restrict__ db, const int32_t* __restrict__ nmap,
                                                             const int32_t* __restrict__ kmap, int ldw, int N, int K, int S) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4, NK = N * K;          // (N K < 2^31: checked by the plan)
    if (i < NK) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int sl = 0;
        for (; sl + 1 < S; sl += 2) { s0 += *(const f32x4*)(part + (size_t)sl * NK + i); s1 += *(const f32x4*)(part + (size_t)(sl + 1) * NK + i); }
        if (sl < S) s0 += *(const f32x4*)(part + (size_t)sl * NK + i);
        s0 += s1;
        int n = i / K;
        const int k = i - n * K;
        if (nmap) n = nmap[n];
        if (n >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = kmap ? kmap[k + e] : k + e;
                if (kk >= 0) dW[(long)n * ldw + kk] += s0[e];
            }
        }
    }
    // bias gradient: the first N / 256 workgroups own 256 entries each; the S * ntk partial rows in four independent chains
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (db && dbpart && n < N) {
        const int rows = S * (K / 256);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = 0;
        for (; j + 3 < rows; j += 4) {
            a0 += dbpart[(size_t)j * N + n]; a1 += dbpart[(size_t)(j + 1) * N + n];
            a2 += dbpart[(size_t)(j + 2) * N + n]; a3 += dbpart[(size_t)(j + 3) * N + n];
        }
        for (; j < rows; ++j) a0 += dbpart[(size_t)j * N + n];
        const int nn = nmap ? nmap[n] : n;
        if (nn >= 0) db[nn] += (a0 + a1) + (a2 + a3);
    }
}

struct TnPlan { int ntn, ntk, tiles, real_slices, slices; size_t ws_bytes; };
TnPlan tn_plan(int M, int N, int K, int splits) {
    TnPlan p;
    p.ntn = cdiv(N, BN); p.ntk = cdiv(K, BN); p.tiles = p.ntn * p.ntk;
    int eff = (p.tiles >= 3) ? splits : splits * 2;               // small outputs: more row slices to fill the chip
    // large outputs need few: 128 slices of a 768 x 3072 gradient (144 tiles) were 18 432 workgroups writing 1.2 GB of
    // partial tiles, and the reduction launch cost as much as a GEMM -- two rounds of workgroups (1024) are plenty
    eff = std::min(eff, std::max(8, (1024 / p.tiles) / 8 * 8));
    p.real_slices = std::min(eff, cdiv(M, TM));
    p.slices = cdiv(p.real_slices, 8) * 8;                        // padded to a multiple of 8 (empty slices exit at once)
    p.ws_bytes = (size_t)p.real_slices * p.tiles * BN * BN * sizeof(float);
    return p;
}
TnOut tn_out(float* dW, float* db, const int32_t* nmap, const int32_t* kmap, int ldw, int M, int N, int K, float* ws) {
    TnOut o;
    o.dW = dW; o.db = db; o.nmap = nmap; o.kmap = kmap; o.ldw = ldw; o.M = M; o.N = N; o.K = K;
    o.ntk = cdiv(K, BN); o.tiles = cdiv(N, BN) * o.ntk; o.ws = ws;
    return o;
}

template <int YK, int XK>
int launch_tn2(const swv2_operand* y, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
               const int32_t* kmap, int ldw, int M, int N, int K, int splits, float* ws, size_t ws_bytes, hipStream_t st) {
    // wide widths: 256 x 256 tiles by LDS-DMA (gemm.hip), workspace path only
    if constexpr ((YK == A_BF16 || YK == A_HEADS) && (XK == A_BF16 || XK == A_HEADS || XK == A_F32)) {
        const TnWidePlan wp = tn_wide_plan(M, N, K);
        const int wide = getenv("SWV2_GEMM_WIDE") ? atoi(getenv("SWV2_GEMM_WIDE")) : 1;
        const bool kinds = !(YK == A_HEADS && XK == A_HEADS) && !y->rowidx && (XK == A_F32 || !x->rowidx);
        if (wide && wp.ok && kinds && ws && ws_bytes >= wp.total) {
            float* part = ws;
            float* dbpart = db ? (float*)((char*)ws + wp.part_bytes) : nullptr;
            swv2_operand xb = *x;
            if constexpr (XK == A_F32) {           // bf16 (gathered) copy of the fp32 rows
                uint16_t* cb = (uint16_t*)((char*)ws + wp.part_bytes + wp.db_bytes);
                const long chunks = (long)M * (K / 8);
                hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const float*)x->ptr, x->rowidx, cb,
                                   M, K, x->ld);
                xb.kind = SWV2_OP_BF16; xb.ptr = cb; xb.rowidx = nullptr; xb.ld = K;
            }
            int rc = swv2_tn_wide_launch(y, &xb, part, dbpart, M, N, K, wp.S, st);
            if (rc) return rc;
            const long quads = (long)N * K / 4;
            hipLaunchKernelGGL(tn_wide_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, (const float*)part,
                               (const float*)dbpart, dW, db, nmap, kmap, ldw, N, K, wp.S);
            SWV2_CHECK_LAUNCH("swv2_linear_wgrad");
            return SWV2_OK;
        }
    }
    const TnPlan p = tn_plan(M, N, K, splits);
    if (ws && ws_bytes < p.ws_bytes) {
        swv2_set_error("swv2_linear_wgrad_ws: workspace of %zu bytes, %zu needed (swv2_linear_wgrad_ws_bytes)", ws_bytes, p.ws_bytes);
        return SWV2_ERR_INVALID;
    }
    const TnOut o = tn_out(dW, db, nmap, kmap, ldw, M, N, K, ws);
    // gathered operands: dY never; X only as fp32 rows (the qkv gradient's window gather)
    if (y->rowidx || (x->rowidx && XK != A_F32)) {
        swv2_set_error("swv2_linear_wgrad: a row gather is supported for a fp32 X operand only (dY kind %d, X kind %d)", y->kind, x->kind);
        return SWV2_ERR_UNSUPPORTED;
    }
    if constexpr (XK == A_F32) {
        if (x->rowidx)
            hipLaunchKernelGGL((gemm_tn_kernel<YK, XK, true>), dim3(p.tiles * p.slices), dim3(NTHREADS), 0, st, make_loader<YK>(y),
                               make_loader<XK>(x), o, p.real_slices);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<YK, XK, false>), dim3(p.tiles * p.slices), dim3(NTHREADS), 0, st, make_loader<YK>(y),
                               make_loader<XK>(x), o, p.real_slices);
    } else {
        hipLaunchKernelGGL((gemm_tn_kernel<YK, XK, false>), dim3(p.tiles * p.slices), dim3(NTHREADS), 0, st, make_loader<YK>(y),
                           make_loader<XK>(x), o, p.real_slices);
    }
    if (ws) {
        TnReduce r = {};
        r.o[0] = o; r.first[0] = 0; r.first[1] = r.first[2] = r.first[3] = r.first[4] = p.tiles; r.slices = p.real_slices;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(BN * BN / 64, p.tiles), dim3(256), 0, st, r);
    }
    SWV2_CHECK_LAUNCH("swv2_linear_wgrad");
    return SWV2_OK;
}

template <int YK>
int launch_tn1(const swv2_operand* y, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
               const int32_t* kmap, int ldw, int M, int N, int K, int splits, float* ws, size_t wsb, hipStream_t st) {
    // only the (dY, X) operand pairs that occur in the model are instantiated (compile time)
    if (x->kind == SWV2_OP_F32) return launch_tn2<YK, A_F32>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
    if constexpr (YK == A_BF16) {
        switch (x->kind) {
            case SWV2_OP_BF16: return launch_tn2<YK, A_BF16>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
            case SWV2_OP_BF16_GELU: return launch_tn2<YK, A_BF16_GELU>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
            case SWV2_OP_HEADS: return launch_tn2<YK, A_HEADS>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
            case SWV2_OP_PATCH: return launch_tn2<YK, A_PATCH>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
        }
    }
    if constexpr (YK == A_F32) {
        if (x->kind == SWV2_OP_MERGE_LN) return launch_tn2<YK, A_MERGE_LN>(y, x, dW, db, nmap, kmap, ldw, M, N, K, splits, ws, wsb, st);
    }
    swv2_set_error("swv2_linear_wgrad: operand pair (dY kind %d, X kind %d) is not instantiated", y->kind, x->kind);
    return SWV2_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" size_t swv2_linear_wgrad_ws_bytes(int M, int N, int K, int splits) {
    if (!(M > 0 && N > 0 && K > 0 && splits > 0)) return 0;
    const TnWidePlan w = tn_wide_plan(M, N, K);
    return std::max(tn_plan(M, N, K, splits).ws_bytes, w.ok ? w.total : (size_t)0);
}

extern "C" int swv2_linear_wgrad_ws(const swv2_operand* dy, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
                                    const int32_t* kmap, int ldw, int splits, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_operand(dy, "swv2_linear_wgrad(dy)");
    if (rc) return rc;
    rc = check_operand(x, "swv2_linear_wgrad(x)");
    if (rc) return rc;
    SWV2_CHECK_ARG(dW && splits > 0 && ldw > 0, "swv2_linear_wgrad: null dW, non-positive splits or bad pitch");
    SWV2_CHECK_ARG(dy->rows == x->rows, "swv2_linear_wgrad: row counts differ (%d vs %d)", dy->rows, x->rows);
    SWV2_CHECK_ARG(!(db && dy->kind == SWV2_OP_PATCH), "swv2_linear_wgrad: no bias gradient with a patch-layout dY (row-fastest staging)");
    const int M = dy->rows, N = dy->cols, K = x->cols;
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    switch (dy->kind) {
        case SWV2_OP_F32: return launch_tn1<A_F32>(dy, x, dW, db, nmap, kmap, ldw, M, N, K, splits, w, ws_bytes, st);
        case SWV2_OP_BF16: return launch_tn1<A_BF16>(dy, x, dW, db, nmap, kmap, ldw, M, N, K, splits, w, ws_bytes, st);
        case SWV2_OP_HEADS: return launch_tn1<A_HEADS>(dy, x, dW, db, nmap, kmap, ldw, M, N, K, splits, w, ws_bytes, st);
        case SWV2_OP_PATCH: return launch_tn1<A_PATCH>(dy, x, dW, db, nmap, kmap, ldw, M, N, K, splits, w, ws_bytes, st);
        case SWV2_OP_BF16_CSCALE: return launch_tn1<A_BF16_CS>(dy, x, dW, db, nmap, kmap, ldw, M, N, K, splits, w, ws_bytes, st);
    }
    swv2_set_error("swv2_linear_wgrad: unsupported dY operand kind %d", dy->kind);
    return SWV2_ERR_INVALID;
}

extern "C" int swv2_linear_wgrad(const swv2_operand* dy, const swv2_operand* x, float* dW, float* db, const int32_t* nmap,
                                 const int32_t* kmap, int ldw, int splits, void* stream) {
    return swv2_linear_wgrad_ws(dy, x, dW, db, nmap, kmap, ldw, splits, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// grouped launch: the four weight gradients of one block (see gemm_tn_group_kernel)
// ------------------------------------------------------------------------------------------------
namespace {
int group_slices(int total_tiles, int requested) {
    if (requested > 0) return std::max(8, requested / 8 * 8);
    return std::max(8, (2 * 256 / total_tiles) / 8 * 8);          // two workgroups (70 KB of LDS each) per CU, one round
}
}  // namespace

extern "C" size_t swv2_block_wgrad_ws_bytes(int C, int hidden, int heads_dp, int slices) {
    if (C <= 0 || hidden <= 0 || heads_dp <= 0) return 0;
    const int tt = 2 * cdiv(C, BN) * cdiv(hidden, BN) + cdiv(C, BN) * cdiv(heads_dp, BN) + cdiv(3 * heads_dp, BN) * cdiv(C, BN);
    return std::max((size_t)group_slices(tt, slices) * tt * BN * BN * sizeof(float), swv2_tn_slab_ws_bytes(C, hidden, heads_dp));
}

extern "C" int swv2_block_wgrad(const swv2_wgrad_item* it, int slices, void* ws, size_t ws_bytes, void* stream) {
    return swv2_block_wgrad_ln(it, slices, ws, ws_bytes, nullptr, stream);
}

int swv2_block_wgrad_ln(const swv2_wgrad_item* it, int slices, void* ws, size_t ws_bytes, const swv2_ln_partials* ln, void* stream) {
    SWV2_CHECK_ARG(it && ws, "swv2_block_wgrad: null items or workspace");
    static const int want[4][2] = {{SWV2_OP_BF16, SWV2_OP_BF16_GELU}, {SWV2_OP_BF16, SWV2_OP_F32}, {SWV2_OP_BF16, SWV2_OP_HEADS},
                                   {SWV2_OP_HEADS, SWV2_OP_F32}};
    TnGroup g = {};
    int tt = 0;
    for (int i = 0; i < 4; ++i) {
        int rc = check_operand(&it[i].dy, "swv2_block_wgrad(dy)");
        if (rc) return rc;
        rc = check_operand(&it[i].x, "swv2_block_wgrad(x)");
        if (rc) return rc;
        SWV2_CHECK_ARG(it[i].dy.kind == want[i][0] && it[i].x.kind == want[i][1],
                       "swv2_block_wgrad: item %d must be (dY kind %d, X kind %d)", i, want[i][0], want[i][1]);
        SWV2_CHECK_ARG(it[i].dW && it[i].ldw > 0 && it[i].dy.rows == it[i].x.rows, "swv2_block_wgrad: item %d: null dW or row mismatch", i);
        SWV2_CHECK_ARG(!it[i].dy.rowidx && (i == 3 ? it[i].x.rowidx != nullptr : it[i].x.rowidx == nullptr),
                       "swv2_block_wgrad: item %d: only the qkv product's X rows are gathered (and they must be)", i);
        g.first[i] = tt;
        tt += cdiv(it[i].dy.cols, BN) * cdiv(it[i].x.cols, BN);
    }
    g.first[4] = tt;
    // second-generation kernel (gemm_tn_slab.hip: operands fetched once, by LDS-DMA) for the shapes it covers; SWV2_WGRAD_SLAB=0
    // and an explicit slice count select the 128 x 128 tile kernel below
    {
        static const int use_slab = getenv("SWV2_WGRAD_SLAB") ? atoi(getenv("SWV2_WGRAD_SLAB")) : 1;
        if (use_slab && slices <= 0) {
            const int rc = swv2_tn_slab_launch(it, ws, ws_bytes, ln, (hipStream_t)stream);
            if (rc <= 0) return rc;
        }
    }
    const int S = group_slices(tt, slices);
    SWV2_CHECK_ARG(ws_bytes >= (size_t)S * tt * BN * BN * sizeof(float), "swv2_block_wgrad: workspace of %zu bytes, %zu needed",
                   ws_bytes, (size_t)S * tt * BN * BN * sizeof(float));
    // (the tile kernel's reduction carries no rider.  Launched only after every argument check: an error return must not leave the
    // LayerNorm gradients accumulated and the weight gradients not -- ADVICE r4)
    if (ln && ln->C > 0)
        swv2_launch_ln_partials_reduce2(ln->ws[0], ln->dgamma[0], ln->dbeta[0], ln->n[0], ln->ws[1], ln->dgamma[1], ln->dbeta[1], ln->n[1], ln->C,
                                        (hipStream_t)stream);
    TnReduce r = {};
    for (int i = 0; i < 4; ++i) {
        const int tiles = cdiv(it[i].dy.cols, BN) * cdiv(it[i].x.cols, BN);
        g.o[i] = tn_out(it[i].dW, it[i].db, it[i].nmap, it[i].kmap, it[i].ldw, it[i].dy.rows, it[i].dy.cols, it[i].x.cols,
                        (float*)ws + (size_t)g.first[i] * S * BN * BN);
        (void)tiles;
        r.o[i] = g.o[i];
    }
    for (int i = 0; i < 5; ++i) r.first[i] = g.first[i];
    g.y0 = make_loader<A_BF16>(&it[0].dy); g.x0 = make_loader<A_BF16_GELU>(&it[0].x);
    g.y1 = make_loader<A_BF16>(&it[1].dy); g.x1 = make_loader<A_F32>(&it[1].x);
    g.y2 = make_loader<A_BF16>(&it[2].dy); g.x2 = make_loader<A_HEADS>(&it[2].x);
    g.y3 = make_loader<A_HEADS>(&it[3].dy); g.x3 = make_loader<A_F32>(&it[3].x);
    g.slices = r.slices = S;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(tt * S), dim3(NTHREADS), 0, st, g);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(BN * BN / 64, tt), dim3(256), 0, st, r);
    SWV2_CHECK_LAUNCH("swv2_block_wgrad");
    return SWV2_OK;
}
