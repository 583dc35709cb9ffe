// Grouped weight gradients of one transformer block, second generation ("slab" kernel): every operand byte is fetched from
// HBM ONCE per launch and reaches LDS by DMA (global_load_lds_dwordx4), no staging registers, no re-reads through L2.
//
//   dW[N][K] = sum_m dY[m][N]^T X[m][K]   for the block's four products (fc2, fc1, proj, qkv), + bias gradients (column sums of dY)
//
// gemm_tn_group_kernel (gemm_tn.hip) cuts every product into 128 x 128 output tiles: the 3 - 4 tiles that share an operand
// re-read its rows through L2 (1.03 GB into the CUs for 597 MB of algorithmic bytes per block), and the rows pass through
// registers (38 % of a wave's time issuing the loads, 26 % converting + storing them to LDS: profiles/r02_stamps_wgrad_tn.txt).
// Here ONE workgroup (512 threads, one per CU) owns the WHOLE N x K output of its product (or a column half of it where the
// accumulators would not fit) for a contiguous slice of the rows:
//   * a stage = SR rows of dY and of X as LDS slabs, written by DMA (1 KB of LDS per wave instruction, per-lane source address);
//     NS slots, NS - 1 stages in flight (80 - 96 KB per CU), counted vmcnt, one workgroup barrier per stage;
//   * row-major operands keep their rows in LDS, the 32-byte unit u of row r at unit u ^ (r & 7) (8 rows x 32 bytes of a
//     transposed read then fall on 8 different bank groups); head-major operands ([Bw][h][parts][Lp][DP]: oh, d(qkv)) are
//     stored as [16-column block][row][16]: a DMA instruction moves 32 rows x 32 bytes that are contiguous in HBM and in LDS;
//   * both MFMA operands are transposed LDS reads (ds_read_b64_tr_b16), 16x16x32 MFMAs, fp32 accumulators for the whole
//     output in registers (96 - 144 per lane);
//   * fp32 operands (the residual stream x, x1: 4 of the 36 bytes per token-channel) arrive as fp32 slabs by DMA as well and are
//     converted to a bf16 slab in LDS by all 512 threads (8 elements each per stage); gathered rows (the window partition of
//     the qkv product) take their source row from an index table the workgroup loads into LDS once;
//   * GELU of the stored pre-activation (fc2's X operand) is applied to the B fragments in registers through the LDS lookup
//     table of gemm_common.h -- every wave owns its own 64 hidden columns, so no element is looked up twice;
//   * bias gradients: one more MFMA of the dY^T fragment against an all-ones operand (spread over the waves that share it);
//   * the workgroups of the four products run side by side in ONE launch of (number of CUs) workgroups, rows split in proportion
//     to the bytes per row, so every CU streams about the same number of bytes; partial outputs -> workspace -> tn_slab_reduce
//     (fixed order: deterministic).
// HBM-bound: 36 B per token-channel; the floor is the CUs' DMA rate (~24 GB/s per CU, MI355X_MICROARCH "ldsdma-fill").
#include "gemm_common.h"
#pragma clang diagnostic ignored "-Wc++20-extensions"      // explicit template parameter lists of lambdas (compile-time unrolling)

namespace {

// Partial tiles in bf16 (default) or fp32 (-DSWV2_SLAB_PART_F32, A/B builds).  Every workgroup leaves its whole N x K accumulator tile for the
// reduction launch -- 56 MB written and read again per block whatever the batch (24 + 12 us of the two launches, round 5's ablations).  In bf16
// that is half; a partial is a sum over >= 1 000 rows rounded once to 8 bits, the fold over the ~64 - 80 slices stays fp32 and in a fixed order
// (bit-deterministic as before), and the reference's own weight gradients under autocast are bf16 GEMM outputs.
#ifdef SWV2_SLAB_PART_F32
typedef float sl_part_t;
#else
typedef uint16_t sl_part_t;
#endif
constexpr int SL_TH = 512;
constexpr int SL_SMEM = 160 * 1024;           // the whole LDS of a CU: one workgroup per CU
constexpr int SL_IDX_CAP = 4096;              // gather index table in LDS: a ring of two halves (2 x 2048 rows), refilled while the other half is in use
enum { F_ROW = 0, F_ROWG = 1, F_F32 = 2, F_F32G = 3, F_HEAD = 4 };

__device__ __forceinline__ void sl_dma(const void* base, uint32_t byte_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base), "s"(lds_addr) : "memory");
}

// In-kernel phase timing (diagnostic builds only, -DSWV2_SLAB_STAMPS, tools/probe_wgrad_slab.py stamps): wave 0 of every workgroup
// sums s_memtime deltas per phase and overwrites the head of its partial tile with them (the results are garbage then).
#ifdef SWV2_SLAB_STAMPS
#define SSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define SSTAMP(k) do { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define SSTAMP_DECL
#define SSTAMP_START() do {} while (0)
#define SSTAMP(k) do {} while (0)
#endif

// GELU of 8 packed bf16 (a B fragment) through the LDS table of gemm_common.h, which sits at LDS address 0 (the gathers then carry no
// base add: the address is the 16-bit byte offset itself).  Per pair of values: magnitude - table origin (16-bit wrap-around for
// arguments below the table), the running maximum of these indices (ONE range check per fragment instead of a clamp + compare per
// pair), + the sign's half-table offset, two 2-byte gathers.  An index outside the table reads some other LDS bytes; the caller then
// discards the fragment and evaluates the formula (wave-uniform, ~2 % of the fragments on N(0, 1) data).  (The d16 / d16_hi forms of
// the gathers, which would merge the two halves for free, zero the other half of the register on this chip: SRAM-ECC registers.)
__device__ __forceinline__ uint4 sl_gelu8_tab(uint4 w, const unsigned char* tab0, bool& bad) {
    const uint32_t in[4] = {w.x, w.y, w.z, w.w};
    uint32_t out[4], off[4];
    u16x2_t mx = {0, 0};
    const u16x2_t lo = {(unsigned short)GT_LO, (unsigned short)GT_LO};
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u16x2_t idx = __builtin_bit_cast(u16x2_t, in[i] & 0x7fff7fffu) - lo;
        mx = __builtin_elementwise_max(mx, idx);
        const uint32_t sg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, in[i]) >> 15) & (uint32_t)(GT_HALF * 0x10001u);
        off[i] = __builtin_bit_cast(uint32_t, (idx + __builtin_bit_cast(u16x2_t, sg)) << 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        out[i] = (uint32_t)*(const uint16_t*)(tab0 + (off[i] & 0xffffu)) | ((uint32_t)*(const uint16_t*)(tab0 + (off[i] >> 16)) << 16);
    const u16x2_t top = {(unsigned short)(GT_HALF - 1), (unsigned short)(GT_HALF - 1)};
    bad = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(mx, top)) != __builtin_bit_cast(uint32_t, mx);
    return make_uint4(out[0], out[1], out[2], out[3]);
}

struct SlOp {                 // one operand (plain data)
    const void* ptr;
    const int32_t* rowidx;    // F_F32G
    int ld;                   // row-major kinds: row pitch in elements
    int heads, parts, Lp, DP; // F_HEAD
    uint32_t mgLp, mgDP, mgH; // fdiv magics
};
struct SlProd {
    SlOp y, x;
    sl_part_t* part;          // [tile][chunk][slice][256] partial outputs, chunks in accumulator layout (see the epilogue)
    float* dbpart;            // [slices][N] partial bias gradients (null: none)
    int M, N, K;
    int first, wgs, ntile;    // workgroups first .. first + wgs - 1; job j -> (tile j % ntile, slice j / ntile)
};
struct SlArgs { SlProd p[4]; unsigned long long* stamps; };

// physical 32-byte unit of logical unit u in row r of a row-major slab of W bf16 columns
template <int W>
__device__ __forceinline__ int sl_swz(int r, int u) {
    if constexpr ((W * 2) % 256 == 0) return u ^ (r & 7);
    else { static_assert((W * 2) % 128 == 0, "slab width"); return u ^ ((r >> 1) & 3); }
}

template <int FY, int FX, int TN, int TK, int WGN, int WGK, int SR, int NS, bool INPLACE = false>
__device__ __forceinline__ void slab_job(const SlProd& P, const int j, unsigned char* __restrict__ smem, [[maybe_unused]] unsigned long long* stamps) {
    static_assert(WGN * WGK == 8, "8 waves");
    constexpr int IA = TN / (16 * WGN), JB = TK / (16 * WGK);
    static_assert(IA * 16 * WGN == TN && JB * 16 * WGK == TK, "wave tiles");
    constexpr bool XF32 = FX == F_F32 || FX == F_F32G;
    constexpr int YP = TN * 2, XP = TK * 2;                       // bf16 row pitches (row-major slabs)
    constexpr int YB = SR * TN * 2, XB = SR * TK * (XF32 ? 4 : 2), STG = YB + XB;
    constexpr int NY = YB / 1024, NX = XB / 1024, NTOT = NY + NX;
    static_assert(YB % 1024 == 0 && XB % 1024 == 0, "whole DMA instructions");
    // instruction ii = wave + 8 q of a stage (q < NIW): the dY slab's NY instructions first, then the X slab's; when NTOT is not a
    // multiple of 8 the waves below NTOT % 8 issue one more than the others (two counted-vmcnt variants, selected per wave)
    constexpr int NIW = (NTOT + 7) / 8, NREM = NTOT % 8, D = NS - 1;
    // LDS: [GELU table (address 0: see sl_gelu8_tab)] [gather index table] [NS stage slots] [bf16 slab of the current fp32 stage]
    constexpr int OFF_TAB = 0, TABB = FX == F_ROWG ? (GT_N * 2 + 1023) / 1024 * 1024 : 0;
    constexpr int OFF_IDX = OFF_TAB + TABB, IDXB = FX == F_F32G ? SL_IDX_CAP * 4 : 0;
    constexpr int OFF_SLOT = OFF_IDX + IDXB;
    // INPLACE: the bf16 copy of an fp32 stage overwrites the head of its own fp32 slab (one more barrier per stage) where the budget is short
    constexpr int OFF_XB = OFF_SLOT + NS * STG, XBB = (XF32 && !INPLACE) ? SR * TK * 2 : 0;
    static_assert(OFF_XB + XBB <= SL_SMEM, "LDS budget");
    static_assert(!INPLACE || XF32, "in-place conversion: fp32 operands only");
    constexpr int KS = SR / 32;                                   // MFMA k-steps per stage

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wgn = wave / WGK, wgk = wave - wgn * WGK;
    const int n_w = wgn * (TN / WGN), k_w = wgk * (TK / WGK);
    const int ntile = P.ntile, ntk = P.K / TK;
    const int slice = j / ntile, tile = j - slice * ntile, S = P.wgs / ntile;
    const int n0 = (tile / ntk) * TN, k0 = (tile % ntk) * TK;
    const int M = P.M, T = (M + SR - 1) / SR;
    // stages: slice s owns stages s, s + S, s + 2 S, ...: at any moment the S workgroups of a product stream ONE contiguous region of
    // S x SR rows, spread over every HBM channel (contiguous row ranges per slice sit a fixed multiple of the channel interleave
    // apart and hit the same channels at the same time: measured on the first-generation kernel, 2.6 - 3.0 vs 3.6 TB/s)
    const int nst = slice < T ? (T - slice + S - 1) / S : 0;
    auto stage_of = [&](int i) { return slice + i * S; };
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem + OFF_SLOT;          // LDS address of slot 0
    uint16_t* const xbs = (uint16_t*)(smem + OFF_XB);
    uint16_t* const tab = (uint16_t*)(smem + OFF_TAB);
    int32_t* const idxs = (int32_t*)(smem + OFF_IDX);

    // ---- one-time tables (before any DMA is in flight: the compiler's own vmcnt bookkeeping is exact here)
    if constexpr (FX == F_ROWG) {
        for (int i = tid; i < GT_N; i += SL_TH) tab[i] = f2bf(gelu_f(bf2f(gelu_tab_arg(i))));
    }
    // gather table: stage index ti lives at ring position ti % (2 HS); fill_idx(h0) loads the HS stages h0 .. h0 + HS - 1
    constexpr int HS = SL_IDX_CAP / 2 / SR, RING = 2 * HS;
    auto fill_idx = [&](int h0) {
        for (int i = tid; i < HS * SR; i += SL_TH) {
            const int ti = h0 + i / SR;
            const int m = stage_of(ti) * SR + (i % SR);
            idxs[(ti % RING) * SR + (i % SR)] = (ti < nst && m < M) ? P.x.rowidx[m] : -1;
        }
    };
    if constexpr (FX == F_F32G) { fill_idx(0); fill_idx(HS); }
    if constexpr (FX == F_ROWG || FX == F_F32G) __syncthreads();

    // ---- DMA geometry.  Instruction q of this wave is instruction ii = 8 q + wave of the stage: LDS bytes ii * 1024 .. + 1023 of (dY | X)
    int orow[NIW];                            // row inside the stage
    uint32_t ocol[NIW];                       // row-major: element column (incl. n0 / k0); head-major: element offset of the column part
#pragma unroll
    for (int q = 0; q < NIW; ++q) {
        const int ii = 8 * q + wave;          // (wave-uniform)
        if (ii < NY) {
            const int o = ii * 1024 + lane * 16;
            if constexpr (FY == F_ROW) {
                const int row = o / YP, rem = o - row * YP, pu = rem >> 5, u = sl_swz<TN>(row, pu);
                orow[q] = row; ocol[q] = (uint32_t)(n0 + u * 16 + ((rem >> 4) & 1) * 8);
            } else {
                static_assert(FY == F_HEAD, "dY kinds");
                const int blk = o / (SR * 32), rem = o - blk * (SR * 32), row = rem >> 5, col = n0 + blk * 16 + ((rem >> 4) & 1) * 8;
                const int ph = fdiv(col, P.y.DP, P.y.mgDP), jj = col - ph * P.y.DP, part = fdiv(ph, P.y.heads, P.y.mgH), hd = ph - part * P.y.heads;
                orow[q] = row; ocol[q] = (uint32_t)(((hd * P.y.parts + part) * P.y.Lp) * P.y.DP + jj);
            }
        } else {
            const int o = (ii - NY) * 1024 + lane * 16;
            if constexpr (FX == F_ROW || FX == F_ROWG) {
                const int row = o / XP, rem = o - row * XP, pu = rem >> 5, u = sl_swz<TK>(row, pu);
                orow[q] = row; ocol[q] = (uint32_t)(k0 + u * 16 + ((rem >> 4) & 1) * 8);
            } else if constexpr (XF32) {
                const int row = o / (TK * 4), rem = o - row * (TK * 4);
                orow[q] = row; ocol[q] = (uint32_t)(k0 + (rem >> 2));
            } else {
                const int blk = o / (SR * 32), rem = o - blk * (SR * 32), row = rem >> 5, col = k0 + blk * 16 + ((rem >> 4) & 1) * 8;
                const int ph = fdiv(col, P.x.DP, P.x.mgDP), jj = col - ph * P.x.DP, part = fdiv(ph, P.x.heads, P.x.mgH), hd = ph - part * P.x.heads;
                orow[q] = row; ocol[q] = (uint32_t)(((hd * P.x.parts + part) * P.x.Lp) * P.x.DP + jj);
            }
        }
    }
    const uint32_t ywst = FY == F_HEAD ? (uint32_t)(P.y.heads * P.y.parts * P.y.Lp * P.y.DP) : 0u;     // elements per window
    const uint32_t xwst = FX == F_HEAD ? (uint32_t)(P.x.heads * P.x.parts * P.x.Lp * P.x.DP) : 0u;
    // DMA instruction q of this wave for stage index ti of this job into `slot`
    auto issue_q = [&](int ti, int slot, auto qtag) {    // rows past M repeat row M - 1 (dY rows there are zeroed below)
        constexpr int q = decltype(qtag)::value;
        const int ii = 8 * q + wave;
        if (NREM != 0 && q == NIW - 1 && ii >= NTOT) return;        // (only the last instruction of a wave can be missing)
        const int t = stage_of(ti);
        const uint32_t lslot = lds0 + (uint32_t)(slot * STG) + (uint32_t)(ii * 1024);     // (the X slab follows the dY slab: ii * 1024 covers both)
        if (ii < NY) {
            const int m = min(t * SR + orow[q], M - 1);
            uint32_t off;
            if constexpr (FY == F_ROW) off = 2u * ((uint32_t)m * (uint32_t)P.y.ld + ocol[q]);
            else {
                const int bw = fdiv(m, P.y.Lp, P.y.mgLp), tt = m - bw * P.y.Lp;
                off = 2u * ((uint32_t)bw * ywst + (uint32_t)(tt * P.y.DP) + ocol[q]);
            }
            sl_dma(P.y.ptr, off, lslot);
        } else {
            uint32_t off;
            if constexpr (FX == F_F32G) {
                const int r = max(idxs[(ti % RING) * SR + orow[q]], 0);
                off = 4u * ((uint32_t)r * (uint32_t)P.x.ld + ocol[q]);
            } else {
                const int m = min(t * SR + orow[q], M - 1);
                if constexpr (FX == F_ROW || FX == F_ROWG) off = 2u * ((uint32_t)m * (uint32_t)P.x.ld + ocol[q]);
                else if constexpr (FX == F_F32) off = 4u * ((uint32_t)m * (uint32_t)P.x.ld + ocol[q]);
                else {
                    const int bw = fdiv(m, P.x.Lp, P.x.mgLp), tt = m - bw * P.x.Lp;
                    off = 2u * ((uint32_t)bw * xwst + (uint32_t)(tt * P.x.DP) + ocol[q]);
                }
            }
            sl_dma(P.x.ptr, off, lslot);
        }
    };
    // instructions [lo, hi) of a stage: the issue of a stage is spread over the compute of the previous one (insertion points below),
    // so that the CU's load queue is topped up all the time instead of being filled in one burst and drained while the waves compute
    auto issue_range = [&](int ti, int slot, auto lo, auto hi) {
        constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (issue_q(ti, slot, std::integral_constant<int, LO + Q>{}), ...); }
        (std::make_integer_sequence<int, HI - LO>{});
    };
    auto issue = [&](int ti, int slot) { issue_range(ti, slot, std::integral_constant<int, 0>{}, std::integral_constant<int, NIW>{}); };
    constexpr int NP = 1 + JB * KS;                      // insertion points per stage: behind the dY fragments of k-step 0, behind every X fragment
    // point p issues instructions [p NIW / NP, (p + 1) NIW / NP)
    auto issue_point = [&](int ti, int slot, auto ptag) {
        constexpr int p_ = decltype(ptag)::value;
        issue_range(ti, slot, std::integral_constant<int, p_ * NIW / NP>{}, std::integral_constant<int, (p_ + 1) * NIW / NP>{});
    };

    f32x4 acc[IA][JB];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NDB = (IA + WGK - 1) / WGK;
    f32x4 accdb[NDB];
#pragma unroll
    for (int e = 0; e < NDB; ++e) accdb[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool want_db = P.dbpart != nullptr && k0 == 0;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    ALoad<A_BF16_GELU> glf;                 // (tab = null: the formula path of gemm_common.h)
    glf.tab = nullptr;

    // transposed fragment: 16 columns c .. c + 15 (c a multiple of 16) x rows r0 + 4 g .. + 3: lane (g, fr) <- column c + fr
    auto tr_row = [&](const uint16_t* slab, auto wtag, int r0, int c) -> bf16x4 {
        constexpr int W = decltype(wtag)::value;
        const int row = r0 + 4 * g + (fr >> 2);
        return lds_tr_read(slab + row * W + (sl_swz<W>(row, c >> 4) << 4) + (fr & 3) * 4);
    };
    auto tr_head = [&](const uint16_t* slab, int r0, int c) -> bf16x4 {
        const int row = r0 + 4 * g + (fr >> 2);
        return lds_tr_read(slab + (c >> 4) * (SR * 16) + row * 16 + (fr & 3) * 4);
    };

    SSTAMP_DECL
    SSTAMP_START();
    if (nst > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(min(d, nst - 1), d);
        SSTAMP(7);
        for (int t = 0; t < nst; ++t) {
            // all but the (D - 1) youngest stages of this wave have landed
            if (NREM == 0 || wave < NREM) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((D - 1) * NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" : : "n"((D - 1) * (NIW - 1)) : "memory");
            SSTAMP(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();             // stage t has landed (every wave's part); every wave is done with stage t - 1
            __builtin_amdgcn_sched_barrier(0);
            SSTAMP(1);
            const int ti_n = min(t + D, nst - 1), slot_n = (t + D) % NS;
            if constexpr (FX == F_F32G) {
                // every HS stages: the half of the gather table that held stages t - HS .. t - 1 (all consumed: issue and conversion of
                // stage t - 1 lie before this stage's barrier) receives stages t + HS ..; first read HS - D stages (barriers) from now.
                // (plain loads: the compiler waits for them -- and with them, in order, for the DMA in flight -- once per HS stages)
                if (t > 0 && t % HS == 0 && t + HS < nst) fill_idx(t + HS);
            }
            const int slot = t % NS;
            unsigned char* const sb = smem + OFF_SLOT + slot * STG;
            const uint16_t* Ys = (const uint16_t*)sb;
            const uint16_t* Xs = (const uint16_t*)(sb + YB);
            if ((stage_of(t) + 1) * SR > M) {         // ragged last stage: dY rows past M contribute nothing
                for (int c = tid; c < YB / 16; c += SL_TH) {
                    const int row = FY == F_ROW ? (c * 16) / YP : ((c * 16) % (SR * 32)) >> 5;
                    if (stage_of(t) * SR + row >= M) *(uint4*)(sb + c * 16) = make_uint4(0, 0, 0, 0);
                }
                __syncthreads();
            }
#if defined(SWV2_SLAB_ABL) && (SWV2_SLAB_ABL & 1)      // timing ablation (tools/build_variant.sh): pure streaming, wrong results
            issue(ti_n, slot_n);
            continue;
#endif
            if constexpr (XF32) {                     // fp32 slab -> bf16 slab (row-major, swizzled), 8 elements per thread and pass
                constexpr int NCH = SR * TK / 8, NPASS = (NCH + SL_TH - 1) / SL_TH;
                uint16_t* const xdst = INPLACE ? (uint16_t*)(sb + YB) : xbs;
                uint4 cv[NPASS];
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = min(tid + ps * SL_TH, NCH - 1), row = c / (TK / 8), ch = c - row * (TK / 8);
                    const float* src = (const float*)(sb + YB) + row * TK + ch * 8;
                    RawF32 v = {*(const f32x4*)src, *(const f32x4*)(src + 4)};
                    cv[ps] = cvt_f32x8(v);
                    if constexpr (FX == F_F32G) {
                        if (idxs[(t % RING) * SR + row] < 0) cv[ps] = make_uint4(0, 0, 0, 0);
                    }
                }
                if constexpr (INPLACE) __syncthreads();          // every thread has read its fp32 chunks before the slab's head is overwritten
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = tid + ps * SL_TH, row = c / (TK / 8), ch = c - row * (TK / 8);
                    if (c < NCH) *(uint4*)(xdst + row * TK + (sl_swz<TK>(row, ch >> 1) << 4) + (ch & 1) * 8) = cv[ps];
                }
                __syncthreads();
                Xs = xdst;
            }
            SSTAMP(3);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                // Fragments in chunks of at most four along the longer side of the wave tile (the other side, <= 4 fragments, stays in
                // registers for the k-step): with all IA + JB fragments live beside 144 accumulator registers the C = 192 shapes spilled,
                // and a scratch access is a VMEM operation that would break the counted vmcnt waits of the DMA pipeline.
                auto read_a = [&](int i) -> bf16x8 {
                    bf16x4 a0, a1;
                    if constexpr (FY == F_ROW) {
                        a0 = tr_row(Ys, std::integral_constant<int, TN>{}, 32 * kk, n_w + 16 * i);
                        a1 = tr_row(Ys, std::integral_constant<int, TN>{}, 32 * kk + 16, n_w + 16 * i);
                    } else {
                        a0 = tr_head(Ys, 32 * kk, n_w + 16 * i);
                        a1 = tr_head(Ys, 32 * kk + 16, n_w + 16 * i);
                    }
                    return __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                auto read_b = [&](auto jtag) -> bf16x8 {            // + GELU, + this fragment's share of the next stage's DMA issue
                    constexpr int jj = decltype(jtag)::value;
                    bf16x4 b0, b1;
                    if constexpr (FX == F_HEAD) {
                        b0 = tr_head(Xs, 32 * kk, k_w + 16 * jj);
                        b1 = tr_head(Xs, 32 * kk + 16, k_w + 16 * jj);
                    } else {
                        b0 = tr_row(Xs, std::integral_constant<int, TK>{}, 32 * kk, k_w + 16 * jj);
                        b1 = tr_row(Xs, std::integral_constant<int, TK>{}, 32 * kk + 16, k_w + 16 * jj);
                    }
                    bf16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#if !defined(SWV2_SLAB_ABL) || !(SWV2_SLAB_ABL & 2)    // (& 2: timing ablation without the GELU lookups, wrong results)
                    if constexpr (FX == F_ROWG) {
                        const uint4 raw = __builtin_bit_cast(uint4, b);
                        bool bad;
                        uint4 gv = sl_gelu8_tab(raw, smem + OFF_TAB, bad);
                        if (__builtin_expect(__any((int)bad), 0)) gv = glf.cvt(raw);       // formula (no table set on glf)
                        b = __builtin_bit_cast(bf16x8, gv);
                    }
#endif
                    [&]<int... KK>(std::integer_sequence<int, KK...>) {          // (kk is a run-time loop variable of an unrolled loop)
                        ((kk == KK ? issue_point(ti_n, slot_n, std::integral_constant<int, 1 + KK * JB + jj>{}) : (void)0), ...);
                    }(std::make_integer_sequence<int, KS>{});
                    return b;
                };
                if (kk == 0) issue_point(ti_n, slot_n, std::integral_constant<int, 0>{});
                if constexpr (JB <= 4) {
                    bf16x8 bf[JB];
                    [&]<int... JJ>(std::integer_sequence<int, JJ...>) { ((bf[JJ] = read_b(std::integral_constant<int, JJ>{})), ...); }
                    (std::make_integer_sequence<int, JB>{});
                    SSTAMP(4);
#pragma unroll
                    for (int i0 = 0; i0 < IA; i0 += 4) {
                        bf16x8 af[4];
#pragma unroll
                        for (int i = i0; i < i0 + 4 && i < IA; ++i) af[i - i0] = read_a(i);
#pragma unroll
                        for (int i = i0; i < i0 + 4 && i < IA; ++i) {
#pragma unroll
                            for (int jj = 0; jj < JB; ++jj) acc[i][jj] = mfma32(af[i - i0], bf[jj], acc[i][jj]);
                            if (want_db && i % WGK == wgk) accdb[i / WGK] = mfma32(af[i - i0], ones, accdb[i / WGK]);
                        }
                    }
                } else {
                    static_assert(IA <= 4, "one side of the wave tile has at most four fragments");
                    bf16x8 af[IA];
#pragma unroll
                    for (int i = 0; i < IA; ++i) af[i] = read_a(i);
                    if (want_db) {
#pragma unroll
                        for (int i = 0; i < IA; ++i)
                            if (i % WGK == wgk) accdb[i / WGK] = mfma32(af[i], ones, accdb[i / WGK]);
                    }
                    [&]<int... J0>(std::integer_sequence<int, J0...>) { ([&] {
                        constexpr int j0 = 4 * J0;
                        bf16x8 bf[4];
                        [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
                            (((j0 + JJ < JB) ? (void)(bf[JJ] = read_b(std::integral_constant<int, (j0 + JJ < JB ? j0 + JJ : 0)>{})) : (void)0), ...);
                        }(std::make_integer_sequence<int, 4>{});
#pragma unroll
                        for (int jj = j0; jj < j0 + 4 && jj < JB; ++jj)
#pragma unroll
                            for (int i = 0; i < IA; ++i) acc[i][jj] = mfma32(af[i], bf[jj - j0], acc[i][jj]);
                    }(), ...); }(std::make_integer_sequence<int, (JB + 3) / 4>{});
                    SSTAMP(4);
                }
#ifdef SWV2_SLAB_STAMPS
                asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[IA - 1][JB - 1][3]));
#endif
                SSTAMP(5);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-issued tail stages)
    }
    // ---- partial outputs -> workspace in the accumulators' own layout: acc[i][jj] of a wave is one 1 KB chunk (64 lanes x 4 floats), so
    // every store instruction writes 1 KB of contiguous memory with no LDS staging and no wait between the stores (through a row-major
    // staging tile the 54 MB of partials took 20 us of the launch).  Chunk c = (wave IA + i) JB + jj of tile `tile`; the S slices of one
    // chunk lie next to each other ([chunk][slice][256]) so that the reduction reads S contiguous KB per chunk.
#if defined(SWV2_SLAB_ABL) && (SWV2_SLAB_ABL & 4)      // timing ablation: no partial stores, wrong results
    if (M < 0)
#endif
    {
        constexpr int CPT = 8 * IA * JB;             // chunks per tile
        sl_part_t* out = P.part + ((size_t)(tile * CPT + wave * IA * JB) * S + slice) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < IA; ++i)
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
#ifdef SWV2_SLAB_PART_F32
                *(f32x4*)(out + (size_t)(i * JB + jj) * S * 256) = acc[i][jj];
#else
                *(bf16x4*)(out + (size_t)(i * JB + jj) * S * 256) = f2bf4(acc[i][jj]);
#endif
            }
    }
#ifdef SWV2_SLAB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SSTAMP(6);
    if (tid == 0 && stamps) {
#pragma unroll
        for (int k = 0; k < 8; ++k) stamps[(size_t)blockIdx.x * 8 + k] = st_acc[k];
    }
#endif
    if (want_db && fr == 0) {
#pragma unroll
        for (int i = 0; i < IA; ++i)
            if (i % WGK == wgk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) P.dbpart[(size_t)slice * P.N + n0 + n_w + 16 * i + 4 * g + r] = accdb[i / WGK][r];
            }
    }
}

// shape set 0: C = 128, hidden = 512, 8 heads x 16 columns (the BASELINE cfg 2 / 3 / 5 block)
#ifndef SWV2_SLAB_NS
#define SWV2_SLAB_NS 3
#endif
__global__ __launch_bounds__(SL_TH) void gemm_tn_slab_c128_kernel(SlArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SL_SMEM];
    const int b = blockIdx.x;
    if (b < a.p[1].first) slab_job<F_ROW, F_ROWG, 128, 512, 1, 8, 32, SWV2_SLAB_NS>(a.p[0], b - a.p[0].first, smem, a.stamps);          // fc2: d(a2)^T GELU(hpre)
    else if (b < a.p[2].first) slab_job<F_ROW, F_F32, 512, 128, 8, 1, 32, SWV2_SLAB_NS>(a.p[1], b - a.p[1].first, smem, a.stamps);      // fc1: d(h)^T x1
    else if (b < a.p[3].first) slab_job<F_ROW, F_HEAD, 128, 128, 2, 4, 64, SWV2_SLAB_NS + 1>(a.p[2], b - a.p[2].first, smem, a.stamps); // proj: d(a1)^T merge(oh)
    else slab_job<F_HEAD, F_F32G, 384, 128, 8, 1, 32, SWV2_SLAB_NS>(a.p[3], b - a.p[3].first, smem, a.stamps);                          // qkv: d(qkv)^T gather(x)
}
// shape set 1: C = 192, hidden = 768, 8 heads x 32 columns (head dim 24 padded: the BASELINE cfg 4 block).  The outputs of fc2 / fc1 /
// qkv (192 x 768) do not fit one workgroup's registers: two column tiles each (the narrow operand is read twice).
__global__ __launch_bounds__(SL_TH) void gemm_tn_slab_c192_kernel(SlArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SL_SMEM];
    const int b = blockIdx.x;
    if (b < a.p[1].first) slab_job<F_ROW, F_ROWG, 192, 384, 1, 8, 32, 3>(a.p[0], b - a.p[0].first, smem, a.stamps);
    else if (b < a.p[2].first) slab_job<F_ROW, F_F32, 384, 192, 8, 1, 32, 3>(a.p[1], b - a.p[1].first, smem, a.stamps);
    else if (b < a.p[3].first) slab_job<F_ROW, F_HEAD, 192, 256, 2, 4, 32, 4>(a.p[2], b - a.p[2].first, smem, a.stamps);
    else slab_job<F_HEAD, F_F32G, 384, 192, 8, 1, 32, 3, true>(a.p[3], b - a.p[3].first, smem, a.stamps);
}

// dW[nmap(n)][kmap(k)] += sum over the slices of a product's partial chunks; db[nmap(n)] += sum of the partial rows (fixed order).
// One workgroup per 1 KB chunk (64 lanes x 4 floats in accumulator layout): its S slices are S contiguous KB; the eight 64-thread groups
// take every eighth slice and are folded through LDS.
// optional rider of the reduction launch: d gamma / d beta of the block's two LayerNorms from the partial rows their backward kernels
// left (rowops.hip: ln_partials_reduce_kernel as its own launch otherwise): [nblocks][2][C] per set
struct SlLnRed { const float* ws[2]; float* dg[2]; float* db[2]; int n[2]; int C; };
struct SlRed { float* dW; float* db; const int32_t* nmap; const int32_t* kmap; const sl_part_t* part; const float* dbpart;
               int ldw, N, K, TN, TK, WGK, IA, JB, ntk, S, first, dbfirst; };
struct SlRedArgs { SlRed p[4]; int total, dbtotal; SlLnRed ln; int cpw; };      // cpw: 1 KB chunks per workgroup (1: 8 slice groups per chunk; 2: 4 each)
__global__ __launch_bounds__(512) void tn_slab_reduce_kernel(SlRedArgs a) {
    __shared__ f32x4 red[8][64];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    if (b >= a.total + a.dbtotal) {
        // ---- LayerNorm d gamma / d beta: workgroup = 16 columns x 32 slices of partial rows, fixed-order fold (deterministic)
        const int bb = b - a.total - a.dbtotal, per_set = (2 * a.ln.C + 15) / 16;
        const int set = bb / per_set, col = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int jc = (bb - set * per_set) * 16 + col;
        float sum = 0.f;
        if (jc < 2 * a.ln.C) {
            const float* w = a.ln.ws[set];
#pragma unroll 4
            for (int r = sl; r < a.ln.n[set]; r += 32) sum += w[(size_t)r * 2 * a.ln.C + jc];
        }
        float* redf = (float*)red;
        redf[sl * 16 + col] = sum;
        __syncthreads();
        for (int hh = 16; hh > 0; hh >>= 1) {
            if (sl < hh) redf[sl * 16 + col] += redf[(sl + hh) * 16 + col];
            __syncthreads();
        }
        if (sl == 0 && jc < 2 * a.ln.C) {
            if (jc < a.ln.C) a.ln.dg[set][jc] += redf[col]; else a.ln.db[set][jc - a.ln.C] += redf[col];
        }
        return;
    }
    if (b >= a.total) {
        // ---- bias gradients: workgroup = 64 entries x 8 slice groups (one thread adding all S partial rows of an entry is a chain of
        // S dependent-latency loads: it alone took 13 us of the first version's 19)
        const int bb = b - a.total;
        const int pi = (bb >= a.p[1].dbfirst) + (bb >= a.p[2].dbfirst) + (bb >= a.p[3].dbfirst);
        const SlRed& p = a.p[pi];
        const int n = (bb - p.dbfirst) * 64 + lane;
        float a0 = 0.f, a1 = 0.f;
        if (p.db && p.dbpart && n < p.N) {
            int q = sg;
            for (; q + 8 < p.S; q += 16) { a0 += p.dbpart[(size_t)q * p.N + n]; a1 += p.dbpart[(size_t)(q + 8) * p.N + n]; }
            if (q < p.S) a0 += p.dbpart[(size_t)q * p.N + n];
        }
        float* redf = (float*)red;
        redf[sg * 64 + lane] = a0 + a1;
        __syncthreads();
        if (sg == 0 && p.db && p.dbpart && n < p.N) {
            const int nn = p.nmap ? p.nmap[n] : n;
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += redf[e * 64 + lane];
            if (nn >= 0) p.db[nn] += t;
        }
        return;
    }
    // cpw = 2 (few slices per chunk: the C = 192 shape set, ~32): two chunks per workgroup, four slice groups each -- half the
    // workgroups, each with the same number of loads in flight per thread (24 -> 16 us for that shape)
    const int pi = (b >= a.p[1].first) + (b >= a.p[2].first) + (b >= a.p[3].first);
    const SlRed& p = a.p[pi];
    const int NG = 8 / a.cpw;                                   // slice groups per chunk
    const int cg = (b - p.first) * a.cpw + sg / NG, sl0 = sg % NG;
    // destination of this lane's four values (group 0 only), requested before the partial sums so that the index-map and dW
    // round trips overlap the slice loads
    int kk = -1, nn[4] = {-1, -1, -1, -1};
    float old[4] = {0.f, 0.f, 0.f, 0.f};
    if (sl0 == 0) {
        const int cpt = 8 * p.IA * p.JB, tile = cg / cpt, c = cg - tile * cpt;
        const int wave = c / (p.IA * p.JB), ij = c - wave * (p.IA * p.JB), i = ij / p.JB, jj = ij - i * p.JB;
        const int wgn = wave / p.WGK, wgk = wave - wgn * p.WGK, g = lane >> 4, fr = lane & 15;
        const int n0 = (tile / p.ntk) * p.TN + wgn * p.IA * 16 + 16 * i + 4 * g;
        const int k = (tile % p.ntk) * p.TK + wgk * p.JB * 16 + 16 * jj + fr;
        kk = p.kmap ? p.kmap[k] : k;
#pragma unroll
        for (int r = 0; r < 4; ++r) nn[r] = p.nmap ? p.nmap[n0 + r] : n0 + r;
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = p.dW[(long)max(nn[r], 0) * p.ldw + max(kk, 0)];
    }
    const sl_part_t* src = p.part + (size_t)cg * p.S * 256 + lane * 4;
#ifdef SWV2_SLAB_PART_F32
    auto ld = [](const sl_part_t* q) -> f32x4 { return *(const f32x4*)q; };
#else
    auto ld = [](const sl_part_t* q) -> f32x4 {
        const bf16x4 v = *(const bf16x4*)q;
        return (f32x4){bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
    };
#endif
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int s = sl0;
#if defined(SWV2_SLAB_ABL) && (SWV2_SLAB_ABL & 8)      // timing ablation: one slice only (fixed cost of the reduction launch)
    s = p.S;
#endif
    for (; s + 3 * NG < p.S; s += 4 * NG) {
        s0 += ld(src + (size_t)s * 256);
        s1 += ld(src + (size_t)(s + NG) * 256);
        s2 += ld(src + (size_t)(s + 2 * NG) * 256);
        s3 += ld(src + (size_t)(s + 3 * NG) * 256);
    }
    for (; s < p.S; s += NG) s0 += ld(src + (size_t)s * 256);
    red[sg][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl0 == 0 && kk >= 0) {
        f32x4 t;
        if (a.cpw == 1) t = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + ((red[4][lane] + red[5][lane]) + (red[6][lane] + red[7][lane]));
        else t = (red[sg][lane] + red[sg + 1][lane]) + (red[sg + 2][lane] + red[sg + 3][lane]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (nn[r] >= 0) p.dW[(long)nn[r] * p.ldw + kk] = old[r] + t[r];
    }
}

struct SlShape { int TN[4], TK[4], SR[4], WGN[4], WGK[4]; };       // (the template arguments of the kernel's four instantiations)
const SlShape SL_SETS[2] = {{{128, 512, 128, 384}, {512, 128, 128, 128}, {32, 32, 64, 32}, {1, 8, 2, 8}, {8, 1, 4, 1}},
                            {{192, 384, 192, 384}, {384, 192, 256, 192}, {32, 32, 32, 32}, {1, 8, 2, 8}, {8, 1, 4, 1}}};
// block shapes: {C, hidden, heads, head columns}
const int SL_BLOCKS[2][4] = {{128, 512, 8, 16}, {192, 768, 8, 32}};
int sl_set(int C, int hidden, int heads_dp) {
    for (int i = 0; i < 2; ++i)
        if (C == SL_BLOCKS[i][0] && hidden == SL_BLOCKS[i][1] && heads_dp == SL_BLOCKS[i][2] * SL_BLOCKS[i][3]) return i;
    return -1;
}

// workgroups of the launch = CUs of the current device, at most SL_CU_BOUND: the workspace SIZE (swv2_tn_slab_ws_bytes, a pure host
// function: no HIP call, no dependence on the thread's current device -- ADVICE r4) is computed for the bound, so a workspace sized on
// any device / before set_device covers every launch
constexpr int SL_CU_BOUND = 320;
int sl_cus() {
    static thread_local int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v < SL_CU_BOUND ? v : SL_CU_BOUND;
    }
    return cus[dev];
}

struct SlPlan { bool ok; int set; int wgs[4], ntile[4], S[4]; size_t part_off[4], db_off[4], total; };

// it: the four items of swv2_block_wgrad (kinds already checked by the caller)
SlPlan sl_plan(const swv2_wgrad_item* it, int cus) {
    SlPlan p = {};
    const int N[4] = {it[0].dy.cols, it[1].dy.cols, it[2].dy.cols, it[3].dy.cols};
    const int K[4] = {it[0].x.cols, it[1].x.cols, it[2].x.cols, it[3].x.cols};
    const int C = N[0], hid = K[0];
    p.set = sl_set(C, hid, K[2]);
    if (p.set < 0) return p;
    const SlShape& sh = SL_SETS[p.set];
    if (!(N[1] == hid && K[1] == C && N[2] == C && N[3] == 3 * K[2] && K[3] == C)) return p;
    if (!(it[2].x.p[3] == SL_BLOCKS[p.set][3] && it[3].dy.p[3] == SL_BLOCKS[p.set][3] && it[2].x.p[0] == SL_BLOCKS[p.set][2] &&
          it[3].dy.p[0] == SL_BLOCKS[p.set][2])) return p;
    if (!(it[0].dy.ld == C && it[0].x.ld == hid && it[1].dy.ld == hid && it[1].x.ld == C && it[2].dy.ld == C && it[3].x.ld == C)) return p;
    if (cus < 16 || cus > 1024) return p;
    // bytes per row of each product; rows in proportion so that every workgroup streams about the same number of bytes
    double cost[4], tot = 0;
    // measured weights on top of the byte counts (tools/probe_wgrad_slab.py, SWV2_SLAB_WEIGHTS sweep): the fc2 workgroups carry the
    // GELU lookups (LDS gathers) and need ~1.5 x the time per byte of the others; the optimum is flat (+- 2 us from 1.4 to 1.7)
    double wgt[4] = {1.55, 1.0, 1.0, 1.1};
    if (const char* e = getenv("SWV2_SLAB_WEIGHTS")) sscanf(e, "%lf,%lf,%lf,%lf", &wgt[0], &wgt[1], &wgt[2], &wgt[3]);
    for (int i = 0; i < 4; ++i) {
        const double xb = it[i].x.kind == SWV2_OP_F32 ? 4.0 : 2.0;
        p.ntile[i] = (N[i] / sh.TN[i]) * (K[i] / sh.TK[i]);
        cost[i] = wgt[i] * (double)it[i].dy.rows * (2.0 * N[i] * (K[i] / sh.TK[i]) + xb * K[i] * (N[i] / sh.TN[i]));
        tot += cost[i];
    }
    int used = 0;
    for (int i = 0; i < 4; ++i) {
        const int T = cdiv(it[i].dy.rows, sh.SR[i]);
        int sl = (int)(cus * cost[i] / tot / p.ntile[i] + 0.5);          // row slices; every slice runs ntile workgroups
        sl = std::max(1, std::min(sl, T));
        p.S[i] = sl;
        p.wgs[i] = sl * p.ntile[i];
        used += p.wgs[i];
    }
    // hand the rounding remainder to / take it from the products with the most rows per slice (one tile each in this shape set)
    for (int guard = 0; used != cus && guard < 64; ++guard) {
        int best = -1;
        double br = 0;
        for (int i = 0; i < 4; ++i) {
            const int T = cdiv(it[i].dy.rows, sh.SR[i]);
            if (used < cus ? (p.S[i] >= T || used + p.ntile[i] > cus) : p.S[i] <= 1) continue;
            const double r = cost[i] / p.S[i];
            if (best < 0 || (used < cus ? r > br : r < br)) { best = i; br = r; }
        }
        if (best < 0) break;
        const int d = used < cus ? 1 : -1;
        p.S[best] += d; p.wgs[best] += d * p.ntile[best]; used += d * p.ntile[best];
    }
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
        const int T = cdiv(it[i].dy.rows, sh.SR[i]);
        if ((double)it[i].dy.rows * std::max(N[i], K[i]) * 4.0 >= 4.29e9) return p;         // 32-bit byte offsets
        p.part_off[i] = off;
        off += (size_t)p.wgs[i] * sh.TN[i] * sh.TK[i] * sizeof(float);
        p.db_off[i] = off;
        off += (size_t)p.S[i] * N[i] * sizeof(float);
        off = (off + 255) / 256 * 256;
    }
    p.total = off;
    p.ok = true;
    return p;
}

SlOp sl_op(const swv2_operand& o) {
    SlOp s = {};
    s.ptr = o.ptr; s.rowidx = o.rowidx; s.ld = (int)o.ld;
    if (o.kind == SWV2_OP_HEADS) {
        s.heads = o.p[0]; s.parts = (int)o.ld; s.Lp = o.p[2]; s.DP = o.p[3];
        s.mgLp = fdiv_magic(s.Lp); s.mgDP = fdiv_magic(s.DP); s.mgH = fdiv_magic(s.heads);
    }
    return s;
}

}  // namespace

// upper bound of the slab path's workspace for this block shape on any device (0: shape not covered): the workgroups of all four
// products together number at most one per CU (<= SL_CU_BOUND), each with at most the largest partial tile, + the partial bias rows
size_t swv2_tn_slab_ws_bytes(int C, int hidden, int heads_dp) {
    const int set = sl_set(C, hidden, heads_dp);
    if (set < 0) return 0;
    const int cus = SL_CU_BOUND;
    const SlShape& sh = SL_SETS[set];
    size_t tile = 0;
    for (int i = 0; i < 4; ++i) tile = std::max(tile, (size_t)sh.TN[i] * sh.TK[i] * 4);
    return (size_t)cus * tile + (size_t)cus * std::max(hidden, 3 * heads_dp) * 4 + 4 * 256;
}

// 0: launched; 1: shape / workspace not covered (the caller takes the 128 x 128 tile kernel); < 0: error code
int swv2_tn_slab_launch(const swv2_wgrad_item* it, void* ws, size_t ws_bytes, const swv2_ln_partials* ln, hipStream_t st) {
    const int cus = sl_cus();
    const SlPlan pl = sl_plan(it, cus);
    if (pl.ok && ws_bytes < pl.total) {
        static int once = 0;
        if (!once++) fprintf(stderr, "swv2: grouped weight gradient declined (workspace %zu bytes, %zu needed): size it with swv2_block_wgrad_ws_bytes; "
                                     "running the 128 x 128 tile kernels\n", ws_bytes, pl.total);
    }
    if (!pl.ok || ws_bytes < pl.total) return 1;
    const SlShape& sh = SL_SETS[pl.set];
    SlArgs a = {};
    SlRedArgs r = {};
    int first = 0, rfirst = 0, dbfirst = 0;
    // two 1 KB chunks per reduction workgroup where every product has few slices and an even number of chunks
    int cpw = 2;
    for (int i = 0; i < 4; ++i)
        if (pl.S[i] > 40 || ((it[i].dy.cols * it[i].x.cols / 256) & 1)) cpw = 1;
    if (const char* e = getenv("SWV2_SLAB_RED_CPW")) cpw = atoi(e) == 2 ? 2 : 1;
    for (int i = 0; i < 4 && cpw == 2; ++i)
        if ((it[i].dy.cols * it[i].x.cols / 256) & 1) cpw = 1;
    r.cpw = cpw;
    for (int i = 0; i < 4; ++i) {
        SlProd& p = a.p[i];
        p.y = sl_op(it[i].dy); p.x = sl_op(it[i].x);
        p.part = (sl_part_t*)((char*)ws + pl.part_off[i]);
        p.dbpart = it[i].db ? (float*)((char*)ws + pl.db_off[i]) : nullptr;
        p.M = it[i].dy.rows; p.N = it[i].dy.cols; p.K = it[i].x.cols;
        p.first = first; p.wgs = pl.wgs[i]; p.ntile = pl.ntile[i];
        first += pl.wgs[i];
        SlRed& q = r.p[i];
        q.dW = it[i].dW; q.db = it[i].db; q.nmap = it[i].nmap; q.kmap = it[i].kmap; q.part = p.part; q.dbpart = p.dbpart;
        q.ldw = it[i].ldw; q.N = p.N; q.K = p.K; q.TN = sh.TN[i]; q.TK = sh.TK[i]; q.WGK = sh.WGK[i];
        q.IA = sh.TN[i] / (16 * sh.WGN[i]); q.JB = sh.TK[i] / (16 * sh.WGK[i]); q.ntk = p.K / sh.TK[i]; q.S = pl.S[i]; q.first = rfirst;
        rfirst += p.N * p.K / 256 / cpw;
        q.dbfirst = dbfirst;
        dbfirst += cdiv(p.N, 64);
    }
    r.total = rfirst;
    r.dbtotal = dbfirst;
    int lnblocks = 0;
    if (ln && ln->C > 0) {
        for (int i = 0; i < 2; ++i) { r.ln.ws[i] = ln->ws[i]; r.ln.dg[i] = ln->dgamma[i]; r.ln.db[i] = ln->dbeta[i]; r.ln.n[i] = ln->n[i]; }
        r.ln.C = ln->C;
        lnblocks = 2 * cdiv(2 * ln->C, 16);
    }
#ifdef SWV2_SLAB_STAMPS       // the last 64 bytes x workgroups of the workspace receive the phase sums; the split goes to stderr once
    if (ws_bytes >= pl.total + (size_t)first * 64) a.stamps = (unsigned long long*)((char*)ws + ws_bytes - (size_t)first * 64);
    { static int once = 0; if (!once++) fprintf(stderr, "slab plan: workgroups %d %d %d %d\n", pl.wgs[0], pl.wgs[1], pl.wgs[2], pl.wgs[3]); }
#endif
    if (pl.set == 0) hipLaunchKernelGGL(gemm_tn_slab_c128_kernel, dim3(first), dim3(SL_TH), 0, st, a);
    else hipLaunchKernelGGL(gemm_tn_slab_c192_kernel, dim3(first), dim3(SL_TH), 0, st, a);
    hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3(rfirst + dbfirst + lnblocks), dim3(512), 0, st, r);
    SWV2_CHECK_LAUNCH("swv2_block_wgrad(slab)");
    return 0;
}
