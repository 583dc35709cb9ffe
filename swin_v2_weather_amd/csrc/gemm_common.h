// Shared pieces of the GEMM engine: tile constants, LDS swizzle, GELU, bf16 packing, operand loaders, host checks.
// Included by gemm.hip (NT products + epilogues) and gemm_tn.hip (weight-gradient products).
#pragma once
#include "common.h"

// gemm.hip: the wide weight-gradient kernel (partial matrices), launched by gemm_tn.hip
int swv2_tn_wide_launch(const swv2_operand* y, const swv2_operand* x, float* part, float* dbpart, int M, int N, int K, int S, hipStream_t st);

// gemm_tn_slab.hip: the block's four weight gradients with every operand byte fetched once (LDS-DMA slabs); launch returns 1 when
// the shape / workspace is not covered (the caller then takes gemm_tn_group_kernel)
size_t swv2_tn_slab_ws_bytes(int C, int hidden, int heads_dp);
int swv2_tn_slab_launch(const swv2_wgrad_item* it, void* ws, size_t ws_bytes, const swv2_ln_partials* ln, hipStream_t st);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KCH = BK / 8;                 // 16-byte chunks per tile row
constexpr int NTHREADS = 256;

// swizzled element offset of chunk kc of row r in a [rows][64] bf16 tile (128-byte rows)
__device__ __forceinline__ int swz(int r, int kc) { return r * BK + ((kc ^ ((r >> 1) & 7)) << 3); }

// erf-GELU (timm Mlp act = nn.GELU) with the Abramowitz-Stegun 7.1.26 rational erf (|err| < 1.5e-7, far below the
// bf16 storage precision): one v_rcp + one v_exp + a few FMAs instead of the ~40-instruction erff().
__device__ __forceinline__ void erf_parts(float x, float& erf_v, float& gauss) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    gauss = __expf(-z * z);                                    // exp(-x^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = fmaf(-poly, gauss, 1.f);
    erf_v = copysignf(e, x);
}
// exact unsigned division by a runtime-constant divisor d with magic = ceil(2^32 / d) (host: fdiv_magic): the estimate
// umulhi(n, magic) is q or q + 1 for every n < 2^32, one compare fixes it.  Replaces ~25-instruction integer divides
// in the loaders' index math (row -> (window, token), column -> (part, head)).
__device__ __forceinline__ int fdiv(int n, int d, uint32_t magic) {
    int q = (int)__umulhi((uint32_t)n, magic);
    return q - (q * d > n);
}
static inline uint32_t fdiv_magic(int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u; }

__device__ __forceinline__ float gelu_f(float x) {
    float e, gs;
    erf_parts(x, e, gs);
    return 0.5f * x * (1.f + e);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float e, gs;
    erf_parts(x, e, gs);
    return fmaf(x * 0.3989422804014327f, gs, 0.5f * (1.f + e));
}

// ------------------------------------------------------------------------------------------------
// Table-driven GELU for bf16 inputs.  The argument of every GELU / GELU' on this path is a STORED bf16 pre-activation, so
// the function has at most 65536 distinct arguments; for |x| in [2^-14, 2^4) (18 binades x 128 mantissas x 2 signs =
// 4608 entries) the value is looked up in an LDS table that each workgroup fills once with the formulas above -- the
// results are bit-identical to evaluating them per element (~14 vector-issue slots incl. two transcendentals), at ~5
// slots + one 2-byte LDS gather.  Arguments outside the table (|x| < 6.1e-5: probability ~5e-5; |x| >= 16) make the
// caller fall back to the formula for that group of values (wave-uniform branch).
// ------------------------------------------------------------------------------------------------
constexpr int GT_LO = (127 - 14) << 7;       // bf16 bits of 2^-14
constexpr int GT_HALF = 18 * 128;            // entries per sign
constexpr int GT_N = 2 * GT_HALF;
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

// entry i <-> bf16 bits ((i / GT_HALF) << 15) | (GT_LO + i % GT_HALF)
__device__ __forceinline__ uint16_t gelu_tab_arg(int i) { return (uint16_t)(((i / GT_HALF) << 15) | (GT_LO + i % GT_HALF)); }

// two packed bf16 arguments -> byte offsets (packed u16 pair) into a table of `ESZ`-byte entries; `bad` is set when either
// argument is outside the table (the offsets are then clamped and the caller must not use the looked-up values)
template <int ESZ>
__device__ __forceinline__ uint32_t gelu_tab_off2(uint32_t w, bool& bad) {
    const u16x2_t a = __builtin_bit_cast(u16x2_t, w & 0x7fff7fffu);
    const u16x2_t lo = {(unsigned short)GT_LO, (unsigned short)GT_LO};
    const u16x2_t mx = {(unsigned short)(GT_HALF - 1), (unsigned short)(GT_HALF - 1)};
    const u16x2_t idx = a - lo;                                      // wraps for arguments below the table
    const u16x2_t idc = __builtin_elementwise_min(idx, mx);
    bad = bad || (__builtin_bit_cast(uint32_t, idx) != __builtin_bit_cast(uint32_t, idc));
    const u16x2_t sg = __builtin_bit_cast(u16x2_t, w) >> 15;         // 0 / 1 per half
    const u16x2_t half = {(unsigned short)GT_HALF, (unsigned short)GT_HALF};
    const u16x2_t off = (idc + sg * half) * (unsigned short)ESZ;
    return __builtin_bit_cast(uint32_t, off);
}

// the same for a table that holds the POSITIVE arguments only (GT_HALF entries): offsets of |x|; the caller applies the sign symmetry
template <int ESZ>
__device__ __forceinline__ uint32_t gelu_tab_off2_abs(uint32_t w, bool& bad) {
    const u16x2_t a = __builtin_bit_cast(u16x2_t, w & 0x7fff7fffu);
    const u16x2_t lo = {(unsigned short)GT_LO, (unsigned short)GT_LO};
    const u16x2_t mx = {(unsigned short)(GT_HALF - 1), (unsigned short)(GT_HALF - 1)};
    const u16x2_t idx = a - lo;
    const u16x2_t idc = __builtin_elementwise_min(idx, mx);
    bad = bad || (__builtin_bit_cast(uint32_t, idx) != __builtin_bit_cast(uint32_t, idc));
    const u16x2_t off = idc * (unsigned short)ESZ;
    return __builtin_bit_cast(uint32_t, off);
}

__device__ __forceinline__ uint4 pack8(const float* v) {
    uint4 r;
    r.x = f2bf2(v[0], v[1]); r.y = f2bf2(v[2], v[3]); r.z = f2bf2(v[4], v[5]); r.w = f2bf2(v[6], v[7]);
    return r;
}
__device__ __forceinline__ void unpack8(uint4 c, float* v) {
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

// ------------------------------------------------------------------------------------------------
// A loaders: chunk(m, k0) returns 8 bf16 (k0 multiple of 8) of logical row m; zeros outside [0,M)x[0,K)
// ------------------------------------------------------------------------------------------------
struct LoadDesc {           // plain-data description shared by all loader kinds (filled by the host)
    const void* ptr;        // primary source
    const int32_t* rowidx;  // optional gather table: logical row -> source row, <0 = zero row
    const float* aux0;      // LN-on-load: mean[M] ; patch: unused
    const float* aux1;      // LN-on-load: rstd[M]
    const float* aux2;      // LN-on-load: gamma[K]
    const float* aux3;      // LN-on-load: beta[K]
    long ld;                // source row pitch in elements
    int M, K;
    int p0, p1, p2, p3;     // kind-specific ints (see loaders)
    uint32_t mg0, mg1, mg2; // fdiv magics of the kind-specific divisors (make_loader)
};

template <int KIND> struct ALoad;

enum { A_F32 = 0, A_BF16 = 1, A_BF16_GELU = 2, A_HEADS = 3, A_PATCH = 4, A_MERGE_LN = 5, A_BF16_CS = 6 };

struct RawF32 { f32x4 a, b; };
__device__ __forceinline__ uint4 cvt_f32x8(const RawF32& r) {
    uint4 o;
    o.x = f2bf2(r.a[0], r.a[1]); o.y = f2bf2(r.a[2], r.a[3]); o.z = f2bf2(r.b[0], r.b[1]); o.w = f2bf2(r.b[2], r.b[3]);
    return o;
}
// Every loader has two phases: raw() issues the global loads and returns the registers untouched, cvt() produces the
// 8 bf16.  The conversion happens when the tile is written to LDS, so the loads stay in flight across the MFMA phase
// (a loader that converts inside the load makes the compiler wait for the data right there).  chunk() = cvt(raw()).

// fp32 rows (optionally gathered): the residual stream x[B*T][C]
template <> struct ALoad<A_F32> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = true;      // un-gathered rows are ptr + row * ld: the kernels may form the offsets incrementally
    typedef RawF32 Raw;
    LoadDesc d;
    // element offset (row * ld + k0, below 2^32: checked by the launchers) -> unconditional load, scalar base + 32-bit offset
    __device__ __forceinline__ Raw raw_lin(uint32_t off) const {
        const float* p = (const float*)d.ptr + off;
        Raw o = {*(const f32x4*)p, *(const f32x4*)(p + 4)};
        return o;
    }
    // source row of logical row m (-1 = zero row): ONE dependent load for gathered operands, hoisted by the kernels
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : (d.rowidx ? d.rowidx[m] : m); }
    __device__ __forceinline__ Raw raw_at(int r, int k0) const {
        Raw z = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (r < 0 || k0 >= d.K) return z;
        const float* p = (const float*)d.ptr + (long)r * d.ld + k0;
        Raw o = {*(const f32x4*)p, *(const f32x4*)(p + 4)};
        return o;
    }
    __device__ __forceinline__ Raw raw(int m, int k0) const { return raw_at(row_of(m), k0); }
    // UNCONDITIONAL load of source row r in [0, rows), k0 in [0, K): no exec-masked load, so the compiler can count it in
    // s_waitcnt and a prefetch stays in flight (a load under `if` makes it guard the destination with vmcnt(0))
    __device__ __forceinline__ Raw raw_unc(int r, int k0) const {
        const float* p = (const float*)d.ptr + (long)r * d.ld + k0;
        Raw o = {*(const f32x4*)p, *(const f32x4*)(p + 4)};
        return o;
    }
    __device__ __forceinline__ uint4 cvt(const Raw& r) const { return cvt_f32x8(r); }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const { return cvt(raw(m, k0)); }
};
// bf16 rows (optionally gathered)
template <> struct ALoad<A_BF16> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = true;
    typedef uint4 Raw;
    LoadDesc d;
    __device__ __forceinline__ Raw raw_lin(uint32_t off) const { return *(const uint4*)((const uint16_t*)d.ptr + off); }
    __device__ __forceinline__ uint4 cvt(const Raw& r) const { return r; }
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : (d.rowidx ? d.rowidx[m] : m); }
    __device__ __forceinline__ Raw raw_at(int r, int k0) const {
        if (r < 0 || k0 >= d.K) return make_uint4(0, 0, 0, 0);
        return *(const uint4*)((const uint16_t*)d.ptr + (long)r * d.ld + k0);
    }
    __device__ __forceinline__ Raw raw_unc(int r, int k0) const { return *(const uint4*)((const uint16_t*)d.ptr + (long)r * d.ld + k0); }
    __device__ __forceinline__ Raw raw(int m, int k0) const { return raw_at(row_of(m), k0); }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const { return raw(m, k0); }
};
// bf16 rows scaled on load by an fp32 factor per (sample, 16-column group): the head's loss-gradient operand
//   G[m][n] = coef[b(m)][n / 16] * R[m][n],  R = quadrature-weighted residual written by the un-patchify + loss epilogue
// p0 = rows per sample (mg0 its magic), p2 = groups per sample in aux0.  The factor is loaded with the data (raw), applied
// when the tile is written to LDS (cvt).
struct RawCS { uint4 v; float s; };
template <> struct ALoad<A_BF16_CS> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = false;
    typedef RawCS Raw;
    LoadDesc d;
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : m; }
    __device__ __forceinline__ Raw raw_unc(int r, int k0) const {
        const int b = fdiv(r, d.p0, d.mg0);
        Raw o = {*(const uint4*)((const uint16_t*)d.ptr + (long)r * d.ld + k0), d.aux0[b * d.p2 + (k0 >> 4)]};
        return o;
    }
    __device__ __forceinline__ Raw raw_at(int r, int k0) const {
        Raw z = {make_uint4(0, 0, 0, 0), 0.f};
        if (r < 0 || k0 >= d.K) return z;
        return raw_unc(r, k0);
    }
    __device__ __forceinline__ Raw raw(int m, int k0) const { return raw_at(row_of(m), k0); }
    __device__ __forceinline__ uint4 cvt(const Raw& r) const {
        float v[8];
        unpack8(r.v, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= r.s;
        return pack8(v);
    }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const { return cvt(raw(m, k0)); }
};
// bf16 rows through GELU (fc2 input = GELU(fc1 output); the pre-activation is what is kept for backward)
template <> struct ALoad<A_BF16_GELU> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = true;
    __device__ __forceinline__ uint4 raw_lin(uint32_t off) const { return *(const uint4*)((const uint16_t*)d.ptr + off); }
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : m; }
    __device__ __forceinline__ auto raw_at(int r, int k0) const { return raw(r < 0 ? d.M : r, k0); }
    typedef uint4 Raw;
    LoadDesc d;
    __device__ __forceinline__ Raw raw(int m, int k0) const {
        if (m >= d.M || k0 >= d.K) return make_uint4(0, 0, 0, 0);
        return *(const uint4*)((const uint16_t*)d.ptr + (long)m * d.ld + k0);
    }
    __device__ __forceinline__ Raw raw_unc(int r, int k0) const { return *(const uint4*)((const uint16_t*)d.ptr + (long)r * d.ld + k0); }
    const uint16_t* tab = nullptr;      // LDS table of bf16(GELU(x)) (see gelu_tab_off2), set by the kernel; null = formula
    __device__ __forceinline__ uint4 cvt(const Raw& r) const {
#ifdef SWV2_TN_GELU_ABL                      // timing ablation only (tools/ab_macro.sh): no GELU, wrong results
        return r;
#endif
        if (tab) {
            bool bad = false;
            const uint32_t o[4] = {gelu_tab_off2<2>(r.x, bad), gelu_tab_off2<2>(r.y, bad), gelu_tab_off2<2>(r.z, bad),
                                   gelu_tab_off2<2>(r.w, bad)};
            if (!__builtin_expect(__any((int)bad), 0)) {
                const unsigned char* tb = (const unsigned char*)tab;
                uint32_t q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    q[i] = (uint32_t)*(const uint16_t*)(tb + (o[i] & 0xffffu)) | ((uint32_t)*(const uint16_t*)(tb + (o[i] >> 16)) << 16);
                return make_uint4(q[0], q[1], q[2], q[3]);
            }
        }
        float v[8];
        unpack8(r, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_f(v[i]);
        return pack8(v);
    }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const { return cvt(raw(m, k0)); }
};
// head-major window layout [Bw][h][S][Lp][DP] -> logical row m = bw*Lp + t, logical k = (part*h + head)*DP + j
// (head dim padded to DP in {16, 32, 64, 96, 128}; the matching weights are padded by swv2_prep_weight).
// p0 = heads, p2 = Lp, p3 = DP ; ld = number of parts S (1 for oh, 3 for dqkvh)
template <> struct ALoad<A_HEADS> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = false;
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : m; }
    __device__ __forceinline__ auto raw_at(int r, int k0) const { return raw(r < 0 ? d.M : r, k0); }
    typedef uint4 Raw;
    LoadDesc d;
    __device__ __forceinline__ uint4 cvt(const Raw& r) const { return r; }
    __device__ __forceinline__ Raw raw(int m, int k0) const { return chunk(m, k0); }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const {
        if (m >= d.M || k0 >= d.K) return make_uint4(0, 0, 0, 0);
        return raw_unc(m, k0);
    }
    __device__ __forceinline__ uint4 raw_unc(int m, int k0) const {
        const int h = d.p0, Lp = d.p2, DP = d.p3, S = (int)d.ld;
        const int bw = fdiv(m, Lp, d.mg0), t = m - bw * Lp;
        const int ph = d.p1 >= 0 ? (k0 >> d.p1) : fdiv(k0, DP, d.mg2), j = k0 - ph * DP;      // p1 = log2(DP) or -1 (DP = 96), set by make_loader
        const int part = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - part * h;
        return *(const uint4*)((const uint16_t*)d.ptr + ((((long)bw * h + hd) * S + part) * Lp + t) * DP + j);
    }
    // element offset of (row m, column k0) for callers that address with 32 bits (tensor below 2^32 bytes, checked by them)
    __device__ __forceinline__ uint32_t elem_off(int m, int k0) const {
        const int h = d.p0, Lp = d.p2, DP = d.p3, S = (int)d.ld;
        const int bw = fdiv(m, Lp, d.mg0), t = m - bw * Lp;
        const int ph = d.p1 >= 0 ? (k0 >> d.p1) : fdiv(k0, DP, d.mg2), j = k0 - ph * DP;
        const int part = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - part * h;
        return (uint32_t)((((bw * h + hd) * S + part) * Lp + t) * DP + j);
    }
    // Walking rows m, m + 16, m + 32, ... of ONE chunk column (the weight-gradient kernel's staging): the window / token split
    // once per step (step_base), then adds -- a row block that runs past its window's Lp rows continues (h S - 1) Lp DP
    // elements further, in the same (head, part) slab of the next window.  Valid while a step spans at most one window
    // boundary (16 * chunks <= Lp, checked by the caller); 32-bit element offsets.
    struct Step { uint32_t off0; int t0; };
    __device__ __forceinline__ Step step_base(int m, int k0) const {
        const int h = d.p0, Lp = d.p2, DP = d.p3, S = (int)d.ld;
        const int bw = fdiv(m, Lp, d.mg0), t = m - bw * Lp;
        const int ph = d.p1 >= 0 ? (k0 >> d.p1) : fdiv(k0, DP, d.mg2), j = k0 - ph * DP;
        const int part = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - part * h;
        Step o = {(uint32_t)((((bw * h + hd) * S + part) * Lp + t) * DP + j), t};
        return o;
    }
    __device__ __forceinline__ uint32_t step_off(const Step& b, int i) const {
        const int Lp = d.p2, DP = d.p3;
        const uint32_t hop = (uint32_t)((d.p0 * (int)d.ld - 1) * Lp * DP);
        return b.off0 + (uint32_t)(16 * i * DP) + ((b.t0 + 16 * i >= Lp) ? hop : 0u);
    }
    __device__ __forceinline__ uint4 raw_lin(uint32_t off) const { return *(const uint4*)((const uint16_t*)d.ptr + off); }
};
// PatchEmbed im2col: x[B][Cin][H][W] fp32, row m = (b, i, j) patch, k = cin*16 + p*4 + q (conv weight order)
// p0 = Cin, p1 = H, p2 = W ; patch = 4.  Adjacent rows are adjacent 16-byte groups -> row-fastest thread map.
template <> struct ALoad<A_PATCH> {
    static constexpr bool ROW_FASTEST = true;
    static constexpr bool LINEAR = false;
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : m; }
    __device__ __forceinline__ auto raw_at(int r, int k0) const { return raw(r < 0 ? d.M : r, k0); }
    typedef RawF32 Raw;
    LoadDesc d;
    __device__ __forceinline__ Raw raw(int m, int k0) const {
        Raw z = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (m >= d.M || k0 >= d.K) return z;
        return raw_unc(m, k0);
    }
    __device__ __forceinline__ Raw raw_unc(int m, int k0) const {
        const int Cin = d.p0, H = d.p1, W = d.p2, gw = W >> 2, gh = H >> 2;
        const int b = fdiv(m, gh * gw, d.mg0), ij = m - b * gh * gw, i = fdiv(ij, gw, d.mg1), j = ij - i * gw;
        const int cin = k0 >> 4, p = (k0 >> 2) & 3;                 // p in {0, 2}
        // p3 = channels per sample of the tensor that holds the Cin planes (0 = Cin: a dense [B][Cin][H][W] tensor); the
        // rollout's step outputs are channel slices of one [B][(n_future+1) Cout][H][W] buffer
        const int Ct = d.p3 ? d.p3 : Cin;
        const float* src = (const float*)d.ptr + (((long)b * Ct + cin) * H + 4 * i + p) * W + 4 * j;
        Raw o = {*(const f32x4*)src, *(const f32x4*)(src + W)};
        if (d.aux0) {   // second source added on load (ld = its channels per sample): gradient of the fed-back prediction
            const float* s2 = d.aux0 + (((long)b * d.ld + cin) * H + 4 * i + p) * W + 4 * j;
            o.a += *(const f32x4*)s2;
            o.b += *(const f32x4*)(s2 + W);
        }
        return o;
    }
    __device__ __forceinline__ uint4 cvt(const Raw& r) const { return cvt_f32x8(r); }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const { return cvt(raw(m, k0)); }
};
// PatchMerging gather + LayerNorm(4C) on load: x[B][H][W][C] fp32, row m = (b, i, j) on the half grid,
// k = (wp*2 + hp)*C + c  (swinv2_global.py:520) ; p0 = H, p1 = W, p2 = C ; aux = mean, rstd, gamma, beta
template <> struct ALoad<A_MERGE_LN> {
    static constexpr bool ROW_FASTEST = false;
    static constexpr bool LINEAR = false;
    __device__ __forceinline__ int row_of(int m) const { return m >= d.M ? -1 : m; }
    __device__ __forceinline__ auto raw_at(int r, int k0) const { return raw(r < 0 ? d.M : r, k0); }
    typedef uint4 Raw;
    LoadDesc d;
    __device__ __forceinline__ uint4 cvt(const Raw& r) const { return r; }
    __device__ __forceinline__ Raw raw(int m, int k0) const { return chunk(m, k0); }
    __device__ __forceinline__ uint4 chunk(int m, int k0) const {
        if (m >= d.M || k0 >= d.K) return make_uint4(0, 0, 0, 0);
        return raw_unc(m, k0);
    }
    __device__ __forceinline__ uint4 raw_unc(int m, int k0) const {
        const int H = d.p0, W = d.p1, C = d.p2, h2 = H >> 1, w2 = W >> 1;
        const int b = m / (h2 * w2), ij = m - b * h2 * w2, i = ij / w2, j = ij - i * w2;
        const float mu = d.aux0[m], rs = d.aux1[m];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e, blk = k / C, c = k - blk * C, wp = blk >> 1, hp = blk & 1;
            const float x = ((const float*)d.ptr)[(((long)b * H + 2 * i + hp) * W + 2 * j + wp) * C + c];
            v[e] = (x - mu) * rs * d.aux2[k] + d.aux3[k];
        }
        return pack8(v);
    }
};

template <int AK>
ALoad<AK> make_loader(const swv2_operand* o) {
    ALoad<AK> l;
    l.d.ptr = o->ptr; l.d.rowidx = o->rowidx; l.d.aux0 = o->aux0; l.d.aux1 = o->aux1; l.d.aux2 = o->aux2;
    l.d.aux3 = o->aux3; l.d.ld = o->ld; l.d.M = o->rows; l.d.K = o->cols;
    l.d.p0 = o->p[0]; l.d.p1 = o->p[1]; l.d.p2 = o->p[2]; l.d.p3 = o->p[3];
    l.d.mg0 = l.d.mg1 = l.d.mg2 = 0;
    if (AK == A_HEADS) {
        l.d.mg0 = fdiv_magic(o->p[2]); l.d.mg1 = fdiv_magic(o->p[0]);
        l.d.p1 = o->p[3] == 16 ? 4 : o->p[3] == 32 ? 5 : o->p[3] == 64 ? 6 : o->p[3] == 128 ? 7 : -1;
        l.d.mg2 = fdiv_magic(o->p[3]);
    }
    if (AK == A_PATCH) { l.d.mg0 = fdiv_magic((o->p[1] / 4) * (o->p[2] / 4)); l.d.mg1 = fdiv_magic(o->p[2] / 4); }
    if (AK == A_BF16_CS) l.d.mg0 = fdiv_magic(o->p[0]);
    return l;
}

int check_operand(const swv2_operand* o, const char* who) {
    SWV2_CHECK_ARG(o && o->ptr, "%s: null operand", who);
    SWV2_CHECK_ARG(o->rows > 0 && o->cols > 0, "%s: empty operand", who);
    SWV2_CHECK_ARG(o->cols % 8 == 0, "%s: operand width %d must be a multiple of 8", who, o->cols);
    SWV2_CHECK_ARG(((uintptr_t)o->ptr & 15) == 0, "%s: operand pointer must be 16-byte aligned", who);
    if (o->kind == SWV2_OP_F32 || o->kind == SWV2_OP_BF16 || o->kind == SWV2_OP_BF16_GELU || o->kind == SWV2_OP_BF16_CSCALE)
        SWV2_CHECK_ARG(o->ld % 8 == 0 && o->ld >= o->cols, "%s: row pitch %ld must be a multiple of 8 and >= cols", who, o->ld);
    if (o->kind == SWV2_OP_F32 || o->kind == SWV2_OP_BF16 || o->kind == SWV2_OP_BF16_GELU)
        SWV2_CHECK_ARG((double)o->rows * (double)o->ld < 4.29e9, "%s: row-major operands are indexed with 32-bit element offsets (%d x %ld too large)", who, o->rows, o->ld);
    if (o->kind == SWV2_OP_BF16_CSCALE)
        SWV2_CHECK_ARG(o->aux0 && o->p[0] > 0 && o->p[2] * 16 >= o->cols && o->cols % 16 == 0 && !o->rowidx,
                       "%s: scaled operand needs aux0, rows per sample p[0] > 0, p[2] >= cols / 16 groups, no gather", who);
    if (o->kind == SWV2_OP_HEADS)
        SWV2_CHECK_ARG(o->p[3] == 16 || o->p[3] == 32 || o->p[3] == 64 || o->p[3] == 96 || o->p[3] == 128, "%s: head pad %d not in {16,32,64,96,128}", who, o->p[3]);
    if (o->kind == SWV2_OP_PATCH)
        SWV2_CHECK_ARG(o->p[1] % 4 == 0 && o->p[2] % 4 == 0 && o->cols == o->p[0] * 16, "%s: bad patch geometry", who);
    if (o->kind == SWV2_OP_MERGE_LN)
        SWV2_CHECK_ARG(o->aux0 && o->aux1 && o->aux2 && o->aux3 && o->cols == 4 * o->p[2], "%s: bad merge operand", who);
    return SWV2_OK;
}


}  // namespace
