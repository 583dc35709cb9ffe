// Shared device/host helpers for the swv2 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "../../include/swv2.h"

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SWV2_LOG2E 1.4426950408889634f
#define SWV2_LN100 4.605170185988092f
#define SWV2_NEG_BIG (-1.0e30f)

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf2f(short v) { return __uint_as_float(((uint32_t)(uint16_t)v) << 16); }
// plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}
// two explicit pairs: a 4-wide convertvector of values produced one by one (v_exp results) compiled to four single
// v_cvt_pk + two v_perm instead of two v_cvt_pk
__device__ __forceinline__ bf16x4 f2bf4(f32x4 v) {
    const uint32_t p[2] = {f2bf2(v[0], v[1]), f2bf2(v[2], v[3])};
    return __builtin_bit_cast(bf16x4, p);
}

// ---- MFMA wrappers (wave64) ---------------------------------------------------------------
// v_mfma_f32_16x16x16_bf16: A[i=l&15][k=4*(l>>4)+j], B[k=4*(l>>4)+j][n=l&15], C[row=4*(l>>4)+r][col=l&15]
__device__ __forceinline__ f32x4 mfma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15], same C map
__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q / cols 4p..4p+3 of a 4x16 block of
// 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 lds_tr_read(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)p);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// wave64 sum on the vector ALU (DPP), total valid in lane 63 only: quad swaps, row mirrors, then the row broadcasts.  The
// __shfl_xor form above goes through the LDS crossbar (ds_bpermute + lgkmcnt wait per step: ~1.2 us for 8 values measured);
// use this one where only one lane consumes the sum.
__device__ __forceinline__ float wave_sum_dpp63(float v) {
    auto dpp = [](float x, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});    // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});    // row_mirror: every lane = its row's sum
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});    // row_bcast:15 into rows 1, 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});    // row_bcast:31 into rows 2, 3
    return v;
}

// All-reduce sums on the vector ALU only (no LDS crossbar): every lane of a group of G consecutive lanes gets the group's sum.
//   G <= 16: DPP (quad swaps, row_half_mirror, row_mirror);  G = 32: + v_permlane16_swap of the value with a copy of itself (rows
//   (x0, x0, x2, x2) + (x1, x1, x3, x3));  G = 64: + v_permlane32_swap likewise.  A __shfl_xor step is a ds_bpermute (an LDS-pipe
//   instruction + an lgkmcnt wait per step, ~100 cycles of latency each); the LayerNorm backward of a block did 160 of them per thread.
// Inline assembly for the swaps: this toolchain compiles element 1 of __builtin_amdgcn_permlane{16,32}_swap's result as element 0
// (see Epi<E_UNPATCH_LOSS>::wave_sum8_rows in gemm.hip); s_nop 1 = the wait the compiler itself puts between a vector write and a swap.
#ifdef SWV2_ALLSUM_SHFL        // (A/B and bisection: the ds_bpermute forms)
__device__ __forceinline__ float xor16_allsum(float x) { return x + __shfl_xor(x, 16); }
__device__ __forceinline__ float xor32_allsum(float x) { return x + __shfl_xor(x, 32); }
template <int G>
__device__ __forceinline__ float group_allsum(float v) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
#else
__device__ __forceinline__ float xor16_allsum(float x) {
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float xor32_allsum(float x) {
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
template <int G>
__device__ __forceinline__ float group_allsum(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "power-of-two lane groups");
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    if constexpr (G >= 2) v += dpp(v, std::integral_constant<int, 0xb1>{});      // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp(v, std::integral_constant<int, 0x4e>{});      // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    if constexpr (G >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});    // row_mirror
    if constexpr (G >= 32) v = xor16_allsum(v);
    if constexpr (G >= 64) v = xor32_allsum(v);
    return v;
}
#endif

// ---- host side ----------------------------------------------------------------------------
void swv2_set_error(const char* fmt, ...);
#define SWV2_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            swv2_set_error(__VA_ARGS__);          \
            return SWV2_ERR_INVALID;              \
        }                                         \
    } while (0)
#define SWV2_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            swv2_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return SWV2_ERR_LAUNCH;                                                  \
        }                                                                            \
    } while (0)

void swv2_launch_ln_partials_reduce(const float* ws, float* dgamma, float* dbeta, int nblocks, int C, hipStream_t st);
// two partial-sum sets (LayerNorm2 of the MLP branch, LayerNorm1 of the attention branch) folded by ONE launch
void swv2_launch_ln_partials_reduce2(const float* ws1, float* dg1, float* db1, int n1, const float* ws2, float* dg2, float* db2,
                                     int n2, int C, hipStream_t st);
// swv2_mlp_bwd / swv2_proj_ln_bwd; deferred != nullptr: the d gamma / d beta partial rows stay in a->ws, their number is
// reported there and the caller folds them later (swv2_block_bwd: one reduction launch per block instead of two)
int swv2_mlp_bwd_impl(const swv2_mlp_bwd_args* a, void* stream, int* deferred, float* zero, long zero_n);
int swv2_proj_ln_bwd_impl(const swv2_proj_ln_bwd_args* a, void* stream, int* deferred);

// swv2_block_wgrad with a rider (block.hip -> gemm_tn.hip): the partial rows of the block's two LayerNorm backward kernels are folded
// into d gamma / d beta by the weight-gradient reduction launch instead of a launch of their own (the 128 x 128 tile path launches
// swv2_launch_ln_partials_reduce2 itself).  ln == nullptr: plain swv2_block_wgrad.
struct swv2_ln_partials { const float* ws[2]; float* dgamma[2]; float* dbeta[2]; int n[2]; int C; };
int swv2_block_wgrad_ln(const swv2_wgrad_item* items4, int slices, void* ws, size_t ws_bytes, const swv2_ln_partials* ln, void* stream);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
