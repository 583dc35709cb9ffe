// bf16 MFMA GEMM engine for the tall-skinny products of the SwinV2 hot path (gfx950 / CDNA4).
//
//   NT:  C[M][N]  = A[M][K] . W[N][K]^T          forward linears, dX = dY . W  (W pre-transposed by swv2_cast_weights)
//   TN:  dW[N][K] = sum_m dY[m][N]^T . X[m][K]    weight gradients (+ bias gradient = column sums of dY)
//
// M is the token count (10^5 .. 10^6), N and K are channel counts (96 .. 1232), so every product is
// HBM-bound on the A / C streams.  The engine therefore spends its structure on the memory side:
//   * A "loader" functor maps (row, 8-element k-chunk) -> 16 bytes of bf16; it folds into the load what PyTorch does
//     as separate passes in the reference: fp32->bf16 conversion, the cyclic roll + window partition gather
//     (swinv2_global.py:457,89-101), the 4x4 patch im2col of PatchEmbed (:537), the 2x2 PatchMerging gather (:519),
//     un-merging of attention heads (:318), GELU (timm Mlp) and LayerNorm-on-load.
//   * An "epilogue" functor receives 16x64 fp32 sub-tiles staged in wave-private LDS and writes them out in
//     whatever layout the consumer wants, always as full 16-byte, row-contiguous stores: bias, the head-major
//     window-ordered q/k/v layout with the L2 normalisation of q and k (:304), GELU', un-patchify + skip (:784-802).
//   * One workgroup owns a 128-row panel of A and walks all N tiles itself, so the panel's re-reads hit its own
//     XCD's L2 instead of crossing XCDs.
// Tile: 128x128x64, 256 threads = 4 waves in a 2x2 grid of 64x64 wave tiles (16 accumulators of 16x16),
// v_mfma_f32_16x16x32_bf16, one LDS tile buffer with register-staged prefetch (issue loads for step s+1, compute
// step s, write LDS after) and 3 workgroups per CU, XOR-swizzled 128-byte LDS rows for conflict-free ds_read_b128.
#include "common.h"

#include <cmath>
#include <cstdlib>
#include "gemm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Epilogues: tile(stage, m0, n0, lane) consumes a 16-row x 64-col fp32 sub-tile held in wave-private LDS
// (pitch EP floats): rows m0..m0+15, cols n0..n0+63 of C.
// ------------------------------------------------------------------------------------------------
constexpr int EP = 68;     // staging pitch in floats

struct EpiDesc {
    void* out;              // primary output
    const float* bias;      // [N] or null
    const void* aux;        // kind specific (pre-activation for GELU', skip input, ...)
    float* aux_out;         // kind specific second output (rnorm)
    const int32_t* rowidx;  // optional scatter table: logical row -> destination row, <0 = skip
    long ld;                // output row pitch (elements)
    int M, N;
    int p0, p1, p2, p3, p4;
    uint32_t mg0, mg1, mg2; // fdiv magics (launch_nt2)
    // E_UNPATCH_LOSS
    const float* loss_tar; const float* loss_qw; float* loss_part; uint16_t* loss_resid; int q0, q1, q2, q3;
};

enum { E_BF16 = 0, E_F32 = 1, E_QKV_HEADS = 2, E_GELU_GRAD = 3, E_UNPATCH = 4, E_HEADS = 5, E_F32_ACC = 6, E_BF16_GELU = 7,
       E_UNPATCH_LOSS = 8, E_UNPATCH_LOSS_SKIP = 9 };      // (9: the same epilogue with a skip tensor; compile-time, see the kernel)

template <int KIND> struct Epi;

// row-major bf16 store (+bias): lane -> (row = lane/4, 16 columns)
template <> struct Epi<E_BF16> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int r = lane >> 2, c0 = (lane & 3) * 16, m = m0 + r, n = n0 + c0;
        if (m >= d.M || n >= d.N) return;
        long dst = m;
        if (d.rowidx) { dst = d.rowidx[m]; if (dst < 0) return; }
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(v + 4 * i) = *(const f32x4*)(st + r * EP + c0 + 4 * i);
        if (d.bias) {
            if (n + 16 <= d.N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b4 = *(const f32x4*)(d.bias + n + 4 * i);
                    v[4 * i] += b4[0]; v[4 * i + 1] += b4[1]; v[4 * i + 2] += b4[2]; v[4 * i + 3] += b4[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += (n + i < d.N) ? d.bias[n + i] : 0.f;
            }
        }
        uint16_t* o = (uint16_t*)d.out + dst * d.ld + n;
        if (n + 16 <= d.N) {
            *(uint4*)o = pack8(v);
            *(uint4*)(o + 8) = pack8(v + 8);
        } else {
            for (int i = 0; i < 16 && n + i < d.N; ++i) o[i] = f2bf(v[i]);
        }
    }
};
// fc1 of the Mlp: out = acc + bias (bf16 pre-activation), aux_out = GELU(out) (bf16), same row-major layout
template <> struct Epi<E_BF16_GELU> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int r = lane >> 2, c0 = (lane & 3) * 16, m = m0 + r, n = n0 + c0;
        if (m >= d.M || n >= d.N) return;
        float v[16], gl[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(v + 4 * i) = *(const f32x4*)(st + r * EP + c0 + 4 * i);
        if (d.bias) {
            if (n + 16 <= d.N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b4 = *(const f32x4*)(d.bias + n + 4 * i);
                    v[4 * i] += b4[0]; v[4 * i + 1] += b4[1]; v[4 * i + 2] += b4[2]; v[4 * i + 3] += b4[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += (n + i < d.N) ? d.bias[n + i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) gl[i] = gelu_f(bf2f(f2bf(v[i])));   // GELU of the stored (bf16) pre-activation
        uint16_t* o = (uint16_t*)d.out + (long)m * d.ld + n;
        uint16_t* o2 = (uint16_t*)d.aux_out + (long)m * d.ld + n;
        if (n + 16 <= d.N) {
            *(uint4*)o = pack8(v);
            *(uint4*)(o + 8) = pack8(v + 8);
            *(uint4*)o2 = pack8(gl);
            *(uint4*)(o2 + 8) = pack8(gl + 8);
        } else {
            for (int i = 0; i < 16 && n + i < d.N; ++i) { o[i] = f2bf(v[i]); o2[i] = f2bf(gl[i]); }
        }
    }
};
// row-major fp32 store (+bias) (+aux: an fp32 tensor of the output's shape added at the destination row, e.g. the
// gradient that arrives over the residual connection)
template <> struct Epi<E_F32> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int r = lane >> 2, c0 = (lane & 3) * 16, m = m0 + r, n = n0 + c0;
        if (m >= d.M || n >= d.N) return;
        long dst = m;
        if (d.rowidx) { dst = d.rowidx[m]; if (dst < 0) return; }
        float* o = (float*)d.out + dst * d.ld + n;
        const float* ax = d.aux ? (const float*)d.aux + dst * d.ld + n : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = *(const f32x4*)(st + r * EP + c0 + 4 * i);
            if (n + 4 * i + 4 <= d.N) {
                if (d.bias) v += *(const f32x4*)(d.bias + n + 4 * i);
                if (ax) v += *(const f32x4*)(ax + 4 * i);
                *(f32x4*)(o + 4 * i) = v;
            } else {
                for (int j = 0; j < 4 && n + 4 * i + j < d.N; ++j)
                    o[4 * i + j] = v[j] + (d.bias ? d.bias[n + 4 * i + j] : 0.f) + (ax ? ax[4 * i + j] : 0.f);
            }
        }
    }
};
// fp32 accumulate into the destination (dx of the second branch summed into the first), no bias
template <> struct Epi<E_F32_ACC> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int r = lane >> 2, c0 = (lane & 3) * 16, m = m0 + r, n = n0 + c0;
        if (m >= d.M || n >= d.N) return;
        long dst = m;
        if (d.rowidx) { dst = d.rowidx[m]; if (dst < 0) return; }
        float* o = (float*)d.out + dst * d.ld + n;
        for (int i = 0; i < 16 && n + i < d.N; ++i) o[i] += st[r * EP + c0 + i];
    }
};
// q/k/v split into the head-major window layout with L2-normalised q and k (swinv2_global.py:300-304).
// logical col n = (part*h + head)*DP + j (head dim padded to DP by swv2_prep_weight, padded columns are exact zeros);
// logical row m = bw*Lp + t (rows t >= L are padding and are written as zeros).
// p0 = heads, p2 = Lp, p3 = DP, p4 = L ; out = qkvh ; aux_out = rnorm [Bw][h][2][Lp] ; bias is padded like n.
// PARTS = 3 with normalisation of parts 0 and 1 (E_QKV_HEADS) or 1 without (E_HEADS).
template <int DPc, bool NORM>
__device__ __forceinline__ void heads_item(const EpiDesc& d, const float* st, int m0, int n0, int lane) {
    const int h = d.p0, Lp = d.p2, L = d.p4, S = NORM ? 3 : 1;
    constexpr int SLOTS = 64 / DPc;                   // heads per 64-column sub-tile
    for (int it = lane; it < 16 * SLOTS; it += 64) {
        const int r = it & 15, slot = it >> 4;
        const int nb = n0 + slot * DPc, m = m0 + r;
        if (m >= d.M || nb >= d.N) continue;
        const int ph = nb / DPc, part = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - part * h;
        const int bw = fdiv(m, Lp, d.mg0), t = m - bw * Lp;
        const bool valid = t < L;
        float v[DPc];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < DPc; j += 4) {
            f32x4 x = *(const f32x4*)(st + r * EP + slot * DPc + j);
            if (d.bias) x += *(const f32x4*)(d.bias + nb + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j + e] = valid ? x[e] : 0.f;
                ss = fmaf(v[j + e], v[j + e], ss);
            }
        }
        float rn = 1.f;
        if (NORM && part < 2) {
            rn = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            d.aux_out[(((long)bw * h + hd) * 2 + part) * Lp + t] = valid ? rn : 0.f;
        }
        uint16_t* o = (uint16_t*)d.out + ((((long)bw * h + hd) * S + part) * Lp + t) * DPc;
#pragma unroll
        for (int j = 0; j < DPc; j += 8) {
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = v[j + e] * rn;
            *(uint4*)(o + j) = pack8(w);
        }
    }
}
// DP = 96 / 128 (head dims 65 .. 128): a wave's 64-column sub-tile is PART of a head, so the squared norm cannot be finished
// here.  Each half adds its partial sum of squares into rnorm (zeroed by the caller; two addends -> order-independent)
// and writes the UN-normalised bf16 values; swv2_qk_normalize then turns the sums into 1 / |.| and rescales q, k in
// place (one extra bf16 rounding of q^, k^ compared with the narrow-head epilogue).  lane = (row, 16-column quarter).
// PB: the lane's 16 bias values were loaded by the caller (once per tile: the wide kernels; a load inside this function sits
// under a condition and makes the compiler wait for every earlier store with s_waitcnt vmcnt(0)); zeros without a bias
template <bool NORM, bool PB = false>
__device__ __forceinline__ void heads_item_wide(const EpiDesc& d, const float* st, int m0, int n0, int lane, const f32x4* pb = nullptr) {
    const int h = d.p0, Lp = d.p2, L = d.p4, S = NORM ? 3 : 1;
    const int r = lane & 15, qd = lane >> 4, nb = n0 + 16 * qd, m = m0 + r;
    const bool in = (m < d.M) && (nb < d.N);
    const int DPv = d.p3, ph = DPv == 128 ? (nb >> 7) : fdiv(nb, DPv, d.mg2), part = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - part * h;
    const int bw = fdiv(min(m, d.M - 1), Lp, d.mg0), t = min(m, d.M - 1) - bw * Lp;
    const bool valid = in && t < L;
    float v[16];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        f32x4 x = *(const f32x4*)(st + r * EP + 16 * qd + j);
        if constexpr (PB) x += pb[j >> 2];
        else if (d.bias && in) x += *(const f32x4*)(d.bias + nb + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j + e] = valid ? x[e] : 0.f;
            ss = fmaf(v[j + e], v[j + e], ss);
        }
    }
    if (NORM) {
        if (DPv == 128) {                 // the wave's 64 columns lie in ONE head: one addend per row
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            if (in && qd == 0 && part < 2 && valid) atomicAdd(d.aux_out + (((long)bw * h + hd) * 2 + part) * Lp + t, ss);
        } else {                          // 96-wide heads: a 64-column sub-tile may straddle two heads -- one addend per 16-column group
            if (in && part < 2 && valid) atomicAdd(d.aux_out + (((long)bw * h + hd) * 2 + part) * Lp + t, ss);
        }
    }
    if (in) {
        uint16_t* o = (uint16_t*)d.out + ((((long)bw * h + hd) * S + part) * Lp + t) * DPv + (nb - ph * DPv);
        *(uint4*)o = pack8(v);
        *(uint4*)(o + 8) = pack8(v + 8);
    }
}
template <> struct Epi<E_QKV_HEADS> {
    EpiDesc d;
    // compile-time head pad (callers that checked p3 on the host): only that variant is inlined
    template <int DPC> __device__ __forceinline__ void tile_fixed(const float* st, int m0, int n0, int lane) const {
        heads_item<DPC, true>(d, st, m0, n0, lane);
    }
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        if (d.p3 == 16) heads_item<16, true>(d, st, m0, n0, lane);
        else if (d.p3 == 32) heads_item<32, true>(d, st, m0, n0, lane);
        else if (d.p3 == 64) heads_item<64, true>(d, st, m0, n0, lane);
        else heads_item_wide<true>(d, st, m0, n0, lane);
    }
};
// heads split without normalisation: [Bw][h][1][Lp][DP] (gradient of the merged attention output)
template <> struct Epi<E_HEADS> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        if (d.p3 == 16) heads_item<16, false>(d, st, m0, n0, lane);
        else if (d.p3 == 32) heads_item<32, false>(d, st, m0, n0, lane);
        else if (d.p3 == 64) heads_item<64, false>(d, st, m0, n0, lane);
        else heads_item_wide<false>(d, st, m0, n0, lane);
    }
};
// dh = (acc) * GELU'(pre-activation) ; aux = pre-activation bf16 [M][N] row-major, same pitch as out
template <> struct Epi<E_GELU_GRAD> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int r = lane >> 2, c0 = (lane & 3) * 16, m = m0 + r, n = n0 + c0;
        if (m >= d.M || n >= d.N) return;
        const uint16_t* pre = (const uint16_t*)d.aux + (long)m * d.ld + n;
        uint16_t* o = (uint16_t*)d.out + (long)m * d.ld + n;
        float v[16], hv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(v + 4 * i) = *(const f32x4*)(st + r * EP + c0 + 4 * i);
        if (n + 16 <= d.N) {
            unpack8(*(const uint4*)pre, hv);
            unpack8(*(const uint4*)(pre + 8), hv + 8);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] *= gelu_grad_f(hv[i]);
            *(uint4*)o = pack8(v);
            *(uint4*)(o + 8) = pack8(v + 8);
        } else {
            for (int i = 0; i < 16 && n + i < d.N; ++i) o[i] = f2bf(v[i] * gelu_grad_f(bf2f(pre[i])));
        }
    }
};
// head Linear + un-patchify (+ residual skip), swinv2_global.py:784-802.  The bf16 head weight is stored with its
// rows permuted to n' = c*16 + p*4 + q (swv2_cast_weights), so 16 consecutive columns are one channel's 4x4 patch:
//   y[b][c][4i+p][4j+q] = acc[m=(b,i,j)][c*16 + p*4 + q] (+ skip[b][c][4i+p][4j+q])
// p0 = Cout, p1 = H, p2 = W, p3 = Cskip (channels of the skip tensor, 0 = no skip) ; aux = skip ; out = y
// p4 = channels per sample of the tensor `out` points into (0 = Cout); aux_out = optional SECOND destination with ld
// channels per sample -- the autoregressive rollout writes a step's prediction straight into the concatenated result and
// into the next step's input buffer (helpers.py:26-41: two torch.cat copies of ~300 MB per sample and step otherwise)
template <> struct Epi<E_UNPATCH> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float* st, int m0, int n0, int lane) const {
        const int Cout = d.p0, H = d.p1, W = d.p2, Cs = d.p3, gw = W >> 2, gh = H >> 2;
        const int Ct = d.p4 ? d.p4 : Cout;
        // 16 rows x 4 channels x 4 p = 256 float4 items; lane -> row fastest so stores run along W
        for (int it = lane; it < 256; it += 64) {
            const int r = it & 15, cp = it >> 4, cl = cp >> 2, p = cp & 3;
            const int m = m0 + r, c = (n0 >> 4) + cl;
            if (m >= d.M || c >= Cout) continue;
            const int b = fdiv(m, gh * gw, d.mg0), ij = m - b * gh * gw, i = fdiv(ij, gw, d.mg1), j = ij - i * gw;
            f32x4 v = *(const f32x4*)(st + r * EP + cl * 16 + p * 4);
            if (Cs) v += *(const f32x4*)((const float*)d.aux + (((long)b * Cs + c) * H + 4 * i + p) * W + 4 * j);
            *(f32x4*)((float*)d.out + (((long)b * Ct + c) * H + 4 * i + p) * W + 4 * j) = v;
            if (d.aux_out) *(f32x4*)((float*)d.aux_out + (((long)b * d.ld + c) * H + 4 * i + p) * W + 4 * j) = v;
        }
    }
};

// un-patchify (+ skip) that also evaluates the geometric l2 loss of the prediction while it is in registers
// (losses.py:188-206: sum_hw q[h] (prd - tar)^2 and sum_hw q[h] tar^2 per (sample, channel); grids.py:115-117) and leaves the
// quadrature-weighted residual q[h] (prd - tar) as a bf16 [M][SWV2_LOSS_RESID_PITCH(N)] matrix in the GEMM's own layout for the head's backward.
// The reference (and rounds 1 - 2 here) read the 303 MB / sample prediction back twice (loss, loss gradient) and wrote a
// gradient of the same size that both backward GEMMs of the head re-read through the 4 x 4 patch gather.
//   row_of      : what a lane has to know about its row of a 16-row tile (sample, image position, quadrature weight), once per kernel
//   load_tar    : the lane's four target float4 of one 16 x 64 tile (four channels), requested a whole N tile ahead
//   tile_loss_t : one 16 x 64 tile on the accumulators of the TRANSPOSED product: lane = (token fr, image row p = g), register r = q, one
//                 channel per 16 x 16 tile; ls[2k], ls[2k + 1] = channel k's partial sums; only the bf16 residual is staged through LDS
//   flush       : per (N tile, 32-row group) wave-reduce the sums (lane swaps + DPP) and STORE them as the group's partial sums
//                 (swv2_loss_part_reduce adds the groups of a sample in a fixed order: the loss value is bit-reproducible).  The
//                 first version added them to the (sample, channel) sums with atomics: 146 addresses hit by 80 K wave-level
//                 atomics made the kernel 1071 us instead of 138 us.
// Rollouts (round 5): `out` may be a channel block of a larger tensor (p4 channels per sample, dump offset q2) and aux_out a second
// destination (the next step's input) written from the same registers.
template <> struct Epi<E_UNPATCH_LOSS> {
    EpiDesc d;
    __device__ __forceinline__ void tile(const float*, int, int, int) const {}
    // What one 16 x 64 tile needs from global memory: this lane's four target float4 (its row / image row, the four channels
    // of the wave's column block) and its quadrature weight.  Requested ONE TILE AHEAD of its use -- before the previous tile's
    // stores are issued, across the MFMA phase between two N tiles -- because vmcnt retires in order and counts stores: waiting
    // for a load that was issued behind a store is waiting for that store to complete (SQ_WAIT_ANY was 79 % of the kernel's
    // wave-cycles with the loads at the top of each tile, profiles/r03_pmc_wait.json).  Everything in load_tar / tile_loss is
    // straight-line code, so the waits are counted vmcnt(N): masked rows / channels load from clamped addresses and store into
    // small dump areas BEHIND the prediction and the residual (same scalar base + 32-bit offset; a select between two base
    // pointers needs 64-bit per-lane addresses, and the kernel spills).
    struct Tar { f32x4 t[4]; float q; };
    // what a lane needs to know about its row of a 16-row tile (token -> sample, image position): the same for every N tile, so the
    // kernel resolves it ONCE per row tile in front of its main loop (inside the epilogue it was ~50 integer instructions per tile)
    struct Row { uint32_t pix, tbase, ybase, nbase, sbase; float q; int b; bool ok; };
    __device__ __forceinline__ Row row_of(int m0, int lane) const {
        const int H = d.p1, W = d.p2, gw = W >> 2, gh = H >> 2;
        const int r = lane & 15, p = lane >> 4, m = m0 + r, mc = min(m, d.M - 1);
        const int b = fdiv(mc, gh * gw, d.mg0), ij = mc - b * gh * gw, i = fdiv(ij, gw, d.mg1), j = ij - i * gw;
        Row o;
        o.pix = (uint32_t)((4 * i + p) * W + 4 * j);
        o.tbase = (uint32_t)b * d.q0 + d.q1;                    // first target channel of the sample
        o.ybase = (uint32_t)b * (d.p4 ? d.p4 : d.p0);           // p4: channels per sample of the tensor `out` points into (rollouts)
        o.nbase = (uint32_t)b * (uint32_t)d.ld;                 // second destination (aux_out), ld channels per sample
        o.sbase = (uint32_t)b * d.p3;
        o.q = d.loss_qw[4 * i + p];
        o.b = b;
        o.ok = m < d.M;
        return o;
    }
    __device__ __forceinline__ void load_tar(const Row& rw, int n0, Tar& o) const {
        const int Cout = d.p0;
        const uint32_t plane = (uint32_t)(d.p1 * d.p2);
        o.q = rw.q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = (uint32_t)min((n0 >> 4) + k, Cout - 1);
#ifdef SWV2_HEAD_ABL_NO_TAR
            o.t[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
            o.t[k] = *(const f32x4*)(d.loss_tar + ((rw.tbase + c) * plane + rw.pix));
#endif
        }
    }
    // The same step on the accumulators of the TRANSPOSED product (D^T = W X^T: lane (token fr, image row p = g) holds the four
    // values q = 0..3 of one channel per 16 x 16 tile -- exactly the float4 of y / tar / skip this lane touches), so the prediction
    // never passes through the LDS: round 5 measured the staged form at 190 us of epilogue work with every memory stream removed
    // (16 + 12 LDS instructions and ~200 vector instructions per 16 x 64 tile; 422 us with the streams).  Only the bf16 residual is
    // staged (one 8-byte write per channel, two 16-byte reads) so that its rows leave as 128 contiguous bytes.
    template <bool HAS_SKIP>
    __device__ __forceinline__ void tile_loss_t(const f32x4 (&acc)[4], uint16_t* st16, const Row& rw, int m0, int n0, int lane, float (&ls)[8],
                                                float (&ls2)[8], int b0, const Tar& in) const {
        constexpr int RP = 72;                                   // bf16 pitch of the residual staging tile [16][64]
        const int Cout = d.p0;
        const int r = lane & 15, p = lane >> 4;
        const uint32_t plane = (uint32_t)(d.p1 * d.p2);
        const float q = in.q;
        float* outp = (float*)d.out;
        // residual rows have a pitch of whole 128-byte lines (SWV2_LOSS_RESID_PITCH): with N = 1168 elements per row every 128-byte
        // piece straddled two lines and the head's two backward products fetched 644 / 708 MB for 368 / 335 MB (profiles/r05_pmc_hbm.json)
        const uint32_t RPITCH = (uint32_t)SWV2_LOSS_RESID_PITCH(d.N);
        const uint32_t ydump = (uint32_t)d.q2 + lane * 4, ndump = (uint32_t)d.q3 + lane * 4, rdump = (uint32_t)d.M * RPITCH + lane * 16;
        const bool same = (rw.b == b0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (n0 >> 4) + k;
            const bool ok = rw.ok && c < Cout;
            f32x4 v = acc[k];
            if constexpr (HAS_SKIP) v += *(const f32x4*)((const float*)d.aux + ((rw.sbase + min(c, Cout - 1)) * plane + rw.pix));
#ifndef SWV2_HEAD_ABL_NO_Y       // (timing ablation: what materialising the prediction costs the loss epilogue)
            *(f32x4*)(outp + (ok ? (rw.ybase + c) * plane + rw.pix : ydump)) = v;
            // rollouts: the next step's input buffer gets the prediction from the same registers (wave-uniform branch around stores
            // only: the counted waits of the common path are unaffected)
            if (d.aux_out) *(f32x4*)(d.aux_out + (ok ? (rw.nbase + c) * plane + rw.pix : ndump)) = v;
#endif
            const f32x4 dd = v - in.t[k];
            const float e0 = q * (dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] + dd[3] * dd[3]);
            const float e1 = q * (in.t[k][0] * in.t[k][0] + in.t[k][1] * in.t[k][1] + in.t[k][2] * in.t[k][2] + in.t[k][3] * in.t[k][3]);
            // rows of the group's first sample -> ls, rows of the next sample (groups that straddle a sample boundary) -> ls2
            ls[2 * k] += (ok && same) ? e0 : 0.f;
            ls[2 * k + 1] += (ok && same) ? e1 : 0.f;
            ls2[2 * k] += (ok && !same) ? e0 : 0.f;
            ls2[2 * k + 1] += (ok && !same) ? e1 : 0.f;
            *(bf16x4*)(st16 + r * RP + 16 * k + 4 * p) = f2bf4(q * dd);
        }
        // residual rows out as bf16, row-major: lane -> (row lane / 4, 16 columns), 128 contiguous bytes per row
        const int r2 = lane >> 2, c0 = (lane & 3) * 16, m2 = m0 + r2, n = n0 + c0;
        const uint4 w0 = *(const uint4*)(st16 + r2 * RP + c0), w1 = *(const uint4*)(st16 + r2 * RP + c0 + 8);
        uint16_t* o = d.loss_resid + ((m2 < d.M && n < d.N) ? (uint32_t)m2 * RPITCH + n : rdump);
#ifndef SWV2_HEAD_ABL_NO_RESID
        *(uint4*)o = w0;
        *(uint4*)(o + 8) = w1;
#else
        if (w0.x == 0x12345678u) { *(uint4*)o = w0; *(uint4*)(o + 8) = w1; }
#endif
    }
    // loss_part[group][slot][Cout][2]: slot 0 = rows of the sample of the group's first row, slot 1 = rows of the following
    // sample (zero unless the group straddles a boundary).  Lane 63 holds the DPP sums and stores 8 floats per slot.
    // Sums of eight per-lane values over the wave with the gfx950 lane swaps: v_permlane32_swap folds the two halves of a value pair
    // into one register (lanes 0..31 = the first value, 32..63 = the second), v_permlane16_swap the 16-lane rows of two such registers,
    // four DPP steps finish the rows: 6 swaps + 6 + 8 adds for 8 values (six DPP steps per value were 96 moves + 96 adds per flush).
    // Row q = lane >> 4 of x0 then holds the sum of value (0, 2, 1, 3)[q], of x1 that of value 4 + (0, 2, 1, 3)[q], in every lane.
    static __device__ __forceinline__ void wave_sum8_rows(const float (&v)[8], float& x0, float& x1) {
        // inline assembly on purpose: with this toolchain (ROCm 7.2 clang) element 1 of __builtin_amdgcn_permlane{32,16}_swap's result
        // pair is compiled as element 0 again (`v_add_f32 v9, v10, v10` behind the swap -- found by the sums coming out 20 % low and
        // confirmed in a standalone kernel); the s_nop 1 in front is the wait the compiler itself puts between a vector write and the swap
        auto fold32 = [](float a, float b) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            return a + b;
        };
        auto fold16 = [](float a, float b) {
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
            return a + b;
        };
        auto rowsum = [](float x) {
            auto dpp = [](float y, auto ctrl) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), decltype(ctrl)::value, 0xf, 0xf, true));
            };
            x += dpp(x, std::integral_constant<int, 0xb1>{});      // quad_perm [1,0,3,2]
            x += dpp(x, std::integral_constant<int, 0x4e>{});      // quad_perm [2,3,0,1]
            x += dpp(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
            x += dpp(x, std::integral_constant<int, 0x140>{});     // row_mirror
            return x;
        };
        x0 = rowsum(fold16(fold32(v[0], v[1]), fold32(v[2], v[3])));
        x1 = rowsum(fold16(fold32(v[4], v[5]), fold32(v[6], v[7])));
    }
    // loss_part[group][slot][Cout][2]: slot 0 = rows of the sample of the group's first row, slot 1 = rows of the following
    // sample.  swv2_loss_part_reduce reads slot 1 only of the group in FRONT of a sample's first group, so slot 1 is written (zeros
    // unless the group straddles the boundary) only by groups whose successor row belongs to another sample (`edge`, wave-uniform: one
    // group in ~2 000).  The first lane of row q stores its two values; every other lane -- and channels past Cout / rows past M --
    // stores the same instruction into the dump area behind the residual, so the common path has no branch around memory operations.
    template <int GR> __device__ __forceinline__ void flush(float (&ls)[8], float (&ls2)[8], bool edge, int m_first, int n0, int lane) const {
        const int c0 = n0 >> 4, qr = lane >> 4, vi = ((qr & 1) << 1) | (qr >> 1);
        float* const dump = (float*)(d.loss_resid + (size_t)d.M * SWV2_LOSS_RESID_PITCH(d.N)) + 2 * lane;
        float* const sp = d.loss_part + ((long)(m_first / GR) * 2 * d.p0 + c0) * 2;
        const bool lead = (lane & 15) == 0 && m_first < d.M;
        const bool ok0 = lead && c0 + (vi >> 1) < d.p0, ok1 = lead && c0 + 2 + (vi >> 1) < d.p0;
        float x0, x1;
        wave_sum8_rows(ls, x0, x1);
        *(ok0 ? sp + vi : dump) = x0;
        *(ok1 ? sp + 4 + vi : dump + 1) = x1;
        if (edge) {
            wave_sum8_rows(ls2, x0, x1);
            *(ok0 ? sp + 2 * d.p0 + vi : dump) = x0;
            *(ok1 ? sp + 2 * d.p0 + 4 + vi : dump + 1) = x1;
        }
    }
};

template <> struct Epi<E_UNPATCH_LOSS_SKIP> : Epi<E_UNPATCH_LOSS> {};

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
// BMT = rows per workgroup: 128 (2 x 2 waves of 64 x 64) or 64 (2 x 2 waves of 32 x 64).  The 64-row variant exists for the
// grid quantisation: at 3 workgroups per CU the chip holds 768; 1013 row tiles of 128 (local batch 2) run as 1 full + 1
// third-full round (66 % of the slots busy on average, measured 12 - 16 us of batch-independent time per launch), 2026
// tiles of 64 as 2.64 of 3 rounds.
#ifndef SWV2_HEAD_OCC
#define SWV2_HEAD_OCC 2
#endif
template <int AK, int EK, int BMT = BM>
__global__ __launch_bounds__(NTHREADS, ((EK == E_UNPATCH_LOSS || EK == E_UNPATCH_LOSS_SKIP) && BMT == 64) ? SWV2_HEAD_OCC : 2) void gemm_nt_kernel(ALoad<AK> al, const uint16_t* __restrict__ Wb, Epi<EK> ep,
                                                              int M, int N, int K) {
    // one A|B tile buffer (32 KB) + wave-private epilogue staging (17 KB): 49 KB -> 3 workgroups (12 waves) per CU.
    // Latency hiding comes from the co-resident workgroups plus the register prefetch of the next step's tiles.
    // A panel: 2 k-steps resident (K <= 128: the panel is loaded and converted ONCE and reused by every N tile; PMC showed
    // the per-N-tile re-reads as real HBM traffic, 548 MB vs 330 MB algorithmic for fc1), B tile, epilogue staging.
    constexpr int ACH = BMT * KCH / NTHREADS;             // A chunks per thread per k-step (4 or 2)
    constexpr int RT = BMT / 32;                          // 16-row MFMA tiles per wave (4 or 2)
    constexpr int LOG_BMT = BMT == 128 ? 7 : 6;
    __shared__ __attribute__((aligned(16))) uint16_t smem[(2 * BMT + BN) * BK];
    __shared__ __attribute__((aligned(16))) float stage[4 * 16 * EP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;              // 2 x 2 waves, (16 RT) x 64 each
    const int m_base = blockIdx.x * BMT;
    const int ntiles = (N + BN - 1) / BN, ksteps = (K + BK - 1) / BK, steps = ntiles * ksteps;
    const bool a_res = (ksteps <= 2) && (ntiles > 1);        // A panel stays in LDS across the N tiles
    uint16_t* As0 = smem;
    uint16_t* Bs = smem + 2 * BMT * BK;

    typename ALoad<AK>::Raw ra[ACH];
    uint4 rb[4];
    // the panel rows this thread stages are the same for every (n-tile, k-step): resolve the gather ONCE
    int arow[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + i * NTHREADS;
        arow[i] = al.row_of(m_base + (ALoad<AK>::ROW_FASTEST ? (c & (BMT - 1)) : (c >> 3)));
    }
    auto issue = [&](int s) {
        const int nt = s / ksteps, ks = s - nt * ksteps;
        const bool need_a = !a_res || nt == 0;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * NTHREADS;
            const int kc = ALoad<AK>::ROW_FASTEST ? (c >> LOG_BMT) : (c & 7);
            if (need_a) ra[i] = al.raw_at(arow[i], ks * BK + kc * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * NTHREADS;
            const int rn = c >> 3, kcb = c & 7, n = nt * BN + rn, k0 = ks * BK + kcb * 8;
            rb[i] = (n < N && k0 < K) ? *(const uint4*)(Wb + (long)n * K + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&](int s) {
        const int nt = s / ksteps, ks = s - nt * ksteps;
        const bool need_a = !a_res || nt == 0;
        uint16_t* As = As0 + (a_res ? ks * BMT * BK : 0);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * NTHREADS;
            int r, kc;
            if (ALoad<AK>::ROW_FASTEST) { r = c & (BMT - 1); kc = c >> LOG_BMT; } else { r = c >> 3; kc = c & 7; }
            if (need_a) *(uint4*)(As + swz(r, kc)) = al.cvt(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * NTHREADS;
            *(uint4*)(Bs + swz(c >> 3, c & 7)) = rb[i];
        }
    };

    f32x4 acc[RT][4];
    constexpr bool LOSS = EK == E_UNPATCH_LOSS || EK == E_UNPATCH_LOSS_SKIP;
    // target sets: lt[i] = this N tile's row tile i, ltn[i] = the NEXT N tile's, requested a whole N tile (two row tiles + one MFMA
    // phase) ahead: one row tile ahead the wave had 4 KB of target loads in flight -- 32 KB per CU, ~16 GB/s per CU at 2 us
    typedef typename std::conditional<LOSS, typename Epi<E_UNPATCH_LOSS>::Tar, int>::type TarT;
    [[maybe_unused]] TarT lt[RT], ltn[RT];
    [[maybe_unused]] typename std::conditional<LOSS, typename Epi<E_UNPATCH_LOSS>::Row, int>::type lrow[RT];
    if constexpr (LOSS) {
#pragma unroll
        for (int i = 0; i < RT; ++i) lrow[i] = ep.row_of(m_base + wr * 16 * RT + 16 * i, lane);
#pragma unroll
        for (int i = 0; i < RT; ++i) ep.load_tar(lrow[i], wc * 64, lt[i]);
    }
    issue(0);
    if constexpr (LOSS) commit(0);
    for (int s = 0; s < steps; ++s) {
        const int nt = s / ksteps, ks = s - nt * ksteps;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // LOSS: the tile of step s was committed at the end of step s - 1 -- inside the branch that ran (or did not run) the
        // epilogue, so the wait for the prefetched weight tile is a counted vmcnt(N) behind the epilogue's straight-line
        // loads / stores instead of the vmcnt(0) a wait behind the merge of the two paths gets (which would wait for every store
        // of the epilogue to complete)
        if constexpr (!LOSS) commit(s);
        __syncthreads();
        if (s + 1 < steps) issue(s + 1);
        const uint16_t* As = As0 + (a_res ? ks * BMT * BK : 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[RT], bf[4];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = *(const bf16x8*)(As + swz(wr * 16 * RT + i * 16 + fr, kk * 4 + g));
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *(const bf16x8*)(Bs + swz(wc * 64 + i * 16 + fr, kk * 4 + g));
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // LOSS: the transposed product (rows = output columns): see Epi<E_UNPATCH_LOSS>::tile_loss_t
                    if constexpr (LOSS) acc[i][j] = mfma32(bf[j], af[i], acc[i][j]);
                    else acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
                }
        }
        __syncthreads();                                   // tile consumed: the next commit may overwrite it
        if (ks == ksteps - 1) {
            float* st = stage + wave * 16 * EP;            // wave-private: no further barrier needed
            const int m_w = m_base + wr * 16 * RT, n_w = nt * BN + wc * 64;
            if constexpr (LOSS) {
                static_assert(!LOSS || RT == 2 || RT == 4, "even number of row tiles: two named target sets");
                // partial sums per group of SWV2_LOSS_GROUP_ROWS = 32 rows: one group (64-row workgroups) or two (128) per wave
#pragma unroll
                for (int grp = 0; grp < RT / 2; ++grp) {
                    const int m_g = m_w + 32 * grp;
                    const int tps = (ep.d.p1 >> 2) * (ep.d.p2 >> 2);                                       // tokens per sample
                    const int b0 = fdiv(min(m_g, M - 1), tps, ep.d.mg0);                                   // sample of the group's first row
                    const bool edge = (m_g + 32 >= M) || fdiv(m_g + 32, tps, ep.d.mg0) != b0;              // the row behind the group: another sample
                    float ls[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ls2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * grp + ii;
                        // request this row tile's targets of the NEXT N tile before this tile's stores (channels clamped inside:
                        // unconditional also behind the last N tile)
                        ep.load_tar(lrow[i], n_w + BN, ltn[i]);
                        ep.template tile_loss_t<EK == E_UNPATCH_LOSS_SKIP>(acc[i], (uint16_t*)st, lrow[i], m_w + 16 * i, n_w, lane, ls, ls2, b0, lt[i]);
                    }
                    ep.template flush<32>(ls, ls2, edge, m_g, n_w, lane);
                }
#pragma unroll
                for (int i = 0; i < RT; ++i) lt[i] = ltn[i];
                if (s + 1 < steps) commit(s + 1);
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[(4 * g + r) * EP + 16 * j + fr] = acc[i][j][r];
                    ep.tile(st, m_w + i * 16, n_w, lane);
                }
            }
        } else if constexpr (LOSS) {
            if (s + 1 < steps) commit(s + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wide NT kernel: 256 x 256 x 64 tiles for the MFMA-bound products of wide models (the reference's own swin_73var is
// embed_dim 768: K and N of 768 .. 3072, where gemm_nt_kernel's 64 x 64 wave tiles and its single LDS buffer with two
// barriers per k-step reach 8 - 23 % of the matrix peak).  One 8-wave workgroup per CU, wave tile 128 x 64 (128 fp32
// accumulators; per K = 32: 12 ds_read_b128 for 32 MFMAs), two LDS tile buffers (2 x 64 KB) with the next k-step's global
// loads in flight in registers across the MFMAs and ONE barrier per k-step.  All staging loads are unconditional (rows
// past M: the operand's first row, zeroed at commit), so the compiler can count them in s_waitcnt.
// Grid: one workgroup per output tile.  Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each XCD
// with its own L2; the index is decoded so that the ~32 tiles an XCD runs concurrently are 8 row panels x 4 column tiles:
// every A line is then fetched from HBM / Infinity Cache once per 4 and every weight line once per 8 workgroups instead of
// once each (speed only: any mapping is correct).
// ------------------------------------------------------------------------------------------------
constexpr int WBM = 256, WBN = 256, WTH = 512;
// LDS-DMA load: 16 bytes per lane from (scalar base + 32-bit byte offset) to LDS at m0 + 16 * lane, no register.  Inline assembly
// on purpose: through the builtin the compiler orders every later LDS access behind the load with s_waitcnt vmcnt(0).  The
// compiler does not count these in its own vmcnt bookkeeping; VMEM operations return in order, so an uncounted operation can
// only make a compiler-placed wait longer, never too short; the waits for the DMA'd tiles are placed by hand.
__device__ __forceinline__ void dma_x4(const void* base, uint32_t byte_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base), "s"(lds_addr) : "memory");
}
// Epilogue of a wave's 128 x 64 accumulator tile in the wide kernels.  Epi<>::tile loads its bias / residual / pre-activation
// operands under conditions, which costs an s_waitcnt vmcnt(0) per 16-row sub-tile -- a wait for the acknowledgement of every
// store of the previous sub-tile (measured: 20 us per tile for 256 KB, half the kernel at K = 768).  Here everything a lane needs
// for all eight sub-tiles is loaded up front (bias: the lane's columns are the same in every sub-tile; scatter rows), the
// per-sub-tile operands are loaded unconditionally (clamped rows) one sub-tile ahead, and only stores sit under the row mask:
// the waits are counted and the stores of a tile leave back to back.  Same arithmetic, in the same order, as Epi<>::tile.
template <int EK>
__device__ __forceinline__ void wide_epilogue(const Epi<EK>& ep, const f32x4 (&acc)[8][4], float* st, int m_w, int n_w, int lane) {
    const EpiDesc& d = ep.d;
    const int fr = lane & 15, g = lane >> 4;
    auto stage = [&](int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(4 * g + r) * EP + 16 * j + fr] = acc[i][j][r];
    };
    if constexpr (EK == E_F32 || EK == E_BF16 || EK == E_BF16_GELU || EK == E_GELU_GRAD) {
        // Row-major outputs.  A store instruction writes WHOLE 128-byte lines: fp32: 4 rows x 256 bytes (lane -> row 4 q + (lane >> 4),
        // 4 columns), bf16: 8 rows x 128 bytes (lane -> row 8 q + (lane >> 3), 8 columns).  With Epi<>::tile's map (a lane owns 16
        // columns of one row: an instruction writes 16-byte pieces 64 bytes apart) a line is completed by four instructions, and once
        // the stores leave back to back lines are evicted half written: measured 67 us per tile of 256 KB instead of 20.
        // N is a multiple of 256 here: no column tail.
        constexpr bool F32O = EK == E_F32;
        constexpr int NQ = F32O ? 4 : 2, RQ = 16 / NQ, CW = F32O ? 4 : 8;      // instructions per sub-tile, rows per instruction, columns per lane
        const int rl = F32O ? (lane >> 4) : (lane >> 3), cl = F32O ? (lane & 15) * 4 : (lane & 7) * 8;
        const int n = n_w + cl;
        f32x4 b4[CW / 4];
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) b4[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (d.bias) {
#pragma unroll
            for (int q = 0; q < CW / 4; ++q) b4[q] = *(const f32x4*)(d.bias + n + 4 * q);
        }
        const bool scatter = (EK == E_F32 || EK == E_BF16) && d.rowidx != nullptr;
        // (compile-time variants for the two uniform run-time conditions, so that every load of a variant is unconditional)
        // Load order: the scatter rows of all eight sub-tiles first; the per-sub-tile operand (residual / pre-activation) of
        // sub-tile i + 1 BEFORE the stores of sub-tile i -- VMEM operations retire in order, so a load issued behind a store is
        // waited for together with that store's acknowledgement.
        auto body = [&](auto scat, auto has_aux) {
            constexpr bool OPL = decltype(has_aux)::value || EK == E_GELU_GRAD;       // a per-sub-tile operand is loaded
            constexpr bool SC = decltype(scat)::value;
            [[maybe_unused]] int dstv[3][NQ];                 // scatter rows of sub-tiles i, i + 1, i + 2 (ring; loaded two ahead)
            auto rows = [&](int i) {
                if constexpr (SC) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) dstv[i % 3][q] = d.rowidx[min(m_w + 16 * i + RQ * q + rl, d.M - 1)];
                }
            };
            rows(0); rows(1);
            auto offs = [&](int i, long (&off)[NQ], bool (&ok)[NQ]) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int m = m_w + 16 * i + RQ * q + rl;
                    int dst = min(m, d.M - 1);
                    if constexpr (SC) dst = dstv[i % 3][q];
                    ok[q] = m < d.M && dst >= 0;
                    off[q] = (long)max(dst, 0) * d.ld + n;
                }
            };
            [[maybe_unused]] f32x4 a[2][NQ];
            [[maybe_unused]] uint4 pv[2][NQ];
            auto opload = [&](int i) {
                if constexpr (OPL) {
                    long off[NQ];
                    bool ok[NQ];
                    offs(i, off, ok);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if constexpr (EK == E_F32) a[i & 1][q] = *(const f32x4*)((const float*)d.aux + off[q]);
                        else pv[i & 1][q] = *(const uint4*)((const uint16_t*)d.aux + off[q]);
                    }
                }
            };
            opload(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long off[NQ];
                bool ok[NQ];
                offs(i, off, ok);
                if (i + 2 < 8) rows(i + 2);
                if (i + 1 < 8) opload(i + 1);
                stage(i);
                if constexpr (EK == E_F32) {
                    float* out = (float*)d.out;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        f32x4 v = *(const f32x4*)(st + (RQ * q + rl) * EP + cl);
                        v += b4[0];
                        if constexpr (decltype(has_aux)::value) v += a[i & 1][q];
                        if (ok[q]) *(f32x4*)(out + off[q]) = v;
                    }
                } else {
                    uint16_t* out = (uint16_t*)d.out;
                    [[maybe_unused]] uint16_t* out2 = (uint16_t*)d.aux_out;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float v[8];
                        *(f32x4*)v = *(const f32x4*)(st + (RQ * q + rl) * EP + cl);
                        *(f32x4*)(v + 4) = *(const f32x4*)(st + (RQ * q + rl) * EP + cl + 4);
                        if constexpr (EK == E_GELU_GRAD) {
                            float hv[8];
                            unpack8(pv[i & 1][q], hv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(hv[e]);
                            if (ok[q]) *(uint4*)(out + off[q]) = pack8(v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += b4[e >> 2][e & 3];
                            if constexpr (EK == E_BF16_GELU) {
                                float gl[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) gl[e] = gelu_f(bf2f(f2bf(v[e])));   // GELU of the stored (bf16) pre-activation
                                if (ok[q]) { *(uint4*)(out + off[q]) = pack8(v); *(uint4*)(out2 + off[q]) = pack8(gl); }
                            } else {
                                if (ok[q]) *(uint4*)(out + off[q]) = pack8(v);
                            }
                        }
                    }
                }
            }
        };
        const bool aux = EK == E_F32 && d.aux != nullptr;
        if (scatter) { if (aux) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
        else { if (aux) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
    } else if constexpr (EK == E_QKV_HEADS || EK == E_HEADS) {
        if (d.p3 > 64) {                 // lane -> (row lane & 15, 16-column quarter lane >> 4)
            const int nb = n_w + 16 * g;
            f32x4 b4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (d.bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4*)(d.bias + nb + 4 * q);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                stage(i);
                heads_item_wide<EK == E_QKV_HEADS, true>(d, st, m_w + 16 * i, n_w, lane, b4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { stage(i); ep.tile(st, m_w + 16 * i, n_w, lane); }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { stage(i); ep.tile(st, m_w + 16 * i, n_w, lane); }
    }
}

__device__ __forceinline__ int swz32(int r, int c) { return r * 32 + ((c ^ ((0 - (r >> 2)) & 3)) << 3); }

// decode the v-th tile of XCD x (see the header comment): false = past the matrix
__device__ __forceinline__ bool wide_tile(int x, int v, int mtiles, int ntn, int& mt, int& nt) {
    const int per_group = 8 * ntn;
    const int gl = v / per_group, r = v - gl * per_group;
    const int nb = r >> 5, rr = r & 31;               // column block of 4 tiles; (panel, column) inside it: rr < 8 * min(4, ntn - 4 nb)
    mt = (gl * 8 + x) * 8 + (rr & 7);
    nt = 4 * nb + (rr >> 3);
    return mt < mtiles && nt < ntn;
}

// ---- raw bf16 operands (A_BF16 without a gather table, A_HEADS): both operands reach LDS by DMA, no staging registers ----
// Persistent workgroups (32 per XCD) walk their XCD's tile sequence; the DMA pipeline runs THROUGH the tile boundaries.
//   LDS: four stage slots of 32 k (A[256][32] | B[256][32] bf16 = 32 KB each; 64-byte rows, chunk c of row r at physical chunk
//   c ^ ((-(r >> 2)) & 3): the 16 lanes ds_read_b128 serves per cycle ({0-3, 12-15, 20-27}, ...) then fall on 16 different 16-byte
//   bank groups) + a 2 KB pad behind slot 3 (the epilogue's staging = slot 3 + pad).  Global stage q = (tile, 32-k step) lives in
//   slot q & 3; K is a multiple of 128, so a tile's last stage sits in slot 3 and the next tile's stages 0, 1, 2 -- issued during
//   the last three steps of this tile -- in slots 0, 1, 2.
// Two wave groups (waves 0-3: rows 0-127 of the tile, waves 4-7: rows 128-255; waves w and w + 4 share a SIMD) run the same loop
// one barrier apart: while one group issues its 32 MFMAs of a stage, the other issues DMA and reads its fragments from LDS, so
// each SIMD's matrix pipe always has one wave feeding it.  Per stage t and wave:
//     LOAD(t): 4 DMA instructions of stage t + 3 -> slot (t + 3) & 3 | 12 ds_read_b128 of stage t | vmcnt(8) | lgkmcnt(0)
//     barrier | MFMA(t): 32 MFMAs | barrier
// group 0 runs LOAD(t) in section 2 t and MFMA(t) in section 2 t + 1, group 1 one section later (sections = the intervals between
// workgroup barriers).  RAW: a wave retires its own DMA of stage t + 1 with the counted vmcnt(8) at the end of LOAD(t) (stages
// t + 2, t + 3 = 8 instructions stay in flight; VMEM operations retire in order, so older epilogue stores only make the wait
// longer), i.e. by the end of section 2 t + 1 at the latest; the first read of stage t + 1 is in section 2 t + 2, behind that
// section's opening barrier.  WAR: slot (t + 3) & 3 held stage t - 1, whose last reads (group 1, LOAD(t - 1), retired by its
// lgkmcnt(0)) are in section 2 t - 1; the DMA into it is issued in sections 2 t (group 0) and 2 t + 1 (group 1).  Every wave issues
// exactly 4 DMA instructions per stage, in stage order; behind the last tile the last stage is re-read into slots nobody reads.
// Tile end (last stage L in slot 3): group 0 waits one section, BOTH groups run the epilogue in the same section (staging in
// slot 3 + pad: the last reads of slot 3 are two sections back, the next DMA into it is issued in LOAD(0) of the next tile, one
// barrier later), then group 1 waits one section: the stagger is restored.  Two idle sections per tile buy the epilogues of all
// eight waves side by side and a prologue that is already in LDS.
constexpr int WSTG = (WBM + WBN) * 32;                 // elements per stage slot
constexpr int WSM_BYTES = 4 * WSTG * 2 + 2048;
template <int AK, int EK>
__global__ __launch_bounds__(WTH) void gemm_nt_wide_dma_kernel(ALoad<AK> al, const uint16_t* __restrict__ Wb, Epi<EK> ep,
                                                               int M, int N, int K, int mtiles, int ntn, int vmax) {
    static_assert(AK == A_BF16 || AK == A_HEADS, "DMA needs a raw bf16 operand");
    __shared__ __attribute__((aligned(1024))) unsigned char smem_b[WSM_BYTES];
    static_assert(3 * WSTG * 2 + 8 * 16 * EP * 4 <= WSM_BYTES, "epilogue staging = slot 3 + pad");
    uint16_t* const smem = (uint16_t*)smem_b;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;              // 2 x 4 waves, 128 x 64 each
    const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3, stages = K / 32;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // DMA geometry: one wave instruction fills 16 stage rows x 64 bytes (1 KB of LDS); instruction i (of 2 per operand) of this
    // wave covers rows 32 wave + 16 i .. + 15; lane -> (row + (lane >> 2), physical chunk lane & 3)
    const int drow = wave * 32 + (lane >> 2);
    int dkc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) dkc[i] = (lane & 3) ^ ((0 - ((drow + 16 * i) >> 2)) & 3);
    struct Tile { int m_base, n_base; uint32_t boff[2], aoff[2]; };
    auto make_tile = [&](int mt, int nt) {
        Tile t;
        t.m_base = mt * WBM; t.n_base = nt * WBN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = drow + 16 * i;
#ifdef SWV2_WIDE_SAME_TILE          // (timing ablation: every workgroup streams the first tile's operands -- all L2 hits; wrong results)
            t.boff[i] = 2u * (uint32_t)(row * K + dkc[i] * 8);
            t.aoff[i] = AK == A_BF16 ? 2u * (uint32_t)(row * (int)al.d.ld + dkc[i] * 8) : 0u;
#else
            t.boff[i] = 2u * (uint32_t)(min(t.n_base + row, N - 1) * K + dkc[i] * 8);      // (N = 192, one partial column tile: rows past N re-read the last one, their columns are never stored)
            t.aoff[i] = AK == A_BF16 ? 2u * (uint32_t)(min(t.m_base + row, M - 1) * (int)al.d.ld + dkc[i] * 8) : 0u;
#endif
        }
        return t;
    };
    auto issue = [&](const Tile& tl, int t, int slot) {
        const uint32_t la = lds0 + (uint32_t)(slot * WSTG * 2) + (uint32_t)(wave * 2048), lb = la + WBM * 32 * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_x4(Wb + t * 32, tl.boff[i], lb + 1024 * i);
        if constexpr (AK == A_BF16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma_x4((const uint16_t*)al.d.ptr + t * 32, tl.aoff[i], la + 1024 * i);
        } else {                             // head-major rows: the element offset of (row, k) is not affine in k
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma_x4(al.d.ptr, 2u * al.elem_off(min(tl.m_base + drow + 16 * i, M - 1), t * 32 + dkc[i] * 8), la + 1024 * i);
        }
    };
    // first tile of this workgroup
    int v = blockIdx.x >> 3, mt = 0, nt = 0;
    while (v < vmax && !wide_tile(xcd, v, mtiles, ntn, mt, nt)) v += nper;
    if (v >= vmax) return;
    Tile cur = make_tile(mt, nt);
    issue(cur, 0, 0); issue(cur, 1, 1); issue(cur, 2, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stage 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // group 1 starts one section later
    for (;;) {
        int v2 = v + nper, mt2 = 0, nt2 = 0;
        while (v2 < vmax && !wide_tile(xcd, v2, mtiles, ntn, mt2, nt2)) v2 += nper;
        const bool has_next = v2 < vmax;
        const Tile nxt = has_next ? make_tile(mt2, nt2) : cur;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < stages; ++t) {
            // ---- LOAD(t)
#ifndef SWV2_WIDE_NO_DMA            // (timing ablations, tools/ab_build.sh: wrong results)
            if (t + 3 < stages) issue(cur, t + 3, (t + 3) & 3);
            else issue(nxt, has_next ? t + 3 - stages : stages - 1, (t + 3) & 3);
#endif
            const uint16_t* As = smem + (t & 3) * WSTG;
            const uint16_t* Bs = As + WBM * 32;
            bf16x8 af[8], bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const bf16x8*)(Bs + swz32(wc * 64 + j * 16 + fr, g));
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(As + swz32(wr * 128 + i * 16 + fr, g));
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA(t)
            __builtin_amdgcn_s_setprio(1);
#ifdef SWV2_WIDE_NO_MMA
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(af[i]), "v"(bf[i & 3]));
#else
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
#endif
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- tile end
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            float* st = (float*)(smem + 3 * WSTG) + wave * 16 * EP;
#ifndef SWV2_WIDE_NO_EPI            // (timing ablation)
            if (cur.n_base + wc * 64 < N) wide_epilogue<EK>(ep, acc, st, cur.m_base + wr * 128, cur.n_base + wc * 64, lane);      // (N: a multiple of 64)
#else
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(acc[i][0]), "v"(acc[i][1]), "v"(acc[i][2]), "v"(acc[i][3]));
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging reads are done before slot 3 is handed back
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        cur = nxt; v = v2;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();             // (every wave passes the same number of barriers)
}

// ------------------------------------------------------------------------------------------------
// Wide TN kernel: weight-gradient partial sums  P_s[n][k] = sum over the rows m of slice s of dY[m][n] X[m][k]  for the wide widths,
// 256 x 256 output tiles, both operands raw bf16 (row-major rows or the head-major layout) by LDS-DMA.  The pipeline is the one
// of gemm_nt_wide_dma_kernel (four stages of 32 rows, loads three stages ahead, two wave groups one barrier apart, counted
// vmcnt); what differs is the contraction index: it is the ROW index of both operands, so a stage is a [32][256] slab of each
// and every fragment is a transposed LDS read (ds_read_b64_tr_b16: two per K = 32 operand, rows 0-15 and 16-31 of the slab;
// the same k order on both sides).
//   LDS stage: dY[32][256] | X[32][256] bf16, 512-byte rows; the 32-byte unit u of row r sits at unit u ^ (r & 7) of its 256-byte
//   half, so the 8 rows x 32 bytes a transposed read touches per 32 lanes fall on 8 different bank groups.  The DMA writes
//   linearly and fetches the permuted source chunk.
//   Rows: the M / 32 stages are cut into S contiguous slices, one workgroup per (slice, tile), S * tiles <= 256: one round.
//   Workgroups of one XCD hold consecutive (slice, tile) indices: about one slice per XCD, whose tiles share the slabs in its L2.
//   Bias gradient (column sums of dY): the workgroup with column tile tk adds up the slab columns of the stages t with
//   t mod ntk = tk from LDS -- 2 ds_read_b128 per thread on 1 / ntk of the stages, shared evenly by all workgroups of a row of
//   tiles -- and writes a partial row; swv2's reduction adds slices and column tiles.
// The partial tiles go to the workspace as S plain [N][K] fp32 matrices through wide_epilogue (whole-line stores).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_pchunk(int row, int c) { return ((((c >> 1) ^ (row & 7)) << 1) | (c & 1)); }    // involution per row
template <int YK, int XK>
__global__ __launch_bounds__(WTH) void gemm_tn_wide_kernel(ALoad<YK> yl, ALoad<XK> xl, float* __restrict__ part, float* __restrict__ dbpart,
                                                           int M, int N, int K, int S, int wgs, int per_xcd) {
    static_assert((YK == A_BF16 || YK == A_HEADS) && (XK == A_BF16 || XK == A_HEADS), "raw bf16 operands");
    constexpr int SLABE = 32 * 256;                          // elements per operand slab
    __shared__ __attribute__((aligned(1024))) unsigned char smem_b[WSM_BYTES];
    uint16_t* const smem = (uint16_t*)smem_b;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;              // 2 x 4 waves, 128 (n) x 64 (k) each
    const int w = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || w >= wgs) return;
    const int ntk = K / WBN, tiles = (N / WBM) * ntk;
    const int slice = w / tiles, tile = w - slice * tiles, tn = tile / ntk, tk = tile - tn * ntk;
    const int n_base = tn * WBM, k_base = tk * WBN;
    const int T = M / 32, t0 = (int)((long)slice * T / S), t1 = (int)((long)(slice + 1) * T / S), stages = t1 - t0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // DMA: instruction i (of 2 per operand) of this wave fills slab rows 4 wave + 2 i, + 1; lane -> (row + (lane >> 5), physical chunk lane & 31)
    int drw[2], dcl[2];
    uint32_t yoff[2], xoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        drw[i] = wave * 4 + 2 * i + (lane >> 5);
        dcl[i] = tn_pchunk(drw[i], lane & 31) * 8;                                   // logical column (elements) inside the slab
        yoff[i] = YK == A_BF16 ? 2u * (uint32_t)((t0 * 32 + drw[i]) * (int)yl.d.ld + n_base + dcl[i]) : 0u;
        xoff[i] = XK == A_BF16 ? 2u * (uint32_t)((t0 * 32 + drw[i]) * (int)xl.d.ld + k_base + dcl[i]) : 0u;
    }
    const long ystep = 32 * (long)yl.d.ld, xstep = 32 * (long)xl.d.ld;              // elements per stage (row-major operands)
    auto issue = [&](int t, int slot) {          // t = stage index inside the slice
        const uint32_t ly = lds0 + (uint32_t)(slot * WSTG * 2) + (uint32_t)(wave * 2048), lx = ly + SLABE * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (YK == A_BF16) dma_x4((const uint16_t*)yl.d.ptr + t * ystep, yoff[i], ly + 1024 * i);
            else dma_x4(yl.d.ptr, 2u * yl.elem_off((t0 + t) * 32 + drw[i], n_base + dcl[i]), ly + 1024 * i);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (XK == A_BF16) dma_x4((const uint16_t*)xl.d.ptr + t * xstep, xoff[i], lx + 1024 * i);
            else dma_x4(xl.d.ptr, 2u * xl.elem_off((t0 + t) * 32 + drw[i], k_base + dcl[i]), lx + 1024 * i);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposed fragment of slab columns c0 .. c0 + 15, rows r0 .. r0 + 15: lane (g, fr) <- rows r0 + 4 g .. + 3 of column c0 + fr
    auto trf = [&](const uint16_t* slab, int r0, int c0) -> bf16x4 {
        const int row = r0 + 4 * g + (fr >> 2), col = c0 + (fr & 3) * 4;
        return lds_tr_read(slab + row * 256 + tn_pchunk(row, col >> 3) * 8 + (col & 7));
    };
    // bias-gradient partial: thread -> (slab row tid >> 5 (+ 16), logical chunk tid & 31): the same 8 columns in every stage
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int cs_off = (tid >> 5) * 256 + tn_pchunk(tid >> 5, tid & 31) * 8;      // (rows r and r + 16 share r & 7)
    const bool want_db = dbpart != nullptr;

    if (stages > 0) {
        issue(0, 0); issue(min(1, stages - 1), 1); issue(min(2, stages - 1), 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one section behind (see gemm_nt_wide_dma_kernel)
        for (int t = 0; t < stages; ++t) {
            // ---- LOAD(t)
            issue(min(t + 3, stages - 1), (t + 3) & 3);
            const uint16_t* Ys = smem + (t & 3) * WSTG;
            const uint16_t* Xs = Ys + SLABE;
            bf16x8 af[8], bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = __builtin_shufflevector(trf(Xs, 0, wc * 64 + 16 * j), trf(Xs, 16, wc * 64 + 16 * j), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                af[i] = __builtin_shufflevector(trf(Ys, 0, wr * 128 + 16 * i), trf(Ys, 16, wr * 128 + 16 * i), 0, 1, 2, 3, 4, 5, 6, 7);
            if (want_db && ((t0 + t) % ntk) == tk) {           // (workgroup-uniform)
                const uint4 a = *(const uint4*)(Ys + cs_off), b = *(const uint4*)(Ys + cs_off + 16 * 256);
                float va[8], vb[8];
                unpack8(a, va); unpack8(b, vb);
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += va[e] + vb[e];
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA(t)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                               // every wave is done with the slots: staging below
    // ---- partial tile -> workspace matrix of this slice (rows n, columns k)
    Epi<E_F32> ep;
    ep.d.out = part + (size_t)slice * N * K; ep.d.bias = nullptr; ep.d.aux = nullptr; ep.d.aux_out = nullptr; ep.d.rowidx = nullptr;
    ep.d.ld = K; ep.d.M = N; ep.d.N = K;
    wide_epilogue<E_F32>(ep, acc, (float*)smem + wave * 16 * EP, n_base + wr * 128, k_base + wc * 64, lane);
    if (want_db) {                                 // 16 row-threads per chunk -> 256 column sums -> dbpart[slice][tk][n]
        __syncthreads();
        float* red = (float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * 256 + (tid & 31) * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 256 + tid];
            dbpart[((size_t)slice * ntk + tk) * N + n_base + tid] = sum;
        }
    }
}

// ---- register-staged A (fp32 rows, gathered rows with zero rows): weights by DMA, two 64-k buffers, one tile per workgroup ----
template <int AK, int EK>
__global__ __launch_bounds__(WTH) void gemm_nt_wide_kernel(ALoad<AK> al, const uint16_t* __restrict__ Wb, Epi<EK> ep,
                                                           int M, int N, int K, int mtiles, int ntn) {
    static_assert(!ALoad<AK>::ROW_FASTEST, "row-major staging map only");
    constexpr int TILE = (WBM + WBN) * BK;                 // elements per 64-k buffer (A | B)
    __shared__ __attribute__((aligned(1024))) uint16_t smem[2 * TILE];
    static_assert(2 * TILE * 2 >= 8 * 16 * EP * 4, "the epilogue staging re-uses the tile buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;              // 2 x 4 waves, 128 x 64 each
    int mt, nt;
    if (!wide_tile(blockIdx.x & 7, blockIdx.x >> 3, mtiles, ntn, mt, nt)) return;
    const int m_base = mt * WBM, n_base = nt * WBN;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int ksteps = K / BK;
        // DMA geometry (weights): one wave instruction fills 8 tile rows x 128 bytes; instruction i of this wave covers tile
        // rows 32 wave + 8 i .. + 7; lane -> (row + (lane >> 3), physical chunk lane & 7), which holds logical chunk (lane & 7) ^ swizzle
        const int drow = wave * 32 + (lane >> 3);
        uint32_t boff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = drow + 8 * i, kcl = (lane & 7) ^ ((row >> 1) & 7);
            boff[i] = 2u * (uint32_t)((n_base + row) * K + kcl * 8);
        }
        // register staging of A: 4 chunks of 16 bytes per thread per k-step; chunk c = tid + 512 i -> (row c >> 3, chunk column c & 7)
        typename ALoad<AK>::Raw ra[4];
        int arow[4];
        uint32_t aok = 0;
        const int kc = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = al.row_of(m_base + (tid >> 3) + 64 * i);
            arow[i] = max(src, 0);
            aok |= (src >= 0 ? 1u : 0u) << i;
        }
        auto issue = [&](int s, int buf) {
            const uint32_t lb = lds0 + (uint32_t)(buf * TILE * 2) + (uint32_t)(wave * 4096) + WBM * BK * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_x4(Wb + s * BK, boff[i], lb + 1024 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = al.raw_unc(arow[i], s * BK + kc * 8);
        };
        auto commit = [&](int buf) {
            uint16_t* As = smem + buf * TILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = al.cvt(ra[i]);
                if (!((aok >> i) & 1)) v = make_uint4(0, 0, 0, 0);
                *(uint4*)(As + swz((tid >> 3) + 64 * i, kc)) = v;
            }
        };
        auto compute = [&](int buf) {
            const uint16_t* As = smem + buf * TILE;
            const uint16_t* Bs = As + WBM * BK;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const bf16x8*)(Bs + swz(wc * 64 + j * 16 + fr, kk * 4 + g));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bf16x8 af = *(const bf16x8*)(As + swz(wr * 128 + i * 16 + fr, kk * 4 + g));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(af, bf[j], acc[i][j]);
                }
            }
        };
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        commit(0);
        __syncthreads();
        for (int s = 0; s + 1 < ksteps; ++s) {
            issue(s + 1, (s + 1) & 1);
            // (without the fences the scheduler sinks the loads behind the MFMAs, next to the waits of commit, or re-issues them
            // there: found in the ISA, 7 400 cycles per k-step against 2 048 of MFMA work)
            __builtin_amdgcn_sched_barrier(0);
            compute(s & 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA'd weight tile of step s + 1 has landed
            commit((s + 1) & 1);
            __syncthreads();           // step s + 1 is complete in its buffer; every wave is done reading buffer s & 1
        }
        compute((ksteps - 1) & 1);
        __syncthreads();               // the staging below overwrites the tile buffers
    }
    wide_epilogue<EK>(ep, acc, (float*)smem + wave * 16 * EP, m_base + wr * 128, n_base + wc * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// Resident-weight kernel for the two per-block row-streaming products at the benchmark width (qkv: N = 384, K = 128 and
// d(qkv) -> dx: N = 128, K = 384).  gemm_nt_kernel above is latency-bound there (measured: ~20 us workgroup lifetime for
// 64 rows = one HBM round trip for the A panel + six dependent L2 round trips for the weight tiles + three epilogues, at
// three workgroups per CU -> 3 TB/s).  Here ONE persistent 8-wave workgroup per CU keeps the whole bf16 weight (96 KB) in
// LDS, walks row tiles t = blockIdx.x, + gridDim.x, ... and always has the NEXT tile's A rows in flight in registers while
// the current tile runs its MFMAs and its epilogue: per tile one A commit, no weight traffic, two barriers.
//   LDS: W[N][K] (swizzled 16-byte chunks) | A tile [BMT][K] bf16, re-used as the waves' epilogue staging
//   waves: 4 row groups x 2 column halves; wave tile = (BMT / 4) x (N / 2)
// ------------------------------------------------------------------------------------------------
#ifdef SWV2_RW_STAMPS
__device__ unsigned long long rw_stamps[256 * 8];
#define RSTAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RSTAMP_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory"); } while (0)
#define RSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); \
                       st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define RSTAMP_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define RSTAMP_DECL
#define RSTAMP_START() do {} while (0)
#define RSTAMP(k) do {} while (0)
#define RSTAMP_WAITV() do {} while (0)
#endif
__device__ __forceinline__ int swzk(int r, int kc, int K) { return r * K + ((kc ^ ((r >> 1) & 7)) << 3); }

template <int AK, int EK, int N, int K, int BMT, int HS = 16>
__global__ __launch_bounds__(512) void gemm_rw_kernel(ALoad<AK> al, const uint16_t* __restrict__ Wb, Epi<EK> ep, int M) {
    constexpr int NTH = 512, KC = K / 8;                    // 16-byte chunks per row
    constexpr int ACH = BMT * KC / NTH;                     // A chunks per thread per tile
    constexpr int RT = BMT / 64;                            // 16-row MFMA tiles per wave (4 row groups)
    constexpr int CT = N / 32;                              // 16-column MFMA tiles per wave (2 column halves)
    constexpr int WCH = N * KC / NTH;                       // weight chunks per thread (one-time load)
    constexpr int A_BYTES = BMT * K * 2, ST_BYTES = 8 * 16 * EP * 4;
    constexpr int AS_BYTES = A_BYTES > ST_BYTES ? A_BYTES : ST_BYTES;
    static_assert(BMT * KC % NTH == 0 && N * KC % NTH == 0 && K % 64 == 0 && N % 64 == 0, "tile shapes");
    static_assert(N * K * 2 + AS_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) uint16_t Ws[N * K];
    __shared__ __attribute__((aligned(16))) unsigned char as_raw[AS_BYTES];
    // The epilogue must not LOAD from global memory: such a load sits behind the next tile's A prefetch in the in-order
    // vmcnt queue, so waiting for it exposes the whole HBM round trip of the prefetch right there (measured with the
    // generic Epi::tile: 52 % / 40 % of the kernel inside the epilogue, 6 % in the explicit wait for A).  The bias lives
    // in LDS; the scatter rows and the residual rows of the d(qkv) -> dx product are prefetched WITH the A rows.
    __shared__ __attribute__((aligned(16))) float bs[N];
    uint16_t* As = (uint16_t*)as_raw;
    float* stage_all = (float*)as_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntiles = (M + BMT - 1) / BMT;
    static_assert(EK != E_F32 || (RT == 1 && CT == 4), "the E_F32 prefetch below covers one 16 x 64 wave tile");
    for (int i = tid; i < N; i += NTH) bs[i] = ep.d.bias ? ep.d.bias[i] : 0.f;
    // E_F32: this lane's epilogue row (lane >> 2 of the wave's 16) and its 16 columns
    int erow_nxt = -1, erow_cur = -1;                       // destination row of the tile in flight / being computed
    f32x4 aux_nxt[4], aux_cur[4];
    const int ecol = wc * (N / 2) + (lane & 3) * 16;

    // ONE tile of A rows in flight per workgroup in registers.  (Two sets used alternately were measured SLOWER, 60.5 vs
    // 54.1 us for qkv: the kernel is not short of outstanding bytes.)  The gather rows of tile t + 2G are resolved while
    // tile t + G's data loads are issued, so the index load is never in front of a data load.
    typename ALoad<AK>::Raw ra[ACH];
    int arow[ACH];
    int erow_res = -1;
    // chunk c of a tile -> (row, 16-byte chunk column).  Row-major operands: consecutive threads walk a row.  Head-major
    // operand ([Bw][h][S][Lp][16]: one (row, head) is 32 bytes, consecutive ROWS of a head are contiguous): consecutive
    // threads take the two halves of a head for consecutive rows, so a wave reads 32 rows x 32 bytes = 1 KB contiguous
    // (row-major order gave 32 separate 32-byte pieces per wave instruction).
    auto crow = [&](int c) { return AK == A_HEADS ? ((c >> 1) & (BMT - 1)) : c / KC; };
    auto ccol = [&](int c) { return AK == A_HEADS ? (((c / (2 * BMT)) << 1) | (c & 1)) : c % KC; };
    static_assert(AK != A_HEADS || (BMT & (BMT - 1)) == 0, "head-major chunk map needs a power-of-two tile height");
    // Every load of the staging pipeline is UNCONDITIONAL (clamped address); validity travels in bit masks and invalid
    // chunks (rows past M, padded window rows) are zeroed when the tile is written to LDS.  Loads under `if` made the
    // compiler guard their destination registers with s_waitcnt vmcnt(0) right behind them, which turned the prefetch into
    // a synchronous load (see gemm_tn.hip).  The launcher guarantees the tables this kernel reads unconditionally: a gather
    // table for the fp32 operand, a scatter table + residual for the E_F32 epilogue.
    uint32_t ain = 0;                 // bit i: chunk row i of the next issue lies inside [0, M)
    uint32_t aok = 0;                 // chunks in flight: bit i = valid
    bool ein = false;                 // E_F32: this lane's epilogue row of the next tile lies inside [0, M)
    auto resolve = [&](int t) {
        ain = 0;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * NTH, m = t * BMT + crow(c);
            const bool in = (t < ntiles) && (m < M);
            ain |= (uint32_t)in << i;
            if constexpr (AK == A_F32) arow[i] = al.d.rowidx[min(m, M - 1)];        // gather table entry (may be -1: padded row)
            else arow[i] = min(m, M - 1);
        }
        if constexpr (EK == E_F32) {
            const int m = t * BMT + wr * 16 + (lane >> 2);
            ein = (t < ntiles) && (m < M);
            erow_res = ein ? ep.d.rowidx[m] : -1;           // (conditional, like the A rows of this product: measured faster)
        }
    };
    auto issue = [&]() {
        aok = 0;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * NTH;
            const int r = ((ain >> i) & 1) ? arow[i] : -1;
            // (the head-major operand of the dx product keeps the conditional load: measured in situ on one box, unconditional
            // 61.2 us vs conditional 55.7 us for that kernel, while the gathered fp32 operand of qkv gains, 47.9 -> 44.6 us)
            if constexpr (AK == A_HEADS) ra[i] = al.raw_at(r, ccol(c) * 8);
            else ra[i] = al.raw_unc(max(r, 0), ccol(c) * 8);
            aok |= (uint32_t)(r >= 0) << i;
        }
        if constexpr (EK == E_F32) {
            erow_nxt = ein ? erow_res : -1;
            const float* ax = (const float*)ep.d.aux + (long)max(erow_nxt, 0) * ep.d.ld + ecol;
#pragma unroll
            for (int i = 0; i < 4; ++i) aux_nxt[i] = *(const f32x4*)(ax + 4 * i);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + i * NTH;
            *(uint4*)(As + swzk(crow(c), ccol(c), K)) = ((aok >> i) & 1) ? al.cvt(ra[i]) : make_uint4(0, 0, 0, 0);
        }
    };
    const int G = gridDim.x;
    RSTAMP_DECL
    auto tile_step = [&](int t) {
        RSTAMP_WAITV();
        RSTAMP(0);
        commit();
        if constexpr (EK == E_F32) {
            erow_cur = erow_nxt;
#pragma unroll
            for (int i = 0; i < 4; ++i) aux_cur[i] = aux_nxt[i];
        }
        RSTAMP(1);
        __syncthreads();                                   // A tile (and, the first time, the weight) visible
        RSTAMP(2);
        issue();                                           // next tile's rows: in flight during the MFMAs and the epilogue
        resolve(t + 2 * G);
        RSTAMP(3);
        f32x4 acc[RT][CT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < K / 32; ++ks) {
            bf16x8 af[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = *(const bf16x8*)(As + swzk(wr * 16 * RT + i * 16 + fr, ks * 4 + g, K));
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const bf16x8 bf = *(const bf16x8*)(Ws + swzk(wc * (N / 2) + j * 16 + fr, ks * 4 + g, K));
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][j] = mfma32(af[i], bf, acc[i][j]);
            }
        }
        asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[RT - 1][CT - 1][3]));
        RSTAMP(4);
        __syncthreads();                                   // every wave is done with the A tile: it becomes the staging area
        RSTAMP(5);
        float* st = stage_all + wave * 16 * EP;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j0 = 0; j0 < CT; j0 += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[(4 * g + r) * EP + 16 * j + fr] = acc[i][j0 + j][r];
                const int m0 = t * BMT + wr * 16 * RT + i * 16, n0 = wc * (N / 2) + j0 * 16;
                if constexpr (EK == E_QKV_HEADS) {
                    // heads_item<HS, true> with the bias from LDS: lane = (row r, 16-column slot); HS = 32: a head is two slots
                    // (adjacent lane groups share the squared norm).  d.p1 = first q / k / v part of this launch's columns (a
                    // launch over one part of a split weight: BASELINE configs[4], see launch_nt2)
                    const EpiDesc& d = ep.d;
                    const int h = d.p0, Lp = d.p2, L = d.p4;
                    const int r = lane & 15, slot = lane >> 4, nb = n0 + slot * 16, m = m0 + r;
                    if (m < d.M) {
                        const int ph = HS == 16 ? nb >> 4 : nb >> 5, lp = (h == 1) ? ph : fdiv(ph, h, d.mg1), hd = ph - lp * h;
                        const int part = lp + d.p1;
                        const int bw = fdiv(m, Lp, d.mg0), tt = m - bw * Lp;
                        const bool valid = tt < L;
                        float v[16];
                        float ss = 0.f;
#pragma unroll
                        for (int j = 0; j < 16; j += 4) {
                            const f32x4 x = *(const f32x4*)(st + r * EP + slot * 16 + j) + *(const f32x4*)(bs + nb + j);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[j + e] = valid ? x[e] : 0.f;
                                ss = fmaf(v[j + e], v[j + e], ss);
                            }
                        }
                        if constexpr (HS == 32) ss += __shfl_xor(ss, 16);      // (rows m and m past d.M differ only in lane bits 0 .. 3)
                        float rn = 1.f;
                        if (part < 2) {
                            rn = 1.f / fmaxf(sqrtf(ss), 1e-12f);
                            if (HS == 16 || !(slot & 1)) d.aux_out[(((long)bw * h + hd) * 2 + part) * Lp + tt] = valid ? rn : 0.f;
                        }
                        uint16_t* o = (uint16_t*)d.out + ((((long)bw * h + hd) * 3 + part) * Lp + tt) * HS + (HS == 32 ? (slot & 1) * 16 : 0);
                        float w8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] = v[e] * rn;
                        *(uint4*)o = pack8(w8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] = v[8 + e] * rn;
                        *(uint4*)(o + 8) = pack8(w8);
                    }
                } else if constexpr (EK == E_F32) {
                    // Epi<E_F32>::tile with the scatter row and the residual prefetched (N == 128: no column tail)
                    if (erow_cur >= 0) {
                        float* o = (float*)ep.d.out + (long)erow_cur * ep.d.ld + ecol;
                        const int r = lane >> 2, c0 = (lane & 3) * 16;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(f32x4*)(o + 4 * q) = *(const f32x4*)(st + r * EP + c0 + 4 * q) + *(const f32x4*)(bs + ecol + 4 * q) + aux_cur[q];
                    }
                } else {
                    ep.tile(st, m0, n0, lane);
                }
            }
        RSTAMP(6);
        __syncthreads();                                   // staging consumed before the next commit overwrites it
        RSTAMP(7);
    };
    int t = blockIdx.x;
    resolve(t);
    issue();                                               // first tile's rows in flight before the weight is fetched
    resolve(t + G);
    // ---- the weight, once (made visible by the first tile's barrier)
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = tid + i * NTH, n = c / KC, kc = c - n * KC;
        *(uint4*)(Ws + swzk(n, kc, K)) = *(const uint4*)(Wb + (long)n * K + kc * 8);
    }

    RSTAMP_START();
    for (; t < ntiles; t += G) tile_step(t);
#ifdef SWV2_RW_STAMPS
    if (tid == 0)
        for (int k = 0; k < 8; ++k) rw_stamps[blockIdx.x * 8 + k] = st_acc[k];
#endif
}

template <int AK, int EK>
int launch_nt2(const swv2_operand* a, const void* w, const swv2_epilogue* e, int M, int N, int K, hipStream_t st) {
    Epi<EK> ep;
    ep.d.out = e->out; ep.d.bias = e->bias; ep.d.aux = e->aux; ep.d.aux_out = e->aux_out; ep.d.rowidx = e->rowidx;
    ep.d.ld = e->ld; ep.d.M = M; ep.d.N = N;
    ep.d.p0 = e->p[0]; ep.d.p1 = e->p[1]; ep.d.p2 = e->p[2]; ep.d.p3 = e->p[3]; ep.d.p4 = e->p[4];
    ep.d.mg0 = ep.d.mg1 = ep.d.mg2 = 0;
    ep.d.loss_tar = nullptr; ep.d.loss_qw = nullptr; ep.d.loss_part = nullptr; ep.d.loss_resid = nullptr; ep.d.q0 = ep.d.q1 = ep.d.q2 = ep.d.q3 = 0;
    if (EK == E_QKV_HEADS || EK == E_HEADS) { ep.d.mg0 = fdiv_magic(e->p[2]); ep.d.mg1 = fdiv_magic(e->p[0]); ep.d.mg2 = fdiv_magic(e->p[3]); }
    if (EK == E_UNPATCH || EK == E_UNPATCH_LOSS || EK == E_UNPATCH_LOSS_SKIP) { ep.d.mg0 = fdiv_magic((e->p[1] / 4) * (e->p[2] / 4)); ep.d.mg1 = fdiv_magic(e->p[2] / 4); }
    if (EK == E_UNPATCH_LOSS || EK == E_UNPATCH_LOSS_SKIP) {
        ep.d.loss_tar = e->loss_tar; ep.d.loss_qw = e->loss_qw; ep.d.loss_part = e->loss_part; ep.d.loss_resid = (uint16_t*)e->loss_resid;
        ep.d.q0 = e->q[0]; ep.d.q1 = e->q[1];
        const long nb = M / ((e->p[1] / 4) * (e->p[2] / 4)), plane = (long)e->p[1] * e->p[2];
        // dump areas of the masked lanes: behind the dense prediction, or where the caller says (q[2] floats from `out`: rollouts, where
        // `out` points into a larger tensor); behind the second destination
        ep.d.q2 = e->q[2] ? e->q[2] : (int)(nb * e->p[0] * plane);
        ep.d.q3 = (int)(nb * e->ld * plane);
    }
    // the two per-block products at the benchmark width: resident-weight persistent kernel
    static const int rw = getenv("SWV2_GEMM_RW") ? atoi(getenv("SWV2_GEMM_RW")) : 1;
    if constexpr (AK == A_F32 && EK == E_QKV_HEADS) {
        if (rw && N == 384 && K == 128 && M >= 256 * 128 && e->p[3] == 16 && a->rowidx) {
            Epi<EK> ep1 = ep;
            ep1.d.p1 = 0;                                 // (first part of the launch's columns; p[1] is not a parameter of this epilogue)
            hipLaunchKernelGGL((gemm_rw_kernel<AK, EK, 384, 128, 128>), dim3(256), dim3(512), 0, st, make_loader<AK>(a),
                               (const uint16_t*)w, ep1, M);
            SWV2_CHECK_LAUNCH("swv2_linear");
            return SWV2_OK;
        }
        // BASELINE configs[4] (192 channels, 8 heads in 32-wide slots): the 768 x 192 weight does not fit LDS beside an A tile, one
        // of its q / k / v parts (256 x 192 = 96 KB) does: three launches, each over all rows (the gathered fp32 rows are read three
        // times, 3 x 108 MB, against the generic kernel's re-read of the weight per 64-row tile: 207 -> 3 x ~45 us)
        if (rw && N == 768 && K == 192 && M >= 256 * 128 && e->p[3] == 32 && e->p[0] == 8 && a->rowidx) {
            for (int part = 0; part < 3; ++part) {
                Epi<EK> ep2 = ep;
                ep2.d.N = 256;
                ep2.d.p1 = part;
                ep2.d.bias = e->bias ? e->bias + 256 * part : nullptr;
                hipLaunchKernelGGL((gemm_rw_kernel<AK, EK, 256, 192, 128, 32>), dim3(256), dim3(512), 0, st, make_loader<AK>(a),
                                   (const uint16_t*)w + (size_t)part * 256 * 192, ep2, M);
            }
            SWV2_CHECK_LAUNCH("swv2_linear");
            return SWV2_OK;
        }
    }
    if constexpr (AK == A_HEADS && EK == E_F32) {
        if (rw && N == 128 && K == 384 && M >= 256 * 64 && e->rowidx && e->aux && !a->rowidx) {
            hipLaunchKernelGGL((gemm_rw_kernel<AK, EK, 128, 384, 64>), dim3(256), dim3(512), 0, st, make_loader<AK>(a),
                               (const uint16_t*)w, ep, M);
            SWV2_CHECK_LAUNCH("swv2_linear");
            return SWV2_OK;
        }
    }
    // wide products (K, N >= 512, N a multiple of 256): 256 x 256 tiles
    if constexpr ((AK == A_F32 || AK == A_BF16 || AK == A_HEADS) &&
                  (EK == E_BF16 || EK == E_F32 || EK == E_F32_ACC || EK == E_QKV_HEADS || EK == E_HEADS || EK == E_GELU_GRAD || EK == E_BF16_GELU)) {
        const int wide = getenv("SWV2_GEMM_WIDE") ? atoi(getenv("SWV2_GEMM_WIDE")) : 1;      // (read per call: the tests toggle it)
        // BASELINE configs[4]'s d(qkv) -> dx product (N = 192, K = 768, head-major operand): one partial column tile of the DMA kernel --
        // the operand is read exactly once, the weight's 192 rows stream from L2; a quarter of the MFMAs multiply clamped rows
        // (the 64-row tile kernel re-reads the weight per tile behind a barrier per 64 k: 165 us)
        const bool part192 = AK == A_HEADS && EK == E_F32 && N == 192 && K == 768 && !a->rowidx && M >= 64 * WBM &&
                             (double)a->rows * a->cols * 2 < 4.29e9;                  // (only the DMA kernel clamps the weight rows)
        if (wide && (N % WBN == 0 || part192) && K % BK == 0 && (N >= 512 || part192) && K >= 512 && M >= 16 * WBM) {
            const int mtiles = cdiv(M, WBM), ntn = cdiv(N, WBN), groups = cdiv(mtiles, 8);
            const int grid = 8 * cdiv(groups, 8) * 8 * ntn;
            // 32-bit byte offsets in the DMA addressing: weights and operand below 4 GB (the operand: checked for its kind below)
            SWV2_CHECK_ARG((double)N * K * 2 < 4.29e9, "swv2_linear: weight too large for the wide kernel's 32-bit offsets");
            const bool small = (double)a->rows * (AK == A_HEADS ? a->cols : a->ld) * 2 < 4.29e9;
            constexpr bool CAN_DMA = AK == A_BF16 || AK == A_HEADS;
            const int vmax = cdiv(groups, 8) * 8 * ntn;            // tile sequence length per XCD
            const int persist = getenv("SWV2_WIDE_PERSIST") ? atoi(getenv("SWV2_WIDE_PERSIST")) : 1;
            if (CAN_DMA && small && K % 128 == 0 && (AK == A_HEADS || !a->rowidx)) {
                if constexpr (CAN_DMA)
                    hipLaunchKernelGGL((gemm_nt_wide_dma_kernel<AK, EK>), dim3(8 * (vmax < 32 || !persist ? vmax : 32)), dim3(WTH), 0, st, make_loader<AK>(a),
                                       (const uint16_t*)w, ep, M, N, K, mtiles, ntn, vmax);
            } else {
                hipLaunchKernelGGL((gemm_nt_wide_kernel<AK, EK>), dim3(grid), dim3(WTH), 0, st, make_loader<AK>(a), (const uint16_t*)w,
                                   ep, M, N, K, mtiles, ntn);
            }
            SWV2_CHECK_LAUNCH("swv2_linear");
            return SWV2_OK;
        }
    }
    // 64-row workgroups where they fill the 768 slots (3 per CU) better; only for the two per-block products
    bool half = false;
    if constexpr ((AK == A_F32 && EK == E_QKV_HEADS) || (AK == A_HEADS && EK == E_F32)) {
        const double w1 = cdiv(M, BM), w2 = cdiv(M, BM / 2), slots = 768.0;
        const double e1 = w1 / (std::ceil(w1 / slots) * slots), e2 = w2 / (std::ceil(w2 / slots) * slots);
        static const int force = getenv("SWV2_GEMM_BM") ? atoi(getenv("SWV2_GEMM_BM")) : 0;
        half = force ? force == 64 : (e2 > e1 + 0.08);
        if (half)
            hipLaunchKernelGGL((gemm_nt_kernel<AK, EK, BM / 2>), dim3(cdiv(M, BM / 2)), dim3(NTHREADS), 0, st, make_loader<AK>(a),
                               (const uint16_t*)w, ep, M, N, K);
    }
    if constexpr (EK == E_UNPATCH_LOSS || EK == E_UNPATCH_LOSS_SKIP) {
        // 64-row workgroups (one 32-row partial-sum group per wave).  128-row workgroups (two groups per wave) measured equal -- 422.8 vs
        // 420.9 us -- with 23 registers spilled: not instantiated
        half = true;
        hipLaunchKernelGGL((gemm_nt_kernel<AK, EK, BM / 2>), dim3(cdiv(M, BM / 2)), dim3(NTHREADS), 0, st, make_loader<AK>(a),
                           (const uint16_t*)w, ep, M, N, K);
    }
    if constexpr (EK != E_UNPATCH_LOSS && EK != E_UNPATCH_LOSS_SKIP)
    if (!half)
        hipLaunchKernelGGL((gemm_nt_kernel<AK, EK>), dim3(cdiv(M, BM)), dim3(NTHREADS), 0, st, make_loader<AK>(a),
                           (const uint16_t*)w, ep, M, N, K);
    SWV2_CHECK_LAUNCH("swv2_linear");
    return SWV2_OK;
}

template <int AK>
int launch_nt1(const swv2_operand* a, const void* w, const swv2_epilogue* e, int M, int N, int K, hipStream_t st) {
    if constexpr (AK == A_BF16_CS) {               // the loss-gradient operand feeds exactly one product: d(e) = G W_head, fp32 out
        if (e->kind == SWV2_EPI_F32) return launch_nt2<AK, E_F32>(a, w, e, M, N, K, st);
        swv2_set_error("swv2_linear: SWV2_OP_BF16_CSCALE supports SWV2_EPI_F32 only (got %d)", e->kind);
        return SWV2_ERR_INVALID;
    } else {
    if constexpr (AK == A_F32) if (e->kind == SWV2_EPI_UNPATCH_LOSS)
        return e->p[3] ? launch_nt2<AK, E_UNPATCH_LOSS_SKIP>(a, w, e, M, N, K, st) : launch_nt2<AK, E_UNPATCH_LOSS>(a, w, e, M, N, K, st);
    switch (e->kind) {
        case SWV2_EPI_BF16: return launch_nt2<AK, E_BF16>(a, w, e, M, N, K, st);
        case SWV2_EPI_F32: return launch_nt2<AK, E_F32>(a, w, e, M, N, K, st);
        case SWV2_EPI_F32_ACC: return launch_nt2<AK, E_F32_ACC>(a, w, e, M, N, K, st);
        case SWV2_EPI_QKV_HEADS: return launch_nt2<AK, E_QKV_HEADS>(a, w, e, M, N, K, st);
        case SWV2_EPI_HEADS: return launch_nt2<AK, E_HEADS>(a, w, e, M, N, K, st);
        case SWV2_EPI_GELU_GRAD: return launch_nt2<AK, E_GELU_GRAD>(a, w, e, M, N, K, st);
        case SWV2_EPI_UNPATCH: return launch_nt2<AK, E_UNPATCH>(a, w, e, M, N, K, st);
        case SWV2_EPI_BF16_GELU: return launch_nt2<AK, E_BF16_GELU>(a, w, e, M, N, K, st);
    }
    swv2_set_error("swv2_linear: unknown epilogue kind %d (or not available for operand kind %d)", e->kind, a->kind);
    return SWV2_ERR_INVALID;
    }
}

}  // namespace

// launched by gemm_tn.hip (swv2_linear_wgrad_ws): partial matrices part[S][N][K] (+ dbpart[S][K / 256][N]); the operands are
// raw bf16 (SWV2_OP_BF16 without a gather table, SWV2_OP_HEADS), M % 32 == 0, N % 256 == 0, K % 256 == 0, S * tiles <= 256
int swv2_tn_wide_launch(const swv2_operand* y, const swv2_operand* x, float* part, float* dbpart, int M, int N, int K, int S, hipStream_t st) {
    const int tiles = (N / WBM) * (K / WBN), wgs = S * tiles, per_xcd = cdiv(wgs, 8);
#define SWV2_TNW(YK_, XK_)                                                                                                           \
    hipLaunchKernelGGL((gemm_tn_wide_kernel<YK_, XK_>), dim3(8 * per_xcd), dim3(WTH), 0, st, make_loader<YK_>(y), make_loader<XK_>(x), part, \
                       dbpart, M, N, K, S, wgs, per_xcd)
    if (y->kind == SWV2_OP_BF16 && x->kind == SWV2_OP_BF16) SWV2_TNW(A_BF16, A_BF16);
    else if (y->kind == SWV2_OP_BF16 && x->kind == SWV2_OP_HEADS) SWV2_TNW(A_BF16, A_HEADS);
    else if (y->kind == SWV2_OP_HEADS && x->kind == SWV2_OP_BF16) SWV2_TNW(A_HEADS, A_BF16);
    else { swv2_set_error("swv2_linear_wgrad: wide kernel: operand pair (%d, %d) not instantiated", y->kind, x->kind); return SWV2_ERR_UNSUPPORTED; }
#undef SWV2_TNW
    SWV2_CHECK_LAUNCH("swv2_linear_wgrad");
    return SWV2_OK;
}

#ifdef SWV2_RW_STAMPS
extern "C" int swv2_debug_rw_stamps(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rw_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int swv2_linear(const swv2_operand* a, const void* w_bf16, const swv2_epilogue* e, int N, void* stream) {
    int rc = check_operand(a, "swv2_linear");
    if (rc) return rc;
    SWV2_CHECK_ARG(w_bf16 && e && e->out && N > 0, "swv2_linear: null weight / epilogue / N");
    SWV2_CHECK_ARG(((uintptr_t)w_bf16 & 15) == 0 && ((uintptr_t)e->out & 15) == 0, "swv2_linear: unaligned pointer");
    if (e->kind == SWV2_EPI_BF16_GELU) SWV2_CHECK_ARG(e->aux_out != nullptr, "swv2_linear: GELU epilogue needs aux_out");
    if (e->kind == SWV2_EPI_BF16 || e->kind == SWV2_EPI_GELU_GRAD || e->kind == SWV2_EPI_BF16_GELU)
        SWV2_CHECK_ARG(e->ld % 8 == 0 && e->ld >= N, "swv2_linear: output pitch %ld must be a multiple of 8 and >= N", e->ld);
    if (e->kind == SWV2_EPI_F32 || e->kind == SWV2_EPI_F32_ACC)
        SWV2_CHECK_ARG(e->ld % 4 == 0 && e->ld >= N, "swv2_linear: output pitch %ld must be a multiple of 4 and >= N", e->ld);
    if (e->kind == SWV2_EPI_QKV_HEADS || e->kind == SWV2_EPI_HEADS)
        SWV2_CHECK_ARG((e->p[3] == 16 || e->p[3] == 32 || e->p[3] == 64 || e->p[3] == 96 || e->p[3] == 128) && N % e->p[3] == 0 && e->p[0] > 0 && e->p[2] > 0,
                       "swv2_linear: head-split epilogue needs DP in {16,32,64,96,128} and N a multiple of DP (DP=%d N=%d)", e->p[3], N);
    if (e->kind == SWV2_EPI_UNPATCH || e->kind == SWV2_EPI_UNPATCH_LOSS)
        SWV2_CHECK_ARG(N == e->p[0] * 16, "swv2_linear: un-patchify needs N == Cout*16");
    if (e->kind == SWV2_EPI_UNPATCH_LOSS) {
        SWV2_CHECK_ARG(a->kind == SWV2_OP_F32, "swv2_linear: the loss epilogue takes an fp32 operand");
        SWV2_CHECK_ARG((e->p[4] == 0 || (e->p[4] >= e->p[0] && e->q[2] > 0)) && (!e->aux_out || e->ld >= e->p[0]) && e->q[2] >= 0,
                       "swv2_linear: loss epilogue: p[4] (channels of the destination tensor) needs q[2] (its dump offset), aux_out needs ld >= Cout");
        SWV2_CHECK_ARG(e->loss_tar && e->loss_qw && e->loss_part && e->loss_resid && e->q[0] >= e->q[1] + e->p[0] && e->q[1] >= 0,
                       "swv2_linear: the loss epilogue needs target, quadrature weights, sums, residual and q[1] + Cout <= q[0]");
        SWV2_CHECK_ARG((((uintptr_t)e->loss_tar | (uintptr_t)e->loss_resid) & 15) == 0, "swv2_linear: unaligned loss pointer");
        const double plane_ = (double)e->p[1] * e->p[2], nb_ = (double)a->rows / ((e->p[1] / 4) * (e->p[2] / 4));
        SWV2_CHECK_ARG((e->p[1] / 4) * (e->p[2] / 4) >= SWV2_LOSS_GROUP_ROWS, "swv2_linear: the loss epilogue needs at least %d patches per sample", SWV2_LOSS_GROUP_ROWS);
        const double cmax_ = fmax(fmax((double)e->p[3], (double)e->p[0]), fmax((double)e->p[4], e->aux_out ? (double)e->ld : 0.0));
        SWV2_CHECK_ARG(nb_ * e->q[0] * plane_ < 4.29e9 && nb_ * cmax_ * plane_ + 1024 < 4.29e9 && (double)a->rows * SWV2_LOSS_RESID_PITCH(N) + 2048 < 4.29e9 && (double)e->q[2] < 4.29e9,
                       "swv2_linear: the loss epilogue indexes its tensors with 32-bit element offsets (tensor too large)");
    }
    const int M = a->rows, K = a->cols;
    hipStream_t st = (hipStream_t)stream;
    switch (a->kind) {
        case SWV2_OP_F32: return launch_nt1<A_F32>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_BF16: return launch_nt1<A_BF16>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_BF16_GELU: return launch_nt1<A_BF16_GELU>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_HEADS: return launch_nt1<A_HEADS>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_PATCH: return launch_nt1<A_PATCH>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_MERGE_LN: return launch_nt1<A_MERGE_LN>(a, w_bf16, e, M, N, K, st);
        case SWV2_OP_BF16_CSCALE: return launch_nt1<A_BF16_CS>(a, w_bf16, e, M, N, K, st);
    }
    swv2_set_error("swv2_linear: unknown operand kind %d", a->kind);
    return SWV2_ERR_INVALID;
}

