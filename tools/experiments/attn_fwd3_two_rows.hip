// RETIRED EXPERIMENT (round 5; VERDICT r4 item 3, hypothesis ii): attn_fwd3_kernel of csrc/attn2.hip with two query tiles of a wave in
// flight.  Parity-green (27 attention tests, bit-identical per row), in situ at local batch 2 on the same box: 57.8 vs 50.8 us per launch
// (step 9.605 vs 9.50 ms): the second row's 44 accumulator registers push the kernel to 256 registers + 49 in scratch at two workgroups per
// CU, and eight waves per CU hide less latency than twelve; the LDS reads saved (one K / V^T fragment read for two rows) were not the bound.
// Not compiled into the library.  To revive: paste into csrc/attn2.hip behind launch_fwd3 (it uses that file's helpers) and dispatch
// launch_fwd3r2<11, 162, 4, 2> from swv2_attn2_fwd.

// ------------------------------------------------------------------------------------------------
// attn_fwd3_kernel with TWO query tiles of a wave in flight (q tiles w and w + 4 of the item together, then w + 8): the K image and V^T
// fragments are read from LDS once for both rows (the LDS pipe is the busiest unit of attn_fwd3_kernel at three workgroups per CU: ~59 %
// by the read costs of tools/ubench_lds.hip), at the price of 88 accumulator registers: two workgroups per CU.  VERDICT r4 item 3 (ii).
// Selected with SWV2_ATTN_FWD3_R2=1; same arithmetic per row as attn_fwd3_kernel (bit-identical results).
// ------------------------------------------------------------------------------------------------
template <int LT, int LFIX, int WAVES, int OCC>
__global__ __launch_bounds__(64 * WAVES, OCC) void attn_fwd3r2_kernel(
    const uint16_t* __restrict__ qkvh, const float* __restrict__ logit_scale, uint16_t* __restrict__ oh, float* __restrict__ lse,
    int Bw, int h, int L, int nW, int nww, int nwh, int mask_thr) {
    constexpr int DP = 16, Lp = 16 * LT, SLAB = Lp * DP;
    constexpr int NT = 64 * WAVES;
    constexpr int CH = 2 * SLAB / 8;
    constexpr int CPT = (CH + NT - 1) / NT;
    constexpr int KIMG = 2 * SLAB, BUF = KIMG + 2 * SLAB;
    constexpr int QCH = SLAB / 8, QPT = (QCH + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int hd = blockIdx.y;
    const bool last_chunk_ok = (wave * 64 + (CPT - 1) * NT) < CH;
    const int Lc = LFIX > 0 ? LFIX : L;
    const float sc2 = __expf(fminf(logit_scale[hd], SWV2_LN100)) * SWV2_LOG2E;
    const bool bounded = sc2 <= 40.f;

    u32x4 stage[CPT], stageq[QPT];
    unsigned soff[CPT], qoff[QPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) soff[j] = (unsigned)(SLAB + min(tid + j * NT, CH - 1) * 8);
#pragma unroll
    for (int j = 0; j < QPT; ++j) qoff[j] = (unsigned)(min(tid + j * NT, QCH - 1) * 8);
    auto issue_loads = [&](int bw) {
        const uint16_t* base = qkvh + ((size_t)bw * h + hd) * 3 * SLAB;
#pragma unroll
        for (int j = 0; j < CPT; ++j) stage[j] = *(const u32x4*)(base + soff[j]);
#pragma unroll
        for (int j = 0; j < QPT; ++j) stageq[j] = *(const u32x4*)(base + qoff[j]);
    };
    auto write_stage = [&](int buf) {
        uint16_t* dst = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (j < CPT - 1 || last_chunk_ok) {
                const int c = tid + j * NT;
                if (c < CH / 2) {
                    const u32x4 lo = {stage[j][0], stage[j][1], stage[j][0], stage[j][1]};
                    const u32x4 hi = {stage[j][2], stage[j][3], stage[j][2], stage[j][3]};
                    *(u32x4*)(dst + (size_t)c * 16) = lo;
                    *(u32x4*)(dst + (size_t)c * 16 + 8) = hi;
                } else {
                    *(u32x4*)(dst + KIMG + (size_t)(c - CH / 2) * 8) = stage[j];
                }
            }
#pragma unroll
        for (int j = 0; j < QPT; ++j)
            if (tid + j * NT < QCH) *(u32x4*)(dst + KIMG + SLAB + (size_t)(tid + j * NT) * 8) = stageq[j];
    };

    int bw = blockIdx.x;
    if (bw >= Bw) return;
    issue_loads(bw);
    write_stage(0);
    __syncthreads();
    const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};

    for (int it = 0; bw < Bw; bw += gridDim.x, ++it) {
        const int buf = it & 1;
        const int bw_next = bw + gridDim.x;
        const uint16_t* Ki = smem + buf * BUF;
        const uint16_t* Vs = Ki + KIMG;
        const uint16_t* Qs = Vs + SLAB;
        if (bw_next < Bw) issue_loads(bw_next);
        const bool do_mask = (mask_thr > 0) && (((bw % nW) / nww) == nwh - 1);
        const bool fixed = bounded && !do_mask;
        const float c0 = fixed ? -sc2 : 0.f;
        f32x4 cpad;
#pragma unroll
        for (int r = 0; r < 4; ++r) cpad[r] = (16 * (LT - 1) + 4 * g + r < Lc) ? c0 : SWV2_NEG_BIG;

        // NR rows (q tiles qt0, qt0 + WAVES, ...) of this wave in one pass over the key tiles
        auto rows = [&](auto nr_c, const int qt0, const bool last) {
            constexpr int NR = decltype(nr_c)::value;
            if (last && bw_next < Bw) write_stage(buf ^ 1);
            bf16x8 qB[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int q = 16 * (qt0 + n * WAVES) + fr;
                const bf16x4 qraw = *(const bf16x4*)(Qs + (size_t)q * DP + 4 * g);
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = bf2f(qraw[j]) * sc2;
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    w[j] = f2bf2(x[2 * j], x[2 * j + 1]);
                    w[2 + j] = f2bf2(x[2 * j] - __uint_as_float(w[j] << 16), x[2 * j + 1] - __uint_as_float(w[j] & 0xffff0000u));
                }
                qB[n] = __builtin_bit_cast(bf16x8, w);
            }
            f32x4 acc[NR][LT];
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                const bf16x8 kA = *(const bf16x8*)(Ki + (16 * t + fr) * 32 + 8 * g);
                const f32x4 c = (t == LT - 1) ? cpad : (f32x4){c0, c0, c0, c0};
#pragma unroll
                for (int n = 0; n < NR; ++n) acc[n][t] = mfma32(kA, qB[n], c);
            }
            __builtin_amdgcn_sched_barrier(0);
            float mx[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                mx[n] = sc2;
                if (!fixed) {
                    const int q = 16 * (qt0 + n * WAVES) + fr;
                    float m_ = SWV2_NEG_BIG;
                    const bool qid = q >= mask_thr;
#pragma unroll
                    for (int t = 0; t < LT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (do_mask) acc[n][t][r] += (((16 * t + 4 * g + r) >= mask_thr) != qid) ? (-100.f * SWV2_LOG2E) : 0.f;
                            m_ = fmaxf(m_, acc[n][t][r]);
                        }
                    m_ = fmaxf(m_, __shfl_xor(m_, 16));
                    m_ = fmaxf(m_, __shfl_xor(m_, 32));
#pragma unroll
                    for (int t = 0; t < LT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[n][t][r] -= m_;
                    mx[n] = m_;
                }
#pragma unroll
                for (int t = 0; t < LT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (LFIX > 0 && 16 * t + r >= LFIX && t == LT - 1) acc[n][t][r] = 0.f;
                        else acc[n][t][r] = __builtin_amdgcn_exp2f(acc[n][t][r]);
                    }
            }
            f32x4 o[NR], rs[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) { o[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; rs[n] = o[n]; }
#pragma unroll
            for (int t = 0; t + 1 < LT; t += 2) {
                const bf16x4 v0 = lds_tr_read(Vs + (16 * t + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const bf16x4 v1 = lds_tr_read(Vs + (16 * (t + 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
                const bf16x8 vA = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    const bf16x4 p0 = f2bf4(acc[n][t]), p1 = f2bf4(acc[n][t + 1]);
                    const bf16x8 pb = __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[n] = mfma32(vA, pb, o[n]);
                    rs[n] = mfma32(ones8, pb, rs[n]);
                }
                __builtin_amdgcn_sched_barrier(0);        // (keeps the bf16 conversions next to their MFMAs: hoisted, they cost 44 registers)
            }
            if (LT & 1) {
                const bf16x4 vf = lds_tr_read(Vs + (16 * (LT - 1) + 4 * g + (fr >> 2)) * DP + (fr & 3) * 4);
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    const bf16x4 pb = f2bf4(acc[n][LT - 1]);
                    o[n] += mfma16(vf, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                    rs[n] += mfma16(ones4, pb, (f32x4){0.f, 0.f, 0.f, 0.f});
                }
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int q = 16 * (qt0 + n * WAVES) + fr;
                const float sum = rs[n][0];
                const float inv = (q < L) ? __builtin_amdgcn_rcpf(sum) : 0.f;
                uint16_t* orow = oh + ((size_t)bw * h + hd) * SLAB + (size_t)q * DP;
                f32x4 v = o[n];
                v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                *(bf16x4*)(orow + 4 * g) = f2bf4(v);
                if (g == 0) lse[((size_t)bw * h + hd) * Lp + q] = (q < L) ? mx[n] + __log2f(sum) : 0.f;
            }
        };
        // q tiles w and w + WAVES together; then w + 2 WAVES if the wave has one (waves 0 .. LT - 2 WAVES - 1)
        const bool third = wave + 2 * WAVES < LT;
        rows(std::integral_constant<int, 2>{}, wave, !third);
        if (third) rows(std::integral_constant<int, 1>{}, wave + 2 * WAVES, true);
        __syncthreads();
    }
}

template <int LT, int LFIX, int WAVES, int OCC>
int launch_fwd3r2(const swv2_attn_args* a, hipStream_t st) {
    static_assert(LT > 2 * WAVES - 1 + 1 && LT <= 3 * WAVES, "every wave owns two or three q tiles");
    int nchunk = (OCC * 256 + a->heads - 1) / a->heads;
    if (nchunk > a->Bw) nchunk = a->Bw;
    dim3 grid(nchunk, a->heads), block(64 * WAVES);
    hipLaunchKernelGGL((attn_fwd3r2_kernel<LT, LFIX, WAVES, OCC>), grid, block, 0, st, (const uint16_t*)a->qkvh, a->logit_scale,
                       (uint16_t*)a->oh, a->lse, a->Bw, a->heads, a->L, a->nwh * a->nww, a->nww, a->nwh, a->mask_thr);
    SWV2_CHECK_LAUNCH("swv2_attn_fwd");
    return SWV2_OK;
}

