// Instruction-throughput micro-benchmarks for gfx950 (build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench).
// Each test runs ITER iterations of an unrolled body on W waves per SIMD (one workgroup of 256*W threads per CU) and
// reports shader cycles (s_memtime) per instruction per wave, i.e. the issue interval one wave observes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITER 2000

template <int TEST>
__global__ void k(float* out, uint64_t* cyc, float seed) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x16 big[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
    bf16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    bf16x8 a8 = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {3, 2, 1, (short)threadIdx.x, 1, 1, 1, 1};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (TEST == 0) {          // 16 independent v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        } else if (TEST == 1) {   // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        } else if (TEST == 2) {   // 8 mfma 16x16x16 bf16 (independent accumulators)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        } else if (TEST == 3) {   // 8 mfma 16x16x32 bf16
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        } else if (TEST == 4) {   // 2 mfma 32x32x16 bf16
#pragma unroll
            for (int i = 0; i < 2; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, big[i], 0, 0, 0);
        } else if (TEST == 5) {   // 2 mfma 32x32x8 bf16_1k
#pragma unroll
            for (int i = 0; i < 2; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, big[i], 0, 0, 0);
        } else if (TEST == 6) {   // softmax-like mix per 16x16 tile: 2 mfma16 + 4 exp + 4 fma + 4 add + 2 cvt_pk
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[2 * i + 1], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[4 * i + r]) : "v"(seed));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[4 * i + r]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(4 * i + r + 5) & 15]) : "v"(x[4 * i + r]));
                }
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[(4 * i + 9) & 15]) : "v"(x[4 * i]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[(4 * i + 10) & 15]) : "v"(x[4 * i + 1]));
            }
        } else if (TEST == 7) {   // exp-only mix: 2 mfma16 + 4 exp + 2 cvt_pk per tile (fixed-max fast path, sum by MFMA)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[2 * i + 1], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("v_exp_f32 %0, %0" : "+v"(x[4 * i + r]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[(4 * i + 9) & 15]) : "v"(x[4 * i]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[(4 * i + 10) & 15]) : "v"(x[4 * i + 1]));
            }
        } else if (TEST == 8) {   // 16 independent v_pk_fma_f32 (2 floats per lane each)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&x[2 * i]));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&x[2 * i]));
        } else if (TEST == 9) {   // 16 v_exp_f16 (one f16 per lane)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
        } else if (TEST == 10) {  // 16 v_pk_fma_f16
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(x[i]));
        } else if (TEST == 11) {  // 16 v_max3_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
        } else if (TEST == 12) {  // 16 v_cvt_pk_bf16_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 15]));
        } else if (TEST == 14) {  // one attention-forward row in registers: 11 mfma32 -> 44 exp -> 22 cvt_pk -> 6 x 2 mfma32 chains
            f32x4 a[11];
#pragma unroll
            for (int t = 0; t < 11; ++t) a[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, (f32x4){x[0], x[1], x[2], x[3]}, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 11; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = __builtin_amdgcn_exp2f(a[t][r]);
            f32x4 o = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t + 1 < 11; t += 2) {
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                unsigned pk[4];
                pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a[t][0], a[t][1]}, bf2));
                pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a[t][2], a[t][3]}, bf2));
                pk[2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a[t + 1][0], a[t + 1][1]}, bf2));
                pk[3] = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a[t + 1][2], a[t + 1][3]}, bf2));
                const bf16x8 pb = __builtin_bit_cast(bf16x8, pk);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, pb, o, 0, 0, 0);
                rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8, pb, rs, 0, 0, 0);
            }
            const float inv = 1.f / (rs[0] + a[10][0]);
            x[0] = o[0] * inv; x[1] = o[1] * inv; x[2] = o[2] * inv; x[3] = o[3] * inv + a[10][3];
        } else if (TEST == 13) {  // exp alternated with fma (8 + 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[2 * i]));
                asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[2 * i + 1]));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int TEST>
void run(const char* name, int per_iter) {
    for (int W = 1; W <= 4; ++W) {
        const int blocks = 256, threads = 256 * W, waves = blocks * threads / 64;
        float* out; uint64_t* cyc;
        hipMalloc(&out, sizeof(float) * blocks * threads);
        hipMalloc(&cyc, sizeof(uint64_t) * waves);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 0.5f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 0.5f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(waves);
        hipMemcpy(h.data(), cyc, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[waves / 2], mx = (double)h[waves - 1], n = (double)ITER * per_iter;
        printf("%-44s W=%d  cyc/inst/wave med %.2f max %.2f  -> per SIMD (max/W) %.2f   (wall %.3f ms, clock ~%.2f GHz)\n", name, W,
               med / n, mx / n, mx / n / W, ms, mx / (ms * 1e6));
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    run<0>("v_exp_f32 x16", 16);
    run<1>("v_fma_f32 x16", 16);
    run<8>("v_pk_fma_f32 x16", 16);
    run<9>("v_exp_f16 x16", 16);
    run<10>("v_pk_fma_f16 x16", 16);
    run<11>("v_max3_f32 x16", 16);
    run<12>("v_cvt_pk_bf16_f32 x16", 16);
    run<13>("exp+fma alternating x16", 16);
    run<2>("mfma_16x16x16_bf16_1k x8", 8);
    run<3>("mfma_16x16x32_bf16 x8", 8);
    run<4>("mfma_32x32x16_bf16 x2", 2);
    run<5>("mfma_32x32x8_bf16_1k x2", 2);
    run<6>("tile mix (2 mfma16+4fma+4exp+4add+2cvt) x4", 4);
    run<7>("tile mix fast (2 mfma16+4exp+2cvt) x4", 4);
    run<14>("fwd row (11mfma32,44exp,22cvt,12mfma32) x1", 1);
    return 0;
}
