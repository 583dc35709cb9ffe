// Stand-alone check of the lane-swap wave reduction used by the loss epilogue (gemm.hip, Epi<E_UNPATCH_LOSS>::wave_sum8_rows) and by
// common.h's xor16_allsum / xor32_allsum -- and the evidence for the toolchain bug it works around: built with -DUSE_BUILTIN the swaps go
// through __builtin_amdgcn_permlane32_swap / _permlane16_swap, whose second result element this ROCm 7.2 clang compiles as the first
// (`v_permlane32_swap v10, v9; v_add_f32 v9, v10, v10`): the rows then print 64480 / 192480 / ... instead of the expected sums.
// build (build container):  hipcc --offload-arch=gfx950 -O3 [-DUSE_BUILTIN] -o /tmp/swapcheck tools/experiments/permlane_swap_check.hip
// run (GPU box): it prints the four row values of x0 / x1 next to the expected sums (rows hold values (0, 2, 1, 3) and 4 + (0, 2, 1, 3)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
__device__ __forceinline__ void wave_sum8_rows(const float (&v)[8], float& x0, float& x1) {
    auto fold32 = [](float a, float b) {
#ifdef USE_BUILTIN
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
#else
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        return a + b;
#endif
    };
    auto fold16 = [](float a, float b) {
#ifdef USE_BUILTIN
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
#else
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        return a + b;
#endif
    };
    auto rowsum = [](float x) {
        auto dpp = [](float y, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        x += dpp(x, std::integral_constant<int, 0xb1>{});
        x += dpp(x, std::integral_constant<int, 0x4e>{});
        x += dpp(x, std::integral_constant<int, 0x141>{});
        x += dpp(x, std::integral_constant<int, 0x140>{});
        return x;
    };
    x0 = rowsum(fold16(fold32(v[0], v[1]), fold32(v[2], v[3])));
    x1 = rowsum(fold16(fold32(v[4], v[5]), fold32(v[6], v[7])));
}
__global__ void k(float* o) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)((i + 1) * 1000 + threadIdx.x);
    float x0, x1;
    wave_sum8_rows(v, x0, x1);
    o[threadIdx.x] = x0; o[64 + threadIdx.x] = x1;
    // individual steps
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), false, false);
    o[128 + threadIdx.x] = __builtin_bit_cast(float, r[0]); o[192 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
    const auto r2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), false, false);
    o[256 + threadIdx.x] = __builtin_bit_cast(float, r2[0]); o[320 + threadIdx.x] = __builtin_bit_cast(float, r2[1]);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); float h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // expected sum of value i over lanes: 64*(i+1)*1000 + 2016
    for (int q = 0; q < 4; ++q) printf("row %d: x0 %.0f x1 %.0f\n", q, h[16 * q], h[64 + 16 * q]);
    for (int i = 0; i < 8; ++i) printf("expect v%d = %.0f\n", i, 64.0 * (i + 1) * 1000 + 2016);
    printf("swap32 r0:"); for (int l = 0; l < 64; l += 8) printf(" %.0f", h[128 + l]); printf("\nswap32 r1:"); for (int l = 0; l < 64; l += 8) printf(" %.0f", h[192 + l]);
    printf("\nswap16 r0:"); for (int l = 0; l < 64; l += 8) printf(" %.0f", h[256 + l]); printf("\nswap16 r1:"); for (int l = 0; l < 64; l += 8) printf(" %.0f", h[320 + l]); printf("\n");
}
