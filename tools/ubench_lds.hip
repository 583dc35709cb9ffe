// LDS read throughput per instruction kind (one workgroup of 256 threads per CU, all CUs): ds_read_b64, ds_read_b64_tr_b16,
// ds_read_b128, each as a long stream of independent reads.  hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, int pitch, int* out) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[176 * 64 + 8 * 64];
    for (int i = threadIdx.x; i < 176 * 64; i += 256) smem[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, fr = lane & 15, g = lane >> 4;
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint16_t* smem_ = smem + (it & 7) * 64;      // iteration-dependent base: nothing can be hoisted
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 11; ++u) {
            if (KIND == 0) { const bf16x4 v = *(const bf16x4*)(smem_ + (16 * u + fr) * pitch + 4 * g); acc += v[0] + v[3]; }
            if (KIND == 1) { const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(smem_ + (16 * u + 4 * g + (fr >> 2)) * pitch + (fr & 3) * 4)); acc += v[0] + v[3]; }
            if (KIND == 2) { const bf16x8 v = *(const bf16x8*)(smem_ + (16 * u + fr) * pitch + 8 * g); acc += v[0] + v[7]; }
        }
    }
    if (acc == 0x7fffffff) out[0] = acc;
}
template <int KIND> float run(int pitch) {
    int* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, 10, pitch, out);
    hipEventRecord(a); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, iters, pitch, out); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / (iters * 11.f * 4.f);      // ns per wave instruction per CU-slot (4 waves per CU share the LDS)
}
int main() {
    printf("ns per wave-instruction (4 waves per CU issuing): b64 pitch16 %.2f | tr_b64 pitch16 %.2f | b128 pitch32 %.2f | b128 pitch40 %.2f | b64 pitch40 %.2f | tr_b64 pitch40 %.2f\n",
           run<0>(16), run<1>(16), run<2>(32), run<2>(40), run<0>(40), run<1>(40));
    return 0;
}
