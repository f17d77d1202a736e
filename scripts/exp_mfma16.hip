#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int K> __global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    s16x4 a = {(short)threadIdx.x, 1, 2, 3}, b = {1, 2, 3, 4};
    bf16x8 a8, b8; for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(float)(threadIdx.x + e); b8[e] = (__bf16)1.f; }
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (K == 16) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c3, 0, 0, 0);
            }
        }
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 12345.f) out[threadIdx.x] = c0[0];
}
template <int K> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K>), dim3(256), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<K>), dim3(256), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() { float* out; hipMalloc(&out, 4096); const int it = 2000;
  float t16 = run<16>(out, it), t32 = run<32>(out, it);
  printf("v_mfma_f32_16x16x16_bf16 (legacy _1k): %.1f us = %.1f cycles/MFMA @2.4GHz | v_mfma_f32_16x16x32_bf16: %.1f us = %.1f cycles/MFMA (one wave per SIMD, 4 accumulators)\n", t16, t16*2400.f/(it*64.f), t32, t32*2400.f/(it*64.f));
  return 0; }
