#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p) {
  float a = (float)threadIdx.x, b = 1000.f + threadIdx.x;
  a = a * 1.0f + 0.f;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  p[threadIdx.x] = a; p[64 + threadIdx.x] = b;
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a: "); for (int i = 0; i < 64; i += 8) printf("%g ", h[i]); printf("\nb: "); for (int i = 0; i < 64; i += 8) printf("%g ", h[64 + i]); printf("\n");
  return 0;
}
