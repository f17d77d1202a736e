// VALU issue-rate microbenchmark: v_fma_f32 vs v_pk_fma_f32 vs v_add_u32, by waves per SIMD (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a[16]; f32x2 p[16]; unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.f}; u[i] = (unsigned)a[i]; }
    const float m = seed * 0.999f, c = seed * 0.001f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            if (MODE == 2) u[i] = u[i] * 3u + (unsigned)it;     // v_mad_u32_u24 / mul+add
            if (MODE == 3) a[i] = fmaxf(a[i] * m, c);           // 2 ops
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* nm, int waves_per_cu, float* out, double flops_per_inst) {
    const int iters = 4096;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_cu), 0, 0, out, iters, 1.0f);
    hipEventRecord(s);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_cu), 0, 0, out, iters, 1.0f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double insts_per_simd = (double)iters * 16 * waves_per_cu / 4.0;   // wave-instructions per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
    const double tf = (double)iters * 16 * 64 * waves_per_cu * 256 * flops_per_inst / (ms * 1e-3) / 1e12;
    printf("  %-12s %2d waves/CU: %7.3f ms  %.2f cycles(@2.4GHz)/wave-instr/SIMD  %.1f TFLOP/s\n", nm, waves_per_cu, ms, cyc, tf);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {4, 8, 9, 12, 16}) {
        run<0>("v_fma_f32", w, out, 2);
        run<1>("v_pk_fma_f32", w, out, 4);
        run<2>("int mul+add", w, out, 0);
        run<3>("mul+max", w, out, 0);
    }
    return 0;
}
