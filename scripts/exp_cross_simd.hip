// do fp32 MFMA waves and a VALU / LDS-store wave on DIFFERENT SIMDs of one CU take the MAX or the SUM
// of their times?  (scripts/exp_mfma_valu.hip measured the same-SIMD case: the SUM.)
// 512-thread workgroups, one per CU (two waves per SIMD).  Roles by the SIMD the wave actually runs on (HW_ID.SIMD_ID):
//   waves on SIMD 0 issue one filler kind (v_pk_fma_f32 | ds_write_b128 | ds_read_b128), waves on SIMDs 1-3 issue
//   v_mfma_f32_16x16x4_f32; each side alone, then both.  Second block: the same with v_mfma_f32_16x16x32_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_cross_simd.hip -o scripts/exp_cross_simd.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int MODE, int BF>   // MODE 1: MFMA waves only, 2: filler waves only, 3: both
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int* simd_of_wave) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) simd_of_wave[threadIdx.x >> 6] = simd;
    const unsigned ldsaddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    if (simd != 0) {
        if (!(MODE & 1)) return;
        float x = threadIdx.x * 1e-3f, y = 1.0f;
        if (BF) {
            bf16x8 xa, xb;
            for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)(x + e); xb[e] = (__bf16)1.0f; }
            f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, c3, 0, 0, 0);
                }
            }
            c0 += c1 + c2 + c3;
            if (c0[0] == 12345.f) out[threadIdx.x] = c0[0] + c0[3];
            return;
        }
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
        a0 += a1 + a2 + a3;
        if (a0[0] == 12345.f) out[threadIdx.x] = a0[0] + a0[1];
    } else {
        if (!(MODE & 2)) return;
        f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
        const f32x2 c = {1.0001f, 0.9999f};
        float s0 = threadIdx.x;
        f32x4 w = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {      // 128 instructions per iteration
                if (KIND == 0) {
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
                } else if (KIND == 1) {
                    asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072"
                                 :: "v"(ldsaddr), "v"(w) : "memory");
                    if ((j & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    f32x4 r0, r1, r2, r3;
                    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ldsaddr) : "memory");
                    s0 += r0[0] + r1[1] + r2[2] + r3[3];
                }
            }
        }
        float r = p0[0] + p1[1] + p2[0] + p3[1] + s0 + lds[threadIdx.x];
        if (r == 12345.f) out[threadIdx.x] = r;
    }
}

template <int KIND, int MODE, int BF> float run(float* out, int iters, int* sw) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, MODE, BF>), dim3(256), dim3(512), 0, 0, out, iters, sw);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, MODE, BF>), dim3(256), dim3(512), 0, 0, out, iters, sw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best * 1e3f;
}
template <int KIND, int BF> void kind(const char* name, float* out, int iters, int* sw) {
    const float m = run<KIND, 1, BF>(out, iters, sw), v = run<KIND, 2, BF>(out, iters, sw), b = run<KIND, 3, BF>(out, iters, sw);
    printf("%-32s MFMA waves (SIMDs 1-3) alone %7.1f us | filler waves (SIMD 0) alone %7.1f us | both %7.1f us | sum %7.1f max %7.1f  -> %s\n",
           name, m, v, b, m + v, m > v ? m : v, b < 0.5f * (m + v + (m > v ? m : v)) ? "MAX (independent)" : "SUM (serialised)");
}
int main() {
    float* out; hipMalloc(&out, 4096);
    int* sw; hipMalloc(&sw, 64); hipMemset(sw, 0xff, 64);
    const int iters = 2000;
    kind<0, 0>("fp32 MFMA | v_pk_fma_f32", out, iters, sw);
    kind<1, 0>("fp32 MFMA | ds_write_b128", out, iters, sw);
    kind<2, 0>("fp32 MFMA | ds_read_b128", out, iters, sw);
    kind<0, 1>("bf16 16x16x32 | v_pk_fma_f32", out, iters, sw);
    kind<1, 1>("bf16 16x16x32 | ds_write_b128", out, iters, sw);
    kind<2, 1>("bf16 16x16x32 | ds_read_b128", out, iters, sw);
    int h[8]; hipMemcpy(h, sw, 32, hipMemcpyDeviceToHost);
    printf("SIMD of waves 0..7 of block 0 (last launch): %d %d %d %d %d %d %d %d\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
