// Does VALU work of one wave run under the fp32 MFMAs of the other wave of the same SIMD?  (gfx950)
// 512-thread workgroups (two waves per SIMD), one per CU.  Waves 0-3 issue NM v_mfma_f32_16x16x4_f32 per iteration (two
// independent accumulator chains), waves 4-7 NV instructions of one VALU kind; each side alone, then both.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_mfma_valu.hip -o scripts/exp_mfma_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int MODE, int BF = 0>   // MODE 1: MFMA half only, 2: VALU half only, 3: both; BF 1: bf16 32x32x16 MFMAs instead
__global__ __launch_bounds__(512, 2) void k(float* out, const float* src, int iters) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
    const int wave = threadIdx.x >> 6;
    const unsigned ldsaddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 65536, 0x00020000);
    const unsigned voff = (threadIdx.x & 63) * 16 + (blockIdx.x & 3) * 1024;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = threadIdx.x * 1e-3f, y = 1.0f;
        if (BF == 2) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 xa, xb;
            for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)(x + e); xb[e] = (__bf16)1.0f; }
            f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {     // 64 MFMAs of 16 cycles per iteration
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
        if (BF) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            bf16x8 xa, xb;
            for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)(x + e); xb[e] = (__bf16)1.0f; }
            f32x16 c0 = {}, c1 = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, c1, 0, 0, 0);
                }
            }
            c0 += c1;
            if (c0[0] == 12345.f) out[threadIdx.x] = c0[0] + c0[5];
            return;
        }
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
        f32x2 q0 = {1.5f, 2.f}, q1 = {3.5f, 4.f}, q2 = {5.5f, 6.f}, q3 = {7.5f, 8.f};
        float t0 = 1, t1 = 2, t2 = 3, t3 = 4;
        float s0 = threadIdx.x, s1 = 2, s2 = 3, s3 = 4;
        int i0 = threadIdx.x, i1 = 5, i2 = 7, i3 = 9;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {      // 128 instructions per iteration
                if (KIND == 0) {
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
                } else if (KIND == 1) {
                    asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(c[0]));
                } else if (KIND == 2) {
                    asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
                } else if (KIND == 3) {
                    asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(c[0]));
                } else if (KIND == 4) {
                    asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                } else if (KIND == 5) {
                    asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i3));
                } else if (KIND == 6) {
                    asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(c[0]));
                } else if (KIND == 8) {     // 8 independent chains
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c));
                } else if (KIND == 9) {     // 8 independent chains, unpacked
                    asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(c[0]));
                    asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(c[0]));
                } else if (KIND == 10) {    // 4 ds_read_b128 (conflict-free, 1 KB each)
                    f32x4 r0, r1, r2, r3;
                    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(ldsaddr) : "memory");
                    s0 += r0[0] + r1[1] + r2[2] + r3[3];
                } else if (KIND == 11) {    // 4 LDS-DMA pieces of 1 KB from an L2-hot buffer
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + (threadIdx.x >> 6) * 1024), 16, (int)voff, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + (threadIdx.x >> 6) * 1024 + 256), 16, (int)voff, 4096, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + (threadIdx.x >> 6) * 1024 + 512), 16, (int)voff, 8192, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + (threadIdx.x >> 6) * 1024 + 768), 16, (int)voff, 12288, 0, 0);
                    if ((j & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (KIND == 12) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0"
                                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                } else if (KIND == 13) {
                    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]\n v_pk_mov_b32 %1, %2, %2 op_sel:[0,1]\n v_pk_mov_b32 %2, %3, %3 op_sel:[0,1]\n v_pk_mov_b32 %3, %0, %0 op_sel:[0,1]"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                } else if (KIND == 14) {
                    asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %1, 0xffff0000, %2\n v_lshlrev_b32 %2, 16, %3\n v_and_b32 %3, 0xffff0000, %0"
                                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
                } else if (KIND == 7) {
                    asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
                }
            }
        }
        float r = q0[0] + q1[1] + q2[0] + q3[1] + t0 + t1 + t2 + t3 + p0[0] + p1[1] + p2[0] + p3[1] + s0 + s1 + s2 + s3 + (float)(i0 + i1 + i2 + i3);
        if (r == 12345.f) out[threadIdx.x] = r;
    }
}

template <int KIND, int MODE, int BF = 0> float run(float* out, const float* src, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, MODE, BF>), dim3(256), dim3(512), 0, 0, out, src, iters);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, MODE, BF>), dim3(256), dim3(512), 0, 0, out, src, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best * 1e3f;
}
template <int KIND> void kind(const char* name, float* out, const float* src, int iters) {
    const float m = run<KIND, 1>(out, src, iters), v = run<KIND, 2>(out, src, iters), b = run<KIND, 3>(out, src, iters);
    // per iteration: 32 MFMAs (x 32 cycles) per MFMA wave, 128 VALU per VALU wave
    const float per = (KIND == 8 || KIND == 9) ? 256.f : 128.f;
    printf("%-14s MFMA alone %7.1f us (%.1f cyc/MFMA @2.4GHz) | VALU alone %7.1f us (%.2f cyc/instr) | both %7.1f us | sum %7.1f max %7.1f\n",
           name, m, m * 2400.f / (iters * 32.f), v, v * 2400.f / (iters * per), b, m + v, m > v ? m : v);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    float* src; hipMalloc(&src, 65536); hipMemset(src, 0, 65536);
    const int iters = 2000;
    kind<0>("v_pk_fma_f32", out, src, iters);
    kind<1>("v_fma_f32", out, src, iters);
    kind<2>("v_pk_add_f32", out, src, iters);
    kind<3>("v_add_f32", out, src, iters);
    kind<4>("v_mov_b32", out, src, iters);
    kind<5>("v_add_u32", out, src, iters);
    kind<6>("v_max_f32", out, src, iters);
    kind<7>("v_pk_mul_f32", out, src, iters);
    kind<8>("pk_fma x8 ilp", out, src, iters);
    kind<9>("v_fma x8 ilp", out, src, iters);
    kind<10>("ds_read_b128", out, src, iters);
    {
        const float m = run<1, 1, 1>(out, src, iters), v = run<1, 2, 1>(out, src, iters), b = run<1, 3, 1>(out, src, iters);
        printf("bf16 32x32x16 MFMA + v_fma_f32: MFMA alone %7.1f us (%.1f cyc/MFMA) | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m, m * 2400.f / (iters * 32.f), v, b, m + v);
        const float m2 = run<10, 1, 1>(out, src, iters), v2 = run<10, 2, 1>(out, src, iters), b2 = run<10, 3, 1>(out, src, iters);
        printf("bf16 32x32x16 MFMA + ds_read_b128: MFMA alone %7.1f us | LDS alone %7.1f | both %7.1f | sum %7.1f\n", m2, v2, b2, m2 + v2);
    }
    kind<11>("lds-dma 1KB", out, src, iters);
    {   // round 4: the same pairing with v_mfma_f32_16x16x32_bf16 (64 per iteration, 16 cycles each)
        const float m = run<1, 1, 2>(out, src, iters), v = run<1, 2, 2>(out, src, iters), b = run<1, 3, 2>(out, src, iters);
        printf("bf16 16x16x32 MFMA + v_fma_f32: MFMA alone %7.1f us (%.1f cyc/MFMA) | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m, m * 2400.f / (iters * 64.f), v, b, m + v);
        const float m3 = run<4, 1, 2>(out, src, iters), v3 = run<4, 2, 2>(out, src, iters), b3 = run<4, 3, 2>(out, src, iters);
        printf("bf16 16x16x32 MFMA + v_mov_b32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m3, v3, b3, m3 + v3);
        const float m4 = run<4, 1, 1>(out, src, iters), v4 = run<4, 2, 1>(out, src, iters), b4 = run<4, 3, 1>(out, src, iters);
        printf("bf16 32x32x16 MFMA + v_mov_b32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m4, v4, b4, m4 + v4);
        const float m5 = run<0, 1, 1>(out, src, iters), v5 = run<0, 2, 1>(out, src, iters), b5 = run<0, 3, 1>(out, src, iters);
        printf("bf16 32x32x16 MFMA + v_pk_fma_f32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m5, v5, b5, m5 + v5);
        {
            const float a = run<2, 1, 2>(out, src, iters), b = run<2, 2, 2>(out, src, iters), c = run<2, 3, 2>(out, src, iters);
            printf("bf16 16x16x32 MFMA + v_pk_add_f32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", a, b, c, a + b);
        }
        {
            const float a = run<12, 1, 2>(out, src, iters), b = run<12, 2, 2>(out, src, iters), c = run<12, 3, 2>(out, src, iters);
            printf("bf16 16x16x32 MFMA + v_cvt_pk_bf16_f32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", a, b, c, a + b);
        }
        {
            const float a = run<13, 1, 2>(out, src, iters), b = run<13, 2, 2>(out, src, iters), c = run<13, 3, 2>(out, src, iters);
            printf("bf16 16x16x32 MFMA + v_pk_mov_b32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", a, b, c, a + b);
        }
        {
            const float a = run<14, 1, 2>(out, src, iters), b = run<14, 2, 2>(out, src, iters), c = run<14, 3, 2>(out, src, iters);
            printf("bf16 16x16x32 MFMA + v_lshlrev_b32 / v_and_b32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", a, b, c, a + b);
        }
        {
            const float a = run<3, 1, 2>(out, src, iters), b = run<3, 2, 2>(out, src, iters), c = run<3, 3, 2>(out, src, iters);
            printf("bf16 16x16x32 MFMA + v_add_f32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", a, b, c, a + b);
        }
        const float m6 = run<0, 1, 2>(out, src, iters), v6 = run<0, 2, 2>(out, src, iters), b6 = run<0, 3, 2>(out, src, iters);
        printf("bf16 16x16x32 MFMA + v_pk_fma_f32: MFMA alone %7.1f us | VALU alone %7.1f | both %7.1f | sum %7.1f\n", m6, v6, b6, m6 + v6);
    }
    return 0;
}
