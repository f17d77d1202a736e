// What limits conv3x3_h2.hip's inner loop: 12 v_mfma_f32_32x32x16_f16 per tap and wave, two waves per SIMD, with NR
// ds_read_b128 fragment reads of the next tap interleaved (one per matrix instruction).  Times one "tap" (ns) against NR, with
// the reads conflict-free (contiguous 512 B per lane half) and, beside it, as fp32 MFMA-free read loops (LDS alone).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_h2_micro.hip -o scripts/exp_h2_micro.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NR, int NM, int WAVES, int ZERO = 0>
__global__ __launch_bounds__(WAVES * 64) void tap_kernel(float* out, int iters, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += WAVES * 64) reinterpret_cast<float*>(sm)[i] = ZERO ? 0.f : 0.001f * ((i * 2654435761u >> 7) & 255);
    __syncthreads();
    const char* base = sm + (wave & 7) * 2176 + (lane >> 5) * 544 + (lane & 31) * 16;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f16x8 f[2][16];
    for (int k = 0; k < 16; ++k) f[0][k] = f[1][k] = *reinterpret_cast<const f16x8*>(base + k * 16);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < (NM > NR ? NM : NR); ++i) {
                if (i < NM) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[half][i % (NR ? NR : 1)], f[half][(i + 1) % (NR ? NR : 1)], acc[i & 3], 0, 0, 0);
                if (i < NR) f[half ^ 1][i] = *reinterpret_cast<const f16x8*>(base + 16384 * half + i * 2176 + (it & 1) * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (NM == 0) for (int k = 0; k < (NR ? NR : 1); ++k) s += (float)f[0][k][0] + (float)f[1][k][1];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NR, int NM, int WAVES, int ZERO = 0>
static void run(const char* name) {
    float* out; (void)hipMalloc(&out, 4096);
    unsigned long long* tk; (void)hipMalloc(&tk, 8);
    const int iters = 2000, blocks = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tap_kernel<NR, NM, WAVES, ZERO>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((tap_kernel<NR, NM, WAVES, ZERO>), dim3(blocks), dim3(WAVES * 64), 163840, 0, out, iters, tk);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    unsigned long long htk = 0; (void)hipMemcpy(&htk, tk, 8, hipMemcpyDeviceToHost);
    printf("%-48s %d waves/WG  NM=%2d NR=%2d: %7.1f ns per tap, %7.1f s_memtime ticks per tap (last launch; %.2f ticks/ns)  (%s)\n", name, WAVES, NM, NR,
           best * 1e6f / (2.f * iters), htk / (2.0 * iters), htk / (best * 1e6), hipGetErrorString(hipGetLastError()));
    (void)hipFree(out);
}

// `long` mode (round 5): the MFMA-only loop for ~3 s per operand set (zeros, then the pattern above) so that a
// 10 Hz log of the clocks and the power (scripts/gpu_clock_log.sh) brackets it; prints the sustained rate of each phase.
template <int ZERO>
static void run_long(const char* name, float seconds) {
    float* out; (void)hipMalloc(&out, 4096);
    unsigned long long* tk; (void)hipMalloc(&tk, 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tap_kernel<0, 12, 8, ZERO>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    double total_ms = 0; long launches = 0;
    while (total_ms < seconds * 1e3) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((tap_kernel<0, 12, 8, ZERO>), dim3(256), dim3(512), 163840, 0, out, iters, tk);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++launches;
    }
    const double taps = 2.0 * iters * launches;
    const double pf = taps * 24 * 32768.0 * 1024 / (total_ms * 1e-3) / 1e15;
    printf("%-40s %.2f s, %.1f ns per tap = %.3f PFLOP/s executed (24 v_mfma_f32_32x32x16_f16 per SIMD and tap, 1024 SIMDs)\n", name, total_ms * 1e-3,
           total_ms * 1e6 / taps, pf);
    fflush(stdout);
    (void)hipFree(out);
}

int main(int argc, char** argv) {
    if (argc > 1) {
        run_long<1>("MFMA only, all operands zero", 3.f);
        run_long<0>("MFMA only, operand pattern", 3.f);
        run_long<1>("MFMA only, all operands zero (again)", 2.f);
        return 0;
    }
    run<0, 12, 8>("MFMA only");
    run<0, 12, 8, 1>("MFMA only, all operands zero");
    run<8, 12, 8, 1>("MFMA + reads, all operands zero");
    run<4, 12, 8>("MFMA + reads");
    run<8, 12, 8>("MFMA + reads");
    run<12, 12, 8>("MFMA + reads");
    run<16, 12, 8>("MFMA + reads");
    run<8, 0, 8>("reads only");
    run<16, 0, 8>("reads only");
    run<0, 12, 4>("MFMA only, one wave per SIMD");
    run<8, 12, 4>("MFMA + reads, one wave per SIMD");
    run<16, 12, 4>("MFMA + reads, one wave per SIMD");
    run<8, 0, 4>("reads only, one wave per SIMD");
    return 0;
}
