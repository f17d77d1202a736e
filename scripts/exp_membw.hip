// Round 5: what the memory system gives the correlation kernel's traffic pattern at level 4 (8 x 112 x 256 pixels):
// reads of 128-byte pixel records (f0, f1) and writes of 336 bytes out of every `ocs`-float pixel record, with nothing else
// in the kernel.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_membw.hip -o scripts/exp_membw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one wave = 16 pixels per iteration (a 4 x 4 block as in the kernel would be 4 row pieces; here: 16 consecutive pixels)
// MODE bit 0: read f0 + f1 records (2 x 2 KB per wave-iteration), bit 1: write the 84-float records, bit 2: nt stores,
// bit 3: writes contiguous (84 floats per pixel packed, no gaps)
template <int MODE>
__global__ __launch_bounds__(256) void bw_kernel(const float* f0, const float* f1, float* out, int npix, int ocs, float* sink) {
    const int lane = threadIdx.x & 63;
    const int wave_g = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * 4;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)f0, 0, npix * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)f1, 0, npix * 128, 0x00020000);
    const size_t obytes = (MODE & 8) ? (size_t)npix * 336 : (size_t)npix * ocs * 4;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)obytes, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nblk = npix / 16;
    // contiguous ranges of blocks per wave (a strip walk touches memory in runs)
    const int per = (nblk + nwaves - 1) / nwaves;
    const int b0 = wave_g * per, b1 = min(b0 + per, nblk);
    for (int b = b0; b < b1; ++b) {
        const int p0 = b * 16;
        if (MODE & 1) {
            f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, p0 * 128 + lane * 16, 0, 0));
            f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, p0 * 128 + 1024 + lane * 16, 0, 0));
            f32x4 c0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, p0 * 128 + lane * 16, 0, 0));
            f32x4 c1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, p0 * 128 + 1024 + lane * 16, 0, 0));
            acc += a0 + a1 + c0 + c1;
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int e = i * 64 + lane;                 // quad e of the block's 16 x 21
                const int p = e / 21, qd = e - p * 21;
                const unsigned vo = e < 336 ? ((MODE & 8) ? (unsigned)((p0 + p) * 336 + qd * 16) : (unsigned)(((p0 + p) * ocs + qd * 4) * 4)) : 0x80000000u;
                const f32x4 v = {(float)e, (float)b, 1.f, 2.f};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, (int)vo, 0, (MODE & 4) ? 2 : 0);
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <typename F>
static float time_us(F&& f, int iters) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) f(i);
    (void)hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f(i);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

int main() {
    const int npix = 8 * 112 * 256;
    const int NS = 3;
    for (int ocs : {128, 160}) {
        std::vector<float*> f0(NS), f1(NS), out(NS);
        for (int s = 0; s < NS; ++s) {
            (void)hipMalloc(&f0[s], (size_t)npix * 128); (void)hipMalloc(&f1[s], (size_t)npix * 128);
            (void)hipMalloc(&out[s], (size_t)npix * ocs * 4);
            (void)hipMemset(f0[s], 0, (size_t)npix * 128); (void)hipMemset(f1[s], 0, (size_t)npix * 128); (void)hipMemset(out[s], 0, (size_t)npix * ocs * 4);
        }
        float* sink; (void)hipMalloc(&sink, 16);
        printf("== level-4 traffic pattern, pixel record stride %d floats (%d B), 512 workgroups of 256 threads\n", ocs, ocs * 4);
        auto rep = [&](const char* nm, double mb, float us) { printf("  %-58s %7.1f us  %6.0f GB/s\n", nm, us, mb / us * 1e3); fflush(stdout); };
        const double rmb = npix * 256.0 / 1e6, wmb = npix * 336.0 / 1e6;
        for (int grid : {512, 1024, 2048}) {
            printf("  grid %d\n", grid);
            rep("read f0 + f1 records (58.7 MB)", rmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<1>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("write 336 B of every record (77.1 MB), default policy", wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<2>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("write 336 B of every record (77.1 MB), nt", wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<6>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("write 77.1 MB contiguous, default policy", wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<10>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("write 77.1 MB contiguous, nt", wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<14>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("read + write records (135.8 MB), default policy", rmb + wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<3>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("read + write records (135.8 MB), nt stores", rmb + wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<7>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
            rep("read + write contiguous (135.8 MB), nt stores", rmb + wmb, time_us([&](int i) { hipLaunchKernelGGL(bw_kernel<15>, dim3(grid), dim3(256), 0, 0, f0[i % NS], f1[i % NS], out[i % NS], npix, ocs, sink); }, 12));
        }
        for (int s = 0; s < NS; ++s) { (void)hipFree(f0[s]); (void)hipFree(f1[s]); (void)hipFree(out[s]); }
    }
    return 0;
}
