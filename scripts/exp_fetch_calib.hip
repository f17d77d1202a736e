// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this repo quotes them for
// (VERDICT r3 item 3).  Every kernel moves a KNOWN byte count of a 1 GiB buffer (4x the 256 MiB Infinity Cache) exactly once:
//   read_lines_dma    buffer_load_dwordx4 ... lds, a wave fetches 1 KB contiguous (whole 128-byte lines)          [weights]
//   read_half_dma     buffer_load_dwordx4 ... lds, 4 lanes fetch 64 contiguous bytes of a 512-byte record; the 8 chunks
//                     of a record are fetched in 8 passes over the workgroup's tile (the F(4x4) patch fetch: 16 of 128
//                     channels of a pixel per stage)                                                              [patch]
//   read_lines_vgpr   global_load_dwordx4, coalesced                                                              [baseline]
//   read_gather_vgpr  global_load_dwordx4 of 128-byte records at pseudo-random record indices (the correlation's bilinear
//                     gather: whole lines per instruction, scattered)                                             [gather]
//   write_lines       global_store_dwordx4, coalesced, default policy;  write_lines_nt: the same with nt
//   write_records336  84-float (336-byte) records written as dwordx4 pieces (the cost-volume output)
// Run each under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes);
// scripts/fetch_calib_table.py divides the counter by the true byte count.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_fetch_calib.hip -o scripts/exp_fetch_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr size_t BYTES = 1ull << 30;

__global__ __launch_bounds__(256) void read_lines_dma(const float* src, float* out) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    __shared__ __attribute__((aligned(16))) float lds[4 * 256 * 4];        // 4 waves x 4 KB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // workgroup b owns 64 KB: 16 pieces of 1 KB per wave
    const float* base = src + (size_t)blockIdx.x * (65536 / 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + wave * 1024 + (i & 3) * 256), 16, lane * 16, (wave * 16 + i) * 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 12345.678f) out[threadIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void read_half_dma(const float* src, float* out) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    __shared__ __attribute__((aligned(16))) float lds[4 * 256 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // workgroup b owns 128 records of 512 bytes (64 KB); pass c fetches chunk c (64 bytes) of every record: a wave
    // instruction covers 16 records x 64 bytes
    const float* base = src + (size_t)blockIdx.x * (65536 / 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rec = (wave * 2 + i) * 16 + (lane >> 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + wave * 1024 + i * 256), 16, rec * 512 + (lane & 3) * 16, c * 64, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (lds[threadIdx.x] == 12345.678f) out[threadIdx.x] = 1.f;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void read_lines_vgpr(const f32x4* src, float* out) {
    const size_t n = BYTES / 16;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[threadIdx.x] = 1.f;
}

__global__ __launch_bounds__(256) void read_gather_vgpr(const f32x4* src, float* out) {
    // 8 lanes read one 128-byte record; records visited in a bijective pseudo-random order (odd multiplier mod 2^23)
    const size_t nrec = BYTES / 128;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3; g < nrec; g += ((size_t)gridDim.x * blockDim.x) >> 3) {
        const size_t rec = (g * 2654435761ull + 12345ull) & (nrec - 1);
        s += src[rec * 8 + (threadIdx.x & 7)];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[threadIdx.x] = 1.f;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_lines(float* dst) {
    const size_t n = BYTES / 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x7FFFFFFF, 0x00020000);
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float* p = dst + i * 4;
        if (NT) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(p));
        else *reinterpret_cast<f32x4*>(p) = __builtin_bit_cast(f32x4, v);
    }
    (void)rsrc;
}

__global__ __launch_bounds__(256) void write_records336(float* dst) {
    // 84-float records, 21 dwordx4 pieces each: lane l of a group of 21 writes piece l (the cost-volume kernel's stage
    // writes 336-byte records as contiguous 16-byte pieces, nt)
    const size_t nrec = BYTES / 336;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t npiece = nrec * 21;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npiece; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + i * 4));
}

int main(int argc, char** argv) {
    float *buf, *out;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, BYTES);
    (void)hipDeviceSynchronize();
    const int reps = 2;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(read_lines_dma, dim3(BYTES / 65536), dim3(256), 0, 0, buf, out);
        hipLaunchKernelGGL(read_half_dma, dim3(BYTES / 65536), dim3(256), 0, 0, buf, out);
        hipLaunchKernelGGL(read_lines_vgpr, dim3(8192), dim3(256), 0, 0, (const f32x4*)buf, out);
        hipLaunchKernelGGL(read_gather_vgpr, dim3(8192), dim3(256), 0, 0, (const f32x4*)buf, out);
        hipLaunchKernelGGL(write_lines<false>, dim3(8192), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(write_lines<true>, dim3(8192), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(write_records336, dim3(8192), dim3(256), 0, 0, buf);
        (void)hipDeviceSynchronize();
    }
    printf("true bytes per launch: read_* and write_lines* %zu; write_records336 %zu\n", BYTES, (BYTES / 336) * 336);
    return 0;
}
