// pwc_common.h -- shared device/host helpers for libpwc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pwc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pwc_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pwc_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pwc_f16x2 __attribute__((ext_vector_type(2)));

#define PWC_WAVE 64

static inline int pwc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PWC_OK : (int)e;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a launch site keeps one flag per device
// (a process that uses the library on a second GPU must set it there too).  Idempotent, benign if raced.
struct PwcDevOnce {
    unsigned long long done[4] = {0, 0, 0, 0};
};
static inline bool pwc_first_on_device(PwcDevOnce* o) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return true;    // unknown: just set it again
    const unsigned long long bit = 1ull << (d & 63);
    if (o->done[d >> 6] & bit) return false;
    o->done[d >> 6] |= bit;
    return true;
}

static inline bool pwc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// TF 'SAME' padding of one axis for a 3-tap kernel (see include/pwc_hip.h, conv).
static inline void pwc_same_pad(int in, int stride, int dil, int* out, int* before) {
    int o = (in + stride - 1) / stride;
    int total = (o - 1) * stride + 2 * dil + 1 - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

// a * b rounded to fp32 and NOT contractible into a following add/sub (hipcc fuses `a * b - c` into
// one fma otherwise -- __fmul_rn does not stop that): the reference computes such products as ops of
// their own (resize coordinate in = i * scale, flow * scale), floor / fraction come from the ROUNDED value
__device__ __forceinline__ float pwc_mul_rounded(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a fence over global memory as well, which
// hipcc lowers to `s_waitcnt vmcnt(0)` in front of the s_barrier: every LDS-DMA piece (buffer_load ... lds) and every
// store in flight is drained at each barrier, and a software pipeline that waits for its pieces with counted
// `s_waitcnt vmcnt(n)` is silently serialised (found in round 3 in the ISA of conv3x3_wino_kernel: all three barriers
// of a stage were preceded by vmcnt(0)).  A wave must have waited for its OWN pieces of a region (s_waitcnt vmcnt)
// before this barrier publishes the region to the other waves.
// (Second finding, late in round 3: a workgroup-scope fence on the "local" address space is lowered to vmcnt(0) too -- the
// backend counts buffer_load ... lds as LDS writes -- so this barrier drains the LDS-DMA pipeline just the same and the
// counted waits in front of it only matter for what they guarantee, not for overlap.  PWC_FENCE_BARRIER=0 spells the barrier
// out instead (`s_waitcnt lgkmcnt(0); s_barrier` with a "memory" clobber): the counted waits then really leave fetches in
// flight across barriers -- checked in the ISA, results identical on every harness shape -- and the fetch/LDS skeleton of the
// F(4x4) kernel drops from 154 to 134 us, but the complete kernels do not move (F(4x4) 252 against 247 us, F(2x2) 296
// against 290 us): fetch time adds to MFMA time on gfx950 whether or not it is in flight early.  The fence
// form stays the default.)
#ifndef PWC_FENCE_BARRIER
#define PWC_FENCE_BARRIER 1
#endif
__device__ __forceinline__ void pwc_lds_barrier() {
#if PWC_FENCE_BARRIER
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ float pwc_lrelu(float v, float slope) {
    // tf.nn.leaky_relu(x, alpha) = max(alpha*x, x)
    return fmaxf(v, slope * v);
}

// Bijective XCD-aware remap of a linear workgroup id (MI355X: 8 XCDs, block b is
// dispatched to XCD b % 8).  After the remap, the blocks that land on one XCD own a
// contiguous range of logical tile ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int pwc_xcd_remap(int b, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = b & 7, k = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// ---- the two-term fp16 operand split of the F16-matrix-pipe kernels (conv3x3_h2.hip, conv3x3_c16pair.hip,
// cost_volume_mfma.hip):  x = h + 2^-11 m',  h = fp16(x),  m' = fp16((x - h) 2^11)  (x - h is exact in fp32; the scaling keeps
// m' out of fp16's subnormals).  Two values in four vector instructions: v_cvt_pk_f16_f32, v_pk_mul_f32 (x 2^11 of both) and
// one mixed-precision FMA per value that reads the fp16 half in place and writes its half of the pair -- the same values as
//   h = (_Float16)x;  m' = (_Float16)fmaf((float)h, -2048.f, x * 2048.f);
// which the compiler turns into seven.  |x| >= 65504 gives h = inf and a NaN m': the range condition of these kernels.
__device__ __forceinline__ void pwc_split2(const float x0, const float x1, unsigned& h_pair, unsigned& m_pair) {
    const f32x2 xs = {x0, x1};
    const pwc_f16x2 h2 = __builtin_convertvector(xs, pwc_f16x2);
    const f32x2 xm = xs * 2048.f;
    const unsigned hp = __builtin_bit_cast(unsigned, h2);
    const float neg = -2048.f;
    unsigned mp;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(mp) : "v"(hp), "s"(neg), "v"(xm[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(mp) : "v"(hp), "s"(neg), "v"(xm[1]));
    h_pair = hp; m_pair = mp;
}
__device__ __forceinline__ void pwc_split4(const f32x4 x, pwc_f16x4& h, pwc_f16x4& m) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hp, mp;
    unsigned a, b;
    pwc_split2(x[0], x[1], a, b); hp[0] = a; mp[0] = b;
    pwc_split2(x[2], x[3], a, b); hp[1] = a; mp[1] = b;
    h = __builtin_bit_cast(pwc_f16x4, hp);
    m = __builtin_bit_cast(pwc_f16x4, mp);
}
