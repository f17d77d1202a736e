// conv3x3_direct.hip -- 3x3 'SAME' convolution straight from the HWIO TF variable.
//
// Used where the implicit-GEMM kernel does not apply: Cin = 3 (first extractor conv,
// modules.py:62), Cout = 2 (flow heads, modules.py:274,324, with the residual adds of
// modules.py:275-277 and :326 fused) -- and as the in-library cross-check of the MFMA
// kernel for arbitrary Cin/Cout.  One thread = one output pixel x CT consecutive
// output channels; the cout block is uniform per workgroup (blockIdx.y) so weight
// addresses are wave-uniform (scalar loads).  HBM-bound layers: the 9-tap re-reads of
// x are served by L1/L2.
#include "pwc_common.h"

struct DirectArgs {
    const float* x;
    const float* w;     // HWIO (3,3,Cin,Cout)
    const float* bias;
    float* y;
    const float* res;
    int x_cs, y_cs, res_cs;
    int H, W, Ho, Wo, Cin, Cout;
    int stride, dil, pad_t, pad_l;
    int apply_act;
    float slope;
    long M;
};

template <int CT, bool VEC4>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const DirectArgs a) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int co0 = blockIdx.y * CT;
    if (m >= a.M) return;
    // 32-bit index math throughout (the host routes here only when M and the per-image extent fit):
    // the first version decomposed a 64-bit flat index with 64-bit divisions per pixel group
    const unsigned HoWo = (unsigned)(a.Ho * a.Wo), M = (unsigned)a.M;
    const unsigned ngroups = (M + 7) / 8;               // 8 pixels per wave iteration
    const unsigned wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned g = wave_id; g < ngroups; g += nwaves) {
        const unsigned m = g * 8 + (lane >> 3);
        const bool mok = m < M;
        const unsigned mm = mok ? m : 0;
        const unsigned n = mm / HoWo;
        const unsigned rem = mm - n * HoWo;
        const int oy = (int)(rem / (unsigned)a.Wo), ox = (int)(rem - (unsigned)oy * (unsigned)a.Wo);
        const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs + q * 4;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int iy = oy * a.stride - a.pad_t + ty * a.dil;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int ix = ox * a.stride - a.pad_l + tx * a.dil;
                const bool ok = mok && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
                f32x4 xv = {0.f, 0.f, 0.f, 0.f};
                if (ok) xv = *reinterpret_cast<const f32x4*>(xn + (iy * a.W + ix) * a.x_cs);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0 = fmaf(xv[e], w0[ty * 3 + tx][e], s0);
                    s1 = fmaf(xv[e], w1[ty * 3 + tx][e], s1);
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            s0 += __shfl_xor(s0, d);
            s1 += __shfl_xor(s1, d);
        }
        if (q == 0 && mok) {
            float v0 = s0 + b0, v1 = s1 + b1;
            if (a.apply_act) { v0 = pwc_lrelu(v0, a.slope); v1 = pwc_lrelu(v1, a.slope); }
            if (a.res) { v0 += a.res[(size_t)m * a.res_cs]; v1 += a.res[(size_t)m * a.res_cs + 1]; }
            a.y[(size_t)m * a.y_cs] = v0;
            a.y[(size_t)m * a.y_cs + 1] = v1;
        }
    }
}

extern "C" int pwc_conv3x3_direct_f32(const float* x, int x_cs, const float* w_hwio, const float* bias,
                                      float* y, int y_cs, const float* residual, int res_cs, int N, int H,
                                      int W, int Cin, int Cout, int stride, int dilation, int apply_act,
                                      float slope, pwc_stream_t stream) {
    if (!x || !w_hwio || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PWC_EINVAL;
    if (stride < 1 || stride > 2 || dilation < 1) return PWC_EINVAL;
    if (x_cs < Cin || y_cs < Cout || (residual && res_cs < Cout)) return PWC_EINVAL;
    DirectArgs a;
    a.x = x; a.w = w_hwio; a.bias = bias; a.y = y; a.res = residual;
    a.x_cs = x_cs; a.y_cs = y_cs; a.res_cs = res_cs;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    pwc_same_pad(H, stride, dilation, &a.Ho, &a.pad_t);
    pwc_same_pad(W, stride, dilation, &a.Wo, &a.pad_l);
    a.stride = stride; a.dil = dilation; a.apply_act = apply_act; a.slope = slope;
    a.M = (long)N * a.Ho * a.Wo;
    const bool vec4 = (Cin % 4 == 0) && (x_cs % 4 == 0) && pwc_aligned16(x);
    const unsigned gx = (unsigned)((a.M + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 3 && Cout == 16 && !residual && (y_cs & 3) == 0 && pwc_aligned16(y) && pwc_aligned16(bias)) {
        long blocks = (a.M * (Cout >> 2) + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL((conv3x3_smallcin_kernel<3, 16>), dim3((unsigned)blocks), dim3(256),
                           (size_t)27 * Cout * sizeof(float), s, a);
        return pwc_launch_status();
    }
    if (Cout == 2 && Cin == 32 && vec4 && pwc_aligned16(w_hwio) && a.M < (1L << 31) && (long)H * W * x_cs < (1L << 31)) {
        long blocks = (a.M + 31) / 32;            // 32 pixels per 256-thread block iteration
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(conv3x3_head2_c32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return pwc_launch_status();
    }
#define PWC_DIRECT(CT)                                                                              \
    do {                                                                                            \
        dim3 grid(gx, (unsigned)((Cout + CT - 1) / CT));                                            \
        if (vec4) hipLaunchKernelGGL((conv3x3_direct_kernel<CT, true>), grid, dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((conv3x3_direct_kernel<CT, false>), grid, dim3(256), 0, s, a);      \
    } while (0)
    if (Cout <= 2) PWC_DIRECT(2);
    else if (Cout <= 4) PWC_DIRECT(4);
    else if (Cout <= 8) PWC_DIRECT(8);
    else PWC_DIRECT(16);
#undef PWC_DIRECT
    return pwc_launch_status();
}
