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
    const long HoWo = (long)a.Ho * a.Wo;
    const int n = (int)(m / HoWo);
    const int rem = (int)(m - (long)n * HoWo);
    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = (co0 + c < a.Cout) ? a.bias[co0 + c] : 0.f;
    const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
    for (int ty = 0; ty < 3; ++ty) {
        const int iy = oy * a.stride - a.pad_t + ty * a.dil;
        if ((unsigned)iy >= (unsigned)a.H) continue;
        for (int tx = 0; tx < 3; ++tx) {
            const int ix = ox * a.stride - a.pad_l + tx * a.dil;
            if ((unsigned)ix >= (unsigned)a.W) continue;
            const float* xp = xn + ((size_t)iy * a.W + ix) * a.x_cs;
            const float* wt = a.w + (size_t)(ty * 3 + tx) * a.Cin * a.Cout + co0;
            if (VEC4) {
                for (int ci = 0; ci < a.Cin; ci += 4) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + ci);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* wr = wt + (size_t)(ci + e) * a.Cout;
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            if (co0 + c < a.Cout) acc[c] = fmaf(xv[e], wr[c], acc[c]);
                    }
                }
            } else {
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float xv = xp[ci];
                    const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        if (co0 + c < a.Cout) acc[c] = fmaf(xv, wr[c], acc[c]);
                }
            }
        }
    }
    float* yo = a.y + (size_t)m * a.y_cs + co0;
    const float* ro = a.res ? a.res + (size_t)m * a.res_cs + co0 : nullptr;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (co0 + c >= a.Cout) break;
        float v = acc[c];
        if (a.apply_act) v = pwc_lrelu(v, a.slope);
        if (ro) v += ro[c];
        yo[c] = v;
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
