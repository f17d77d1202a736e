// pwc_ops.hip -- the HBM-bound pointwise/gather ops of the PWC-Net forward for gfx950:
//   backward warping (bilinear / nearest), TF-legacy bilinear resize, channel-slice copy.
// Every kernel maps consecutive lanes to consecutive 16-byte channel quads of one NHWC
// pixel, so each wave instruction touches whole contiguous pixel records.
#include "pwc_common.h"

// ------------------------------------------------------------------ warp (a2 / a3)
// WarpingLayer.__call__ -> bilinear_warp / nearest_warp, reference modules.py:83-154.
struct WarpArgs {
    const float* x;
    const float* flow;
    float* out;
    int x_cs, flow_cs, out_cs;
    int H, W, C4;     // C4 = C / 4
    float flow_scale;
    int rows;         // N * H
    // optional second job of the same launch: copy CC4 channel quads of every pixel of cp_src into
    // cp_dst (the features_0 part of the estimator input's tf.concat, modules.py:264)
    const float* cp_src;
    float* cp_dst;
    int cp_src_cs, cp_dst_cs, CC4;
};

// grid.y walks the rows (n, y), grid.x the (x, channel quad) elements of a row -- first the warp's,
// then the copy job's; 32-bit index math (the first version decomposed a flat 64-bit index with
// four 64-bit divisions per element).
template <bool BILINEAR>
__global__ __launch_bounds__(256) void warp_kernel(const WarpArgs a) {
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    const unsigned nw = (unsigned)a.W * (unsigned)a.C4;
    if (e >= nw + (unsigned)a.W * (unsigned)a.CC4) return;
    const bool is_copy = e >= nw;
    const unsigned ee = is_copy ? e - nw : e, cv = is_copy ? (unsigned)a.CC4 : (unsigned)a.C4;
    const int gx = (int)(ee / cv), cq = (int)(ee - (unsigned)gx * cv);
    for (int row = blockIdx.y; row < a.rows; row += gridDim.y) {
        const size_t pix = (size_t)row * a.W + gx;
        if (is_copy) {
            *reinterpret_cast<f32x4*>(a.cp_dst + pix * a.cp_dst_cs + cq * 4) =
                *reinterpret_cast<const f32x4*>(a.cp_src + pix * a.cp_src_cs + cq * 4);
            continue;
        }
        const int n = row / a.H, gy = row - n * a.H;
        const float* fp = a.flow + pix * a.flow_cs;
        // the product is ROUNDED before floor / weights are taken from it (the reference multiplies in
        // its own op, model.py:109); a contracted fma(flow, scale, -floor) is not the same number
        const float fx = pwc_mul_rounded(fp[0], a.flow_scale), fy = pwc_mul_rounded(fp[1], a.flow_scale);
        const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs + cq * 4;
        f32x4 v;
        if (BILINEAR) {
            // modules.py:107-137: weights from un-clipped floors, corners clipped independently
            const float fx0 = floorf(fx), fy0 = floorf(fy);
            const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
            const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
            const int y0 = (int)fminf(fmaxf((float)gy + fy0, 0.f), hl);
            const int y1 = (int)fminf(fmaxf((float)gy + fy1, 0.f), hl);
            const int x0 = (int)fminf(fmaxf((float)gx + fx0, 0.f), wl);
            const int x1 = (int)fminf(fmaxf((float)gx + fx1, 0.f), wl);
            const float c00 = (fy1 - fy) * (fx1 - fx), c01 = (fy1 - fy) * (fx - fx0);
            const float c10 = (fy - fy0) * (fx1 - fx), c11 = (fy - fy0) * (fx - fx0);
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y0 * a.W + x0) * a.x_cs);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y0 * a.W + x1) * a.x_cs);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y1 * a.W + x0) * a.x_cs);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y1 * a.W + x1) * a.x_cs);
            v = c00 * v00 + c01 * v01 + c10 * v10 + c11 * v11;
        } else {
            // modules.py:85-92: int32 cast truncates toward zero, then clip
            int yy = gy + (int)fy, xx = gx + (int)fx;
            yy = min(max(yy, 0), a.H - 1);
            xx = min(max(xx, 0), a.W - 1);
            v = *reinterpret_cast<const f32x4*>(xn + ((size_t)yy * a.W + xx) * a.x_cs);
        }
        *reinterpret_cast<f32x4*>(a.out + pix * a.out_cs + cq * 4) = v;
    }
}

static int warp_common(bool bilinear, const float* x, int x_cs, const float* flow, int flow_cs,
                       float flow_scale, float* out, int out_cs, int N, int H, int W, int C,
                       const float* cp_src, int cp_src_cs, float* cp_dst, int cp_dst_cs, int cp_C,
                       pwc_stream_t stream) {
    if (!x || !flow || !out) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PWC_EINVAL;
    if (x_cs < C || out_cs < C || flow_cs < 2) return PWC_EINVAL;
    if ((C & 3) || (x_cs & 3) || (out_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(out)) return PWC_EALIGN;
    if (cp_C < 0 || (cp_C > 0 && (!cp_src || !cp_dst || cp_src_cs < cp_C || cp_dst_cs < cp_C))) return PWC_EINVAL;
    if (cp_C > 0 && ((cp_C & 3) || (cp_src_cs & 3) || (cp_dst_cs & 3) || !pwc_aligned16(cp_src) || !pwc_aligned16(cp_dst)))
        return PWC_EALIGN;
    if ((long)N * H >= (1L << 31) || (long)W * (C + cp_C) >= (1L << 31)) return PWC_ERANGE;
    WarpArgs a;
    a.x = x; a.flow = flow; a.out = out;
    a.x_cs = x_cs; a.flow_cs = flow_cs; a.out_cs = out_cs;
    a.H = H; a.W = W; a.C4 = C / 4; a.flow_scale = flow_scale;
    a.rows = N * H;
    a.cp_src = cp_src; a.cp_dst = cp_dst; a.cp_src_cs = cp_src_cs; a.cp_dst_cs = cp_dst_cs; a.CC4 = cp_C / 4;
    const dim3 grid((unsigned)(((long)W * (a.C4 + a.CC4) + 255) / 256), (unsigned)(a.rows < 65535 ? a.rows : 65535));
    if (bilinear)
        hipLaunchKernelGGL(warp_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(warp_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_warp_bilinear_f32(const float* x, int x_cs, const float* flow, int flow_cs,
                                     float flow_scale, float* out, int out_cs, int N, int H, int W, int C,
                                     pwc_stream_t stream) {
    return warp_common(true, x, x_cs, flow, flow_cs, flow_scale, out, out_cs, N, H, W, C, nullptr, 0, nullptr, 0, 0, stream);
}

extern "C" int pwc_warp_nearest_f32(const float* x, int x_cs, const float* flow, int flow_cs,
                                    float flow_scale, float* out, int out_cs, int N, int H, int W, int C,
                                    pwc_stream_t stream) {
    return warp_common(false, x, x_cs, flow, flow_cs, flow_scale, out, out_cs, N, H, W, C, nullptr, 0, nullptr, 0, 0, stream);
}

extern "C" int pwc_warp_copy_f32(int bilinear, const float* x, int x_cs, const float* flow, int flow_cs,
                                 float flow_scale, float* out, int out_cs, int N, int H, int W, int C,
                                 const float* copy_src, int copy_src_cs, float* copy_dst, int copy_dst_cs,
                                 int copy_C, pwc_stream_t stream) {
    return warp_common(bilinear != 0, x, x_cs, flow, flow_cs, flow_scale, out, out_cs, N, H, W, C, copy_src,
                       copy_src_cs, copy_dst, copy_dst_cs, copy_C, stream);
}

// ------------------------------------------------------------------ resize (a7)
// tf.image.resize_bilinear, TF 1.8, align_corners=False (modules.py:283-284, model.py:127):
//   src = dst * (in/out); lo = floor(src); hi = min(lo+1, in-1); t = src - lo
//   top = tl + (tr-tl)*tx; bot = bl + (br-bl)*tx; out = (top + (bot-top)*ty) * mul
struct ResizeArgs {
    const float* x;
    float* y;
    int x_cs, y_cs;
    int H, W, C, OH, OW;
    float sy, sx, mul;
    int rows;     // N * OH
    unsigned* status;   // null, or the status words (PWC_STATUS_NONFINITE)
};

// VW floats per thread (4, 2 or 1).  grid.y walks the output rows (n, oy) -- their decomposition is
// wave-uniform -- and grid.x the (ox, channel group) elements of a row with 32-bit math.  (The first
// version decomposed a flat 64-bit index with four 64-bit divisions per element and was
// division-bound: 41 us for the 29 MB two-channel x4 upsampling of the final flows.)
template <int VW, bool STATUS = false>
__global__ __launch_bounds__(256) void resize_kernel(const ResizeArgs a) {
    typedef float vec_t __attribute__((ext_vector_type(VW)));
    const unsigned CV = (unsigned)a.C / VW;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)a.OW * CV) return;
    const unsigned ox = CV == 1 ? e : e / CV, cv = CV == 1 ? 0 : e - ox * CV;
    const float fx = pwc_mul_rounded((float)ox, a.sx);   // rounded product, then floor / fraction (TF: in = i * scale)
    const int x0 = (int)floorf(fx);
    const int x1 = min(x0 + 1, a.W - 1);
    const float xl = fx - (float)x0;
    bool bad = false;
    for (int row = blockIdx.y; row < a.rows; row += gridDim.y) {
        const int n = row / a.OH, oy = row - n * a.OH;
        const float fy = pwc_mul_rounded((float)oy, a.sy);
        const int y0 = (int)floorf(fy);
        const int y1 = min(y0 + 1, a.H - 1);
        const float yl = fy - (float)y0;
        const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs + cv * VW;
        const float* r0 = xn + (size_t)y0 * a.W * a.x_cs;
        const float* r1 = xn + (size_t)y1 * a.W * a.x_cs;
        const vec_t tl = *reinterpret_cast<const vec_t*>(r0 + (size_t)x0 * a.x_cs);
        const vec_t tr = *reinterpret_cast<const vec_t*>(r0 + (size_t)x1 * a.x_cs);
        const vec_t bl = *reinterpret_cast<const vec_t*>(r1 + (size_t)x0 * a.x_cs);
        const vec_t br = *reinterpret_cast<const vec_t*>(r1 + (size_t)x1 * a.x_cs);
        const vec_t top = tl + (tr - tl) * xl;
        const vec_t bot = bl + (br - bl) * xl;
        const vec_t o = (top + (bot - top) * yl) * a.mul;
        *reinterpret_cast<vec_t*>(a.y + ((size_t)row * a.OW + ox) * a.y_cs + cv * VW) = o;
        if (STATUS) {
            // exponent bits all ones = inf or NaN.  (Not `o - o != 0`: the compiler contracts the subtraction with the multiply
            // that made o into an fma and gets the product's rounding error -- non-zero for finite values.)
#pragma unroll
            for (int k = 0; k < VW; ++k) bad = bad || (__builtin_bit_cast(unsigned, (float)o[k]) & 0x7F800000u) == 0x7F800000u;
        }
    }
    if (STATUS && bad) atomicOr(a.status, (unsigned)PWC_STATUS_NONFINITE);
}

// Round 6: the forward's LAST launch -- the x4 up-sampling of the 2-channel flows to the frame size, times 20 (reference
// model.py:125-127; 29 MB written for 1.8 MB read at batch 8) -- one thread per SOURCE cell: its four corners are requested once
// and its 4 x 4 output pixels leave as four 32-byte runs (a wave: 2 KB contiguous per output row) instead of one 8-byte store and
// four corner requests per output pixel.  in / out = 1/4 exactly: src = dst / 4 is exact, lo = dst >> 2, t = (dst & 3) / 4 -- the
// values of resize_kernel, same expressions.  Needs C == 2, OH == 4 H, OW == 4 W, y dense (y_cs == 2) and 16-byte aligned.
template <bool STATUS>
__global__ __launch_bounds__(256) void resize_x4_c2_kernel(const ResizeArgs a, int ncells) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ncells) return;
    const int xs = e % a.W;
    const int r = e / a.W;
    const int ys = r % a.H, n = r / a.H;
    const int x1 = min(xs + 1, a.W - 1), y1 = min(ys + 1, a.H - 1);
    const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
    const f32x2 tl = *reinterpret_cast<const f32x2*>(xn + ((size_t)ys * a.W + xs) * a.x_cs);
    const f32x2 tr = *reinterpret_cast<const f32x2*>(xn + ((size_t)ys * a.W + x1) * a.x_cs);
    const f32x2 bl = *reinterpret_cast<const f32x2*>(xn + ((size_t)y1 * a.W + xs) * a.x_cs);
    const f32x2 br = *reinterpret_cast<const f32x2*>(xn + ((size_t)y1 * a.W + x1) * a.x_cs);
    bool bad = false;
    float* yo = a.y + (((size_t)n * a.OH + 4 * ys) * a.OW + 4 * xs) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float yl = 0.25f * (float)j;
        f32x2 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xl = 0.25f * (float)i;
            const f32x2 top = tl + (tr - tl) * xl;
            const f32x2 bot = bl + (br - bl) * xl;
            o[i] = (top + (bot - top) * yl) * a.mul;
            if (STATUS)
                bad = bad || (__builtin_bit_cast(unsigned, (float)o[i][0]) & 0x7F800000u) == 0x7F800000u ||
                      (__builtin_bit_cast(unsigned, (float)o[i][1]) & 0x7F800000u) == 0x7F800000u;
        }
        f32x4* dst = reinterpret_cast<f32x4*>(yo + (size_t)j * a.OW * 2);
        dst[0] = f32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
        dst[1] = f32x4{o[2][0], o[2][1], o[3][0], o[3][1]};
    }
    if (STATUS && bad) atomicOr(a.status, (unsigned)PWC_STATUS_NONFINITE);
}

static int resize_run(const float* x, int x_cs, float* y, int y_cs, int N, int H, int W,
                      int C, int OH, int OW, float mul, uint32_t* status, pwc_stream_t stream) {
    if (!x || !y) return PWC_EINVAL;
    if (reinterpret_cast<uintptr_t>(status) & 7u) return PWC_EALIGN;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return PWC_EINVAL;
    if (x_cs < C || y_cs < C) return PWC_EINVAL;
    if ((long)N * OH >= (1L << 31) || (long)OW * C >= (1L << 31)) return PWC_ERANGE;
    ResizeArgs a;
    a.x = x; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW;
    a.sy = (float)H / (float)OH; a.sx = (float)W / (float)OW; a.mul = mul;
    a.rows = N * OH;
    a.status = status;
    if (C == 2 && OH == 4 * H && OW == 4 * W && y_cs == 2 && (x_cs % 2 == 0) && ((uintptr_t)x % 8 == 0) && pwc_aligned16(y) &&
        (long)N * H * W < (1L << 31)) {
        const int ncells = N * H * W;
        const dim3 g((unsigned)((ncells + 255) / 256));
        if (status) hipLaunchKernelGGL(resize_x4_c2_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, a, ncells);
        else hipLaunchKernelGGL(resize_x4_c2_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, a, ncells);
        return pwc_launch_status();
    }
    const bool vec4 = (C % 4 == 0) && (x_cs % 4 == 0) && (y_cs % 4 == 0) && pwc_aligned16(x) && pwc_aligned16(y);
    const bool vec2 = (C % 2 == 0) && (x_cs % 2 == 0) && (y_cs % 2 == 0) && ((uintptr_t)x % 8 == 0) &&
                      ((uintptr_t)y % 8 == 0);
    const int vw = vec4 ? 4 : vec2 ? 2 : 1;
    const unsigned gx = (unsigned)(((long)OW * (C / vw) + 255) / 256);
    const unsigned gy = (unsigned)(a.rows < 65535 ? a.rows : 65535);
    const dim3 grid(gx, gy);
    if (status) {
        if (vw == 4)
            hipLaunchKernelGGL((resize_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (vw == 2)
            hipLaunchKernelGGL((resize_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((resize_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        return pwc_launch_status();
    }
    if (vw == 4)
        hipLaunchKernelGGL(resize_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (vw == 2)
        hipLaunchKernelGGL(resize_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(resize_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_resize_bilinear_f32(const float* x, int x_cs, float* y, int y_cs, int N, int H, int W,
                                       int C, int OH, int OW, float mul, pwc_stream_t stream) {
    return resize_run(x, x_cs, y, y_cs, N, H, W, C, OH, OW, mul, nullptr, stream);
}

extern "C" int pwc_resize_bilinear_status_f32(const float* x, int x_cs, float* y, int y_cs, int N, int H, int W,
                                              int C, int OH, int OW, float mul, uint32_t* status, pwc_stream_t stream) {
    return resize_run(x, x_cs, y, y_cs, N, H, W, C, OH, OW, mul, status, stream);
}

// status[1] = max(status[1], bits of max |x|) -- see include/pwc_hip.h
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int x_cs, long npix, int C, unsigned* status) {
    float m = 0.f;
    const long total = npix * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long p = i / C;
        const int c = (int)(i - p * C);
        m = fmaxf(m, fabsf(x[p * x_cs + c]));             // (fmaxf drops a NaN operand)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(status + 1, __builtin_bit_cast(unsigned, m));
}

extern "C" int pwc_absmax_f32(const float* x, int x_cs, long npix, int C, uint32_t* status, pwc_stream_t stream) {
    if (!x || !status || npix <= 0 || C <= 0 || x_cs < C) return PWC_EINVAL;
    if (reinterpret_cast<uintptr_t>(status) & 7u) return PWC_EALIGN;
    long blocks = (npix * C + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_cs, npix, C, status);
    return pwc_launch_status();
}


// Two resizes of the same geometry in one launch: the 2-channel flows and the feature map that
// the estimator hands to the next pyramid level (modules.py:283-284) -- at the coarse levels each
// separate launch costs its ~6 us floor for kilobytes of data.
struct ResizePairArgs {
    const float* xa;    // 2 channels (float2 units)
    float* ya;
    const float* xb;    // CB channels, CB % 4 == 0 (float4 units)
    float* yb;
    int xa_cs, ya_cs, xb_cs, yb_cs;
    int H, W, CB, OH, OW;
    float sy, sx;
    int rows;
};

__global__ __launch_bounds__(256) void resize_pair_kernel(const ResizePairArgs a) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned upp = 1u + (unsigned)a.CB / 4;            // units per output pixel: 1 float2 + CB/4 float4
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)a.OW * upp) return;
    const unsigned ox = e / upp, un = e - ox * upp;
    const float fx = pwc_mul_rounded((float)ox, a.sx);   // rounded product, then floor / fraction (TF: in = i * scale)
    const int x0 = (int)floorf(fx);
    const int x1 = min(x0 + 1, a.W - 1);
    const float xl = fx - (float)x0;
    for (int row = blockIdx.y; row < a.rows; row += gridDim.y) {
        const int n = row / a.OH, oy = row - n * a.OH;
        const float fy = pwc_mul_rounded((float)oy, a.sy);
        const int y0 = (int)floorf(fy);
        const int y1 = min(y0 + 1, a.H - 1);
        const float yl = fy - (float)y0;
        const size_t p00 = ((size_t)n * a.H + y0) * a.W, p10 = ((size_t)n * a.H + y1) * a.W;
        const size_t po = (size_t)row * a.OW + ox;
        if (un == 0) {
            const f32x2 tl = *reinterpret_cast<const f32x2*>(a.xa + (p00 + x0) * a.xa_cs);
            const f32x2 tr = *reinterpret_cast<const f32x2*>(a.xa + (p00 + x1) * a.xa_cs);
            const f32x2 bl = *reinterpret_cast<const f32x2*>(a.xa + (p10 + x0) * a.xa_cs);
            const f32x2 br = *reinterpret_cast<const f32x2*>(a.xa + (p10 + x1) * a.xa_cs);
            const f32x2 top = tl + (tr - tl) * xl, bot = bl + (br - bl) * xl;
            *reinterpret_cast<f32x2*>(a.ya + po * a.ya_cs) = top + (bot - top) * yl;
        } else {
            const int c = (int)(un - 1) * 4;
            const f32x4 tl = *reinterpret_cast<const f32x4*>(a.xb + (p00 + x0) * a.xb_cs + c);
            const f32x4 tr = *reinterpret_cast<const f32x4*>(a.xb + (p00 + x1) * a.xb_cs + c);
            const f32x4 bl = *reinterpret_cast<const f32x4*>(a.xb + (p10 + x0) * a.xb_cs + c);
            const f32x4 br = *reinterpret_cast<const f32x4*>(a.xb + (p10 + x1) * a.xb_cs + c);
            const f32x4 top = tl + (tr - tl) * xl, bot = bl + (br - bl) * xl;
            *reinterpret_cast<f32x4*>(a.yb + po * a.yb_cs + c) = top + (bot - top) * yl;
        }
    }
}

// The same for an exact x2 up-sampling of WIDE feature maps (the dense-connection estimators hand 736 ... 2632 channels to the
// next level: 2.4 GB written at the 112 x 256 level of configs[3]): one thread per (source pixel, unit) writes the 2 x 2 output
// pixels of its cell -- every source value is requested once per neighbour instead of once per output pixel.  Same formulas,
// same values: in = out / 2 exactly, weights 0 or 1/2.
__global__ __launch_bounds__(256) void resize_pair2x_kernel(const ResizePairArgs a, long total) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned upp = 1u + (unsigned)a.CB / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pix = e / upp;
        const unsigned un = (unsigned)(e - pix * upp);
        const int x = (int)(pix % a.W);
        const long r = pix / a.W;
        const int y = (int)(r % a.H), n = (int)(r / a.H);
        const int x1 = min(x + 1, a.W - 1), y1 = min(y + 1, a.H - 1);
        const size_t p00 = ((size_t)n * a.H + y) * a.W, p10 = ((size_t)n * a.H + y1) * a.W;
        const size_t o00 = ((size_t)n * a.OH + 2 * y) * a.OW + 2 * x, o10 = o00 + a.OW;
        if (un == 0) {
            const f32x2 tl = *reinterpret_cast<const f32x2*>(a.xa + (p00 + x) * a.xa_cs);
            const f32x2 tr = *reinterpret_cast<const f32x2*>(a.xa + (p00 + x1) * a.xa_cs);
            const f32x2 bl = *reinterpret_cast<const f32x2*>(a.xa + (p10 + x) * a.xa_cs);
            const f32x2 br = *reinterpret_cast<const f32x2*>(a.xa + (p10 + x1) * a.xa_cs);
            const f32x2 top0 = tl + (tr - tl) * 0.f, bot0 = bl + (br - bl) * 0.f;
            const f32x2 top1 = tl + (tr - tl) * 0.5f, bot1 = bl + (br - bl) * 0.5f;
            *reinterpret_cast<f32x2*>(a.ya + o00 * a.ya_cs) = top0 + (bot0 - top0) * 0.f;
            *reinterpret_cast<f32x2*>(a.ya + (o00 + 1) * a.ya_cs) = top1 + (bot1 - top1) * 0.f;
            *reinterpret_cast<f32x2*>(a.ya + o10 * a.ya_cs) = top0 + (bot0 - top0) * 0.5f;
            *reinterpret_cast<f32x2*>(a.ya + (o10 + 1) * a.ya_cs) = top1 + (bot1 - top1) * 0.5f;
        } else {
            const int c = (int)(un - 1) * 4;
            const f32x4 tl = *reinterpret_cast<const f32x4*>(a.xb + (p00 + x) * a.xb_cs + c);
            const f32x4 tr = *reinterpret_cast<const f32x4*>(a.xb + (p00 + x1) * a.xb_cs + c);
            const f32x4 bl = *reinterpret_cast<const f32x4*>(a.xb + (p10 + x) * a.xb_cs + c);
            const f32x4 br = *reinterpret_cast<const f32x4*>(a.xb + (p10 + x1) * a.xb_cs + c);
            const f32x4 top0 = tl + (tr - tl) * 0.f, bot0 = bl + (br - bl) * 0.f;
            const f32x4 top1 = tl + (tr - tl) * 0.5f, bot1 = bl + (br - bl) * 0.5f;
            *reinterpret_cast<f32x4*>(a.yb + o00 * a.yb_cs + c) = top0 + (bot0 - top0) * 0.f;
            *reinterpret_cast<f32x4*>(a.yb + (o00 + 1) * a.yb_cs + c) = top1 + (bot1 - top1) * 0.f;
            *reinterpret_cast<f32x4*>(a.yb + o10 * a.yb_cs + c) = top0 + (bot0 - top0) * 0.5f;
            *reinterpret_cast<f32x4*>(a.yb + (o10 + 1) * a.yb_cs + c) = top1 + (bot1 - top1) * 0.5f;
        }
    }
}

extern "C" int pwc_resize_bilinear_pair_f32(const float* xa, int xa_cs, float* ya, int ya_cs, const float* xb,
                                            int xb_cs, float* yb, int yb_cs, int N, int H, int W, int CB, int OH,
                                            int OW, pwc_stream_t stream) {
    if (!xa || !ya || !xb || !yb) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || CB <= 0 || OH <= 0 || OW <= 0) return PWC_EINVAL;
    if (xa_cs < 2 || ya_cs < 2 || xb_cs < CB || yb_cs < CB) return PWC_EINVAL;
    if ((CB & 3) || (xb_cs & 3) || (yb_cs & 3) || !pwc_aligned16(xb) || !pwc_aligned16(yb)) return PWC_EALIGN;
    if ((xa_cs & 1) || (ya_cs & 1) || ((uintptr_t)xa & 7) || ((uintptr_t)ya & 7)) return PWC_EALIGN;
    if ((long)N * OH >= (1L << 31) || (long)OW * (1 + CB / 4) >= (1L << 31)) return PWC_ERANGE;
    ResizePairArgs a;
    a.xa = xa; a.ya = ya; a.xb = xb; a.yb = yb;
    a.xa_cs = xa_cs; a.ya_cs = ya_cs; a.xb_cs = xb_cs; a.yb_cs = yb_cs;
    a.H = H; a.W = W; a.CB = CB; a.OH = OH; a.OW = OW;
    a.sy = (float)H / (float)OH; a.sx = (float)W / (float)OW;
    a.rows = N * OH;
    // per source cell from 64 feature channels on, and from 32 on where a launch has at least 8192 source pixels (measured, batch 8,
    // flow 2 + features 32 into a 128-channel buffer: 28 x 64: 5.0 against 5.9 us, 56 x 128: 12.7 against 14.3; the two coarsest
    // levels are level or slower: profiles/r06_exp_resize_ab.txt)
    int min_cb_2x = (long)N * H * W >= 8192 ? 32 : 64;
#ifdef PWC_HARNESS
    if (const char* e = getenv("PWC_RESIZE2X_MIN_CB")) min_cb_2x = atoi(e);     // libpwc_hip_harness.so only (scripts/exp_resize_ab.py)
#endif
    if (OH == 2 * H && OW == 2 * W && CB >= min_cb_2x) {
        const long total = (long)N * H * W * (1 + CB / 4);
        long blocks = (total + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(resize_pair2x_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, total);
        return pwc_launch_status();
    }
    const dim3 grid((unsigned)(((long)OW * (1 + CB / 4) + 255) / 256), (unsigned)(a.rows < 65535 ? a.rows : 65535));
    hipLaunchKernelGGL(resize_pair_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return pwc_launch_status();
}

// ------------------------------------------------------------------ channel-slice copy
// tf.concat (modules.py:264,305) when an input arrives as its own tensor: copies C
// channels of every pixel into a channel slice of the destination.
struct CopyArgs {
    const float* src;
    float* dst;
    int src_cs, dst_cs, CV;
    long total;
};

template <bool VEC4>
__global__ __launch_bounds__(256) void copy_channels_kernel(const CopyArgs a) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total;
         idx += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % a.CV);
        const long pix = idx / a.CV;
        if (VEC4)
            *reinterpret_cast<f32x4*>(a.dst + (size_t)pix * a.dst_cs + cv * 4) =
                *reinterpret_cast<const f32x4*>(a.src + (size_t)pix * a.src_cs + cv * 4);
        else
            a.dst[(size_t)pix * a.dst_cs + cv] = a.src[(size_t)pix * a.src_cs + cv];
    }
}

extern "C" int pwc_copy_channels_f32(const float* src, int src_cs, float* dst, int dst_cs, long npix, int C,
                                     pwc_stream_t stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || src_cs < C || dst_cs < C) return PWC_EINVAL;
    const bool vec4 = (C % 4 == 0) && (src_cs % 4 == 0) && (dst_cs % 4 == 0) && pwc_aligned16(src) &&
                      pwc_aligned16(dst);
    CopyArgs a;
    a.src = src; a.dst = dst; a.src_cs = src_cs; a.dst_cs = dst_cs;
    a.CV = vec4 ? C / 4 : C;
    a.total = npix * a.CV;
    long blocks = (a.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (vec4)
        hipLaunchKernelGGL(copy_channels_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(copy_channels_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                           (hipStream_t)stream, a);
    return pwc_launch_status();
}

// ------------------------------------------------------------------ misc
extern "C" int pwc_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* pwc_error_string(int code) {
    switch (code) {
        case PWC_OK: return "ok";
        case PWC_EINVAL: return "invalid argument (null pointer, non-positive size or stride < channels)";
        case PWC_EALIGN: return "pointer or channel stride not aligned as the kernel requires";
        case PWC_ERANGE: return "tensor too large for the kernel's 32-bit indices";
        case PWC_EUNSUPPORTED: return "unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown pwc error";
    }
}

// ------------------------------------------------------------------ losses (forward)
// reference losses.py:4-13 (L1loss / L2loss / EPE) and the per-level term of multiscale_loss /
// multirobust_loss (losses.py:15-48): sum over the pixels of image n of
//     || pred[n,y,x,0:2] - gt[n, floor(y*GH/H), floor(x*GW/W), 0:2] / gt_div ||_ord ,  ord in {1, 2}
// -- tf.image.resize_nearest_neighbor (TF 1.8, align_corners=False: src = floor(dst * in/out),
// clipped) is folded into the read; GH = H, GW = W, gt_div = 1: plain norm of the difference.  The ground truth is DIVIDED
// (losses.py:20 `flows_gt/20.`), not multiplied by a reciprocal: x/20 and x*(1/20) differ by 1 ulp.
// Deterministic: fixed-shape block partial sums, then one block per image adds them in order.
struct FlowNormArgs {
    const float* pred;
    const float* gt;
    float* partial;      // [N][gridDim.x]
    int pred_cs, gt_cs;
    int H, W, GH, GW;
    float sy, sx, gt_div;
    int ord;
};

__global__ __launch_bounds__(256) void flow_norm_partial_kernel(const FlowNormArgs a) {
    __shared__ float red[256];
    const int n = blockIdx.y;
    const int npix = a.H * a.W;
    float s = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int y = p / a.W, x = p - y * a.W;
        const int gy = min((int)floorf(pwc_mul_rounded((float)y, a.sy)), a.GH - 1), gx = min((int)floorf(pwc_mul_rounded((float)x, a.sx)), a.GW - 1);
        const float* pp = a.pred + ((size_t)n * npix + p) * a.pred_cs;
        const float* gp = a.gt + (((size_t)n * a.GH + gy) * a.GW + gx) * a.gt_cs;
        const float dx = gp[0] / a.gt_div - pp[0], dy = gp[1] / a.gt_div - pp[1];
        s += a.ord == 1 ? fabsf(dx) + fabsf(dy) : sqrtf(dx * dx + dy * dy);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[(size_t)n * gridDim.x + blockIdx.x] = red[0];
}

__global__ void flow_norm_final_kernel(const float* __restrict__ partial, int nparts, int nimg, float* __restrict__ out) {
    // one thread per image: the few hundred partials are added in index order (deterministic)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nimg) return;
    float s = 0.f;
    for (int i = 0; i < nparts; ++i) s += partial[(size_t)n * nparts + i];
    out[n] = s;
}

extern "C" size_t pwc_flow_norm_workspace_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    long parts = ((long)H * W + 255) / 256;
    if (parts > 256) parts = 256;
    return (size_t)N * parts;
}

extern "C" int pwc_flow_norm_sums_f32(const float* pred, int pred_cs, const float* gt, int gt_cs, int N, int H, int W,
                                      int GH, int GW, float gt_div, int ord, float* workspace,
                                      size_t workspace_floats, float* out_sums, pwc_stream_t stream) {
    if (!pred || !gt || !workspace || !out_sums) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || GH <= 0 || GW <= 0 || pred_cs < 2 || gt_cs < 2) return PWC_EINVAL;
    if (ord != 1 && ord != 2) return PWC_EUNSUPPORTED;
    if (!(gt_div != 0.f)) return PWC_EINVAL;
    if ((long)H * W >= (1L << 31) || N > 65535) return PWC_ERANGE;
    if (workspace_floats < pwc_flow_norm_workspace_floats(N, H, W)) return PWC_EINVAL;
    FlowNormArgs a;
    a.pred = pred; a.gt = gt; a.partial = workspace; a.pred_cs = pred_cs; a.gt_cs = gt_cs;
    a.H = H; a.W = W; a.GH = GH; a.GW = GW;
    a.sy = (float)GH / (float)H; a.sx = (float)GW / (float)W; a.gt_div = gt_div; a.ord = ord;
    const int parts = (int)(pwc_flow_norm_workspace_floats(N, H, W) / N);
    hipLaunchKernelGGL(flow_norm_partial_kernel, dim3((unsigned)parts, (unsigned)N), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(flow_norm_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       (const float*)workspace, parts, N, out_sums);
    return pwc_launch_status();
}

// ---------------------------------------------------------------- stream placement probe (host side: pwcnet_amd/model.py)
// Keeps `stream` busy for about `ticks` cycles of the device's s_memtime counter without touching memory.  Used ONLY to find
// out whether two HIP streams are served by one hardware queue (a kernel on the second stream then cannot start before this
// one ends); never part of a forward.
__global__ void pwc_spin_kernel(long long ticks) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void pwc_touch_kernel(float* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}

extern "C" int pwc_device_spin(long long ticks, pwc_stream_t stream) {
    if (ticks < 0 || ticks > (1LL << 34)) return PWC_EINVAL;
    hipLaunchKernelGGL(pwc_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
    return pwc_launch_status();
}

extern "C" int pwc_device_touch(float* p, pwc_stream_t stream) {
    if (!p) return PWC_EINVAL;
    hipLaunchKernelGGL(pwc_touch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    return pwc_launch_status();
}

