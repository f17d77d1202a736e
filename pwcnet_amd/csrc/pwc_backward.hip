// pwc_backward.hip -- backward (gradient) kernels of the PWC-Net ops for gfx950 and the optimiser
// update: the training path of the reference (train.py:66-92 builds them with tf.gradients /
// tf.train.AdamOptimizer).  Conventions as in the forward kernels: NHWC fp32,
// (pointer, channel stride) tensors, everything enqueued on the caller's stream.
//
// Gradient buffers have the layout of the activation they belong to.  Kernels whose name ends in
// `_acc` ADD into their output (an activation with several consumers collects its gradient from all of
// them); the others overwrite it.
//
//   pwc_lrelu_grad_f32          dy *= (y > 0 ? 1 : slope)          tf.nn.leaky_relu's gradient (y = its output)
//   pwc_channel_sums_f32        s[c] = sum_p dy[p, c]               bias gradient of tf.layers.Conv2D
//   pwc_lrelu_grad_channel_sums_f32   the two above in one pass over dy
//   pwc_add_f32                 dst (+)= alpha * src                channel-slice accumulate
//   pwc_resize_bilinear_grad    transpose of the TF-legacy resize  modules.py:283-284
//   pwc_warp_bilinear_grad      d/dx and d/dflow of bilinear_warp  modules.py:99-137
//   pwc_cost_volume_grad        d/df0 and d/df1w of the cost volume modules.py:158-204
//   pwc_flow_norm_grad_f32      gradient of L1loss / L2loss terms  losses.py:4-8,20-29
//   pwc_adam_step_f32           tf.train.AdamOptimizer update + the weights' L2 term  train.py:75,90
#include "pwc_common.h"
#include <cstdint>

// ------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void lrelu_grad_kernel(const float* __restrict__ y, int y_cs, float* __restrict__ dy,
                                                         int dy_cs, long npix, int C4, float slope) {
    const long total = npix * C4;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e / C4;
        const int c = (int)(e - p * C4) * 4;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + p * y_cs + c);
        f32x4 g = *reinterpret_cast<f32x4*>(dy + p * dy_cs + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = yv[i] > 0.f ? g[i] : g[i] * slope;
        *reinterpret_cast<f32x4*>(dy + p * dy_cs + c) = g;
    }
}

extern "C" int pwc_lrelu_grad_f32(const float* y, int y_cs, float* dy, int dy_cs, long npix, int C, float slope,
                                  pwc_stream_t stream) {
    if (!y || !dy || npix <= 0 || C <= 0 || y_cs < C || dy_cs < C) return PWC_EINVAL;
    if ((C & 3) || (y_cs & 3) || (dy_cs & 3) || !pwc_aligned16(y) || !pwc_aligned16(dy)) return PWC_EALIGN;
    long blocks = (npix * (C / 4) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lrelu_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, y_cs, dy, dy_cs,
                       npix, C / 4, slope);
    return pwc_launch_status();
}

// dst[p, 0:C] = beta * dst + alpha * src[p, 0:C]   (beta in {0, 1})
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ src, int src_cs, float* __restrict__ dst,
                                                  int dst_cs, long npix, int C, float alpha, int accumulate) {
    const long total = npix * C;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e / C;
        const int c = (int)(e - p * C);
        const float v = alpha * src[p * src_cs + c];
        float* d = dst + p * dst_cs + c;
        *d = accumulate ? *d + v : v;
    }
}

extern "C" int pwc_add_f32(const float* src, int src_cs, float* dst, int dst_cs, long npix, int C, float alpha,
                           int accumulate, pwc_stream_t stream) {
    if (!src || !dst || npix <= 0 || C <= 0 || src_cs < C || dst_cs < C) return PWC_EINVAL;
    long blocks = (npix * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, src_cs, dst, dst_cs, npix,
                       C, alpha, accumulate);
    return pwc_launch_status();
}

// s[c] = sum over pixels of dy[p, c]: fixed-shape partial sums (deterministic), then a second kernel adds them.
// Block = 256 threads = (256 / CP) pixel lanes x CP channel lanes (CP = channels rounded up to a power of two <= 256:
// consecutive threads read consecutive channels of a pixel -- coalesced); every block walks a fixed pixel range, the
// pixel lanes are summed through LDS.  With Y != nullptr the kernel first applies the leaky-relu mask to dy IN PLACE
// (dy *= y > 0 ? 1 : slope) -- bias gradient and activation gradient in one pass over dy.
// partial: [nparts][C]
// VEC = 4: channels in float4 lanes (C, strides and pointers multiples of 4 floats), CP = lanes per pixel (power of two).
template <int VEC>
__global__ __launch_bounds__(256) void channel_sums_partial_kernel(const float* __restrict__ y, int y_cs, float slope,
                                                                   float* __restrict__ dy, int dy_cs, long npix, int C, int CP,
                                                                   float* __restrict__ partial) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ vec_t red[256];
    const int c = threadIdx.x & (CP - 1), pl = threadIdx.x / CP, npl = 256 / CP;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
    for (int cb = 0; cb < C; cb += CP * VEC) {           // (more than 256 lanes of channels: blocks of them)
        const int cc = cb + c * VEC;
        vec_t s = 0.f;
        if (cc < C) {
            long p = p0 + pl;
            for (; p + npl < p1; p += 2 * npl) {          // two pixels in flight per lane
                vec_t g0 = *(const vec_t*)(dy + p * dy_cs + cc);
                vec_t g1 = *(const vec_t*)(dy + (p + npl) * dy_cs + cc);
                if (y) {
                    const vec_t y0 = *(const vec_t*)(y + p * y_cs + cc), y1 = *(const vec_t*)(y + (p + npl) * y_cs + cc);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        g0[e] = y0[e] > 0.f ? g0[e] : g0[e] * slope;
                        g1[e] = y1[e] > 0.f ? g1[e] : g1[e] * slope;
                    }
                    *(vec_t*)(dy + p * dy_cs + cc) = g0;
                    *(vec_t*)(dy + (p + npl) * dy_cs + cc) = g1;
                }
                s += g0;
                s += g1;
            }
            if (p < p1) {
                vec_t g0 = *(const vec_t*)(dy + p * dy_cs + cc);
                if (y) {
                    const vec_t y0 = *(const vec_t*)(y + p * y_cs + cc);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g0[e] = y0[e] > 0.f ? g0[e] : g0[e] * slope;
                    *(vec_t*)(dy + p * dy_cs + cc) = g0;
                }
                s += g0;
            }
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int k = npl >> 1; k > 0; k >>= 1) {         // fixed-order tree over the pixel lanes
            if (pl < k) red[threadIdx.x] += red[threadIdx.x + k * CP];
            __syncthreads();
        }
        if (pl == 0 && cc < C) *(vec_t*)(partial + (long)blockIdx.x * C + cc) = red[c];
        __syncthreads();
    }
}
// out[c] (+)= sum_i partial[i][c]: block = 16 channels x 16 part lanes
__global__ __launch_bounds__(256) void channel_sums_final_kernel(const float* __restrict__ partial, int nparts, int C,
                                                                 float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C)
        for (int i = pl; i < nparts; i += 16) s += partial[(long)i * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 8; k > 0; k >>= 1) {
        if (pl < k) red[threadIdx.x] += red[threadIdx.x + k * 16];
        __syncthreads();
    }
    if (pl == 0 && c < C) out[c] = accumulate ? out[c] + red[cl] : red[cl];
}

extern "C" size_t pwc_channel_sums_workspace_floats(long npix, int C) {
    if (npix <= 0 || C <= 0) return 0;
    long parts = (npix + 255) / 256;
    if (parts > 2048) parts = 2048;
    return (size_t)parts * C;
}

static int channel_sums_launch(const float* y, int y_cs, float slope, float* dy, int dy_cs, long npix, int C, float* workspace,
                               size_t workspace_floats, float* out, int accumulate, pwc_stream_t stream) {
    const size_t need = pwc_channel_sums_workspace_floats(npix, C);
    if (workspace_floats < need) return PWC_EINVAL;
    const int parts = (int)(need / C);
    const bool vec4 = C % 4 == 0 && dy_cs % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)workspace & 15) == 0 &&
                      (!y || (y_cs % 4 == 0 && ((uintptr_t)y & 15) == 0));
    const int lanes = vec4 ? C / 4 : C;
    int cp = 1;
    while (cp < lanes && cp < 256) cp <<= 1;
    if (vec4)
        hipLaunchKernelGGL(channel_sums_partial_kernel<4>, dim3((unsigned)parts), dim3(256), 0, (hipStream_t)stream, y, y_cs,
                           slope, dy, dy_cs, npix, C, cp, workspace);
    else
        hipLaunchKernelGGL(channel_sums_partial_kernel<1>, dim3((unsigned)parts), dim3(256), 0, (hipStream_t)stream, y, y_cs,
                           slope, dy, dy_cs, npix, C, cp, workspace);
    hipLaunchKernelGGL(channel_sums_final_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, parts, C, out, accumulate);
    return pwc_launch_status();
}

extern "C" int pwc_channel_sums_f32(const float* dy, int dy_cs, long npix, int C, float* workspace, size_t workspace_floats,
                                    float* out, int accumulate, pwc_stream_t stream) {
    if (!dy || !workspace || !out || npix <= 0 || C <= 0 || dy_cs < C) return PWC_EINVAL;
    return channel_sums_launch(nullptr, 0, 0.f, const_cast<float*>(dy), dy_cs, npix, C, workspace, workspace_floats, out,
                               accumulate, stream);
}

// leaky-relu gradient (in place on dy) + bias gradient in one pass: dy *= (y > 0 ? 1 : slope); out[c] (+)= sum_p dy[p, c]
extern "C" int pwc_lrelu_grad_channel_sums_f32(const float* y, int y_cs, float* dy, int dy_cs, long npix, int C, float slope,
                                               float* workspace, size_t workspace_floats, float* out, int accumulate,
                                               pwc_stream_t stream) {
    if (!y || !dy || !workspace || !out || npix <= 0 || C <= 0 || dy_cs < C || y_cs < C) return PWC_EINVAL;
    return channel_sums_launch(y, y_cs, slope, dy, dy_cs, npix, C, workspace, workspace_floats, out, accumulate, stream);
}

// ------------------------------------------------------------------ resize (legacy bilinear) backward
// Forward (pwc_resize_bilinear_f32, integer factor k = OH/H = OW/W): out[o] = in[lo] + (in[hi] - in[lo]) * t,
// lo = o / k, hi = min(lo + 1, n - 1), t = (o % k) / k, applied along x then y, times `mul`.  The transpose
// as a GATHER (no atomics): input pixel i collects the outputs whose lo or hi it is.
struct ResizeGradArgs {
    const float* dy;
    float* dx;
    int dy_cs, dx_cs;
    int N, H, W, C, k;
    float mul;
    int accumulate;
};

__device__ __forceinline__ int rg_taps(int i, int n, int k, int* o_first, float* w) {
    // outputs o in [k*(i-1)+1, k*i + k - 1] touch input i; weight of output o on input i:
    //   as lo (o/k == i): 1 - t;  as hi (min(o/k + 1, n-1) == i): t   (both when i == n-1 and o/k == n-1)
    const int first = max(k * (i - 1) + 1, 0), last = min(k * i + k - 1, k * n - 1);
    *o_first = first;
    int cnt = 0;
    for (int o = first; o <= last; ++o) {
        const int lo = o / k, hi = min(lo + 1, n - 1);
        const float t = (float)(o - lo * k) / (float)k;
        float ww = 0.f;
        if (lo == i) ww += 1.f - t;
        if (hi == i) ww += t;
        w[cnt++] = ww;
    }
    return cnt;
}

__global__ __launch_bounds__(256) void resize_grad_kernel(const ResizeGradArgs a) {
    const long total = (long)a.N * a.H * a.W * a.C;
    const int OW = a.W * a.k, OH = a.H * a.k;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % a.C);
        long r = e / a.C;
        const int ix = (int)(r % a.W);
        r /= a.W;
        const int iy = (int)(r % a.H);
        const int n = (int)(r / a.H);
        float wx[8], wy[8];
        int ox0, oy0;
        const int nx = rg_taps(ix, a.W, a.k, &ox0, wx), ny = rg_taps(iy, a.H, a.k, &oy0, wy);
        float s = 0.f;
        for (int j = 0; j < ny; ++j) {
            const float* row = a.dy + (((long)n * OH + oy0 + j) * OW) * a.dy_cs + c;
            float sr = 0.f;
            for (int i = 0; i < nx; ++i) sr += wx[i] * row[(long)(ox0 + i) * a.dy_cs];
            s += wy[j] * sr;
        }
        float* d = a.dx + (((long)n * a.H + iy) * a.W + ix) * a.dx_cs + c;
        *d = a.accumulate ? *d + s * a.mul : s * a.mul;
    }
}

extern "C" int pwc_resize_bilinear_grad_f32(const float* dy, int dy_cs, float* dx, int dx_cs, int N, int H, int W, int C,
                                            int OH, int OW, float mul, int accumulate, pwc_stream_t stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || dy_cs < C || dx_cs < C) return PWC_EINVAL;
    if (OH % H || OW % W || OH / H != OW / W || OH / H < 1 || OH / H > 4) return PWC_EUNSUPPORTED;   // integer factors 1..4
    ResizeGradArgs a;
    a.dy = dy; a.dx = dx; a.dy_cs = dy_cs; a.dx_cs = dx_cs; a.N = N; a.H = H; a.W = W; a.C = C; a.k = OH / H;
    a.mul = mul; a.accumulate = accumulate;
    long blocks = ((long)N * H * W * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resize_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return pwc_launch_status();
}

// ------------------------------------------------------------------ bilinear warp backward
// out = sum_ij c_ij * x[corner_ij]  (modules.py:107-137); floor and clip have zero gradient, so
//   dx[corner_ij] += c_ij * dy                       (scatter: several pixels may share a corner.  With a workspace the
//                                                     contributions are added as 64-bit FIXED-POINT integers -- integer
//                                                     addition is associative, so the sums do not depend on the order
//                                                     in which the atomics land: bit-reproducible training steps; the
//                                                     fp32 atomics of the first version remain as the workspace-free form)
//   dflow_x = scale * sum_c dy * [(fy1-fy)(x01-x00) + (fy-fy0)(x11-x10)]
//   dflow_y = scale * sum_c dy * [(fx1-fx)(x10-x00) + (fx-fx0)(x11-x01)]
// One wave per pixel row segment: lane = channel quad, the flow gradient is reduced over the pixel's lanes.
struct WarpGradArgs {
    const float* x;
    const float* flow;
    const float* dy;
    float* dx;       // accumulated (atomics); may be null
    long long* fix;  // null, or N*H*W*C zeroed 64-bit accumulators for dx (dense, channel stride C): deterministic form
    float* dflow;    // 2 channels, written or accumulated
    int x_cs, flow_cs, dy_cs, dx_cs, dflow_cs;
    int N, H, W, C;
    float flow_scale;
    int dflow_accumulate;
};

__global__ __launch_bounds__(256) void warp_grad_kernel(const WarpGradArgs a) {
    // 8 lanes per pixel (each strides over the channel quads), 32 pixels per block
    const int sub = threadIdx.x & 7;
    const long pix = blockIdx.x * 32L + (threadIdx.x >> 3);
    const long npix = (long)a.N * a.H * a.W;
    float gx_sum = 0.f, gy_sum = 0.f;
    if (pix < npix) {
        const int gx = (int)(pix % a.W);
        const long r = pix / a.W;
        const int gy = (int)(r % a.H), n = (int)(r / a.H);
        const float* fp = a.flow + pix * a.flow_cs;
        const float fx = pwc_mul_rounded(fp[0], a.flow_scale), fy = pwc_mul_rounded(fp[1], a.flow_scale);
        const float fx0 = floorf(fx), fy0 = floorf(fy);
        const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
        const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
        const int y0 = (int)fminf(fmaxf((float)gy + fy0, 0.f), hl), y1 = (int)fminf(fmaxf((float)gy + fy1, 0.f), hl);
        const int x0 = (int)fminf(fmaxf((float)gx + fx0, 0.f), wl), x1 = (int)fminf(fmaxf((float)gx + fx1, 0.f), wl);
        const float wy0 = fy1 - fy, wy1 = fy - fy0, wx0 = fx1 - fx, wx1 = fx - fx0;
        const long img = (long)n * a.H * a.W;
        const long o00 = img + (long)y0 * a.W + x0, o01 = img + (long)y0 * a.W + x1;
        const long o10 = img + (long)y1 * a.W + x0, o11 = img + (long)y1 * a.W + x1;
        for (int c = sub * 4; c < a.C; c += 32) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.dy + pix * a.dy_cs + c);
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(a.x + o00 * a.x_cs + c);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(a.x + o01 * a.x_cs + c);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(a.x + o10 * a.x_cs + c);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(a.x + o11 * a.x_cs + c);
            const f32x4 dfx = wy0 * (v01 - v00) + wy1 * (v11 - v10);
            const f32x4 dfy = wx0 * (v10 - v00) + wx1 * (v11 - v01);
#pragma unroll
            for (int i = 0; i < 4; ++i) { gx_sum += g[i] * dfx[i]; gy_sum += g[i] * dfy[i]; }
            if (a.fix) {
                // 2^36 steps per unit: |sum| < 1.3e8 representable, 1.5e-11 resolution (fp32 gradients carry ~1e-7 relative)
                // Range of the fixed point: contributions below 1.5e-11 vanish, sums beyond +-1.3e8 would wrap, and
                // __float2ll_rn maps NaN to 0 / saturates Inf -- a diverging step must not come out finite: a
                // contribution that is not finite or reaches 2^26 raises the poison word behind the sums, and the finish
                // kernel then writes NaN to every dx it owns (what fp32 atomics would have produced downstream).
                constexpr float FIX = 68719476736.f;
                unsigned long long* f = reinterpret_cast<unsigned long long*>(a.fix);
                {
                    const bool ok = (fabsf(g[0]) < 67108864.f) & (fabsf(g[1]) < 67108864.f) & (fabsf(g[2]) < 67108864.f) &
                                    (fabsf(g[3]) < 67108864.f);            // (every comparison is false for a NaN)
                    if (!ok) atomicOr(f + (long)a.N * a.H * a.W * a.C, 1ull);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    atomicAdd(f + o00 * a.C + c + i, (unsigned long long)__float2ll_rn(wy0 * wx0 * g[i] * FIX));
                    atomicAdd(f + o01 * a.C + c + i, (unsigned long long)__float2ll_rn(wy0 * wx1 * g[i] * FIX));
                    atomicAdd(f + o10 * a.C + c + i, (unsigned long long)__float2ll_rn(wy1 * wx0 * g[i] * FIX));
                    atomicAdd(f + o11 * a.C + c + i, (unsigned long long)__float2ll_rn(wy1 * wx1 * g[i] * FIX));
                }
            } else if (a.dx) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    atomicAdd(a.dx + o00 * a.dx_cs + c + i, wy0 * wx0 * g[i]);
                    atomicAdd(a.dx + o01 * a.dx_cs + c + i, wy0 * wx1 * g[i]);
                    atomicAdd(a.dx + o10 * a.dx_cs + c + i, wy1 * wx0 * g[i]);
                    atomicAdd(a.dx + o11 * a.dx_cs + c + i, wy1 * wx1 * g[i]);
                }
            }
        }
    }
    // reduce over the 8 lanes of the pixel
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        gx_sum += __shfl_xor(gx_sum, o, 64);
        gy_sum += __shfl_xor(gy_sum, o, 64);
    }
    if (pix < npix && sub == 0 && a.dflow) {
        float* d = a.dflow + pix * a.dflow_cs;
        const float vx = gx_sum * a.flow_scale, vy = gy_sum * a.flow_scale;
        d[0] = a.dflow_accumulate ? d[0] + vx : vx;
        d[1] = a.dflow_accumulate ? d[1] + vy : vy;
    }
}

// dx += fixed-point sums (deterministic form)
__global__ __launch_bounds__(256) void warp_grad_fix_finish_kernel(const long long* __restrict__ fix, float* __restrict__ dx,
                                                                   int dx_cs, long npix, int C) {
    const long total = npix * C;
    const bool poisoned = fix[total] != 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / C;
        const int c = (int)(i - p * C);
        // a sum at or beyond 2^61 (|dx| >= 2^25 from contributions that are each below 2^26) is where the 64-bit fixed-point
        // sum wraps: many in-range contributions to one cell can get there without tripping the per-contribution poison
        //  -- such a cell becomes NaN instead of a finite wrong value
        const long long v = fix[i];
        const bool wild = v >= (1LL << 61) || v <= -(1LL << 61);
        dx[p * dx_cs + c] += (poisoned || wild) ? __builtin_nanf("") : (float)((double)v * (1.0 / 68719476736.0));
    }
}

static int warp_grad_run(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                         const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                         int dflow_accumulate, int N, int H, int W, int C, long long* fix, pwc_stream_t stream);

extern "C" size_t pwc_warp_bilinear_grad_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return ((size_t)N * H * W * C + 1) * sizeof(long long);     // + one word: "a contribution was not representable"
}

extern "C" int pwc_warp_bilinear_grad_det_f32(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                                              const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                                              int dflow_accumulate, int N, int H, int W, int C, void* workspace,
                                              size_t workspace_bytes, pwc_stream_t stream) {
    if (dx && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 7u) ||
               workspace_bytes < pwc_warp_bilinear_grad_workspace_bytes(N, H, W, C)))
        return PWC_EINVAL;
    return warp_grad_run(x, x_cs, flow, flow_cs, flow_scale, dy, dy_cs, dx, dx_cs, dflow, dflow_cs, dflow_accumulate, N, H, W,
                         C, dx ? reinterpret_cast<long long*>(workspace) : nullptr, stream);
}

extern "C" int pwc_warp_bilinear_grad_f32(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                                          const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                                          int dflow_accumulate, int N, int H, int W, int C, pwc_stream_t stream) {
    return warp_grad_run(x, x_cs, flow, flow_cs, flow_scale, dy, dy_cs, dx, dx_cs, dflow, dflow_cs, dflow_accumulate, N, H, W,
                         C, nullptr, stream);
}

static int warp_grad_run(const float* x, int x_cs, const float* flow, int flow_cs, float flow_scale,
                         const float* dy, int dy_cs, float* dx, int dx_cs, float* dflow, int dflow_cs,
                         int dflow_accumulate, int N, int H, int W, int C, long long* fix, pwc_stream_t stream) {
    if (!x || !flow || !dy || N <= 0 || H <= 0 || W <= 0 || C <= 0) return PWC_EINVAL;
    if (x_cs < C || dy_cs < C || flow_cs < 2 || (dx && dx_cs < C) || (dflow && dflow_cs < 2)) return PWC_EINVAL;
    if ((C & 3) || (x_cs & 3) || (dy_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(dy)) return PWC_EALIGN;
    WarpGradArgs a;
    a.x = x; a.flow = flow; a.dy = dy; a.dx = dx; a.dflow = dflow; a.fix = fix;
    a.x_cs = x_cs; a.flow_cs = flow_cs; a.dy_cs = dy_cs; a.dx_cs = dx_cs; a.dflow_cs = dflow_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = flow_scale; a.dflow_accumulate = dflow_accumulate;
    const long npix = (long)N * H * W;
    const long blocks = (npix + 31) / 32;
    if (blocks >= (1L << 31)) return PWC_ERANGE;
    if (fix) {
        const hipError_t me = hipMemsetAsync(fix, 0, ((size_t)npix * C + 1) * sizeof(long long), (hipStream_t)stream);
        if (me != hipSuccess) { (void)hipGetLastError(); return (int)me; }      // the memset's OWN error, never PWC_OK
    }
    hipLaunchKernelGGL(warp_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    if (fix) {
        long fb = (npix * C + 255) / 256;
        if (fb > 8192) fb = 8192;
        hipLaunchKernelGGL(warp_grad_fix_finish_kernel, dim3((unsigned)fb), dim3(256), 0, (hipStream_t)stream,
                           (const long long*)fix, dx, dx_cs, npix, C);
    }
    return pwc_launch_status();
}

// ------------------------------------------------------------------ cost volume backward
// cv[y,x,d(v,h)] = lrelu( (1/C) sum_c f0[y,x,c] * f1[y+v,x+h,c] ), zero outside.  With
//   g[y,x,d] = dcv[y,x,d] * (cv[y,x,d] > 0 ? 1 : slope) / C :
//   df0[y,x,c]   (+)= sum_{v,h} g[y,x,d(v,h)]       * f1[y+v, x+h, c]
//   df1[y',x',c] (+)= sum_{v,h} g[y'-v,x'-h,d(v,h)] * f0[y'-v, x'-h, c]
// Both are gathers (no atomics).  Workgroup = 8 x 8 output pixels: the 16 x 16 neighbourhood of the OTHER map
// and the tile's (or neighbourhood's) g values go through LDS per 16-channel chunk; thread = (pixel, channel quad).
struct CvGradArgs {
    const float* f0;
    const float* f1;
    const float* cv;      // forward output (after leaky-relu)
    const float* dcv;
    float* df0;
    float* df1;
    int f0_cs, f1_cs, cv_cs, dcv_cs, df0_cs, df1_cs;
    int N, H, W, C;
    float slope;
    int accumulate;
    int tiles_x, tiles_y;
    int vec_g;            // cv / dcv records 16-byte aligned: float4 loads
};

// WHICH = 0: df0 (own pixel's g, neighbours of f1);  1: df1 (neighbours' g and f0)
template <int WHICH>
__global__ __launch_bounds__(256) void cost_volume_grad_kernel(const CvGradArgs a) {
    constexpr int R = 4, D = 9, T = 8, NB = T + 2 * R;        // 16 x 16 neighbourhood
    extern __shared__ __attribute__((aligned(16))) float cvg_smem[];
    float* s_nb = cvg_smem;                                    // other map: 16 channels + 4 padding floats per pixel
    float* s_g = cvg_smem + NB * NB * 20;                      // g of the tile (WHICH 0) / of the neighbourhood (WHICH 1)
    const int t = threadIdx.x;
    int blk = blockIdx.x;
    const int bx = blk % a.tiles_x;
    blk /= a.tiles_x;
    const int by = blk % a.tiles_y;
    const int n = blk / a.tiles_y;
    const int y0 = by * T, x0 = bx * T;
    const float inv_c = 1.f / (float)a.C;
    const long img = (long)n * a.H * a.W;
    const float* other = WHICH ? a.f0 : a.f1;
    const int other_cs = WHICH ? a.f0_cs : a.f1_cs;

    // ---- g values.  A pixel's 81 values are 20 float4 + 1 float when the records are 16-byte aligned (the estimator
    // buffers are): 21 pieces per pixel instead of 81 scalar loads; e / 21 as a multiply-shift (exact below 2^13).
    constexpr int NG = (WHICH ? NB * NB : T * T);
    if (a.vec_g) {
        for (int e = t; e < NG * 21; e += 256) {
            const int p = (e * 49933) >> 20, j = e - p * 21;
            const int py = WHICH ? p / NB : p / T, px = WHICH ? p - py * NB : p - py * T;
            const int y = WHICH ? y0 - R + py : y0 + py, x = WHICH ? x0 - R + px : x0 + px;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
                const long o = img + (long)y * a.W + x;
                f32x4 c, dv;
                if (j < 20) {
                    c = *reinterpret_cast<const f32x4*>(a.cv + o * a.cv_cs + 4 * j);
                    dv = *reinterpret_cast<const f32x4*>(a.dcv + o * a.dcv_cs + 4 * j);
                } else {
                    c = f32x4{a.cv[o * a.cv_cs + 80], 0.f, 0.f, 0.f};
                    dv = f32x4{a.dcv[o * a.dcv_cs + 80], 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = dv[i] * (c[i] > 0.f ? 1.f : a.slope) * inv_c;
            }
            float* dst = s_g + p * 81 + 4 * j;
            dst[0] = g[0];
            if (j < 20) { dst[1] = g[1]; dst[2] = g[2]; dst[3] = g[3]; }
        }
    } else {
        for (int e = t; e < NG * 81; e += 256) {
            const int p = (e * 25891) >> 21, d = e - p * 81;        // e / 81 (exact below 2^15)
            const int py = WHICH ? p / NB : p / T, px = WHICH ? p - py * NB : p - py * T;
            const int y = WHICH ? y0 - R + py : y0 + py, x = WHICH ? x0 - R + px : x0 + px;
            float g = 0.f;
            if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
                const long o = img + (long)y * a.W + x;
                const float c = a.cv[o * a.cv_cs + d];
                g = a.dcv[o * a.dcv_cs + d] * (c > 0.f ? 1.f : a.slope) * inv_c;
            }
            s_g[e] = g;
        }
    }
    const int pl = t >> 2, q = t & 3;                         // pixel of the tile, channel quad of the chunk
    const int ty = pl >> 3, tx = pl & 7;
    const int oy = y0 + ty, ox = x0 + tx;
    const bool valid = oy < a.H && ox < a.W;
    float* dst = WHICH ? a.df1 : a.df0;
    const int dst_cs = WHICH ? a.df1_cs : a.df0_cs;
    for (int c0 = 0; c0 < a.C; c0 += 16) {
        __syncthreads();
        // ---- 16 channels of the other map's neighbourhood
        for (int e = t; e < NB * NB * 4; e += 256) {
            const int p = e >> 2, cq = e & 3;
            const int py = p / NB, px = p - py * NB;
            const int y = y0 - R + py, x = x0 - R + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && c0 + cq * 4 < a.C)
                v = *reinterpret_cast<const f32x4*>(other + (img + (long)y * a.W + x) * other_cs + c0 + cq * 4);
            *reinterpret_cast<f32x4*>(s_nb + p * 20 + cq * 4) = v;
        }
        __syncthreads();
        if (c0 + q * 4 < a.C) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < D; ++v)
#pragma unroll
                for (int h = 0; h < D; ++h) {
                    if (WHICH == 0) {
                        // own g, f1 at (ty + v, tx + h) of the neighbourhood
                        const float g = s_g[pl * 81 + v * D + h];
                        const f32x4 w = *reinterpret_cast<const f32x4*>(s_nb + ((ty + v) * NB + tx + h) * 20 + q * 4);
                        acc += g * w;
                    } else {
                        // source pixel (y' - (v-4), x' - (h-4)) = neighbourhood (ty + 8 - v, tx + 8 - h): its g for shift (v, h)
                        const int sp = (ty + 2 * R - v) * NB + tx + 2 * R - h;
                        const float g = s_g[sp * 81 + v * D + h];
                        const f32x4 w = *reinterpret_cast<const f32x4*>(s_nb + sp * 20 + q * 4);
                        acc += g * w;
                    }
                }
            if (valid) {
                float* d = dst + (img + (long)oy * a.W + ox) * dst_cs + c0 + q * 4;
                f32x4 o = acc;
                if (a.accumulate) o += *reinterpret_cast<const f32x4*>(d);
                *reinterpret_cast<f32x4*>(d) = o;
            }
        }
    }
}

extern "C" int pwc_cost_volume_grad_f32(const float* f0, int f0_cs, const float* f1w, int f1w_cs, const float* cv, int cv_cs,
                                        const float* dcv, int dcv_cs, float* df0, int df0_cs, float* df1w, int df1w_cs,
                                        int accumulate, int N, int H, int W, int C, int search_range, float slope,
                                        pwc_stream_t stream) {
    if (!f0 || !f1w || !cv || !dcv || (!df0 && !df1w)) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PWC_EINVAL;
    if (search_range != 4) return PWC_EUNSUPPORTED;
    if (f0_cs < C || f1w_cs < C || cv_cs < 81 || dcv_cs < 81 || (df0 && df0_cs < C) || (df1w && df1w_cs < C)) return PWC_EINVAL;
    if ((C & 3) || (f0_cs & 3) || (f1w_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1w)) return PWC_EALIGN;
    if ((df0 && ((df0_cs & 3) || !pwc_aligned16(df0))) || (df1w && ((df1w_cs & 3) || !pwc_aligned16(df1w)))) return PWC_EALIGN;
    CvGradArgs a;
    a.f0 = f0; a.f1 = f1w; a.cv = cv; a.dcv = dcv; a.df0 = df0; a.df1 = df1w;
    a.f0_cs = f0_cs; a.f1_cs = f1w_cs; a.cv_cs = cv_cs; a.dcv_cs = dcv_cs; a.df0_cs = df0_cs; a.df1_cs = df1w_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.slope = slope; a.accumulate = accumulate;
    a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
    a.vec_g = (!(cv_cs & 3) && !(dcv_cs & 3) && pwc_aligned16(cv) && pwc_aligned16(dcv)) ? 1 : 0;
    const long nblk = (long)N * a.tiles_x * a.tiles_y;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    const size_t lds0 = (size_t)(16 * 16 * 20 + 8 * 8 * 81) * 4, lds1 = (size_t)(16 * 16 * 20 + 16 * 16 * 81) * 4;
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_grad_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    }
    if (df0) hipLaunchKernelGGL(cost_volume_grad_kernel<0>, dim3((unsigned)nblk), dim3(256), lds0, (hipStream_t)stream, a);
    if (df1w) hipLaunchKernelGGL(cost_volume_grad_kernel<1>, dim3((unsigned)nblk), dim3(256), lds1, (hipStream_t)stream, a);
    return pwc_launch_status();
}

// ------------------------------------------------------------------ loss gradient
// d/dpred of  scale * sum_p || pred[p] - gt[nearest(p)] / gt_div ||_ord   (losses.py:4-8 inside :20-29,38-45):
//   ord 2: (pred - g) / ||pred - g||_2   (0 where the norm is 0);   ord 1: sign(pred - g)
struct FlowNormGradArgs {
    const float* pred;
    const float* gt;
    float* dpred;
    int pred_cs, gt_cs, dpred_cs;
    int N, H, W, GH, GW;
    float sy, sx, gt_div, scale;
    int ord, accumulate;
};

__global__ __launch_bounds__(256) void flow_norm_grad_kernel(const FlowNormGradArgs a) {
    const long npix = (long)a.N * a.H * a.W;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const int x = (int)(p % a.W);
        const long r = p / a.W;
        const int y = (int)(r % a.H), n = (int)(r / a.H);
        const int gy = min((int)floorf(pwc_mul_rounded((float)y, a.sy)), a.GH - 1);
        const int gx = min((int)floorf(pwc_mul_rounded((float)x, a.sx)), a.GW - 1);
        const float* pp = a.pred + p * a.pred_cs;
        const float* gp = a.gt + (((long)n * a.GH + gy) * a.GW + gx) * a.gt_cs;
        // the loss is norm(gt_down - pred): its gradient w.r.t. pred is -(gt_down - pred) / norm
        const float dx = pp[0] - gp[0] / a.gt_div, dy = pp[1] - gp[1] / a.gt_div;
        float ox, oy;
        if (a.ord == 1) {
            ox = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f);
            oy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
        } else {
            const float nrm = sqrtf(dx * dx + dy * dy);
            ox = nrm > 0.f ? dx / nrm : 0.f;
            oy = nrm > 0.f ? dy / nrm : 0.f;
        }
        float* d = a.dpred + p * a.dpred_cs;
        d[0] = a.accumulate ? d[0] + a.scale * ox : a.scale * ox;
        d[1] = a.accumulate ? d[1] + a.scale * oy : a.scale * oy;
    }
}

extern "C" int pwc_flow_norm_grad_f32(const float* pred, int pred_cs, const float* gt, int gt_cs, int N, int H, int W, int GH,
                                      int GW, float gt_div, int ord, float scale, float* dpred, int dpred_cs,
                                      int accumulate, pwc_stream_t stream) {
    if (!pred || !gt || !dpred || N <= 0 || H <= 0 || W <= 0 || GH <= 0 || GW <= 0) return PWC_EINVAL;
    if (pred_cs < 2 || gt_cs < 2 || dpred_cs < 2 || !(gt_div != 0.f)) return PWC_EINVAL;
    if (ord != 1 && ord != 2) return PWC_EUNSUPPORTED;
    FlowNormGradArgs a;
    a.pred = pred; a.gt = gt; a.dpred = dpred; a.pred_cs = pred_cs; a.gt_cs = gt_cs; a.dpred_cs = dpred_cs;
    a.N = N; a.H = H; a.W = W; a.GH = GH; a.GW = GW;
    a.sy = (float)GH / (float)H; a.sx = (float)GW / (float)W; a.gt_div = gt_div; a.scale = scale; a.ord = ord;
    a.accumulate = accumulate;
    long blocks = ((long)N * H * W + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(flow_norm_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return pwc_launch_status();
}

// ------------------------------------------------------------------ Adam
// tf.train.AdamOptimizer (TF 1.8 training/adam.py, reference train.py:90) over a FLAT parameter buffer, with the
// gradient of gamma * sum(l2_loss(var)) (train.py:75) folded in:  g <- g + gamma * p;
//   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - lr_t * m / (sqrt(v) + eps),
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the host.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                   float gamma, float gscale) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * gscale + gamma * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int pwc_adam_step_f32(float* params, const float* grads, float* m, float* v, long n, float lr_t, float beta1,
                                 float beta2, float eps, float l2_gamma, float grad_scale, pwc_stream_t stream) {
    if (!params || !grads || !m || !v || n <= 0) return PWC_EINVAL;
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n, lr_t,
                       beta1, beta2, eps, l2_gamma, grad_scale);
    return pwc_launch_status();
}
