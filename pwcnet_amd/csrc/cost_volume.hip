// cost_volume.hip -- the +-R px correlation ("cost volume") of PWC-Net for gfx950, with
// an optional fused bilinear warp of the second feature map.
//
// Replaces CostVolumeLayer.__call__ / get_cost / pad2d / crop2d (reference
// modules.py:158-204: 81 x (2 tf.pad + multiply + Cropping2D + reduce_mean) + tf.stack +
// leaky_relu) and, in the fused form, WarpingLayer/bilinear_warp (modules.py:99-137)
// as called from model.py:109-112.
//
//   out[n,y,x,(v+R)*D+(h+R)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] )
//
// HBM-bound op (AI = 2*81*C / ((2C+81)*4) = 8.9 flop/B at C = 32).  Work decomposition:
//   workgroup = 4 x 64 output pixels of one image, D = 2R+1 waves; wave w owns the
//     vertical shift v = w - R (wave-uniform), lane = a strip of 4 consecutive pixels,
//     so one lane accumulates 4 pixels x D horizontal shifts = 36 outputs in VGPRs;
//   channels are processed in chunks of 16.  Per chunk the f1w halo tile
//     (4+2R) x (64+8) and the f0 tile are loaded NHWC-coalesced from HBM and TRANSPOSED
//     into channel planes in LDS ([c][y][x], x contiguous), so the inner loop is 3
//     ds_read_b128 (12-pixel window) + 1 ds_read_b128 (f0) per 36 FMAs, conflict-free:
//     the lane -> strip map follows the four 16-lane groups a ds_read_b128 is serviced in;
//   the 256 x 81 result tile goes back through LDS so that HBM stores are contiguous
//     (324 B per pixel) instead of 4-byte scatters.
#include "pwc_common.h"

struct CvArgs {
    const float* f0;
    const float* f1;
    const float* flow;   // fused warp only
    float* out;
    int f0_cs, f1_cs, flow_cs, out_cs;
    int N, H, W, C;
    float flow_scale;
    float slope;
    int tiles_x, tiles_y;
};

constexpr int CV_TH = 4, CV_TW = 64, CV_CK = 16, CV_HW = CV_TW + 8;

__device__ __forceinline__ f32x4 cv_warp_gather(const float* f1n, int f1_cs, int c, int H, int W,
                                                int y, int x, float fx, float fy) {
    // bilinear_warp, modules.py:107-137: un-clipped floors give the weights, the four
    // corner indices are clipped independently.
    const float fx0 = floorf(fx), fy0 = floorf(fy);
    const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
    const float hl = (float)(H - 1), wl = (float)(W - 1);
    const int y0 = (int)fminf(fmaxf((float)y + fy0, 0.f), hl);
    const int y1 = (int)fminf(fmaxf((float)y + fy1, 0.f), hl);
    const int x0 = (int)fminf(fmaxf((float)x + fx0, 0.f), wl);
    const int x1 = (int)fminf(fmaxf((float)x + fx1, 0.f), wl);
    const float c00 = (fy1 - fy) * (fx1 - fx), c01 = (fy1 - fy) * (fx - fx0);
    const float c10 = (fy - fy0) * (fx1 - fx), c11 = (fy - fy0) * (fx - fx0);
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(f1n + ((size_t)y0 * W + x0) * f1_cs + c);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(f1n + ((size_t)y0 * W + x1) * f1_cs + c);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(f1n + ((size_t)y1 * W + x0) * f1_cs + c);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(f1n + ((size_t)y1 * W + x1) * f1_cs + c);
    return c00 * v00 + c01 * v01 + c10 * v10 + c11 * v11;
}

template <int R, bool FUSED>
__global__ __launch_bounds__(64 * (2 * R + 1)) void cost_volume_kernel(const CvArgs a) {
    constexpr int D = 2 * R + 1, DD = D * D;
    constexpr int T = 64 * D;
    constexpr int HH = CV_TH + 2 * R;                        // halo rows
    constexpr int PS1 = ((HH * CV_HW + 24 + 31) / 32) * 32;  // f1 plane stride (floats)
    constexpr int PS0 = ((CV_TH * CV_TW + 24 + 31) / 32) * 32;
    constexpr int F0_BASE = CV_CK * PS1;
    constexpr int N1 = HH * CV_HW * 4;                       // f1 load items (pixel x 4-ch quad)
    constexpr int N0 = CV_TH * CV_TW * 4;
    constexpr int NI = N1 + N0;
    static_assert(CV_CK * (PS1 + PS0) >= 128 * DD, "output stage must fit in the tile LDS");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // vertical shift index, v = w - R

    // tile decode (XCD-aware: consecutive logical tiles share an XCD's L2)
    const int nb = a.tiles_x * a.tiles_y * a.N;
    const int lb = pwc_xcd_remap(blockIdx.x, nb);
    const int tx_ = lb % a.tiles_x;
    const int ty_ = (lb / a.tiles_x) % a.tiles_y;
    const int n = lb / (a.tiles_x * a.tiles_y);
    const int x0 = tx_ * CV_TW, y0 = ty_ * CV_TH;

    const float* f0n = a.f0 + (size_t)n * a.H * a.W * a.f0_cs;
    const float* f1n = a.f1 + (size_t)n * a.H * a.W * a.f1_cs;
    const float* fln = FUSED ? a.flow + (size_t)n * a.H * a.W * a.flow_cs : nullptr;

    // lane -> (row, strip): each ds_read_b128 service group (16 lanes) reads one row's
    // 16 consecutive 16-byte slots.
    const int b = (lane & 31) >> 2;
    const int row = 2 * (lane >> 5) + (__builtin_popcount(b) & 1);
    const int x4 = (b >> 1) * 4 + (lane & 3);

    float acc[4][D];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < D; ++h) acc[i][h] = 0.f;

    for (int c0 = 0; c0 < a.C; c0 += CV_CK) {
        // ---------------- load + transpose this channel chunk into LDS planes
        constexpr int U = 4;
        for (int q0 = t; q0 < NI; q0 += T * U) {
            f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * T;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[u] = -1;
                if (q < N1) {
                    const int cq = q & 3, px = q >> 2;
                    const int py = px / CV_HW, pxx = px - py * CV_HW;
                    const int y = y0 - R + py, x = x0 - 4 + pxx;
                    dst[u] = cq * 4 * PS1 + 8 * cq + py * CV_HW + pxx;
                    const int c = c0 + cq * 4;
                    if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && c < a.C) {
                        if (FUSED) {
                            const float* fp = fln + ((size_t)y * a.W + x) * a.flow_cs;
                            v[u] = cv_warp_gather(f1n, a.f1_cs, c, a.H, a.W, y, x, fp[0] * a.flow_scale,
                                                  fp[1] * a.flow_scale);
                        } else {
                            v[u] = *reinterpret_cast<const f32x4*>(f1n + ((size_t)y * a.W + x) * a.f1_cs + c);
                        }
                    }
                } else if (q < NI) {
                    const int qq = q - N1;
                    const int cq = qq & 3, px = qq >> 2;
                    const int py = px >> 6, pxx = px & 63;
                    const int y = y0 + py, x = x0 + pxx;
                    dst[u] = F0_BASE + cq * 4 * PS0 + 8 * cq + py * CV_TW + pxx;
                    const int c = c0 + cq * 4;
                    if (y < a.H && x < a.W && c < a.C)
                        v[u] = *reinterpret_cast<const f32x4*>(f0n + ((size_t)y * a.W + x) * a.f0_cs + c);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] >= 0) {
                    const int ps = dst[u] >= F0_BASE ? PS0 : PS1;
                    smem[dst[u]] = v[u][0];
                    smem[dst[u] + ps] = v[u][1];
                    smem[dst[u] + 2 * ps] = v[u][2];
                    smem[dst[u] + 3 * ps] = v[u][3];
                }
            }
        }
        __syncthreads();

        // ---------------- correlate: 16 channels x (4 pixels x D shifts) per lane
        const float* p1 = smem + (row + w) * CV_HW + 4 * x4;
        const float* p0 = smem + F0_BASE + row * CV_TW + 4 * x4;
#pragma unroll
        for (int c = 0; c < CV_CK; ++c) {
            const int sk = 8 * (c >> 2);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0 + c * PS0 + sk);
            float win[12];
            *reinterpret_cast<f32x4*>(&win[0]) = *reinterpret_cast<const f32x4*>(p1 + c * PS1 + sk);
            *reinterpret_cast<f32x4*>(&win[4]) = *reinterpret_cast<const f32x4*>(p1 + c * PS1 + sk + 4);
            *reinterpret_cast<f32x4*>(&win[8]) = *reinterpret_cast<const f32x4*>(p1 + c * PS1 + sk + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < D; ++h) acc[i][h] = fmaf(a0[i], win[i + h + 4 - R], acc[i][h]);
        }
        __syncthreads();
    }

    // ---------------- mean over C, leaky-relu, stage through LDS, contiguous stores
    const float fC = (float)a.C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < D; ++h) acc[i][h] = pwc_lrelu(acc[i][h] / fC, a.slope);

    for (int hf = 0; hf < 2; ++hf) {
        if ((row >> 1) == hf) {
            float* st = smem + (((row & 1) * CV_TW + 4 * x4) * DD) + w * D;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < D; ++h) st[i * DD + h] = acc[i][h];
        }
        __syncthreads();
        for (int e = t; e < 128 * DD; e += T) {
            const int px = e / DD, d = e - px * DD;
            const int y = y0 + 2 * hf + (px >> 6), x = x0 + (px & 63);
            if (y < a.H && x < a.W) a.out[(((size_t)n * a.H + y) * a.W + x) * a.out_cs + d] = smem[e];
        }
        __syncthreads();
    }
}

template <int R, bool FUSED>
static int cv_launch(const CvArgs& a, hipStream_t s) {
    constexpr int D = 2 * R + 1;
    constexpr int HH = CV_TH + 2 * R;
    constexpr int PS1 = ((HH * CV_HW + 24 + 31) / 32) * 32;
    constexpr int PS0 = ((CV_TH * CV_TW + 24 + 31) / 32) * 32;
    const size_t lds = (size_t)CV_CK * (PS1 + PS0) * sizeof(float);
    static bool attr_set = false;   // idempotent, benign if raced
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_kernel<R, FUSED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned nb = (unsigned)(a.tiles_x * a.tiles_y * a.N);
    hipLaunchKernelGGL((cost_volume_kernel<R, FUSED>), dim3(nb), dim3(64 * D), lds, s, a);
    return pwc_launch_status();
}

static int cv_dispatch(CvArgs& a, int R, bool fused, hipStream_t s) {
    a.tiles_x = (a.W + CV_TW - 1) / CV_TW;
    a.tiles_y = (a.H + CV_TH - 1) / CV_TH;
    if ((long)a.tiles_x * a.tiles_y * a.N >= (1L << 31)) return PWC_ERANGE;
#define PWC_CV(RR)                                                   \
    case RR:                                                         \
        return fused ? cv_launch<RR, true>(a, s) : cv_launch<RR, false>(a, s);
    switch (R) {
        PWC_CV(1)
        PWC_CV(2)
        PWC_CV(3)
        PWC_CV(4)
        default: return PWC_EUNSUPPORTED;
    }
#undef PWC_CV
}

static int cv_check(const float* f0, int f0_cs, const float* f1, int f1_cs, float* out, int out_cs, int N,
                    int H, int W, int C, int R) {
    if (!f0 || !f1 || !out) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PWC_EINVAL;
    if (R < 1 || R > 4) return PWC_EUNSUPPORTED;
    if (f0_cs < C || f1_cs < C || out_cs < (2 * R + 1) * (2 * R + 1)) return PWC_EINVAL;
    if ((C & 3) || (f0_cs & 3) || (f1_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1)) return PWC_EALIGN;
    return PWC_OK;
}

extern "C" int pwc_cost_volume_f32(const float* f0, int f0_cs, const float* f1w, int f1w_cs, float* out,
                                   int out_cs, int N, int H, int W, int C, int search_range, float slope,
                                   pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1w, f1w_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    CvArgs a;
    a.f0 = f0; a.f1 = f1w; a.flow = nullptr; a.out = out;
    a.f0_cs = f0_cs; a.f1_cs = f1w_cs; a.flow_cs = 0; a.out_cs = out_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = 1.f; a.slope = slope;
    return cv_dispatch(a, search_range, false, (hipStream_t)stream);
}

extern "C" int pwc_warp_cost_volume_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                        const float* flow, int flow_cs, float flow_scale, float* out,
                                        int out_cs, int N, int H, int W, int C, int search_range, float slope,
                                        pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1, f1_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if (!flow || flow_cs < 2) return PWC_EINVAL;
    CvArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = flow_scale; a.slope = slope;
    return cv_dispatch(a, search_range, true, (hipStream_t)stream);
}
