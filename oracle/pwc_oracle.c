/*
 * pwc_oracle.c -- CPU restatement of the PWC-Net inference arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pwcnet_amd/ may import, link or
 * call this file; it is the checker for tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference (daigo0927/pwcnet) ships no tests, no golden
 * vectors and cannot be executed here (it needs tensorflow-gpu==1.8.0,
 * requirements.txt:63, which is absent).  Every function below restates the
 * TF-1.8 primitive the reference calls, with the reference call site cited.
 * It is cross-checked against an independent literal numpy restatement
 * (oracle/np_literal.py) and against torch CPU ops in tests/test_oracle.py.
 *
 * Layout everywhere: NHWC float32, kernels HWIO (3,3,Cin,Cout), as
 * tf.layers.Conv2D stores them (reference modules.py:62-66).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -fopenmp -shared).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_MAX_COUT 65536

int oracle_version(void) { return 1; }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* TF 'SAME' padding for one spatial axis (tensorflow/core/framework/
 * common_shape_fns.cc GetWindowedOutputSizeVerbose, restated):
 *   out = ceil(in / stride); k_eff = (k-1)*dilation + 1
 *   pad_total = max((out-1)*stride + k_eff - in, 0); pad_before = pad_total/2
 * Reference call sites: tf.layers.Conv2D(..., 'same') modules.py:62-66,267,
 * 274,306-324. */
static void same_pad(int in, int stride, int dilation, int* out, int* pad_before) {
    int o = (in + stride - 1) / stride;
    int k_eff = 2 * dilation + 1;
    int total = (o - 1) * stride + k_eff - in;
    if (total < 0) total = 0;
    *out = o;
    *pad_before = total / 2;
}

/* y[n,oy,ox,co] = act( bias[co] + sum_{ty,tx,ci} x[n, oy*s-pt+ty*d, ox*s-pl+tx*d, ci]
 *                                               * w[ty,tx,ci,co] )
 * act = leaky_relu(v, slope) = max(v, slope*v) when apply_act, identity otherwise
 * (tf.nn.leaky_relu, reference modules.py:63-67,268,308-323).
 * `residual` (may be NULL) is added AFTER the activation-less conv: it restates
 * `flows += flows_up_prev` (modules.py:275-277) and `flows + x` (modules.py:326).
 * x has channel stride x_cs (>= Cin) so a channel slice of a wider tensor can be
 * convolved; y is written dense (N,Ho,Wo,Cout). */
int oracle_conv3x3(const float* x, int N, int H, int W, int Cin, int x_cs,
                   const float* w_hwio, const float* bias, int Cout,
                   int stride, int dilation, int apply_act, float slope,
                   const float* residual, int res_cs, float* y) {
    if (Cout > ORACLE_MAX_COUT || Cout <= 0 || Cin <= 0) return -1;
    int Ho, Wo, pt, pl;
    same_pad(H, stride, dilation, &Ho, &pt);
    same_pad(W, stride, dilation, &Wo, &pl);
    enum { PB = 4 };
    const long rows = (long)N * Ho;
#pragma omp parallel for schedule(dynamic, 1)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / Ho), oy = (int)(r % Ho);
        float acc[PB][256];
        for (int ox0 = 0; ox0 < Wo; ox0 += PB) {
            int pb = Wo - ox0 < PB ? Wo - ox0 : PB;
            /* Cout > 256 is handled in column blocks of 256 */
            for (int c0 = 0; c0 < Cout; c0 += 256) {
                int cb = Cout - c0 < 256 ? Cout - c0 : 256;
                for (int p = 0; p < pb; ++p)
                    for (int co = 0; co < cb; ++co) acc[p][co] = bias ? bias[c0 + co] : 0.0f;
                for (int ty = 0; ty < 3; ++ty) {
                    int iy = oy * stride - pt + ty * dilation;
                    if (iy < 0 || iy >= H) continue;
                    for (int tx = 0; tx < 3; ++tx) {
                        const float* wt = w_hwio + ((size_t)(ty * 3 + tx) * Cin) * Cout + c0;
                        const float* xp[PB];
                        int any = 0;
                        for (int p = 0; p < pb; ++p) {
                            int ix = (ox0 + p) * stride - pl + tx * dilation;
                            if (ix < 0 || ix >= W) { xp[p] = NULL; continue; }
                            xp[p] = x + (((size_t)n * H + iy) * W + ix) * x_cs;
                            any = 1;
                        }
                        if (!any) continue;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float* wr = wt + (size_t)ci * Cout;
                            for (int p = 0; p < pb; ++p) {
                                if (!xp[p]) continue;
                                float xv = xp[p][ci];
                                float* a = acc[p];
                                for (int co = 0; co < cb; ++co) a[co] += xv * wr[co];
                            }
                        }
                    }
                }
                for (int p = 0; p < pb; ++p) {
                    size_t pix = ((size_t)n * Ho + oy) * Wo + ox0 + p;
                    float* yo = y + pix * Cout + c0;
                    for (int co = 0; co < cb; ++co) {
                        float v = acc[p][co];
                        if (apply_act) v = v > slope * v ? v : slope * v;
                        if (residual) v += residual[pix * res_cs + c0 + co];
                        yo[co] = v;
                    }
                }
            }
        }
    }
    return 0;
}

/* Cost volume, reference modules.py:158-204 (CostVolumeLayer.__call__ + get_cost):
 *   cv[n,y,x,(v+R)*(2R+1)+(h+R)] = lrelu_slope( (1/C) * sum_c f0[n,y,x,c]*f1[n,y+v,x+h,c] )
 * zero where (y+v,x+h) is outside the image (tf.pad zero padding, modules.py:159,
 * 176-177); v is the OUTER loop (modules.py:197), h the inner (modules.py:198);
 * reduce_mean divides by C (modules.py:181); leaky_relu 0.1 (modules.py:203). */
int oracle_cost_volume(const float* f0, const float* f1, int N, int H, int W, int C,
                       int search_range, float slope, float* cv) {
    const int R = search_range, D = 2 * R + 1;
    const long rows = (long)N * H;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / H), y = (int)(r % H);
        for (int x = 0; x < W; ++x) {
            const float* a = f0 + (((size_t)n * H + y) * W + x) * C;
            float* o = cv + (((size_t)n * H + y) * W + x) * (D * D);
            for (int v = -R; v <= R; ++v)
                for (int h = -R; h <= R; ++h) {
                    int yy = y + v, xx = x + h;
                    float s = 0.0f;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const float* b = f1 + (((size_t)n * H + yy) * W + xx) * C;
                        for (int c = 0; c < C; ++c) s += a[c] * b[c];
                    }
                    s = s / (float)C;
                    o[(v + R) * D + (h + R)] = s > slope * s ? s : slope * s;
                }
        }
    }
    return 0;
}

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* bilinear_warp, reference modules.py:99-137.  flow[...,0]=fx, flow[...,1]=fy
 * (modules.py:106); the caller's `flows_up*self.scales[l]` (model.py:109) is the
 * `flow_scale` multiply here.  Corner indices are clipped independently and cast
 * to int32 (modules.py:116-124); weights use the UN-clipped floors
 * (modules.py:132-135); sum order c00*x00 + c01*x01 + c10*x10 + c11*x11
 * (modules.py:137). */
int oracle_warp_bilinear(const float* x, const float* flow, int flow_cs, float flow_scale,
                         int N, int H, int W, int C, float* out) {
    const long rows = (long)N * H;
    const float hlim = (float)(H - 1), wlim = (float)(W - 1);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / H), gy = (int)(r % H);
        for (int gx = 0; gx < W; ++gx) {
            size_t pix = ((size_t)n * H + gy) * W + gx;
            float fx = flow[pix * flow_cs + 0] * flow_scale;
            float fy = flow[pix * flow_cs + 1] * flow_scale;
            float fx0 = floorf(fx), fx1 = fx0 + 1.0f;
            float fy0 = floorf(fy), fy1 = fy0 + 1.0f;
            int y0 = (int)clampf((float)gy + fy0, 0.0f, hlim);
            int y1 = (int)clampf((float)gy + fy1, 0.0f, hlim);
            int x0 = (int)clampf((float)gx + fx0, 0.0f, wlim);
            int x1 = (int)clampf((float)gx + fx1, 0.0f, wlim);
            float c00 = (fy1 - fy) * (fx1 - fx), c01 = (fy1 - fy) * (fx - fx0);
            float c10 = (fy - fy0) * (fx1 - fx), c11 = (fy - fy0) * (fx - fx0);
            const float* p00 = x + (((size_t)n * H + y0) * W + x0) * C;
            const float* p01 = x + (((size_t)n * H + y0) * W + x1) * C;
            const float* p10 = x + (((size_t)n * H + y1) * W + x0) * C;
            const float* p11 = x + (((size_t)n * H + y1) * W + x1) * C;
            float* o = out + pix * C;
            for (int c = 0; c < C; ++c)
                o[c] = c00 * p00[c] + c01 * p01[c] + c10 * p10[c] + c11 * p11[c];
        }
    }
    return 0;
}

/* nearest_warp, reference modules.py:83-97: tf.cast(flow, int32) truncates toward
 * zero (modules.py:85), then add to the grid and clip (modules.py:87-92). */
int oracle_warp_nearest(const float* x, const float* flow, int flow_cs, float flow_scale,
                        int N, int H, int W, int C, float* out) {
    const long rows = (long)N * H;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / H), gy = (int)(r % H);
        for (int gx = 0; gx < W; ++gx) {
            size_t pix = ((size_t)n * H + gy) * W + gx;
            int ifx = (int)(flow[pix * flow_cs + 0] * flow_scale);
            int ify = (int)(flow[pix * flow_cs + 1] * flow_scale);
            int yy = gy + ify, xx = gx + ifx;
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            memcpy(out + pix * C, x + (((size_t)n * H + yy) * W + xx) * C, sizeof(float) * C);
        }
    }
    return 0;
}

/* tf.image.resize_bilinear, TF 1.8, align_corners=False (legacy, no half-pixel
 * centres); tensorflow/core/kernels/resize_bilinear_op.cc restated:
 *   scale = in/out (float); src = dst*scale; lo = (int)floor(src);
 *   hi = min(lo+1, in-1); lerp = src - lo
 *   top = tl + (tr-tl)*xl; bot = bl + (br-bl)*xl; out = top + (bot-top)*yl
 * Reference call sites: modules.py:283-284 (x2, no value scaling) and
 * model.py:127 (x4 then `*20.` == `mul`). */
int oracle_resize_bilinear(const float* x, int x_cs, int N, int H, int W, int C,
                           int OH, int OW, float mul, float* y) {
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const long rows = (long)N * OH;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        int n = (int)(r / OH), oy = (int)(r % OH);
        float fy = (float)oy * sy;
        int y0 = (int)floorf(fy);
        int y1 = y0 + 1 < H - 1 ? y0 + 1 : H - 1;
        float yl = fy - (float)y0;
        for (int ox = 0; ox < OW; ++ox) {
            float fx = (float)ox * sx;
            int x0 = (int)floorf(fx);
            int x1 = x0 + 1 < W - 1 ? x0 + 1 : W - 1;
            float xl = fx - (float)x0;
            const float* tl = x + (((size_t)n * H + y0) * W + x0) * x_cs;
            const float* tr = x + (((size_t)n * H + y0) * W + x1) * x_cs;
            const float* bl = x + (((size_t)n * H + y1) * W + x0) * x_cs;
            const float* br = x + (((size_t)n * H + y1) * W + x1) * x_cs;
            float* o = y + (((size_t)n * OH + oy) * OW + ox) * C;
            for (int c = 0; c < C; ++c) {
                float top = tl[c] + (tr[c] - tl[c]) * xl;
                float bot = bl[c] + (br[c] - bl[c]) * xl;
                o[c] = (top + (bot - top) * yl) * mul;
            }
        }
    }
    return 0;
}
