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
#include <cstdlib>

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

// ---------------------------------------------------------------- Cin = 3 (first layer)
// modules.py:62, l = 0: 3 -> 16 channels, stride 2, on the full-resolution images: the
// HBM-bound head of the extractor (reads 12 B, writes 64 B per output pixel).  Lane =
// (output pixel, 4-channel quad): consecutive lanes store consecutive 16 bytes, so the
// NHWC output is written in whole contiguous lines; the 27 x Cout weights sit in LDS.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_smallcin_kernel(const DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];   // [27*CIN/3... = 9*CIN][Cout]
    const int nw = 9 * CIN * COUT;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) wlds[i] = a.w[i];
    __syncthreads();
    constexpr int qpp = COUT >> 2;
    const long total = a.M * qpp;
    const long HoWo = (long)a.Ho * a.Wo;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(gid % qpp);
        const long m = gid / qpp;
        const int n = (int)(m / HoWo);
        const int rem = (int)(m - (long)n * HoWo);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        f32x4 acc = *reinterpret_cast<const f32x4*>(a.bias + cq * 4);
        const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int iy = oy * a.stride - a.pad_t + ty * a.dil;
            const bool yok = (unsigned)iy < (unsigned)a.H;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int ix = ox * a.stride - a.pad_l + tx * a.dil;
                const bool ok = yok && ((unsigned)ix < (unsigned)a.W);
                const float* xp = xn + ((size_t)(ok ? iy : 0) * a.W + (ok ? ix : 0)) * a.x_cs;
                float xv[CIN];
                if (CIN == 3) {
                    // one 12-byte load per tap (pixels are 4-byte aligned triples)
                    struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
                    const f3 v = *reinterpret_cast<const f3*>(xp);
                    xv[0] = ok ? v.x : 0.f; xv[1] = ok ? v.y : 0.f; xv[2] = ok ? v.z : 0.f;
                } else {
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) xv[ci] = ok ? xp[ci] : 0.f;
                }
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wlds + ((ty * 3 + tx) * CIN + ci) * COUT + cq * 4);
                    acc += xv[ci] * w4;
                }
            }
        }
        if (a.apply_act) {
            acc[0] = pwc_lrelu(acc[0], a.slope); acc[1] = pwc_lrelu(acc[1], a.slope);
            acc[2] = pwc_lrelu(acc[2], a.slope); acc[3] = pwc_lrelu(acc[3], a.slope);
        }
        *reinterpret_cast<f32x4*>(a.y + (size_t)m * a.y_cs + cq * 4) = acc;
    }
}

// ---------------------------------------------------------------- Cout = 2 flow heads
// modules.py:274 / :324 (+ residual adds :275-277, :326) with Cin = 32: 8 lanes share one
// output pixel, lane q holds input channels 4q..4q+3 (one coalesced 128-byte read per
// pixel tap) and its 9 x 4 x 2 weights in VGPRs; the 8 partial sums are combined with
// wave shuffles; persistent grid-stride loop over pixel groups.
__global__ __launch_bounds__(256) void conv3x3_head2_c32_kernel(const DirectArgs a) {
    const int lane = threadIdx.x & 63;
    const int q = lane & 7;
    float w0[9][4], w1[9][4];
    // HWIO (3,3,32,2): the 4 channels x 2 couts of lane q are 8 contiguous floats per tap
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.w + ((size_t)tap * 32 + q * 4) * 2);
        const f32x4 lo = wp[0], hi = wp[1];
        w0[tap][0] = lo[0]; w1[tap][0] = lo[1]; w0[tap][1] = lo[2]; w1[tap][1] = lo[3];
        w0[tap][2] = hi[0]; w1[tap][2] = hi[1]; w0[tap][3] = hi[2]; w1[tap][3] = hi[3];
    }
    const float b0 = a.bias[0], b1 = a.bias[1];
    // 32-bit index math (the host routes here only when M and the per-image extent fit in 31 bits)
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

// Tile geometry of the LDS-staged flow-head kernel for wide inputs below ((8+2) x (32+2)-pixel patches, quad-major planes).
// (The 32-channel heads of the large levels ran a kernel of this shape in rounds 1-2 -- 72 ds_read_b128 + 576 FMAs per
// output -- and take conv3x3_head2_mfma_kernel now.)
constexpr int HT_R = 8, HT_C = 32, HT_PW = HT_C + 2, HT_NP = (HT_R + 2) * HT_PW, HT_PLANE = HT_NP * 4 + 4;

// Flow heads of the dense-connection estimators (modules.py:269-274 with use_dc: Cin = 725 ... 3169 -> 2): the same
// tile decomposition with a loop over 32-channel chunks.  The generic kernel below reads every input value 9 times
// with one thread per pixel walking all channels (3.1 ms average, 10 ms at level 4: a third of the use_dc forward);
// here the next chunk's patch is fetched into registers while the current one is reduced from LDS.
__global__ __launch_bounds__(256) void conv3x3_head2_wide_kernel(const DirectArgs a, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float patch[8 * HT_PLANE];
    __shared__ __attribute__((aligned(16))) float wsm[9 * 64];   // this chunk's weights [tap][32 ci][2 co] (broadcast reads)
    constexpr int NLD = (HT_NP * 8 + 255) / 256;             // b128 pieces per thread and chunk (11)
    const int t = threadIdx.x;
    int blk = blockIdx.x;
    const int bx = blk % tiles_x; blk /= tiles_x;
    const int by = blk % tiles_y;
    const int n = blk / tiles_y;
    const int y0 = by * HT_R, x0 = bx * HT_C;
    const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
    // this thread's pieces: patch pixel and channel quad (fixed over the chunks), global offset or -1
    long goff[NLD];
    int lslot[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = t + 256 * j;
        const int p = e >> 3, q = e & 7;
        const int py = p / HT_PW, px = p - py * HT_PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = e < HT_NP * 8 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        goff[j] = ok ? ((long)y * a.W + x) * a.x_cs + q * 4 : -1;
        lslot[j] = e < HT_NP * 8 ? q * HT_PLANE + p * 4 : -1;
    }
    f32x4 st[NLD];
    f32x4 wst = {0.f, 0.f, 0.f, 0.f};                        // threads 0..143: one float4 of the chunk's 9 x 64 weights
    const int wtap = t >> 4, wq4 = t & 15;                   // tap, float4 within the tap's 64 floats (2 channels x 2 co)
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (goff[j] >= 0 && c0 + ((t + 256 * j) & 7) * 4 < a.Cin) v = *reinterpret_cast<const f32x4*>(xn + goff[j] + c0);
            st[j] = v;
        }
        if (t < 144) {
            const float* wp = a.w + ((size_t)wtap * a.Cin + c0) * 2 + wq4 * 4;     // channels c0 + 2*wq4, +1
            const int ch = c0 + wq4 * 2;
            wst[0] = ch < a.Cin ? wp[0] : 0.f; wst[1] = ch < a.Cin ? wp[1] : 0.f;
            wst[2] = ch + 1 < a.Cin ? wp[2] : 0.f; wst[3] = ch + 1 < a.Cin ? wp[3] : 0.f;
        }
    };
    const int r = t >> 5, c = t & 31;
    float s0a = 0.f, s0b = 0.f, s1a = 0.f, s1b = 0.f;
    fetch(0);
    for (int c0 = 0; c0 < a.Cin; c0 += 32) {
        __syncthreads();                                      // the previous chunk has been read
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (lslot[j] >= 0) *reinterpret_cast<f32x4*>(patch + lslot[j]) = st[j];
        if (t < 144) *reinterpret_cast<f32x4*>(wsm + wtap * 64 + wq4 * 4) = wst;
        __syncthreads();
        if (c0 + 32 < a.Cin) fetch(c0 + 32);
        const int nq = min(8, (a.Cin - c0) >> 2);             // channel quads of this chunk (Cin % 4 == 0)
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const float* pp = patch + ((r + ty) * HT_PW + c + tx) * 4;
                const float* wt = wsm + (ty * 3 + tx) * 64;                           // [32 ci][2 co] of this tap and chunk
                for (int q = 0; q < nq; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(pp + q * HT_PLANE);
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(wt + q * 8), wb = *reinterpret_cast<const f32x4*>(wt + q * 8 + 4);
                    s0a = fmaf(v[0], wa[0], s0a); s1a = fmaf(v[0], wa[1], s1a);
                    s0b = fmaf(v[1], wa[2], s0b); s1b = fmaf(v[1], wa[3], s1b);
                    s0a = fmaf(v[2], wb[0], s0a); s1a = fmaf(v[2], wb[1], s1a);
                    s0b = fmaf(v[3], wb[2], s0b); s1b = fmaf(v[3], wb[3], s1b);
                }
            }
    }
    const int oy = y0 + r, ox = x0 + c;
    if (oy < a.H && ox < a.W) {
        const size_t m = ((size_t)n * a.H + oy) * a.W + ox;
        float v0 = (s0a + s0b) + a.bias[0], v1 = (s1a + s1b) + a.bias[1];
        if (a.apply_act) { v0 = pwc_lrelu(v0, a.slope); v1 = pwc_lrelu(v1, a.slope); }
        if (a.res) { v0 += a.res[m * a.res_cs]; v1 += a.res[m * a.res_cs + 1]; }
        a.y[m * a.y_cs] = v0;
        a.y[m * a.y_cs + 1] = v1;
    }
}

// ---------------------------------------------------------------- 3 -> 16 channels on the matrix pipe
// The first convolution of the extractor (reference modules.py:64, `fp_extractor/conv2d`: 3 -> 16 channels, stride 2,
// 448x1024 images) as a GEMM per 16 consecutive output pixels of a row: D[cout 16][pixel 16] = W[16][K] x X[K][16],
// K = 27 = (tap, channel) padded to 28 = 7 x v_mfma_f32_16x16x4_f32.  conv3x3_smallcin_kernel spends ~150 VALU / LDS
// instructions per 16 pixels x 16 couts (4 threads per pixel); here a wave spends 7 loads, 7 MFMAs and a 10-instruction
// epilogue (measured 31 us against 38 us on 8 x 448 x 1024; with the next group's loads issued ahead: 34 us).
//   A (weights): lane (cout = lane & 15, k = 4 i + (lane >> 4)), 7 registers for the whole kernel;
//   B (pixels):  lane (k = 4 i + (lane >> 4), pixel = lane & 15): one buffer_load_dword per k-step whose per-lane offset
//                is a kernel constant and whose scalar offset is the group's origin -- no address arithmetic in the loop
//                (groups that touch the SAME padding take a masked path: out-of-image taps become out-of-range offsets);
//   D: lane holds couts 4 (lane >> 4) .. + 3 of pixel lane & 15: one 16-byte store.
constexpr unsigned C3M_OOB = 0x7FFF0000u;
__global__ __launch_bounds__(256) void conv3x3_cin3_mfma_kernel(const DirectArgs a, int groups_per_row, int nrows, int rstep, int N) {
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fq = lane >> 4;
    float wa[7];
    int kdy[7], kdx[7], kci[7];
    unsigned koff[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = 4 * i + fq;
        const bool kv = k < 27;
        const int tap = kv ? k / 3 : 0, ci = kv ? k - 3 * tap : 0;
        const int ty = tap / 3, tx = tap - 3 * ty;
        wa[i] = kv ? a.w[k * 16 + fr] : 0.f;
        kdy[i] = kv ? ty * a.dil : -(1 << 20);           // k = 27: never inside the image
        kdx[i] = tx * a.dil + fr * a.stride;
        kci[i] = ci;
        koff[i] = kv ? (unsigned)(((kdy[i] * a.W + kdx[i]) * a.x_cs + ci) * 4) : C3M_OOB;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((long)N * a.H * a.W * a.x_cs * 4), 0x00020000);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * fq);
    // wave w owns column group w % groups_per_row and walks the rows (n, oy) from w / groups_per_row in steps of `rstep`
    // (no division in the loop)
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wave >= rstep * groups_per_row) return;
    const int gx = wave % groups_per_row;
    int row = wave / groups_per_row;
    int n = row / a.Ho, oy = row - n * a.Ho;
    const int ox0 = gx * 16;
    for (; row < nrows; row += rstep, oy += rstep) {
        while (oy >= a.Ho) { oy -= a.Ho; ++n; }
        const int iy0 = oy * a.stride - a.pad_t, ix0 = ox0 * a.stride - a.pad_l;
        const bool interior = iy0 >= 0 && iy0 + 2 * a.dil < a.H && ix0 >= 0 && ix0 + 15 * a.stride + 2 * a.dil < a.W;
        const long base = (((long)n * a.H + iy0) * a.W + ix0) * a.x_cs * 4;       // bytes; negative only on the masked path
        float b[7];
        if (interior) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
                b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)koff[i], (int)base, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int iy = iy0 + kdy[i], ix = ix0 + kdx[i];
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const long off = (((long)n * a.H + iy) * a.W + ix) * a.x_cs * 4 + kci[i] * 4;
                b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ok ? (int)off : (int)C3M_OOB, 0, 0));
            }
        }
        f32x4 acc = b4;
#pragma unroll
        for (int i = 0; i < 7; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], b[i], acc, 0, 0, 0);
        if (a.apply_act) {
            acc[0] = pwc_lrelu(acc[0], a.slope); acc[1] = pwc_lrelu(acc[1], a.slope);
            acc[2] = pwc_lrelu(acc[2], a.slope); acc[3] = pwc_lrelu(acc[3], a.slope);
        }
        if (ox0 + fr < a.Wo)
            *reinterpret_cast<f32x4*>(a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox0 + fr) * a.y_cs + 4 * fq) = acc;
    }
}


// ---------------------------------------------------------------- 32 -> 2 flow head as a 1x1 GEMM + 9 shifted adds
// y[p, co] = sum_t sum_ci x[p + t, ci] w[t, ci, co]  =  sum_t Z[p + t][t, co],   Z[q][t, co] = sum_ci x[q, ci] w[t, ci, co]:
// the inner sums are ONE small GEMM per input pixel (18 = 9 taps x 2 couts outputs, K = 32) on the matrix pipe -- 16 MFMAs
// per 16 pixels instead of 576 FMAs per pixel and 216 ds_read_b128 per thread in the tiled kernel above -- and the outer
// sum is 9 ds_read_b64 + 18 adds per output pixel from a Z tile in LDS (8 x 32 outputs need Z on 10 x 34 pixels; out-of-image
// pixels load as zeros through the buffer range check, so their Z is the zero of the SAME padding).
//   A (weights): rows m = 2 t + co (18 of 32), lane (m & 15, kq = lane >> 4) holds channels 8 kq + i, i = 0..7, of both M tiles;
//   B (pixels):  lane (kq, pixel = lane & 15) loads channels 8 kq .. 8 kq + 7 of its pixel (two 16-byte loads).
constexpr int HM_R = 8, HM_C = 32, HM_PW = HM_C + 2, HM_NP = (HM_R + 2) * HM_PW, HM_NG = (HM_NP + 15) / 16, HM_ZS = 20;
__global__ __launch_bounds__(256) void conv3x3_head2_mfma_kernel(const DirectArgs a, int tiles_x, int tiles_y, int N) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float zs[HM_NG * 16 * HM_ZS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int y0 = by * HM_R, x0 = bx * HM_C;
    float wa[2][8];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = T * 16 + fr;                       // = 2 tap + co
            wa[T][i] = m < 18 ? a.w[((m >> 1) * 32 + 8 * kq + i) * 2 + (m & 1)] : 0.f;
        }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
    // Round 6: ALL of the wave's pixel groups are requested before the first one is used (12 16-byte loads per lane in flight).  The
    // loop this replaces asked for one group, waited, multiplied, stored, and asked for the next: six memory round trips in
    // series per workgroup -- the "15 us even on warm operands" of rounds 3-5 was that chain, not the launch.
    constexpr int GPW = (HM_NG + 3) / 4;                     // groups per wave
    f32x4 lo[GPW], hi[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int j = (wave + 4 * gi) * 16 + fr;
        const int py = j / HM_PW, px = j - py * HM_PW;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool ok = wave + 4 * gi < HM_NG && j < HM_NP && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const int vo = ok ? ((yy * a.W + xx) * a.x_cs + 8 * kq) * 4 : (int)C3M_OOB;
        lo[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vo, 0, 0));
        hi[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vo, 16, 0));
    }
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        if (wave + 4 * gi >= HM_NG) break;                   // uniform
        const int j = (wave + 4 * gi) * 16 + fr;
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = z0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xv = i < 4 ? lo[gi][i & 3] : hi[gi][i & 3];
            z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][i], xv, z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][i], xv, z1, 0, 0, 0);
        }
        // D: rows 4 kq + r of the M tile, column = pixel fr
        *reinterpret_cast<f32x4*>(zs + j * HM_ZS + 4 * kq) = z0;
        if (kq == 0) *reinterpret_cast<f32x2*>(zs + j * HM_ZS + 16) = f32x2{z1[0], z1[1]};
    }
    __syncthreads();
    const int r = t >> 5, c = t & 31;
    f32x2 acc = {a.bias[0], a.bias[1]};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
            acc += *reinterpret_cast<const f32x2*>(zs + ((r + ty) * HM_PW + c + tx) * HM_ZS + (ty * 3 + tx) * 2);
    const int oy = y0 + r, ox = x0 + c;
    if (oy < a.H && ox < a.W) {
        const size_t m = ((size_t)n * a.H + oy) * a.W + ox;
        float v0 = acc[0], v1 = acc[1];
        if (a.apply_act) { v0 = pwc_lrelu(v0, a.slope); v1 = pwc_lrelu(v1, a.slope); }
        if (a.res) { v0 += a.res[m * a.res_cs]; v1 += a.res[m * a.res_cs + 1]; }
        a.y[m * a.y_cs] = v0;
        a.y[m * a.y_cs + 1] = v1;
    }
}


// ---------------------------------------------------------------- the same for WIDE inputs (round 5): the dense-connection heads
// modules.py:274 with use_dc=True reads the whole estimator buffer (736 ... 3200 physical channels: every level's buffer holds the
// up-sampled buffer of the level below).  The 1x1-GEMM form above with the channels in chunks of 32: a wave keeps the Z
// accumulators of its (up to six) 16-pixel groups of the 10 x 34 patch in registers and walks the chunks -- weights of a chunk into
// registers once, 8 channels of every group pixel per lane, 16 matrix instructions per group and chunk.  These launches are
// bandwidth: 2.9 GB of input at the 112 x 256 level of configs[3] -- 1094 us = 2.7 TB/s here, 1455 us on the LDS-tiled FMA kernel it
// replaces (conv3x3_head2_wide_kernel; 115 / 199 / 284 / 490 us at the levels below, 51 / 103 / 156 / 244 here).  (A group-major
// walk with the weights in the LDS -- whole pixel records front to back -- was slower: 1900 us, one workgroup per CU.)
__global__ __launch_bounds__(256) void conv3x3_head2_widemfma_kernel(const DirectArgs a, int tiles_x, int tiles_y, int N) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int GPW = (HM_NG + 3) / 4;                     // groups per wave
    __shared__ __attribute__((aligned(16))) float zs[HM_NG * 16 * HM_ZS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int y0 = by * HM_R, x0 = bx * HM_C;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 9 * a.Cin * 2 * 4, 0x00020000);
    // this lane's pixel of each of its groups: byte offset of channel 8 kq (out-of-image / beyond the patch: out of range)
    int pvo[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int j = (wave + 4 * gi) * 16 + fr;
        const int py = j / HM_PW, px = j - py * HM_PW;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        const bool ok = wave + 4 * gi < HM_NG && j < HM_NP && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        pvo[gi] = ok ? ((yy * a.W + xx) * a.x_cs + 8 * kq) * 4 : -1;
    }
    f32x4 z0[GPW], z1[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) { z0[gi] = f32x4{0.f, 0.f, 0.f, 0.f}; z1[gi] = z0[gi]; }
    for (int c0 = 0; c0 < a.Cin; c0 += 32) {
        const bool cok = c0 + 8 * kq < a.Cin;                 // (a last chunk of 16 channels: lanes kq >= 2 carry zeros)
        // weights of the chunk: row m = 2 tap + co of M tile T, channels c0 + 8 kq + i -- HWIO (tap, channel, co): the 8 channels
        // of one tap are 16 consecutive floats (both co)
        float wa[2][8];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const int m = T * 16 + fr;
            const bool wok = m < 18 && cok;
            const unsigned wo = wok ? (unsigned)((((m >> 1) * a.Cin + c0 + 8 * kq) * 2) * 4) : 0x80000000u;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 w4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wo, q4 * 16, 0));
                wa[T][2 * q4] = (m & 1) ? w4[1] : w4[0];
                wa[T][2 * q4 + 1] = (m & 1) ? w4[3] : w4[2];
            }
        }
        f32x4 lo[GPW], hi[GPW];
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const int vo = (pvo[gi] >= 0 && cok) ? pvo[gi] + c0 * 4 : (int)C3M_OOB;
            lo[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vo, 0, 0));
            hi[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vo, 16, 0));
        }
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xv = i < 4 ? lo[gi][i & 3] : hi[gi][i & 3];
                z0[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][i], xv, z0[gi], 0, 0, 0);
                z1[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][i], xv, z1[gi], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int g = wave + 4 * gi;
        if (g < HM_NG) {
            const int j = g * 16 + fr;
            *reinterpret_cast<f32x4*>(zs + j * HM_ZS + 4 * kq) = z0[gi];
            if (kq == 0) *reinterpret_cast<f32x2*>(zs + j * HM_ZS + 16) = f32x2{z1[gi][0], z1[gi][1]};
        }
    }
    __syncthreads();
    const int r = t >> 5, c = t & 31;
    f32x2 acc = {a.bias[0], a.bias[1]};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
            acc += *reinterpret_cast<const f32x2*>(zs + ((r + ty) * HM_PW + c + tx) * HM_ZS + (ty * 3 + tx) * 2);
    const int oy = y0 + r, ox = x0 + c;
    if (oy < a.H && ox < a.W) {
        const size_t m = ((size_t)n * a.H + oy) * a.W + ox;
        float v0 = acc[0], v1 = acc[1];
        if (a.apply_act) { v0 = pwc_lrelu(v0, a.slope); v1 = pwc_lrelu(v1, a.slope); }
        if (a.res) { v0 += a.res[m * a.res_cs]; v1 += a.res[m * a.res_cs + 1]; }
        a.y[m * a.y_cs] = v0;
        a.y[m * a.y_cs + 1] = v1;
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
    if (Cin == 3 && Cout == 16 && !residual && (y_cs & 3) == 0 && pwc_aligned16(y) && pwc_aligned16(bias) &&
        (long)N * H * W * x_cs * 4 < (long)C3M_OOB) {
        const int gpr = (a.Wo + 15) / 16;
        const long nrows = (long)N * a.Ho;
        if (nrows < (1L << 30) && gpr <= 8192) {
            long rstep = 8192 / gpr;                     // ~8 waves per SIMD, each walks the rows of its column group
            if (rstep < 1) rstep = 1;
            if (rstep > nrows) rstep = nrows;
            const long blocks = (rstep * gpr + 3) / 4;
            hipLaunchKernelGGL(conv3x3_cin3_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, gpr, (int)nrows,
                               (int)rstep, N);
            return pwc_launch_status();
        }
    }
    if (Cin == 3 && Cout == 16 && !residual && (y_cs & 3) == 0 && pwc_aligned16(y) && pwc_aligned16(bias)) {
        long blocks = (a.M * (Cout >> 2) + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL((conv3x3_smallcin_kernel<3, 16>), dim3((unsigned)blocks), dim3(256),
                           (size_t)27 * Cout * sizeof(float), s, a);
        return pwc_launch_status();
    }
    if (Cout == 2 && Cin == 32 && vec4 && stride == 1 && dilation == 1 && (long)H * W >= 4096 &&
        (long)H * W * x_cs * 4 < (long)C3M_OOB) {
        const int tiles_x = (W + HM_C - 1) / HM_C, tiles_y = (H + HM_R - 1) / HM_R;
        const long nblk = (long)N * tiles_x * tiles_y;
        if (nblk < (1L << 31)) {
            hipLaunchKernelGGL(conv3x3_head2_mfma_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a, tiles_x, tiles_y, N);
            return pwc_launch_status();
        }
    }
    if (Cout == 2 && Cin >= 64 && Cin % 16 == 0 && vec4 && stride == 1 && dilation == 1 && pwc_aligned16(w_hwio) &&
        (long)H * W * x_cs * 4 < (long)C3M_OOB) {
        // wide heads (use_dc=True): the 1x1-GEMM form on the matrix pipe
        const int tiles_x = (W + HM_C - 1) / HM_C, tiles_y = (H + HM_R - 1) / HM_R;
        const long nblk = (long)N * tiles_x * tiles_y;
        if (nblk < (1L << 31)) {
            hipLaunchKernelGGL(conv3x3_head2_widemfma_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a, tiles_x, tiles_y, N);
            return pwc_launch_status();
        }
    }
    if (Cout == 2 && Cin >= 64 && vec4 && stride == 1 && dilation == 1) {
        const int tiles_x = (W + HT_C - 1) / HT_C, tiles_y = (H + HT_R - 1) / HT_R;
        const long nblk = (long)N * tiles_x * tiles_y;
        if (nblk < (1L << 31)) {
            hipLaunchKernelGGL(conv3x3_head2_wide_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a, tiles_x, tiles_y);
            return pwc_launch_status();
        }
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
