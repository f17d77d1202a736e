// conv3x3_mfma.hip -- 3x3 convolution (TF 'SAME', stride 1/2, dilation 1..16) as an
// fp32-MFMA implicit GEMM for gfx950 (MI355X).
//
// Replaces tf.layers.Conv2D(Cout,(3,3),(s,s),'same',dilation_rate=d) + tf.nn.leaky_relu
// of the reference (modules.py:62-67 extractor, :267-268 flow estimator, :306-323
// context network).
//
// GEMM view (per launch):  D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],
//   pixel = flattened (n, oy, ox) output position, k = (tap, physical input channel).
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, 32 cyc/issue, 157 TFLOP/s chip peak) with
//   A operand = weights   (row i = cout  = lane & 15, k-slot = lane >> 4)
//   B operand = activations (col j = pixel = lane & 15, k-slot = lane >> 4)
//   D: lane holds couts 4*(lane>>4)+{0..3} of pixel (lane & 15)  -> one float4 NHWC store.
// k is consumed in groups of 16: lane k-slot q reads ONE ds_read_b128 = channels
// 4q..4q+3 and feeds element s of it to MFMA step s, i.e. step s contracts
// k = {s, 4+s, 8+s, 12+s}; both operands use the same map, so the sum is complete.
//
// LDS tiles are [k-group][row][16 floats] (64-byte rows, no padding) with the 16-byte
// chunk index XOR-swizzled by {0,3,2,1}[(row>>2)&3]: conflict-free for the four
// 16-lane groups a ds_read_b128 is serviced in.  The weight image is pre-swizzled at
// pack time (pwc_conv3x3_pack_f32), so its LDS copy is a linear memcpy.
//
// Pipeline: register-staged prefetch of stage s+1 (global -> VGPR) is issued before the
// MFMAs of stage s; the VGPRs are written to the other LDS buffer after them; one
// barrier per stage.
#include "pwc_common.h"

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int H, W, Ho, Wo;
    int Cin_phys, Cout, Cout_pad;
    int stride, dil, pad_t, pad_l;
    int apply_act;
    float slope;
    int M;      // N*Ho*Wo output pixels
    int y_vec4; // y pointer/stride allow float4 stores
};

__device__ __forceinline__ int swz4(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <int WM, int WN, int WGM, int WGN, int KC>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs a) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int BM = 16 * WM * WGM;
    constexpr int BN = 16 * WN * WGN;
    constexpr int KG = KC / 16;       // 16-wide k-groups per stage
    constexpr int CPP = KC / 4;       // float4 chunks per pixel per stage
    constexpr int A_F4 = BM * CPP;
    constexpr int B_F4 = BN * CPP;
    constexpr int A_LD = (A_F4 + 255) / 256;
    constexpr int B_LD = (B_F4 + 255) / 256;
    constexpr int PIX_STEP = 256 / CPP;   // pixel rows covered by one pass of 256 threads
    constexpr int STAGE = (BM + BN) * KC; // floats per LDS buffer

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread A-load bookkeeping (fixed over the k loop)
    const int a_ch = t % CPP;          // 16-byte chunk within the pixel's KC channels
    const int a_row0 = t / CPP;
    const float* a_base[A_LD];
    int a_iy0[A_LD], a_ix0[A_LD], a_lds[A_LD];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int row = a_row0 + i * PIX_STEP;
        const int m = m0 + row;
        const bool ok = (row < BM) && (m < a.M);
        const int mm = ok ? m : 0;
        const int n_img = mm / HoWo;
        const int rem = mm - n_img * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        a_base[i] = a.x + (size_t)n_img * a.H * a.W * a.x_cs + a_ch * 4;
        // an invalid row gets an iy0 that fails every bounds test
        a_iy0[i] = ok ? oy * a.stride - a.pad_t : -(1 << 28);
        a_ix0[i] = ox * a.stride - a.pad_l;
        const int g = a_ch >> 2, j = a_ch & 3;
        a_lds[i] = ((g * BM + row) * 16) + ((j ^ swz4(row)) << 2);
    }
    // ---- per-thread B-load bookkeeping
    int b_src[B_LD], b_lds[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int q = t + i * 256;
        const int j = q & 3;
        const int rr = q >> 2;
        const int row = rr % BN, g = rr / BN;
        b_ok[i] = (q < B_F4) && (n0 + row < a.Cout_pad);
        b_src[i] = (g * a.Cout_pad + n0 + row) * 16 + j * 4;
        b_lds[i] = (g * BN + row) * 16 + j * 4;
    }

    const int CC = a.Cin_phys / KC; // stages per tap
    const int S = 9 * CC;
    const int nc16 = a.Cin_phys >> 4;

    f32x4 ra[A_LD], rb[B_LD];
    auto load_stage = [&](int tap, int cc) {
        const int ty = tap / 3, tx = tap - ty * 3;
        const int dy = ty * a.dil, dx = tx * a.dil;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(a_base[i] + (size_t)(iy * a.W + ix) * a.x_cs + cc * KC);
            ra[i] = v;
        }
        const float* wsrc = a.wp + (size_t)(tap * nc16 + cc * KG) * a.Cout_pad * 16;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(wsrc + b_src[i]);
            rb[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
        float* Ab = smem + buf * STAGE;
        float* Bb = Ab + BM * KC;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (A_F4 % 256 == 0 || t + i * 256 < A_F4) *reinterpret_cast<f32x4*>(Ab + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if (B_F4 % 256 == 0 || t + i * 256 < B_F4) *reinterpret_cast<f32x4*>(Bb + b_lds[i]) = rb[i];
    };

    f32x4 acc[WN][WM];
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int m = 0; m < WM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;            // fragment row (cout for W, pixel for X)
    const int fq = lane >> 4;            // k-slot
    const int f_off = fr * 16 + ((fq ^ swz4(fr)) << 2);

    int tap = 0, cc = 0;
    load_stage(0, 0);
    store_stage(0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        int ntap = tap, ncc = cc + 1;
        if (ncc == CC) { ncc = 0; ntap = tap + 1; }
        const bool more = (s + 1 < S);
        if (more) load_stage(ntap, ncc);

        const float* Ab = smem + buf * STAGE;
        const float* Bb = Ab + BM * KC;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            f32x4 wf[WN], xf[WM];
#pragma unroll
            for (int n = 0; n < WN; ++n)
                wf[n] = *reinterpret_cast<const f32x4*>(Bb + (g * BN + (wn * WN + n) * 16) * 16 + f_off);
#pragma unroll
            for (int m = 0; m < WM; ++m)
                xf[m] = *reinterpret_cast<const f32x4*>(Ab + (g * BM + (wm * WM + m) * 16) * 16 + f_off);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < WN; ++n)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[n][m], 0, 0, 0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
        tap = ntap;
        cc = ncc;
    }

    // ---- epilogue: bias + leaky-relu, NHWC float4 stores (4 consecutive couts per lane)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int co = n0 + (wn * WN + n) * 16 + fq * 4;
        if (co >= a.Cout) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int pix = m0 + (wm * WM + m) * 16 + fr;
            if (pix >= a.M) continue;
            f32x4 v = acc[n][m] + b4;
            if (a.apply_act) {
                v[0] = pwc_lrelu(v[0], a.slope);
                v[1] = pwc_lrelu(v[1], a.slope);
                v[2] = pwc_lrelu(v[2], a.slope);
                v[3] = pwc_lrelu(v[3], a.slope);
            }
            float* dst = a.y + (size_t)pix * a.y_cs + co;
            if (a.y_vec4) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
    }
}

// ---------------------------------------------------------------- weight packing
// packed[tap][c16][cout_pad][16]: element (j*4+e) of row `co` holds
// w_hwio[tap][cin_map[c16*16 + ((j ^ swz4(co))*4 + e)]][co]   (0 for padding)
// i.e. the 16-byte chunk index is pre-swizzled so the LDS image is a linear copy.
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map,
                                    int Cin, int Cin_phys, int Cout, int Cout_pad, float* __restrict__ packed) {
    const size_t total = (size_t)9 * Cin_phys * Cout_pad;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e16 = (int)(idx & 15);
        size_t r = idx >> 4;
        const int co = (int)(r % Cout_pad);
        r /= Cout_pad;
        const int c16 = (int)(r % (Cin_phys >> 4));
        const int tap = (int)(r / (Cin_phys >> 4));
        const int jpos = e16 >> 2, e = e16 & 3;
        const int j = jpos ^ swz4(co);
        const int cphys = c16 * 16 + j * 4 + e;
        int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        float v = 0.f;
        if (clog >= 0 && clog < Cin && co < Cout) v = w[((size_t)tap * Cin + clog) * Cout + co];
        packed[idx] = v;
    }
}

// ---------------------------------------------------------------- host side
typedef void (*conv_kernel_t)(const ConvArgs);

struct TileCfg {
    int BM, BN;
    conv_kernel_t k16, k32;
};

#define PWC_TILE(WM, WN, WGM, WGN)                                                        \
    { 16 * WM * WGM, 16 * WN * WGN, conv3x3_mfma_kernel<WM, WN, WGM, WGN, 16>,            \
      conv3x3_mfma_kernel<WM, WN, WGM, WGN, 32> }

static const TileCfg g_tiles[] = {
    PWC_TILE(4, 4, 2, 2), // 0: 128 x 128
    PWC_TILE(4, 3, 2, 2), // 1: 128 x 96
    PWC_TILE(4, 2, 2, 2), // 2: 128 x 64
    PWC_TILE(4, 2, 4, 1), // 3: 256 x 32
    PWC_TILE(4, 1, 4, 1), // 4: 256 x 16
    PWC_TILE(2, 4, 2, 2), // 5: 64 x 128
    PWC_TILE(2, 3, 2, 2), // 6: 64 x 96
    PWC_TILE(2, 2, 2, 2), // 7: 64 x 64
    PWC_TILE(2, 2, 4, 1), // 8: 128 x 32
    PWC_TILE(2, 1, 4, 1), // 9: 128 x 16
    PWC_TILE(1, 4, 2, 2), // 10: 32 x 128
    PWC_TILE(1, 3, 2, 2), // 11: 32 x 96
    PWC_TILE(1, 2, 2, 2), // 12: 32 x 64
    PWC_TILE(1, 2, 4, 1), // 13: 64 x 32
    PWC_TILE(1, 1, 4, 1), // 14: 64 x 16
};
static const int g_ntiles = (int)(sizeof(g_tiles) / sizeof(g_tiles[0]));

static int pick_tile(int M, int Cout_pad) {
    // widest BN that divides Cout_pad, then the largest BM that still yields >= 512
    // workgroups (2 per CU on 256 CUs); smallest BM if none does.
    static const int by_bn[5][3] = {
        /* BN=128 */ {0, 5, 10}, /* 96 */ {1, 6, 11}, /* 64 */ {2, 7, 12},
        /* 32 */ {3, 8, 13},     /* 16 */ {4, 9, 14}};
    int row;
    if (Cout_pad % 128 == 0) row = 0;
    else if (Cout_pad % 96 == 0) row = 1;
    else if (Cout_pad % 64 == 0) row = 2;
    else if (Cout_pad % 32 == 0) row = 3;
    else row = 4;
    for (int c = 0; c < 3; ++c) {
        const TileCfg& tc = g_tiles[by_bn[row][c]];
        const long wgs = (long)((M + tc.BM - 1) / tc.BM) * (Cout_pad / tc.BN);
        if (wgs >= 512 || c == 2) return by_bn[row][c];
    }
    return by_bn[row][2];
}

extern "C" int pwc_conv3x3_select_tile(int M, int Cout, int Cin_phys, int* bm, int* bn, int* kc) {
    if (M <= 0 || Cout <= 0 || Cin_phys <= 0) return PWC_EINVAL;
    const int tile = pick_tile(M, (Cout + 15) & ~15);
    if (bm) *bm = g_tiles[tile].BM;
    if (bn) *bn = g_tiles[tile].BN;
    if (kc) *kc = (Cin_phys % 32 == 0) ? 32 : 16;
    return tile;
}

extern "C" size_t pwc_conv3x3_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    const int Cout_pad = (Cout + 15) & ~15;
    return (size_t)9 * Cin_phys * Cout_pad;
}

extern "C" int pwc_conv3x3_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                    int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int Cout_pad = (Cout + 15) & ~15;
    const size_t total = (size_t)9 * Cin_phys * Cout_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, Cout_pad, packed);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_f32(const float* x, int x_cs, const float* packed, const float* bias, float* y,
                               int y_cs, int N, int H, int W, int Cin_phys, int Cout, int stride,
                               int dilation, int apply_act, float slope, int tile, pwc_stream_t stream) {
    if (!x || !packed || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0) return PWC_EINVAL;
    if (stride < 1 || stride > 2 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(packed) || !pwc_aligned16(bias)) return PWC_EALIGN;
    ConvArgs a;
    a.x = x; a.wp = packed; a.bias = bias; a.y = y;
    a.x_cs = x_cs; a.y_cs = y_cs;
    a.H = H; a.W = W;
    pwc_same_pad(H, stride, dilation, &a.Ho, &a.pad_t);
    pwc_same_pad(W, stride, dilation, &a.Wo, &a.pad_l);
    a.Cin_phys = Cin_phys; a.Cout = Cout; a.Cout_pad = (Cout + 15) & ~15;
    a.stride = stride; a.dil = dilation;
    a.apply_act = apply_act; a.slope = slope;
    const long M = (long)N * a.Ho * a.Wo;
    // within-image offsets and the pixel count are 32-bit in the kernel
    if (M >= (1L << 31) || (long)H * W * x_cs >= (1L << 31)) return PWC_ERANGE;
    a.M = (int)M;
    a.y_vec4 = ((y_cs & 3) == 0 && pwc_aligned16(y)) ? 1 : 0;
    if (tile < 0) tile = pick_tile(a.M, a.Cout_pad);
    if (tile >= g_ntiles) return PWC_EINVAL;
    const TileCfg& tc = g_tiles[tile];
    if (a.Cout_pad % tc.BN) return PWC_EINVAL;
    const int KC = (Cin_phys % 32 == 0) ? 32 : 16;
    conv_kernel_t k = KC == 32 ? tc.k32 : tc.k16;
    const size_t lds = (size_t)2 * (tc.BM + tc.BN) * KC * sizeof(float);
    dim3 grid((unsigned)((a.M + tc.BM - 1) / tc.BM), (unsigned)(a.Cout_pad / tc.BN));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, a);
    return pwc_launch_status();
}
