// conv3x3_wgrad.hip -- weight gradient of the 3x3 'SAME' convolution on the fp32 MFMA units of gfx950
// (training path; the reference gets it from tf.gradients of tf.layers.Conv2D,
// train.py:90 / modules.py:62-66,267,274,306-324):
//
//   dW[ty][tx][ci][co] = sum_{n,oy,ox} X[n, oy*s + ty*d - pt, ox*s + tx*d - pl, ci] * dY[n,oy,ox,co]
//
// Per tap a GEMM  D[ci][co] = sum over output pixels  X_tap[pixel][ci] * dY[pixel][co]  with the PIXELS as the
// reduction dimension.  v_mfma_f32_16x16x4_f32 takes 4 pixels per instruction; lane (m = lane % 16, k = lane / 16)
// loads VA consecutive input channels and VB consecutive output channels of pixel k of the group with ONE
// vector load each -- register i of the X vector is then the A operand of the MFMAs for input channels
// {VA*m + i}, register j of the dY vector the B operand for output channels {VB*n + j}: VA * VB MFMAs per two
// loads, a (16*VA) x (16*VB) tile of dW per wave, straight from L2 (no LDS staging: both operands are
// pixel-major, the MFMA wants them exactly so).  The pixels are split over the 4 waves of a workgroup and
// over `ksplit` workgroups; partial tiles are summed in a fixed order (deterministic) by the reduce kernel,
// which also maps physical input channels back to the TensorFlow variable's logical order (cin_map).
#include "pwc_common.h"

struct WgradArgs {
    const float* x;
    const float* dy;
    float* partial;          // [ksplit][9][Cin_phys][Cout]
    int x_cs, dy_cs;
    int N, H, W, Ho, Wo;
    int Cin_phys, Cout;
    int stride, dil, pt, pl;
    int ksplit, ci_tiles, co_tiles;
    long npix, chunk;        // output pixels in total / per k-split workgroup (multiple of 16)
};

template <int V> struct WgVec;
template <> struct WgVec<1> { typedef float T; };
template <> struct WgVec<2> { typedef f32x2 T; };
template <> struct WgVec<4> { typedef f32x4 T; };

template <int V>
__device__ __forceinline__ void wg_load(const float* p, bool ok, float (&r)[V]) {
    if (V == 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = v[i];
    } else if (V == 2) {
        f32x2 v = {0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x2*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = v[i];
    } else {
        r[0] = ok ? *p : 0.f;
    }
}

template <int VA, int VB>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradArgs a) {
    __shared__ float red[3 * 64 * VA * VB * 4];           // tiles of waves 1-3 for the final sum
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, k = lane >> 4;
    int blk = blockIdx.x;
    const int cot = blk % a.co_tiles; blk /= a.co_tiles;
    const int cit = blk % a.ci_tiles; blk /= a.ci_tiles;
    const int tap = blk % 9;
    const int ks = blk / 9;
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int ci = cit * 16 * VA + m * VA, co = cot * 16 * VB + m * VB;
    const bool ci_ok = ci < a.Cin_phys, co_ok = co < a.Cout;     // (channel counts are multiples of VA / VB)
    const long p_begin = ks * a.chunk, p_end = min(a.npix, p_begin + a.chunk);

    f32x4 acc[VA][VB];
#pragma unroll
    for (int i = 0; i < VA; ++i)
#pragma unroll
        for (int j = 0; j < VB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int hw = a.Ho * a.Wo;
    for (long p0 = p_begin + 4 * wave; p0 < p_end; p0 += 16) {
        const long p = p0 + k;
        const bool pv = p < p_end;
        const int n = (int)(p / hw);
        const int rem = (int)(p - (long)n * hw);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        const int iy = oy * a.stride + ty * a.dil - a.pt, ix = ox * a.stride + tx * a.dil - a.pl;
        const bool in = pv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        float av[VA], bv[VB];
        wg_load<VA>(a.x + (((long)n * a.H + iy) * a.W + ix) * a.x_cs + ci, in && ci_ok, av);
        wg_load<VB>(a.dy + p * a.dy_cs + co, pv && co_ok, bv);
#pragma unroll
        for (int i = 0; i < VA; ++i)
#pragma unroll
            for (int j = 0; j < VB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }

    // ---- sum the 4 waves' tiles (fixed order), write the partial tile
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < VA; ++i)
#pragma unroll
            for (int j = 0; j < VB; ++j)
                *reinterpret_cast<f32x4*>(red + (((wave - 1) * VA * VB + i * VB + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + ((long)ks * 9 + tap) * a.Cin_phys * a.Cout;
#pragma unroll
        for (int i = 0; i < VA; ++i)
#pragma unroll
            for (int j = 0; j < VB; ++j) {
                f32x4 s = acc[i][j];
#pragma unroll
                for (int w = 0; w < 3; ++w) s += *reinterpret_cast<const f32x4*>(red + ((w * VA * VB + i * VB + j) * 64 + lane) * 4);
                // D rows m' = 4*k + r (input channel VA*m' + i), column n = m (output channel VB*m + j)
                const int oc = cot * 16 * VB + m * VB + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ic = cit * 16 * VA + (4 * k + r) * VA + i;
                    if (ic < a.Cin_phys && oc < a.Cout) out[(long)ic * a.Cout + oc] = s[r];
                }
            }
    }
}

// dW[tap][ci_log][co] = sum_ks partial[ks][tap][ci_phys][co]   (fixed order), dW zero-filled first by the caller's pass below
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ cin_map,
                                                                  int ksplit, int Cin_phys, int Cin, int Cout, float* __restrict__ dw,
                                                                  int accumulate) {
    const long total = 9L * Cin_phys * Cout;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int co = (int)(e % Cout);
        long r = e / Cout;
        const int cp = (int)(r % Cin_phys);
        const int tap = (int)(r / Cin_phys);
        const int cl = cin_map ? cin_map[cp] : (cp < Cin ? cp : -1);
        if (cl < 0 || cl >= Cin) continue;
        float s = 0.f;
        for (int ks = 0; ks < ksplit; ++ks) s += partial[(long)ks * total + e];
        float* d = dw + ((long)tap * Cin + cl) * Cout + co;
        *d = accumulate ? *d + s : s;
    }
}

// channels per lane: 4 from 48 channels up, 2 from 32, else 1 (a 16-channel tile)
static int wg_vec(int c) { return (c % 4 == 0 && c >= 48) ? 4 : (c % 2 == 0 && c >= 32) ? 2 : 1; }

static void wgrad_plan(int N, int Ho, int Wo, int Cin_phys, int Cout, int* va, int* vb, int* ksplit, long* chunk) {
    *va = wg_vec(Cin_phys);
    *vb = wg_vec(Cout);
    const long npix = (long)N * Ho * Wo;
    const long tiles = (long)((Cin_phys + 16 * *va - 1) / (16 * *va)) * ((Cout + 16 * *vb - 1) / (16 * *vb)) * 9;
    long ks = (2048 + tiles - 1) / tiles;                  // ~2048 workgroups
    const long max_ks = (npix + 255) / 256;                 // at least 256 pixels per workgroup
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    long ch = ((npix + ks - 1) / ks + 15) / 16 * 16;
    *ksplit = (int)((npix + ch - 1) / ch);
    *chunk = ch;
}

extern "C" size_t pwc_conv3x3_wgrad_workspace_floats(int N, int H, int W, int Cin_phys, int Cout, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || stride < 1) return 0;
    int va, vb, ks;
    long chunk;
    wgrad_plan(N, (H + stride - 1) / stride, (W + stride - 1) / stride, Cin_phys, Cout, &va, &vb, &ks, &chunk);
    return (size_t)ks * 9 * Cin_phys * Cout;
}

extern "C" int pwc_conv3x3_wgrad_f32(const float* x, int x_cs, const float* dy, int dy_cs, const int32_t* cin_map, int Cin,
                                     int Cin_phys, int Cout, float* dw_hwio, int accumulate, int N, int H, int W, int stride,
                                     int dilation, float* workspace, size_t workspace_floats, pwc_stream_t stream) {
    if (!x || !dy || !dw_hwio || !workspace) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin_phys < 1 || Cout <= 0 || stride < 1 || stride > 2 || dilation < 1) return PWC_EINVAL;
    if (x_cs < Cin_phys || dy_cs < Cout) return PWC_EINVAL;
    if (workspace_floats < pwc_conv3x3_wgrad_workspace_floats(N, H, W, Cin_phys, Cout, stride)) return PWC_EINVAL;
    WgradArgs a;
    a.x = x; a.dy = dy; a.partial = workspace; a.x_cs = x_cs; a.dy_cs = dy_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.stride = stride; a.dil = dilation;
    pwc_same_pad(H, stride, dilation, &a.Ho, &a.pt);
    pwc_same_pad(W, stride, dilation, &a.Wo, &a.pl);
    a.npix = (long)N * a.Ho * a.Wo;
    int va, vb;
    wgrad_plan(N, a.Ho, a.Wo, Cin_phys, Cout, &va, &vb, &a.ksplit, &a.chunk);
    // vector loads need aligned pointers / strides
    if (va > 1 && ((x_cs % va) || (reinterpret_cast<uintptr_t>(x) % (4 * va)) || (Cin_phys % va))) va = 1;
    if (vb > 1 && ((dy_cs % vb) || (reinterpret_cast<uintptr_t>(dy) % (4 * vb)) || (Cout % vb))) vb = 1;
    a.ci_tiles = (Cin_phys + 16 * va - 1) / (16 * va);
    a.co_tiles = (Cout + 16 * vb - 1) / (16 * vb);
    const long nblk = (long)a.ksplit * 9 * a.ci_tiles * a.co_tiles;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    // (a.ksplit was planned for the preferred vector widths; narrower vectors only mean more tiles)
    if ((size_t)a.ksplit * 9 * Cin_phys * Cout > workspace_floats) return PWC_EINVAL;
#define WG_LAUNCH(A, B) hipLaunchKernelGGL((conv3x3_wgrad_kernel<A, B>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a)
    if (va == 4 && vb == 4) WG_LAUNCH(4, 4);
    else if (va == 4 && vb == 2) WG_LAUNCH(4, 2);
    else if (va == 4 && vb == 1) WG_LAUNCH(4, 1);
    else if (va == 2 && vb == 4) WG_LAUNCH(2, 4);
    else if (va == 2 && vb == 2) WG_LAUNCH(2, 2);
    else if (va == 2 && vb == 1) WG_LAUNCH(2, 1);
    else if (va == 1 && vb == 4) WG_LAUNCH(1, 4);
    else if (va == 1 && vb == 2) WG_LAUNCH(1, 2);
    else WG_LAUNCH(1, 1);
#undef WG_LAUNCH
    long rb = (9L * Cin_phys * Cout + 255) / 256;
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, cin_map, a.ksplit, Cin_phys, Cin, Cout, dw_hwio, accumulate);
    return pwc_launch_status();
}
