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
#include <cstdlib>

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

    // Lane (m, k) walks pixels p_begin + 4*wave + k, + 16, ...: (n, oy, ox) are advanced incrementally (the divisions of
    // a per-iteration decomposition cost about as many VALU cycles as the 16 MFMAs of the widest tile), and the operands
    // of the next pixel group are requested before the MFMAs of the current one are issued.
    const int hw = a.Ho * a.Wo;
    long p = p_begin + 4 * wave + k;
    int n, oy, ox;
    {
        const long pc = min(p, a.npix - 1);
        n = (int)(pc / hw);
        const int rem = (int)(pc - (long)n * hw);
        oy = rem / a.Wo;
        ox = rem - oy * a.Wo;
    }
    const int ty_off = ty * a.dil - a.pt, tx_off = tx * a.dil - a.pl;
    float av[VA], bv[VB];
    auto fetch = [&](float (&fa)[VA], float (&fb)[VB]) {
        const bool pv = p < p_end;
        const int iy = oy * a.stride + ty_off, ix = ox * a.stride + tx_off;
        const bool in = pv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        wg_load<VA>(a.x + (((long)n * a.H + iy) * a.W + ix) * a.x_cs + ci, in && ci_ok, fa);
        wg_load<VB>(a.dy + p * a.dy_cs + co, pv && co_ok, fb);
        p += 16;
        ox += 16;
        while (ox >= a.Wo) {
            ox -= a.Wo;
            if (++oy == a.Ho) { oy = 0; ++n; }
        }
    };
    fetch(av, bv);
    for (long p0 = p_begin + 4 * wave; p0 < p_end; p0 += 16) {
        float an[VA], bn[VB];
        fetch(an, bn);                                    // (past the end: all-false predicates, zeros)
#pragma unroll
        for (int i = 0; i < VA; ++i)
#pragma unroll
            for (int j = 0; j < VB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VA; ++i) av[i] = an[i];
#pragma unroll
        for (int j = 0; j < VB; ++j) bv[j] = bn[j];
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

// ---------------------------------------------------------------- stride 1, LDS-staged (the large layers)
// The kernel above streams both operands from L2 once per (tap, 64x64 channel tile): a 128 -> 128 layer moves
// 18 KB per pixel through the vector memory path and is bound there (measured 17 TB/s, 43 % of the MFMA peak).
// conv3x3_wgrad_lds_kernel stages a tile of output pixels ONCE per channel tile for all 9 taps:
//
//   workgroup = 512 threads = 8 waves = NCI x NCO x NPX: a (32 NCI) ci x (32 NCO) co x 9 taps tile of dW, a chunk of
//               pixel tiles (k-split as above).  (NCI, NCO) is chosen per layer so that the channel counts fill the
//               tile: 128 -> 96 runs as (4, 1), 96 -> 64 as (1, 2), 32 -> 32 as (1, 1) with 8 pixel waves;
//   tile      = TH x TW = P = 4 G NPX output pixels of one dilation sub-lattice (y mod d, x mod d) of one image --
//               with dilation d the taps of a sub-lattice pixel are its sub-lattice neighbours, so every dilation is
//               the d = 1 problem on d*d sub-images -- TW in {64, ..., 8} by the sub-image width;
//   LDS       = the (TH+2) x (TW+2) input patch and the TH x TW dY tile as 32-channel planes [plane][pixel][32 ch]
//               (pixel stride 128 B: a wave's ds_read_b64 of 4 pixels x 16 channel pairs is 512 contiguous bytes),
//               two images: tile s is computed from one while the registers holding tile s+1 (buffer loads issued a
//               tile earlier; out-of-image pixels and padding channels are out-of-range offsets = zeros) are written
//               to the other -- one barrier per tile;
//   wave      = (ci block, co block, pixel part): a 32 x 32 tile x 9 taps = 36 accumulator tiles; per 4-pixel group one
//               dY read and 9 shifted X reads (ds_read_b64, fetched one group ahead) feed 36 MFMAs.  The pixel parts
//               are summed through LDS at the end (fixed order), partial tiles go to the same workspace / reduce
//               kernel as above.
struct WgLdsArgs {
    const float* x;
    const float* dy;
    float* partial;          // [ksplit][9][Cin_phys][Cout]
    int x_cs, dy_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int dil;
    int th, tw, tw_log;      // tile shape (th * tw == P)
    int tiles_y, tiles_x;    // tiles per sub-lattice image
    int ksplit, ci_tiles, co_tiles;
    int ntiles, chunk;       // tiles in total (N * d * d * tiles_y * tiles_x) / per workgroup
};

constexpr int wl_max_patch(int P) { return P >= 64 ? (P / 64 + 2) * 66 : 3 * (P + 2); }   // widest shape: TW = min(P, 64)

typedef float f32x1 __attribute__((ext_vector_type(1)));
template <int V> __device__ __forceinline__ float wl_elem(f32x1 v, int) { return v[0]; }
template <int V> __device__ __forceinline__ float wl_elem(f32x2 v, int i) { return v[i]; }

template <int NCI, int NCO, int G, int V = 2> struct WgLdsGeom {
    static constexpr int CPP = 16 * V, QPP = 4 * V;           // channels / 16-byte quads per plane (a wave tile is CPP wide)
    static constexpr int NPX = 8 / (NCI * NCO);               // pixel parts (waves)
    static constexpr int P = 4 * G * NPX;                     // output pixels per tile
    static constexpr int QX = QPP * NCI, QY = QPP * NCO;      // channel quads per pixel
    static constexpr int NX = (wl_max_patch(P) * QX + 511) / 512;   // b128 pieces per thread: patch
    static constexpr int ND = (P * QY + 511) / 512;                 //                         dY tile
    static constexpr int XPIX = NX * 512 / QX, DPIX = ND * 512 / QY;   // LDS pixels per plane (>= patch / tile pixels)
    static constexpr int XS = NCI * XPIX * CPP, DS = NCO * DPIX * CPP;   // floats
    static constexpr int IMG = XS + DS;
    static constexpr int RED = (NPX - 1) * NCI * NCO * 3 * V * V * 256;   // floats of the final pixel-part reduction
    static constexpr int LDS_BYTES = (2 * IMG > RED ? 2 * IMG : RED) * 4;
    static_assert(NCI * NCO * NPX == 8 && LDS_BYTES <= 160 * 1024, "wave split / LDS budget");
};

template <int NCI, int NCO, int G, int V>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_lds_kernel(const WgLdsArgs a) {
    typedef WgLdsGeom<NCI, NCO, G, V> Geo;
    constexpr int CPP = Geo::CPP, QPP = Geo::QPP;
    typedef float vec_t __attribute__((ext_vector_type(V)));
    constexpr int NPX = Geo::NPX, NX = Geo::NX, ND = Geo::ND, QX = Geo::QX, QY = Geo::QY;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, k = lane >> 4;
    const int wci = wave % NCI, wco = (wave / NCI) % NCO, wpx = wave / (NCI * NCO);
    // XCD-aware order (workgroup b runs on XCD b % 8): the channel tiles of one pixel chunk share an XCD's L2, so every
    // X / dY tile leaves HBM once.  The grid is rounded up to 8 * ceil(ksplit / 8) chunks; the surplus exits.
    const int npairs = a.ci_tiles * a.co_tiles;
    const int idx = blockIdx.x >> 3;
    const int ks = (blockIdx.x & 7) + 8 * (idx / npairs);
    if (ks >= a.ksplit) return;
    const int pair = idx % npairs;
    const int cot = pair % a.co_tiles, cit = pair / a.co_tiles;
    const int t_begin = ks * a.chunk, t_end = min(a.ntiles, t_begin + a.chunk);
    const int pw = a.tw + 2, npatch = (a.th + 2) * pw;
    const int d = a.dil;

    // ---- this thread's staging pieces (same for every tile).  Piece q = t + 512 * j covers channel quad q % QX of
    // patch pixel q / QX (j < NX; pixels past the patch are out-of-range loads = zeros into unused LDS pixels) or quad
    // q % QY of dY tile pixel q / QY (j < ND): no predicates anywhere, LDS slots advance by a constant per piece.
    const int xquad = t & (QX - 1), xpp0 = t / QX;
    const int dquad = t & (QY - 1), dpp0 = t / QY;
    int pc_py[NX], pc_px[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int pp = xpp0 + (512 / QX) * j;
        pc_py[j] = pp < npatch ? pp / pw : (1 << 20);         // far outside every image
        pc_px[j] = pp < npatch ? pp - pc_py[j] * pw : 0;
    }
    const int xslot = ((xquad / QPP) * Geo::XPIX + xpp0) * QPP + (xquad % QPP);      // 16-byte units; + (512 / QX) * QPP per piece
    const int dslot = Geo::XS / 4 + ((dquad / QPP) * Geo::DPIX + dpp0) * QPP + (dquad % QPP);
    constexpr unsigned WL_OOB = 0x7FFF0000u;
    f32x4 stx[NX], std_[ND];
    const bool xch_ok = cit * CPP * NCI + xquad * 4 < a.Cin_phys, dch_ok = cot * CPP * NCO + dquad * 4 < a.Cout;
    auto issue = [&](int tile) {
        int r = tile;
        const int bx = r % a.tiles_x; r /= a.tiles_x;
        const int by = r % a.tiles_y; r /= a.tiles_y;
        const int rx = r % d; r /= d;
        const int ry = r % d;
        const int n = r / d;
        const int sy0 = by * a.th, sx0 = bx * a.tw;          // tile origin in sub-lattice coordinates
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs + cit * CPP * NCI), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.dy + (size_t)n * a.H * a.W * a.dy_cs + cot * CPP * NCO), 0, a.H * a.W * a.dy_cs * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int iy = ry + d * (sy0 + pc_py[j] - 1), ix = rx + d * (sx0 + pc_px[j] - 1);
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && xch_ok;
            const unsigned off = ok ? (unsigned)(((iy * a.W + ix) * a.x_cs + xquad * 4) * 4) : WL_OOB;
            stx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int pix = dpp0 + (512 / QY) * j;
            const int oy = ry + d * (sy0 + (pix >> a.tw_log)), ox = rx + d * (sx0 + (pix & (a.tw - 1)));
            const bool ok = pix < Geo::P && oy < a.H && ox < a.W && dch_ok;
            const unsigned off = ok ? (unsigned)(((oy * a.W + ox) * a.dy_cs + dquad * 4) * 4) : WL_OOB;
            std_[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, (int)off, 0, 0));
        }
    };
    auto stash = [&](float* img) {
        f32x4* v = reinterpret_cast<f32x4*>(img);
#pragma unroll
        for (int j = 0; j < NX; ++j) v[xslot + (512 / QX) * QPP * j] = stx[j];
#pragma unroll
        for (int j = 0; j < ND; ++j) v[dslot + (512 / QY) * QPP * j] = std_[j];
    };

    f32x4 acc[9][V][V];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int xa_off = wci * Geo::XPIX * 16 + m;               // in V-float vectors; + pixel * 16
    const int db_off = Geo::XS / V + wco * Geo::DPIX * 16 + m;
    if (t_begin < t_end) {
        issue(t_begin);
        stash(smem);
        if (t_begin + 1 < t_end) issue(t_begin + 1);
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        const float* img = smem + ((tile - t_begin) & 1) * Geo::IMG;
        float* nxt = smem + (((tile - t_begin) & 1) ^ 1) * Geo::IMG;
        if (tile + 1 < t_end) {
            stash(nxt);
            if (tile + 2 < t_end) issue(tile + 2);
        }
        // operands of group g+1 are read from LDS before the MFMAs of group g are issued
        const vec_t* img2 = reinterpret_cast<const vec_t*>(img);
        vec_t bv, av[9];
        auto fetch = [&](int g, vec_t& fb, vec_t (&fa)[9]) {
            const int pix = (wpx * G + g) * 4;              // first tile pixel of the group (one tile row: tw % 4 == 0)
            const int r = pix >> a.tw_log, c0 = pix & (a.tw - 1);
            fb = img2[db_off + (pix + k) * 16];
            const vec_t* xg = img2 + xa_off + (r * pw + c0 + k) * 16;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) fa[ty * 3 + tx] = xg[(ty * pw + tx) * 16];
        };
        fetch(0, bv, av);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            vec_t bn, an[9];
            if (g + 1 < G) fetch(g + 1, bn, an);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int i = 0; i < V; ++i)
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        acc[tp][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl_elem<V>(av[tp], i), wl_elem<V>(bv, j), acc[tp][i][j], 0, 0, 0);
            if (g + 1 < G) {
                bv = bn;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) av[tp] = an[tp];
            }
        }
        __syncthreads();
    }

    // ---- sum the pixel parts through LDS (3 taps at a time: 12 tiles x 1 KB per wave), write the partial tile
    float* out = a.partial + (long)ks * 9 * a.Cin_phys * a.Cout;
    const int wch = wave % (NCI * NCO);
#pragma unroll
    for (int tr = 0; tr < 3; ++tr) {
        if (NPX > 1) {
            if (tr > 0) __syncthreads();
            if (wpx > 0) {
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int i = 0; i < V; ++i)
#pragma unroll
                        for (int j = 0; j < V; ++j)
                            reinterpret_cast<f32x4*>(smem)[((((wpx - 1) * NCI * NCO + wch) * 3 + tx) * V * V + i * V + j) * 64 + lane] =
                                acc[tr * 3 + tx][i][j];
            }
            __syncthreads();
        }
        if (wpx == 0) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int i = 0; i < V; ++i)
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        f32x4 s = acc[tr * 3 + tx][i][j];
#pragma unroll
                        for (int w = 1; w < NPX; ++w)
                            s += reinterpret_cast<const f32x4*>(smem)[((((w - 1) * NCI * NCO + wch) * 3 + tx) * V * V + i * V + j) * 64 + lane];
                        // D rows 4*k + r' (input channel V * row + i), column m (output channel V * m + j)
                        const int oc = (cot * NCO + wco) * CPP + V * m + j;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int ic = (cit * NCI + wci) * CPP + (4 * k + rr) * V + i;
                            if (ic < a.Cin_phys && oc < a.Cout)
                                out[((long)(tr * 3 + tx) * a.Cin_phys + ic) * a.Cout + oc] = s[rr];
                        }
                    }
        }
    }
}

// (NCI, NCO, G) configurations instantiated; the plan picks the one with the least padded channel area
struct WgLdsCfg { int nci, nco, g, v; };
static const WgLdsCfg WL_CFGS[] = {{2, 2, 8, 2}, {4, 1, 4, 2}, {2, 1, 4, 2}, {1, 2, 4, 2}, {1, 1, 4, 2}, {1, 1, 16, 1}};

// the LDS-staged kernel pays from 32 channels on either side and enough tiles to give every workgroup a few
static bool wgrad_lds_plan(int N, int H, int W, int Cin_phys, int Cout, int stride, int dil, WgLdsArgs* a, int* cfg_out) {
    if (stride != 1 || Cin_phys < 16 || Cout < 16 || (Cin_phys & 3) || (Cout & 3)) return false;
    int best = -1;
    long best_area = 0;
    const bool narrow = Cin_phys <= 16 && Cout <= 16;                  // 16-wide wave tiles only for the 16 -> 16 layers
    if (!narrow && (Cin_phys < 32 || Cout < 32)) return false;
    for (int c = 0; c < (int)(sizeof(WL_CFGS) / sizeof(WL_CFGS[0])); ++c) {
        if ((WL_CFGS[c].v == 1) != narrow) continue;
        const int tci = 16 * WL_CFGS[c].v * WL_CFGS[c].nci, tco = 16 * WL_CFGS[c].v * WL_CFGS[c].nco;
        const long area = (long)((Cin_phys + tci - 1) / tci) * tci * ((Cout + tco - 1) / tco) * tco;
        if (best < 0 || area < best_area) { best = c; best_area = area; }       // earlier configurations win ties
    }
    const WgLdsCfg cf = WL_CFGS[best];
    const int P = 4 * cf.g * (8 / (cf.nci * cf.nco));
    const int hs = (H + dil - 1) / dil, ws = (W + dil - 1) / dil;     // sub-lattice image (largest)
    int tw = P < 64 ? P : 64;
    while (tw > 8 && tw / 2 >= ws) tw >>= 1;                           // smallest TW that covers the width
    int tw_log = 0;
    while ((1 << tw_log) < tw) ++tw_log;
    const int th = P / tw;
    const int tiles_x = (ws + tw - 1) / tw, tiles_y = (hs + th - 1) / th;
    const long ntiles = (long)N * dil * dil * tiles_y * tiles_x;
    const int tci = 16 * cf.v * cf.nci, tco = 16 * cf.v * cf.nco;
    const int ci_tiles = (Cin_phys + tci - 1) / tci, co_tiles = (Cout + tco - 1) / tco;
    // one workgroup per CU, and the channel tiles of a chunk on ONE XCD (32 CUs): chunks per XCD = 32 / tile pairs
    long ks = 8 * (32 / (ci_tiles * co_tiles));
    if (ks < 8) ks = 8;                                                 // many channel tiles: one chunk per XCD, several rounds
    if (ks > ntiles / 3) ks = ntiles / 3 / 8 * 8;                       // at least 3 tiles per workgroup
    if (ks < 8 || ks * ci_tiles * co_tiles < 128 || ntiles >= (1L << 30)) return false;
    if ((long)H * W * (Cin_phys > Cout ? Cin_phys : Cout) >= (1L << 28)) return false;   // 32-bit byte offsets per image
    const int chunk = (int)((ntiles + ks - 1) / ks);
    if (a) {
        a->th = th; a->tw = tw; a->tw_log = tw_log; a->tiles_x = tiles_x; a->tiles_y = tiles_y;
        a->ci_tiles = ci_tiles; a->co_tiles = co_tiles; a->ntiles = (int)ntiles; a->chunk = chunk;
        a->ksplit = (int)((ntiles + chunk - 1) / chunk);
    }
    if (cfg_out) *cfg_out = best;
    return true;
}

template <int NCI, int NCO, int G, int V>
static void wgrad_lds_launch(const WgLdsArgs& l, hipStream_t stream) {
    typedef WgLdsGeom<NCI, NCO, G, V> Geo;
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_lds_kernel<NCI, NCO, G, V>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS_BYTES);
    }
    hipLaunchKernelGGL((conv3x3_wgrad_lds_kernel<NCI, NCO, G, V>), dim3((unsigned)((l.ksplit + 7) / 8 * 8 * l.ci_tiles * l.co_tiles)), dim3(512),
                       Geo::LDS_BYTES, stream, l);
}

// dW[tap][ci_log][co] (+)= sum_ks partial[ks][tap][ci_phys][co] in a fixed order: block = 64 elements x 4 k-lanes, lane q
// adds the partials ks = q, q + 4, ... (two running sums), the lanes are combined through LDS as ((0 + 1) + (2 + 3)).
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ cin_map,
                                                                  int ksplit, int Cin_phys, int Cin, int Cout, float* __restrict__ dw,
                                                                  int accumulate) {
    __shared__ float red[256];
    const long total = 9L * Cin_phys * Cout;
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long e = blockIdx.x * 64L + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int ks = q;
        for (; ks + 4 < ksplit; ks += 8) {
            s0 += partial[(long)ks * total + e];
            s1 += partial[(long)(ks + 4) * total + e];
        }
        if (ks < ksplit) s0 += partial[(long)ks * total + e];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (q == 0 && e < total) {
        const int co = (int)(e % Cout);
        const long r = e / Cout;
        const int cp = (int)(r % Cin_phys);
        const int tap = (int)(r / Cin_phys);
        const int cl = cin_map ? cin_map[cp] : (cp < Cin ? cp : -1);
        if (cl >= 0 && cl < Cin) {
            const float s = (red[el] + red[64 + el]) + (red[128 + el] + red[192 + el]);
            float* d = dw + ((long)tap * Cin + cl) * Cout + co;
            *d = accumulate ? *d + s : s;
        }
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
    // the LDS-staged plan's k-split depends on the dilation (through the tile count), which this query does not take:
    // bound it by its largest value for these channel counts (the smallest channel tiles give the most chunks)
    if (stride == 1 && Cin_phys >= 16 && Cout >= 16) {
        long bound = 8;
        for (int c = 0; c < (int)(sizeof(WL_CFGS) / sizeof(WL_CFGS[0])); ++c) {
            const int tci = 16 * WL_CFGS[c].v * WL_CFGS[c].nci, tco = 16 * WL_CFGS[c].v * WL_CFGS[c].nco;
            const long pairs = (long)((Cin_phys + tci - 1) / tci) * ((Cout + tco - 1) / tco);
            const long k = 8 * (32 / pairs) < 8 ? 8 : 8 * (32 / pairs);
            if (k > bound) bound = k;
        }
        if (ks < bound) ks = (int)bound;
    }
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
    {
        WgLdsArgs l;
        int cfg = 0;
        const bool al = !(reinterpret_cast<uintptr_t>(x) & 15) && !(reinterpret_cast<uintptr_t>(dy) & 15) && !(x_cs & 3) && !(dy_cs & 3);
        if (al && (long)H * W * (x_cs > dy_cs ? x_cs : dy_cs) < (1L << 29) &&
            wgrad_lds_plan(N, H, W, Cin_phys, Cout, stride, dilation, &l, &cfg) &&
            (size_t)l.ksplit * 9 * Cin_phys * Cout <= workspace_floats) {
            l.x = x; l.dy = dy; l.partial = workspace; l.x_cs = x_cs; l.dy_cs = dy_cs;
            l.N = N; l.H = H; l.W = W; l.Cin_phys = Cin_phys; l.Cout = Cout; l.dil = dilation;
            switch (cfg) {
                case 0: wgrad_lds_launch<2, 2, 8, 2>(l, (hipStream_t)stream); break;
                case 1: wgrad_lds_launch<4, 1, 4, 2>(l, (hipStream_t)stream); break;
                case 2: wgrad_lds_launch<2, 1, 4, 2>(l, (hipStream_t)stream); break;
                case 3: wgrad_lds_launch<1, 2, 4, 2>(l, (hipStream_t)stream); break;
                case 4: wgrad_lds_launch<1, 1, 4, 2>(l, (hipStream_t)stream); break;
                default: wgrad_lds_launch<1, 1, 16, 1>(l, (hipStream_t)stream); break;
            }
            const long rb = (9L * Cin_phys * Cout + 63) / 64;
            hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream,
                               (const float*)workspace, cin_map, l.ksplit, Cin_phys, Cin, Cout, dw_hwio, accumulate);
            return pwc_launch_status();
        }
    }
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
    const long rb = (9L * Cin_phys * Cout + 63) / 64;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, cin_map, a.ksplit, Cin_phys, Cin, Cout, dw_hwio, accumulate);
    return pwc_launch_status();
}
