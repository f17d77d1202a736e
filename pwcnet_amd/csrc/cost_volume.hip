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
// HBM-bound op (AI = 2*81*C / ((2C+81)*4) = 8.9 flop/B at C = 32).  Decomposition:
//   workgroup = 4 x 64 output pixels of one image; channels in chunks of 16.
//   LDS holds the f1w halo tile (4+2R) x 72 pixels PIXEL-major, one 80-byte row per pixel
//     (16 channels + 16 bytes of padding): 64 lanes reading the same channel quad of 64
//     consecutive pixels touch 16 distinct 16-byte slots per ds_read_b128 service group
//     (row*5 mod 16 is a bijection), and every read address is lane_base + a compile-time
//     offset -- no per-read address arithmetic.  The loader moves 16-byte channel quads
//     global -> VGPR -> ds_write_b128 (plain form) or gathers the 4 bilinear corners and
//     blends them in registers first (fused-warp form); pixels outside the image are 0.
//   Wave = (row pair, group of 3 vertical shifts); lane = output column.  A thread owns 2
//     vertically adjacent pixels x 3 shifts x D horizontal shifts = 54 accumulators; the 4
//     f1w rows it needs are shared by its two pixels (36 ds_read_b128 per 216 FMAs); its
//     two f0 pixels are read straight from global memory into registers.
//   The 256 x 81 result tile goes back through LDS (stride 81 floats: odd, conflict-free)
//     so that HBM stores are contiguous 324-byte pixel records, not 4-byte scatters.
//   Tiles are visited in XCD-aware order (neighbouring tiles share halo rows in one L2).
#include "pwc_common.h"
#include "cost_volume_roll.hip"   // rolling-window kernel for C = 32 (static: compiled into this unit)
#include "cost_volume_mfma.hip"   // matrix-pipe kernel with the warp and the concat copy fused in (static: compiled into this unit)
#include "cost_volume_h2.hip"     // round 5: the same launch on the F16 matrix pipe, gathers a step ahead (static: compiled into this unit)
#include "cost_volume_blk.hip"    // round 5: small pyramid levels, one 4 x 4 block per workgroup, F16 matrix pipe (static: compiled into this unit)

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
    int out_vec4;        // out pointer / stride allow 16-byte stores
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

constexpr int CV_RS = 20;   // LDS row stride in floats (16 channels + 4 padding)

template <int R>
struct CvGeom {
    static constexpr int D = 2 * R + 1, DD = D * D;
    static constexpr int NG = (D + 2) / 3;                 // groups of 3 vertical shifts
    static constexpr int NW = (CV_TH / 2) * NG;            // waves per workgroup
    static constexpr int T = 64 * NW;
    static constexpr int HH = CV_TH + 2 * R;               // halo rows
    static constexpr int NR1 = HH * CV_HW;                 // f1 pixels in LDS
    static constexpr int LDS_FLOATS = (NR1 * CV_RS > 128 * DD) ? NR1 * CV_RS : 128 * DD;
};

template <int R, bool FUSED>
__global__ __launch_bounds__(64 * CvGeom<R>::NW) void cost_volume_kernel(const CvArgs a) {
    using G = CvGeom<R>;
    constexpr int D = G::D, DD = G::DD, T = G::T;
    constexpr int NI = G::NR1 * 4;                         // loader items: (halo pixel, channel quad)
    constexpr int IPT = (NI + T - 1) / T;                  // items per thread per chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int yp = wave / G::NG;          // row pair: output rows 2*yp, 2*yp+1 of the tile
    const int vg = wave % G::NG;          // vertical shift indices 3*vg .. 3*vg+2

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

    // ---- loader bookkeeping (fixed over the channel loop): item q = t + T*i -> (pixel, quad)
    int ld_src[IPT];     // plain: element offset of the pixel in f1 (+ quad*4), <0 = zero fill
    int ld_dst[IPT];     // LDS float offset, <0 = no item
    float ld_fx[FUSED ? IPT : 1], ld_fy[FUSED ? IPT : 1];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int q = t + T * i;
        const int cq = q & 3, row = q >> 2;
        const int py = row / CV_HW, pxx = row - py * CV_HW;
        const int y = y0 - R + py, x = x0 - 4 + pxx;
        const bool inside = (q < NI) && ((unsigned)y < (unsigned)a.H) && ((unsigned)x < (unsigned)a.W);
        ld_dst[i] = (q < NI) ? row * CV_RS + cq * 4 : -1;
        ld_src[i] = inside ? (y * a.W + x) * a.f1_cs + cq * 4 : -1;
        if (FUSED) {
            // for the fused form ld_src keeps (y*W + x) and the flow is read once per tile
            ld_src[i] = inside ? ((y * a.W + x) << 2) | cq : -1;
            float fx = 0.f, fy = 0.f;
            if (inside) {
                const float* fp = fln + (size_t)(y * a.W + x) * a.flow_cs;
                fx = pwc_mul_rounded(fp[0], a.flow_scale);   // rounded product (see warp_kernel)
                fy = pwc_mul_rounded(fp[1], a.flow_scale);
            }
            ld_fx[i] = fx;
            ld_fy[i] = fy;
        }
    }
    // the two f0 pixels of this thread (rows 2yp, 2yp+1, column lane)
    const int oy0 = y0 + 2 * yp, ox = x0 + lane;
    const bool f0ok0 = (oy0 < a.H) && (ox < a.W), f0ok1 = (oy0 + 1 < a.H) && (ox < a.W);
    const float* f0p0 = f0n + (size_t)(f0ok0 ? oy0 * a.W + ox : 0) * a.f0_cs;
    const float* f0p1 = f0n + (size_t)(f0ok1 ? (oy0 + 1) * a.W + ox : 0) * a.f0_cs;

    float acc[2][3][D];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int h = 0; h < D; ++h) acc[j][v][h] = 0.f;

    // lane base of the f1 reads: halo row (2yp + 3vg), column lane + 4 - R
    const float* rd = smem + ((2 * yp + 3 * vg) * CV_HW + lane + 4 - R) * CV_RS;

    for (int c0 = 0; c0 < a.C; c0 += CV_CK) {
        // ---------------- f0 channel chunk of the two own pixels -> registers
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            a0[cq] = f32x4{0.f, 0.f, 0.f, 0.f};
            a1[cq] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c0 + cq * 4 < a.C) {
                if (f0ok0) a0[cq] = *reinterpret_cast<const f32x4*>(f0p0 + c0 + cq * 4);
                if (f0ok1) a1[cq] = *reinterpret_cast<const f32x4*>(f0p1 + c0 + cq * 4);
            }
        }
        // ---------------- f1w halo chunk -> LDS (batches of LB items keep the loads in flight
        // within the register budget: the fused form holds 4 corner quads per item)
        {
            constexpr int LB = FUSED ? 1 : IPT;
#pragma unroll
            for (int i0 = 0; i0 < IPT; i0 += LB) {
                f32x4 v[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int i = i0 + u;
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (i < IPT && ld_src[i < IPT ? i : 0] >= 0) {
                        if (FUSED) {
                            const int cq = ld_src[i] & 3, pix = ld_src[i] >> 2;
                            const int y = pix / a.W, x = pix - y * a.W;
                            if (c0 + cq * 4 < a.C)
                                v[u] = cv_warp_gather(f1n, a.f1_cs, c0 + cq * 4, a.H, a.W, y, x, ld_fx[i], ld_fy[i]);
                        } else {
                            v[u] = *reinterpret_cast<const f32x4*>(f1n + ld_src[i] + c0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int i = i0 + u;
                    if (i < IPT && ld_dst[i < IPT ? i : 0] >= 0) *reinterpret_cast<f32x4*>(smem + ld_dst[i]) = v[u];
                }
                if (FUSED) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();

        // ---------------- correlate: 4 channel quads x (4 f1w rows x D columns)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int h = 0; h < D; ++h) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(rd + (r * CV_HW + h) * CV_RS + cq * 4);
                    if (r < 3) {
                        float s = acc[0][r][h];
                        s = fmaf(a0[cq][0], w[0], s); s = fmaf(a0[cq][1], w[1], s);
                        s = fmaf(a0[cq][2], w[2], s); s = fmaf(a0[cq][3], w[3], s);
                        acc[0][r][h] = s;
                    }
                    if (r >= 1) {
                        float s = acc[1][r - 1][h];
                        s = fmaf(a1[cq][0], w[0], s); s = fmaf(a1[cq][1], w[1], s);
                        s = fmaf(a1[cq][2], w[2], s); s = fmaf(a1[cq][3], w[3], s);
                        acc[1][r - 1][h] = s;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // keep one f1w row of reads in flight
            }
        }
        __syncthreads();
    }

    // ---------------- mean over C, leaky-relu, stage through LDS, contiguous stores
    const float inv_c = 1.0f / (float)a.C;   // reduce_mean: x * (1/C), within 1 ulp of x / C
    for (int hf = 0; hf < CV_TH / 2; ++hf) {
        if (yp == hf) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int vi = 3 * vg + v;
                    if (vi < D) {
                        float* st = smem + (j * CV_TW + lane) * DD + vi * D;
#pragma unroll
                        for (int h = 0; h < D; ++h) st[h] = pwc_lrelu(acc[j][v][h] * inv_c, a.slope);
                    }
                }
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

// ---------------------------------------------------------------- persistent LDS-DMA form
// The production kernel for the plain cost volume (the warp, when needed, runs first as
// its own HBM-streaming kernel).  Differences from cost_volume_kernel above:
//   * LDS rows are dense 64-byte pixel records, so both tiles (f1w halo, f0) are filled by
//     global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR staging, no ds_write;
//   * bank conflicts are avoided on the READ side by a per-lane chunk rotation: at step s
//     lane l reads channel quad (s + (l>>2)) & 3 of its pixel row -- within a 16-lane
//     ds_read_b128 service group the 4 lanes that share a quad sit on 4 different rows
//     mod 4, and the 4 quads differ, whatever the row shift K; every address is
//     lane_base[s] + K*64 bytes (an immediate);
//   * two LDS stages: the DMA of the next stage (next 16-channel chunk, or the first chunk
//     of the workgroup's NEXT tile) is issued before the current stage is computed;
//   * one workgroup per CU walks its tiles persistently, so the HBM stream never stops
//     at tile boundaries; the result tile is staged through the stage just consumed.
__device__ float cv_zero_page[4];

template <int R>
struct CvPGeom {
    static constexpr int D = 2 * R + 1, DD = D * D;
    static constexpr int NG = (D + 2) / 3;                 // groups of 3 vertical shifts
    static constexpr int NHS = (D >= 5) ? 2 : 1;           // horizontal shifts split over NHS waves
    static constexpr int DH = (D + NHS - 1) / NHS;         // horizontal shifts per wave
    static constexpr int NW = (CV_TH / 2) * NG * NHS;      // waves per workgroup (12 for R = 4)
    static constexpr int T = 64 * NW;
    static constexpr int HH = CV_TH + 2 * R;
    static constexpr int NR1 = HH * CV_HW;
    static constexpr int NR0 = CV_TH * CV_TW;
    static constexpr int NB1 = NR1 / 16, NB0 = NR0 / 16, NBT = NB1 + NB0;
    static constexpr int F0_BASE = NR1 * 16;
    static constexpr int BUF = (NR1 + NR0) * 16;          // floats per stage
    static_assert(NR1 % 16 == 0, "halo tile must be whole 16-row DMA blocks");
    static_assert(BUF >= 128 * ((DD + 3) & ~3), "output stage must fit in one LDS stage");
};

// ABL: ablation switch for scripts/exp_cv.hip only (0 in the library): 1 = no FMAs,
// 2 = no DMA, 4 = no result stores.
template <int R, int ABL = 0>
__global__ __launch_bounds__(64 * CvPGeom<R>::NW) void cost_volume_dma_kernel(const CvArgs a) {
    using G = CvPGeom<R>;
    constexpr int D = G::D, DD = G::DD, NW = G::NW, T = G::T, DH = G::DH;
    constexpr int BPW = (G::NBT + NW - 1) / NW;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hs = wave % G::NHS;                          // which half of the horizontal shifts
    const int yp = (wave / G::NHS) / G::NG, vg = (wave / G::NHS) % G::NG;
    const int h0 = hs * DH;
    const float* zero = cv_zero_page;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = tiles_per_img * a.N;
    const int nwg = gridDim.x;
    // XCD-aware start: workgroups of one XCD own neighbouring tiles
    const int first = pwc_xcd_remap(blockIdx.x, nwg);
    const int spt = (a.C + CV_CK - 1) / CV_CK;             // stages per tile
    const int cl = (lane & 3) * 4;                         // this lane's channel quad within a chunk

    // ---- DMA bookkeeping.  Block b = wave + NW*i covers LDS rows 16b..16b+15; this lane
    // fills row 16b + (lane>>2), chunk lane&3.  The (row -> tile pixel) map is tile
    // independent: packed as (py << 8) | px with bit 31 = f0 block.
    int blk_px[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int b = wave + NW * i;
        const int row = b * 16 + (lane >> 2);
        int v = -1;
        if (b < G::NB1) {
            const int py = row / CV_HW;
            v = (py << 8) | (row - py * CV_HW);
        } else if (b < G::NBT) {
            const int r0 = row - G::NR1;
            v = (int)0x40000000 | ((r0 >> 6) << 8) | (r0 & 63);
        }
        blk_px[i] = v;
    }
    // per-tile source element offsets of the tile whose stages are being ISSUED (-1: zeros)
    int src_off[BPW];
    const float* src_img0 = a.f0;
    const float* src_img1 = a.f1;
    auto tile_offsets = [&](int lt) {
        const int n = lt / tiles_per_img;
        const int rem = lt - n * tiles_per_img;
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
        const int x0 = tx_ * CV_TW, y0 = ty_ * CV_TH;
        src_img0 = a.f0 + (size_t)n * a.H * a.W * a.f0_cs;
        src_img1 = a.f1 + (size_t)n * a.H * a.W * a.f1_cs;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int v = blk_px[i];
            const bool is0 = (v & 0x40000000) != 0;
            const int py = (v >> 8) & 0xFFFF, px = v & 0xFF;
            const int y = is0 ? y0 + py : y0 - R + py;
            const int x = is0 ? x0 + px : x0 - 4 + px;
            const bool ok = (v >= 0) && ((unsigned)y < (unsigned)a.H) && ((unsigned)x < (unsigned)a.W);
            src_off[i] = ok ? (y * a.W + x) * (is0 ? a.f0_cs : a.f1_cs) + cl : -1;
        }
    };
    auto issue_stage = [&](int c0, int buf) {
        float* dst = smem + buf * G::BUF;
        const bool cok = (c0 + cl) < a.C;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int b = wave + NW * i;                   // wave-uniform
            if (b < G::NBT) {
                const float* base = (b < G::NB1) ? src_img1 : src_img0;
                const float* src = (src_off[i] >= 0 && cok) ? base + src_off[i] + c0 : zero;
                if (!(ABL & 2)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + b * 256), 16, 0, 0);
            }
        }
    };

    // per-lane read bases: chunk rotation cs(s) = (s + (lane >> 2)) & 3
    const int r_base = 2 * yp + 3 * vg;
    int rd1[4], rd0[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cs = (s + (lane >> 2)) & 3;
        rd1[s] = (r_base * CV_HW + lane + 4 - R + h0) * 16 + cs * 4;
        rd0[s] = G::F0_BASE + ((2 * yp) * CV_TW + lane) * 16 + cs * 4;
    }

    // issue side runs one stage ahead of the compute side
    int itile = first, ichunk = 0;
    int cur = 0;
    if (itile < ntiles) {
        tile_offsets(itile);
        issue_stage(0, 0);
        if (++ichunk == spt) { ichunk = 0; itile += nwg; if (itile < ntiles) tile_offsets(itile); }
    }
    for (int lt = first; lt < ntiles; lt += nwg) {
        const int n = lt / tiles_per_img;
        const int rem = lt - n * tiles_per_img;
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
        const int x0 = tx_ * CV_TW, y0 = ty_ * CV_TH;

        float acc[2][3][DH];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int h = 0; h < DH; ++h) acc[j][v][h] = 0.f;

        for (int ci = 0; ci < spt; ++ci) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's DMA has landed
            __syncthreads();                                  // ... for every wave; previous stage fully read
            if (itile < ntiles) {
                issue_stage(ichunk * CV_CK, cur ^ 1);
                if (++ichunk == spt) { ichunk = 0; itile += nwg; if (itile < ntiles) tile_offsets(itile); }
            }

            const float* sb = smem + cur * G::BUF;
            // 16 steps (4 channel quads x 4 f1w rows); the D reads of step k+1 are issued
            // before the FMAs of step k (explicit register double buffer), so LDS latency
            // is covered by 36..72 FMAs instead of being exposed once per row.
            f32x4 wb[2][DH];
            f32x4 fa[2][2];
            auto load_row = [&](int k, int slot) {
                const int s = k >> 2, r = k & 3;
                // partial last shift group (D % 3 != 0): rows past the halo feed unused sums
                const int rr = (D % 3 != 0 && r_base + r >= G::HH) ? G::HH - 1 - r_base : r;
#pragma unroll
                for (int h = 0; h < DH; ++h)   // (a column past D on the last split reads a valid halo pixel)
                    wb[slot][h] = *reinterpret_cast<const f32x4*>(sb + rd1[s] + (rr * CV_HW + h) * 16);
                if (r == 0) {
                    fa[s & 1][0] = *reinterpret_cast<const f32x4*>(sb + rd0[s]);
                    fa[s & 1][1] = *reinterpret_cast<const f32x4*>(sb + rd0[s] + CV_TW * 16);
                }
            };
            load_row(0, 0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int s = k >> 2, r = k & 3;
                if (k + 1 < 16) load_row(k + 1, (k + 1) & 1);
                const f32x4 a0 = fa[s & 1][0], a1 = fa[s & 1][1];
#pragma unroll
                for (int h = 0; h < DH; ++h) {
                    const f32x4 w = wb[k & 1][h];
                    if (ABL & 1) {
                        asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
                        continue;
                    }
                    if (r < 3) {
                        float q = acc[0][r][h];
                        q = fmaf(a0[0], w[0], q); q = fmaf(a0[1], w[1], q);
                        q = fmaf(a0[2], w[2], q); q = fmaf(a0[3], w[3], q);
                        acc[0][r][h] = q;
                    }
                    if (r >= 1) {
                        float q = acc[1][r - 1][h];
                        q = fmaf(a1[0], w[0], q); q = fmaf(a1[1], w[1], q);
                        q = fmaf(a1[2], w[2], q); q = fmaf(a1[3], w[3], q);
                        acc[1][r - 1][h] = q;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            cur ^= 1;
        }

        // ---- epilogue: mean over C, leaky-relu, through the stage just consumed (cur ^ 1).
        // Stage rows are padded to SROW floats (multiple of 4) so the copy-out moves 16-byte
        // quads: ds_read_b128 -> global_store_dwordx4 (a pixel record = DD floats, contiguous).
        constexpr int SROW = (DD + 3) & ~3;
        constexpr int QPP = (DD + 3) / 4;                     // quads per pixel (last may be partial)
        float* stg = smem + (cur ^ 1) * G::BUF;
        const float inv_c = 1.0f / (float)a.C;
        for (int hf = 0; hf < CV_TH / 2; ++hf) {
            __syncthreads();
            if (yp == hf) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        const int vi = 3 * vg + v;
                        if (vi < D) {
                            float* st = stg + (j * CV_TW + lane) * SROW + vi * D + h0;
#pragma unroll
                            for (int h = 0; h < DH; ++h)
                                if (h0 + h < D) st[h] = pwc_lrelu(acc[j][v][h] * inv_c, a.slope);
                        }
                    }
            }
            __syncthreads();
            for (int e = t; e < 128 * QPP; e += T) {
                const int px = e / QPP, q = e - px * QPP;
                const int y = y0 + 2 * hf + (px >> 6), x = x0 + (px & 63);
                if (y < a.H && x < a.W && !(ABL & 4)) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + px * SROW + q * 4);
                    float* dst = a.out + (((size_t)n * a.H + y) * a.W + x) * a.out_cs + q * 4;
                    if (a.out_vec4 && q * 4 + 3 < DD) {
                        *reinterpret_cast<f32x4*>(dst) = v4;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (q * 4 + u < DD) dst[u] = v4[u];
                    }
                }
            }
        }
    }
}

template <int R>
static int cv_launch_dma(const CvArgs& a, hipStream_t s) {
    using G = CvPGeom<R>;
    const size_t lds = (size_t)2 * G::BUF * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_dma_kernel<R, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.N;
    // one persistent workgroup per CU; smaller LDS footprints (R < 4) may co-reside
    const int per_cu = (int)((size_t)160 * 1024 / lds);
    long nwg = 256L * (per_cu < 1 ? 1 : per_cu);
    if (nwg > ntiles) nwg = ntiles;
    hipLaunchKernelGGL((cost_volume_dma_kernel<R, 0>), dim3((unsigned)nwg), dim3(G::T), lds, s, a);
    return pwc_launch_status();
}

template <int R, bool FUSED>
static int cv_launch(const CvArgs& a, hipStream_t s) {
    using G = CvGeom<R>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_kernel<R, FUSED>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const unsigned nb = (unsigned)(a.tiles_x * a.tiles_y * a.N);
    hipLaunchKernelGGL((cost_volume_kernel<R, FUSED>), dim3(nb), dim3(G::T), lds, s, a);
    return pwc_launch_status();
}

static int cv_dispatch(CvArgs& a, int R, bool fused, hipStream_t s) {
    a.tiles_x = (a.W + CV_TW - 1) / CV_TW;
    a.tiles_y = (a.H + CV_TH - 1) / CV_TH;
    a.out_vec4 = ((a.out_cs & 3) == 0 && pwc_aligned16(a.out)) ? 1 : 0;
    if ((long)a.tiles_x * a.tiles_y * a.N >= (1L << 31)) return PWC_ERANGE;
    // in-image element offsets are 32-bit in the kernel
    if ((long)a.H * a.W * a.f0_cs >= (1L << 31) || (long)a.H * a.W * a.f1_cs >= (1L << 31)) return PWC_ERANGE;
#define PWC_CV(RR)                                                   \
    case RR:                                                         \
        return fused ? cv_launch<RR, true>(a, s) : cv_launch_dma<RR>(a, s);
    switch (R) {
        PWC_CV(1)
        PWC_CV(2)
        PWC_CV(3)
        PWC_CV(4)
        default: return PWC_EUNSUPPORTED;
    }
#undef PWC_CV
}

// ---------------------------------------------------------------- coarse levels, fused
// cost_volume_coarse_kernel: warp (optional) + cost volume + the tf.concat copy of f0, for the
// coarse pyramid levels (7x16 ... 28x64 pixels per image).  There the persistent 4x64-tile kernel
// above is a latency-bound chain (a 7x16 image occupies 16 workgroups that each walk 12 channel
// stages: 40 us for 896 pixels), and the warp and copy launches cost ~6 us apiece for kilobytes.
//
//   workgroup  = (image n, 8x8-pixel tile, vertical shift v): 9x more workgroups than tiles, each
//                produces the 9 horizontal shifts of its v -- 36 contiguous bytes per pixel record;
//   loader     = per 32-channel stage every thread fetches one of the 8 x 16 patch pixels
//                (rows y0+v .. y0+v+7, columns x0-4 .. x0+11) for 4 channel quads -- through the
//                bilinear warp of modules.py:99-137 when a flow is given -- and one f0 tile pixel
//                for 2 quads, into registers; the values go to LDS after the previous stage is
//                computed (register-staged double buffer: the loads of stage s+1 fly during stage s);
//   compute    = wave w owns channel quads 2w, 2w+1 of the stage, lane = tile pixel: 1 + 9
//                ds_read_b128 per quad (patch row stride 24 pixels: the 64 lanes of one shift hit 64
//                distinct 16-byte slots in every ds_read_b128 lane group);
//   epilogue   = the 4 waves' partial sums are added through LDS, mean, leaky-relu, store.
struct CvCoarseArgs {
    const float* f0;
    const float* f1;
    const float* flow;      // null: f1 is used as is (pyramid level 0, model.py:105-106)
    float* out;
    float* f0_copy;         // null or the `features_0` slice of the estimator input (modules.py:264)
    int f0_cs, f1_cs, flow_cs, out_cs, f0_copy_cs;
    int N, H, W, C;
    float flow_scale, slope;
    int tiles_x, tiles_y;
};

constexpr int CC_PSTR = 24;                          // patch row stride in pixels (16 used)
constexpr int CC_QS = 8;                             // channel quads per stage
constexpr int CC_F1 = CC_QS * 8 * CC_PSTR * 4;       // floats of the f1 patch image of a stage
constexpr int CC_STAGE = CC_F1 + CC_QS * 64 * 4;     // + f0 tile: 8192 floats = 32 KB
// LDS images of a stage: 1 (the next stage waits in registers until the current one is read; 3
// workgroups per CU hide the latencies) measured faster than 2 (2 workgroups per CU)
#ifndef CC_NBUF
#define CC_NBUF 1
#endif

template <bool WARP>
__global__ __launch_bounds__(256, 3) void cost_volume_coarse_kernel(const CvCoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // CC_NBUF stage images
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int blk = blockIdx.x;
    const int dyi = blk % 9;                         // vertical shift v = dyi - 4
    blk /= 9;
    const int bx = blk % a.tiles_x;
    blk /= a.tiles_x;
    const int by = blk % a.tiles_y;
    const int n = blk / a.tiles_y;
    const int y0 = by * 8, x0 = bx * 8;
    const float* f0n = a.f0 + (size_t)n * a.H * a.W * a.f0_cs;
    const float* f1n = a.f1 + (size_t)n * a.H * a.W * a.f1_cs;

    // ---- loader role
    const int pp = t >> 1, qb = (t & 1) * 4;         // patch pixel, first of its 4 quads
    const int prow = pp >> 4, pcol = pp & 15;
    const int yy = y0 + prow + dyi - 4, xx = x0 + pcol - 4;
    const bool pin = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;   // outside: zeros (pad2d)
    int o00 = 0, o01 = 0, o10 = 0, o11 = 0;
    float c00 = 1.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
    if (pin) {
        if (WARP) {
            // bilinear_warp, modules.py:107-137: un-clipped floors give the weights, the four corner
            // indices are clipped independently
            const float* fl = a.flow + (((size_t)n * a.H + yy) * a.W + xx) * a.flow_cs;
            const float fx = pwc_mul_rounded(fl[0], a.flow_scale), fy = pwc_mul_rounded(fl[1], a.flow_scale);   // rounded product (see warp_kernel)
            const float fx0 = floorf(fx), fy0 = floorf(fy);
            const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
            const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
            const int iy0 = (int)fminf(fmaxf((float)yy + fy0, 0.f), hl), iy1 = (int)fminf(fmaxf((float)yy + fy1, 0.f), hl);
            const int ix0 = (int)fminf(fmaxf((float)xx + fx0, 0.f), wl), ix1 = (int)fminf(fmaxf((float)xx + fx1, 0.f), wl);
            c00 = (fy1 - fy) * (fx1 - fx); c01 = (fy1 - fy) * (fx - fx0);
            c10 = (fy - fy0) * (fx1 - fx); c11 = (fy - fy0) * (fx - fx0);
            o00 = (iy0 * a.W + ix0) * a.f1_cs; o01 = (iy0 * a.W + ix1) * a.f1_cs;
            o10 = (iy1 * a.W + ix0) * a.f1_cs; o11 = (iy1 * a.W + ix1) * a.f1_cs;
        } else {
            o00 = (yy * a.W + xx) * a.f1_cs;
        }
    }
    const int fp = t >> 2, fqb = (t & 3) * 2;        // f0 tile pixel, first of its 2 quads
    const int fy_ = y0 + (fp >> 3), fx_ = x0 + (fp & 7);
    const bool fin = fy_ < a.H && fx_ < a.W;
    const int f0off = fin ? (fy_ * a.W + fx_) * a.f0_cs : 0;
    float* cpy = (a.f0_copy && dyi == 4 && fin) ? a.f0_copy + (((size_t)n * a.H + fy_) * a.W + fx_) * a.f0_copy_cs : nullptr;

    f32x4 g1[4], g0[2];
    auto gload = [&](int st) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = st * 32 + (qb + k) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pin && c < a.C) {
                if (WARP) {
                    v = c00 * *reinterpret_cast<const f32x4*>(f1n + o00 + c) + c01 * *reinterpret_cast<const f32x4*>(f1n + o01 + c) +
                        c10 * *reinterpret_cast<const f32x4*>(f1n + o10 + c) + c11 * *reinterpret_cast<const f32x4*>(f1n + o11 + c);
                } else {
                    v = *reinterpret_cast<const f32x4*>(f1n + o00 + c);
                }
            }
            g1[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = st * 32 + (fqb + k) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (fin && c < a.C) v = *reinterpret_cast<const f32x4*>(f0n + f0off + c);
            g0[k] = v;
        }
    };
    auto lstore = [&](int st, int buf) {
        float* b = smem + buf * CC_STAGE;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<f32x4*>(b + ((qb + k) * (8 * CC_PSTR) + prow * CC_PSTR + pcol) * 4) = g1[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            *reinterpret_cast<f32x4*>(b + CC_F1 + ((fqb + k) * 64 + fp) * 4) = g0[k];
            const int c = st * 32 + (fqb + k) * 4;
            if (cpy && c < a.C) *reinterpret_cast<f32x4*>(cpy + c) = g0[k];
        }
    };

    // ---- compute role
    const int py = lane >> 3, px = lane & 7;
    const int rbase = (py * CC_PSTR + px) * 4;
    float acc[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) acc[d] = 0.f;

    const int nst = (a.C + 31) >> 5;
    gload(0);
    lstore(0, 0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const bool more = st + 1 < nst;
        if (more) gload(st + 1);
        const float* b = smem + (CC_NBUF == 2 ? (st & 1) * CC_STAGE : 0);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * wave + qq;
            const f32x4 u = *reinterpret_cast<const f32x4*>(b + CC_F1 + (q * 64 + lane) * 4);
            const float* pq = b + q * (8 * CC_PSTR * 4) + rbase;
#pragma unroll
            for (int d = 0; d < 9; ++d) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(pq + d * 4);
                acc[d] = fmaf(u[0], w[0], fmaf(u[1], w[1], fmaf(u[2], w[2], fmaf(u[3], w[3], acc[d]))));
            }
        }
        if (CC_NBUF == 1) __syncthreads();               // single LDS image: everyone is done reading it
        if (more) lstore(st + 1, CC_NBUF == 2 ? (st + 1) & 1 : 0);
        __syncthreads();
    }

    // ---- sum the 4 waves' partial sums, mean over channels, leaky-relu, store
    float* red = smem;
#pragma unroll
    for (int d = 0; d < 9; ++d) red[(wave * 9 + d) * 64 + lane] = acc[d];
    __syncthreads();
    const float inv = 1.f / (float)a.C;
    for (int idx = t; idx < 64 * 9; idx += 256) {
        const int l = idx / 9, d = idx - l * 9;
        const int y = y0 + (l >> 3), x = x0 + (l & 7);
        if (y < a.H && x < a.W) {
            const float v = (red[d * 64 + l] + red[(9 + d) * 64 + l] + red[(18 + d) * 64 + l] + red[(27 + d) * 64 + l]) * inv;
            a.out[(((size_t)n * a.H + y) * a.W + x) * a.out_cs + dyi * 9 + d] = pwc_lrelu(v, a.slope);
        }
    }
}

extern "C" int pwc_cost_volume_coarse_f32(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow,
                                          int flow_cs, float flow_scale, float* out, int out_cs, float* f0_copy,
                                          int f0_copy_cs, int N, int H, int W, int C, int search_range, float slope,
                                          pwc_stream_t stream) {
    if (!f0 || !f1 || !out) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PWC_EINVAL;
    if (search_range != 4) return PWC_EUNSUPPORTED;
    if (f0_cs < C || f1_cs < C || out_cs < 81 || (flow && flow_cs < 2) || (f0_copy && f0_copy_cs < C)) return PWC_EINVAL;
    if ((C & 3) || (f0_cs & 3) || (f1_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1)) return PWC_EALIGN;
    if (f0_copy && ((f0_copy_cs & 3) || !pwc_aligned16(f0_copy))) return PWC_EALIGN;
    if ((long)H * W * f0_cs >= (1L << 31) || (long)H * W * f1_cs >= (1L << 31)) return PWC_ERANGE;
    CvCoarseArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = flow_scale; a.slope = slope;
    a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
    const long nblk = (long)N * a.tiles_x * a.tiles_y * 9;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    const size_t lds = (size_t)CC_NBUF * CC_STAGE * sizeof(float);
    if (flow)
        hipLaunchKernelGGL(cost_volume_coarse_kernel<true>, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(cost_volume_coarse_kernel<false>, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    return pwc_launch_status();
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

extern "C" int pwc_cost_volume_uses_rolling_kernel(int H, int W, int C, int search_range, int f0_cs, int f1_cs, int out_cs) {
    // pointers are taken as 16-byte aligned (what pwc_cost_volume_f32 checks at run time)
    const float* al = reinterpret_cast<const float*>(16);
    return cv_roll_eligible(al, f0_cs, al, f1_cs, al, out_cs, nullptr, 0, H, W, C, search_range) ? 1 : 0;
}

extern "C" int pwc_cost_volume_f32(const float* f0, int f0_cs, const float* f1w, int f1w_cs, float* out,
                                   int out_cs, int N, int H, int W, int C, int search_range, float slope,
                                   pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1w, f1w_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if (cv_roll_eligible(f0, f0_cs, f1w, f1w_cs, out, out_cs, nullptr, 0, H, W, C, search_range))
        return cv_roll_launch(f0, f0_cs, f1w, f1w_cs, out, out_cs, nullptr, 0, N, H, W, slope, (hipStream_t)stream);
    CvArgs a;
    a.f0 = f0; a.f1 = f1w; a.flow = nullptr; a.out = out;
    a.f0_cs = f0_cs; a.f1_cs = f1w_cs; a.flow_cs = 0; a.out_cs = out_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = 1.f; a.slope = slope;
    return cv_dispatch(a, search_range, false, (hipStream_t)stream);
}

extern "C" int pwc_warp_cost_volume_concat_supported(int H, int W, int C, int search_range, int f0_cs, int f1_cs,
                                                     int flow_cs, int out_cs, int f0_copy_cs) {
    const float* al = reinterpret_cast<const float*>(16);
    return cvm_eligible(al, f0_cs, al, f1_cs, flow_cs ? al : nullptr, flow_cs, al, out_cs, f0_copy_cs ? al : nullptr,
                        f0_copy_cs, H, W, C, search_range) ? 1 : 0;
}

extern "C" int pwc_warp_cost_volume_concat_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                               const float* flow, int flow_cs, float flow_scale, float* out,
                                               int out_cs, int out_pad_writable, float* f0_copy, int f0_copy_cs,
                                               int N, int H, int W, int C, int search_range, float slope,
                                               pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1, f1_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if ((flow && flow_cs < 2) || (f0_copy && f0_copy_cs < C)) return PWC_EINVAL;
    if (out_pad_writable && out_cs < 84) return PWC_EINVAL;
    if (out_pad_writable == 2) return PWC_EUNSUPPORTED;      // (the flow in channels 81, 82: pwc_warp_cost_volume_concat_h2_f32 only)
    if (search_range != 4 || !(C == 32 || C == 64 || C == 96)) return PWC_EUNSUPPORTED;
    if (!cvm_eligible(f0, f0_cs, f1, f1_cs, flow, flow_cs, out, out_cs, f0_copy, f0_copy_cs, H, W, C, search_range))
        return ((long)H * W * (long)(out_cs > f0_cs ? out_cs : f0_cs) * 4 >= (1L << 31)) ? PWC_ERANGE : PWC_EALIGN;
    return cvm_launch(f0, f0_cs, f1, f1_cs, flow, flow_cs, flow_scale, out, out_cs, out_pad_writable ? 1 : 0, f0_copy,
                      f0_copy_cs, N, H, W, C, slope, (hipStream_t)stream);
}

// Round 5: the same launch with the correlation on the F16 matrix pipe (two-term fp16 operand splits, fp32 accumulation;
// cost_volume_h2.hip).
extern "C" int pwc_warp_cost_volume_concat_h2_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                                  const float* flow, int flow_cs, float flow_scale, float* out,
                                                  int out_cs, int out_pad_writable, float* f0_copy, int f0_copy_cs,
                                                  int N, int H, int W, int C, int search_range, float slope,
                                                  pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1, f1_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if ((flow && flow_cs < 2) || (f0_copy && f0_copy_cs < C)) return PWC_EINVAL;
    if (out_pad_writable && out_cs < 84) return PWC_EINVAL;
    if (search_range != 4 || !(C == 32 || C == 64 || C == 96)) return PWC_EUNSUPPORTED;
    if (!cvm_eligible(f0, f0_cs, f1, f1_cs, flow, flow_cs, out, out_cs, f0_copy, f0_copy_cs, H, W, C, search_range))
        return ((long)H * W * (long)(out_cs > f0_cs ? out_cs : f0_cs) * 4 >= (1L << 31)) ? PWC_ERANGE : PWC_EALIGN;
    // out_pad_writable == 2 (round 6): channels 81, 82 of every record receive the pixel's flow (cost_volume_h2.hip, FLOWPAD)
    if (out_pad_writable == 2 && !flow) return PWC_EINVAL;
    return cvh_launch(f0, f0_cs, f1, f1_cs, flow, flow_cs, flow_scale, out, out_cs,
                      out_pad_writable == 2 ? 2 : (out_pad_writable ? 1 : 0), f0_copy, f0_copy_cs, N, H, W, C, slope,
                      (hipStream_t)stream);
}

// Round 5: the same operation for the small pyramid levels (C = 96 / 128 / 192): one 4 x 4 block per workgroup, the whole
// 12 x 12 window requested at once, correlation on the F16 matrix pipe (cost_volume_blk.hip).
extern "C" int pwc_warp_cost_volume_concat_blk_supported(int H, int W, int C, int search_range, int f0_cs, int f1_cs,
                                                         int flow_cs, int out_cs, int f0_copy_cs) {
    const float* al = reinterpret_cast<const float*>(16);
    return cvb_eligible(al, f0_cs, al, f1_cs, flow_cs ? al : nullptr, flow_cs, al, out_cs, f0_copy_cs ? al : nullptr,
                        f0_copy_cs, H, W, C, search_range) ? 1 : 0;
}

#ifdef PWC_HARNESS
extern "C" int pwc_debug_cost_volume_blk_rows(int rows) { cvb_rows_override = rows; return 0; }
#endif

extern "C" int pwc_warp_cost_volume_concat_blk_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                                   const float* flow, int flow_cs, float flow_scale, float* out,
                                                   int out_cs, int out_pad_writable, float* f0_copy, int f0_copy_cs,
                                                   int N, int H, int W, int C, int search_range, float slope,
                                                   pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1, f1_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if ((flow && flow_cs < 2) || (f0_copy && f0_copy_cs < C)) return PWC_EINVAL;
    if (out_pad_writable && out_cs < 84) return PWC_EINVAL;
    if (out_pad_writable == 2) return PWC_EUNSUPPORTED;      // (the flow in channels 81, 82: pwc_warp_cost_volume_concat_h2_f32 only)
    if (search_range != 4 || !(C == 64 || C == 96 || C == 128 || C == 192)) return PWC_EUNSUPPORTED;
    if (!cvb_eligible(f0, f0_cs, f1, f1_cs, flow, flow_cs, out, out_cs, f0_copy, f0_copy_cs, H, W, C, search_range))
        return ((long)H * W * (long)(out_cs > f0_cs ? out_cs : f0_cs) * 4 >= (1L << 31)) ? PWC_ERANGE : PWC_EALIGN;
    return cvb_launch(f0, f0_cs, f1, f1_cs, flow, flow_cs, flow_scale, out, out_cs, out_pad_writable ? 1 : 0, f0_copy,
                      f0_copy_cs, N, H, W, C, slope, (hipStream_t)stream);
}

extern "C" int pwc_warp_cost_volume_f32(const float* f0, int f0_cs, const float* f1, int f1_cs,
                                        const float* flow, int flow_cs, float flow_scale, float* out,
                                        int out_cs, int N, int H, int W, int C, int search_range, float slope,
                                        pwc_stream_t stream) {
    int rc = cv_check(f0, f0_cs, f1, f1_cs, out, out_cs, N, H, W, C, search_range);
    if (rc) return rc;
    if (!flow || flow_cs < 2) return PWC_EINVAL;
    if (cvm_eligible(f0, f0_cs, f1, f1_cs, flow, flow_cs, out, out_cs, nullptr, 0, H, W, C, search_range))
        return cvm_launch(f0, f0_cs, f1, f1_cs, flow, flow_cs, flow_scale, out, out_cs, 0, nullptr, 0, N, H, W, C, slope,
                          (hipStream_t)stream);
    CvArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs;
    a.N = N; a.H = H; a.W = W; a.C = C; a.flow_scale = flow_scale; a.slope = slope;
    return cv_dispatch(a, search_range, true, (hipStream_t)stream);
}
