// cost_volume_tile.hip -- warp + cost volume (+ the optional f0 concat copy) in ONE launch, correlation on the F16 matrix pipe, as
// independent 2-D TILES (round 6).  Search range 4, C = 32 (the 1/4-resolution level: two thirds of the correlation's bytes).
// Same operands, same results and the same arithmetic as cost_volume_h2.hip (reference model.py:105-112, modules.py:99-137,
// 158-204,264):
//
//   f1w[n,y,x,:]             = bilinear_warp(f1, flow * flow_scale)          (fp32, never written to memory)
//   out[n,y,x,(v+4)*9+(h+4)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  f1w zero outside
//   f0_copy[n,y,x,0:C]       = f0[n,y,x,0:C]                                  (optional)
//
// Why another organisation.  cost_volume_h2.hip walks a 16-column strip segment down the image: one workgroup per CU, 14 + 3 steps
// of ~3000 cycles each at the level this file is for -- a chain of LDS round trips, barriers and dependent matrix instructions at
// ONE consumer wave per SIMD: 33-34 us where the memory system's floor for the traffic is 21.  Here a workgroup
// (4 waves, 256 threads, 73 KB of LDS: TWO per CU) takes a tile of TBY = 2 block rows x 4 block columns (8 x 16 pixels), requests
// the TBY + 2 Q rows it meets (4 x 24 pixels each, all four bilinear corners) up front -- two rows in flight, the blend + split of
// a row under the requests of the next -- and then every wave computes and stores its block column with no further workgroup
// barrier.  No ring, no fill steps, no roles: the overlap of one tile's gather with another's matrix work and stores is the
// hardware's (two workgroups per CU, seven tiles per CU at batch 8), not a hand-written pipeline.  The price is the halo: a tile
// reads (TBY + 2) / TBY = 2 x its rows of f1 (the walking kernel 1.0 x + 3 fill rows per 14) -- from the L2, which the 3.7 MB of an
// image's f1 fit (XCD-aware tile order: an image's tiles on one XCD).
//
// Lane roles, LDS layouts (Q-row image, stage), the operand split, the K mapping and the copy-out are those of
// cost_volume_h2.hip / cost_volume_mfma.hip (see there).
//
// NOT IN THE LIBRARY (round 6 experiment, profiles/r06_exp_cost_volume_tile.txt): correct on every test of the entry point
// (tests/test_gpu_ops.py::test_tile_cost_volume_kernel_vs_oracle ran against it), bit-identical flows, and NOT faster: 35.2 us at
// batch 8 x 112 x 256 x 32 against the walking kernel's 34.0, whole forwards 2.606 against 2.593 ms.  To build it again: include it
// from cost_volume_h2.hip and route cvh_launch through cvt_launch where cvt_pays.
#pragma once
#include "../../pwcnet_amd/csrc/cost_volume_mfma.hip"

template <int TBY>
struct CvtGeom {
    using M = CvmGeom<2>;
    static constexpr int NQ = TBY + 2;                                  // Q-row images of a tile
    static constexpr int IMG_F = NQ * M::BUF * 4;                       // floats
    static constexpr int STG_F = M::NW * M::WSTG;                       // the waves' stages; the corner tables live here until the gather is done
    static constexpr int TAB_F = NQ * M::TAB;
    static constexpr int LDS_F = IMG_F + STG_F;
    static constexpr int NTE = NQ * M::NPIX;                            // corner table entries
    static constexpr int TPT = (NTE + 255) / 256;                       // ... per thread
    static_assert(TAB_F <= STG_F, "the tables do not fit the stage area");
    static_assert(2 * LDS_F * 4 <= 160 * 1024, "two workgroups per CU");
};

template <int TBY, bool WARP, bool PAD, bool FLOWPAD>
__global__ __launch_bounds__(256, 2) void cost_volume_tile_kernel(const CvmArgs a) {
    static_assert(!FLOWPAD || (WARP && PAD), "the flow rides in the padding channels of a warping launch");
    constexpr int CG = 2;
    using G = CvmGeom<CG>;
    using GT = CvtGeom<TBY>;
    constexpr int RS = G::RS, PLANE = G::PLANE, BUF = G::BUF, ITEMS = G::ITEMS, NQ = GT::NQ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* qimg = reinterpret_cast<f32x4*>(smem);                       // NQ Q-row images of BUF slots
    float* stg_all = smem + GT::IMG_F;
    float* tabf = stg_all;                                              // NQ corner tables of TAB dwords (dead before the stage is used)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);            // block column of the strip
    float* stg = stg_all + wave * G::WSTG;

    // ---- work item: (image, tile row, strip); XCD-aware order (the tiles of an image meet in one L2)
    const int id = pwc_xcd_remap(blockIdx.x, gridDim.x);
    const int sx = id % a.nstrips;
    const int rest = id / a.nstrips;
    const int ty = rest % a.nseg;
    const int n = rest / a.nseg;
    const int x0 = sx * G::SW;
    const int pb0 = ty * TBY;
    const int pb1 = min(pb0 + TBY, a.nbrows);
    const int qa = max(pb0 - 1, 0), qb = min(pb1, a.nbrows - 1);        // Q rows that hold image pixels

    const size_t npx = (size_t)a.H * a.W;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f0 + (size_t)n * npx * a.f0_cs), 0, (int)(npx * a.f0_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f1 + (size_t)n * npx * a.f1_cs), 0, (int)(npx * a.f1_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.out + (size_t)n * npx * a.out_cs), 0, (int)(npx * a.out_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(WARP ? a.flow + (size_t)n * npx * a.flow_cs : a.f1), 0, WARP ? (int)(npx * a.flow_cs * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f0_copy ? a.f0_copy + (size_t)n * npx * a.f0_copy_cs : a.out), 0,
        a.f0_copy ? (int)(npx * a.f0_copy_cs * 4) : 0, 0x00020000);

    // ---- lane roles of the matrix instructions (cost_volume_mfma.hip): lane = (block pixel m = lane & 15 -> row m >> 2, column
    // m & 3; channel quad kq = lane >> 4)
    const int mrow = (lane & 15) >> 2, mcol = lane & 3, kq = lane >> 4;
    const int ax = x0 + 4 * wave + mcol;                                // image column of this lane's f0 pixel
    const int bslot = mrow * RS + (4 * (wave + 1) + mcol) * 4 + kq;     // B operand slot of block column offset 0
    const bool a_in = ax < a.W;
    const unsigned a_rel = (unsigned)(mrow * a.W + ax);

    // ---- the block rows' f0 operands (and, FLOWPAD, their flows): requested first, they land under the gather
    f32x4 A[TBY][CG];
    float flx[TBY][2];
#pragma unroll
    for (int j = 0; j < TBY; ++j) {
        const int pb = pb0 + j;
        const bool ok = a_in && pb < pb1 && 4 * pb + mrow < a.H;
        const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g)
            A[j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)vo, g * 64, CVM_F0_AUX));
        flx[j][0] = 0.f; flx[j][1] = 0.f;
        if constexpr (FLOWPAD) {
            const unsigned fo = (ok && lane < 16) ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.flow_cs * 4) : CVM_OOB;
            flx[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)fo, 0, 0));
            flx[j][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)fo, 4, 0));
        }
    }

    // ---- corner tables of the NQ Q rows (WARP): entry e = i * 256 + t -> (Q row e / 96, pixel e % 96)
    if constexpr (WARP) {
        float fv[GT::TPT][2];
#pragma unroll
        for (int i = 0; i < GT::TPT; ++i) {
            const int e = i * 256 + t;
            const int r = (e * 2731) >> 18, p = e - r * G::NPIX;        // e / 96 for e < 576
            const int pr = (p * 2731) >> 16, pxi = p - pr * 24;         // p / 24
            const int qq = pb0 - 1 + r;
            const int gy = 4 * qq + pr, gx = x0 - 4 + pxi;
            const bool ok = e < GT::NTE && qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)((gy * a.W + gx) * a.flow_cs) * 4u : CVM_OOB;
            fv[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 0, 0));
            fv[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 4, 0));
        }
#pragma unroll
        for (int i = 0; i < GT::TPT; ++i) {
            const int e = i * 256 + t;
            if (e < GT::NTE) {
                const int r = (e * 2731) >> 18, p = e - r * G::NPIX;
                const int pr = (p * 2731) >> 16, pxi = p - pr * 24;
                const int qq = pb0 - 1 + r;
                const int gy = 4 * qq + pr, gx = x0 - 4 + pxi;
                const bool ok = qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                // bilinear_warp, modules.py:107-137: the product flow * scale is rounded first (model.py:109 is an op of its own),
                // weights from the un-clipped floors, the four corner indices clipped independently
                const float fx = pwc_mul_rounded(fv[i][0], a.flow_scale), fy = pwc_mul_rounded(fv[i][1], a.flow_scale);
                const float fx0 = floorf(fx), fy0 = floorf(fy);
                const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
                const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
                const int iy0 = (int)fminf(fmaxf((float)gy + fy0, 0.f), hl), iy1 = (int)fminf(fmaxf((float)gy + fy1, 0.f), hl);
                const int ix0 = (int)fminf(fmaxf((float)gx + fx0, 0.f), wl), ix1 = (int)fminf(fmaxf((float)gx + fx1, 0.f), wl);
                const f32x4 w = {(fy1 - fy) * (fx1 - fx), (fy1 - fy) * (fx - fx0), (fy - fy0) * (fx1 - fx), (fy - fy0) * (fx - fx0)};
                const unsigned cs4 = (unsigned)a.f1_cs * 4u;
                cvm_u32x4 off = {(unsigned)(iy0 * a.W + ix0) * cs4, (unsigned)(iy0 * a.W + ix1) * cs4,
                                 (unsigned)(iy1 * a.W + ix0) * cs4, (unsigned)(iy1 * a.W + ix1) * cs4};
                if (!ok) off = cvm_u32x4{CVM_OOB, CVM_OOB, CVM_OOB, CVM_OOB};
                float* en = tabf + r * G::TAB + p * 8;
                *reinterpret_cast<cvm_u32x4*>(en) = off;
                *reinterpret_cast<f32x4*>(en + 4) = w;
            }
        }
        cvm_barrier();
    }

    // ---- gather items of a Q row: item e = i * 256 + t -> (pixel, plane g, quad): the C/4 quads of a pixel sit in consecutive
    // lanes, so one instruction asks for whole 128-byte lines of a corner pixel (cost_volume_h2.hip)
    unsigned k_chan[ITEMS];
    int k_tab[ITEMS], k_img[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int e = i * 256 + t;
        const int pix = e >> 3;                                         // e / (C / 4)
        const int cq = e - pix * (CG * 4);
        const int g = cq >> 2, kqi = cq & 3;
        const int r = (pix * 2731) >> 16;                               // pix / 24 for pix < 96
        const int xi = pix - r * 24;
        k_chan[i] = (unsigned)(g * 64 + kqi * 16);                      // byte offset of the channel quad
        k_tab[i] = pix * 8;                                             // table entry (dwords)
        k_img[i] = ((g >> 1) * 2 * PLANE + r * RS + xi * 4 + kqi) * 16 + (g & 1) * 8;   // byte offset of the item's 8 bytes of h
    }
    auto g_issue = [&](f32x4 (*gv)[WARP ? 4 : 1], int r) {
        const int qq = pb0 - 1 + r;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            cvm_u32x4 off;
            if constexpr (WARP) {
                off = *reinterpret_cast<const cvm_u32x4*>(tabf + r * G::TAB + k_tab[i]);
            } else {
                const int pix = k_tab[i] >> 3;
                const int pr = (pix * 2731) >> 16, xi = pix - pr * 24;
                const int gy = 4 * qq + pr, gx = x0 - 4 + xi;
                const bool ok = qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                off[0] = ok ? (unsigned)((gy * a.W + gx) * a.f1_cs) * 4u : CVM_OOB;
            }
#pragma unroll
            for (int c = 0; c < (WARP ? 4 : 1); ++c)                    // (out-of-range + chan stays out of range)
                gv[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, (int)(off[c] + k_chan[i]), 0, 0));
        }
    };
    auto g_commit = [&](f32x4 (*gv)[WARP ? 4 : 1], int r) {
        char* image = reinterpret_cast<char*>(qimg + r * BUF);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            f32x4 v;
            if constexpr (WARP) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(tabf + r * G::TAB + k_tab[i] + 4);
                // modules.py:132-135: c00*x00 + c01*x01 + c10*x10 + c11*x11, summed left to right (1/C multiplies the finished sum)
                v = w[0] * gv[i][0];
                v = __builtin_elementwise_fma(f32x4{w[1], w[1], w[1], w[1]}, gv[i][1], v);
                v = __builtin_elementwise_fma(f32x4{w[2], w[2], w[2], w[2]}, gv[i][2], v);
                v = __builtin_elementwise_fma(f32x4{w[3], w[3], w[3], w[3]}, gv[i][3], v);
            } else {
                v = gv[i][0];
            }
            pwc_f16x4 h, m;
            pwc_split4(v, h, m);
            *reinterpret_cast<pwc_f16x4*>(image + k_img[i]) = h;
            *reinterpret_cast<pwc_f16x4*>(image + k_img[i] + PLANE * 16) = m;
        }
    };
    {
        // two Q rows in flight: the blend + split of a row runs under the requests of the next
        f32x4 gva[ITEMS][WARP ? 4 : 1], gvb[ITEMS][WARP ? 4 : 1];
        g_issue(gva, 0);
        g_issue(gvb, 1);
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            if (r & 1) {
                g_commit(gvb, r);
                if (r + 2 < NQ) g_issue(gvb, r + 2);
            } else {
                g_commit(gva, r);
                if (r + 2 < NQ) g_issue(gva, r + 2);
            }
        }
    }
    cvm_barrier();                                                      // the images are complete; the tables are dead

    // ---- D fragment: lane holds P pixels (row kq, column r = 0..3) x Q pixel (row mrow, column mcol).  Stage address of
    // entry (by, bx, r) = sbase + 83 r + 36 by + 4 bx floats; entries with |dx| > 4 go to the dump area, entries with
    // |dy| > 4 are masked off per lane
    const int sbase = kq * (4 * G::SROW) + (mrow - kq + 4) * 9 + mcol + 4;
    float* sxa[3][4];
#pragma unroll
    for (int bxi = 0; bxi < 3; ++bxi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool vx = bxi == 0 ? (mcol >= r) : (bxi == 2 ? (mcol <= r) : true);
            sxa[bxi][r] = stg + (vx ? sbase + r * (G::SROW - 1) + 4 * (bxi - 1) - 36 : G::DUMP + lane);
        }
    const int vy_m_i = mrow >= kq, vy_p_i = mrow <= kq;                 // by = -1 / +1: |dy| <= 4
    // copy-out items: e = i * 64 + lane -> (pixel p = e / 21 of the block, quad e % 21)
    unsigned co_rel[6];
    unsigned q20 = 0;                                                   // bit i: item i is quad 20 of its pixel (channels 80..83)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = i * 64 + lane;
        const int p = (e * 3121) >> 16, qd = e - p * 21;                // e / 21 for e < 336
        if (qd == 20) q20 |= 1u << i;
        const bool in = e < 16 * 21 && x0 + 4 * wave + (p & 3) < a.W && (PAD || qd < 20);
        co_rel[i] = in ? (unsigned)((((p >> 2) * a.W + (p & 3)) * a.out_cs + qd * 4) * 4) : CVM_OOB;
    }
    const unsigned c80_rel = (lane < 16 && x0 + 4 * wave + (lane & 3) < a.W)
                                 ? (unsigned)((((lane >> 2) * a.W + (lane & 3)) * a.out_cs + 80) * 4) : CVM_OOB;
    // (the stage area held the tables until the barrier above) padding channels 81..83 stay zero from here on
    if (lane < 48) stg[(lane / 3) * G::SROW + 81 + (lane % 3)] = 0.f;

    // ---- per block row of the tile: split the f0 operand (+ its concat copy), the nine tiles, stage, copy-out
#pragma unroll
    for (int j = 0; j < TBY; ++j) {
        const int pb = pb0 + j;
        if (pb >= pb1) break;                                           // uniform
        const bool okr = a_in && 4 * pb + mrow < a.H;
        pwc_f16x8 AH, AM;
        {
            const unsigned vo = okr ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_copy_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
            for (int g = 0; g < CG; ++g)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, A[j][g]), rc, (int)vo, g * 64, CVM_COPY_AUX);
            pwc_f16x4 h0, m0, h1, m1;
            pwc_split4(A[j][0], h0, m0);
            pwc_split4(A[j][1], h1, m1);
            AH = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            AM = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        // B operands: images j, j + 1, j + 2 = Q rows pb - 1, pb, pb + 1; all eighteen reads together
        pwc_f16x8 Lh[3][3], Lm[3][3];
#pragma unroll
        for (int byi = 0; byi < 3; ++byi) {
            const pwc_f16x8* imgh = reinterpret_cast<const pwc_f16x8*>(qimg + (j + byi) * BUF);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) {
                Lh[byi][bx] = imgh[bslot + (bx - 1) * 16];
                Lm[byi][bx] = imgh[PLANE + bslot + (bx - 1) * 16];
            }
        }
        f32x4 acc[3][3];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int byi = 0; byi < 3; ++byi) {
            f32x4 hh[3], xx[3];
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) xx[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, Lm[byi][bx], zero, 0, 0, 0);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) hh[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, Lh[byi][bx], zero, 0, 0, 0);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) xx[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AM, Lh[byi][bx], xx[bx], 0, 0, 0);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx)
                acc[byi][bx] = __builtin_elementwise_fma(xx[bx], f32x4{1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f}, hh[bx]);
        }
        cvm_wave_sync();                                                // the previous copy-out has read the stage
#pragma unroll
        for (int byi = 0; byi < 3; ++byi)
            if (byi == 1 || (byi == 0 ? vy_m_i : vy_p_i)) {
#pragma unroll
                for (int bxi = 0; bxi < 3; ++bxi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sxa[bxi][r][36 * byi] = acc[byi][bxi][r];
            }
        if constexpr (FLOWPAD) {
            if (lane < 16) {
                stg[lane * G::SROW + 81] = flx[j][0];
                stg[lane * G::SROW + 82] = flx[j][1];
            }
        }
        cvm_wave_sync();
        // copy-out: the six stage reads together, then mean = sum * (1/C) (reduce_mean, modules.py:181), leaky-relu max(x, slope x)
        // and a 16-byte store per quad
        {
            const int ylim = a.H - 4 * pb;                              // rows of this block inside the image
            const unsigned base = (unsigned)((4 * pb * a.W + x0 + 4 * wave) * a.out_cs) * 4u;
            f32x4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const f32x4*>(stg + (i * 64 + lane < 16 * 21 ? (i * 64 + lane) * 4 : 0));
            float x80 = 0.f;
            if constexpr (!PAD) x80 = stg[(lane & 15) * G::SROW + 80];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 mv = v[i] * a.inv_c;
                const f32x4 sv = mv * a.slope;
                f32x4 y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float yk;
                    asm("v_max_f32 %0, %1, %2" : "=v"(yk) : "v"(mv[k]), "v"(sv[k]));
                    y[k] = yk;
                }
                if constexpr (FLOWPAD) {                                // channels 81, 82 carry the flow as read: no mean, no activation
                    const bool f = (q20 >> i) & 1u;
                    y[1] = f ? v[i][1] : y[1];
                    y[2] = f ? v[i][2] : y[2];
                }
                const bool ok = i * 64 + lane < 84 * ylim;              // 84 items per block row
                const unsigned vo = ok ? base + co_rel[i] : CVM_OOB;    // out-of-range + base stays out of range
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, y), ro, (int)vo, 0, CVM_STORE_AUX);
            }
            if constexpr (!PAD) {
                const float y = pwc_lrelu(x80 * a.inv_c, a.slope);
                const bool ok = (lane >> 2) < ylim;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ro, (int)(ok ? base + c80_rel : CVM_OOB), 0, CVM_STORE_AUX);
            }
        }
    }
}

constexpr int CVT_TBY = 2;
#ifdef PWC_HARNESS
static int cvt_tile_mode = 0;      // libpwc_hip_harness.so only (pwc_debug_cost_volume_tile): 0 = cvt_pays decides, 1 = never, 2 = every C = 32 launch
#else
constexpr int cvt_tile_mode = 0;
#endif

// The tile form pays where it fills the chip at two workgroups per CU for at least two rounds (batch 8 at 112 x 256: 1792 tiles);
// smaller launches stay with the row-walking kernel (whose strips then have few steps to walk anyway).
static bool cvt_pays(int N, int H, int W, int C) {
    if (C != 32) return false;
    const long tiles = (long)N * ((W + 15) / 16) * (((H + 3) / 4 + CVT_TBY - 1) / CVT_TBY);
    return tiles >= 1024;
}

template <bool WARP, bool PAD, bool FLOWPAD = false>
static int cvt_launch_t(CvmArgs& a, hipStream_t s) {
    using GT = CvtGeom<CVT_TBY>;
    const size_t lds = (size_t)GT::LDS_F * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_tile_kernel<CVT_TBY, WARP, PAD, FLOWPAD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    a.nstrips = (a.W + 15) / 16;
    a.seg_brows = CVT_TBY;
    a.nseg = (a.nbrows + CVT_TBY - 1) / CVT_TBY;
    const long items = (long)a.N * a.nstrips * a.nseg;
    if (items >= (1L << 31)) return PWC_ERANGE;
    hipLaunchKernelGGL((cost_volume_tile_kernel<CVT_TBY, WARP, PAD, FLOWPAD>), dim3((unsigned)items), dim3(256), lds, s, a);
    return pwc_launch_status();
}

// (a: filled by cvh_launch; C = 32)
static int cvt_launch(CvmArgs& a, bool has_flow, int pad_ok, hipStream_t s) {
    if (has_flow && pad_ok == 2) return cvt_launch_t<true, true, true>(a, s);
    return has_flow ? (pad_ok ? cvt_launch_t<true, true>(a, s) : cvt_launch_t<true, false>(a, s))
                    : (pad_ok ? cvt_launch_t<false, true>(a, s) : cvt_launch_t<false, false>(a, s));
}
