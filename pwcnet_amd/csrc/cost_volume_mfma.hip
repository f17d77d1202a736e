// cost_volume_mfma.hip -- warp + cost volume (+ the f0 part of the estimator input's concat) in ONE launch,
// correlation on the matrix pipe, for gfx950.  Search range 4, C = 32 / 64 / 96 (C = 128 / 192, the two coarsest levels, stay on cost_volume_coarse_kernel: pwc_warp_cost_volume_concat_supported).
//
// Replaces, for pyramid levels with C in {32, 64, 96} (reference model.py:105-112, modules.py:99-137,158-204,264):
//
//   f1w[n,y,x,:]             = bilinear_warp(f1, flow * flow_scale)          (never written to memory)
//   out[n,y,x,(v+4)*9+(h+4)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  f1w zero outside
//   f0_copy[n,y,x,0:C]       = f0[n,y,x,0:C]                                  (optional)
//
// Structure
//   blocks  the +-4 window has an exact block structure: for a 4 x 4-pixel block P of f0 (16 pixels = the M
//           side of v_mfma_f32_16x16x4_f32) the 12 x 12 window pixels are a 3 x 3 grid of 4 x 4-pixel blocks
//           Q of f1w (16 pixels = the N side).  One MFMA = 16 x 16 dot-product pieces over 4 channels; 81 of
//           the 144 pairs of a P pixel are wanted (56 % of the matrix pipe's work is useful: an effective
//           88 TFLOP/s, what the fp32 VALU reaches on paper -- but with one ds_read_b128 feeding 12 MFMAs
//           instead of 9 FMAs, and with the VALU free for the warp, the activation and the addressing).
//   stream  a workgroup (4 waves, wave = block column) owns a 16-column STRIP SEGMENT and walks down it one
//           block row of Q (4 image rows x 24 pixels: the strip + 4 columns of halo on both sides) per step.
//           Q row q meets the P rows q-1, q, q+1: a wave keeps THREE generations of accumulators (3 x 9 tiles
//           x 4 registers = 108 ... of which 18 tiles = 72 registers are live) and the f0 operands of the three
//           P rows in registers; after the step in which Q row q arrived, P row q-1 is complete.  LDS holds two
//           Q-row images only (25 KB at C = 32), so two or three workgroups share a CU and overlap their
//           gather / matrix / store phases without any hand-written software pipeline across workgroups.
//   gather  every lane fetches (pixel, 16-byte channel quad) items of the NEXT Q row: the four bilinear corners
//           straight from f1 (global_load_dwordx4 each, L2 / L1 absorb the shared corners), blended in
//           registers, ds_write_b128 into the other Q-row image.  Corner offsets and weights of the 96 pixels
//           of a Q row come from a small LDS table written one step earlier by 96 lanes (flow read two steps
//           ahead), so the 4 x C/4 items of a pixel do not repeat the floor / clip arithmetic.
//   f0      the A operand goes global -> registers directly in its MFMA layout (lane = pixel x channel quad):
//           no LDS; the same registers are the source of the concat copy.
//   banks   Q-row image = C/16 planes of [4 rows][24 pixels][64 bytes], row stride 98 slots of 16 bytes (skew
//           2): the 16 lanes of every ds_read_b128 service group (4 pixels x {2 rows x 2 quads}) cover the 16
//           slots of a bank row exactly once.
//   out     accumulators -> mean, leaky-relu -> wave-private LDS stage of 84-float pixel records (entries
//           outside the +-4 window go to a dump slot) -> 16-byte buffer stores of contiguous records; pixels
//           beyond the image edge are dropped by the buffer range check.
//
// Algorithmic bytes N*H*W*(2C+2+81)*4; HBM-bound (AI 8.9 flop/B at C = 32).
#pragma once
#include "pwc_common.h"
#include <type_traits>

struct CvmArgs {
    const float* f0;
    const float* f1;
    const float* flow;      // null (WARP = false): f1 is used as is
    float* out;
    float* f0_copy;         // null: no concat copy
    int f0_cs, f1_cs, flow_cs, out_cs, f0_copy_cs;
    int N, H, W;
    float flow_scale, slope, inv_c;
    int nstrips, nseg, seg_brows, nbrows;
    int pad_ok;             // channels 81..83 of every `out` record may be written (with zeros)
    long long* dbg;         // scripts/exp_cv3.hip only (ABL & 8): s_memtime stamps of two workgroups
};

template <int CG>
struct CvmGeom {
    static constexpr int C = 16 * CG;
    static constexpr int NW = 4, T = 64 * NW, SW = 4 * NW, QW = SW + 8;
    static constexpr int NPIX = 4 * QW;                  // pixels of a Q row (96)
    static constexpr int RS = QW * 4 + 2;                // 16-byte slots per plane row (98 = 2 mod 4: see `banks`)
    static constexpr int PLANE = 4 * RS + 4;             // slots per 16-channel plane (+64 B: the two planes a gather
                                                         // instruction writes land in different banks)
    static constexpr int BUF = CG * PLANE;               // slots per Q-row image
    static constexpr int ITEMS = CG * NPIX * 4 / T;      // gather items per lane and Q row
    static constexpr int NB = ITEMS / 3;                 // batches of 3 items (48 registers in flight)
    static constexpr int SROW = 84;                      // stage floats per pixel record
    static constexpr int DUMP = 16 * SROW;               // dump slots: 64 lanes, reached with 0 / 36 / 72-float immediates
    static constexpr int WSTG = 16 * SROW + 144;         // stage floats per wave: its 4 x 4 block + the dump area
    static constexpr int TAB = NPIX * 8;                 // dwords per corner table
    static constexpr int LDS_F = 2 * BUF * 4 + NW * WSTG + 2 * TAB;
    static constexpr int WGPC = LDS_F * 4 * 2 <= 160 * 1024 ? 2 : 1;   // workgroups per CU (LDS; registers follow)
    static_assert(CG % 2 == 0 && ITEMS % 3 == 0 && NB >= 1 && NB <= 3, "C must be 32, 64 or 96");
    static_assert(RS % 4 == 2, "row skew");
    static_assert(LDS_F * 4 <= 160 * 1024, "does not fit the LDS");
};

#define CVM_OOB 0x80000000u
#ifndef CVM_ITEM_LINES
#define CVM_ITEM_LINES 1
#endif
// cache policy bits of the builtins' aux operand (gfx940+): 0 = default, 2 = nt (streaming), 16 = sc1
#ifndef CVM_STORE_AUX
#define CVM_STORE_AUX 2
#endif
#ifndef CVM_F0_AUX
#define CVM_F0_AUX 0
#endif
#ifndef CVM_COPY_AUX
#define CVM_COPY_AUX 0
#endif
// start delay (s_sleep units of 64 cycles) of the workgroups of the second half of the grid: the two workgroups of
// a CU then alternate their matrix and memory phases instead of meeting in them
#ifndef CVM_STAGGER
#define CVM_STAGGER 0
#endif
// phase boundaries of a step (scripts/exp_cv3.hip builds a variant without them)
#ifndef CVM_NO_SCHED_BARRIER
#define CVM_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define CVM_SCHED_BARRIER()
#endif
typedef unsigned int cvm_u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only (__syncthreads() would also drain vmcnt: the gathers in flight
// and the copy-out stores)
__device__ __forceinline__ void cvm_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// LDS hand-over between the lanes of ONE wave (a wave's LDS instructions execute in order: compiler fence only)
__device__ __forceinline__ void cvm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// f(std::integral_constant<int, i>{}) for i = 0 .. N-1
template <int N, int I = 0, class F>
__device__ __forceinline__ void cvm_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cvm_for<N, I + 1>(f);
    }
}

// ABL (scripts/exp_cv3.hip only; 0 in the library): 1 = no MFMAs, 2 = no gather loads, 4 = no stores
// PAD: channels 81..83 of every `out` record are the kernel's to zero (estimator buffers: padding channels) -- the
// record is then 21 full 16-byte quads.
template <int CG, bool WARP, bool PAD, int ABL = 0>
__global__ __launch_bounds__(256, CvmGeom<CG>::WGPC) void cost_volume_mfma_kernel(const CvmArgs a) {
    using G = CvmGeom<CG>;
    constexpr int RS = G::RS, PLANE = G::PLANE, BUF = G::BUF, NB = G::NB, ITEMS = G::ITEMS;
    constexpr bool KEEP = CG <= 2;                                      // per-item constants live in registers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* qimg = reinterpret_cast<f32x4*>(smem);                       // 2 Q-row images of BUF slots
    float* stg_all = smem + 2 * BUF * 4;
    float* tabf = stg_all + G::NW * G::WSTG;                            // 2 corner tables of TAB dwords

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);            // block column of the strip
    float* stg = stg_all + wave * G::WSTG;

    // ---- work item: (image, strip, segment); XCD-aware order (neighbouring strips meet in one L2)
    const int id = pwc_xcd_remap(blockIdx.x, gridDim.x);
    const int sx = id % a.nstrips;
    const int rest = id / a.nstrips;
    const int sg = rest % a.nseg;
    const int n = rest / a.nseg;
    const int x0 = sx * G::SW;
    const int pb0 = sg * a.seg_brows;
    const int pb1 = min(pb0 + a.seg_brows, a.nbrows);
    if (pb0 >= pb1) return;                                             // uniform
    if (CVM_STAGGER > 0 && blockIdx.x >= 256 && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(CVM_STAGGER);
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

    // ---- lane roles
    // MFMA operands: lane = (block pixel m = lane & 15 -> row m >> 2, column m & 3; channel quad kq = lane >> 4)
    const int mrow = (lane & 15) >> 2, mcol = lane & 3, kq = lane >> 4;
    const int ax = x0 + 4 * wave + mcol;                                // image column of this lane's f0 pixel
    const int bslot = mrow * RS + (4 * (wave + 1) + mcol) * 4 + kq;     // B operand slot of block column offset 0
    // f0 pixel of this lane relative to the block row's first image row, in pixels; columns beyond the image never load
    const bool a_in = ax < a.W;
    const unsigned a_rel = (unsigned)(mrow * a.W + ax);

    // D fragment: lane holds P pixels (row kq, column r = 0..3) x Q pixel (row mrow, column mcol).  Stage address of
    // entry (by, bx, r) = sbase + 83 r + 36 by + 4 bx floats; entries with |dx| > 4 go to the dump area (whose
    // +-36-float `by` immediates stay inside it), entries with |dy| > 4 are masked off per lane
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

    // copy-out items: e = i * 64 + lane -> (pixel p = e / 21 of the block, quad e % 21); byte offset relative to
    // the block's first pixel (out-of-range: the strip's columns beyond the image, e >= 336, and -- without PAD --
    // the last quad, whose first float goes out through a 4-byte store of its own)
    unsigned co_rel[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int e = i * 64 + lane;
        const int p = (e * 3121) >> 16, qd = e - p * 21;                // e / 21 for e < 336
        const bool in = e < 16 * 21 && x0 + 4 * wave + (p & 3) < a.W && (PAD || qd < 20);
        co_rel[i] = in ? (unsigned)((((p >> 2) * a.W + (p & 3)) * a.out_cs + qd * 4) * 4) : CVM_OOB;
    }
    const unsigned c80_rel = (lane < 16 && x0 + 4 * wave + (lane & 3) < a.W)
                                 ? (unsigned)((((lane >> 2) * a.W + (lane & 3)) * a.out_cs + 80) * 4) : CVM_OOB;

    // stage padding channels 81..83 stay zero for the whole launch
    if (lane < 48) stg[(lane / 3) * G::SROW + 81 + (lane % 3)] = 0.f;

    // ---- gather items of a Q row: item e = i * 256 + t -> (plane g, pixel, quad t & 3)
    auto item_decode = [&](int i, unsigned& chan, int& tab_d, int& img_s) {
        int tt = t;
        if (!KEEP) asm volatile("" : "+v"(tt));                         // recomputed per use: too many to keep
#if CVM_ITEM_LINES
        // the C/4 quads of a pixel sit in consecutive lanes: one instruction asks for whole 128-byte lines of a
        // corner pixel (with plane-major items every line was requested by C/16 different instructions)
        const int e = i * 256 + tt;
        const int pix = (CG == 2) ? (e >> 3) : (CG == 4) ? (e >> 4) : ((e * 2731) >> 16);   // e / (C/4); e / 24 for e < 2304
        const int cq = e - pix * (CG * 4);
        const int g = cq >> 2, kqi = cq & 3;
#else
        const int u = i * 64 + (tt >> 2);
        const int g = (u * 683) >> 16;                                  // u / 96 for u < 1152
        const int pix = u - g * 96;
        const int kqi = tt & 3;
#endif
        const int r = (pix * 2731) >> 16;                               // pix / 24 for pix < 96
        const int xi = pix - r * 24;
        chan = (unsigned)(g * 64 + kqi * 16);                           // byte offset of the channel quad
        tab_d = pix * 8;                                                // table entry (dwords)
        img_s = g * PLANE + r * RS + xi * 4 + kqi;                      // Q-row image slot
    };
    unsigned k_chan[KEEP ? ITEMS : 1];
    int k_tab[KEEP ? ITEMS : 1], k_img[KEEP ? ITEMS : 1];
    if (KEEP) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) item_decode(i, k_chan[i], k_tab[i], k_img[i]);
    }
    auto item = [&](int i, unsigned& chan, int& tab_d, int& img_s) {
        if (KEEP) { chan = k_chan[i]; tab_d = k_tab[i]; img_s = k_img[i]; }
        else item_decode(i, chan, tab_d, img_s);
    };
    // !WARP: pixel of item i relative to the Q row's first pixel (row 4 qq, column x0 - 4), or out of range
    auto nowarp_off = [&](int tab_d, int qq) -> unsigned {
        const int pix = tab_d >> 3;
        const int r = (pix * 2731) >> 16, xi = pix - r * 24;
        const int gy = 4 * qq + r, gx = x0 - 4 + xi;
        const bool ok = qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        return ok ? (unsigned)((gy * a.W + gx) * a.f1_cs) * 4u : CVM_OOB;
    };

    int stamp_i = 0;
    auto stamp = [&]() {
        if (ABL & 8) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tk;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0 && (wave == 0 || wave == 3) && stamp_i < 128)
                a.dbg[(blockIdx.x ? 256 : 0) + (wave ? 128 : 0) + stamp_i] = (long long)tk;
            ++stamp_i;
        }
    };

    // ---- gather of a Q row, as micro-operations that the step places into the shadows of its MFMAs:
    //   gi_tab(b, qq)     the three table reads of batch b (one LDS round trip)
    //   gi_load(b, j)     the 4 (1) corner loads of item j
    //   gc_item(b, j, qq) blend + ds_write_b128 of item j into Q-row image qq & 1
    f32x4 gv[3][WARP ? 4 : 1];
    unsigned gchan[3];
    cvm_u32x4 goff[3];
    auto gi_tab = [&](int batch, int qq) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int tab_d, img_s;
            item(batch * 3 + j, gchan[j], tab_d, img_s);
            if (WARP) goff[j] = *reinterpret_cast<const cvm_u32x4*>(tabf + (qq & 1) * G::TAB + tab_d);
            else goff[j][0] = nowarp_off(tab_d, qq);
        }
    };
    auto gi_load = [&](int j) {
#pragma unroll
        for (int c = 0; c < (WARP ? 4 : 1); ++c) {
            const unsigned vo = (ABL & 2) ? CVM_OOB : goff[j][c] + gchan[j];       // out-of-range + chan stays out of range
            gv[j][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, (int)vo, 0, 0));
        }
    };
    auto gc_item = [&](int batch, int j, int qq) {
        unsigned chan;
        int tab_d, img_s;
        item(batch * 3 + j, chan, tab_d, img_s);
        f32x4 v;
        if (WARP) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(tabf + (qq & 1) * G::TAB + tab_d + 4);
            // modules.py:132-135: c00*x00 + c01*x01 + c10*x10 + c11*x11, summed left to right (the weights carry the
            // 1/C of the mean: exact for C = 32, 64)
            v = w[0] * gv[j][0];
            v = __builtin_elementwise_fma(f32x4{w[1], w[1], w[1], w[1]}, gv[j][1], v);
            v = __builtin_elementwise_fma(f32x4{w[2], w[2], w[2], w[2]}, gv[j][2], v);
            v = __builtin_elementwise_fma(f32x4{w[3], w[3], w[3], w[3]}, gv[j][3], v);
        } else {
            v = gv[j][0] * a.inv_c;
        }
        // rows without pixels are never read: their (out-of-range, zero) items may land anywhere in the image
        qimg[(qq & 1) * BUF + img_s] = v;
    };

    // ---- corner table of a Q row (WARP): 96 lanes, one pixel each
    float fl0 = 0.f, fl1 = 0.f;
    const int t_r = (t * 2731) >> 16, t_xi = t - t_r * 24;              // this lane's table pixel (t < 96)
    auto flow_issue = [&](int qq, float& f0v, float& f1v) {
        const int gy = 4 * qq + t_r, gx = x0 - 4 + t_xi;
        const bool ok = t < G::NPIX && qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const unsigned vo = ok ? (unsigned)((gy * a.W + gx) * a.flow_cs) * 4u : CVM_OOB;
        f0v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 0, 0));
        f1v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 4, 0));
    };
    auto table_write = [&](int qq, float f0v, float f1v) {
        if (t < G::NPIX) {
            const int gy = 4 * qq + t_r, gx = x0 - 4 + t_xi;
            const bool ok = qq >= qa && qq <= qb && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            // bilinear_warp, modules.py:107-137: the product flow * scale is rounded first (model.py:109 is an op of
            // its own), weights from the un-clipped floors, the four corner indices clipped independently
            const float fx = pwc_mul_rounded(f0v, a.flow_scale), fy = pwc_mul_rounded(f1v, a.flow_scale);
            const float fx0 = floorf(fx), fy0 = floorf(fy);
            const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
            const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
            const int iy0 = (int)fminf(fmaxf((float)gy + fy0, 0.f), hl), iy1 = (int)fminf(fmaxf((float)gy + fy1, 0.f), hl);
            const int ix0 = (int)fminf(fmaxf((float)gx + fx0, 0.f), wl), ix1 = (int)fminf(fmaxf((float)gx + fx1, 0.f), wl);
            f32x4 w = {(fy1 - fy) * (fx1 - fx), (fy1 - fy) * (fx - fx0), (fy - fy0) * (fx1 - fx), (fy - fy0) * (fx - fx0)};
            w = w * a.inv_c;
            const unsigned cs4 = (unsigned)a.f1_cs * 4u;
            cvm_u32x4 off = {(unsigned)(iy0 * a.W + ix0) * cs4, (unsigned)(iy0 * a.W + ix1) * cs4,
                             (unsigned)(iy1 * a.W + ix0) * cs4, (unsigned)(iy1 * a.W + ix1) * cs4};
            if (!ok) off = cvm_u32x4{CVM_OOB, CVM_OOB, CVM_OOB, CVM_OOB};
            float* e = tabf + (qq & 1) * G::TAB + t * 8;
            *reinterpret_cast<cvm_u32x4*>(e) = off;
            *reinterpret_cast<f32x4*>(e + 4) = w;
        }
    };

    f32x4 acc[3][3][3];      // [P-row slot][by + 1][bx + 1]; slot of P row pb = pb mod 3 (static per unrolled step)
    f32x4 A[3][CG];          // f0 operands of the three P rows
    auto load_A = [&](f32x4* dst, int pb) {
        const bool ok = a_in && pb >= pb0 && pb < pb1 && 4 * pb + mrow < a.H;
        const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g)
            dst[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)(ok ? vo : CVM_OOB), g * 64, CVM_F0_AUX));
    };

    // ---- epilogue of a complete P row (slot sl), as micro-operations:
    //   e1(i)   value i = (byi, bxi, r) of the 36 the lane holds: leaky-relu -> stage (|dx| > 4: dump area through
    //           the address, |dy| > 4: dump area by select)
    //   e2(i)   i < 6: one 16-byte store instruction of the copy-out; 6: the concat copy from the f0 registers;
    //           7: channel 80 when the padding channels are not the kernel's to write (!PAD)
    // Always executed (rows outside the segment store nothing: out-of-range offsets) -- the compiler counts vmcnt along
    // every path, and a branch around these stores would make each wait for a gathered corner wait for the stores too.
    // e1(t): tile t = (byi, bxi) of the 9 the lane holds -> stage, raw sums (the activation is applied to the 16-byte
    //        quads of the copy-out: 24 values per lane there against 36 here).  |dx| > 4: dump area through the
    //        address; |dy| > 4: the lane sits the tile out.
    auto e1 = [&](auto sl_c, auto t_c) {
        constexpr int sl = decltype(sl_c)::value, tile = decltype(t_c)::value;
        constexpr int byi = tile / 3, bxi = tile % 3;
        if (byi == 1 || (byi == 0 ? vy_m_i : vy_p_i)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sxa[bxi][r][36 * byi] = acc[sl][byi][bxi][r];
        }
    };
    auto e2 = [&](auto sl_c, auto i_c, int pb) {
        constexpr int sl = decltype(sl_c)::value, i = decltype(i_c)::value;
        const bool pv = pb >= pb0;                                      // uniform (pb < pb1 always)
        const int ylim = pv ? a.H - 4 * pb : 0;                         // rows of this block inside the image
        const unsigned base = (unsigned)((4 * pb * a.W + x0 + 4 * wave) * a.out_cs) * 4u;
        if constexpr (i < 6) {
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + (i * 64 + lane < 16 * 21 ? (i * 64 + lane) * 4 : 0));
            // leaky-relu max(x, slope * x): ONE v_max_f32 per value (fmaxf costs a second one that quiets a possible
            // signalling NaN first; fmed3 with +inf is folded back into fmaxf).  The multiply in front is
            // compiler-visible and reads the same registers.
            const f32x4 sv = v * a.slope;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y;
                asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(v[k]), "v"(sv[k]));
                v[k] = y;
            }
            const bool ok = i * 64 + lane < 84 * ylim && !(ABL & 4);    // 84 items per block row
            const unsigned vo = ok ? base + co_rel[i] : CVM_OOB;        // out-of-range + base stays out of range
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, v), ro, (int)vo, 0, CVM_STORE_AUX);
        } else if constexpr (i == 6) {
            const bool ok = a_in && pv && mrow < ylim && !(ABL & 4);
            const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_copy_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
            for (int g = 0; g < CG; ++g)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, A[sl][g]), rc, (int)(ok ? vo : CVM_OOB), g * 64, CVM_COPY_AUX);
        } else if constexpr (!PAD) {
            const float x = stg[(lane & 15) * G::SROW + 80];
            const float v = pwc_lrelu(x, a.slope);
            const bool ok = (lane >> 2) < ylim && !(ABL & 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, (int)(ok ? base + c80_rel : CVM_OOB), 0, CVM_STORE_AUX);
        }
    };

    // ---- one vertical block offset of a step: 4 CG chunks of 3 MFMAs (P row in slot `sl` against Q-row image
    // `img`; its three tiles start from zero: every tile row of a P row is first touched by exactly one group), and
    // after each chunk the share [c n / NCH, (c+1) n / NCH) of the group's n side micro-operations -- they issue
    // while the matrix pipe works (an MFMA occupies it for 32 cycles, the issuing wave for 4).  sched_barrier pins
    // the interleave: the compiler would otherwise cluster the MFMAs.
    auto group = [&](auto sl_c, auto byi_c, const f32x4* img, bool on, auto nside_c, auto&& side) {
        constexpr int sl = decltype(sl_c)::value, byi = decltype(byi_c)::value, NS = decltype(nside_c)::value;
        constexpr int NCH = 4 * CG;
        const bool run = on && !(ABL & 1);                              // uniform
        if (!run) {
            // Q row without pixels / P row outside the segment (fill and tail steps): zero tiles, side work only
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) acc[sl][byi][bx] = f32x4{0.f, 0.f, 0.f, 0.f};
            cvm_for<NS>(side);
            return;
        }
        f32x4 B[3], Bn[3];
#pragma unroll
        for (int bx = 0; bx < 3; ++bx) B[bx] = img[bslot + (bx - 1) * 16];
        cvm_for<NCH>([&](auto c_c) {
            constexpr int c = decltype(c_c)::value, g = c / 4, k = c % 4;
            if (k == 0 && g + 1 < CG) {
#pragma unroll
                for (int bx = 0; bx < 3; ++bx) Bn[bx] = img[(g + 1) * PLANE + bslot + (bx - 1) * 16];
            }
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) {
                const f32x4 cin = c == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[sl][byi][bx];
                acc[sl][byi][bx] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[sl][g][k], B[bx][k], cin, 0, 0, 0);
            }
            if (k == 3 && g + 1 < CG) {
#pragma unroll
                for (int bx = 0; bx < 3; ++bx) B[bx] = Bn[bx];
            }
            cvm_for<(c + 1) * NS / NCH - c * NS / NCH>([&](auto i_c) {
                side(std::integral_constant<int, c * NS / NCH + decltype(i_c)::value>{});
            });
            CVM_SCHED_BARRIER();
        });
    };

    // ---- the walk.  Step q: Q row q (image q & 1) is multiplied; Q row q+1 is gathered into the other image; the
    // corner table of Q row q+2 is built.  One fill step comes first (its tables are built in front of it).  S0 / S1 /
    // S2 = slots of the P rows q-1 / q / q+1 (static: the loop is unrolled three steps deep, nothing rotates)
    auto step = [&](auto s0_c, int q) {
        constexpr int S0 = decltype(s0_c)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;
        using IS0 = std::integral_constant<int, S0>;
        using IS1 = std::integral_constant<int, S1>;
        using IS2 = std::integral_constant<int, S2>;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        const bool mm = q >= qa && q <= qb;                             // uniform: Q row q holds pixels
        const f32x4* img = qimg + (q & 1) * BUF;
        stamp();
        load_A(A[S2], q + 1);
        if (WARP) flow_issue(q + 2, fl0, fl1);
        CVM_SCHED_BARRIER();
        stamp();

        // The P row q-1 (slot S0) leaves in three parts: its by = -1 and by = 0 tiles have been complete since the
        // previous steps and go to the stage WHILE the by = +1 group runs (24 registers free before the gather's 48
        // are taken), the by = +1 tiles follow in the next group, the copy-out in the last.
        // by = +1 (completes the P row q-1)  ||  activation -> stage, tiles 0..5  [NB >= 2: + batch 0 loads]
        cvm_wave_sync();                                                // the previous copy-out has read the stage
        group(IS0{}, I2{}, img, mm && q - 1 >= pb0, std::integral_constant<int, 6 + (NB >= 2 ? 4 : 0)>{}, [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (i < 6) e1(IS0{}, i_c);
            else if constexpr (i == 6) gi_tab(0, q + 1);
            else gi_load(i - 7);
        });
        stamp();
        // by = 0  ||  tiles 6..8, then the gather traffic of this phase
        //   NB = 1: batch 0 loads;  NB >= 2: batch 0 -> image, batch 1 loads
        group(IS1{}, I1{}, img, mm && q >= pb0 && q < pb1, std::integral_constant<int, 3 + (NB >= 2 ? 7 : 4)>{}, [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (i < 3) e1(IS0{}, std::integral_constant<int, 6 + i>{});
            else if constexpr (NB == 1) {
                if constexpr (i == 3) gi_tab(0, q + 1);
                else gi_load(i - 4);
            } else {
                if constexpr (i < 6) gc_item(0, i - 3, q + 1);
                else if constexpr (i == 6) gi_tab(1, q + 1);
                else gi_load(i - 7);
            }
        });
        stamp();
        // by = -1  ||  [NB = 3: batch 1 -> image, batch 2 loads, then] the copy-out of the P row q-1
        cvm_wave_sync();
        group(IS2{}, I0{}, img, mm && q + 1 < pb1, std::integral_constant<int, 8 + (NB == 3 ? 7 : 0)>{}, [&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr int E0 = NB == 3 ? 7 : 0;
            if constexpr (i < E0) {
                if constexpr (i < 3) gc_item(1, i, q + 1);
                else if constexpr (i == 3) gi_tab(2, q + 1);
                else gi_load(i - 4);
            } else {
                e2(IS0{}, std::integral_constant<int, i - E0>{}, q - 1);
            }
        });
        stamp();
#pragma unroll
        for (int j = 0; j < 3; ++j) gc_item(NB - 1, j, q + 1);
        if (WARP) table_write(q + 2, fl0, fl1);
        stamp();
        cvm_barrier();
    };
    if (WARP) {
        // tables of the first two Q rows
        float g0, g1;
        flow_issue(qa, fl0, fl1);
        flow_issue(qa + 1, g0, g1);
        table_write(qa, fl0, fl1);
        table_write(qa + 1, g0, g1);
    }
    cvm_barrier();
    {
        int q = qa - 1;
        for (;;) {
            step(std::integral_constant<int, 0>{}, q);
            if (++q > pb1) break;
            step(std::integral_constant<int, 1>{}, q);
            if (++q > pb1) break;
            step(std::integral_constant<int, 2>{}, q);
            if (++q > pb1) break;
        }
    }
}

// Work decomposition: strips of 16 columns, each cut into segments of `seg_brows` block rows (4 image rows
// each).  A segment costs its block rows + 1 halo Q row + the fill steps; a second round of workgroups costs a
// whole segment: the split with the smallest estimated makespan on 256 CUs x `wgpc` resident workgroups.
static void cvm_plan(int N, int H, int W, int wgpc, int* nstrips, int* nseg, int* seg_brows) {
    const int ns = (W + 15) / 16, nb = (H + 3) / 4;
    const long slots = 256L * wgpc;
    long best = -1;
    int best_k = nb;
    for (int k = nb; k >= 1; --k) {
        const int segs = (nb + k - 1) / k;
        const long items = (long)N * ns * segs;
        const long rounds = (items + slots - 1) / slots;
        const long cost = rounds * (k + 4);
        if (best < 0 || cost < best) { best = cost; best_k = k; }
    }
    *nstrips = ns; *seg_brows = best_k; *nseg = (nb + best_k - 1) / best_k;
}

static bool cvm_eligible(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow, int flow_cs,
                         const float* out, int out_cs, const float* f0_copy, int f0_copy_cs, int H, int W, int C, int R) {
    if (R != 4 || !(C == 32 || C == 64 || C == 96)) return false;
    if ((f0_cs & 3) || (f1_cs & 3) || (out_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1) || !pwc_aligned16(out)) return false;
    if (f0_copy && ((f0_copy_cs & 3) || !pwc_aligned16(f0_copy))) return false;
    if (flow && (reinterpret_cast<uintptr_t>(flow) & 3u)) return false;
    // buffer resources are per image: byte extents must stay below 2^31 (the out-of-range marker)
    const long px = (long)H * W;
    if (px * f0_cs * 4 >= (1L << 31) || px * f1_cs * 4 >= (1L << 31) || px * out_cs * 4 >= (1L << 31)) return false;
    if (f0_copy && px * f0_copy_cs * 4 >= (1L << 31)) return false;
    if (flow && px * flow_cs * 4 >= (1L << 31)) return false;
    return true;
}

template <int CG, bool WARP, bool PAD>
static int cvm_launch_t(CvmArgs& a, hipStream_t s) {
    using G = CvmGeom<CG>;
    const size_t lds = (size_t)G::LDS_F * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_mfma_kernel<CG, WARP, PAD, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    cvm_plan(a.N, a.H, a.W, G::WGPC, &a.nstrips, &a.nseg, &a.seg_brows);
    const long items = (long)a.N * a.nstrips * a.nseg;
    if (items >= (1L << 31)) return PWC_ERANGE;
    hipLaunchKernelGGL((cost_volume_mfma_kernel<CG, WARP, PAD, 0>), dim3((unsigned)items), dim3(G::T), lds, s, a);
    return pwc_launch_status();
}

static int cvm_launch(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow, int flow_cs,
                      float flow_scale, float* out, int out_cs, int pad_ok, float* f0_copy, int f0_copy_cs, int N, int H,
                      int W, int C, float slope, hipStream_t s) {
    CvmArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.flow_scale = flow_scale; a.slope = slope;
    a.inv_c = 1.0f / (float)C;               // reduce_mean: x * (1/C), within 1 ulp of x / C
    a.nbrows = (H + 3) / 4;
    a.pad_ok = pad_ok; a.dbg = nullptr;
#define CVM_CASE(CGV)                                                                          \
    case CGV * 16:                                                                             \
        return flow ? (pad_ok ? cvm_launch_t<CGV, true, true>(a, s) : cvm_launch_t<CGV, true, false>(a, s))         \
                    : (pad_ok ? cvm_launch_t<CGV, false, true>(a, s) : cvm_launch_t<CGV, false, false>(a, s));
    switch (C) {
        CVM_CASE(2)
        CVM_CASE(4)
        CVM_CASE(6)
        default: return PWC_EUNSUPPORTED;
    }
#undef CVM_CASE
}
