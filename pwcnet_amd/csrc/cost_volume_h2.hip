// cost_volume_h2.hip -- warp + cost volume (+ the optional f0 concat copy) in ONE launch, correlation on the F16 matrix pipe,
// for gfx950 (round 5).  Search range 4, C = 32 / 64 / 96.  Same operands, same results contract and the same strip / block
// structure as cost_volume_mfma.hip (reference model.py:105-112, modules.py:99-137,158-204,264) -- what changes is the
// arithmetic of the dot products and, above all, WHEN things happen:
//
//   f1w[n,y,x,:]             = bilinear_warp(f1, flow * flow_scale)          (fp32, never written to memory)
//   out[n,y,x,(v+4)*9+(h+4)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  f1w zero outside
//   f0_copy[n,y,x,0:C]       = f0[n,y,x,0:C]                                  (optional)
//
// Arithmetic.  Every operand (f0; f1w after its fp32 blend) is used as the two-term fp16 split of conv3x3_h2.hip,
// x = h + 2^-11 m' (pwc_split2), and a tile of 16 P pixels x 16 Q pixels over 32 channels is three v_mfma_f32_16x16x32_f16:
// cross = AH x BM' + AM' x BH, hh = AH x BH, result hh + 2^-11 cross (fp32 accumulators).  27 matrix instructions of 16 cycles
// per 4 x 4-pixel block at C = 32 where the fp32 form has 72 of 32 cycles, and an F16-pipe instruction does not
// keep the SIMD's vector instructions out.  Range of the split: |x| < 65504 (beyond: NaN outputs; PWC_STATUS_NONFINITE).
//
// Organisation.  A workgroup is EIGHT waves, one workgroup per CU:
//   waves 0-3  "consumers", one per 4-pixel block column of the 16-column strip: matrix instructions, stage, copy-out.  A P block
//              row is computed in ONE step against the three Q rows it meets, which all sit in LDS (ring of three Q-row images):
//              9 accumulator tiles = 36 registers -- no three generations of accumulators (108 registers in cost_volume_mfma.hip),
//              no unrolled slot rotation.  The 18 operand reads of a step (C = 32) go out together at its top.  The f0 rows are
//              requested two steps before they are split.
//   waves 4-7  "producers": nothing but the gather -- flows, corner tables, the corner requests of Q row p+3 in step p, blend +
//              split + Q-row image of row p+2 (requested a step earlier), so a trip to memory has a whole step.
// Per step p two barriers: A (every consumer has read the image of Q row p-1: the producers may overwrite it with row p+2) and B
// (row p+2 and the next table are complete).  Three fill steps (p = pb0-3 .. pb0-1) run with the tile work switched off.
//
// What was measured on the way (batch 8, cold operands, iid N(0, 3^2) px flows; profiles/r05_exp_cv*.txt), level 4 / 3 / 2, us:
//   cost_volume_mfma.hip (fp32 MFMA), with / without the f0 copy        46.7 / 40.5    31.4 / 28.0    21.4 / 19.5
//   the same kernel with F16-pipe products only                          44.5 / 38.4    26.9 / 24.7    19.0 / 17.8
//   ring of Q images, one step per P row, gathers a step ahead (4 waves) 46.3 / 36.5    25.9 / 24.3    19.9 / 18.8
//   + producer / consumer waves (this file)                              47.6 / 33.4    22.6 / 20.2    17.5 / 16.8
// and what did NOT move it: the corners requested three steps ahead instead of one (34.6 at level 4), the copy-out given to the
// producers (37.2), the f0 operands fetched and split by the producers and handed over through LDS (34.6), lane-dependent
// addresses instead of execution-mask branches in the stage writes.  The skeleton without corner requests and stores takes 21-23
// us at level 4 in EVERY form -- about 3000 cycles per step for the ~250 instructions of a consumer: the step is a chain of LDS
// round trips and dependent matrix / vector instructions at one consumer wave per SIMD, and the memory system's own floor for this
// traffic pattern (58.7 MB of 128-byte reads + 77 MB written as 336 bytes of every 512 / 640: scripts/exp_membw.hip) is 25 - 32 us.
// One thing the compiler forces: a wave that waits for ANY load waits for ALL of its stores (loads and stores share one counter and
// the waits are not counted across the two kinds) -- the consumers' f0 wait drains their copy-out stores once per step.
//
// Lane roles, LDS layouts, the stage and the copy-out are those of cost_volume_mfma.hip (see there); the Q-row image holds, per
// 32 channels, a plane of h slots [row 4][pixel 24][quad kq: 8 fp16 = channels 4 kq..+3 and 16 + 4 kq..+3] and a plane of m'
// slots: the bytes and the bank pattern of the fp32 image.
#pragma once
#include "cost_volume_mfma.hip"

template <int CG>
struct CvhGeom {
    using M = CvmGeom<CG>;
    static constexpr int RING = 3;                                      // Q-row images
    static constexpr int TRING = 3;                                     // corner tables: row r's is written in step r-4, read in steps r-3 (requests) and r-2 (blend)
    static constexpr int LDS_F = RING * M::BUF * 4 + M::NW * M::WSTG + TRING * M::TAB;
    static_assert(LDS_F * 4 <= 160 * 1024, "does not fit the LDS");
};

// ABL (scripts/exp_cv5.hip only; 0 in the library): 1 = no MFMAs, 2 = no gather loads, 4 = no stores, 8 = s_memtime stamps
// FLOWPAD (round 6; needs WARP and PAD): channels 81, 82 of every `out` record receive the pixel's flow (x, y) as read -- the record
// is then [cv 81 | flows_up_prev 2 | 0] of the estimator's concat (reference modules.py:261-264: the flow keeps its LOGICAL place
// behind features_0, the first conv's packed weights follow the physical order), `out` can be a dense tensor of 84-channel records
// (336 contiguous bytes per pixel: no record-strided writes) and the estimator's first conv reads it as one of three operands.
template <int CG, bool WARP, bool PAD, int ABL = 0, bool FLOWPAD = false>
__global__ __launch_bounds__(512, 1) void cost_volume_h2_kernel(const CvmArgs a) {
    static_assert(!FLOWPAD || (WARP && PAD), "the flow rides in the padding channels of a warping launch");
    using G = CvmGeom<CG>;
    using GH = CvhGeom<CG>;
    constexpr int RS = G::RS, PLANE = G::PLANE, BUF = G::BUF, ITEMS = G::ITEMS, NP = CG / 2;
    constexpr bool KEEP = CG <= 2;                                      // per-item constants live in registers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* qimg = reinterpret_cast<f32x4*>(smem);                       // ring of 3 Q-row images of BUF slots
    float* stg_all = smem + GH::RING * BUF * 4;
    float* tabf = stg_all + G::NW * G::WSTG;                            // ring of corner tables of TAB dwords

    const int t = threadIdx.x & 255, lane = t & 63;                     // thread index inside its role
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);            // block column of the strip (consumers)
    const bool consumer = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0;      // wave-uniform
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

    // ---- lane roles (cost_volume_mfma.hip): MFMA operands: lane = (block pixel m = lane & 15 -> row m >> 2, column m & 3;
    // channel quad kq = lane >> 4)
    const int mrow = (lane & 15) >> 2, mcol = lane & 3, kq = lane >> 4;
    const int ax = x0 + 4 * wave + mcol;                                // image column of this lane's f0 pixel
    const int bslot = mrow * RS + (4 * (wave + 1) + mcol) * 4 + kq;     // B operand slot of block column offset 0
    const bool a_in = ax < a.W;
    const unsigned a_rel = (unsigned)(mrow * a.W + ax);

    // D fragment: lane holds P pixels (row kq, column r = 0..3) x Q pixel (row mrow, column mcol).  Stage address of
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
    // stage padding channels 81..83 stay zero for the whole launch
    if (lane < 48) stg[(lane / 3) * G::SROW + 81 + (lane % 3)] = 0.f;

    // ---- gather items of a Q row: item e = i * 256 + t -> (pixel, plane g, quad): the C/4 quads of a pixel sit in consecutive
    // lanes, so one instruction asks for whole 128-byte lines of a corner pixel
    auto item_decode = [&](int i, unsigned& chan, int& tab_d, int& img_b) {
        int tt = t;
        if (!KEEP) asm volatile("" : "+v"(tt));                         // recomputed per use: too many to keep
        const int e = i * 256 + tt;
        const int pix = (CG == 2) ? (e >> 3) : (CG == 4) ? (e >> 4) : ((e * 2731) >> 16);   // e / (C/4); e / 24 for e < 2304
        const int cq = e - pix * (CG * 4);
        const int g = cq >> 2, kqi = cq & 3;
        const int r = (pix * 2731) >> 16;                               // pix / 24 for pix < 96
        const int xi = pix - r * 24;
        chan = (unsigned)(g * 64 + kqi * 16);                           // byte offset of the channel quad
        tab_d = pix * 8;                                                // table entry (dwords)
        img_b = ((g >> 1) * 2 * PLANE + r * RS + xi * 4 + kqi) * 16 + (g & 1) * 8;   // byte offset of the item's 8 bytes of h
    };
    unsigned k_chan[KEEP ? ITEMS : 1];
    int k_tab[KEEP ? ITEMS : 1], k_img[KEEP ? ITEMS : 1];
    if (KEEP) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) item_decode(i, k_chan[i], k_tab[i], k_img[i]);
    }
    auto item = [&](int i, unsigned& chan, int& tab_d, int& img_b) {
        if (KEEP) { chan = k_chan[i]; tab_d = k_tab[i]; img_b = k_img[i]; }
        else item_decode(i, chan, tab_d, img_b);
    };
    // !WARP: pixel of an item relative to the image, or out of range
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

    // ---- gather of a Q row: request (g_issue, table slot ts) and, a step later, blend + split -> image slot (g_commit)
    f32x4 gv[ITEMS][WARP ? 4 : 1];
    auto g_issue = [&](int qq, int ts) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            unsigned chan;
            int tab_d, img_b;
            item(i, chan, tab_d, img_b);
            cvm_u32x4 off;
            if (WARP) off = *reinterpret_cast<const cvm_u32x4*>(tabf + ts * G::TAB + tab_d);
            else off[0] = nowarp_off(tab_d, qq);
#pragma unroll
            for (int c = 0; c < (WARP ? 4 : 1); ++c) {
                const unsigned vo = (ABL & 2) ? CVM_OOB : off[c] + chan;           // out-of-range + chan stays out of range
                gv[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, (int)vo, 0, 0));
            }
        }
    };
    auto g_commit = [&](int ts, int is) {
        char* image = reinterpret_cast<char*>(qimg + is * BUF);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            unsigned chan;
            int tab_d, img_b;
            item(i, chan, tab_d, img_b);
            f32x4 v;
            if (WARP) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(tabf + ts * G::TAB + tab_d + 4);
                // modules.py:132-135: c00*x00 + c01*x01 + c10*x10 + c11*x11, summed left to right.  (The 1/C of the mean is NOT
                // folded into the weights here as the fp32 kernel does: a feature of 1e-3 over C = 64 would be split in fp16's
                // subnormal range and lose bits; it multiplies the finished sum in the copy-out.)
                v = w[0] * gv[i][0];
                v = __builtin_elementwise_fma(f32x4{w[1], w[1], w[1], w[1]}, gv[i][1], v);
                v = __builtin_elementwise_fma(f32x4{w[2], w[2], w[2], w[2]}, gv[i][2], v);
                v = __builtin_elementwise_fma(f32x4{w[3], w[3], w[3], w[3]}, gv[i][3], v);
            } else {
                v = gv[i][0];
            }
            pwc_f16x4 h, m;
            pwc_split4(v, h, m);
            *reinterpret_cast<pwc_f16x4*>(image + img_b) = h;
            *reinterpret_cast<pwc_f16x4*>(image + img_b + PLANE * 16) = m;
        }
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
    auto table_write = [&](int qq, int ts, float f0v, float f1v) {
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
            const f32x4 w = {(fy1 - fy) * (fx1 - fx), (fy1 - fy) * (fx - fx0), (fy - fy0) * (fx1 - fx), (fy - fy0) * (fx - fx0)};
            const unsigned cs4 = (unsigned)a.f1_cs * 4u;
            cvm_u32x4 off = {(unsigned)(iy0 * a.W + ix0) * cs4, (unsigned)(iy0 * a.W + ix1) * cs4,
                             (unsigned)(iy1 * a.W + ix0) * cs4, (unsigned)(iy1 * a.W + ix1) * cs4};
            if (!ok) off = cvm_u32x4{CVM_OOB, CVM_OOB, CVM_OOB, CVM_OOB};
            float* e = tabf + ts * G::TAB + t * 8;
            *reinterpret_cast<cvm_u32x4*>(e) = off;
            *reinterpret_cast<f32x4*>(e + 4) = w;
        }
    };

    // ---- f0 operand of a P row: global -> registers in MFMA layout; split (and copied out) one step later
    f32x4 A2[2][CG];                                                    // raw rows in flight: requested TWO steps ahead
    pwc_f16x8 AH[NP], AM[NP];
    auto load_A = [&](f32x4* A, int pb) {
        const bool ok = a_in && pb >= pb0 && pb < pb1 && 4 * pb + mrow < a.H;
        const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g)
            A[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)(ok ? vo : CVM_OOB), g * 64, CVM_F0_AUX));
    };
    // (k slot e of lane quarter kq = channel 32 j + 4 kq + e for e < 4, 32 j + 16 + 4 kq + e - 4 else: what the image holds)
    auto split_A = [&](const f32x4* A, int pb) {
        const bool ok = a_in && pb >= pb0 && pb < pb1 && 4 * pb + mrow < a.H && !(ABL & 4);
        const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.f0_copy_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, A[g]), rc, (int)(ok ? vo : CVM_OOB), g * 64, CVM_COPY_AUX);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            pwc_f16x4 h0, m0, h1, m1;
            pwc_split4(A[2 * j], h0, m0);
            pwc_split4(A[2 * j + 1], h1, m1);
            AH[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            AM[j] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };

    // ---- the three tiles of one vertical block offset: per 32 channels cross = AH x BM' + AM' x BH, hh = AH x BH
    f32x4 acc[3][3];                                                    // [by + 1][bx + 1]
    // B operands of one vertical offset (image slot `is`): reads(); its nine (C = 32) matrix instructions: mfmas().  The three
    // images a step reads are complete when it starts, so the reads of a later group can be in flight under an earlier one's
    // instructions (C = 32: all eighteen at the top of the step).
    pwc_f16x8 Lh[3][NP][3], Lm[3][NP][3];
    auto reads = [&](auto byi_c, int is) {
        constexpr int byi = decltype(byi_c)::value;
        const pwc_f16x8* imgh = reinterpret_cast<const pwc_f16x8*>(qimg + is * BUF);
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) {
                Lh[byi][j][bx] = imgh[(2 * j) * PLANE + bslot + (bx - 1) * 16];
                Lm[byi][j][bx] = imgh[(2 * j + 1) * PLANE + bslot + (bx - 1) * 16];
            }
    };
    auto mfmas = [&](auto byi_c) {
        constexpr int byi = decltype(byi_c)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (ABL & 1) {
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) acc[byi][bx] = zero;
            return;
        }
        f32x4 hh[3], xx[3];
        cvm_for<NP>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) xx[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH[j], Lm[byi][j][bx], j == 0 ? zero : xx[bx], 0, 0, 0);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) hh[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH[j], Lh[byi][j][bx], j == 0 ? zero : hh[bx], 0, 0, 0);
#pragma unroll
            for (int bx = 0; bx < 3; ++bx) xx[bx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AM[j], Lh[byi][j][bx], xx[bx], 0, 0, 0);
        });
#pragma unroll
        for (int bx = 0; bx < 3; ++bx)
            acc[byi][bx] = __builtin_elementwise_fma(xx[bx], f32x4{1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f}, hh[bx]);
    };
    // ---- the nine tiles -> stage, raw sums (the activation is applied to the 16-byte quads of the copy-out)
    auto to_stage = [&]() {
#pragma unroll
        for (int byi = 0; byi < 3; ++byi)
            if (byi == 1 || (byi == 0 ? vy_m_i : vy_p_i)) {
#pragma unroll
                for (int bxi = 0; bxi < 3; ++bxi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sxa[bxi][r][36 * byi] = acc[byi][bxi][r];
            }
    };
    // ---- copy-out of a P row: the six stage reads together (one LDS round trip), then activation and store per quad.  Always
    // executed (rows outside the segment store nothing: out-of-range offsets): a branch around stores would make every wait
    // for a load behind it a wait for the stores (the compiler counts along every path).
    auto copy_out = [&](int pb) {
        const bool pv = pb >= pb0;                                      // uniform (pb < pb1 always)
        const int ylim = pv ? a.H - 4 * pb : 0;                         // rows of this block inside the image
        const unsigned base = (unsigned)((4 * pb * a.W + x0 + 4 * wave) * a.out_cs) * 4u;
        f32x4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const f32x4*>(stg + (i * 64 + lane < 16 * 21 ? (i * 64 + lane) * 4 : 0));
        float x80 = 0.f;
        if constexpr (!PAD) x80 = stg[(lane & 15) * G::SROW + 80];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // mean = sum * (1/C) (reduce_mean, modules.py:181), then leaky-relu max(x, slope * x): ONE v_max_f32 per value (fmaxf
            // costs a second one that quiets a possible signalling NaN)
            const f32x4 mv = v[i] * a.inv_c;
            const f32x4 sv = mv * a.slope;
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float yk;
                asm("v_max_f32 %0, %1, %2" : "=v"(yk) : "v"(mv[k]), "v"(sv[k]));
                y[k] = yk;
            }
            if constexpr (FLOWPAD) {                                    // channels 81, 82 carry the flow as read: no mean, no activation
                const bool f = (q20 >> i) & 1u;
                y[1] = f ? v[i][1] : y[1];
                y[2] = f ? v[i][2] : y[2];
            }
            const bool ok = i * 64 + lane < 84 * ylim && !(ABL & 4);    // 84 items per block row
            const unsigned vo = ok ? base + co_rel[i] : CVM_OOB;        // out-of-range + base stays out of range
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, y), ro, (int)vo, 0, CVM_STORE_AUX);
        }
        if constexpr (!PAD) {
            const float y = pwc_lrelu(x80 * a.inv_c, a.slope);
            const bool ok = (lane >> 2) < ylim && !(ABL & 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ro, (int)(ok ? base + c80_rel : CVM_OOB), 0, CVM_STORE_AUX);
        }
    };

    // ---- FLOWPAD: the flow of the P row's 16 pixels (lanes 0-15: one pixel each), requested a step before its copy-out
    float flx[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    auto load_flow = [&](float* f, int pb) {
        const bool ok = lane < 16 && a_in && pb >= pb0 && pb < pb1 && 4 * pb + mrow < a.H;
        const unsigned vo = ok ? (a_rel + (unsigned)(4 * pb * a.W)) * (unsigned)(a.flow_cs * 4) : CVM_OOB;
        f[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 0, 0));
        f[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 4, 0));
    };
    auto flow_to_stage = [&](const float* f) {
        if (lane < 16) {
            stg[lane * G::SROW + 81] = f[0];
            stg[lane * G::SROW + 82] = f[1];
        }
    };

    auto slot = [](int r) { return ((r % 3) + 3) % 3; };                // ring slot of Q row r (images and tables alike)
    if (consumer) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
#pragma unroll
        for (int j = 0; j < NP; ++j) { AH[j] = pwc_f16x8{}; AM[j] = pwc_f16x8{}; }
        int s_m = slot(pb0 - 4), s_0 = slot(pb0 - 3), s_p = slot(pb0 - 2);          // ring slots of Q rows p-1, p, p+1
        // one step; PAR = which of the two f0 buffers receives row p+2 (the other one holds row p+1, requested a step ago)
        auto cstep = [&](auto par_c, int p) {
            constexpr int PAR = decltype(par_c)::value;
            const bool real = p >= pb0;                                 // uniform: fill steps compute nothing
            stamp();
            load_A(A2[PAR], p + 2);
            if constexpr (FLOWPAD) load_flow(flx[PAR], p + 1);
            if (real) {
                reads(I0{}, s_m);
                if constexpr (NP == 1) { reads(I1{}, s_0); reads(I2{}, s_p); }
            }
            stamp();
            cvm_barrier();                                              // A: the image of Q row p-1 has been read
            stamp();
            if (real) {
                if constexpr (NP > 1) reads(I1{}, s_0);
                mfmas(I0{});
                if constexpr (NP > 1) reads(I2{}, s_p);
                mfmas(I1{});
                mfmas(I2{});
                cvm_wave_sync();                                        // the previous copy-out has read the stage
                to_stage();
            }
            stamp();
            split_A(A2[1 - PAR], p + 1);                                // (+ its concat copy: stores)
            if constexpr (FLOWPAD) flow_to_stage(flx[1 - PAR]);        // (row p's, requested in step p - 1; behind the wave sync above
                                                                        // when the step is real, nothing reads the stage otherwise)
            stamp();
            cvm_wave_sync();
            // Always executed (rows outside the segment store nothing: out-of-range offsets)
            copy_out(p);
            stamp();
            cvm_barrier();                                              // B: Q row p+2 and the table of row p+4 are complete
            stamp();
            stamp();
            const int s_n = s_m; s_m = s_0; s_0 = s_p; s_p = s_n;      // (the freed image is Q row p+2's)
        };
        cvm_barrier();                                                  // (the producers' first tables)
        load_A(A2[1], pb0 - 2);                                         // (no such row: zeros)
        for (int p = pb0 - 3; p < pb1; p += 2) {
            cstep(I0{}, p);
            if (p + 1 < pb1) cstep(I1{}, p + 1);
        }
    } else {
        // tables of Q rows pb0-1 and pb0, the corners of row pb0-1 requested; then per step: flow of row p+4, [A], blend of row
        // p+2, requests of row p+3, table of row p+4, [B]
        if (WARP) {
            float g0, g1;
            flow_issue(pb0 - 1, fl0, fl1);
            flow_issue(pb0, g0, g1);
            table_write(pb0 - 1, slot(pb0 - 1), fl0, fl1);
            table_write(pb0, slot(pb0), g0, g1);
        }
        cvm_barrier();
        g_issue(pb0 - 1, slot(pb0 - 1));
        for (int p = pb0 - 3; p < pb1; ++p) {
            if (WARP) flow_issue(p + 4, fl0, fl1);
            cvm_barrier();                                              // A
            g_commit(slot(p + 2), slot(p - 1));                         // Q row p+2 (requested a step ago) -> the freed image
            g_issue(p + 3, slot(p + 3));
            if (WARP) table_write(p + 4, slot(p + 4), fl0, fl1);
            cvm_barrier();                                              // B
        }
    }
}

template <int CG, bool WARP, bool PAD, bool FLOWPAD = false>
static int cvh_launch_t(CvmArgs& a, hipStream_t s) {
    using GH = CvhGeom<CG>;
    const size_t lds = (size_t)GH::LDS_F * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_h2_kernel<CG, WARP, PAD, 0, FLOWPAD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    cvm_plan(a.N, a.H, a.W, 1, &a.nstrips, &a.nseg, &a.seg_brows);
    const long items = (long)a.N * a.nstrips * a.nseg;
    if (items >= (1L << 31)) return PWC_ERANGE;
    hipLaunchKernelGGL((cost_volume_h2_kernel<CG, WARP, PAD, 0, FLOWPAD>), dim3((unsigned)items), dim3(512), lds, s, a);
    return pwc_launch_status();
}

static int cvh_launch(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow, int flow_cs,
                      float flow_scale, float* out, int out_cs, int pad_ok, float* f0_copy, int f0_copy_cs, int N, int H,
                      int W, int C, float slope, hipStream_t s) {
    CvmArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.flow_scale = flow_scale; a.slope = slope;
    a.inv_c = 1.0f / (float)C;               // reduce_mean: x * (1/C), within 1 ulp of x / C
    a.nbrows = (H + 3) / 4;
    a.pad_ok = pad_ok; a.dbg = nullptr;
#define CVH_CASE(CGV)                                                                          \
    case CGV * 16:                                                                             \
        if (flow && pad_ok == 2) return cvh_launch_t<CGV, true, true, true>(a, s);                                  \
        return flow ? (pad_ok ? cvh_launch_t<CGV, true, true>(a, s) : cvh_launch_t<CGV, true, false>(a, s))         \
                    : (pad_ok ? cvh_launch_t<CGV, false, true>(a, s) : cvh_launch_t<CGV, false, false>(a, s));
    switch (C) {
        CVH_CASE(2)
        CVH_CASE(4)
        CVH_CASE(6)
        default: return PWC_EUNSUPPORTED;
    }
#undef CVH_CASE
}
