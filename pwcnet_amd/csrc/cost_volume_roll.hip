// cost_volume_roll.hip -- rolling-window cost volume for gfx950 (search range 4, all channels of a
// pixel resident in LDS: C = 32, the full-resolution pyramid level that holds 65 % of the
// correlation bytes of a PWC-Net forward).
//
// Replaces CostVolumeLayer.__call__ (reference modules.py:158-204) and, optionally, the
// `features_0` operand of the estimator input's tf.concat (modules.py:264):
//
//   out[n,y,x,(v+4)*9+(h+4)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  zero outside
//   f0_copy[n,y,x,0:C]       = f0[n,y,x,0:C]                                            (optional)
//
// Why a second kernel: the tile kernel (cost_volume.hip) loads a (4+8) x (64+8) halo per 4 x 64
// output tile, i.e. every f1w row travels L2 -> LDS three times and the LDS-DMA stream is 2.2x the
// input bytes (54 us for the 133 MB of the 112x256 level = 31 % of HBM peak).  Here a persistent
// workgroup owns a 32-column STRIP SEGMENT and walks DOWN it four rows at a time:
//
//   ring   16 f1w rows x 40 pixels, as two 16-channel PLANES of 64-byte pixel records, filled by
//          buffer_load_dwordx4 ... lds (1 KiB per instruction, out-of-image pixels = the zeros of the
//          buffer range check).  A step needs window rows [4s, 4s+12) of the segment; the 4 rows of
//          step s+1 are fetched into the 4 slots that expired at step s-1 while step s is computed:
//          every f1w row is fetched once per segment (x1.25 horizontal halo, x(rows+8)/rows vertical
//          halo at segment starts).
//   f0     double-buffered 4 x 32 pixel tile (same two-plane layout, same DMA path); also the source
//          of the concat copy.
//   lanes  wave = horizontal shift h (9 waves); lane = (strip column x, channel half): the lane owns
//          the 4 pixels of column x in this step x all 9 vertical shifts on ITS 16 channels -- 36
//          accumulators.  Per channel quad it reads its column of the 12 window rows once (12
//          ds_read_b128, each feeding up to 4 pixels) + 4 f0 quads, for 144 FMAs: 9 FMAs per LDS
//          read (the first version -- 2 pixels x 3 x 3 shifts -- had 5 and was LDS-bound: 39 k LDS
//          cycles per CU and segment).  The two channel halves are added with v_permlane32_swap.
//   banks  64-byte records: lane x visits the 4 quads of its plane in the rotated order
//          (sq + (x >> 2)) mod 4 -- in every ds_read_b128 service group the 4 pixel residues mod 4 x 4
//          distinct quads cover the 16 slots of a bank row once, for every shift.
//   out    accumulators -> LDS stage (82-float pixel records: ds_write_b32 2-way = free) -> 16-byte
//          buffer stores of contiguous 324-byte records; pixels beyond the image edge are dropped by
//          the range check (no divergent branches: every wave issues a FIXED number of stores).
//   sync   every wave issues 4 of a step's 36 DMA pieces at the start of the step and its copy-out
//          stores at the end; `s_waitcnt vmcnt(#stores)` then waits for exactly the (older) DMA
//          pieces -- loads and stores of a wave retire in issue order on gfx9 -- and the stores are
//          never waited for.
//
// Algorithmic bytes: N*H*W*(2C+81)*4; HBM-bound (8.9 flop/B).
//
// Measured (8 x 112 x 256 x 32, 133 MB; scripts/exp_cv2.hip, operands rotated through > 256 MB): 47 us against the
// tile kernel's 62 us; inside the forward (operands partly in the Infinity Cache) 40.7 us against 54-60 us.  What
// bounds it now: the workgroup moves 76.6 MB of LDS-DMA + 74 MB of stores for the 133 algorithmic MB, and the
// steady state (reads and writes mixed, all 256 workgroups in step) runs at ~3.8 TB/s of that traffic, the
// 12-row prologue (reads only) at 4.5 TB/s, the last copy-out with nothing behind it.  The fp32 VALU is the
// second wall: scripts/exp_valu.hip measures 3.3-4.5 cycles per wave64 v_fma_f32 (82-95 TFLOP/s) and 5.9-8.1
// per v_pk_fma_f32 (up to 107 TFLOP/s) -- 1.19 GFMA cannot take less than ~13 us of VALU issue, bookkeeping
// not counted.  A matrix-pipe form (4 x 4-pixel blocks: the +-4 window is an exact 3 x 3 grid of 16-pixel
// N-blocks, 56 % of the products useful, LDS reads 10x fewer) was built and measures the same 46-49 us
// (scripts/exp_cost_volume_roll_mfma.hip): with the arithmetic out of the way the memory system is the bound.
#pragma once
#include "pwc_common.h"

struct CvRollArgs {
    const float* f0;
    const float* f1;
    float* out;
    float* f0_copy;       // null: no concat copy
    int f0_cs, f1_cs, out_cs, f0_copy_cs;
    int N, H, W;
    float slope;
    int nstrips, nseg, seg_rows;
    long long* dbg;      // scripts/exp_cv2.hip only (ABL & 8): per-phase s_memtime stamps of workgroup 0
};

struct CvRollGeom {
    static constexpr int C = 32, R = 4, D = 9, DD = 81;
    static constexpr int WS = 32, HW = WS + 2 * R, Q = 4, RING = 16;
    static constexpr int PROW = HW * 16;                // floats per ring row of one plane (2560 B)
    static constexpr int RPLANE = RING * PROW;          // floats per ring plane
    static constexpr int RING_F = 2 * RPLANE;
    static constexpr int F0ROW = WS * 16;               // floats per f0 tile row of one plane
    static constexpr int F0PL = Q * F0ROW;              // floats per f0 tile plane
    static constexpr int F0BUF = 2 * F0PL;              // floats per f0 tile
    static constexpr int F0_F = 2 * F0BUF;              // two tiles
    static constexpr int SROW = 82;                     // stage floats per pixel
    static constexpr int STG_F = Q * WS * SROW;
    static constexpr int LDS_F = RING_F + F0_F + STG_F;
    static constexpr int NW = 9, T = 64 * NW;
    static constexpr int G1 = Q * PROW / 256;           // DMA pieces per 4-row group and plane (10)
    static constexpr int G0 = F0PL / 256;               // DMA pieces per f0 tile plane (8)
    static constexpr int NPIECE = 2 * G1 + 2 * G0;      // 36 pieces per step
    static constexpr int PPW = NPIECE / NW;             // pieces per wave (4)
    static constexpr int NQUADS = Q * WS * 21;          // copy-out items: 20 quads + 1 scalar per pixel
    static constexpr int NST_OUT = (NQUADS + T - 1) / T;        // store instructions per wave: cost volume (5)
    static constexpr int NST_CPY = (Q * WS * 8 + T - 1) / T;    // ... concat copy (2)
    static_assert(NPIECE % NW == 0, "pieces must split evenly over the waves");
    static_assert(LDS_F * 4 <= 160 * 1024, "does not fit the LDS");
};

// LDS reads in flight per lane in the correlation loop (measured 3 / 6 / 10: no difference -- the loop is
// bound by VALU issue, not by LDS latency; 6 keeps the kernel at 147 VGPRs).
#ifndef CVR_PD
#define CVR_PD 6
#endif
#define CVR_OOB 0x80000000u
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
#define CVR_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

typedef unsigned int cvr_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cvr_u32x2 __attribute__((ext_vector_type(2)));

// a <- (a[lanes 0-31], b[lanes 0-31]), b <- (a[lanes 32-63], b[lanes 32-63]).  Inline asm (the builtin
// loses its second result in this toolchain); the s_nops are the VALU <-> permlane wait states that the
// hazard recogniser cannot add for instructions it does not see.
__device__ __forceinline__ void cvr_swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt (its fence covers
// global memory), which would wait for the prefetch DMA and for the copy-out stores at every barrier.
__device__ __forceinline__ void cvr_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ABL (scripts/exp_cv2.hip only; 0 in the library): 1 = no FMAs, 2 = no DMA, 4 = no stores, 8 = phase stamps
template <int ABL = 0>
__global__ __launch_bounds__(CvRollGeom::T) void cost_volume_roll_kernel(const CvRollArgs a) {
    using G = CvRollGeom;
    constexpr int WS = G::WS, Q = G::Q, C = G::C;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    float* f0s = smem + G::RING_F;
    float* stg = f0s + G::F0_F;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // = horizontal shift index h + 4

    const int nitems = a.N * a.nseg * a.nstrips;
    const float inv_c = 1.0f / (float)C;                 // C is a power of two: exact
    const unsigned rowb1 = (unsigned)(a.W * a.f1_cs * 4), rowb0 = (unsigned)(a.W * a.f0_cs * 4);

    int stamp_i = 0;
    auto stamp = [&]() {
        if (ABL & 8) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tk;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 8) && stamp_i < 128)
                a.dbg[(wave ? 128 : 0) + stamp_i] = (long long)tk;
        }
        ++stamp_i;
    };
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        // XCD-aware item order: the workgroups of one XCD own neighbouring strips / segments,
        // whose halos then meet in that XCD's L2
        const int id = pwc_xcd_remap(it, nitems);
        const int sx = id % a.nstrips;
        const int rest = id / a.nstrips;
        const int sg = rest % a.nseg;
        const int n = rest / a.nseg;
        const int x0 = sx * WS;
        const int Y0 = sg * a.seg_rows;
        if (Y0 >= a.H) continue;                          // uniform
        const int rows = min(a.seg_rows, a.H - Y0);
        const int nsteps = (rows + Q - 1) / Q;

        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.f1 + (size_t)n * a.H * a.W * a.f1_cs), 0, a.H * a.W * a.f1_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.f0 + (size_t)n * a.H * a.W * a.f0_cs), 0, a.H * a.W * a.f0_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.out + (size_t)n * a.H * a.W * a.out_cs), 0, a.H * a.W * a.out_cs * 4, 0x00020000);

        // ---- this wave's 4 DMA pieces of a step.  Piece id pid = p*9 + wave:
        //   pid <  20: f1 4-row group, plane pid/10, 1 KiB chunk k = pid%10 of the group's plane image
        //              (lane -> pixel (k*64+lane)/4 of the 160 group pixels, quad lane%4);
        //   pid >= 20: f0 tile, plane (pid-20)/8, chunk k = (pid-20)%8 (tile row k/2, pixel (k&1)*16 + lane/4).
        // Nothing per-lane is kept across the correlation loop for the DMA: the offsets are recomputed from
        // an opaque copy of the lane id at every issue (a few dozen VALU instructions per step; holding them,
        // or letting the compiler turn them into loop-carried induction variables, spilled accumulators).
        // yb1: image row of the f1 group's first row, slot4: its ring slot (multiple of 4); yb0 / buf: f0 tile
        auto issue_pieces = [&](int yb1, int slot4, int yb0, int buf, bool with_f0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int p = 0; p < G::PPW; ++p) {
                const int pid = p * G::NW + wave;                 // uniform
                if (pid < 2 * G::G1) {
                    const int plane = pid / G::G1, k = pid - plane * G::G1;
                    const int pxi = (k * 64 + ln) >> 2;
                    const int i = pxi / G::HW, px = pxi - i * G::HW;
                    const int xx = x0 - 4 + px, y = yb1 + i;
                    const bool ok = (unsigned)xx < (unsigned)a.W && (unsigned)y < (unsigned)a.H;
                    const unsigned vo = ok ? (unsigned)y * rowb1 + (unsigned)((xx * a.f1_cs + plane * 16 + (ln & 3) * 4) * 4) : CVR_OOB;
                    if (!(ABL & 2))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lptr_t)(smem + plane * G::RPLANE + k * 256 + slot4 * G::PROW), 16,
                                                                 (int)vo, 0, 0, 0);
                } else if (with_f0) {
                    const int p0 = pid - 2 * G::G1;
                    const int plane = p0 / G::G0, k = p0 - plane * G::G0;
                    const int xx = x0 + (k & 1) * 16 + (ln >> 2), y = yb0 + (k >> 1);
                    const bool ok = xx < a.W && y < a.H;
                    const unsigned vo = ok ? (unsigned)y * rowb0 + (unsigned)((xx * a.f0_cs + plane * 16 + (ln & 3) * 4) * 4) : CVR_OOB;
                    if (!(ABL & 2))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lptr_t)(smem + G::RING_F + plane * G::F0PL + k * 256 + buf * G::F0BUF), 16,
                                                                 (int)vo, 0, 0, 0);
                }
            }
        };

        // ---- prologue: window rows 0..11 (image rows Y0-4 .. Y0+7) and the first f0 tile
        stamp();
        issue_pieces(Y0 - 4, 0, Y0, 0, true);
        issue_pieces(Y0, 4, 0, 0, false);
        issue_pieces(Y0 + 4, 8, 0, 0, false);
        CVR_WAIT_VM(0);
        stamp();
        cvr_barrier();
        stamp();

        for (int s = 0; s < nsteps; ++s) {
            // ---- prefetch of step s+1: window rows 4s+12 .. 4s+15 (slots of the rows that expired
            // at step s-1) and the next f0 tile
            const bool more = s + 1 < nsteps;                 // uniform
            if (more) issue_pieces(Y0 + 4 * s + 8, (4 * s + 12) & (G::RING - 1), Y0 + 4 * (s + 1), (s + 1) & 1, true);
            stamp();

            // ---- correlate: 4 quad steps x 12 window rows, one column, on this lane's 16 channels
            // Packed accumulators: acc[j][v] = (sum over even channels, sum over odd channels) -- one
            // v_pk_fma_f32 does two of the four products of a channel quad.  Plain v_fma_f32 issues at 16
            // lanes per clock on gfx950 (the 157 TFLOP/s fp32 vector peak is the PACKED rate): with scalar
            // FMAs this loop was VALU-bound at 4.3 us per step.
            f32x2 acc[4][9];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < 9; ++v) acc[j][v] = f32x2{0.f, 0.f};

            int lc = lane;                                     // opaque per step: see issue_pieces
            asm volatile("" : "+v"(lc));
            const int x = lc & 31, half = lc >> 5;             // strip column, channel half (plane)
            const int rot = (x >> 2) & 3;
            const float* f0b = f0s + (s & 1) * G::F0BUF + half * G::F0PL + x * 16;
            const float* rb = ring + half * G::RPLANE + (x + wave) * 16;
            int goff[3];                                       // float offsets of the 3 slot groups of the window
#pragma unroll
            for (int g = 0; g < 3; ++g) goff[g] = ((s + g) & 3) * 4 * G::PROW;

            constexpr int NK = 4 * 12, PD = CVR_PD;            // (quad step, window row) pairs; LDS prefetch distance
            f32x4 wb[PD + 1];
            f32x4 fa[4];                                       // f0 quads of the 4 pixels (single buffer, see below)
            auto qoff = [&](int sq) { return ((sq + rot) & 3) * 4; };
            auto load_w = [&](int k) {
                const int sq = k / 12, r = k - sq * 12;
                wb[k % (PD + 1)] = *reinterpret_cast<const f32x4*>(rb + goff[r >> 2] + (r & 3) * G::PROW + qoff(sq));
            };
            // pixel j's f0 quad is used by window rows j .. j+8 of a quad step: the next step's quad is
            // loaded as soon as row j+8 is done (rows 9, 10, 11, and row 0 of the next step for j = 3)
            auto load_a = [&](int sq, int j) {
                fa[j] = *reinterpret_cast<const f32x4*>(f0b + j * G::F0ROW + qoff(sq));
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) load_a(0, j);
#pragma unroll
            for (int k = 0; k < PD; ++k) load_w(k);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int sq = k / 12, r = k - sq * 12;
                if (k + PD < NK) load_w(k + PD);
                if (r >= 9 && sq + 1 < 4) load_a(sq + 1, r - 9);
                if (r == 0 && sq > 0) load_a(sq, 3);
                const f32x4 w = wb[k % (PD + 1)];
                if (ABL & 1) {
                    asm volatile("" ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
                } else {
                    const f32x2 wlo = __builtin_shufflevector(w, w, 0, 1), whi = __builtin_shufflevector(w, w, 2, 3);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int v = r - j;
                        if (v >= 0 && v < 9) {
                            const f32x4 av = fa[j];
                            const f32x2 alo = __builtin_shufflevector(av, av, 0, 1), ahi = __builtin_shufflevector(av, av, 2, 3);
                            acc[j][v] = __builtin_elementwise_fma(alo, wlo, acc[j][v]);
                            acc[j][v] = __builtin_elementwise_fma(ahi, whi, acc[j][v]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }

            stamp();
            // ---- add the two channel halves: after the swap of the pair (A, B) the lower half-wave holds
            // both halves of A, the upper both halves of B; then mean over C, leaky-relu -> stage
            // stage address of pair member B relative to A: the next vertical shift of the same pixel (+9
            // floats) or, after the last shift, the first one of the next pixel row -- two per-lane bases
            // and an immediate instead of 18 selected addresses
            constexpr int DNEXT = WS * G::SROW - 8 * 9;
            int le = lane;
            asm volatile("" : "+v"(le));
            const int ex = le & 31, ehalf = le >> 5;
            float* stp9 = stg + ex * G::SROW + wave + ehalf * 9;
            float* stpn = stg + ex * G::SROW + wave + ehalf * DNEXT;
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int ia = 2 * i, ib = 2 * i + 1;
                const int ja = ia / 9, va = ia - ja * 9, jb = ib / 9, vb = ib - jb * 9;
                float A = acc[ja][va][0] + acc[ja][va][1], B = acc[jb][vb][0] + acc[jb][vb][1];
                cvr_swap32(A, B);
                const float sum = A + B;
                const int offa = ja * WS * G::SROW + va * 9;
                (va < 8 ? stp9 : stpn)[offa] = pwc_lrelu(sum * inv_c, a.slope);
            }
            stamp();
            cvr_barrier();
            stamp();

            // ---- copy-out: 21 items per pixel (20 quads + the last float), consecutive lanes =
            // consecutive items of a pixel record
            {
                const int yb = Y0 + 4 * s;
                // the item decode below is step-invariant; recomputing it from an opaque copy of the thread
                // id keeps ~30 VGPRs from living across the correlation loop (they spilled)
                int tt = t;
                asm volatile("" : "+v"(tt));
#pragma unroll
                for (int i = 0; i < G::NST_OUT; ++i) {
                    const int e = tt + i * G::T;
                    const int p = e / 21, q = e - p * 21;
                    const int y = yb + (p >> 5), xx = x0 + (p & 31);
                    const bool in = e < G::NQUADS;
                    const float* sp = stg + (in ? p * G::SROW + q * 4 : 0);
                    const f32x2 lo = *reinterpret_cast<const f32x2*>(sp);
                    f32x2 hi = {0.f, 0.f};
                    if (q < 20) hi = *reinterpret_cast<const f32x2*>(sp + 2);
                    const bool ok = in && (y < a.H) && (xx < a.W) && !(ABL & 4);
                    const unsigned vo = ok ? (unsigned)(((y * a.W + xx) * a.out_cs + q * 4) * 4) : CVR_OOB;
                    // q == 20: only the first float belongs to the record; the b128 store of the other lanes is
                    // disabled for it and a b32 store takes its place (both instructions are always issued)
                    // (whole-vector bit_cast: __builtin_bit_cast of a vector ELEMENT is miscompiled by this hipcc --
                    // every element reads as element 0)
                    const f32x4 f4 = {lo[0], lo[1], hi[0], hi[1]};
                    const cvr_u32x4 v4 = __builtin_bit_cast(cvr_u32x4, f4);
                    __builtin_amdgcn_raw_buffer_store_b128(v4, ro, (int)(q < 20 ? vo : CVR_OOB), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v4[0], ro, (int)(q == 20 ? vo : CVR_OOB), 0, 0);
                }
                if (a.f0_copy) {
                    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(a.f0_copy + (size_t)n * a.H * a.W * a.f0_copy_cs), 0, a.H * a.W * a.f0_copy_cs * 4, 0x00020000);
                    const float* fb = f0s + (s & 1) * G::F0BUF;
#pragma unroll
                    for (int i = 0; i < G::NST_CPY; ++i) {
                        const int e = tt + i * G::T;                // (pixel, plane, quad), quad fastest
                        const int p = e >> 3, pl = (e >> 2) & 1, q = e & 3;
                        const int y = yb + (p >> 5), xx = x0 + (p & 31);
                        const bool ok = (e < Q * WS * 8) && (y < a.H) && (xx < a.W) && !(ABL & 4);
                        const f32x4 v4 = *reinterpret_cast<const f32x4*>(fb + (ok ? pl * G::F0PL + p * 16 + q * 4 : 0));
                        const unsigned vo = ok ? (unsigned)(((y * a.W + xx) * a.f0_copy_cs + pl * 16 + q * 4) * 4) : CVR_OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvr_u32x4, v4), rc, (int)vo, 0, 0);
                    }
                }
            }
            stamp();
            // the DMA pieces of step s+1 are older than this step's stores: wait for them only
            if (more) {
                if (a.f0_copy) CVR_WAIT_VM(2 * G::NST_OUT + G::NST_CPY);
                else CVR_WAIT_VM(2 * G::NST_OUT);
            }
            stamp();
            cvr_barrier();
            stamp();
        }
    }
}

// Work decomposition: strips of 32 columns, each cut into `nseg` segments of `seg_rows` rows (a
// multiple of 4).  A segment start costs 8 halo rows + a prologue without stores, a second round of
// workgroups costs a whole segment: pick the split with the smallest estimated makespan on 256 CUs.
static void cv_roll_plan(int N, int H, int W, int* nstrips, int* nseg, int* seg_rows) {
    const int ns = (W + 31) / 32;
    long best = -1;
    int best_k = 1, best_rows = ((H + 3) / 4) * 4;
    const int kmax = (H + 3) / 4;
    for (int k = 1; k <= kmax; ++k) {
        const int rows = ((((H + k - 1) / k) + 3) / 4) * 4;
        const int segs = (H + rows - 1) / rows;
        const long items = (long)N * ns * segs;
        const long rounds = (items + 255) / 256;
        const long cost = rounds * (rows + 10);
        if (best < 0 || cost < best) { best = cost; best_k = segs; best_rows = rows; }
    }
    *nstrips = ns; *nseg = best_k; *seg_rows = best_rows;
}

static bool cv_roll_eligible(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* out, int out_cs,
                             const float* f0_copy, int f0_copy_cs, int H, int W, int C, int R) {
    if (R != 4 || C != 32) return false;
    if ((f0_cs & 3) || (f1_cs & 3) || (out_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1) || !pwc_aligned16(out)) return false;
    if (f0_copy && ((f0_copy_cs & 3) || !pwc_aligned16(f0_copy))) return false;
    // buffer resources are per image: byte extents must stay below 2^31 (the OOB marker)
    const long px = (long)H * W;
    if (px * f0_cs * 4 >= (1L << 31) || px * f1_cs * 4 >= (1L << 31) || px * out_cs * 4 >= (1L << 31)) return false;
    if (f0_copy && px * f0_copy_cs * 4 >= (1L << 31)) return false;
    return (long)H * W >= 4096;     // small maps: the tile / coarse kernels have more workgroups
}

static int cv_roll_launch(const float* f0, int f0_cs, const float* f1, int f1_cs, float* out, int out_cs, float* f0_copy,
                          int f0_copy_cs, int N, int H, int W, float slope, hipStream_t s) {
    using G = CvRollGeom;
    CvRollArgs a;
    a.f0 = f0; a.f1 = f1; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.slope = slope; a.dbg = nullptr;
    cv_roll_plan(N, H, W, &a.nstrips, &a.nseg, &a.seg_rows);
    const long items = (long)N * a.nstrips * a.nseg;
    if (items >= (1L << 31)) return PWC_ERANGE;
    const size_t lds = (size_t)G::LDS_F * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_roll_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const unsigned nwg = (unsigned)(items < 256 ? items : 256);
    hipLaunchKernelGGL((cost_volume_roll_kernel<0>), dim3(nwg), dim3(G::T), lds, s, a);
    return pwc_launch_status();
}
