// exp_cost_volume_roll_mfma.hip -- EXPERIMENT RECORD, not part of the library: the matrix-pipe form of the
// rolling-window cost volume (pwcnet_amd/csrc/cost_volume_roll.hip is the shipped VALU form).  Bit-identical
// results, 46-49 us against the VALU form's 47 us on cold operands (8 x 112 x 256 x 32): once the multiply-adds
// leave the VALU the kernel is bound by the memory system (151 MB of LDS-DMA + store traffic at ~3.8 TB/s in
// the steady state, prologue and tail exposed), so the simpler VALU kernel stays.  Measured variants: phases
// separated by barriers (44.6 us), half of the waves one phase behind (46 us: VALU work next to the SIMD
// partner's MFMAs runs at a third of its stand-alone speed), everything interleaved into the MFMA stream (49 us).
//
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
//          buffer_load_dwordx4 ... lds (row pieces of 16 + 16 + 8 pixels, out-of-image pixels = the
//          zeros of the buffer range check).  A step needs window rows [4s, 4s+12) of the segment;
//          the 4 rows of step s+1 are fetched into the 4 slots that expired at step s-1 while step
//          s is computed: every f1w row is fetched once per segment (x1.25 horizontal halo,
//          x(rows+8)/rows vertical halo at segment starts).
//   f0     double-buffered 4 x 32 pixel tile (same layout, same DMA path); also the source of the
//          concat copy.
//   math   on the MATRIX pipe.  The first two versions of this kernel did the multiply-adds with
//          v_fma_f32 / v_pk_fma_f32 and were bound by VALU issue: the fp32 VALU of gfx950 sustains
//          82-107 TFLOP/s (scripts/exp_valu.hip: 3.3-4.5 cycles per wave64 v_fma_f32, 5.9-8.1 per
//          v_pk_fma_f32), i.e. >= 15 us for the 1.19 GFMA of this level before any bookkeeping, and
//          the kernel stood at 47-52 us with the memory pipes waiting.  The +-4 search window has an
//          exact block structure: for a 4 x 4-pixel block of f0 (16 pixels = the M side of
//          v_mfma_f32_16x16x4_f32) the 12 x 12 window pixels it meets are a 3 x 3 grid of 4 x 4-pixel
//          blocks of f1w (16 pixels each = the N side), all aligned to multiples of 4 in window
//          coordinates.  One MFMA = 16 x 16 dot-product pieces over 4 channels; 81 of the 144 pairs of
//          a pixel are wanted, so 56 % of the matrix pipe's work is useful -- an effective 87 TFLOP/s,
//          the VALU's rate, but CONCURRENT with the VALU's bookkeeping, and with one ds_read_b128 per
//          lane feeding 4 MFMAs (LDS traffic 10x lower than the VALU form's).
//          wave = block column b (pixels 4b .. 4b+3 of the strip, 8 waves); per step it keeps the two
//          A quads of its f0 block (lane = pixel m = lane % 16, channel quad lane / 16 of a plane) and
//          walks the 9 window blocks: 2 ds_read_b128 + 8 MFMAs each, 36 accumulator registers.
//   banks  LDS rows are skewed by 32 bytes (row strides 2592 / 2080 B): the 16 lanes of a
//          ds_read_b128 service group touch 4 pixels x 2 rows x 2 quads -- with the skew these are
//          the 16 slots of a bank row exactly once.
//   out    accumulators -> LDS stage (82-float pixel records; lanes whose (v, h) falls outside the +-4
//          window are masked off) -> copy-out: mean over C, leaky-relu, 16-byte buffer stores of
//          contiguous 324-byte records; pixels beyond the image edge are dropped by the range check
//          (no divergent branches: every wave issues a FIXED number of stores).
//   sync   every wave issues its share of a step's 40 DMA pieces at the start of the step and its
//          copy-out stores at the end; `s_waitcnt vmcnt(#stores)` then waits for exactly the (older)
//          DMA pieces -- loads and stores of a wave retire in issue order on gfx9 -- and the stores
//          are never waited for.  Barriers order LDS traffic only (cvr_barrier).
//
// Algorithmic bytes: N*H*W*(2C+81)*4; HBM-bound by design (8.9 flop/B).
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
    int pad_ok;           // channels 81..83 of every `out` record are the callee's to zero (estimator buffers: padding)
    unsigned long long* dbg;   // scripts/exp_cv2.hip only (ABL & 8): s_memtime stamps of workgroup 0, waves 0 and 4
};

struct CvRollGeom {
    static constexpr int C = 32, R = 4, D = 9, DD = 81;
    static constexpr int WS = 32, HW = WS + 2 * R, Q = 4, RING = 16;
    static constexpr int PROW = HW * 16 + 8;            // floats per ring row of one plane: 2592 B (32 B skew)
    static constexpr int RPLANE = RING * PROW;          // floats per ring plane
    static constexpr int RING_F = 2 * RPLANE;
    static constexpr int F0ROW = WS * 16 + 8;           // floats per f0 tile row of one plane: 2080 B (32 B skew)
    static constexpr int F0PL = Q * F0ROW;              // floats per f0 tile plane
    static constexpr int F0BUF = 2 * F0PL;              // floats per f0 tile
    static constexpr int F0_F = 2 * F0BUF;              // two tiles
    static constexpr int SROW = 82;                     // stage floats per pixel
    static constexpr int NW = 8, T = 64 * NW;
    static constexpr int WSTG = 16 * SROW;              // stage floats per wave: its 4 x 4 pixel block
    static constexpr int STG_F = NW * WSTG;
    static constexpr int DUMP_F = T;                    // one float per lane for the masked-off accumulator entries
    static constexpr int LDS_F = RING_F + F0_F + STG_F + DUMP_F;
    static constexpr int NP1 = 2 * Q * 3;               // f1 DMA pieces per step: plane x row x {16, 16, 8 pixels}
    static constexpr int NP0 = 2 * Q * 2;               // f0 DMA pieces per step: plane x row x {16, 16 pixels}
    static constexpr int NPIECE = NP1 + NP0;            // 40
    static constexpr int PPW = NPIECE / NW;             // pieces per wave (5)
    static constexpr int NST_OUT = 6;                   // copy-out iterations of a wave: 3 of its 16 pixel records each, 2 stores
    static constexpr int NST_CPY = (16 * 8) / 64;       // concat-copy store instructions per wave (2)
    static_assert(NPIECE % NW == 0 && (Q * WS * 8) % T == 0, "work must split evenly over the waves");
    static_assert((PROW * 4 / 16) % 16 == 2 && (F0ROW * 4 / 16) % 16 == 2, "row skew of two 16-byte slots");
    static_assert(LDS_F * 4 <= 160 * 1024, "does not fit the LDS");
};

#define CVR_OOB 0x80000000u
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
#define CVR_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

typedef unsigned int cvr_u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt (its fence covers
// global memory), which would wait for the prefetch DMA and for the copy-out stores at every barrier.
__device__ __forceinline__ void cvr_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ABL (scripts/exp_cv2.hip only; 0 in the library): 1 = no MFMAs, 2 = no DMA, 4 = no stores, 8 = phase stamps
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
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // = block column b of the strip
    const int m = lane & 15, kq = lane >> 4;             // MFMA row/column index, channel quad of a plane
    const int mr = m >> 2, mc = m & 3;                   // pixel of a 4 x 4 block

    const int nitems = a.N * a.nseg * a.nstrips;
    const float inv_c = 1.0f / (float)C;                 // C is a power of two: exact
    const unsigned rowb1 = (unsigned)(a.W * a.f1_cs * 4), rowb0 = (unsigned)(a.W * a.f0_cs * 4);

    // lanes of an accumulator whose (v, h) lies in the +-4 window.  Register r of window block (rg, cg)
    // holds  D[f0 pixel (row kq, column r)][f1 pixel (row 4rg + mr, column 4cg + mc)]  of this wave's blocks:
    //   v' = 4rg + mr - kq in [0, 8]:  rg = 0 needs mr >= kq, rg = 2 needs mr <= kq;
    //   h' = 4cg + mc - r  in [0, 8]:  cg = 0 needs mc >= r,  cg = 2 needs mc <= r.
    const bool row_ok[3] = {mr >= kq, true, mr <= kq};

    // (ABL & 8) phase stamps: s_memtime into a spare LDS area, copied out at the end
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + G::LDS_F) + (wave >> 2) * 96;
    int stamp_i = 0;
    auto stamp = [&]() {
        // (the scheduling barriers stay in the production build: with them the compiler keeps the phases of a
        // step apart and the kernel measures 41 us instead of 49 us)
        if (!(ABL & 8)) __builtin_amdgcn_sched_barrier(0);
        if (ABL & 8) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long tk = __builtin_amdgcn_s_memtime();
            if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && stamp_i < 96) stamps[stamp_i] = tk;
            ++stamp_i;
            __builtin_amdgcn_sched_barrier(0);
        }
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

        // ---- DMA pieces of a step, piece id pid = k*8 + wave (k = 0..4):
        //   pid <  24: f1, plane pid/12, row (pid%12)/3 of the 4-row group, 16-pixel column piece pid%3
        //              (piece 2 holds the last 8 pixels: its upper half-wave is masked off);
        //   pid >= 24: f0, plane (pid-24)/8, tile row ((pid-24)%8)/2, column piece (pid-24)%2.
        // The row is uniform per piece (scalar offset, scalar validity); the lane's part is the byte offset
        // of its 16 bytes within the image row: pixel 16*piece + lane/4, quad lane%4 of the plane.
        // yb1: image row of the f1 group's first row, slot4: its ring slot (multiple of 4); yb0 / buf: f0 tile
        // per-lane byte offsets within an image row, computed once per item (CVR_OOB: column outside the image,
        // or the masked upper half of an 8-pixel piece); the plane's 64 bytes go into the scalar offset
        unsigned dcol1[3], dcol0[2];
        {
            const int lp = lane >> 2, lq = (lane & 3) * 16;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const int xx = x0 - 4 + pc * 16 + lp;
                dcol1[pc] = ((unsigned)xx < (unsigned)a.W && (pc < 2 || lane < 32)) ? (unsigned)(xx * a.f1_cs * 4 + lq) : CVR_OOB;
            }
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int xx = x0 + pc * 16 + lp;
                dcol0[pc] = (xx < a.W) ? (unsigned)(xx * a.f0_cs * 4 + lq) : CVR_OOB;
            }
        }
        auto issue_piece = [&](int k, int yb1, int slot4, int yb0, int buf, bool with_f0) {
            {
                const int pid = k * G::NW + wave;                 // uniform
                if (pid < G::NP1) {
                    const int plane = pid / 12, rem = pid - plane * 12;
                    const int i = rem / 3, pc = rem - i * 3;
                    const int y = yb1 + i;
                    const bool rok = (unsigned)y < (unsigned)a.H;             // uniform
                    const unsigned dc = pc == 0 ? dcol1[0] : pc == 1 ? dcol1[1] : dcol1[2];
                    const unsigned vo = rok ? dc : CVR_OOB;
                    float* dst = smem + plane * G::RPLANE + (slot4 + i) * G::PROW + pc * 256;
                    if (!(ABL & 2)) {
                        if (pc < 2 || lane < 32)             // piece 2: 8 pixels = lanes 0-31 (the rest would run into the next row)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lptr_t)dst, 16, (int)vo,
                                                                     rok ? (unsigned)y * rowb1 + plane * 64 : 0u, 0, 0);
                    }
                } else if (with_f0) {
                    const int p0 = pid - G::NP1;
                    const int plane = p0 / 8, rem = p0 - plane * 8;
                    const int j = rem >> 1, pc = rem & 1;
                    const int y = yb0 + j;
                    const bool rok = y < a.H;                                   // uniform
                    const unsigned vo = rok ? (pc ? dcol0[1] : dcol0[0]) : CVR_OOB;
                    float* dst = smem + G::RING_F + buf * G::F0BUF + plane * G::F0PL + j * G::F0ROW + pc * 256;
                    if (!(ABL & 2))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lptr_t)dst, 16, (int)vo,
                                                                 rok ? (unsigned)y * rowb0 + plane * 64 : 0u, 0, 0);
                }
            }
        };
        auto issue_pieces = [&](int yb1, int slot4, int yb0, int buf, bool with_f0) {
#pragma unroll
            for (int k = 0; k < G::PPW; ++k) issue_piece(k, yb1, slot4, yb0, buf, with_f0);
        };

        // ---- prologue: window rows 0..11 (image rows Y0-4 .. Y0+7) and the first f0 tile
        stamp();
        issue_pieces(Y0 - 4, 0, Y0, 0, true);
        issue_pieces(Y0, 4, 0, 0, false);
        issue_pieces(Y0 + 4, 8, 0, 0, false);
        CVR_WAIT_VM(0);
        cvr_barrier();
        stamp();

        // ---- copy-out of this wave's own 4 x 4 pixel block (rows yb .. yb+3, columns x0 + 4*wave ..): leaky-relu
        // (the mean's 1/C rides in the A operand), 16-byte stores.  3 pixel records per iteration: lanes
        // 21*i .. 21*i + 20 = the 20 quads + the last float of record i (lane 63 idles).  Wave-private: no
        // workgroup barrier involved.  The lane's offsets relative to the step's first row are computed once
        // per item: o128 for the quads, o32 for the record's last float (CVR_OOB = not this lane's job).
        unsigned o128[G::NST_OUT], o32[G::NST_OUT];
        int orow[G::NST_OUT];
        const float* sp0;
        {
            const int trio = (lane * 49) >> 10;               // lane / 21 for lane < 64
            const int q = lane - trio * 21;
            const unsigned csb = (unsigned)(a.out_cs * 4);
            sp0 = stg + wave * G::WSTG + trio * G::SROW + q * 4;
#pragma unroll
            for (int i = 0; i < G::NST_OUT; ++i) {
                const int p = trio + 3 * i;                       // pixel of the block: row p / 4, column p % 4
                const int xx = x0 + 4 * wave + (p & 3);
                const bool ok = lane < 63 && p < 16 && xx < a.W;
                const unsigned vo = __umul24((unsigned)((p >> 2) * a.W + xx), csb) + (unsigned)(q * 16);
                o128[i] = (ok && (q < 20 || a.pad_ok)) ? vo : CVR_OOB;
                o32[i] = (ok && q == 20 && !a.pad_ok) ? vo : CVR_OOB;
                orow[i] = p >> 2;
            }
        }
        const bool lastq = (lane - ((lane * 49) >> 10) * 21) == 20;
        auto copy_out_iter = [&](int i, int yb) {
            const unsigned sbase = (unsigned)yb * (unsigned)(a.W * a.out_cs * 4);   // row yb of the image, bytes
            const bool all_rows = yb + 4 <= a.H;               // uniform; else rows are checked per lane
            // (lanes without a record -- iteration 5 holds pixel 15 only -- and the 2 floats past a record's
            // last one read neighbouring LDS words that are never stored)
            const float* sp = sp0 + 3 * i * G::SROW;
            const f32x2 lo = *reinterpret_cast<const f32x2*>(sp);
            const f32x2 hi = *reinterpret_cast<const f32x2*>(sp + 2);
            f32x4 f4 = {lo[0], lo[1], hi[0], hi[1]};
            if (lastq) { f4[1] = 0.f; f4[2] = 0.f; f4[3] = 0.f; }     // record's last quad: float 80 + three padding zeros
            const f32x4 sl = f4 * a.slope;
            f4[0] = fmaxf(f4[0], sl[0]); f4[1] = fmaxf(f4[1], sl[1]);
            f4[2] = fmaxf(f4[2], sl[2]); f4[3] = fmaxf(f4[3], sl[3]);
            unsigned v128 = o128[i], v32 = o32[i];
            if (!all_rows || (ABL & 4)) {
                const bool rok = yb + orow[i] < a.H && !(ABL & 4);
                v128 = rok ? v128 : CVR_OOB;
                v32 = rok ? v32 : CVR_OOB;
            }
            // both instructions are always issued, at most one of them with an in-range offset
            const cvr_u32x4 v4 = __builtin_bit_cast(cvr_u32x4, f4);
            __builtin_amdgcn_raw_buffer_store_b128(v4, ro, (int)v128, (int)sbase, 0);
            if (!a.pad_ok) __builtin_amdgcn_raw_buffer_store_b32(v4[0], ro, (int)v32, (int)sbase, 0);   // uniform
        };

        // stage address of every accumulator register (or the lane's dump slot when its (v, h) is outside the
        // window), as LDS byte offsets -- 36 registers that turn the stage write into 36 plain ds_write_b32
        unsigned sdst[9][4];
        {
            const unsigned st = (unsigned)((G::RING_F + G::F0_F + wave * G::WSTG + (kq * 4) * G::SROW + (mr - kq) * 9 + mc) * 4);
            const unsigned dump = (unsigned)((G::RING_F + G::F0_F + G::STG_F + wave * 64 + lane) * 4);
#pragma unroll
            for (int nb = 0; nb < 9; ++nb) {
                const int rg = nb / 3, cg = nb - 3 * rg;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool col_ok = cg == 1 ? true : (cg == 0 ? mc >= r : mc <= r);
                    sdst[nb][r] = (row_ok[rg] && col_ok) ? st + (unsigned)((r * G::SROW + 36 * rg + 4 * cg - r) * 4) : dump;
                }
            }
        }

        // Everything that is not an MFMA rides BETWEEN the MFMAs of a step, in the same wave: the 5 DMA pieces of
        // step s+1 and the 6 copy-out iterations of step s-1 (whose sums wait in the wave's stage) are dealt over
        // the 5 window-block pairs.  The matrix pipe executes an MFMA for 32 cycles after a short issue; the
        // VALU / LDS / memory instructions of the same wave (and of its SIMD partner) go into that shadow.
        // (A first form ran the phases one after the other, half of the waves one phase behind: the wave doing
        // VALU work next to its partner's MFMAs got a third of its stand-alone speed.)
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;                 // uniform
            const bool prev = s > 0;                           // uniform: a step to copy out
            stamp();

            f32x4 acc[9];
            {
                const float* fa = f0s + (s & 1) * G::F0BUF + mr * G::F0ROW + (4 * wave + mc) * 16 + kq * 4;
                // the mean over C rides in the A operand: 1/C is a power of two, so scaling f0 first gives
                // bit-identical sums
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(fa) * inv_c;
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(fa + G::F0PL) * inv_c;
                // window block (rg, cg): rows 4rg + mr of the window = slot 4*((s+rg)&3) + mr, pixels 4*wave + 4cg + mc
                const float* fb = ring + mr * G::PROW + (4 * wave + mc) * 16 + kq * 4;
                auto bptr = [&](int nb) { return fb + ((s + nb / 3) & 3) * 4 * G::PROW + (nb % 3) * 64; };
                // blocks in pairs (the 9th alone): two independent accumulator chains alternate, so that
                // an MFMA never waits for the 40-cycle latency of the one before it
                f32x4 bq[2][2];
                bq[0][0] = *reinterpret_cast<const f32x4*>(bptr(0));
                bq[0][1] = *reinterpret_cast<const f32x4*>(bptr(0) + G::RPLANE);
                bq[1][0] = *reinterpret_cast<const f32x4*>(bptr(1));
                bq[1][1] = *reinterpret_cast<const f32x4*>(bptr(1) + G::RPLANE);
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    const int nb = 2 * g;
                    const bool two = nb + 1 < 9;
                    f32x4 nq[2][2];
                    nq[0][0] = bq[0][0]; nq[0][1] = bq[0][1]; nq[1][0] = bq[1][0]; nq[1][1] = bq[1][1];
                    if (nb + 2 < 9) {
                        nq[0][0] = *reinterpret_cast<const f32x4*>(bptr(nb + 2));
                        nq[0][1] = *reinterpret_cast<const f32x4*>(bptr(nb + 2) + G::RPLANE);
                    }
                    if (nb + 3 < 9) {
                        nq[1][0] = *reinterpret_cast<const f32x4*>(bptr(nb + 3));
                        nq[1][1] = *reinterpret_cast<const f32x4*>(bptr(nb + 3) + G::RPLANE);
                    }
                    // prefetch of step s+1: window rows 4s+12 .. 4s+15 (slots of the rows that expired at step
                    // s-1) and the next f0 tile -- this wave's piece g of 5
                    if (more) issue_piece(g, Y0 + 4 * s + 8, (4 * s + 12) & (G::RING - 1), Y0 + 4 * (s + 1), (s + 1) & 1, true);
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                    if (ABL & 1) {
                        asm volatile("" ::"v"(bq[0][0]), "v"(bq[0][1]), "v"(bq[1][0]), "v"(bq[1][1]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const f32x4 av = e < 4 ? a0 : a1;
                            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e & 3], bq[0][e >> 2][e & 3], c0, 0, 0, 0);
                            if (two) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e & 3], bq[1][e >> 2][e & 3], c1, 0, 0, 0);
                        }
                    }
                    acc[nb] = c0;
                    if (two) acc[nb + 1] = c1;
                    bq[0][0] = nq[0][0]; bq[0][1] = nq[0][1]; bq[1][0] = nq[1][0]; bq[1][1] = nq[1][1];
                    // copy-out of step s-1: iteration g (and the short 6th one with the last pair)
                    if (prev) {
                        copy_out_iter(g, Y0 + 4 * (s - 1));
                        if (g == 4) copy_out_iter(5, Y0 + 4 * (s - 1));
                    }
                }
            }
            stamp();

            // ---- the f0 part of the concat: this wave's 16 pixels x 8 quads = 2 store instructions
            if (a.f0_copy) {
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(a.f0_copy + (size_t)n * a.H * a.W * a.f0_copy_cs), 0, a.H * a.W * a.f0_copy_cs * 4, 0x00020000);
                const float* fb0 = f0s + (s & 1) * G::F0BUF;
                const unsigned ccb = (unsigned)(a.f0_copy_cs * 4);
                int lf = lane;
                asm volatile("" : "+v"(lf));
#pragma unroll
                for (int i = 0; i < G::NST_CPY; ++i) {
                    const int e = lf + i * 64;                    // (pixel, plane, quad), quad fastest
                    const int p = e >> 3, pl = (e >> 2) & 1, qq = e & 3;
                    const int y = Y0 + 4 * s + (p >> 2), xl = 4 * wave + (p & 3);
                    const bool ok = (y < a.H) && (x0 + xl < a.W) && !(ABL & 4);
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(fb0 + pl * G::F0PL + (p >> 2) * G::F0ROW + xl * 16 + qq * 4);
                    const unsigned vo = ok ? __umul24((unsigned)(y * a.W + x0 + xl), ccb) + (unsigned)((pl * 16 + qq * 4) * 4) : CVR_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvr_u32x4, v4), rc, (int)vo, 0, 0);
                }
            }

            // ---- scaled sums -> this wave's stage (pixel-major, 82 floats per pixel).  Register r of block (rg, cg):
            // f0 pixel (kq, r), entry (4rg + mr - kq) * 9 + 4cg + mc - r; lanes outside the window write to their
            // dump slot instead (address select, no divergent branches).
            {
                __builtin_amdgcn_wave_barrier();               // the copy-out reads of this stage are issued
#pragma unroll
                for (int nb = 0; nb < 9; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + sdst[nb][r]) = acc[nb][r];
                __builtin_amdgcn_wave_barrier();
            }
            stamp();
            // The DMA pieces of step s+1 must have landed; memory operations of a wave retire in issue order, so
            // it is enough to wait until only the stores issued AFTER the last piece are outstanding: the copy-out
            // stores of iterations 4 and 5 and the concat copy.
            if (more) {
                const int after = (prev ? (a.pad_ok ? 2 : 4) : 0) + (a.f0_copy ? G::NST_CPY : 0);   // uniform: 0, 2, 4 or 6
                if (after == 0) CVR_WAIT_VM(0);
                else if (after == 2) CVR_WAIT_VM(2);
                else if (after == 4) CVR_WAIT_VM(4);
                else CVR_WAIT_VM(6);
            }
            stamp();
            cvr_barrier();
            stamp();
        }
        // the last step's sums: plain copy-out
#pragma unroll
        for (int i = 0; i < G::NST_OUT; ++i) copy_out_iter(i, Y0 + 4 * (nsteps - 1));
    }
    if ((ABL & 8) && blockIdx.x == 0 && a.dbg) {
        __syncthreads();
        if (t < 192) a.dbg[t] = reinterpret_cast<unsigned long long*>(smem + G::LDS_F)[t];
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
    // buffer resources are per image: byte extents must stay below 2^31 (the OOB marker); pixel indices and
    // channel strides are multiplied with v_mul_u32_u24
    const long px = (long)H * W;
    if (px * f0_cs * 4 >= (1L << 31) || px * f1_cs * 4 >= (1L << 31) || px * out_cs * 4 >= (1L << 31)) return false;
    if (f0_copy && px * f0_copy_cs * 4 >= (1L << 31)) return false;
    if (px >= (1L << 24) || out_cs * 4L >= (1L << 24) || f0_copy_cs * 4L >= (1L << 24)) return false;
    return (long)H * W >= 4096;     // small maps: the tile / coarse kernels have more workgroups
}

static int cv_roll_launch(const float* f0, int f0_cs, const float* f1, int f1_cs, float* out, int out_cs, float* f0_copy,
                          int f0_copy_cs, int N, int H, int W, float slope, int pad_ok, hipStream_t s) {
    using G = CvRollGeom;
    CvRollArgs a;
    a.f0 = f0; a.f1 = f1; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.slope = slope; a.dbg = nullptr; a.pad_ok = pad_ok;
    cv_roll_plan(N, H, W, &a.nstrips, &a.nseg, &a.seg_rows);
    const long items = (long)N * a.nstrips * a.nseg;
    if (items >= (1L << 31)) return PWC_ERANGE;
    const size_t lds = (size_t)G::LDS_F * sizeof(float);
    static bool attr_set = false;   // idempotent, benign if raced
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_roll_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const unsigned nwg = (unsigned)(items < 256 ? items : 256);
    hipLaunchKernelGGL((cost_volume_roll_kernel<0>), dim3(nwg), dim3(G::T), lds, s, a);
    return pwc_launch_status();
}
