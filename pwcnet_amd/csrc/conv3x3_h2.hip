// conv3x3_h2.hip -- 3x3 stride-1 'SAME' convolution, DIRECT, on the F16 matrix pipe of gfx950 with scaled two-term operand
// splits (round 4).  fp32 in, fp32 out, fp32 accumulation; more accurate than an fp32 MFMA chain (below).
//
// Replaces the same tf.layers.Conv2D(...,(3,3),(1,1),'same',dilation_rate=d) + tf.nn.leaky_relu calls as conv3x3_wino4.hip
// (reference modules.py:266-268 `optflow_l/conv2d*`, modules.py:306-323 `context/conv2d*`).
//
// Arithmetic: every fp32 operand is written  x = h + 2^-11 m',  h = fp16(x),  m' = fp16((x - h) * 2^11)  (x - h is exact in
// fp32; 11 + 11 significant bits; the scaling keeps m' out of fp16's subnormal range) and
//     u v ~= uh vh + 2^-11 (uh vm' + um' vh)            three products, TWO fp32 accumulators (hh and cross), combined once.
// Measured (scripts/exp_f16x2.hip, profiles/r04_exp_f16x2_numerics.txt): error against float64 = x0.37 - 0.47 of the
// v_mfma_f32_16x16x4_f32 chain's on every distribution tried (the matrix pipe rounds once per 16 products).  No Winograd
// transform: no transform rounding (F(4x4) rounds ~6x coarser than F(2x2)), no per-position operand split, no 36-position
// accumulator set -- the kernel is an implicit GEMM whose tiles are bounded by LDS and registers like any GEMM.
// Range: |x| must stay below 65504 (fp16); the fp32 kernels remain for anything else.
// The matrix work: 9 taps x 3 products at the F16 rate (16x the fp32 MFMA rate) = 27/16 fp32-MFMA-equivalents per
// multiply-add against F(4x4)'s 36/16 -- 0.75x the matrix time of the fp32 Winograd kernel, with nothing else on the SIMD that
// serialises with it (no packed fp32 VALU: the split is plain v_cvt / v_sub / v_mul, once per workgroup and 16-channel stage).
//
// Work decomposition (512 threads = 8 waves, one workgroup per CU), template <CT, PT, WCG>:
//   wave      = CT cout tiles of 32 x PT pixel tiles of 32 (one image row each) of v_mfma_f32_32x32x16_f16: 2 CT PT
//               accumulator tiles (hh, cross) of 16 registers;
//   workgroup = WCG cout groups x WPG = 8 / WCG pixel groups: 32 CT WCG couts x (PT WPG rows x 32 columns).
//               (2,2,2): 128 couts x 8 rows   (2,2,1): 64 couts x 16 rows   (3,1,1): 96 couts x 8 rows   (1,2,1): 32 couts x 16 rows
//               (1,2,2): 64 couts x 8 rows (more, smaller workgroups for the small pyramid levels)
//   Per 16-channel stage: the raw fp32 patch ((rows + 2) x 34 pixels x 64 bytes) arrives by buffer_load ... lds into a staging
//   image S; all threads split it (4 channels per item) into the operand image B[stage & 1] = [patch row][chunk: vh 0-7, vh 8-15,
//   vm' 0-7, vm' 8-15][pixel 34][16 bytes] -- 32 consecutive pixels of a chunk are 512 contiguous bytes = conflict-free
//   fragments at any tap shift; the pre-split weights of a TAP ROW r, A[r] = [cout tile][dx 3][chunk: uh 0-7, uh 8-15, um' 0-7,
//   um' 8-15][cout 32][16 bytes], are a linear copy of the packed global image.  K of an MFMA = [8 channels | 8 channels] over the
//   two lane halves, four fragments per (cout tile, pixel tile) pair:
//     hh = UH x VH      cross = UH x VM' + UM' x VH          UH = [uh 0-7 | uh 8-15], UM' = [um' 0-7 | um' 8-15], VH, VM' alike.
//   Pipeline: a stage is three PARTS (tap rows).  Barrier of part (g, r): every wave is done with A[r - 1] -> the weights of
//   (g + 1, r - 1) are fetched into it (r = 0: (g, 2) into A[2]); the barrier also publishes that part (g, r + 1) has landed, so
//   the fragments of the next tap are always fetched one tap ahead, across part and stage boundaries (two fragment sets in
//   registers).  A tap is a sequence of issue slots: matrix instruction i, then ONE other thing (a fragment fetch of the next
//   tap, a fetch piece, a piece of the split).  The split of stage g + 1 runs inside taps (g, 1, 1..2) from S into B[(g + 1) & 1]; the
//   patch of stage g + 2 is fetched into S in tap (g, 2, 1).  Three barriers per stage; 1 % of the time is spent waiting for
//   fetches.
//   Stream-K: with a workspace and more tiles than CUs the launch is ONE workgroup per CU; the (tile, stage) sequence of the
//   launch is dealt in equal contiguous ranges, the pipeline above runs across tile boundaries (one prologue per workgroup),
//   and a tile cut in two is finished by the workgroup holding its first piece (finish() below).
//   Where the time goes (profiles/r04_exp_h2.txt, r04_exp_h2_micro.txt, r04_exp_h2_tap_timeline*.txt): a loop of nothing but
//   these matrix instructions on random data sustains 1.6 PFLOP/s = 20.5 ns per instruction and SIMD (the chip is power-limited
//   under the F16 matrix pipe: 2.2 PFLOP/s on all-zero operands) = 124 us for the 128 -> 128 layer at 8 x 112 x 256.  The kernel
//   takes 175 us: inside the main loop the pipe delivers one instruction per 24.8 ns whether one or both waves of a SIMD are
//   feeding it (83 % of the sustained rate; the three barriers per stage and the taps that carry fetch pieces or the split make
//   up the rest), piece ends (16 x 128 KB of stores per workgroup and tile) take 9 %, the prologue 3 %.  Moving the fetch pieces
//   (spread, staggered between the waves of a SIMD, issued right behind the barriers, given to half of the waves, s_setprio) or
//   the weights off the LDS-DMA path (through registers: r04_exp_h2_weights_through_registers.txt) changes the per-tap pattern
//   and not the time: the matrix pipe is what the loop waits for.
//   The piece ends were tried as part of the NEXT tile's first tap (that tap pair-major -- hh, cross, cross of pair p, the four
//   16-byte chunks of pair p + 1's output between them, each pair stored ahead of the instruction that starts its accumulators
//   again): the register count is unchanged on paper (an old and a new accumulator of a pair never live together), the
//   compiler's allocation is not -- 256 registers and 201 - 629 spilled in the 128 / 96 / 64 x 16 variants, as a third copy of
//   the tap and as a predicated form of the fresh tap alike (the allocator keeps the accumulators in two sets of tuples, one per
//   instantiation of the first tap, and copies between them).  With ONE copy of the first tap -- its accumulate-or-start choice a
//   uniform branch per instruction, the matrix instructions as inline assembly with the accumulator tied through them -- the
//   allocation holds (0 - 27 registers spilled), but the launches got slower, 128 -> 128 202 - 212 us against 175 - 180 and
//   64 -> 32 41 against 39 (a pair-major first tap in EVERY stage, instructions the scheduler cannot see into), before its
//   results were right.  Left there.
//   The split in five instructions per value pair instead of seven (conv3x3_c16pair.hip's c16_split2 with plain multiplies)
//   changes no launch by more than the noise (128 -> 128, 96 -> 64, 64 -> 32, 32 -> 32, d = 16): the split already hides
//   behind the matrix instructions of its taps.
#pragma once
#include "pwc_common.h"
#include <type_traits>

typedef float pwc_f32x16 __attribute__((ext_vector_type(16)));

struct H2Args {
    const float* x;
    const void* wp;      // packed split weights [c16][tap row 3][cout tile of 32][dx 3][chunk 4][cout 32][8 fp16]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // pixel tiles per (sub-)image, cout blocks of the workgroup's 32 CT WCG couts
    int dil_y, dil_x;    // steps of the pixel lattice a workgroup's tile lives on (dilation d: (d, d), or (d, d / 2) with XS = 2)
    int ntiles;
    float* ws_partial;   // stream-K: one partial tile (32 CT WCG couts x PT WPG x 32 pixels, fp32) per workgroup, words 0xFFFFFFFF
                         // while nothing is published; null = every workgroup owns whole tiles
    unsigned* dbg;       // harness only
    int Ho, Wo;          // output size: (H, W), or (H / 2, W / 2) of the stride-2 form (S2)
    // round 5: the input as TWO tensors -- stages c16 < nc16_a are channels [16 c16, 16 c16 + 16) of x, the others channels
    // [16 (c16 - nc16_a), ...) of x2 (same pixel grid, own channel stride).  The estimator's first layer reads features_0
    // straight from the pyramid tensor: the concat copy of reference modules.py:264 does not exist.  x2 = null: all of x.
    const float* x2;
    int x2_cs, nc16_a;
    // round 6: a THIRD tensor -- stages c16 >= nc16_a + nc16_b are channels [16 (c16 - nc16_a - nc16_b), ...) of x3 (null: none).
    // The estimator's first layer then reads [cv | flow] (a dense tensor of 84-channel records), features_0 (the pyramid tensor)
    // and features_up (a dense 32-channel tensor): no producer writes into a slice of a wider record any more.  An operand whose
    // channel stride is below its stage count x 16 (84 < 96) lets its LAST stage run into the next pixel's record (beyond the
    // image: zeros); the packed weights of those channels are zero (cin_map = -1).
    const float* x3;
    int x3_cs, nc16_b;
    // round 5: caller-owned status words (null: none): PWC_STATUS_STREAMK_TIMEOUT is OR-ed into [0] when a bounded wait runs out.
    // (The RANGE of the split is not watched here: tracking the largest operand in the split cost every launch 2-3 %
    // (profiles/r05_timeline_range_tracking_cost.txt).  An operand beyond fp16's range makes NaN outputs, NaN survives every later
    // layer, and the model's last launch -- pwc_resize_bilinear_status_f32 -- reports non-finite flows.)
    unsigned* status;
};

constexpr unsigned H2_OOB = 0x7FFF0000u;
constexpr int H2_MAX_COUT = 512;             // (the bias lives in LDS)
constexpr int H2_SC1 = 16;                   // aux bit of the buffer intrinsics: device-scope access (gfx940+)
constexpr int H2_MIN_STAGES = 3;              // stream-K only where a CU gets at least this many 16-channel stages
constexpr unsigned H2_EMPTY = 0xFFFFFFFFu;    // workspace word that holds no published sum
constexpr int H2_TAPB = 4 * 32 * 16;         // bytes of the weights of one (cout tile, tap): 2048

// XS: the lattice columns a tap step spans.  1: the lattice of a dilation-d convolution is (y mod d, x mod d).  2: the lattice
// is (y mod d, x mod d/2) and a tap moves TWO lattice columns -- a sub-lattice of only 16 columns (d = 16 at W = 256) would fill
// half of a 32-column tile, the twice-as-dense lattice fills it (patch 36 instead of 34 pixels wide).
template <int CT, int PT, int WCG, int XS = 1> struct H2Cfg {
    static constexpr int PW = 32 + 2 * XS;       // patch width in pixels: 34 / 36
    static constexpr int CHK = PW * 16;          // bytes of a chunk row in the operand image: 544
    static constexpr int ROWB = 4 * CHK;         // bytes of a patch row in the operand image: 4 chunks x 34 pixels x 16 B = 2176
    static constexpr int WPG = 8 / WCG;          // pixel groups (PT rows each)
    static constexpr int NCT = CT * WCG;         // cout tiles of the workgroup
    static constexpr int TR = PT * WPG;          // tile rows: 8 or 16
    static constexpr int PH = TR + 2;            // patch rows
    static constexpr int NREC = PH * PW;         // patch pixels: 340 / 612
    static constexpr int NBP = (NREC + 15) / 16; // 1 KB pieces of the staging image (16 records of 64 B): 22 / 39
    static constexpr int PPW = (NBP + 7) / 8;    // patch pieces per wave: 3 / 5
    static constexpr int S_BYTES = (PPW * 8) * 1024;                 // staging (incl. the surplus pieces)
    static constexpr int B_BYTES = PH * ROWB;                        // one operand image: 21 760 / 39 168
    static constexpr int AP = NCT * 3 * H2_TAPB;                     // weights of a part (tap row): 24 576 / 12 288 / 18 432 / 6144
    static constexpr int NAP = AP / 1024;                            // ... in 1 KB pieces
    static constexpr int APW = (NAP + 7) / 8;                        // pieces per wave (the last round may be partial)
    static constexpr int S0 = 0, B0 = S_BYTES, A0 = B0 + ((2 * B_BYTES + 1023) / 1024) * 1024;
    static constexpr int BIAS = A0 + 3 * AP;                          // the layer's bias (H2_MAX_COUT floats)
    static constexpr int LDS = BIAS + H2_MAX_COUT * 4;
    static_assert(LDS <= 160 * 1024, "LDS");
};

// ABL (harness only): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA, 8 = m' = 0, 16 = no split at all, 32 = no fragment reads,
// 64 = the published sums are replaced by position tags and checked by the reader (self-test of the exchange), 2048 = s_memtime
// totals of the waits, barriers and piece ends per wave
// S2 (round 5): stride 2, 'SAME', even H and W -- the extractor's down-sampling layers (reference modules.py:57-60).  With the input
// seen as its four PARITY planes x_ab[y', x'] = x[2 y' + a, 2 x' + b] the strided convolution is a stride-1 one with taps at 0 / +1:
//     y[oy, ox] = sum_{a,b} sum_{ty,tx in {0,1}} w[2 ty + a, 2 tx + b] x_ab[oy + ty, ox + tx]        (w = 0 beyond index 2),
// i.e. a channel stage here is (16 channels, parity) -- stage s = (2 a + b) (C / 16) + channel group, Cin_phys = 4 C -- fetched from the
// pixels of ITS plane (the lane's pixel index + a W + b), and of the nine tap slots of a stage the four with tap row >= 1 and tap
// column >= 1 carry matrix instructions (weights: h2_pack with S2; the others keep their fetch / split / barrier duties, so the
// pipeline is the stride-1 one).  16 C products per output against the 9 C a strided convolution needs and the 36 C of "the
// stride-1 launch that stores every second sum" (round 4: slower than the fp32 kernel).
template <int CT, int PT, int WCG, int ABL = 0, int XS = 1, bool S2 = false>
__global__ __launch_bounds__(512) void conv3x3_h2_kernel(const H2Args a) {
    typedef H2Cfg<CT, PT, WCG, XS> C;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pg = wave % C::WPG, cgw = wave / C::WPG;     // pixel group (rows PT pg ..), cout group (CT tiles)
    const int ln = lane & 31, kh = lane >> 5;

    const int dly = a.dil_y, dlx = a.dil_x;
    const int nc16 = a.Cin_phys >> 4;
    const int nct_all = a.Cout >> 5;                // cout tiles of 32 in the packed image
    // ---- the workgroup's range of the launch's (tile, channel stage) sequence.  Tiles: cout block fastest, then the pixel
    // tiles of a (sub-)image; logical workgroup ids are XCD-aware (neighbouring ranges share an L2).  A tile is cut into as many
    // pieces as ranges touch it: the piece that holds the tile's FIRST stage sits at the end of workgroup w's range, the others
    // belong to workgroups w + 1, w + 2, ... (the last one at the START of its owner's range).
    const int G = (int)gridDim.x;
    const int lw = pwc_xcd_remap(blockIdx.x, G);
    const long total = (long)a.ntiles * nc16;
    const int g0 = (int)((long)lw * total / G), g1 = (int)((long)(lw + 1) * total / G);
    // tile coordinates (cout block fastest, then the pixel tiles of a (sub-)image, the sub-lattices (y mod d, x mod d) of a dilated
    // conv, the images), decoded ONCE per workgroup and stepped from then on: five divisions per tile are ~1000 cycles
    struct Tile { int cb, bx, by, rx, ry, n; };
    auto decode = [&](int tile) {
        Tile tl;
        tl.cb = tile % a.ncb;
        int rest = tile / a.ncb;
        tl.bx = rest % a.tiles_x;
        rest /= a.tiles_x;
        tl.by = rest % a.tiles_y;
        rest /= a.tiles_y;
        const int sub = rest % (dly * dlx);
        tl.n = rest / (dly * dlx);
        tl.ry = sub / dlx; tl.rx = sub - tl.ry * dlx;
        return tl;
    };
    auto next_tile = [&](Tile& tl) {
        if (++tl.cb < a.ncb) return;
        tl.cb = 0;
        if (++tl.bx < a.tiles_x) return;
        tl.bx = 0;
        if (++tl.by < a.tiles_y) return;
        tl.by = 0;
        if (++tl.rx < dlx) return;
        tl.rx = 0;
        if (++tl.ry < dly) return;
        tl.ry = 0;
        ++tl.n;
    };
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wp, 0, nc16 * nct_all * 9 * H2_TAPB, 0x00020000);

    // ---- patch fetch of a stage (tile, c16): piece b = wave + 8 i holds records 16 b .. 16 b + 15 (record = patch pixel, 64 bytes
    // = 16 channels).  The lane offsets (and the image's buffer resource) change with the tile only.
    // (the lane keeps the PIXEL index of each of its records -- H W, one past the image, for records outside it: any channel
    // stride turns that into an out-of-range offset, which the fetch answers with zeros -- and forms the byte offset for the
    // tensor a stage reads from when the piece is issued: one v_mad per piece instead of a second set of offsets)
    unsigned p_pix[C::PPW];
    const unsigned p_q16 = (unsigned)(lane & 3) * 16u;
    // ONE buffer resource: the tensor (x, x2 or x3) the stages being fetched come from, re-made where the stage sequence enters
    // another operand or another tile (select_operand; uniform, a few scalar instructions, two or three times per tile).  Round 5
    // kept one resource per operand and selected per fetch; a third one does not fit: the kernel already spills ~130 SGPRs into
    // VGPR lanes, six more pushed the 128-cout variants over 256 VGPRs into scratch (2 x slower).
    __amdgpu_buffer_rsrc_t xrsrc;
    unsigned x_cs4 = 0;                                    // channel stride of that tensor, bytes
    int x_c0 = 0;                                          // its first stage
    auto select_operand = [&](const Tile& tl, int c16) {
        if constexpr (!S2) {
            const float* base = a.x;
            int cs = a.x_cs, c0 = 0;
            if (c16 >= a.nc16_a + a.nc16_b) { base = a.x3; cs = a.x3_cs; c0 = a.nc16_a + a.nc16_b; }     // (nc16_b = the rest without an x3)
            else if (c16 >= a.nc16_a) { base = a.x2; cs = a.x2_cs; c0 = a.nc16_a; }
            xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)tl.n * a.H * a.W * cs), 0, a.H * a.W * cs * 4, 0x00020000);
            x_cs4 = (unsigned)cs * 4u;
            x_c0 = c0;
        }
    };
    auto patch_tile = [&](const Tile& tl) {
        const int y0 = tl.by * C::TR, x0 = tl.bx * 32;      // output origin of the tile, in sub-lattice coordinates
        if constexpr (S2)                                   // (one operand: the resource changes with the tile only)
            xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)tl.n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
            const int rec = (wave + 8 * i) * 16 + (lane >> 2);
            const int py = rec / C::PW, px = rec - py * C::PW;
            const int yy = S2 ? 2 * (y0 - 1 + py) : tl.ry + dly * (y0 - 1 + py);     // (S2: the pixel of plane (0, 0))
            const int xx = S2 ? 2 * (x0 - 1 + px) : tl.rx + dlx * (x0 - XS + px);
            const bool ok = rec < C::NREC && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            p_pix[i] = ok ? (unsigned)(yy * a.W + xx) : (unsigned)(a.H * a.W);
        }
    };
    auto issue_patch_piece = [&](int i, int c16) {
        if (ABL & 1) return;
        // (locals: see issue_w_piece)
        // S2: stage = parity * (channel groups) + channel group -- the 16-channel groups that share a 128-byte line of a pixel are
        // consecutive stages; the same pixel of parity plane (a, b).  (H and W are even: a pixel of plane (0, 0) inside the image
        // has its three neighbours inside too; the marker H W of a pixel outside stays out of range.)
        const int ncc = nc16 >> 2, par = S2 ? c16 / ncc : 0, cc = S2 ? c16 - par * ncc : 0;
        const int soff = S2 ? cc * 64 : (c16 - x_c0) * 64;
        const unsigned ppix = S2 ? p_pix[i] + (unsigned)((par >> 1) * a.W + (par & 1)) : p_pix[i];
        const int voff = (int)(ppix * (S2 ? (unsigned)(a.x_cs * 4) : x_cs4) + p_q16);
        lptr_t dst = (lptr_t)(sm + C::S0 + (wave + 8 * i) * 1024);
        // ONE fetch instruction on ONE resource (not one per branch: the compiler counts outstanding fetches along every path, and
        // two paths made its waits conservative, +3-4 % on every launch)
        const __amdgpu_buffer_rsrc_t rs = xrsrc;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
    };
    // ---- weights of part (c16, r) for cout block cb: NCT cout tiles x 3 taps x 2 KB, contiguous in the packed image; pieces wave,
    // wave + 8, ...
    const unsigned w_lane = (unsigned)lane * 16u;
    auto issue_w_piece = [&](int j, int c16, int cb, int r) {
        const int pc = wave + 8 * j;                   // uniform
        // (a local: with the expression inline the HOST pass of hipcc drops the kernel's stub without a diagnostic)
        const int soff = ((c16 * 3 + r) * nct_all + cb * C::NCT) * 3 * H2_TAPB;
        if (pc < C::NAP && !(ABL & 2))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(sm + C::A0 + r * C::AP + pc * 1024), 16, (int)w_lane,
                                                     soff + pc * 1024, 0, 0);
    };
#define H2_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- split of the staging image into an operand image: item = (patch pixel, 4-channel group); item j of a thread is
    // t + 512 j.  cv_read / cv_write are the two halves of an item so that they can sit in different issue slots of a tap.
    constexpr int NIT = C::NREC * 4;
    constexpr int NITJ = (NIT + 511) / 512;          // 3 / 5
    constexpr int NFULL = NIT / 512;                 // items every thread has (the last one, if any, only threads t < NIT % 512)
    auto cv_read = [&](int j) -> f32x4 {
        int it = t + 512 * j;
        if (j >= NFULL) it = it < NIT ? it : 0;      // (any valid address: the write is what is predicated)
        return *reinterpret_cast<const f32x4*>(sm + C::S0 + it * 16);
    };
    auto cv_write = [&](int j, int buf, const f32x4 v) {
        const int it = t + 512 * j;
        if (j < NFULL || it < NIT) {
            const int rec = it >> 2, g = it & 3;
            pwc_f16x4 h, m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (_Float16)v[e];
                // (x - h) 2^11 = x 2^11 - h 2^11, all three exact: one multiply and one mixed-precision FMA per element
                m[e] = (ABL & 8) ? (_Float16)0.f : (_Float16)__builtin_fmaf((float)h[e], -2048.f, v[e] * 2048.f);
            }
            const int py = rec / C::PW, px = rec - py * C::PW;
            char* dst = sm + C::B0 + buf * C::B_BYTES + py * C::ROWB + (g >> 1) * C::CHK + px * 16 + (g & 1) * 8;
            *reinterpret_cast<pwc_f16x4*>(dst) = h;
            *reinterpret_cast<pwc_f16x4*>(dst + 2 * C::CHK) = m;
        }
    };
    auto convert = [&](int buf) {
        if (ABL & 16) return;
#pragma unroll
        for (int j = 0; j < NITJ; ++j) cv_write(j, buf, cv_read(j));
    };

    // ---- fragments.  B: patch row (PT pg + pt + r), pixel ln + dx, chunk kh (h) / 2 + kh (m');  A: slot r, cout tile (CT cgw + ct),
    // tap dx, chunk kh / 2 + kh, cout ln.  Fetch order = the order the matrix instructions of the next tap want them in.
    struct Frags { pwc_f16x8 ah[CT], am[CT], bh[PT], bm[PT]; };
    constexpr int NL = 2 * (CT + PT);
    constexpr int NM = 3 * CT * PT;
    const char* const bbase = sm + C::B0 + (PT * pg) * C::ROWB + kh * C::CHK + ln * 16;
    const char* const abase = sm + C::A0 + (CT * cgw) * 3 * H2_TAPB + kh * 512 + ln * 16;
    auto load_i = [&](Frags& f, int i, int buf, int r, int dx) {
        if (ABL & 32) return;
        const char* const bb = bbase + buf * C::B_BYTES + r * C::ROWB + dx * XS * 16;
        const char* const ab = abase + r * C::AP + dx * H2_TAPB;
        // order: AH0, BH0 .. BH(PT-1), AH1 .. AH(CT-1), BM0 .., AM0 ..
        if (i == 0) f.ah[0] = *reinterpret_cast<const pwc_f16x8*>(ab);
        else if (i <= PT) f.bh[i - 1] = *reinterpret_cast<const pwc_f16x8*>(bb + (i - 1) * C::ROWB);
        else if (i < PT + CT) f.ah[i - PT] = *reinterpret_cast<const pwc_f16x8*>(ab + (i - PT) * 3 * H2_TAPB);
        else if (i < 2 * PT + CT) f.bm[i - PT - CT] = *reinterpret_cast<const pwc_f16x8*>(bb + (i - PT - CT) * C::ROWB + 2 * C::CHK);
        else f.am[i - 2 * PT - CT] = *reinterpret_cast<const pwc_f16x8*>(ab + (i - 2 * PT - CT) * 3 * H2_TAPB + 1024);
    };

    pwc_f32x16 acc[CT][PT], accx[CT][PT];
    // matrix instruction i of a tap: group i / (CT PT) (hh, UH x VM', UM' x VH), tile i % (CT PT) -- MFMAs on one accumulator
    // stay CT PT apart
    // fresh: the first tap of a piece starts the accumulators from the matrix pipe's constant 0 (no 128 v_mov per tile)
    auto mfma_i = [&](const Frags& f, int i, bool fresh) {
        const int grp = i / (CT * PT), tl = i % (CT * PT), ct = tl / PT, pt = tl % PT;
        const pwc_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ABL & 4) {
            if (i == 0) asm volatile("" ::"v"(f.ah[0]), "v"(f.am[CT - 1]), "v"(f.bh[0]), "v"(f.bm[PT - 1]));
            if (fresh && grp == 0) acc[ct][pt] = zero16;
            if (fresh && grp == 1) accx[ct][pt] = zero16;
            return;
        }
        if (grp == 0) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ct], f.bh[pt], fresh ? zero16 : acc[ct][pt], 0, 0, 0);
        if (grp == 1) accx[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ct], f.bm[pt], fresh ? zero16 : accx[ct][pt], 0, 0, 0);
        if (grp == 2) accx[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.am[ct], f.bh[pt], accx[ct][pt], 0, 0, 0);
    };

    // ---- end of a piece.  kind 0: a whole tile, or the FIRST piece of a tile whose second piece workgroup lw + 1 publishes
    // (that workgroup computes it at the start of its range, this one reaches its piece at the end of its own):
    // y = own + other + bias, leaky-relu.  kind 1: the SECOND piece of a tile (start of the range): publish hh + 2^-11 cross
    // to the workspace.  own + other is one fp32 addition of two finished sums: its result does not depend on who adds.
    // The exchange needs no flag and no fence: the workspace holds 0xFFFFFFFF words (a NaN no arithmetic produces) wherever
    // nothing is published; the reader polls each 16 bytes with device-scope loads until none of its words is the sentinel
    // and then puts the sentinel back.  (A release / acquire pair would write back and invalidate a whole L2 -- every
    // output line of the launch -- per workgroup; a flag behind plain device-scope stores was seen to overtake them.)
    // Lane = (pixel column ln, couts (r & 3) + 8 (r >> 2) + 4 kh of a tile).
    constexpr int PART_FLOATS = C::NCT * 32 * C::TR * 32;
    auto finish = [&](const Tile& tl, int kind, int nother) {
        const bool with_other = nother > 0;
        if (kind == 1) {
            // buffer addressing (base in SGPRs, one lane offset): flat pointers would cost two VGPRs per access, hoisted
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.ws_partial + (size_t)lw * PART_FLOATS + (size_t)wave * (CT * PT * 16 * 64)), 0, CT * PT * 16 * 64 * 4, 0x00020000);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = (ABL & 64) ? __builtin_bit_cast(float, 0x40000000u | ((unsigned)(lw & 255) << 16) | ((unsigned)wave << 12) | ((unsigned)((pt * CT + ct) * 4 + q) << 8) | (lane << 2) | e)
                                              : acc[ct][pt][4 * q + e] + accx[ct][pt][4 * q + e] * (1.f / 2048.f);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), prs, lane * 16, ((pt * CT + ct) * 4 + q) * 1024, H2_SC1);
                        // Seen on gfx950: with nothing but four VALU instructions between two 16-byte stores, the NEXT iteration's
                        // first result lands in the store's first data register before the store has read it (first word of the
                        // chunk wrong in the last lanes of each 16-lane pass).  The compiler knows the hazard only for stores
                        // without an SGPR offset.
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 7" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
            return;
        }
        const int y0 = tl.by * C::TR, x0 = tl.bx * 32, n0 = tl.cb * 32 * C::NCT;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (size_t)tl.n * a.Ho * a.Wo * a.y_cs), 0, a.Ho * a.Wo * a.y_cs * 4, 0x00020000);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int py = tl.ry + dly * (y0 + PT * pg + pt), px = tl.rx + dlx * (x0 + ln);      // (S2: dly = dlx = 1, output coordinates)
            const bool inside = py < a.Ho && px < a.Wo;
            const int opy = py, opx = px;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[q][e] = acc[ct][pt][4 * q + e] + accx[ct][pt][4 * q + e] * (1.f / 2048.f);
                for (int k = 1; k <= nother; ++k) {      // the other pieces, in the order of their stages; FOUR chunks per round trip
                    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(a.ws_partial + (size_t)(lw + k) * PART_FLOATS + (size_t)wave * (CT * PT * 16 * 64)), 0, CT * PT * 16 * 64 * 4, 0x00020000);
                    const int po = (pt * CT + ct) * 4 * 1024;
                    u32x4 p4[4];
                    // bounded (~1 s): a sum that never arrives (it cannot, short of a fault) leaves the sentinel -- a NaN -- in the
                    // output instead of hanging the device
                    bool missing = false;
                    for (int tries = 0; tries < (1 << 20); ++tries) {
                        missing = false;
#pragma unroll
                        for (int q = 0; q < 4; ++q) p4[q] = __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, po + q * 1024, H2_SC1);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            missing = missing || p4[q][0] == H2_EMPTY || p4[q][1] == H2_EMPTY || p4[q][2] == H2_EMPTY || p4[q][3] == H2_EMPTY;
                        missing = __builtin_amdgcn_ballot_w64(missing) != 0;
                        if (!missing) break;
                        __builtin_amdgcn_s_sleep(16);
                    }
                    // the bounded wait ran out: the sentinel (a NaN) goes into the output, and -- round 5 -- the caller is TOLD: a late
                    // publisher would leave its sums in the slot for the next launch to mistake for its own; the host refills
                    // the workspace and repeats the forward when it sees the bit
                    if (missing && a.status && lane == 0) atomicOr(a.status, (unsigned)PWC_STATUS_STREAMK_TIMEOUT);
                    const u32x4 empty = {H2_EMPTY, H2_EMPTY, H2_EMPTY, H2_EMPTY};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_raw_buffer_store_b128(empty, prs, lane * 16, po + q * 1024, H2_SC1);       // clean for the next launch
                        const f32x4 pf = __builtin_bit_cast(f32x4, p4[q]);
                        if ((ABL & 64) && a.dbg) {
                            for (int e = 0; e < 4; ++e)
                                if (p4[q][e] != (0x40000000u | ((unsigned)((lw + k) & 255) << 16) | ((unsigned)wave << 12) | ((unsigned)((pt * CT + ct) * 4 + q) << 8) | (lane << 2) | e)) {
                                    atomicAdd(a.dbg + ((lane >> 2) & 3), 1u); a.dbg[4 + ((lane >> 2) & 3)] = p4[q][e]; a.dbg[8] = pt * 100 + ct * 10 + q;
                                    a.dbg[9] = (0x40000000u | ((unsigned)((lw + k) & 255) << 16) | ((unsigned)wave << 12) | ((unsigned)((pt * CT + ct) * 4 + q) << 8) | (lane << 2) | e);
                                }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[q][e] += pf[e];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n0 + (CT * cgw + ct) * 32 + 8 * q + 4 * kh;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(sm + C::BIAS + co * 4);      // (copied into LDS by the prologue)
                    f32x4 o = o4[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += b4[e];
                    if (a.apply_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], o[e] * a.slope);
                    }
                    const unsigned vo = inside ? (unsigned)(((opy * a.Wo + opx) * a.y_cs + co) * 4) : H2_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, (int)vo, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);                    // (also bounds the registers the hoisted loads take)
                    asm volatile("s_nop 7" ::: "memory");                 // see the published sums above
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // ---- stage bookkeeping (uniform): the current stage (tile, c16, cb), the next one (weights are fetched one stage ahead) and
    // the one after (patches two ahead), advanced incrementally -- a division per stage and wave would cost as much as a tap
    unsigned long long tk_wait = 0, tk_bar = 0, tk_fin = 0;
    const unsigned long long tk_start = (ABL & 2048) ? __builtin_readcyclecounter() : 0;
    Tile tcur = decode(g0 / nc16);                    // the tile of stage g
    int c16 = g0 - (g0 / nc16) * nc16;
    Tile t1 = tcur; int c1 = c16;                     // ... of stage g + 1 (weights are fetched one stage ahead)
    auto step = [&](Tile& tl, int& c) {
        if (++c == nc16) { c = 0; next_tile(tl); }
    };
    step(t1, c1);
    Tile t2 = t1; int c2 = c1;                        // ... of stage g + 2 (patches two ahead)
    step(t2, c2);

    // ---- prologue: patch g0, parts (g0, 0) and (g0, 1); split patch g0; patch g0 + 1
    patch_tile(tcur);
    select_operand(tcur, c16);
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) issue_patch_piece(i, c16);
#pragma unroll
    for (int j = 0; j < C::APW; ++j) issue_w_piece(j, c16, tcur.cb, 0);
#pragma unroll
    for (int j = 0; j < C::APW; ++j) issue_w_piece(j, c16, tcur.cb, 1);
    if (t < a.Cout) reinterpret_cast<float*>(sm + C::BIAS)[t] = a.bias[t];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H2_BAR();
    convert(0);
    H2_BAR();
    if (g0 + 1 < g1) {
        if (c1 == 0) patch_tile(t1);
        select_operand(t1, c1);
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) issue_patch_piece(i, c1);
    }
    Frags cur, nxt;
#pragma unroll
    for (int i = 0; i < NL; ++i) load_i(cur, i, 0, 0, 0);

    // One tap = NM issue slots: matrix instruction i, then ONE other instruction group (a fragment fetch of the next tap, a
    // fetch piece, a piece of the split) that issues while the matrix pipe works.
    auto tap = [&](auto Rc, auto DXc, int g, int buf, bool more, bool fresh) {
        constexpr int R = decltype(Rc)::value, DX = decltype(DXc)::value;
        constexpr int NR = DX < 2 ? R : (R + 1) % 3, NDX = (DX + 1) % 3;
        constexpr int NMU = (!S2 || (R >= 1 && DX >= 1)) ? NM : 0;       // S2: taps 0 / +1 of the parity plane only
        const int nbuf = (R == 2 && DX == 2) ? (buf ^ 1) : buf;
        const bool do_load = !(R == 2 && DX == 2) || more;
        // Placement of the fetch pieces, measured with per-tap s_memtime stamps (profiles/r04_exp_h2_tap_timeline*.txt): all weight
        // pieces in the first tap of a part and the patch pieces in tap (2, 1) -- spreading them one per tap takes the landing slack
        // away (7 % of the time waiting for fetches, 184 vs 179 us); giving the two waves of a SIMD different taps, only one of
        // them the pieces (waves 0-3 issuing all weight pieces: 180 vs 179 us), issuing them right behind the barrier (182 vs 179)
        // or s_setprio move the delay between the waves without shortening the part.
        const bool do_w = R == 0 || more;
        const bool do_p = R == 2 && DX == 1 && g + 2 < g1;
        // the split of stage g + 1: items 0-2 in tap (1, 1), items 3-4 (16-row tiles) in tap (1, 2).  The staging reads go out in the
        // FIRST slots of the tap, ahead of that slot's fragment fetch: LDS returns in order, so a read issued behind the tap's
        // fragment fetches could only be waited for by draining all of them (lgkmcnt(0), three times per stage)
        constexpr int CV0 = (R == 1 && DX == 1) ? 0 : (R == 1 && DX == 2) ? 3 : NITJ;
        constexpr int CV1 = (R == 1 && DX == 1) ? (NITJ < 3 ? NITJ : 3) : NITJ;
        constexpr int NEXC = CV1 - CV0;
        const bool do_cv = NEXC > 0 && more && !(ABL & 16);
        constexpr int NEXA = DX == 0 ? C::APW : 0;
        constexpr int NEXP = (R == 2 && DX == 1) ? C::PPW : 0;
        constexpr int NEX = NEXA + NEXP > NEXC ? NEXA + NEXP : NEXC;
        constexpr int NSLOT = NMU > NL + NEX ? NMU : NL + NEX;
        f32x4 cvv[3];
        if (do_p && c2 == 0) patch_tile(t2);
        if constexpr (!S2) {
            if (do_p && (c2 == 0 || c2 == a.nc16_a || c2 == a.nc16_a + a.nc16_b)) select_operand(t2, c2);    // (another tile or operand)
        }
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            if (i < NMU) mfma_i(cur, i, fresh);
            if (i < NEXC && do_cv) cvv[i] = cv_read(CV0 + i);
            if (i < NL) {
                if (do_load) load_i(nxt, i, nbuf, NR, NDX);
            } else {
                const int e = i - NL;
                if (e < NEXA && do_w) {
                    if (R == 0) issue_w_piece(e, c16, tcur.cb, 2);
                    else issue_w_piece(e, c1, t1.cb, R - 1);
                }
                if (e >= NEXA && e < NEXA + NEXP && do_p) issue_patch_piece(e - NEXA, c2);
                if (e < NEXC && do_cv) cv_write(CV0 + e, buf ^ 1, cvv[e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur = nxt;
    };

    int piece_c0 = c16;                               // the stage of its tile the current piece began with
    bool drain = false;                               // output stores of the previous piece may be in flight
#define H2_STAMP(k) do { if ((ABL & 4096) && a.dbg && lane == 0 && (lw == 0 || lw == 5) && g - g0 >= 8 && g - g0 < 12) \
        a.dbg[(((lw ? 1 : 0) * 8 + wave) * 4 + (g - g0 - 8)) * 16 + (k)] = (unsigned)__builtin_readcyclecounter(); } while (0)
    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        const bool more = g + 1 < g1;
        const bool fresh0 = c16 == 0 || g == g0;        // the first stage of a piece
        // part (g, 0): this wave's pieces of part (g, 1) have landed (the patch pieces behind them may be in flight; stores and
        // fetches retire out of order with each other, so behind an epilogue everything is waited for) ...
        unsigned long long tk0 = (ABL & 2048) ? __builtin_readcyclecounter() : 0;
        H2_STAMP(0);
        if (more && !drain) {
            // patch pieces issued behind the last weight piece of part (g - 1, 2)
            constexpr int BEHIND = C::PPW;
            static_assert((C::PPW == 3 || C::PPW == 5) && C::APW >= 1 && C::APW <= 3, "vmcnt immediates");
            if (BEHIND == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        drain = false;
        if (ABL & 2048) { const unsigned long long tk1 = __builtin_readcyclecounter(); tk_wait += tk1 - tk0; tk0 = tk1; }
        H2_BAR();                          // ... everybody's; every wave is done with slot 2
        if (ABL & 2048) tk_bar += __builtin_readcyclecounter() - tk0;
        H2_STAMP(1);
        if (fresh0 && !S2) tap(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, g, buf, more, true);
        else tap(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, g, buf, more, false);
        H2_STAMP(2);
        tap(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, g, buf, more, false);
        H2_STAMP(3);
        tap(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, g, buf, more, false);
        H2_STAMP(4);
        if (ABL & 2048) tk0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // part (g, 2) and the patch of g + 1
        if (ABL & 2048) { const unsigned long long tk1 = __builtin_readcyclecounter(); tk_wait += tk1 - tk0; tk0 = tk1; }
        H2_BAR();
        if (ABL & 2048) tk_bar += __builtin_readcyclecounter() - tk0;
        H2_STAMP(5);
        tap(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, g, buf, more, false);
        H2_STAMP(6);
        if (fresh0 && S2) tap(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, g, buf, more, true);      // (its first tap with products)
        else tap(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, g, buf, more, false);
        H2_STAMP(7);
        tap(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, g, buf, more, false);
        H2_STAMP(8);
        if (ABL & 2048) tk0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // part (g + 1, 0)
        if (ABL & 2048) { const unsigned long long tk1 = __builtin_readcyclecounter(); tk_wait += tk1 - tk0; tk0 = tk1; }
        H2_BAR();
        if (ABL & 2048) tk_bar += __builtin_readcyclecounter() - tk0;
        H2_STAMP(9);
        tap(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, g, buf, more, false);
        H2_STAMP(10);
        tap(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, g, buf, more, false);
        H2_STAMP(11);
        tap(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, g, buf, more, false);
        H2_STAMP(12);
        // end of a piece: the tile's last stage, or the range's
        const bool tile_end = c16 == nc16 - 1;
        if (tile_end || !more) {
            if (ABL & 2048) tk0 = __builtin_readcyclecounter();
            if (piece_c0 != 0) {
                finish(tcur, 1, 0);                     // not the tile's first stage: publish
            } else if (tile_end) {
                finish(tcur, 0, 0);                     // a whole tile
            } else {
                // the tile's first piece: the rest belongs to workgroups lw + 1, lw + 2, ... until one's range reaches the tile's end
                const long tile_end_g = (long)(g - c16) + nc16;
                int nother = 0;
                for (int w2 = lw + 1; w2 < G && (long)w2 * total / G < tile_end_g; ++w2) ++nother;
                finish(tcur, 0, nother);
            }
            drain = true;
            piece_c0 = 0;
            if (ABL & 2048) tk_fin += __builtin_readcyclecounter() - tk0;
        }
        step(tcur, c16);
        step(t1, c1);
        step(t2, c2);
    }
    if ((ABL & 2048) && a.dbg && lane == 0) {
        unsigned* o = a.dbg + (lw * 8 + wave) * 4;
        o[0] = (unsigned)(__builtin_readcyclecounter() - tk_start); o[1] = (unsigned)tk_wait; o[2] = (unsigned)tk_bar; o[3] = (unsigned)tk_fin;
    }
#undef H2_BAR
#undef H2_STAMP
}

// ---------------------------------------------------------------- weight split + packing
// packed[c16][tap row 3][cout tile of 32][dx 3][chunk 4: uh ch 0-7, uh 8-15, um' 0-7, um' 8-15][cout 32][8 fp16];
// uh = fp16(w), um' = fp16((w - uh) * 2^11), both round-to-nearest; cin_map as in pwc_conv3x3_pack_f32.
// s2: the stride-2 form -- Cin_phys = 4 x (the input's physical channels), stage = parity (channel groups) + channel group, tap (R, DX) of
// parity (a, b) = w[2 (R - 1) + a, 2 (DX - 1) + b] (zero where R or DX is 0 or the index passes 2); cin_map over the INPUT's
// physical channels.
__global__ void conv3x3_h2_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                       int Cin_phys, int Cout, int nct, unsigned short* __restrict__ packed, int s2) {
    const size_t total = (size_t)(Cin_phys >> 4) * nct * 9 * 32 * 16;     // one thread per (c16, ct, tap, cout, channel)
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx & 15);
        size_t r = idx >> 4;
        const int i = (int)(r & 31);
        r >>= 5;
        const int tap = (int)(r % 9);
        r /= 9;
        const int ct = (int)(r % nct);
        const int c16 = (int)(r / nct);
        const int ncc = Cin_phys >> 6, par = s2 ? c16 / ncc : 0;          // (s2: stage = parity * channel groups + channel group)
        const int cphys = s2 ? (c16 - par * ncc) * 16 + ch : c16 * 16 + ch;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        const int co = ct * 32 + i;
        int wtap = tap;
        if (s2) {
            const int R = tap / 3, DX = tap - 3 * R, dy = 2 * (R - 1) + (par >> 1), dx = 2 * (DX - 1) + (par & 1);
            wtap = (R >= 1 && DX >= 1 && dy <= 2 && dx <= 2) ? dy * 3 + dx : -1;
        }
        float u = 0.f;
        if (wtap >= 0 && clog >= 0 && clog < Cin && co < Cout) u = w[((size_t)wtap * Cin + clog) * Cout + co];
        const _Float16 h = (_Float16)u;
        const _Float16 m = (_Float16)((u - (float)h) * 2048.f);
        const int tr = tap / 3, dx = tap - 3 * tr;
        unsigned short* base = packed + (((((size_t)c16 * 3 + tr) * nct + ct) * 3 + dx) * 4) * 256;      // 4 chunks x 32 couts x 8
        base[(ch >> 3) * 256 + i * 8 + (ch & 7)] = __builtin_bit_cast(unsigned short, h);
        base[(2 + (ch >> 3)) * 256 + i * 8 + (ch & 7)] = __builtin_bit_cast(unsigned short, m);
    }
}

extern "C" size_t pwc_conv3x3_h2_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    return (size_t)(Cin_phys >> 4) * ((Cout + 31) / 32) * 9 * (H2_TAPB / 4);
}

extern "C" int pwc_conv3x3_h2_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                       int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int nct = (Cout + 31) / 32;
    const size_t total = (size_t)(Cin_phys >> 4) * nct * 9 * 32 * 16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_h2_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, nct, reinterpret_cast<unsigned short*>(packed), 0);
    return pwc_launch_status();
}

// Weights of the stride-2 form (pwc_conv3x3_h2_stride2_f32): Cin_phys = the input's physical channels (cin_map over them).
extern "C" size_t pwc_conv3x3_h2_stride2_packed_floats(int Cin_phys, int Cout) {
    return pwc_conv3x3_h2_packed_floats(4 * Cin_phys, Cout);
}
extern "C" int pwc_conv3x3_h2_stride2_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                               int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int nct = (Cout + 31) / 32;
    const size_t total = (size_t)(Cin_phys >> 2) * nct * 9 * 32 * 16;        // (4 Cin_phys / 16 stages)
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_h2_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, 4 * Cin_phys, Cout, nct, reinterpret_cast<unsigned short*>(packed), 1);
    return pwc_launch_status();
}

// ---------------------------------------------------------------- tile variants and the launch plan
// variant: 1 = (2,2,2) 128 couts x 8 rows, 2 = (2,2,1) 64 couts x 16 rows, 3 = (3,1,1) 96 couts x 8 rows,
//          4 = (1,2,1) 32 couts x 16 rows, 5 = (1,2,2) 64 couts x 8 rows            (all x 32 columns)
struct H2Variant { int couts, rows, nm; };
static inline H2Variant h2_variant(int v) {
    switch (v) {
        case 1: return {128, 8, 12};
        case 2: return {64, 16, 12};
        case 3: return {96, 8, 9};
        case 4: return {32, 16, 6};
        default: return {64, 8, 6};
    }
}
// The lattice a dilation-d launch works on (see H2Cfg): sub-lattices of fewer than 24 columns are taken two at a time.
struct H2Geo { int dy, dx, xs, hs, ws; };
static inline H2Geo h2_geometry(int H, int W, int d, int Cout) {
    H2Geo g = {d, d, 1, (H + d - 1) / d, (W + d - 1) / d};
    if (g.ws < 24 && d % 2 == 0 && (W + d / 2 - 1) / (d / 2) >= 24 && Cout % 64 == 0) {      // (8-row tiles: 64 couts or more)
        g.dx = d / 2; g.xs = 2; g.ws = (W + g.dx - 1) / g.dx;
    }
    return g;
}
static inline long h2_blocks(int v, int N, const H2Geo& g, int Cout) {
    const H2Variant t = h2_variant(v);
    return (long)N * g.dy * g.dx * ((g.ws + 31) / 32) * ((g.hs + t.rows - 1) / t.rows) * (Cout / t.couts);
}
// The variant whose launch is estimated shortest: tiles per CU x (matrix instructions per tap and wave + 3.6), the 3.6 being
// the measured fixed part of a tap (profiles/r04_exp_h2.txt: 52 / 40.5 / 34 us per round of 8 stages for 12 / 9 / 6).
static int h2_plan(int N, int H, int W, int Cin_phys, int Cout, int dilation, long* blocks_out) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation < 1 || Cin_phys < 16 || (Cin_phys % 16) || Cout < 32 || (Cout % 32) || Cout > H2_MAX_COUT) return 0;
    const H2Geo geo = h2_geometry(H, W, dilation, Cout);
    int best = 0;
    double best_cost = 0.;
    long best_blocks = 0;
    const int order[5] = {1, 2, 3, 5, 4};          // (ties go to the wider tile)
    for (int oi = 0; oi < 5; ++oi) {
        const int v = order[oi];
        const H2Variant t = h2_variant(v);
        if (Cout % t.couts) continue;
        if (geo.xs == 2 && t.rows != 8) continue;  // (the 36-pixel patch exists for the 8-row tiles)
        const long nb = h2_blocks(v, N, geo, Cout);
        if (nb >= (1L << 31)) continue;
        const double cost = (nb <= 256 ? 1.0 : (double)nb / 256.0) * (t.nm + 3.6);      // stream-K: no rounding up to whole rounds
        if (!best || cost < best_cost) { best = v; best_cost = cost; best_blocks = nb; }
    }
    if (blocks_out) *blocks_out = best_blocks;
    return best;
}

// Stream-K (one workgroup per CU, equal shares of the (tile, stage) sequence) pays where it removes a partly filled last round
// (more tiles than CUs), or where the launch has so few tiles that a CU's share is at least 5.5 stages shorter than a whole
// tile's channel loop (measured, profiles/r04_exp_h2.txt: 192 -> 128 at 8 x 28 x 64 27 us against 34 us as 128 whole tiles; at
// 224 tiles the exchange costs more than the 1.5 stages it saves: 76 against 70 us).
static inline bool h2_split_pays(long ntiles, int nc16, int cus) {
    if (ntiles > cus) return ntiles * nc16 >= (long)H2_MIN_STAGES * cus;
    return ntiles < cus && ntiles * nc16 >= (long)H2_MIN_STAGES * cus && 2L * nc16 * (cus - ntiles) >= 11L * cus;
}

#ifdef PWC_HARNESS
static int h2_reserve_cus = 0;     // libpwc_hip_harness.so only (pwc_debug_h2_reserve_cus, scripts/exp_pipeline_reserve.py): CUs a stream-K launch leaves free
extern "C" void pwc_debug_h2_reserve_cus(int n) { h2_reserve_cus = n < 0 ? 0 : n; }
#else
constexpr int h2_reserve_cus = 0;
#endif

static int h2_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev] - h2_reserve_cus > 0 ? cus[dev] - h2_reserve_cus : 1;
}

// 1 where this kernel is the faster one for the shape (measured against conv3x3_wino4.hip / conv3x3_wino.hip on isolated layers:
// profiles/r04_exp_h2.txt): sub-lattices of at least 7 x 24 pixels (h2_geometry pairs narrower ones), a launch that fills at least three quarters of the CUs, two or more channel stages
// (32 -> 32 channels at 16 x 112 x 256: 48 us against F(2x2)'s 56).
extern "C" int pwc_conv3x3_h2_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    long nb = 0;
    if (!h2_plan(N, H, W, Cin_phys, Cout, dilation, &nb)) return 0;
    const H2Geo geo = h2_geometry(H, W, dilation, Cout);
    const int hs = geo.hs, ws = geo.ws;
    const bool filled = nb >= 192 || (nb >= 64 && h2_split_pays(nb, Cin_phys >> 4, h2_cu_count()));     // (the latter needs the workspace)
    return hs >= 7 && ws >= 24 && filled && Cin_phys >= 32 ? 1 : 0;
}

#ifdef PWC_HARNESS
static unsigned* h2_debug_counters = nullptr;      // scripts/exp_h2.hip only: s_memtime totals / self-test of the exchange
#else
constexpr unsigned* h2_debug_counters = nullptr;
#endif

template <int CT, int PT, int WCG, int ABL, int XS = 1, bool S2 = false>
static int h2_launch(H2Args& a, int hs, int ws, float* workspace, size_t workspace_floats, hipStream_t stream) {
    typedef H2Cfg<CT, PT, WCG, XS> C;
    a.tiles_x = (ws + 31) / 32; a.tiles_y = (hs + C::TR - 1) / C::TR; a.ncb = a.Cout / (32 * C::NCT);
    const long nblk = (long)a.N * a.dil_y * a.dil_x * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk * (a.Cin_phys >> 4) >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    // one workgroup per tile, or -- with a workspace and at least H2_MIN_STAGES stages per CU -- one workgroup per CU, each with an
    // equal share of the (tile, stage) sequence (a tile is then cut into as many pieces as ranges touch it)
    const int cus = h2_cu_count();
    int grid = a.ntiles;
    a.ws_partial = nullptr;
    const size_t part = (size_t)C::NCT * 32 * C::TR * 32;
    if (workspace && h2_split_pays(a.ntiles, a.Cin_phys >> 4, cus) && workspace_floats >= (size_t)cus * part) {
        grid = cus;
        a.ws_partial = workspace;
    }
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h2_kernel<CT, PT, WCG, ABL, XS, S2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    }
    hipLaunchKernelGGL((conv3x3_h2_kernel<CT, PT, WCG, ABL, XS, S2>), dim3((unsigned)grid), dim3(512), C::LDS, stream, a);
    return pwc_launch_status();
}

// variant 0 = h2_plan's choice
template <int ABL = 0>
static int h2_run(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs, int N, int H, int W,
                  int Cin_phys, int Cout, int dilation, int apply_act, float slope, pwc_stream_t stream, int variant = 0,
                  float* workspace = nullptr, size_t workspace_floats = 0, int stride = 1, const float* x2 = nullptr, int x2_cs = 0,
                  int Cin_a_phys = 0, uint32_t* status = nullptr, const float* x3 = nullptr, int x3_cs = 0, int Cin_b_phys = 0) {
    if (!x || !packed_w || !bias || !y) return PWC_EINVAL;
    if (x3 && !x2) return PWC_EINVAL;
    if (x2) {
        // (three operands: an operand's channel stride may be up to 12 channels short of its stage count x 16 -- its last stage then
        // runs into the next pixel's record, see H2Args::x3)
        const int slack = x3 ? 12 : 0;
        const int cb = x3 ? Cin_b_phys : Cin_phys - Cin_a_phys;
        if (Cin_a_phys <= 0 || Cin_a_phys >= Cin_phys || (Cin_a_phys % 16) || x_cs < Cin_a_phys - slack || x2_cs < cb - slack) return PWC_EINVAL;
        if ((x2_cs & 3) || !pwc_aligned16(x2)) return PWC_EALIGN;
        if ((long)H * W * x2_cs * 4 >= (long)H2_OOB) return PWC_ERANGE;
    }
    if (x3) {
        const int cc = Cin_phys - Cin_a_phys - Cin_b_phys;
        if (Cin_b_phys <= 0 || (Cin_b_phys % 16) || cc <= 0 || x3_cs < cc - 12 || dilation != 1) return PWC_EINVAL;
        if ((x3_cs & 3) || !pwc_aligned16(x3)) return PWC_EALIGN;
        if ((long)H * W * x3_cs * 4 >= (long)H2_OOB) return PWC_ERANGE;
    }
    if (reinterpret_cast<uintptr_t>(status) & 7u) return PWC_EALIGN;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1 || (stride != 1 && stride != 2)) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 32 || Cout > H2_MAX_COUT || (stride == 2 && (dilation != 1 || (H & 1) || (W & 1) || x2))) return PWC_EUNSUPPORTED;
    if ((!x2 && x_cs < Cin_phys) || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_w) || !pwc_aligned16(bias) ||
        !pwc_aligned16(workspace))
        return PWC_EALIGN;
    if (((long)H * W + W + 2) * x_cs * 4 >= (long)H2_OOB || (long)H * W * y_cs * 4 >= (long)H2_OOB) return PWC_ERANGE;
    H2Args a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.apply_act = apply_act; a.slope = slope;
    // stride 2: the launch is laid out over the OUTPUT (H / 2 x W / 2) and walks 4 x Cin_phys / 16 (channel group, parity) stages
    const bool s2 = stride == 2;
    a.Ho = s2 ? H / 2 : H; a.Wo = s2 ? W / 2 : W;
    if (s2) a.Cin_phys = 4 * Cin_phys;
    const H2Geo geo = h2_geometry(a.Ho, a.Wo, dilation, Cout);
    a.dil_y = geo.dy; a.dil_x = geo.dx;
    a.dbg = h2_debug_counters;
    a.x2 = x2; a.x2_cs = x2_cs; a.nc16_a = x2 ? Cin_a_phys >> 4 : a.Cin_phys >> 4; a.status = status;
    a.x3 = x3; a.x3_cs = x3_cs; a.nc16_b = x3 ? Cin_b_phys >> 4 : (a.Cin_phys >> 4) - a.nc16_a;
    const int hs = geo.hs, ws = geo.ws;
    if (variant == 0) variant = h2_plan(N, a.Ho, a.Wo, a.Cin_phys, Cout, dilation, nullptr);
    if (variant < 1 || variant > 5) return PWC_EUNSUPPORTED;
    if (Cout % h2_variant(variant).couts) return PWC_EUNSUPPORTED;
    if (s2) {
        switch (variant) {
            case 1: return h2_launch<2, 2, 2, ABL, 1, true>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            case 2: return h2_launch<2, 2, 1, ABL, 1, true>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            case 3: return h2_launch<3, 1, 1, ABL, 1, true>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            case 4: return h2_launch<1, 2, 1, ABL, 1, true>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            default: return h2_launch<1, 2, 2, ABL, 1, true>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
        }
    }
    if (geo.xs == 2) {
        switch (variant) {
            case 1: return h2_launch<2, 2, 2, ABL, 2>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            case 3: return h2_launch<3, 1, 1, ABL, 2>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            case 5: return h2_launch<1, 2, 2, ABL, 2>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
            default: return PWC_EUNSUPPORTED;
        }
    }
    switch (variant) {
        case 1: return h2_launch<2, 2, 2, ABL>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
        case 2: return h2_launch<2, 2, 1, ABL>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
        case 3: return h2_launch<3, 1, 1, ABL>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
        case 4: return h2_launch<1, 2, 1, ABL>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
        default: return h2_launch<1, 2, 2, ABL>(a, hs, ws, workspace, workspace_floats, (hipStream_t)stream);
    }
}

// Workspace of the stream-K form: one partial tile per CU, every word 0xFFFFFFFF before the first launch that uses the buffer
// (every launch leaves it so).  0 where the launch has no more tiles than CUs (no workspace is used).
extern "C" size_t pwc_conv3x3_h2_workspace_floats(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    long nb = 0;
    const int v = h2_plan(N, H, W, Cin_phys, Cout, dilation, &nb);
    const int cus = h2_cu_count();
    if (!v || !h2_split_pays(nb, Cin_phys >> 4, cus)) return 0;
    const H2Variant t = h2_variant(v);
    return (size_t)cus * t.couts * t.rows * 32;
}

extern "C" int pwc_conv3x3_h2_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y,
                                  int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                  int apply_act, float slope, float* workspace, size_t workspace_floats, pwc_stream_t stream) {
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, stream, 0,
                     workspace, workspace_floats);
}

// Round 5: the same launch with (a) the input given as two tensors -- the first Cin_a_phys physical channels from x, the rest
// from x2 (null: everything from x) -- and (b) a caller-owned status word pair (null: none), see include/pwc_hip.h.
extern "C" int pwc_conv3x3_h2_ex_f32(const float* x, int x_cs, int Cin_a_phys, const float* x2, int x2_cs,
                                     const float* packed_w, const float* bias, float* y, int y_cs, int N, int H, int W,
                                     int Cin_phys, int Cout, int dilation, int apply_act, float slope, float* workspace,
                                     size_t workspace_floats, uint32_t* status, pwc_stream_t stream) {
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, stream, 0,
                     workspace, workspace_floats, 1, x2, x2_cs, Cin_a_phys, status);
}

// Round 6: the input channels as THREE tensors over the same pixel grid: physical channels [0, Cin_a_phys) from x, the next
// Cin_b_phys from x2, the rest from x3 (Cin_a_phys, Cin_b_phys multiples of 16; dilation 1).  A channel stride may be up to 12
// channels SHORT of the operand's stage count x 16 (x_cs = 84 with Cin_a_phys = 96): the operand's last 16-channel stage then
// reads the first channels of the NEXT pixel's record (zeros behind the image's last pixel) and the caller's packed weights are
// zero there (cin_map = -1).  That is what makes a dense [cost volume 81 | flow 2 | 0] tensor of 336-byte records an operand.
extern "C" int pwc_conv3x3_h2_ex3_f32(const float* x, int x_cs, int Cin_a_phys, const float* x2, int x2_cs, int Cin_b_phys,
                                      const float* x3, int x3_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                                      int N, int H, int W, int Cin_phys, int Cout, int apply_act, float slope, float* workspace,
                                      size_t workspace_floats, uint32_t* status, pwc_stream_t stream) {
    if (!x2 || !x3) return PWC_EINVAL;
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, 1, apply_act, slope, stream, 0,
                     workspace, workspace_floats, 1, x2, x2_cs, Cin_a_phys, status, x3, x3_cs, Cin_b_phys);
}

// Stride 2 ('SAME', dilation 1, even H and W: the extractor's down-sampling layers, reference modules.py:57-60): the kernel's S2
// form (parity planes, see the kernel).  packed_w: pwc_conv3x3_h2_stride2_pack_f32; y is (N, H / 2, W / 2) at channel stride
// y_cs; workspace: pwc_conv3x3_h2_stride2_workspace_floats.  _supported: 1 where it is the faster kernel for the shape.  Measured
// in the forward (batch 8 = 16 images; profiles/r05_timeline_stride2.txt), us, S2 against pwc_conv3x3_f32 (fp32 MFMA):
//     16 -> 32 at 224 x 512: 61.0 - 62.3 / 64.8     32 -> 64 at 112 x 256: 45.7 - 47.0 / 57.8
//     64 -> 96 at  56 x 128: 49.4 - 50.3 / 42.9     96 -> 128 at 28 x 64:  31.7 - 32.2 / 30.5
// A stage carries 4 taps of matrix instructions instead of 9 and costs its fixed 5 - 6 us all the same (three parts, each waiting
// for weights that were requested one -- now nearly empty -- part earlier): with 4 C / 16 stages per tile the form pays only while
// the channel loop is short.  So: inputs of up to 32 channels.
extern "C" int pwc_conv3x3_h2_stride2_supported(int N, int H, int W, int Cin_phys, int Cout) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cin_phys % 16 || Cin_phys > 32) return 0;
    return pwc_conv3x3_h2_supported(N, H / 2, W / 2, 4 * Cin_phys, Cout, 1);
}

extern "C" size_t pwc_conv3x3_h2_stride2_workspace_floats(int N, int H, int W, int Cin_phys, int Cout) {
    if ((H & 1) || (W & 1)) return 0;
    return pwc_conv3x3_h2_workspace_floats(N, H / 2, W / 2, 4 * Cin_phys, Cout, 1);
}

extern "C" int pwc_conv3x3_h2_stride2_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y,
                                          int y_cs, int N, int H, int W, int Cin_phys, int Cout, int apply_act, float slope,
                                          float* workspace, size_t workspace_floats, uint32_t* status, pwc_stream_t stream) {
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, 1, apply_act, slope, stream, 0,
                     workspace, workspace_floats, 2, nullptr, 0, 0, status);
}

// The tile variant pwc_conv3x3_h2_f32 uses for a shape (1 - 5, see h2_variant; 0 = none fits), and the same convolution with
// the variant given (tests and tuning: every variant must give the same result on every shape whose Cout it divides).
extern "C" int pwc_conv3x3_h2_plan(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    return h2_plan(N, H, W, Cin_phys, Cout, dilation, nullptr);
}

extern "C" int pwc_conv3x3_h2_variant_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y,
                                          int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                          int apply_act, float slope, int variant, float* workspace, size_t workspace_floats,
                                          pwc_stream_t stream) {
    if (variant < 1 || variant > 5) return PWC_EINVAL;
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, stream, variant,
                     workspace, workspace_floats);
}
