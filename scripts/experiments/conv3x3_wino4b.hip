// conv3x3_wino4b.hip -- 3x3 stride-1 'SAME' convolution by Winograd F(4x4, 3x3) whose 36 per-position GEMMs run on the
// BF16 matrix pipe of gfx950 with EXACT three-way operand splits (round 4).
//
// Replaces the same tf.layers.Conv2D(...,(3,3),(1,1),'same',dilation_rate=d) + tf.nn.leaky_relu calls as
// conv3x3_wino4.hip (reference modules.py:266-268 `optflow_l/conv2d .. conv2d_3`, modules.py:306-323 `context/conv2d*`).
//
// Why: on gfx950 an fp32 MFMA runs at the fp32 VECTOR rate and excludes every other instruction on its SIMD (DESIGN.md
// 3.4): the fp32 F(4x4) kernel sits at 0.38 of the fp32 matrix peak under a 0.62 instruction-mix cap.  A bf16 MFMA runs at
// 16x the rate and overlaps VALU work of the other wave of its SIMD.  An fp32 value is EXACTLY the sum of three bf16 values
// (round-to-nearest splits: x = h + m + l, 8 + 8 + 8 significand bits, same exponent range -- no scaling, no overflow
// hazard), a product of two bf16 values is exact in fp32, and of the nine cross products of two split operands the six
// kept here carry everything above 2^-24 |u v| (dropped: m.l + l.m + l.l <= 2^-24 + 2^-32 of the product, below the
// rounding error of ONE fp32 fused multiply-add, of which the fp32 MFMA chain performs one per product):
//
//     u v  ~=  uh vh + uh vm + um vh + um vm + uh vl + ul vh            (accumulated in fp32 by the matrix pipe)
//
// The six products of a 16-channel stage are THREE v_mfma_f32_16x16x32_bf16: the split terms are laid out along K,
//     MFMA 1:  A = [uh | uh]   B = [vh | vm]        MFMA 2:  A = [um | um]   B = [vh | vm]       MFMA 3:  A = [ul | uh]   B = [vh | vl]
// (k = 0..15 | 16..31).  6 bf16 products at 16x the fp32 rate = 2.67x the fp32 MFMA throughput, and the transform /
// split VALU work of one wave runs under the MFMAs of the other wave of its SIMD.  Measured error against a float64
// convolution: see profiles/r04_wino4b_numerics.txt (per layer, next to the fp32 kernels on the same inputs).
//
// Work decomposition (512 threads = 8 waves, ONE workgroup per CU, two waves per SIMD):
//   workgroup = 4 x 8 Winograd tiles (16 x 32 output pixels) x 32 output channels;
//   wave      = (tile group g: tile rows 2g, 2g+1 = 16 tiles = the N side of the MFMA;
//                position block (ah, bh): rows a = 3ah..3ah+2 and columns b = 3bh..3bh+2 of the 6 x 6 transformed tile =
//                9 of the 36 positions) x both 16-cout tiles: 18 accumulator tiles = 72 registers.  The input transform of
//                a (tile, channel) is done ONCE per workgroup and position (the fp32 kernel: once per 16 couts), in the
//                registers of the lane that feeds it to the matrix pipe -- the transformed tile never touches LDS.
//   lane      = (tile j = lane & 15, k-slot q = lane >> 4): reads the 6 x 6 input pixels of its tile for channels
//               4q..4q+3 (36 ds_read_b128), row pass for its three rows a, column pass for its three columns b, then per
//               position: split the 4 values into h / m / l bf16 pairs (v_cvt_pk_bf16_f32, exact residuals by
//               v_pk_add_f32) and trade halves with lane ^ 32 (v_permlane32_swap): lanes 0-31 end up with the h terms of
//               8 channels (k = 0..15 side of the MFMA), lanes 32-63 with the m (or l) terms of the same 8 channels.
//   LDS per 16-channel stage: the raw 18 x 34 pixel patch (conv3x3_wino4.hip's image, 42 KB) and the split transformed
//   weights of the workgroup's 32 couts, [position 36][cout 32][uh 16 | um 16 | ul 16] bf16 = 108 KB, a LINEAR copy of
//   the packed global image (96-byte rows are conflict-free for ds_read_b128 as they are); both filled by
//   buffer_load_dwordx4 ... lds; single-buffered, the weights in three parts that are re-fetched for the next stage as
//   soon as every wave has read them (the pipeline of conv3x3_wino4.hip).
#pragma once
#include "../../pwcnet_amd/csrc/pwc_common.h"
#include <type_traits>

typedef __bf16 pwc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pwc_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned pwc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pwc_u32x2 __attribute__((ext_vector_type(2)));

struct Wino4bArgs {
    const float* x;
    const void* up;      // packed split weights [c16][cout group of 32][xi 36][cout 32][48 bf16]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // 16x32-pixel blocks per (sub-)image, cout groups of 32
    int dil;
    int ntiles;
};

constexpr unsigned WB_OOB = 0x7FFF0000u;
constexpr int WB_NW = 8, WB_T = 64 * WB_NW;
constexpr int WB_PS = 36;                    // patch records per patch row (conv3x3_wino4.hip's image)
constexpr int WB_PH = 18, WB_PW = 34;
constexpr int WB_PPW = 6;                    // patch DMA pieces per wave and stage: 48 requests for the 41 blocks
constexpr int WB_NBP = 42;                   // 41 blocks + block 41 that swallows the surplus (out-of-range) requests
constexpr int WB_PATCH_BYTES = WB_NBP * 1024;
constexpr int WB_UX = 3072;                  // bytes of split weights per position: 32 couts x (16 + 16 + 16) bf16
constexpr int WB_U_BYTES = 36 * WB_UX;
constexpr int WB_UPW = 5;                    // weight DMA pieces per wave and part: 40 requests for the 36 KB of a part
constexpr int WB_DUMMY = WB_PATCH_BYTES + WB_U_BYTES;   // 1 KB that swallows the 4 surplus weight requests of a part
constexpr int WB_LDS = WB_DUMMY + 1024;      // 154 624 B: one workgroup per CU
constexpr int WB_XCH = WB_NW * 12 * 1024;    // output exchange: 8 waves x 12 slots x 64 lanes x 16 B
static_assert(WB_XCH <= WB_U_BYTES && WB_LDS <= 160 * 1024, "the exchange reuses the weight area");

__device__ __forceinline__ int wb_pswz(int py) { return ((py >> 2) & 1) << 1; }   // = w4_pswz

// three rows (half = 0: rows 0..2, half = 1: rows 3..5) of  B^T e,
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
#define WBSUB(p, q) __builtin_elementwise_fma((q), M1, (p))                       /* p - q, packable */
#define WBFMA(x, c, y) __builtin_elementwise_fma((x), f32x4{c, c, c, c}, (y))     /* x * c + y */

// ABL (scripts/exp_wino4b.hip only; 0 in the library): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA,
// 32 = no split / lane exchange (the h terms stand in for m and l), 64 = no transform arithmetic
template <int ABL = 0>
__global__ __launch_bounds__(WB_T, 2) void conv3x3_wino4b_kernel(const Wino4bArgs a) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave >> 2, qb = wave & 3;         // tile group, position block
    const int ah = qb >> 1, bh = qb & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int trl = fr >> 3, tc = fr & 7;           // tile (2g + trl, tc)
    float m1s;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1s));   // -1.0f the optimiser cannot see through (see conv3x3_wino.hip)
    const f32x4 M1 = {m1s, m1s, m1s, m1s};

    const int d = a.dil;
    const int nc16 = a.Cin_phys >> 4;

    // ---- block decode: cout group fastest, XCD-aware (the cout groups of a pixel block share its patch in one L2)
    int lb = pwc_xcd_remap(blockIdx.x, a.ntiles);
    const int cb = lb % a.ncb;
    int rest = lb / a.ncb;
    const int bx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int by = rest % a.tiles_y;
    rest /= a.tiles_y;
    const int sub = rest % (d * d);
    const int n = rest / (d * d);
    const int ry = sub / d, rx = sub - ry * d;      // pixel sub-lattice (y mod d, x mod d) of a dilated conv
    const int y0 = by * 16, x0 = bx * 32;           // output origin of the block, in sub-lattice coordinates
    const int n0 = cb * 32;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.up, 0, nc16 * a.ncb * WB_U_BYTES, 0x00020000);

    // ---- LDS-DMA bookkeeping
    unsigned p_voff[WB_PPW];                        // patch: per-lane byte offsets fixed over the channel loop
#pragma unroll
    for (int i = 0; i < WB_PPW; ++i) {
        const int rec = (wave + WB_NW * i) * 16 + (lane >> 2);
        const int py = rec / WB_PS, rem = rec - py * WB_PS;
        const int q = rem / 9, ci = rem - q * 9;
        const int px = 4 * ci + q;
        const int yy = ry + d * (y0 - 1 + py), xx = rx + d * (x0 - 1 + px);
        const int ch = (lane & 3) ^ wb_pswz(py);                       // source chunk for this LDS slot
        const bool ok = py < WB_PH && px < WB_PW && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        p_voff[i] = ok ? (unsigned)(((yy * a.W + xx) * a.x_cs + ch * 4) * 4) : WB_OOB;
    }
    auto issue_patch = [&](int c16, int i0, int i1) {      // pieces i0 .. i1-1 of this wave's WB_PPW
#pragma unroll
        for (int i = 0; i < WB_PPW; ++i)
            if (i >= i0 && i < i1 && !(ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xrsrc, (lptr_t)(sm + (wave + WB_NW * i < WB_NBP - 1 ? wave + WB_NW * i : WB_NBP - 1) * 1024), 16,
                    (int)p_voff[i], c16 * 64, 0, 0);
    };
    // weights: part P = the positions of row 3 ah + P of both row halves: two runs of 6 positions = 2 x 18 pieces of 1 KB,
    // everything but the lane's 16 bytes is wave-uniform (scalar offset)
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_lane4 = wave < 4 ? u_lane : WB_OOB;            // piece wave + 32 exists for waves 0..3 only
    const int u_cb = cb * WB_U_BYTES;
    auto issue_u = [&](int c16, int part, int j0, int j1) {   // pieces j0 .. j1-1 of this wave's WB_UPW
        const int sbase = c16 * a.ncb * WB_U_BYTES + u_cb;
#pragma unroll
        for (int j = 0; j < WB_UPW; ++j) {
            const int i = wave + WB_NW * j;                            // uniform
            const int run = i >= 18 ? 1 : 0;
            const int rel = (18 * run + 6 * part) * WB_UX + (i - 18 * run) * 1024;
            const bool real = j < WB_UPW - 1 || wave < 4;
            if (j >= j0 && j < j1 && !(ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lptr_t)(sm + (real ? WB_PATCH_BYTES + rel : WB_DUMMY)), 16,
                                                         (int)(j < WB_UPW - 1 ? u_lane : u_lane4), sbase + rel, 0, 0);
        }
    };
#define WB_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#define WB_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // ABL & 128 (harness only): s_memtime stamps of wave-lane 0 at the four points of every phase top, kept in the 9 KB of LDS
    // above the stage buffers and dumped by workgroup 0 (a.y is not written in this mode)
    int stamp_n = 0;
    auto stamp = [&]() {
        if (ABL & 128) {
            const unsigned long long tm = __builtin_readcyclecounter();
            if (lane == 0 && stamp_n < 144) *reinterpret_cast<unsigned long long*>(sm + WB_LDS + (wave * 144 + stamp_n) * 8) = tm;
            ++stamp_n;
        }
    };

    // ---- this lane's patch reads (conv3x3_wino4.hip): record (4 trow + i) * 36 + (j & 3) * 9 + (j >> 2) + tc, chunk
    // fq ^ pswz(py); pswz flips between window rows i < 4 and i >= 4: two per-lane bases, everything else an immediate
    const int trow = 2 * g + trl;
    const float* pb_lo = smem + ((4 * trow) * WB_PS + tc) * 16 + ((fq ^ wb_pswz(4 * trow)) << 2);
    const float* pb_hi = smem + ((4 * trow) * WB_PS + tc) * 16 + ((fq ^ wb_pswz(4 * trow + 4)) << 2);
    // ... and its weight fragments: per position 3072 bytes = [cout tile 2][k-slot 4][cout 16] x (uh 4 | um 4 channels) bf16, then
    // [k-slot 4][cout 16] x (ul of tile 0 | ul of tile 1): three conflict-free ds_read_b128 of 1 KB per position
    const char* const ub = sm + WB_PATCH_BYTES + (18 * ah + 3 * bh) * WB_UX + fq * 256 + fr * 16;

    f32x4 acc[9][2];
    f32x4 V[3][6];
#define WB_SCHED() __builtin_amdgcn_sched_barrier(0)
    // ---- row pass of the input transform, one column j of the window: this wave's rows a = 3ah .. 3ah+2 of  B^T d.
    // The six pixels of a column are READ one column ahead of their use (a wave has one partner on its SIMD: a read that is
    // waited for at once costs its whole latency).
    auto loadcol = [&](int j, f32x4 (&dd)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            dd[i] = *reinterpret_cast<const f32x4*>((i < 4 ? pb_lo : pb_hi) + (i * WB_PS + (j & 3) * 9 + (j >> 2)) * 16);
    };
    auto rowcol = [&](auto ahc, int j, const f32x4 (&dd)[6]) {     // ahc: the wave's row half as a compile-time constant (one basic block)
        constexpr int AH = decltype(ahc)::value;
        if (ABL & 64) {
            V[0][j] = dd[0] + dd[3]; V[1][j] = dd[1] + dd[4]; V[2][j] = dd[2] + dd[5];
        } else if (AH == 0) {
            V[0][j] = WBFMA(dd[0], 4.f, WBFMA(dd[2], -5.f, dd[4]));
            const f32x4 s = dd[1] + dd[2], tt = dd[3] + dd[4], u = WBSUB(dd[1], dd[2]), v = WBSUB(dd[4], dd[3]);
            V[1][j] = WBFMA(s, -4.f, tt);
            V[2][j] = WBFMA(u, 4.f, v);
        } else {
            const f32x4 p = WBSUB(dd[4], dd[2]), q = WBSUB(dd[3], dd[1]);
            V[0][j] = WBFMA(q, 2.f, p);
            V[1][j] = WBFMA(q, -2.f, p);
            V[2][j] = WBFMA(dd[1], 4.f, WBFMA(dd[3], -5.f, dd[5]));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) asm("" : "+v"(V[r][j]));              // keep the packed ops (see conv3x3_wino.hip)
    };
    // ---- one position.  Weights: [cout tile][k-slot][cout] x (uh 4 | um 4), then [k-slot][cout] x (ul tile 0 | ul tile 1):
    // three ds_read_b128, issued one position ahead.  Split: v = h + m + l exactly (round-to-nearest bf16 terms; the residuals
    // are exact in fp32).  k-slot layout of a lane (tile n, q): [term X of channels 4q..4q+3 | term Y of the same]:
    //   MFMA 3: A = [ul | uh]  B = [vh | vl]     MFMA 2: A = [uh | um]  B = [vm | vm]     MFMA 1: A = [uh | um]  B = [vh | vh]
    auto loadA = [&](const char* up, pwc_u32x4 (&A)[3]) {
        A[0] = *reinterpret_cast<const pwc_u32x4*>(up);
        A[1] = *reinterpret_cast<const pwc_u32x4*>(up + 1024);
        A[2] = *reinterpret_cast<const pwc_u32x4*>(up + 2048);
    };
    auto split = [&](const f32x4 v, unsigned (&S)[6]) {
        S[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, pwc_bf16x2));
        S[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, pwc_bf16x2));
        if (ABL & 32) {
            S[2] = S[0]; S[3] = S[1]; S[4] = S[0]; S[5] = S[1];
        } else {
            f32x4 hf = {__builtin_bit_cast(float, S[0] << 16), __builtin_bit_cast(float, S[0] & 0xffff0000u),
                        __builtin_bit_cast(float, S[1] << 16), __builtin_bit_cast(float, S[1] & 0xffff0000u)};
            asm("" : "+v"(hf));                      // (a vector the optimiser cannot take apart: v_pk_add_f32)
            const f32x4 r1 = v - hf;
            S[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r1[0], r1[1]}, pwc_bf16x2));
            S[3] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r1[2], r1[3]}, pwc_bf16x2));
            f32x4 mf = {__builtin_bit_cast(float, S[2] << 16), __builtin_bit_cast(float, S[2] & 0xffff0000u),
                        __builtin_bit_cast(float, S[3] << 16), __builtin_bit_cast(float, S[3] & 0xffff0000u)};
            asm("" : "+v"(mf));
            const f32x4 r2 = r1 - mf;
            S[4] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r2[0], r2[1]}, pwc_bf16x2));
            S[5] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r2[2], r2[3]}, pwc_bf16x2));
        }
    };
    auto mfmas = [&](auto first, int XL, const pwc_u32x4 (&A)[3], const unsigned (&S)[6]) {
        constexpr bool FIRST = decltype(first)::value;
        typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
        const u32x6 W = {S[0], S[1], S[0], S[1], S[4], S[5]};       // (h, h) and (h, l) are overlapping register windows of it
        const pwc_bf16x8 Bhh = __builtin_bit_cast(pwc_bf16x8, __builtin_shufflevector(W, W, 0, 1, 2, 3));
        const pwc_bf16x8 Bhl = __builtin_bit_cast(pwc_bf16x8, __builtin_shufflevector(W, W, 2, 3, 4, 5));
        const pwc_bf16x8 Bmm = __builtin_bit_cast(pwc_bf16x8, pwc_u32x4{S[2], S[3], S[2], S[3]});
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const pwc_bf16x8 Ahm = __builtin_bit_cast(pwc_bf16x8, A[ct]);
            const pwc_bf16x8 Alh = __builtin_bit_cast(pwc_bf16x8, pwc_u32x4{A[2][2 * ct], A[2][2 * ct + 1], A[ct][0], A[ct][1]});
            if (ABL & 4) {
                if (FIRST) acc[XL][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" ::"v"(Ahm), "v"(Alh), "v"(Bhh), "v"(Bmm), "v"(Bhl));
                continue;
            }
            f32x4 c = FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[XL][ct];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Alh, Bhl, c, 0, 0, 0);      // ul vh + uh vl
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahm, Bmm, c, 0, 0, 0);      // uh vm + um vm
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ahm, Bhh, c, 0, 0, 0);      // uh vh + um vh
            acc[XL][ct] = c;
        }
    };
    // ---- part P of a stage: column pass of row P (columns b = 3bh .. 3bh+2), then its three positions; in part 2 the
    // row pass of the NEXT stage (its patch is resident since two phases) is spread between the positions.  Order within
    // the part, fixed by scheduling barriers: the LDS reads of step k+1 are issued before the arithmetic of step k.
    // ---- part P of a stage: column pass of row P (columns b = 3bh .. 3bh+2), then its three positions; in part 2 (NR) the
    // row pass of the NEXT stage (its patch is resident since two phases) is spread between the positions.  Software pipeline
    // inside the part: the LDS reads of step k+1 are issued first, then one scheduling region holds the split of position
    // k+1, the row pass of two columns and the six MFMAs of position k -- independent work the scheduler interleaves (two
    // waves of a SIMD that run the same code arrive at their MFMAs together: VALU and matrix work have to alternate INSIDE a
    // wave; measured scripts/exp_pos.hip: 515 -> 430 cycles per position and wave pair).
    auto part = [&](auto first, auto pc, auto nr, auto ahc, auto&& dma) {
        constexpr int P = decltype(pc)::value;
        constexpr bool NR = decltype(nr)::value;
        const char* const up = ub + (6 * P) * WB_UX;
        pwc_u32x4 A[3];
        f32x4 d0[6], d1[6];
        loadA(up, A);
        if (NR) { loadcol(0, d0); loadcol(1, d1); }
        f32x4 o[3];
        {
            const f32x4 e0 = V[P][0], e1 = V[P][1], e2 = V[P][2], e3 = V[P][3], e4 = V[P][4], e5 = V[P][5];
            if (ABL & 64) {
                o[0] = e0 + e3; o[1] = e1 + e4; o[2] = e2 + e5;
            } else if (bh == 0) {
                const f32x4 s = e1 + e2, tt = e3 + e4, u = WBSUB(e1, e2), v = WBSUB(e4, e3);
                o[0] = WBFMA(e0, 4.f, WBFMA(e2, -5.f, e4));
                o[1] = WBFMA(s, -4.f, tt);
                o[2] = WBFMA(u, 4.f, v);
            } else {
                const f32x4 p = WBSUB(e4, e2), q = WBSUB(e3, e1);
                o[0] = WBFMA(q, 2.f, p);
                o[1] = WBFMA(q, -2.f, p);
                o[2] = WBFMA(e1, 4.f, WBFMA(e3, -5.f, e5));
            }
#pragma unroll
            for (int bi = 0; bi < 3; ++bi) asm("" : "+v"(o[bi]));
        }
        unsigned S0[6], S1[6];
        split(o[0], S0);
        WB_SCHED();
        // step 0: one region = split of position 1, two columns of the next stage's row pass, the MFMAs of position 0; then the
        // LDS reads of step 1 (their latency is covered by the split and the row pass that open the next region)
        dma(0);
        split(o[1], S1);
        if (NR) { rowcol(ahc, 0, d0); rowcol(ahc, 1, d1); }
        mfmas(first, P * 3 + 0, A, S0);
        WB_SCHED();
        loadA(up + WB_UX, A);
        if (NR) { loadcol(2, d0); loadcol(3, d1); }
        WB_SCHED();
        // step 1
        dma(1);
        split(o[2], S0);
        if (NR) { rowcol(ahc, 2, d0); rowcol(ahc, 3, d1); }
        mfmas(first, P * 3 + 1, A, S1);
        WB_SCHED();
        loadA(up + 2 * WB_UX, A);
        if (NR) { loadcol(4, d0); loadcol(5, d1); }
        WB_SCHED();
        // step 2
        dma(2);
        if (NR) { rowcol(ahc, 4, d0); rowcol(ahc, 5, d1); }
        mfmas(first, P * 3 + 2, A, S0);
    };
    // ---- pipeline.  LDS: one patch buffer, the weights of a stage in three parts (fixed regions).  A phase = one part.
    // At the top of a phase (after its barrier) the region read in the PREVIOUS phase is re-fetched for its next use,
    // two phases ahead: top of part 0: weights part 2 of this stage + the next stage's patch (the patch is only read in
    // part 2); top of part 1: part 0 of the next stage; top of part 2: part 1 of the next stage.
    auto stage = [&](auto first, int c16) {
        const bool has_next = c16 + 1 < nc16;
        // in flight here (oldest first): weight parts 0 and 1 of c
        stamp();
        WB_WAIT_VM(WB_UPW);                          // part 0 landed
        stamp();
        WB_BAR();                                    // ... for every wave; part 2 of c-1 and patch(c) fully read
        stamp();
        stamp();
        // the 5 (+ 6) fetches of a phase are issued in three slots between its positions: a burst of all waves at the top
        // of the phase keeps every wave in the issue queue of the 64 B/clk fetch path (measured: 600 - 1400 cycles per phase)
        part(first, std::integral_constant<int, 0>{}, std::false_type{}, std::integral_constant<int, 0>{}, [&](int slot) {
            issue_u(c16, 2, slot == 0 ? 0 : slot == 1 ? 2 : 4, slot == 0 ? 2 : slot == 1 ? 4 : 5);
            if (has_next) issue_patch(c16 + 1, 2 * slot, 2 * slot + 2);
        });
        stamp();
        if (has_next) WB_WAIT_VM(WB_UPW + WB_PPW); else WB_WAIT_VM(WB_UPW);   // part 1 landed
        stamp();
        WB_BAR();                                    // ... for every wave; part 0 fully read
        stamp();
        stamp();
        part(first, std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<int, 0>{}, [&](int slot) {
            if (has_next) issue_u(c16 + 1, 0, slot == 0 ? 0 : slot == 1 ? 2 : 4, slot == 0 ? 2 : slot == 1 ? 4 : 5);
        });
        stamp();
        if (has_next) WB_WAIT_VM(WB_UPW); else WB_WAIT_VM(0);                 // part 2 and patch(c+1) landed
        stamp();
        WB_BAR();                                    // ... for every wave; part 1 fully read
        stamp();
        stamp();
        auto dma2 = [&](int slot) {
            if (has_next) issue_u(c16 + 1, 1, slot == 0 ? 0 : slot == 1 ? 2 : 4, slot == 0 ? 2 : slot == 1 ? 4 : 5);
        };
        if (!has_next) part(first, std::integral_constant<int, 2>{}, std::false_type{}, std::integral_constant<int, 0>{}, dma2);
        else if (ah == 0) part(first, std::integral_constant<int, 2>{}, std::true_type{}, std::integral_constant<int, 0>{}, dma2);
        else part(first, std::integral_constant<int, 2>{}, std::true_type{}, std::integral_constant<int, 1>{}, dma2);
    };
    issue_patch(0, 0, WB_PPW);
    issue_u(0, 0, 0, WB_UPW);
    issue_u(0, 1, 0, WB_UPW);
    WB_WAIT_VM(2 * WB_UPW);                          // patch(0) landed
    WB_BAR();
    {
        f32x4 d0[6], d1[6];
        loadcol(0, d0);
#pragma unroll
        for (int j = 0; j < 6; j += 2) {
            loadcol(j + 1, d1);
            WB_SCHED();
            if (ah == 0) rowcol(std::integral_constant<int, 0>{}, j, d0); else rowcol(std::integral_constant<int, 1>{}, j, d0);
            if (j + 2 < 6) loadcol(j + 2, d0);
            WB_SCHED();
            if (ah == 0) rowcol(std::integral_constant<int, 0>{}, j + 1, d1); else rowcol(std::integral_constant<int, 1>{}, j + 1, d1);
        }
    }
    stage(std::true_type{}, 0);
    for (int c16 = 1; c16 < nc16; ++c16) stage(std::false_type{}, c16);
    stamp();
#undef WB_SCHED

    // ---- output transform  Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]: this wave holds the
    // 3 x 3 block (ah, bh) of M and forms its 4 x 4 partial sums; wave qb finishes output row qb of the tile and gets
    // the three other blocks' partials of that row through LDS (12 slots of 16 bytes per lane and cout tile).
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (size_t)n * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
    char* const xch = sm + WB_PATCH_BYTES;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f32x4 Yp[4][4];                              // [i'][j'] partial sums over this block
        {
            f32x4 Z[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 m0 = acc[r * 3 + 0][ct], m1 = acc[r * 3 + 1][ct], m2 = acc[r * 3 + 2][ct];
                if (bh == 0) {                       // b = 0, 1, 2: columns [1 0 0 0], [1 1 1 1], [1 -1 1 -1] of A^T
                    const f32x4 s = m1 + m2, dd = WBSUB(m1, m2);
                    Z[r][0] = m0 + s; Z[r][1] = dd; Z[r][2] = s; Z[r][3] = dd;
                } else {                             // b = 3, 4, 5: [1 2 4 8], [1 -2 4 -8], [0 0 0 1]
                    const f32x4 s = m0 + m1, dd = WBSUB(m0, m1);
                    Z[r][0] = s; Z[r][1] = dd * 2.f; Z[r][2] = s * 4.f; Z[r][3] = WBFMA(dd, 8.f, m2);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ah == 0) {
                    const f32x4 s = Z[1][j] + Z[2][j], dd = WBSUB(Z[1][j], Z[2][j]);
                    Yp[0][j] = Z[0][j] + s; Yp[1][j] = dd; Yp[2][j] = s; Yp[3][j] = dd;
                } else {
                    const f32x4 s = Z[0][j] + Z[1][j], dd = WBSUB(Z[0][j], Z[1][j]);
                    Yp[0][j] = s; Yp[1][j] = dd * 2.f; Yp[2][j] = s * 4.f; Yp[3][j] = WBFMA(dd, 8.f, Z[2][j]);
                }
            }
        }
        WB_BAR();                                    // every wave is past its last LDS read of the stage / of cout tile 0
        f32x4 own[4];
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            if (I == qb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) own[j] = Yp[I][j];
            } else {
                char* dst = xch + (((4 * g + I) * 3 + (qb < I ? qb : qb - 1)) * 4) * 1024 + lane * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + j * 1024) = Yp[I][j];
            }
        }
        WB_BAR();
        const char* src = xch + (wave * 12) * 1024 + lane * 16;
        const int co = n0 + ct * 16 + fq * 4;
        if (co < a.Cout) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
            const int py = ry + d * (y0 + 4 * trow + qb), px0 = rx + d * (x0 + 4 * tc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 yv = own[j];
#pragma unroll
                for (int s = 0; s < 3; ++s) yv += *reinterpret_cast<const f32x4*>(src + (s * 4 + j) * 1024);
                yv += b4;
                if (a.apply_act) {                   // tf.nn.leaky_relu = max(v, slope * v)
                    const f32x4 sv = yv * a.slope;
                    yv[0] = fmaxf(yv[0], sv[0]); yv[1] = fmaxf(yv[1], sv[1]);
                    yv[2] = fmaxf(yv[2], sv[2]); yv[3] = fmaxf(yv[3], sv[3]);
                }
                const int px = px0 + j * d;
                const unsigned vo = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.y_cs + co) * 4) : WB_OOB;
                if (!(ABL & 128)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwc_u32x4, yv), yrsrc, (int)vo, 0, 0);
            }
        }
    }
    if (ABL & 128) {
        stamp();
        WB_BAR();
        if (blockIdx.x == 0)
            for (int i = t; i < WB_NW * 144 * 2; i += WB_T) reinterpret_cast<unsigned*>(a.y)[i] = reinterpret_cast<unsigned*>(sm + WB_LDS)[i];
    }
#undef WB_WAIT_VM
#undef WB_BAR
}
#undef WBSUB
#undef WBFMA

// ---------------------------------------------------------------- weight transform, split and packing
// U_xi = (G g G^T)[a][b], xi = 6a + b, in double (G as conv3x3_wino4.hip); u = h + m + l with round-to-nearest bf16
// terms (|u - h - m - l| <= 2^-27 |u|); image [c16][cout group][xi] x 3072 bytes: [cout tile 2][k-slot 4][cout 16][uh 4 | um 4],
// then [k-slot 4][cout 16][ul of tile 0: 4 | ul of tile 1: 4]; k-slot q holds channels 4q..4q+3 of the 16-channel stage
// (what lane (n, q) of the kernel transforms).
__global__ void conv3x3_wino4b_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                           int Cin_phys, int Cout, int ncb, unsigned short* __restrict__ packed) {
    const size_t total = (size_t)(Cin_phys >> 4) * ncb * 36 * 32 * 16;      // one thread per (c16, cg, xi, cout, channel)
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx & 15);
        size_t r = idx >> 4;
        const int co32 = (int)(r & 31);
        r >>= 5;
        const int xi = (int)(r % 36);
        r /= 36;
        const int cg = (int)(r % ncb);
        const int c16 = (int)(r / ncb);
        const int cphys = c16 * 16 + ch;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        const int co = cg * 32 + co32;
        double u = 0.;
        if (clog >= 0 && clog < Cin && co < Cout) {
            const int ua = xi / 6, ubb = xi % 6;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    u += G[ua][p] * G[ubb][q] * (double)w[((size_t)(p * 3 + q) * Cin + clog) * Cout + co];
        }
        const __bf16 h = (__bf16)(float)u;
        const double r1 = u - (double)(float)h;
        const __bf16 m = (__bf16)(float)r1;
        const double r2 = r1 - (double)(float)m;
        const __bf16 l = (__bf16)(float)r2;
        const int ct = co32 >> 4, i = co32 & 15, q = ch >> 2, e = ch & 3;
        unsigned short* pos = packed + (((size_t)c16 * ncb + cg) * 36 + xi) * (WB_UX / 2);
        pos[ct * 512 + (q * 16 + i) * 8 + e] = __builtin_bit_cast(unsigned short, h);
        pos[ct * 512 + (q * 16 + i) * 8 + 4 + e] = __builtin_bit_cast(unsigned short, m);
        pos[1024 + (q * 16 + i) * 8 + ct * 4 + e] = __builtin_bit_cast(unsigned short, l);
    }
}

extern "C" size_t pwc_conv3x3_wino4b_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    return (size_t)(Cin_phys >> 4) * ((Cout + 31) / 32) * (WB_U_BYTES / 4);
}

extern "C" int pwc_conv3x3_wino4b_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                           int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int ncb = (Cout + 31) / 32;
    const size_t total = (size_t)(Cin_phys >> 4) * ncb * 36 * 32 * 16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_wino4b_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, ncb, reinterpret_cast<unsigned short*>(packed));
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_wino4b_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation < 1 || Cin_phys < 32 || Cin_phys > 1024 || (Cin_phys % 16) || Cout < 32 || (Cout % 32)) return 0;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    if (hs < 14 || ws < 28) return 0;
    const long blocks = (long)N * dilation * dilation * ((hs + 15) / 16) * ((ws + 31) / 32) * (Cout / 32);
    const double fill = (double)hs * ws / ((double)(((hs + 15) / 16) * 16) * (((ws + 31) / 32) * 32));
    return blocks >= 128 && fill >= 0.8 ? 1 : 0;
}

template <int ABL>
static int wino4b_launch(const Wino4bArgs& a, hipStream_t stream) {
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4b_kernel<ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, WB_LDS + ((ABL & 128) ? 9216 : 0));
    }
    hipLaunchKernelGGL((conv3x3_wino4b_kernel<ABL>), dim3((unsigned)a.ntiles), dim3(WB_T), WB_LDS + ((ABL & 128) ? 9216 : 0), stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_wino4b_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                      int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                      int apply_act, float slope, pwc_stream_t stream) {
    if (!x || !packed_u || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 32) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_u) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)H * W * x_cs * 4 >= (long)WB_OOB || (long)H * W * y_cs * 4 >= (long)WB_OOB) return PWC_ERANGE;
    if ((long)(Cin_phys >> 4) * (Cout / 32) * WB_U_BYTES >= (long)WB_OOB) return PWC_ERANGE;
    Wino4bArgs a;
    a.x = x; a.up = packed_u; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.apply_act = apply_act; a.slope = slope;
    a.dil = dilation;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    a.tiles_x = (ws + 31) / 32; a.tiles_y = (hs + 15) / 16; a.ncb = Cout / 32;
    const long nblk = (long)N * dilation * dilation * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    return wino4b_launch<0>(a, (hipStream_t)stream);
}
