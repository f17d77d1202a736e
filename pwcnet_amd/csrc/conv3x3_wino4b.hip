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
#include "pwc_common.h"
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
    const bool hi32 = lane >= 32;
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
    auto issue_patch = [&](int c16) {
#pragma unroll
        for (int i = 0; i < WB_PPW; ++i)
            if (!(ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xrsrc, (lptr_t)(sm + (wave + WB_NW * i < WB_NBP - 1 ? wave + WB_NW * i : WB_NBP - 1) * 1024), 16,
                    (int)p_voff[i], c16 * 64, 0, 0);
    };
    // weights: part P = the positions of row 3 ah + P of both row halves: two runs of 6 positions = 2 x 18 pieces of 1 KB,
    // everything but the lane's 16 bytes is wave-uniform (scalar offset)
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_lane4 = wave < 4 ? u_lane : WB_OOB;            // piece wave + 32 exists for waves 0..3 only
    const int u_cb = cb * WB_U_BYTES;
    auto issue_u = [&](int c16, int part) {
        const int sbase = c16 * a.ncb * WB_U_BYTES + u_cb;
#pragma unroll
        for (int j = 0; j < WB_UPW; ++j) {
            const int i = wave + WB_NW * j;                            // uniform
            const int run = i >= 18 ? 1 : 0;
            const int rel = (18 * run + 6 * part) * WB_UX + (i - 18 * run) * 1024;
            const bool real = j < WB_UPW - 1 || wave < 4;
            if (!(ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lptr_t)(sm + (real ? WB_PATCH_BYTES + rel : WB_DUMMY)), 16,
                                                         (int)(j < WB_UPW - 1 ? u_lane : u_lane4), sbase + rel, 0, 0);
        }
    };
#define WB_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#define WB_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- this lane's patch reads (conv3x3_wino4.hip): record (4 trow + i) * 36 + (j & 3) * 9 + (j >> 2) + tc, chunk
    // fq ^ pswz(py); pswz flips between window rows i < 4 and i >= 4: two per-lane bases, everything else an immediate
    const int trow = 2 * g + trl;
    const float* pb_lo = smem + ((4 * trow) * WB_PS + tc) * 16 + ((fq ^ wb_pswz(4 * trow)) << 2);
    const float* pb_hi = smem + ((4 * trow) * WB_PS + tc) * 16 + ((fq ^ wb_pswz(4 * trow + 4)) << 2);
    // ... and its weight fragments: row (cout) fr of a 16-cout tile, 96 bytes per row:
    // [uh set 0 | uh set 1 | um set 0 | um set 1 | ul set 0 | ul set 1], set f = channels {4f..4f+3, 4f+8..4f+11}
    const char* const ub = sm + WB_PATCH_BYTES + (18 * ah + 3 * bh) * WB_UX + fr * 96 + (fq & 1) * 16;
    const int u3 = fq < 2 ? 64 : 0;                 // MFMA 3: A = [ul | uh]

    f32x4 acc[9][2];
    auto stage = [&](auto first, int c16) {
        constexpr bool FIRST = decltype(first)::value;
        const bool has_next = c16 + 1 < nc16;
        // in flight here (oldest first): patch(c), weight parts 0 and 1 of c
        WB_WAIT_VM(2 * WB_UPW);                      // patch(c) landed
        WB_BAR();                                    // ... for every wave; part 2 of c-1 fully read
        issue_u(c16, 2);

        // ---- row pass of the input transform: this wave's rows a = 3ah .. 3ah+2 of  B^T d, all six columns
        f32x4 V[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4 dd[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                dd[i] = *reinterpret_cast<const f32x4*>((i < 4 ? pb_lo : pb_hi) + (i * WB_PS + (j & 3) * 9 + (j >> 2)) * 16);
            if (ABL & 64) {
                V[0][j] = dd[0] + dd[3]; V[1][j] = dd[1] + dd[4]; V[2][j] = dd[2] + dd[5];
            } else if (ah == 0) {
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
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j) asm("" : "+v"(V[r][j]));      // keep the packed ops (see conv3x3_wino.hip)

        // ---- part P: column pass of row P (columns b = 3bh .. 3bh+2), split, lane exchange, 3 positions x 2 cout
        // tiles x 3 MFMAs
        auto part = [&](auto pc) {
            constexpr int P = decltype(pc)::value;
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
            }
#pragma unroll
            for (int bi = 0; bi < 3; ++bi) {
                asm("" : "+v"(o[bi]));
                // ---- exact split  v = h + m + l  (round-to-nearest bf16 terms; the residuals are exact in fp32)
                const f32x4 v = o[bi];
                unsigned H0, H1, M0, M1b, L0, L1;
                {
                    H0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, pwc_bf16x2));
                    H1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, pwc_bf16x2));
                    if (ABL & 32) {
                        M0 = H0; M1b = H1; L0 = H0; L1 = H1;
                    } else {
                        const f32x4 hf = {__builtin_bit_cast(float, H0 << 16), __builtin_bit_cast(float, H0 & 0xffff0000u),
                                          __builtin_bit_cast(float, H1 << 16), __builtin_bit_cast(float, H1 & 0xffff0000u)};
                        const f32x4 r1 = v - hf;
                        M0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r1[0], r1[1]}, pwc_bf16x2));
                        M1b = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r1[2], r1[3]}, pwc_bf16x2));
                        const f32x4 mf = {__builtin_bit_cast(float, M0 << 16), __builtin_bit_cast(float, M0 & 0xffff0000u),
                                          __builtin_bit_cast(float, M1b << 16), __builtin_bit_cast(float, M1b & 0xffff0000u)};
                        const f32x4 r2 = r1 - mf;
                        L0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r2[0], r2[1]}, pwc_bf16x2));
                        L1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r2[2], r2[3]}, pwc_bf16x2));
                    }
                }
                // ---- trade halves with lane ^ 32: B1 = [vh | vm], B3 = [vh | vl] along k (8 channels per lane)
                pwc_u32x4 b1, b3;
                if (ABL & 32) {
                    b1 = pwc_u32x4{H0, H1, M0, M1b};
                    b3 = pwc_u32x4{H0, H1, L0, L1};
                } else {
                    const pwc_u32x2 s0 = __builtin_amdgcn_permlane32_swap(H0, M0, false, false);
                    const pwc_u32x2 s1 = __builtin_amdgcn_permlane32_swap(H1, M1b, false, false);
                    // s[0]: lanes < 32 own h, lanes >= 32 m of lane - 32;  s[1]: lanes < 32 h of lane + 32, lanes >= 32 own m
                    b1 = pwc_u32x4{s0[0], s1[0], s0[1], s1[1]};
                    const pwc_u32x2 t0 = __builtin_amdgcn_permlane32_swap(s0[0], L0, false, false);
                    const pwc_u32x2 t1 = __builtin_amdgcn_permlane32_swap(s1[0], L1, false, false);
                    // t[0]: lanes < 32 own h, lanes >= 32 l of lane - 32;  t[1] lanes >= 32: own l
                    b3 = pwc_u32x4{t0[0], t1[0], hi32 ? t0[1] : s0[1], hi32 ? t1[1] : s1[1]};
                }
                const pwc_bf16x8 B1 = __builtin_bit_cast(pwc_bf16x8, b1), B3 = __builtin_bit_cast(pwc_bf16x8, b3);
                const int XL = P * 3 + bi;
                const char* const up = ub + (6 * P + bi) * WB_UX;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const pwc_bf16x8 A1 = *reinterpret_cast<const pwc_bf16x8*>(up + ct * 1536);
                    const pwc_bf16x8 A2 = *reinterpret_cast<const pwc_bf16x8*>(up + ct * 1536 + 32);
                    const pwc_bf16x8 A3 = *reinterpret_cast<const pwc_bf16x8*>(up + ct * 1536 + u3);
                    if (ABL & 4) {
                        if (FIRST) acc[XL][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                        asm volatile("" ::"v"(A1), "v"(A2), "v"(A3), "v"(B1), "v"(B3));
                        continue;
                    }
                    f32x4 c = FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[XL][ct];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B3, c, 0, 0, 0);     // ul vh + uh vl
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, c, 0, 0, 0);     // um vh + um vm
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, c, 0, 0, 0);     // uh vh + uh vm
                    acc[XL][ct] = c;
                }
            }
        };
        WB_WAIT_VM(2 * WB_UPW);                      // weight part 0 of c landed (parts 1, 2 may be in flight)
        WB_BAR();                                    // ... for every wave; patch(c) fully read
        if (has_next) issue_patch(c16 + 1);
        part(std::integral_constant<int, 0>{});
        if (has_next) WB_WAIT_VM(WB_UPW + WB_PPW); else WB_WAIT_VM(WB_UPW);   // part 1 landed
        WB_BAR();                                    // ... for every wave; part 0 fully read
        if (has_next) issue_u(c16 + 1, 0);
        part(std::integral_constant<int, 1>{});
        if (has_next) WB_WAIT_VM(WB_PPW + WB_UPW); else WB_WAIT_VM(0);        // part 2 landed
        WB_BAR();                                    // ... for every wave; part 1 fully read
        if (has_next) issue_u(c16 + 1, 1);
        part(std::integral_constant<int, 2>{});
    };
    issue_patch(0);
    issue_u(0, 0);
    issue_u(0, 1);
    stage(std::true_type{}, 0);
    for (int c16 = 1; c16 < nc16; ++c16) stage(std::false_type{}, c16);

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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwc_u32x4, yv), yrsrc, (int)vo, 0, 0);
            }
        }
    }
#undef WB_WAIT_VM
#undef WB_BAR
}
#undef WBSUB
#undef WBFMA

// ---------------------------------------------------------------- weight transform, split and packing
// U_xi = (G g G^T)[a][b], xi = 6a + b, in double (G as conv3x3_wino4.hip); u = h + m + l with round-to-nearest bf16
// terms (|u - h - m - l| <= 2^-27 |u|); image [c16][cout group][xi][cout 32][term 3][set 2][8]: set f holds the channels
// {4f..4f+3, 4f+8..4f+11} of the 16-channel stage in that order (the k order the lane exchange of the kernel produces).
__global__ void conv3x3_wino4b_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                           int Cin_phys, int Cout, int ncb, unsigned short* __restrict__ packed) {
    const size_t total = (size_t)(Cin_phys >> 4) * ncb * 36 * 32 * 16;      // one thread per (c16, cg, xi, cout, channel)
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int s16 = (int)(idx & 15);             // slot: set f = s16 >> 3, element e = s16 & 7
        size_t r = idx >> 4;
        const int co32 = (int)(r & 31);
        r >>= 5;
        const int xi = (int)(r % 36);
        r /= 36;
        const int cg = (int)(r % ncb);
        const int c16 = (int)(r / ncb);
        const int f = s16 >> 3, e = s16 & 7;
        const int ch = e < 4 ? 4 * f + e : 8 + 4 * f + (e - 4);
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
        unsigned short* row = packed + ((((size_t)c16 * ncb + cg) * 36 + xi) * 32 + co32) * 48;
        row[s16] = __builtin_bit_cast(unsigned short, h);
        row[16 + s16] = __builtin_bit_cast(unsigned short, m);
        row[32 + s16] = __builtin_bit_cast(unsigned short, l);
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
                                  hipFuncAttributeMaxDynamicSharedMemorySize, WB_LDS);
    }
    hipLaunchKernelGGL((conv3x3_wino4b_kernel<ABL>), dim3((unsigned)a.ntiles), dim3(WB_T), WB_LDS, stream, a);
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
