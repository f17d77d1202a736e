// conv3x3_wino4.hip -- 3x3 stride-1 'SAME' convolution by Winograd F(4x4, 3x3) on the fp32 MFMA units of gfx950.
//
// Replaces the tf.layers.Conv2D(...,(3,3),(1,1),'same',dilation_rate=d) + tf.nn.leaky_relu calls with Cin, Cout
// >= 64 on the full-resolution pyramid level (reference modules.py:267-268 `optflow_4/conv2d .. conv2d_3`,
// modules.py:308-316 `context/conv2d_1 .. conv2d_3`) -- 1.7 ms of the 3.6 ms forward on F(2x2,3x3)
// (conv3x3_wino.hip).  36 multiplies per 4x4 outputs instead of 64: 1.78x fewer MFMA instructions.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      per 6x6 input tile d, 4x4 output Y
//
// Numerics: the transforms have the constants of Lavin & Gray (|B^T| entries up to 5, G down to 1/24); measured
// on the whole network (scripts/exp_f4x4_numerics.py, fp32 emulation against a float64 forward) F(4x4) on these
// layers changes the flow error by x1.0 - 1.4 -- the level of the direct fp32 convolution (profiles/
// r03_f4x4_numerics.txt).  U = G g G^T is computed in double and rounded once.
//
// Work decomposition (256 threads = 4 waves, TWO workgroups per CU = two waves per SIMD):
//   workgroup = 4 x 8 Winograd tiles (16 x 32 output pixels) x 16 output channels;
//   wave      = (tile group g: tile rows 2g, 2g+1 = 16 tiles = one MFMA column block;
//                position half h: rows a = 3h .. 3h+2 of the 6 x 6 transformed tile = 18 of the 36 positions):
//               18 accumulator tiles = 72 registers.  Splitting the POSITIONS over waves halves the accumulators AND the
//               input-transform work per wave (the row pass produces 3 of 6 rows).
//               History: 4 waves with two cout tiles each (364 registers, one wave per SIMD) 256 us on the 128 -> 128
//               layer; 8 waves = (g, h, cout tile) in ONE workgroup of 32 couts 247 us; this form -- the same waves as
//               two independent workgroups of 16 couts, each with its own copy of the patch -- 249 us there, but 197
//               against 210 us on 128 -> 96 and 107 against 117 us on 96 -> 64 (cout granularity 16; the two
//               workgroups of a CU are not in lock step).  A 16 x 64-pixel form with both cout tiles per wave (one
//               transform per 144 MFMAs) needs 256 registers to the last one, spilled 21 and measured 281 us.
//   lane      = (tile j = lane & 15, k-slot q = lane >> 4): reads the 6 x 6 input pixels of its tile for channels
//               4q..4q+3 (36 ds_read_b128), transforms them IN REGISTERS -- the result is the MFMA B-operand
//               fragment V_xi[k = 4q+s][tile j] -- and at the end holds its 18 M_xi: the column pass of the
//               output transform is register-local, the row pass needs the other half's partial sums, exchanged
//               through LDS once per tile (wave h finishes output rows 2h, 2h+1).
//   LDS per 16-channel stage: the raw 18 x 34 pixel patch (64-byte records, see w4_rec) and the transformed
//   weights U[xi 36][16 cout][16 ch] (pre-swizzled by the packer), both filled by buffer_load_dwordx4 ... lds;
//   out-of-image pixels are out-of-range buffer offsets = the zeros of the SAME padding.  One stage buffer (78 KB),
//   fetched in three parts (patch, positions {0-5, 18-23}, ...) that are each re-fetched for the next stage as soon
//   as they have been read.
// What bounds it: on gfx950 nothing overlaps an fp32 MFMA on its SIMD; per wave and stage 72 MFMAs carry
// 180 packed VALU instructions of transform, 54 ds_read_b128 and 20 fetch pieces: a cap of 0.6 on the MFMA fraction.
#pragma once
#include "pwc_common.h"
#include <type_traits>
#ifndef WINO_SYNC
#define WINO_SYNC() pwc_lds_barrier()
#endif

struct Wino4Args {
    const float* x;
    const float* up;     // packed transformed weights [xi 36][c16][Cout_pad][16], chunk-swizzled
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // 16x32-pixel blocks per (sub-)image, cout blocks of 16
    int dil;
    int ntiles;
};

constexpr unsigned W4_OOB = 0x7FFF0000u;
constexpr int W4_PS = 36;                    // patch records per patch row: 4 quarter rows (px & 3) of 9 (px >> 2)
constexpr int W4_PH = 18, W4_PW = 34;
constexpr int W4_NW = 4, W4_T = 64 * W4_NW;
constexpr int W4_PPW = 11;                   // patch DMA pieces per wave and stage: 44 requests for the 41 blocks (648 records of 64 B)
constexpr int W4_NBP = 42;                   // ... of the patch image, + block 41 that swallows the three surplus (out-of-range) requests
constexpr int W4_PREC = W4_NBP * 16;
constexpr int W4_NBU = 36;                   // 16-row DMA blocks of the weights: 36 positions x 16 couts
constexpr int W4_STAGE = (W4_PREC + W4_NBU * 16) * 16;     // floats per LDS stage: 79 872 B -- two workgroups per CU
constexpr int W4_XCH = W4_NW * 64 * 36;      // floats of the output exchange (4 waves x 64 lanes x (32 + 4 pad)): 36 864 B
static_assert(W4_XCH <= W4_STAGE && 2 * W4_STAGE * 4 <= 160 * 1024, "the exchange reuses the stage buffer; two workgroups share a CU");

__device__ __forceinline__ int w4_wswz(int row) { return (4 - ((row >> 2) & 3)) & 3; }   // weight rows (as conv3x3_wino.hip)
// Patch image: pixel (py, px) of the 18 x 34 patch sits in record py * 36 + (px & 3) * 9 + (px >> 2) -- the 8 tile
// columns a wave reads at one (i, j) of the 6 x 6 window are then CONSECUTIVE records -- and its 16-byte chunk c at
// slot c ^ w4_pswz(py): the two tile rows of a wave differ in (py >> 2) & 1, so every ds_read_b128 lane group
// ({tile row 0, k-slot 0}, {row 1, slot 0}, {row 0, slot 1}, {row 1, slot 1} x 4 consecutive records) covers the 16
// slots of a 256-byte bank row exactly once.
__device__ __forceinline__ int w4_pswz(int py) { return ((py >> 2) & 1) << 1; }

// ABL (scripts/exp_wino4.hip only; 0 in the library): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA, 64 = no
// transform arithmetic, 8 / 16 = the same patch / weight requests from a few hot cache lines
template <int ABL = 0>
__global__ __launch_bounds__(W4_T, 2) void conv3x3_wino4_kernel(const Wino4Args a) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave >> 1, h = wave & 1;          // tile group, position half
    const int fr = lane & 15, fq = lane >> 4;
    const int trl = fr >> 3, tc = fr & 7;           // tile (2g + trl, tc)
    float m1s;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1s));   // -1.0f the optimiser cannot see through (see conv3x3_wino.hip)
    const f32x4 M1 = {m1s, m1s, m1s, m1s};
#define W4SUB(p, q) __builtin_elementwise_fma((q), M1, (p))    /* p - q, packable */
#define W4FMA(x, c, y) __builtin_elementwise_fma((x), f32x4{c, c, c, c}, (y))   /* x * c + y */

    const int d = a.dil;
    const int Cout_pad = (a.Cout + 15) & ~15;
    const int nc16 = a.Cin_phys >> 4;
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.up, 0, 36 * a.Cin_phys * Cout_pad * 4, 0x00020000);

    // ---- block decode: cout block fastest, XCD-aware (the cout blocks of a pixel block share its patch in one L2)
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
    const int n0 = cb * 16;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);

    // ---- LDS-DMA bookkeeping: per-lane byte offsets fixed over the channel loop, the stage in the scalar offset
    constexpr int PPW = W4_PPW;                     // patch pieces per wave: blocks wave, wave + 4, ... (>= 41: nothing to fetch)
    unsigned p_voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int rec = (wave + W4_NW * i) * 16 + (lane >> 2);
        const int py = rec / W4_PS, rem = rec - py * W4_PS;
        const int q = rem / 9, ci = rem - q * 9;
        const int px = 4 * ci + q;
        const int yy = ry + d * (y0 - 1 + py), xx = rx + d * (x0 - 1 + px);
        const int ch = (lane & 3) ^ w4_pswz(py);                       // source chunk for this LDS slot
        const bool ok = py < W4_PH && px < W4_PW && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        p_voff[i] = ok ? (unsigned)(((yy * a.W + xx) * a.x_cs + ch * 4) * 4) : W4_OOB;
    }
    // weights, in three parts P of 6 positions per half: wave w fetches positions 18 (w & 1) + 6 P + 3 (w >> 1) + i, i = 0..2
    const int u_xi0 = 18 * (wave & 1) + 3 * (wave >> 1);
    const int u_co = n0 + (lane >> 2);
    const unsigned u_voff = (u_co < Cout_pad) ? (unsigned)((((u_xi0 * nc16) * Cout_pad + u_co) * 16 + (lane & 3) * 4) * 4) : W4_OOB;
    const int u_step = nc16 * Cout_pad * 64;        // bytes between consecutive positions
    auto issue_patch = [&](int c16) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (!(ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xrsrc, (lptr_t)(smem + (wave + W4_NW * i < W4_NBP - 1 ? wave + W4_NW * i : W4_NBP - 1) * 256), 16,
                    (ABL & 8) ? (int)(p_voff[i] & 4095u) : (int)p_voff[i], (ABL & 8) ? 0 : c16 * 64, 0, 0);
    };
    auto issue_u = [&](int c16, int part) {
        const int us = c16 * Cout_pad * 64;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int xi = u_xi0 + 6 * part + i;    // uniform
            if (!(ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lptr_t)(smem + (W4_NBP + xi) * 256), 16,
                                                         (int)u_voff, (ABL & 16) ? 0 : us + (6 * part + i) * u_step, 0, 0);
        }
    };
#define W4_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

    // ---- this lane's patch reads: record (4 trow + i) * 36 + (j & 3) * 9 + (j >> 2) + tc, chunk fq ^ pswz(py);
    // pswz flips between window rows i < 4 and i >= 4: two per-lane bases, everything else is an immediate
    const int trow = 2 * g + trl;
    const float* pb_lo = smem + ((4 * trow) * W4_PS + tc) * 16 + ((fq ^ w4_pswz(4 * trow)) << 2);
    const float* pb_hi = smem + ((4 * trow) * W4_PS + tc) * 16 + ((fq ^ w4_pswz(4 * trow + 4)) << 2);
    const int u_off = W4_PREC * 16 + ((18 * h) * 16 + fr) * 16 + ((fq ^ w4_wswz(fr)) << 2);   // this wave's A-fragment rows

    f32x4 acc[18];
    auto stage = [&](auto first, int c16) {
        constexpr bool FIRST = decltype(first)::value;
        const bool has_next = c16 + 1 < nc16;
        // in flight here (oldest first): patch(c), weight parts 0 and 1 of c
        W4_WAIT_VM(6);                               // patch(c) landed
        WINO_SYNC();                             // ... for every wave; part 2 of c-1 fully read
        issue_u(c16, 2);

        // ---- input transform, this wave's three rows a = 3h .. 3h+2 of  V = B^T d B  (both cout-tile waves of a
        // (tile group, half) pair do it: the registers it would take to hand V over do not exist)
        //   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        f32x4 V[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                // row pass, column j of the window
            f32x4 dd[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                dd[i] = *reinterpret_cast<const f32x4*>((i < 4 ? pb_lo : pb_hi) + (i * W4_PS + (j & 3) * 9 + (j >> 2)) * 16);
            if (ABL & 64) {                          // (ablation: no transform arithmetic)
                V[0][j] = dd[0] + dd[3]; V[1][j] = dd[1] + dd[4]; V[2][j] = dd[2] + dd[5];
            } else if (h == 0) {
                V[0][j] = W4FMA(dd[0], 4.f, W4FMA(dd[2], -5.f, dd[4]));
                const f32x4 s = dd[1] + dd[2], tt = dd[3] + dd[4], u = W4SUB(dd[1], dd[2]), v = W4SUB(dd[4], dd[3]);
                V[1][j] = W4FMA(s, -4.f, tt);
                V[2][j] = W4FMA(u, 4.f, v);
            } else {
                const f32x4 p = W4SUB(dd[4], dd[2]), q = W4SUB(dd[3], dd[1]);
                V[0][j] = W4FMA(q, 2.f, p);
                V[1][j] = W4FMA(q, -2.f, p);
                V[2][j] = W4FMA(dd[1], 4.f, W4FMA(dd[3], -5.f, dd[5]));
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j) asm("" : "+v"(V[r][j]));      // keep the packed ops (see conv3x3_wino.hip)
        if (!(ABL & 64)) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {            // column pass, in place
                const f32x4 e0 = V[r][0], e1 = V[r][1], e2 = V[r][2], e3 = V[r][3], e4 = V[r][4], e5 = V[r][5];
                const f32x4 s = e1 + e2, tt = e3 + e4, u = W4SUB(e1, e2), v = W4SUB(e4, e3);
                const f32x4 p = W4SUB(e4, e2), q = W4SUB(e3, e1);
                V[r][0] = W4FMA(e0, 4.f, W4FMA(e2, -5.f, e4));
                V[r][1] = W4FMA(s, -4.f, tt);
                V[r][2] = W4FMA(u, 4.f, v);
                V[r][3] = W4FMA(q, 2.f, p);
                V[r][4] = W4FMA(q, -2.f, p);
                V[r][5] = W4FMA(e1, 4.f, W4FMA(e3, -5.f, e5));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) asm("" : "+v"(V[r][j]));
        }

        // ---- 18 positions x 4 k-steps of MFMA in three parts of 6 positions; two positions alternate (a dependent
        // 16x16x4 MFMA would wait 8 cycles for its accumulator)
        auto mfma_part = [&](int part) {
#pragma unroll
            for (int xp = 6 * part; xp < 6 * part + 6; xp += 2) {
                f32x4 wf[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) wf[e] = *reinterpret_cast<const f32x4*>(smem + u_off + (xp + e) * 16 * 16);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = kk >> 1, xl = xp + (kk & 1);
                    if (ABL & 4) {
                        if (FIRST && k == 0) acc[xl] = f32x4{0.f, 0.f, 0.f, 0.f};
                        asm volatile("" ::"v"(wf[kk & 1][k]), "v"(V[xl / 6][xl % 6][k]));
                        continue;
                    }
                    const f32x4 c = (FIRST && k == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[xl];
                    acc[xl] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk & 1][k], V[xl / 6][xl % 6][k], c, 0, 0, 0);
                }
            }
        };
        W4_WAIT_VM(6);                               // weight part 0 of c landed (parts 1, 2 may be in flight)
        WINO_SYNC();                             // ... for every wave; patch(c) fully read
        if (has_next) issue_patch(c16 + 1);
        mfma_part(0);
        if (has_next) W4_WAIT_VM(3 + PPW); else W4_WAIT_VM(3);   // part 1 landed (part 2, patch(c+1) may be in flight)
        WINO_SYNC();                             // ... for every wave; part 0 fully read
        if (has_next) issue_u(c16 + 1, 0);
        mfma_part(1);
        if (has_next) W4_WAIT_VM(PPW + 3); else W4_WAIT_VM(0);   // part 2 landed (patch(c+1), part 0 of c+1 may be in flight)
        WINO_SYNC();                             // ... for every wave; part 1 fully read
        if (has_next) issue_u(c16 + 1, 1);
        mfma_part(2);
    };
    issue_patch(0);
    issue_u(0, 0);
    issue_u(0, 1);
    stage(std::true_type{}, 0);
    for (int c16 = 1; c16 < nc16; ++c16) stage(std::false_type{}, c16);

    // ---- output transform  Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
    // Column pass (over b) on this wave's three rows a, then the partial row pass; the other half's partial sums come
    // through LDS: wave h finishes output rows 2h, 2h+1 of the tile and sends the partials of the other two.
    // (h is wave-uniform; the two halves are separate instantiations so that every array index is static)
    auto epilogue = [&](auto h_c) {
        constexpr int HH = decltype(h_c)::value;
        f32x4 Yp[4][4];                             // [i'][j'] partial sums over a in this half
        {
            f32x4 Z[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 m0 = acc[r * 6 + 0], m1 = acc[r * 6 + 1], m2 = acc[r * 6 + 2];
                const f32x4 m3 = acc[r * 6 + 3], m4 = acc[r * 6 + 4], m5 = acc[r * 6 + 5];
                const f32x4 s12 = m1 + m2, d12 = W4SUB(m1, m2), s34 = m3 + m4, d34 = W4SUB(m3, m4);
                Z[r][0] = m0 + s12 + s34;
                Z[r][1] = W4FMA(d34, 2.f, d12);
                Z[r][2] = W4FMA(s34, 4.f, s12);
                Z[r][3] = W4FMA(d34, 8.f, d12) + m5;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (HH == 0) {                       // a = 0, 1, 2: columns [1 1 1], [0 1 -1], [0 1 1], [0 1 -1] of A^T
                    const f32x4 s = Z[1][j] + Z[2][j], dd = W4SUB(Z[1][j], Z[2][j]);
                    Yp[0][j] = Z[0][j] + s;
                    Yp[1][j] = dd;
                    Yp[2][j] = s;
                    Yp[3][j] = dd;
                } else {                             // a = 3, 4, 5: [1 1 0], [2 -2 0], [4 4 0], [8 -8 1]
                    const f32x4 s = Z[0][j] + Z[1][j], dd = W4SUB(Z[0][j], Z[1][j]);
                    Yp[0][j] = s;
                    Yp[1][j] = dd * 2.f;
                    Yp[2][j] = s * 4.f;
                    Yp[3][j] = W4FMA(dd, 8.f, Z[2][j]);
                }
            }
        }
        WINO_SYNC();                             // every wave is past its last LDS read of the stage
        {
            // exchange with the wave of the other half (same tile group = wave ^ 1): 8 f32x4 per lane,
            // 32 + 4 floats per lane (conflict-free b128 accesses)
            float* dst = smem + ((wave ^ 1) * 64 + lane) * 36;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (i * 4 + j) * 4) = Yp[2 * (1 - HH) + i][j];
        }
        WINO_SYNC();
        const float* src = smem + (wave * 64 + lane) * 36;
        const int co = n0 + fq * 4;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (size_t)n * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
        if (co < a.Cout) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
            const int py0 = ry + d * (y0 + 4 * trow + 2 * HH), px0 = rx + d * (x0 + 4 * tc);   // real coordinates of this wave's first output
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 yv = Yp[2 * HH + i][j] + *reinterpret_cast<const f32x4*>(src + (i * 4 + j) * 4) + b4;
                    if (a.apply_act) {               // tf.nn.leaky_relu = max(v, slope * v)
                        const f32x4 sv = yv * a.slope;
                        yv[0] = fmaxf(yv[0], sv[0]); yv[1] = fmaxf(yv[1], sv[1]);
                        yv[2] = fmaxf(yv[2], sv[2]); yv[3] = fmaxf(yv[3], sv[3]);
                    }
                    const int py = py0 + i * d, px = px0 + j * d;
                    const unsigned vo = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.y_cs + co) * 4) : W4_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), yrsrc, (int)vo, 0, 0);
                }
        }
    };
    if (h == 0) epilogue(std::integral_constant<int, 0>{});
    else epilogue(std::integral_constant<int, 1>{});
#undef W4SUB
#undef W4FMA
#undef W4_WAIT_VM
}

// ---------------------------------------------------------------- weight transform + packing
// packed[xi][c16][cout_pad][16]: U_xi = (G g G^T)[a][b], xi = 6a + b,
//   G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]   (double, rounded once),
// chunk-swizzled like conv3x3_wino.hip's image; cin_map as in pwc_conv3x3_pack_f32.
__global__ void conv3x3_wino4_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                          int Cin_phys, int Cout, int Cout_pad, float* __restrict__ packed) {
    const size_t total = (size_t)36 * Cin_phys * Cout_pad;
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e16 = (int)(idx & 15);
        size_t r = idx >> 4;
        const int co = (int)(r % Cout_pad);
        r /= Cout_pad;
        const int c16 = (int)(r % (Cin_phys >> 4));
        const int xi = (int)(r / (Cin_phys >> 4));
        const int jpos = e16 >> 2, e = e16 & 3;
        const int j = jpos ^ w4_wswz(co);
        const int cphys = c16 * 16 + j * 4 + e;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        double u = 0.;
        if (clog >= 0 && clog < Cin && co < Cout) {
            const int ua = xi / 6, ub = xi % 6;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    u += G[ua][p] * G[ub][q] * (double)w[((size_t)(p * 3 + q) * Cin + clog) * Cout + co];
        }
        packed[idx] = (float)u;
    }
}

extern "C" size_t pwc_conv3x3_wino4_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    return (size_t)36 * Cin_phys * ((Cout + 15) & ~15);
}

extern "C" int pwc_conv3x3_wino4_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                          int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int Cout_pad = (Cout + 15) & ~15;
    const size_t total = (size_t)36 * Cin_phys * Cout_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_wino4_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, Cout_pad, packed);
    return pwc_launch_status();
}

// Where F(4x4) pays (measured against conv3x3_wino.hip: scripts/exp_wino4.hip on isolated layers, profiles/r03_exp_wino4.txt,
// and in the forward): launches that fill the GPU with 16 x 32-pixel blocks x 16 couts and whose channel loop is long --
// Cin_phys >= 48 and Cout >= 32 on maps (or dilation sub-lattices) of at least 14 x 28 pixels that fill their blocks to
// 80 % and make 256 workgroups.  x1.15 - 1.2 on the undilated 160/128 -> 128, 128 -> 96 and 96 -> 64 layers and on d = 8;
// d = 2, 4 are worth x1.05, the 48/64-channel layers +0.6 % of the forward together; 7-row sub-lattices (d = 16:
// conv3x3_wino.hip's SPLIT geometry) and shorter launches stay on F(2x2), and so do the 288 ... 576-channel inputs of the
// dense-connection estimators (use_dc=True: 205 pairs/s with F(4x4) on them, 212 - 215 without).  (The rule was swept in the
// forward: 256 / 384 / 512 / 1024 workgroups, with and without d = 2, 4, Cin >= 96 / 64 / 48 / 32, Cout >= 64 / 32.)
extern "C" int pwc_conv3x3_wino4_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation < 1 || Cin_phys < 48 || Cin_phys > 256 || (Cin_phys % 16) || Cout < 32 || (Cout % 16)) return 0;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    if (hs < 14 || ws < 28) return 0;
    const long blocks = (long)N * dilation * dilation * ((hs + 15) / 16) * ((ws + 31) / 32) * (Cout / 16);
    const double fill = (double)hs * ws / ((double)(((hs + 15) / 16) * 16) * (((ws + 31) / 32) * 32));
    return blocks >= 256 && fill >= 0.8 ? 1 : 0;
}

template <int ABL>
static int wino4_launch(const Wino4Args& a, hipStream_t stream) {
    const size_t lds = (size_t)W4_STAGE * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((conv3x3_wino4_kernel<ABL>), dim3((unsigned)a.ntiles), dim3(W4_T), lds, stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_wino4_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                     int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                     int apply_act, float slope, pwc_stream_t stream) {
    if (!x || !packed_u || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_u) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)H * W * x_cs * 4 >= (long)W4_OOB || (long)H * W * y_cs * 4 >= (long)W4_OOB) return PWC_ERANGE;
    if ((long)36 * Cin_phys * ((Cout + 15) & ~15) * 4 >= (long)W4_OOB) return PWC_ERANGE;
    Wino4Args a;
    a.x = x; a.up = packed_u; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.apply_act = apply_act; a.slope = slope;
    a.dil = dilation;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    a.tiles_x = (ws + 31) / 32; a.tiles_y = (hs + 15) / 16; a.ncb = Cout / 16;
    const long nblk = (long)N * dilation * dilation * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    return wino4_launch<0>(a, (hipStream_t)stream);
}
