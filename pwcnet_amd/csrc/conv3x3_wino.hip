// conv3x3_wino.hip -- 3x3 stride-1 'SAME' convolution (any dilation) by Winograd F(2x2, 3x3)
// on the fp32 MFMA units of gfx950.
//
// Replaces the same tf.layers.Conv2D(...,(3,3),(1,1),'same',dilation_rate=d) + tf.nn.leaky_relu
// calls as conv3x3_mfma.hip (reference modules.py:64-67, 267-268, 306-323) wherever stride = 1
// and Cout % 16 == 0: 16 multiplies per 2x2 outputs instead of 36, i.e. 2.25x fewer MFMA
// instructions for the same result (fp32 error ~1e-7 relative).  A dilation-d convolution is d*d
// independent ordinary convolutions on the pixel sub-lattices (y mod d, x mod d).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      per 4x4 input tile d, 2x2 output Y
//
// GEMM view: 16 independent products  M_xi[cout][tile] = sum_c U_xi[cout][c] * V_xi[c][tile],
// xi = (a,b) the position in the 4x4 transformed tile.
//
// Work decomposition (256 threads = 4 waves):
//   workgroup = 8 x 8 Winograd tiles (16 x 16 output pixels) x 32 output channels;
//   wave w    = tile rows 2w, 2w+1 (16 tiles = one MFMA column block), ALL 16 positions xi
//               and both 16-cout MFMA tiles: 16 x 2 accumulator tiles = 128 registers;
//   lane      = (tile j = lane & 15, k-slot q = lane >> 4): it reads the 4x4 input pixels of
//               its tile for channels 4q..4q+3 (16 ds_read_b128 from the raw patch), does the
//               input transform B^T d B IN REGISTERS -- the result is exactly the MFMA
//               B-operand fragment V_xi[k = 4q+s][tile j] -- and, at the end, holds all 16
//               M_xi of its (4 couts x 1 tile) outputs, so the output transform A^T M A is
//               register-local too.  No transformed data ever goes through LDS.
//   LDS per 16-channel stage: the raw 18 x 18 pixel patch (64-byte rows, see wpswz) and the
//   transformed weights U[xi][32 cout][16 ch] (pre-swizzled by the packer), both filled by
//   buffer_load_dwordx4 ... lds; out-of-image pixels are out-of-range buffer offsets and
//   read as zeros (SAME padding).  One stage buffer, fetched in three parts that are each
//   refetched as soon as they have been read (PIPE); 2 workgroups per CU.
//   Other tile shapes of the same kernel (WinoGeom): two short sub-lattice images per
//   workgroup, 4 x 64-pixel blocks; 16 instead of 32 output channels (NT = 1).
#include "pwc_common.h"
#include <cstdlib>

#ifndef WINO_BN32_MIN_WG
#define WINO_BN32_MIN_WG 384
#endif

struct WinoArgs {
    const float* x;
    const float* up;     // packed transformed weights [xi 16][c16][Cout_pad][16], chunk-swizzled
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // 16x16-pixel blocks per (sub-)image, cout blocks of 32
    int y_vec4;
    int dil;                     // dilation d: the conv splits into d*d ordinary convs on the pixel sub-lattices
    int ntiles;                  // (pixel block, cout block[, channel split]) tiles of the launch (= gridDim.x unless PERSIST)
    int csplit;                  // > 1: the input channel stages are split over `csplit` workgroups per tile; each writes its
    long slab;                   //      raw partial outputs (no bias, no activation) to y + z * slab (floats), see wino_split
};

__device__ __forceinline__ int wswz(int row) { return (4 - ((row >> 2) & 3)) & 3; }   // weight rows
// Patch image in LDS: pixel (py, px) of the 18 x 18 patch sits in 64-byte row
//   py * 20 + (px & 1) * 10 + (px >> 1)
// (even pixel columns first, then the odd ones; rows 9 and 19 of every 20 are unused), its
// 16-byte chunk c at slot c ^ wpswz(row).  The 16 tiles a wave reads at one (i, j) of the 4x4
// window then lie in two runs of 8 CONSECUTIVE rows, and with this chunk swizzle every
// ds_read_b128 lane group ({0-3,12-15,20-27}, ... MI355X_MICROARCH.md) covers the 16 slots of the
// 256-byte bank row exactly once.  (The first layout, rows py*18+px, had 2-way conflicts on all
// 16 patch reads of a stage: SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE.)
__device__ __forceinline__ int wpswz(int row) { return (row >> 1) & 2; }

constexpr unsigned WN_OOB = 0x7FFF0000u;  // byte offset beyond every buffer: loads return 0, stores are dropped
// NT = 16-cout MFMA tiles per workgroup (2: 32 output channels, 1: 16).
// GEO = shape of the workgroup's 64 Winograd tiles:
//   0  8 x 8 tiles = 16 x 16 pixels (patch 18 x 18);
//   1  SPLIT: 4 + 4 tile rows of TWO sub-lattice images (dilated convs whose sub-lattices are at
//      most 8 pixels high, e.g. d = 16 on a 112-row level: 7 x 16-pixel sub-lattices would fill
//      44 % of a 16 x 16 block); patch 2 x (8 + 2) rows x 18;
//   2  WIDE: 2 x 32 tiles = 4 x 64 pixels (patch 6 x 66), for images whose height is a multiple
//      of 4 but not of 16 (56- and 28-row levels and sub-lattices: 12.5 % of a 16-row grid is waste).
template <int NT, int GEO = 0> struct WinoGeom {
    static constexpr int BN = 16 * NT;                // output channels per workgroup
    static constexpr int UROWS = 16 * BN;             // weight rows (xi, cout) per stage
    static constexpr int NBU = UROWS / 16;
    static constexpr int BH = GEO == 2 ? 4 : 16, BW = GEO == 2 ? 64 : 16;   // output pixels per workgroup
    static constexpr int PH = GEO == 2 ? 6 : GEO == 1 ? 20 : 18;           // patch height in pixels
    static constexpr int PW = BW + 2;                                       // patch width in pixels
    static constexpr int PS = GEO == 2 ? 68 : 20;     // LDS rows per patch row: even pixel columns, then odd ones
    static_assert(PS / 2 >= PW / 2 && PS % 4 == 0, "patch row stride");
    static constexpr int PRP = GEO == 0 ? 384 : 448;  // PH * PS LDS rows, padded to a multiple of 64 (16-row DMA blocks x 4 waves)
    static_assert(PH * PS <= PRP, "patch does not fit");
    static constexpr int NBP = PRP / 16;
    static constexpr int STAGE = (PRP + UROWS) * 16;  // floats per LDS stage (NT = 2: 57 344 B, GEO 1/2: 61 440 B)
};

template <bool B> struct WinoBool { static constexpr bool value = B; };

// -1.0f in an SGPR the optimiser cannot see through: p - q is written fma(q, -1, p) so that it
// can become one v_pk_fma_f32 per two floats (a vector fsub is scalarised by the backend: there
// is v_pk_add_f32 but no packed subtract).  The VALU instructions of the transforms share the
// issue port with the MFMAs and their time ADDS to the MFMA time (measured: removing the 128
// scalar transform instructions of a stage saved 10 % of the kernel).
// (Inline-asm v_pk_add_f32 with neg modifiers was tried: fully packed, same speed, but every
// VALU write an MFMA reads next needs 2 wait states that the hazard recogniser only inserts
// for instructions it can see -- results were wrong until the s_nop moved into the asm.)
__device__ __forceinline__ float wino_minus_one() {
    float m;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m));
    return m;
}

// ABL (scripts/exp_wino.hip only, 0 in the library): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA,
// 8 = s_setprio(1) around the MFMA clusters, 16 = k innermost in the MFMA order (both measured: no gain),
// 64 = no input transform
// PIPE = 1: the stage is fetched in three parts (patch, weights of positions 0-7, of 8-15), each
// re-fetched for the next stage as soon as its LDS region is free, one barrier per part.
// PERSIST = 1 (needs PIPE; scripts/exp_wino.hip only): the grid is 2 workgroups per CU and each walks
// tiles blockIdx.x, + gridDim.x, ...; the first stage of the next tile is requested before the
// current tile's output transform and stores, so its fetch latency and the per-tile setup hide behind
// them.  Correct, but the 7 offsets of the next tile live across the epilogue push the kernel over
// 256 VGPRs (84 B of scratch per lane) and it measures 289 us against 255 us: not used by the library.
// barriers of the stage pipeline: LDS only (pwc_lds_barrier) -- WINO_DRAIN_BARRIERS=1 restores the round-1/2 behaviour
// (__syncthreads(), which drains every DMA piece in flight) for A/B measurements
#ifndef WINO_DRAIN_BARRIERS
#define WINO_SYNC() pwc_lds_barrier()
#else
#define WINO_SYNC() __syncthreads()
#endif
template <int ABL = 0, int NT = 2, int PIPE = 0, int GEO = 0, int PERSIST = 0>
__global__ __launch_bounds__(256, NT >= 4 ? 1 : 2) void conv3x3_wino_kernel(const WinoArgs a) {
    static_assert(!PERSIST || PIPE, "the persistent loop is built on the pipelined stage");
    typedef WinoGeom<NT, GEO> Geo;
    constexpr int SPLIT = GEO == 1, WIDE = GEO == 2;
    constexpr int WN_BN = Geo::BN, WN_NBU = Geo::NBU, WN_PRP = Geo::PRP, WN_NBP = Geo::NBP, WN_PH = Geo::PH;
    constexpr int WN_PW = Geo::PW, WN_PS = Geo::PS;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const float m1 = wino_minus_one();
    const f32x4 M1 = {m1, m1, m1, m1};
#define WSUB(p, q) __builtin_elementwise_fma((q), M1, (p))    /* p - q */

    const int d = a.dil;
    const int Cout_pad = (a.Cout + 15) & ~15;
    const int nc16 = a.Cin_phys >> 4;
    constexpr int PPW = WN_NBP / 4;                // patch blocks per wave (6)
    constexpr int UPW = WN_NBU / 4;                // weight blocks per wave (8 for NT = 2)
    constexpr int UH = UPW / 2;                    // ... per half of the positions
    static_assert(WN_NBU % 8 == 0 && WN_NBP % 4 == 0, "blocks must split evenly over the waves");
    static_assert(64 % WN_BN == 0, "weight block stride must be a whole number of positions");
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.up, 0, 16 * a.Cin_phys * Cout_pad * 4, 0x00020000);

    // ---- per-tile state (set by setup): block decode -- cout block fastest, XCD-aware (the cout
    // blocks of one pixel block share their input patch in one XCD's L2) -- and the LDS-DMA
    // bookkeeping.  Both operands are fetched with buffer_load_dwordx4 ... lds: a per-lane BYTE
    // offset that is fixed over the channel loop (VGPR), the stage advance in the scalar offset
    // (SGPR) -- no vector instruction per fetch -- and the buffer range check gives the zeros of
    // the SAME padding: out-of-image lanes carry the offset WN_OOB.
    int n, n0, y0, x0, ry_g[2], rx_g[2];
    int zsplit = 0, c_begin = 0, c_end = nc16;     // this workgroup's share of the channel stages
    __amdgpu_buffer_rsrc_t xrsrc;
    unsigned p_voff[PPW];
    unsigned u_voff;
    auto setup = [&](int tile) {
        int lb = pwc_xcd_remap(tile, a.ntiles);
        zsplit = lb % a.csplit;                    // (csplit = 1: 0)
        lb /= a.csplit;
        {
            const int base = nc16 / a.csplit, rem = nc16 - base * a.csplit;
            c_begin = zsplit * base + min(zsplit, rem);
            c_end = c_begin + base + (zsplit < rem ? 1 : 0);
        }
        const int cb = lb % a.ncb;
        int rest = lb / a.ncb;
        const int bx = rest % a.tiles_x;
        rest /= a.tiles_x;
        const int by = rest % a.tiles_y;
        rest /= a.tiles_y;
        const int nsub = SPLIT ? (d * d) >> 1 : d * d;   // sub-lattices (SPLIT: pairs of them) of a dilated conv
        const int sub = rest % nsub;
        n = rest / nsub;
        // sub-lattice (y mod d, x mod d) of tile-row group g (SPLIT: g = 0 for tile rows 0-3, 1 for 4-7)
        const int sub_g[2] = {SPLIT ? 2 * sub : sub, SPLIT ? 2 * sub + 1 : sub};
        ry_g[0] = sub_g[0] / d; ry_g[1] = sub_g[1] / d;
        rx_g[0] = sub_g[0] - ry_g[0] * d; rx_g[1] = sub_g[1] - ry_g[1] * d;
        y0 = by * Geo::BH; x0 = bx * Geo::BW;          // output origin of the block, in sub-lattice coordinates
        n0 = cb * WN_BN;
        xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0,
                                                  a.H * a.W * a.x_cs * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int b = wave + 4 * i;
            const int pr = b * 16 + (lane >> 2);                       // LDS row this lane fills
            const int py = pr / WN_PS, rem = pr - py * WN_PS;
            const int half = rem >= WN_PS / 2 ? 1 : 0, col = rem - half * (WN_PS / 2);
            const int px = 2 * col + half;
            const int g = SPLIT ? (py >= 10 ? 1 : 0) : 0, pyl = py - 10 * g;   // SPLIT: patch rows 0-9 / 10-19
            const int y = ry_g[g] + d * (y0 - 1 + pyl), x = rx_g[g] + d * (x0 - 1 + px);
            const int ch = (lane & 3) ^ wpswz(pr);                     // source chunk for this LDS slot
            const bool ok = py < WN_PH && col < WN_PW / 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            p_voff[i] = ok ? (unsigned)(((y * a.W + x) * a.x_cs + ch * 4) * 4) : WN_OOB;
        }
        // weight blocks: block wave + 4*i holds rows (xi, cout) = ((wave + 4*i) * 16 + lane/4); 4 blocks
        // = 64 rows = 64 / WN_BN positions, so one per-lane offset plus a uniform stride covers all i
        const int ur = wave * 16 + (lane >> 2);
        const int xi = ur / WN_BN, co = ur - xi * WN_BN;
        u_voff = (n0 + co < Cout_pad) ? (unsigned)((((xi * nc16) * Cout_pad + n0 + co) * 16 + (lane & 3) * 4) * 4) : WN_OOB;
    };
    int tile = blockIdx.x;
    setup(tile);
    const int u_step = (64 / WN_BN) * nc16 * Cout_pad * 64;       // bytes between a wave's consecutive weight blocks
    auto issue_patch = [&](int c16) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (!(ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(smem + (wave + 4 * i) * 256), 16, (int)p_voff[i],
                                                         c16 * 64, 0, 0);
    };
    auto issue_u = [&](int c16, int half) {        // half 0: positions 0-7, 1: positions 8-15
        const int us = c16 * Cout_pad * 64;
#pragma unroll
        for (int i = half * UH; i < (half + 1) * UH; ++i)
            if (!(ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lptr_t)(smem + (WN_NBP + wave + 4 * i) * 256), 16,
                                                         (int)u_voff, us + i * u_step, 0, 0);
    };
    // s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

    // ---- this lane's tile and its 16 patch read offsets (floats, swizzled for k-slot fq)
    // tile (tr, tc) of this lane: 8 x 8 tiles (wave = 2 tile rows) or, WIDE, 2 x 32 tiles (wave = half a row)
    const int tr = WIDE ? wave >> 1 : 2 * wave + (fr >> 3), tc = WIDE ? (wave & 1) * 16 + fr : fr & 7;
    int poff[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int prow = SPLIT ? (tr >> 2) * 10 + 2 * (tr & 3) + i : 2 * tr + i;
            const int row = prow * WN_PS + (j & 1) * (WN_PS / 2) + tc + (j >> 1);
            poff[i][j] = row * 16 + ((fq ^ wpswz(row)) << 2);
        }
    const int u_off = WN_PRP * 16 + fr * 16 + ((fq ^ wswz(fr)) << 2);   // A-fragment row fr of a 16-row tile

    f32x4 acc[16][NT];
    // one 16-channel stage; FIRST: the accumulators start from the MFMA's zero C operand
    auto stage = [&](auto first, int c16) {
        constexpr bool FIRST = decltype(first)::value;
        const bool has_next = c16 + 1 < c_end;
        if (PIPE) {
            WAIT_VM(UH);                             // patch(c) landed (weights A(c) may be in flight)
            WINO_SYNC();                         // ... for every wave; positions 8-15 of c-1 fully read
            issue_u(c16, 1);
        } else {
            WINO_SYNC();                         // previous stage fully read
            issue_patch(c16); issue_u(c16, 0); issue_u(c16, 1);
            WAIT_VM(0);
            WINO_SYNC();
        }

        // ---- input transform  V = B^T d B  (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]), in place
        f32x4 v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = *reinterpret_cast<const f32x4*>(smem + poff[i][j]);
        if (!(ABL & 64)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {            // rows
                const f32x4 d0 = v[0][j], d1 = v[1][j], d2 = v[2][j], d3 = v[3][j];
                v[0][j] = WSUB(d0, d2); v[1][j] = d1 + d2; v[2][j] = WSUB(d2, d1); v[3][j] = WSUB(d1, d3);
            }
            // opaque uses: the MFMAs take single floats out of these vectors and the backend would
            // rewrite extract(vector op) as scalar ops, losing the packed instructions
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm("" : "+v"(v[i][j]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {            // columns
                const f32x4 e0 = v[i][0], e1 = v[i][1], e2 = v[i][2], e3 = v[i][3];
                v[i][0] = WSUB(e0, e2); v[i][1] = e1 + e2; v[i][2] = WSUB(e2, e1); v[i][3] = WSUB(e1, e3);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm("" : "+v"(v[i][j]));
        }
        // ---- 16 positions x NT cout tiles x 4 k-steps of MFMA, in two halves of 8 positions
        auto mfma_half = [&](int h) {
#pragma unroll
            for (int xi = 8 * h; xi < 8 * h + 8; ++xi) {
                f32x4 wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[nt] = *reinterpret_cast<const f32x4*>(smem + u_off + (xi * WN_BN + nt * 16) * 16);
#pragma unroll
                for (int kk = 0; kk < 4 * NT; ++kk) {
                    // ABL & 16: the 4 k-steps of one accumulator back to back (k innermost); default: the
                    // NT accumulators of a position alternate
                    const int k = (ABL & 16) ? kk % 4 : kk / NT, nt = (ABL & 16) ? kk / 4 : kk % NT;
                    if (ABL & 4) {
                        if (FIRST && k == 0) acc[xi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        asm volatile("" ::"v"(wf[nt][k]), "v"(v[xi >> 2][xi & 3][k]));
                        continue;
                    }
                    const f32x4 c = (FIRST && k == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[xi][nt];
                    acc[xi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][k], v[xi >> 2][xi & 3][k], c, 0, 0, 0);
                }
            }
        };
        if (PIPE) {
            WAIT_VM(UH);                             // weights A(c) landed (B(c) may be in flight)
            WINO_SYNC();                         // ... for every wave; patch(c) fully read
            if (has_next) issue_patch(c16 + 1);
        }
        if (ABL & 8) __builtin_amdgcn_s_setprio(1);
        mfma_half(0);
        if (ABL & 8) __builtin_amdgcn_s_setprio(0);
        if (PIPE) {
            if (has_next) WAIT_VM(PPW); else WAIT_VM(0);   // weights B(c) landed (patch(c+1) may be in flight)
            WINO_SYNC();                         // ... for every wave; positions 0-7 of c fully read
            if (has_next) issue_u(c16 + 1, 0);
        }
        if (ABL & 8) __builtin_amdgcn_s_setprio(1);
        mfma_half(1);
        if (ABL & 8) __builtin_amdgcn_s_setprio(0);
    };
    if (PIPE) { issue_patch(c_begin); issue_u(c_begin, 0); }
    for (;;) {
    stage(WinoBool<true>{}, c_begin);
    for (int c16 = c_begin + 1; c16 < c_end; ++c16) stage(WinoBool<false>{}, c16);

    // the tile's output coordinates, before setup() moves on to the next tile
    const int on = n, on0 = n0, oy0 = y0, ox0 = x0, oz = zsplit;
    const int ory[2] = {ry_g[0], ry_g[1]}, orx[2] = {rx_g[0], rx_g[1]};
    const int next = tile + (int)gridDim.x;
    const bool more = PERSIST && next < a.ntiles;
    if (more) {
        // all waves are past the last stage's patch and positions-0-7 reads (its barriers): those two
        // LDS regions take the next tile's first stage while this tile's outputs are transformed and stored
        setup(next);
        issue_patch(c_begin);
        issue_u(c_begin, 0);
    }

    // ---- output transform  Y = A^T M A  (A^T = [1 1 1 0; 0 1 -1 -1]), bias, leaky-relu; stores go
    // through a buffer resource of image n so that pixels beyond the image edge are dropped by the
    // range check (no divergent branches)
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (size_t)oz * a.slab + (size_t)on * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
    const int og = SPLIT ? tr >> 2 : 0, otr = SPLIT ? tr & 3 : tr;
    const int py0 = ory[og] + d * (oy0 + 2 * otr), px0 = orx[og] + d * (ox0 + 2 * tc);   // real coordinates of output (0,0)
    unsigned y_voff[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int py = py0 + i * d, px = px0 + j * d;
            y_voff[i][j] = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.y_cs + on0 + fq * 4) * 4) : WN_OOB;
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = on0 + nt * 16 + fq * 4;
        if (co >= a.Cout) continue;
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[0 * 4 + j][nt] + acc[1 * 4 + j][nt] + acc[2 * 4 + j][nt];
            s[1][j] = WSUB(WSUB(acc[1 * 4 + j][nt], acc[2 * 4 + j][nt]), acc[3 * 4 + j][nt]);
        }
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};            // (channel split: the reduce kernel adds the bias)
        if (a.csplit == 1) b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 yv = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : WSUB(WSUB(s[i][1], s[i][2]), s[i][3]);
                yv += b4;
                if (a.apply_act) {               // tf.nn.leaky_relu = max(v, slope * v)
                    const f32x4 sv = yv * a.slope;
                    yv[0] = fmaxf(yv[0], sv[0]); yv[1] = fmaxf(yv[1], sv[1]);
                    yv[2] = fmaxf(yv[2], sv[2]); yv[3] = fmaxf(yv[3], sv[3]);
                }
                if (a.y_vec4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), yrsrc, (int)y_voff[i][j], nt * 64, 0);
                } else {
                    const u32x4 yu = __builtin_bit_cast(u32x4, yv);
                    __builtin_amdgcn_raw_buffer_store_b32(yu[0], yrsrc, (int)y_voff[i][j], nt * 64 + 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(yu[1], yrsrc, (int)y_voff[i][j], nt * 64 + 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(yu[2], yrsrc, (int)y_voff[i][j], nt * 64 + 8, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(yu[3], yrsrc, (int)y_voff[i][j], nt * 64 + 12, 0);
                }
            }
        }
    }
    if (!more) break;
    tile = next;
    }   // tiles
#undef WSUB
#undef WAIT_VM
}

// ---------------------------------------------------------------- weight transform + packing
// packed[xi][c16][cout_pad][16]: U_xi = (G g G^T)[a][b], xi = 4a + b, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
// chunk-swizzled like the direct kernel's image; cin_map as in pwc_conv3x3_pack_f32.
__global__ void conv3x3_wino_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                         int Cin_phys, int Cout, int Cout_pad, float* __restrict__ packed) {
    const size_t total = (size_t)16 * Cin_phys * Cout_pad;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e16 = (int)(idx & 15);
        size_t r = idx >> 4;
        const int co = (int)(r % Cout_pad);
        r /= Cout_pad;
        const int c16 = (int)(r % (Cin_phys >> 4));
        const int xi = (int)(r / (Cin_phys >> 4));
        const int jpos = e16 >> 2, e = e16 & 3;
        const int j = jpos ^ wswz(co);
        const int cphys = c16 * 16 + j * 4 + e;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        float u = 0.f;
        if (clog >= 0 && clog < Cin && co < Cout) {
            const int ua = xi >> 2, ub = xi & 3;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    u += G[ua][p] * G[ub][q] * w[((size_t)(p * 3 + q) * Cin + clog) * Cout + co];
        }
        packed[idx] = u;
    }
}

extern "C" size_t pwc_conv3x3_wino_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    return (size_t)16 * Cin_phys * ((Cout + 15) & ~15);
}

extern "C" int pwc_conv3x3_wino_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                         int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int Cout_pad = (Cout + 15) & ~15;
    const size_t total = (size_t)16 * Cin_phys * Cout_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_wino_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, Cout_pad, packed);
    return pwc_launch_status();
}

// output channels per workgroup: 32, or 16 when 32 would leave most of the 512 workgroup slots
// (256 CUs x 2) empty -- measured on the 28 x 64 level (14 336 pixels): 16 is faster below ~384 workgroups
static int wino_bn(long pix_blocks, int Cout) {
    if (Cout % 32) return 16;
#ifdef PWC_HARNESS
    if (const char* e = getenv("PWC_WINO_FORCE_BN")) {          // libpwc_hip_harness.so only (scripts/tune_wino_split.py)
        const int v = atoi(e);
        if (v == 16 || v == 32) return v;
    }
#endif
    return pix_blocks * (Cout / 32) >= WINO_BN32_MIN_WG ? 32 : 16;
}

// geometry of a launch (see WinoGeom): 1 = two sub-lattice images per workgroup when they are at most
// 8 pixels high; 2 = 4 x 64-pixel blocks when they cover the (sub-)image with at least 5 % less
// waste than 16 x 16 ones; 0 otherwise
static int wino_geo(int H, int W, int dilation) {
    const long hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    if ((dilation * dilation) % 2 == 0 && hs <= 8) return 1;
    const double sq = (double)(((hs + 15) / 16) * 16) * (((ws + 15) / 16) * 16);
    const double wd = (double)(((hs + 3) / 4) * 4) * (((ws + 63) / 64) * 64);
    return wd < 0.95 * sq ? 2 : 0;
}

static void wino_grid(int N, int H, int W, int dilation, int geo, int* tiles_x, int* tiles_y, long* pix_blocks) {
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    const int bh = geo == 2 ? 4 : 16, bw = geo == 2 ? 64 : 16;
    *tiles_x = (ws + bw - 1) / bw;
    *tiles_y = (hs + bh - 1) / bh;
    *pix_blocks = (long)N * (geo == 1 ? dilation * dilation / 2 : dilation * dilation) * *tiles_x * *tiles_y;
}

extern "C" long pwc_conv3x3_wino_workgroups(int N, int H, int W, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || dilation < 1 || Cout % 16) return 0;
    int tx, ty;
    long pix_blocks;
    wino_grid(N, H, W, dilation, wino_geo(H, W, dilation), &tx, &ty, &pix_blocks);
    return pix_blocks * (Cout / wino_bn(pix_blocks, Cout));
}

// sum of the channel-split partial outputs (fixed order) + bias + leaky-relu -> y
__global__ __launch_bounds__(256) void conv3x3_wino_split_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                        float* __restrict__ y, int y_cs, int y_vec4, long M,
                                                                        int Cout, int Cout_pad, int nsplit, int apply_act,
                                                                        float slope) {
    const int c4n = Cout >> 2;
    const long total = M * c4n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        f32x4 v = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
        for (int z = 0; z < nsplit; ++z) v += *reinterpret_cast<const f32x4*>(ws + ((size_t)z * M + pix) * Cout_pad + c4 * 4);
        if (apply_act) {
            v[0] = pwc_lrelu(v[0], slope); v[1] = pwc_lrelu(v[1], slope);
            v[2] = pwc_lrelu(v[2], slope); v[3] = pwc_lrelu(v[3], slope);
        }
        float* dst = y + (size_t)pix * y_cs + c4 * 4;
        if (y_vec4) *reinterpret_cast<f32x4*>(dst) = v;
        else { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    }
}

static int wino_run(const float* x, int x_cs, const float* packed_u, const float* bias, float* y, int y_cs, int N, int H,
                    int W, int Cin_phys, int Cout, int dilation, int apply_act, float slope, int csplit, float* workspace,
                    size_t workspace_floats, pwc_stream_t stream);

// Channel split for launches that leave most of the 512 workgroup slots (256 CUs x 2) empty -- the 14 x 32 and
// 28 x 64 pyramid levels: 32 ... 128 workgroups that each walk 6 ... 16 channel stages whose fetch latency nothing
// hides.  With csplit > 1 the stages are dealt to csplit workgroups per tile, each writes its raw partial outputs to a
// slab of the caller's workspace, and a reduce kernel adds the slabs in a fixed order, the bias and the activation.
// pwc_conv3x3_wino_split_plan returns the csplit pwc_conv3x3_wino_split_f32 should be given (1: no split).
extern "C" int pwc_conv3x3_wino_split_plan(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys < 16 || Cout <= 0 || dilation < 1 || (Cin_phys % 16) || (Cout % 16)) return 1;
#ifdef PWC_HARNESS
    if (const char* e = getenv("PWC_WINO_FORCE_SPLIT")) {       // libpwc_hip_harness.so only (scripts/tune_wino_split.py)
        const int v = atoi(e);
        if (v >= 1 && v <= Cin_phys / 16) return v;
    }
#endif
    const long wgs = pwc_conv3x3_wino_workgroups(N, H, W, Cout, dilation);
    const int nc16 = Cin_phys / 16;
    // measured (batch 8, levels 14x32 / 28x64): the split pays from a quarter-filled GPU down and from 6 stages up --
    // 33 -> 21 us for 128 workgroups x 16 stages; at 256 workgroups x 8 stages the reduce launch costs more than it saves
    if (wgs <= 0 || wgs > 128 || nc16 < 6) return 1;
    long s = 256 / wgs;                                        // fill half of the slots (scripts/tune_wino_split.py: at 128
    if (s > 4) s = 4;                                          // workgroups 2 parts beat 4 by 1-2 us)
    if (s < 2) s = 2;
    if (s > nc16 / 2) s = nc16 / 2;
    return s < 2 ? 1 : (int)s;
}

extern "C" size_t pwc_conv3x3_wino_split_workspace_floats(int N, int H, int W, int Cout, int csplit) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || csplit <= 1) return 0;
    return (size_t)csplit * N * H * W * ((Cout + 15) & ~15);
}

extern "C" int pwc_conv3x3_wino_split_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                          int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                          int apply_act, float slope, int csplit, float* workspace,
                                          size_t workspace_floats, pwc_stream_t stream) {
    if (csplit < 1 || (csplit > 1 && (!workspace || !pwc_aligned16(workspace)))) return PWC_EINVAL;
    if (csplit > 1 && (Cin_phys / 16 < csplit || workspace_floats < pwc_conv3x3_wino_split_workspace_floats(N, H, W, Cout, csplit)))
        return PWC_EINVAL;
    return wino_run(x, x_cs, packed_u, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, csplit, workspace,
                    workspace_floats, stream);
}

extern "C" int pwc_conv3x3_wino_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                    int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                    int apply_act, float slope, pwc_stream_t stream) {
    return wino_run(x, x_cs, packed_u, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, 1, nullptr, 0, stream);
}

static int wino_run(const float* x, int x_cs, const float* packed_u, const float* bias, float* y, int y_cs, int N, int H,
                    int W, int Cin_phys, int Cout, int dilation, int apply_act, float slope, int csplit, float* workspace,
                    size_t workspace_floats, pwc_stream_t stream) {
    (void)workspace_floats;
    if (!x || !packed_u || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(packed_u) || !pwc_aligned16(bias)) return PWC_EALIGN;
    // per-image slabs are addressed with 32-bit byte offsets through buffer resources
    if ((long)H * W * x_cs * 4 >= (long)WN_OOB || (long)H * W * y_cs * 4 >= (long)WN_OOB) return PWC_ERANGE;
    if ((long)16 * Cin_phys * ((Cout + 15) & ~15) * 4 >= (long)WN_OOB) return PWC_ERANGE;
    const int Cout_pad = (Cout + 15) & ~15;
    const long M = (long)N * H * W;
    if (csplit > 1 && (long)H * W * Cout_pad * 4 >= (long)WN_OOB) return PWC_ERANGE;
    WinoArgs a;
    a.x = x; a.up = packed_u; a.bias = bias; a.x_cs = x_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout;
    a.slope = slope;
    a.dil = dilation;
    a.csplit = csplit;
    if (csplit > 1) {            // partial outputs: dense [z][pixel][Cout_pad] slabs, no bias / activation
        a.y = workspace; a.y_cs = Cout_pad; a.apply_act = 0; a.slab = M * Cout_pad;
    } else {
        a.y = y; a.y_cs = y_cs; a.apply_act = apply_act; a.slab = 0;
    }
    const int geo = wino_geo(H, W, dilation);
    long pix_blocks;
    wino_grid(N, H, W, dilation, geo, &a.tiles_x, &a.tiles_y, &pix_blocks);
    const int bn = wino_bn(pix_blocks, Cout);
    a.ncb = Cout / bn;
    a.y_vec4 = csplit > 1 ? 1 : (((y_cs & 3) == 0 && pwc_aligned16(y)) ? 1 : 0);
    const long nblk = pix_blocks * a.ncb * csplit;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    // measured (scripts/exp_wino.hip): one LDS stage fetched in three pipelined parts with 2 co-resident
    // workgroups per CU beats a double-buffered whole stage (1 workgroup per CU)
#define WINO_LAUNCH_P(NT, GEO, PERS)                                                                        \
    do {                                                                                                    \
        const size_t lds = (size_t)WinoGeom<NT, GEO>::STAGE * sizeof(float);                                \
        static PwcDevOnce attr_once;                                                                          \
        if (pwc_first_on_device(&attr_once)) {                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<0, NT, 1, GEO, PERS>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        }                                                                                                   \
        const long grid = (PERS && nblk > 512) ? 512 : nblk;   /* persistent: 2 workgroups per CU walk the tiles */ \
        hipLaunchKernelGGL((conv3x3_wino_kernel<0, NT, 1, GEO, PERS>), dim3((unsigned)grid), dim3(256), lds, \
                           (hipStream_t)stream, a);                                                         \
    } while (0)
#define WINO_LAUNCH(NT, GEO) WINO_LAUNCH_P(NT, GEO, 0)
    // One- and two-stage launches of the 16-cout variant (the 16 -> 16 and 32 -> 16 layers: Cin_phys <= 32) spend
    // most of a workgroup's life waiting for its only patch: there the persistent form -- the next tile's first
    // stage is requested before the current tile's output transform and stores -- pays (64 accumulator registers,
    // no spill; with 32 couts it spills and loses, see the kernel comment).
    // (A dedicated 16 -> 16 kernel -- weights resident in registers, double-buffered patch, LDS-only barriers -- was built
    // and measured in round 3: 76.7 us against this kernel's 75.4 us on 16 x 224 x 512; its memory side alone runs in 38 us,
    // its compute side alone in 49 us, together 77: removed again, see DESIGN.md.)
#ifdef PWC_HARNESS
    const char* pe = getenv("PWC_WINO_PERSIST");                // libpwc_hip_harness.so only (scripts/exp_wino_bn.py)
#else
    const char* pe = nullptr;
#endif
    const bool persist = (pe ? atoi(pe) != 0 : true) && bn == 16 && Cin_phys <= 32 && nblk > 1024 && csplit == 1;
    if (persist && geo == 0) { WINO_LAUNCH_P(1, 0, 1); return pwc_launch_status(); }
    if (persist && geo == 2) { WINO_LAUNCH_P(1, 2, 1); return pwc_launch_status(); }
    if (bn == 32) { if (geo == 1) WINO_LAUNCH(2, 1); else if (geo == 2) WINO_LAUNCH(2, 2); else WINO_LAUNCH(2, 0); }
    else          { if (geo == 1) WINO_LAUNCH(1, 1); else if (geo == 2) WINO_LAUNCH(1, 2); else WINO_LAUNCH(1, 0); }
#undef WINO_LAUNCH
#undef WINO_LAUNCH_P
    if (csplit > 1) {
        long blocks = (M * (Cout >> 2) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv3x3_wino_split_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)workspace, bias, y, y_cs, ((y_cs & 3) == 0 && pwc_aligned16(y)) ? 1 : 0, M, Cout,
                           Cout_pad, csplit, apply_act, slope);
    }
    return pwc_launch_status();
}

#include "conv3x3_wino4.hip"   // F(4x4, 3x3) variant for the big full-resolution layers (same translation unit)
