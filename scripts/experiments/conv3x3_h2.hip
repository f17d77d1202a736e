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
// serialises with it (no packed fp32 VALU: the split is plain v_cvt / v_sub / v_mul, 11 % of the matrix time, once per
// workgroup and 16-channel stage).
//
// Work decomposition (512 threads = 8 waves, one workgroup per CU):
//   workgroup = (2 WP rows) x 32 columns of output pixels x (64 WC) output channels, WC x WP = 8: WC = 2 (128 couts, 8 x 32
//   pixels) or WC = 1 (64 couts, 16 x 32 pixels); wave = 64 couts x 64 pixels (two rows of 32) = 2 x 2 tiles of
//   v_mfma_f32_32x32x16_f16: 8 accumulator tiles (hh, cross) = 128 registers.
//   Per 16-channel stage: the raw fp32 patch ((rows + 2) x 34 pixels x 64 bytes) arrives by buffer_load ... lds into a staging
//   image; all threads split it (4 channels per item) into the operand image [patch row][chunk: vh 0-7, vh 8-15, vm' 0-7,
//   vm' 8-15][pixel][16 bytes] -- 32 consecutive pixels of a chunk are 512 contiguous bytes = conflict-free B fragments at any
//   tap shift; the pre-split weights [cout tile][tap][chunk: uh 0-7, uh 8-15, um' 0-7, um' 8-15][cout 32][16 bytes] are a linear
//   copy of the packed global image.  K of an MFMA = [8 channels | 8 channels] over the two lane halves:
//     hh:      A = [uh 0-7 | uh 8-15]   B = [vh 0-7 | vh 8-15]
//     cross 0: A = [uh 0-7 | um' 0-7]   B = [vm' 0-7 | vh 0-7]         cross 1: the same for channels 8-15.
#pragma once
#include "../../pwcnet_amd/csrc/pwc_common.h"
#include <type_traits>

typedef _Float16 pwc_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pwc_f16x4 __attribute__((ext_vector_type(4)));
typedef float pwc_f32x16 __attribute__((ext_vector_type(16)));

struct H2Args {
    const float* x;
    const void* wp;      // packed split weights [c16][cout tile of 32][tap 9][chunk 4][cout 32][8 fp16]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // pixel tiles per (sub-)image, cout blocks of 64 WC
    int dil;
    int ntiles;
};

constexpr unsigned H2_OOB = 0x7FFF0000u;
constexpr int H2_PW = 34;                    // patch width in pixels
constexpr int H2_ROWB = 4 * H2_PW * 16;      // bytes of a patch row in the operand image: 4 chunks x 34 pixels x 16 B = 2176
constexpr int H2_TAPB = 4 * 32 * 16;         // bytes of the weights of one (cout tile, tap): 2048

template <int WC> struct H2Cfg {
    static constexpr int WP = 8 / WC;            // pixel groups (2 rows each)
    static constexpr int TR = 2 * WP;            // tile rows: 8 or 16
    static constexpr int PH = TR + 2;            // patch rows
    static constexpr int NREC = PH * H2_PW;      // patch pixels: 340 / 612
    static constexpr int NBP = (NREC + 15) / 16; // 1 KB pieces of the staging image (16 records of 64 B): 22 / 39
    static constexpr int PPW = (NBP + 7) / 8;    // patch pieces per wave: 3 / 5
    static constexpr int S_BYTES = (PPW * 8) * 1024;                 // staging (incl. the surplus pieces)
    static constexpr int B_BYTES = PH * H2_ROWB;                     // operand image: 21 760 / 39 168
    static constexpr int A_BYTES = 2 * WC * 9 * H2_TAPB;             // weights of a stage: 73 728 / 36 864
    static constexpr int UPW = A_BYTES / 1024 / 8;                   // weight pieces per wave: 9 / 4.5 -> see below
    static constexpr int S0 = 0, B0 = S_BYTES, A0 = B0 + ((B_BYTES + 1023) / 1024) * 1024;
    static constexpr int LDS = A0 + A_BYTES;
};

// ABL (harness only): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA, 8 = no split
template <int WC, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_h2_kernel(const H2Args a) {
    typedef H2Cfg<WC> C;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pg = wave % C::WP, cgw = wave / C::WP;     // pixel group (rows 2pg, 2pg+1), cout group (64 couts)
    const int ln = lane & 31, kh = lane >> 5;

    const int d = a.dil;
    const int nc16 = a.Cin_phys >> 4;
    // ---- block decode: cout block fastest, XCD-aware
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
    const int y0 = by * C::TR, x0 = bx * 32;        // output origin of the tile, in sub-lattice coordinates
    const int n0 = cb * 64 * WC;
    const int nct_all = a.Cout >> 5;                // cout tiles of 32 in the packed image
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wp, 0, nc16 * nct_all * 9 * H2_TAPB, 0x00020000);

    // ---- patch fetch: piece b = wave + 8 i holds records 16 b .. 16 b + 15 (record = patch pixel, 64 bytes = 16 channels)
    unsigned p_voff[C::PPW];
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) {
        const int rec = (wave + 8 * i) * 16 + (lane >> 2);
        const int py = rec / H2_PW, px = rec - py * H2_PW;
        const int yy = ry + d * (y0 - 1 + py), xx = rx + d * (x0 - 1 + px);
        const bool ok = rec < C::NREC && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        p_voff[i] = ok ? (unsigned)(((yy * a.W + xx) * a.x_cs + (lane & 3) * 4) * 4) : H2_OOB;
    }
    auto issue_patch = [&](int c16) {
#pragma unroll
        for (int i = 0; i < C::PPW; ++i)
            if (!(ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sm + C::S0 + (wave + 8 * i) * 1024), 16, (int)p_voff[i], c16 * 64, 0, 0);
    };
    // ---- weights of a stage: 2 WC cout tiles x 18 KB, contiguous in the packed image; pieces wave, wave + 8, ...
    constexpr int NWP = C::A_BYTES / 1024;          // 72 / 36 pieces
    constexpr int WPW = (NWP + 7) / 8;              // 9 / 5 (the last round of WC = 1 is half empty)
    const unsigned w_lane = (unsigned)lane * 16u;
    auto issue_w = [&](int c16) {
        const int sbase = (c16 * nct_all + (n0 >> 5)) * 9 * H2_TAPB;
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int pc = wave + 8 * j;               // uniform
            if (pc < NWP && !(ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(sm + C::A0 + pc * 1024), 16, (int)w_lane, sbase + pc * 1024, 0, 0);
        }
    };
#define H2_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // ---- split of the staging image into the operand image: item = (patch pixel, 4-channel group)
    auto convert = [&]() {
        constexpr int NIT = C::NREC * 4;
#pragma unroll
        for (int j = 0; j < (NIT + 511) / 512; ++j) {
            const int it = t + 512 * j;
            if (it < NIT) {
                const int rec = it >> 2, g = it & 3;
                const f32x4 v = *reinterpret_cast<const f32x4*>(sm + C::S0 + it * 16);
                pwc_f16x4 h, m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (_Float16)v[e];
                    m[e] = (ABL & 8) ? (_Float16)0.f : (_Float16)((v[e] - (float)h[e]) * 2048.f);
                }
                const int py = rec / H2_PW, px = rec - py * H2_PW;
                char* dst = sm + C::B0 + py * H2_ROWB + (g >> 1) * (H2_PW * 16) + px * 16 + (g & 1) * 8;
                *reinterpret_cast<pwc_f16x4*>(dst) = h;
                *reinterpret_cast<pwc_f16x4*>(dst + 2 * (H2_PW * 16)) = m;
            }
        }
    };

    // ---- fragment addresses.  B: patch row (2 pg + pt + dy), pixel ln + dx; A: cout tile (2 cgw + ct), tap, cout ln
    const char* const bbase = sm + C::B0 + (2 * pg) * H2_ROWB + ln * 16;
    const int b_hh = kh * (H2_PW * 16), b_x0 = (kh ? 0 : 2) * (H2_PW * 16), b_x1 = (kh ? 1 : 3) * (H2_PW * 16);
    const char* const abase = sm + C::A0 + (2 * cgw) * 9 * H2_TAPB + ln * 16;
    const int a_hh = kh * 512, a_x0 = (kh ? 2 : 0) * 512, a_x1 = (kh ? 3 : 1) * 512;

    pwc_f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = accx[ct][pt][r] = 0.f;

    issue_patch(0);
    issue_w(0);
    for (int c16 = 0; c16 < nc16; ++c16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of patch(c) and weights(c) landed
        H2_BAR();                                              // ... everybody's; the MFMAs of c-1 are done with the operand image
        convert();
        H2_BAR();                                              // operand image complete; staging free
        if (c16 + 1 < nc16) issue_patch(c16 + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            pwc_f16x8 Bh[2], Bx0[2], Bx1[2], Ah[2], Ax0[2], Ax1[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const char* bp = bbase + (pt + dy) * H2_ROWB + dx * 16;
                Bh[pt] = *reinterpret_cast<const pwc_f16x8*>(bp + b_hh);
                Bx0[pt] = *reinterpret_cast<const pwc_f16x8*>(bp + b_x0);
                Bx1[pt] = *reinterpret_cast<const pwc_f16x8*>(bp + b_x1);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const char* ap = abase + (ct * 9 + tap) * H2_TAPB;
                Ah[ct] = *reinterpret_cast<const pwc_f16x8*>(ap + a_hh);
                Ax0[ct] = *reinterpret_cast<const pwc_f16x8*>(ap + a_x0);
                Ax1[ct] = *reinterpret_cast<const pwc_f16x8*>(ap + a_x1);
            }
            if (ABL & 4) { asm volatile("" ::"v"(Bh[0]), "v"(Bx0[1]), "v"(Bx1[0]), "v"(Ah[1]), "v"(Ax0[0]), "v"(Ax1[1])); continue; }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ct], Bh[pt], acc[ct][pt], 0, 0, 0);
                    accx[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ax0[ct], Bx0[pt], accx[ct][pt], 0, 0, 0);
                    accx[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ax1[ct], Bx1[pt], accx[ct][pt], 0, 0, 0);
                }
        }
        H2_BAR();                                              // every wave is done with the weights of c
        if (c16 + 1 < nc16) issue_w(c16 + 1);
    }
    // ---- epilogue: y = hh + 2^-11 cross + bias, leaky-relu; lane = (pixel column ln, couts (r & 3) + 8 (r >> 2) + 4 kh of a tile)
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (size_t)n * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int py = ry + d * (y0 + 2 * pg + pt), px = rx + d * (x0 + ln);
        const bool inside = py < a.H && px < a.W;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = n0 + (2 * cgw + ct) * 32 + 8 * q + 4 * kh;
                if (co >= a.Cout) continue;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[ct][pt][4 * q + e] + accx[ct][pt][4 * q + e] * (1.f / 2048.f) + b4[e];
                if (a.apply_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], o[e] * a.slope);
                }
                const unsigned vo = inside ? (unsigned)(((py * a.W + px) * a.y_cs + co) * 4) : H2_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, (int)vo, 0, 0);
            }
    }
#undef H2_BAR
}

// ---------------------------------------------------------------- weight split + packing
// packed[c16][cout tile of 32][tap 9][chunk 4: uh ch 0-7, uh 8-15, um' 0-7, um' 8-15][cout 32][8 fp16];
// uh = fp16(w), um' = fp16((w - uh) * 2^11), both round-to-nearest; cin_map as in pwc_conv3x3_pack_f32.
__global__ void conv3x3_h2_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                       int Cin_phys, int Cout, int nct, unsigned short* __restrict__ packed) {
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
        const int cphys = c16 * 16 + ch;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        const int co = ct * 32 + i;
        float u = 0.f;
        if (clog >= 0 && clog < Cin && co < Cout) u = w[((size_t)tap * Cin + clog) * Cout + co];
        const _Float16 h = (_Float16)u;
        const _Float16 m = (_Float16)((u - (float)h) * 2048.f);
        unsigned short* base = packed + ((((size_t)c16 * nct + ct) * 9 + tap) * 4) * 256;      // 4 chunks x 32 couts x 8
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
                       Cin, Cin_phys, Cout, nct, reinterpret_cast<unsigned short*>(packed));
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_h2_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation < 1 || Cin_phys < 16 || (Cin_phys % 16) || Cout < 64 || (Cout % 64)) return 0;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    return hs >= 8 && ws >= 24 ? 1 : 0;
}

template <int WC, int ABL>
static int h2_launch(const H2Args& a, hipStream_t stream) {
    typedef H2Cfg<WC> C;
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_h2_kernel<WC, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    }
    hipLaunchKernelGGL((conv3x3_h2_kernel<WC, ABL>), dim3((unsigned)a.ntiles), dim3(512), C::LDS, stream, a);
    return pwc_launch_status();
}

template <int ABL = 0>
static int h2_run(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs, int N, int H, int W,
                  int Cin_phys, int Cout, int dilation, int apply_act, float slope, pwc_stream_t stream) {
    if (!x || !packed_w || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 64) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_w) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)H * W * x_cs * 4 >= (long)H2_OOB || (long)H * W * y_cs * 4 >= (long)H2_OOB) return PWC_ERANGE;
    H2Args a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.apply_act = apply_act; a.slope = slope;
    a.dil = dilation;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    const bool wide = (Cout % 128) == 0;
    const int TR = wide ? 8 : 16;
    a.tiles_x = (ws + 31) / 32; a.tiles_y = (hs + TR - 1) / TR; a.ncb = wide ? Cout / 128 : Cout / 64;
    const long nblk = (long)N * dilation * dilation * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    return wide ? h2_launch<2, ABL>(a, (hipStream_t)stream) : h2_launch<1, ABL>(a, (hipStream_t)stream);
}

extern "C" int pwc_conv3x3_h2_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y,
                                  int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                  int apply_act, float slope, pwc_stream_t stream) {
    return h2_run<0>(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, dilation, apply_act, slope, stream);
}
