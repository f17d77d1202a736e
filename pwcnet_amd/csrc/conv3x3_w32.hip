// conv3x3_w32.hip -- 3x3 stride-1 'SAME' convolution (+ bias, + leaky-relu) of 32 or 64 input channels to 32 OUTPUT channels on the
// F16 matrix pipe, the whole weight tensor resident in the LDS (round 6).  libpwc_hip.so, gfx950 only.
//
// Replaces (reference modules.py:58-71, :266-268, :321-323): tf.layers.Conv2D(32, (3, 3), (1, 1), 'same') + LeakyReLU(0.1) on NHWC
// fp32 -- fp_extractor/conv2d_4 and conv2d_5 (32 -> 32 at 112 x 256 of both frames), optflow_l/conv2d_4 (64 -> 32, the last hidden
// layer of every estimator) and context/conv2d_5 (64 -> 32).  At batch 8 these were six launches of conv3x3_h2_kernel /
// conv3x3_wino4_kernel: 45 + 45 + 46 + 40 + 19 us (profiles/r05_forward_trace_b8.txt).
//
// Why another kernel.  With 32 output channels conv3x3_h2_kernel's tile is 32 couts x 16 rows: six matrix instructions per tap and
// wave against a per-tap fixed cost worth 3.6 (weights and patches staged per 16-channel stage, three barriers per stage) -- it runs
// these layers at 0.55 - 0.63 PFLOP/s where the 128-cout layers reach 1.1.  conv3x3_t32.hip showed the other way for thin layers:
// all 32 output channels are the ROW operand of one 32 x 32 x 16 instruction, so the weights never move and only pixel fragments
// stream -- but it keeps the weights in REGISTERS, which ends at 16 input channels (144 registers at 32).  Here they live in the
// LDS for the life of the workgroup (9 taps x C_in / 16 K steps x 2 KB of split halves: 36 or 72 KB, copied once, linear), a K step
// reads its two weight fragments and the wave's two pixel fragments (four conflict-free ds_read_b128 for three matrix
// instructions) two steps ahead, and there is no per-stage staging at all: per tile the input patch is requested into REGISTERS
// (pixel-major: neighbouring lanes ask for neighbouring 32 bytes) while the previous tile computes, split ONCE into the operand image
// (h and m' halves per 8-channel group, 16 bytes per position) between two barriers, and the K loop runs on it.
//
// Work decomposition: 512 threads = 8 waves, one workgroup per CU, persistent over tiles of TR rows x 32 columns.
//   C_in = 32: TR = 8, a wave = one row, K = 9 taps x 2 steps.
//   C_in = 64: TR = 4, a wave = (row, K half): waves 4-7 take input channels 32-63 and hand their finished sums to waves 0-3 through
//              the LDS (one fp32 addition of two finished sums, fixed order: launches repeat bitwise).  The operand image of 8 rows
//              would not fit beside 72 KB of weights.
// Arithmetic and RANGE of conv3x3_h2.hip: x = h + 2^-11 m' per operand, hh and cross terms in separate fp32 accumulators, combined
// once; |x| >= 65504 gives NaN outputs (PWC_STATUS_NONFINITE at the end of the forward).
#include "pwc_common.h"

typedef float w32_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int w32_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* w32_lptr;
#define W32_OOB 0x80000000u

struct W32Args {
    const float* x;
    const float* wp;        // packed split weights: [tap 9][C_in / 16][h | m'][64 lanes][8 fp16]  (conv3x3_w32_pack_kernel)
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ntiles;
};

template <int CIN>
struct W32Geom {
    static constexpr int KS = CIN / 32;                            // K halves (wave groups): 1 or 2
    static constexpr int TR = 8 / KS, TW = 32;                     // output rows x columns of a tile
    static constexpr int PR = TR + 2, PC = TW + 2;                 // patch rows x columns
    static constexpr int NPOS = PR * PC;                           // patch positions: 340 / 204
    // bytes of an (8-channel group, half) plane: positions x 16, padded so that PLANE = 16 mod 64 -- the eight lanes of a store
    // group write eight 8-channel groups of one pixel (planes 2 PLANE apart: 32 bytes mod 128, a 2-way overlap the store's own
    // cycles cover), and the h and m' planes of a group start 4 banks apart
    static constexpr int PLANE = ((NPOS * 16 + 63) / 64) * 64 + 16;
    static constexpr int NC8 = CIN / 8;                            // 8-channel groups
    static constexpr int OPI = NC8 * 2 * PLANE;                    // operand image
    static constexpr int J16 = CIN / 16;
    static constexpr int NSW = 9 * J16;                            // K steps of the whole tensor
    static constexpr int WB = NSW * 2048;                          // weights: 36 / 72 KB
    static constexpr int NITEM = NPOS * NC8;                       // (position, 8-channel group) items of a patch: 32 bytes each
    static constexpr int IPT = (NITEM + 511) / 512;                // ... per thread: 3 / 4
    static constexpr int XCH = KS == 2 ? 4 * 64 * 64 : 0;          // finished sums of the upper K half: 4 waves x 64 lanes x 16 floats
    static constexpr int NS = 9 * (J16 / KS);                      // K steps of a wave: 18
    static constexpr int W0 = 0, O0 = WB, X0 = WB + OPI, LDS = WB + OPI + XCH;
    static_assert(CIN == 32 || CIN == 64, "C_in");
    static_assert(LDS <= 160 * 1024, "does not fit the LDS");
    static_assert(PLANE % 64 == 16, "plane skew");
};

template <int CIN>
__global__ __launch_bounds__(512, 1) void conv3x3_w32_kernel(const W32Args a) {
    using G = W32Geom<CIN>;
    constexpr int KS = G::KS, TR = G::TR, PC = G::PC, NC8 = G::NC8, PLANE = G::PLANE, NS = G::NS, IPT = G::IPT;
    extern __shared__ __attribute__((aligned(16))) char w32_smem[];
    char* const wl = w32_smem + G::W0;
    char* const opi = w32_smem + G::O0;
    char* const xch = w32_smem + G::X0;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row = KS == 1 ? wave : (wave & 3);                   // this wave's tile row
    const int kk = KS == 1 ? 0 : (wave >> 2);                      // ... and K half (input channels 32 kk ..)
    const int p = lane & 31, kh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.x_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, (int)((size_t)a.N * a.H * a.W * a.y_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, G::WB, 0x00020000);

    // ---- the weights, for good: a linear copy of the packed image (1 KB pieces, wave + 8 i)
#pragma unroll
    for (int i = 0; i < (G::WB / 1024 + 7) / 8; ++i) {
        const int pc = wave + 8 * i;                               // uniform
        if (pc < G::WB / 1024)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (w32_lptr)(wl + pc * 1024), 16, lane * 16, pc * 1024, 0, 0);
    }
    f32x4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + 8 * q + 4 * kh);

    auto decode = [&](int tile, int& n, int& y0, int& x0) {
        const int bx = tile % a.tiles_x;
        const int r = tile / a.tiles_x;
        n = r / a.tiles_y;
        y0 = (r - n * a.tiles_y) * TR;
        x0 = bx * G::TW;
    };
    // ---- the raw patch of a tile, into registers: item it = t + 512 i = (patch position it / NC8, 8-channel group it % NC8), the
    // groups of a pixel in neighbouring lanes (whole lines per request); positions outside the image read zeros ('SAME' padding).
    // What does not depend on the tile is computed once: the item's patch row / column, its byte offset from the patch origin and
    // its place in the operand image (a tile then costs each item two range tests, an add and a select: the walk is lock-step --
    // whatever the waves do between the barriers the matrix pipe waits for).
    f32x4 pre[IPT][2];
    int i_rc[IPT], i_rel[IPT], i_dst[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int it = t + 512 * i;
        const int pos = it / NC8, c8 = it - pos * NC8;
        const int pr = pos / PC, pcx = pos - pr * PC;
        i_rc[i] = it < G::NITEM ? (pr << 8 | pcx) : (1 << 28);              // (row 2^20: no image has it, the item is never in range)
        i_rel[i] = ((pr * a.W + pcx) * a.x_cs + c8 * 8) * 4;
        i_dst[i] = (c8 * 2) * PLANE + pos * 16;
    }
    auto fetch = [&](int n, int y0, int x0) {
        const int base = (((n * a.H + y0 - 1) * a.W) + x0 - 1) * a.x_cs * 4;      // (of the patch origin: may lie outside the image)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int gy = y0 - 1 + (i_rc[i] >> 8), gx = x0 - 1 + (i_rc[i] & 255);
            const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)(base + i_rel[i]) : W32_OOB;
            pre[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
            pre[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 16, 0));
        }
    };
    auto split = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            pwc_f16x4 h0, m0, h1, m1;
            pwc_split4(pre[i][0], h0, m0);
            pwc_split4(pre[i][1], h1, m1);
            char* dst = opi + i_dst[i];
            if (i < IPT - 1 || t + 512 * i < G::NITEM) {
                *reinterpret_cast<pwc_f16x8*>(dst) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<pwc_f16x8*>(dst + PLANE) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    // K step s of this wave = (tap s / JW, 16-channel group kk JW + s % JW): weight fragments (row = output channel lane & 31,
    // K = channels 8 kh ..) and the pixel fragments of the wave's row (column = pixel p, K alike)
    constexpr int JW = G::J16 / KS;                                // 16-channel groups of a wave: 2
    auto frags = [&](int s, pwc_f16x8& ah, pwc_f16x8& am, pwc_f16x8& bh, pwc_f16x8& bm) {
        const int tap = s / JW, j = kk * JW + (s - tap * JW);
        const int dy = tap / 3, dx = tap - 3 * dy;
        const char* ws = wl + (tap * G::J16 + j) * 2048 + lane * 16;
        ah = *reinterpret_cast<const pwc_f16x8*>(ws);
        am = *reinterpret_cast<const pwc_f16x8*>(ws + 1024);
        const char* bs = opi + ((2 * j + kh) * 2) * PLANE + ((row + dy) * PC + p + dx) * 16;
        bh = *reinterpret_cast<const pwc_f16x8*>(bs);
        bm = *reinterpret_cast<const pwc_f16x8*>(bs + PLANE);
    };
    // D fragment: register r of a lane = (output channel 8 (r >> 2) + 4 kh + (r & 3), pixel p)
    auto store_row = [&](const f32x4* v, int n, int y0, int x0) {
        const int oy = y0 + row, ox = x0 + p;
        const bool ok = oy < a.H && ox < a.W;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const unsigned vo = ok ? (unsigned)(((n * a.H + oy) * a.W + ox) * a.y_cs + 8 * q4 + 4 * kh) * 4u : W32_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w32_u32x4, v[q4]), ry, (int)vo, 0, 0);
        }
    };
    auto finish = [&](f32x4* v) {                                  // + bias, leaky-relu
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sum = v[q4][e] + bias4[q4][e];
                v[q4][e] = a.apply_act ? pwc_lrelu(sum, a.slope) : sum;
            }
    };

    int tile = blockIdx.x;
    {
        int n, y0, x0;
        decode(tile < a.ntiles ? tile : 0, n, y0, x0);
        if (tile < a.ntiles) fetch(n, y0, x0);
    }
    // The sums of a tile leave one tile LATER (right behind the next tile's first barrier): their stores then retire under that
    // tile's K loop instead of in front of a barrier that waits for everything in flight.  KS == 2: waves 0-3 add the upper K
    // half's sums (published through the LDS before that barrier) first.
    f32x4 own[4];
    int o_n = 0, o_y0 = 0, o_x0 = 0;
    bool have_prev = false;
    auto flush = [&]() {
        if (kk == 0 && have_prev) {                                // (wave-uniform)
            if (KS == 2) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) own[q4] += *reinterpret_cast<const f32x4*>(xch + ((wave * 64 + lane) * 4 + q4) * 16);
            }
            finish(own);
            store_row(own, o_n, o_y0, o_x0);
        }
    };
    for (; tile < a.ntiles; tile += gridDim.x) {
        int n, y0, x0;
        decode(tile, n, y0, x0);
        // this thread's patch values (and, the first time, its weight pieces) are here; the barrier says that every wave is done
        // with the operand image of the previous tile and that the upper K half has published its sums
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        flush();
        split();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the operand image is complete (and the sums read)
        const int nxt = tile + (int)gridDim.x;
        if (nxt < a.ntiles) {
            int nn, ny0, nx0;
            decode(nxt, nn, ny0, nx0);
            fetch(nn, ny0, nx0);                                   // lands under the K loop
        }
        w32_f32x16 hh, xx;
#pragma unroll
        for (int e = 0; e < 16; ++e) { hh[e] = 0.f; xx[e] = 0.f; }
        pwc_f16x8 ah[3], am[3], bh[3], bm[3];
        frags(0, ah[0], am[0], bh[0], bm[0]);
        frags(1, ah[1], am[1], bh[1], bm[1]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 2 < NS) frags(s + 2, ah[(s + 2) % 3], am[(s + 2) % 3], bh[(s + 2) % 3], bm[(s + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 3], bm[s % 3], xx, 0, 0, 0);
            hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 3], bh[s % 3], hh, 0, 0, 0);
            xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[s % 3], bh[s % 3], xx, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 v[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q4][e] = fmaf(xx[4 * q4 + e], 1.f / 2048.f, hh[4 * q4 + e]);
        if (kk == 1) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<f32x4*>(xch + (((wave - 4) * 64 + lane) * 4 + q4) * 16) = v[q4];
        } else {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) own[q4] = v[q4];
            o_n = n; o_y0 = y0; o_x0 = x0; have_prev = true;
        }
    }
    if (KS == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    flush();
}

// packed[tap][j][hm][lane][e] (fp16): weight of output channel lane & 31, tap, physical input channel 16 j + 8 (lane >> 5) + e
// (the layout of conv3x3_t32.hip, for 32 or 64 physical input channels)
__global__ void conv3x3_w32_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin, int Cin_phys,
                                        _Float16* __restrict__ packed) {
    const int j16 = Cin_phys >> 4;
    const int total = 9 * j16 * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63;
        const int r = idx >> 9;
        const int j = r % j16, tap = r / j16;
        const int cphys = j * 16 + (lane >> 5) * 8 + e, co = lane & 31;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        float v = 0.f;
        if (clog >= 0 && clog < Cin) v = w[((size_t)tap * Cin + clog) * 32 + co];
        const _Float16 h = (_Float16)v;
        const _Float16 mm = (_Float16)fmaf((float)h, -2048.f, v * 2048.f);
        _Float16* dst = packed + (size_t)r * 1024 + lane * 8 + e;
        dst[0] = h;
        dst[512] = mm;
    }
}

extern "C" size_t pwc_conv3x3_w32_packed_floats(int Cin_phys) {
    if (Cin_phys != 32 && Cin_phys != 64) return 0;
    return (size_t)9 * (Cin_phys / 16) * 512;
}

extern "C" int pwc_conv3x3_w32_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, float* packed_w,
                                        pwc_stream_t stream) {
    if (!w_hwio || !packed_w || Cin <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys != 32 && Cin_phys != 64) return PWC_EUNSUPPORTED;
    if (!pwc_aligned16(packed_w)) return PWC_EALIGN;
    hipLaunchKernelGGL(conv3x3_w32_pack_kernel, dim3(72), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map, Cin, Cin_phys,
                       reinterpret_cast<_Float16*>(packed_w));
    return pwc_launch_status();
}

static int w32_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

// C_in (physical) 32 or 64, C_out 32, stride 1, no dilation; 1 where it is the fastest kernel of the library for the shape: a
// launch of at least one tile (8 / 4 rows x 32 columns) per CU
extern "C" int pwc_conv3x3_w32_supported(int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || (Cin_phys != 32 && Cin_phys != 64) || Cout != 32 || stride != 1 || dilation != 1) return 0;
    if ((long)N * H * W * Cin_phys * 4 >= (1L << 31)) return 0;
    const int tr = Cin_phys == 32 ? 8 : 4;
    return (long)N * ((H + tr - 1) / tr) * ((W + 31) / 32) >= 256 ? 1 : 0;
}

template <int CIN>
static int w32_launch(W32Args& a, hipStream_t s) {
    using G = W32Geom<CIN>;
    a.tiles_x = (a.W + 31) / 32; a.tiles_y = (a.H + G::TR - 1) / G::TR;
    const long nt = (long)a.N * a.tiles_x * a.tiles_y;
    if (nt >= (1L << 30)) return PWC_ERANGE;
    a.ntiles = (int)nt;
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_w32_kernel<CIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    }
    int grid = w32_cus();
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL((conv3x3_w32_kernel<CIN>), dim3((unsigned)grid), dim3(512), G::LDS, s, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_w32_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                                   int N, int H, int W, int Cin_phys, int Cout, int apply_act, float slope,
                                   pwc_stream_t stream) {
    if (!x || !packed_w || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0) return PWC_EINVAL;
    if ((Cin_phys != 32 && Cin_phys != 64) || Cout != 32) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_w) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)N * H * W * x_cs * 4 >= (1L << 31) || (long)N * H * W * y_cs * 4 >= (1L << 31)) return PWC_ERANGE;
    W32Args a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W;
    a.apply_act = apply_act; a.slope = slope;
    a.tiles_x = a.tiles_y = a.ntiles = 0;
    hipStream_t s = (hipStream_t)stream;
    return Cin_phys == 32 ? w32_launch<32>(a, s) : w32_launch<64>(a, s);
}
