// conv3x3_t32.hip -- 3x3 convolution (+ bias, + leaky-relu) of THIN inputs (C_in = 16 or 32) to 32 output channels, stride 1 or 2,
// on the F16 matrix pipe with the WEIGHTS STATIONARY in registers (round 5).  libpwc_hip.so, gfx950 only.
//
// Replaces (reference modules.py:58-71, the feature extractor's full-resolution layers): tf.layers.Conv2D(32, (3, 3), strides,
// 'same') + LeakyReLU(0.1) on NHWC fp32 -- fp_extractor/conv2d_3 (16 -> 32, stride 2, 224 x 512) and conv2d_4 / conv2d_5
// (32 -> 32, 112 x 256): 64 + 48 + 48 us of conv3x3_h2_kernel launches for 22 + 15 + 15 us of memory traffic.
//
// Why another kernel.  With 16 or 32 input channels the whole weight tensor is 9 or 18 K steps of a 32 x 32 x 16 matrix
// instruction: 72 or 144 registers hold its split halves for the life of a wave, and all 32 output channels are the ROW operand of
// ONE instruction -- so the only operand that moves is the pixel fragment (1 KB per instruction from the LDS: 2 KB of h and m'
// fragments feed three instructions = 85 B/clk per CU, under the LDS's 128).  conv3x3_h2_kernel stages patches AND weight parts
// per 16-channel stage and pays a fixed cost per stage for a channel loop that is over after one or two; a fetch straight from
// global memory in operand order is slower still (profiles/r05_exp_ws_thin_input_conv_dropped.txt).  Here a workgroup walks over
// tiles of 8 rows x 32 pixels; the input patch of tile t + 1 arrives by LDS-DMA (16 bytes per lane, pixel-major: neighbouring lanes
// ask for neighbouring bytes -- a fetch in operand order, lane = pixel, runs at a quarter of the rate) while the four
// waves compute tile t, two rows each; the raw patch is split ONCE into an operand image (h and m' halves, 16 bytes per position
// and 8 channels), so a K step is two ds_read_b128 per row and three matrix instructions per row, no arithmetic per tap; fragments
// are read two steps ahead.  Two barriers per tile.  Stride 2 keeps even and odd patch columns in separate planes, so that the
// 32 pixels of a fragment stay neighbours.
//
// Arithmetic and RANGE of conv3x3_h2.hip: x = h + 2^-11 m' per operand, hh and cross terms in separate fp32 accumulators.
#include "pwc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int t32_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* t32_lptr;
#define T32_OOB 0x80000000u

struct T32Args {
    const float* x;
    const float* wp;        // packed split weights: [tap 9][C_in / 16][h | m'][64 lanes][8 fp16]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W, Ho, Wo;
    int pad_t, pad_l;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ntiles;
    int dbg;                // experiment knob (pwc_debug_conv3x3_t32): 1 no fetches, 2 no matrix work, 4 no stores, 8 no split pass
};

template <int CIN, int S>
struct T32Geom {
    static constexpr int RPW = 1;                                  // output rows per wave (two: 64 more registers, one workgroup per CU: slower)
    static constexpr int TR = 4 * RPW, TW = 32;                    // output rows x columns of a tile
    static constexpr int PR = (TR - 1) * S + 3;                    // patch rows
    static constexpr int PC = (TW - 1) * S + 3;                    // patch columns
    static constexpr int NPAR = S;                                 // column-parity planes of the operand image (stride 2: even | odd)
    static constexpr int PCC = S == 1 ? PC : 33;                   // columns of a parity plane
    static constexpr int QPU = PR * PCC;                           // positions of a plane in use
    static constexpr int PLANE = QPU * 16 + 64;                    // (+64: neighbouring planes start 16 banks apart)
    static constexpr int NCH = CIN / 4;                            // 16-byte channel chunks of a pixel
    static constexpr int NU = PR * PC * NCH;                       // 16-byte units of the raw patch, pixel-major
    static constexpr int NI = (NU + 63) / 64;                      // fetch instructions of a patch
    static constexpr int NFW = (NI + 3) / 4;                       // ... per wave
    static constexpr int RAW = NI * 1024;
    static constexpr int NP8 = (CIN / 8) * NPAR;                   // (8-channel group, parity) planes of the operand image
    static constexpr int OPI = NP8 * 2 * PLANE;                    // h plane + m' plane each
    static constexpr int OPX = 144;                                // bytes of an output pixel record in the LDS (128 + 16: conflict-free)
    static constexpr bool OUTLDS = false;                          // (true: finished rows through the LDS as 1 KB stores -- costs the second workgroup per CU)
    static constexpr int OUT = OUTLDS ? 4 * RPW * 32 * OPX : 0;   // finished rows, one region per wave
    static constexpr int NOW = RPW * 4;                            // 1 KB store instructions per wave and tile
    static constexpr int LDS = RAW + OPI + OUT;
    static constexpr int J16 = CIN / 16;
    static constexpr int NS = 9 * J16;                             // K steps
    static_assert(LDS <= 160 * 1024, "patch buffers exceed the LDS");
};

template <int CIN, int S>
__global__ __launch_bounds__(256) void conv3x3_t32_kernel(const T32Args a) {
    using G = T32Geom<CIN, S>;
    constexpr int J16 = G::J16, NI = G::NI, PCC = G::PCC, RPW = G::RPW, NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) char t32_smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 31, kh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.x_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, (int)((size_t)a.N * a.Ho * a.Wo * a.y_cs * 4), 0x00020000);

    // ---- the weights, for good: K step (tap, 16 channels) -> h and m' fragments (row = output channel lane & 31)
    pwc_f16x8 wh[9][J16], wm[9][J16];
    {
        const t32_u32x4* src = reinterpret_cast<const t32_u32x4*>(a.wp) + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < J16; ++j) {
                wh[tap][j] = __builtin_bit_cast(pwc_f16x8, src[(tap * J16 + j) * 128]);
                wm[tap][j] = __builtin_bit_cast(pwc_f16x8, src[(tap * J16 + j) * 128 + 64]);
            }
    }
    f32x4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + 8 * q + 4 * kh);

    auto decode = [&](int tile, int& n, int& y0, int& x0) {
        const int bx = tile % a.tiles_x;
        const int r = tile / a.tiles_x;
        n = r / a.tiles_y;
        y0 = (r - n * a.tiles_y) * G::TR;
        x0 = bx * G::TW;
    };
    // fetch instruction jj of this wave for the raw patch of a tile: 16-byte unit u = (patch pixel u / NCH, chunk u % NCH),
    // pixel-major -- neighbouring lanes ask for neighbouring 16 bytes (a pixel's channels, then the next pixel of the row: whole
    // lines), and the LDS image is the same order
    auto fetch_one = [&](int jj, int n, int gy0, int gx0) {
        const int i = wave + 4 * jj;                              // uniform
        if (i < NI) {
            const int u = i * 64 + lane;
            const int px = u / G::NCH, c = u - px * G::NCH;
            const int pr = px / G::PC, pc = px - pr * G::PC;
            const int gy = gy0 + pr, gx = gx0 + pc;
            const bool ok = pr < G::PR && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * a.x_cs + c * 4) * 4u : T32_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (t32_lptr)(t32_smem + i * 1024), 16, (int)vo, 0, 0, 0);
        }
    };

    // LDS: the raw patch (fp32, where the fetches land), the OPERAND image made from it once per tile -- per 8-channel group and
    // parity plane the h halves of every position (16 bytes each), then the m' halves: a fragment of a K step is one 16-byte read
    // of each, no arithmetic per tap -- and the finished rows of each wave (pixel records of 32 channels), which leave one tile
    // later as whole 1 KB store instructions
    char* const raw = t32_smem;
    char* const opi = t32_smem + G::RAW;
    char* const outw = t32_smem + G::RAW + G::OPI + wave * (RPW * 32 * G::OPX);

    // store instruction k of this wave for the rows finished one tile ago: unit u = 64 k + lane = (row, pixel, 16-byte piece)
    int o_n = 0, o_y0 = 0, o_x0 = 0;
    bool have_out = false;
    f32x4 held[RPW][4];
    auto store_one = [&](int k) {
        if constexpr (G::OUTLDS) {
            const int u = k * 64 + lane;
            const int r = u >> 8, pix = (u >> 3) & 31, piece = u & 7;
            const t32_u32x4 v = *reinterpret_cast<const t32_u32x4*>(outw + (r * 32 + pix) * G::OPX + piece * 16);
            const int oy = o_y0 + wave * RPW + r, ox = o_x0 + pix;
            const unsigned vo = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((o_n * a.Ho + oy) * a.Wo + ox) * a.y_cs + piece * 4) * 4u : T32_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, ry, (int)vo, 0, 0);
        } else {
            const int r = k >> 2, q4 = k & 3;
            const int oy = o_y0 + wave * RPW + r, ox = o_x0 + p;
            const unsigned vo = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((o_n * a.Ho + oy) * a.Wo + ox) * a.y_cs + 8 * q4 + 4 * kh) * 4u : T32_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(t32_u32x4, held[r][q4]), ry, (int)vo, 0, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.ntiles && !(a.dbg & 1)) {
        int n, y0, x0;
        decode(tile, n, y0, x0);
#pragma unroll
        for (int jj = 0; jj < G::NFW; ++jj) fetch_one(jj, n, y0 * S - a.pad_t, x0 * S - a.pad_l);
    }
    for (; tile < a.ntiles; tile += gridDim.x) {
        // this wave's pieces have landed; the barrier publishes everybody's -- and says that every wave is done with the operand image
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // item = (patch pixel, 8-channel group), group fastest: neighbouring lanes read neighbouring 32 bytes of the raw patch
        if (!(a.dbg & 8))
        for (int it = t; it < G::PR * G::PC * (CIN / 8); it += 256) {
            const int px = it / (CIN / 8), c8 = it - px * (CIN / 8);
            const int pr = px / G::PC, pc = px - pr * G::PC;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(raw + it * 32);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(raw + it * 32 + 16);
            pwc_f16x4 h0, m0, h1, m1;
            pwc_split4(r0, h0, m0);
            pwc_split4(r1, h1, m1);
            const int par = S == 1 ? 0 : (pc & 1);
            const int q = pr * PCC + (S == 1 ? pc : (pc >> 1));
            char* dst = opi + ((c8 * G::NPAR + par) * 2) * G::PLANE + q * 16;
            *reinterpret_cast<pwc_f16x8*>(dst) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<pwc_f16x8*>(dst + G::PLANE) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        // the operand image is complete, the raw patch free
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int n, y0, x0;
        decode(tile, n, y0, x0);
        const int nxt = tile + (int)gridDim.x;
        const bool more = nxt < a.ntiles && !(a.dbg & 1);
        int nn = 0, ny0 = 0, nx0 = 0;
        if (more) decode(nxt, nn, ny0, nx0);
        const int ngy0 = ny0 * S - a.pad_t, ngx0 = nx0 * S - a.pad_l;
        const bool outs = have_out && !(a.dbg & 4);
        // the wave's rows in one walk over the K steps (one weight fragment, RPW pixel fragments: independent accumulator chains);
        // fragments are read two steps ahead.  The next tile's fetches and the previous tile's stores are dealt out over the steps:
        // a wave that issues them in a bunch waits for the address units before it can issue a matrix instruction.
        f32x16 hh[RPW], xx[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hh[r][e] = 0.f; xx[r][e] = 0.f; }
        auto frag = [&](int s, int r, pwc_f16x8& fh, pwc_f16x8& fm) {
            const int tap = s / J16, jj = s - tap * J16;
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int par = S == 1 ? 0 : (dx & 1);
            const int q = ((wave * RPW + r) * S + dy) * PCC + (S == 1 ? p + dx : p + (dx >> 1));
            const char* src = opi + (((jj * 2 + kh) * G::NPAR + par) * 2) * G::PLANE + q * 16;
            fh = *reinterpret_cast<const pwc_f16x8*>(src);
            fm = *reinterpret_cast<const pwc_f16x8*>(src + G::PLANE);
        };
        constexpr int FPS = (G::NFW + NS - 1) / NS, OPS = (G::NOW + NS - 1) / NS;     // fetches / stores per K step
        pwc_f16x8 fh[3][RPW], fm[3][RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) { frag(0, r, fh[0][r], fm[0][r]); frag(1, r, fh[1][r], fm[1][r]); }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 2 < NS) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) frag(s + 2, r, fh[(s + 2) % 3][r], fm[(s + 2) % 3][r]);
            }
            if (outs) {
#pragma unroll
                for (int k = s * OPS; k < (s + 1) * OPS && k < G::NOW; ++k) store_one(k);
            }
            if (more) {
#pragma unroll
                for (int jj = s * FPS; jj < (s + 1) * FPS && jj < G::NFW; ++jj) fetch_one(jj, nn, ngy0, ngx0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(a.dbg & 2)) {
                const int tap = s / J16, jj = s - tap * J16;
#pragma unroll
                for (int r = 0; r < RPW; ++r) xx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tap][jj], fm[s % 3][r], xx[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RPW; ++r) hh[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tap][jj], fh[s % 3][r], hh[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RPW; ++r) xx[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm[tap][jj], fh[s % 3][r], xx[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // D fragment: register r of lane = (output channel 8 (r >> 2) + 4 kh + (r & 3), pixel p) -> this wave's region of finished
        // rows (its reads of the previous tile's rows are behind it: LDS operations of a wave complete in order)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sum = fmaf(xx[r][4 * q4 + e], 1.f / 2048.f, hh[r][4 * q4 + e]) + bias4[q4][e];
                    v[e] = a.apply_act ? pwc_lrelu(sum, a.slope) : sum;
                }
                if constexpr (G::OUTLDS) *reinterpret_cast<f32x4*>(outw + (r * 32 + p) * G::OPX + q4 * 32 + kh * 16) = v;
                else held[r][q4] = v;
            }
        o_n = n; o_y0 = y0; o_x0 = x0; have_out = true;
    }
    if (have_out && !(a.dbg & 4)) {
#pragma unroll
        for (int k = 0; k < G::NOW; ++k) store_one(k);
    }
}

// packed[tap][j][hm][lane][e] (fp16): weight of output channel lane & 31, tap, physical input channel 16 j + 8 (lane >> 5) + e
__global__ void conv3x3_t32_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin, int Cin_phys,
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

extern "C" size_t pwc_conv3x3_t32_packed_floats(int Cin_phys) {
    if (Cin_phys != 16 && Cin_phys != 32) return 0;
    return (size_t)9 * (Cin_phys / 16) * 512;
}

extern "C" int pwc_conv3x3_t32_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, float* packed_w,
                                        pwc_stream_t stream) {
    if (!w_hwio || !packed_w || Cin <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys != 16 && Cin_phys != 32) return PWC_EUNSUPPORTED;
    if (!pwc_aligned16(packed_w)) return PWC_EALIGN;
    hipLaunchKernelGGL(conv3x3_t32_pack_kernel, dim3(36), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map, Cin, Cin_phys,
                       reinterpret_cast<_Float16*>(packed_w));
    return pwc_launch_status();
}

// C_in (physical) 16 (stride 1 | 2) or 32 (stride 1), C_out 32, no dilation; 1 where it is the fastest kernel of the library for the shape
// (at least 256 tiles of 8 x 32 -- stride 2: 4 x 32 -- output pixels: one per CU)
extern "C" int pwc_conv3x3_t32_supported(int N, int H, int W, int Cin_phys, int Cout, int stride) {
    if (N <= 0 || H <= 0 || W <= 0 || (Cin_phys != 16 && Cin_phys != 32) || Cout != 32 || stride < 1 || stride > 2) return 0;
    if (Cin_phys == 32 && stride == 2) return 0;                 // (two parity planes of 32 channels do not fit the LDS twice)
    // 32 input channels: the entry point takes them, but 144 weight registers beside the fragments and accumulators leave one wave
    // per SIMD (or spill) and the launch loses to conv3x3_h2_kernel (52 - 77 us against 48 at 16 x 112 x 256; in the forward
    // 2.599 against 2.582 ms)
    if (Cin_phys == 32) return 0;
    // 32 input channels: the entry point takes them, but 144 weight registers beside the fragments and accumulators spill into
    // the accumulation registers and the launch loses to conv3x3_h2_kernel (52 - 73 us against 48 at 16 x 112 x 256)
    if ((long)N * H * W * Cin_phys * 4 >= (1L << 31)) return 0;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    if ((long)N * Ho * Wo * Cout * 4 >= (1L << 31)) return 0;   // (a dense output; a wider y_cs is the entry point's PWC_ERANGE)
    const int tr = 4;
    return (long)N * ((Ho + tr - 1) / tr) * ((Wo + 31) / 32) >= 256 ? 1 : 0;
}

#ifdef PWC_HARNESS
// libpwc_hip_harness.so only (scripts/exp_t32_ab.py: ablations -- 1 no fetches, 2 no matrix work, 4 no stores, 8 no split pass)
static int t32_dbg = 0;
extern "C" int pwc_debug_conv3x3_t32(int bits) { t32_dbg = bits; return 0; }
#endif

template <int CIN, int S>
static int t32_launch(T32Args& a, hipStream_t s) {
#ifdef PWC_HARNESS
    a.dbg = t32_dbg;
#else
    a.dbg = 0;
#endif
    using G = T32Geom<CIN, S>;
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_t32_kernel<CIN, S>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int per_cu = (160 * 1024) / G::LDS > 0 ? (160 * 1024) / G::LDS : 1;
    int grid = cus * per_cu;
    if (grid > a.ntiles) grid = a.ntiles;
    hipLaunchKernelGGL((conv3x3_t32_kernel<CIN, S>), dim3((unsigned)grid), dim3(256), G::LDS, s, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_t32_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                                   int N, int H, int W, int Cin_phys, int Cout, int stride, int apply_act, float slope,
                                   pwc_stream_t stream) {
    if (!x || !packed_w || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || stride < 1 || stride > 2) return PWC_EINVAL;
    if ((Cin_phys != 16 && Cin_phys != 32) || Cout != 32 || (Cin_phys == 32 && stride == 2)) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_w) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)N * H * W * x_cs * 4 >= (1L << 31)) return PWC_ERANGE;
    {   // the OUTPUT's buffer resource and store offsets are 32-bit too (a stride-1 launch writes twice the input's
        // bytes, a slice of a wide buffer more)
        const long Ho_ = (H + stride - 1) / stride, Wo_ = (W + stride - 1) / stride;
        if ((long)N * Ho_ * Wo_ * y_cs * 4 >= (1L << 31)) return PWC_ERANGE;
    }
    T32Args a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W;
    pwc_same_pad(H, stride, 1, &a.Ho, &a.pad_t);
    pwc_same_pad(W, stride, 1, &a.Wo, &a.pad_l);
    a.apply_act = apply_act; a.slope = slope;
    const int tr = 4;          // T32Geom::TR
    a.tiles_x = (a.Wo + 31) / 32; a.tiles_y = (a.Ho + tr - 1) / tr;
    const long nt = (long)N * a.tiles_x * a.tiles_y;
    if (nt >= (1L << 30)) return PWC_ERANGE;
    a.ntiles = (int)nt;
    hipStream_t s = (hipStream_t)stream;
    if (Cin_phys == 16) return stride == 1 ? t32_launch<16, 1>(a, s) : t32_launch<16, 2>(a, s);
    return t32_launch<32, 1>(a, s);
}
