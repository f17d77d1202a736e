// conv3x3_sk.hip -- 3x3 convolution (+ bias, + leaky-relu) for SMALL launches -- the 7 x 16 and 14 x 32 pyramid levels of a
// batch of 8, every level of a single pair -- on the F16 matrix pipe (round 5).  libpwc_hip.so, gfx950 only.
//
// Replaces (reference modules.py:58-71 feature extractor, modules.py:266-272 estimator convs): tf.layers.Conv2D(f, (3, 3),
// strides, 'same', dilation_rate) + LeakyReLU(0.1) on NHWC fp32.
//
// Why another kernel.  A dispatch on this device does not finish in under ~4.8 us (kernel trace of a forward, round 5:
// profiles/r05_forward_trace_b8.txt), and at 896 ... 3584 output pixels the tiled kernels need a second dispatch to be
// parallel at all: conv3x3_mfma_glds_kernel and conv3x3_wino_kernel deal the taps / channel stages of a tile to several
// workgroups and a reduce launch adds the parts (13.6 + 5.2 us for 0.6 GFLOP at 7 x 16 x 288 -> 128).  Such a layer is a
// chain of round trips, not arithmetic: 1.3 MB of weights and 1 MB of activations that live in the L2, 110 000 matrix
// instructions for 1024 SIMDs.  Here the K dimension (9 taps x C_in) is dealt to the EIGHT WAVES of a workgroup instead: a
// workgroup owns MT x NT tiles of (16 pixels = one 4 x 4 block) x (16 output channels), wave w takes the K steps (one tap x 32
// channels) s = w, w + 8, ... -- at most 11 of them -- and requests ALL its operands at once, straight into registers: nothing
// is shared between waves, so nothing goes through the LDS on the way in.  Activations are fetched as fp32 (one pixel's 8
// channels per lane, taps outside the image from an out-of-range offset: zeros) and split in registers; weights come packed
// as the split halves in fragment order (pwc_conv3x3_sk_pack_f32).  The eight partial tiles meet in the LDS and are added in
// wave order -- a fixed order: launches repeat bitwise.  One dispatch, one round trip for the operands, one for the store.
//
// Arithmetic: the two-term fp16 split of conv3x3_h2.hip (x = h + 2^-11 m', three v_mfma_f32_16x16x32_f16 per K step and tile,
// fp32 accumulation: hh and the cross terms in separate accumulators).  RANGE as there: |x|, |w| < 65504, beyond: NaN.
// Stride 1 or 2, any dilation (a tap is an address).  C_in (physical) % 32 == 0, C_out % 16 == 0.
#include "pwc_common.h"

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));
#define SK_OOB 0x80000000u

struct SkArgs {
    const float* x;
    const float* wp;        // packed split weights: [C_out / 16][nsteps][h | m'][64 lanes][8 fp16]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W, Ho, Wo;
    int Cin_phys, Cout;
    int stride, dil, pad_t, pad_l;
    int apply_act;
    float slope;
    int nbx, nby;           // 4 x 4 output blocks per image row / column
    int ntx;                // workgroup tiles per block row: ceil(nbx / MT)
    int nct;                // output-channel tiles: C_out / (16 NT)
    int cg;                 // C_in / 32
    int nsteps;             // 9 cg
};

// PB: K steps a wave requests at a time (registers: PB (MT + NT) 8)
template <int MT, int NT, int PB>
__global__ __launch_bounds__(512, 2) void conv3x3_sk_kernel(const SkArgs a) {
    __shared__ __attribute__((aligned(16))) float part[8 * MT * NT * 256];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // output-channel tile fastest: the workgroups of one XCD (blockIdx % 8) then share few weight tiles
    int b = blockIdx.x;
    const int ct = b % a.nct;
    b /= a.nct;
    const int tx = b % a.ntx;
    b /= a.ntx;
    const int by = b % a.nby;
    const int n = b / a.nby;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.x_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wp, 0, (int)((size_t)a.Cout * a.nsteps * 128), 0x00020000);

    const int m = lane & 15, kq = lane >> 4;
    // this lane's output pixel in each of the MT blocks: input row / column of tap (0, 0)
    int iy0[MT], ix0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int oy = 4 * by + (m >> 2), ox = 4 * (tx * MT + mt) + (m & 3);
        const bool ok = oy < a.Ho && ox < a.Wo;
        iy0[mt] = ok ? oy * a.stride - a.pad_t : -(1 << 20);
        ix0[mt] = ox * a.stride - a.pad_l;
    }
    const unsigned img = (unsigned)n * (unsigned)(a.H * a.W);

    f32x4 hh[MT][NT], xx[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            hh[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            xx[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    for (int s0 = wave; s0 < a.nsteps; s0 += 8 * PB) {
        f32x4 av[PB][MT][2];
        sk_u32x4 bv[PB][NT][2];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int s = s0 + 8 * i;                       // wave-uniform
            const bool live = s < a.nsteps;
            const int tap = s / a.cg, c32 = s - tap * a.cg;
            const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int iy = iy0[mt] + dy * a.dil, ix = ix0[mt] + dx * a.dil;
                const bool ok = live && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = ok ? ((img + (unsigned)(iy * a.W + ix)) * (unsigned)a.x_cs + (unsigned)(c32 * 32 + kq * 8)) * 4u
                                       : SK_OOB;
                av[i][mt][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
                av[i][mt][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 16, 0));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned vo = live ? (unsigned)(((ct * NT + nt) * a.nsteps + s) * 2048 + lane * 16) : SK_OOB;
                bv[i][nt][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, 0, 0);
                bv[i][nt][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, 1024, 0);
            }
        }
        // every request of the batch is issued before the first value is used (the scheduler otherwise sinks each fetch
        // next to its use to save registers: a round trip per K step)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            pwc_f16x8 ah[MT], am[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                pwc_f16x4 h0, m0, h1, m1;
                pwc_split4(av[i][mt][0], h0, m0);
                pwc_split4(av[i][mt][1], h1, m1);
                ah[mt] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                am[mt] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const pwc_f16x8 bh = __builtin_bit_cast(pwc_f16x8, bv[i][nt][0]);
                const pwc_f16x8 bm = __builtin_bit_cast(pwc_f16x8, bv[i][nt][1]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bm, xx[mt][nt], 0, 0, 0);
                    hh[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, hh[mt][nt], 0, 0, 0);
                    xx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[mt], bh, xx[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- the eight partial tiles meet in the LDS: D fragment = (pixel 4 kq + r, output channel m)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 s = __builtin_elementwise_fma(xx[mt][nt], f32x4{1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f},
                                                      hh[mt][nt]);
            float* p = part + ((wave * MT + mt) * NT + nt) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(4 * kq + r) * 16 + m] = s[r];
        }
    __syncthreads();
    // ---- sum in wave order, bias, leaky-relu, 16-byte stores: item = (tile, pixel, channel quad)
    for (int it = t; it < MT * NT * 64; it += 512) {
        const int tile = it >> 6, px = (it >> 2) & 15, q = it & 3;
        const int mt = tile / NT, nt = tile - mt * NT;
        const int co = (ct * NT + nt) * 16 + q * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int w = 0; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(part + ((w * MT + mt) * NT + nt) * 256 + px * 16 + q * 4);
        if (a.apply_act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = pwc_lrelu(v[k], a.slope);
        }
        const int oy = 4 * by + (px >> 2), ox = 4 * (tx * MT + mt) + (px & 3);
        if (oy < a.Ho && ox < a.Wo)
            *reinterpret_cast<f32x4*>(a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.y_cs + co) = v;
    }
}

// ---- the same with the input PATCH of the workgroup staged in the LDS (stride 1, dilation 1): the workgroup's 6 x (4 MT + 2)
// input pixels are fetched once, 16 bytes per lane in pixel order (whole lines), and the waves read their tap-shifted fragments
// from the LDS.  The direct form asks the texture addressers for every fragment of every tap in operand order (lane = pixel:
// no two neighbouring lanes in one line) -- nine times the bytes at a quarter of the rate; that is what bounds it from a few
// thousand pixels on.  Pixel records are padded by 16 bytes so that the 16 pixels of a fragment read fall into different banks.
template <int MT, int NT, int PB, int S>
__global__ __launch_bounds__(512, 2) void conv3x3_skp_kernel(const SkArgs a) {
    // (stride S: the patch is 3 S + 3 rows x (4 MT - 1) S + 3 columns; up to 288 input channels at stride 1, 128 at stride 2)
    constexpr int PCW = (4 * MT - 1) * S + 3, NPX = (3 * S + 3) * PCW;
    extern __shared__ __attribute__((aligned(16))) char skp_smem[];
    float* const part = reinterpret_cast<float*>(skp_smem);                 // 8 MT NT partial tiles: they take the patch's place
    char* const patch = skp_smem;                                           // when every wave is done with it
    const int ps = a.Cin_phys * 4 + 16;                                     // bytes of a pixel record
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int b = blockIdx.x;
    const int ct = b % a.nct;
    b /= a.nct;
    const int tx = b % a.ntx;
    b /= a.ntx;
    const int by = b % a.nby;
    const int n = b / a.nby;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.x_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wp, 0, (int)((size_t)a.Cout * a.nsteps * 128), 0x00020000);
    const int m = lane & 15, kq = lane >> 4;

    // ---- this wave's weight fragments: all requests first (they land under the patch fetch)
    constexpr int NB = (88 + 8 * PB - 1) / (8 * PB);                        // batches that cover 88 K steps (C_in <= 288)
    sk_u32x4 bv[PB][NT][2];
    auto fetch_b = [&](int s0) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int s = s0 + 8 * i;
            const bool live = s < a.nsteps;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned vo = live ? (unsigned)(((ct * NT + nt) * a.nsteps + s) * 2048 + lane * 16) : SK_OOB;
                bv[i][nt][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, 0, 0);
                bv[i][nt][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)vo, 1024, 0);
            }
        }
    };
    fetch_b(wave);
    // ---- the patch: 16-byte unit u = (pixel u / cq, chunk u % cq), neighbouring lanes ask for neighbouring bytes
    {
        const int cq = a.Cin_phys >> 2;
        const int nu = NPX * cq;
        const int gy0 = 4 * by * S - a.pad_t, gx0 = 4 * tx * MT * S - a.pad_l;
        constexpr int UMAX = (NPX * (S == 1 ? 72 : 32) + 511) / 512;         // units per thread at the widest input
        f32x4 v[UMAX];
        int dst[UMAX];
#pragma unroll
        for (int k = 0; k < UMAX; ++k) {
            const int u = t + 512 * k;
            const int px = u / cq, c = u - px * cq;
            const int pr = px / PCW, pc = px - pr * PCW;
            const int gy = gy0 + pr, gx = gx0 + pc;
            const bool ok = u < nu && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned vo = ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * a.x_cs + c * 4) * 4u : SK_OOB;
            v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vo, 0, 0));
            dst[k] = u < nu ? px * ps + c * 16 : -1;
        }
#pragma unroll
        for (int k = 0; k < UMAX; ++k)
            if (dst[k] >= 0) *reinterpret_cast<f32x4*>(patch + dst[k]) = v[k];
    }
    __syncthreads();

    f32x4 hh[MT][NT], xx[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            hh[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            xx[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // this lane's pixel of block mt in the patch (tap (0, 0)), as a byte offset incl. its channel group
    int pbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pbase[mt] = ((m >> 2) * S * PCW + (4 * mt + (m & 3)) * S) * ps + kq * 32;

#pragma unroll 1
    for (int bi = 0; bi < NB; ++bi) {
        const int s0 = wave + 8 * PB * bi;
        if (s0 >= a.nsteps) break;
        if (bi > 0) fetch_b(s0);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int s = s0 + 8 * i;                               // wave-uniform
            if (s < a.nsteps) {
                const int tap = s / a.cg, c32 = s - tap * a.cg;
                const int dy = tap / 3, dx = tap - 3 * dy;
                const int toff = (dy * PCW + dx) * ps + c32 * 128;
                pwc_f16x8 ah[MT], am[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(patch + pbase[mt] + toff);
                    const f32x4 r1 = *reinterpret_cast<const f32x4*>(patch + pbase[mt] + toff + 16);
                    pwc_f16x4 h0, m0, h1, m1;
                    pwc_split4(r0, h0, m0);
                    pwc_split4(r1, h1, m1);
                    ah[mt] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    am[mt] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const pwc_f16x8 bh = __builtin_bit_cast(pwc_f16x8, bv[i][nt][0]);
                    const pwc_f16x8 bm = __builtin_bit_cast(pwc_f16x8, bv[i][nt][1]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        xx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bm, xx[mt][nt], 0, 0, 0);
                        hh[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, hh[mt][nt], 0, 0, 0);
                        xx[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[mt], bh, xx[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    __syncthreads();          // (the last fragment reads of every wave are behind this)
    // ---- the eight partial tiles meet in the LDS: D fragment = (pixel 4 kq + r, output channel m)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 s = __builtin_elementwise_fma(xx[mt][nt], f32x4{1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f},
                                                      hh[mt][nt]);
            float* p = part + ((wave * MT + mt) * NT + nt) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(4 * kq + r) * 16 + m] = s[r];
        }
    __syncthreads();
    for (int it = t; it < MT * NT * 64; it += 512) {
        const int tile = it >> 6, px = (it >> 2) & 15, q = it & 3;
        const int mt = tile / NT, nt = tile - mt * NT;
        const int co = (ct * NT + nt) * 16 + q * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int w = 0; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(part + ((w * MT + mt) * NT + nt) * 256 + px * 16 + q * 4);
        if (a.apply_act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = pwc_lrelu(v[k], a.slope);
        }
        const int oy = 4 * by + (px >> 2), ox = 4 * (tx * MT + mt) + (px & 3);
        if (oy < a.Ho && ox < a.Wo)
            *reinterpret_cast<f32x4*>(a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.y_cs + co) = v;
    }
}

// packed[cb][s][hm][lane][e] (fp16): weight of output channel 16 cb + (lane & 15), tap s / cg, physical input channel
// 32 (s % cg) + 8 (lane >> 4) + e -- the B fragment of K step s, h halves then m' halves
__global__ void conv3x3_sk_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin, int Cin_phys,
                                       int Cout, _Float16* __restrict__ packed) {
    const int cg = Cin_phys >> 5, nsteps = 9 * cg;
    const size_t total = (size_t)(Cout >> 4) * nsteps * 512;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        size_t r = idx >> 9;
        const int s = (int)(r % nsteps);
        const int cb = (int)(r / nsteps);
        const int tap = s / cg, c32 = s - tap * cg;
        const int cphys = c32 * 32 + (lane >> 4) * 8 + e;
        const int co = cb * 16 + (lane & 15);
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        float v = 0.f;
        if (clog >= 0 && clog < Cin) v = w[((size_t)tap * Cin + clog) * Cout + co];
        const _Float16 h = (_Float16)v;
        const _Float16 mm = (_Float16)fmaf((float)h, -2048.f, v * 2048.f);
        _Float16* dst = packed + ((size_t)(cb * nsteps + s) * 2) * 512 + lane * 8 + e;
        dst[0] = h;
        dst[512] = mm;
    }
}

static bool sk_shape_ok(int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0) return false;
    if (Cin_phys % 32 || Cout % 16 || stride < 1 || stride > 2 || dilation < 1) return false;
    return true;
}

extern "C" size_t pwc_conv3x3_sk_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0 || Cin_phys % 32 || Cout % 16) return 0;
    return (size_t)9 * Cin_phys * Cout;          // two fp16 per weight
}

extern "C" int pwc_conv3x3_sk_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys, int Cout,
                                       float* packed_w, pwc_stream_t stream) {
    if (!w_hwio || !packed_w || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 32 || Cout % 16) return PWC_EUNSUPPORTED;
    if (!pwc_aligned16(packed_w)) return PWC_EALIGN;
    const size_t total = (size_t)9 * Cin_phys * Cout * 2;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_sk_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map, Cin, Cin_phys,
                       Cout, reinterpret_cast<_Float16*>(packed_w));
    return pwc_launch_status();
}

// 1 where this is the fastest kernel of the library for the shape, 0 otherwise; the entry point accepts every shape that meets
// the requirements.  Measured (scripts/exp_sk_ab.py and the kernel traces of forwards with / without this kernel,
// profiles/r05_exp_sk_ab.txt, r05_forward_trace_b8*.txt).  With the patch in the LDS (stride 1, no dilation, up to 288 input
// channels) it wins up to about 2.4e8 multiply-adds (output pixels x C_in x C_out) and 8 K output pixels: 14 x 32 x 256 -> 128 of
// a batch of 8: 16.1 us against 20.8 + 5.0 for the tiled kernel and its reduce dispatch (19.6 in the forward); at 28 x 64 (14 K
// pixels) it is level with the Winograd kernels in the forward (33.8 / 28.2 / 17.6 against 31.8 / 29.9 / 17.1 us) and loses on
// the widest layer (28 x 64 x 224 -> 128: 48.9 against 33.4 on conv3x3_h2_kernel): every workgroup fetches its tile's weights on
// its own, and that traffic grows with the pixel count while the tiled kernels' does not.  With fragments straight from global
// memory (stride 2, dilation): up to 1e8 multiply-adds and 4096 output pixels, beyond that pixel count the stride-2 layers
// (the tiled stride-2 path is the fp32 one: 25.5 against 29.7 at 28 x 64 x 96 -> 128 / 2) and thin layers.
#define PWC_SK_MAX_MACS 100000000L
#define PWC_SK_THIN_MACS 32000000L
#define PWC_SK_LP_MAX_MACS 240000000L
#define PWC_SK_LP_MAX_PIXELS 8192L
#define PWC_SK_S2_MAX_MACS 200000000L
extern "C" int pwc_conv3x3_sk_supported(int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation) {
    if (!sk_shape_ok(N, H, W, Cin_phys, Cout, stride, dilation)) return 0;
    int Ho, Wo, pt, pl;
    pwc_same_pad(H, stride, dilation, &Ho, &pt);
    pwc_same_pad(W, stride, dilation, &Wo, &pl);
    if ((long)N * H * W * Cin_phys * 4 >= (1L << 31)) return 0;
    const long M = (long)N * Ho * Wo, macs = M * Cin_phys * Cout;
    if (stride == 1 && dilation == 1 && Cin_phys >= 96 && Cin_phys <= 288 && M <= PWC_SK_LP_MAX_PIXELS && macs <= PWC_SK_LP_MAX_MACS)
        return 1;
    // stride 2: the tiled alternative is the fp32-pipe kernel (56 x 128 x 64 -> 96 / 2 of 16 images: 34 us against 43 - 46)
    if (stride == 2 && dilation == 1 && Cin_phys >= 64 && Cin_phys <= 128 && macs <= PWC_SK_S2_MAX_MACS) return 1;
    if (macs > PWC_SK_MAX_MACS) return 0;
    return (M <= 4096 || stride == 2 || macs <= PWC_SK_THIN_MACS) ? 1 : 0;
}

template <int MT, int NT, int PB>
static int sk_launch(SkArgs a, hipStream_t s) {
    a.ntx = (a.nbx + MT - 1) / MT;
    a.nct = a.Cout / (16 * NT);
    const long wgs = (long)a.N * a.nby * a.ntx * a.nct;
    if (wgs >= (1L << 31)) return PWC_ERANGE;
    hipLaunchKernelGGL((conv3x3_sk_kernel<MT, NT, PB>), dim3((unsigned)wgs), dim3(512), 0, s, a);
    return pwc_launch_status();
}

template <int MT, int NT, int PB, int S>
static int skp_launch(SkArgs a, hipStream_t s) {
    a.ntx = (a.nbx + MT - 1) / MT;
    a.nct = a.Cout / (16 * NT);
    const long wgs = (long)a.N * a.nby * a.ntx * a.nct;
    if (wgs >= (1L << 31)) return PWC_ERANGE;
    constexpr int NPX = (3 * S + 3) * ((4 * MT - 1) * S + 3);
    size_t lds = (size_t)NPX * (a.Cin_phys * 4 + 16);                      // the patch; the partial tiles reuse its space
    if (lds < (size_t)8 * MT * NT * 1024) lds = (size_t)8 * MT * NT * 1024;
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_skp_kernel<MT, NT, PB, S>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, NPX * ((S == 1 ? 288 : 128) * 4 + 16));
    }
    hipLaunchKernelGGL((conv3x3_skp_kernel<MT, NT, PB, S>), dim3((unsigned)wgs), dim3(512), lds, s, a);
    return pwc_launch_status();
}

#ifdef PWC_HARNESS
// libpwc_hip_harness.so only (scripts/exp_sk_ab.py, exp_sk_tiles.py: pins the tile of every launch of a model forward).  The
// production library has no such state: a caller who wants a given tile passes it to pwc_conv3x3_sk_variant_f32.
static int sk_tile_override = 0;
extern "C" int pwc_debug_conv3x3_sk_tile(int tile) { sk_tile_override = tile; return 0; }
#endif

// tile_req: 0 = the library's choice for the shape; 11, 21, 22 (fragments from global memory), 31, 41, 42 (patch in the LDS) = that
// workgroup tile, PWC_EUNSUPPORTED where the shape does not admit it (x2 tiles: C_out % 32; 3x / 4x: no dilation, up to 288
// input channels at stride 1, 128 at stride 2).
static int sk_run(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                  int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation, int apply_act,
                  float slope, int tile_req, pwc_stream_t stream) {
    if (!x || !packed_w || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || stride < 1 || stride > 2 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 32 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_w) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)N * H * W * x_cs * 4 >= (1L << 31)) return PWC_ERANGE;
    SkArgs a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W;
    pwc_same_pad(H, stride, dilation, &a.Ho, &a.pad_t);
    pwc_same_pad(W, stride, dilation, &a.Wo, &a.pad_l);
    a.Cin_phys = Cin_phys; a.Cout = Cout; a.stride = stride; a.dil = dilation;
    a.apply_act = apply_act; a.slope = slope;
    a.nbx = (a.Wo + 3) / 4; a.nby = (a.Ho + 3) / 4;
    a.cg = Cin_phys / 32; a.nsteps = 9 * a.cg;
    a.ntx = a.nct = 0;
    // Form and tile (measured per layer shape, scripts/exp_sk_ab.py, profiles/r05_exp_sk_ab.txt).  Fragments straight from
    // global memory: 16 pixels x 16 channels up to one workgroup per CU, 32 x 16 up to four, else 32 x 32.  Patch in the LDS
    // (stride 1, no dilation, up to 288 input channels): from 192 input channels on always, from 96 on where the launch has more
    // than one workgroup per CU -- 7 x 16 x 288 -> 128 of a batch of 8: 8.5 us against 12.8; 14 x 32 x 128 -> 128: 9.8 against 13.6;
    // 7 x 16 x 192 -> 192 of 16 images: 11.5 against 17.9; thin layers (7 x 16 x 96 -> 64: 4.6 against 5.0) stay on the direct form.
    const long wgs11 = (long)N * a.nbx * a.nby * (Cout / 16);
    int tile = wgs11 <= 256 ? 11 : (wgs11 <= 1024 || (Cout % 32)) ? 21 : 22;
    const bool lp_ok = dilation == 1 && ((stride == 1 && Cin_phys <= 288) || (stride == 2 && Cin_phys <= 128));
    if (lp_ok && stride == 1 && (Cin_phys >= 192 || (Cin_phys >= 96 && wgs11 > 256)))
        tile = wgs11 <= 256 ? 31 : (wgs11 <= 640 || (Cout % 32)) ? 41 : 42;
    // stride 2 (a 9 x 17-pixel patch for 4 x 8 outputs): 28 x 64 x 96 -> 128 / 2 of 16 images 16.8 us against 22.3 direct,
    // 14 x 32 x 128 -> 192 / 2: 11.3 against 13.8; launches of up to one workgroup per CU stay direct (5.0 against 5.2)
    if (lp_ok && stride == 2 && Cin_phys >= 64 && wgs11 > 256) tile = (wgs11 <= 640 || (Cout % 32)) ? 41 : 42;
    if (tile_req) {
        if (tile_req != 11 && tile_req != 21 && tile_req != 22 && tile_req != 31 && tile_req != 41 && tile_req != 42) return PWC_EINVAL;
        if (((tile_req % 10) == 2 && Cout % 32) || (tile_req > 30 && !lp_ok)) return PWC_EUNSUPPORTED;
        tile = tile_req;
    }
#ifdef PWC_HARNESS
    else if (sk_tile_override && !((sk_tile_override % 10) == 2 && Cout % 32) && !(sk_tile_override > 30 && !lp_ok)) tile = sk_tile_override;
#endif
    hipStream_t s = (hipStream_t)stream;
    switch (tile) {
        case 11: return sk_launch<1, 1, 5>(a, s);
        case 21: return sk_launch<2, 1, 3>(a, s);
        case 22: return sk_launch<2, 2, 2>(a, s);
        case 31: return stride == 1 ? skp_launch<1, 1, 11, 1>(a, s) : skp_launch<1, 1, 11, 2>(a, s);
        case 41: return stride == 1 ? skp_launch<2, 1, 7, 1>(a, s) : skp_launch<2, 1, 7, 2>(a, s);
        default: return stride == 1 ? skp_launch<2, 2, 3, 1>(a, s) : skp_launch<2, 2, 3, 2>(a, s);
    }
}

extern "C" int pwc_conv3x3_sk_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                                  int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation, int apply_act,
                                  float slope, pwc_stream_t stream) {
    return sk_run(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, stride, dilation, apply_act, slope, 0, stream);
}

// The same convolution with the workgroup tile given (tests and tuning: every tile must give the same result on every shape it
// admits) -- an argument of the call, not a process-wide setting.
extern "C" int pwc_conv3x3_sk_variant_f32(const float* x, int x_cs, const float* packed_w, const float* bias, float* y, int y_cs,
                                          int N, int H, int W, int Cin_phys, int Cout, int stride, int dilation, int apply_act,
                                          float slope, int tile, pwc_stream_t stream) {
    if (tile == 0) return PWC_EINVAL;
    return sk_run(x, x_cs, packed_w, bias, y, y_cs, N, H, W, Cin_phys, Cout, stride, dilation, apply_act, slope, tile, stream);
}
