// conv3x3_c16pair.hip -- TWO chained 3x3 stride-1 'SAME' convolutions of 16 channels each (16 -> 16 -> 16), leaky-relu behind
// each, in ONE launch, on the F16 matrix pipe of gfx950 with the exact-to-22-bit operand splits of conv3x3_h2.hip (round 4).
//
// Replaces the two tf.layers.Conv2D(16, (3,3), (1,1), 'same') + tf.nn.leaky_relu calls behind the stride-2 convolution of
// pyramid level 1 (reference modules.py:62-67 `fp_extractor/conv2d_1`, `conv2d_2`).  The intermediate (117 MB for a batch of 8
// pairs at 448 x 1024) never leaves the CU: a workgroup computes a 16 x 32-pixel output tile from the 20 x 36-pixel input patch
// through the 18 x 34-pixel intermediate held in LDS.  As two launches of the fp32 Winograd kernel the pair takes 160 us (each
// moves 235 MB and runs at 30 % of the fp32 matrix pipe); here the matrix work is 14 v_mfma_f32_16x16x32_f16 per 16 pixels and
// layer.
//
// Arithmetic: as conv3x3_h2.hip -- x = h + 2^-11 m', three products, two fp32 accumulators (hh, cross); the intermediate is
// rounded to fp32 once (bias, leaky-relu) and split again, exactly like an fp32 tensor read back from memory would be.
// K packing of a 16x16x32 MFMA (M = couts, N = 16 pixels; lane quarter kq holds K 8 kq .. 8 kq + 7):
//   hh, tap pair (2j, 2j+1):  A = [uh(tap 2j) ch 0-7 | ch 8-15 | uh(tap 2j+1) ch 0-7 | ch 8-15]   B = vh shifted by the quarter's tap
//   cross, tap t:             A = [uh ch 0-7 | uh ch 8-15 | um' ch 0-7 | um' ch 8-15]              B = [vm' 0-7 | vm' 8-15 | vh 0-7 | vh 8-15]
//   = 5 + 9 = 14 instructions per 16 pixels and layer; the 28 weight operands of the two layers stay in registers.
// LDS images are [chunk: h 0-7, h 8-15, m' 0-7, m' 8-15][pixel, row pitch 36][16 bytes]: a tile of 16 CONSECUTIVE linear pixel
// indices p (it may wrap from one patch row to the next) reads 16 consecutive 16-byte records per chunk at p + (dy-1) 36 + (dx-1)
// -- wrapped pixels are columns nobody needs.  Intermediate pixels outside the IMAGE are the second convolution's zero padding,
// not convolution results: they are zeroed before they are split.
// Where the time goes (profiles/r04_exp_c16pair.txt, 16 x 224 x 512): 100 us; the two layers alone 74, fetch + split + barriers
// alone 28.  A 16x16x32 MFMA with only 16 couts as M reads a fresh 1 KB pixel operand per instruction: 4 SIMDs x 1 KB per 16
// cycles IS the LDS's 256 B/clk, so the layers are LDS-read-bound at the matrix pipe's own rate (35 us each way, not
// overlapping fully); re-fetching a consumed fragment for the next tile behind its instruction (rolling prefetch, three
// accumulator chains) made it 109 us.  What would halve the reads: v_mfma_f32_32x32x16_f16 with M = [uh | um'] stacked (hh and
// um' vh from ONE pixel operand, rows r and r + 16 of a lane's accumulator) -- 2 KB and 64 pipe cycles per tap and 32 pixels
// instead of 3.1 KB and 50.
// Persistent: one workgroup per CU loops over tiles; the patch of the next tile is fetched (LDS-DMA into the staging image)
// under the two layers of the current one.  Three barriers per tile.
//
// FIRST form (pwc_conv3x3_c3c16pair_f32): ALL of pyramid level 1 from the raw frames.  The staging area receives the 41 x 73-pixel
// 3-channel raw patch instead (one 16-byte-chunk fetch instruction per raw row, rows of whole chunks: W0 % 4 == 0; chunks beyond
// the row end and rows beyond the image are the 'SAME' zeros), and the split stage becomes the stride-2 convolution
// (`fp_extractor/conv2d`, reference modules.py:57-61): per 16 patch pixels each lane reads its 8 of the 27 (+5 zero) K values
// straight from the raw rows, splits them, and three 16x16x32 instructions (hh, two cross) produce the 16 channels that are
// biased, activated, zeroed outside the image, split and written into the operand image the pair reads.  165 -> 123 us for the
// three layers at 16 x 448 x 1024 (profiles/r04_exp_c16pair.txt): the 117 MB level-1 input of the pair is never written or read.
// The stride-2 stage costs 42 us of the 123 and is VALU ISSUE (2 waves per SIMD x 6 tiles x 4 cycles an instruction; about 100
// vector instructions per 16 pixels with the compiler's 3.5-instruction operand split, 131 us; c16_split2 made it 123).  Running it for
// tile t + 1 BESIDE layer 2 of tile t (waves 0-3 one order, waves 4-7 the other, two barriers per tile) changed nothing
// (130.8 us): every wave still executes both in series and neither saturates a unit the other needs.  Delaying waves 4-7 by
// 256 - 900 cycles behind every barrier (so that one wave of a SIMD reads while the other multiplies) ADDS the delay: 126 -> 126 /
// 128 / 129 us.  The layers are the serial latency of a wave's read -> 14 dependent-pair MFMAs -> epilogue chain at two waves
// per SIMD, not contention for the LDS or the matrix pipe.  Issuing a wave's NEXT tile's 14 reads behind the matrix
// instructions of the current one (in front of its epilogue) needs a second fragment set in the allocator's eyes: 256
// registers + 44 / 54 spilled -- the 28 weight operands (112 registers) leave no room for it.
#pragma once
#include "pwc_common.h"
#include <type_traits>

typedef _Float16 c16_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c16_f16x4 __attribute__((ext_vector_type(4)));

typedef _Float16 c16_f16x2 __attribute__((ext_vector_type(2)));

// The operand split of TWO fp32 values in four vector instructions: h = fp16(x) of both (v_cvt_pk_f16_f32), x 2^11 of both
// (v_pk_mul_f32), and m' = fp16(fma(h, -2^11, x 2^11)) = fp16((x - h) 2^11) per value with the fp16 half read in place and
// the result written to its half of the pair (v_fma_mixlo / mixhi_f16): the same values as the scalar form
//   h = (_Float16)x;  m' = (_Float16)fmaf((float)h, -2048.f, x * 2048.f);
// which the compiler turns into seven.
__device__ __forceinline__ void c16_split2(const float x0, const float x1, unsigned& h_pair, unsigned& m_pair) {
    const f32x2 xs = {x0, x1};
    const c16_f16x2 h2 = __builtin_convertvector(xs, c16_f16x2);
    const f32x2 xm = xs * 2048.f;
    const unsigned hp = __builtin_bit_cast(unsigned, h2);
    const float neg = -2048.f;
    unsigned mp;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(mp) : "v"(hp), "s"(neg), "v"(xm[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(mp) : "v"(hp), "s"(neg), "v"(xm[1]));
    h_pair = hp; m_pair = mp;
}
__device__ __forceinline__ void c16_split4(const float x0, const float x1, const float x2, const float x3, c16_f16x4& h, c16_f16x4& m) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hp, mp;
    unsigned a, b;
    c16_split2(x0, x1, a, b); hp[0] = a; mp[0] = b;
    c16_split2(x2, x3, a, b); hp[1] = a; mp[1] = b;
    h = __builtin_bit_cast(c16_f16x4, hp);
    m = __builtin_bit_cast(c16_f16x4, mp);
}

struct C16Args {
    const float* x;
    const void* wp;          // [layer 2][operand 14][lane 64][8 fp16]
    const float* b1;
    const float* b2;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    float slope;
    int tiles_x, tiles_y, ntiles;
    // FIRST form only: the raw images (N_a from x, the others from x_b; 3 channels, channel stride 3), H0 x W0, stride-2 'SAME'
    const float* x_b;
    const float* b0;
    int N_a, H0, W0, pad_t;
};

constexpr int C16_P = 36;                      // patch width = row pitch of the LDS images
constexpr int C16_PR = 20;                     // patch rows
constexpr int C16_NPIX = C16_P * C16_PR;       // 720
constexpr int C16_CH = C16_NPIX * 16;          // bytes of a chunk plane: 11 520
constexpr int C16_IMG = 4 * C16_CH;            // 46 080
constexpr int C16_NPC = 48;                    // 1 KB staging pieces (45 hold records)
constexpr int C16_S0 = 0, C16_IN0 = C16_NPC * 1024, C16_MID0 = C16_IN0 + C16_IMG, C16_LDS = C16_MID0 + C16_IMG + 1024;
constexpr int C16_MT_A = 41;                   // 16-pixel tiles of the intermediate: linear pixels [36, 692)
constexpr int C16_MT_B = 36;                   // ... of the output: [72, 648) = rows 2 .. 17
constexpr unsigned C16_OOB = 0x7FFF0000u;
// FIRST form: the raw patch is 41 rows x 73 pixels x 3 floats = 55 16-byte chunks a row; a row's fetch instruction writes 1 KB, the
// row pitch of 1072 bytes keeps the three rows a pixel reads 12 banks apart (41 x 1072 = 43 952 bytes of the 48 KB staging area)
constexpr int C16_RAW_ROWS = 41, C16_RAW_CHUNKS = 55, C16_RAW_PITCH = 1072;

// ABL (harness only): 1 = no patch DMA, 2 = no split of the staging image (FIRST: no stride-2 convolution), 4 = no layer 1,
// 8 = no layer 2
// FIRST: the input is the raw 3-channel image pair and the stride-2 convolution in front of the two layers (`conv2d`, reference
// modules.py:60-61) is computed into the operand image instead of the split of a fetched 16-channel patch.
template <int ABL = 0, bool FIRST = false>
__global__ __launch_bounds__(512) void conv3x3_c16pair_kernel(const C16Args a) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float c16_smem[];
    char* const sm = reinterpret_cast<char*>(c16_smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n16 = lane & 15, kq = lane >> 4;

    // ---- the 28 weight operands and the lane's four biases of each layer
    c16_f16x8 A1[14], A2[14];
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        A1[m] = reinterpret_cast<const c16_f16x8*>(a.wp)[m * 64 + lane];
        A2[m] = reinterpret_cast<const c16_f16x8*>(a.wp)[(14 + m) * 64 + lane];
    }
    c16_f16x8 A0h = {}, A0m = {};
    f32x4 b0v = {0.f, 0.f, 0.f, 0.f};
    if (FIRST) {
        A0h = reinterpret_cast<const c16_f16x8*>(a.wp)[28 * 64 + lane];
        A0m = reinterpret_cast<const c16_f16x8*>(a.wp)[29 * 64 + lane];
        b0v = *reinterpret_cast<const f32x4*>(a.b0 + 4 * kq);
    }
    const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + 4 * kq);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + 4 * kq);

    // ---- per-lane fragment offsets (bytes into an image, for the tile at linear pixel 0)
    auto tap_shift = [](int tp) { return (tp / 3) * C16_P + (tp % 3) - (C16_P + 1); };
    int hoff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tp = 2 * j + (kq >> 1) > 8 ? 8 : 2 * j + (kq >> 1);      // (the tenth tap has zero weights)
        hoff[j] = (kq & 1) * C16_CH + (n16 + tap_shift(tp)) * 16;
    }
    const int coff = ((kq + 2) & 3) * C16_CH + n16 * 16;

    const int G = (int)gridDim.x;
    const int lw = pwc_xcd_remap(blockIdx.x, G);
    const int per = (a.ntiles + G - 1) / G;
    const int tile0 = lw * per, tile1 = tile0 + per < a.ntiles ? tile0 + per : a.ntiles;      // a contiguous run: neighbours share halos in L2

    // ---- patch fetch: piece pc = wave + 8 i holds records 16 pc .. 16 pc + 15 (record = patch pixel, 64 bytes)
    unsigned p_voff[6];
    __amdgpu_buffer_rsrc_t xrsrc;
    auto patch_prepare = [&](int tile) {
        const int bx = tile % a.tiles_x;
        const int rest = tile / a.tiles_x;
        const int by = rest % a.tiles_y, n = rest / a.tiles_y;
        if (FIRST) {
            const float* img = n < a.N_a ? a.x + (size_t)n * a.H0 * a.W0 * 3 : a.x_b + (size_t)(n - a.N_a) * a.H0 * a.W0 * 3;
            xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, a.H0 * a.W0 * 12, 0x00020000);
            const int rowb = a.W0 * 12;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int r = wave + 8 * i;
                const int c = lane;
                const int yy = 2 * (by * 16 - 2) - a.pad_t + r;
                const int cb = (2 * (bx * 32 - 2)) * 12 + 16 * c;                    // first byte of the chunk within the image row
                const bool ok = r < C16_RAW_ROWS && (unsigned)c < (unsigned)C16_RAW_CHUNKS && (unsigned)yy < (unsigned)a.H0 &&
                                (unsigned)cb < (unsigned)rowb;
                p_voff[i] = ok ? (unsigned)(yy * rowb + cb) : C16_OOB;
            }
            return;
        }
        xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int rec = (wave + 8 * i) * 16 + (lane >> 2);
            const int pr = rec / C16_P, pc = rec - pr * C16_P;
            const int yy = by * 16 - 2 + pr, xx = bx * 32 - 2 + pc;
            const bool ok = rec < C16_NPIX && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            p_voff[i] = ok ? (unsigned)(((yy * a.W + xx) * a.x_cs + (lane & 3) * 4) * 4) : C16_OOB;
        }
    };
    auto patch_issue = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (FIRST) {
                const int roff = (wave + 8 * i) * C16_RAW_PITCH;
                if (!(ABL & 1) && wave + 8 * i < C16_RAW_ROWS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sm + C16_S0 + roff), 16, (int)p_voff[i], 0, 0, 0);
            } else if (!(ABL & 1)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sm + C16_S0 + (wave + 8 * i) * 1024), 16, (int)p_voff[i], 0, 0, 0);
            }
        }
    };
#define C16_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // one 16-pixel tile of a layer: 14 fragment reads, 14 matrix instructions; returns hh + 2^-11 cross
    auto mtile = [&](const char* img, int p0, const c16_f16x8* A, const f32x4 bias) -> f32x4 {
        c16_f16x8 B[14];
        const char* base = img + p0 * 16;
#pragma unroll
        for (int j = 0; j < 5; ++j) B[j] = *reinterpret_cast<const c16_f16x8*>(base + hoff[j]);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) B[5 + tp] = *reinterpret_cast<const c16_f16x8*>(base + coff + tap_shift(tp) * 16);
        f32x4 hh = bias, cx = {0.f, 0.f, 0.f, 0.f};        // (the bias rides in the accumulator)
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            // (alternating accumulators keeps two dependent chains in flight)
            if (m < 5) hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m], B[m], hh, 0, 0, 0);
            if (m + 5 < 14) cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[m + 5], B[m + 5], cx, 0, 0, 0);
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hh[e] + cx[e] * (1.f / 2048.f);
        return o;
    };

    // FIRST: the lane's pixel of the stride-2 layer's tile `wave` (16 consecutive linear patch pixels a tile), its source in the
    // raw staging rows (quarter dy < 3: floats 0 .. 7 of raw row 2 pr + dy at column 2 pc; quarter 3: float 8 of the three rows)
    // and its destination in the operand image
    const int pr_first = (16 * wave + n16) / C16_P, pc_first = (16 * wave + n16) - pr_first * C16_P;
    const int src_lane = kq < 3 ? kq * C16_RAW_PITCH : 32;
    const int dst_lane = (kq >> 1) * C16_CH + (16 * wave + n16) * 16 + (kq & 1) * 8;
    if (tile0 < tile1) {
        patch_prepare(tile0);
        patch_issue();
    }
    for (int tile = tile0; tile < tile1; ++tile) {
        const int bx = tile % a.tiles_x;
        const int rest = tile / a.tiles_x;
        const int by = rest % a.tiles_y, n = rest / a.tiles_y;
        const int y0 = by * 16, x0 = bx * 32;
        const bool interior = y0 >= 2 && y0 + 18 <= a.H && x0 >= 2 && x0 + 34 <= a.W;      // no patch pixel outside the image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C16_BAR();                                   // the patch has landed; everybody is done with the images of the previous tile
        if (FIRST) {
            // ---- the stride-2 convolution of the raw patch: 45 tiles of 16 patch pixels, K = 27 of 32 (quarter dy < 3: the first
            // 8 of the 9 floats of raw row 2 pr + dy at columns 2 pc .. 2 pc + 2; quarter 3: the ninth float of the three rows),
            // one hh and two cross instructions; bias, leaky-relu, zero outside the image, split, into the operand image
            // (three tiles at a time: their reads go out together, the latencies of the three chains overlap; a lane's pixel of
            // tile wave + 8 i is 128 i linear pixels = 3 rows and 20 columns behind its pixel of tile `wave`)
            auto stage0 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            int pr = pr_first, pc = pc_first;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float v[3][8];
                bool in_image[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = 3 * g + u;
                    const char* src = sm + C16_S0 + pr * (2 * C16_RAW_PITCH) + pc * 24 + src_lane;
                    in_image[u] = INTERIOR || ((unsigned)(y0 - 2 + pr) < (unsigned)a.H && (unsigned)(x0 - 2 + pc) < (unsigned)a.W);
                    pc += 20; pr += 3;
                    if (pc >= C16_P) { pc -= C16_P; pr += 1; }
                    if (wave + 8 * i >= ((ABL & 2) ? 0 : C16_NPIX / 16)) continue;
                    if (kq < 3) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 q = *reinterpret_cast<const f32x2*>(src + 8 * e);
                            v[u][2 * e] = q[0]; v[u][2 * e + 1] = q[1];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
#pragma unroll
                        for (int e = 0; e < 3; ++e) v[u][e] = *reinterpret_cast<const float*>(src + e * C16_RAW_PITCH);
                    }
                }
                f32x4 hh[3], cx[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    c16_f16x4 hl, ml, hu, mu;
                    c16_split4(v[u][0], v[u][1], v[u][2], v[u][3], hl, ml);
                    c16_split4(v[u][4], v[u][5], v[u][6], v[u][7], hu, mu);
                    const c16_f16x8 Bh = __builtin_shufflevector(hl, hu, 0, 1, 2, 3, 4, 5, 6, 7);
                    const c16_f16x8 Bm = __builtin_shufflevector(ml, mu, 0, 1, 2, 3, 4, 5, 6, 7);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    hh[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0h, Bh, b0v, 0, 0, 0);        // (the bias rides in the accumulator)
                    cx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0h, Bm, z, 0, 0, 0);
                    cx[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0m, Bh, cx[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = 3 * g + u;
                    if (wave + 8 * i >= ((ABL & 2) ? 0 : C16_NPIX / 16)) continue;
                    c16_f16x4 h, m;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = hh[u][e] + cx[u][e] * (1.f / 2048.f);
                        o[e] = fmaxf(o[e], o[e] * a.slope);
                        if (!INTERIOR) o[e] = in_image[u] ? o[e] : 0.f;
                    }
                    c16_split4(o[0], o[1], o[2], o[3], h, m);
                    char* dst = sm + C16_IN0 + dst_lane + i * 2048;
                    *reinterpret_cast<c16_f16x4*>(dst) = h;
                    *reinterpret_cast<c16_f16x4*>(dst + 2 * C16_CH) = m;
                }
            }
            };
            if (interior) stage0(std::true_type{}); else stage0(std::false_type{});
        } else {
        // ---- split the staging image: item = (patch pixel, 4-channel group)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int it = t + 512 * j;
            if (it < C16_NPIX * 4 && !(ABL & 2)) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sm + C16_S0 + it * 16);
                const int rec = it >> 2, g = it & 3;
                c16_f16x4 h, m;
                c16_split4(v[0], v[1], v[2], v[3], h, m);
                char* dst = sm + C16_IN0 + (g >> 1) * C16_CH + rec * 16 + (g & 1) * 8;
                *reinterpret_cast<c16_f16x4*>(dst) = h;
                *reinterpret_cast<c16_f16x4*>(dst + 2 * C16_CH) = m;
            }
        }
        }
        C16_BAR();
        if (tile + 1 < tile1) {                      // the next patch, under the two layers
            patch_prepare(tile + 1);
            patch_issue();
        }
        // ---- layer 1: intermediate pixels [36, 692) -> bias, leaky-relu, zero outside the image, split, into LDS
        auto layer1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            for (int mt = wave; mt < ((ABL & 4) ? 0 : C16_MT_A); mt += 8) {
                const int p0 = C16_P + mt * 16;
                f32x4 o = mtile(sm + C16_IN0, p0, A1, b1v);
                const int p = p0 + n16;
                bool in_image = true;
                if (!INTERIOR) {
                    const int pr = p / C16_P, pc = p - pr * C16_P;
                    in_image = (unsigned)(y0 - 2 + pr) < (unsigned)a.H && (unsigned)(x0 - 2 + pc) < (unsigned)a.W;
                }
                c16_f16x4 h, m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = fmaxf(o[e], o[e] * a.slope);
                    if (!INTERIOR) o[e] = in_image ? o[e] : 0.f;
                }
                c16_split4(o[0], o[1], o[2], o[3], h, m);
                char* dst = sm + C16_MID0 + (kq >> 1) * C16_CH + p * 16 + (kq & 1) * 8;
                *reinterpret_cast<c16_f16x4*>(dst) = h;
                *reinterpret_cast<c16_f16x4*>(dst + 2 * C16_CH) = m;
            }
        };
        if (interior) layer1(std::true_type{}); else layer1(std::false_type{});
        C16_BAR();
        // ---- layer 2: output rows 2 .. 17 of the patch -> bias, leaky-relu, 16 bytes (4 couts) per lane
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (size_t)n * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
        for (int mt = wave; mt < ((ABL & 8) ? 0 : C16_MT_B); mt += 8) {
            const int p0 = 2 * C16_P + mt * 16;
            f32x4 o = mtile(sm + C16_MID0, p0, A2, b2v);
            const int p = p0 + n16, pr = p / C16_P, pc = p - pr * C16_P;
            const int yy = y0 - 2 + pr, xx = x0 - 2 + pc;
            const bool ok = pc >= 2 && pc < 34 && yy < a.H && xx < a.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], o[e] * a.slope);
            const unsigned vo = ok ? (unsigned)(((yy * a.W + xx) * a.y_cs + 4 * kq) * 4) : C16_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, (int)vo, 0, 0);
        }
    }
#undef C16_BAR
}

// ---------------------------------------------------------------- weight split + packing
// packed[layer][operand m 14][lane 64][8 fp16]; lane = (cout i = lane & 15, K quarter kq = lane >> 4), see the header comment.
__global__ void conv3x3_c16pair_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                            unsigned short* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (layer, m, lane, e)
    if (idx >= 2 * 14 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, m = (idx >> 9) % 14, layer = idx / (14 * 512);
    const int i = lane & 15, kq = lane >> 4;
    const float* w = layer ? w2 : w1;                           // HWIO (3,3,16,16)
    int tap, ch;
    bool low;                                                   // um' (else uh)
    if (m < 5) { tap = 2 * m + (kq >> 1); ch = 8 * (kq & 1) + e; low = false; }
    else { tap = m - 5; ch = 8 * (kq & 1) + e; low = kq >= 2; }
    float u = 0.f;
    if (tap < 9) u = w[(tap * 16 + ch) * 16 + i];
    const _Float16 h = (_Float16)u;
    const _Float16 lo = (_Float16)((u - (float)h) * 2048.f);
    packed[idx] = __builtin_bit_cast(unsigned short, low ? lo : h);
}

// The stride-2 convolution's two operands behind the 28 of the layers: packed[28 + (0: uh, 1: um')][lane 64][8 fp16], K as in the
// kernel: quarter kq < 3 = tap row kq, floats j = 3 dx + c = 0 .. 7; quarter 3 = (tap row e, j = 8) for e < 3, zeros behind.
__global__ void conv3x3_c16pair_pack0_kernel(const float* __restrict__ w0, unsigned short* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (low, lane, e)
    if (idx >= 2 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, low = idx >> 9;
    const int i = lane & 15, kq = lane >> 4;
    int dy = -1, j = 0;
    if (kq < 3) { dy = kq; j = e; }
    else if (e < 3) { dy = e; j = 8; }
    float u = 0.f;
    if (dy >= 0) u = w0[((dy * 3 + j / 3) * 3 + j % 3) * 16 + i];   // HWIO (3,3,3,16)
    const _Float16 h = (_Float16)u;
    const _Float16 lo = (_Float16)((u - (float)h) * 2048.f);
    packed[28 * 64 * 8 + idx] = __builtin_bit_cast(unsigned short, low ? lo : h);
}

extern "C" size_t pwc_conv3x3_c16pair_packed_floats(void) { return 2 * 14 * 64 * 8 / 2; }
extern "C" size_t pwc_conv3x3_c3c16pair_packed_floats(void) { return (2 * 14 + 2) * 64 * 8 / 2; }

extern "C" int pwc_conv3x3_c3c16pair_pack_f32(const float* w0_hwio, const float* w1_hwio, const float* w2_hwio, float* packed,
                                              pwc_stream_t stream) {
    if (!w0_hwio || !w1_hwio || !w2_hwio || !packed) return PWC_EINVAL;
    hipLaunchKernelGGL(conv3x3_c16pair_pack_kernel, dim3((2 * 14 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w1_hwio, w2_hwio, reinterpret_cast<unsigned short*>(packed));
    hipLaunchKernelGGL(conv3x3_c16pair_pack0_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, w0_hwio,
                       reinterpret_cast<unsigned short*>(packed));
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_c16pair_pack_f32(const float* w1_hwio, const float* w2_hwio, float* packed, pwc_stream_t stream) {
    if (!w1_hwio || !w2_hwio || !packed) return PWC_EINVAL;
    hipLaunchKernelGGL(conv3x3_c16pair_pack_kernel, dim3((2 * 14 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w1_hwio, w2_hwio, reinterpret_cast<unsigned short*>(packed));
    return pwc_launch_status();
}

// 1 where the fused launch is the faster way to run the pair (at least one tile per CU).
extern "C" int pwc_conv3x3_c16pair_supported(int N, int H, int W) {
    if (N <= 0 || H < 16 || W < 32) return 0;
    return (long)N * ((H + 15) / 16) * ((W + 31) / 32) >= 256 ? 1 : 0;
}

// 1 where the three-layer launch is supported and the faster way: raw rows that are whole 16-byte chunks, a level-1 image the
// pair kernel takes.
extern "C" int pwc_conv3x3_c3c16pair_supported(int N, int H0, int W0) {
    if (N <= 0 || H0 < 2 || W0 < 4 || (W0 & 3)) return 0;
    return pwc_conv3x3_c16pair_supported(N, (H0 + 1) / 2, W0 / 2);
}

static int c16pair_grid(int ntiles) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return ntiles < cus ? ntiles : cus;
}

// images: N_a at x_a and N_b at x_b (x_b may be null with N_b = 0), NHWC with exactly 3 channels (channel stride 3), H0 x W0,
// W0 % 4 == 0; y: (N_a + N_b) x ceil(H0 / 2) x W0 / 2 x 16 at channel stride y_cs.
template <int ABL = 0>
static int c3c16pair_run(const float* x_a, int N_a, const float* x_b, int N_b, const float* packed, const float* bias0,
                         const float* bias1, const float* bias2, float* y, int y_cs, int H0, int W0, float slope,
                         pwc_stream_t stream) {
    if (!x_a || !packed || !bias0 || !bias1 || !bias2 || !y || (N_b > 0 && !x_b)) return PWC_EINVAL;
    if (N_a <= 0 || N_b < 0 || H0 <= 0 || W0 <= 0 || y_cs < 16) return PWC_EINVAL;
    if (W0 & 3) return PWC_EUNSUPPORTED;
    if ((y_cs & 3) || !pwc_aligned16(x_a) || !pwc_aligned16(x_b) || !pwc_aligned16(y) || !pwc_aligned16(packed) ||
        !pwc_aligned16(bias0) || !pwc_aligned16(bias1) || !pwc_aligned16(bias2))
        return PWC_EALIGN;
    const int H = (H0 + 1) / 2, W = W0 / 2;
    if ((long)H0 * W0 * 12 >= (long)C16_OOB || (long)H * W * y_cs * 4 >= (long)C16_OOB) return PWC_ERANGE;
    C16Args a;
    a.x = x_a; a.x_b = x_b; a.N_a = N_a; a.wp = packed; a.b0 = bias0; a.b1 = bias1; a.b2 = bias2; a.y = y; a.x_cs = 3; a.y_cs = y_cs;
    a.N = N_a + N_b; a.H = H; a.W = W; a.H0 = H0; a.W0 = W0; a.pad_t = H0 & 1; a.slope = slope;      // TF 'SAME', stride 2: odd sizes pad one row on top
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 15) / 16;
    const long nt = (long)a.N * a.tiles_x * a.tiles_y;
    if (nt >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nt;
    const int grid = c16pair_grid(a.ntiles);
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c16pair_kernel<ABL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS);
    hipLaunchKernelGGL((conv3x3_c16pair_kernel<ABL, true>), dim3((unsigned)grid), dim3(512), C16_LDS, (hipStream_t)stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_c3c16pair_f32(const float* x_a, int N_a, const float* x_b, int N_b, const float* packed,
                                         const float* bias0, const float* bias1, const float* bias2, float* y, int y_cs,
                                         int H0, int W0, float slope, pwc_stream_t stream) {
    return c3c16pair_run<0>(x_a, N_a, x_b, N_b, packed, bias0, bias1, bias2, y, y_cs, H0, W0, slope, stream);
}

template <int ABL = 0>
static int c16pair_run(const float* x, int x_cs, const float* packed, const float* bias1, const float* bias2,
                       float* y, int y_cs, int N, int H, int W, float slope, pwc_stream_t stream) {
    if (!x || !packed || !bias1 || !bias2 || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || x_cs < 16 || y_cs < 16) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed) || !pwc_aligned16(bias1) ||
        !pwc_aligned16(bias2))
        return PWC_EALIGN;
    if ((long)H * W * x_cs * 4 >= (long)C16_OOB || (long)H * W * y_cs * 4 >= (long)C16_OOB) return PWC_ERANGE;
    C16Args a;
    a.x = x; a.wp = packed; a.b1 = bias1; a.b2 = bias2; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.slope = slope;
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 15) / 16;
    const long nt = (long)N * a.tiles_x * a.tiles_y;
    if (nt >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nt;
    a.x_b = nullptr; a.b0 = nullptr; a.N_a = N; a.H0 = 0; a.W0 = 0; a.pad_t = 0;
    const int grid = c16pair_grid(a.ntiles);
    static PwcDevOnce attr_once;
    if (pwc_first_on_device(&attr_once))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c16pair_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS);
    hipLaunchKernelGGL((conv3x3_c16pair_kernel<ABL>), dim3((unsigned)grid), dim3(512), C16_LDS, (hipStream_t)stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_c16pair_f32(const float* x, int x_cs, const float* packed, const float* bias1, const float* bias2,
                                       float* y, int y_cs, int N, int H, int W, float slope, pwc_stream_t stream) {
    return c16pair_run<0>(x, x_cs, packed, bias1, bias2, y, y_cs, N, H, W, slope, stream);
}
