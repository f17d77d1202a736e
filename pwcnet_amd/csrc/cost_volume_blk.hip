// cost_volume_blk.hip -- warp + cost volume (+ the optional f0 concat copy) for the SMALL pyramid levels, one 4 x 4-pixel block
// per workgroup, correlation on the F16 matrix pipe (round 5).  Search range 4, C = 64 / 96 / 128 / 192.
//
// Replaces (reference model.py:105-112, modules.py:99-137,158-204,264), like pwc_cost_volume_coarse_f32:
//   f1w[n,y,x,:]             = bilinear_warp(f1, flow * flow_scale)          (fp32; flow == null: f1 as is -- pyramid level 0)
//   out[n,y,x,(v+4)*9+(h+4)] = lrelu( (1/C) * sum_c f0[n,y,x,c] * f1w[n,y+v,x+h,c] ),  f1w zero outside
//   f0_copy[n,y,x,0:C]       = f0[n,y,x,0:C]                                  (optional)
//
// Why another kernel.  At 7 x 16 ... 28 x 64 pixels per image a launch is a chain of round trips to memory, not bandwidth:
// cost_volume_coarse_kernel walks C / 32 channel stages one after the other (a trip each), cost_volume_h2.hip walks block rows with
// three fill steps per workgroup (a trip each): 12 / 22 / 20 us for the three coarsest levels of a batch of 8, i.e. for 3 % of the
// correlation's bytes.  Here a workgroup takes ONE 4 x 4 block of f0 and requests its 12 x 12-pixel window of f1 at once -- all
// channels, all four bilinear corners, in rounds of six (pixel, channel quad) items per lane -- so a launch is: flow -> corner
// table -> corner requests -> blend + split -> 27 C / 32 matrix instructions -> stage -> stores.  Where a launch has fewer blocks
// than the device has CUs (QR = 1) a workgroup takes ONE of the window's three block rows (4 x 12 pixels): three times the
// workgroups, a third of the chain each.  A window row gives a pixel of f0 the displacement rows v that fall into it (for a pixel
// row, every v belongs to exactly one window row): the three workgroups of a block write disjoint channel runs of the same
// 81-float records.  The window is re-gathered by each of the (up to) nine blocks that meet it; at these sizes the requests hit
// the L2 (f1 of a level: 0.7 - 5.5 MB) -- and that is what ends this form's range: from about 8192 pixels per launch on the
// row-walking kernel, which reads f1 three times, is faster.
//
// Measured (batch 8, graph replays over operand sets rotating through 300 MB, us per launch; profiles/r05_exp_blk_ab.txt):
//   7 x 16 x 192 (no warp)   coarse fp32 10.0   QR=3  8.7   QR=1  6.1
//   14 x 32 x 128            coarse fp32 19.3   QR=3  9.1   QR=1  9.9
//   28 x 64 x 96             row-walking 16.4   QR=3 19.5   QR=1 21.1     (batch 1: 15.8 / 8.6 / 6.8)
//
// Arithmetic, operand layouts, the K mapping of an instruction and the stage are those of cost_volume_h2.hip: two-term fp16 splits
// (pwc_split2), cross = AH x BM' + AM' x BH and hh = AH x BH per 32 channels in two fp32 accumulators, 1/C applied to the finished sum.
#pragma once
#include "cost_volume_mfma.hip"

struct CvbArgs {
    const float* f0;
    const float* f1;
    const float* flow;      // null: f1 is used as is
    float* out;
    float* f0_copy;         // null: no concat copy
    int f0_cs, f1_cs, flow_cs, out_cs, f0_copy_cs;
    int N, H, W;
    float flow_scale, slope, inv_c;
    int nbx, nby;           // 4 x 4 blocks per image row / column
    int pad_ok;             // channels 81..83 of every `out` record may be written (with zeros)
};

template <int CG, int QR>
struct CvbGeom {
    static constexpr int C = 16 * CG, NQ = C / 4, NP = CG / 2;
    static constexpr int RSB = 12 * 64 + 16;                 // bytes of a window row in a plane (+16: the four rows of a block
                                                             // start four banks apart: conflict-free 16-byte operand reads)
    static constexpr int ROWS = 4 * QR;                      // window rows of a workgroup (QR of the window's three block rows)
    static constexpr int PLANE = ROWS * RSB;                 // bytes of a 32-channel plane (h or m') of the ROWS x 12 window part
    static constexpr int IMG = 2 * NP * PLANE;               // h planes, then m' planes
    static constexpr int TAB = 12 * ROWS * 32;               // corner table: 4 offsets + 4 weights per window pixel
    static constexpr int STG = (16 * ROWS * 9 + 64) * 4;     // stage: [16 pixels][window row][9 h] + a dump slot per lane
    static constexpr int LDS = IMG + TAB + STG;
    static constexpr int NITEMS = 12 * ROWS * NQ;            // (window pixel, channel quad) items
    static constexpr int PER_LANE = (NITEMS + 255) / 256;
    static_assert(CG % 2 == 0 && LDS <= 160 * 1024, "C must be a multiple of 32 that fits the LDS");
};

template <int CG, bool WARP, bool PAD, int QR>
__global__ __launch_bounds__(256) void cost_volume_blk_kernel(const CvbArgs a) {
    using G = CvbGeom<CG, QR>;
    constexpr int ROWS = G::ROWS;
    constexpr int NQ = G::NQ, NP = G::NP, RSB = G::RSB, PLANE = G::PLANE;
    constexpr int R = WARP ? 6 : 9;                          // items of a round (96 / 36 registers of requests in flight)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const img = reinterpret_cast<char*>(smem);
    float* const tab = reinterpret_cast<float*>(img + G::IMG);
    float* const stg = tab + G::TAB / 4;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int blk = blockIdx.x;
    const int qbr0 = QR == 3 ? 0 : blk % 3;                  // first window block row: displacement rows 4 (qbr0 - 1) + window row - pixel row
    if (QR == 1) blk /= 3;
    const int bx = blk % a.nbx;
    blk /= a.nbx;
    const int by = blk % a.nby;
    const int n = blk / a.nby;
    const int y0 = 4 * by, x0 = 4 * bx;

    const size_t npx = (size_t)a.H * a.W;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f0 + (size_t)n * npx * a.f0_cs), 0, (int)(npx * a.f0_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f1 + (size_t)n * npx * a.f1_cs), 0, (int)(npx * a.f1_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.out + (size_t)n * npx * a.out_cs), 0, (int)(npx * a.out_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(WARP ? a.flow + (size_t)n * npx * a.flow_cs : a.f1), 0, WARP ? (int)(npx * a.flow_cs * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.f0_copy ? a.f0_copy + (size_t)n * npx * a.f0_copy_cs : a.out), 0,
        a.f0_copy ? (int)(npx * a.f0_copy_cs * 4) : 0, 0x00020000);

    // ---- the block's f0 operand: lane = (pixel m = lane & 15 -> row m >> 2, column m & 3; channel quad kq = lane >> 4), every
    // wave its own copy (requested first: it lands under the gather)
    const int mrow = (lane & 15) >> 2, mcol = lane & 3, kq = lane >> 4;
    const bool a_in = y0 + mrow < a.H && x0 + mcol < a.W;
    const unsigned a_pix = (unsigned)((y0 + mrow) * a.W + x0 + mcol);
    f32x4 A[CG];
    {
        const unsigned vo = a_in ? a_pix * (unsigned)(a.f0_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g) A[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)vo, g * 64, 0));
    }

    // ---- corner table of the 4 x 12 window row (lanes 0..47: one window pixel each); pixels outside the image: out-of-range offsets
    if (t < 12 * ROWS) {
        const int qr = (t * 2731) >> 15, qc = t - qr * 12;            // t / 12 for t < 144
        const int gy = y0 + 4 * (qbr0 - 1) + qr, gx = x0 - 4 + qc;
        const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        cvm_u32x4 off = {CVM_OOB, CVM_OOB, CVM_OOB, CVM_OOB};
        f32x4 w = {1.f, 0.f, 0.f, 0.f};
        const unsigned cs4 = (unsigned)a.f1_cs * 4u;
        if (WARP) {
            const unsigned vo = ok ? (unsigned)((gy * a.W + gx) * a.flow_cs) * 4u : CVM_OOB;
            const float f0v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 0, 0));
            const float f1v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, (int)vo, 4, 0));
            // bilinear_warp, modules.py:107-137: the product flow * scale is rounded first (model.py:109 is an op of its own),
            // weights from the un-clipped floors, the four corner indices clipped independently
            const float fx = pwc_mul_rounded(f0v, a.flow_scale), fy = pwc_mul_rounded(f1v, a.flow_scale);
            const float fx0 = floorf(fx), fy0 = floorf(fy);
            const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
            const float hl = (float)(a.H - 1), wl = (float)(a.W - 1);
            const int iy0 = (int)fminf(fmaxf((float)gy + fy0, 0.f), hl), iy1 = (int)fminf(fmaxf((float)gy + fy1, 0.f), hl);
            const int ix0 = (int)fminf(fmaxf((float)gx + fx0, 0.f), wl), ix1 = (int)fminf(fmaxf((float)gx + fx1, 0.f), wl);
            w = f32x4{(fy1 - fy) * (fx1 - fx), (fy1 - fy) * (fx - fx0), (fy - fy0) * (fx1 - fx), (fy - fy0) * (fx - fx0)};
            if (ok) off = cvm_u32x4{(unsigned)(iy0 * a.W + ix0) * cs4, (unsigned)(iy0 * a.W + ix1) * cs4,
                                    (unsigned)(iy1 * a.W + ix0) * cs4, (unsigned)(iy1 * a.W + ix1) * cs4};
        } else if (ok) {
            off[0] = (unsigned)(gy * a.W + gx) * cs4;
        }
        *reinterpret_cast<cvm_u32x4*>(tab + t * 8) = off;
        *reinterpret_cast<f32x4*>(tab + t * 8 + 4) = w;
    }
    cvm_barrier();

    // ---- the gather: item e = (window pixel e / NQ, channel quad e % NQ) -- the quads of a pixel in consecutive lanes, so one
    // instruction asks for whole lines of a corner pixel -- in rounds of R items per lane: requests, then blend + split -> image
    for (int r0i = 0; r0i < G::PER_LANE; r0i += R) {
        f32x4 gv[R][WARP ? 4 : 1];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int e = (r0i + i) * 256 + t;
            const bool live = r0i + i < G::PER_LANE && e < G::NITEMS;
            const int qp = live ? e / NQ : 0, cq = e - qp * NQ;
            const cvm_u32x4 off = *reinterpret_cast<const cvm_u32x4*>(tab + qp * 8);
#pragma unroll
            for (int c = 0; c < (WARP ? 4 : 1); ++c) {
                const unsigned vo = live ? off[c] + (unsigned)(cq * 16) : CVM_OOB;      // out-of-range + chan stays out of range
                gv[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, (int)vo, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int e = (r0i + i) * 256 + t;
            const bool live = r0i + i < G::PER_LANE && e < G::NITEMS;
            const int qp = live ? e / NQ : 0, cq = e - qp * NQ;
            f32x4 v = gv[i][0];
            if (WARP) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(tab + qp * 8 + 4);
                // modules.py:132-135: c00*x00 + c01*x01 + c10*x10 + c11*x11, summed left to right
                v = w[0] * gv[i][0];
                v = __builtin_elementwise_fma(f32x4{w[1], w[1], w[1], w[1]}, gv[i][1], v);
                v = __builtin_elementwise_fma(f32x4{w[2], w[2], w[2], w[2]}, gv[i][2], v);
                v = __builtin_elementwise_fma(f32x4{w[3], w[3], w[3], w[3]}, gv[i][3], v);
            }
            pwc_f16x4 h, m;
            pwc_split4(v, h, m);
            const int qr = (qp * 2731) >> 15, qc = qp - qr * 12;
            const int g = cq >> 2, kqi = cq & 3;
            char* dst = img + (g >> 1) * PLANE + qr * RSB + qc * 64 + kqi * 16 + (g & 1) * 8;
            if (live) {
                *reinterpret_cast<pwc_f16x4*>(dst) = h;
                *reinterpret_cast<pwc_f16x4*>(dst + NP * PLANE) = m;
            }
        }
    }

    // ---- the block's own operand: concat copy (wave 0 of the middle window row), split
    if (wave == 0 && (QR == 3 || qbr0 == 1)) {
        const unsigned vo = a_in ? a_pix * (unsigned)(a.f0_copy_cs * 4) + (unsigned)(kq * 16) : CVM_OOB;
#pragma unroll
        for (int g = 0; g < CG; ++g)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cvm_u32x4, A[g]), rc, (int)vo, g * 64, 0);
    }
    pwc_f16x8 AH[NP], AM[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        pwc_f16x4 h0, m0, h1, m1;
        pwc_split4(A[2 * j], h0, m0);
        pwc_split4(A[2 * j + 1], h1, m1);
        AH[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        AM[j] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    cvm_barrier();

    // ---- tiles: the window blocks (row qr4, column qbc) over the waves.  D fragment: lane holds P pixels (row kq, column r = 0..3)
    // x Q pixel (row mrow, column mcol); entry (v, h) = (4 (qbr0 + qr4 - 1) + mrow - kq, 4 (qbc - 1) + mcol - r) goes to
    // stage[P pixel][window row 4 qr4 + mrow][h + 4], entries beyond h = +-4 to the lane's dump slot (v is checked at the copy-out)
    for (int qb = wave; qb < 3 * QR; qb += 4) {
        const int qr4 = qb / 3, qbc = qb - 3 * qr4;
        const char* base = img + (4 * qr4 + mrow) * RSB + (4 * qbc + mcol) * 64 + kq * 16;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 hh = zero, xx = zero;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const pwc_f16x8 bh = *reinterpret_cast<const pwc_f16x8*>(base + j * PLANE);
            const pwc_f16x8 bm = *reinterpret_cast<const pwc_f16x8*>(base + (NP + j) * PLANE);
            xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH[j], bm, xx, 0, 0, 0);
            hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH[j], bh, hh, 0, 0, 0);
            xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(AM[j], bh, xx, 0, 0, 0);
        }
        const f32x4 s = __builtin_elementwise_fma(xx, f32x4{1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f, 1.f / 2048.f}, hh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hx = 4 * (qbc - 1) + mcol - r;
            const bool ok = hx >= -4 && hx <= 4;
            stg[ok ? ((kq * 4 + r) * ROWS + 4 * qr4 + mrow) * 9 + hx + 4 : 16 * ROWS * 9 + lane] = s[r];
        }
    }
    cvm_barrier();

    // ---- copy-out: entry e = (pixel p, window row mr, h): channel (v + 4) * 9 + h + 4 with v = 4 (qbr0 - 1) + mr - (p >> 2), where
    // |v| <= 4 -- for a pixel a contiguous run of channels; mean = sum * (1/C) (reduce_mean, modules.py:181), leaky-relu
    constexpr int NE = 16 * ROWS * 9;
#pragma unroll
    for (int i = 0; i < (NE + 255) / 256; ++i) {
        const int e = i * 256 + t;
        const int pm = (e * 7282) >> 16, h = e - pm * 9;             // e / 9 for e < 1728
        const int p = pm / ROWS, mr = pm - p * ROWS;
        const int v = 4 * (qbr0 - 1) + mr - (p >> 2);
        const int py = y0 + (p >> 2), px = x0 + (p & 3);
        const bool ok = e < NE && v >= -4 && v <= 4 && py < a.H && px < a.W;
        const float y = pwc_lrelu(stg[e < NE ? e : 0] * a.inv_c, a.slope);
        const unsigned vo = ok ? (unsigned)(((py * a.W + px) * a.out_cs + (v + 4) * 9 + h) * 4) : CVM_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ro, (int)vo, 0, 0);
    }
    if constexpr (PAD) {
        if ((QR == 3 || qbr0 == 2) && t < 48) {                                    // padding channels 81..83
            const int p = (t * 171) >> 9, c = t - 3 * p;             // t / 3 for t < 48
            const int py = y0 + (p >> 2), px = x0 + (p & 3);
            const unsigned vo = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.out_cs + 81 + c) * 4) : CVM_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(0u, ro, (int)vo, 0, 0);
        }
    }
}

static bool cvb_eligible(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow, int flow_cs,
                         const float* out, int out_cs, const float* f0_copy, int f0_copy_cs, int H, int W, int C, int R) {
    if (R != 4 || !(C == 64 || C == 96 || C == 128 || C == 192)) return false;
    if ((f0_cs & 3) || (f1_cs & 3) || (out_cs & 3) || !pwc_aligned16(f0) || !pwc_aligned16(f1) || !pwc_aligned16(out)) return false;
    if (f0_copy && ((f0_copy_cs & 3) || !pwc_aligned16(f0_copy))) return false;
    if (flow && (reinterpret_cast<uintptr_t>(flow) & 3u)) return false;
    const long px = (long)H * W;
    if (px * f0_cs * 4 >= (1L << 31) || px * f1_cs * 4 >= (1L << 31) || px * out_cs * 4 >= (1L << 31)) return false;
    if (f0_copy && px * f0_copy_cs * 4 >= (1L << 31)) return false;
    if (flow && px * flow_cs * 4 >= (1L << 31)) return false;
    return true;
}

template <int CG, bool WARP, bool PAD, int QR>
static int cvb_launch_q(const CvbArgs& a, hipStream_t s) {
    using G = CvbGeom<CG, QR>;
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_blk_kernel<CG, WARP, PAD, QR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    }
    const long items = (long)a.N * a.nbx * a.nby * (QR == 1 ? 3 : 1);
    if (items >= (1L << 31)) return PWC_ERANGE;
    hipLaunchKernelGGL((cost_volume_blk_kernel<CG, WARP, PAD, QR>), dim3((unsigned)items), dim3(256), G::LDS, s, a);
    return pwc_launch_status();
}

#ifdef PWC_HARNESS
static int cvb_rows_override = 0;   // libpwc_hip_harness.so only (pwc_debug_cost_volume_blk_rows, scripts/exp_blk_ab.py)
#else
constexpr int cvb_rows_override = 0;
#endif

template <int CG, bool WARP, bool PAD>
static int cvb_launch_t(const CvbArgs& a, hipStream_t s) {
    // fewer blocks than CUs: a workgroup per (block, window block row) -- three times the workgroups, a third of the chain each
    // (7 x 16 level of a batch of 8: 6.1 us against 8.7); otherwise the whole window per workgroup (14 x 32 level: 9.1 against 9.9)
    const int qr = cvb_rows_override ? cvb_rows_override : ((long)a.N * a.nbx * a.nby < 256 ? 1 : 3);
    return qr == 3 ? cvb_launch_q<CG, WARP, PAD, 3>(a, s) : cvb_launch_q<CG, WARP, PAD, 1>(a, s);
}

static int cvb_launch(const float* f0, int f0_cs, const float* f1, int f1_cs, const float* flow, int flow_cs,
                      float flow_scale, float* out, int out_cs, int pad_ok, float* f0_copy, int f0_copy_cs, int N, int H,
                      int W, int C, float slope, hipStream_t s) {
    CvbArgs a;
    a.f0 = f0; a.f1 = f1; a.flow = flow; a.out = out; a.f0_copy = f0_copy;
    a.f0_cs = f0_cs; a.f1_cs = f1_cs; a.flow_cs = flow_cs; a.out_cs = out_cs; a.f0_copy_cs = f0_copy_cs;
    a.N = N; a.H = H; a.W = W; a.flow_scale = flow_scale; a.slope = slope;
    a.inv_c = 1.0f / (float)C;               // reduce_mean: x * (1/C), within 1 ulp of x / C
    a.nbx = (W + 3) / 4; a.nby = (H + 3) / 4; a.pad_ok = pad_ok;
#define CVB_CASE(CGV)                                                                          \
    case CGV * 16:                                                                             \
        return flow ? (pad_ok ? cvb_launch_t<CGV, true, true>(a, s) : cvb_launch_t<CGV, true, false>(a, s))         \
                    : (pad_ok ? cvb_launch_t<CGV, false, true>(a, s) : cvb_launch_t<CGV, false, false>(a, s));
    switch (C) {
        CVB_CASE(4)
        CVB_CASE(6)
        CVB_CASE(8)
        CVB_CASE(12)
        default: return PWC_EUNSUPPORTED;
    }
#undef CVB_CASE
}
