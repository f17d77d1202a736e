// conv3x3_mfma.hip -- 3x3 convolution (TF 'SAME', stride 1/2, dilation 1..16) as an
// fp32-MFMA implicit GEMM for gfx950 (MI355X).
//
// Replaces tf.layers.Conv2D(Cout,(3,3),(s,s),'same',dilation_rate=d) + tf.nn.leaky_relu
// of the reference (modules.py:62-67 extractor, :267-268 flow estimator, :306-323
// context network).
//
// GEMM view (per launch):  D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],
//   pixel = flattened (n, oy, ox) output position, k = (tap, physical input channel).
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, 32 cyc/issue, 157 TFLOP/s chip peak) with
//   A operand = weights   (row i = cout  = lane & 15, k-slot = lane >> 4)
//   B operand = activations (col j = pixel = lane & 15, k-slot = lane >> 4)
//   D: lane holds couts 4*(lane>>4)+{0..3} of pixel (lane & 15)  -> one float4 NHWC store.
// k is consumed in groups of 16: lane k-slot q reads ONE ds_read_b128 = channels
// 4q..4q+3 and feeds element s of it to MFMA step s, i.e. step s contracts
// k = {s, 4+s, 8+s, 12+s}; both operands use the same map, so the sum is complete.
//
// LDS tiles are [k-group][row][16 floats] (64-byte rows, no padding) with the 16-byte
// chunk index XOR-swizzled by {0,3,2,1}[(row>>2)&3]: conflict-free for the four
// 16-lane groups a ds_read_b128 is serviced in.  The weight image is pre-swizzled at
// pack time (pwc_conv3x3_pack_f32), so its LDS copy is a linear memcpy.
//
// Pipeline: register-staged prefetch of stage s+1 (global -> VGPR) is issued before the
// MFMAs of stage s; the VGPRs are written to the other LDS buffer after them; one
// barrier per stage.
#include "pwc_common.h"

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int H, W, Ho, Wo;
    int Cin_phys, Cout, Cout_pad;
    int stride, dil, pad_t, pad_l;
    int apply_act;
    float slope;
    int M;      // N*Ho*Wo output pixels
    int y_vec4; // y pointer/stride allow float4 stores
    int m_begin, m_end;   // pixel range of this launch (tail launches use a smaller tile)
    int taps_per_split;   // 9 (no split), 3 or 1: blockIdx.z owns taps [z*tps, (z+1)*tps)
    float* ws;            // split-K: raw partial sums ws[z][pixel][Cout_pad] (no bias/act)
    int xcd_remap;        // consecutive pixel tiles on the same XCD (shared halo rows hit its L2)
};

__device__ __forceinline__ int swz4(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// ABL: ablation switch for scripts/exp_conv_ablate.hip only (0 in every shipped
// instantiation): 1 = no global loads in the k loop, 2 = no LDS store / barrier,
// 3 = both (MFMA + ds_read only).
template <int WM, int WN, int WGM, int WGN, int KC, int ABL = 0>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const ConvArgs a) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int BM = 16 * WM * WGM;
    constexpr int BN = 16 * WN * WGN;
    constexpr int KG = KC / 16;       // 16-wide k-groups per stage
    constexpr int CPP = KC / 4;       // float4 chunks per pixel per stage
    constexpr int A_F4 = BM * CPP;
    constexpr int B_F4 = BN * CPP;
    constexpr int A_LD = (A_F4 + 255) / 256;
    constexpr int B_LD = (B_F4 + 255) / 256;
    constexpr int PIX_STEP = 256 / CPP;   // pixel rows covered by one pass of 256 threads
    constexpr int STAGE = (BM + BN) * KC; // floats per LDS buffer

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = a.m_begin + (a.xcd_remap ? pwc_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int tap0 = blockIdx.z * a.taps_per_split;

    // ---- per-thread A-load bookkeeping (fixed over the k loop)
    const int a_ch = t % CPP;          // 16-byte chunk within the pixel's KC channels
    const int a_row0 = t / CPP;
    const float* a_base[A_LD];
    int a_iy0[A_LD], a_ix0[A_LD], a_lds[A_LD];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int row = a_row0 + i * PIX_STEP;
        const int m = m0 + row;
        const bool ok = (row < BM) && (m < a.m_end);
        const int mm = ok ? m : 0;
        const int n_img = mm / HoWo;
        const int rem = mm - n_img * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        a_base[i] = a.x + (size_t)n_img * a.H * a.W * a.x_cs + a_ch * 4;
        // an invalid row gets an iy0 that fails every bounds test
        a_iy0[i] = ok ? oy * a.stride - a.pad_t : -(1 << 28);
        a_ix0[i] = ox * a.stride - a.pad_l;
        const int g = a_ch >> 2, j = a_ch & 3;
        a_lds[i] = ((g * BM + row) * 16) + ((j ^ swz4(row)) << 2);
    }
    // ---- per-thread B-load bookkeeping
    int b_src[B_LD], b_lds[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int q = t + i * 256;
        const int j = q & 3;
        const int rr = q >> 2;
        const int row = rr % BN, g = rr / BN;
        b_ok[i] = (q < B_F4) && (n0 + row < a.Cout_pad);
        b_src[i] = (g * a.Cout_pad + n0 + row) * 16 + j * 4;
        b_lds[i] = (g * BN + row) * 16 + j * 4;
    }

    const int CC = a.Cin_phys / KC; // stages per tap
    const int S = a.taps_per_split * CC;
    const int nc16 = a.Cin_phys >> 4;

    f32x4 ra[A_LD], rb[B_LD];
    auto load_stage = [&](int tap, int cc) {
        const int ty = tap / 3, tx = tap - ty * 3;
        const int dy = ty * a.dil, dx = tx * a.dil;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(a_base[i] + (size_t)(iy * a.W + ix) * a.x_cs + cc * KC);
            ra[i] = v;
        }
        const float* wsrc = a.wp + (size_t)(tap * nc16 + cc * KG) * a.Cout_pad * 16;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(wsrc + b_src[i]);
            rb[i] = v;
        }
    };
    auto store_stage = [&](int buf) {
        float* Ab = smem + buf * STAGE;
        float* Bb = Ab + BM * KC;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (A_F4 % 256 == 0 || t + i * 256 < A_F4) *reinterpret_cast<f32x4*>(Ab + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            if (B_F4 % 256 == 0 || t + i * 256 < B_F4) *reinterpret_cast<f32x4*>(Bb + b_lds[i]) = rb[i];
    };

    f32x4 acc[WN][WM];
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int m = 0; m < WM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;            // fragment row (cout for W, pixel for X)
    const int fq = lane >> 4;            // k-slot
    const int f_off = fr * 16 + ((fq ^ swz4(fr)) << 2);

    int tap = tap0, cc = 0;
    load_stage(tap0, 0);
    store_stage(0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        int ntap = tap, ncc = cc + 1;
        if (ncc == CC) { ncc = 0; ntap = tap + 1; }
        const bool more = (s + 1 < S);
        if (more && !(ABL & 1)) load_stage(ntap, ncc);

        const float* Ab = smem + buf * STAGE;
        const float* Bb = Ab + BM * KC;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            f32x4 wf[WN], xf[WM];
#pragma unroll
            for (int n = 0; n < WN; ++n)
                wf[n] = *reinterpret_cast<const f32x4*>(Bb + (g * BN + (wn * WN + n) * 16) * 16 + f_off);
#pragma unroll
            for (int m = 0; m < WM; ++m)
                xf[m] = *reinterpret_cast<const f32x4*>(Ab + (g * BM + (wm * WM + m) * 16) * 16 + f_off);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < WN; ++n)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[n][m], 0, 0, 0);
        }
        if (!(ABL & 2)) {
            if (more) store_stage(buf ^ 1);
            __syncthreads();
        }
        tap = ntap;
        cc = ncc;
    }

    if (a.ws) {
        // ---- split-K epilogue: raw partial sums to the workspace slab of this tap group
        float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout_pad;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = n0 + (wn * WN + n) * 16 + fq * 4;
            if (co >= a.Cout_pad) continue;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int pix = m0 + (wm * WM + m) * 16 + fr;
                if (pix < a.m_end) *reinterpret_cast<f32x4*>(slab + (size_t)pix * a.Cout_pad + co) = acc[n][m];
            }
        }
        return;
    }
    // ---- epilogue: bias + leaky-relu, NHWC float4 stores (4 consecutive couts per lane)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int co = n0 + (wn * WN + n) * 16 + fq * 4;
        if (co >= a.Cout) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int pix = m0 + (wm * WM + m) * 16 + fr;
            if (pix >= a.m_end) continue;
            f32x4 v = acc[n][m] + b4;
            if (a.apply_act) {
                v[0] = pwc_lrelu(v[0], a.slope);
                v[1] = pwc_lrelu(v[1], a.slope);
                v[2] = pwc_lrelu(v[2], a.slope);
                v[3] = pwc_lrelu(v[3], a.slope);
            }
            float* dst = a.y + (size_t)pix * a.y_cs + co;
            if (a.y_vec4) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
    }
}

// ---------------------------------------------------------------- LDS-DMA variant
// Same tiling, LDS image and MFMA loop as conv3x3_mfma_kernel, but both operand tiles go
// HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass): one
// wave instruction fills one 16-row x 64-byte block (1 KiB, lane-linear).  The XOR swizzle
// of the activation rows is applied on the SOURCE side (lane l fetches chunk
// (l&3) ^ swz(row) of its pixel, a permutation inside the pixel's 64-byte segment); the
// weight image is pre-swizzled by the packer.  Out-of-image taps (zero padding) and
// rows past the end read a 16-byte zero page instead.
__device__ float pwc_zero_page[4];

template <int WM, int WN, int WGM, int WGN, int KC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_mfma_glds_kernel(const ConvArgs a) {
    constexpr int NWV = WGM * WGN;               // waves per workgroup (4 or 8)
    constexpr int BM = 16 * WM * WGM;
    constexpr int BN = 16 * WN * WGN;
    constexpr int KG = KC / 16;
    constexpr int NBA = (BM / 16) * KG;          // 1-KiB blocks of the activation tile
    constexpr int NBB = (BN / 16) * KG;          // ... of the weight tile
    constexpr int A_PW = (NBA + NWV - 1) / NWV;  // blocks per wave
    constexpr int B_PW = (NBB + NWV - 1) / NWV;
    constexpr int STAGE = (BM + BN) * KC;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = a.m_begin + (a.xcd_remap ? pwc_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x) * BM;
    const int n0 = blockIdx.y * BN;
    const int tap0 = blockIdx.z * a.taps_per_split;
    const int lrow = lane >> 2, lslot = lane & 3;

    // ---- activation blocks owned by this wave: block b = wave + 4*i -> (kgroup g, 16-row group rb)
    const float* a_base[A_PW];
    int a_iy0[A_PW], a_ix0[A_PW];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int b = wave + NWV * i;
        const int g = b / (BM / 16), rb = b % (BM / 16);
        const int row = rb * 16 + lrow;
        const int m = m0 + row;
        const bool ok = (b < NBA) && (m < a.m_end);
        const int mm = ok ? m : 0;
        const int n_img = mm / HoWo;
        const int rem = mm - n_img * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        const int j = lslot ^ swz4(row);
        a_base[i] = a.x + (size_t)n_img * a.H * a.W * a.x_cs + g * 16 + j * 4;
        a_iy0[i] = ok ? oy * a.stride - a.pad_t : -(1 << 28);
        a_ix0[i] = ox * a.stride - a.pad_l;
    }
    int b_src[B_PW];
    bool b_ok[B_PW];
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        const int b = wave + NWV * i;
        const int g = b / (BN / 16), rb = b % (BN / 16);
        const int row = rb * 16 + lrow;
        b_ok[i] = (b < NBB) && (n0 + row < a.Cout_pad);
        b_src[i] = (g * a.Cout_pad + n0 + row) * 16 + lslot * 4;
    }

    const int CC = a.Cin_phys / KC;
    const int S = a.taps_per_split * CC;
    const int nc16 = a.Cin_phys >> 4;
    const float* zero = pwc_zero_page;

    auto issue_stage = [&](int tap, int cc, int buf) {
        const int ty = tap / 3, tx = tap - ty * 3;
        const int dy = ty * a.dil, dx = tx * a.dil;
        float* Ab = smem + buf * STAGE;
        float* Bb = Ab + BM * KC;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const int b = wave + NWV * i;
            if (NBA % NWV != 0 && b >= NBA) break;
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
            const float* src = ok ? a_base[i] + (size_t)(iy * a.W + ix) * a.x_cs + cc * KC : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ab + b * 256), 16, 0, 0);
        }
        const float* wsrc = a.wp + (size_t)(tap * nc16 + cc * KG) * a.Cout_pad * 16;
#pragma unroll
        for (int i = 0; i < B_PW; ++i) {
            const int b = wave + NWV * i;
            if (NBB % NWV != 0 && b >= NBB) break;
            const float* src = b_ok[i] ? wsrc + b_src[i] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bb + b * 256), 16, 0, 0);
        }
    };

    f32x4 acc[WN][WM];
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int m = 0; m < WM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int f_off = fr * 16 + ((fq ^ swz4(fr)) << 2);

    int tap = tap0, cc = 0;
    issue_stage(tap0, 0, 0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        int ntap = tap, ncc = cc + 1;
        if (ncc == CC) { ncc = 0; ntap = tap + 1; }
        if (s + 1 < S) issue_stage(ntap, ncc, buf ^ 1);

        const float* Ab = smem + buf * STAGE;
        const float* Bb = Ab + BM * KC;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            f32x4 wf[WN], xf[WM];
#pragma unroll
            for (int n = 0; n < WN; ++n)
                wf[n] = *reinterpret_cast<const f32x4*>(Bb + (g * BN + (wn * WN + n) * 16) * 16 + f_off);
#pragma unroll
            for (int m = 0; m < WM; ++m)
                xf[m] = *reinterpret_cast<const f32x4*>(Ab + (g * BM + (wm * WM + m) * 16) * 16 + f_off);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < WN; ++n)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[n][m], 0, 0, 0);
        }
        __syncthreads();   // drains this stage's LDS-DMA (vmcnt) and orders the buffer swap
        tap = ntap;
        cc = ncc;
    }

    if (a.ws) {
        float* slab = a.ws + (size_t)blockIdx.z * a.M * a.Cout_pad;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int co = n0 + (wn * WN + n) * 16 + fq * 4;
            if (co >= a.Cout_pad) continue;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int pix = m0 + (wm * WM + m) * 16 + fr;
                if (pix < a.m_end) *reinterpret_cast<f32x4*>(slab + (size_t)pix * a.Cout_pad + co) = acc[n][m];
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int co = n0 + (wn * WN + n) * 16 + fq * 4;
        if (co >= a.Cout) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int pix = m0 + (wm * WM + m) * 16 + fr;
            if (pix >= a.m_end) continue;
            f32x4 v = acc[n][m] + b4;
            if (a.apply_act) {
                v[0] = pwc_lrelu(v[0], a.slope);
                v[1] = pwc_lrelu(v[1], a.slope);
                v[2] = pwc_lrelu(v[2], a.slope);
                v[3] = pwc_lrelu(v[3], a.slope);
            }
            float* dst = a.y + (size_t)pix * a.y_cs + co;
            if (a.y_vec4) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
    }
}

// ---------------------------------------------------------------- small-Cin "halo" variant
// Stride-1, dilation-1 convolutions with few channels (16 -> 16, 32 -> 32: the first
// full-resolution extractor layers, modules.py:64-67 at l = 0, 1) are HBM-bound: K = 9*Cin is
// so short that the generic kernel spends its time re-loading the same pixels once per tap
// and synchronising 9 tiny stages.  Here
//   * the packed weights of ALL 9 taps stay resident in LDS for the life of the workgroup;
//   * a workgroup owns a TH x 32 pixel tile and DMA-loads its (TH+2) x 34 input patch ONCE
//     (global_load_lds, zero page outside the image); the 9 taps are row/column shifts of
//     the LDS read address;
//   * workgroups are persistent (the resident weights are loaded once per workgroup); NB = 2
//     keeps the next tile's patch in flight during the MFMAs, NB = 1 relies on 4 co-resident
//     workgroups per CU instead -- measured faster (TH = 4, NB = 1: 98-106 TFLOP/s).
// MFMA/LDS conventions are those of conv3x3_mfma_kernel (same packed weight image, same
// XOR swizzle of the 16-byte chunks, recomputed per tap for the shifted patch rows).
struct HaloArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int apply_act;
    float slope;
    int tiles_x, tiles_y;
    int y_vec4;
};

template <int CIN, int COUT, int TH, int NB>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const HaloArgs a) {
    constexpr int KG = CIN / 16;                  // 16-channel groups
    constexpr int NT = COUT / 16;                 // cout MFMA tiles
    constexpr int TW = 32;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PR = PH * PW;                   // patch pixels
    constexpr int PRP = (PR + 15) & ~15;          // padded to whole 16-row DMA blocks
    constexpr int NBP = (PRP / 16) * KG;          // DMA blocks per patch
    constexpr int BPW = (NBP + 3) / 4;            // per wave
    constexpr int WFL = 9 * CIN * COUT;           // resident weight floats
    constexpr int PFL = KG * PRP * 16;            // floats per patch buffer
    constexpr int MT = TH * TW / 16 / 4;          // pixel MFMA tiles per wave (16 px each)
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;
    float* pbuf = smem + WFL;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const float* zero = pwc_zero_page;

    // resident weights: linear copy of the packed image [tap][kg][cout][16]
    for (int i = t * 4; i < WFL; i += 256 * 4)
        *reinterpret_cast<f32x4*>(wl + i) = *reinterpret_cast<const f32x4*>(a.wp + i);

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = tiles_per_img * a.N;
    const int nwg = gridDim.x;
    const int first = pwc_xcd_remap(blockIdx.x, nwg);

    // tile-independent DMA map: block b = wave + 4*i -> kgroup g, patch rows 16*rb .. +15
    int blk[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int b = wave + 4 * i;
        int v = -1;
        if (b < NBP) {
            const int g = b / (PRP / 16), rb = b - g * (PRP / 16);
            const int pr = rb * 16 + (lane >> 2);
            if (pr < PR) {
                const int py = pr / PW, px = pr - py * PW;
                const int j = (lane & 3) ^ swz4(pr);          // source chunk for this LDS slot
                v = (py << 16) | (px << 8) | (g * 16 + j * 4);
            }
        }
        blk[i] = v;
    }
    auto issue_patch = [&](int lt, int buf) {
        const int n = lt / tiles_per_img;
        const int rem = lt - n * tiles_per_img;
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
        const int y0 = ty_ * TH - 1, x0 = tx_ * TW - 1;      // SAME padding: one pixel of halo
        const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;
        float* dst = pbuf + buf * PFL;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int b = wave + 4 * i;
            if (b < NBP) {
                const int v = blk[i];
                const int y = y0 + (v >> 16), x = x0 + ((v >> 8) & 0xFF);
                const bool ok = (v >= 0) && ((unsigned)y < (unsigned)a.H) && ((unsigned)x < (unsigned)a.W);
                const float* src = ok ? xn + (size_t)(y * a.W + x) * a.x_cs + (v & 0xFF) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + b * 256), 16, 0, 0);
            }
        }
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 b4[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b4[n] = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + fq * 4);

    int cur = 0;
    if (NB == 2 && first < ntiles) issue_patch(first, 0);
    for (int lt = first; lt < ntiles; lt += nwg) {
        if (NB == 1) {
            __syncthreads();                   // previous tile fully read before the patch is overwritten
            issue_patch(lt, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // patch `cur` landed; weights visible; previous tile read
        if (NB == 2 && lt + nwg < ntiles) issue_patch(lt + nwg, cur ^ 1);

        const float* pb = pbuf + cur * PFL;
        f32x4 acc[NT][MT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = zero4;

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                f32x4 wf[NT], xf[MT];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    wf[n] = *reinterpret_cast<const f32x4*>(wl + ((tap * KG + g) * COUT + n * 16 + fr) * 16 +
                                                            ((fq ^ swz4(fr)) << 2));
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int pt = wave * MT + m;               // pixel tile: row pt/2, columns 16*(pt&1)..
                    const int pr = ((pt >> 1) + ty) * PW + (pt & 1) * 16 + fr + tx;
                    xf[m] = *reinterpret_cast<const f32x4*>(pb + (g * PRP + pr) * 16 + ((fq ^ swz4(pr)) << 2));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][k], acc[n][m], 0, 0, 0);
            }
        }

        // epilogue: bias + leaky-relu, 16-byte NHWC stores
        const int n_img = lt / tiles_per_img;
        const int rem = lt - n_img * tiles_per_img;
        const int ty_ = rem / a.tiles_x, tx_ = rem - ty_ * a.tiles_x;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int pt = wave * MT + m;
            const int y = ty_ * TH + (pt >> 1), x = tx_ * TW + (pt & 1) * 16 + fr;
            if (y < a.H && x < a.W) {
                float* dst = a.y + ((size_t)(n_img * a.H + y) * a.W + x) * a.y_cs + fq * 4;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = acc[n][m] + b4[n];
                    if (a.apply_act) {
                        v[0] = pwc_lrelu(v[0], a.slope); v[1] = pwc_lrelu(v[1], a.slope);
                        v[2] = pwc_lrelu(v[2], a.slope); v[3] = pwc_lrelu(v[3], a.slope);
                    }
                    if (a.y_vec4) *reinterpret_cast<f32x4*>(dst + n * 16) = v;
                    else { dst[n * 16] = v[0]; dst[n * 16 + 1] = v[1]; dst[n * 16 + 2] = v[2]; dst[n * 16 + 3] = v[3]; }
                }
            }
        }
        if (NB == 2) cur ^= 1;
    }
}

template <int CIN, int COUT, int TH, int NB>
static int launch_halo(const HaloArgs& a0, hipStream_t s) {
    HaloArgs a = a0;
    constexpr int KG = CIN / 16, PW = 34, PH = TH + 2;
    constexpr int PRP = (PH * PW + 15) & ~15;
    const size_t lds = ((size_t)9 * CIN * COUT + (size_t)NB * KG * PRP * 16) * sizeof(float);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<CIN, COUT, TH, NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + TH - 1) / TH;
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.N;
    int per_cu = (int)((size_t)160 * 1024 / lds);
    if (per_cu > 4) per_cu = 4;
    long nwg = 256L * (per_cu < 1 ? 1 : per_cu);
    if (nwg > ntiles) nwg = ntiles;
    hipLaunchKernelGGL((conv3x3_halo_kernel<CIN, COUT, TH, NB>), dim3((unsigned)nwg), dim3(256), lds, s, a);
    return pwc_launch_status();
}

// ---------------------------------------------------------------- weight packing
// packed[tap][c16][cout_pad][16]: element (j*4+e) of row `co` holds
// w_hwio[tap][cin_map[c16*16 + ((j ^ swz4(co))*4 + e)]][co]   (0 for padding)
// i.e. the 16-byte chunk index is pre-swizzled so the LDS image is a linear copy.
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map,
                                    int Cin, int Cin_phys, int Cout, int Cout_pad, float* __restrict__ packed) {
    const size_t total = (size_t)9 * Cin_phys * Cout_pad;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e16 = (int)(idx & 15);
        size_t r = idx >> 4;
        const int co = (int)(r % Cout_pad);
        r /= Cout_pad;
        const int c16 = (int)(r % (Cin_phys >> 4));
        const int tap = (int)(r / (Cin_phys >> 4));
        const int jpos = e16 >> 2, e = e16 & 3;
        const int j = jpos ^ swz4(co);
        const int cphys = c16 * 16 + j * 4 + e;
        int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        float v = 0.f;
        if (clog >= 0 && clog < Cin && co < Cout) v = w[((size_t)tap * Cin + clog) * Cout + co];
        packed[idx] = v;
    }
}

// ---------------------------------------------------------------- split-K reduce
// y[pix][co] = act(bias[co] + sum_z ws[z][pix][co]); z summed in a fixed order, so the
// result is deterministic.
__global__ __launch_bounds__(256) void conv3x3_splitk_reduce_kernel(const float* __restrict__ ws,
                                                                    const float* __restrict__ bias, float* y,
                                                                    int y_cs, int y_vec4, int M, int Cout,
                                                                    int Cout_pad, int nsplit, int apply_act,
                                                                    float slope) {
    const int c4n = Cout >> 2;
    const long total = (long)M * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long pix = idx / c4n;
        f32x4 v = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
        for (int z = 0; z < nsplit; ++z)
            v += *reinterpret_cast<const f32x4*>(ws + ((size_t)z * M + pix) * Cout_pad + c4 * 4);
        if (apply_act) {
            v[0] = pwc_lrelu(v[0], slope); v[1] = pwc_lrelu(v[1], slope);
            v[2] = pwc_lrelu(v[2], slope); v[3] = pwc_lrelu(v[3], slope);
        }
        float* dst = y + (size_t)pix * y_cs + c4 * 4;
        if (y_vec4) *reinterpret_cast<f32x4*>(dst) = v;
        else { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    }
}

// ---------------------------------------------------------------- host side
typedef void (*conv_kernel_t)(const ConvArgs);

struct TileCfg {
    int BM, BN;
    int wgs_per_cu;   // resident workgroups per CU (LDS / VGPR limited), KC = 32 build
    conv_kernel_t k16, k32;
};

#define PWC_TILE(WM, WN, WGM, WGN, OCC)                                                   \
    { 16 * WM * WGM, 16 * WN * WGN, OCC, conv3x3_mfma_kernel<WM, WN, WGM, WGN, 16>,       \
      conv3x3_mfma_glds_kernel<WM, WN, WGM, WGN, 32> }

// KC = 16 (Cin_phys % 32 != 0) uses the register-staged kernel, KC = 32 the LDS-DMA one
// (measured on MI355X: LDS-DMA +2 % at KC = 32, -8 % at KC = 16).
static const TileCfg g_tiles[] = {
    PWC_TILE(4, 4, 2, 2, 2), // 0: 128 x 128
    PWC_TILE(4, 3, 2, 2, 2), // 1: 128 x 96
    PWC_TILE(4, 2, 2, 2, 3), // 2: 128 x 64
    PWC_TILE(4, 2, 4, 1, 2), // 3: 256 x 32
    PWC_TILE(4, 1, 4, 1, 2), // 4: 256 x 16
    PWC_TILE(2, 4, 2, 2, 3), // 5: 64 x 128
    PWC_TILE(2, 3, 2, 2, 3), // 6: 64 x 96
    PWC_TILE(2, 2, 2, 2, 4), // 7: 64 x 64
    PWC_TILE(2, 2, 4, 1, 4), // 8: 128 x 32
    PWC_TILE(2, 1, 4, 1, 4), // 9: 128 x 16
    PWC_TILE(1, 4, 2, 2, 3), // 10: 32 x 128
    PWC_TILE(1, 3, 2, 2, 4), // 11: 32 x 96
    PWC_TILE(1, 2, 2, 2, 6), // 12: 32 x 64
    PWC_TILE(1, 2, 4, 1, 6), // 13: 64 x 32
    PWC_TILE(1, 1, 4, 1, 6), // 14: 64 x 16
};
static const int g_ntiles = (int)(sizeof(g_tiles) / sizeof(g_tiles[0]));

// Launch plan of one convolution: tile configuration + optional split of the 9 taps over
// blockIdx.z (small M: too few tiles to fill 256 CUs).  tail_tile/m_main describe an
// optional second launch with a smaller tile for the last pixels (measured: no gain on
// MI355X -- a lone workgroup per CU gets the whole MFMA pipe -- so the planner never
// emits it; the kernel-side pixel range stays for explicit use).
struct ConvPlan {
    int tile, tail_tile, m_main, nsplit;
};

// Measured on MI355X (scripts/tune_conv.py, all layer shapes of the 8x448x1024 forward):
// the fastest configuration is the LARGEST tile that still yields >= ~384 workgroups,
// scanning BN downwards before resorting to the tap split.  Preferred tiles per BN, large
// BM first (256x32 is never the best: lower occupancy), BN >= 64 whenever Cout allows.
static ConvPlan plan_conv(int M, int Cout_pad, int Cin_phys, bool allow_split) {
    static const int kBN[5] = {128, 96, 64, 32, 16};
    static const int pref[5][3] = {{0, 5, 10}, {1, 6, 11}, {2, 7, 12}, {8, 13, -1}, {4, 9, 14}};
    const long target = 384;
    ConvPlan p;
    p.tail_tile = -1;
    p.m_main = M;
    p.nsplit = 1;
    p.tile = -1;
    const int max_split = (allow_split && Cin_phys >= 64) ? 9 : 1;
    int last_valid = -1;
    for (int split = 1; split <= max_split; split *= 3) {
        for (int r = 0; r < 5; ++r) {
            if (Cout_pad % kBN[r]) continue;
            if (kBN[r] < 64 && Cout_pad >= 64) continue;   // narrow tiles re-read the pixels too often
            for (int c = 0; c < 3; ++c) {
                const int t = pref[r][c];
                if (t < 0) continue;
                const TileCfg& tc = g_tiles[t];
                const long wgs = (long)((M + tc.BM - 1) / tc.BM) * (Cout_pad / tc.BN) * split;
                last_valid = t;
                if (wgs >= target) { p.tile = t; p.nsplit = split; return p; }
            }
        }
    }
    // nothing reaches the target: smallest tile of the widest usable BN, deepest split
    for (int r = 0; r < 5 && p.tile < 0; ++r) {
        if (Cout_pad % kBN[r]) continue;
        if (Cout_pad / kBN[r] * ((M + 31) / 32) * max_split >= 64 || r == 4) {
            for (int c = 2; c >= 0; --c)
                if (pref[r][c] >= 0) { p.tile = pref[r][c]; break; }
        }
    }
    if (p.tile < 0) p.tile = last_valid;
    p.nsplit = max_split;
    return p;
}

extern "C" int pwc_conv3x3_uses_halo_kernel(int M, int Cin_phys, int Cout, int stride, int dilation) {
    return (stride == 1 && dilation == 1 && Cin_phys == Cout && (Cout == 16 || Cout == 32) && M >= (1 << 16)) ? 1 : 0;
}

extern "C" int pwc_conv3x3_plan(int M, int Cout, int Cin_phys, int* plan4) {
    if (M <= 0 || Cout <= 0 || Cin_phys <= 0 || !plan4) return PWC_EINVAL;
    const ConvPlan p = plan_conv(M, (Cout + 15) & ~15, Cin_phys, true);
    plan4[0] = p.tile; plan4[1] = p.tail_tile; plan4[2] = p.m_main; plan4[3] = p.nsplit;
    return PWC_OK;
}

extern "C" int pwc_conv3x3_tile_shape(int tile, int* bm, int* bn) {
    if (tile < 0 || tile >= g_ntiles) return PWC_EINVAL;
    if (bm) *bm = g_tiles[tile].BM;
    if (bn) *bn = g_tiles[tile].BN;
    return PWC_OK;
}

extern "C" size_t pwc_conv3x3_workspace_floats(int M, int Cout) {
    // worst case: 9-way tap split of the whole output
    if (M <= 0 || Cout <= 0) return 0;
    return (size_t)9 * M * ((Cout + 15) & ~15);
}

extern "C" size_t pwc_conv3x3_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    const int Cout_pad = (Cout + 15) & ~15;
    return (size_t)9 * Cin_phys * Cout_pad;
}

extern "C" int pwc_conv3x3_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                    int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int Cout_pad = (Cout + 15) & ~15;
    const size_t total = (size_t)9 * Cin_phys * Cout_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, Cout_pad, packed);
    return pwc_launch_status();
}

static int launch_tile(const ConvArgs& base, int tile, int m_begin, int m_end, int nsplit, hipStream_t s) {
    const TileCfg& tc = g_tiles[tile];
    ConvArgs a = base;
    a.m_begin = m_begin;
    a.m_end = m_end;
    a.taps_per_split = 9 / nsplit;
    if (nsplit == 1) a.ws = nullptr;
    const int KC = (a.Cin_phys % 32 == 0) ? 32 : 16;
    conv_kernel_t k = KC == 32 ? tc.k32 : tc.k16;
    const size_t lds = (size_t)2 * (tc.BM + tc.BN) * KC * sizeof(float);
    dim3 grid((unsigned)((m_end - m_begin + tc.BM - 1) / tc.BM), (unsigned)(a.Cout_pad / tc.BN), (unsigned)nsplit);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_f32(const float* x, int x_cs, const float* packed, const float* bias, float* y,
                               int y_cs, int N, int H, int W, int Cin_phys, int Cout, int stride,
                               int dilation, int apply_act, float slope, int tile, int split,
                               float* workspace, size_t workspace_floats, pwc_stream_t stream) {
    if (!x || !packed || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0) return PWC_EINVAL;
    if (stride < 1 || stride > 2 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(packed) || !pwc_aligned16(bias)) return PWC_EALIGN;
    if (split != 0 && split != 1 && split != 3 && split != 9) return PWC_EINVAL;
    ConvArgs a;
    a.x = x; a.wp = packed; a.bias = bias; a.y = y;
    a.x_cs = x_cs; a.y_cs = y_cs;
    a.H = H; a.W = W;
    pwc_same_pad(H, stride, dilation, &a.Ho, &a.pad_t);
    pwc_same_pad(W, stride, dilation, &a.Wo, &a.pad_l);
    a.Cin_phys = Cin_phys; a.Cout = Cout; a.Cout_pad = (Cout + 15) & ~15;
    a.stride = stride; a.dil = dilation;
    a.apply_act = apply_act; a.slope = slope;
    const long M = (long)N * a.Ho * a.Wo;
    // within-image offsets and the pixel count are 32-bit in the kernel
    if (M >= (1L << 31) || (long)H * W * x_cs >= (1L << 31)) return PWC_ERANGE;
    a.M = (int)M;
    a.y_vec4 = ((y_cs & 3) == 0 && pwc_aligned16(y)) ? 1 : 0;
    a.ws = workspace;
    a.xcd_remap = 1;
    hipStream_t s = (hipStream_t)stream;

    // small-Cin full-resolution layers: resident-weights / halo-patch kernel
    if (tile < 0 && split == 0 && stride == 1 && dilation == 1 && Cin_phys == Cout && (Cout == 16 || Cout == 32) &&
        M >= (1L << 16) && (long)N * H * W * x_cs < (1L << 31)) {
        HaloArgs h;
        h.x = x; h.wp = packed; h.bias = bias; h.y = y; h.x_cs = x_cs; h.y_cs = y_cs;
        h.N = N; h.H = H; h.W = W; h.apply_act = apply_act; h.slope = slope; h.y_vec4 = a.y_vec4;
        h.tiles_x = h.tiles_y = 0;
        // measured (scripts/exp_halo.hip): 4-row tiles, one patch buffer, 4 workgroups per CU
        return Cout == 16 ? launch_halo<16, 16, 4, 1>(h, s) : launch_halo<32, 32, 4, 1>(h, s);
    }
    const bool ws_ok = workspace && pwc_aligned16(workspace);
    ConvPlan p;
    if (tile < 0) {
        p = plan_conv(a.M, a.Cout_pad, Cin_phys, ws_ok && split != 1);
        if (split > 1) p.nsplit = split;
    } else {
        if (tile >= g_ntiles) return PWC_EINVAL;
        p.tile = tile; p.tail_tile = -1; p.m_main = a.M; p.nsplit = split > 1 ? split : 1;
    }
    if (a.Cout_pad % g_tiles[p.tile].BN) return PWC_EINVAL;
    if (p.nsplit > 1) {
        if (!ws_ok || workspace_floats < (size_t)p.nsplit * a.M * a.Cout_pad) return PWC_EINVAL;
        int rc = launch_tile(a, p.tile, 0, a.M, p.nsplit, s);
        if (rc) return rc;
        const long total = (long)a.M * (Cout >> 2);
        long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv3x3_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, bias,
                           y, y_cs, a.y_vec4, a.M, Cout, a.Cout_pad, p.nsplit, apply_act, slope);
        return pwc_launch_status();
    }
    int rc = launch_tile(a, p.tile, 0, p.m_main, 1, s);
    if (rc) return rc;
    if (p.tail_tile >= 0 && p.m_main < a.M) rc = launch_tile(a, p.tail_tile, p.m_main, a.M, 1, s);
    return rc;
}
