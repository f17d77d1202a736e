// conv3x3_wino.hip -- 3x3 stride-1 dilation-1 'SAME' convolution by Winograd F(2x2, 3x3)
// on the fp32 MFMA units of gfx950.
//
// Replaces the same tf.layers.Conv2D(...,(3,3),(1,1),'same') + tf.nn.leaky_relu calls as
// conv3x3_mfma.hip (reference modules.py:64-67, 267-268, 306-307, 322-323) wherever
// stride = dilation = 1 and Cout % 32 == 0: 16 multiplies per 2x2 outputs instead of 36,
// i.e. 2.25x fewer MFMA instructions for the same result (fp32 error ~1e-7 relative).
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
//   LDS per 16-channel stage: the raw 18 x 18 pixel patch (64-byte rows, XOR-swizzled) and
//   the transformed weights U[xi][32 cout][16 ch] (pre-swizzled by the packer), both filled
//   by global_load_lds_dwordx4; out-of-image pixels read a zero page (SAME padding).
#include "pwc_common.h"

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
};

__device__ float wino_zero_page[4];

__device__ __forceinline__ int wswz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

constexpr int WN_PW = 18;                 // patch width/height (8 tiles * 2 + 2)
constexpr int WN_PR = WN_PW * WN_PW;      // 324 patch pixels
constexpr int WN_PRP = 336;               // padded to 21 DMA blocks of 16 rows
constexpr int WN_NBP = WN_PRP / 16;       // 21
// NT = 16-cout MFMA tiles per workgroup (2: 32 output channels, 1: 16)
template <int NT> struct WinoGeom {
    static constexpr int BN = 16 * NT;                // output channels per workgroup
    static constexpr int UROWS = 16 * BN;             // weight rows (xi, cout) per stage
    static constexpr int NBU = UROWS / 16;
    static constexpr int STAGE = (WN_PRP + UROWS) * 16;   // floats per LDS stage (NT = 2: 54 272 B)
};

// ABL (scripts/exp_wino.hip only, 0 in the library): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA
template <int NSTG, int ABL = 0, int NT = 2>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const WinoArgs a) {
    constexpr int WN_BN = WinoGeom<NT>::BN, WN_NBU = WinoGeom<NT>::NBU, WN_STAGE = WinoGeom<NT>::STAGE;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const float* zero = wino_zero_page;

    if (ABL & 24) {
        // experiments: break the lockstep of the two co-resident workgroups of a CU
        const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | 4);   // HW_REG_HW_ID[3:0] = wave slot
        if (__builtin_amdgcn_readfirstlane(hw) & 1) {
            if (ABL & 8) __builtin_amdgcn_s_setprio(1);
            if (ABL & 16) { __builtin_amdgcn_s_sleep(32); __builtin_amdgcn_s_sleep(32); }
        }
    }

    // block decode: cout block fastest, XCD-aware (the cout blocks of one pixel block share
    // their input patch in one XCD's L2)
    const int nblk = gridDim.x;
    const int lb = pwc_xcd_remap(blockIdx.x, nblk);
    const int cb = lb % a.ncb;
    int rest = lb / a.ncb;
    const int bx = rest % a.tiles_x;
    rest /= a.tiles_x;
    const int by = rest % a.tiles_y;
    rest /= a.tiles_y;
    const int d = a.dil;
    const int sub = rest % (d * d);                // sub-lattice (y mod d, x mod d) of a dilated conv
    const int n = rest / (d * d);
    const int ry = sub / d, rx = sub - ry * d;
    const int y0 = by * 16, x0 = bx * 16;          // output origin of the block, in sub-lattice coordinates
    const int n0 = cb * WN_BN;
    const int Cout_pad = (a.Cout + 15) & ~15;
    const int nc16 = a.Cin_phys >> 4;
    const float* xn = a.x + (size_t)n * a.H * a.W * a.x_cs;

    // ---- DMA bookkeeping (fixed over the channel loop).  Patch blocks (21) and weight blocks
    // (16*NT) are dealt to the 4 waves separately, so each unrolled issue knows its base at
    // compile time: per stage a lane does "offset + stage advance -> 64-bit add -> DMA".
    constexpr int PPW = (WN_NBP + 3) / 4;          // patch blocks per wave (6, the last partly unused)
    constexpr int UPW = WN_NBU / 4;                // weight blocks per wave (8 for NT = 2)
    static_assert(WN_NBU % 4 == 0, "weight blocks must split evenly over the waves");
    int p_off[PPW];                                // element offset of the lane's 16-byte chunk in the image, -1: zeros
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int b = wave + 4 * i;
        const int pr = b * 16 + (lane >> 2);
        int v = -1;
        if (b < WN_NBP && pr < WN_PR) {
            const int py = pr / WN_PW, px = pr - py * WN_PW;
            const int y = ry + d * (y0 - 1 + py), x = rx + d * (x0 - 1 + px);
            const int j = (lane & 3) ^ wswz(pr);                   // source chunk for this LDS slot
            if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) v = (y * a.W + x) * a.x_cs + j * 4;
        }
        p_off[i] = v;
    }
    // weight blocks: block wave + 4*i holds rows (xi, cout) = ((wave + 4*i) * 16 + lane/4); 4 blocks
    // = 64 rows = 64 / WN_BN positions, so one per-lane offset plus a uniform stride covers all i
    static_assert(64 % WN_BN == 0, "weight block stride must be a whole number of positions");
    int u_src0;
    {
        const int ur = wave * 16 + (lane >> 2);
        const int xi = ur / WN_BN, co = ur - xi * WN_BN;
        u_src0 = (n0 + co < Cout_pad) ? ((xi * nc16) * Cout_pad + n0 + co) * 16 + (lane & 3) * 4 : -1;
    }
    const int u_step = (64 / WN_BN) * nc16 * Cout_pad * 16;
    auto issue_stage = [&](int c16, int buf) {
        float* dst = smem + buf * WN_STAGE;
        const float* xs = xn + c16 * 16;
        const float* us = a.up + (size_t)c16 * Cout_pad * 16;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int b = wave + 4 * i;
            if (b < WN_NBP) {
                const float* src = p_off[i] >= 0 ? xs + p_off[i] : zero;
                if (!(ABL & 1)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + b * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const float* src = u_src0 >= 0 ? us + u_src0 + i * u_step : zero;
            if (!(ABL & 2))
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (WN_NBP + wave + 4 * i) * 256), 16, 0, 0);
        }
    };

    // ---- this lane's tile and its 16 patch read offsets (floats, swizzled for k-slot fq)
    const int tr = 2 * wave + (fr >> 3), tc = fr & 7;
    int poff[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (2 * tr + i) * WN_PW + 2 * tc + j;
            poff[i][j] = row * 16 + ((fq ^ wswz(row)) << 2);
        }
    const int u_off = WN_PRP * 16 + fr * 16 + ((fq ^ wswz(fr)) << 2);   // A-fragment row fr of a 16-row tile

    f32x4 acc[16][NT];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[xi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int cur = 0;
    if (NSTG == 2) issue_stage(0, 0);
    for (int c16 = 0; c16 < nc16; ++c16) {
        if (NSTG == 1) {
            __syncthreads();                         // previous stage fully read
            issue_stage(c16, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NSTG == 2 && c16 + 1 < nc16) issue_stage(c16 + 1, cur ^ 1);
        const float* sb = smem + cur * WN_STAGE;

        // ---- input transform  V = B^T d B  (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]), in place
        f32x4 v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = *reinterpret_cast<const f32x4*>(sb + poff[i][j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                // rows
            const f32x4 d0 = v[0][j], d1 = v[1][j], d2 = v[2][j], d3 = v[3][j];
            v[0][j] = d0 - d2; v[1][j] = d1 + d2; v[2][j] = d2 - d1; v[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                // columns
            const f32x4 e0 = v[i][0], e1 = v[i][1], e2 = v[i][2], e3 = v[i][3];
            v[i][0] = e0 - e2; v[i][1] = e1 + e2; v[i][2] = e2 - e1; v[i][3] = e1 - e3;
        }
        // ---- 16 positions x NT cout tiles x 4 k-steps of MFMA, two positions interleaved so
        // that 2*NT independent accumulators rotate (no dependent back-to-back issue)
        f32x4 wf[2][NT];                         // [ring slot][cout tile]
        auto load_w = [&](int slot, int xi) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[slot][nt] = *reinterpret_cast<const f32x4*>(sb + u_off + (xi * WN_BN + nt * 16) * 16);
        };
        load_w(0, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int slot = xi & 1;
            if (ABL & 32) {
                // (experiment; spills at 256 VGPRs and measured slower) the NEXT position's weights are requested before this position's 4*NT MFMAs
                // (~250 cycles of cover for the LDS latency); the scheduling barrier keeps them here
                if (xi + 1 < 16) load_w(slot ^ 1, xi + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else if (xi > 0) load_w(slot, xi);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (ABL & 4) { asm volatile("" ::"v"(wf[slot][nt][k]), "v"(v[xi >> 2][xi & 3][k])); continue; }
                    acc[xi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][nt][k], v[xi >> 2][xi & 3][k],
                                                                       acc[xi][nt], 0, 0, 0);
                }
        }
        if (NSTG == 2) cur ^= 1;
    }

    // ---- output transform  Y = A^T M A  (A^T = [1 1 1 0; 0 1 -1 -1]), bias, leaky-relu, stores
    const int oy = y0 + 2 * tr, ox = x0 + 2 * tc;
    const int py0 = ry + d * oy, px0 = rx + d * ox;                   // real coordinates of output (0,0)
    const bool okr[2] = {py0 < a.H, py0 + d < a.H};
    const bool okc[2] = {px0 < a.W, px0 + d < a.W};
    float* out00 = a.y + ((size_t)(n * a.H + py0) * a.W + px0) * a.y_cs;
    const size_t dcol = (size_t)d * a.y_cs, drow = (size_t)d * a.W * a.y_cs;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 16 + fq * 4;
        if (co >= a.Cout) continue;
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[0 * 4 + j][nt] + acc[1 * 4 + j][nt] + acc[2 * 4 + j][nt];
            s[1][j] = acc[1 * 4 + j][nt] - acc[2 * 4 + j][nt] - acc[3 * 4 + j][nt];
        }
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 yv = (j == 0) ? s[i][0] + s[i][1] + s[i][2] : s[i][1] - s[i][2] - s[i][3];
                yv += b4;
                if (a.apply_act) {
                    yv[0] = pwc_lrelu(yv[0], a.slope); yv[1] = pwc_lrelu(yv[1], a.slope);
                    yv[2] = pwc_lrelu(yv[2], a.slope); yv[3] = pwc_lrelu(yv[3], a.slope);
                }
                if (okr[i] && okc[j]) {
                    float* dst = out00 + i * drow + j * dcol + co;
                    if (a.y_vec4) *reinterpret_cast<f32x4*>(dst) = yv;
                    else { dst[0] = yv[0]; dst[1] = yv[1]; dst[2] = yv[2]; dst[3] = yv[3]; }
                }
            }
        }
    }
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

extern "C" long pwc_conv3x3_wino_workgroups(int N, int H, int W, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || dilation < 1 || Cout % 16) return 0;
    const long tx = ((W + dilation - 1) / dilation + 15) / 16, ty = ((H + dilation - 1) / dilation + 15) / 16;
    return (long)N * dilation * dilation * tx * ty * (Cout % 32 == 0 ? Cout / 32 : Cout / 16);
}

extern "C" int pwc_conv3x3_wino_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                    int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                    int apply_act, float slope, pwc_stream_t stream) {
    if (!x || !packed_u || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 16) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(packed_u) || !pwc_aligned16(bias)) return PWC_EALIGN;
    if ((long)H * W * x_cs >= (1L << 31)) return PWC_ERANGE;
    WinoArgs a;
    a.x = x; a.up = packed_u; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout;
    a.apply_act = apply_act; a.slope = slope;
    a.dil = dilation;
    a.tiles_x = ((W + dilation - 1) / dilation + 15) / 16;
    a.tiles_y = ((H + dilation - 1) / dilation + 15) / 16;
    const int bn = (Cout % 32 == 0) ? 32 : 16;
    a.ncb = Cout / bn;
    a.y_vec4 = ((y_cs & 3) == 0 && pwc_aligned16(y)) ? 1 : 0;
    const long nblk = (long)N * dilation * dilation * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    // measured (scripts/exp_wino.hip): one LDS stage with 2 co-resident workgroups per CU beats both a
    // double-buffered stage (1 workgroup per CU) and a split-weights pipeline
    if (bn == 32)
        hipLaunchKernelGGL((conv3x3_wino_kernel<1, 0, 2>), dim3((unsigned)nblk), dim3(256),
                           (size_t)WinoGeom<2>::STAGE * sizeof(float), (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv3x3_wino_kernel<1, 0, 1>), dim3((unsigned)nblk), dim3(256),
                           (size_t)WinoGeom<1>::STAGE * sizeof(float), (hipStream_t)stream, a);
    return pwc_launch_status();
}
