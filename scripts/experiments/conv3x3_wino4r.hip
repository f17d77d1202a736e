// conv3x3_wino4r.hip -- 3x3 stride-1 'SAME' convolution by Winograd F(4x4, 3x3) on the fp32 MFMA units of gfx950,
// ROLE-SPECIALISED: the input transform runs on one SIMD of the CU, the matrix instructions on the other three (round 4).
//
// Replaces the same tf.layers.Conv2D(...,(3,3),(1,1),'same',dilation_rate=d) + tf.nn.leaky_relu calls as
// conv3x3_wino4.hip (reference modules.py:266-268 `optflow_l/conv2d .. conv2d_3`, modules.py:306-323 `context/conv2d*`);
// same arithmetic: fp32 throughout, U = G g G^T computed in double and rounded once, B^T d B and A^T M A as there.
//
// Why: on gfx950 an fp32 MFMA excludes every other instruction of its SIMD (VALU, LDS, fetch issue: the times ADD,
// DESIGN.md 3.4), so conv3x3_wino4.hip -- every wave transforms AND multiplies, once per 16 couts -- sits at 0.38 of the
// fp32 matrix peak under an instruction-mix cap of 0.62.  Different SIMDs of a CU ARE independent (scripts/exp_cross_simd.hip,
// profiles/r04_exp_cross_simd.txt: MFMA waves on SIMDs 1-3 beside a VALU / ds_write wave on SIMD 0 take the MAX).  So:
//
//   workgroup = 512 threads = 8 waves, one per CU: 4 x 8 Winograd tiles (16 x 32 output pixels) x 32 output channels.
//   Every SIMD hosts ONE producer and ONE consumer wave (a workgroup's waves go round the four SIMDs: waves w and w + 4
//   share one):
//   Waves 0..3 are PRODUCERS: wave = (tile group g = tile rows 2g, 2g+1; row half ah = rows a = 3ah .. 3ah+2 of the 6 x 6
//     transformed tile).  Lane (tile j = lane & 15, k-slot q = lane >> 4) reads the 6 x 6 input pixels of its tile for
//     channels 4q..4q+3 from the raw patch in LDS (conv3x3_wino4.hip's image), forms its three rows of  B^T d B  (row
//     pass for 3 rows, column pass for all 6 columns: 84 f32x4 operations -- ONCE per workgroup = per 32 couts; the
//     16-cout kernel does it once per 16 couts on every wave) and publishes V[position][tile group] -- 1 KB each, already in
//     B-fragment order -- to LDS, one row (6 positions) per phase, one phase ahead of its use.  The producers also issue
//     every fetch (buffer_load ... lds) of the workgroup.
//   Waves 4..7 are CONSUMERS: wave c owns 3 of the 12 positions of each part (9 of the 36) for both tile groups and both
//     16-cout tiles: 36 accumulator tiles = 144 registers.  Per position and 16-channel stage: 4 ds_read_b128 (two weight
//     fragments, two V fragments: 4 k-steps each) feed 16 v_mfma_f32_16x16x4_f32 -- nothing else is issued by these waves.
//   A phase = one part p (rows a = p and p + 3: 12 positions) of a 16-channel stage; one s_barrier per phase.  LDS: patch
//   42 KB (single buffer, read by the producers only), weights 2 x 24 KB and V 2 x 24 KB rings of parts.
//   Epilogue: the accumulators go through LDS once (144 KB, M[position][tile group][cout tile]), then all 8 waves apply
//   A^T M A, bias and leaky-relu to (tile, 4 couts) units and store.
#pragma once
#include "../../pwcnet_amd/csrc/pwc_common.h"
#include <type_traits>

struct Wino4rArgs {
    const float* x;
    const float* up;     // packed transformed weights [c16][cout group of 32][xi 36][cout tile 2][k-slot 4][cout 16][4]
    const float* bias;
    float* y;
    int x_cs, y_cs;
    int N, H, W;
    int Cin_phys, Cout;
    int apply_act;
    float slope;
    int tiles_x, tiles_y, ncb;   // 16x32-pixel blocks per (sub-)image, cout groups of 32
    int dil;
    int ntiles;
};

constexpr unsigned WR_OOB = 0x7FFF0000u;
constexpr int WR_NW = 8, WR_T = 64 * WR_NW;
constexpr int WR_PS = 36;                    // patch records per patch row (conv3x3_wino4.hip's image)
constexpr int WR_PH = 18, WR_PW = 34;
constexpr int WR_PPW = 11;                   // patch DMA pieces per PRODUCER wave and stage: 44 requests for the 41 blocks
constexpr int WR_NBP = 42;                   // 41 blocks + block 41 that swallows the surplus (out-of-range) request
constexpr int WR_PATCH_BYTES = WR_NBP * 1024;
constexpr int WR_PART = 12 * 2048;           // bytes of a part of the weights (12 positions x 2 cout tiles x 1 KB) = of a part of V
constexpr int WR_UPW = 6;                    // weight DMA pieces per producer wave and part
constexpr int WR_U0 = WR_PATCH_BYTES;        // two weight slots
constexpr int WR_V0 = WR_U0 + 2 * WR_PART;   // two V slots
constexpr int WR_MAIN = WR_V0 + 2 * WR_PART; // 141 312 B
constexpr int WR_M_BYTES = 36 * 4096;        // epilogue: M[position][tile group][cout tile] x 1 KB = 147 456 B
constexpr int WR_LDS = WR_M_BYTES > WR_MAIN ? WR_M_BYTES : WR_MAIN;
static_assert(WR_LDS <= 160 * 1024, "one workgroup per CU");

__device__ __forceinline__ int wr_pswz(int py) { return ((py >> 2) & 1) << 1; }   // = w4_pswz

#define WRSUB(p, q) __builtin_elementwise_fma((q), M1, (p))                       /* p - q, packable */
#define WRFMA(x, c, y) __builtin_elementwise_fma((x), f32x4{c, c, c, c}, (y))     /* x * c + y */

// ABL (scripts/exp_wino4r.hip only; 0 in the library): 1 = no patch DMA, 2 = no weight DMA, 4 = no MFMA,
// 64 = no transform arithmetic, 128 = s_memtime stamps instead of the output
template <int ABL = 0>
__global__ __launch_bounds__(WR_T, 2) void conv3x3_wino4r_kernel(const Wino4rArgs a) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = wave < 4;
    const int fr = lane & 15, fq = lane >> 4;
    float m1s;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1s));   // -1.0f the optimiser cannot see through (see conv3x3_wino.hip)
    const f32x4 M1 = {m1s, m1s, m1s, m1s};

    const int d = a.dil;
    const int nc16 = a.Cin_phys >> 4;
    const int nphase = 3 * nc16;

    // ---- block decode: cout group fastest, XCD-aware (the cout groups of a pixel block share its patch in one L2)
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
    const int n0 = cb * 32;

#define WR_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#define WR_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    int stamp_n = 0;
    auto stamp = [&]() {
        if (ABL & 128) {
            const unsigned long long tm = __builtin_readcyclecounter();
            if (lane == 0 && stamp_n < 144) *reinterpret_cast<unsigned long long*>(sm + WR_LDS + (wave * 144 + stamp_n) * 8) = tm;
            ++stamp_n;
        }
    };

    if (producer) {
        // =========================================================== producers: waves 0..3 = (tile group, row half), one per SIMD
        const int g = wave & 1, ah = wave >> 1;
        const int trl = fr >> 3, tc = fr & 7;       // tile (2g + trl, tc)
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.x + (size_t)n * a.H * a.W * a.x_cs), 0, a.H * a.W * a.x_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.up, 0, nc16 * a.ncb * 36 * 2048, 0x00020000);
        // patch fetch: blocks wave, wave + 4, ... of 16 records x 64 bytes; per-lane byte offsets fixed over the channel loop
        unsigned p_voff[WR_PPW];
#pragma unroll
        for (int i = 0; i < WR_PPW; ++i) {
            const int rec = (wave + 4 * i) * 16 + (lane >> 2);
            const int py = rec / WR_PS, rem = rec - py * WR_PS;
            const int q = rem / 9, ci = rem - q * 9;
            const int px = 4 * ci + q;
            const int yy = ry + d * (y0 - 1 + py), xx = rx + d * (x0 - 1 + px);
            const int ch = (lane & 3) ^ wr_pswz(py);                       // source chunk for this LDS slot
            const bool ok = py < WR_PH && px < WR_PW && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            p_voff[i] = ok ? (unsigned)(((yy * a.W + xx) * a.x_cs + ch * 4) * 4) : WR_OOB;
        }
        auto issue_patch = [&](int c16) {
#pragma unroll
            for (int i = 0; i < WR_PPW; ++i)
                if (!(ABL & 1))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        xrsrc, (lptr_t)(sm + (wave + 4 * i < WR_NBP - 1 ? wave + 4 * i : WR_NBP - 1) * 1024), 16, (int)p_voff[i], c16 * 64, 0, 0);
        };
        // weights: part k = 3 c + p is 24 KB contiguous in the packed image; pieces wave, wave + 4, ...; everything but the
        // lane's 16 bytes is wave-uniform (scalar offset)
        const unsigned u_lane = (unsigned)lane * 16u;
        auto issue_u = [&](int k) {
            const int c16 = k / 3, p = k - 3 * c16;
            const int sbase = ((c16 * a.ncb + cb) * 36 + 12 * p) * 2048;
            char* const dst = sm + WR_U0 + (k & 1) * WR_PART;
#pragma unroll
            for (int j = 0; j < WR_UPW; ++j)
                if (!(ABL & 2))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lptr_t)(dst + (wave + 4 * j) * 1024), 16, (int)u_lane,
                                                             sbase + (wave + 4 * j) * 1024, 0, 0);
        };
        // this lane's patch reads (conv3x3_wino4.hip): record (4 trow + i) * 36 + (j & 3) * 9 + (j >> 2) + tc, chunk
        // fq ^ pswz(py); pswz flips between window rows i < 4 and i >= 4
        const int trow = 2 * g + trl;
        const float* pb_lo = smem + ((4 * trow) * WR_PS + tc) * 16 + ((fq ^ wr_pswz(4 * trow)) << 2);
        const float* pb_hi = smem + ((4 * trow) * WR_PS + tc) * 16 + ((fq ^ wr_pswz(4 * trow + 4)) << 2);

        f32x4 V[3][6];
        // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        f32x4 Wn[3][6];                              // row-pass results of the NEXT stage (V still holds this stage's rows)
        auto rowpass = [&](int j0, int j1) {         // columns j0 .. j1-1 of the window: this wave's rows a = 3ah .. 3ah+2 of B^T d
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j < j0 || j >= j1) continue;
                f32x4 dd[6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    dd[i] = *reinterpret_cast<const f32x4*>((i < 4 ? pb_lo : pb_hi) + (i * WR_PS + (j & 3) * 9 + (j >> 2)) * 16);
                if (ABL & 64) {
                    Wn[0][j] = dd[0] + dd[3]; Wn[1][j] = dd[1] + dd[4]; Wn[2][j] = dd[2] + dd[5];
                } else if (ah == 0) {
                    Wn[0][j] = WRFMA(dd[0], 4.f, WRFMA(dd[2], -5.f, dd[4]));
                    const f32x4 s = dd[1] + dd[2], tt = dd[3] + dd[4], u = WRSUB(dd[1], dd[2]), v = WRSUB(dd[4], dd[3]);
                    Wn[1][j] = WRFMA(s, -4.f, tt);
                    Wn[2][j] = WRFMA(u, 4.f, v);
                } else {
                    const f32x4 p = WRSUB(dd[4], dd[2]), q = WRSUB(dd[3], dd[1]);
                    Wn[0][j] = WRFMA(q, 2.f, p);
                    Wn[1][j] = WRFMA(q, -2.f, p);
                    Wn[2][j] = WRFMA(dd[1], 4.f, WRFMA(dd[3], -5.f, dd[5]));
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) asm("" : "+v"(Wn[r][j]));         // keep the packed ops (see conv3x3_wino.hip)
            }
        };
        auto colpass = [&]() {                       // V = (rows of B^T d) B, all six columns b
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 e0 = Wn[r][0], e1 = Wn[r][1], e2 = Wn[r][2], e3 = Wn[r][3], e4 = Wn[r][4], e5 = Wn[r][5];
                if (ABL & 64) {
                    V[r][0] = e0; V[r][1] = e1; V[r][2] = e2; V[r][3] = e3; V[r][4] = e4; V[r][5] = e5;
                } else {
                    const f32x4 s = e1 + e2, tt = e3 + e4, u = WRSUB(e1, e2), v = WRSUB(e4, e3);
                    const f32x4 p = WRSUB(e4, e2), q = WRSUB(e3, e1);
                    V[r][0] = WRFMA(e0, 4.f, WRFMA(e2, -5.f, e4));
                    V[r][1] = WRFMA(s, -4.f, tt);
                    V[r][2] = WRFMA(u, 4.f, v);
                    V[r][3] = WRFMA(q, 2.f, p);
                    V[r][4] = WRFMA(q, -2.f, p);
                    V[r][5] = WRFMA(e1, 4.f, WRFMA(e3, -5.f, e5));
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) asm("" : "+v"(V[r][j]));
            }
        };
        auto publish_row = [&](auto pc, int k) {     // part k = 3 c + p: this wave's row a = 3ah + p -> V slot k & 1, entries 6ah + b
            constexpr int PP = decltype(pc)::value;  // (compile-time row: V must stay in registers)
            char* const dst = sm + WR_V0 + (k & 1) * WR_PART + (6 * ah) * 2048 + g * 1024 + lane * 16;
#pragma unroll
            for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x4*>(dst + b * 2048) = V[PP][b];
        };
        // ---- prologue: patch(0), weight part 0 -> transform stage 0, publish part 0
        issue_patch(0);
        issue_u(0);
        WR_WAIT_VM(WR_UPW);                          // patch(0) landed (this wave's pieces)
        WR_BAR();                                    // B0: ... every producer's pieces
        rowpass(0, 6);
        colpass();
        publish_row(std::integral_constant<int, 0>{}, 0);
        WR_WAIT_VM(0);                               // weight part 0 landed
        // A phase with its part index as a compile-time constant (V stays in registers: a run-time row selection made hipcc
        // keep V in SCRATCH memory -- 512 us instead of 238).  Schedules measured on the 128 -> 128 layer at 8 x 112 x 256
        // (profiles/r04_exp_wino4r_role_specialised.txt): whole transform in part 2 (this form) 237.8 us; row pass in part 1 and
        // column pass in part 2: 242.7 us; row pass split over parts 0 and 1 with a second register image: 254.2 us -- a
        // producer instruction issued while its SIMD's consumer has MFMAs queued waits for the 32-cycle MFMA in front of it,
        // so spreading the producer's work makes every phase longer instead of shortening the long one.
        auto phase = [&](auto pc, int c16) {
            constexpr int P = decltype(pc)::value;
            const int k = 3 * c16 + P;
            const bool has_next = c16 + 1 < nc16;
            stamp();
            WR_BAR();                                // top of phase k: part k of weights and V published; slots of part k-1 free
            stamp();
            if (P < 2 || has_next) issue_u(k + 1);
            if (P == 0 && has_next) issue_patch(c16 + 1);      // (the patch buffer was last read in phase (c-1, 2))
            stamp();
            if (P < 2) {
                publish_row(std::integral_constant<int, (P + 1) % 3>{}, k + 1);
                // fetches in flight (oldest first): P == 0: weights k+1, patch(c+1); P == 1: patch(c+1), weights k+1
                if (P == 0) { if (has_next) WR_WAIT_VM(WR_PPW); else WR_WAIT_VM(0); }
                else WR_WAIT_VM(0);                  // weights k+1 AND the next patch landed: the next phase transforms
            } else if (has_next) {
                rowpass(0, 6);                       // stage c+1 (patch(c+1) was published by the barrier above)
                colpass();
                publish_row(std::integral_constant<int, 0>{}, k + 1);
                WR_WAIT_VM(0);
            }
            stamp();
        };
        for (int c16 = 0; c16 < nc16; ++c16) {
            phase(std::integral_constant<int, 0>{}, c16);
            phase(std::integral_constant<int, 1>{}, c16);
            phase(std::integral_constant<int, 2>{}, c16);
        }
        stamp();
        WR_BAR();                                    // last phase read by every consumer
    } else {
        // =========================================================== consumers: waves 4..7, one per SIMD
        const int cj = wave - 4;                     // positions e = 3 cj .. 3 cj + 2 of every part
        const char* const ubase = sm + WR_U0 + (3 * cj) * 2048 + lane * 16;
        const char* const vbase = sm + WR_V0 + (3 * cj) * 2048 + lane * 16;
        f32x4 acc[9][2][2];                          // [3 p + i: part p, position 3 cj + i][tile group][cout tile]
#pragma unroll
        for (int pe = 0; pe < 9; ++pe)
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) acc[pe][tg][0] = acc[pe][tg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        WR_BAR();                                    // B0
        for (int k = 0; k < nphase; ++k) {
            const int p = k % 3;
            stamp();
            WR_BAR();                                // top of phase k
            stamp();
            stamp();
            const char* const us = ubase + (k & 1) * WR_PART;
            const char* const vs = vbase + (k & 1) * WR_PART;
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
                if (pp == p) {
                    f32x4 A0[3], A1[3], B0[3], B1[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) {    // the twelve reads of the phase first: a consumer has no partner wave to hide them
                        A0[e] = *reinterpret_cast<const f32x4*>(us + e * 2048);
                        A1[e] = *reinterpret_cast<const f32x4*>(us + e * 2048 + 1024);
                        B0[e] = *reinterpret_cast<const f32x4*>(vs + e * 2048);
                        B1[e] = *reinterpret_cast<const f32x4*>(vs + e * 2048 + 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);       // (the scheduler would sink the reads behind the MFMAs of the position before)
#pragma unroll
                    for (int e = 0; e < 3; ++e)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct) {
                                    f32x4& c = acc[3 * pp + e][tg][ct];
                                    if (ABL & 4) { asm volatile("" ::"v"(A0[e]), "v"(A1[e]), "v"(B0[e]), "v"(B1[e])); continue; }
                                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(ct ? A1[e][s] : A0[e][s], tg ? B1[e][s] : B0[e][s], c, 0, 0, 0);
                                }
                }
            stamp();
        }
        stamp();
        WR_BAR();                                    // every wave is past its last read of the rings
        // ---- accumulators -> LDS: M[position 36][tile group 2][cout tile 2] x 1 KB (lane-contiguous 16 bytes)
#pragma unroll
        for (int pe = 0; pe < 9; ++pe) {
            const int e = 3 * cj + (pe % 3);         // entry of part p = pe / 3: rows a = p (e < 6) and p + 3
            const int xi = e < 6 ? 6 * (pe / 3) + e : 6 * (pe / 3 + 3) + (e - 6);
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    *reinterpret_cast<f32x4*>(sm + xi * 4096 + (tg * 2 + ct) * 1024 + lane * 16) = acc[pe][tg][ct];
        }
    }
    WR_BAR();                                        // M complete
    // ---- output transform  Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1], by all 8 waves:
    // wave w: unit (tile group, cout tile) = w & 3, output rows 2 (w >> 2), 2 (w >> 2) + 1 of every tile; lane = (tile
    // fr, couts 4 fq .. 4 fq + 3) as the accumulators were
    {
        const int unit = wave & 3, tg = unit >> 1, ct = unit & 1, half = wave >> 2;
        const char* const mb = sm + unit * 1024 + lane * 16;
        f32x4 Z[6][4];                               // column pass: Z[a][j'] = sum_b M[a][b] A^T[j'][b]
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x4 m[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) m[b] = *reinterpret_cast<const f32x4*>(mb + (6 * r + b) * 4096);
            const f32x4 s12 = m[1] + m[2], d12 = WRSUB(m[1], m[2]), s34 = m[3] + m[4], d34 = WRSUB(m[3], m[4]);
            Z[r][0] = m[0] + s12 + s34;
            Z[r][1] = WRFMA(d34, 2.f, d12);
            Z[r][2] = WRFMA(s34, 4.f, s12);
            Z[r][3] = WRFMA(d34, 8.f, d12) + m[5];
        }
        const int co = n0 + ct * 16 + fq * 4;
        const int trl = fr >> 3, tc = fr & 7, trow = 2 * tg + trl;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (size_t)n * a.H * a.W * a.y_cs), 0, a.H * a.W * a.y_cs * 4, 0x00020000);
        if (co < a.Cout) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + co);
            const int px0 = rx + d * (x0 + 4 * tc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 s12 = Z[1][j] + Z[2][j], d12 = WRSUB(Z[1][j], Z[2][j]);
                const f32x4 s34 = Z[3][j] + Z[4][j], d34 = WRSUB(Z[3][j], Z[4][j]);
                f32x4 yv[2];
                if (half == 0) { yv[0] = Z[0][j] + s12 + s34; yv[1] = WRFMA(d34, 2.f, d12); }
                else { yv[0] = WRFMA(s34, 4.f, s12); yv[1] = WRFMA(d34, 8.f, d12) + Z[5][j]; }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 o = yv[i] + b4;
                    if (a.apply_act) {               // tf.nn.leaky_relu = max(v, slope * v)
                        const f32x4 sv = o * a.slope;
                        o[0] = fmaxf(o[0], sv[0]); o[1] = fmaxf(o[1], sv[1]); o[2] = fmaxf(o[2], sv[2]); o[3] = fmaxf(o[3], sv[3]);
                    }
                    const int py = ry + d * (y0 + 4 * trow + 2 * half + i), px = px0 + j * d;
                    const unsigned vo = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.y_cs + co) * 4) : WR_OOB;
                    if (!(ABL & 128)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, (int)vo, 0, 0);
                }
            }
        }
    }
    if (ABL & 128) {
        stamp();
        WR_BAR();
        if (blockIdx.x == 0)
            for (int i = t; i < WR_NW * 144 * 2; i += WR_T) reinterpret_cast<unsigned*>(a.y)[i] = reinterpret_cast<unsigned*>(sm + WR_LDS)[i];
    }
#undef WR_WAIT_VM
#undef WR_BAR
}
#undef WRSUB
#undef WRFMA

// ---------------------------------------------------------------- weight transform + packing
// packed[c16][cout group][part 3][entry 12][cout tile 2][k-slot q 4][cout 16][4 channels 4q..4q+3]: part p holds the rows
// a = p (entries 0..5: b) and a = p + 3 (entries 6..11) of U = G g G^T (G as conv3x3_wino4.hip, double, rounded once); a part is
// 24 KB contiguous = the LDS image of a ring slot.
__global__ void conv3x3_wino4r_pack_kernel(const float* __restrict__ w, const int32_t* __restrict__ cin_map, int Cin,
                                           int Cin_phys, int Cout, int ncb, float* __restrict__ packed) {
    const size_t total = (size_t)(Cin_phys >> 4) * ncb * 36 * 512;
    const double G[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        const int i = (int)((idx >> 2) & 15);
        const int q = (int)((idx >> 6) & 3);
        const int ct = (int)((idx >> 8) & 1);
        size_t r = idx >> 9;
        const int slot = (int)(r % 36);              // 12 * part + entry
        r /= 36;
        const int cg = (int)(r % ncb);
        const int c16 = (int)(r / ncb);
        const int part = slot / 12, ent = slot % 12;
        const int xi = ent < 6 ? 6 * part + ent : 6 * (part + 3) + (ent - 6);
        const int cphys = c16 * 16 + 4 * q + e;
        const int clog = cin_map ? cin_map[cphys] : (cphys < Cin ? cphys : -1);
        const int co = cg * 32 + ct * 16 + i;
        double u = 0.;
        if (clog >= 0 && clog < Cin && co < Cout) {
            const int ua = xi / 6, ubb = xi % 6;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq)
                    u += G[ua][p] * G[ubb][qq] * (double)w[((size_t)(p * 3 + qq) * Cin + clog) * Cout + co];
        }
        packed[idx] = (float)u;
    }
}

extern "C" size_t pwc_conv3x3_wino4r_packed_floats(int Cin_phys, int Cout) {
    if (Cin_phys <= 0 || Cout <= 0) return 0;
    return (size_t)(Cin_phys >> 4) * ((Cout + 31) / 32) * 36 * 512;
}

extern "C" int pwc_conv3x3_wino4r_pack_f32(const float* w_hwio, const int32_t* cin_map, int Cin, int Cin_phys,
                                           int Cout, float* packed, pwc_stream_t stream) {
    if (!w_hwio || !packed || Cin <= 0 || Cout <= 0 || Cin_phys < Cin) return PWC_EINVAL;
    if (Cin_phys % 16) return PWC_EALIGN;
    const int ncb = (Cout + 31) / 32;
    const size_t total = (size_t)(Cin_phys >> 4) * ncb * 36 * 512;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_wino4r_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, cin_map,
                       Cin, Cin_phys, Cout, ncb, packed);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_wino4r_supported(int N, int H, int W, int Cin_phys, int Cout, int dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation < 1 || Cin_phys < 32 || Cin_phys > 1024 || (Cin_phys % 16) || Cout < 32 || (Cout % 32)) return 0;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    if (hs < 14 || ws < 28) return 0;
    const long blocks = (long)N * dilation * dilation * ((hs + 15) / 16) * ((ws + 31) / 32) * (Cout / 32);
    const double fill = (double)hs * ws / ((double)(((hs + 15) / 16) * 16) * (((ws + 31) / 32) * 32));
    return blocks >= 128 && fill >= 0.8 ? 1 : 0;
}

template <int ABL>
static int wino4r_launch(const Wino4rArgs& a, hipStream_t stream) {
    const int lds = WR_LDS + ((ABL & 128) ? 9216 : 0);
    static PwcDevOnce attr_once;   // the attribute is per device
    if (pwc_first_on_device(&attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4r_kernel<ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL((conv3x3_wino4r_kernel<ABL>), dim3((unsigned)a.ntiles), dim3(WR_T), lds, stream, a);
    return pwc_launch_status();
}

extern "C" int pwc_conv3x3_wino4r_f32(const float* x, int x_cs, const float* packed_u, const float* bias, float* y,
                                      int y_cs, int N, int H, int W, int Cin_phys, int Cout, int dilation,
                                      int apply_act, float slope, pwc_stream_t stream) {
    if (!x || !packed_u || !bias || !y) return PWC_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || Cin_phys <= 0 || Cout <= 0 || dilation < 1) return PWC_EINVAL;
    if (Cin_phys % 16 || Cout % 32) return PWC_EUNSUPPORTED;
    if (x_cs < Cin_phys || y_cs < Cout) return PWC_EINVAL;
    if ((x_cs & 3) || (y_cs & 3) || !pwc_aligned16(x) || !pwc_aligned16(y) || !pwc_aligned16(packed_u) || !pwc_aligned16(bias))
        return PWC_EALIGN;
    if ((long)H * W * x_cs * 4 >= (long)WR_OOB || (long)H * W * y_cs * 4 >= (long)WR_OOB) return PWC_ERANGE;
    if ((long)(Cin_phys >> 4) * (Cout / 32) * 36 * 2048 >= (long)WR_OOB) return PWC_ERANGE;
    Wino4rArgs a;
    a.x = x; a.up = packed_u; a.bias = bias; a.y = y; a.x_cs = x_cs; a.y_cs = y_cs;
    a.N = N; a.H = H; a.W = W; a.Cin_phys = Cin_phys; a.Cout = Cout; a.apply_act = apply_act; a.slope = slope;
    a.dil = dilation;
    const int hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
    a.tiles_x = (ws + 31) / 32; a.tiles_y = (hs + 15) / 16; a.ncb = Cout / 32;
    const long nblk = (long)N * dilation * dilation * a.tiles_x * a.tiles_y * a.ncb;
    if (nblk >= (1L << 31)) return PWC_ERANGE;
    a.ntiles = (int)nblk;
    return wino4r_launch<0>(a, (hipStream_t)stream);
}
