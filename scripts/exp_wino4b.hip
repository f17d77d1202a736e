// Microbenchmark + correctness harness for pwcnet_amd/csrc/conv3x3_wino4b.hip (F(4x4,3x3) on the bf16 matrix pipe with exact
// three-way operand splits) against the shipped fp32 kernels (conv3x3_wino4.hip, conv3x3_wino.hip) and a double-precision
// CPU convolution on sampled outputs.  Part 1: the arithmetic alone (one 16x16 tile, K-chain) -- fp32 MFMA chain against the
// split bf16 forms, errors against float64.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_wino4b.hip -o scripts/exp_wino4b.bin
#include "../pwcnet_amd/csrc/conv3x3_wino.hip"
#include "experiments/conv3x3_wino4b.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

template <typename F>
static float time_us(F&& f, int iters) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) f(i);
    (void)hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f(i);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

// ------------------------------------------------------------------ part 1: arithmetic of one 16 x 16 tile over K
// C[i][n] = sum_k U[i][k] V[k][n].  MODE 0: v_mfma_f32_16x16x4_f32 chain.  MODE 1: six products, three K=32 bf16 MFMAs per
// 16 k, small terms first (the kernel's order).  MODE 2: the same, large terms first.  MODE 3: + the two dropped 2^-24 terms
// (um vl + ul vm) as a fourth MFMA.  MODE 4: as 1 with the small terms (MFMA 2, 3) in a SECOND accumulator.  MODE 5: as 1
// with truncating instead of round-to-nearest splits.  MODE 6: hh + hm + mh only (what a two-term split would give).
__device__ __forceinline__ void split3(float x, bool trunc, __bf16& h, __bf16& m, __bf16& l) {
    if (trunc) {
        const float hf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
        const float r1 = x - hf;
        const float mf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
        const float r2 = r1 - mf;
        h = (__bf16)hf; m = (__bf16)mf; l = (__bf16)r2;
    } else {
        h = (__bf16)x;
        const float r1 = x - (float)h;
        m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        l = (__bf16)r2;
    }
}
template <int MODE>
__global__ void gemm_tile_kernel(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ C, int K) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const float* u = U + (size_t)tile * 16 * K;     // [16][K]
    const float* v = V + (size_t)tile * K * 16;     // [K][16]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[fr * K + k + fq], v[(k + fq) * 16 + fr], acc, 0, 0, 0);
    } else {
        const bool tr = MODE == 5;
        for (int k0 = 0; k0 < K; k0 += 16) {
            pwc_bf16x8 a1, a2, a3, a4, b1, b3, b4;
            const int f = fq & 1;
            for (int e = 0; e < 8; ++e) {
                const int ch = k0 + (e < 4 ? 4 * f + e : 8 + 4 * f + e - 4);
                __bf16 uh, um, ul, vh, vm, vl;
                split3(u[fr * K + ch], tr, uh, um, ul);
                split3(v[ch * 16 + fr], tr, vh, vm, vl);
                a1[e] = uh; a2[e] = um; a3[e] = fq < 2 ? ul : uh; a4[e] = fq < 2 ? um : ul;
                b1[e] = fq < 2 ? vh : vm; b3[e] = fq < 2 ? vh : vl; b4[e] = fq < 2 ? vl : vm;
            }
            if (MODE == 6) {
                pwc_bf16x8 z = {};
                for (int e = 0; e < 8; ++e) if (fq >= 2) a2[e] = z[e];       // keep um vh only
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
            } else if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);
            } else if (MODE == 4) {
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
            } else {
                if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a4, b4, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
            }
        }
        if (MODE == 4) acc += acc2;
    }
    for (int r = 0; r < 4; ++r) C[(size_t)tile * 256 + (fq * 4 + r) * 16 + fr] = acc[r];
}

static void arithmetic_part() {
    const int tiles = 256;
    struct Dist { const char* name; int kind; };
    const Dist dists[] = {{"U,V ~ N(0,1)", 0}, {"V = |N(0,1)| (post-activation), U ~ N(0,1)", 1}, {"V ~ N(0,1) x 10^uniform(-2,2), U ~ N(0,1)", 2},
                          {"V = 1 + 1e-3 N(0,1) (large mean), U = 1 + 1e-3 N", 3}};
    const int Ks[] = {64, 128, 160, 576, 1152};
    printf("# part 1: one 16x16 tile, K-chain; error against float64, relative to rms |C| (max over %d tiles x 256 entries / rms)\n", tiles);
    for (const Dist& ds : dists) {
        for (int K : Ks) {
            std::vector<float> hu((size_t)tiles * 16 * K), hv((size_t)tiles * K * 16);
            unsigned r = 12345 + K + 77 * ds.kind;
            auto uni = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) + 0.5f) / 16777216.f; };
            auto nrm = [&]() { const float a = uni(), b = uni(); return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b); };
            for (auto& x : hu) x = ds.kind == 3 ? 1.f + 1e-3f * nrm() : nrm();
            for (auto& x : hv) {
                const float g = nrm();
                x = ds.kind == 0 ? g : ds.kind == 1 ? fabsf(g) : ds.kind == 2 ? g * powf(10.f, 4.f * uni() - 2.f) : 1.f + 1e-3f * g;
            }
            std::vector<double> ref((size_t)tiles * 256);
            double rms = 0;
            for (int t = 0; t < tiles; ++t)
                for (int i = 0; i < 16; ++i)
                    for (int n = 0; n < 16; ++n) {
                        double s = 0;
                        for (int k = 0; k < K; ++k) s += (double)hu[((size_t)t * 16 + i) * K + k] * hv[((size_t)t * K + k) * 16 + n];
                        ref[(size_t)t * 256 + i * 16 + n] = s; rms += s * s;
                    }
            rms = sqrt(rms / ref.size());
            float *du, *dv, *dc;
            (void)hipMalloc(&du, hu.size() * 4); (void)hipMalloc(&dv, hv.size() * 4); (void)hipMalloc(&dc, ref.size() * 4);
            (void)hipMemcpy(du, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
            std::vector<float> hc(ref.size());
            double emax[7], erms[7];
            auto run = [&](int mode) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(gemm_tile_kernel<0>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 1: hipLaunchKernelGGL(gemm_tile_kernel<1>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 2: hipLaunchKernelGGL(gemm_tile_kernel<2>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 3: hipLaunchKernelGGL(gemm_tile_kernel<3>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 4: hipLaunchKernelGGL(gemm_tile_kernel<4>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 5: hipLaunchKernelGGL(gemm_tile_kernel<5>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    default: hipLaunchKernelGGL(gemm_tile_kernel<6>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                }
                (void)hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, sq = 0;
                for (size_t i = 0; i < hc.size(); ++i) { const double e = hc[i] - ref[i]; mx = fmax(mx, fabs(e)); sq += e * e; }
                emax[mode] = mx / rms; erms[mode] = sqrt(sq / hc.size()) / rms;
            };
            for (int m = 0; m < 7; ++m) run(m);
            printf("%-48s K=%4d | fp32 MFMA max %.2e rms %.2e | 6 products: small first max %.2e rms %.2e (x%.2f rms) ; large first x%.2f ; "
                   "8 products x%.2f ; two accumulators x%.2f ; truncating splits x%.2f ; 3 products (2 terms) x%.1f\n",
                   ds.name, K, emax[0], erms[0], emax[1], erms[1], erms[1] / erms[0], erms[2] / erms[0], erms[3] / erms[0],
                   erms[4] / erms[0], erms[5] / erms[0], erms[6] / erms[0]);
            (void)hipFree(du); (void)hipFree(dv); (void)hipFree(dc);
        }
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    if (only < 0 || only == 100) arithmetic_part();
    if (only == 100) return 0;
    struct Shape { int N, H, W, Cin, Cout, dil, xcs; float in_scale; };
    Shape shapes[] = {{8, 112, 256, 128, 128, 1, 128, 1.f}, {8, 112, 256, 160, 128, 1, 160, 1.f}, {8, 112, 256, 128, 96, 1, 128, 1.f},
                      {8, 112, 256, 96, 64, 1, 96, 1.f}, {8, 112, 256, 64, 32, 1, 64, 1.f}, {8, 112, 256, 128, 128, 2, 128, 1.f},
                      {8, 112, 256, 128, 128, 4, 128, 1.f}, {8, 112, 256, 128, 96, 8, 128, 1.f}, {8, 56, 128, 192, 128, 1, 192, 1.f},
                      {2, 50, 70, 64, 64, 1, 80, 1.f}, {1, 16, 32, 64, 64, 1, 64, 1.f}, {2, 112, 256, 128, 128, 1, 128, 300.f}};
    int idx = -1;
    for (auto sh : shapes) {
        ++idx;
        if (only >= 0 && idx != only) continue;
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        const int ycs = sh.Cout + 16;
        std::vector<float> hx(npix * sh.xcs), hw((size_t)9 * sh.Cin * sh.Cout), hb(sh.Cout);
        unsigned r = 4242 + idx;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; };
        // post-activation statistics: leaky-relu of a centred variable
        for (auto& v : hx) { const float g = 2.f * rnd(); v = sh.in_scale * (g > 0.f ? g : 0.1f * g); }
        const float wl = sqrtf(6.f / (9.f * (sh.Cin + sh.Cout)));
        for (auto& v : hw) v = 2.f * wl * rnd();
        for (auto& v : hb) v = 0.2f * rnd();
        float *x, *w, *b, *y2, *y4, *yb, *u2, *u4, *ub;
        (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
        (void)hipMalloc(&y2, npix * ycs * 4); (void)hipMalloc(&y4, npix * ycs * 4); (void)hipMalloc(&yb, npix * ycs * 4);
        (void)hipMalloc(&u2, pwc_conv3x3_wino_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMalloc(&u4, pwc_conv3x3_wino4_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMalloc(&ub, pwc_conv3x3_wino4b_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemset(y2, 0, npix * ycs * 4); (void)hipMemset(y4, 0, npix * ycs * 4); (void)hipMemset(yb, 0, npix * ycs * 4);
        int rc = pwc_conv3x3_wino_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u2, 0);
        rc |= pwc_conv3x3_wino4_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u4, 0);
        rc |= pwc_conv3x3_wino4b_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, ub, 0);
        const double gf = 2.0 * npix * 9.0 * sh.Cin * sh.Cout / 1e9;
        printf("== [%d] N=%d %dx%d Cin=%d (cs %d) Cout=%d d=%d input scale %.0f: %.1f GFLOP (direct), wino4 supported=%d wino4b supported=%d, pack rc %d\n",
               idx, sh.N, sh.H, sh.W, sh.Cin, sh.xcs, sh.Cout, sh.dil, sh.in_scale, gf,
               pwc_conv3x3_wino4_supported(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil),
               pwc_conv3x3_wino4b_supported(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil), rc);
        rc = pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        int rc4 = pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        int rcb = pwc_conv3x3_wino4b_f32(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        (void)hipDeviceSynchronize();
        printf("  launch rc: F(2x2) %d, F(4x4) %d, F(4x4) bf16x3 %d; hip: %s\n", rc, rc4, rcb, hipGetErrorString(hipGetLastError()));
        std::vector<float> h2(npix * ycs), h4(npix * ycs), hbb(npix * ycs);
        (void)hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h4.data(), y4, h4.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hbb.data(), yb, hbb.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0, mpad = 0; size_t bad = 0, nan = 0;
        size_t hy[16] = {0}, hxm[32] = {0}, hc[8] = {0};
        const double tol = 2e-4 * sh.in_scale;
        for (size_t p = 0; p < npix; ++p) {
            for (int c = 0; c < sh.Cout; ++c) {
                const double a = hbb[p * ycs + c], e = h4[p * ycs + c];
                if (a != a) { ++nan; continue; }
                md = fmax(md, fabs(a - e)); mx = fmax(mx, fabs(e));
                if (fabs(a - e) > tol) {
                    if (bad < 6) printf("    mismatch n %zu y %zu x %zu c %d: bf16x3 %.6f fp32 F(4x4) %.6f\n", p / ((size_t)sh.H * sh.W), (p / sh.W) % sh.H, p % sh.W, c, a, e);
                    ++bad; ++hy[((p / sh.W) % sh.H) & 15]; ++hxm[(p % sh.W) & 31]; ++hc[(c >> 2) & 7];
                }
            }
            for (int c = sh.Cout; c < ycs; ++c) mpad = fmax(mpad, fabs((double)hbb[p * ycs + c]));
        }
        if (bad) {
            printf("    bad by y%%16: "); for (int i = 0; i < 16; ++i) printf("%zu ", hy[i]);
            printf("\n    bad by x%%32: "); for (int i = 0; i < 32; ++i) printf("%zu ", hxm[i]);
            printf("\n    bad by (c/4)%%8: "); for (int i = 0; i < 8; ++i) printf("%zu ", hc[i]);
            printf("\n");
        }
        // double-precision direct convolution on sampled outputs: errors of the three kernels
        double e2 = 0, e4 = 0, eb = 0, s2 = 0, s4 = 0, sb = 0, sv = 0;
        unsigned rs = 99;
        const int NS = 3000;
        for (int s = 0; s < NS; ++s) {
            rs = rs * 1664525u + 1013904223u; const size_t p = (rs >> 4) % npix;
            rs = rs * 1664525u + 1013904223u; const int co = (rs >> 4) % sh.Cout;
            const int n = (int)(p / ((size_t)sh.H * sh.W)), yy = (int)((p / sh.W) % sh.H), xx = (int)(p % sh.W);
            double acc = hb[co];
            for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
                const int sy = yy + (ty - 1) * sh.dil, sx = xx + (tx - 1) * sh.dil;
                if (sy < 0 || sy >= sh.H || sx < 0 || sx >= sh.W) continue;
                const float* xp = &hx[(((size_t)n * sh.H + sy) * sh.W + sx) * sh.xcs];
                for (int ci = 0; ci < sh.Cin; ++ci) acc += (double)xp[ci] * hw[((size_t)(ty * 3 + tx) * sh.Cin + ci) * sh.Cout + co];
            }
            acc = fmax(acc, 0.1 * acc);
            const double d2 = h2[p * ycs + co] - acc, d4 = h4[p * ycs + co] - acc, db = hbb[p * ycs + co] - acc;
            e2 = fmax(e2, fabs(d2)); e4 = fmax(e4, fabs(d4)); eb = fmax(eb, fabs(db));
            s2 += d2 * d2; s4 += d4 * d4; sb += db * db; sv += acc * acc;
        }
        printf("  bf16x3 vs fp32 F(4x4): max |diff| %.3e (max |value| %.3f), %zu entries > %.0e, %zu NaN; channels beyond Cout max %.1e\n", md, mx, bad, tol, nan, mpad);
        printf("  NUMERICS vs float64 direct conv (%d samples, rms |y| %.3e): F(2x2) fp32 max %.3e rms %.3e | F(4x4) fp32 max %.3e rms %.3e | F(4x4) bf16x3 max %.3e rms %.3e  (bf16x3 / fp32 F(4x4): max x%.2f rms x%.2f)\n",
               NS, sqrt(sv / NS), e2, sqrt(s2 / NS), e4, sqrt(s4 / NS), eb, sqrt(sb / NS), eb / e4, sqrt(sb / s4));
        fflush(stdout);
        for (int round = 0; round < 2; ++round) {
            const float t2 = time_us([&](int) { pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            const float t4 = time_us([&](int) { pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            const float tb = time_us([&](int) { pwc_conv3x3_wino4b_f32(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            printf("  F(2x2) %8.1f us %6.1f TF | F(4x4) fp32 %8.1f us %6.1f TF | F(4x4) bf16x3 %8.1f us %6.1f TF (direct-conv flops)  x%.2f vs fp32 F(4x4)\n",
                   t2, gf / t2 * 1e3, t4, gf / t4 * 1e3, tb, gf / tb * 1e3, t4 / tb);
            fflush(stdout);
        }
        if (idx == 0 || idx == 3) {
            Wino4bArgs a{};
            a.x = x; a.up = ub; a.bias = b; a.y = yb; a.x_cs = sh.xcs; a.y_cs = ycs; a.N = sh.N; a.H = sh.H; a.W = sh.W;
            a.Cin_phys = sh.Cin; a.Cout = sh.Cout; a.apply_act = 1; a.slope = 0.1f; a.dil = 1;
            a.tiles_x = (sh.W + 31) / 32; a.tiles_y = (sh.H + 15) / 16; a.ncb = sh.Cout / 32;
            a.ntiles = sh.N * a.tiles_x * a.tiles_y * a.ncb;
            printf("  ablations: no patch DMA %.1f | no weight DMA %.1f | no DMA %.1f | no MFMA %.1f | no split/exchange %.1f | no transform, no split %.1f | "
                   "MFMA + LDS reads only (no DMA, transform, split) %.1f | DMA + LDS only (no MFMA, transform, split) %.1f us\n",
                   time_us([&](int) { wino4b_launch<1>(a, 0); }, 10), time_us([&](int) { wino4b_launch<2>(a, 0); }, 10),
                   time_us([&](int) { wino4b_launch<3>(a, 0); }, 10), time_us([&](int) { wino4b_launch<4>(a, 0); }, 10),
                   time_us([&](int) { wino4b_launch<32>(a, 0); }, 10), time_us([&](int) { wino4b_launch<32 | 64>(a, 0); }, 10),
                   time_us([&](int) { wino4b_launch<3 | 32 | 64>(a, 0); }, 10), time_us([&](int) { wino4b_launch<4 | 32 | 64>(a, 0); }, 10));
            fflush(stdout);
            // timeline of workgroup 0 (s_memtime stamps, 100 MHz ticks -> printed raw): once alone on the GPU, once in the full launch
            for (int full = 0; full < 2; ++full) {
                Wino4bArgs b2 = a;
                if (!full) { b2.N = 1; b2.ntiles = 1; }
                (void)hipMemset(yb, 0, 8 * 144 * 8);
                wino4b_launch<128>(b2, 0);
                (void)hipDeviceSynchronize();
                std::vector<unsigned long long> st(8 * 144);
                (void)hipMemcpy(st.data(), yb, st.size() * 8, hipMemcpyDeviceToHost);
                printf("  timeline (%s), per wave: per phase [vm-wait, barrier, dma-issue, part] in ticks; stamps are s_memtime\n", full ? "full launch, block 0" : "one workgroup alone");
                const int nph = 3 * (sh.Cin / 16);
                for (int wv = 0; wv < 8; wv += 1) {
                    const unsigned long long* q = &st[wv * 144];
                    printf("    wave %d total %llu :", wv, q[4 * nph + 1] - q[0]);
                    for (int ph = 0; ph < nph && ph < 9; ++ph)
                        printf(" [%llu %llu %llu %llu]", q[4 * ph + 1] - q[4 * ph], q[4 * ph + 2] - q[4 * ph + 1], q[4 * ph + 3] - q[4 * ph + 2], q[4 * ph + 4] - q[4 * ph + 3]);
                    printf(" ... epilogue %llu\n", q[4 * nph + 1] - q[4 * nph]);
                }
            }
            fflush(stdout);
        }
        (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(y2); (void)hipFree(y4); (void)hipFree(yb);
        (void)hipFree(u2); (void)hipFree(u4); (void)hipFree(ub);
    }
    return 0;
}
