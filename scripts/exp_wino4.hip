// Microbenchmark + correctness harness: Winograd F(4x4,3x3) (pwcnet_amd/csrc/conv3x3_wino4.hip) against the shipped
// F(2x2,3x3) kernel (conv3x3_wino.hip) and a double-precision CPU convolution on sampled pixels.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_wino4.hip -o scripts/exp_wino4.bin
#include "../pwcnet_amd/csrc/conv3x3_wino.hip"
#include "../pwcnet_amd/csrc/conv3x3_wino4.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

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

int main(int argc, char** argv) {
    struct Shape { int N, H, W, Cin, Cout, dil, xcs; };
    Shape shapes[] = {{8, 112, 256, 128, 128, 1, 128}, {8, 112, 256, 160, 128, 1, 160}, {8, 112, 256, 128, 96, 1, 128},
                      {8, 112, 256, 96, 64, 1, 96}, {8, 112, 256, 128, 128, 2, 128}, {8, 112, 256, 128, 128, 4, 128},
                      {8, 112, 256, 128, 96, 8, 128}, {2, 50, 70, 64, 64, 1, 80}, {1, 16, 32, 64, 64, 1, 64}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int idx = -1;
    for (auto sh : shapes) {
        ++idx;
        if (only >= 0 && idx != only) continue;
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        const int ycs = sh.Cout + 16;
        std::vector<float> hx(npix * sh.xcs), hw((size_t)9 * sh.Cin * sh.Cout), hb(sh.Cout);
        unsigned r = 4242 + idx;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; };
        for (auto& v : hx) v = rnd();
        const float wl = sqrtf(6.f / (9.f * (sh.Cin + sh.Cout)));
        for (auto& v : hw) v = 2.f * wl * rnd();
        for (auto& v : hb) v = 0.2f * rnd();
        float *x, *w, *b, *y2, *y4, *u2, *u4;
        (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
        (void)hipMalloc(&y2, npix * ycs * 4); (void)hipMalloc(&y4, npix * ycs * 4);
        (void)hipMalloc(&u2, pwc_conv3x3_wino_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMalloc(&u4, pwc_conv3x3_wino4_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemset(y2, 0, npix * ycs * 4); (void)hipMemset(y4, 0, npix * ycs * 4);
        int rc = pwc_conv3x3_wino_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u2, 0);
        rc |= pwc_conv3x3_wino4_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u4, 0);
        const double gf = 2.0 * npix * 9.0 * sh.Cin * sh.Cout / 1e9;
        printf("== N=%d %dx%d Cin=%d (cs %d) Cout=%d d=%d : %.1f GFLOP (direct), supported=%d, pack rc %d\n", sh.N, sh.H, sh.W, sh.Cin,
               sh.xcs, sh.Cout, sh.dil, gf, pwc_conv3x3_wino4_supported(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil), rc);
        rc = pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        int rc4 = pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        (void)hipDeviceSynchronize();
        printf("  launch rc: F(2x2) %d, F(4x4) %d; hip: %s\n", rc, rc4, hipGetErrorString(hipGetLastError()));
        std::vector<float> h2(npix * ycs), h4(npix * ycs);
        (void)hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h4.data(), y4, h4.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0, mpad = 0; size_t bad = 0, nan = 0;
        size_t hy[16] = {0}, hxm[32] = {0}, hc[8] = {0};
        for (size_t p = 0; p < npix; ++p) {
            for (int c = 0; c < sh.Cout; ++c) {
                const double a = h4[p * ycs + c], e = h2[p * ycs + c];
                if (a != a) { ++nan; continue; }
                md = fmax(md, fabs(a - e)); mx = fmax(mx, fabs(e));
                if (fabs(a - e) > 2e-4) {
                    if (bad < 6) printf("    mismatch n %zu y %zu x %zu c %d: F(4x4) %.6f F(2x2) %.6f\n", p / ((size_t)sh.H * sh.W), (p / sh.W) % sh.H, p % sh.W, c, a, e);
                    ++bad; ++hy[((p / sh.W) % sh.H) & 15]; ++hxm[(p % sh.W) & 31]; ++hc[(c >> 2) & 7];
                }
            }
            for (int c = sh.Cout; c < ycs; ++c) mpad = fmax(mpad, fabs((double)h4[p * ycs + c]));
        }
        if (bad) {
            printf("    bad by y%%16: "); for (int i = 0; i < 16; ++i) printf("%zu ", hy[i]);
            printf("\n    bad by x%%32: "); for (int i = 0; i < 32; ++i) printf("%zu ", hxm[i]);
            printf("\n    bad by (c/4)%%8: "); for (int i = 0; i < 8; ++i) printf("%zu ", hc[i]);
            printf("\n");
        }
        // double-precision direct convolution on sampled outputs: errors of both kernels
        double e2 = 0, e4 = 0;
        unsigned rs = 99;
        for (int s = 0; s < 400; ++s) {
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
            e2 = fmax(e2, fabs(h2[p * ycs + co] - acc)); e4 = fmax(e4, fabs(h4[p * ycs + co] - acc));
        }
        printf("  F(4x4) vs F(2x2): max |diff| %.3e (max |value| %.3f), %zu entries > 2e-4, %zu NaN; channels beyond Cout max %.1e\n", md, mx, bad, nan, mpad);
        printf("  vs double-precision direct conv (400 samples): F(2x2) max err %.3e, F(4x4) max err %.3e\n", e2, e4);
        fflush(stdout);
        for (int round = 0; round < 2; ++round) {
            const float t2 = time_us([&](int) { pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            const float t4 = time_us([&](int) { pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            printf("  F(2x2) %8.1f us  %6.1f TFLOP/s (direct-conv flops)   |   F(4x4) %8.1f us  %6.1f TFLOP/s   x%.2f\n", t2, gf / t2 * 1e-3 * 1e3, t4,
                   gf / t4 * 1e-3 * 1e3, t2 / t4);
            fflush(stdout);
        }
        if (idx == 0) {
            Wino4Args a{};
            a.x = x; a.up = u4; a.bias = b; a.y = y4; a.x_cs = sh.xcs; a.y_cs = ycs; a.N = sh.N; a.H = sh.H; a.W = sh.W;
            a.Cin_phys = sh.Cin; a.Cout = sh.Cout; a.apply_act = 1; a.slope = 0.1f; a.dil = 1;
            a.tiles_x = (sh.W + 31) / 32; a.tiles_y = (sh.H + 15) / 16; a.ncb = sh.Cout / 16;
            a.ntiles = sh.N * a.tiles_x * a.tiles_y * a.ncb;
            printf("  ablations: no patch DMA %.1f us | no weight DMA %.1f us | no DMA %.1f us | no MFMA %.1f us | no MFMA, no transform %.1f us | no transform %.1f us\n",
                   time_us([&](int) { wino4_launch<1>(a, 0); }, 10), time_us([&](int) { wino4_launch<2>(a, 0); }, 10),
                   time_us([&](int) { wino4_launch<3>(a, 0); }, 10), time_us([&](int) { wino4_launch<4>(a, 0); }, 10),
                   time_us([&](int) { wino4_launch<4 | 64>(a, 0); }, 10), time_us([&](int) { wino4_launch<64>(a, 0); }, 10));
            printf("  same instructions, fetches from hot lines: patch %.1f us | weights %.1f us | both %.1f us | both, no MFMA no transform %.1f us\n",
                   time_us([&](int) { wino4_launch<8>(a, 0); }, 10), time_us([&](int) { wino4_launch<16>(a, 0); }, 10),
                   time_us([&](int) { wino4_launch<24>(a, 0); }, 10), time_us([&](int) { wino4_launch<24 | 4 | 64>(a, 0); }, 10));
        }
        (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(y2); (void)hipFree(y4); (void)hipFree(u2); (void)hipFree(u4);
    }
    return 0;
}
