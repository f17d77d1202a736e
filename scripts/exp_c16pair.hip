// Harness for pwcnet_amd/csrc/conv3x3_c16pair.hip (two chained 16 -> 16 3x3 convolutions in one launch, F16 matrix pipe with
// exact operand splits) against two launches of the fp32 Winograd F(2x2) kernel and a float64 convolution pair on samples.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_c16pair.hip -o scripts/exp_c16pair.bin
#include "../pwcnet_amd/csrc/conv3x3_wino.hip"
#include "../pwcnet_amd/csrc/conv3x3_direct.hip"
#include "../pwcnet_amd/csrc/conv3x3_c16pair.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

template <class F>
static float time_us(F&& f, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f(i);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    struct Shape { int N, H, W; };
    Shape shapes[] = {{16, 224, 512}, {2, 50, 70}, {1, 16, 32}, {3, 100, 130}, {4, 224, 512}};
    for (auto sh : shapes) {
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        std::vector<float> hx(npix * 16), hw1(9 * 256), hw2(9 * 256), hb1(16), hb2(16);
        unsigned r = 777 + sh.H;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; };
        for (auto& v : hx) { const float g = 2.f * rnd(); v = g > 0.f ? g : 0.1f * g; }
        const float wl = sqrtf(6.f / (9.f * 32));
        for (auto& v : hw1) v = 2.f * wl * rnd();
        for (auto& v : hw2) v = 2.f * wl * rnd();
        for (auto& v : hb1) v = 0.2f * rnd();
        for (auto& v : hb2) v = 0.2f * rnd();
        float *x, *w1, *w2, *b1, *b2, *ym, *yr, *yf, *u1, *u2, *up;
        (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&w1, 9 * 256 * 4); (void)hipMalloc(&w2, 9 * 256 * 4);
        (void)hipMalloc(&b1, 64); (void)hipMalloc(&b2, 64);
        (void)hipMalloc(&ym, npix * 64); (void)hipMalloc(&yr, npix * 64); (void)hipMalloc(&yf, npix * 64);
        (void)hipMalloc(&u1, pwc_conv3x3_wino_packed_floats(16, 16) * 4); (void)hipMalloc(&u2, pwc_conv3x3_wino_packed_floats(16, 16) * 4);
        (void)hipMalloc(&up, pwc_conv3x3_c16pair_packed_floats() * 4);
        (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(w1, hw1.data(), 9 * 256 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(w2, hw2.data(), 9 * 256 * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b1, hb1.data(), 64, hipMemcpyHostToDevice); (void)hipMemcpy(b2, hb2.data(), 64, hipMemcpyHostToDevice);
        (void)hipMemset(yf, 0, npix * 64);
        int rc = pwc_conv3x3_wino_pack_f32(w1, nullptr, 16, 16, 16, u1, 0) | pwc_conv3x3_wino_pack_f32(w2, nullptr, 16, 16, 16, u2, 0) |
                 pwc_conv3x3_c16pair_pack_f32(w1, w2, up, 0);
        auto ref = [&]() {
            pwc_conv3x3_wino_f32(x, 16, u1, b1, ym, 16, sh.N, sh.H, sh.W, 16, 16, 1, 1, 0.1f, 0);
            pwc_conv3x3_wino_f32(ym, 16, u2, b2, yr, 16, sh.N, sh.H, sh.W, 16, 16, 1, 1, 0.1f, 0);
        };
        ref();
        const int rcf = pwc_conv3x3_c16pair_f32(x, 16, up, b1, b2, yf, 16, sh.N, sh.H, sh.W, 0.1f, 0);
        (void)hipDeviceSynchronize();
        printf("== N=%d %dx%d: pack rc %d, launch rc %d, hip %s, supported %d\n", sh.N, sh.H, sh.W, rc, rcf, hipGetErrorString(hipGetLastError()),
               pwc_conv3x3_c16pair_supported(sh.N, sh.H, sh.W));
        std::vector<float> hr(npix * 16), hf(npix * 16);
        (void)hipMemcpy(hr.data(), yr, hr.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hf.data(), yf, hf.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0; size_t bad = 0, nan = 0;
        for (size_t i = 0; i < hr.size(); ++i) {
            if (hf[i] != hf[i]) { ++nan; continue; }
            const double dlt = fabs((double)hf[i] - hr[i]); md = fmax(md, dlt); mx = fmax(mx, fabs((double)hr[i]));
            if (dlt > 2e-4) { if (bad < 5) printf("    mismatch pixel %zu (y %zu x %zu) c %zu: fused %.6f two launches %.6f\n", i / 16, (i / 16 / sh.W) % sh.H, (i / 16) % sh.W, i % 16, hf[i], hr[i]); ++bad; }
        }
        printf("  fused vs two fp32 F(2x2) launches: max |diff| %.3e (max |value| %.3f), %zu entries > 2e-4, %zu NaN\n", md, mx, bad, nan);
        // float64 pair on samples
        auto xat = [&](int n, int y, int xx, int c) -> double { return (y < 0 || y >= sh.H || xx < 0 || xx >= sh.W) ? 0.0 : hx[(((size_t)n * sh.H + y) * sh.W + xx) * 16 + c]; };
        double ef = 0, er = 0, sf = 0, sr = 0; unsigned rs = 5;
        const int NS = 400;
        for (int s = 0; s < NS; ++s) {
            rs = rs * 1664525u + 1013904223u; const size_t p = (rs >> 4) % npix;
            rs = rs * 1664525u + 1013904223u; const int co = (rs >> 4) % 16;
            const int n = (int)(p / ((size_t)sh.H * sh.W)), yy = (int)((p / sh.W) % sh.H), xx = (int)(p % sh.W);
            double acc2 = hb2[co];
            for (int t2 = 0; t2 < 9; ++t2) {
                const int my = yy + t2 / 3 - 1, mxx = xx + t2 % 3 - 1;
                if (my < 0 || my >= sh.H || mxx < 0 || mxx >= sh.W) continue;
                for (int cm = 0; cm < 16; ++cm) {
                    double a1 = hb1[cm];
                    for (int t1 = 0; t1 < 9; ++t1) for (int ci = 0; ci < 16; ++ci) a1 += xat(n, my + t1 / 3 - 1, mxx + t1 % 3 - 1, ci) * hw1[(t1 * 16 + ci) * 16 + cm];
                    a1 = fmax(a1, 0.1 * a1);
                    a1 = (double)(float)a1;                       // the intermediate is an fp32 tensor in both paths
                    acc2 += a1 * hw2[(t2 * 16 + cm) * 16 + co];
                }
            }
            acc2 = fmax(acc2, 0.1 * acc2);
            const double df = hf[p * 16 + co] - acc2, dr = hr[p * 16 + co] - acc2;
            ef = fmax(ef, fabs(df)); er = fmax(er, fabs(dr)); sf += df * df; sr += dr * dr;
        }
        printf("  vs float64 pair (%d samples): fused max %.3e rms %.3e | two F(2x2) launches max %.3e rms %.3e\n", NS, ef, sqrt(sf / NS), er, sqrt(sr / NS));
        std::vector<float> ta, tb;
        for (int rep = 0; rep < 7; ++rep) {
            ta.push_back(time_us([&](int) { pwc_conv3x3_c16pair_f32(x, 16, up, b1, b2, yf, 16, sh.N, sh.H, sh.W, 0.1f, 0); }, 10));
            tb.push_back(time_us([&](int) { ref(); }, 10));
        }
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("  medians of 7 interleaved rounds: fused %.1f us (min %.1f), two F(2x2) launches %.1f us (min %.1f)\n", ta[3], ta[0], tb[3], tb[0]);
        if (sh.N == 16) {
            auto ab = [&](auto tag) { return time_us([&](int) { c16pair_run<decltype(tag)::value>(x, 16, up, b1, b2, yf, 16, sh.N, sh.H, sh.W, 0.1f, 0); }, 10); };
            printf("  ablations: no patch DMA %.1f | no split %.1f | no layer 1 %.1f | no layer 2 %.1f | no layers %.1f | no DMA, no split %.1f | nothing %.1f us\n",
                   ab(std::integral_constant<int, 1>{}), ab(std::integral_constant<int, 2>{}), ab(std::integral_constant<int, 4>{}), ab(std::integral_constant<int, 8>{}),
                   ab(std::integral_constant<int, 12>{}), ab(std::integral_constant<int, 3>{}), ab(std::integral_constant<int, 15>{}));
        }
        (void)hipFree(x); (void)hipFree(w1); (void)hipFree(w2); (void)hipFree(b1); (void)hipFree(b2); (void)hipFree(ym); (void)hipFree(yr); (void)hipFree(yf);
        (void)hipFree(u1); (void)hipFree(u2); (void)hipFree(up);
    }
    {   // ---- all of pyramid level 1 from the raw frames: 8 + 8 images of 448 x 1024 x 3
        const int Na = 8, Nb = 8, H0 = 448, W0 = 1024, H = 224, W = 512;
        const size_t nraw = (size_t)Na * H0 * W0 * 3, npix = (size_t)(Na + Nb) * H * W;
        std::vector<float> ha(nraw), hb(nraw), hw0(27 * 16), hw1(9 * 256), hw2(9 * 256), hb0(16), hb1(16), hb2(16);
        unsigned r = 4321;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; };
        for (auto& v : ha) v = rnd() + 0.5f;
        for (auto& v : hb) v = rnd() + 0.5f;
        for (auto& v : hw0) v = 2.f * sqrtf(6.f / (9.f * 19)) * rnd();
        for (auto& v : hw1) v = 2.f * sqrtf(6.f / (9.f * 32)) * rnd();
        for (auto& v : hw2) v = 2.f * sqrtf(6.f / (9.f * 32)) * rnd();
        for (auto& v : hb0) v = 0.2f * rnd();
        for (auto& v : hb1) v = 0.2f * rnd();
        for (auto& v : hb2) v = 0.2f * rnd();
        float *xa, *xb, *w0, *w1, *w2, *b0, *b1, *b2, *y0, *yr, *yf, *up, *up3;
        (void)hipMalloc(&xa, nraw * 4); (void)hipMalloc(&xb, nraw * 4); (void)hipMalloc(&w0, 27 * 64); (void)hipMalloc(&w1, 9 * 1024); (void)hipMalloc(&w2, 9 * 1024);
        (void)hipMalloc(&b0, 64); (void)hipMalloc(&b1, 64); (void)hipMalloc(&b2, 64);
        (void)hipMalloc(&y0, npix * 64); (void)hipMalloc(&yr, npix * 64); (void)hipMalloc(&yf, npix * 64);
        (void)hipMalloc(&up, pwc_conv3x3_c16pair_packed_floats() * 4); (void)hipMalloc(&up3, pwc_conv3x3_c3c16pair_packed_floats() * 4);
        (void)hipMemcpy(xa, ha.data(), nraw * 4, hipMemcpyHostToDevice); (void)hipMemcpy(xb, hb.data(), nraw * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(w0, hw0.data(), 27 * 64, hipMemcpyHostToDevice); (void)hipMemcpy(w1, hw1.data(), 9 * 1024, hipMemcpyHostToDevice);
        (void)hipMemcpy(w2, hw2.data(), 9 * 1024, hipMemcpyHostToDevice);
        (void)hipMemcpy(b0, hb0.data(), 64, hipMemcpyHostToDevice); (void)hipMemcpy(b1, hb1.data(), 64, hipMemcpyHostToDevice); (void)hipMemcpy(b2, hb2.data(), 64, hipMemcpyHostToDevice);
        int rc = pwc_conv3x3_c16pair_pack_f32(w1, w2, up, 0) | pwc_conv3x3_c3c16pair_pack_f32(w0, w1, w2, up3, 0);
        auto ref = [&]() {
            pwc_conv3x3_direct_f32(xa, 3, w0, b0, y0, 16, nullptr, 0, Na, H0, W0, 3, 16, 2, 1, 1, 0.1f, 0);
            pwc_conv3x3_direct_f32(xb, 3, w0, b0, y0 + (size_t)Na * H * W * 16, 16, nullptr, 0, Nb, H0, W0, 3, 16, 2, 1, 1, 0.1f, 0);
            pwc_conv3x3_c16pair_f32(y0, 16, up, b1, b2, yr, 16, Na + Nb, H, W, 0.1f, 0);
        };
        ref();
        const int rcf = pwc_conv3x3_c3c16pair_f32(xa, Na, xb, Nb, up3, b0, b1, b2, yf, 16, H0, W0, 0.1f, 0);
        (void)hipDeviceSynchronize();
        printf("== level 1 from the raw frames, %d + %d x %dx%dx3: pack rc %d, launch rc %d, hip %s, supported %d\n", Na, Nb, H0, W0, rc, rcf,
               hipGetErrorString(hipGetLastError()), pwc_conv3x3_c3c16pair_supported(Na + Nb, H0, W0));
        std::vector<float> hr(npix * 16), hf(npix * 16);
        (void)hipMemcpy(hr.data(), yr, hr.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hf.data(), yf, hf.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0; size_t nan = 0;
        for (size_t i = 0; i < hr.size(); ++i) { if (hf[i] != hf[i]) { ++nan; continue; } md = fmax(md, fabs((double)hf[i] - hr[i])); mx = fmax(mx, fabs((double)hr[i])); }
        printf("  three-layer launch vs fp32 stride-2 launches + pair launch: max |diff| %.3e (max |value| %.3f), %zu NaN\n", md, mx, nan);
        std::vector<float> ta, tb;
        for (int rep = 0; rep < 7; ++rep) {
            ta.push_back(time_us([&](int) { pwc_conv3x3_c3c16pair_f32(xa, Na, xb, Nb, up3, b0, b1, b2, yf, 16, H0, W0, 0.1f, 0); }, 10));
            tb.push_back(time_us([&](int) { ref(); }, 10));
        }
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("  medians of 7 interleaved rounds: three-layer launch %.1f us (min %.1f), three launches %.1f us (min %.1f)\n", ta[3], ta[0], tb[3], tb[0]);
        auto ab = [&](auto tag) { return time_us([&](int) { c3c16pair_run<decltype(tag)::value>(xa, Na, xb, Nb, up3, b0, b1, b2, yf, 16, H0, W0, 0.1f, 0); }, 10); };
        printf("  ablations: no patch DMA %.1f | no stride-2 layer %.1f | no layer 1 %.1f | no layer 2 %.1f | no layers 1, 2 %.1f | no DMA, no stride-2 layer %.1f | nothing %.1f us\n",
               ab(std::integral_constant<int, 1>{}), ab(std::integral_constant<int, 2>{}), ab(std::integral_constant<int, 4>{}), ab(std::integral_constant<int, 8>{}),
               ab(std::integral_constant<int, 12>{}), ab(std::integral_constant<int, 3>{}), ab(std::integral_constant<int, 15>{}));
    }
    return 0;
}
