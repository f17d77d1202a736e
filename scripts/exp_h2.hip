// Microbenchmark + correctness harness for scripts/experiments/conv3x3_h2.hip (direct 3x3 convolution on the F16 matrix pipe
// with scaled two-term operand splits) against the shipped fp32 kernels (conv3x3_wino4.hip, conv3x3_wino.hip) and a
// double-precision CPU convolution on sampled outputs.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_h2.hip -o scripts/exp_h2.bin
#define PWC_HARNESS 1
#include "../pwcnet_amd/csrc/conv3x3_wino.hip"
#include "../pwcnet_amd/csrc/conv3x3_h2.hip"
#include <cstdio>
#include <algorithm>
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

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    struct Shape { int N, H, W, Cin, Cout, dil, xcs; float in_scale; };
    Shape shapes[] = {{8, 112, 256, 128, 128, 1, 128, 1.f}, {8, 112, 256, 160, 128, 1, 160, 1.f}, {8, 112, 256, 128, 96, 1, 128, 1.f},
                      {8, 112, 256, 96, 64, 1, 96, 1.f}, {8, 112, 256, 64, 32, 1, 64, 1.f}, {8, 112, 256, 128, 128, 2, 128, 1.f},
                      {8, 112, 256, 128, 128, 4, 128, 1.f}, {8, 112, 256, 128, 96, 8, 128, 1.f}, {8, 56, 128, 192, 128, 1, 192, 1.f},
                      {2, 50, 70, 64, 64, 1, 80, 1.f}, {1, 16, 32, 64, 64, 1, 64, 1.f}, {2, 112, 256, 128, 128, 1, 128, 300.f},
                      {16, 112, 256, 32, 32, 1, 32, 1.f}, {16, 56, 128, 64, 64, 1, 64, 1.f}, {16, 28, 64, 96, 96, 1, 96, 1.f}, {8, 56, 128, 128, 96, 1, 128, 1.f},
                      {8, 56, 128, 96, 64, 1, 96, 1.f}, {8, 56, 128, 64, 32, 1, 64, 1.f}, {8, 112, 256, 48, 128, 1, 48, 1.f}, {8, 28, 64, 192, 128, 1, 192, 1.f},
                      {8, 112, 256, 96, 64, 16, 96, 1.f}, {8, 28, 64, 128, 128, 1, 128, 1.f}, {8, 28, 64, 128, 96, 1, 128, 1.f}, {8, 28, 64, 96, 64, 1, 96, 1.f},
                      {8, 28, 64, 64, 32, 1, 64, 1.f}, {4, 28, 64, 192, 128, 1, 192, 1.f}};
    const size_t WSF = (size_t)304 * 32768;
    float* wsp; (void)hipMalloc(&wsp, WSF * 4); (void)hipMemset(wsp, 0xFF, WSF * 4);
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
        (void)hipMalloc(&ub, pwc_conv3x3_h2_packed_floats(sh.Cin, sh.Cout) * 4);
        (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemset(y2, 0, npix * ycs * 4); (void)hipMemset(y4, 0, npix * ycs * 4); (void)hipMemset(yb, 0, npix * ycs * 4);
        int rc = pwc_conv3x3_wino_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u2, 0);
        rc |= pwc_conv3x3_wino4_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, u4, 0);
        rc |= pwc_conv3x3_h2_pack_f32(w, nullptr, sh.Cin, sh.Cin, sh.Cout, ub, 0);
        const double gf = 2.0 * npix * 9.0 * sh.Cin * sh.Cout / 1e9;
        printf("== [%d] N=%d %dx%d Cin=%d (cs %d) Cout=%d d=%d input scale %.0f: %.1f GFLOP (direct), wino4 supported=%d h2 supported=%d, pack rc %d\n",
               idx, sh.N, sh.H, sh.W, sh.Cin, sh.xcs, sh.Cout, sh.dil, sh.in_scale, gf,
               pwc_conv3x3_wino4_supported(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil),
               pwc_conv3x3_h2_supported(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil), rc);
        { long nb = 0; const int pv = h2_plan(sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, &nb); printf("  plan: variant %d, %ld workgroups\n", pv, nb); if (!pv) continue; }
        rc = pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        int rc4 = pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0);
        int rcb = pwc_conv3x3_h2_f32(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, wsp, WSF, 0);
        (void)hipDeviceSynchronize();
        {
            std::vector<unsigned> hws(WSF);
            (void)hipMemcpy(hws.data(), wsp, WSF * 4, hipMemcpyDeviceToHost);
            size_t dirty = 0, byq[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < WSF; ++i) if (hws[i] != 0xFFFFFFFFu) { ++dirty; ++byq[(i / 16) & 3]; }
            printf("  workspace after the launch: %zu words are not the sentinel (by 64-byte quarter of 256 B: %zu %zu %zu %zu)\n", dirty, byq[0], byq[1], byq[2], byq[3]);
        }
        {
            unsigned* dc; (void)hipMalloc(&dc, 64); (void)hipMemset(dc, 0, 64);
            h2_debug_counters = dc;
            for (int rep = 0; rep < 5; ++rep) h2_run<64>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, wsp, WSF);
            (void)hipDeviceSynchronize();
            unsigned hc[16]; (void)hipMemcpy(hc, dc, 64, hipMemcpyDeviceToHost);
            printf("  DEBUG exchange of a constant: wrong words by lane quarter %u %u %u %u; last wrong values %08x %08x %08x %08x; at (pt ct q) %u, expected there %08x\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], hc[8], hc[9]);
            h2_debug_counters = nullptr; (void)hipFree(dc);
            rcb = pwc_conv3x3_h2_f32(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, wsp, WSF, 0);
            (void)hipDeviceSynchronize();
        }
        printf("  launch rc: F(2x2) %d, F(4x4) %d, F(4x4) f16x2 direct %d; hip: %s\n", rc, rc4, rcb, hipGetErrorString(hipGetLastError()));
        std::vector<float> h2(npix * ycs), h4(npix * ycs), hbb(npix * ycs);
        (void)hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h4.data(), y4, h4.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hbb.data(), yb, hbb.size() * 4, hipMemcpyDeviceToHost);
        { size_t b2 = 0, b42 = 0; for (size_t p = 0; p < npix; ++p) for (int c = 0; c < sh.Cout; ++c) { if (fabs((double)hbb[p * ycs + c] - h2[p * ycs + c]) > 2e-4 * sh.in_scale) ++b2; if (fabs((double)h4[p * ycs + c] - h2[p * ycs + c]) > 2e-4 * sh.in_scale) ++b42; }
          printf("  entries off by more than 2e-4: f16x2 direct vs F(2x2) %zu, F(4x4) vs F(2x2) %zu\n", b2, b42);
          if (b2 && sh.dil == 1 && sh.Cout == 128) {
              size_t bym7[7] = {0}, byrow[8] = {0}; int shown = 0;
              for (size_t p = 0; p < npix; ++p) for (int c = 0; c < sh.Cout; ++c) if (fabs((double)hbb[p * ycs + c] - h2[p * ycs + c]) > 2e-4 * sh.in_scale) {
                  const int n = (int)(p / ((size_t)sh.H * sh.W)), yy = (int)((p / sh.W) % sh.H), xx = (int)(p % sh.W);
                  const int tile = (n * ((sh.H + 7) / 8) + yy / 8) * ((sh.W + 31) / 32) + xx / 32;
                  ++bym7[tile % 7]; ++byrow[yy % 8];
                  if (shown < 8 && c == 0) { printf("    bad tile %d (n %d y %d x %d): got %.6f want %.6f\n", tile, n, yy, xx, hbb[p * ycs + c], h2[p * ycs + c]); ++shown; }
              }
              printf("    bad by tile %% 7:"); for (int i = 0; i < 7; ++i) printf(" %zu", bym7[i]); printf("   by row %% 8:"); for (int i = 0; i < 8; ++i) printf(" %zu", byrow[i]); printf("\n");
          } }
        double md = 0, mx = 0, mpad = 0; size_t bad = 0, nan = 0;
        size_t hy[16] = {0}, hxm[32] = {0}, hc[8] = {0};
        const double tol = 2e-4 * sh.in_scale;
        for (size_t p = 0; p < npix; ++p) {
            for (int c = 0; c < sh.Cout; ++c) {
                const double a = hbb[p * ycs + c], e = h4[p * ycs + c];
                if (a != a) { ++nan; continue; }
                md = fmax(md, fabs(a - e)); mx = fmax(mx, fabs(e));
                if (fabs(a - e) > tol) {
                    if (bad < 6) printf("    mismatch n %zu y %zu x %zu c %d: f16x2 direct %.6f fp32 F(4x4) %.6f\n", p / ((size_t)sh.H * sh.W), (p / sh.W) % sh.H, p % sh.W, c, a, e);
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
        printf("  f16x2 direct vs fp32 F(4x4): max |diff| %.3e (max |value| %.3f), %zu entries > %.0e, %zu NaN; channels beyond Cout max %.1e\n", md, mx, bad, tol, nan, mpad);
        printf("  NUMERICS vs float64 direct conv (%d samples, rms |y| %.3e): F(2x2) fp32 max %.3e rms %.3e | F(4x4) fp32 max %.3e rms %.3e | F(4x4) f16x2 direct max %.3e rms %.3e  (f16x2 direct / fp32 F(4x4): max x%.2f rms x%.2f)\n",
               NS, sqrt(sv / NS), e2, sqrt(s2 / NS), e4, sqrt(s4 / NS), eb, sqrt(sb / NS), eb / e4, sqrt(sb / s4));
        fflush(stdout);
        for (int round = 0; round < 2; ++round) {
            const float t2 = time_us([&](int) { pwc_conv3x3_wino_f32(x, sh.xcs, u2, b, y2, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            const float t4 = time_us([&](int) { pwc_conv3x3_wino4_f32(x, sh.xcs, u4, b, y4, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0); }, 10);
            const float tb = time_us([&](int) { pwc_conv3x3_h2_f32(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, wsp, WSF, 0); }, 10);
            printf("  F(2x2) %8.1f us %6.1f TF | F(4x4) fp32 %8.1f us %6.1f TF | F(4x4) f16x2 direct %8.1f us %6.1f TF (direct-conv flops)  x%.2f vs fp32 F(4x4)\n",
                   t2, gf / t2 * 1e3, t4, gf / t4 * 1e3, tb, gf / tb * 1e3, t4 / tb);
            fflush(stdout);
        }
        for (int vv = 2; vv <= 11; ++vv) {
            const int v = vv >> 1; const bool use_ws = vv & 1;
            (void)hipMemset(yb, 0, npix * ycs * 4);
            const int rv = h2_run<0>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, v, use_ws ? wsp : nullptr, use_ws ? WSF : 0);
            if (rv) continue;
            (void)hipMemcpy(hbb.data(), yb, hbb.size() * 4, hipMemcpyDeviceToHost);
            double mdv = 0; size_t nanv = 0;
            for (size_t p = 0; p < npix; ++p)
                for (int c = 0; c < ycs; ++c) {
                    const double a = hbb[p * ycs + c], e = c < sh.Cout ? h2[p * ycs + c] : 0.0;
                    if (a != a) { ++nanv; continue; }
                    mdv = fmax(mdv, fabs(a - e));
                }
            const float tv = time_us([&](int) { h2_run<0>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, v, use_ws ? wsp : nullptr, use_ws ? WSF : 0); }, 10);
            printf("  variant %d %s: max |diff| vs fp32 F(2x2) %.3e, %zu NaN, %8.1f us %6.1f TF\n", v, use_ws ? "stream-K    " : "tile per WG ", mdv, nanv, tv, gf / tv * 1e3);
            fflush(stdout);
        }
        if (idx == 0 || idx == 2) {
            {
                std::vector<float> ta, tc;
                for (int rep = 0; rep < 9; ++rep) {
                    ta.push_back(time_us([&](int) { h2_run<0>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, wsp, WSF); }, 10));
                    tc.push_back(time_us([&](int) { h2_run<0>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, nullptr, 0); }, 10));
                }
                std::sort(ta.begin(), ta.end()); std::sort(tc.begin(), tc.end());
                printf("  medians of 9 interleaved rounds: stream-K %.1f us (min %.1f), one workgroup per tile %.1f us (min %.1f)\n", ta[4], ta[0], tc[4], tc[0]);
            }
            unsigned* dc; (void)hipMalloc(&dc, 256 * 8 * 16); (void)hipMemset(dc, 0, 256 * 8 * 16);
            h2_debug_counters = dc;
            for (int rep = 0; rep < 3; ++rep) h2_run<2048>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, wsp, WSF);
            (void)hipDeviceSynchronize();
            std::vector<unsigned> hc(256 * 8 * 4); (void)hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
            double st = 0, sw = 0, sb = 0, sf = 0; for (int i = 0; i < 256 * 8; ++i) { st += hc[i * 4]; sw += hc[i * 4 + 1]; sb += hc[i * 4 + 2]; sf += hc[i * 4 + 3]; }
            printf("  s_memtime, mean over waves: kernel %.0f ticks; waiting for fetches %.0f (%.1f %%), in barriers %.0f (%.1f %%), piece ends %.0f (%.1f %%)\n", st / 2048, sw / 2048, 100 * sw / st, sb / 2048, 100 * sb / st, sf / 2048, 100 * sf / st);
            printf("    workgroup 0: "); for (int w = 0; w < 8; ++w) printf("[%u %u %u %u] ", hc[w * 4], hc[w * 4 + 1], hc[w * 4 + 2], hc[w * 4 + 3]); printf("\n");
            (void)hipMemset(dc, 0, 256 * 8 * 16);
            h2_run<4096>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, wsp, WSF);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(hc.data(), dc, 2 * 8 * 4 * 16 * 4, hipMemcpyDeviceToHost);
            printf("  tap timeline (s_memtime ticks; stages 8-11 of the range; per stage: wait+barrier | taps (0,0) (0,1) (0,2) | w+b | (1,0) (1,1) (1,2) | w+b | (2,0) (2,1) (2,2))\n");
            for (int wgi = 0; wgi < 2; ++wgi) for (int w = 0; w < 8; w += (wgi ? 4 : 1)) {
                printf("    workgroup %d wave %d:", wgi ? 5 : 0, w);
                for (int st = 0; st < 4; ++st) { const unsigned* q = &hc[((wgi * 8 + w) * 4 + st) * 16]; printf("  [%u | %u %u %u | %u | %u %u %u | %u | %u %u %u]", q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[7] - q[6], q[8] - q[7], q[9] - q[8], q[10] - q[9], q[11] - q[10], q[12] - q[11]); }
                printf("\n");
            }
            h2_debug_counters = nullptr; (void)hipFree(dc);
        }
        if (idx == 0 || idx == 2) {
            auto ab = [&](auto tag) { return time_us([&](int) { h2_run<decltype(tag)::value>(x, sh.xcs, ub, b, yb, ycs, sh.N, sh.H, sh.W, sh.Cin, sh.Cout, sh.dil, 1, 0.1f, 0, 0, wsp, WSF); }, 10); };
            printf("  ablations: no patch DMA %.1f | no weight DMA %.1f | no DMA %.1f | no MFMA %.1f | m' = 0 %.1f | no split %.1f | no fragment reads %.1f |"
                   " no DMA, no MFMA %.1f | no MFMA, no fragment reads %.1f | MFMA only %.1f | nothing %.1f us\n",
                   ab(std::integral_constant<int, 1>{}), ab(std::integral_constant<int, 2>{}), ab(std::integral_constant<int, 3>{}),
                   ab(std::integral_constant<int, 4>{}), ab(std::integral_constant<int, 8>{}), ab(std::integral_constant<int, 16>{}),
                   ab(std::integral_constant<int, 32>{}), ab(std::integral_constant<int, 7>{}), ab(std::integral_constant<int, 36>{}),
                   ab(std::integral_constant<int, 51>{}), ab(std::integral_constant<int, 55>{}));
            fflush(stdout);
        }
        (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(y2); (void)hipFree(y4); (void)hipFree(yb);
        (void)hipFree(u2); (void)hipFree(u4); (void)hipFree(ub);
    }
    return 0;
}
