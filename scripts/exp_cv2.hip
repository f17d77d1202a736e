// Microbenchmark: rolling-window cost volume (cost_volume_roll.hip) vs the tile kernel (cost_volume.hip).
// Not part of the library.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_cv2.hip -o scripts/exp_cv2.bin
#include "../pwcnet_amd/csrc/cost_volume.hip"
#include "../pwcnet_amd/csrc/cost_volume_roll.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

template <typename F>
static float time_us(F&& f, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) f(i);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f(i);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

template <int ABL>
static void launch_roll(CvRollArgs a) {
    using G = CvRollGeom;
    const size_t lds = (size_t)G::LDS_F * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_roll_kernel<ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long items = (long)a.N * a.nstrips * a.nseg;
    hipLaunchKernelGGL((cost_volume_roll_kernel<ABL>), dim3((unsigned)(items < 256 ? items : 256)), dim3(G::T), lds, 0, a);
}

int main(int argc, char** argv) {
    struct Shape { int N, H, W, C; };
    Shape shapes[] = {{8, 112, 256, 32}, {8, 240, 480, 32}, {1, 112, 256, 32}, {3, 100, 75, 32}};
    const int NSETS = 4;      // operand sets rotated so that neither L2 nor the Infinity Cache (256 MB) serves them
    const int only = argc > 1 ? atoi(argv[1]) : -1;     // shape index (PMC runs: one shape)
    const bool quick = argc > 2;                       // quick: correctness + the two kernels only
    int shape_idx = -1;
    for (auto sh : shapes) {
        ++shape_idx;
        if (only >= 0 && shape_idx != only) continue;
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        const int ocs = 160;
        float *f0[NSETS], *f1[NSETS], *outA[NSETS], *outB;
        std::vector<float> h(npix * sh.C);
        unsigned r = 777;
        for (int s = 0; s < NSETS; ++s) {
            hipMalloc(&f0[s], npix * sh.C * 4); hipMalloc(&f1[s], npix * sh.C * 4); hipMalloc(&outA[s], npix * ocs * 4);
            for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
            hipMemcpy(f0[s], h.data(), h.size() * 4, hipMemcpyHostToDevice);
            for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
            hipMemcpy(f1[s], h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipMemset(outA[s], 0, npix * ocs * 4);
        }
        hipMalloc(&outB, npix * ocs * 4); hipMemset(outB, 0, npix * ocs * 4);
        const double mb = npix * (2.0 * sh.C + 81) * 4 / 1e6;
        printf("== N=%d %dx%d C=%d : %.1f MB algorithmic\n", sh.N, sh.H, sh.W, sh.C, mb);
        auto rep = [&](const char* nm, float us) { printf("  %-34s %8.1f us  %7.0f GB/s (%.1f%% of 8 TB/s)\n", nm, us, mb / us * 1e3, mb / us * 1e3 / 80.0); };

        auto old_args = [&](int s, float* out) {
            CvArgs a{};
            a.f0 = f0[s]; a.f1 = f1[s]; a.flow = nullptr; a.out = out; a.f0_cs = sh.C; a.f1_cs = sh.C; a.flow_cs = 0; a.out_cs = ocs;
            a.N = sh.N; a.H = sh.H; a.W = sh.W; a.C = sh.C; a.flow_scale = 1.f; a.slope = 0.1f;
            return a;
        };
        auto new_args = [&](int s, float* out, bool copy) {
            CvRollArgs a{};
            a.f0 = f0[s]; a.f1 = f1[s]; a.out = out; a.f0_copy = copy ? out + 84 : nullptr;
            a.f0_cs = sh.C; a.f1_cs = sh.C; a.out_cs = ocs; a.f0_copy_cs = ocs;
            a.N = sh.N; a.H = sh.H; a.W = sh.W; a.slope = 0.1f;
            cv_roll_plan(a.N, a.H, a.W, &a.nstrips, &a.nseg, &a.seg_rows);
            a.dbg = nullptr;
            return a;
        };
        {   // correctness: new vs old on set 0
            CvArgs a = old_args(0, outB);
            cv_dispatch(a, 4, false, 0);
            CvRollArgs b = new_args(0, outA[0], true);
            printf("  plan: strips %d, segments %d x %d rows -> %d items\n", b.nstrips, b.nseg, b.seg_rows, b.N * b.nstrips * b.nseg);
            launch_roll<0>(b);
            hipDeviceSynchronize();
            std::vector<float> ha(npix * ocs), hb(npix * ocs), hf(npix * sh.C);
            hipMemcpy(ha.data(), outA[0], ha.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hb.data(), outB, hb.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hf.data(), f0[0], hf.size() * 4, hipMemcpyDeviceToHost);
            double md = 0, mx = 0, mc = 0; size_t bad = 0;
            size_t hd[81] = {0}, hx[32] = {0}, hy[4] = {0};
            for (size_t p = 0; p < npix; ++p) {
                for (int d = 0; d < 81; ++d) {
                    const double x = ha[p * ocs + d], y = hb[p * ocs + d];
                    md = fmax(md, fabs(x - y)); mx = fmax(mx, fabs(y));
                    if (fabs(x - y) > 1e-5) {
                        if (bad < 6) printf("    mismatch px %zu (y %zu x %zu) d %d (v %d h %d): got %.6f exp %.6f\n", p, (p / sh.W) % sh.H, p % sh.W, d, d / 9, d % 9, x, y);
                        ++bad; ++hd[d]; ++hx[(p % sh.W) & 31]; ++hy[((p / sh.W) % sh.H) & 3];
                    }
                }
                for (int c = 0; c < sh.C; ++c) mc = fmax(mc, fabs((double)ha[p * ocs + 84 + c] - hf[p * sh.C + c]));
                for (int d = 81; d < 84; ++d) mc = fmax(mc, fabs((double)ha[p * ocs + d]));
                for (int d = 84 + sh.C; d < ocs; ++d) mc = fmax(mc, fabs((double)ha[p * ocs + d]));
            }
            if (bad) {
                printf("    bad by v: "); for (int v = 0; v < 9; ++v) { size_t c = 0; for (int h = 0; h < 9; ++h) c += hd[v * 9 + h]; printf("%zu ", c); }
                printf("\n    bad by h: "); for (int h = 0; h < 9; ++h) { size_t c = 0; for (int v = 0; v < 9; ++v) c += hd[v * 9 + h]; printf("%zu ", c); }
                printf("\n    bad by x%%32: "); for (int i = 0; i < 32; ++i) printf("%zu ", hx[i]);
                printf("\n    bad by y%%4: "); for (int i = 0; i < 4; ++i) printf("%zu ", hy[i]);
                printf("\n");
            }
            printf("  new vs old: max |diff| %.3e (max |value| %.3f), %zu entries > 1e-5; f0-copy / padding max err %.3e\n", md, mx, bad, mc);
        }
        for (int round = 0; round < (quick ? 1 : 2); ++round) {
            if (quick) {
                long long* dbg; hipMalloc(&dbg, 256 * 8); hipMemset(dbg, 0, 256 * 8);
                for (int rep_i = 0; rep_i < 2; ++rep_i) { CvRollArgs b = new_args(1, outA[1], false); b.dbg = dbg; launch_roll<8>(b); }
                hipDeviceSynchronize();
                long long hdbg[256]; hipMemcpy(hdbg, dbg, sizeof(hdbg), hipMemcpyDeviceToHost);
                const char* nm[] = {"dma-issue", "compute", "epilogue", "barrierA", "copy-out", "vmwait", "barrierB"};
                for (int wv = 0; wv < 2; ++wv) {
                    long long* d = hdbg + wv * 128;
                    printf("  wave %d stamps (cycles, 100 MHz memtime units?): prologue issue+wait %lld, barrier %lld\n", wv ? 8 : 0, d[1] - d[0], d[2] - d[1]);
                    for (int st = 0; st < 7; ++st) {
                        long long* e = d + 2 + st * 7;
                        printf("    step %d:", st);
                        for (int ph = 0; ph < 7; ++ph) printf(" %s %lld", nm[ph], e[ph + 1] - e[ph]);
                        printf("\n");
                    }
                    printf("    total %lld\n", d[2 + 49] - d[0]);
                }
                { CvRollArgs b = new_args(1, outA[1], false); b.dbg = dbg;
                  rep("  roll: instrumented (stamps)", time_us([&](int i) { launch_roll<8>(b); }, 6)); }
                hipFree(dbg);
                rep("rolling kernel", time_us([&](int i) { launch_roll<0>(new_args(i % NSETS, outA[i % NSETS], false)); }, 6));
                rep("  roll: LDS reads + FMAs only", time_us([&](int i) { launch_roll<6>(new_args(i % NSETS, outA[i % NSETS], false)); }, 6));
                rep("  roll: LDS reads only", time_us([&](int i) { launch_roll<7>(new_args(i % NSETS, outA[i % NSETS], false)); }, 6));
                break;
            }
            rep("tile kernel (old)", time_us([&](int i) { CvArgs a = old_args(i % NSETS, outA[i % NSETS]); cv_dispatch(a, 4, false, 0); }, 12));
            rep("rolling kernel", time_us([&](int i) { launch_roll<0>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("rolling kernel + f0 concat copy", time_us([&](int i) { launch_roll<0>(new_args(i % NSETS, outA[i % NSETS], true)); }, 12));
            rep("  roll: no FMAs", time_us([&](int i) { launch_roll<1>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("  roll: no DMA", time_us([&](int i) { launch_roll<2>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("  roll: no stores", time_us([&](int i) { launch_roll<4>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("  roll: DMA + sync only", time_us([&](int i) { launch_roll<5>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("  roll: LDS reads + FMAs only", time_us([&](int i) { launch_roll<6>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
            rep("  roll: stores only", time_us([&](int i) { launch_roll<3>(new_args(i % NSETS, outA[i % NSETS], false)); }, 12));
        }
        for (int s = 0; s < NSETS; ++s) { hipFree(f0[s]); hipFree(f1[s]); hipFree(outA[s]); }
        hipFree(outB);
    }
    return 0;
}
