// Round 5: the harness of exp_cv3.hip for cost_volume_h2.hip (F16 pipe, producer / consumer waves) beside the fp32 kernel of round 3.
// (The intermediate forms of the round -- the F16 products inside the round-3 kernel: scripts/exp_cv4.hip at commit c50e90d; the ring
// kernel with four one-role waves, deeper request pipelines, other divisions of the roles: profiles/r05_exp_cv5_*.txt.)
// Microbenchmark + correctness harness: fused warp + cost volume + concat copy on the matrix pipe
// (pwcnet_amd/csrc/cost_volume_mfma.hip) against the production pair of round 2 (warp_kernel + concat copy, then the
// rolling / tile cost-volume kernel).  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize scripts/exp_cv5.hip -o scripts/exp_cv5.bin
#include "../pwcnet_amd/csrc/cost_volume.hip"
#include "../pwcnet_amd/csrc/pwc_ops.hip"
// (cost_volume.hip includes cost_volume_mfma.hip and cost_volume_h2.hip)
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

template <int CG, bool WARP, bool PAD>
static void launch_cvm_old(CvmArgs a);
template <int CG, bool WARP, bool PAD, int ABL, bool H2>
static void launch_cvm_t(CvmArgs a) {
    if constexpr (H2) {
        using GH = CvhGeom<CG>;
        const size_t lds = (size_t)GH::LDS_F * sizeof(float);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_h2_kernel<CG, WARP, PAD, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (a.seg_brows <= 0) cvm_plan(a.N, a.H, a.W, 1, &a.nstrips, &a.nseg, &a.seg_brows);
        else { a.nstrips = (a.W + 15) / 16; a.nseg = (a.nbrows + a.seg_brows - 1) / a.seg_brows; }
        const long items = (long)a.N * a.nstrips * a.nseg;
        hipLaunchKernelGGL((cost_volume_h2_kernel<CG, WARP, PAD, ABL>), dim3((unsigned)items), dim3(512), lds, 0, a);
        return;
    } else {
    launch_cvm_old<CG, WARP, PAD>(a);
    }
}
template <int CG, bool WARP, bool PAD>
static void launch_cvm_old(CvmArgs a) {
    constexpr int ABL = 0;
    using G = CvmGeom<CG>;
    const size_t lds = (size_t)G::LDS_F * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_mfma_kernel<CG, WARP, PAD, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (a.seg_brows <= 0) cvm_plan(a.N, a.H, a.W, G::WGPC, &a.nstrips, &a.nseg, &a.seg_brows);
    else { a.nstrips = (a.W + 15) / 16; a.nseg = (a.nbrows + a.seg_brows - 1) / a.seg_brows; }
    const long items = (long)a.N * a.nstrips * a.nseg;
    hipLaunchKernelGGL((cost_volume_mfma_kernel<CG, WARP, PAD, ABL>), dim3((unsigned)items), dim3(G::T), lds, 0, a);
}
template <int CG, int ABL, bool H2>
static void launch_cvm_c(const CvmArgs& a) {
    if (a.flow) { if (a.pad_ok) launch_cvm_t<CG, true, true, ABL, H2>(a); else launch_cvm_t<CG, true, false, ABL, H2>(a); }
    else { if (a.pad_ok) launch_cvm_t<CG, false, true, ABL, H2>(a); else launch_cvm_t<CG, false, false, ABL, H2>(a); }
}
template <int ABL, bool H2 = false>
static void launch_cvm(const CvmArgs& a, int C) {
    if (C == 32) launch_cvm_c<2, ABL, H2>(a); else if (C == 64) launch_cvm_c<4, ABL, H2>(a); else launch_cvm_c<6, ABL, H2>(a);
}

int main(int argc, char** argv) {
    struct Shape { int N, H, W, C; float sigma; };
    Shape shapes[] = {{8, 112, 256, 32, 3.f}, {8, 56, 128, 64, 3.f}, {8, 28, 64, 96, 3.f}, {3, 100, 75, 32, 6.f},
                      {2, 30, 60, 64, 2.f}, {1, 15, 30, 96, 2.f}, {8, 240, 480, 32, 3.f}, {1, 112, 256, 32, 3.f},
                      {8, 112, 256, 32, 0.f}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int seg_override = argc > 2 ? atoi(argv[2]) : 0;
    const int ocs_override = argc > 3 ? atoi(argv[3]) : 0;
    const bool stamps = argc > 4;            // estimator buffer channel stride (0: the model's)
    int shape_idx = -1;
    for (auto sh : shapes) {
        ++shape_idx;
        if (only >= 0 && shape_idx != only) continue;
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        const int C = sh.C;
        const int ocs = ocs_override ? ocs_override : ((84 + C + 4 + 32 + 15) / 16) * 16;   // estimator buffer: [cv 84 | f0 C | flow 4 | feat_up 32]
        const double set_mb = npix * (3.0 * C + 2 + ocs) * 4 / 1e6;
        int NSETS = (int)(300.0 / set_mb) + 1; if (NSETS < 2) NSETS = 2; if (NSETS > 24) NSETS = 24;
        std::vector<float*> f0(NSETS), f1(NSETS), fl(NSETS), E(NSETS), f1w(NSETS);
        std::vector<float> h(npix * C), hf(npix * 2);
        unsigned r = 777 + shape_idx;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 65536.f; };
        auto gauss = [&]() { float u1 = rnd() + 1e-6f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
        for (int s = 0; s < NSETS; ++s) {
            (void)hipMalloc(&f0[s], npix * C * 4); (void)hipMalloc(&f1[s], npix * C * 4); (void)hipMalloc(&fl[s], npix * 2 * 4);
            (void)hipMalloc(&E[s], npix * ocs * 4); (void)hipMalloc(&f1w[s], npix * C * 4);
            for (auto& v : h) v = rnd() - 0.5f;
            (void)hipMemcpy(f0[s], h.data(), h.size() * 4, hipMemcpyHostToDevice);
            for (auto& v : h) v = rnd() - 0.5f;
            (void)hipMemcpy(f1[s], h.data(), h.size() * 4, hipMemcpyHostToDevice);
            for (auto& v : hf) v = gauss() * sh.sigma / 5.0f;          // the kernels multiply by flow_scale = 5
            if (s == 0 && sh.sigma > 0) { hf[0] = 60.f; hf[1] = -60.f; hf[2 * (sh.W + 1)] = -40.f; hf[2 * (sh.W + 1) + 1] = 35.f; }
            (void)hipMemcpy(fl[s], hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemset(E[s], 0, npix * ocs * 4);
        }
        float* Eref; (void)hipMalloc(&Eref, npix * ocs * 4); (void)hipMemset(Eref, 0, npix * ocs * 4);
        const double mb = npix * (2.0 * C + 2 + 81) * 4 / 1e6;
        printf("== N=%d %dx%d C=%d sigma=%.1f : %.1f MB algorithmic (fused), est. buffer cs %d, %d operand sets\n", sh.N, sh.H, sh.W, C, sh.sigma, mb, ocs, NSETS);
        auto rep = [&](const char* nm, float us) { printf("  %-44s %8.1f us  %7.0f GB/s (%.1f%% of 8 TB/s)\n", nm, us, mb / us * 1e3, mb / us * 1e3 / 80.0); fflush(stdout); };

        auto ref_pair = [&](int s, float* out) {     // production pair: warp (+ f0 concat copy), then cost volume
            warp_common(true, f1[s], C, fl[s], 2, 5.0f, f1w[s], C, sh.N, sh.H, sh.W, C, f0[s], C, out + 84, ocs, C, 0);
            if (cv_roll_eligible(f0[s], C, f1w[s], C, out, ocs, nullptr, 0, sh.H, sh.W, C, 4))
                cv_roll_launch(f0[s], C, f1w[s], C, out, ocs, nullptr, 0, sh.N, sh.H, sh.W, 0.1f, 0);
            else {
                CvArgs a{};
                a.f0 = f0[s]; a.f1 = f1w[s]; a.flow = nullptr; a.out = out; a.f0_cs = C; a.f1_cs = C; a.flow_cs = 0; a.out_cs = ocs;
                a.N = sh.N; a.H = sh.H; a.W = sh.W; a.C = C; a.flow_scale = 1.f; a.slope = 0.1f;
                cv_dispatch(a, 4, false, 0);
            }
        };
        auto new_args = [&](int s, float* out, bool warp, bool copy, bool pad) {
            CvmArgs a{};
            a.f0 = f0[s]; a.f1 = f1[s]; a.flow = warp ? fl[s] : nullptr; a.out = out; a.f0_copy = copy ? out + 84 : nullptr;
            a.f0_cs = C; a.f1_cs = C; a.flow_cs = 2; a.out_cs = ocs; a.f0_copy_cs = ocs;
            a.N = sh.N; a.H = sh.H; a.W = sh.W; a.flow_scale = 5.0f; a.slope = 0.1f; a.inv_c = 1.0f / C;
            a.nbrows = (sh.H + 3) / 4; a.pad_ok = pad ? 1 : 0; a.seg_brows = seg_override;
            return a;
        };
        auto compare = [&](const char* what, float* got, float* ref, bool copy) {
            std::vector<float> ha(npix * ocs), hb(npix * ocs);
            (void)hipMemcpy(ha.data(), got, ha.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hb.data(), ref, hb.size() * 4, hipMemcpyDeviceToHost);
            double md = 0, mx = 0, mc = 0; size_t bad = 0, nanc = 0;
            size_t hd[81] = {0}, hx[16] = {0}, hy[4] = {0};
            for (size_t p = 0; p < npix; ++p) {
                for (int d = 0; d < 81; ++d) {
                    const double x = ha[p * ocs + d], y = hb[p * ocs + d];
                    if (x != x) { ++nanc; continue; }
                    md = fmax(md, fabs(x - y)); mx = fmax(mx, fabs(y));
                    if (fabs(x - y) > 2e-6) {
                        if (bad < 8) printf("    mismatch n %zu y %zu x %zu d %d (v %d h %d): got %.7f exp %.7f\n", p / ((size_t)sh.H * sh.W), (p / sh.W) % sh.H, p % sh.W, d, d / 9 - 4, d % 9 - 4, x, y);
                        ++bad; ++hd[d]; ++hx[(p % sh.W) & 15]; ++hy[((p / sh.W) % sh.H) & 3];
                    }
                }
                for (int d = 81; d < ocs; ++d) {
                    if (!copy && d >= 84 && d < 84 + C) continue;
                    mc = fmax(mc, fabs((double)ha[p * ocs + d] - hb[p * ocs + d]));
                }
            }
            if (bad) {
                printf("    bad by v: "); for (int v = 0; v < 9; ++v) { size_t c = 0; for (int hh = 0; hh < 9; ++hh) c += hd[v * 9 + hh]; printf("%zu ", c); }
                printf("\n    bad by h: "); for (int hh = 0; hh < 9; ++hh) { size_t c = 0; for (int v = 0; v < 9; ++v) c += hd[v * 9 + hh]; printf("%zu ", c); }
                printf("\n    bad by x%%16: "); for (int i = 0; i < 16; ++i) printf("%zu ", hx[i]);
                printf("\n    bad by y%%4: "); for (int i = 0; i < 4; ++i) printf("%zu ", hy[i]);
                printf("\n");
            }
            printf("  %-22s max |diff| %.3e (max |value| %.3f), %zu entries > 2e-6, %zu NaN; other channels (padding / f0 copy) max err %.3e\n", what, md, mx, bad, nanc, mc);
            fflush(stdout);
        };
        {   // correctness: fused (warp + copy + pad) vs the pair
            ref_pair(0, Eref);
            CvmArgs b = new_args(0, E[0], true, true, true);
            launch_cvm<0>(b, C);
            (void)hipDeviceSynchronize();
            printf("  hip status: %s\n", hipGetErrorString(hipGetLastError()));
            {
                using G2 = CvmGeom<2>;
                CvmArgs pl = b; int wg = 2; cvm_plan(pl.N, pl.H, pl.W, wg, &pl.nstrips, &pl.nseg, &pl.seg_brows);
                printf("  plan (2 WG/CU): strips %d, segments %d x %d block rows -> %d items; LDS C=32 %d B\n", pl.nstrips, pl.nseg, pl.seg_brows, pl.N * pl.nstrips * pl.nseg, G2::LDS_F * 4);
            }
            compare("fused+copy+pad:", E[0], Eref, true);
            (void)hipMemset(E[0], 0, npix * ocs * 4);
            launch_cvm<0, true>(b, C);
            (void)hipDeviceSynchronize();
            compare("H2 fused+copy+pad:", E[0], Eref, true);
            // no pad, no copy: the b32 store of channel 80, channels 81.. untouched
            (void)hipMemset(E[1 % NSETS], 0, npix * ocs * 4);
            (void)hipMemset(Eref, 0, npix * ocs * 4);
            ref_pair(1 % NSETS, Eref);
            CvmArgs c2 = new_args(1 % NSETS, E[1 % NSETS], true, false, false);
            launch_cvm<0>(c2, C);
            (void)hipDeviceSynchronize();
            compare("fused, no copy/pad:", E[1 % NSETS], Eref, false);
            (void)hipMemset(E[1 % NSETS], 0, npix * ocs * 4);
            launch_cvm<0, true>(c2, C);
            (void)hipDeviceSynchronize();
            compare("H2 fused, no copy/pad:", E[1 % NSETS], Eref, false);
            // no warp: plain cost volume of (f0, f1)
            (void)hipMemset(E[0], 0, npix * ocs * 4);
            (void)hipMemset(Eref, 0, npix * ocs * 4);
            {
                CvArgs a{};
                a.f0 = f0[0]; a.f1 = f1[0]; a.flow = nullptr; a.out = Eref; a.f0_cs = C; a.f1_cs = C; a.flow_cs = 0; a.out_cs = ocs;
                a.N = sh.N; a.H = sh.H; a.W = sh.W; a.C = C; a.flow_scale = 1.f; a.slope = 0.1f;
                cv_dispatch(a, 4, false, 0);
            }
            CvmArgs c3 = new_args(0, E[0], false, false, false);
            launch_cvm<0>(c3, C);
            (void)hipDeviceSynchronize();
            compare("no warp:", E[0], Eref, false);
            (void)hipMemset(E[0], 0, npix * ocs * 4);
            launch_cvm<0, true>(c3, C);
            (void)hipDeviceSynchronize();
            compare("H2 no warp:", E[0], Eref, false);
            printf("  hip status: %s\n", hipGetErrorString(hipGetLastError()));
        }
        if (stamps) {
            long long* dbg; (void)hipMalloc(&dbg, 512 * 8); (void)hipMemset(dbg, 0, 512 * 8);
            for (int rep_i = 0; rep_i < 3; ++rep_i) { CvmArgs b = new_args(rep_i % NSETS, E[rep_i % NSETS], true, false, true); b.dbg = dbg; launch_cvm<8, true>(b, C); }
            (void)hipDeviceSynchronize();
            long long hd[512]; (void)hipMemcpy(hd, dbg, sizeof(hd), hipMemcpyDeviceToHost);
            const char* nm[] = {"issue", "grp+1", "grp0", "grp-1", "blend+tab+splitA", "copyout", "barrier", "(loop)"};
            for (int w = 0; w < 4; ++w) {
                long long* d = hd + w * 128;
                printf("  workgroup %d wave %d: stamps per step (cycles)\n", w >= 2 ? 300 : 0, (w & 1) ? 3 : 0);
                for (int st = 0; st < 12 && d[st * 8 + 8] != 0; ++st) {
                    printf("    step %2d:", st);
                    for (int ph = 0; ph < 8; ++ph) printf(" %s %5lld", nm[ph], d[st * 8 + ph + 1] - d[st * 8 + ph]);
                    printf("  | total %lld\n", d[st * 8 + 8] - d[st * 8]);
                }
            }
            { CvmArgs b = new_args(0, E[0], true, false, true); b.dbg = dbg;
              rep("  H2 no copy: instrumented (stamps)", time_us([&](int i) { launch_cvm<8, true>(b, C); }, 6)); }
            (void)hipFree(dbg);
        }
        for (int round = 0; round < 2; ++round) {
            rep("fp32 fused (warp + cv + copy + pad)", time_us([&](int i) { launch_cvm<0>(new_args(i % NSETS, E[i % NSETS], true, true, true), C); }, 12));
            rep("fp32 fused, no copy", time_us([&](int i) { launch_cvm<0>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("H2 fused (warp + cv + copy + pad)", time_us([&](int i) { launch_cvm<0, true>(new_args(i % NSETS, E[i % NSETS], true, true, true), C); }, 12));
            rep("H2 fused, no copy", time_us([&](int i) { launch_cvm<0, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("H2, no warp (cv + copy)", time_us([&](int i) { launch_cvm<0, true>(new_args(i % NSETS, E[i % NSETS], false, true, true), C); }, 12));
            rep("  H2 no copy: no MFMAs", time_us([&](int i) { launch_cvm<1, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("  H2 no copy: no gather loads", time_us([&](int i) { launch_cvm<2, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("  H2 no copy: no stores", time_us([&](int i) { launch_cvm<4, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("  H2 no copy: MFMAs only (skeleton)", time_us([&](int i) { launch_cvm<6, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
            rep("  H2 no copy: gather + stores only", time_us([&](int i) { launch_cvm<1, true>(new_args(i % NSETS, E[i % NSETS], true, false, true), C); }, 12));
        }
        for (int s = 0; s < NSETS; ++s) { (void)hipFree(f0[s]); (void)hipFree(f1[s]); (void)hipFree(fl[s]); (void)hipFree(E[s]); (void)hipFree(f1w[s]); }
        (void)hipFree(Eref);
    }
    return 0;
}
