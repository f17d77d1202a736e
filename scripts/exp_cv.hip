// Microbenchmark of cost-volume kernel variants (not part of the library).
#include "../pwcnet_amd/csrc/cost_volume.hip"
#include "../pwcnet_amd/csrc/pwc_ops.hip"
#include <cstdio>
#include <vector>

template <int R, bool FUSED>
static float run(CvArgs a, int iters) {
    using G = CvGeom<R>;
    const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_kernel<R, FUSED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    a.tiles_x = (a.W + CV_TW - 1) / CV_TW; a.tiles_y = (a.H + CV_TH - 1) / CV_TH;
    const unsigned nb = a.tiles_x * a.tiles_y * a.N;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((cost_volume_kernel<R, FUSED>), dim3(nb), dim3(G::T), lds, 0, a);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((cost_volume_kernel<R, FUSED>), dim3(nb), dim3(G::T), lds, 0, a);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

template <int R, int ABL>
static void launch_abl(CvArgs a) {
    using G = CvPGeom<R>;
    const size_t lds = (size_t)2 * G::BUF * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_volume_dma_kernel<R, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long ntiles = (long)a.tiles_x * a.tiles_y * a.N, nwg = 256;
    if (nwg > ntiles) nwg = ntiles;
    hipLaunchKernelGGL((cost_volume_dma_kernel<R, ABL>), dim3((unsigned)nwg), dim3(G::T), lds, 0, a);
}

template <int R, int ABL>
static float run_dma(CvArgs a, int iters) {
    a.tiles_x = (a.W + CV_TW - 1) / CV_TW; a.tiles_y = (a.H + CV_TH - 1) / CV_TH;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) launch_abl<R, ABL>(a);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) launch_abl<R, ABL>(a);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

static double checksum(const float* d, size_t n) {
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    double s = 0; for (size_t i = 0; i < n; ++i) s += (double)h[i] * (1 + (i % 13));
    return s;
}

int main() {
    struct Shape { int N, H, W, C; };
    Shape shapes[] = {{8, 112, 256, 32}, {8, 56, 128, 64}, {8, 28, 64, 96}, {8, 14, 32, 128}, {8, 7, 16, 192}};
    for (auto sh : shapes) {
        const size_t npix = (size_t)sh.N * sh.H * sh.W;
        float *f0, *f1, *f1w, *fl, *out;
        hipMalloc(&f0, npix * sh.C * 4); hipMalloc(&f1, npix * sh.C * 4); hipMalloc(&f1w, npix * sh.C * 4);
        hipMalloc(&fl, npix * 2 * 4); hipMalloc(&out, npix * 81 * 4);
        std::vector<float> h(npix * sh.C);
        unsigned r = 777;
        for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
        hipMemcpy(f0, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(f1, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hf(npix * 2);
        for (auto& v : hf) { r = r * 1664525u + 1013904223u; v = (((r >> 8) & 0xFFFF) / 65536.f - 0.5f) * 1.2f; }   // ~N(0,3 px) after x5
        hipMemcpy(fl, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
        CvArgs a{};
        a.f0 = f0; a.f1 = f1; a.flow = fl; a.out = out; a.f0_cs = sh.C; a.f1_cs = sh.C; a.flow_cs = 2; a.out_cs = 81;
        a.N = sh.N; a.H = sh.H; a.W = sh.W; a.C = sh.C; a.flow_scale = 5.f; a.slope = 0.1f; a.out_vec4 = 0;
        const double mb_f = npix * (2.0 * sh.C + 83) * 4 / 1e6, mb_u = npix * (2.0 * sh.C + 81) * 4 / 1e6;
        printf("== N=%d %dx%d C=%d : fused %.1f MB, unfused %.1f MB\n", sh.N, sh.H, sh.W, sh.C, mb_f, mb_u);
        auto rep = [&](const char* nm, float us, double mb) { printf("  %-26s %8.1f us  %7.0f GB/s (%.1f%% of 8 TB/s)\n", nm, us, mb / us * 1e3, mb / us * 1e3 / 80.0); };
        for (int round = 0; round < 2; ++round) {
            rep("fused", run<4, true>(a, 10), mb_f);
            a.f1 = f1w;
            rep("unfused", run<4, false>(a, 10), mb_u);
            double c0 = checksum(out, npix * 81);
            rep("unfused persistent DMA", run_dma<4, 0>(a, 10), mb_u);
            double c1 = checksum(out, npix * 81);
            printf("  checksums %.6f %.6f  diff %.3e\n", c0, c1, c1 - c0);
            {   // the model's case: 81 channels written into a 160-channel buffer (16-byte stores)
                float* out160; hipMalloc(&out160, npix * 160 * 4);
                CvArgs b = a; b.out = out160; b.out_cs = 160; b.out_vec4 = 1;
                rep("  DMA: out_cs=160, 16B stores", run_dma<4, 0>(b, 10), mb_u);
                hipFree(out160);
            }
            rep("  DMA: no FMAs", run_dma<4, 1>(a, 10), mb_u);
            rep("  DMA: no DMA", run_dma<4, 2>(a, 10), mb_u);
            rep("  DMA: no stores", run_dma<4, 4>(a, 10), mb_u);
            rep("  DMA: no FMA, no stores", run_dma<4, 5>(a, 10), mb_u);
            rep("  DMA: only DMA+sync", run_dma<4, 5>(a, 10), mb_u);
            rep("  DMA: nothing but LDS reads", run_dma<4, 7>(a, 10), mb_u);
            a.f1 = f1;
            // standalone warp kernel
            WarpArgs w{}; w.x = f1; w.flow = fl; w.out = f1w; w.x_cs = sh.C; w.flow_cs = 2; w.out_cs = sh.C; w.H = sh.H; w.W = sh.W;
            w.C4 = sh.C / 4; w.flow_scale = 5.f; w.total = (long)npix * w.C4;
            long blocks = (w.total + 255) / 256; if (blocks > 4096) blocks = 4096;
            hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
            hipLaunchKernelGGL(warp_kernel<true>, dim3(blocks), dim3(256), 0, 0, w);
            hipEventRecord(s);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(warp_kernel<true>, dim3(blocks), dim3(256), 0, 0, w);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            rep("warp kernel alone", ms * 100.f, npix * (2.0 * sh.C + 2) * 4 / 1e6);
        }
        hipFree(f0); hipFree(f1); hipFree(f1w); hipFree(fl); hipFree(out);
    }
    return 0;
}
