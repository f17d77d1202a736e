// Ablation microbenchmark of the 128x128 implicit-GEMM conv kernel (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_conv_ablate.hip -o /tmp/abl && /tmp/abl
#include "../pwcnet_amd/csrc/conv3x3_mfma.hip"
#include <cstdio>
#include <vector>

template <int WM, int WN, int WGM, int WGN, int KC, int ABL>
static float run(const ConvArgs& a, int BM, int BN, int iters) {
    const size_t lds = (size_t)2 * (BM + BN) * KC * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM, a.Cout_pad / BN, 1);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, WGM, WGN, KC, ABL>), grid, dim3(256), lds, 0, a);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, WGM, WGN, KC, ABL>), grid, dim3(256), lds, 0, a);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

template <int WM, int WN, int WGM, int WGN, int KC>
static float run_glds(const ConvArgs& a, int BM, int BN, int iters) {
    const size_t lds = (size_t)2 * (BM + BN) * KC * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM, a.Cout_pad / BN, 1);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_glds_kernel<WM, WN, WGM, WGN, KC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv3x3_mfma_glds_kernel<WM, WN, WGM, WGN, KC>), grid, dim3(64 * WGM * WGN), lds, 0, a);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv3x3_mfma_glds_kernel<WM, WN, WGM, WGN, KC>), grid, dim3(64 * WGM * WGN), lds, 0, a);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

static double checksum(const float* d, size_t n) {
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    double s = 0; for (size_t i = 0; i < n; i += 7) s += h[i] * (1 + (i % 13));
    return s;
}

int main() {
    const int N = 8, H = 112, W = 256, Cin = 128;
    const size_t nx = (size_t)N * H * W * Cin, nw = (size_t)9 * Cin * 128;
    float *x, *w, *b, *y;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&b, 128 * 4); hipMalloc(&y, (size_t)N * H * W * 128 * 4);
    std::vector<float> h(nx);
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, 128 * 4);
    for (int Cout : {96, 64}) {
        ConvArgs a{};
        a.x = x; a.wp = w; a.bias = b; a.y = y; a.x_cs = Cin; a.y_cs = Cout; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
        a.Cin_phys = Cout == 64 ? 96 : 128; a.Cout = Cout; a.Cout_pad = Cout; a.stride = 1; a.dil = 1; a.pad_t = 1; a.pad_l = 1;
        a.apply_act = 1; a.slope = 0.1f; a.M = N * H * W; a.y_vec4 = 1; a.m_begin = 0; a.m_end = a.M;
        a.taps_per_split = 9; a.ws = nullptr; a.xcd_remap = 1;
        const double gf = 2.0 * a.M * 9.0 * a.Cin_phys * Cout / 1e9;
        auto rep = [&](const char* nm, float us) { printf("Cout=%d %-30s %8.1f us %7.1f TFLOP/s\n", Cout, nm, us, gf / us * 1e3); };
        for (int round = 0; round < 2; ++round) {
            if (Cout == 96) {
                rep("64x96 4w <2,3,2,2>", run_glds<2, 3, 2, 2, 32>(a, 64, 96, 10));
                rep("128x96 4w <4,3,2,2>", run_glds<4, 3, 2, 2, 32>(a, 128, 96, 10));
                rep("128x96 8w <2,3,4,2>", run_glds<2, 3, 4, 2, 32>(a, 128, 96, 10));
                rep("128x96 8w <4,1,2,... 6x?>", run_glds<4, 1, 2, 6 / 2 * 0 + 2, 32>(a, 128, 32, 1));
                rep("64x96 8w <1,3,4,2>", run_glds<1, 3, 4, 2, 32>(a, 64, 96, 10));
                rep("128x96 6w <4,2,2,3>", run_glds<4, 2, 2, 3, 32>(a, 128, 96, 10));
            } else {
                rep("128x64 4w <4,2,2,2>", run_glds<4, 2, 2, 2, 32>(a, 128, 64, 10));
                rep("128x64 8w <2,2,4,2>", run_glds<2, 2, 4, 2, 32>(a, 128, 64, 10));
                rep("256x64 8w <4,2,4,2>", run_glds<4, 2, 4, 2, 32>(a, 256, 64, 10));
                rep("64x64 4w <2,2,2,2>", run_glds<2, 2, 2, 2, 32>(a, 64, 64, 10));
            }
        }
    }
    return 0;
}
