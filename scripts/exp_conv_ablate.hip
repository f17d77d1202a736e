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
    const int N = 8, H = 112, W = 256, Cin = 128, Cout = 128;
    const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout, nw = (size_t)9 * Cin * Cout;
    float *x, *w, *b, *y;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&b, Cout * 4); hipMalloc(&y, ny * 4);
    std::vector<float> h(nx);
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, Cout * 4);
    ConvArgs a{};
    a.x = x; a.wp = w; a.bias = b; a.y = y; a.x_cs = Cin; a.y_cs = Cout; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
    a.Cin_phys = Cin; a.Cout = Cout; a.Cout_pad = Cout; a.stride = 1; a.dil = 1; a.pad_t = 1; a.pad_l = 1;
    a.apply_act = 1; a.slope = 0.1f; a.M = N * H * W; a.y_vec4 = 1; a.m_begin = 0; a.m_end = a.M;
    a.taps_per_split = 9; a.ws = nullptr;
    const double gf = 2.0 * a.M * 9.0 * Cin * Cout / 1e9;
    auto rep = [&](const char* nm, float us) { printf("%-34s %8.1f us %7.1f TFLOP/s\n", nm, us, gf / us * 1e3); };
    a.xcd_remap = 1;
    for (int round = 0; round < 3; ++round) {
        printf("-- round %d\n", round);
        rep("128x128 KC32 GLDS 2-stage (ref)", run_glds<4, 4, 2, 2, 32>(a, 128, 128, 10));
        double c0 = checksum(y, ny);
        double c1 = c0, c2 = c0, c3 = c0;
        if (round == 0) printf("checksums %.4f %.4f %.4f %.4f\n", c0, c1, c2, c3);
    }
    return 0;
}
