// microbench: halo conv kernel variants on the 16->16 (M = 16*224*512) and 32->32 (16*112*256) layers
#include "../pwcnet_amd/csrc/conv3x3_mfma.hip"
#include <cstdio>
#include <vector>
template <int CIN, int COUT, int TH, int NB>
static float run(HaloArgs a, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) launch_halo<CIN, COUT, TH, NB>(a, 0);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) launch_halo<CIN, COUT, TH, NB>(a, 0);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / iters * 1e3f;
}
int main() {
    for (int c : {16, 32}) {
        const int N = 16, H = c == 16 ? 224 : 112, W = c == 16 ? 512 : 256;
        const size_t nx = (size_t)N * H * W * c;
        float *x, *w, *b, *y;
        hipMalloc(&x, nx * 4); hipMalloc(&y, nx * 4); hipMalloc(&w, 9 * c * c * 4); hipMalloc(&b, c * 4);
        std::vector<float> h(nx); unsigned r = 1;
        for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.f - 0.5f; }
        hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), 9 * c * c * 4, hipMemcpyHostToDevice); hipMemset(b, 0, c * 4);
        HaloArgs a{}; a.x = x; a.wp = w; a.bias = b; a.y = y; a.x_cs = c; a.y_cs = c; a.N = N; a.H = H; a.W = W; a.apply_act = 1; a.slope = 0.1f; a.y_vec4 = 1;
        const double gf = 2.0 * N * H * W * 9.0 * c * c / 1e9, mb = 2.0 * nx * 4 / 1e6;
        for (int round = 0; round < 2; ++round) {
            float t;
            if (c == 16) {
                t = run<16, 16, 8, 2>(a, 10); printf("16->16 TH8 NB2: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
                t = run<16, 16, 8, 1>(a, 10); printf("16->16 TH8 NB1: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
                t = run<16, 16, 4, 2>(a, 10); printf("16->16 TH4 NB2: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
                t = run<16, 16, 4, 1>(a, 10); printf("16->16 TH4 NB1: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
            } else {
                t = run<32, 32, 4, 2>(a, 10); printf("32->32 TH4 NB2: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
                t = run<32, 32, 4, 1>(a, 10); printf("32->32 TH4 NB1: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
                t = run<32, 32, 8, 1>(a, 10); printf("32->32 TH8 NB1: %7.1f us %6.1f TF %6.0f GB/s\n", t, gf / t * 1e3, mb / t * 1e3);
            }
        }
    }
    return 0;
}
