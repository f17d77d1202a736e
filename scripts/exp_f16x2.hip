// Arithmetic of an fp32 product sum on the F16 matrix pipe with SCALED two-term operand splits (round 4, for a direct
// 3x3 convolution that needs no Winograd transform and therefore no in-kernel split per position):
//     x = h + m,   h = fp16(x),   m' = fp16((x - h) * 2^11)           (x - h is exact in fp32; 11 + 11 = 22 significant bits,
//                                                                      the scaling keeps m' out of fp16's subnormal range)
//     u v ~= uh vh + 2^-11 (uh vm' + um' vh) [+ 2^-22 um' vm']        three (four) products, two (three) accumulators
// One 16x16 tile per wave, K-chain, error against float64, next to the fp32 MFMA chain and the bf16 x 3 form.
//   MODE 0: v_mfma_f32_16x16x4_f32 chain          MODE 1: f16 x 2 scaled, 3 products, accumulators {hh}, {cross}
//   MODE 2: + the um' vm' product (third accumulator)              MODE 3: f16 x 2 UNscaled, one accumulator
//   MODE 4: bf16 x 3, six products (conv3x3_wino4b.hip)             MODE 5: as 1 with a single accumulator per 32 k (cross terms
//                                                                    pre-scaled by 2^-11 in the weights: exact, but um' 2^-11 can go subnormal)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_f16x2.hip -o scripts/exp_f16x2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split2(float x, bool scaled, _Float16& h, _Float16& m) {
    h = (_Float16)x;
    const float r = x - (float)h;
    m = (_Float16)(scaled ? r * 2048.f : r);
}
__device__ __forceinline__ void split3b(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

template <int MODE>
__global__ void gemm_tile_kernel(const float* __restrict__ U, const float* __restrict__ V, float* __restrict__ C, int K) {
    const int lane = threadIdx.x & 63, tile = blockIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const float* u = U + (size_t)tile * 16 * K;     // [16][K]
    const float* v = V + (size_t)tile * K * 16;     // [K][16]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accx = acc, accm = acc;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(u[fr * K + k + fq], v[(k + fq) * 16 + fr], acc, 0, 0, 0);
    } else if (MODE == 4) {
        for (int k0 = 0; k0 < K; k0 += 16) {
            bf16x8 a1, a2, a3, b1, b3;
            for (int e = 0; e < 8; ++e) {
                const int ch = k0 + 4 * fq + (e & 3);       // 4 channels x 2 terms per lane
                __bf16 uh, um, ul, vh, vm, vl;
                split3b(u[fr * K + ch], uh, um, ul);
                split3b(v[ch * 16 + fr], vh, vm, vl);
                a1[e] = e < 4 ? uh : um; a3[e] = e < 4 ? ul : uh;
                b1[e] = vh; a2[e] = e < 4 ? uh : um; b3[e] = e < 4 ? vh : vl;
                (void)vm;
            }
            bf16x8 bm;
            for (int e = 0; e < 8; ++e) { __bf16 vh, vm, vl; split3b(v[(k0 + 4 * fq + (e & 3)) * 16 + fr], vh, vm, vl); bm[e] = vm; }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
        }
    } else {
        const bool scaled = MODE != 3;
        for (int k0 = 0; k0 < K; k0 += 32) {
            // hh: 32 channels in one MFMA (8 per lane); cross terms: A = [uh 16 | um' 16], B = [vm' 16 | vh 16] per 16 channels
            f16x8 ah, bh;
            for (int e = 0; e < 8; ++e) {
                _Float16 h, m;
                split2(u[fr * K + k0 + 8 * fq + e], scaled, h, m); ah[e] = h;
                split2(v[(k0 + 8 * fq + e) * 16 + fr], scaled, h, m); bh[e] = h;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            for (int half = 0; half < 2; ++half) {
                f16x8 ax, bx, am, bm;
                for (int e = 0; e < 8; ++e) {
                    const int ch = k0 + 16 * half + 8 * (fq & 1) + e;
                    _Float16 uh, um, vh, vm;
                    split2(u[fr * K + ch], scaled, uh, um);
                    split2(v[ch * 16 + fr], scaled, vh, vm);
                    if (MODE == 5) um = (_Float16)((float)um * (1.f / 2048.f)), vm = (_Float16)((float)vm * (1.f / 2048.f));
                    ax[e] = fq < 2 ? uh : um; bx[e] = fq < 2 ? vm : vh;
                    am[e] = fq < 2 ? um : (_Float16)0.f; bm[e] = fq < 2 ? vm : (_Float16)0.f;
                }
                if (MODE == 3 || MODE == 5) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax, bx, acc, 0, 0, 0);
                else accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax, bx, accx, 0, 0, 0);
                if (MODE == 2) accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm, accm, 0, 0, 0);
            }
        }
        if (MODE == 1 || MODE == 2) acc += accx * (1.f / 2048.f);
        if (MODE == 2) acc += accm * (1.f / 4194304.f);
    }
    for (int r = 0; r < 4; ++r) C[(size_t)tile * 256 + (fq * 4 + r) * 16 + fr] = acc[r];
}

int main() {
    const int tiles = 256;
    struct Dist { const char* name; int kind; };
    const Dist dists[] = {{"U,V ~ N(0,1)", 0}, {"V = |N(0,1)| (post-activation), U ~ 0.03 N(0,1) (weights)", 1},
                          {"V ~ N(0,1) x 10^uniform(-3,3), U ~ 0.03 N(0,1)", 2}, {"V ~ 1e-3 N(0,1) (small activations), U ~ 0.03 N", 3},
                          {"V ~ 300 |N(0,1)| (large activations), U ~ 0.03 N", 4}, {"V = 1 + 1e-3 N (large mean), U = 1 + 1e-3 N", 5}};
    const int Ks[] = {288, 1152, 1440};
    printf("# one 16x16 tile, K-chain; error against float64 relative to rms |C| (max and rms over %d tiles x 256 entries)\n", tiles);
    for (const Dist& ds : dists) {
        for (int K : Ks) {
            std::vector<float> hu((size_t)tiles * 16 * K), hv((size_t)tiles * K * 16);
            unsigned r = 2345 + K + 77 * ds.kind;
            auto uni = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) + 0.5f) / 16777216.f; };
            auto nrm = [&]() { const float a = uni(), b = uni(); return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b); };
            for (auto& x : hu) x = ds.kind == 0 ? nrm() : ds.kind == 5 ? 1.f + 1e-3f * nrm() : 0.03f * nrm();
            for (auto& x : hv) {
                const float g = nrm();
                x = ds.kind == 0 ? g : ds.kind == 1 ? fabsf(g) : ds.kind == 2 ? g * powf(10.f, 6.f * uni() - 3.f) : ds.kind == 3 ? 1e-3f * g
                    : ds.kind == 4 ? 300.f * fabsf(g) : 1.f + 1e-3f * g;
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
            double emax[6], erms[6];
            auto run = [&](int mode) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(gemm_tile_kernel<0>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 1: hipLaunchKernelGGL(gemm_tile_kernel<1>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 2: hipLaunchKernelGGL(gemm_tile_kernel<2>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 3: hipLaunchKernelGGL(gemm_tile_kernel<3>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    case 4: hipLaunchKernelGGL(gemm_tile_kernel<4>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                    default: hipLaunchKernelGGL(gemm_tile_kernel<5>, dim3(tiles), dim3(64), 0, 0, du, dv, dc, K); break;
                }
                (void)hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, sq = 0; size_t bad = 0;
                for (size_t i = 0; i < hc.size(); ++i) { const double e = hc[i] - ref[i]; if (!(e == e)) { ++bad; continue; } mx = fmax(mx, fabs(e)); sq += e * e; }
                emax[mode] = bad ? NAN : mx / rms; erms[mode] = bad ? NAN : sqrt(sq / hc.size()) / rms;
            };
            for (int m = 0; m < 6; ++m) run(m);
            printf("%-58s K=%4d | fp32 MFMA chain max %.2e rms %.2e | f16x2 scaled, 3 products: max %.2e rms %.2e (x%.2f rms) ; + mm product x%.2f ; "
                   "unscaled x%.2f ; one accumulator x%.2f ; bf16x3 x%.2f\n",
                   ds.name, K, emax[0], erms[0], emax[1], erms[1], erms[1] / erms[0], erms[2] / erms[0], erms[3] / erms[0], erms[5] / erms[0],
                   erms[4] / erms[0]);
            (void)hipFree(du); (void)hipFree(dv); (void)hipFree(dc);
        }
    }
    return 0;
}
