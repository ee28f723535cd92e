// wino_bench - the fused conv1 + conv2 + pool1 inference kernel in direct form (conv_mfma.h) against the Winograd form (conv_wino.h)
// on random patches: results compared, both timed on the same GPU.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sub-cortical_segmentation_amd/csrc -I tools/experiments tools/wino_bench.hip -o build/wino_bench
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>

#include "conv_wino.h"

namespace scs { void set_error(const char*, ...) {} }
using namespace scs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using Direct = ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2>;
#ifndef WDBG
#define WDBG 0
#endif
using Wino8 = Wino12Cfg8<SRC_CONV1, WDBG>;

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 25000, reps = argc > 2 ? atoi(argv[2]) : 10;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x((size_t)n * 1024), W1(20 * 9), b1(20), W2(20 * 20 * 9), b2(20);
    for (auto& v : x) v = nd(rng);
    for (auto& v : W1) v = nd(rng) * 0.4f;
    for (auto& v : b1) v = nd(rng) * 0.2f;
    for (auto& v : W2) v = nd(rng) * 0.1f;
    for (auto& v : b2) v = nd(rng) * 0.2f;
    // conv1 operands [12][16] + [12][4] + bias[20] (pack_conv1_kernel), direct conv2 chunks (pack_conv_kernel), Winograd U
    std::vector<float> w1p(kW1Floats, 0.f), wp(conv_pack_floats(1)), bp(32, 0.f), up(kWinoUFloats);
    for (int t = 0; t < 240; ++t) {
        const int k = t < 192 ? t / 16 : (t - 192) / 4, c = t < 192 ? t % 16 : 16 + (t - 192) % 4;
        if (k < 9) w1p[t] = W1[(c * 3 + (2 - k / 3)) * 3 + (2 - k % 3)];
    }
    for (int c = 0; c < 20; ++c) w1p[240 + c] = b1[c];
    for (size_t e = 0; e < wp.size(); ++e) {
        const PackIdx ix = conv_pack_index((int)e, 20);
        wp[e] = ix.n >= 0 ? W2[((ix.n * 20 + ix.c4 * 4 + ix.kk) * 3 + (2 - ix.tap / 3)) * 3 + (2 - ix.tap % 3)] : 0.f;
    }
    for (int c = 0; c < 20; ++c) bp[c] = b2[c];
    for (int e = 0; e < kWinoUFloats; ++e) {
        const WinoIdx ix = wino_pack_index(e);
        float g[9];
        for (int k = 0; k < 9; ++k) g[k] = W2[((ix.n * 20 + ix.cin) * 3 + (2 - k / 3)) * 3 + (2 - k % 3)];
        up[e] = wino_u(g, ix.p / 4, ix.p % 4);
    }
    float *dx, *dw1, *dwp, *dbp, *dup, *o1, *o2;
    const size_t on = (size_t)n * 20 * 196;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dw1, w1p.size() * 4)); CK(hipMalloc(&dwp, wp.size() * 4));
    CK(hipMalloc(&dbp, bp.size() * 4)); CK(hipMalloc(&dup, up.size() * 4));
    CK(hipMalloc(&o1, 3 * on * 4)); CK(hipMalloc(&o2, 3 * on * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw1, w1p.data(), w1p.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbp, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dup, up.data(), up.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(o1, 0, 3 * on * 4)); CK(hipMemset(o2, 0, 3 * on * 4));
    CK(conv_set_attr<Direct>()); CK(wino12x8_set_attr<Wino8>());
    if (getenv("ONE_PER_CU")) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_wino8_kernel<Wino8>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    ConvArgs a{}, b{};
    a.n = b.n = n;
    for (int t = 0; t < kTowers; ++t) {
        a.in[t] = b.in[t] = dx; a.w1[t] = b.w1[t] = dw1; a.bias[t] = b.bias[t] = dbp;
        a.w[t] = dwp; b.w[t] = dup; a.out[t] = o1 + t * on; b.out[t] = o2 + t * on;
    }
    unsigned long long* dst = nullptr;
    if (WDBG == 7) { CK(hipMalloc(&dst, (size_t)3 * n * 2 * 64 * 8)); CK(hipMemset(dst, 0, (size_t)3 * n * 2 * 64 * 8)); b.stamps = dst; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float msd = 0, msw = 0;
    for (int form = 0; form < 2; ++form) {
        for (int w = 0; w < 2; ++w) { if (form == 0) conv_launch<Direct>(a, 0); else { if (getenv("ONE_PER_CU")) hipLaunchKernelGGL(conv12_wino8_kernel<Wino8>, dim3(b.n * 2, 3), dim3(512), 100 * 1024, 0, b); else wino12x8_launch<Wino8>(b, 0); } }
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) { if (form == 0) conv_launch<Direct>(a, 0); else { if (getenv("ONE_PER_CU")) hipLaunchKernelGGL(conv12_wino8_kernel<Wino8>, dim3(b.n * 2, 3), dim3(512), 100 * 1024, 0, b); else wino12x8_launch<Wino8>(b, 0); } }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(form == 0 ? &msd : &msw, e0, e1));
    }
    CK(hipDeviceSynchronize());
    std::vector<float> h1(on), h2(on);
    CK(hipMemcpy(h1.data(), o1, on * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), o2, on * 4, hipMemcpyDeviceToHost));
    double maxd = 0, scale = 0; size_t bad = 0;
    for (size_t i = 0; i < on; ++i) {
        scale = std::max(scale, (double)std::fabs(h1[i]));
        const double d = std::fabs((double)h1[i] - h2[i]);
        if (!(d <= 1e-3)) ++bad;
        maxd = std::max(maxd, d);
    }
    const double flop = 2.0 * 3 * (162000.0 + 2822400.0) * n;
    if (WDBG == 7) {
        const size_t wgs = (size_t)3 * n * 2;
        std::vector<unsigned long long> st(wgs * 64);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        double acc[2][7] = {}, cnt[2] = {0, 0};
        for (size_t g = 0; g < wgs; ++g) {
            const int band = g & 1;
            for (int wv = 0; wv < 8; ++wv) {
                const unsigned long long* q = &st[g * 64 + wv * 8];
                for (int k = 1; k < 7; ++k) acc[band][k] += (double)(q[k] - q[k - 1]);
                cnt[band] += 1;
            }
        }
        for (int band = 0; band < 2; ++band)
            printf("band %d (memtime ticks per wave): stage+U %.0f  conv1 %.0f  wait %.0f  main %.0f  wait %.0f  restage+store %.0f\n", band, acc[band][1] / cnt[band], acc[band][2] / cnt[band],
                   acc[band][3] / cnt[band], acc[band][4] / cnt[band], acc[band][5] / cnt[band], acc[band][6] / cnt[band]);
    }
    printf("n %d: direct %.3f ms (%.1f TFLOP/s)  winograd %.3f ms (%.1f nominal TFLOP/s)  ratio %.3f\n", n, msd / reps, flop / (msd / reps * 1e-3) / 1e12,
           msw / reps, flop / (msw / reps * 1e-3) / 1e12, msd / msw);
    printf("max |direct - winograd| %.3e of scale %.3f (%.2e relative), %zu elements beyond 1e-3\n", maxd, scale, maxd / scale, bad);
    return bad ? 2 : 0;
}
