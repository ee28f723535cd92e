// wgbench - A/B timing of wgrad_mfma_kernel configurations (weight gradients of the 3x3 convs) in one process.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sub-cortical_segmentation_amd/csrc tools/wgbench.hip -o build/wgbench
// Ablation builds (timings only, results wrong on purpose): -DSCS_WG_ABLATE=1 no MFMAs, 2 no staging, 3 no partial store.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wgrad_mfma.h"

namespace scs { void set_error(const char*, ...) {} }
using namespace scs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_random(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <class C>
static void run(const char* name, int n, int slices, int rounds) {
    CK(wgrad_set_attr<C>());
    WgArgs a{};
    a.n = n; a.slices = slices;
    for (int t = 0; t < kTowers; ++t) {
        a.X[t] = dev_random((size_t)n * C::CIN * C::HIN * C::HIN, 1 + t);
        a.dZ[t] = dev_random((size_t)n * C::COUT * C::HO * C::HO, 7 + t);
    }
    CK(hipMalloc(&a.part, (size_t)kTowers * slices * C::NB * C::PARTS * C::COUT * C::J * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) wgrad_launch<C>(a, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, nullptr));
        wgrad_launch<C>(a, nullptr);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    CK(hipGetLastError());
    const double flop = 2.0 * C::CIN * 9.0 * C::COUT * C::HO * C::HO * 3.0 * n;
    printf("%-34s n=%d slices=%3d WGs=%4d  min %7.1f us  mean %7.1f us  %5.1f TF/s  LDS %zu B\n", name, n, slices,
           slices * C::NB * 3, best * 1e3, sum / rounds * 1e3, flop / (best * 1e-3) / 1e12, (size_t)C::LDS_BYTES);
    for (int t = 0; t < kTowers; ++t) { CK(hipFree((void*)a.X[t])); CK(hipFree((void*)a.dZ[t])); }
    CK(hipFree(a.part));
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    for (int rep = 0; rep < 2; ++rep) {
        run<WgCfg<20, 20, 30, 1, 1>>("conv2 NB1 WPC1", n, 64, rounds);
        run<WgCfg<20, 20, 30, 2, 2>>("conv2 NB2 WPC2", n, 128, rounds);
        run<WgCfg<20, 20, 30, 4, 4>>("conv2 NB4 WPC4", n, 64, rounds);
        run<WgCfg<20, 20, 30, 4, 4>>("conv2 NB4 WPC4", n, 256, rounds);
        run<WgCfg<20, 20, 30, 4, 3>>("conv2 NB4 WPC3", n, 64, rounds);
        run<WgCfg<20, 20, 30, 2, 2>>("conv2 NB2 WPC2", n, 86, rounds);
        run<WgCfg<20, 20, 30, 7, 3>>("conv2 NB7 WPC3", n, 37, rounds);
        run<WgCfg<20, 40, 14, 1, 3>>("conv3 NB1 WPC3", n, 256, rounds);
        run<WgCfg<20, 40, 14, 2, 3>>("conv3 NB2 WPC3", n, 128, rounds);
        run<WgCfg<20, 40, 14, 3, 3>>("conv3 NB3 WPC3", n, 86, rounds);
        run<WgCfg<40, 40, 12, 1, 3>>("conv4 NB1 WPC3", n, 256, rounds);
        run<WgCfg<40, 40, 12, 1, 3>>("conv4 NB1 WPC3", n, 128, rounds);
        run<WgCfg<40, 40, 12, 2, 3>>("conv4 NB2 WPC3", n, 128, rounds);
        run<WgCfg<40, 60, 5, 1, 2>>("conv5 NB1 WPC2", n, 64, rounds);
        run<WgCfg<1, 20, 32, 3, 4>>("conv1 NB3 WPC4", n, 64, rounds);
        printf("\n");
    }
    return 0;
}
