// kbench - A/B timing of conv3x3_mfma_kernel configurations in ONE process (cdna_hip_programming.md rule 24), plus
// per-phase s_memtime stamps of diagnostic (STAMP) instantiations.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sub-cortical_segmentation_amd/csrc tools/kbench.hip -o build/kbench
// (The round-1 ablation builds - kernels with one phase removed, wrong results on purpose - lived behind SCS_ABLATE in the
// kernel source; that code is gone from the shipped headers, their timings are recorded in profiles/r01_kbench_final.txt.)
// Inputs are random (rule 25); outputs are not checked here (tests/ do that through the C ABI).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "conv_mfma.h"

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
static void run(const char* name, int n, int rounds, double macs_per_patch_tower) {
    CK(conv_set_attr<C>());
    ConvArgs a{};
    a.n = n;
    const size_t in_per = (C::SRC == SRC_CONV1) ? 1024 : (size_t)C::CIN * C::HSRC * C::HSRC;
    for (int t = 0; t < kTowers; ++t) {
        a.in[t] = dev_random((size_t)n * in_per, 1 + t);
        a.w[t] = dev_random((size_t)C::NC4 * C::WCHUNK, 10 + t);
        a.bias[t] = dev_random(C::NT * 16, 20 + t);
        a.w1[t] = dev_random(kW1Floats, 30 + t);
        float* o; CK(hipMalloc(&o, (size_t)n * C::COUT * C::OUTPIX * 4)); a.out[t] = o;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) conv_launch<C>(a, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, nullptr));
        conv_launch<C>(a, nullptr);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    CK(hipGetLastError());
    const double flop = 2.0 * macs_per_patch_tower * 3.0 * n;
    printf("%-36s n=%d  min %.3f ms  mean %.3f ms  %.1f TF/s (min)  LDS %zu B  TPW %d\n", name, n, best, sum / rounds,
           flop / (best * 1e-3) / 1e12, (size_t)C::LDS_BYTES, C::TPW);
    for (int t = 0; t < kTowers; ++t) {
        CK(hipFree((void*)a.in[t])); CK(hipFree((void*)a.w[t])); CK(hipFree((void*)a.bias[t]));
        CK(hipFree((void*)a.w1[t])); CK(hipFree(a.out[t]));
    }
}

template <class C>
static void stamps(const char* name, int n) {
    CK(conv_set_attr<C>());
    ConvArgs a{};
    a.n = n;
    const size_t in_per = (C::SRC == SRC_CONV1) ? 1024 : (size_t)C::CIN * C::HSRC * C::HSRC;
    for (int t = 0; t < kTowers; ++t) {
        a.in[t] = dev_random((size_t)n * in_per, 1 + t);
        a.w[t] = dev_random((size_t)C::NC4 * C::WCHUNK, 10 + t);
        a.bias[t] = dev_random(C::NT * 16, 20 + t);
        a.w1[t] = dev_random(kW1Floats, 30 + t);
        float* o; CK(hipMalloc(&o, (size_t)n * C::COUT * C::OUTPIX * 4)); a.out[t] = o;
    }
    const size_t nwg = (size_t)((n + C::G - 1) / C::G) * C::NB * kTowers;
    CK(hipMalloc(&a.stamps, nwg * 8 * sizeof(unsigned long long)));
    conv_launch<C>(a, nullptr);
    conv_launch<C>(a, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double d[4] = {0, 0, 0, 0}, in_wait = 0, epi_lds = 0, epi_bar = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nwg; ++w) {
        for (int i = 0; i < 4; ++i) d[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]);
        if (C::SRC == SRC_CONV1) in_wait += (double)(h[w * 8 + 5] - h[w * 8 + 0]);
        epi_lds += (double)(h[w * 8 + 6] - h[w * 8 + 3]);
        epi_bar += (double)(h[w * 8 + 7] - h[w * 8 + 6]);
        tmin = h[w * 8] < tmin ? h[w * 8] : tmin;
        tmax = h[w * 8 + 4] > tmax ? h[w * 8 + 4] : tmax;
    }
    printf("%-18s per-WG s_memtime ticks: stage %.0f | w0+abase+barrier %.0f | main loop %.0f | epilogue %.0f | total %.0f ; kernel span %.0f ; WGs %zu\n",
           name, d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg, (d[0] + d[1] + d[2] + d[3]) / nwg, (double)(tmax - tmin), nwg);
    printf("                     of stage: input-in-LDS at %.0f ; of epilogue: acc->LDS %.0f, barrier %.0f, copy-out %.0f\n", in_wait / nwg,
           epi_lds / nwg, epi_bar / nwg, (d[3] - epi_lds - epi_bar) / nwg);
}

// training-batch view: the forward (EPI_RAW_STATS) and data-gradient (SRC_NCHW_PAD2) configurations at n = batch size,
// where a launch is only a few rounds of workgroups and latency per workgroup matters more than steady-state rate
static void train_configs(int n, int rounds) {
    const double c2 = 2822400.0, c3 = 1036800.0, c4 = 1440000.0, c5 = 194400.0;
    for (int rep = 0; rep < 2; ++rep) {
        run<ConvCfg<20, 20, 30, 1, EPI_RAW, SRC_NCHW, 496, 2, 2>>("TF2 NB2 WPS2 (shipped)", n, rounds, c2);
        run<ConvCfg<20, 20, 30, 1, EPI_RAW, SRC_NCHW, 496, 2, 3>>("TF2 NB2 WPS3", n, rounds, c2);
        run<ConvCfg<20, 20, 30, 1, EPI_RAW, SRC_NCHW, 272, 4, 4>>("TF2 NB4 WPS4", n, rounds, c2);
        run<ConvCfg<20, 20, 30, 1, EPI_RAW, SRC_NCHW, 180, 7, 4>>("TF2 NB7 WPS4", n, rounds, c2);
        run<ConvCfg<20, 40, 14, 2, EPI_RAW, SRC_NCHW, 208, 1, 2>>("TF3 G2 WPS2 (shipped)", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 1, EPI_RAW, SRC_NCHW, 208, 1, 4>>("TF3 G1 WPS4", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 1, EPI_RAW, SRC_NCHW, 112, 2, 4>>("TF3 G1 NB2 WPS4", n, rounds, c3);
        run<ConvCfg<40, 40, 12, 1, EPI_RAW, SRC_NCHW, 144, 1, 4>>("TF4 G1 WPS4 (shipped)", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 1, EPI_RAW, SRC_NCHW, 112, 2, 4>>("TF4 G1 NB2 WPS4", n, rounds, c4);
        run<ConvCfg<40, 60, 5, 8, EPI_RAW, SRC_NCHW, 27, 1, 2>>("TF5 G8 WPS2 (shipped)", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 2, EPI_RAW, SRC_NCHW, 27, 1, 4>>("TF5 G2 WPS4", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 1, EPI_RAW, SRC_NCHW, 27, 1, 4>>("TF5 G1 WPS4", n, rounds, c5);
        run<ConvCfg<60, 40, 7, 8, EPI_RAW, SRC_NCHW_PAD2, 49, 1, 1>>("DG5 G8 WPS1 (shipped)", n, rounds, c5);
        run<ConvCfg<60, 40, 7, 2, EPI_RAW, SRC_NCHW_PAD2, 49, 1, 3>>("DG5 G2 WPS3", n, rounds, c5);
        run<ConvCfg<60, 40, 7, 1, EPI_RAW, SRC_NCHW_PAD2, 49, 1, 4>>("DG5 G1 WPS4", n, rounds, c5);
        run<ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_NCHW_PAD2, 208, 1, 3>>("DG4 NB1 WPS3 (shipped)", n, rounds, c4);
        run<ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_NCHW_PAD2, 112, 2, 4>>("DG4 NB2 WPS4", n, rounds, c4);
        run<ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_NCHW_PAD2, 84, 3, 4>>("DG4 NB3 WPS4", n, rounds, c4);
        run<ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_NCHW_PAD2, 144, 2, 4>>("DG3 NB2 WPS4 (shipped)", n, rounds, c3);
        run<ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_NCHW_PAD2, 80, 7, 4>>("DG3 NB7 WPS4", n, rounds, c3);
        run<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 544, 2, 3>>("DG2 NB2 WPS3 (shipped)", n, rounds, c2);
        run<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 384, 3, 4>>("DG2 NB3 WPS4", n, rounds, c2);
        run<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 256, 5, 4>>("DG2 NB5 WPS4", n, rounds, c2);
        run<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 224, 6, 4>>("DG2 NB6 WPS4", n, rounds, c2);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    if (argc > 3 && !strcmp(argv[3], "train")) { train_configs(n, rounds); return 0; }
    const double c12 = 162000.0 + 2822400.0, c3 = 1036800.0, c4 = 1440000.0, c5 = 194400.0;
    stamps<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 912, 1, 1, 1>>("conv12 NB1 WPS1", n);
    stamps<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2, 1>>("conv12 NB2 WPS2", n);
    stamps<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 180, 7, 4, 1>>("conv12 NB7 WPS4", n);
    stamps<ConvCfg<20, 40, 14, 4, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 1, 1>>("conv3 G4", n);
    stamps<ConvCfg<20, 40, 14, 1, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 4, 1>>("conv3 G1", n);
    stamps<ConvCfg<40, 40, 12, 1, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 4, 1>>("conv4 G1", n);
    for (int rep = 0; rep < 2; ++rep) {
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 912, 1, 1>>("conv12 NB1 WPS1 (912)", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 1>>("conv12 NB2 WPS1", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2>>("conv12 NB2 WPS2", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 180, 7, 4>>("conv12 NB7 WPS4", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 180, 7, 5>>("conv12 NB7 WPS5", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 180, 7, 6>>("conv12 NB7 WPS6", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 3>>("conv12 NB2 WPS3", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 3, 0, 5>>("conv12 NB2 WPS3 NW5", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2, 0, 5>>("conv12 NB2 WPS2 NW5", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2, 0, 7>>("conv12 NB2 WPS2 NW7", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 912, 1, 1, 0, 7>>("conv12 NB1 WPS1 NW7", n, rounds, c12);
        run<ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 912, 1, 1, 0, 14>>("conv12 NB1 WPS1 NW14", n, rounds, c12);
        run<ConvCfg<20, 40, 14, 4, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 1>>("conv3 G4 WPS1", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 2, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 3>>("conv3 G2 WPS3", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 2, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 2>>("conv3 G2 WPS2", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 1, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 4>>("conv3 G1 WPS4", n, rounds, c3);
        run<ConvCfg<40, 40, 12, 4, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 1>>("conv4 G4 WPS1", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 2, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 2>>("conv4 G2 WPS2", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 1, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 4>>("conv4 G1 WPS4", n, rounds, c4);
        run<ConvCfg<20, 40, 14, 4, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 1, 0, 12>>("conv3 G4 NW12", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 2, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 2, 0, 6>>("conv3 G2 NW6 WPS2", n, rounds, c3);
        run<ConvCfg<20, 40, 14, 4, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 1, 0, 16>>("conv3 G4 NW16", n, rounds, c3);
        run<ConvCfg<40, 40, 12, 4, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 1, 0, 16>>("conv4 G4 NW16", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 4, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 1, 0, 13>>("conv4 G4 NW13", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 2, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 2, 0, 7>>("conv4 G2 NW7 WPS2", n, rounds, c4);
        run<ConvCfg<40, 40, 12, 2, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 2, 0, 8>>("conv4 G2 NW8 WPS2", n, rounds, c4);
        run<ConvCfg<40, 60, 5, 16, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 1>>("conv5 G16 WPS1", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 8, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 2>>("conv5 G8 WPS2", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 4, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 4>>("conv5 G4 WPS4", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 7, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 3>>("conv5 G7 WPS3", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 6, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 3>>("conv5 G6 WPS3", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 14, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 2>>("conv5 G14 WPS2", n, rounds, c5);
        run<ConvCfg<40, 60, 5, 12, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 2>>("conv5 G12 WPS2", n, rounds, c5);
        printf("\n");
    }
    return 0;
}
