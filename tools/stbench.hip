// stbench - stand-alone timing of the split training forward kernel conv12_split_train_kernel (csrc/conv_split_train.h):
//   stbench [n] [workgroups per tower]      n patches x 3 towers, us per launch (best of 8), the separate-finalize form (no tickets)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSPT_ABLATE=k] -Isub-cortical_segmentation_amd/csrc tools/stbench.hip -o build/stbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "conv_split_train.h"
namespace scs { void set_error(const char*, ...) {} }
using namespace scs;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned g_seed = 99u;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xFFFFFF) / 16777216.0f - 0.5f; }
__global__ void pack_kernel(const float* W, uint4* wp, float* inv) { __shared__ float s_red[256]; pack_split_raw(W, 20, 20, wp, inv, s_red); }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, g = argc > 2 ? atoi(argv[2]) : 160;
    SplitTrain12Args a{};
    a.n = n;
    std::vector<float> x((size_t)n * 1024), w1(kSp12W1Floats, 0.f), w2(20 * 20 * 9), gam(20, 1.f);
    for (auto& v : x) v = 2.f * rnd();
    for (int s = 0; s < 12; ++s) for (int c = 0; c < 20; ++c) if ((s & 3) < 3) w1[s * 20 + c] = 0.6f * rnd() * kSplitActScale;
    for (int c = 0; c < 20; ++c) w1[240 + c] = 0.3f * rnd() * kSplitActScale;
    for (auto& v : w2) v = 0.15f * rnd();
    float *dW2, *dstats; unsigned* flag;
    CK(hipMalloc(&dW2, w2.size() * 4)); CK(hipMemcpy(dW2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dstats, (size_t)3 * n * 40 * 4)); CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
    a.stats = dstats; a.flag = flag;
    for (int t = 0; t < 3; ++t) {
        float *dx, *dw1, *dinv, *dg, *dz, *dp; uint4* dwp;
        CK(hipMalloc(&dx, x.size() * 4)); CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dw1, w1.size() * 4)); CK(hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dg, 80)); CK(hipMemcpy(dg, gam.data(), 80, hipMemcpyHostToDevice));
        CK(hipMalloc(&dinv, 16)); CK(hipMalloc(&dwp, (size_t)kSp12W2Vec * 16));
        CK(hipMalloc(&dz, (size_t)n * 20 * 784 * 4)); CK(hipMalloc(&dp, (size_t)n * 20 * 196 * 4));
        hipLaunchKernelGGL(pack_kernel, dim3(1), dim3(256), 0, nullptr, dW2, dwp, dinv);
        a.x0.views[t] = dx; a.w1[t] = dw1; a.w2[t] = dwp; a.inv2[t] = dinv; a.gamma2[t] = dg; a.out[t] = dz; a.pout[t] = dp;
    }
    CK(conv12_split_train_set_attr());
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0, nullptr)); conv12_split_train_launch(a, nullptr, g); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 2 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    unsigned hf; CK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
    printf("ablate %d: n %d, %d workgroups per tower: %.1f us per launch (overflow flag %u)\n", SPT_ABLATE, n, g < n ? g : n, best * 1000.f, hf);
    return 0;
}
