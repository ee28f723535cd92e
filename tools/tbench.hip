// tbench - micro-benchmark of the training-step kernels (conv forward / data gradient / weight gradient configurations of
// csrc/train_cfg.h and candidates) at the training batch size, one kernel at a time on random data, plus per-phase
// s_memtime stamps of diagnostic (STAMP) instantiations of the conv kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sub-cortical_segmentation_amd/csrc tools/tbench.hip -o build/tbench
// Outputs are not checked here (tests/ do that through the C ABI).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "train_cfg.h"

namespace scs { void set_error(const char*, ...) {} }
namespace scs {
// ---- experiment kept out of the product headers (profiles/r03_tbench_c5.txt): not faster than the per-patch kernel
// Weight gradient of a conv layer with SMALL planes (conv5: 5x5 -> 3x3) as a split-K GEMM with an on-the-fly im2col operand:
//   D[j = (c, tap)][o] = sum_{k = (n, pixel)}  X[n][c][pixel + tap] * dz[n][o][pixel]
// The address of X[k; j] separates into a k part and a j part (n * CIN * HIN^2 + py * HIN + px  +  c * HIN^2 + dy * HIN + dx), so
// every lane fetches its own MFMA operands straight from global memory (the latency form of gemm_mfma.h: dense_wgrad_kernel):
// no LDS staging, no per-patch loop - per patch the planes hold 9 output pixels, and staging them patch by patch left the
// matrix cores idle behind one global round trip per patch.  Workgroup = one 16-row tile of (c, tap) x all output channels
// (four 16-column tiles, columns >= COUT are zero) x one K slice; the four waves split the slice's k-steps, their partial
// tiles are added through LDS in wave order; the slices' partials go to wgrad_final_kernel like those of the other layers.
template <int CIN_, int COUT_, int HIN_>
struct WgSmallCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, HO = HIN_ - 2, PIX = HO * HO, J = CIN_ * 9, MT = (J + 15) / 16;
    static constexpr int NT = (COUT_ + 15) / 16, NB = 1;
    static constexpr int CH = 20;                       // k-steps in flight per wave and round
    static_assert(NT <= 4, "one workgroup covers all output channels");
};

template <class C>
__global__ __launch_bounds__(256, 2) void wgrad_small_kernel(WgArgs a) {
    __shared__ f32x4 s_r[4 * C::NT * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int mt = blockIdx.x, slice = blockIdx.y, tower = blockIdx.z;
    const int ktot = a.n * C::PIX;                                   // reduction length: (patch, output pixel) pairs
    const int steps = (ktot + 3) / 4, per_slice = (steps + a.slices - 1) / a.slices;
    const int sl_beg = min(steps, slice * per_slice), sl_end = min(steps, sl_beg + per_slice);
    const int per = (sl_end - sl_beg + 3) / 4;
    const int s_beg = min(sl_end, sl_beg + wave * per), s_end = min(sl_end, s_beg + per);
    const int j = min(mt * 16 + lr, C::J - 1), c = j / 9, tap = j - c * 9;
    const float* __restrict__ xcol = pick3(a.X, tower) + c * C::HIN * C::HIN + (tap / 3) * C::HIN + tap % 3;
    const float* __restrict__ zcol = pick3(a.dZ, tower) + lr * C::PIX;
    bool col_ok[C::NT];
#pragma unroll
    for (int q = 0; q < C::NT; ++q) col_ok[q] = q * 16 + lr < C::COUT;
    f32x4 acc[C::NT];
#pragma unroll
    for (int q = 0; q < C::NT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_beg; s0 < s_end; s0 += C::CH) {
        float av[C::CH], bv[C::CH][C::NT];
#pragma unroll
        for (int i = 0; i < C::CH; ++i) {
            const int k = 4 * (s0 + i) + kq;
            const bool on = (s0 + i) < s_end && k < ktot;
            const int kk = on ? k : 0, nn = kk / C::PIX, p = kk - nn * C::PIX, py = p / C::HO, px = p - py * C::HO;
            av[i] = on ? xcol[(size_t)nn * C::CIN * C::HIN * C::HIN + py * C::HIN + px] : 0.f;
#pragma unroll
            for (int q = 0; q < C::NT; ++q) bv[i][q] = (on && col_ok[q]) ? zcol[(size_t)nn * C::COUT * C::PIX + q * 16 * C::PIX + p] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < C::CH; ++i)
#pragma unroll
            for (int q = 0; q < C::NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][q], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < C::NT; ++q) s_r[(wave * C::NT + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (wave < C::NT) {
        const int q = wave;
        f32x4 v = s_r[q * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const f32x4 u = s_r[(w * C::NT + q) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        const int o = q * 16 + lr;
        float* __restrict__ out = a.part + ((size_t)tower * a.slices + slice) * C::COUT * C::J;      // [tower][slice][j][o]
        if (o < C::COUT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = mt * 16 + kq * 4 + r;
                if (jj < C::J) out[jj * C::COUT + o] = v[r];
            }
        }
    }
}

template <class C>
inline void wgrad_small_launch(const WgArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_small_kernel<C>, dim3(C::MT, a.slices, kTowers), dim3(256), 0, st, a);
}

}  // namespace scs
using namespace scs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_random(size_t n, unsigned seed, float lo = -1.f, float hi = 1.f) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * (((s >> 8) & 0xFFFF) / 65536.0f); }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

struct Pool { double* grp; float* in[3]; float* in2[3]; float* in3[3]; float* out[3]; float* out2[3]; float* coef[3]; float* fin[3]; unsigned int* counter; float* stats; float* part; size_t per; };
static Pool g_pool;
static void make_pool(int n) {
    g_pool.per = (size_t)n * 20 * 1024;
    for (int t = 0; t < 3; ++t) {
        g_pool.in[t] = dev_random(g_pool.per, 1 + t); g_pool.in2[t] = dev_random(g_pool.per, 11 + t); g_pool.in3[t] = dev_random(g_pool.per, 21 + t);
        CK(hipMalloc(&g_pool.out[t], g_pool.per * 4)); CK(hipMalloc(&g_pool.out2[t], g_pool.per * 4));
        g_pool.coef[t] = dev_random(4096, 31 + t, 0.2f, 1.2f);
        g_pool.fin[t] = dev_random(4096, 41 + t, 0.2f, 1.2f);      // finalize outputs of the FIN configurations
    }
    CK(hipMalloc(&g_pool.grp, 3 * fin_groups(n) * kFinRec * 2 * 8));
    CK(hipMalloc(&g_pool.counter, 3 * fin_counter_words(n) * 4)); CK(hipMemset(g_pool.counter, 0, 3 * fin_counter_words(n) * 4));
    CK(hipMalloc(&g_pool.stats, (size_t)3 * n * 32 * 256 * 4));      // up to 32 bands x 256 floats per patch and tower
    CK(hipMalloc(&g_pool.part, (size_t)3 * 1024 * 60 * 40 * 9 * 4));     // up to 1024 partials of the largest layer per tower
}

template <class F>
static void time_it(const char* name, int rounds, double flop, F&& launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms(rounds);
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, nullptr)); launch(); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-44s min %7.1f us  median %7.1f us  %6.1f TF/s (median)\n", name, ms[0] * 1e3, ms[rounds / 2] * 1e3, flop / (ms[rounds / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

template <class C>
static ConvArgs conv_args(int n) {
    static_assert(C::NB <= 32, "tbench statistics pool holds 32 bands per patch");
    ConvArgs a{};
    a.n = n;
    for (int t = 0; t < kTowers; ++t) {
        a.in[t] = g_pool.in[t]; a.w[t] = g_pool.in3[t]; a.bias[t] = g_pool.coef[t]; a.out[t] = g_pool.out[t];
        a.w1[t] = g_pool.coef[t] + 1024; a.bn[t] = g_pool.coef[t] + 512;
        a.xout[t] = (C::SRC == SRC_BN || C::SRC == SRC_BN_POOL) ? g_pool.out2[t] : nullptr;
        a.dup[t] = g_pool.in2[t]; a.zlow[t] = g_pool.in2[t]; a.bnlow[t] = g_pool.coef[t] + 3072;
        a.x0.views[t] = g_pool.in2[t];
        a.fin.bn[t] = g_pool.fin[t]; a.fin.ss[t] = g_pool.fin[t] + 1024; a.fin.coef[t] = g_pool.fin[t] + 2048;
    }
    a.fin.counter = g_pool.counter; a.fin.grp = g_pool.grp; a.fin.count = (double)n * C::PIX;
    a.stats = g_pool.stats;
    return a;
}

template <class C>
static void run_conv(const char* name, int n, int rounds, double macs_per_patch_tower, bool dz_out = false) {
    CK(conv_set_attr<C>());
    ConvArgs a = conv_args<C>(n);
    if (dz_out) for (int t = 0; t < kTowers; ++t) a.xout[t] = g_pool.out2[t];
    char buf[128]; snprintf(buf, sizeof buf, "%s [LDS %zu]", name, (size_t)C::LDS_BYTES);
    time_it(buf, rounds, 2.0 * macs_per_patch_tower * 3.0 * n, [&] { conv_launch<C>(a, nullptr); });
}

template <class C>
static void stamps(const char* name, int n) {
    CK(conv_set_attr<C>());
    ConvArgs a = conv_args<C>(n);
    const size_t nwg = (size_t)((n + C::G - 1) / C::G) * C::NB * kTowers;
    CK(hipMalloc(&a.stamps, nwg * 8 * sizeof(unsigned long long)));
    conv_launch<C>(a, nullptr);
    conv_launch<C>(a, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double d[4] = {0, 0, 0, 0}, epi_lds = 0, epi_bar = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nwg; ++w) {
        for (int i = 0; i < 4; ++i) d[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]);
        epi_lds += (double)(h[w * 8 + 6] - h[w * 8 + 3]);
        epi_bar += (double)(h[w * 8 + 7] - h[w * 8 + 6]);
        tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + 4]);
    }
    // device-side span per XCD (s_memtime is an XCD-local clock; workgroups b and b + 8 share an XCD): first start -> last end
    double span = 0, wg_max = 0;
    for (int x = 0; x < 8; ++x) {
        unsigned long long lo = ~0ull, hi = 0;
        for (size_t w = x; w < nwg; w += 8) { lo = std::min(lo, h[w * 8]); hi = std::max(hi, h[w * 8 + 4]); wg_max = std::max(wg_max, (double)(h[w * 8 + 4] - h[w * 8])); }
        if (hi > lo) span = std::max(span, (double)(hi - lo));
    }
    printf("%-28s device span (max over XCDs) %.0f cycles, slowest workgroup %.0f cycles\n", name, span, wg_max);
    printf("%-28s per-WG ticks (100 MHz): stage %.0f | w0+barrier %.0f | main loop %.0f | epilogue %.0f (acc->LDS %.0f, barrier %.0f, rest %.0f) | total %.0f ; kernel span %.0f ; WGs %zu\n",
           name, d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg, epi_lds / nwg, epi_bar / nwg, (d[3] - epi_lds - epi_bar) / nwg,
           (d[0] + d[1] + d[2] + d[3]) / nwg, (double)(tmax - tmin), nwg);
    CK(hipFree(a.stamps));
}

template <class C>
static void run_wg(const char* name, int n, int slices, int rounds, double macs_per_patch_tower) {
    CK(wgrad_set_attr<C>());
    WgArgs a{};
    for (int t = 0; t < kTowers; ++t) {
        a.X[t] = g_pool.in[t]; a.w1[t] = g_pool.coef[t] + 1024; a.dZ[t] = g_pool.in2[t]; a.x0.views[t] = g_pool.in3[t];
    }
    a.part = g_pool.part; a.n = n; a.slices = slices;
    if ((size_t)slices * C::NB > 1024) { printf("%s: too many partials for the tbench pool\n", name); return; }
    char buf[128]; snprintf(buf, sizeof buf, "%s slices %d [LDS %zu]", name, slices, (size_t)C::LDS_BYTES);
    time_it(buf, rounds, 2.0 * macs_per_patch_tower * 3.0 * n, [&] { wgrad_launch<C>(a, nullptr); });
}

template <class C>
static void run_wg_small(const char* name, int n, int slices, int rounds, double macs_per_patch_tower) {
    WgArgs a{};
    for (int t = 0; t < kTowers; ++t) { a.X[t] = g_pool.in[t]; a.dZ[t] = g_pool.in2[t]; }
    a.part = g_pool.part; a.n = n; a.slices = slices;
    char buf[128]; snprintf(buf, sizeof buf, "%s K slices %d", name, slices);
    time_it(buf, rounds, 2.0 * macs_per_patch_tower * 3.0 * n, [&] { wgrad_small_launch<C>(a, nullptr); });
}

// a latency-bound stand-in for the tiny finalize launches between the conv kernels: `blocks` workgroups, each a chain of
// `hops` dependent loads
__global__ void small_kernel(const int* __restrict__ chain, int hops, int* out) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    for (int h = 0; h < hops; ++h) i = chain[i & 4095];
    if (i == -1) out[0] = i;
}

// Forward conv chain [TF12, small, TF3, small, TF4, small, TF5, small] either as one stream with all three towers per launch,
// or as three per-tower streams (optionally with descending priorities so that the towers run out of phase)
static void chain_experiment(int n, int rounds) {
    CK(conv_set_attr<TF12>()); CK(conv_set_attr<TF3>()); CK(conv_set_attr<TF4>()); CK(conv_set_attr<TF5>());
    int* chain; CK(hipMalloc(&chain, 4096 * 4));
    { std::vector<int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 97 + 13) & 4095; CK(hipMemcpy(chain, h.data(), 4096 * 4, hipMemcpyHostToDevice)); }
    int* sink; CK(hipMalloc(&sink, 4));
    ConvArgs a12 = conv_args<TF12>(n), a3 = conv_args<TF3>(n), a4 = conv_args<TF4>(n), a5 = conv_args<TF5>(n);
    auto chain_on = [&](hipStream_t st, int t0, int nt) {
        ConvArgs b;
        b = a12; b.tower0 = t0; conv_launch<TF12>(b, st, nt); hipLaunchKernelGGL(small_kernel, dim3(20 * nt), dim3(256), 0, st, chain, 6, sink);
        b = a3; b.tower0 = t0; conv_launch<TF3>(b, st, nt); hipLaunchKernelGGL(small_kernel, dim3(40 * nt), dim3(256), 0, st, chain, 6, sink);
        b = a4; b.tower0 = t0; conv_launch<TF4>(b, st, nt); hipLaunchKernelGGL(small_kernel, dim3(40 * nt), dim3(256), 0, st, chain, 6, sink);
        b = a5; b.tower0 = t0; conv_launch<TF5>(b, st, nt); hipLaunchKernelGGL(small_kernel, dim3(60 * nt), dim3(256), 0, st, chain, 6, sink);
    };
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t sp[3], sq[3];
    for (int t = 0; t < 3; ++t) {
        CK(hipStreamCreateWithFlags(&sq[t], hipStreamNonBlocking));
        const int pr = hi + t <= lo ? hi + t : lo;
        CK(hipStreamCreateWithPriority(&sp[t], hipStreamNonBlocking, pr));
    }
    hipEvent_t e0, e1, ej[3]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int t = 0; t < 3; ++t) CK(hipEventCreateWithFlags(&ej[t], hipEventDisableTiming));
    auto timed = [&](const char* name, auto&& body) {
        std::vector<float> ms(rounds);
        for (int r = 0; r < rounds + 3; ++r) {
            CK(hipEventRecord(e0, s0));
            body();
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            if (r >= 3) CK(hipEventElapsedTime(&ms[r - 3], e0, e1));
        }
        std::sort(ms.begin(), ms.end());
        printf("%-60s min %7.1f us  median %7.1f us\n", name, ms[0] * 1e3, ms[rounds / 2] * 1e3); fflush(stdout);
    };
    timed("forward conv chain, one stream, 3 towers per launch", [&] { chain_on(s0, 0, 3); });
    auto fanout = [&](hipStream_t* ss) {
        hipEvent_t ef; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CK(hipEventRecord(ef, s0));
        for (int t = 0; t < 3; ++t) { CK(hipStreamWaitEvent(ss[t], ef, 0)); chain_on(ss[t], t, 1); CK(hipEventRecord(ej[t], ss[t])); }
        for (int t = 0; t < 3; ++t) CK(hipStreamWaitEvent(s0, ej[t], 0));
        CK(hipEventDestroy(ef));
    };
    timed("forward conv chain, 3 per-tower streams (equal priority)", [&] { fanout(sq); });
    timed("forward conv chain, 3 per-tower streams (priorities)", [&] { fanout(sp); });
    timed("forward conv chain, one stream, 3 towers per launch (again)", [&] { chain_on(s0, 0, 3); });
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    const char* what = argc > 3 ? argv[3] : "all";
    make_pool(n);
    const double c1 = 162000.0, c2 = 2822400.0, c3 = 1036800.0, c4 = 1440000.0, c5 = 194400.0;
    const bool all = !strcmp(what, "all");
    if (!strcmp(what, "chain")) { chain_experiment(n, rounds); return 0; }
    if (all || !strcmp(what, "fwd")) {
        run_conv<TF12>("TF12 (shipped)", n, rounds, c1 + c2);
        run_conv<TF3>("TF3 (shipped)", n, rounds, c3);
        run_conv<TF4>("TF4 (shipped)", n, rounds, c4);
        run_conv<TF5>("TF5 (shipped)", n, rounds, c5);
    }
    if (!strcmp(what, "dg2")) {
        // the dominant kernel of the step (conv2 data gradient + layer-1 sums): three bands of 10 rows against the shipped five of 6
        run_conv<DG2>("DG2 (shipped: 5 bands, WPS4)", n, rounds, c2);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 384, 3, 3>>("DG2 3 bands, PLANE 384, WPS3", n, rounds, c2);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 400, 3, 3>>("DG2 3 bands, PLANE 400, WPS3", n, rounds, c2);
        run_conv<DG2>("DG2 (shipped: 5 bands, WPS4)", n, rounds, c2);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 384, 3, 3>>("DG2 3 bands, PLANE 384, WPS3", n, rounds, c2);
        run_conv<TF12>("TF12 (shipped: 7 bands, WPS4)", n, rounds, c1 + c2);
        return 0;
    }
    if (!strcmp(what, "band")) {
        // two workgroups per patch-tower (row bands) against one: data-gradient kernels whose staging already takes bands
        run_conv<DG4>("DG4 (shipped)", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 112, 2, 4, 0, 4, 1>>("DG4 2 bands, fin, WPS4", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 112, 2, 3, 0, 4, 1>>("DG4 2 bands, fin, WPS3", n, rounds, c4);
        run_conv<TF3>("TF3 (shipped)", n, rounds, c3);
        run_conv<ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN_POOL, 112, 2, 4, 0, 4, 1>>("TF3 2 bands, fin, WPS4", n, rounds, c3);
        run_conv<ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN_POOL, 112, 2, 3, 0, 4, 1>>("TF3 2 bands, fin, WPS3", n, rounds, c3);
        run_conv<TF4>("TF4 (shipped)", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 84, 2, 4, 0, 4, 1>>("TF4 2 bands, fin, WPS4", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 84, 2, 3, 0, 4, 1>>("TF4 2 bands, fin, WPS3", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 100, 2, 4, 0, 4, 1>>("TF4 2 bands, PLANE 100", n, rounds, c4);
        run_conv<DG3>("DG3 (shipped)", n, rounds, c3);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 144, 2, 4, 0, 4, 1>>("DG3 2 bands, fin, WPS4", n, rounds, c3);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 144, 2, 3, 0, 4, 1>>("DG3 2 bands, fin, WPS3", n, rounds, c3);
        return 0;
    }
    if (all || !strcmp(what, "dg")) {
        run_conv<DG5>("DG5 (shipped)", n, rounds, c5);
        run_conv<DG4>("DG4 (shipped)", n, rounds, c4);
        run_conv<DG3>("DG3 (shipped)", n, rounds, c3);
        run_conv<DG2>("DG2 (shipped)", n, rounds, c2);
        // the same tilings with plain staging / plain epilogue: what the fused BN work costs
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_NCHW_PAD2, 208, 1, 3>>("DG4 plain", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_DZ_POOL_PAD2, 208, 1, 3>>("DG4 dz staging, plain epilogue", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_NCHW_PAD2, 208, 1, 3>>("DG4 plain staging, stats epilogue", n, rounds, c4);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_NCHW_PAD2, 144, 2, 4>>("DG3 plain", n, rounds, c3);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_DZ_PAD2, 144, 2, 4>>("DG3 dz staging, plain epilogue", n, rounds, c3);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_NCHW_PAD2, 144, 2, 4>>("DG3 plain staging, stats epilogue", n, rounds, c3);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 256, 5, 4>>("DG2 plain", n, rounds, c2);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_DZ_POOL_PAD2, 256, 5, 4>>("DG2 dz staging, plain epilogue", n, rounds, c2);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_NCHW_PAD2, 256, 5, 4>>("DG2 plain staging, L1 epilogue", n, rounds, c2);
        run_conv<DG2>("DG2 + d z stored", n, rounds, c2, true);
        run_conv<DG3>("DG3 + d z stored", n, rounds, c3, true);
        run_conv<DG4>("DG4 + d z stored", n, rounds, c4, true);
        run_conv<DG5>("DG5 + d z stored", n, rounds, c5, true);
    }
    if (!strcmp(what, "c5")) {
        // the conv5 block as K-split GEMM-style kernels: patches per workgroup x K parts (WPS 1: up to 256 registers, one
        // workgroup of 6+ waves per CU; WPS 2: 168 registers, two per CU)
        run_conv<TF5>("TF5 (shipped)", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 2, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 6, 0, 2>>("TF5 G2 KS2 WPS1", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 3, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 6, 0, 2>>("TF5 G3 KS2 WPS1", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 6, 0, 2>>("TF5 G4 KS2 WPS1", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 2, 0, 6, 0, 2>>("TF5 G4 KS2 WPS2", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 7, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 6, 0, 2>>("TF5 G7 KS2 WPS1", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 15, 0, 5>>("TF5 G4 KS5 WPS1", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 4>>("TF5 one patch, tiles over waves (round 2 form)", n, rounds, c5);
        run_conv<DG5>("DG5 (shipped)", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 3, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 1, 0, 6, 0, 3>>("DG5 G3 KS3 WPS1 no FIN", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 0, 4, 1>>("DG5 one patch, tiles over waves (FIN)", n, rounds, c5, true);
        for (int sl : {4, 8, 16, 32}) run_wg_small<WgSmallCfg<40, 60, 5>>("WG5 split-K GEMM", n, sl, rounds, c5);
        run_wg<WgCfg<40, 60, 5, 1, 2, 0>>("WG5 per-patch staging (round 2 form)", n, 64, rounds, c5);
        return 0;
    }
    if (!strcmp(what, "c5stamps")) {
        stamps<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 1, 6, 0, 2>>("TF5 G4 KS2", n);
        stamps<ConvCfg<40, 60, 5, 3, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 1, 6, 0, 2>>("TF5 G3 KS2", n);
        stamps<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 1, 15, 0, 5>>("TF5 G4 KS5", n);
        stamps<ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 4, 1>>("TF5 one patch, tiles over waves", n);
        stamps<ConvCfg<60, 40, 7, 3, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 1, 1, 6, 0, 3>>("DG5 G3 KS3", n);
        stamps<ConvCfg<60, 40, 7, 4, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 1, 1, 6, 0, 3>>("DG5 G4 KS3", n);
        stamps<ConvCfg<60, 40, 7, 2, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 1, 1, 6, 0, 3>>("DG5 G2 KS3", n);
        stamps<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 1>>("DG5 one patch, tiles over waves", n);
        stamps<ConvCfg<60, 40, 7, 1, EPI_RAW, SRC_NCHW_PAD2, 49, 1, 4, 1>>("DG5 one patch, plain staging + epilogue", n);
        return 0;
    }
    if (!strcmp(what, "fin")) {
        // what the in-kernel finalize (last workgroup of a tower) adds to the statistics kernels
        run_conv<TF3>("TF3 (shipped, FIN)", n, rounds, c3);
        run_conv<ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN_POOL, 208, 1, 4>>("TF3 without finalize", n, rounds, c3);
        run_conv<TF4>("TF4 (shipped, FIN)", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 144, 1, 4>>("TF4 without finalize", n, rounds, c4);
        run_conv<DG5>("DG5 (shipped, FIN)", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4>>("DG5 without finalize", n, rounds, c5, true);
        run_conv<DG4>("DG4 (shipped, FIN)", n, rounds, c4, true);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 208, 1, 3>>("DG4 without finalize", n, rounds, c4, true);
        run_conv<DG3>("DG3 (shipped, FIN)", n, rounds, c3, true);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 256, 1, 3>>("DG3 without finalize", n, rounds, c3, true);
        return 0;
    }
    if (!strcmp(what, "plane")) {
        // LDS plane strides: the four k lanes of an A operand read planes 4c..4c+3, a stride of 16 mod 32 floats keeps the
        // two planes of a half-wave on disjoint banks
        run_conv<DG3>("DG3 (shipped)", n, rounds, c3, true);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 272, 1, 3, 0, 4, 1>>("DG3 PLANE 272", n, rounds, c3, true);
        run_conv<DG2>("DG2 (shipped)", n, rounds, c2, true);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 272, 5, 4>>("DG2 PLANE 272", n, rounds, c2, true);
        run_conv<TF12>("TF12 (shipped)", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 208, 7, 4>>("TF12 PLANE 208", n, rounds, c1 + c2);
        run_conv<TF5>("TF5 (shipped)", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_BN_POOL, 48, 1, 4>>("TF5 PLANE 48", n, rounds, c5);
        run_conv<DG5>("DG5 (shipped)", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 80, 1, 4, 0, 4, 1>>("DG5 PLANE 80", n, rounds, c5, true);
        return 0;
    }
    if (!strcmp(what, "ns")) {
        // conv5 block (one / two M tiles per patch): tiles dealt to waves + weight chunks through LDS, against the waves
        // splitting the output channels with operands straight from L2 (NSPLIT)
        run_conv<ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 4>>("TF5 tiles over waves", n, rounds, c5);
        run_conv<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 0, 4, 1>>("DG5 tiles over waves (FIN)", n, rounds, c5, true);
        return 0;
    }
    if (!strcmp(what, "wgs")) {
        // weight-gradient kernels with and without the software-pipelined staging, over the number of batch slices
        for (int sl : {64, 32}) run_wg<WgCfg<20, 20, 30, 4, 3, 1>>("WG2", n, sl, rounds, c2);
        for (int sl : {64, 32}) run_wg<WgCfg<20, 20, 30, 4, 3, 1, 1>>("WG2 pipelined", n, sl, rounds, c2);
        for (int sl : {128, 64, 32}) run_wg<WgCfg<20, 40, 14, 2, 3, 0>>("WG3", n, sl, rounds, c3);
        for (int sl : {128, 64, 32}) run_wg<WgCfg<20, 40, 14, 2, 3, 0, 1>>("WG3 pipelined", n, sl, rounds, c3);
        for (int sl : {64, 32, 16}) run_wg<WgCfg<40, 40, 12, 5, 3, 0>>("WG4", n, sl, rounds, c4);
        for (int sl : {64, 32, 16}) run_wg<WgCfg<40, 40, 12, 5, 2, 0, 1>>("WG4 pipelined WPC2", n, sl, rounds, c4);
        for (int sl : {128, 64, 32}) run_wg<WgCfg<40, 60, 5, 1, 2, 0>>("WG5", n, sl, rounds, c5);
        for (int sl : {128, 64, 32}) run_wg<WgCfg<40, 60, 5, 1, 2, 0, 1>>("WG5 pipelined", n, sl, rounds, c5);
        return 0;
    }
    if (!strcmp(what, "sweep")) {
        run_conv<TF12>("TF12 NB7 WPS4 (shipped)", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 180, 7, 6>>("TF12 NB7 WPS6", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 272, 4, 4>>("TF12 NB4 WPS4", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 496, 2, 3>>("TF12 NB2 WPS3", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 496, 2, 2>>("TF12 NB2 WPS2", n, rounds, c1 + c2);
        run_conv<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 120, 14, 6>>("TF12 NB14 WPS6", n, rounds, c1 + c2);
        run_conv<DG2>("DG2 NB5 WPS4 (shipped)", n, rounds, c2, true);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 384, 3, 3>>("DG2 NB3 WPS3", n, rounds, c2, true);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 384, 3, 4>>("DG2 NB3 WPS4", n, rounds, c2, true);
        run_conv<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 128, 15, 6>>("DG2 NB15 WPS6", n, rounds, c2, true);
        run_conv<DG3>("DG3 NB2 WPS4 (shipped)", n, rounds, c3, true);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 256, 1, 3>>("DG3 NB1 WPS3", n, rounds, c3, true);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 144, 2, 5>>("DG3 NB2 WPS5", n, rounds, c3, true);
        run_conv<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 64, 7, 6>>("DG3 NB7 WPS6", n, rounds, c3, true);
        run_conv<DG4>("DG4 NB1 WPS3 (shipped)", n, rounds, c4, true);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 112, 2, 4>>("DG4 NB2 WPS4", n, rounds, c4, true);
        run_conv<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 84, 3, 4>>("DG4 NB3 WPS4", n, rounds, c4, true);
        run_conv<DG5>("DG5 G2 WPS3 (shipped)", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4>>("DG5 G1 WPS4", n, rounds, c5, true);
        run_conv<ConvCfg<60, 40, 7, 4, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 2>>("DG5 G4 WPS2", n, rounds, c5, true);
        run_conv<TF5>("TF5 G1 WPS4 (shipped)", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 2, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 4>>("TF5 G2 WPS4", n, rounds, c5);
        run_conv<ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 3>>("TF5 G4 WPS3", n, rounds, c5);
        run_conv<TF3>("TF3 G1 WPS4 (shipped)", n, rounds, c3);
        run_conv<ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN_POOL, 208, 1, 3>>("TF3 G1 WPS3", n, rounds, c3);
        run_conv<ConvCfg<20, 40, 14, 2, EPI_RAW_STATS, SRC_BN_POOL, 208, 1, 3>>("TF3 G2 WPS3", n, rounds, c3);
        run_conv<TF4>("TF4 G1 WPS4 (shipped)", n, rounds, c4);
        run_conv<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 144, 1, 3>>("TF4 G1 WPS3", n, rounds, c4);
        run_wg<WG2>("WG2 (shipped)", n, 64, rounds, c2);
        run_wg<WG2>("WG2 (shipped)", n, 32, rounds, c2);
        run_wg<WgCfg<20, 20, 30, 7, 4, 1>>("WG2 NB7 WPC4", n, 64, rounds, c2);
        run_wg<WgCfg<20, 20, 30, 7, 4, 1>>("WG2 NB7 WPC4", n, 32, rounds, c2);
        run_wg<WgCfg<20, 20, 30, 2, 2, 1>>("WG2 NB2 WPC2", n, 128, rounds, c2);
        run_wg<WG3>("WG3 (shipped)", n, 128, rounds, c3);
        run_wg<WG3>("WG3 (shipped)", n, 64, rounds, c3);
        run_wg<WgCfg<20, 40, 14, 1, 3, 0>>("WG3 NB1", n, 256, rounds, c3);
        run_wg<WgCfg<20, 40, 14, 3, 4, 0>>("WG3 NB3 WPC4", n, 128, rounds, c3);
        run_wg<WG4>("WG4 (shipped)", n, 128, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 2, 2, 0>>("WG4 NB2 WPC2", n, 128, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 5, 3, 0>>("WG4 NB5 WPC3", n, 64, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 5, 2, 0>>("WG4 NB5 WPC2", n, 64, rounds, c4);
        run_wg<WG5>("WG5 (shipped)", n, 64, rounds, c5);
        run_wg<WG5>("WG5 (shipped)", n, 128, rounds, c5);
        run_wg<WG5>("WG5 (shipped)", n, 32, rounds, c5);
        return 0;
    }
    if (all || !strcmp(what, "stamps")) {
        stamps<ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 256, 5, 4, 1>>("DG2", n);
        stamps<ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 256, 5, 4, 1>>("DG2 plain", n);
        stamps<ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 256, 1, 3, 1>>("DG3", n);
        stamps<ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_NCHW_PAD2, 256, 1, 3, 1>>("DG3 plain", n);
        stamps<ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 208, 1, 3, 1>>("DG4", n);
        stamps<ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 180, 7, 4, 1>>("TF12", n);
        stamps<ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN_POOL, 208, 1, 4, 1>>("TF3", n);
        stamps<ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 144, 1, 4, 1>>("TF4", n);
        stamps<ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 4, 1>>("TF5 tiles over waves", n);
        stamps<ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 1>>("DG5 tiles over waves", n);
    }
    if (all || !strcmp(what, "wg")) {
        run_wg<WG2>("WG2 (shipped)", n, 64, rounds, c2);
        run_wg<WG3>("WG3 (shipped)", n, 128, rounds, c3);
        run_wg<WG4>("WG4 (shipped)", n, 256, rounds, c4);
        run_wg<WG5>("WG5 (shipped)", n, 64, rounds, c5);
        run_wg<WgCfg<20, 20, 30, 4, 3, 0>>("WG2 plain (X, dz from memory)", n, 64, rounds, c2);
        run_wg<WgCfg<20, 20, 30, 4, 3, 1>>("WG2 conv1 recompute, dz from memory", n, 64, rounds, c2);
        run_wg<WgCfg<20, 40, 14, 2, 3, 0>>("WG3 plain GP1", n, 128, rounds, c3);
        run_wg<WgCfg<40, 40, 12, 1, 3, 0>>("WG4 plain", n, 256, rounds, c4);
        run_wg<WgCfg<40, 60, 5, 1, 2, 0>>("WG5 plain GP1", n, 64, rounds, c5);
        run_wg<WgCfg<40, 60, 5, 1, 3, 0>>("WG5 plain GP1 WPC3", n, 64, rounds, c5);
        run_wg<WgCfg<40, 40, 12, 2, 3, 0>>("WG4 plain NB2 GP1", n, 128, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 2, 3, 0>>("WG4 plain NB2 GP1", n, 256, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 1, 3, 0>>("WG4 plain", n, 128, rounds, c4);
        run_wg<WgCfg<40, 40, 12, 1, 2, 0>>("WG4 plain WPC2", n, 256, rounds, c4);
    }
    return 0;
}
