// split_conv_probe - research prototype for the next round (NOT part of libscs): conv3 of the towers (20 -> 40 channels,
// 14x14 -> 12x12, bias + ReLU) computed with the error-free bf16x6 split on v_mfma_f32_16x16x32_bf16.
//
//   * one persistent 9-wave workgroup per CU; the split weights (7 k-blocks x 3 column tiles x 3 pieces, 63 KB, MFMA lane
//     order) stay resident in LDS; each iteration stages 3 patches (fp32 NCHW from global) as three bf16 planes with the
//     channel innermost ([patch][pixel][24 channels], 16-byte reads of 8 channels of one tap), splitting on the fly;
//   * 27 output tiles of 16 pixels per iteration = 3 per wave, 6 MFMAs per (tile, column tile, k-block);
//   * no overlap of staging and matrix phases yet (single activation buffer) - a lower bound.
// Prints the error against a double-precision CPU convolution and the rate next to what libscs' fp32 conv3 reaches
// (0.54 ms for 8192 patches x 3 towers = 45.5 M patch-towers/s, profiles/r01_kbench_final.txt).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/split_conv_probe.hip -o build/split_conv_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CIN = 20, CP = 24, COUT = 40, HIN = 14, HO = 12, PIN = HIN * HIN, POUT = HO * HO;
constexpr int G = 3, NW = 9, NTH = 64 * NW, TILES = G * POUT / 16, TPW = TILES / NW;          // 27 tiles, 3 per wave
constexpr int NGRP = 9 * (CP / 8), NBLK = (NGRP + 3) / 4, NT = 3;                            // 27 groups of 8 k, 7 blocks
constexpr int W_V = NBLK * NT * 3 * 64;                                                       // uint4 units (63 KB)
constexpr int PIECE_BYTES = G * PIN * CP * 2;                                                 // one bf16 plane of the 3 patches
constexpr size_t LDS_BYTES = (size_t)W_V * 16 + 3 * PIECE_BYTES;
static_assert(TILES % NW == 0 && (G * POUT) % 16 == 0, "tiling");

__host__ __device__ inline unsigned bf16_rne_bits(float x) {       // upper 16 bits of x rounded to nearest even (finite x)
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}
__host__ __device__ inline float bits_f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

#ifndef PROBE_PRESPLIT
#define PROBE_PRESPLIT 0          // 1: the input arrives already split ([patch][piece][pixel][24] bf16, what a split-aware
#endif                            //    producer epilogue would write): staging is plain 16-byte copies
__global__ __launch_bounds__(NTH, 1) void split_conv3(const float* __restrict__ in, const uint4* __restrict__ in_split, const uint4* __restrict__ wpk,
                                                      const float* __restrict__ bias, float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_w = reinterpret_cast<uint4*>(smem);
    unsigned char* s_a = smem + (size_t)W_V * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, kq = lane >> 4;
    for (int e = tid; e < W_V; e += NTH) s_w[e] = wpk[e];
    for (int e = tid; e < 3 * PIECE_BYTES / 4; e += NTH) reinterpret_cast<unsigned*>(s_a)[e] = 0u;   // channel padding stays zero
    // per-lane offsets of the k-groups this lane feeds (bytes): group = 4*block + kq -> (tap, 8-channel group)
    int goff[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int gi = 4 * b + kq, tap = gi / 3, cg = gi - tap * 3;
        goff[b] = gi < NGRP ? ((tap / 3) * HIN + tap % 3) * CP * 2 + cg * 16 : 0;             // padded groups: weights are zero
    }
    int tbase[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int m = (wave + NW * j) * 16 + lr, g = m / POUT, r = m - g * POUT, y = r / HO, x = r - y * HO;
        tbase[j] = (g * PIN + y * HIN + x) * CP * 2;
    }
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (nt * 16 + lr < COUT) ? bias[nt * 16 + lr] : 0.f;
    const int units = (n + G - 1) / G;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        __syncthreads();                                             // previous iteration's reads are done
        const int p0 = u * G, np = min(G, n - p0);
#ifndef PROBE_ABLATE
#define PROBE_ABLATE 0            // 1: stage only the first unit, 2: no MFMAs, 3: no output stores (timing only)
#endif
        // stage: fp32 NCHW -> three bf16 planes, pixel-major / channel-innermost
        if (PROBE_PRESPLIT) {
            constexpr int V = PIN * CP * 2 / 16;                      // uint4 per patch and piece (588)
            constexpr int ITER = (G * 3 * V + NTH - 1) / NTH;         // all of a lane's loads in flight, then the LDS writes
            uint4 v[ITER];
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int e = min(tid + NTH * i, G * 3 * V - 1), g = e / (3 * V), r = e - g * 3 * V;
                v[i] = in_split[((size_t)(p0 + min(g, np - 1)) * 3) * V + r];
            }
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int e = tid + NTH * i, g = e / (3 * V), r = e - g * 3 * V, p = r / V, q = r - p * V;
                if (e < G * 3 * V && g < np) reinterpret_cast<uint4*>(s_a + p * PIECE_BYTES + g * PIN * CP * 2)[q] = v[i];
            }
        } else
        if (PROBE_ABLATE != 1 || u == (int)blockIdx.x)
        for (int e = tid; e < G * CIN * PIN; e += NTH) {
            const int g = e / (CIN * PIN), r = e - g * (CIN * PIN), c = r / PIN, px = r - c * PIN;
            const float x = g < np ? in[(size_t)(p0 + g) * CIN * PIN + r] : 0.f;
            const unsigned a1 = bf16_rne_bits(x);
            const float r1 = x - bits_f(a1);
            const unsigned a2 = bf16_rne_bits(r1);
            const unsigned a3 = bf16_rne_bits(r1 - bits_f(a2));
            unsigned short* d = reinterpret_cast<unsigned short*>(s_a) + (g * PIN + px) * CP + c;
            d[0] = (unsigned short)(a1 >> 16);
            d[PIECE_BYTES / 2] = (unsigned short)(a2 >> 16);
            d[PIECE_BYTES] = (unsigned short)(a3 >> 16);
        }
        __syncthreads();
        f32x4 acc[TPW][NT];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            bf16x8 a[3][TPW], w[3][NT];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) { const uint4 v = *reinterpret_cast<const uint4*>(s_a + p * PIECE_BYTES + tbase[j] + goff[b]); a[p][j] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { const uint4 v = s_w[((b * NT + nt) * 3 + p) * 64 + lane]; w[p][nt] = *reinterpret_cast<const bf16x8*>(&v); }
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 c = acc[j][nt];                            // smallest products first
                    if (PROBE_ABLATE == 2) { c[0] += (float)a[0][j][0] * (float)w[0][nt][0] + (float)a[1][j][1] * (float)w[2][nt][3] + (float)a[2][j][2] * (float)w[1][nt][5]; acc[j][nt] = c; continue; }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][j], w[0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][j], w[1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[2][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][j], w[0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[0][nt], c, 0, 0, 0);
                    acc[j][nt] = c;
                }
        }
        // epilogue: D[row = pixel 4kq + r][col = cout lr] -> NCHW, bias + ReLU
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int o = nt * 16 + lr;
                if (o >= COUT) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = (wave + NW * j) * 16 + 4 * kq + r, g = m / POUT, pix = m - g * POUT;
                    if (g < np && (PROBE_ABLATE != 3 || acc[j][nt][r] == 123.456f))
                        out[((size_t)(p0 + g) * COUT + o) * POUT + pix] = fmaxf(acc[j][nt][r] + bv[nt], 0.f);
                }
            }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192 * 3;
    std::vector<float> h_in((size_t)n * CIN * PIN), h_w(COUT * CIN * 9), h_b(COUT);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f; };
    for (auto& v : h_in) v = fmaxf(0.f, 2.5f * rnd() - 1.0f);                  // ReLU-like activations
    for (auto& v : h_w) v = 0.2f * (rnd() - 0.5f);
    for (auto& v : h_b) v = 0.1f * (rnd() - 0.5f);
    // pack the split weights in MFMA lane order: [block][column tile][piece][lane] x 8 bf16 (k = 8 channels of one tap)
    std::vector<unsigned short> h_pk((size_t)W_V * 8, 0);
    for (int b = 0; b < NBLK; ++b)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int lr = lane & 15, kq = lane >> 4, gi = 4 * b + kq, tap = gi / 3, cg = gi % 3, c = cg * 8 + j, o = nt * 16 + lr;
                    const float wv = (gi < NGRP && c < CIN && o < COUT) ? h_w[(o * CIN + c) * 9 + tap] : 0.f;
                    const unsigned a1 = bf16_rne_bits(wv);
                    const float r1 = wv - bits_f(a1);
                    const unsigned a2 = bf16_rne_bits(r1), a3 = bf16_rne_bits(r1 - bits_f(a2));
                    const unsigned pc[3] = {a1, a2, a3};
                    for (int p = 0; p < 3; ++p) h_pk[((((size_t)b * NT + nt) * 3 + p) * 64 + lane) * 8 + j] = (unsigned short)(pc[p] >> 16);
                }
    std::vector<unsigned short> h_sp((size_t)n * 3 * PIN * CP, 0);
    for (size_t p = 0; p < (size_t)n; ++p)
        for (int c = 0; c < CIN; ++c)
            for (int px = 0; px < PIN; ++px) {
                const float x = h_in[(p * CIN + c) * PIN + px];
                const unsigned a1 = bf16_rne_bits(x);
                const float r1 = x - bits_f(a1);
                const unsigned a2 = bf16_rne_bits(r1), a3 = bf16_rne_bits(r1 - bits_f(a2));
                const unsigned pc[3] = {a1, a2, a3};
                for (int q = 0; q < 3; ++q) h_sp[((p * 3 + q) * PIN + px) * CP + c] = (unsigned short)(pc[q] >> 16);
            }
    uint4* d_sp; CK(hipMalloc(&d_sp, h_sp.size() * 2)); CK(hipMemcpy(d_sp, h_sp.data(), h_sp.size() * 2, hipMemcpyHostToDevice));
    float *d_in, *d_b, *d_out; uint4* d_pk;
    CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pk, h_pk.size() * 2)); CK(hipMemcpy(d_pk, h_pk.data(), h_pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_b, COUT * 4)); CK(hipMemcpy(d_b, h_b.data(), COUT * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)n * COUT * POUT * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&split_conv3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    const int grid = 256;
    hipLaunchKernelGGL(split_conv3, dim3(grid), dim3(NTH), LDS_BYTES, nullptr, d_in, d_sp, d_pk, d_b, d_out, n);
    CK(hipDeviceSynchronize());
    // accuracy on the first 48 patches against a double-precision convolution of the same fp32 numbers
    const int nchk = n < 48 ? n : 48;
    std::vector<float> h_out((size_t)nchk * COUT * POUT);
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int p = 0; p < nchk; ++p)
        for (int o = 0; o < COUT; ++o)
            for (int y = 0; y < HO; ++y)
                for (int x = 0; x < HO; ++x) {
                    double acc = h_b[o];
                    for (int c = 0; c < CIN; ++c)
                        for (int t = 0; t < 9; ++t)
                            acc += (double)h_w[(o * CIN + c) * 9 + t] * (double)h_in[((size_t)p * CIN + c) * PIN + (y + t / 3) * HIN + x + t % 3];
                    const double ref = acc > 0 ? acc : 0;
                    worst = fmax(worst, fabs(ref - (double)h_out[((size_t)p * COUT + o) * POUT + y * HO + x]));
                    scale = fmax(scale, fabs(ref));
                }
    printf("accuracy: max |out - double reference| = %.3g (largest output %.3g) on %d patches\n", worst, scale, nchk);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(split_conv3, dim3(grid), dim3(NTH), LDS_BYTES, nullptr, d_in, d_sp, d_pk, d_b, d_out, n);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const double flop = 2.0 * CIN * 9.0 * COUT * POUT * n;
    printf("split conv3 (bf16x6, resident weights, no staging overlap): %d patch-towers in %.3f ms = %.1f M patch-towers/s = %.1f fp32-equivalent TFLOP/s  (LDS %zu B)\n",
           n, best, n / (best * 1e-3) / 1e6, flop / (best * 1e-3) / 1e12, LDS_BYTES);
    return 0;
}
