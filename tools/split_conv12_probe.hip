// split_conv12_probe - research prototype for the next round (NOT part of libscs): the dominant kernel of the forward
// pass, conv1 + ReLU -> conv2 -> 2x2 max-pool -> bias + ReLU of one tower, with conv2 on the error-free bf16x6 split.
//
//   * persistent 4-wave workgroups, two per CU; a unit of work = (patch, band of 4 conv2 rows = 2 pooled rows);
//   * conv2's split weights (7 k-blocks x 2 column tiles x 3 pieces = 42 KB, MFMA lane order) stay resident in LDS;
//   * per unit: 8 raw rows -> conv1 on the fp32 MFMA (K = 9 taps padded to 12, 20 channels as two 16-wide column
//     tiles) -> bias + ReLU -> three bf16 planes [pixel][24 channels] in LDS (26 KB) -> conv2 as 7 tiles x 7 blocks x
//     2 column tiles x 6 v_mfma_f32_16x16x32_bf16 -> in-lane max-pool -> bias + ReLU -> global.
// Prints the error against a double-precision CPU evaluation and the rate next to libscs' fused fp32 kernel
// (1.467 ms for 8192 patches x 3 towers = 16.75 M patch-towers/s, profiles/r01_kbench_final.txt).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/split_conv12_probe.hip -o build/split_conv12_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int C1 = 20, CP = 24, C2 = 20, NB = 7, HOB = 4, H1B = HOB + 2, W1 = 30, W2 = 28;     // band: 4 conv2 rows, 6 conv1 rows
constexpr int NGRP = 9 * (CP / 8), NBLK = (NGRP + 3) / 4, NT = 2;
constexpr int W_V = NBLK * NT * 3 * 64;                              // resident conv2 weights, uint4 units (42 KB)
constexpr int PIECE_BYTES = H1B * W1 * CP * 2;                       // one bf16 plane of the band's conv1 output (8640 B)
constexpr int T1 = (H1B * W1 + 15) / 16;                             // conv1 tiles per band: 180 px -> 12 (11.25)
constexpr int T2 = HOB * W2 / 16;                                    // conv2 tiles per band: 28 windows x 4 positions = 7
constexpr int NTH = 256, TPW2 = (T2 + 3) / 4, TPW1 = T1 / 4;
constexpr int RAW_FLOATS = (H1B + 2) * 32 + 8, W1_FLOATS = 12 * 32 + 32;   // raw rows; conv1 weights [12][32] + bias[32]
constexpr size_t LDS_BYTES = (size_t)W_V * 16 + 3 * PIECE_BYTES + (RAW_FLOATS + W1_FLOATS) * 4;
static_assert(T1 % 4 == 0, "conv1 tiles split evenly over the waves");

__host__ __device__ inline unsigned bf16_rne_bits(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return u & 0xFFFF0000u; }
__host__ __device__ inline float bits_f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

__global__ __launch_bounds__(NTH, 2) void split_conv12(const float* __restrict__ in, const float* __restrict__ w1pk, const uint4* __restrict__ w2pk,
                                                       const float* __restrict__ b2, float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* s_w = reinterpret_cast<uint4*>(smem);
    unsigned char* s_a = smem + (size_t)W_V * 16;
    float* s_x = reinterpret_cast<float*>(s_a + 3 * PIECE_BYTES);
    float* s_w1 = s_x + RAW_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, kq = lane >> 4;
    for (int e = tid; e < W_V; e += NTH) s_w[e] = w2pk[e];
    for (int e = tid; e < W1_FLOATS; e += NTH) s_w1[e] = w1pk[e];
    for (int e = tid; e < 3 * PIECE_BYTES / 4; e += NTH) reinterpret_cast<unsigned*>(s_a)[e] = 0u;   // channels 20..23 stay zero
    for (int e = tid; e < RAW_FLOATS; e += NTH) s_x[e] = 0.f;
    __syncthreads();
    // conv1 operands of this lane: tap offsets in the raw rows for k = 4*s3 + kq, weights / bias for couts lr and 16 + lr
    int toff[3];
    float w1a[3], w1b[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int k = 4 * s3 + kq, tk = k < 9 ? k : 8;
        toff[s3] = (tk / 3) * 32 + tk % 3;
        w1a[s3] = s_w1[k * 32 + lr];
        w1b[s3] = s_w1[k * 32 + 16 + lr];
    }
    const float b1a = s_w1[12 * 32 + lr], b1b = s_w1[12 * 32 + 16 + lr];
    // conv2: per-lane byte offsets of the k-groups (group = 4*block + kq -> tap, 8-channel group) and of the wave's tiles
    int goff[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int gi = 4 * b + kq, tap = gi / 3, cg = gi - tap * 3;
        goff[b] = gi < NGRP ? ((tap / 3) * W1 + tap % 3) * CP * 2 + cg * 16 : 0;
    }
    int tbase[TPW2];
#pragma unroll
    for (int j = 0; j < TPW2; ++j) {
        const int t = min(wave + 4 * j, T2 - 1);                      // waves without a last tile recompute tile T2-1, never store it
        const int widx = t * 4 + (lr >> 2), q = lr & 3, wy = widx / 14, wx = widx - wy * 14;
        tbase[j] = ((2 * wy + (q >> 1)) * W1 + 2 * wx + (q & 1)) * CP * 2;
    }
    const float b2a = b2[lr], b2b = lr < 4 ? b2[16 + lr] : 0.f;
    const int units = n * NB;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int p = u / NB, band = u - p * NB;
        __syncthreads();                                             // previous unit consumed
        if (tid < (H1B + 2) * 8) reinterpret_cast<float4*>(s_x)[tid] = reinterpret_cast<const float4*>(in + (size_t)p * 1024 + band * HOB * 32)[tid];
        __syncthreads();
#ifndef PROBE_ABLATE
#define PROBE_ABLATE 0            // timing only - 1: conv1 stage only for a workgroup's first unit, 2: no conv2 MFMAs, 3: no stores
#endif
        // ---- conv1 (fp32 MFMA): 12 tiles of 16 pixels, 3 per wave; D = [pixel 4kq + r][cout lr] -> split -> [pixel][24] planes
        if (PROBE_ABLATE != 1 || u == (int)blockIdx.x)
#pragma unroll
        for (int jj = 0; jj < TPW1; ++jj) {
            const int t = wave + 4 * jj;
            const int pp = min(t * 16 + lr, H1B * W1 - 1), y = pp / W1, x = pp - y * W1;
            f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const float av = s_x[y * 32 + x + toff[s3]];
                da = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1a[s3], da, 0, 0, 0);
                db = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1b[s3], db, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int px = t * 16 + 4 * kq + r;
                if (px < H1B * W1) {
                    unsigned short* d = reinterpret_cast<unsigned short*>(s_a) + px * CP;
                    const float va = fmaxf(da[r] + b1a, 0.f);
                    unsigned a1 = bf16_rne_bits(va); float r1 = va - bits_f(a1); unsigned a2 = bf16_rne_bits(r1), a3 = bf16_rne_bits(r1 - bits_f(a2));
                    d[lr] = (unsigned short)(a1 >> 16); d[PIECE_BYTES / 2 + lr] = (unsigned short)(a2 >> 16); d[PIECE_BYTES + lr] = (unsigned short)(a3 >> 16);
                    if (lr < 4) {
                        const float vb = fmaxf(db[r] + b1b, 0.f);
                        a1 = bf16_rne_bits(vb); r1 = vb - bits_f(a1); a2 = bf16_rne_bits(r1); a3 = bf16_rne_bits(r1 - bits_f(a2));
                        d[16 + lr] = (unsigned short)(a1 >> 16); d[PIECE_BYTES / 2 + 16 + lr] = (unsigned short)(a2 >> 16); d[PIECE_BYTES + 16 + lr] = (unsigned short)(a3 >> 16);
                    }
                }
            }
        }
        __syncthreads();
        // ---- conv2 on the bf16 matrix cores
        f32x4 acc[TPW2][NT];
#pragma unroll
        for (int j = 0; j < TPW2; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < (PROBE_ABLATE == 2 ? 1 : NBLK); ++b) {
            bf16x8 a[3][TPW2], w[3][NT];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                for (int j = 0; j < TPW2; ++j) { const uint4 v = *reinterpret_cast<const uint4*>(s_a + pc * PIECE_BYTES + tbase[j] + goff[b]); a[pc][j] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { const uint4 v = s_w[((b * NT + nt) * 3 + pc) * 64 + lane]; w[pc][nt] = *reinterpret_cast<const bf16x8*>(&v); }
            }
#pragma unroll
            for (int j = 0; j < TPW2; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 c = acc[j][nt];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][j], w[0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][j], w[1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[2][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][j], w[0][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[1][nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][j], w[0][nt], c, 0, 0, 0);
                    acc[j][nt] = c;
                }
        }
        // ---- epilogue: rows 4kq..4kq+3 of a tile are one pooling window -> in-lane max, bias, ReLU -> [20][14][14]
        float* op = out + (size_t)p * C2 * 196 + band * 2 * 14;
#pragma unroll
        for (int j = 0; j < TPW2; ++j) {
            const int t = wave + 4 * j;
            if (t >= T2) continue;
            if (PROBE_ABLATE == 3 && acc[j][0][0] != 123.25f) continue;
            const int widx = t * 4 + kq;                              // window of this lane group: D row 4kq + r <-> positions r
            const f32x4 va = acc[j][0], vb = acc[j][1];
            op[lr * 196 + widx] = fmaxf(fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3])) + b2a, 0.f);
            if (lr < 4) op[(16 + lr) * 196 + widx] = fmaxf(fmaxf(fmaxf(vb[0], vb[1]), fmaxf(vb[2], vb[3])) + b2b, 0.f);
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192 * 3;
    std::vector<float> h_in((size_t)n * 1024), h_w1(C1 * 9), h_b1(C1), h_w2(C2 * C1 * 9), h_b2(C2);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f; };
    for (auto& v : h_in) v = 4.0f * rnd() - 2.0f;
    for (auto& v : h_w1) v = 0.6f * (rnd() - 0.5f);
    for (auto& v : h_b1) v = 0.4f * (rnd() - 0.3f);
    for (auto& v : h_w2) v = 0.2f * (rnd() - 0.5f);
    for (auto& v : h_b2) v = 0.2f * (rnd() - 0.5f);
    std::vector<float> h_w1pk(W1_FLOATS, 0.f);                       // [12][32] (k = tap, couts 0..19) + bias[32]
    for (int k = 0; k < 9; ++k) for (int c = 0; c < C1; ++c) h_w1pk[k * 32 + c] = h_w1[c * 9 + k];
    for (int c = 0; c < C1; ++c) h_w1pk[12 * 32 + c] = h_b1[c];
    std::vector<unsigned short> h_pk((size_t)W_V * 8, 0);
    for (int b = 0; b < NBLK; ++b)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int lr = lane & 15, kq = lane >> 4, gi = 4 * b + kq, tap = gi / 3, cg = gi % 3, c = cg * 8 + j, o = nt * 16 + lr;
                    const float wv = (gi < NGRP && c < C1 && o < C2) ? h_w2[(o * C1 + c) * 9 + tap] : 0.f;
                    const unsigned a1 = bf16_rne_bits(wv);
                    const float r1 = wv - bits_f(a1);
                    const unsigned a2 = bf16_rne_bits(r1), a3 = bf16_rne_bits(r1 - bits_f(a2));
                    const unsigned pc[3] = {a1, a2, a3};
                    for (int q = 0; q < 3; ++q) h_pk[((((size_t)b * NT + nt) * 3 + q) * 64 + lane) * 8 + j] = (unsigned short)(pc[q] >> 16);
                }
    float *d_in, *d_w1, *d_b2, *d_out; uint4* d_pk;
    CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_w1, h_w1pk.size() * 4)); CK(hipMemcpy(d_w1, h_w1pk.data(), h_w1pk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pk, h_pk.size() * 2)); CK(hipMemcpy(d_pk, h_pk.data(), h_pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_b2, C2 * 4)); CK(hipMemcpy(d_b2, h_b2.data(), C2 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)n * C2 * 196 * 4)); CK(hipMemset(d_out, 0, (size_t)n * C2 * 196 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&split_conv12), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    const int grid = 512;
    hipLaunchKernelGGL(split_conv12, dim3(grid), dim3(NTH), LDS_BYTES, nullptr, d_in, d_w1, d_pk, d_b2, d_out, n);
    CK(hipDeviceSynchronize());
    const int nchk = n < 6 ? n : 6;
    std::vector<float> h_out((size_t)nchk * C2 * 196);
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    std::vector<double> a1((size_t)C1 * 900);
    for (int p = 0; p < nchk; ++p) {
        for (int c = 0; c < C1; ++c)
            for (int y = 0; y < 30; ++y)
                for (int x = 0; x < 30; ++x) {
                    double acc = h_b1[c];
                    for (int t = 0; t < 9; ++t) acc += (double)h_w1[c * 9 + t] * (double)h_in[(size_t)p * 1024 + (y + t / 3) * 32 + x + t % 3];
                    a1[(c * 30 + y) * 30 + x] = (double)(float)(acc > 0 ? acc : 0);       // the layer boundary is fp32
                }
        for (int o = 0; o < C2; ++o)
            for (int wy = 0; wy < 14; ++wy)
                for (int wx = 0; wx < 14; ++wx) {
                    double m = -1e300;
                    for (int q = 0; q < 4; ++q) {
                        const int y = 2 * wy + (q >> 1), x = 2 * wx + (q & 1);
                        double acc = 0;
                        for (int c = 0; c < C1; ++c)
                            for (int t = 0; t < 9; ++t) acc += (double)h_w2[(o * C1 + c) * 9 + t] * a1[(c * 30 + y + t / 3) * 30 + x + t % 3];
                        m = acc > m ? acc : m;
                    }
                    const double ref = m + h_b2[o] > 0 ? m + h_b2[o] : 0;
                    worst = fmax(worst, fabs(ref - (double)h_out[((size_t)p * C2 + o) * 196 + wy * 14 + wx]));
                    scale = fmax(scale, fabs(ref));
                }
    }
    printf("accuracy: max |out - double reference| = %.3g (largest output %.3g) on %d patches\n", worst, scale, nchk);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(split_conv12, dim3(grid), dim3(NTH), LDS_BYTES, nullptr, d_in, d_w1, d_pk, d_b2, d_out, n);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const double flop = 2.0 * (162000.0 + 2822400.0) * n;
    printf("split conv1+conv2+pool (conv2 on bf16x6): %d patch-towers in %.3f ms = %.2f M patch-towers/s = %.1f fp32-equivalent TFLOP/s (LDS %zu B, 2 WG/CU)\n",
           n, best, n / (best * 1e-3) / 1e6, flop / (best * 1e-3) / 1e12, LDS_BYTES);
    return 0;
}
