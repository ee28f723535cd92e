// f16split_probe - would a two-piece f16 split of the f32 operands on the f16 matrix cores keep f32-grade results, and at what rate?
//
// a = ah + al with ah = f16(a), al = f16(a - ah): 22 significant bits; a*b ~ ah*bh + ah*bl + al*bh (the dropped al*bl is 2^-22 of
// the product), every f16 x f16 product exact in the f32 accumulator.  Three v_mfma_f32_16x16x32_f16 do the work of eight
// v_mfma_f32_16x16x4_f32 (16 against 32 cycles each).
// Part 1 (accuracy): 16 x 16 outputs of K = 192 dot products (ReLU-like activations x signed weights), operands split on the
//   device with the conversion code a kernel would use, at several operand scales (small scales push the low pieces into the f16
//   subnormals: shows whether the matrix cores flush them): error against double of (a) the f32 MFMA chain, (b) three products into
//   one accumulator, (c) the two cross products into their own accumulator, added at the end.
// Part 2 (rate): the A operand re-read from LDS (two 16-byte reads per tile and k block), B resident in registers, TM tiles per wave.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/f16split_probe.hip -o build/f16split_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline void split2(float v, _Float16& h, _Float16& l) {
    h = (_Float16)v;                     // round to nearest even
    l = (_Float16)(v - (float)h);        // the remainder is exact in f32
}

constexpr int K = 192;

// one wave per workgroup: out[blk][3][16][16]
__global__ __launch_bounds__(64) void acc_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out) {
    const int lane = threadIdx.x, lr = lane & 15, kq = lane >> 4;
    const float* a = A + (size_t)blockIdx.x * 16 * K;      // [16][K]
    const float* b = B + (size_t)blockIdx.x * 16 * K;      // [16 cols][K]
    f32x4 d32 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, dm = {0, 0, 0, 0}, dc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) d32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[lr * K + k0 + kq], b[lr * K + k0 + kq], d32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        f16x8 ah, al, bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 h, l;
            split2(a[lr * K + k0 + 8 * kq + j], h, l); ah[j] = h; al[j] = l;
            split2(b[lr * K + k0 + 8 * kq + j], h, l); bh[j] = h; bl[j] = l;
        }
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, d1, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, d1, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d1, 0, 0, 0);
        dc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, dc, 0, 0, 0);
        dc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, dc, 0, 0, 0);
        dm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, dm, 0, 0, 0);
    }
    float* o = out + (size_t)blockIdx.x * 3 * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * kq + i;
        o[0 * 256 + row * 16 + lr] = d32[i];
        o[1 * 256 + row * 16 + lr] = d1[i];
        o[2 * 256 + row * 16 + lr] = dm[i] + dc[i];
    }
}

template <int TM, int NW, int NT>
__global__ __launch_bounds__(64 * NW) void rate_f16(const uint4* __restrict__ src, float* __restrict__ out, int reps) {
    constexpr int KB = 6;
    constexpr int KA = 2;                                   // resident A k blocks, walked in turn
    constexpr int A_V = 2 * NW * TM * KA * 64;
    extern __shared__ uint4 lds[];
    for (int e = threadIdx.x; e < A_V; e += 64 * NW) lds[e] = src[e % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 bh[KB][NT], bl[KB][NT];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            uint4 v = src[(kb * NT + n) * 64 + lane]; bh[kb][n] = *reinterpret_cast<f16x8*>(&v);
            v = src[2048 + (kb * NT + n) * 64 + lane]; bl[kb][n] = *reinterpret_cast<f16x8*>(&v);
        }
    f32x4 acc[TM][NT];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                uint4 v = lds[(((0 * NW + wave) * TM + m) * KA + kb % KA) * 64 + lane]; ah[m] = *reinterpret_cast<f16x8*>(&v);
                v = lds[(((1 * NW + wave) * TM + m) * KA + kb % KA) * 64 + lane]; al[m] = *reinterpret_cast<f16x8*>(&v);
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[kb][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[kb][n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[kb][n], acc[m][n], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * 64 * NW + threadIdx.x] = s;
}

// legacy K = 16 form (4 halves per lane): is it still half the work per instruction at the same cycles?
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__global__ __launch_bounds__(256) void rate_k16(const uint2* __restrict__ src, float* __restrict__ out, int reps) {
    const int lane = threadIdx.x & 63;
    f16x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint2 v = src[i * 64 + lane]; a[i] = *reinterpret_cast<f16x4*>(&v); v = src[256 + i * 64 + lane]; b[i] = *reinterpret_cast<f16x4*>(&v); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[i], b[(i + j) & 3], acc[j], 0, 0, 0);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__global__ __launch_bounds__(256) void rate_k32(const uint4* __restrict__ src, float* __restrict__ out, int reps) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint4 v = src[i * 64 + lane]; a[i] = *reinterpret_cast<f16x8*>(&v); v = src[256 + i * 64 + lane]; b[i] = *reinterpret_cast<f16x8*>(&v); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(i + j) & 3], acc[j], 0, 0, 0);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <class L>
static float time_kernel(L launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 5; ++i) {
        CK(hipEventRecord(e0, nullptr)); launch(); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    return best;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

template <int TM, int NW, int NT>
static void rate(int wgs_per_cu, int reps, const uint4* src, float* out) {
    const int wgs = 256 * wgs_per_cu * 4;
    const size_t lds = (size_t)2 * NW * TM * 2 * 64 * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_f16<TM, NW, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float t = time_kernel([&] { hipLaunchKernelGGL((rate_f16<TM, NW, NT>), dim3(wgs), dim3(64 * NW), lds, nullptr, src, out, reps); });
    const double macs = (double)wgs * NW * TM * NT * 256.0 * 32.0 * 6 * reps;
    printf("rate: TM %d NT %d, %d waves/WG, %d WG/CU (LDS %3zu KB): f16x2 split (3 products) %7.1f f32-equivalent TFLOP/s (%.0f TFLOP/s of f16 MFMA)\n",
           TM, NT, NW, wgs_per_cu, lds >> 10, 2 * macs / (t * 1e-3) / 1e12, 6 * macs / (t * 1e-3) / 1e12);
}

int main() {
    // ---- accuracy
    const int NB = 512;
    std::vector<float> hA((size_t)NB * 16 * K), hB((size_t)NB * 16 * K), hO((size_t)NB * 3 * 256);
    float *dA, *dB, *dO;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dO, hO.size() * 4));
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f; };
    auto gauss = [&]() { float u1 = rnd() + 1e-7f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (int e = -12; e <= 12; e += 4) {
        const float sa = ldexpf(1.f, e), sb = ldexpf(1.f, e);
        for (auto& v : hA) { float g = gauss(); v = g > 0 ? g * sa : 0.f; }
        for (auto& v : hB) v = 0.1f * gauss() * sb;
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(acc_kernel, dim3(NB), dim3(64), 0, nullptr, dA, dB, dO);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost));
        double err[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, scale = 0;
        for (int b = 0; b < NB; ++b)
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)hA[((size_t)b * 16 + r) * K + k] * (double)hB[((size_t)b * 16 + c) * K + k];
                    scale = fmax(scale, fabs(ref));
                    for (int v = 0; v < 3; ++v) {
                        const double d = fabs((double)hO[((size_t)b * 3 + v) * 256 + r * 16 + c] - ref);
                        err[v] += d; mx[v] = fmax(mx[v], d);
                    }
                }
        const double n = (double)NB * 256;
        printf("operand scale 2^%+3d: output scale %.3g | mean (max) |err| / scale: f32 MFMA %.3g (%.3g) | f16x2 one accumulator %.3g (%.3g) | f16x2 cross terms apart %.3g (%.3g)\n",
               e, scale, err[0] / n / scale, mx[0] / scale, err[1] / n / scale, mx[1] / scale, err[2] / n / scale, mx[2] / scale);
    }
    // ---- rate
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = f2h((i & 1 ? -1.f : 1.f) * (0.25f + 3.f * rnd()));
    uint4* src; CK(hipMalloc(&src, 4096 * 16)); CK(hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, (size_t)256 * 16 * 1024 * 4));
    const int reps = 200;
    {
        const int wgs = 256 * 8;
        const float t16 = time_kernel([&] { hipLaunchKernelGGL(rate_k16, dim3(wgs), dim3(256), 0, nullptr, reinterpret_cast<const uint2*>(src), out, 2000); });
        const float t32 = time_kernel([&] { hipLaunchKernelGGL(rate_k32, dim3(wgs), dim3(256), 0, nullptr, src, out, 2000); });
        const double n = (double)wgs * 4 * 16 * 2000;
        printf("bare loops, 2 waves per SIMD: 16x16x16 f16 %.2f ns per MFMA per SIMD (%.0f TFLOP/s) | 16x16x32 f16 %.2f ns (%.0f TFLOP/s)\n",
               t16 * 1e6 / (n / 1024 ), n * 8192 / (t16 * 1e-3) / 1e12, t32 * 1e6 / (n / 1024), n * 16384 / (t32 * 1e-3) / 1e12);
    }
    rate<6, 4, 2>(1, reps, src, out);
    rate<6, 4, 2>(2, reps, src, out);
    rate<4, 4, 2>(2, reps, src, out);
    rate<4, 4, 3>(2, reps, src, out);
    rate<3, 4, 4>(2, reps, src, out);
    rate<2, 4, 2>(4, reps, src, out);
    rate<2, 8, 2>(2, reps, src, out);
    return 0;
}
