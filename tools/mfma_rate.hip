// Issue-rate probe for f32 MFMA mixes on gfx950: cycles per instruction for pure 16x16x4, pure 4x4x1 and the
// 1:1 interleave the conv kernels use, at 1..4 waves per SIMD, operands in registers (no LDS / memory).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void rate(float* out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    const float* lp = lds + (threadIdx.x & 63);
    f32x4 a16[8], a4[8];
    for (int i = 0; i < 8; ++i) { a16[i] = f32x4{0, 0, 0, 0}; a4[i] = f32x4{0, 0, 0, 0}; }
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) a16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[i], 0, 0, 0);
            if (MODE == 1 || MODE == 2) a4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a4[i], 0, 0, 0);
            if (MODE == 3) { a16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[i], 0, 0, 0); x = x * 1.0001f + y; }   // + 1 VALU per MFMA
            if (MODE == 4) { const float v = lp[((it * 8 + i) * 64) & 4031]; a16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, y, a16[i], 0, 0, 0); }   // A operand from LDS
            if (MODE == 5) { const float v = lp[((it * 8 + i) * 64) & 4031]; a16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, y, a16[i], 0, 0, 0);
                             a4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(y, v, a4[i], 0, 0, 0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a16[i][0] + a4[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + x;
    if (threadIdx.x == 0) out[(1 << 20) + blockIdx.x] = (float)(t1 - t0);
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, ((1 << 20) + 4096) * 4);
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;      // one block per CU, waves_per_simd waves on each SIMD
    hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static float h[4096];
    hipMemcpy(h, d + (1 << 20), 256 * 4, hipMemcpyDeviceToHost);
    double cyc0 = 0; for (int i = 0; i < 256; ++i) cyc0 += h[i]; cyc0 /= 256;
    const double cyc = ms * 1e-3 * 2.4e9;      // whole-kernel wall time in nominal 2.4 GHz cycles (all waves of the block)
    const int per_iter = (MODE == 2 || MODE == 5) ? 16 : 8;
    const double pipe = (MODE == 0 || MODE == 3 || MODE == 4) ? 8 * 32.0 : (MODE == 1 ? 8 * 8.0 : 8 * 40.0);
    printf("[wave0 stream %.0f] %-34s waves/SIMD %d: %.1f cycles per loop body (%d MFMAs) per wave; pipe-bound would be %.0f x %d waves = %.0f -> %.0f %% of pipe rate\n",
           cyc0 / iters, name, waves_per_simd, cyc / iters, per_iter, pipe, waves_per_simd, pipe * waves_per_simd, 100.0 * pipe * waves_per_simd / (cyc / iters));
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0>("16x16x4 only", w);
        run<1>("4x4x1 only", w);
        run<2>("16x16x4 + 4x4x1 interleaved", w);
        run<3>("16x16x4 + one v_fma each", w);
        run<4>("16x16x4, A operand via ds_read", w);
        run<5>("16x16x4 + 4x4x1, A via ds_read", w);
    }
    return 0;
}
