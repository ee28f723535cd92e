// mfma_valu - issue cost of the f32 MFMA forms the conv kernels use (16x16x4 and the 16-block 4x4x1), alone, interleaved, and with
// 1..8 independent v_add_f32 behind every 16x16x4: wall time and s_memtime ticks per instruction, one and four waves per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_valu.hip -o build/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = f32x4{0, 0, 0, 0}; b[i] = f32x4{0, 0, 0, 0}; }
    const float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = x * i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[i], 0, 0, 0);
            if (MODE == 1 || MODE == 2) b[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, b[i], 0, 0, 0);
            if (MODE >= 3) {                                   // a 16x16x4 MFMA followed by MODE - 2 independent vector adds
                a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < MODE - 2; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(i * 2 + q) & 15]) : "v"(y));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3] + b[i][0] + b[i][1] + b[i][2] + b[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 9; ++mode) {
        if (mode == 7) continue;
        for (int wgs : {256, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int r = 0; r < 2; ++r) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                if (mode == 8) hipLaunchKernelGGL(k<10>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
            const double n = (double)iters * 8 * (mode == 2 ? 2 : 1);   // (modes >= 3: per MFMA)
            printf("mode %d (%s) %d workgroups: %.3f ms; %.1f ns per instruction per wave (wall), memtime ticks per instruction %.2f\n", mode,
                   mode == 0 ? "16x16x4" : mode == 1 ? "4x4x1" : mode == 2 ? "both interleaved" : mode == 3 ? "16x16x4 + 1 v_add" : mode == 4 ? "+ 2 v_add" : mode == 5 ? "+ 3 v_add" : mode == 6 ? "+ 4 v_add" : "+ 8 v_add", wgs, ms, ms * 1e6 / n, (double)h[0] / n);
        }
    }
    return 0;
}
