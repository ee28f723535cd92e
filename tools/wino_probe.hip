// Feasibility probe for a Winograd F(2x2,3x3) main loop on gfx950 f32 MFMA: per (M-tile of 16 tiles, k-step of 4
// channels): 8 ds_read_b64 (the 4x4 input patch of this lane's tile/channel), 32 vector adds (B^T d B), then for
// each of the 16 transform positions 2 weight reads (16-wide + 4-wide) and the MFMA pair.  Reports the fraction of
// the MFMA pipe rate the loop sustains at 1..3 waves per SIMD (operands are random, results are not meaningful).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void wino(float* out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) lds[i] = (i % 97) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* dpf = lds + (lane >> 4) * 912 + 2 * (lane & 15);       // tile i at x = 2i, channel kq on its own plane
    const float* wp = lds + 4096 + (lane & 15) + (lane >> 4) * 24;
    f32x4 acc[16], acc4[16];
    for (int i = 0; i < 16; ++i) { acc[i] = f32x4{0, 0, 0, 0}; acc4[i] = f32x4{0, 0, 0, 0}; }
    for (int it = 0; it < iters; ++it) {
        const int koff = (it & 7) * 512;
        float d[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float2 v = *reinterpret_cast<const float2*>(dpf + (koff & 2047) + (r >> 1) * 30 + (r & 1) * 2); d[2 * r] = v.x; d[2 * r + 1] = v.y; }
        // B^T d B for F(2x2,3x3): rows then columns (32 adds)
        float t[16], v[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
            t[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
            t[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
            t[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r * 4 + 0] = t[r * 4 + 0] - t[r * 4 + 2];
            v[r * 4 + 1] = t[r * 4 + 1] + t[r * 4 + 2];
            v[r * 4 + 2] = t[r * 4 + 2] - t[r * 4 + 1];
            v[r * 4 + 3] = t[r * 4 + 1] - t[r * 4 + 3];
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const float b = wp[xi * 96 + (koff & 1023)], b4 = wp[xi * 96 + 16 + (koff & 1023)];
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi], b, acc[xi], 0, 0, 0);
            acc4[xi] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4, v[xi], acc4[xi], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3] + acc4[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, (1 << 20) * 4);
    for (int w = 1; w <= 2; ++w) {
        const int iters = 1000, threads = 256 * w;
        auto launch = [&]() {
            if (w == 1) hipLaunchKernelGGL(wino<256>, dim3(256), dim3(256), 49152, 0, d, iters);
            else hipLaunchKernelGGL(wino<512>, dim3(256), dim3(512), 49152, 0, d, iters);
        };
        launch();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / iters, pipe = 16 * 40.0 * w;
        printf("winograd loop body, %d wave(s)/SIMD: %.0f cycles (2.4 GHz nominal) per body per SIMD; MFMA pipe time %.0f -> %.0f %% of pipe rate\n",
               w, cyc, pipe, 100.0 * pipe / cyc);
    }
    return 0;
}
