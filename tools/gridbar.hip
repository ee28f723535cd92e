// Micro-benchmark: what a dependent launch costs against an in-kernel grid barrier on this GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/gridbar.hip -o gpurun_out/gridbar && gpurun_out/gridbar
// (a) a chain of S dependent kernels, each workgroup reading 32 KB and writing 4 KB;
// (b) the same S stages inside one kernel, separated by a grid barrier (monotonic counter, one atomic per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void stage(const float* __restrict__ in, float* __restrict__ out, int s) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    const float4* p = reinterpret_cast<const float4*>(in) + (size_t)((wg * 37 + s * 11) % nwg) * 2048;   // another workgroup's output
    float4 acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) { float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4* q = reinterpret_cast<float4*>(out) + (size_t)wg * 2048;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) q[i] = acc;
}
__global__ __launch_bounds__(256) void stage_kernel(const float* in, float* out, int s) { stage(in, out, s); }

#ifndef VARIANT
#define VARIANT 0
#endif
// ctr[0]: arrivals (monotonic); ctr[64]: released generation, written by the last arriver (another cache line)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
#if VARIANT == 0
        __threadfence();
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
#elif VARIANT == 1
        const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == target) __hip_atomic_store(ctr + 64, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else while (__hip_atomic_load(ctr + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
#else
        const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == target) __hip_atomic_store(ctr + 64, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else while (__hip_atomic_load(ctr + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(32);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void fused_kernel(float* a, float* b, int S, unsigned* ctr, unsigned base) {
    for (int s = 0; s < S; ++s) {
        stage((s & 1) ? b : a, (s & 1) ? a : b, s);
        if (s + 1 < S) grid_barrier(ctr, base + (unsigned)(s + 1) * gridDim.x);
    }
}

int main() {
    const int S = 9, iters = 200;
    for (int nwg : {64, 128, 256, 512}) {
        float *a, *b; unsigned* ctr;
        CK(hipMalloc(&a, (size_t)nwg * 2048 * 16)); CK(hipMalloc(&b, (size_t)nwg * 2048 * 16)); CK(hipMalloc(&ctr, 1024));
        CK(hipMemset(a, 0, (size_t)nwg * 2048 * 16)); CK(hipMemset(ctr, 0, 1024));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms_chain = 0, ms_fused = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it)
                for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage_kernel, dim3(nwg), dim3(256), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, s);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_chain, e0, e1));
        }
        unsigned base = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) {
                hipLaunchKernelGGL(fused_kernel, dim3(nwg), dim3(256), 0, st, a, b, S, ctr, base);
                base += (unsigned)(S - 1) * nwg;
            }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_fused, e0, e1));
        }
        printf("workgroups %4d: chain of %d launches %.2f us (%.2f us per stage); one launch with %d grid barriers %.2f us (%.2f us per stage)\n",
               nwg, S, ms_chain * 1e3 / iters, ms_chain * 1e3 / iters / S, S - 1, ms_fused * 1e3 / iters, ms_fused * 1e3 / iters / S);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctr));
    }
    return 0;
}
