// Micro-benchmark: an in-kernel GRID barrier built like the in-kernel batch-norm finalize of conv_mfma.h (write-through partial
// stores, two-level arrival tickets, relaxed device-scope atomics, no fences) plus a released-generation flag that every
// workgroup polls with write-through-coherent loads - against a dependent launch.  Each stage: a workgroup leaves a 64-float
// partial, the last arriver combines all group records into 64 coefficients, everybody reads them.
//   hipcc -O3 --offload-arch=gfx950 -I sub-cortical_segmentation_amd/csrc tools/gridbar2.hip -o tools/bin/gridbar2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kGroup = 16, kLine = 32;
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ bool take_ticket(unsigned* c, unsigned full, volatile int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == full - 1;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}
// one stage's statistics hand-off + barrier; returns the coefficient of this thread's channel
__device__ __forceinline__ float stage_and_barrier(float* part, float* rec, float* coef, unsigned* cnt, unsigned* flag, unsigned epoch, float mine, volatile int* s_flag, int sleep) {
    const int wg = blockIdx.x, nwg = gridDim.x, grp = wg / kGroup, groups = (nwg + kGroup - 1) / kGroup;
    if (threadIdx.x < 64) st_sc1(part + (size_t)wg * 64 + threadIdx.x, mine);
    if (take_ticket(cnt + (size_t)kLine * (1 + grp), min(kGroup, nwg - grp * kGroup), s_flag)) {
        if (threadIdx.x < 64) {
            float s = 0.f;
            for (int i = 0; i < kGroup && grp * kGroup + i < nwg; ++i) s += ld_sc1(part + (size_t)(grp * kGroup + i) * 64 + threadIdx.x);
            st_sc1(rec + (size_t)grp * 64 + threadIdx.x, s);
        }
        if (take_ticket(cnt, groups, s_flag)) {
            if (threadIdx.x < 64) {
                float s = 0.f;
                for (int g = 0; g < groups; ++g) s += ld_sc1(rec + (size_t)g * 64 + threadIdx.x);
                st_sc1(coef + threadIdx.x, s);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // everybody: wait for the generation, then read the coefficients
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && spins < (1u << 22)) { __builtin_amdgcn_s_sleep(1); if (sleep >= 8) __builtin_amdgcn_s_sleep(8); if (sleep >= 32) { __builtin_amdgcn_s_sleep(24); } ++spins; }
    }
    __syncthreads();
    return ld_sc1(coef + (threadIdx.x & 63));
}
__global__ __launch_bounds__(256) void fused_kernel(float* part, float* rec, float* coef, unsigned* cnt, unsigned* flag, unsigned epoch0, int S, float* sink, int sleep) {
    extern __shared__ float pad[];
    __shared__ int s_flag;
    float v = (float)(blockIdx.x + threadIdx.x);
    for (int s = 0; s < S; ++s) v = stage_and_barrier(part, rec, coef + 64 * (s & 1), cnt, flag, epoch0 + s + 1, v * 1e-3f, &s_flag, sleep);
    if (v == 12345.f) sink[0] = v + pad[0];
}
__global__ __launch_bounds__(256) void stage_kernel(float* part, float* rec, float* coef, unsigned* cnt, float* sink) {
    extern __shared__ float pad[];
    __shared__ int s_flag;
    const int wg = blockIdx.x, nwg = gridDim.x, grp = wg / kGroup, groups = (nwg + kGroup - 1) / kGroup;
    float mine = coef[threadIdx.x & 63] * 1e-3f + (float)wg;
    if (threadIdx.x < 64) st_sc1(part + (size_t)wg * 64 + threadIdx.x, mine);
    if (take_ticket(cnt + (size_t)kLine * (1 + grp), min(kGroup, nwg - grp * kGroup), &s_flag)) {
        if (threadIdx.x < 64) {
            float s = 0.f;
            for (int i = 0; i < kGroup && grp * kGroup + i < nwg; ++i) s += ld_sc1(part + (size_t)(grp * kGroup + i) * 64 + threadIdx.x);
            st_sc1(rec + (size_t)grp * 64 + threadIdx.x, s);
        }
        if (take_ticket(cnt, groups, &s_flag)) {
            if (threadIdx.x < 64) {
                float s = 0.f;
                for (int g = 0; g < groups; ++g) s += ld_sc1(rec + (size_t)g * 64 + threadIdx.x);
                coef[threadIdx.x] = s;
            }
        }
    }
    if (mine == 12345.f) sink[0] = pad[0];
}
int main(int argc, char** argv) {
    const int S = 9, iters = 200;
    const int lds = 50 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int nwg : {256, 384, 768}) {
        for (int sleep : {1, 8, 32}) {
        float *part, *rec, *coef, *sink; unsigned *cnt, *flag;
        CK(hipMalloc(&part, (size_t)nwg * 64 * 4)); CK(hipMalloc(&rec, 64 * 64 * 4)); CK(hipMalloc(&coef, 128 * 4)); CK(hipMalloc(&sink, 64));
        CK(hipMalloc(&cnt, 128 * kLine * 4)); CK(hipMalloc(&flag, 256));
        CK(hipMemset(cnt, 0, 128 * kLine * 4)); CK(hipMemset(flag, 0, 256)); CK(hipMemset(coef, 0, 128 * 4));
        hipStream_t st; CK(hipStreamCreate(&st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms_chain = 0, ms_fused = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it)
                for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage_kernel, dim3(nwg), dim3(256), lds, st, part, rec, coef, cnt, sink);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_chain, e0, e1));
        }
        unsigned epoch = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) { hipLaunchKernelGGL(fused_kernel, dim3(nwg), dim3(256), lds, st, part, rec, coef, cnt, flag, epoch, S, sink, sleep); epoch += S; }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms_fused, e0, e1));
        }
        printf("workgroups %4d sleep %2d: %d dependent launches %.2f us (%.2f per stage); one launch, %d ticket-tree grid barriers %.2f us (%.2f per stage)\n",
               nwg, sleep, S, ms_chain * 1e3 / iters, ms_chain * 1e3 / iters / S, S, ms_fused * 1e3 / iters, ms_fused * 1e3 / iters / S);
        CK(hipFree(part)); CK(hipFree(rec)); CK(hipFree(coef)); CK(hipFree(cnt)); CK(hipFree(flag)); CK(hipFree(sink));
        }
    }
    return 0;
}
