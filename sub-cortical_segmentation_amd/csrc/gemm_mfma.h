// Dense layers on the fp32 matrix cores: C[M][N] = act(A[M][K] * W[K][N] + b), v_mfma_f32_16x16x4_f32.
// M = patches of the chunk.  Replaces the DenseLayer(+rectify) nodes of /root/reference/cnn_cort/nets.py:
// <tower>_d1 (:179), FC1 (:217), fc_2 (:227).  W arrives packed [K4][NS] (zero padded rows/columns).
//
// Workgroup = 4 waves, tile 128 rows x 96 columns, K walked in slabs of 32 staged through LDS (register
// prefetch of slab s+1 while slab s feeds the MFMAs); each wave owns 2 M-tiles x 6 N-tiles (12 accumulators).
// LDS images: A rows padded to 34 floats (bank = 2*row + k: the 16 rows x 2 k of a 32-lane read group hit 32
// distinct banks), W rows padded to 112 (112 % 32 == 16).
#pragma once
#include "scs_common.h"

namespace scs {

struct GemmProblem {
    const float* A; int lda;
    const float* W; int NS;      // packed [K4][NS]
    const float* bias;           // [>= N]
    float* C; int ldc;
};
struct GemmArgs {
    GemmProblem p[kTowers];      // blockIdx.z selects
    int M, K4, N;                // K4 = reduction length walked (multiple of 4)
    int K;                       // rows of W that exist (k >= K reads as zero); 0 = K4
    int relu;
};

constexpr int kGemmBM = 128, kGemmBN = 96, kGemmBK = 32, kGemmAS = 34, kGemmBS = 112;
constexpr size_t kGemmLds = (size_t)(2 * kGemmBM * kGemmAS + 2 * kGemmBK * kGemmBS) * 4;

template <int VARIANT = 0>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                                  // [2][128][34]
    float* s_b = smem + 2 * kGemmBM * kGemmAS;          // [2][32][112]
    const int z = blockIdx.z;
    const GemmProblem P = z == 0 ? g.p[0] : (z == 1 ? g.p[1] : g.p[2]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * kGemmBM, n0 = blockIdx.y * kGemmBN;

    const int kmax = g.K ? g.K : g.K4;
    float ra[16], rb[12];
    const int a_col = tid & 31, a_row = tid >> 5;       // A slab: rows a_row + 8*i, column a_col
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + a_row + 8 * i, k = k0 + a_col;
            ra[i] = (row < g.M && k < g.K4) ? P.A[(size_t)row * P.lda + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / kGemmBN, c = e - kr * kGemmBN;
            const int k = k0 + kr, n = n0 + c;
            rb[i] = (k < kmax && n < P.NS) ? P.W[(size_t)k * P.NS + n] : 0.f;
        }
    };
    auto store_slab = [&](int buf) {
        float* da = s_a + buf * kGemmBM * kGemmAS;
        float* db = s_b + buf * kGemmBK * kGemmBS;
#pragma unroll
        for (int i = 0; i < 16; ++i) da[(a_row + 8 * i) * kGemmAS + a_col] = ra[i];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / kGemmBN, c = e - kr * kGemmBN;
            db[kr * kGemmBS + c] = rb[i];
        }
    };

    f32x4 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nslab = (g.K4 + kGemmBK - 1) / kGemmBK;
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) load_slab((s + 1) * kGemmBK);
        const float* sa = s_a + (s & 1) * kGemmBM * kGemmAS + (wave * 32 + lr) * kGemmAS + kq;
        const float* sb = s_b + (s & 1) * kGemmBK * kGemmBS + kq * kGemmBS + lr;
#pragma unroll
        for (int ks = 0; ks < kGemmBK / 4; ++ks) {
            float a[2], b[6];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sa[i * 16 * kGemmAS + ks * 4];
#pragma unroll
            for (int j = 0; j < 6; ++j) b[j] = sb[ks * 4 * kGemmBS + j * 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab) store_slab((s + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int col = n0 + j * 16 + lr;
            if (col >= g.N) continue;
            const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * 32 + i * 16 + kq * 4 + r;
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.relu) v = fmaxf(v, 0.f);
                    P.C[(size_t)row * P.ldc + col] = v;
                }
            }
        }
}

inline void gemm_launch(const GemmArgs& g, int nproblems, hipStream_t st) {
    if (g.M <= 0) return;
    dim3 grid((g.M + kGemmBM - 1) / kGemmBM, (g.N + kGemmBN - 1) / kGemmBN, nproblems);
    hipLaunchKernelGGL(gemm_mfma_kernel<0>, grid, dim3(256), kGemmLds, st, g);
}

}  // namespace scs
