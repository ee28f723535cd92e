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

constexpr int kGemmBK = 32, kGemmAS = 34;

// MT = 16-row tiles per wave, NT = 16-column tiles per workgroup: workgroup tile (64*MT) x (16*NT)
template <int MT, int NT>
struct GemmTile {
    static constexpr int BM = 64 * MT, BN = 16 * NT, BS = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int A_PER = BM * kGemmBK / 256, B_PER = (kGemmBK * BN + 255) / 256;
    static constexpr size_t LDS = (size_t)(2 * BM * kGemmAS + 2 * kGemmBK * BS) * 4;
};

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmArgs g) {
    using T = GemmTile<MT, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                                  // [2][BM][34]
    float* s_b = smem + 2 * T::BM * kGemmAS;            // [2][32][BS]
    const int z = blockIdx.z;
    const GemmProblem P = z == 0 ? g.p[0] : (z == 1 ? g.p[1] : g.p[2]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * T::BM, n0 = blockIdx.y * T::BN;

    const int kmax = g.K ? g.K : g.K4;
    float ra[T::A_PER], rb[T::B_PER];
    const int a_col = tid & 31, a_row = tid >> 5;       // A slab: rows a_row + 8*i, column a_col
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < T::A_PER; ++i) {
            const int row = m0 + a_row + 8 * i, k = k0 + a_col;
            ra[i] = (row < g.M && k < g.K4) ? P.A[(size_t)row * P.lda + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < T::B_PER; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / T::BN, c = e - kr * T::BN;
            const int k = k0 + kr, n = n0 + c;
            rb[i] = (kr < kGemmBK && k < kmax && n < P.NS) ? P.W[(size_t)k * P.NS + n] : 0.f;
        }
    };
    auto store_slab = [&](int buf) {
        float* da = s_a + buf * T::BM * kGemmAS;
        float* db = s_b + buf * kGemmBK * T::BS;
#pragma unroll
        for (int i = 0; i < T::A_PER; ++i) da[(a_row + 8 * i) * kGemmAS + a_col] = ra[i];
#pragma unroll
        for (int i = 0; i < T::B_PER; ++i) {
            const int e = tid + 256 * i;
            const int kr = e / T::BN, c = e - kr * T::BN;
            if (kr < kGemmBK) db[kr * T::BS + c] = rb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nslab = (g.K4 + kGemmBK - 1) / kGemmBK;
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) load_slab((s + 1) * kGemmBK);
        const float* sa = s_a + (s & 1) * T::BM * kGemmAS + (wave * 16 * MT + lr) * kGemmAS + kq;
        const float* sb = s_b + (s & 1) * kGemmBK * T::BS + kq * T::BS + lr;
#pragma unroll
        for (int ks = 0; ks < kGemmBK / 4; ++ks) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = sa[i * 16 * kGemmAS + ks * 4];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = sb[ks * 4 * T::BS + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab) store_slab((s + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + j * 16 + lr;
            if (col >= g.N) continue;
            const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * 16 * MT + i * 16 + kq * 4 + r;
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.relu) v = fmaxf(v, 0.f);
                    P.C[(size_t)row * P.ldc + col] = v;
                }
            }
        }
}

// big tile (128 x 96) when it fills the chip, small tile (64 x 48) for the short-M / short-N training GEMMs
inline void gemm_launch(const GemmArgs& g, int nproblems, hipStream_t st) {
    if (g.M <= 0) return;
    using Big = GemmTile<2, 6>;
    using Small = GemmTile<1, 3>;
    const long big_wgs = (long)((g.M + Big::BM - 1) / Big::BM) * ((g.N + Big::BN - 1) / Big::BN) * nproblems;
    if (big_wgs >= 192) {
        dim3 grid((g.M + Big::BM - 1) / Big::BM, (g.N + Big::BN - 1) / Big::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<2, 6>), grid, dim3(256), Big::LDS, st, g);
    } else {
        dim3 grid((g.M + Small::BM - 1) / Small::BM, (g.N + Small::BN - 1) / Small::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<1, 3>), grid, dim3(256), Small::LDS, st, g);
    }
}

}  // namespace scs
