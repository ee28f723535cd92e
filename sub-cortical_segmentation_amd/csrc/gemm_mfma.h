// Dense layers on the fp32 matrix cores: C[M][N] = act(A[M][K] * W[K][N] + b), v_mfma_f32_16x16x4_f32.
// M = patches of the chunk.  Replaces the DenseLayer(+rectify) nodes of /root/reference/cnn_cort/nets.py:
// <tower>_d1 (:179), FC1 (:217), fc_2 (:227).  W arrives packed [K4][NS] (zero padded rows/columns).
//
// Workgroup = 4 waves, tile 128 rows x 96 columns, K walked in slabs of BK staged through LDS (register
// prefetch of slab s+1 while slab s feeds the MFMAs); each wave owns 2 M-tiles x 6 N-tiles (12 accumulators).
// LDS images: A rows padded to 34 floats (bank = 2*row + k: the 16 rows x 2 k of a 32-lane read group hit 32
// distinct banks), W rows padded to 112 (112 % 32 == 16).
#pragma once
#include "scs_common.h"
#include <cstdint>

namespace scs {

struct GemmProblem {
    const float* A; int lda;
    const float* W; int NS;      // packed [K4][NS]
    const float* bias;           // [>= N]
    float* C; int ldc;
    // optional epilogue of the training step (all null / 0 at inference), applied after bias (+ ReLU):
    //   C2 set   (forward through a dropout layer): C = v, C2 = mask ? 2v : 0
    //   C2 null  (backward): v = mask ? 2v : 0 (dropout backward), then v = gate > 0 ? v : 0 (ReLU backward), C = v
    const uint8_t* mask; int ldm;
    const float* gate; int ldg;
    float* C2; int ldc2;
};

__device__ __forceinline__ void gemm_store(const GemmProblem& P, int relu, int row, int col, float v) {
    if (relu) v = fmaxf(v, 0.f);
    if (P.C2) {
        P.C[(size_t)row * P.ldc + col] = v;
        P.C2[(size_t)row * P.ldc2 + col] = P.mask[(size_t)row * P.ldm + col] ? 2.f * v : 0.f;
        return;
    }
    if (P.mask) v = P.mask[(size_t)row * P.ldm + col] ? 2.f * v : 0.f;
    if (P.gate) v = P.gate[(size_t)row * P.ldg + col] > 0.f ? v : 0.f;
    P.C[(size_t)row * P.ldc + col] = v;
}
struct GemmArgs {
    GemmProblem p[kTowers];      // blockIdx.z selects
    int M, K4, N;                // K4 = reduction length walked (multiple of 4)
    int K;                       // rows of W that exist (k >= K reads as zero); 0 = K4
    int relu;
    const unsigned int* guard;   // GUARDED instantiations only (conv_mfma.h: conv3x3_mfma_kernel): the launch does nothing while this device word is zero
};

// MT = 16-row tiles per wave, NT = 16-column tiles per workgroup: workgroup tile (64*MT) x (16*NT); BK = K slab.
template <int MT, int NT, int BK = 32>
struct GemmTile {
    static constexpr int BM = 64 * MT, BN = 16 * NT, BS = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int AS = BK + 2;                    // bank = 2*row + k
    static constexpr int A_PER = BM * BK / 256, B_PER = (BK * BN + 255) / 256;
    static constexpr size_t LDS = (size_t)(2 * BM * AS + 2 * BK * BS) * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int MT, int NT, int BK = 32, int WPC = 2, bool GUARDED = false>
__global__ __launch_bounds__(256, WPC) void gemm_mfma_kernel(GemmArgs g) {
    if constexpr (GUARDED) { if (*g.guard == 0u) return; }
    using T = GemmTile<MT, NT, BK>;
    constexpr int kGemmBK = BK, kGemmAS = T::AS;
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
    constexpr int RSTEP = 256 / BK;                     // A slab: rows a_row + RSTEP*i, column a_col
    const int a_col = tid % BK, a_row = tid / BK;
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < T::A_PER; ++i) {
            const int row = m0 + a_row + RSTEP * i, k = k0 + a_col;
            ra[i] = (row < g.M && k < kmax) ? P.A[(size_t)row * P.lda + k] : 0.f;
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
        for (int i = 0; i < T::A_PER; ++i) da[(a_row + RSTEP * i) * kGemmAS + a_col] = ra[i];
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
                if (row < g.M) gemm_store(P, g.relu, row, col, acc[i][j][r] + bv);
            }
        }
}

// The throughput configuration with 16-byte global loads (every operand row a multiple of 4 floats and 16-byte aligned - the
// inference GEMMs: lda 540 / 556, packed W rows of 208 / 560 / 272 floats): workgroup tile (64 MT) x (16 NT), K slabs of 16.
// A thread fetches MT float4 of A (row (t >> 2) + 64 i, k = 4 (t & 3) .. + 3) and up to two float4 of W per slab instead of 14
// single floats, and parks them in LDS with 8- / 16-byte stores; the LDS images and the fragment reads are those of
// gemm_mfma_kernel (A rows of 18 floats: bank = 2 row + k; W rows padded to 16 mod 32).  Rows >= M and k >= K read as zero.
template <int MT, int NT, int WPC, bool GUARDED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPC, WPC))) void gemm_mfma_v4_kernel(GemmArgs g) {
    if constexpr (GUARDED) { if (*g.guard == 0u) return; }
    constexpr int BK = 16, BM = 64 * MT, BN = 16 * NT, BS = (BN % 32 == 16) ? BN : BN + 16, AS = BK + 2;
    constexpr int B4 = BK * BN / 4;                                    // float4 of a W slab
    static_assert(B4 <= 512 && BN % 4 == 0, "two W pieces per thread at most");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                                  // [2][BM][AS]
    float* s_b = smem + 2 * BM * AS;                    // [2][BK][BS]
    const int z = blockIdx.z;
    const GemmProblem P = z == 0 ? g.p[0] : (z == 1 ? g.p[1] : g.p[2]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kmax = g.K ? g.K : g.K4;

    // this thread's pieces: A rows a_row + 64 i at k offset 4 a_k4; W rows b_kr[j], columns n0 + 4 b_c4[j]
    const int a_k4 = tid & 3, a_row = tid >> 2;
    const float* __restrict__ ap0 = P.A + (size_t)min(m0 + a_row, g.M - 1) * P.lda + 4 * a_k4;      // row a_row + 64 i: ap0 + 64 i lda
    const size_t a_step = (size_t)64 * P.lda;
    const int a_rows_left = g.M - (m0 + a_row);                        // row a_row + 64 i exists while 64 i < a_rows_left
    // (named scalars, not small arrays indexed inside a lambda: those end up in scratch memory)
    const int b_kr0 = tid / (BN / 4), b_c40 = tid - b_kr0 * (BN / 4);
    const int b_kr1 = (tid + 256) / (BN / 4), b_c41 = (tid + 256) - b_kr1 * (BN / 4);
    const bool b_ok0 = tid < B4 && n0 + 4 * b_c40 < P.NS, b_ok1 = tid + 256 < B4 && n0 + 4 * b_c41 < P.NS;
    const float* __restrict__ bp0 = P.W + (size_t)b_kr0 * P.NS + n0 + 4 * b_c40;
    const float* __restrict__ bp1 = P.W + (size_t)b_kr1 * P.NS + n0 + 4 * b_c41;
    float4 ra[MT], rb0, rb1;
#define SCS_V4_LOAD(k0)                                                                                                   \
    {                                                                                                                     \
        const bool ka = (k0) + 4 * a_k4 < g.K4;                                                                           \
        /* (assign-then-overwrite, not `cond ? *p : zero`: the compiler turns that into a select of POINTERS - one of them into   \
           scratch memory - behind a flat load) */                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                   \
            ra[i] = float4{0.f, 0.f, 0.f, 0.f};                                                                            \
            if (64 * i < a_rows_left && ka) ra[i] = *reinterpret_cast<const float4*>(ap0 + i * a_step + (k0));              \
        }                                                                                                                 \
        rb0 = float4{0.f, 0.f, 0.f, 0.f}; rb1 = float4{0.f, 0.f, 0.f, 0.f};                                                 \
        if (b_ok0 && (k0) + b_kr0 < kmax) rb0 = *reinterpret_cast<const float4*>(bp0 + (size_t)(k0) * P.NS);                \
        if (b_ok1 && (k0) + b_kr1 < kmax) rb1 = *reinterpret_cast<const float4*>(bp1 + (size_t)(k0) * P.NS);                \
    }
#define SCS_V4_STORE(buf)                                                                                                 \
    {                                                                                                                     \
        float* da = s_a + (buf) * BM * AS + a_row * AS + 4 * a_k4;                                                         \
        float* db = s_b + (buf) * BK * BS;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                   \
            *reinterpret_cast<float2*>(da + 64 * i * AS) = float2{ra[i].x, ra[i].y};                                        \
            *reinterpret_cast<float2*>(da + 64 * i * AS + 2) = float2{ra[i].z, ra[i].w};                                    \
        }                                                                                                                 \
        if (tid < B4) *reinterpret_cast<float4*>(db + b_kr0 * BS + 4 * b_c40) = rb0;                                        \
        if (tid + 256 < B4) *reinterpret_cast<float4*>(db + b_kr1 * BS + 4 * b_c41) = rb1;                                  \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nslab = (g.K4 + BK - 1) / BK;
    SCS_V4_LOAD(0)
    SCS_V4_STORE(0)
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        if (s + 1 < nslab) SCS_V4_LOAD((s + 1) * BK)
        const float* sa = s_a + (s & 1) * BM * AS + (wave * 16 * MT + lr) * AS + kq;
        const float* sb = s_b + (s & 1) * BK * BS + kq * BS + lr;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = sa[i * 16 * AS + ks * 4];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = sb[ks * 4 * BS + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab) SCS_V4_STORE((s + 1) & 1)
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
                if (row < g.M) gemm_store(P, g.relu, row, col, acc[i][j][r] + bv);
            }
        }
}

#undef SCS_V4_LOAD
#undef SCS_V4_STORE

template <int MT, int NT>
struct GemmV4Tile {
    static constexpr int BM = 64 * MT, BN = 16 * NT, BS = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr size_t LDS = (size_t)(2 * BM * 18 + 2 * 16 * BS) * 4;
};

// Latency configuration for the training batches (M = 256: a few dozen 64-row tiles on 256 CUs, one wave per SIMD, every
// k-step waiting for its own operands): workgroup tile 16 rows x 48 columns, and the FOUR WAVES SPLIT K - wave w owns a
// contiguous quarter of the k-steps.  Since no operand is shared between the waves there is nothing to stage: every lane
// loads its own MFMA operands straight from global memory, up to 36 k-steps (36 + 108 loads) in flight at once, so a
// K = 540 layer is ONE global round trip per wave instead of a slab pipeline.  The four partial accumulators are added
// through LDS in a fixed order (deterministic).
struct GemmSplitK {
    static constexpr int BM = 16, NT = 3, BN = 48, CH = 36;      // CH = k-steps in flight per wave
    static constexpr size_t LDS = (size_t)4 * NT * 256 * 4;
};

template <int CH_ = GemmSplitK::CH, bool GUARDED = false>     // a template so that both translation units may instantiate it
__global__ __launch_bounds__(256, 2) void gemm_splitk_kernel(GemmArgs g) {
    if constexpr (GUARDED) { if (*g.guard == 0u) return; }
    using T = GemmSplitK;
    __shared__ f32x4 s_r[4 * T::NT * 64];
    const int z = blockIdx.z;
    const GemmProblem P = z == 0 ? g.p[0] : (z == 1 ? g.p[1] : g.p[2]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * T::BM, n0 = blockIdx.y * T::BN;
    const int kmax = g.K ? g.K : g.K4;
    const int steps = g.K4 / 4, per = (steps + 3) / 4;
    const int s_beg = min(steps, wave * per), s_end = min(steps, s_beg + per);
    const int row = m0 + lr;
    const bool row_ok = row < g.M;
    const float* __restrict__ arow = P.A + (size_t)(row_ok ? row : 0) * P.lda + kq;
    const float* __restrict__ wcol = P.W + (size_t)kq * P.NS + n0 + lr;
    bool col_ok[T::NT];
#pragma unroll
    for (int j = 0; j < T::NT; ++j) col_ok[j] = n0 + j * 16 + lr < P.NS;
    f32x4 acc[T::NT];
#pragma unroll
    for (int j = 0; j < T::NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_beg; s0 < s_end; s0 += T::CH) {
        float av[T::CH], bv[T::CH][T::NT];
#pragma unroll
        for (int i = 0; i < T::CH; ++i) {
            const int ks = s0 + i, k = 4 * ks + kq;
            const bool on = ks < s_end;
            av[i] = (on && row_ok && k < kmax) ? arow[4 * ks] : 0.f;
#pragma unroll
            for (int j = 0; j < T::NT; ++j) bv[i][j] = (on && k < kmax && col_ok[j]) ? wcol[(size_t)4 * ks * P.NS + j * 16] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < T::CH; ++i)
#pragma unroll
            for (int j = 0; j < T::NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][j], acc[j], 0, 0, 0);
    }
    // partial accumulators -> LDS [wave][tile][lane], summed in wave order by the wave that owns the column tile
#pragma unroll
    for (int j = 0; j < T::NT; ++j) s_r[(wave * T::NT + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (wave < T::NT) {
        const int j = wave;
        f32x4 v = s_r[(0 * T::NT + j) * 64 + lane];
#pragma unroll
        for (int q = 1; q < 4; ++q) { const f32x4 u = s_r[(q * T::NT + j) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        const int col = n0 + j * 16 + lr;
        if (col < g.N) {
            const float bvv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                if (orow < g.M) gemm_store(P, g.relu, orow, col, v[r] + bvv);
            }
        }
    }
}

// The same latency configuration with 16-byte loads (training batches; every operand row a multiple of 4 floats and 16-byte
// aligned): workgroup tile 16 rows x 64 columns, EIGHT waves split K in groups of 16.  Per group a lane loads ONE float4 of A
// (row lr, k = 16 G + 4 kq + 0..3) and FOUR float4 of W (rows k = 16 G + 4 kq + i, columns n0 + 4 lr + 0..3) and issues 16
// MFMAs: MFMA (i, j) contracts the k-slots {16 G + 4 q + i : q = 0..3} - the same four for the A and the B operand of a lane,
// which is all the instruction asks - into the column set {n0 + 4 c + j : c = 0..15}.  A K = 540 layer is 5 groups = 25 loads
// per lane, all in flight at once, instead of 136 four-byte loads.  Partial accumulators are added through LDS in wave order.
struct GemmSplitK4 { static constexpr int BM = 16, BN = 64, NW = 8, GMAX = 5; };

template <int GMAX_ = GemmSplitK4::GMAX, bool GUARDED = false>     // a template so that both translation units may instantiate it
__global__ __launch_bounds__(512, 1) void gemm_splitk4_kernel(GemmArgs g) {
    if constexpr (GUARDED) { if (*g.guard == 0u) return; }
    using T = GemmSplitK4;
    __shared__ f32x4 s_r[T::NW * 4 * 64];
    const int z = blockIdx.z;
    const GemmProblem P = z == 0 ? g.p[0] : (z == 1 ? g.p[1] : g.p[2]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * T::BM, n0 = blockIdx.y * T::BN;
    const int kmax = g.K ? g.K : g.K4;
    const int groups = (g.K4 + 15) / 16, per = (groups + T::NW - 1) / T::NW;
    const int g_beg = min(groups, wave * per), g_end = min(groups, g_beg + per);
    const int row = m0 + lr;
    const bool row_ok = row < g.M;
    const float* __restrict__ arow = P.A + (size_t)(row_ok ? row : 0) * P.lda + 4 * kq;
    const int col0 = n0 + 4 * lr;
    const bool col_ok = col0 < P.NS;                       // NS is a multiple of 4: a piece is inside or outside as a whole
    const float* __restrict__ wcol = P.W + col0;
    // what the epilogue of waves 0..3 needs (bias, dropout mask, ReLU gate of its 4 outputs) is fetched with the operands:
    // one global round trip per launch, not two
    const int ecol = n0 + 4 * lr + wave;
    float e_bias = 0.f, e_gate[4] = {1.f, 1.f, 1.f, 1.f};
    unsigned char e_mask[4] = {1, 1, 1, 1};
    if (wave < 4 && ecol < g.N) {
        if (P.bias) e_bias = P.bias[ecol];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = min(m0 + kq * 4 + r, g.M - 1);
            if (P.mask) e_mask[r] = P.mask[(size_t)orow * P.ldm + ecol];
            if (P.gate) e_gate[r] = P.gate[(size_t)orow * P.ldg + ecol];
        }
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g0 = g_beg; g0 < g_end; g0 += T::GMAX) {
        float4 av[T::GMAX], wv[T::GMAX][4];
#pragma unroll
        for (int u = 0; u < T::GMAX; ++u) {
            const int k = 16 * (g0 + u) + 4 * kq;
            const bool on = (g0 + u) < g_end;
            av[u] = (on && row_ok && k < g.K4) ? *reinterpret_cast<const float4*>(arow + 16 * (g0 + u)) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wv[u][i] = (on && col_ok && k + i < kmax) ? *reinterpret_cast<const float4*>(wcol + (size_t)(k + i) * P.NS) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < T::GMAX; ++u) {
            const float a4[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float w4[4] = {wv[u][i].x, wv[u][i].y, wv[u][i].z, wv[u][i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], w4[j], acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_r[(wave * 4 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (wave < 4) {
        const int j = wave;
        f32x4 v = s_r[(0 * 4 + j) * 64 + lane];
#pragma unroll
        for (int q = 1; q < T::NW; ++q) { const f32x4 u = s_r[(q * 4 + j) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        if (ecol < g.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                if (orow < g.M) {                                   // gemm_store with the prefetched mask / gate
                    float o = v[r] + e_bias;
                    if (g.relu) o = fmaxf(o, 0.f);
                    if (P.C2) {
                        P.C[(size_t)orow * P.ldc + ecol] = o;
                        P.C2[(size_t)orow * P.ldc2 + ecol] = e_mask[r] ? 2.f * o : 0.f;
                    } else {
                        if (P.mask) o = e_mask[r] ? 2.f * o : 0.f;
                        if (P.gate) o = e_gate[r] > 0.f ? o : 0.f;
                        P.C[(size_t)orow * P.ldc + ecol] = o;
                    }
                }
            }
        }
    }
}

inline bool gemm_vec4_ok(const GemmArgs& g, int nproblems) {
    if (g.K4 % 4 != 0) return false;
    for (int i = 0; i < nproblems; ++i) {
        const GemmProblem& P = g.p[i];
        if (P.lda % 4 != 0 || P.NS % 4 != 0 || (reinterpret_cast<uintptr_t>(P.A) & 15) || (reinterpret_cast<uintptr_t>(P.W) & 15)) return false;
    }
    return true;
}

// Weight (and bias) gradients of the dense layers, all layers in ONE launch (blockIdx.z = problem):
//   C[m][n] = sum_k A[k][m] * W[k][n],  k < K = batch;  row m = M (when bias_out is set) = sum_k W[k][n]
// A = the layer's input activations, row-major [batch][features] (read "transposed": the 16 lanes of an operand group
// load 16 consecutive floats of one batch row), W = the upstream gradient [batch][outputs].  Same latency structure as
// gemm_splitk_kernel: the four waves split the batch, every lane loads its own operands straight from global memory, the
// four partial accumulators are added through LDS in wave order (deterministic).
struct WgProblem {
    const float* A; int lda;
    const float* W; int ldw;
    float* C; int ldc;
    float* bias_out;             // [N] or null
    int M, N;
};
constexpr int kWgProblems = 6;
struct WgGroupArgs {
    WgProblem p[kWgProblems]; int K;
    const float* loss_rows; float* loss_mean;      // one extra workgroup (blockIdx.z = kWgProblems): mean of the K per-row losses
    const unsigned int* poison;                    // or null: a non-zero word turns the mean into NaN (option train_split: a value left the f16 range)
};

// Workgroup tile: 16 MT rows x 16 NT columns.  32 x 64 makes the six problems of the training step 405 workgroups - one
// generation at two per CU - where 16 x 48 made 1 043 (profiles/r03_notes.txt: 28 -> 17 us).
constexpr int kWgMT = 2, kWgNT = 4;
template <int CH_ = 16>
__global__ __launch_bounds__(256, 2) void dense_wgrad_kernel(WgGroupArgs g) {
    constexpr int MT = kWgMT, NT = kWgNT, CH = CH_;
    __shared__ f32x4 s_r[4 * MT * NT * 64];
    if ((int)blockIdx.z == kWgProblems) {
        // the batch-mean loss: one workgroup, double, fixed order (lanes by shuffle, then waves 0..3)
        if (blockIdx.x != 0 || blockIdx.y != 0 || !g.loss_rows) return;
        double v = 0;
        for (int i = threadIdx.x; i < g.K; i += 256) v += (double)g.loss_rows[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        double* s_d = reinterpret_cast<double*>(s_r);
        if ((threadIdx.x & 63) == 0) s_d[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool bad = g.poison && *g.poison != 0u;
            g.loss_mean[0] = bad ? __builtin_nanf("") : (float)(((s_d[0] + s_d[1]) + (s_d[2] + s_d[3])) / g.K);
        }
        return;
    }
    WgProblem P = g.p[0];
#pragma unroll
    for (int i = 1; i < kWgProblems; ++i) if ((int)blockIdx.z == i) P = g.p[i];
    const int rows = P.M + (P.bias_out ? 1 : 0);
    const int m0 = blockIdx.x * 16 * MT, n0 = blockIdx.y * 16 * NT;
    if (m0 >= rows || n0 >= P.N) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int steps = (g.K + 3) / 4, per = (steps + 3) / 4;
    const int s_beg = min(steps, wave * per), s_end = min(steps, s_beg + per);
    bool m_act[MT], m_one[MT], col_ok[NT];
    const float* __restrict__ acol[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + lr;
        m_act[t] = m < P.M; m_one[t] = (m == P.M) && P.bias_out != nullptr;
        acol[t] = P.A + (m_act[t] ? m : 0);
    }
    const float* __restrict__ wcol = P.W + n0 + lr;
#pragma unroll
    for (int j = 0; j < NT; ++j) col_ok[j] = n0 + j * 16 + lr < P.N;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_beg; s0 < s_end; s0 += CH) {
        float av[CH][MT], bv[CH][NT];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = 4 * (s0 + i) + kq;
            const bool on = (s0 + i) < s_end && k < g.K;
#pragma unroll
            for (int t = 0; t < MT; ++t) av[i][t] = on ? (m_act[t] ? acol[t][(size_t)k * P.lda] : (m_one[t] ? 1.f : 0.f)) : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[i][j] = (on && col_ok[j]) ? wcol[(size_t)k * P.ldw + j * 16] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][t], bv[i][j], acc[t][j], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) s_r[(wave * MT * NT + t * NT + j) * 64 + lane] = acc[t][j];
    __syncthreads();
    // the four waves' partial tiles are added in wave order; wave w finishes tiles w, w + 4 of the MT x NT
    for (int e = wave; e < MT * NT; e += 4) {
        const int t = e / NT, j = e - t * NT;
        f32x4 v = s_r[(0 * MT * NT + e) * 64 + lane];
#pragma unroll
        for (int q = 1; q < 4; ++q) { const f32x4 u = s_r[(q * MT * NT + e) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        const int col = n0 + j * 16 + lr;
        if (col < P.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + 16 * t + kq * 4 + r;
                if (orow < P.M) P.C[(size_t)orow * P.ldc + col] = v[r];
                else if (orow == P.M && P.bias_out) P.bias_out[col] = v[r];
            }
        }
    }
}

inline void dense_wgrad_launch(const WgGroupArgs& g, int nproblems, hipStream_t st) {
    int mr = 0, mc = 0;
    for (int i = 0; i < nproblems; ++i) {
        const int rows = g.p[i].M + (g.p[i].bias_out ? 1 : 0);
        mr = rows > mr ? rows : mr; mc = g.p[i].N > mc ? g.p[i].N : mc;
    }
    hipLaunchKernelGGL(dense_wgrad_kernel<>, dim3((mr + 16 * kWgMT - 1) / (16 * kWgMT), (mc + 16 * kWgNT - 1) / (16 * kWgNT), nproblems + (g.loss_rows ? 1 : 0)),
                       dim3(256), 0, st, g);
}

#ifndef SCS_GEMM_MT
#define SCS_GEMM_MT 2            // 16-row tiles per wave of the vectorised throughput configuration (workgroup tile 64 MT x 96) and
#define SCS_GEMM_V4WPC 4         // workgroups per CU: d1 / FC1 / fc_2 at 91 / 92 / 75 TFLOP/s (71 / 64 / 56 with MT 4 and 2 per CU, 76 / 76 / 71
                                 // with 3 and 3; the scalar-load form they replace: 77 / 78 / 65) - profiles/r03_gemm_ab.txt
#endif
#ifndef SCS_GEMM_BK
#define SCS_GEMM_BK 16           // K slab and workgroups per CU of the throughput (128 x 96) configuration: a 16-wide
#define SCS_GEMM_WPC 3           // slab keeps the LDS image at 32 KB, so 3-4 workgroups share a CU (BK 32 / 2 per CU: -9 %)
#endif

inline hipError_t gemm_set_attrs() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<2, 6, SCS_GEMM_BK, SCS_GEMM_WPC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)GemmTile<2, 6, SCS_GEMM_BK>::LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<1, 3, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GemmTile<1, 3, 32>::LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_v4_kernel<SCS_GEMM_MT, 6, SCS_GEMM_V4WPC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GemmV4Tile<SCS_GEMM_MT, 6>::LDS);
    if (e != hipSuccess) return e;
    return hipSuccess;                                  // gemm_splitk_kernel uses static LDS only
}

// big tile (128 x 96) when it fills the chip, small tile (64 x 48) for the short-M / short-N GEMMs, and the split-K
// latency configuration when even that leaves most CUs without a workgroup (training batches)
inline void gemm_launch(const GemmArgs& g, int nproblems, hipStream_t st) {
    if (g.M <= 0) return;
    using Big = GemmTile<2, 6, SCS_GEMM_BK>;
    using Small = GemmTile<1, 3, 32>;
    using Lat = GemmSplitK;
    const long big_wgs = (long)((g.M + Big::BM - 1) / Big::BM) * ((g.N + Big::BN - 1) / Big::BN) * nproblems;
    const long small_wgs = (long)((g.M + Small::BM - 1) / Small::BM) * ((g.N + Small::BN - 1) / Small::BN) * nproblems;
    using V4 = GemmV4Tile<SCS_GEMM_MT, 6>;
    const long v4_wgs = (long)((g.M + V4::BM - 1) / V4::BM) * ((g.N + V4::BN - 1) / V4::BN) * nproblems;
    if (v4_wgs >= 192 && gemm_vec4_ok(g, nproblems)) {
        dim3 grid((g.M + V4::BM - 1) / V4::BM, (g.N + V4::BN - 1) / V4::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_v4_kernel<SCS_GEMM_MT, 6, SCS_GEMM_V4WPC>), grid, dim3(256), V4::LDS, st, g);
    } else if (big_wgs >= 192) {
        dim3 grid((g.M + Big::BM - 1) / Big::BM, (g.N + Big::BN - 1) / Big::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<2, 6, SCS_GEMM_BK, SCS_GEMM_WPC>), grid, dim3(256), Big::LDS, st, g);
    } else if (small_wgs > 256) {
        dim3 grid((g.M + Small::BM - 1) / Small::BM, (g.N + Small::BN - 1) / Small::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<1, 3, 32>), grid, dim3(256), Small::LDS, st, g);
    } else if (gemm_vec4_ok(g, nproblems)) {
        dim3 grid((g.M + GemmSplitK4::BM - 1) / GemmSplitK4::BM, (g.N + GemmSplitK4::BN - 1) / GemmSplitK4::BN, nproblems);
        hipLaunchKernelGGL(gemm_splitk4_kernel<>, grid, dim3(512), 0, st, g);
    } else {
        dim3 grid((g.M + Lat::BM - 1) / Lat::BM, (g.N + Lat::BN - 1) / Lat::BN, nproblems);
        hipLaunchKernelGGL(gemm_splitk_kernel<>, grid, dim3(256), 0, st, g);
    }
}

// ---- the same launches behind a device-side guard (g.guard; conv_mfma.h: conv3x3_mfma_kernel<C, true>): the same kernel choice and
// the same grids as gemm_launch, so a guarded repeat gives gemm_launch's bits.  Templates so that only the translation unit that
// uses them (scs_api.hip: the f32 repeat behind the split inference kernels) instantiates the guarded kernels.
template <int DUMMY = 0>
inline hipError_t gemm_set_attrs_guarded() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<2, 6, SCS_GEMM_BK, SCS_GEMM_WPC, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)GemmTile<2, 6, SCS_GEMM_BK>::LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<1, 3, 32, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GemmTile<1, 3, 32>::LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_v4_kernel<SCS_GEMM_MT, 6, SCS_GEMM_V4WPC, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)GemmV4Tile<SCS_GEMM_MT, 6>::LDS);
}

template <int DUMMY = 0>
inline void gemm_launch_guarded(const GemmArgs& g, int nproblems, hipStream_t st) {
    if (g.M <= 0 || !g.guard) return;
    using Big = GemmTile<2, 6, SCS_GEMM_BK>;
    using Small = GemmTile<1, 3, 32>;
    using Lat = GemmSplitK;
    const long big_wgs = (long)((g.M + Big::BM - 1) / Big::BM) * ((g.N + Big::BN - 1) / Big::BN) * nproblems;
    const long small_wgs = (long)((g.M + Small::BM - 1) / Small::BM) * ((g.N + Small::BN - 1) / Small::BN) * nproblems;
    using V4 = GemmV4Tile<SCS_GEMM_MT, 6>;
    const long v4_wgs = (long)((g.M + V4::BM - 1) / V4::BM) * ((g.N + V4::BN - 1) / V4::BN) * nproblems;
    if (v4_wgs >= 192 && gemm_vec4_ok(g, nproblems)) {
        dim3 grid((g.M + V4::BM - 1) / V4::BM, (g.N + V4::BN - 1) / V4::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_v4_kernel<SCS_GEMM_MT, 6, SCS_GEMM_V4WPC, true>), grid, dim3(256), V4::LDS, st, g);
    } else if (big_wgs >= 192) {
        dim3 grid((g.M + Big::BM - 1) / Big::BM, (g.N + Big::BN - 1) / Big::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<2, 6, SCS_GEMM_BK, SCS_GEMM_WPC, true>), grid, dim3(256), Big::LDS, st, g);
    } else if (small_wgs > 256) {
        dim3 grid((g.M + Small::BM - 1) / Small::BM, (g.N + Small::BN - 1) / Small::BN, nproblems);
        hipLaunchKernelGGL((gemm_mfma_kernel<1, 3, 32, 2, true>), grid, dim3(256), Small::LDS, st, g);
    } else if (gemm_vec4_ok(g, nproblems)) {
        dim3 grid((g.M + GemmSplitK4::BM - 1) / GemmSplitK4::BM, (g.N + GemmSplitK4::BN - 1) / GemmSplitK4::BN, nproblems);
        hipLaunchKernelGGL((gemm_splitk4_kernel<GemmSplitK4::GMAX, true>), grid, dim3(512), 0, st, g);
    } else {
        dim3 grid((g.M + Lat::BM - 1) / Lat::BM, (g.N + Lat::BN - 1) / Lat::BN, nproblems);
        hipLaunchKernelGGL((gemm_splitk_kernel<GemmSplitK::CH, true>), grid, dim3(256), 0, st, g);
    }
}

}  // namespace scs
