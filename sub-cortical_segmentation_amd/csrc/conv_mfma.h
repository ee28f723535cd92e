// im2col-free 3x3 "valid" convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// One workgroup (4 waves) owns G whole patches.  Their input planes sit in LDS ([g][cin][PLANE]); the
// GEMM view is  out[pixel][cout] = sum_k in[pixel @ (cin,tap)] * W[(cin,tap)][cout]  with
//   M = G * output pixels (16 per MFMA tile), N = Cout padded to 16s, K = Cin*9 walked as
//   (cin-group of 4) x (9 taps) so that one k-step = 4 consecutive input channels at one tap:
//   lane l supplies A[pixel l&15][cin 4c+(l>>4)] (one ds_read_b32 at  base + tap offset) and
//   B[cin 4c+(l>>4)][cout l&15] (one ds_read_b32 of the packed weight chunk).
// All of a wave's accumulator tiles stay live (K outermost), so each weight fragment is read once per
// k-step per wave and each activation once per N-sweep; weights stream through LDS in per-cin-group
// chunks (9*4*NS floats), double buffered against the MFMAs.
//
// With POOL the 16 rows of an M tile are 4 pooling windows x 4 window positions, so the 4 accumulator
// registers of a lane (rows 4*(l>>4)+r) are exactly one 2x2 window: max-pool is an in-lane max, then
// bias + ReLU (both monotone, so pooling first is exact).
//
// Replaces the Conv2DLayer -> BatchNormLayer -> rectify -> (MaxPool2DLayer) chains of
// /root/reference/cnn_cort/nets.py:171-177 (BN folded into W/bias at inference; raw conv in training).
#pragma once
#include "scs_common.h"

namespace scs {

enum ConvEpi { EPI_BIAS_RELU = 0, EPI_BIAS_RELU_POOL = 1, EPI_RAW = 2 };
enum ConvSrc { SRC_NCHW = 0, SRC_CONV1 = 1, SRC_VOLUME = 2, SRC_NCHW_PAD2 = 3 };

template <int CIN_, int COUT_, int HIN_, int G_, int EPI_, int SRC_, int PLANE_>
struct ConvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, G = G_, EPI = EPI_, SRC = SRC_;
    // SRC_NCHW_PAD2: the global input is (HIN_-4)^2 and is zero-padded by 2 on every side in LDS (dgrad)
    static constexpr int HIN = HIN_, WIN = HIN_;
    static constexpr int HSRC = (SRC_ == SRC_NCHW_PAD2) ? HIN_ - 4 : HIN_;
    static constexpr int HO = HIN_ - 2;
    static constexpr bool POOL = (EPI_ == EPI_BIAS_RELU_POOL);
    static constexpr int HP = HO / 2;
    static constexpr int PIX = HO * HO;                 // conv outputs per patch
    static constexpr int OUTPIX = POOL ? HP * HP : PIX; // stored outputs per patch and channel
    static constexpr int PLANE = PLANE_;
    static constexpr int NT = n_tiles(COUT_);
    static constexpr int NS = n_stride(COUT_);
    static constexpr int NC4 = CIN_ / 4;
    static constexpr int MTOT = G_ * PIX;
    static constexpr int TILES = (MTOT + 15) / 16;
    static constexpr int TPW = (TILES + 3) / 4;         // tiles per wave
    static constexpr int WCHUNK = 36 * NS;              // floats per weight chunk
    static constexpr int WCHUNK4 = WCHUNK / 4;
    static constexpr int WPT = (WCHUNK4 + 255) / 256;   // float4 per thread per chunk
    static constexpr int IN_FLOATS = G_ * CIN_ * PLANE_;
    static constexpr int IN_FLOATS_AL = (IN_FLOATS + 3) / 4 * 4;
    static constexpr int XS_FLOATS = (SRC_ == SRC_CONV1 || SRC_ == SRC_VOLUME) ? (32 * 32 + 9 * 20 + 32) : 0;
    static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS_AL + 2 * WCHUNK + XS_FLOATS) * 4;
    static_assert(CIN_ % 4 == 0, "k-steps are groups of 4 input channels");
    static_assert(!POOL || (HO % 2 == 0), "pooling needs an even conv output");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct ConvArgs {
    const float* in[kTowers];     // SRC_NCHW*: [n][CIN][HSRC][HSRC]; SRC_CONV1: [n][32][32]; SRC_VOLUME: the volume
    const float* w[kTowers];      // packed [CIN/4][9][4][NS]
    const float* bias[kTowers];   // [NT*16] (unused for EPI_RAW)
    float* out[kTowers];          // [n][COUT][OUTPIX]
    const float* w1[kTowers];     // SRC_CONV1/VOLUME: conv1 weights [9][20] and bias [32]
    const float* b1[kTowers];
    const int32_t* centers;       // SRC_VOLUME: [n][3]
    int dimx, dimy, dimz;
    int n;                        // patches
};

// kernel-argument arrays indexed by the (wave-uniform) tower id: select, so the struct stays in SGPRs
template <class T>
__device__ __forceinline__ T pick3(T const (&arr)[kTowers], int t) {
    return t == 0 ? arr[0] : (t == 1 ? arr[1] : arr[2]);
}

// conv1 (1 -> 20 channels, K = 9) on the vector ALU, 32x32 LDS tile -> 20 planes of 30x30 in LDS, bias + ReLU.
// (nets.py:171; 2.8 % of a tower's MACs: not worth a K-padded MFMA pass.)
template <int PLANE>
__device__ __forceinline__ void conv1_to_lds(const float* __restrict__ s_x, const float* __restrict__ s_w1,
                                             const float* __restrict__ s_b1, float* __restrict__ s_out, int tid) {
    for (int p = tid; p < 900; p += 256) {
        const int y = p / 30, x = p - y * 30;
        float in[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) in[t] = s_x[(y + t / 3) * 32 + x + t % 3];
        float o[20];
#pragma unroll
        for (int c = 0; c < 20; ++c) o[c] = s_b1[c];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 20; ++c) o[c] = fmaf(in[t], s_w1[t * 20 + c], o[c]);
#pragma unroll
        for (int c = 0; c < 20; ++c) s_out[c * PLANE + p] = fmaxf(o[c], 0.f);
    }
}

template <class C>
__global__ __launch_bounds__(256, 1) void conv3x3_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + C::IN_FLOATS_AL;
    float* s_x = s_w + 2 * C::WCHUNK;     // only when XS_FLOATS > 0: [1024] patch, [180] w1, [32] b1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int tower = blockIdx.y;
    const int g0 = blockIdx.x * C::G;
    const int ng = min(C::G, a.n - g0);

    const float* __restrict__ wsrc = pick3(a.w, tower);
    const float* __restrict__ insrc = pick3(a.in, tower);

    // ---- weight chunk 0 -> registers (lands in LDS after the input staging)
    // (named registers, clamped indices: an indexed array here ends up in scratch behind a vmcnt(0))
    static_assert(C::WPT <= 3, "weight chunk prefetch holds at most 3 float4 per thread");
    const int wi0 = min(tid, C::WCHUNK4 - 1), wi1 = min(tid + 256, C::WCHUNK4 - 1), wi2 = min(tid + 512, C::WCHUNK4 - 1);
    float4 wr0, wr1, wr2;
    wr0 = reinterpret_cast<const float4*>(wsrc)[wi0];
    if constexpr (C::WPT > 1) wr1 = reinterpret_cast<const float4*>(wsrc)[wi1];
    if constexpr (C::WPT > 2) wr2 = reinterpret_cast<const float4*>(wsrc)[wi2];

    // ---- stage the input planes of this workgroup's patches
    if constexpr (C::SRC == SRC_NCHW) {
        constexpr int HW = C::HSRC * C::HSRC;
        const float* __restrict__ src = insrc + (size_t)g0 * C::CIN * HW;
        const int total = ng * C::CIN * HW;
        for (int e = tid; e < C::G * C::CIN * HW; e += 256) {
            const int pl = e / HW, r = e - pl * HW;
            s_in[pl * C::PLANE + r] = (e < total) ? src[e] : 0.f;
        }
    } else if constexpr (C::SRC == SRC_NCHW_PAD2) {
        constexpr int HS = C::HSRC, HW = HS * HS;
        const float* __restrict__ src = insrc + (size_t)g0 * C::CIN * HW;
        for (int e = tid; e < C::G * C::CIN * C::PLANE; e += 256) s_in[e] = 0.f;
        __syncthreads();
        const int total = ng * C::CIN * HW;
        for (int e = tid; e < total; e += 256) {
            const int pl = e / HW, r = e - pl * HW;
            const int y = r / HS, x = r - y * HS;
            s_in[pl * C::PLANE + (y + 2) * C::WIN + x + 2] = src[e];
        }
    } else {
        static_assert(C::SRC == SRC_NCHW || C::SRC == SRC_NCHW_PAD2 || C::G == 1, "fused conv1 variants take one patch per workgroup");
        float* s_w1 = s_x + 1024;
        float* s_b1 = s_w1 + 180;
        if (tid < 180) s_w1[tid] = pick3(a.w1, tower)[tid];
        if (tid < 32) s_b1[tid] = pick3(a.b1, tower)[tid];
        if constexpr (C::SRC == SRC_CONV1) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(insrc + (size_t)g0 * 1024);
            reinterpret_cast<float4*>(s_x)[tid] = src[tid];
        } else {
            // tri-planar gather, zero padded (base.py:293-306): axial [x-16+i][y-16+j][z], coronal
            // [x-16+i][y][z-16+j], sagittal [x][y-16+i][z-16+j]
            const int cx = a.centers[(size_t)g0 * 3 + 0], cy = a.centers[(size_t)g0 * 3 + 1], cz = a.centers[(size_t)g0 * 3 + 2];
            const float* __restrict__ vol = insrc;
            for (int e = tid; e < 1024; e += 256) {
                const int i = e >> 5, j = e & 31;
                int vx, vy, vz;
                if (tower == 0) { vx = cx - 16 + i; vy = cy - 16 + j; vz = cz; }
                else if (tower == 1) { vx = cx - 16 + i; vy = cy; vz = cz - 16 + j; }
                else { vx = cx; vy = cy - 16 + i; vz = cz - 16 + j; }
                const bool ok = (unsigned)vx < (unsigned)a.dimx && (unsigned)vy < (unsigned)a.dimy && (unsigned)vz < (unsigned)a.dimz;
                s_x[e] = ok ? vol[((size_t)vx * a.dimy + vy) * a.dimz + vz] : 0.f;
            }
        }
        __syncthreads();
        conv1_to_lds<C::PLANE>(s_x, s_w1, s_b1, s_in, tid);
    }

    // ---- weight chunk 0 -> LDS
    {
        float4* dst = reinterpret_cast<float4*>(s_w);
        if (tid < C::WCHUNK4) dst[tid] = wr0;
        if constexpr (C::WPT > 1) { if (tid + 256 < C::WCHUNK4) dst[tid + 256] = wr1; }
        if constexpr (C::WPT > 2) { if (tid + 512 < C::WCHUNK4) dst[tid + 512] = wr2; }
    }

    // ---- per-lane A base offsets (floats) for the wave's tiles; out-of-range tiles are clamped, never stored
    int abase[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        int m = (wave + 4 * j) * 16 + lr;
        int g, y, x;
        if constexpr (C::POOL) {
            int widx = min(m >> 2, C::G * C::HP * C::HP - 1);
            const int q = m & 3;
            g = widx / (C::HP * C::HP);
            const int r = widx - g * (C::HP * C::HP);
            const int wy = r / C::HP, wx = r - wy * C::HP;
            y = 2 * wy + (q >> 1);
            x = 2 * wx + (q & 1);
        } else {
            m = min(m, C::MTOT - 1);
            g = m / C::PIX;
            const int r = m - g * C::PIX;
            y = r / C::HO;
            x = r - y * C::HO;
        }
        abase[j] = (g * C::CIN + kq) * C::PLANE + y * C::WIN + x;
    }

    f32x4 acc[C::TPW][C::NT];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();

    const int boff = kq * C::NS + lr;
    for (int c4 = 0; c4 < C::NC4; ++c4) {
        if (c4 + 1 < C::NC4) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wsrc + (size_t)(c4 + 1) * C::WCHUNK);
            wr0 = src[wi0];
            if constexpr (C::WPT > 1) wr1 = src[wi1];
            if constexpr (C::WPT > 2) wr2 = src[wi2];
        }
        const float* __restrict__ sw = s_w + (c4 & 1) * C::WCHUNK + boff;
        const float* __restrict__ sa = s_in + c4 * 4 * C::PLANE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * C::WIN + (tap % 3);
            float b[C::NT];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) b[nt] = sw[tap * 4 * C::NS + nt * 16];
#pragma unroll
            for (int j = 0; j < C::TPW; ++j) {
                const float av = sa[abase[j] + toff];
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt)
                    acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[j][nt], 0, 0, 0);
            }
        }
        if (c4 + 1 < C::NC4) {
            float4* dst = reinterpret_cast<float4*>(s_w + ((c4 + 1) & 1) * C::WCHUNK);
            if (tid < C::WCHUNK4) dst[tid] = wr0;
            if constexpr (C::WPT > 1) { if (tid + 256 < C::WCHUNK4) dst[tid + 256] = wr1; }
            if constexpr (C::WPT > 2) { if (tid + 512 < C::WCHUNK4) dst[tid + 512] = wr2; }
        }
        __syncthreads();
    }

    // ---- epilogue: D row = 4*kq + r (pixel / window position), column = lr (cout)
    const float* __restrict__ biasp = pick3(a.bias, tower);
    float* __restrict__ outp = pick3(a.out, tower) + (size_t)g0 * C::COUT * C::OUTPIX;
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int t = wave + 4 * j;
        if (t >= C::TILES) continue;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
            const int co = nt * 16 + lr;
            if (co >= C::COUT) continue;
            float bv = 0.f;
            if constexpr (C::EPI != EPI_RAW) bv = biasp[co];
            if constexpr (C::POOL) {
                const int widx = t * 4 + kq;
                if (widx < C::G * C::HP * C::HP) {
                    const int g = widx / (C::HP * C::HP);
                    const int r = widx - g * (C::HP * C::HP);
                    if (g < ng) {
                        const f32x4 v = acc[j][nt];
                        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        outp[((size_t)g * C::COUT + co) * C::OUTPIX + r] = fmaxf(mx + bv, 0.f);
                    }
                }
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int m = t * 16 + kq * 4 + r4;
                    if (m < C::MTOT) {
                        const int g = m / C::PIX;
                        const int r = m - g * C::PIX;
                        if (g < ng) {
                            float v = acc[j][nt][r4];
                            if constexpr (C::EPI != EPI_RAW) v = fmaxf(v + bv, 0.f);
                            outp[((size_t)g * C::COUT + co) * C::OUTPIX + r] = v;
                        }
                    }
                }
            }
        }
    }
}

template <class C>
inline hipError_t conv_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<C>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}

template <class C>
inline void conv_launch(const ConvArgs& a, hipStream_t st) {
    if (a.n <= 0) return;
    dim3 grid((a.n + C::G - 1) / C::G, kTowers);
    hipLaunchKernelGGL(conv3x3_mfma_kernel<C>, grid, dim3(256), C::LDS_BYTES, st, a);
}

}  // namespace scs
