// im2col-free 3x3 "valid" convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// One workgroup (NW waves, 4 unless stated) owns G whole patches.  Their input planes sit in LDS ([g][cin][PLANE]); the
// GEMM view is  out[pixel][cout] = sum_k in[pixel @ (cin,tap)] * W[(cin,tap)][cout]  with
//   M = G * output pixels (16 per MFMA tile), N = Cout = 16-wide blocks + 4-wide blocks (scs_common.h: no padding),
//   K = Cin*9 walked as
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

constexpr int kW1Floats = 12 * 16 + 12 * 4 + 20;   // conv1 operands: [12][16] + [12][4] + bias[20]
static_assert(kW1Floats > 256 && kW1Floats <= 512, "conv1 operand staging assumes two loads per thread");

// EPI_RAW_STATS: raw outputs plus, per (patch, band, channel), their sum and sum of squares (batch-norm statistics
// of the training forward pass without a second pass over the activations)
enum ConvEpi { EPI_BIAS_RELU = 0, EPI_BIAS_RELU_POOL = 1, EPI_RAW = 2, EPI_RAW_STATS = 3 };
enum ConvSrc { SRC_NCHW = 0, SRC_CONV1 = 1, SRC_VOLUME = 2, SRC_NCHW_PAD2 = 3 };

template <int CIN_, int COUT_, int HIN_, int G_, int EPI_, int SRC_, int PLANE_, int NB_ = 1, int WPS_ = 1, int STAMP_ = 0, int NW_ = 4>
struct ConvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, G = G_, EPI = EPI_, SRC = SRC_, NB = NB_, WPS = WPS_, STAMP = STAMP_;
    static constexpr int NW = NW_, NTH = 64 * NW_;          // waves / threads per workgroup
    // SRC_NCHW_PAD2: the global input is (HIN_-4)^2 and is zero-padded by 2 on every side in LDS (dgrad)
    static constexpr int HIN = HIN_, WIN = HIN_;
    static constexpr int HSRC = (SRC_ == SRC_NCHW_PAD2) ? HIN_ - 4 : HIN_;
    static constexpr int HO = HIN_ - 2;
    static constexpr bool POOL = (EPI_ == EPI_BIAS_RELU_POOL);
    static constexpr bool RAW = (EPI_ == EPI_RAW || EPI_ == EPI_RAW_STATS);
    static constexpr int HP = HO / 2;
    static constexpr int PIX = HO * HO;                 // conv outputs per patch
    static constexpr int OUTPIX = POOL ? HP * HP : PIX; // stored outputs per patch and channel
    // a workgroup owns band `b` of NB horizontal bands of its G patches: HOB conv-output rows, HINB input rows
    static constexpr int HOB = HO / NB_;
    static constexpr int HINB = HOB + 2;
    static constexpr int PIXB = HOB * HO;               // conv outputs per patch and band
    static constexpr int WB = (HOB / 2) * HP;           // pooling windows per patch and band
    static constexpr int OUTB = POOL ? WB : PIXB;       // stored outputs per patch, channel and band
    static constexpr int XROWS = HINB + 2;              // rows of the raw 32x32 patch a fused-conv1 band needs
    static constexpr int PLANE = PLANE_;
    static constexpr int NT = n_tiles(COUT_);           // bias granularity
    static constexpr int NT16 = nt16(COUT_), NT4 = nt4(COUT_), NS16 = ns16(COUT_), L4 = 4 * NT4;
    static constexpr int NC4 = CIN_ / 4;
    static constexpr int MTOT = G_ * PIXB;
    static constexpr int TILES = (MTOT + 15) / 16;
    static constexpr int TPW = (TILES + NW - 1) / NW;   // tiles per wave (wave w owns tiles w, w+NW, ...)
    static constexpr bool LAST_RAGGED = (NW - 1 + NW * (TPW - 1)) >= TILES;   // some waves lack their last tile
    static constexpr int WCHUNK = conv_chunk_floats(COUT_);   // floats per weight chunk
    static constexpr int WCHUNK4 = WCHUNK / 4;
    static constexpr int WPT = (WCHUNK4 + NTH - 1) / NTH;   // float4 per thread per chunk
    static constexpr int IN_FLOATS = G_ * CIN_ * PLANE_;
    static constexpr int IN_FLOATS_AL = (IN_FLOATS + 3) / 4 * 4;
    static constexpr int XS_ROWS_FLOATS = (XROWS * 32 + 3) / 4 * 4;
    static constexpr int XS_FLOATS = (SRC_ == SRC_CONV1 || SRC_ == SRC_VOLUME) ? XS_ROWS_FLOATS + 4 + kW1Floats : 0;
    // the epilogue restages the outputs in LDS ([g][cout][OSTR], odd stride) and stores them as whole rows
    static constexpr bool OUT_VEC4 = (OUTB % 4 == 0) && (OUTPIX % 4 == 0);
    static constexpr int OSTR = OUT_VEC4 ? OUTB + 4 : (OUTB | 1);
    static constexpr int OUT_FLOATS = G_ * COUT_ * OSTR;
    static constexpr int MAIN_FLOATS = IN_FLOATS_AL + 2 * WCHUNK + XS_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)(MAIN_FLOATS > OUT_FLOATS ? MAIN_FLOATS : OUT_FLOATS) * 4;
    static_assert(CIN_ % 4 == 0, "k-steps are groups of 4 input channels");
    static_assert(COUT_ % 4 == 0, "output channels are covered by 16-wide and 4-wide MFMA blocks");
    static_assert(HO % NB_ == 0 && (!POOL || (HOB % 2 == 0)), "bands must split the output rows (evenly for pooling)");
    static_assert(NB_ == 1 || G_ == 1, "bands: one patch per workgroup");
    static_assert(PLANE_ >= HINB * HIN_, "plane stride too small");
    static_assert(LDS_BYTES * WPS_ <= 160 * 1024, "LDS budget (WPS workgroups per CU)");
    static_assert(NTH >= 8 * XROWS || (SRC_ != SRC_CONV1 && SRC_ != SRC_VOLUME), "raw patch rows are staged in one pass");
};

struct ConvArgs {
    const float* in[kTowers];     // SRC_NCHW*: [n][CIN][HSRC][HSRC]; SRC_CONV1: [n][32][32]; SRC_VOLUME: the volume
    const float* w[kTowers];      // packed chunks per 4 input channels: [9][4][NS16] + [9][4][L4] (scs_common.h)
    const float* bias[kTowers];   // [NT*16] (unused for EPI_RAW)
    float* out[kTowers];          // [n][COUT][OUTPIX]
    const float* w1[kTowers];     // SRC_CONV1/VOLUME: conv1 operands [12][16] + [12][4] + bias[20] (kW1Floats)
    const int32_t* centers;       // SRC_VOLUME: [n][3]
    int dimx, dimy, dimz;
    int n;                        // patches
    float* stats;                 // EPI_RAW_STATS: [tower][n * NB][COUT][2] (sum, sum of squares) per patch band
    unsigned long long* stamps;   // diagnostic builds only (STAMP): [workgroup][8] s_memtime values
};

// kernel-argument arrays indexed by the (wave-uniform) tower id: select, so the struct stays in SGPRs
template <class T>
__device__ __forceinline__ T pick3(T const (&arr)[kTowers], int t) {
    return t == 0 ? arr[0] : (t == 1 ? arr[1] : arr[2]);
}

// ---- cross-lane helpers on the VALU (no LDS crossbar): v + v[lane^16] + v[lane^32] + v[lane^48], and quad maxima
__device__ __forceinline__ float sum_xor16(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // odd rows of one copy <-> even rows of the other
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_ksub(float v) { return sum_xor32(sum_xor16(v)); }

// The 4x4x1 accumulators hold, per lane group kq = lane >> 4, the partial sums of k-subgroup kq for 4 output channels
// (vgpr i).  Lane group kq stores channel i = kq, so what is needed is a 4x4 transpose-and-add across the lane groups:
// two exchange rounds (lane ^ 16 through the LDS crossbar's swizzle, lane ^ 32 through ds_bpermute - neither occupies
// the vector ALU, which the f32 MFMAs of the other waves need) and three adds.
__device__ __forceinline__ float ksub_reduce(const f32x4 v, int kq, int xor32_addr) {
    const bool odd = kq & 1, hi = kq & 2;
    const float keepA = odd ? v[1] : v[0], sendA = odd ? v[0] : v[1];
    const float keepB = odd ? v[3] : v[2], sendB = odd ? v[2] : v[3];
    const float recvA = __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(sendA), 0x401F));   // lane ^ 16
    const float recvB = __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(sendB), 0x401F));
    const float sA = keepA + recvA, sB = keepB + recvB;
    const float keep = hi ? sB : sA, send = hi ? sA : sB;
    const float recv = __uint_as_float(__builtin_amdgcn_ds_bpermute(xor32_addr, __float_as_uint(send)));  // lane ^ 32
    return keep + recv;
}
__device__ __forceinline__ float max_quad(float v) {     // max over the 4 lanes of a quad (lane ^ 1, lane ^ 2)
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xF, 0xF, true)));
    return v;
}

// conv1 (1 -> 20 channels, 9 taps) on the matrix cores as well: K = 9 padded to 12 (3 k-steps, tap index on the
// k axis), N = 20 = 16 + 4, M = ROWS*30 pixels.  Raw tile rows in LDS -> 20 planes of ROWS x 30 in LDS with bias +
// ReLU (nets.py:171).  w1p: [12][16] main weights, [12][4] leftover weights, [20] bias (rows 9..11 are zero).
template <int PLANE, int ROWS, int NW = 4>
__device__ __forceinline__ void conv1_to_lds(const float* __restrict__ s_x, const float* __restrict__ s_w1,
                                             float* __restrict__ s_out, int wave, int lane) {
    constexpr int NPIX = ROWS * 30, T1 = (NPIX + 15) / 16;
    const int lr = lane & 15, kq = lane >> 4;
    int toff[3];
    float bm[3], bl[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int k = 4 * s3 + kq, tk = k < 9 ? k : 8;
        toff[s3] = (tk / 3) * 32 + tk % 3;
        bm[s3] = s_w1[k * 16 + lr];
        bl[s3] = s_w1[192 + k * 4 + (lane & 3)];
    }
    const float bias_m = s_w1[240 + lr], bias_l = s_w1[240 + 16 + kq];
    // all of a wave's tiles are independent straight-line chains (clamped indices, predicated stores), so the
    // scheduler overlaps their LDS-read -> MFMA -> lane-exchange -> LDS-write latencies
    constexpr int TW = (T1 + NW - 1) / NW;
#pragma unroll
    for (int jj = 0; jj < TW; ++jj) {
        const int t = wave + NW * jj;
        const int tc = min(t, T1 - 1);
        const int p = min(tc * 16 + lr, NPIX - 1), y = p / 30, x = p - y * 30;
        f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const float av = s_x[y * 32 + x + toff[s3]];
            am = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bm[s3], am, 0, 0, 0);
            al = __builtin_amdgcn_mfma_f32_4x4x1f32(bl[s3], av, al, 0, 0, 0);
        }
        // main: cout = lr, pixels 16t + 4kq + r (contiguous in the plane: rows are stored unpadded, 30 wide)
        const int p0 = tc * 16 + kq * 4;
        float* dst = s_out + lr * PLANE + p0;
        const bool tile_ok = t < T1;
        if (tile_ok && (NPIX % 16 == 0 || p0 + 3 < NPIX)) {
            *reinterpret_cast<float4*>(dst) = float4{fmaxf(am[0] + bias_m, 0.f), fmaxf(am[1] + bias_m, 0.f),
                                                     fmaxf(am[2] + bias_m, 0.f), fmaxf(am[3] + bias_m, 0.f)};
        } else if (tile_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (p0 + r < NPIX) dst[r] = fmaxf(am[r] + bias_m, 0.f);
        }
        // leftover: cout 16 + i in vgpr i, pixel = lr, partial over kq -> combine, lane group kq stores cout 16 + kq
        const float val = ksub_reduce(al, kq, (lane ^ 32) << 2);
        if (tile_ok && (NPIX % 16 == 0 || tc * 16 + lr < NPIX)) s_out[(16 + kq) * PLANE + tc * 16 + lr] = fmaxf(val + bias_l, 0.f);
    }
}

// batched copy helper: K independent loads in flight per thread before the LDS writes (a plain strided loop
// serialises one global-load latency per iteration)
template <int TOTAL, class V, int NTH = 256, class SrcF, class DstF>
__device__ __forceinline__ void staged_copy(int tid, SrcF&& src_of, DstF&& dst_of) {
    constexpr int ITER = (TOTAL + NTH - 1) / NTH, K = 8;
#pragma unroll
    for (int i0 = 0; i0 < ITER; i0 += K) {
        V v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (i0 + k < ITER) { const int e = min(tid + NTH * (i0 + k), TOTAL - 1); v[k] = src_of(e); }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (i0 + k < ITER) { const int e = tid + NTH * (i0 + k); if (e < TOTAL) dst_of(e, v[k]); }
        }
    }
}

template <class C>
__global__ __launch_bounds__(C::NTH, (C::WPS * C::NW + 3) / 4) void conv3x3_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + C::IN_FLOATS_AL;
    float* s_x = s_w + 2 * C::WCHUNK;     // only when XS_FLOATS > 0: raw rows of the 32x32 patch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int tower = blockIdx.y;
    const int band = (C::NB > 1) ? (int)(blockIdx.x % C::NB) : 0;
    const int g0 = (int)(blockIdx.x / C::NB) * C::G;
    const int ng = min(C::G, a.n - g0);

    unsigned long long* stp = nullptr;
    if constexpr (C::STAMP) {
        stp = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        if (tid == 0) stp[0] = __builtin_amdgcn_s_memtime();
    }
    const float* __restrict__ wsrc = pick3(a.w, tower);
    const float* __restrict__ insrc = pick3(a.in, tower);

    // ---- weight chunk 0 -> registers (lands in LDS after the input staging)
    // (named registers, clamped indices: an indexed array here ends up in scratch behind a vmcnt(0))
    static_assert(C::WPT <= 3, "weight chunk prefetch holds at most 3 float4 per thread");
    const int wi0 = min(tid, C::WCHUNK4 - 1), wi1 = min(tid + C::NTH, C::WCHUNK4 - 1), wi2 = min(tid + 2 * C::NTH, C::WCHUNK4 - 1);
    float4 wr0, wr1, wr2;
    wr0 = reinterpret_cast<const float4*>(wsrc)[wi0];
    if constexpr (C::WPT > 1) wr1 = reinterpret_cast<const float4*>(wsrc)[wi1];
    if constexpr (C::WPT > 2) wr2 = reinterpret_cast<const float4*>(wsrc)[wi2];

    // per-lane epilogue bias, fetched now so its latency hides under the staging
    float bvs[C::NT16 > 0 ? C::NT16 : 1], bv4[C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
    for (int nt = 0; nt < C::NT16; ++nt) {
        bvs[nt] = 0.f;
        if constexpr (!C::RAW) bvs[nt] = pick3(a.bias, tower)[nt * 16 + lr];
    }
#pragma unroll
    for (int u = 0; u < C::NT4; ++u) {
        bv4[u] = 0.f;
        if constexpr (!C::RAW) bv4[u] = pick3(a.bias, tower)[C::NT16 * 16 + u * 4 + kq];
    }

    // ---- stage the input planes of this workgroup's patches
    if constexpr (C::SRC == SRC_NCHW) {
        constexpr int HW = C::HSRC * C::HSRC, BW = C::HINB * C::WIN;      // full plane / band slice of it
        const float* __restrict__ src = insrc + (size_t)g0 * C::CIN * HW + band * C::HOB * C::WIN;
        const int planes = ng * C::CIN;
        if constexpr (BW % 4 == 0 && HW % 4 == 0 && C::PLANE % 4 == 0 && (C::HOB * C::WIN) % 4 == 0) {
            constexpr int BW4 = BW / 4;
            const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
            float4* dst4 = reinterpret_cast<float4*>(s_in);
            staged_copy<C::G * C::CIN * BW4, float4, C::NTH>(
                tid,
                [&](int e) { const int pl = e / BW4, r = e - pl * BW4;
                             return (pl < planes) ? src4[(size_t)pl * (HW / 4) + r] : float4{0.f, 0.f, 0.f, 0.f}; },
                [&](int e, float4 v) { const int pl = e / BW4, r = e - pl * BW4; dst4[pl * (C::PLANE / 4) + r] = v; });
        } else if constexpr (C::PLANE == HW && C::NB == 1 && (C::G * C::CIN * HW) % 4 == 0) {
            // unpadded planes: the LDS image is the global block itself (16-byte aligned: g0 is a multiple of G)
            const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
            float4* dst4 = reinterpret_cast<float4*>(s_in);
            const int tot4 = planes * HW / 4;      // ng*CIN*HW is a multiple of 4 whenever CIN*HW is
            staged_copy<C::G * C::CIN * HW / 4, float4, C::NTH>(
                tid, [&](int e) { return (e < tot4) ? src4[e] : float4{0.f, 0.f, 0.f, 0.f}; },
                [&](int e, float4 v) { dst4[e] = v; });
        } else {
            staged_copy<C::G * C::CIN * BW, float, C::NTH>(
                tid,
                [&](int e) { const int pl = e / BW, r = e - pl * BW; return (pl < planes) ? src[(size_t)pl * HW + r] : 0.f; },
                [&](int e, float v) { const int pl = e / BW, r = e - pl * BW; s_in[pl * C::PLANE + r] = v; });
        }
    } else if constexpr (C::SRC == SRC_NCHW_PAD2) {
        // destination-driven: every element of the (band of the) zero-padded plane is written once, taken from the
        // source when it falls inside it (padded row yp = band*HOB + yl <-> source row yp - 2)
        constexpr int HS = C::HSRC, HW = HS * HS, BW = C::HINB * C::WIN;
        const float* __restrict__ src = insrc + (size_t)g0 * C::CIN * HW;
        const int planes = ng * C::CIN;
        staged_copy<C::G * C::CIN * BW, float, C::NTH>(
            tid,
            [&](int e) {
                const int pl = e / BW, r = e - pl * BW, yl = r / C::WIN, xp = r - yl * C::WIN;
                const int ys = band * C::HOB + yl - 2, xs = xp - 2;
                const bool ok = pl < planes && (unsigned)ys < (unsigned)HS && (unsigned)xs < (unsigned)HS;
                return ok ? src[(size_t)pl * HW + ys * HS + xs] : 0.f;
            },
            [&](int e, float v) { const int pl = e / BW, r = e - pl * BW; s_in[pl * C::PLANE + r] = v; });
    } else {
        static_assert(C::SRC == SRC_NCHW || C::SRC == SRC_NCHW_PAD2 || C::G == 1, "fused conv1 variants take one patch per workgroup");
        // conv1 operands and the raw rows are fetched together: one global round trip, not two
        const float* __restrict__ w1g = pick3(a.w1, tower);
        const float w1a = w1g[min(tid, kW1Floats - 1)];
        const float w1b = w1g[min(tid + C::NTH, kW1Floats - 1)];
        if constexpr (C::SRC == SRC_CONV1) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(insrc + (size_t)g0 * 1024 + band * C::HOB * 32);
            if (tid < C::XROWS * 8) reinterpret_cast<float4*>(s_x)[tid] = src[tid];
        } else {
            // tri-planar gather, zero padded (base.py:293-306): axial [x-16+i][y-16+j][z], coronal
            // [x-16+i][y][z-16+j], sagittal [x][y-16+i][z-16+j]
            const int cx = a.centers[(size_t)g0 * 3 + 0], cy = a.centers[(size_t)g0 * 3 + 1], cz = a.centers[(size_t)g0 * 3 + 2];
            const float* __restrict__ vol = insrc;
            for (int e = tid; e < C::XROWS * 32; e += C::NTH) {
                const int i = band * C::HOB + (e >> 5), j = e & 31;
                int vx, vy, vz;
                if (tower == 0) { vx = cx - 16 + i; vy = cy - 16 + j; vz = cz; }
                else if (tower == 1) { vx = cx - 16 + i; vy = cy; vz = cz - 16 + j; }
                else { vx = cx; vy = cy - 16 + i; vz = cz - 16 + j; }
                const bool ok = (unsigned)vx < (unsigned)a.dimx && (unsigned)vy < (unsigned)a.dimy && (unsigned)vz < (unsigned)a.dimz;
                s_x[e] = ok ? vol[((size_t)vx * a.dimy + vy) * a.dimz + vz] : 0.f;
            }
        }
        float* s_w1 = s_x + C::XS_ROWS_FLOATS + 4;
        if (tid < kW1Floats) s_w1[tid] = w1a;              // kW1Floats = 260: two loads per lane at 256 threads
        if (tid + C::NTH < kW1Floats) s_w1[tid + C::NTH] = w1b;
        __syncthreads();
        if constexpr (C::STAMP) { if (tid == 0) stp[5] = __builtin_amdgcn_s_memtime(); }
#if !(defined(SCS_ABLATE) && SCS_ABLATE == 3)
        conv1_to_lds<C::PLANE, C::HINB, C::NW>(s_x, s_w1, s_in, wave, lane);
#endif
    }

    if constexpr (C::STAMP) { if (tid == 0) stp[1] = __builtin_amdgcn_s_memtime(); }
    // ---- weight chunk 0 -> LDS
    {
        float4* dst = reinterpret_cast<float4*>(s_w);
        if (tid < C::WCHUNK4) dst[tid] = wr0;
        if constexpr (C::WPT > 1) { if (tid + C::NTH < C::WCHUNK4) dst[tid + C::NTH] = wr1; }
        if constexpr (C::WPT > 2) { if (tid + 2 * C::NTH < C::WCHUNK4) dst[tid + 2 * C::NTH] = wr2; }
    }

    // ---- per-lane A base offsets (floats) for the wave's tiles; out-of-range tiles are clamped, never stored
    int abase[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        int m = (wave + C::NW * j) * 16 + lr;
        int g, y, x;
        if constexpr (C::POOL) {
            int widx = min(m >> 2, C::G * C::WB - 1);
            const int q = m & 3;
            g = widx / C::WB;
            const int r = widx - g * C::WB;
            const int wy = r / C::HP, wx = r - wy * C::HP;
            y = 2 * wy + (q >> 1);
            x = 2 * wx + (q & 1);
        } else {
            m = min(m, C::MTOT - 1);
            g = m / C::PIXB;
            const int r = m - g * C::PIXB;
            y = r / C::HO;
            x = r - y * C::HO;
        }
        abase[j] = (g * C::CIN + kq) * C::PLANE + y * C::WIN + x;
#if defined(SCS_ABLATE) && SCS_ABLATE == 6
        abase[j] = (g * C::CIN + kq) * C::PLANE + lr + 16 * j;      // conflict-free A reads (wrong data): what would it buy?
#endif
    }

    f32x4 acc[C::TPW][C::NT16 > 0 ? C::NT16 : 1], acc4[C::TPW][C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
#pragma unroll
        for (int nt = 0; nt < C::NT16; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < C::NT4; ++u) acc4[j][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    __syncthreads();
    if constexpr (C::STAMP) { if (tid == 0) stp[2] = __builtin_amdgcn_s_memtime(); }

    const int boff = kq * C::NS16 + lr;                       // main part: W[k = 4s + kq][cout = 16 nt + lr]
    const int loff = 36 * C::NS16 + kq * C::L4 + (lane & 3);   // leftover: W[k = 4s + kq][cout = 16 NT16 + 4u + (lane&3)]
    const bool last_ok = (wave + C::NW * (C::TPW - 1)) < C::TILES;
    for (int c4 = 0; c4 < C::NC4; ++c4) {
        if (c4 + 1 < C::NC4) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wsrc + (size_t)(c4 + 1) * C::WCHUNK);
            wr0 = src[wi0];
            if constexpr (C::WPT > 1) wr1 = src[wi1];
            if constexpr (C::WPT > 2) wr2 = src[wi2];
        }
        const float* __restrict__ sw = s_w + (c4 & 1) * C::WCHUNK;
        const float* __restrict__ sa = s_in + c4 * 4 * C::PLANE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * C::WIN + (tap % 3);
            float b[C::NT16 > 0 ? C::NT16 : 1], b4[C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
            for (int nt = 0; nt < C::NT16; ++nt) b[nt] = sw[boff + tap * 4 * C::NS16 + nt * 16];
#pragma unroll
            for (int u = 0; u < C::NT4; ++u) b4[u] = sw[loff + tap * 4 * C::L4 + u * 4];
#pragma unroll
            for (int j = 0; j < C::TPW; ++j) {
                if (C::LAST_RAGGED && j == C::TPW - 1 && !last_ok) continue;   // wave-uniform: this wave has no such tile
#if defined(SCS_ABLATE) && SCS_ABLATE == 2
                const float av = __builtin_bit_cast(float, abase[j] + toff);       // ablation: no activation reads
#else
                const float av = sa[abase[j] + toff];
#endif
#if defined(SCS_ABLATE) && SCS_ABLATE == 1
#pragma unroll
                for (int nt = 0; nt < C::NT16; ++nt) acc[j][nt][0] += av * b[nt];  // ablation: no MFMA
#pragma unroll
                for (int u = 0; u < C::NT4; ++u) acc4[j][u][0] += av * b4[u];
#else
#pragma unroll
                for (int nt = 0; nt < C::NT16; ++nt)
                    acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[j][nt], 0, 0, 0);
                // 16 blocks of 4 couts x 4 pixels: block (l>>2) = (k-subgroup kq, pixel quad); partial sums over kq
                // are combined in the epilogue
#pragma unroll
                for (int u = 0; u < C::NT4; ++u)
                    acc4[j][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, acc4[j][u], 0, 0, 0);
#endif
            }
        }
        if (c4 + 1 < C::NC4) {
            float4* dst = reinterpret_cast<float4*>(s_w + ((c4 + 1) & 1) * C::WCHUNK);
            if (tid < C::WCHUNK4) dst[tid] = wr0;
            if constexpr (C::WPT > 1) { if (tid + C::NTH < C::WCHUNK4) dst[tid + C::NTH] = wr1; }
            if constexpr (C::WPT > 2) { if (tid + 2 * C::NTH < C::WCHUNK4) dst[tid + 2 * C::NTH] = wr2; }
        }
        __syncthreads();
    }

    if constexpr (C::STAMP) { if (tid == 0) stp[3] = __builtin_amdgcn_s_memtime(); }

    // ---- epilogue: D row = 4*kq + r (pixel / window position), column = lr (cout).  Results are restaged in
    // LDS as [g][cout][OSTR] (every wave has passed the loop's last barrier, so the operand buffers are free) and
    // leave as whole rows: 64 lanes -> 256 contiguous bytes, instead of 64 scattered dwords per store.
    float* s_o = smem;
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int t = wave + C::NW * j;
        if (t >= C::TILES) continue;
#pragma unroll
        for (int nt = 0; nt < C::NT16; ++nt) {
            const int co = nt * 16 + lr;
            const float bv = bvs[nt];
            if constexpr (C::POOL) {
                const int widx = t * 4 + kq;
                if (widx < C::G * C::WB) {
                    const int g = widx / C::WB;
                    const int r = widx - g * C::WB;
                    const f32x4 v = acc[j][nt];
                    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    s_o[(g * C::COUT + co) * C::OSTR + r] = fmaxf(mx + bv, 0.f);
                }
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int m = t * 16 + kq * 4 + r4;
                    if (m < C::MTOT) {
                        const int g = m / C::PIXB;
                        const int r = m - g * C::PIXB;
                        float v = acc[j][nt][r4];
                        if constexpr (!C::RAW) v = fmaxf(v + bv, 0.f);
                        s_o[(g * C::COUT + co) * C::OSTR + r] = v;
                    }
                }
            }
        }
    }
    // leftover channels: lane l holds, for pixel (l & 15), the partial sums of k-subgroup kq for 4 couts (vgpr i).
    // All cross-lane reductions first, as one branch-free block (independent chains the scheduler can overlap: each
    // has two LDS-crossbar round trips), then the guarded stores.
    const int xor32_addr = (lane ^ 32) << 2;
    float lval[C::TPW][C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j)
#pragma unroll
        for (int u = 0; u < C::NT4; ++u) {
            float val = ksub_reduce(acc4[j][u], kq, xor32_addr);                         // this lane stores cout 4u + kq
            if constexpr (C::POOL) val = max_quad(val);
            lval[j][u] = val;
        }
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int t = wave + C::NW * j;
        if (t >= C::TILES) continue;
#if defined(SCS_ABLATE) && SCS_ABLATE == 4
        if (t >= 0) { s_o[lane] = lval[j][0]; continue; }
#endif
#pragma unroll
        for (int u = 0; u < C::NT4; ++u) {
            float val = lval[j][u];
            const int co = C::NT16 * 16 + u * 4 + kq;
            const float bv = bv4[u];
            if constexpr (C::POOL) {
                const int widx = t * 4 + ((lane & 15) >> 2);
                if ((lane & 3) == 0 && widx < C::G * C::WB) {
                    const int g = widx / C::WB;
                    const int r = widx - g * C::WB;
                    s_o[(g * C::COUT + co) * C::OSTR + r] = fmaxf(val + bv, 0.f);
                }
            } else {
                const int m = t * 16 + (lane & 15);
                if (m < C::MTOT) {
                    const int g = m / C::PIXB;
                    const int r = m - g * C::PIXB;
                    if constexpr (!C::RAW) val = fmaxf(val + bv, 0.f);
                    s_o[(g * C::COUT + co) * C::OSTR + r] = val;
                }
            }
        }
    }
    if constexpr (C::STAMP) { if (tid == 0) stp[6] = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if constexpr (C::STAMP) { if (tid == 0) stp[7] = __builtin_amdgcn_s_memtime(); }
    {
        float* __restrict__ outp = pick3(a.out, tower) + (size_t)g0 * C::COUT * C::OUTPIX + band * C::OUTB;
        if constexpr (C::OUT_VEC4) {
            constexpr int O4 = C::OUTB / 4, ITER = (C::G * C::COUT * O4 + C::NTH - 1) / C::NTH;
            const int total = ng * C::COUT * O4;
#if defined(SCS_ABLATE) && SCS_ABLATE == 5
            if (tid < 0)
#endif
#pragma unroll 4
            for (int it = 0; it < ITER; ++it) {
                const int e = tid + C::NTH * it;
                if (e < total) {
                    const int gc = e / O4, r = e - gc * O4;
                    *reinterpret_cast<float4*>(outp + (size_t)gc * C::OUTPIX + 4 * r) =
                        *reinterpret_cast<const float4*>(s_o + gc * C::OSTR + 4 * r);
                }
            }
        } else {
            const int total = ng * C::COUT * C::OUTB;
            constexpr int ITER = (C::G * C::COUT * C::OUTB + C::NTH - 1) / C::NTH;
#if defined(SCS_ABLATE) && SCS_ABLATE == 5
            if (tid < 0)
#endif
#pragma unroll 2
            for (int it = 0; it < ITER; ++it) {
                const int e = tid + C::NTH * it;
                if (e < total) {
                    const int gc = e / C::OUTB, r = e - gc * C::OUTB;
                    outp[(size_t)gc * C::OUTPIX + r] = s_o[gc * C::OSTR + r];
                }
            }
        }
    }
    if constexpr (C::EPI == EPI_RAW_STATS) {
        // rows of s_o are (patch, channel); LPR lanes share a row, then combine through the crossbar
        constexpr int ROWS = C::G * C::COUT;
        constexpr int LPR = ROWS * 16 <= C::NTH ? 16 : ROWS * 8 <= C::NTH ? 8 : ROWS * 4 <= C::NTH ? 4 : ROWS * 2 <= C::NTH ? 2 : 1;
        constexpr int RPI = C::NTH / LPR;                    // rows per iteration
        float* __restrict__ sp = a.stats + ((size_t)tower * a.n * C::NB) * C::COUT * 2;
        const int sub = tid % LPR;
#pragma unroll
        for (int row0 = 0; row0 < ROWS; row0 += RPI) {
            const int row = row0 + tid / LPR;
            float sm = 0.f, sq = 0.f;
            if (row < ng * C::COUT) {
                const float* __restrict__ rp = s_o + row * C::OSTR;
                for (int i = sub; i < C::OUTB; i += LPR) { const float v = rp[i]; sm += v; sq = fmaf(v, v, sq); }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
            if (sub == 0 && row < ng * C::COUT) {
                const int g = row / C::COUT, c = row - g * C::COUT;
                float2* dst = reinterpret_cast<float2*>(sp + (((size_t)(g0 + g) * C::NB + band) * C::COUT + c) * 2);
                *dst = make_float2(sm, sq);
            }
        }
    }
    if constexpr (C::STAMP) { if (tid == 0) stp[4] = __builtin_amdgcn_s_memtime(); }
}

template <class C>
inline hipError_t conv_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<C>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}

template <class C>
inline void conv_launch(const ConvArgs& a, hipStream_t st) {
    if (a.n <= 0) return;
    dim3 grid(((a.n + C::G - 1) / C::G) * C::NB, kTowers);
    hipLaunchKernelGGL(conv3x3_mfma_kernel<C>, grid, dim3(C::NTH), C::LDS_BYTES, st, a);
}

}  // namespace scs
