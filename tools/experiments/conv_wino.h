// conv1 -> conv2 -> pool1 at inference with conv2 in Winograd form F(2x2, 3x3) on the fp32 matrix cores, gfx950.
//
// The 28x28 conv2 outputs of a patch are 14x14 tiles of 2x2 - exactly the windows of pool1.  For every tile and input channel the
// 4x4 input window d is transformed (V = B'dB, 32 additions per lane, in registers), the 16 Winograd positions are 16 independent
// small GEMMs over the input channels (M = tiles, N = 20 output channels = one 16-wide + one 4-wide MFMA block, K = 20), and the
// output transform Y = A'MA, the 2x2 max, bias and ReLU happen in the lanes that hold the accumulators: 16 multiplies per output
// tile and channel pair instead of 36 (2.25x fewer matrix-core cycles than the direct form in conv_mfma.h).
//   A operand: lane (tile lr, channel 4c + kq) reads its 4x4 window of the conv1 plane in LDS (eight 8-byte reads) once per
//              k-step and feeds the 16 positions from registers;
//   B operand: U = G g G' of the BN-folded, flipped filters, packed at set-params time in the order the lanes read it (wino_pack_index),
//              resident in LDS (25.6 KB): four 16-byte reads per lane and k-step.
// Same decomposition as the direct kernel otherwise: one workgroup = one band (half) of a patch, conv1 (+BN+ReLU) runs first and
// its 16 output rows stay in LDS (conv1_to_lds), two workgroups per CU.
// Replaces Conv2DLayer(conv2) -> BatchNormLayer -> rectify -> MaxPool2DLayer of /root/reference/cnn_cort/nets.py:172-174.
#pragma once
#include "conv_mfma.h"

namespace scs {

constexpr int kWinoBlk = 640;                          // per (4 input channels, half): 2 x [64 lanes][4 positions] main + 2 x [16][4] leftover
constexpr int kWinoUFloats = 5 * 2 * kWinoBlk;         // 6400

// element e of the packed U array -> (position p = 4i + j of the Winograd domain, input channel, output channel).
// Block (c4, half h) holds, for the 8 positions of the half - quad 0 = domain row 0 (h = 0) / 3 (h = 1), quad 1 = row 1 / 2 - the B
// operands exactly as the lanes read them: main [quad][lane = 16 kq + lr][j] (one 16-byte read per lane and quad: W[cin 4 c4 + kq][cout lr]),
// leftover [quad][4 kq + i][j] (cout 16 + i).
struct WinoIdx { int p, cin, n; };
__host__ __device__ constexpr WinoIdx wino_pack_index(int e) {
    const int blk = e / kWinoBlk, r = e - blk * kWinoBlk, c4 = blk / 2, h = blk - 2 * c4;
    const bool main = r < 512;
    const int r2 = main ? r : r - 512, quad = main ? r2 / 256 : r2 / 64, s = main ? (r2 % 256) / 4 : (r2 % 64) / 4, j = r2 % 4;
    const int i = h ? (quad ? 2 : 3) : (quad ? 1 : 0);
    return main ? WinoIdx{4 * i + j, 4 * c4 + s / 16, s % 16} : WinoIdx{4 * i + j, 4 * c4 + s / 4, 16 + s % 4};
}
// U[i][j] = (G g G')[i][j] of a 3x3 filter g (row-major), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__host__ __device__ inline float wino_u(const float g[9], int i, int j) {
    float t[3];                                        // row i of G g
    for (int b = 0; b < 3; ++b)
        t[b] = i == 0 ? g[b] : (i == 3 ? g[6 + b] : 0.5f * ((g[b] + g[6 + b]) + (i == 1 ? g[3 + b] : -g[3 + b])));
    return j == 0 ? t[0] : (j == 3 ? t[2] : 0.5f * ((t[0] + t[2]) + (j == 1 ? t[1] : -t[1])));
}

// ---- second form: 8 waves per workgroup, ONE M tile per wave, the 16 positions in two halves of 8 (rows {0, 1} / {3, 2} of the
// Winograd domain) so that 64 accumulator registers are live instead of 128 and four waves share a SIMD (two workgroups per CU).
// Half h needs rows h .. h + 2 of the 4x4 window:  with t = (row) B:  V_a = t_h - t_(h+2)  (domain row 0 / 3),
// V_b = t_(h+1) + t_(h+2)  (h = 0: domain row 1)  or  t_(h+1) - t_h  (h = 1: domain row 2);  its share of Y = A'MA is
//   h = 0:  s0 = M0 + M1, s1 = M1;   h = 1:  s0 = M2, s1 = -M2 - M3       (s_r[j], then Y[r][0] = s0+s1+s2, Y[r][1] = s1-s2-s3 over j)
// Bands: tile rows 0..7 (112 tiles = 7 full M tiles) and 8..13 (84 tiles = 5 full M tiles + 4 tiles): 13 M tiles per patch.
// The fp32 MFMAs run on the vector ALUs' multipliers (tools/mfma_rate.hip: every vector instruction beside them costs ~2 cycles of
// the same pipe), so the kernel counts vector instructions (selects kept out of the k-loop: the two halves are two loops) and LDS
// instructions (B operands as 16-byte reads in lane order, window rows as plain 8-byte reads).
template <int SRC_, int DBG_ = 0>
struct Wino12Cfg8 {
    static constexpr int DBG = DBG_, SRC = SRC_, NB = 2, NW = 8, NTH = 512;
    static constexpr int TR0 = 8, TR1 = 6;                                  // tile rows of band 0 / 1
    static constexpr int PLANE = 544;                                       // 18 conv1 rows x 30, stride = 32 (mod 64): the k-subgroups of a 32-lane read group on disjoint banks
    static constexpr int XROWS = 2 * TR0 + 4;                               // raw rows of band 0 (band 1: 2 TR1 + 4)
    static constexpr int IN_FLOATS = 20 * PLANE, XS_ROWS_FLOATS = XROWS * 32, XS_FLOATS = XS_ROWS_FLOATS + 4 + kW1Floats;
    static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS + kWinoUFloats + XS_FLOATS) * 4;
    static_assert(LDS_BYTES * 2 <= 160 * 1024, "LDS budget");
    static_assert((TR0 * 14 + 15) / 16 <= NW && TR0 + TR1 == 14, "one M tile per wave");
    static_assert(SRC_ == SRC_CONV1 || SRC_ == SRC_VOLUME || SRC_ == SRC_X0, "the fused conv1 sources");
};

template <class C>
__global__ __launch_bounds__(C::NTH, 4) void conv12_wino8_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_u = smem + C::IN_FLOATS;
    float* s_x = s_u + kWinoUFloats;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int tower = blockIdx.y + a.tower0;
    const int band = (int)(blockIdx.x & 1), g0 = (int)(blockIdx.x >> 1);
    const float* __restrict__ insrc = pick3(a.in, tower);
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 0] = __builtin_amdgcn_s_memtime(); }
#ifdef SCS_WINO_STAGGER
    if (blockIdx.y == 0 && blockIdx.x < 512u) {                            // experiment: the first generation starts out of phase
        const int k = (int)((blockIdx.x * 2654435761u) >> 29) * SCS_WINO_STAGGER;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const int xrow0 = band ? 2 * C::TR0 : 0;                               // first raw row of the band: 0 / 16
    const int xrows = band ? 2 * C::TR1 + 4 : 2 * C::TR0 + 4;              // 20 / 16
    {
        const float4* __restrict__ usrc = reinterpret_cast<const float4*>(pick3(a.w, tower));
        constexpr int U4 = kWinoUFloats / 4, UIT = (U4 + C::NTH - 1) / C::NTH;
        float4 ur[UIT];
#pragma unroll
        for (int k = 0; k < UIT; ++k) ur[k] = usrc[min(tid + C::NTH * k, U4 - 1)];
        const float* __restrict__ w1g = pick3(a.w1, tower);
        const float w1a = w1g[min(tid, kW1Floats - 1)];
        if constexpr (C::SRC == SRC_CONV1) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(insrc + (size_t)g0 * 1024 + xrow0 * 32);
            if (tid < xrows * 8) reinterpret_cast<float4*>(s_x)[tid] = src[tid];
        } else if constexpr (C::SRC == SRC_X0) {
            if (band) load_x0_rows<2 * C::TR1 + 4, C::NTH>(a.x0, tower, g0, xrow0, s_x, tid);
            else load_x0_rows<2 * C::TR0 + 4, C::NTH>(a.x0, tower, g0, xrow0, s_x, tid);
        } else {
            const int cx = a.centers[(size_t)g0 * 3 + 0], cy = a.centers[(size_t)g0 * 3 + 1], cz = a.centers[(size_t)g0 * 3 + 2];
            for (int e = tid; e < xrows * 32; e += C::NTH)
                s_x[e] = gather_voxel(insrc, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, xrow0 + (e >> 5), e & 31);
        }
        float* s_w1 = s_x + C::XS_ROWS_FLOATS + 4;
        if (tid < kW1Floats) s_w1[tid] = w1a;
#pragma unroll
        for (int k = 0; k < UIT; ++k) { const int e = tid + C::NTH * k; if (e < U4) reinterpret_cast<float4*>(s_u)[e] = ur[k]; }
        __syncthreads();
        if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 1] = __builtin_amdgcn_s_memtime(); }
        if constexpr (C::DBG != 2) {
            if (band) conv1_to_lds<C::PLANE, 2 * C::TR1 + 2, C::NW>(s_x, s_w1, s_in, wave, lane);
            else conv1_to_lds<C::PLANE, 2 * C::TR0 + 2, C::NW>(s_x, s_w1, s_in, wave, lane);
        }
    }
    const float bias_m = pick3(a.bias, tower)[lr], bias_l = pick3(a.bias, tower)[16 + kq];
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 2] = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 3] = __builtin_amdgcn_s_memtime(); }

    const int tiles = band ? C::TR1 * 14 : C::TR0 * 14;
    const int t = (wave + g0) & (C::NW - 1);                               // the idle slots rotate over the waves (SIMDs)
    float o16[4] = {0.f, 0.f, 0.f, 0.f}, o4 = 0.f;                         // pooled outputs: tiles 16 t + 4 kq + r of cout lr; tile 16 t + lr of cout 16 + kq
    if (t * 16 < tiles) {
        const int T = min(t * 16 + lr, tiles - 1), ty = T / 14, tx = T - ty * 14;
        const float* __restrict__ wa = s_in + kq * C::PLANE + (2 * ty) * 30 + 2 * tx;
        const float4* __restrict__ sb16 = reinterpret_cast<const float4*>(s_u) + lane;                       // + quad * 64 (+ block * 160)
        const float4* __restrict__ sb4 = reinterpret_cast<const float4*>(s_u + 512) + kq * 4 + (lane & 3);    // + quad * 16
        const int xor32_addr = (lane ^ 32) << 2;
        f32x4 am[8], al[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { am[p] = f32x4{0.f, 0.f, 0.f, 0.f}; al[p] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // one half's share of Y from the live accumulators (am / al [0..3] = domain row a = 0 / 3, [4..7] = row b = 1 / 2)
        auto y_main = [&](auto hh, int r, float (&y)[4]) {
            constexpr int h = decltype(hh)::value;
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ma = am[j][r], mb = am[4 + j][r];
                s0[j] = h ? mb : ma + mb; s1[j] = h ? -mb - ma : mb;
            }
            y[0] = (s0[0] + s0[1]) + s0[2]; y[1] = (s0[1] - s0[2]) - s0[3];
            y[2] = (s1[0] + s1[1]) + s1[2]; y[3] = (s1[1] - s1[2]) - s1[3];
        };
        auto y_left = [&](auto hh, f32x4 (&y)[4]) {                         // y[2x2 position][cout 16 + i], still per k-subgroup
            constexpr int h = decltype(hh)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s0[4], s1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ma = al[j][i], mb = al[4 + j][i];
                    s0[j] = h ? mb : ma + mb; s1[j] = h ? -mb - ma : mb;
                }
                y[0][i] = (s0[0] + s0[1]) + s0[2]; y[1][i] = (s0[1] - s0[2]) - s0[3];
                y[2][i] = (s1[0] + s1[1]) + s1[2]; y[3][i] = (s1[1] - s1[2]) - s1[3];
            }
        };
        // Five k-steps per half, software-pipelined: the three window rows and the B operands of the NEXT step (the other half's
        // first step behind half 0's last) are requested while this step's MFMAs run - each quad of B registers is refilled right
        // behind the MFMAs that consumed it.  (Rolled loops with scheduling fences: left alone, the compiler chains the k-steps of
        // a position back to back or waits for every B operand in front of its MFMA.)
        float2 w[6];
        auto window = [&](const float* __restrict__ p) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { w[2 * r] = *reinterpret_cast<const float2*>(p + r * 30); w[2 * r + 1] = *reinterpret_cast<const float2*>(p + r * 30 + 2); }
        };
        float4 bma, bmb, bla, blb;
        window(wa);
        bma = sb16[0]; bmb = sb16[64]; bla = sb4[0]; blb = sb4[16];
        auto half = [&](auto hh) {
            constexpr int h = decltype(hh)::value;
            constexpr int STEPS = C::DBG == 1 ? 1 : 5;
#pragma unroll 1
            for (int c = 0; c < STEPS; ++c) {
                float v[8];
                {
                    float tr[3][4];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float d0 = w[2 * r].x, d1 = w[2 * r].y, d2 = w[2 * r + 1].x, d3 = w[2 * r + 1].y;
                        tr[r][0] = d0 - d2; tr[r][1] = d1 + d2; tr[r][2] = d2 - d1; tr[r][3] = d1 - d3;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = tr[0][j] - tr[2][j]; v[4 + j] = h ? tr[1][j] - tr[0][j] : tr[1][j] + tr[2][j]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                // next step: (h, c + 1), or (1, 0) behind half 0's last step; half 1's last step re-reads its own operands
                const bool wrap = h == 0 && c == STEPS - 1;
                const int cn = wrap ? 0 : min(c + 1, STEPS - 1);
                const int hn = (h == 1 || wrap) ? 1 : 0;
                if constexpr (C::DBG != 5 && C::DBG != 6) window(wa + cn * 4 * C::PLANE + hn * 30);
                __builtin_amdgcn_sched_barrier(0);
                const float4* __restrict__ nb16 = sb16 + (cn * 2 + hn) * (kWinoBlk / 4);
                const float4* __restrict__ nb4 = sb4 + (cn * 2 + hn) * (kWinoBlk / 4);
                am[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], bma.x, am[0], 0, 0, 0);
                al[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(bla.x, v[0], al[0], 0, 0, 0);
                am[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], bma.y, am[1], 0, 0, 0);
                al[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bla.y, v[1], al[1], 0, 0, 0);
                am[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], bma.z, am[2], 0, 0, 0);
                al[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(bla.z, v[2], al[2], 0, 0, 0);
                am[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[3], bma.w, am[3], 0, 0, 0);
                al[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(bla.w, v[3], al[3], 0, 0, 0);
                if constexpr (C::DBG != 4 && C::DBG != 6) { bma = nb16[0]; bla = nb4[0]; }
                __builtin_amdgcn_sched_barrier(0);
                am[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4], bmb.x, am[4], 0, 0, 0);
                al[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(blb.x, v[4], al[4], 0, 0, 0);
                am[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[5], bmb.y, am[5], 0, 0, 0);
                al[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(blb.y, v[5], al[5], 0, 0, 0);
                am[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[6], bmb.z, am[6], 0, 0, 0);
                al[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(blb.z, v[6], al[6], 0, 0, 0);
                am[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[7], bmb.w, am[7], 0, 0, 0);
                al[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(blb.w, v[7], al[7], 0, 0, 0);
                if constexpr (C::DBG != 4 && C::DBG != 6) { bmb = nb16[64]; blb = nb4[16]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        half(std::integral_constant<int, 0>{});
        float yp[4][4];                                                     // half 0's share: 16-wide block [register r][2x2]
        float yq[4];                                                        // 4-wide block [2x2] of cout 16 + kq, tile 16 t + lr
#pragma unroll
        for (int r = 0; r < 4; ++r) y_main(std::integral_constant<int, 0>{}, r, yp[r]);
        {
            f32x4 y[4];
            y_left(std::integral_constant<int, 0>{}, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) yq[k] = ksub_reduce(y[k], kq, xor32_addr);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) { am[p] = f32x4{0.f, 0.f, 0.f, 0.f}; al[p] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        half(std::integral_constant<int, 1>{});
        // ---- half 1's share, 2x2 max (= pool1), bias, ReLU
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y[4];
            y_main(std::integral_constant<int, 1>{}, r, y);
            o16[r] = fmaxf(fmaxf(fmaxf(yp[r][0] + y[0], yp[r][1] + y[1]), fmaxf(yp[r][2] + y[2], yp[r][3] + y[3])) + bias_m, 0.f);
        }
        {
            f32x4 y[4];
            y_left(std::integral_constant<int, 1>{}, y);
            float q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = yq[k] + ksub_reduce(y[k], kq, xor32_addr);
            o4 = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])) + bias_l, 0.f);
        }
    }
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 4] = __builtin_amdgcn_s_memtime(); }
    // ---- the pooled rows leave as whole rows: restaged in LDS over the conv1 planes (every wave has finished reading them)
    __syncthreads();
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 5] = __builtin_amdgcn_s_memtime(); }
    constexpr int OSTR = C::TR0 * 14 + 4;                                   // 116: rows 16-byte aligned
    float* s_o = s_in;
    if (t * 16 + kq * 4 < tiles) *reinterpret_cast<float4*>(s_o + lr * OSTR + t * 16 + kq * 4) = float4{o16[0], o16[1], o16[2], o16[3]};   // tiles % 4 == 0: whole quads
    if (t * 16 + lr < tiles) s_o[(16 + kq) * OSTR + t * 16 + lr] = o4;
    __syncthreads();
    {
        float* __restrict__ outp = pick3(a.out, tower) + (size_t)g0 * 20 * 196 + band * (C::TR0 * 14);
        const int q4 = tiles >> 2;                                          // float4 per row: 28 / 21
        for (int e = tid; e < 20 * q4; e += C::NTH) {
            const int co = e / q4, r = e - co * q4;
            if (C::DBG != 3 || o4 == 12345.f)
                *reinterpret_cast<float4*>(outp + (size_t)co * 196 + 4 * r) = *reinterpret_cast<const float4*>(s_o + co * OSTR + 4 * r);
        }
    }
    if constexpr (C::DBG == 7) { if (lane == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + 6] = __builtin_amdgcn_s_memtime(); }
}

template <class C>
inline hipError_t wino12x8_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_wino8_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}
template <class C>
inline void wino12x8_launch(const ConvArgs& a, hipStream_t st, int ntowers = kTowers) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(conv12_wino8_kernel<C>, dim3(a.n * C::NB, ntowers), dim3(C::NTH), C::LDS_BYTES, st, a);
}

}  // namespace scs
