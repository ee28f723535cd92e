// conv3 -> conv4 + pool2 -> conv5 of ONE patch-tower as one inference kernel: the 12x12x40 and 5x5x40 activations between the
// layers never leave LDS.  Replaces the three launches CfgConv3 / CfgConv4 / CfgConv5 of scs_api.hip
// (Conv2DLayer -> BatchNorm (folded) -> rectify chains and MaxPool2DLayer of /root/reference/cnn_cort/nets.py:174-177).
//
// Same arithmetic as conv_mfma.h, output for output: one k-step = 4 input channels at one tap, chunks 0 .. CIN/4 - 1, taps 0 .. 8,
// 16-wide v_mfma_f32_16x16x4_f32 blocks + 4-wide v_mfma_f32_4x4x1_16b_f32 blocks, bias + ReLU (after the in-lane pooling
// maximum where a layer pools) - the results are bit-identical to the three-kernel path (tests/test_gpu_forward.py).
//
// Why one workgroup = one patch-tower works here: 9 M tiles of conv3 on 4 waves are 2 + 2 + 2 + 2 and one tile whose channel
// blocks are dealt to the waves (16-wide block b to wave b, the 4-wide blocks to the next wave: 94 % of the matrix pipe against
// 75 % as 3,2,2,2); conv5's single tile (9 of 16 rows) is dealt the same way, three waves one 16-wide block each and the fourth
// the three 4-wide blocks.  LDS: the conv4 input planes [40][144] + one double-buffered weight chunk of the widest layer =
// 40.3 KB - four workgroups (16 waves) per CU; the conv3 input planes and the conv5 input planes live inside the conv4 planes'
// space before / after those are in use.  HBM traffic per patch-tower: 15.7 KB in, 2.2 KB out (three kernels: 15.7 + 2 x 23 +
// 2 x 4 + 2.2 KB).
#pragma once
#include "conv_mfma.h"

namespace scs {

// geometry of one layer inside the fused kernel (input planes [CIN][PLANE] in LDS, rows of WIN = HIN floats)
template <int CIN_, int COUT_, int HIN_, int PLANE_, bool POOL_>
struct FusedLayer {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, WIN = HIN_, PLANE = PLANE_, HO = HIN_ - 2, HP = HO / 2, PIX = HO * HO;
    static constexpr bool POOL = POOL_;
    static constexpr int WINDOWS = HP * HP;
    static constexpr int MROWS = POOL_ ? WINDOWS * 4 : PIX;       // rows of the GEMM's M (pooling: 4 positions per window)
    static constexpr int TILES = (MROWS + 15) / 16;
    static constexpr bool SHARED = TILES % 4 == 1;                // the last tile's channel blocks are dealt to the waves
    static constexpr int TPW = (TILES + 3) / 4;
    static constexpr int NT16 = nt16(COUT_), NT4 = nt4(COUT_), NS16 = ns16(COUT_), L4 = 4 * NT4, NC4 = CIN_ / 4;
    static constexpr int WCHUNK = conv_chunk_floats(COUT_), WCHUNK4 = WCHUNK / 4, WPT = (WCHUNK4 + 255) / 256;
    static constexpr int OUTPIX = POOL_ ? WINDOWS : PIX;          // stored outputs per channel
    static_assert(!SHARED || NT16 + (NT4 > 0 ? 1 : 0) <= 4, "the odd tile's channel blocks go to distinct waves");
    static_assert(WPT <= 3, "a weight chunk is at most three 16-byte pieces per thread");
};

using F3 = FusedLayer<20, 40, 14, 208, false>;     // conv3: 14x14 -> 12x12
using F4 = FusedLayer<40, 40, 12, 144, true>;      // conv4 + pool2: 12x12 -> 10x10 -> 5x5
using F5 = FusedLayer<40, 60, 5, 27, false>;       // conv5: 5x5 -> 3x3
constexpr int kF3In = F3::CIN * F3::PLANE, kF4In = F4::CIN * F4::PLANE, kF5In = F5::CIN * F5::PLANE;
constexpr int kFusedSlot = F5::WCHUNK > F4::WCHUNK ? F5::WCHUNK : F4::WCHUNK;              // one slot holds the widest chunk
constexpr int kFusedW = 2 * kFusedSlot;
constexpr int kF5Out = 60 * 12;                                                          // conv5 outputs restaged [60][12]
static_assert(kF3In <= kF4In && kF5In + kF5Out <= kF4In, "the conv3 / conv5 planes live inside the conv4 planes' space");
static_assert(F3::WCHUNK <= kFusedSlot, "weight buffer");
constexpr size_t kFusedLdsBytes = (size_t)(kF4In + kFusedW) * 4;
static_assert(kFusedLdsBytes * 4 <= 160 * 1024, "four workgroups per CU");

struct FusedArgs {
    const float* in[kTowers];       // pool1 output [n][20][14][14]
    const float* w[kTowers][3];     // packed chunks of conv3, conv4, conv5 (BN folded, scs_common.h)
    const float* bias[kTowers][3];
    float* out[kTowers];            // conv5 output [n][540]
    float* dbg3[kTowers];           // optional: conv3 output [n][40][144] and pool2 output [n][40][25] (scs_get_activation), or null
    float* dbg4[kTowers];
    int n;
};

// One layer: s_in (planes in LDS, complete and visible) -> s_out [COUT][OSTR] in LDS (bias + ReLU, pooled where the layer pools).
// s_out may overlap s_in / s_w: it is written behind the main loop's last barrier.  Ends with a barrier: s_out is visible.
// The three layers' weight chunks (5 + 10 + 10) form ONE stream through the double buffer (slots of kFusedSlot floats, slot =
// running chunk number & 1): the chunk behind a layer's last one is the next layer's first (NEXT), so no layer start waits for
// a weight fetch.  On entry this layer's chunk 0 is in its slot and visible.
struct NoLayer { static constexpr int WCHUNK4 = 1, WPT = 1; };
template <class L, int OSTR, int GI0, class NEXT>
__device__ __forceinline__ void fused_layer(const float* __restrict__ s_in, float* __restrict__ s_w, const float* __restrict__ wsrc,
                                            const float* __restrict__ bias, float* __restrict__ s_out, const float* __restrict__ wnext,
                                            int wt, int tid, int lane) {
    constexpr bool HAS_NEXT = !std::is_same<NEXT, NoLayer>::value;
    const int lr = lane & 15, kq = lane >> 4;
    const int wi0 = min(tid, L::WCHUNK4 - 1), wi1 = min(tid + 256, L::WCHUNK4 - 1), wi2 = min(tid + 512, L::WCHUNK4 - 1);
    const int ni0 = min(tid, NEXT::WCHUNK4 - 1), ni1 = min(tid + 256, NEXT::WCHUNK4 - 1), ni2 = min(tid + 512, NEXT::WCHUNK4 - 1);
    float4 wr0 = {0.f, 0.f, 0.f, 0.f}, wr1 = wr0, wr2 = wr0;
    float bvs[L::NT16], bv4[L::NT4 > 0 ? L::NT4 : 1];
#pragma unroll
    for (int nt = 0; nt < L::NT16; ++nt) bvs[nt] = bias[nt * 16 + lr];
#pragma unroll
    for (int u = 0; u < L::NT4; ++u) bv4[u] = bias[L::NT16 * 16 + u * 4 + kq];

    // ---- per-lane A base offsets of the wave's tiles (the last one of a SHARED layer is the odd tile)
    int abase[L::TPW];
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
        int m = ((L::SHARED && j == L::TPW - 1) ? L::TILES - 1 : wt + 4 * j) * 16 + lr;
        int y, x;
        if constexpr (L::POOL) {
            const int widx = min(m >> 2, L::WINDOWS - 1), q = m & 3;
            const int wy = widx / L::HP, wx = widx - wy * L::HP;
            y = 2 * wy + (q >> 1); x = 2 * wx + (q & 1);
        } else {
            m = min(m, L::PIX - 1);
            y = m / L::HO; x = m - y * L::HO;
        }
        abase[j] = kq * L::PLANE + y * L::WIN + x;
    }
    f32x4 acc[L::TPW][L::NT16], acc4[L::TPW][L::NT4 > 0 ? L::NT4 : 1];
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
#pragma unroll
        for (int nt = 0; nt < L::NT16; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < L::NT4; ++u) acc4[j][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int boff = kq * L::NS16 + lr;                        // main part: W[k = 4s + kq][cout = 16 nt + lr]
    const int loff = 36 * L::NS16 + kq * L::L4 + (lane & 3);    // leftover: W[k = 4s + kq][cout = 16 NT16 + 4u + (lane & 3)]
    const bool last_ok = L::SHARED || (wt + 4 * (L::TPW - 1)) < L::TILES;
#ifndef SCS_FUSED_UNROLL
#define SCS_FUSED_UNROLL 1
#endif
#pragma unroll SCS_FUSED_UNROLL
    for (int c4 = 0; c4 < L::NC4; ++c4) {
        if (c4 + 1 < L::NC4) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wsrc + (size_t)(c4 + 1) * L::WCHUNK);
            wr0 = src[wi0];
            if constexpr (L::WPT > 1) wr1 = src[wi1];
            if constexpr (L::WPT > 2) wr2 = src[wi2];
        } else if constexpr (HAS_NEXT) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wnext);
            wr0 = src[ni0];
            if constexpr (NEXT::WPT > 1) wr1 = src[ni1];
            if constexpr (NEXT::WPT > 2) wr2 = src[ni2];
        }
        const float* __restrict__ sw = s_w + ((GI0 + c4) & 1) * kFusedSlot;
        const float* __restrict__ sa = s_in + c4 * 4 * L::PLANE;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * L::WIN + (tap % 3);
            float b[L::NT16], b4[L::NT4 > 0 ? L::NT4 : 1];
#pragma unroll
            for (int nt = 0; nt < L::NT16; ++nt) b[nt] = sw[boff + tap * 4 * L::NS16 + nt * 16];
#pragma unroll
            for (int u = 0; u < L::NT4; ++u) b4[u] = sw[loff + tap * 4 * L::L4 + u * 4];
#pragma unroll
            for (int j = 0; j < L::TPW; ++j) {
                if (L::SHARED && j == L::TPW - 1) {             // the odd tile: this wave's share of its channel blocks
                    if (wt <= L::NT16) {
                        const float av = sa[abase[j] + toff];
#pragma unroll
                        for (int nt = 0; nt < L::NT16; ++nt)
                            if (wt == nt) acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[j][nt], 0, 0, 0);
                        if (wt == L::NT16) {
#pragma unroll
                            for (int u = 0; u < L::NT4; ++u) acc4[j][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, acc4[j][u], 0, 0, 0);
                        }
                    }
                    continue;
                }
                if (j == L::TPW - 1 && !last_ok) continue;      // wave-uniform: this wave has no such tile
                const float av = sa[abase[j] + toff];
#pragma unroll
                for (int nt = 0; nt < L::NT16; ++nt) acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[j][nt], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < L::NT4; ++u) acc4[j][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, acc4[j][u], 0, 0, 0);
            }
        }
        float4* dst = reinterpret_cast<float4*>(s_w + ((GI0 + c4 + 1) & 1) * kFusedSlot);
        if (c4 + 1 < L::NC4) {
            if (tid < L::WCHUNK4) dst[tid] = wr0;
            if constexpr (L::WPT > 1) { if (tid + 256 < L::WCHUNK4) dst[tid + 256] = wr1; }
            if constexpr (L::WPT > 2) { if (tid + 512 < L::WCHUNK4) dst[tid + 512] = wr2; }
        } else if constexpr (HAS_NEXT) {
            if (tid < NEXT::WCHUNK4) dst[tid] = wr0;
            if constexpr (NEXT::WPT > 1) { if (tid + 256 < NEXT::WCHUNK4) dst[tid + 256] = wr1; }
            if constexpr (NEXT::WPT > 2) { if (tid + 512 < NEXT::WCHUNK4) dst[tid + 512] = wr2; }
        }
        __syncthreads();
    }

    // ---- epilogue: D row = 4 kq + r (pixel / window position), column = lr (cout); bias + ReLU (+ pooling maximum first) -> s_out
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
        const bool odd = L::SHARED && j == L::TPW - 1;
        const int t = odd ? L::TILES - 1 : wt + 4 * j;
        if (t >= L::TILES) continue;
#pragma unroll
        for (int nt = 0; nt < L::NT16; ++nt) {
            if (odd && wt != nt) continue;
            const int co = nt * 16 + lr;
            if constexpr (L::POOL) {
                const int widx = t * 4 + kq;
                if (widx < L::WINDOWS) {
                    const f32x4 v = acc[j][nt];
                    s_out[co * OSTR + widx] = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) + bvs[nt], 0.f);
                }
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int m = t * 16 + kq * 4 + r4;
                    if (m < L::PIX) s_out[co * OSTR + m] = fmaxf(acc[j][nt][r4] + bvs[nt], 0.f);
                }
            }
        }
    }
    const int xor32_addr = (lane ^ 32) << 2;
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
        const bool odd = L::SHARED && j == L::TPW - 1;
        const int t = odd ? L::TILES - 1 : wt + 4 * j;
#pragma unroll
        for (int u = 0; u < L::NT4; ++u) {
            float val = ksub_reduce(acc4[j][u], kq, xor32_addr);                   // this lane holds cout 16 NT16 + 4u + kq
            if constexpr (L::POOL) val = max_quad(val);
            if (t >= L::TILES || (odd && wt != L::NT16)) continue;
            const int co = L::NT16 * 16 + u * 4 + kq;
            if constexpr (L::POOL) {
                const int widx = t * 4 + ((lane & 15) >> 2);
                if ((lane & 3) == 0 && widx < L::WINDOWS) s_out[co * OSTR + widx] = fmaxf(val + bv4[u], 0.f);
            } else {
                const int m = t * 16 + (lane & 15);
                if (m < L::PIX) s_out[co * OSTR + m] = fmaxf(val + bv4[u], 0.f);
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256, 4) void conv345_fused_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* s_p = smem_f;                    // conv3 input [20][208], then conv4 input [40][144], then conv5 input [40][27] + its outputs
    float* s_w = smem_f + kF4In;            // double-buffered weight chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tower = blockIdx.y, g = blockIdx.x;
    const int wt = (wave + (int)blockIdx.x) & 3;                // tile ownership rotates with the workgroup (conv_mfma.h)
    const float* w3 = tower == 0 ? a.w[0][0] : (tower == 1 ? a.w[1][0] : a.w[2][0]);
    const float* w4 = tower == 0 ? a.w[0][1] : (tower == 1 ? a.w[1][1] : a.w[2][1]);
    const float* w5 = tower == 0 ? a.w[0][2] : (tower == 1 ? a.w[1][2] : a.w[2][2]);
    const float* b3 = tower == 0 ? a.bias[0][0] : (tower == 1 ? a.bias[1][0] : a.bias[2][0]);
    const float* b4 = tower == 0 ? a.bias[0][1] : (tower == 1 ? a.bias[1][1] : a.bias[2][1]);
    const float* b5 = tower == 0 ? a.bias[0][2] : (tower == 1 ? a.bias[1][2] : a.bias[2][2]);
    // ---- conv3's input planes and its first weight chunk: one global round trip
    {
        constexpr int HW4 = 196 / 4;
        const float4* __restrict__ src4 = reinterpret_cast<const float4*>(pick3(a.in, tower) + (size_t)g * F3::CIN * 196);
        float4* dst4 = reinterpret_cast<float4*>(s_p);
        const float4 c0 = reinterpret_cast<const float4*>(w3)[min(tid, F3::WCHUNK4 - 1)];
        const float4 c1 = reinterpret_cast<const float4*>(w3)[min(tid + 256, F3::WCHUNK4 - 1)];
        static_assert(F3::WPT == 2, "conv3's chunk: two pieces per thread");
        staged_copy<F3::CIN * HW4, float4, 256>(
            tid, [&](int e) { return src4[e]; },
            [&](int e, float4 v) { const int pl = e / HW4, r = e - pl * HW4; dst4[pl * (F3::PLANE / 4) + r] = v; });
        if (tid < F3::WCHUNK4) reinterpret_cast<float4*>(s_w)[tid] = c0;
        if (tid + 256 < F3::WCHUNK4) reinterpret_cast<float4*>(s_w)[tid + 256] = c1;
    }
    __syncthreads();
    fused_layer<F3, F4::PLANE, 0, F4>(s_p, s_w, w3, b3, s_p, w4, wt, tid, lane);       // -> conv4's input planes, in place
    if (float* d3 = pick3(a.dbg3, tower)) {
        for (int e = tid; e < 40 * 144; e += 256) d3[(size_t)g * 5760 + e] = s_p[e];
    }
    fused_layer<F4, F5::PLANE, F3::NC4, F5>(s_p, s_w, w4, b4, s_p, w5, wt, tid, lane); // -> conv5's input planes
    if (float* d4 = pick3(a.dbg4, tower)) {
        for (int e = tid; e < 1000; e += 256) d4[(size_t)g * 1000 + e] = s_p[(e / 25) * F5::PLANE + e % 25];
    }
    fused_layer<F5, 12, F3::NC4 + F4::NC4, NoLayer>(s_p, s_w, w5, b5, s_p + kF5In, nullptr, wt, tid, lane);   // -> [60][12] behind them
    float* __restrict__ outp = pick3(a.out, tower) + (size_t)g * 540;
    for (int e = tid; e < 540; e += 256) outp[e] = s_p[kF5In + (e / 9) * 12 + e % 9];
}

inline hipError_t conv345_fused_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv345_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLdsBytes);
}
inline void conv345_fused_launch(const FusedArgs& a, hipStream_t st) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(conv345_fused_kernel, dim3(a.n, kTowers), dim3(256), kFusedLdsBytes, st, a);
}

}  // namespace scs
