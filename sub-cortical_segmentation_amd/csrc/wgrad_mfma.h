// Weight gradient of the 3x3 true convolutions on the fp32 matrix cores.
//
//   dW[o][c][2-dy][2-dx] = sum_{n,y,x} dz[n][o][y][x] * X[n][c][y+dy][x+dx]        (Theano autodiff of nets.py:171-177)
//
// GEMM view per patch:  D[(c,tap)][o] += sum_pixels  Xs[(c,tap)][pixel] * dz[pixel][o]  with
//   rows  = (input channel, tap) pairs (16 per MFMA tile; the row's LDS base = c*plane + tap offset),
//   cols  = output channels (16-wide v_mfma_f32_16x16x4_f32 blocks + 4-wide v_mfma_f32_4x4x1_16b_f32 blocks),
//   k     = output pixels, 4 per step: the output row is padded to a multiple of 4 with ZERO dz columns so a k-step
//           never straddles image rows and every operand address is base + immediate.
// A workgroup walks a slice of the batch, one patch per staging round, with its accumulators live, then writes one
// partial dW; partials are summed in a fixed order (deterministic gradients) by wgrad_final_kernel.
//
// Operand sources (the training step keeps neither X1 nor any d z pass of its own):
//   XSRC 0  X is read from global memory ([n][CIN][HIN][HIN]);
//   XSRC 1  X is the conv1 block output (conv2's weight gradient), recomputed per patch band from the 32x32 input rows with
//           the batch-statistics-folded conv1 operands (conv_mfma.h: conv1_to_lds);
//   d z is read from global memory ([n][COUT][HO][HO]): the data-gradient kernel of the layer stores it as a by-product of its
//   staging.  (Forming d z here instead, and staging several patches per round, were measured in round 2 and cost 3-15 us per
//   layer: profiles/r02_tbench_final.txt; those variants are no longer part of this header.)
#pragma once
#include <type_traits>

#include "conv_mfma.h"

namespace scs {

template <int CIN_, int COUT_, int HIN_, int NB_ = 1, int WPC_ = 1, int XSRC_ = 0, int PIPE_ = 0>
struct WgCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, HO = HIN_ - 2, XSRC = XSRC_;
    static constexpr bool PIPE = PIPE_ != 0;                 // one patch per round, the next patch's loads in flight during the matrix phase
    // a workgroup owns band `b` of NB bands of output rows (HOB rows of dz, HOB + 2 rows of X) of the patches of its
    // batch slice: smaller LDS footprint -> WPC workgroups share a CU and overlap staging with the matrix phase
    static constexpr int NB = NB_, WPC = WPC_, HOB = HO / NB_, HXB = HOB + 2;
    static_assert(HO % NB_ == 0, "bands must split the output rows evenly");
    static constexpr int WOP = (HO + 3) / 4 * 4;             // padded output row
    static constexpr int XPL = (HXB * HIN_ + 8 + 3) / 4 * 4; // X plane stride (tail absorbs the padded-column reads)
    static constexpr int ZPL = HOB * WOP;                    // dz plane stride
    static constexpr int J = CIN_ * 9;                       // (c,tap) rows
    static constexpr int MT = (J + 15) / 16;
    static constexpr int MTW = (MT + 3) / 4;                 // row tiles per wave
    static constexpr int NT16 = nt16(COUT_), NT4 = nt4(COUT_);
    static constexpr int X_FLOATS = (CIN_ * XPL + 3) / 4 * 4, Z_FLOATS = (COUT_ * ZPL + 3) / 4 * 4;   // per patch, 16-byte aligned pieces
    static constexpr int X0_ROWS = HXB + 2, X0_FLOATS = XSRC_ ? X0_ROWS * 32 + 4 : 0;                   // input rows per patch
    static constexpr int W1_FLOATS = XSRC_ ? (kW1Floats + 3) / 4 * 4 : 0;
    static constexpr size_t LDS_BYTES = (size_t)(X_FLOATS + Z_FLOATS + X0_FLOATS + W1_FLOATS) * 4;
    static_assert(LDS_BYTES * WPC_ <= 160 * 1024, "LDS budget");
    static_assert(!XSRC_ || (CIN_ == 20 && HIN_ == 30), "the recomputed X is the conv1 block output");
};

struct WgArgs {
    const float* X[kTowers];      // XSRC 0: [n][CIN][HIN][HIN]
    const float* w1[kTowers];     // XSRC 1: conv1 operands (kW1Floats, batch statistics folded)
    X0Src x0;                     // XSRC 1: the input patches
    const float* dZ[kTowers];     // [n][COUT][HO][HO]
    float* part;                  // [tower][slice][band][CIN*9][COUT]
    int n, slices;
};

template <class C>
__global__ __launch_bounds__(256, C::WPC) void wgrad_mfma_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_x = smem;                                   // [X_FLOATS]
    float* s_z = s_x + C::X_FLOATS;                      // [Z_FLOATS]
    float* s_x0 = s_z + C::Z_FLOATS;                     // [X0_FLOATS]
    float* s_w1 = s_x0 + C::X0_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int slice = blockIdx.x / C::NB, band = blockIdx.x % C::NB, tower = blockIdx.y;
    const int y0 = band * C::HOB;
    const int per = (a.n + a.slices - 1) / a.slices, n0 = slice * per, n1 = min(a.n, n0 + per);

    // zero once: dz padding columns and the X plane tails stay zero for the whole kernel
    for (int e = tid; e < C::X_FLOATS + C::Z_FLOATS; e += 256) smem[e] = 0.f;
    if constexpr (C::XSRC == 1) {
        const float* __restrict__ w1g = pick3(a.w1, tower);
        for (int e = tid; e < kW1Floats; e += 256) s_w1[e] = w1g[e];
    }

    // per-lane row bases of this wave's (c,tap) tiles; rows beyond J are clamped (computed, never stored)
    int xbase[C::MTW];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m) {
        const int j = min((wave + 4 * m) * 16 + lr, C::J - 1);
        const int c = j / 9, tap = j - c * 9;
        xbase[m] = c * C::XPL + (tap / 3) * C::HIN + tap % 3 + kq;
    }
    const int zoff16 = lr * C::ZPL + kq;                                   // dz[o = 16 nt + lr][pixel 4s + kq]
    const int zoff4 = (C::NT16 * 16 + (lane & 3)) * C::ZPL + kq;           // dz[o = 16 NT16 + 4u + (lane&3)][pixel 4s + kq]

    f32x4 acc[C::MTW][C::NT16 > 0 ? C::NT16 : 1], acc4[C::MTW][C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m) {
#pragma unroll
        for (int nt = 0; nt < C::NT16; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < C::NT4; ++u) acc4[m][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // the matrix phase of one staged patch (all of its band's output rows)
    auto mma_patch = [&](int) {
#pragma unroll 1
        for (int y = 0; y < C::HOB; ++y) {
            const float* __restrict__ sxr = s_x + y * C::HIN;
            const float* __restrict__ szr = s_z + y * C::WOP;
#pragma unroll
            for (int xg = 0; xg < C::WOP / 4; ++xg) {
                float b[C::NT16 > 0 ? C::NT16 : 1], b4[C::NT4 > 0 ? C::NT4 : 1];
#pragma unroll
                for (int nt = 0; nt < C::NT16; ++nt) b[nt] = szr[zoff16 + nt * 16 * C::ZPL + 4 * xg];
#pragma unroll
                for (int u = 0; u < C::NT4; ++u) b4[u] = szr[zoff4 + u * 4 * C::ZPL + 4 * xg];
#pragma unroll
                for (int m = 0; m < C::MTW; ++m) {
                    if ((3 + 4 * m) >= C::MT && (wave + 4 * m) >= C::MT) continue;   // wave-uniform: no such row tile
                    const float av = sxr[xbase[m] + 4 * xg];
#pragma unroll
                    for (int nt = 0; nt < C::NT16; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < C::NT4; ++u)
                        acc4[m][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, acc4[m][u], 0, 0, 0);
                }
            }
        }
    };

    if constexpr (C::PIPE) {
        // One patch per round, software-pipelined: the global loads of patch p + 1 are issued before the matrix phase of patch p
        // and land in LDS after it (a workgroup walks 4-8 patches; without this every one of them exposes a global round trip
        // that only the one or two other workgroups of the CU could hide).
        constexpr int HW = C::HIN * C::HIN, XB = C::HXB * C::HIN, PZ = C::HO * C::HO, ZB = C::HOB * C::HO;
        constexpr int XV = (HW % 4 == 0 && XB % 4 == 0 && (C::HOB * C::HIN) % 4 == 0) ? 4
                         : (HW % 2 == 0 && XB % 2 == 0 && (C::HOB * C::HIN) % 2 == 0) ? 2 : 1;
        constexpr int ZV = (C::WOP == C::HO && PZ % 4 == 0 && ZB % 4 == 0) ? 4 : 1;
        using VX = typename std::conditional<XV == 4, float4, typename std::conditional<XV == 2, float2, float>::type>::type;
        using VZ = typename std::conditional<ZV == 4, float4, float>::type;
        StagedRegs<C::XSRC == 1 ? 1 : C::CIN * XB / XV, VX> rx;
        StagedRegs<C::COUT * ZB / ZV, VZ> rz;
        X0RowRegs<C::X0_ROWS, 256> r0;
        auto request = [&](int nb) {
            if constexpr (C::XSRC == 1) {
                r0.load(a.x0, tower, nb, y0, tid);
            } else {
                const float* __restrict__ xs = pick3(a.X, tower) + (size_t)nb * C::CIN * HW + y0 * C::HIN;
                rx.load(tid, [&](int e) { const int pl = e / (XB / XV), r = e - pl * (XB / XV); return reinterpret_cast<const VX*>(xs + (size_t)pl * HW)[r]; });
            }
            const float* __restrict__ zs = pick3(a.dZ, tower) + (size_t)nb * C::COUT * PZ + y0 * C::HO;
            rz.load(tid, [&](int e) { const int pl = e / (ZB / ZV), r = e - pl * (ZB / ZV); return reinterpret_cast<const VZ*>(zs + (size_t)pl * PZ)[r]; });
        };
        auto commit = [&]() {
            if constexpr (C::XSRC == 1) {
                r0.store(s_x0, tid);
            } else {
                rx.store(tid, [&](int e, VX v) { const int c = e / (XB / XV), r = e - c * (XB / XV); reinterpret_cast<VX*>(s_x)[c * (C::XPL / XV) + r] = v; });
            }
            if constexpr (ZV == 4) {
                rz.store(tid, [&](int e, VZ v) { const int o = e / (ZB / 4), r = e - o * (ZB / 4); reinterpret_cast<VZ*>(s_z)[o * (C::ZPL / 4) + r] = v; });
            } else {
                rz.store(tid, [&](int e, VZ v) { const int o = e / ZB, r = e - o * ZB, y = r / C::HO, x = r - y * C::HO; s_z[o * C::ZPL + y * C::WOP + x] = v; });
            }
        };
        if (n0 < n1) request(n0);
        for (int nb = n0; nb < n1; ++nb) {
            __syncthreads();                                               // previous patch consumed (and the zero fill)
            commit();
            __syncthreads();
            if constexpr (C::XSRC == 1) {
                conv1_to_lds<C::XPL, C::HXB, 4>(s_x0, s_w1, s_x, wave, lane);
                __syncthreads();
            }
            if (nb + 1 < n1) request(nb + 1);
            mma_patch(0);
        }
    } else
    for (int nb = n0; nb < n1; ++nb) {
        __syncthreads();                                                   // previous patch consumed (and the zero fill)
        // ---- X
        if constexpr (C::XSRC == 1) {
            load_x0_rows<C::X0_ROWS, 256>(a.x0, tower, nb, y0, s_x0, tid);
        } else {
            // rows y0 .. y0+HXB of every X plane are one contiguous run of XB floats
            constexpr int HW = C::HIN * C::HIN, XB = C::HXB * C::HIN;
            const float* __restrict__ xs = pick3(a.X, tower) + (size_t)nb * C::CIN * HW + y0 * C::HIN;
            if constexpr (HW % 4 == 0 && XB % 4 == 0 && (C::HOB * C::HIN) % 4 == 0) {
                staged_copy<C::CIN * XB / 4, float4>(
                    tid, [&](int e) { const int pl = e / (XB / 4), r = e % (XB / 4); return reinterpret_cast<const float4*>(xs + (size_t)pl * HW)[r]; },
                    [&](int e, float4 v) { const int c = e / (XB / 4), r = e - c * (XB / 4); reinterpret_cast<float4*>(s_x)[c * (C::XPL / 4) + r] = v; });
            } else if constexpr (HW % 2 == 0 && XB % 2 == 0 && (C::HOB * C::HIN) % 2 == 0) {
                staged_copy<C::CIN * XB / 2, float2>(
                    tid, [&](int e) { const int pl = e / (XB / 2), r = e % (XB / 2); return reinterpret_cast<const float2*>(xs + (size_t)pl * HW)[r]; },
                    [&](int e, float2 v) { const int c = e / (XB / 2), r = e - c * (XB / 2); reinterpret_cast<float2*>(s_x)[c * (C::XPL / 2) + r] = v; });
            } else {
                staged_copy<C::CIN * XB, float>(
                    tid, [&](int e) { const int pl = e / XB, r = e % XB; return xs[(size_t)pl * HW + r]; },
                    [&](int e, float v) { const int c = e / XB, r = e - c * XB; s_x[c * C::XPL + r] = v; });
            }
        }
        // ---- d z
        constexpr int PZ = C::HO * C::HO, ZB = C::HOB * C::HO;
        const float* __restrict__ zs = pick3(a.dZ, tower) + (size_t)nb * C::COUT * PZ + y0 * C::HO;
        if constexpr (C::WOP == C::HO && PZ % 4 == 0 && ZB % 4 == 0) {
            // no row padding: a band of a dz plane is one contiguous run, in global memory and in LDS
            staged_copy<C::COUT * ZB / 4, float4>(
                tid, [&](int e) { const int pl = e / (ZB / 4), r = e - pl * (ZB / 4); return reinterpret_cast<const float4*>(zs + (size_t)pl * PZ)[r]; },
                [&](int e, float4 v) { const int o = e / (ZB / 4), r = e - o * (ZB / 4); reinterpret_cast<float4*>(s_z)[o * (C::ZPL / 4) + r] = v; });
        } else {
            staged_copy<C::COUT * ZB, float>(
                tid, [&](int e) { const int pl = e / ZB, r = e - pl * ZB; return zs[(size_t)pl * PZ + r]; },
                [&](int e, float v) { const int o = e / ZB, r = e - o * ZB, y = r / C::HO, x = r - y * C::HO; s_z[o * C::ZPL + y * C::WOP + x] = v; });
        }
        __syncthreads();
        if constexpr (C::XSRC == 1) {
            conv1_to_lds<C::XPL, C::HXB, 4>(s_x0, s_w1, s_x, wave, lane);
            __syncthreads();
        }
        mma_patch(0);
    }

    // ---- partial dW of this slice: part[tower][slice][band][j = c*9 + tap][o]  (output channel fastest: the 16 lanes of an
    // accumulator column group store 64 contiguous bytes)
    float* __restrict__ out = a.part + ((size_t)tower * a.slices * C::NB + (size_t)blockIdx.x) * C::COUT * C::J;
#pragma unroll
    for (int m = 0; m < C::MTW; ++m) {
        const int t = wave + 4 * m;
        if (t >= C::MT) continue;
#pragma unroll
        for (int nt = 0; nt < C::NT16; ++nt) {
            const int o = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + kq * 4 + r;
                if (j < C::J) out[j * C::COUT + o] = acc[m][nt][r];
            }
        }
#pragma unroll
        for (int u = 0; u < C::NT4; ++u) {
            const float val = ksub_reduce(acc4[m][u], kq, (lane ^ 32) << 2);
            const int o = C::NT16 * 16 + u * 4 + kq, j = t * 16 + lr;
            if (j < C::J) out[j * C::COUT + o] = val;
        }
    }
}

template <class C>
inline hipError_t wgrad_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)C::LDS_BYTES);
}

template <class C>
inline void wgrad_launch(const WgArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_mfma_kernel<C>, dim3(a.slices * C::NB, kTowers), dim3(256), C::LDS_BYTES, st, a);
}

}  // namespace scs
