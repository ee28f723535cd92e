// The dense layers of the inference head (d1 x3, FC1, fc_2) on the f16 matrix cores with f32-grade results - the split of
// conv_split.h (every f32 operand as two f16 pieces, three MFMAs per product block, f32 accumulation; operands scaled by powers of
// two) applied to  out[patch][f] = relu(sum_k x[patch][k] W[k][f] + b[f]).
//
// GEMM view as in the conv kernels ("D transposed"): A operand = weights (rows = output features, 16 per tile), resident in
// registers for the whole launch: a wave owns 3 feature tiles over ALL of K (24 registers per k-step: 408 / 432 - one wave per SIMD),
// a workgroup (4 waves) 192 features, `nfb` workgroup kinds cover the layer;  B operand = activations from LDS, a fragment read
// once and used for the wave's 3 feature tiles (9 MFMAs per 2 KB).  Workgroups are persistent over blocks of 32 patches.
// Activation rows, in memory and in LDS alike:  [32 KS halves high piece][32 KS halves low piece][8 halves pad]  (values times
// kSplitActScale): a block of 32 rows is ONE contiguous range, copied global -> LDS by global_load_lds_dwordx4 (1 KB per wave
// instruction, no registers: with ~470 weight / accumulator registers live there is no room for a register-staged pipeline), the
// next block's copy in flight under this block's MFMAs; the 16-byte pad makes the row stride 16 (mod 128) bytes, so the 16-byte
// fragment reads of 8 consecutive patches fall on 8 distinct bank groups.  The pad and the K padding columns are never written by
// anything but zeros (the buffers are cleared when they are allocated and belong to the split head alone).
//
// Replaces, at inference with option fwd_split, DenseLayer (+ rectify) of /root/reference/cnn_cort/nets.py:179, 217, 227.
#pragma once
#include "conv_mfma.h"
#include "conv_split.h"

namespace scs {

constexpr int kDsPB = 32;            // patches per block (two MFMA tiles)
constexpr int ds_row_bytes(int kp) { return 4 * kp + 16; }

template <int KS_, bool OUT_F32_, int MTW_>
struct DsCfg {
    static constexpr int KS = KS_, KP = 32 * KS_;                 // k-steps, halves per piece of a row
    static constexpr int MTW = MTW_;                              // feature tiles per wave (a workgroup kind: 4 MTW tiles = 64 MTW features)
    static constexpr bool OUT_F32 = OUT_F32_;
    static constexpr int RB = ds_row_bytes(KP), BUF_B = kDsPB * RB;
    static constexpr int CHUNKS = (BUF_B + 1023) / 1024;          // 1 KB wave-instructions per block
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF_B;
    static_assert(RB % 128 == 16, "row stride: 8 consecutive rows on 8 distinct 16-byte bank groups");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct DenseSplitArgs {
    const unsigned char* in[kTowers];   // rows [n] of ds_row_bytes(32 KS) bytes
    const uint4* w[kTowers];      // [k-step][feature tile][piece][lane] x 16 bytes (pack_dense_split_kernel), mt_total feature tiles
    const float* bias[kTowers];   // [mt_total * 16] (zero beyond the layer's features), then one float: 1 / (kSplitActScale * weight scale)
    void* out[kTowers];           // OUT_F32: f32 rows [n][ld_out]; else rows [n] of ds_row_bytes(kp_out) bytes
    int col0[kTowers];            // first output column of the problem (d1: 180 per tower into FC1's input row)
    int n_store[kTowers];         // features stored (a multiple of 4; beyond the layer's own: zeros = the next layer's K padding)
    int ld_out, kp_out;
    int mt_total, nfb;            // feature tiles in the packed arrays, workgroup kinds (4 MTW tiles each)
    int n;
    unsigned int* flag;
};

typedef __attribute__((address_space(3))) void* ds_lds_ptr;
typedef const __attribute__((address_space(1))) void* ds_glb_ptr;

template <class C>
__global__ __launch_bounds__(256, 1) void dense_split_kernel(DenseSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ds_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, j = lane >> 4;
    const int prob = blockIdx.y, fb = blockIdx.x % a.nfb, wg = blockIdx.x / a.nfb, nwg = gridDim.x / a.nfb;
    const int mt0 = fb * 4 * C::MTW + wave * C::MTW;                     // this wave's first feature tile
    const uint4* __restrict__ w = pick3(a.w, prob);
    f16x8 wh[C::KS][C::MTW], wl[C::KS][C::MTW];
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
#pragma unroll
        for (int m = 0; m < C::MTW; ++m) {
            const int mt = min(mt0 + m, a.mt_total - 1);                 // (tiles beyond the layer: any finite operand - nothing of them is stored)
            wh[s][m] = __builtin_bit_cast(f16x8, w[((size_t)(s * a.mt_total + mt) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, w[((size_t)(s * a.mt_total + mt) * 2 + 1) * 64 + lane]);
        }
    const float* __restrict__ bias = pick3(a.bias, prob);
    const float inv = bias[a.mt_total * 16];
    const unsigned char* __restrict__ inp = pick3(a.in, prob);
    unsigned char* __restrict__ outp = reinterpret_cast<unsigned char*>(pick3(a.out, prob));
    const int col0 = pick3(a.col0, prob), n_store = pick3(a.n_store, prob);
    const size_t in_bytes = (size_t)a.n * C::RB;
    float vmax = 0.f;
    // the block of rows at patch p0 -> LDS buffer `buf`: chunks of 1 KB, one wave instruction each (bytes beyond the last row: the
    // last 16 bytes of the array once more - finite, and nothing computed from them is stored)
    auto copy_block = [&](int p0, unsigned char* buf) {
        const size_t base = (size_t)p0 * C::RB;
#pragma unroll
        for (int c0 = 0; c0 < C::CHUNKS; c0 += 4) {
            const int c = c0 + wave;
            if (c < C::CHUNKS) {
                const int off = c * 1024 + lane * 16;
                if (off < C::BUF_B) {
                    size_t g = base + off;
                    if (g > in_bytes - 16) g = in_bytes - 16;
                    __builtin_amdgcn_global_load_lds((ds_glb_ptr)(inp + g), (ds_lds_ptr)(buf + c * 1024), 16, 0, 0);
                }
            }
        }
    };
    int p0 = wg * kDsPB, b = 0;
    if (p0 < a.n) copy_block(p0, ds_smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's copies have landed (an LDS-DMA is a pending write on the VM counter) ...
    __syncthreads();                                         // ... and so have everybody else's
    for (; p0 < a.n; p0 += nwg * kDsPB) {
        const unsigned char* cur = ds_smem + (size_t)b * C::BUF_B;
        const int pn = p0 + nwg * kDsPB;
        if (pn < a.n) copy_block(pn, ds_smem + (size_t)(b ^ 1) * C::BUF_B);
        f32x4 acc[2][C::MTW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < C::MTW; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment (k-step s, tile i): 16 bytes of this lane's patch row at 64 s + 16 j, high and low piece; one ahead in flight
        const unsigned char* fr = cur + n * C::RB + 16 * j;
        u32x4 fh[2], fl[2];
        auto frag = [&](int s, int i, int slot) {
            const unsigned char* p = fr + i * 16 * C::RB + 64 * s;
            fh[slot] = *reinterpret_cast<const u32x4*>(p);
            fl[slot] = *reinterpret_cast<const u32x4*>(p + 2 * C::KP);
        };
        frag(0, 0, 0);
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int slot = (2 * s + i) & 1;
                if (2 * s + i + 1 < 2 * C::KS) frag(i == 0 ? s : s + 1, i ^ 1, slot ^ 1);
                const f16x8 bh = __builtin_bit_cast(f16x8, fh[slot]), bl = __builtin_bit_cast(f16x8, fl[slot]);
#pragma unroll
                for (int m = 0; m < C::MTW; ++m) {
                    acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[i][m], 0, 0, 0);
                    acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[i][m], 0, 0, 0);
                    acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[i][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: lane (n, j) holds features 16 mt + 4 j .. + 3 of patch p0 + 16 i + n
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = p0 + 16 * i + n;
#pragma unroll
            for (int m = 0; m < C::MTW; ++m) {
                const int f = 16 * (mt0 + m) + 4 * j;
                if (p >= a.n || f >= n_store) continue;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + f);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(acc[i][m][q], inv, bv[q]), 0.f);
                if constexpr (C::OUT_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + (size_t)p * a.ld_out + col0 + f) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= kSplitActScale;
                    vmax = fmaxf(fmaxf(vmax, v[0]), fmaxf(v[1], fmaxf(v[2], v[3])));
                    f16x4 hh, ll;
                    sp_split4(v, hh, ll);
                    unsigned char* o = outp + (size_t)p * ds_row_bytes(a.kp_out) + 2 * (col0 + f);
                    *reinterpret_cast<f16x4*>(o) = hh;
                    *reinterpret_cast<f16x4*>(o + 2 * a.kp_out) = ll;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the next block's rows has landed (and its stores have left)
        __syncthreads();                                     // the whole block has; everybody is done with this one's rows
        b ^= 1;
    }
    if (!(vmax <= kSplitOverflow)) atomicOr(a.flag, 1u);
}

template <class C>
inline hipError_t dense_split_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_split_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}
// nprob problems (towers) x nfb workgroup kinds x persistent workgroups: one workgroup per CU
template <class C>
inline void dense_split_launch(const DenseSplitArgs& a, int nprob, hipStream_t st, int cus = 256) {
    const int blocks = (a.n + kDsPB - 1) / kDsPB;
    int g = cus / (nprob * a.nfb);
    if (g > blocks) g = blocks;
    if (g < 1) return;
    hipLaunchKernelGGL(dense_split_kernel<C>, dim3(a.nfb * g, nprob), dim3(256), C::LDS_BYTES, st, a);
}

using DsK17 = DsCfg<17, false, 3>;       // d1 (conv5's rows [540 -> 544]) and FC1 (rows [540 -> 544]): rows out; 192 features per workgroup kind
using DsFC2 = DsCfg<18, true, 2>;        // fc_2: rows [555 -> 576] -> f32 rows [272]; 128 features per workgroup kind (432 weight registers
                                         // for three tiles over 18 k-steps leave no room for the rest: spills)

// Split operands of a dense layer W [K][N] (x W + b) by ONE workgroup of 256 threads: weight scale = the power of two that puts max |W|
// into [2^12, 2^13); lane (r = lane & 15, j = lane >> 4) of feature tile mt holds the 8 halves k = 32 s + 8 j .. + 7 of feature
// 16 mt + r, high and low piece; bias [mt_total * 16] zero padded + 1 / (kSplitActScale * weight scale)
static __global__ __launch_bounds__(256) void pack_dense_split_kernel(const float* __restrict__ W, const float* __restrict__ b, int K, int N, int KS, int mt_total,
                                                               uint4* __restrict__ wp, float* __restrict__ bias) {
    __shared__ float s_red[256];
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int e = tid; e < K * N; e += 256) mx = fmaxf(mx, fabsf(W[e]));
    s_red[tid] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) s_red[tid] = fmaxf(s_red[tid], s_red[tid + k]); __syncthreads(); }
    mx = s_red[0];
    int ex = 0;
    if (mx > 0.f && mx < 3e38f) (void)frexpf(mx, &ex); else ex = 13;
    const float sw = ldexpf(1.f, 13 - ex);
    for (int idx = tid; idx < KS * mt_total * 64; idx += 256) {
        const int lane = idx & 63, mt = (idx >> 6) % mt_total, s = (idx >> 6) / mt_total, r = lane & 15, j = lane >> 4, o = 16 * mt + r;
        unsigned short hb[8], lb[8];
        for (int pos = 0; pos < 8; ++pos) {
            const int k = 32 * s + 8 * j + pos;
            const float v = (k < K && o < N) ? W[(size_t)k * N + o] * sw : 0.f;
            const _Float16 hh = (_Float16)v, ll = (_Float16)(v - (float)hh);
            hb[pos] = __builtin_bit_cast(unsigned short, hh); lb[pos] = __builtin_bit_cast(unsigned short, ll);
        }
        wp[((size_t)(s * mt_total + mt) * 2 + 0) * 64 + lane] = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
        wp[((size_t)(s * mt_total + mt) * 2 + 1) * 64 + lane] = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
    }
    for (int f = tid; f < mt_total * 16; f += 256) bias[f] = f < N ? b[f] : 0.f;
    if (tid == 0) bias[mt_total * 16] = 1.f / (kSplitActScale * sw);
}

constexpr int kDsD1Tiles = 12, kDsFC1Tiles = 36, kDsFC2Tiles = 24;       // feature tiles of the packed arrays (12 per workgroup kind)
constexpr int kDsKpFC1 = 544, kDsKpFC2 = 576;                           // halves per activation row in front of FC1 / fc_2

}  // namespace scs
