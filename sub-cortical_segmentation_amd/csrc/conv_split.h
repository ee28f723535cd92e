// Inference convolutions on the f16 matrix cores with f32-grade results: every f32 operand is carried as TWO f16 pieces
// (a = ah + al, ah = f16(a), al = f16(a - ah): 22 significant bits) and a product is the three MFMAs ah*bh + ah*bl + al*bh with
// f32 accumulation (every f16 x f16 product is exact in f32; the dropped al*bl is 2^-22 of the product).  Operands are scaled by
// powers of two so that their low pieces stay clear of the f16 subnormals (tools/f16split_probe.hip: at operand magnitudes
// >= 2^4 the error against double equals the f32 MFMA chain's, 2.8e-8 of the output scale).  Three v_mfma_f32_16x16x32_f16
// (16 cycles each) do the work of eight v_mfma_f32_16x16x4_f32 (32 cycles each).
//
// GEMM view ("D transposed"):  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]
//   A operand = weights (rows = output channels, 16 per tile), resident in registers for the whole launch (persistent workgroups
//   of one tower);  B operand = activations from LDS, stored [pixel][channel] (channel innermost) as two f16 planes, so that for
//   a fixed tap row dy the K index (dx, cin) is ONE contiguous run of 3 * Cin halves starting at pixel (y + dy, x): a lane's 8
//   consecutive k are 16 contiguous bytes.  A run is padded to a multiple of 32 k (20 channels: 60 -> 64; the pad reads the next
//   pixel's first channels against zero weights).  D: lane (n = l & 15, j = l >> 4) holds output channels 4j .. 4j+3 of pixel n,
//   i.e. 8 contiguous bytes of the next layer's [pixel][channel] f16 plane.
//   Pooling: the four N tiles of a group are the four positions of the 2x2 windows of 16 pooled pixels: max-pool is an in-lane max.
//   LDS banks: 16 consecutive pooled pixels are 80 bytes apart (20 dwords, 5 mod 16 in units of 4 dwords): their 8-byte reads
//   cover every fourth bank pair exactly once; lane groups with odd j read the two halves of their 16 bytes in the opposite order
//   (the weights of those lanes are packed with the same swap), which puts them on the remaining bank pairs: conflict-free.
//
// conv1 (Cin = 1, K = 9) stays on the f32 MFMA (exact) in the same transposed form: its output is scaled, split and written to
// the LDS planes by the lanes that hold it.
//
// Replaces, at inference, Conv2DLayer -> BatchNormLayer -> rectify (-> MaxPool2DLayer) of /root/reference/cnn_cort/nets.py:171-173
// (BN folded into weights and bias, true convolution = correlation with the flipped filter).
#pragma once
#include "scs_common.h"

namespace scs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr float kSplitActScale = 64.f;         // activations are stored times 2^6 (overflow beyond |a| = 1023: flagged, see below)
constexpr float kSplitOverflow = 60000.f;      // a scaled activation above this sets the launch's overflow flag
constexpr int kSp12PlanePx = 904;              // conv1 plane: 30 x 30 pixels + 4 pad pixels (run overrun of the last rows)
constexpr int kSp12PlaneBytes = kSp12PlanePx * 40;
constexpr int kSp12XFloats = 1024 + 64;        // raw patch + pad (tiles of the 32-wide domain read up to 2 rows + 3 past the end)
constexpr size_t kSp12LdsBytes = 2 * (size_t)kSp12PlaneBytes + kSp12XFloats * 4;
constexpr int kSp12W1Floats = 12 * 20 + 20;    // conv1 operands: [slot 4 dy + dx (dx = 3: zero)][20 couts] + bias[20], both times kSplitActScale
constexpr int kSp12W2Vec = 6 * 2 * 2 * 64;     // conv2 operands: [k-step][M tile][piece][lane] x 16 bytes

struct Split12Args {
    const float* in[kTowers];     // [n][32][32]
    const float* w1[kTowers];     // kSp12W1Floats
    const uint4* w2[kTowers];     // kSp12W2Vec
    const float* b2[kTowers];     // [20] folded bias of conv2 (unscaled)
    float inv2[kTowers];          // 1 / (kSplitActScale * weight scale of conv2): power of two
    float* out[kTowers];          // [n][20][14][14] f32 (the f32 kernels' pool1 layout)
    int n;
    unsigned int* flag;           // set to 1 when a scaled activation left the f16 range (results of the launch are then void)
};

// element e (0..63) of the run of tap row dy -> (dx, cin); e >= 60: padding
__host__ __device__ constexpr int sp_run_dx(int e, int cin) { return e / cin; }

// lane's element order inside its 16-byte chunk: lane groups with odd j hold the two 8-byte halves swapped
__host__ __device__ constexpr int sp_chunk_elem(int j, int pos) { return (j & 1) ? (pos ^ 4) : pos; }

__device__ __forceinline__ void sp_split4_store(f32x4 v, unsigned char* hi, unsigned char* lo, int off, float& vmax) {
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    vmax = fmaxf(fmaxf(vmax, v[0]), v[1]);
    vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);
    f16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (_Float16)v[i]; l[i] = (_Float16)(v[i] - (float)h[i]); }
    *reinterpret_cast<f16x4*>(hi + off) = h;
    *reinterpret_cast<f16x4*>(lo + off) = l;
}

__device__ __forceinline__ f16x8 sp_frag(const unsigned char* p0, const unsigned char* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(f16x8, v);
}

template <int LANES_MASKED = 0>
__device__ __forceinline__ float sp_quad_max(float v) {
    // max over the 4 lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1])
    const int a = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
    v = fmaxf(v, __builtin_bit_cast(float, a));
    const int b = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);
    return fmaxf(v, __builtin_bit_cast(float, b));
}

// grid = 3 * G workgroups of 256 threads: workgroup b serves tower b % 3, patches b / 3, b / 3 + G, ...
__global__ __launch_bounds__(256, 2) void conv12_split_kernel(Split12Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned char* s_hi = sp_smem;
    unsigned char* s_lo = sp_smem + kSp12PlaneBytes;
    float* s_x = reinterpret_cast<float*>(sp_smem + 2 * kSp12PlaneBytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    // pads: zero once (never written afterwards)
    for (int e = tid; e < 160 / 4; e += 256) {
        reinterpret_cast<unsigned*>(s_hi + 900 * 40)[e] = 0u;
        reinterpret_cast<unsigned*>(s_lo + 900 * 40)[e] = 0u;
    }
    if (tid < 64) s_x[1024 + tid] = 0.f;
    // ---- resident operands
    const float* w1 = a.w1[tower];
    float a1[3], a4[9];
    f32x4 b1m, b1r;
#pragma unroll
    for (int s = 0; s < 3; ++s) a1[s] = w1[(4 * s + j) * 20 + n];                        // couts 0..15 (n = row of the A operand)
#pragma unroll
    for (int t = 0; t < 9; ++t) a4[t] = w1[(4 * (t / 3) + t % 3) * 20 + 16 + (lane & 3)];   // couts 16..19 on the 4x4x1 blocks
#pragma unroll
    for (int i = 0; i < 4; ++i) { b1m[i] = w1[240 + 4 * j + i]; b1r[i] = w1[240 + 16 + i]; }
    f16x8 wh[6][2], wl[6][2];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wh[s][m] = __builtin_bit_cast(f16x8, a.w2[tower][((s * 2 + m) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, a.w2[tower][((s * 2 + m) * 2 + 1) * 64 + lane]);
        }
    f32x4 bias0, bias1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bias0[i] = a.b2[tower][4 * j + i]; bias1[i] = a.b2[tower][16 + i]; }
    const float inv = a.inv2[tower];
    // per-lane byte offsets of the wave's three pooling groups (16 pooled pixels each) and of the odd tile (pooled 192..195)
    int rd0[3], rd1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int w = 16 * (wave + 4 * k) + n, p0 = (2 * (w / 14)) * 30 + 2 * (w % 14);
        const int x = 40 * p0 + 16 * j;
        rd0[k] = x + 8 * (j & 1);
        rd1[k] = x + 8 - 8 * (j & 1);
    }
    int rs0, rs1;
    {
        const int w = 192 + (n >> 2), q = n & 3, p0 = (2 * (w / 14) + (q >> 1)) * 30 + 2 * (w % 14) + (q & 1);
        const int x = 40 * p0 + 16 * j;
        rs0 = x + 8 * (j & 1);
        rs1 = x + 8 - 8 * (j & 1);
    }
    float vmax = 0.f;
    const float* inp = a.in[tower];
    float* outp = a.out[tower];
    float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wg < a.n) xr = reinterpret_cast<const float4*>(inp + (size_t)wg * 1024)[tid];
    int it = 0;
    for (int p = wg; p < a.n; p += nwg, ++it) {
        reinterpret_cast<float4*>(s_x)[tid] = xr;
        __syncthreads();                                     // raw patch in place; every wave is done with the planes of the previous patch
        if (p + nwg < a.n) xr = reinterpret_cast<const float4*>(inp + (size_t)(p + nwg) * 1024)[tid];
        // ---- conv1 on the f32 matrix cores: 60 tiles of 16 pixels over the 32-wide row domain (columns 30, 31 are not stored)
#pragma unroll 3
        for (int tt = 0; tt < 15; ++tt) {
            const int T = wave + 4 * tt;
            f32x4 d = b1m;
#pragma unroll
            for (int s = 0; s < 3; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], s_x[16 * T + n + j + 32 * s], d, 0, 0, 0);
            const int y = T >> 1, x = 16 * (T & 1) + n;
            if (x < 30) sp_split4_store(d, s_hi, s_lo, (y * 30 + x) * 40 + 8 * j, vmax);
        }
        for (int Q = wave; Q < 15; Q += 4) {
            f32x4 d = b1r;
#pragma unroll
            for (int t = 0; t < 9; ++t) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[t], s_x[64 * Q + lane + (t / 3) * 32 + t % 3], d, 0, 0, 0);
            const int pp = 64 * Q + lane, y = pp >> 5, x = pp & 31;
            if (x < 30) sp_split4_store(d, s_hi, s_lo, (y * 30 + x) * 40 + 32, vmax);
        }
        __syncthreads();                                     // planes complete
        // ---- conv2 on the f16 matrix cores
        float* op = outp + (size_t)p * (20 * 196);
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const unsigned char* h0 = s_hi + rd0[k];
            const unsigned char* h1 = s_hi + rd1[k];
            const unsigned char* l0 = s_lo + rd0[k];
            const unsigned char* l1 = s_lo + rd1[k];
            f32x4 acc[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int s = 0; s < 6; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = ((s >> 1) + (q >> 1)) * 1200 + (q & 1) * 40 + (s & 1) * 64;
                    const f16x8 bh = sp_frag(h0 + off, h1 + off), bl = sp_frag(l0 + off, l1 + off);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[q][m], 0, 0, 0);
                    }
                }
            }
            const int w = 16 * (wave + 4 * k) + n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = fmaxf(fmaxf(acc[0][0][i], acc[1][0][i]), fmaxf(acc[2][0][i], acc[3][0][i]));
                op[(4 * j + i) * 196 + w] = fmaxf(v0 * inv + bias0[i], 0.f);
                const float v1 = fmaxf(fmaxf(acc[0][1][i], acc[1][1][i]), fmaxf(acc[2][1][i], acc[3][1][i]));
                if (j == 0) op[(16 + i) * 196 + w] = fmaxf(v1 * inv + bias1[i], 0.f);
            }
        }
        if (wave == (it & 3)) {                              // the odd tile: pooled pixels 192..195, the four positions of each in one quad of lanes
            const unsigned char* h0 = s_hi + rs0;
            const unsigned char* h1 = s_hi + rs1;
            const unsigned char* l0 = s_lo + rs0;
            const unsigned char* l1 = s_lo + rs1;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int off = (s >> 1) * 1200 + (s & 1) * 64;
                const f16x8 bh = sp_frag(h0 + off, h1 + off), bl = sp_frag(l0 + off, l1 + off);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[m], 0, 0, 0);
                }
            }
            const int w = 192 + (n >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = sp_quad_max(acc[0][i]), v1 = sp_quad_max(acc[1][i]);
                if ((n & 3) == 0) {
                    op[(4 * j + i) * 196 + w] = fmaxf(v0 * inv + bias0[i], 0.f);
                    if (j == 0) op[(16 + i) * 196 + w] = fmaxf(v1 * inv + bias1[i], 0.f);
                }
            }
        }
    }
    if (vmax > kSplitOverflow) atomicOr(a.flag, 1u);
}

inline hipError_t conv12_split_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSp12LdsBytes);
}

inline void conv12_split_launch(const Split12Args& a, hipStream_t st, int wgs_per_tower = 170) {
    int g = wgs_per_tower < a.n ? wgs_per_tower : a.n;
    if (g < 1) return;
    hipLaunchKernelGGL(conv12_split_kernel, dim3(3 * g), dim3(256), kSp12LdsBytes, st, a);
}

}  // namespace scs
