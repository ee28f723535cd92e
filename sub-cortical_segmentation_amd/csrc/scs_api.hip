// libscs - C ABI (include/scs.h) and the inference pipeline of the tri-planar CNN on gfx950.
//
// Inference per chunk of patches (all launches on the caller's stream):
//   conv1+conv2+pool1 (fused, LDS-resident conv1 output)        nets.py:171-173
//   conv3                                                         nets.py:174
//   conv4+pool2                                                   nets.py:175-176
//   conv5                                                         nets.py:177
//   d1 x3 -> concat [n][540]                                      nets.py:179, 215
//   FC1 -> [n][556] (+ atlas columns 540..554)                    nets.py:217, 222-224
//   fc_2 -> [n][272]                                              nets.py:227
//   out_layer + softmax + argmax                                  nets.py:231; base.py:425-438
// Dropout layers are identity at inference, PReLU after ReLU is the identity for any alpha.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "conv_mfma.h"
#include "gemm_mfma.h"
#include "scs_ctx.h"
#include "dense_eval.h"
#include "conv_fused.h"

namespace scs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ----------------------------------------------------------------------------- layer configurations
// <CIN, COUT, HIN, patches per workgroup, epilogue, source, LDS plane stride, bands, workgroups per CU>
using CfgConv12 = ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_CONV1, 496, 2, 2>;    // half a patch per workgroup
using CfgConv12V = ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_VOLUME, 496, 2, 2>;
using CfgConv12X = ConvCfg<20, 20, 30, 1, EPI_BIAS_RELU_POOL, SRC_X0, 496, 2, 2>;     // patches cut from a set of volumes (sample table)
using CfgConv3 = ConvCfg<20, 40, 14, 2, EPI_BIAS_RELU, SRC_NCHW, 208, 1, 3>;
using CfgConv4 = ConvCfg<40, 40, 12, 1, EPI_BIAS_RELU_POOL, SRC_NCHW, 144, 1, 4>;
using CfgConv5 = ConvCfg<40, 60, 5, 7, EPI_BIAS_RELU, SRC_NCHW, 27, 1, 3>;     // 7 x 9 = 63 of 64 rows: four full tiles
using CfgConv4NP = ConvCfg<40, 40, 12, 1, EPI_BIAS_RELU, SRC_NCHW, 144, 1, 4>;  // conv4 without pool2 (shared-tower evaluation: pool2 has four window origins)

// ----------------------------------------------------------------------------- packing kernels
// conv i (Cin >= 4): packed chunks (scs_common.h), element = W[n][4*c4+kk][2-dy][2-dx] * scale[n]  (true conv -> correlation)
__global__ void pack_conv_kernel(const float* __restrict__ W, const float* __restrict__ bn, int cin, int cout,
                                 int fold, float* __restrict__ wp, float* __restrict__ bias) {
    const int total = (cin / 4) * conv_chunk_floats(cout);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const PackIdx ix = conv_pack_index(e, cout);
        float v = 0.f;
        if (ix.n >= 0) {
            const int c = ix.c4 * 4 + ix.kk, dy = ix.tap / 3, dx = ix.tap % 3;
            v = W[((ix.n * cin + c) * 3 + (2 - dy)) * 3 + (2 - dx)];
            if (fold) v *= bn[cout + ix.n] * bn[3 * cout + ix.n];          // gamma * inv_std
        }
        wp[e] = v;
    }
    const int nb = n_tiles(cout) * 16;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nb; n += gridDim.x * blockDim.x) {
        float b = 0.f;
        if (fold && n < cout) b = bn[n] - bn[2 * cout + n] * bn[cout + n] * bn[3 * cout + n];   // beta - mean*scale
        bias[n] = b;
    }
}

// conv1 operands for the MFMA form (conv_mfma.h: conv1_to_lds): [12][16] main, [12][4] leftover, [20] bias;
// element (k = tap, c) = W[c][0][2-dy][2-dx] * scale[c] (true conv -> correlation, BN folded), rows 9..11 zero
__global__ void pack_conv1_kernel(const float* __restrict__ W, const float* __restrict__ bn, int fold,
                                  float* __restrict__ w1p) {
    const int t = threadIdx.x;
    if (t >= kW1Floats) return;
    float v = 0.f;
    if (t < 240) {
        const int k = t < 192 ? t / 16 : (t - 192) / 4;
        const int c = t < 192 ? t % 16 : 16 + (t - 192) % 4;
        if (k < 9) {
            v = W[(c * 3 + (2 - k / 3)) * 3 + (2 - k % 3)];
            if (fold) v *= bn[20 + c] * bn[60 + c];                          // gamma * inv_std
        }
    } else {
        const int c = t - 240;
        v = fold ? bn[c] - bn[40 + c] * bn[20 + c] * bn[60 + c] : 0.f;       // beta - mean*scale
    }
    w1p[t] = v;
}

// ---- split operands (conv_split.h).  One workgroup per layer: weight scale = the power of two that puts max |folded weight| into
// [2^12, 2^13), then every lane's 8 halves per (k-step, channel tile), high and low piece; bias [tiles * 16] + 1 / (act scale * weight scale)
// stack: conv2 only - behind the plain tiles, the stacked remainder tiles of conv_split.h (kSp12W2StackVec; sp12_stack_row)
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ W, const float* __restrict__ bn, int cin, int cout,
                                                         uint4* __restrict__ wp, float* __restrict__ bias, int stack) {
    __shared__ float s_red[256];
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int e = tid; e < cout * cin * 9; e += 256) { const int n = e / (cin * 9); mx = fmaxf(mx, fabsf(W[e] * (bn[cout + n] * bn[3 * cout + n]))); }
    s_red[tid] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) s_red[tid] = fmaxf(s_red[tid], s_red[tid + k]); __syncthreads(); }
    mx = s_red[0];
    int ex = 0;
    if (mx > 0.f && mx < 3e38f) (void)frexpf(mx, &ex); else ex = 13;
    const float sw = ldexpf(1.f, 13 - ex);
    const int KSR = (3 * cin + 31) / 32, KS = 3 * KSR, MT = (cout + 15) / 16;
    for (int idx = tid; idx < KS * MT * 64; idx += 256) {
        const int lane = idx & 63, m = (idx >> 6) % MT, st = (idx >> 6) / MT, r = lane & 15, j = lane >> 4, dy = st / KSR, ks = st % KSR, o = 16 * m + r;
        unsigned short hb[8], lb[8];
        for (int pos = 0; pos < 8; ++pos) {
            const int e = 32 * ks + 8 * j + sp_chunk_elem(j, pos);
            float v = 0.f;
            if (e < 3 * cin && o < cout) {
                const int c = e % cin, dx = e / cin;
                v = W[((o * cin + c) * 3 + (2 - dy)) * 3 + (2 - dx)];                 // true convolution -> correlation (as pack_conv_kernel)
                v *= bn[cout + o] * bn[3 * cout + o];                                 // gamma * inv_std
                v *= sw;
            }
            const _Float16 hh = (_Float16)v, ll = (_Float16)(v - (float)hh);
            hb[pos] = __builtin_bit_cast(unsigned short, hh); lb[pos] = __builtin_bit_cast(unsigned short, ll);
        }
        wp[((st * MT + m) * 2 + 0) * 64 + lane] = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
        wp[((st * MT + m) * 2 + 1) * 64 + lane] = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
    }
    if (stack) {
        uint4* ws = wp + KS * MT * 2 * 64;
        for (int idx = tid; idx < 4 * 2 * 64; idx += 256) {
            const int lane = idx & 63, hh = (idx >> 6) & 1, R = idx >> 7, r = lane & 15, j = lane >> 4;
            int o = 0, dy = 0;
            const bool row = sp12_stack_row(R, r, o, dy);
            unsigned short hb[8], lb[8];
            for (int pos = 0; pos < 8; ++pos) {
                const int e = 32 * hh + 8 * j + sp_chunk_elem(j, pos);
                float v = 0.f;
                if (row && e < 3 * cin && o < cout) {
                    const int c = e % cin, dx = e / cin;
                    v = W[((o * cin + c) * 3 + (2 - dy)) * 3 + (2 - dx)];
                    v *= bn[cout + o] * bn[3 * cout + o];
                    v *= sw;                                                              // (the roundings of the plain tiles above)
                }
                const _Float16 hv = (_Float16)v, lv = (_Float16)(v - (float)hv);
                hb[pos] = __builtin_bit_cast(unsigned short, hv); lb[pos] = __builtin_bit_cast(unsigned short, lv);
            }
            ws[((R * 2 + hh) * 2 + 0) * 64 + lane] = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
            ws[((R * 2 + hh) * 2 + 1) * 64 + lane] = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
        }
    }
    for (int n = tid; n < MT * 16; n += 256) bias[n] = n < cout ? bn[n] - bn[2 * cout + n] * bn[cout + n] * bn[3 * cout + n] : 0.f;
    if (tid == 0) bias[MT * 16] = 1.f / (kSplitActScale * sw);
}
// conv1 for the transposed f32 MFMA form: [slot 4 dy + dx][20] + bias[20], both times the activation scale
__global__ void pack_split1_kernel(const float* __restrict__ W, const float* __restrict__ bn, float* __restrict__ w1p) {
    const int t = threadIdx.x;
    if (t >= kSp12W1Floats) return;
    float v = 0.f;
    if (t < 240) {
        const int slot = t / 20, c = t % 20, dy = slot >> 2, dx = slot & 3;
        if (dx < 3) v = W[(c * 3 + (2 - dy)) * 3 + (2 - dx)] * (bn[20 + c] * bn[60 + c]) * kSplitActScale;
    } else {
        const int c = t - 240;
        v = (bn[c] - bn[40 + c] * bn[20 + c] * bn[60 + c]) * kSplitActScale;
    }
    w1p[t] = v;
}

// dense: Wp[k][n] over [K4][NS] zero padded, bias padded to NS
__global__ void pack_dense_kernel(const float* __restrict__ W, const float* __restrict__ b, int K, int N, int K4,
                                  int NS, float* __restrict__ wp, float* __restrict__ bp) {
    const int total = K4 * NS;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int n = e % NS, k = e / NS;
        wp[e] = (k < K && n < N) ? W[(size_t)k * N + n] : 0.f;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < NS; n += gridDim.x * blockDim.x) bp[n] = n < N ? b[n] : 0.f;
}

// ----------------------------------------------------------------------------- small kernels of the head
// atlas -> columns 540..554 of the FC1 output rows (ConcatLayer elem_channels2, nets.py:224); column 555 = 0
// (guard, here and in the next two kernels: null, or a device word - the kernel does nothing while it is zero; forward_chunk_split)
__global__ void atlas_concat_kernel(const float* __restrict__ atlas, const int32_t* __restrict__ index, int n_rows, float* __restrict__ fc1, int n,
                                    const unsigned int* __restrict__ guard) {
    if (guard && *guard == 0u) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 16) return;
    const int row = e >> 4, c = e & 15;
    const int src = index ? max(0, min(index[row], n_rows - 1)) : row;
    fc1[(size_t)row * kFC1ld + kFC1 + c] = c < 15 ? atlas[(size_t)src * 15 + c] : 0.f;
}

// same from the 4-D prior volume at the patch centres, with the zero-row -> background rule (base.py:388-394)
__global__ void atlas_gather_kernel(const float* __restrict__ atlas4d, const int32_t* __restrict__ centers,
                                    int dx, int dy, int dz, float* __restrict__ fc1, int n, const unsigned int* __restrict__ guard) {
    if (guard && *guard == 0u) return;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int x = centers[row * 3], y = centers[row * 3 + 1], z = centers[row * 3 + 2];
    const bool inside = (unsigned)x < (unsigned)dx && (unsigned)y < (unsigned)dy && (unsigned)z < (unsigned)dz;
    const float* src = atlas4d + (inside ? (((size_t)x * dy + y) * dz + z) * 15 : 0);
    float v[15], s = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) { v[c] = inside ? src[c] : 0.f; s += v[c]; }     // a centre outside the volume reads as "no prior"
    if (s == 0.f) v[14] = 1.f;
    float* dst = fc1 + (size_t)row * kFC1ld + kFC1;
#pragma unroll
    for (int c = 0; c < 15; ++c) dst[c] = v[c];
    dst[15] = 0.f;
}

// the same two for the split dense head (dense_split.h): columns 540..554 of fc_2's input rows as high / low halves times kSplitActScale
__device__ __forceinline__ void atlas_row_split(const float (&v)[15], unsigned char* __restrict__ rows, int row) {
    _Float16* hi = reinterpret_cast<_Float16*>(rows + (size_t)row * ds_row_bytes(kDsKpFC2)) + kFC1;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        const float x = v[c] * kSplitActScale;
        const _Float16 hh = (_Float16)x;
        hi[c] = hh; hi[kDsKpFC2 + c] = (_Float16)(x - (float)hh);
    }
}
__global__ void atlas_concat_split_kernel(const float* __restrict__ atlas, const int32_t* __restrict__ index, int n_rows, unsigned char* __restrict__ rows, int n) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int src = index ? max(0, min(index[row], n_rows - 1)) : row;
    float v[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) v[c] = atlas[(size_t)src * 15 + c];
    atlas_row_split(v, rows, row);
}
__global__ void atlas_gather_split_kernel(const float* __restrict__ atlas4d, const int32_t* __restrict__ centers,
                                          int dx, int dy, int dz, unsigned char* __restrict__ rows, int n) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int x = centers[row * 3], y = centers[row * 3 + 1], z = centers[row * 3 + 2];
    const bool inside = (unsigned)x < (unsigned)dx && (unsigned)y < (unsigned)dy && (unsigned)z < (unsigned)dz;
    const float* src = atlas4d + (inside ? (((size_t)x * dy + y) * dz + z) * 15 : 0);
    float v[15], s = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) { v[c] = inside ? src[c] : 0.f; s += v[c]; }     // (as atlas_gather_kernel: base.py:388-394)
    if (s == 0.f) v[14] = 1.f;
    atlas_row_split(v, rows, row);
}

// out_layer (270 -> 15) + softmax + argmax, one lane per patch row (4 050 MACs per row: 0.02 % of the net).
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ fc2, const float* __restrict__ wout /*[270][16]*/,
                                                      const float* __restrict__ bout /*[16]*/, int n,
                                                      float* __restrict__ logits /*[n][16] or null*/,
                                                      float* __restrict__ probs /*[n][15] or null*/,
                                                      int32_t* __restrict__ labels /*[n] or null*/,
                                                      const unsigned int* __restrict__ guard) {
    if (guard && *guard == 0u) return;
    __shared__ __attribute__((aligned(16))) float s_w[kFC2 * 16 + 16];
    for (int e = threadIdx.x; e < kFC2 * 16; e += blockDim.x) s_w[e] = wout[e];
    if (threadIdx.x < 16) s_w[kFC2 * 16 + threadIdx.x] = bout[threadIdx.x];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float z[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) z[c] = s_w[kFC2 * 16 + c];
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(fc2 + (size_t)row * kFC2ld);
    for (int k4 = 0; k4 < kFC2ld / 4; ++k4) {
        const float4 xv = x4[k4];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k4 * 4 + u;
            if (k < kFC2) {
#pragma unroll
                for (int c = 0; c < 15; ++c) z[c] = fmaf(xs[u], s_w[k * 16 + c], z[c]);
            }
        }
    }
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < 15; ++c) mx = fmaxf(mx, z[c]);
    float p[15], s = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) { p[c] = expf(z[c] - mx); s += p[c]; }
    const float inv = 1.f / s;
    int best = 0;
    float bp = -1.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        p[c] *= inv;
        if (p[c] > bp) { bp = p[c]; best = c; }       // strict '>' keeps the first maximum (np.argmax)
    }
    if (logits) {
#pragma unroll
        for (int c = 0; c < 15; ++c) logits[(size_t)row * 16 + c] = z[c];
        logits[(size_t)row * 16 + 15] = 0.f;
    }
    if (probs) {
#pragma unroll
        for (int c = 0; c < 15; ++c) probs[(size_t)row * 15 + c] = p[c];
    }
    if (labels) labels[row] = best;
}

// one step of binary dilation with the 6-neighbour cross (scipy.ndimage.binary_dilation's default structure, border value
// 0): out = in | its six face neighbours.  Four consecutive z voxels per thread (the volume is [X][Y][Z], z fastest).
__global__ __launch_bounds__(256) void dilate_step_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int dx, int dy, int dz) {
    const int zq = (dz + 3) / 4;
    const long total = (long)dx * dy * zq;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const int q = (int)(e % zq);
        const long xy = e / zq;
        const int y = (int)(xy % dy), x = (int)(xy / dy);
        const long row = ((long)x * dy + y) * dz;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = 4 * q + k;
            if (z >= dz) break;
            const long at = row + z;
            uint8_t v = in[at];
            if (z > 0) v |= in[at - 1];
            if (z + 1 < dz) v |= in[at + 1];
            if (y > 0) v |= in[at - dz];
            if (y + 1 < dy) v |= in[at + dz];
            if (x > 0) v |= in[at - (long)dy * dz];
            if (x + 1 < dx) v |= in[at + (long)dy * dz];
            out[at] = v ? 1 : 0;
        }
    }
}


// ----------------------------------------------------------------------------- whole-volume preparation (test_scan)
// Element `lin` (C order of [X][Y][Z], z fastest) of a raw volume as it lies in a NIfTI file (Fortran order: x fastest) or in C
// order; any of the scalar types a T1 / mask file holds.  dtype: 0 u8, 1 i16, 2 u16, 3 i32, 4 f32, 5 f64.
template <class T> __device__ __forceinline__ double raw_at(const void* p, size_t i) { return (double)reinterpret_cast<const T*>(p)[i]; }
__device__ __forceinline__ double raw_value(const void* raw, int dtype, size_t i) {
    switch (dtype) {
        case 0: return raw_at<uint8_t>(raw, i);
        case 1: return raw_at<int16_t>(raw, i);
        case 2: return raw_at<uint16_t>(raw, i);
        case 3: return raw_at<int32_t>(raw, i);
        case 4: return raw_at<float>(raw, i);
        default: return raw_at<double>(raw, i);
    }
}
__device__ __forceinline__ size_t file_index(long lin, int dx, int dy, int dz, int fortran) {
    if (!fortran) return (size_t)lin;
    const int z = (int)(lin % dz);
    const long xy = lin / dz;
    const int y = (int)(xy % dy), x = (int)(xy / dy);
    return (size_t)x + (size_t)dx * ((size_t)y + (size_t)dy * z);
}

// Statistics of the non-zero voxels in TWO passes (the volume is 25 MB: a pass is a few microseconds), per-block partials in a
// fixed order (deterministic), in double: PASS 0 count and sum -> mean; PASS 1 the sum of squared deviations about that mean -
// no E[x^2] - mean^2, whose digits go when the mean is large against the spread.
constexpr int kStatBlocks = 1024;
template <int PASS>
__global__ __launch_bounds__(256) void zscore_stats_kernel(const void* __restrict__ raw, int dtype, long n, const double* __restrict__ stat,
                                                          double* __restrict__ part /*[kStatBlocks][2]*/) {
    __shared__ double s_red[4][2];
    const double mean = PASS ? stat[0] : 0.0;
    double a = 0, b = 0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)kStatBlocks * 256L) {
        const double v = raw_value(raw, dtype, (size_t)e);           // the statistics do not depend on the order
        if (v != 0.0) {
            if (PASS == 0) { a += 1.0; b += v; } else { const double d = v - mean; a += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_red[wv][0] = a; s_red[wv][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2)
        part[blockIdx.x * 2 + threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}
// PASS 0: stat[0] = mean, stat[2] = count;  PASS 1: stat[1] = population standard deviation
template <int PASS>
__global__ __launch_bounds__(256) void zscore_final_kernel(const double* __restrict__ part, double* __restrict__ stat) {
    __shared__ double s_red[4][2];
    double a = 0, b = 0;
    for (int k = threadIdx.x; k < kStatBlocks; k += 256) { a += part[k * 2]; b += part[k * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_red[wv][0] = a; s_red[wv][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        b = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
        if (PASS == 0) { stat[0] = a > 0 ? b / a : 0.0; stat[2] = a; }
        else stat[1] = stat[2] > 0 ? sqrt(a / stat[2]) : 0.0;
    }
}
// out[x][y][z] = float((v - mean) / std), in double like the reference's inference path (base.py:358), then one rounding
__global__ __launch_bounds__(256) void zscore_apply_kernel(const void* __restrict__ raw, int dtype, int dx, int dy, int dz, int fortran,
                                                          const double* __restrict__ stat, float* __restrict__ out) {
    const long n = (long)dx * dy * dz;
    const double mean = stat[0], sd = stat[1];
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L)
        out[e] = (float)((raw_value(raw, dtype, file_index(e, dx, dy, dz, fortran)) - mean) / sd);
}
// mask != 0 as 0/1 bytes in C order
__global__ __launch_bounds__(256) void binarize_kernel(const void* __restrict__ raw, int dtype, int dx, int dy, int dz, int fortran, uint8_t* __restrict__ out) {
    const long n = (long)dx * dy * dz;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L)
        out[e] = raw_value(raw, dtype, file_index(e, dx, dy, dz, fortran)) != 0.0 ? 1 : 0;
}
// ROI voxel list in C order (np.nonzero order, base.py:310-331): set voxels per run of kRoiRun consecutive voxels ...
constexpr int kRoiRun = 1024;
__global__ __launch_bounds__(256) void roi_count_kernel(const uint8_t* __restrict__ m, long n, int32_t* __restrict__ counts) {
    __shared__ int s_red[4];
    const long base = blockIdx.x * (long)kRoiRun + threadIdx.x * 4;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n && m[base + k]) ++c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
// ... exclusive scan of the run counts by one workgroup (counts[i] <- offset of run i, counts[nb] <- total) ...
__global__ __launch_bounds__(1024) void roi_scan_kernel(int32_t* __restrict__ counts, int nb) {
    __shared__ long s_part[1024];
    const int per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = min(nb, lo + per);
    long sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { long run = 0; for (int i = 0; i < 1024; ++i) { const long v = s_part[i]; s_part[i] = run; run += v; } counts[nb] = (int32_t)run; }
    __syncthreads();
    long run = s_part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = counts[i]; counts[i] = (int32_t)run; run += v; }
}
// ... and the (x, y, z) rows themselves, each run writing behind its offset
__global__ __launch_bounds__(256) void roi_write_kernel(const uint8_t* __restrict__ m, long n, int dy, int dz, const int32_t* __restrict__ offsets,
                                                       int32_t* __restrict__ centers, long cap) {
    __shared__ int s_scan[256];
    const long base = blockIdx.x * (long)kRoiRun + threadIdx.x * 4;
    bool on[4];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { on[k] = base + k < n && m[base + k]; c += on[k] ? 1 : 0; }
    s_scan[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                            // inclusive Hillis-Steele scan of the 256 thread counts
        const int v = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    long at = (long)offsets[blockIdx.x] + s_scan[threadIdx.x] - c;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (on[k]) {
            const long lin = base + k;
            const int z = (int)(lin % dz);
            const long xy = lin / dz;
            if (at < cap) { centers[at * 3] = (int)(xy / dy); centers[at * 3 + 1] = (int)(xy % dy); centers[at * 3 + 2] = z; }
            ++at;
        }
}

// ----------------------------------------------------------------------------- packed-weights arena
static size_t align64(size_t f) { return (f + 63) / 64 * 64; }

static int alloc_packed(scs_ctx* h) {
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += align64(floats); return o; };
    size_t o_c1w[kTowers], o_c1b[kTowers], o_cw[kTowers][kConvs], o_cb[kTowers][kConvs], o_d1w[kTowers], o_d1b[kTowers];
    for (int t = 0; t < kTowers; ++t) {
        o_c1w[t] = take(kW1Floats); o_c1b[t] = take(1);
        for (int i = 1; i < kConvs; ++i) { o_cw[t][i] = take(conv_pack_floats(i)); o_cb[t][i] = take(n_tiles(kCout[i]) * 16); }
        o_d1w[t] = take((size_t)kFlat * gemm_ns(kD1)); o_d1b[t] = take(gemm_ns(kD1));
    }
    const size_t o_f1w = take((size_t)kFC1 * gemm_ns(kFC1)), o_f1b = take(gemm_ns(kFC1));
    const size_t o_f2w = take((size_t)kFC1ld * gemm_ns(kFC2)), o_f2b = take(gemm_ns(kFC2));
    const size_t o_ow = take(kFC2 * 16), o_ob = take(16);
    h->packed_floats = off;
    SCS_HIP(hipMalloc(&h->packed, off * sizeof(float)));
    float* b = h->packed;
    for (int t = 0; t < kTowers; ++t) {
        h->pk.conv1w[t] = b + o_c1w[t]; h->pk.conv1b[t] = b + o_c1b[t];
        for (int i = 1; i < kConvs; ++i) { h->pk.convw[t][i] = b + o_cw[t][i]; h->pk.convb[t][i] = b + o_cb[t][i]; }
        h->pk.d1w[t] = b + o_d1w[t]; h->pk.d1b[t] = b + o_d1b[t];
    }
    h->pk.fc1w = b + o_f1w; h->pk.fc1b = b + o_f1b;
    h->pk.fc2w = b + o_f2w; h->pk.fc2b = b + o_f2b;
    h->pk.outw = b + o_ow; h->pk.outb = b + o_ob;
    // split operands
    size_t so = 0;
    auto stake = [&](size_t bytes) { size_t o = so; so += (bytes + 255) / 256 * 256; return o; };
    size_t s_w1[kTowers], s_w[kTowers][kConvs], s_b[kTowers][kConvs], s_dw[5], s_db[5];
    for (int t = 0; t < kTowers; ++t) {
        s_w1[t] = stake(kSp12W1Floats * 4);
        for (int i = 1; i < kConvs; ++i) {
            const int KS = 3 * ((3 * kCin[i] + 31) / 32), MT = (kCout[i] + 15) / 16;
            s_w[t][i] = stake((size_t)KS * MT * 2 * 64 * 16 + (i == 1 ? (size_t)kSp12W2StackVec * 16 : 0)); s_b[t][i] = stake((MT * 16 + 1) * 4);
        }
    }
    for (int l = 0; l < 5; ++l) {
        const int KS = l == 4 ? DsFC2::KS : DsK17::KS, MT = l < 3 ? kDsD1Tiles : (l == 3 ? kDsFC1Tiles : kDsFC2Tiles);
        s_dw[l] = stake((size_t)KS * MT * 2 * 64 * 16); s_db[l] = stake((MT * 16 + 1) * 4);
    }
    SCS_HIP(hipMalloc(&h->sp_pack, so));
    for (int t = 0; t < kTowers; ++t) {
        h->sp.w1[t] = reinterpret_cast<const float*>(h->sp_pack + s_w1[t]);
        for (int i = 1; i < kConvs; ++i) {
            h->sp.w[t][i] = reinterpret_cast<const uint4*>(h->sp_pack + s_w[t][i]);
            h->sp.bias[t][i] = reinterpret_cast<const float*>(h->sp_pack + s_b[t][i]);
        }
    }
    for (int l = 0; l < 5; ++l) {
        h->sp.dw[l] = reinterpret_cast<const uint4*>(h->sp_pack + s_dw[l]);
        h->sp.db[l] = reinterpret_cast<const float*>(h->sp_pack + s_db[l]);
    }
    SCS_HIP(hipMalloc(&h->sp_flag, 4 * sizeof(unsigned int)));
    SCS_HIP(hipMemset(h->sp_flag, 0, 4 * sizeof(unsigned int)));
    SCS_HIP(hipHostMalloc(&h->sp_flag_host, 4 * sizeof(unsigned int)));
    memset(h->sp_flag_host, 0, 4 * sizeof(unsigned int));
    return SCS_OK;
}

// rebuild the inference operands from the parameter arena (BN folded, flipped, fragment order)
int repack_inference(scs_ctx* h, hipStream_t st) {
    const float* P = h->params;
    constexpr HeadOff H = head_off();
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        hipLaunchKernelGGL(pack_conv1_kernel, dim3(1), dim3(320), 0, st, P + T.convW[0], P + T.bn[0], 1, h->pk.conv1w[t]);
        for (int i = 1; i < kConvs; ++i)
            hipLaunchKernelGGL(pack_conv_kernel, dim3(32), dim3(256), 0, st, P + T.convW[i], P + T.bn[i], kCin[i],
                               kCout[i], 1, h->pk.convw[t][i], h->pk.convb[t][i]);
        hipLaunchKernelGGL(pack_dense_kernel, dim3(64), dim3(256), 0, st, P + T.d1W, P + T.d1b, kFlat, kD1, kFlat,
                           gemm_ns(kD1), h->pk.d1w[t], h->pk.d1b[t]);
        hipLaunchKernelGGL(pack_split1_kernel, dim3(1), dim3(320), 0, st, P + T.convW[0], P + T.bn[0], const_cast<float*>(h->sp.w1[t]));
        for (int i = 1; i < kConvs; ++i)
            hipLaunchKernelGGL(pack_split_kernel, dim3(1), dim3(256), 0, st, P + T.convW[i], P + T.bn[i], kCin[i], kCout[i],
                               const_cast<uint4*>(h->sp.w[t][i]), const_cast<float*>(h->sp.bias[t][i]), i == 1 ? 1 : 0);
    }
    hipLaunchKernelGGL(pack_dense_kernel, dim3(128), dim3(256), 0, st, P + H.fc1W, P + H.fc1b, kFC1, kFC1, kFC1,
                       gemm_ns(kFC1), h->pk.fc1w, h->pk.fc1b);
    hipLaunchKernelGGL(pack_dense_kernel, dim3(128), dim3(256), 0, st, P + H.fc2W, P + H.fc2b, kFC1 + kAtlas, kFC2,
                       kFC1ld, gemm_ns(kFC2), h->pk.fc2w, h->pk.fc2b);
    hipLaunchKernelGGL(pack_dense_kernel, dim3(8), dim3(256), 0, st, P + H.outW, P + H.outb, kFC2, kClasses, kFC2, 16,
                       h->pk.outw, h->pk.outb);
    // the dense head's split operands (dense_split.h)
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        hipLaunchKernelGGL(pack_dense_split_kernel, dim3(1), dim3(256), 0, st, P + T.d1W, P + T.d1b, kFlat, kD1, DsK17::KS, kDsD1Tiles,
                           const_cast<uint4*>(h->sp.dw[t]), const_cast<float*>(h->sp.db[t]));
    }
    hipLaunchKernelGGL(pack_dense_split_kernel, dim3(1), dim3(256), 0, st, P + H.fc1W, P + H.fc1b, kFC1, kFC1, DsK17::KS, kDsFC1Tiles,
                       const_cast<uint4*>(h->sp.dw[3]), const_cast<float*>(h->sp.db[3]));
    hipLaunchKernelGGL(pack_dense_split_kernel, dim3(1), dim3(256), 0, st, P + H.fc2W, P + H.fc2b, kFC1 + kAtlas, kFC2, DsFC2::KS, kDsFC2Tiles,
                       const_cast<uint4*>(h->sp.dw[4]), const_cast<float*>(h->sp.db[4]));
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

// ----------------------------------------------------------------------------- workspace
// workspace for chunks of up to `need` patches (grown on demand, never beyond the chunk bound: a 64-patch predict
// must not reserve the 4.6 GB a 32768-patch chunk needs)
int ensure_workspace(scs_ctx* h, int need) {
    if (need > h->chunk) need = h->chunk;
    need = (need + 15) / 16 * 16;
    if (h->ws && h->ws_chunk >= need) return SCS_OK;
    SCS_HIP(hipDeviceSynchronize());                       // a previous call may still be using the old buffers
    if (h->ws) { SCS_HIP(hipFree(h->ws)); h->ws = nullptr; }
    if (h->r_base) { SCS_HIP(hipFree(h->r_base)); h->r_base = nullptr; }
    if (h->stage_dev) { SCS_HIP(hipFree(h->stage_dev)); h->stage_dev = nullptr; h->stage_floats = 0; }
    const size_t c = (size_t)need;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += align64(floats); return o; };
    size_t o1[kTowers], o3[kTowers], o2[kTowers], o5[kTowers];
    for (int t = 0; t < kTowers; ++t) {
        o1[t] = take(c * 20 * 196); o3[t] = take(c * 40 * 144); o2[t] = take(c * 40 * 25); o5[t] = take(c * kFlat);
    }
    const size_t of = take(c * kFlat), o_fc1 = take(c * kFC1ld), o_fc2 = take(c * kFC2ld), o_lg = take(c * 16);
    SCS_HIP(hipMalloc(&h->ws, off * sizeof(float)));
    for (int t = 0; t < kTowers; ++t) {
        h->a_pool1[t] = h->ws + o1[t]; h->a_conv3[t] = h->ws + o3[t];
        h->a_pool2[t] = h->ws + o2[t]; h->a_conv5[t] = h->ws + o5[t];
    }
    h->a_feat = h->ws + of; h->a_fc1 = h->ws + o_fc1; h->a_fc2 = h->ws + o_fc2; h->a_logits = h->ws + o_lg;
    h->ws_chunk = need;
    h->r_chunk = 0;                                        // (the split head's rows, if any, were freed above: ensure_split_rows)
    h->last_n = 0;
    return SCS_OK;
}

// rows of the split dense head (dense_split.h) for chunks of the workspace's size: allocated on the first chunk that takes the
// split-head branch - 11 KB per patch (363 MB at the default chunk) that a handle on the f32 kernels, or with split_head = 0, never
// pays for - and zeroed once (pads and K padding are never written afterwards)
static int ensure_split_rows(scs_ctx* h) {
    if (h->r_base && h->r_chunk == h->ws_chunk) return SCS_OK;
    SCS_HIP(hipDeviceSynchronize());                       // a previous call may still be using the old rows
    if (h->r_base) { SCS_HIP(hipFree(h->r_base)); h->r_base = nullptr; }
    const size_t c = (size_t)h->ws_chunk;
    const size_t rb17 = (size_t)ds_row_bytes(DsK17::KP), rb18 = (size_t)ds_row_bytes(DsFC2::KP);
    const size_t tot = c * (4 * rb17 + rb18);
    SCS_HIP(hipMalloc(&h->r_base, tot));
    SCS_HIP(hipMemset(h->r_base, 0, tot));
    SCS_HIP(hipDeviceSynchronize());                       // (the kernels that fill these rows may run on a non-blocking stream)
    for (int t = 0; t < kTowers; ++t) h->r_c5[t] = h->r_base + (size_t)t * c * rb17;
    h->r_feat = h->r_base + 3 * c * rb17;
    h->r_fc1 = h->r_base + 4 * c * rb17;
    h->r_chunk = h->ws_chunk;
    return SCS_OK;
}

// ----------------------------------------------------------------------------- the fusion head on the split kernels
// d1 x3 -> FC1 -> [FC1 | atlas] -> fc_2 on dense_split.h's kernels (rows of halves between the layers), then the f32 out layer +
// softmax + argmax as in forward_head; the towers' block-5 rows are in h->r_c5; marks 5..8 of the profiled forward.
static int forward_head_split(scs_ctx* h, const float* atlas, const float* vol, int dx, int dy, int dz, const int32_t* centers,
                              const float* atlas4d, int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev,
                              int atlas_rows, const int32_t* atlas_index) {
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], st); };
    DenseSplitArgs d{};
    d.n = n; d.flag = h->sp_flag;
    d.mt_total = kDsD1Tiles; d.nfb = 1; d.kp_out = DsK17::KP;
    for (int t = 0; t < kTowers; ++t) {
        d.in[t] = h->r_c5[t]; d.w[t] = h->sp.dw[t]; d.bias[t] = h->sp.db[t]; d.out[t] = h->r_feat;
        d.col0[t] = t * kD1; d.n_store[t] = t == kTowers - 1 ? kD1 + (DsK17::KP - kFC1) : kD1;      // the last tower also clears FC1's K padding
    }
    dense_split_launch<DsK17>(d, kTowers, st);
    mark(5);
    d.mt_total = kDsFC1Tiles; d.nfb = 3; d.kp_out = DsFC2::KP;
    d.in[0] = h->r_feat; d.w[0] = h->sp.dw[3]; d.bias[0] = h->sp.db[3]; d.out[0] = h->r_fc1; d.col0[0] = 0; d.n_store[0] = DsFC2::KP;
    dense_split_launch<DsK17>(d, 1, st);
    if (vol && atlas4d)
        hipLaunchKernelGGL(atlas_gather_split_kernel, dim3((n + 255) / 256), dim3(256), 0, st, atlas4d, centers, dx, dy, dz, h->r_fc1, n);
    else
        hipLaunchKernelGGL(atlas_concat_split_kernel, dim3((n + 255) / 256), dim3(256), 0, st, atlas, atlas_index, atlas_rows, h->r_fc1, n);
    mark(6);
    d.mt_total = kDsFC2Tiles; d.nfb = 3; d.ld_out = kFC2ld;
    d.in[0] = h->r_fc1; d.w[0] = h->sp.dw[4]; d.bias[0] = h->sp.db[4]; d.out[0] = h->a_fc2; d.col0[0] = 0; d.n_store[0] = kFC2ld;
    dense_split_launch<DsFC2>(d, 1, st);
    mark(7);
    hipLaunchKernelGGL(head_out_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->a_fc2, h->pk.outw, h->pk.outb, n,
                       h->a_logits, probs, labels, (const unsigned int*)nullptr);
    mark(8);
    SCS_HIP(hipGetLastError());
    h->last_n = n;
    return SCS_OK;
}

// ----------------------------------------------------------------------------- one chunk of the forward
int forward_chunk(scs_ctx* h, const float* ax, const float* co, const float* sa, const float* atlas,
                  const float* vol, int dx, int dy, int dz, const int32_t* centers, const float* atlas4d,
                  int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev, const X0Src* x0,
                  const int32_t* atlas_index, int mode) {
    const PackedNet& pk = h->pk;
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], st); };
    mark(0);
    const bool split = mode == kFwdSplit;
    const unsigned int* guard = mode == kFwdF32Guarded ? h->sp_flag : nullptr;
    if (!guard) {                         // (a guarded repeat leaves the record of the split pass in front of it alone)
        h->last_split = false;
        h->last_split_head = false;
    }
    if (split && (ax || vol) && !x0) {
        if (h->opt.split_head != 0) { const int rc = ensure_split_rows(h); if (rc != SCS_OK) return rc; }
        // the towers on the f16 matrix cores (conv_split.h); activations between the launches are two f16 planes [pixel][channel]
        // per patch-tower in the f32 path's buffers (same bytes per element); conv5 leaves f32 [n][540] for the head
        Split12Args a12{};
        a12.n = n; a12.flag = h->sp_flag;
        const float* in3[kTowers] = {vol ? vol : ax, co, sa};
        for (int t = 0; t < kTowers; ++t) {
            a12.in[t] = in3[t]; a12.w1[t] = h->sp.w1[t]; a12.w2[t] = h->sp.w[t][1]; a12.b2[t] = h->sp.bias[t][1]; a12.out[t] = h->a_pool1[t];
        }
        a12.centers = centers; a12.dimx = dx; a12.dimy = dy; a12.dimz = dz;
        if (vol) conv12_split_launch<true, true>(a12, st, 170);          // patches cut inside the kernel (scs_forward_volume*)
        else conv12_split_launch<true>(a12, st, 170);
        mark(1);
        SplitArgs sa3{}, sa4{}, sa5{};
        sa3.n = sa4.n = sa5.n = n; sa3.flag = sa4.flag = sa5.flag = h->sp_flag;
        for (int t = 0; t < kTowers; ++t) {
            sa3.in[t] = reinterpret_cast<const unsigned char*>(h->a_pool1[t]); sa3.w[t] = h->sp.w[t][2]; sa3.bias[t] = h->sp.bias[t][2]; sa3.out[t] = h->a_conv3[t];
            sa4.in[t] = reinterpret_cast<const unsigned char*>(h->a_conv3[t]); sa4.w[t] = h->sp.w[t][3]; sa4.bias[t] = h->sp.bias[t][3]; sa4.out[t] = h->a_pool2[t];
            sa5.in[t] = reinterpret_cast<const unsigned char*>(h->a_pool2[t]); sa5.w[t] = h->sp.w[t][4]; sa5.bias[t] = h->sp.bias[t][4]; sa5.out[t] = h->a_conv5[t];
        }
        conv_split_all_launch<SpConv3>(sa3, st, 85);
        mark(2);
        conv_split_all_launch<SpConv4>(sa4, st, 85);
        mark(3);
        const bool shead = h->opt.split_head != 0;
        if (shead) {
            for (int t = 0; t < kTowers; ++t) sa5.out[t] = h->r_c5[t];
            sa5.out_kp = DsK17::KP;
            conv_split_launch<SpConv5R>(sa5, st, 170);
        } else conv_split_launch<SpConv5>(sa5, st, 170);
        mark(4);
        h->last_split = true;
        h->last_split_head = shead;
        if (shead) return forward_head_split(h, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, ev, n, atlas_index);
        const float* c5s[kTowers] = {h->a_conv5[0], h->a_conv5[1], h->a_conv5[2]};
        return forward_head(h, c5s, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, ev, n, atlas_index);
    }
    ConvArgs a{};
    a.n = n; a.guard = guard;
    // conv1 + conv2 + pool1
    for (int t = 0; t < kTowers; ++t) {
        a.w[t] = pk.convw[t][1]; a.bias[t] = pk.convb[t][1]; a.out[t] = h->a_pool1[t];
        a.w1[t] = pk.conv1w[t];
    }
    if (x0) {
        a.x0 = *x0;
        conv_launch<CfgConv12X>(a, st);
    } else if (vol) {
        a.in[0] = a.in[1] = a.in[2] = vol;
        a.centers = centers; a.dimx = dx; a.dimy = dy; a.dimz = dz;
        if (guard) conv_launch_guarded<CfgConv12V>(a, st); else conv_launch<CfgConv12V>(a, st);
    } else {
        a.in[0] = ax; a.in[1] = co; a.in[2] = sa;
        if (guard) conv_launch_guarded<CfgConv12>(a, st); else conv_launch<CfgConv12>(a, st);
    }
    mark(1);
    if (h->opt.fwd_fused > 0 && !guard) {        // (a guarded repeat takes the three launches below: the same bits)
        // conv3 -> conv4 + pool2 -> conv5 as ONE kernel per patch-tower (conv_fused.h): the two activations in between stay in LDS;
        // same bits as the three launches below.  An OPTION (fwd_fused = 1), not the default: it cuts the net's HBM traffic from
        // ~280 to ~113 KB per patch but streams every layer's weights through LDS once per patch-tower and runs 20 % slower than
        // the three launches (profiles/NOTES_r04.md).  The profiled forward books the whole kernel under conv3.
        FusedArgs f{};
        f.n = n;
        for (int t = 0; t < kTowers; ++t) {
            f.in[t] = h->a_pool1[t]; f.out[t] = h->a_conv5[t];
            for (int l = 0; l < 3; ++l) { f.w[t][l] = pk.convw[t][2 + l]; f.bias[t][l] = pk.convb[t][2 + l]; }
            f.dbg3[t] = h->opt.fwd_fused == 2 ? h->a_conv3[t] : nullptr;
            f.dbg4[t] = h->opt.fwd_fused == 2 ? h->a_pool2[t] : nullptr;
        }
        conv345_fused_launch(f, st);
        mark(2); mark(3); mark(4);
        const float* c5f[kTowers] = {h->a_conv5[0], h->a_conv5[1], h->a_conv5[2]};
        return forward_head(h, c5f, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, ev, x0 ? x0->n_samples : n, atlas_index);
    }
    // conv3
    for (int t = 0; t < kTowers; ++t) { a.in[t] = h->a_pool1[t]; a.w[t] = pk.convw[t][2]; a.bias[t] = pk.convb[t][2]; a.out[t] = h->a_conv3[t]; }
    if (guard) conv_launch_guarded<CfgConv3>(a, st); else conv_launch<CfgConv3>(a, st);
    mark(2);
    // conv4 + pool2
    for (int t = 0; t < kTowers; ++t) { a.in[t] = h->a_conv3[t]; a.w[t] = pk.convw[t][3]; a.bias[t] = pk.convb[t][3]; a.out[t] = h->a_pool2[t]; }
    if (guard) conv_launch_guarded<CfgConv4>(a, st); else conv_launch<CfgConv4>(a, st);
    mark(3);
    // conv5
    for (int t = 0; t < kTowers; ++t) { a.in[t] = h->a_pool2[t]; a.w[t] = pk.convw[t][4]; a.bias[t] = pk.convb[t][4]; a.out[t] = h->a_conv5[t]; }
    if (guard) conv_launch_guarded<CfgConv5>(a, st); else conv_launch<CfgConv5>(a, st);
    mark(4);
    const float* c5[kTowers] = {h->a_conv5[0], h->a_conv5[1], h->a_conv5[2]};
    return forward_head(h, c5, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, ev, x0 ? x0->n_samples : n, atlas_index, guard);
}

// One chunk with option fwd_split, entirely on the stream (nothing here waits for the device): the split kernels; then the f32 kernels
// of the same chunk in their guarded forms - every workgroup of them reads the overflow flag the split kernels may have raised
// (conv_split.h: a scaled activation beyond the f16 range) and leaves at once while it is clear, so a flagged chunk is done again on
// the exact-f32 kernels (the bits scs_forward gives without the option) and a clean one costs ten near-empty launches; then one
// thread counts the repeat, records whether this chunk's buffers now hold the f32 layouts (scs_get_activation) and clears the flag
// for the next chunk.
__global__ void split_settle_kernel(unsigned int* __restrict__ words) {
    const unsigned int f = words[0];
    words[1] = f ? 1u : 0u;
    if (f) { words[2] += 1u; words[0] = 0u; }
}
static int forward_chunk_split(scs_ctx* h, const float* ax, const float* co, const float* sa, const float* atlas,
                               const float* vol, int dx, int dy, int dz, const int32_t* centers, const float* atlas4d,
                               int n, float* probs, int32_t* labels, hipStream_t st) {
    int rc = forward_chunk(h, ax, co, sa, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, nullptr, nullptr, nullptr, kFwdSplit);
    if (rc != SCS_OK) return rc;
    rc = forward_chunk(h, ax, co, sa, atlas, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, st, nullptr, nullptr, nullptr, kFwdF32Guarded);
    if (rc != SCS_OK) return rc;
    hipLaunchKernelGGL(split_settle_kernel, dim3(1), dim3(1), 0, st, h->sp_flag);
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

// The fusion head of one chunk: c5[t] = the towers' flattened block-5 outputs [n][540] (nets.py:176-231: d1 x3, FC1 on their
// concatenation, fc_2 on [FC1 | atlas prior], out + softmax); marks 5..8 of the profiled forward.
int forward_head(scs_ctx* h, const float* const c5[kTowers], const float* atlas, const float* vol, int dx, int dy, int dz,
                 const int32_t* centers, const float* atlas4d, int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev,
                 int atlas_rows, const int32_t* atlas_index, const unsigned int* guard) {
    const PackedNet& pk = h->pk;
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], st); };
    auto gemm = [&](const GemmArgs& g, int np) { if (guard) gemm_launch_guarded(g, np, st); else gemm_launch(g, np, st); };
    // d1 x3 -> feat[:, 180 t : 180 t + 180]
    GemmArgs g{};
    g.M = n; g.K4 = kFlat; g.N = kD1; g.relu = 1; g.guard = guard;
    for (int t = 0; t < kTowers; ++t) g.p[t] = GemmProblem{c5[t], kFlat, pk.d1w[t], gemm_ns(kD1), pk.d1b[t], h->a_feat + t * kD1, kFlat};
    gemm(g, kTowers);
    mark(5);
    // FC1 -> fc1[:, 0:540]; atlas -> fc1[:, 540:556]
    g.K4 = kFC1; g.N = kFC1;
    g.p[0] = GemmProblem{h->a_feat, kFlat, pk.fc1w, gemm_ns(kFC1), pk.fc1b, h->a_fc1, kFC1ld};
    gemm(g, 1);
    if (vol && atlas4d)
        hipLaunchKernelGGL(atlas_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, atlas4d, centers, dx, dy, dz, h->a_fc1, n, guard);
    else
        hipLaunchKernelGGL(atlas_concat_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, st, atlas, atlas_index, atlas_rows, h->a_fc1, n, guard);
    mark(6);
    // fc_2
    g.K4 = kFC1ld; g.N = kFC2;
    g.p[0] = GemmProblem{h->a_fc1, kFC1ld, pk.fc2w, gemm_ns(kFC2), pk.fc2b, h->a_fc2, kFC2ld};
    gemm(g, 1);
    mark(7);
    // out + softmax + argmax
    hipLaunchKernelGGL(head_out_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->a_fc2, pk.outw, pk.outb, n,
                       h->a_logits, probs, labels, guard);
    mark(8);
    SCS_HIP(hipGetLastError());
    h->last_n = n;
    return SCS_OK;
}

// ----------------------------------------------------------------------------- whole subject with shared tower work (dense_eval.h)
static int grow(float** buf, size_t* have, size_t need) {
    if (*buf && *have >= need) return SCS_OK;
    SCS_HIP(hipDeviceSynchronize());                       // a previous call may still be using the old buffer
    if (*buf) { SCS_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
    SCS_HIP(hipMalloc(buf, need * sizeof(float)));
    *have = need;
    return SCS_OK;
}

constexpr size_t kDenseBudgetFloats = (size_t)192 << 20;   // tile buffers of one slice batch: at most 768 MB

// towers of all n ROI voxels -> h->dense_feat [3][n][540]
int dense_towers(scs_ctx* h, const float* vol, int dx, int dy, int dz, const int32_t* centers, int64_t n, const int32_t* box, hipStream_t st) {
    const PackedNet& pk = h->pk;
    int rc = grow(&h->dense_feat, &h->dense_feat_floats, (size_t)kTowers * n * kFlat);
    if (rc != SCS_OK) return rc;
    for (int v = 0; v < kTowers; ++v) {
        const DenseView V = dense_view(v, box);
        const size_t per_slice = dense_buffers(V, 1).total;
        const int sb = (int)std::max<size_t>(1, std::min<size_t>((size_t)V.S, kDenseBudgetFloats / per_slice));
        const DenseBuffers B = dense_buffers(V, sb);
        rc = grow(&h->dense_ws, &h->dense_ws_floats, B.total);
        if (rc != SCS_OK) return rc;
        float* W = h->dense_ws;
        int32_t* ctr = reinterpret_cast<int32_t*>(W + B.ctr);
        float* feat = h->dense_feat + (size_t)v * n * kFlat;
        for (int s0 = 0; s0 < V.S; s0 += sb) {
            const int cur = std::min(sb, V.S - s0);
            ConvArgs a{};
            a.tower0 = v;
            // conv1 + conv2 + pool1 on the 32x32 windows of the pool1 tiles
            a.n = V.n1(cur);
            hipLaunchKernelGGL(dense_centres_kernel, dense_grid(a.n), dim3(256), 0, st, V, s0, cur, ctr);
            a.in[v] = vol; a.centers = ctr; a.dimx = dx; a.dimy = dy; a.dimz = dz;
            a.w[v] = pk.convw[v][1]; a.bias[v] = pk.convb[v][1]; a.w1[v] = pk.conv1w[v]; a.out[v] = W + B.t1;
            conv_launch<CfgConv12V>(a, st, 1);
            // conv3 on 14x14 tiles every 12
            hipLaunchKernelGGL((dense_retile_kernel<20, 14, 14, 12>), dense_tile_grid(V.n3(cur)), dim3(256), 0, st,
                               W + B.t1, V.t1r, V.t1c, W + B.i3, V.t3r, V.t3c, cur * 4);
            a.n = V.n3(cur); a.centers = nullptr;
            a.in[v] = W + B.i3; a.w[v] = pk.convw[v][2]; a.bias[v] = pk.convb[v][2]; a.out[v] = W + B.o3;
            conv_launch<CfgConv3>(a, st, 1);
            // conv4 on 12x12 tiles every 10
            hipLaunchKernelGGL((dense_retile_kernel<40, 12, 12, 10>), dense_tile_grid(V.n4(cur)), dim3(256), 0, st,
                               W + B.o3, V.t3r, V.t3c, W + B.i4, V.t4r, V.t4c, cur * 4);
            a.n = V.n4(cur);
            a.in[v] = W + B.i4; a.w[v] = pk.convw[v][3]; a.bias[v] = pk.convb[v][3]; a.out[v] = W + B.o4;
            conv_launch<CfgConv4NP>(a, st, 1);
            // pool2 (four window origins) and conv5 on 5x5 tiles every 3
            hipLaunchKernelGGL(dense_pool_retile_kernel, dense_tile_grid(V.n5(cur)), dim3(256), 0, st,
                               W + B.o4, V.t4r, V.t4c, W + B.i5, V.t5r, V.t5c, cur);
            a.n = V.n5(cur);
            a.in[v] = W + B.i5; a.w[v] = pk.convw[v][4]; a.bias[v] = pk.convb[v][4]; a.out[v] = W + B.o5;
            conv_launch<CfgConv5>(a, st, 1);
            // the 3x3 windows of the voxels of these slices
            hipLaunchKernelGGL(dense_features_kernel, dense_grid(n * 60), dim3(256), 0, st, V, s0, cur, W + B.o5, centers, (long)n, feat);
        }
    }
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

static int set_kernel_attrs() {
    SCS_HIP(conv_set_attr<CfgConv12>());
    SCS_HIP(conv_set_attr<CfgConv12V>());
    SCS_HIP(conv_set_attr<CfgConv12X>());
    SCS_HIP(conv_set_attr<CfgConv3>());
    SCS_HIP(conv_set_attr<CfgConv4>());
    SCS_HIP(conv_set_attr<CfgConv5>());
    SCS_HIP(conv345_fused_set_attr());
    SCS_HIP(conv_set_attr<CfgConv4NP>());
    SCS_HIP(conv12_split_set_attr());
    SCS_HIP(conv_split_all_set_attr<SpConv3>());
    SCS_HIP(conv_split_all_set_attr<SpConv4>());
    SCS_HIP(conv_split_set_attr<SpConv5>());
    SCS_HIP(conv_split_set_attr<SpConv5R>());
    SCS_HIP(dense_split_set_attr<DsK17>());
    SCS_HIP(dense_split_set_attr<DsFC2>());
    SCS_HIP(gemm_set_attrs());
    // the guarded forms of the f32 inference kernels (forward_chunk_split)
    SCS_HIP(conv_set_attr_guarded<CfgConv12>());
    SCS_HIP(conv_set_attr_guarded<CfgConv12V>());
    SCS_HIP(conv_set_attr_guarded<CfgConv3>());
    SCS_HIP(conv_set_attr_guarded<CfgConv4>());
    SCS_HIP(conv_set_attr_guarded<CfgConv5>());
    SCS_HIP(gemm_set_attrs_guarded());
    return SCS_OK;
}

}  // namespace scs

using namespace scs;

// =============================================================================== C ABI
extern "C" {

const char* scs_last_error(void) { return g_err; }
int scs_abi_version(void) { return 1; }
const char* scs_target_arch(void) { return "gfx950"; }

int scs_create(int device, scs_handle* out) {
    SCS_ARG(out != nullptr, "out handle is NULL");
    *out = nullptr;
    int ndev = 0;
    SCS_HIP(hipGetDeviceCount(&ndev));
    SCS_ARG(device >= 0 && device < ndev, "device index out of range");
    SCS_HIP(hipSetDevice(device));
    scs_ctx* h = new (std::nothrow) scs_ctx();
    SCS_ARG(h != nullptr, "out of host memory");
    h->device = device;
    {   // the behaviour switches of the training step: the environment is read here, once per handle
        auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        scs_options& o = h->opt;
        o.fin_max_batch = env_int("SCS_FIN_MAX_BATCH", -1);
        o.bn1_fused_max_batch = env_int("SCS_BN1_FUSED_MAX_BATCH", -1);
        o.dense_chain_max_batch = env_int("SCS_DENSE_CHAIN_MAX_BATCH", -1);
        o.c5_fused_max_batch = env_int("SCS_C5_FUSED_MAX_BATCH", -1);
        o.train_one_stream = getenv("SCS_TRAIN_ONE_STREAM") != nullptr;
        o.tail_side = getenv("SCS_TAIL_SIDE") != nullptr;
        o.tail_fused = env_int("SCS_TAIL_FUSED", -1);
        o.split_max_batch = env_int("SCS_SPLIT_MAX_BATCH", -1);
        o.fwd_fused = env_int("SCS_FWD_FUSED", -1);
        o.fwd_split = env_int("SCS_FWD_SPLIT", -1);
        o.train_split = env_int("SCS_TRAIN_SPLIT", -1);
        o.split_head = env_int("SCS_SPLIT_HEAD", -1);
        o.wgfin_early = env_int("SCS_WGFIN_EARLY", 1);
        o.c5_split = env_int("SCS_C5_SPLIT", -1);
        if (const char* e = getenv("SCS_WG_SLICES")) {
            int x[4];
            if (sscanf(e, "%d,%d,%d,%d", &x[0], &x[1], &x[2], &x[3]) == 4)
                for (int i = 0; i < 4; ++i) if (x[i] >= 1 && x[i] <= 256) o.wg_slices[i] = x[i];
        }
    }
    int rc = set_kernel_attrs();
    if (rc != SCS_OK) { delete h; return rc; }
    const size_t bytes = (size_t)SCS_N_PARAMS * sizeof(float);
    hipError_t e = hipMalloc(&h->params, bytes);
    if (e == hipSuccess) e = hipMalloc(&h->grads, bytes);
    if (e == hipSuccess) e = hipMalloc(&h->adam_m, bytes);
    if (e == hipSuccess) e = hipMalloc(&h->adam_v, bytes);
    if (e != hipSuccess) { set_error("hipMalloc of the arenas failed: %s", hipGetErrorString(e)); scs_destroy(h); return SCS_ERR_HIP; }
    (void)hipMemset(h->grads, 0, bytes);
    (void)hipMemset(h->adam_m, 0, bytes);
    (void)hipMemset(h->adam_v, 0, bytes);
    rc = alloc_packed(h);
    if (rc != SCS_OK) { scs_destroy(h); return rc; }
    *out = h;
    return SCS_OK;
}

int scs_destroy(scs_handle h) {
    if (!h) return SCS_OK;
    (void)hipSetDevice(h->device);
    train_free(h);
    if (h->params) (void)hipFree(h->params);
    if (h->grads) (void)hipFree(h->grads);
    if (h->adam_m) (void)hipFree(h->adam_m);
    if (h->adam_v) (void)hipFree(h->adam_v);
    if (h->packed) (void)hipFree(h->packed);
    if (h->sp_pack) (void)hipFree(h->sp_pack);
    if (h->sp_flag) (void)hipFree(h->sp_flag);
    if (h->sp_flag_host) (void)hipHostFree(h->sp_flag_host);
    if (h->ws) (void)hipFree(h->ws);
    if (h->r_base) (void)hipFree(h->r_base);
    if (h->dense_ws) (void)hipFree(h->dense_ws);
    if (h->dense_feat) (void)hipFree(h->dense_feat);
    if (h->box_flag) (void)hipFree(h->box_flag);
    if (h->box_flag_host) (void)hipHostFree(h->box_flag_host);
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    if (h->last_train_event) (void)hipEventDestroy(h->last_train_event);
    if (h->host_compute) (void)hipStreamDestroy(h->host_compute);
    if (h->host_copy) (void)hipStreamDestroy(h->host_copy);
    delete h;
    return SCS_OK;
}

int scs_set_params(scs_handle h, const float* host_arena, size_t n) {
    SCS_ARG(h && host_arena, "null handle or arena");
    SCS_ARG(n == (size_t)SCS_N_PARAMS, "arena must hold SCS_N_PARAMS floats");
    SCS_HIP(hipSetDevice(h->device));
    SCS_HIP(hipMemcpy(h->params, host_arena, n * sizeof(float), hipMemcpyHostToDevice));
    SCS_HIP(hipMemset(h->adam_m, 0, n * sizeof(float)));
    SCS_HIP(hipMemset(h->adam_v, 0, n * sizeof(float)));
    h->adam_t = 0;
    int rc = repack_inference(h, nullptr);
    if (rc != SCS_OK) return rc;
    SCS_HIP(hipDeviceSynchronize());
    h->packed_dirty = false;
    h->has_params = true;
    return SCS_OK;
}

int scs_get_params(scs_handle h, float* host_arena, size_t n) {
    SCS_ARG(h && host_arena, "null handle or arena");
    SCS_ARG(n == (size_t)SCS_N_PARAMS, "arena must hold SCS_N_PARAMS floats");
    if (!h->has_params) { set_error("no parameters loaded"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    SCS_HIP(hipDeviceSynchronize());
    SCS_HIP(hipMemcpy(host_arena, h->params, n * sizeof(float), hipMemcpyDeviceToHost));
    return SCS_OK;
}

static int* option_slot(scs_ctx* h, const char* name) {
    scs_options& o = h->opt;
    struct { const char* n; int* p; } tab[] = {
        {"fin_max_batch", &o.fin_max_batch}, {"bn1_fused_max_batch", &o.bn1_fused_max_batch},
        {"dense_chain_max_batch", &o.dense_chain_max_batch}, {"c5_fused_max_batch", &o.c5_fused_max_batch},
        {"train_one_stream", &o.train_one_stream}, {"tail_side", &o.tail_side}, {"tail_fused", &o.tail_fused},
        {"split_max_batch", &o.split_max_batch},
        {"fwd_fused", &o.fwd_fused}, {"fwd_split", &o.fwd_split}, {"split_fallbacks", &o.split_fallbacks}, {"train_split", &o.train_split}, {"split_head", &o.split_head},
        {"wgfin_early", &o.wgfin_early},
        {"c5_split", &o.c5_split},
        {"wg_slices_conv2", &o.wg_slices[0]}, {"wg_slices_conv3", &o.wg_slices[1]}, {"wg_slices_conv4", &o.wg_slices[2]},
        {"wg_slices_conv5", &o.wg_slices[3]}};
    for (auto& t : tab) if (strcmp(t.n, name) == 0) return t.p;
    return nullptr;
}

int scs_set_option(scs_handle h, const char* name, int64_t value) {
    SCS_ARG(h && name, "null handle or option name");
    int* slot = option_slot(h, name);
    SCS_ARG(slot != nullptr, "unknown option name");
    SCS_ARG(value >= -1 && value <= (1 << 20), "option value out of range");
    *slot = (int)value;
    if (slot == &h->opt.split_fallbacks) {                 // the counter lives on the device (sp_flag[2]): set it behind everything enqueued
        SCS_HIP(hipSetDevice(h->device));
        SCS_HIP(hipDeviceSynchronize());
        const unsigned int v = value > 0 ? (unsigned int)value : 0u;
        SCS_HIP(hipMemcpy(h->sp_flag + 2, &v, sizeof(v), hipMemcpyHostToDevice));
    }
    return SCS_OK;
}

int scs_get_option(scs_handle h, const char* name, int64_t* value) {
    SCS_ARG(h && name && value, "null handle, option name or output");
    int* slot = option_slot(h, name);
    SCS_ARG(slot != nullptr, "unknown option name");
    if (slot == &h->opt.split_fallbacks) {                 // counted on the device by the chunks' settle launches: read behind everything enqueued
        SCS_HIP(hipSetDevice(h->device));
        SCS_HIP(hipDeviceSynchronize());
        unsigned int v = 0;
        SCS_HIP(hipMemcpy(&v, h->sp_flag + 2, sizeof(v), hipMemcpyDeviceToHost));
        *slot = (int)v;
    }
    *value = *slot;
    return SCS_OK;
}

// ---- scs_probe_f32_mfma[_clock]: issue rate of v_mfma_f32_16x16x4_f32 (independent accumulators, operands in registers), and the
// shader clock the device holds while it runs them: one lane per workgroup stamps s_memtime (tick = shader cycle) and s_memrealtime
// (the constant 100 MHz counter) around the loop - clock = d memtime / d memrealtime x 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" (6))
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, unsigned long long* stamps /*[workgroup][2]: d memtime, d memrealtime*/, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 12345.678f) out[blockIdx.x] = s;                    // keeps the accumulators alive; never true for these operands
}

int scs_probe_f32_mfma_clock(scs_handle h, double* tflops, double* shader_ghz) {
    SCS_ARG(h && tflops, "null handle or output");
    SCS_HIP(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    SCS_HIP(hipGetDeviceProperties(&prop, h->device));
    const int cus = prop.multiProcessorCount, wgs = cus * 4, iters = 8000;     // 4 workgroups of 4 waves per CU: four waves per SIMD; ~4 ms per launch
    float* out = nullptr;
    unsigned long long* stamps = nullptr;
    SCS_HIP(hipMalloc(&out, wgs * sizeof(float)));
    hipError_t err = hipMalloc(&stamps, (size_t)wgs * 2 * sizeof(unsigned long long));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    for (int rep = 0; rep < 3 && err == hipSuccess; ++rep) {                     // the last of three launches is the one timed
        err = hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(wgs), dim3(256), 0, nullptr, out, stamps, iters);
        if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> st;
    if (err == hipSuccess && shader_ghz) {
        try { st.resize((size_t)wgs * 2); } catch (...) { err = hipErrorOutOfMemory; }
        if (err == hipSuccess) err = hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (stamps) (void)hipFree(stamps);
    if (err != hipSuccess || !(ms > 0.f)) { set_error("scs_probe_f32_mfma: %s", hipGetErrorString(err)); return SCS_ERR_HIP; }
    *tflops = (double)wgs * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;         // 16x16x4 = 1024 MACs per wave-instruction
    if (shader_ghz) {
        std::vector<double> ghz;
        for (int i = 0; i < wgs; ++i) if (st[2 * i + 1] > 0) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        *shader_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];                   // median over the workgroups
    }
    return SCS_OK;
}

int scs_probe_f32_mfma(scs_handle h, double* tflops) { return scs_probe_f32_mfma_clock(h, tflops, nullptr); }

int scs_set_chunk(scs_handle h, int patches_per_chunk) {
    if (!h) return SCS_ERR_ARG;
    if (patches_per_chunk >= 16 && patches_per_chunk <= (1 << 20)) h->chunk = (patches_per_chunk + 15) / 16 * 16;
    return h->chunk;
}

// split n patches into equal chunks no larger than the workspace chunk (a short tail chunk would leave the
// dense-head GEMMs with too few workgroups to fill 256 CUs)
static int balanced_chunk(int64_t n, int max_chunk) {
    const int64_t k = (n + max_chunk - 1) / max_chunk;
    int64_t c = ((n + k - 1) / k + 15) / 16 * 16;
    return (int)(c > max_chunk ? max_chunk : c);
}

// (scs_forward_profiled only: the timed split pass has no f32 repeat behind it) wait for the stream and read (and clear) the overflow
// flag: 1 = set, 0 = clean, < 0 = error
static int split_overflowed(scs_ctx* h, hipStream_t st) {
    SCS_HIP(hipMemcpyAsync(h->sp_flag_host, h->sp_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    SCS_HIP(hipStreamSynchronize(st));
    if (*h->sp_flag_host == 0) return 0;
    SCS_HIP(hipMemsetAsync(h->sp_flag, 0, sizeof(unsigned int), st));
    return 1;
}

static int forward_common(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas,
                          const float* vol, int dx, int dy, int dz, const int32_t* centers, const float* atlas4d,
                          int64_t n, float* probs, int32_t* labels, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    if (!h->has_params) { set_error("scs_forward before scs_set_params"); return SCS_ERR_STATE; }
    if (n == 0) return SCS_OK;
    SCS_HIP(hipSetDevice(h->device));
    const int step = balanced_chunk(n, h->chunk);
    int rc = ensure_workspace(h, step);
    if (rc != SCS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->packed_dirty) { rc = repack_inference(h, st); if (rc != SCS_OK) return rc; h->packed_dirty = false; }
    // fwd_split: every chunk on the split kernels with its guarded f32 repeat behind it (forward_chunk_split) - the call enqueues
    // and returns like the f32 form.  The flag word is cleared in front of the first chunk on THIS stream (a call that failed half
    // way, or the profiled entry, may have left it set).
    const bool split = h->opt.fwd_split > 0;
    if (split) SCS_HIP(hipMemsetAsync(h->sp_flag, 0, sizeof(unsigned int), st));
    for (int64_t s = 0; s < n; s += step) {
        const int m = (int)((n - s < step) ? (n - s) : step);
        const float *cax = ax ? ax + s * 1024 : nullptr, *cco = co ? co + s * 1024 : nullptr, *csa = sa ? sa + s * 1024 : nullptr;
        const float* cat = atlas ? atlas + s * 15 : nullptr;
        const int32_t* cc = centers ? centers + s * 3 : nullptr;
        float* cp = probs ? probs + s * 15 : nullptr;
        int32_t* cl = labels ? labels + s : nullptr;
        rc = split ? forward_chunk_split(h, cax, cco, csa, cat, vol, dx, dy, dz, cc, atlas4d, m, cp, cl, st)
                   : forward_chunk(h, cax, cco, csa, cat, vol, dx, dy, dz, cc, atlas4d, m, cp, cl, st, nullptr, nullptr, nullptr, kFwdF32);
        if (rc != SCS_OK) return rc;
    }
    return SCS_OK;
}

int scs_forward(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, int64_t n,
                float* probs, int32_t* labels, void* stream) {
    SCS_ARG(n == 0 || (ax && co && sa && atlas), "null input pointer");
    return forward_common(h, ax, co, sa, atlas, nullptr, 0, 0, 0, nullptr, nullptr, n, probs, labels, stream);
}

int scs_forward_volume(scs_handle h, const float* vol, int dx, int dy, int dz, const int32_t* centers,
                       const float* atlas4d, int64_t n, float* probs, int32_t* labels, void* stream) {
    SCS_ARG(n == 0 || (vol && centers && atlas4d), "null input pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0, "volume dims must be positive");
    return forward_common(h, nullptr, nullptr, nullptr, nullptr, vol, dx, dy, dz, centers, atlas4d, n, probs, labels, stream);
}

int scs_forward_volume_rows(scs_handle h, const float* vol, int dx, int dy, int dz, const int32_t* centers,
                            const float* atlas_rows, int64_t n, float* probs, int32_t* labels, void* stream) {
    SCS_ARG(n == 0 || (vol && centers && atlas_rows), "null input pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0, "volume dims must be positive");
    return forward_common(h, nullptr, nullptr, nullptr, atlas_rows, vol, dx, dy, dz, centers, nullptr, n, probs, labels, stream);
}

// centres outside the caller's box (xmin, xmax, ymin, ymax, zmin, zmax): the shared-tower path would skip them
__global__ void box_check_kernel(const int32_t* __restrict__ centers, long n, int x0, int x1, int y0, int y1, int z0, int z1,
                                 unsigned int* __restrict__ outside) {
    unsigned int bad = 0;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        const int x = centers[3 * k], y = centers[3 * k + 1], z = centers[3 * k + 2];
        bad += (x < x0 || x > x1 || y < y0 || y > y1 || z < z0 || z > z1) ? 1u : 0u;
    }
    if (bad) atomicAdd(outside, bad);
}

int scs_forward_volume_shared(scs_handle h, const float* vol, int dx, int dy, int dz, const int32_t* centers,
                              const float* atlas_rows, int64_t n, const int32_t* box, float* probs, int32_t* labels, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    SCS_ARG(n == 0 || (vol && centers && atlas_rows && box), "null input pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0, "volume dims must be positive");
    if (!h->has_params) { set_error("scs_forward_volume_shared before scs_set_params"); return SCS_ERR_STATE; }
    if (n == 0) return SCS_OK;
    for (int a = 0; a < 3; ++a) {
        SCS_ARG(box[2 * a] <= box[2 * a + 1], "box: min > max");
        SCS_ARG((int64_t)box[2 * a + 1] - box[2 * a] < 4096, "box: extent too large");
    }
    SCS_HIP(hipSetDevice(h->device));
    const int step = balanced_chunk(n, h->chunk);
    int rc = ensure_workspace(h, step);
    if (rc != SCS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->packed_dirty) { rc = repack_inference(h, st); if (rc != SCS_OK) return rc; h->packed_dirty = false; }
    // every centre must lie inside the box (the tile maps cover nothing else): counted on the device in front of the work, read
    // back behind it - a centre outside makes the call fail with SCS_ERR_ARG instead of leaving a stale row of results
    if (!h->box_flag) {
        SCS_HIP(hipMalloc(&h->box_flag, sizeof(unsigned int)));
        SCS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->box_flag_host), sizeof(unsigned int), hipHostMallocDefault));
    }
    SCS_HIP(hipMemsetAsync(h->box_flag, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(box_check_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, centers, (long)n,
                       box[0], box[1], box[2], box[3], box[4], box[5], h->box_flag);
    rc = dense_towers(h, vol, dx, dy, dz, centers, n, box, st);
    if (rc != SCS_OK) return rc;
    for (int64_t s = 0; s < n; s += step) {
        const int m = (int)((n - s < step) ? (n - s) : step);
        const float* c5[kTowers];
        for (int t = 0; t < kTowers; ++t) c5[t] = h->dense_feat + ((size_t)t * n + s) * kFlat;
        rc = forward_head(h, c5, atlas_rows + s * 15, nullptr, 0, 0, 0, nullptr, nullptr, m, probs ? probs + s * 15 : nullptr,
                          labels ? labels + s : nullptr, st, nullptr, m, nullptr);
        if (rc != SCS_OK) return rc;
    }
    SCS_HIP(hipMemcpyAsync(h->box_flag_host, h->box_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    SCS_HIP(hipStreamSynchronize(st));
    if (*h->box_flag_host != 0) {
        set_error("scs_forward_volume_shared: %u of %lld centres lie outside the box; the results are invalid", *h->box_flag_host, (long long)n);
        return SCS_ERR_ARG;
    }
    return SCS_OK;
}

// Host-pointer entry.  The chunks are double buffered on the device: a helper thread uploads chunk i+1 on the
// handle's copy stream (the runtime stages pageable memory and blocks the thread that issued the copy, not the device)
// while the kernels of chunk i run on the compute stream; results come back per chunk.
int scs_forward_host(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, int64_t n,
                     float* probs, int32_t* labels) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    SCS_ARG(n == 0 || (ax && co && sa && atlas), "null input pointer");
    if (!h->has_params) { set_error("scs_forward_host before scs_set_params"); return SCS_ERR_STATE; }
    if (n == 0) return SCS_OK;
    SCS_HIP(hipSetDevice(h->device));
    const int hstep = balanced_chunk(n, h->chunk);
    int rc = ensure_workspace(h, hstep);
    if (rc != SCS_OK) return rc;
    if (!h->host_compute) {
        SCS_HIP(hipStreamCreateWithFlags(&h->host_compute, hipStreamNonBlocking));
        SCS_HIP(hipStreamCreateWithFlags(&h->host_copy, hipStreamNonBlocking));
    }
    hipStream_t sc = h->host_compute, su = h->host_copy;
    if (h->packed_dirty) {
        // the parameter arena may still be written by training work the caller enqueued on a stream of theirs
        if (h->last_train_event) SCS_HIP(hipStreamWaitEvent(sc, h->last_train_event, 0));
        rc = repack_inference(h, sc);
        if (rc != SCS_OK) return rc;
        h->packed_dirty = false;
    }
    const size_t c = (size_t)h->ws_chunk;
    const size_t per = 3 * 1024 + 15 + 15 + 1;                  // views, atlas, probs, label
    if (!h->stage_dev || h->stage_floats < 2 * c * per) {
        if (h->stage_dev) SCS_HIP(hipFree(h->stage_dev));
        h->stage_dev = nullptr; h->stage_floats = 0;
        SCS_HIP(hipMalloc(&h->stage_dev, 2 * c * per * sizeof(float)));
        h->stage_floats = 2 * c * per;
    }
    struct Buf { float *ax, *co, *sa, *at, *pr; int32_t* lb; } buf[2];
    for (int i = 0; i < 2; ++i) {
        float* base = h->stage_dev + (size_t)i * c * per;
        buf[i] = Buf{base, base + c * 1024, base + 2 * c * 1024, base + 3 * c * 1024, base + 3 * c * 1024 + c * 15,
                     reinterpret_cast<int32_t*>(base + 3 * c * 1024 + 2 * c * 15)};
    }
    const int device = h->device;
    auto upload = [&](int64_t s, int which, hipError_t* err) {
        *err = hipSetDevice(device);
        const size_t m = (size_t)((n - s < hstep) ? (n - s) : hstep);
        const Buf& d = buf[which];
        if (*err == hipSuccess) *err = hipMemcpyAsync(d.ax, ax + s * 1024, m * 4096, hipMemcpyHostToDevice, su);
        if (*err == hipSuccess) *err = hipMemcpyAsync(d.co, co + s * 1024, m * 4096, hipMemcpyHostToDevice, su);
        if (*err == hipSuccess) *err = hipMemcpyAsync(d.sa, sa + s * 1024, m * 4096, hipMemcpyHostToDevice, su);
        if (*err == hipSuccess) *err = hipMemcpyAsync(d.at, atlas + s * 15, m * 60, hipMemcpyHostToDevice, su);
        if (*err == hipSuccess) *err = hipStreamSynchronize(su);
    };
    hipError_t uerr = hipSuccess;
    upload(0, 0, &uerr);
    SCS_HIP(uerr);
    int which = 0;
    for (int64_t s = 0; s < n; s += hstep, which ^= 1) {
        const size_t m = (size_t)((n - s < hstep) ? (n - s) : hstep);
        const bool more = s + hstep < n;
        std::thread next;
        hipError_t nerr = hipSuccess;
        bool inline_upload = false;
        if (more) {
            try {
                next = std::thread(upload, s + hstep, which ^ 1, &nerr);      // the other buffer: its results were read back
            } catch (...) {                                                     // no thread to be had: upload after the kernels instead
                inline_upload = true;
            }
        }
        const Buf& d = buf[which];
        hipError_t e = hipSuccess;
        // fwd_split: the chunk on the split kernels with its guarded f32 repeat behind it (forward_chunk_split)
        if (h->opt.fwd_split > 0) {
            if (s == 0) e = hipMemsetAsync(h->sp_flag, 0, sizeof(unsigned int), sc);
            rc = e == hipSuccess ? forward_chunk_split(h, d.ax, d.co, d.sa, d.at, nullptr, 0, 0, 0, nullptr, nullptr, (int)m, d.pr, d.lb, sc) : SCS_OK;
        } else rc = forward_chunk(h, d.ax, d.co, d.sa, d.at, nullptr, 0, 0, 0, nullptr, nullptr, (int)m, d.pr, d.lb, sc, nullptr, nullptr, nullptr, kFwdF32);
        if (rc == SCS_OK && e == hipSuccess && probs) e = hipMemcpyAsync(probs + s * 15, d.pr, m * 60, hipMemcpyDeviceToHost, sc);
        if (rc == SCS_OK && e == hipSuccess && labels) e = hipMemcpyAsync(labels + s, d.lb, m * 4, hipMemcpyDeviceToHost, sc);
        if (e == hipSuccess) e = hipStreamSynchronize(sc);
        if (next.joinable()) next.join();
        if (inline_upload) upload(s + hstep, which ^ 1, &nerr);
        if (rc != SCS_OK) return rc;
        SCS_HIP(e);
        SCS_HIP(nerr);
    }
    return SCS_OK;
}

int scs_forward_profiled(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas,
                         int64_t n, float* probs, int32_t* labels, void* stream, float* stage_ms, int* launches) {
    SCS_ARG(h && stage_ms && launches, "null pointer");
    SCS_ARG(n > 0 && ax && co && sa && atlas, "null input pointer or n <= 0");
    if (!h->has_params) { set_error("scs_forward_profiled before scs_set_params"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    const int step = balanced_chunk(n, h->chunk);
    int rc = ensure_workspace(h, step);
    if (rc != SCS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->packed_dirty) { rc = repack_inference(h, st); if (rc != SCS_OK) return rc; h->packed_dirty = false; }
    if (h->opt.fwd_split > 0) SCS_HIP(hipMemsetAsync(h->sp_flag, 0, 2 * sizeof(unsigned int), st));      // (words 0, 1; the profiled pass has no f32 repeat)
    const int nchunks = (int)((n + step - 1) / step);
    std::vector<hipEvent_t> ev;
    try { ev.resize((size_t)nchunks * 9); } catch (...) { set_error("out of host memory"); return SCS_ERR_HIP; }
    for (auto& e : ev) SCS_HIP(hipEventCreate(&e));
    int c = 0;
    for (int64_t s = 0; s < n; s += step, ++c) {
        const int m = (int)((n - s < step) ? (n - s) : step);
        rc = forward_chunk(h, ax + s * 1024, co + s * 1024, sa + s * 1024, atlas + s * 15, nullptr, 0, 0, 0, nullptr, nullptr,
                           m, probs ? probs + s * 15 : nullptr, labels ? labels + s : nullptr, st, &ev[(size_t)c * 9], nullptr, nullptr,
                           h->opt.fwd_split > 0 ? kFwdSplit : kFwdF32);
        if (rc != SCS_OK) break;
    }
    if (rc == SCS_OK) {
        SCS_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 8; ++i) stage_ms[i] = 0.f;
        for (int k = 0; k < nchunks; ++k)
            for (int i = 0; i < 8; ++i) {
                float ms = 0.f;
                SCS_HIP(hipEventElapsedTime(&ms, ev[(size_t)k * 9 + i], ev[(size_t)k * 9 + i + 1]));
                stage_ms[i] += ms;
            }
        *launches = nchunks;
        if (h->opt.fwd_split > 0) {
            const int ov = split_overflowed(h, st);
            if (ov < 0) rc = ov;
            else if (ov > 0) { set_error("scs_forward_profiled: a scaled activation left the f16 range of the split kernels (results void; scs_forward repeats such a batch on the f32 kernels)"); rc = SCS_ERR_STATE; }
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int scs_get_activation(scs_handle h, int which, float* host_out, size_t count) {
    SCS_ARG(h && host_out, "null pointer");
    if (!h->ws || h->last_n <= 0) { set_error("no forward has run"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    SCS_HIP(hipDeviceSynchronize());
    bool last_split = h->last_split, last_split_head = h->last_split_head;
    if (last_split) {                                      // a chunk the guarded f32 kernels repeated holds THEIR layouts (sp_flag[1])
        unsigned int repeated = 0;
        SCS_HIP(hipMemcpy(&repeated, h->sp_flag + 1, sizeof(repeated), hipMemcpyDeviceToHost));
        if (repeated) last_split = last_split_head = false;
    }
    const size_t n = (size_t)h->last_n;
    size_t per = 0;
    float* const* tow = nullptr;
    const float* single = nullptr;
    if ((which == 1 || which == 2) && h->opt.fwd_fused == 1) {
        set_error("activations 1 / 2 stay in LDS in the fused forward: set option fwd_fused to 0 or 2 before the forward call");
        return SCS_ERR_STATE;
    }
    switch (which) {
        case 0: per = 20 * 196; tow = h->a_pool1; break;
        case 1: per = 40 * 144; tow = h->a_conv3; break;     // (1, 2: written only by the three-launch path or with option fwd_fused = 2)
        case 2: per = 40 * 25; tow = h->a_pool2; break;
        case 3: per = kFlat; tow = h->a_conv5; break;
        case 4: per = kFlat; single = h->a_feat; break;
        case 5: per = kFC1ld; single = h->a_fc1; break;
        case 6: per = kFC2ld; single = h->a_fc2; break;
        case 7: per = 16; single = h->a_logits; break;
        default: SCS_ARG(false, "unknown activation id");
    }
    if (tow) {
        SCS_ARG(count == 3 * n * per, "count does not match 3*n*per-patch size");
        for (int t = 0; t < kTowers; ++t)
            SCS_HIP(hipMemcpy(host_out + t * n * per, tow[t], n * per * sizeof(float), hipMemcpyDeviceToHost));
        if (last_split && which <= 2) {
            // the split forward left two f16 planes [pixel][channel] (hi, lo; times the activation scale) per patch-tower in the same
            // bytes: back to f32 [channel][pixel]
            const int ch = which == 0 ? 20 : 40, px = (int)per / ch;
            std::vector<float> tmp(per);
            for (size_t r = 0; r < 3 * n; ++r) {
                const _Float16* hi = reinterpret_cast<const _Float16*>(host_out + r * per);
                const _Float16* lo = hi + per;
                for (int q = 0; q < px; ++q)
                    for (int c = 0; c < ch; ++c) tmp[(size_t)c * px + q] = ((float)hi[q * ch + c] + (float)lo[q * ch + c]) / kSplitActScale;
                memcpy(host_out + r * per, tmp.data(), per * sizeof(float));
            }
        }
    } else {
        SCS_ARG(count == n * per, "count does not match n*row size");
        SCS_HIP(hipMemcpy(host_out, single, n * per * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (last_split_head && which >= 3 && which <= 5) {
        // the split head keeps these as rows of halves (dense_split.h): back to the f32 layouts of the ids
        const int kp = which == 5 ? DsFC2::KP : DsK17::KP, cols = which == 5 ? kFC1 + kAtlas : kFlat, rb = ds_row_bytes(kp);
        std::vector<unsigned char> rows(n * rb);
        for (int t = 0; t < (which == 3 ? kTowers : 1); ++t) {
            const unsigned char* src = which == 3 ? h->r_c5[t] : (which == 4 ? h->r_feat : h->r_fc1);
            SCS_HIP(hipMemcpy(rows.data(), src, rows.size(), hipMemcpyDeviceToHost));
            for (size_t r = 0; r < n; ++r) {
                const _Float16* hi = reinterpret_cast<const _Float16*>(rows.data() + r * rb);
                float* o = host_out + ((size_t)t * n + r) * per;
                for (int c = 0; c < (int)per; ++c) o[c] = c < cols ? ((float)hi[c] + (float)hi[kp + c]) / kSplitActScale : 0.f;
            }
        }
    }
    return SCS_OK;
}

int scs_dilate_mask(scs_handle h, uint8_t* mask_dev, uint8_t* scratch_dev, int dx, int dy, int dz, int iterations, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(mask_dev && scratch_dev, "null pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0 && iterations >= 0, "bad dims or iteration count");
    SCS_HIP(hipSetDevice(h->device));
    const long total = (long)dx * dy * ((dz + 3) / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
    uint8_t *a = mask_dev, *b = scratch_dev;
    for (int i = 0; i < iterations; ++i) {
        hipLaunchKernelGGL(dilate_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, dx, dy, dz);
        uint8_t* t = a; a = b; b = t;
    }
    if (a != mask_dev)            // odd iteration count: the result sits in the scratch buffer
        SCS_HIP(hipMemcpyAsync(mask_dev, a, (size_t)dx * dy * dz, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}


static int dtype_ok(int dtype) { return dtype >= 0 && dtype <= 5; }

int scs_zscore_volume(scs_handle h, const void* raw_dev, int dtype, int dx, int dy, int dz, int fortran_order, float* out_dev,
                      double* scratch_dev, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(raw_dev && out_dev && scratch_dev, "null pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0 && dtype_ok(dtype), "bad dims or dtype");
    SCS_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)dx * dy * dz;
    double* stat = scratch_dev + 3 * kStatBlocks;
    hipLaunchKernelGGL(zscore_stats_kernel<0>, dim3(kStatBlocks), dim3(256), 0, st, raw_dev, dtype, n, stat, scratch_dev);
    hipLaunchKernelGGL(zscore_final_kernel<0>, dim3(1), dim3(256), 0, st, scratch_dev, stat);
    hipLaunchKernelGGL(zscore_stats_kernel<1>, dim3(kStatBlocks), dim3(256), 0, st, raw_dev, dtype, n, stat, scratch_dev);
    hipLaunchKernelGGL(zscore_final_kernel<1>, dim3(1), dim3(256), 0, st, scratch_dev, stat);
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 65535L * 8);
    hipLaunchKernelGGL(zscore_apply_kernel, dim3(blocks), dim3(256), 0, st, raw_dev, dtype, dx, dy, dz, fortran_order,
                       scratch_dev + 3 * kStatBlocks, out_dev);
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

int scs_roi_prepare(scs_handle h, const void* mask_raw_dev, int dtype, int dx, int dy, int dz, int fortran_order, int iterations,
                    uint8_t* mask_dev, uint8_t* scratch_dev, int32_t* run_offsets_dev, int64_t* total_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(mask_raw_dev && mask_dev && scratch_dev && run_offsets_dev && total_host, "null pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0 && dtype_ok(dtype) && iterations >= 0, "bad dims, dtype or iteration count");
    SCS_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)dx * dy * dz;
    SCS_ARG(n < (1L << 31), "volume too large for 32-bit voxel offsets");
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 65535L * 8);
    hipLaunchKernelGGL(binarize_kernel, dim3(blocks), dim3(256), 0, st, mask_raw_dev, dtype, dx, dy, dz, fortran_order, mask_dev);
    int rc = scs_dilate_mask(h, mask_dev, scratch_dev, dx, dy, dz, iterations, stream);
    if (rc != SCS_OK) return rc;
    const int nb = (int)((n + kRoiRun - 1) / kRoiRun);
    hipLaunchKernelGGL(roi_count_kernel, dim3(nb), dim3(256), 0, st, mask_dev, n, run_offsets_dev);
    hipLaunchKernelGGL(roi_scan_kernel, dim3(1), dim3(1024), 0, st, run_offsets_dev, nb);
    SCS_HIP(hipGetLastError());
    int32_t total = 0;
    SCS_HIP(hipMemcpyAsync(&total, run_offsets_dev + nb, sizeof(total), hipMemcpyDeviceToHost, st));
    SCS_HIP(hipStreamSynchronize(st));
    *total_host = total;
    return SCS_OK;
}

int scs_roi_centers(scs_handle h, const uint8_t* mask_dev, int dx, int dy, int dz, const int32_t* run_offsets_dev, int32_t* centers_dev,
                    int64_t n_centers, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(mask_dev && run_offsets_dev && (centers_dev || n_centers == 0), "null pointer");
    SCS_ARG(dx > 0 && dy > 0 && dz > 0 && n_centers >= 0, "bad dims or count");
    if (n_centers == 0) return SCS_OK;
    SCS_HIP(hipSetDevice(h->device));
    const long n = (long)dx * dy * dz;
    const int nb = (int)((n + kRoiRun - 1) / kRoiRun);
    hipLaunchKernelGGL(roi_write_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, mask_dev, n, dy, dz, run_offsets_dev, centers_dev, (long)n_centers);
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

int scs_param_arena(scs_handle h, float** dev_ptr, size_t* n) {
    SCS_ARG(h && dev_ptr && n, "null pointer");
    *dev_ptr = h->params; *n = SCS_N_PARAMS;
    h->packed_dirty = true;            // the caller may write through the pointer (data-parallel parameter averaging)
    return SCS_OK;
}
int scs_grad_arena(scs_handle h, float** dev_ptr, size_t* n) {
    SCS_ARG(h && dev_ptr && n, "null pointer");
    *dev_ptr = h->grads; *n = SCS_N_PARAMS;
    return SCS_OK;
}

}  // extern "C"
