// libscs - training step (forward in training mode, backward, Adam) and labelled evaluation.
//
// Replaces what nolearn's train_iter_/eval_iter_ compile out of the Lasagne graph of
// /root/reference/cnn_cort/nets.py:170-246: conv -> BatchNorm (batch statistics, eps 1e-4, running averages
// alpha 0.1) -> ReLU -> PReLU (identity on ReLU output, zero alpha gradient: SURVEY F2) -> max-pool, dropout 0.5,
// dense layers, softmax + mean categorical cross-entropy (nets.py:235), Adam (nets.py:236-237).
//
// Forward convolutions (raw, unfolded weights; their epilogue also leaves the per-patch BN sums) and the data gradients
// (correlation of the zero-padded output gradient with the channel-transposed filters) run on the same fp32 MFMA
// kernel as inference (conv_mfma.h); weight gradients on wgrad_mfma.h; every dense product (forward, data and weight
// gradients) on the split-K MFMA GEMM of gemm_mfma.h with dropout / ReLU masks in its epilogue.  BN passes are one
// workgroup per (channel, tower, batch slice) with double-precision, fixed-order reductions: the step is deterministic.
// Weight-gradient work runs on a second stream beside the data-gradient path (fork / join events inside the call).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "conv_mfma.h"
#include "gemm_mfma.h"
#include "scs_ctx.h"
#include "wgrad_mfma.h"

namespace scs {

// ------------------------------------------------------------------------------------------- configurations
// tuned at the training batch size (tools/kbench.hip train; profiles/r01_kbench_train_batch.txt): finer bands / one
// patch per workgroup than at inference, where a launch is many rounds of workgroups
using TF2 = ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_NCHW, 180, 7, 4>;
using TF3 = ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_NCHW, 208, 1, 4>;
using TF4 = ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_NCHW, 144, 1, 4>;
using TF5 = ConvCfg<40, 60, 5, 1, EPI_RAW_STATS, SRC_NCHW, 27, 1, 4>;
// data gradients: input = dZ_i zero-padded by 2, Cin' = Cout_i, Cout' = Cin_i
using DG5 = ConvCfg<60, 40, 7, 2, EPI_RAW, SRC_NCHW_PAD2, 49, 1, 3>;
using DG4 = ConvCfg<40, 40, 14, 1, EPI_RAW, SRC_NCHW_PAD2, 208, 1, 3>;
using DG3 = ConvCfg<40, 20, 16, 1, EPI_RAW, SRC_NCHW_PAD2, 144, 2, 4>;
using DG2 = ConvCfg<20, 20, 32, 1, EPI_RAW, SRC_NCHW_PAD2, 256, 5, 4>;

constexpr int kHout[kConvs] = {30, 28, 12, 10, 3};
constexpr int kRedSplit = 16;     // batch slices per channel in the two-stage BN reductions
// <CIN, COUT, HIN, row bands, workgroups per CU>
using WG2 = WgCfg<20, 20, 30, 4, 3>;
using WG3 = WgCfg<20, 40, 14, 2, 3>;
using WG4 = WgCfg<40, 40, 12, 1, 3>;
using WG1 = WgCfg<1, 20, 32, 3, 4>;
using WG5 = WgCfg<40, 60, 5, 1, 2>;
constexpr size_t kWgPartFloats = (size_t)256 * 60 * 40 * 9;   // per tower: room for 256 partials of the largest layer

}  // namespace scs

struct TrainWs {       // global namespace: forward-declared by scs_ctx.h
    static constexpr int kTowers = scs::kTowers, kConvs = scs::kConvs;
    int cap = 0;                               // samples the buffers hold
    float* base = nullptr;
    float* Z[kTowers][kConvs]{};               // raw conv outputs
    float* X[kTowers][kConvs]{};               // conv inputs (X[0] = caller's patches; X[i] = block i-1 output)
    float* H5[kTowers]{};                      // block-5 output after dropout [n][540]
    float* dZ[kTowers]{};                      // gradient wrt the current Z (max size)
    float* dX[kTowers][2]{};                   // gradient wrt block outputs, ping-pong (max size)
    float *feat = nullptr, *featd = nullptr, *R1 = nullptr, *cat2 = nullptr, *R2 = nullptr;
    float *probs = nullptr, *dlog = nullptr, *dR2 = nullptr, *dcat = nullptr, *dfeat = nullptr;
    float *loss_rows = nullptr, *loss_dev = nullptr;
    float* stats = nullptr;                    // [3][5][4][64]: mean, inv_std, dbeta, dgamma
    uint8_t* masks = nullptr;                  // [3][n][540] + [n][540] + [n][540]
    float* wpk[kTowers][kConvs]{};             // unfolded forward packs (i >= 1), conv1 [20][10] at i = 0
    float* wdg[kTowers][kConvs]{};             // data-gradient packs (i >= 1)
    float* pack_base = nullptr;
    double* eval_acc = nullptr;                // [2]: loss sum, correct count
    float* actT = nullptr;                     // transposed activations for the dense weight-gradient GEMMs [5][556][cap]: H5 x3, featd, cat2
    float* wT = nullptr;                       // transposed dense weights for the dense data-gradient GEMMs
    float* stat_part = nullptr;                // [3][n * bands][C][2] per-patch-band (sum, sum of squares) left by the forward convs
    double* red_part = nullptr;                // [3][64][kRedSplit][2] partial sums of the BN backward reductions
    float* wg_part = nullptr;                  // [3][kWgPartFloats] partial weight gradients
    // the weight-gradient kernels of a layer do not feed the data path: they run on a second stream beside the
    // data-gradient kernels (fork / join with events; the caller still sees one stream)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace scs {

static size_t al64(size_t f) { return (f + 63) / 64 * 64; }

void train_free(scs_ctx* h) {
    if (!h->tws) return;
    TrainWs* w = h->tws;
    if (w->base) (void)hipFree(w->base);
    if (w->masks) (void)hipFree(w->masks);
    if (w->pack_base) (void)hipFree(w->pack_base);
    if (w->eval_acc) (void)hipFree(w->eval_acc);
    if (w->red_part) (void)hipFree(w->red_part);
    if (w->wT) (void)hipFree(w->wT);
    if (w->wg_part) (void)hipFree(w->wg_part);
    if (w->side) (void)hipStreamDestroy(w->side);
    if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
    if (w->ev_join) (void)hipEventDestroy(w->ev_join);
    delete w;
    h->tws = nullptr;
}

static int train_ensure(scs_ctx* h, int n) {
    if (!h->tws) {
        h->tws = new (std::nothrow) TrainWs();
        if (!h->tws) { set_error("out of host memory"); return SCS_ERR_HIP; }
    }
    TrainWs* w = h->tws;
    if (!w->pack_base) {
        size_t off = 0;
        size_t of[kTowers][kConvs], og[kTowers][kConvs];
        for (int t = 0; t < kTowers; ++t)
            for (int i = 0; i < kConvs; ++i) {
                of[t][i] = off; off += al64(i == 0 ? 200 : conv_pack_floats(i));
                og[t][i] = off; off += al64(i == 0 ? 0 : (size_t)(kCout[i] / 4) * conv_chunk_floats(kCin[i]));
            }
        SCS_HIP(hipMalloc(&w->pack_base, off * sizeof(float)));
        for (int t = 0; t < kTowers; ++t)
            for (int i = 0; i < kConvs; ++i) { w->wpk[t][i] = w->pack_base + of[t][i]; w->wdg[t][i] = w->pack_base + og[t][i]; }
        SCS_HIP(hipMalloc(&w->eval_acc, 2 * sizeof(double)));
        SCS_HIP(hipMalloc(&w->wT, (size_t)(270 * 540 + 540 * 540 + 3 * 180 * 540) * sizeof(float)));
        SCS_HIP(hipMalloc(&w->red_part, (size_t)kTowers * 64 * kRedSplit * 2 * sizeof(double)));
        SCS_HIP(hipMalloc(&w->wg_part, (size_t)kTowers * kWgPartFloats * sizeof(float)));
        SCS_HIP(hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking));
        SCS_HIP(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
        SCS_HIP(hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
    }
    if (n <= w->cap) return SCS_OK;
    if (w->base) { SCS_HIP(hipFree(w->base)); w->base = nullptr; }
    if (w->masks) { SCS_HIP(hipFree(w->masks)); w->masks = nullptr; }
    const size_t c = (size_t)((n + 63) / 64 * 64);
    size_t off = 0;
    auto take = [&](size_t f) { size_t o = off; off += al64(f); return o; };
    size_t oZ[kTowers][kConvs], oX[kTowers][kConvs], oH[kTowers], odZ[kTowers], odX[kTowers][2];
    for (int t = 0; t < kTowers; ++t) {
        for (int i = 0; i < kConvs; ++i) {
            oZ[t][i] = take(c * kCout[i] * kHout[i] * kHout[i]);
            oX[t][i] = (i == 0) ? 0 : take(c * kCin[i] * kHin[i] * kHin[i]);
        }
        oH[t] = take(c * kFlat);
        odZ[t] = take(c * 20 * 900);
        odX[t][0] = take(c * 20 * 900);
        odX[t][1] = take(c * 20 * 900);
    }
    const size_t o_feat = take(c * kFlat), o_featd = take(c * kFlat), o_R1 = take(c * kFC1), o_cat2 = take(c * kFC1ld),
                 o_R2 = take(c * kFC2ld), o_probs = take(c * 16), o_dlog = take(c * 16), o_dR2 = take(c * kFC2ld),
                 o_dcat = take(c * kFC1), o_dfeat = take(c * kFlat), o_lr = take(c), o_ld = take(64),
                 o_st = take((size_t)kTowers * kConvs * 4 * 64), o_at = take((size_t)5 * kFC1ld * c),
                 o_sp = take((size_t)kTowers * c * 8 * 64 * 2);      // up to 8 bands x 64 channels x (sum, sumsq)
    SCS_HIP(hipMalloc(&w->base, off * sizeof(float)));
    SCS_HIP(hipMemset(w->base, 0, off * sizeof(float)));
    for (int t = 0; t < kTowers; ++t) {
        for (int i = 0; i < kConvs; ++i) { w->Z[t][i] = w->base + oZ[t][i]; w->X[t][i] = (i == 0) ? nullptr : w->base + oX[t][i]; }
        w->H5[t] = w->base + oH[t]; w->dZ[t] = w->base + odZ[t];
        w->dX[t][0] = w->base + odX[t][0]; w->dX[t][1] = w->base + odX[t][1];
    }
    w->feat = w->base + o_feat; w->featd = w->base + o_featd; w->R1 = w->base + o_R1; w->cat2 = w->base + o_cat2;
    w->R2 = w->base + o_R2; w->probs = w->base + o_probs; w->dlog = w->base + o_dlog; w->dR2 = w->base + o_dR2;
    w->dcat = w->base + o_dcat; w->dfeat = w->base + o_dfeat; w->loss_rows = w->base + o_lr; w->loss_dev = w->base + o_ld;
    w->stats = w->base + o_st; w->stat_part = w->base + o_sp;
    w->actT = w->base + o_at;
    SCS_HIP(hipMalloc(&w->masks, c * kFlat * 5));
    w->cap = (int)c;
    return SCS_OK;
}

// ------------------------------------------------------------------------------------------- packing (per step)
struct PackJob { const float* W; float* fwd; float* dg; int cin, cout; };
struct PackJobs { PackJob j[kTowers * kConvs]; };

// fwd (i >= 1): Wp[c4][tap][kk][n] = W[n][4c4+kk][2-dy][2-dx];  conv1: w1b[c][10] = flipped taps, bias 0
// dg  (i >= 1): Kp[o4][tap][kk][c] = W[4o4+kk][c][a][b]   (tap = 3a+b, no flip, channels transposed)
__global__ void pack_train_kernel(PackJobs jobs) {
    const PackJob J = jobs.j[blockIdx.y];
    const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (J.cin == 1) {
        for (int e = t0; e < 200; e += stride) {
            const int c = e / 10, k = e % 10;
            J.fwd[e] = k < 9 ? J.W[(c * 3 + (2 - k / 3)) * 3 + (2 - k % 3)] : 0.f;
        }
        return;
    }
    const int tot = (J.cin / 4) * conv_chunk_floats(J.cout);
    for (int e = t0; e < tot; e += stride) {
        const PackIdx ix = conv_pack_index(e, J.cout);
        float v = 0.f;
        if (ix.n >= 0) v = J.W[((ix.n * J.cin + ix.c4 * 4 + ix.kk) * 3 + (2 - ix.tap / 3)) * 3 + (2 - ix.tap % 3)];
        J.fwd[e] = v;
    }
    // data gradient: roles swapped (reduction over the conv's output channels, outputs = its input channels)
    const int totd = (J.cout / 4) * conv_chunk_floats(J.cin);
    for (int e = t0; e < totd; e += stride) {
        const PackIdx ix = conv_pack_index(e, J.cin);
        float v = 0.f;
        if (ix.n >= 0) v = J.W[(((ix.c4 * 4 + ix.kk) * J.cin + ix.n) * 3 + ix.tap / 3) * 3 + ix.tap % 3];
        J.dg[e] = v;
    }
}

// ------------------------------------------------------------------------------------------- small kernels
struct Ptr3 { float* p[kTowers]; };
struct CPtr3 { const float* p[kTowers]; };
template <class T> __device__ __forceinline__ T sel3(const T (&a)[kTowers], int t) { return t == 0 ? a[0] : (t == 1 ? a[1] : a[2]); }

// counter-based dropout keep-mask: one hash per element (splitmix64 finaliser), keep when the top bit is set
__global__ void dropout_mask_kernel(uint8_t* m, size_t total, uint64_t seed) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (e + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        m[e] = (uint8_t)(z >> 63);
    }
}

// conv1, raw (no BN): [n][32][32] -> Z1 [n][20][30][30]; weights [20][10] (bias slot 0) via scalar loads
// also writes, per patch and channel, the sum and sum of squares of the outputs (BN statistics): stat [tower][n][20][2]
__global__ __launch_bounds__(256) void conv1_raw_kernel(CPtr3 in, CPtr3 w, Ptr3 out, float* stat) {
    __shared__ __attribute__((aligned(16))) float s_x[1024];
    __shared__ float s_st[4][40];
    const int t = blockIdx.y, tid = threadIdx.x;
    const float* __restrict__ src = sel3(in.p, t) + (size_t)blockIdx.x * 1024;
    reinterpret_cast<float4*>(s_x)[tid] = reinterpret_cast<const float4*>(src)[tid];
    __syncthreads();
    const float* __restrict__ w1 = sel3(w.p, t);
    float* __restrict__ o = sel3(out.p, t) + (size_t)blockIdx.x * 20 * 900;
    float inr[4][9];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = min(tid + 256 * r, 899), y = p / 30, x = p - y * 30;
#pragma unroll
        for (int k = 0; k < 9; ++k) inr[r][k] = s_x[(y + k / 3) * 32 + x + k % 3];
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int c = 0; c < 20; ++c) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(inr[r][k], w1[c * 10 + k], acc);
            const int p = tid + 256 * r;
            if (p < 900) { o[c * 900 + p] = acc; sm += acc; sq = fmaf(acc, acc, sq); }
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) { sm += __shfl_xor(sm, ofs); sq += __shfl_xor(sq, ofs); }
        if (lane == 0) { s_st[wv][2 * c] = sm; s_st[wv][2 * c + 1] = sq; }
    }
    __syncthreads();
    if (tid < 40)
        stat[((size_t)t * gridDim.x + blockIdx.x) * 40 + tid] = (s_st[0][tid] + s_st[1][tid]) + (s_st[2][tid] + s_st[3][tid]);
}

__device__ __forceinline__ double block_sum(double v, double* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wv] = v;
    __syncthreads();
    double r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += s_red[i];
    return r;
}


// Training-mode BN + ReLU (+ 2x2 max-pool, + dropout) in one pass per (channel, tower, batch slice) workgroup.
// Every workgroup first finishes the channel's batch statistics from the per-(patch band, channel) float sums the conv
// kernels left behind - `units` partials added in double, in a fixed order - then applies
//   y = gamma*(z-mean)*inv_std + beta,  inv_std = 1/sqrt(var + eps)   (Lasagne BatchNormLayer, nets.py:172-177)
// to its slice.  Slice 0 also publishes mean / inv_std for the backward pass and moves the running averages in the
// arena (alpha 0.1).
template <bool POOL, int H>
__global__ __launch_bounds__(256) void bn_relu_kernel(CPtr3 Z, Ptr3 out, int n, int C, const float* part, int units, float* stats, int layer,
                                                     float* params, int bn_off0, int tower_stride, const uint8_t* mask) {
    __shared__ double s_red[4];
    __shared__ float s_mi[2];
    const int c = blockIdx.x, t = blockIdx.y, sp = blockIdx.z, split = gridDim.z;
    constexpr int pix = H * H;
    {
        const float2* __restrict__ p = reinterpret_cast<const float2*>(part) + (size_t)t * units * C + c;
        double s = 0, q = 0;
        for (int u = threadIdx.x; u < units; u += 256) { const float2 v = p[(size_t)u * C]; s += v.x; q += v.y; }
        s = block_sum(s, s_red);
        q = block_sum(q, s_red);
        if (threadIdx.x == 0) {
            const double total = (double)n * pix;
            const double mean = s / total, var = fmax(q / total - mean * mean, 0.0);
            const double inv = 1.0 / sqrt(var + (double)kBnEps);
            s_mi[0] = (float)mean; s_mi[1] = (float)inv;
            if (sp == 0) {
                float* st = stats + ((size_t)(t * kConvs + layer) * 4) * 64;
                st[c] = (float)mean; st[64 + c] = (float)inv;
                float* bn = params + bn_off0 + t * tower_stride;           // beta, gamma, mean, inv_std
                bn[2 * C + c] = (1.f - kBnAlpha) * bn[2 * C + c] + kBnAlpha * (float)mean;
                bn[3 * C + c] = (1.f - kBnAlpha) * bn[3 * C + c] + kBnAlpha * (float)inv;
            }
        }
        __syncthreads();
    }
    const float* bn = params + bn_off0 + t * tower_stride;
    const float sc = bn[C + c] * s_mi[1], sh = bn[c] - s_mi[0] * sc;
    const float* __restrict__ z = sel3(Z.p, t);
    float* __restrict__ o = sel3(out.p, t);
    const int per = (n + split - 1) / split, n0 = min(n, sp * per), n1 = min(n, n0 + per);
    if constexpr (POOL) {
        constexpr int HO = H / 2, W = HO * HO;
        const int total = (n1 - n0) * W;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / W, wdx = idx % W, wy = wdx / HO, wx = wdx - wy * HO;
            const size_t plane = (size_t)ni * C + c;
            const float* zp = z + plane * pix + (2 * wy) * H + 2 * wx;
            const float2 r0 = *reinterpret_cast<const float2*>(zp), r1 = *reinterpret_cast<const float2*>(zp + H);
            const float m = fmaxf(fmaxf(fmaf(r0.x, sc, sh), fmaf(r0.y, sc, sh)), fmaxf(fmaf(r1.x, sc, sh), fmaf(r1.y, sc, sh)));
            o[plane * W + wdx] = fmaxf(m, 0.f);
        }
    } else if constexpr (pix % 4 == 0) {
        constexpr int Q = pix / 4;
        const int total = (n1 - n0) * Q;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / Q, q = idx % Q;
            const size_t plane = (size_t)ni * C + c;
            const float4 v = reinterpret_cast<const float4*>(z + plane * pix)[q];
            reinterpret_cast<float4*>(o + plane * pix)[q] = make_float4(fmaxf(fmaf(v.x, sc, sh), 0.f), fmaxf(fmaf(v.y, sc, sh), 0.f),
                                                                      fmaxf(fmaf(v.z, sc, sh), 0.f), fmaxf(fmaf(v.w, sc, sh), 0.f));
        }
    } else {
        const int total = (n1 - n0) * pix;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / pix, e = idx % pix;
            const size_t at = ((size_t)ni * C + c) * pix + e;
            float v = fmaxf(fmaf(z[at], sc, sh), 0.f);
            if (mask) v = mask[(size_t)t * n * kFlat + at] ? v * 2.f : 0.f;     // dropout in front of d1 (conv5: C*pix = 540)
            o[at] = v;
        }
    }
}

// atlas -> columns 540..554 of the fc_2 input rows (ConcatLayer elem_channels2, nets.py:224); column 555 = 0
__global__ void atlas_cols_kernel(const float* __restrict__ atlas, float* __restrict__ cat2, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 16) return;
    const int row = e >> 4, c = e & 15;
    cat2[(size_t)row * kFC1ld + kFC1 + c] = c < 15 ? atlas[(size_t)row * 15 + c] : 0.f;
}

// out_layer + softmax + cross-entropy on unpacked arena weights W[270][15], b[15]
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ R2, const float* __restrict__ W,
                                                        const float* __restrict__ b, const uint8_t* __restrict__ y, int n,
                                                        float* probs /*[n][16]*/, float* dlog /*[n][16] or null*/,
                                                        float* loss_rows /*[n] or null*/, double* eval_acc /*null or [2]*/) {
    __shared__ float s_w[kFC2 * 15 + 16];
    for (int e = threadIdx.x; e < kFC2 * 15; e += blockDim.x) s_w[e] = W[e];
    if (threadIdx.x < 15) s_w[kFC2 * 15 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float z[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) z[c] = s_w[kFC2 * 15 + c];
    const float* x = R2 + (size_t)row * kFC2ld;
    for (int k = 0; k < kFC2; ++k) {
        const float xv = x[k];
#pragma unroll
        for (int c = 0; c < 15; ++c) z[c] = fmaf(xv, s_w[k * 15 + c], z[c]);
    }
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < 15; ++c) mx = fmaxf(mx, z[c]);
    float p[15], s = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) { p[c] = expf(z[c] - mx); s += p[c]; }
    const float inv = 1.f / s;
    const int yi = y[row];
    int best = 0; float bp = -1.f, py = 0.f, zy = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        p[c] *= inv;
        if (p[c] > bp) { bp = p[c]; best = c; }
        if (c == yi) { py = p[c]; zy = z[c]; }
        probs[(size_t)row * 16 + c] = p[c];
    }
    const float loss = -(zy - mx - logf(s));                       // -log softmax, stable
    if (loss_rows) loss_rows[row] = loss;
    if (dlog) {
        const float in = 1.f / (float)n;
#pragma unroll
        for (int c = 0; c < 15; ++c) dlog[(size_t)row * 16 + c] = (p[c] - (c == yi ? 1.f : 0.f)) * in;
        dlog[(size_t)row * 16 + 15] = 0.f;
    }
    if (eval_acc) {
        atomicAdd(&eval_acc[0], (double)loss);
        if (best == yi) atomicAdd(&eval_acc[1], 1.0);
    }
    (void)py;
}

// Training head in one pass per row: out_layer (270 -> 15) + softmax + cross-entropy, d logits = (p - onehot)/n, and
// the gradient that flows back through fc_2's ReLU:  dR2[row][k] = (R2[row][k] > 0) * sum_c dlog[c] * W[k][c].
// 16 lanes per row (lane = class), 16 rows per workgroup.
__global__ __launch_bounds__(256) void head_train_kernel(const float* __restrict__ R2, const float* __restrict__ W,
                                                        const float* __restrict__ b, const uint8_t* __restrict__ y, int n,
                                                        float* __restrict__ probs /*[n][16]*/, float* __restrict__ dlog /*[n][16]*/,
                                                        float* __restrict__ loss_rows /*[n]*/, float* __restrict__ dR2 /*[n][kFC2ld]*/) {
    __shared__ float s_w[kFC2 * 15 + 16];
    __shared__ float s_x[16][kFC2ld];
    __shared__ float s_d[16][16];
    for (int e = threadIdx.x; e < kFC2 * 15; e += 256) s_w[e] = W[e];
    if (threadIdx.x < 15) s_w[kFC2 * 15 + threadIdx.x] = b[threadIdx.x];
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int row0 = blockIdx.x * 16, row = row0 + r;
    for (int e = threadIdx.x; e < 16 * kFC2ld; e += 256) {
        const int rr = e / kFC2ld, k = e - rr * kFC2ld;
        s_x[rr][k] = (row0 + rr < n) ? R2[(size_t)(row0 + rr) * kFC2ld + k] : 0.f;
    }
    __syncthreads();
    const int cc = c < 15 ? c : 14;                                // lane 15 shadows class 14, its results are dropped
    float z = s_w[kFC2 * 15 + cc];
    for (int k = 0; k < kFC2; ++k) z = fmaf(s_x[r][k], s_w[k * 15 + cc], z);
    float mx = z;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = c < 15 ? expf(z - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float pr = e / sum;
    const int yi = row < n ? y[row] : 0;
    const float dl = (c < 15 && row < n) ? (pr - (c == yi ? 1.f : 0.f)) / (float)n : 0.f;
    s_d[r][c] = dl;
    if (row < n) {
        probs[(size_t)row * 16 + c] = c < 15 ? pr : 0.f;
        dlog[(size_t)row * 16 + c] = dl;
        if (c == yi) loss_rows[row] = -(z - mx - logf(sum));       // -log softmax, stable
    }
    __syncthreads();
    if (row < n) {
        for (int k = c; k < kFC2ld; k += 16) {
            float g = 0.f;
            if (k < kFC2 && s_x[r][k] > 0.f) {
#pragma unroll
                for (int j = 0; j < 15; ++j) g = fmaf(s_d[r][j], s_w[k * 15 + j], g);
            }
            dR2[(size_t)row * kFC2ld + k] = g;
        }
    }
}

// d out_layer.W[k][c] = sum_rows R2[row][k] * dlog[row][c]  and  d out_layer.b[c] = sum_rows dlog[row][c].
// 16 outputs x 16 row groups per workgroup (output index 4050 + c = the bias gradient), fixed-order combination.
__global__ __launch_bounds__(256) void outw_grad_kernel(const float* __restrict__ R2, const float* __restrict__ dlog, int n,
                                                       float* __restrict__ gW /*[270][15]*/, float* __restrict__ gb /*[15]*/) {
    __shared__ float s_p[16][17];
    const int ex = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + ex;
    const bool isw = e < kFC2 * 15, isb = !isw && e < kFC2 * 15 + 15;
    const int k = isw ? e / 15 : 0, c = isw ? e - k * 15 : (isb ? e - kFC2 * 15 : 0);
    float s = 0.f;
    if (isw || isb) {
#pragma unroll 4
        for (int row = rg; row < n; row += 16) {
            const float x = isw ? R2[(size_t)row * kFC2ld + k] : 1.f;
            s = fmaf(x, dlog[(size_t)row * 16 + c], s);
        }
    }
    s_p[rg][ex] = s;
    __syncthreads();
    if (rg == 0 && (isw || isb)) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += s_p[i][ex];
        if (isw) gW[e] = v; else gb[c] = v;
    }
}

__global__ void mean_kernel(const float* v, int n, float* out) {
    __shared__ double s_red[4];
    double s = 0;
    for (int e = threadIdx.x; e < n; e += blockDim.x) s += v[e];
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) out[0] = (float)(s / n);
}

// out[j] = sum_r g[r][ldg + j]   (64 columns x 4 row groups per block, fixed summation order)
struct ColProblem { const float* g; int ldg, cols; float* out; };
constexpr int kColProblems = 5;
struct ColArgs { ColProblem p[kColProblems]; int n; };
// out[j] = sum_rows g[row][j] for up to 5 matrices in one launch (the bias gradients of the dense layers); blockIdx.y = problem
__global__ __launch_bounds__(256) void colsum_kernel(ColArgs a) {
    __shared__ double s_p[4][64];
    ColProblem P = a.p[0];
#pragma unroll
    for (int i = 1; i < kColProblems; ++i) if ((int)blockIdx.y == i) P = a.p[i];
    const int cj = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cj;
    if ((int)blockIdx.x * 64 >= P.cols) return;
    double s = 0;
    if (j < P.cols)
        for (int r = rg; r < a.n; r += 4) s += P.g[(size_t)r * P.ldg + j];
    s_p[rg][cj] = s;
    __syncthreads();
    if (rg == 0 && j < P.cols) P.out[j] = (float)(s_p[0][cj] + s_p[1][cj] + s_p[2][cj] + s_p[3][cj]);
}

// out[c][r] (ldo) = in[r][c] (ldi), r < rows, c < cols; blockIdx.z selects one of up to 10 problems of different sizes
struct TrProblem { const float* in; int ldi; float* out; int ldo; int rows, cols; };
constexpr int kTrProblems = 10;
struct TrArgs { TrProblem p[kTrProblems]; };
__global__ __launch_bounds__(256) void transpose_kernel(TrArgs a) {
    __shared__ float tile[32][33];
    TrProblem P = a.p[0];
#pragma unroll
    for (int i = 1; i < kTrProblems; ++i) if ((int)blockIdx.z == i) P = a.p[i];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= P.rows || c0 >= P.cols) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < P.rows && c < P.cols) ? P.in[(size_t)r * P.ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < P.cols && r < P.rows) P.out[(size_t)c * P.ldo + r] = tile[tx][ty + 8 * i];
    }
}

static void transpose(const TrArgs& a, int nprob, hipStream_t st) {
    int mr = 0, mc = 0;
    for (int i = 0; i < nprob; ++i) { mr = a.p[i].rows > mr ? a.p[i].rows : mr; mc = a.p[i].cols > mc ? a.p[i].cols : mc; }
    hipLaunchKernelGGL(transpose_kernel, dim3((mc + 31) / 32, (mr + 31) / 32, nprob), dim3(256), 0, st, a);
}

// ---- BN backward.  dy(n,c,y,x) = dOut routed through max-pool (every tied maximum receives it, as Theano's
// MaxPoolGrad does) and ReLU; then  dbeta = sum dy,  dgamma = sum dy*xhat,
//   dz = gamma*inv_std * (dy - dbeta/M - xhat*dgamma/M),   M = n*H*H.
// Two passes over (Z, dOut), both as one workgroup per (channel, tower, batch slice): pass 1 leaves per-slice partial
// sums in double, pass 2 adds them in a fixed order and writes dz.  With pooling a thread owns one 2x2 window (4 z
// values as two 8-byte loads + one dOut); without, 4 consecutive pixels (16-byte loads) when the plane allows.
struct BnCh { float mean, inv, sc, sh; };
__device__ __forceinline__ BnCh bn_channel(const float* stats, int layer, const float* params, int bn_off0, int tower_stride, int t, int C, int c) {
    const float* st = stats + ((size_t)(t * kConvs + layer) * 4) * 64;
    const float* bn = params + bn_off0 + t * tower_stride;
    BnCh r;
    r.mean = st[c]; r.inv = st[64 + c]; r.sc = bn[C + c] * r.inv; r.sh = bn[c] - r.mean * r.sc;
    return r;
}

// F(z, dy_routed, element index within the (ni, c) plane) for every element of patches [n0, n1) of channel c
template <bool POOL, int H, class F>
__device__ __forceinline__ void bn_for_each(const float* __restrict__ z, const float* __restrict__ d, int C, int c, int n0, int n1,
                                            const BnCh& b, F&& f) {
    constexpr int pix = H * H;
    if constexpr (POOL) {
        constexpr int HO = H / 2, W = HO * HO;
        const int total = (n1 - n0) * W;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / W, wdx = idx % W, wy = wdx / HO, wx = wdx - wy * HO;
            const size_t plane = (size_t)ni * C + c;
            const int e0 = (2 * wy) * H + 2 * wx;
            const float* zp = z + plane * pix + e0;
            const float2 r0 = *reinterpret_cast<const float2*>(zp), r1 = *reinterpret_cast<const float2*>(zp + H);
            const float dv = d[plane * W + wdx];
            const float y0 = fmaf(r0.x, b.sc, b.sh), y1 = fmaf(r0.y, b.sc, b.sh), y2 = fmaf(r1.x, b.sc, b.sh), y3 = fmaf(r1.y, b.sc, b.sh);
            const float m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
            const bool on = m > 0.f;
            f(plane, e0, r0.x, (on && y0 == m) ? dv : 0.f, r0.y, (on && y1 == m) ? dv : 0.f,
              e0 + H, r1.x, (on && y2 == m) ? dv : 0.f, r1.y, (on && y3 == m) ? dv : 0.f);
        }
    } else if constexpr (pix % 4 == 0) {
        constexpr int Q = pix / 4;
        const int total = (n1 - n0) * Q;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / Q, q = idx % Q;
            const size_t plane = (size_t)ni * C + c;
            const float4 zv = reinterpret_cast<const float4*>(z + plane * pix)[q];
            const float4 dv = reinterpret_cast<const float4*>(d + plane * pix)[q];
            f(plane, 4 * q, zv.x, fmaf(zv.x, b.sc, b.sh) > 0.f ? dv.x : 0.f, zv.y, fmaf(zv.y, b.sc, b.sh) > 0.f ? dv.y : 0.f,
              4 * q + 2, zv.z, fmaf(zv.z, b.sc, b.sh) > 0.f ? dv.z : 0.f, zv.w, fmaf(zv.w, b.sc, b.sh) > 0.f ? dv.w : 0.f);
        }
    } else {
        const int total = (n1 - n0) * pix;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int ni = n0 + idx / pix, e = idx % pix;
            const size_t plane = (size_t)ni * C + c;
            const float zv = z[plane * pix + e];
            f.one(plane, e, zv, fmaf(zv, b.sc, b.sh) > 0.f ? d[plane * pix + e] : 0.f);
        }
    }
}

template <bool POOL, int H>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(CPtr3 Z, CPtr3 dOut, int n, int C, const float* stats, int layer,
                                                            const float* params, int bn_off0, int tower_stride, double* part) {
    __shared__ double s_red[4];
    const int c = blockIdx.x, t = blockIdx.y, sp = blockIdx.z;
    const float* __restrict__ z = sel3(Z.p, t);
    const float* __restrict__ d = sel3(dOut.p, t);
    const BnCh b = bn_channel(stats, layer, params, bn_off0, tower_stride, t, C, c);
    const int per = (n + kRedSplit - 1) / kRedSplit, n0 = min(n, sp * per), n1 = min(n, n0 + per);
    struct Acc {
        float fb = 0.f, fg = 0.f; BnCh b;
        __device__ void one(size_t, int, float zv, float dy) { fb += dy; fg = fmaf(dy, (zv - b.mean) * b.inv, fg); }
        __device__ void operator()(size_t pl, int e0, float z0, float d0, float z1, float d1, int e1, float z2, float d2, float z3, float d3) {
            one(pl, e0, z0, d0); one(pl, e0, z1, d1); one(pl, e1, z2, d2); one(pl, e1, z3, d3);
        }
    } acc;
    acc.b = b;
    bn_for_each<POOL, H>(z, d, C, c, n0, n1, b, acc);
    const double sb = block_sum((double)acc.fb, s_red);
    const double sg = block_sum((double)acc.fg, s_red);
    if (threadIdx.x == 0) { double* o = part + (((size_t)t * 64 + c) * kRedSplit + sp) * 2; o[0] = sb; o[1] = sg; }
}

// pass 2; `split` batch slices (gridDim.z).  Slice 0 also stores d beta / d gamma into the gradient arena.
template <bool POOL, int H>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(CPtr3 Z, CPtr3 dOut, Ptr3 dZ, int n, int C, const float* stats, int layer,
                                                          const float* params, int bn_off0, int tower_stride, const double* part,
                                                          float* grads) {
    __shared__ float s_sum[2];
    const int c = blockIdx.x, t = blockIdx.y, sp = blockIdx.z, split = gridDim.z;
    const float* __restrict__ z = sel3(Z.p, t);
    const float* __restrict__ d = sel3(dOut.p, t);
    float* __restrict__ dz = sel3(dZ.p, t);
    const BnCh b = bn_channel(stats, layer, params, bn_off0, tower_stride, t, C, c);
    if (threadIdx.x == 0) {
        double sb = 0, sg = 0;
        for (int k = 0; k < kRedSplit; ++k) { const double* o = part + (((size_t)t * 64 + c) * kRedSplit + k) * 2; sb += o[0]; sg += o[1]; }
        s_sum[0] = (float)sb; s_sum[1] = (float)sg;
        if (sp == 0) { float* gb = grads + bn_off0 + t * tower_stride; gb[c] = (float)sb; gb[C + c] = (float)sg; }   // mean / inv_std stay 0
    }
    __syncthreads();
    constexpr int pix = H * H;
    const float invM = 1.f / ((float)n * pix);
    const int per = (n + split - 1) / split, n0 = min(n, sp * per), n1 = min(n, n0 + per);
    struct Wr {
        float* dz; BnCh b; float mb, mg;
        __device__ float val(float zv, float dy) const { return b.sc * (dy - mb - (zv - b.mean) * b.inv * mg); }
        __device__ void one(size_t pl, int e, float zv, float dy) { dz[pl * pix + e] = val(zv, dy); }
        __device__ void operator()(size_t pl, int e0, float z0, float d0, float z1, float d1, int e1, float z2, float d2, float z3, float d3) {
            float* p = dz + pl * pix;
            if constexpr (POOL) {
                *reinterpret_cast<float2*>(p + e0) = make_float2(val(z0, d0), val(z1, d1));
                *reinterpret_cast<float2*>(p + e1) = make_float2(val(z2, d2), val(z3, d3));
            } else {
                *reinterpret_cast<float4*>(p + e0) = make_float4(val(z0, d0), val(z1, d1), val(z2, d2), val(z3, d3));
            }
        }
    } wr;
    wr.dz = dz; wr.b = b; wr.mb = s_sum[0] * invM; wr.mg = s_sum[1] * invM;
    bn_for_each<POOL, H>(z, d, C, c, n0, n1, b, wr);
}

// sum of the partial weight gradients in a fixed order (deterministic): 64 elements x 4 groups of partials per block,
// 8 independent loads in flight per thread
__global__ __launch_bounds__(256) void wgrad_final_kernel(const float* part, int nsplit, int Cin, int Cout, float* grads, int w_off0,
                                                          int tower_stride) {
    __shared__ float s_red[4][64];
    const int t = blockIdx.y, tot = Cout * Cin * 9;
    const int ex = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + ex;
    const int per = (nsplit + 3) / 4, sp0 = grp * per, sp1 = min(nsplit, sp0 + per);
    float v = 0.f;
    if (e < tot) {
        const float* __restrict__ base = part + (size_t)t * nsplit * tot + e;
        int sp = sp0;
        for (; sp + 8 <= sp1; sp += 8) {
            float r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = base[(size_t)(sp + u) * tot];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += r[u];
        }
        for (; sp < sp1; ++sp) v += base[(size_t)sp * tot];
    }
    s_red[grp][ex] = v;
    __syncthreads();
    if (grp == 0 && e < tot) {
        v = (s_red[0][ex] + s_red[1][ex]) + (s_red[2][ex] + s_red[3][ex]);
        const int o = e % Cout, j = e / Cout, c = j / 9, k = j - c * 9;        // partials are [c*9 + tap][o]
        grads[w_off0 + t * tower_stride + (o * Cin + c) * 9 + (2 - k / 3) * 3 + (2 - k % 3)] = v;   // flip back to Lasagne's W
    }
}

// batch slices of the weight-gradient kernels per conv layer (workgroups = slices x bands x 3 towers; partials to sum =
// slices x bands x parts).  SCS_WG_SLICES="a,b,c,d,e" overrides for tuning runs.
static const int* wg_slices() {
    static int v[kConvs] = {64, 64, 128, 256, 64};
    static bool done = false;
    if (!done) {
        done = true;
        if (const char* e = getenv("SCS_WG_SLICES")) {
            int x[kConvs];
            if (sscanf(e, "%d,%d,%d,%d,%d", &x[0], &x[1], &x[2], &x[3], &x[4]) == kConvs)
                for (int i = 0; i < kConvs; ++i) if (x[i] >= 1 && x[i] <= 256) v[i] = x[i];
        }
    }
    return v;
}

// lasagne.updates.adam over the arena
__global__ void adam_kernel(float* p, float* g, float* m, float* v, int n, float a_t, float b1, float b2, float eps, float gscale) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const float gr = g[e] * gscale;
        const float mn = b1 * m[e] + (1.f - b1) * gr;
        const float vn = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = mn; v[e] = vn;
        p[e] -= a_t * mn / (sqrtf(vn) + eps);
    }
}

static int set_train_attrs() {
    static bool done = false;
    if (done) return SCS_OK;
    SCS_HIP(conv_set_attr<TF2>()); SCS_HIP(conv_set_attr<TF3>()); SCS_HIP(conv_set_attr<TF4>()); SCS_HIP(conv_set_attr<TF5>());
    SCS_HIP(conv_set_attr<DG5>()); SCS_HIP(conv_set_attr<DG4>()); SCS_HIP(conv_set_attr<DG3>()); SCS_HIP(conv_set_attr<DG2>());
    SCS_HIP(wgrad_set_attr<WG2>()); SCS_HIP(wgrad_set_attr<WG3>()); SCS_HIP(wgrad_set_attr<WG4>());
    SCS_HIP(wgrad_set_attr<WG1>()); SCS_HIP(wgrad_set_attr<WG5>());
    done = true;
    return SCS_OK;
}

static inline dim3 ew_grid(long total) { long b = (total + 255) / 256; return dim3((unsigned)(b < 4096 ? (b > 0 ? b : 1) : 4096)); }

}  // namespace scs

using namespace scs;

extern "C" {

int scs_train_step(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, const uint8_t* y,
                   int64_t n64, uint64_t dropout_seed, const uint8_t* masks_dev, float* loss_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n64 > 0 && n64 <= (1 << 20), "n out of range");
    SCS_ARG(ax && co && sa && atlas && y, "null input pointer");
    if (!h->has_params) { set_error("scs_train_step before scs_set_params"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    int rc = set_train_attrs();
    if (rc != SCS_OK) return rc;
    const int n = (int)n64;
    rc = train_ensure(h, n);
    if (rc != SCS_OK) return rc;
    TrainWs* w = h->tws;
    hipStream_t st = (hipStream_t)stream;
    float* P = h->params;
    float* G = h->grads;
    constexpr HeadOff H = head_off();
    constexpr TowerOff T0 = tower_off(0);
    const float* xin[kTowers] = {ax, co, sa};

    // every trainable entry of the gradient arena is overwritten below; the rest (BN running statistics, PReLU alphas)
    // has been zero since the arena was allocated

    // ---- masks
    const uint8_t* masks = masks_dev;
    if (!masks) {
        const size_t total = (size_t)n * kFlat * 5;
        hipLaunchKernelGGL(dropout_mask_kernel, ew_grid((long)total), dim3(256), 0, st, w->masks, total, dropout_seed);
        masks = w->masks;
    }
    const uint8_t* mask5 = masks;                                  // [3][n][540]
    const uint8_t* mask1 = masks + (size_t)3 * n * kFlat;          // f1_drop [n][540]
    const uint8_t* mask2 = mask1 + (size_t)n * kFlat;              // f2_drop [n][540]

    // ---- per-step operand packs (unfolded)
    PackJobs jobs;
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        for (int i = 0; i < kConvs; ++i) jobs.j[t * kConvs + i] = PackJob{P + T.convW[i], w->wpk[t][i], w->wdg[t][i], kCin[i], kCout[i]};
    }
    hipLaunchKernelGGL(pack_train_kernel, dim3(8, kTowers * kConvs), dim3(256), 0, st, jobs);

    // ---- forward, training mode
    auto c3 = [&](auto& a, int i) { CPtr3 r; for (int t = 0; t < kTowers; ++t) r.p[t] = a[t][i]; return r; };
    auto p3 = [&](auto& a, int i) { Ptr3 r; for (int t = 0; t < kTowers; ++t) r.p[t] = a[t][i]; return r; };
    for (int i = 0; i < kConvs; ++i) {
        if (i == 0) {
            CPtr3 in{{ax, co, sa}}, wp{{w->wpk[0][0], w->wpk[1][0], w->wpk[2][0]}};
            hipLaunchKernelGGL(conv1_raw_kernel, dim3(n, kTowers), dim3(256), 0, st, in, wp, p3(w->Z, 0), w->stat_part);
        } else {
            ConvArgs a{};
            a.n = n;
            for (int t = 0; t < kTowers; ++t) { a.in[t] = w->X[t][i]; a.w[t] = w->wpk[t][i]; a.bias[t] = nullptr; a.out[t] = w->Z[t][i]; }
            a.stats = w->stat_part;
            if (i == 1) conv_launch<TF2>(a, st);
            else if (i == 2) conv_launch<TF3>(a, st);
            else if (i == 3) conv_launch<TF4>(a, st);
            else conv_launch<TF5>(a, st);
        }
        const int C = kCout[i], Hh = kHout[i];
        constexpr int kStatBands[kConvs] = {1, TF2::NB, TF3::NB, TF4::NB, TF5::NB};
        Ptr3 out;
        for (int t = 0; t < kTowers; ++t) out.p[t] = (i == kConvs - 1) ? w->H5[t] : w->X[t][i + 1];
        const int split = max(1, min(n, (int)((long)n * Hh * Hh / 8192)));
        auto bn_fwd = [&](auto pool_tag, auto h_tag) {
            constexpr bool PL = decltype(pool_tag)::value;
            constexpr int HH = decltype(h_tag)::value;
            hipLaunchKernelGGL((bn_relu_kernel<PL, HH>), dim3(C, kTowers, split), dim3(256), 0, st, c3(w->Z, i), out, n, C, w->stat_part,
                               n * kStatBands[i], w->stats, i, P, T0.bn[i], kTowerFloats, (i == kConvs - 1) ? mask5 : (const uint8_t*)nullptr);
        };
        using std::integral_constant;
        if (i == 0) bn_fwd(integral_constant<bool, false>{}, integral_constant<int, 30>{});
        else if (i == 1) bn_fwd(integral_constant<bool, true>{}, integral_constant<int, 28>{});
        else if (i == 2) bn_fwd(integral_constant<bool, false>{}, integral_constant<int, 12>{});
        else if (i == 3) bn_fwd(integral_constant<bool, true>{}, integral_constant<int, 10>{});
        else bn_fwd(integral_constant<bool, false>{}, integral_constant<int, 3>{});
    }
    // dense head (weights straight from the arena: [K][N] row-major)
    GemmArgs g{};
    g.M = n; g.K4 = kFlat; g.K = kFlat; g.N = kD1; g.relu = 1;
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        // feat = relu(H5 W + b) (kept: it gates the backward pass), featd = f1_drop(feat), both from the GEMM epilogue
        g.p[t] = GemmProblem{w->H5[t], kFlat, P + T.d1W, kD1, P + T.d1b, w->feat + t * kD1, kFlat,
                             mask1 + t * kD1, kFlat, nullptr, 0, w->featd + t * kD1, kFlat};
    }
    gemm_launch(g, kTowers, st);
    g.K4 = kFC1; g.K = kFC1; g.N = kFC1;
    g.p[0] = GemmProblem{w->featd, kFlat, P + H.fc1W, kFC1, P + H.fc1b, w->R1, kFC1, mask2, kFC1, nullptr, 0, w->cat2, kFC1ld};
    gemm_launch(g, 1, st);                    // R1 = relu(.), cat2[:, :540] = f2_drop(R1)
    hipLaunchKernelGGL(atlas_cols_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, st, atlas, w->cat2, n);   // cols 540..555
    g.K4 = kFC1ld; g.K = kFC1 + kAtlas; g.N = kFC2;
    g.p[0] = GemmProblem{w->cat2, kFC1ld, P + H.fc2W, kFC2, P + H.fc2b, w->R2, kFC2ld};
    gemm_launch(g, 1, st);
    hipLaunchKernelGGL(head_train_kernel, dim3((n + 15) / 16), dim3(256), 0, st, w->R2, P + H.outW, P + H.outb, y, n, w->probs, w->dlog,
                       w->loss_rows, w->dR2);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, w->loss_rows, n, w->loss_dev);

    // ---- backward: dense head.  d out_layer and dR2 (through fc_2's ReLU) came out of head_train_kernel's pass.
    // Weight / bias gradients go to the side stream (sd), data gradients stay on the caller's stream (st).
    hipStream_t sd = w->side;
    auto fork = [&]() { (void)hipEventRecord(w->ev_fork, st); (void)hipStreamWaitEvent(sd, w->ev_fork, 0); };   // sd sees what st has done
    auto join = [&]() { (void)hipEventRecord(w->ev_join, sd); (void)hipStreamWaitEvent(st, w->ev_join, 0); };   // st waits for sd
    // The remaining dense gradients run on the MFMA GEMM: weight gradients take the transposed activation as A
    // ([in][n]) and the upstream gradient as W ([n][out]); data gradients take the transposed weight as W.  All ten
    // transposes (5 weights, 5 activations) are one launch.
    const int ldT = w->cap;
    float* wT2 = w->wT;                       // [270][540]  = fc_2.W[:540]^T
    float* wT1 = wT2 + 270 * 540;             // [540][540]  = FC1.W^T
    float* wTd = wT1 + 540 * 540;             // [3][180][540] = d1.W^T
    float* aT5 = w->actT;                                         // [3][540][ldT]  H5^T per tower
    float* aTf = w->actT + (size_t)3 * kFC1ld * ldT;              // [540][ldT]     featd^T
    float* aTc = w->actT + (size_t)4 * kFC1ld * ldT;              // [555][ldT]     cat2^T
    {
        TrArgs tr{};
        tr.p[0] = TrProblem{P + H.fc2W, kFC2, wT2, kFC1, kFC1, kFC2};
        tr.p[1] = TrProblem{P + H.fc1W, kFC1, wT1, kFlat, kFlat, kFC1};
        for (int t = 0; t < kTowers; ++t) {
            const TowerOff T = tower_off(t * kTowerFloats);
            tr.p[2 + t] = TrProblem{P + T.d1W, kD1, wTd + t * kD1 * kFlat, kFlat, kFlat, kD1};
            tr.p[5 + t] = TrProblem{w->H5[t], kFlat, aT5 + (size_t)t * kFC1ld * ldT, ldT, n, kFlat};
        }
        tr.p[8] = TrProblem{w->featd, kFlat, aTf, ldT, n, kFlat};
        tr.p[9] = TrProblem{w->cat2, kFC1ld, aTc, ldT, n, kFC1 + kAtlas};
        transpose(tr, kTrProblems, st);
    }
    GemmArgs bg{};
    bg.relu = 0;
    fork();                                   // dR2, dlog, the transposes are complete for sd
    hipLaunchKernelGGL(outw_grad_kernel, dim3((kFC2 * 15 + 15 + 15) / 16), dim3(256), 0, sd, w->R2, w->dlog, n, G + H.outW, G + H.outb);
    // fc_2: dW2 = cat2^T dPre2 (sd), dcat = dPre2 W2[:540]^T (st)
    bg.M = kFC1 + kAtlas; bg.K4 = (n + 3) / 4 * 4; bg.K = n; bg.N = kFC2;
    bg.p[0] = GemmProblem{aTc, ldT, w->dR2, kFC2ld, nullptr, G + H.fc2W, kFC2};
    gemm_launch(bg, 1, sd);
    bg.M = n; bg.K4 = kFC2ld; bg.K = kFC2; bg.N = kFC1;
    bg.p[0] = GemmProblem{w->dR2, kFC2ld, wT2, kFC1, nullptr, w->dcat, kFC1, mask2, kFC1, w->R1, kFC1, nullptr, 0};
    gemm_launch(bg, 1, st);                   // through f2_drop and FC1's ReLU in the epilogue
    // FC1: dW1 (sd) needs the masked dcat; dfeat (st)
    fork();
    bg.M = kFlat; bg.K4 = (n + 3) / 4 * 4; bg.K = n; bg.N = kFC1;
    bg.p[0] = GemmProblem{aTf, ldT, w->dcat, kFC1, nullptr, G + H.fc1W, kFC1};
    gemm_launch(bg, 1, sd);
    bg.M = n; bg.K4 = kFC1; bg.K = kFC1; bg.N = kFlat;
    bg.p[0] = GemmProblem{w->dcat, kFC1, wT1, kFlat, nullptr, w->dfeat, kFlat, mask1, kFlat, w->feat, kFlat, nullptr, 0};
    gemm_launch(bg, 1, st);                   // through f1_drop and the d1 ReLUs in the epilogue
    fork();                                   // the masked dfeat is final
    // bias gradients of fc_2, FC1 and the three d1 layers: column sums of the (masked) upstream gradients, one launch
    {
        ColArgs ca{};
        ca.n = n;
        ca.p[0] = ColProblem{w->dR2, kFC2ld, kFC2, G + H.fc2b};
        ca.p[1] = ColProblem{w->dcat, kFC1, kFC1, G + H.fc1b};
        for (int t = 0; t < kTowers; ++t) {
            const TowerOff T = tower_off(t * kTowerFloats);
            ca.p[2 + t] = ColProblem{w->dfeat + t * kD1, kFlat, kD1, G + T.d1b};
        }
        hipLaunchKernelGGL(colsum_kernel, dim3((kFC1 + 63) / 64, kColProblems), dim3(256), 0, sd, ca);
    }
    // d1 x3: dW (sd), dX (st)
    bg.M = kFlat; bg.K4 = (n + 3) / 4 * 4; bg.K = n; bg.N = kD1;
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        bg.p[t] = GemmProblem{aT5 + (size_t)t * kFC1ld * ldT, ldT, w->dfeat + t * kD1, kFlat, nullptr, G + T.d1W, kD1};
    }
    gemm_launch(bg, kTowers, sd);
    bg.M = n; bg.K4 = kD1; bg.K = kD1; bg.N = kFlat;
    for (int t = 0; t < kTowers; ++t)         // through the dropout in front of d1 (mask5 of the tower) in the epilogue
        bg.p[t] = GemmProblem{w->dfeat + t * kD1, kFlat, wTd + t * kD1 * kFlat, kFlat, nullptr, w->dX[t][0], kFlat,
                              mask5 + (size_t)t * n * kFlat, kFlat, nullptr, 0, nullptr, 0};
    gemm_launch(bg, kTowers, st);

    // ---- backward: conv blocks 5..1 (dX ping-pongs between the two buffers)
    int cur = 0;
    bool dz_busy = false;                     // a weight-gradient kernel on the side stream still reads dZ
    for (int i = kConvs - 1; i >= 0; --i) {
        const int C = kCout[i], Hh = kHout[i];
        CPtr3 Zi = c3(w->Z, i), dOut{{w->dX[0][cur], w->dX[1][cur], w->dX[2][cur]}};
        Ptr3 dZ{{w->dZ[0], w->dZ[1], w->dZ[2]}};
        const long tot = (long)n * C * Hh * Hh;
        auto bn_bwd = [&](auto pool_tag, auto h_tag) {
            constexpr bool PL = decltype(pool_tag)::value;
            constexpr int HH = decltype(h_tag)::value;
            hipLaunchKernelGGL((bn_bwd_partial_kernel<PL, HH>), dim3(C, kTowers, kRedSplit), dim3(256), 0, st, Zi, dOut, n, C, w->stats, i, P,
                               T0.bn[i], kTowerFloats, w->red_part);
            const int split = max(1, min(n, (int)(tot / C / 8192)));           // ~8k elements of a channel per workgroup
            if (dz_busy) (void)hipStreamWaitEvent(st, w->ev_join, 0);          // previous layer's weight gradient has read dZ
            hipLaunchKernelGGL((bn_bwd_apply_kernel<PL, HH>), dim3(C, kTowers, split), dim3(256), 0, st, Zi, dOut, dZ, n, C, w->stats, i, P,
                               T0.bn[i], kTowerFloats, w->red_part, G);
        };
        using std::integral_constant;
        if (i == 0) bn_bwd(integral_constant<bool, false>{}, integral_constant<int, 30>{});
        else if (i == 1) bn_bwd(integral_constant<bool, true>{}, integral_constant<int, 28>{});
        else if (i == 2) bn_bwd(integral_constant<bool, false>{}, integral_constant<int, 12>{});
        else if (i == 3) bn_bwd(integral_constant<bool, true>{}, integral_constant<int, 10>{});
        else bn_bwd(integral_constant<bool, false>{}, integral_constant<int, 3>{});
        CPtr3 Xi;
        for (int t = 0; t < kTowers; ++t) Xi.p[t] = (i == 0) ? xin[t] : w->X[t][i];
        {
            // weight gradient of this layer on the side stream, beside the data gradient below; the next layer's
            // bn_bwd_apply overwrites dZ, so the caller's stream re-joins before it (top of the next iteration)
            fork();
            WgArgs wa{};
            for (int t = 0; t < kTowers; ++t) { wa.X[t] = Xi.p[t]; wa.dZ[t] = w->dZ[t]; }
            wa.part = w->wg_part; wa.n = n; wa.slices = n < wg_slices()[i] ? n : wg_slices()[i];
            if (i == 0) wgrad_launch<WG1>(wa, sd);
            else if (i == 1) wgrad_launch<WG2>(wa, sd);
            else if (i == 2) wgrad_launch<WG3>(wa, sd);
            else if (i == 3) wgrad_launch<WG4>(wa, sd);
            else wgrad_launch<WG5>(wa, sd);
            (void)hipEventRecord(w->ev_join, sd);               // dZ has been consumed
            const int tot = kCin[i] * C * 9;
            constexpr int kParts[kConvs] = {WG1::NB * WG1::PARTS, WG2::NB * WG2::PARTS, WG3::NB * WG3::PARTS, WG4::NB * WG4::PARTS, WG5::NB * WG5::PARTS};
            hipLaunchKernelGGL(wgrad_final_kernel, dim3((tot + 63) / 64, kTowers), dim3(256), 0, sd, w->wg_part, wa.slices * kParts[i], kCin[i], C, G,
                               T0.convW[i], kTowerFloats);
        }
        if (i > 0) {
            ConvArgs a{};
            a.n = n;
            for (int t = 0; t < kTowers; ++t) { a.in[t] = w->dZ[t]; a.w[t] = w->wdg[t][i]; a.bias[t] = nullptr; a.out[t] = w->dX[t][cur ^ 1]; }
            if (i == 4) conv_launch<DG5>(a, st);
            else if (i == 3) conv_launch<DG4>(a, st);
            else if (i == 2) conv_launch<DG3>(a, st);
            else conv_launch<DG2>(a, st);
            cur ^= 1;
        }
        dz_busy = true;
    }
    join();                                   // every gradient is in the arena before the caller's stream goes on
    SCS_HIP(hipGetLastError());
    h->packed_dirty = true;          // BN running statistics moved
    if (loss_host) {
        SCS_HIP(hipMemcpyAsync(loss_host, w->loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
        SCS_HIP(hipStreamSynchronize(st));
    }
    return SCS_OK;
}

int scs_train_loss(scs_handle h, float** dev_ptr) {
    SCS_ARG(h && dev_ptr, "null pointer");
    if (!h->tws || !h->tws->loss_dev) { set_error("scs_train_loss before scs_train_step"); return SCS_ERR_STATE; }
    *dev_ptr = h->tws->loss_dev;
    return SCS_OK;
}

int scs_adam_apply(scs_handle h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    if (!h->has_params) { set_error("scs_adam_apply before scs_set_params"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    h->adam_t += 1;
    const double t = (double)h->adam_t;
    const float a_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
    hipLaunchKernelGGL(adam_kernel, dim3(864), dim3(256), 0, (hipStream_t)stream, h->params, h->grads, h->adam_m, h->adam_v,
                       SCS_N_PARAMS, a_t, beta1, beta2, eps, grad_scale);
    SCS_HIP(hipGetLastError());
    h->packed_dirty = true;
    return SCS_OK;
}

int scs_eval_batch(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, const uint8_t* y,
                   int64_t n, double* loss_sum_host, int64_t* n_correct_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    SCS_ARG(loss_sum_host && n_correct_host, "null output pointer");
    *loss_sum_host = 0.0; *n_correct_host = 0;
    if (n == 0) return SCS_OK;
    SCS_ARG(ax && co && sa && atlas && y, "null input pointer");
    if (!h->has_params) { set_error("scs_eval_batch before scs_set_params"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    int rc = train_ensure(h, 1);
    if (rc != SCS_OK) return rc;
    const int estep = (int)(n < h->chunk ? (n + 15) / 16 * 16 : h->chunk);
    rc = ensure_workspace(h, estep);
    if (rc != SCS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->packed_dirty) { rc = repack_inference(h, st); if (rc != SCS_OK) return rc; h->packed_dirty = false; }
    TrainWs* w = h->tws;
    SCS_HIP(hipMemsetAsync(w->eval_acc, 0, 2 * sizeof(double), st));
    constexpr HeadOff H = head_off();
    for (int64_t s = 0; s < n; s += estep) {
        const int m = (int)((n - s < estep) ? (n - s) : estep);
        rc = forward_chunk(h, ax + s * 1024, co + s * 1024, sa + s * 1024, atlas + s * 15, nullptr, 0, 0, 0, nullptr, nullptr, m,
                           nullptr, nullptr, st, nullptr);
        if (rc != SCS_OK) return rc;
        // loss / accuracy from the fc_2 activations of this chunk (probs scratch = logits buffer)
        hipLaunchKernelGGL(softmax_ce_kernel, dim3((m + 255) / 256), dim3(256), 0, st, h->a_fc2, h->params + H.outW, h->params + H.outb,
                           y + s, m, h->a_logits, (float*)nullptr, (float*)nullptr, w->eval_acc);
    }
    double acc[2];
    SCS_HIP(hipMemcpyAsync(acc, w->eval_acc, sizeof(acc), hipMemcpyDeviceToHost, st));
    SCS_HIP(hipStreamSynchronize(st));
    *loss_sum_host = acc[0];
    *n_correct_host = (int64_t)(acc[1] + 0.5);
    return SCS_OK;
}

}  // extern "C"
