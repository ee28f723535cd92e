// Backward pass of the conv5 block (conv 3x3, 40 -> 60 channels, 5x5 -> 3x3) of the training step as ONE launch of two small
// GEMMs per workgroup.  Replaces, up to batch kC5MaxBatch, the data-gradient kernel ConvCfg<60,40,7,..> + the weight-gradient
// kernel WgCfg<40,60,5,..> of train_cfg.h (Theano autodiff of /root/reference/cnn_cort/nets.py:176-177).
//
// With 3x3 outputs the generic kernels waste most of their work: the data gradient convolves zero-padded 7x7 planes (2.8x the
// useful multiplications, 25 pixels = two 16-row tiles of which two of four waves own one each), the weight gradient contracts
// nine pixels per patch.  Here a workgroup owns G = 4 patch-towers = 36 (patch, pixel) rows and forms
//   d z[r][o] = [sc z + sh > 0] sc dUp + ca z + cb                                   once, in LDS (BN-backward form of conv_mfma.h)
//   D[(c,tap)][o] += sum_r X[g][c][y+dy][x+dx] d z[r][o]       M = 360, N = 60, K = 36   the weight-gradient partial of these patches
//   Y[r][(c,tap)]  = sum_o d z[r][o] Wc[o][c][tap]             M = 36, N = 360, K = 60   then col2im: dX[g][c][y+dy][x+dx] += Y
// on v_mfma_f32_16x16x4_f32 from LDS operands (the correlation-form weights Wc[o][c][tap] = W[o][c][8 - tap], packed by the prep
// launch, are copied to LDS once per workgroup together with everything else it reads), no padded planes anywhere; the epilogue leaves the (d beta, d gamma) sums of the conv4 block - dX through pool2 and the
// ReLU, against the raw conv4 output - one unit per workgroup.  Every sum has a fixed order: deterministic.
// Partials: part[tower][workgroup][c*9 + tap][o], reduced by wgrad_final_kernel with the other layers'.
#pragma once
#include "conv_mfma.h"

namespace scs {

constexpr int kC5G = 4;                       // patch-towers per workgroup: 192 workgroups at batch 256 (3 per workgroup would be 258: a second generation)
constexpr int kC5Rows = 48;                   // 36 (patch, pixel) rows padded to three 16-row tiles (rows 36.. are zero)
constexpr int kC5DzS = 66;                    // row stride of d z in LDS (floats)
constexpr int kC5WS = 368;                    // row stride of Wc / Y in LDS: 360 + 8 (368 % 32 == 16: the four k-rows of a fragment read hit distinct banks)
constexpr int kC5NW = 8, kC5Threads = 64 * kC5NW;
constexpr int kC5J = 360, kC5JT = 23;         // (c, tap) pairs, 16-wide tiles of them
constexpr int kC5MaxBatch = 1024;             // 256 partials per tower is what the partial buffer holds (kWgPartFloats)
constexpr int kC5WcFloats = 60 * kC5WS;       // the correlation-form weights of one tower as the prep launch packs them
constexpr int kC5DzFloats = kC5Rows * kC5DzS, kC5XFloats = kC5G * 1000, kC5XPad = (kC5XFloats + 7) / 8 * 8, kC5Z4Floats = kC5G * 4000;
constexpr int kC5RedFloats = kC5G * 40 * 5 * 2;
constexpr size_t kC5LdsBytes = (size_t)(kC5DzFloats + kC5XPad + kC5WcFloats + kC5RedFloats) * 4;   // (Z4: 16 000 floats, in Wc's space)
static_assert(kC5Z4Floats <= kC5WcFloats && kC5Threads * 3 * 2 <= kC5WcFloats, "the raw conv4 planes and the finalize scratch reuse the weights' space");
static_assert(kC5LdsBytes <= 160 * 1024, "LDS budget");
static_assert(8 * 60 * 2 * 2 <= kC5DzFloats && 64 * 4 <= kC5RedFloats, "part5 scratch: the per-part sums in the d z space, the coefficient table in s_red");

struct C5BArgs {
    const float* Z5[kTowers];     // raw conv5 output [n][60][9]
    const float* dH5[kTowers];    // gradient wrt the block-5 output [n][60][9]
    const float* cf5[kTowers];    // float4 per channel: sc, sh, ca, cb (bn_bwd_final_kernel<true>)
    const float* X5[kTowers];     // block-4 output = conv5 input [n][40][5][5]
    const float* Wc[kTowers];     // correlation-form conv5 weights [60][368]: Wc[o][c*9 + tap] = W[o][c][8 - tap], columns 360.. zero (prep launch)
    float* dX[kTowers];           // gradient wrt X5 [n][40][5][5]
    const float* Z4[kTowers];     // raw conv4 output [n][40][10][10]
    const float* low4[kTowers];   // float4 per channel of the conv4 block: sc, sh, mean, inv_std
    float* stats;                 // [tower][workgroups][40][2]: (sum dy, sum dy xhat) of the conv4 block, one unit per workgroup
    float* part;                  // [tower][workgroups][360][60]
    unsigned long long* stamps;   // diagnostic (SCS_C5_STAMPS): 8 time stamps (100 MHz) per workgroup, or null
    // part5 set (tail_fused): the conv5 block's own (d beta, d gamma) arrive as per-(row block, column) sums from the dense head's
    // last stage (dense_chain.h) instead of as the coefficient table cf5 of a launch in between: every workgroup adds them per
    // channel (fixed order, double: the same bits everywhere), forms (sc, sh, ca, cb) in LDS, and workgroup 0 of a tower
    // stores the two gradients
    const float* part5;           // [tower][blocks5][540] float2, or null
    int blocks5;
    const float* low5[kTowers];   // float4 per channel of the conv5 block: sc, sh, mean, inv_std
    float* gbn5[kTowers];         // (d beta, d gamma) slot of the conv5 block in the gradient arena
    FinArgs fin;                  // fin_on: the last workgroup of a tower turns the sums into the conv4 block's BN-backward coefficients
    int fin_on;                   // (conv_mfma.h: take_ticket / bn_grp_backward / bn_top_backward; sums then leave write-through)
    int n;
};

// MODE 0: everything in one launch (round 3).  MODE 1: the data path only (d z, data gradient, col2im, the conv4 block's sums) - what the
// caller's stream has to wait for; MODE 2: d z and the weight-gradient partial only, launched on the side stream beside the data
// gradients (the partial store and its GEMM are 6 us of the one-launch form's 27; the side stream is idle at that point).
template <bool P5, int MODE = 0>     // P5: the conv5 block's BN-backward coefficients from the dense head's column sums (a.part5)
__global__ __launch_bounds__(kC5Threads, 1) void conv5_backward_kernel(C5BArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem5[];
    float* s_dz = smem5;                                  // [48][66]
    float* s_x = s_dz + kC5DzFloats;                      // [4][40][25] X5, later dX
    float* s_w = s_x + kC5XPad;                           // [60][368] Wc, later Y [48][368], later the raw conv4 output [4][40][100]
    float* s_z4 = s_w;
    float* s_red = s_w + kC5WcFloats;                     // [4][40][5][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int tower = blockIdx.y, g0 = blockIdx.x * kC5G, ng = min(kC5G, a.n - g0);
    const float* __restrict__ z5 = pick3(a.Z5, tower) + (size_t)g0 * 540;
    const float* __restrict__ dh = pick3(a.dH5, tower) + (size_t)g0 * 540;
    const float4* __restrict__ cf = reinterpret_cast<const float4*>(pick3(a.cf5, tower));
    const float4* __restrict__ x5 = reinterpret_cast<const float4*>(pick3(a.X5, tower) + (size_t)g0 * 1000);
    const float4* __restrict__ wc = reinterpret_cast<const float4*>(pick3(a.Wc, tower));
    const float4* __restrict__ z4g = reinterpret_cast<const float4*>(pick3(a.Z4, tower) + (size_t)g0 * 4000);
    unsigned long long* stp = a.stamps ? a.stamps + ((size_t)tower * gridDim.x + blockIdx.x) * 8 : nullptr;
    auto stamp = [&](int i) { if (stp && tid == 0) stp[i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    constexpr int Z_IT = (kC5Z4Floats / 4 + kC5Threads - 1) / kC5Threads;          // 8
    float4 z4v[Z_IT];
    // ---- part5: the conv5 block's BN-backward coefficients from the dense head's column sums.  Thread (c, p): channel c = tid % 60,
    // part p = tid / 60 (8 parts) takes row blocks p, p + 8, .. - requested here, ahead of everything else (loads retire in order)
    constexpr int P5_MAXB = 3;                                 // row blocks per part: 8 x 3 = 24 blocks = batch 384 (the chain's bound)
    float2 p5v[P5_MAXB][9];
    float4 lo5 = {0.f, 0.f, 0.f, 0.f};
    constexpr bool p5 = P5;
    const int p5c = tid % 60, p5p = tid / 60;
    if constexpr (p5) {                                        // (straight-line loads from clamped addresses: no branch around the batch)
        const int cc = min(p5c, 59), pp = min(p5p, 7);
        const float2* __restrict__ ps = reinterpret_cast<const float2*>(a.part5) + (size_t)tower * a.blocks5 * 540 + cc * 9;
#pragma unroll
        for (int i = 0; i < P5_MAXB; ++i)
#pragma unroll
            for (int k = 0; k < 9; ++k) p5v[i][k] = ps[(size_t)min(pp + 8 * i, a.blocks5 - 1) * 540 + k];
        lo5 = reinterpret_cast<const float4*>(pick3(a.low5, tower))[cc];
    }

    // ---- stage: d z (formed here), X5, Wc - every load of a thread in flight before the first LDS write; the raw conv4 planes
    // the epilogue needs are requested right behind and wait in registers (their LDS space is Wc's, free after the col2im step)
    {
        constexpr int DZ_IT = kC5Rows * 64 / kC5Threads;                           // 6
        constexpr int X_IT = (kC5XFloats / 4 + kC5Threads - 1) / kC5Threads;       // 2
        constexpr int W_IT = (kC5WcFloats / 4 + kC5Threads - 1) / kC5Threads;      // 11
        float zv[DZ_IT], dv[DZ_IT];
        float4 cv[DZ_IT], xv[X_IT], wv[W_IT];
#pragma unroll
        for (int k = 0; k < DZ_IT; ++k) {
            const int e = tid + kC5Threads * k, r = e >> 6, o = e & 63;
            const bool ok = r < ng * 9 && o < 60;
            const int g = r / 9, px = r - g * 9;
            const int at = ok ? g * 540 + o * 9 + px : 0;
            zv[k] = z5[at]; dv[k] = dh[at]; if constexpr (p5) cv[k] = float4{0.f, 0.f, 0.f, 0.f}; else cv[k] = cf[ok ? o : 0];
        }
#pragma unroll
        for (int k = 0; k < X_IT; ++k) { const int e = tid + kC5Threads * k; xv[k] = (MODE != 1 && e < ng * 250) ? x5[e] : float4{0.f, 0.f, 0.f, 0.f}; }
        // (with part5 the weights - which only the second GEMM needs - are requested behind the coefficient step: held across its
        // barriers the compiler spilled all 44 registers of them to scratch)
        if constexpr (!p5 && MODE != 2) {
#pragma unroll
            for (int k = 0; k < W_IT; ++k) { const int e = tid + kC5Threads * k; wv[k] = wc[min(e, kC5WcFloats / 4 - 1)]; }
        }
        if constexpr (p5) {
            // sums per (channel, part) -> LDS (the d z space: nothing has been written there yet), the 8 parts in order, then the
            // coefficient table in the s_red space; bn_bwd_final_kernel's formulas
            double* s_p5 = reinterpret_cast<double*>(s_dz);                        // [8][60][2]
            float4* s_cf = reinterpret_cast<float4*>(s_red);                       // [64]
            {
                float fb = 0.f, fg = 0.f;                                          // a thread's <= 27 values in float, the parts in double
#pragma unroll
                for (int i = 0; i < P5_MAXB; ++i) {
                    const bool on = p5p + 8 * i < a.blocks5;
#pragma unroll
                    for (int k = 0; k < 9; ++k) { fb += on ? p5v[i][k].x : 0.f; fg += on ? p5v[i][k].y : 0.f; }
                }
                if (tid < 480) { s_p5[(p5p * 60 + p5c) * 2] = (double)fb; s_p5[(p5p * 60 + p5c) * 2 + 1] = (double)fg; }
            }
            __syncthreads();
            if (tid < 60) {
                double sb = 0, sg = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { sb += s_p5[(q * 60 + tid) * 2]; sg += s_p5[(q * 60 + tid) * 2 + 1]; }
                const double count = (double)a.n * 9.0, mb = sb / count, mg = sg / count, sc = lo5.x, inv = lo5.w, mean = lo5.z;
                s_cf[tid] = make_float4(lo5.x, lo5.y, (float)(-sc * inv * mg), (float)(sc * (inv * mean * mg - mb)));
                if (MODE != 2 && blockIdx.x == 0) { float* gb = pick3(a.gbn5, tower); gb[tid] = (float)sb; gb[60 + tid] = (float)sg; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DZ_IT; ++k) { const int o = (tid + kC5Threads * k) & 63; cv[k] = s_cf[min(o, 59)]; }
            __syncthreads();                                                       // s_p5 has been read: d z may be written
        }
#pragma unroll
        for (int k = 0; k < DZ_IT; ++k) {
            const int e = tid + kC5Threads * k, r = e >> 6, o = e & 63;
            const bool ok = r < ng * 9 && o < 60;
            const float dz = ((fmaf(zv[k], cv[k].x, cv[k].y) > 0.f) ? cv[k].x * dv[k] : 0.f) + fmaf(cv[k].z, zv[k], cv[k].w);
            s_dz[r * kC5DzS + o] = ok ? dz : 0.f;
        }
#pragma unroll
        for (int k = 0; k < X_IT; ++k) { const int e = tid + kC5Threads * k; if (e < kC5XFloats / 4) reinterpret_cast<float4*>(s_x)[e] = xv[k]; }
        if constexpr (MODE == 2) {
            // (the weight-gradient part does not touch the weights)
        } else if constexpr (!p5) {
#pragma unroll
            for (int k = 0; k < W_IT; ++k) { const int e = tid + kC5Threads * k; if (e < kC5WcFloats / 4) reinterpret_cast<float4*>(s_w)[e] = wv[k]; }
        } else {
            float4 w2[W_IT];                                                       // (its own registers: see above)
#pragma unroll
            for (int k = 0; k < W_IT; ++k) { const int e = tid + kC5Threads * k; w2[k] = wc[min(e, kC5WcFloats / 4 - 1)]; }
#pragma unroll
            for (int k = 0; k < W_IT; ++k) { const int e = tid + kC5Threads * k; if (e < kC5WcFloats / 4) reinterpret_cast<float4*>(s_w)[e] = w2[k]; }
        }
    }
    // (requested behind the staging loads: memory operations retire in order, so in front of them they would hold up the first LDS write)
#pragma unroll
    for (int k = 0; k < Z_IT; ++k) { const int e = tid + kC5Threads * k; z4v[k] = (MODE != 2 && e < ng * 1000) ? z4g[e] : float4{0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    stamp(1);

    // ---- weight-gradient partial: D[j][o] = sum_r Xcol[r][j] dz[r][o]; wave w owns row tiles w, w + 8, w + 16 of the 23
    if constexpr (MODE != 1) {
        constexpr int MTW = 3;
        f32x4 acc[MTW][4];
        int offj[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int j = min((wave + kC5NW * m) * 16 + lr, kC5J - 1), c = j / 9, t = j - c * 9;
            offj[m] = c * 25 + (t / 3) * 5 + (t % 3);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < kC5G * 9 / 4; ++s) {                          // K = the 36 real rows: nine steps (rows 36..47 are zero)
            const int r = 4 * s + kq, g = r / 9, px = r - g * 9;
            const int base = g * 1000 + (px / 3) * 5 + (px % 3);
            float b[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b[nt] = s_dz[r * kC5DzS + min(nt * 16 + lr, 63)];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                if (wave + kC5NW * m >= kC5JT) continue;
                const float av = s_x[base + offj[m]];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[m][nt], 0, 0, 0);
            }
        }
        float* __restrict__ out = a.part + ((size_t)tower * gridDim.x + blockIdx.x) * (size_t)(kC5J * 60);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int t = wave + kC5NW * m;
            if (t >= kC5JT) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int o = nt * 16 + lr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = t * 16 + kq * 4 + i;
                    if (j < kC5J && o < 60) out[j * 60 + o] = acc[m][nt][i];
                }
            }
        }
    }

    stamp(2);
    if constexpr (MODE == 2) return;
    // ---- data gradient: Y[r][j] = sum_o dz[r][o] Wc[o][j]; wave w owns column tiles w, w + 8, w + 16 and all row tiles
    {
        constexpr int NTW = 3, MT = kC5Rows / 16;
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 15; ++s) {
            float av[MT], bv[NTW];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = s_dz[(m * 16 + lr) * kC5DzS + 4 * s + kq];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bv[nt] = s_w[(4 * s + kq) * kC5WS + min(wave + kC5NW * nt, kC5JT - 1) * 16 + lr];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if (wave + kC5NW * nt >= kC5JT) continue;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[nt], acc[m][nt], 0, 0, 0);
            }
        }
        __syncthreads();                                                  // every wave is done with Wc: its space becomes Y
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int t = wave + kC5NW * nt;
            if (t >= kC5JT) continue;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) s_w[(m * 16 + kq * 4 + i) * kC5WS + t * 16 + lr] = acc[m][nt][i];
        }
    }
    __syncthreads();
    stamp(3);

    // ---- col2im: dX[g][c][y + dy][x + dx] += Y[(g, y, x)][c][dy][dx]; one thread per (patch, channel) walks its 9 x 9 values in
    // a fixed order into 25 registers (static indices: no address arithmetic, no branches)
    if (tid < kC5G * 40) {
        const int g = tid / 40, c = tid - g * 40;
        float o25[25];
#pragma unroll
        for (int p = 0; p < 25; ++p) o25[p] = 0.f;
        const float* yr = s_w + (g * 9) * kC5WS + c * 9;
#pragma unroll
        for (int px = 0; px < 9; ++px)
#pragma unroll
            for (int t = 0; t < 9; ++t) o25[(px / 3 + t / 3) * 5 + (px % 3 + t % 3)] += yr[px * kC5WS + t];
        float* __restrict__ dxo = pick3(a.dX, tower) + (size_t)(g0 + g) * 1000 + c * 25;
#pragma unroll
        for (int p = 0; p < 25; ++p) { s_x[tid * 25 + p] = o25[p]; if (g < ng) dxo[p] = o25[p]; }      // X5 is no longer needed
    }
    __syncthreads();                                                       // Y has been read: its space takes the raw conv4 planes
    stamp(4);
#pragma unroll
    for (int k = 0; k < Z_IT; ++k) { const int e = tid + kC5Threads * k; if (e < kC5Z4Floats / 4) reinterpret_cast<float4*>(s_z4)[e] = z4v[k]; }
    __syncthreads();

    // ---- (d beta, d gamma) sums of the conv4 block: dX through pool2 (every position that attains the window maximum, if it
    // survives the ReLU: Theano MaxPoolGrad) against the raw conv4 output; one thread per (patch, channel, window row), then the
    // five rows and the patches of a channel in a fixed order
    for (int e = tid; e < kC5G * 40 * 5; e += kC5Threads) {
        const int pc = e / 5, wy = e - pc * 5, g = pc / 40, c = pc - g * 40;
        float sm = 0.f, sq = 0.f;
        if (g < ng) {
            const float4 lo = reinterpret_cast<const float4*>(pick3(a.low4, tower))[c];       // sc, sh, mean, inv
            const float* zp = s_z4 + pc * 100 + (2 * wy) * 10;
            const float* dp = s_x + pc * 25 + wy * 5;
#pragma unroll
            for (int wx = 0; wx < 5; ++wx) {
                const float z0 = zp[2 * wx], z1 = zp[2 * wx + 1], z2 = zp[10 + 2 * wx], z3 = zp[10 + 2 * wx + 1];
                const float y0 = fmaf(z0, lo.x, lo.y), y1 = fmaf(z1, lo.x, lo.y), y2 = fmaf(z2, lo.x, lo.y), y3 = fmaf(z3, lo.x, lo.y);
                const float m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
                const float dv = m > 0.f ? dp[wx] : 0.f;
                const float d0 = y0 == m ? dv : 0.f, d1 = y1 == m ? dv : 0.f, d2 = y2 == m ? dv : 0.f, d3 = y3 == m ? dv : 0.f;
                sm += (d0 + d1) + (d2 + d3);
                sq = fmaf(d0, (z0 - lo.z) * lo.w, sq); sq = fmaf(d1, (z1 - lo.z) * lo.w, sq);
                sq = fmaf(d2, (z2 - lo.z) * lo.w, sq); sq = fmaf(d3, (z3 - lo.z) * lo.w, sq);
            }
        }
        s_red[2 * e] = sm; s_red[2 * e + 1] = sq;
    }
    __syncthreads();
    stamp(5);
    if (tid < 40) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int g = 0; g < kC5G; ++g)
#pragma unroll
            for (int wy = 0; wy < 5; ++wy) { sm += s_red[2 * ((g * 40 + tid) * 5 + wy)]; sq += s_red[2 * ((g * 40 + tid) * 5 + wy) + 1]; }
        float2* dst = reinterpret_cast<float2*>(a.stats + (((size_t)tower * gridDim.x + blockIdx.x) * 40 + tid) * 2);
        if (a.fin_on) st_sc1(dst, make_float2(sm, sq)); else *dst = make_float2(sm, sq);
    }
    stamp(6);
    if (a.fin_on) {
        __syncthreads();                                                   // s_red has been read: the LDS is free
        double* s_acc = reinterpret_cast<double*>(s_w);                    // [512][3]
        volatile int* s_flag = reinterpret_cast<volatile int*>(s_red);
        const unsigned int nwg = gridDim.x, grp = blockIdx.x / kFinGroup, groups = (nwg + kFinGroup - 1) / kFinGroup;
        unsigned int* cnt = a.fin.counter + tower * fin_counter_words(nwg);
        if (take_ticket(cnt + (size_t)kFinLine * (1 + grp), min((unsigned int)kFinGroup, nwg - grp * kFinGroup), s_flag)) {
            const float2* part = reinterpret_cast<const float2*>(a.stats + (size_t)tower * nwg * 40 * 2);
            double* grec = a.fin.grp + ((size_t)tower * groups + grp) * kFinRec * 2;
            bn_grp_backward<40>(part, (int)(grp * kFinGroup), (int)min((unsigned int)kFinGroup, nwg - grp * kFinGroup), grec);
            if (take_ticket(cnt, groups, s_flag)) {
                const double* rec = a.fin.grp + (size_t)tower * groups * kFinRec * 2;
                bn_top_backward<40, kC5Threads>(rec, (int)groups, a.fin.count, reinterpret_cast<const float4*>(pick3(a.low4, tower)),
                                                pick3(a.fin.bn, tower), reinterpret_cast<float4*>(pick3(a.fin.coef, tower)), s_acc);
            }
        }
        stamp(7);
    }
}

inline int conv5_backward_workgroups(int n) { return (n + kC5G - 1) / kC5G; }

template <bool P5, int MODE>
inline hipError_t conv5_backward_set_attr_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_backward_kernel<P5, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC5LdsBytes);
}
inline hipError_t conv5_backward_set_attr() {
    hipError_t e = conv5_backward_set_attr_one<false, 0>();
    if (e == hipSuccess) e = conv5_backward_set_attr_one<true, 0>();
    if (e == hipSuccess) e = conv5_backward_set_attr_one<true, 1>();
    if (e == hipSuccess) e = conv5_backward_set_attr_one<true, 2>();
    return e;
}

// mode 0: one launch; 1 / 2: the data path / the weight-gradient partial as launches of their own (part5 form only)
inline void conv5_backward_launch(const C5BArgs& a, hipStream_t st, int mode = 0) {
    const dim3 grid(conv5_backward_workgroups(a.n), kTowers), block(kC5Threads);
    if (!a.part5) hipLaunchKernelGGL((conv5_backward_kernel<false, 0>), grid, block, kC5LdsBytes, st, a);
    else if (mode == 1) hipLaunchKernelGGL((conv5_backward_kernel<true, 1>), grid, block, kC5LdsBytes, st, a);
    else if (mode == 2) hipLaunchKernelGGL((conv5_backward_kernel<true, 2>), grid, block, kC5LdsBytes, st, a);
    else hipLaunchKernelGGL((conv5_backward_kernel<true, 0>), grid, block, kC5LdsBytes, st, a);
}

}  // namespace scs
