// The dense head of the training step as ONE launch (small batches): d1 x3 -> FC1 -> fc_2 -> out + softmax + cross-entropy ->
// fc_2 / FC1 / d1 data gradients.  Replaces, for the batch sizes nolearn trains at, the seven dependent GEMM launches + the head
// launch that scs_train.hip issues otherwise (DenseLayer nodes of /root/reference/cnn_cort/nets.py:179, 217, 227, 231 and their
// Theano gradients): at batch 256 each of those is 144 workgroups that live for ONE global round trip and 80 MFMAs per wave -
// 8 - 10 us as a launch of its own, about 1.5 us of work.
//
// The chain is row-local: rows [16 b, 16 b + 16) of every layer depend on the same rows of the layer before and on nothing else.
// A CLUSTER of 9 workgroups owns one block of 16 rows; member j computes column block j (64 columns) of each layer - the tiling,
// the 8-wave K split and the order of every sum are those of gemm_splitk4_kernel / head_train_kernel, so both forms of the step
// give the same bits (tests/test_gpu_train.py).  Between layers the members exchange their columns through global memory and
// meet at a cluster barrier.
//
// A hand-off between XCDs costs ~2 us per hop on this GPU (write-through store, memory-side atomic, device-scope load: measured,
// profiles/r03_notes.txt - the chain then takes as long as the launches it replaces), one inside an XCD 0.7 - 0.9 us for the whole
// barrier: the L2 of an XCD is shared by its CUs.  So a cluster is formed INSIDE one XCD, whatever the dispatcher does: a
// workgroup reads the XCC_ID register, takes a ticket from THAT XCD's counter (an L2 atomic; only workgroups of this XCD ever
// touch the line) and becomes member (ticket % 9) of this XCD's cluster (ticket / 9), which owns row block 8 * cluster + xcd.
// With the round-robin dispatch of gfx942 / gfx950 every XCD receives grid / 8 workgroups = whole clusters.  Inside a cluster:
//   * outputs are plain stores (the CU's L1 writes through to the XCD's L2); the storing waves drain them (vmcnt(0)) in front of
//     the workgroup barrier behind which ONE lane adds its arrival to the cluster's counter - an L2 atomic - and polls it with
//     returning L2 atomics (a load could be served by the CU's L1 for ever);
//   * operands written by other members are loaded behind that barrier, for the first time in this launch by this CU, so the L1
//     cannot hold them (it is invalid at kernel start); a 16-row block of every exchanged array is a whole number of 128-byte
//     lines (row strides 540, 556, 272, 16 floats from 256-byte aligned bases), so no line is shared with another cluster;
//   * the counters are zeroed by the prep launch at the start of every step.  A poll that does not complete within
//     kChainSpinLimit rounds (an XCD that did not receive whole clusters) gives up, poisons the loss of its rows with NaN and lets
//     the kernel end; the row losses are NaN from the prep launch until a cluster writes them, and the host takes this form only
//     after a probe launch on the device (chain_placement_probe): wrong placement is loud, never silent.
// The grid (72 x ceil(row blocks / 8) workgroups of 512 threads, at most kChainMaxBatch rows) is co-resident by size.  Tickets are
// taken in arrival order, so the first nine workgroups of a launch that become resident on an XCD always form a complete
// cluster, finish and free their CUs: two training processes sharing one GPU (whose launches together may exceed the CUs) still
// make progress - one of them holds at least nine of an XCD's 32 slots; four or more could starve each other into the spin limit.
//
// What it buys (profiles/r03_chain_stamps.txt, r03_notes.txt): a 16 x 64 x 540 tile costs 7.5 - 8 us in either form - 2.7 us until
// the first wave has its operands (1.4 us when they are hot), 3.5 us until the last wave is through its 80 MFMAs (two waves share a
// SIMD's matrix pipe: 2.4 us is the floor of such a tile on one CU), 1 us reduction + stores, 0.9 us drain + barrier against 2 - 3 us
// between dependent launches: with the layers one tile each the chain takes as long as the launches (65 - 68 us).  It pays through
// what a launch cannot do: the three d1 data gradients of a member (K = 180 each: six of eight waves busy, three operand round
// trips as launches or tiles) are ONE pass of 18 (tower, wave-partial) jobs over the 8 waves (chain_d1_dgrad), and the out layer +
// loss - a stage that kept one member of nine busy - is computed by every member in front of its fc_2 data-gradient tile
// (chain_head_dcat): the chain drops to ~48 us and the step by 3 % at batch 256 (0.582 against 0.600 ms) and 4 % at 128 (0.438
// against 0.455), 2.7 % at 320 and 384 - shipped up to batch 384 (what the co-resident grid allows).
// Prefetching the weights - from the matrix waves, from extra waves, from spare workgroups on the idle CUs - did not move the
// operand phase and is not kept.
#pragma once
#include "gemm_mfma.h"

// XCC_ID (hardware register 20), an L2 shared by the CUs of an XCD, vector stores counted by vmcnt: the gfx942 / gfx950 family
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "dense_chain.h: the XCD-local cluster hand-off relies on gfx942/gfx950 hardware (XCC_ID, per-XCD L2, vmcnt-counted stores)"
#endif

namespace scs {

constexpr int kChainMembers = 9;                 // column blocks of 64 of a 540-wide layer = workgroups per cluster
constexpr int kChainMaxBatch = 384;              // 24 clusters = 216 workgroups <= 256 CUs
constexpr int kChainDefaultMaxBatch = kChainMaxBatch;   // the step takes this form up to here unless SCS_DENSE_CHAIN_MAX_BATCH says otherwise
constexpr int kChainLine = 32;                   // counter stride in words: one 128-byte line per counter
constexpr int kXcds = 8;
constexpr int kChainClustersPerXcd = (kChainMaxBatch / 16 + kXcds - 1) / kXcds;
constexpr int kChainCounterWords = kXcds * (1 + kChainClustersPerXcd) * kChainLine;   // per XCD: ticket line, then one barrier line per cluster
constexpr unsigned int kChainSpinLimit = 1u << 20;

struct DenseChainArgs {
    GemmProblem d1[kTowers], fc1, fc2;           // forward problems (as scs_train.hip hands them to gemm_launch)
    GemmProblem dcat, dfeat, dh5[kTowers];       // data-gradient problems
    const float* outW; const float* outb;        // out layer [270][15], [15]
    const uint8_t* y;                            // batch labels
    float* probs; float* dlog; float* loss_rows; // [n][16], [n][16], [n]
    float* R2; float* dR2;                       // [n][kFC2ld]
    // conv5 block's batch norm folded into the chain (tail_fused; units[0] == null: H5 comes from the launch before, as round 3):
    // every workgroup finalizes the statistics of ITS tower from the per-workgroup units conv5's forward kernel left (<= 96 units
    // of 36 values: one round of loads, double arithmetic in a fixed order - every workgroup gets the same bits), then
    // applies BN + ReLU + the dropout in front of d1 to the raw conv5 rows while they are its d1 A operand; the member that
    // owns column block 0 of a tower also stores those rows as H5 (the d1 weight gradient reads them) and, in row block 0,
    // the (scale, shift) / (sc, sh, mean, inv_std) tables and the running averages (nets.py:177-178)
    struct Bn5 {
        const float* units[kTowers];             // [nwg][60] (mean, M2) float2
        float* bn[kTowers];                      // beta, gamma, mean, inv_std of the conv5 block in the parameter arena
        float* ss[kTowers];                      // float2 [60] out
        float* low4[kTowers];                    // float4 [60] out
        const uint8_t* mask[kTowers];            // dropout keep-flags [n][540]
        float* H5[kTowers];                      // [n][540] out
        int nwg;
    } bn5;
    // conv5 block's batch-norm BACKWARD sums folded into the last stage (part5 == null: bn_bwd_final_kernel<true> as round 3):
    // per (tower, row block, column) the sums over the block's 16 rows of dy and dy * xhat (dy = dH5 through the ReLU, from the raw
    // conv5 output Z5); conv5_backward_kernel's prologue adds them per channel
    float* part5;                                // [tower][row blocks][540] float2
    const float* Z5[kTowers];                    // raw conv5 output [n][540]
    float* clow;                                 // [row blocks][3][60] float4: the towers' (sc, sh, mean, inv_std) handed round inside a cluster
    unsigned int* counters;                      // kChainCounterWords, zero at launch
    unsigned long long* stamps;                  // diagnostic (SCS_CHAIN_STAMPS): kChainStamps 100 MHz time stamps per workgroup, or null
    int n;
};
constexpr int kChainStamps = 64;
struct ChainStamper {
    unsigned long long* p; int i;
    __device__ __forceinline__ void seek(int to) { i = to; }
    __device__ __forceinline__ void hit() { if (p && threadIdx.x == 0) { p[i] = __builtin_amdgcn_s_memrealtime(); ++i; } }
    __device__ __forceinline__ void drained() { if (p) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); hit(); } }
};

// XCC_ID of the XCD this wave runs on (hardware register 20, bits 3:0)
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

// L2 atomics of one XCD (no device scope: they execute in the XCD's L2 and are coherent for the workgroups of that XCD)
__device__ __forceinline__ unsigned int l2_add(unsigned int* p, unsigned int v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned int l2_read(unsigned int* p) {        // a compare-and-swap that never swaps: the current value, from L2
    unsigned int expected = 0xFFFFFFFFu;
    __hip_atomic_compare_exchange_strong(p, &expected, 0xFFFFFFFFu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expected;
}

// Arrival of this workgroup at barrier number `round` (1-based) of its cluster, then wait for all kChainMembers.
// Returns false in every thread when the wait gave up.
// ROLE of a wave: 0 = waves 0..3 (matrix work, reduction, stores - they drain their stores here), 1 = waves 4..7 (matrix work
// only).  Both roles execute the same workgroup barriers in the same order.
template <int ROLE>
__device__ __forceinline__ bool chain_barrier(unsigned int* ctr, unsigned int round, volatile int* s_flag, ChainStamper& ts) {
    if (ROLE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are in the XCD's L2
    __syncthreads();
    ts.hit();                                                   // [4] stores drained, workgroup together
    if (threadIdx.x == 0) {
        const unsigned int target = round * kChainMembers;
        unsigned int seen = l2_add(ctr, 1u) + 1u;
        unsigned int spins = 0;
        while (seen < target && spins < kChainSpinLimit) {
            __builtin_amdgcn_s_sleep(1);
            seen = l2_read(ctr);
            ++spins;
        }
        *s_flag = seen >= target ? 1 : 0;
    }
    __syncthreads();
    ts.hit();                                                   // [5] cluster together
    return *s_flag != 0;
}

// One 16 x 64 output tile of C = epilogue(A W + b): gemm_splitk4_kernel's arithmetic (same K split over the 8 matrix waves, same
// order of sums).  Straight-line loads from clamped (always valid) addresses; what lies outside the problem becomes zero by
// select.  s_r: 8 * 4 * 64 f32x4 of LDS.
// what the BN5 form of the first tile needs beside its GemmProblem (A = the raw conv5 rows)
struct ChainBn5Tile {
    const float2* units; float* bn; float* ss; float* low4; const uint8_t* mask; float* H5; float* clow;   // clow: this cluster's [60] float4 of the tower
    int nwg, n; bool store_h5, store_tables;
    double* s_part;        // LDS [8][64][3]
    float2* s_sc;          // LDS [64]
    float2* s_bn;          // LDS [544]: (scale, shift) per k = (channel, pixel)
};
constexpr int kBn5MaxUnits = 12;                 // units per thread: 8 parts x 12 = 96 workgroups of 4 patches = batch 384

template <int ROLE, bool BN5 = false>
__device__ __forceinline__ void chain_tile(const GemmProblem& P, int M, int K4, int K, int N, int relu, int m0, int n0,
                                           f32x4* s_r, ChainStamper& ts, const ChainBn5Tile* bx = nullptr) {
    ts.hit();                                                   // [0] tile start
    using T = GemmSplitK4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int ecol = n0 + 4 * lr + wave;
    f32x4 acc[4];
    float e_bias_raw = 0.f, e_gate[4] = {1.f, 1.f, 1.f, 1.f};
    unsigned char e_mask[4] = {1, 1, 1, 1};
    // BN5: the statistics units of this tower, requested FIRST (loads retire in order: they are back before the operands)
    float2 bu[BN5 ? kBn5MaxUnits : 1];
    float2 bu0 = make_float2(0.f, 0.f);
    float b_beta = 0.f, b_gamma = 0.f, b_mean = 0.f, b_inv = 0.f;
    if constexpr (BN5) {
        const int c = min(tid & 63, 59), p = tid >> 6;
        bu0 = bx->units[c];
#pragma unroll
        for (int i = 0; i < kBn5MaxUnits; ++i) bu[i] = bx->units[(size_t)min(p + 8 * i, bx->nwg - 1) * 60 + c];
        if (tid < 60) { b_beta = bx->bn[tid]; b_gamma = bx->bn[60 + tid]; b_mean = bx->bn[120 + tid]; b_inv = bx->bn[180 + tid]; }
    }
    {
        const int kmax = K ? K : K4;
        const int groups = (K4 + 15) / 16, per = (groups + T::NW - 1) / T::NW;  // per <= GMAX: K4 <= 640 (checked on the host)
        const int g_beg = min(groups, wave * per), g_end = min(groups, g_beg + per);
        const int row = m0 + lr;
        const bool row_ok = row < M;
        const float* arow = P.A + (size_t)(row_ok ? row : 0) * P.lda;
        const bool col_ok = n0 + 4 * lr < P.NS;
        const float* __restrict__ wcol = P.W + min(n0 + 4 * lr, P.NS - 4);
        float4 av[T::GMAX], wv[T::GMAX][4];
        unsigned int am[BN5 ? T::GMAX : 1];                     // BN5: four dropout keep-flags beside every four raw values
#pragma unroll
        for (int u = 0; u < T::GMAX; ++u) {
            const bool on = (g_beg + u) < g_end;
            const int k = on ? 16 * (g_beg + u) + 4 * kq : 0;   // a group this wave does not own: rows 0..3 (cached)
            av[u] = *reinterpret_cast<const float4*>(arow + (k < K4 ? k : 0));
            if constexpr (BN5) am[u] = *reinterpret_cast<const unsigned int*>(bx->mask + (size_t)(row_ok ? row : 0) * kFlat + (k < K4 ? k : 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[u][i] = *reinterpret_cast<const float4*>(wcol + (size_t)min(k + i, kmax - 1) * P.NS);
        }
        if (ROLE == 0) {
            // bias, dropout mask and ReLU gate of the outputs this wave will store, requested with the operands (absent operands:
            // a load from a valid address whose value is ignored - a branch here would serialise the loads)
            const int ecl = min(ecol, N - 1);
            const float* __restrict__ bp = P.bias ? P.bias + ecl : P.W;
            const unsigned char* __restrict__ mp = P.mask ? P.mask + ecl : reinterpret_cast<const unsigned char*>(P.W);
            const float* gp = P.gate ? P.gate + ecl : P.W;
            const int ldm = P.mask ? P.ldm : 0, ldg = P.gate ? P.ldg : 0;
            e_bias_raw = *bp;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = min(m0 + kq * 4 + r, M - 1);
                e_mask[r] = mp[(size_t)orow * ldm];
                e_gate[r] = gp[(size_t)orow * ldg];
            }
        }
        if constexpr (BN5) {
            // ---- finalize the tower's batch statistics (every workgroup of the tower does, in the same order: the same bits):
            // weighted parallel-variance combine of the units about unit 0's mean, eight strided sub-sums per channel added in order
            {
                const int c = tid & 63, p = tid >> 6;
                const double m0 = (double)bu0.x;
                double sw = 0, sw2 = 0, q = 0;
#pragma unroll
                for (int i = 0; i < kBn5MaxUnits; ++i) {
                    const int uu = p + 8 * i;
                    if (uu < bx->nwg) {
                        const double wgt = 9.0 * (double)min(4, bx->n - 4 * uu), d = (double)bu[i].x - m0;
                        sw += wgt * d; sw2 += wgt * d * d; q += (double)bu[i].y;
                    }
                }
                double* sp = bx->s_part + (p * 64 + c) * 3;
                sp[0] = sw; sp[1] = sw2; sp[2] = q;
            }
            __syncthreads();
            if (tid < 60) {
                double sw = 0, sw2 = 0, q = 0;
#pragma unroll
                for (int p = 0; p < 8; ++p) { const double* sp = bx->s_part + (p * 64 + tid) * 3; sw += sp[0]; sw2 += sp[1]; q += sp[2]; }
                const double cnt = 9.0 * (double)bx->n, dm = sw / cnt, mean = (double)bu0.x + dm;
                const double var = (q + fmax(sw2 - sw * dm, 0.0)) / cnt;
                const double inv = 1.0 / sqrt(var + (double)kBnEps);
                const float sc = b_gamma * (float)inv, sh = b_beta - (float)mean * sc;
                bx->s_sc[tid] = make_float2(sc, sh);
                if (bx->clow) reinterpret_cast<float4*>(bx->clow)[tid] = make_float4(sc, sh, (float)mean, (float)inv);   // for this cluster's last stage
                if (bx->store_tables) {
                    reinterpret_cast<float2*>(bx->ss)[tid] = make_float2(sc, sh);
                    reinterpret_cast<float4*>(bx->low4)[tid] = make_float4(sc, sh, (float)mean, (float)inv);
                    bx->bn[120 + tid] = (1.f - kBnAlpha) * b_mean + kBnAlpha * (float)mean;
                    bx->bn[180 + tid] = (1.f - kBnAlpha) * b_inv + kBnAlpha * (float)inv;
                }
            }
            __syncthreads();
            // ---- the A operand: raw conv5 values -> BN + ReLU + dropout (bn_fwd_final_kernel<true>'s expression); H5 as a by-product.
            // Four consecutive k = (channel, pixel) indices touch at most two channels: c0 = k / 9 and, from element 9 (c0 + 1) - k on, c0 + 1
#pragma unroll
            for (int u = 0; u < T::GMAX; ++u) {
                const bool on = (g_beg + u) < g_end;
                const int k = 16 * (g_beg + u) + 4 * kq;
                if (on && k < K4) {
                    const int c0 = (k * 7282) >> 16;                          // k / 9 for k < 16384
                    const int nb = 9 * (c0 + 1) - k;                          // elements 0 .. nb - 1 belong to channel c0
                    const float2 ca = bx->s_sc[c0], cb = bx->s_sc[min(c0 + 1, 59)];
                    const unsigned int mk = am[u];
                    float4 h;
                    h.x = (mk & 0xFFu) ? 2.f * fmaxf(fmaf(av[u].x, ca.x, ca.y), 0.f) : 0.f;
                    h.y = (mk & 0xFF00u) ? 2.f * fmaxf(nb > 1 ? fmaf(av[u].y, ca.x, ca.y) : fmaf(av[u].y, cb.x, cb.y), 0.f) : 0.f;
                    h.z = (mk & 0xFF0000u) ? 2.f * fmaxf(nb > 2 ? fmaf(av[u].z, ca.x, ca.y) : fmaf(av[u].z, cb.x, cb.y), 0.f) : 0.f;
                    h.w = (mk & 0xFF000000u) ? 2.f * fmaxf(nb > 3 ? fmaf(av[u].w, ca.x, ca.y) : fmaf(av[u].w, cb.x, cb.y), 0.f) : 0.f;
                    av[u] = h;
                    if (bx->store_h5 && row_ok) *reinterpret_cast<float4*>(bx->H5 + (size_t)row * kFlat + k) = h;
                }
            }
        }
        ts.drained();                                           // [1] operands have arrived (diagnostic runs only)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < T::GMAX; ++u) {
            const bool on = (g_beg + u) < g_end;
            const int k = 16 * (g_beg + u) + 4 * kq;
            const float4 a = (on && row_ok && k < K4) ? av[u] : zero4;
            const float a4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w = (on && col_ok && k + i < kmax) ? wv[u][i] : zero4;
                const float w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], w4[j], acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                            // s_r may still be read by the previous tile's epilogue
    ts.hit();                                                   // [2] matrix work done in every wave
#pragma unroll
    for (int j = 0; j < 4; ++j) s_r[(wave * 4 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (ROLE == 0) {
        const int j = wave;
        f32x4 v = s_r[(0 * 4 + j) * 64 + lane];
#pragma unroll
        for (int q = 1; q < T::NW; ++q) { const f32x4 u = s_r[(q * 4 + j) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        if (ecol < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                if (orow < M) {
                    float x = v[r] + (P.bias ? e_bias_raw : 0.f);
                    if (relu) x = fmaxf(x, 0.f);
                    if (P.C2) {
                        P.C[(size_t)orow * P.ldc + ecol] = x;
                        P.C2[(size_t)orow * P.ldc2 + ecol] = e_mask[r] ? 2.f * x : 0.f;
                    } else {
                        if (P.mask) x = e_mask[r] ? 2.f * x : 0.f;
                        if (P.gate) x = e_gate[r] > 0.f ? x : 0.f;
                        P.C[(size_t)orow * P.ldc + ecol] = x;
                    }
                }
            }
        }
    }
    ts.hit();                                                   // [3] outputs issued
}

// out layer + loss + fc_2's data gradient as ONE stage.  The head of 16 rows is a few hundred MFMAs: as a stage of its own it ran
// on one of the nine members while eight waited (7.8 us with its barrier).  Here EVERY member computes the head of its cluster's
// rows (logits, softmax, d logits, dR2 = d logits outW' gated by R2) into LDS - member 0 also writes probs, d logits, the row losses
// and dR2 for the weight gradients - and goes straight on with its dcat tile, A = dR2 from LDS; the tile's weights are requested
// behind the head's own operands (loads retire in order) and arrive while the head computes.  Arithmetic and order of sums are
// head_train_kernel's and gemm_splitk4_kernel's.  s_dr2: [16][kFC2ld] floats of LDS behind the partial tiles.
template <int ROLE>
__device__ __forceinline__ void chain_head_dcat(const DenseChainArgs& a, int j, int m0, int n0, f32x4* s_r, float (*s_d)[17], float* s_dr2,
                                                ChainStamper& ts) {
    ts.hit();
    using T = GemmSplitK4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int n = a.n;
    const bool writer = j == 0;
    // ---- the head's operands (waves 0..3: their K slice of R2 and outW)
    constexpr int SPW = 17;
    float hav[SPW], hbv[SPW];
    if (wave < 4) {
        const float* ap = a.R2 + (size_t)min(m0 + lr, n - 1) * kFC2ld + kq;
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int k = 4 * (wave * SPW + i) + kq;
            hav[i] = ap[k < kFC2 ? 4 * (wave * SPW + i) : 0];
            hbv[i] = a.outW[min(k, kFC2 - 1) * 15 + min(lr, 14)];
        }
    }
    // ---- the dcat tile's weights and epilogue operands, behind them
    const GemmProblem& P = a.dcat;
    constexpr int K4 = kFC2ld, KM = kFC2, N = kFC1;
    constexpr int groups = (K4 + 15) / 16, per = (groups + T::NW - 1) / T::NW;
    const int g_beg = min(groups, wave * per), g_end = min(groups, g_beg + per);
    const bool col_ok = n0 + 4 * lr < P.NS;
    const float* __restrict__ wcol = P.W + min(n0 + 4 * lr, P.NS - 4);
    float4 wv[per][4];
#pragma unroll
    for (int u = 0; u < per; ++u) {
        const int k = (g_beg + u) < g_end ? 16 * (g_beg + u) + 4 * kq : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[u][i] = *reinterpret_cast<const float4*>(wcol + (size_t)min(k + i, KM - 1) * P.NS);
    }
    const int ecol = n0 + 4 * lr + wave;
    float e_gate[4] = {1.f, 1.f, 1.f, 1.f};
    unsigned char e_mask[4] = {1, 1, 1, 1};
    if (ROLE == 0) {
        const int ecl = min(ecol, N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = min(m0 + kq * 4 + r, n - 1);
            e_mask[r] = P.mask[(size_t)orow * P.ldm + ecl];
            e_gate[r] = P.gate[(size_t)orow * P.ldg + ecl];
        }
    }
    // ---- logits: wave w < 4 owns k-steps [17 w, 17 w + 17) of the 68
    __syncthreads();                                            // s_r: the previous tile's epilogue is done with it
    if (wave < 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int k = 4 * (wave * SPW + i) + kq;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k < kFC2 ? hav[i] : 0.f, (k < kFC2 && lr < 15) ? hbv[i] : 0.f, acc, 0, 0, 0);
        }
        s_r[wave * 64 + lane] = acc;
    }
    __syncthreads();
    if (tid < 256) {
        const int r = tid >> 4, c = tid & 15;
        const int row = m0 + r;
        const int src = (r >> 2) * 16 + c;
        const float z = (c < 15 ? a.outb[c] : 0.f) + ((s_r[0 * 64 + src][r & 3] + s_r[1 * 64 + src][r & 3]) + (s_r[2 * 64 + src][r & 3] + s_r[3 * 64 + src][r & 3]));
        float mx = c < 15 ? z : -3.0e38f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = c < 15 ? expf(z - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float pr = e / sum;
        const int yi = row < n ? a.y[row] : 0;
        const float dl = (c < 15 && row < n) ? (pr - (c == yi ? 1.f : 0.f)) / (float)n : 0.f;
        s_d[r][c] = dl;
        if (writer && row < n) {
            a.probs[(size_t)row * 16 + c] = c < 15 ? pr : 0.f;
            a.dlog[(size_t)row * 16 + c] = dl;
            if (c == yi) a.loss_rows[row] = -(z - mx - logf(sum));
        }
    }
    __syncthreads();
    {   // dR2: 17 column tiles of 16 over the 8 waves; K = the 15 classes (4 k-steps)
        float a4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a4[s4] = s_d[lr][4 * s4 + kq];
        for (int t = wave; t < 17; t += 8) {
            const int col = 16 * t + lr;
            float w4[4], gate[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { const int cl = 4 * s4 + kq; w4[s4] = a.outW[min(col, kFC2 - 1) * 15 + min(cl, 14)]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) gate[i] = a.R2[(size_t)min(m0 + 4 * kq + i, n - 1) * kFC2ld + min(col, kFC2 - 1)];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], (col < kFC2 && 4 * s4 + kq < 15) ? w4[s4] : 0.f, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int orow = m0 + 4 * kq + i;
                const float v = (col < kFC2 && gate[i] > 0.f) ? acc[i] : 0.f;       // columns 270, 271: zero
                s_dr2[(4 * kq + i) * kFC2ld + col] = v;
                if (writer && orow < n) a.dR2[(size_t)orow * kFC2ld + col] = v;
            }
        }
    }
    __syncthreads();
    ts.hit();
    // ---- the dcat tile: A = dR2 rows from LDS
    f32x4 acc[4];
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) acc[jc] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool row_ok = m0 + lr < n;
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < T::GMAX; ++u) {                         // (the slots beyond this stage's `per` multiply zeros, as in the stand-alone kernel)
        const bool on = u < per && (g_beg + u) < g_end;
        const int k = 16 * (g_beg + u) + 4 * kq;
        float4 av = zero4;
        if (u < per) av = *reinterpret_cast<const float4*>(s_dr2 + lr * kFC2ld + (on && k < K4 ? k : 0));
        const float4 a4v = (on && row_ok && k < K4) ? av : zero4;
        const float a4[4] = {a4v.x, a4v.y, a4v.z, a4v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 w = (u < per && on && col_ok && k + i < KM) ? wv[u < per ? u : 0][i] : zero4;
            const float w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int jc = 0; jc < 4; ++jc) acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], w4[jc], acc[jc], 0, 0, 0);
        }
    }
    __syncthreads();                                            // s_r: the softmax is done with the logits' partials
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) s_r[(wave * 4 + jc) * 64 + lane] = acc[jc];
    __syncthreads();
    if (ROLE == 0) {
        const int jc = wave;
        f32x4 v = s_r[(0 * 4 + jc) * 64 + lane];
#pragma unroll
        for (int q = 1; q < T::NW; ++q) { const f32x4 u = s_r[(q * 4 + jc) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        if (ecol < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                if (orow < n) {
                    float x = v[r] + 0.f;                          // (no bias)
                    x = e_mask[r] ? 2.f * x : 0.f;
                    x = e_gate[r] > 0.f ? x : 0.f;
                    P.C[(size_t)orow * P.ldc + ecol] = x;
                }
            }
        }
    }
    ts.hit();
}

// The three d1 data gradients of a member (towers t = 0..2: dH5_t[16 x 64] = dfeat[:, 180 t ..] wTd_t, K = 180) in ONE pass.
// As tiles of their own each is 12 groups of 16 k over 8 waves - six waves with two groups, two idle - and the three tiles cost
// three operand round trips (17 us of the chain's 65).  Here the 18 (tower, wave-partial) jobs - partial p of a tile = the MFMA
// chain over groups 2p, 2p + 1, exactly what wave p computes in gemm_splitk4_kernel - are dealt to the 8 waves (3, 3, 2, ...),
// all their operands are requested together, and the partials are added per tile in the same order (p = 0..5, then the two
// zero partials of the idle waves: x + 0 is not x for x = -0), so the bits are those of the layer launches.
// s_p: 18 * 4 * 64 f32x4 of LDS.
constexpr int kChainS7Jobs = 18, kChainS7Floats = kChainS7Jobs * 4 * 64 * 4;
__device__ __forceinline__ void chain_d1_dgrad(const DenseChainArgs& a, int M, int m0, int n0, f32x4* s_p, ChainStamper& ts) {
    ts.hit();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int row = m0 + lr;
    const bool row_ok = row < M;
    constexpr int JW = 3;                                       // jobs per wave (the third only in waves 0, 1)
    float4 av[JW][2], wv[JW][2][4];
    unsigned char e_mask[2][4];
    float e_z[2][4];                                            // part5: the raw conv5 values of the outputs this wave finishes ...
    float4 e_lo[2];                                             // ... and their channel's (sc, sh, mean, inv_std)
    // the two (tower, column set) pairs this wave finishes: e = wave, wave + 8 of the 12
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int e = wave + 8 * f, t = min(e >> 2, kTowers - 1), jc = e & 3;
        const GemmProblem& P = t == 0 ? a.dh5[0] : (t == 1 ? a.dh5[1] : a.dh5[2]);
        const int ecl = min(n0 + 4 * lr + jc, kFlat - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) e_mask[f][r] = P.mask[(size_t)min(m0 + kq * 4 + r, M - 1) * P.ldm + ecl];
        e_lo[f] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) e_z[f][r] = 0.f;
        if (a.part5) {
            const float* z5 = t == 0 ? a.Z5[0] : (t == 1 ? a.Z5[1] : a.Z5[2]);
#pragma unroll
            for (int r = 0; r < 4; ++r) e_z[f][r] = z5[(size_t)min(m0 + kq * 4 + r, M - 1) * kFlat + ecl];
            // the tower's coefficients as its member of THIS cluster left them in front of barrier 1 (same XCD: L2)
            e_lo[f] = reinterpret_cast<const float4*>(a.clow + ((size_t)(m0 / 16) * kTowers + t) * 240)[ecl / 9];
        }
    }
#pragma unroll
    for (int q = 0; q < JW; ++q) {
        const int job = min(wave + 8 * q, kChainS7Jobs - 1), t = job / 6, p = job - t * 6;
        const GemmProblem& P = t == 0 ? a.dh5[0] : (t == 1 ? a.dh5[1] : a.dh5[2]);
        const float* arow = P.A + (size_t)(row_ok ? row : 0) * P.lda;
        const float* __restrict__ wcol = P.W + min(n0 + 4 * lr, P.NS - 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 16 * (2 * p + u) + 4 * kq;
            av[q][u] = *reinterpret_cast<const float4*>(arow + (k < kD1 ? k : 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[q][u][i] = *reinterpret_cast<const float4*>(wcol + (size_t)min(k + i, kD1 - 1) * P.NS);
        }
    }
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
    const bool col_ok = n0 + 4 * lr < a.dh5[0].NS;
    __syncthreads();                                            // s_p may still be read by the previous tile's epilogue
#pragma unroll
    for (int q = 0; q < JW; ++q) {
        const int job = wave + 8 * q;
        if (job >= kChainS7Jobs) continue;
        const int p = job % 6;
        f32x4 acc[4];
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) acc[jc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 16 * (2 * p + u) + 4 * kq;
            const float4 av4 = (row_ok && k < kD1) ? av[q][u] : zero4;
            const float a4[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w = (col_ok && k + i < kD1) ? wv[q][u][i] : zero4;
                const float w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int jc = 0; jc < 4; ++jc) acc[jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], w4[jc], acc[jc], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jc = 0; jc < 4; ++jc) s_p[(job * 4 + jc) * 64 + lane] = acc[jc];
    }
    __syncthreads();
    ts.hit();
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int e = wave + 8 * f;
        if (e >= 4 * kTowers) continue;
        const int t = e >> 2, jc = e & 3;
        const GemmProblem& P = t == 0 ? a.dh5[0] : (t == 1 ? a.dh5[1] : a.dh5[2]);
        f32x4 v = s_p[((t * 6 + 0) * 4 + jc) * 64 + lane];
#pragma unroll
        for (int p = 1; p < 6; ++p) { const f32x4 u = s_p[((t * 6 + p) * 4 + jc) * 64 + lane]; v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
#pragma unroll
        for (int z = 0; z < 2; ++z) { v[0] += 0.f; v[1] += 0.f; v[2] += 0.f; v[3] += 0.f; }    // the idle waves' partials of the layer launch
        const int ecol = n0 + 4 * lr + jc;
        if (ecol < kFlat) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                if (orow < M) P.C[(size_t)orow * P.ldc + ecol] = e_mask[f][r] ? 2.f * v[r] : 0.f;
            }
        }
        if (a.part5) {
            // (d beta, d gamma) of the conv5 block, this column's share of the row block: dy = dH5 where the block's ReLU let the
            // value through; the four lane groups hold four rows each (fixed order: rows, then lane groups 0+1, 2+3)
            float pb = 0.f, pg = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + kq * 4 + r;
                const float dh = e_mask[f][r] ? 2.f * v[r] : 0.f;
                const float dy = (orow < M && fmaf(e_z[f][r], e_lo[f].x, e_lo[f].y) > 0.f) ? dh : 0.f;
                pb += dy; pg = fmaf(dy, (e_z[f][r] - e_lo[f].z) * e_lo[f].w, pg);
            }
            pb += __shfl_xor(pb, 16); pg += __shfl_xor(pg, 16);
            pb += __shfl_xor(pb, 32); pg += __shfl_xor(pg, 32);
            if (kq == 0 && ecol < kFlat)
                reinterpret_cast<float2*>(a.part5)[((size_t)t * ((M + 15) / 16) + m0 / 16) * kFlat + ecol] = make_float2(pb, pg);
        }
    }
    ts.hit();
}

// The chain as a wave of role ROLE executes it: the same tiles and the same workgroup barriers in the same order.
template <int ROLE>
__device__ __forceinline__ void chain_body(const DenseChainArgs& a, int j, int m0, unsigned int* ctr, f32x4* s_r, float (*s_d)[17],
                                           volatile int* s_flag, ChainStamper& ts) {
    const int n = a.n, n0 = j * GemmSplitK4::BN;
    bool ok = true;
    // d1 x3: member j = (tower j / 3, column block j % 3); H5 comes from the launch before
    const int t1 = j / 3, n1 = (j - 3 * t1) * GemmSplitK4::BN;
    const GemmProblem P1 = t1 == 0 ? a.d1[0] : (t1 == 1 ? a.d1[1] : a.d1[2]);
    if (a.bn5.units[0]) {
        // the conv5 block's batch norm + ReLU + dropout happen here, on the way into the matrix cores (P1.A = the raw conv5 rows)
        float* lds = reinterpret_cast<float*>(s_r);
        ChainBn5Tile bx;
        bx.units = reinterpret_cast<const float2*>(t1 == 0 ? a.bn5.units[0] : (t1 == 1 ? a.bn5.units[1] : a.bn5.units[2]));
        bx.bn = t1 == 0 ? a.bn5.bn[0] : (t1 == 1 ? a.bn5.bn[1] : a.bn5.bn[2]);
        bx.ss = t1 == 0 ? a.bn5.ss[0] : (t1 == 1 ? a.bn5.ss[1] : a.bn5.ss[2]);
        bx.low4 = t1 == 0 ? a.bn5.low4[0] : (t1 == 1 ? a.bn5.low4[1] : a.bn5.low4[2]);
        bx.mask = t1 == 0 ? a.bn5.mask[0] : (t1 == 1 ? a.bn5.mask[1] : a.bn5.mask[2]);
        bx.H5 = t1 == 0 ? a.bn5.H5[0] : (t1 == 1 ? a.bn5.H5[1] : a.bn5.H5[2]);
        bx.nwg = a.bn5.nwg; bx.n = n;
        bx.store_h5 = n1 == 0;                                  // one of the tower's three members stores the rows
        bx.store_tables = n1 == 0 && m0 == 0;                   // ... and in row block 0 the tables / running averages
        bx.clow = (n1 == 0 && a.clow) ? a.clow + ((size_t)(m0 / 16) * kTowers + t1) * 240 : nullptr;
        bx.s_part = reinterpret_cast<double*>(lds + 8192);      // behind the 8 x 4 partial tiles (32 KB): 12 KB
        bx.s_sc = reinterpret_cast<float2*>(lds + 8192 + 3072);
        bx.s_bn = reinterpret_cast<float2*>(lds + 8192 + 3072 + 128);
        chain_tile<ROLE, true>(P1, n, kFlat, kFlat, kD1, 1, m0, n1, s_r, ts, &bx);
    } else {
        chain_tile<ROLE>(P1, n, kFlat, kFlat, kD1, 1, m0, n1, s_r, ts);
    }
    ok = chain_barrier<ROLE>(ctr, 1, s_flag, ts);
    ts.seek(8);
    if (ok) {                                                   // FC1 (K = 540 features of the three towers)
        chain_tile<ROLE>(a.fc1, n, kFC1, kFC1, kFC1, 1, m0, n0, s_r, ts);
        ok = chain_barrier<ROLE>(ctr, 2, s_flag, ts);
    }
    ts.seek(16);
    if (ok) {                                                   // fc_2 on [FC1 | atlas]: 5 column blocks
        if (n0 < kFC2) chain_tile<ROLE>(a.fc2, n, kFC1ld, kFC1 + kAtlas, kFC2, 1, m0, n0, s_r, ts);
        ok = chain_barrier<ROLE>(ctr, 3, s_flag, ts);
    }
    ts.seek(24);
    if (ok) {                                                   // out + softmax + loss + fc_2's data gradient, one stage
        chain_head_dcat<ROLE>(a, j, m0, n0, s_r, s_d, reinterpret_cast<float*>(s_r) + GemmSplitK4::NW * 4 * 64 * 4, ts);
        ok = chain_barrier<ROLE>(ctr, 4, s_flag, ts);
    }
    ts.seek(40);
    if (ok) {
        chain_tile<ROLE>(a.dfeat, n, kFC1, kFC1, kFlat, 0, m0, n0, s_r, ts);
        ok = chain_barrier<ROLE>(ctr, 5, s_flag, ts);
    }
    ts.seek(48);
    if (ok) chain_d1_dgrad(a, n, m0, n0, s_r, ts);              // d1 x3 data gradients: the member's three tiles in one pass
    if (!ok && threadIdx.x == 0) a.loss_rows[m0] = __builtin_nanf("");   // a barrier gave up: the step's loss says so
}

__global__ __launch_bounds__(512, 1) void dense_chain_kernel(DenseChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_chain[];   // kChainLdsBytes
    f32x4* s_r = reinterpret_cast<f32x4*>(smem_chain);                  // 8 * 4 * 64 partial tiles; 18 * 4 * 64 in the last stage
    __shared__ float s_d[16][17];
    __shared__ int s_flag;
    const int n = a.n, nblocks = (n + 15) / 16;
    const unsigned long long t_start = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // membership: a ticket from the counter of the XCD this workgroup runs on
    const int xcd = __builtin_amdgcn_readfirstlane(xcc_id() & (kXcds - 1));
    unsigned int* const xbase = a.counters + (size_t)xcd * (1 + kChainClustersPerXcd) * kChainLine;
    if (threadIdx.x == 0) s_flag = (int)l2_add(xbase, 1u);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_flag);   // wave-uniform: member, row block and column block live in SGPRs
    __syncthreads();                                            // s_flag is reused by the barriers
    const int cluster = ticket / kChainMembers, j = ticket - cluster * kChainMembers;
    const int block = cluster * kXcds + xcd;
    if (cluster >= kChainClustersPerXcd) {
        // more workgroups on this XCD than a launch can place there: another XCD is short of them and its row blocks will never
        // be computed (their row losses keep the NaN the prep launch wrote; this one says so too).  The host only uses this
        // kernel on devices where a probe launch saw 8 equal shares (scs_train.hip: chain_placement_probe).
        if (threadIdx.x == 0) a.loss_rows[0] = __builtin_nanf("");
        return;
    }
    if (block >= nblocks) return;                               // the grid is rounded up to whole rounds of 8 row blocks: a whole cluster leaves together
    const int m0 = block * 16;
    unsigned int* ctr = xbase + (size_t)(1 + cluster) * kChainLine;
    ChainStamper ts{a.stamps ? a.stamps + (size_t)(block * kChainMembers + j) * kChainStamps : nullptr, 0};
    if (ts.p && threadIdx.x == 0) { ts.p[62] = t_start; ts.p[63] = __builtin_amdgcn_s_memrealtime(); }
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) chain_body<0>(a, j, m0, ctr, s_r, s_d, &s_flag, ts);
    else chain_body<1>(a, j, m0, ctr, s_r, s_d, &s_flag, ts);
}

constexpr size_t kChainLdsBytes = (size_t)kChainS7Floats * 4;
static_assert(kChainS7Floats >= 8192 + 3072 + 128 + 2 * 544, "the BN5 tables of the first tile sit behind the partial tiles");
static_assert(kChainS7Floats >= GemmSplitK4::NW * 4 * 64 * 4 + 16 * kFC2ld, "the last stage's partials are the larger set; dR2 [16][272] sits behind the 8 x 4 partial tiles");
inline hipError_t dense_chain_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsBytes);
}
inline dim3 dense_chain_grid(int n) { return dim3(kXcds * kChainMembers * (((n + 15) / 16 + kXcds - 1) / kXcds)); }

inline bool dense_chain_ok(int n, const GemmArgs& probe, int nproblems) { return n > 0 && n <= kChainMaxBatch && gemm_vec4_ok(probe, nproblems); }

}  // namespace scs
