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
#include <type_traits>

#include "scs_common.h"

namespace scs {

constexpr int kW1Floats = 12 * 16 + 12 * 4 + 20;   // conv1 operands: [12][16] + [12][4] + bias[20]
static_assert(kW1Floats > 256 && kW1Floats <= 512, "conv1 operand staging assumes two loads per thread");

// Epilogues.
//   EPI_RAW_STATS      raw outputs plus, per (patch, band, channel) unit of OUTB values, their mean and sum of squared deviations
//                      (batch-norm statistics of the training forward pass without a second pass over the activations;
//                      accumulated about the unit's first value and combined by the parallel-variance formula, so nothing
//                      of the form E[v^2] - E[v]^2 is ever evaluated)
//   EPI_RAW_BNB[_POOL] data gradient of a training conv: raw outputs dX plus, per (patch, band, channel), the partial
//                      sums (d beta, d gamma) of the batch-norm layer BELOW (whose block output X this kernel differentiates):
//                      dy = dX routed through that block's (max-pool and) ReLU, from its raw conv output `zlow`
//   EPI_L1             data gradient of conv2 in training: dX1 is never stored.  The block below is conv1 -> BN -> ReLU,
//                      whose raw output is recomputed from the 32x32 input rows; the kernel leaves, per (patch, band,
//                      channel c), A[c][tap] = sum_px dy[c][px] * x0[px + tap] and d beta[c] = sum_px dy[c][px], from which
//                      d W1, d gamma1 follow algebraically (scs_train.hip: l1_final_kernel)
enum ConvEpi { EPI_BIAS_RELU = 0, EPI_BIAS_RELU_POOL = 1, EPI_RAW = 2, EPI_RAW_STATS = 3, EPI_RAW_BNB = 4, EPI_RAW_BNB_POOL = 5, EPI_L1 = 6 };
// Sources.
//   SRC_NCHW           [n][CIN][HIN][HIN] activations
//   SRC_CONV1 / SRC_VOLUME / SRC_X0   the 32x32 input patch (given / cut from one volume at inference / given or cut from
//                      a set of volumes in training); conv1 (+ folded BN + ReLU) runs first and its output stays in LDS
//   SRC_NCHW_PAD2      [n][CIN][HIN-4][HIN-4], zero padded by 2 on every side in LDS (data gradients)
//   SRC_BN[_POOL]      training forward: the previous layer's RAW conv output; y = relu(sc * z + sh) (then 2x2 max-pool of a
//                      [2 HIN][2 HIN] plane) is applied while staging, and the block output X is stored as a by-product
//   SRC_DZ[_POOL]_PAD2 training backward: d z = [y > 0 (and arg max of its window)] * sc * dUp + ca * z + cb is computed
//                      while staging from this layer's raw output z and the upstream gradient dUp (pooled grid for _POOL);
//                      zero padded by 2; d z is stored as a by-product (the weight-gradient kernel reads it)
enum ConvSrc { SRC_NCHW = 0, SRC_CONV1 = 1, SRC_VOLUME = 2, SRC_NCHW_PAD2 = 3, SRC_BN = 4, SRC_BN_POOL = 5, SRC_DZ_PAD2 = 6,
               SRC_DZ_POOL_PAD2 = 7, SRC_X0 = 8 };

// FIN_ = 1 (EPI_RAW_STATS / EPI_RAW_BNB*): the workgroup of a tower that finishes last also finalizes the batch-norm sums
// (what a separate tiny launch would do: in the stream that launch costs ~9 us - two-level tree of tickets and partial
// reductions, see take_ticket / bn_grp_* / bn_top_*).
// NSPLIT_ = KS > 0 (the 3x3 / 5x5 planes of the conv5 block: a handful of M tiles per workgroup, whose matrix work is tiny next
// to the latency of staging and weight streaming): GEMM-style K split.  The NT16 * KS waves of a workgroup are (channel pair
// p, K part ks): wave (p, ks) owns the 16-wide block p and the 4-wide block p of ALL M tiles of the workgroup's G patches, over
// the input-channel chunks [ks * NC4 / KS, (ks + 1) * NC4 / KS).  Its B operands come straight from global memory (every
// workgroup reads the same packed weights: L2), ALL of them requested at kernel start - one round trip that hides behind the
// staging - and each is used for every tile, so a weight element is fetched once per G patches instead of once per patch.
// No weight buffer in LDS, no barrier inside the loop; the KS partial sums are added through LDS in a fixed order.
// Derived: SHARED (4q + 1 tiles on 4 waves: the odd tile's channel blocks are dealt to the waves); tile ownership rotates
// with the workgroup index (see the kernel).
template <int CIN_, int COUT_, int HIN_, int G_, int EPI_, int SRC_, int PLANE_, int NB_ = 1, int WPS_ = 1, int STAMP_ = 0, int NW_ = 4, int FIN_ = 0,
          int NSPLIT_ = 0>
struct ConvCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, G = G_, EPI = EPI_, SRC = SRC_, NB = NB_, WPS = WPS_, STAMP = STAMP_, FIN = FIN_;
    static_assert(!FIN_ || (NW_ >= 4 && (EPI_ == EPI_RAW_STATS || EPI_ == EPI_RAW_BNB || EPI_ == EPI_RAW_BNB_POOL)), "in-kernel finalize: statistics kernels of at least 256 threads");
    static_assert(!FIN_ || G_ == 1, "in-kernel finalize: one statistics unit per workgroup (a patch, or one band of it)");
    static constexpr int NW = NW_, NTH = 64 * NW_;          // waves / threads per workgroup
    static constexpr bool NSPLIT = NSPLIT_ != 0;            // K split over waves of (channel pair, K part) (see above)
    static constexpr int KS = NSPLIT_ > 0 ? NSPLIT_ : 1;
    static constexpr bool PAD2 = (SRC_ == SRC_NCHW_PAD2 || SRC_ == SRC_DZ_PAD2 || SRC_ == SRC_DZ_POOL_PAD2);
    static constexpr bool FUSED1 = (SRC_ == SRC_CONV1 || SRC_ == SRC_VOLUME || SRC_ == SRC_X0);
    // PAD2: the global input is (HIN_-4)^2 and is zero-padded by 2 on every side in LDS (dgrad)
    static constexpr int HIN = HIN_, WIN = HIN_;
    static constexpr int HSRC = PAD2 ? HIN_ - 4 : HIN_;
    static constexpr int HO = HIN_ - 2;
    static constexpr bool POOL = (EPI_ == EPI_BIAS_RELU_POOL);
    static constexpr bool RAW = (EPI_ >= EPI_RAW);
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
    // TILES = NW * q + 1 (9 or 13 tiles on 4 waves): instead of one wave owning the odd tile - and every other wave waiting
    // for it - its channel blocks are dealt to the waves: 16-block b to wave b, the 4-blocks to wave NT16 (training kernels)
#ifndef SCS_SHARED_INFER
#define SCS_SHARED_INFER 0
#endif
    static constexpr bool SHARED = ((EPI_ >= EPI_RAW && !POOL) || SCS_SHARED_INFER) && !NSPLIT_ && TILES > NW && TILES % NW == 1 &&
                                   (nt16(COUT_) + (nt4(COUT_) > 0 ? 1 : 0)) <= NW;
    static constexpr int WCHUNK = conv_chunk_floats(COUT_);   // floats per weight chunk
    static constexpr int WCHUNK4 = WCHUNK / 4;
    static constexpr int WPT = (WCHUNK4 + NTH - 1) / NTH;   // float4 per thread per chunk
    static constexpr int IN_FLOATS = G_ * CIN_ * PLANE_;
    static constexpr int IN_FLOATS_AL = (IN_FLOATS + 3) / 4 * 4;
    static constexpr int XS_ROWS_FLOATS = (XROWS * 32 + 3) / 4 * 4;
    static constexpr int XS_FLOATS = FUSED1 ? XS_ROWS_FLOATS + 4 + kW1Floats : 0;
    // the epilogue restages the outputs in LDS ([g][cout][OSTR], odd stride) and stores them as whole rows
    static constexpr bool OUT_VEC4 = (OUTB % 4 == 0) && (OUTPIX % 4 == 0);
    static constexpr int OSTR = OUT_VEC4 ? OUTB + 4 : (OUTB | 1);
    static constexpr int OUT_FLOATS = G_ * COUT_ * OSTR;
    // EPI_L1: the input rows of the band ((HOB + 2) x 32) and the conv1 tap table sit behind the main buffers (they are
    // fetched at kernel start); the waves' partial sums reuse the space behind the restaged outputs
    static constexpr int L1_X_FLOATS = (HOB + 2) * 32 + 8, L1_TAB_FLOATS = 240, L1_RED_FLOATS = NW_ * 20 * 16;
    static constexpr int L1_PRE_FLOATS = EPI_ == EPI_L1 ? L1_X_FLOATS + L1_TAB_FLOATS : 0;
    // NSPLIT: the waves' partial tiles [KS][TILES * 16 rows][COUT] sit behind the restaged outputs
    static constexpr int OUT_FLOATS_AL = (OUT_FLOATS + 3) / 4 * 4;
    static constexpr int KRED_FLOATS = NSPLIT_ ? KS * TILES * 16 * COUT_ : 0;
    static constexpr int EPI_FLOATS = (NSPLIT_ ? OUT_FLOATS_AL + KRED_FLOATS : OUT_FLOATS) + (EPI_ == EPI_L1 ? L1_RED_FLOATS : 0);
    static constexpr int WBUF_FLOATS = NSPLIT_ ? 0 : 2 * WCHUNK;   // double-buffered weight chunks (NSPLIT: operands come straight from global memory)
    static constexpr int MAIN_FLOATS = IN_FLOATS_AL + WBUF_FLOATS + XS_FLOATS;
    static_assert(EPI_ != EPI_L1 || EPI_FLOATS <= MAIN_FLOATS, "EPI_L1 scratch must not reach the prefetched input rows");
    static constexpr size_t LDS_BYTES = (size_t)((MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS) + L1_PRE_FLOATS) * 4;
    static_assert(CIN_ % 4 == 0, "k-steps are groups of 4 input channels");
    static_assert(COUT_ % 4 == 0, "output channels are covered by 16-wide and 4-wide MFMA blocks");
    static_assert(HO % NB_ == 0 && (!POOL || (HOB % 2 == 0)), "bands must split the output rows (evenly for pooling)");
    static_assert(NB_ == 1 || G_ == 1, "bands: one patch per workgroup");
    static_assert(PLANE_ >= HINB * HIN_, "plane stride too small");
    static_assert(LDS_BYTES * WPS_ <= 160 * 1024, "LDS budget (WPS workgroups per CU)");
    static_assert(NTH >= 8 * XROWS || !FUSED1, "raw patch rows are staged in one pass");
    static_assert(SRC_ != SRC_BN || ((HIN_ * HIN_) % 4 == 0 && PLANE_ % 4 == 0 && (HOB * HIN_) % 4 == 0 && (HINB * HIN_) % 4 == 0),
                  "SRC_BN stages 16-byte pieces (of a band: its first row and its row count must keep them aligned)");
    static_assert(SRC_ != SRC_DZ_POOL_PAD2 || (HINB % 2 == 0 && HIN_ % 2 == 0 && PLANE_ % 2 == 0 && (NB_ == 1 || HOB % 2 == 0)),
                  "pooled d z staging walks 2x2 windows");
    static_assert(EPI_ != EPI_L1 || (COUT_ == 20 && HO == 30 && G_ == 1 && NW_ == 4), "EPI_L1 is the conv2 data gradient");
    static_assert(!NSPLIT_ || (NW_ == NT16 * KS && RAW && !FUSED1 && EPI_ != EPI_L1 && NT16 == NT4 && NT16 >= 1 && NB_ == 1 &&
                               NC4 % KS == 0 && TILES <= 8),
                  "NSPLIT: raw training kernels, Cout = pairs of a 16-wide and a 4-wide block, NT16 * KS waves, KS divides the chunks");
    static_assert(!FIN_ || LDS_BYTES >= (size_t)NTH * 3 * 8 + 16, "finalize scratch");
};

// the 32x32 input patches of a training batch: given as three [n][32][32] arrays, or cut on the fly from z-scored
// volumes resident on the device (tri-planar gather of base.py:293-306 around each sample's centre voxel)
struct X0Src {
    const float* views[kTowers];  // null: gather
    const float* vols;            // all volumes, concatenated
    const scs_volume* descs;      // per volume: offset into vols, dims
    const int32_t* samples;       // [N][4]: volume, x, y, z
    const int32_t* index;         // [n] rows of `samples` that form this batch (null: rows 0..n-1)
    int n_samples, n_vols;        // bounds: a row / volume number outside them is clamped, never dereferenced
    const uint8_t* flips;         // [n] or null: per batch row, two bits per view (bit 2t: mirror the patch rows, bit 2t+1: mirror
                                  // its columns) - the rotate / flip augmentation of nets.py:41-124 applied inside the gather
};

// sample row of patch g: (volume, x, y, z), row and volume numbers clamped into the tables (voxels outside a volume read as
// zero padding in gather_voxel; nothing a caller passes can make the kernels leave their buffers)
__device__ __forceinline__ int4 x0_sample(const X0Src& s, int g) {
    int r = s.index ? s.index[g] : g;
    r = max(0, min(r, s.n_samples - 1));
    int4 smp = reinterpret_cast<const int4*>(s.samples)[r];
    smp.x = max(0, min(smp.x, s.n_vols - 1));
    return smp;
}

// what the last workgroup of a tower needs to turn the partial sums into batch-norm coefficients (FIN configurations)
struct FinArgs {
    unsigned int* counter;        // arrival tickets, zero between launches: fin_counter_words(workgroups per tower) words per tower
    double* grp;                  // group records: [tower][fin_groups(workgroups per tower)][kFinRec][2]
    double count;                 // values per channel over the batch
    float* bn[kTowers];           // forward: this layer's (beta, gamma, mean, inv_std) in the parameter arena; backward: the
                                  // layer below's (d beta, d gamma, -, -) slot in the gradient arena
    float* ss[kTowers];           // forward: float2 [COUT] (scale, shift) out
    float* coef[kTowers];         // forward: float4 [COUT] (sc, sh, mean, inv_std) out; backward: float4 [COUT] (sc, sh, ca, cb) out
};

// The in-kernel finalize is a two-level tree over the workgroups of a tower, in arrival tickets AND in arithmetic: groups of
// kFinGroup workgroups share a ticket counter (each on its own 128-byte line); the workgroup that completes a group reduces
// that group's partial sums to one record per channel, then takes a ticket on the tower's top counter; the workgroup that
// completes the top counter combines the group records.  Same-address device-scope atomics retire one at a time (~50 ns
// each on MI355X, tools/gridbar.hip): a flat counter makes the last of 256 simultaneously finishing workgroups wait ~13 us
// for its ticket, the tree ~1.5 us; and the last workgroup reads units / kFinGroup records instead of every unit, so the tail
// of the kernel does not grow with the batch.  Which workgroup arrives last changes nothing in the arithmetic: groups are
// fixed by the workgroup index and every sum runs in index order (deterministic).
constexpr int kFinGroup = 16, kFinLine = 32;
constexpr size_t fin_counter_words(size_t wgs_per_tower) { return (1 + (wgs_per_tower + kFinGroup - 1) / kFinGroup) * kFinLine; }
constexpr size_t fin_groups(size_t wgs_per_tower) { return (wgs_per_tower + kFinGroup - 1) / kFinGroup; }
constexpr int kFinRec = 64;                                   // group records per group: [64 channels] x (double, double)

// Cross-workgroup hand-off of the partial sums without any fence: the sums are stored write-through (`sc1`: they never sit
// dirty in an XCD's L2), every storing wave drains its stores (vmcnt(0)) in front of the workgroup barrier behind which ONE
// lane takes the ticket, and every load of handed-off data is an `sc1` load to registers (served by L2 / memory, never by a
// CU's L1) issued after the ticket that completed the count has returned (MI355X_MICROARCH.md, "Valid forms").
// This hand-off is valid only where vector stores are counted by vmcnt and `sc1` stores write through - the gfx942 / gfx950
// family.  On a target that counts stores in vscnt the ticket could overtake the sums.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "conv_mfma.h: the in-kernel batch-norm finalize (FIN configurations) relies on gfx942/gfx950 memory-counter semantics"
#endif
__device__ __forceinline__ void st_sc1(float2* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_sc1(const float2* p) {
    return __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// One ticket on `c` behind this workgroup's drained stores; true in every thread when it completed `full` arrivals (the
// counter is re-armed for the next launch).
__device__ __forceinline__ bool take_ticket(unsigned int* c, unsigned int full, volatile int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == full - 1;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}

constexpr int kFinBatch = 8;

// ---- forward statistics.  A unit is (mean, M2 = sum of squared deviations) of `per_unit` values.
// group step: units [u0, u0 + k) of channel c -> (mean_g, M2_g) by the parallel-variance formula in double, about the first
// unit's mean (exact differences)
template <int CH>
__device__ __forceinline__ void bn_grp_forward(const float2* __restrict__ part /*[units][CH]*/, int u0, int k, double per_unit,
                                               double* __restrict__ rec /*[kFinRec][2]*/) {
    const int c = threadIdx.x;
    if (c < CH) {
        float2 v[kFinGroup];
#pragma unroll
        for (int i = 0; i < kFinGroup; ++i) v[i] = ld_sc1(part + (size_t)(u0 + min(i, k - 1)) * CH + c);
        const double m0 = (double)v[0].x;
        double s = 0, s2 = 0, q = 0;
#pragma unroll
        for (int i = 0; i < kFinGroup; ++i)
            if (i < k) { const double d = (double)v[i].x - m0; s += d; s2 += d * d; q += (double)v[i].y; }
        const double dm = s / k;
        st_sc1(rec + 2 * c, m0 + dm);
        st_sc1(rec + 2 * c + 1, q + per_unit * fmax(s2 - s * dm, 0.0));
    }
}
// top step: group records (group g holds min(kFinGroup, units - g kFinGroup) units) -> batch mean / biased variance -> scale /
// shift for the next conv's staging, (sc, sh, mean, inv_std) for the backward pass, running averages (alpha 0.1)
template <int CH, int NTH>
__device__ __forceinline__ void bn_top_forward(const double* __restrict__ rec /*[groups][kFinRec][2]*/, int groups, int units, double per_unit,
                                               float* bn, float2* __restrict__ ss, float4* __restrict__ low4, double* s_acc /*[NTH][3]*/) {
    constexpr int J = NTH / CH;
    const int tid = threadIdx.x, j = tid / CH, c = tid - j * CH;
    double s = 0, s2 = 0, q = 0, m0 = 0;
    if (j < J) {
        m0 = ld_sc1(rec + 2 * c);
        for (int g0 = j; g0 < groups; g0 += kFinBatch * J) {
            double mg[kFinBatch], qg[kFinBatch];
#pragma unroll
            for (int i = 0; i < kFinBatch; ++i) {
                const double* r = rec + ((size_t)min(g0 + i * J, groups - 1) * kFinRec + c) * 2;
                mg[i] = ld_sc1(r); qg[i] = ld_sc1(r + 1);
            }
#pragma unroll
            for (int i = 0; i < kFinBatch; ++i) {
                const int g = g0 + i * J;
                if (g < groups) {
                    const double ng = per_unit * min(kFinGroup, units - g * kFinGroup), d = mg[i] - m0;
                    s += ng * d; s2 += ng * d * d; q += qg[i];
                }
            }
        }
    }
    s_acc[tid * 3] = s; s_acc[tid * 3 + 1] = s2; s_acc[tid * 3 + 2] = q;
    __syncthreads();
    if (tid < CH) {
        s = s2 = q = 0;
        for (int i = 0; i < J; ++i) { s += s_acc[(tid + i * CH) * 3]; s2 += s_acc[(tid + i * CH) * 3 + 1]; q += s_acc[(tid + i * CH) * 3 + 2]; }
        const double count = per_unit * units, dm = s / count, mean = m0 + dm;
        const double var = (q + fmax(s2 - s * dm, 0.0)) / count;
        const double inv = 1.0 / sqrt(var + (double)kBnEps);
        const float sc = bn[CH + c] * (float)inv, sh = bn[c] - (float)mean * sc;
        ss[c] = make_float2(sc, sh);
        low4[c] = make_float4(sc, sh, (float)mean, (float)inv);
        bn[2 * CH + c] = (1.f - kBnAlpha) * bn[2 * CH + c] + kBnAlpha * (float)mean;
        bn[3 * CH + c] = (1.f - kBnAlpha) * bn[3 * CH + c] + kBnAlpha * (float)inv;
    }
}

// ---- backward: units are (d beta, d gamma) partial sums
template <int CH>
__device__ __forceinline__ void bn_grp_backward(const float2* __restrict__ part, int u0, int k, double* __restrict__ rec) {
    const int c = threadIdx.x;
    if (c < CH) {
        float2 v[kFinGroup];
#pragma unroll
        for (int i = 0; i < kFinGroup; ++i) v[i] = ld_sc1(part + (size_t)(u0 + min(i, k - 1)) * CH + c);
        double sb = 0, sg = 0;
#pragma unroll
        for (int i = 0; i < kFinGroup; ++i)
            if (i < k) { sb += (double)v[i].x; sg += (double)v[i].y; }
        st_sc1(rec + 2 * c, sb);
        st_sc1(rec + 2 * c + 1, sg);
    }
}
// (d beta, d gamma) -> gradients + the coefficients of d z = [y > 0] sc dUp + ca z + cb,
// ca = -sc inv mg,  cb = sc (inv mean mg - mb),  mb = d beta / M,  mg = d gamma / M.
template <int CH, int NTH>
__device__ __forceinline__ void bn_top_backward(const double* __restrict__ rec, int groups, double count, const float4* __restrict__ low4,
                                                float* __restrict__ gbn, float4* __restrict__ cf4, double* s_acc /*[NTH][3]*/) {
    constexpr int J = NTH / CH;
    const int tid = threadIdx.x, j = tid / CH, c = tid - j * CH;
    double sb = 0, sg = 0;
    if (j < J) {
        for (int g0 = j; g0 < groups; g0 += kFinBatch * J) {
            double bg[kFinBatch], gg[kFinBatch];
#pragma unroll
            for (int i = 0; i < kFinBatch; ++i) {
                const double* r = rec + ((size_t)min(g0 + i * J, groups - 1) * kFinRec + c) * 2;
                bg[i] = ld_sc1(r); gg[i] = ld_sc1(r + 1);
            }
#pragma unroll
            for (int i = 0; i < kFinBatch; ++i)
                if (g0 + i * J < groups) { sb += bg[i]; sg += gg[i]; }
        }
    }
    s_acc[tid * 3] = sb; s_acc[tid * 3 + 1] = sg;
    __syncthreads();
    if (tid < CH) {
        sb = sg = 0;
        for (int i = 0; i < J; ++i) { sb += s_acc[(tid + i * CH) * 3]; sg += s_acc[(tid + i * CH) * 3 + 1]; }
        gbn[c] = (float)sb; gbn[CH + c] = (float)sg;                                // mean / inv_std entries stay 0
        const float4 lo = low4[c];                                                  // sc, sh, mean, inv
        const double mb = sb / count, mg = sg / count, sc = lo.x, inv = lo.w, mean = lo.z;
        cf4[c] = make_float4(lo.x, lo.y, (float)(-sc * inv * mg), (float)(sc * (inv * mean * mg - mb)));
    }
}

struct ConvArgs {
    const float* in[kTowers];     // SRC_NCHW*: [n][CIN][HSRC][HSRC]; SRC_CONV1: [n][32][32]; SRC_VOLUME: the volume;
                                  // SRC_BN*: raw conv output of the layer below; SRC_DZ*: raw conv output of THIS layer
    const float* w[kTowers];      // packed chunks per 4 input channels: [9][4][NS16] + [9][4][L4] (scs_common.h)
    const float* bias[kTowers];   // [NT*16] (unused for EPI_RAW*)
    float* out[kTowers];          // [n][COUT][OUTPIX]
    const float* w1[kTowers];     // fused conv1: operands [12][16] + [12][4] + bias[20] (kW1Floats)
    const int32_t* centers;       // SRC_VOLUME: [n][3]
    int dimx, dimy, dimz;
    int n;                        // patches
    float* stats;                 // EPI_RAW_STATS: [tower][n * NB][COUT][2] (mean, M2) per patch band; EPI_RAW_BNB*: same shape, partial sums;
                                  // EPI_L1: [tower][n * NB][20][10]
    unsigned long long* stamps;   // diagnostic builds only (STAMP): [workgroup][8] s_memtime values
    // ---- training only
    const float* bn[kTowers];     // SRC_BN*: float2 [CIN] (scale, shift); SRC_DZ*: float4 [CIN] (sc, sh, ca, cb)
    float* xout[kTowers];         // SRC_BN*: block output X [n][CIN][HIN][HIN]; SRC_DZ*: d z [n][CIN][HSRC][HSRC] (null: not stored)
    const float* dup[kTowers];    // SRC_DZ*: upstream gradient
    const float* zlow[kTowers];   // EPI_RAW_BNB*: raw conv output of the layer below [n][COUT][HO or 2 HO]^2
    const float* bnlow[kTowers];  // EPI_RAW_BNB*: float4 [COUT] (sc, sh, mean, inv_std) of the layer below; EPI_L1: [20][12] flipped
                                  // conv1 filters (9), sc, sh, pad
    X0Src x0;                     // SRC_X0 / EPI_L1: the input patches
    int tower0;                   // first tower of this launch (grid.y towers from here; per-tower streams launch one each)
    FinArgs fin;                  // FIN configurations
    float* pout[kTowers];         // fused conv1 + conv2 in training (EPI_RAW_STATS), or null: [n][COUT][HP][HP], per 2x2 window of the raw output the
                                  // value the block's batch norm + ReLU + max-pool will keep - the window's maximum where the layer's gamma
                                  // (bias[]: [COUT]) is >= 0, its minimum otherwise (z -> sc z + sh and the ReLU are monotone, sign(sc) = sign(gamma):
                                  // pool(relu(bn(z))) = relu(bn(that value)) exactly).  The next conv stages this quarter-size array (SRC_BN)
                                  // instead of pooling the full-size one (SRC_BN_POOL); the backward pass still reads the full-size output.
    float* wgstats;               // EPI_RAW_STATS with G > 1 (conv5 block), or null: [tower][workgroup][COUT] (mean, M2) of the workgroup's
                                  // G * OUTB values per channel - the units the one-launch dense head finalizes (dense_chain.h)
    const unsigned int* guard;    // GUARDED instantiations only (see the kernel): the launch does nothing while this device word is zero
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
// K_ = 0: as few round trips as possible with at most 12 loads in flight per thread (equal batches)
template <int TOTAL, class V, int NTH = 256, int K_ = 8, class SrcF, class DstF>
__device__ __forceinline__ void staged_copy(int tid, SrcF&& src_of, DstF&& dst_of) {
    constexpr int ITER = (TOTAL + NTH - 1) / NTH;
    constexpr int TRIPS = (ITER + 11) / 12;
    constexpr int K = K_ > 0 ? K_ : (ITER + TRIPS - 1) / TRIPS;
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

// one element of the tri-planar gather, zero padded (base.py:293-306): patch pixel (i, j) of `tower` around (cx, cy, cz):
// axial [x-16+i][y-16+j][z], coronal [x-16+i][y][z-16+j], sagittal [x][y-16+i][z-16+j]
__device__ __forceinline__ float gather_voxel(const float* __restrict__ vol, int dx, int dy, int dz, int tower, int cx, int cy, int cz,
                                              int i, int j) {
    int vx, vy, vz;
    if (tower == 0) { vx = cx - 16 + i; vy = cy - 16 + j; vz = cz; }
    else if (tower == 1) { vx = cx - 16 + i; vy = cy; vz = cz - 16 + j; }
    else { vx = cx; vy = cy - 16 + i; vz = cz - 16 + j; }
    const bool ok = (unsigned)vx < (unsigned)dx && (unsigned)vy < (unsigned)dy && (unsigned)vz < (unsigned)dz;
    return ok ? vol[((size_t)vx * dy + vy) * dz + vz] : 0.f;
}

// rows [row0, row0 + NROWS) of the 32x32 input patch g of `tower` -> s_x[NROWS][32]
template <int NROWS, int NTH>
__device__ __forceinline__ void load_x0_rows(const X0Src& s, int tower, int g, int row0, float* __restrict__ s_x, int tid) {
    const float* __restrict__ v = pick3(s.views, tower);
    if (v) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(v + (size_t)g * 1024 + row0 * 32);
#pragma unroll
        for (int e0 = 0; e0 < NROWS * 8; e0 += NTH) {
            const int e = e0 + tid;
            if (e < NROWS * 8) reinterpret_cast<float4*>(s_x)[e] = src[e];
        }
    } else {
        const int4 smp = x0_sample(s, g);
        const scs_volume d = s.descs[smp.x];
        const float* __restrict__ vol = s.vols + d.offset;
        const int fl = s.flips ? (s.flips[g] >> (2 * tower)) & 3 : 0;
        const int fi = (fl & 1) ? 31 : 0, fj = (fl & 2) ? 31 : 0;          // mirrored index = 31 - index = index ^ 31
#pragma unroll
        for (int e0 = 0; e0 < NROWS * 32; e0 += NTH) {
            const int e = e0 + tid;
            if (e < NROWS * 32) s_x[e] = gather_voxel(vol, d.dim_x, d.dim_y, d.dim_z, tower, smp.y, smp.z, smp.w, (row0 + (e >> 5)) ^ fi, (e & 31) ^ fj);
        }
    }
}

// The same rows held in registers between the request and the LDS write (software-pipelined callers: wgrad_mfma.h)
template <int NROWS, int NTH>
struct X0RowRegs {
    static constexpr int N4 = (NROWS * 8 + NTH - 1) / NTH, NS = (NROWS * 32 + NTH - 1) / NTH;
    float4 v4[N4];
    float vs[NS];
    bool dense;
    __device__ __forceinline__ void load(const X0Src& s, int tower, int g, int row0, int tid) {
        const float* __restrict__ v = pick3(s.views, tower);
        dense = v != nullptr;
        if (dense) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(v + (size_t)g * 1024 + row0 * 32);
#pragma unroll
            for (int k = 0; k < N4; ++k) v4[k] = src[min(k * NTH + tid, NROWS * 8 - 1)];
        } else {
            const int4 smp = x0_sample(s, g);
            const scs_volume d = s.descs[smp.x];
            const float* __restrict__ vol = s.vols + d.offset;
            const int fl = s.flips ? (s.flips[g] >> (2 * tower)) & 3 : 0;
            const int fi = (fl & 1) ? 31 : 0, fj = (fl & 2) ? 31 : 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int e = min(k * NTH + tid, NROWS * 32 - 1);
                vs[k] = gather_voxel(vol, d.dim_x, d.dim_y, d.dim_z, tower, smp.y, smp.z, smp.w, (row0 + (e >> 5)) ^ fi, (e & 31) ^ fj);
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ s_x, int tid) const {
        if (dense) {
#pragma unroll
            for (int k = 0; k < N4; ++k) { const int e = k * NTH + tid; if (e < NROWS * 8) reinterpret_cast<float4*>(s_x)[e] = v4[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < NS; ++k) { const int e = k * NTH + tid; if (e < NROWS * 32) s_x[e] = vs[k]; }
        }
    }
};

// TOTAL elements of type V, element e = tid + NTH k, held in registers between load() and store()
template <int TOTAL, class V, int NTH = 256>
struct StagedRegs {
    static constexpr int ITER = (TOTAL + NTH - 1) / NTH;
    V v[ITER];
    template <class SrcF> __device__ __forceinline__ void load(int tid, SrcF&& src_of) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) v[k] = src_of(min(tid + NTH * k, TOTAL - 1));
    }
    template <class DstF> __device__ __forceinline__ void store(int tid, DstF&& dst_of) const {
#pragma unroll
        for (int k = 0; k < ITER; ++k) { const int e = tid + NTH * k; if (e < TOTAL) dst_of(e, v[k]); }
    }
};

// GUARDED (forward_chunk_split in scs_api.hip: the f32 repeat behind the split kernels of a chunk, option fwd_split): every workgroup
// reads the device word a.guard (one scalar load) and leaves at once while it is zero - the launch does the chunk again only if a
// split kernel raised the overflow flag, without a host round trip.  `if constexpr`: the plain instantiations are untouched.
template <class C, bool GUARDED = false>
__global__ __launch_bounds__(C::NTH, (C::WPS * C::NW + 3) / 4) void conv3x3_mfma_kernel(ConvArgs a) {
    if constexpr (GUARDED) { if (*a.guard == 0u) return; }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + C::IN_FLOATS_AL;
    float* s_x = s_w + C::WBUF_FLOATS;    // only when XS_FLOATS > 0: raw rows of the 32x32 patch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int tower = blockIdx.y + a.tower0;
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
    float4 wr0 = {0.f, 0.f, 0.f, 0.f}, wr1 = wr0, wr2 = wr0;
    if constexpr (!C::NSPLIT) {
        wr0 = reinterpret_cast<const float4*>(wsrc)[wi0];
        if constexpr (C::WPT > 1) wr1 = reinterpret_cast<const float4*>(wsrc)[wi1];
        if constexpr (C::WPT > 2) wr2 = reinterpret_cast<const float4*>(wsrc)[wi2];
    }
    // NSPLIT: wave (pr, ks) = (channel pair, K part); every B operand of its k-steps (chunks ks * CPW .. + CPW, 9 taps each)
    // is requested now and lands under the staging below
    constexpr int KS_CPW = C::NC4 / C::KS, KS_SPW = C::NSPLIT ? KS_CPW * 9 : 1;
    const int ks_pr = C::NSPLIT ? wave % C::NT16 : 0, ks_part = C::NSPLIT ? wave / C::NT16 : 0;
    float ks_b16[KS_SPW], ks_b4[KS_SPW];
    if constexpr (C::NSPLIT) {
        const float* __restrict__ wb = wsrc + (size_t)ks_part * KS_CPW * C::WCHUNK + kq * C::NS16 + lr + ks_pr * 16;                       // W[k = 4s + kq][cout = 16 p + lr]
        const float* __restrict__ wl = wsrc + (size_t)ks_part * KS_CPW * C::WCHUNK + 36 * C::NS16 + kq * C::L4 + (lane & 3) + ks_pr * 4;   // W[k = 4s + kq][cout = 16 NT16 + 4 p + (lane & 3)]
#pragma unroll
        for (int i = 0; i < KS_SPW; ++i) {
            ks_b16[i] = wb[(i / 9) * C::WCHUNK + (i % 9) * 4 * C::NS16];
            ks_b4[i] = wl[(i / 9) * C::WCHUNK + (i % 9) * 4 * C::L4];
        }
    }

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

    // EPI_L1: the band's input rows and the conv1 tap table, fetched now, parked in LDS behind the main buffers after the
    // staging (the main loop's barriers publish them long before the epilogue reads them)
    constexpr int L1_NX = (C::EPI == EPI_L1) ? ((C::HOB + 2) * 32 + C::NTH - 1) / C::NTH : 1;      // gathered voxels per thread
    float4 l1x4 = {0.f, 0.f, 0.f, 0.f};
    float l1xs[L1_NX], l1tab = 0.f;
    if constexpr (C::EPI == EPI_L1) {
        constexpr int XR = C::HOB + 2;
        static_assert(XR * 8 <= C::NTH && 240 <= C::NTH, "one 16-byte piece / table entry per thread");
        const float* __restrict__ v = pick3(a.x0.views, tower);
        if (v) {
            if (tid < XR * 8) l1x4 = reinterpret_cast<const float4*>(v + (size_t)g0 * 1024 + band * C::HOB * 32)[tid];
        } else {
            const int4 smp = x0_sample(a.x0, g0);
            const scs_volume d = a.x0.descs[smp.x];
            const int fl = a.x0.flips ? (a.x0.flips[g0] >> (2 * tower)) & 3 : 0;
            const int fi = (fl & 1) ? 31 : 0, fj = (fl & 2) ? 31 : 0;
#pragma unroll
            for (int q = 0; q < L1_NX; ++q) {
                const int e = min(tid + C::NTH * q, XR * 32 - 1);
                l1xs[q] = gather_voxel(a.x0.vols + d.offset, d.dim_x, d.dim_y, d.dim_z, tower, smp.y, smp.z, smp.w,
                                       (band * C::HOB + (e >> 5)) ^ fi, (e & 31) ^ fj);
            }
        }
        if (tid < 240) l1tab = pick3(a.bnlow, tower)[tid];
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
    } else if constexpr (C::SRC == SRC_BN) {
        // training forward: x = relu(sc * z + sh) while staging; x is also stored (the weight-gradient kernel reads it)
        // (NB > 1: rows [band * HOB, band * HOB + HINB) of every plane; a band stores the rows it owns - the last one its halo too)
        constexpr int HW = C::HIN * C::HIN, HW4 = HW / 4, BW4 = C::HINB * C::HIN / 4, B04 = C::HOB * C::HIN / 4;
        const float4* __restrict__ src4 = reinterpret_cast<const float4*>(insrc + (size_t)g0 * C::CIN * HW) + band * B04;
        const float2* __restrict__ bn2 = reinterpret_cast<const float2*>(pick3(a.bn, tower));
        float* __restrict__ xo = pick3(a.xout, tower);
        float4* __restrict__ xo4 = xo ? reinterpret_cast<float4*>(xo + (size_t)g0 * C::CIN * HW) + band * B04 : nullptr;
        float4* dst4 = reinterpret_cast<float4*>(s_in);
        const int planes = ng * C::CIN;
        const int own4 = (C::NB == 1 || band == C::NB - 1) ? BW4 : B04;
        struct V { float4 v; float2 ss; };
        staged_copy<C::G * C::CIN * BW4, V, C::NTH, 0>(
            tid,
            [&](int e) { const int pl = e / BW4, q = e - pl * BW4; V r; r.ss = bn2[pl % C::CIN];
                         r.v = (pl < planes) ? src4[(size_t)pl * HW4 + q] : float4{0.f, 0.f, 0.f, 0.f}; return r; },
            [&](int e, V r) {
                const int pl = e / BW4, q = e - pl * BW4;
                const float4 y{fmaxf(fmaf(r.v.x, r.ss.x, r.ss.y), 0.f), fmaxf(fmaf(r.v.y, r.ss.x, r.ss.y), 0.f),
                               fmaxf(fmaf(r.v.z, r.ss.x, r.ss.y), 0.f), fmaxf(fmaf(r.v.w, r.ss.x, r.ss.y), 0.f)};
                dst4[pl * (C::PLANE / 4) + q] = y;
                if (xo4 && pl < planes && q < own4) xo4[(size_t)pl * HW4 + q] = y;
            });
    } else if constexpr (C::SRC == SRC_BN_POOL) {
        // training forward: x = maxpool2x2(relu(sc * z + sh)) while staging (one thread per window), x also stored
        // (NB > 1: pooled rows [band * HOB, band * HOB + HINB) of every plane; a band stores the rows it owns - the last one its halo too)
        constexpr int HZ = 2 * C::HIN, HWZ = HZ * HZ, W = C::HIN * C::HIN, BW = C::HINB * C::HIN, B0 = C::HOB * C::HIN;
        const float* __restrict__ src = insrc + (size_t)g0 * C::CIN * HWZ + (size_t)(2 * band * C::HOB) * HZ;
        const float2* __restrict__ bn2 = reinterpret_cast<const float2*>(pick3(a.bn, tower));
        float* __restrict__ xo = pick3(a.xout, tower);
        if (xo) xo += (size_t)g0 * C::CIN * W + band * B0;
        const int planes = ng * C::CIN;
        const int own = (C::NB == 1 || band == C::NB - 1) ? BW : B0;
        struct V { float2 r0, r1, ss; };
        staged_copy<C::G * C::CIN * BW, V, C::NTH, 0>(
            tid,
            [&](int e) {
                const int pl = e / BW, wdx = e - pl * BW, wy = wdx / C::HIN, wx = wdx - wy * C::HIN;
                V r; r.ss = bn2[pl % C::CIN];
                const float* zp = src + (size_t)min(pl, planes - 1) * HWZ + (2 * wy) * HZ + 2 * wx;
                r.r0 = *reinterpret_cast<const float2*>(zp); r.r1 = *reinterpret_cast<const float2*>(zp + HZ);
                return r;
            },
            [&](int e, V r) {
                const int pl = e / BW, wdx = e - pl * BW;
                const float m = fmaxf(fmaxf(fmaf(r.r0.x, r.ss.x, r.ss.y), fmaf(r.r0.y, r.ss.x, r.ss.y)),
                                      fmaxf(fmaf(r.r1.x, r.ss.x, r.ss.y), fmaf(r.r1.y, r.ss.x, r.ss.y)));
                const float y = (pl < planes) ? fmaxf(m, 0.f) : 0.f;
                s_in[pl * C::PLANE + wdx] = y;
                if (xo && pl < planes && wdx < own) xo[(size_t)pl * W + wdx] = y;
            });
    } else if constexpr (C::SRC == SRC_DZ_PAD2 && C::NB == 1 && C::HSRC % 4 == 0) {
        // training backward, whole planes whose rows are float4 multiples: d z = [sc z + sh > 0] sc dUp + ca z + cb from 16-byte
        // pieces of z and dUp (ONE round of loads; the scalar walk over the padded plane below takes four), the zero
        // border is written while the loads are in flight
        constexpr int HS = C::HSRC, HS4 = HS / 4, HW4 = HS * HS4, TOTAL = C::G * C::CIN * HW4;
        constexpr int ITER = (TOTAL + C::NTH - 1) / C::NTH;
        static_assert(ITER <= 8, "one round of loads");
        static_assert(C::PLANE % 2 == 0 && C::WIN % 2 == 0, "8-byte LDS stores");
        const float4* __restrict__ z4 = reinterpret_cast<const float4*>(insrc + (size_t)g0 * C::CIN * HS * HS);
        const float4* __restrict__ d4 = reinterpret_cast<const float4*>(pick3(a.dup, tower) + (size_t)g0 * C::CIN * HS * HS);
        const float4* __restrict__ bn4 = reinterpret_cast<const float4*>(pick3(a.bn, tower));
        float* __restrict__ dzo = pick3(a.xout, tower);
        float4* __restrict__ dzo4 = dzo ? reinterpret_cast<float4*>(dzo + (size_t)g0 * C::CIN * HS * HS) : nullptr;
        const int planes = ng * C::CIN;
        float4 vz[ITER], vd[ITER], vc[ITER];
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int e = min(tid + C::NTH * k, TOTAL - 1), pl = e / HW4;
            const bool ok = pl < planes;
            vz[k] = ok ? z4[e] : float4{0.f, 0.f, 0.f, 0.f};
            vd[k] = ok ? d4[e] : float4{0.f, 0.f, 0.f, 0.f};
            vc[k] = ok ? bn4[pl % C::CIN] : float4{0.f, 0.f, 0.f, 0.f};
        }
        {
            float4* z0 = reinterpret_cast<float4*>(s_in);
            for (int e = tid; e < C::IN_FLOATS_AL / 4; e += C::NTH) z0[e] = float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int e = tid + C::NTH * k;
            if (e < TOTAL) {
                const int pl = e / HW4, r = e - pl * HW4, ys = r / HS4, x4 = r - ys * HS4;
                const float4 z = vz[k], d = vd[k], cf = vc[k];
                float4 o;
                o.x = ((fmaf(z.x, cf.x, cf.y) > 0.f) ? cf.x * d.x : 0.f) + fmaf(cf.z, z.x, cf.w);
                o.y = ((fmaf(z.y, cf.x, cf.y) > 0.f) ? cf.x * d.y : 0.f) + fmaf(cf.z, z.y, cf.w);
                o.z = ((fmaf(z.z, cf.x, cf.y) > 0.f) ? cf.x * d.z : 0.f) + fmaf(cf.z, z.z, cf.w);
                o.w = ((fmaf(z.w, cf.x, cf.y) > 0.f) ? cf.x * d.w : 0.f) + fmaf(cf.z, z.w, cf.w);
                float* q = s_in + pl * C::PLANE + (ys + 2) * C::WIN + 4 * x4 + 2;             // even offset: two 8-byte stores
                *reinterpret_cast<float2*>(q) = float2{o.x, o.y};
                *reinterpret_cast<float2*>(q + 2) = float2{o.z, o.w};
                if (dzo4 && pl < planes) dzo4[e] = o;
            }
        }
    } else if constexpr (C::SRC == SRC_DZ_PAD2) {
        // training backward: d z = [sc z + sh > 0] sc dUp + ca z + cb, zero padded by 2; rows the band owns are stored
        constexpr int HS = C::HSRC, HW = HS * HS, BW = C::HINB * C::WIN;
        const float* __restrict__ zsrc = insrc + (size_t)g0 * C::CIN * HW;
        const float* __restrict__ dsrc = pick3(a.dup, tower) + (size_t)g0 * C::CIN * HW;
        const float4* __restrict__ bn4 = reinterpret_cast<const float4*>(pick3(a.bn, tower));
        float* __restrict__ dzo = pick3(a.xout, tower);
        if (dzo) dzo += (size_t)g0 * C::CIN * HW;
        const int planes = ng * C::CIN;
        struct V { float z, d; float4 cf; };
        auto locate = [&](int e, int& pl, int& r, int& idx, bool& own) {
            pl = e / BW; r = e - pl * BW;
            const int yl = r / C::WIN, xp = r - yl * C::WIN;
            const int ys = band * C::HOB + yl - 2, xs = xp - 2;
            idx = pl * HW + ys * HS + xs;
            own = yl < C::HOB;
            return pl < planes && (unsigned)ys < (unsigned)HS && (unsigned)xs < (unsigned)HS;
        };
        staged_copy<C::G * C::CIN * BW, V, C::NTH, 0>(
            tid,
            [&](int e) {
                int pl, r, idx; bool own;
                const bool ok = locate(e, pl, r, idx, own);
                V v; v.z = ok ? zsrc[idx] : 0.f; v.d = ok ? dsrc[idx] : 0.f;
                v.cf = ok ? bn4[pl % C::CIN] : float4{0.f, 0.f, 0.f, 0.f};
                return v;
            },
            [&](int e, V v) {
                int pl, r, idx; bool own;
                const bool ok = locate(e, pl, r, idx, own);
                const float dz = ((fmaf(v.z, v.cf.x, v.cf.y) > 0.f) ? v.cf.x * v.d : 0.f) + fmaf(v.cf.z, v.z, v.cf.w);
                s_in[pl * C::PLANE + r] = dz;                      // all-zero coefficients outside the plane: dz = 0
                if (dzo && ok && own) dzo[idx] = dz;
            });
    } else if constexpr (C::SRC == SRC_DZ_POOL_PAD2) {
        // same through a 2x2 max-pool: one thread per window of the padded band; dUp lives on the pooled grid and goes
        // to every position that attains the window maximum (Theano MaxPoolGrad), if the maximum survives the ReLU
        constexpr int HS = C::HSRC, HW = HS * HS, HPS = HS / 2, WR = C::HINB / 2, WC = C::WIN / 2, WP = WR * WC;
        const float* __restrict__ zsrc = insrc + (size_t)g0 * C::CIN * HW;
        const float* __restrict__ dsrc = pick3(a.dup, tower) + (size_t)g0 * C::CIN * HPS * HPS;
        const float4* __restrict__ bn4 = reinterpret_cast<const float4*>(pick3(a.bn, tower));
        float* __restrict__ dzo = pick3(a.xout, tower);
        if (dzo) dzo += (size_t)g0 * C::CIN * HW;
        const int planes = ng * C::CIN;
        struct V { float2 r0, r1; float d; float4 cf; };
        auto locate = [&](int e, int& pl, int& wyl, int& wxp, int& wsy, int& wsx) {
            pl = e / WP;
            const int w = e - pl * WP;
            wyl = w / WC; wxp = w - wyl * WC;
            wsy = (band * C::HOB) / 2 + wyl - 1; wsx = wxp - 1;
            return pl < planes && (unsigned)wsy < (unsigned)HPS && (unsigned)wsx < (unsigned)HPS;
        };
        staged_copy<C::G * C::CIN * WP, V, C::NTH, 0>(
            tid,
            [&](int e) {
                int pl, wyl, wxp, wsy, wsx;
                const bool ok = locate(e, pl, wyl, wxp, wsy, wsx);
                V v;
                if (ok) {
                    const float* zp = zsrc + (size_t)pl * HW + (2 * wsy) * HS + 2 * wsx;
                    v.r0 = *reinterpret_cast<const float2*>(zp); v.r1 = *reinterpret_cast<const float2*>(zp + HS);
                    v.d = dsrc[(size_t)pl * HPS * HPS + wsy * HPS + wsx];
                    v.cf = bn4[pl % C::CIN];
                } else {
                    v.r0 = v.r1 = float2{0.f, 0.f}; v.d = 0.f; v.cf = float4{0.f, 0.f, 0.f, 0.f};
                }
                return v;
            },
            [&](int e, V v) {
                int pl, wyl, wxp, wsy, wsx;
                const bool ok = locate(e, pl, wyl, wxp, wsy, wsx);
                const float y0 = fmaf(v.r0.x, v.cf.x, v.cf.y), y1 = fmaf(v.r0.y, v.cf.x, v.cf.y);
                const float y2 = fmaf(v.r1.x, v.cf.x, v.cf.y), y3 = fmaf(v.r1.y, v.cf.x, v.cf.y);
                const float m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
                const float g = (m > 0.f) ? v.cf.x * v.d : 0.f;
                const float2 o0{(y0 == m ? g : 0.f) + fmaf(v.cf.z, v.r0.x, v.cf.w), (y1 == m ? g : 0.f) + fmaf(v.cf.z, v.r0.y, v.cf.w)};
                const float2 o1{(y2 == m ? g : 0.f) + fmaf(v.cf.z, v.r1.x, v.cf.w), (y3 == m ? g : 0.f) + fmaf(v.cf.z, v.r1.y, v.cf.w)};
                float* d = s_in + pl * C::PLANE + (2 * wyl) * C::WIN + 2 * wxp;
                *reinterpret_cast<float2*>(d) = o0;
                *reinterpret_cast<float2*>(d + C::WIN) = o1;
                if (dzo && ok && 2 * wyl < C::HOB) {
                    float* q = dzo + (size_t)pl * HW + (2 * wsy) * HS + 2 * wsx;
                    *reinterpret_cast<float2*>(q) = o0;
                    *reinterpret_cast<float2*>(q + HS) = o1;
                }
            });
    } else {
        static_assert(C::FUSED1 && C::G == 1, "fused conv1 variants take one patch per workgroup");
        // conv1 operands and the raw rows are fetched together: one global round trip, not two
        const float* __restrict__ w1g = pick3(a.w1, tower);
        const float w1a = w1g[min(tid, kW1Floats - 1)];
        const float w1b = w1g[min(tid + C::NTH, kW1Floats - 1)];
        if constexpr (C::SRC == SRC_CONV1) {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(insrc + (size_t)g0 * 1024 + band * C::HOB * 32);
            if (tid < C::XROWS * 8) reinterpret_cast<float4*>(s_x)[tid] = src[tid];
        } else if constexpr (C::SRC == SRC_X0) {
            load_x0_rows<C::XROWS, C::NTH>(a.x0, tower, g0, band * C::HOB, s_x, tid);
        } else {
            const int cx = a.centers[(size_t)g0 * 3 + 0], cy = a.centers[(size_t)g0 * 3 + 1], cz = a.centers[(size_t)g0 * 3 + 2];
            for (int e = tid; e < C::XROWS * 32; e += C::NTH)
                s_x[e] = gather_voxel(insrc, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, band * C::HOB + (e >> 5), e & 31);
        }
        float* s_w1 = s_x + C::XS_ROWS_FLOATS + 4;
        if (tid < kW1Floats) s_w1[tid] = w1a;              // kW1Floats = 260: two loads per lane at 256 threads
        if (tid + C::NTH < kW1Floats) s_w1[tid + C::NTH] = w1b;
        __syncthreads();
        if constexpr (C::STAMP) { if (tid == 0) stp[5] = __builtin_amdgcn_s_memtime(); }
        conv1_to_lds<C::PLANE, C::HINB, C::NW>(s_x, s_w1, s_in, wave, lane);
    }

    if constexpr (C::STAMP) { if (tid == 0) stp[1] = __builtin_amdgcn_s_memtime(); }
    if constexpr (C::EPI == EPI_L1) {
        constexpr int XR = C::HOB + 2;
        float* s_x0 = smem + (C::MAIN_FLOATS > C::EPI_FLOATS ? C::MAIN_FLOATS : C::EPI_FLOATS);
        if (pick3(a.x0.views, tower)) { if (tid < XR * 8) reinterpret_cast<float4*>(s_x0)[tid] = l1x4; }
        else {
#pragma unroll
            for (int q = 0; q < L1_NX; ++q) { const int e = tid + C::NTH * q; if (e < XR * 32) s_x0[e] = l1xs[q]; }
        }
        if (tid < 8) s_x0[XR * 32 + tid] = 0.f;
        if (tid < 240) s_x0[C::L1_X_FLOATS + tid] = l1tab;
    }
    // ---- weight chunk 0 -> LDS
    if constexpr (!C::NSPLIT) {
        float4* dst = reinterpret_cast<float4*>(s_w);
        if (tid < C::WCHUNK4) dst[tid] = wr0;
        if constexpr (C::WPT > 1) { if (tid + C::NTH < C::WCHUNK4) dst[tid + C::NTH] = wr1; }
        if constexpr (C::WPT > 2) { if (tid + 2 * C::NTH < C::WCHUNK4) dst[tid + 2 * C::NTH] = wr2; }
    }

    if constexpr (C::NSPLIT) {
        // GEMM-style K split (see ConvCfg): this wave multiplies ALL M tiles of the workgroup with its channel pair over its
        // share of the input-channel chunks; A operands from the staged planes, B operands already in registers.
        int abase_n[C::TILES];
#pragma unroll
        for (int j = 0; j < C::TILES; ++j) {
            const int m = min(j * 16 + lr, C::MTOT - 1);
            const int g = m / C::PIXB, r = m - g * C::PIXB, y = r / C::HO, x = r - y * C::HO;
            abase_n[j] = (g * C::CIN + kq) * C::PLANE + y * C::WIN + x;
        }
        f32x4 accn[C::TILES], acc4n[C::TILES];
#pragma unroll
        for (int j = 0; j < C::TILES; ++j) { accn[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc4n[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        __syncthreads();                                     // the staged planes are visible
        if constexpr (C::STAMP) { if (tid == 0) stp[2] = __builtin_amdgcn_s_memtime(); }
        {
            const float* __restrict__ sa = s_in + ks_part * KS_CPW * 4 * C::PLANE;
#pragma unroll
            for (int i = 0; i < KS_SPW; ++i) {
                const int off = (i / 9) * 4 * C::PLANE + ((i % 9) / 3) * C::WIN + (i % 9) % 3;
#pragma unroll
                for (int j = 0; j < C::TILES; ++j) {
                    const float av = sa[abase_n[j] + off];
                    accn[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ks_b16[i], accn[j], 0, 0, 0);
                    acc4n[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(ks_b4[i], av, acc4n[j], 0, 0, 0);
                }
            }
        }
        if constexpr (C::STAMP) { if (tid == 0) stp[3] = __builtin_amdgcn_s_memtime(); }
        __syncthreads();                                     // every wave has read its planes: the LDS becomes the output stage
        // partial tiles -> s_red[ks][row m][cout] (the 4-wide blocks' k-subgroup partials are combined across the lanes first)
        constexpr int MROWS = C::TILES * 16;
        float* s_red = smem + C::OUT_FLOATS_AL;
        {
            float* rw = s_red + (size_t)ks_part * MROWS * C::COUT;
            const int xor32_addr = (lane ^ 32) << 2;
#pragma unroll
            for (int j = 0; j < C::TILES; ++j) {
                const float val4 = ksub_reduce(acc4n[j], kq, xor32_addr);                               // row j * 16 + (lane & 15), cout 16 NT16 + 4 p + kq
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) rw[(j * 16 + kq * 4 + r4) * C::COUT + ks_pr * 16 + lr] = accn[j][r4];
                rw[(j * 16 + (lane & 15)) * C::COUT + C::NT16 * 16 + ks_pr * 4 + kq] = val4;
            }
        }
        if constexpr (C::STAMP) { if (tid == 0) stp[6] = __builtin_amdgcn_s_memtime(); }
        __syncthreads();
        // sum of the K parts in part order -> the restaged outputs [g][cout][OSTR]
        {
            float* s_o = smem;
            constexpr int TOT = C::G * C::COUT * C::PIXB;
#pragma unroll 2
            for (int e = tid; e < TOT; e += C::NTH) {
                const int co = e % C::COUT, m = e / C::COUT;                   // consecutive lanes: consecutive channels of one row
                const int g = m / C::PIXB, r = m - g * C::PIXB;
                float v = s_red[m * C::COUT + co];
#pragma unroll
                for (int q = 1; q < C::KS; ++q) v += s_red[(q * MROWS + m) * C::COUT + co];
                s_o[(g * C::COUT + co) * C::OSTR + r] = v;
            }
        }
        __syncthreads();
        if constexpr (C::STAMP) { if (tid == 0) stp[7] = __builtin_amdgcn_s_memtime(); }
    } else {
    // Tile ownership rotates with the workgroup: wave w of workgroup b owns tiles (w + b) mod NW, + NW, ...  When the tiles do
    // not divide by the waves (25 = 7 + 6 + 6 + 6, 7 = 2 + 2 + 2 + 1) the short share would otherwise always fall on the same
    // waves - and with one wave of every workgroup per SIMD, always on the same SIMDs of the CU.
#ifndef SCS_TILE_ROT
#define SCS_TILE_ROT 1
#endif
    const int wt = (SCS_TILE_ROT && (C::NW & (C::NW - 1)) == 0) ? ((wave + (int)blockIdx.x) & (C::NW - 1)) : wave;
    // ---- per-lane A base offsets (floats) for the wave's tiles; out-of-range tiles are clamped, never stored
    int abase[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        int m = ((C::SHARED && j == C::TPW - 1) ? C::TILES - 1 : wt + C::NW * j) * 16 + lr;
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
    const bool last_ok = (wt + C::NW * (C::TPW - 1)) < C::TILES;
    // The first k-step (chunk 0, tap 0) of every tile a wave always owns may start its accumulators from the instruction's inline
    // zero instead of zero-initialised registers (4 v_mov per accumulator tile: 56 per wave in the fused conv1+conv2 kernel, on
    // the vector pipe the f32 MFMAs share): the inference kernels peel chunk 0 off the loop for that (fused conv1+conv2+pool:
    // 18.03 -> 17.80 ms per 100 000 patches).  The training kernels - one generation of workgroups per launch, bound by latency,
    // not by vector issue - keep the plain loop (a same-GPU A/B of the step with the peeled form: 0.625 against 0.619 ms).
    const f32x4 zero_acc = {0.f, 0.f, 0.f, 0.f};
#define SCS_CONV_CHUNK(c4, FIRST)                                                                                          \
    {                                                                                                                      \
        if (c4 + 1 < C::NC4) {                                                                                             \
            const float4* __restrict__ src = reinterpret_cast<const float4*>(wsrc + (size_t)(c4 + 1) * C::WCHUNK);         \
            wr0 = src[wi0];                                                                                                \
            if constexpr (C::WPT > 1) wr1 = src[wi1];                                                                      \
            if constexpr (C::WPT > 2) wr2 = src[wi2];                                                                      \
        }                                                                                                                  \
        const float* __restrict__ sw = s_w + (c4 & 1) * C::WCHUNK;                                                         \
        const float* __restrict__ sa = s_in + c4 * 4 * C::PLANE;                                                           \
        _Pragma("unroll")                                                                                                  \
        for (int tap = 0; tap < 9; ++tap) {                                                                                \
            const int toff = (tap / 3) * C::WIN + (tap % 3);                                                               \
            float b[C::NT16 > 0 ? C::NT16 : 1], b4[C::NT4 > 0 ? C::NT4 : 1];                                               \
        _Pragma("unroll")                                                                                                  \
            for (int nt = 0; nt < C::NT16; ++nt) b[nt] = sw[boff + tap * 4 * C::NS16 + nt * 16];                           \
        _Pragma("unroll")                                                                                                  \
            for (int u = 0; u < C::NT4; ++u) b4[u] = sw[loff + tap * 4 * C::L4 + u * 4];                                   \
        _Pragma("unroll")                                                                                                  \
            for (int j = 0; j < C::TPW; ++j) {                                                                             \
                if constexpr (C::SHARED) {                                                                                 \
                    if (j == C::TPW - 1) {                                     /* the odd tile: this wave's share of its channel blocks */ \
                        if (wt <= C::NT16) {                                                                               \
                            const float av = sa[abase[j] + toff];                                                          \
        _Pragma("unroll")                                                                                                  \
                            for (int nt = 0; nt < C::NT16; ++nt)                                                           \
                                if (wt == nt) acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], acc[j][nt], 0, 0, 0); \
                            if (wt == C::NT16) {                                                                           \
        _Pragma("unroll")                                                                                                  \
                                for (int u = 0; u < C::NT4; ++u)                                                           \
                                    acc4[j][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, acc4[j][u], 0, 0, 0);       \
                            }                                                                                              \
                        }                                                                                                  \
                        continue;                                                                                          \
                    }                                                                                                      \
                }                                                                                                          \
                if (!C::SHARED && C::LAST_RAGGED && j == C::TPW - 1 && !last_ok) continue;   /* wave-uniform: this wave has no such tile */ \
                const float av = sa[abase[j] + toff];                                                                      \
                /* (a tile some waves skip keeps its zero-initialised registers) */                                        \
                const bool from_zero = FIRST && tap == 0 && !(C::LAST_RAGGED && j == C::TPW - 1);                          \
        _Pragma("unroll")                                                                                                  \
                for (int nt = 0; nt < C::NT16; ++nt)                                                                       \
                    acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt], from_zero ? zero_acc : acc[j][nt], 0, 0, 0); \
                /* 16 blocks of 4 couts x 4 pixels: block (l>>2) = (k-subgroup kq, pixel quad); partial sums over kq */    \
                /* are combined in the epilogue */                                                                         \
        _Pragma("unroll")                                                                                                  \
                for (int u = 0; u < C::NT4; ++u)                                                                           \
                    acc4[j][u] = __builtin_amdgcn_mfma_f32_4x4x1f32(b4[u], av, from_zero ? zero_acc : acc4[j][u], 0, 0, 0); \
            }                                                                                                              \
        }                                                                                                                  \
        if (c4 + 1 < C::NC4) {                                                                                             \
            float4* dst = reinterpret_cast<float4*>(s_w + ((c4 + 1) & 1) * C::WCHUNK);                                     \
            if (tid < C::WCHUNK4) dst[tid] = wr0;                                                                          \
            if constexpr (C::WPT > 1) { if (tid + C::NTH < C::WCHUNK4) dst[tid + C::NTH] = wr1; }                          \
            if constexpr (C::WPT > 2) { if (tid + 2 * C::NTH < C::WCHUNK4) dst[tid + 2 * C::NTH] = wr2; }                  \
        }                                                                                                                  \
        __syncthreads();                                                                                                   \
    }
    if constexpr (C::RAW) {
        for (int c4 = 0; c4 < C::NC4; ++c4) SCS_CONV_CHUNK(c4, false)
    } else {
        // (as a lambda: the same text placed inline schedules worse - fused conv1+conv2+pool at 0.621 instead of 0.636 of peak, same GPU)
        auto chunk = [&](const int c4, auto first_chunk) {
            constexpr bool FIRST = decltype(first_chunk)::value;
            SCS_CONV_CHUNK(c4, FIRST)
        };
        chunk(0, std::true_type{});
        for (int c4 = 1; c4 < C::NC4; ++c4) chunk(c4, std::false_type{});
    }
#undef SCS_CONV_CHUNK

    if constexpr (C::STAMP) { if (tid == 0) stp[3] = __builtin_amdgcn_s_memtime(); }

    // ---- epilogue: D row = 4*kq + r (pixel / window position), column = lr (cout).  Results are restaged in
    // LDS as [g][cout][OSTR] (every wave has passed the loop's last barrier, so the operand buffers are free) and
    // leave as whole rows: 64 lanes -> 256 contiguous bytes, instead of 64 scattered dwords per store.
    float* s_o = smem;
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const bool odd = C::SHARED && j == C::TPW - 1;
        const int t = odd ? C::TILES - 1 : wt + C::NW * j;
        if (t >= C::TILES) continue;
#pragma unroll
        for (int nt = 0; nt < C::NT16; ++nt) {
            if (odd && wt != nt) continue;                                   // the odd tile: wave nt holds this block
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
        const bool odd = C::SHARED && j == C::TPW - 1;
        const int t = odd ? C::TILES - 1 : wt + C::NW * j;
        if (t >= C::TILES || (odd && wt != C::NT16)) continue;               // the odd tile: wave NT16 holds the 4-blocks
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
    }   // !NSPLIT
    float* s_o = smem;
    if constexpr (C::EPI != EPI_L1) {
        float* __restrict__ outp = pick3(a.out, tower) + (size_t)g0 * C::COUT * C::OUTPIX + band * C::OUTB;
        if constexpr (C::OUT_VEC4) {
            constexpr int O4 = C::OUTB / 4, ITER = (C::G * C::COUT * O4 + C::NTH - 1) / C::NTH;
            const int total = ng * C::COUT * O4;
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
    if constexpr (C::EPI == EPI_RAW_STATS && C::FUSED1 && C::HOB % 2 == 0 && C::HO % 2 == 0) {
        // the pooled-by-sign copy of the band's raw outputs (ConvArgs::pout): rows of s_o are [cout][HOB x HO]
        float* __restrict__ po = pick3(a.pout, tower);
        if (po) {
            constexpr int WB2 = (C::HOB / 2) * C::HP, PR = C::HP / 2, ITEMS = C::COUT * (C::HOB / 2) * PR;   // windows per channel and band; window pairs per row
            static_assert(C::HP % 2 == 0 && C::OSTR % 4 == 0 && C::HO % 4 == 0, "two windows = one 16-byte piece of each of their two rows");
            po += (size_t)g0 * C::COUT * (C::HP * C::HP) + band * WB2;
            const float* __restrict__ gm = pick3(a.bias, tower);
#pragma unroll
            for (int e0 = 0; e0 < ITEMS; e0 += C::NTH) {
                const int e = e0 + tid;
                if (e < ITEMS) {
                    const int row = e / PR, wp = e - row * PR, co = row / (C::HOB / 2), wy = row - co * (C::HOB / 2);
                    const float* __restrict__ q = s_o + co * C::OSTR + (2 * wy) * C::HO + 4 * wp;
                    const float4 r0 = *reinterpret_cast<const float4*>(q), r1 = *reinterpret_cast<const float4*>(q + C::HO);
                    const bool up = gm[co] >= 0.f;
                    const float a0 = up ? fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y)) : fminf(fminf(r0.x, r0.y), fminf(r1.x, r1.y));
                    const float a1 = up ? fmaxf(fmaxf(r0.z, r0.w), fmaxf(r1.z, r1.w)) : fminf(fminf(r0.z, r0.w), fminf(r1.z, r1.w));
                    *reinterpret_cast<float2*>(po + (size_t)co * (C::HP * C::HP) + wy * C::HP + 2 * wp) = float2{a0, a1};
                }
            }
        }
    }
    if constexpr (C::EPI == EPI_RAW_STATS || C::EPI == EPI_RAW_BNB || C::EPI == EPI_RAW_BNB_POOL) {
        // rows of s_o are (patch, channel); LPR lanes share a row, then combine through the crossbar
        constexpr int ROWS = C::G * C::COUT;
        constexpr int LPR = ROWS * 16 <= C::NTH ? 16 : ROWS * 8 <= C::NTH ? 8 : ROWS * 4 <= C::NTH ? 4 : ROWS * 2 <= C::NTH ? 2 : 1;
        constexpr int RPI = C::NTH / LPR;                    // rows per iteration
        float* __restrict__ sp = a.stats + ((size_t)tower * a.n * C::NB) * C::COUT * 2;
        const int sub = tid % LPR;
        float2 unit_val = make_float2(0.f, 0.f);              // this thread's row of the last pass (G > 1: the only pass)
#pragma unroll
        for (int row0 = 0; row0 < ROWS; row0 += RPI) {
            const int row = row0 + tid / LPR;
            const bool live = row < ng * C::COUT;
            const int g = row / C::COUT, c = row - g * C::COUT;
            float sm = 0.f, sq = 0.f;
            if (live) {
                const float* __restrict__ rp = s_o + row * C::OSTR;
                if constexpr (C::EPI == EPI_RAW_STATS) {
                    // sums about the row's own first value: no cancellation however far the channel's mean is from zero
                    const float p0 = rp[0];
                    for (int i = sub; i < C::OUTB; i += LPR) { const float v = rp[i] - p0; sm += v; sq = fmaf(v, v, sq); }
                } else {
                    // (d beta, d gamma) partials of the block below: dy = dX through (pool and) ReLU, xhat = (z - mean) inv_std
                    const float4 cf = reinterpret_cast<const float4*>(pick3(a.bnlow, tower))[c];     // sc, sh, mean, inv
                    if constexpr (C::EPI == EPI_RAW_BNB) {
                        const float* __restrict__ zp = pick3(a.zlow, tower) + ((size_t)(g0 + g) * C::COUT + c) * C::PIX + band * C::OUTB;
                        constexpr int IT = (C::OUTB + LPR - 1) / LPR, KB = IT <= 12 ? IT : (IT + (IT + 11) / 12 - 1) / ((IT + 11) / 12);
#pragma unroll
                        for (int i0 = 0; i0 < IT; i0 += KB) {
                            float zv[KB];
#pragma unroll
                            for (int k = 0; k < KB; ++k) zv[k] = zp[min(sub + LPR * (i0 + k), C::OUTB - 1)];
#pragma unroll
                            for (int k = 0; k < KB; ++k) {
                                const int i = sub + LPR * (i0 + k);
                                if (i0 + k < IT && i < C::OUTB) {
                                    const float dy = fmaf(zv[k], cf.x, cf.y) > 0.f ? rp[i] : 0.f;
                                    sm += dy; sq = fmaf(dy, (zv[k] - cf.z) * cf.w, sq);
                                }
                            }
                        }
                    } else {
                        constexpr int HZ = 2 * C::HO;
                        const float* __restrict__ zp = pick3(a.zlow, tower) + ((size_t)(g0 + g) * C::COUT + c) * HZ * HZ + (2 * band * C::HOB) * HZ;
                        constexpr int IT = (C::OUTB + LPR - 1) / LPR, KB = IT <= 8 ? IT : (IT + (IT + 7) / 8 - 1) / ((IT + 7) / 8);
#pragma unroll
                        for (int i0 = 0; i0 < IT; i0 += KB) {
                            float2 r0[KB], r1[KB];
#pragma unroll
                            for (int k = 0; k < KB; ++k) {
                                const int i = min(sub + LPR * (i0 + k), C::OUTB - 1), yy = i / C::HO, xx = i - yy * C::HO;
                                const float* q = zp + (2 * yy) * HZ + 2 * xx;
                                r0[k] = *reinterpret_cast<const float2*>(q); r1[k] = *reinterpret_cast<const float2*>(q + HZ);
                            }
#pragma unroll
                            for (int k = 0; k < KB; ++k) {
                                const int i = sub + LPR * (i0 + k);
                                if (i0 + k < IT && i < C::OUTB) {
                                    const float y0 = fmaf(r0[k].x, cf.x, cf.y), y1 = fmaf(r0[k].y, cf.x, cf.y);
                                    const float y2 = fmaf(r1[k].x, cf.x, cf.y), y3 = fmaf(r1[k].y, cf.x, cf.y);
                                    const float m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
                                    const float dv = m > 0.f ? rp[i] : 0.f;
                                    const float d0 = y0 == m ? dv : 0.f, d1 = y1 == m ? dv : 0.f, d2 = y2 == m ? dv : 0.f, d3 = y3 == m ? dv : 0.f;
                                    sm += (d0 + d1) + (d2 + d3);
                                    sq = fmaf(d0, (r0[k].x - cf.z) * cf.w, sq); sq = fmaf(d1, (r0[k].y - cf.z) * cf.w, sq);
                                    sq = fmaf(d2, (r1[k].x - cf.z) * cf.w, sq); sq = fmaf(d3, (r1[k].y - cf.z) * cf.w, sq);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
            if (sub == 0 && live) {
                float2* dst = reinterpret_cast<float2*>(sp + (((size_t)(g0 + g) * C::NB + band) * C::COUT + c) * 2);
                float2 val;
                if constexpr (C::EPI == EPI_RAW_STATS) {
                    // (mean, sum of squared deviations) of this unit of OUTB values - combined exactly by the finalize step
                    const float mu = sm * (1.f / C::OUTB);
                    val = make_float2(s_o[row * C::OSTR] + mu, fmaxf(sq - sm * mu, 0.f));
                } else {
                    val = make_float2(sm, sq);
                }
                if constexpr (C::FIN) st_sc1(dst, val); else *dst = val;
                unit_val = val;
            }
        }
        if constexpr (C::EPI == EPI_RAW_STATS && C::G > 1 && LPR == 1 && ROWS <= C::NTH && C::NB == 1) {
            // the same statistics once more as ONE unit per (workgroup, channel): the G patches' units combined in patch order
            // by the parallel-variance formula, in double (a.wgstats; the per-patch units above stay what the finalize launch reads)
            // (tilings with one thread per (patch, channel) row and one pass: the conv5 forward kernel of train_cfg.h)
            if (a.wgstats) {
                __syncthreads();                                           // s_o has been read: the LDS is free
                float2* s_u = reinterpret_cast<float2*>(smem);
                if (tid < ROWS) s_u[tid] = unit_val;
                __syncthreads();
                if (tid < C::COUT) {
                    const double m0 = (double)s_u[tid].x;
                    double sd = 0, sd2 = 0, q = 0;
                    for (int g = 0; g < ng; ++g) { const float2 u = s_u[g * C::COUT + tid]; const double d = (double)u.x - m0; sd += d; sd2 += d * d; q += (double)u.y; }
                    const double dm = sd / ng;
                    reinterpret_cast<float2*>(a.wgstats)[((size_t)tower * gridDim.x + blockIdx.x) * C::COUT + tid] =
                        make_float2((float)(m0 + dm), (float)(q + (double)C::OUTB * fmax(sd2 - sd * dm, 0.0)));
                }
            }
        }
        if constexpr (C::FIN) {
            __syncthreads();                                               // s_o has been read: the LDS is free
            double* s_acc = reinterpret_cast<double*>(smem);               // [NTH][3]
            volatile int* s_flag = reinterpret_cast<volatile int*>(smem + C::NTH * 6);
            const unsigned int nwg = gridDim.x, grp = blockIdx.x / kFinGroup, groups = (nwg + kFinGroup - 1) / kFinGroup;
            unsigned int* cnt = a.fin.counter + tower * fin_counter_words(nwg);
            if (take_ticket(cnt + (size_t)kFinLine * (1 + grp), min((unsigned int)kFinGroup, nwg - grp * kFinGroup), s_flag)) {
                // this workgroup completed its group: the group's units (G per workgroup) -> one record per channel
                const float2* part = reinterpret_cast<const float2*>(sp);
                const int units = a.n * C::NB;                             // one unit per workgroup: unit index = blockIdx.x
                const int u0 = (int)grp * kFinGroup;
                double* grec = a.fin.grp + ((size_t)tower * groups + grp) * kFinRec * 2;
                const double per_unit = a.fin.count / units;
                // (a group of kFinGroup workgroups holds kFinGroup * upw units: reduced in slices of kFinGroup units when upw > 1)
                static_assert(C::G == 1, "in-kernel finalize: one unit per workgroup");
                if constexpr (C::EPI == EPI_RAW_STATS) bn_grp_forward<C::COUT>(part, u0, min(kFinGroup, units - u0), per_unit, grec);
                else bn_grp_backward<C::COUT>(part, u0, min(kFinGroup, units - u0), grec);
                if (take_ticket(cnt, groups, s_flag)) {
                    const double* rec = a.fin.grp + (size_t)tower * groups * kFinRec * 2;
                    if constexpr (C::EPI == EPI_RAW_STATS)
                        bn_top_forward<C::COUT, C::NTH>(rec, (int)groups, units, per_unit, pick3(a.fin.bn, tower), reinterpret_cast<float2*>(pick3(a.fin.ss, tower)),
                                                        reinterpret_cast<float4*>(pick3(a.fin.coef, tower)), s_acc);
                    else
                        bn_top_backward<C::COUT, C::NTH>(rec, (int)groups, a.fin.count, reinterpret_cast<const float4*>(pick3(a.bnlow, tower)),
                                                         pick3(a.fin.bn, tower), reinterpret_cast<float4*>(pick3(a.fin.coef, tower)), s_acc);
                }
            }
        }
    }
    if constexpr (C::EPI == EPI_L1) {
        // s_o = dX1 of this band: [20][OSTR], pixel p = y * 30 + x over HOB rows.  (1) ReLU mask of the block below from the
        // recomputed conv1 output (folded BN: y = sc z + sh); (2) A[c][tap] = sum_p dy[c][p] * x0[p + tap] and d beta[c] =
        // sum_p dy[c][p] on the matrix cores (k = pixels, n = taps, tap 9 = constant one); (3) the four waves' partial sums
        // are added in wave order and leave as [20][10] per (patch, band).
        const float* s_x0 = smem + (C::MAIN_FLOATS > C::EPI_FLOATS ? C::MAIN_FLOATS : C::EPI_FLOATS);
        const float* s_l1 = s_x0 + C::L1_X_FLOATS;                      // [20][12]: 9 flipped taps, sc, sh, pad
        float* s_red = smem + C::OUT_FLOATS;
        {
            // (1) conv1 on the matrix cores once more (K = 9 taps padded to 12, N = 20 = 16 + 4, M = the band's 180 pixels)
            // and zero dX1 wherever sc z + sh <= 0
            constexpr int NPIX = C::PIXB, T1 = (NPIX + 15) / 16, TW = (T1 + C::NW - 1) / C::NW;
            int koff[3];
            float bm[3], bl[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int k = 4 * s3 + kq, tk = k < 9 ? k : 8;
                koff[s3] = (tk / 3) * 32 + tk % 3;
                bm[s3] = k < 9 ? s_l1[lr * 12 + k] : 0.f;
                bl[s3] = k < 9 ? s_l1[(16 + (lane & 3)) * 12 + k] : 0.f;
            }
            const float sc_m = s_l1[lr * 12 + 9], sh_m = s_l1[lr * 12 + 10];
            const float sc_l = s_l1[(16 + kq) * 12 + 9], sh_l = s_l1[(16 + kq) * 12 + 10];
#pragma unroll
            for (int jj = 0; jj < TW; ++jj) {
                const int t = wave + C::NW * jj, tc = min(t, T1 - 1);
                const int p = min(tc * 16 + lr, NPIX - 1), y = p / 30, x = p - y * 30;
                f32x4 zm = {0.f, 0.f, 0.f, 0.f}, zl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const float av = s_x0[y * 32 + x + koff[s3]];
                    zm = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bm[s3], zm, 0, 0, 0);      // D[pixel 4 kq + i][c = lr]
                    zl = __builtin_amdgcn_mfma_f32_4x4x1f32(bl[s3], av, zl, 0, 0, 0);
                }
                const float zv = ksub_reduce(zl, kq, (lane ^ 32) << 2);                       // c = 16 + kq, pixel lr of the tile
                if (t < T1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pp = tc * 16 + kq * 4 + i;
                        if (pp < NPIX && !(fmaf(zm[i], sc_m, sh_m) > 0.f)) s_o[lr * C::OSTR + pp] = 0.f;
                    }
                    const int pl = tc * 16 + lr;
                    if (pl < NPIX && !(fmaf(zv, sc_l, sh_l) > 0.f)) s_o[(16 + kq) * C::OSTR + pl] = 0.f;
                }
            }
        }
        __syncthreads();
        const int toff = lr < 9 ? (lr / 3) * 32 + lr % 3 : 0;
        // four independent accumulator sets: under the other workgroups' matrix phases a dependent MFMA chain advances one
        // link per pipe round trip, so a 16-deep chain costs four times a 4-deep one
        f32x4 am4[4], al4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { am4[q] = f32x4{0.f, 0.f, 0.f, 0.f}; al4[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ri = 0; ri < (C::HOB + C::NW - 1) / C::NW; ++ri) {     // unrolled: a runtime trip count makes the eight accumulators
            const int r = wave + C::NW * ri;                             // loop-carried copies (32 v_mov per round)
            if (r >= C::HOB) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = 4 * j + kq;
                const bool ok = x < 30;
                const float bx = ok ? (lr < 9 ? s_x0[r * 32 + x + toff] : (lr == 9 ? 1.f : 0.f)) : 0.f;
                const float a16 = ok ? s_o[lr * C::OSTR + r * 30 + x] : 0.f;
                const float a4 = ok ? s_o[(16 + (lane & 3)) * C::OSTR + r * 30 + x] : 0.f;
                am4[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16, bx, am4[j & 3], 0, 0, 0);       // D[c = 4 kq + i][tap = lr]
                al4[j & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, bx, al4[j & 3], 0, 0, 0);          // k-subgroup partials of c = 16 + i
            }
        }
        f32x4 am, al;
#pragma unroll
        for (int i = 0; i < 4; ++i) { am[i] = (am4[0][i] + am4[1][i]) + (am4[2][i] + am4[3][i]); al[i] = (al4[0][i] + al4[1][i]) + (al4[2][i] + al4[3][i]); }
        const float lv = ksub_reduce(al, kq, (lane ^ 32) << 2);                          // c = 16 + kq, tap = lr
        float* rw = s_red + wave * 320;
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[(4 * kq + i) * 16 + lr] = am[i];
        rw[(16 + kq) * 16 + lr] = lv;
        __syncthreads();
        if (tid < 200) {
            const int c = tid / 10, t = tid - c * 10;
            const float v = (s_red[c * 16 + t] + s_red[320 + c * 16 + t]) + (s_red[640 + c * 16 + t] + s_red[960 + c * 16 + t]);
            a.stats[(((size_t)tower * a.n + g0) * C::NB + band) * 200 + tid] = v;
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
inline void conv_launch(const ConvArgs& a, hipStream_t st, int ntowers = kTowers) {
    if (a.n <= 0) return;
    dim3 grid(((a.n + C::G - 1) / C::G) * C::NB, ntowers);
    hipLaunchKernelGGL(conv3x3_mfma_kernel<C>, grid, dim3(C::NTH), C::LDS_BYTES, st, a);
}
// the guarded form (a.guard set; instantiated only where a caller asks for it)
template <class C>
inline hipError_t conv_set_attr_guarded() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<C, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}
template <class C>
inline void conv_launch_guarded(const ConvArgs& a, hipStream_t st, int ntowers = kTowers) {
    if (a.n <= 0 || !a.guard) return;
    dim3 grid(((a.n + C::G - 1) / C::G) * C::NB, ntowers);
    hipLaunchKernelGGL((conv3x3_mfma_kernel<C, true>), grid, dim3(C::NTH), C::LDS_BYTES, st, a);
}

}  // namespace scs
