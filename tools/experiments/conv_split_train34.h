// EXPERIMENT (round 5) - NOT part of libscs: conv3 / conv4 of the training forward on the split f16 matrix-core form, one wave per
// patch-tower (weights resident in the wave's registers, staging / matrix phase / epilogue without workgroup barriers).  Built into
// the step behind option train_split, correct (all of tests/test_gpu_train_split.py passed: gradients within 1e-4 of scale of the
// oracle at 14 batch sizes, raw outputs within 1e-5 of the f32 kernels'), and SLOWER than the f32-MFMA kernels it replaced:
// conv3 fwd 35.2 us against 32.0, conv4 fwd 41.3 against 42.1 at batch 256, and the two extra weight-pack jobs doubled the prep
// launch (20 -> 38 us): step 0.543 ms against 0.538.  One wave per SIMD walking load -> convert -> multiply -> reduce -> store
// serially hides nothing; the f32 kernels' twelve waves per CU do.  Kept here for the record (profiles/NOTES_r05.md); to build it
// again, include it behind conv_split_train.h (it uses SplitTrain12Args' helpers: sp_row_sum, pack_split_raw, FinArgs).
#pragma once
#include "conv_split_train.h"

namespace scs {

// =====================================================================================================================
// conv3 / conv4 of the training forward on the split.  ONE WAVE per patch-tower, four per workgroup, one workgroup per CU (the
// layer's split weights - 144 / 288 registers - stay in the wave's registers: one wave per SIMD): a wave stages its own patch-tower
// (raw output z of the block below, f32 [CIN][HIN][HIN] -> x = relu(fma(z, sc, sh)), stored as the block output X for the weight-
// gradient kernel, times kSplitActScale, split, into two f16 planes [pixel][channel] of its own LDS slice), multiplies (tiles of 16
// consecutive output pixels, all channel tiles of a fragment), stores the raw conv output z [COUT][HO][HO] and leaves the patch-
// tower's statistics unit (mean, M2 per channel) - no workgroup barrier between those phases, so the four waves of a CU drift apart
// and one's staging sits under another's matrix phase.  At batch 256 the 768 patch-towers are ONE generation on 192 CUs.
// The workgroup that finishes last finalizes the layer's batch statistics (fin.counter != null; conv_mfma.h).
template <int CIN_, int COUT_, int HIN_>
struct SpTCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, HO = HIN_ - 2, PIX = HO * HO, IN_PX = HIN_ * HIN_;
    static constexpr int NT = (PIX + 15) / 16, MT = (COUT_ + 15) / 16, NW = 4;
    static constexpr int PS = 2 * CIN_, RS = HIN_ * PS;                        // pixel / row stride of a plane, bytes
    static constexpr int KSR = (3 * CIN_ + 31) / 32, KS = 3 * KSR;
    static constexpr int IN_B = IN_PX * PS, PLANE_B = IN_B + 32;               // + pad: the last runs overrun (against zero weights; kept zero)
    static constexpr int WV = KS * MT * 2;                                     // uint4 of weights per lane
    static constexpr int TB = 3;                                               // tiles per pass (the last pass may be short)
    static constexpr int CG = CIN_ / 4;                                        // channel groups of the staging
    static constexpr int ROUNDS = (IN_PX + 63) / 64;                           // pixels per lane in the staging
    static constexpr size_t PLANES_BYTES = (size_t)NW * 2 * PLANE_B, W_BYTES = (size_t)WV * 64 * 16;
    static constexpr int NPASS = (NT + TB - 1) / TB;
    static constexpr size_t UNITS_OFF = (PLANES_BYTES > W_BYTES ? PLANES_BYTES : W_BYTES);          // per wave: [NPASS][COUT] float2 (mean, M2) of a pass
    static constexpr size_t LDS_BYTES = UNITS_OFF + (size_t)NW * NPASS * COUT_ * 8 + 64;
    static_assert(IN_B % 16 == 0 && CIN_ % 4 == 0 && COUT_ % 4 == 0, "16-byte pieces, channel groups of 4");
    static_assert(32 * KSR - 3 * CIN_ <= 16, "run padding stays inside the plane pad");
    static_assert(LDS_BYTES >= 256 * 3 * 8 + 16, "finalize scratch");
};
using SpT3 = SpTCfg<20, 40, 14>;
using SpT4 = SpTCfg<40, 40, 12>;

struct SplitTrainArgs {
    const float* in[kTowers];     // raw conv output of the block below [n][CIN][HIN][HIN] (conv3: the pooled-by-sign plane of conv2)
    const float* bn[kTowers];     // float2 [CIN]: (scale, shift) of the block below
    float* xout[kTowers];         // block output x = relu(sc z + sh) [n][CIN][HIN][HIN] (the weight-gradient kernel reads it)
    const uint4* w[kTowers];      // split raw weights [k-step][channel tile][piece][lane] (pack_split_raw)
    const float* inv[kTowers];    // one float: 1 / (kSplitActScale * weight scale)
    float* out[kTowers];          // raw conv output z [n][COUT][HO][HO]
    float* stats;                 // [tower][n][COUT][2] (mean, M2): one unit per patch-tower
    int n;
    FinArgs fin;
    unsigned int* flag;
};

template <class C>
__global__ __launch_bounds__(256, 1) void conv_split_train_kernel(SplitTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    // ---- the layer's weights: once per workgroup through LDS into every wave's registers
    {
        const uint4* w = pick3(a.w, tower);
        constexpr int PER = (C::WV * 64 + 255) / 256;
#pragma unroll
        for (int r0 = 0; r0 < PER; r0 += 8) {
            uint4 t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) if (r0 + r < PER) t[r] = w[min((r0 + r) * 256 + tid, C::WV * 64 - 1)];
#pragma unroll
            for (int r = 0; r < 8; ++r) if (r0 + r < PER && (r0 + r) * 256 + tid < C::WV * 64) reinterpret_cast<uint4*>(sp_smem)[(r0 + r) * 256 + tid] = t[r];
        }
    }
    __syncthreads();
    f16x8 wh[C::KS][C::MT], wl[C::KS][C::MT];
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            wh[s][m] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(sp_smem)[((s * C::MT + m) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(sp_smem)[((s * C::MT + m) * 2 + 1) * 64 + lane]);
        }
    __syncthreads();                                        // the planes are free
    unsigned char* const pl_hi = sp_smem + (size_t)wave * 2 * C::PLANE_B;
    unsigned char* const pl_lo = pl_hi + C::PLANE_B;
    if (lane < 8) { reinterpret_cast<unsigned*>(pl_hi + C::IN_B)[lane] = 0u; reinterpret_cast<unsigned*>(pl_lo + C::IN_B)[lane] = 0u; }   // pads
    const float inv = *pick3(a.inv, tower);
    const float2* __restrict__ bn2 = reinterpret_cast<const float2*>(pick3(a.bn, tower));
    const float* __restrict__ inp = pick3(a.in, tower);
    float* __restrict__ xo = pick3(a.xout, tower);
    float* __restrict__ outp = pick3(a.out, tower);
    float2* const statp = reinterpret_cast<float2*>(a.stats) + (size_t)tower * a.n * C::COUT;
    // per-lane byte offsets of this lane's pixel in every tile (tile t: output pixels 16 t .. 16 t + 15, clamped), its lane group's
    // 16-byte chunk with the two halves in the order of conv_split.h's bank note
    int toff[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) { const int p = min(16 * t + n, C::PIX - 1), y = p / C::HO, x = p - y * C::HO; toff[t] = y * C::RS + x * C::PS + 16 * j; }
    const int h8 = 8 * (j & 1);
    float vmax = 0.f;
    for (int p = wg * C::NW + wave; p < a.n; p += nwg * C::NW) {
        // ---- staging: lane = pixel (ROUNDS of 64), channel groups of 4; every load is one coalesced row segment of a channel plane
        {
            const float* __restrict__ src = inp + (size_t)p * C::CIN * C::IN_PX;
            float* __restrict__ xd = xo + (size_t)p * C::CIN * C::IN_PX;
#pragma unroll
            for (int r = 0; r < C::ROUNDS; ++r) {
                const int px = 64 * r + lane, pc = min(px, C::IN_PX - 1);
                float v[C::CG][4];
#pragma unroll
                for (int g = 0; g < C::CG; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[g][i] = src[(4 * g + i) * C::IN_PX + pc];
#pragma unroll
                for (int g = 0; g < C::CG; ++g) {
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float2 ss = bn2[4 * g + i];
                        y[i] = fmaxf(fmaf(v[g][i], ss.x, ss.y), 0.f);
                    }
                    if (px < C::IN_PX) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) xd[(4 * g + i) * C::IN_PX + px] = y[i];
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] *= kSplitActScale;
                        vmax = fmaxf(fmaxf(vmax, y[0]), fmaxf(y[1], fmaxf(y[2], y[3])));
                        f16x4 hh, ll;
                        sp_split4(y, hh, ll);
                        *reinterpret_cast<f16x4*>(pl_hi + px * C::PS + 8 * g) = hh;
                        *reinterpret_cast<f16x4*>(pl_lo + px * C::PS + 8 * g) = ll;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // this wave's LDS writes are ordered in front of its reads (same wave: in order)
        // ---- matrix phase, TB tiles per pass; every pass leaves (mean, M2) of its pixels per channel in the wave's unit table (LDS)
        float2* const ut = reinterpret_cast<float2*>(sp_smem + C::UNITS_OFF) + wave * C::NPASS * C::COUT;
        float* const op = outp + (size_t)p * C::COUT * C::PIX;
#pragma unroll
        for (int i0 = 0; i0 < C::NT; i0 += C::TB) {
            constexpr int TBX = C::TB;
            f32x4 acc[TBX][C::MT];
#pragma unroll
            for (int i = 0; i < TBX; ++i)
#pragma unroll
                for (int m = 0; m < C::MT; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            f16x8 fh[2][TBX], fl[2][TBX];
            auto frags = [&](int s, int slot, int i) {
                if (i0 + i >= C::NT) return;
                const int off = (s / C::KSR) * C::RS + (s % C::KSR) * 64 + toff[i0 + i < C::NT ? i0 + i : 0];
                fh[slot][i] = sp_frag(pl_hi + off + h8, pl_hi + off + 8 - h8);
                fl[slot][i] = sp_frag(pl_lo + off + h8, pl_lo + off + 8 - h8);
            };
#pragma unroll
            for (int i = 0; i < TBX; ++i) frags(0, 0, i);
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
#pragma unroll
                for (int i = 0; i < TBX; ++i) {
                    if (i0 + i < C::NT) {
#pragma unroll
                        for (int m = 0; m < C::MT; ++m) {
                            acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], fh[s & 1][i], acc[i][m], 0, 0, 0);
                            acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fl[s & 1][i], acc[i][m], 0, 0, 0);
                            acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fh[s & 1][i], acc[i][m], 0, 0, 0);
                        }
                    }
                    if (s + 1 < C::KS) frags(s + 1, (s + 1) & 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // epilogue of the pass: stores, and per channel the mean and the sum of squared deviations of the pass's NP pixels (differences
            // to the lane's first value r, then about the pass mean: sum (v - mp)^2 = s2 - 2 e s1 + cnt e^2, e = mp - r)
            const int NP = (C::PIX - 16 * i0) < 16 * TBX ? (C::PIX - 16 * i0) : 16 * TBX;        // (a constant once the pass loop is unrolled)
            float cnt = 0.f;
#pragma unroll
            for (int i = 0; i < TBX; ++i) if (i0 + i < C::NT && 16 * (i0 + i) + n < C::PIX) cnt += 1.f;
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
                if (16 * m + 4 * j >= C::COUT) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 16 * m + 4 * j + q;
                    const float r = acc[0][m][q] * inv;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < TBX; ++i) {
                        if (i0 + i >= C::NT) continue;
                        const int px = 16 * (i0 + i) + n;
                        const float z = acc[i][m][q] * inv;
                        if ((16 * (i0 + i) + 15 < C::PIX) || px < C::PIX) {
                            op[ch * C::PIX + px] = z;
                            const float d = z - r;
                            s1 += d; s2 = fmaf(d, d, s2);
                        }
                    }
                    const float mp = sp_row_sum(fmaf(cnt, r, s1)) * (1.f / NP), e = mp - r;
                    const float q2 = sp_row_sum(fmaf(cnt * e, e, fmaf(-2.f * e, s1, s2)));
                    if (n == 0) ut[(i0 / TBX) * C::COUT + ch] = make_float2(mp, q2);
                }
            }
        }
        // ---- the patch-tower's statistics unit from the passes' (lane = channel; parallel-variance formula about the first pass's mean)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < C::COUT) {
            const float m0 = ut[lane].x;
            float sd = 0.f, q = 0.f;
#pragma unroll
            for (int ps = 0; ps < C::NPASS; ++ps) {
                constexpr int TBX = C::TB;
                const float np = (float)((C::PIX - 16 * TBX * ps) < 16 * TBX ? (C::PIX - 16 * TBX * ps) : 16 * TBX);
                const float2 u = ut[ps * C::COUT + lane];
                sd = fmaf(np, u.x - m0, sd); q += u.y;
            }
            const float mean = m0 + sd * (1.f / C::PIX);
#pragma unroll
            for (int ps = 0; ps < C::NPASS; ++ps) {
                constexpr int TBX = C::TB;
                const float np = (float)((C::PIX - 16 * TBX * ps) < 16 * TBX ? (C::PIX - 16 * TBX * ps) : 16 * TBX);
                const float d = ut[ps * C::COUT + lane].x - mean;
                q = fmaf(np * d, d, q);
            }
            if (a.fin.counter) st_sc1(statp + (size_t)p * C::COUT + lane, make_float2(mean, q));
            else statp[(size_t)p * C::COUT + lane] = make_float2(mean, q);
        }
    }
    if (!(vmax <= kSplitOverflow)) atomicOr(a.flag, 1u);
    if (a.fin.counter) {
        // in-kernel finalize: as conv12_split_train_kernel; workgroup w wrote the units NW w .. NW w + NW - 1 (+ rounds of nwg NW)
        __syncthreads();
        double* s_acc = reinterpret_cast<double*>(sp_smem);
        volatile int* s_flag = reinterpret_cast<volatile int*>(sp_smem + 256 * 3 * 8);
        const unsigned int nw = (unsigned int)nwg, grp = (unsigned int)wg / kFinGroup, groups = (nw + kFinGroup - 1) / kFinGroup;
        unsigned int* cnt = a.fin.counter + tower * fin_counter_words(nw);
        if (take_ticket(cnt + (size_t)kFinLine * (1 + grp), min((unsigned int)kFinGroup, nw - grp * kFinGroup), s_flag)) {
            const int recs = (a.n + kFinGroup - 1) / kFinGroup;
            double* rec = a.fin.grp + (size_t)tower * recs * kFinRec * 2;
            // the group's units: [NW kFinGroup grp, + NW kFinGroup) of every round: slices of kFinGroup units, one record each
            for (int r0 = (int)grp * kFinGroup * C::NW; r0 < a.n; r0 += nwg * C::NW)
                for (int u0 = r0; u0 < min(r0 + kFinGroup * C::NW, a.n); u0 += kFinGroup)
                    bn_grp_forward<C::COUT>(statp, u0, min(kFinGroup, a.n - u0), (double)C::PIX, rec + (size_t)(u0 / kFinGroup) * kFinRec * 2);
            if (take_ticket(cnt, groups, s_flag))
                bn_top_forward<C::COUT, 256>(rec, recs, a.n, (double)C::PIX, pick3(a.fin.bn, tower), reinterpret_cast<float2*>(pick3(a.fin.ss, tower)),
                                             reinterpret_cast<float4*>(pick3(a.fin.coef, tower)), s_acc);
        }
    }
}

template <class C>
inline hipError_t conv_split_train_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_train_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}
// one workgroup (four patch-towers at a time) per CU and tower third: up to 85 workgroups per tower, a multiple of kFinGroup beyond
// one round (the in-kernel finalize's slices: a workgroup's round-r units must stay inside its group's slice range)
template <class C>
inline void conv_split_train_launch(const SplitTrainArgs& a, hipStream_t st) {
    const int units = (a.n + C::NW - 1) / C::NW;
    const int g = units <= 85 ? units : 80;
    if (g < 1) return;
    hipLaunchKernelGGL(conv_split_train_kernel<C>, dim3(3 * g), dim3(256), C::LDS_BYTES, st, a);
}

}  // namespace scs
