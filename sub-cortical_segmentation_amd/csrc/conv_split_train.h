// Training forward of the conv1 + conv2 block on the f16 matrix cores with f32-grade results (the split of conv_split.h: every f32
// operand as two f16 pieces, three MFMAs per product block, f32 accumulation) - option train_split of the training step.
//
// Same arithmetic scheme as conv12_split_kernel (conv1 on the exact f32 MFMA with this step's batch statistics folded into its
// operands, its output scaled, split and kept in LDS as two f16 planes [pixel][channel]; conv2 = 3 x 3 correlation as runs of
// 3 * Cin halves per tap row), but what leaves the kernel is what the training step needs (conv_mfma.h, EPI_RAW_STATS of the
// fused conv1 + conv2 configuration):
//   out    raw conv2 output z [n][20][28][28] (the batch norm of the block follows in the next kernel's staging)
//   pout   per 2 x 2 window the value pool1 keeps after BN + ReLU: the window's maximum where gamma >= 0, its minimum otherwise
//   stats  per (patch, channel) the mean and the sum of squared deviations of the 784 values (one unit per patch-tower: the
//          bn2 finalize launch combines the units by the parallel-variance formula, scs_train.hip: bn_fwd_final_kernel)
// conv2 main loop: an LDS fragment (tap row R of the window's 4 x 2 pixel rows, column parity, half of the run) is read ONCE and
// used for both window rows it serves (output row 0 with tap row R, output row 1 with tap row R - 1): 16 fragment pairs per
// pooling group instead of 24.
//
// Replaces, in training, Conv2DLayer -> BatchNormLayer -> rectify -> Conv2DLayer (raw) of /root/reference/cnn_cort/nets.py:171-172.
#pragma once
#include "conv_mfma.h"
#include "conv_split.h"

#ifndef SPT_ABLATE
#define SPT_ABLATE 0         // tools/stbench timing experiments only (results are wrong when set): 1 no z2 / pout stores, 2 no statistics,
#endif                       // 3 neither, 4 conv1 stage only for a workgroup's first patch, 5 no conv2 MFMAs behind the first fragment

namespace scs {

constexpr int kSpT12StatFloats = 16 * 20 * 2 + 4 + kSp12W1Floats;      // (mean, M2) per (wave, tile group) and channel, the overflow word, conv1's operands
constexpr size_t kSpT12LdsBytes = kSp12LdsBytes + (size_t)kSpT12StatFloats * 4;

struct SplitTrain12Args {
    X0Src x0;                     // the input patches (given, or cut from the volumes: conv_mfma.h)
    const float* w1[kTowers];     // kSp12W1Floats: conv1 taps [slot 4 dy + dx][20] + shift [20] with the batch statistics folded, times kSplitActScale
    const uint4* w2[kTowers];     // kSp12W2Vec: raw conv2 weights (correlation form) times a power of two, split (pack_split_raw)
    const float* inv2[kTowers];   // one float: 1 / (kSplitActScale * weight scale)
    const float* gamma2[kTowers]; // [20]: sign picks the extremum pout keeps
    float* out[kTowers];          // z2 [n][20][28][28]
    float* pout[kTowers];         // [n][20][14][14]
    float* stats;                 // [tower][n][20][2] (mean, M2)
    int n;
    FinArgs fin;                  // fin.counter != null: the workgroup of a tower that finishes last also finalizes the block's batch statistics
                                  // (conv_mfma.h: take_ticket / bn_grp_forward / bn_top_forward; needs a grid of n or a multiple of kFinGroup
                                  // workgroups per tower, see conv12_split_train_launch) - no bn2 finalize launch
    unsigned int* flag;           // set when a scaled conv1 activation left the f16 range (zeroed by the step's prep launch; the step's
                                  // loss-mean workgroup turns the loss into NaN: a NaN in the activations would be swallowed by the next ReLU)
};

// sum over the 16 lanes of a DPP row (lanes with the same lane >> 4), result in every lane; fixed order
__device__ __forceinline__ float sp_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));   // row_ror:1
    return v;
}

// One workgroup per CU slot, persistent: workgroup b serves tower b % 3, patches b / 3, b / 3 + G, ...
__global__ __launch_bounds__(256, 2) void conv12_split_train_kernel(SplitTrain12Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned char* s_hi = sp_smem;
    unsigned char* s_lo = sp_smem + kSp12PlaneBytes;
    float* s_x = reinterpret_cast<float*>(sp_smem + 2 * kSp12PlaneBytes);
    float* s_st = s_x + kSp12XFloats;                       // [4 waves][4 groups: three of 64 values, the odd tile's 16][20][2]
    volatile unsigned* s_ovf = reinterpret_cast<volatile unsigned*>(s_st + 640);
    float* s_w1 = s_st + 644;                               // conv1's operands: read back at the top of every patch (registers are for the conv2 phase)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    for (int e = tid; e < 160 / 4; e += 256) {              // pads: zero once (never written afterwards)
        reinterpret_cast<unsigned*>(s_hi + 900 * 40)[e] = 0u;
        reinterpret_cast<unsigned*>(s_lo + 900 * 40)[e] = 0u;
    }
    if (tid < 64) s_x[1024 + tid] = 0.f;
    if (tid == 0) *s_ovf = 0u;
    // ---- resident operands.  Every wave holds ALL of conv2's split weights (24 x 16 bytes per lane): fetched once per workgroup into
    // the (still empty) planes and read from there by the four waves - a quarter of the L2 traffic of 4 x 64 lanes fetching them
    // (one generation of workgroups starts at the same time: 510 x 98 KB)
    for (int e = tid; e < kSp12W1Floats; e += 256) s_w1[e] = pick3(a.w1, tower)[e];
    {
        const uint4* w2 = pick3(a.w2, tower);
        uint4 t6[kSp12W2Vec / 256];
#pragma unroll
        for (int r = 0; r < kSp12W2Vec / 256; ++r) t6[r] = w2[r * 256 + tid];
#pragma unroll
        for (int r = 0; r < kSp12W2Vec / 256; ++r) reinterpret_cast<uint4*>(sp_smem)[r * 256 + tid] = t6[r];
    }
    __syncthreads();
    f16x8 wh[6][2], wl[6][2];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wh[s][m] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(sp_smem)[((s * 2 + m) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, reinterpret_cast<const uint4*>(sp_smem)[((s * 2 + m) * 2 + 1) * 64 + lane]);
        }
    __syncthreads();                                        // the planes are free again
    const float inv = *pick3(a.inv2, tower);
    // sign of gamma per channel slot cs = 4 m + i (channel 4 j + i, or 16 + i from lane group 0): bit cs set = gamma < 0 = pout keeps the
    // MINIMUM (taken as the maximum of the values with their sign bits flipped, flipped back)
    unsigned negmask = 0u;
    {
        const float* g2 = pick3(a.gamma2, tower);
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (g2[4 * j + i] < 0.f) negmask |= 1u << i; if (g2[16 + i] < 0.f) negmask |= 16u << i; }
    }
    auto flip = [&](float v, int cs) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ ((negmask >> cs) << 31)); };
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
    float* const outp = pick3(a.out, tower);
    float* const poutp = pick3(a.pout, tower);
    float2* const statp = reinterpret_cast<float2*>(a.stats) + (size_t)tower * a.n * 20;
    if (wg < a.n) {
        X0RowRegs<32, 256> xr;
        xr.load(a.x0, tower, wg, 0, tid);
        xr.store(s_x, tid);
    }
    // the (mean, M2) of the 12 tile groups of 64 values and of the odd tile's 16 (computed by wave `oddw`) -> the patch-tower's statistics
    // unit, by the parallel-variance formula about the first group's mean (threads 0..19, behind a barrier; fixed order)
    auto combine = [&](int p, int oddw) {
        if (tid < 20) {
            const float2* u = reinterpret_cast<const float2*>(s_st) + tid;
            const float m0 = u[0].x;
            float sd = 0.f, q = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float cnt = (g & 3) < 3 ? 64.f : ((g >> 2) == oddw ? 16.f : 0.f);
                const float2 v = u[g * 20];
                if (cnt > 0.f) { sd = fmaf(cnt, v.x - m0, sd); q += v.y; }
            }
            const float mean = m0 + sd * (1.f / 784.f);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float cnt = (g & 3) < 3 ? 64.f : ((g >> 2) == oddw ? 16.f : 0.f);
                const float d = u[g * 20].x - mean;
                if (cnt > 0.f) q = fmaf(cnt * d, d, q);
            }
            if (a.fin.counter) st_sc1(statp + (size_t)p * 20 + tid, make_float2(mean, q)); else statp[(size_t)p * 20 + tid] = make_float2(mean, q);
        }
    };
    int it = 0, pprev = -1;
    for (int p = wg; p < a.n; p += nwg, ++it) {
        __syncthreads();                                     // raw patch in place; every wave is done with the planes and the statistics slots of the previous patch
        if (pprev >= 0) combine(pprev, (it - 1) & 3);
        X0RowRegs<32, 256> xr;                               // the next patch: requested now, parked in LDS behind the conv1 stage (which reads this one)
        const bool more = p + nwg < a.n;
        if (more) xr.load(a.x0, tower, p + nwg, 0, tid);
        float vmax = 0.f;
        float a1[3], a4[9];
        f32x4 b1m, b1r;
#pragma unroll
        for (int s = 0; s < 3; ++s) a1[s] = s_w1[(4 * s + j) * 20 + n];
#pragma unroll
        for (int t = 0; t < 9; ++t) a4[t] = s_w1[(4 * (t / 3) + t % 3) * 20 + 16 + (lane & 3)];
#pragma unroll
        for (int i = 0; i < 4; ++i) { b1m[i] = s_w1[240 + 4 * j + i]; b1r[i] = s_w1[240 + 16 + i]; }
        // ---- conv1 on the f32 matrix cores (as conv12_split_kernel): 60 tiles of 16 pixels over the 32-wide row domain
#pragma unroll 1
        for (int g5 = 0; g5 < ((SPT_ABLATE == 4 && it > 0) ? 0 : 3); ++g5) {
            float bx[5][3];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int s = 0; s < 3; ++s) bx[u][s] = s_x[16 * (wave + 4 * (5 * g5 + u)) + n + j + 32 * s];
            f32x4 d[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) d[u] = b1m;
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int u = 0; u < 5; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bx[u][s], d[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int T = wave + 4 * (5 * g5 + u), y = T >> 1, x = 16 * (T & 1) + n;
                sp_split4_store(d[u], s_hi, s_lo, x < 30 ? (y * 30 + x) * 40 + 8 * j : 900 * 40 + 32 + 8 * j, vmax);
            }
        }
        for (int Q = wave; Q < 15; Q += 4) {
            float bx[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) bx[t] = s_x[64 * Q + lane + (t / 3) * 32 + t % 3];
            f32x4 d = b1r;
#pragma unroll
            for (int t = 0; t < 9; ++t) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[t], bx[t], d, 0, 0, 0);
            const int pp = 64 * Q + lane, y = pp >> 5, x = pp & 31;
            sp_split4_store(d, s_hi, s_lo, x < 30 ? (y * 30 + x) * 40 + 32 : 900 * 40 + 64, vmax);
        }
        if (!(vmax <= kSplitOverflow)) *s_ovf = 1u;          // (an infinite value too; NOT a NaN: vmax is built with fmaxf, which drops NaN
                                                             // operands - a NaN input goes the way it goes in the f32 kernels, to 0 at the next ReLU)
        __syncthreads();                                     // planes complete; nobody reads the raw patch any more
        if (more) xr.store(s_x, tid);
        // ---- conv2 on the f16 matrix cores
        float* const op = outp + (size_t)p * (20 * 784);
        float* const pp_ = poutp + (size_t)p * (20 * 196);
        float2* const stw = reinterpret_cast<float2*>(s_st) + wave * 4 * 20;
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const unsigned char* h0 = s_hi + rd0[k];
            const unsigned char* h1 = s_hi + rd1[k];
            const unsigned char* l0 = s_lo + rd0[k];
            const unsigned char* l1 = s_lo + rd1[k];
            f32x4 acc[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // fragment f = 4 R + 2 half + column parity: pixel row R (0..3) of the window pair rows, read once, used by output row 0
            // (tap row R) and output row 1 (tap row R - 1); three fragments in flight
            f16x8 fh[3], fl[3];
            auto rdfrag = [&](int f) {
                const int off = (f >> 2) * 1200 + (f & 1) * 40 + ((f >> 1) & 1) * 64;
                fh[f % 3] = sp_frag(h0 + off, h1 + off); fl[f % 3] = sp_frag(l0 + off, l1 + off);
            };
            rdfrag(0); rdfrag(1);
#pragma unroll
            for (int f = 0; f < (SPT_ABLATE == 5 ? 1 : 16); ++f) {
                const int R = f >> 2, hh = (f >> 1) & 1, qc = f & 1;
                if (f + 2 < 16) rdfrag(f + 2);
                const f16x8 bh = fh[f % 3], bl = fl[f % 3];
#pragma unroll
                for (int piece = 0; piece < 3; ++piece) {
#pragma unroll
                    for (int use = 0; use < 2; ++use) {
                        const int dy = R - use;                      // use 0: output row 0, use 1: output row 1
                        if (dy < 0 || dy > 2) continue;
                        const int s = 2 * dy + hh, q = 2 * use + qc;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            if (piece == 0) acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[q][m], 0, 0, 0);
                            else if (piece == 1) acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[q][m], 0, 0, 0);
                            else acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[q][m], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the group: window w of this lane, positions q = 2 qr + qc
            const int w = 16 * (wave + 4 * k) + n, wy = w / 14, wx = w - 14 * wy;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (m == 1 && j != 0) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cs = 4 * m + i, ch = m == 0 ? 4 * j + i : 16 + i;
                    float z[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] = acc[q][m][i] * inv;
                    float* zo = op + ch * 784 + (2 * wy) * 28 + 2 * wx;
                    if ((SPT_ABLATE == 1 || SPT_ABLATE == 3) && z[0] != 123.25f) continue;
                    *reinterpret_cast<float2*>(zo) = make_float2(z[0], z[1]);
                    *reinterpret_cast<float2*>(zo + 28) = make_float2(z[2], z[3]);
                    pp_[ch * 196 + w] = flip(fmaxf(fmaxf(flip(z[0], cs), flip(z[1], cs)), fmaxf(flip(z[2], cs), flip(z[3], cs))), cs);
                    // the group's 64 values of this channel: mean and sum of squared deviations (differences to the lane's first value,
                    // then about the group's mean: sum (v - mg)^2 = sum (d - e)^2 = s2 - 2 e s1 + 4 e^2, d = v - z[0], e = mg - z[0])
                    if (SPT_ABLATE == 2 || SPT_ABLATE == 3) continue;
                    const float d1 = z[1] - z[0], d2 = z[2] - z[0], d3 = z[3] - z[0];
                    const float s1 = d1 + d2 + d3, s2 = fmaf(d3, d3, fmaf(d2, d2, d1 * d1));
                    const float mg = sp_row_sum(fmaf(4.f, z[0], s1)) * (1.f / 64.f), e = mg - z[0];
                    const float q2 = sp_row_sum(fmaf(4.f * e, e, fmaf(-2.f * e, s1, s2)));
                    if (n == 0) stw[k * 20 + ch] = make_float2(mg, q2);
                }
            }
        }
        const bool odd = wave == (it & 3);
        if (odd) {                                           // the odd tile: windows 192..195, the four positions of each in one quad of lanes
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
            const int w = 192 + (n >> 2), q = n & 3, wy = w / 14, wx = w - 14 * wy;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cs = 4 * m + i, ch = m == 0 ? 4 * j + i : 16 + i;
                    const float z = acc[m][i] * inv;
                    const float ext = flip(sp_quad_max(flip(z, cs)), cs);        // (every lane takes part in the quad / row exchanges)
                    const float mg = sp_row_sum(z) * (1.f / 16.f), d = z - mg;
                    const float q2 = sp_row_sum(d * d);
                    if (m == 1 && j != 0) continue;
                    op[ch * 784 + (2 * wy + (q >> 1)) * 28 + 2 * wx + (q & 1)] = z;
                    if (q == 0) pp_[ch * 196 + w] = ext;
                    if (n == 0) stw[3 * 20 + ch] = make_float2(mg, q2);
                }
            }
        }
        pprev = p;
    }
    __syncthreads();
    if (pprev >= 0) combine(pprev, (it - 1) & 3);
    if (*s_ovf != 0u && tid == 0) atomicOr(a.flag, 1u);
    if (a.fin.counter) {
        // in-kernel finalize (as the FIN configurations of conv_mfma.h): workgroups in groups of kFinGroup; the workgroup that completes
        // its group reduces the group's statistics units - unit = patch, workgroup w wrote units w, w + nwg, ...: with nwg a multiple
        // of kFinGroup (or one round) every slice of kFinGroup consecutive units belongs to ONE group - to one record per slice, the
        // workgroup that completes the tower combines the records in index order (the arithmetic does not depend on who arrives last)
        __syncthreads();                                     // the planes are free: scratch of the finalize
        double* s_acc = reinterpret_cast<double*>(sp_smem);  // [256][3]
        volatile int* s_flag = reinterpret_cast<volatile int*>(sp_smem + 256 * 3 * 8);
        const unsigned int nw = (unsigned int)nwg, grp = (unsigned int)wg / kFinGroup, groups = (nw + kFinGroup - 1) / kFinGroup;
        unsigned int* cnt = a.fin.counter + tower * fin_counter_words(nw);
        if (take_ticket(cnt + (size_t)kFinLine * (1 + grp), min((unsigned int)kFinGroup, nw - grp * kFinGroup), s_flag)) {
            const int recs = (a.n + kFinGroup - 1) / kFinGroup;
            double* rec = a.fin.grp + (size_t)tower * recs * kFinRec * 2;
            for (int u0 = (int)grp * kFinGroup; u0 < a.n; u0 += nwg)
                bn_grp_forward<20>(statp, u0, min(kFinGroup, a.n - u0), 784.0, rec + (size_t)(u0 / kFinGroup) * kFinRec * 2);
            if (take_ticket(cnt, groups, s_flag))
                bn_top_forward<20, 256>(rec, recs, a.n, 784.0, pick3(a.fin.bn, tower), reinterpret_cast<float2*>(pick3(a.fin.ss, tower)),
                                        reinterpret_cast<float4*>(pick3(a.fin.coef, tower)), s_acc);
        }
    }
}

inline hipError_t conv12_split_train_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_split_train_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSpT12LdsBytes);
}
// Two workgroups per CU: 512 slots for the 3 towers.  Up to batch 256 one workgroup per patch-tower (at 256: 768 workgroups, the
// last third starts as slots free up - 41.6 us against 45.7 us for 160 persistent workgroups per tower, tools/stbench.hip); beyond,
// 160 persistent workgroups per tower (a multiple of kFinGroup: the in-kernel finalize's slices).  wgs_per_tower > 0 overrides.
inline void conv12_split_train_launch(const SplitTrain12Args& a, hipStream_t st, int wgs_per_tower = 0) {
    static_assert(160 % kFinGroup == 0, "finalize slices");
    if (wgs_per_tower <= 0) wgs_per_tower = a.n <= 256 ? a.n : 160;
    int g = wgs_per_tower < a.n ? wgs_per_tower : a.n;
    if (a.fin.counter && g < a.n && g % kFinGroup != 0) {                        // several rounds: the finalize's slices need whole groups
        g = g - g % kFinGroup < kFinGroup ? kFinGroup : g - g % kFinGroup;
        if (g > a.n) g = a.n;
    }
    if (g < 1) return;
    hipLaunchKernelGGL(conv12_split_train_kernel, dim3(3 * g), dim3(256), kSpT12LdsBytes, st, a);
}

// Split operands of a RAW conv layer (training: no batch-norm fold) by ONE workgroup of 256 threads: weight scale = the power of
// two that puts max |w| into [2^12, 2^13), every lane's 8 halves per (k-step, channel tile), high and low piece (the layout of
// pack_split_kernel, scs_api.hip); scale_out[0] = 1 / (kSplitActScale * weight scale).
__device__ __forceinline__ void pack_split_raw(const float* __restrict__ W, int cin, int cout, uint4* __restrict__ wp, float* __restrict__ scale_out,
                                               float* s_red /* [256] */) {
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int e = tid; e < cout * cin * 9; e += 256) mx = fmaxf(mx, fabsf(W[e]));
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
                v = W[((o * cin + c) * 3 + (2 - dy)) * 3 + (2 - dx)] * sw;            // true convolution -> correlation
            }
            const _Float16 hh = (_Float16)v, ll = (_Float16)(v - (float)hh);
            hb[pos] = __builtin_bit_cast(unsigned short, hh); lb[pos] = __builtin_bit_cast(unsigned short, ll);
        }
        wp[((st * MT + m) * 2 + 0) * 64 + lane] = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
        wp[((st * MT + m) * 2 + 1) * 64 + lane] = make_uint4(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16), lb[4] | (lb[5] << 16), lb[6] | (lb[7] << 16));
    }
    if (tid == 0) scale_out[0] = 1.f / (kSplitActScale * sw);
}

}  // namespace scs
