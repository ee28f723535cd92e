// libscs - training step (forward in training mode, backward, Adam) and labelled evaluation.
//
// Replaces what nolearn's train_iter_/eval_iter_ compile out of the Lasagne graph of
// /root/reference/cnn_cort/nets.py:170-246: conv -> BatchNorm (batch statistics, eps 1e-4, running averages
// alpha 0.1) -> ReLU -> PReLU (identity on ReLU output, zero alpha gradient: SURVEY F2) -> max-pool, dropout 0.5,
// dense layers, softmax + mean categorical cross-entropy (nets.py:235), Adam (nets.py:236-237).
//
// Shape of one step (every contraction on the fp32 matrix cores; BN / ReLU / pool never run as passes of their own):
//   forward   moments of the 32x32 inputs (9x9 tap autocorrelation R and tap sums S per tower, one MFMA product per patch)
//             -> conv1's batch statistics follow algebraically (mean = w.S/M, var = w'Cov w), so conv1 + BN1 + ReLU is
//             recomputed inside the conv2 kernel exactly as at inference and its 72 KB/patch output never exists in HBM;
//             conv i+1 applies BN_i + ReLU (+ pool) while STAGING the raw output of conv i (conv_mfma.h: SRC_BN*), the
//             raw outputs carry per-(patch band, channel) statistics partials from the epilogue (EPI_RAW_STATS) and a
//             tiny finalize launch turns them into scale / shift.
//   backward  the data-gradient conv of layer i computes d z_i while staging (SRC_DZ*: ReLU / pool routing + the BN
//             backward affine form) and its epilogue leaves the (d beta, d gamma) partials of layer i-1 (EPI_RAW_BNB*);
//             for layer 1 the epilogue (EPI_L1) leaves A[c][tap] = sum dy x0 and sum dy, from which d W1, d gamma1,
//             d beta1 follow with R and S (l1_final_kernel) - no d X1, d Z1 or conv1 weight-gradient pass.
//             Weight gradients (wgrad_mfma.h) run on a second stream beside the data-gradient chain; all dense-layer
//             weight / bias gradients are one grouped launch.
// All reductions have a fixed order: the step is deterministic.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>
#include <algorithm>

#include "conv_mfma.h"
#include "gemm_mfma.h"
#include "dense_chain.h"
#include "conv5_bwd.h"
#include "scs_ctx.h"
#include "wgrad_mfma.h"
#include "train_cfg.h"
#include "conv_split_train.h"

namespace scs {

// profiling stages of scs_train_step_profiled (one per launch)
enum Stage { ST_PREP, ST_FIN1, ST_CONV12, ST_FIN2, ST_CONV3, ST_FIN3, ST_CONV4, ST_FIN4, ST_CONV5, ST_BN5, ST_D1, ST_FC1,
             ST_FC2, ST_HEAD, ST_DCAT, ST_DFEAT, ST_DH5, ST_BN5B, ST_DG5, ST_FINB4, ST_DG4, ST_FINB3, ST_DG3, ST_FINB2, ST_DG2,
             ST_L1FINAL, ST_WGDENSE, ST_WG5, ST_WG4, ST_WG3, ST_WG2, ST_WGFINAL, ST_WGFINAL2, ST_CHAIN, ST_C5B, ST_ADAM1, ST_ADAM2, ST_ADAM3, ST_C5W, ST_COUNT };
static const char* const kStageNames[ST_COUNT] = {
    "prep (weight packs, dropout masks, atlas columns) + input moments", "bn1 finalize", "conv1+conv2 fwd", "bn2 finalize",
    "conv3 fwd", "bn3 finalize", "conv4 fwd", "bn4 finalize", "conv5 fwd", "bn5 + dropout", "d1 x3 fwd", "FC1 fwd", "fc_2 fwd",
    "out + softmax + CE", "fc_2 dgrad", "FC1 dgrad", "d1 x3 dgrad", "bn5 backward", "conv5 dgrad", "bn4 bwd finalize", "conv4 dgrad",
    "bn3 bwd finalize", "conv3 dgrad", "bn2 bwd finalize", "conv2 dgrad + layer-1 sums", "layer-1 gradients", "dense wgrad (grouped)",
    "conv5 wgrad", "conv4 wgrad", "conv3 wgrad", "conv2 wgrad", "conv3..5 wgrad reduce", "conv2 wgrad reduce",
    "dense head, one launch (d1 x3, FC1, fc_2, out + softmax + CE, fc_2 / FC1 / d1 dgrad)",
    "conv5 backward, one launch (dgrad + wgrad + bn4 bwd sums)", "adam (dense layers, fused)", "adam (bn2..5 scale / shift, fused)", "adam (conv2..5 weights, fused)", "conv5 wgrad partial (side stream)"};

}  // namespace scs

struct TrainWs {       // global namespace: forward-declared by scs_ctx.h
    static constexpr int kTowers = scs::kTowers, kConvs = scs::kConvs;
    int cap = 0;                               // samples the buffers hold
    int attr_device = -1;                      // device whose kernels carry the dynamic-LDS attributes
    float* base = nullptr;
    float* Z[kTowers][kConvs]{};               // raw conv outputs of conv2..conv5 (index 1..4)
    float* ZP[kTowers]{};                      // conv2's raw output pooled by the sign of gamma ([n][20][14][14]: what conv3's forward stages; conv_mfma.h ConvArgs::pout)
    float* X[kTowers][kConvs]{};               // block outputs = conv inputs of conv3..conv5 (index 2..4)
    float* H5[kTowers]{};                      // block-5 output after dropout [n][540]
    float* dZ[kTowers][kConvs]{};              // d z of conv2..conv5 (index 1..4), left by the data-gradient kernels
    float* dX[kTowers][kConvs]{};              // gradient wrt the block outputs X[2..4] (index 2..4)
    float* dH5[kTowers]{};                     // gradient wrt H5 before the dropout mask is undone: [n][540]
    float *feat = nullptr, *featd = nullptr, *R1 = nullptr, *cat2 = nullptr, *R2 = nullptr;
    float *probs = nullptr, *dlog = nullptr, *dR2 = nullptr, *dcat = nullptr, *dfeat = nullptr;
    float *loss_rows = nullptr, *loss_dev = nullptr;
    uint8_t* masks = nullptr;                  // [3][n][540] + [n][540] + [n][540]
    uint8_t* yb = nullptr;                     // [n] labels of the batch (gathered through the index)
    float* wpk[kTowers][kConvs]{};             // unfolded forward packs (conv2..5)
    float* wdg[kTowers][kConvs]{};             // data-gradient packs (conv2..5)
    float* pack_base = nullptr;
    double* eval_acc = nullptr;                // [2]: loss sum, correct count
    float* eval_rows = nullptr;                // [chunk] per-row losses / hits of scs_eval_batch
    size_t eval_cap = 0;
    float* wT = nullptr;                       // transposed dense weights for the dense data-gradient GEMMs
    float* stat_part = nullptr;                // [3][n * bands][C][2] per-patch-band partial sums (forward statistics, then BN backward)
    float* l1_part = nullptr;                  // [3][n * 5][20][10] layer-1 sums left by the conv2 data gradient
    float* mom_part = nullptr;                 // [3][units][128] input moments per group of patches
    float* small = nullptr;                    // per-layer coefficient tables (below)
    float2* ss[kTowers][kConvs]{};             // forward scale / shift per channel (conv2..5 blocks: index 1..4)
    float4* low4[kTowers][kConvs]{};           // (sc, sh, mean, inv_std)
    float4* cf4[kTowers][kConvs]{};            // (sc, sh, ca, cb): d z = [y>0] sc dUp + ca z + cb
    float* w1p[kTowers]{};                     // conv1 MFMA operands with the batch statistics folded (kW1Floats)
    float* l1p[kTowers]{};                     // [20][12]: flipped conv1 taps (9), sc, sh, pad
    float* l1m[kTowers]{};                     // [20][2]: mean, inv_std of conv1's batch statistics
    double* mom = nullptr;                     // [3][100] R (9x9), S (column 9), count
    // option train_split (conv_split_train.h): conv1 operands of the transposed split form, raw conv2 weights as split pieces, accumulator scale
    float* w1s[kTowers]{};                     // kSp12W1Floats (the dx = 3 slots stay zero)
    uint4* w2s[kTowers]{};                     // kSp12W2Vec
    float* inv2s[kTowers]{};                   // one float
    unsigned int* split_flag = nullptr;        // set when a scaled activation left the f16 range (the step's loss is then NaN)
    bool tickets_dirty = false;                // a step failed between launches: the tickets are zeroed before the next one
    double* fin_grp = nullptr;                 // group records of the in-kernel finalize steps (inside `base`)
    unsigned long long* chain_stamps = nullptr; // diagnostic time stamps of dense_chain_kernel (SCS_CHAIN_STAMPS)
    unsigned int* chain_counter = nullptr;     // cluster barriers of dense_chain_kernel (inside `base`; zeroed by the prep launch)
    unsigned int* fin_counter = nullptr;       // arrival tickets of the in-kernel finalize steps (inside `base`; zero between launches)
    float* wg_part = nullptr;                  // [4 layers][3][kWgPartFloats] partial weight gradients
    // tail_fused (the one-launch dense head takes the conv5 block's batch norm on both sides):
    float* c5_units = nullptr;                 // [3][ceil(cap / 4)][60] float2: per-workgroup statistics units of conv5's forward kernel (inside `base`)
    float* part5 = nullptr;                    // [3][24][540] float2: (d beta, d gamma) column sums left by the head's last stage
    float* clow = nullptr;                     // [24][3][60] float4: the conv5 block's coefficients handed round inside a cluster
    // the weight-gradient kernels of a layer do not feed the data path: they run on a second stream beside the
    // data-gradient kernels (fork / join with events; the caller still sees one stream)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_head = nullptr;              // recorded behind the grouped dense weight-gradient launch: the head's gradients are final
    bool head_recorded = false;
    // scs_train_step_profiled
    bool prof = false;
    hipEvent_t pev[2 * scs::ST_COUNT]{};
    bool pused[scs::ST_COUNT]{};
};

namespace scs {

static size_t al64(size_t f) { return (f + 63) / 64 * 64; }
constexpr int kW2Floats = kFC1ld * kFC2ld;              // fc_2.W padded to [556][272] (16-byte rows for the float4 GEMM)
// Leading dimension of the transposed dense weights: 544 floats = 17 lines of 128 bytes, so that the 64-column slice a GEMM
// workgroup reads of a row is exactly two lines (with 540 it straddles three).  No measurable effect on the kernels' times.
constexpr int kTld = 544;

void train_free(scs_ctx* h) {
    if (!h->tws) return;
    TrainWs* w = h->tws;
    if (w->base) (void)hipFree(w->base);
    if (w->masks) (void)hipFree(w->masks);
    if (w->pack_base) (void)hipFree(w->pack_base);
    if (w->eval_acc) (void)hipFree(w->eval_acc);
    if (w->eval_rows) (void)hipFree(w->eval_rows);
    if (w->wT) (void)hipFree(w->wT);
    if (w->wg_part) (void)hipFree(w->wg_part);
    if (w->part5) (void)hipFree(w->part5);
    if (w->small) (void)hipFree(w->small);
    if (w->mom) (void)hipFree(w->mom);
    if (w->side) (void)hipStreamDestroy(w->side);
    if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
    if (w->ev_join) (void)hipEventDestroy(w->ev_join);
    if (w->ev_head) (void)hipEventDestroy(w->ev_head);
    for (auto& e : w->pev) if (e) (void)hipEventDestroy(e);
    delete w;
    h->tws = nullptr;
}

// Patches per input-moment unit: one per workgroup up to batch 256, never more than kMomentPpb - a unit's raw second moments
// are ONE fp32 MFMA accumulator (900 products per patch); the units are combined in double (bn1_final_kernel).
constexpr int kMomentPpb = 4;
constexpr int64_t kMaxTrainBatch = 65536;    // largest n the training entry points take (workspace: ~0.75 MB per sample)
static int moment_ppb(int n) { return std::min(kMomentPpb, (n + 255) / 256); }
static int moment_units(int n) { const int ppb = moment_ppb(n); return (n + ppb - 1) / ppb; }

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
            for (int i = 1; i < kConvs; ++i) {
                of[t][i] = off; off += al64(conv_pack_floats(i));
                og[t][i] = off; off += al64((size_t)(kCout[i] / 4) * conv_chunk_floats(kCin[i]));
            }
        size_t os2[kTowers];
        for (int t = 0; t < kTowers; ++t) { os2[t] = off; off += al64((size_t)kSp12W2Vec * 4); }
        SCS_HIP(hipMalloc(&w->pack_base, off * sizeof(float)));
        for (int t = 0; t < kTowers; ++t)
            for (int i = 1; i < kConvs; ++i) { w->wpk[t][i] = w->pack_base + of[t][i]; w->wdg[t][i] = w->pack_base + og[t][i]; }
        for (int t = 0; t < kTowers; ++t) w->w2s[t] = reinterpret_cast<uint4*>(w->pack_base + os2[t]);
        SCS_HIP(hipMalloc(&w->eval_acc, 2 * sizeof(double)));
        SCS_HIP(hipMalloc(&w->wT, (size_t)(270 * kTld + 540 * kTld + 3 * 180 * kTld + kW2Floats + kTowers * kC5WcFloats) * sizeof(float)));
        SCS_HIP(hipMemset(w->wT, 0, (size_t)(270 * kTld + 540 * kTld + 3 * 180 * kTld + kW2Floats + kTowers * kC5WcFloats) * sizeof(float)));   // pad columns stay zero
        SCS_HIP(hipMalloc(&w->wg_part, (size_t)4 * kTowers * kWgPartFloats * sizeof(float)));
        {
            const size_t f5 = (size_t)kTowers * (kChainMaxBatch / 16) * kFlat * 2, fc = (size_t)(kChainMaxBatch / 16) * kTowers * 240;
            SCS_HIP(hipMalloc(&w->part5, (f5 + fc) * sizeof(float)));
            SCS_HIP(hipMemset(w->part5, 0, (f5 + fc) * sizeof(float)));
            w->clow = w->part5 + f5;
        }
        SCS_HIP(hipMalloc(&w->mom, (size_t)kTowers * 100 * sizeof(double)));
        // coefficient tables: per tower and conv layer 64 x (float2 + float4 + float4), then the conv1 tables
        size_t so = 0;
        auto stake = [&](size_t f) { size_t o = so; so += al64(f); return o; };
        size_t o_ss[kTowers][kConvs], o_lo[kTowers][kConvs], o_cf[kTowers][kConvs], o_w1[kTowers], o_l1[kTowers], o_lm[kTowers];
        for (int t = 0; t < kTowers; ++t) {
            for (int i = 0; i < kConvs; ++i) { o_ss[t][i] = stake(128); o_lo[t][i] = stake(256); o_cf[t][i] = stake(256); }
            o_w1[t] = stake(kW1Floats + 4); o_l1[t] = stake(240); o_lm[t] = stake(40);
        }
        size_t o_w1s[kTowers], o_i2[kTowers];
        for (int t = 0; t < kTowers; ++t) { o_w1s[t] = stake(kSp12W1Floats); o_i2[t] = stake(4); }
        const size_t o_sf = stake(4);
        SCS_HIP(hipMalloc(&w->small, so * sizeof(float)));
        SCS_HIP(hipMemset(w->small, 0, so * sizeof(float)));
        for (int t = 0; t < kTowers; ++t) {
            for (int i = 0; i < kConvs; ++i) {
                w->ss[t][i] = reinterpret_cast<float2*>(w->small + o_ss[t][i]);
                w->low4[t][i] = reinterpret_cast<float4*>(w->small + o_lo[t][i]);
                w->cf4[t][i] = reinterpret_cast<float4*>(w->small + o_cf[t][i]);
            }
            w->w1p[t] = w->small + o_w1[t]; w->l1p[t] = w->small + o_l1[t]; w->l1m[t] = w->small + o_lm[t];
            w->w1s[t] = w->small + o_w1s[t]; w->inv2s[t] = w->small + o_i2[t];
        }
        w->split_flag = reinterpret_cast<unsigned int*>(w->small + o_sf);
        SCS_HIP(hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking));
        SCS_HIP(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
        SCS_HIP(hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
        SCS_HIP(hipEventCreateWithFlags(&w->ev_head, hipEventDisableTiming));
    }
    if (n <= w->cap) return SCS_OK;
    SCS_HIP(hipDeviceSynchronize());           // a previous step may still be using the old buffers
    if (w->base) { SCS_HIP(hipFree(w->base)); w->base = nullptr; }
    if (w->masks) { SCS_HIP(hipFree(w->masks)); w->masks = nullptr; }
    w->cap = 0;
    const size_t c = (size_t)((n + 63) / 64 * 64);
    size_t off = 0;
    auto take = [&](size_t f) { size_t o = off; off += al64(f); return o; };
    size_t oZ[kTowers][kConvs], oX[kTowers][kConvs], odZ[kTowers][kConvs], odX[kTowers][kConvs], oH[kTowers], odH[kTowers], oZP[kTowers];
    for (int t = 0; t < kTowers; ++t) {
        for (int i = 1; i < kConvs; ++i) {
            oZ[t][i] = take(c * kCout[i] * kHout[i] * kHout[i]);
            odZ[t][i] = take(c * kCout[i] * kHout[i] * kHout[i]);
            oX[t][i] = (i >= 2) ? take(c * kCin[i] * kHin[i] * kHin[i]) : 0;
            odX[t][i] = (i >= 2) ? take(c * kCin[i] * kHin[i] * kHin[i]) : 0;
        }
        oH[t] = take(c * kFlat);
        odH[t] = take(c * kFlat);
        oZP[t] = take(c * kCout[1] * kHin[2] * kHin[2]);
    }
    const size_t o_feat = take(c * kFlat), o_featd = take(c * kFlat), o_R1 = take(c * kFC1), o_cat2 = take(c * kFC1ld),
                 o_R2 = take(c * kFC2ld), o_probs = take(c * 16), o_dlog = take(c * 16), o_dR2 = take(c * kFC2ld),
                 o_dcat = take(c * kFC1), o_dfeat = take(c * kFlat), o_lr = take(c), o_ld = take(64),
                 o_sp = take((size_t)kTowers * c * 8 * 64 * 2),     // up to 8 bands x 64 channels x 2 sums
                 o_l1 = take((size_t)kTowers * c * kL1Bands * 200), o_mp = take((size_t)kTowers * std::max<size_t>(256, (c + kMomentPpb - 1) / kMomentPpb) * 128),
                 o_fc = take(kTowers * fin_counter_words(2 * c)),             // up to two workgroups per patch in the FIN configurations (banded tilings)
                 o_fg = take(kTowers * fin_groups(2 * c) * kFinRec * 2 * 2),  // group records (doubles)
                 o_cc = take((size_t)kChainCounterWords),
                 o_c5u = take((size_t)kTowers * ((c + 3) / 4) * 120);
    SCS_HIP(hipMalloc(&w->base, off * sizeof(float)));
    SCS_HIP(hipMemset(w->base, 0, off * sizeof(float)));
    for (int t = 0; t < kTowers; ++t) {
        for (int i = 1; i < kConvs; ++i) {
            w->Z[t][i] = w->base + oZ[t][i]; w->dZ[t][i] = w->base + odZ[t][i];
            w->X[t][i] = (i >= 2) ? w->base + oX[t][i] : nullptr; w->dX[t][i] = (i >= 2) ? w->base + odX[t][i] : nullptr;
        }
        w->H5[t] = w->base + oH[t]; w->dH5[t] = w->base + odH[t]; w->ZP[t] = w->base + oZP[t];
    }
    w->feat = w->base + o_feat; w->featd = w->base + o_featd; w->R1 = w->base + o_R1; w->cat2 = w->base + o_cat2;
    w->R2 = w->base + o_R2; w->probs = w->base + o_probs; w->dlog = w->base + o_dlog; w->dR2 = w->base + o_dR2;
    w->dcat = w->base + o_dcat; w->dfeat = w->base + o_dfeat; w->loss_rows = w->base + o_lr; w->loss_dev = w->base + o_ld;
    w->stat_part = w->base + o_sp; w->l1_part = w->base + o_l1; w->mom_part = w->base + o_mp;
    w->fin_counter = reinterpret_cast<unsigned int*>(w->base + o_fc);
    w->fin_grp = reinterpret_cast<double*>(w->base + o_fg);
    w->chain_counter = reinterpret_cast<unsigned int*>(w->base + o_cc);
    w->c5_units = w->base + o_c5u;
    SCS_HIP(hipMalloc(&w->masks, c * kFlat * 5 + c));
    w->yb = w->masks + c * kFlat * 5;
    w->cap = (int)c;
    return SCS_OK;
}

// ------------------------------------------------------------------------------------------- prep (one launch per step)
struct PackJob { const float* W; float* fwd; float* dg; int cin, cout; };
struct TrProblem { const float* in; int ldi; float* out; int ldo; int rows, cols; };
constexpr int kPackJobs = kTowers * (kConvs - 1), kTrJobs = 5;
struct PrepArgs {
    PackJob pack[kPackJobs];
    TrProblem tr[kTrJobs];
    const float* atlas; const uint8_t* y; const int32_t* index; float* cat2; uint8_t* yb; int n, n_rows;   // n_rows: rows atlas / y hold
    uint8_t* masks; size_t mask_total; uint64_t seed;       // masks == null: the caller supplied them
    const float* fc2W; float* w2p;                           // fc_2.W [555][270] -> [556][272] zero padded
    unsigned int* chain_counter;                             // cluster barrier counters of the dense chain: zeroed here
    const float* c5W; float* wc5; int c5_tower_stride;       // conv5.W [60][40][9] of tower 0 (towers c5_tower_stride apart) -> Wc [3][60][368]
    float* loss_rows;                                        // [n] row losses: NaN until the head of the step has written them
};

// blockIdx.y selects the job:
//   [0, 12)  conv2..5 operand packs: fwd Wp[c4][tap][kk][n] = W[n][4c4+kk][2-dy][2-dx] (true conv -> correlation, unfolded);
//            dg  Kp[o4][tap][kk][c] = W[4o4+kk][c][a][b] (tap = 3a+b, no flip, channels transposed)
//   [12, 17) dense weights transposed for the data-gradient GEMMs
//   17       atlas -> columns 540..555 of the fc_2 input rows (ConcatLayer elem_channels2, nets.py:224) + batch labels
//   18       counter-based dropout keep-masks: one hash per element (splitmix64 finaliser), keep when the top bit is set
constexpr int kPrepJobs = kPackJobs + kTrJobs + 2, kPrepBlocks = 64;      // kPrepBlocks workgroups per job ...
// ... and kPrepExtra * kPrepBlocks more for the one job that was the launch's long pole: the 540 x 540 transpose of FC1.W is 289
// tiles - 4.5 dependent (load, barrier, store) rounds per workgroup at 64 (10 us of prep against 6.6 us without that job)
constexpr int kPrepBigJob = kPackJobs + 1, kPrepExtra = 3;
constexpr int kPrepWorkgroups = (kPrepJobs + kPrepExtra) * kPrepBlocks;
__device__ __forceinline__ void prep_body(const PrepArgs& a, int job, int bx) {
    __shared__ float tile[32][33];
    const int stride = kPrepBlocks * 256, t0 = bx * 256 + threadIdx.x;
    if (job < kPackJobs) {
        const PackJob J = a.pack[job];
        const int tot = (J.cin / 4) * conv_chunk_floats(J.cout);
        for (int e = t0; e < tot; e += stride) {
            const PackIdx ix = conv_pack_index(e, J.cout);
            float v = 0.f;
            if (ix.n >= 0) v = J.W[((ix.n * J.cin + ix.c4 * 4 + ix.kk) * 3 + (2 - ix.tap / 3)) * 3 + (2 - ix.tap % 3)];
            J.fwd[e] = v;
        }
        const int totd = (J.cout / 4) * conv_chunk_floats(J.cin);
        for (int e = t0; e < totd; e += stride) {
            const PackIdx ix = conv_pack_index(e, J.cin);
            float v = 0.f;
            if (ix.n >= 0) v = J.W[(((ix.c4 * 4 + ix.kk) * J.cin + ix.n) * 3 + ix.tap / 3) * 3 + ix.tap % 3];
            J.dg[e] = v;
        }
    } else if (job < kPackJobs + kTrJobs) {
        const TrProblem P = a.tr[job - kPackJobs];           // out[c][r] = in[r][c]
        const int tr_ = (P.rows + 31) / 32, tc_ = (P.cols + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int tl = bx; tl < tr_ * tc_; tl += (job == kPrepBigJob ? (1 + kPrepExtra) * kPrepBlocks : kPrepBlocks)) {
            const int r0 = (tl / tc_) * 32, c0 = (tl % tc_) * 32;
            __syncthreads();
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
    } else if (job == kPackJobs + kTrJobs) {
        for (int e = t0; e < a.n * 16; e += stride) {
            const int row = e >> 4, c = e & 15;
            const int src = max(0, min(a.index ? a.index[row] : row, a.n_rows - 1));
            a.cat2[(size_t)row * kFC1ld + kFC1 + c] = c < 15 ? a.atlas[(size_t)src * 15 + c] : 0.f;
            if (c == 0) a.yb[row] = a.y[src];
        }
        for (int e = t0; e < kChainCounterWords; e += stride) a.chain_counter[e] = 0u;
        // a row block the dense head never reaches (dense_chain.h: an XCD that did not receive its workgroups) leaves NaN in
        // the step's loss instead of last step's value
        for (int e = t0; e < a.n; e += stride) a.loss_rows[e] = __builtin_nanf("");
        for (int e = t0; e < kTowers * 60 * 360; e += stride) {           // Wc[t][o][c*9 + tap] = W[t][o][c][8 - tap]
            const int t = e / 21600, f = e - t * 21600, o = f / 360, rem = f - o * 360, c = rem / 9, k = rem - c * 9;
            a.wc5[(size_t)t * kC5WcFloats + o * kC5WS + c * 9 + (8 - k)] = a.c5W[(size_t)t * a.c5_tower_stride + f];
        }
        for (int e = t0; e < kW2Floats; e += stride) {
            const int k = e / kFC2ld, c = e - k * kFC2ld;
            a.w2p[e] = (k < kFC1 + kAtlas && c < kFC2) ? a.fc2W[(size_t)k * kFC2 + c] : 0.f;
        }
    } else if (a.masks) {
        for (size_t e = t0; e < a.mask_total; e += stride) {
            uint64_t z = a.seed + 0x9E3779B97F4A7C15ull * (e + 1);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            a.masks[e] = (uint8_t)(z >> 63);
        }
    }
}

// ------------------------------------------------------------------------------------------- small kernels
struct Ptr3 { float* p[kTowers]; };
struct CPtr3 { const float* p[kTowers]; };
template <class T> __device__ __forceinline__ T sel3(const T (&a)[kTowers], int t) { return t == 0 ? a[0] : (t == 1 ? a[1] : a[2]); }

// lasagne.updates.adam on ONE element, for the kernels that apply the update to the gradients they have just written (the fused
// form of the step, scs_set_fused_adam): adam_kernel's arithmetic with grad_scale 1
struct AdamCoef { float a_t, b1, b2, eps; };
struct AdamFuse { float* p; float* m; float* v; AdamCoef k; };       // p == null: not fused
// (explicitly rounded operations: every kernel that applies the update - adam_kernel, adam_seg_kernel, the fused reductions -
// gives the same bits whatever the compiler would contract around the inlined text)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float gr, const AdamCoef& k) {
    const float mn = __fmaf_rn(k.b1, m, __fmul_rn(1.f - k.b1, gr));
    const float vn = __fmaf_rn(k.b2, v, __fmul_rn(__fmul_rn(1.f - k.b2, gr), gr));
    m = mn; v = vn;
    p = __fsub_rn(p, __fdiv_rn(__fmul_rn(k.a_t, mn), __fadd_rn(__fsqrt_rn(vn), k.eps)));
}
__device__ __forceinline__ void adam_elem(const AdamFuse& a, int at, float g) { adam_update(a.p[at], a.m[at], a.v[at], g, a.k); }

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

// three block sums with ONE exchange through LDS (fixed order: lanes by shuffle, then waves 0..3)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double (*s_red)[3]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); c += __shfl_down(c, o); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_red[wv][0] = a; s_red[wv][1] = b; s_red[wv][2] = c; }
    __syncthreads();
    a = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
    b = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
    c = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
}

// Second moments of the 32x32 inputs over the 30x30 conv1 output grid: with P[px][k] = x0[px + tap k] (k < 9) and
// P[px][9] = 1, D = P'P holds R[k][k'] = sum_px x0[px+k] x0[px+k'], S[k] = D[k][9] and the pixel count D[9][9].  One
// matrix product per patch (both MFMA operands are the same register), `ppb` patches per workgroup, the four waves split
// the rows.  Everything conv1's batch statistics and weight gradient need from the inputs (see bn1_final / l1_final).
// what turns the input moments into conv1's batch statistics and folded operands (bn1_final_body)
struct Bn1Args {
    float* params; int w_off0, bn_off0, tower_stride;
    double* mom;                  // [3][100] out: R, S, count (l1_final_kernel reads them)
    Ptr3 w1p, l1p, l1m;
    unsigned int* counter;        // in-launch finalize: arrival tickets (FinArgs layout, fin_counter_words(units) words per tower), or null
    double* rec;                  // in-launch finalize: group records [tower][groups][128]
    Ptr3 w1s;                     // option train_split: the same folded taps as [slot 4 dy + dx][20] + shift [20], times kSplitActScale (or null)
};

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// conv1's batch statistics from the input moments (z = sum_k wf[k] x0[. + k], wf = flipped filter):
//   mean[c] = wf.S / M,  var[c] = wf' (R/M - S S'/M^2) wf,  inv_std = 1/sqrt(var + eps)      (all in double)
// then everything later kernels need: the MFMA operands of the fused conv1 with (gamma inv_std, beta - mean gamma inv_std)
// folded (conv_mfma.h: conv1_to_lds), the plain tap table for the backward pass, and the running averages (alpha 0.1).
// The arithmetic, whatever runs it: units are summed in GROUPS of kFinGroup (ascending, in double), moment e of the 100 is the
// sum of eight strided sub-sums (groups = h mod 8, ascending), added in order h = 0..7.  Run by NT threads (a multiple of 128)
// for tower t.  REC: the group sums are records left inside this launch by the workgroups that completed the groups
// (write-through stores behind tickets: loads that no L1 can serve); otherwise they are formed here from the unit sums.
template <bool SC1>
__device__ __forceinline__ double mom_group_sum(const float* __restrict__ p /* unit 0 of this moment */, int u0, int k) {
    float r[kFinGroup];
#pragma unroll
    for (int i = 0; i < kFinGroup; ++i) { const float* q = p + (size_t)(u0 + min(i, k - 1)) * 128; r[i] = SC1 ? ld_sc1(q) : *q; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < kFinGroup; ++i) if (i < k) s += r[i];
    return s;
}
template <int NT, bool REC>
__device__ __forceinline__ void bn1_final_body(const float* __restrict__ part, const double* __restrict__ rec /* [t][groups][128], REC */, int units, int n,
                                               const Bn1Args& b, int t, double (*s_p)[128], double* s_R, double* s_Q) {
    static_assert(NT % 128 == 0 && 8 % (NT / 128) == 0, "thread rows of 128");
    constexpr int HN = NT / 128, HP = 8 / HN;               // thread rows, sub-sums per thread
    const int e = threadIdx.x & 127, h0 = threadIdx.x >> 7;
    // the channel threads' filter taps and BN entries: requested first, in flight with the unit sums
    const int c = threadIdx.x;
    float* bn = b.params + b.bn_off0 + t * b.tower_stride;  // beta, gamma, mean, inv_std
    // (the three per-tower tables are picked HERE: with the picks between the rolled loops below, hipcc 7.2 left the third one's
    // register pair unset on the t == 2 path - a store through a stale pointer)
    float* const w1 = sel3(b.w1p.p, t);
    float* const l1 = sel3(b.l1p.p, t) + c * 12;
    float* const lm = sel3(b.l1m.p, t);
    float* const w1s = sel3(b.w1s.p, t);
    float wr[9], b_beta = 0.f, b_gamma = 0.f, b_mean = 0.f, b_inv = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = 0.f;
    if (c < 20) {
        const float* W = b.params + b.w_off0 + t * b.tower_stride + c * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[k] = W[k];
        b_beta = bn[c]; b_gamma = bn[20 + c]; b_mean = bn[40 + c]; b_inv = bn[60 + c];
    }
    const double M = (double)n * 900.0;
    double s[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) s[i] = 0;
    if (e < 100) {
        const int G = (units + kFinGroup - 1) / kFinGroup;
        if constexpr (REC) {
            // up to 4 groups per sub-sum in one round trip (G <= 32: the batches the in-launch form takes), then one by one
            const double* __restrict__ q = rec + (size_t)t * G * 128 + e;
            double r[HP][4];
#pragma unroll
            for (int i = 0; i < HP; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r[i][j] = ld_sc1(q + (size_t)min(h0 + HN * i + 8 * j, G - 1) * 128);
#pragma unroll
            for (int i = 0; i < HP; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (h0 + HN * i + 8 * j < G) s[i] += r[i][j];
                for (int g = h0 + HN * i + 32; g < G; g += 8) s[i] += ld_sc1(q + (size_t)g * 128);
            }
        } else {
            const float* __restrict__ p = part + (size_t)t * units * 128 + e;
#pragma unroll
            for (int i = 0; i < HP; ++i) {
                int g = h0 + HN * i;
                for (; g + 8 < G; g += 16) {                // two groups (32 unit sums) in flight
                    const double s0 = mom_group_sum<false>(p, kFinGroup * g, kFinGroup);
                    const double s1 = mom_group_sum<false>(p, kFinGroup * (g + 8), min(kFinGroup, units - kFinGroup * (g + 8)));
                    s[i] += s0; s[i] += s1;
                }
                if (g < G) s[i] += mom_group_sum<false>(p, kFinGroup * g, min(kFinGroup, units - kFinGroup * g));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < HP; ++i) s_p[h0 + HN * i][e] = s[i];
    __syncthreads();
    if (threadIdx.x < 100) {
        double v = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += s_p[q][threadIdx.x];
        s_R[threadIdx.x] = v;
        s_Q[threadIdx.x] = v / M;                           // the quotients of the variance formula, one division each
        b.mom[t * 100 + threadIdx.x] = v;
    }
    __syncthreads();
    if (c >= 32 && c < 32 + 48) {                           // rows 9..11 of the operand tables are zero
        const int z = c - 32;
        if (z < 48) w1[144 + z] = 0.f;                      // main [12][16]: rows 9..11 = elements 144..191
        if (z < 12) w1[192 + 36 + z] = 0.f;                 // leftover [12][4]: rows 9..11 = elements 228..239
    }
    if (c >= 20) return;
    // (rolled loops over LDS operands: inside the moments launch this body must not raise the kernel's register count)
    double* wf = &s_p[0][0] + c * 9;                        // s_p has been consumed (barrier above): this thread's flipped taps
#pragma unroll
    for (int k = 0; k < 9; ++k) wf[k] = (double)wr[(2 - k / 3) * 3 + (2 - k % 3)];
    double mean = 0, var = 0;
#pragma unroll 1
    for (int k = 0; k < 9; ++k) mean += wf[k] * s_R[k * 10 + 9];
    mean /= M;
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
        const double wk = wf[k], qk = s_Q[k * 10 + 9];
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) var += wk * wf[k2] * (s_Q[k * 10 + k2] - qk * s_Q[k2 * 10 + 9]);
    }
    const double inv = 1.0 / sqrt(fmax(var, 0.0) + (double)kBnEps);
    const float sc = b_gamma * (float)inv, sh = b_beta - (float)mean * sc;
    bn[40 + c] = (1.f - kBnAlpha) * b_mean + kBnAlpha * (float)mean;
    bn[60 + c] = (1.f - kBnAlpha) * b_inv + kBnAlpha * (float)inv;
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
        l1[k] = (float)wf[k];
        const float v = (float)wf[k] * sc;
        if (c < 16) w1[k * 16 + c] = v; else w1[192 + k * 4 + (c - 16)] = v;
        if (w1s) w1s[(4 * (k / 3) + k % 3) * 20 + c] = v * kSplitActScale;
    }
    l1[9] = sc; l1[10] = sh; l1[11] = 0.f;
    w1[240 + c] = sh;
    if (w1s) w1s[240 + c] = sh * kSplitActScale;
    lm[2 * c] = (float)mean; lm[2 * c + 1] = (float)inv;
}

// With b.counter set, the workgroup that completes a tower's arrival tickets (two levels, as the conv kernels' finalize:
// conv_mfma.h take_ticket) runs bn1_final_body in this launch.
__device__ __forceinline__ void moments_body(const X0Src& x0, float* __restrict__ part /*[t][units][128]*/, int n, int ppb, int units, int t,
                                             int bx, const Bn1Args& b) {
    // one buffer: the patch rows and the waves' partial products, then (the finalizing workgroup, behind its tickets' barriers)
    // the finalize's sub-sums over the same bytes
    __shared__ __attribute__((aligned(16))) double s_buf[8 * 128 + 200];
    float* s_x = reinterpret_cast<float*>(s_buf);                           // [1024 + 8]
    float (*s_red)[256] = reinterpret_cast<float (*)[256]>(s_x + 1032);     // [4][256]
    double (*s_p)[128] = reinterpret_cast<double (*)[128]>(s_buf);          // [8][128]
    double* s_R = s_buf + 8 * 128; double* s_Q = s_R + 100;
    __shared__ int s_flag;
    static_assert((1032 + 4 * 256) * sizeof(float) <= sizeof(s_buf), "moments scratch");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int toff = lr < 9 ? (lr / 3) * 32 + lr % 3 : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int g0 = bx * ppb, g1 = min(n, g0 + ppb);
    for (int g = g0; g < g1; ++g) {
        __syncthreads();
        load_x0_rows<32, 256>(x0, t, g, 0, s_x, tid);
        __syncthreads();
        for (int r = wave; r < 30; r += 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = 4 * j + kq;
                const float v = x < 30 ? (lr < 9 ? s_x[r * 32 + x + toff] : (lr == 9 ? 1.f : 0.f)) : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s_red[wave][(4 * kq + i) * 16 + lr] = acc[i];
    __syncthreads();
    const int m = tid >> 4, k = tid & 15;
    if (m < 10 && k < 10) {
        float* dst = part + ((size_t)t * units + bx) * 128 + m * 10 + k;
        const float v = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        if (b.counter) st_sc1(dst, v); else *dst = v;
    }
    if (b.counter) {
        const unsigned int nwg = (unsigned int)units, grp = bx / kFinGroup, groups = (nwg + kFinGroup - 1) / kFinGroup;
        unsigned int* cnt = b.counter + t * fin_counter_words(nwg);
        if (take_ticket(cnt + (size_t)kFinLine * (1 + grp), min((unsigned int)kFinGroup, nwg - grp * kFinGroup), &s_flag)) {
            // this workgroup completed its group: the group's unit sums -> one record per moment
            if (tid < 100)
                st_sc1(b.rec + ((size_t)t * groups + grp) * 128 + tid,
                       mom_group_sum<true>(part + (size_t)t * units * 128 + tid, kFinGroup * (int)grp, min(kFinGroup, units - kFinGroup * (int)grp)));
            if (take_ticket(cnt, groups, &s_flag)) bn1_final_body<256, true>(part, b.rec, units, n, b, t, s_p, s_R, s_Q);
        }
    }
}

// First launch of a step, a flat grid: workgroups [0, kTowers * units) are the input moments of the towers - dispatched FIRST, so
// that the in-launch finalize of conv1's statistics (small batches) runs beside the prep jobs instead of behind them - and the
// next kPrepWorkgroups the prep jobs (nothing in either half depends on the other; one launch instead of two on the
// step's critical path.  The prep jobs on the side stream beside the moments was tried: the fork / join events cost more than
// the 8 us the jobs take - 0.628 against 0.616 ms per step, profiles/r03_notes.txt).
struct SplitPackJobs { const float* W[kTowers]; uint4* wp[kTowers]; float* inv[kTowers]; int cin, cout; unsigned int* flag; };   // option train_split: one workgroup per job, behind the prep jobs
__global__ __launch_bounds__(256) void prep_moments_kernel(PrepArgs a, X0Src x0, float* __restrict__ mom_part, int n, int ppb, int units, Bn1Args b,
                                                           SplitPackJobs sj) {
    const int bx = blockIdx.x - kTowers * units;
    if (bx >= kPrepWorkgroups) {
        __shared__ float s_red[256];
        const int t = bx - kPrepWorkgroups;
        if (t == 0 && threadIdx.x == 0) *sj.flag = 0u;          // this step's overflow flag (the loss-mean workgroup reads it)
        pack_split_raw(sel3(sj.W, t), sj.cin, sj.cout, sel3(sj.wp, t), sel3(sj.inv, t), s_red);
    } else if (bx >= kPrepJobs * kPrepBlocks) prep_body(a, kPrepBigJob, bx - (kPrepJobs - 1) * kPrepBlocks);   // the big job's extra workgroups: blocks 64 ..
    else if (bx >= 0) prep_body(a, bx / kPrepBlocks, bx % kPrepBlocks);
    else moments_body(x0, mom_part, n, ppb, units, (int)blockIdx.x / units, (int)blockIdx.x % units, b);
}
inline dim3 prep_moments_grid(int units, int split_jobs = 0) { return dim3(kPrepWorkgroups + kTowers * units + split_jobs); }

// The finalize as a launch of its own (batches beyond kBn1FusedMaxBatch): one workgroup per tower.
__global__ __launch_bounds__(1024) void bn1_final_kernel(const float* __restrict__ part, int units, int n, Bn1Args b) {
    __shared__ double s_p[8][128];
    __shared__ double s_R[100], s_Q[100];
    bn1_final_body<1024, false>(part, nullptr, units, n, b, blockIdx.x, s_p, s_R, s_Q);
}

// Batch statistics of a conv layer from the per-(patch band, channel) units its epilogue left - each the mean and the sum of
// squared deviations of `per_unit` values - combined by the parallel-variance formula in double, in a fixed order:
//   mean = avg(mean_u),  M2 = sum M2_u + per_unit * sum (mean_u - mean)^2,  var = M2 / count          (biased, as Lasagne's)
// one workgroup per (channel, tower).  Leaves scale / shift for the next conv's staging, (sc, sh, mean, inv_std) for the
// backward pass, and moves the running averages (Lasagne BatchNormLayer, alpha 0.1, nets.py:172-177).
// APPLY5 (conv5 block): also applies BN + ReLU + the dropout in front of d1 to the channel's n x 9 values -> H5.
template <bool APPLY5>
__global__ __launch_bounds__(256) void bn_fwd_final_kernel(const float* __restrict__ part, int units, double count, int C, float* params,
                                                          int bn_off0, int tower_stride, Ptr3 ss, Ptr3 low4, CPtr3 Z, Ptr3 H5,
                                                          const uint8_t* __restrict__ mask, int n) {
    __shared__ double s_red[4][3];
    __shared__ float s_ss[2];
    const int c = blockIdx.x, t = blockIdx.y;
    const float2* __restrict__ p = reinterpret_cast<const float2*>(part) + (size_t)t * units * C + c;
    float* bn = params + bn_off0 + t * tower_stride;               // beta, gamma, mean, inv_std
    float b_beta = 0.f, b_gamma = 0.f, b_mean = 0.f, b_inv = 0.f;
    if (threadIdx.x == 0) { b_beta = bn[c]; b_gamma = bn[C + c]; b_mean = bn[2 * C + c]; b_inv = bn[3 * C + c]; }   // in flight with the partials
    // APPLY5: the first nine values (and keep-flags) this thread will transform do not depend on the statistics: requested now
    float zv0[9];
    uint8_t mk0[9];
    if constexpr (APPLY5) {
        const size_t at = ((size_t)min((int)threadIdx.x, n - 1) * C + c) * 9;
        const float* __restrict__ z = sel3(Z.p, t);
#pragma unroll
        for (int e = 0; e < 9; ++e) { zv0[e] = z[at + e]; mk0[e] = mask[(size_t)t * n * kFlat + at + e]; }
    }
    // one pass: sum of unit means, of their squares (about the first unit's mean: exact in double) and of the unit M2
    const double m0 = (double)p[0].x;
    double s = 0, s2 = 0, q = 0;
    for (int u0 = threadIdx.x; u0 < units; u0 += 8 * 256) {        // eight units per thread in flight (conv2 at batch 256: 7 per thread, one round trip)
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)min(u0 + 256 * j, units - 1) * C];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (u0 + 256 * j < units) { const double d = (double)v[j].x - m0; s += d; s2 += d * d; q += v[j].y; }
    }
    block_sum3(s, s2, q, s_red);
    if (threadIdx.x == 0) {
        const double dm = s / units, mean = m0 + dm;
        const double dev2 = fmax(s2 - s * dm, 0.0);                // sum (mean_u - mean)^2
        const double var = (q + (count / units) * dev2) / count;
        const double inv = 1.0 / sqrt(var + (double)kBnEps);
        const float sc = b_gamma * (float)inv, sh = b_beta - (float)mean * sc;
        reinterpret_cast<float2*>(sel3(ss.p, t))[c] = make_float2(sc, sh);
        reinterpret_cast<float4*>(sel3(low4.p, t))[c] = make_float4(sc, sh, (float)mean, (float)inv);
        bn[2 * C + c] = (1.f - kBnAlpha) * b_mean + kBnAlpha * (float)mean;
        bn[3 * C + c] = (1.f - kBnAlpha) * b_inv + kBnAlpha * (float)inv;
        s_ss[0] = sc; s_ss[1] = sh;
    }
    if constexpr (APPLY5) {
        __syncthreads();
        const float sc = s_ss[0], sh = s_ss[1];
        const float* __restrict__ z = sel3(Z.p, t);
        float* __restrict__ o = sel3(H5.p, t);
        for (int ni = threadIdx.x; ni < n; ni += 256) {
            const size_t at = ((size_t)ni * C + c) * 9;
            float zv[9];
            uint8_t mk[9];
            if (ni == (int)threadIdx.x) {
#pragma unroll
                for (int e = 0; e < 9; ++e) { zv[e] = zv0[e]; mk[e] = mk0[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 9; ++e) { zv[e] = z[at + e]; mk[e] = mask[(size_t)t * n * kFlat + at + e]; }
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const float v = fmaxf(fmaf(zv[e], sc, sh), 0.f);
                o[at + e] = mk[e] ? v * 2.f : 0.f;                   // dropout 0.5 in front of d1 (nets.py:178)
            }
        }
    }
}

// BN backward of a conv layer from the (d beta, d gamma) partials:  d z = sc (dy - d beta/M - xhat d gamma/M), written as
// d z = [y > 0] sc dUp + ca z + cb with ca = -sc inv mg, cb = sc (inv mean mg - mb), mb = d beta/M, mg = d gamma/M.
// DIRECT5 (conv5 block, 3x3 planes): no partials - the workgroup reads the channel's z and upstream gradient itself.
template <bool DIRECT5>
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int units, double count, int C, CPtr3 low4, Ptr3 cf4,
                                                          float* grads, int bn_off0, int tower_stride, CPtr3 Z, CPtr3 dUp, int n) {
    __shared__ double s_red[4][3];
    const int c = blockIdx.x, t = blockIdx.y;
    const float4 lo = reinterpret_cast<const float4*>(sel3(low4.p, t))[c];      // sc, sh, mean, inv
    double sb = 0, sg = 0;
    if constexpr (DIRECT5) {
        const float* __restrict__ z = sel3(Z.p, t);
        const float* __restrict__ d = sel3(dUp.p, t);
        float fb = 0.f, fg = 0.f;
        for (int ni = threadIdx.x; ni < n; ni += 256) {
            const size_t at = ((size_t)ni * C + c) * 9;
            float zv[9], dv[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) { zv[e] = z[at + e]; dv[e] = d[at + e]; }
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const float dy = fmaf(zv[e], lo.x, lo.y) > 0.f ? dv[e] : 0.f;
                fb += dy; fg = fmaf(dy, (zv[e] - lo.z) * lo.w, fg);
            }
        }
        sb = fb; sg = fg;
    } else {
        const float2* __restrict__ p = reinterpret_cast<const float2*>(part) + (size_t)t * units * C + c;
        for (int u0 = threadIdx.x; u0 < units; u0 += 8 * 256) {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(size_t)min(u0 + 256 * j, units - 1) * C];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (u0 + 256 * j < units) { sb += v[j].x; sg += v[j].y; }
        }
    }
    double unused = 0;
    block_sum3(sb, sg, unused, s_red);
    if (threadIdx.x == 0) {
        float* gb = grads + bn_off0 + t * tower_stride;
        gb[c] = (float)sb; gb[C + c] = (float)sg;                                   // mean / inv_std entries stay 0
        const double mb = sb / count, mg = sg / count;
        const double sc = lo.x, inv = lo.w, mean = lo.z;
        reinterpret_cast<float4*>(sel3(cf4.p, t))[c] = make_float4(lo.x, lo.y, (float)(-sc * inv * mg), (float)(sc * (inv * mean * mg - mb)));
    }
}

// Gradients of the conv1 block from the sums the conv2 data-gradient kernel left (A[c][k] = sum dy x0[. + k], sum dy) and the
// input moments (R, S): with xhat = (z - mean) inv, z = wf . x0,
//   d beta = sum dy,  d gamma = inv (wf . A[c] - mean d beta),
//   d wf[k] = sc (A[c][k] - mb S[k] - mg inv (sum_k' wf[k'] R[k'][k] - mean S[k])),   mb = d beta/M, mg = d gamma/M,
// and d W = d wf flipped back to Lasagne's true-convolution layout.  One workgroup per (channel, tower).
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int units, int n, const double* __restrict__ mom, CPtr3 l1p,
                                                      CPtr3 l1m, float* grads, int w_off0, int bn_off0, int tower_stride, AdamFuse af) {
    __shared__ double s_w[4][10];
    __shared__ double s_A[10];
    const int c = blockIdx.x, t = blockIdx.y;
    const float2* __restrict__ p = reinterpret_cast<const float2*>(part + (size_t)t * units * 200 + c * 10);
    double a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = 0;
    for (int u0 = threadIdx.x; u0 < units; u0 += 4 * 256) {
        float2 v[4][5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = min(u0 + 256 * q, units - 1);
#pragma unroll
            for (int k = 0; k < 5; ++k) v[q][k] = p[(size_t)u * 100 + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (u0 + 256 * q < units) {
#pragma unroll
                for (int k = 0; k < 5; ++k) { a[2 * k] += v[q][k].x; a[2 * k + 1] += v[q][k].y; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_down(a[k], o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s_w[threadIdx.x >> 6][k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 10) s_A[threadIdx.x] = (s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + (s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float* l1 = sel3(l1p.p, t) + c * 12;
    const float* lm = sel3(l1m.p, t);
    const double* R = mom + t * 100;
    const double M = (double)n * 900.0, mean = lm[2 * c], inv = lm[2 * c + 1], sc = l1[9];
    const double db = s_A[9];
    double wa = 0;
    for (int k = 0; k < 9; ++k) wa += (double)l1[k] * s_A[k];
    const double dg = inv * (wa - mean * db);
    float* gb = grads + bn_off0 + t * tower_stride;
    gb[c] = (float)db; gb[20 + c] = (float)dg;
    if (af.p) { adam_elem(af, bn_off0 + t * tower_stride + c, (float)db); adam_elem(af, bn_off0 + t * tower_stride + 20 + c, (float)dg); }
    const double mb = db / M, mg = dg / M;
    float* gw = grads + w_off0 + t * tower_stride + c * 9;
    for (int k = 0; k < 9; ++k) {
        double wr = 0;
        for (int k2 = 0; k2 < 9; ++k2) wr += (double)l1[k2] * R[k2 * 10 + k];
        const double xs = inv * (wr - mean * R[k * 10 + 9]);
        const float gv = (float)(sc * (s_A[k] - mb * R[k * 10 + 9] - mg * xs));
        const int kk = (2 - k / 3) * 3 + (2 - k % 3);
        gw[kk] = gv;
        if (af.p) adam_elem(af, w_off0 + t * tower_stride + c * 9 + kk, gv);
    }
}

// out_layer + softmax + cross-entropy on unpacked arena weights W[270][15], b[15]; per-row loss and hit flag
// (scs_eval_batch adds them in a fixed order)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ R2, const float* __restrict__ W,
                                                        const float* __restrict__ b, const uint8_t* __restrict__ y, const int32_t* __restrict__ index,
                                                        int n_rows, int n, float* __restrict__ rows /*[n][2]*/) {
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
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) s += expf(z[c] - mx);
    const int yi = y[index ? max(0, min(index[row], n_rows - 1)) : row];
    int best = 0; float bz = z[0], zy = 0.f;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        if (z[c] > bz) { bz = z[c]; best = c; }
        if (c == yi) zy = z[c];
    }
    rows[2 * row] = -(zy - mx - logf(s));                          // -log softmax, stable
    rows[2 * row + 1] = best == yi ? 1.f : 0.f;
}

// acc[0] += sum of losses, acc[1] += hits over `n` rows: one workgroup, double, fixed order
__global__ __launch_bounds__(256) void eval_sum_kernel(const float* __restrict__ rows, int n, double* acc) {
    __shared__ double s_red[4];
    double l = 0, hct = 0;
    for (int e = threadIdx.x; e < n; e += 256) { l += rows[2 * e]; hct += rows[2 * e + 1]; }
    l = block_sum(l, s_red);
    hct = block_sum(hct, s_red);
    if (threadIdx.x == 0) { acc[0] += l; acc[1] += hct; }
}

// Training head, 16 rows per workgroup, both small products on the matrix cores:
//   logits[16][15] = R2[16][270] W[270][15] + b      (the four waves split K, partial tiles added through LDS in wave order)
//   softmax + cross-entropy per row, d logits = (p - onehot)/n                                  (nets.py:231, 235)
//   dR2[16][270]  = (R2 > 0) * d logits[16][15] W'[15][270]   - the gradient through fc_2's ReLU (the waves split the columns)
// (The batch-mean loss is one ordered sum of loss_rows inside the grouped dense weight-gradient launch, gemm_mfma.h.)
__global__ __launch_bounds__(256) void head_train_kernel(const float* __restrict__ R2, const float* __restrict__ W,
                                                        const float* __restrict__ b, const uint8_t* __restrict__ y, int n,
                                                        float* __restrict__ probs /*[n][16]*/, float* __restrict__ dlog /*[n][16]*/,
                                                        float* __restrict__ loss_rows /*[n]*/, float* __restrict__ dR2 /*[n][kFC2ld]*/) {
    __shared__ f32x4 s_p[4][64];
    __shared__ float s_d[16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 16;
    // ---- logits: wave w owns k-steps [17 w, 17 w + 17) of the 68 (K = 270 padded to 272)
    {
        constexpr int SPW = 17;
        const int arow = min(row0 + lr, n - 1);
        const float* __restrict__ ap = R2 + (size_t)arow * kFC2ld + kq;
        float av[SPW], bv[SPW];
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
            const int k = 4 * (wave * SPW + i) + kq;
            av[i] = k < kFC2 ? ap[4 * (wave * SPW + i)] : 0.f;
            bv[i] = (k < kFC2 && lr < 15) ? W[k * 15 + lr] : 0.f;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SPW; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
        s_p[wave][lane] = acc;                                       // D[row 4 kq + r][class lr]
    }
    __syncthreads();
    // ---- softmax / cross-entropy: thread (r, c) = (row, class); its logit sits in lane (c, r >> 2), register r & 3
    const int r = tid >> 4, c = tid & 15;
    const int row = row0 + r;
    const int src = (r >> 2) * 16 + c;
    const float z = (c < 15 ? b[c] : 0.f) + ((s_p[0][src][r & 3] + s_p[1][src][r & 3]) + (s_p[2][src][r & 3] + s_p[3][src][r & 3]));
    float mx = c < 15 ? z : -3.0e38f;
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
    // ---- dR2: 17 column tiles of 16, wave w owns tiles w, w + 4, ...; K = the 15 classes (4 k-steps)
    {
        float a4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a4[s4] = s_d[lr][4 * s4 + kq];             // d logits[row lr][class 4 s + kq] (class 15 = 0)
        for (int t = wave; t < 17; t += 4) {
            const int col = 16 * t + lr;
            float w4[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { const int cl = 4 * s4 + kq; w4[s4] = (col < kFC2 && cl < 15) ? W[col * 15 + cl] : 0.f; }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], w4[s4], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int orow = row0 + 4 * kq + i;
                if (orow < n) {
                    const float gate = col < kFC2 ? R2[(size_t)orow * kFC2ld + col] : 0.f;
                    dR2[(size_t)orow * kFC2ld + col] = gate > 0.f ? acc[i] : 0.f;          // columns 270, 271: zero
                }
            }
        }
    }
}

// sum of the partial weight gradients of ALL conv layers in a fixed order (deterministic): blockIdx.z = layer, 64 elements x
// 4 groups of partials per block, 8 independent loads in flight per thread
struct WgFinalJob { const float* part; int nsplit, cin, cout, w_off0; };
struct WgFinalArgs { WgFinalJob j[4]; float* grads; int tower_stride; int z0;          // blockIdx.z + z0 selects the job
                     AdamFuse af; };                                                   // af.p set: the update of the element, fused
__global__ __launch_bounds__(256) void wgrad_final_kernel(WgFinalArgs a) {
    __shared__ float s_red[4][64];
    WgFinalJob J = a.j[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) if ((int)blockIdx.z + a.z0 == i) J = a.j[i];
    const int t = blockIdx.y, tot = J.cout * J.cin * 9;
    if ((int)blockIdx.x * 64 >= tot) return;
    const int ex = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + ex;
    const int per = (J.nsplit + 3) / 4, sp0 = grp * per, sp1 = min(J.nsplit, sp0 + per);
    float v = 0.f;
    if (e < tot) {
        const float* __restrict__ base = J.part + (size_t)t * J.nsplit * tot + e;
        int sp = sp0;
#ifndef SCS_WGFIN_BATCH
#define SCS_WGFIN_BATCH 32
#endif
        // (the partials of a thread in flight SCS_WGFIN_BATCH at a time - the launch is a chain of dependent round trips and the last
        // one of the step; added in the same order whatever the batch: same bits)
        for (; sp + SCS_WGFIN_BATCH <= sp1; sp += SCS_WGFIN_BATCH) {
            float r[SCS_WGFIN_BATCH];
#pragma unroll
            for (int u = 0; u < SCS_WGFIN_BATCH; ++u) r[u] = base[(size_t)(sp + u) * tot];
#pragma unroll
            for (int u = 0; u < SCS_WGFIN_BATCH; ++u) v += r[u];
        }
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
        const int o = e % J.cout, j = e / J.cout, c = j / 9, k = j - c * 9;    // partials are [c*9 + tap][o]
        const int at = J.w_off0 + t * a.tower_stride + (o * J.cin + c) * 9 + (2 - k / 3) * 3 + (2 - k % 3);   // flip back to Lasagne's W
        a.grads[at] = v;
        if (a.af.p) adam_elem(a.af, at, v);
    }
}

// batch slices of the weight-gradient kernels per conv layer (workgroups = slices x bands x 3 towers; partials to sum =
// slices x bands x parts).  The handle's wg_slices options (SCS_WG_SLICES="b,c,d,e" at scs_create: conv2..conv5) override for tuning runs.
struct WgSlices { int v[kConvs]; };
static WgSlices wg_slices(const scs_options& o) {
    WgSlices s{{0, 64, 64, 32, 32}};
    for (int i = 0; i < 4; ++i) if (o.wg_slices[i] >= 1 && o.wg_slices[i] <= 256) s.v[i + 1] = o.wg_slices[i];
    return s;
}

// lasagne.updates.adam over the arena
__global__ void adam_kernel(float* p, float* g, float* m, float* v, int n, float a_t, float b1, float b2, float eps, float gscale) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        adam_update(p[e], m[e], v[e], __fmul_rn(g[e], gscale), AdamCoef{a_t, b1, b2, eps});
    }
}

// the same update over up to kAdamSegs ranges of the arena (blockIdx.y = range): the fused form of the step applies it group by
// group as the gradients become final (scs_set_fused_adam)
constexpr int kAdamSegs = 12;
struct AdamSegs { int off[kAdamSegs]; int len[kAdamSegs]; };
__global__ void adam_seg_kernel(float* p, const float* g, float* m, float* v, AdamSegs sg, AdamCoef k) {
    int off = sg.off[0], len = sg.len[0];
#pragma unroll
    for (int i = 1; i < kAdamSegs; ++i) if ((int)blockIdx.y == i) { off = sg.off[i]; len = sg.len[i]; }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < len; e += gridDim.x * blockDim.x) {
        const int i = off + e;
        adam_update(p[i], m[i], v[i], g[i], k);
    }
}
static AdamCoef adam_coef(double t, float lr, float b1, float b2, float eps) {
    return AdamCoef{(float)((double)lr * std::sqrt(1.0 - std::pow((double)b2, t)) / (1.0 - std::pow((double)b1, t))), b1, b2, eps};
}

// dynamic-LDS attributes are per device: set them for the device of every handle that trains (cheap; once per handle)
static int set_train_attrs(TrainWs* w, int device) {
    if (w->attr_device == device) return SCS_OK;
    SCS_HIP(conv12_split_train_set_attr());
    SCS_HIP(conv_set_attr<TF12>()); SCS_HIP(conv_set_attr<TF3>()); SCS_HIP(conv_set_attr<TF4>()); SCS_HIP(conv_set_attr<TF5>());
    SCS_HIP(conv_set_attr<DG5>()); SCS_HIP(conv_set_attr<DG4>()); SCS_HIP(conv_set_attr<DG3>()); SCS_HIP(conv_set_attr<DG2>());
    SCS_HIP(conv_set_attr<TF3N>()); SCS_HIP(conv_set_attr<TF4N>()); SCS_HIP(conv_set_attr<DG5N>()); SCS_HIP(conv_set_attr<DG4N>());
    SCS_HIP(conv_set_attr<DG3N>());
    SCS_HIP(conv_set_attr<TF3B>()); SCS_HIP(conv_set_attr<TF4B>()); SCS_HIP(conv_set_attr<DG4B>()); SCS_HIP(conv_set_attr<DG3B>());
    SCS_HIP(wgrad_set_attr<WG2>()); SCS_HIP(wgrad_set_attr<WG3>()); SCS_HIP(wgrad_set_attr<WG4>()); SCS_HIP(wgrad_set_attr<WG5>());
    SCS_HIP(conv5_backward_set_attr());
    SCS_HIP(dense_chain_set_attr());
    w->attr_device = device;
    return SCS_OK;
}

// The one-launch dense head forms its clusters from per-XCD tickets and assumes that every XCD receives grid / 8 workgroups
// (dense_chain.h).  That holds for the round-robin dispatch of a whole MI300X / MI355X (SPX mode, 8 XCDs, >= 216 CUs); it does not
// on a CPX/DPX/QPX partition, a 6-XCD part or under a CU mask.  Checked at run time, once per handle: a probe launch of the chain's
// largest grid in which every workgroup counts itself on its XCD.  Anything but 8 equal counts -> the per-layer launches.
__global__ __launch_bounds__(512, 1) void chain_probe_kernel(unsigned int* counts /*[16]*/) {
    if (threadIdx.x == 0) atomicAdd(counts + (xcc_id() & 15), 1u);
}
static int chain_placement_probe(scs_ctx* h, hipStream_t st) {
    if (h->chain_placement != 0) return SCS_OK;
    hipDeviceProp_t prop;
    SCS_HIP(hipGetDeviceProperties(&prop, h->device));
    const dim3 grid = dense_chain_grid(kChainMaxBatch);
    h->chain_placement = -1;
    if (prop.multiProcessorCount < (int)grid.x) return SCS_OK;              // the grid must be co-resident
    unsigned int* d = nullptr;
    unsigned int host[16];
    SCS_HIP(hipMalloc(&d, sizeof(host)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(host), st);
    if (e == hipSuccess) { hipLaunchKernelGGL(chain_probe_kernel, grid, dim3(512), kChainLdsBytes, st, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    SCS_HIP(e);
    bool ok = true;
    for (int x = 0; x < 16; ++x) ok = ok && host[x] == (x < kXcds ? grid.x / kXcds : 0u);
    h->chain_placement = ok ? 1 : -1;
    return SCS_OK;
}

// ------------------------------------------------------------------------------------------- the step
static int train_step_impl(scs_ctx* h, const X0Src& x0, const float* atlas, const uint8_t* y, const int32_t* index, int n,
                           uint64_t dropout_seed, const uint8_t* masks_dev, float* loss_host, hipStream_t st, float* stage_ms) {
    if (!h->has_params) { set_error("scs_train_step before scs_set_params"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    int rc = train_ensure(h, n);
    if (rc != SCS_OK) return rc;
    TrainWs* w = h->tws;
    rc = set_train_attrs(w, h->device);
    if (rc != SCS_OK) return rc;
    rc = chain_placement_probe(h, st);
    if (rc != SCS_OK) return rc;
    float* P = h->params;
    float* G = h->grads;
    constexpr HeadOff H = head_off();
    constexpr TowerOff T0 = tower_off(0);
    const scs_options& opt = h->opt;                                               // behaviour switches: per handle (scs_set_option), no getenv per step
    // scs_set_fused_adam: the Lasagne-Adam update of this step's gradients is part of the step
    const bool fused = h->fused_adam;
    if (fused) h->adam_t += 1;
    const AdamCoef ak = adam_coef((double)h->adam_t, h->fa_lr, h->fa_b1, h->fa_b2, h->fa_eps);
    const AdamFuse af = fused ? AdamFuse{h->params, h->adam_m, h->adam_v, ak} : AdamFuse{nullptr, nullptr, nullptr, ak};
    const bool one_stream = opt.train_one_stream != 0;                             // tuning runs and the stream-order test: no side stream
    // batch-norm finalize inside the producing conv kernel up to kFinMaxBatch, as a launch of its own beyond (train_cfg.h);
    // option fin_max_batch overrides the bound (tests run both forms at one batch size)
    const bool fin_in_kernel = n <= (opt.fin_max_batch >= 0 ? opt.fin_max_batch : kFinMaxBatch);
    // banded tilings of the middle layers (train_cfg.h: two workgroups per patch-tower) at small batches; they carry the in-kernel finalize
    const bool split = fin_in_kernel && n <= (opt.split_max_batch >= 0 ? opt.split_max_batch : kSplitMaxBatch);
    hipStream_t sd = one_stream ? st : w->side;
    // The tail of the step.  The conv2 weight gradient - the last big kernel, which cannot start before the conv2 data gradient ends -
    // follows that kernel on the CALLER's stream (in order: no cross-stream wait on the critical path) with its reduction behind
    // it; the layer-1 launch and the reduction of the conv3..5 partials go to the side stream, behind the conv3 weight gradient,
    // and run beside it.  option tail_side (tuning runs) restores the round-2 order: everything but the layer-1 launch on the side stream.
    const bool tail_side = opt.tail_side != 0;
    const bool tail_on_main = !one_stream && !tail_side;
    // the conv3..5 partial-sum reduction (and, fused update, those weights' Adam launch) right behind the conv3 weight gradient on
    // the side stream - beside the conv2 data gradient, where that stream has slack - instead of behind the layer-1 launch at its end
    // (option wgfin_early = 0 restores that order): 0.5547 -> 0.5449 ms at batch 256, 0.4131 -> 0.4035 at 128, same GPU
    const bool wgfin_early = opt.wgfin_early != 0;
    const bool adam_early = wgfin_early && tail_on_main;
    // option train_split: the conv1 + conv2 block of the forward pass on the f16 matrix cores with split operands (conv_split_train.h)
    const bool tsplit = opt.train_split > 0;

    w->prof = stage_ms != nullptr;
    if (w->prof) {
        for (int i = 0; i < 2 * ST_COUNT; ++i) if (!w->pev[i]) SCS_HIP(hipEventCreate(&w->pev[i]));
        for (int i = 0; i < ST_COUNT; ++i) w->pused[i] = false;
    }
    auto run = [&](int stage, hipStream_t s, auto&& launch) {
        if (w->prof) (void)hipEventRecord(w->pev[2 * stage], s);
        launch();
        if (w->prof) { (void)hipEventRecord(w->pev[2 * stage + 1], s); w->pused[stage] = true; }
    };
    auto fork = [&]() { if (sd != st) { (void)hipEventRecord(w->ev_fork, st); (void)hipStreamWaitEvent(sd, w->ev_fork, 0); } };   // sd sees what st has done
    auto join = [&]() { if (sd != st) { (void)hipEventRecord(w->ev_join, sd); (void)hipStreamWaitEvent(st, w->ev_join, 0); } };   // st waits for sd
    auto c3 = [&](auto& a, int i) { CPtr3 r; for (int t = 0; t < kTowers; ++t) r.p[t] = reinterpret_cast<const float*>(a[t][i]); return r; };
    auto p3 = [&](auto& a, int i) { Ptr3 r; for (int t = 0; t < kTowers; ++t) r.p[t] = reinterpret_cast<float*>(a[t][i]); return r; };

    // every trainable entry of the gradient arena is overwritten below; the rest (BN running statistics, PReLU alphas)
    // has been zero since the arena was allocated

    // ---- prep: operand packs (unfolded), transposed dense weights, atlas columns + batch labels, dropout masks
    const int units1 = moment_units(n), ppb = moment_ppb(n);
    if (w->tickets_dirty) {                     // a launch of the previous step failed in mid-flight: re-arm the arrival tickets
        SCS_HIP(hipMemsetAsync(w->fin_counter, 0, kTowers * fin_counter_words((size_t)2 * w->cap) * sizeof(unsigned int), st));
        w->tickets_dirty = false;
    }
    const uint8_t* masks = masks_dev ? masks_dev : w->masks;
    float* wT2 = w->wT;                       // [270][544]  = fc_2.W[:540]^T
    float* wT1 = wT2 + 270 * kTld;            // [540][544]  = FC1.W^T
    float* wTd = wT1 + 540 * kTld;            // [3][180][544] = d1.W^T
    float* w2p = wTd + 3 * 180 * kTld;        // [556][272]  = fc_2.W zero padded
    float* wc5 = w2p + kW2Floats;             // [3][60][368] = conv5.W of every tower in correlation form (conv5_bwd.h)
    {
        PrepArgs pa{};
        for (int t = 0; t < kTowers; ++t) {
            const TowerOff T = tower_off(t * kTowerFloats);
            for (int i = 1; i < kConvs; ++i) pa.pack[t * (kConvs - 1) + i - 1] = PackJob{P + T.convW[i], w->wpk[t][i], w->wdg[t][i], kCin[i], kCout[i]};
            pa.tr[2 + t] = TrProblem{P + T.d1W, kD1, wTd + t * kD1 * kTld, kTld, kFlat, kD1};
        }
        pa.tr[0] = TrProblem{P + H.fc2W, kFC2, wT2, kTld, kFC1, kFC2};
        pa.tr[1] = TrProblem{P + H.fc1W, kFC1, wT1, kTld, kFlat, kFC1};
        pa.atlas = atlas; pa.y = y; pa.index = index; pa.cat2 = w->cat2; pa.yb = w->yb; pa.n = n;
        pa.n_rows = index ? x0.n_samples : n;
        pa.fc2W = P + H.fc2W; pa.w2p = w2p; pa.chain_counter = w->chain_counter; pa.loss_rows = w->loss_rows;
        pa.c5W = P + T0.convW[4]; pa.wc5 = wc5; pa.c5_tower_stride = kTowerFloats;
        pa.masks = masks_dev ? nullptr : w->masks; pa.mask_total = (size_t)n * kFlat * 5; pa.seed = dropout_seed;
        // ---- forward, training mode: the input moments ride in the same launch; up to kBn1FusedMaxBatch the workgroup that
        // completes a tower's moments also finalizes conv1's batch statistics there (beside the prep jobs: no launch of its own on
        // the step's critical path); option bn1_fused_max_batch overrides the bound (0: always a launch; tests run both forms)
        Bn1Args b1{P, T0.convW[0], T0.bn[0], kTowerFloats, w->mom, Ptr3{{w->w1p[0], w->w1p[1], w->w1p[2]}}, Ptr3{{w->l1p[0], w->l1p[1], w->l1p[2]}},
                   Ptr3{{w->l1m[0], w->l1m[1], w->l1m[2]}}, nullptr, nullptr, Ptr3{}};
        SplitPackJobs sj{};
        if (tsplit) {
            for (int t = 0; t < kTowers; ++t) {
                b1.w1s.p[t] = w->w1s[t];
                sj.W[t] = P + T0.convW[1] + t * kTowerFloats; sj.wp[t] = w->w2s[t]; sj.inv[t] = w->inv2s[t];
            }
            sj.cin = kCin[1]; sj.cout = kCout[1]; sj.flag = w->split_flag;
        }
        const bool bn1_fused = n <= (opt.bn1_fused_max_batch >= 0 ? std::min(opt.bn1_fused_max_batch, w->cap) : kBn1FusedMaxBatch);
        if (bn1_fused) { b1.counter = w->fin_counter; b1.rec = w->fin_grp; }
        run(ST_PREP, st, [&] {
            hipLaunchKernelGGL(prep_moments_kernel, prep_moments_grid(units1, tsplit ? kTowers : 0), dim3(256), 0, st, pa, x0, w->mom_part, n,
                               ppb, units1, b1, sj);
        });
        if (!bn1_fused) run(ST_FIN1, st, [&] { hipLaunchKernelGGL(bn1_final_kernel, dim3(kTowers), dim3(1024), 0, st, w->mom_part, units1, n, b1); });
    }
    // dense head (weights straight from the arena: [K][N] row-major).  Decided HERE, ahead of the conv blocks: with the one-launch
    // head the conv5 block's batch norm is the head's business on both sides (tail_fused) and the launches around it go away
    GemmArgs g{};
    g.M = n; g.K4 = kFlat; g.K = kFlat; g.N = kD1; g.relu = 1;
    const uint8_t* const mask5 = masks;                            // [3][n][540]
    const uint8_t* const mask1 = masks + (size_t)3 * n * kFlat;    // f1_drop [n][540]
    const uint8_t* const mask2 = mask1 + (size_t)n * kFlat;        // f2_drop [n][540]
    for (int t = 0; t < kTowers; ++t) {
        const TowerOff T = tower_off(t * kTowerFloats);
        // feat = relu(H5 W + b) (kept: it gates the backward pass), featd = f1_drop(feat), both from the GEMM epilogue
        g.p[t] = GemmProblem{w->H5[t], kFlat, P + T.d1W, kD1, P + T.d1b, w->feat + t * kD1, kFlat,
                             mask1 + t * kD1, kFlat, nullptr, 0, w->featd + t * kD1, kFlat};
    }
    // Small batches: the whole dense head - these three layers, the out layer with its loss, and the three data-gradient
    // layers below - is ONE launch of 9-workgroup clusters (dense_chain.h; same tiles, same sums, same bits); otherwise one
    // launch per layer.  option dense_chain_max_batch overrides the bound (0: never; tests run both forms at one batch size);
    // SCS_CHAIN_STAMPS prints the one-launch form's per-tile phase times to stderr (diagnostic, synchronises).
    // ... and only on a device whose dispatcher gives every XCD grid / 8 workgroups (probed once per handle: chain_placement_ok)
    const int chain_max = opt.dense_chain_max_batch >= 0 ? std::min(opt.dense_chain_max_batch, kChainMaxBatch) : kChainDefaultMaxBatch;
    const bool chain = n <= chain_max && dense_chain_ok(n, g, kTowers) && h->chain_placement > 0;
    const bool c5_fused = n <= (opt.c5_fused_max_batch >= 0 ? std::min(opt.c5_fused_max_batch, kC5MaxBatch) : kC5MaxBatch);
    // tail_fused: bn5 + dropout and bn5 backward are no launches of their own - the head finalizes the conv5 block's statistics and
    // applies them while it stages its first A operand, its last stage leaves the (d beta, d gamma) column sums and the conv5
    // backward launch turns them into its coefficients.  Needs both one-launch forms; option tail_fused = 0 restores the launches.
    const bool tail = chain && c5_fused && opt.tail_fused != 0;
    static_assert(TF5::G > 1 && TF5::G * TF5::COUT <= TF5::NTH && TF5::G * TF5::COUT * 2 > TF5::NTH && TF5::NB == 1,
                  "tail_fused reads the per-workgroup statistics units of the conv5 forward kernel (conv_mfma.h: a.wgstats)");

    constexpr int kStatBands[kConvs] = {1, TF12::NB, TF3::NB, TF4::NB, TF5::NB};
    for (int i = 1; i < kConvs; ++i) {
        ConvArgs a{};
        a.n = n;
        a.stats = w->stat_part;
        a.wgstats = (tail && i == kConvs - 1) ? w->c5_units : nullptr;
        for (int t = 0; t < kTowers; ++t) {
            a.w[t] = w->wpk[t][i]; a.out[t] = w->Z[t][i];
            if (i == 1) { a.w1[t] = w->w1p[t]; a.pout[t] = w->ZP[t]; a.bias[t] = P + T0.bn[1] + t * kTowerFloats + kCout[1]; }   // gamma2: which extremum pool1 keeps
            else { a.in[t] = (i == 2) ? w->ZP[t] : w->Z[t][i - 1]; a.bn[t] = reinterpret_cast<const float*>(w->ss[t][i - 1]); a.xout[t] = w->X[t][i]; }
        }
        a.x0 = x0;
        const int C = kCout[i], Hh = kHout[i];
        const double count = (double)n * Hh * Hh;
        a.fin.counter = w->fin_counter; a.fin.grp = w->fin_grp; a.fin.count = count;
        for (int t = 0; t < kTowers; ++t) {
            a.fin.bn[t] = P + T0.bn[i] + t * kTowerFloats;
            a.fin.ss[t] = reinterpret_cast<float*>(w->ss[t][i]); a.fin.coef[t] = reinterpret_cast<float*>(w->low4[t][i]);
        }
        const bool kFwdFin[kConvs] = {false, TF12::FIN != 0 || (tsplit && fin_in_kernel), TF3::FIN != 0 && fin_in_kernel, TF4::FIN != 0 && fin_in_kernel, TF5::FIN != 0};
        if (i == 1 && tsplit) {
            SplitTrain12Args sa{};
            sa.x0 = x0; sa.n = n; sa.stats = w->stat_part; sa.flag = w->split_flag;
            for (int t = 0; t < kTowers; ++t) {
                sa.w1[t] = w->w1s[t]; sa.w2[t] = w->w2s[t]; sa.inv2[t] = w->inv2s[t];
                sa.gamma2[t] = P + T0.bn[1] + t * kTowerFloats + kCout[1];
                sa.out[t] = w->Z[t][1]; sa.pout[t] = w->ZP[t];
            }
            if (fin_in_kernel) sa.fin = a.fin;                    // the kernel's last workgroup finalizes BN2 (no launch of its own)
            run(ST_CONV12, st, [&] { conv12_split_train_launch(sa, st); });
        } else if (i == 1) run(ST_CONV12, st, [&] { conv_launch<TF12>(a, st); });
        else if (i == 2) run(ST_CONV3, st, [&] { if (split) conv_launch<TF3B>(a, st); else if (fin_in_kernel) conv_launch<TF3>(a, st); else conv_launch<TF3N>(a, st); });
        else if (i == 3) run(ST_CONV4, st, [&] { if (split) conv_launch<TF4B>(a, st); else if (fin_in_kernel) conv_launch<TF4>(a, st); else conv_launch<TF4N>(a, st); });
        else run(ST_CONV5, st, [&] { conv_launch<TF5>(a, st); });
        const int units = (i == 1 && tsplit) ? n : n * kStatBands[i];      // the split kernel leaves one statistics unit per patch-tower
        if (kFwdFin[i]) {
            // the conv kernel's last workgroup has already finalized this layer's statistics
        } else if (i < kConvs - 1) {
            run(ST_FIN2 + 2 * (i - 1), st, [&] {
                hipLaunchKernelGGL(bn_fwd_final_kernel<false>, dim3(C, kTowers), dim3(256), 0, st, w->stat_part, units, count, C, P, T0.bn[i],
                                   kTowerFloats, p3(w->ss, i), p3(w->low4, i), CPtr3{}, Ptr3{}, (const uint8_t*)nullptr, n);
            });
        } else if (!tail) {
            run(ST_BN5, st, [&] {
                hipLaunchKernelGGL(bn_fwd_final_kernel<true>, dim3(C, kTowers), dim3(256), 0, st, w->stat_part, units, count, C, P, T0.bn[i],
                                   kTowerFloats, p3(w->ss, i), p3(w->low4, i), c3(w->Z, i), Ptr3{{w->H5[0], w->H5[1], w->H5[2]}}, mask5, n);
            });
        }
    }
    DenseChainArgs ch{};
    for (int t = 0; t < kTowers; ++t) ch.d1[t] = g.p[t];
    if (tail) {
        const int nwg5 = (n + TF5::G - 1) / TF5::G;
        for (int t = 0; t < kTowers; ++t) {
            ch.d1[t].A = w->Z[t][kConvs - 1];                      // the raw conv5 rows: BN + ReLU + dropout on the way in
            ch.bn5.units[t] = w->c5_units + (size_t)t * nwg5 * 120;
            ch.bn5.bn[t] = P + T0.bn[kConvs - 1] + t * kTowerFloats;
            ch.bn5.ss[t] = reinterpret_cast<float*>(w->ss[t][kConvs - 1]);
            ch.bn5.low4[t] = reinterpret_cast<float*>(w->low4[t][kConvs - 1]);
            ch.bn5.mask[t] = mask5 + (size_t)t * n * kFlat;
            ch.bn5.H5[t] = w->H5[t];
            ch.Z5[t] = w->Z[t][kConvs - 1];
        }
        ch.bn5.nwg = nwg5;
        ch.part5 = w->part5; ch.clow = w->clow;
    }
    if (!chain) run(ST_D1, st, [&] { gemm_launch(g, kTowers, st); });
    g.K4 = kFC1; g.K = kFC1; g.N = kFC1;
    g.p[0] = GemmProblem{w->featd, kFlat, P + H.fc1W, kFC1, P + H.fc1b, w->R1, kFC1, mask2, kFC1, nullptr, 0, w->cat2, kFC1ld};
    ch.fc1 = g.p[0];
    if (!chain) run(ST_FC1, st, [&] { gemm_launch(g, 1, st); });        // R1 = relu(.), cat2[:, :540] = f2_drop(R1)
    g.K4 = kFC1ld; g.K = kFC1 + kAtlas; g.N = kFC2;
    g.p[0] = GemmProblem{w->cat2, kFC1ld, w2p, kFC2ld, P + H.fc2b, w->R2, kFC2ld};
    ch.fc2 = g.p[0];
    if (!chain) {
        run(ST_FC2, st, [&] { gemm_launch(g, 1, st); });
        run(ST_HEAD, st, [&] {
            hipLaunchKernelGGL(head_train_kernel, dim3((n + 15) / 16), dim3(256), 0, st, w->R2, P + H.outW, P + H.outb, w->yb, n, w->probs, w->dlog,
                               w->loss_rows, w->dR2);
        });
    }

    // ---- backward: dense head.  d logits and dR2 (through fc_2's ReLU) came out of head_train_kernel's pass.  Data
    // gradients stay on the caller's stream (st); every dense weight / bias gradient is one grouped launch on the side
    // stream (sd) once the last upstream gradient (dfeat) exists.
    GemmArgs bg{};
    bg.relu = 0;
    bg.M = n; bg.K4 = kFC2ld; bg.K = kFC2; bg.N = kFC1;
    bg.p[0] = GemmProblem{w->dR2, kFC2ld, wT2, kTld, nullptr, w->dcat, kFC1, mask2, kFC1, w->R1, kFC1, nullptr, 0};
    ch.dcat = bg.p[0];
    if (!chain) run(ST_DCAT, st, [&] { gemm_launch(bg, 1, st); });      // through f2_drop and FC1's ReLU in the epilogue
    bg.M = n; bg.K4 = kFC1; bg.K = kFC1; bg.N = kFlat;
    bg.p[0] = GemmProblem{w->dcat, kFC1, wT1, kTld, nullptr, w->dfeat, kFlat, mask1, kFlat, w->feat, kFlat, nullptr, 0};
    ch.dfeat = bg.p[0];
    if (!chain) run(ST_DFEAT, st, [&] { gemm_launch(bg, 1, st); });     // through f1_drop and the d1 ReLUs in the epilogue
    if (chain) {
        for (int t = 0; t < kTowers; ++t)
            ch.dh5[t] = GemmProblem{w->dfeat + t * kD1, kFlat, wTd + t * kD1 * kTld, kTld, nullptr, w->dH5[t], kFlat,
                                    mask5 + (size_t)t * n * kFlat, kFlat, nullptr, 0, nullptr, 0};
        ch.outW = P + H.outW; ch.outb = P + H.outb; ch.y = w->yb; ch.probs = w->probs; ch.dlog = w->dlog; ch.loss_rows = w->loss_rows;
        ch.R2 = w->R2; ch.dR2 = w->dR2; ch.counters = w->chain_counter; ch.n = n;
        static const bool stamps_on = getenv("SCS_CHAIN_STAMPS") != nullptr;
        if (stamps_on && !w->chain_stamps) {
            SCS_HIP(hipMalloc(&w->chain_stamps, (size_t)(kChainMaxBatch / 16) * kChainMembers * kChainStamps * sizeof(unsigned long long)));
        }
        ch.stamps = stamps_on ? w->chain_stamps : nullptr;
        if (stamps_on) SCS_HIP(hipMemsetAsync(w->chain_stamps, 0, (size_t)(kChainMaxBatch / 16) * kChainMembers * kChainStamps * sizeof(unsigned long long), st));
        run(ST_CHAIN, st, [&] { hipLaunchKernelGGL(dense_chain_kernel, dense_chain_grid(n), dim3(512), kChainLdsBytes, st, ch); });
        if (stamps_on) {                                                // diagnostic: where the chain's time goes, per stage (stderr)
            SCS_HIP(hipStreamSynchronize(st));
            const int wgs = ((n + 15) / 16) * kChainMembers;
            std::vector<unsigned long long> hs((size_t)wgs * kChainStamps);
            SCS_HIP(hipMemcpy(hs.data(), w->chain_stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int g2 = 0; g2 < wgs; ++g2) { t0 = std::min(t0, hs[(size_t)g2 * kChainStamps + 62]); for (int i = 0; i < kChainStamps; ++i) t1 = std::max(t1, hs[(size_t)g2 * kChainStamps + i]); }
            fprintf(stderr, "dense chain, n = %d, %d workgroups: span %.2f us (first start to last stamp)\n", n, wgs, (double)(t1 - t0) * 0.01);
            static const char* const kNames[7] = {"d1", "FC1", "fc_2", "head+dcat", "-", "dfeat", "dH5 x3"};
            for (int sidx = 0; sidx < 7; ++sidx) {
                double d[6] = {0, 0, 0, 0, 0, 0}, at0 = 0, at5 = 0; int cnt = 0;
                for (int g2 = 0; g2 < wgs; ++g2) {
                    const unsigned long long* q = hs.data() + (size_t)g2 * kChainStamps + 8 * sidx;
                    const int last = sidx == 6 ? 3 : 5;
                    bool full = true; for (int i = 0; i <= last; ++i) full = full && q[i] != 0;
                    if (!full) continue;
                    for (int i = 1; i <= last; ++i) d[i] += (double)(q[i] - q[i - 1]) * 0.01;
                    at0 += (double)(q[0] - t0) * 0.01; at5 += (double)(q[last] - t0) * 0.01; ++cnt;
                }
                if (cnt) fprintf(stderr, "  %-7s (%3d wgs) starts %6.2f ends %6.2f us | operands %5.2f  matrix %5.2f  reduce+store %5.2f  drain %5.2f  barrier %5.2f\n",
                                 kNames[sidx], cnt, at0 / cnt, at5 / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt);
            }
        }
    }
    fork();                                                             // dlog, dR2, dcat, dfeat are final for sd
    {
        WgGroupArgs wg{};
        wg.K = n;
        wg.loss_rows = w->loss_rows; wg.loss_mean = w->loss_dev;
        wg.poison = tsplit ? w->split_flag : nullptr;
        wg.p[0] = WgProblem{w->R2, kFC2ld, w->dlog, 16, G + H.outW, kClasses, G + H.outb, kFC2, kClasses};
        wg.p[1] = WgProblem{w->cat2, kFC1ld, w->dR2, kFC2ld, G + H.fc2W, kFC2, G + H.fc2b, kFC1 + kAtlas, kFC2};
        wg.p[2] = WgProblem{w->featd, kFlat, w->dcat, kFC1, G + H.fc1W, kFC1, G + H.fc1b, kFlat, kFC1};
        for (int t = 0; t < kTowers; ++t) {
            const TowerOff T = tower_off(t * kTowerFloats);
            wg.p[3 + t] = WgProblem{w->H5[t], kFlat, w->dfeat + t * kD1, kFlat, G + T.d1W, kD1, G + T.d1b, kFlat, kD1};
        }
        run(ST_WGDENSE, sd, [&] { dense_wgrad_launch(wg, kWgProblems, sd); });
        // every gradient of FC1 / fc_2 / out (the contiguous tail of the arena) is final behind this launch, ~0.3 ms before the
        // step ends: a data-parallel caller may start exchanging that bucket now (scs_wait_head_grads)
        (void)hipEventRecord(w->ev_head, sd);
        w->head_recorded = true;
        if (fused) {
            // nothing reads FC1 / fc_2 / out / d1 parameters from the arena any more in this step (the head's forward is done, the
            // data gradients use the transposed copies of the prep launch): 84 % of the update runs here, beside the conv backward pass
            AdamSegs sg{};
            sg.off[0] = H.fc1W; sg.len[0] = H.end - H.fc1W;
            for (int t = 0; t < kTowers; ++t) { const TowerOff T = tower_off(t * kTowerFloats); sg.off[1 + t] = T.d1W; sg.len[1 + t] = (t + 1) * kTowerFloats - T.d1W; }
            run(ST_ADAM1, sd, [&] { hipLaunchKernelGGL(adam_seg_kernel, dim3(216, 4), dim3(256), 0, sd, P, G, h->adam_m, h->adam_v, sg, ak); });
        }
    }
    bg.M = n; bg.K4 = kD1; bg.K = kD1; bg.N = kFlat;
    for (int t = 0; t < kTowers; ++t)         // through the dropout in front of d1 (mask5 of the tower) in the epilogue
        bg.p[t] = GemmProblem{w->dfeat + t * kD1, kFlat, wTd + t * kD1 * kTld, kTld, nullptr, w->dH5[t], kFlat,
                              mask5 + (size_t)t * n * kFlat, kFlat, nullptr, 0, nullptr, 0};
    if (!chain) run(ST_DH5, st, [&] { gemm_launch(bg, kTowers, st); });

    // ---- backward: conv blocks 5..2.  Per layer i: BN-backward coefficients from the partial sums (left by the data
    // gradient of layer i+1, or computed directly for the 3x3 planes of conv5), then the data-gradient kernel of layer i
    // (d z_i while staging, stored for the weight gradient; epilogue: partial sums of layer i-1), then the weight gradient
    // of layer i on the side stream.
    const WgSlices slices = wg_slices(opt);
    WgFinalArgs fa{};
    fa.grads = G; fa.tower_stride = kTowerFloats;
    // (the update inside these reductions was tried: their stores are a transpose - stride Cin * 9 floats between neighbouring lanes -
    // and three more arrays accessed that way doubled their time; the conv weights are updated by one coalesced launch behind them)
    constexpr int kDgBands[kConvs] = {0, DG2::NB, DG3::NB, DG4::NB, DG5::NB};
    // conv5 block: data gradient + weight gradient as one launch of two small GEMMs per workgroup (conv5_bwd.h) up to
    // kC5MaxBatch (c5_fused, decided above); option c5_fused_max_batch overrides the bound (0: the generic kernels; tests run both forms)
    const int c5_units = conv5_backward_workgroups(n);
    const bool kDgFin[kConvs + 1] = {false, DG2::FIN != 0, DG3::FIN != 0 && fin_in_kernel, DG4::FIN != 0 && fin_in_kernel,
                                     c5_fused ? fin_in_kernel : (DG5::FIN != 0 && fin_in_kernel), false};
    for (int i = kConvs - 1; i >= 1; --i) {
        const int C = kCout[i], Hh = kHout[i];
        const double count = (double)n * Hh * Hh;
        CPtr3 dUp;
        for (int t = 0; t < kTowers; ++t) dUp.p[t] = (i == kConvs - 1) ? w->dH5[t] : w->dX[t][i + 1];
        if (i == kConvs - 1 && tail) {
            // the head's last stage left the conv5 block's (d beta, d gamma) column sums; the launch below adds them itself
        } else if (i == kConvs - 1) {
            run(ST_BN5B, st, [&] {
                hipLaunchKernelGGL(bn_bwd_final_kernel<true>, dim3(C, kTowers), dim3(256), 0, st, (const float*)nullptr, 0, count, C, c3(w->low4, i),
                                   p3(w->cf4, i), G, T0.bn[i], kTowerFloats, c3(w->Z, i), dUp, n);
            });
        } else if (!kDgFin[i + 1]) {
            run(ST_FINB4 + 2 * (kConvs - 2 - i), st, [&] {
                const int units = (i + 1 == kConvs - 1 && c5_fused) ? c5_units : n * kDgBands[i + 1];
                hipLaunchKernelGGL(bn_bwd_final_kernel<false>, dim3(C, kTowers), dim3(256), 0, st, w->stat_part, units, count, C,
                                   c3(w->low4, i), p3(w->cf4, i), G, T0.bn[i], kTowerFloats, CPtr3{}, CPtr3{}, n);
            });
        }                                                               // else: the data-gradient kernel of layer i+1 finalized layer i
        if (i == kConvs - 1 && c5_fused) {
            // conv5 block: d z, the weight-gradient partial of every four patch-towers, the data gradient and the conv4 block's
            // (d beta, d gamma) sums in one launch on the caller's stream; nothing of this layer goes to the side stream
            C5BArgs b{};
            float* part = w->wg_part + (size_t)(i - 1) * kTowers * kWgPartFloats;
            for (int t = 0; t < kTowers; ++t) {
                b.Z5[t] = w->Z[t][i]; b.dH5[t] = dUp.p[t]; b.cf5[t] = reinterpret_cast<const float*>(w->cf4[t][i]); b.X5[t] = w->X[t][i];
                b.Wc[t] = wc5 + (size_t)t * kC5WcFloats; b.dX[t] = w->dX[t][i];
                b.Z4[t] = w->Z[t][i - 1]; b.low4[t] = reinterpret_cast<const float*>(w->low4[t][i - 1]);
            }
            b.stats = w->stat_part; b.part = part; b.n = n;
            if (tail) {
                b.part5 = w->part5; b.blocks5 = (n + 15) / 16;
                for (int t = 0; t < kTowers; ++t) { b.low5[t] = reinterpret_cast<const float*>(w->low4[t][i]); b.gbn5[t] = G + T0.bn[i] + t * kTowerFloats; }
            }
            b.fin_on = fin_in_kernel ? 1 : 0;
            b.fin.counter = w->fin_counter; b.fin.grp = w->fin_grp; b.fin.count = (double)n * kHout[i - 1] * kHout[i - 1];
            for (int t = 0; t < kTowers; ++t) { b.fin.bn[t] = G + T0.bn[i - 1] + t * kTowerFloats; b.fin.coef[t] = reinterpret_cast<float*>(w->cf4[t][i - 1]); }
            static const bool c5_stamps = getenv("SCS_C5_STAMPS") != nullptr;          // diagnostic: phase times of the launch (stderr)
            if (c5_stamps && !w->chain_stamps)
                SCS_HIP(hipMalloc(&w->chain_stamps, (size_t)(kChainMaxBatch / 16) * kChainMembers * kChainStamps * sizeof(unsigned long long)));
            b.stamps = c5_stamps ? w->chain_stamps : nullptr;
            // With the head's column sums (tail) the launch splits: the data path on the caller's stream, the weight-gradient partial
            // - which feeds nothing before the partial-sum reduction - on the side stream behind the dense weight gradients
            // up to batch kC5SplitMaxBatch (option c5_split: the bound, 0 = never).  Same GPU, two rounds: batch 128 0.3974 / 0.3994 ->
            // 0.3898 / 0.3907 ms; batch 256 0.5349 / 0.5365 -> 0.5424 / 0.5438 (there the side stream is busy with the dense weight
            // gradients and the early Adam launch at that point, and the partial launch only takes the data path's CUs)
            const bool c5_split = tail && n <= (opt.c5_split >= 0 ? opt.c5_split : kC5SplitMaxBatch);
            run(ST_C5B, st, [&] { conv5_backward_launch(b, st, c5_split ? 1 : 0); });
            if (c5_split) run(ST_C5W, sd, [&] { conv5_backward_launch(b, sd, 2); });
            if (c5_stamps && (size_t)kTowers * c5_units * 8 <= (size_t)(kChainMaxBatch / 16) * kChainMembers * kChainStamps) {
                SCS_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> hs((size_t)kTowers * c5_units * 8);
                SCS_HIP(hipMemcpy(hs.data(), w->chain_stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double d[7] = {0, 0, 0, 0, 0, 0, 0};
                unsigned long long t0 = ~0ull, t1 = 0;
                const int wgs = kTowers * c5_units;
                for (int g2 = 0; g2 < wgs; ++g2) {
                    for (int k = 1; k <= 6; ++k) d[k] += (double)(hs[g2 * 8 + k] - hs[g2 * 8 + k - 1]) * 0.01 / wgs;
                    t0 = std::min(t0, hs[g2 * 8]); t1 = std::max(t1, hs[g2 * 8 + 6]);
                }
                fprintf(stderr, "conv5 backward, n = %d, %d workgroups, span %.2f us: stage %.2f | wgrad GEMM + partial store %.2f | dgrad GEMM + Y %.2f | "
                                "col2im %.2f | Z4 to LDS + sums %.2f | reduce + store %.2f\n", n, wgs, (double)(t1 - t0) * 0.01, d[1], d[2], d[3], d[4], d[5], d[6]);
            }
            fa.j[i - 1] = WgFinalJob{part, c5_units, kCin[i], C, T0.convW[i]};
            continue;
        }
        // data gradient (d z_i while staging, stored for the weight gradient; epilogue: partial sums of layer i-1)
        ConvArgs a{};
        a.n = n;
        for (int t = 0; t < kTowers; ++t) {
            a.in[t] = w->Z[t][i]; a.dup[t] = dUp.p[t]; a.bn[t] = reinterpret_cast<const float*>(w->cf4[t][i]); a.xout[t] = w->dZ[t][i];
            a.w[t] = w->wdg[t][i];
            if (i >= 2) { a.out[t] = w->dX[t][i]; a.zlow[t] = w->Z[t][i - 1]; a.bnlow[t] = reinterpret_cast<const float*>(w->low4[t][i - 1]); }
            else { a.bnlow[t] = w->l1p[t]; }
        }
        a.stats = (i >= 2) ? w->stat_part : w->l1_part;
        a.x0 = x0;
        if (i >= 2) {                                                   // finalize of the layer below, done by the kernel's last workgroup
            a.fin.counter = w->fin_counter; a.fin.grp = w->fin_grp; a.fin.count = (double)n * kHout[i - 1] * kHout[i - 1];
            for (int t = 0; t < kTowers; ++t) {
                a.fin.bn[t] = G + T0.bn[i - 1] + t * kTowerFloats;
                a.fin.coef[t] = reinterpret_cast<float*>(w->cf4[t][i - 1]);
            }
        }
        if (i == 4) run(ST_DG5, st, [&] { if (fin_in_kernel) conv_launch<DG5>(a, st); else conv_launch<DG5N>(a, st); });
        else if (i == 3) run(ST_DG4, st, [&] { if (split) conv_launch<DG4B>(a, st); else if (fin_in_kernel) conv_launch<DG4>(a, st); else conv_launch<DG4N>(a, st); });
        else if (i == 2) run(ST_DG3, st, [&] { if (split) conv_launch<DG3B>(a, st); else if (fin_in_kernel) conv_launch<DG3>(a, st); else conv_launch<DG3N>(a, st); });
        else run(ST_DG2, st, [&] { conv_launch<DG2>(a, st); });
        // weight gradient of this layer on the side stream (it reads the d z the kernel above stored)
        fork();
        if (fused && i == 2) {
            // (d beta, d gamma) of the conv2..conv5 blocks are final (the last of them left by the kernel above, or by its finalize
            // launch): their update, a tiny launch on the side stream in front of the conv3 weight gradient
            AdamSegs sb{};
            for (int t = 0; t < kTowers; ++t)
                for (int q = 1; q < kConvs; ++q) { sb.off[t * 4 + q - 1] = T0.bn[q] + t * kTowerFloats; sb.len[t * 4 + q - 1] = 2 * kCout[q]; }
            run(ST_ADAM2, sd, [&] { hipLaunchKernelGGL(adam_seg_kernel, dim3(1, 12), dim3(128), 0, sd, P, G, h->adam_m, h->adam_v, sb, ak); });
        }
        WgArgs wa{};
        float* part = w->wg_part + (size_t)(i - 1) * kTowers * kWgPartFloats;
        for (int t = 0; t < kTowers; ++t) {
            wa.X[t] = w->X[t][i]; wa.w1[t] = w->w1p[t]; wa.dZ[t] = w->dZ[t][i];
        }
        wa.x0 = x0;
        constexpr int kParts[kConvs] = {0, WG2::NB, WG3::NB, WG4::NB, WG5::NB};
        const int max_slices = (int)(kWgPartFloats / ((size_t)kParts[i] * kCin[i] * 9 * C));      // what the partial buffer holds
        wa.part = part; wa.n = n; wa.slices = std::min(std::min(n, slices.v[i]), max_slices);
        if (i == 4) run(ST_WG5, sd, [&] { wgrad_launch<WG5>(wa, sd); });
        else if (i == 3) run(ST_WG4, sd, [&] { wgrad_launch<WG4>(wa, sd); });
        else if (i == 2) run(ST_WG3, sd, [&] { wgrad_launch<WG3>(wa, sd); });
        else { hipStream_t s2 = tail_on_main ? st : sd; run(ST_WG2, s2, [&] { wgrad_launch<WG2>(wa, s2); }); }
        fa.j[i - 1] = WgFinalJob{part, wa.slices * kParts[i], kCin[i], C, T0.convW[i]};
        if (i == 2 && (!tail_on_main || wgfin_early)) {
            // the partial sums of conv5..conv3 reduced on the side stream right behind the conv3 weight gradient - in FRONT of the
            // side stream's wait for the conv2 data gradient, so beside that kernel and not beside the conv2 weight gradient (the
            // round-2 order did the same, but with the conv2 weight gradient behind it on this stream)
            fa.z0 = 1;
            run(ST_WGFINAL, sd, [&] { hipLaunchKernelGGL(wgrad_final_kernel, dim3((60 * 40 * 9 + 63) / 64, kTowers, 3), dim3(256), 0, sd, fa); });
            if (fused && adam_early) {
                // ... and their update right behind it on this stream (coalesced, 9 ranges); conv2.W is updated by its own reduction
                AdamSegs sc{};
                for (int t = 0; t < kTowers; ++t)
                    for (int q = 2; q < kConvs; ++q) { sc.off[t * 3 + q - 2] = T0.convW[q] + t * kTowerFloats; sc.len[t * 3 + q - 2] = kCout[q] * kCin[q] * 9; }
                run(ST_ADAM3, sd, [&] { hipLaunchKernelGGL(adam_seg_kernel, dim3(22, 9), dim3(256), 0, sd, P, G, h->adam_m, h->adam_v, sc, ak); });
            }
        }
    }
    // conv1 block: gradients from the layer-1 sums and the input moments; reductions of the weight-gradient partials (streams: see
    // tail_on_main above; the last fork made the side stream wait for the conv2 data gradient)
    hipStream_t sl = tail_on_main ? sd : st, s2 = tail_on_main ? st : sd;
    run(ST_L1FINAL, sl, [&] {
        hipLaunchKernelGGL(l1_final_kernel, dim3(20, kTowers), dim3(256), 0, sl, w->l1_part, n * kL1Bands, n, w->mom,
                           CPtr3{{w->l1p[0], w->l1p[1], w->l1p[2]}}, CPtr3{{w->l1m[0], w->l1m[1], w->l1m[2]}}, G, T0.convW[0], T0.bn[0], kTowerFloats, af);
    });
    if (tail_on_main && !wgfin_early) {
        fa.z0 = 1;
        run(ST_WGFINAL, sd, [&] { hipLaunchKernelGGL(wgrad_final_kernel, dim3((60 * 40 * 9 + 63) / 64, kTowers, 3), dim3(256), 0, sd, fa); });
    }
    fa.z0 = 0;
    if (fused && adam_early) fa.af = af;      // conv2.W: 10 800 elements, updated by the reduction that has just summed them (its transposed stores
                                              // are the price: +1 us on this launch against a launch of its own behind the join)
    run(ST_WGFINAL2, s2, [&] { hipLaunchKernelGGL(wgrad_final_kernel, dim3((20 * 20 * 9 + 63) / 64, kTowers, 1), dim3(256), 0, s2, fa); });
    join();                                   // every gradient is in the arena before the caller's stream goes on
    if (fused && !adam_early) {
        // what is left of the update: the conv2..conv5 weights (16 % of the arena), whose gradients the two reductions above
        // have just finished - one coalesced launch of 12 ranges
        AdamSegs sc{};
        for (int t = 0; t < kTowers; ++t)
            for (int q = 1; q < kConvs; ++q) { sc.off[t * 4 + q - 1] = T0.convW[q] + t * kTowerFloats; sc.len[t * 4 + q - 1] = kCout[q] * kCin[q] * 9; }
        run(ST_ADAM3, st, [&] { hipLaunchKernelGGL(adam_seg_kernel, dim3(22, 12), dim3(256), 0, st, P, G, h->adam_m, h->adam_v, sc, ak); });
    }
    if (hipError_t le = hipGetLastError(); le != hipSuccess) {
        w->tickets_dirty = true;
        set_error("a launch of the training step failed: %s", hipGetErrorString(le));
        return SCS_ERR_HIP;
    }
    h->packed_dirty = true;          // BN running statistics moved
    if (!h->last_train_event) SCS_HIP(hipEventCreateWithFlags(&h->last_train_event, hipEventDisableTiming));
    SCS_HIP(hipEventRecord(h->last_train_event, st));
    if (w->prof) {
        SCS_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < ST_COUNT; ++i) {
            stage_ms[i] = 0.f;
            if (w->pused[i]) SCS_HIP(hipEventElapsedTime(&stage_ms[i], w->pev[2 * i], w->pev[2 * i + 1]));
        }
        w->prof = false;
    }
    if (loss_host) {
        SCS_HIP(hipMemcpyAsync(loss_host, w->loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
        SCS_HIP(hipStreamSynchronize(st));
    }
    return SCS_OK;
}

__global__ void gather_labels_kernel(const uint8_t* __restrict__ lab, const scs_volume* __restrict__ descs, int n_vols,
                                     const int32_t* __restrict__ samples, long n, uint8_t* __restrict__ y) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int4 s = reinterpret_cast<const int4*>(samples)[e];
        const scs_volume d = descs[max(0, min(s.x, n_vols - 1))];
        const bool ok = (unsigned)s.y < (unsigned)d.dim_x && (unsigned)s.z < (unsigned)d.dim_y && (unsigned)s.w < (unsigned)d.dim_z;
        const uint8_t v = ok ? lab[d.offset + ((size_t)s.y * d.dim_y + s.z) * d.dim_z + s.w] : 0;
        y[e] = v == 15 ? 0 : v;                                      // boundary negatives -> background (base.py:88-89)
    }
}

}  // namespace scs

using namespace scs;

extern "C" {

int scs_train_stage_count(void) { return ST_COUNT; }
const char* scs_train_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : ""; }

int scs_train_step(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, const uint8_t* y,
                   int64_t n64, uint64_t dropout_seed, const uint8_t* masks_dev, float* loss_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n64 > 0 && n64 <= kMaxTrainBatch, "n out of range (1..65536)");
    SCS_ARG(ax && co && sa && atlas && y, "null input pointer");
    X0Src x0{};
    x0.views[0] = ax; x0.views[1] = co; x0.views[2] = sa;
    return train_step_impl(h, x0, atlas, y, nullptr, (int)n64, dropout_seed, masks_dev, loss_host, (hipStream_t)stream, nullptr);
}

int scs_train_step_profiled(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, const uint8_t* y,
                            int64_t n64, uint64_t dropout_seed, float* stage_ms, void* stream) {
    SCS_ARG(h != nullptr && stage_ms != nullptr, "null handle or output");
    SCS_ARG(n64 > 0 && n64 <= kMaxTrainBatch, "n out of range (1..65536)");
    SCS_ARG(ax && co && sa && atlas && y, "null input pointer");
    X0Src x0{};
    x0.views[0] = ax; x0.views[1] = co; x0.views[2] = sa;
    return train_step_impl(h, x0, atlas, y, nullptr, (int)n64, dropout_seed, nullptr, nullptr, (hipStream_t)stream, stage_ms);
}

int scs_train_step_volume(scs_handle h, const float* vols, const scs_volume* descs, int32_t n_vols, const int32_t* samples,
                          int64_t n_samples, const float* atlas_rows, const uint8_t* y, const int32_t* index, const uint8_t* flips,
                          int64_t n64, uint64_t dropout_seed, const uint8_t* masks_dev, float* loss_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n64 > 0 && n64 <= kMaxTrainBatch, "n out of range (1..65536)");
    SCS_ARG(vols && descs && samples && atlas_rows && y, "null input pointer");
    SCS_ARG(n_vols > 0 && n_samples > 0 && n_samples <= 0x7fffffff, "empty volume / sample table");
    SCS_ARG(index || n64 <= n_samples, "more samples asked for than the table holds");
    X0Src x0{};
    x0.vols = vols; x0.descs = descs; x0.samples = samples; x0.index = index; x0.n_samples = (int)n_samples; x0.n_vols = n_vols;
    x0.flips = flips;
    return train_step_impl(h, x0, atlas_rows, y, index, (int)n64, dropout_seed, masks_dev, loss_host, (hipStream_t)stream, nullptr);
}

int scs_gather_labels(scs_handle h, const uint8_t* label_vols, const scs_volume* descs, int32_t n_vols, const int32_t* samples,
                      int64_t n, uint8_t* y, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0 && n_vols > 0, "n < 0 or no volumes");
    if (n == 0) return SCS_OK;
    SCS_ARG(label_vols && descs && samples && y, "null pointer");
    SCS_HIP(hipSetDevice(h->device));
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(gather_labels_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, label_vols,
                       descs, n_vols, samples, (long)n, y);
    SCS_HIP(hipGetLastError());
    return SCS_OK;
}

int scs_get_train_buffer(scs_handle h, int which, float* host_out, size_t count) {
    SCS_ARG(h && host_out, "null pointer");
    if (!h->tws || !h->tws->base) { set_error("no training step has run"); return SCS_ERR_STATE; }
    TrainWs* w = h->tws;
    const float* src = nullptr;
    size_t ld = 0, rows = (size_t)w->cap;
    if (which >= 100 && which < 300) {
        // 100 + 10 t + i: raw output of conv block i (2..5) of tower t, [n][Cout_i][Ho][Ho];
        // 200 + 10 t + i: that block's batch-norm (scale, shift) pairs as applied in the step, float2 [Cout_i] (i = 1..5; block 1:
        //                 the folded conv1 table)
        const int t = (which % 100) / 10, i = which % 10;
        SCS_ARG(t >= 0 && t < kTowers, "unknown tower");
        if (which < 200) {
            SCS_ARG(i >= 2 && i <= kConvs, "raw outputs exist for conv blocks 2..5");
            src = w->Z[t][i - 1]; ld = (size_t)kCout[i - 1] * kHout[i - 1] * kHout[i - 1];
        } else {
            SCS_ARG(i >= 1 && i <= kConvs, "unknown conv block");
            SCS_ARG(count == 2 * (size_t)kCout[i - 1], "count must be 2 * Cout");
            SCS_HIP(hipSetDevice(h->device));
            SCS_HIP(hipDeviceSynchronize());
            if (i == 1) {                                    // l1p: [20][12] rows of (9 taps, sc, sh, pad)
                float tab[240];
                SCS_HIP(hipMemcpy(tab, w->l1p[t], sizeof(tab), hipMemcpyDeviceToHost));
                for (int c = 0; c < 20; ++c) { host_out[2 * c] = tab[c * 12 + 9]; host_out[2 * c + 1] = tab[c * 12 + 10]; }
            } else {
                SCS_HIP(hipMemcpy(host_out, w->ss[t][i - 1], count * sizeof(float), hipMemcpyDeviceToHost));
            }
            return SCS_OK;
        }
    } else {
        switch (which) {
            case 0: src = w->dR2; ld = kFC2ld; break;     // [n][272]
            case 1: src = w->dcat; ld = kFC1; break;      // [n][540]
            case 2: src = w->dfeat; ld = kFlat; break;    // [n][540]
            case 3: src = w->featd; ld = kFlat; break;    // [n][540]
            case 4: src = w->R1; ld = kFC1; break;        // [n][540]
            case 5: src = w->cat2; ld = kFC1ld; break;    // [n][556]
            case 6: src = w->dlog; ld = 16; break;        // [n][16]
            case 7: src = w->R2; ld = kFC2ld; break;      // [n][272]
            case 8: src = w->feat; ld = kFlat; break;     // [n][540]
            default: SCS_ARG(false, "unknown buffer id");
        }
    }
    SCS_ARG(count <= rows * ld, "count exceeds the buffer");
    SCS_HIP(hipSetDevice(h->device));
    SCS_HIP(hipDeviceSynchronize());
    SCS_HIP(hipMemcpy(host_out, src, count * sizeof(float), hipMemcpyDeviceToHost));
    return SCS_OK;
}

int scs_wait_head_grads(scs_handle h, void* stream, size_t* first_float, size_t* n_floats) {
    SCS_ARG(h != nullptr, "null handle");
    if (!h->tws || !h->tws->head_recorded) { set_error("scs_wait_head_grads before scs_train_step"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    SCS_HIP(hipStreamWaitEvent((hipStream_t)stream, h->tws->ev_head, 0));
    constexpr HeadOff H = head_off();
    if (first_float) *first_float = (size_t)H.fc1W;
    if (n_floats) *n_floats = (size_t)(H.end - H.fc1W);
    return SCS_OK;
}

int scs_train_loss(scs_handle h, float** dev_ptr) {
    SCS_ARG(h && dev_ptr, "null pointer");
    if (!h->tws || !h->tws->loss_dev) { set_error("scs_train_loss before scs_train_step"); return SCS_ERR_STATE; }
    *dev_ptr = h->tws->loss_dev;
    return SCS_OK;
}

int scs_set_fused_adam(scs_handle h, int enable, float lr, float beta1, float beta2, float eps) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(!enable || (lr > 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f), "Adam coefficients out of range");
    h->fused_adam = enable != 0;
    if (enable) { h->fa_lr = lr; h->fa_b1 = beta1; h->fa_b2 = beta2; h->fa_eps = eps; }
    return SCS_OK;
}

int scs_adam_apply(scs_handle h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    if (!h->has_params) { set_error("scs_adam_apply before scs_set_params"); return SCS_ERR_STATE; }
    if (h->fused_adam) { set_error("scs_adam_apply on a handle with the fused update on (scs_set_fused_adam): the step applied it"); return SCS_ERR_STATE; }
    SCS_HIP(hipSetDevice(h->device));
    h->adam_t += 1;
    const double t = (double)h->adam_t;
    const float a_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
    hipLaunchKernelGGL(adam_kernel, dim3(864), dim3(256), 0, (hipStream_t)stream, h->params, h->grads, h->adam_m, h->adam_v,
                       SCS_N_PARAMS, a_t, beta1, beta2, eps, grad_scale);
    SCS_HIP(hipGetLastError());
    h->packed_dirty = true;
    if (!h->last_train_event) SCS_HIP(hipEventCreateWithFlags(&h->last_train_event, hipEventDisableTiming));
    SCS_HIP(hipEventRecord(h->last_train_event, (hipStream_t)stream));
    return SCS_OK;
}

static int eval_impl(scs_handle h, const float* ax, const float* co, const float* sa, const X0Src* x0, const float* atlas, const uint8_t* y,
                     const int32_t* index, int64_t n, double* loss_sum_host, int64_t* n_correct_host, void* stream) {
    *loss_sum_host = 0.0; *n_correct_host = 0;
    if (n == 0) return SCS_OK;
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
    if (w->eval_cap < (size_t)estep) {
        SCS_HIP(hipDeviceSynchronize());
        if (w->eval_rows) { SCS_HIP(hipFree(w->eval_rows)); w->eval_rows = nullptr; w->eval_cap = 0; }
        SCS_HIP(hipMalloc(&w->eval_rows, (size_t)estep * 2 * sizeof(float)));
        w->eval_cap = (size_t)estep;
    }
    SCS_HIP(hipMemsetAsync(w->eval_acc, 0, 2 * sizeof(double), st));
    constexpr HeadOff H = head_off();
    for (int64_t s = 0; s < n; s += estep) {
        const int m = (int)((n - s < estep) ? (n - s) : estep);
        if (x0) {
            X0Src xs = *x0;                                     // this chunk's rows: index + s, or samples s.. when there is no index
            if (index) xs.index = index + s; else { xs.samples += 4 * s; xs.n_samples -= (int)s; }
            rc = forward_chunk(h, nullptr, nullptr, nullptr, index ? atlas : atlas + s * 15, nullptr, 0, 0, 0, nullptr, nullptr, m, nullptr,
                               nullptr, st, nullptr, &xs, index ? index + s : nullptr);
        } else {
            rc = forward_chunk(h, ax + s * 1024, co + s * 1024, sa + s * 1024, atlas + s * 15, nullptr, 0, 0, 0, nullptr, nullptr, m,
                               nullptr, nullptr, st, nullptr);
        }
        if (rc != SCS_OK) return rc;
        // loss / hit per row from the fc_2 activations of this chunk, then one ordered sum
        hipLaunchKernelGGL(softmax_ce_kernel, dim3((m + 255) / 256), dim3(256), 0, st, h->a_fc2, h->params + H.outW, h->params + H.outb,
                           index ? y : y + s, index ? index + s : (const int32_t*)nullptr, x0 ? x0->n_samples : 0, m, w->eval_rows);
        hipLaunchKernelGGL(eval_sum_kernel, dim3(1), dim3(256), 0, st, w->eval_rows, m, w->eval_acc);
    }
    double acc[2];
    SCS_HIP(hipMemcpyAsync(acc, w->eval_acc, sizeof(acc), hipMemcpyDeviceToHost, st));
    SCS_HIP(hipStreamSynchronize(st));
    *loss_sum_host = acc[0];
    *n_correct_host = (int64_t)(acc[1] + 0.5);
    return SCS_OK;
}

int scs_eval_batch(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas, const uint8_t* y,
                   int64_t n, double* loss_sum_host, int64_t* n_correct_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    SCS_ARG(loss_sum_host && n_correct_host, "null output pointer");
    SCS_ARG(n == 0 || (ax && co && sa && atlas && y), "null input pointer");
    return eval_impl(h, ax, co, sa, nullptr, atlas, y, nullptr, n, loss_sum_host, n_correct_host, stream);
}

int scs_eval_batch_volume(scs_handle h, const float* vols, const scs_volume* descs, int32_t n_vols, const int32_t* samples,
                          int64_t n_samples, const float* atlas_rows, const uint8_t* y, const int32_t* index, int64_t n,
                          double* loss_sum_host, int64_t* n_correct_host, void* stream) {
    SCS_ARG(h != nullptr, "null handle");
    SCS_ARG(n >= 0, "n < 0");
    SCS_ARG(loss_sum_host && n_correct_host, "null output pointer");
    SCS_ARG(n == 0 || (vols && descs && samples && atlas_rows && y), "null input pointer");
    SCS_ARG(n == 0 || (n_vols > 0 && n_samples > 0 && n_samples <= 0x7fffffff), "empty volume / sample table");
    SCS_ARG(index || n <= n_samples, "more samples asked for than the table holds");
    X0Src x0{};
    x0.vols = vols; x0.descs = descs; x0.samples = samples; x0.index = index; x0.n_samples = (int)n_samples; x0.n_vols = n_vols;
    return eval_impl(h, nullptr, nullptr, nullptr, &x0, atlas_rows, y, index, n, loss_sum_host, n_correct_host, stream);
}

}  // extern "C"
