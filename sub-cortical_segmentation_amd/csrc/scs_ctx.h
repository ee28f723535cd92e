// Per-instance state of libscs (opaque to callers: include/scs.h).
#pragma once
#include "conv_mfma.h"
#include "conv_split.h"
#include "dense_split.h"

struct TrainWs;

// Behaviour switches of the training step, per handle: read ONCE from the environment when the handle is created (SCS_* below),
// changed afterwards only through scs_set_option (include/scs.h).  -1 = the library's default bound.
struct scs_options {
    int fin_max_batch = -1;          // SCS_FIN_MAX_BATCH: batch-norm finalize inside the producing conv kernel up to this batch
    int bn1_fused_max_batch = -1;    // SCS_BN1_FUSED_MAX_BATCH: conv1's statistics finalized inside the prep + moments launch
    int dense_chain_max_batch = -1;  // SCS_DENSE_CHAIN_MAX_BATCH: the dense head as ONE launch up to this batch
    int c5_fused_max_batch = -1;     // SCS_C5_FUSED_MAX_BATCH: the conv5 block's backward as one launch
    int train_one_stream = 0;        // SCS_TRAIN_ONE_STREAM: every launch in order on the caller's stream
    int tail_side = 0;               // SCS_TAIL_SIDE: the round-2 order of the step's tail
    int wg_slices[4] = {0, 0, 0, 0}; // SCS_WG_SLICES="b,c,d,e": batch slices of the conv2..5 weight-gradient kernels (0 = default)
    int tail_fused = -1;             // SCS_TAIL_FUSED: bn5 / bn5-backward / bn2 folded into their neighbours (-1 = default on)
    int split_max_batch = -1;        // SCS_SPLIT_MAX_BATCH: channel-split conv tilings up to this batch (-1 = default)
    int c5_split = -1;               // SCS_C5_SPLIT: largest batch at which the conv5 block's backward is two launches (data path / weight-
                                     // gradient partial on the side stream); -1 = default (128), 0 = always one launch
    int wgfin_early = 1;             // SCS_WGFIN_EARLY: 0 = the conv3..5 partial-sum reduction at the END of the side stream (the round-3 order)
    int fwd_fused = -1;              // SCS_FWD_FUSED: 1 = inference conv3 -> conv4 + pool2 -> conv5 as one kernel per patch-tower, 2 = that kernel
                                     // storing the two intermediates too (scs_get_activation); default (-1, 0): three launches - they are faster
    int fwd_split = -1;              // SCS_FWD_SPLIT: 1 = inference convolutions on the f16 matrix cores with every f32 operand as two f16 pieces
                                     // (conv_split.h: f32-grade results, ~2x the f32-MFMA rate); scs_forward / scs_forward_profiled on patch inputs;
                                     // the call then returns after the work has finished (it reads the overflow flag and, if a scaled activation
                                     // left the f16 range, repeats the batch on the f32 kernels).  Default (-1, 0): the exact-f32 MFMA kernels
    int split_fallbacks = 0;         // (counter, read with scs_get_option: the device word sp_flag[2]) chunks repeated on the f32 kernels after such an overflow
    int split_head = -1;             // SCS_SPLIT_HEAD: with fwd_split, 0 = the dense head on the f32-MFMA GEMMs (round-5's first form); default
                                     // (-1, 1): d1 x3, FC1, fc_2 on the split kernels of dense_split.h as well
    int train_split = -1;            // SCS_TRAIN_SPLIT: 1 = the training forward's conv1 + conv2 block on the f16 matrix cores with split operands
                                     // (conv_split_train.h: f32-grade results; a scaled activation beyond the f16 range turns the step's loss into NaN);
                                     // default (-1, 0): the exact-f32 MFMA kernel
};

struct scs_ctx {
    int device = 0;
    scs_options opt{};
    int chain_placement = 0;        // dense_chain.h: 0 = not probed yet, 1 = every XCD receives grid / 8 workgroups, -1 = it does not
    bool has_params = false;
    bool packed_dirty = false;     // parameter arena changed since the inference operands were packed

    // Lasagne-layout arenas (SCS_N_PARAMS floats each)
    float* params = nullptr;
    float* grads = nullptr;
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    long adam_t = 0;
    // scs_set_fused_adam: the update is applied INSIDE every training step (single-rank training: nothing sits between the
    // gradients and the update), layer group by layer group as the gradients become final
    bool fused_adam = false;
    float fa_lr = 1e-3f, fa_b1 = 0.9f, fa_b2 = 0.999f, fa_eps = 1e-8f;

    // packed inference operands
    float* packed = nullptr;
    size_t packed_floats = 0;
    scs::PackedNet pk{};

    // split (f16 x 2 pieces) inference operands: conv_split.h
    unsigned char* sp_pack = nullptr;
    struct SplitPacked {
        const float* w1[scs::kTowers];                        // conv1: kSp12W1Floats
        const uint4* w[scs::kTowers][scs::kConvs];            // i = 1..4: [k-step][channel tile][piece][lane] x 16 bytes
        const float* bias[scs::kTowers][scs::kConvs];         // i = 1..4: [tiles * 16] folded bias + the accumulator scale
        const uint4* dw[5];                                   // dense_split.h: d1 of the three towers, FC1, fc_2
        const float* db[5];                                   // [tiles * 16] bias + the accumulator scale
    } sp{};
    // activation rows of the split dense head (dense_split.h: rows of 4 * kp + 16 bytes, cleared at allocation, written by the split
    // kernels alone: pads and K padding stay zero)
    unsigned char* r_c5[scs::kTowers]{};      // [chunk] rows of 544 halves per piece: the towers' flattened block-5 outputs
    unsigned char* r_feat = nullptr;          // [chunk] rows of 544: d1 x3 side by side (FC1's input)
    unsigned char* r_fc1 = nullptr;           // [chunk] rows of 576: FC1 output + atlas prior (fc_2's input)
    unsigned char* r_base = nullptr;          // (allocated by the first chunk that takes the split-head branch: scs_api.hip, ensure_split_rows)
    int r_chunk = 0;                          // chunk size those rows were allocated for
    unsigned int* sp_flag = nullptr;           // device words: [0] overflow flag of the split kernels of the chunk in flight (the guard of
                                               // the f32 repeat behind them), [1] whether the LAST chunk was repeated (its buffers then hold the f32
                                               // kernels' layouts), [2] chunks repeated so far (option split_fallbacks)
    unsigned int* sp_flag_host = nullptr;      // pinned host copy of those words (read behind a synchronisation only)
    bool last_split = false;                   // the last forward left split planes in a_pool1 / a_conv3 / a_pool2
    bool last_split_head = false;              // ... and rows of halves in r_c5 / r_feat / r_fc1 (the split dense head)

    // forward workspace (one chunk)
    int chunk = 32768;             // patches per internal chunk (workspace ~140 KB per patch)
    int ws_chunk = 0;              // chunk the workspace was allocated for
    float* ws = nullptr;
    float* a_pool1[scs::kTowers]{};   // [chunk][20][14][14]
    float* a_conv3[scs::kTowers]{};   // [chunk][40][12][12]
    float* a_pool2[scs::kTowers]{};   // [chunk][40][5][5]
    float* a_conv5[scs::kTowers]{};   // [chunk][540]
    float* a_feat = nullptr;          // [chunk][540]
    float* a_fc1 = nullptr;           // [chunk][556]
    float* a_fc2 = nullptr;           // [chunk][272]
    float* a_logits = nullptr;        // [chunk][16]
    int last_n = 0;

    // shared-tower evaluation of a whole subject (dense_eval.h): tile buffers of one slice batch, block-5 features of all voxels
    float* dense_ws = nullptr; size_t dense_ws_floats = 0;
    float* dense_feat = nullptr; size_t dense_feat_floats = 0;
    unsigned int* box_flag = nullptr;          // scs_forward_volume_shared: centres outside the caller's box (device counter ...
    unsigned int* box_flag_host = nullptr;     // ... and its pinned host copy)

    // host staging for the *_host entry points
    float* stage_dev = nullptr;       // [chunk] x (3*1024 + 15 + 15) floats + labels
    size_t stage_floats = 0;
    hipStream_t host_compute = nullptr, host_copy = nullptr;    // scs_forward_host: kernels / uploads of the next chunk

    // training state (scs_train.hip)
    TrainWs* tws = nullptr;
    hipEvent_t last_train_event = nullptr;   // recorded behind the last scs_train_step / scs_adam_apply work: entry points that
                                             // repack on a stream of their own (scs_forward_host) wait for it first
};

namespace scs {
int ensure_workspace(scs_ctx* h, int need);
int repack_inference(scs_ctx* h, hipStream_t st);
int forward_chunk(scs_ctx* h, const float* ax, const float* co, const float* sa, const float* atlas,
                  const float* vol, int dx, int dy, int dz, const int32_t* centers, const float* atlas4d,
                  int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev,
                  const X0Src* x0 = nullptr, const int32_t* atlas_index = nullptr, int mode = 0);
int forward_head(scs_ctx* h, const float* const c5[kTowers], const float* atlas, const float* vol, int dx, int dy, int dz,
                 const int32_t* centers, const float* atlas4d, int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev,
                 int atlas_rows, const int32_t* atlas_index, const unsigned int* guard = nullptr);
// forward_chunk modes: the f32 kernels; the split kernels (option fwd_split); the f32 kernels behind the device-side guard
// h->sp_flag (they run only if a split kernel of the same chunk raised the overflow flag: scs_api.hip, forward_chunk_split)
enum { kFwdF32 = 0, kFwdSplit = 1, kFwdF32Guarded = 2 };
void train_free(scs_ctx* h);
}
