// Per-instance state of libscs (opaque to callers: include/scs.h).
#pragma once
#include "conv_mfma.h"

struct TrainWs;

struct scs_ctx {
    int device = 0;
    bool has_params = false;
    bool packed_dirty = false;     // parameter arena changed since the inference operands were packed

    // Lasagne-layout arenas (SCS_N_PARAMS floats each)
    float* params = nullptr;
    float* grads = nullptr;
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    long adam_t = 0;

    // packed inference operands
    float* packed = nullptr;
    size_t packed_floats = 0;
    scs::PackedNet pk{};

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
                  const X0Src* x0 = nullptr, const int32_t* atlas_index = nullptr);
int forward_head(scs_ctx* h, const float* const c5[kTowers], const float* atlas, const float* vol, int dx, int dy, int dz,
                 const int32_t* centers, const float* atlas4d, int n, float* probs, int32_t* labels, hipStream_t st, hipEvent_t* ev,
                 int atlas_rows, const int32_t* atlas_index);
void train_free(scs_ctx* h);
}
