/*
 * scs.h - C ABI of libscs.so, the MI355X (gfx950) implementation of the tri-planar voxelwise CNN of
 * sergivalverde/sub-cortical_segmentation.
 *
 * The reference has no C/FFI boundary: its boundary is the Python object protocol of the nolearn `net`
 * object plus three functions of cnn_cort/base.py.  This library is what the py3
 * `cnn_cort` package in this repo binds (ctypes) to provide that protocol; each entry point names the
 * reference call it replaces.  Plain pointers and sizes only - no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative scs_status on failure; scs_last_error() gives text.
 *   - "dev" pointers are HIP device pointers on the handle's device; "host" pointers are ordinary memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue work;
 *     the caller synchronises the stream (the *_host variants synchronise before returning).
 *   - tensors are C-contiguous float32: views [n,1,32,32], atlas [n,15], probs [n,15]; labels int32 [n].
 *   - parameters travel as ONE flat float32 arena in the reference's own order ("Lasagne layout"): the 88
 *     layers of nets.py:170-231 in graph order, per layer the arrays of get_params() (conv W; BN beta,
 *     gamma, mean, inv_std; PReLU alpha; dense W (in,out), b) - 883 455 floats.  Nothing is flipped or
 *     folded in the arena; the library derives its packed MFMA operand layouts from it on the device.
 */
#ifndef SCS_H
#define SCS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCS_N_PARAMS 883455          /* floats in the Lasagne-layout arena (the shipped miccai2012_v1 weights) */
#define SCS_N_CLASSES 15
#define SCS_PATCH 32

typedef struct scs_ctx* scs_handle;

/* One z-scored volume inside a concatenated device buffer of volumes (gather-fused training, scs_train_step_volume):
 * `offset` floats from the start of the buffer, C order [dim_x][dim_y][dim_z]. */
typedef struct {
    int64_t offset;
    int32_t dim_x, dim_y, dim_z;
    int32_t reserved;
} scs_volume;

typedef enum {
    SCS_OK = 0,
    SCS_ERR_ARG = -1,        /* bad argument (null pointer, wrong size, n < 0 ...) */
    SCS_ERR_HIP = -2,        /* a HIP runtime call failed (no device, launch failure, out of memory) */
    SCS_ERR_STATE = -3       /* call order (e.g. forward before scs_set_params) */
} scs_status;

/* Text of the last error raised on the calling thread ("" if none). */
const char* scs_last_error(void);

/* Library/ABI version, and the gfx target the kernels were compiled for ("gfx950"). */
int scs_abi_version(void);
const char* scs_target_arch(void);

/* Create / destroy a net instance bound to HIP device `device`.
 * Replaces: nolearn NeuralNet(...) construction in build_model (nets.py:233-246). */
int scs_create(int device, scs_handle* out);
int scs_destroy(scs_handle h);

/* Upload / download all parameters (host pointers, Lasagne-layout arena, n must be SCS_N_PARAMS).
 * Replaces: net.load_params_from / net.save_params_to (nets.py:251, SaveWeights at nets.py:155).
 * scs_set_params also (re)builds the inference operands (BN folded into the conv weights, filters flipped
 * to correlation form, MFMA fragment order) and resets the Adam state. */
int scs_set_params(scs_handle h, const float* host_arena, size_t n);
int scs_get_params(scs_handle h, float* host_arena, size_t n);

/* Upper bound of the batch chunk used internally by the forward entry points (default 32768 patches; workspace
 * ~140 KB per patch is allocated for one chunk; a call splits its n patches into equal chunks below this bound).
 * Returns the value in effect (pass 0 to query). */
int scs_set_chunk(scs_handle h, int patches_per_chunk);

/* Behaviour switches of the training step, per handle.  Each is initialised ONCE, at scs_create, from the environment variable
 * named below (tuning scripts), and changed afterwards only here - the step never reads the environment.  value -1 = the
 * library's default.  Names:
 *   "fin_max_batch"          SCS_FIN_MAX_BATCH          batch-norm finalize inside the producing conv kernel up to this batch
 *   "bn1_fused_max_batch"    SCS_BN1_FUSED_MAX_BATCH    conv1's statistics finalized inside the step's first launch up to this batch
 *   "dense_chain_max_batch"  SCS_DENSE_CHAIN_MAX_BATCH  the dense head as one launch up to this batch (at most 384; only on devices
 *                                                       where a probe launch finds 8 XCDs with equal shares of a grid)
 *   "c5_fused_max_batch"     SCS_C5_FUSED_MAX_BATCH     the conv5 block's backward as one launch up to this batch
 *   "tail_fused"             SCS_TAIL_FUSED             0: bn5 / bn5-backward as launches of their own (the round-3 step)
 *   "split_max_batch"        SCS_SPLIT_MAX_BATCH        channel-split conv tilings (two workgroups per patch-tower) up to this batch
 *   "fwd_fused"              SCS_FWD_FUSED              inference: 1 = conv3 -> conv4 + pool2 -> conv5 as ONE kernel per patch-tower (the activations
 *                                                       between them stay in LDS: 113 instead of 280 KB of HBM traffic per patch, same bits, 20 %
 *                                                       slower), 2 = that kernel also storing them (scs_get_activation); default 0: three launches
 *   "c5_split"               SCS_C5_SPLIT               largest batch at which the conv5 block's backward is two launches (data path on the caller's
 *                                                       stream, weight-gradient partial on the side stream); default 128, 0 = always one launch
 *   "wgfin_early"            SCS_WGFIN_EARLY            0: the conv3..5 partial-sum reduction at the end of the side stream (the round-3 order)
 *   "train_one_stream"       SCS_TRAIN_ONE_STREAM       1: every launch in order on the caller's stream
 *   "tail_side"              SCS_TAIL_SIDE              1: the round-2 stream order of the step's last launches
 *   "wg_slices_conv2".."wg_slices_conv5"  SCS_WG_SLICES="b,c,d,e"  batch slices of the weight-gradient kernels (0 = default)
 *   "fwd_split"              SCS_FWD_SPLIT              inference: 1 = conv2..conv5 and the dense head on the f16 matrix cores with every f32 operand
 *                                                       as two f16 pieces, f32 accumulation (f32-grade results at ~2.4x the rate).  scs_forward,
 *                                                       scs_forward_host, scs_forward_volume / _rows, scs_forward_profiled.  The entries stay
 *                                                       asynchronous: behind the split kernels of every internal chunk the f32 kernels of the same
 *                                                       chunk are enqueued in a guarded form - each of their workgroups reads the overflow flag the
 *                                                       split kernels raise when a scaled activation leaves the f16 range and exits at once while it
 *                                                       is clear - so a flagged chunk comes out of the exact-f32 kernels (the bits the call gives
 *                                                       without the option) with no host round trip.  "split_fallbacks": chunks repeated so far (a
 *                                                       device counter: scs_get_option waits for the device to read it; setting it resets it).
 *                                                       (scs_forward_volume_shared keeps the f32 kernels.)  Default 0
 *   "split_head"             SCS_SPLIT_HEAD             0: with fwd_split, the dense head stays on the f32 GEMMs.  Default 1
 *   "train_split"            SCS_TRAIN_SPLIT            training: 1 = the forward pass of the conv1 + conv2 block on the split kernels; a scaled
 *                                                       activation beyond the f16 range makes that step's loss NaN - and its gradients: there is
 *                                                       no f32 repeat of a training step, an update applied from them (scs_adam_apply, or the
 *                                                       fused update inside the step) leaves the WEIGHTS INVALID.  A caller that uses the option
 *                                                       checks the loss before it keeps the weights (Net.fit raises at the end of the epoch);
 *                                                       the batch norm in front bounds the conv1 output, so this takes inputs far outside a
 *                                                       z-scored volume's range.  Default 0
 * Unknown name or value outside [-1, 2^20]: SCS_ERR_ARG. */
int scs_set_option(scs_handle h, const char* name, int64_t value);
int scs_get_option(scs_handle h, const char* name, int64_t* value);

/* Measurement aid (bench.py): the rate at which THIS device issues back-to-back v_mfma_f32_16x16x4_f32 with independent accumulators
 * and register operands - four waves per SIMD on every CU, three launches of ~4 ms, the last one timed - as TFLOP/s.  The roofline's 157.3 TFLOP/s prices the instruction
 * at 32 cycles of a 2.4 GHz clock; what the part sustains under an f32-MFMA load is lower, and this is that figure, measured in the
 * same process as the kernels it is set beside.  No reference counterpart.  Synchronises the device. */
int scs_probe_f32_mfma(scs_handle h, double* tflops);
/* The same probe, and the SHADER CLOCK the device held while it ran (GHz; median over the workgroups of d s_memtime / d s_memrealtime
 * x 100 MHz around the loop), so that a reader can tell a clock difference between devices or runs (126-139 TFLOP/s seen on this pool)
 * from an issue-rate difference: rate / (CUs x 4 SIMDs x 2048 FLOP / 32 cycles x clock) is the share of the pipe the loop reaches. */
int scs_probe_f32_mfma_clock(scs_handle h, double* tflops, double* shader_ghz);

/* Deterministic forward on n patches (device pointers).  probs and/or labels may be NULL.
 * labels[i] = argmax_c probs[i][c] (first maximum), i.e. nolearn's predict = argmax(predict_proba).
 * Replaces: net.predict / net.predict_proba (base.py:425-428, 435-438) - one pass yields both. */
int scs_forward(scs_handle h, const float* ax_dev, const float* co_dev, const float* sa_dev,
                const float* atlas_dev, int64_t n, float* probs_dev, int32_t* labels_dev, void* stream);

/* Same, host pointers: uploads chunk by chunk, synchronises before returning. */
int scs_forward_host(scs_handle h, const float* ax, const float* co, const float* sa, const float* atlas,
                     int64_t n, float* probs, int32_t* labels);

/* Gather-fused inference: patches are cut on the fly from a z-scored volume resident on the device.
 *   vol_dev     float32 [X][Y][Z] (C order), already normalised (base.py:358)
 *   centers_dev int32 [n][3] voxel coordinates (x,y,z); patches are zero padded outside the volume and the
 *               centre lands at patch index [16][16] (base.py:293-306)
 *   atlas4d_dev float32 [X][Y][Z][15]; rows that sum to 0 become background, channel 14 = 1 (base.py:391-394);
 *               a centre outside the volume (the Python layer rejects those) reads an all-zero prior, never memory
 * Replaces: the host loop load_patch_batch + get_patches (base.py:380-397, 272-308) feeding net.predict. */
int scs_forward_volume(scs_handle h, const float* vol_dev, int dim_x, int dim_y, int dim_z,
                       const int32_t* centers_dev, const float* atlas4d_dev, int64_t n,
                       float* probs_dev, int32_t* labels_dev, void* stream);

/* Same patch gather, but the priors arrive as rows: atlas_rows_dev float32 [n][15], one row per centre, ALREADY carrying
 * the background rule (what load_patch_batch yields as x_atlas, base.py:388-394).  For hosts that keep the 4-D prior in
 * the file's own (Fortran) order and gather the n rows themselves instead of shipping X*Y*Z*15 floats to the device. */
int scs_forward_volume_rows(scs_handle h, const float* vol_dev, int dim_x, int dim_y, int dim_z,
                            const int32_t* centers_dev, const float* atlas_rows_dev, int64_t n,
                            float* probs_dev, int32_t* labels_dev, void* stream);

/* `iterations` steps of binary dilation of a uint8 0/1 volume [X][Y][Z] with the 6-neighbour cross, border value 0, in
 * place (scratch_dev: X*Y*Z bytes).  Non-zero input voxels count as set; the result is 0/1.
 * Replaces: scipy.ndimage.binary_dilation(mask, iterations=10) growing the registered sub-cortical mask into the ROI
 * that test_scan classifies (base.py:369). */
int scs_dilate_mask(scs_handle h, uint8_t* mask_dev, uint8_t* scratch_dev, int dim_x, int dim_y, int dim_z, int iterations,
                    void* stream);

/* ---- whole-volume preparation on the device (test_scan, base.py:401-458): the raw T1 and the registered mask go up as they
 * lie in their files and everything the classifier needs from them is derived there. ------------------------------------ */

/* Element types of a raw volume. */
#define SCS_U8 0
#define SCS_I16 1
#define SCS_U16 2
#define SCS_I32 3
#define SCS_F32 4
#define SCS_F64 5

/* out[x][y][z] (float32, C order) = (v - mean) / std, mean and population std over the NON-ZERO voxels of the raw volume,
 * evaluated in double (two passes: mean, then the squared deviations about it) and rounded once - the inference normalisation
 * `(image - image[nz].mean()) / image[nz].std()` of load_patch_batch (base.py:357-358) followed by the float32 cast of the
 * patches (base.py:380-386).  Deviation: for a FLOAT32 T1 the reference evaluates that formula in float32 throughout (NumPy
 * pairwise sums); the result here differs from it by a few float32 ulps (bounded in tests/test_gpu_scan.py).
 *   raw_dev        the voxels as stored (dtype: SCS_U8 .. SCS_F64), in the file's Fortran order (x fastest; fortran_order = 1)
 *                  or in C order (z fastest; 0)
 *   scratch_dev    3 * 1024 + 3 doubles: per-workgroup partial sums in a fixed order (the result is deterministic), then
 *                  (mean, std, non-zero count), which the caller may read back
 * Replaces: the whole-volume NumPy passes of base.py:357-358 (about 0.15 CPU-seconds per 256x256x192 subject). */
int scs_zscore_volume(scs_handle h, const void* raw_dev, int dtype, int dim_x, int dim_y, int dim_z, int fortran_order,
                      float* out_dev, double* scratch_dev, void* stream);

/* The ROI of test_scan: mask != 0, dilated `iterations` times with the 6-neighbour cross (scs_dilate_mask), and the number of
 * its voxels.  mask_dev (X*Y*Z bytes, C order) receives the dilated 0/1 mask; scratch_dev: X*Y*Z bytes; run_offsets_dev:
 * ceil(X*Y*Z / 1024) + 1 int32 - start of every run of 1024 consecutive voxels in the voxel list, and the total.
 * Synchronises the stream to return *total_host.
 * Replaces: binary_dilation + np.nonzero of base.py:369-370, 310-331. */
int scs_roi_prepare(scs_handle h, const void* mask_raw_dev, int dtype, int dim_x, int dim_y, int dim_z, int fortran_order,
                    int iterations, uint8_t* mask_dev, uint8_t* scratch_dev, int32_t* run_offsets_dev, int64_t* total_host,
                    void* stream);

/* The voxel list itself: centers_dev int32 [n_centers][3] = (x, y, z) of the set voxels of mask_dev in C order - the order of
 * np.nonzero, in which get_mask_voxels (base.py:310-331) lists them - n_centers = the total scs_roi_prepare returned. */
int scs_roi_centers(scs_handle h, const uint8_t* mask_dev, int dim_x, int dim_y, int dim_z, const int32_t* run_offsets_dev,
                    int32_t* centers_dev, int64_t n_centers, void* stream);

/* test_scan for a whole subject with the tower work SHARED between neighbouring voxels: the same inputs and the same results
 * - bit for bit - as scs_forward_volume_rows, at about 1/20 of the tower arithmetic when the centres fill their bounding box
 * densely (an ROI).  The towers are valid convolutions and two 2x2 poolings, so the block-5 output of a patch is a 3x3 window of
 * a per-slice map that depends only on the patch origin modulo 4; the maps are computed tile by tile with the kernels of the
 * patch path (csrc/dense_eval.h).  box (HOST memory): xmin, xmax, ymin, ymax, zmin, zmax of the centres - every centre must lie
 * inside it: the centres are checked against the box on the device and the call returns SCS_ERR_ARG (results invalid) when one
 * lies outside; the work grows with the box volume, not with n.  The call synchronises `stream` before it returns (it reads that
 * check back).  Scratch is grown inside the handle (at most ~0.8 GB of tiles + 6.5 KB per centre).
 * Replaces: load_patch_batch + get_patches + net.predict_proba of base.py:380-397, 272-308, 425-438 for dense centre sets. */
int scs_forward_volume_shared(scs_handle h, const float* vol_dev, int dim_x, int dim_y, int dim_z, const int32_t* centers_dev,
                              const float* atlas_rows_dev, int64_t n, const int32_t* box_host, float* probs_dev,
                              int32_t* labels_dev, void* stream);

/* Measurement hook (bench.py): scs_forward with HIP events recorded on `stream` around each stage of every
 * chunk; synchronises.  stage_ms[8] receives the summed elapsed milliseconds of
 *   0 conv1+conv2+pool1, 1 conv3, 2 conv4+pool2, 3 conv5, 4 d1 x3, 5 FC1 (+atlas concat), 6 fc_2, 7 out+softmax
 * and *launches the number of chunks (= launches per stage). */
int scs_forward_profiled(scs_handle h, const float* ax_dev, const float* co_dev, const float* sa_dev,
                         const float* atlas_dev, int64_t n, float* probs_dev, int32_t* labels_dev, void* stream,
                         float* stage_ms, int* launches);

/* Test hook: copy an intermediate activation of the LAST chunk processed by scs_forward to the host.
 * which: 0 pool1 [3][n][20][14][14], 1 conv3 [3][n][40][12][12], 2 pool2 [3][n][40][5][5],
 *        3 conv5 [3][n][540], 4 d1 concat [n][540], 5 FC1+atlas [n][556], 6 fc_2 [n][272], 7 logits [n][16]
 * n = patches in that chunk; count = floats the caller's buffer holds (must match). */
int scs_get_activation(scs_handle h, int which, float* host_out, size_t count);

/* ---- training (replaces net.fit's per-batch train_iter_/eval_iter_, nets.py:233-246 + train_model.py:57) -- */

/* One forward+backward in training mode (BN batch statistics, dropout p=0.5, mean cross-entropy) on n
 * samples (device pointers; y = uint8 class ids 0..14).  Gradients land in the gradient arena (Lasagne
 * layout, zeros at BN mean/inv_std); BN running statistics in the parameter arena are updated with
 * alpha = 0.1.  Dropout keep-masks come from `masks_dev` when non-NULL (uint8 0/1, laid out
 * [3][n][540] tower masks in (c,h,w) order, then [n][540] f1_drop, then [n][540] f2_drop), else from a
 * counter-based generator keyed by dropout_seed.  *loss_host receives the batch mean loss (the call
 * synchronises the stream to return it) unless NULL.  1 <= n <= 65536 (the step's workspace is ~0.75 MB per sample). */
int scs_train_step(scs_handle h, const float* ax_dev, const float* co_dev, const float* sa_dev,
                   const float* atlas_dev, const uint8_t* y_dev, int64_t n, uint64_t dropout_seed,
                   const uint8_t* masks_dev, float* loss_host, void* stream);

/* Gather-fused training step: the same step, but the 32x32 tri-planar patches are cut inside the kernels from z-scored
 * volumes resident on the device, so a training set is 77 bytes per sample (sample row + prior row + label) instead of
 * 12.3 KB of materialised patches.
 *   vols_dev       float32, all training volumes concatenated (each [X][Y][Z], C order, normalised as base.py:146)
 *   descs_dev      scs_volume[*] (device memory): where each volume lies in vols_dev and its dims
 *   samples_dev    int32 [N][4]: (volume, x, y, z) of every sample of the training set; the patch of a sample is cut around
 *                  its voxel exactly as get_patches does (base.py:293-306; zero padded outside the volume)
 *   atlas_rows_dev float32 [N][15], y_dev uint8 [N]: prior vector and class id (0..14) of every sample
 *   index_dev      int32 [n]: the rows of samples / atlas_rows / y that form THIS batch - generate_training_set's permutation
 *                  (base.py:92-103) as an index array - or NULL for rows 0..n-1
 *   flips_dev      uint8 [n] or NULL: per batch row two bits per view t (bit 2t: mirror the patch rows, bit 2t+1: mirror its
 *                  columns) - the rotate-180 / flip variants of the reference's Rotate_batch_Iterator (nets.py:64-84), applied
 *                  inside the gather
 *   n_vols, n_samples  sizes of the two tables: a volume number or index entry outside them is clamped on the device,
 *                  never dereferenced (voxels outside a volume read as zero padding)
 * Replaces: load_patch_vectors + generate_training_set materialising [N,1,32,32] x 3 (base.py:120-184, 53-117) in front of
 * net.fit.  Results are bit-identical to scs_train_step on the materialised patches. */
int scs_train_step_volume(scs_handle h, const float* vols_dev, const scs_volume* descs_dev, int32_t n_vols,
                          const int32_t* samples_dev, int64_t n_samples, const float* atlas_rows_dev, const uint8_t* y_dev,
                          const int32_t* index_dev, const uint8_t* flips_dev, int64_t n, uint64_t dropout_seed,
                          const uint8_t* masks_dev, float* loss_host, void* stream);

/* Class id of every sample: the label volume at the sample's voxel, with the boundary class 15 mapped to background 0
 * (generate_training_set, base.py:85-89).  label_vols_dev: uint8, concatenated like vols_dev (same descs). */
int scs_gather_labels(scs_handle h, const uint8_t* label_vols_dev, const scs_volume* descs_dev, int32_t n_vols,
                      const int32_t* samples_dev, int64_t n_samples, uint8_t* y_dev, void* stream);

/* Labelled evaluation / inference on the same compact form (deterministic forward, patches cut in the first kernel). */
int scs_eval_batch_volume(scs_handle h, const float* vols_dev, const scs_volume* descs_dev, int32_t n_vols,
                          const int32_t* samples_dev, int64_t n_samples, const float* atlas_rows_dev, const uint8_t* y_dev,
                          const int32_t* index_dev, int64_t n, double* loss_sum_host, int64_t* n_correct_host, void* stream);

/* Measurement hook (bench.py --mode train): one scs_train_step with HIP events recorded around every launch on the stream
 * it is issued on (the weight-gradient kernels run on an internal side stream); synchronises.  stage_ms[i] = summed
 * duration of the launches of stage i, i < scs_train_stage_count(); scs_train_stage_name(i) names it. */
int scs_train_stage_count(void);
const char* scs_train_stage_name(int stage);
int scs_train_step_profiled(scs_handle h, const float* ax_dev, const float* co_dev, const float* sa_dev,
                            const float* atlas_dev, const uint8_t* y_dev, int64_t n, uint64_t dropout_seed,
                            float* stage_ms, void* stream);

/* Test hook: copy an intermediate of the LAST training step to the host.  which: 0 d(fc_2 pre-activation) [n][272],
 * 1 d(FC1 pre-activation) [n][540], 2 d(d1 pre-activations) [n][540], 3 f1_drop(feat) [n][540], 4 relu(FC1) [n][540],
 * 5 fc_2 input [n][556], 6 d logits [n][16], 7 relu(fc_2) [n][272], 8 relu(d1) of the three towers [n][540];
 * 100 + 10 t + i: raw conv output of block i (2..5) of tower t, [n][Cout_i][Ho][Ho]; 200 + 10 t + i: the (scale, shift) pairs
 * the step applied as that block's batch norm (i = 1..5), float2 [Cout_i].  With these a test can read off every ReLU gate and
 * max-pool routing decision the float32 step took.  count = floats to copy. */
int scs_get_train_buffer(scs_handle h, int which, float* host_out, size_t count);

/* Device pointers to the parameter and gradient arenas (SCS_N_PARAMS floats each) - the gradient arena is
 * what a data-parallel caller all-reduces (RCCL) between scs_train_step and scs_adam_apply. */
int scs_param_arena(scs_handle h, float** dev_ptr, size_t* n);   /* marks the inference operands stale: the caller may write */
int scs_grad_arena(scs_handle h, float** dev_ptr, size_t* n);

/* Two-bucket gradient exchange (data-parallel training): makes `stream` wait until the gradients of the dense HEAD (FC1, fc_2,
 * out_layer: floats [*first_float, *first_float + *n_floats) of the gradient arena, its contiguous tail, 51 % of it) written by
 * the most recent scs_train_step are final - the grouped dense weight-gradient launch, which ends ~0.3 ms before the step does.
 * A caller may all-reduce that bucket on `stream` while the conv backward pass is still running and exchange the rest
 * ([0, *first_float): the towers) behind the step; scs_adam_apply must be ordered behind both.  SCS_ERR_STATE before the first step. */
int scs_wait_head_grads(scs_handle h, void* stream, size_t* first_float, size_t* n_floats);

/* Device address of the mean loss of the most recent scs_train_step (one float, valid once that step's work on its
 * stream has completed).  Lets a training loop collect the per-batch losses with device-side copies and read them back
 * once per epoch instead of synchronising every step (nolearn's train_loss = mean of the batch losses). */
int scs_train_loss(scs_handle h, float** dev_ptr);

/* lasagne.updates.adam over the whole arena (nets.py:236-237): t += 1, bias-corrected step size,
 * p -= a_t * m / (sqrt(v) + eps).  grad_scale multiplies the gradient first (1/world_size after a
 * sum all-reduce).  Then refreshes the inference operands. */
int scs_adam_apply(scs_handle h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Single-rank training: apply that update INSIDE every following scs_train_step* call (enable = 1), layer group by layer group
 * as the step finishes their gradients - the dense layers' 84 % of the arena beside the conv backward pass, the conv blocks
 * beside the last weight-gradient kernel, conv2.W inside its reduction - instead of as a launch of its own behind the step.
 * Same arithmetic per element as scs_adam_apply with grad_scale 1 (bit-identical parameters).  While enabled scs_adam_apply
 * returns SCS_ERR_STATE (the update must not run twice); a data-parallel caller, whose all-reduce sits between gradients and
 * update, leaves this off. */
int scs_set_fused_adam(scs_handle h, int enable, float lr, float beta1, float beta2, float eps);

/* Deterministic forward + summed cross-entropy + number of correct argmax predictions on n labelled
 * samples (nolearn eval_iter_).  Synchronises.  The sums are formed in a fixed order (bit-reproducible). */
int scs_eval_batch(scs_handle h, const float* ax_dev, const float* co_dev, const float* sa_dev,
                   const float* atlas_dev, const uint8_t* y_dev, int64_t n,
                   double* loss_sum_host, int64_t* n_correct_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCS_H */
