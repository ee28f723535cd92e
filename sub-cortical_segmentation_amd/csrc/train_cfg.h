// Kernel configurations of the training step (scs_train.hip), shared with the micro-benchmark tools/tbench.hip.
#pragma once
#include "conv_mfma.h"
#include "wgrad_mfma.h"

namespace scs {

// ------------------------------------------------------------------------------------------- configurations
// <CIN, COUT, HIN, patches per workgroup, epilogue, source, LDS plane stride, bands, workgroups per CU>; tilings picked at
// batch 256 with tools/tbench.hip (profiles/r02_tbench_sweep.txt)
using TF12 = ConvCfg<20, 20, 30, 1, EPI_RAW_STATS, SRC_X0, 180, 7, 4>;          // conv1 (recomputed) + conv2
using TF3 = ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN, 208, 1, 4, 0, 4, 1>;        // BN2 + ReLU while staging the pooled-by-sign conv2 output (ConvArgs::pout); finalizes BN3
using TF4 = ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 144, 1, 4, 0, 4, 1>;        // BN3 + ReLU while staging; finalizes BN4
using TF5 = ConvCfg<40, 60, 5, 4, EPI_RAW_STATS, SRC_BN_POOL, 27, 1, 1, 0, 6, 0, 2>;   // BN4 + ReLU + pool2 while staging; 4 patches (3 M
                                                                                 // tiles) per workgroup, 6 waves = 3 channel pairs x 2 K parts (K split)
// data gradients: input = dZ_i (computed while staging) zero-padded by 2, Cin' = Cout_i, Cout' = Cin_i; the epilogue
// leaves the BN-backward partial sums of the block below and (last template argument) finalizes them in its last workgroup
using DG5 = ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 0, 4, 1>;     // below: conv4 block (pooled)
using DG4 = ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 208, 1, 3, 0, 4, 1>;   // below: conv3 block
using DG3 = ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 256, 1, 3, 0, 4, 1>;   // below: conv2 block (pooled)
using DG2 = ConvCfg<20, 20, 32, 1, EPI_L1, SRC_DZ_POOL_PAD2, 256, 5, 4>;        // below: conv1 block (recomputed)

// The same tilings without the in-kernel finalize.  The finalize costs every workgroup a drained write-through store and a
// returning ticket (~1.5 us during which it holds its CU slot): with ONE generation of workgroups (batch 256: 768 patch-towers on
// 256 CUs x 3) that is a 5 us tail against the ~9 us of a separate finalize launch, but it grows with the generations (+20 us per
// kernel at batch 2048, profiles/r03_tbench_fin.txt), so beyond kFinMaxBatch the step uses these and launches the finalize.
using TF3N = ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN, 208, 1, 4, 0, 4, 0>;
using TF4N = ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 144, 1, 4, 0, 4, 0>;
using DG5N = ConvCfg<60, 40, 7, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 49, 1, 4, 0, 4, 0>;
using DG4N = ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 208, 1, 3, 0, 4, 0>;
using DG3N = ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 256, 1, 3, 0, 4, 0>;
// Two workgroups per patch-tower (row bands) for the four one-generation kernels of the middle layers at SMALL batches: at batch
// 128 the unbanded tilings launch 384 workgroups on 256 CUs - half the machine runs a full-length latency chain twice, the other
// half once - the banded ones 768 of half the work (tools/tbench.hip "band", same GPU, batch 128: conv3 fwd 27.6 -> 25.1 us, conv4
// fwd 30.6 -> 26.1, conv4 dgrad 40.0 -> 34.6, conv3 dgrad 34.6 -> 31.0; no gain at 160, 20 % SLOWER at 256: two generations).
// Taken up to kSplitMaxBatch (option split_max_batch) together with the in-kernel finalize.
using TF3B = ConvCfg<20, 40, 14, 1, EPI_RAW_STATS, SRC_BN, 112, 2, 3, 0, 4, 1>;
using TF4B = ConvCfg<40, 40, 12, 1, EPI_RAW_STATS, SRC_BN, 100, 2, 4, 0, 4, 1>;
using DG4B = ConvCfg<40, 40, 14, 1, EPI_RAW_BNB, SRC_DZ_POOL_PAD2, 112, 2, 3, 0, 4, 1>;
using DG3B = ConvCfg<40, 20, 16, 1, EPI_RAW_BNB_POOL, SRC_DZ_PAD2, 144, 2, 3, 0, 4, 1>;
constexpr int kSplitMaxBatch = 128;
constexpr int kC5SplitMaxBatch = 128;     // conv5 block's backward as two launches (data path / weight-gradient partial on the side stream) up to here
constexpr int kFinMaxBatch = 512;
// conv1's statistics (from the input moments) finalized inside the prep + moments launch up to this batch, by bn1_final_kernel beyond
constexpr int kBn1FusedMaxBatch = 512;

constexpr int kHout[kConvs] = {30, 28, 12, 10, 3};
// <CIN, COUT, HIN, row bands, workgroups per CU, X source, software-pipelined staging>.  d z is read from memory (the data-gradient
// kernel of the layer stores it as a by-product of its staging, which costs it nothing, while forming it again in the weight-
// gradient kernel cost 3-15 us per layer - profiles/r02_tbench_final.txt); conv2's X is the recomputed conv1 block output.
using WG2 = WgCfg<20, 20, 30, 4, 3, 1>;
using WG3 = WgCfg<20, 40, 14, 2, 3, 0>;
using WG4 = WgCfg<40, 40, 12, 5, 2, 0, 1>;           // software-pipelined staging (8 patches per workgroup)
using WG5 = WgCfg<40, 60, 5, 1, 2, 0>;
constexpr size_t kWgPartFloats = (size_t)256 * 60 * 40 * 9;   // per tower and layer: room for 256 partials of the largest layer
constexpr int kL1Bands = DG2::NB;


}  // namespace scs
