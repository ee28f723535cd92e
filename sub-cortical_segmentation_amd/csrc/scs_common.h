// Shared definitions of libscs: network geometry, arena offsets, context, error plumbing.
// Geometry follows /root/reference/cnn_cort/nets.py:159-231 (see include/scs.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/scs.h"

namespace scs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTowers = 3;
constexpr int kConvs = 5;
constexpr int kCin[kConvs] = {1, 20, 20, 40, 40};
constexpr int kCout[kConvs] = {20, 20, 40, 40, 60};
constexpr int kHin[kConvs] = {32, 30, 14, 12, 5};      // input height=width of conv i (nets.py:171-177)
constexpr int kFlat = 540;                               // 60*3*3 (c,h,w order)
constexpr int kD1 = 180, kFC1 = 540, kFC2 = 270, kAtlas = 15, kClasses = 15;
constexpr int kFC1ld = 556;                              // FC1 output row: 540 features + 15 atlas + 1 zero pad
constexpr int kFC2ld = 272;                              // fc_2 output row: 270 + 2 pad
constexpr float kBnEps = 1e-4f, kBnAlpha = 0.1f;         // lasagne BatchNormLayer defaults

// ------------------------------------------------------------------ Lasagne-layout arena offsets
struct TowerOff { int convW[kConvs], bn[kConvs], prelu[kConvs], d1W, d1b, preluD1, end; };
struct HeadOff { int fc1W, fc1b, prelu1, fc2W, fc2b, prelu2, outW, outb, end; };

constexpr TowerOff tower_off(int base) {
    TowerOff t{};
    int o = base;
    for (int i = 0; i < kConvs; ++i) {
        t.convW[i] = o; o += kCout[i] * kCin[i] * 9;
        t.bn[i] = o;    o += 4 * kCout[i];                // beta, gamma, mean, inv_std
        t.prelu[i] = o; o += kCout[i];
    }
    t.d1W = o; o += kFlat * kD1;
    t.d1b = o; o += kD1;
    t.preluD1 = o; o += kD1;
    t.end = o;
    return t;
}
constexpr int kTowerFloats = tower_off(0).end;           // 145 440
constexpr HeadOff head_off() {
    HeadOff h{};
    int o = kTowers * kTowerFloats;
    h.fc1W = o; o += 3 * kD1 * kFC1;
    h.fc1b = o; o += kFC1;
    h.prelu1 = o; o += kFC1;
    h.fc2W = o; o += (kFC1 + kAtlas) * kFC2;
    h.fc2b = o; o += kFC2;
    h.prelu2 = o; o += kFC2;
    h.outW = o; o += kFC2 * kClasses;
    h.outb = o; o += kClasses;
    h.end = o;
    return h;
}
static_assert(head_off().end == SCS_N_PARAMS, "arena layout must match the 883455-float reference pickle");

// ------------------------------------------------------------------ MFMA operand geometry
// Output channels are covered without padding waste: Cout = 16*NT16 + 4*NT4.  The 16-wide part runs on
// v_mfma_f32_16x16x4_f32, the remainder (4, 8 or 12 channels) on v_mfma_f32_4x4x1_16b_f32, whose 16 independent
// 4x4 blocks take the SAME pixel operand register (block b = k-subgroup b>>2, pixel quad b&3).
// Conv weights are packed per group of 4 input channels as one chunk:
//   main     [9 taps][4][NS16]   NS16 = 16*NT16 padded so NS16 % 32 == 16 (two k-rows of a 32-lane LDS read group
//                                land on disjoint banks)
//   leftover [9 taps][4][L4]     L4 = 4*NT4
constexpr int n_tiles(int cout) { return (cout + 15) / 16; }              // bias array granularity
constexpr int n_stride(int cout) { int n = n_tiles(cout) * 16; return (n % 32 == 16) ? n : n + 16; }   // GEMM packs
constexpr int nt16(int cout) { return cout / 16; }
constexpr int nt4(int cout) { return (cout % 16) / 4; }
constexpr int ns16(int cout) { int n = nt16(cout) * 16; return n == 0 ? 0 : ((n % 32 == 16) ? n : n + 16); }
constexpr int conv_chunk_floats(int cout) { return 36 * (ns16(cout) + 4 * nt4(cout)); }
// activation planes are padded so that plane stride % 32 == 16 where the LDS budget allows
constexpr int plane_stride(int hw) { int p = hw; while (p % 32 != 16) ++p; return p; }

constexpr int conv_pack_floats(int i) { return kCin[i] < 4 ? 0 : (kCin[i] / 4) * conv_chunk_floats(kCout[i]); }

// source coordinates of element e of a packed conv weight array: input-channel group c4, tap, channel-in-group kk,
// output channel n (-1 = padding, stored as zero)
struct PackIdx { int c4, tap, kk, n; };
__host__ __device__ constexpr PackIdx conv_pack_index(int e, int cout) {
    const int ch = conv_chunk_floats(cout), n16 = ns16(cout), l4 = 4 * nt4(cout);
    const int c4 = e / ch;
    int r = e - c4 * ch, row = 0, n = -1;
    if (r < 36 * n16) { row = r / n16; const int col = r - row * n16; n = col < nt16(cout) * 16 ? col : -1; }
    else { r -= 36 * n16; row = r / l4; n = nt16(cout) * 16 + (r - row * l4); }
    return PackIdx{c4, row >> 2, row & 3, n};
}

// GEMM weights are packed [K4][NS]: K padded to a multiple of 4, NS = padded N (multiple of 16, %32==16)
constexpr int gemm_ns(int n) { return n_stride(n); }
constexpr int round4(int k) { return (k + 3) / 4 * 4; }

struct PackedNet {                 // device pointers into one packed-weights allocation
    float* conv1w[kTowers];        // conv1 MFMA operands: [12][16] + [12][4] + bias[20]
    float* conv1b[kTowers];        // (unused)
    float* convw[kTowers][kConvs]; // i = 1..4: packed MFMA layout
    float* convb[kTowers][kConvs]; // [NT*16] zero padded
    float* d1w[kTowers];           // [540][NS(180)]
    float* d1b[kTowers];
    float* fc1w; float* fc1b;      // [540][NS(540)]
    float* fc2w; float* fc2b;      // [556][NS(270)]
    float* outw; float* outb;      // [270][16], [16]
};

}  // namespace scs

// ------------------------------------------------------------------ error plumbing
namespace scs {
void set_error(const char* fmt, ...);
}
#define SCS_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            scs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SCS_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)
#define SCS_ARG(cond, msg)                                   \
    do {                                                     \
        if (!(cond)) { scs::set_error("bad argument: %s", msg); return SCS_ERR_ARG; } \
    } while (0)
