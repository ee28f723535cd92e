// Inference convolutions on the f16 matrix cores with f32-grade results: every f32 operand is carried as TWO f16 pieces
// (a = ah + al, ah = f16(a), al = f16(a - ah): 22 significant bits) and a product is the three MFMAs ah*bh + ah*bl + al*bh with
// f32 accumulation (every f16 x f16 product is exact in f32; the dropped al*bl is 2^-22 of the product).  Operands are scaled by
// powers of two so that their low pieces stay clear of the f16 subnormals (tools/f16split_probe.hip: at operand magnitudes
// >= 2^4 the error against double equals the f32 MFMA chain's, 2.8e-8 of the output scale).  Three v_mfma_f32_16x16x32_f16
// (16 cycles each) do the work of eight v_mfma_f32_16x16x4_f32 (32 cycles each).
//
// GEMM view ("D transposed"):  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]
//   A operand = weights (rows = output channels, 16 per tile), resident in registers for the whole launch (persistent workgroups
//   of one tower);  B operand = activations from LDS, stored [pixel][channel] (channel innermost) as two f16 planes, so that for
//   a fixed tap row dy the K index (dx, cin) is ONE contiguous run of 3 * Cin halves starting at pixel (y + dy, x): a lane's 8
//   consecutive k are 16 contiguous bytes.  A run is padded to a multiple of 32 k (20 channels: 60 -> 64; the pad reads the next
//   pixel's first channels against zero weights).  D: lane (n = l & 15, j = l >> 4) holds output channels 4j .. 4j+3 of pixel n,
//   i.e. 8 contiguous bytes of the next layer's [pixel][channel] f16 plane.
//   Pooling: the four N tiles of a group are the four positions of the 2x2 windows of 16 pooled pixels: max-pool is an in-lane max.
//   LDS banks: 16 consecutive pooled pixels are 80 bytes apart (20 dwords, 5 mod 16 in units of 4 dwords): their 8-byte reads
//   cover every fourth bank pair exactly once; lane groups with odd j read the two halves of their 16 bytes in the opposite order
//   (the weights of those lanes are packed with the same swap), which puts them on the remaining bank pairs: conflict-free.
//
// conv1 (Cin = 1, K = 9) stays on the f32 MFMA (exact) in the same transposed form: its output is scaled, split and written to
// the LDS planes by the lanes that hold it.
//
// Replaces, at inference, Conv2DLayer -> BatchNormLayer -> rectify (-> MaxPool2DLayer) of /root/reference/cnn_cort/nets.py:171-173
// (BN folded into weights and bias, true convolution = correlation with the flipped filter).
#pragma once
#include "scs_common.h"

namespace scs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef SP_ROWREUSE
#define SP_ROWREUSE 1        // conv1+conv2+pool1: every LDS fragment of the conv2 stage read once for both window rows it serves (0: per tile and k-step)
#endif
#ifndef SP_GLDS
#define SP_GLDS 1            // conv3 / conv4 (conv_split_all_kernel): the next iteration's planes copied global -> LDS by global_load_lds_dwordx4 (no
#endif                       // registers, no LDS-write pass) instead of through 48-64 prefetch registers per thread (0: the register-staged form)
#ifndef SP_ABLATE
#define SP_ABLATE 0          // tools/sbench timing experiments only (results are wrong when set): 1 conv1 stage only on a workgroup's first
#endif                       // patch, 2 no conv2 MFMAs after the first k-step, 3 no stores, 4 no overflow tracking; conv_split_all_kernel:
                             // 5 no staging after the first iteration, 6 no epilogue, 7 one k-step only, 8 no barrier
constexpr float kSplitActScale = 64.f;         // activations are stored times 2^6 (overflow beyond |a| = 1023: flagged, see below)
constexpr float kSplitOverflow = 60000.f;      // a scaled activation above this sets the launch's overflow flag
constexpr int kSp12PlanePx = 904;              // conv1 plane: 30 x 30 pixels + 4 pad pixels: the last runs overrun 8 bytes into the pad (kept zero);
                                               // pad bytes 32.. are the dummy slot of the conv1 columns that do not exist
constexpr int kSp12PlaneBytes = kSp12PlanePx * 40;
constexpr int kSp12XFloats = 1024 + 64;        // raw patch + pad (tiles of the 32-wide domain read up to 2 rows + 3 past the end)
constexpr size_t kSp12LdsBytes = 2 * (size_t)kSp12PlaneBytes + kSp12XFloats * 4;
constexpr int kSp12W1Floats = 12 * 20 + 20;    // conv1 operands: [slot 4 dy + dx (dx = 3: zero)][20 couts] + bias[20], both times kSplitActScale
constexpr int kSp12W2Vec = 6 * 2 * 2 * 64;     // conv2 operands: [k-step][M tile][piece][lane] x 16 bytes
// ... followed (inference pack only) by the STACKED remainder tiles [pixel row R = 0..3][k half][piece][lane]: conv2's 20 output channels
// are one full M tile and 4 rows of a second one; in the row-reuse walk below a fragment (pixel row R of a pooling group's 4 x 2
// pixels) serves tap row R of the windows' upper output row and tap row R - 1 of their lower one, so the remainder channels' weights
// for both uses share ONE A operand - rows 0..3: channels 16..19 at tap row R, rows 4..7: the same channels at tap row R - 1 (rows
// that do not exist: zero) - and one MFMA triple per fragment does the work of two: 16 x 3 instead of 24 x 3 MFMAs per pooling group
// on the remainder tile, -17 % of the conv2 MFMAs.  The upper row's sums come out in accumulator rows 0..3 (lane group 0), the lower
// row's in rows 4..7 (lane group 1).
constexpr int kSp12W2StackVec = 4 * 2 * 2 * 64;
#ifndef SP_STACK
#define SP_STACK 1           // conv1+conv2+pool1: the stacked remainder tiles (0: the plain second M tile, 37.5 % of whose rows are zero)
#endif
// element of a stacked tile: row r of the A operand of (pixel row R) -> (output channel, tap row) or nothing
__host__ __device__ constexpr bool sp12_stack_row(int R, int r, int& o, int& dy) {
    if (r < 4) { o = 16 + r; dy = R; return R <= 2; }
    if (r < 8) { o = 16 + (r - 4); dy = R - 1; return R >= 1; }
    o = 0; dy = 0; return false;
}

struct Split12Args {
    const float* in[kTowers];     // [n][32][32]; VOL: in[0] = the z-scored volume [dimx][dimy][dimz], patches are cut around `centers`
    const int32_t* centers;       // VOL: [n][3] centre voxels (tri-planar gather of base.py:293-306, zero padded)
    int dimx, dimy, dimz;
    const float* w1[kTowers];     // kSp12W1Floats
    const uint4* w2[kTowers];     // kSp12W2Vec
    const float* b2[kTowers];     // [32] folded bias of conv2 (20 used), then one float: 1 / (kSplitActScale * weight scale of conv2), a power of two
    void* out[kTowers];           // OUT_SPLIT: [n][2][196 x 40 bytes] f16 planes [pixel][channel] (hi, lo; times kSplitActScale);
                                  // else [n][20][14][14] f32 (the f32 kernels' pool1 layout)
    int n;
    unsigned int* flag;           // set to 1 when a scaled activation left the f16 range (results of the launch are then void)
};

// element e (0..63) of the run of tap row dy -> (dx, cin); e >= 60: padding
__host__ __device__ constexpr int sp_run_dx(int e, int cin) { return e / cin; }

// lane's element order inside its 16-byte chunk: lane groups with odd j hold the two 8-byte halves swapped
__host__ __device__ constexpr int sp_chunk_elem(int j, int pos) { return (j & 1) ? (pos ^ 4) : pos; }
// ReLU, overflow tracking and the two-piece split of four f32 values (channels 4j .. 4j+3 of one pixel) -> 8 + 8 bytes of the LDS planes.
// ReLU as an integer max on the bits (no NaN canonicalisation); the low piece as ONE mixed-precision fma per value (f16 operand
// widened inside the instruction: v_fma_mixlo / mixhi_f16), rounded to nearest like the high piece.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sp_relu(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
__device__ __forceinline__ void sp_split4(f32x4 v, f16x4& h, f16x4& l) {
    const f16x2 h01 = __builtin_convertvector((f32x2){v[0], v[1]}, f16x2), h23 = __builtin_convertvector((f32x2){v[2], v[3]}, f16x2);
    h = f16x4{h01[0], h01[1], h23[0], h23[1]};
    // l = f16(v - h): v_fma_mix{lo,hi}_f16 take the f16 operand as it is (op_sel_hi) and round the f32 result once, to nearest -
    // one instruction per value where the compiler's own selection is widen + packed subtract + narrow
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(u01), "v"(v[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(u01), "v"(v[1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(u23), "v"(v[2]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(u23), "v"(v[3]));
    const f16x2 a = __builtin_bit_cast(f16x2, l01), b = __builtin_bit_cast(f16x2, l23);
    l = f16x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ void sp_split4_store(f32x4 v, unsigned char* hi, unsigned char* lo, int off, float& vmax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = sp_relu(v[i]);
    if (SP_ABLATE != 4) {
        vmax = fmaxf(fmaxf(vmax, v[0]), v[1]);
        vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);
    }
    f16x4 h, l;
    sp_split4(v, h, l);
    *reinterpret_cast<f16x4*>(hi + off) = h;
    *reinterpret_cast<f16x4*>(lo + off) = l;
}

// One B fragment = two 8-byte LDS reads (the two halves of the lane's 16 bytes, order by lane group: see the bank note above).
// Volatile: the compiler otherwise pairs reads off the same base register ACROSS fragments into ds_read2_b64 (half the LDS rate,
// 32-bank mapping) and reassembles the fragments with register moves - more vector instructions than MFMAs.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) u32x2* sp_lds_cv8;   // explicit LDS address space: a volatile access through a
                                                                               // generic pointer is compiled to a flat load
__device__ __forceinline__ f16x8 sp_frag(const unsigned char* p0, const unsigned char* p1) {
    const u32x2 a = *(sp_lds_cv8)(p0), b = *(sp_lds_cv8)(p1);
    const u32x4 v = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(f16x8, v);
}

template <int LANES_MASKED = 0>
__device__ __forceinline__ float sp_quad_max(float v) {
    // max over the 4 lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1])
    const int a = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
    v = fmaxf(v, __builtin_bit_cast(float, a));
    const int b = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);
    return fmaxf(v, __builtin_bit_cast(float, b));
}

// pooled pixel w of a patch-tower: channels 4 j .. 4 j + 3 (v0) and, from lane group 0, 16 .. 19 (v1)
template <bool OUT_SPLIT>
__device__ __forceinline__ void sp12_store(unsigned char* op, int w, int j, f32x4 v0, f32x4 v1, float& vmax) {
    if (SP_ABLATE == 3 && v0[0] != 123.25f) return;
    if constexpr (OUT_SPLIT) {
        f16x4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) v0[i] *= kSplitActScale;
        sp_split4(v0, h, l);
        vmax = fmaxf(fmaxf(vmax, v0[0]), v0[1]); vmax = fmaxf(fmaxf(vmax, v0[2]), v0[3]);
        *reinterpret_cast<f16x4*>(op + w * 40 + 8 * j) = h;
        *reinterpret_cast<f16x4*>(op + 7840 + w * 40 + 8 * j) = l;
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v1[i] *= kSplitActScale;
            sp_split4(v1, h, l);
            vmax = fmaxf(fmaxf(vmax, v1[0]), v1[1]); vmax = fmaxf(fmaxf(vmax, v1[2]), v1[3]);
            *reinterpret_cast<f16x4*>(op + w * 40 + 32) = h;
            *reinterpret_cast<f16x4*>(op + 7840 + w * 40 + 32) = l;
        }
    } else {
        float* o = reinterpret_cast<float*>(op);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[(4 * j + i) * 196 + w] = v0[i];
            if (j == 0) o[(16 + i) * 196 + w] = v1[i];
        }
    }
}

// grid = 3 * G workgroups of 256 threads: workgroup b serves tower b % 3, patches b / 3, b / 3 + G, ...
// one element of the tri-planar gather, zero padded (base.py:293-306): patch pixel (i, j) of `tower` around (cx, cy, cz):
// axial [x-16+i][y-16+j][z], coronal [x-16+i][y][z-16+j], sagittal [x][y-16+i][z-16+j]  (as gather_voxel of conv_mfma.h)
__device__ __forceinline__ float sp_gather(const float* __restrict__ vol, int dx, int dy, int dz, int tower, int cx, int cy, int cz, int i, int j) {
    int vx, vy, vz;
    if (tower == 0) { vx = cx - 16 + i; vy = cy - 16 + j; vz = cz; }
    else if (tower == 1) { vx = cx - 16 + i; vy = cy; vz = cz - 16 + j; }
    else { vx = cx; vy = cy - 16 + i; vz = cz - 16 + j; }
    const bool ok = (unsigned)vx < (unsigned)dx && (unsigned)vy < (unsigned)dy && (unsigned)vz < (unsigned)dz;
    return ok ? vol[((size_t)vx * dy + vy) * dz + vz] : 0.f;
}

template <bool OUT_SPLIT, bool VOL>
__global__ __launch_bounds__(256, 2) void conv12_split_kernel(Split12Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned char* s_hi = sp_smem;
    unsigned char* s_lo = sp_smem + kSp12PlaneBytes;
    float* s_x = reinterpret_cast<float*>(sp_smem + 2 * kSp12PlaneBytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    // pads: zero once (never written afterwards)
    for (int e = tid; e < 160 / 4; e += 256) {
        reinterpret_cast<unsigned*>(s_hi + 900 * 40)[e] = 0u;
        reinterpret_cast<unsigned*>(s_lo + 900 * 40)[e] = 0u;
    }
    if (tid < 64) s_x[1024 + tid] = 0.f;
    // ---- resident operands
    const float* w1 = a.w1[tower];
    float a1[3], a4[9];
    f32x4 b1m, b1r;
#pragma unroll
    for (int s = 0; s < 3; ++s) a1[s] = w1[(4 * s + j) * 20 + n];                        // couts 0..15 (n = row of the A operand)
#pragma unroll
    for (int t = 0; t < 9; ++t) a4[t] = w1[(4 * (t / 3) + t % 3) * 20 + 16 + (lane & 3)];   // couts 16..19 on the 4x4x1 blocks
#pragma unroll
    for (int i = 0; i < 4; ++i) { b1m[i] = w1[240 + 4 * j + i]; b1r[i] = w1[240 + 16 + i]; }
    constexpr int MREG = (SP_STACK && SP_ROWREUSE) ? 1 : 2;      // M tiles held in the plain form (stacked: the full tile only)
    f16x8 wh[6][MREG], wl[6][MREG];
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int m = 0; m < MREG; ++m) {
            wh[s][m] = __builtin_bit_cast(f16x8, a.w2[tower][((s * 2 + m) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, a.w2[tower][((s * 2 + m) * 2 + 1) * 64 + lane]);
        }
#if SP_STACK && SP_ROWREUSE
    f16x8 wsh[4][2], wsl[4][2];                                  // the stacked remainder tiles [pixel row R][k half]
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            wsh[R][hh] = __builtin_bit_cast(f16x8, a.w2[tower][kSp12W2Vec + ((R * 2 + hh) * 2 + 0) * 64 + lane]);
            wsl[R][hh] = __builtin_bit_cast(f16x8, a.w2[tower][kSp12W2Vec + ((R * 2 + hh) * 2 + 1) * 64 + lane]);
        }
#endif
    f32x4 bias0, bias1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bias0[i] = a.b2[tower][4 * j + i]; bias1[i] = a.b2[tower][16 + i]; }
    const float inv = a.b2[tower][32];
    // per-lane byte offsets of the wave's three pooling groups (16 pooled pixels each) and of the odd tile (pooled 192..195)
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
    const int rs_up = ((n & 3) >> 1) * 1200;      // (stacked remainder of the odd tile: every column starts at its window's TOP pixel row)
    float vmax = 0.f;
    const float* inp = VOL ? a.in[0] : a.in[tower];
    unsigned char* outp = reinterpret_cast<unsigned char*>(a.out[tower]);
    auto load_x = [&](int p) -> float4 {                     // elements 4 tid .. 4 tid + 3 of patch p (row tid / 8, columns 4 (tid % 8) ..)
        if constexpr (!VOL) return reinterpret_cast<const float4*>(inp + (size_t)p * 1024)[tid];
        else {
            const int cx = a.centers[3 * p], cy = a.centers[3 * p + 1], cz = a.centers[3 * p + 2], i = tid >> 3, j = 4 * (tid & 7);
            return make_float4(sp_gather(inp, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, i, j), sp_gather(inp, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, i, j + 1),
                               sp_gather(inp, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, i, j + 2), sp_gather(inp, a.dimx, a.dimy, a.dimz, tower, cx, cy, cz, i, j + 3));
        }
    };
    float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wg < a.n) xr = load_x(wg);
    int it = 0;
    for (int p = wg; p < a.n; p += nwg, ++it) {
        reinterpret_cast<float4*>(s_x)[tid] = xr;
        __syncthreads();                                     // raw patch in place; every wave is done with the planes of the previous patch
        if (p + nwg < a.n) xr = load_x(p + nwg);
        // ---- conv1 on the f32 matrix cores: 60 tiles of 16 pixels over the 32-wide row domain (columns 30, 31 are not stored)
        if (SP_ABLATE != 1 || it == 0) {
        // five tiles at a time: their 15 operand reads, then the MFMAs of the five accumulators interleaved (a tile's three MFMAs
        // depend on each other), then the five epilogues; columns 30, 31 of the 32-wide domain go to a dummy slot behind the planes
#pragma unroll 1
        for (int g5 = 0; g5 < 3; ++g5) {
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
        }
        __syncthreads();                                     // planes complete
        // ---- conv2 on the f16 matrix cores
        unsigned char* op = outp + (size_t)p * (20 * 196 * 4);      // both formats: 15 680 bytes per patch-tower
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const unsigned char* h0 = s_hi + rd0[k];
            const unsigned char* h1 = s_hi + rd1[k];
            const unsigned char* l0 = s_lo + rd0[k];
            const unsigned char* l1 = s_lo + rd1[k];
            f32x4 acc[4][MREG];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < MREG; ++m) acc[q][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#if SP_STACK && SP_ROWREUSE
            f32x4 accs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // remainder channels: [column parity]; rows 0..3 upper, 4..7 lower output row
#endif
#if SP_ROWREUSE
            // fragment f = 4 R + 2 half + column parity: pixel row R (0..3) of the windows' 4 x 2 pixels, read ONCE and used by the window
            // row it serves as tap row R (output row 0) and as tap row R - 1 (output row 1): 16 fragment pairs per group instead of 24;
            // three in flight
            f16x8 fh[3], fl[3];
            auto rdfrag = [&](int f) {
                const int off = (f >> 2) * 1200 + (f & 1) * 40 + ((f >> 1) & 1) * 64;
                fh[f % 3] = sp_frag(h0 + off, h1 + off); fl[f % 3] = sp_frag(l0 + off, l1 + off);
            };
            rdfrag(0); rdfrag(1);
#pragma unroll
            for (int f = 0; f < (SP_ABLATE == 2 ? 4 : 16); ++f) {
                const int R = f >> 2, hh = (f >> 1) & 1, qc = f & 1;
                if (f + 2 < 16) rdfrag(f + 2);
                const f16x8 bh = fh[f % 3], bl = fl[f % 3];
#pragma unroll
                for (int piece = 0; piece < 3; ++piece) {
#pragma unroll
                    for (int use = 0; use < 2; ++use) {
                        const int dy = R - use;
                        if (dy < 0 || dy > 2) continue;
                        const int s = 2 * dy + hh, q = 2 * use + qc;
#pragma unroll
                        for (int m = 0; m < MREG; ++m) {
                            if (piece == 0) acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[q][m], 0, 0, 0);
                            else if (piece == 1) acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[q][m], 0, 0, 0);
                            else acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[q][m], 0, 0, 0);
                        }
                    }
#if SP_STACK
                    if (piece == 0) accs[qc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsl[R][hh], bh, accs[qc], 0, 0, 0);
                    else if (piece == 1) accs[qc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsh[R][hh], bl, accs[qc], 0, 0, 0);
                    else accs[qc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsh[R][hh], bh, accs[qc], 0, 0, 0);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            // fragments of k-step s + 1 are requested behind the MFMAs of the same tile of k-step s (at most 12 reads newer than the
            // one a tile waits for: the LDS counter has 4 bits)
            f16x8 fh[2][4], fl[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (q >> 1) * 1200 + (q & 1) * 40;
                fh[0][q] = sp_frag(h0 + off, h1 + off); fl[0][q] = sp_frag(l0 + off, l1 + off);
            }
#pragma unroll
            for (int s = 0; s < (SP_ABLATE == 2 ? 1 : 6); ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], fh[s & 1][q], acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fl[s & 1][q], acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fh[s & 1][q], acc[q][m], 0, 0, 0);
                    }
                    if (s + 1 < 6) {
                        const int off = (((s + 1) >> 1) + (q >> 1)) * 1200 + (q & 1) * 40 + ((s + 1) & 1) * 64;
                        fh[(s + 1) & 1][q] = sp_frag(h0 + off, h1 + off); fl[(s + 1) & 1][q] = sp_frag(l0 + off, l1 + off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
            const int w = 16 * (wave + 4 * k) + n;
            f32x4 v0, v1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v0[i] = fmaxf(fmaxf(fmaxf(acc[0][0][i], acc[1][0][i]), fmaxf(acc[2][0][i], acc[3][0][i])) * inv + bias0[i], 0.f);
#if SP_STACK && SP_ROWREUSE
                // the window's four positions: two column parities in this lane, the lower output row in lane + 16 (accumulator rows 4..7)
                const float up = fmaxf(accs[0][i], accs[1][i]);
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, __builtin_bit_cast(int, up)));
                v1[i] = fmaxf(fmaxf(up, lo) * inv + bias1[i], 0.f);          // (read by lane group 0 only)
#else
                v1[i] = fmaxf(fmaxf(fmaxf(acc[0][1][i], acc[1][1][i]), fmaxf(acc[2][1][i], acc[3][1][i])) * inv + bias1[i], 0.f);
#endif
            }
            sp12_store<OUT_SPLIT>(op, w, j, v0, v1, vmax);
        }
        if (wave == (it & 3)) {                              // the odd tile: pooled pixels 192..195, the four positions of each in one quad of lanes
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
                for (int m = 0; m < MREG; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[m], 0, 0, 0);
                }
            }
#if SP_STACK && SP_ROWREUSE
            // the remainder channels in the stacked form: every column walks the four pixel rows of ITS window from the top (columns of the
            // lower output row repeat those of the upper one: their sums are the same numbers), rows 0..3 / 4..7 = upper / lower output row
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int off = (f >> 1) * 1200 + (f & 1) * 64 - rs_up;
                const f16x8 bh = sp_frag(h0 + off, h1 + off), bl = sp_frag(l0 + off, l1 + off);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsl[f >> 1][f & 1], bh, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsh[f >> 1][f & 1], bl, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wsh[f >> 1][f & 1], bh, acc[1], 0, 0, 0);
            }
#endif
            const int w = 192 + (n >> 2);
            f32x4 v0, v1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v0[i] = fmaxf(sp_quad_max(acc[0][i]) * inv + bias0[i], 0.f);
#if SP_STACK && SP_ROWREUSE
                const float up = sp_quad_max(acc[1][i]);          // both column parities (and the repeated rows) of the window
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, __builtin_bit_cast(int, up)));
                v1[i] = fmaxf(fmaxf(up, lo) * inv + bias1[i], 0.f);
#else
                v1[i] = fmaxf(sp_quad_max(acc[1][i]) * inv + bias1[i], 0.f);
#endif
            }
            if ((n & 3) == 0) sp12_store<OUT_SPLIT>(op, w, j, v0, v1, vmax);
        }
    }
    if (vmax > kSplitOverflow) atomicOr(a.flag, 1u);
}

inline hipError_t conv12_split_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_split_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSp12LdsBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_split_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSp12LdsBytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_split_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSp12LdsBytes);
}

template <bool OUT_SPLIT, bool VOL = false>
inline void conv12_split_launch(const Split12Args& a, hipStream_t st, int wgs_per_tower = 170) {
    int g = wgs_per_tower < a.n ? wgs_per_tower : a.n;
    if (g < 1) return;
    hipLaunchKernelGGL((conv12_split_kernel<OUT_SPLIT, VOL>), dim3(3 * g), dim3(256), kSp12LdsBytes, st, a);
}


// =====================================================================================================================
// conv3 / conv4 (+ pool2) / conv5 on the split: activations arrive and leave as two f16 planes [pixel][channel] per patch-tower
// (hi plane, then lo plane; values times kSplitActScale) - the same bytes per element as the f32 layout.
// One wave per 16-channel output tile (its weights: KS k-steps x 2 pieces x 16 bytes per lane, in registers for the whole launch);
// every wave walks all pixel tiles of the workgroup's G patch-towers, so the B operand is read by each of the MT waves.
// Pixel tiles: 4 x 4 blocks of output pixels (plain), the four window positions of 16 pooled pixels (POOL), or the 3 x 3 outputs of
// one patch (conv5: 9 of 16 lanes).
template <int CIN_, int COUT_, int HIN_, bool POOL_, int OUTF_, int G_, int WPC_>
struct SpCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, G = G_, OUTF = OUTF_, WPC = WPC_;
    static constexpr bool POOL = POOL_;
    static constexpr int HO = HIN_ - 2, HP = HO / 2;
    static constexpr int MT = (COUT_ + 15) / 16, NW = MT, NTH = 64 * MT;
    static constexpr int PS = 2 * CIN_, RS = HIN_ * PS;                       // pixel / row stride of a plane, bytes
    static constexpr int KSR = (3 * CIN_ + 31) / 32, KS = 3 * KSR;            // k-steps per tap row / in all
    static constexpr int IN_PX = HIN_ * HIN_, IN_B = IN_PX * PS;              // one plane of one patch-tower in global memory
    static constexpr int PLANE_B = IN_B + 32;                                 // in LDS: + pad for the last run's overrun
    static constexpr int V = IN_B / 16;                                       // 16-byte pieces per plane
    static constexpr int PF = (2 * V * G_ + NTH - 1) / NTH;                   // prefetch registers (uint4) per thread
    static constexpr bool SMALL = (HO < 4);                                   // conv5: one tile per patch, lanes 0..8
    static constexpr int NPOOL = HP * HP, NGRP = (NPOOL + 15) / 16;
    static constexpr int TPP = SMALL ? 1 : (POOL_ ? 4 * NGRP : (HO / 4) * (HO / 4));   // pixel tiles per patch-tower
    static constexpr int NT = TPP * G_;
    static constexpr int OUT_PX = POOL_ ? NPOOL : HO * HO;
    static constexpr int OUT_PS = 2 * COUT_, OUT_B = OUT_PX * OUT_PS;         // output plane (OUTF 0)
    static constexpr size_t LDS_BYTES = (size_t)2 * G_ * PLANE_B;
    static_assert(IN_B % 16 == 0, "planes are staged in 16-byte pieces");
    static_assert(SMALL || POOL_ || HO % 4 == 0, "plain tiles are 4 x 4 pixel blocks");
    static_assert(!SMALL || HO * HO <= 16, "small form: all outputs of a patch in one tile");
    static_assert(32 * KSR - 3 * CIN_ <= 16 && 32 * KSR >= 3 * CIN_, "run padding stays inside the plane pad");
};

// pooled pixel (window) that lane n of pooling group k serves, -1: none
template <class C>
__device__ __forceinline__ int sp_window(int k, int n) {
    const int w = 16 * k + n;
    return w < C::NPOOL ? w : -1;
}
// byte offsets (inside an LDS plane) of the two 8-byte halves this lane reads for tile family k: its pixel, its lane group's
// 16-byte chunk, odd lane groups in the opposite order (bank note at the top of the file).
// (Measured and not kept: rows padded to 968 bytes with a window-to-lane partition that makes every conv4 read conflict-free, and a
// 576-byte row / chunk order 0,2,1,3 for conv3 - no gain on either: the reads are not what these two kernels wait for.)
template <class C>
__device__ __forceinline__ void sp_lane_base(int k, int n, int j, int& rd0, int& rd1) {
    int row, col;
    if constexpr (C::POOL) { const int w = max(sp_window<C>(k, n), 0); row = 2 * (w / C::HP); col = 2 * (w % C::HP); }
    else if constexpr (C::SMALL) { row = n < 9 ? n / 3 : 0; col = n < 9 ? n % 3 : 0; }
    else { row = n >> 2; col = n & 3; }
    const int x = row * C::RS + col * C::PS + 16 * j;
    rd0 = x + 8 * (j & 1); rd1 = x + 8 - 8 * (j & 1);
}

struct SplitArgs {
    const unsigned char* in[kTowers];   // [n][2][IN_B]
    const uint4* w[kTowers];            // [KS][MT][2][64] x 16 bytes
    const float* bias[kTowers];         // [MT * 16] folded bias, then one float: 1 / (kSplitActScale * weight scale), a power of two
    void* out[kTowers];                 // OUTF 0: [n][2][OUT_B] planes; OUTF 1: f32 [n][COUT][HO][HO]; OUTF 2: rows of the split dense head
                                        // (dense_split.h): [n] x (4 * out_kp + 16) bytes, feature = channel * OUT_PX + pixel (the flatten order
                                        // of nets.py:178): high halves, low halves at + 2 * out_kp bytes, values times kSplitActScale
    int out_kp;                         // OUTF 2: halves per piece of a row
    int n;
    unsigned int* flag;
};

template <class C>
__global__ __launch_bounds__(C::NTH, C::WPC) void conv_split_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int tid = threadIdx.x, lane = tid & 63, m = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    // plane g: hi at g * 2 * PLANE_B, lo behind it
    if (tid < 16 * C::G) {                                      // pads: zero once
        const int g = tid >> 4, e = tid & 15;
        reinterpret_cast<unsigned*>(sp_smem + (size_t)(2 * g) * C::PLANE_B + C::IN_B)[e & 7] = 0u;
        reinterpret_cast<unsigned*>(sp_smem + (size_t)(2 * g + 1) * C::PLANE_B + C::IN_B)[e & 7] = 0u;
    }
    f16x8 wh[C::KS], wl[C::KS];
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
        wh[s] = __builtin_bit_cast(f16x8, a.w[tower][((s * C::MT + m) * 2 + 0) * 64 + lane]);
        wl[s] = __builtin_bit_cast(f16x8, a.w[tower][((s * C::MT + m) * 2 + 1) * 64 + lane]);
    }
    f32x4 bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = a.bias[tower][16 * m + 4 * j + i];
    const float c0 = a.bias[tower][16 * C::MT];
    const bool jvalid = 16 * m + 4 * j < C::COUT;
    // per-lane byte offset of this lane's pixel inside a tile family
    int rd0[C::POOL ? C::NGRP : 1], rd1[C::POOL ? C::NGRP : 1];
#pragma unroll
    for (int k = 0; k < (C::POOL ? C::NGRP : 1); ++k) sp_lane_base<C>(k, n, j, rd0[k], rd1[k]);
    float vmax = 0.f;
    const unsigned char* inp = a.in[tower];
    constexpr int SP_NTH = C::NTH, SP_PF = C::PF;
    u32x4 pf[SP_PF];
    // global -> registers (clamped, always valid addresses: straight-line loads) and registers -> LDS; macros, not lambdas: with the
    // array captured by a closure the compiler kept it in scratch memory
#define SP_FETCH(P_)                                                                                                       \
    _Pragma("unroll") for (int r = 0; r < SP_PF; ++r) {                                                                    \
        const int e_ = min(tid + r * SP_NTH, 2 * C::V * C::G - 1), g_ = e_ / (2 * C::V), v_ = e_ - g_ * 2 * C::V;         \
        pf[r] = reinterpret_cast<const u32x4*>(inp + (size_t)min((P_) + g_, a.n - 1) * 2 * C::IN_B)[v_];                   \
    }
#define SP_STASH(BASE_)                                                                                                    \
    _Pragma("unroll") for (int r = 0; r < SP_PF; ++r) {                                                                    \
        const int e_ = tid + r * SP_NTH;                                                                                   \
        if (e_ < 2 * C::V * C::G) {                                                                                        \
            const int pl_ = e_ / C::V, v_ = e_ - pl_ * C::V;                                                               \
            reinterpret_cast<u32x4*>((BASE_) + (size_t)pl_ * C::PLANE_B)[v_] = pf[r];                                      \
        }                                                                                                                  \
    }
    const int step = nwg * C::G;
    if (wg * C::G < a.n) { SP_FETCH(wg * C::G) }
    for (int p = wg * C::G; p < a.n; p += step) {
        __syncthreads();                                     // every wave is done with the planes of the previous iteration
        SP_STASH(sp_smem)
        __syncthreads();
        if (p + step < a.n) { SP_FETCH(p + step) }
        f32x4 acc[C::NT];
#pragma unroll
        for (int t = 0; t < C::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            const int soff = (s / C::KSR) * C::RS + (s % C::KSR) * 64;
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                const int g = t / C::TPP, tt = t % C::TPP;
                int off = g * 2 * C::PLANE_B + soff, k = 0;
                if constexpr (C::POOL) { k = tt >> 2; off += ((tt >> 1) & 1) * C::RS + (tt & 1) * C::PS; }
                else if constexpr (!C::SMALL) off += (tt / (C::HO / 4)) * 4 * C::RS + (tt % (C::HO / 4)) * 4 * C::PS;
                const unsigned char* h0 = sp_smem + rd0[k] + off;
                const unsigned char* h1 = sp_smem + rd1[k] + off;
                const f16x8 bh = sp_frag(h0, h1), bl = sp_frag(h0 + C::PLANE_B, h1 + C::PLANE_B);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh, acc[t], 0, 0, 0);
            }
        }
        // ---- epilogue: lane (n, j) holds channels 16 m + 4 j .. + 3 of its pixel
#pragma unroll
        for (int g = 0; g < C::G; ++g) {
            if (p + g >= a.n) break;
            constexpr int NOUT = C::POOL ? C::NGRP : C::TPP;
#pragma unroll
            for (int t = 0; t < NOUT; ++t) {
                f32x4 v;
                int px;                                          // output pixel of this lane, -1: none
                if constexpr (C::POOL) {
                    const f32x4* q = &acc[g * C::TPP + 4 * t];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(fmaxf(q[0][i], q[1][i]), fmaxf(q[2][i], q[3][i]));
                    px = sp_window<C>(t, n);
                } else {
                    v = acc[g * C::TPP + t];
                    if constexpr (C::SMALL) px = n < 9 ? n : -1;
                    else px = ((t / (C::HO / 4)) * 4 + (n >> 2)) * C::HO + (t % (C::HO / 4)) * 4 + (n & 3);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * c0 + bias[i], 0.f);
                if (px < 0 || !jvalid) continue;
                if constexpr (C::OUTF == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= kSplitActScale;
                    vmax = fmaxf(fmaxf(vmax, v[0]), v[1]); vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);
                    f16x4 h, l;
                    sp_split4(v, h, l);
                    unsigned char* o = reinterpret_cast<unsigned char*>(a.out[tower]) + (size_t)(p + g) * 2 * C::OUT_B + px * C::OUT_PS + 2 * (16 * m + 4 * j);
                    *reinterpret_cast<f16x4*>(o) = h;
                    *reinterpret_cast<f16x4*>(o + C::OUT_B) = l;
                } else if constexpr (C::OUTF == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= kSplitActScale;
                    vmax = fmaxf(fmaxf(vmax, v[0]), v[1]); vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);
                    f16x4 h, l;
                    sp_split4(v, h, l);
                    _Float16* o = reinterpret_cast<_Float16*>(reinterpret_cast<unsigned char*>(a.out[tower]) + (size_t)(p + g) * (4 * a.out_kp + 16)) + (16 * m + 4 * j) * C::OUT_PX + px;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o[i * C::OUT_PX] = h[i]; o[a.out_kp + i * C::OUT_PX] = l[i]; }
                } else {
                    float* o = reinterpret_cast<float*>(a.out[tower]) + (size_t)(p + g) * C::COUT * C::OUT_PX + (16 * m + 4 * j) * C::OUT_PX + px;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i * C::OUT_PX] = v[i];
                }
            }
        }
    }
    if (C::OUTF != 1 && vmax > kSplitOverflow) atomicOr(a.flag, 1u);
}

// ---- the same layers with ALL output tiles per wave: one wave per SIMD (512 registers), every wave holds the weights of all MT
// output tiles (MT * KS * 8 registers: 144 for conv3, 288 for conv4) and walks its own quarter of the iteration's pixel tiles, so
// every B fragment is read from LDS once.  A workgroup (4 waves) stages G patch-towers per iteration into one of two LDS buffers
// (copied global -> LDS by global_load_lds_dwordx4 under the previous iteration's matrix phase - no registers, no LDS-write pass;
// SP_GLDS = 0: through 48-64 prefetch registers per thread, the first form: conv3 26 % slower): one barrier per iteration.
template <class C>
__global__ __launch_bounds__(256, 1) void conv_split_all_kernel(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    constexpr int TPW = C::NT / 4, BUF_B = 2 * C::G * C::PLANE_B;
#if !SP_GLDS
    constexpr int NTH = 256, PF = (2 * C::V * C::G + NTH - 1) / NTH;
#endif
    static_assert(C::NT % 4 == 0 && (C::TPP % TPW == 0 || TPW % C::TPP == 0), "tiles split evenly over the four waves");
    static_assert(!C::POOL || TPW == 4, "a wave owns one pooling group");
    static_assert(C::OUTF == 0 && !C::SMALL, "split-plane outputs");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, j = lane >> 4;
    const int tower = blockIdx.x % kTowers, wg = blockIdx.x / kTowers, nwg = gridDim.x / kTowers;
    if (tid < 32 * C::G) {                                      // pads of both buffers: zero once
        const int b = tid / (16 * C::G), g = (tid >> 4) % C::G, e = tid & 7;
        reinterpret_cast<unsigned*>(sp_smem + (size_t)b * BUF_B + (size_t)(2 * g) * C::PLANE_B + C::IN_B)[e] = 0u;
        reinterpret_cast<unsigned*>(sp_smem + (size_t)b * BUF_B + (size_t)(2 * g + 1) * C::PLANE_B + C::IN_B)[e] = 0u;
    }
    f16x8 wh[C::KS][C::MT], wl[C::KS][C::MT];
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            wh[s][m] = __builtin_bit_cast(f16x8, a.w[tower][((s * C::MT + m) * 2 + 0) * 64 + lane]);
            wl[s][m] = __builtin_bit_cast(f16x8, a.w[tower][((s * C::MT + m) * 2 + 1) * 64 + lane]);
        }
    f32x4 bias[C::MT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[m][i] = a.bias[tower][16 * m + 4 * j + i];
    const float c0 = a.bias[tower][16 * C::MT];
    // this wave's tiles: t = wave * TPW + i -> patch g = t / TPP, tile tt = t % TPP
    const int t0 = wave * TPW, g0 = t0 / C::TPP;
    int rd0, rd1;
    sp_lane_base<C>(C::POOL ? (t0 % C::TPP) >> 2 : 0, n, j, rd0, rd1);
    rd0 += g0 * 2 * C::PLANE_B; rd1 += g0 * 2 * C::PLANE_B;
    float vmax = 0.f;
    const unsigned char* inp = a.in[tower];
    const int step = nwg * C::G;
    int p = wg * C::G, buf = 0;
#if SP_GLDS
    // plane pl (patch-tower g = pl / 2, piece pl & 1) of the iteration at patch P: IN_B contiguous bytes -> LDS at pl * PLANE_B, in
    // 1 KB wave instructions (the last one of a plane partly masked); the four waves take the planes' chunks round robin
    constexpr int CPP = (C::IN_B + 1023) / 1024, NCH = 2 * C::G * CPP;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto copy_planes = [&](int P, unsigned char* dstbuf) {
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 4) {
            const int c = c0 + wv;
            if (c < NCH) {
                const int pl = c / CPP, k = c - pl * CPP, off = k * 1024 + lane * 16;
                if (off < C::IN_B) {
                    const unsigned char* src = inp + (size_t)min(P + (pl >> 1), a.n - 1) * 2 * C::IN_B + (size_t)(pl & 1) * C::IN_B + off;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dstbuf + pl * C::PLANE_B + k * 1024), 16, 0, 0);
                }
            }
        }
    };
    if (p < a.n) copy_planes(p, sp_smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#else
    constexpr int SP_NTH = NTH, SP_PF = PF;
    u32x4 pf[SP_PF];
    if (p < a.n) { SP_FETCH(p) SP_STASH(sp_smem) }
    __syncthreads();
    if (p + step < a.n) { SP_FETCH(p + step) }
#endif
    for (; p < a.n; p += step) {
        const unsigned char* base = sp_smem + (size_t)buf * BUF_B;
#if SP_GLDS
        if (p + step < a.n && SP_ABLATE != 5) copy_planes(p + step, sp_smem + (size_t)(buf ^ 1) * BUF_B);
#endif
        // TB tiles per pass (POOL: the four positions of the wave's pooling group); the B fragments of k-step s + 1 are requested in
        // front of the MFMAs of k-step s and the scheduler is kept from hoisting further (fragments of ALL k-steps in flight spill)
        constexpr int TB = C::POOL ? 4 : 3;
        static_assert(TPW % TB == 0, "passes");
#pragma unroll
        for (int i0 = 0; i0 < TPW; i0 += TB) {
            f32x4 acc[TB][C::MT];
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int m = 0; m < C::MT; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#if SP_ROWREUSE
            if constexpr (C::POOL) {
                // pooling group: pixel row R (0..3) of the windows' 4 x 2 pixels is tap row R of the upper window row and tap row R - 1 of
                // the lower one: fragment f = (R KSR + ks) 2 + column parity is read ONCE and used for both (32 fragment pairs per group
                // instead of 48; conv1+conv2+pool1 does the same); three in flight
                constexpr int NF = 4 * C::KSR * 2;
                f16x8 rh[3], rl[3];
                auto rdf = [&](int f) {
                    const int off = ((f >> 1) / C::KSR) * C::RS + ((f >> 1) % C::KSR) * 64 + (f & 1) * C::PS;
                    const unsigned char* h0 = base + rd0 + off;
                    const unsigned char* h1 = base + rd1 + off;
                    rh[f % 3] = sp_frag(h0, h1);
                    rl[f % 3] = sp_frag(h0 + C::PLANE_B, h1 + C::PLANE_B);
                };
                rdf(0); rdf(1);
#pragma unroll
                for (int f = 0; f < (SP_ABLATE == 7 ? 2 : NF); ++f) {
                    const int R = (f >> 1) / C::KSR, ks = (f >> 1) % C::KSR, qc = f & 1;
                    if (f + 2 < NF) rdf(f + 2);
                    const f16x8 bh = rh[f % 3], bl = rl[f % 3];
#pragma unroll
                    for (int piece = 0; piece < 3; ++piece) {
#pragma unroll
                        for (int use = 0; use < 2; ++use) {
                            const int dy = R - use;
                            if (dy < 0 || dy > 2) continue;
                            const int s = dy * C::KSR + ks, i = 2 * use + qc;
#pragma unroll
                            for (int m = 0; m < C::MT; ++m) {
                                if (piece == 0) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], bh, acc[i][m], 0, 0, 0);
                                else if (piece == 1) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bl, acc[i][m], 0, 0, 0);
                                else acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], bh, acc[i][m], 0, 0, 0);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#endif
            {
            f16x8 fh[2][TB], fl[2][TB];
            auto frags = [&](int s, int slot, int i) {
                int off = (s / C::KSR) * C::RS + (s % C::KSR) * 64;
                if constexpr (C::POOL) off += ((i >> 1) & 1) * C::RS + (i & 1) * C::PS;
                else {
                    const int tt = (i0 + i) % C::TPP;
                    off += ((i0 + i) / C::TPP) * 2 * C::PLANE_B + (tt / (C::HO / 4)) * 4 * C::RS + (tt % (C::HO / 4)) * 4 * C::PS;
                }
                const unsigned char* h0 = base + rd0 + off;
                const unsigned char* h1 = base + rd1 + off;
                fh[slot][i] = sp_frag(h0, h1);
                fl[slot][i] = sp_frag(h0 + C::PLANE_B, h1 + C::PLANE_B);
            };
#pragma unroll
            for (int i = 0; i < TB; ++i) frags(0, 0, i);
#pragma unroll
            for (int s = 0; s < (SP_ABLATE == 7 ? 1 : C::KS); ++s) {
#pragma unroll
                for (int i = 0; i < TB; ++i) {
#pragma unroll
                    for (int m = 0; m < C::MT; ++m) {
                        acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s][m], fh[s & 1][i], acc[i][m], 0, 0, 0);
                        acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fl[s & 1][i], acc[i][m], 0, 0, 0);
                        acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s][m], fh[s & 1][i], acc[i][m], 0, 0, 0);
                    }
                    if (s + 1 < C::KS) frags(s + 1, (s + 1) & 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            // ---- epilogue of the pass
            constexpr int NOUT = C::POOL ? 1 : TB;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) {
                const int t = t0 + i0 + (C::POOL ? 0 : i), g = t / C::TPP, tt = t % C::TPP;
                int px;
                if constexpr (C::POOL) px = sp_window<C>(tt >> 2, n);
                else px = ((tt / (C::HO / 4)) * 4 + (n >> 2)) * C::HO + (tt % (C::HO / 4)) * 4 + (n & 3);
                if (p + g >= a.n) px = -1;
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
                    f32x4 v;
                    if constexpr (C::POOL) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaxf(acc[0][m][q], acc[1][m][q]), fmaxf(acc[2][m][q], acc[3][m][q]));
                    } else v = acc[i][m];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q] * c0 + bias[m][q], 0.f) * kSplitActScale;
                    if (px < 0 || 16 * m + 4 * j >= C::COUT) continue;
                    if (SP_ABLATE == 6 && v[0] != 123.25f) continue;
                    vmax = fmaxf(fmaxf(vmax, v[0]), v[1]); vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);
                    f16x4 h, l;
                    sp_split4(v, h, l);
                    unsigned char* o = reinterpret_cast<unsigned char*>(a.out[tower]) + (size_t)(p + g) * 2 * C::OUT_B + px * C::OUT_PS + 2 * (16 * m + 4 * j);
                    *reinterpret_cast<f16x4*>(o) = h;
                    *reinterpret_cast<f16x4*>(o + C::OUT_B) = l;
                }
            }
        }
#if SP_GLDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the next planes has landed
#else
        if (p + step < a.n && SP_ABLATE != 5) {
            SP_STASH(sp_smem + (size_t)(buf ^ 1) * BUF_B)
            if (p + 2 * step < a.n) { SP_FETCH(p + 2 * step) }
        }
#endif
        if (SP_ABLATE != 8) __syncthreads();
        buf ^= 1;
    }
    if (vmax > kSplitOverflow) atomicOr(a.flag, 1u);
}

using SpConv3 = SpCfg<20, 40, 14, false, 0, 4, 1>;     // 4 patch-towers per iteration: one per wave (9 tiles x 3 channel tiles)
using SpConv4 = SpCfg<40, 40, 12, true, 0, 2, 1>;      // 2 per iteration: a wave owns one pooling group (4 tiles x 3 channel tiles)
using SpConv5 = SpCfg<40, 60, 5, false, 1, 4, 2>;      // one wave per channel tile (conv_split_kernel)
using SpConv5R = SpCfg<40, 60, 5, false, 2, 4, 2>;     // the same, leaving rows of the split dense head

template <class C>
inline hipError_t conv_split_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
}
template <class C>
inline hipError_t conv_split_all_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_all_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * C::LDS_BYTES));
}

template <class C>
inline void conv_split_launch(const SplitArgs& a, hipStream_t st, int wgs_per_tower) {
    const int units = (a.n + C::G - 1) / C::G;
    const int g = wgs_per_tower < units ? wgs_per_tower : units;
    if (g < 1) return;
    hipLaunchKernelGGL(conv_split_kernel<C>, dim3(3 * g), dim3(C::NTH), C::LDS_BYTES, st, a);
}
template <class C>
inline void conv_split_all_launch(const SplitArgs& a, hipStream_t st, int wgs_per_tower) {
    const int units = (a.n + C::G - 1) / C::G;
    const int g = wgs_per_tower < units ? wgs_per_tower : units;
    if (g < 1) return;
    hipLaunchKernelGGL(conv_split_all_kernel<C>, dim3(3 * g), dim3(256), 2 * C::LDS_BYTES, st, a);
}

}  // namespace scs
