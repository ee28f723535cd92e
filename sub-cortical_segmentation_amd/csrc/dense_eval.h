// Whole-subject inference with SHARED tower work (test_scan, /root/reference/cnn_cort/base.py:372-438).
//
// The reference cuts one 32x32 patch per ROI voxel and view (load_patch_batch -> get_patches, base.py:380-397, 272-308) and runs
// the tower on each: neighbouring voxels share 31/32 of their pixels and the tower recomputes them.  The towers are stacks of
// VALID 3x3 convolutions and two 2x2 max-poolings, so the block-5 output of the patch with origin (dr, dc) in a slice is a 3x3
// window of a map that depends only on (dr mod 4, dc mod 4):
//   Z   = conv2(conv1(slice))                         one map per slice
//   P_p = pool1 of Z with window origin p             p = (dr & 1, dc & 1): 4 maps;  patch rows u0 + [0, 14), u0 = (dr - p) / 2
//   C4_p = conv4(conv3(P_p))                          patch rows u0 + [0, 10)
//   P2_pq = pool2 of C4_p with window origin q        q = (u0 & 1, ..) = ((dr >> 1) & 1, ..): 16 maps; rows a0 + [0, 5), a0 = dr >> 2
//   C5_pq = conv5(P2_pq)                              patch rows a0 + [0, 3)
// 47 k instead of 5.75 M multiply-adds per slice pixel and view.  The maps are computed with THE SAME kernels as the patch path:
// each map is covered by fixed-size tiles that are fed to the patch kernels as if they were patches - conv1+conv2+pool1 on
// 32x32 windows every 28 pixels, conv3 on 14x14 tiles every 12, conv4 on 12x12 tiles every 10, conv5 on 5x5 tiles every 3 - and
// small copy kernels re-cut the tiles between the layers (and apply pool2 with its phase).  A kernel's arithmetic for an output
// pixel does not depend on where the pixel sits in its patch, so every feature - and therefore every probability - equals the
// patch path's bit for bit (tests/test_gpu_scan.py).
//
// Views (conv_mfma.h: gather_voxel): 0 axial rows x, columns y, slice z;  1 coronal rows x, columns z, slice y;  2 sagittal rows y,
// columns z, slice x.  Zero padding outside the volume is the gather's, as in the patch path.
#pragma once
#include "conv_mfma.h"
#include <algorithm>

namespace scs {

struct DenseView {                // one view of one subject: the ROI's bounding box in view axes and the tile grids of its maps
    int view;
    int r0, c0, s0;               // smallest centre row / column / slice of the ROI
    int R, C, S;                  // extents of the box (centres)
    int t1r, t1c, t3r, t3c, t4r, t4c, t5r, t5c;   // tile grid (rows, columns) of: pool1 tiles, conv3, conv4, conv5 tiles
    __host__ __device__ int n1(int sb) const { return sb * 4 * t1r * t1c; }
    __host__ __device__ int n3(int sb) const { return sb * 4 * t3r * t3c; }
    __host__ __device__ int n4(int sb) const { return sb * 4 * t4r * t4c; }
    __host__ __device__ int n5(int sb) const { return sb * 16 * t5r * t5c; }
};

// Tile counts along one axis of extent E (centres): the conv5 map needs rows a0 + [0, 3) for a0 <= (E - 1) >> 2, and every
// earlier layer needs what the tiles of the next one read.
inline void dense_axis(int E, int& t1, int& t3, int& t4, int& t5) {
    const int A = ((E - 1) >> 2) + 1;                        // values of a0
    const int U = ((E - 1) >> 1) + 1;                        // values of u0
    t5 = (A + 2 + 2) / 3;                                    // conv5 output rows [0, A + 2)
    t4 = std::max((U + 10 + 9) / 10, (6 * t5 + 5 + 9) / 10); // conv4 output rows: what the patches need, and 2 (3 t5 + 1) + 1 + 1 for pool2
    t3 = (10 * t4 + 2 + 11) / 12;                            // conv3 output rows [0, 10 t4 + 2)
    t1 = (12 * t3 + 2 + 13) / 14;                            // pool1 rows [0, 12 t3 + 2)
}

inline DenseView dense_view(int view, const int box[6]) {   // box: xmin, xmax, ymin, ymax, zmin, zmax of the centres
    DenseView v{};
    v.view = view;
    const int lo[3] = {box[0], box[2], box[4]}, hi[3] = {box[1], box[3], box[5]};
    const int ra = view == 2 ? 1 : 0, ca = view == 0 ? 1 : 2, sa = view == 0 ? 2 : (view == 1 ? 1 : 0);
    v.r0 = lo[ra]; v.c0 = lo[ca]; v.s0 = lo[sa];
    v.R = hi[ra] - lo[ra] + 1; v.C = hi[ca] - lo[ca] + 1; v.S = hi[sa] - lo[sa] + 1;
    dense_axis(v.R, v.t1r, v.t3r, v.t4r, v.t5r);
    dense_axis(v.C, v.t1c, v.t3c, v.t4c, v.t5c);
    return v;
}

// floats of tile buffers one batch of `sb` slices needs: pool1 tiles | conv3 inputs | conv3 outputs | conv4 inputs | conv4 outputs |
// conv5 inputs | conv5 outputs, and the int32 centres of the pool1 tiles
struct DenseBuffers { size_t t1, i3, o3, i4, o4, i5, o5, ctr, total; };
inline DenseBuffers dense_buffers(const DenseView& v, int sb) {
    auto al = [](size_t f) { return (f + 63) / 64 * 64; };
    DenseBuffers b{};
    size_t off = 0;
    b.t1 = off; off += al((size_t)v.n1(sb) * 20 * 196);
    b.i3 = off; off += al((size_t)v.n3(sb) * 20 * 196);
    b.o3 = off; off += al((size_t)v.n3(sb) * 40 * 144);
    b.i4 = off; off += al((size_t)v.n4(sb) * 40 * 144);
    b.o4 = off; off += al((size_t)v.n4(sb) * 40 * 100);
    b.i5 = off; off += al((size_t)v.n5(sb) * 40 * 25);
    b.o5 = off; off += al((size_t)v.n5(sb) * 540);
    b.ctr = off; off += al((size_t)v.n1(sb) * 3);
    b.total = off;
    return b;
}

// ---- kernels.  Tile index = ((slice * PH + phase) * tr + i) * tc + j, phase = 2 * pr + pc (pool1) or 4 * (2 pr + pc) + 2 qr + qc.

// centres of the 32x32 windows of the pool1 tiles: window origin (28 i + pr, 28 j + pc) in the canvas whose origin is the
// top-left pixel of the patch around (r0, c0) -> centre (r0 + 28 i + pr, c0 + 28 j + pc)
__global__ void dense_centres_kernel(DenseView v, int s_begin, int sb, int32_t* __restrict__ ctr) {
    const int n = v.n1(sb);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        int q = t;
        const int j = q % v.t1c; q /= v.t1c;
        const int i = q % v.t1r; q /= v.t1r;
        const int ph = q & 3, s = q >> 2;
        const int r = v.r0 + 28 * i + (ph >> 1), c = v.c0 + 28 * j + (ph & 1), sl = v.s0 + s_begin + s;
        int x, y, z;
        if (v.view == 0) { x = r; y = c; z = sl; }
        else if (v.view == 1) { x = r; z = c; y = sl; }
        else { y = r; z = c; x = sl; }
        ctr[3 * t] = x; ctr[3 * t + 1] = y; ctr[3 * t + 2] = z;
    }
}

// The two re-tiling kernels below work one DESTINATION tile per workgroup pass: the divisions by run-time tile counts are per
// tile, the per-element ones are by constants (as one grid-stride loop over single elements with 64-bit indices they took 2x / 1.3x
// as long).  The features kernel stays element-wise: a wave per voxel, and a wave testing 64 voxels then copying its hits one
// after the other, were both slower (136 and 236 us per launch against 125: the hits of a slice batch are neighbours in the list).

// Re-cut a map held as tiles [..][CH][HS][HS] that tessellate it with stride HS into tiles [..][CH][HD][HD] taken every SD rows
// and columns (the inputs of the next valid convolution: HD = SD + 2).  Same slices and phases on both sides.
template <int CH, int HS, int HD, int SD>
__global__ __launch_bounds__(256) void dense_retile_kernel(const float* __restrict__ src, int trs, int tcs, float* __restrict__ dst, int trd, int tcd,
                                                           int planes /* slices x phases */) {
    constexpr int TILE = CH * HD * HD;
    const unsigned tiles = (unsigned)planes * trd * tcd;
    for (unsigned T = blockIdx.x; T < tiles; T += gridDim.x) {
        const unsigned j = T % (unsigned)tcd, q = T / (unsigned)tcd, i = q % (unsigned)trd, pl = q / (unsigned)trd;
        const float* __restrict__ sp = src + (size_t)pl * trs * tcs * CH * HS * HS;
        float* __restrict__ d = dst + (size_t)T * TILE;
        for (int e = threadIdx.x; e < TILE; e += 256) {
            const int x = e % HD, y = (e / HD) % HD, c = e / (HD * HD);
            const int u = SD * (int)i + y, w = SD * (int)j + x;
            const int ti = u / HS, tj = w / HS;
            d[e] = sp[(((size_t)ti * tcs + tj) * CH + c) * HS * HS + (u - ti * HS) * HS + (w - tj * HS)];
        }
    }
}

// conv4 outputs (tiles [slices x 4][t4r][t4c][40][10][10], post-ReLU) -> pool2 with window origin q -> conv5 input tiles
// [slices x 16][t5r][t5c][40][5][5] taken every 3 rows / columns of the pooled map
__global__ __launch_bounds__(256) void dense_pool_retile_kernel(const float* __restrict__ src, int t4r, int t4c, float* __restrict__ dst, int t5r, int t5c,
                                                                int slices) {
    const unsigned tiles = (unsigned)slices * 16u * t5r * t5c;
    for (unsigned T = blockIdx.x; T < tiles; T += gridDim.x) {
        const unsigned j = T % (unsigned)t5c, k1 = T / (unsigned)t5c, i = k1 % (unsigned)t5r, k = k1 / (unsigned)t5r;
        const int ph = (int)(k & 15u), sl = (int)(k >> 4);
        const int p = ph >> 2, qr = (ph >> 1) & 1, qc = ph & 1;
        const float* __restrict__ base = src + ((size_t)(sl * 4 + p) * t4r * t4c) * 4000;
        float* __restrict__ d = dst + (size_t)T * 1000;
        for (int e = threadIdx.x; e < 1000; e += 256) {
            const int x = e % 5, y = (e / 5) % 5, c = e / 25;
            const int a = 3 * (int)i + y, b = 3 * (int)j + x;
            float m = -3.0e38f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int u = 2 * a + qr + dy, w = 2 * b + qc + dx;
                    const int ti = u / 10, tj = w / 10;
                    m = fmaxf(m, base[((size_t)ti * t4c + tj) * 4000 + c * 100 + (u - ti * 10) * 10 + (w - tj * 10)]);
                }
            d[e] = m;
        }
    }
}

// block-5 features of the ROI voxels whose slice lies in [s_begin, s_begin + sb): 60 x 3 x 3 window of the conv5 map of their phase
// -> feat[k][540] (c, h, w order: what d1 contracts, nets.py:176-179)
__global__ __launch_bounds__(256) void dense_features_kernel(DenseView v, int s_begin, int sb, const float* __restrict__ t5 /*[..][540]*/,
                                                             const int32_t* __restrict__ centers, long n, float* __restrict__ feat) {
    const long total = n * 60;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long k = e / 60;
        const int c = (int)(e - k * 60);
        const int x = centers[3 * k], y = centers[3 * k + 1], z = centers[3 * k + 2];
        const int r = v.view == 2 ? y : x, cc = v.view == 0 ? y : z, sl = v.view == 0 ? z : (v.view == 1 ? y : x);
        const int s = sl - v.s0 - s_begin;
        if (s < 0 || s >= sb) continue;
        const int dr = r - v.r0, dc = cc - v.c0;
        if ((unsigned)dr >= (unsigned)v.R || (unsigned)dc >= (unsigned)v.C) continue;   // a centre outside the caller's box: no read
        const int ph = 4 * (2 * (dr & 1) + (dc & 1)) + 2 * ((dr >> 1) & 1) + ((dc >> 1) & 1);
        const int a0 = dr >> 2, b0 = dc >> 2;
        const float* __restrict__ base = t5 + ((size_t)(s * 16 + ph) * v.t5r * v.t5c) * 540 + c * 9;
        float* __restrict__ o = feat + (size_t)k * 540 + c * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int a = a0 + i, b = b0 + j;
                const int ti = a / 3, tj = b / 3;
                o[3 * i + j] = base[((size_t)ti * v.t5c + tj) * 540 + (a - ti * 3) * 3 + (b - tj * 3)];
            }
    }
}

inline dim3 dense_grid(long total) { return dim3((unsigned)std::min<long>((total + 255) / 256, 65536)); }
inline dim3 dense_tile_grid(long tiles) { return dim3((unsigned)std::max<long>(1, std::min<long>(tiles, 1 << 20))); }   // one workgroup per tile

}  // namespace scs
