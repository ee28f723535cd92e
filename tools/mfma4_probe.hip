// Probe of v_mfma_f32_4x4x1_16b_f32 lane maps (integer-valued data, asymmetric operands):
// assumed: block b = lanes 4b..4b+3; A_b[i] = a[lane 4b+i]; B_b[j] = b[lane 4b+j]; D_b[i][j] -> lane 4b+j, vgpr i.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], c, 0, 0, 0);     // accumulate a second k
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}
int main() {
    float ha[128], hb[128], hd[256];
    for (int l = 0; l < 128; ++l) { ha[l] = 1 + (l * 7) % 13; hb[l] = 2 + (l * 5) % 11; }
    float *a, *b, *d;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&d, sizeof(hd));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int blk = l / 4, j = l % 4;
            const float exp = ha[4 * blk + i] * hb[4 * blk + j] + ha[64 + 4 * blk + i] * hb[64 + 4 * blk + j];
            if (hd[l * 4 + i] != exp) { if (bad < 8) printf("lane %d vgpr %d: got %g expected %g\n", l, i, hd[l * 4 + i], exp); ++bad; }
        }
    printf("mfma_f32_4x4x1 lane-map probe: %d mismatches of 256\n", bad);
    return bad != 0;
}
