/* Plain-C training host of libscs.so: include/scs.h plus the HIP runtime's C API for device buffers - what a maintainer of
 * a C / C++ / cgo host writes; no Python, no torch.
 *   abi_train <weights.f32> <n> <steps> <out.bin>
 * Fills n tri-view patches, prior rows and labels from a fixed LCG (the pytest side generates the same numbers with NumPy),
 * uploads them, runs <steps> x (scs_train_step with the counter-based dropout masks of seed 1000 + step, scs_adam_apply with
 * the reference's Adam constants) and writes the float32 losses [steps] followed by the updated parameter arena
 * [SCS_N_PARAMS] to <out.bin>.  Built and run by tests/test_gpu_cabi.py. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "scs.h"

static uint32_t lcg_state = 4242u;
static float lcg_unit(void) {                     /* uniform in [0, 1) with 24 random bits */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) / 16777216.0f;
}
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define SCS_OK_(x) do { int rc_ = (x); if (rc_ != SCS_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, scs_last_error()); return 1; } } while (0)

static void* upload(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s weights.f32 n steps out.bin\n", argv[0]); return 2; }
    const long n = atol(argv[2]), steps = atol(argv[3]);
    float* arena = (float*)malloc(sizeof(float) * SCS_N_PARAMS);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(arena, sizeof(float), SCS_N_PARAMS, f) != SCS_N_PARAMS) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    HIP_OK(hipSetDevice(0));
    float* views_dev[3];
    float* tmp = (float*)malloc(sizeof(float) * 1024 * (size_t)n);
    for (int v = 0; v < 3; ++v) {
        for (size_t i = 0; i < 1024 * (size_t)n; ++i) tmp[i] = 4.0f * lcg_unit() - 2.0f;
        views_dev[v] = (float*)upload(tmp, sizeof(float) * 1024 * (size_t)n);
        if (!views_dev[v]) { fprintf(stderr, "upload failed\n"); return 1; }
    }
    float* atlas = (float*)malloc(sizeof(float) * 15 * (size_t)n);
    uint8_t* y = (uint8_t*)malloc((size_t)n);
    for (long r = 0; r < n; ++r) {
        float s = 0.f;
        for (int c = 0; c < 15; ++c) { atlas[r * 15 + c] = lcg_unit(); s += atlas[r * 15 + c]; }
        for (int c = 0; c < 15; ++c) atlas[r * 15 + c] /= s;
        y[r] = (uint8_t)(lcg_unit() * 15.0f);
    }
    float* atlas_dev = (float*)upload(atlas, sizeof(float) * 15 * (size_t)n);
    uint8_t* y_dev = (uint8_t*)upload(y, (size_t)n);
    if (!atlas_dev || !y_dev) { fprintf(stderr, "upload failed\n"); return 1; }

    scs_handle h = NULL;
    SCS_OK_(scs_create(0, &h));
    SCS_OK_(scs_set_params(h, arena, SCS_N_PARAMS));
    float* losses = (float*)malloc(sizeof(float) * (size_t)steps);
    for (long s = 0; s < steps; ++s) {
        SCS_OK_(scs_train_step(h, views_dev[0], views_dev[1], views_dev[2], atlas_dev, y_dev, n, 1000u + (uint64_t)s, NULL, &losses[s], NULL));
        SCS_OK_(scs_adam_apply(h, 1e-3f, 0.9f, 0.999f, 1e-8f, 1.0f, NULL));
    }
    SCS_OK_(scs_get_params(h, arena, SCS_N_PARAMS));
    SCS_OK_(scs_destroy(h));
    f = fopen(argv[4], "wb");
    if (!f || fwrite(losses, sizeof(float), (size_t)steps, f) != (size_t)steps || fwrite(arena, sizeof(float), SCS_N_PARAMS, f) != SCS_N_PARAMS) {
        fprintf(stderr, "cannot write %s\n", argv[4]);
        return 2;
    }
    fclose(f);
    printf("abi_train ok: %ld steps of %ld samples, last loss %.6f\n", steps, n, losses[steps - 1]);
    return 0;
}
