/* Plain-C host of libscs.so: no Python, no torch, no HIP headers - only include/scs.h.
 *   abi_host <weights.f32> <n> <out.bin>
 * Fills n tri-view patches + prior rows from a fixed LCG (the pytest side generates the same numbers with NumPy),
 * runs scs_forward_host and writes float32 probs [n][15] followed by int32 labels [n] to <out.bin>.
 * Built and run by tests/test_gpu_cabi.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "scs.h"

static uint32_t lcg_state = 12345u;
static float lcg_unit(void) {                     /* uniform in [0, 1) with 24 random bits */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) / 16777216.0f;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.f32 n out.bin\n", argv[0]); return 2; }
    const long n = atol(argv[2]);
    float* arena = (float*)malloc(sizeof(float) * SCS_N_PARAMS);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(arena, sizeof(float), SCS_N_PARAMS, f) != SCS_N_PARAMS) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    float* views[3];
    for (int v = 0; v < 3; ++v) {
        views[v] = (float*)malloc(sizeof(float) * 1024 * (size_t)n);
        for (size_t i = 0; i < 1024 * (size_t)n; ++i) views[v][i] = 4.0f * lcg_unit() - 2.0f;
    }
    float* atlas = (float*)malloc(sizeof(float) * 15 * (size_t)n);
    for (long r = 0; r < n; ++r) {
        float s = 0.f;
        for (int c = 0; c < 15; ++c) { atlas[r * 15 + c] = lcg_unit(); s += atlas[r * 15 + c]; }
        for (int c = 0; c < 15; ++c) atlas[r * 15 + c] /= s;
    }
    float* probs = (float*)malloc(sizeof(float) * 15 * (size_t)n);
    int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);

    scs_handle h = NULL;
    if (scs_abi_version() < 1) { fprintf(stderr, "bad ABI version\n"); return 1; }
    if (scs_create(0, &h) != SCS_OK) { fprintf(stderr, "scs_create: %s\n", scs_last_error()); return 1; }
    /* call order is enforced: forward before set_params is an error, not a crash */
    if (scs_forward_host(h, views[0], views[1], views[2], atlas, n, probs, labels) != SCS_ERR_STATE) { fprintf(stderr, "expected SCS_ERR_STATE\n"); return 1; }
    if (scs_set_params(h, arena, SCS_N_PARAMS) != SCS_OK) { fprintf(stderr, "scs_set_params: %s\n", scs_last_error()); return 1; }
    if (scs_forward_host(h, views[0], views[1], views[2], atlas, n, probs, labels) != SCS_OK) { fprintf(stderr, "scs_forward_host: %s\n", scs_last_error()); return 1; }
    scs_destroy(h);

    f = fopen(argv[3], "wb");
    if (!f) return 2;
    fwrite(probs, sizeof(float), 15 * (size_t)n, f);
    fwrite(labels, sizeof(int32_t), (size_t)n, f);
    fclose(f);
    printf("%s: %ld patches on %s through the C ABI v%d\n", argv[0], n, scs_target_arch(), scs_abi_version());
    return 0;
}
