/* asan_check.c -- runs every oracle entry point on a small random scene under
 * AddressSanitizer + UBSan (CPU only; GPU ASan is not available on this pool).
 * Built and run by tests/test_sanitizers_cpu.py.  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arvx_oracle.h"

static unsigned rng_state = 12345u;
static unsigned rnd(void) { return rng_state = rng_state * 1664525u + 1013904223u; }
static float frand(void) { return (float)(rnd() >> 8) / 16777216.0f; }

int main(void) {
    enum { X = 13, Y = 9, Z = 7, V = 3, W = 37, H = 23, C = 3 };
    const long N = (long)X * Y * Z, stride = W * C + 5;
    const float s = 0.04f;
    float K[9] = {30.f, 0, 18.f, 0, 30.f, 11.f, 0, 0, 1.f};
    float Rt[V * 12], M[V * 12], campos[V * 3];
    for (int v = 0; v < V; ++v) {
        for (int i = 0; i < 12; ++i) Rt[12 * v + i] = frand() - 0.5f;
        Rt[12 * v + 11] += 1.0f;
        arvx_oracle_compose(K, Rt + 12 * v, M + 12 * v);
        campos[3 * v] = Rt[12 * v + 3];
        campos[3 * v + 1] = Rt[12 * v + 7];
        campos[3 * v + 2] = Rt[12 * v + 11];
    }
    uint8_t *masks = malloc((size_t)V * H * stride), *images = malloc((size_t)V * H * W * 3);
    for (long i = 0; i < (long)V * H * stride; ++i) masks[i] = (rnd() & 3) ? 0 : (uint8_t)rnd();
    for (long i = 0; i < (long)V * H * W * 3; ++i) images[i] = (uint8_t)rnd();
    uint8_t *st = malloc(N), *st2 = malloc(N), *st3 = malloc(N);
    memset(st, 1, N);
    memset(st2, 1, N);
    memset(st3, 1, N);
    arvx_oracle_carve(X, Y, Z, s, V, M, masks, W, H, C, stride, st);
    arvx_oracle_carve_mt(X, Y, Z, s, V, M, masks, W, H, C, stride, st2, 2);
    if (memcmp(st, st2, N)) { puts("carve_mt != carve"); return 1; }
    arvx_oracle_fast_carve(X, Y, Z, s, V, M, masks, W, H, C, stride, st3);
    float *rgba = malloc((size_t)N * 16), *ref = malloc((size_t)N * 16);
    uint64_t *seen = calloc((N + 63) / 64, 8);
    arvx_oracle_model_init(ref, N);
    arvx_oracle_carve_ref(X, Y, Z, s, V, K, Rt, masks, W, H, C, stride, ref, seen, 0, Z);
    for (long i = 0; i < N; ++i)
        if ((ref[4 * i + 3] != 0) != ((st[i] & 1) != 0)) { puts("carve_ref != carve"); return 1; }
    arvx_oracle_state_to_model(st, rgba, N);
    arvx_oracle_color(X, Y, Z, s, V, M, campos, images, W, H, (long)W * 3, 0, rgba);
    arvx_oracle_color(X, Y, Z, s, V, M, campos, images, W, H, (long)W * 3, 1, rgba);
    arvx_oracle_handle_unseen(st, rgba, N);
    arvx_oracle_closure(X, Y, Z, rgba);
    arvx_oracle_model_to_state(rgba, st2, N);
    float out[5];
    int px, py;
    arvx_oracle_project_raw(M, s, X - 1, Y - 1, Z - 1, out);
    (void)arvx_oracle_project(M, s, 0, 0, 0, W, H, &px, &py);
    (void)arvx_oracle_depth(campos, s, 1, 2, 3);
    free(masks); free(images); free(st); free(st2); free(st3); free(rgba); free(ref); free(seen);
    puts("asan_check ok");
    return 0;
}
