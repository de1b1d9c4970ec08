/* developer tool (survey / build container only: needs oracle/_ref built from /root/reference): times the reference's own SAD and
 * overlap-add object code (scalar C and AVX2 builds) against the oracle's scalar restatement on this host, single thread, so that
 * bench.py's cpu_baseline ("kind: port") can be read in "reference SIMD" terms.  BASELINE.md section 3.3 quotes its output.
 *   gcc -O2 tools/cpu_calibrate.c -o /tmp/cpu_calibrate -ldl && /tmp/cpu_calibrate oracle/libmvoracle.so oracle/_ref/libmvref.so */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned (*ref_sad_t)(int, int, int, int, const uint8_t *, intptr_t, const uint8_t *, intptr_t);
typedef unsigned (*mvo_sad_t)(int, int, int, const uint8_t *, intptr_t, const uint8_t *, intptr_t);
typedef void (*ref_ov_t)(int, int, int, int, uint8_t *, intptr_t, const uint8_t *, intptr_t, int16_t *, intptr_t);
typedef void (*mvo_ov_t)(int, int, int, uint8_t *, intptr_t, const uint8_t *, intptr_t, int16_t *, intptr_t);
typedef int (*ref_has_t)(int, int, int);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s libmvoracle.so libmvref.so\n", argv[0]); return 2; }
    void *o = dlopen(argv[1], RTLD_NOW), *r = dlopen(argv[2], RTLD_NOW);
    if (!o || !r) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    mvo_sad_t osad = (mvo_sad_t)dlsym(o, "mvo_sad");
    ref_sad_t rsad = (ref_sad_t)dlsym(r, "ref_sad");
    mvo_ov_t oov = (mvo_ov_t)dlsym(o, "mvo_overlaps");
    ref_ov_t rov = (ref_ov_t)dlsym(r, "ref_overlaps");
    ref_has_t has_sad = (ref_has_t)dlsym(r, "ref_has_sad_avx2");
    const int pitch = 8192;
    uint8_t *a = malloc(pitch * 256), *b = malloc(pitch * 256), *acc = calloc(pitch * 256, 4);
    int16_t win[128 * 128];
    for (int i = 0; i < pitch * 256; i++) { a[i] = (uint8_t)(i * 7 + (i >> 9)); b[i] = (uint8_t)(i * 13 + (i >> 7)); }
    for (int i = 0; i < 128 * 128; i++) win[i] = (int16_t)(i & 2047);
    const int cases[][3] = { { 8, 8, 8 }, { 16, 16, 8 }, { 8, 8, 16 }, { 16, 16, 16 }, { 32, 32, 16 } };
    printf("%-22s %12s %12s %12s   %s\n", "kernel", "oracle ns", "ref C ns", "ref AVX2 ns", "oracle / ref-best");
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); c++) {
        const int w = cases[c][0], h = cases[c][1], bits = cases[c][2], N = 400000;
        volatile unsigned sink = 0;
        double t[3];
        for (int v = 0; v < 3; v++) {
            if (v == 2 && !has_sad(w, h, bits)) { t[2] = 0; continue; } /* no AVX2 kernel for this size in SADFunctions_AVX2.cpp */
            const double t0 = now();
            for (int i = 0; i < N; i++) {
                const uint8_t *s = a + (i & 63) * 64, *q = b + (i & 31) * 2 + ((i >> 5) & 15) * pitch;
                sink += v == 0 ? osad(w, h, bits, s, pitch, q, pitch) : rsad(w, h, bits, v == 2, s, pitch, q, pitch);
            }
            t[v] = (now() - t0) / N * 1e9;
        }
        char name[64]; snprintf(name, sizeof(name), "SAD %dx%d %d-bit", w, h, bits);
        const double best = (t[2] > 0 && t[2] < t[1]) ? t[2] : t[1];
        printf("%-22s %12.1f %12.1f %12.1f   %.2fx\n", name, t[0], t[1], t[2], t[0] / best);
        for (int v = 0; v < 3; v++) {
            if (v == 2 && bits > 8) { t[2] = 0; continue; } /* Overlap_AVX2.cpp has 8-bit kernels only (MVDegrains.cpp:362) */
            const double t0 = now();
            for (int i = 0; i < N / 4; i++) {
                uint8_t *d = acc + ((i & 15) * 32) * 4;
                const uint8_t *s = a + (i & 63) * 64;
                if (v == 0) oov(w, h, bits, d, pitch, s, pitch, win, w); else rov(w, h, bits, v == 2, d, pitch, s, pitch, win, w);
            }
            t[v] = (now() - t0) / (N / 4) * 1e9;
        }
        snprintf(name, sizeof(name), "overlaps %dx%d %d-bit", w, h, bits);
        const double best2 = (t[2] > 0 && t[2] < t[1]) ? t[2] : t[1];
        printf("%-22s %12.1f %12.1f %12.1f   %.2fx\n", name, t[0], t[1], t[2], t[0] / best2);
    }
    return 0;
}
