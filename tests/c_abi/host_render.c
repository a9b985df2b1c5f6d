/* host_render.c -- a C host of the C-ABI (include/sonicsim_hip.h), no Python and no torch in the process.
 * The boundary exists so that a non-Python host can bind it: this program renders two small moving-source cases through
 * ss_convolve_moving_seg_f32 with HOST pointers (pageable memory, then a pinned output from ss_host_alloc), checks them against a
 * double-precision direct-form evaluation of the reference formula (SonicSim_moving.py:86-94: y[c,t] = (1-w) (x*h[k,c])[t] + w (x*h[k+1,c])[t],
 * w = i / n_k inside segment k, :42-45) and exercises the error convention (negative code + ss_last_error()).
 * Build: gcc -O2 -std=c99 tests/c_abi/host_render.c -Iinclude -Lsonicsim_amd/lib -lsonicsim_hip -lm -Wl,-rpath,$PWD/sonicsim_amd/lib
 * Test infrastructure: built and run by tests/test_c_abi.py (run needs a GPU). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sonicsim_hip.h"

static unsigned long long rng_state = 88172645463325252ULL;
static double rnd(void) { /* xorshift64*, uniform in (-1, 1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ULL) >> 11) / 4503599627370496.0 - 1.0;
}

static int run_case(int64_t T, int P, int C, int L, int pinned_out) {
    float* x = (float*)malloc(sizeof(float) * T);
    float* h = (float*)malloc(sizeof(float) * (size_t)P * C * L);
    int64_t* seg = (int64_t*)malloc(sizeof(int64_t) * (P - 1));
    double* ref = (double*)calloc((size_t)C * T, sizeof(double));
    float* y = NULL;
    if (pinned_out) {
        if (ss_host_alloc((void**)&y, (int64_t)sizeof(float) * C * T)) { fprintf(stderr, "ss_host_alloc: %s\n", ss_last_error()); return 1; }
    } else {
        y = (float*)malloc(sizeof(float) * (size_t)C * T);
    }
    for (int64_t t = 0; t < T; ++t) x[t] = (float)(0.1 * rnd());
    for (int64_t i = 0; i < (int64_t)P * C * L; ++i) h[i] = (float)(rnd() * exp(-4.0 * (double)(i % L) / L));
    int64_t left = T;                                   /* ragged segments, one of them empty when P allows */
    for (int k = 0; k < P - 1; ++k) {
        int64_t n = (k == P - 2) ? left : (int64_t)((0.5 + 0.5 * (rnd() + 1.0)) * T / P);
        if (k == 1 && P > 3) n = 0;
        if (n > left) n = left;
        seg[k] = n;
        left -= n;
    }
    int64_t s0 = 0;
    for (int k = 0; k < P - 1; ++k) {                  /* direct form, double precision */
        const int64_t n = seg[k];
        for (int64_t i = 0; i < n; ++i) {
            const int64_t t = s0 + i;
            const double w = (double)(float)((double)i * (1.0 / (double)n));     /* linspace(0, 1, n, endpoint=False).astype(float32) */
            for (int c = 0; c < C; ++c) {
                const float* h0 = h + ((size_t)k * C + c) * L;
                const float* h1 = h + ((size_t)(k + 1) * C + c) * L;
                double a0 = 0.0, a1 = 0.0;
                const int64_t m = t + 1 < L ? t + 1 : L;
                for (int64_t u = 0; u < m; ++u) { a0 += (double)h0[u] * x[t - u]; a1 += (double)h1[u] * x[t - u]; }
                ref[(size_t)c * T + t] = (1.0 - w) * a0 + w * a1;
            }
        }
        s0 += n;
    }
    memset(y, 0xff, sizeof(float) * (size_t)C * T);    /* NaN pattern: every sample must be written */
    const int rc = ss_convolve_moving_seg_f32(x, T, h, P, C, L, seg, y, 0 /* host pointers */, NULL);
    if (rc) { fprintf(stderr, "ss_convolve_moving_seg_f32 -> %d: %s\n", rc, ss_last_error()); return 1; }
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < (size_t)C * T; ++i) { const double d = (double)y[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
    const double rel = sqrt(num / (den > 0 ? den : 1.0));
    printf("case T=%lld P=%d C=%d L=%d %s output: rel RMS vs double-precision direct form %.3e\n", (long long)T, P, C, L, pinned_out ? "pinned" : "pageable", rel);
    free(x); free(h); free(seg); free(ref);
    if (pinned_out) ss_host_free(y); else free(y);
    return !(rel <= 1e-4);                              /* the north star's gate */
}

int main(void) {
    if (ss_version() != SS_VERSION) { fprintf(stderr, "header / library version mismatch\n"); return 2; }
    int bad = 0;
    bad |= run_case(5000, 4, 2, 100, 0);                /* direct-form engine */
    bad |= run_case(30000, 5, 2, 9000, 0);              /* assembly engine, staged through the pinned rings */
    bad |= run_case(30000, 5, 2, 9000, 1);              /* the same into pinned memory: direct DMA */
    /* error convention: a negative code and a message, nothing thrown across the boundary */
    float xx[8] = {0}, hh[2 * 1 * 4] = {0}, yy[8];
    int64_t neg[1] = {-3};
    const int rc = ss_convolve_moving_seg_f32(xx, 8, hh, 2, 1, 4, neg, yy, 0, NULL);
    if (rc != SS_EINVAL || strstr(ss_last_error(), "negative") == NULL) { fprintf(stderr, "error convention: rc=%d msg=%s\n", rc, ss_last_error()); bad = 1; }
    printf(bad ? "FAILED\n" : "C host: all cases within 1e-4, error convention ok\n");
    ss_shutdown();
    return bad;
}
