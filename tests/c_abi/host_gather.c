/* host_gather.c -- two PROCESSES of a plain C host gather rendered scenes through the C-ABI's CU-free gather (include/sonicsim_hip.h:
 * ss_gather_create / attach / slot / put / flush / close), no Python in either process.  What SonicSet.py:183-211 does in one serial loop,
 * sharded: the root exports its result array as a HIP IPC handle (64 bytes, handed over through a file here -- any control plane does),
 * the peer opens it, renders its scenes and copies each into its slot with the copy engines; the root renders its own scenes in place,
 * then re-renders EVERY scene and compares bits.
 *   host_gather root <dir>      host_gather peer <dir>        (run both; <dir> is an empty scratch directory)
 * Build: gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ tests/c_abi/host_gather.c -Iinclude -I/opt/rocm/include -Lsonicsim_amd/lib -lsonicsim_hip
 *        -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/sonicsim_amd/lib
 * Test infrastructure: built and run by tests/test_c_abi.py (needs a GPU). */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "sonicsim_hip.h"

enum { NUM = 5, T = 20000, P = 4, C = 2, L = 5000 };
static const int64_t SCENE_BYTES = (int64_t)sizeof(float) * C * T;

static unsigned long long rng_state;
static double rnd(void) {
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ULL) >> 11) / 4503599627370496.0 - 1.0;
}

/* scene s, rendered through the host-pointer entry point into y[C][T] */
static int render_scene(int s, float* y) {
    static float x[T], h[P * C * L];
    int64_t seg[P - 1] = {T / 2, T / 5, T - T / 2 - T / 5};
    rng_state = 88172645463325252ULL + 7919ULL * (unsigned long long)(s + 1);
    for (int t = 0; t < T; ++t) x[t] = (float)(0.1 * rnd());
    for (int i = 0; i < P * C * L; ++i) h[i] = (float)(rnd() * (1.0 - (double)(i % L) / L));
    const int rc = ss_convolve_moving_seg_f32(x, T, h, P, C, L, seg, y, 0 /* host pointers */, NULL);
    if (rc) fprintf(stderr, "render of scene %d failed: %s\n", s, ss_last_error());
    return rc;
}

static int wait_for(const char* path) {
    for (int i = 0; i < 6000; ++i) {                     /* <= 60 s */
        if (access(path, F_OK) == 0) return 0;
        struct timespec ts = {0, 10 * 1000 * 1000};
        nanosleep(&ts, NULL);
    }
    fprintf(stderr, "timed out waiting for %s\n", path);
    return 1;
}

static int touch(const char* path) { FILE* f = fopen(path, "wb"); if (!f) return 1; fclose(f); return 0; }

#define HIPOK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: host_gather root|peer <dir>\n"); return 2; }
    const int root = strcmp(argv[1], "root") == 0;
    char hpath[512], tpath[512], pdone[512], rdone[512];
    snprintf(hpath, sizeof hpath, "%s/handle.bin", argv[2]);
    snprintf(tpath, sizeof tpath, "%s/handle.tmp", argv[2]);
    snprintf(pdone, sizeof pdone, "%s/peer.done", argv[2]);
    snprintf(rdone, sizeof rdone, "%s/root.done", argv[2]);
    if (ss_init(0)) { fprintf(stderr, "ss_init: %s\n", ss_last_error()); return 1; }
    float* y = (float*)malloc((size_t)SCENE_BYTES);
    void* g = NULL;
    unsigned char ipc[SS_IPC_HANDLE_BYTES];
    /* scenes 0, 2, 4 belong to the root, 1 and 3 to the peer */
    if (root) {
        if (ss_gather_create(&g, NUM, SCENE_BYTES, ipc)) { fprintf(stderr, "ss_gather_create: %s\n", ss_last_error()); return 1; }
        FILE* f = fopen(tpath, "wb");
        if (!f || fwrite(ipc, 1, sizeof ipc, f) != sizeof ipc) return 1;
        fclose(f);
        if (rename(tpath, hpath)) return 1;
        for (int s = 0; s < NUM; s += 2) {               /* the root's own scenes: straight into their slots */
            void* slot = NULL;
            if (render_scene(s, y) || ss_gather_slot(g, s, &slot)) return 1;
            HIPOK(hipMemcpy(slot, y, (size_t)SCENE_BYTES, hipMemcpyHostToDevice));
        }
        if (wait_for(pdone)) return 1;
        int bad = 0;
        float* got = (float*)malloc((size_t)SCENE_BYTES);
        for (int s = 0; s < NUM; ++s) {
            void* slot = NULL;
            if (render_scene(s, y) || ss_gather_slot(g, s, &slot)) return 1;
            HIPOK(hipMemcpy(got, slot, (size_t)SCENE_BYTES, hipMemcpyDeviceToHost));
            if (memcmp(got, y, (size_t)SCENE_BYTES) != 0) { fprintf(stderr, "scene %d differs\n", s); ++bad; }
        }
        free(got);
        touch(rdone);
        ss_gather_close(g);
        printf("root: %d scenes gathered, %d mismatching\n", NUM, bad);
        return bad ? 1 : 0;
    }
    if (wait_for(hpath)) return 1;
    FILE* f = fopen(hpath, "rb");
    if (!f || fread(ipc, 1, sizeof ipc, f) != sizeof ipc) return 1;
    fclose(f);
    if (ss_gather_attach(&g, ipc, NUM, SCENE_BYTES)) { fprintf(stderr, "ss_gather_attach: %s\n", ss_last_error()); return 1; }
    void* dev = NULL;
    HIPOK(hipMalloc(&dev, (size_t)SCENE_BYTES));
    for (int s = 1; s < NUM; s += 2) {
        if (render_scene(s, y)) return 1;
        if (ss_gather_flush(g)) return 1;                /* (one source buffer: the previous copy must have read it) */
        HIPOK(hipMemcpy(dev, y, (size_t)SCENE_BYTES, hipMemcpyHostToDevice));
        if (ss_gather_put(g, s, dev, NULL /* the null stream */)) { fprintf(stderr, "ss_gather_put: %s\n", ss_last_error()); return 1; }
    }
    if (ss_gather_flush(g)) { fprintf(stderr, "ss_gather_flush: %s\n", ss_last_error()); return 1; }
    touch(pdone);
    if (wait_for(rdone)) return 1;                       /* nobody closes while the root still reads */
    ss_gather_close(g);
    HIPOK(hipFree(dev));
    printf("peer: scenes 1, 3 delivered\n");
    return 0;
}
