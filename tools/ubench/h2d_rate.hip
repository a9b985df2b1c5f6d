// h2d_rate.hip -- what the host-pointer path of the render can hope for on this box (round 4).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/ubench/h2d_rate.hip -o tools/ubench/h2d_rate -lpthread
// Measures, for a config-2 bank (307.2 MB) and a config-2 output (30.7 MB):
//   (a) hipMemcpy from / to PAGEABLE memory (what rounds 1-3 did)
//   (b) hipHostRegister + DMA + hipHostUnregister of the caller's buffer
//   (c) DMA from / to PINNED memory (the PCIe ceiling)
//   (d) N host threads copying pageable -> pinned ring slots, DMA behind them (the pipelined staging of hostpipe.h)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvd;
    char* dst = nullptr;
    const char* src = nullptr;
    size_t n = 0;
    int parts = 0, gen = 0, left = 0;
    bool stop = false;
    explicit Pool(int k) {
        for (int i = 0; i < k; ++i) th.emplace_back([this, i] { run(i); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void piece(int i) {
        const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
        const size_t a = per * i, b = a + per < n ? a + per : n;
        if (a < b) memcpy(dst + a, src + a, b - a);
    }
    void run(int i) {
        int seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            l.unlock();
            piece(i + 1);
            l.lock();
            if (--left == 0) cvd.notify_one();
        }
    }
    void copy(void* d, const void* s, size_t bytes) {
        dst = (char*)d; src = (const char*)s; n = bytes; parts = (int)th.size() + 1;
        if (!th.empty()) {
            { std::lock_guard<std::mutex> l(mu); left = (int)th.size(); ++gen; }
            cv.notify_all();
        }
        piece(0);
        if (!th.empty()) {
            std::unique_lock<std::mutex> l(mu);
            cvd.wait(l, [&] { return left == 0; });
        }
    }
};

int main(int argc, char** argv) {
    const size_t NB = 307200000, NY = 30720000;
    const int reps = 3;
    char* dev;
    CK(hipMalloc((void**)&dev, NB));
    char* page = (char*)malloc(NB);
    memset(page, 1, NB);
    char* pin;
    double t0 = now();
    CK(hipHostMalloc((void**)&pin, NB, hipHostMallocDefault));
    printf("hipHostMalloc(307 MB) %.2f ms\n", (now() - t0) * 1e3);
    memset(pin, 2, NB);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int r = 0; r < reps; ++r) {
        t0 = now();
        CK(hipMemcpy(dev, page, NB, hipMemcpyHostToDevice));
        double a = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(dev, pin, NB, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double c = now() - t0;
        t0 = now();
        CK(hipHostRegister(page, NB, hipHostRegisterDefault));
        double b1 = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(dev, page, NB, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double b2 = now() - t0;
        t0 = now();
        CK(hipHostUnregister(page));
        double b3 = now() - t0;
        printf("H2D 307 MB: pageable hipMemcpy %.2f ms (%.1f GB/s) | pinned DMA %.2f ms (%.1f GB/s) | register %.2f + DMA %.2f + unregister %.2f ms\n",
               a * 1e3, NB / a / 1e9, c * 1e3, NB / c / 1e9, b1 * 1e3, b2 * 1e3, b3 * 1e3);
    }
    for (int r = 0; r < reps; ++r) {
        t0 = now();
        CK(hipMemcpy(page, dev, NY, hipMemcpyDeviceToHost));
        double a = now() - t0;
        t0 = now();
        CK(hipMemcpyAsync(pin, dev, NY, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double c = now() - t0;
        printf("D2H 30.7 MB: pageable hipMemcpy %.3f ms (%.1f GB/s) | pinned DMA %.3f ms (%.1f GB/s)\n", a * 1e3, NY / a / 1e9, c * 1e3, NY / c / 1e9);
    }
    // both directions at once (PCIe is full duplex)
    t0 = now();
    CK(hipMemcpyAsync(dev, pin, NB / 2, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(pin + NB / 2, dev + NB / 2, NB / 2, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s));
    CK(hipStreamSynchronize(s2));
    printf("duplex 153.6 MB each way: %.2f ms\n", (now() - t0) * 1e3);
    // small-copy latency
    for (size_t sz : {(size_t)4096, (size_t)(1 << 20), (size_t)3840000}) {
        t0 = now();
        for (int i = 0; i < 20; ++i) {
            CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
        }
        printf("pinned H2D of %zu bytes + sync: %.1f us\n", sz, (now() - t0) / 20 * 1e6);
    }
    // host memcpy rates
    for (int k : {1, 2, 4, 8, 12, 16, 24, 32}) {
        Pool p(k - 1);
        p.copy(pin, page, NB);
        t0 = now();
        p.copy(pin, page, NB);
        double a = now() - t0;
        t0 = now();
        for (int i = 0; i < 8; ++i) p.copy(pin + (size_t)i * (8 << 20), page + (size_t)i * (8 << 20), 8 << 20);
        double b = now() - t0;
        printf("host memcpy pageable -> pinned, %2d threads: 307 MB in %.2f ms (%.1f GB/s); 8 MB pieces %.1f GB/s\n", k, a * 1e3, NB / a / 1e9,
               8.0 * (8 << 20) / b / 1e9);
    }
    // pipelined staging: ring of slots
    for (size_t slot : {(size_t)(4 << 20), (size_t)(8 << 20), (size_t)(16 << 20)})
        for (int k : {4, 8, 16}) {
            const int NS = 6;
            Pool p(k - 1);
            hipEvent_t ev[NS];
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            bool used[NS] = {};
            double best = 1e9;
            for (int r = 0; r < reps; ++r) {
                t0 = now();
                int i = 0;
                for (size_t off = 0; off < NB; off += slot, ++i) {
                    const int q = i % NS;
                    const size_t n = off + slot <= NB ? slot : NB - off;
                    if (used[q]) CK(hipEventSynchronize(ev[q]));
                    p.copy(pin + (size_t)q * slot, page + off, n);
                    CK(hipMemcpyAsync(dev + off, pin + (size_t)q * slot, n, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[q], s));
                    used[q] = true;
                }
                CK(hipStreamSynchronize(s));
                double a = now() - t0;
                if (a < best) best = a;
            }
            printf("staged H2D 307 MB: slot %2zu MB x %d, %2d threads: %.2f ms (%.1f GB/s)\n", slot >> 20, NS, k, best * 1e3, NB / best / 1e9);
            for (auto& e : ev) CK(hipEventDestroy(e));
        }
    printf("host cores: %u\n", std::thread::hardware_concurrency());
    return 0;
}
