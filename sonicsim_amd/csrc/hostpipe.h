// hostpipe.h -- pipelined staging between the caller's HOST buffers and HBM (round 4).
//
// The reference hands NumPy arrays in and takes a NumPy array back (SonicSim_moving.py:122-125: convolve_moving_receiver(
// source1_audio.numpy()[0], np.array(ir1_list).squeeze(1), ...) -> torch.from_numpy(...)): a 307 MB bank of pageable memory per
// moving source at config 2.  Rounds 1-3 pushed it through hipMemcpyAsync on pageable pointers and a synchronous device-to-host
// copy.  This file is the replacement:
//   * a ring of pinned slots, filled by a small pool of host threads (one memcpy stream cannot keep a PCIe Gen5 x16 link busy) while
//     the DMA engine drains the slots already filled -- `up` stream, host-to-device;
//   * the mirror image for results -- `down` stream, device-to-host into pinned slots, copied out to the caller's array while the
//     next piece is in flight;
//   * buffers the caller has already pinned (hipHostMalloc / hipHostRegister) are recognised and moved by DMA directly.
// The render code (render() in sonicsim_hip.hip) hangs its launches on events of these streams, so the render of bank chunk i runs
// while chunk i + 1 is on the wire and finished stretches of the output travel back while later chunks are still coming in.
// Included by sonicsim_hip.hip only (needs its fail() / HIPCHK).
#pragma once
// (<condition_variable>, <deque>, <thread> are included at the top of sonicsim_hip.hip: this file sits inside its anonymous namespace)

struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvd;
    struct Seg { char* d; const char* s; size_t n; };
    std::vector<Seg> segs;      // the current job: independent (dst, src, bytes) pieces, cut into `parts` slices over all bytes
    size_t total = 0;
    int parts = 1, gen = 0, left = 0;
    bool stop = false;

    ~CopyPool() { shutdown(); }
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
        th.clear();
        stop = false;
    }
    std::vector<int> cpus;      // workers are bound to these CPUs; empty = unbound (then the caller's thread takes a slice itself)
    std::vector<std::vector<int>> groups;   // `cpus` by last-level cache (one CCD each on the EPYC hosts): worker i runs on group i -- a CCD's link to
                                            // the memory fabric carries ~55 GB/s, so four copy threads that the scheduler happens to put on ONE CCD stage
                                            // a 307 MB bank in 5.7 ms instead of 3.1 (and the render then takes 7.2 instead of 6.1 ms, profiles/r04al)
    int bound_node = -2;        // NUMA node the workers currently follow (-2: never bound, -1: unbound)
    bool cpus_all_workers = false;   // the caller's thread only waits (it may sit on the far socket) even while no CPU list is set
    void apply_affinity() {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (cpus.empty()) {
            for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &set);
        } else {
            for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
        }
        if (!cpus.empty() && groups.size() >= 2) {
            for (size_t i = 0; i < th.size(); ++i) {
                cpu_set_t one;
                CPU_ZERO(&one);
                for (int c : groups[i % groups.size()]) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &one);
                pthread_setaffinity_np(th[i].native_handle(), sizeof(one), &one);               // best effort
            }
            return;
        }
        for (auto& t : th) pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);      // best effort
    }
    void resize(int n) {      // n = threads in all, the caller's included
        if (n < 1) n = 1;
        // bound workers: the caller's thread may sit on the far socket, so it only waits; unbound: it takes a slice itself
        const bool all_workers = !cpus.empty() || cpus_all_workers;
        const int workers = all_workers ? n : n - 1;
        if ((int)th.size() == workers) return;
        shutdown();
        const int g0 = gen;     // a new worker must not mistake jobs that ran before it existed for a pending one
        const int first = all_workers ? 0 : 1;
        for (int i = 0; i < workers; ++i) th.emplace_back([this, i, g0, first] { run(i + first, g0); });
        if (!cpus.empty()) apply_affinity();
    }
    // slice i of the concatenation of all pieces
    void slice(int i) {
        const size_t per = ((total + (size_t)parts - 1) / (size_t)parts + 63) & ~(size_t)63;
        size_t a = per * (size_t)i, b = a + per < total ? a + per : total;
        size_t base = 0;
        for (const Seg& sg : segs) {
            if (a >= b) break;
            const size_t lo = a > base ? a - base : 0;
            if (lo < sg.n) {
                const size_t hi = b - base < sg.n ? b - base : sg.n;
                memcpy(sg.d + lo, sg.s + lo, hi - lo);
                a = base + hi;
            }
            base += sg.n;
        }
    }
    void run(int i, int seen) {
        for (;;) {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            l.unlock();
            slice(i);
            l.lock();
            if (--left == 0) cvd.notify_one();
        }
    }
    void go() {
        total = 0;
        for (const Seg& sg : segs) total += sg.n;
        const bool caller_copies = cpus.empty() && !cpus_all_workers;      // (matches resize(): all-worker layouts leave the caller waiting)
        parts = (int)th.size() + (caller_copies ? 1 : 0);
        if (total < ((size_t)256 << 10) || th.empty()) {      // small jobs: the wake-up costs more than it saves
            parts = 1;
            slice(0);
            return;
        }
        {
            std::lock_guard<std::mutex> l(mu);
            left = (int)th.size();
            ++gen;
        }
        cv.notify_all();
        if (caller_copies) slice(0);
        std::unique_lock<std::mutex> l(mu);
        cvd.wait(l, [&] { return left == 0; });
    }
    void copy(void* d, const void* s, size_t n) {
        segs.assign(1, Seg{(char*)d, (const char*)s, n});
        go();
    }
};

struct HostPipe {
    static constexpr int NUP = 6, NDOWN = 4;
    size_t slot_bytes = (size_t)32 << 20;    // upload slots (6 of them); with the copy threads on separate CCDs 32 MiB beats 16 by 0.13 ms per config-2
                                             // render (fewer DMA submissions), 48-64 MiB are no better (profiles/r04al); the download slots hold <= 4 MiB pieces
    size_t chunk_bytes = (size_t)24 << 20;   // render() cuts a host bank into chunks of about this size (whole trajectory positions)
    int threads = 0;                    // 0 = choose at first use
    int bind = 2;                       // copy threads: 0 = left to the scheduler, 1 = bound to the CPUs next to the GPU, 2 (default) = they FOLLOW THE CALLER'S
                                        // PAGES -- bound to the NUMA node the array being staged lives on (move_pages query), so the memcpy READS locally and
                                        // posts its writes to the slots across the socket link.  Left alone the threads drift: the same call measured 6.45 or
                                        // 8.5 ms from one run to the next; bound to the GPU's node they read remotely: 8.4 ms (profiles/r04d, r04p)
    bool bind_set = false;              // `bind` was chosen by ss_set_host_pipe or read from SS_HOST_BIND
    bool dirty = false;                 // a host-pointer call is under way (or ended in an error before hp_finish): the next call drains both streams first
    hipStream_t up = nullptr, down = nullptr;
    char* ups[NUP] = {};
    hipEvent_t upev[NUP] = {};
    bool upbusy[NUP] = {};
    int upnext = 0;
    size_t ramp = (size_t)2 << 20;      // size of the next upload piece (reset at the start of every host-pointer call)
    char* dns[NDOWN] = {};
    hipEvent_t dnev[NDOWN] = {};
    int dnnext = 0;
    std::vector<hipEvent_t> evpool;     // chunk-ready / chunk-done events (no timing)
    size_t evused = 0;
    CopyPool pool;
    struct Pending {                    // a device-to-host piece on the wire: slot -> rows of the caller's array
        int slot;
        char* dst;                      // first row's destination
        size_t row_bytes, dst_pitch;
        int rows;
    };
    std::deque<Pending> pending;
    std::vector<std::vector<int>> node_cpus;                 // per NUMA node: its CPUs ...
    std::vector<std::vector<std::vector<int>>> node_groups;  // ... and those by last-level cache (filled on first use, hp_follow)
    // statistics of the last host-pointer render (ss_host_path_stats)
    double st_bytes_up = 0, st_bytes_down = 0, st_seconds = 0;
    int st_chunks = 0, st_direct = 0;
    int st_aborted = 0;                // host-pointer calls that ended in an error after hp_begin and were drained on the way out (render()'s HpAbort)
    double st_mark[8] = {};            // seconds since the call started: staging ready, x on its way, plan built, spectra launched, bank + launches
                                       // enqueued, results enqueued, everything copied out (ss_host_path_stats entries 6..13)
};

static void hp_destroy(HostPipe& h);
static int hp_ensure_build(HostPipe& h);
static void hp_llc_groups(const std::vector<int>& cpus, std::vector<std::vector<int>>& out);
static int hp_ensure(HostPipe& h) {
    if (h.up) return SS_OK;
    const int rc = hp_ensure_build(h);
    if (rc) hp_destroy(h);           // (out of pinned memory half way: nothing half-built survives, the next call starts over)
    return rc;
}
static int hp_ensure_build(HostPipe& h) {
    if (!h.bind_set) {               // SS_HOST_BIND=0|1|2: the placement policy without a code change (a job whose scheduler already pins its workers sets 0)
        if (const char* e = getenv("SS_HOST_BIND")) { const int b = atoi(e); if (b >= 0 && b <= 2) h.bind = b; }
        h.bind_set = true;
    }
    HIPCHK(hipStreamCreateWithFlags(&h.up, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h.down, hipStreamNonBlocking));
    for (int i = 0; i < HostPipe::NUP; ++i) {
        HIPCHK(hipHostMalloc((void**)&h.ups[i], h.slot_bytes, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&h.upev[i], hipEventDisableTiming));
    }
    for (int i = 0; i < HostPipe::NDOWN; ++i) {
        HIPCHK(hipHostMalloc((void**)&h.dns[i], h.slot_bytes < ((size_t)4 << 20) ? h.slot_bytes : ((size_t)4 << 20), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&h.dnev[i], hipEventDisableTiming));
    }
    // optional: the CPUs next to this GPU (sysfs local_cpulist of its PCI function), where the pinned slots live
    h.pool.cpus.clear();
    h.pool.bound_node = -2;
    if (h.bind == 1) {
        int dev = 0;
        char bus[64] = {0};
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bus, sizeof(bus), dev) == hipSuccess) {
            for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
            const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
            if (FILE* f = fopen(path.c_str(), "r")) {
                char line[4096] = {0};
                if (fgets(line, sizeof(line), f)) {
                    h.pool.cpus.clear();
                    for (char* tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
                        int a = 0, b = 0;
                        const int k = sscanf(tok, "%d-%d", &a, &b);
                        if (k == 1) b = a;
                        if (k >= 1) for (int c = a; c <= b && h.pool.cpus.size() < 4096; ++c) h.pool.cpus.push_back(c);
                    }
                }
                fclose(f);
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (h.threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        h.threads = hw >= 8 ? 4 : (hw >= 2 ? 2 : 1);      // measured (profiles/r04a): 4 memcpy threads feed a Gen5 x16 link, more only contend
    }
    h.pool.groups.clear();
    if (!h.pool.cpus.empty()) hp_llc_groups(h.pool.cpus, h.pool.groups);
    h.pool.cpus_all_workers = h.bind == 2;           // the real CPU list follows the first staged array (hp_follow)
    h.pool.resize(h.threads);
    if (h.bind == 2) h.pool.bound_node = -1;
    return SS_OK;
}

// NUMA node of the page that holds `p` (-1: unknown / not faulted in yet)
static int hp_page_node(const void* p) {
#if defined(__linux__) && defined(SYS_move_pages)
    void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1UL, &page, (const int*)nullptr, &status, 0) == 0 && status >= 0) return status;
#endif
    return -1;
}

static bool hp_node_cpus(int node, std::vector<int>& out) {
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char line[4096] = {0};
    out.clear();
    if (fgets(line, sizeof(line), f))
        for (char* tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            const int k = sscanf(tok, "%d-%d", &a, &b);
            if (k == 1) b = a;
            if (k >= 1) for (int c = a; c <= b && out.size() < 4096; ++c) out.push_back(c);
        }
    fclose(f);
    return !out.empty();
}

// the CPUs of `cpus` grouped by the last-level cache they share (sysfs cache/index3/id; one group if the host does not say)
static void hp_llc_groups(const std::vector<int>& cpus, std::vector<std::vector<int>>& out) {
    out.clear();
    std::vector<int> ids;
    for (int c : cpus) {
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/id", c);
        int id = -1;
        if (FILE* f = fopen(path, "r")) {
            if (fscanf(f, "%d", &id) != 1) id = -1;
            fclose(f);
        }
        if (id < 0) { out.clear(); return; }
        size_t g = 0;
        while (g < ids.size() && ids[g] != id) ++g;
        if (g == ids.size()) { ids.push_back(id); out.emplace_back(); }
        out[g].push_back(c);
    }
}

// bind == 2: the workers follow the pages of the array that is about to be staged
static void hp_follow(HostPipe& h, const void* p, size_t bytes) {
    if (h.bind != 2 || h.pool.th.empty()) return;
    if (bytes < ((size_t)8 << 20) && h.pool.bound_node >= 0) return;      // a small array (the dry signal) does not move the threads off the bank's node
    int node = hp_page_node(p);
    if (node < 0 && bytes > 4096) node = hp_page_node((const char*)p + bytes / 2);
    if (node < 0 || node == h.pool.bound_node) return;
    // (CPU list and cache groups of a node are read from sysfs once: a caller whose arrays alternate between the sockets flips per call)
    if ((size_t)node >= h.node_cpus.size()) { h.node_cpus.resize((size_t)node + 1); h.node_groups.resize((size_t)node + 1); }
    if (h.node_cpus[(size_t)node].empty()) {
        if (!hp_node_cpus(node, h.node_cpus[(size_t)node])) return;
        hp_llc_groups(h.node_cpus[(size_t)node], h.node_groups[(size_t)node]);
    }
    h.pool.cpus = h.node_cpus[(size_t)node];
    h.pool.groups = h.node_groups[(size_t)node];
    h.pool.apply_affinity();
    h.pool.bound_node = node;
}

static void hp_destroy(HostPipe& h) {
    h.pool.shutdown();
    for (int i = 0; i < HostPipe::NUP; ++i) {
        if (h.ups[i]) hipHostFree(h.ups[i]);
        if (h.upev[i]) hipEventDestroy(h.upev[i]);
        h.ups[i] = nullptr; h.upev[i] = nullptr; h.upbusy[i] = false;
    }
    for (int i = 0; i < HostPipe::NDOWN; ++i) {
        if (h.dns[i]) hipHostFree(h.dns[i]);
        if (h.dnev[i]) hipEventDestroy(h.dnev[i]);
        h.dns[i] = nullptr; h.dnev[i] = nullptr;
    }
    for (hipEvent_t e : h.evpool) hipEventDestroy(e);
    h.evpool.clear();
    if (h.up) hipStreamDestroy(h.up);
    if (h.down) hipStreamDestroy(h.down);
    h.up = h.down = nullptr;
}

static int hp_event(HostPipe& h, hipEvent_t* out) {
    if (h.evused == h.evpool.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h.evpool.push_back(e);
    }
    *out = h.evpool[h.evused++];
    return SS_OK;
}

// true when the DMA engines can address `p` directly (hipHostMalloc / hipHostRegister memory)
static bool hp_is_pinned(const void* p) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();        // an ordinary malloc'ed pointer: not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// copies finished device-to-host pieces out of their slots; all = false: only those whose DMA has already completed
static int hp_drain(HostPipe& h, bool all) {
    while (!h.pending.empty()) {
        HostPipe::Pending& p = h.pending.front();
        if (!all) {
            const hipError_t q = hipEventQuery(h.dnev[p.slot]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); return SS_OK; }
            if (q != hipSuccess) return fail(SS_EHIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        } else {
            HIPCHK(hipEventSynchronize(h.dnev[p.slot]));
        }
        h.pool.segs.clear();
        for (int r = 0; r < p.rows; ++r)
            h.pool.segs.push_back(CopyPool::Seg{p.dst + (size_t)r * p.dst_pitch, h.dns[p.slot] + (size_t)r * p.row_bytes, p.row_bytes});
        h.pool.go();
        h.pending.pop_front();
    }
    return SS_OK;
}

// host [src, src + bytes) -> device dst on the `up` stream, in pieces of one slot.  Returns when everything is on its way (the last
// pieces may still be in flight: the caller records an event on h.up).  `progress(done_bytes)` is called after every piece has been
// ENQUEUED (render() hangs the launch of a bank chunk on it); finished device-to-host pieces are copied out in between.
template <class F> static int hp_upload(HostPipe& h, void* dst, const void* src, size_t bytes, F progress) {
    h.st_bytes_up += (double)bytes;
    const bool direct = hp_is_pinned(src);
    if (direct) ++h.st_direct;
    else hp_follow(h, src, bytes);
    for (size_t off = 0; off < bytes;) {
        // pieces ramp up from 2 MiB to one slot: the DMA engine starts behind a 2 MiB copy instead of waiting for the first 16 MiB to be staged
        size_t piece = h.ramp < h.slot_bytes ? h.ramp : h.slot_bytes;
        if (h.ramp < h.slot_bytes) h.ramp *= 2;
        const size_t n = bytes - off < piece ? bytes - off : piece;
        if (direct) {
            HIPCHK(hipMemcpyAsync((char*)dst + off, (const char*)src + off, n, hipMemcpyHostToDevice, h.up));
        } else {
            const int q = h.upnext;
            h.upnext = (h.upnext + 1) % HostPipe::NUP;
            if (h.upbusy[q]) {
                HIPCHK(hipEventSynchronize(h.upev[q]));
                h.upbusy[q] = false;
            }
            h.pool.copy(h.ups[q], (const char*)src + off, n);
            HIPCHK(hipMemcpyAsync((char*)dst + off, h.ups[q], n, hipMemcpyHostToDevice, h.up));
            HIPCHK(hipEventRecord(h.upev[q], h.up));
            h.upbusy[q] = true;
        }
        int rc = progress(off + n);
        if (rc) return rc;
        if ((rc = hp_drain(h, false))) return rc;
        off += n;
    }
    return SS_OK;
}
static int hp_upload(HostPipe& h, void* dst, const void* src, size_t bytes) {
    return hp_upload(h, dst, src, bytes, [](size_t) { return (int)SS_OK; });
}

// device rows -> host rows on the `down` stream: `rows` rows of row_bytes, src_pitch apart on the device, dst_pitch apart on the host
// (rows == 1: a flat copy).  The caller has made h.down wait for the producer.  Pageable destinations go through the slot ring and are
// completed by hp_drain.
static int hp_download(HostPipe& h, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, int rows) {
    h.st_bytes_down += (double)row_bytes * rows;
    if (row_bytes == 0 || rows == 0) return SS_OK;
    if (hp_is_pinned(dst)) {
        ++h.st_direct;
        if (rows == 1) HIPCHK(hipMemcpyAsync(dst, src, row_bytes, hipMemcpyDeviceToHost, h.down));
        else HIPCHK(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, row_bytes, (size_t)rows, hipMemcpyDeviceToHost, h.down));
        return SS_OK;
    }
    // pieces of at most 4 MiB (and one slot): whole rows cut along their length.  Small pieces keep the tail short -- the last piece's copy
    // into the caller's (often freshly allocated, not yet faulted-in) array is all that is left once the last DMA has landed
    const size_t piece = h.slot_bytes < ((size_t)4 << 20) ? h.slot_bytes : ((size_t)4 << 20);
    size_t w = piece / (size_t)rows & ~(size_t)63;
    if (w == 0) return fail(SS_EINVAL, "too many rows for the staging slots");
    for (size_t off = 0; off < row_bytes; off += w) {
        const size_t n = row_bytes - off < w ? row_bytes - off : w;
        if ((int)h.pending.size() == HostPipe::NDOWN) {      // the slot we are about to reuse is the oldest pending one
            HostPipe::Pending& p = h.pending.front();
            HIPCHK(hipEventSynchronize(h.dnev[p.slot]));
            h.pool.segs.clear();
            for (int r = 0; r < p.rows; ++r)
                h.pool.segs.push_back(CopyPool::Seg{p.dst + (size_t)r * p.dst_pitch, h.dns[p.slot] + (size_t)r * p.row_bytes, p.row_bytes});
            h.pool.go();
            h.pending.pop_front();
        }
        const int q = h.dnnext;
        h.dnnext = (h.dnnext + 1) % HostPipe::NDOWN;
        if (rows == 1) HIPCHK(hipMemcpyAsync(h.dns[q], (const char*)src + off, n, hipMemcpyDeviceToHost, h.down));
        else HIPCHK(hipMemcpy2DAsync(h.dns[q], n, (const char*)src + off, src_pitch, n, (size_t)rows, hipMemcpyDeviceToHost, h.down));
        HIPCHK(hipEventRecord(h.dnev[q], h.down));
        h.pending.push_back(HostPipe::Pending{q, (char*)dst + off, n, dst_pitch, rows});
    }
    return SS_OK;
}

// everything on both streams done, every pending piece copied out
static int hp_finish(HostPipe& h) {
    int rc = hp_drain(h, true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h.down));
    HIPCHK(hipStreamSynchronize(h.up));
    for (bool& b : h.upbusy) b = false;
    h.evused = 0;
    h.dirty = false;
    return SS_OK;
}

// start of a host-pointer call: per-call state, and -- if the previous call ended in an error half way -- nothing of it still in flight
static int hp_begin(HostPipe& h) {
    if (h.dirty) {
        HIPCHK(hipStreamSynchronize(h.down));
        HIPCHK(hipStreamSynchronize(h.up));
        for (bool& b : h.upbusy) b = false;
    }
    h.dirty = true;
    h.pending.clear();
    h.evused = 0;
    h.ramp = (size_t)2 << 20;
    h.st_bytes_up = h.st_bytes_down = h.st_seconds = 0;
    h.st_chunks = h.st_direct = 0;
    for (double& m : h.st_mark) m = 0;
    return SS_OK;
}
