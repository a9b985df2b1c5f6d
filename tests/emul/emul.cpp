// emul.cpp -- CPU "workgroup emulator" for the HIP kernel bodies (TEST INFRASTRUCTURE ONLY).
// Compiles sonicsim_amd/csrc/tvfir_core.h with g++ and runs each 256-thread workgroup as 256
// std::threads synchronised by a std::barrier, so the exact kernel source (FFT index mapping,
// LDS exchange layout, task planning, epilogue) is verified on machines without a GPU.
// Built and used only by tests/test_emul.py.
#include <barrier>
#include <cstring>
#include <thread>
#include <vector>

#include "../../sonicsim_amd/csrc/plan.h"
#include "../../sonicsim_amd/csrc/tvfir_core.h"
#include "../../sonicsim_amd/csrc/stream13.h"

using namespace ss;

struct HostEnv {
    int tid_;
    std::barrier<>* bar;
    c32* lds_;
    int tid() const { return tid_; }
    void barrier() const { bar->arrive_and_wait(); }
    void wave_sync() const { bar->arrive_and_wait(); }   // host threads are not lock-step: use the full barrier
    c32* lds() const { return lds_; }
    int uniform(int v) const { return v; }
};

template <class F> static void launch(int grid, F&& body, int nthreads = NT) {
    std::vector<c32> lds(LDS13_C32 > LDS12_C32 ? LDS13_C32 : LDS12_C32);
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([&, t]() {
            HostEnv env{t, &bar, lds.data()};
            for (int b = 0; b < grid; ++b) {
                body(env, b);
                bar.arrive_and_wait();
            }
        });
    }
    for (auto& x : th) x.join();
}

extern "C" {

// forward transform of one 2048-point complex vector; out in slot order (c32 index tid*8 + r)
int emul_fft_roundtrip(const float* zin /*[2048][2]*/, float* slots /*[2048][2]*/, float* back /*[2048][2]*/) {
    std::vector<c32> consts;
    build_consts(consts);
    launch(1, [&](HostEnv& env, int) {
        LdsView l; l.base = env.lds();
        load_consts(env, l, consts.data());
        c32 v[8];
        const int tid = env.tid();
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = mk(zin[2 * (n1 * 256 + tid)], zin[2 * (n1 * 256 + tid) + 1]);
        int par = 0;
        fft_fwd(env, l, v, par);
        for (int r = 0; r < 8; ++r) { slots[2 * (tid * 8 + r)] = v[r].x; slots[2 * (tid * 8 + r) + 1] = v[r].y; }
        fft_inv(env, l, v, par);
        for (int n1 = 0; n1 < 8; ++n1) { back[2 * (n1 * 256 + tid)] = v[n1].x; back[2 * (n1 * 256 + tid) + 1] = v[n1].y; }
    });
    return 0;
}

int emul_fft12_roundtrip(const float* zin /*[4096][2]*/, float* slots, float* back) {
    std::vector<c32> consts;
    build_consts12(consts);
    launch(1, [&](HostEnv& env, int) {
        Lds12 l; l.base = env.lds();
        load_consts12(env, l, consts.data());
        c32 v[8];
        const int tid = env.tid();
        for (int n1 = 0; n1 < 8; ++n1) v[n1] = mk(zin[2 * (n1 * 512 + tid)], zin[2 * (n1 * 512 + tid) + 1]);
        int par = 0;
        fft12_fwd(env, l, v, par);
        for (int r = 0; r < 8; ++r) { slots[2 * (tid * 8 + r)] = v[r].x; slots[2 * (tid * 8 + r) + 1] = v[r].y; }
        fft12_inv(env, l, v, par);
        for (int n1 = 0; n1 < 8; ++n1) { back[2 * (n1 * 512 + tid)] = v[n1].x; back[2 * (n1 * 512 + tid) + 1] = v[n1].y; }
    }, NT12);
    return 0;
}

// the segment planner's task list (row, chan, j0, nj per task) as the assembly engine gets it; returns the number of tasks
int emul_plan_dump(const int64_t* seg_len, int P, int C, int L, int groups, int nwg, int32_t* out, int max_tasks) {
    std::vector<int64_t> seg_start(P);
    int64_t s = 0;
    for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
    seg_start[P - 1] = s;
    const int NP = (L + B12 - 1) / B12;
    std::vector<Task> t;
    std::vector<int32_t> scratch;
    plan_seg_lpt(seg_start, P, C, B12, JMAX12, NP, t, scratch, groups, nwg);
    const int n = (int)std::min<size_t>(t.size(), (size_t)max_tasks);
    for (int i = 0; i < n; ++i) { out[4 * i] = t[i].row; out[4 * i + 1] = t[i].chan; out[4 * i + 2] = t[i].j0; out[4 * i + 3] = t[i].nj; }
    return (int)t.size();
}

// the same with every option of the assembly engine's planner: XCD ranges, hop-unit block starts (rs), the shared tail queue
// (tail_pct; *main_out = number of tasks in the per-XCD part)
int emul_plan_dump_ex(const int64_t* seg_len, int P, int C, int L, int groups, int nwg, int rs, int tail_pct, int32_t* main_out, int32_t* out,
                      int max_tasks) {
    std::vector<int64_t> seg_start(P);
    int64_t s = 0;
    for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
    seg_start[P - 1] = s;
    const int NP = (L + B12 - 1) / B12;
    std::vector<Task> t;
    std::vector<int32_t> scratch;
    int32_t m = -1;
    plan_seg_lpt(seg_start, P, C, B12, JMAX12, NP, t, scratch, groups, nwg, rs, tail_pct, &m);
    if (main_out) *main_out = m;
    const int n = (int)std::min<size_t>(t.size(), (size_t)max_tasks);
    for (int i = 0; i < n; ++i) { out[4 * i] = t[i].row; out[4 * i + 1] = t[i].chan; out[4 * i + 2] = t[i].j0; out[4 * i + 3] = t[i].nj; }
    return (int)t.size();
}

// the multi-source (scene) planner: segs = the concatenated segment lengths of the moving sources (P[s] - 1 each; nothing for P[s] == 1)
int emul_plan_scene(const int64_t* segs, const int32_t* Ps, int nsrc, int64_t T, int C, int L, int groups, int tail_pct, int32_t* main_out,
                    int32_t* out, int max_tasks) {
    std::vector<std::vector<int64_t>> starts(nsrc);
    SceneSrc src[8];
    const int64_t* p = segs;
    for (int s = 0; s < nsrc; ++s) {
        starts[s].assign(Ps[s], 0);
        if (Ps[s] > 1) {
            int64_t acc = 0;
            for (int k = 0; k < Ps[s] - 1; ++k) { starts[s][k] = acc; acc += p[k]; }
            starts[s][Ps[s] - 1] = acc;
            p += Ps[s] - 1;
        }
        src[s].seg_start = starts[s].data();
        src[s].P = Ps[s];
    }
    const int NP = (L + B12 - 1) / B12;
    std::vector<Task> t;
    int32_t m = -1;
    plan_scene_lpt(src, nsrc, T, C, B12, JMAX12, NP, t, groups, tail_pct, &m);
    if (main_out) *main_out = m;
    const int n = (int)std::min<size_t>(t.size(), (size_t)max_tasks);
    for (int i = 0; i < n; ++i) { out[4 * i] = t[i].row; out[4 * i + 1] = t[i].chan; out[4 * i + 2] = t[i].j0; out[4 * i + 3] = t[i].nj; }
    return (int)t.size();
}

// property tests (tests/test_plan_properties.py): the segment planner with EVERY knob the library and its tuning build can turn -- jmax, XCD
// ranges, workgroup count, hop-unit block starts, tail queue, row pairing, balanced cut -- followed by the long-row marking (flag_long_rows:
// hrow_min tasks per row, spectra budget in bytes, row-table size).  out = raw tasks (nj with its flag bits), rows_out = the marked rows.
int emul_plan_prop(const int64_t* seg_len, int P, int C, int L, int jmax, int groups, int nwg, int rs, int tail_pct, int pair_rows, int balanced,
                   int hrow_min, int64_t hrow_budget, int hrow_max_rows, int32_t* main_out, int32_t* out, int max_tasks, int32_t* rows_out,
                   int32_t* nrows_out) {
    std::vector<int64_t> seg_start(P);
    int64_t s = 0;
    for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
    seg_start[P - 1] = s;
    const int NP = (L + B12 - 1) / B12;
    std::vector<Task> t;
    std::vector<int32_t> scratch;
    int32_t m = -1;
    const int keep = g_plan_balanced;
    g_plan_balanced = balanced;
    plan_seg_lpt(seg_start, P, C, B12, jmax, NP, t, scratch, groups, nwg, rs, tail_pct, &m, pair_rows != 0);
    g_plan_balanced = keep;
    if (main_out) *main_out = m;
    std::vector<int32_t> rows;
    int nr = 0;
    if (hrow_min > 0) {
        const int32_t Pone = P;
        nr = flag_long_rows(t, &Pone, 1, C, NP, hrow_min, hrow_budget, hrow_max_rows, rows);
    }
    if (nrows_out) *nrows_out = nr;
    for (int i = 0; i < nr && rows_out; ++i) rows_out[i] = rows[(size_t)i];
    const int n = (int)std::min<size_t>(t.size(), (size_t)max_tasks);
    for (int i = 0; i < n; ++i) { out[4 * i] = t[i].row; out[4 * i + 1] = t[i].chan; out[4 * i + 2] = t[i].j0; out[4 * i + 3] = t[i].nj; }
    return (int)t.size();
}

// the host planner of an EXPLICIT (idx, w) schedule as the library runs it for the assembly engine (sonicsim_hip.hip render(): per-tile min / max of
// idx, build_plan, merge_lpt_xcd -- the host twin of k_plan_explicit) + the long-row marking
int emul_plan_explicit(const int64_t* idx, int64_t T, int P, int C, int L, int jmax, int groups, int hrow_min, int64_t hrow_budget, int32_t* out,
                       int max_tasks, int32_t* nrows_out) {
    const int64_t nfine = (T + DTILE - 1) / DTILE;
    std::vector<int32_t> bmin((size_t)nfine), bmax((size_t)nfine);
    for (int64_t b = 0; b < nfine; ++b) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int64_t t = b * DTILE; t < std::min<int64_t>(T, (b + 1) * DTILE); ++t) { lo = std::min(lo, idx[t]); hi = std::max(hi, idx[t]); }
        bmin[(size_t)b] = (int32_t)lo; bmax[(size_t)b] = (int32_t)hi;
    }
    const int NP = (L + B12 - 1) / B12;
    Plan plan;
    build_plan(bmin, bmax, P, C, B12 / DTILE, jmax, plan);
    std::vector<Task> t;
    std::vector<int32_t> scratch;
    merge_lpt_xcd(plan, NP, groups, t, scratch);
    std::vector<int32_t> rows;
    int nr = 0;
    if (hrow_min > 0) {
        const int32_t Pone = P;
        nr = flag_long_rows(t, &Pone, 1, C, NP, hrow_min, hrow_budget, HROW_MAX, rows);
    }
    if (nrows_out) *nrows_out = nr;
    const int n = (int)std::min<size_t>(t.size(), (size_t)max_tasks);
    for (int i = 0; i < n; ++i) { out[4 * i] = t[i].row; out[4 * i + 1] = t[i].chan; out[4 * i + 2] = t[i].j0; out[4 * i + 3] = t[i].nj; }
    return (int)t.size();
}

// planner cross-check: the direct O(P*C) segment planner must emit exactly the tasks of the generic (min/max driven) planner
// whose row actually owns samples; returns 0 when consistent, otherwise a positive diagnostic code
int emul_plan_compare(const int64_t* seg_len, int P, int C, int L, int64_t* nfast, int64_t* ngeneric) {
    std::vector<int64_t> seg_start(P);
    int64_t s = 0;
    for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
    seg_start[P - 1] = s;
    const int64_t T = s;
    const int NP = (L + B12 - 1) / B12;
    std::vector<Task> fast;
    std::vector<int32_t> scratch;
    plan_seg_lpt(seg_start, P, C, B12, JMAX12, NP, fast, scratch);
    std::vector<int32_t> bmin, bmax;
    seg_minmax(seg_start, T, bmin, bmax);
    Plan plan;
    build_plan(bmin, bmax, P, C, B12 / DTILE, JMAX12, plan);
    std::vector<Task> gen;
    merge_lpt(plan, NP, gen);
    *nfast = (int64_t)fast.size();
    *ngeneric = (int64_t)gen.size();
    auto key = [](const Task& t) { return ((int64_t)t.row << 40) | ((int64_t)t.chan << 32) | (uint32_t)t.j0; };
    // every sample of every (row, channel) must be covered exactly once by the fast plan
    std::vector<int64_t> covered((size_t)P * C, 0);
    for (const Task& t : fast) {
        if (t.nj < 1 || t.nj > JMAX12) return 1;
        const int64_t a0 = seg_start[t.row > 0 ? t.row - 1 : t.row], a2 = seg_start[t.row < P - 1 ? t.row + 1 : t.row];
        const int64_t lo = std::max<int64_t>(a0, (int64_t)t.j0 * B12), hi = std::min<int64_t>(a2, (int64_t)(t.j0 + t.nj) * B12);
        if (hi <= lo) return 2;                                       // a task that owns no sample
        covered[(size_t)t.row * C + t.chan] += hi - lo;
    }
    for (int r = 0; r < P; ++r) {
        const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
        for (int c = 0; c < C; ++c)
            if (covered[(size_t)r * C + c] != std::max<int64_t>(0, a2 - a0)) return 3;
    }
    // descending cost order
    auto cost = [NP](const Task& t) { const int np_eff = std::min(NP, t.j0 + t.nj); return task_cost(np_eff, t.nj); };
    for (size_t i = 1; i < fast.size(); ++i)
        if (cost(fast[i]) > cost(fast[i - 1])) return 4;
    // the generic plan covers a superset of (row, chan, block) pairs
    std::vector<int64_t> gk;
    for (const Task& t : gen) for (int j = 0; j < t.nj; ++j) { Task u = t; u.j0 = t.j0 + j; gk.push_back(key(u)); }
    std::sort(gk.begin(), gk.end());
    for (const Task& t : fast)
        for (int j = 0; j < t.nj; ++j) { Task u = t; u.j0 = t.j0 + j; if (!std::binary_search(gk.begin(), gk.end(), key(u))) return 5; }
    return 0;
}

// streaming render with persistent state (stream13.h): the kernel bodies and the host-side cut of a push into pieces, pushes of the
// given sizes (the last push takes whatever is left); y[C][T].  Returns the number of pieces (kernel launches), < 0 on error.
int emul_stream(const float* x, int64_t T, const float* bank, int P, int C, int L, const int64_t* seg_len, const int64_t* sizes, int nsizes, float* y) {
    std::vector<int64_t> seg_start(P);
    int64_t s = 0;
    for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
    seg_start[P - 1] = s;
    if (s != T) return -1;
    std::vector<c32> c13;
    build_consts13(c13);
    StreamDev d;
    d.bank = bank; d.P = P; d.C = C; d.L = L; d.NP = (L + B13 - 1) / B13; d.NR = d.NP + 1; d.consts = c13.data();
    std::vector<c32> Hs((size_t)STREAM_ROW_SLOTS * C * d.NP * B13), Xr((size_t)d.NR * B13);
    std::vector<float> xh((size_t)T);
    d.Hs = Hs.data(); d.Xr = Xr.data(); d.xh = xh.data();
    int slot_row[STREAM_ROW_SLOTS] = {-1, -1, -1, -1};
    auto prepare = [&](int row) {
        if (row < 0 || row >= P || slot_row[row & (STREAM_ROW_SLOTS - 1)] == row) return;
        launch(d.NP * C, [&](HostEnv& env, int b) { stream_row_body(env, d, row, b % d.NP, b / d.NP); }, NT13);
        slot_row[row & (STREAM_ROW_SLOTS - 1)] = row;
    };
    for (int r = 0; r < 3; ++r) prepare(r);
    int64_t pos = 0;
    int k = 0, pieces = 0;
    for (int i = 0; pos < T; ++i) {
        int64_t n = i < nsizes ? sizes[i] : T - pos;
        if (n > T - pos) n = T - pos;
        if (n <= 0) { if (i >= nsizes) break; continue; }
        std::vector<float> out((size_t)C * n);
        int64_t left = n, off = 0;
        while (left > 0) {
            int64_t len = 0;
            if (!stream_next_piece(seg_start.data(), P, pos, left, k, len)) return -2;
            for (int r = k; r <= k + 2; ++r) prepare(r);
            StreamPiece pc;
            pc.pos = pos; pc.n = (int32_t)len; pc.j = (int32_t)(pos / B13); pc.k = k;
            pc.seg_start = seg_start[k];
            const int64_t nk = seg_start[k + 1] - seg_start[k];
            pc.inv_len = nk > 0 ? 1.0 / (double)nk : 0.0;
            pc.chunk = x + pos; pc.out = out.data(); pc.out_stride = n; pc.out_off = off;
            launch(C, [&](HostEnv& env, int c) { stream_push_body(env, d, pc, c); }, NT13);
            ++pieces;
            pos += len; off += len; left -= len;
        }
        for (int c = 0; c < C; ++c) std::memcpy(y + (int64_t)c * T + (pos - n), out.data() + (int64_t)c * n, sizeof(float) * (size_t)n);
    }
    return pieces;
}

// mode: 0 fixed (P==1), 1 seg (seg_len[P-1]), 2 explicit (idx,w).  path: 0 = overlap-save, 1 = direct
int emul_render(const float* x, int64_t T, const float* bank, int P, int C, int L, int mode,
                const int64_t* seg_len, const int64_t* idx, const float* w, float* y, int path, int64_t* ntasks, int xd) {
    const bool g13 = (path == 0 && xd == 13);
    const bool g12 = (path == 0 && xd == 12) || g13;
    const int BB = g12 ? B12 : B;
    std::vector<c32> consts;
    if (g12) build_consts12(consts); else build_consts(consts);
    const int M = (int)((T + BB - 1) / BB);
    std::vector<c32> Xs((size_t)(M + 1) * BB);
    if (path == 0 && !g12) launch(M + 1, [&](HostEnv& env, int m) { xspec_body(env, x, T, consts.data(), Xs.data(), m, M); });
    if (g13) {
        std::vector<c32> c13;
        build_consts13(c13);
        launch(M + 1, [&](HostEnv& env, int m) { xspec13_body(env, x, T, c13.data(), Xs.data(), m, M, y, (int64_t)C * T); }, NT13);
    } else if (g12) launch(M + 1, [&](HostEnv& env, int m) { xspec12_body(env, x, T, consts.data(), Xs.data(), m, M, y, (int64_t)C * T); }, NT12);

    Plan plan;
    std::vector<int64_t> seg_start;
    std::vector<int32_t> bmin, bmax;
    const int fine = path == 0 ? BB / DTILE : 1;
    const int jmax = path == 0 ? (g12 ? JMAX12 : JMAX) : 1;
    if (mode == 0) {
        build_plan_fixed(T, C, path == 0 ? BB : DTILE, jmax, plan);
    } else if (mode == 1) {
        seg_start.resize(P);
        int64_t s = 0;
        for (int k = 0; k < P - 1; ++k) { seg_start[k] = s; s += seg_len[k]; }
        seg_start[P - 1] = s;
        if (s != T) return -1;
        seg_minmax(seg_start, T, bmin, bmax);
        build_plan(bmin, bmax, P, C, fine, jmax, plan);
    } else {
        const int64_t nb = (T + DTILE - 1) / DTILE;
        bmin.assign(nb, INT32_MAX);
        bmax.assign(nb, INT32_MIN);
        for (int64_t t = 0; t < T; ++t) {
            bmin[t / DTILE] = std::min<int32_t>(bmin[t / DTILE], (int32_t)idx[t]);
            bmax[t / DTILE] = std::max<int32_t>(bmax[t / DTILE], (int32_t)idx[t]);
        }
        build_plan(bmin, bmax, P, C, fine, jmax, plan);
    }
    RenderParams prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.x = x; prm.T = T; prm.bank = bank; prm.P = P; prm.C = C; prm.L = L;
    prm.NP = (L + BB - 1) / BB; prm.Xs = Xs.data(); prm.M = M; prm.consts = consts.data();
    prm.mode = mode; prm.seg_start = seg_start.data(); prm.idx = idx; prm.w = w; prm.y = y;
    *ntasks = 0;
    if (g12) {   // single persistent launch, atomic accumulation onto the zeroed y, LPT task order
        std::vector<Task> all;
        merge_lpt(plan, prm.NP, all);
        prm.tasks = all.data();
        prm.ntasks = (int)all.size();
        prm.accumulate = 2;
        *ntasks = (int64_t)all.size();
        if (g13) {
            std::vector<c32> c13;
            build_consts13(c13);
            int counter = 0;
            Params13 p13;
            p13.r = prm;
            p13.r.consts = c13.data();
            p13.counter = &counter;
            p13.nwg = std::min<int>(prm.ntasks, 3);
            launch(p13.nwg, [&](HostEnv& env, int b) { os13_body(env, p13, b); }, NT13);
            return 0;
        }
        const int nwg = std::min<int>(prm.ntasks, 5);
        launch(nwg, [&](HostEnv& env, int b) { os12_body(env, prm, b, nwg); }, NT12);
        return 0;
    }
    for (int parity = 0; parity < 2; ++parity) {
        if (plan.tasks[parity].empty()) continue;
        prm.tasks = plan.tasks[parity].data();
        prm.accumulate = parity;
        *ntasks += (int64_t)plan.tasks[parity].size();
        xcd_interleave(plan.tasks[parity]);
        prm.tasks = plan.tasks[parity].data();
        if (path == 0 && xd == 0) launch((int)plan.tasks[parity].size(), [&](HostEnv& env, int b) { os_body<HostEnv, 0>(env, prm, b); });
        else if (path == 0 && xd == 2) launch((int)plan.tasks[parity].size(), [&](HostEnv& env, int b) { os_body<HostEnv, 2>(env, prm, b); });
        else if (path == 0) launch((int)plan.tasks[parity].size(), [&](HostEnv& env, int b) { os_body<HostEnv, 3>(env, prm, b); });
        else launch((int)plan.tasks[parity].size(), [&](HostEnv& env, int b) { direct_body(env, prm, b); });
    }
    return 0;
}
}
