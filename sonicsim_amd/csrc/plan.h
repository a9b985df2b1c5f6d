// plan.h -- host-side task planning for the row-stationary render (plain C++, shared by the HIP
// library and the CPU workgroup emulator used in tests).
//
// Work decomposition: filter row r contributes to output sample t iff idx[t] == r (start filter)
// or idx[t]+1 == r (end filter) -- SonicSim_moving.py:89-90.  From the per-block min/max of idx
// (a fine grid of DTILE samples) we derive, for every row, the set of output blocks it touches,
// split it into runs of consecutive blocks of at most JMAX, and emit one Task per (run, channel).
// Rows are separated by parity: every sample has exactly one even and one odd responsible row, so
// the even pass STORES and the odd pass ADDS -- deterministic, no atomics, no zero fill.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "tvfir_core.h"
#include "tvfir13.h"

namespace ss {

struct Plan {
    std::vector<Task> tasks[2];   // [parity]
    int64_t pairs = 0;            // number of (row, block) pairs
};

// per-fine-block (DTILE grid) min/max of idx for the implicit (segment) schedule
inline void seg_minmax(const std::vector<int64_t>& seg_start /*[P], last == T*/, int64_t T,
                       std::vector<int32_t>& bmin, std::vector<int32_t>& bmax) {
    const int64_t nb = (T + DTILE - 1) / DTILE;
    bmin.resize(nb);
    bmax.resize(nb);
    const int P = (int)seg_start.size();
    // segment of sample t: last k in [0, P-2] with seg_start[k] <= t
    auto seg_of = [&](int64_t t) {
        int lo = 0, hi = P - 2;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_start[mid] <= t) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * DTILE, t1 = std::min<int64_t>(T, t0 + DTILE) - 1;
        bmin[b] = seg_of(t0);
        bmax[b] = seg_of(t1);
    }
}

// Build tasks.  `fine_per_block` = B/DTILE for the overlap-save engine, 1 for the direct engine.
inline void build_plan(const std::vector<int32_t>& bmin, const std::vector<int32_t>& bmax, int P, int C,
                       int fine_per_block, int jmax, Plan& plan) {
    plan.tasks[0].clear();
    plan.tasks[1].clear();
    plan.pairs = 0;
    const int64_t nfine = (int64_t)bmin.size();
    const int64_t nblk = (nfine + fine_per_block - 1) / fine_per_block;
    std::vector<std::vector<int32_t>> rows(P);
    for (int64_t j = 0; j < nblk; ++j) {
        int32_t lo = INT32_MAX, hi = INT32_MIN;
        for (int f = 0; f < fine_per_block; ++f) {
            const int64_t fb = j * fine_per_block + f;
            if (fb < nfine) { lo = std::min(lo, bmin[fb]); hi = std::max(hi, bmax[fb]); }
        }
        for (int32_t r = lo; r <= hi + 1 && r < P; ++r) rows[r].push_back((int32_t)j);
    }
    for (int r = 0; r < P; ++r) {
        const auto& bl = rows[r];
        plan.pairs += (int64_t)bl.size();
        size_t i = 0;
        while (i < bl.size()) {
            size_t e = i + 1;
            while (e < bl.size() && bl[e] == bl[e - 1] + 1 && (int)(e - i) < jmax) ++e;
            for (int c = 0; c < C; ++c) {
                Task t;
                t.row = r;
                t.chan = c;
                t.j0 = bl[i];
                t.nj = (int32_t)(e - i);
                plan.tasks[r & 1].push_back(t);
            }
            i = e;
        }
    }
}

// XCD-aware launch order.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed
// only): give each XCD a CONTIGUOUS chunk of the (row-major) task list so that the tasks co-resident
// on one XCD belong to neighbouring rows and share their input-spectra window in that XCD's 4 MiB L2.
inline void xcd_interleave(std::vector<Task>& tasks, int nxcd = 8) {
    const size_t n = tasks.size();
    if (n < (size_t)nxcd * 2) return;
    const size_t q = n / nxcd, r = n % nxcd;
    std::vector<Task> out(n);
    for (size_t b = 0; b < n; ++b) {
        const size_t x = b % nxcd, i = b / nxcd;
        const size_t start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        out[b] = tasks[start + i];
    }
    tasks.swap(out);
}

// Merge both parity lists into one launch order sorted by descending cost (LPT): with atomic accumulation
// the rows no longer need two passes.  cost ~ partitions * (fft + nj * mac) + nj * ifft.
inline void merge_lpt(Plan& plan, int NP, std::vector<Task>& out) {
    out.clear();
    out.reserve(plan.tasks[0].size() + plan.tasks[1].size());
    out.insert(out.end(), plan.tasks[0].begin(), plan.tasks[0].end());
    out.insert(out.end(), plan.tasks[1].begin(), plan.tasks[1].end());
    auto cost = [NP](const Task& t) {
        const int np_eff = std::min(NP, t.j0 + t.nj);
        return (int64_t)task_cost(np_eff, t.nj);
    };
    std::stable_sort(out.begin(), out.end(), [&](const Task& a, const Task& b) { return cost(a) > cost(b); });
}

// XCD-aware LPT order of the generic planner's lists (explicit schedule, fixed receiver) for the persistent kernels: see
// plan_seg_lpt's `groups`.  Both parity lists come out of build_plan in time order (row, first block), so one linear merge gives
// the time order; `groups` contiguous ranges of equal total cost; a counting sort by cost inside every range; ranges
// interleaved task by task (position i <- range i % groups).  O(n), no allocation once the buffers have grown.
inline void merge_lpt_xcd(Plan& plan, int NP, int groups, std::vector<Task>& out, std::vector<int32_t>& scratch) {
    constexpr int MAXCOST = 4096;
    auto cost = [NP](const Task& t) {
        const int np_eff = std::min(NP, t.j0 + t.nj);
        const int c = task_cost(np_eff, t.nj);
        return c < MAXCOST ? c : MAXCOST - 1;
    };
    static thread_local std::vector<Task> byt, tmp;
    const std::vector<Task>&A = plan.tasks[0], &B = plan.tasks[1];
    const size_t n = A.size() + B.size();
    byt.resize(n);
    auto before = [](const Task& a, const Task& b) { return a.row != b.row ? a.row < b.row : (a.j0 != b.j0 ? a.j0 < b.j0 : a.chan < b.chan); };
    std::merge(A.begin(), A.end(), B.begin(), B.end(), byt.begin(), before);
    if (!std::is_sorted(byt.begin(), byt.end(), before)) std::stable_sort(byt.begin(), byt.end(), before);   // (not expected)
    // moving source: the tasks of one filter row stay together -- channel-major inside the row and keyed by the row's most expensive task
    // (plan_seg_lpt, round 4: the second reader of a row's taps then finds them in its XCD's L2).  A fixed receiver has ONE row per channel
    // read by every task: its list stays in time order.
    static thread_local std::vector<int32_t> rowmax;
    const bool pair_rows = n > 0 && byt.front().row != byt.back().row;
    if (pair_rows) {
        std::stable_sort(byt.begin(), byt.end(), [](const Task& a, const Task& b) { return a.row != b.row ? a.row < b.row : (a.chan != b.chan ? a.chan < b.chan : a.j0 < b.j0); });
        rowmax.assign((size_t)byt.back().row + 1, 0);
        for (const Task& t : byt) rowmax[(size_t)t.row] = std::max<int32_t>(rowmax[(size_t)t.row], cost(t));
    }
    auto key = [&](const Task& t) { return pair_rows ? (int)rowmax[(size_t)t.row] : cost(t); };
    int64_t total = 0;
    for (const Task& t : byt) total += cost(t);
    if (groups > 64) groups = 64;
    if (groups < 1 || n < (size_t)groups * 2) groups = 1;
    scratch.assign((size_t)groups * MAXCOST + groups + n, 0);
    int32_t* hist = scratch.data();
    int32_t* gcount = scratch.data() + (size_t)groups * MAXCOST;
    int32_t* gof = gcount + groups;                                   // range of every task
    int64_t acc = 0;
    for (size_t i = 0; i < n; ++i) {
        const int c = cost(byt[i]);
        int g = total > 0 ? (int)((__int128)(acc + c / 2) * groups / total) : 0;
        if (g >= groups) g = groups - 1;
        gof[i] = g;
        hist[(size_t)g * MAXCOST + key(byt[i])]++;
        gcount[g]++;
        acc += c;
    }
    int32_t run = 0;
    for (int g = 0; g < groups; ++g)
        for (int c = MAXCOST - 1; c >= 0; --c) { int32_t& h = hist[(size_t)g * MAXCOST + c]; const int32_t k = h; h = run; run += k; }
    tmp.resize(n);
    for (size_t i = 0; i < n; ++i) tmp[(size_t)hist[(size_t)gof[i] * MAXCOST + key(byt[i])]++] = byt[i];
    out.resize(n);
    int32_t next[64], end[64];
    int32_t off = 0;
    for (int g = 0; g < groups; ++g) { next[g] = off; off += gcount[g]; end[g] = off; }
    for (size_t i = 0; i < n; ++i) {
        int g = (int)(i % (size_t)groups);
        for (int k = 0; k < groups && next[g] >= end[g]; ++k) g = (g + 1) % groups;
        out[i] = tmp[(size_t)next[g]++];
    }
}

// Direct O(P*C) planner of the implicit (segment) schedule for the single-launch geometries (12/13/14): row r is responsible
// for the samples [seg_start[r-1], seg_start[r+1]) (end filter of segment r-1, start filter of segment r -- SonicSim_moving.py
// :89-94), i.e. for a run of consecutive output blocks, cut into tasks of at most `jmax` blocks.  Tasks come out in descending
// cost order (the persistent workgroups take them round-robin = LPT), row-major inside a cost class (neighbouring rows share
// their input-spectra window in L2).  No per-call allocation once `out` / `scratch` have grown.
// groups > 1 (XCD-aware order for persistent kernels whose workgroup b runs on XCD b % groups and takes tasks b, b + nwg, ...):
// the rows are cut into `groups` contiguous time ranges of equal total cost, every range is sorted by descending cost on its
// own, and the ranges are interleaved task by task, so that position i of the list belongs to range i % groups.  Each XCD then
// works on one stretch of the trajectory and its L2 only has to hold that stretch's input spectra (1/groups of the set).
// nwg = number of persistent workgroups (0 = unknown: plain round-robin dealing).
// rs > 0 (assembly engine): the input spectra exist on a grid of block >> rs samples, so a row's first block may start at any
// multiple of that hop instead of a multiple of `block` -- a row of 2.3 blocks then never spills into a fourth block and fewer
// rows need a second task.  Task::j0 is in HOP units (first sample = j0 * (block >> rs)); blocks of a task stay `block` apart.
inline int g_plan_balanced = 1;
template <class F> inline void row_tasks(int64_t a0, int64_t a2, int block, int jmax, int rs, F f) {
    if (a2 <= a0) return;
    const int64_t hop = block >> rs;
    int64_t jh = a0 / hop;
    int64_t nb = (a2 - jh * hop + block - 1) / block;
    // a row of jmax + 2 or more blocks is cut into the FEWEST tasks, of (nearly) EQUAL size (round 4; rounds 1-3 cut greedily: 4 + 2 instead of
    // 3 + 3): tasks of one row then run at the same pace, so the two that read the same taps stay within the L2's reach of each other
    // (plan_seg_lpt puts them next to each other in the queue), and the largest task is smaller.  Same (row, block) pairs, same bits.
    if (!g_plan_balanced || nb < jmax + 2) {      // jmax + 1 blocks stay jmax + 1 (config 2's rare long rows: measured better, profiles/r04u);
                                                   // SS_PLAN_SPLIT=0 (tuning build): the greedy cut of rounds 1-3 throughout
        while (nb > 0) {
            const int nj = (int)std::min<int64_t>(jmax, nb);
            f((int)jh, nj);
            jh += (int64_t)nj << rs;
            nb -= nj;
        }
        return;
    }
    const int64_t ntask = (nb + jmax - 1) / jmax;
    for (int64_t k = 0; k < ntask; ++k) {
        const int nj = (int)(nb / ntask + (k < nb % ntask ? 1 : 0));
        f((int)jh, nj);
        jh += (int64_t)nj << rs;
    }
}

inline void plan_seg_lpt(const std::vector<int64_t>& seg_start /*[P], last == T*/, int P, int C, int block, int jmax, int NP,
                         std::vector<Task>& out, std::vector<int32_t>& scratch, int groups = 1, int nwg = 0, int rs = 0, int tail_pct = 0,
                         int32_t* main_out = nullptr, bool pair_rows = true) {
    out.clear();
    constexpr int MAXCOST = 4096;
    auto cost = [NP, rs](int j0, int nj) {
        const int np_eff = std::min(NP, ((j0 + (1 << rs) - 1) >> rs) + nj);
        const int c = task_cost(np_eff, nj);
        return c < MAXCOST ? c : MAXCOST - 1;
    };
    if (groups > 1) {
        static thread_local std::vector<Task> tmp;
        static thread_local std::vector<int64_t> rowcost;
        rowcost.assign((size_t)P + 1, 0);
        // per-row cost and the range of every row
        for (int r = 0; r < P; ++r) {
            const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
            int64_t rc = 0;
            row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) { rc += cost(j, nj); });
            rowcost[(size_t)r + 1] = rowcost[(size_t)r] + rc;
        }
        const int64_t total_cost = rowcost[(size_t)P];
        auto group_of = [&](int r) {
            if (total_cost <= 0) return 0;
            const int64_t mid = rowcost[(size_t)r] + (rowcost[(size_t)r + 1] - rowcost[(size_t)r]) / 2;
            const int g = (int)((__int128)mid * groups / total_cost);
            return g < groups ? g : groups - 1;
        };
        scratch.assign((size_t)groups * MAXCOST + groups + 1, 0);
        int32_t* hist = scratch.data();                       // [groups][MAXCOST]
        int32_t* gcount = scratch.data() + (size_t)groups * MAXCOST;
        // Sort key of a task = the cost of its row's MOST EXPENSIVE task (round 4): a row that spans more than jmax blocks is cut into
        // several tasks (config 5: every row, 4 + 2..3 blocks) and each of them streams the whole row from HBM.  Keyed by their own costs the
        // tasks of a row sit far apart in the queue and the row is read twice (PMC: 1.73 GB fetched for a 0.77 GB bank, profiles/r04j/cfg5);
        // keyed by the row they are NEIGHBOURS in their XCD's queue, are taken by workgroups of that XCD at the same moment, and the second
        // reader of every 16 KB partition finds it in the XCD's L2.  (Rows of one task -- config 2, mostly -- keep their order.)
        auto rowkey = [&](int64_t a0, int64_t a2) {
            int k = 0;
            row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) { const int c = cost(j, nj); if (c > k) k = c; });
            return k;
        };
        for (int r = 0; r < P; ++r) {
            const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
            if (a2 <= a0) continue;
            const int g = group_of(r);
            const int key = pair_rows ? rowkey(a0, a2) : -1;
            row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) {
                hist[(size_t)g * MAXCOST + (pair_rows ? key : cost(j, nj))] += C;
                gcount[g] += C;
            });
        }
        int32_t total = 0;
        for (int g = 0; g < groups; ++g)
            for (int c = MAXCOST - 1; c >= 0; --c) { int32_t& h = hist[(size_t)g * MAXCOST + c]; const int32_t n = h; h = total; total += n; }
        tmp.resize((size_t)total);
        for (int r = 0; r < P; ++r) {
            const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
            if (a2 <= a0) continue;
            const int g = group_of(r);
            const int key = pair_rows ? rowkey(a0, a2) : -1;
            if (pair_rows) {
                // channel-major inside a row: (task A, c), (task B, c), (task A, c + 1), ... -- the two readers of filter (r, c) take
                // CONSECUTIVE tickets, i.e. start ~one task-start interval apart, well inside the time a tap line survives in the XCD's L2
                // (any number of tasks per row: a row of P = 2 or 3 over a long T spans hundreds of blocks.  Round 4 kept them in a 16-entry array
                // and silently dropped the rest while the histogram had counted them -- tests/test_emul.py::test_planner_covers_long_rows)
                static thread_local std::vector<std::pair<int32_t, int32_t>> rt;
                rt.clear();
                row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) { rt.emplace_back((int32_t)j, (int32_t)nj); });
                int32_t& at = hist[(size_t)g * MAXCOST + key];
                for (int c = 0; c < C; ++c)
                    for (const auto& jn : rt) {
                        Task t;
                        t.row = r; t.chan = c; t.j0 = jn.first; t.nj = jn.second;
                        tmp[(size_t)at++] = t;
                    }
                continue;
            }
            row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) {
                int32_t& at = hist[(size_t)g * MAXCOST + cost(j, nj)];
                for (int c = 0; c < C; ++c) {
                    Task t;
                    t.row = r; t.chan = c; t.j0 = (int32_t)j; t.nj = nj;
                    tmp[(size_t)at++] = t;
                }
            });
        }
        // interleave: position i takes the next task of range i % groups (or of the next range that still has one)
        out.resize((size_t)total);
        int32_t next[64], end[64];
        int32_t off = 0;
        for (int g = 0; g < groups; ++g) { next[g] = off; off += gcount[g]; end[g] = off; }
        if (main_out) *main_out = total;
        if (tail_pct > 0 && tail_pct < 100) {
            // dynamic queues with a shared tail (ss_set_task_queue(1)): the first m tasks of every range stay in that range's queue
            // (list positions g, g + groups, ...: XCD g's queue), the smallest tail_pct % of the shortest range -- and whatever the
            // longer ranges have beyond m -- form ONE queue behind them, in descending cost, which every workgroup turns to when its
            // own queue is drained: the ranges' different speeds and the task granularity are evened out across the XCDs.
            int32_t m = INT32_MAX;
            for (int g = 0; g < groups; ++g) m = std::min(m, gcount[g]);
            m = (int32_t)((int64_t)m * (100 - tail_pct) / 100);
            out.resize((size_t)total);
            int32_t off = 0, start[64];
            for (int g = 0; g < groups; ++g) { start[g] = off; off += gcount[g]; }
            for (int32_t k = 0; k < m; ++k)
                for (int g = 0; g < groups; ++g) out[(size_t)k * groups + g] = tmp[(size_t)start[g] + k];
            size_t w = (size_t)m * groups;
            for (int g = 0; g < groups; ++g)
                for (int32_t k = m; k < gcount[g]; ++k) out[w++] = tmp[(size_t)start[g] + k];
            std::stable_sort(out.begin() + (size_t)m * groups, out.end(),
                             [&](const Task& a, const Task& b) { return cost(a.j0, a.nj) > cost(b.j0, b.nj); });
            if (main_out) *main_out = m * groups;
            return;
        }
        // the nwg / groups workgroups of a range take its list round-robin; dealing every second round in the opposite direction
        // (boustrophedon) keeps the first workgroup from collecting the largest task of every round
        const int bins = nwg > 0 && nwg % groups == 0 ? nwg / groups : 0;
        if (bins > 1)
            for (int g = 0; g < groups; ++g)
                for (int32_t r0 = next[g] + bins; r0 + bins <= end[g]; r0 += 2 * bins) std::reverse(tmp.begin() + r0, tmp.begin() + r0 + bins);
        for (int32_t i = 0; i < total; ++i) {
            int g = i % groups;
            for (int k = 0; k < groups && next[g] >= end[g]; ++k) g = (g + 1) % groups;
            out[(size_t)i] = tmp[(size_t)next[g]++];
        }
        return;
    }
    scratch.assign(MAXCOST + 1, 0);
    // pass 1: histogram of costs
    for (int r = 0; r < P; ++r) {
        const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
        row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) { scratch[cost(j, nj)] += C; });
    }
    // descending-cost start offsets
    int32_t total = 0;
    for (int c = MAXCOST - 1; c >= 0; --c) { const int32_t n = scratch[c]; scratch[c] = total; total += n; }
    out.resize((size_t)total);
    // pass 2: scatter
    for (int r = 0; r < P; ++r) {
        const int64_t a0 = seg_start[r > 0 ? r - 1 : r], a2 = seg_start[r < P - 1 ? r + 1 : r];
        row_tasks(a0, a2, block, jmax, rs, [&](int j, int nj) {
            int32_t& at = scratch[cost(j, nj)];
            for (int c = 0; c < C; ++c) {
                Task t;
                t.row = r; t.chan = c; t.j0 = (int32_t)j; t.nj = nj;
                out[(size_t)at++] = t;
            }
        });
    }
}

// One task list for SEVERAL sources rendered by one persistent launch (a SonicSet scene: 3 moving + 2 static renders).  Task.chan carries
// the source in its upper half (source << 16 | channel: the assembly kernel looks the source's bank / spectra / output / segment table up
// in its argument table).  Per source the rows are cut into `groups` time ranges of equal cost exactly as in plan_seg_lpt (a static source,
// P == 1, is one row over every block: its ranges are stretches of blocks); queue g = the g-th range of source 0, then of source 1, ...
// (an XCD works through one source's stretch at a time, so its L2 holds one stretch of input spectra), every (source, range) in descending
// cost.  Output layout = plan_seg_lpt's two-level form: the first m tasks of every queue interleaved (position k * groups + g), the rest --
// the smallest tail_pct % -- in ONE shared queue behind them, descending cost; *main_out = m * groups.
struct SceneSrc {
    const int64_t* seg_start;   // [P] (last == T) for a moving source; ignored for P == 1
    int P;
};
inline void plan_scene_lpt(const SceneSrc* src, int nsrc, int64_t T, int C, int block, int jmax, int NP, std::vector<Task>& out, int groups,
                           int tail_pct, int32_t* main_out) {
    auto cost = [NP](int j0, int nj) { return task_cost(std::min(NP, j0 + nj), nj); };
    static thread_local std::vector<std::vector<Task>> q;
    static thread_local std::vector<Task> part;
    if (groups < 1) groups = 1;
    if (groups > 64) groups = 64;
    q.resize((size_t)groups);
    for (auto& v : q) v.clear();
    const int64_t nblk = (T + block - 1) / block;
    for (int s = 0; s < nsrc; ++s) {
        // (row, first block, blocks) of this source in time order + its total cost
        struct RT { int32_t row, j0, nj, c; };
        static thread_local std::vector<RT> rt;
        rt.clear();
        int64_t total = 0;
        if (src[s].P <= 1) {
            for (int64_t j = 0; j < nblk; j += jmax) {
                const int nj = (int)std::min<int64_t>(jmax, nblk - j);
                rt.push_back(RT{0, (int32_t)j, nj, cost((int)j, nj)});
                total += rt.back().c;
            }
        } else {
            const int P = src[s].P;
            const int64_t* ss = src[s].seg_start;
            for (int r = 0; r < P; ++r) {
                const int64_t a0 = ss[r > 0 ? r - 1 : r], a2 = ss[r < P - 1 ? r + 1 : r];
                row_tasks(a0, a2, block, jmax, 0, [&](int j, int nj) { rt.push_back(RT{r, j, nj, cost(j, nj)}); total += rt.back().c; });
            }
        }
        // ranges of equal cost (whole rows stay together like in plan_seg_lpt: the range of a row is that of its cost midpoint)
        int64_t acc = 0;
        size_t i = 0;
        while (i < rt.size()) {
            size_t e = i;
            int64_t rc = 0;
            while (e < rt.size() && (src[s].P > 1 ? rt[e].row == rt[i].row : e == i)) rc += rt[e++].c;
            int g = total > 0 ? (int)((__int128)(acc + rc / 2) * groups / total) : 0;
            if (g >= groups) g = groups - 1;
            part.clear();
            for (size_t k = i; k < e; ++k)
                for (int c = 0; c < C; ++c) part.push_back(Task{rt[k].row, (int32_t)((s << 16) | c), rt[k].j0, rt[k].nj});
            q[(size_t)g].insert(q[(size_t)g].end(), part.begin(), part.end());
            acc += rc;
            i = e;
        }
        // descending cost inside this source's share of every queue (stable: time order inside a cost class); the key of a task is the cost
        // of its row's most expensive task, so that the tasks of one row stay neighbours and share the row's taps in their XCD's L2 (plan_seg_lpt)
        static thread_local std::vector<int32_t> rowmax;
        rowmax.assign((size_t)(src[s].P > 1 ? src[s].P : 1), 0);
        if (src[s].P > 1)
            for (const RT& e : rt) rowmax[(size_t)e.row] = std::max(rowmax[(size_t)e.row], e.c);
        auto key = [&](const Task& t) { return src[s].P > 1 ? (int)rowmax[(size_t)t.row] : cost(t.j0, t.nj); };
        for (int g = 0; g < groups; ++g) {
            auto& v = q[(size_t)g];
            auto first = std::find_if(v.begin(), v.end(), [s](const Task& t) { return (t.chan >> 16) == s; });
            std::stable_sort(first, v.end(), [&](const Task& a, const Task& b) { return key(a) > key(b); });
        }
    }
    size_t total_n = 0, m = (size_t)-1;
    for (const auto& v : q) { total_n += v.size(); m = std::min(m, v.size()); }
    if (groups == 1) tail_pct = 0;
    m = m * (size_t)(100 - (tail_pct > 0 && tail_pct < 100 ? tail_pct : 0)) / 100;
    out.resize(total_n);
    for (size_t k = 0; k < m; ++k)
        for (int g = 0; g < groups; ++g) out[k * (size_t)groups + (size_t)g] = q[(size_t)g][k];
    size_t w = m * (size_t)groups;
    for (int g = 0; g < groups; ++g)
        for (size_t k = m; k < q[(size_t)g].size(); ++k) out[w++] = q[(size_t)g][k];
    std::stable_sort(out.begin() + (std::ptrdiff_t)(m * (size_t)groups), out.end(),
                     [&](const Task& a, const Task& b) { return cost(a.j0, a.nj) > cost(b.j0, b.nj); });
    if (main_out) *main_out = (int32_t)(m * (size_t)groups);
}

// Rows cut into MANY tasks: transform the row once.  A trajectory of a few points over a long signal -- what SonicSet.py:40 takes from
// SonicSim_rir.get_nav_idx (SonicSim_rir.py:1064: a navmesh shortest path, a handful to a few dozen points over 60 s) -- makes every filter
// row span tens of output blocks, i.e. many tasks of <= jmax blocks, and each of them used to stream and forward-transform the whole row
// again (P = 12 at config-2 shapes: ~11 transforms of every tap; P = 3: ~60).  Rows with at least `min_tasks` tasks (per channel) get a SLOT
// in a partition-spectra array that a pre-pass fills once (k_row_spectra: [slot][NP][4096] c32 in the transform's slot order) and their
// tasks are marked  Task.nj = blocks | 0x100 | slot << 9 : the assembly kernel then only loads the row's spectrum per partition and
// multiply-accumulates (tools/gen_asm/os13.py, HROW).  Any task list of the assembly engine can be post-processed (implicit, explicit,
// scene); the list's order is untouched.  rows_out[k] = source << 24 | row for slot k * C + chan.  Returns the number of rows marked.
// `budget_bytes` bounds the spectra array (it should stay cache resident: 2 x the rows' taps), `max_rows` the pre-pass's row table.
constexpr int32_t TASK_NJ_MASK = 0xff, TASK_SPECTRA_READY = 0x100, TASK_SLOT_SHIFT = 9;
constexpr int HROW_MAX = 256;
inline int flag_long_rows(std::vector<Task>& tasks, const int32_t* Ps, int nsrc, int C, int NP, int min_tasks, int64_t budget_bytes,
                          int max_rows, std::vector<int32_t>& rows_out) {
    rows_out.clear();
    if (tasks.empty() || nsrc < 1 || nsrc > 8 || C < 1) return 0;
    static thread_local std::vector<int32_t> cnt;
    int32_t off[9];
    off[0] = 0;
    for (int s = 0; s < nsrc; ++s) off[s + 1] = off[s] + (Ps[s] > 1 ? Ps[s] : 1);
    cnt.assign((size_t)off[nsrc], 0);
    for (const Task& t : tasks)
        if ((t.chan & 0xffff) == 0) cnt[(size_t)off[t.chan >> 16] + (size_t)t.row]++;
    const int64_t per_row = (int64_t)C * NP * (int64_t)(sizeof(c32) * B13);
    int n = 0;
    for (int s = 0; s < nsrc; ++s)
        for (int32_t r = 0; r < off[s + 1] - off[s]; ++r) {
            int32_t& k = cnt[(size_t)off[s] + (size_t)r];
            if (k >= min_tasks && n < max_rows && (int64_t)(n + 1) * per_row <= budget_bytes && (int64_t)(n + 1) * C < ((int64_t)1 << 22)) {
                rows_out.push_back((int32_t)(s << 24) | r);
                k = -(++n);                                   // -(slot row + 1)
            } else {
                k = 0;
            }
        }
    if (!n) return 0;
    for (Task& t : tasks) {
        const int32_t k = cnt[(size_t)off[t.chan >> 16] + (size_t)t.row];
        if (k < 0) t.nj = (t.nj & TASK_NJ_MASK) | TASK_SPECTRA_READY | (((-k - 1) * C + (t.chan & 0xffff)) << TASK_SLOT_SHIFT);
    }
    return n;
}

// fixed receiver: one row, every block, store pass only
inline void build_plan_fixed(int64_t T, int C, int block, int jmax, Plan& plan) {
    plan.tasks[0].clear();
    plan.tasks[1].clear();
    const int64_t nblk = (T + block - 1) / block;
    plan.pairs = nblk;
    for (int64_t j = 0; j < nblk; j += jmax) {
        for (int c = 0; c < C; ++c) {
            Task t;
            t.row = 0;
            t.chan = c;
            t.j0 = (int32_t)j;
            t.nj = (int32_t)std::min<int64_t>(jmax, nblk - j);
            plan.tasks[0].push_back(t);
        }
    }
}

// constant tables in double precision (see tvfir_core.h layout)
inline void build_consts(std::vector<c32>& tab) {
    tab.assign(CONST_C32, c32{0.f, 0.f});
    const double PI = 3.14159265358979323846264338327950288;
    auto W = [&](double num, double den) {   // exp(-2 pi i num/den)
        const double a = -2.0 * PI * num / den;
        return c32{(float)std::cos(a), (float)std::sin(a)};
    };
    for (int k = 1; k < 8; ++k)
        for (int t = 0; t < 256; ++t) tab[TW1_OFF + (k - 1) * 256 + t] = W((double)((t * k) % 2048), 2048.0);
    for (int k = 1; k < 8; ++k)
        for (int m = 0; m < 32; ++m) tab[TW2_OFF + (k - 1) * 32 + m] = W((double)((m * k) % 256), 256.0);
    for (int k = 1; k < 8; ++k)
        for (int n = 0; n < 4; ++n) tab[TW3_OFF + (k - 1) * 4 + n] = W((double)((n * k) % 32), 32.0);
    for (int n = 0; n < 2048; ++n) tab[TWIST_OFF + n] = W((double)n, 8192.0);   // exp(-i pi n / 4096)
}

inline void build_consts12(std::vector<c32>& tab) {
    tab.assign(CONST12_C32, c32{0.f, 0.f});
    const double PI = 3.14159265358979323846264338327950288;
    auto W = [&](double num, double den) {
        const double a = -2.0 * PI * num / den;
        return c32{(float)std::cos(a), (float)std::sin(a)};
    };
    for (int k = 1; k < 8; ++k)
        for (int t = 0; t < 512; ++t) tab[TW1_12 + (k - 1) * 512 + t] = W((double)((t * k) % 4096), 4096.0);
    for (int k = 1; k < 8; ++k)
        for (int m = 0; m < 64; ++m) tab[TW2_12 + (k - 1) * 64 + m] = W((double)((m * k) % 512), 512.0);
    for (int k = 1; k < 8; ++k)
        for (int n = 0; n < 8; ++n) tab[TW3_12 + (k - 1) * 8 + n] = W((double)((n * k) % 64), 64.0);
    for (int n = 0; n < 4096; ++n) tab[TWIST_12 + n] = W((double)n, 16384.0);   // exp(-i pi n / 8192)
}


// geometry 13: TW1P merges the per-thread part of the right-angle twist into the pass-1 twiddles
inline void build_consts13(std::vector<c32>& tab) {
    tab.assign(CONST13_C32, c32{0.f, 0.f});
    const double PI = 3.14159265358979323846264338327950288;
    auto W = [&](double num, double den) {
        const double a = -2.0 * PI * num / den;
        return c32{(float)std::cos(a), (float)std::sin(a)};
    };
    for (int k = 0; k < 8; ++k)
        for (int t = 0; t < 512; ++t) tab[TW1P_13 + k * 512 + t] = W((double)((t * (1 + 4 * k)) % 16384), 16384.0);   // exp(-i pi t/8192) W_4096^(t k)
    for (int k = 1; k < 8; ++k)
        for (int m = 0; m < 64; ++m) tab[TW2_13 + (k - 1) * 64 + m] = W((double)((m * k) % 512), 512.0);
    for (int k = 1; k < 8; ++k)
        for (int n = 0; n < 8; ++n) tab[TW3_13 + (k - 1) * 8 + n] = W((double)((n * k) % 64), 64.0);
}


// assembly engine (tools/gen_asm/os13.py): row-per-reader tables, row stride 10 c32; image = LDS bytes 0x10000..0x1C000
constexpr int CONST14_C32 = 12 * 512;
inline void build_consts14(std::vector<c32>& tab) {
    tab.assign(CONST14_C32, c32{0.f, 0.f});
    const double PI = 3.14159265358979323846264338327950288;
    auto W = [&](double num, double den) {
        const double a = -2.0 * PI * num / den;
        return c32{(float)std::cos(a), (float)std::sin(a)};
    };
    for (int t = 0; t < 512; ++t)
        for (int k = 0; k < 8; ++k) tab[t * 10 + k] = W((double)((t * (1 + 4 * k)) % 16384), 16384.0);
    const int o2 = (0x1A000 - 0x10000) / 8, o3 = (0x1B400 - 0x10000) / 8;
    for (int m = 0; m < 64; ++m)
        for (int k = 0; k < 8; ++k) tab[o2 + m * 10 + k] = W((double)((m * k) % 512), 512.0);
    for (int n = 0; n < 8; ++n)
        for (int k = 0; k < 8; ++k) tab[o3 + n * 10 + k] = W((double)((n * k) % 64), 64.0);
}

}  // namespace ss
