// sonicsim_hip.hip -- libsonicsim_hip.so: gfx950 kernels + the C-ABI of include/sonicsim_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see sonicsim_amd/build.py).
// Written for CDNA4 only (wave64, 160 KiB LDS, 256 CUs / 8 XCDs); no CUDA compatibility layer.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sonicsim_hip.h"
#include "plan.h"
#include "tvfir_core.h"
#include "stream13.h"

// Tuning / experiment switches (SS_OS_GEOM, SS_HSACO, SS_TRACE_FILE, SS_HOP_RS, ...) exist only in the library the tools build with
// -DSS_TUNING_KNOBS (sonicsim_amd/build.py::build_tuning, lib/libsonicsim_hip_tuning.so); the product library reads no environment variable.
#ifdef SS_TUNING_KNOBS
static inline const char* knob(const char* name) { return getenv(name); }
#else
static inline const char* knob(const char*) { return nullptr; }
#endif


using namespace ss;

// =============================================================================================
// device side
// =============================================================================================
struct DevEnv {
    c32* smem;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    // same-wave LDS hand-off: DS instructions of one wave execute in order, so only the compiler must be
    // kept from reordering across this point
    __device__ __forceinline__ void wave_sync() const {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ c32* lds() const { return smem; }
    __device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
};

// input spectra: one workgroup per 2B window
__global__ __launch_bounds__(256, 2) void k_xspec(const float* __restrict__ x, int64_t T, const c32* __restrict__ consts,
                                                  c32* __restrict__ Xs, int M) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS_C32];
    DevEnv env{smem};
    xspec_body(env, x, T, consts, Xs, (int)blockIdx.x, M);
}

// row-stationary partitioned overlap-save: one workgroup per Task, 2 workgroups resident per CU
// XD = number of input spectra prefetched ahead of the MAC loop (0 = load at use)
template <int XD, int ABL = 0> __global__ __launch_bounds__(256, 2) void k_os(RenderParams prm) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS_C32];
    DevEnv env{smem};
    os_body<DevEnv, XD, ABL>(env, prm, (int)blockIdx.x);
}

// geometry 12 (B = 4096, 512 threads, sliding spectrum window): one workgroup per CU
__global__ __launch_bounds__(512, 2) void k_xspec12(const float* __restrict__ x, int64_t T, const c32* __restrict__ consts,
                                                    c32* __restrict__ Xs, int M, float* __restrict__ yzero, int64_t nzero,
                                                    int* __restrict__ counter) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS12_C32];
    DevEnv env{smem};
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;     // task-queue head of the render kernel that follows
    xspec12_body(env, x, T, consts, Xs, (int)blockIdx.x, M, yzero, nzero);
}
template <int ABL = 0> __global__ __launch_bounds__(512, 2) void k_os12(RenderParams prm) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS12_C32];
    DevEnv env{smem};
    os12_body<DevEnv, ABL>(env, prm, (int)blockIdx.x, (int)gridDim.x);
}

// partition spectra of the filter rows that are cut into many tasks (plan.h flag_long_rows): the row is transformed ONCE here, its tasks in
// k_os13_asm only multiply-accumulate.  Workgroup (partition p, slot = table row * C + channel); layout [slot][NP][4096] c32 in slot order
// (stream_store_slots = what the render kernel's own forward transform leaves in its pending-spectrum registers).  Reference: the row's
// convolution, SonicSim_moving.py:86 -- the same taps, transformed by the same geometry-13 transform as the streaming renderer's rows.
struct HRowArgs {
    const float* bank[8];     // per source
    c32* Hs;
    const c32* consts;        // geometry-13 table
    int32_t nrows, C, L, NP;
    int32_t rows[HROW_MAX];   // source << 24 | row
};
constexpr int HROW_PPW = 3;      // partitions per workgroup
__device__ __forceinline__ void row_spectra_wg(DevEnv& env, const HRowArgs& a, int b) {
    const int npw = (a.NP + HROW_PPW - 1) / HROW_PPW;
    const int p0 = (b % npw) * HROW_PPW, slot = b / npw;
    const int e = a.rows[slot / a.C], c = slot % a.C;
    const float* h = a.bank[e >> 24] + ((int64_t)(e & 0xffffff) * a.C + c) * a.L;
    row_spectra_body<HROW_PPW>(env, h, a.L, a.NP, p0, a.consts, a.Hs + (int64_t)slot * a.NP * B13);
}
static inline int hrow_workgroups(const HRowArgs& a) { return a.nrows > 0 ? a.nrows * a.C * ((a.NP + HROW_PPW - 1) / HROW_PPW) : 0; }
// stand-alone launch (host-pointer renders: the bank arrives behind the input-spectra kernel); resident banks ride on k_xspec13's launch
__global__ __launch_bounds__(512, 2) void k_row_spectra(HRowArgs a) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDSFWD13_C32];
    DevEnv env{smem};
    row_spectra_wg(env, a, (int)blockIdx.x);
}

// input spectra for the assembly / geometry-13 render kernels (no zero spectrum: their descriptors return zeros out of range)
// counter: task-queue heads of the render kernel that follows -- ncnt words, 64 bytes apart, each set to cnt_init
__global__ __launch_bounds__(512, 2) void k_xspec13(const float* __restrict__ x, int64_t T, const c32* __restrict__ consts,
                                                    c32* __restrict__ Xs, int M, float* __restrict__ yzero, int64_t nzero,
                                                    int* __restrict__ counter, int ncnt, int cnt_init, const float* __restrict__ xdiv, int rs,
                                                    const uint4* __restrict__ plan_src, uint4* __restrict__ plan_dst, int plan_n16,
                                                    const int32_t* __restrict__ fail_flag, const HRowArgs hrow, int nrow_wg) {
    // nrow_wg > 0: the FIRST nrow_wg workgroups form the partition spectra of the rows that are cut into many tasks (k_row_spectra's body: three
    // transforms each, so they start first); the other M + 1 are the input spectra.  One launch, one boundary (round 6).
    __shared__ __attribute__((aligned(16))) c32 smem[LDSFWD13_C32];
    DevEnv env{smem};
    if (counter && blockIdx.x == 0 && (int)threadIdx.x < ncnt) counter[16 * threadIdx.x] = cnt_init;
    // the render's plan (segment table + task list, ~31 KB at config 2) sits in pinned host memory; every workgroup moves its slice to
    // HBM here, the PCIe round trip hidden behind its transform -- the render kernel then never reads across PCIe (round 3: its
    // ~10 000 scalar loads per render from host memory stalled ONE launch in ~250 for 0.3-0.65 ms, profiles/r03b, and an in-stream
    // hipMemcpyAsync instead costs 11 us per render, profiles/r03c)
    uint4 pv = make_uint4(0, 0, 0, 0);
    const int pi = 4 * ((int)blockIdx.x + (int)gridDim.x * ((int)threadIdx.x >> 2)) + ((int)threadIdx.x & 3);   // 64-byte units dealt round-robin
    const bool pok = plan_src && pi < plan_n16;                                                                  // to the workgroups: a few PCIe reads each
    if (pok) pv = plan_src[pi];
    // SS_FLAG_ASYNC_PLAN: k_plan_explicit (earlier on this stream) found the schedule too irregular for its task buffer and planned
    // nothing -- y is then filled with NaN instead of zeros: the failure cannot pass as valid silence even if nobody polls
    // ss_async_status (word 2 of the status record is written by every planner run; words 0-1 are the latched error)
    const float fill = (fail_flag && fail_flag[2] != 0) ? __builtin_nanf("") : 0.0f;
    if ((int)blockIdx.x < nrow_wg) row_spectra_wg(env, hrow, (int)blockIdx.x);
    else xspec13_body<DevEnv, LdsFwd13>(env, x, T, consts, Xs, (int)blockIdx.x - nrow_wg, M, yzero, nzero, xdiv, rs, fill);
    if (pok) plan_dst[pi] = pv;
    if (plan_src)       // (a plan larger than the grid's 8 KB per workgroup: the remainder in a strided loop)
        for (int i = pi + (int)gridDim.x * 512; i < plan_n16; i += (int)gridDim.x * 512) plan_dst[i] = plan_src[i];
}

// streaming render with persistent state (stream13.h): filter row -> partition spectra (grid NP x C); one piece of a push (grid C)
__global__ __launch_bounds__(512, 2) void k_stream_rows(StreamDev a, int row) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS13_C32];
    DevEnv env{smem};
    stream_row_body(env, a, row, (int)blockIdx.x, (int)blockIdx.y);
}
__global__ __launch_bounds__(512, 2) void k_stream_push(StreamDev a, StreamPiece pc) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS13_C32];
    DevEnv env{smem};
    stream_push_body(env, a, pc, (int)blockIdx.x);
}

// the same for the sources of ONE scene launch (ss_convolve_scene_f32): blockIdx.y = source; every source has its own dry signal,
// divisor (deferred peak normalisation, may be null), spectra array and output (zero fill)
struct XspecSrcTab {
    const float* x[8];
    const float* xdiv[8];
    c32* Xs[8];
    float* y[8];
};
__global__ __launch_bounds__(512, 2) void k_xspec13_multi(XspecSrcTab tab, int64_t T, const c32* __restrict__ consts, int M, int64_t nzero,
                                                          int* __restrict__ counter, int ncnt, int cnt_init, const uint4* __restrict__ plan_src,
                                                          uint4* __restrict__ plan_dst, int plan_n16, const HRowArgs hrow, int nrow_wg) {
    // 1-D grid: nrow_wg row-spectra workgroups (see k_xspec13), then (M + 1) input-spectra workgroups per source
    __shared__ __attribute__((aligned(16))) c32 smem[LDSFWD13_C32];
    DevEnv env{smem};
    const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
    const int s = bid < nrow_wg ? 0 : (bid - nrow_wg) / (M + 1), m = bid < nrow_wg ? 0 : (bid - nrow_wg) % (M + 1);
    if (counter && bid == 0 && (int)threadIdx.x < ncnt) counter[16 * threadIdx.x] = cnt_init;
    uint4 pv = make_uint4(0, 0, 0, 0);
    const int pi = 4 * (bid + nblk * ((int)threadIdx.x >> 2)) + ((int)threadIdx.x & 3);      // plan staging: see k_xspec13
    const bool pok = plan_src && pi < plan_n16;
    if (pok) pv = plan_src[pi];
    if (bid < nrow_wg) row_spectra_wg(env, hrow, bid);
    else xspec13_body<DevEnv, LdsFwd13>(env, tab.x[s], T, consts, tab.Xs[s], m, M, tab.y[s], nzero, tab.xdiv[s], 0, 0.0f);
    if (pok) plan_dst[pi] = pv;
    if (plan_src)
        for (int i = pi + nblk * 512; i < plan_n16; i += nblk * 512) plan_dst[i] = plan_src[i];
}

// geometry 13 (tvfir13.h): software-pipelined FFT/MAC, one barrier per transform, buffer addressing, dynamic task queue
__global__ __launch_bounds__(512, 2) void k_os13(Params13 prm) {
    __shared__ __attribute__((aligned(16))) c32 smem[LDS13_C32];
    DevEnv env{smem};
    os13_body(env, prm, (int)blockIdx.x);
}

// direct-form fallback / cross-check
__global__ __launch_bounds__(256) void k_direct(RenderParams prm) {
    __shared__ __attribute__((aligned(16))) float smemf[DCHUNK + DTILE + DCHUNK];
    DevEnv env{reinterpret_cast<c32*>(smemf)};
    direct_body(env, prm, (int)blockIdx.x);
}

// per-DTILE min/max of idx (explicit schedule): one workgroup per tile
__global__ __launch_bounds__(256) void k_idx_minmax(const int64_t* __restrict__ idx, int64_t T, int32_t* __restrict__ bmin,
                                                    int32_t* __restrict__ bmax) {
    __shared__ long long smin[4], smax[4];
    const int64_t t0 = (int64_t)blockIdx.x * DTILE;
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    for (int i = threadIdx.x; i < DTILE; i += 256) {
        const int64_t t = t0 + i;
        if (t < T) {
            const long long v = idx[t];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { lo = smin[i] < lo ? smin[i] : lo; hi = smax[i] > hi ? smax[i] : hi; }
        // clamp into int32 so out-of-range values are still detected on the host
        const long long big = 0x7fffffffLL;
        bmin[blockIdx.x] = (int32_t)(lo < -big ? -big : (lo > big ? big : lo));
        bmax[blockIdx.x] = (int32_t)(hi < -big ? -big : (hi > big ? big : hi));
    }
}

#ifdef SS_DEBUG_CLK
__device__ unsigned long long g_dbg_clk[12][2][256];
#define DBG_CLK_T(kid, which, thr) do { if (threadIdx.x == (thr)) { const int wg_ = blockIdx.x + gridDim.x * blockIdx.y; if (wg_ < 256) g_dbg_clk[kid][which][wg_] = wall_clock64(); } } while (0)
#else
#define DBG_CLK_T(kid, which, thr) do {} while (0)
#endif
#define DBG_CLK(kid, which) DBG_CLK_T(kid, which, 0)
// ---------------------------------------------------------------------------------------------
// Device-side planner of the explicit (idx, w) schedule for the assembly engine (SS_FLAG_ASYNC_PLAN): the task list of plan.h's
// build_plan + merge_lpt_xcd, produced on the stream from k_idx_minmax's tile bounds -- no copy to the host, no synchronisation.
// One workgroup of 1024 threads; every stage is a strided loop, so any T / P works.  The ORDER of the list only matters for speed (every output sample receives
// exactly two float atomics, a commutative sum), so the rule is restated in a form that needs no sequential pass:
//   * row-tasks (row, j0, nj) in time order; cost as in plan.h; `groups` contiguous ranges of equal total cost (one per XCD);
//   * inside a range by descending cost (ties in time order), channels of a row-task adjacent;
//   * list position = round-robin over the ranges (element e of range g precedes element e of range g + 1).
// Out-of-range interp_index values select no filter (the kernel's epilogue compares row - idx with 0 / 1); the first offending
// tile is latched in status[] for ss_async_status.
struct PlanDevArgs {
    const int32_t* bmin;
    const int32_t* bmax;
    int64_t nfine;
    int32_t fine_per_block, nblk, P, C, jmax, NP, groups, cap_rows, rs;      // rs: Task::j0 is emitted in hop units (j0 << rs)
    int32_t* lo;       // [nblk]
    int32_t* hi;       // [nblk]
    int32_t* first;    // [P]
    int32_t* last;     // [P]
    int32_t* rcount;   // [P + 1]
    int32_t* rtask;    // [cap_rows][4]: row, j0, nj, cost
    unsigned long long* keys;   // [cap_rows]
    int32_t* bins;     // [groups * 4096] counting-sort bins
    int32_t* out;      // header (16 bytes: ntasks, 0, 0, 0) + Task[cap_rows * C]
    int32_t* status;   // [0] code (1 = interp_index out of range, 2 = plan capacity exceeded), [1] tile / count, latched: ONE record per device (atomicCAS)
    int32_t* call;     // THIS call's verdict, words [2] too irregular, [3] out of range, [4] its tile: a record of the stream's workspace LANE (round 6:
                       // in the per-device record a planner on another stream could overwrite them between this planner's write and this render's
                       // spectra kernel reading word 2 -- a failed render zero-filled, a valid one NaN-filled)
    int32_t* status_host;   // pinned host mirror of THIS run's words {out of range, tile, too irregular} (null: nobody asked): the validating call
                            // reads it after its stream synchronisation instead of paying a device-to-host copy of 32 bytes
};

__device__ inline int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline int plan_cost(int NP, int j0, int nj) {
    const int np_eff = NP < j0 + nj ? NP : j0 + nj;
    const int c = task_cost(np_eff, nj);
    return c < 4096 ? c : 4095;
}

// exclusive prefix sum of v[0..n) in place (64-bit running total returned to every thread), one workgroup
template <typename T>
__device__ inline long long block_exclusive_scan(T* v, int n, long long* sh /*[17]*/) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    long long carry = 0;
    for (int base = 0; base < n; base += nt) {
        const int i = base + tid;
        const long long x = i < n ? (long long)__hip_atomic_load(&v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;   // (entries may have been updated by atomics)
        long long incl = x;
        for (int o = 1; o < 64; o <<= 1) {
            const long long y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) sh[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            long long run = 0;
            for (int k = 0; k < nw; ++k) { const long long t = sh[k]; sh[k] = run; run += t; }
            sh[16] = run;
        }
        __syncthreads();
        if (i < n) v[i] = (T)(carry + sh[wave] + incl - x);
        carry += sh[16];
        __syncthreads();
    }
    return carry;
}

__device__ __forceinline__ void plan_explicit_body(PlanDevArgs a) {
    __shared__ long long sh[17];
    __shared__ int gcount[64];
    __shared__ int crange[2];
    __shared__ int oor_tile;               // first tile of THIS call with an index out of range (status[3..4]: per call, not latched)
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) oor_tile = INT32_MAX;
    DBG_CLK(0, 0);
    // round 4: the planner's scratch arrays live in LDS whenever they fit (they do at every BASELINE.json shape: 4 KB at config 2, 17 KB at
    // config 5) -- a dozen dependent phases then cost LDS round trips instead of trips to the L2 (27.5 -> 23.9 us per call, profiles/r04aq: most of
    // the kernel is the phases' own serial work and synchronisations);
    // the global arrays of PlanDevArgs remain the fallback for schedules that do not fit.  `a` is this kernel's own copy of the arguments.
    constexpr int POOL = 14336;                                    // 56 KB of the 64 KB a workgroup may declare statically
    __shared__ __attribute__((aligned(16))) int32_t pool[POOL];
    int used = 0;
    {
        const long long need = 2LL * a.nblk + 3LL * a.P + 1;
        if (need <= POOL) {
            a.lo = pool; a.hi = pool + a.nblk; a.first = pool + 2 * a.nblk; a.last = a.first + a.P; a.rcount = a.last + a.P;
            used = (int)((need + 3) & ~3LL);
        }
    }
    __syncthreads();
    // ---- 1: per output block min/max of idx, range check, clamp
    // round 5: ONE tile per thread and a shuffle reduction over the fine_per_block lanes of a block (16 at B = 4096) instead of one thread walking the
    // 16 tiles of its block through 32 dependent trips to the L2 (the branch with the status atomics keeps the compiler from batching the loads)
    const int fpb = a.fine_per_block;
    const bool lanes_per_block = fpb >= 2 && fpb <= 64 && (fpb & (fpb - 1)) == 0 && (nt % fpb) == 0;
    if (lanes_per_block) {
        const long long units = (long long)a.nblk * fpb;
        // every round's loads are issued before any of them is used: the rounds' trips to the L2 overlap instead of following each other
        constexpr int PF = 8;
        int pl[PF], ph[PF];
        const bool batched = units <= (long long)PF * nt;
        if (batched) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const long long u = (long long)k * nt + tid;
                const bool ok = u < units && u < a.nfine;
                pl[k] = ok ? ld_agent(a.bmin + u) : 0;
                ph[k] = ok ? ld_agent(a.bmax + u) : 0;
            }
        }
        int round = 0;
        for (long long base = 0; base < units; base += nt, ++round) {
            const long long u = base + tid;                        // tile u of the launch: block u / fpb (whole groups of fpb lanes stay inside a wave)
            int lo = INT32_MAX, hi = INT32_MIN;
            if (u < units && u < a.nfine) {
                int l, h;
                if (batched) {
                    l = pl[0]; h = ph[0];
#pragma unroll
                    for (int k = 1; k < PF; ++k) if (round == k) { l = pl[k]; h = ph[k]; }
                } else { l = ld_agent(a.bmin + u); h = ld_agent(a.bmax + u); }
                if (l < 0 || h > a.P - 2) {
                    if (atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = (int32_t)(u < INT32_MAX ? u : INT32_MAX);
                    atomicMin(&oor_tile, (int)(u < INT32_MAX ? u : INT32_MAX - 1));
                }
                lo = l; hi = h;
            }
            for (int o = 1; o < fpb; o <<= 1) {
                const int l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if (u < units && (tid & (fpb - 1)) == 0) {
                if (lo < 0) lo = 0;
                if (hi > a.P - 2) hi = a.P - 2;
                if (lo > hi) { lo = 1; hi = -1; }
                a.lo[u / fpb] = lo;
                a.hi[u / fpb] = hi;
            }
        }
    } else
    for (int j = tid; j < a.nblk; j += nt) {
        int lo = INT32_MAX, hi = INT32_MIN;
        for (int f = 0; f < a.fine_per_block; ++f) {
            const int64_t fb = (int64_t)j * a.fine_per_block + f;
            if (fb < a.nfine) {
                const int l = ld_agent(a.bmin + fb), h = ld_agent(a.bmax + fb);
                if (l < 0 || h > a.P - 2) {
                    if (atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = (int32_t)(fb < INT32_MAX ? fb : INT32_MAX);
                    atomicMin(&oor_tile, (int)(fb < INT32_MAX ? fb : INT32_MAX - 1));
                }
                lo = l < lo ? l : lo;
                hi = h > hi ? h : hi;
            }
        }
        if (lo < 0) lo = 0;
        if (hi > a.P - 2) hi = a.P - 2;
        if (lo > hi) { lo = 1; hi = -1; }          // nothing valid in this block: the row range lo .. hi + 1 is empty
        a.lo[j] = lo;
        a.hi[j] = hi;
    }
    for (int r = tid; r < a.P; r += nt) { a.first[r] = INT32_MAX; a.last[r] = -1; }
    if (tid < 64) gcount[tid] = 0;
    __syncthreads();
    if (tid == 0) { a.call[3] = oor_tile != INT32_MAX ? 1 : 0; a.call[4] = oor_tile != INT32_MAX ? oor_tile : 0; }
    DBG_CLK(0, 1);
    // ---- 2: first / last block of every row
    for (int j = tid; j < a.nblk; j += nt) {
        const int lo = a.lo[j], hi = a.hi[j];
        for (int r = lo; r <= hi + 1 && r < a.P; ++r) { atomicMin(&a.first[r], j); atomicMax(&a.last[r], j); }
    }
    __syncthreads();
    DBG_CLK(1, 0);
    // ---- 3: row-tasks per row (runs of consecutive blocks cut into pieces of at most jmax)
    for (int r = tid; r < a.P; r += nt) {
        const int f = ld_agent(&a.first[r]), l = ld_agent(&a.last[r]);
        int cnt = 0, run = 0;
        for (int j = f; j <= l; ++j) {
            const bool has = a.lo[j] <= r && r <= a.hi[j] + 1;
            if (has) { if (run % a.jmax == 0) ++cnt; ++run; } else run = 0;
        }
        a.rcount[r] = cnt;
    }
    __syncthreads();
    long long nrow = block_exclusive_scan(a.rcount, a.P, sh);
    DBG_CLK(1, 1);
    if (tid == 0) a.call[2] = nrow > a.cap_rows ? 1 : 0;      // per call (not latched): the spectra kernel of THIS render fills y with NaN
    if (nrow > a.cap_rows) {
        if (tid == 0 && atomicCAS(&a.status[0], 0, 2) == 0) a.status[1] = (int32_t)(nrow < INT32_MAX ? nrow : INT32_MAX);
        nrow = 0;                                   // render nothing rather than part of the schedule
    }
    const int N = (int)nrow;
    if (N > 0 && used + 6LL * N <= POOL) {                          // row-tasks (16 B) and sort keys (8 B) of the rows that exist, not of the capacity
        a.rtask = pool + used;
        a.keys = reinterpret_cast<unsigned long long*>(pool + used + 4 * N);     // (used and 4 N are multiples of 4 ints: 16-byte aligned)
        used += 6 * N;
    }
    // ---- 4: emit the row-tasks in time order
    if (N > 0)
        for (int r = tid; r < a.P; r += nt) {
            const int f = ld_agent(&a.first[r]), l = ld_agent(&a.last[r]);
            int k = a.rcount[r] - 1, run = 0, j0 = 0, nj = 0;
            for (int j = f; j <= l + 1; ++j) {
                const bool has = j <= l && a.lo[j] <= r && r <= a.hi[j] + 1;
                if (has && run % a.jmax == 0) {
                    if (nj) { int32_t* t = a.rtask + 4 * (size_t)k; t[0] = r; t[1] = j0; t[2] = nj; t[3] = plan_cost(a.NP, j0, nj); }
                    ++k; j0 = j; nj = 0;
                }
                if (has) { ++nj; ++run; }
                else {
                    if (nj) { int32_t* t = a.rtask + 4 * (size_t)k; t[0] = r; t[1] = j0; t[2] = nj; t[3] = plan_cost(a.NP, j0, nj); nj = 0; }
                    run = 0;
                }
            }
        }
    __syncthreads();
    // ---- 5: ranges of equal total cost; counting sort by (range, descending cost) -- O(N + bins)
    DBG_CLK(2, 0);
    if (tid == 0) { crange[0] = 4095; crange[1] = 0; }
    __syncthreads();
    {   // (round 5: one pair of LDS atomics per WAVE instead of per row-task -- N threads on two addresses serialise: 6 of the kernel's 24 us)
        int cmin_ = 4095, cmax_ = 0;
        for (int i = tid; i < N; i += nt) {
            const int c = a.rtask[4 * (size_t)i + 3];
            a.keys[i] = (unsigned long long)c;
            cmin_ = c < cmin_ ? c : cmin_;
            cmax_ = c > cmax_ ? c : cmax_;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int l2 = __shfl_xor(cmin_, o), h2 = __shfl_xor(cmax_, o);
            cmin_ = l2 < cmin_ ? l2 : cmin_;
            cmax_ = h2 > cmax_ ? h2 : cmax_;
        }
        if ((tid & 63) == 0 && cmin_ <= cmax_) { atomicMin(&crange[0], cmin_); atomicMax(&crange[1], cmax_); }
    }
    __syncthreads();
    int groups = a.groups;
    if (groups > 8) groups = 8;                          // bins[] holds 8 x 4096 classes
    if (groups < 1 || (long long)N * a.C < 2LL * groups) groups = 1;
    const int cmax = crange[1], span = N > 0 ? cmax - crange[0] + 1 : 1;     // only the costs that occur get a bin
    const int nbins = groups * span;
    if (used + nbins <= POOL) a.bins = pool + used;
    for (int i = tid; i < nbins; i += nt) a.bins[i] = 0;
    __syncthreads();
    const long long total = block_exclusive_scan(a.keys, N, sh);
    DBG_CLK(2, 1);
    for (int i = tid; i < N; i += nt) {
        const long long acc = (long long)a.keys[i];
        const int c = a.rtask[4 * (size_t)i + 3];
        int g = total > 0 ? (int)((acc + c / 2) * groups / total) : 0;        // total < 2^43: no overflow with groups <= 64
        if (g >= groups) g = groups - 1;
        atomicAdd(&gcount[g], 1);
        const int cls = g * span + (cmax - c);
        atomicAdd(&a.bins[cls], 1);
        a.keys[i] = ((unsigned long long)g << 32) | (unsigned long long)cls;
    }
    __syncthreads();
    block_exclusive_scan(a.bins, nbins, sh);             // start of every (range, cost) class in the sorted order
    DBG_CLK(3, 0);
    // ---- 6: place every row-task (the order inside a class is whatever the atomics give: equal cost, same range), then the
    //         round-robin position of each of its C tasks.  Round 5: the rank first (one LDS atomic per row-task), then ALL N * C tasks spread
    //         over the workgroup's threads, each written with ONE 16-byte store (it was N threads writing 4 C dwords each: 7.6 of the 24 us)
    __shared__ int gstart[65];
    for (int i = tid; i < N; i += nt) {
        const int cls = (int)(a.keys[i] & 0xffffffffu);
        const int rank = atomicAdd(&a.bins[cls], 1);
        a.keys[i] = (a.keys[i] & 0xffffffff00000000ull) | (unsigned long long)(unsigned int)rank;
    }
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < groups; ++q) { gstart[q] = run; run += gcount[q]; }
        gstart[groups] = run;
    }
    __syncthreads();
    {
        const int Cc = a.C;
        const long long NC = (long long)N * Cc;
        for (long long e1 = tid; e1 < NC; e1 += nt) {
            const int i = (int)(e1 / Cc), c = (int)(e1 - (long long)i * Cc);
            const unsigned long long key = a.keys[i];
            const int g = (int)(key >> 32), rank = (int)(key & 0xffffffffu);
            const int32_t* t = a.rtask + 4 * (size_t)i;
            const long long e = (long long)(rank - gstart[g]) * Cc + c;
            long long pos = 0;
            for (int q = 0; q < groups; ++q) {
                const long long cq = (long long)gcount[q] * Cc;
                pos += cq < e ? cq : e;
                if (q < g && cq > e) ++pos;
            }
            int4 o4;
            o4.x = t[0]; o4.y = c; o4.z = t[1] << a.rs; o4.w = t[2];
            *reinterpret_cast<int4*>(a.out + 4 + 4 * (size_t)pos) = o4;
        }
    }
    if (tid == 0) { a.out[0] = N * a.C; a.out[1] = 0; a.out[2] = 0; a.out[3] = 0; }
    DBG_CLK(3, 1);
    if (tid == 0 && a.status_host) {               // (thread 0 wrote all three words itself)
        a.status_host[1] = a.call[4];
        a.status_host[2] = a.call[2];
        __threadfence_system();                    // word 0 is the flag the host polls (round 6: the validating call returns when the PLANNER has spoken, not
        __hip_atomic_store(a.status_host, a.call[3], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // when the render has finished): it goes last
    }
}
__global__ __launch_bounds__(1024) void k_plan_explicit(PlanDevArgs a) { plan_explicit_body(a); }

// Everything an EXPLICIT (idx, w) render needs ahead of its persistent launch, in ONE launch (round 6): the per-tile min / max of idx
// (workgroups [0, n_mm): FRONT_MMT tiles each), the device planner (workgroup n_mm: it waits for the n_mm arrivals, then runs k_plan_explicit's body),
// the input spectra + zero fill (the other M + 1 workgroups).  The two planner kernels used to run serially ahead of the spectra kernel, which needs
// nothing of theirs: 4.5 + 15 us + two boundaries on the critical path of convolve_moving_receiver (SonicSim_moving.py:63-96).  Workgroups are
// dispatched in index order, so when the planner spins every workgroup it waits for is already resident: no deadlock.  (Two streams instead cost
// two cross-queue dependencies of ~8 us each: measured slower than the serial form, profiles/r06o.)  The hand-off of the bounds uses write-through
// stores, a drain and agent-scope loads -- NOT a release fence, which on this chip writes the XCD's dirty L2 back and cost 20-40 us beside the
// spectra workgroups' 38 MB of stores (profiles/r06p); with it the launch takes ~23 us where the two launches took 27 + a boundary (profiles/r06aa).
constexpr int FRONT_MMT = 8;
template <bool PLAN> __global__ __launch_bounds__(512, PLAN ? 1 : 2) void k_front_explicit(const float* __restrict__ x, int64_t T, const c32* __restrict__ consts, c32* __restrict__ Xs, int M,
                                                           float* __restrict__ yzero, int64_t nzero, int* __restrict__ counter, int ncnt, int cnt_init,
                                                           const float* __restrict__ xdiv, const int64_t* __restrict__ idx, int32_t* __restrict__ bmin,
                                                           int32_t* __restrict__ bmax, int n_mm, int* __restrict__ mm_done, PlanDevArgs pa) {
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (counter && b == 0 && tid < ncnt) counter[16 * tid] = cnt_init;
    if (b < n_mm) {      // one tile per WAVE (16 samples per lane, all loads in flight at once): FRONT_MMT = 8 tiles per workgroup, one trip to memory
        const int64_t nfine = (T + DTILE - 1) / DTILE;
        const int64_t tile = (int64_t)b * FRONT_MMT + (tid >> 6);
        const int lane = tid & 63;
        if (tile < nfine) {
            long long v[DTILE / 64];
#pragma unroll
            for (int k = 0; k < DTILE / 64; ++k) {
                const int64_t t = tile * DTILE + k * 64 + lane;
                v[k] = t < T ? idx[t] : idx[tile * DTILE];        // (a tile that exists has its first sample)
            }
            long long lo = v[0], hi = v[0];
#pragma unroll
            for (int k = 1; k < DTILE / 64; ++k) { lo = v[k] < lo ? v[k] : lo; hi = v[k] > hi ? v[k] : hi; }
            for (int o = 32; o > 0; o >>= 1) {
                const long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            if (lane == 0) {
                const long long big = 0x7fffffffLL;          // clamp into int32 so out-of-range values are still detected
                const int32_t vlo = (int32_t)(lo < -big ? -big : (lo > big ? big : lo)), vhi = (int32_t)(hi < -big ? -big : (hi > big ? big : hi));
                if constexpr (PLAN) {
                    // write-through stores + a drain instead of a release fence: an agent-scope RELEASE on this chip writes the XCD's dirty L2 lines back, and
                    // the spectra workgroups have 38 MB of stores in flight beside us (one fence per workgroup: 46 us for this launch; per wave: 67).  The
                    // planner reads the bounds with agent-scope loads (ld_agent), so no cache holds a stale copy -- the pattern of k_rir_synth's peak slots.
                    __hip_atomic_store(bmin + tile, vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(bmax + tile, vhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    bmin[tile] = vlo;
                    bmax[tile] = vhi;
                }
            }
        }
        if constexpr (PLAN) {
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(mm_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (PLAN && b == n_mm) {
        if (tid == 0) {
            while (__hip_atomic_load(mm_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_mm) __builtin_amdgcn_s_sleep(4);
            __hip_atomic_store(mm_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (the next call on this lane starts from zero)
        }
        __syncthreads();
        if constexpr (PLAN) plan_explicit_body(pa);
        return;
    }
    __shared__ __attribute__((aligned(16))) c32 smem[LDSFWD13_C32];
    DevEnv env{smem};
    xspec13_body<DevEnv, LdsFwd13>(env, x, T, consts, Xs, b - n_mm - (PLAN ? 1 : 0), M, yzero, nzero, xdiv, 0, 0.0f);
}


// ---------------------------------------------------------------------------------------------
// K1: synthetic RIR bank (row R).  Thread per (c,t), sequential AR(1) over positions.
struct RirDev {
    int32_t P, C, L;
    float tail_gain, rho, srho;
    double inv_tau;   // 6.91 / (rt60 * fs)
    uint32_t seed;
    const int32_t* delay;
    const float* dgain;
};

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t stream, uint64_t ctr) {
    const uint32_t k = seed * 0x9E3779B9u + stream;
    uint32_t h = fmix32((uint32_t)(ctr >> 32) ^ k);
    return fmix32((uint32_t)ctr ^ h);
}
// Unit-variance noise for a PAIR of taps from ONE hash (round 6; oracle/rir_synth.py::gauss is the definition): an Irwin-Hall sum of four uniform
// bytes per tap -- zero mean, unit variance, kurtosis 2.7, |g| <= 3.45 -- the bytes of a for the even tap, of remix(a) for the odd one.
//     g = fma(float(b0 + b1 + b2 + b3), 1 / sqrt(4 (256^2 - 1) / 12), -510 / sqrt(...))          remix(a) = h ^ (h >> 13),  h = (a >> 7)[23:0] * 0xB5297A + a
// Seven full-rate integer / float instructions per pair (v_sad_u8 sums the bytes) where rounds 3-5 ran a Box-Muller pair: a log2, a square root, a
// sine and a cosine -- four quarter-rate transcendentals -- per pair.  The generator is this repository's own definition (row R's parity is
// unpinned: the reference's RIRs come from closed-source RLR); whiteness and moments: tests/test_oracle_rir_stats.py.
__device__ __forceinline__ void noise_pair(uint32_t a, float& g0, float& g1) {
    constexpr float SC = 0.006765875034034252f;           // 1 / sqrt(21845)
    constexpr float OF = -3.450596332550049f;             // -510 * SC  (both rounded to float32 exactly as oracle/rir_synth.py: IH_SCALE, IH_OFFSET)
    uint32_t h = __umul24(a >> 7, 0xB5297Au) + a;
    h ^= h >> 13;
    g0 = __builtin_fmaf((float)__builtin_amdgcn_sad_u8(a, 0u, 0u), SC, OF);
    g1 = __builtin_fmaf((float)__builtin_amdgcn_sad_u8(h, 0u, 0u), SC, OF);
}

// FAST32: the bank has fewer than 2^33 samples, so the high word of every PAIR counter is 0 and the first mixing round of hash32 is one
// constant.  V consecutive taps per thread; V = 2 or 4 needs an even L (then every thread's first counter is even at every
// position and its taps are whole pairs); V = 1 evaluates its pair's hash per tap (odd L).  The chain over the positions is sequential,
// so the parallelism is C * L / V threads: two taps per thread give a config-2 bank 3 000 waves for the 1 024 SIMDs; the position loop is
// unrolled by two (two positions' hash / Box-Muller chains in flight per thread).
// peak_bits (may be null): max |bank| over the whole bank -- row G's abs().max() for free; slots: 1 + gridDim.x words of workspace.
constexpr int RIR_GEOM_LDS = 2048;     // (position, channel) pairs of geometry a workgroup keeps in LDS: P <= 1024 for the usual two channels per workgroup
template <bool FAST32, int V>
__device__ __forceinline__ void rir_synth_body(const RirDev& p, float* __restrict__ bank, unsigned int* __restrict__ peak_bits,
                                               unsigned int* __restrict__ slots, const unsigned bx, const unsigned gx) {
    const int64_t i = ((int64_t)bx * 256 + threadIdx.x) * V;
    const int64_t CL = (int64_t)p.C * p.L;
    const bool live = i < CL;
    const int64_t ii = live ? i : CL - V;
    const int c = (int)(ii / p.L);
    const int t0 = (int)(ii - (int64_t)c * p.L);
    float te[V], n[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { te[v] = p.tail_gain * (float)exp(-(double)(t0 + v) * p.inv_tau); n[v] = 0.0f; }
    float peak = 0.0f;
    uint64_t ctr = (uint64_t)c * (uint64_t)p.L + (uint64_t)t0;
    float* out = bank + ii;
    const int32_t* dl = p.delay + c;
    const float* dg = p.dgain + c;
    const uint32_t k1 = p.seed * 0x9E3779B9u + 1u;
    const uint32_t h1c = fmix32(k1);
    // The geometry of the workgroup's channel(s) goes to LDS once (round 6): a global load inside the position loop shares `vmcnt` with the stores, and
    // on this chip the counter retires in order -- waiting for the next position's delay meant waiting for the acknowledgement of every store issued
    // before it (~0.8 us each time: 420 ns per position and wave, the "latency" a bank of 200 positions ran at: 84 us where its stores alone take 43).
    __shared__ int32_t s_del[RIR_GEOM_LDS];
    __shared__ float s_dg[RIR_GEOM_LDS];
    const int64_t i_lo = (int64_t)bx * 256 * V, i_hi = (i_lo + 256 * V < CL ? i_lo + 256 * V : CL) - 1;
    const int c_lo = (int)(i_lo / p.L), nspan = (int)(i_hi / p.L) - c_lo + 1;
    const bool geom_lds = (int64_t)nspan * p.P <= RIR_GEOM_LDS;
    if (geom_lds) {
        for (int kk = threadIdx.x; kk < nspan * p.P; kk += 256) {
            const int cc = kk / p.P, q = kk - cc * p.P;
            s_del[kk] = p.delay[(int64_t)q * p.C + c_lo + cc];
            s_dg[kk] = p.dgain[(int64_t)q * p.C + c_lo + cc];
        }
        __syncthreads();
    }
    const int32_t* sd = s_del + (c - c_lo) * p.P;
    const float* sg = s_dg + (c - c_lo) * p.P;
    // Waves whose taps all lie behind every direct-path delay of their channel(s) -- at config-2 shapes all but the first wave or two of each channel --
    // need neither the delay gate nor the impulse test: their loop has no LDS read, no compare, no select (a fifth of its instructions; same values).
    bool ungated = false;
    if (geom_lds) {
        __shared__ int s_dmax[4];
        int dm = INT32_MIN;
        for (int kk = threadIdx.x; kk < nspan * p.P; kk += 256) dm = s_del[kk] > dm ? s_del[kk] : dm;
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(dm, o); dm = v > dm ? v : dm; }
        if ((threadIdx.x & 63) == 0) s_dmax[threadIdx.x >> 6] = dm;
        __syncthreads();
        dm = max(max(s_dmax[0], s_dmax[1]), max(s_dmax[2], s_dmax[3]));
        ungated = __all(t0 > dm) != 0;
    }
    int d_next = geom_lds ? sd[0] : dl[0];
    uint32_t pr32 = (uint32_t)(ctr >> 1);                            // FAST32, V >= 2: the pair counter is 32 bits and advances by C L / 2 per position
    const uint32_t pr_step = (uint32_t)(CL >> 1);
    auto step = [&](int q, auto lds_tag, auto gate_tag) {            // (two loop bodies per address space: one pointer of either would be a FLAT load -- vmcnt again)
        constexpr bool GL = decltype(lds_tag)::value, GATE = decltype(gate_tag)::value;
        int d = 0;
        if (GATE) {
            d = d_next;                                              // this position's direct-path delay was requested an iteration ago
            if (q + 1 < p.P) d_next = GL ? sd[q + 1] : dl[(int64_t)(q + 1) * p.C];
        }
        float g[V];
        if (V == 1) {
            const uint64_t pr = ctr >> 1;
            float g0, g1;
            noise_pair(FAST32 ? fmix32((uint32_t)pr ^ h1c) : fmix32((uint32_t)pr ^ fmix32((uint32_t)(pr >> 32) ^ k1)), g0, g1);
            g[0] = (ctr & 1) ? g1 : g0;
        } else {
#pragma unroll
            for (int v = 0; v < V; v += 2) {
                const uint64_t pr = (ctr + (uint64_t)v) >> 1;
                noise_pair(FAST32 ? fmix32((pr32 + (uint32_t)(v >> 1)) ^ h1c) : fmix32((uint32_t)pr ^ fmix32((uint32_t)(pr >> 32) ^ k1)), g[v], g[v + (V > 1 ? 1 : 0)]);
            }
        }
        float val[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            n[v] = (q == 0) ? g[v] : (p.rho * n[v] + p.srho * g[v]);
            const int t = t0 + v;
            val[v] = (!GATE || t > d) ? te[v] * n[v] : 0.0f;
        }
        if (GATE && (unsigned)(d - t0) < (unsigned)V) {              // the direct-path impulse falls on one of this thread's taps: one thread
            const float dgain = GL ? sg[q] : dg[(int64_t)q * p.C];   // per (position, channel) -- a rarely taken branch, not selects per tap
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (t0 + v == d) val[v] += dgain;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) peak = fmaxf(peak, fabsf(val[v]));
        if (live) {
            if (V == 4) *reinterpret_cast<float4*>(out) = make_float4(val[0], val[1], val[2], val[3]);
            else if (V == 2) *reinterpret_cast<float2*>(out) = make_float2(val[0], val[V - 1]);
            else
#pragma unroll
                for (int v = 0; v < V; ++v) out[v] = val[v];
        }
        out += CL;
        if (FAST32 && V >= 2) pr32 += pr_step;
        else ctr += (uint64_t)CL;
    };
    auto run = [&](auto lds_tag, auto gate_tag) {
        int q = 0;
        for (; q + 1 < p.P; q += 2) { step(q, lds_tag, gate_tag); step(q + 1, lds_tag, gate_tag); }
        for (; q < p.P; ++q) step(q, lds_tag, gate_tag);
    };
    if (ungated) run(std::true_type{}, std::false_type{});
    else if (geom_lds) run(std::true_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{});
    if (peak_bits) {
        // max |bank| without an initialised result word (a hipMemsetAsync ahead of the kernel is 4 us + a 6 us boundary, profiles/r03g):
        // every workgroup publishes its maximum in its own slot (write-through store), draws an arrival ticket, and the LAST arriver
        // reduces the slots, stores the result and resets the ticket for the next launch (slot 0 = ticket, zeroed when allocated).
        __shared__ float wmax[4];
        __shared__ int is_last;
        if (!live) peak = 0.0f;
        for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = peak;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            __hip_atomic_store(slots + 1 + bx, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(slots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = t == gx - 1;
        }
        __syncthreads();
        if (is_last) {
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            unsigned m = 0;
            for (unsigned b = threadIdx.x; b < gx; b += 256) {
                const unsigned v = __hip_atomic_load(slots + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m = v > m ? v : m;                 // non-negative floats order like their bit patterns
            }
            for (int o = 32; o > 0; o >>= 1) { const unsigned v = __shfl_xor(m, o); m = v > m ? v : m; }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = __uint_as_float(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                *peak_bits = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
                __hip_atomic_store(slots, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <bool FAST32, int V>
__global__ __launch_bounds__(256) void k_rir_synth(RirDev p, float* __restrict__ bank, unsigned int* __restrict__ peak_bits,
                                                   unsigned int* __restrict__ slots) {
    rir_synth_body<FAST32, V>(p, bank, peak_bits, slots, blockIdx.x, gridDim.x);
}

// the banks of ONE scene in one launch (SonicSet.py:61-63 + :86-91: three trajectories + two static positions): blockIdx.y = bank.  All
// banks share C * L (so one grid width fits them all); P, geometry, seed, decay, peak word and peak slots are per bank.
struct RirBatch {
    RirDev p[8];
    float* bank[8];
    unsigned int* peak[8];
    unsigned int* slots[8];
};
template <int V>
__global__ __launch_bounds__(256) void k_rir_synth_batch(RirBatch tab) {
    const int b = (int)blockIdx.y;
    rir_synth_body<true, V>(tab.p[b], tab.bank[b], tab.peak[b], tab.slots[b], blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// reductions / elementwise (rows G, M, U)
// 16-byte loads, 8 independent loads in flight per thread, at most 512 workgroups (grid sweep in profiles/r02f: 512 -> 41 us, 2048 -> 53 us,
// 8192 -> 60 us for 307 MB): a pure streaming max must run at the copy ceiling
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ out_bits) {
    unsigned int m = 0;
    const int64_t stride = (int64_t)gridDim.x * 256, tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t head = 0;
    if (((uintptr_t)x & 15) == 0) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* x4 = reinterpret_cast<const u32x4*>(x);
        const int64_t n4 = n >> 2;
        int64_t i = tid;
        for (; i + 7 * stride < n4; i += 8 * stride) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(x4 + i + u * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned int a = max(v[u].x & 0x7fffffffu, v[u].y & 0x7fffffffu), b = max(v[u].z & 0x7fffffffu, v[u].w & 0x7fffffffu);
                m = max(m, max(a, b));
            }
        }
        for (; i < n4; i += stride) {
            const u32x4 v = x4[i];
            m = max(m, max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        head = n4 << 2;
    }
    for (int64_t i = head + tid; i < n; i += stride) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    // one atomic per workgroup, and only when it can raise the maximum: atomics on ONE address serialise (~12 ns each; the
    // first version's 16 K wave-level atomics cost more than streaming the 307 MB)
    __shared__ unsigned int wmax[4];
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_bits, m);
    }
}

// x /= peak (IEEE division, bit-exact with the reference's elementwise true division, degenerate peaks included: an all-zero
// bank becomes NaN (0/0) and a NaN anywhere makes the peak -- the largest bit pattern -- and thus everything NaN, exactly like
// torch's ir_output /= ir_output.abs().max())
__global__ __launch_bounds__(256) void k_divide(float* __restrict__ x, int64_t n, const unsigned int* __restrict__ peak_bits) {
    const float peak = __uint_as_float(*peak_bits);
    const int64_t stride = (int64_t)gridDim.x * 256, tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t head = 0;
    if (((uintptr_t)x & 15) == 0) {
        float4* x4 = reinterpret_cast<float4*>(x);
        const int64_t n4 = n >> 2;
        int64_t i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u].x = v[u].x / peak; v[u].y = v[u].y / peak; v[u].z = v[u].z / peak; v[u].w = v[u].w / peak;
                x4[i + u * stride] = v[u];
            }
        }
        for (; i < n4; i += stride) {
            float4 v = x4[i];
            v.x = v.x / peak; v.y = v.y / peak; v.z = v.z / peak; v.w = v.w / peak;
            x4[i] = v;
        }
        head = n4 << 2;
    }
    for (int64_t i = head + tid; i < n; i += stride) x[i] = x[i] / peak;
}

// deterministic two-stage float64 sums: partial[a][blockIdx.x] then fixed-order final
template <int KIND>   // 0: sum x^2, 1: sum x
__global__ __launch_bounds__(256) void k_partial_sum(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double sw[4];
    const float* xa = x + (int64_t)blockIdx.y * n;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = (double)xa[i];
        s += KIND == 0 ? v * v : v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// the same with 16-byte loads (n % 4 == 0, 16-byte aligned rows): the 4-byte form streams at ~2 TB/s
template <int KIND>
__global__ __launch_bounds__(256) void k_partial_sum4(const float* __restrict__ x, int64_t n4, double* __restrict__ partial) {
    __shared__ double sw[4];
    const float4* xa = reinterpret_cast<const float4*>(x) + (int64_t)blockIdx.y * n4;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = xa[i];
        const double a = v.x, b = v.y, c = v.z, d = v.w;
        s += KIND == 0 ? (a * a + b * b) + (c * c + d * d) : (a + b) + (c + d);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// fixed-order final sum: lane l adds partial[l], partial[l + 64], ... then a butterfly over the 64 lanes (deterministic)
// SS_FLAG_RESULT_DEVICE: {loudness, gain, sum(out), sum(in)} of stem g straight into a caller's device array -- the final additions in
// k_final_sum's association (64 strided lanes, then a butterfly), so host-finished and device-finished sums are the same bits
__device__ __forceinline__ double wave_final_sum(const double* __restrict__ partial, int nb, int lane) {
    double s = 0.0;
    for (int i = lane; i < nb; i += 64) s += partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
__global__ __launch_bounds__(64) void k_lufs_result(const double* __restrict__ res, const double* __restrict__ part, int nb, double* __restrict__ out,
                                                    const double* __restrict__ part_sq = nullptr, double* __restrict__ sumsq = nullptr,
                                                    int S = 0, const double* __restrict__ part_x = nullptr) {
    const int g = blockIdx.x;
    if (part_x && g >= S) {          // blocks S ..: the cross sums of the speaker pairs, behind the S energies
        const double v = wave_final_sum(part_x + (int64_t)(g - S) * nb, nb, (int)threadIdx.x);
        if (threadIdx.x == 0) sumsq[g] = v;
        return;
    }
    double s0 = 0.0, s1 = 0.0;
    if (part_sq) {                   // sum(out^2) of stem g for the mix that follows (ss_mix_presum_f32)
        const double v = wave_final_sum(part_sq + (int64_t)g * nb, nb, (int)threadIdx.x);
        if (threadIdx.x == 0) sumsq[g] = v;
    }
    for (int i = threadIdx.x; i < nb; i += 64) {
        s0 += part[((int64_t)2 * g + 0) * nb + i];
        s1 += part[((int64_t)2 * g + 1) * nb + i];
    }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o); s1 += __shfl_down(s1, o); }
    if (threadIdx.x == 0) {
        out[4 * g + 0] = res[4 * g + 0];
        out[4 * g + 1] = res[4 * g + 1];
        out[4 * g + 2] = s0;
        out[4 * g + 3] = s1;
    }
}

__global__ __launch_bounds__(64) void k_final_sum(const double* __restrict__ partial, int nb, double* __restrict__ out) {
    DBG_CLK(3, 0);
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[(int64_t)blockIdx.x * nb + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
    DBG_CLK(3, 1);
}
// out = gain * in with the float64 partial sums of out and in in the same pass (pyloudnorm.normalize.loudness + :78-79).
// gain_dev != nullptr: the gain is the float64 the gating kernel left on the device (rounded to float32 like the host path).
// blockIdx.y = group (stem): its n elements start at group * n, its gain is gain_dev[4 * group], its partial sums go to
// partial[(2 * group + {0,1}) * gridDim.x + blockIdx.x].
// nspk > 1 (round 6, ss_lufs_norm_batch_sqx_f32): the first nspk stems are the speakers of the mix that follows; the workgroups of speaker j < nspk also read
// the inputs of the speakers i < j at the same indices, form out_i = gain_i * in_i as that stem's own workgroups do, and leave the partial sums of
// out_i * out_j in partial_x[pair][grid] (pairs in the order (0,1), (0,2), (1,2), (0,3) ...): with them the energy of the speech sum is known without a pass
// over it and the mix is ONE pass (ss_mix_onepass_f32).
constexpr int SCALE_XMAX = 3;        // speakers i < j a workgroup reads beside its own stem: nspk <= 4
__global__ __launch_bounds__(256) void k_scale_sums(const float* __restrict__ in, float* __restrict__ out, int64_t n, float gain,
                                                    const double* __restrict__ gain_dev, double* __restrict__ partial /*[groups][2][grid]*/,
                                                    double* __restrict__ partial_sq = nullptr /*[groups][grid]: sum(out^2), or null*/,
                                                    int nspk = 0, double* __restrict__ partial_x = nullptr /*[pairs][grid]*/) {
    __shared__ double sw[3][4];
    double sq = 0.0;                 // round 5: the energy the mix needs of every normalised stem rides on the pass that writes it (ss_mix_presum_f32)
    if (gain_dev) gain = (float)gain_dev[4 * blockIdx.y];
    if (partial_x && (int)blockIdx.y >= 1 && (int)blockIdx.y < nspk && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && (n & 3) == 0) {
        // speaker j >= 1: the same pass with the cross sums (aligned stems of whole float4s: the mix's own requirement)
        const int j = (int)blockIdx.y;
        float gi[SCALE_XMAX];
        const float4* oth[SCALE_XMAX];
        double cx[SCALE_XMAX];
#pragma unroll
        for (int i = 0; i < SCALE_XMAX; ++i) {
            gi[i] = i < j ? (float)gain_dev[4 * i] : 0.0f;
            oth[i] = (const float4*)(in + (int64_t)(i < j ? i : 0) * n);
            cx[i] = 0.0;
        }
        const float4* in4 = (const float4*)(in + (int64_t)j * n);
        float4* out4 = (float4*)(out + (int64_t)j * n);
        double so = 0.0, si = 0.0;
        const int64_t stride = (int64_t)gridDim.x * 256, n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 v = in4[i];
            float4 o;
            o.x = gain * v.x; o.y = gain * v.y; o.z = gain * v.z; o.w = gain * v.w;
            out4[i] = o;
            so += ((double)o.x + (double)o.y) + ((double)o.z + (double)o.w);
            si += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            sq += ((double)o.x * (double)o.x + (double)o.y * (double)o.y) + ((double)o.z * (double)o.z + (double)o.w * (double)o.w);
#pragma unroll
            for (int k = 0; k < SCALE_XMAX; ++k)
                if (k < j) {
                    const float4 u = oth[k][i];
                    const float ux = gi[k] * u.x, uy = gi[k] * u.y, uz = gi[k] * u.z, uw = gi[k] * u.w;
                    cx[k] += ((double)ux * (double)o.x + (double)uy * (double)o.y) + ((double)uz * (double)o.z + (double)uw * (double)o.w);
                }
        }
        __shared__ double swx[SCALE_XMAX][4];
        for (int o = 32; o > 0; o >>= 1) {
            so += __shfl_xor(so, o); si += __shfl_xor(si, o); sq += __shfl_xor(sq, o);
#pragma unroll
            for (int k = 0; k < SCALE_XMAX; ++k) cx[k] += __shfl_xor(cx[k], o);
        }
        if ((threadIdx.x & 63) == 0) {
            sw[0][threadIdx.x >> 6] = so; sw[1][threadIdx.x >> 6] = si; sw[2][threadIdx.x >> 6] = sq;
#pragma unroll
            for (int k = 0; k < SCALE_XMAX; ++k) swx[k][threadIdx.x >> 6] = cx[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* pj = partial + (int64_t)2 * j * gridDim.x;
            pj[blockIdx.x] = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
            pj[gridDim.x + blockIdx.x] = (sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]);
            if (partial_sq) partial_sq[(int64_t)j * gridDim.x + blockIdx.x] = (sw[2][0] + sw[2][1]) + (sw[2][2] + sw[2][3]);
            for (int k = 0; k < j && k < SCALE_XMAX; ++k)
                partial_x[((int64_t)(j * (j - 1) / 2 + k)) * gridDim.x + blockIdx.x] = (swx[k][0] + swx[k][1]) + (swx[k][2] + swx[k][3]);
        }
        return;
    }
    in += (int64_t)blockIdx.y * n;
    out += (int64_t)blockIdx.y * n;
    partial += (int64_t)2 * blockIdx.y * gridDim.x;
    double so = 0.0, si = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256, tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const float4* in4 = (const float4*)in;
        float4* out4 = (float4*)out;
        for (int64_t i = tid; i < n4; i += stride) {
            const float4 v = in4[i];
            float4 o;
            o.x = gain * v.x; o.y = gain * v.y; o.z = gain * v.z; o.w = gain * v.w;
            out4[i] = o;
            so += ((double)o.x + (double)o.y) + ((double)o.z + (double)o.w);
            si += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            sq += ((double)o.x * (double)o.x + (double)o.y * (double)o.y) + ((double)o.z * (double)o.z + (double)o.w * (double)o.w);
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
            const float v = in[i], o = gain * v;
            out[i] = o;
            so += (double)o;
            si += (double)v;
            sq += (double)o * (double)o;
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float v = in[i];
            const float o = gain * v;
            out[i] = o;
            so += (double)o;
            si += (double)v;
            sq += (double)o * (double)o;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { so += __shfl_xor(so, o); si += __shfl_xor(si, o); sq += __shfl_xor(sq, o); }
    if ((threadIdx.x & 63) == 0) { sw[0][threadIdx.x >> 6] = so; sw[1][threadIdx.x >> 6] = si; sw[2][threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
        partial[gridDim.x + blockIdx.x] = (sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]);
        if (partial_sq) partial_sq[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sw[2][0] + sw[2][1]) + (sw[2][2] + sw[2][3]);
    }
}

__device__ __forceinline__ double rms_db_from_sumsq(double ss_, double n) {
    const double ms = ss_ / n;
    return 10.0 * log10(ms > 1e-20 ? ms : 1e-20);
}

// one wave adds nb partial sums in k_final_sum's association (64 strided lanes, then a butterfly); every lane gets the sum
// mix step 2: interferer gains from speaker energies (movingdatamodule.py:106-113); finishes the energy sums itself
// (partial[S][nb], wave per speaker) instead of a separate final-sum launch
struct SirTab { float v[64]; };        // the S - 1 drawn SIRs travel in the kernel arguments (an upload costs a 4 us copy + a 6 us boundary)
__global__ __launch_bounds__(1024) void k_mix_gains1(const double* __restrict__ partial, int nb, int S, double n, const SirTab sir_tab,
                                                     float* __restrict__ g, double* __restrict__ sumsq_out) {
    const float* sirs = sir_tab.v;
    __shared__ double sumsq[64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int sp = w; sp < S; sp += 16) {
        const double v = wave_final_sum(partial + (int64_t)sp * nb, nb, lane);
        if (lane == 0) { sumsq[sp] = v; sumsq_out[sp] = v; }
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i == 0) g[0] = 1.0f;
    if (i >= 1 && i < S) {
        const double e0 = rms_db_from_sumsq(sumsq[0], n), ei = rms_db_from_sumsq(sumsq[i], n);
        double gain = e0 - ei - (double)sirs[i - 1];
        gain = gain < 40.0 ? gain : 40.0;
        g[i] = (float)pow(10.0, gain / 20.0);
    }
}
// mix step 3: scale interferers in place, speech sum -> mix, accumulate energies of speech / noise sums
__global__ __launch_bounds__(256) void k_mix_scale_sum(float* __restrict__ spk, int S, const float* __restrict__ noises, int N,
                                                       int64_t n, const float* __restrict__ g, float* __restrict__ mix,
                                                       double* __restrict__ partial /*[2][grid]*/, int write_back) {
    __shared__ double sw[2][4];
    double e_s = 0.0, e_n = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float sp = spk[i];
        for (int s = 1; s < S; ++s) {
            const float v = spk[(int64_t)s * n + i] * g[s];
            if (write_back) spk[(int64_t)s * n + i] = v;
            sp += v;
        }
        float nz = 0.0f;
        for (int k = 0; k < N; ++k) nz += noises[(int64_t)k * n + i];
        mix[i] = sp;
        e_s += (double)sp * (double)sp;
        e_n += (double)nz * (double)nz;
    }
    for (int o = 32; o > 0; o >>= 1) { e_s += __shfl_xor(e_s, o); e_n += __shfl_xor(e_n, o); }
    if ((threadIdx.x & 63) == 0) { sw[0][threadIdx.x >> 6] = e_s; sw[1][threadIdx.x >> 6] = e_n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
        partial[gridDim.x + blockIdx.x] = (sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]);
    }
}
// the same pass with 16-byte accesses (n % 4 == 0, 16-byte aligned stems): the scalar form runs at 2.4 TB/s, this one at the copy rate
__global__ __launch_bounds__(256) void k_mix_scale_sum4(float* __restrict__ spk, int S, const float* __restrict__ noises, int N,
                                                        int64_t n4, const float* __restrict__ g, float* __restrict__ mix,
                                                        double* __restrict__ partial /*[2][grid]*/, int write_back) {
    __shared__ double sw[2][4];
    double e_s = 0.0, e_n = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* spk4 = reinterpret_cast<float4*>(spk);
    const float4* noi4 = reinterpret_cast<const float4*>(noises);
    float4* mix4 = reinterpret_cast<float4*>(mix);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 sp = spk4[i];
        for (int s = 1; s < S; ++s) {
            const float gs = g[s];
            float4 v = spk4[(int64_t)s * n4 + i];
            v.x *= gs; v.y *= gs; v.z *= gs; v.w *= gs;
            if (write_back) spk4[(int64_t)s * n4 + i] = v;
            sp.x += v.x; sp.y += v.y; sp.z += v.z; sp.w += v.w;
        }
        float4 nz = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int k = 0; k < N; ++k) {
            const float4 v = noi4[(int64_t)k * n4 + i];
            nz.x += v.x; nz.y += v.y; nz.z += v.z; nz.w += v.w;
        }
        mix4[i] = sp;
        e_s += ((double)sp.x * (double)sp.x + (double)sp.y * (double)sp.y) + ((double)sp.z * (double)sp.z + (double)sp.w * (double)sp.w);
        e_n += ((double)nz.x * (double)nz.x + (double)nz.y * (double)nz.y) + ((double)nz.z * (double)nz.z + (double)nz.w * (double)nz.w);
    }
    for (int o = 32; o > 0; o >>= 1) { e_s += __shfl_xor(e_s, o); e_n += __shfl_xor(e_n, o); }
    if ((threadIdx.x & 63) == 0) { sw[0][threadIdx.x >> 6] = e_s; sw[1][threadIdx.x >> 6] = e_n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
        partial[gridDim.x + blockIdx.x] = (sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]);
    }
}

__global__ __launch_bounds__(128) void k_mix_gains2(const double* __restrict__ partial /*[2][nb]: speech, noise*/, int nb, double n, float snr,
                                                    float* __restrict__ g, int S) {
    __shared__ double sums[2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double v = wave_final_sum(partial + (int64_t)w * nb, nb, lane);
    if (lane == 0) sums[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double gain = rms_db_from_sumsq(sums[0], n) - rms_db_from_sumsq(sums[1], n) - (double)snr;
        gain = gain < 40.0 ? gain : 40.0;
        g[S] = (float)pow(10.0, gain / 20.0);
    }
}
// ---- round 5: the mix of a scene whose stems come straight out of the loudness pass (ss_mix_presum_f32).  sum(x^2) of every stem was
// accumulated by the pass that wrote it (k_scale_sums), so the energy pass over the speakers and both one-workgroup gain kernels disappear:
// every workgroup derives the interferer gains from the S sums itself, and the second pass reduces the first one's partial sums itself.
// Same arithmetic per sample as k_mix_scale_sum4 / k_mix_final (movingdatamodule.py:105-124); two launches instead of five.
__global__ __launch_bounds__(256) void k_mix_pre_scale_sum4(float* __restrict__ spk, int S, int64_t n4, const double* __restrict__ sumsq,
                                                            double n_elems, const SirTab sir_tab, float* __restrict__ mix,
                                                            double* __restrict__ partial /*[grid]*/, float* __restrict__ g_out, int write_back) {
    __shared__ double sw[4];
    __shared__ float gs_[64];
    if ((int)threadIdx.x < S) {
        float g = 1.0f;
        if (threadIdx.x >= 1) {
            const double e0 = rms_db_from_sumsq(sumsq[0], n_elems), ei = rms_db_from_sumsq(sumsq[threadIdx.x], n_elems);
            double gain = e0 - ei - (double)sir_tab.v[threadIdx.x - 1];
            gain = gain < 40.0 ? gain : 40.0;
            g = (float)pow(10.0, gain / 20.0);
        }
        gs_[threadIdx.x] = g;
        if (blockIdx.x == 0) g_out[threadIdx.x] = g;
    }
    __syncthreads();
    double e_s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* spk4 = reinterpret_cast<float4*>(spk);
    float4* mix4 = reinterpret_cast<float4*>(mix);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 sp = spk4[i];
        for (int s = 1; s < S; ++s) {
            const float gs = gs_[s];
            float4 v = spk4[(int64_t)s * n4 + i];
            v.x *= gs; v.y *= gs; v.z *= gs; v.w *= gs;
            if (write_back) spk4[(int64_t)s * n4 + i] = v;
            sp.x += v.x; sp.y += v.y; sp.z += v.z; sp.w += v.w;
        }
        mix4[i] = sp;
        e_s += ((double)sp.x * (double)sp.x + (double)sp.y * (double)sp.y) + ((double)sp.z * (double)sp.z + (double)sp.w * (double)sp.w);
    }
    for (int o = 32; o > 0; o >>= 1) e_s += __shfl_xor(e_s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = e_s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void k_mix_pre_final4(const float* __restrict__ noise, int64_t n4, const double* __restrict__ partial, int nb,
                                                        double n_elems, const double* __restrict__ sumsq_noise, float snr,
                                                        float* __restrict__ mix, float* __restrict__ g_out /*[S + 1]*/, int S) {
    __shared__ float gn_;
    if (threadIdx.x < 64) {
        const double es = wave_final_sum(partial, nb, (int)threadIdx.x);
        if (threadIdx.x == 0) {
            double gain = rms_db_from_sumsq(es, n_elems) - rms_db_from_sumsq(sumsq_noise[0], n_elems) - (double)snr;
            gain = gain < 40.0 ? gain : 40.0;
            gn_ = (float)pow(10.0, gain / 20.0);
            if (blockIdx.x == 0) g_out[S] = gn_;
        }
    }
    __syncthreads();
    const float gn = gn_;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float4* noi4 = reinterpret_cast<const float4*>(noise);
    float4* mix4 = reinterpret_cast<float4*>(mix);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 nz = noi4[i];
        float4 m = mix4[i];
        m.x = m.x + (0.0f + nz.x) * gn; m.y = m.y + (0.0f + nz.y) * gn; m.z = m.z + (0.0f + nz.z) * gn; m.w = m.w + (0.0f + nz.w) * gn;
        mix4[i] = m;
    }
}

// round 6: the mix in ONE pass (ss_mix_onepass_f32).  With the speakers' energies E_ss and cross sums C_st known (k_scale_sums) the energy of the speech
// sum sp = s_0 + sum g_s s_s is E = sum_s g_s^2 E_ss + 2 sum_{s<t} g_s g_t C_st (g_0 = 1) -- float64, from the float64 sums; it differs from the sum over
// the float32-rounded sp by ~1e-10 relative (the roundings of sp are uncorrelated), so the noise gain is the two-pass form's up to a rare last bit.
// Per sample the same float32 operations as k_mix_pre_scale_sum4 + k_mix_pre_final4; 123 MB instead of 184 MB for a 2-speaker 8 x 960 000 mix.
__global__ __launch_bounds__(256) void k_mix_onepass4(float* __restrict__ spk, int S, const float* __restrict__ noise, int64_t n4,
                                                      const double* __restrict__ sums /*[S]*/, const double* __restrict__ cross /*[S (S - 1) / 2]*/,
                                                      const double* __restrict__ sumsq_noise,
                                                      double n_elems, const SirTab sir_tab, float snr, float* __restrict__ mix,
                                                      float* __restrict__ g_out /*[S + 1]*/, int write_back) {
    __shared__ float gs_[64];
    __shared__ float gn_;
    if ((int)threadIdx.x < S) {
        float g = 1.0f;
        if (threadIdx.x >= 1) {
            const double e0 = rms_db_from_sumsq(sums[0], n_elems), ei = rms_db_from_sumsq(sums[threadIdx.x], n_elems);
            double gain = e0 - ei - (double)sir_tab.v[threadIdx.x - 1];
            gain = gain < 40.0 ? gain : 40.0;
            g = (float)pow(10.0, gain / 20.0);
        }
        gs_[threadIdx.x] = g;
        if (blockIdx.x == 0) g_out[threadIdx.x] = g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double es = 0.0;
        for (int a = 0; a < S; ++a) es += (double)gs_[a] * (double)gs_[a] * sums[a];
        for (int b = 1; b < S; ++b)
            for (int a = 0; a < b; ++a) es += 2.0 * (double)gs_[a] * (double)gs_[b] * cross[b * (b - 1) / 2 + a];
        double gain = rms_db_from_sumsq(es, n_elems) - rms_db_from_sumsq(sumsq_noise[0], n_elems) - (double)snr;
        gain = gain < 40.0 ? gain : 40.0;
        gn_ = (float)pow(10.0, gain / 20.0);
        if (blockIdx.x == 0) g_out[S] = gn_;
    }
    __syncthreads();
    const float gn = gn_;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* spk4 = reinterpret_cast<float4*>(spk);
    const float4* noi4 = reinterpret_cast<const float4*>(noise);
    float4* mix4 = reinterpret_cast<float4*>(mix);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 sp = spk4[i];
        const float4 nz = noi4[i];
        for (int s = 1; s < S; ++s) {
            const float gs = gs_[s];
            float4 v = spk4[(int64_t)s * n4 + i];
            v.x *= gs; v.y *= gs; v.z *= gs; v.w *= gs;
            if (write_back) spk4[(int64_t)s * n4 + i] = v;
            sp.x += v.x; sp.y += v.y; sp.z += v.z; sp.w += v.w;
        }
        sp.x = sp.x + (0.0f + nz.x) * gn; sp.y = sp.y + (0.0f + nz.y) * gn; sp.z = sp.z + (0.0f + nz.z) * gn; sp.w = sp.w + (0.0f + nz.w) * gn;
        mix4[i] = sp;
    }
}

__global__ __launch_bounds__(256) void k_mix_final(const float* __restrict__ noises, int N, int64_t n, const float* __restrict__ g,
                                                   int S, float* __restrict__ mix) {
    const float gn = g[S];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float nz = 0.0f;
        for (int k = 0; k < N; ++k) nz += noises[(int64_t)k * n + i];
        const float scaled = nz * gn;
        mix[i] = mix[i] + scaled;
    }
}

__global__ __launch_bounds__(256) void k_div_by(const float* __restrict__ in, float* __restrict__ out, int64_t n, const float* __restrict__ div) {
    const float r = 1.0f / *div;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * r;
}
__global__ __launch_bounds__(256) void k_scale(const float* __restrict__ in, float* __restrict__ out, int64_t n, float gain) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = gain * in[i];
}

// ---- row N2: dataset-side crop + silence rejection + batched SIR/SNR mix (separation/look2hear/datas/movingdatamodule.py:81-126)
// mono[t] = (x[0][t] + x[1][t] + ... ) / C  -- torch's wav.mean(dim=0) (:63, :77): float32 sum in channel order, IEEE division
__global__ __launch_bounds__(256) void k_mean_channels(const float* __restrict__ x, int C, int64_t T, float* __restrict__ out) {
    const float fc = (float)C;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += stride) {
        float s = x[t];
        for (int c = 1; c < C; ++c) s += x[(int64_t)c * T + t];
        out[t] = s / fc;
    }
}
// sum of squares of K crops: crop k = C rows of n samples, row c at ptr[k] + c * chan_stride.  One workgroup per crop
// (a 4 s mono crop is 256 KB): fixed association -> deterministic.
__global__ __launch_bounds__(1024) void k_crop_sumsq(const float* const* __restrict__ ptr, int C, int64_t chan_stride, int64_t n,
                                                     double* __restrict__ out) {
    __shared__ double sw[16];
    const float* base = ptr[blockIdx.x];
    double s = 0.0;
    for (int c = 0; c < C; ++c) {
        const float* row = base + (int64_t)c * chan_stride;
        for (int64_t i = threadIdx.x; i < n; i += 1024) { const double v = (double)row[i]; s += v * v; }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sw[i];
        out[blockIdx.x] = t;
    }
}
// batched mix, step 1: interferer gains of item b from the speaker energies (movingdatamodule.py:106-113); one workgroup per item
__global__ __launch_bounds__(64) void k_bmix_gains1(const double* __restrict__ sumsq /*[B][S]*/, int S, double cnt, const float* __restrict__ sirs /*[B][S-1]*/,
                                                    float* __restrict__ g /*[B][S+1]*/) {
    const int b = blockIdx.x, i = threadIdx.x;
    if (i == 0) g[(int64_t)b * (S + 1)] = 1.0f;
    if (i >= 1 && i < S) {
        const double e0 = rms_db_from_sumsq(sumsq[(int64_t)b * S], cnt), ei = rms_db_from_sumsq(sumsq[(int64_t)b * S + i], cnt);
        double gain = e0 - ei - (double)sirs[(int64_t)b * (S - 1) + i - 1];
        gain = gain < 40.0 ? gain : 40.0;
        g[(int64_t)b * (S + 1) + i] = (float)pow(10.0, gain / 20.0);
    }
}
// step 2: scaled speaker crops -> spk_out[b][s][c][i], speech sum -> mix[b][c][i], energies of the speech and noise sums.
// grid (chunks, B); partial[b][2][chunks]
__global__ __launch_bounds__(256) void k_bmix_scale_sum(const float* const* __restrict__ sp /*[B][S]*/, const float* const* __restrict__ np /*[B][N]*/, int S, int N,
                                                        int C, int64_t chan_stride, int64_t n, const float* __restrict__ g, float* __restrict__ spk_out,
                                                        float* __restrict__ mix, double* __restrict__ partial) {
    __shared__ double sw[2][4];
    const int b = blockIdx.y;
    const int64_t cn = (int64_t)C * n;
    double e_s = 0.0, e_n = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cn; i += stride) {
        const int64_t c = i / n, t = i - c * n, src = c * chan_stride + t;
        float acc = sp[(int64_t)b * S][src];
        spk_out[((int64_t)b * S) * cn + i] = acc;
        for (int s2 = 1; s2 < S; ++s2) {
            const float v = sp[(int64_t)b * S + s2][src] * g[(int64_t)b * (S + 1) + s2];
            spk_out[((int64_t)b * S + s2) * cn + i] = v;
            acc += v;
        }
        float nz = 0.0f;
        for (int k = 0; k < N; ++k) nz += np[(int64_t)b * N + k][src];
        mix[(int64_t)b * cn + i] = acc;
        e_s += (double)acc * (double)acc;
        e_n += (double)nz * (double)nz;
    }
    for (int o = 32; o > 0; o >>= 1) { e_s += __shfl_xor(e_s, o); e_n += __shfl_xor(e_n, o); }
    if ((threadIdx.x & 63) == 0) { sw[0][threadIdx.x >> 6] = e_s; sw[1][threadIdx.x >> 6] = e_n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((int64_t)b * 2) * gridDim.x + blockIdx.x] = (sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]);
        partial[((int64_t)b * 2 + 1) * gridDim.x + blockIdx.x] = (sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]);
    }
}
// step 3: noise gain of item b (movingdatamodule.py:118-122); one workgroup of two waves per item
__global__ __launch_bounds__(128) void k_bmix_gains2(const double* __restrict__ partial, int nb, double cnt, const float* __restrict__ snrs, float* __restrict__ g, int S) {
    __shared__ double sums[2];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double v = wave_final_sum(partial + ((int64_t)b * 2 + w) * nb, nb, lane);
    if (lane == 0) sums[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double gain = rms_db_from_sumsq(sums[0], cnt) - rms_db_from_sumsq(sums[1], cnt) - (double)snrs[b];
        gain = gain < 40.0 ? gain : 40.0;
        g[(int64_t)b * (S + 1) + S] = (float)pow(10.0, gain / 20.0);
    }
}
// step 4: mix += g_n * (sum of the noise crops)
__global__ __launch_bounds__(256) void k_bmix_final(const float* const* __restrict__ np, int N, int C, int64_t chan_stride, int64_t n,
                                                    const float* __restrict__ g, int S, float* __restrict__ mix) {
    const int b = blockIdx.y;
    const int64_t cn = (int64_t)C * n;
    const float gn = g[(int64_t)b * (S + 1) + S];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cn; i += stride) {
        const int64_t c = i / n, t = i - c * n, src = c * chan_stride + t;
        float nz = 0.0f;
        for (int k = 0; k < N; ++k) nz += np[(int64_t)b * N + k][src];
        const float scaled = nz * gn;
        mix[(int64_t)b * cn + i] += scaled;
    }
}
// enhancement/look2hear/datas/movingdatamodule.py:34-48 overlap_audio: (x delayed by d) + (x advanced by d) + x, zero filled
__global__ __launch_bounds__(256) void k_overlap_audio(const float* __restrict__ x, float* __restrict__ out, int64_t T, int64_t d) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += stride) {
        const float fwd = t >= d ? x[t - d] : 0.0f;
        const float bwd = t + d < T ? x[t + d] : 0.0f;
        out[t] = (fwd + bwd) + x[t];
    }
}

// ---- row N3: polyphase sinc resampler (torchaudio.transforms.Resample as called at SonicSim_audio.py:249,297).
// out[row][f * nnew + p] = sum_j taps[j][p] * x[row][f * orig - width + first[p] + j]  (zeros outside [0, L)): the dense
// (2 width + orig)-tap convolution of the published algorithm restricted to each phase's non-zero taps (the Hann window clamps the
// rest to exactly 0).  taps is stored tap-major so that the 64 lanes (consecutive phases) read it coalesced; the input window of a
// frame (orig + 2 width samples) is shared by its nnew outputs and stays in L1.
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ x, int64_t L, const float* __restrict__ taps, const int32_t* __restrict__ first,
                                                  int ntap, int orig, int nnew, int width, float* __restrict__ out, int64_t Lout) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= Lout) return;
    const float* xr = x + (int64_t)blockIdx.y * L;
    const int64_t f = n / nnew;
    const int p = (int)(n - f * nnew);
    const int64_t i0 = f * orig - width + first[p];
    float acc = 0.0f;
    for (int j = 0; j < ntap; ++j) {
        const int64_t i = i0 + j;
        const float v = (i >= 0 && i < L) ? xr[i] : 0.0f;
        acc = fmaf(taps[(int64_t)j * nnew + p], v, acc);
    }
    out[(int64_t)blockIdx.y * Lout + n] = acc;
}

// ---- K-weighting (row U): two cascaded biquads, transposed direct form II (scipy.signal.lfilter),
// float64 state.  Chunk-parallel: (1) zero-state run per chunk -> end state, (2) sequential state
// propagation across chunks with the 4x4 chunk transition matrix, (3) re-run from the true state.
// Row U's two biquads in DELTA form (round 6).  With a1 = -2 + d1, a2 = 1 - d2 the direct-form-II recurrence w[n] = x[n] - a1 w[n-1] - a2 w[n-2]
// becomes, in the states w1 = w[n-1] and d = w[n-1] - w[n-2],
//     d' = x + d - d2 d - eps w1,   w1' = w1 + d',   y = beta w1 + b0 d' - b2 d        (eps = A(1) = 1 + a1 + a2, beta = B(1) = b0 + b1 + b2)
// i.e. the SMALL quantities (K-weighting's high-pass has eps = 2.5e-5, d2 = 0.01 at 48 kHz) are the coefficients, each held to full relative
// precision.  In float64 this equals SciPy's lfilter to 1e-14 in the loudness; in FLOAT32 it stays within 4e-6 dB of it over noise, tones down to
// 50 Hz, brown noise and a DC offset of 0.3 at 8 / 16 / 44.1 / 48 kHz (2e-7 dB at 16 kHz), where the textbook forms with float32-rounded
// a1, a2 are off by up to 5e-4 in |H|^2 below 100 Hz (tools/lab/r06_kw_f32.py).  7 operations per stage, a dependent chain of 3.
template <class R>
struct KwTabT {
    R b[2][3], a[2][3];   // the normalised biquads as given (host-side identity of the table set)
    R c[2][6];            // per stage: d2, eps, beta, b0, b2, (pad)
    R Mp[13][16];         // Mp[i] = (state transition over one full chunk) ^ (2^i), row-major 4x4; state = (w1, d) of stage 1, (w1, d) of stage 2
    R W[64][4];           // zero-state END state of a full chunk as a linear map of its samples: z = sum_t W[t] * x[t]
};
using KwCoef = KwTabT<double>;
using KwCoefF = KwTabT<float>;
static_assert(sizeof(((KwCoef*)0)->W) == 64 * 4 * 8, "W is [KW_CHUNK][4]");
constexpr int KW_CHUNK = 64;     // samples per thread in the sample-level passes (divides the 0.1 s block step at 16 kHz)
constexpr int KW_ROW = KW_CHUNK + 4;   // floats per LDS row of the fused kernel: 16-lane groups of b128 accesses hit distinct banks
constexpr int KW_SER = 8;        // chunks each scan thread walks serially
constexpr int KW_TILE = 512;     // threads of the scan workgroup: one tile = KW_TILE * KW_SER chunks
static_assert(KW_SER == 8 && KW_TILE == 512, "the scan's power indices (Mp[3+b], Mp[9+b], Mp[12]) assume 8 chunks/thread, 8 waves");

template <class R>
__host__ __device__ __forceinline__ R kw_stage(const R* __restrict__ c, R& w1, R& d, R x) {
    R t = x + d;
    t = fma(-c[1], w1, t);
    const R dn = fma(-c[0], d, t);
    R y = c[2] * w1;
    y = fma(c[3], dn, y);
    y = fma(-c[4], d, y);
    w1 += dn;
    d = dn;
    return y;
}
template <class R>
__host__ __device__ __forceinline__ R kw_step(const KwTabT<R>& k, R s[4], R xin) {
    const R y1 = kw_stage<R>(k.c[0], s[0], s[1], xin);
    return kw_stage<R>(k.c[1], s[2], s[3], y1);
}
template <class R>
__device__ __forceinline__ void kw_matvec(const R* M, const R s[4], R o[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = M[r * 4 + 0] * s[0] + M[r * 4 + 1] * s[1] + M[r * 4 + 2] * s[2] + M[r * 4 + 3] * s[3];
}
// visit the samples of one chunk in order; 16-byte loads when the chunk is whole, contiguous and aligned
template <class F>
__device__ __forceinline__ void kw_walk(const float* __restrict__ a, int64_t t0, int64_t t1, int64_t st, F&& f) {
    if (st == 1 && t1 - t0 == KW_CHUNK && (((uintptr_t)(a + t0)) & 15) == 0) {
        const float4* p = (const float4*)(a + t0);
        float4 q[KW_CHUNK / 4];                         // the whole chunk in flight at once: one memory latency, not four
#pragma unroll
        for (int i = 0; i < KW_CHUNK / 4; ++i) q[i] = p[i];
#pragma unroll
        for (int i = 0; i < KW_CHUNK / 4; ++i) { f(q[i].x); f(q[i].y); f(q[i].z); f(q[i].w); }
    } else {
#pragma unroll 8
        for (int64_t t = t0; t < t1; ++t) f(a[t * st]);
    }
}

// zero-state end state of one chunk by its FIR form: 4 independent dot products (256 FMAs, no recurrence) instead of 64
// dependent biquad steps (~700 float64 ops).  A short last chunk of n samples uses the rows W[64-n..63].
template <class R>
__device__ __forceinline__ void kw_endstate(const float* __restrict__ a, int64_t t0, int64_t t1, int64_t st, const KwTabT<R>& k, R v[4]) {
    if (st == 1 && t1 - t0 == KW_CHUNK && (((uintptr_t)(a + t0)) & 15) == 0) {
        const float4* p = (const float4*)(a + t0);
        float4 q[KW_CHUNK / 4];
#pragma unroll
        for (int i = 0; i < KW_CHUNK / 4; ++i) q[i] = p[i];
#pragma unroll
        for (int i = 0; i < KW_CHUNK / 4; ++i) {
            const R x0 = (R)q[i].x, x1 = (R)q[i].y, x2 = (R)q[i].z, x3 = (R)q[i].w;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = fma(k.W[4 * i + 3][r], x3, fma(k.W[4 * i + 2][r], x2, fma(k.W[4 * i + 1][r], x1, fma(k.W[4 * i][r], x0, v[r]))));
        }
    } else {
        const int off = KW_CHUNK - (int)(t1 - t0);
        for (int64_t t = t0; t < t1; ++t) {
            const R x = (R)a[t * st];
            const R* w = k.W[off + (int)(t - t0)];
            v[0] += w[0] * x; v[1] += w[1] * x; v[2] += w[2] * x; v[3] += w[3] * x;
        }
    }
}
// pass 1: zero-state response of every chunk -> its end state z
__global__ __launch_bounds__(256) void k_kw_state(const float* __restrict__ audio, int64_t T, int C, int64_t st, int64_t sc,
                                                  const KwCoef* __restrict__ kp, int nchunks, double* __restrict__ states /*[C][nchunks][4]*/) {
    const KwCoef& k = *kp;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * nchunks) return;
    const int c = id / nchunks, ch = id - c * nchunks;
    double s[4] = {0, 0, 0, 0};
    const int64_t t0 = (int64_t)ch * KW_CHUNK;
    const int64_t t1 = t0 + KW_CHUNK < T ? t0 + KW_CHUNK : T;
    kw_endstate(audio + c * sc, t0, t1, st, k, s);
    double* o = states + ((int64_t)c * nchunks + ch) * 4;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
}
// pass 2: states[] := TRUE start state of every chunk.  The chunk recurrence s' = M s + z has the same M for every chunk, so
// an inclusive scan only needs powers of M.  Two launches, both with workgroup = (tile of 4096 chunks, channel) and
// thread = KW_SER consecutive chunks:
//   k_kw_scan_local: (a) serial walk of the thread's chunks from a zero state, (b) Hillis-Steele inside each wave (6 shuffle
//     steps, spans M^8 .. M^256), (c) Hillis-Steele over the 8 wave totals (wave 0), (d) thread start = previous lane's
//     prefix + (M^8)^lane * (wave carry-in), serial walk again storing the chunk start states RELATIVE TO A ZERO TILE START,
//     plus the tile's end state tot[c][tile].
//   k_kw_scan_carry: tile carry = sum over the earlier tiles' totals (a handful of M^4096 steps), added to every chunk of the
//     tile as M^n * carry, n = chunk index in the tile = i + 8*lane + 512*wave.
// ptab: [0..63] = (M^8)^lane, [64..71] = (M^512)^wave (row-major 4x4 each).
__global__ __launch_bounds__(KW_TILE) void k_kw_scan_local(const KwCoef* __restrict__ kp, int nchunks, int ntiles, const double* __restrict__ ptab,
                                                           double* __restrict__ states, double* __restrict__ tot /*[C][ntiles][4]*/) {
    DBG_CLK(0, 0);
    __shared__ double wt[KW_TILE / 64][4];
    __shared__ double mp[13][16];                     // the powers of M: one global round trip instead of one scalar-cache miss per use
    if (threadIdx.x < 13 * 16) (&mp[0][0])[threadIdx.x] = (&kp->Mp[0][0])[threadIdx.x];
    const int c = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* st = states + (int64_t)c * nchunks * 4;
    const int c0 = tile * (KW_TILE * KW_SER) + tid * KW_SER;
    double v[4] = {0, 0, 0, 0}, m[4], u[4];
    double2 z[KW_SER][2];                             // the thread's chunk end states: all loads in flight together
#pragma unroll
    for (int i = 0; i < KW_SER; ++i) {
        const int ch = c0 + i < nchunks ? c0 + i : nchunks - 1;
        const double2* q = (const double2*)(st + (int64_t)ch * 4);
        z[i][0] = q[0];
        z[i][1] = q[1];
    }
    double P[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 q = ((const double2*)(ptab + lane * 16))[i];
        P[2 * i] = q.x;
        P[2 * i + 1] = q.y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KW_SER; ++i) {
        if (c0 + i >= nchunks) z[i][0] = z[i][1] = make_double2(0.0, 0.0);
        kw_matvec(mp[0], v, m);
        v[0] = m[0] + z[i][0].x; v[1] = m[1] + z[i][0].y; v[2] = m[2] + z[i][1].x; v[3] = m[3] + z[i][1].y;
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int d = 1 << b;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = __shfl_up(v[r], d, 64);
        kw_matvec(mp[3 + b], u, m);
        if (lane >= d) { v[0] += m[0]; v[1] += m[1]; v[2] += m[2]; v[3] += m[3]; }
    }
    if (lane == 63) { wt[w][0] = v[0]; wt[w][1] = v[1]; wt[w][2] = v[2]; wt[w][3] = v[3]; }
    __syncthreads();
    if (w == 0) {
        double t[4] = {0, 0, 0, 0};
        if (lane < KW_TILE / 64) { t[0] = wt[lane][0]; t[1] = wt[lane][1]; t[2] = wt[lane][2]; t[3] = wt[lane][3]; }
#pragma unroll
        for (int b = 0; (1 << b) < KW_TILE / 64; ++b) {
            const int d = 1 << b;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = __shfl_up(t[r], d, 64);
            kw_matvec(mp[9 + b], u, m);
            if (lane >= d) { t[0] += m[0]; t[1] += m[1]; t[2] += m[2]; t[3] += m[3]; }
        }
        if (lane < KW_TILE / 64) { wt[lane][0] = t[0]; wt[lane][1] = t[1]; wt[lane][2] = t[2]; wt[lane][3] = t[3]; }   // END state of wave `lane`
        if (lane == KW_TILE / 64 - 1) {
            double* o = tot + ((int64_t)c * ntiles + tile) * 4;
            o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
        }
    }
    __syncthreads();
    double cw[4], S[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cw[r] = w == 0 ? 0.0 : wt[w > 0 ? w - 1 : 0][r];
        u[r] = __shfl_up(v[r], 1, 64);
        if (lane == 0) u[r] = 0.0;
    }
    kw_matvec(P, cw, m);
    S[0] = u[0] + m[0]; S[1] = u[1] + m[1]; S[2] = u[2] + m[2]; S[3] = u[3] + m[3];
#pragma unroll
    for (int i = 0; i < KW_SER; ++i) {
        const int ch = c0 + i;
        if (ch < nchunks) {
            double2* q = (double2*)(st + (int64_t)ch * 4);
            q[0] = make_double2(S[0], S[1]);
            q[1] = make_double2(S[2], S[3]);
        }
        kw_matvec(mp[0], S, m);
        S[0] = m[0] + z[i][0].x; S[1] = m[1] + z[i][0].y; S[2] = m[2] + z[i][1].x; S[3] = m[3] + z[i][1].y;
    }
    DBG_CLK(0, 1);
}
__global__ __launch_bounds__(KW_TILE) void k_kw_scan_carry(const KwCoef* __restrict__ kp, int nchunks, int ntiles, const double* __restrict__ ptab,
                                                           double* __restrict__ states, const double* __restrict__ tot) {
    DBG_CLK(1, 0);
    const KwCoef& k = *kp;
    const int c = blockIdx.y, tile = blockIdx.x + 1, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;   // tile 0 has no carry
    double* st = states + (int64_t)c * nchunks * 4;
    const int c0 = tile * (KW_TILE * KW_SER) + tid * KW_SER;
    double carry[4] = {0, 0, 0, 0}, m[4], x[4], y[4];
    for (int i = 0; i < tile; ++i) {
        const double* t = tot + ((int64_t)c * ntiles + i) * 4;
        kw_matvec(k.Mp[12], carry, m);
        carry[0] = m[0] + t[0]; carry[1] = m[1] + t[1]; carry[2] = m[2] + t[2]; carry[3] = m[3] + t[3];
    }
    double2 z[KW_SER][2];                             // the thread's chunk states: all loads in flight together
#pragma unroll
    for (int i = 0; i < KW_SER; ++i) {
        const int ch = c0 + i < nchunks ? c0 + i : nchunks - 1;
        const double2* q = (const double2*)(st + (int64_t)ch * 4);
        z[i][0] = q[0];
        z[i][1] = q[1];
    }
    double Q[16], P[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 a = ((const double2*)(ptab + (64 + w) * 16))[i], b = ((const double2*)(ptab + lane * 16))[i];
        Q[2 * i] = a.x; Q[2 * i + 1] = a.y;
        P[2 * i] = b.x; P[2 * i + 1] = b.y;
    }
    kw_matvec(Q, carry, x);
    kw_matvec(P, x, y);
#pragma unroll
    for (int i = 0; i < KW_SER; ++i) {
        const int ch = c0 + i;
        if (ch < nchunks) {
            double2* q = (double2*)(st + (int64_t)ch * 4);
            q[0] = make_double2(z[i][0].x + y[0], z[i][0].y + y[1]);
            q[1] = make_double2(z[i][1].x + y[2], z[i][1].y + y[3]);
        }
        kw_matvec(k.Mp[0], y, m);
        y[0] = m[0]; y[1] = m[1]; y[2] = m[2]; y[3] = m[3];
    }
    DBG_CLK(1, 1);
}
// Passes 1-3 in ONE launch when the filter forgets fast enough (every practical K-weighting: the slowest pole, 38 Hz, decays
// below 1e-20 within H <= 256 chunks): workgroup = (NT - H new chunks + H history chunks, channel), thread = chunk.  The
// history chunks start from a zero state; what that ignores is below 1e-20 of the state by the first new chunk, far under
// float64 resolution.  Phase 1: zero-state walk -> chunk end state.  Phase 2: Hillis-Steele inside each wave (M^1..M^32) and
// over the wave totals (M^64..), chunk start = previous lane's prefix + M^lane * (wave carry-in).  Phase 3: the new chunks
// walk again from their start state and keep their energy (+ the start state, for k_block_power_chunks' edge chunks).
// R = float (the product's form since round 6: the delta-form recurrence keeps float32 within ~1e-6 dB of the float64 walk, see KwTabT) or double
// (the round-2..5 form; tuning knob SS_KW_F64, and what the exact multi-launch path below still computes in).  The chunk energies and the start
// states leave the kernel as float64 either way: block powers, gating and the gain are float64 sums (k_block_power_chunks, k_gate).
template <int NT, class R>
__global__ __launch_bounds__(NT) void k_kw_fused(const float* __restrict__ audio, int64_t T, int64_t st, int64_t sc,
                                                 const KwTabT<R>* __restrict__ kp, const R* __restrict__ plane /*[64][16] = M^lane*/,
                                                 int nchunks, int H, double* __restrict__ states, double* __restrict__ energy) {
    constexpr int NW = NT / 64;
    __shared__ float xs[NW][64][KW_ROW];              // the wave's 64 chunks, one padded row per chunk
    __shared__ R wt[NW][4];
    __shared__ R mp[10][16];
    const KwTabT<R>& k = *kp;
    const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ch = blockIdx.x * (NT - H) - H + tid;
    if (tid < 160) (&mp[0][0])[tid] = (&kp->Mp[0][0])[tid];
    const bool live = ch >= 0 && ch < nchunks;
    const float* a = audio + c * sc;
    // the wave's 64 chunks are 16 KiB of contiguous samples: fetch them with fully coalesced 16-byte loads and hand each
    // thread its chunk through LDS (a per-thread walk of global memory touches 64 cache lines per load instruction)
    const int chw = ch - lane;
    const int64_t tw = (int64_t)chw * KW_CHUNK;
    const bool fast = st == 1 && chw >= 0 && tw + 64 * KW_CHUNK <= T && (((uintptr_t)(a + tw)) & 15) == 0;
    if (fast) {
        const float4* g = (const float4*)(a + tw);
        float4 q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] = g[j * 64 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) *(float4*)&xs[w][4 * j + (lane >> 4)][(lane & 15) * 4] = q[j];
    }
    const float* row = &xs[w][lane][0];
    const int64_t t0 = (int64_t)ch * KW_CHUNK;
    const int64_t t1 = t0 + KW_CHUNK < T ? t0 + KW_CHUNK : T;
    R v[4] = {0, 0, 0, 0}, m[4], u[4];
    DBG_CLK(0, 0);
    DBG_CLK_T(8, 0, 64);
    __syncthreads();
    DBG_CLK_T(8, 1, 64);
    if (live) {
        if (fast) kw_endstate<R>(row, 0, KW_CHUNK, 1, k, v);
        else kw_endstate<R>(a, t0, t1, st, k, v);
    }
    DBG_CLK(1, 0);
    DBG_CLK(1, 1);
    DBG_CLK_T(9, 0, 64);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int d = 1 << b;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = __shfl_up(v[r], d, 64);
        kw_matvec<R>(mp[b], u, m);
        if (lane >= d) { v[0] += m[0]; v[1] += m[1]; v[2] += m[2]; v[3] += m[3]; }
    }
    if (lane == 63) { wt[w][0] = v[0]; wt[w][1] = v[1]; wt[w][2] = v[2]; wt[w][3] = v[3]; }
    __syncthreads();
    if (w == 0) {
        R t[4] = {0, 0, 0, 0};
        if (lane < NW) { t[0] = wt[lane][0]; t[1] = wt[lane][1]; t[2] = wt[lane][2]; t[3] = wt[lane][3]; }
#pragma unroll
        for (int b = 0; (1 << b) < NW; ++b) {
            const int d = 1 << b;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = __shfl_up(t[r], d, 64);
            kw_matvec<R>(mp[6 + b], u, m);
            if (lane >= d) { t[0] += m[0]; t[1] += m[1]; t[2] += m[2]; t[3] += m[3]; }
        }
        if (lane < NW) { wt[lane][0] = t[0]; wt[lane][1] = t[1]; wt[lane][2] = t[2]; wt[lane][3] = t[3]; }   // END state of wave `lane`
    }
    __syncthreads();
    R cw[4], S[4], P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = plane[lane * 16 + i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cw[r] = w == 0 ? (R)0 : wt[w > 0 ? w - 1 : 0][r];
        u[r] = __shfl_up(v[r], 1, 64);
        if (lane == 0) u[r] = (R)0;
    }
    kw_matvec<R>(P, cw, m);
    S[0] = u[0] + m[0]; S[1] = u[1] + m[1]; S[2] = u[2] + m[2]; S[3] = u[3] + m[3];
    DBG_CLK(7, 0);
    DBG_CLK_T(9, 1, 64);
    if (live && tid >= H) {
        double2* q = (double2*)(states + ((int64_t)c * nchunks + ch) * 4);
        q[0] = make_double2((double)S[0], (double)S[1]);
        q[1] = make_double2((double)S[2], (double)S[3]);
        R e = 0;
        auto f = [&](float x) {
            const R y = kw_step<R>(k, S, (R)x);
            e = fma(y, y, e);
        };
        if (fast) kw_walk(row, 0, KW_CHUNK, 1, f);
        else kw_walk(a, t0, t1, st, f);
        energy[(int64_t)c * nchunks + ch] = (double)e;
    }
    DBG_CLK(0, 1);
    DBG_CLK_T(10, 0, 64);
}
// The float32 form of k_kw_fused with the thread's chunk held in REGISTERS: the wave's 16 KiB arrive as 16 float4 per lane (quarter q of 16 chunks per
// load: 64 contiguous bytes per 4 lanes), and four rounds through a 5 KiB per-wave transit buffer turn them, in place, into the lane's own chunk (round q:
// write the four registers of quarter q, read back samples 16q .. 16q+15 of chunk `lane`; rows of 20 floats: conflict-free b128 on both sides).  LDS per
// workgroup 21 KiB instead of 70 (the staged chunks), 4 waves per SIMD instead of 2, no barrier around the staging (the transit buffer is the wave's own),
// and neither walk reads LDS.  Same arithmetic and the same order of operations as k_kw_fused<NT, float>: identical bits (tests/test_gpu_aux.py).
constexpr int KW_TROW = 20;
typedef float kw_v4f __attribute__((ext_vector_type(4)));
// HPS: the second stage's numerator is a scaled double difference b = g (1, -2, 1) -- BS.1770's high-pass as pyloudnorm generates it (beta = 0, b0 = b2 = g):
// y = g (d' - d).  The walk keeps d' - d (three operations fewer per sample than the general form) and the chunk energy is scaled by g^2 in float64 at the end.
template <bool HPS>
__device__ __forceinline__ float kw_step32(const KwCoefF& k, float s[4], float xin) {
    const float y1 = kw_stage<float>(k.c[0], s[0], s[1], xin);
    if (!HPS) return kw_stage<float>(k.c[1], s[2], s[3], y1);
    float t = y1 + s[3];
    t = fma(-k.c[1][1], s[2], t);
    const float dn = fma(-k.c[1][0], s[3], t);
    const float y = dn - s[3];
    s[2] += dn;
    s[3] = dn;
    return y;
}
// The end states are a (4 x 64) x (64 x 64 chunks) product whose columns are the lanes: 64 rank-one updates v_mfma_f32_4x4x1 (16 blocks of 4 lanes: A = the
// lane's W[t][lane & 3], B = the lane's own sample t, D = the lane's four state components) -- on the matrix pipe, beside the other waves' walks on the VALU,
// with the coefficients from a 1 KiB LDS table instead of 16 dependent scalar-cache round trips (1.6 -> 0.5 us per workgroup, profiles/r06ag).
template <int NT, bool HPS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_kw_fused32(const float* __restrict__ audio, int64_t T, int64_t st, int64_t sc,
                                                   const KwCoefF* __restrict__ kp, const float* __restrict__ plane /*[64][16] = M^lane*/,
                                                   int nchunks, int H, double* __restrict__ states, double* __restrict__ energy, double g2 /* HPS: b0^2 of stage 2 */) {
    constexpr int NW = NT / 64;
    __shared__ float xt[NW][64 * KW_TROW];
    __shared__ float wt[NW][4];
    __shared__ float mp[10][16];
    __shared__ float wtab[4][KW_CHUNK];                // W transposed: wtab[r][t]
    const KwCoefF& k = *kp;
    const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ch = blockIdx.x * (NT - H) - H + tid;
    if (tid < 160) (&mp[0][0])[tid] = (&kp->Mp[0][0])[tid];
    if (tid < 256) wtab[tid & 3][tid >> 2] = (&kp->W[0][0])[tid];
    const bool live = ch >= 0 && ch < nchunks;
    const float* a = audio + c * sc;
    const int chw = ch - lane;
    const int64_t tw = (int64_t)chw * KW_CHUNK;
    const bool fast = st == 1 && chw >= 0 && tw + 64 * KW_CHUNK <= T && (((uintptr_t)(a + tw)) & 15) == 0;
    float4 X[16];
    if (fast) {
        const float4* g = (const float4*)(a + tw) + (lane >> 2) * 16 + (lane & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) X[4 * q + j] = g[(16 * j) * 16 + 4 * q];
        float* tr = &xt[w][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(float4*)&tr[(16 * j + (lane >> 2)) * KW_TROW + (lane & 3) * 4] = X[4 * q + j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) X[4 * q + i] = *(const float4*)&tr[lane * KW_TROW + 4 * i];
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int64_t t0 = (int64_t)ch * KW_CHUNK;
    const int64_t t1 = t0 + KW_CHUNK < T ? t0 + KW_CHUNK : T;
    float v[4] = {0, 0, 0, 0}, m[4], u[4];
    DBG_CLK(0, 0);
    DBG_CLK_T(8, 0, 64);
    __syncthreads();                                  // (mp[], wtab[])
    DBG_CLK_T(8, 1, 64);
    if (fast) {
        kw_v4f d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
        const float* wl = &wtab[lane & 3][0];
#pragma unroll
        for (int i = 0; i < KW_CHUNK / 4; ++i) {
            const float4 wv = *(const float4*)&wl[4 * i];
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, X[i].x, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, X[i].y, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, X[i].z, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, X[i].w, d1, 0, 0, 0);
        }
        v[0] = d0[0] + d1[0]; v[1] = d0[1] + d1[1]; v[2] = d0[2] + d1[2]; v[3] = d0[3] + d1[3];
    } else if (live) kw_endstate<float>(a, t0, t1, st, k, v);
    DBG_CLK(1, 0);
    DBG_CLK(1, 1);
    DBG_CLK_T(9, 0, 64);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int d = 1 << b;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = __shfl_up(v[r], d, 64);
        kw_matvec<float>(mp[b], u, m);
        if (lane >= d) { v[0] += m[0]; v[1] += m[1]; v[2] += m[2]; v[3] += m[3]; }
    }
    if (lane == 63) { wt[w][0] = v[0]; wt[w][1] = v[1]; wt[w][2] = v[2]; wt[w][3] = v[3]; }
    __syncthreads();
    if (w == 0) {
        float t[4] = {0, 0, 0, 0};
        if (lane < NW) { t[0] = wt[lane][0]; t[1] = wt[lane][1]; t[2] = wt[lane][2]; t[3] = wt[lane][3]; }
#pragma unroll
        for (int b = 0; (1 << b) < NW; ++b) {
            const int d = 1 << b;
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = __shfl_up(t[r], d, 64);
            kw_matvec<float>(mp[6 + b], u, m);
            if (lane >= d) { t[0] += m[0]; t[1] += m[1]; t[2] += m[2]; t[3] += m[3]; }
        }
        if (lane < NW) { wt[lane][0] = t[0]; wt[lane][1] = t[1]; wt[lane][2] = t[2]; wt[lane][3] = t[3]; }   // END state of wave `lane`
    }
    __syncthreads();
    float cw[4], S[4], P[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 p4 = ((const float4*)(plane + lane * 16))[i];
        P[4 * i] = p4.x; P[4 * i + 1] = p4.y; P[4 * i + 2] = p4.z; P[4 * i + 3] = p4.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cw[r] = w == 0 ? 0.f : wt[w > 0 ? w - 1 : 0][r];
        u[r] = __shfl_up(v[r], 1, 64);
        if (lane == 0) u[r] = 0.f;
    }
    kw_matvec<float>(P, cw, m);
    S[0] = u[0] + m[0]; S[1] = u[1] + m[1]; S[2] = u[2] + m[2]; S[3] = u[3] + m[3];
    DBG_CLK(7, 0);
    DBG_CLK_T(9, 1, 64);
    if (live && tid >= H) {
        double2* q = (double2*)(states + ((int64_t)c * nchunks + ch) * 4);
        q[0] = make_double2((double)S[0], (double)S[1]);
        q[1] = make_double2((double)S[2], (double)S[3]);
        float e = 0;
        auto f = [&](float x) {
            const float y = kw_step32<HPS>(k, S, x);
            e = fma(y, y, e);
        };
        if (fast) {
#pragma unroll
            for (int i = 0; i < KW_CHUNK / 4; ++i) { f(X[i].x); f(X[i].y); f(X[i].z); f(X[i].w); }
        } else kw_walk(a, t0, t1, st, f);
        energy[(int64_t)c * nchunks + ch] = HPS ? (double)e * g2 : (double)e;
    }
    DBG_CLK(0, 1);
    DBG_CLK_T(10, 0, 64);
}
// pass 3: re-run every chunk from its true start state and keep only the chunk's K-weighted energy (float64): the gating
// blocks are sums of whole chunks plus two partial edge chunks (k_block_power_chunks), so the filtered signal is never stored
__global__ __launch_bounds__(256) void k_kw_energy(const float* __restrict__ audio, int64_t T, int C, int64_t st, int64_t sc, const KwCoef* __restrict__ kp,
                                                   int nchunks, const double* __restrict__ states, double* __restrict__ energy /*[C][nchunks]*/) {
    const KwCoef& k = *kp;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * nchunks) return;
    const int c = id / nchunks, ch = id - c * nchunks;
    const double* z = states + ((int64_t)c * nchunks + ch) * 4;
    double s[4] = {z[0], z[1], z[2], z[3]};
    const int64_t t0 = (int64_t)ch * KW_CHUNK;
    const int64_t t1 = t0 + KW_CHUNK < T ? t0 + KW_CHUNK : T;
    double e = 0.0;
    kw_walk(audio + c * sc, t0, t1, st, [&](float x) {
        const double v = kw_step(k, s, (double)x);
        e += v * v;
    });
    energy[(int64_t)c * nchunks + ch] = e;
}
// z[c][j] = (1/norm) * sum_{t in [lo_j, hi_j)} k(x_c)[t]^2 from the chunk energies: whole chunks are added in a fixed order by
// the 64 lanes, the (at most two) partial edge chunks are re-filtered from their stored start state by lanes 0 and 1.
__global__ __launch_bounds__(64) void k_block_power_chunks(const float* __restrict__ audio, int64_t T, int64_t st, int64_t sc, const KwCoef* __restrict__ kp,
                                                           int nchunks, const double* __restrict__ states, const double* __restrict__ energy,
                                                           const int64_t* __restrict__ lo, const int64_t* __restrict__ hi, int nblocks,
                                                           double inv_norm, double* __restrict__ z /*[C][nblocks]*/) {
    const KwCoef& k = *kp;
    const int j = blockIdx.x, c = blockIdx.y;
    int64_t a = lo[j], b = hi[j];
    a = a < 0 ? 0 : a;
    b = b > T ? T : b;
    __shared__ float eb[2][KW_CHUNK];
    double s = 0.0;
    int64_t edge0 = -1, edge1 = -1;                                                // partial chunks (uniform over the wave)
    if (b > a) {
        const int64_t ca = a / KW_CHUNK, cb = (b - 1) / KW_CHUNK;                 // first / last chunk touched
        const int64_t cb_end = (cb + 1) * KW_CHUNK < T ? (cb + 1) * KW_CHUNK : T;
        const bool head_whole = a == ca * KW_CHUNK, tail_whole = b == cb_end;
        const double* e = energy + (int64_t)c * nchunks;
        if (ca == cb) {
            if (head_whole && tail_whole) { if (threadIdx.x == 0) s += e[ca]; }
            else edge0 = ca;
        } else {
            const int64_t f0 = head_whole ? ca : ca + 1, f1 = tail_whole ? cb + 1 : cb;
            for (int64_t q = f0 + threadIdx.x; q < f1; q += 64) s += e[q];
            if (!head_whole) edge0 = ca;
            if (!tail_whole) edge1 = cb;
        }
    }
    const float* au = audio + c * sc;
    for (int i = threadIdx.x; i < KW_CHUNK; i += 64) {                             // edge samples: one coalesced load each
        const int64_t ta = edge0 * KW_CHUNK + i, tb = edge1 * KW_CHUNK + i;
        eb[0][i] = (edge0 >= 0 && ta < T) ? au[ta * st] : 0.f;
        eb[1][i] = (edge1 >= 0 && tb < T) ? au[tb * st] : 0.f;
    }
    __syncthreads();
    {
        const int64_t edge = threadIdx.x == 0 ? edge0 : (threadIdx.x == 1 ? edge1 : -1);
        if (edge >= 0) {                                                           // lanes 0 / 1 re-filter their edge chunk
            const double* zz = states + ((int64_t)c * nchunks + edge) * 4;
            double sv[4] = {zz[0], zz[1], zz[2], zz[3]};
            const int64_t t0 = edge * KW_CHUNK;
            int64_t t1 = t0 + KW_CHUNK < T ? t0 + KW_CHUNK : T;
            t1 = t1 < b ? t1 : b;
            const float* eb_ = eb[threadIdx.x];
#pragma unroll 8
            for (int64_t t = t0; t < t1; ++t) {
                const double v = kw_step(k, sv, (double)eb_[t - t0]);
                if (t >= a) s += v * v;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) z[(int64_t)c * nblocks + j] = inv_norm * s;
}
// BS.1770-4 two-stage gating (pyloudnorm.Meter.integrated_loudness) over z[C][nblocks], then the gain of lufs_norm
// (SonicSim_audio.py:68-77): res[0] = integrated loudness (-inf when no block survives), res[1] = 10^((target - L)/20) with
// L := -40 when the loudness is -inf (the reference's fallback).  One workgroup; wave w reduces channels w, w+4, ...
struct GateTargets { double t[16]; };   // drawn class loudness per stem
__global__ __launch_bounds__(1024) void k_gate(const double* __restrict__ z, int C, int nblocks, const double* __restrict__ gw /*[C] channel weights*/, GateTargets targets,
                                               double* __restrict__ lbuf /*[nblocks] scratch*/, int use_lds, double* __restrict__ res) {
    // blockIdx.x = group (stem) of C channels: its own z rows, scratch, target and result slot
    z += (int64_t)blockIdx.x * C * nblocks;
    lbuf += (int64_t)blockIdx.x * nblocks;
    res += 4 * blockIdx.x;
    const double target = targets.t[blockIdx.x];
    DBG_CLK(2, 0);
    extern __shared__ double gsm[];                      // use_lds: [C][nblocks] copy of z, then [nblocks] block loudness
    __shared__ double part[65];                          // [channel] sums over the kept blocks, [64] = kept-block count
    __shared__ double rel_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* lb = use_lds ? gsm + (size_t)C * nblocks : lbuf;
    const double* zz = use_lds ? gsm : z;
    for (int j = tid; j < nblocks; j += 1024) {
        double s = 0.0;
        for (int c = 0; c < C; ++c) {
            const double v = z[(int64_t)c * nblocks + j];
            if (use_lds) gsm[(size_t)c * nblocks + j] = v;
            s += gw[c] * v;
        }
        lb[j] = -0.691 + 10.0 * log10(s);
    }
    __syncthreads();
    DBG_CLK(4, 0);
    double loud = -INFINITY;
    for (int stage = 0; stage < 2; ++stage) {
        const double rel = stage ? rel_s : 0.0;
        for (int c = w; c <= C; c += 16) {               // wave per channel; c == C: the kept-block count
            double s = 0.0;
#pragma unroll 4
            for (int j = lane; j < nblocks; j += 64) {
                const double l = lb[j];
                const bool keep = stage ? (l > rel && l > -70.0) : (l >= -70.0);
                const double zv = c < C ? zz[(int64_t)c * nblocks + j] : 1.0;
                if (keep) s += zv;
            }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) part[c < C ? c : 64] = s;
        }
        __syncthreads();
        DBG_CLK(5, stage);
        if (tid == 0) {
            const double nk = part[64];
            double acc = 0.0;
            for (int c = 0; c < C; ++c) acc += gw[c] * (part[c] / nk);
            const double l = -0.691 + 10.0 * log10(acc);
            if (stage == 0) rel_s = nk > 0.0 ? l - 10.0 : NAN;
            else loud = nk > 0.0 ? l : -INFINITY;
        }
        __syncthreads();
        DBG_CLK(6, stage);
    }
    if (tid == 0) {
        res[0] = loud;
        const double used = isinf(loud) ? -40.0 : loud;
        res[1] = pow(10.0, (target - used) / 20.0);
    }
    DBG_CLK(2, 1);
}

// =============================================================================================
// host side
// =============================================================================================
namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? SS_ENOMEM : SS_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#include "hostpipe.h"

enum WsSlot { WS_XS, WS_PLAN, WS_BMIN, WS_BMAX, WS_X, WS_BANK, WS_IDX, WS_W, WS_Y, WS_SCR, WS_SCR2, WS_FILT, WS_META, WS_CNT, WS_LUFS, WS_RES, WS_KWP, WS_KWT, WS_GW, WS_DPLAN, WS_DTASKS, WS_K1, WS_SQ, WS_HS, WS_STATUS, WS_COUNT };

struct Pinned {
    void* host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct EvPair {
    hipEvent_t a, b;
    int kind;
};

struct Ctx {
    int device = -1;
    bool inited = false;
    c32* consts = nullptr;
    c32* consts12 = nullptr;
    c32* consts13 = nullptr;
    c32* consts14 = nullptr;
    hipModule_t mod13 = nullptr;          // hand-scheduled gfx950 render kernel (tools/gen_asm/os13.py -> lib/k_os13_gfx950.hsaco)
    hipFunction_t fn13 = nullptr;
    bool mod13_tried = false;
    hipModule_t mod13q = nullptr;         // the same kernel with dynamic per-XCD task queues (OS13_OPT=dynq -> lib/k_os13_gfx950_dynq.hsaco)
    hipFunction_t fn13q = nullptr;
    bool mod13q_tried = false;
    int os_geom = 0;        // SS_OS_GEOM: 11 (B=2048, 256 thr) / 12 (B=4096, 512 thr, spectrum window); 0 = by filter length
    void* ws[WS_COUNT] = {};              // the ACTIVE lane's workspace (see Lane / stream_enter)
    size_t ws_cap[WS_COUNT] = {};
    static constexpr int NRING = 8;
    Pinned ring[NRING];
    int ring_next = 0;
    // Workspace LANES (round 5): the workspace is stream-ordered, and rounds 1-4 serialised a stream switch against the previous stream.  A caller
    // that alternates two streams -- independent renders, render i + 1's spectra launch and first tickets under render i's draining tail -- now gets
    // one private workspace per stream: up to NLANE streams are live at once, a further one takes over the least recently used lane after
    // synchronising that lane's stream.  ws[] / ws_cap[] above are the active lane's slots; WS_K1 (the bank generator's own lane) is shared.
    static constexpr int NLANE = 4;
    struct Lane {
        void* ws[WS_COUNT] = {};
        size_t cap[WS_COUNT] = {};
        hipStream_t stream = nullptr;
        bool used = false;
        bool dev_planned = false;         // this lane's last render planned its schedule on the device (ss_plan_status_last is about the CALLING stream's lane)
        bool status_zeroed = false;       // WS_STATUS of this lane has been cleared once (k_front_explicit's arrival counter lives in its word 7)
        uint64_t tick = 0;
    } lanes[NLANE];
    int cur_lane = 0;
    uint64_t lane_tick = 0;
    int lane_switches = 0, lane_evictions = 0;
    size_t k1_batch_per = 0;              // slot words per bank of the batched generator's layout in WS_K1 (0: the single-bank layout)
    hipStream_t k1_stream = nullptr;      // the bank generator's own lane (it only touches WS_K1): see stream_enter_k1
    bool have_k1 = false;
    bool prof = false;
    std::vector<EvPair> evs;
    int prof_every = 1;                    // ss_prof_enable(N > 1): time every N-th launch of each kind only
    unsigned prof_seen[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> ev_pool;      // recycled timing events (creating events costs host time inside the timed loop)
    int os_variant = 3;     // SS_OS_VARIANT: prefetch depth of the render kernel (0, 2, 3) -- tuning knob
    int os_ablate = 0;      // SS_OS_ABLATE: profiling-only ablation mask (results are WRONG when != 0)
    int plan_mode = 2;      // how the assembly engine's plan reaches the GPU: 2 = staged into HBM by the spectra kernel (default), 1 = read in
                            // place from device-mapped pinned memory by the render kernel, 0 = stream-ordered hipMemcpyAsync (SS_ZERO_COPY_PLAN knob)
    bool xcd_order = true;  // SS_XCD_ORDER=0 disables the XCD-aware task order -- tuning knob
    bool dynq = true;       // default; ss_set_task_queue(0) selects the static lists: per-XCD dynamic task queues (robust when anything else holds compute units; alone as fast as the static lists)
    // host scratch reused across calls
    std::vector<int64_t> seg_start;
    std::vector<int32_t> bmin, bmax;
    Plan plan;
    std::vector<Task> merged;
    std::vector<int32_t> plan_scratch;
    std::vector<int64_t> lufs_bounds;     // block bounds currently resident in ws[WS_LUFS]
    void* lufs_bounds_dev = nullptr;
    double kw_cached[12] = {0};           // biquad coefficients whose KwCoef + carry-power tables are resident in ws[WS_KWP]
    double gw_cached[64] = {0};           // channel weights resident in ws[WS_GW]
    void* gw_cached_dev = nullptr;
    void* kw_cached_dev = nullptr;
    int num_cu = 0;
    bool last_dev_planned = false;        // the last render() on this device planned its schedule on the device (ss_plan_status_last is about it)
    int32_t* async_status = nullptr;      // device: {code, where} latched by k_plan_explicit (SS_FLAG_ASYNC_PLAN), read by ss_async_status
    int32_t* status_pin = nullptr;        // pinned host words the planner mirrors this run's verdict into (ss_convolve_moving_checked_f32)
    HostPipe pipe;          // host-pointer mode: pinned staging rings, copy streams, copy threads (hostpipe.h)
    std::vector<Task> chunk_tmp;
    std::mutex mu;          // one lock per device context: entry points are re-entrant per device (one host thread per GPU works)
};

std::mutex g_mu;            // guards the context map and one-time initialisation only
std::map<int, Ctx*> g_ctx;

int get_ctx(Ctx** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(SS_ENODEV, "hipGetDevice failed: %s (no usable GPU; the HIP path has no CPU fallback)", hipGetErrorString(e));
    std::lock_guard<std::mutex> map_lock(g_mu);
    auto it = g_ctx.find(dev);
    Ctx* c;
    if (it == g_ctx.end()) {
        c = new Ctx();
        c->device = dev;
        g_ctx[dev] = c;
    } else {
        c = it->second;
    }
    if (!c->inited) {
        std::vector<c32> tab;
        build_consts(tab);
        HIPCHK(hipMalloc((void**)&c->consts, sizeof(c32) * CONST_C32));
        HIPCHK(hipMemcpy(c->consts, tab.data(), sizeof(c32) * CONST_C32, hipMemcpyHostToDevice));
        build_consts12(tab);
        HIPCHK(hipMalloc((void**)&c->consts12, sizeof(c32) * CONST12_C32));
        HIPCHK(hipMemcpy(c->consts12, tab.data(), sizeof(c32) * CONST12_C32, hipMemcpyHostToDevice));
        build_consts13(tab);
        HIPCHK(hipMalloc((void**)&c->consts13, sizeof(c32) * CONST13_C32));
        HIPCHK(hipMemcpy(c->consts13, tab.data(), sizeof(c32) * CONST13_C32, hipMemcpyHostToDevice));
        build_consts14(tab);
        HIPCHK(hipMalloc((void**)&c->consts14, sizeof(c32) * CONST14_C32));
        HIPCHK(hipMemcpy(c->consts14, tab.data(), sizeof(c32) * CONST14_C32, hipMemcpyHostToDevice));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char* e = knob("SS_OS_GEOM")) c->os_geom = atoi(e);
        if (const char* e = knob("SS_OS_VARIANT")) c->os_variant = atoi(e);
#ifdef SS_ABLATE
        if (const char* e = knob("SS_OS_ABLATE")) c->os_ablate = atoi(e);
#endif
        if (const char* e = knob("SS_XCD_ORDER")) c->xcd_order = atoi(e) != 0;
        if (const char* e = knob("SS_ZERO_COPY_PLAN")) c->plan_mode = atoi(e);
        if (const char* e = knob("SS_DYNQ")) c->dynq = atoi(e) != 0;
        c->inited = true;
    }
    *out = c;
    return SS_OK;
}

int ws_ensure(Ctx* c, int slot, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (c->ws_cap[slot] >= bytes) return SS_OK;
    if (c->ws[slot]) HIPCHK(hipFree(c->ws[slot]));   // hipFree synchronises the device
    c->ws[slot] = nullptr;
    c->ws_cap[slot] = 0;
    size_t cap = bytes + bytes / 8;
    HIPCHK(hipMalloc(&c->ws[slot], cap));
    c->ws_cap[slot] = cap;
    return SS_OK;
}

// stream switch: the workspace is stream-ordered -- every stream works in its own lane (Ctx::Lane), so two streams never share a buffer and a
// switch costs a few pointer copies instead of a device synchronisation.  Only when more than NLANE streams are live is the least recently used
// lane's stream synchronised and its buffers handed to the newcomer.
int stream_enter(Ctx* c, hipStream_t s) {
    Ctx::Lane& cur = c->lanes[c->cur_lane];
    ++c->lane_tick;
    if (cur.used && cur.stream == s) { cur.tick = c->lane_tick; return SS_OK; }
    int pick = -1;
    for (int i = 0; i < Ctx::NLANE; ++i) if (c->lanes[i].used && c->lanes[i].stream == s) pick = i;
    if (pick < 0) for (int i = 0; i < Ctx::NLANE && pick < 0; ++i) if (!c->lanes[i].used) pick = i;
    if (pick < 0) {                          // every lane belongs to another stream: take over the least recently used one
        for (int i = 0; i < Ctx::NLANE; ++i) if (i != c->cur_lane && (pick < 0 || c->lanes[i].tick < c->lanes[pick].tick)) pick = i;
        HIPCHK(hipStreamSynchronize(c->lanes[pick].stream));
        ++c->lane_evictions;
    }
    if (pick != c->cur_lane) {
        for (int i = 0; i < WS_COUNT; ++i) {
            if (i == WS_K1) continue;        // the bank generator's slots are ordered by ITS stream (stream_enter_k1), whatever lane is active
            cur.ws[i] = c->ws[i]; cur.cap[i] = c->ws_cap[i];
            c->ws[i] = c->lanes[pick].ws[i]; c->ws_cap[i] = c->lanes[pick].cap[i];
        }
        c->cur_lane = pick;
        ++c->lane_switches;
    }
    c->lanes[pick].used = true;
    c->lanes[pick].stream = s;
    c->lanes[pick].tick = c->lane_tick;
    return SS_OK;
}

// host reads of state that any lane may have written (the device planner's status words): every OTHER live stream is drained first
int sync_other_lanes(Ctx* c, hipStream_t s) {
    for (int i = 0; i < Ctx::NLANE; ++i)
        if (c->lanes[i].used && c->lanes[i].stream != s) HIPCHK(hipStreamSynchronize(c->lanes[i].stream));
    return SS_OK;
}

// The bank generator with device-resident geometry touches no shared workspace but its own peak slots (WS_K1): it is ordered against
// other generator launches only, so a scene pipeline can run the NEXT scene's generator on a second stream while the current scene's
// loudness / mix kernels run on the first (pipeline.SceneRenderer, round 4) without the stream switch synchronising the device.
int stream_enter_k1(Ctx* c, hipStream_t s) {
    if (c->have_k1 && c->k1_stream != s) HIPCHK(hipStreamSynchronize(c->k1_stream));
    c->k1_stream = s;
    c->have_k1 = true;
    return SS_OK;
}

int pinned_acquire(Ctx* c, size_t bytes, Pinned** out) {
    Pinned& p = c->ring[c->ring_next];
    c->ring_next = (c->ring_next + 1) % Ctx::NRING;
    if (p.pending) {
        HIPCHK(hipEventSynchronize(p.ev));
        p.pending = false;
    }
    if (!p.ev) HIPCHK(hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
    if (p.cap < bytes) {
        if (p.host) HIPCHK(hipHostFree(p.host));
        p.host = nullptr;
        p.cap = 0;
        HIPCHK(hipHostMalloc(&p.host, bytes + bytes / 4 + 4096, hipHostMallocDefault));
        p.cap = bytes + bytes / 4 + 4096;
    }
    *out = &p;
    return SS_OK;
}

struct ProfScope {
    Ctx* c;
    hipStream_t s;
    int kind;
    bool on;
    EvPair ev;
    ProfScope(Ctx* c_, hipStream_t s_, int kind_) : c(c_), s(s_), kind(kind_), on(c_->prof) {
        if (on && (c->prof_seen[kind_ & 3]++ % c->prof_every) != 0) on = false;   // sampled: every N-th launch
        if (on) {
            ev.kind = kind;
            auto get = [&](hipEvent_t* e) {
                if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
                return hipEventCreate(e) == hipSuccess;
            };
            if (!get(&ev.a) || !get(&ev.b)) { on = false; return; }
            hipEventRecord(ev.a, s);
        }
    }
    ~ProfScope() {
        if (on) {
            hipEventRecord(ev.b, s);
            c->evs.push_back(ev);
        }
    }
};

inline int grid_for(int64_t n, int cap = 2048) {
    int64_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// kernel arguments of k_os13_asm (layout fixed by tools/gen_asm/os13.py: ARG)
struct Os13AsmArgs {
    const void* bank;
    const void* Xs;
    const void* tasks;
    const void* seg_start;
    const void* inv_seg;
    void* y;
    int64_t T;
    int32_t P, C, L, NP, M, ntasks, mode, nwg;
    const void* consts;
    void* counter;
    const void* idx;       // explicit schedule (mode 2): interp_index int64[T], interp_weight float[T]
    const void* w;
    int32_t qgroups;       // dynamic task queues: 0 = static stride-nwg assignment, G = workgroup b pulls from queue b % G (counter[16 * g])
    int32_t rs;            // input spectra every 4096 >> rs samples; Task::j0 in those hop units (plan.h row_tasks)
    // ---- multi-source launches (tools/gen_asm/os13.py: ARG_NSRC, SRC_TAB): Task.chan = source << 16 | channel
    int32_t nsrc;          // <= 1: the fields above describe the one source
    int32_t pad_a;
    const void* hspec;     // partition spectra of the rows marked TASK_SPECTRA_READY (k_row_spectra; os13.py: ARG_HSPEC)
    const void* verdict;   // device-planned schedule: the lane's verdict words (word 2 != 0: nothing was planned -> the kernel fills y with NaN); else null
    int32_t pad0[26];
    struct Src {
        const void* bank;
        const void* Xs;
        const void* seg_start;
        const void* inv_seg;
        void* y;
        int32_t P, C, mode, nwg;
        int32_t pad[2];
    } src[8];
};
static_assert(sizeof(Os13AsmArgs) == 768 && offsetof(Os13AsmArgs, nsrc) == 128 && offsetof(Os13AsmArgs, hspec) == 136 && offsetof(Os13AsmArgs, verdict) == 144 && offsetof(Os13AsmArgs, src) == 256 && sizeof(Os13AsmArgs::Src) == 64,
              "Os13AsmArgs layout");

// The code object sits next to this shared library (built by sonicsim_amd/build.py); a missing file is an error
// for the callers that asked for the assembly engine, never a silent fallback.
int load_mod13(Ctx* c, bool dynq = false) {
    hipModule_t& mod = dynq ? c->mod13q : c->mod13;
    hipFunction_t& fn = dynq ? c->fn13q : c->fn13;
    bool& tried = dynq ? c->mod13q_tried : c->mod13_tried;
    const char* file = dynq ? "k_os13_gfx950_dynq.hsaco" : "k_os13_gfx950.hsaco";
    if (fn) return SS_OK;
    if (tried) return fail(SS_EHIP, "%s could not be loaded (see the first error)", file);
    tried = true;
    Dl_info info;
    if (!dladdr((const void*)&ss_version, &info) || !info.dli_fname) return fail(SS_EHIP, "dladdr failed: cannot locate %s", file);
    std::string path(info.dli_fname);
    const size_t slash = path.find_last_of('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/" + file;
    if (const char* e = knob("SS_HSACO")) path = e;
    hipError_t e = hipModuleLoad(&mod, path.c_str());
    if (e != hipSuccess) return fail(SS_EHIP, "hipModuleLoad(%s) failed: %s", path.c_str(), hipGetErrorString(e));
    e = hipModuleGetFunction(&fn, mod, "k_os13_asm");
    if (e != hipSuccess) return fail(SS_EHIP, "hipModuleGetFunction(k_os13_asm) failed: %s", hipGetErrorString(e));
    return SS_OK;
}

// Rows cut into many tasks are transformed once (plan.h flag_long_rows): marks the tasks of `tasks`, sizes the spectra array and fills
// `ha` for the pre-pass.  Policy: automatic = rows of >= 5 tasks while the array stays within 96 MB.  Measured (profiles/r06w, trajectories of
// 2 .. 200 points at config-2 shapes): a pre-pass transform costs ~1.2 x a transforming interval of the render kernel and a spectra-ready
// interval saves ~2/3 of one, so rows of 3-4 tasks (P = 40 .. 64 there, config 5's longest rows) LOSE -- 0.132 -> 0.151 ms at P = 40 with the
// round's first threshold of 3 -- rows of 5-6 tasks are even, rows of >= 11 (P <= 12, static sources) gain 15-20 %; the array is read once
// per task and has to stay in the L2s / the Infinity Cache to pay (LAB round 6);
// SS_FLAG_ROW_SPECTRA marks every row (tests, measurements), SS_FLAG_NO_ROW_SPECTRA none.
int hrow_mark(Ctx* c, std::vector<Task>& tasks, const int32_t* Ps, int nsrc, int C, int NPart, uint32_t flags, HRowArgs& ha) {
    ha.nrows = 0;
    if (flags & SS_FLAG_NO_ROW_SPECTRA) return SS_OK;
    static const int hrow_min = knob("SS_HROW_MIN") ? atoi(knob("SS_HROW_MIN")) : 5;
    static const int64_t hrow_mb = knob("SS_HROW_MB") ? atoll(knob("SS_HROW_MB")) : 96;
    const bool force = (flags & SS_FLAG_ROW_SPECTRA) != 0;
    for (int s = 0; s < nsrc; ++s) if (Ps[s] > 0xffffff) return SS_OK;
    static thread_local std::vector<int32_t> rows;
    const int n = flag_long_rows(tasks, Ps, nsrc, C, NPart, force ? 1 : hrow_min, force ? ((int64_t)1 << 40) : (hrow_mb << 20), HROW_MAX, rows);
    if (!n) return SS_OK;
    int rc = ws_ensure(c, WS_HS, sizeof(c32) * (size_t)B13 * (size_t)NPart * (size_t)C * (size_t)n);
    if (rc) return rc;
    ha.nrows = n;
    memcpy(ha.rows, rows.data(), sizeof(int32_t) * (size_t)n);
    return SS_OK;
}

void hrow_fill(Ctx* c, HRowArgs& ha, int C, int L, int NPart) {
    ha.Hs = (c32*)c->ws[WS_HS];
    ha.consts = c->consts13;
    ha.C = C; ha.L = L; ha.NP = NPart;
}

int hrow_launch(Ctx* c, HRowArgs& ha, int C, int L, int NPart, hipStream_t stream) {
    if (ha.nrows <= 0) return SS_OK;
    ProfScope ps(c, stream, 3);
    hrow_fill(c, ha, C, L, NPart);
    hipLaunchKernelGGL(k_row_spectra, dim3((unsigned)hrow_workgroups(ha)), dim3(NT13), 0, stream, ha);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// the render engine shared by rows V / I+V / F
int render(int mode, const float* x, int64_t T, const float* bank, int32_t P, int32_t C, int32_t L,
           const int64_t* seg_len_host, const int64_t* idx, const float* w, float* y, uint32_t flags, void* stream_,
           const float* xdiv = nullptr /* device scalar: render with bank / *xdiv (deferred peak normalisation) */,
           int64_t* status_out = nullptr /* [3]: THIS call's device-planner verdict {out_of_range (-1: validated on the host), where, too_irregular};
                                            asking for it synchronises the stream before the context lock is released */) {
    if (T < 0 || P < 1 || C < 1 || L < 1) return fail(SS_EINVAL, "bad shape: T=%lld P=%d C=%d L=%d", (long long)T, P, C, L);
    if (T == 0) return SS_OK;
    if (!x || !bank || !y) return fail(SS_EINVAL, "NULL data pointer");
    if (mode != COEF_FIXED && P < 2) return fail(SS_EINVAL, "moving render needs at least 2 positions (P=%d)", P);
    if (mode == COEF_SEG && !seg_len_host) return fail(SS_EINVAL, "seg_len is NULL");
    if (mode == COEF_EXPLICIT && (!idx || !w)) return fail(SS_EINVAL, "idx / w is NULL");
    if (T > (int64_t)2000000000LL * 4) return fail(SS_EINVAL, "T too large");
    if (mode == COEF_SEG) {        // validated BEFORE anything is put on the wire (an error return must not leave uploads of the caller's arrays in flight)
        int64_t s = 0;
        for (int k = 0; k < P - 1; ++k) {
            if (seg_len_host[k] < 0) return fail(SS_EINVAL, "seg_len[%d] = %lld is negative", k, (long long)seg_len_host[k]);
            s += seg_len_host[k];
        }
        if (s != T) return fail(SS_EINVAL, "sum(seg_len) = %lld != T = %lld", (long long)s, (long long)T);
    }
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    // host-pointer mode: whatever way this function is left, nothing of the call stays in flight -- an error return after hp_begin() used to leave
    // host-to-device DMAs reading the caller's (possibly pinned) arrays and device-to-host pieces pending, drained only by the NEXT host-pointer
    // call; the caller may free its arrays right after the error, and other entry points reuse WS_X / WS_BANK / WS_Y on `stream`.
    struct HpAbort {
        HostPipe* h = nullptr;
        hipStream_t s = nullptr;
        ~HpAbort() {
            if (!h || !h->dirty) return;                   // hp_finish() ran: the call completed
            (void)hipStreamSynchronize(h->up);
            (void)hipStreamSynchronize(h->down);
            (void)hipStreamSynchronize(s);
            (void)hipGetLastError();
            h->pending.clear();
            for (bool& b : h->upbusy) b = false;
            h->evused = 0;
            h->dirty = false;
            ++h->st_aborted;
        }
    } hp_abort;

    // ---- staging for host-pointer mode (hostpipe.h): x (and idx / w) go up first through the pinned ring; the bank follows further
    //      down, behind the spectra kernel -- in chunks interleaved with the render launches where the engine allows it
    const bool bank_dev = dev || (flags & SS_FLAG_BANK_DEVICE) != 0;
    const float* dx = x;
    const float* dbank = bank;
    const int64_t* didx = idx;
    const float* dw = w;
    float* dy = y;
    const size_t bank_bytes = sizeof(float) * (size_t)P * C * L;
    HostPipe& hp = c->pipe;
    const auto host_t0 = std::chrono::steady_clock::now();
    auto mark = [&](int i) { if (!dev) hp.st_mark[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count(); };
    if (!dev) {
        if ((rc = hp_ensure(hp))) return rc;
        if ((rc = hp_begin(hp))) return rc;
        hp_abort.h = &hp;
        hp_abort.s = stream;
        if ((rc = ws_ensure(c, WS_X, sizeof(float) * T))) return rc;
        if (!bank_dev && (rc = ws_ensure(c, WS_BANK, bank_bytes))) return rc;
        if ((rc = ws_ensure(c, WS_Y, sizeof(float) * (size_t)C * T))) return rc;
        if (mode == COEF_EXPLICIT) {
            if ((rc = ws_ensure(c, WS_IDX, sizeof(int64_t) * T))) return rc;
            if ((rc = ws_ensure(c, WS_W, sizeof(float) * T))) return rc;
        }
        mark(0);
        if ((rc = hp_upload(hp, c->ws[WS_X], x, sizeof(float) * T))) return rc;
        dx = (const float*)c->ws[WS_X];
        if (!bank_dev) dbank = (const float*)c->ws[WS_BANK];
        dy = (float*)c->ws[WS_Y];
        if (mode == COEF_EXPLICIT) {
            if ((rc = hp_upload(hp, c->ws[WS_IDX], idx, sizeof(int64_t) * T))) return rc;
            if ((rc = hp_upload(hp, c->ws[WS_W], w, sizeof(float) * T))) return rc;
            didx = (const int64_t*)c->ws[WS_IDX];
            dw = (const float*)c->ws[WS_W];
        }
        hipEvent_t e_in;
        if ((rc = hp_event(hp, &e_in))) return rc;
        HIPCHK(hipEventRecord(e_in, hp.up));
        HIPCHK(hipStreamWaitEvent(stream, e_in, 0));
        mark(1);
    }

    // ---- engine choice
    bool use_os = L > 128;
    if (flags & SS_FLAG_PATH_OS) use_os = true;
    if (flags & SS_FLAG_PATH_DIRECT) use_os = false;
    int geom = c->os_geom;
    if (flags & SS_FLAG_GEOM_2048) geom = 11;
    if (flags & SS_FLAG_GEOM_4096) geom = 12;
    if (flags & SS_FLAG_GEOM_13) geom = 13;
    const bool g13 = use_os && T < ((int64_t)1 << 30) && (int64_t)L * 4 < ((int64_t)1 << 31) && geom == 13;
    const bool g14 = use_os && T < ((int64_t)1 << 30) && (int64_t)L * 4 < ((int64_t)1 << 31) &&
                     (geom == 14 || (flags & SS_FLAG_GEOM_ASM) || geom == 0);   // hand-scheduled assembly engine (k_os13_asm): the default whenever the
                                                                                 // transform engine is (L > 128).  Until round 6 only for L > 4096; it is
                                                                                 // 0.23-0.94 x the B = 2048 engine's time on every shorter shape tried
                                                                                 // (static and moving, T = 16 000 .. 960 000: profiles/r06cc)
    const bool g12 = g13 || g14 || (use_os && T < ((int64_t)1 << 30) && geom == 12);     // 13/14 share 12's block size, spectra and plan
    if (g14 && (rc = load_mod13(c, c->dynq))) return rc;
    if (xdiv && !(g13 || g14)) {       // engines without the fused scaling: divide a copy of x (linear in x)
        if ((rc = ws_ensure(c, WS_W, sizeof(float) * T))) return rc;
        hipLaunchKernelGGL(k_div_by, dim3(grid_for(T)), dim3(256), 0, stream, dx, (float*)c->ws[WS_W], T, xdiv);
        HIPCHK(hipGetLastError());
        dx = (const float*)c->ws[WS_W];
        xdiv = nullptr;
    }
    const int BB = g12 ? B12 : B;
    const int JM = g12 ? JMAX12 : JMAX;
    // assembly engine, EXPERIMENT (SS_HOP_RS = 1 / 2; default 0 = the block grid): input spectra on a grid of B >> rs samples so that a
    // row's first block starts within one hop of its first sample (plan.h row_tasks) -- 8 % fewer blocks and 6 % fewer tasks at config 2,
    // but the kernel gains only 1.5 % (twice the spectra in L2), the spectra kernel loses as much, and the implicit schedule is no longer
    // bit-identical to the explicit one (different block decomposition): profiles/r02r.
    static const int hop_rs_env = knob("SS_HOP_RS") ? atoi(knob("SS_HOP_RS")) : 0;
    const int rs = g14 ? (hop_rs_env < 0 ? 0 : (hop_rs_env > 2 ? 2 : hop_rs_env)) : 0;
    const int M = g14 ? (int)((T + (BB >> rs) - 1) / (BB >> rs)) + (1 << rs) - 1 : (int)((T + BB - 1) / BB);   // number of input spectra
    const int NPart = (L + BB - 1) / BB;

    // ---- schedule -> per-tile min/max of idx
    const int64_t nfine = (T + DTILE - 1) / DTILE;
    if (mode == COEF_SEG) {
        c->seg_start.resize(P);
        int64_t s = 0;
        for (int k = 0; k < P - 1; ++k) {
            if (seg_len_host[k] < 0) return fail(SS_EINVAL, "seg_len[%d] = %lld is negative", k, (long long)seg_len_host[k]);
            c->seg_start[k] = s;
            s += seg_len_host[k];
        }
        c->seg_start[P - 1] = s;
        if (s != T) return fail(SS_EINVAL, "sum(seg_len) = %lld != T = %lld", (long long)s, (long long)T);
    }
    // explicit schedule planned on the device (assembly engine, device pointers): nothing comes back to the host
    const bool dev_plan = mode == COEF_EXPLICIT && g14 && dev && (flags & SS_FLAG_ASYNC_PLAN);
    if (mode == COEF_EXPLICIT) {
        if ((rc = ws_ensure(c, WS_BMIN, sizeof(int32_t) * nfine))) return rc;
        if ((rc = ws_ensure(c, WS_BMAX, sizeof(int32_t) * nfine))) return rc;
        if (!dev_plan) {       // (planned on the device: the tile bounds are formed by the first workgroups of k_front_explicit, below)
            hipLaunchKernelGGL(k_idx_minmax, dim3((unsigned)nfine), dim3(256), 0, stream, didx, T, (int32_t*)c->ws[WS_BMIN],
                               (int32_t*)c->ws[WS_BMAX]);
            HIPCHK(hipGetLastError());
        }
    }
    PlanDevArgs pa;
    memset(&pa, 0, sizeof(pa));
    c->last_dev_planned = dev_plan;
    c->lanes[c->cur_lane].dev_planned = dev_plan;
    int32_t* dplan_out = nullptr;
    if (dev_plan) {
        const int nblk = (int)((T + BB - 1) / BB);
        const int64_t cap64 = 4 * ((int64_t)P + 2 * (int64_t)nblk) + 64;
        if (cap64 * C > ((int64_t)1 << 27)) return fail(SS_EINVAL, "schedule too large for the device planner (use the synchronous explicit path)");
        const int32_t cap_rows = (int32_t)cap64;
        auto al = [](size_t n) { return (n + 255) & ~(size_t)255; };
        const size_t o_lo = 0, o_hi = o_lo + al(4 * (size_t)nblk), o_first = o_hi + al(4 * (size_t)nblk), o_last = o_first + al(4 * (size_t)P),
                     o_rc = o_last + al(4 * (size_t)P), o_rt = o_rc + al(4 * ((size_t)P + 1)), o_keys = o_rt + al(16 * (size_t)cap_rows),
                     o_bins = o_keys + al(8 * (size_t)cap_rows), o_end = o_bins + al(4 * 8 * 4096);
        if ((rc = ws_ensure(c, WS_DPLAN, o_end))) return rc;
        if ((rc = ws_ensure(c, WS_DTASKS, 16 + sizeof(Task) * (size_t)cap_rows * C))) return rc;
        if (!c->async_status) {
            HIPCHK(hipMalloc((void**)&c->async_status, 32));
            HIPCHK(hipMemset(c->async_status, 0, 32));                 // synchronous: ordered against every lane's stream, once per device
        }
        if ((rc = ws_ensure(c, WS_STATUS, 32))) return rc;             // this lane's per-call words
        char* base = (char*)c->ws[WS_DPLAN];
        pa.bmin = (const int32_t*)c->ws[WS_BMIN]; pa.bmax = (const int32_t*)c->ws[WS_BMAX]; pa.nfine = nfine;
        pa.fine_per_block = BB / DTILE; pa.nblk = nblk; pa.P = P; pa.C = C; pa.jmax = JM; pa.NP = NPart; pa.groups = 8; pa.cap_rows = cap_rows; pa.rs = rs;
        pa.lo = (int32_t*)(base + o_lo); pa.hi = (int32_t*)(base + o_hi); pa.first = (int32_t*)(base + o_first); pa.last = (int32_t*)(base + o_last);
        pa.rcount = (int32_t*)(base + o_rc); pa.rtask = (int32_t*)(base + o_rt); pa.keys = (unsigned long long*)(base + o_keys); pa.bins = (int32_t*)(base + o_bins);
        pa.out = dplan_out = (int32_t*)c->ws[WS_DTASKS];
        pa.status = c->async_status;
        pa.call = (int32_t*)c->ws[WS_STATUS];
        pa.status_host = nullptr;
        if (status_out) {
            if (!c->status_pin) HIPCHK(hipHostMalloc((void**)&c->status_pin, 64, hipHostMallocCoherent));     // (fine-grained: the planner's words are polled while the stream runs on)
            c->status_pin[0] = c->status_pin[1] = c->status_pin[2] = -2;       // (-2: the planner has not reported)
            void* dp = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp, c->status_pin, 0));
            pa.status_host = (int32_t*)dp;
        }
        if (!c->ws[WS_STATUS] || !c->lanes[c->cur_lane].status_zeroed) {      // the arrival counter of k_front_explicit (word 7 of the lane's record) starts at zero
            HIPCHK(hipMemsetAsync(c->ws[WS_STATUS], 0, 32, stream));
            c->lanes[c->cur_lane].status_zeroed = true;
        }
        c->plan.tasks[0].clear();
        c->plan.tasks[1].clear();
    } else if (mode == COEF_EXPLICIT) {
        c->bmin.resize(nfine);
        c->bmax.resize(nfine);
        HIPCHK(hipMemcpyAsync(c->bmin.data(), c->ws[WS_BMIN], sizeof(int32_t) * nfine, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(c->bmax.data(), c->ws[WS_BMAX], sizeof(int32_t) * nfine, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int64_t b = 0; b < nfine; ++b)
            if (c->bmin[b] < 0 || c->bmax[b] > P - 2)
                return fail(SS_EINVAL, "interp_index out of range [0, %d] near sample %lld (min %d, max %d)", P - 2,
                            (long long)(b * DTILE), c->bmin[b], c->bmax[b]);
    }
    const bool fast_plan = g12 && mode == COEF_SEG;        // single-launch geometries, implicit schedule: O(P*C) direct planner
    int32_t qmain = 0;                                     // dynamic queues: tasks in the per-XCD part of the list (0 = all)
    // host-pointer mode, assembly engine, implicit schedule: the bank travels in `nchunk` chunks of whole positions and the rows of
    // chunk k are rendered (one launch per chunk, static task lists) while chunk k + 1 is on the wire; the stretch of the output that
    // no later chunk touches travels back at once.  Every row is still one task and y is accumulated with the same two commutative
    // additions per sample onto zeros: same bits as the one-launch render.
    int nchunk = 1;
    if (!dev && !bank_dev && g14 && fast_plan) {
        nchunk = (int)std::min<size_t>(16, bank_bytes / hp.chunk_bytes);
        if (nchunk > P / 2) nchunk = P / 2;
        if (nchunk < 2) nchunk = 1;
    } else if (!dev && bank_dev && g14 && fast_plan && sizeof(float) * (size_t)C * T >= ((size_t)8 << 20)) {
        // a RESIDENT bank rendered for host x / y: nothing to upload, but the 4CT bytes of y take ~3x as long over PCIe as the render itself --
        // four launches in trajectory order let the first quarter of y travel while the other three are rendered (1.13 -> ~0.8 ms at config 2)
        nchunk = P / 2 < 4 ? P / 2 : 4;
        if (nchunk < 2) nchunk = 1;
    }
    const bool chunked = nchunk > 1;
    if (chunked && (rc = load_mod13(c, false))) return rc;
    std::vector<int32_t> chunk_off;
    if (dev_plan) {
    } else if (fast_plan) {
        // XCD-aware task order for the persistent assembly kernel (workgroup b -> XCD b % 8 takes tasks b, b + nwg, ...)
        static const int plan_groups = knob("SS_PLAN_GROUPS") ? atoi(knob("SS_PLAN_GROUPS")) : 8;
        static const int plan_snake = knob("SS_PLAN_SNAKE") ? atoi(knob("SS_PLAN_SNAKE")) : 1;
        static const int plan_tail = knob("SS_PLAN_TAIL") ? atoi(knob("SS_PLAN_TAIL")) : 12;      // % of a range's tasks that go to the shared tail queue
        static const int plan_split = knob("SS_PLAN_SPLIT") ? (g_plan_balanced = atoi(knob("SS_PLAN_SPLIT")) != 0) : 1;
        (void)plan_split;
        static const int plan_pair = knob("SS_PLAN_PAIR") ? atoi(knob("SS_PLAN_PAIR")) : 1;       // tasks of one row adjacent in the queue (plan.h; tuning build: 0 = by their own cost)
        const bool two_level = g14 && c->dynq && plan_groups == 8 && !chunked;
        plan_seg_lpt(c->seg_start, P, C, BB, JM, NPart, c->plan.tasks[0], c->plan_scratch, (g14 && plan_groups > 1 && plan_groups <= 64) ? plan_groups : 1,
                     (plan_snake && !two_level) ? c->num_cu : 0, rs, two_level ? plan_tail : 0, &qmain, plan_pair != 0);
        c->plan.tasks[1].clear();
        if (chunked) {      // stable counting sort of the list by the chunk of the task's row (the LPT / XCD order survives inside a chunk)
            std::vector<Task>& tk = c->plan.tasks[0];
            chunk_off.assign((size_t)nchunk + 1, 0);
            auto chunk_of = [&](int row) {
                int k = (int)(((int64_t)row * nchunk + nchunk - 1) / P);          // smallest k with floor(P k / nchunk) ... refined below
                if (k > nchunk - 1) k = nchunk - 1;
                while (k > 0 && (int)((int64_t)P * k / nchunk) > row) --k;
                while (k + 1 < nchunk && (int)((int64_t)P * (k + 1) / nchunk) <= row) ++k;
                return k;
            };
            for (const Task& t : tk) chunk_off[(size_t)chunk_of(t.row) + 1]++;
            for (int k = 0; k < nchunk; ++k) chunk_off[(size_t)k + 1] += chunk_off[(size_t)k];
            c->chunk_tmp.resize(tk.size());
            std::vector<int32_t> at(chunk_off.begin(), chunk_off.end() - 1);
            for (const Task& t : tk) c->chunk_tmp[(size_t)at[(size_t)chunk_of(t.row)]++] = t;
            tk.swap(c->chunk_tmp);
        }
    } else {
        if (mode == COEF_SEG) seg_minmax(c->seg_start, T, c->bmin, c->bmax);
        if (mode == COEF_FIXED) build_plan_fixed(T, C, use_os ? BB : DTILE, use_os ? JM : 1, c->plan);
        else build_plan(c->bmin, c->bmax, P, C, use_os ? BB / DTILE : 1, use_os ? JM : 1, c->plan);
        if (c->xcd_order) { xcd_interleave(c->plan.tasks[0]); xcd_interleave(c->plan.tasks[1]); }
    }

    // ---- upload plan blob: [seg_start (P int64)][tasks parity 0][tasks parity 1]
    //      geometry 12: ONE list in LPT order (atomic accumulation, persistent workgroups)
    if (g12 && !fast_plan && !dev_plan) {
        if (g14) merge_lpt_xcd(c->plan, NPart, 8, c->merged, c->plan_scratch);     // same XCD-aware order as the implicit schedule's planner
        else merge_lpt(c->plan, NPart, c->merged);
        c->plan.tasks[0].swap(c->merged);
        c->plan.tasks[1].clear();
        if (rs)                                     // these planners work on the block grid: the same tasks in hop units
            for (Task& t : c->plan.tasks[0]) t.j0 <<= rs;
    }
    HRowArgs hrow;
    memset(&hrow, 0, sizeof(hrow));
    bool hrow_done = false;
    if (g14 && !chunked && !dev_plan && rs == 0) {
        const int32_t Pone = P;
        if ((rc = hrow_mark(c, c->plan.tasks[0], &Pone, 1, C, NPart, flags, hrow))) return rc;
        hrow.bank[0] = dbank;
    }
    const size_t n0 = c->plan.tasks[0].size(), n1 = c->plan.tasks[1].size();
    const size_t seg_bytes = 2 * sizeof(int64_t) * (size_t)P;      // [seg_start P x i64][inv_seg P x f64: 1/len(segment k), IEEE double division]
    const size_t blob = seg_bytes + sizeof(Task) * (n0 + n1);
    Pinned* pin;
    if ((rc = pinned_acquire(c, (blob + 15) / 16 * 16, &pin))) return rc;      // k_xspec13 stages whole 16-byte units
    memset(pin->host, 0, seg_bytes);
    if (mode == COEF_SEG) {
        memcpy(pin->host, c->seg_start.data(), sizeof(int64_t) * (size_t)P);
        double* inv = reinterpret_cast<double*>((char*)pin->host + sizeof(int64_t) * (size_t)P);
        for (int k = 0; k + 1 < P; ++k) {
            const int64_t n = c->seg_start[k + 1] - c->seg_start[k];
            inv[k] = n > 0 ? 1.0 / (double)n : 0.0;
        }
    }
    if (n0) memcpy((char*)pin->host + seg_bytes, c->plan.tasks[0].data(), sizeof(Task) * n0);
    if (n1) memcpy((char*)pin->host + seg_bytes + sizeof(Task) * n0, c->plan.tasks[1].data(), sizeof(Task) * n1);
    // The assembly engine reads its plan (28 KB of task descriptors fetched a task ahead by scalar loads + the segment table)
    // straight from the pinned, device-mapped host buffer: no in-stream DMA copy ahead of the kernels (its start-up latency
    // would sit on the critical path of every render).  Other geometries take the copy.
    const bool xspec_stages_plan = g14 && c->plan_mode == 2;     // default: the spectra kernel moves the plan from pinned host memory to HBM
    const bool zero_copy_plan = g14 && c->plan_mode == 1;        // (round 1-2 default, now a tuning knob: the render kernel reads the pinned buffer in place)
    const size_t blob16 = (blob + 15) / 16;
    const char* plan_base;
    if (xspec_stages_plan) {
        if ((rc = ws_ensure(c, WS_PLAN, blob16 * 16))) return rc;
        plan_base = (const char*)c->ws[WS_PLAN];
    } else if (zero_copy_plan) {
        plan_base = (const char*)pin->host;
    } else {
        if ((rc = ws_ensure(c, WS_PLAN, blob))) return rc;
        HIPCHK(hipMemcpyAsync(c->ws[WS_PLAN], pin->host, blob, hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(pin->ev, stream));
        pin->pending = true;
        plan_base = (const char*)c->ws[WS_PLAN];
    }

    mark(2);
    int qgroups = 0, qinit = 0;
    const char* trace_env = knob("SS_TRACE_FILE");      // timeline trace of a code object built with OS13_OPT=trace (tools/)
    RenderParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.x = dx; prm.T = T; prm.bank = dbank; prm.P = P; prm.C = C; prm.L = L; prm.NP = NPart;
    prm.M = M; prm.consts = g12 ? c->consts12 : c->consts; prm.mode = mode;
    prm.seg_start = (const int64_t*)plan_base;
    prm.idx = didx; prm.w = dw; prm.y = dy;

    if (use_os) {
        if ((rc = ws_ensure(c, WS_XS, sizeof(c32) * (size_t)(M + 1) * BB))) return rc;
        prm.Xs = (const c32*)c->ws[WS_XS];
        {
            ProfScope ps(c, stream, 1);
            if ((g13 || g14) && (rc = ws_ensure(c, WS_CNT, trace_env ? 512 * 1024 : 16 * 64))) return rc;     // queue heads: 8 XCD queues + the shared tail queue, 64 bytes apart
            if (g14 && trace_env) HIPCHK(hipMemsetAsync(c->ws[WS_CNT], 0, 512 * 1024, stream));   // stamps / trace records land behind the queue heads
            if (g14) {      // dynamic task queues of the assembly kernel: one head per XCD (workgroup b runs on XCD b % 8), preloaded
                            // with the tasks the workgroups start on
                const int nwg = (int)(n0 > (size_t)c->num_cu ? (size_t)c->num_cu : n0);
                qgroups = !c->dynq ? 0 : ((nwg >= 8 && nwg % 8 == 0) ? 8 : 1);
                qinit = 0;                                   // every task, the first one included, comes from the queue
            }
            if (g13 || g14) {
                if (dev_plan) {
                    const int n_mm = (int)((nfine + FRONT_MMT - 1) / FRONT_MMT);
                    static const int front_fused = knob("SS_FRONT_FUSED") ? atoi(knob("SS_FRONT_FUSED")) : 1;
                    if (front_fused) {        // the planner as one more workgroup of this launch, waiting for the tile-bound workgroups' arrivals (default since the
                                              // hand-off needs no release fence: write-through stores + agent-scope loads; SS_FRONT_FUSED=0 in the tuning build: two launches)
                        hipLaunchKernelGGL(k_front_explicit<true>, dim3((unsigned)(n_mm + 1 + M + 1)), dim3(NT13), 0, stream, dx, T, (const c32*)c->consts13, (c32*)c->ws[WS_XS], M,
                                           dy, (int64_t)C * T, qgroups ? (int*)c->ws[WS_CNT] : (int*)nullptr, qgroups + 1, qinit, xdiv, didx,
                                           (int32_t*)c->ws[WS_BMIN], (int32_t*)c->ws[WS_BMAX], n_mm, (int*)c->ws[WS_STATUS] + 7, pa);
                    } else {
                        // the tile bounds ride on the input-spectra launch (its first n_mm workgroups: one wave per tile); the planner follows as its
                        // own one-workgroup launch of 1024 threads -- one kernel and one boundary less than k_idx_minmax + k_plan_explicit + k_xspec13
                        hipLaunchKernelGGL(k_front_explicit<false>, dim3((unsigned)(n_mm + M + 1)), dim3(NT13), 0, stream, dx, T, (const c32*)c->consts13, (c32*)c->ws[WS_XS], M,
                                           dy, (int64_t)C * T, qgroups ? (int*)c->ws[WS_CNT] : (int*)nullptr, qgroups + 1, qinit, xdiv, didx,
                                           (int32_t*)c->ws[WS_BMIN], (int32_t*)c->ws[WS_BMAX], n_mm, (int*)c->ws[WS_STATUS] + 7, pa);
                        hipLaunchKernelGGL(k_plan_explicit, dim3(1), dim3(1024), 0, stream, pa);
                    }
                } else {
                const bool rows_ride = hrow.nrows > 0 && bank_dev;        // a resident bank: its row spectra are formed by this launch's first workgroups
                if (rows_ride) { hrow_fill(c, hrow, C, L, NPart); }
                const int nrow_wg = rows_ride ? hrow_workgroups(hrow) : 0;
                if (rows_ride) hrow_done = true;
                hipLaunchKernelGGL(k_xspec13, dim3(M + 1 + nrow_wg), dim3(NT13), 0, stream, dx, T, (const c32*)c->consts13, (c32*)c->ws[WS_XS], M,
                                   // a static source on the assembly engine is STORED (one task per channel and output block): no zero fill
                                   (knob("SS_NO_ZFILL") /* (tuning build: results WRONG) */ || (g14 && mode == COEF_FIXED)) ? (float*)nullptr : dy,
                                   (int64_t)C * T, (g13 || qgroups) ? (int*)c->ws[WS_CNT] : (int*)nullptr, g13 ? 1 : qgroups + 1,
                                   g13 ? 0 : qinit, xdiv, rs, xspec_stages_plan ? (const uint4*)pin->host : (const uint4*)nullptr,
                                   xspec_stages_plan ? (uint4*)c->ws[WS_PLAN] : (uint4*)nullptr, xspec_stages_plan ? (int)blob16 : 0,
                                   (const int32_t*)nullptr /* (the NaN fill on a failed plan moved into the render kernel's prologue) */, hrow, nrow_wg);
                }
                if (xspec_stages_plan) {       // the ring slot may be rewritten once the spectra kernel has consumed it
                    HIPCHK(hipEventRecord(pin->ev, stream));
                    pin->pending = true;
                }
            }
            else if (g12) hipLaunchKernelGGL(k_xspec12, dim3(M + 1), dim3(NT12), 0, stream, dx, T, (const c32*)c->consts12, (c32*)c->ws[WS_XS], M,
                                        dy, (int64_t)C * T, (int*)nullptr);
            else hipLaunchKernelGGL(k_xspec, dim3(M + 1), dim3(NT), 0, stream, dx, T, (const c32*)c->consts, (c32*)c->ws[WS_XS], M);
        }
        HIPCHK(hipGetLastError());
    }
    const Task* dtasks = (const Task*)(plan_base + seg_bytes);
    mark(3);
    if (chunked) {
        const size_t pos_bytes = sizeof(float) * (size_t)C * L;      // one trajectory position of the bank
        int64_t done = 0;
        int k = 0;                                                   // next chunk to launch
        auto launch_ready = [&](size_t up_bytes) -> int {            // every chunk whose last byte is on its way: launch behind the transfer
            int rc2;
            while (k < nchunk) {
                const int r0 = (int)((int64_t)P * k / nchunk), r1 = (int)((int64_t)P * (k + 1) / nchunk);
                if (pos_bytes * (size_t)r1 > up_bytes) break;
                (void)r0;
                if (!bank_dev) {
                    hipEvent_t e_up;
                    if ((rc2 = hp_event(hp, &e_up))) return rc2;
                    HIPCHK(hipEventRecord(e_up, hp.up));
                    HIPCHK(hipStreamWaitEvent(stream, e_up, 0));
                }
                const int32_t nt = chunk_off[(size_t)k + 1] - chunk_off[(size_t)k];
                if (nt > 0) {
                    ProfScope ps(c, stream, 0);
                    Os13AsmArgs a;
                    memset(&a, 0, sizeof(a));
                    a.bank = dbank; a.Xs = prm.Xs; a.tasks = dtasks + chunk_off[(size_t)k];
                    a.seg_start = plan_base;
                    a.inv_seg = plan_base + sizeof(int64_t) * (size_t)P;
                    a.y = dy; a.T = T; a.P = P; a.C = C; a.L = L; a.NP = NPart; a.M = M;
                    a.ntasks = nt; a.mode = mode; a.nwg = nt < c->num_cu ? nt : c->num_cu;
                    a.consts = c->consts14; a.counter = nullptr;
                    a.qgroups = 0;
                    a.rs = rs;
                    size_t asz = sizeof(a);
                    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
                    HIPCHK(hipModuleLaunchKernel(c->fn13, (unsigned)a.nwg, 1, 1, NT13, 1, 1, 0, stream, nullptr, cfg));
                }
                // samples before the segment that ends at row r1 - 1 have both their rows in chunks <= k: they are final
                const int64_t fin = k == nchunk - 1 ? T : c->seg_start[(size_t)(r1 - 1)];
                if (fin > done) {
                    hipEvent_t e_done;
                    if ((rc2 = hp_event(hp, &e_done))) return rc2;
                    HIPCHK(hipEventRecord(e_done, stream));
                    HIPCHK(hipStreamWaitEvent(hp.down, e_done, 0));
                    if ((rc2 = hp_download(hp, y + done, sizeof(float) * (size_t)T, dy + done, sizeof(float) * (size_t)T, sizeof(float) * (size_t)(fin - done), C)))
                        return rc2;
                    done = fin;
                }
                ++hp.st_chunks;
                ++k;
            }
            return SS_OK;
        };
        if (bank_dev) {
            if ((rc = launch_ready((size_t)-1))) return rc;          // every chunk is "uploaded": launched back to back, y slices follow each
        } else if ((rc = hp_upload(hp, c->ws[WS_BANK], bank, bank_bytes, launch_ready))) return rc;
        mark(4); mark(5);
        if ((rc = hp_finish(hp))) return rc;
        mark(6);
        HIPCHK(hipStreamSynchronize(stream));
        hp.st_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count();
        return SS_OK;
    }
    if (!dev && !bank_dev) {       // the whole bank behind the spectra kernel, pipelined through the pinned ring
        if ((rc = hp_upload(hp, c->ws[WS_BANK], bank, bank_bytes))) return rc;
        hipEvent_t e_up;
        if ((rc = hp_event(hp, &e_up))) return rc;
        HIPCHK(hipEventRecord(e_up, hp.up));
        HIPCHK(hipStreamWaitEvent(stream, e_up, 0));
        hp.st_chunks = 1;
    }
    if (!hrow_done && (rc = hrow_launch(c, hrow, C, L, NPart, stream))) return rc;
    for (int parity = 0; parity < 2; ++parity) {
        size_t nt = parity ? n1 : n0;
        if (dev_plan) nt = parity ? 0 : (size_t)c->num_cu;      // the count lives in the list's header: every CU gets a workgroup
        if (!nt) continue;
        prm.tasks = dev_plan ? (const Task*)dplan_out : dtasks + (parity ? n0 : 0);
        prm.ntasks = dev_plan ? -1 : (int32_t)nt;
        prm.accumulate = g12 ? 2 : parity;
        if (g12 && nt > (size_t)c->num_cu) nt = (size_t)c->num_cu;     // persistent: one workgroup per CU
        ProfScope ps(c, stream, use_os ? 0 : 2);
        if (g14) {
            Os13AsmArgs a;
            memset(&a, 0, sizeof(a));
            a.bank = dbank; a.Xs = prm.Xs; a.tasks = prm.tasks;
            a.seg_start = plan_base;
            a.inv_seg = plan_base + sizeof(int64_t) * (size_t)P;
            a.y = dy; a.T = T; a.P = P; a.C = C; a.L = L; a.NP = NPart; a.M = M;
            a.ntasks = prm.ntasks; a.mode = mode; a.nwg = (int32_t)nt;
            a.consts = c->consts14; a.counter = qgroups ? c->ws[WS_CNT] : nullptr;
            a.idx = didx; a.w = dw;
            a.hspec = hrow.nrows > 0 ? c->ws[WS_HS] : nullptr;
            a.verdict = dev_plan ? c->ws[WS_STATUS] : nullptr;
            a.qgroups = qgroups;
            a.rs = rs | ((qgroups == 8 && qmain > 0 && qmain < (1 << 22) && (size_t)qmain < n0) ? qmain << 8 : 0);   // bits 8..: the queue split
            const char* trace_file = trace_env;
            if (trace_file) a.counter = c->ws[WS_CNT];     // (zeroed ahead of the spectra kernel, which then sets the queue heads)
            size_t asz = sizeof(a);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
            HIPCHK(hipModuleLaunchKernel(c->dynq ? c->fn13q : c->fn13, (unsigned)nt, 1, 1, NT13, 1, 1, 0, stream, nullptr, cfg));
            if (zero_copy_plan) {          // the ring slot may be rewritten only after this kernel has consumed it
                HIPCHK(hipEventRecord(pin->ev, stream));
                pin->pending = true;
            }
            if (trace_file) {
                std::vector<char> hb(512 * 1024);
                HIPCHK(hipMemcpyAsync(hb.data(), c->ws[WS_CNT], hb.size(), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if (FILE* f = fopen(trace_file, "wb")) { fwrite(hb.data(), 1, hb.size(), f); fclose(f); }
            }
        }
        else if (g13) {
            Params13 p13;
            p13.r = prm;
            p13.r.consts = c->consts13;
            p13.counter = (int*)c->ws[WS_CNT];
            p13.nwg = (int32_t)nt;
            hipLaunchKernelGGL(k_os13, dim3((unsigned)nt), dim3(NT13), 0, stream, p13);
        }
        else if (!use_os) hipLaunchKernelGGL(k_direct, dim3((unsigned)nt), dim3(NT), 0, stream, prm);
#ifdef SS_ABLATE      // profiling-only ablation ladder of the HIP engines (results are WRONG when a mask is set): tools/ablate.sh builds with -DSS_ABLATE
        else if (g12 && c->os_ablate == 1) hipLaunchKernelGGL(k_os12<1>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 2) hipLaunchKernelGGL(k_os12<2>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 4) hipLaunchKernelGGL(k_os12<4>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 8) hipLaunchKernelGGL(k_os12<8>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 16) hipLaunchKernelGGL(k_os12<16>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 32) hipLaunchKernelGGL(k_os12<32>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 48) hipLaunchKernelGGL(k_os12<48>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 56) hipLaunchKernelGGL(k_os12<56>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 63) hipLaunchKernelGGL(k_os12<63>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
        else if (g12 && c->os_ablate == 7) hipLaunchKernelGGL(k_os12<7>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
#endif
        else if (g12) hipLaunchKernelGGL(k_os12<0>, dim3((unsigned)nt), dim3(NT12), 0, stream, prm);
#ifdef SS_ABLATE
        else if (c->os_ablate == 1) hipLaunchKernelGGL((k_os<3, 1>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_ablate == 2) hipLaunchKernelGGL((k_os<3, 2>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_ablate == 4) hipLaunchKernelGGL((k_os<3, 4>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_ablate == 3) hipLaunchKernelGGL((k_os<3, 3>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_ablate == 6) hipLaunchKernelGGL((k_os<3, 6>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_ablate == 7) hipLaunchKernelGGL((k_os<3, 7>), dim3((unsigned)nt), dim3(NT), 0, stream, prm);
#endif
        else if (c->os_variant == 0) hipLaunchKernelGGL(k_os<0>, dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else if (c->os_variant == 2) hipLaunchKernelGGL(k_os<2>, dim3((unsigned)nt), dim3(NT), 0, stream, prm);
        else hipLaunchKernelGGL(k_os<3>, dim3((unsigned)nt), dim3(NT), 0, stream, prm);
    }
    HIPCHK(hipGetLastError());

    if (status_out) {
        status_out[0] = -1; status_out[1] = 0; status_out[2] = 0;
        if (dev_plan && c->async_status) {
            // The planner mirrors its verdict into pinned host words ~25 us after the front launch starts; the call returns then and the render kernel
            // runs on -- device output is ordered by the stream as in every other entry point (round 5 synchronised the stream here: 0.223 ms per
            // validating call against 0.199 asynchronous).  Host output (below) waits for the render anyway.
            volatile int32_t* sp = c->status_pin;
            if (sp) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int spin = 0; sp[0] == -2; ++spin) {
                    if ((spin & 63) == 63) {
                        if (hipStreamQuery(stream) == hipSuccess) break;            // the stream has drained: whatever was written is there
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { HIPCHK(hipStreamSynchronize(stream)); break; }
                    }
                    __builtin_ia32_pause();
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            } else {
                HIPCHK(hipStreamSynchronize(stream));
            }
            if (sp && sp[0] != -2) {                // the planner's pinned mirror (no copy to wait for)
                status_out[0] = sp[0]; status_out[1] = (int64_t)sp[1] * DTILE; status_out[2] = sp[2];
            } else {
                int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                HIPCHK(hipMemcpy(h, c->ws[WS_STATUS], 32, hipMemcpyDeviceToHost));
                status_out[0] = h[3]; status_out[1] = (int64_t)h[4] * DTILE; status_out[2] = h[2];
            }
        }
    }
    if (!dev) {
        hipEvent_t e_done;
        if ((rc = hp_event(hp, &e_done))) return rc;
        HIPCHK(hipEventRecord(e_done, stream));
        HIPCHK(hipStreamWaitEvent(hp.down, e_done, 0));
        mark(4);
        if ((rc = hp_download(hp, y, 0, dy, 0, sizeof(float) * (size_t)C * T, 1))) return rc;
        mark(5);
        if ((rc = hp_finish(hp))) return rc;
        mark(6);
        HIPCHK(hipStreamSynchronize(stream));
        hp.st_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - host_t0).count();
    }
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// ONE persistent launch for all the renders of a scene (SonicSet.py:61-94: three moving speakers + two static sources): one spectra
// launch (grid = spectra x sources), one task list over all sources (plan_scene_lpt), one k_os13_asm launch whose tasks look their
// source up in the argument table.  Saves, per extra source, a spectra launch, two kernel boundaries and an LPT tail.
int render_scene(int nsrc, const float* const* xs, int64_t T, const float* const* banks, const int32_t* Ps, int32_t C, int32_t L,
                 const int64_t* const* seg_lens, const float* const* divisors, float* const* ys, uint32_t flags, void* stream_) {
    if (nsrc < 1 || nsrc > 8) return fail(SS_EINVAL, "a scene launch takes 1..8 sources (got %d)", nsrc);
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "the scene launch takes device pointers (SS_FLAG_DEVICE_PTR)");
    if (T < 1 || C < 1 || C > 65535 || L < 1) return fail(SS_EINVAL, "bad shape: T=%lld C=%d L=%d", (long long)T, C, L);
    if (!(T < ((int64_t)1 << 30) && (int64_t)L * 4 < ((int64_t)1 << 31) && L > 128))
        return fail(SS_EINVAL, "the scene launch exists for the assembly engine's shapes (L > 128, T < 2^30): T=%lld L=%d", (long long)T, L);
    if (!xs || !banks || !Ps || !ys) return fail(SS_EINVAL, "NULL argument");
    for (int s = 0; s < nsrc; ++s) {
        if (!xs[s] || !banks[s] || !ys[s] || Ps[s] < 1) return fail(SS_EINVAL, "source %d: NULL pointer or P < 1", s);
        if (Ps[s] > 1 && (!seg_lens || !seg_lens[s])) return fail(SS_EINVAL, "source %d: a moving source (P = %d) needs its segment lengths", s, Ps[s]);
    }
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    if ((rc = load_mod13(c, c->dynq))) return rc;
    c->last_dev_planned = false;
    c->lanes[c->cur_lane].dev_planned = false;
    const int M = (int)((T + B12 - 1) / B12);
    const int NPart = (L + B12 - 1) / B12;
    // ---- segment tables (host) + task list
    static thread_local std::vector<std::vector<int64_t>> segs;
    segs.resize((size_t)nsrc);
    SceneSrc ssrc[8];
    size_t seg_off[8], seg_total = 0;
    for (int s = 0; s < nsrc; ++s) {
        const int P = Ps[s];
        auto& st = segs[(size_t)s];
        st.assign((size_t)P, 0);
        if (P > 1) {
            int64_t acc = 0;
            for (int k = 0; k < P - 1; ++k) {
                if (seg_lens[s][k] < 0) return fail(SS_EINVAL, "source %d: seg_len[%d] = %lld is negative", s, k, (long long)seg_lens[s][k]);
                st[(size_t)k] = acc;
                acc += seg_lens[s][k];
            }
            st[(size_t)P - 1] = acc;
            if (acc != T) return fail(SS_EINVAL, "source %d: sum(seg_len) = %lld != T = %lld", s, (long long)acc, (long long)T);
        }
        ssrc[s].seg_start = st.data();
        ssrc[s].P = P;
        seg_off[s] = seg_total;
        seg_total += 2 * sizeof(int64_t) * (size_t)P;
    }
    int32_t qmain = 0;
    static const int plan_tail = knob("SS_PLAN_TAIL") ? atoi(knob("SS_PLAN_TAIL")) : 12;
    plan_scene_lpt(ssrc, nsrc, T, C, B12, JMAX12, NPart, c->plan.tasks[0], 8, c->dynq ? plan_tail : 0, &qmain);
    c->plan.tasks[1].clear();
    HRowArgs hrow;
    memset(&hrow, 0, sizeof(hrow));
    if ((rc = hrow_mark(c, c->plan.tasks[0], Ps, nsrc, C, NPart, flags, hrow))) return rc;      // the static sources' single rows, long rows of few-point paths
    for (int s = 0; s < nsrc; ++s) hrow.bank[s] = banks[s];
    const size_t n0 = c->plan.tasks[0].size();
    const size_t blob = seg_total + sizeof(Task) * n0, blob16 = (blob + 15) / 16;
    Pinned* pin;
    if ((rc = pinned_acquire(c, blob16 * 16, &pin))) return rc;
    memset(pin->host, 0, seg_total);
    for (int s = 0; s < nsrc; ++s) {
        const int P = Ps[s];
        char* base = (char*)pin->host + seg_off[s];
        memcpy(base, segs[(size_t)s].data(), sizeof(int64_t) * (size_t)P);
        double* inv = reinterpret_cast<double*>(base + sizeof(int64_t) * (size_t)P);
        for (int k = 0; k + 1 < P; ++k) {
            const int64_t n = segs[(size_t)s][(size_t)k + 1] - segs[(size_t)s][(size_t)k];
            inv[k] = n > 0 ? 1.0 / (double)n : 0.0;
        }
    }
    memcpy((char*)pin->host + seg_total, c->plan.tasks[0].data(), sizeof(Task) * n0);
    if ((rc = ws_ensure(c, WS_PLAN, blob16 * 16))) return rc;
    const char* plan_base = (const char*)c->ws[WS_PLAN];
    const size_t xs_one = sizeof(c32) * (size_t)(M + 1) * B12;
    if ((rc = ws_ensure(c, WS_XS, xs_one * (size_t)nsrc))) return rc;
    if ((rc = ws_ensure(c, WS_CNT, 16 * 64))) return rc;
    const int nwg = (int)(n0 > (size_t)c->num_cu ? (size_t)c->num_cu : n0);
    const int qgroups = !c->dynq ? 0 : ((nwg >= 8 && nwg % 8 == 0) ? 8 : 1);
    {
        ProfScope ps(c, stream, 1);
        XspecSrcTab tab;
        memset(&tab, 0, sizeof(tab));
        for (int s = 0; s < nsrc; ++s) {
            tab.x[s] = xs[s];
            tab.xdiv[s] = divisors ? divisors[s] : nullptr;
            tab.Xs[s] = (c32*)((char*)c->ws[WS_XS] + xs_one * (size_t)s);
            tab.y[s] = Ps[s] > 1 ? ys[s] : nullptr;       // static sources are stored by the render kernel, not added onto zeros
        }
        if (hrow.nrows > 0) hrow_fill(c, hrow, C, L, NPart);
        const int nrow_wg = hrow_workgroups(hrow);
        hipLaunchKernelGGL(k_xspec13_multi, dim3((unsigned)(nrow_wg + (M + 1) * nsrc)), dim3(NT13), 0, stream, tab, T, (const c32*)c->consts13, M, (int64_t)C * T,
                           qgroups ? (int*)c->ws[WS_CNT] : (int*)nullptr, qgroups + 1, 0, (const uint4*)pin->host, (uint4*)c->ws[WS_PLAN], (int)blob16,
                           hrow, nrow_wg);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(pin->ev, stream));
        pin->pending = true;
    }
    if (!n0) return SS_OK;
    {
        ProfScope ps(c, stream, 0);
        Os13AsmArgs a;
        memset(&a, 0, sizeof(a));
        a.hspec = hrow.nrows > 0 ? c->ws[WS_HS] : nullptr;
        for (int s = 0; s < nsrc; ++s) {
            Os13AsmArgs::Src& e = a.src[s];
            e.bank = banks[s];
            e.Xs = (const char*)c->ws[WS_XS] + xs_one * (size_t)s;
            e.seg_start = plan_base + seg_off[s];
            e.inv_seg = plan_base + seg_off[s] + sizeof(int64_t) * (size_t)Ps[s];
            e.y = ys[s];
            e.P = Ps[s]; e.C = C; e.mode = Ps[s] > 1 ? COEF_SEG : COEF_FIXED; e.nwg = nwg;
        }
        a.bank = a.src[0].bank; a.Xs = a.src[0].Xs; a.seg_start = a.src[0].seg_start; a.inv_seg = a.src[0].inv_seg; a.y = a.src[0].y;
        a.P = a.src[0].P; a.mode = a.src[0].mode;
        a.tasks = plan_base + seg_total;
        a.T = T; a.C = C; a.L = L; a.NP = NPart; a.M = M; a.ntasks = (int32_t)n0; a.nwg = nwg;
        a.consts = c->consts14; a.counter = qgroups ? c->ws[WS_CNT] : nullptr;
        a.qgroups = qgroups;
        a.rs = (qgroups == 8 && qmain > 0 && qmain < (1 << 22) && (size_t)qmain < n0) ? qmain << 8 : 0;
        a.nsrc = nsrc;
        size_t asz = sizeof(a);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        HIPCHK(hipModuleLaunchKernel(c->dynq ? c->fn13q : c->fn13, (unsigned)nwg, 1, 1, NT13, 1, 1, 0, stream, nullptr, cfg));
    }
    return SS_OK;
}

// stage a host array to a workspace slot (or pass a device pointer through)
int stage_in(Ctx* c, int slot, const void* p, size_t bytes, bool dev, hipStream_t s, const void** out) {
    if (dev) { *out = p; return SS_OK; }
    int rc = ws_ensure(c, slot, bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->ws[slot], p, bytes, hipMemcpyHostToDevice, s));
    *out = c->ws[slot];
    return SS_OK;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int ss_version(void) { return SS_VERSION; }
const char* ss_last_error(void) { return g_err.c_str(); }

int ss_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(SS_ENODEV, "no HIP device available (%s); the HIP path has no CPU fallback", hipGetErrorString(e));
    if (device >= 0) {
        if (device >= n) return fail(SS_EINVAL, "device %d out of range (%d devices)", device, n);
        HIPCHK(hipSetDevice(device));
    }
    Ctx* c;
    return get_ctx(&c);
}

int ss_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_ctx) {
        Ctx* c = kv.second;
        hipSetDevice(c->device);
        hipDeviceSynchronize();
        if (c->consts) hipFree(c->consts);
        if (c->consts12) hipFree(c->consts12);
        if (c->consts13) hipFree(c->consts13);
        if (c->consts14) hipFree(c->consts14);
        if (c->mod13) hipModuleUnload(c->mod13);
        if (c->mod13q) hipModuleUnload(c->mod13q);
        hp_destroy(c->pipe);
        for (int i = 0; i < WS_COUNT; ++i) if (c->ws[i]) hipFree(c->ws[i]);
        for (int l = 0; l < Ctx::NLANE; ++l)
            if (l != c->cur_lane)
                for (int i = 0; i < WS_COUNT; ++i) if (i != WS_K1 && c->lanes[l].ws[i]) hipFree(c->lanes[l].ws[i]);
        if (c->async_status) hipFree(c->async_status);
        if (c->status_pin) hipHostFree(c->status_pin);
        for (auto& p : c->ring) { if (p.host) hipHostFree(p.host); if (p.ev) hipEventDestroy(p.ev); }
        for (auto& e : c->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        for (auto& e : c->ev_pool) hipEventDestroy(e);
        delete c;
    }
    g_ctx.clear();
    return SS_OK;
}

int ss_convolve_moving_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L, const int64_t* idx,
                           const float* w, float* y, uint32_t flags, void* stream) {
    return render(COEF_EXPLICIT, x, T, rirs, P, C, L, nullptr, idx, w, y, flags, stream);
}

// ---------------------------------------------------------------------------------------------
// streaming render with persistent state (stream13.h)
struct SsStream {
    int device = -1;
    StreamDev d{};
    std::vector<int64_t> seg_start;      // [P], last == total
    int64_t total = 0, pos = 0;
    int k = 0;                           // current segment (monotone)
    int slot_row[STREAM_ROW_SLOTS] = {-1, -1, -1, -1};
    int64_t pushes = 0, pieces = 0, rows_prepared = 0;
};

static int stream_prepare_row(SsStream* st, int row, hipStream_t stream) {
    if (row < 0 || row >= st->d.P) return SS_OK;
    int& have = st->slot_row[row & (STREAM_ROW_SLOTS - 1)];
    if (have == row) return SS_OK;
    hipLaunchKernelGGL(k_stream_rows, dim3((unsigned)st->d.NP, (unsigned)st->d.C), dim3(NT13), 0, stream, st->d, row);
    HIPCHK(hipGetLastError());
    have = row;
    ++st->rows_prepared;
    return SS_OK;
}

int ss_stream_open(void** handle, const float* rirs, int32_t P, int32_t C, int32_t L, const int64_t* seg_len, uint32_t flags, void* stream_) {
    if (!handle || !rirs || !seg_len) return fail(SS_EINVAL, "ss_stream_open: NULL argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "ss_stream_open takes a DEVICE bank (SS_FLAG_DEVICE_PTR): the persistent state lives in HBM");
    if (P < 2 || C < 1 || C > 65535 || L < 1) return fail(SS_EINVAL, "bad shape: P=%d C=%d L=%d (a moving source has at least 2 positions)", P, C, L);
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    SsStream* st = new SsStream();
    st->device = c->device;
    st->seg_start.resize((size_t)P);
    int64_t acc = 0;
    for (int k = 0; k < P - 1; ++k) {
        if (seg_len[k] < 0) { delete st; return fail(SS_EINVAL, "seg_len[%d] = %lld is negative", k, (long long)seg_len[k]); }
        st->seg_start[(size_t)k] = acc;
        acc += seg_len[k];
    }
    st->seg_start[(size_t)P - 1] = acc;
    st->total = acc;
    StreamDev& d = st->d;
    d.bank = rirs; d.P = P; d.C = C; d.L = L;
    d.NP = (L + B13 - 1) / B13;
    d.NR = d.NP + 1;
    d.consts = c->consts13;
    const size_t hs = sizeof(c32) * (size_t)STREAM_ROW_SLOTS * C * d.NP * B13, xr = sizeof(c32) * (size_t)d.NR * B13;
    hipError_t e = hipMalloc((void**)&d.Hs, hs);
    if (e == hipSuccess) e = hipMalloc((void**)&d.Xr, xr);
    if (e == hipSuccess) e = hipMalloc((void**)&d.xh, sizeof(float) * (size_t)(acc > 0 ? acc : 1));
    if (e != hipSuccess) {
        if (d.Hs) hipFree(d.Hs);
        if (d.Xr) hipFree(d.Xr);
        delete st;
        return fail(SS_ENOMEM, "ss_stream_open: %s", hipGetErrorString(e));
    }
    // the first segment's two rows, and the row after them, are transformed now; every later row one whole segment ahead of its use
    for (int r = 0; r < 3; ++r)
        if ((rc = stream_prepare_row(st, r, stream))) {
            hipFree(d.Hs); hipFree(d.Xr); hipFree(d.xh);
            delete st;
            return rc;
        }
    *handle = st;
    return SS_OK;
}

int ss_stream_push(void* handle, const float* chunk, int64_t n, float* out, uint32_t flags, void* stream_) {
    SsStream* st = static_cast<SsStream*>(handle);
    if (!st || (n > 0 && (!chunk || !out))) return fail(SS_EINVAL, "ss_stream_push: NULL argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "ss_stream_push takes device pointers (SS_FLAG_DEVICE_PTR)");
    if (n < 0 || st->pos + n > st->total) return fail(SS_EINVAL, "more input (%lld + %lld samples) than the trajectory schedule covers (%lld)",
                                                       (long long)st->pos, (long long)n, (long long)st->total);
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    if (c->device != st->device) return fail(SS_EINVAL, "the stream was opened on device %d, the current device is %d", st->device, c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    ++st->pushes;
    int64_t left = n, off = 0;
    while (left > 0) {
        int64_t len = 0;
        if (!stream_next_piece(st->seg_start.data(), st->d.P, st->pos, left, st->k, len)) return fail(SS_EINVAL, "push past the end of the schedule");
        for (int r = st->k; r <= st->k + 2; ++r)       // k, k + 1: needed now (prepared a segment ago unless the stream just started / jumped
            if ((rc = stream_prepare_row(st, r, stream))) return rc;     // over empty segments); k + 2: for the next segment
        StreamPiece pc;
        pc.pos = st->pos; pc.n = (int32_t)len; pc.j = (int32_t)(st->pos / B13); pc.k = st->k;
        pc.seg_start = st->seg_start[(size_t)st->k];
        const int64_t nk = st->seg_start[(size_t)st->k + 1] - st->seg_start[(size_t)st->k];
        pc.inv_len = nk > 0 ? 1.0 / (double)nk : 0.0;
        pc.chunk = chunk + off; pc.out = out; pc.out_stride = n; pc.out_off = off;
        hipLaunchKernelGGL(k_stream_push, dim3((unsigned)st->d.C), dim3(NT13), 0, stream, st->d, pc);
        HIPCHK(hipGetLastError());
        ++st->pieces;
        st->pos += len; off += len; left -= len;
    }
    return SS_OK;
}

int ss_stream_info(void* handle, int64_t* out, int32_t n) {
    SsStream* st = static_cast<SsStream*>(handle);
    if (!st || !out) return fail(SS_EINVAL, "ss_stream_info: NULL argument");
    const int64_t v[6] = {st->pos, st->total, st->pushes, st->pieces, st->rows_prepared,
                          (int64_t)(sizeof(c32) * ((size_t)STREAM_ROW_SLOTS * st->d.C * st->d.NP + (size_t)st->d.NR) * B13 + sizeof(float) * (size_t)st->total)};
    for (int i = 0; i < n && i < 6; ++i) out[i] = v[i];
    return SS_OK;
}

int ss_stream_close(void* handle) {
    SsStream* st = static_cast<SsStream*>(handle);
    if (!st) return SS_OK;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != st->device) (void)hipSetDevice(st->device);
    if (st->d.Hs) hipFree(st->d.Hs);          // (hipFree synchronises the device: no kernel of the stream is still reading)
    if (st->d.Xr) hipFree(st->d.Xr);
    if (st->d.xh) hipFree(st->d.xh);
    if (cur >= 0 && cur != st->device) (void)hipSetDevice(cur);
    delete st;
    return SS_OK;
}

int ss_set_host_pipe(int threads, int64_t slot_bytes, int64_t chunk_bytes, int bind) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    HostPipe& h = c->pipe;
    if (threads > 256 || (slot_bytes > 0 && (slot_bytes < (1 << 16) || slot_bytes > ((int64_t)1 << 30))) || (chunk_bytes > 0 && chunk_bytes < (1 << 20)))
        return fail(SS_EINVAL, "ss_set_host_pipe: threads <= 256, 64 KiB <= slot_bytes <= 1 GiB, chunk_bytes >= 1 MiB (0 / negative = keep)");
    const bool rebind = bind >= 0 && bind <= 2 && bind != h.bind;
    if (bind >= 0 && bind <= 2) h.bind_set = true;
    if (rebind) h.bind = bind;
    if ((slot_bytes > 0 && (size_t)slot_bytes != h.slot_bytes) || rebind) {
        if (h.up) {
            HIPCHK(hipStreamSynchronize(h.up));
            HIPCHK(hipStreamSynchronize(h.down));
        }
        hp_destroy(h);
        if (slot_bytes > 0) h.slot_bytes = (size_t)slot_bytes;
    }
    if (chunk_bytes > 0) h.chunk_bytes = (size_t)chunk_bytes;
    if (threads > 0) {
        h.threads = threads;
        if (h.up) h.pool.resize(threads);
    }
    return SS_OK;
}

int ss_workspace_lanes(int32_t* out, int32_t n) {
    if (!out || n < 1) return fail(SS_EINVAL, "ss_workspace_lanes: out is NULL");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    int live = 0;
    for (int i = 0; i < Ctx::NLANE; ++i) live += c->lanes[i].used ? 1 : 0;
    const int32_t v[4] = {Ctx::NLANE, live, c->lane_switches, c->lane_evictions};
    for (int i = 0; i < n && i < 4; ++i) out[i] = v[i];
    return SS_OK;
}

int ss_stream_release(void* stream_) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < Ctx::NLANE; ++i) {
        Ctx::Lane& l = c->lanes[i];
        if (!l.used || l.stream != stream) continue;
        HIPCHK(hipStreamSynchronize(stream));
        void** ws = i == c->cur_lane ? c->ws : l.ws;             // the active lane's slots live in the context
        size_t* cap = i == c->cur_lane ? c->ws_cap : l.cap;
        for (int k = 0; k < WS_COUNT; ++k) {
            if (k == WS_K1) continue;                            // (the bank generator's slots are shared by all lanes)
            if (ws[k]) HIPCHK(hipFree(ws[k]));
            ws[k] = nullptr;
            cap[k] = 0;
        }
        c->lufs_bounds_dev = c->kw_cached_dev = c->gw_cached_dev = nullptr;      // (pointer-identity caches of tables that lived in the freed slots:
                                                                                   //  a later allocation may land on the same address)
        l.used = false;
        l.stream = nullptr;
        l.dev_planned = false;
        l.status_zeroed = false;
        l.tick = 0;
    }
    return SS_OK;
}

int ss_host_path_stats(double* out, int32_t n) {
    if (!out || n < 1) return fail(SS_EINVAL, "ss_host_path_stats: out is NULL");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    const HostPipe& h = c->pipe;
    double v[18] = {h.st_seconds, h.st_bytes_up, h.st_bytes_down, (double)h.st_chunks, (double)h.st_direct, (double)h.threads};
    for (int i = 0; i < 8; ++i) v[6 + i] = h.st_mark[i];
    v[14] = (double)h.st_aborted;
    v[15] = (double)h.bind;
    v[16] = (double)h.pool.groups.size();          // last-level-cache groups the copy threads are spread over right now (0: unbound)
    v[17] = (double)h.pool.bound_node;             // NUMA node they follow (-2: never bound, -1: unbound)
    for (int i = 0; i < n && i < 18; ++i) out[i] = v[i];
    return SS_OK;
}

int ss_host_alloc(void** out, int64_t bytes) {
    if (!out || bytes < 1) return fail(SS_EINVAL, "ss_host_alloc: bad argument");
    Ctx* c;
    int rc = get_ctx(&c);      // (selects / initialises the device the allocation is mapped for)
    if (rc) return rc;
    HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return SS_OK;
}

int ss_host_free(void* p) {
    if (p) HIPCHK(hipHostFree(p));
    return SS_OK;
}

int ss_convolve_moving_checked_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L, const int64_t* idx,
                                   const float* w, float* y, uint32_t flags, void* stream, int64_t* status) {
    if (!status) return fail(SS_EINVAL, "status is NULL");
    return render(COEF_EXPLICIT, x, T, rirs, P, C, L, nullptr, idx, w, y, flags | SS_FLAG_ASYNC_PLAN, stream, nullptr, status);
}

int ss_set_task_queue(int dynamic) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    c->dynq = dynamic != 0;
    return SS_OK;
}

int ss_async_status(int32_t* code, int64_t* where, void* stream_) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (code) *code = 0;
    if (where) *where = 0;
    if (!c->async_status) return SS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    // the status word is per device: the planner that may have latched it ran on the stream of the last render -- wait for THAT stream
    // first, or a poll through another stream could read the word before k_plan_explicit has run and miss the error
    { const int rcl = sync_other_lanes(c, stream); if (rcl) return rcl; }
    int32_t h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, c->async_status, 16, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (h[0]) HIPCHK(hipMemsetAsync(c->async_status, 0, 16, stream));
    if (code) *code = h[0];
    if (where) *where = h[0] == 1 ? (int64_t)h[1] * DTILE : (int64_t)h[1];
    return SS_OK;
}

int ss_plan_status_last(int32_t* out_of_range, int64_t* where, int32_t* too_irregular, void* stream_) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (out_of_range) *out_of_range = 0;
    if (where) *where = 0;
    if (too_irregular) *too_irregular = 0;
    hipStream_t stream = (hipStream_t)stream_;
    // the verdict of the last render ON THIS STREAM: its words live in the stream's workspace lane (round 6), so renders pending on other
    // streams neither have to be waited for nor can they have overwritten them
    int lane = -1;
    for (int i = 0; i < Ctx::NLANE; ++i) if (c->lanes[i].used && c->lanes[i].stream == stream) lane = i;
    void* words = lane < 0 ? nullptr : (lane == c->cur_lane ? c->ws[WS_STATUS] : c->lanes[lane].ws[WS_STATUS]);
    if (lane < 0 || !c->lanes[lane].dev_planned || !words) {      // that render validated (or had nothing to validate) on the host: -1 = "not device-planned"
        if (out_of_range) *out_of_range = -1;
        return SS_OK;
    }
    int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, words, 32, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (out_of_range) *out_of_range = h[3];
    if (where) *where = (int64_t)h[4] * DTILE;
    if (too_irregular) *too_irregular = h[2];
    return SS_OK;
}

int ss_convolve_moving_seg_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L,
                               const int64_t* seg_len, float* y, uint32_t flags, void* stream) {
    return render(COEF_SEG, x, T, rirs, P, C, L, seg_len, nullptr, nullptr, y, flags, stream);
}

int ss_convolve_scene_f32(int32_t nsrc, const float* const* x, int64_t T, const float* const* rirs, const int32_t* P, int32_t C, int32_t L,
                          const int64_t* const* seg_len, const float* const* divisor, float* const* y, uint32_t flags, void* stream) {
    return render_scene(nsrc, x, T, rirs, P, C, L, seg_len, divisor, y, flags, stream);
}

int ss_convolve_moving_seg_div_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L,
                                   const int64_t* seg_len, const float* divisor, float* y, uint32_t flags, void* stream) {
    if (!divisor) return fail(SS_EINVAL, "divisor is NULL");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "the deferred-normalisation render takes device pointers (SS_FLAG_DEVICE_PTR)");
    return render(COEF_SEG, x, T, rirs, P, C, L, seg_len, nullptr, nullptr, y, flags, stream, divisor);
}

int ss_convolve_fixed_f32(const float* x, int64_t T, const float* h, int32_t C, int32_t L, float* y, uint32_t flags,
                          void* stream) {
    return render(COEF_FIXED, x, T, h, 1, C, L, nullptr, nullptr, nullptr, y, flags, stream);
}

// peak (may be NULL): receives max |bank| -- a DEVICE float when flags has SS_FLAG_DEVICE_PTR (no synchronisation), else a host float
static int rir_synth(const SsRirParams* p, float* bank, float* peak, uint32_t flags, void* stream_) {
    if (!p || !bank || !p->delay || !p->dgain) return fail(SS_EINVAL, "NULL pointer");
    if (p->P < 1 || p->C < 1 || p->L < 1 || !(p->fs > 0) || !(p->rt60 > 0)) return fail(SS_EINVAL, "bad RIR parameters");
    if (!(p->rho >= 0.0f && p->rho < 1.0f)) return fail(SS_EINVAL, "rho must be in [0,1)");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    const size_t pc = (size_t)p->P * p->C;
    const size_t meta = pc * (sizeof(int32_t) + sizeof(float));
    const bool meta_dev = (flags & SS_FLAG_META_DEVICE) != 0;      // delay / dgain already live in HBM (a scene generator keeps its geometry there)
    if (meta_dev && !dev) return fail(SS_EINVAL, "SS_FLAG_META_DEVICE needs SS_FLAG_DEVICE_PTR");
    if (!(dev && meta_dev) && (rc = stream_enter(c, stream))) return rc;      // (staging buffers of the shared workspace)
    if ((rc = stream_enter_k1(c, stream))) return rc;
    if (!meta_dev) {
        Pinned* pin;
        if ((rc = pinned_acquire(c, meta, &pin))) return rc;
        memcpy(pin->host, p->delay, pc * sizeof(int32_t));
        memcpy((char*)pin->host + pc * sizeof(int32_t), p->dgain, pc * sizeof(float));
        if ((rc = ws_ensure(c, WS_META, meta))) return rc;
        HIPCHK(hipMemcpyAsync(c->ws[WS_META], pin->host, meta, hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(pin->ev, stream));
        pin->pending = true;
    }
    const size_t bytes = sizeof(float) * pc * p->L;
    float* dbank = bank;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_BANK, bytes))) return rc;
        dbank = (float*)c->ws[WS_BANK];
    }
    unsigned int* dpeak = nullptr;
    if (peak) {
        if (dev) dpeak = reinterpret_cast<unsigned int*>(peak);
        else {
            if ((rc = ws_ensure(c, WS_SCR, 64))) return rc;
            dpeak = (unsigned int*)c->ws[WS_SCR];
        }
    }
    RirDev d;
    d.P = p->P; d.C = p->C; d.L = p->L;
    d.tail_gain = p->tail_gain; d.rho = p->rho;
    d.srho = (float)std::sqrt(1.0 - (double)p->rho * (double)p->rho);
    d.inv_tau = 6.91 / ((double)p->rt60 * (double)p->fs);
    d.seed = p->seed;
    d.delay = meta_dev ? p->delay : (const int32_t*)c->ws[WS_META];
    d.dgain = meta_dev ? p->dgain : (const float*)((const char*)c->ws[WS_META] + pc * sizeof(int32_t));
    const int64_t CL = (int64_t)p->C * p->L;
    const bool fast32 = (uint64_t)pc * (uint64_t)p->L < ((uint64_t)1 << 33);      // pair counters below 2^32
    // taps per thread: the chain over the positions is sequential, so the parallelism is C * L / V threads.  Two taps (one Box-Muller
    // pair, 8-byte stores) per thread: a config-2 bank (384 000 taps per position) runs 3 000 waves on the 1 024 SIMDs; four taps per
    // thread (16-byte stores) once that still leaves 8 waves per SIMD.  An odd L takes the one-tap form.
    static const int synth_v = knob("SS_SYNTH_V") ? atoi(knob("SS_SYNTH_V")) : 0;
    const bool even = p->L % 2 == 0;
    const bool big = CL / 4 / 64 >= (int64_t)c->num_cu * 32;
    const bool vec4 = even && p->L % 4 == 0 && ((uintptr_t)dbank & 15) == 0 && (synth_v ? synth_v == 4 : big);
    const bool vec2 = !vec4 && even && ((uintptr_t)dbank & 7) == 0 && (synth_v ? synth_v != 1 : true);
    const dim3 grid((unsigned)((CL / (vec4 ? 4 : (vec2 ? 2 : 1)) + 255) / 256));
    unsigned int* slots = nullptr;
    if (dpeak) {       // per-workgroup maxima + the arrival ticket (word 0; the last arriver of every launch leaves it at 0 again)
        const size_t need = sizeof(unsigned int) * ((size_t)grid.x + 1);
        if (c->ws_cap[WS_K1] < need) {
            if ((rc = ws_ensure(c, WS_K1, need))) return rc;
            HIPCHK(hipMemsetAsync(c->ws[WS_K1], 0, sizeof(unsigned int), stream));
        }
        slots = (unsigned int*)c->ws[WS_K1];
        c->k1_batch_per = 0;               // (the batched form lays its tickets out differently: it re-zeroes them when it runs next)
    }
    if (fast32 && vec4) hipLaunchKernelGGL((k_rir_synth<true, 4>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    else if (fast32 && vec2) hipLaunchKernelGGL((k_rir_synth<true, 2>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    else if (fast32) hipLaunchKernelGGL((k_rir_synth<true, 1>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    else if (vec4) hipLaunchKernelGGL((k_rir_synth<false, 4>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    else if (vec2) hipLaunchKernelGGL((k_rir_synth<false, 2>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    else hipLaunchKernelGGL((k_rir_synth<false, 1>), grid, dim3(256), 0, stream, d, dbank, dpeak, slots);
    HIPCHK(hipGetLastError());
    if (!dev) {
        HIPCHK(hipMemcpyAsync(bank, dbank, bytes, hipMemcpyDeviceToHost, stream));
        if (peak) HIPCHK(hipMemcpyAsync(peak, dpeak, sizeof(float), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return SS_OK;
}

int ss_rir_bank_synth_f32(const SsRirParams* p, float* bank, uint32_t flags, void* stream) { return rir_synth(p, bank, nullptr, flags, stream); }

int ss_rir_bank_synth_batch_f32(int32_t n, const SsRirParams* prm, float* const* banks, float* const* peaks, uint32_t flags, void* stream_) {
    if (n < 1 || n > 8 || !prm || !banks) return fail(SS_EINVAL, "ss_rir_bank_synth_batch_f32: 1..8 banks");
    const uint32_t need = SS_FLAG_DEVICE_PTR | SS_FLAG_META_DEVICE;
    if ((flags & need) != need) return fail(SS_EINVAL, "the batched generator takes device banks and device-resident geometry (SS_FLAG_DEVICE_PTR | SS_FLAG_META_DEVICE)");
    const int64_t CL = (int64_t)prm[0].C * prm[0].L;
    bool one_launch = CL % 2 == 0;
    for (int b = 0; b < n; ++b) {
        const SsRirParams& p = prm[b];
        if (!banks[b] || !p.delay || !p.dgain) return fail(SS_EINVAL, "bank %d: NULL pointer", b);
        if (p.P < 1 || p.C < 1 || p.L < 1 || !(p.fs > 0) || !(p.rt60 > 0) || !(p.rho >= 0.0f && p.rho < 1.0f)) return fail(SS_EINVAL, "bank %d: bad RIR parameters", b);
        if ((int64_t)p.C * p.L != CL || p.L % 2 != 0) one_launch = false;
        if ((uint64_t)p.P * (uint64_t)CL >= ((uint64_t)1 << 33)) one_launch = false;            // the 32-bit pair counter of the fast path
        if (((uintptr_t)banks[b] & 15) != 0) one_launch = false;
    }
    if (!one_launch) {      // shapes the one-launch form does not cover: bank by bank (same values either way)
        for (int b = 0; b < n; ++b) {
            const int rc = rir_synth(&prm[b], banks[b], peaks ? peaks[b] : nullptr, flags, stream_);
            if (rc) return rc;
        }
        return SS_OK;
    }
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter_k1(c, stream))) return rc;
    // the widest form every bank admits (the same choice rir_synth makes for one bank: values do not depend on it)
    bool all4 = true;
    for (int b = 0; b < n; ++b) all4 = all4 && prm[b].L % 4 == 0;
    const bool vec4 = all4 && CL / 4 / 64 >= (int64_t)c->num_cu * 32;        // (four taps per thread in a scene's five-bank launch: 192-193 us against 187-195, profiles/r06ao)
    const unsigned gx = (unsigned)((CL / (vec4 ? 4 : 2) + 255) / 256);
    const size_t per = (size_t)gx + 1, needw = sizeof(unsigned int) * per * 8;
    if (c->ws_cap[WS_K1] < needw) {
        if ((rc = ws_ensure(c, WS_K1, needw))) return rc;
        HIPCHK(hipMemsetAsync(c->ws[WS_K1], 0, needw, stream));      // the arrival tickets (word 0 of every bank's slot block) start at 0
        c->k1_batch_per = per;
    } else if (c->k1_batch_per != per) {                             // another slot layout was in use: its tickets sit elsewhere
        HIPCHK(hipMemsetAsync(c->ws[WS_K1], 0, c->ws_cap[WS_K1] < needw ? c->ws_cap[WS_K1] : needw, stream));
        c->k1_batch_per = per;
    }
    RirBatch tab;
    memset(&tab, 0, sizeof(tab));
    for (int b = 0; b < n; ++b) {
        const SsRirParams& p = prm[b];
        RirDev& d = tab.p[b];
        d.P = p.P; d.C = p.C; d.L = p.L;
        d.tail_gain = p.tail_gain; d.rho = p.rho;
        d.srho = (float)std::sqrt(1.0 - (double)p.rho * (double)p.rho);
        d.inv_tau = 6.91 / ((double)p.rt60 * (double)p.fs);
        d.seed = p.seed;
        d.delay = p.delay; d.dgain = p.dgain;
        tab.bank[b] = banks[b];
        tab.peak[b] = (peaks && peaks[b]) ? reinterpret_cast<unsigned int*>(peaks[b]) : nullptr;
        tab.slots[b] = (unsigned int*)c->ws[WS_K1] + per * (size_t)b;
    }
    // SS_FLAG_BACKGROUND: 16 000 B of dynamic LDS nobody touches cap the generator at five workgroups per CU (eight by its wave slots), so that the kernels of
    // another stream -- a scene's loudness / mix while the NEXT scene's banks are generated -- find wave slots: a generator workgroup lives for the whole
    // launch (every thread walks all positions), the short kernels beside it otherwise queue behind it.  A scene 0.957-0.967 -> 0.928-0.933 ms; the five-bank
    // launch alone 186 -> 194 us (profiles/r06az; tuning knob SS_K1_LDS_PAD overrides the bytes).
    static const int k1_pad_knob = knob("SS_K1_LDS_PAD") ? atoi(knob("SS_K1_LDS_PAD")) : -1;
    const int k1_pad = k1_pad_knob >= 0 ? k1_pad_knob : ((flags & SS_FLAG_BACKGROUND) ? 16000 : 0);
    if (vec4) hipLaunchKernelGGL((k_rir_synth_batch<4>), dim3(gx, (unsigned)n), dim3(256), (size_t)k1_pad, stream, tab);
    else hipLaunchKernelGGL((k_rir_synth_batch<2>), dim3(gx, (unsigned)n), dim3(256), (size_t)k1_pad, stream, tab);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_rir_bank_synth_peak_f32(const SsRirParams* p, float* bank, float* peak, uint32_t flags, void* stream) {
    if (!peak) return fail(SS_EINVAL, "peak is NULL");
    return rir_synth(p, bank, peak, flags, stream);
}

// divisor: the peak already known (device float with SS_FLAG_DEVICE_PTR, else host float); NULL = find it (k_absmax)
static int normalize(float* data, int64_t n, const float* divisor, float* peak_out, uint32_t flags, void* stream_) {
    if (n < 0 || (n > 0 && !data)) return fail(SS_EINVAL, "bad argument");
    if (n == 0) { if (peak_out) *peak_out = 0.0f; return SS_OK; }
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    float* d = data;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_BANK, sizeof(float) * n))) return rc;
        HIPCHK(hipMemcpyAsync(c->ws[WS_BANK], data, sizeof(float) * n, hipMemcpyHostToDevice, stream));
        d = (float*)c->ws[WS_BANK];
    }
    if ((rc = ws_ensure(c, WS_SCR, 64))) return rc;
    const unsigned int* bits = (const unsigned int*)c->ws[WS_SCR];
    if (divisor && dev) {
        bits = reinterpret_cast<const unsigned int*>(divisor);
    } else if (divisor) {
        HIPCHK(hipMemcpyAsync(c->ws[WS_SCR], divisor, sizeof(float), hipMemcpyHostToDevice, stream));   // (pageable source: returns after the copy)
    } else {
        HIPCHK(hipMemsetAsync(c->ws[WS_SCR], 0, sizeof(unsigned int), stream));
        hipLaunchKernelGGL(k_absmax, dim3(grid_for((n + 31) >> 5, 512)), dim3(256), 0, stream, (const float*)d, n, (unsigned int*)c->ws[WS_SCR]);
    }
    hipLaunchKernelGGL(k_divide, dim3(grid_for((n + 3) >> 2, 8192)), dim3(256), 0, stream, d, n, bits);
    HIPCHK(hipGetLastError());
    if (peak_out) {
        unsigned int hb = 0;
        HIPCHK(hipMemcpyAsync(&hb, bits, sizeof(hb), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        memcpy(peak_out, &hb, sizeof(float));
    }
    if (!dev) {
        HIPCHK(hipMemcpyAsync(data, d, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return SS_OK;
}

int ss_peak_normalize_f32(float* data, int64_t n, float* peak_out, uint32_t flags, void* stream) {
    return normalize(data, n, nullptr, peak_out, flags, stream);
}

int ss_divide_by_f32(float* data, int64_t n, const float* divisor, uint32_t flags, void* stream) {
    if (!divisor) return fail(SS_EINVAL, "divisor is NULL");
    return normalize(data, n, divisor, nullptr, flags, stream);
}

int ss_rms_db_f32(const float* x, int64_t n, int32_t count, double* out_db, uint32_t flags, void* stream_) {
    if (n <= 0 || count <= 0 || !x || !out_db) return fail(SS_EINVAL, "bad argument");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const void* dx;
    if ((rc = stage_in(c, WS_Y, x, sizeof(float) * (size_t)n * count, (flags & SS_FLAG_DEVICE_PTR) != 0, stream, &dx))) return rc;
    const int nb = grid_for(n, 512);
    if ((rc = ws_ensure(c, WS_SCR, sizeof(double) * (size_t)nb * count))) return rc;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(double) * count))) return rc;
    hipLaunchKernelGGL(k_partial_sum<0>, dim3(nb, count), dim3(256), 0, stream, (const float*)dx, n, (double*)c->ws[WS_SCR]);
    hipLaunchKernelGGL(k_final_sum, dim3(count), dim3(64), 0, stream, (const double*)c->ws[WS_SCR], nb, (double*)c->ws[WS_SCR2]);
    HIPCHK(hipGetLastError());
    std::vector<double> ssq(count);
    HIPCHK(hipMemcpyAsync(ssq.data(), c->ws[WS_SCR2], sizeof(double) * count, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    for (int i = 0; i < count; ++i) {
        const double ms = ssq[i] / (double)n;
        out_db[i] = 10.0 * std::log10(ms > 1e-20 ? ms : 1e-20);
    }
    return SS_OK;
}

int ss_mix_f32(float* speakers, int32_t S, const float* noises, int32_t N, int64_t n, const float* sirs, float snr, float* mix,
               float* gains_out, uint32_t flags, void* stream_) {
    if (S < 1 || N < 1 || n <= 0 || !speakers || !noises || !mix || (S > 1 && !sirs)) return fail(SS_EINVAL, "bad argument");
    if (S > 64) return fail(SS_EINVAL, "too many speakers");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    const int write_back = (flags & SS_FLAG_KEEP_SPEAKERS) ? 0 : 1;     // the reference scales the interferers in place (:113); a pipeline that only
                                                                         // wants the mix keeps its normalised stems (no clone, one stem less to write)
    float* dspk = speakers;
    const float* dnoise = noises;
    float* dmix = mix;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_Y, sizeof(float) * (size_t)S * n))) return rc;
        if ((rc = ws_ensure(c, WS_BANK, sizeof(float) * (size_t)N * n))) return rc;
        if ((rc = ws_ensure(c, WS_X, sizeof(float) * (size_t)n))) return rc;
        HIPCHK(hipMemcpyAsync(c->ws[WS_Y], speakers, sizeof(float) * (size_t)S * n, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(c->ws[WS_BANK], noises, sizeof(float) * (size_t)N * n, hipMemcpyHostToDevice, stream));
        dspk = (float*)c->ws[WS_Y];
        dnoise = (const float*)c->ws[WS_BANK];
        dmix = (float*)c->ws[WS_X];
    }
    const int nb = grid_for(n, 512);
    const size_t scr = sizeof(double) * (size_t)nb * (S > 2 ? S : 2);
    if ((rc = ws_ensure(c, WS_SCR, scr))) return rc;
    // WS_SCR2: [S doubles sumsq][2 doubles][S+1 floats gains][S floats sirs]
    const size_t off_s2 = sizeof(double) * S, off_g = off_s2 + sizeof(double) * 2, off_sir = off_g + sizeof(float) * (S + 1);
    if ((rc = ws_ensure(c, WS_SCR2, off_sir + sizeof(float) * S + 16))) return rc;
    char* s2 = (char*)c->ws[WS_SCR2];
    double* d_sumsq = (double*)s2;
    double* d_s2 = (double*)(s2 + off_s2);
    float* d_g = (float*)(s2 + off_g);
    float* d_sir = (float*)(s2 + off_sir);
    SirTab sir_tab;
    memset(&sir_tab, 0, sizeof(sir_tab));
    for (int i = 0; i + 1 < S; ++i) sir_tab.v[i] = sirs[i];
    (void)d_sir;
    if (n % 4 == 0 && ((uintptr_t)dspk & 15) == 0)
        hipLaunchKernelGGL(k_partial_sum4<0>, dim3(nb, S), dim3(256), 0, stream, (const float*)dspk, n / 4, (double*)c->ws[WS_SCR]);
    else
        hipLaunchKernelGGL(k_partial_sum<0>, dim3(nb, S), dim3(256), 0, stream, (const float*)dspk, n, (double*)c->ws[WS_SCR]);
    hipLaunchKernelGGL(k_mix_gains1, dim3(1), dim3(1024), 0, stream, (const double*)c->ws[WS_SCR], nb, S, (double)n, sir_tab, d_g, d_sumsq);
    if (n % 4 == 0 && (((uintptr_t)dspk | (uintptr_t)dnoise | (uintptr_t)dmix) & 15) == 0)
        hipLaunchKernelGGL(k_mix_scale_sum4, dim3(nb), dim3(256), 0, stream, dspk, S, dnoise, N, n / 4, (const float*)d_g, dmix,
                           (double*)c->ws[WS_SCR], write_back);
    else
        hipLaunchKernelGGL(k_mix_scale_sum, dim3(nb), dim3(256), 0, stream, dspk, S, dnoise, N, n, (const float*)d_g, dmix,
                           (double*)c->ws[WS_SCR], write_back);
    hipLaunchKernelGGL(k_mix_gains2, dim3(1), dim3(128), 0, stream, (const double*)c->ws[WS_SCR], nb, (double)n, snr, d_g, S);
    hipLaunchKernelGGL(k_mix_final, dim3(grid_for(n)), dim3(256), 0, stream, dnoise, N, n, (const float*)d_g, S, dmix);
    HIPCHK(hipGetLastError());
    if (gains_out) {
        std::vector<float> g(S + 1);
        HIPCHK(hipMemcpyAsync(g.data(), d_g, sizeof(float) * (S + 1), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int i = 1; i <= S; ++i) gains_out[i - 1] = g[i];
    }
    if (!dev) {
        HIPCHK(hipMemcpyAsync(speakers, dspk, sizeof(float) * (size_t)S * n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(mix, dmix, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return SS_OK;
}

int ss_mix_presum_f32(float* speakers, int32_t S, const float* noise, int64_t n, const float* sirs, float snr, float* mix,
                      const double* sumsq_speakers, const double* sumsq_noise, float* gains_dev, uint32_t flags, void* stream_) {
    if (S < 1 || S > 64 || n <= 0 || !speakers || !noise || !mix || !sumsq_speakers || !sumsq_noise || (S > 1 && !sirs)) return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR): the stem energies are device-side by-products");
    if (n % 4 != 0 || (((uintptr_t)speakers | (uintptr_t)noise | (uintptr_t)mix) & 15) != 0)
        return fail(SS_EINVAL, "ss_mix_presum_f32 needs 16-byte aligned stems of a multiple of four samples (use ss_mix_f32)");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const int write_back = (flags & SS_FLAG_KEEP_SPEAKERS) ? 0 : 1;
    const int nb = grid_for(n, 512);                      // (the grid of ss_mix_f32's pass: the same partial sums of the speech energy)
    if ((rc = ws_ensure(c, WS_SCR, sizeof(double) * (size_t)nb))) return rc;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(float) * ((size_t)S + 1) + 16))) return rc;
    float* d_g = gains_dev ? gains_dev : (float*)c->ws[WS_SCR2];
    SirTab sir_tab;
    memset(&sir_tab, 0, sizeof(sir_tab));
    for (int i = 0; i + 1 < S; ++i) sir_tab.v[i] = sirs[i];
    hipLaunchKernelGGL(k_mix_pre_scale_sum4, dim3(nb), dim3(256), 0, stream, speakers, S, n / 4, sumsq_speakers, (double)n, sir_tab, mix,
                       (double*)c->ws[WS_SCR], d_g, write_back);
    hipLaunchKernelGGL(k_mix_pre_final4, dim3(grid_for(n / 4)), dim3(256), 0, stream, noise, n / 4, (const double*)c->ws[WS_SCR], nb, (double)n,
                       sumsq_noise, snr, mix, d_g, S);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_mix_onepass_f32(float* speakers, int32_t S, const float* noise, int64_t n, const float* sirs, float snr, float* mix,
                       const double* sumsq_speakers, const double* cross_speakers, const double* sumsq_noise, float* gains_dev, uint32_t flags, void* stream_) {
    if (S < 1 || S > 64 || n <= 0 || !speakers || !noise || !mix || !sumsq_speakers || !sumsq_noise || (S > 1 && (!sirs || !cross_speakers)))
        return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR): the stem energies are device-side by-products");
    if (n % 4 != 0 || (((uintptr_t)speakers | (uintptr_t)noise | (uintptr_t)mix) & 15) != 0)
        return fail(SS_EINVAL, "ss_mix_onepass_f32 needs 16-byte aligned stems of a multiple of four samples (use ss_mix_f32)");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const int write_back = (flags & SS_FLAG_KEEP_SPEAKERS) ? 0 : 1;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(float) * ((size_t)S + 1) + 16))) return rc;
    float* d_g = gains_dev ? gains_dev : (float*)c->ws[WS_SCR2];
    SirTab sir_tab;
    memset(&sir_tab, 0, sizeof(sir_tab));
    for (int i = 0; i + 1 < S; ++i) sir_tab.v[i] = sirs[i];
    hipLaunchKernelGGL(k_mix_onepass4, dim3(grid_for(n / 4)), dim3(256), 0, stream, speakers, S, noise, n / 4, sumsq_speakers, cross_speakers, sumsq_noise,
                       (double)n, sir_tab, snr, mix, d_g, write_back);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

// K-weighting coefficients -> KwCoef (normalised biquads + the powers of the chunk transition matrix the scan needs)
static int kw_setup(const double* coef, KwCoef& k) {
    // the tables depend on the 12 coefficients only (one set per sample rate): reuse the last set's tables
    static double last_coef[12];
    static KwCoef last_k;
    static bool have_last = false;
    static std::mutex kw_mu;                 // process-wide cache shared by every device context
    std::lock_guard<std::mutex> kw_lock(kw_mu);
    if (have_last && memcmp(last_coef, coef, sizeof(last_coef)) == 0) { k = last_k; return SS_OK; }
    for (int s = 0; s < 2; ++s) {
        const double a0 = coef[s * 6 + 3];
        if (a0 == 0.0) return fail(SS_EINVAL, "a0 == 0");
        for (int i = 0; i < 3; ++i) { k.b[s][i] = coef[s * 6 + i] / a0; k.a[s][i] = coef[s * 6 + 3 + i] / a0; }
    }
    for (int st = 0; st < 2; ++st) {                    // delta form (KwTabT): d2 = 1 - a2, eps = A(1), beta = B(1), b0, b2
        k.c[st][0] = 1.0 - k.a[st][2];
        k.c[st][1] = (1.0 + k.a[st][1]) + k.a[st][2];
        k.c[st][2] = (k.b[st][0] + k.b[st][1]) + k.b[st][2];
        k.c[st][3] = k.b[st][0];
        k.c[st][4] = k.b[st][2];
        k.c[st][5] = 0.0;
    }
    // chunk transition matrix: zero-input response of the cascade from the 4 unit states
    for (int u = 0; u < 4; ++u) {
        double s[4] = {0, 0, 0, 0};
        s[u] = 1.0;
        for (int t = 0; t < KW_CHUNK; ++t) kw_step<double>(k, s, 0.0);
        for (int r = 0; r < 4; ++r) k.Mp[0][r * 4 + u] = s[r];
    }
    // W[t] = end state of a chunk whose only non-zero sample is x[t] = 1
    for (int t = 0; t < KW_CHUNK; ++t) {
        double s[4] = {0, 0, 0, 0};
        for (int u = t; u < KW_CHUNK; ++u) kw_step<double>(k, s, u == t ? 1.0 : 0.0);
        for (int r = 0; r < 4; ++r) k.W[t][r] = s[r];
    }
    for (int i = 1; i < 13; ++i)                        // Mp[i] = Mp[i-1]^2
        for (int r = 0; r < 4; ++r)
            for (int q = 0; q < 4; ++q) {
                double acc = 0;
                for (int m = 0; m < 4; ++m) acc += k.Mp[i - 1][r * 4 + m] * k.Mp[i - 1][m * 4 + q];
                k.Mp[i][r * 4 + q] = acc;
            }
    memcpy(last_coef, coef, sizeof(last_coef));
    last_k = k;
    have_last = true;
    return SS_OK;
}

// device part of row U's measurement: leaves z[C][nblocks] (float64) in ws[WS_SCR2].  da is a device pointer.
static int kw_block_power_dev(Ctx* c, const float* da, int64_t T, int32_t C, int64_t st, int64_t sc, const KwCoef& k, const int64_t* lo,
                              const int64_t* hi, int32_t nblocks, double norm, hipStream_t stream) {
    int rc;
    if ((T + KW_CHUNK - 1) / KW_CHUNK > (int64_t)INT32_MAX / (4 * C)) return fail(SS_EINVAL, "audio too long");
    const int nchunks = (int)((T + KW_CHUNK - 1) / KW_CHUNK);
    if ((rc = ws_ensure(c, WS_SCR, sizeof(double) * 4 * (size_t)C * nchunks))) return rc;
    if ((rc = ws_ensure(c, WS_FILT, sizeof(double) * (size_t)C * nchunks))) return rc;          // chunk energies
    const size_t bb = sizeof(int64_t) * (size_t)nblocks;
    if ((rc = ws_ensure(c, WS_LUFS, 2 * bb))) return rc;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(double) * ((size_t)C + 16) * nblocks))) return rc;   // z + the gate's per-block scratch (one row per stem)
    // block bounds: uploaded only when they differ from the previous call's (same T / rate / block size -> same bounds)
    if (c->lufs_bounds.size() != 2 * (size_t)nblocks || memcmp(c->lufs_bounds.data(), lo, bb) != 0 ||
        memcmp(c->lufs_bounds.data() + nblocks, hi, bb) != 0 || c->lufs_bounds_dev != c->ws[WS_LUFS]) {
        Pinned* pin;
        if ((rc = pinned_acquire(c, 2 * bb, &pin))) return rc;
        memcpy(pin->host, lo, bb);
        memcpy((char*)pin->host + bb, hi, bb);
        HIPCHK(hipMemcpyAsync(c->ws[WS_LUFS], pin->host, 2 * bb, hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(pin->ev, stream));
        pin->pending = true;
        c->lufs_bounds.assign(lo, lo + nblocks);
        c->lufs_bounds.insert(c->lufs_bounds.end(), hi, hi + nblocks);
        c->lufs_bounds_dev = c->ws[WS_LUFS];
    }
    // per-lane carry powers (M^KW_SER)^lane of the scan: uploaded when the coefficients change (i.e. once per sample rate)
    const int ntiles = (nchunks + KW_TILE * KW_SER - 1) / (KW_TILE * KW_SER);
    const size_t koff = (sizeof(KwCoef) + 255) & ~(size_t)255;      // ws[WS_KWP] = [KwCoef][72 4x4 carry-power matrices]
    const size_t foff = koff + sizeof(double) * (72 + 64) * 16;         // + [64] M^lane for the fused kernel
    const size_t fpoff = foff + ((sizeof(KwCoefF) + 255) & ~(size_t)255);                        // float32 copies for k_kw_fused<., float>: [KwCoefF][64 x M^lane]
    const size_t ptab_bytes = fpoff + sizeof(float) * 64 * 16;
    if ((rc = ws_ensure(c, WS_KWP, ptab_bytes))) return rc;                                       // fixed size: never reallocated
    if ((rc = ws_ensure(c, WS_KWT, sizeof(double) * 4 * (size_t)C * ntiles))) return rc;           // tile totals
    if (memcmp(c->kw_cached, k.b, sizeof(c->kw_cached)) != 0 || c->kw_cached_dev != c->ws[WS_KWP]) {
        Pinned* pin;
        if ((rc = pinned_acquire(c, ptab_bytes, &pin))) return rc;
        memcpy(pin->host, &k, sizeof(KwCoef));
        double* pt = (double*)((char*)pin->host + koff);
        for (int i = 0; i < 16; ++i) pt[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int l = 1; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q) {
                    double acc = 0;
                    for (int m = 0; m < 4; ++m) acc += pt[(l - 1) * 16 + r * 4 + m] * k.Mp[3][m * 4 + q];
                    pt[l * 16 + r * 4 + q] = acc;
                }
        double* qt = pt + 64 * 16;                      // (M^512)^wave
        for (int i = 0; i < 16; ++i) qt[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int l = 1; l < 8; ++l)
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q) {
                    double acc = 0;
                    for (int m = 0; m < 4; ++m) acc += qt[(l - 1) * 16 + r * 4 + m] * k.Mp[9][m * 4 + q];
                    qt[l * 16 + r * 4 + q] = acc;
                }
        double* lt = pt + 72 * 16;                      // M^lane
        for (int i = 0; i < 16; ++i) lt[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int l = 1; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q) {
                    double acc = 0;
                    for (int m = 0; m < 4; ++m) acc += lt[(l - 1) * 16 + r * 4 + m] * k.Mp[0][m * 4 + q];
                    lt[l * 16 + r * 4 + q] = acc;
                }
        {                                                // the float32 table set: every entry the float64 one's nearest float
            KwCoefF* kf = (KwCoefF*)((char*)pin->host + foff);
            const double* src = (const double*)&k;
            float* dst = (float*)kf;
            for (size_t i = 0; i < sizeof(KwCoef) / sizeof(double); ++i) dst[i] = (float)src[i];
            float* lf = (float*)((char*)pin->host + fpoff);
            for (int i = 0; i < 64 * 16; ++i) lf[i] = (float)lt[i];
        }
        HIPCHK(hipMemcpyAsync(c->ws[WS_KWP], pin->host, ptab_bytes, hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(pin->ev, stream));
        pin->pending = true;
        memcpy(c->kw_cached, k.b, sizeof(c->kw_cached));
        c->kw_cached_dev = c->ws[WS_KWP];
    }
    const int nthreads = C * nchunks;
    double* states = (double*)c->ws[WS_SCR];
    double* energy = (double*)c->ws[WS_FILT];
    const KwCoef* kd = (const KwCoef*)c->ws[WS_KWP];
    double* ptab = (double*)((char*)c->ws[WS_KWP] + koff);
    double* tot = (double*)c->ws[WS_KWT];
    // history the fused kernel needs: the first power of two H with |M^H| <= 1e-20 entrywise (1e-12 for the float32 walk, whose states carry 6e-8)
    static const int kw_f64 = knob("SS_KW_F64") ? atoi(knob("SS_KW_F64")) : 0;      // (tuning build: the float64 walk of rounds 2-5)
    int hp = 0;
    for (; hp < 10; ++hp) {
        double mx = 0;
        for (int i = 0; i < 16; ++i) mx = std::max(mx, std::fabs(k.Mp[hp][i]));
        if (mx <= (kw_f64 ? 1e-20 : 1e-12)) break;
    }
    const char* fe = getenv("SS_KW_EXACT");        // test switch (the only environment variable the product library reads): force the exact multi-launch scan
    const bool force_exact = fe && atoi(fe);
    if (hp <= 8 && !force_exact) {
        const int H = 1 << hp;
        const double* plane = ptab + 72 * 16;
        const KwCoefF* kf = (const KwCoefF*)((char*)c->ws[WS_KWP] + foff);
        const float* planef = (const float*)((char*)c->ws[WS_KWP] + fpoff);
        const int nt = H <= 64 ? 256 : 512, tiles = (nchunks + (nt - H) - 1) / (nt - H);
        if (kw_f64) {
            if (nt == 256) hipLaunchKernelGGL((k_kw_fused<256, double>), dim3(tiles, C), dim3(nt), 0, stream, da, T, st, sc, kd, plane, nchunks, H, states, energy);
            else hipLaunchKernelGGL((k_kw_fused<512, double>), dim3(tiles, C), dim3(nt), 0, stream, da, T, st, sc, kd, plane, nchunks, H, states, energy);
        } else {
            static const int kw_lds = knob("SS_KW_LDS") ? atoi(knob("SS_KW_LDS")) : 0;  // (tuning build: the float32 walk with the chunks staged in LDS)
            if (kw_lds) {
                if (nt == 256) hipLaunchKernelGGL((k_kw_fused<256, float>), dim3(tiles, C), dim3(nt), 0, stream, da, T, st, sc, kf, planef, nchunks, H, states, energy);
                else hipLaunchKernelGGL((k_kw_fused<512, float>), dim3(tiles, C), dim3(nt), 0, stream, da, T, st, sc, kf, planef, nchunks, H, states, energy);
            } else {
                const bool hps = k.c[1][2] == 0.0 && k.c[1][3] == k.c[1][4];                    // BS.1770's high-pass: b = g (1, -2, 1)
                static const int kw_nt = knob("SS_KW_NT") ? atoi(knob("SS_KW_NT")) : 0;        // (tuning build: workgroup size of the walk)
                const int nt2 = kw_nt == 512 || H > 64 ? 512 : 256, tiles2 = (nchunks + (nt2 - H) - 1) / (nt2 - H);
                auto kern = nt2 == 256 ? (hps ? k_kw_fused32<256, true> : k_kw_fused32<256, false>) : (hps ? k_kw_fused32<512, true> : k_kw_fused32<512, false>);
                hipLaunchKernelGGL(kern, dim3(tiles2, C), dim3(nt2), 0, stream, da, T, st, sc, kf, planef, nchunks, H, states, energy, k.c[1][3] * k.c[1][3]);
            }
        }
    } else {
        hipLaunchKernelGGL(k_kw_state, dim3((nthreads + 255) / 256), dim3(256), 0, stream, da, T, C, st, sc, kd, nchunks, states);
        hipLaunchKernelGGL(k_kw_scan_local, dim3(ntiles, C), dim3(KW_TILE), 0, stream, kd, nchunks, ntiles, (const double*)ptab, states, tot);
        if (ntiles > 1)
            hipLaunchKernelGGL(k_kw_scan_carry, dim3(ntiles - 1, C), dim3(KW_TILE), 0, stream, kd, nchunks, ntiles, (const double*)ptab,
                               states, (const double*)tot);
        hipLaunchKernelGGL(k_kw_energy, dim3((nthreads + 255) / 256), dim3(256), 0, stream, da, T, C, st, sc, kd, nchunks,
                           (const double*)states, energy);
    }
    hipLaunchKernelGGL(k_block_power_chunks, dim3(nblocks, C), dim3(64), 0, stream, da, T, st, sc, kd, nchunks,
                       (const double*)states, (const double*)energy, (const int64_t*)c->ws[WS_LUFS],
                       (const int64_t*)((const char*)c->ws[WS_LUFS] + bb), nblocks, 1.0 / norm, (double*)c->ws[WS_SCR2]);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_kweighted_block_power_f32(const float* audio, int64_t T, int32_t C, const double* coef, const int64_t* lo,
                                 const int64_t* hi, int32_t nblocks, double norm, double* z_out, uint32_t flags, void* stream_) {
    if (!audio || T <= 0 || C < 1 || C > 64 || !coef || nblocks < 0 || (nblocks && (!lo || !hi)) || !z_out || !(norm > 0))
        return fail(SS_EINVAL, "bad argument");
    if (nblocks == 0) return SS_OK;
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const void* da;
    if ((rc = stage_in(c, WS_Y, audio, sizeof(float) * (size_t)C * T, (flags & SS_FLAG_DEVICE_PTR) != 0, stream, &da))) return rc;
    const bool tc = (flags & SS_FLAG_LAYOUT_TC) != 0;
    KwCoef k;
    if ((rc = kw_setup(coef, k))) return rc;
    if ((rc = kw_block_power_dev(c, (const float*)da, T, C, tc ? C : 1, tc ? 1 : T, k, lo, hi, nblocks, norm, stream))) return rc;
    HIPCHK(hipMemcpyAsync(z_out, c->ws[WS_SCR2], sizeof(double) * (size_t)C * nblocks, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return SS_OK;
}

static int lufs_norm_batch(const float* audio, float* out, int64_t T, int32_t C, int32_t S, const double* coef, const int64_t* lo,
                           const int64_t* hi, int32_t nblocks, double block_norm, const double* weights, const double* targets,
                           double* result, uint32_t flags, void* stream_, double* sumsq_dev /* device [S]: sum(out^2) per stem, or null */,
                           int nspk = 0 /* > 1: sumsq_dev has S + nspk (nspk - 1) / 2 entries, the cross sums of the first nspk stems behind the energies */) {
    if (nspk > 1 && (!sumsq_dev || nspk > S || nspk > SCALE_XMAX + 1 || ((int64_t)C * T) % 4 != 0 || (((uintptr_t)audio | (uintptr_t)out) & 15) != 0))
        return fail(SS_EINVAL, "cross sums: 2 <= speakers <= %d of the stems, 16-byte aligned stems of a multiple of four samples", SCALE_XMAX + 1);
    if (sumsq_dev && (flags & (SS_FLAG_DEVICE_PTR | SS_FLAG_RESULT_DEVICE)) != (SS_FLAG_DEVICE_PTR | SS_FLAG_RESULT_DEVICE))
        return fail(SS_EINVAL, "the stem energies are a device-side by-product: SS_FLAG_DEVICE_PTR | SS_FLAG_RESULT_DEVICE");
    if (!audio || !out || T <= 0 || C < 1 || S < 1 || S > 16 || (int64_t)C * S > 64 || !coef || nblocks < 0 || (nblocks && (!lo || !hi)) ||
        !weights || !targets || !result || !(block_norm > 0))
        return fail(SS_EINVAL, "bad argument");
    const bool tc = (flags & SS_FLAG_LAYOUT_TC) != 0;
    if (tc && S > 1) return fail(SS_EINVAL, "a batch of stems must be channel-first [S][C][T]");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    const int64_t ng = (int64_t)C * T, n = ng * S;                 // elements per stem / in total
    const int CC = C * S;                                          // [S][C][T] is a [S*C][T] signal for the K-weighting kernels
    const void* da;
    if ((rc = stage_in(c, WS_Y, audio, sizeof(float) * (size_t)n, dev, stream, &da))) return rc;
    float* dout = out;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_X, sizeof(float) * (size_t)n))) return rc;
        dout = (float*)c->ws[WS_X];
    }
    KwCoef k;
    if ((rc = kw_setup(coef, k))) return rc;
    const int nb = grid_for(ng, 1024);
    const size_t res_doubles = 4 * (size_t)S + 2 * (size_t)S * nb;  // [S][4] results, [S][2][nb] partial sums
    if ((rc = ws_ensure(c, WS_RES, sizeof(double) * res_doubles))) return rc;
    double* res = (double*)c->ws[WS_RES];
    if (nblocks) {
        if ((rc = kw_block_power_dev(c, (const float*)da, T, CC, tc ? C : 1, tc ? 1 : T, k, lo, hi, nblocks, block_norm, stream))) return rc;
    } else if ((rc = ws_ensure(c, WS_SCR2, 64))) return rc;
    if ((rc = ws_ensure(c, WS_GW, sizeof(double) * 64))) return rc;       // channel weights: uploaded when they change
    {
        double gw[64];
        for (int i = 0; i < 64; ++i) gw[i] = i < C ? weights[i] : 0.0;
        if (memcmp(c->gw_cached, gw, sizeof(gw)) != 0 || c->gw_cached_dev != c->ws[WS_GW]) {
            Pinned* pin;
            if ((rc = pinned_acquire(c, sizeof(gw), &pin))) return rc;
            memcpy(pin->host, gw, sizeof(gw));
            HIPCHK(hipMemcpyAsync(c->ws[WS_GW], pin->host, sizeof(gw), hipMemcpyHostToDevice, stream));
            HIPCHK(hipEventRecord(pin->ev, stream));
            pin->pending = true;
            memcpy(c->gw_cached, gw, sizeof(gw));
            c->gw_cached_dev = c->ws[WS_GW];
        }
    }
    double* zdev = (double*)c->ws[WS_SCR2];
    const size_t gate_lds = sizeof(double) * ((size_t)C + 1) * nblocks;
    const int use_lds = gate_lds <= 60 * 1024;          // z and the block loudness in LDS when they fit the default dynamic limit
    GateTargets gt;
    for (int i = 0; i < 16; ++i) gt.t[i] = i < S ? targets[i] : 0.0;
    hipLaunchKernelGGL(k_gate, dim3(S), dim3(1024), use_lds ? gate_lds : 0, stream, (const double*)zdev, (int)C, (int)nblocks,
                       (const double*)c->ws[WS_GW], gt, zdev + (size_t)CC * nblocks, use_lds, res);
    double* part = res + 4 * (size_t)S;
    double* part_sq = nullptr;
    double* part_x = nullptr;
    const int npairs = nspk > 1 ? nspk * (nspk - 1) / 2 : 0;
    if (sumsq_dev) {
        if ((rc = ws_ensure(c, WS_SQ, sizeof(double) * ((size_t)S + npairs) * nb))) return rc;
        part_sq = (double*)c->ws[WS_SQ];
        if (npairs) part_x = part_sq + (size_t)S * nb;
    }
    hipLaunchKernelGGL(k_scale_sums, dim3(nb, S), dim3(256), 0, stream, (const float*)da, dout, ng, 0.f, (const double*)(res + 1), part, part_sq,
                       nspk, part_x);
    HIPCHK(hipGetLastError());
    if (flags & SS_FLAG_RESULT_DEVICE) {      // no host synchronisation: the four numbers per stem land in the caller's device array
        if (!dev) return fail(SS_EINVAL, "SS_FLAG_RESULT_DEVICE needs SS_FLAG_DEVICE_PTR");
        hipLaunchKernelGGL(k_lufs_result, dim3(S + npairs), dim3(64), 0, stream, (const double*)res, (const double*)part, nb, result, (const double*)part_sq,
                           sumsq_dev, (int)S, (const double*)part_x);
        HIPCHK(hipGetLastError());
        return SS_OK;
    }
    // {loudness, gain} + the partial sums of every stem come back in one go; the last (fixed-order) additions are done here
    Pinned* pin;
    if ((rc = pinned_acquire(c, sizeof(double) * res_doubles, &pin))) return rc;
    double* hp_ = (double*)pin->host;
    HIPCHK(hipMemcpyAsync(hp_, res, sizeof(double) * res_doubles, hipMemcpyDeviceToHost, stream));
    if (!dev) HIPCHK(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    for (int g = 0; g < S; ++g) {
        result[4 * g + 0] = hp_[4 * g + 0];
        result[4 * g + 1] = hp_[4 * g + 1];
        for (int h = 0; h < 2; ++h) {
            const double* pp = hp_ + 4 * (size_t)S + ((size_t)2 * g + h) * nb;
            double lanes[64] = {0};                     // same association as k_final_sum: 64 strided lanes, then a butterfly
            for (int i = 0; i < nb; ++i) lanes[i & 63] += pp[i];
            for (int o = 32; o > 0; o >>= 1)
                for (int i = 0; i < o; ++i) lanes[i] += lanes[i + o];
            result[4 * g + 2 + h] = lanes[0];
        }
    }
    return SS_OK;
}

int ss_lufs_norm_batch_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S, const double* coef, const int64_t* lo,
                           const int64_t* hi, int32_t nblocks, double block_norm, const double* weights, const double* targets,
                           double* result, uint32_t flags, void* stream_) {
    return lufs_norm_batch(audio, out, T, C, S, coef, lo, hi, nblocks, block_norm, weights, targets, result, flags, stream_, nullptr);
}

int ss_lufs_norm_batch_sq_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S, const double* coef, const int64_t* lo,
                              const int64_t* hi, int32_t nblocks, double block_norm, const double* weights, const double* targets,
                              double* result, double* sumsq, uint32_t flags, void* stream_) {
    if (!sumsq) return fail(SS_EINVAL, "sumsq is NULL");
    return lufs_norm_batch(audio, out, T, C, S, coef, lo, hi, nblocks, block_norm, weights, targets, result, flags, stream_, sumsq);
}

int ss_lufs_norm_batch_sqx_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S, int32_t nspk, const double* coef, const int64_t* lo,
                               const int64_t* hi, int32_t nblocks, double block_norm, const double* weights, const double* targets,
                               double* result, double* sums, uint32_t flags, void* stream_) {
    if (!sums) return fail(SS_EINVAL, "sums is NULL");
    if (nspk < 2) return fail(SS_EINVAL, "nspk < 2: use ss_lufs_norm_batch_sq_f32");
    return lufs_norm_batch(audio, out, T, C, S, coef, lo, hi, nblocks, block_norm, weights, targets, result, flags, stream_, sums, nspk);
}

int ss_lufs_norm_f32(const float* audio, float* out, int64_t T, int32_t C, const double* coef, const int64_t* lo, const int64_t* hi,
                     int32_t nblocks, double block_norm, const double* weights, double target_lufs, double* result, uint32_t flags,
                     void* stream_) {
    return ss_lufs_norm_batch_f32(audio, out, T, C, 1, coef, lo, hi, nblocks, block_norm, weights, &target_lufs, result, flags, stream_);
}

// upload a small host table through the pinned ring into a workspace slot (stream ordered)
static int upload_small(Ctx* c, int slot, const void* src, size_t bytes, hipStream_t stream, void** dst) {
    int rc;
    Pinned* pin;
    if ((rc = pinned_acquire(c, bytes, &pin))) return rc;
    memcpy(pin->host, src, bytes);
    if ((rc = ws_ensure(c, slot, bytes))) return rc;
    HIPCHK(hipMemcpyAsync(c->ws[slot], pin->host, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(pin->ev, stream));
    pin->pending = true;
    *dst = c->ws[slot];
    return SS_OK;
}

int ss_mean_channels_f32(const float* x, int32_t C, int64_t T, float* out, uint32_t flags, void* stream_) {
    if (!x || !out || C < 1 || T < 1) return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR)");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    hipLaunchKernelGGL(k_mean_channels, dim3(grid_for(T)), dim3(256), 0, stream, x, C, T, out);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_crop_rms_db_f32(const float* const* crops, int32_t K, int32_t C, int64_t chan_stride, int64_t n, double* out_db, uint32_t flags,
                       void* stream_) {
    if (!crops || !out_db || K < 1 || C < 1 || n < 1) return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR)");
    if (K > 4096) return fail(SS_EINVAL, "too many crops in one call");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    void* dptr;
    if ((rc = upload_small(c, WS_META, crops, sizeof(float*) * (size_t)K, stream, &dptr))) return rc;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(double) * (size_t)K))) return rc;
    hipLaunchKernelGGL(k_crop_sumsq, dim3(K), dim3(1024), 0, stream, (const float* const*)dptr, C, chan_stride, n, (double*)c->ws[WS_SCR2]);
    HIPCHK(hipGetLastError());
    std::vector<double> ssq((size_t)K);
    HIPCHK(hipMemcpyAsync(ssq.data(), c->ws[WS_SCR2], sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const double cnt = (double)C * (double)n;
    for (int i = 0; i < K; ++i) {
        const double ms = ssq[(size_t)i] / cnt;
        out_db[i] = 10.0 * std::log10(ms > 1e-20 ? ms : 1e-20);
    }
    return SS_OK;
}

int ss_mix_batch_f32(const float* const* speakers, const float* const* noises, int32_t B, int32_t S, int32_t N, int32_t C, int64_t chan_stride,
                     int64_t n, const float* sirs, const float* snrs, float* speakers_out, float* mix_out, float* gains_out, uint32_t flags,
                     void* stream_) {
    if (!speakers || !noises || !snrs || !speakers_out || !mix_out || B < 1 || S < 1 || N < 1 || C < 1 || n < 1 || (S > 1 && !sirs))
        return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR)");
    if (S > 64 || B > 65535) return fail(SS_EINVAL, "too many speakers / items");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    // one table: [B*S speaker pointers][B*N noise pointers][B*(S-1) sirs][B snrs]
    const size_t o_np = sizeof(float*) * (size_t)B * S, o_sir = o_np + sizeof(float*) * (size_t)B * N;
    const size_t o_snr = o_sir + sizeof(float) * (size_t)B * (S > 1 ? S - 1 : 0), bytes = o_snr + sizeof(float) * (size_t)B;
    std::vector<char> tab(bytes);
    memcpy(tab.data(), speakers, o_np);
    memcpy(tab.data() + o_np, noises, o_sir - o_np);
    if (S > 1) memcpy(tab.data() + o_sir, sirs, o_snr - o_sir);
    memcpy(tab.data() + o_snr, snrs, sizeof(float) * (size_t)B);
    void* dtab;
    if ((rc = upload_small(c, WS_META, tab.data(), bytes, stream, &dtab))) return rc;
    const float* const* d_sp = (const float* const*)dtab;
    const float* const* d_np = (const float* const*)((char*)dtab + o_np);
    const float* d_sir = (const float*)((char*)dtab + o_sir);
    const float* d_snr = (const float*)((char*)dtab + o_snr);
    const int64_t cn = (int64_t)C * n;
    const int nb = grid_for(cn, 64);
    if ((rc = ws_ensure(c, WS_SCR, sizeof(double) * ((size_t)B * S + (size_t)B * 2 * nb)))) return rc;
    if ((rc = ws_ensure(c, WS_SCR2, sizeof(float) * (size_t)B * (S + 1)))) return rc;
    double* d_ssq = (double*)c->ws[WS_SCR];
    double* d_part = d_ssq + (size_t)B * S;
    float* d_g = (float*)c->ws[WS_SCR2];
    hipLaunchKernelGGL(k_crop_sumsq, dim3(B * S), dim3(1024), 0, stream, d_sp, C, chan_stride, n, d_ssq);
    hipLaunchKernelGGL(k_bmix_gains1, dim3(B), dim3(64), 0, stream, (const double*)d_ssq, S, (double)cn, d_sir, d_g);
    hipLaunchKernelGGL(k_bmix_scale_sum, dim3(nb, B), dim3(256), 0, stream, d_sp, d_np, S, N, C, chan_stride, n, (const float*)d_g, speakers_out, mix_out,
                       d_part);
    hipLaunchKernelGGL(k_bmix_gains2, dim3(B), dim3(128), 0, stream, (const double*)d_part, nb, (double)cn, d_snr, d_g, S);
    hipLaunchKernelGGL(k_bmix_final, dim3(nb, B), dim3(256), 0, stream, d_np, N, C, chan_stride, n, (const float*)d_g, S, mix_out);
    HIPCHK(hipGetLastError());
    if (gains_out) {
        std::vector<float> g((size_t)B * (S + 1));
        HIPCHK(hipMemcpyAsync(g.data(), d_g, sizeof(float) * g.size(), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int b = 0; b < B; ++b)
            for (int i = 1; i <= S; ++i) gains_out[(size_t)b * S + i - 1] = g[(size_t)b * (S + 1) + i];
    }
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// Row R, optional: image-source early reflections of a shoebox room added onto a synthetic bank (SURVEY.md section 8f, N4's second
// half -- "a geometric RIR model richer than K1 behind render_ir"; content is synthetic by definition, oracle/rir_synth.py::early_reflections).
// For source position p and microphone c, every image (n in [-N, N]^3, q in {0,1}^3) with 1 <= reflections <= N contributes
//   g = pat[p][c] * beta^reflections / max(d, 0.1),  d = |(1 - 2q) * src_p + 2 n * room - mic_c|,  at the fractional delay fs * d / 343,
// split linearly over the two neighbouring taps.  One workgroup per (p, c): the images are scattered into an LDS window in FIXED POINT
// (integer atomics commute: the result does not depend on the order), then added to the bank by one writer per tap.
struct EarlyArgs {
    const float* src;    // [P][3]
    const float* mic;    // [C][3]
    const float* pat;    // [P][C]
    float room[3];
    float beta, fs;
    int32_t P, C, L, order, W;
};

constexpr int EARLY_WMAX = 16384;
constexpr float EARLY_SCALE = 262144.0f;      // 2^18

__global__ __launch_bounds__(256) void k_rir_early(EarlyArgs a, float* __restrict__ bank) {
    __shared__ int win[EARLY_WMAX];
    const int p = blockIdx.x / a.C, c = blockIdx.x % a.C;
    for (int t = threadIdx.x; t < a.W; t += 256) win[t] = 0;
    __syncthreads();
    const float sx = a.src[3 * p], sy = a.src[3 * p + 1], sz = a.src[3 * p + 2];
    const float mx = a.mic[3 * c], my = a.mic[3 * c + 1], mz = a.mic[3 * c + 2];
    const float g0 = a.pat[(int64_t)p * a.C + c];
    const int K = 2 * a.order + 1, nimg = K * K * K * 8;
    for (int i = threadIdx.x; i < nimg; i += 256) {
        const int q = i & 7, m = i >> 3;
        const int nx = m % K - a.order, ny = (m / K) % K - a.order, nz = m / (K * K) - a.order;
        const int qx = q & 1, qy = (q >> 1) & 1, qz = q >> 2;
        const int refl = abs(nx - qx) + abs(nx) + abs(ny - qy) + abs(ny) + abs(nz - qz) + abs(nz);
        if (refl < 1 || refl > a.order) continue;
        const float dx = (1 - 2 * qx) * sx + 2.0f * nx * a.room[0] - mx;
        const float dy = (1 - 2 * qy) * sy + 2.0f * ny * a.room[1] - my;
        const float dz = (1 - 2 * qz) * sz + 2.0f * nz * a.room[2] - mz;
        const float d = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 0.1f);
        const float tau = a.fs * d / 343.0f;
        const int i0 = (int)floorf(tau);
        if (i0 + 1 >= a.W) continue;
        const float fr = tau - (float)i0;
        const float g = g0 * powf(a.beta, (float)refl) / d;
        atomicAdd(&win[i0], (int)lrintf(g * (1.0f - fr) * EARLY_SCALE));
        atomicAdd(&win[i0 + 1], (int)lrintf(g * fr * EARLY_SCALE));
    }
    __syncthreads();
    float* row = bank + ((int64_t)p * a.C + c) * a.L;
    for (int t = threadIdx.x; t < a.W; t += 256)
        if (win[t]) row[t] += (float)win[t] * (1.0f / EARLY_SCALE);
}

int ss_rir_early_add_f32(float* bank, int32_t P, int32_t C, int32_t L, float fs, const float* src, const float* mic, const float* pat,
                         const float* room, float beta, int32_t order, uint32_t flags, void* stream_) {
    if (!bank || !src || !mic || !pat || !room || P < 1 || C < 1 || L < 2 || !(fs > 0) || order < 0 || order > 6 || !(beta >= 0) || beta > 1 ||
        !(room[0] > 0) || !(room[1] > 0) || !(room[2] > 0))
        return fail(SS_EINVAL, "bad argument (order 0..6, 0 <= beta <= 1, positive room dimensions)");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "the bank must be a device pointer (SS_FLAG_DEVICE_PTR); src / mic / pat / room are host arrays");
    if (order == 0) return SS_OK;
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const size_t nb = sizeof(float) * ((size_t)3 * P + (size_t)3 * C + (size_t)P * C);
    std::vector<float> host((size_t)3 * P + (size_t)3 * C + (size_t)P * C);
    memcpy(host.data(), src, sizeof(float) * 3 * P);
    memcpy(host.data() + 3 * (size_t)P, mic, sizeof(float) * 3 * C);
    memcpy(host.data() + 3 * (size_t)P + 3 * (size_t)C, pat, sizeof(float) * (size_t)P * C);
    void* dmeta;
    if ((rc = upload_small(c, WS_META, host.data(), nb, stream, &dmeta))) return rc;
    EarlyArgs a;
    a.src = (const float*)dmeta;
    a.mic = a.src + 3 * (size_t)P;
    a.pat = a.mic + 3 * (size_t)C;
    a.room[0] = room[0]; a.room[1] = room[1]; a.room[2] = room[2];
    a.beta = beta; a.fs = fs; a.P = P; a.C = C; a.L = L; a.order = order; a.W = L < EARLY_WMAX ? L : EARLY_WMAX;
    hipLaunchKernelGGL(k_rir_early, dim3((unsigned)((int64_t)P * C)), dim3(256), 0, stream, a, bank);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

// enhancement/look2hear/datas/movingdatamodule_remix.py:136-146: crops of resident stems summed without gains --
// out[t] = (a_0[t] + a_1[t] + ...) + (b_0[t] + b_1[t] + ...), float32, left to right inside a group (torch.sum over the stack dim)
struct CropSumArgs {
    const float* src[8];
    int32_t na, nb;
};

__global__ __launch_bounds__(256) void k_crop_sum(CropSumArgs a, float* __restrict__ out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float sa = a.src[0][t];
    for (int i = 1; i < a.na; ++i) sa += a.src[i][t];
    if (a.nb > 0) {
        float sb = a.src[a.na][t];
        for (int i = 1; i < a.nb; ++i) sb += a.src[a.na + i][t];
        sa += sb;
    }
    out[t] = sa;
}

int ss_crop_sum_f32(const float* const* first, int32_t n_first, const float* const* second, int32_t n_second, int64_t n, float* out,
                    uint32_t flags, void* stream_) {
    if (!first || !out || n_first < 1 || n_second < 0 || n_first + n_second > 8 || n < 1 || (n_second > 0 && !second))
        return fail(SS_EINVAL, "bad argument (1..8 sources in all)");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR)");
    CropSumArgs a;
    a.na = n_first;
    a.nb = n_second;
    for (int i = 0; i < 8; ++i) a.src[i] = nullptr;
    for (int i = 0; i < n_first; ++i) { if (!first[i]) return fail(SS_EINVAL, "NULL source"); a.src[i] = first[i]; }
    for (int i = 0; i < n_second; ++i) { if (!second[i]) return fail(SS_EINVAL, "NULL source"); a.src[n_first + i] = second[i]; }
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    hipLaunchKernelGGL(k_crop_sum, dim3(grid_for(n)), dim3(256), 0, stream, a, out, n);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_overlap_audio_f32(const float* x, float* out, int64_t T, int64_t delay_samples, uint32_t flags, void* stream_) {
    if (!x || !out || T < 1 || delay_samples < 0 || x == out) return fail(SS_EINVAL, "bad argument");
    if (!(flags & SS_FLAG_DEVICE_PTR)) return fail(SS_EINVAL, "device pointers only (SS_FLAG_DEVICE_PTR)");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    hipLaunchKernelGGL(k_overlap_audio, dim3(grid_for(T)), dim3(256), 0, stream, x, out, T, delay_samples);
    HIPCHK(hipGetLastError());
    return SS_OK;
}

int ss_resample_f32(const float* x, int32_t rows, int64_t L, int32_t orig, int32_t nnew, int32_t width, const float* taps, const int32_t* first,
                    int32_t ntap, float* out, int64_t Lout, uint32_t flags, void* stream_) {
    if (!x || !out || !taps || !first || rows < 1 || L < 1 || orig < 1 || nnew < 1 || width < 0 || ntap < 1 || Lout < 1 || rows > 65535)
        return fail(SS_EINVAL, "bad argument");
    if (Lout > ((L + orig - 1) / orig + 1) * (int64_t)nnew) return fail(SS_EINVAL, "Lout = %lld is longer than the resampled signal", (long long)Lout);
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    const size_t tb = sizeof(float) * (size_t)ntap * nnew, fb = sizeof(int32_t) * (size_t)nnew;
    std::vector<char> tab(tb + fb);
    memcpy(tab.data(), taps, tb);
    memcpy(tab.data() + tb, first, fb);
    void* dtab;
    if ((rc = upload_small(c, WS_META, tab.data(), tb + fb, stream, &dtab))) return rc;
    const void* dx;
    if ((rc = stage_in(c, WS_X, x, sizeof(float) * (size_t)rows * L, dev, stream, &dx))) return rc;
    float* dout = out;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_Y, sizeof(float) * (size_t)rows * Lout))) return rc;
        dout = (float*)c->ws[WS_Y];
    }
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((Lout + 255) / 256), rows), dim3(256), 0, stream, (const float*)dx, L, (const float*)dtab,
                       (const int32_t*)((const char*)dtab + tb), ntap, orig, nnew, width, dout, Lout);
    HIPCHK(hipGetLastError());
    if (!dev) {
        HIPCHK(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)rows * Lout, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return SS_OK;
}

int ss_scale_f32(const float* in, float* out, int64_t n, float gain, double* sums_out, uint32_t flags, void* stream_) {
    if (n <= 0 || !in || !out) return fail(SS_EINVAL, "bad argument");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    hipStream_t stream = (hipStream_t)stream_;
    if ((rc = stream_enter(c, stream))) return rc;
    const bool dev = (flags & SS_FLAG_DEVICE_PTR) != 0;
    const void* din;
    if ((rc = stage_in(c, WS_Y, in, sizeof(float) * (size_t)n, dev, stream, &din))) return rc;
    float* dout = out;
    if (!dev) {
        if ((rc = ws_ensure(c, WS_X, sizeof(float) * (size_t)n))) return rc;
        dout = (float*)c->ws[WS_X];
    }
    if (!sums_out) {
        hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, stream, (const float*)din, dout, n, gain);
        HIPCHK(hipGetLastError());
    } else {
        const int nb = grid_for(n, 1024);
        if ((rc = ws_ensure(c, WS_SCR, sizeof(double) * (size_t)nb * 2))) return rc;
        if ((rc = ws_ensure(c, WS_SCR2, sizeof(double) * 2))) return rc;
        double* part = (double*)c->ws[WS_SCR];
        hipLaunchKernelGGL(k_scale_sums, dim3(nb), dim3(256), 0, stream, (const float*)din, dout, n, gain, (const double*)nullptr, part);
        hipLaunchKernelGGL(k_final_sum, dim3(2), dim3(64), 0, stream, (const double*)part, nb, (double*)c->ws[WS_SCR2]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sums_out, c->ws[WS_SCR2], sizeof(double) * 2, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    if (!dev) {
        HIPCHK(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return SS_OK;
}

#ifdef SS_DEBUG_CLK
int ss_debug_clk(unsigned long long* out) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_clk), sizeof(unsigned long long) * 12 * 2 * 256));
    return SS_OK;
}
#endif
int ss_prof_enable(int on) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipDeviceSynchronize());
    for (auto& e : c->evs) { c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b); }
    c->evs.clear();
    if (on && c->ev_pool.size() < 256)
        for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) c->ev_pool.push_back(e); }
    c->prof = on != 0;
    c->prof_every = on > 1 ? on : 1;
    memset(c->prof_seen, 0, sizeof(c->prof_seen));
    return SS_OK;
}

int ss_prof_seen(int kind, int64_t* launches) {
    if (!launches || kind < 0 || kind > 3) return fail(SS_EINVAL, "bad argument");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    *launches = c->prof_seen[kind];
    return SS_OK;
}

int ss_prof_read(int kind, int64_t* launches, double* total_ms) {
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipDeviceSynchronize());
    int64_t n = 0;
    double tot = 0.0;
    for (auto& e : c->evs) {
        if (e.kind != kind) continue;
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        tot += ms;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return SS_OK;
}

int ss_prof_list(int kind, double* ms_out, int64_t cap, int64_t* launches) {
    if (!launches || cap < 0 || (cap > 0 && !ms_out)) return fail(SS_EINVAL, "bad argument");
    Ctx* c;
    int rc = get_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipDeviceSynchronize());
    int64_t n = 0;
    for (auto& e : c->evs) {
        if (e.kind != kind) continue;
        if (n < cap) {
            float ms = 0.0f;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            ms_out[n] = ms;
        }
        ++n;
    }
    *launches = n;
    return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// CU-free scene gather (include/sonicsim_hip.h): IPC-shared result array on the root, copy-engine transfers from the other ranks
struct SsGather {
    bool root = false;
    char* base = nullptr;          // root: hipMalloc'ed; others: hipIpcOpenMemHandle
    int64_t num = 0, bytes = 0;
    hipStream_t copy = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int device = 0;
};

int ss_gather_create(void** handle, int64_t num_scenes, int64_t scene_bytes, void* ipc_handle_out) {
    if (!handle || !ipc_handle_out || num_scenes < 1 || scene_bytes < 1) return fail(SS_EINVAL, "bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= SS_IPC_HANDLE_BYTES, "IPC handle size");
    SsGather* g = new SsGather();
    g->root = true; g->num = num_scenes; g->bytes = scene_bytes;
    if (hipGetDevice(&g->device) != hipSuccess) { delete g; return fail(SS_ENODEV, "hipGetDevice failed"); }
    hipError_t e = hipMalloc((void**)&g->base, (size_t)num_scenes * (size_t)scene_bytes);
    if (e != hipSuccess) { delete g; return fail(SS_ENOMEM, "hipMalloc of the gather array (%lld bytes) failed: %s", (long long)(num_scenes * scene_bytes), hipGetErrorString(e)); }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, g->base);
    if (e != hipSuccess) { hipFree(g->base); delete g; return fail(SS_EHIP, "hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 needed on dmabuf-only hosts)", hipGetErrorString(e)); }
    memset(ipc_handle_out, 0, SS_IPC_HANDLE_BYTES);
    memcpy(ipc_handle_out, &h, sizeof(h));
    if (hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_out, hipEventDisableTiming) != hipSuccess) { hipFree(g->base); delete g; return fail(SS_EHIP, "stream / event creation failed"); }
    (void)hipEventRecord(g->ev_out, g->copy);          // a stream gets its hardware queue on first use (milliseconds): here, not inside the caller's first scene
    (void)hipStreamSynchronize(g->copy);
    *handle = g;
    return SS_OK;
}

int ss_gather_attach(void** handle, const void* ipc_handle_in, int64_t num_scenes, int64_t scene_bytes) {
    if (!handle || !ipc_handle_in || num_scenes < 1 || scene_bytes < 1) return fail(SS_EINVAL, "bad argument");
    SsGather* g = new SsGather();
    g->num = num_scenes; g->bytes = scene_bytes;
    if (hipGetDevice(&g->device) != hipSuccess) { delete g; return fail(SS_ENODEV, "hipGetDevice failed"); }
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle_in, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle((void**)&g->base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { delete g; return fail(SS_EHIP, "hipIpcOpenMemHandle failed: %s", hipGetErrorString(e)); }
    if (hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_out, hipEventDisableTiming) != hipSuccess) { hipIpcCloseMemHandle(g->base); delete g; return fail(SS_EHIP, "stream / event creation failed"); }
    (void)hipEventRecord(g->ev_out, g->copy);          // (first use of the copy stream: see ss_gather_create)
    (void)hipStreamSynchronize(g->copy);
    *handle = g;
    return SS_OK;
}

int ss_gather_slot(void* handle, int64_t scene, void** ptr) {
    SsGather* g = (SsGather*)handle;
    if (!g || !ptr || scene < 0 || scene >= g->num) return fail(SS_EINVAL, "bad argument");
    *ptr = g->base + scene * g->bytes;
    return SS_OK;
}

int ss_gather_put(void* handle, int64_t scene, const void* src, void* stream_) {
    SsGather* g = (SsGather*)handle;
    if (!g || !src || scene < 0 || scene >= g->num) return fail(SS_EINVAL, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipEventRecord(g->ev_in, stream));                    // the copy starts once the producer (the render on `stream`) has finished
    HIPCHK(hipStreamWaitEvent(g->copy, g->ev_in, 0));
    HIPCHK(hipMemcpyAsync(g->base + scene * g->bytes, src, (size_t)g->bytes, hipMemcpyDeviceToDevice, g->copy));      // copy engines, no kernel
    return SS_OK;
}

int ss_gather_wait_src(void* handle, void* stream_) {
    SsGather* g = (SsGather*)handle;
    if (!g) return fail(SS_EINVAL, "bad argument");
    HIPCHK(hipEventRecord(g->ev_out, g->copy));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream_, g->ev_out, 0));
    return SS_OK;
}

int ss_gather_flush(void* handle) {
    SsGather* g = (SsGather*)handle;
    if (!g) return fail(SS_EINVAL, "bad argument");
    HIPCHK(hipStreamSynchronize(g->copy));
    return SS_OK;
}

int ss_gather_close(void* handle) {
    SsGather* g = (SsGather*)handle;
    if (!g) return SS_OK;
    (void)hipStreamSynchronize(g->copy);
    if (g->base) { if (g->root) (void)hipFree(g->base); else (void)hipIpcCloseMemHandle(g->base); }
    if (g->copy) (void)hipStreamDestroy(g->copy);
    if (g->ev_in) (void)hipEventDestroy(g->ev_in);
    if (g->ev_out) (void)hipEventDestroy(g->ev_out);
    delete g;
    return SS_OK;
}

}  // extern "C"
