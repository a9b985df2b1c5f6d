// stream13.h -- streaming moving-source render with PERSISTENT state ("chunked real-time rendering with
// persistent x-history"; round 4).
//
// A push of n new dry samples must cost O(n + state touched), not O(L).  The state that makes that possible lives in HBM:
//   * Hs  -- the partition spectra of the filter rows the trajectory is currently between (4 row slots x C x NP x 32 KB): a row is
//            transformed ONCE, when the trajectory reaches the segment before it, not once per push;
//   * Xr  -- a ring of the final input spectra X_m of the completed blocks (window x[(m-1)B, (m+1)B) of the global block grid);
//   * xh  -- the dry signal received so far (the current, incomplete block's window is re-transformed per push: ONE transform).
// A push is cut into pieces that lie inside one block AND one trajectory segment (usually one piece).  Per piece and channel, one
// workgroup forms the current block's spectrum X_j from xh + the new samples (future samples = 0: outputs up to the newest sample do
// not depend on them), accumulates  A_e = sum_p X_{j-p} . H_{row k+e, p}  for the segment's two rows e = 0, 1 (X loaded once for both),
// runs the two inverse transforms and stores  y[t] = (1 - w[t]) a_0[t] + w[t] a_1[t]  for the piece's samples -- the reference's
// gather + lerp (SonicSim_moving.py:89-94) with its bit-exact ramp w = float(double(t - s_k) * (1.0 / double(n_k))) (:43: np.linspace multiplies by the step, it does not divide).  No atomics, no
// zero fill, deterministic.  Same transforms, slot order and tables as geometry 13 (tvfir13.h), so the bodies below also run on the
// CPU workgroup emulator (tests/emul).
#pragma once
#include "tvfir13.h"

namespace ss {

constexpr int STREAM_ROW_SLOTS = 4;

struct StreamDev {
    const float* bank;      // [P][C][L]
    int32_t P, C, L, NP, NR;
    c32* Hs;                // [STREAM_ROW_SLOTS][C][NP][B13]
    c32* Xr;                // [NR][B13]
    float* xh;              // [total]
    const c32* consts;      // geometry-13 table (plan.h build_consts13)
};

struct StreamPiece {
    int64_t pos;            // absolute index of the piece's first sample
    int32_t n;              // samples in the piece (inside block j and segment k)
    int32_t j;              // block of the global grid: [j B, (j + 1) B)
    int32_t k;              // trajectory segment: rows k (start filter, coefficient 1 - w) and k + 1 (end filter, w)
    int64_t seg_start;      // s_k
    double inv_len;         // 1 / n_k
    const float* chunk;     // the piece's new samples (chunk[i] = x[pos + i])
    float* out;             // [C][out_stride], the piece starts at column out_off
    int64_t out_stride, out_off;
};

// host-side cut of a push: the next piece starting at `pos` with at most `n` samples left (plain C++: shared with the emulator).
// seg_start[P] (last == total).  Returns false when pos is past the schedule.
inline bool stream_next_piece(const int64_t* seg_start, int P, int64_t pos, int64_t n, int& k /* in: a segment <= the right one */, int64_t& len) {
    while (k < P - 2 && seg_start[k + 1] <= pos) ++k;
    if (pos >= seg_start[P - 1]) return false;
    const int64_t j = pos / B13;
    len = n;
    if ((j + 1) * (int64_t)B13 - pos < len) len = (j + 1) * (int64_t)B13 - pos;
    if (seg_start[k + 1] - pos < len) len = seg_start[k + 1] - pos;
    return len > 0;
}

// the forward transform of geometry 13 from a folded window (lo = first half, hi = second half; a filter partition has hi = 0):
// z[n] = (lo - i hi) exp(-i pi n / 8192), B-point transform, result in slot order.  Twiddles straight from the (L2-resident) table.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float ss_mul_rn(float a, float b) {      // a product ROUNDED on its own: the compiler contracts `a * b` (and __fmul_rn(a, b)) with
    float r;                                                           // a following addition into one FMA
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
inline float ss_mul_rn(float a, float b) { volatile float p = a * b; return p; }
#endif
struct Tw13 {
    c32 tw1[8], tw2v[7], tw3v[7];
};
template <class Env> SS_HD void fwd13_twiddles(Env& env, const c32* consts, Tw13& t) {
    const int tid = env.tid();
    const int lane = tid & 63, n4 = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) t.tw1[k] = consts[TW1P_13 + k * 512 + tid];
#pragma unroll
    for (int k = 1; k < 8; ++k) { t.tw2v[k - 1] = consts[TW2_13 + (k - 1) * 64 + lane]; t.tw3v[k - 1] = consts[TW3_13 + (k - 1) * 8 + n4]; }
}
// the four passes on v[n1] = twisted window samples n1 * 512 + tid (see fwd13_lohi); result in slot order
// (LdsT: where the exchange regions lie -- Lds13, or the compact LdsFwd13 (tvfir13.h) of kernels that only transform)
template <class Env, class LdsT = Lds13> SS_HD void fwd13_core(Env& env, const Tw13& t, c32 (&v)[8]) {
    const int tid = env.tid();
    LdsT l; l.base = env.lds();
    const int wave = tid >> 6, lane = tid & 63;
    const int k2 = lane >> 3, n4 = lane & 7;
    c32* Pv = l.priv(wave);
    dft8f<false>(v);
    c32* C = l.cross(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) C[k * 512 + tid] = cmul(v[k], t.tw1[k]);
    env.barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = C[wave * 512 + n * 64 + lane];
    dft8f<false>(v);
    Pv[lane] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[k * 72 + lane] = cmul(v[k], t.tw2v[k - 1]);
    env.wave_sync();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[k2 * 72 + n * 8 + n4];
    dft8f<false>(v);
    env.wave_sync();
    Pv[(k2 * 8) * 9 + n4] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[(k2 * 8 + k) * 9 + n4] = cmul(v[k], t.tw3v[k - 1]);
    env.wave_sync();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[lane * 9 + n];
    dft8f<false>(v);
    env.wave_sync();
}
template <class Env> SS_HD void fwd13_lohi(Env& env, const float (&lo)[8], const float (&hi)[8], const c32* consts, c32 (&v)[8]) {
    Tw13 t;
    fwd13_twiddles(env, consts, t);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
        v[n1] = mk(lo[n1] * SS_C16(n1) - hi[n1] * SS_S16(n1), -(lo[n1] * SS_S16(n1)) - hi[n1] * SS_C16(n1));
    fwd13_core(env, t, v);
}

SS_HD void stream_store_slots(c32* out, int tid, const c32 (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[(q * 512 + tid) * 2 + 0] = v[2 * q];
        out[(q * 512 + tid) * 2 + 1] = v[2 * q + 1];
    }
}

SS_HD void stream_load_slots(const c32* in, int tid, c32 (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f4 a = *reinterpret_cast<const f4*>(in + (q * 512 + tid) * 2);
        v[2 * q] = mk(a.x, a.y);
        v[2 * q + 1] = mk(a.z, a.w);
    }
}

// filter row -> its NP partition spectra in row slot (row & 3).  Workgroup (p, c).
template <class Env> SS_HD void stream_row_body(Env& env, const StreamDev& a, int row, int p, int c) {
    const int tid = env.tid();
    float lo[8], hi[8];
    const float* h = a.bank + ((int64_t)row * a.C + c) * a.L;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int64_t i = (int64_t)p * B13 + n1 * 512 + tid;
        lo[n1] = i < a.L ? h[i] : 0.0f;
        hi[n1] = 0.0f;
    }
    c32 v[8];
    fwd13_lohi(env, lo, hi, a.consts, v);
    stream_store_slots(a.Hs + (((int64_t)(row & (STREAM_ROW_SLOTS - 1)) * a.C + c) * a.NP + p) * B13, tid, v);
}

// Partition spectra of one filter row for the render kernel's spectra-ready tasks (plan.h flag_long_rows): PPW consecutive partitions per
// workgroup -- the taps of all of them are requested up front (one HBM round trip), the twiddles are fetched once.  h = taps of (row, channel),
// out = the slot's [NP][B13] array.
template <int PPW, class Env> SS_HD void row_spectra_body(Env& env, const float* h, int L, int NP, int p0, const c32* consts, c32* out) {
    const int tid = env.tid();
    float tap[PPW][8];
#pragma unroll
    for (int i = 0; i < PPW; ++i)
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int64_t k = (int64_t)(p0 + i) * B13 + n1 * 512 + tid;
            tap[i][n1] = (p0 + i < NP && k < L) ? h[k] : 0.0f;
        }
    Tw13 t;
    fwd13_twiddles(env, consts, t);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        if (p0 + i >= NP) break;
        c32 v[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1)      // two ROUNDED products, as the render kernel's own pass 1 forms them (v_mul_f32): left to the compiler they fuse
            v[n1] = mk(ss_mul_rn(tap[i][n1], SS_C16(n1)), -ss_mul_rn(tap[i][n1], SS_S16(n1)));      // into the first butterfly's additions and the
        fwd13_core<Env, LdsFwd13>(env, t, v);                                                        // spectra differ from the kernel's in the last bits
        stream_store_slots(out + (int64_t)(p0 + i) * B13, tid, v);
        env.barrier();                                  // the cross buffer is rewritten by the next partition's first pass
    }
}

// one piece of a push, channel c (workgroup c of a grid of C)
template <class Env> SS_HD void stream_push_body(Env& env, const StreamDev& a, const StreamPiece& pc, int c) {
    const int tid = env.tid();
    Lds13 l; l.base = env.lds();
    // ---- the current block's window from the history + the new samples (future samples are zero)
    float lo[8], hi[8];
    const int64_t tb = (int64_t)pc.j * B13, tend = pc.pos + pc.n;
    auto xat = [&](int64_t t) -> float {
        if (t < 0 || t >= tend) return 0.0f;
        return t < pc.pos ? a.xh[t] : pc.chunk[t - pc.pos];
    };
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = n1 * 512 + tid;
        lo[n1] = xat(tb - B13 + n);
        hi[n1] = xat(tb + n);
    }
    c32 X[8];
    fwd13_lohi(env, lo, hi, a.consts, X);
    // tables for the inverse transforms into LDS (the forward pass used registers; its exchange regions do not overlap the tables)
    for (int i = tid; i < CONST13_C32; i += NT13) l.base[i] = a.consts[i];
    if (c == 0) {
        stream_store_slots(a.Xr + (int64_t)(pc.j % a.NR) * B13, tid, X);        // final once the block is complete; rewritten until then
        for (int64_t i = tid; i < pc.n; i += NT13) a.xh[pc.pos + i] = pc.chunk[i];
    }
    // ---- A_e = sum_p X_{j-p} H_{k+e, p}
    c32 acc0[8], acc1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc0[r] = acc1[r] = mk(0.0f, 0.0f);
    const int npe = a.NP < pc.j + 1 ? a.NP : pc.j + 1;
    const c32* H0 = a.Hs + ((int64_t)(pc.k & (STREAM_ROW_SLOTS - 1)) * a.C + c) * a.NP * B13;
    const c32* H1 = a.Hs + ((int64_t)((pc.k + 1) & (STREAM_ROW_SLOTS - 1)) * a.C + c) * a.NP * B13;
    for (int p = 0; p < npe; ++p) {
        c32 h0[8], h1[8];
        stream_load_slots(H0 + (int64_t)p * B13, tid, h0);
        stream_load_slots(H1 + (int64_t)p * B13, tid, h1);
        if (p > 0) stream_load_slots(a.Xr + (int64_t)((pc.j - p) % a.NR) * B13, tid, X);
#pragma unroll
        for (int r = 0; r < 8; ++r) { cmac_a(acc0[r], X[r], h0[r]); cmac_a(acc1[r], X[r], h1[r]); }
#pragma unroll
        for (int r = 0; r < 8; ++r) { cmac_b(acc0[r], X[r], h0[r]); cmac_b(acc1[r], X[r], h1[r]); }
    }
    env.barrier();                                  // tables are in LDS
    g13_inv(env, l, acc0, 0);
    env.barrier();
    g13_inv(env, l, acc1, 1);
    // ---- gather + lerp of SonicSim_moving.py:89-94 for the piece's samples
    const float scale = 1.0f / (float)B13;
    float* out = pc.out + (int64_t)c * pc.out_stride + pc.out_off;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int64_t t = tb + n1 * 512 + tid;
        if (t >= pc.pos && t < tend) {
            const float v0 = -(acc0[n1].x * SS_S16(n1) + acc0[n1].y * SS_C16(n1)) * scale;
            const float v1 = -(acc1[n1].x * SS_S16(n1) + acc1[n1].y * SS_C16(n1)) * scale;
            const float w = (float)((double)(t - pc.seg_start) * pc.inv_len);
            const float s0 = (1.0f - w) * v0, s1 = w * v1;
            out[t - pc.pos] = s0 + s1;
        }
    }
}

}  // namespace ss
