// tvfir13.h -- geometry 13 of the row-stationary overlap-save render (B = 4096, 512 threads, persistent).
//
// Same algorithm, slot order and X-spectra layout as geometry 12 (tvfir_core.h); what changes is the schedule:
//   * software pipeline: the four block MACs of partition p are interleaved into the LDS round trips of the
//     forward transform of partition p+1 (the wave has independent VALU work while its exchange is in flight);
//   * ONE workgroup barrier per transform: the pass-1 -> pass-2 cross-wave buffer is double buffered;
//   * no register-to-register window copies: the four window slots rotate (4x unrolled partition loop) and the
//     ONE new spectrum per partition is loaded straight into the slot that was just consumed;
//   * the right-angle twist exp(-i pi n / 8192), n = n1*512 + tid, is split into a per-thread factor that is
//     merged into the pass-1 twiddle table (TW1P) and a per-n1 literal: no twist table, 36 KB of constants;
//   * taps, spectra and output go through buffer instructions: the descriptor's num_records does the tail /
//     out-of-range masking (loads return 0, atomics are dropped), so there is no address or select VALU;
//   * radix-8 butterflies fold the 1/sqrt(2) scaling into fused multiply-adds (26 packed instructions);
//   * dynamic task queue (one atomic per task, fetched a whole task ahead) and next-task prefetch under the
//     inverse transforms.
// Reference arithmetic: SonicSim-SonicSet/SonicSim_moving.py:86-94 (see tvfir_core.h).
#pragma once
#include "tvfir_core.h"

namespace ss {

constexpr int B13 = 4096;
constexpr int NT13 = 512;
constexpr int JMAX13 = 4;
constexpr int TW1P_13 = 0;        // [8][512]  exp(-i pi t/8192) * W_4096^(t*k),  k = 0..7
constexpr int TW2_13 = 4096;      // [7][64]   W_512^(m*k),  k = 1..7
constexpr int TW3_13 = 4544;      // [7][8]    W_64^(n*k),   k = 1..7
constexpr int CONST13_C32 = 4608;
constexpr int CROSS13_OFF = CONST13_C32;                 // 2 x [4096] cross-wave exchange (double buffered)
constexpr int PRIV13_OFF = CROSS13_OFF + 2 * 4096;       // 8 x [576]  wave-private exchange regions
constexpr int MISC13_OFF = PRIV13_OFF + 8 * 576;         // next-task mailbox
constexpr int LDS13_C32 = MISC13_OFF + 8;                // 17416 c32 = 139328 bytes

struct Lds13 {
    c32* base;
    SS_HD const c32* tw1p() const { return base + TW1P_13; }
    SS_HD const c32* tw2() const { return base + TW2_13; }
    SS_HD const c32* tw3() const { return base + TW3_13; }
    SS_HD c32* cross(int par) const { return base + CROSS13_OFF + (par & 1) * 4096; }
    SS_HD c32* priv(int wave) const { return base + PRIV13_OFF + wave * 576; }
    SS_HD int* mailbox() const { return reinterpret_cast<int*>(base + MISC13_OFF); }
};

// kernels that only run forward transforms (input spectra, row spectra) need one cross buffer and the wave-private regions: 70 KB,
// two workgroups per CU
struct LdsFwd13 {
    c32* base;
    SS_HD c32* cross(int) const { return base; }
    SS_HD c32* priv(int wave) const { return base + 4096 + wave * 576; }
};
constexpr int LDSFWD13_C32 = 4096 + 8 * 576;

// ---------------------------------------------------------------------------------------------
// two packed spectrum bins as loaded (16 bytes per lane)
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c32 f4lo(f4v a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ c32 f4hi(f4v a) { return __builtin_shufflevector(a, a, 2, 3); }
#else
typedef f4 f4v;
SS_HD c32 f4lo(const f4v& a) { return mk(a.x, a.y); }
SS_HD c32 f4hi(const f4v& a) { return mk(a.z, a.w); }
#endif

// buffer access: descriptor = (base, valid bytes).  Loads beyond `valid` return 0, atomics beyond it are dropped.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t BufRes;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BufRes mk_buf(const void* base, uint32_t nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)nbytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld_f32(BufRes r, uint32_t off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ f4v buf_ld_f4(BufRes r, uint32_t voff, uint32_t soff) {
    const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(f4v, q);
}
__device__ __forceinline__ void buf_atomic_add_f32(BufRes r, uint32_t off, float v) { __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, off, 0, 0); }
__device__ __forceinline__ int fetch_add_i32(int* p) { return atomicAdd(p, 1); }
#else
struct BufRes {
    const char* base;
    uint32_t n;
};
SS_HD BufRes mk_buf(const void* base, uint32_t nbytes) { return BufRes{static_cast<const char*>(base), nbytes}; }
SS_HD float buf_ld_f32(BufRes r, uint32_t off) { return (uint64_t)off + 4 <= r.n ? *reinterpret_cast<const float*>(r.base + off) : 0.0f; }
SS_HD f4v buf_ld_f4(BufRes r, uint32_t voff, uint32_t soff) {
    f4 o{0.f, 0.f, 0.f, 0.f};
    if ((uint64_t)voff + soff + 16 <= r.n) o = *reinterpret_cast<const f4*>(r.base + voff + soff);
    return o;
}
SS_HD void buf_atomic_add_f32(BufRes r, uint32_t off, float v) {
    if ((uint64_t)off + 4 <= r.n) *reinterpret_cast<float*>(const_cast<char*>(r.base) + off) += v;
}
SS_HD int fetch_add_i32(int* p) { return __atomic_fetch_add(p, 1, __ATOMIC_RELAXED); }
#endif

// ---------------------------------------------------------------------------------------------
// fused complex helpers: c + s*U, c - s*U, c -+ i*s*V with S = (s, s)
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_PKF(name, text)                                                                        \
    __device__ __forceinline__ c32 name(c32 u, c32 s, c32 c) {                                     \
        c32 r;                                                                                     \
        asm("v_pk_fma_f32 %0, %1, %2, %3 " text : "=v"(r) : "v"(u), "v"(s), "v"(c));              \
        return r;                                                                                  \
    }
SS_PKF(cfma_s, "")                                                          // c + s U
SS_PKF(cfms_s, "neg_lo:[1,0,0] neg_hi:[1,0,0]")                             // c - s U
SS_PKF(cfma_mi_s, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")        // c + (-i) s U = (c.x + s U.y, c.y - s U.x)
SS_PKF(cfma_pi_s, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")        // c + (+i) s U = (c.x - s U.y, c.y + s U.x)
#undef SS_PKF
#else
SS_HD c32 cfma_s(c32 u, c32 s, c32 c) { return mk(c.x + s.x * u.x, c.y + s.y * u.y); }
SS_HD c32 cfms_s(c32 u, c32 s, c32 c) { return mk(c.x - s.x * u.x, c.y - s.y * u.y); }
SS_HD c32 cfma_mi_s(c32 u, c32 s, c32 c) { return mk(c.x + s.x * u.y, c.y - s.y * u.x); }
SS_HD c32 cfma_pi_s(c32 u, c32 s, c32 c) { return mk(c.x - s.x * u.y, c.y + s.y * u.x); }
#endif

// in-place complex multiply-accumulate, split in two half-steps so that a block's 8 first halves can be issued before its
// 8 second halves (the dependent pair then never needs a wait state): acc += x * h
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void cmac_a(c32& acc, c32 x, c32 h) {   // acc += (x.x h.x, x.x h.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(h));
}
__device__ __forceinline__ void cmac_b(c32& acc, c32 x, c32 h) {   // acc += (-x.y h.y, x.y h.x)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(x), "v"(h));
}
#else
SS_HD void cmac_a(c32& acc, c32 x, c32 h) { acc = mk(acc.x + x.x * h.x, acc.y + x.x * h.y); }
SS_HD void cmac_b(c32& acc, c32 x, c32 h) { acc = mk(acc.x - x.y * h.y, acc.y + x.y * h.x); }
#endif

// 8-point DFT, 26 packed instructions (the W8 scalings ride on fused multiply-adds)
template <bool INV> SS_HD void dft8f(c32* v) {
    const c32 S = mk(0.70710678118654752440f, 0.70710678118654752440f);
    const c32 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const c32 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    const c32 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    const c32 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    {
        const c32 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), d = csub(a1, a3);
        v[0] = cadd(c0, c2);
        v[4] = csub(c0, c2);
        v[2] = INV ? csub_mi(c1, d) : cadd_mi(c1, d);
        v[6] = INV ? cadd_mi(c1, d) : csub_mi(c1, d);
    }
    {
        // unscaled odd-branch rotations: t5 = a5 (1 -+ i), t7 = a7 (-1 -+ i); the common 1/sqrt(2) is applied by the FMAs
        const c32 t5 = INV ? csub_mi(a5, a5) : cadd_mi(a5, a5);
        const c32 t7 = INV ? nsub_mi2(a7, a7) : nadd_mi2(a7, a7);
        const c32 c0 = INV ? csub_mi(a4, a6) : cadd_mi(a4, a6);
        const c32 c1 = INV ? cadd_mi(a4, a6) : csub_mi(a4, a6);
        const c32 u = cadd(t5, t7), w = csub(t5, t7);
        v[1] = cfma_s(u, S, c0);
        v[5] = cfms_s(u, S, c0);
        v[3] = INV ? cfma_pi_s(w, S, c1) : cfma_mi_s(w, S, c1);
        v[7] = INV ? cfma_mi_s(w, S, c1) : cfma_pi_s(w, S, c1);
    }
}

// exp(-i pi n1 / 16) = (C16, -S16)
#define SS_C16(n) ((n) == 0 ? 1.0f : (n) == 1 ? 0.98078528040323044913f : (n) == 2 ? 0.92387953251128675613f : (n) == 3 ? 0.83146961230254523708f : \
                   (n) == 4 ? 0.70710678118654752440f : (n) == 5 ? 0.55557023301960222474f : (n) == 6 ? 0.38268343236508977173f : 0.19509032201612826785f)
#define SS_S16(n) ((n) == 0 ? 0.0f : (n) == 1 ? 0.19509032201612826785f : (n) == 2 ? 0.38268343236508977173f : (n) == 3 ? 0.55557023301960222474f : \
                   (n) == 4 ? 0.70710678118654752440f : (n) == 5 ? 0.83146961230254523708f : (n) == 6 ? 0.92387953251128675613f : 0.98078528040323044913f)

#if defined(SS_G13_ALLMAC)
#define SS_NJ(j) true
#else
#define SS_NJ(j) (tk.nj > (j))
#endif

struct Params13 {
    RenderParams r;        // x/T/bank/P/C/L/NP/Xs/M/consts(->13 table)/tasks/ntasks/mode/seg_start/idx/w/y as in geometry 12
    int* counter;          // dynamic task queue head (zeroed by the spectra kernel of the same render)
    int32_t nwg;           // workgroups launched
};

template <class Env> SS_HD void load_consts13(Env& env, const Lds13& l, const c32* consts) {
    const int tid = env.tid();
    for (int i = tid; i < CONST13_C32; i += NT13) l.base[i] = consts[i];
    env.barrier();
}

// per-thread register state of one task
struct State13 {
    c32 acc[JMAX13][8];
    f4v W[JMAX13][4];      // window slots: at partition p block j uses slot (j - p) & 3
    c32 hs[8];             // spectrum of the partition whose MACs are pending
    float t[8];            // taps of the next partition to transform
};

// forward pass 1 (radix-8 over n1, stride 512) of partition q: taps -> cross buffer q&1
template <class Env> SS_HD void g13_pass1(Env& env, const Lds13& l, const float (&t)[8], int par) {
    const int tid = env.tid();
    c32 v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = mk(t[n1] * SS_C16(n1), -(t[n1] * SS_S16(n1)));
    dft8f<false>(v);
    c32* C = l.cross(par);
#pragma unroll
    for (int k = 0; k < 8; ++k) C[k * 512 + tid] = cmul(v[k], l.tw1p()[k * 512 + tid]);
}

// MAC of one block: acc[j] += X (window slot) * hs
SS_HD void g13_mac(c32 (&acc)[8], const f4v (&X)[4], const c32 (&hs)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cmac_a(acc[2 * q], f4lo(X[q]), hs[2 * q]);
        cmac_a(acc[2 * q + 1], f4hi(X[q]), hs[2 * q + 1]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cmac_b(acc[2 * q], f4lo(X[q]), hs[2 * q]);
        cmac_b(acc[2 * q + 1], f4hi(X[q]), hs[2 * q + 1]);
    }
}

SS_HD void g13_ldX(f4v (&X)[4], BufRes xs, uint32_t voff) {
#pragma unroll
    for (int q = 0; q < 4; ++q) X[q] = buf_ld_f4(xs, voff, (uint32_t)q * 8192u);
}

struct Task13 {
    BufRes row;            // whole filter row (L*4 bytes valid)
    const char* rowp;
    int32_t j0, nj, np, chan, rowi;
    uint32_t rowbytes;
};

// descriptor of spectrum m (zeros outside [0, M))
template <class P> SS_HD BufRes g13_xdesc(const P& prm, int m) {
    const bool ok = m >= 0 && m < prm.r.M;
    const char* base = reinterpret_cast<const char*>(prm.r.Xs) + (ok ? (int64_t)m * (B13 * 8) : 0);
    return mk_buf(base, ok ? (uint32_t)(B13 * 8) : 0u);
}
// descriptor of partition q of the row: base advanced by q*16 KB, valid = bytes left in the row
SS_HD BufRes g13_tdesc(const Task13& tk, int q) {
    const int64_t off = (int64_t)q * (B13 * 4);
    const int64_t left = (int64_t)tk.rowbytes - off;
    return mk_buf(tk.rowp + (left > 0 ? off : 0), left > 0 ? (uint32_t)left : 0u);
}
SS_HD void g13_ldtaps(float (&t)[8], BufRes d, uint32_t voff) {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) t[n1] = buf_ld_f32(d, voff + (uint32_t)n1 * 2048u);
}

// One pipelined iteration: [FFT] forward transform of partition q (taps in s.t) -> s.hs, and
// [MAC] the four block MACs of partition q-1 (spectrum in s.hs, rotation phase PH = (q-1) & 3) in its LDS shadows.
template <class Env, class P, int PH, bool FFT, bool MAC>
SS_HD void g13_iter(Env& env, const Lds13& l, const P& prm, const Task13& tk, State13& s, int q, int& par, uint32_t tvoff, uint32_t xvoff) {
    const int tid = env.tid();
    const int wave = tid >> 6, lane = tid & 63;
    c32* Pv = l.priv(wave);
    c32 v[8];
    if (FFT) {
        g13_pass1(env, l, s.t, par);
        if (q + 1 < tk.np) g13_ldtaps(s.t, g13_tdesc(tk, q + 1), tvoff);       // next partition's taps: a whole iteration of cover
    }
    if (MAC) {
        if (tk.nj > 3) g13_mac(s.acc[3], s.W[(3 - PH) & 3], s.hs);
        // the slot block 3 just used is dead: load the ONE new spectrum the next partition needs (block 0 at p+1)
        if (q < tk.np) g13_ldX(s.W[(3 - PH) & 3], g13_xdesc(prm, tk.j0 - q), xvoff);
    }
    if (FFT) {
        env.barrier();
        const c32* C = l.cross(par);
        par ^= 1;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = C[wave * 512 + n * 64 + lane];
    }
    if (MAC && tk.nj > 2) g13_mac(s.acc[2], s.W[(2 - PH) & 3], s.hs);
    if (FFT) {
        dft8f<false>(v);
        Pv[lane] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) Pv[k * 72 + lane] = cmul(v[k], l.tw2()[(k - 1) * 64 + lane]);
        env.wave_sync();
        const int k2 = lane >> 3, n4 = lane & 7;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = Pv[k2 * 72 + n * 8 + n4];
    }
    if (MAC && tk.nj > 1) g13_mac(s.acc[1], s.W[(1 - PH) & 3], s.hs);
    if (FFT) {
        const int k2 = lane >> 3, n4 = lane & 7;
        dft8f<false>(v);
        env.wave_sync();
        Pv[(k2 * 8) * 9 + n4] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) Pv[(k2 * 8 + k) * 9 + n4] = cmul(v[k], l.tw3()[(k - 1) * 8 + n4]);
        env.wave_sync();
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = Pv[lane * 9 + n];
    }
    if (MAC) g13_mac(s.acc[0], s.W[(0 - PH) & 3], s.hs);
    if (FFT) {
        dft8f<false>(v);
        env.wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) s.hs[r] = v[r];
    }
}

// Un-pipelined step (lower register pressure): forward transform of partition q, then its four block MACs (phase PH = q & 3)
template <class Env, class P, int PH>
SS_HD void g13_step(Env& env, const Lds13& l, const P& prm, const Task13& tk, State13& s, int q, int& par, uint32_t tvoff, uint32_t xvoff) {
    const int tid = env.tid();
    const int wave = tid >> 6, lane = tid & 63;
    c32* Pv = l.priv(wave);
    c32 v[8];
    g13_pass1(env, l, s.t, par);
    if (q + 1 < tk.np) g13_ldtaps(s.t, g13_tdesc(tk, q + 1), tvoff);
    env.barrier();
    {
        const c32* C = l.cross(par);
        par ^= 1;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = C[wave * 512 + n * 64 + lane];
    }
    dft8f<false>(v);
    Pv[lane] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[k * 72 + lane] = cmul(v[k], l.tw2()[(k - 1) * 64 + lane]);
    env.wave_sync();
    const int k2 = lane >> 3, n4 = lane & 7;
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[k2 * 72 + n * 8 + n4];
    dft8f<false>(v);
    env.wave_sync();
    Pv[(k2 * 8) * 9 + n4] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[(k2 * 8 + k) * 9 + n4] = cmul(v[k], l.tw3()[(k - 1) * 8 + n4]);
    env.wave_sync();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[lane * 9 + n];
    dft8f<false>(v);
    env.wave_sync();
    if (SS_NJ(3)) g13_mac(s.acc[3], s.W[(3 - PH) & 3], v);
    if (q + 1 < tk.np) g13_ldX(s.W[(3 - PH) & 3], g13_xdesc(prm, tk.j0 - q - 1), xvoff);
    if (SS_NJ(2)) g13_mac(s.acc[2], s.W[(2 - PH) & 3], v);
    if (SS_NJ(1)) g13_mac(s.acc[1], s.W[(1 - PH) & 3], v);
    g13_mac(s.acc[0], s.W[(0 - PH) & 3], v);
}

// Input spectra on the geometry-13 tables (36 KB of constants instead of 64 KB, the x window is requested before the
// constants are staged, 16-byte zero fill): window m = x[(m-1)B, (m+1)B) folded to z[n] = (lo - i hi) * exp(-i pi n / 8192),
// spectrum in the slot layout the render kernels read (c32 index ((r>>1)*512 + tid)*2 + (r&1)).
// xdiv != nullptr: the spectra are those of x / *xdiv -- the render is linear in x, so this is how a bank's deferred global
// peak normalisation (SonicSim_audio.py:398, ir_output /= ir_output.abs().max()) reaches the output without a pass over the bank
// rs > 0: spectra on a grid of B13 >> rs samples for the assembly engine's hop-aligned tasks (plan.h row_tasks): entry m is the
// window that starts at (m - (2^rs - 1)) * (B13 >> rs) - B13, i.e. the 2^rs - 1 windows that begin before -B13 + hop .. are kept too
// (they still cover samples of x); rs = 0 is the block grid the other engines use.
template <class Env, class LdsT = Lds13> SS_HD void xspec13_body(Env& env, const float* x, int64_t T, const c32* consts, c32* Xs, int m, int M,
                                             float* yzero, int64_t nzero, const float* xdiv = nullptr, int rs = 0, float fill = 0.0f) {
    const int tid = env.tid();
    float lo[8], hi[8];
    if (m < M) {
        const float xs = xdiv ? 1.0f / *xdiv : 1.0f;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int n = n1 * 512 + tid;
            const int64_t thi = ((int64_t)m - ((1 << rs) - 1)) * (B13 >> rs) + n, tlo = thi - B13;
            lo[n1] = (tlo >= 0 && tlo < T) ? x[tlo] : 0.0f;
            hi[n1] = (thi >= 0 && thi < T) ? x[thi] : 0.0f;
            if (xdiv) { lo[n1] *= xs; hi[n1] *= xs; }
        }
    }
    if (yzero) {   // this workgroup's slice of y (the render kernel accumulates with float atomics onto zero)
        const bool al = (reinterpret_cast<uintptr_t>(yzero) & 15) == 0;
        const int64_t n4 = al ? nzero / 4 : 0;
        const int64_t chunk = (n4 + M) / (M + 1);
        const int64_t zlo = (int64_t)m * chunk, zhi = zlo + chunk < n4 ? zlo + chunk : n4;
        f4* y4 = reinterpret_cast<f4*>(yzero);
        const f4 z{fill, fill, fill, fill};        // 0; NaN when the device planner latched an error (the output must not pass as valid silence)
        for (int64_t i = zlo + tid; i < zhi; i += NT13) y4[i] = z;        // (write-through `sc1` stores instead: measured, no gain -- profiles/r03i)
        if (m == 0)
            for (int64_t i = n4 * 4 + tid; i < nzero; i += NT13) yzero[i] = fill;
    }
    if (m >= M) return;              // (there is no zero spectrum any more: the render kernels' descriptors return zeros)
    LdsT l; l.base = env.lds();
    const int wave = tid >> 6, lane = tid & 63;
    const int k2 = lane >> 3, n4 = lane & 7;
    // this thread's 22 twiddles straight from the (L2-resident) table into registers: a workgroup forms ONE spectrum, so staging the
    // 36 KB table through LDS (nine load + store rounds and a barrier) only adds latency in front of the transform (round 3)
    c32 tw1[8], tw2v[7], tw3v[7];
#pragma unroll
    for (int k = 0; k < 8; ++k) tw1[k] = consts[TW1P_13 + k * 512 + tid];
#pragma unroll
    for (int k = 1; k < 8; ++k) { tw2v[k - 1] = consts[TW2_13 + (k - 1) * 64 + lane]; tw3v[k - 1] = consts[TW3_13 + (k - 1) * 8 + n4]; }
    c32* Pv = l.priv(wave);
    c32 v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
        v[n1] = mk(lo[n1] * SS_C16(n1) - hi[n1] * SS_S16(n1), -(lo[n1] * SS_S16(n1)) - hi[n1] * SS_C16(n1));
    dft8f<false>(v);
    c32* C = l.cross(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) C[k * 512 + tid] = cmul(v[k], tw1[k]);
    env.barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = C[wave * 512 + n * 64 + lane];
    dft8f<false>(v);
    Pv[lane] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[k * 72 + lane] = cmul(v[k], tw2v[k - 1]);
    env.wave_sync();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[k2 * 72 + n * 8 + n4];
    dft8f<false>(v);
    env.wave_sync();
    Pv[(k2 * 8) * 9 + n4] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) Pv[(k2 * 8 + k) * 9 + n4] = cmul(v[k], tw3v[k - 1]);
    env.wave_sync();
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = Pv[lane * 9 + n];
    dft8f<false>(v);
    c32* out = Xs + (int64_t)m * B13;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[(q * 512 + tid) * 2 + 0] = v[2 * q];
        out[(q * 512 + tid) * 2 + 1] = v[2 * q + 1];
    }
}

// inverse transform (slot order in, v[n1] = conj(tau[tid]) * z[n1*512 + tid] * B out); `par` picks the cross buffer
template <class Env> SS_HD void g13_inv(Env& env, const Lds13& l, c32* v, int par) {
    const int tid = env.tid();
    const int wave = tid >> 6, lane = tid & 63;
    c32* Pv = l.priv(wave);
    c32* C = l.cross(par);
    {
        dft8f<true>(v);
        c32* p0 = Pv + lane * 9;
#pragma unroll
        for (int n = 0; n < 8; ++n) p0[n] = v[n];
    }
    env.wave_sync();
    {
        const int k2 = lane >> 3, n4 = lane & 7;
        v[0] = Pv[(k2 * 8) * 9 + n4];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(Pv[(k2 * 8 + k) * 9 + n4], l.tw3()[(k - 1) * 8 + n4]);
        dft8f<true>(v);
        env.wave_sync();
#pragma unroll
        for (int n = 0; n < 8; ++n) Pv[k2 * 72 + n * 8 + n4] = v[n];
    }
    env.wave_sync();
    {
        v[0] = Pv[lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(Pv[k * 72 + lane], l.tw2()[(k - 1) * 64 + lane]);
        dft8f<true>(v);
#pragma unroll
        for (int n = 0; n < 8; ++n) C[wave * 512 + n * 64 + lane] = v[n];
    }
    env.wave_sync();
    env.barrier();
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = cmulc(C[k * 512 + tid], l.tw1p()[k * 512 + tid]);
        dft8f<true>(v);
    }
}

template <class P> SS_HD Task13 g13_task(const P& prm, int id) {
    const Task t = prm.r.tasks[id];
    Task13 tk;
    tk.rowi = t.row; tk.chan = t.chan; tk.j0 = t.j0; tk.nj = t.nj;
    tk.rowp = reinterpret_cast<const char*>(prm.r.bank + ((int64_t)t.row * prm.r.C + t.chan) * prm.r.L);
    tk.rowbytes = (uint32_t)prm.r.L * 4u;
    tk.row = mk_buf(tk.rowp, tk.rowbytes);
    int np = prm.r.NP;
    if (np > t.j0 + t.nj) np = t.j0 + t.nj;      // later partitions only meet windows before t = 0
    tk.np = np;
    return tk;
}

// loads that open a task: the window (slot s = X_{j0+s}, zeros for s >= nj) and the taps of partition 0
template <class P> SS_HD void g13_open(const P& prm, const Task13& tk, State13& s, uint32_t tvoff, uint32_t xvoff) {
#pragma unroll
    for (int j = 0; j < JMAX13; ++j) g13_ldX(s.W[j], g13_xdesc(prm, j < tk.nj ? tk.j0 + j : -1), xvoff);
    g13_ldtaps(s.t, g13_tdesc(tk, 0), tvoff);
}

template <class Env, class P, int PH> SS_HD void g13_tail(Env& env, const Lds13& l, const P& prm, const Task13& tk, State13& s, int& par, uint32_t tvoff, uint32_t xvoff) {
    g13_iter<Env, P, PH, false, true>(env, l, prm, tk, s, tk.np, par, tvoff, xvoff);
}

template <class Env, class P> SS_HD void os13_body(Env& env, const P& prm, int wg) {
    Lds13 l; l.base = env.lds();
    load_consts13(env, l, prm.r.consts);
    const int tid = env.tid();
    const uint32_t tvoff = (uint32_t)tid * 4u, xvoff = (uint32_t)tid * 16u;
    int id = wg;
    if (id >= prm.r.ntasks) return;
    State13 s;
    Task13 tk = g13_task(prm, id);
    g13_open(prm, tk, s, tvoff, xvoff);
    int par = 0;      // cross-buffer parity: toggles at EVERY transform (forward or inverse), see Lds13::cross
    for (;;) {
        // ticket for the NEXT task, a whole task ahead of its use
        int next_id = 0;
        if (tid == 0) next_id = prm.nwg + fetch_add_i32(prm.counter);
#pragma unroll
        for (int j = 0; j < JMAX13; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) s.acc[j][r] = mk(0.0f, 0.0f);

#if defined(SS_G13_NOPIPE)
        int q = 0;
        for (;;) {
            if (q >= tk.np) break;
            g13_step<Env, P, 0>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_step<Env, P, 1>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_step<Env, P, 2>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_step<Env, P, 3>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
        }
#else
        g13_iter<Env, P, 0, true, false>(env, l, prm, tk, s, 0, par, tvoff, xvoff);
        int q = 1;
        for (;;) {
            if (q >= tk.np) break;
            g13_iter<Env, P, 0, true, true>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_iter<Env, P, 1, true, true>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_iter<Env, P, 2, true, true>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
            if (q >= tk.np) break;
            g13_iter<Env, P, 3, true, true>(env, l, prm, tk, s, q, par, tvoff, xvoff); ++q;
        }
        switch ((tk.np - 1) & 3) {
            case 0: g13_tail<Env, P, 0>(env, l, prm, tk, s, par, tvoff, xvoff); break;
            case 1: g13_tail<Env, P, 1>(env, l, prm, tk, s, par, tvoff, xvoff); break;
            case 2: g13_tail<Env, P, 2>(env, l, prm, tk, s, par, tvoff, xvoff); break;
            default: g13_tail<Env, P, 3>(env, l, prm, tk, s, par, tvoff, xvoff); break;
        }
#endif

        // hand the next ticket to the whole workgroup, start its loads, then finish this task
        if (tid == 0) l.mailbox()[0] = next_id;
        env.barrier();
        const int nid = env.uniform(l.mailbox()[0]);      // wave-uniform by construction: keep the descriptors in SGPRs
        const bool more = nid < prm.r.ntasks;
        const Task13 cur = tk;
#if !defined(SS_G13_NOPREFETCH)
        if (more) {
            tk = g13_task(prm, nid);
            g13_open(prm, tk, s, tvoff, xvoff);
        }
#endif

        const RowCoef rc = make_rowcoef(prm.r, cur.rowi);
        const float scale = 1.0f / (float)B13;
#pragma unroll
        for (int j = 0; j < JMAX13; ++j) {
            if (j < cur.nj) {
                c32 v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = s.acc[j][r];
                g13_inv(env, l, v, par);
                par ^= 1;
                const int64_t t0 = (int64_t)(cur.j0 + j) * B13;
                const BlockCoef bc = make_blockcoef(rc, t0, prm.r.T);
                int64_t left = prm.r.T - t0;
                if (left > B13) left = B13;
                const BufRes yb = mk_buf(prm.r.y + (int64_t)cur.chan * prm.r.T + t0, (uint32_t)(left > 0 ? left * 4 : 0));
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const int n = n1 * 512 + tid;
                    const float val = -(v[n1].x * SS_S16(n1) + v[n1].y * SS_C16(n1)) * scale;   // -Im(z conj(twist)) / B
                    float coef = 0.0f;
                    const bool ok = block_coef(bc, n, coef);
                    buf_atomic_add_f32(yb, ok ? (uint32_t)n * 4u : 0x7ffffff0u, coef * val);
                }
            }
        }
        if (!more) break;
#if defined(SS_G13_NOPREFETCH)
        tk = g13_task(prm, nid);
        g13_open(prm, tk, s, tvoff, xvoff);
#endif
    }
}

}  // namespace ss
