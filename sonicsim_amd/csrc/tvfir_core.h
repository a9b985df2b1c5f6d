// tvfir_core.h -- time-varying multichannel FIR (moving-source render) kernel bodies.
//
// One source, two compilations:
//   * hipcc --offload-arch=gfx950 : the kernel bodies run as HIP workgroups (256 threads = 4 wave64)
//   * g++ (tests/emul)            : the SAME bodies run under a std::thread/std::barrier
//                                   "workgroup emulator" so the index gymnastics can be verified on
//                                   a machine without a GPU.  The emulator is test infrastructure.
//
// Replaces the hot loops of SonicSim-SonicSet/SonicSim_moving.py:86-94 (oaconvolve with EVERY
// position + fancy-index gather + lerp) by a row-stationary, uniformly partitioned overlap-save:
//   y[c,t] = sum_r coef_r(t) * (x * h[r,c])[t],  coef_r(t) = (1-w[t]) [idx[t]==r] + w[t] [idx[t]+1==r]
// Each workgroup owns ONE filter row (r,c) (read from HBM exactly once, coalesced) and up to JMAX
// output blocks of B samples; partition spectra H_p are produced on the fly in LDS/registers and
// multiplied with the shared input spectra X_m (L2 resident); accumulators live in VGPRs.
//
// Transform: "right-angle" (odd-frequency) DFT.  A real 2B window a[] is folded to
//   z[n] = (a[n] - i a[n+B]) * exp(-i pi n / 2B),  n < B
// and transformed by ONE B-point complex FFT.  All B bins are ordinary complex numbers (no
// DC/Nyquist special case, no real-FFT split pass); bin products realise the negacyclic convolution
// whose samples [B,2B) equal the linear convolution with a B-tap partition (overlap-save valid part).
//
// FFT: B = 2048 = 8*8*8*4, decimation in frequency, 8 points per thread, 3 LDS exchanges.
// Forward leaves bins in a permuted "slot" order (thread tid, register r); the inverse is the
// transposed flow graph and consumes exactly that order, so no reordering pass exists anywhere.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SS_HD __host__ __device__ __forceinline__
#else
#define SS_HD inline
#endif

namespace ss {

// complex value.  Under hipcc it is a 2-wide float vector so that it lives in an aligned VGPR pair and
// every complex add/sub/multiply maps onto ONE or TWO packed-f32 instructions (v_pk_add/mul/fma_f32).
#if defined(__HIPCC__)
typedef float c32 __attribute__((ext_vector_type(2)));
#else
struct c32 {
    float x, y;
};
#endif
struct alignas(16) f4 {
    float x, y, z, w;
};

constexpr int B = 2048;   // output block = filter partition = complex FFT length
constexpr int NT = 256;   // threads per workgroup
constexpr int JMAX = 6;   // output blocks accumulated in registers per task
constexpr int DTILE = 1024;  // direct-form output tile
constexpr int DCHUNK = 256;  // direct-form tap chunk

// constant table layout (c32 units), built on the host in double precision (ss_build_consts)
constexpr int TW1_OFF = 0;       // [7][256]  W_2048^(t*k1),   k1 = 1..7
constexpr int TW2_OFF = 1792;    // [7][32]   W_256^(m2*k2),   k2 = 1..7
constexpr int TW3_OFF = 2016;    // [7][4]    W_32^(n4*k3),    k3 = 1..7
constexpr int TWIST_OFF = 2048;  // [2048]    exp(-i pi n / 4096)
constexpr int CONST_C32 = 4096;
constexpr int EX_C32 = 2304;     // one padded exchange buffer
constexpr int LDS_C32 = CONST_C32 + 2 * EX_C32;   // 8704 c32 = 69632 bytes

struct Task {
    int32_t row;    // filter row index r (position); 0 for the fixed-receiver path
    int32_t chan;   // channel c
    int32_t j0;     // first output block (OS: B grid, direct: DTILE grid)
    int32_t nj;     // number of blocks (1..JMAX; direct: 1)
};

// Cost of a task of the persistent single-launch engines in planner units (1 unit ~ 0.13 us on MI355X), for the LPT order and the
// equal-cost time ranges of the XCD-aware plan: np_eff forward transforms, np_eff x nj block MACs, nj inverse transforms + outputs.
// Round 4 tried the model FITTED to per-workgroup busy times of the assembly kernel (round 2, DESIGN.md section 6: 9 + 10 np + np nj / 2 +
// 20 nj) in its place: no difference with the dynamic queues + shared tail (profiles/r04h: kernel median 169.4-170.0 vs 167.9-169.5 us at
// config 2, 668-671 vs 666-670 us at config 5, three interleaved rounds) -- the queues absorb the model error.  The analytic form stays.
SS_HD int task_cost(int np_eff, int nj) { return np_eff * (10 + 2 * nj) + 12 * nj; }

enum CoefMode : int32_t { COEF_FIXED = 0, COEF_SEG = 1, COEF_EXPLICIT = 2 };

struct RenderParams {
    const float* x;          // [T]
    int64_t T;
    const float* bank;       // [P][C][L]
    int32_t P, C, L;
    int32_t NP;              // ceil(L / B)
    const c32* Xs;           // [M+1][B] input spectra in slot order; Xs[M] is all zero
    int32_t M;               // ceil(T / B)
    const c32* consts;       // [CONST_C32]
    const Task* tasks;
    int32_t ntasks;          // persistent kernels iterate over all of them
    int32_t mode;            // CoefMode
    int32_t accumulate;      // 0: y = contribution (even rows), 1: y += contribution (odd rows), 2: atomic add onto zeroed y
    const int64_t* seg_start;  // [P]  COEF_SEG: segment k covers [seg_start[k], seg_start[k+1]); seg_start[P-1] == T
    const int64_t* idx;      // [T]  COEF_EXPLICIT
    const float* w;          // [T]  COEF_EXPLICIT
    float* y;                // [C][T]
};

// ---------------------------------------------------------------------------------------------
// complex helpers.  On the device the swizzled forms are written as single packed-f32 instructions with
// explicit op_sel / neg modifiers: hipcc's SLP vectoriser otherwise spends ~27 % of the issued VALU on
// v_mov shuffles around packed ops (rocprofv3 / ISA histogram, profiles/r01a).
//   VOP3P f32 semantics: D.lo = op(S0[op_sel[0]], S1[op_sel[1]]), D.hi = op(S0[op_sel_hi[0]], S1[op_sel_hi[1]]),
//   neg_lo / neg_hi negate a source for the low / high result.
#if defined(__HIPCC__)
SS_HD c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
SS_HD c32 cadd(c32 a, c32 b) { return a + b; }
SS_HD c32 csub(c32 a, c32 b) { return a - b; }
SS_HD c32 cscale(c32 a, float s) { return a * s; }
#else
SS_HD c32 mk(float x, float y) { c32 r; r.x = x; r.y = y; return r; }
SS_HD c32 cadd(c32 a, c32 b) { return mk(a.x + b.x, a.y + b.y); }
SS_HD c32 csub(c32 a, c32 b) { return mk(a.x - b.x, a.y - b.y); }
SS_HD c32 cscale(c32 a, float s) { return mk(a.x * s, a.y * s); }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define SS_PK2(name, text)                                                        \
    __device__ __forceinline__ c32 name(c32 a, c32 b) {                            \
        c32 r;                                                                     \
        asm("v_pk_add_f32 %0, %1, %2 " text : "=v"(r) : "v"(a), "v"(b));         \
        return r;                                                                  \
    }
SS_PK2(cadd_mi, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                      // a + (-i) b = (a.x + b.y, a.y - b.x)
SS_PK2(csub_mi, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                      // a - (-i) b = (a.x - b.y, a.y + b.x)
SS_PK2(nadd_mi2, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]")        // -a + (-i) b = (-a.x + b.y, -a.y - b.x)
SS_PK2(nsub_mi2, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]")        // -a - (-i) b = (-a.x - b.y, -a.y + b.x)
#undef SS_PK2
__device__ __forceinline__ c32 cmul(c32 a, c32 w) {       // a * w
    c32 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c32 cmulc(c32 a, c32 w) {      // a * conj(w)
    c32 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c32 cmac(c32 acc, c32 x, c32 h) {   // acc + x * h
    c32 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(x), "v"(h), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x), "v"(h), "v"(t));
    return r;
}
#else
SS_HD c32 cadd_mi(c32 a, c32 b) { return mk(a.x + b.y, a.y - b.x); }
SS_HD c32 csub_mi(c32 a, c32 b) { return mk(a.x - b.y, a.y + b.x); }
SS_HD c32 nadd_mi2(c32 a, c32 b) { return mk(-a.x + b.y, -a.y - b.x); }
SS_HD c32 nsub_mi2(c32 a, c32 b) { return mk(-a.x - b.y, -a.y + b.x); }
SS_HD c32 cmul(c32 a, c32 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
SS_HD c32 cmulc(c32 a, c32 b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
SS_HD c32 cmac(c32 acc, c32 x, c32 h) { return cadd(acc, cmul(x, h)); }
#endif

// 8-point DFT, natural in -> natural out (register renaming is free): 28 packed instructions.
// Forward uses W8 = exp(-i pi/4); INV uses the conjugates.
template <bool INV> SS_HD void dft8(c32* v) {
    const float s = 0.70710678118654752440f;
    const c32 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const c32 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    const c32 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    const c32 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // odd-branch twiddles: b5 = a5 W8^(+-1), b7 = a7 W8^(+-3); the W8^(+-2) = -+i on a6 is folded below
    const c32 b5 = cscale(INV ? csub_mi(a5, a5) : cadd_mi(a5, a5), s);
    const c32 b7 = cscale(INV ? nsub_mi2(a7, a7) : nadd_mi2(a7, a7), s);
    {   // even outputs X0 X2 X4 X6 = DFT4(a0, a1, a2, a3)
        const c32 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), d = csub(a1, a3);
        v[0] = cadd(c0, c2);
        v[4] = csub(c0, c2);
        v[2] = INV ? csub_mi(c1, d) : cadd_mi(c1, d);
        v[6] = INV ? cadd_mi(c1, d) : csub_mi(c1, d);
    }
    {   // odd outputs X1 X3 X5 X7 = DFT4(a4, b5, -+i a6, b7)
        const c32 c0 = INV ? csub_mi(a4, a6) : cadd_mi(a4, a6);
        const c32 c1 = INV ? cadd_mi(a4, a6) : csub_mi(a4, a6);
        const c32 c2 = cadd(b5, b7), d = csub(b5, b7);
        v[1] = cadd(c0, c2);
        v[5] = csub(c0, c2);
        v[3] = INV ? csub_mi(c1, d) : cadd_mi(c1, d);
        v[7] = INV ? cadd_mi(c1, d) : csub_mi(c1, d);
    }
}

// 4-point DFT (forward: W4 = -i).  in/out natural order.
template <bool INV> SS_HD void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
    const c32 c0 = cadd(a0, a2), c1 = csub(a0, a2), c2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(c0, c2);
    a2 = csub(c0, c2);
    a1 = INV ? csub_mi(c1, d) : cadd_mi(c1, d);
    a3 = INV ? cadd_mi(c1, d) : csub_mi(c1, d);
}

// ---------------------------------------------------------------------------------------------
// LDS view of one workgroup
struct LdsView {
    c32* base;
    SS_HD const c32* tw1() const { return base + TW1_OFF; }
    SS_HD const c32* tw2() const { return base + TW2_OFF; }
    SS_HD const c32* tw3() const { return base + TW3_OFF; }
    SS_HD const c32* twist() const { return base + TWIST_OFF; }
    // the two exchange buffers swap roles every transform (see hazard note in fft_fwd)
    SS_HD c32* exA(int par) const { return base + CONST_C32 + (par ? EX_C32 : 0); }
    SS_HD c32* exB(int par) const { return base + CONST_C32 + (par ? 0 : EX_C32); }
};

// Forward B-point FFT.  In: v[n1] = z[n1*256 + tid].  Out: v[r] = Z[bin(tid,r)] ("slot order"):
//   G = tid + 256*(r>>2) = k1*64 + k2*8 + k3,  k4 = r&3,  bin = k1 + 8*k2 + 64*k3 + 512*k4.
// Buffer use is A,B,A with (A,B) swapped every call (par ^= 1): a transform's first write goes to
// the buffer whose last read is separated from it by the previous transform's final barrier, so
// three barriers per transform are sufficient.
template <class Env> SS_HD void fft_fwd(Env& env, const LdsView& l, c32* v, int& par) {
    const int tid = env.tid();
    c32* A = l.exA(par);
    c32* Bf = l.exB(par);
    par ^= 1;
    // pass 1: radix-8 over n1 (stride 256), twiddle W_2048^(tid*k1)
    dft8<false>(v);
    A[tid] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) A[k * 256 + tid] = cmul(v[k], l.tw1()[(k - 1) * 256 + tid]);
    env.barrier();
    // pass 2: thread (k1 = tid>>5, m2 = tid&31): radix-8 over n2 (stride 32), twiddle W_256^(m2*k2)
    {
        const int k1 = tid >> 5, m2 = tid & 31;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = A[k1 * 256 + n * 32 + m2];
        dft8<false>(v);
        Bf[(k1 * 8) * 36 + m2] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) Bf[(k1 * 8 + k) * 36 + m2] = cmul(v[k], l.tw2()[(k - 1) * 32 + m2]);
    }
    env.barrier();
    // pass 3: thread (g = tid>>2 = k1*8+k2, n4 = tid&3): radix-8 over n3 (stride 4), twiddle W_32^(n4*k3)
    {
        const int g = tid >> 2, n4 = tid & 3;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = Bf[g * 36 + n * 4 + n4];
        dft8<false>(v);
        A[(g * 9) * 4 + n4] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) A[(g * 9 + k) * 4 + n4] = cmul(v[k], l.tw3()[(k - 1) * 4 + n4]);
    }
    env.barrier();
    // pass 4: two radix-4 over n4 for groups G0 = tid, G1 = tid + 256  (G = g*8 + k3)
    {
        const int g0 = tid >> 3, k3 = tid & 7;
        const c32* p0 = A + (g0 * 9 + k3) * 4;
        const c32* p1 = A + ((g0 + 32) * 9 + k3) * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) { v[n] = p0[n]; v[4 + n] = p1[n]; }
        dft4<false>(v[0], v[1], v[2], v[3]);
        dft4<false>(v[4], v[5], v[6], v[7]);
    }
}

// Inverse (unnormalised, conjugate twiddles): transposed flow graph of fft_fwd.
// In: slot order.  Out: v[n1] = z[n1*256 + tid] * B.
template <class Env> SS_HD void fft_inv(Env& env, const LdsView& l, c32* v, int& par) {
    const int tid = env.tid();
    c32* A = l.exA(par);
    c32* Bf = l.exB(par);
    par ^= 1;
    {
        const int g0 = tid >> 3, k3 = tid & 7;
        dft4<true>(v[0], v[1], v[2], v[3]);
        dft4<true>(v[4], v[5], v[6], v[7]);
        c32* p0 = A + (g0 * 9 + k3) * 4;
        c32* p1 = A + ((g0 + 32) * 9 + k3) * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) { p0[n] = v[n]; p1[n] = v[4 + n]; }
    }
    env.barrier();
    {
        const int g = tid >> 2, n4 = tid & 3;
        v[0] = A[(g * 9) * 4 + n4];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(A[(g * 9 + k) * 4 + n4], l.tw3()[(k - 1) * 4 + n4]);
        dft8<true>(v);
#pragma unroll
        for (int n = 0; n < 8; ++n) Bf[g * 36 + n * 4 + n4] = v[n];
    }
    env.barrier();
    {
        const int k1 = tid >> 5, m2 = tid & 31;
        v[0] = Bf[(k1 * 8) * 36 + m2];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(Bf[(k1 * 8 + k) * 36 + m2], l.tw2()[(k - 1) * 32 + m2]);
        dft8<true>(v);
#pragma unroll
        for (int n = 0; n < 8; ++n) A[k1 * 256 + n * 32 + m2] = v[n];
    }
    env.barrier();
    {
        v[0] = A[tid];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(A[k * 256 + tid], l.tw1()[(k - 1) * 256 + tid]);
        dft8<true>(v);
    }
}

template <class Env> SS_HD void load_consts(Env& env, const LdsView& l, const c32* consts) {
    const int tid = env.tid();
    for (int i = tid; i < CONST_C32; i += NT) l.base[i] = consts[i];
    env.barrier();
}

// ---------------------------------------------------------------------------------------------
// Per-sample interpolation coefficient of filter row r at output sample t (rows I/V of SURVEY 8a).
// Returns false when row r is not one of the two rows responsible for t.
//   COEF_SEG reproduces np.linspace(0,1,n,endpoint=False).astype(float32) bit-exactly:
//   w = float(double(i) * (1.0/double(n)))  (SonicSim_moving.py:43-45), start coef = 1.0f - w (:94).
struct RowCoef {
    int32_t mode, row;
    int64_t a0, a1, a2;      // COEF_SEG: [a0,a1) row is END filter of segment row-1; [a1,a2) START of segment row
    double inv0, inv1;
    const int64_t* idx;
    const float* w;
};

SS_HD RowCoef make_rowcoef(const RenderParams& prm, int row) {
    RowCoef rc;
    rc.mode = prm.mode;
    rc.row = row;
    rc.idx = prm.idx;
    rc.w = prm.w;
    rc.a0 = rc.a1 = rc.a2 = 0;
    rc.inv0 = rc.inv1 = 0.0;
    if (prm.mode == COEF_SEG) {
        rc.a1 = prm.seg_start[row];
        rc.a0 = row > 0 ? prm.seg_start[row - 1] : rc.a1;
        rc.a2 = row < prm.P - 1 ? prm.seg_start[row + 1] : rc.a1;
        if (rc.a1 > rc.a0) rc.inv0 = 1.0 / (double)(rc.a1 - rc.a0);
        if (rc.a2 > rc.a1) rc.inv1 = 1.0 / (double)(rc.a2 - rc.a1);
    }
    return rc;
}

SS_HD bool row_coef(const RowCoef& rc, int64_t t, float& coef) {
    if (rc.mode == COEF_FIXED) { coef = 1.0f; return true; }
    if (rc.mode == COEF_SEG) {
        if (t >= rc.a1) {
            if (t >= rc.a2) return false;
            const float wv = (float)((double)(t - rc.a1) * rc.inv1);
            coef = 1.0f - wv;
            return true;
        }
        if (t < rc.a0) return false;
        coef = (float)((double)(t - rc.a0) * rc.inv0);
        return true;
    }
    const int64_t k = rc.idx[t];
    if (k == rc.row) { coef = 1.0f - rc.w[t]; return true; }
    if (k + 1 == rc.row) { coef = rc.w[t]; return true; }
    return false;
}

// Block-relative form used by geometry 12: everything per sample is 32-bit (sample offset r inside the
// block, segment bounds relative to the block start; requires T < 2^30), one int->double conversion, one
// double multiply, one double->float conversion per sample -- the int64 compares and int64->double
// conversions of row_coef() cost ~10 % of the whole render when done per sample.
struct BlockCoef {
    int32_t mode, row;
    int32_t A0, A1, A2, TL;   // segment bounds and T, relative to the block start, clamped to +-2^30
    double inv0, inv1;
    const int64_t* idx;       // already offset to the block start
    const float* w;
};
SS_HD int32_t rel30(int64_t v) {
    const int64_t lim = (int64_t)1 << 30;
    return (int32_t)(v < -lim ? -lim : (v > lim ? lim : v));
}
SS_HD BlockCoef make_blockcoef(const RowCoef& rc, int64_t t0, int64_t T) {
    BlockCoef bc;
    bc.mode = rc.mode; bc.row = rc.row;
    bc.A0 = rel30(rc.a0 - t0); bc.A1 = rel30(rc.a1 - t0); bc.A2 = rel30(rc.a2 - t0); bc.TL = rel30(T - t0);
    bc.inv0 = rc.inv0; bc.inv1 = rc.inv1;
    bc.idx = rc.idx ? rc.idx + t0 : rc.idx;
    bc.w = rc.w ? rc.w + t0 : rc.w;
    return bc;
}
SS_HD bool block_coef(const BlockCoef& bc, int32_t r, float& coef) {
    if (r >= bc.TL) return false;
    if (bc.mode == COEF_FIXED) { coef = 1.0f; return true; }
    if (bc.mode == COEF_SEG) {
        if (r >= bc.A1) {
            if (r >= bc.A2) return false;
            coef = 1.0f - (float)((double)(r - bc.A1) * bc.inv1);
            return true;
        }
        if (r < bc.A0) return false;
        coef = (float)((double)(r - bc.A0) * bc.inv0);
        return true;
    }
    const int64_t k = bc.idx[r];
    if (k == bc.row) { coef = 1.0f - bc.w[r]; return true; }
    if (k + 1 == bc.row) { coef = bc.w[r]; return true; }
    return false;
}

// Accumulation modes.  Mode 2 (geometry 12): y is zeroed by the spectra kernel and every sample receives
// exactly TWO hardware float atomic adds (its start row and its end row) -- 0 + a + b is the same bit
// pattern in either order, so the result stays deterministic while all rows run in ONE launch.
SS_HD void atomic_add_f32(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsafeAtomicAdd(p, v);
#else
    *p += v;
#endif
}
SS_HD void emit(const RenderParams& prm, const RowCoef& rc, int chan, int64_t t, float val) {
    float coef;
    if (t < prm.T && row_coef(rc, t, coef)) {
        float* yp = prm.y + (int64_t)chan * prm.T + t;
        const float contrib = coef * val;
        if (prm.accumulate == 2) atomic_add_f32(yp, contrib);
        else if (prm.accumulate) *yp = *yp + contrib;
        else *yp = contrib;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel body 1: input spectra.  One workgroup per window m (0 <= m < M):
//   window m = x[(m-1)B, (m+1)B), zero outside [0,T);  Xs[m][slot] in the layout
//   c32 index ((r>>1)*256 + tid)*2 + (r&1)  (so the MAC loop reads 16 B per lane, lane-contiguous).
//   Workgroup m == M writes an all-zero spectrum: out-of-range (block, partition) pairs of the render
//   kernel read it instead of branching, which keeps the MAC loop free of control flow.
template <class Env> SS_HD void xspec_body(Env& env, const float* x, int64_t T, const c32* consts, c32* Xs, int m, int M) {
    const int tid = env.tid();
    if (m >= M) {   // whole workgroup takes this branch
        c32* z = Xs + (int64_t)M * B;
        for (int i = tid; i < B; i += NT) z[i] = mk(0.0f, 0.0f);
        return;
    }
    LdsView l; l.base = env.lds();
    load_consts(env, l, consts);
    c32 v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = n1 * 256 + tid;
        const int64_t tlo = (int64_t)(m - 1) * B + n, thi = (int64_t)m * B + n;
        const float lo = (tlo >= 0 && tlo < T) ? x[tlo] : 0.0f;
        const float hi = (thi < T) ? x[thi] : 0.0f;
        const c32 tw = l.twist()[n];
        v[n1] = mk(lo * tw.x + hi * tw.y, lo * tw.y - hi * tw.x);   // (lo - i hi) * tw
    }
    int par = 0;
    fft_fwd(env, l, v, par);
    c32* out = Xs + (int64_t)m * B;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[(q * 256 + tid) * 2 + 0] = v[2 * q];
        out[(q * 256 + tid) * 2 + 1] = v[2 * q + 1];
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel body 2: row-stationary partitioned overlap-save.  One workgroup per Task.
//
// Per partition p:  [issue X loads for the first XD blocks] -> twist -> forward FFT (3 barriers; the X
// loads and the next partition's taps are in flight underneath) -> 6 block MACs, each consuming a
// prefetched spectrum and re-issuing the load XD blocks ahead.  (block, partition) pairs that fall
// before t = 0 or beyond the task's nj blocks read the all-zero spectrum Xs[M]: no control flow in the loop.
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_KEEP(x) asm volatile("" ::"v"(x))
#else
#define SS_KEEP(x) (void)(x)
#endif
// ABL: ablation mask for profiling only (0 in production): 1 = skip forward FFT, 2 = skip MAC, 4 = skip inverse FFT
template <class Env, int XD, int ABL = 0> SS_HD void os_body(Env& env, const RenderParams& prm, int task_id) {
    LdsView l; l.base = env.lds();
    load_consts(env, l, prm.consts);
    const int tid = env.tid();
    const Task tk = prm.tasks[task_id];
    if (tk.nj <= 0) return;   // padding task (whole workgroup)
    const float* h = prm.bank + ((int64_t)tk.row * prm.C + tk.chan) * prm.L;
    const int nj = tk.nj, j0 = tk.j0;

    c32 acc[JMAX][8];
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[j][r] = mk(0.0f, 0.0f);

    // partitions p > j0+nj-1 only meet windows before t=0 (all zero): skip them
    int np_eff = prm.NP;
    if (np_eff > j0 + nj) np_eff = j0 + nj;

    const f4* Xq = reinterpret_cast<const f4*>(prm.Xs) + tid;     // + m*(B/2) + q*256
    auto xaddr = [&](int j, int p) -> const f4* {
        const int m = j0 + j - p;
        const int me = (j < nj && m >= 0) ? m : prm.M;            // Xs[M] is the zero spectrum
        return Xq + (int64_t)me * (B / 2);
    };

    float hn[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = n1 * 256 + tid;
        hn[n1] = (n < prm.L) ? h[n] : 0.0f;
    }
    int par = 0;
    constexpr int XB = XD > 0 ? XD : 1;
    f4 xb[XB][4];
    for (int p = 0; p < np_eff; ++p) {
        if (XD > 0) {
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const f4* a = xaddr(j, p);
#pragma unroll
                for (int q = 0; q < 4; ++q) xb[j][q] = a[q * 256];
            }
        }
        c32 v[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const c32 tw = l.twist()[n1 * 256 + tid];
            v[n1] = mk(hn[n1] * tw.x, hn[n1] * tw.y);
        }
        // software prefetch of the next partition (HBM latency hides under this partition's FFT+MAC)
        if (p + 1 < np_eff) {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int n = (p + 1) * B + n1 * 256 + tid;
                hn[n1] = (n < prm.L) ? h[n] : 0.0f;
            }
        }
        if (!(ABL & 1)) fft_fwd(env, l, v, par);
        if (ABL & 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { SS_KEEP(v[r].x); SS_KEEP(v[r].y); }
        }
#pragma unroll
        for (int j = 0; j < ((ABL & 2) ? 0 : JMAX); ++j) {
            f4 xv[4];
            if (XD > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[q] = xb[j % XB][q];
                if (j + XB < JMAX) {
                    const f4* a = xaddr(j + XB, p);
#pragma unroll
                    for (int q = 0; q < 4; ++q) xb[j % XB][q] = a[q * 256];
                }
            } else {
                const f4* a = xaddr(j, p);
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[q] = a[q * 256];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[j][2 * q] = cmac(acc[j][2 * q], mk(xv[q].x, xv[q].y), v[2 * q]);
                acc[j][2 * q + 1] = cmac(acc[j][2 * q + 1], mk(xv[q].z, xv[q].w), v[2 * q + 1]);
            }
        }
    }

    const RowCoef rc = make_rowcoef(prm, tk.row);
    const float scale = 1.0f / (float)B;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        if (j < nj) {
            c32 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = acc[j][r];
            if (!(ABL & 4)) fft_inv(env, l, v, par);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int n = n1 * 256 + tid;
                const c32 tw = l.twist()[n];
                const float val = (v[n1].x * tw.y - v[n1].y * tw.x) * scale;   // -Im(z * conj(tw)) / B
                emit(prm, rc, tk.chan, (int64_t)(j0 + j) * B + n, val);
            }
        }
    }
}

// =============================================================================================
// Geometry 12: B = 4096 = 8^4, 512 threads (8 wave64), JMAX = 4, sliding window of input spectra in VGPRs.
//
// Why: with B = 2048 the MAC loop re-reads 6 spectra (96 KB per workgroup) per partition through the
// 64 B/clk L1 path and is bound there.  With B = 4096 a row touches <= 4 blocks most of the time, the
// accumulators shrink to 64 VGPRs and the 4 spectra a partition needs are exactly the previous
// partition's shifted by one block: keep them in registers and load ONE new spectrum per partition
// (4.4x less L1/L2 traffic, half the barriers per output sample).
// =============================================================================================
constexpr int B12 = 4096;
constexpr int NT12 = 512;
constexpr int JMAX12 = 4;
constexpr int TW1_12 = 0;        // [7][512] W_4096^(t*k1)
constexpr int TW2_12 = 3584;     // [7][64]  W_512^(m2*k2)
constexpr int TW3_12 = 4032;     // [7][8]   W_64^(n4*k3)
constexpr int TWIST_12 = 4096;   // [4096]   exp(-i pi n / 8192)
constexpr int CONST12_C32 = 8192;
constexpr int C12_C32 = 4096;        // cross-wave exchange buffer (pass 1 -> pass 2)
constexpr int PRIV12_C32 = 576;      // per-wave private exchange region: E2 8 rows x 72, E3 64 groups x 9
constexpr int LDS12_C32 = CONST12_C32 + C12_C32 + 8 * PRIV12_C32;   // 16896 c32 = 135168 bytes

struct Lds12 {
    c32* base;
    SS_HD const c32* tw1() const { return base + TW1_12; }
    SS_HD const c32* tw2() const { return base + TW2_12; }
    SS_HD const c32* tw3() const { return base + TW3_12; }
    SS_HD const c32* twist() const { return base + TWIST_12; }
    SS_HD c32* cross() const { return base + CONST12_C32; }
    SS_HD c32* priv(int wave) const { return base + CONST12_C32 + C12_C32 + wave * PRIV12_C32; }
};

// Forward 4096-point FFT.  In: v[n1] = z[n1*512 + tid].  Out: slot (tid, r): tid = k1*64 + k2*8 + k3, r = k4,
// bin = k1 + 8*k2 + 64*k3 + 512*k4.
//
// Only pass 1 -> pass 2 crosses waves (wave k1 then owns sub-transform k1: 512 points = 64 lanes x 8).  The
// later exchanges stay inside the wave's private LDS region and need NO workgroup barrier -- LDS executes
// one wave's instructions in order -- so the 8 waves drift apart and the two waves sharing a SIMD overlap
// each other's LDS latency.  Barriers per transform: one RAW (cross buffer written -> read) and one WAR
// right after the reads (cheap: the waves have just been released together), instead of three.
template <class Env> SS_HD void fft12_fwd(Env& env, const Lds12& l, c32* v, int& par) {
    (void)par;
    const int tid = env.tid();
    c32* C = l.cross();
    c32* P = l.priv(tid >> 6);
    const int lane = tid & 63;
    dft8<false>(v);
    C[tid] = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) C[k * 512 + tid] = cmul(v[k], l.tw1()[(k - 1) * 512 + tid]);
    env.barrier();
    {
        const int k1 = tid >> 6;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = C[k1 * 512 + n * 64 + lane];
        env.barrier();                                   // WAR: the next transform may overwrite C
        dft8<false>(v);
        P[lane] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) P[k * 72 + lane] = cmul(v[k], l.tw2()[(k - 1) * 64 + lane]);
    }
    env.wave_sync();
    {
        const int k2 = lane >> 3, n4 = lane & 7;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = P[k2 * 72 + n * 8 + n4];
        dft8<false>(v);
        env.wave_sync();                                 // all lanes have read before the region is rewritten
        P[(k2 * 8) * 9 + n4] = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) P[(k2 * 8 + k) * 9 + n4] = cmul(v[k], l.tw3()[(k - 1) * 8 + n4]);
    }
    env.wave_sync();
    {
        const c32* p0 = P + lane * 9;
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = p0[n];
        dft8<false>(v);
    }
    env.wave_sync();                                     // region free for the next transform of this wave
}

template <class Env> SS_HD void fft12_inv(Env& env, const Lds12& l, c32* v, int& par) {
    (void)par;
    const int tid = env.tid();
    c32* C = l.cross();
    c32* P = l.priv(tid >> 6);
    const int lane = tid & 63;
    {
        dft8<true>(v);
        c32* p0 = P + lane * 9;
#pragma unroll
        for (int n = 0; n < 8; ++n) p0[n] = v[n];
    }
    env.wave_sync();
    {
        const int k2 = lane >> 3, n4 = lane & 7;
        v[0] = P[(k2 * 8) * 9 + n4];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(P[(k2 * 8 + k) * 9 + n4], l.tw3()[(k - 1) * 8 + n4]);
        dft8<true>(v);
        env.wave_sync();
#pragma unroll
        for (int n = 0; n < 8; ++n) P[k2 * 72 + n * 8 + n4] = v[n];
    }
    env.wave_sync();
    {
        const int k1 = tid >> 6;
        v[0] = P[lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(P[k * 72 + lane], l.tw2()[(k - 1) * 64 + lane]);
        dft8<true>(v);
#pragma unroll
        for (int n = 0; n < 8; ++n) C[k1 * 512 + n * 64 + lane] = v[n];
    }
    env.barrier();
    {
        v[0] = C[tid];
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmulc(C[k * 512 + tid], l.tw1()[(k - 1) * 512 + tid]);
        env.barrier();                                   // WAR
        dft8<true>(v);
    }
}

template <class Env> SS_HD void load_consts12(Env& env, const Lds12& l, const c32* consts) {
    const int tid = env.tid();
    for (int i = tid; i < CONST12_C32; i += NT12) l.base[i] = consts[i];
    env.barrier();
}

// input spectra, B = 4096: slot layout c32 index ((r>>1)*512 + tid)*2 + (r&1); Xs[M] = 0
template <class Env> SS_HD void xspec12_body(Env& env, const float* x, int64_t T, const c32* consts, c32* Xs, int m, int M,
                                             float* yzero = nullptr, int64_t nzero = 0) {
    const int tid = env.tid();
    if (yzero) {   // zero this workgroup's slice of y for the atomic accumulation of the render kernel (16-byte stores)
        const bool al = (reinterpret_cast<uintptr_t>(yzero) & 15) == 0;
        const int64_t n4 = al ? nzero / 4 : 0;
        const int64_t chunk = (n4 + M) / (M + 1);
        const int64_t lo = (int64_t)m * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
        f4* y4 = reinterpret_cast<f4*>(yzero);
        const f4 z{0.0f, 0.0f, 0.0f, 0.0f};
        for (int64_t i = lo + tid; i < hi; i += NT12) y4[i] = z;
        if (m == 0)
            for (int64_t i = n4 * 4 + tid; i < nzero; i += NT12) yzero[i] = 0.0f;
    }
    if (m >= M) {
        c32* z = Xs + (int64_t)M * B12;
        for (int i = tid; i < B12; i += NT12) z[i] = mk(0.0f, 0.0f);
        return;
    }
    Lds12 l; l.base = env.lds();
    load_consts12(env, l, consts);
    c32 v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = n1 * 512 + tid;
        const int64_t tlo = (int64_t)(m - 1) * B12 + n, thi = (int64_t)m * B12 + n;
        const float lo = (tlo >= 0 && tlo < T) ? x[tlo] : 0.0f;
        const float hi = (thi < T) ? x[thi] : 0.0f;
        const c32 tw = l.twist()[n];
        v[n1] = mk(lo * tw.x + hi * tw.y, lo * tw.y - hi * tw.x);
    }
    int par = 0;
    fft12_fwd(env, l, v, par);
    c32* out = Xs + (int64_t)m * B12;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[(q * 512 + tid) * 2 + 0] = v[2 * q];
        out[(q * 512 + tid) * 2 + 1] = v[2 * q + 1];
    }
}

// render body, B = 4096, sliding spectrum window in registers.
// One partition step: twist -> forward FFT -> 4 block MACs -> slide the window.
template <class Env, int ABL> SS_HD void os12_step(Env& env, const Lds12& l, const float* h, int lastn, int p, const f4* nxaddr,
                                                    float (&hcur)[8], c32 (&acc)[JMAX12][8], f4 (&win)[JMAX12][4], int& par) {
    const int tid = env.tid();
    // the one new spectrum the NEXT partition needs (block 0 at step p+1): in flight under this FFT + MAC
    f4 nx[4];
    if (ABL & 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) nx[q] = win[1][q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) nx[q] = nxaddr[q * 512];
    }
    c32 v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = p * B12 + n1 * 512 + tid;
        const float hv = (n <= lastn) ? hcur[n1] : 0.0f;     // the tail mask is applied at use, never right after the load
        const c32 tw = l.twist()[n1 * 512 + tid];
        v[n1] = mk(hv * tw.x, hv * tw.y);
    }
    // taps for partition p+2 go into the buffer just consumed (ping-pong with the other buffer): the HBM
    // latency is covered by two full steps, and no register copy ever waits on an in-flight load
    if (!(ABL & 16)) {
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int n = (p + 2) * B12 + n1 * 512 + tid;
            hcur[n1] = h[n < lastn ? n : lastn];              // clamped address: branch-free, always in bounds
        }
    }
    if (!(ABL & 1)) fft12_fwd(env, l, v, par);
    if (ABL & 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { SS_KEEP(v[r].x); SS_KEEP(v[r].y); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { SS_KEEP(nx[q].x); SS_KEEP(nx[q].w); }
    }
#pragma unroll
    for (int j = 0; j < ((ABL & 2) ? 0 : JMAX12); ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[j][2 * q] = cmac(acc[j][2 * q], mk(win[j][q].x, win[j][q].y), v[2 * q]);
            acc[j][2 * q + 1] = cmac(acc[j][2 * q + 1], mk(win[j][q].z, win[j][q].w), v[2 * q + 1]);
        }
    }
    // slide: block j of the next partition needs what block j-1 used now
#pragma unroll
    for (int j = JMAX12 - 1; j > 0; --j)
#pragma unroll
        for (int q = 0; q < 4; ++q) win[j][q] = win[j - 1][q];
#pragma unroll
    for (int q = 0; q < 4; ++q) win[0][q] = nx[q];
}

// Persistent: workgroup `wg` of `nwg` walks tasks wg, wg+nwg, ... (the host sorts tasks by descending cost,
// so this static round-robin is an LPT schedule); the 64 KB constant table is loaded into LDS once.
template <class Env, int ABL = 0> SS_HD void os12_body(Env& env, const RenderParams& prm, int wg, int nwg) {
    Lds12 l; l.base = env.lds();
    load_consts12(env, l, prm.consts);
    const int tid = env.tid();
    int par = 0;
    const int lastn = prm.L - 1;
    const f4* Xq = reinterpret_cast<const f4*>(prm.Xs) + tid;     // + m*(B12/2) + q*512
    auto xaddr = [&](int m) -> const f4* {
        const int me = (m >= 0 && m < prm.M) ? m : prm.M;         // Xs[M] is the zero spectrum
        return Xq + (int64_t)me * (B12 / 2);
    };
    for (int task_id = wg; task_id < prm.ntasks; task_id += nwg) {
        const Task tk = prm.tasks[task_id];
        if (tk.nj <= 0) continue;
        const float* h = prm.bank + ((int64_t)tk.row * prm.C + tk.chan) * prm.L;
        const int nj = tk.nj, j0 = tk.j0;

        c32 acc[JMAX12][8];
#pragma unroll
        for (int j = 0; j < JMAX12; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[j][r] = mk(0.0f, 0.0f);

        // partitions p > j0+nj-1 only meet windows before t=0 (all zero): skip them
        int np_eff = prm.NP;
        if (np_eff > j0 + nj) np_eff = j0 + nj;

        // window: win[j] = X_{j0 + j - p} for the current partition p
        f4 win[JMAX12][4];
#pragma unroll
        for (int j = 0; j < JMAX12; ++j) {
            const f4* a = xaddr(j < nj ? j0 + j : -1);
#pragma unroll
            for (int q = 0; q < 4; ++q) win[j][q] = a[q * 512];
        }
        float hA[8], hB[8];
        if (ABL & 16) {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) hA[n1] = hB[n1] = 0.25f * (float)(tid + n1);
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int na = n1 * 512 + tid, nb = B12 + n1 * 512 + tid;
                hA[n1] = h[na < lastn ? na : lastn];
                hB[n1] = h[nb < lastn ? nb : lastn];
            }
        }
        // two partitions per trip (ping-pong tap buffers); an odd count runs one extra all-zero-tap step
        for (int p = 0; p < np_eff; p += 2) {
            os12_step<Env, ABL>(env, l, h, lastn, p, xaddr(j0 - (p + 1)), hA, acc, win, par);
            if (p + 1 < np_eff) os12_step<Env, ABL>(env, l, h, lastn, p + 1, xaddr(j0 - (p + 2)), hB, acc, win, par);
        }

        const RowCoef rc = make_rowcoef(prm, tk.row);
        const float scale = 1.0f / (float)B12;
#pragma unroll
        for (int j = 0; j < JMAX12; ++j) {
            if (j < nj) {
                c32 v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = acc[j][r];
                if (!(ABL & 4)) fft12_inv(env, l, v, par);
                const int64_t t0 = (int64_t)(j0 + j) * B12;
                const BlockCoef bc = make_blockcoef(rc, t0, prm.T);
                float* yb = prm.y + (int64_t)tk.chan * prm.T + t0;
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const int n = n1 * 512 + tid;
                    const c32 tw = l.twist()[n];
                    const float val = (v[n1].x * tw.y - v[n1].y * tw.x) * scale;   // -Im(z * conj(tw)) / B
                    float coef;
                    if (ABL & 8) SS_KEEP(val);
                    else if (block_coef(bc, n, coef)) atomic_add_f32(yb + n, coef * val);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel body 3: direct-form fallback / cross-check (exact fp32 fmaf chain, no transform).
// One workgroup per Task{row, chan, tile}; thread owns 4 consecutive outputs of a DTILE tile.
// LDS: hs[DCHUNK] taps, xs[DTILE + DCHUNK] input window (float).
template <class Env> SS_HD void direct_body(Env& env, const RenderParams& prm, int task_id) {
    float* lf = (float*)env.lds();
    float* hs = lf;                 // [DCHUNK]
    float* xs = lf + DCHUNK;        // [DTILE + DCHUNK]  xs[i] = x[t0 - tau0 - (DCHUNK-1) + i]
    const int tid = env.tid();
    const Task tk = prm.tasks[task_id];
    const float* h = prm.bank + ((int64_t)tk.row * prm.C + tk.chan) * prm.L;
    const int64_t t0 = (int64_t)tk.j0 * DTILE;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    // taps beyond t0+DTILE-1 only meet x before 0
    int64_t lmax = prm.L;
    if (lmax > t0 + DTILE) lmax = t0 + DTILE;
    for (int64_t tau0 = 0; tau0 < lmax; tau0 += DCHUNK) {
        env.barrier();
        {
            const int64_t ti = tau0 + tid;
            hs[tid] = (ti < prm.L) ? h[ti] : 0.0f;
        }
        for (int i = tid; i < DTILE + DCHUNK; i += NT) {
            const int64_t ti = t0 - tau0 - (DCHUNK - 1) + i;
            xs[i] = (ti >= 0 && ti < prm.T) ? prm.x[ti] : 0.0f;
        }
        env.barrier();
        // output t = t0 + 4*tid + q uses xs[4*tid + q + (DCHUNK-1) - tau']
        const float* xb = xs + 4 * tid + (DCHUNK - 1);
#pragma unroll 8
        for (int u = 0; u < DCHUNK; ++u) {
            const float hv = hs[u];
            o0 += hv * xb[0 - u];
            o1 += hv * xb[1 - u];
            o2 += hv * xb[2 - u];
            o3 += hv * xb[3 - u];
        }
    }
    const RowCoef rc = make_rowcoef(prm, tk.row);
    const int64_t t = t0 + 4 * tid;
    emit(prm, rc, tk.chan, t + 0, o0);
    emit(prm, rc, tk.chan, t + 1, o1);
    emit(prm, rc, tk.chan, t + 2, o2);
    emit(prm, rc, tk.chan, t + 3, o3);
}

}  // namespace ss
