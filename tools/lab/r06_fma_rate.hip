// VALU rates the loudness kernel depends on: v_fma_f64 / v_fma_f32 / v_pk_fma_f32, independent and dependent chains, at 1 / 2 / 4 / 8 waves per SIMD
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/r06_fma_rate.hip -o tools/lab/r06_fma_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, int ILP> __global__ void k(float* out, int iters, float seed) {
    // MODE 0: f64 fma, 1: f32 fma, 2: pk f32 fma ; ILP independent chains
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (MODE == 0) {
        double a[ILP]; const double b = 1.0000001 + seed, c = 1e-9;
        for (int i = 0; i < ILP; ++i) a[i] = t * 1e-6 + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        double s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
        if (s == 12345.678) out[t] = (float)s;
    } else if (MODE == 1) {
        float a[ILP]; const float b = 1.0000001f + seed, c = 1e-9f;
        for (int i = 0; i < ILP; ++i) a[i] = t * 1e-6f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        float s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
        if (s == 12345.678f) out[t] = s;
    } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 a[ILP]; const f2 b = {1.0000001f + seed, 1.0000002f}, c = {1e-9f, 2e-9f};
        for (int i = 0; i < ILP; ++i) a[i] = f2{t * 1e-6f + i, 1.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < ILP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        float s = 0; for (int i = 0; i < ILP; ++i) s += a[i].x + a[i].y;
        if (s == 12345.678f) out[t] = s;
    }
}
template <int MODE, int ILP> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int nt = 256, nb = 256 * wps;                    // 256-thread workgroups: one wave per SIMD each
        k<MODE, ILP><<<nb, nt>>>(out, 10, 0.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE, ILP><<<nb, nt>>>(out, iters, 0.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ninstr = (double)iters * 16 * ILP;          // per wave
        const double wave_instr_per_simd = ninstr * wps;
        printf("%-8s ilp %d waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz), per-wave %.2f ns per instruction\n", name, ILP, wps, ms,
               ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4, ms * 1e6 / ninstr);
    }
}
template <int OP, int ILP> __global__ void ki(unsigned* out, int iters, unsigned seed) {
    // integer / conversion ops of the bank generator: 0 v_mul_lo_u32, 1 v_mad_u32_u24, 2 v_xor_b32, 3 v_sad_u8, 4 v_cvt_f32_u32 + v_cvt_u32_f32
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned a[ILP]; const unsigned b = 0x85EBCA6Bu + seed, c = 0x1234567u;
    for (int i = 0; i < ILP; ++i) a[i] = t * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 3) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            }
    }
    unsigned s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
    if (s == 12345u) out[t] = s;
}
template <int OP> void runi(const char* name, unsigned* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {2, 4, 8}) {
        ki<OP, 4><<<256 * wps, 256>>>(out, 10, 0u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        ki<OP, 4><<<256 * wps, 256>>>(out, iters, 0u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s ilp 4 waves/SIMD %d: %.2f ns per instruction per SIMD\n", name, wps, ms * 1e6 / ((double)iters * 16 * 4 * wps));
    }
}
int main() {
    { unsigned* o; hipMalloc(&o, 256 * 2048 * 4); runi<0>("v_mul_lo_u32", o); runi<1>("v_mad_u32_u24", o); runi<2>("v_xor_b32", o); runi<3>("v_sad_u8", o); runi<4>("v_cvt_f32_u32", o); }
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    run<0, 1>("f64", out); run<0, 4>("f64", out);
    run<1, 1>("f32", out); run<1, 4>("f32", out);
    run<2, 1>("pk_f32", out); run<2, 4>("pk_f32", out);
    return 0;
}
