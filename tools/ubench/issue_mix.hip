// issue_mix.hip -- micro-benchmark (GPU box): do SALU / LDS / waitcnt instructions consume VALU issue bandwidth at 2 waves per SIMD?
// Each variant runs 64 packed VALU ops per iteration plus N extra instructions of another kind interleaved; if the extra kind is free
// the time equals the VALU-only time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c32 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define PK8 "v_pk_fma_f32 %0, %0, %[m], %0\n v_pk_fma_f32 %1, %1, %[m], %1\n v_pk_fma_f32 %2, %2, %[m], %2\n v_pk_fma_f32 %3, %3, %[m], %3\n" \
            "v_pk_fma_f32 %4, %4, %[m], %4\n v_pk_fma_f32 %5, %5, %[m], %5\n v_pk_fma_f32 %6, %6, %[m], %6\n v_pk_fma_f32 %7, %7, %[m], %7\n"
#define PK8S(x) "v_pk_fma_f32 %0, %0, %[m], %0\n" x "v_pk_fma_f32 %1, %1, %[m], %1\n" x "v_pk_fma_f32 %2, %2, %[m], %2\n" x "v_pk_fma_f32 %3, %3, %[m], %3\n" x \
                "v_pk_fma_f32 %4, %4, %[m], %4\n" x "v_pk_fma_f32 %5, %5, %[m], %5\n" x "v_pk_fma_f32 %6, %6, %[m], %6\n" x "v_pk_fma_f32 %7, %7, %[m], %7\n" x
#define OPS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : [m] "v"(m)

template <int KIND> __global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    c32 p0 = {1.f * threadIdx.x, 1.f}, p1 = {1.f, 2.f}, p2 = {2.f, 1.f}, p3 = {3.f, 1.f}, p4 = {4.f, 1.f}, p5 = {5.f, 1.f}, p6 = {6.f, 1.f}, p7 = {7.f, 1.f};
    c32 m = {1.0001f, 0.9999f};
    unsigned a = (threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 8;
    unsigned a16 = (threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
    c32 r0 = {0, 0};
    float g0 = 0, g1 = 0;
    f4v q0 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile(PK8 OPS);
            else if (KIND == 1) asm volatile(PK8S("s_add_u32 s90, s90, 1\n") OPS : "s90");
            else if (KIND == 2) asm volatile(PK8S("s_add_u32 s90, s90, 1\n s_add_u32 s91, s91, 1\n") OPS : "s90", "s91");
            else if (KIND == 3) asm volatile(PK8S("s_nop 0\n") OPS);
            else if (KIND == 4) asm volatile(PK8 "ds_read_b64 %[r], %[a]\n ds_read_b64 %[r], %[a] offset:512\n s_waitcnt lgkmcnt(0)\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), [r] "=&v"(r0) : [m] "v"(m), [a] "v"(a));
            else if (KIND == 5) asm volatile(PK8 "ds_read_b128 %[r], %[a]\n s_waitcnt lgkmcnt(0)\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), [r] "=&v"(q0) : [m] "v"(m), [a] "v"(a16));
            else if (KIND == 6) asm volatile(PK8 "ds_write_b64 %[a], %0\n ds_write_b64 %[a], %1 offset:512\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : [m] "v"(m), [a] "v"(a));
            else if (KIND == 7) asm volatile(PK8S("v_mov_b32 v200, v201\n") OPS : "v200");
            else if (KIND == 8) asm volatile(PK8S("s_waitcnt lgkmcnt(0)\n") OPS);
            else if (KIND == 9) asm volatile(PK8 "global_load_dword %[r], %[a], %[gp]\n global_load_dword %[r2], %[a], %[gp] offset:2048\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), [r] "=&v"(g0), [r2] "=&v"(g1) : [m] "v"(m), [a] "v"(a), [gp] "s"(out));
            else if (KIND == 10) asm volatile(PK8 "global_load_dwordx4 %[r], %[a], %[gp]\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), [r] "=&v"(q0) : [m] "v"(m), [a] "v"(a16), [gp] "s"(out));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    out[blockIdx.x * 512 + threadIdx.x] = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + r0.x + q0.x + g0 + g1 + smem[threadIdx.x];
}

template <int KIND> void run(const char* name) {
    float* out;
    hipMalloc(&out, 4 * 512 * 256);
    const int iters = 500;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-52s %9.2f us  (%.2f ns per 8-VALU group per SIMD)\n", name, ms * 1e3, ms * 1e6 / (iters * 8.0 * 2));
    hipFree(out);
}

int main() {
    run<0>("64 x v_pk_fma (2 waves/SIMD)");
    run<1>("+ 1 SALU per VALU");
    run<2>("+ 2 SALU per VALU");
    run<3>("+ 1 s_nop per VALU");
    run<4>("+ 2 ds_read_b64 + waitcnt per 8 VALU");
    run<5>("+ 1 ds_read_b128 + waitcnt per 8 VALU");
    run<6>("+ 2 ds_write_b64 per 8 VALU");
    run<7>("+ 1 v_mov_b32 per VALU");
    run<8>("+ 1 s_waitcnt per VALU");
    run<9>("+ 2 global_load_dword per 8 VALU (L2 hits)");
    run<10>("+ 1 global_load_dwordx4 per 8 VALU (L2 hits)");
    return 0;
}
