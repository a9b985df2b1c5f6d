// valu_rate.hip -- micro-benchmark (GPU box): issue cost of plain vs packed f32 VALU on gfx950, 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float c32 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int KIND> __global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    c32 p0 = {a0, 1.f}, p1 = {1.f, 2.f}, p2 = {2.f, 1.f}, p3 = {3.f, 1.f}, p4 = {4.f, 1.f}, p5 = {5.f, 1.f}, p6 = {6.f, 1.f}, p7 = {7.f, 1.f};
    c32 m = {1.0001f, 0.9999f};
    float ms = 1.0001f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms));)
        } else if (KIND == 1) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms));)
        } else if (KIND == 4) {   // packed op with op_sel swizzle (complex rotate-add)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));)
        } else if (KIND == 5) {   // dependent chain of packed fma (latency)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
                         : "+v"(p0) : "v"(m));)
        } else if (KIND == 6) {   // v_mul_f64
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3; double dm = 1.0000001;
            REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));)
            a0 += (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 7) {   // v_pk_mul with SGPR-pair operand
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[0,1]\n"
                         "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 1024 * 2048); hipMalloc(&cyc, 8);
    const int iters = 2000, grid = 256;    // one block per CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<grid, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<grid, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 64.0;   // wave-instructions per wave
    const int waves_per_simd = threads / 256;
    printf("%-28s threads %4d (%d waves/SIMD): %.2f us, s_memtime-cycles/inst/wave %.2f, SIMD-cycles per inst @2.4GHz %.2f\n", name, threads,
           waves_per_simd ? waves_per_simd : 1, ms * 1e3, (double)hc / ninst, ms * 1e-3 * 2.4e9 / (ninst * (waves_per_simd ? waves_per_simd : 1)));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", th);
        run<1>("v_pk_fma_f32", th);
        run<2>("v_pk_add_f32", th);
        run<3>("v_add_f32", th);
        run<4>("v_pk_add_f32 op_sel/neg", th);
        run<5>("v_pk_fma_f32 dependent", th);
        run<6>("v_mul_f64", th);
        run<7>("v_pk_mul_f32 op_sel_hi", th);
    }
    return 0;
}
