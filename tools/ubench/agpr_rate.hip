// agpr_rate.hip -- micro-benchmark (GPU box), round 3: can a ONE-wave-per-SIMD kernel use the accumulation half of the 512-entry register
// file for VALU state?  On gfx950 VALU instructions address arch VGPRs only (v0-v255; llvm-mc rejects `v_pk_fma_f32 v[0:1], a[0:1], ...`),
// an AGPR reaches the VALU through v_accvgpr_read_b32 (one 32-bit move per instruction).  This measures what such moves cost beside packed
// math at 1 and 2 waves per SIMD: the "block MAC with its spectrum window in AGPRs" pattern (2 v_pk_fma_f32 + 2 v_accvgpr_read_b32 per bin).
// Build: hipcc --offload-arch=gfx950 -O3 agpr_rate.hip -o agpr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
typedef float c32 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x

template <int KIND> __global__ void k(float* out, long long* cyc, int iters) {
    c32 p0 = {(float)threadIdx.x, 1.f}, p1 = {1.f, 2.f}, p2 = {2.f, 1.f}, p3 = {3.f, 1.f};
    c32 m = {1.0001f, 0.9999f};
    float r0 = 1.f, r1 = 2.f, r2 = 3.f, r3 = 4.f;
    float g0, g1, g2, g3;
    asm volatile("v_accvgpr_write_b32 %0, %4\n v_accvgpr_write_b32 %1, %4\n v_accvgpr_write_b32 %2, %4\n v_accvgpr_write_b32 %3, %4\n"
                 : "=a"(g0), "=a"(g1), "=a"(g2), "=a"(g3) : "v"(r0));
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // 8 packed fma (reference)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));)
        } else if (KIND == 1) {   // 8 accvgpr reads
            REP8(asm volatile("v_accvgpr_read_b32 %0, %4\n v_accvgpr_read_b32 %1, %5\n v_accvgpr_read_b32 %2, %6\n v_accvgpr_read_b32 %3, %7\n"
                              "v_accvgpr_read_b32 %0, %5\n v_accvgpr_read_b32 %1, %6\n v_accvgpr_read_b32 %2, %7\n v_accvgpr_read_b32 %3, %4\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "a"(g0), "a"(g1), "a"(g2), "a"(g3));)
        } else {                  // the MAC pattern: per bin 2 reads (re, im of the window value) + 2 packed fma -> 4 bins = 16 instructions
            REP8(asm volatile("v_accvgpr_read_b32 %4, %8\n v_accvgpr_read_b32 %5, %9\n v_pk_fma_f32 %0, %0, %12, %0\n v_pk_fma_f32 %1, %1, %12, %1\n"
                              "v_accvgpr_read_b32 %6, %10\n v_accvgpr_read_b32 %7, %11\n v_pk_fma_f32 %2, %2, %12, %2\n v_pk_fma_f32 %3, %3, %12, %3\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
                              : "a"(g0), "a"(g1), "a"(g2), "a"(g3), "v"(m));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p1.x + p2.y + p3.y + r0 + r1 + r2 + r3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, int threads, int ninst_per_rep) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 1024 * 2048); hipMalloc(&cyc, 8);
    const int iters = 2000, grid = 256;    // one block per CU: 256 threads = 1 wave per SIMD, 512 = 2
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<grid, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<grid, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 8.0 * ninst_per_rep;
    // wall clock: instructions per second per SIMD -> with the clock implied by (cycles counted / time) of workgroup 0
    const double secs = ms * 1e-3, clk = (double)hc / secs;            // workgroup 0 runs for (almost) the whole kernel
    const double tflops = KIND == 0 ? ninst * (threads / 64.0) * grid * 256.0 / secs * 1e-12 : 0.0;      // packed FMA: 2 x 2 x 64 flop per wave-instruction
    // the first wave's s_memtime count is the OLDER wave's view (it keeps its issue slots, the younger wave of the pair gets the rest):
    // the per-SIMD rate comes from the wall clock of the whole launch
    printf("%-44s %d wave(s)/SIMD: first wave %.2f s_memtime cycles per instruction | launch %.1f us = %.2f ns per instruction per SIMD%s\n", name,
           threads / 256, (double)hc / ninst, ms * 1e3, secs * 1e9 / (ninst * (threads / 256)),
           KIND == 0 ? (std::string(" = ") + std::to_string(tflops).substr(0, 6) + " TFLOP/s chip-wide").c_str() : "");
    (void)clk;
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512}) {
        run<0>("v_pk_fma_f32", th, 8);
        run<1>("v_accvgpr_read_b32", th, 8);
        run<2>("2 v_accvgpr_read_b32 + 2 v_pk_fma_f32 (MAC)", th, 8);
    }
    return 0;
}
