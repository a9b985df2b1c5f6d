// lds_rate.hip -- micro-benchmark (GPU box): cost of the LDS instructions the render kernel uses, at ITS occupancy
// (one 512-thread workgroup per CU = 2 waves per SIMD), all 8 waves issuing the same instruction mix.
// Prints LDS-unit cycles per wave-instruction (kernel cycles * / (8 waves * instructions per wave)).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND> __global__ __launch_bounds__(512, 2) void k(float* out, int iters, int stride_b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // addresses: wave-private 8 KiB regions
    unsigned base = wave * 8192;
    unsigned a_lin8 = base + lane * 8;                  // contiguous 8 B per lane
    unsigned a_row = base + lane * stride_b;            // row per lane (stride_b bytes)
    unsigned a_e3w = base + (lane >> 3) * 8 * stride_b + (lane & 7) * 8;      // E3 write pattern: row k2*8+k, column n4
    unsigned a_e2w = base + (lane & 7) * stride_b + (lane >> 3) * 8;          // E2 write pattern: row k*8+(lane&7), column lane>>3
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    f2v v0 = {1.f * tid, 2.f}, v1 = {3.f, 4.f};
    f4v q0 = {1.f, 2.f, 3.f, 4.f * tid}, q1 = q0;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 8 x ds_write_b64 contiguous (cross write, conflict free)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lin8), "v"(v0), "i"(k * 512));
        } else if (KIND == 1) { // 8 x ds_write_b64 E3 pattern (row stride = stride_b, k-th row: offset k*stride_b)
            asm volatile("ds_write_b64 %0, %1 offset:0\n ds_write_b64 %0, %1 offset:80\n ds_write_b64 %0, %1 offset:160\n ds_write_b64 %0, %1 offset:240\n"
                         "ds_write_b64 %0, %1 offset:320\n ds_write_b64 %0, %1 offset:400\n ds_write_b64 %0, %1 offset:480\n ds_write_b64 %0, %1 offset:560\n" ::"v"(a_e3w), "v"(v0));
        } else if (KIND == 2) { // 8 x ds_write_b64 E2 pattern (k-th: offset k*8*stride)
            asm volatile("ds_write_b64 %0, %1 offset:0\n ds_write_b64 %0, %1 offset:640\n ds_write_b64 %0, %1 offset:1280\n ds_write_b64 %0, %1 offset:1920\n"
                         "ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3200\n ds_write_b64 %0, %1 offset:3840\n ds_write_b64 %0, %1 offset:4480\n" ::"v"(a_e2w), "v"(v0));
        } else if (KIND == 3) { // 4 x ds_write_b128 row per lane
            asm volatile("ds_write_b128 %0, %1 offset:0\n ds_write_b128 %0, %1 offset:16\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:48\n" ::"v"(a_row), "v"(q0));
        } else if (KIND == 4) { // 8 x ds_read_b64 contiguous
            asm volatile("ds_read_b64 %0, %2 offset:0\n ds_read_b64 %1, %2 offset:512\n ds_read_b64 %0, %2 offset:1024\n ds_read_b64 %1, %2 offset:1536\n"
                         "ds_read_b64 %0, %2 offset:2048\n ds_read_b64 %1, %2 offset:2560\n ds_read_b64 %0, %2 offset:3072\n ds_read_b64 %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1) : "v"(a_lin8));
        } else if (KIND == 5) { // 4 x ds_read_b128 row per lane
            asm volatile("ds_read_b128 %0, %2 offset:0\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1) : "v"(a_row));
        } else if (KIND == 6) { // 8 x ds_read_b64 E3 pattern (inverse reads)
            asm volatile("ds_read_b64 %0, %2 offset:0\n ds_read_b64 %1, %2 offset:80\n ds_read_b64 %0, %2 offset:160\n ds_read_b64 %1, %2 offset:240\n"
                         "ds_read_b64 %0, %2 offset:320\n ds_read_b64 %1, %2 offset:400\n ds_read_b64 %0, %2 offset:480\n ds_read_b64 %1, %2 offset:560\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1) : "v"(a_e3w));
        } else if (KIND == 7) { // 8 x ds_read_b64 row per lane (what 4 x b128 replaced)
            asm volatile("ds_read_b64 %0, %2 offset:0\n ds_read_b64 %1, %2 offset:8\n ds_read_b64 %0, %2 offset:16\n ds_read_b64 %1, %2 offset:24\n"
                         "ds_read_b64 %0, %2 offset:32\n ds_read_b64 %1, %2 offset:40\n ds_read_b64 %0, %2 offset:48\n ds_read_b64 %1, %2 offset:56\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1) : "v"(a_row));
        }
        acc += v0.x + v1.y + q0.x + q1.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * 512 + tid] = acc;
}

template <int KIND> void run(const char* name, int ninst, int stride_b) {
    float* out;
    hipMalloc(&out, 4 * 512 * 256);
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 65536, 0, out, iters, stride_b);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 65536, 0, out, iters, stride_b);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.0e9;   // ~2.0 GHz under load
    printf("%-44s stride %3d: %.2f us, LDS-unit cycles per wave-instruction (@2.0 GHz) %.2f\n", name, stride_b, ms * 1e3, cyc / ((double)iters * ninst * 8));
    hipFree(out);
}

int main() {
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("ds_write_b64 contiguous (cross)", 8, 80);
    run<1>("ds_write_b64 E3 pattern", 8, 80);
    run<1>("ds_write_b64 E3 pattern", 8, 72);
    run<2>("ds_write_b64 E2 pattern", 8, 80);
    run<3>("ds_write_b128 row per lane", 4, 80);
    run<3>("ds_write_b128 row per lane", 4, 64);
    run<4>("ds_read_b64 contiguous (cross)", 8, 80);
    run<5>("ds_read_b128 row per lane", 4, 80);
    run<5>("ds_read_b128 row per lane", 4, 64);
    run<6>("ds_read_b64 E3 pattern", 8, 80);
    run<6>("ds_read_b64 E3 pattern", 8, 72);
    run<7>("ds_read_b64 row per lane x8", 8, 80);
    run<7>("ds_read_b64 row per lane x8", 8, 72);
    return 0;
}
