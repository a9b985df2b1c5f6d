// Cold instruction-fetch cost of straight-line code: kernels that execute N KiB of code exactly once.
// hipcc --offload-arch=gfx950 -O3 icache.hip -o icache && ./icache
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256) void k_line(float* out, float a, float b) {
    float x = threadIdx.x, y = blockIdx.x;
#pragma unroll
    for (int i = 0; i < N * 64; ++i) {      // 2 x 8-byte VOP3 per iteration -> 1 KiB per 64 iterations
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(b));
    }
    if (x + y == 12345.f) out[0] = x;
}
template <int N>
__global__ __launch_bounds__(256) void k_loop(float* out, float a, float b, int reps) {
    float x = threadIdx.x, y = blockIdx.x;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(b));
        }
    }
    if (x + y == 12345.f) out[0] = x;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 50 * 1e3f;
}
int main() {
    float* out; hipMalloc(&out, 4);
    const int G = 24;
    printf("grid %d x 256 threads; us per launch (back-to-back launches)\n", G);
    printf("straight 1 KiB  %.2f\n", timeit([&] { k_line<1><<<G, 256>>>(out, 1.f, 0.f); }));
    printf("straight 4 KiB  %.2f\n", timeit([&] { k_line<4><<<G, 256>>>(out, 1.f, 0.f); }));
    printf("straight 16 KiB %.2f\n", timeit([&] { k_line<16><<<G, 256>>>(out, 1.f, 0.f); }));
    printf("straight 48 KiB %.2f\n", timeit([&] { k_line<48><<<G, 256>>>(out, 1.f, 0.f); }));
    printf("loop 1 KiB x1   %.2f\n", timeit([&] { k_loop<1><<<G, 256>>>(out, 1.f, 0.f, 1); }));
    printf("loop 1 KiB x4   %.2f\n", timeit([&] { k_loop<1><<<G, 256>>>(out, 1.f, 0.f, 4); }));
    printf("loop 1 KiB x16  %.2f\n", timeit([&] { k_loop<1><<<G, 256>>>(out, 1.f, 0.f, 16); }));
    printf("loop 1 KiB x48  %.2f\n", timeit([&] { k_loop<1><<<G, 256>>>(out, 1.f, 0.f, 48); }));
    return 0;
}
