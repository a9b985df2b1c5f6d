// A kernel that occupies `n` compute units for `usec` microseconds (each workgroup takes the whole LDS of a CU and spins on the wall
// clock): stands in for the RCCL send / recv kernels that share the GPU with the persistent render kernel in a multi-GPU run.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/squat.hip -o tools/var/libsquat.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void k_squat(long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == -1) *sink = 1;
}

extern "C" int squat(int n, int usec, void* stream) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)k_squat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        once = true;
    }
    static int* sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_squat, dim3(n), dim3(256), 160 * 1024, (hipStream_t)stream, (long long)usec * 100, sink);   // wall clock: 100 MHz
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
