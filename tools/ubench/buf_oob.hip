// buf_oob.hip -- does the raw-buffer range check on gfx950 include the SGPR offset?  (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 64, 0x00020000);   // 16 floats valid
    const unsigned t = threadIdx.x;
    out[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, t * 4, 0, 0));            // voffset only: lanes >= 16 out of range
    out[64 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 0, 128, 0));        // soffset beyond num_records, voffset in range
    out[128 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 60, 8, 0));        // voffset in range, voffset+soffset out of range
    out[192 + t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, t * 4, 32, 0));    // mixed
}
int main() {
    float h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("voffset only : lane15 %.0f lane16 %.0f (expect 115, 0)\n", r[15], r[16]);
    printf("soffset 128, voffset 0 : %.0f  (0 => soffset IS range checked; 132 => it is not)\n", r[64]);
    printf("voffset 60 + soffset 8 : %.0f  (0 => sum checked; 117 => only voffset checked)\n", r[128]);
    printf("voffset t*4 + soffset 32: lane7 %.0f lane8 %.0f lane15 %.0f lane16 %.0f\n", r[192 + 7], r[192 + 8], r[192 + 15], r[192 + 16]);
    return 0;
}
