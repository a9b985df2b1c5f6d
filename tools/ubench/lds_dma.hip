// lds_dma.hip -- probe (GPU box): semantics of `buffer_load_dwordx4 voff, srd, soff offen lds` on gfx950:
// LDS address = M0 + lane*16 ?  per-lane global address honoured ?  what lands for out-of-range lanes ?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    // lane l fetches 16 bytes of row (l/16), chunk (l%16); rows are 512 floats apart in memory
    unsigned voff = (threadIdx.x >> 4) * 2048 + (threadIdx.x & 15) * 16;
    unsigned ldsbase = 256 * 4;          // destination: lds[256 ...]
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n s_mov_b32 m0, %2\n s_nop 0\n buffer_load_dwordx4 %1, %3, 0 offen lds\n s_mov_b32 m0, %0\n s_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(ldsbase), "s"(r) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float* h = new float[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int nbytes : {16384, 4096 + 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nbytes);
        float r[1024];
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("num_records %d:\n", nbytes);
        printf("  lds[252..259]  : %g %g %g %g | %g %g %g %g   (expect -1 x4 | 0 1 2 3)\n", r[252], r[253], r[254], r[255], r[256], r[257], r[258], r[259]);
        printf("  lds[256+60..]  : %g %g %g %g | row1: %g %g %g %g (expect 60..63 | 512..515)\n", r[316], r[317], r[318], r[319], r[320], r[321], r[322], r[323]);
        printf("  row2 start, row3 end: %g %g ... %g %g then %g (expect 1024 1025 ... 1598 1599 then -1; with num_records 4160: rows>=2 out of range)\n",
               r[256 + 128], r[256 + 129], r[256 + 254], r[256 + 255], r[512]);
        printf("  row2 chunk0 when partially in range: %g %g\n", r[256 + 128 + 16], r[256 + 128 + 17]);
    }
    return 0;
}
