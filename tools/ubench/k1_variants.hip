// k1_variants.hip -- micro-benchmark (GPU box), round 3: where does the time of the synthetic-bank generator (k_rir_synth) go?
// Stand-alone copies of its inner loop at config-2 size (P 200, C 8, L 48000: 307 MB), two taps per thread, with variants:
//   0  as shipped: two murmur finalisers per tap pair (u1, u2), log2 / sqrt / sin / cos, AR(1), envelope, delay gate, peak, 8-byte store
//   1  ONE finaliser per pair, u1 / u2 from its upper / lower 16 bits
//   2  variant 1 without the peak tracking
//   3  variant 1 with the positions unrolled by two (two independent chains per thread)
//   6  variant 1 with the positions unrolled by four
//   4  the hash + Box-Muller only (no AR, gate, store): what the arithmetic alone costs
//   5  stores only (zeros): the write-bandwidth floor of this access pattern
// Build: hipcc --offload-arch=gfx950 -O3 k1_variants.hip -o k1_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ __forceinline__ float unit24(uint32_t h) { return __builtin_fmaf((float)(h >> 8), 5.9604644775390625e-8f, 2.98023223876953125e-8f); }
__device__ __forceinline__ float unit16(uint32_t h) { return __builtin_fmaf((float)(h & 0xFFFFu), 1.52587890625e-5f, 7.62939453125e-6f); }
__device__ __forceinline__ void bm2(float u1, float u2, float& g0, float& g1) {
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    g0 = r * __builtin_amdgcn_cosf(u2);
    g1 = r * __builtin_amdgcn_sinf(u2);
}

template <int VAR>
__global__ __launch_bounds__(256) void k(int P, int C, int L, const int* __restrict__ delay, float rho, float srho, float* __restrict__ bank,
                                          unsigned* __restrict__ peak_out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const int64_t CL = (int64_t)C * L;
    if (i >= CL) return;
    const int c = (int)(i / L), t0 = (int)(i - (int64_t)c * L);
    const float te0 = 0.05f * __expf(-(float)t0 * 1e-4f), te1 = 0.05f * __expf(-(float)(t0 + 1) * 1e-4f);
    const uint32_t h1c = fmix32(0x1234567u), h2c = fmix32(0x7654321u);
    float n0 = 0.f, n1 = 0.f, peak = 0.f, sink = 0.f;
    uint64_t ctr = (uint64_t)c * L + t0;
    float* out = bank + i;
    const int* dl = delay + c;
    int d_next = dl[0];
    auto gauss = [&](uint64_t pr, float& g0, float& g1) {
        if (VAR == 0) {
            bm2(unit24(fmix32((uint32_t)pr ^ h1c)), unit24(fmix32((uint32_t)pr ^ h2c)), g0, g1);
        } else {
            const uint32_t a = fmix32((uint32_t)pr ^ h1c);
            bm2(unit16(a >> 16), unit16(a), g0, g1);
        }
    };
    auto step = [&](int q) {
        const int d = d_next;
        if (q + 1 < P) d_next = dl[(int64_t)(q + 1) * C];
        float g0, g1;
        gauss(ctr >> 1, g0, g1);
        if (VAR == 4) { sink += g0 + g1; ctr += (uint64_t)CL; return; }
        n0 = q == 0 ? g0 : rho * n0 + srho * g0;
        n1 = q == 0 ? g1 : rho * n1 + srho * g1;
        float v0 = t0 > d ? te0 * n0 : 0.f, v1 = t0 + 1 > d ? te1 * n1 : 0.f;
        if ((unsigned)(d - t0) < 2u) { if (t0 == d) v0 += 1.f; else v1 += 1.f; }
        if (VAR != 2) peak = fmaxf(peak, fmaxf(fabsf(v0), fabsf(v1)));
        *reinterpret_cast<float2*>(out) = make_float2(v0, v1);
        out += CL;
        ctr += (uint64_t)CL;
    };
    if (VAR == 5) {
        for (int q = 0; q < P; ++q) { *reinterpret_cast<float2*>(out) = make_float2(0.f, 0.f); out += CL; }
        return;
    }
    if (VAR == 3) {
        int q = 0;
        for (; q + 1 < P; q += 2) { step(q); step(q + 1); }
        for (; q < P; ++q) step(q);
    } else if (VAR == 6) {
        int q = 0;
        for (; q + 3 < P; q += 4) { step(q); step(q + 1); step(q + 2); step(q + 3); }
        for (; q < P; ++q) step(q);
    } else {
        for (int q = 0; q < P; ++q) step(q);
    }
    if (VAR == 4) { if (sink == 123.456f) bank[0] = sink; return; }
    if (peak_out && peak > 100.f) atomicMax(peak_out, __float_as_uint(peak));
}

template <int VAR> void run(const char* name, int P, int C, int L, int* delay, float* bank, unsigned* peak) {
    const int64_t CL = (int64_t)C * L;
    const dim3 grid((unsigned)((CL / 2 + 255) / 256));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 20; ++w) k<VAR><<<grid, 256>>>(P, C, L, delay, 0.9f, 0.4359f, bank, peak);      // warm clocks
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) k<VAR><<<grid, 256>>>(P, C, L, delay, 0.9f, 0.4359f, bank, peak);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("variant %d  %-62s %7.1f us per 307 MB bank (sustained, %d back-to-back launches)\n", VAR, name, ms * 1e3 / reps, reps);
}

int main() {
    const int P = 200, C = 8, L = 48000;
    int* delay; float* bank; unsigned* peak;
    hipMalloc(&delay, sizeof(int) * P * C); hipMalloc(&bank, sizeof(float) * (size_t)P * C * L); hipMalloc(&peak, 4);
    std::vector<int> d(P * C);
    for (int i = 0; i < P * C; ++i) d[i] = 100 + (i * 37) % 400;
    hipMemcpy(delay, d.data(), sizeof(int) * P * C, hipMemcpyHostToDevice);
    hipMemset(peak, 0, 4);
    run<0>("as shipped (two finalisers per pair)", P, C, L, delay, bank, peak);
    run<1>("one finaliser per pair, 16 + 16 bit uniforms", P, C, L, delay, bank, peak);
    run<2>("  ... without the peak", P, C, L, delay, bank, peak);
    run<3>("  ... positions unrolled by two", P, C, L, delay, bank, peak);
    run<6>("  ... positions unrolled by four", P, C, L, delay, bank, peak);
    run<4>("hash + Box-Muller only", P, C, L, delay, bank, peak);
    run<5>("stores only", P, C, L, delay, bank, peak);
    return 0;
}
