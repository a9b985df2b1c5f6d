// wg_placement.hip -- where does the dispatcher put the workgroups of a grid that fits the machine several times over?  Each workgroup records its
// (XCC, SE, CU) and spins ~20 us so that all of them are resident together.  usage: ./wg_placement <workgroups> <threads> <lds_bytes>
// Build: hipcc --offload-arch=gfx950 -O3 wg_placement.hip -o wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k(unsigned* out, int spin, int lds) {
    extern __shared__ char sm[];
    if (lds && threadIdx.x == 0) sm[0] = 1;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 750, nt = argc > 2 ? atoi(argv[2]) : 256, lds = argc > 3 ? atoi(argv[3]) : 0;
    unsigned* d; hipMalloc(&d, 8 * nwg);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(nt), lds, 0, d, 2000, lds);   // 100 MHz clock: 2000 ticks = 20 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * nwg); hipMemcpy(h.data(), d, 8 * nwg, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu; std::map<unsigned, int> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; per_xcc[xcc]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%d workgroups of %d threads, %d B LDS: %zu distinct CUs;", nwg, nt, lds, per_cu.size());
    for (auto& kv : hist) printf(" %d CUs hold %d;", kv.second, kv.first);
    printf(" per XCC:");
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    printf("\n");
    return 0;
}
