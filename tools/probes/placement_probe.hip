// Where does the dispatcher put the workgroups of a 4 096-wave launch?  Waves per CU / per XCD for the
// launch shapes of the step kernel (diagnostic; hipcc --offload-arch=gfx950 -O2 -o placement_probe placement_probe.hip)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#define GETREG(id, off, sz) __builtin_amdgcn_s_getreg((((sz) - 1) << 11) | ((off) << 6) | (id))

__global__ __launch_bounds__(1024) void probe(unsigned* out, int spin) {
    extern __shared__ char smem[];
    const unsigned hw = GETREG(4, 0, 32);    // HW_REG_HW_ID
    const unsigned xcc = GETREG(20, 0, 4);   // HW_REG_XCC_ID
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
    if (acc == 12345u) smem[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[w * 4 + 0] = hw; out[w * 4 + 1] = xcc; out[w * 4 + 2] = (unsigned)t0; out[w * 4 + 3] = acc;
    }
}

static void run(const char* tag, int blocks, int threads, size_t lds, int spin) {
    const int waves = blocks * threads / 64;
    unsigned* d;
    hipMalloc(&d, waves * 16);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<unsigned> h(waves * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, spin);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_xcc;
    for (int w = 0; w < waves; ++w) {
        const unsigned hw = h[w * 4], xcc = h[w * 4 + 1];
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%-40s waves=%d CUs used=%zu  waves-per-CU histogram:", tag, waves, per_cu.size());
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("   per XCD:");
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    printf("\n");
    hipFree(d);
}

int main() {
    run("512 x 512 thr, 35.7 KB LDS, spin 300", 512, 512, 35712, 300);
    run("512 x 512 thr, 35.7 KB LDS, spin 0", 512, 512, 35712, 0);
    run("512 x 512 thr, 64 KB LDS, spin 300", 512, 512, 65536, 300);
    run("512 x 512 thr, 80 KB LDS, spin 300", 512, 512, 81920, 300);
    run("1024 x 256 thr, 17.9 KB LDS, spin 300", 1024, 256, 17856, 300);
    run("256 x 1024 thr, 71.4 KB LDS, spin 300", 256, 1024, 71424, 300);
    run("4096 x 64 thr, 4.4 KB LDS, spin 300", 4096, 64, 4464, 300);
    return 0;
}
