// How fast does the MI355X dispatcher start 4096 waves?  (diagnostic; hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int VG>
__global__ __launch_bounds__(256) void probe(unsigned long long* t, int spin, float* sink) {
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) acc[i] = threadIdx.x * 0.5f + i;
    for (int k = 0; k < spin; ++k)
#pragma unroll
        for (int i = 0; i < VG; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    float s = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += acc[i];
    if (s == 12345.f) sink[0] = s + smem[threadIdx.x];
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t0;
}

template <int VG>
void run(const char* tag, int blocks, int threads, size_t lds, int spin) {
    const int waves = blocks * threads / 64;
    unsigned long long* d; float* sink;
    hipMalloc(&d, waves * 8); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(waves);
    double spread = 0, med = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(probe<VG>, dim3(blocks), dim3(threads), lds, 0, d, spin, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        if (rep >= 2) { spread += (h.back() - h.front()) / 100.0 / 4; med += (h[waves / 2] - h.front()) / 100.0 / 4; }
    }
    printf("%-44s waves=%d  first->last start %.2f us, median start %.2f us\n", tag, waves, spread, med);
    hipFree(d); hipFree(sink);
}

int main() {
    run<1>("VGPR sweep: 1", 1024, 256, 0, 2);
    run<8>("VGPR sweep: 8", 1024, 256, 0, 2);
    run<16>("VGPR sweep: 16", 1024, 256, 0, 2);
    run<24>("VGPR sweep: 24", 1024, 256, 0, 2);
    run<32>("VGPR sweep: 32", 1024, 256, 0, 2);
    run<48>("VGPR sweep: 48", 1024, 256, 0, 2);
    run<64>("VGPR sweep: 64", 1024, 256, 0, 2);
    run<96>("VGPR sweep: 96", 1024, 256, 0, 2);
    run<1>("trivial, 1024 x 256 thr", 1024, 256, 0, 0);
    run<1>("trivial + 17.8 KB LDS, 1024 x 256", 1024, 256, 17856, 0);
    run<1>("trivial, 4096 x 64 thr", 4096, 64, 0, 0);
    run<1>("trivial, spins ~3 us, 1024 x 256", 1024, 256, 0, 1500);
    run<40>("40 live VGPRs, spins, 1024 x 256", 1024, 256, 0, 40);
    run<40>("40 VGPRs + 17.8 KB LDS, spins", 1024, 256, 17856, 40);
    run<1>("trivial, 2048 x 256 (8192 waves)", 2048, 256, 0, 0);
    return 0;
}
