// Back-to-back launches of kernels that do (almost) nothing: the per-launch floor of one stream on this
// GPU for the step kernel's grid shape.   hipcc --offload-arch=gfx950 -O2 -o launch_floor_probe launch_floor_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void k_empty() {}
__global__ void k_store(unsigned* p) { if (threadIdx.x == 0) p[blockIdx.x] = blockIdx.x; }
__global__ void k_store_nt(unsigned* p) { if (threadIdx.x == 0) __builtin_nontemporal_store(blockIdx.x, &p[blockIdx.x]); }
#define ASM_STORE_KERNEL(name, bits)                                                                       \
    __global__ void name(unsigned* p) {                                                                    \
        if (threadIdx.x == 0) {                                                                            \
            unsigned* q = p + blockIdx.x; unsigned v = blockIdx.x;                                          \
            asm volatile("global_store_dword %0, %1, off " bits :: "v"(q), "v"(v) : "memory");             \
        }                                                                                                  \
    }
ASM_STORE_KERNEL(k_store_sc0, "sc0")
ASM_STORE_KERNEL(k_store_sc1, "sc1")
ASM_STORE_KERNEL(k_store_sc0sc1, "sc0 sc1")
ASM_STORE_KERNEL(k_store_ntsc0sc1, "nt sc0 sc1")
// every lane of every wave stores 16 bytes (the step kernel's copy-out shape: 4 KiB per wave in four instructions)
__global__ void k_store_wide(uint4* p, int nt) {
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 0, 0);
    uint4* q = p + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (int i = 0; i < 4; ++i) { if (nt) __builtin_nontemporal_store(v.x, &q[i].x); else q[i] = v; }
}
__global__ void k_load_store(const unsigned* __restrict__ a, unsigned* __restrict__ b) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    b[i] = a[i] + 1u;
}

template <typename F>
static double per_launch_us(F launch, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> v;
    for (int r = 0; r < 7; ++r) {
        for (int i = 0; i < 64; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        v.push_back(ms * 1e3 / iters);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    unsigned *a, *b;
    hipMalloc(&a, 1024 * 256 * 4); hipMalloc(&b, 1024 * 256 * 4);
    hipMemset(a, 0, 1024 * 256 * 4);
    const int shapes[][2] = {{1, 64}, {256, 256}, {1024, 256}, {2048, 256}, {4096, 256}};
    for (auto& s : shapes) {
        const dim3 g(s[0]), t(s[1]);
        const double e = per_launch_us([&] { hipLaunchKernelGGL(k_empty, g, t, 0, 0); }, 2000);
        const double st = per_launch_us([&] { hipLaunchKernelGGL(k_store, g, t, 0, 0, b); }, 2000);
        const double ls = (size_t)s[0] * s[1] <= 1024u * 256u ? per_launch_us([&] { hipLaunchKernelGGL(k_load_store, g, t, 0, 0, a, b); }, 2000) : 0.0;
        printf("grid %4d x %3d threads (%5d waves): empty %.2f us, one store per workgroup %.2f us, load+store per thread %.2f us per launch\n",
               s[0], s[1], s[0] * s[1] / 64, e, st, ls);
        printf("      one store per workgroup with nt %.2f, sc0 %.2f, sc1 %.2f, sc0 sc1 %.2f, nt sc0 sc1 %.2f us\n",
               per_launch_us([&] { hipLaunchKernelGGL(k_store_nt, g, t, 0, 0, b); }, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_store_sc0, g, t, 0, 0, b); }, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_store_sc1, g, t, 0, 0, b); }, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_store_sc0sc1, g, t, 0, 0, b); }, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_store_ntsc0sc1, g, t, 0, 0, b); }, 2000));
    }
    return 0;
}
