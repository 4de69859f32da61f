// Probe: DS_MSKOR_B32 on gfx950 -- MEM = (MEM & ~mask) | data, atomically per lane, also when two
// lanes of one instruction hit the same dword with disjoint masks.
// Build: hipcc --offload-arch=gfx950 -O2 -o ds_mskor_probe ds_mskor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(uint32_t* out) {
    __shared__ uint32_t lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0x11111111u * (uint32_t)(i & 15);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(&lds[lane >> 1]);
    const uint32_t mask = (lane & 1) ? 0xFFFF0000u : 0x000000FFu;
    const uint32_t data = (lane & 1) ? 0xABCD0000u : 0x00000042u;
    asm volatile("ds_mskor_b32 %0, %1, %2" ::"v"(addr), "v"(mask), "v"(data) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    uint32_t* d; hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(256);
    if (hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
        uint32_t want = 0x11111111u * (uint32_t)(i & 15);
        if (i < 32) want = (want & ~0xFFFF00FFu) | 0xABCD0042u;
        if (h[i] != want) { if (!bad) printf("dword %d: got %08x want %08x\n", i, h[i], want); ++bad; }
    }
    printf("ds_mskor probe: %d of 256 dwords wrong\n", bad);
    return bad != 0;
}
