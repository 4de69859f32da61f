// Probe: does global_load_lds_dwordx4 (gfx950) place lane l's 16 bytes at M0base + 16*l, and is the
// data visible to ds_read after s_waitcnt vmcnt(0)?  Build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    uint8_t* img = lds + w * 4096;
    for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(img + q * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int q = 0; q < 4; ++q)
        out[(blockIdx.x * 4 + w) * 256 + q * 64 + lane] = reinterpret_cast<uint4*>(img)[q * 64 + lane];
}

int main() {
    const int blocks = 64, waves = blocks * 4;
    std::vector<uint32_t> h(1024), o((size_t)waves * 1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x9E3779B9u * (uint32_t)(i + 1);
    uint4 *ds, *dd;
    hipMalloc(&ds, 4096);
    hipMalloc(&dd, (size_t)waves * 4096);
    hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    hipMemset(dd, 0, (size_t)waves * 4096);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, ds, dd);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(o.data(), dd, (size_t)waves * 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < waves; ++w)
        if (memcmp(&o[(size_t)w * 1024], h.data(), 4096)) ++bad;
    printf("lds dma probe: %d of %d waves differ\n", bad, waves);
    if (bad) {
        for (int i = 0; i < 1024 && bad; ++i)
            if (o[i] != h[i]) { printf("first diff at dword %d: got %08x want %08x\n", i, o[i], h[i]); break; }
    }
    return bad ? 1 : 0;
}
