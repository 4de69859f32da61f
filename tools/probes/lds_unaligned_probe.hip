// Probe (result: correct but SLOW -- 7 such stores per wave doubled the step kernel, 7.8 -> 16.5 us):
// do misaligned ds_write_b64 (the compiler emits them for align-1 stores on gfx950) land
// correctly in LDS?  Lane l writes 9 bytes at offset 9*l + shift for every shift 0..15.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_unaligned_probe lds_unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint64_t __attribute__((aligned(1))) u64_u;

__global__ void k(int shift, uint8_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = 0;
    __syncthreads();
    const int off = 9 * lane + shift;
    *reinterpret_cast<u64_u*>(lds + off) = 0x0807060504030201ull + 0x0101010101010101ull * (uint64_t)(lane & 7);
    lds[off + 8] = (uint8_t)(9 + (lane & 7));
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
    uint8_t* d; hipMalloc(&d, 1024);
    int bad = 0;
    for (int shift = 0; shift < 16; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, shift, d);
        std::vector<uint8_t> h(1024), want(1024, 0);
        if (hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
        for (int l = 0; l < 64; ++l)
            for (int b = 0; b < 9; ++b) want[9 * l + shift + b] = (uint8_t)(1 + b + (l & 7));
        if (memcmp(h.data(), want.data(), 1024)) { ++bad; printf("shift %d: mismatch\n", shift); }
    }
    printf("lds unaligned probe: %d of 16 shifts wrong\n", bad);
    return bad != 0;
}
