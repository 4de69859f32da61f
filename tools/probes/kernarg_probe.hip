// Probe (shared library, loaded into the measuring process with ctypes AFTER torch): where do this process's kernel
// arguments live?  Every launch gets a fresh slice of the HIP runtime's kernel-argument pool; the kernel times one
// scalar load from a part of its own argument segment that nothing has touched yet (s_memrealtime, 100 MHz).  Arguments
// in device memory answer in well under a microsecond, arguments in host memory cost a PCIe round trip.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libkernarg_probe.so kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
struct Big { uint32_t x[96]; };  // 384 bytes by value: far beyond anything a previous packet's prefetch could have pulled in

__global__ void kernarg_latency(unsigned long long* out, int slot, Big big) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t v = big.x[95];            // scalar load from the argument segment (cold)
    asm volatile("" : "+s"(v));        // (wait for it here)
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        out[slot * 4 + 0] = t1 - t0;   // cost of one s_memrealtime round trip (calibration)
        out[slot * 4 + 1] = t2 - t1;   // argument load + one s_memrealtime round trip
        out[slot * 4 + 2] = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        out[slot * 4 + 3] = v;
    }
}

// which XCD does workgroup b of a grid land on in this process's queue?  out[b] = XCC id
__global__ void xcc_of_block(unsigned long long* out) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 15u;
}

extern "C" int xcc_probe(unsigned long long* out_dev, int blocks, void* stream) {
    hipLaunchKernelGGL(xcc_of_block, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out_dev);
    return (int)hipGetLastError();
}

extern "C" int kernarg_probe(unsigned long long* out_dev, int n, void* stream) {
    Big b;
    for (int i = 0; i < 96; ++i) b.x[i] = 1000u + (uint32_t)i;
    for (int i = 0; i < n; ++i) {
        b.x[95] = 7000u + (uint32_t)i;
        hipLaunchKernelGGL(kernarg_latency, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, i, b);
    }
    return (int)hipGetLastError();
}
