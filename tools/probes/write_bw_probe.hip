// Probe: what does a pure store stream reach on this chip?  One kernel writes N bytes with 16-byte-per-lane stores
// (1 KiB contiguous per wave instruction, like the observation copy-out), plain or nt, at several sizes; hipMemsetAsync
// for comparison.  The step kernel's copy-out is judged against THIS figure, not against the 8 TB/s spec peak alone.
// Build: hipcc --offload-arch=gfx950 -O2 -o write_bw_probe write_bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int PER_WAVE_KIB>
__global__ __launch_bounds__(256) void fill(u32x4* __restrict__ out, size_t n16, unsigned v) {
    // each wave owns PER_WAVE_KIB consecutive KiB (an "image"), waves of a block consecutive images
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const u32x4 x = {v, v + 1, v + 2, v + 3};
    size_t base = wave * (size_t)PER_WAVE_KIB * 64;
#pragma unroll
    for (int k = 0; k < PER_WAVE_KIB; ++k) {
        const size_t i = base + (size_t)k * 64 + lane;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(x, &out[i]);
            else out[i] = x;
        }
    }
}

// persistent variant: `nwaves` resident waves sweep the images; COOP = the four waves of a workgroup write every image
// together (each instruction covers 4 KiB contiguous across the workgroup) instead of one image per wave
template <int NT, int COOP>
__global__ __launch_bounds__(256) void sweep(u32x4* __restrict__ out, size_t nimg, int img16, unsigned v) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32x4 x = {v, v + 1, v + 2, v + 3};
    if (COOP) {
        for (size_t im = (size_t)blockIdx.x * 4; im < nimg; im += (size_t)gridDim.x * 4) {
            const size_t base = im * (size_t)img16;  // 4 consecutive images = one contiguous run
            const size_t n = ((im + 4 <= nimg) ? 4 : (nimg - im)) * (size_t)img16;
            for (size_t i = (size_t)w * 64 + lane; i < n; i += 256) {
                if (NT) __builtin_nontemporal_store(x, &out[base + i]); else out[base + i] = x;
            }
        }
    } else {
        for (size_t im = (size_t)blockIdx.x * 4 + w; im < nimg; im += (size_t)gridDim.x * 4) {
            const size_t base = im * (size_t)img16;
            for (int i = lane; i < img16; i += 64) {
                if (NT) __builtin_nontemporal_store(x, &out[base + i]); else out[base + i] = x;
            }
        }
    }
}

template <int NT, int COOP>
static double run_sweep(u32x4* d, size_t bytes, int img_bytes, int blocks, int iters) {
    const size_t nimg = bytes / (size_t)img_bytes;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((sweep<NT, COOP>), dim3(blocks), dim3(256), 0, 0, d, nimg, img_bytes / 16, 1u);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((sweep<NT, COOP>), dim3(blocks), dim3(256), 0, 0, d, nimg, img_bytes / 16, (unsigned)i);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)nimg * img_bytes * iters / (ms * 1e-3) / 1e12;
}

template <int NT, int PW>
static double run(u32x4* d, size_t bytes, int iters) {
    const size_t n16 = bytes / 16;
    const size_t waves = (n16 + (size_t)PW * 64 - 1) / ((size_t)PW * 64);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill<NT, PW>), dim3(blocks), dim3(256), 0, 0, d, n16, 1u);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fill<NT, PW>), dim3(blocks), dim3(256), 0, 0, d, n16, (unsigned)i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)bytes * iters / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t sizes[] = {16ull << 20, 130ull << 20, 260ull << 20, 1040ull << 20, 2080ull << 20, 4160ull << 20};
    u32x4* d;
    if (hipMalloc(&d, sizes[5]) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("%10s %12s %12s %12s %12s %12s\n", "MiB", "plain 4KiB/w", "nt 4KiB/w", "plain 62K/w", "nt 62KiB/w", "memset");
    for (size_t s : sizes) {
        const int iters = s <= (260ull << 20) ? 200 : 40;
        const double a = run<0, 4>(d, s, iters), b = run<1, 4>(d, s, iters), c = run<0, 62>(d, s, iters), e = run<1, 62>(d, s, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipMemsetAsync(d, 1, s, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) hipMemsetAsync(d, i, s, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%10zu %12.2f %12.2f %12.2f %12.2f %12.2f   TB/s\n", s >> 20, a, b, c, e, (double)s * iters / (ms * 1e-3) / 1e12);
    }
    printf("\npersistent sweeps of 2080 MiB in 63 504-byte images (the fused 84x84x9 frame): TB/s by resident workgroups (4 waves each)\n");
    printf("%8s %14s %14s %14s %14s\n", "blocks", "wave/img plain", "wave/img nt", "coop plain", "coop nt");
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
        const size_t s = 2080ull << 20;
        printf("%8d %14.2f %14.2f %14.2f %14.2f\n", blocks, run_sweep<0, 0>(d, s, 63504, blocks, 20), run_sweep<1, 0>(d, s, 63504, blocks, 20),
               run_sweep<0, 1>(d, s, 63504, blocks, 20), run_sweep<1, 1>(d, s, 63504, blocks, 20));
    }
    printf("\nthe same for 1040 MiB in 3 969-byte images rounded to 3 968 (native 21x21x9 frame)\n");
    for (int blocks : {512, 2048, 8192, 65536}) {
        const size_t s = 1040ull << 20;
        printf("%8d %14.2f %14.2f %14.2f %14.2f\n", blocks, run_sweep<0, 0>(d, s, 3968, blocks, 20), run_sweep<1, 0>(d, s, 3968, blocks, 20),
               run_sweep<0, 1>(d, s, 3968, blocks, 20), run_sweep<1, 1>(d, s, 3968, blocks, 20));
    }
    hipFree(d);
    return 0;
}
