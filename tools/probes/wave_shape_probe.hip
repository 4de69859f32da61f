// How does the per-launch time of a "load record -> think -> write a 3 969-byte image per env" kernel
// depend on the number of waves that carry a fixed batch of 4 096 envs?  (diagnostic for DESIGN.md
// section 4: envs per wave; hipcc --offload-arch=gfx950 -O2 -o wave_shape_probe wave_shape_probe.hip)
// EPW envs per wave are processed TOGETHER (one latency chain of `spin` dependent multiply-adds for
// all of them, as a lane-parallel kernel would), only the memory traffic scales with EPW.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;

template <int EPW, bool NT>
__global__ __launch_bounds__(1024) void probe(uint32_t* __restrict__ state, uint8_t* __restrict__ obs, int nenv, int spin,
                                             int S) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * (blockDim.x >> 6) + wave;
    const int e0 = w * EPW;
    if (e0 >= nenv) return;
    uint32_t rec[EPW], body[EPW];
#pragma unroll
    for (int k = 0; k < EPW; ++k) {
        rec[k] = state[(size_t)(e0 + k) * 64 + lane];
        body[k] = state[(size_t)nenv * 64 + (size_t)(e0 + k) * 96 + lane] + state[(size_t)nenv * 64 + (size_t)(e0 + k) * 96 + 32 + lane];
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < EPW; ++k) acc += rec[k] ^ body[k];
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;  // one dependent chain
    uint8_t* img = smem + (size_t)wave * EPW * 4096;
#pragma unroll
    for (int k = 0; k < EPW; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) reinterpret_cast<u32x4*>(img + k * 4096)[j * 64 + lane] = u32x4{acc, acc + 1, acc + 2, acc + 3};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < EPW; ++k) {
        uint8_t* o = obs + (size_t)(e0 + k) * S;
        const int nfull = S >> 4;
        for (int j = lane; j < nfull; j += 64) {
            const u32x4 v = reinterpret_cast<const u32x4*>(img + k * 4096)[j];
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_unaligned*>(o + 16 * j));
            else *reinterpret_cast<u32x4_unaligned*>(o + 16 * j) = v;
        }
        const int tail = (nfull << 4) + lane;
        if (tail < S) o[tail] = img[k * 4096 + tail];
        state[(size_t)(e0 + k) * 64 + lane] = acc + k;
    }
}

template <int EPW, bool NT>
static double run(uint32_t* state, uint8_t* obs, int nenv, int spin, int wpb, int iters) {
    const int waves = (nenv + EPW - 1) / EPW;
    const dim3 grid((waves + wpb - 1) / wpb), block(64 * wpb);
    const size_t lds = (size_t)wpb * EPW * 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((probe<EPW, NT>), grid, block, lds, 0, state, obs, nenv, spin, 3969);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((probe<EPW, NT>), grid, block, lds, 0, state, obs, nenv, spin, 3969);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const int nenv = argc > 1 ? atoi(argv[1]) : 4096;
    const int iters = nenv <= 8192 ? 2000 : 200;
    uint32_t* state; uint8_t* obs;
    hipMalloc(&state, (size_t)nenv * (256 + 384) + 4096);
    hipMalloc(&obs, (size_t)nenv * 3969 + 4096);
    hipMemset(state, 1, (size_t)nenv * (256 + 384));
    printf("nenv=%d: us per launch (back-to-back launches on one stream)\n", nenv);
    printf("%-28s %8s %8s %8s %8s\n", "variant \\ spin", "0", "100", "200", "400");
    const int spins[4] = {0, 100, 200, 400};
#define ROW(EPW, NT, WPB)                                                                \
    {                                                                                    \
        printf("epw=%d nt=%d waves/wg=%d waves=%-6d", EPW, NT, WPB, (nenv + EPW - 1) / EPW); \
        for (int s = 0; s < 4; ++s) printf(" %8.2f", run<EPW, NT>(state, obs, nenv, spins[s], WPB, iters)); \
        printf("\n");                                                                    \
    }
    ROW(1, true, 16) ROW(1, true, 8) ROW(1, true, 4) ROW(1, true, 2) ROW(1, true, 1) ROW(1, false, 8)
    ROW(2, true, 8) ROW(2, true, 4) ROW(2, false, 4)
    ROW(4, true, 4) ROW(4, true, 2) ROW(4, false, 4)
    ROW(8, true, 2) ROW(8, true, 1)
    // latency of ONE wave (grid of one workgroup): launch-to-launch period of a 1-wave kernel
    {
        printf("single wave (1 env), spin 0/100/200/400:");
        for (int s = 0; s < 4; ++s) printf(" %8.2f", run<1, true>(state, obs, 1, spins[s], 1, 500));
        printf("\n");
    }
    hipFree(state); hipFree(obs);
    return 0;
}
