// lanebench.hip -- does a wave with FEWER ACTIVE LANES issue its quarter-rate multiply-adds faster?  The decompression kernels
// run one chain of ~1000 dependent S29 products per point on 8192 points = 128 full waves on 128 of the chip's 1024 SIMDs
// (DESIGN.md 3.6): if a wave of 16 active lanes took a quarter of the issue time, four times as many quarter-full waves on four
// times as many SIMDs would shorten the chain fourfold with a change of launch geometry alone.
// One workgroup of 64 lanes per CU-SIMD slot (grid = 128 waves), `active` lanes of each run a chain of S29 squarings, the others
// exit at once.  Prints ns per squaring for 64 / 32 / 16 / 8 active lanes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o lanebench lanebench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "fp381_s29.h"

using namespace posevo;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_chain(int32_t* out, int iters, int active)
{
    const int lane = threadIdx.x;
    if (lane >= active) return;
    fq a;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) a.l[i] = (int32_t)((blockIdx.x * 64 + lane) * 2654435761u + i * 40503u) >> 4;
#pragma nounroll
    for (int k = 0; k < iters; ++k) fq_sqr(a, a);
    int32_t x = 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) x ^= a.l[i];
    out[blockIdx.x * 64 + lane] = x;
}

int main()
{
    int32_t* d;
    CHECK(hipMalloc(&d, 4 * 64 * 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int waves : {128, 1024}) {
        for (int active : {64, 32, 16, 8, 1}) {
            hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, d, 100, active);
            CHECK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, d, iters, active);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("waves %4d active lanes %2d: %.3f ms for %d dependent squarings = %.1f ns per squaring (301 multiply-adds: %.2f ns each)\n",
                   waves, active, best, iters, best * 1e6 / iters, best * 1e6 / iters / 301);
        }
    }
    return 0;
}
