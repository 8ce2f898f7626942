// clockramp.hip -- does the shader clock ramp after an idle moment?  Launches N multiplier-bound kernels back to back from an
// idle device; lane 0 of block 0 of each reports its duration by the fixed 100 MHz counter (wall_clock64) and by the shader
// clock (clock64): the ratio is the clock the kernel ran at.  build: hipcc --offload-arch=gfx950 -O3 tools/clockramp.hip -o clockramp
// usage: clockramp [launches=60] [idle_ms=50] [iters=20000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mads(unsigned long long* rec, unsigned iters, unsigned long long* sink)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    long long acc = threadIdx.x;
    int a = threadIdx.x * 2654435761u, b = blockIdx.x * 40503u + 17;
#pragma unroll 8
    for (unsigned i = 0; i < iters; ++i) {
        acc += (long long)a * b;  // v_mad_i64_i32
        a += (int)(acc >> 7);
        b ^= (int)acc;
    }
    if (acc == 0x1234567) sink[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        rec[0] = w0;
        rec[1] = wall_clock64() - w0;
        rec[2] = clock64() - c0;
    }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 60;
    const int idle_ms = argc > 2 ? atoi(argv[2]) : 50;
    const unsigned iters = argc > 3 ? (unsigned)atoi(argv[3]) : 20000u;
    unsigned long long *rec, *sink;
    CHECK(hipHostMalloc((void**)&rec, 24 * (size_t)n * 3));
    CHECK(hipMalloc((void**)&sink, 8));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int round = 0; round < 3; ++round) {
        CHECK(hipDeviceSynchronize());
        std::this_thread::sleep_for(std::chrono::milliseconds(round == 2 ? 0 : idle_ms));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_mads, dim3(1024), dim3(256), 0, s, rec + 3 * (round * n + i), iters, sink);
        CHECK(hipStreamSynchronize(s));
    }
    for (int round = 0; round < 3; ++round) {
        printf("# round %d (%s): launch, start_us, dur_us, shader MHz\n", round, round == 2 ? "straight after the previous" : "after an idle pause");
        const unsigned long long t0 = rec[3 * round * n];
        for (int i = 0; i < n; ++i) {
            const unsigned long long* r = rec + 3 * (round * n + i);
            if (i < 12 || i % 8 == 0 || i == n - 1)
                printf("%3d %9.1f %8.1f %7.0f\n", i, (r[0] - t0) / 100.0, r[1] / 100.0, r[2] * 100.0 / r[1]);
        }
    }
    return 0;
}
