// sgtest.hip -- device check of the variable-time divsteps against the constant-time form (same header as the engine).
// Build (in tools/): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos-evolution_amd/csrc -o sgtest sgtest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include "fp_inv_safegcd.h"
using namespace posevo;

__global__ void k(const int32_t* delta, const uint32_t* f0, const uint32_t* g0, int32_t* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sg_trans a, b;
    out[10 * i + 0] = sg_divsteps_30(delta[i], f0[i], g0[i], a);
    out[10 * i + 1] = a.u; out[10 * i + 2] = a.v; out[10 * i + 3] = a.q; out[10 * i + 4] = a.r;
    out[10 * i + 5] = sg_divsteps_30_var(delta[i], f0[i], g0[i], b);
    out[10 * i + 6] = b.u; out[10 * i + 7] = b.v; out[10 * i + 8] = b.q; out[10 * i + 9] = b.r;
}

int main()
{
    const int n = 1 << 16;
    std::vector<int32_t> d(n), out(10 * n);
    std::vector<uint32_t> f(n), g(n);
    std::mt19937 rng(3);
    for (int i = 0; i < n; ++i) { d[i] = (int)(rng() % 800) - 400; f[i] = rng() | 1u; g[i] = rng(); }
    int32_t *dd, *dout; uint32_t *df, *dg;
    hipMalloc(&dd, 4 * n); hipMalloc(&df, 4 * n); hipMalloc(&dg, 4 * n); hipMalloc(&dout, 40 * n);
    hipMemcpy(dd, d.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(df, f.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dd, df, dg, dout, n);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(out.data(), dout, 40 * n, hipMemcpyDeviceToHost);
    int bad = 0, bad_host = 0;
    for (int i = 0; i < n; ++i) {
        bool ok = true;
        for (int j = 0; j < 5; ++j) ok &= out[10 * i + j] == out[10 * i + 5 + j];
        sg_trans h; int32_t hd = sg_divsteps_30(d[i], f[i], g[i], h);
        bool okh = hd == out[10 * i] && h.u == out[10 * i + 1] && h.v == out[10 * i + 2] && h.q == out[10 * i + 3] && h.r == out[10 * i + 4];
        if (!ok && bad < 5) printf("mismatch %d: delta %d f %08x g %08x | ct %d %d %d %d %d | var %d %d %d %d %d\n", i, d[i], f[i], g[i],
                                   out[10*i], out[10*i+1], out[10*i+2], out[10*i+3], out[10*i+4], out[10*i+5], out[10*i+6], out[10*i+7], out[10*i+8], out[10*i+9]);
        bad += !ok; bad_host += !okh;
    }
    printf("device var != device const-time: %d of %d;  device const-time != host: %d\n", bad, n, bad_host);
    return bad != 0;
}
