// fpbench.hip -- correctness + throughput of the device Fp381 multiply and the mixed G1 add.
// Build: hipcc --offload-arch=gfx950 -O3 -I../pos_evolution_amd/csrc -o fpbench fpbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "g1.h"

using namespace posevo;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- host reference (6x64 CIOS), independent of the device code ----
typedef unsigned __int128 u128;
static const uint64_t HP[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                               0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static void host_mont_mul(uint64_t* r, const uint64_t* a, const uint64_t* b)
{
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; ++i) {
        u128 c = 0;
        for (int j = 0; j < 6; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * 0x89f3fffcfffcfffdULL;
        c = (u128)m * HP[0] + t[0]; c >>= 64;
        for (int j = 1; j < 6; ++j) { c += (u128)m * HP[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
    }
    bool ge = t[6] != 0;
    if (!ge) { ge = true; for (int i = 5; i >= 0; --i) { if (t[i] > HP[i]) break; if (t[i] < HP[i]) { ge = false; break; } } }
    if (ge) { u128 br = 0; for (int i = 0; i < 6; ++i) { u128 d = (u128)t[i] - HP[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    memcpy(r, t, 48);
}

// comparison arm: one asm statement per MAC (274 s_nops per product) -- what fp_mul was before tools/gen_fp_mul.py
#define POSEVO_MAC1(lo64, hi, a, b) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo64), "+v"(hi) : "v"(a), "v"(b) : "vcc")
__device__ __forceinline__ void fp_mul_fused(fp& r, const fp& a, const fp& b)  // name kept: the "other" arm
{
    uint32_t m[12], t[13];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) POSEVO_MAC1(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) { uint32_t pj = fp_p_limb(k - i); POSEVO_MAC1(lo, hi, m[i], pj); }
        m[k] = (uint32_t)lo * FP_N0;
        { uint32_t p0 = fp_p_limb(0); POSEVO_MAC1(lo, hi, m[k], p0); }
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = 12; k < 23; ++k) {
#pragma unroll
        for (int i = k - 11; i < 12; ++i) POSEVO_MAC1(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 11; i < 12; ++i) { uint32_t pj = fp_p_limb(k - i); POSEVO_MAC1(lo, hi, m[i], pj); }
        t[k - 12] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[11] = (uint32_t)lo;
    t[12] = (uint32_t)(lo >> 32);
    uint32_t s[12], br = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = __builtin_subc(t[j], fp_p_limb(j), br, &br);
    const bool ge = (t[12] != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = ge ? s[j] : t[j];
}
__global__ void k_mulf_check(const fp* a, const fp* b, fp* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { fp r; fp_mul_fused(r, a[i], b[i]); out[i] = r; }
}
__global__ void __launch_bounds__(256) k_mulf_chain(fp* x, int iters, uint64_t* cyc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    fp a = x[i], b = x[i + 1];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) { fp c; fp_mul_fused(c, a, b); a = b; b = c; }
    uint64_t t1 = __builtin_readcyclecounter();
    x[i] = b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_mul_check(const fp* a, const fp* b, fp* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { fp r; fp_mul(r, a[i], b[i]); out[i] = r; }
}
__global__ void k_addsub_check(const fp* a, const fp* b, fp* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { fp s, d, r; fp_add(s, a[i], b[i]); fp_sub(d, a[i], b[i]); fp_mul(r, s, d); out[i] = r; } // (a+b)(a-b)
}

__global__ void __launch_bounds__(256) k_mul_chain(fp* x, int iters, uint64_t* cyc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    fp a = x[i], b = x[i + 1];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) { fp c; fp_mul(c, a, b); a = b; b = c; }
    uint64_t t1 = __builtin_readcyclecounter();
    x[i] = b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// two independent chains per lane (ILP)
__global__ void __launch_bounds__(256) k_mul_chain2(fp* x, int iters, uint64_t* cyc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    fp a = x[i], b = x[i + 1], c = x[i + 2], d = x[i + 3];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) { fp e, f; fp_mul(e, a, b); fp_mul(f, c, d); a = b; b = e; c = d; d = f; }
    uint64_t t1 = __builtin_readcyclecounter();
    fp_add(b, b, d);
    x[i] = b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_madd_chain(const fp* pts, g1x* out, int iters, uint64_t* cyc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    g1x acc;
    acc.x = pts[2 * i]; acc.y = pts[2 * i + 1]; fp_set_one(acc.zz); fp_set_one(acc.zzz);
    fp qx = pts[2 * i + 2], qy = pts[2 * i + 3];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; ++k) { g1x_add_affine(acc, qx, qy, false); fp_add(qx, qx, acc.zz); }
    uint64_t t1 = __builtin_readcyclecounter();
    out[i] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int N = 1 << 16;
    std::mt19937_64 rng(42);
    std::vector<uint64_t> ha(N * 6), hb(N * 6), hr(N * 6), hexp(N * 6);
    auto rnd_fp = [&](uint64_t* o) { for (int j = 0; j < 6; ++j) o[j] = rng(); o[5] &= 0x0fffffffffffffffULL; /* < 2^380 < p */ };
    for (int i = 0; i < N; ++i) { rnd_fp(&ha[6 * i]); rnd_fp(&hb[6 * i]); }
    // edge values
    memset(&ha[0], 0, 48); for (int j = 0; j < 6; ++j) { ha[6 + j] = HP[j]; hb[6 + j] = HP[j]; } ha[6] -= 1; hb[6] -= 1; // p-1
    for (int j = 0; j < 6; ++j) ha[12 + j] = ~0ULL >> (j == 5 ? 4 : 0);
    fp *da, *db, *dr;
    CHECK(hipMalloc(&da, N * 48)); CHECK(hipMalloc(&db, N * 48)); CHECK(hipMalloc(&dr, (N + 8) * 48));
    CHECK(hipMemcpy(da, ha.data(), N * 48, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), N * 48, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul_check, dim3(N / 256), dim3(256), 0, 0, da, db, dr, N);
    CHECK(hipMemcpy(hr.data(), dr, N * 48, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) { host_mont_mul(&hexp[6 * i], &ha[6 * i], &hb[6 * i]); if (memcmp(&hexp[6 * i], &hr[6 * i], 48)) ++bad; }
    printf("fp_mul check: %d / %d mismatches\n", bad, N);
    hipLaunchKernelGGL(k_mulf_check, dim3(N / 256), dim3(256), 0, 0, da, db, dr, N);
    CHECK(hipMemcpy(hr.data(), dr, N * 48, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 0; i < N; ++i) if (memcmp(&hexp[6 * i], &hr[6 * i], 48)) ++bad;
    printf("fp_mul (one asm per MAC) check: %d / %d mismatches\n", bad, N);
    // (a+b)(a-b) == a^2 - b^2 : checks add/sub against mul
    hipLaunchKernelGGL(k_addsub_check, dim3(N / 256), dim3(256), 0, 0, da, db, dr, N);
    CHECK(hipMemcpy(hr.data(), dr, N * 48, hipMemcpyDeviceToHost));
    bad = 0;
    for (int i = 3; i < N; ++i) {
        uint64_t a2[6], b2[6], d[6];
        host_mont_mul(a2, &ha[6 * i], &ha[6 * i]); host_mont_mul(b2, &hb[6 * i], &hb[6 * i]);
        u128 br = 0; for (int j = 0; j < 6; ++j) { u128 x = (u128)a2[j] - b2[j] - br; d[j] = (uint64_t)x; br = (x >> 64) & 1; }
        if (br) { u128 c = 0; for (int j = 0; j < 6; ++j) { c += (u128)d[j] + HP[j]; d[j] = (uint64_t)c; c >>= 64; } }
        if (memcmp(d, &hr[6 * i], 48)) ++bad;
    }
    printf("fp_add/sub check: %d mismatches\n", bad);

    uint64_t* dcyc; CHECK(hipMalloc(&dcyc, 4096 * 8));
    g1x* dj; CHECK(hipMalloc(&dj, 256 * 4 * 256 * sizeof(g1x)));
    fp* dx; CHECK(hipMalloc(&dx, (256 * 4 * 256 * 2 + 8) * 48));
    for (size_t off = 0; off < (size_t)256 * 4 * 256 * 2; off += N) CHECK(hipMemcpy(dx + off, da, (size_t)N * 48, hipMemcpyDeviceToDevice));
    auto bench = [&](const char* name, int which, int iters, double mul_per_iter) -> int {
        for (int wps = 1; wps <= 4; wps *= 2) {
            int blocks = 256 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_mul_chain, dim3(blocks), dim3(256), 0, 0, dx, iters, dcyc);
                if (which == 1) hipLaunchKernelGGL(k_mul_chain2, dim3(blocks), dim3(256), 0, 0, dx, iters, dcyc);
                if (which == 3) hipLaunchKernelGGL(k_mulf_chain, dim3(blocks), dim3(256), 0, 0, dx, iters, dcyc);
                if (which == 2) hipLaunchKernelGGL(k_madd_chain, dim3(blocks), dim3(256), 0, 0, dx, dj, iters, dcyc);
                hipEventRecord(e1);
                CHECK(hipDeviceSynchronize());
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> cyc(blocks); hipMemcpy(cyc.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : cyc) avg += c; avg /= blocks;
            double total_ops = (double)blocks * 256 * iters;
            printf("%-12s wps=%d: %.0f ticks/iter/wave, %.3f ms, %.2f G iter/s chip (%.2f G fpmul/s)\n", name, wps,
                   avg / iters, ms, total_ops / ms / 1e6, total_ops * mul_per_iter / ms / 1e6);
        }
        return 0;
    };
    bench("mul_chain", 0, 2000, 1);
    bench("mul_1asm", 3, 2000, 1);
    bench("mul_chain", 0, 2000, 1);
    bench("mul_1asm", 3, 2000, 1);
    bench("mul_chain2", 1, 1000, 2);
    bench("madd_chain", 2, 300, 10);
    return 0;
}
