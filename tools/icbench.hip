// icbench.hip -- does the accumulation pay for its code size?  k_g1_accumulate's loop body (ten fully unrolled Montgomery
// products) is ~47 KB of instructions, the S29 kernel ~214 KB in all, k_g1_tree 252 KB; two CUs share a 64 KB instruction
// cache.  This benchmark runs the SAME dependent chain of mixed adds with U distinct copies of the add's code in the loop
// (U = 1, 2, 3: ~45 / 90 / 135 KB), and with the field product as a CALLED function (one copy of the product's code).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o icbench icbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "g1.h"
#include "g1_s29.h"

using namespace posevo;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __noinline__ fp fp_mul_call(fp a, fp b) { fp r; fp_mul(r, a, b); return r; }
__device__ __noinline__ fq fq_mul_call(fq a, fq b) { fq r; fq_mul(r, a, b); return r; }
__device__ __noinline__ fq fq_sqr_call(fq a) { fq r; fq_sqr(r, a); return r; }

template <int CALL> __device__ __forceinline__ void MUL(fp& r, const fp& a, const fp& b)
{
    if (CALL) r = fp_mul_call(a, b); else fp_mul(r, a, b);
}
template <int CALL> __device__ __forceinline__ void MULQ(fq& r, const fq& a, const fq& b)
{
    if (CALL) r = fq_mul_call(a, b); else fq_mul(r, a, b);
}
template <int CALL> __device__ __forceinline__ void SQRQ(fq& r, const fq& a)
{
    if (CALL) r = fq_sqr_call(a); else fq_sqr(r, a);
}

// the general body of g1x_add_affine (madd-2008-s), no edge cases: what a lane's steady state executes
template <int CALL> __device__ __forceinline__ void madd32(g1x& acc, const fp& qx, const fp& qy)
{
    fp U2, S2, P, R, PP, PPP, Q, X3, t;
    MUL<CALL>(U2, qx, acc.zz);
    MUL<CALL>(S2, qy, acc.zzz);
    fp_sub(P, U2, acc.x);
    fp_sub(R, S2, acc.y);
    MUL<CALL>(PP, P, P);
    MUL<CALL>(PPP, P, PP);
    MUL<CALL>(Q, acc.x, PP);
    MUL<CALL>(X3, R, R);
    fp_sub(X3, X3, PPP);
    fp_dbl(t, Q);
    fp_sub(X3, X3, t);
    fp_sub(t, Q, X3);
    MUL<CALL>(t, R, t);
    MUL<CALL>(Q, acc.y, PPP);
    fp_sub(acc.y, t, Q);
    MUL<CALL>(acc.zz, acc.zz, PP);
    MUL<CALL>(acc.zzz, acc.zzz, PPP);
    acc.x = X3;
}
template <int CALL> __device__ __forceinline__ void madd29(g1q& acc, const fq& qx, const fq& qy)
{
    fq U2, S2, P, R, PP, PPP, Q, X3, t, u, zz, zzz;
    MULQ<CALL>(U2, qx, acc.zz);
    MULQ<CALL>(S2, qy, acc.zzz);
    fq_sub(P, U2, acc.x);
    fq_sub(R, S2, acc.y);
    SQRQ<CALL>(PP, P);
    MULQ<CALL>(PPP, P, PP);
    MULQ<CALL>(Q, acc.x, PP);
    SQRQ<CALL>(X3, R);
    fq_sub_sub2_norm(X3, X3, PPP, Q);
    fq_sub(t, Q, X3);
    MULQ<CALL>(t, R, t);
    MULQ<CALL>(u, acc.y, PPP);
    fq_sub_norm(acc.y, t, u);
    MULQ<CALL>(zz, acc.zz, PP);
    MULQ<CALL>(zzz, acc.zzz, PPP);
    acc.zz = zz;
    acc.zzz = zzz;
    acc.x = X3;
}

template <int U, int CALL> __global__ void __launch_bounds__(256) k_chain32(const fp* pts, g1x* out, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    g1x acc;
    acc.x = pts[2 * i]; acc.y = pts[2 * i + 1]; fp_set_one(acc.zz); fp_set_one(acc.zzz);
    fp qx = pts[2 * i + 2], qy = pts[2 * i + 3];
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        madd32<CALL>(acc, qx, qy); fp_add(qx, qx, acc.zz);
        if constexpr (U >= 2) { madd32<CALL>(acc, qx, qy); fp_add(qx, qx, acc.zz); }
        if constexpr (U >= 3) { madd32<CALL>(acc, qx, qy); fp_add(qx, qx, acc.zz); }
        if constexpr (U >= 4) { madd32<CALL>(acc, qx, qy); fp_add(qx, qx, acc.zz); }
    }
    out[i] = acc;
}
template <int U, int CALL> __global__ void __launch_bounds__(256) k_chain29(const fq* pts, fq* out, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    g1q acc;
    acc.x = pts[2 * i]; acc.y = pts[2 * i + 1]; fq_set_one(acc.zz); fq_set_one(acc.zzz); acc.inf = false; acc.affine = false;
    fq qx = pts[2 * i + 2], qy = pts[2 * i + 3];
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        madd29<CALL>(acc, qx, qy); fq_sub_norm(qx, qx, acc.zz);
        if constexpr (U >= 2) { madd29<CALL>(acc, qx, qy); fq_sub_norm(qx, qx, acc.zz); }
        if constexpr (U >= 3) { madd29<CALL>(acc, qx, qy); fq_sub_norm(qx, qx, acc.zz); }
        if constexpr (U >= 4) { madd29<CALL>(acc, qx, qy); fq_sub_norm(qx, qx, acc.zz); }
    }
    out[i] = acc.x;
}

template <typename K, typename... A> static float run(K kern, int blocks, A... args)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main()
{
    const size_t LANES = (size_t)256 * 4 * 256;
    std::vector<uint32_t> h((LANES * 2 + 8) * 14);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s & 0x0fffffffu; }
    void* dpts; void* dout;
    CHECK(hipMalloc(&dpts, h.size() * 4)); CHECK(hipMalloc(&dout, LANES * sizeof(g1x)));
    CHECK(hipMemcpy(dpts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int ITERS = 240;  // madds per lane = ITERS (U divides 240)
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;
        const double ops = (double)blocks * 256 * ITERS;
#define R32(U, C) { float ms = run(k_chain32<U, C>, blocks, (const fp*)dpts, (g1x*)dout, ITERS / U); \
        printf("12x32  copies=%d call=%d wps=%d: %.3f ms  %.2f G madds/s\n", U, C, wps, ms, ops / ms / 1e6); }
#define R29(U, C) { float ms = run(k_chain29<U, C>, blocks, (const fq*)dpts, (fq*)dout, ITERS / U); \
        printf("S29    copies=%d call=%d wps=%d: %.3f ms  %.2f G madds/s\n", U, C, wps, ms, ops / ms / 1e6); }
        R32(1, 0) R32(2, 0) R32(3, 0) R32(4, 0) R32(1, 1) R32(3, 1)
        R29(1, 0) R29(2, 0) R29(3, 0) R29(4, 0) R29(1, 1) R29(3, 1)
    }
    return 0;
}
