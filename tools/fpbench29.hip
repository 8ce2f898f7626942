// fpbench29.hip -- the S29 field form (pos_evolution_amd/csrc/fp381_s29.h, g1_s29.h) on the device: results against
// the SAME source run on the host (which tests/test_host_fp29.py holds against Python integers), and throughput of the
// product and of the mixed add beside the 12 x 32 form's (tools/fpbench.hip prints those).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o fpbench29 fpbench29.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "g1_s29.h"

using namespace posevo;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mul_check(const fq* a, const fq* b, fq* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fq r, s, d, t;
    fq_mul(r, a[i], b[i]);
    fq_sqr(s, a[i]);
    fq_sub(d, r, s);       // a b - a^2, unreduced difference of two products
    fq_mul(t, d, b[i]);    // ... straight into the next product
    fq_canonical(out[i], t);
}
__global__ void k_run_check(const uint32_t* rows24, int run, uint32_t* out48, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1q acc;
    g1q_set_inf(acc);
    for (int j = 0; j < run; ++j) {
        const uint32_t* row = rows24 + 24 * ((size_t)i * run + j);
        uint32_t any = 0;
        for (int k = 0; k < 24; ++k) any |= row[k];
        fq qx, qy;
        fq_from_mont32(qx, row);
        fq_from_mont32(qy, row + 12);
        g1q_add_affine(acc, qx, qy, any == 0);
    }
    g1q_to_words32(out48 + 48 * (size_t)i, acc);
}
__global__ void __launch_bounds__(256) k_mul_chain(fq* x, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    fq a = x[i], b = x[i + 1];
    for (int k = 0; k < iters; ++k) { fq c; fq_mul(c, a, b); a = b; b = c; }
    x[i] = b;
}
__global__ void __launch_bounds__(256) k_madd_chain(const fq* pts, fq* out, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    g1q acc;
    g1q_set_inf(acc);
    fq qx = pts[2 * i], qy = pts[2 * i + 1];
    g1q_add_affine(acc, qx, qy, false);
    acc.affine = false;                      // time the general body
    qx = pts[2 * i + 2]; qy = pts[2 * i + 3];
    for (int k = 0; k < iters; ++k) { g1q_add_affine(acc, qx, qy, false); fq_sub_norm(qx, qx, acc.zz); }
    out[i] = acc.x;
}

int main()
{
    const int N = 1 << 15;
    std::mt19937_64 rng(29);
    auto rnd = [&](fq& o) {  // canonical limbs of a value below 2^380
        for (int k = 0; k < FQ_N - 1; ++k) o.l[k] = (int32_t)(rng() & FQ_MASK);
        o.l[FQ_N - 1] = (int32_t)(rng() & 7);
    };
    std::vector<fq> ha(N), hb(N), hr(N), he(N);
    for (int i = 0; i < N; ++i) { rnd(ha[i]); rnd(hb[i]); }
    fq *da, *db, *dr;
    CHECK(hipMalloc(&da, N * sizeof(fq))); CHECK(hipMalloc(&db, N * sizeof(fq))); CHECK(hipMalloc(&dr, (N + 8) * sizeof(fq)));
    CHECK(hipMemcpy(da, ha.data(), N * sizeof(fq), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), N * sizeof(fq), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul_check, dim3(N / 256), dim3(256), 0, 0, da, db, dr, N);
    CHECK(hipMemcpy(hr.data(), dr, N * sizeof(fq), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        fq r, s, d, t;
        fq_mul(r, ha[i], hb[i]); fq_sqr(s, ha[i]); fq_sub(d, r, s); fq_mul(t, d, hb[i]); fq_canonical(he[i], t);
        if (memcmp(&he[i], &hr[i], sizeof(fq))) ++bad;
    }
    printf("S29 mul / sqr / sub / canonical, device vs host: %d / %d mismatches\n", bad, N);

    // accumulation runs: rows = Montgomery (R = 2^384) words of arbitrary field elements are NOT curve points; the group
    // law's formulas do not care, and host and device must agree word for word (the CPU test uses real points)
    const int RUNS = 4096, RUN = 8;
    std::vector<uint32_t> rows((size_t)RUNS * RUN * 24), o_dev((size_t)RUNS * 48), o_host((size_t)RUNS * 48);
    for (auto& w : rows) w = (uint32_t)rng();
    for (size_t r = 0; r < (size_t)RUNS * RUN; ++r) { rows[24 * r + 11] &= 0x0fffffffu; rows[24 * r + 23] &= 0x0fffffffu; }
    for (int r = 0; r < RUNS; r += 7) memset(&rows[24 * ((size_t)r * RUN + 3)], 0, 96);            // a row without a point
    for (int r = 1; r < RUNS; r += 5) memcpy(&rows[24 * ((size_t)r * RUN + 1)], &rows[24 * ((size_t)r * RUN)], 96);  // P, P
    uint32_t *drows, *dout;
    CHECK(hipMalloc(&drows, rows.size() * 4)); CHECK(hipMalloc(&dout, o_dev.size() * 4));
    CHECK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_run_check, dim3(RUNS / 256), dim3(256), 0, 0, drows, RUN, dout, RUNS);
    CHECK(hipMemcpy(o_dev.data(), dout, o_dev.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < RUNS; ++i) {
        g1q acc;
        g1q_set_inf(acc);
        for (int j = 0; j < RUN; ++j) {
            const uint32_t* row = &rows[24 * ((size_t)i * RUN + j)];
            uint32_t any = 0;
            for (int k = 0; k < 24; ++k) any |= row[k];
            fq qx, qy;
            fq_from_mont32(qx, row); fq_from_mont32(qy, row + 12);
            g1q_add_affine(acc, qx, qy, any == 0);
        }
        g1q_to_words32(&o_host[48 * (size_t)i], acc);
    }
    bad = 0;
    for (int i = 0; i < RUNS; ++i) if (memcmp(&o_host[48 * (size_t)i], &o_dev[48 * (size_t)i], 192)) ++bad;
    printf("S29 accumulation runs of %d, device vs host: %d / %d mismatches\n", RUN, bad, RUNS);

    fq* dx; CHECK(hipMalloc(&dx, ((size_t)256 * 4 * 256 * 2 + 8) * sizeof(fq)));
    for (size_t off = 0; off < (size_t)256 * 4 * 256 * 2; off += N) CHECK(hipMemcpy(dx + off, da, (size_t)N * sizeof(fq), hipMemcpyDeviceToDevice));
    fq* dj; CHECK(hipMalloc(&dj, (size_t)256 * 4 * 256 * sizeof(fq)));
    for (int which = 0; which < 2; ++which)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = 256 * wps, iters = which ? 200 : 2000;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k_mul_chain, dim3(blocks), dim3(256), 0, 0, dx, iters);
                else hipLaunchKernelGGL(k_madd_chain, dim3(blocks), dim3(256), 0, 0, dx, dj, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (double)blocks * 256 * iters;
            printf("%-10s waves/SIMD=%d: %.3f ms, %.2f G %s/s on the chip\n", which ? "s29 madd" : "s29 mul", wps, ms,
                   ops / ms / 1e6, which ? "mixed adds" : "products");
        }
    printf("compare: tools/fpbench (12 x 32 form) -- 57 G products/s, 4.6 G mixed adds/s at two waves per SIMD\n");
    return 0;
}
