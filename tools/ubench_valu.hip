// ubench_valu.hip -- integer/fp64 VALU issue rates on gfx950, to choose the Fp381 limb scheme.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on the GPU box.
// Reports cycles per wave-instruction per SIMD (s_memtime ticks) at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int CHAINS = 8;

#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint64_t* out, uint64_t* cyc, uint32_t seed) { \
    uint32_t a = seed * (threadIdx.x + 1) | 1u, b = a * 2654435761u + 12345u; \
    uint64_t acc[CHAINS]; \
    for (int k = 0; k < CHAINS; ++k) acc[k] = (uint64_t)a * (k + 3) + b; \
    uint64_t t0 = __builtin_readcyclecounter(); \
    for (int it = 0; it < ITERS; ++it) {

#define KERNEL_END \
    } \
    uint64_t t1 = __builtin_readcyclecounter(); \
    uint64_t s = 0; \
    for (int k = 0; k < CHAINS; ++k) s ^= acc[k]; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; \
}

KERNEL_BEGIN(k_mad_u64_u32)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
KERNEL_END

KERNEL_BEGIN(k_mul_lo_u32)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[k] = lo;
    }
KERNEL_END

KERNEL_BEGIN(k_mul_hi_u32)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[k] = lo | 1;
    }
KERNEL_END

KERNEL_BEGIN(k_mad_u32_u24)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));
        acc[k] = lo;
    }
KERNEL_END

KERNEL_BEGIN(k_mul_hi_u32_u24)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[k] = lo | 1;
    }
KERNEL_END

KERNEL_BEGIN(k_add_co_addc)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                     : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[k] = ((uint64_t)hi << 32) | lo;
    }
KERNEL_END

KERNEL_BEGIN(k_lshl_add_u64)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint64_t ab = ((uint64_t)a << 32) | b;
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(ab));
    }
KERNEL_END

KERNEL_BEGIN(k_add_u32)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
        acc[k] = lo;
    }
KERNEL_END

KERNEL_BEGIN(k_fma_f64)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        double d = __longlong_as_double((acc[k] & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
        double x = 1.0000001, y = 1e-9;
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y));
        acc[k] = __double_as_longlong(d);
    }
KERNEL_END

KERNEL_BEGIN(k_lshrrev_b64)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t sh = 1;
        asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(acc[k]) : "v"(sh));
        acc[k] |= 0x8000000000000000ull;
    }
KERNEL_END

KERNEL_BEGIN(k_alignbit)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t lo = (uint32_t)acc[k];
        asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(lo) : "v"(a));
        acc[k] = lo;
    }
KERNEL_END

// mad_u64_u32 followed by an addc into a third word (Comba 3-word column accumulate)
KERNEL_BEGIN(k_mad_plus_addc)
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) {
        uint32_t hi = (uint32_t)(acc[k] >> 40);
        asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                     : "+v"(acc[k]), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[k] ^= hi;
    }
KERNEL_END

typedef void (*kern_t)(uint64_t*, uint64_t*, uint32_t);

static int run(const char* name, kern_t k, int instr_per_chain_step, uint64_t* d_out, uint64_t* d_cyc)
{
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD
        int blocks = 256 * wps;               // 256 threads = 4 waves = one per SIMD of a CU
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 777u);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> cyc(blocks);
        CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : cyc) avg += (double)c; avg /= blocks;
        double n_instr = (double)ITERS * CHAINS * instr_per_chain_step;   // per wave
        // memtime ticks per wave-instruction, with wps waves sharing the SIMD
        printf("%-18s wps=%d  ticks/wave-instr=%.2f  ticks/instr/SIMD=%.2f  wall=%.3f ms  (ns/instr/SIMD=%.3f)\n",
               name, wps, avg / n_instr, avg / n_instr / wps, ms, ms * 1e6 / (n_instr * wps));
    }
    return 0;
}

int main()
{
    uint64_t *d_out, *d_cyc;
    CHECK(hipMalloc(&d_out, 256 * 4 * 256 * sizeof(uint64_t)));
    CHECK(hipMalloc(&d_cyc, 256 * 4 * sizeof(uint64_t)));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run("v_mad_u64_u32", k_mad_u64_u32, 1, d_out, d_cyc);
    run("v_mul_lo_u32", k_mul_lo_u32, 1, d_out, d_cyc);
    run("v_mul_hi_u32", k_mul_hi_u32, 1, d_out, d_cyc);
    run("v_mad_u32_u24", k_mad_u32_u24, 1, d_out, d_cyc);
    run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1, d_out, d_cyc);
    run("add_co+addc (2)", k_add_co_addc, 2, d_out, d_cyc);
    run("v_lshl_add_u64", k_lshl_add_u64, 1, d_out, d_cyc);
    run("v_add_u32", k_add_u32, 1, d_out, d_cyc);
    run("v_fma_f64", k_fma_f64, 1, d_out, d_cyc);
    run("v_lshrrev_b64", k_lshrrev_b64, 1, d_out, d_cyc);
    run("v_alignbit_b32", k_alignbit, 1, d_out, d_cyc);
    run("mad64+addc (2)", k_mad_plus_addc, 2, d_out, d_cyc);
    return 0;
}
