// mfmabench.hip -- can the matrix cores take the CONSTANT half of the S29 Montgomery product?  
//
// fq_mul (fp381_s29.h) is 392 v_mad_i64_i32: 196 for a x b and 196 for m x p, p a constant.  As an int8 matrix product the
// constant multiplications become Toeplitz matrices of p's (and of -p^-1's) bytes times the lanes' operand bytes: a batch of 64
// lanes is a 64-column right-hand side, one v_mfma_i32_16x16x64_i8 per 16 rows x 16 lanes x 64 bytes.  What that form needs:
//   * the SEPARATED product (the interleaved form takes each m_k from the column it has just completed -- no batch there):
//     T = a b (196 multiply-adds, 28 live 64-bit columns), m = T_lo (-p^-1) mod 2^406, r = (T + m p) / 2^406;
//   * T_lo as 51 signed bytes per lane (406 bits), transposed into the B-operand layout (a lane holds 16 bytes of ONE
//     column; the column's other bytes sit in the three other 16-lane rows): a 4 x 4 block transpose by
//     v_permlane32_swap / v_permlane16_swap;
//   * 4 x 4 = 16 MFMAs for m (64 rows, lower triangle), their 64 int32 sums per lane transposed back, a carry chain over 51
//     columns, bytes again, transposed again;
//   * 7 x 4 = 28 MFMAs for m p (102 rows: the low half is needed for its carry), 112 int32 sums per lane transposed back, a
//     carry chain over 102 columns, 29-bit limbs again, + T_hi.
// This file times the PIECES with the real instructions on real dependencies (it does not compute a correct product -- the
// question is what the instruction mix costs, and every piece below is a lower bound of its real counterpart):
//   full      the interleaved product as the engine runs it (392 multiply-adds, dependent chain of products)
//   ab        its a x b half alone (196)                          -> full - ab = what the m p half costs today
//   mfma      44 v_mfma_i32_16x16x64_i8 per 64 products, A tiles (the constant matrices, 44 x 1 KB) read from LDS
//   glue      the VALU work between the matrix products: words / signed bytes, 4 transposes (16 + 64 + 16 + 112 swaps), carry
//             chains (51 + 102 columns), limbs
//   path      ab + glue + mfma in one loop (what a product would cost in that form; the matrix pipe may overlap the VALU)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pos_evolution_amd/csrc -o mfmabench mfmabench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "fp381_s29.h"

using namespace posevo;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void seed(fq& a, unsigned salt)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) a.l[i] = (int32_t)((blockIdx.x * 64 + threadIdx.x + salt) * 2654435761u + i * 40503u) >> 4;
}
__device__ __forceinline__ int32_t fold(const fq& a)
{
    int32_t x = 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) x ^= a.l[i];
    return x;
}

// ---- full: the engine's product
__global__ void __launch_bounds__(64) k_full(int32_t* out, int iters)
{
    fq a, b;
    seed(a, 1);
    seed(b, 7);
#pragma nounroll
    for (int k = 0; k < iters; ++k) fq_mul(a, a, b);
    out[blockIdx.x * 64 + threadIdx.x] = fold(a);
}

// ---- ab: the a x b half (28 columns), folded back to 14 limbs without a reduction
__device__ __forceinline__ void mul_ab(int64_t* T, const fq& a, const fq& b)
{
#pragma unroll
    for (int k = 0; k < 2 * FQ_N - 1; ++k) {
        int64_t acc = 0;
#pragma unroll
        for (int i = (k < FQ_N ? 0 : k - (FQ_N - 1)); i <= (k < FQ_N ? k : FQ_N - 1); ++i) acc += (int64_t)a.l[i] * b.l[k - i];
        T[k] = acc;
    }
    T[2 * FQ_N - 1] = 0;
}
// one carry pass over the 28 columns: 28 limbs of 29 bits (what the interleaved product does column by column)
__device__ __forceinline__ void columns_to_limbs(int32_t* l, const int64_t* T)
{
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 2 * FQ_N; ++k) {
        const int64_t v = T[k] + c;
        l[k] = fq_digit(v);
        c = (v - l[k]) >> FQ_B;
    }
}
__global__ void __launch_bounds__(64) k_ab(int32_t* out, int iters)
{
    fq a, b;
    seed(a, 1);
    seed(b, 7);
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        int64_t T[2 * FQ_N];
        mul_ab(T, a, b);
        int32_t l[2 * FQ_N];
        columns_to_limbs(l, T);
#pragma unroll
        for (int i = 0; i < FQ_N; ++i) a.l[i] = fq_digit32(l[i] + l[i + FQ_N]);
    }
    out[blockIdx.x * 64 + threadIdx.x] = fold(a);
}

// ---- the matrix products: n MFMAs over B operands in registers, A tiles from LDS (one ds_read_b128 per row tile)
__device__ __forceinline__ v4i lds_tile(const v4i* lds, int tile) { return lds[tile * 64 + threadIdx.x]; }
template <int ROW_TILES>
__device__ __forceinline__ void matmul(v4i (&acc)[ROW_TILES][4], const v4i (&B)[4], const v4i* lds, int first_tile)
{
#pragma unroll
    for (int r = 0; r < ROW_TILES; ++r) {
        const v4i A = lds_tile(lds, first_tile + r);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[c], (v4i){0, 0, 0, 0}, 0, 0, 0);
    }
}
__global__ void __launch_bounds__(64) k_mfma(int32_t* out, int iters)
{
    __shared__ v4i lds[11 * 64];
    for (int t = 0; t < 11; ++t) lds[t * 64 + threadIdx.x] = (v4i){(int)threadIdx.x * 0x01010101, t * 0x11, 3, 4};
    __syncthreads();
    v4i B[4];
    for (int c = 0; c < 4; ++c) B[c] = (v4i){(int)threadIdx.x, c, 5, 6};
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        v4i m[4][4], r[7][4];
        matmul<4>(m, B, lds, 0);
        v4i B2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) B2[c] = m[0][c] ^ m[1][c] ^ m[2][c] ^ m[3][c];  // (the second product depends on the first)
        matmul<7>(r, B2, lds, 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) B[c] = r[0][c] ^ r[1][c] ^ r[2][c] ^ r[3][c] ^ r[4][c] ^ r[5][c] ^ r[6][c];
    }
    out[blockIdx.x * 64 + threadIdx.x] = B[0].x ^ B[1].y ^ B[2].z ^ B[3].w;
}

// ---- the glue
// 4 x 4 block transpose of N dwords per lane between the wave's four 16-lane rows: two stages of swaps, N / 2 each
template <int N> __device__ __forceinline__ void transpose4(uint32_t (&v)[N])
{
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        auto s = __builtin_amdgcn_permlane32_swap(v[i], v[i + 1], false, false);
        v[i] = s[0];
        v[i + 1] = s[1];
    }
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        auto s = __builtin_amdgcn_permlane16_swap(v[i], v[i + 1], false, false);
        v[i] = s[0];
        v[i + 1] = s[1];
    }
}
// 14 limbs of 29 bits -> 13 words + signed byte digits (a borrow chain through the words) -> 16 dwords of B operand
__device__ __forceinline__ void limbs_to_bytes(uint32_t (&w)[16], const int32_t* l)
{
#pragma unroll
    for (int j = 0; j < 13; ++j) {
        const int bit = 32 * j, i = bit / 29, s = bit % 29;
        uint32_t x = (uint32_t)l[i] >> s;
        if (i + 1 < FQ_N) x |= (uint32_t)l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < FQ_N) x |= (uint32_t)l[i + 2] << (58 - s);
        w[j] = x;
    }
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 13; ++j) {  // byte d -> d - 256 [d >= 128], the borrow into the next byte; word-parallel
        const uint64_t t = (uint64_t)w[j] + 0x80808080u + carry;
        carry = (uint32_t)(t >> 32);
        w[j] = (uint32_t)t ^ 0x80808080u;
    }
    w[13] = carry;
    w[14] = w[15] = 0;
}
// column sums -> bytes: a carry chain, the digits packed four to a word
template <int COLS, int WORDS> __device__ __forceinline__ void carry_chain(uint32_t (&w)[WORDS], const uint32_t* col)
{
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < WORDS; ++j) w[j] = 0;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        c += (int32_t)col[j];
        const uint32_t d = (uint32_t)c & 0xffu;
        c >>= 8;
        w[j >> 2] |= d << (8 * (j & 3));
    }
    w[WORDS - 1] ^= (uint32_t)c;
}
__device__ __forceinline__ void glue_in(uint32_t (&B)[16], const int32_t* limbs)
{
    limbs_to_bytes(B, limbs);
    transpose4<16>(B);
}
__device__ __forceinline__ void glue_mid(uint32_t (&B2)[16], uint32_t (&sums)[64])
{
    transpose4<64>(sums);
    carry_chain<51, 16>(B2, sums);
    transpose4<16>(B2);
}
__device__ __forceinline__ void glue_out(int32_t* limbs, uint32_t (&sums)[112], const int64_t* T_hi)
{
    transpose4<112>(sums);
    uint32_t w[28];
    carry_chain<102, 28>(w, sums);
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) {  // bytes 51.. -> 29-bit limbs, + T_hi
        const int bit = 408 + 29 * i, j = bit >> 5, s = bit & 31;
        uint64_t x = ((uint64_t)w[j] | ((uint64_t)w[j + 1] << 32)) >> s;
        limbs[i] = fq_digit((int64_t)(x & (uint64_t)FQ_MASK) + T_hi[i]);
    }
}
__global__ void __launch_bounds__(64) k_glue(int32_t* out, int iters)
{
    fq a;
    seed(a, 3);
    int64_t T_hi[FQ_N];
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) T_hi[i] = a.l[i] * 3;
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        uint32_t B[16], sums[64], B2[16], sums2[112];
        glue_in(B, a.l);
#pragma unroll
        for (int j = 0; j < 64; ++j) sums[j] = B[j & 15] * (j + 1);   // stands for the 16 MFMAs' outputs
        glue_mid(B2, sums);
#pragma unroll
        for (int j = 0; j < 112; ++j) sums2[j] = B2[j & 15] + j;     // ... the 28 MFMAs' outputs
        glue_out(a.l, sums2, T_hi);
    }
    out[blockIdx.x * 64 + threadIdx.x] = fold(a);
}

// ---- path: a x b + glue + matrix products
__global__ void __launch_bounds__(64) k_path(int32_t* out, int iters)
{
    __shared__ v4i lds[11 * 64];
    for (int t = 0; t < 11; ++t) lds[t * 64 + threadIdx.x] = (v4i){(int)threadIdx.x * 0x01010101, t * 0x11, 3, 4};
    __syncthreads();
    fq a, b;
    seed(a, 1);
    seed(b, 7);
#pragma nounroll
    for (int k = 0; k < iters; ++k) {
        int64_t T[2 * FQ_N];
        mul_ab(T, a, b);
        int32_t l[2 * FQ_N];
        columns_to_limbs(l, T);
        int64_t T_hi[FQ_N];
#pragma unroll
        for (int i = 0; i < FQ_N; ++i) T_hi[i] = l[FQ_N + i];
        uint32_t Bw[16], B2w[16];
        glue_in(Bw, l);
        v4i B[4], m[4][4], r[7][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) B[c] = (v4i){(int)Bw[4 * c], (int)Bw[4 * c + 1], (int)Bw[4 * c + 2], (int)Bw[4 * c + 3]};
        matmul<4>(m, B, lds, 0);
        uint32_t sums[64];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sums[16 * rr + 4 * c] = m[rr][c].x; sums[16 * rr + 4 * c + 1] = m[rr][c].y;
                sums[16 * rr + 4 * c + 2] = m[rr][c].z; sums[16 * rr + 4 * c + 3] = m[rr][c].w;
            }
        glue_mid(B2w, sums);
#pragma unroll
        for (int c = 0; c < 4; ++c) B[c] = (v4i){(int)B2w[4 * c], (int)B2w[4 * c + 1], (int)B2w[4 * c + 2], (int)B2w[4 * c + 3]};
        matmul<7>(r, B, lds, 4);
        uint32_t sums2[112];
#pragma unroll
        for (int rr = 0; rr < 7; ++rr)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sums2[16 * rr + 4 * c] = r[rr][c].x; sums2[16 * rr + 4 * c + 1] = r[rr][c].y;
                sums2[16 * rr + 4 * c + 2] = r[rr][c].z; sums2[16 * rr + 4 * c + 3] = r[rr][c].w;
            }
        glue_out(a.l, sums2, T_hi);
    }
    out[blockIdx.x * 64 + threadIdx.x] = fold(a);
}

template <class K> static float time_kernel(K kernel, int waves, int32_t* d, int iters, hipEvent_t e0, hipEvent_t e1)
{
    hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d, 50);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e6f / iters;  // ns per iteration (= per product of every lane of a wave)
}

int main()
{
    int32_t* d;
    CHECK(hipMalloc(&d, 4 * 64 * 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("ns per product and wave (64 products), dependent chain, best of 3 launches of %d iterations\n", iters);
    printf("%-22s %10s %10s %10s %10s %10s %14s\n", "waves (per SIMD)", "full", "ab", "mfma x44", "glue", "path", "path / full");
    for (int waves : {1024, 2048, 4096}) {
        const float full = time_kernel(k_full, waves, d, iters, e0, e1);
        const float ab = time_kernel(k_ab, waves, d, iters, e0, e1);
        const float mf = time_kernel(k_mfma, waves, d, iters, e0, e1);
        const float gl = time_kernel(k_glue, waves, d, iters, e0, e1);
        const float pa = time_kernel(k_path, waves, d, iters, e0, e1);
        const float f = (float)(waves / 1024);
        printf("%4d (%d)              %10.1f %10.1f %10.1f %10.1f %10.1f %14.2f\n", waves, waves / 1024, full / f, ab / f,
               mf / f, gl / f, pa / f, pa / full);
    }
    printf("(columns at 2 / 4 waves per SIMD are divided by the waves per SIMD: ns of SIMD time per product-wave)\n");
    printf("the m p half as multiply-adds costs full - ab; in the matrix form it costs path - ab\n");
    CHECK(hipGetLastError());
    return 0;
}
