// fp381.cuh -- BLS12-381 base-field arithmetic for gfx950 (CDNA4), device only.
//
// Representation: 12 little-endian 32-bit limbs, Montgomery form with R = 2^384,
// always fully reduced to [0, p) so equality is limb equality.
//
// Why this shape on CDNA4 (measured with tools/ubench_valu.hip):
// the only full 32x32->64 multiplier is v_mad_u64_u32 (dst64 = a32*b32 + c64,
// carry-out to an SGPR pair, no carry-in).  The multiply is therefore written
// column-wise (product scanning with the Montgomery reduction interleaved --
// "FIPS"), each column summed in a 96-bit accumulator {hi32 : lo64}:
//     v_mad_u64_u32 lo64, vcc, a_i, b_j, lo64 ; v_addc_co_u32 hi, vcc, 0, hi, vcc
// = 2 VALU instructions per limb product, 288 products + 12 v_mul_lo_u32 per
// Montgomery product and ~80 moves.  A row-wise (CIOS) formulation in plain C
// compiles to the same 289 multiplies but ~1300 extra v_mov/v_lshl_add_u64 to
// marshal 64-bit register pairs.
//
// This is integer VALU work: no MFMA (there is no shared operand to contract
// over; north_star says the same).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace posevo {

struct fp {
    uint32_t l[12];
};

__device__ __forceinline__ constexpr uint32_t fp_p_limb(int j)
{
    return j == 0 ? 0xffffaaabu : j == 1 ? 0xb9feffffu : j == 2 ? 0xb153ffffu : j == 3 ? 0x1eabfffeu
         : j == 4 ? 0xf6b0f624u : j == 5 ? 0x6730d2a0u : j == 6 ? 0xf38512bfu : j == 7 ? 0x64774b84u
         : j == 8 ? 0x434bacd7u : j == 9 ? 0x4b1ba7b6u : j == 10 ? 0x397fe69au : 0x1a0111eau;
}
// R mod p  (Montgomery form of 1)
__device__ __forceinline__ constexpr uint32_t fp_r1_limb(int j)
{
    return j == 0 ? 0x0002fffdu : j == 1 ? 0x76090000u : j == 2 ? 0xc40c0002u : j == 3 ? 0xebf4000bu
         : j == 4 ? 0x53c758bau : j == 5 ? 0x5f489857u : j == 6 ? 0x70525745u : j == 7 ? 0x77ce5853u
         : j == 8 ? 0xa256ec6du : j == 9 ? 0x5c071a97u : j == 10 ? 0xfa80e493u : 0x15f65ec3u;
}
// R^2 mod p
__device__ __forceinline__ constexpr uint32_t fp_r2_limb(int j)
{
    return j == 0 ? 0x1c341746u : j == 1 ? 0xf4df1f34u : j == 2 ? 0x09d104f1u : j == 3 ? 0x0a76e6a6u
         : j == 4 ? 0x4c95b6d5u : j == 5 ? 0x8de5476cu : j == 6 ? 0x939d83c0u : j == 7 ? 0x67eb88a9u
         : j == 8 ? 0xb519952du : j == 9 ? 0x9a793e85u : j == 10 ? 0x92cae3aau : 0x11988fe5u;
}
constexpr uint32_t FP_N0 = 0xfffcfffdu;  // -p^-1 mod 2^32

__device__ __forceinline__ void fp_set_zero(fp& r)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = 0;
}
__device__ __forceinline__ void fp_set_one(fp& r)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = fp_r1_limb(j);
}
__device__ __forceinline__ void fp_set_r2(fp& r)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = fp_r2_limb(j);
}
__device__ __forceinline__ bool fp_is_zero(const fp& a)
{
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) o |= a.l[j];
    return o == 0;
}
__device__ __forceinline__ bool fp_eq(const fp& a, const fp& b)
{
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) o |= a.l[j] ^ b.l[j];
    return o == 0;
}

// r = a + b mod p     (a, b < p < 2^381: the 384-bit sum cannot carry out)
__device__ __forceinline__ void fp_add(fp& r, const fp& a, const fp& b)
{
    uint32_t t[12], s[12], c = 0, br = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) t[j] = __builtin_addc(a.l[j], b.l[j], c, &c);
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = __builtin_subc(t[j], fp_p_limb(j), br, &br);
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = br ? t[j] : s[j];
}
// r = a - b mod p
__device__ __forceinline__ void fp_sub(fp& r, const fp& a, const fp& b)
{
    uint32_t t[12], s[12], c = 0, br = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) t[j] = __builtin_subc(a.l[j], b.l[j], br, &br);
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = __builtin_addc(t[j], fp_p_limb(j), c, &c);
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = br ? s[j] : t[j];
}
__device__ __forceinline__ void fp_dbl(fp& r, const fp& a) { fp_add(r, a, a); }
__device__ __forceinline__ void fp_neg(fp& r, const fp& a)
{
    fp z;
    fp_set_zero(z);
    fp_sub(r, z, a);
}

// 96-bit column accumulator {hi : lo64}:  lo64 += a*b ; hi += carry-out
#define POSEVO_MAC(lo64, hi, a, b)                                                                 \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"                   \
        : "+v"(lo64), "+v"(hi)                                                                     \
        : "v"(a), "v"(b)                                                                           \
        : "vcc")

// r = a * b * R^-1 mod p   (product scanning, reduction interleaved)
__device__ __forceinline__ void fp_mul(fp& r, const fp& a, const fp& b)
{
    uint32_t m[12], t[13];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) POSEVO_MAC(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) {
            uint32_t pj = fp_p_limb(k - i);
            POSEVO_MAC(lo, hi, m[i], pj);
        }
        m[k] = (uint32_t)lo * FP_N0;
        {
            uint32_t p0 = fp_p_limb(0);
            POSEVO_MAC(lo, hi, m[k], p0);  // low word becomes 0
        }
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = 12; k < 23; ++k) {
#pragma unroll
        for (int i = k - 11; i < 12; ++i) POSEVO_MAC(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 11; i < 12; ++i) {
            uint32_t pj = fp_p_limb(k - i);
            POSEVO_MAC(lo, hi, m[i], pj);
        }
        t[k - 12] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[11] = (uint32_t)lo;
    t[12] = (uint32_t)(lo >> 32);
    // a, b < p  =>  result < 2p: one conditional subtract
    uint32_t s[12], br = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = __builtin_subc(t[j], fp_p_limb(j), br, &br);
    const bool ge = (t[12] != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = ge ? s[j] : t[j];
}
__device__ __forceinline__ void fp_sqr(fp& r, const fp& a) { fp_mul(r, a, a); }

// Montgomery reduction of a plain 384-bit value: r = a * R^-1 mod p  (= fp_mul(a, 1))
__device__ __forceinline__ void fp_from_mont(fp& r, const fp& a)
{
    fp one;
    fp_set_zero(one);
    one.l[0] = 1;
    fp_mul(r, a, one);
}
__device__ __forceinline__ void fp_to_mont(fp& r, const fp& a)
{
    fp r2;
    fp_set_r2(r2);
    fp_mul(r, a, r2);
}

// 48 big-endian bytes -> limbs (top three bits of byte 0 are serialisation flags: masked)
__device__ __forceinline__ void fp_load_be48(fp& r, const uint8_t* __restrict__ be)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(be);  // 4-byte aligned by layout
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = __builtin_bswap32(w[11 - j]);
    r.l[11] &= 0x1fffffffu;
}
__device__ __forceinline__ void fp_store_be48(uint8_t* __restrict__ be, const fp& a)
{
    uint32_t* w = reinterpret_cast<uint32_t*>(be);
#pragma unroll
    for (int j = 0; j < 12; ++j) w[11 - j] = __builtin_bswap32(a.l[j]);
}

}  // namespace posevo
