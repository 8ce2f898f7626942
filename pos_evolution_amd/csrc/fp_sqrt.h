// fp_sqrt.h -- square roots in Fp381 (p = 3 mod 4) for the point decompression kernels, device only.
#pragma once
#include "g1.h"
#include "fp381_s29.h"

namespace posevo {

#ifndef POSEVO_POW_INLINE
#define POSEVO_POW_INLINE __noinline__
#endif
static __device__ const uint32_t FP_HALF[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                                0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};     // (p-1)/2

__device__ __forceinline__ bool limbs_gt(const fp& a, const uint32_t* b)  // a > b as 384-bit integers
{
    bool gt = false, eq = true;
#pragma unroll
    for (int j = 11; j >= 0; --j) {
        gt = gt || (eq && a.l[j] > b[j]);
        eq = eq && a.l[j] == b[j];
    }
    return gt;
}
__device__ __forceinline__ bool fp_is_canonical(const fp& x)  // x < p
{
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) (void)__builtin_subc(x.l[j], fp_p_limb(j), br, &br);
    return br != 0;
}
// a (Montgomery form, any representation of a residue) is "larger" than its negation: plain value > (p-1)/2
__device__ __forceinline__ bool fp_is_larger_half(const fp& a_mont)
{
    fp plain;
    fp_from_mont(plain, a_mont);
    return limbs_gt(plain, FP_HALF);
}
// r = a / 2 (works on Montgomery representatives: halving is linear)
__device__ __forceinline__ void fp_half(fp& r, const fp& a)
{
    uint32_t t[13], c = 0;
    const uint32_t odd = a.l[0] & 1u;
#pragma unroll
    for (int j = 0; j < 12; ++j) t[j] = __builtin_addc(a.l[j], odd ? fp_p_limb(j) : 0u, c, &c);
    t[12] = c;
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = (t[j] >> 1) | (t[j + 1] << 31);
}
// w = a^((p-3)/4).  With it: a * w = a^((p+1)/4) is the square root of a when a is a quadratic residue (the caller checks
// its square), and then w = 1 / sqrt(a) as well ((a w) w = a^((p-1)/2) = 1) -- the G2 decompression needs both.  When a is NOT a
// residue, (a w)^2 = -a and (a w) w = -1.
// The chain itself runs in the 29-bit form (fp381_s29.h, fq_pow_pm3d4: why, and the windows); Montgomery words in, canonical
// Montgomery words out, one product each way.
__device__ POSEVO_POW_INLINE void fp_pow_pm3d4(fp& w, const fp& a)
{
    fq x, r;
    fq_from_mont32(x, a.l);
    fq_pow_pm3d4(r, x);
    fq_to_mont32(w.l, r);
}
__device__ __forceinline__ void fp_sqrt_candidate(fp& r, const fp& a)
{
    fp w;
    fp_pow_pm3d4(w, a);
    fp_mul(r, w, a);
}
// true iff a is a square; then r^2 == a
__device__ __forceinline__ bool fp_sqrt(fp& r, const fp& a)
{
    fp t;
    fp_sqrt_candidate(r, a);
    fp_sqr(t, r);
    return fp_eq(t, a);
}

}  // namespace posevo
