// fp_inv_safegcd.h -- modular inversion in Fp381 by Bernstein-Yang "safegcd" divsteps.
//
// Why: normalising a projective (XYZZ) sum to affine needs one inversion per output.  Fermat
// (a^(p-2)) is ~570 DEPENDENT Montgomery products = ~1 ms of pure latency per lane on
// gfx950 (tools/fpbench: 1.8 us per dependent fp_mul) -- it was the single largest item
// of the first profile (profiles/r01_kernel_stats_first.txt: k_g1_finish 1012 us).
// divsteps work on the low bits only: 30 of them are folded into one 2x2 transition
// matrix computed on 32-bit scalars, then applied once to the 13 x 30-bit signed limbs
// of f, g (exact division by 2^30) and of d, e (division mod p).  ~40x fewer instructions.
//
// Algorithm and limb discipline follow the published construction (Bernstein, Yang:
// "Fast constant-time gcd computation and modular inversion", 2019; the 30-bit batching
// is the one popularised by libsecp256k1's modinv32, restated here for a 381-bit modulus).
// The loop runs until g == 0 (at most 1101 divsteps are ever needed for 381-bit inputs,
// Theorem 11.2 of the paper: (49*381+57)/17; 37 batches of 30 cover that).
//
// Plain integer code, no inline asm: compiles for host too, so tests/ checks it against
// Python's pow(x, -1, p) on the CPU (tests/test_host_safegcd.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define POSEVO_HD __host__ __device__ __forceinline__
#else
#define POSEVO_HD static inline
#endif

namespace posevo {

constexpr int SG_LIMBS = 13;           // 13 x 30 = 390 bits
constexpr int32_t SG_M30 = (1 << 30) - 1;

// p in 30-bit limbs and p^-1 mod 2^30
POSEVO_HD constexpr int32_t sg_p_limb(int i)
{
    return i == 0 ? 0x3fffaaab : i == 1 ? 0x27fbffff : i == 2 ? 0x153ffffb : i == 3 ? 0x2affffac : i == 4 ? 0x30f6241e : i == 5 ? 0x34a83da : i == 6 ? 0x112bf673 : i == 7 ? 0x12e13ce1 : i == 8 ? 0x2cd76477 : i == 9 ? 0x1ed90d2e : i == 10 ? 0x29a4b1ba : i == 11 ? 0x3a8e5ff9
         : 0x1a0111;
}
constexpr uint32_t SG_PINV30 = 0x30003u;  // p^-1 mod 2^30

struct sg_int {
    int32_t v[SG_LIMBS];  // limbs 0..11 in [0, 2^30), limb 12 signed
};

// 12 x u32 (little-endian, value < 2^384) -> 13 x 30-bit limbs
POSEVO_HD void sg_from_u32(sg_int& r, const uint32_t* a)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < SG_LIMBS; ++i) {
        const int bit = 30 * i;
        const int w = bit >> 5, s = bit & 31;
        uint64_t lo = a[w];
        uint64_t hi = (w + 1 < 12) ? a[w + 1] : 0;
        uint64_t x = (lo | (hi << 32)) >> s;
        r.v[i] = (int32_t)(x & SG_M30);
    }
    // top limb: bits 360..383 (24 bits) -- the generic expression above already yields them
}
POSEVO_HD void sg_to_u32(uint32_t* a, const sg_int& r)
{
    // r is non-negative and < 2^384 here
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int w = 0; w < 12; ++w) {
        const int bit = 32 * w;
        const int i = bit / 30, s = bit % 30;
        uint64_t x = (uint64_t)(uint32_t)r.v[i] >> s;
        int have = 30 - s;
        if (i + 1 < SG_LIMBS) { x |= (uint64_t)(uint32_t)r.v[i + 1] << have; have += 30; }
        if (have < 32 && i + 2 < SG_LIMBS) x |= (uint64_t)(uint32_t)r.v[i + 2] << have;
        a[w] = (uint32_t)x;
    }
}

struct sg_trans {
    int32_t u, v, q, r;
};

// 30 divsteps on the low bits.  `delta` holds TWICE the paper's delta and starts at 1, i.e. delta = 1/2: the
// "half-delta" variant (the starting point libsecp256k1's modinv uses).  Measured on 20 000 random 380-bit values:
// 26.0 batches of 30 on average, 27 at most, against 26.9 / 28 with delta = 1 -- a 3 % shorter chain, not the sixth
// its worst-case bound suggests; kept because it is free.  The 40-batch cap of sg_modinv covers either variant's
// worst case (1101 divsteps = 37 batches for delta = 1, Theorem 11.2; fewer for 1/2).  The state stays odd:
// "delta > 0" is the test, "+ 2" the step.  f0 must be odd.
POSEVO_HD int32_t sg_divsteps_30(int32_t delta, uint32_t f0, uint32_t g0, sg_trans& t)
{
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
    for (int i = 0; i < 30; ++i) {
        // c1 = all-ones iff (delta > 0 and g odd); c2 = all-ones iff g odd
        const uint32_t c2 = (uint32_t)0 - (g & 1u);
        const uint32_t c1 = c2 & (uint32_t)((int32_t)(-delta) >> 31);
        // swap+negate branch: (f, g) <- (g, g - f), (u,v,q,r) <- (q, r, q - u, r - v), delta <- -delta
        // otherwise:          g <- g + (g odd ? f : 0), (q, r) <- (q + u, r + v) if g odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // conditionally negated f,u,v
        g += x & c2;
        q += y & c2;
        r += z & c2;
        // if c1: f <- old g, u <- old q, v <- old r.  Since g_new = g_old - f_old etc.: f_new = g_new + f_old
        f += g & c1;
        u += q & c1;
        v += r & c1;
        delta = (int32_t)(((uint32_t)delta ^ c1) - c1);  // negate if c1
        delta += 2;                                       // (2 delta) <- 2 (1 +- delta)
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return delta;
}

// A variable-time form (count-trailing-zeros runs + six bits cancelled per pass) was measured and dropped: inside a
// 64-lane wave the data-dependent trip counts and the swap branch diverge, and k_g1_finish came out 3 % slower than
// with the branch-free loop above (0.140 vs 0.1355 ms, same box).

// (f, g) <- (u f + v g, q f + r g) / 2^30   (exact)
POSEVO_HD void sg_update_fg(sg_int& f, sg_int& g, const sg_trans& t)
{
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 1; i < SG_LIMBS; ++i) {
        cf += u * f.v[i] + v * g.v[i];
        cg += q * f.v[i] + r * g.v[i];
        f.v[i - 1] = (int32_t)cf & SG_M30;
        g.v[i - 1] = (int32_t)cg & SG_M30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[SG_LIMBS - 1] = (int32_t)cf;
    g.v[SG_LIMBS - 1] = (int32_t)cg;
}

// (d, e) <- (u d + v e, q d + r e) / 2^30 mod p, keeping d, e in (-2p, p)
POSEVO_HD void sg_update_de(sg_int& d, sg_int& e, const sg_trans& t, uint32_t pinv30)
{
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[SG_LIMBS - 1] >> 31, se = e.v[SG_LIMBS - 1] >> 31;  // sign masks
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = u * d.v[0] + v * e.v[0];
    int64_t ce = q * d.v[0] + r * e.v[0];
    // choose md, me so that the low 30 bits of cd + p0*md (ce + p0*me) vanish
    md -= (int32_t)((pinv30 * (uint32_t)cd + (uint32_t)md) & SG_M30);
    me -= (int32_t)((pinv30 * (uint32_t)ce + (uint32_t)me) & SG_M30);
    cd += (int64_t)sg_p_limb(0) * md;
    ce += (int64_t)sg_p_limb(0) * me;
    cd >>= 30;
    ce >>= 30;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 1; i < SG_LIMBS; ++i) {
        cd += u * d.v[i] + v * e.v[i] + (int64_t)sg_p_limb(i) * md;
        ce += q * d.v[i] + r * e.v[i] + (int64_t)sg_p_limb(i) * me;
        d.v[i - 1] = (int32_t)cd & SG_M30;
        e.v[i - 1] = (int32_t)ce & SG_M30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[SG_LIMBS - 1] = (int32_t)cd;
    e.v[SG_LIMBS - 1] = (int32_t)ce;
}

// r in (-2p, p) -> [0, p), negated first when sign < 0
POSEVO_HD void sg_normalize(sg_int& r, int32_t sign)
{
    // add p if negative
    int32_t cond_add = r.v[SG_LIMBS - 1] >> 31;
    const int32_t cond_negate = sign >> 31;
    int32_t c = 0;
    for (int i = 0; i < SG_LIMBS; ++i) {
        int32_t x = r.v[i] + (sg_p_limb(i) & cond_add);
        x = (x ^ cond_negate) - cond_negate;
        x += c;
        c = x >> 30;
        r.v[i] = (i < SG_LIMBS - 1) ? (x & SG_M30) : x;
    }
    // result in (-p, p): add p once more if negative
    cond_add = r.v[SG_LIMBS - 1] >> 31;
    c = 0;
    for (int i = 0; i < SG_LIMBS; ++i) {
        int32_t x = r.v[i] + (sg_p_limb(i) & cond_add) + c;
        c = x >> 30;
        r.v[i] = (i < SG_LIMBS - 1) ? (x & SG_M30) : x;
    }
}

POSEVO_HD bool sg_is_zero(const sg_int& a)
{
    int32_t o = 0;
    for (int i = 0; i < SG_LIMBS; ++i) o |= a.v[i];
    return o == 0;
}

// out = x^-1 mod p for a plain (non-Montgomery) 0 < x < p given as 12 little-endian u32.
// Returns the number of 30-divstep batches used.  x == 0 yields 0.
POSEVO_HD int sg_modinv(uint32_t* out, const uint32_t* x, uint32_t pinv30)
{
    sg_int f, g, d, e;
    for (int i = 0; i < SG_LIMBS; ++i) {
        f.v[i] = sg_p_limb(i);
        d.v[i] = 0;
        e.v[i] = 0;
    }
    e.v[0] = 1;
    sg_from_u32(g, x);
    int32_t delta = 1;
    int batches = 0;
    for (; batches < 40; ++batches) {
        if (sg_is_zero(g)) break;
        sg_trans t;
        delta = sg_divsteps_30(delta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30),
                               (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
        sg_update_de(d, e, t, pinv30);
        sg_update_fg(f, g, t);
    }
    // f = +-1 (gcd); d * x == f (mod p)
    sg_normalize(d, f.v[SG_LIMBS - 1]);
    sg_to_u32(out, d);
    return batches;
}

}  // namespace posevo
