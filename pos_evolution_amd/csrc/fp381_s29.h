// fp381_s29.h -- BLS12-381 base field in 14 signed limbs of 29 bits ("S29"), Montgomery constant R' = 2^406.
//
// Why this form next to fp381.h (12 x 32 bits, R = 2^384): half of that product's 689 instructions are carry
// handling -- v_mad_u64_u32 has a carry-out but no carry-in, a full 32 x 32 product plus a 64-bit accumulator overflows,
// every limb product pays a v_addc_co_u32 (288 per product), plus ~65 moves that slide the 96-bit column accumulator and
// the conditional subtraction.  With 29-bit limbs a column of the interleaved (FIPS) Montgomery product -- up to 14
// a_i b_j + 14 m_i n_j terms of <= 2^58 -- fits a signed 64-bit accumulator (28 x 2^58 < 2^63): one v_mad_i64_i32 per
// limb product and nothing else, ~505 instructions per product instead of 689, written in plain C++ (no inline assembly:
// the compiler sees the whole product and the same source compiles for the host, where tests/test_host_fp29.py holds it
// against Python integers).
// The price is 392 multiply-adds per product instead of 288.  Round 3 wrote this form and could not run it; by the
// per-instruction timings of round 1 it predicted a wash (the carry add behind a multiply-add "three quarters hidden").
// Round 4 measured (tools/fpbench29, tools/icbench, tools/accbench): 14 % more dependent products per second,
// 25 % more mixed adds (the squaring is 301 multiply-adds), and the accumulation kernel 158 us against 183 for a million
// points -- a mixed add of 3 738 multiply-adds runs at the multiplier's issue rate, the simple instructions around them are
// free and carries are not.  It is the accumulation's form and, since round 6, the tree's; finish and the wire formats stay in fp381.h.
//
// Lazy, signed values.  R' / p > 2^25, so a product of operands of magnitude < 2^386 (32 p) comes out in (-eps, p + eps)
// with no final subtraction; a - b is a plain limb-wise subtraction (limbs of both signs are fine in the next product as
// long as |limb| <= 2^29 + small; products and carry passes leave BALANCED digits, |limb| <= 2^28 + small, so one
// subtraction stays inside); sums of three terms pass through ONE carry pass (fq_norm: four instructions per limb, no
// dependency chain).  A value is a residue mod p in redundant form: equality with zero is a filter on the low 29 bits
// against the few multiples of p the value can be, and an exact comparison behind it (fq_is_zero_modp).
//
// Replaces (as fp381.h does): the field arithmetic under bls.Aggregate's point additions, reference call sites pe:736,
// pe:976 (the reference holds no BLS arithmetic; oracle/g1.py restates it).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PE_HD __host__ __device__ __forceinline__
#define PE_HD_MEMBER static __host__ __device__ __forceinline__
#else
#define PE_HD static inline
#define PE_HD_MEMBER static inline
#endif
#define PE_HD_CONST static constexpr  // constant-initialised: hipcc emits them for the device where device code reads them

namespace posevo {

constexpr int FQ_B = 29, FQ_N = 14;
constexpr int32_t FQ_MASK = (1 << FQ_B) - 1;

#include "fp381_s29_consts.inc"

struct fq {
    int32_t l[FQ_N];  // value = sum l[i] 2^(29 i); limbs 0..12 nominally balanced digits in [-2^28, 2^28), the top limb the rest
};

PE_HD void fq_set_zero(fq& r)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) r.l[i] = 0;
}
PE_HD void fq_set_one(fq& r)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) r.l[i] = FQ_ONE[i];
}
PE_HD bool fq_limbs_zero(const fq& a)  // all limbs zero (the table's encoding of "no point"); NOT a test mod p
{
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) o |= a.l[i];
    return o == 0;
}

// r = a - b, limb by limb.  |limbs| add up: two balanced operands (|limb| <= 2^28 + c) or two table rows (canonical limbs
// in [0, 2^29)) give |limb| <= 2^29 + 2c -- what the products accept.
PE_HD void fq_sub(fq& r, const fq& a, const fq& b)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) r.l[i] = a.l[i] - b.l[i];
}
// The balanced digit of a value: its low 29 bits read as a signed number in [-2^28, 2^28).  Balanced on purpose: a limb the
// compiler can prove non-negative turns the next product's sign extension into a zero extension, and a signed x unsigned
// 32 x 32 -> 64 multiply is TWO v_mad_u64_u32 plus fix-ups on gfx950 where signed x signed is one v_mad_i64_i32.
PE_HD int32_t fq_digit(int64_t v) { return (int32_t)((uint32_t)v << (32 - FQ_B)) >> (32 - FQ_B); }
PE_HD int32_t fq_digit32(int32_t v) { return (int32_t)((uint32_t)v << (32 - FQ_B)) >> (32 - FQ_B); }
// One carry pass: limbs 0..12 back to balanced digits plus the neighbour's carry (|carry| <= 4 for inputs below 2^31 in
// magnitude), the top limb absorbs its carry-in.  No chain: every limb looks at its lower neighbour only.
PE_HD void fq_norm(fq& r, const fq& a)
{
    int32_t c[FQ_N], o[FQ_N];
#pragma unroll
    for (int i = 0; i < FQ_N - 1; ++i) {
        o[i] = fq_digit32(a.l[i]);
        c[i] = (a.l[i] - o[i]) >> FQ_B;  // exact
    }
#pragma unroll
    for (int i = 1; i < FQ_N - 1; ++i) o[i] += c[i - 1];
    o[FQ_N - 1] = a.l[FQ_N - 1] + c[FQ_N - 2];
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) r.l[i] = o[i];
}
// r = a + b (one carry pass), r = 2a, r = a - b - c and r = a - b - 2c (one carry pass each): the shapes the XYZZ
// formulas need.
PE_HD void fq_add(fq& r, const fq& a, const fq& b)
{
    fq t;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) t.l[i] = a.l[i] + b.l[i];
    fq_norm(r, t);
}
PE_HD void fq_sub_norm(fq& r, const fq& a, const fq& b)
{
    fq t;
    fq_sub(t, a, b);
    fq_norm(r, t);
}
PE_HD void fq_sub_sub2_norm(fq& r, const fq& a, const fq& b, const fq& c)  // a - b - 2c
{
    fq t;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) t.l[i] = a.l[i] - b.l[i] - 2 * c.l[i];
    fq_norm(r, t);
}

// r = a b / R' mod p in lazy form: (a b + m p) / 2^406 with m in [0, 2^406), i.e. r in (a b / R', a b / R' + p).
// Operand limbs |.| <= 2^29 + 16.  Output limbs 0..12 balanced digits in [-2^28, 2^28), top limb small.
PE_HD void fq_mul(fq& r, const fq& a, const fq& b)
{
    int32_t m[FQ_N];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * FQ_P[k - i];
        m[k] = (int32_t)(((uint32_t)acc * FQ_N0INV) & (uint32_t)FQ_MASK);
        acc += (int64_t)m[k] * FQ_P[0];
        acc >>= FQ_B;  // exact: the low 29 bits are zero now
    }
#pragma unroll
    for (int k = FQ_N; k < 2 * FQ_N - 1; ++k) {
#pragma unroll
        for (int i = k - (FQ_N - 1); i < FQ_N; ++i) acc += (int64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - (FQ_N - 1); i < FQ_N; ++i) acc += (int64_t)m[i] * FQ_P[k - i];
        r.l[k - FQ_N] = fq_digit(acc);
        acc = (acc + (int64_t(1) << (FQ_B - 1))) >> FQ_B;  // = (acc - digit) / 2^29: round to nearest
    }
    r.l[FQ_N - 1] = (int32_t)acc;
}
// r = a^2 / R': the cross products once, against the doubled operand (|2 a_i a_j| <= 2^59: a column of 7 of them, one
// square and 14 m n terms stays below 2^63).
PE_HD void fq_sqr(fq& r, const fq& a)
{
    int32_t m[FQ_N], d[FQ_N];
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) d[i] = 2 * a.l[i];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (int64_t)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * FQ_P[k - i];
        m[k] = (int32_t)(((uint32_t)acc * FQ_N0INV) & (uint32_t)FQ_MASK);
        acc += (int64_t)m[k] * FQ_P[0];
        acc >>= FQ_B;
    }
#pragma unroll
    for (int k = FQ_N; k < 2 * FQ_N - 1; ++k) {
#pragma unroll
        for (int i = k - (FQ_N - 1); 2 * i < k; ++i) acc += (int64_t)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (int64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - (FQ_N - 1); i < FQ_N; ++i) acc += (int64_t)m[i] * FQ_P[k - i];
        r.l[k - FQ_N] = fq_digit(acc);
        acc = (acc + (int64_t(1) << (FQ_B - 1))) >> FQ_B;  // = (acc - digit) / 2^29: round to nearest
    }
    r.l[FQ_N - 1] = (int32_t)acc;
}

// ---- exact, slow: canonical limbs and comparisons mod p (rare paths and the hand-over to the 12 x 32 form) ----
// Full carry propagation: limbs 0..12 in [0, 2^29), the top limb signed -- the unique such representation of the value.
PE_HD void fq_carry(fq& r, const fq& a)
{
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FQ_N - 1; ++i) {
        const int32_t v = a.l[i] + c;
        r.l[i] = v & FQ_MASK;
        c = v >> FQ_B;
    }
    r.l[FQ_N - 1] = a.l[FQ_N - 1] + c;
}
PE_HD bool fq_eq_limbs(const fq& a, const int32_t* b)
{
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) o |= a.l[i] ^ b[i];
    return o == 0;
}
// value == 0 mod p, for a value in [FQ_KP_LO p, (FQ_KP_LO + FQ_KP_N) p) (everything the formulas below produce is far
// inside).  The filter first: value = k p  =>  k = value p^-1 (mod 2^29), and the value's low 29 bits are limb 0's
// (every other limb weighs a multiple of 2^29): one multiply says whether k is one of the few small multiples possible.
PE_HD bool fq_maybe_zero_modp(const fq& a)
{
    const uint32_t k = (0u - (uint32_t)a.l[0] * FQ_N0INV) & (uint32_t)FQ_MASK;  // FQ_N0INV = -p^-1
    return ((k - (uint32_t)FQ_KP_LO) & (uint32_t)FQ_MASK) < (uint32_t)FQ_KP_N;
}
PE_HD bool fq_is_zero_modp_exact(const fq& a)
{
    fq c;
    fq_carry(c, a);
    bool hit = false;
    for (int k = 0; k < FQ_KP_N; ++k) hit = hit || fq_eq_limbs(c, FQ_KP + FQ_N * k);
    return hit;
}
PE_HD bool fq_is_zero_modp(const fq& a) { return fq_maybe_zero_modp(a) && fq_is_zero_modp_exact(a); }

// The unique representative in [0, p) with canonical limbs.  fq_canonical: any value in [-8 p, 9 p); fq_canonical_near:
// a value in (-p, 2 p) -- what a product gives -- in three carry chains.
PE_HD void fq_canonical(fq& r, const fq& a)
{
    fq c, t, u;
    fq_carry(c, a);
    for (int round = 0; round < 9 && c.l[FQ_N - 1] < 0; ++round) {  // negative: add p until it is not
#pragma unroll
        for (int i = 0; i < FQ_N; ++i) t.l[i] = c.l[i] + FQ_P[i];
        fq_carry(c, t);
    }
    for (int round = 0; round < 9; ++round) {  // subtract p while the result stays non-negative
#pragma unroll
        for (int i = 0; i < FQ_N; ++i) t.l[i] = c.l[i] - FQ_P[i];
        fq_carry(u, t);
        if (u.l[FQ_N - 1] < 0) break;  // went below zero: c is the representative
        c = u;
    }
    r = c;
}
PE_HD void fq_canonical_near(fq& r, const fq& a)
{
    fq c, t, lo, hi;
    fq_carry(c, a);
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) t.l[i] = c.l[i] + FQ_P[i];
    fq_carry(lo, t);  // value + p: the answer when the value is negative
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) t.l[i] = c.l[i] - FQ_P[i];
    fq_carry(hi, t);  // value - p: the answer when that is not negative
    const bool neg = c.l[FQ_N - 1] < 0, big = hi.l[FQ_N - 1] >= 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) r.l[i] = neg ? lo.l[i] : big ? hi.l[i] : c.l[i];
}

// ---- hand-over to / from the 12 x 32-bit Montgomery form of fp381.h (R = 2^384, canonical) ----
// words[12] (little-endian 32-bit limbs of a value < 2^384) -> 29-bit limbs of the same integer
PE_HD void fq_from_words32(fq& r, const uint32_t* w)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) {
        const int bit = FQ_B * i, lo = bit >> 5, sh = bit & 31;
        uint64_t v = lo < 12 ? (uint64_t)w[lo] : 0u;
        if (lo + 1 < 12) v |= (uint64_t)w[lo + 1] << 32;
        r.l[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)FQ_MASK);
    }
}
// canonical limbs of a value in [0, 2^384) -> words[12]
PE_HD void fq_to_words32(uint32_t* w, const fq& a)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) w[j] = 0;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) {
        const int bit = FQ_B * i, lo = bit >> 5, sh = bit & 31;
        const uint64_t v = (uint64_t)(uint32_t)a.l[i] << sh;
        if (lo < 12) w[lo] |= (uint32_t)v;
        if (lo + 1 < 12) w[lo + 1] |= (uint32_t)(v >> 32);
    }
}
// x 2^384 mod p (words, canonical) -> x R' mod p, canonical S29 limbs: what the registry table of this form stores
PE_HD void fq_from_mont32(fq& r, const uint32_t* w)
{
    fq a, k, t;
    fq_from_words32(a, w);
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) k.l[i] = FQ_FROM_R32[i];
    fq_mul(t, a, k);
    fq_canonical_near(r, t);
}
// x R' (lazy) -> x 2^384 mod p, canonical words: what k_g1_tree / k_g1_finish read
PE_HD void fq_to_mont32(uint32_t* w, const fq& a)
{
    fq k, t, c;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) k.l[i] = FQ_TO_R32[i];
    fq_mul(t, a, k);
    fq_canonical_near(c, t);
    fq_to_words32(w, c);
}

// ---- a^((p-3)/4): the exponentiation under every square root of the decompression kernels (fp_sqrt.h) ----
// One lane per point there: a chain of ~460 dependent products.  This form's column sums are independent multiply-adds: the
// chain runs at the multiplier's issue rate (301 / 392 multiply-adds per squaring / product).
// The exponent is a constant, so the schedule is too (tools/gen_fp29_consts.py: sliding windows of at most four bits over the
// ODD powers a, a^3 .. a^15): 375 squarings + 78 products + 8 for the table.  The table is EIGHT values held in registers and
// picked by a wave-uniform selector -- rounds 4-5 indexed a 16-entry table at run time, i.e. kept it in scratch: 1120 bytes per
// lane, 147 MB for a chip full of waves, which the runtime hands out per dispatch (profiles/r06_sig_*: the kernel measured
// 15 ms, the call around it 35-50).  No scratch now.
PE_HD void fq_pick_odd(fq& r, uint32_t v, const fq& t1, const fq& t3, const fq& t5, const fq& t7, const fq& t9, const fq& t11,
                       const fq& t13, const fq& t15)
{
#pragma unroll
    for (int i = 0; i < FQ_N; ++i)
        r.l[i] = v == 1 ? t1.l[i] : v == 3 ? t3.l[i] : v == 5 ? t5.l[i] : v == 7 ? t7.l[i] : v == 9 ? t9.l[i]
               : v == 11 ? t11.l[i] : v == 13 ? t13.l[i] : t15.l[i];
}
PE_HD void fq_pow_pm3d4(fq& w, const fq& a)  // a: limbs as the products accept them (|limb| <= 2^29 + 16), any residue
{
    fq t1, t3, t5, t7, t9, t11, t13, t15, a2;
    fq_norm(t1, a);  // balanced digits: signed x signed multiplies below
    fq_sqr(a2, t1);
    fq_mul(t3, t1, a2);
    fq_mul(t5, t3, a2);
    fq_mul(t7, t5, a2);
    fq_mul(t9, t7, a2);
    fq_mul(t11, t9, a2);
    fq_mul(t13, t11, a2);
    fq_mul(t15, t13, a2);
    fq acc;
    fq_pick_odd(acc, FQ_PM3D4_FIRST, t1, t3, t5, t7, t9, t11, t13, t15);
    constexpr int n_sched = (int)(sizeof(FQ_PM3D4_SW) / sizeof(FQ_PM3D4_SW[0]));
#pragma nounroll
    for (int i = 0; i < n_sched; ++i) {
        const uint32_t e = FQ_PM3D4_SW[i];
        const uint32_t n_sq = e & 0xFFu, v = e >> 8;
#pragma nounroll
        for (uint32_t k = 0; k < n_sq; ++k) fq_sqr(acc, acc);
        if (v) {
            fq m;
            fq_pick_odd(m, v, t1, t3, t5, t7, t9, t11, t13, t15);
            fq_mul(acc, acc, m);
        }
    }
    w = acc;
}

}  // namespace posevo
