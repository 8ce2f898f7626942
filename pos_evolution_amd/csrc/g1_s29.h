// g1_s29.h -- the accumulation side of g1.h over the S29 field form (fp381_s29.h): XYZZ accumulator += affine point,
// and the hand-over of a finished accumulator to the 12 x 32-bit XYZZ words k_g1_tree / k_g1_finish read.
//
// Same formulas as g1.h (madd-2008-s / mmadd-2008-s / dbl-2008-s-1), same exact edge cases; what differs is the
// bookkeeping of the lazy form: products come out in (-eps, p + eps) with balanced limbs (|limb| <= 2^28); a difference of
// two such values goes into the next product as it is; X3 (three terms) and Y3 (stored, subtracted from a product in the next
// add) take one carry pass each.  Two carry passes, six limb-wise subtractions, eight products and two squarings per
// mixed add.  Infinity is a flag beside the accumulator, not a zero test.
//
// Host + device (see fp381_s29.h): tests/test_host_fp29.py runs these functions on the CPU against oracle/g1.py.
#pragma once
#include "fp381_s29.h"

namespace posevo {

// Where a function's products come from.  FqInline: fq_mul / fq_sqr expanded in place (straight-line, ~3 KB of code per
// product): for the one hot loop body.  A kernel may pass a policy whose mul / sqr CALL a single non-inlined copy instead:
// everything that runs once per lane (the hand-over below) or almost never (doubling, the complete add) then costs a few
// hundred bytes of code, not tens of kilobytes -- two CUs share a 64 KB instruction cache, and tools/icbench.hip measured
// what leaving it costs (a dependent chain of mixed adds: -20 % at two waves per SIMD, -55 % at one).
struct FqInline {
    PE_HD_MEMBER void mul(fq& r, const fq& a, const fq& b) { fq_mul(r, a, b); }
    PE_HD_MEMBER void sqr(fq& r, const fq& a) { fq_sqr(r, a); }
};

struct g1q {
    fq x, y, zz, zzz;
    bool inf;     // the point at infinity (the coordinates are then meaningless)
    bool affine;  // x, y are a table row and zz = zzz = 1 is implied (zz / zzz do hold the constant): a lane's first point
};

PE_HD void g1q_set_inf(g1q& p)
{
    fq_set_zero(p.x);
    fq_set_zero(p.y);
    fq_set_zero(p.zz);
    fq_set_zero(p.zzz);
    p.inf = true;
    p.affine = false;
}

// dbl-2008-s-1 (a = 0) of an XYZZ point.  Rare (an accumulator meets an equal point): carry passes used freely.
template <class MP = FqInline> PE_HD void g1q_double(g1q& p)
{
    if (p.inf) return;
    if (fq_is_zero_modp(p.y)) {  // a point of order two: none on this curve, kept for exactness
        g1q_set_inf(p);
        return;
    }
    fq U, V, W, S, M, t, X3, Y3, yn, xn;
    fq_norm(yn, p.y);
    fq_norm(xn, p.x);
    fq_add(U, yn, yn);           // 2 Y
    MP::sqr(V, U);
    MP::mul(W, U, V);
    MP::mul(S, xn, V);
    MP::sqr(M, xn);
    fq_add(t, M, M);
    fq_add(M, M, t);             // 3 X^2
    MP::sqr(X3, M);
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) t.l[i] = X3.l[i] - 2 * S.l[i];
    fq_norm(X3, t);              // M^2 - 2 S
    fq_sub_norm(t, S, X3);
    MP::mul(Y3, M, t);
    MP::mul(t, W, yn);
    fq_sub_norm(Y3, Y3, t);
    fq zz = p.zz, zzz = p.zzz;
    MP::mul(p.zz, V, zz);
    MP::mul(p.zzz, W, zzz);
    p.x = X3;
    p.y = Y3;
}

// acc += (qx, qy), an affine point with canonical limbs (the registry table's rows); q_none: the row encodes "no point".
// acc.x / acc.y hold a table row (the first point) or carry-passed values (|limb| <= 2^28 + 4), acc.zz / acc.zzz products
// (or the constant one).  The ten products sit in ONE basic block on purpose: hipcc selects v_mad_i64_i32 only where it
// sees the 32 -> 64-bit sign extension next to the multiply; operands extended in another block (the first version of
// this function branched around the ZZ / ZZZ products of a lane's first add) become generic 64 x 64 multiplies, four
// instructions each.  A lane's first add (affine + affine, six products) is therefore a second straight-line body.
template <class MP = FqInline> PE_HD void g1q_add_affine(g1q& acc, const fq& qx, const fq& qy, bool q_none)
{
    if (q_none) return;
    if (acc.inf) {
        acc.x = qx;
        acc.y = qy;
        fq_set_one(acc.zz);
        fq_set_one(acc.zzz);
        acc.inf = false;
        acc.affine = true;
        return;
    }
    if (acc.affine) {
        // mmadd-2008-s, affine + affine (4M + 2S): a body of its own, straight-line like the general one below -- sharing
        // the tail with it would put products and their operands' sign extensions into different basic blocks
        fq P, R;
        fq_sub(P, qx, acc.x);
        fq_sub(R, qy, acc.y);
        acc.affine = false;
        if (fq_maybe_zero_modp(P) && fq_is_zero_modp_exact(P)) {
            if (fq_is_zero_modp(R)) g1q_double<MP>(acc);
            else g1q_set_inf(acc);
            return;
        }
        fq PP, PPP, Q, X3, t, u;
        MP::sqr(PP, P);
        MP::mul(PPP, P, PP);
        MP::mul(Q, acc.x, PP);
        MP::sqr(X3, R);
        fq_sub_sub2_norm(X3, X3, PPP, Q);
        fq_sub(t, Q, X3);
        MP::mul(t, R, t);
        MP::mul(u, acc.y, PPP);
        fq_sub_norm(acc.y, t, u);
        acc.zz = PP;
        acc.zzz = PPP;
        acc.x = X3;
        return;
    }
    fq U2, S2, P, R;
    MP::mul(U2, qx, acc.zz);
    MP::mul(S2, qy, acc.zzz);
    fq_sub(P, U2, acc.x);
    fq_sub(R, S2, acc.y);
    if (fq_maybe_zero_modp(P) && fq_is_zero_modp_exact(P)) {  // same x: the same point or its negative
        if (fq_is_zero_modp(R)) g1q_double<MP>(acc);
        else g1q_set_inf(acc);
        return;
    }
    fq PP, PPP, Q, X3, t, u, zz, zzz;
    MP::sqr(PP, P);
    MP::mul(PPP, P, PP);
    MP::mul(Q, acc.x, PP);
    MP::sqr(X3, R);
    fq_sub_sub2_norm(X3, X3, PPP, Q);  // R^2 - PPP - 2 Q: |limb| <= 2^30 before the pass
    fq_sub(t, Q, X3);
    MP::mul(t, R, t);
    MP::mul(u, acc.y, PPP);
    fq_sub_norm(acc.y, t, u);
    MP::mul(zz, acc.zz, PP);
    MP::mul(zzz, acc.zzz, PPP);
    acc.zz = zz;
    acc.zzz = zzz;
    acc.x = X3;
}

// The general body alone, for the accumulation kernel's loop: acc is a finite point whose x / y are carry-passed (a
// table row goes through fq_norm once when it becomes the accumulator, zz = zzz = one) -- no infinity, no first-add body,
// no doubling: ONE straight-line body of eight products and two squarings is all the loop holds.  The same-x case (the
// same point or its negative) is only DETECTED (the one-multiply filter on P, the exact comparison behind it); a lane that ever raises `exc` has its
// whole run redone by g1q_add_affine afterwards (its accumulator is garbage from here on: integers, nothing traps).
PE_HD void g1q_madd_fast(g1q& acc, const fq& qx, const fq& qy, bool& exc)
{
    fq U2, S2, P, R;
    fq_mul(U2, qx, acc.zz);
    fq_mul(S2, qy, acc.zzz);
    fq_sub(P, U2, acc.x);
    fq_sub(R, S2, acc.y);
    // the filter passes ~25 values in 2^29 that are no multiples of p: settle those here (cold code), or one such lane
    // in a launch costs its wave a whole second run
    if (__builtin_expect(fq_maybe_zero_modp(P), 0)) exc = exc || fq_is_zero_modp_exact(P);
    fq PP, PPP, Q, X3, t, u, zz, zzz;
    fq_sqr(PP, P);
    fq_mul(PPP, P, PP);
    fq_mul(Q, acc.x, PP);
    fq_sqr(X3, R);
    fq_sub_sub2_norm(X3, X3, PPP, Q);
    fq_sub(t, Q, X3);
    fq_mul(t, R, t);
    fq_mul(u, acc.y, PPP);
    fq_sub_norm(acc.y, t, u);
    fq_mul(zz, acc.zz, PP);
    fq_mul(zzz, acc.zzz, PPP);
    acc.zz = zz;
    acc.zzz = zzz;
    acc.x = X3;
}
// a table row as the accumulator of g1q_madd_fast
PE_HD void g1q_set_first(g1q& acc, const fq& qx, const fq& qy)
{
    fq_norm(acc.x, qx);
    fq_norm(acc.y, qy);
    fq_set_one(acc.zz);
    fq_set_one(acc.zzz);
    acc.inf = false;
    acc.affine = false;
}

// A finished accumulator as the 48 words of a g1x in the 12 x 32-bit Montgomery form (fp381.h): X, Y, ZZ, ZZZ, all
// zero for infinity.  Four products and four exact reductions: once per lane.
template <class MP = FqInline> PE_HD void fq_to_mont32_via(uint32_t* w, const fq& a)  // fq_to_mont32 with the policy's product
{
    fq k, t, c;
#pragma unroll
    for (int i = 0; i < FQ_N; ++i) k.l[i] = FQ_TO_R32[i];
    MP::mul(t, a, k);
    fq_canonical_near(c, t);
    fq_to_words32(w, c);
}
// p += q, both in XYZZ form with lazy coordinates -- X and Y carry-passed values (or a table row's canonical limbs), ZZ and ZZZ
// products (or the constant one) --, every case of the group law (add-2008-s: 12 products + 2 squarings).  What k_g1_tree adds
// the lanes' accumulators with (round 6); the cooperative two- and four-lane versions in g1_kernels.hip are this formula spread
// over lanes, and fall back to it for an infinity operand or P1 = +-P2.
template <class MP = FqInline> PE_HD void g1q_add(g1q& p, const g1q& q)
{
    if (q.inf) return;
    if (p.inf) { p = q; return; }
    fq U1, U2, S1, S2, P, R;
    MP::mul(U1, p.x, q.zz);
    MP::mul(U2, q.x, p.zz);
    MP::mul(S1, p.y, q.zzz);
    MP::mul(S2, q.y, p.zzz);
    fq_sub(P, U2, U1);
    fq_sub(R, S2, S1);
    if (fq_is_zero_modp(P)) {
        if (fq_is_zero_modp(R)) g1q_double<MP>(p);
        else g1q_set_inf(p);
        return;
    }
    fq PP, PPP, Q, X3, t;
    MP::sqr(PP, P);
    MP::mul(PPP, P, PP);
    MP::mul(Q, U1, PP);
    MP::sqr(X3, R);
    fq_sub_sub2_norm(X3, X3, PPP, Q);
    fq_sub(t, Q, X3);
    MP::mul(t, R, t);
    MP::mul(S1, S1, PPP);
    fq_sub_norm(p.y, t, S1);
    MP::mul(t, p.zz, q.zz);
    MP::mul(p.zz, t, PP);
    MP::mul(t, p.zzz, q.zzz);
    MP::mul(p.zzz, t, PPP);
    p.x = X3;
    p.affine = false;
}

template <class MP = FqInline> PE_HD void g1q_to_words32(uint32_t* w48, const g1q& p)
{
    if (p.inf) {
#pragma unroll
        for (int k = 0; k < 48; ++k) w48[k] = 0;
        return;
    }
    fq_to_mont32_via<MP>(w48, p.x);
    fq_to_mont32_via<MP>(w48 + 12, p.y);
    fq_to_mont32_via<MP>(w48 + 24, p.zz);
    fq_to_mont32_via<MP>(w48 + 36, p.zzz);
}

}  // namespace posevo
