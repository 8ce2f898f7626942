// g1.h -- BLS12-381 G1 (y^2 = x^3 + 4 over Fp) point arithmetic for gfx950, device only.
//
// Accumulators use extended Jacobian "XYZZ" coordinates (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; ZZ == 0 <=> infinity.  Inputs are affine.  XYZZ costs 10 Montgomery products for a
// mixed add (8M + 2S) and 14 for a full add (12M + 2S) against 11 / 16 for plain Jacobian -- on a
// part whose only wide multiplier is v_mad_u64_u32 the product count is the whole cost.
//
// Every edge case of the group law is exact (inf + Q, P + inf, P + P -> double, P + (-P) -> inf):
// structured synthetic keys ((i+1)*G) hit the doubling branch and P/-P pairs hit infinity
// (SURVEY.md 7 "hard parts" (ii)).  The rare branches are divergent on purpose and INLINED: a
// non-inlined call inside the accumulation loop made the register allocator spill 160 B per point
// to scratch (profiles/: WRITE_SIZE 163 MB per launch before, 0.3 MB after).
//
// Replaces: the point additions inside bls.Aggregate / the pubkey sum of bls.FastAggregateVerify --
// called by is_valid_indexed_attestation (reference call sites pe:736, pe:976; the reference
// itself contains no BLS arithmetic, see oracle/g1.py header).
#pragma once
#include "fp381.h"
#include "fp_inv_safegcd.h"

namespace posevo {

struct g1x {
    fp x, y, zz, zzz;
};
constexpr int G1X_WORDS = 48;  // u32 words of one XYZZ point (192 bytes = PE_G1_PARTIAL_BYTES)

__device__ __forceinline__ void g1x_set_inf(g1x& p)
{
    fp_set_zero(p.x);
    fp_set_zero(p.y);
    fp_set_zero(p.zz);
    fp_set_zero(p.zzz);
}
__device__ __forceinline__ bool g1x_is_inf(const g1x& p) { return fp_is_zero(p.zz); }

// dbl-2008-s-1 (a = 0).  Only ever taken when an accumulator meets an equal point.
// Value in, value out: keeps every coordinate in SSA registers (a by-reference form left the accumulator in
// private memory behind a pointer phi -- 372 B of scratch per lane in the hot loop).
__device__ __forceinline__ g1x g1x_double(const g1x p)
{
    g1x r;
    fp U, V, W, S, M, t, X3, Y3;
    fp_dbl(U, p.y);
    fp_sqr(V, U);
    fp_mul(W, U, V);
    fp_mul(S, p.x, V);
    fp_sqr(M, p.x);
    fp_dbl(t, M);
    fp_add(M, M, t);  // 3 X^2
    fp_sqr(X3, M);
    fp_dbl(t, S);
    fp_sub(X3, X3, t);
    fp_sub(t, S, X3);
    fp_mul(Y3, M, t);
    fp_mul(t, W, p.y);
    fp_sub(Y3, Y3, t);
    fp_mul(r.zz, V, p.zz);
    fp_mul(r.zzz, W, p.zzz);
    r.x = X3;
    r.y = Y3;
    // infinity in, or a point of order two (none exist on this curve, kept for exactness): infinity out
    const bool inf = fp_is_zero(p.zz) || fp_is_zero(p.y);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        r.x.l[j] = inf ? 0u : r.x.l[j];
        r.y.l[j] = inf ? 0u : r.y.l[j];
        r.zz.l[j] = inf ? 0u : r.zz.l[j];
        r.zzz.l[j] = inf ? 0u : r.zzz.l[j];
    }
    return r;
}

// acc += q  (madd-2008-s: 8M + 2S)
__device__ __forceinline__ void g1x_add_affine(g1x& acc, const fp& qx, const fp& qy, bool q_inf)
{
    if (q_inf) return;
    if (g1x_is_inf(acc)) {
        acc.x = qx;
        acc.y = qy;
        fp_set_one(acc.zz);
        fp_set_one(acc.zzz);
        return;
    }
    fp U2, S2, P, R;
    fp_mul(U2, qx, acc.zz);
    fp_mul(S2, qy, acc.zzz);
    fp_sub(P, U2, acc.x);
    fp_sub(R, S2, acc.y);
    if (fp_is_zero(P)) {
        if (fp_is_zero(R)) acc = g1x_double(acc);
        else g1x_set_inf(acc);
        return;
    }
    fp PP, PPP, Q, X3, t;
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, acc.x, PP);
    fp_sqr(X3, R);
    fp_sub(X3, X3, PPP);
    fp_dbl(t, Q);
    fp_sub(X3, X3, t);
    fp_sub(t, Q, X3);
    fp_mul(t, R, t);
    fp_mul(Q, acc.y, PPP);
    fp_sub(acc.y, t, Q);
    fp_mul(acc.zz, acc.zz, PP);
    fp_mul(acc.zzz, acc.zzz, PPP);
    acc.x = X3;
}

// p += q, both XYZZ (add-2008-s: 12M + 2S), single lane, all edge cases
__device__ __forceinline__ void g1x_add(g1x& p, const g1x& q)
{
    if (g1x_is_inf(q)) return;
    if (g1x_is_inf(p)) {
        p = q;
        return;
    }
    fp U1, U2, S1, S2, P, R;
    fp_mul(U1, p.x, q.zz);
    fp_mul(U2, q.x, p.zz);
    fp_mul(S1, p.y, q.zzz);
    fp_mul(S2, q.y, p.zzz);
    fp_sub(P, U2, U1);
    fp_sub(R, S2, S1);
    if (fp_is_zero(P)) {
        if (fp_is_zero(R)) p = g1x_double(p);
        else g1x_set_inf(p);
        return;
    }
    fp PP, PPP, Q, X3, t;
    fp_sqr(PP, P);
    fp_mul(PPP, P, PP);
    fp_mul(Q, U1, PP);
    fp_sqr(X3, R);
    fp_sub(X3, X3, PPP);
    fp_dbl(t, Q);
    fp_sub(X3, X3, t);
    fp_sub(t, Q, X3);
    fp_mul(t, R, t);
    fp_mul(S1, S1, PPP);
    fp_sub(p.y, t, S1);
    fp_mul(t, p.zz, q.zz);
    fp_mul(p.zz, t, PP);
    fp_mul(t, p.zzz, q.zzz);
    fp_mul(p.zzz, t, PPP);
    p.x = X3;
}

// ---- two-lane cooperative add (lanes 2w and 2w+1 of a wave, identical instruction stream) -------------
// A full add is 14 products of dependency depth 7 when split over two lanes.  Both lanes run the SAME
// seven fp_mul's on role-selected operands (role = lane & 1) and swap four field elements through DPP:
//   role 0 ("own" = P1): U1, S1, PP, PPP, Q, R*(Q-X3), S1*PPP                -> produces X3, Y3
//   role 1 ("own" = P2): U2, S2, RR, ZZ1*ZZ2, ZZZ1*ZZZ2, (..)*PP, (..)*PPP   -> produces ZZ3, ZZZ3
// In the LDS tree half of the lanes idle anyway, so the second lane is free.
__device__ __forceinline__ void fp_select(fp& r, bool c, const fp& a, const fp& b)  // r = c ? a : b
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = c ? a.l[j] : b.l[j];
}
__device__ __forceinline__ void fp_xchg(fp& r, const fp& a)  // r = a of the partner lane (lane ^ 1)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = (uint32_t)__shfl_xor((int)a.l[j], 1, 64);
}

// Inputs (per lane): own point's X, Y, ZZ, ZZZ and the partner point's ZZ, ZZZ.  role 0 owns P1, role 1 owns P2.
// Precondition (checked by the caller): neither point is infinity.
// Returns false (outputs undefined) when P == 0, i.e. P1 = +-P2: the caller takes the slow path.
// Outputs: role 0: out_a = X3, out_b = Y3;  role 1: out_a = ZZ3, out_b = ZZZ3.
__device__ __forceinline__ bool g1x_add_pair(fp& out_a, fp& out_b, bool role, const fp& x_own, const fp& y_own,
                                             const fp& zz_own, const fp& zzz_own, const fp& zz_oth, const fp& zzz_oth)
{
    fp m1, m2, o1, o2;
    fp_mul(m1, x_own, zz_oth);   // U1 | U2
    fp_mul(m2, y_own, zzz_oth);  // S1 | S2
    fp_xchg(o1, m1);
    fp_xchg(o2, m2);
    fp U1, S1, P, R;
    {
        fp U2, S2;
        fp_select(U1, role, o1, m1);
        fp_select(U2, role, m1, o1);
        fp_select(S1, role, o2, m2);
        fp_select(S2, role, m2, o2);
        fp_sub(P, U2, U1);
        fp_sub(R, S2, S1);
    }
    if (fp_is_zero(P)) return false;  // identical in both lanes of the pair
    fp a, b, m3, x3, m4, x4, m5;
    fp_select(a, role, R, P);
    fp_sqr(m3, a);    // PP | RR
    fp_xchg(x3, m3);  // role 0 receives RR, role 1 receives PP
    fp_select(a, role, zz_own, P);
    fp_select(b, role, zz_oth, m3);
    fp_mul(m4, a, b);  // PPP | ZZ1*ZZ2
    fp_xchg(x4, m4);   // role 1 receives PPP
    fp_select(a, role, zzz_own, U1);
    fp_select(b, role, zzz_oth, m3);
    fp_mul(m5, a, b);  // Q | ZZZ1*ZZZ2
    // role 0: X3 = RR - PPP - 2Q, T = Q - X3   (role 1 computes don't-cares)
    fp X3, T, t;
    fp_sub(X3, x3, m4);
    fp_dbl(t, m5);
    fp_sub(X3, X3, t);
    fp_sub(T, m5, X3);
    fp m6, m7;
    fp_select(a, role, m4, R);   // ZZ1*ZZ2   | R
    fp_select(b, role, x3, T);   // PP        | Q - X3
    fp_mul(m6, a, b);            // role 0: R*(Q-X3), role 1: ZZ3
    fp_select(a, role, m5, S1);  // ZZZ1*ZZZ2 | S1
    fp_select(b, role, x4, m4);  // PPP (received) | PPP (own)
    fp_mul(m7, a, b);            // role 0: S1*PPP, role 1: ZZZ3
    fp_sub(t, m6, m7);           // role 0: Y3
    fp_select(out_a, role, m6, X3);
    fp_select(out_b, role, m7, t);
    return true;
}

// ---- four-lane cooperative add (lanes 4w .. 4w+3) ---------------------------------------------------------
// Past the first tree level at most a quarter of the lanes hold work, so a pair of points can take four lanes: the 14
// products of an XYZZ add in four rounds of one product per lane (dependency depth 4 instead of 7):
//   round 1   q0: U1 = X1 ZZ2     q1: U2 = X2 ZZ1     q2: S1 = Y1 ZZZ2    q3: S2 = Y2 ZZZ1
//   round 2   q0: PP = P^2        q1: A = ZZ1 ZZ2     q2: RR = R^2        q3: B = ZZZ1 ZZZ2
//   round 3   q0: PPP = P PP      q1: ZZ3 = A PP      q2: Q = U1 PP       q3: (Q)
//   round 4   q0: T2 = S1 PPP     q1: --              q2: T1 = R (Q - X3) q3: ZZZ3 = B PPP
// with P = U2 - U1 (q0), R = S2 - S1 (q2), X3 = RR - PPP - 2Q and Y3 = T1 - T2 (q2).
// Inputs per lane: a, b = (X1, ZZ2) | (X2, ZZ1) | (Y1, ZZZ2) | (Y2, ZZZ1).  Neither point may be infinity (caller).
// Returns false when P == 0 (P1 = +-P2: the caller takes the complete single-lane add).
// Outputs: q2: out_a = X3, out_b = Y3;  q1: out_a = ZZ3;  q3: out_a = ZZZ3.
__device__ __forceinline__ void fp_from_lane(fp& r, const fp& a, int src_lane)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = (uint32_t)__shfl((int)a.l[j], src_lane, 64);
}
__device__ __forceinline__ bool g1x_add_quad(fp& out_a, fp& out_b, int q, const fp& a, const fp& b)
{
    const int base = (int)(threadIdx.x & 63) & ~3;
    const bool even = (q & 1) == 0;
    fp m1, t, d, send;
    fp_mul(m1, a, b);                 // U1 | U2 | S1 | S2
    fp_select(send, even, b, m1);     // q0: ZZ2, q1: U2, q2: ZZZ2, q3: S2
    fp_xchg(t, send);                 // q0: U2,  q1: ZZ2, q2: S2,  q3: ZZZ2
    fp_sub(d, t, m1);                 // q0: P,   q2: R   (odd lanes: unused)
    const int p_zero = __shfl((int)fp_is_zero(d), base, 64);
    if (p_zero) return false;         // identical in the four lanes
    fp x, y, m2;
    fp_select(x, even, d, b);
    fp_select(y, even, d, t);
    fp_mul(m2, x, y);                 // PP | A | RR | B
    fp pp, u1;
    fp_from_lane(pp, m2, base);       // PP to everyone
    fp_from_lane(u1, m1, base);       // U1 to everyone (q3 needs it)
    fp m3, sel;
    fp_select(sel, q == 1, m2, u1);
    fp_select(x, q == 0, d, sel);     // q0: P, q1: A, q2 and q3: U1
    fp_select(y, q == 0, m2, pp);     // q0: PP, others: PP
    fp_mul(m3, x, y);                 // q0: PPP, q1: ZZ3, q2 and q3: Q = U1 PP (q2 keeps its own copy)
    fp ppp, s1;
    fp_from_lane(ppp, m3, base);      // PPP to everyone
    fp_from_lane(s1, m1, base + 2);   // S1 to everyone (q0 needs it)
    fp X3, T, tmp;
    fp_sub(X3, m2, ppp);              // q2: RR - PPP
    fp_dbl(tmp, m3);
    fp_sub(X3, X3, tmp);              // q2: X3 = RR - PPP - 2Q
    fp_sub(T, m3, X3);                // q2: Q - X3
    fp m4;
    fp_select(sel, q == 2, d, m2);
    fp_select(x, q == 0, s1, sel);    // q0: S1, q2: R, q3: B (q1: A, unused)
    fp_select(sel, q == 2, T, ppp);
    fp_select(y, q == 0, m3, sel);    // q0: PPP, q2: Q - X3, q3: PPP
    fp_mul(m4, x, y);                 // q0: T2, q2: T1, q3: ZZZ3
    fp t2;
    fp_from_lane(t2, m4, base);
    fp_sub(tmp, m4, t2);              // q2: Y3
    fp_select(sel, q == 1, m3, m4);
    fp_select(out_a, q == 2, X3, sel);
    out_b = tmp;
    return true;
}

// R^3 mod p: fp_mul(t, R^3) = t * R^2, lifting (xR)^-1 = x^-1 R^-1 back to Montgomery form x^-1 R
__device__ __forceinline__ constexpr uint32_t fp_r3_limb(int j)
{
    return j == 0 ? 0xd94ca1e0u : j == 1 ? 0xed48ac6bu : j == 2 ? 0x03a7adf8u : j == 3 ? 0x315f831eu
         : j == 4 ? 0x615e29ddu : j == 5 ? 0x9a53352au : j == 6 ? 0x921e1761u : j == 7 ? 0x34c04e5eu
         : j == 8 ? 0x65724728u : j == 9 ? 0x2512d435u : j == 10 ? 0x91755d4du : 0x0aa63460u;
}

// r = a^-1 (Montgomery in, Montgomery out) by safegcd divsteps; a == 0 yields 0.
__device__ __noinline__ void fp_inv(fp& r, const fp& a)
{
    uint32_t plain[12];
    sg_modinv(plain, a.l, SG_PINV30);  // (aR)^-1 as a plain integer = a^-1 R^-1
    fp t, r3;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        t.l[j] = plain[j];
        r3.l[j] = fp_r3_limb(j);
    }
    fp_mul(r, t, r3);
}

// r = a^-1 as a PLAIN residue (Montgomery in): a Montgomery product of it with a Montgomery-form value yields the
// plain product, so a chain that ends in serialisation needs no separate fp_from_mont.
__device__ __noinline__ void fp_inv_plain(fp& r, const fp& a)
{
    uint32_t plain[12];
    sg_modinv(plain, a.l, SG_PINV30);  // a^-1 R^-1
    fp t, r2;
#pragma unroll
    for (int j = 0; j < 12; ++j) t.l[j] = plain[j];
    fp_set_r2(r2);
    fp_mul(r, t, r2);  // a^-1 R^-1 * R^2 * R^-1 = a^-1
}

// XYZZ -> affine (x, y) as PLAIN residues ready for fp_store_be48.  One inversion: i = 1/(ZZ*ZZZ) = Z^-5;
// 1/ZZ = i*ZZZ, 1/ZZZ = i*ZZ.  i is taken in the plain domain, which makes every later product plain as well.
__device__ __forceinline__ void g1x_to_affine_plain(fp& x, fp& y, const g1x& p)
{
    fp t, i, izz, izzz;
    fp_mul(t, p.zz, p.zzz);
    fp_inv_plain(i, t);
    fp_mul(izz, i, p.zzz);
    fp_mul(izzz, i, p.zz);
    fp_mul(x, p.x, izz);
    fp_mul(y, p.y, izzz);
}

}  // namespace posevo
