// g2.h -- BLS12-381 G2 (the twist y^2 = x^3 + 4(1+u) over Fp2 = Fp[u]/(u^2+1)) for gfx950, device only.
//
// bls.Aggregate over real BLSSignature points (types pe:37, pe:717; prose pe:659, pe:1536).
//
// Component-parallel Fp2: a lane PAIR (2w, 2w+1) holds one Fp2 value, lane role r = lane & 1 keeps component c_r
// (c0 + c1 u).  Additions are lane-local; a product swaps the partner's halves through DPP (fp_xchg) and costs two
// Montgomery products per lane (schoolbook, 4 per Fp2 product against Karatsuba's 3 in one lane, but the latency of
// 2 and -- the point -- half the registers: a single-lane XYZZ accumulator over Fp2 is 96 VGPRs before any
// temporary, the pair form keeps the G1 kernel's footprint).  A squaring is ONE product per lane:
// (a0+a1)(a0-a1) on role 0, (2 a1) a0 on role 1.
//
// Every predicate (is-zero, equality) is combined across the pair, so the two lanes always take the same branch
// and the exchanges inside the rare branches stay well defined.
#pragma once
#include "g1.h"

namespace posevo {

// both lanes of the pair must execute the exchange: no short-circuit around it
__device__ __forceinline__ bool pair_and(bool v)
{
    const int o = __shfl_xor((int)v, 1, 64);
    return v & (o != 0);
}
__device__ __forceinline__ bool f2_is_zero(const fp& a) { return pair_and(fp_is_zero(a)); }

// r = a * b in Fp2 (this lane's component)
__device__ __forceinline__ void f2_mul(fp& r, const fp& a, const fp& b, bool role)
{
    fp ao, bo, bx, by, t1, t2, n2;
    fp_xchg(ao, a);
    fp_xchg(bo, b);
    fp_select(bx, role, bo, b);  // role 0: a0*b0 - a1*b1      role 1: a1*b0 + a0*b1
    fp_select(by, role, b, bo);
    fp_mul(t1, a, bx);
    fp_mul(t2, ao, by);
    fp_neg(n2, t2);
    fp_select(t2, role, t2, n2);
    fp_add(r, t1, t2);
}
__device__ __forceinline__ void f2_sqr(fp& r, const fp& a, bool role)
{
    fp ao, s, d, dd, m, n;
    fp_xchg(ao, a);
    fp_add(s, a, ao);
    fp_sub(d, a, ao);
    fp_dbl(dd, a);
    fp_select(m, role, dd, s);  // role 0: (a0+a1)(a0-a1)      role 1: (2 a1) a0
    fp_select(n, role, ao, d);
    fp_mul(r, m, n);
}
// r = a^-1 = conj(a) / (a0^2 + a1^2) as a PLAIN Fp2 value (see fp_inv_plain); a == 0 yields 0
__device__ __forceinline__ void f2_inv_plain(fp& r, const fp& a, bool role)
{
    fp sq, sqo, n, ni, t, nt;
    fp_sqr(sq, a);
    fp_xchg(sqo, sq);
    fp_add(n, sq, sqo);
    fp_inv_plain(ni, n);
    fp_mul(t, a, ni);
    fp_neg(nt, t);
    fp_select(r, role, nt, t);
}

// XYZZ point, this lane's halves
struct g2x {
    fp x, y, zz, zzz;
};
constexpr int G2X_WORDS = 96;  // u32 words of one XYZZ point over Fp2 (384 bytes)

__device__ __forceinline__ void g2x_set_inf(g2x& p)
{
    fp_set_zero(p.x);
    fp_set_zero(p.y);
    fp_set_zero(p.zz);
    fp_set_zero(p.zzz);
}
__device__ __forceinline__ bool g2x_is_inf(const g2x& p) { return f2_is_zero(p.zz); }
__device__ __forceinline__ void f2_set_one(fp& r, bool role)
{
#pragma unroll
    for (int j = 0; j < 12; ++j) r.l[j] = role ? 0u : fp_r1_limb(j);
}

// dbl-2008-s-1 (a = 0)
__device__ __forceinline__ g2x g2x_double(const g2x p, bool role)
{
    g2x r;
    fp U, V, W, S, M, t, X3, Y3;
    fp_dbl(U, p.y);
    f2_sqr(V, U, role);
    f2_mul(W, U, V, role);
    f2_mul(S, p.x, V, role);
    f2_sqr(M, p.x, role);
    fp_dbl(t, M);
    fp_add(M, M, t);
    f2_sqr(X3, M, role);
    fp_dbl(t, S);
    fp_sub(X3, X3, t);
    fp_sub(t, S, X3);
    f2_mul(Y3, M, t, role);
    f2_mul(t, W, p.y, role);
    fp_sub(Y3, Y3, t);
    f2_mul(r.zz, V, p.zz, role);
    f2_mul(r.zzz, W, p.zzz, role);
    r.x = X3;
    r.y = Y3;
    const bool inf = f2_is_zero(p.zz) || f2_is_zero(p.y);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        r.x.l[j] = inf ? 0u : r.x.l[j];
        r.y.l[j] = inf ? 0u : r.y.l[j];
        r.zz.l[j] = inf ? 0u : r.zz.l[j];
        r.zzz.l[j] = inf ? 0u : r.zzz.l[j];
    }
    return r;
}

// acc += q (affine), madd-2008-s: 8M + 2S in Fp2 = 18 Montgomery products per lane
__device__ __forceinline__ void g2x_add_affine(g2x& acc, const fp& qx, const fp& qy, bool q_inf, bool role)
{
    if (q_inf) return;
    if (g2x_is_inf(acc)) {
        acc.x = qx;
        acc.y = qy;
        f2_set_one(acc.zz, role);
        f2_set_one(acc.zzz, role);
        return;
    }
    fp U2, S2, P, R;
    f2_mul(U2, qx, acc.zz, role);
    f2_mul(S2, qy, acc.zzz, role);
    fp_sub(P, U2, acc.x);
    fp_sub(R, S2, acc.y);
    if (f2_is_zero(P)) {
        if (f2_is_zero(R)) acc = g2x_double(acc, role);
        else g2x_set_inf(acc);
        return;
    }
    fp PP, PPP, Q, X3, t;
    f2_sqr(PP, P, role);
    f2_mul(PPP, P, PP, role);
    f2_mul(Q, acc.x, PP, role);
    f2_sqr(X3, R, role);
    fp_sub(X3, X3, PPP);
    fp_dbl(t, Q);
    fp_sub(X3, X3, t);
    fp_sub(t, Q, X3);
    f2_mul(t, R, t, role);
    f2_mul(Q, acc.y, PPP, role);
    fp_sub(acc.y, t, Q);
    f2_mul(acc.zz, acc.zz, PP, role);
    f2_mul(acc.zzz, acc.zzz, PPP, role);
    acc.x = X3;
}

// p += q, both XYZZ (add-2008-s: 12M + 2S in Fp2 = 26 products per lane), all edge cases
__device__ __forceinline__ void g2x_add(g2x& p, const g2x& q, bool role)
{
    if (g2x_is_inf(q)) return;
    if (g2x_is_inf(p)) {
        p = q;
        return;
    }
    fp U1, U2, S1, S2, P, R;
    f2_mul(U1, p.x, q.zz, role);
    f2_mul(U2, q.x, p.zz, role);
    f2_mul(S1, p.y, q.zzz, role);
    f2_mul(S2, q.y, p.zzz, role);
    fp_sub(P, U2, U1);
    fp_sub(R, S2, S1);
    if (f2_is_zero(P)) {
        if (f2_is_zero(R)) p = g2x_double(p, role);
        else g2x_set_inf(p);
        return;
    }
    fp PP, PPP, Q, X3, t;
    f2_sqr(PP, P, role);
    f2_mul(PPP, P, PP, role);
    f2_mul(Q, U1, PP, role);
    f2_sqr(X3, R, role);
    fp_sub(X3, X3, PPP);
    fp_dbl(t, Q);
    fp_sub(X3, X3, t);
    fp_sub(t, Q, X3);
    f2_mul(t, R, t, role);
    f2_mul(S1, S1, PPP, role);
    fp_sub(p.y, t, S1);
    f2_mul(t, p.zz, q.zz, role);
    f2_mul(p.zz, t, PP, role);
    f2_mul(t, p.zzz, q.zzz, role);
    f2_mul(p.zzz, t, PPP, role);
    p.x = X3;
}

// XYZZ -> affine (this lane's halves of x and y) as PLAIN residues; one Fp inversion per pair (done in both lanes)
__device__ __forceinline__ void g2x_to_affine_plain(fp& x, fp& y, const g2x& p, bool role)
{
    fp t, i, izz, izzz;
    f2_mul(t, p.zz, p.zzz, role);
    f2_inv_plain(i, t, role);
    f2_mul(izz, i, p.zzz, role);
    f2_mul(izzz, i, p.zz, role);
    f2_mul(x, p.x, izz, role);
    f2_mul(y, p.y, izzz, role);
}

}  // namespace posevo
