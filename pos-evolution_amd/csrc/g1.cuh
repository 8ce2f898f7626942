// g1.cuh -- BLS12-381 G1 (y^2 = x^3 + 4 over Fp) point arithmetic for gfx950, device only.
//
// Accumulators are Jacobian (X : Y : Z), Z == 0 <=> infinity; inputs are affine.
// Every edge case of the group law is exact (inf + Q, P + inf, P + P -> double,
// P + (-P) -> inf): structured synthetic keys ((i+1)*G) hit the doubling branch
// and P/-P pairs hit infinity (SURVEY.md 7 "hard parts" (ii)).  The rare
// branches are divergent on purpose: the common path stays straight-line.
//
// Replaces: the point additions inside bls.Aggregate / the pubkey sum of
// bls.FastAggregateVerify -- called by is_valid_indexed_attestation
// (reference call sites pe:736, pe:976; the reference itself contains no BLS
// arithmetic, see oracle/g1.py header).
#pragma once
#include "fp381.cuh"
#include "fp_inv_safegcd.h"

namespace posevo {

struct g1j {
    fp x, y, z;
};
struct g1a {
    fp x, y;
    uint32_t inf;
};

__device__ __forceinline__ void g1j_set_inf(g1j& p)
{
    fp_set_zero(p.x);
    fp_set_zero(p.y);
    fp_set_zero(p.z);
}
__device__ __forceinline__ bool g1j_is_inf(const g1j& p) { return fp_is_zero(p.z); }

// dbl-2009-l (a = 0)
__device__ __forceinline__ void g1j_double(g1j& r, const g1j& p)
{
    if (g1j_is_inf(p) || fp_is_zero(p.y)) {
        g1j_set_inf(r);
        return;
    }
    fp A, B, C, D, E, F, t;
    fp_sqr(A, p.x);
    fp_sqr(B, p.y);
    fp_sqr(C, B);
    fp_add(t, p.x, B);
    fp_sqr(t, t);
    fp_sub(t, t, A);
    fp_sub(t, t, C);
    fp_dbl(D, t);
    fp_dbl(E, A);
    fp_add(E, E, A);
    fp_sqr(F, E);
    fp X3, Y3, Z3;
    fp_dbl(t, D);
    fp_sub(X3, F, t);
    fp_mul(Z3, p.y, p.z);
    fp_dbl(Z3, Z3);
    fp_sub(t, D, X3);
    fp_mul(Y3, E, t);
    fp_dbl(C, C);
    fp_dbl(C, C);
    fp_dbl(C, C);
    fp_sub(Y3, Y3, C);
    r.x = X3;
    r.y = Y3;
    r.z = Z3;
}

// acc += q (mixed addition, 8M + 3S).  q.inf handled by the caller-visible flag.
__device__ __forceinline__ void g1j_add_affine(g1j& acc, const fp& qx, const fp& qy, bool q_inf)
{
    if (q_inf) return;
    if (g1j_is_inf(acc)) {
        acc.x = qx;
        acc.y = qy;
        fp_set_one(acc.z);
        return;
    }
    fp Z2, U2, S2, H, Rr;
    fp_sqr(Z2, acc.z);
    fp_mul(U2, qx, Z2);
    fp_mul(S2, qy, Z2);
    fp_mul(S2, S2, acc.z);
    fp_sub(H, U2, acc.x);
    fp_sub(Rr, S2, acc.y);
    if (fp_is_zero(H)) {
        if (fp_is_zero(Rr)) {
            g1j t = acc;
            g1j_double(acc, t);
        } else {
            g1j_set_inf(acc);
        }
        return;
    }
    fp HH, HHH, V, t;
    fp_sqr(HH, H);
    fp_mul(HHH, HH, H);
    fp_mul(V, acc.x, HH);
    fp X3;
    fp_sqr(X3, Rr);
    fp_sub(X3, X3, HHH);
    fp_dbl(t, V);
    fp_sub(X3, X3, t);
    fp_sub(t, V, X3);
    fp_mul(t, Rr, t);
    fp_mul(HHH, acc.y, HHH);
    fp_sub(acc.y, t, HHH);
    fp_mul(acc.z, acc.z, H);
    acc.x = X3;
}

// p += q, both Jacobian (add-2007-bl shape, 12M + 4S)
__device__ __forceinline__ void g1j_add(g1j& p, const g1j& q)
{
    if (g1j_is_inf(q)) return;
    if (g1j_is_inf(p)) {
        p = q;
        return;
    }
    fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr;
    fp_sqr(Z1Z1, p.z);
    fp_sqr(Z2Z2, q.z);
    fp_mul(U1, p.x, Z2Z2);
    fp_mul(U2, q.x, Z1Z1);
    fp_mul(S1, p.y, Z2Z2);
    fp_mul(S1, S1, q.z);
    fp_mul(S2, q.y, Z1Z1);
    fp_mul(S2, S2, p.z);
    fp_sub(H, U2, U1);
    fp_sub(Rr, S2, S1);
    if (fp_is_zero(H)) {
        if (fp_is_zero(Rr)) {
            g1j t = p;
            g1j_double(p, t);
        } else {
            g1j_set_inf(p);
        }
        return;
    }
    fp HH, HHH, V, t;
    fp_sqr(HH, H);
    fp_mul(HHH, HH, H);
    fp_mul(V, U1, HH);
    fp X3;
    fp_sqr(X3, Rr);
    fp_sub(X3, X3, HHH);
    fp_dbl(t, V);
    fp_sub(X3, X3, t);
    fp_sub(t, V, X3);
    fp_mul(t, Rr, t);
    fp_mul(S1, S1, HHH);
    fp_sub(p.y, t, S1);
    fp_mul(t, p.z, q.z);
    fp_mul(p.z, t, H);
    p.x = X3;
}

// a^(p-2): plain square-and-multiply over the 381 exponent bits (Fermat).
__device__ __noinline__ void fp_inv_fermat(fp& r, const fp& a)
{
    fp acc, base = a;
    fp_set_one(acc);
    for (int i = 0; i < 12; ++i) {
        uint32_t e = fp_p_limb(0);
        // p - 2: only limb 0 differs (…aaab - 2 = …aaa9)
        switch (i) {
            case 0: e = fp_p_limb(0) - 2u; break;
            case 1: e = fp_p_limb(1); break;
            case 2: e = fp_p_limb(2); break;
            case 3: e = fp_p_limb(3); break;
            case 4: e = fp_p_limb(4); break;
            case 5: e = fp_p_limb(5); break;
            case 6: e = fp_p_limb(6); break;
            case 7: e = fp_p_limb(7); break;
            case 8: e = fp_p_limb(8); break;
            case 9: e = fp_p_limb(9); break;
            case 10: e = fp_p_limb(10); break;
            default: e = fp_p_limb(11); break;
        }
        for (int b = 0; b < 32; ++b) {
            if ((e >> b) & 1u) fp_mul(acc, acc, base);
            fp_sqr(base, base);
        }
    }
    r = acc;
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

}  // namespace posevo
