// Host build of the S29 field / G1 code (pos_evolution_amd/csrc/fp381_s29.h, g1_s29.h) behind a C interface, for
// tests/test_host_fp29.py:  g++ -O2 -shared -fPIC tests/native/fp29_host.cpp -o <tmp>/libfp29.so
// The SAME source the gfx950 kernels compile; no GPU involved.
#include <string.h>

#include "../../pos_evolution_amd/csrc/g1_s29.h"

using namespace posevo;

extern "C" {

void fq29_mul(const int32_t* a, const int32_t* b, int32_t* r)
{
    fq x, y, z;
    memcpy(x.l, a, sizeof(x.l));
    memcpy(y.l, b, sizeof(y.l));
    fq_mul(z, x, y);
    memcpy(r, z.l, sizeof(z.l));
}
void fq29_sqr(const int32_t* a, int32_t* r)
{
    fq x, z;
    memcpy(x.l, a, sizeof(x.l));
    fq_sqr(z, x);
    memcpy(r, z.l, sizeof(z.l));
}
void fq29_norm(const int32_t* a, int32_t* r)
{
    fq x, z;
    memcpy(x.l, a, sizeof(x.l));
    fq_norm(z, x);
    memcpy(r, z.l, sizeof(z.l));
}
void fq29_canonical(const int32_t* a, int32_t* r, int near)
{
    fq x, z;
    memcpy(x.l, a, sizeof(x.l));
    if (near) fq_canonical_near(z, x);
    else fq_canonical(z, x);
    memcpy(r, z.l, sizeof(z.l));
}
int fq29_is_zero_modp(const int32_t* a, int* filter)
{
    fq x;
    memcpy(x.l, a, sizeof(x.l));
    *filter = fq_maybe_zero_modp(x) ? 1 : 0;
    return fq_is_zero_modp(x) ? 1 : 0;
}
void fq29_from_mont32(const uint32_t* w12, int32_t* r)
{
    fq z;
    fq_from_mont32(z, w12);
    memcpy(r, z.l, sizeof(z.l));
}
void fq29_to_mont32(const int32_t* a, uint32_t* w12)
{
    fq x;
    memcpy(x.l, a, sizeof(x.l));
    fq_to_mont32(w12, x);
}
void fq29_words(const uint32_t* w12, int32_t* r, uint32_t* back)
{
    fq z;
    fq_from_words32(z, w12);
    memcpy(r, z.l, sizeof(z.l));
    fq_to_words32(back, z);
}

// One lane's run: n table rows (x, y as 12-word Montgomery values of the 32-bit form, all zero = no point) added into
// an empty accumulator in order; out = the 48 XYZZ words k_g1_tree would read.  max_abs_limb (optional) receives the
// largest |limb| any accumulator coordinate held between adds: the bound the products rely on.
void g1q_run(const uint32_t* rows24, int n, uint32_t* out48, int32_t* max_abs_limb)
{
    g1q acc;
    g1q_set_inf(acc);
    int32_t worst = 0;
    for (int j = 0; j < n; ++j) {
        const uint32_t* row = rows24 + 24 * j;
        uint32_t any = 0;
        for (int k = 0; k < 24; ++k) any |= row[k];
        fq qx, qy;
        fq_from_mont32(qx, row);
        fq_from_mont32(qy, row + 12);
        g1q_add_affine(acc, qx, qy, any == 0);
        const fq* cs[4] = {&acc.x, &acc.y, &acc.zz, &acc.zzz};
        for (const fq* c : cs)
            for (int i = 0; i < FQ_N - 1; ++i) {
                const int32_t v = c->l[i] < 0 ? -c->l[i] : c->l[i];
                if (v > worst) worst = v;
            }
    }
    if (max_abs_limb) *max_abs_limb = worst;
    g1q_to_words32(out48, acc);
}

// The same run the way k_g1_accumulate does it: the first point becomes the accumulator as it is (g1q_set_first), every
// later one goes through the general body alone (g1q_madd_fast), a same-x case only raises the flag and the whole run is
// then redone by the complete add.  *took_slow_path says which way the run went.
void g1q_run_kernel_way(const uint32_t* rows24, int n, uint32_t* out48, int32_t* max_abs_limb, int* took_slow_path)
{
    g1q acc;
    g1q_set_inf(acc);
    bool exc = false;
    int32_t worst = 0;
    for (int j = 0; j < n; ++j) {
        const uint32_t* row = rows24 + 24 * j;
        uint32_t any = 0;
        for (int k = 0; k < 24; ++k) any |= row[k];
        if (!any) continue;
        fq qx, qy;
        fq_from_mont32(qx, row);
        fq_from_mont32(qy, row + 12);
        if (acc.inf) g1q_set_first(acc, qx, qy);
        else g1q_madd_fast(acc, qx, qy, exc);
        if (exc) break;  // the kernel's lane goes on over garbage; nothing of it is used
        const fq* cs[4] = {&acc.x, &acc.y, &acc.zz, &acc.zzz};
        for (const fq* c : cs)
            for (int i = 0; i < FQ_N - 1; ++i) {
                const int32_t v = c->l[i] < 0 ? -c->l[i] : c->l[i];
                if (v > worst) worst = v;
            }
    }
    *took_slow_path = exc ? 1 : 0;
    if (max_abs_limb) *max_abs_limb = worst;
    if (exc) {
        g1q_run(rows24, n, out48, nullptr);
        return;
    }
    g1q_to_words32(out48, acc);
}

// k_g1_accumulate + k_g1_tree at the level of their formulas (round 6: the tree adds in S29): the n rows go to lanes of k rows each,
// every lane accumulates the kernel's way, and the lanes' accumulators -- handed over as they are, lazy limbs and all -- are
// reduced pairwise, level by level, with the complete add g1q_add.  out = the 48 words k_g1_finish would read; max_abs_limb = the
// largest |limb| a coordinate held between two adds of the tree.
void g1q_tree_run(const uint32_t* rows24, int n, int k, uint32_t* out48, int32_t* max_abs_limb)
{
    const int lanes = (n + k - 1) / k;
    g1q* acc = new g1q[lanes > 0 ? lanes : 1];
    for (int l = 0; l < lanes; ++l) {
        g1q_set_inf(acc[l]);
        for (int j = l * k; j < n && j < (l + 1) * k; ++j) {
            const uint32_t* row = rows24 + 24 * j;
            uint32_t any = 0;
            for (int w = 0; w < 24; ++w) any |= row[w];
            fq qx, qy;
            fq_from_mont32(qx, row);
            fq_from_mont32(qy, row + 12);
            g1q_add_affine(acc[l], qx, qy, any == 0);
        }
        if (acc[l].inf) g1q_set_inf(acc[l]);  // all limbs zero: what the hand-over writes for an empty lane
    }
    int32_t worst = 0;
    for (int m = lanes; m > 1; m = (m + 1) / 2) {
        for (int i = 0; i < m / 2; ++i) {
            g1q a = acc[2 * i];
            // the tree learns "infinity" from the limbs (all zero), not from a flag
            a.inf = fq_limbs_zero(a.zz);
            g1q b = acc[2 * i + 1];
            b.inf = fq_limbs_zero(b.zz);
            a.affine = b.affine = false;
            g1q_add(a, b);
            if (a.inf) g1q_set_inf(a);
            acc[i] = a;
            const fq* cs[4] = {&a.x, &a.y, &a.zz, &a.zzz};
            for (const fq* c : cs)
                for (int t = 0; t < FQ_N - 1; ++t) {
                    const int32_t v = c->l[t] < 0 ? -c->l[t] : c->l[t];
                    if (v > worst) worst = v;
                }
        }
        if (m & 1) acc[m / 2] = acc[m - 1];
    }
    if (max_abs_limb) *max_abs_limb = worst;
    if (lanes == 0) { for (int w = 0; w < 48; ++w) out48[w] = 0; }
    else {
        g1q r = acc[0];
        r.inf = fq_limbs_zero(r.zz);
        g1q_to_words32(out48, r);
    }
    delete[] acc;
}

// fp_sqrt.h's fp_pow_pm3d4 as the decompression kernels run it: Montgomery words (R = 2^384) in and out
void fq29_pow_pm3d4_words(const uint32_t* in12, uint32_t* out12, int32_t* max_abs_limb)
{
    fq x, r;
    fq_from_mont32(x, in12);
    fq_pow_pm3d4(r, x);
    int32_t worst = 0;
    for (int i = 0; i < FQ_N - 1; ++i) {
        const int32_t v = r.l[i] < 0 ? -r.l[i] : r.l[i];
        if (v > worst) worst = v;
    }
    if (max_abs_limb) *max_abs_limb = worst;
    fq_to_mont32(out12, r);
}

}  // extern "C"
