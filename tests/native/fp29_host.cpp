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
