// Host build of the engine's safegcd inversion (plain C++ path of fp_inv_safegcd.h) for CPU tests.
#include "../../pos-evolution_amd/csrc/fp_inv_safegcd.h"
extern "C" int host_modinv(uint32_t* out, const uint32_t* x) { return posevo::sg_modinv(out, x, posevo::SG_PINV30); }
extern "C" int host_modinv_many(uint32_t* out, const uint32_t* x, int n)
{
    int worst = 0;
    for (int i = 0; i < n; ++i) {
        int b = posevo::sg_modinv(out + 12 * i, x + 12 * i, posevo::SG_PINV30);
        if (b > worst) worst = b;
    }
    return worst;
}
// both divstep kernels on the same low words: out = {delta, u, v, q, r} of the constant-time and of the variable-time form
extern "C" void host_divsteps_both(int32_t delta, uint32_t f0, uint32_t g0, int32_t* out)
{
    posevo::sg_trans a, b;
    out[0] = posevo::sg_divsteps_30(delta, f0, g0, a);
    out[1] = a.u; out[2] = a.v; out[3] = a.q; out[4] = a.r;
    out[5] = posevo::sg_divsteps_30_var(delta, f0, g0, b);
    out[6] = b.u; out[7] = b.v; out[8] = b.q; out[9] = b.r;
}
