// Host build of the engine's safegcd inversion (plain C++ path of fp_inv_safegcd.h) for CPU tests.
#include "../../pos_evolution_amd/csrc/fp_inv_safegcd.h"
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
extern "C" long host_modinv_total_batches(const uint32_t* x, int n)
{
    long total = 0;
    uint32_t out[12];
    for (int i = 0; i < n; ++i) total += posevo::sg_modinv(out, x + 12 * i, posevo::SG_PINV30);
    return total;
}
