// g1_kernels.hip -- BLS12-381 G1 point-sum kernels for gfx950.
//
// Replaces the point additions of bls.Aggregate / the pubkey sum inside
// bls.FastAggregateVerify (is_valid_indexed_attestation, reference call sites
// pe:736 and pe:976; aggregation prose pe:474, pe:659, pe:715, pe:1536).
//
// Kernels
//   k_g1_convert     96-B big-endian affine -> Montgomery limbs (registry load, once)
//   k_g1_table_s29   the registry in the accumulation's field form (14 x 29-bit limbs, fp381_s29.h), once per registry
//   k_g1_accumulate  HOT: per-lane XYZZ accumulation of k gathered points (mixed adds over the S29 form), one
//                    partial per lane slot
//   k_g1_tree        the compacting pairwise tree over a workgroup's 256 partials staged in LDS (limb-major), one XYZZ
//                    partial out per (group, workgroup)
//   k_g1_finish      per group: add the few workgroup (or rank) partials, normalise
//                    to canonical affine, store big-endian
//
// Bound: integer VALU (a mixed add = 8 products + 2 squarings = 3 738 v_mad_[iu]64_[iu]32 per 100 bytes gathered), not
// HBM and not MFMA -- see DESIGN.md "G1 roofline".  Two field forms live here: the accumulation computes in S29 (one
// multiply-add per limb product, no carry instructions: 25 % faster per mixed add); tree, finish, wire formats and key
// validation compute in 12 x 32-bit limbs (fp381.h) -- latency-bound guests whose products are CALLS to one copy of the
// code, so that what they cost the accumulation is not a 64 KB instruction cache full of their unrolled products.
#define POSEVO_FP_MUL_CALLED 1  // every 12 x 32-bit product of this file's kernels is a call (see fp381.h)
#include <algorithm>
#include <hip/hip_ext.h>
#include "g1.h"
#include "g1_s29.h"
#include "fp_sqrt.h"
#include "kernels.h"

namespace posevo {

// ---------------------------------------------------------------- convert
__global__ void __launch_bounds__(256) k_g1_convert(const uint8_t* __restrict__ be96, uint32_t* __restrict__ mont24,
                                                     uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = be96 + 96 * i;
    uint32_t* dst = mont24 + (uint64_t)G1_ROW_WORDS * i;
    if (src[0] & 0x40) {  // infinity flag
#pragma unroll
        for (int j = 0; j < 24; ++j) dst[j] = 0;
        return;
    }
    fp x, y, xm, ym;
    fp_load_be48(x, src);
    fp_load_be48(y, src + 48);
    y.l[11] = __builtin_bswap32(reinterpret_cast<const uint32_t*>(src + 48)[0]);  // y carries no flag bits
    fp_to_mont(xm, x);
    fp_to_mont(ym, y);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        dst[j] = xm.l[j];
        dst[12 + j] = ym.l[j];
    }
}

void launch_g1_convert(hipStream_t s, const uint8_t* be96, uint32_t* mont24, uint64_t n)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g1_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, be96, mont24, n);
}

// ---------------------------------------------------------------- decompress
// BLSPubkey wire format (Validator.pubkey, pe:37): 48 bytes big-endian x with three flag bits in the leading byte
// (bit 7 compressed, bit 6 infinity, bit 5 "y is the lexicographically larger root", i.e. y > (p-1)/2).
// y = (x^3 + 4)^((p+1)/4) (p = 3 mod 4), verified by squaring.  ~620 Montgomery products per key, once per registry
// load.  status: 0 ok, 1 malformed encoding (flag bits / x >= p), 2 x is not the abscissa of a curve point.
// No subgroup check here (KeyValidate's r*P == infinity is a separate scalar multiplication).
__global__ void __launch_bounds__(256)
k_g1_decompress(const uint8_t* __restrict__ in48, uint64_t n, uint32_t* __restrict__ out_mont24,
                uint8_t* __restrict__ out_be96, int32_t* __restrict__ status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = in48 + 48 * i;
    const uint32_t lead = src[0];
    const bool c_flag = lead & 0x80, inf_flag = lead & 0x40, sign_flag = lead & 0x20;
    fp x;
    fp_load_be48(x, src);  // flag bits masked
    int32_t st = 0;
    bool is_inf = false;
    if (!c_flag) st = 1;
    else if (inf_flag) {
        if (sign_flag || !fp_is_zero(x)) st = 1;
        is_inf = true;
    } else if (!fp_is_canonical(x)) {
        st = 1;
    }
    fp xm, ym;
    fp_set_zero(xm);
    fp_set_zero(ym);
    if (st == 0 && !is_inf) {
        fp t, rhs, four, r;
        fp_to_mont(xm, x);
        fp_sqr(t, xm);
        fp_mul(t, t, xm);
        fp_set_zero(four);
        four.l[0] = 4;
        fp_to_mont(four, four);
        fp_add(rhs, t, four);
        if (!fp_sqrt(r, rhs)) st = 2;
        else {
            fp neg;
            fp_neg(neg, r);
            fp_select(ym, fp_is_larger_half(r) != sign_flag, neg, r);
        }
    }
    status[i] = st;
    const bool bad = st != 0;
    if (out_mont24) {
        uint32_t* d = out_mont24 + (uint64_t)G1_ROW_WORDS * i;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            d[j] = (bad || is_inf) ? 0u : xm.l[j];
            d[12 + j] = (bad || is_inf) ? 0u : ym.l[j];
        }
    }
    if (out_be96) {
        uint8_t* o = out_be96 + 96 * i;
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);
        if (bad || is_inf) {
#pragma unroll
            for (int j = 0; j < 24; ++j) ow[j] = 0;
            if (is_inf && !bad) o[0] = 0x40;
        } else {
            fp yp;
            fp_from_mont(yp, ym);
            fp_store_be48(o, x);
            fp_store_be48(o + 48, yp);
        }
    }
}

void launch_g1_decompress(hipStream_t s, const uint8_t* in48, uint64_t n, uint32_t* out_mont24, uint8_t* out_be96,
                          int32_t* status)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g1_decompress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in48, n, out_mont24,
                       out_be96, status);
}

// ---------------------------------------------------------------- KeyValidate
// FastAggregateVerify = KeyValidate each pubkey first (IETF BLS draft 2.5): a valid curve point, not the
// identity, and in the prime-order subgroup: r * P == infinity.  One lane per key: 254 doublings + one mixed add per set
// bit of r (MSB first; the scalar is a constant, so the branch is uniform) ~ 3.5 k Montgomery products per key -- a
// once-per-registry-load pass (1 M keys ~ 65 ms of the chip).  status: 0 ok, 3 not in the subgroup, 4 identity.
__constant__ uint32_t G1_R_ORDER_BE_WORDS[8] = {0x73eda753u, 0x299d7d48u, 0x3339d808u, 0x09a1d805u,
                                                0x53bda402u, 0xfffe5bfeu, 0xffffffffu, 0x00000001u};
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g1_key_validate(const uint32_t* __restrict__ pts, uint64_t n, int32_t* __restrict__ status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* row = pts + (uint64_t)G1_ROW_WORDS * i;
    fp px, py;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        px.l[j] = row[j];
        py.l[j] = row[12 + j];
    }
    if (fp_is_zero(px) && fp_is_zero(py)) {  // (0, 0) encodes infinity in the table
        status[i] = 4;
        return;
    }
    g1x acc;
    acc.x = px;
    acc.y = py;
    fp_set_one(acc.zz);
    fp_set_one(acc.zzz);
    // bit 254 (the top bit of r) is the initial value; bits 253 .. 0 follow
    for (int b = 253; b >= 0; --b) {
        acc = g1x_double(acc);
        const uint32_t w = G1_R_ORDER_BE_WORDS[7 - (b >> 5)];
        if ((w >> (b & 31)) & 1u) g1x_add_affine(acc, px, py, false);
    }
    status[i] = g1x_is_inf(acc) ? 0 : 3;
}

void launch_g1_key_validate(hipStream_t s, const uint32_t* points_mont, uint64_t n, int32_t* status)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g1_key_validate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points_mont, n, status);
}

// ---------------------------------------------------------------- accumulate
__device__ __forceinline__ void load_point(fp& x, fp& y, const uint32_t* __restrict__ pts, uint32_t idx)
{
    const uint4* p = reinterpret_cast<const uint4*>(pts + (uint64_t)G1_ROW_WORDS * idx);  // one 128-byte line per point
    uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5];
    x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w;
    x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
    x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w;
    y.l[0] = v3.x; y.l[1] = v3.y; y.l[2] = v3.z; y.l[3] = v3.w;
    y.l[4] = v4.x; y.l[5] = v4.y; y.l[6] = v4.z; y.l[7] = v4.w;
    y.l[8] = v5.x; y.l[9] = v5.y; y.l[10] = v5.z; y.l[11] = v5.w;
}

// LDS staging of the workgroup's XYZZ partials, limb-major: word k of slot s at lds[k*256 + s]
// (coordinate c occupies words 12c .. 12c+11).
__device__ __forceinline__ void lds_store_fp(uint32_t* lds, int coord, int slot, const fp& v)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) lds[(12 * coord + k) * G1_WG + slot] = v.l[k];
}
__device__ __forceinline__ void lds_load_fp(fp& v, const uint32_t* lds, int coord, int slot)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) v.l[k] = lds[(12 * coord + k) * G1_WG + slot];
}
__device__ __forceinline__ void lds_store_x(uint32_t* lds, int slot, const g1x& p)
{
    lds_store_fp(lds, 0, slot, p.x);
    lds_store_fp(lds, 1, slot, p.y);
    lds_store_fp(lds, 2, slot, p.zz);
    lds_store_fp(lds, 3, slot, p.zzz);
}
__device__ __forceinline__ void lds_load_x(g1x& p, const uint32_t* lds, int slot)
{
    lds_load_fp(p.x, lds, 0, slot);
    lds_load_fp(p.y, lds, 1, slot);
    lds_load_fp(p.zz, lds, 2, slot);
    lds_load_fp(p.zzz, lds, 3, slot);
}
__device__ __forceinline__ void global_store_fp(uint32_t* __restrict__ dst, const fp& v)  // 48 B, 16-byte aligned
{
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    d[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    d[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
__device__ __forceinline__ void global_load_fp(fp& v, const uint32_t* __restrict__ src)
{
    const uint4* s = reinterpret_cast<const uint4*>(src);
    const uint4 a = s[0], b = s[1], c = s[2];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    v.l[8] = c.x; v.l[9] = c.y; v.l[10] = c.z; v.l[11] = c.w;
}
__device__ __forceinline__ void global_store_x(uint32_t* __restrict__ dst, const g1x& p)  // 192-byte rows
{
    global_store_fp(dst, p.x);
    global_store_fp(dst + 12, p.y);
    global_store_fp(dst + 24, p.zz);
    global_store_fp(dst + 36, p.zzz);
}
__device__ __forceinline__ void global_load_x(g1x& p, const uint32_t* __restrict__ src)
{
    global_load_fp(p.x, src);
    global_load_fp(p.y, src + 12);
    global_load_fp(p.zz, src + 24);
    global_load_fp(p.zzz, src + 36);
}

// Slot -> group: last g with slot_base[g] <= slot (groups are laid out in slot order).
__device__ __forceinline__ uint32_t find_group(const G1Group* __restrict__ groups, uint32_t n_groups, uint32_t slot)
{
    uint32_t lo = 0, hi = n_groups;  // invariant: slot_base[lo] <= slot < slot_base[hi]
    // Plans made on the device give every group the same block (k_att_plan), and so do the host's for committees of one
    // size: try the group a uniform layout puts the slot in before searching (2 loads instead of 11 dependent ones:
    // ~10 us at the head of a 170 us kernel).
    {
        const uint32_t lb = groups[0].log2_block;
        const uint32_t g0 = min(lb <= 8 ? slot >> lb : 0u, n_groups - 1);
        if (groups[g0].slot_base <= slot) {
            if (g0 + 1 == n_groups || groups[g0 + 1].slot_base > slot) return g0;
            lo = g0 + 1;
        } else {
            hi = g0;
        }
    }
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (groups[mid].slot_base <= slot) lo = mid; else hi = mid;
    }
    return lo;
}

// tools/g1_phases.hip builds this file with POSEVO_G1_PHASE_TIMING to stamp the phases of one workgroup.
#ifdef POSEVO_G1_PHASE_TIMING
__device__ unsigned long long g1_phase_stamps[16 * 4096];
#define G1_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g1_phase_stamps[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
__device__ unsigned long long g1_wave_info[3 * 4 * 4096];  // per wave: HW_ID | XCC_ID << 32, accumulate start, accumulate end
#define G1_WAVE_BEGIN() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) { \
        unsigned long long* w_ = &g1_wave_info[3 * (blockIdx.x * 4 + (threadIdx.x >> 6))]; \
        w_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) | \
                ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32); \
        w_[1] = wall_clock64(); } } while (0)
#define G1_WAVE_END() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) \
        g1_wave_info[3 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 2] = wall_clock64(); } while (0)
#else
#define G1_WAVE_BEGIN() do { } while (0)
#define G1_WAVE_END() do { } while (0)
#define G1_STAMP(i) do { } while (0)
#endif

// ONE copy of the product and of the squaring for everything that is not the loop body (the hand-over of a finished
// accumulator, the complete add of the rare path): called, arguments and result in registers.
__device__ __noinline__ fq fq_mul_nc(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6,
                                     int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13,
                                     int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t b4, int32_t b5, int32_t b6,
                                     int32_t b7, int32_t b8, int32_t b9, int32_t b10, int32_t b11, int32_t b12, int32_t b13)
{
    const fq a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}};
    const fq b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13}};
    fq r;
    fq_mul(r, a, b);
    return r;
}
__device__ __noinline__ fq fq_sqr_nc(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6,
                                     int32_t a7, int32_t a8, int32_t a9, int32_t a10, int32_t a11, int32_t a12, int32_t a13)
{
    const fq a = {{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}};
    fq r;
    fq_sqr(r, a);
    return r;
}
struct FqCalled {
    __device__ __forceinline__ static void mul(fq& r, const fq& a, const fq& b)
    {
        r = fq_mul_nc(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11],
                      a.l[12], a.l[13], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9],
                      b.l[10], b.l[11], b.l[12], b.l[13]);
    }
    __device__ __forceinline__ static void sqr(fq& r, const fq& a)
    {
        r = fq_sqr_nc(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11],
                      a.l[12], a.l[13]);
    }
};


// ---------------------------------------------------------------- the tree's S29 helpers (round 6)
// The lanes' partials reach the tree in the accumulation's own field form: X | Y | ZZ | ZZZ as 14 limbs each (56 words, X and
// Y carry-passed, ZZ and ZZZ products; all limbs zero = infinity).  Round 4-5 converted every lane's accumulator to the 12 x 32
// form first (four products + four exact reductions per LANE, inside the kernel that paces the step); now the tree adds in
// S29 and converts once per GROUP, at the block's last level.  Same formulas as g1.h's two- and four-lane cooperative adds;
// what differs is the lazy form's bookkeeping: a difference of two balanced values goes into the next product as it is,
// X3 = RR - PPP - 2 Q (three terms) and Y3 (stored) take one carry pass each.
constexpr int G1S_WORDS = 4 * FQ_N;
__device__ __forceinline__ void fq_select(fq& r, bool c, const fq& a, const fq& b)  // r = c ? a : b
{
#pragma unroll
    for (int j = 0; j < FQ_N; ++j) r.l[j] = c ? a.l[j] : b.l[j];
}
__device__ __forceinline__ void fq_xchg(fq& r, const fq& a)  // r = a of the partner lane (lane ^ 1)
{
#pragma unroll
    for (int j = 0; j < FQ_N; ++j) r.l[j] = __shfl_xor(a.l[j], 1, 64);
}
__device__ __forceinline__ void fq_from_lane(fq& r, const fq& a, int src_lane)
{
#pragma unroll
    for (int j = 0; j < FQ_N; ++j) r.l[j] = __shfl(a.l[j], src_lane, 64);
}
__device__ __forceinline__ void lds_store_fq(uint32_t* lds, int coord, int slot, const fq& v)
{
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) lds[(FQ_N * coord + k) * G1_WG + slot] = (uint32_t)v.l[k];
}
__device__ __forceinline__ void lds_load_fq(fq& v, const uint32_t* lds, int coord, int slot)
{
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) v.l[k] = (int32_t)lds[(FQ_N * coord + k) * G1_WG + slot];
}
__device__ __forceinline__ void lds_load_q(g1q& p, const uint32_t* lds, int slot)
{
    lds_load_fq(p.x, lds, 0, slot);
    lds_load_fq(p.y, lds, 1, slot);
    lds_load_fq(p.zz, lds, 2, slot);
    lds_load_fq(p.zzz, lds, 3, slot);
    p.inf = fq_limbs_zero(p.zz);
    p.affine = false;
}
// one coordinate -> its 12 canonical Montgomery words of the 12 x 32 form (what k_g1_finish and the exchange read); zero limbs
// (infinity) give zero words
__device__ __forceinline__ void global_store_fq_as_mont32(uint32_t* __restrict__ dst, const fq& v)
{
    uint32_t w[12];
    fq_to_mont32_via<FqCalled>(w, v);
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(w[0], w[1], w[2], w[3]);
    d[1] = make_uint4(w[4], w[5], w[6], w[7]);
    d[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
// p += q, both XYZZ in S29, every case of the group law (g1_s29.h: g1q_add) over the called products: the rare path of the
// cooperative adds (an infinity operand, P1 = +-P2)
__device__ __noinline__ void g1q_add_full(g1q& p, const g1q& q) { g1q_add<FqCalled>(p, q); }
// two lanes per pair (g1.h: g1x_add_pair).  role 0 owns P1, role 1 owns P2; neither is infinity (caller).  false: P1 = +-P2.
// Outputs: role 0: out_a = X3, out_b = Y3;  role 1: out_a = ZZ3, out_b = ZZZ3.
__device__ __forceinline__ bool g1s_add_pair(fq& out_a, fq& out_b, bool role, const fq& x_own, const fq& y_own,
                                             const fq& zz_own, const fq& zzz_own, const fq& zz_oth, const fq& zzz_oth)
{
    fq m1, m2, o1, o2;
    FqCalled::mul(m1, x_own, zz_oth);   // U1 | U2
    FqCalled::mul(m2, y_own, zzz_oth);  // S1 | S2
    fq_xchg(o1, m1);
    fq_xchg(o2, m2);
    fq U1, S1, P, R;
    {
        fq U2, S2;
        fq_select(U1, role, o1, m1);
        fq_select(U2, role, m1, o1);
        fq_select(S1, role, o2, m2);
        fq_select(S2, role, m2, o2);
        fq_sub(P, U2, U1);
        fq_sub(R, S2, S1);
    }
    if (fq_is_zero_modp(P)) return false;  // identical in both lanes of the pair
    fq a, b, m3, x3, m4, x4, m5;
    fq_select(a, role, R, P);
    FqCalled::sqr(m3, a);    // PP | RR
    fq_xchg(x3, m3);         // role 0 receives RR, role 1 receives PP
    fq_select(a, role, zz_own, P);
    fq_select(b, role, zz_oth, m3);
    FqCalled::mul(m4, a, b);  // PPP | ZZ1*ZZ2
    fq_xchg(x4, m4);          // role 1 receives PPP
    fq_select(a, role, zzz_own, U1);
    fq_select(b, role, zzz_oth, m3);
    FqCalled::mul(m5, a, b);  // Q | ZZZ1*ZZZ2
    fq X3, T, t;
    fq_sub_sub2_norm(X3, x3, m4, m5);  // role 0: RR - PPP - 2 Q   (role 1 computes don't-cares of the same magnitudes)
    fq_sub(T, m5, X3);
    fq m6, m7;
    fq_select(a, role, m4, R);   // ZZ1*ZZ2   | R
    fq_select(b, role, x3, T);   // PP        | Q - X3
    FqCalled::mul(m6, a, b);     // role 0: R*(Q-X3), role 1: ZZ3
    fq_select(a, role, m5, S1);  // ZZZ1*ZZZ2 | S1
    fq_select(b, role, x4, m4);  // PPP (received) | PPP (own)
    FqCalled::mul(m7, a, b);     // role 0: S1*PPP, role 1: ZZZ3
    fq_sub_norm(t, m6, m7);      // role 0: Y3
    fq_select(out_a, role, m6, X3);
    fq_select(out_b, role, m7, t);
    return true;
}
// four lanes per pair (g1.h: g1x_add_quad).  Inputs per lane: a, b = (X1, ZZ2) | (X2, ZZ1) | (Y1, ZZZ2) | (Y2, ZZZ1).
// Outputs: q2: out_a = X3, out_b = Y3;  q1: out_a = ZZ3;  q3: out_a = ZZZ3.  false: P1 = +-P2.
__device__ __forceinline__ bool g1s_add_quad(fq& out_a, fq& out_b, int q, const fq& a, const fq& b)
{
    const int base = (int)(threadIdx.x & 63) & ~3;
    const bool even = (q & 1) == 0;
    fq m1, t, d, send;
    FqCalled::mul(m1, a, b);          // U1 | U2 | S1 | S2
    fq_select(send, even, b, m1);     // q0: ZZ2, q1: U2, q2: ZZZ2, q3: S2
    fq_xchg(t, send);                 // q0: U2,  q1: ZZ2, q2: S2,  q3: ZZZ2
    fq_sub(d, t, m1);                 // q0: P,   q2: R   (odd lanes: unused)
    const int p_zero = __shfl((int)fq_is_zero_modp(d), base, 64);
    if (p_zero) return false;         // identical in the four lanes
    fq x, y, m2;
    fq_select(x, even, d, b);
    fq_select(y, even, d, t);
    FqCalled::mul(m2, x, y);          // PP | A | RR | B
    fq pp, u1;
    fq_from_lane(pp, m2, base);       // PP to everyone
    fq_from_lane(u1, m1, base);       // U1 to everyone (q3 needs it)
    fq m3, sel;
    fq_select(sel, q == 1, m2, u1);
    fq_select(x, q == 0, d, sel);     // q0: P, q1: A, q2 and q3: U1
    fq_select(y, q == 0, m2, pp);     // q0: PP, others: PP
    FqCalled::mul(m3, x, y);          // q0: PPP, q1: ZZ3, q2 and q3: Q = U1 PP
    fq ppp, s1;
    fq_from_lane(ppp, m3, base);      // PPP to everyone
    fq_from_lane(s1, m1, base + 2);   // S1 to everyone (q0 needs it)
    fq X3, T, tmp;
    fq_sub_sub2_norm(X3, m2, ppp, m3);  // q2: X3 = RR - PPP - 2 Q
    fq_sub(T, m3, X3);                  // q2: Q - X3
    fq m4;
    fq_select(sel, q == 2, d, m2);
    fq_select(x, q == 0, s1, sel);    // q0: S1, q2: R, q3: B (q1: A, unused)
    fq_select(sel, q == 2, T, ppp);
    fq_select(y, q == 0, m3, sel);    // q0: PPP, q2: Q - X3, q3: PPP
    FqCalled::mul(m4, x, y);          // q0: T2, q2: T1, q3: ZZZ3
    fq t2;
    fq_from_lane(t2, m4, base);
    fq_sub_norm(tmp, m4, t2);         // q2: Y3
    fq_select(sel, q == 1, m3, m4);
    fq_select(out_a, q == 2, X3, sel);
    out_b = tmp;
    return true;
}

// The sum of one launch runs in THREE kernels since round 2:
//   k_g1_accumulate  per-lane XYZZ accumulation of k gathered points (mixed adds, S29 form since round 4): throughput-
//                    bound, at the multiplier's issue floor; writes one partial per lane slot, limb-major per workgroup
//                    (coalesced), already in the 12 x 32-bit words the tree reads;
//   k_g1_tree        the compacting pairwise tree over each workgroup's 256 partials (LDS, two- and four-lane
//                    cooperative adds): latency-bound, a level is one full add deep;
//   k_g1_finish      adds the few workgroup partials of a wide group and normalises.
// Fused (round 1) the tree of the younger workgroup of every CU ran exposed at the end of the launch (45-70 us at < 50 %
// lane use).  Split, the tree is a short kernel of its own that -- in a stream of pipelined steps -- runs on the finish
// stream BESIDE the next aggregate's accumulation, which leaves it the issue slots a latency-bound kernel needs: the
// accumulation alone is the step's critical path.  Stand-alone (synchronous calls) the two launches cost what the
// fused one did.
// Slot -> (output slot of its block, block size): shared by the two kernels.
__device__ __forceinline__ void g1_slot_block(const G1Group* __restrict__ groups, uint32_t n_groups, uint32_t n_slots,
                                              uint32_t slot, uint32_t& my_out, uint32_t& my_size, G1Group& d, uint32_t& t)
{
    my_out = NONE32;
    my_size = 0;
    t = 0;
    if (slot >= n_slots) return;
    const uint32_t g = find_group(groups, n_groups, slot);
    d = groups[g];
    t = slot - d.slot_base;
    const bool wide = d.log2_block > 8;                  // group spans whole workgroups
    const uint32_t block_slots = d.n_tasks == 0 ? 0u
                               : wide ? ((d.n_tasks + G1_WG - 1) / G1_WG) * G1_WG : (1u << d.log2_block);
    if (t < block_slots) {  // inside the group's padded block (padding lanes carry infinity)
        my_size = wide ? (uint32_t)G1_WG : (1u << d.log2_block);
        my_out = d.out_base + (wide ? (t >> 8) : 0u);
    }
}

// lane partials in HBM: word k of lane `tid` of workgroup `wg` at ((wg * 56 + k) * 256 + tid): a wave's 64 lanes write
// 256 contiguous bytes per word (k_g1_accumulate's hand-over writes them -- S29 limbs, round 6 --, k_g1_tree reads them into LDS)

// Two waves per SIMD: the shape that fits beside ONE wave of the next aggregate's k_g1_accumulate (<= 256 VGPRs) on every SIMD,
// which is how streaming steps run it (round 4).
//
// The body is shared by two kernels that differ in their resource signature only (launch_g1_tree):
//   k_g1_tree       <= 256 registers: two of its workgroups fit a CU's register file; the 84 KB LDS request keeps them apart;
//   k_g1_tree_solo  the same code + eight accumulation registers nobody reads (264 in all): two of ITS workgroups never share
//                   a SIMD whatever the LDS says, while one still fits beside an accumulation wave (232 + 264 <= 512).  That
//                   frees the LDS for the accumulation's own exclusion (launch_g1_accumulate, `exclusive`).
template <bool SOLO>
__device__ __forceinline__ void g1_tree_body(const uint32_t* __restrict__ lane_partials, const G1Group* __restrict__ groups,
                                             uint32_t n_groups, uint32_t n_slots, uint32_t* __restrict__ wg_partials,
                                             const AttPlan* __restrict__ plan_dev)
{
    if (SOLO) asm volatile("" ::: "a7");
    if (plan_dev) {
        n_groups = plan_dev->n_groups;
        n_slots = plan_dev->n_slots;
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // 56*256 partial words + 2*256 block info
    uint32_t* lds_out = lds + G1S_WORDS * G1_WG;  // output slot of the block a partial belongs to
    uint32_t* lds_sz = lds_out + G1_WG;           // current block size (0 = empty / retired)
    const int tid = threadIdx.x;
    // a workgroup takes slabs wg, wg + gridDim.x, ... (launch_g1_tree starts one workgroup per slab)
    for (uint32_t wg = blockIdx.x; wg * G1_WG < n_slots; wg += gridDim.x) {
    const uint32_t slot = wg * G1_WG + tid;
    uint32_t my_out, my_size, t;
    G1Group d{};  // padding lanes (slot >= n_slots) read zeros, not an unwritten struct
    g1_slot_block(groups, n_groups, n_slots, slot, my_out, my_size, d, t);
    if (my_size == 1) my_size = 0;  // written by k_g1_accumulate
    {
        const uint32_t* b = lane_partials + (size_t)wg * G1S_WORDS * G1_WG + tid;
#pragma unroll
        for (int k = 0; k < G1S_WORDS; ++k) lds[k * G1_WG + tid] = b[k * G1_WG];
    }
    lds_out[tid] = my_out;
    lds_sz[tid] = my_size;
    __syncthreads();
    G1_STAMP(2);
#ifdef POSEVO_G1_PHASE_TIMING
    int g1_level = 0;
#endif
    for (int n = G1_WG / 2; n >= 1; n >>= 1) {  // n = number of pairs at this level
        if (n <= G1_WG / 4) {  // four lanes per pair from the second level on (g1s_add_quad: depth 4 instead of 7)
            const int w = tid >> 2, q = tid & 3;
            const bool active = w < n;
            fq out_a, out_b;
            uint32_t out = NONE32, sz = 0;
            bool store_lds = false;
            if (active) {
                sz = lds_sz[2 * w];
                out = lds_out[2 * w];
                if (sz >= 2) {
                    fq a, b;
                    lds_load_fq(a, lds, q >> 1, 2 * w + (q & 1));             // X1 | X2 | Y1 | Y2
                    lds_load_fq(b, lds, 2 + (q >> 1), 2 * w + ((q & 1) ^ 1));  // ZZ2 | ZZ1 | ZZZ2 | ZZZ1
                    // all limbs zero <=> infinity (ZZZ with ZZ), so every lane sees "its" partner point's infinity in b
                    const unsigned long long inf_lanes = __ballot(fq_limbs_zero(b));
                    bool fast = ((inf_lanes >> ((tid & 63) & ~3)) & 0xFull) == 0;
                    if (fast) fast = g1s_add_quad(out_a, out_b, q, a, b);
                    if (!fast) {  // an infinity operand or P1 = +-P2: rare; the four lanes run the complete add
                        g1q p1, p2;
                        lds_load_q(p1, lds, 2 * w);
                        lds_load_q(p2, lds, 2 * w + 1);
                        g1q_add_full(p1, p2);
                        if (p1.inf) g1q_set_inf(p1);  // all limbs zero
                        fq sel;
                        fq_select(sel, q == 1, p1.zz, p1.zzz);
                        fq_select(out_a, q == 2, p1.x, sel);
                        out_b = p1.y;
                    }
                    sz >>= 1;
                    if (sz == 1) {  // block finished: q2 writes X|Y, q1 ZZ, q3 ZZZ of the 192-byte partial (12 x 32 words)
                        uint32_t* dst = wg_partials + (size_t)G1X_WORDS * out;
                        if (q == 2) { global_store_fq_as_mont32(dst, out_a); global_store_fq_as_mont32(dst + 12, out_b); }
                        else if (q == 1) global_store_fq_as_mont32(dst + 24, out_a);
                        else if (q == 3) global_store_fq_as_mont32(dst + 36, out_a);
                        sz = 0;
                    } else {
                        store_lds = true;
                    }
                } else {
                    sz = 0;
                }
            }
            __syncthreads();
            if (active) {
                if (store_lds) {
                    if (q == 2) { lds_store_fq(lds, 0, w, out_a); lds_store_fq(lds, 1, w, out_b); }
                    else if (q == 1) lds_store_fq(lds, 2, w, out_a);
                    else if (q == 3) lds_store_fq(lds, 3, w, out_a);
                }
                if (q == 0) {
                    lds_out[w] = out;
                    lds_sz[w] = sz;
                }
            }
            __syncthreads();
#ifdef POSEVO_G1_PHASE_TIMING
            G1_STAMP(3 + g1_level);
            ++g1_level;
#endif
            continue;
        }
        const int w = tid >> 1;
        const bool role = tid & 1;
        const bool active = w < n;
        fq out_a, out_b;
        uint32_t out = NONE32, sz = 0;
        bool store_lds = false;
        if (active) {
            sz = lds_sz[2 * w];
            out = lds_out[2 * w];
            if (sz >= 2) {
                const int own = 2 * w + (role ? 1 : 0), oth = 2 * w + (role ? 0 : 1);
                fq x_own, y_own, zz_own, zzz_own, zz_oth, zzz_oth;
                lds_load_fq(zz_own, lds, 2, own);
                lds_load_fq(zz_oth, lds, 2, oth);
                const bool own_inf = fq_limbs_zero(zz_own), oth_inf = fq_limbs_zero(zz_oth);
                bool fast = !own_inf && !oth_inf;  // identical in both lanes of the pair
                if (fast) {
                    lds_load_fq(x_own, lds, 0, own);
                    lds_load_fq(y_own, lds, 1, own);
                    lds_load_fq(zzz_own, lds, 3, own);
                    lds_load_fq(zzz_oth, lds, 3, oth);
                    fast = g1s_add_pair(out_a, out_b, role, x_own, y_own, zz_own, zzz_own, zz_oth, zzz_oth);
                }
                if (!fast) {  // an infinity operand or P1 = +-P2: rare; both lanes run the complete single-lane add
                    g1q p1, p2;
                    lds_load_q(p1, lds, 2 * w);
                    lds_load_q(p2, lds, 2 * w + 1);
                    g1q_add_full(p1, p2);
                    if (p1.inf) g1q_set_inf(p1);
                    fq_select(out_a, role, p1.zz, p1.x);
                    fq_select(out_b, role, p1.zzz, p1.y);
                }
                sz >>= 1;
                if (sz == 1) {  // block finished: role 0 writes X|Y, role 1 writes ZZ|ZZZ of the 192-byte partial
                    uint32_t* dst = wg_partials + (size_t)G1X_WORDS * out + (role ? 24 : 0);
                    global_store_fq_as_mont32(dst, out_a);
                    global_store_fq_as_mont32(dst + 12, out_b);
                    sz = 0;
                } else {
                    store_lds = true;
                }
            } else {
                sz = 0;
            }
        }
        __syncthreads();
        if (active) {
            if (store_lds) {
                lds_store_fq(lds, role ? 2 : 0, w, out_a);
                lds_store_fq(lds, role ? 3 : 1, w, out_b);
            }
            if (!role) {
                lds_out[w] = out;
                lds_sz[w] = sz;
            }
        }
        __syncthreads();
#ifdef POSEVO_G1_PHASE_TIMING
        G1_STAMP(3 + g1_level);
        ++g1_level;
#endif
    }
    }  // slabs (the level loop ends behind a barrier: the next slab may overwrite the LDS)
}

__global__ void __launch_bounds__(G1_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g1_tree(const uint32_t* __restrict__ lane_partials, const G1Group* __restrict__ groups, uint32_t n_groups,
          uint32_t n_slots, uint32_t* __restrict__ wg_partials, const AttPlan* __restrict__ plan_dev)
{
    g1_tree_body<false>(lane_partials, groups, n_groups, n_slots, wg_partials, plan_dev);
}
__global__ void __launch_bounds__(G1_WG) __attribute__((amdgpu_waves_per_eu(1, 1), amdgpu_num_vgpr(256)))
k_g1_tree_solo(const uint32_t* __restrict__ lane_partials, const G1Group* __restrict__ groups, uint32_t n_groups,
               uint32_t n_slots, uint32_t* __restrict__ wg_partials, const AttPlan* __restrict__ plan_dev)
{
    g1_tree_body<true>(lane_partials, groups, n_groups, n_slots, wg_partials, plan_dev);
}

// ---------------------------------------------------------------- the accumulation (S29 field form)
// (fp381_s29.h / g1_s29.h: 14 signed limbs of 29 bits, one v_mad_i64_i32 per limb product, no carry instructions.)
// Tree and finish read 12 x 32-bit words: a lane converts its finished accumulator (four products + four exact
// reductions per lane).  Round 3 kept a 12 x 32-bit accumulation kernel beside this one (inline-asm columns of
// v_mad_u64_u32 + v_addc_co_u32: 288 multiply-adds AND 288 carry adds per product); measured back to back on one box by
// tools/accbench.hip at 1 M points it took 183 us where this kernel takes 158 (170 at one wave per SIMD, where the other
// took 219) -- identical lane partials, word for word -- and it was deleted.
// The registry table of this form: one 128-byte row per validator like the 32-bit table -- x limbs in words 0..13,
// y limbs in words 14..27 (canonical, Montgomery constant 2^406), word 28 = 1 when the row holds a point.
__global__ void __launch_bounds__(256)
k_g1_table_s29(const uint32_t* __restrict__ pts32, uint32_t* __restrict__ pts29, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = pts32 + (uint64_t)G1_ROW_WORDS * i;
    uint32_t w[24], any = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
    uint32_t* dst = pts29 + (uint64_t)G1_ROW_WORDS * i;
    fq x, y;
    fq_set_zero(x);
    fq_set_zero(y);
    if (any) {
        fq_from_mont32(x, w);
        fq_from_mont32(y, w + 12);
    }
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) { dst[k] = (uint32_t)x.l[k]; dst[FQ_N + k] = (uint32_t)y.l[k]; }
    dst[28] = any ? 1u : 0u;
    dst[29] = dst[30] = dst[31] = 0;
}

__device__ __forceinline__ bool load_point_s29(fq& x, fq& y, const uint32_t* __restrict__ pts29, uint32_t idx)
{
    const uint4* p = reinterpret_cast<const uint4*>(pts29 + (uint64_t)G1_ROW_WORDS * idx);  // one 128-byte line per point
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5], v6 = p[6], v7 = p[7];
    x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w;
    x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
    x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w;
    x.l[12] = v3.x; x.l[13] = v3.y;
    y.l[0] = v3.z; y.l[1] = v3.w;
    y.l[2] = v4.x; y.l[3] = v4.y; y.l[4] = v4.z; y.l[5] = v4.w;
    y.l[6] = v5.x; y.l[7] = v5.y; y.l[8] = v5.z; y.l[9] = v5.w;
    y.l[10] = v6.x; y.l[11] = v6.y; y.l[12] = v6.z; y.l[13] = v6.w;
    return v7.x != 0;  // the row holds a point
}

// The accumulation.  What shapes it (round 4, tools/icbench.hip): the instruction cache.  A lane's loop holds ONE
// straight-line body -- the general mixed add, g1q_madd_fast: eight products + two squarings, ~38 KB -- and nothing else:
//   * a lane's first point becomes the accumulator (x, y, 1, 1) as it is, and its first add is a general one (ten products
//     instead of the six of an affine + affine add: +5 % products, against a second 27 KB body that every wave fetched
//     once per run of eight);
//   * the same-x cases (P + P, P - P) are detected by a one-multiply filter and NOT handled in the loop: a lane that saw
//     one redoes its whole run afterwards with the complete add (g1q_add_affine over CALLED products: a few hundred bytes
//     of code that is fetched only then) -- exact results, and structured keys ((i + 1) G) take that path all the time;
//   * the hand-over to the 48 words k_g1_tree reads (four products + four exact reductions per lane) goes through the same
//     called product, one coordinate per iteration of a rolled loop.
// Round 3's version of this kernel (three inlined bodies + doubling twice + an unrolled hand-over: 214 KB of code) ran its
// 1 M-point launch in 0.273 ms where the 12 x 32-bit kernel took 0.226 -- with a product that is 14 % and a mixed add that
// is 25 % FASTER in a loop that fits the cache (tools/fpbench29).
__global__ void __launch_bounds__(G1_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g1_accumulate(const uint32_t* __restrict__ pts29, const uint32_t* __restrict__ members,
                    const uint32_t* __restrict__ bit_arena, const G1Group* __restrict__ groups, uint32_t n_groups,
                    uint32_t n_slots, uint32_t* __restrict__ lane_partials, uint32_t* __restrict__ wg_partials,
                    const AttPlan* __restrict__ plan_dev, const uint32_t* __restrict__ members1,
                    unsigned long long* __restrict__ clock_rec)
{
    const int tid = threadIdx.x;
    if (plan_dev) {
        n_groups = plan_dev->n_groups;
        n_slots = plan_dev->n_slots;
        if (blockIdx.x * G1_WG >= n_slots) return;
    }
    // the clock this launch ran at (pe_profile_accumulate_mhz): workgroup 0 counts shader cycles against the fixed 100 MHz counter
    unsigned long long rec_w0 = 0, rec_c0 = 0;
    if (clock_rec && blockIdx.x == 0) {
        rec_w0 = wall_clock64();
        rec_c0 = clock64();
    }
    const uint32_t slot = blockIdx.x * G1_WG + tid;
    G1_STAMP(0);
    G1_WAVE_BEGIN();
#ifdef POSEVO_ACC_PRIO_LEVEL   // tools/build_variant.sh accprio -DPOSEVO_ACC_PRIO_LEVEL=1: an experiment for the signed step
    __builtin_amdgcn_s_setprio(POSEVO_ACC_PRIO_LEVEL);
#endif
    g1q acc;
    g1q_set_inf(acc);
    bool exc = false;
    uint32_t my_out, my_size, t;
    G1Group d{};  // padding lanes (slot >= n_slots) read zeros, not an unwritten struct
    g1_slot_block(groups, n_groups, n_slots, slot, my_out, my_size, d, t);
    uint32_t first = 0, count = 0;
    if (slot < n_slots && t < d.n_tasks) {
        const uint32_t gk = d.k & 0x7FFFFFFFu;
        if (d.k >> 31) members = members1;
        first = t * gk;
        count = min(gk, d.n_members - first);
    }
    // Two loads deep: while point j is added, the row of point j + 1 is in flight (its index arrived an iteration ago) and
    // so are the index and the bit word of point j + 2 -- the wave never waits for an address it needs at once.  (One
    // deep -- index, wait, row -- every iteration stalled for two dependent memory round trips IN FRONT of its add: the
    // kernel ran 27 % below the rate of the same adds in a loop without loads, tools/accbench.hip.)
    const bool gated = d.bits_word != NONE32;
    auto want = [&](uint32_t j, uint32_t& idx) -> bool {  // member j of the run: its table row, whether its bit is set
        const uint32_t i = first + j;
        idx = members ? members[d.member_start + i] : d.member_start + i;
        if (!gated) return true;
        return (bit_arena[d.bits_word + (i >> 5)] >> (i & 31)) & 1u;
    };
    auto fetch = [&](uint32_t j, fq& ox, fq& oy) -> bool {  // complete, for the rare path
        uint32_t idx;
        if (!want(j, idx)) return false;
        return load_point_s29(ox, oy, pts29, idx);
    };
    {
        fq qx, qy, nx, ny;
        bool have = false, nhave = false, w1 = false, w2 = false;
        uint32_t i1 = 0, i2 = 0;
        if (count > 0) {
            uint32_t i0;
            if (want(0, i0)) have = load_point_s29(qx, qy, pts29, i0);
        }
        if (count > 1) w1 = want(1, i1);
        for (uint32_t j = 0; j < count; ++j) {
            w2 = false;
            if (j + 2 < count) w2 = want(j + 2, i2);
            nhave = false;
            if (w1) nhave = load_point_s29(nx, ny, pts29, i1);
            if (have) {
                if (acc.inf) g1q_set_first(acc, qx, qy);
                else g1q_madd_fast(acc, qx, qy, exc);
            }
            qx = nx; qy = ny; have = nhave;
            i1 = i2; w1 = w2;
        }
    }
    G1_STAMP(1);
    G1_WAVE_END();
    if (exc) {  // rare: the run again, every case of the group law (rows come from L2 this time)
        g1q_set_inf(acc);
        for (uint32_t j = 0; j < count; ++j) {
            fq qx, qy;
            if (fetch(j, qx, qy)) g1q_add_affine<FqCalled>(acc, qx, qy, false);
        }
    }
    // hand-over: the accumulator as it is -- X | Y | ZZ | ZZZ, 14 limbs each, limb-major per workgroup; all limbs zero for
    // infinity.  (Rounds 4-5 converted every lane's accumulator to the 12 x 32 words here: four products and four exact
    // reductions per lane inside the kernel that paces the step; k_g1_tree adds in S29 now and converts once per group.)
    uint32_t* b = lane_partials + (size_t)blockIdx.x * G1S_WORDS * G1_WG + tid;
    const bool inf = acc.inf;
#pragma unroll
    for (int k = 0; k < FQ_N; ++k) {
        b[k * G1_WG] = inf ? 0u : (uint32_t)acc.x.l[k];
        b[(FQ_N + k) * G1_WG] = inf ? 0u : (uint32_t)acc.y.l[k];
        b[(2 * FQ_N + k) * G1_WG] = inf ? 0u : (uint32_t)acc.zz.l[k];
        b[(3 * FQ_N + k) * G1_WG] = inf ? 0u : (uint32_t)acc.zzz.l[k];
    }
    if (my_size == 1) {  // a group of one task: done here, in the words k_g1_finish reads (rare: committees of <= k members)
        uint32_t* dst = wg_partials + (size_t)G1X_WORDS * my_out;
#pragma nounroll
        for (int c = 0; c < 4; ++c) {
            uint32_t w[12];
            fq_to_mont32_via<FqCalled>(w, acc.x);
#pragma unroll
            for (int k = 0; k < 12; ++k) dst[12 * c + k] = inf ? 0u : w[k];
            acc.x = acc.y;
            acc.y = acc.zz;
            acc.zz = acc.zzz;
        }
    }
    if (clock_rec && blockIdx.x == 0 && tid == 0) {
        clock_rec[0] = wall_clock64() - rec_w0;
        clock_rec[1] = clock64() - rec_c0;
    }
}

void launch_g1_table_s29(hipStream_t s, const uint32_t* points_mont24, uint32_t* points_s29, uint64_t n)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g1_table_s29, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points_mont24, points_s29, n);
}
void launch_g1_accumulate(hipStream_t s, const uint32_t* points_s29, const uint32_t* members,
                              const uint32_t* bit_arena, const G1Group* groups, uint32_t n_groups, uint32_t n_slots,
                              uint32_t* lane_partials, uint32_t* wg_partials48, const AttPlan* plan_dev,
                              const uint32_t* members1, int exclusive, unsigned long long* clock_rec)
{
    if (n_groups == 0 || n_slots == 0) return;
    const unsigned blocks = (n_slots + G1_WG - 1) / G1_WG;
    // exclusive: the kernel uses no LDS; it ASKS for more than half of a CU's 160 KB so that no CU ever holds two of its
    // workgroups -- of one launch or of two consecutive ones.  Its 232 registers per wave allow two waves per SIMD, and the
    // dispatcher does double workgroups up on a CU whenever it finds the others busy for a moment (k_g1_tree of the previous
    // step arriving in the same microsecond; the first CUs to retire a predecessor's workgroup): those eight waves then run
    // at half speed for the whole launch, 330-370 us instead of 205-240 in up to six steps of twenty
    // (profiles/r05_engine_timeline_cold20_before_exclusive.txt).  78 KB stay for the guests: k_g1_tree_solo (58 KB), the fork-choice
    // tree up to 4096 blocks (66 KB), the vote histograms (32 KB each).
    size_t lds_bytes = 0;
    if (exclusive) {
        lds_bytes = G1_ACC_EXCLUSIVE_LDS;
        if (first_use_on_this_device<82>())
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_g1_accumulate), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)G1_ACC_EXCLUSIVE_LDS);
    }
    hipLaunchKernelGGL(k_g1_accumulate, dim3(blocks), dim3(G1_WG), lds_bytes, s, points_s29, members, bit_arena,
                       groups, n_groups, n_slots, lane_partials, wg_partials48, plan_dev, members1, clock_rec);
}

// one_per_cu (the tree of a streaming step, on its own stream): ask for 84 KB of LDS instead of the 58 KB the kernel uses,
// so that a CU (160 KB) never holds two of its workgroups.  Two reasons, one per round:
//  * round 2: the tree of step N runs when the accumulation of step N retires -- which is when the fork-choice kernels of step
//    N+1 arrive, and k_tree's single workgroup (82 KB of LDS at 4096 blocks) found no CU with room until this kernel had
//    drained (+60 us on every get_head, profiles/r02_timeline_tree_collision.txt).  With the accumulation launched behind
//    k_tree the two now rarely meet; k_tree measured 22-28 us in every step of the runs below.
//  * round 4: the accumulation runs ONE 4-wave workgroup per CU (232 registers per wave) and this kernel's waves take 256:
//    two tree workgroups fill a CU's registers, the dispatcher skips that CU and puts a second accumulate workgroup on
//    another one, whose waves then run at half speed -- 330-360 us instead of 205-240 in up to five steps of twenty
//    (tools/engine_timeline.py --cold 20, 77 KB: two per CU fit); with at most one tree workgroup per CU every CU keeps room
//    for its accumulate workgroup: 0-1 such steps in twenty over four runs, 291-297 us per step instead of 289-323.
//    (At most 128 tree workgroups of two slabs each -- half of the CUs free of them -- was tried first: the tree took
//    twice as long beside the accumulation and stretched it more.)
// The accumulation has the same problem with ITSELF when it is queued behind its predecessor on the side stream: its
// workgroups are dispatched as CUs free up and the first CUs to finish take two; hence its launch behind the step's k_tree
// (engine_internal.h, g1_chain_idle).
//  * round 5 (`solo`, the default of streaming steps): the padding moved to the ACCUMULATION, which asks for 82 KB it never
//    touches (launch_g1_accumulate, `exclusive`) -- that rules its doubling-up out whatever arrives when, which the 84 KB here
//    only made rarer (6 steps of 20 at 340-370 us on one box, profiles/r05_engine_timeline_cold20_before_exclusive.txt); this kernel then
//    keeps its workgroups apart by registers instead (k_g1_tree_solo: 264 + 264 > 512) and asks for the 58 KB it uses.
void launch_g1_tree(hipStream_t s, const uint32_t* lane_partials, const G1Group* groups, uint32_t n_groups,
                    uint32_t n_slots, uint32_t* wg_partials48, int one_per_cu, const AttPlan* plan_dev, int solo)
{
    if (n_groups == 0 || n_slots == 0) return;
    const unsigned blocks = (n_slots + G1_WG - 1) / G1_WG;
    size_t lds_bytes = (G1S_WORDS + 2) * G1_WG * sizeof(uint32_t);
    if (solo) {
        // beside an accumulation that asks for G1_ACC_EXCLUSIVE_LDS: one workgroup per CU by registers (264), the LDS request
        // is what the kernel uses (58 KB)
        hipLaunchKernelGGL(k_g1_tree_solo, dim3(blocks), dim3(G1_WG), lds_bytes, s, lane_partials, groups, n_groups, n_slots,
                           wg_partials48, plan_dev);
        return;
    }
    if (one_per_cu) {
        constexpr size_t padded = 84 * 1024;
        if (first_use_on_this_device<84>())
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_g1_tree), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)padded);
        lds_bytes = padded;
    }
    hipLaunchKernelGGL(k_g1_tree, dim3(blocks), dim3(G1_WG), lds_bytes, s, lane_partials, groups, n_groups, n_slots,
                       wg_partials48, plan_dev);
}

// ---------------------------------------------------------------- finish
__global__ void __launch_bounds__(64)
k_g1_finish(const uint32_t* __restrict__ partials, const G1Group* __restrict__ groups, uint32_t n_groups,
            uint32_t n_parts_fixed, uint32_t part_stride, uint8_t* __restrict__ out_be96,
            uint32_t* __restrict__ out_jac, const AttPlan* __restrict__ plan_dev)
{
    // the launch's n_groups BOUNDS the device-resident count: the output block (96 B per group) and, across ranks, the
    // gathered partials' stride are sized by it (an aggregate that formed more groups than pe_dist_set_max_groups
    // allows wrote past both; the completion reports PE_ERR_CAPACITY)
    if (plan_dev) n_groups = min(n_groups, plan_dev->n_groups);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    // 32 waves of one long dependent chain each (the inversion).  In a stream of pipelined steps they run beside the
    // next aggregate's k_g1_accumulate and stretch from 90 to ~360 us (profiles/r02_timeline.txt): every instruction
    // of the chain queues behind the multi-cycle v_mad_u64_u32 of two older waves on its SIMD.  s_setprio(3) here was
    // measured and changes nothing (the wait is for the ALU, not for the arbiter); the pipeline's lag depth of two
    // absorbs it (the finish stream keeps up as long as a launch stays under the step period).
    uint32_t first, n_parts, stride;
    if (groups) {  // single-GPU: the group's workgroup partials are consecutive
        const G1Group d = groups[g];
        first = d.out_base;
        n_parts = d.log2_block > 8 ? (d.n_tasks + G1_WG - 1) / G1_WG : 1u;
        if (d.n_tasks == 0) n_parts = 0;
        stride = 1;
    } else {       // multi-GPU: rank r's partial of group g at r*part_stride + g
        first = g;
        n_parts = n_parts_fixed;
        stride = part_stride;
    }
    g1x acc;
    g1x_set_inf(acc);
    for (uint32_t k = 0; k < n_parts; ++k) {
        g1x q;
        global_load_x(q, partials + (size_t)G1X_WORDS * (first + (uint64_t)k * stride));
        g1x_add(acc, q);
    }
    if (out_jac) global_store_x(out_jac + (size_t)G1X_WORDS * g, acc);
    if (!out_be96) return;
    uint8_t* o = out_be96 + 96ull * g;
    uint32_t* ow = reinterpret_cast<uint32_t*>(o);
    if (g1x_is_inf(acc)) {
#pragma unroll
        for (int j = 0; j < 24; ++j) ow[j] = 0;
        o[0] = 0x40;
        return;
    }
    fp x, y;
    g1x_to_affine_plain(x, y, acc);
    fp_store_be48(o, x);
    fp_store_be48(o + 48, y);
}

void launch_g1_finish(hipStream_t s, const uint32_t* partials48, const G1Group* groups, uint32_t n_groups,
                      uint32_t n_parts_fixed, uint32_t part_stride, uint8_t* out_be96, uint32_t* out_xyzz48,
                      const AttPlan* plan_dev)
{
    if (n_groups == 0) return;
    hipLaunchKernelGGL(k_g1_finish, dim3((n_groups + 63) / 64), dim3(64), 0, s, partials48, groups, n_groups,
                       n_parts_fixed, part_stride, out_be96, out_xyzz48, plan_dev);
}

}  // namespace posevo
