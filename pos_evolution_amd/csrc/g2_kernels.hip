// g2_kernels.hip -- BLS12-381 G2 point-sum kernels for gfx950.
//
// bls.Aggregate over BLSSignature points (pe:37, pe:717; prose pe:659, pe:1536): the same accumulate -> workgroup
// tree -> finish structure as g1_kernels.hip, with every point handled by a LANE PAIR (g2.h: one Fp2 component
// per lane).  A 256-lane workgroup therefore covers G2_WG_SLOTS = 128 task slots.
//
//   k_g2_convert     192-B big-endian affine (x.c1 | x.c0 | y.c1 | y.c0) -> Montgomery limbs [x0 x1 y0 y1]
//   k_g2_accumulate  per-pair XYZZ accumulation of k gathered points (mixed adds, 18 products per lane each), then a
//                    compacting pairwise tree over the workgroup's 128 partials staged in LDS
//   k_g2_finish      per group: add the workgroup partials, normalise to canonical affine, store big-endian
//
// Bound: integer VALU, 36 Montgomery products per gathered 192-byte point.
// (Round 4 measured this file's kernels with their Fp products as CALLS, like g1_kernels.hip's -- DESIGN.md 3.8: k_g2_accumulate
// 1.222 ms against 1.218 ms inlined, 2048 x 512 points (gpurun_out/r04g2): its five LDS tree levels of 26 dependent products
// are what it waits for, not instruction fetch.  The inlined form stays: no scratch.)
// The square roots' exponentiation inlined into this file's kernels: as a call it cost the decompression 208-448 bytes of scratch
// per lane (what lives across the call), i.e. 110-235 MB for a chip full of waves, which the runtime hands out per dispatch when
// it exceeds its 140 MiB bound (the kernel took 15 ms, the call around it 35-50: profiles/r06_sig_*).  Inlined: no scratch.
#ifndef POSEVO_POW_INLINE
#define POSEVO_POW_INLINE __forceinline__
#endif
#include "g2.h"
#include "fp_sqrt.h"
#include "kernels.h"

namespace posevo {

// ---------------------------------------------------------------- convert
__global__ void __launch_bounds__(256) k_g2_convert(const uint8_t* __restrict__ be192, uint32_t* __restrict__ mont48,
                                                     uint64_t n)
{
    // one lane per Fp element: element e of point i; wire order c1 first, stored order c0 first
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 4 * n) return;
    const uint64_t i = gid >> 2;
    const uint32_t e = (uint32_t)(gid & 3);             // wire element: 0 x.c1, 1 x.c0, 2 y.c1, 3 y.c0
    const uint32_t dst_e = (e & 2) | ((e & 1) ^ 1);     // stored: 0 x0, 1 x1, 2 y0, 3 y1
    const uint8_t* base = be192 + 192 * i;
    uint32_t* dst = mont48 + 48 * i + 12 * dst_e;
    if (base[0] & 0x40) {  // infinity flag
#pragma unroll
        for (int j = 0; j < 12; ++j) dst[j] = 0;
        return;
    }
    fp v, m;
    fp_load_be48(v, base + 48 * e);  // masks the three flag bits of the leading byte
    if (e != 0) v.l[11] = __builtin_bswap32(reinterpret_cast<const uint32_t*>(base + 48 * e)[0]);  // only x.c1 has flags
    fp_to_mont(m, v);
#pragma unroll
    for (int j = 0; j < 12; ++j) dst[j] = m.l[j];
}

void launch_g2_convert(hipStream_t s, const uint8_t* be192, uint32_t* mont48, uint64_t n)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g2_convert, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, s, be192, mont48, n);
}

// ---------------------------------------------------------------- decompress
// BLSSignature wire format (pe:37, pe:717): 96 bytes = x.c1 | x.c0 big-endian, flag bits in the leading byte (bit 7
// compressed, bit 6 infinity, bit 5 "y is the lexicographically larger root", compared on (c1, c0)).
// One lane per point.  y = sqrt(x^3 + 4(1+u)) in Fp2 by the "complex method" (p = 3 mod 4): s = sqrt(norm) in Fp,
// d = (a0 + s)/2, w = d^((p-3)/4): y = d w + (a1 w / 2) u if d is a residue, -(a1 w / 2) + (d w) u otherwise -- two windowed
// Fp exponentiations (fp_sqrt.h), no inversion, no second attempt: ~1000 Montgomery products per point.  status: 0 ok, 1 malformed encoding, 2 not on the curve.
__device__ __forceinline__ void g2_decompress_one(const uint64_t i, const uint8_t* __restrict__ in96, uint32_t* __restrict__ out_mont48,
                                                  uint8_t* __restrict__ out_be192, int32_t* __restrict__ status)
{
    const uint8_t* src = in96 + 96 * i;
    const uint32_t lead = src[0];
    const bool c_flag = lead & 0x80, inf_flag = lead & 0x40, sign_flag = lead & 0x20;
    fp x1, x0;
    fp_load_be48(x1, src);  // flag bits masked
    fp_load_be48(x0, src + 48);
    x0.l[11] = __builtin_bswap32(reinterpret_cast<const uint32_t*>(src + 48)[0]);  // no flags in x.c0
    int32_t st = 0;
    bool is_inf = false;
    if (!c_flag) st = 1;
    else if (inf_flag) {
        if (sign_flag || !fp_is_zero(x0) || !fp_is_zero(x1)) st = 1;
        is_inf = true;
    } else if (!fp_is_canonical(x0) || !fp_is_canonical(x1)) {
        st = 1;
    }
    fp xm0, xm1, y0, y1;
    fp_set_zero(xm0); fp_set_zero(xm1); fp_set_zero(y0); fp_set_zero(y1);
    if (st == 0 && !is_inf) {
        fp_to_mont(xm0, x0);
        fp_to_mont(xm1, x1);
        fp a0, a1;
        {   // a = x^3 + 4(1+u)
            fp t0, t1, t2, sq0, sq1, four;
            fp_sqr(t0, xm0);
            fp_sqr(t1, xm1);
            fp_mul(t2, xm0, xm1);
            fp_sub(sq0, t0, t1);
            fp_dbl(sq1, t2);
            fp_mul(t0, sq0, xm0);
            fp_mul(t1, sq1, xm1);
            fp_sub(a0, t0, t1);
            fp_mul(t0, sq0, xm1);
            fp_mul(t1, sq1, xm0);
            fp_add(a1, t0, t1);
            fp_set_zero(four);
            four.l[0] = 4;
            fp_to_mont(four, four);
            fp_add(a0, a0, four);
            fp_add(a1, a1, four);
        }
        if (fp_is_zero(a1)) {  // a in Fp: t = a0^((p+1)/4) squares to a0 (y = t) or to -a0 (y = u t: -1 is a non-residue)
            fp t, c;
            fp_sqrt_candidate(t, a0);
            fp_sqr(c, t);
            if (fp_eq(c, a0)) y0 = t;
            else y1 = t;
        } else {
            fp n, t, s, d, w, v, c;
            fp_sqr(n, a0);
            fp_sqr(t, a1);
            fp_add(n, n, t);
            if (!fp_sqrt(s, n)) st = 2;  // the norm of a square is a square in Fp
            else {
                // exactly one of d = (a0 + s)/2 and d' = (a0 - s)/2 = -a1^2 / (4 d) is a square.  ONE exponentiation serves both
                // cases and the division: w = d^((p-3)/4), t = d w, v = a1 w / 2.
                //   d a residue:  t^2 = d, w = 1/t:        y = t + v u       (y0^2 - y1^2 = d - a1^2/(4d) = a0, 2 y0 y1 = a1)
                //   otherwise:    t^2 = -d, w^2 = -1/d:    y = -v + t u      (v^2 + d = d' + d = a0, -2 v t = -a1 d w^2 = a1)
                fp_add(d, a0, s);
                fp_half(d, d);
                fp_pow_pm3d4(w, d);
                fp_mul(t, d, w);
                fp_mul(v, a1, w);
                fp_half(v, v);
                fp_sqr(c, t);
                if (fp_eq(c, d)) { y0 = t; y1 = v; }
                else { fp_neg(y0, v); y1 = t; }
            }
        }
        if (st == 0) {  // belt and braces: y^2 == a
            fp t0, t1, t2;
            fp_sqr(t0, y0);
            fp_sqr(t1, y1);
            fp_sub(t0, t0, t1);
            fp_mul(t2, y0, y1);
            fp_dbl(t2, t2);
            if (!fp_eq(t0, a0) || !fp_eq(t2, a1)) st = 2;
        }
        if (st == 0) {
            const bool larger = fp_is_zero(y1) ? fp_is_larger_half(y0) : fp_is_larger_half(y1);
            if (larger != sign_flag) {
                fp_neg(y0, y0);
                fp_neg(y1, y1);
            }
        }
    }
    status[i] = st;
    const bool blank = st != 0 || is_inf;
    if (out_mont48) {
        uint32_t* d = out_mont48 + 48 * i;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            d[j] = blank ? 0u : xm0.l[j];
            d[12 + j] = blank ? 0u : xm1.l[j];
            d[24 + j] = blank ? 0u : y0.l[j];
            d[36 + j] = blank ? 0u : y1.l[j];
        }
    }
    if (out_be192) {
        uint8_t* o = out_be192 + 192 * i;
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);
        if (blank) {
#pragma unroll
            for (int j = 0; j < 48; ++j) ow[j] = 0;
            if (is_inf && st == 0) o[0] = 0x40;
        } else {
            fp p0, p1;
            fp_from_mont(p0, y0);
            fp_from_mont(p1, y1);
            fp_store_be48(o, x1);
            fp_store_be48(o + 48, x0);
            fp_store_be48(o + 96, p1);
            fp_store_be48(o + 144, p0);
        }
    }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_g2_decompress(const uint8_t* __restrict__ in96, uint64_t n, uint32_t* __restrict__ out_mont48,
                uint8_t* __restrict__ out_be192, int32_t* __restrict__ status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_decompress_one(i, in96, out_mont48, out_be192, status);
}
// Several arrays of signatures in ONE launch (the signature legs of consecutive streaming steps, engine_g1.cpp): a launch is as
// long as one lane's chain of ~970 dependent products whatever its size, so the legs of B steps cost what one cost.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_g2_decompress_batch(const G2DecompressBatch b)
{
    uint32_t k = 0;
    while (k + 1 < b.count && blockIdx.x >= b.first_block[k + 1]) ++k;
    const uint64_t i = (uint64_t)(blockIdx.x - b.first_block[k]) * 64 + threadIdx.x;
    if (i >= b.n[k]) return;
    g2_decompress_one(i, b.in96[k], b.out_mont48[k], nullptr, b.status[k]);
}

void launch_g2_decompress(hipStream_t s, const uint8_t* in96, uint64_t n, uint32_t* out_mont48, uint8_t* out_be192,
                          int32_t* status)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g2_decompress, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, in96, n, out_mont48, out_be192,
                       status);
}
void launch_g2_decompress_batch(hipStream_t s, G2DecompressBatch& b)
{
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < b.count; ++k) {
        b.first_block[k] = blocks;
        blocks += (b.n[k] + 63) / 64;
    }
    b.first_block[b.count] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_g2_decompress_batch, dim3(blocks), dim3(64), 0, s, b);
}

// ---------------------------------------------------------------- accumulate
constexpr int G2_WG = 256;  // lanes; G2_WG_SLOTS (kernels.h) = 128 point slots

__device__ __forceinline__ void ld12(fp& v, const uint32_t* __restrict__ src)
{
    const uint4* s = reinterpret_cast<const uint4*>(src);
    const uint4 a = s[0], b = s[1], c = s[2];
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    v.l[8] = c.x; v.l[9] = c.y; v.l[10] = c.z; v.l[11] = c.w;
}
__device__ __forceinline__ void st12(uint32_t* __restrict__ dst, const fp& v)
{
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    d[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    d[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
// XYZZ partial in global memory: [x0 x1 y0 y1 zz0 zz1 zzz0 zzz1] x 12 words; a lane moves its own halves
__device__ __forceinline__ void g2_global_store(uint32_t* __restrict__ dst, const g2x& p, bool role)
{
    const int o = role ? 12 : 0;
    st12(dst + o, p.x);
    st12(dst + 24 + o, p.y);
    st12(dst + 48 + o, p.zz);
    st12(dst + 72 + o, p.zzz);
}
__device__ __forceinline__ void g2_global_load(g2x& p, const uint32_t* __restrict__ src, bool role)
{
    const int o = role ? 12 : 0;
    ld12(p.x, src + o);
    ld12(p.y, src + 24 + o);
    ld12(p.zz, src + 48 + o);
    ld12(p.zzz, src + 72 + o);
}
// LDS staging, limb-major over lane positions: word k of position L at lds[k * G2_WG + L]; slot s = positions 2s, 2s+1
__device__ __forceinline__ void g2_lds_store(uint32_t* lds, int pos, const g2x& p)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        lds[k * G2_WG + pos] = p.x.l[k];
        lds[(12 + k) * G2_WG + pos] = p.y.l[k];
        lds[(24 + k) * G2_WG + pos] = p.zz.l[k];
        lds[(36 + k) * G2_WG + pos] = p.zzz.l[k];
    }
}
__device__ __forceinline__ void g2_lds_load(g2x& p, const uint32_t* lds, int pos)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        p.x.l[k] = lds[k * G2_WG + pos];
        p.y.l[k] = lds[(12 + k) * G2_WG + pos];
        p.zz.l[k] = lds[(24 + k) * G2_WG + pos];
        p.zzz.l[k] = lds[(36 + k) * G2_WG + pos];
    }
}

__device__ __forceinline__ uint32_t g2_find_group(const G1Group* __restrict__ groups, uint32_t n_groups, uint32_t slot)
{
    uint32_t lo = 0, hi = n_groups;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (groups[mid].slot_base <= slot) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(G2_WG) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g2_accumulate(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ members,
                const G1Group* __restrict__ groups, uint32_t n_groups, uint32_t n_slots,
                uint32_t* __restrict__ wg_partials)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[48 * G2_WG];
    __shared__ uint32_t lds_out[G2_WG_SLOTS];
    __shared__ uint32_t lds_sz[G2_WG_SLOTS];

    const int tid = threadIdx.x;
    const bool role = tid & 1;
    const int ws = tid >> 1;  // slot inside the workgroup
    const uint32_t slot = blockIdx.x * G2_WG_SLOTS + ws;

    g2x acc;
    g2x_set_inf(acc);
    uint32_t my_out = NONE32, my_size = 0;

    if (slot < n_slots) {
        const uint32_t g = g2_find_group(groups, n_groups, slot);
        const G1Group d = groups[g];
        const uint32_t t = slot - d.slot_base;
        const bool wide = d.log2_block > 8;
        const uint32_t block_slots = d.n_tasks == 0 ? 0u
                                   : wide ? ((d.n_tasks + G2_WG_SLOTS - 1) / G2_WG_SLOTS) * G2_WG_SLOTS
                                          : (1u << d.log2_block);
        if (t < block_slots) {
            my_size = wide ? (uint32_t)G2_WG_SLOTS : (1u << d.log2_block);
            my_out = d.out_base + (wide ? t / G2_WG_SLOTS : 0u);
        }
        if (t < d.n_tasks) {
            const uint32_t first = t * d.k;
            const uint32_t count = min(d.k, d.n_members - first);
            fp qx, qy, nx, ny;
            auto fetch = [&](uint32_t j, fp& ox, fp& oy) {
                const uint32_t i = first + j;
                const uint32_t idx = members ? members[d.member_start + i] : d.member_start + i;
                const uint32_t* p = pts + 48ull * idx + (role ? 12 : 0);
                ld12(ox, p);
                ld12(oy, p + 24);
            };
            if (count > 0) fetch(0, qx, qy);
            for (uint32_t j = 0; j < count; ++j) {
                if (j + 1 < count) fetch(j + 1, nx, ny);
                const bool q_inf = pair_and(fp_is_zero(qx) && fp_is_zero(qy));  // (0,0) encodes infinity
                g2x_add_affine(acc, qx, qy, q_inf, role);
                qx = nx; qy = ny;
            }
        }
    }
    // ---- workgroup tree over the 128 slot partials: level with n pairs, slot w < n adds slots 2w and 2w+1 ----
    if (my_size == 1) {
        g2_global_store(wg_partials + (size_t)G2X_WORDS * my_out, acc, role);
        my_size = 0;
    }
    g2_lds_store(lds, tid, acc);
    if (!role) {
        lds_out[ws] = my_out;
        lds_sz[ws] = my_size;
    }
    __syncthreads();
    for (int n = G2_WG_SLOTS / 2; n >= 1; n >>= 1) {
        const bool active = ws < n;
        g2x p1;
        uint32_t out = NONE32, sz = 0;
        bool store_lds = false;
        if (active) {
            sz = lds_sz[2 * ws];
            out = lds_out[2 * ws];
            if (sz >= 2) {
                g2x p2;
                g2_lds_load(p1, lds, 4 * ws + (role ? 1 : 0));
                g2_lds_load(p2, lds, 4 * ws + 2 + (role ? 1 : 0));
                g2x_add(p1, p2, role);
                sz >>= 1;
                if (sz == 1) {
                    g2_global_store(wg_partials + (size_t)G2X_WORDS * out, p1, role);
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
            if (store_lds) g2_lds_store(lds, tid, p1);
            if (!role) {
                lds_out[ws] = out;
                lds_sz[ws] = sz;
            }
        }
        __syncthreads();
    }
}

void launch_g2_accumulate(hipStream_t s, const uint32_t* points_mont48, const uint32_t* members,
                          const G1Group* groups, uint32_t n_groups, uint32_t n_slots, uint32_t* wg_partials96)
{
    if (n_groups == 0 || n_slots == 0) return;
    const unsigned blocks = (n_slots + G2_WG_SLOTS - 1) / G2_WG_SLOTS;
    hipLaunchKernelGGL(k_g2_accumulate, dim3(blocks), dim3(G2_WG), 0, s, points_mont48, members, groups, n_groups,
                       n_slots, wg_partials96);
}

// ---------------------------------------------------------------- finish
__global__ void __launch_bounds__(64)
k_g2_finish(const uint32_t* __restrict__ partials, const G1Group* __restrict__ groups, uint32_t n_groups,
            uint8_t* __restrict__ out_be192)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = lane >> 1;
    const bool role = lane & 1;
    if (g >= n_groups) return;
    const G1Group d = groups[g];
    uint32_t n_parts = d.log2_block > 8 ? (d.n_tasks + G2_WG_SLOTS - 1) / G2_WG_SLOTS : 1u;
    if (d.n_tasks == 0) n_parts = 0;
    g2x acc;
    g2x_set_inf(acc);
    for (uint32_t k = 0; k < n_parts; ++k) {
        g2x q;
        g2_global_load(q, partials + (size_t)G2X_WORDS * (d.out_base + k), role);
        g2x_add(acc, q, role);
    }
    // wire order: x.c1 | x.c0 | y.c1 | y.c0 -- role 1 (c1) writes the leading 48 bytes of each coordinate
    uint8_t* o = out_be192 + 192ull * g;
    uint8_t* ox = o + (role ? 0 : 48);
    uint8_t* oy = o + 96 + (role ? 0 : 48);
    if (g2x_is_inf(acc)) {
        uint32_t* wx = reinterpret_cast<uint32_t*>(ox);
        uint32_t* wy = reinterpret_cast<uint32_t*>(oy);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            wx[j] = 0;
            wy[j] = 0;
        }
        if (role) o[0] = 0x40;
        return;
    }
    fp x, y;
    g2x_to_affine_plain(x, y, acc, role);
    fp_store_be48(ox, x);
    fp_store_be48(oy, y);
}

void launch_g2_finish(hipStream_t s, const uint32_t* partials96, const G1Group* groups, uint32_t n_groups,
                      uint8_t* out_be192)
{
    if (n_groups == 0) return;
    hipLaunchKernelGGL(k_g2_finish, dim3((2 * n_groups + 63) / 64), dim3(64), 0, s, partials96, groups, n_groups,
                       out_be192);
}

// ---------------------------------------------------------------- subgroup check
// Membership of G2 for every decoded point (is_valid_indexed_attestation, pe:736 / pe:976, presumes signatures in G2: a point
// of the curve outside the r-torsion is not a BLSSignature) by the endomorphism test (round 4; M. Scott, "A note on group
// membership tests for G1, G2 and GT on BLS pairing-friendly curves", 2021):   P in G2  <=>  psi(P) = [z] P,   z = the curve's
// parameter -0xd201000000010000, psi = untwist-Frobenius-twist: psi(x, y) = (c_x conj(x), c_y conj(y)) with
// c_x = 1 / (1 + u)^((p - 1) / 3), c_y = 1 / (1 + u)^((p - 1) / 2).  63 doublings + 5 mixed adds (|z| has six bits set) + six
// Fp2 products instead of the 254 doublings + 127 adds of r * P == infinity (rounds 1-3): ~1.6 k instead of ~6.4 k Montgomery
// products per lane.  Lane pair per point, uniform control flow (z is a constant).  The constants and the criterion are
// derived and checked against oracle/g2.py in tests/test_oracle_g2_psi.py (CPU): psi(P) == [z]P on multiples of the generator,
// psi(P) != [z]P on curve points outside the subgroup, exactly where r * P != infinity.
// status[i]: 0 stays 0 when the point is in G2 (or is infinity), becomes 3 otherwise; non-zero entries are left alone.
namespace {
// Montgomery limbs (R = 2^384) of c_x = (0, CX1) and c_y = (CY0, CY1)
__device__ __forceinline__ constexpr uint32_t psi_cx1_limb(int j)
{
    return j == 0 ? 0x867545c3u : j == 1 ? 0x890dc9e4u : j == 2 ? 0x3285a5d5u : j == 3 ? 0x2af32253u
         : j == 4 ? 0x309b7e2cu : j == 5 ? 0x50880866u : j == 6 ? 0x7e881024u : j == 7 ? 0xa20d1b8cu
         : j == 8 ? 0xe2db9068u : j == 9 ? 0x14e4f04fu : j == 10 ? 0x1564853au : 0x14e56d3fu;
}
__device__ __forceinline__ constexpr uint32_t psi_cy0_limb(int j)
{
    return j == 0 ? 0xa55c9ad1u : j == 1 ? 0x3e2f585du : j == 2 ? 0x86c18183u : j == 3 ? 0x4294213du
         : j == 4 ? 0x8b623732u : j == 5 ? 0x382844c8u : j == 6 ? 0x19103e18u : j == 7 ? 0x92ad2afdu
         : j == 8 ? 0xac7cf0b9u : j == 9 ? 0x1d794e4fu : j == 10 ? 0x7d825ec8u : 0x0bd592fcu;
}
__device__ __forceinline__ constexpr uint32_t psi_cy1_limb(int j)
{
    return j == 0 ? 0x5aa30fdau : j == 1 ? 0x7bcfa7a2u : j == 2 ? 0x2a927e7cu : j == 3 ? 0xdc17dec1u
         : j == 4 ? 0x6b4ebef1u : j == 5 ? 0x2f088dd8u : j == 6 ? 0xda74d4a7u : j == 7 ? 0xd1ca2087u
         : j == 8 ? 0x96cebc1du : j == 9 ? 0x2da25966u : j == 10 ? 0xbbfd87d2u : 0x0e2b7eedu;
}
}  // namespace

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g2_subgroup_check(const uint32_t* __restrict__ pts, uint64_t n, int32_t* __restrict__ status)
{
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = lane >> 1;
    const bool role = lane & 1;
    if (i >= n) return;
    if (status[i] != 0) return;  // both lanes of the pair read the same word
    fp px, py;
    const uint32_t* p = pts + 48ull * i + (role ? 12 : 0);
    ld12(px, p);
    ld12(py, p + 24);
    if (pair_and(fp_is_zero(px) && fp_is_zero(py))) return;  // infinity: in every subgroup
    // Q = [|z|] P, left to right over |z| = 0xd201000000010000 (bit 63 is the leading one)
    const uint32_t Z_HI = 0xd2010000u, Z_LO = 0x00010000u;
    g2x acc;
    g2x_set_inf(acc);
    g2x_add_affine(acc, px, py, false, role);
    for (int b = 62; b >= 0; --b) {
        acc = g2x_double(acc, role);
        const uint32_t w = b >= 32 ? Z_HI : Z_LO;
        if ((w >> (b & 31)) & 1u) g2x_add_affine(acc, px, py, false, role);
    }
    // psi(P): conjugate (the c1 lane negates its half), times the constants
    fp xc, yc, cx, cy, nx, ny, X, Y;
    fp_neg(nx, px);
    fp_neg(ny, py);
    fp_select(xc, role, nx, px);
    fp_select(yc, role, ny, py);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        cx.l[j] = role ? psi_cx1_limb(j) : 0u;
        cy.l[j] = role ? psi_cy1_limb(j) : psi_cy0_limb(j);
    }
    f2_mul(X, xc, cx, role);
    f2_mul(Y, yc, cy, role);
    // psi(P) == [z] P = -Q   <=>   Q.X == X ZZ   and   Q.Y + Y ZZZ == 0   (Q = (X/ZZ, Y/ZZZ), not infinity)
    fp t1, t2, d1, d2;
    f2_mul(t1, X, acc.zz, role);
    f2_mul(t2, Y, acc.zzz, role);
    fp_sub(d1, acc.x, t1);
    fp_add(d2, acc.y, t2);
    const bool q_inf = g2x_is_inf(acc);
    const bool same = f2_is_zero(d1) & f2_is_zero(d2);   // both exchanges executed by both lanes
    if ((q_inf || !same) && !role) status[i] = 3;
}

void launch_g2_subgroup_check(hipStream_t s, const uint32_t* points_mont48, uint64_t n, int32_t* status)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g2_subgroup_check, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, s, points_mont48, n, status);
}

// Rows whose status is not 0 become the (0, 0) row = infinity, so that a plain sum leaves them out (pe_aggregate_signatures:
// a signature outside G2 keeps its decoded point until here; undecodable ones are zero rows already).
__global__ void __launch_bounds__(256)
k_g2_mask_bad(uint32_t* __restrict__ points_mont48, uint64_t n, const int32_t* __restrict__ status)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // 12 lanes of 16 bytes per 192-byte row
    const uint64_t i = t / 12;
    if (i >= n || status[i] == 0) return;
    reinterpret_cast<uint4*>(points_mont48)[t] = make_uint4(0, 0, 0, 0);
}
void launch_g2_mask_bad(hipStream_t s, uint32_t* points_mont48, uint64_t n, const int32_t* status)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_g2_mask_bad, dim3((unsigned)((12 * n + 255) / 256)), dim3(256), 0, s, points_mont48, n, status);
}

// ---------------------------------------------------------------- the signature leg of pe_aggregate
// bls.Aggregate (pe:659, pe:714-717) per aggregate: group g's signature = the sum of its members' signature points,
// members listed by input row (member_row[list_start .. + n_atts), the lists the bitfield union runs over).  One lane
// pair per group: an epoch's partial aggregates are a handful per committee, so the run is a few dependent mixed adds
// and one normalisation deep -- latency-sized, on the state-transition stream beside everything else.  Output: the
// 96-byte COMPRESSED BLSSignature (x.c1 | x.c0, flag bits: 0x80 compressed, 0x40 infinity, 0x20 y is the larger root)
// and bad[g] = number of members whose signature did not decode (status != 0; they are left out of the sum -- the caller
// clears PE_ATT_FLAG_SIGNATURE_VALID on the row).
__device__ __forceinline__ void g2_aggregate_rows_body(const uint32_t lane, const uint32_t* __restrict__ pts,
                                                       const int32_t* __restrict__ status, const UnionGroup* __restrict__ ug,
                                                       const uint32_t* __restrict__ member_row, uint32_t n_groups,
                                                       const AttPlan* __restrict__ plan_dev, uint8_t* __restrict__ out96,
                                                       uint32_t* __restrict__ out_bad)
{
    if (plan_dev) n_groups = min(n_groups, plan_dev->n_groups);
    const uint32_t g = lane >> 1;
    const bool role = lane & 1;
    if (g >= n_groups) return;
    const UnionGroup u = ug[g];
    g2x acc;
    g2x_set_inf(acc);
    uint32_t bad = 0;
    for (uint32_t j = 0; j < u.n_atts; ++j) {
        const uint32_t row = member_row[u.list_start + j];
        if (status[row] != 0) { ++bad; continue; }
        fp qx, qy;
        const uint32_t* p = pts + 48ull * row + (role ? 12 : 0);
        ld12(qx, p);
        ld12(qy, p + 24);
        const bool q_inf = pair_and(fp_is_zero(qx) && fp_is_zero(qy));
        g2x_add_affine(acc, qx, qy, q_inf, role);
    }
    if (!role) out_bad[g] = bad;
    uint8_t* o = out96 + 96ull * g;
    uint8_t* ox = o + (role ? 0 : 48);  // wire order: x.c1 | x.c0 -- role 1 (c1) writes the leading 48 bytes
    if (g2x_is_inf(acc)) {
        uint32_t* wx = reinterpret_cast<uint32_t*>(ox);
#pragma unroll
        for (int j = 0; j < 12; ++j) wx[j] = 0;
        if (role) o[0] = 0xC0;
        return;
    }
    fp x, y;
    g2x_to_affine_plain(x, y, acc, role);  // plain residues: this lane's component of x and of y
    // y > -y on (c1, c0): c1 decides unless it is zero
    const bool mine_larger = limbs_gt(y, FP_HALF);
    const bool mine_zero = fp_is_zero(y);
    const bool other_larger = __shfl_xor((int)mine_larger, 1, 64) != 0;
    const bool other_zero = __shfl_xor((int)mine_zero, 1, 64) != 0;
    const bool c1_zero = role ? mine_zero : other_zero;
    const bool c1_larger = role ? mine_larger : other_larger;
    const bool c0_larger = role ? other_larger : mine_larger;
    const bool sign = c1_zero ? c0_larger : c1_larger;
    fp_store_be48(ox, x);
    if (role) o[0] |= (uint8_t)(0x80 | (sign ? 0x20 : 0));
}
// the legs of up to G2_BATCH_MAX steps in one launch (first_block is filled by the launcher): the kernel is one chain of a few
// additions and one normalisation deep -- ~180 us whether it serves one step's groups or eight steps'
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_g2_aggregate_rows(const G2AggregateRowsBatch b)
{
    uint32_t k = 0;
    while (k + 1 < b.count && blockIdx.x >= b.first_block[k + 1]) ++k;
    g2_aggregate_rows_body((blockIdx.x - b.first_block[k]) * 64 + threadIdx.x, b.pts[k], b.status[k], b.ug[k], b.member_row[k],
                           b.n_groups[k], b.plan_dev[k], b.out96[k], b.out_bad[k]);
}

void launch_g2_aggregate_rows(hipStream_t s, G2AggregateRowsBatch& b)
{
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < b.count; ++k) {
        b.first_block[k] = blocks;
        blocks += (2 * b.n_groups[k] + 63) / 64;
    }
    b.first_block[b.count] = blocks;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_g2_aggregate_rows, dim3(blocks), dim3(64), 0, s, b);
}

// ---------------------------------------------------------------- pe_aggregate_signatures: the index list, on the device
// A caller's DEVICE-resident index list cannot be range-checked by the host: this pass copies it into the engine's scratch with
// every entry >= n replaced by 0 and an error word set (the sums then read defined memory and the call fails);
// k_g2_count_bad counts, per group, the members whose signature did not decode -- one wave per group -- so that neither the
// statuses nor the index have to travel to the host for it.
__global__ void __launch_bounds__(256)
k_g2_index_check(const uint32_t* __restrict__ index, uint32_t total, uint32_t n, uint32_t* __restrict__ out_index,
                 uint32_t* __restrict__ err)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    uint32_t v = index[j];
    if (v >= n) { atomicOr(err, 1u); v = 0; }
    out_index[j] = v;
}
__global__ void __launch_bounds__(256)
k_g2_count_bad(const int32_t* __restrict__ status, const uint32_t* __restrict__ index, const G1Group* __restrict__ groups,
               uint32_t n_groups, uint32_t* __restrict__ out_bad)
{
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const uint32_t lane = threadIdx.x & 63, first = groups[g].member_start, cnt = groups[g].n_members;
    uint32_t bad = 0;
    for (uint32_t j = lane; j < cnt; j += 64) bad += status[index ? index[first + j] : first + j] != 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bad += __shfl_xor(bad, off, 64);
    if (lane == 0) out_bad[g] = bad;
}
__global__ void __launch_bounds__(256)
k_g2_status_out(const G2StatusOutBatch b)
{
    const uint32_t seg = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.n[seg]) b.dst_host[seg][i] = b.src[seg][i];
}
void launch_g2_status_out(hipStream_t s, const G2StatusOutBatch& b)
{
    uint32_t most = 0;
    for (uint32_t k = 0; k < b.count; ++k) most = std::max(most, b.n[k]);
    if (b.count == 0 || most == 0) return;
    hipLaunchKernelGGL(k_g2_status_out, dim3((most + 255) / 256, b.count), dim3(256), 0, s, b);
}
void launch_g2_index_check(hipStream_t s, const uint32_t* index, uint32_t total, uint32_t n, uint32_t* out_index, uint32_t* err)
{
    if (total == 0) return;
    hipLaunchKernelGGL(k_g2_index_check, dim3((total + 255) / 256), dim3(256), 0, s, index, total, n, out_index, err);
}
void launch_g2_count_bad(hipStream_t s, const int32_t* status, const uint32_t* index, const G1Group* groups, uint32_t n_groups,
                         uint32_t* out_bad)
{
    if (n_groups == 0) return;
    hipLaunchKernelGGL(k_g2_count_bad, dim3((n_groups + 3) / 4), dim3(256), 0, s, status, index, groups, n_groups, out_bad);
}

}  // namespace posevo
