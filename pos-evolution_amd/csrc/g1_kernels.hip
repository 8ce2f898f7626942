// g1_kernels.hip -- BLS12-381 G1 point-sum kernels for gfx950.
//
// Replaces the point additions of bls.Aggregate / the pubkey sum inside
// bls.FastAggregateVerify (is_valid_indexed_attestation, reference call sites
// pe:736 and pe:976; aggregation prose pe:474, pe:659, pe:715, pe:1536).
//
// Kernels
//   k_g1_convert     96-B big-endian affine -> Montgomery limbs (registry load, once)
//   k_g1_accumulate  HOT: per-lane Jacobian accumulation of k gathered points (mixed
//                    adds), then a compacting pairwise tree over the workgroup's 256
//                    partials staged in LDS (limb-major, conflict-light), one Jacobian
//                    partial out per (group, workgroup)
//   k_g1_finish      per group: add the few workgroup (or rank) partials, normalise
//                    to canonical affine, store big-endian
//
// Bound: integer VALU (about 11 Montgomery products = 6.4k v_mad_u64_u32/v_addc per
// 100 bytes gathered), not HBM and not MFMA -- see DESIGN.md "G1 roofline".
#include "g1.cuh"
#include "kernels.h"

namespace posevo {

// ---------------------------------------------------------------- convert
__global__ void __launch_bounds__(256) k_g1_convert(const uint8_t* __restrict__ be96, uint32_t* __restrict__ mont24,
                                                     uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* src = be96 + 96 * i;
    uint32_t* dst = mont24 + 24 * i;
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

// ---------------------------------------------------------------- accumulate
__device__ __forceinline__ void load_point(fp& x, fp& y, const uint32_t* __restrict__ pts, uint32_t idx)
{
    const uint4* p = reinterpret_cast<const uint4*>(pts + 24ull * idx);  // 96-byte rows are 16-byte aligned
    uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5];
    x.l[0] = v0.x; x.l[1] = v0.y; x.l[2] = v0.z; x.l[3] = v0.w;
    x.l[4] = v1.x; x.l[5] = v1.y; x.l[6] = v1.z; x.l[7] = v1.w;
    x.l[8] = v2.x; x.l[9] = v2.y; x.l[10] = v2.z; x.l[11] = v2.w;
    y.l[0] = v3.x; y.l[1] = v3.y; y.l[2] = v3.z; y.l[3] = v3.w;
    y.l[4] = v4.x; y.l[5] = v4.y; y.l[6] = v4.z; y.l[7] = v4.w;
    y.l[8] = v5.x; y.l[9] = v5.y; y.l[10] = v5.z; y.l[11] = v5.w;
}

// LDS staging of the workgroup's Jacobian partials, limb-major: word k of slot s at lds[k*256 + s].
__device__ __forceinline__ void lds_store_j(uint32_t* lds, int slot, const g1j& p)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        lds[k * G1_WG + slot] = p.x.l[k];
        lds[(12 + k) * G1_WG + slot] = p.y.l[k];
        lds[(24 + k) * G1_WG + slot] = p.z.l[k];
    }
}
__device__ __forceinline__ void lds_load_j(g1j& p, const uint32_t* lds, int slot)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        p.x.l[k] = lds[k * G1_WG + slot];
        p.y.l[k] = lds[(12 + k) * G1_WG + slot];
        p.z.l[k] = lds[(24 + k) * G1_WG + slot];
    }
}
__device__ __forceinline__ void global_store_j(uint32_t* __restrict__ dst, const g1j& p)
{
    uint4* d = reinterpret_cast<uint4*>(dst);  // 144-byte rows are 16-byte aligned
    d[0] = make_uint4(p.x.l[0], p.x.l[1], p.x.l[2], p.x.l[3]);
    d[1] = make_uint4(p.x.l[4], p.x.l[5], p.x.l[6], p.x.l[7]);
    d[2] = make_uint4(p.x.l[8], p.x.l[9], p.x.l[10], p.x.l[11]);
    d[3] = make_uint4(p.y.l[0], p.y.l[1], p.y.l[2], p.y.l[3]);
    d[4] = make_uint4(p.y.l[4], p.y.l[5], p.y.l[6], p.y.l[7]);
    d[5] = make_uint4(p.y.l[8], p.y.l[9], p.y.l[10], p.y.l[11]);
    d[6] = make_uint4(p.z.l[0], p.z.l[1], p.z.l[2], p.z.l[3]);
    d[7] = make_uint4(p.z.l[4], p.z.l[5], p.z.l[6], p.z.l[7]);
    d[8] = make_uint4(p.z.l[8], p.z.l[9], p.z.l[10], p.z.l[11]);
}
__device__ __forceinline__ void global_load_j(g1j& p, const uint32_t* __restrict__ src)
{
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = s[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.x.l[4 * k] = v[k].x; p.x.l[4 * k + 1] = v[k].y; p.x.l[4 * k + 2] = v[k].z; p.x.l[4 * k + 3] = v[k].w;
        p.y.l[4 * k] = v[3 + k].x; p.y.l[4 * k + 1] = v[3 + k].y; p.y.l[4 * k + 2] = v[3 + k].z; p.y.l[4 * k + 3] = v[3 + k].w;
        p.z.l[4 * k] = v[6 + k].x; p.z.l[4 * k + 1] = v[6 + k].y; p.z.l[4 * k + 2] = v[6 + k].z; p.z.l[4 * k + 3] = v[6 + k].w;
    }
}

// Slot -> group: last g with slot_base[g] <= slot (groups are laid out in slot order).
__device__ __forceinline__ uint32_t find_group(const G1Group* __restrict__ groups, uint32_t n_groups, uint32_t slot)
{
    uint32_t lo = 0, hi = n_groups;  // invariant: slot_base[lo] <= slot < slot_base[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (groups[mid].slot_base <= slot) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(G1_WG, 2)
k_g1_accumulate(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ members,
                const uint32_t* __restrict__ bit_arena, const G1Group* __restrict__ groups, uint32_t n_groups,
                uint32_t n_slots, uint32_t* __restrict__ wg_partials)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // 36*256 partial words + 2*256 block info
    uint32_t* lds_out = lds + 36 * G1_WG;   // output slot of the block a partial belongs to
    uint32_t* lds_sz = lds_out + G1_WG;     // current block size (0 = empty / retired)

    const int tid = threadIdx.x;
    const uint32_t slot = blockIdx.x * G1_WG + tid;

    g1j acc;
    g1j_set_inf(acc);
    uint32_t my_out = NONE32, my_size = 0;

    if (slot < n_slots) {
        const uint32_t g = find_group(groups, n_groups, slot);
        const G1Group d = groups[g];
        const uint32_t t = slot - d.slot_base;
        const bool wide = d.log2_block > 8;                  // group spans whole workgroups
        const uint32_t block_slots = d.n_tasks == 0 ? 0u
                                   : wide ? ((d.n_tasks + G1_WG - 1) / G1_WG) * G1_WG : (1u << d.log2_block);
        if (t < block_slots) {  // inside the group's padded block (padding lanes carry infinity)
            my_size = wide ? (uint32_t)G1_WG : (1u << d.log2_block);
            my_out = d.out_base + (wide ? (t >> 8) : 0u);
        }
        if (t < d.n_tasks) {
            const uint32_t first = t * d.k;
            const uint32_t count = min(d.k, d.n_members - first);
            // gather + mixed adds; the next point's loads are issued before the current add
            fp qx, qy, nx, ny;
            bool have = false, nhave = false;
            auto fetch = [&](uint32_t j, fp& ox, fp& oy) -> bool {
                const uint32_t i = first + j;
                if (d.bits_word != NONE32) {
                    const uint32_t w = bit_arena[d.bits_word + (i >> 5)];
                    if (!((w >> (i & 31)) & 1u)) return false;
                }
                const uint32_t idx = members ? members[d.member_start + i] : d.member_start + i;
                load_point(ox, oy, pts, idx);
                return true;
            };
            if (count > 0) have = fetch(0, qx, qy);
            for (uint32_t j = 0; j < count; ++j) {
                nhave = false;
                if (j + 1 < count) nhave = fetch(j + 1, nx, ny);
                if (have) {
                    const bool q_inf = fp_is_zero(qx) && fp_is_zero(qy);  // (0,0) encodes infinity in the table
                    g1j_add_affine(acc, qx, qy, q_inf);
                }
                qx = nx; qy = ny; have = nhave;
            }
        }
    }
    // ---- workgroup tree over the 256 partials (compacting: level L uses the first 128>>L lanes) ----
    if (my_size == 1) {  // single-task group: done
        global_store_j(wg_partials + 36ull * my_out, acc);
        my_size = 0;
    }
    lds_store_j(lds, tid, acc);
    lds_out[tid] = my_out;
    lds_sz[tid] = my_size;
    __syncthreads();
    for (int n = G1_WG / 2; n >= 1; n >>= 1) {  // n = number of pairs at this level
        g1j a;
        uint32_t out = NONE32, sz = 0;
        const bool active = tid < n;
        if (active) {
            sz = lds_sz[2 * tid];
            out = lds_out[2 * tid];
            if (sz >= 2) {
                g1j b;
                lds_load_j(a, lds, 2 * tid);
                lds_load_j(b, lds, 2 * tid + 1);
                g1j_add(a, b);
                sz >>= 1;
                if (sz == 1) {
                    global_store_j(wg_partials + 36ull * out, a);
                    sz = 0;
                }
            } else {
                sz = 0;
            }
        }
        __syncthreads();
        if (active) {
            if (sz) lds_store_j(lds, tid, a);
            lds_out[tid] = out;
            lds_sz[tid] = sz;
        }
        __syncthreads();
    }
}

void launch_g1_accumulate(hipStream_t s, const uint32_t* points_mont24, const uint32_t* members,
                          const uint32_t* bit_arena, const G1Group* groups, uint32_t n_groups, uint32_t n_slots,
                          uint32_t* wg_partials36)
{
    if (n_groups == 0 || n_slots == 0) return;
    const unsigned blocks = (n_slots + G1_WG - 1) / G1_WG;
    const size_t lds_bytes = (36 + 2) * G1_WG * sizeof(uint32_t);
    hipLaunchKernelGGL(k_g1_accumulate, dim3(blocks), dim3(G1_WG), lds_bytes, s, points_mont24, members, bit_arena,
                       groups, n_groups, n_slots, wg_partials36);
}

// ---------------------------------------------------------------- finish
__global__ void __launch_bounds__(64)
k_g1_finish(const uint32_t* __restrict__ partials, const G1Group* __restrict__ groups, uint32_t n_groups,
            uint32_t n_parts_fixed, uint32_t part_stride, uint8_t* __restrict__ out_be96,
            uint32_t* __restrict__ out_jac)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
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
    g1j acc;
    g1j_set_inf(acc);
    for (uint32_t k = 0; k < n_parts; ++k) {
        g1j q;
        global_load_j(q, partials + 36ull * (first + (uint64_t)k * stride));
        g1j_add(acc, q);
    }
    if (out_jac) global_store_j(out_jac + 36ull * g, acc);
    if (!out_be96) return;
    uint8_t* o = out_be96 + 96ull * g;
    uint32_t* ow = reinterpret_cast<uint32_t*>(o);
    if (g1j_is_inf(acc)) {
#pragma unroll
        for (int j = 0; j < 24; ++j) ow[j] = 0;
        o[0] = 0x40;
        return;
    }
    fp zi, zi2, zi3, x, y;
    fp_inv(zi, acc.z);
    fp_sqr(zi2, zi);
    fp_mul(zi3, zi2, zi);
    fp_mul(x, acc.x, zi2);
    fp_mul(y, acc.y, zi3);
    fp_from_mont(x, x);
    fp_from_mont(y, y);
    fp_store_be48(o, x);
    fp_store_be48(o + 48, y);
}

void launch_g1_finish(hipStream_t s, const uint32_t* partials36, const G1Group* groups, uint32_t n_groups,
                      uint32_t n_parts_fixed, uint32_t part_stride, uint8_t* out_be96, uint32_t* out_jac36)
{
    if (n_groups == 0) return;
    hipLaunchKernelGGL(k_g1_finish, dim3((n_groups + 63) / 64), dim3(64), 0, s, partials36, groups, n_groups,
                       n_parts_fixed, part_stride, out_be96, out_jac36);
}

}  // namespace posevo
