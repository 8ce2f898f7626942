// att_kernels.hip -- attestation rows resident in device memory: what the host does per row in engine_attest.cpp,
// done on the device so that the host only enqueues launches (the host paced the step).
//
//   k_att_ingest          one lane per input row: hash the 128-byte AttestationData (pe:689-697) + len(aggregation_bits),
//                         insert into an open-addressing table in HBM; a slot's value is the SMALLEST row index of its
//                         class (atomicMin), i.e. the row of first appearance -- the grouping rule of pe_aggregate
//                         ("attestations with identical AttestationData and n_bits form one group, groups ordered by
//                         first appearance").
//   k_att_plan            one workgroup: group ids by a prefix count over the representatives, committee resolution
//                         (get_beacon_committee's index arithmetic, A.6) against the store's current / previous
//                         epoch tables, offsets of the unions (prefix sums), member counts, the G1 plan, the
//                         rows-per-committee lists the handlers walk.  Writes the very descriptor arrays the host
//                         path uploads (UnionGroup, G1Group), so k_bits_union / k_g1_* run unchanged.
//   k_att_members         one lane per input row: group_of[], the group's member list, AND of the signature verdicts,
//                         the output rows; clears the hash table for the next call.
//   k_att_validate_fc     validate_on_attestation (A.4, called at pe:970) + is_valid_indexed_attestation's structural
//                         part per group: root -> block lookups in a device hash table, the LMD/FFG consistency walk
//                         (at most SLOTS_PER_EPOCH parent steps once the two slot checks passed), one AttRow + one
//                         pe_att_status per group.
//   k_att_validate_state  the asserts of process_attestation (pe:724-730) and
//                         get_attestation_participation_flag_indices (A.9) per group.
//   k_lmd_vm_tables       update_latest_messages (pe:1435-1441), validator-major, both candidate tables in one launch.
//   k_participation_tables the flag loop of pe:745-749, one wave per committee, rows of a committee in batch order.
//
// Integer / byte work, latency- and HBM-bound: no MFMA.
#include <algorithm>
#include <cstdlib>
#include "kernels.h"

namespace posevo {

namespace {

constexpr uint32_t VAL_EQUIVOCATING_BIT = 0x04u;
constexpr uint32_t FLAG_SIG_VALID = 0x1u, FLAG_FROM_BLOCK = 0x2u, FLAG_OVERLAPPING = 0x4u;  // PE_ATT_FLAG_* of include/posevo.h
// pe_att_status values (include/posevo.h)
constexpr int32_t ST_OK = 0, ST_EPOCH_TIME = 1, ST_EPOCH_SLOT = 2, ST_UNKNOWN_TARGET = 3, ST_UNKNOWN_BLOCK = 4,
                  ST_BLOCK_AFTER_SLOT = 5, ST_TARGET_NOT_ANCESTOR = 6, ST_SLOT_NOT_PAST = 7, ST_NO_TABLE = 8,
                  ST_INDEX_RANGE = 9, ST_BITS_LENGTH = 10, ST_EMPTY = 11, ST_BAD_SIGNATURE = 12, ST_INCLUSION = 13,
                  ST_SOURCE = 14;
// -pe_status values reported through AttPlan::error
constexpr uint32_t ERR_INVALID_ARG = 1, ERR_CAPACITY = 10, ERR_NO_COMMITTEES = 11;

// a pe_attestation is 9 x 16 bytes: [0] slot, index  [1-2] beacon_block_root  [3] source_epoch, source_root[0:8]
// [4] source_root[8:24]  [5] source_root[24:32], target_epoch  [6-7] target_root  [8] bits_offset, n_bits, flags, reserved
struct Row9 { uint4 q[9]; };
__device__ __forceinline__ void load_row(Row9& r, const uint4* __restrict__ rows, uint32_t i)
{
    const uint4* p = rows + (size_t)9 * i;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.q[k] = p[k];
}
__device__ __forceinline__ unsigned long long u64_of(uint32_t lo, uint32_t hi) { return ((unsigned long long)hi << 32) | lo; }
__device__ __forceinline__ unsigned long long row_slot(const Row9& r) { return u64_of(r.q[0].x, r.q[0].y); }
__device__ __forceinline__ unsigned long long row_index(const Row9& r) { return u64_of(r.q[0].z, r.q[0].w); }
__device__ __forceinline__ unsigned long long row_source_epoch(const Row9& r) { return u64_of(r.q[3].x, r.q[3].y); }
__device__ __forceinline__ unsigned long long row_target_epoch(const Row9& r) { return u64_of(r.q[5].z, r.q[5].w); }

__device__ __forceinline__ uint32_t mix32(uint32_t h, uint32_t v)
{
    h ^= v;
    h *= 0x9E3779B1u;
    return h ^ (h >> 15);
}
__device__ __forceinline__ uint32_t att_hash(const Row9& r)
{
    uint32_t h = 0x85EBCA6Bu ^ r.q[8].y;  // n_bits is part of the key
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h = mix32(h, r.q[k].x);
        h = mix32(h, r.q[k].y);
        h = mix32(h, r.q[k].z);
        h = mix32(h, r.q[k].w);
    }
    return h;
}
__device__ __forceinline__ bool same4(const uint4& a, const uint4& b)
{
    return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) == 0;
}

}  // namespace

// ------------------------------------------------------------------ ingest: hash + insert
// (Bodies take their block index and grid size as arguments: the same code runs as a kernel of its own and as one block
// range of a paired launch, pair_kernels.hip.)
namespace {
__device__ __forceinline__ void att_ingest_body(const uint32_t bid, const uint32_t nblk, const IngestArgs& a)
{
    const uint4* __restrict__ rows = static_cast<const uint4*>(a.rows);
    uint32_t* __restrict__ tab = a.tab;
    uint32_t* __restrict__ cnt_tab = a.cnt_tab;
    uint32_t* __restrict__ slot_of = a.slot_of;
    AttPlan* __restrict__ plan = a.plan;
    uint4* __restrict__ arena_pad = static_cast<uint4*>(a.arena_pad32);
    const uint4* __restrict__ arena_src = static_cast<const uint4*>(a.arena_src);
    uint4* __restrict__ arena_dst = static_cast<uint4*>(a.arena_dst);
    const uint32_t mask = a.tab_mask;
    const unsigned long long arena_len = a.arena_len;
    uint32_t n = a.n;
    if (a.n_dev) n = min(n, *a.n_dev);  // the row count is itself a device result (pe_aggregate_exchange): n is its bound
    const uint32_t i = bid * 256 + threadIdx.x;
    // the caller's bits lie in device memory (16-byte aligned): this launch brings them into the staging arena itself -- a
    // copy command in front of it cost the step's chain 10-13 us (profiles/r03_timeline.txt: __amd_rocclr_copyBuffer)
    if (arena_src) {
        const unsigned long long whole = arena_len >> 4;
        for (unsigned long long q = i; q < whole; q += (unsigned long long)nblk * 256) arena_dst[q] = arena_src[q];
        if (i == 0 && (arena_len & 15)) {
            const uint8_t* sb = reinterpret_cast<const uint8_t*>(arena_src + whole);
            uint8_t* db = reinterpret_cast<uint8_t*>(arena_dst + whole);
            for (uint32_t b = 0; b < (uint32_t)(arena_len & 15); ++b) db[b] = sb[b];
        }
    }
    if (i == 0 && arena_pad && n == 0) { arena_pad[0] = make_uint4(0, 0, 0, 0); arena_pad[1] = make_uint4(0, 0, 0, 0); }
    if (i >= n) return;
    // k_bits_union reads whole dwords: up to 8 bytes past the last member's bits, which must read zero (no copy command
    // for 16 bytes: this kernel runs between the arena's copy and the union)
    if (i == 0 && arena_pad) { arena_pad[0] = make_uint4(0, 0, 0, 0); arena_pad[1] = make_uint4(0, 0, 0, 0); }
    Row9 r;
    load_row(r, rows, i);
    const uint32_t b0 = r.q[8].x, nb = r.q[8].y;
    // "attestation bits exceed the arena" / "target epoch must fit 32 bits" of the host path: the whole call fails
    if (nb > 0x7FFFFFFFu || (unsigned long long)b0 + ((unsigned long long)nb + 7) / 8 > arena_len || row_target_epoch(r) >= 0xFFFFFFFEull) atomicMax(&plan->error, ERR_INVALID_ARG);
    uint32_t h = att_hash(r) & mask;
    for (;;) {
        const uint32_t prev = atomicCAS(&tab[h], ATT_EMPTY, i);
        if (prev == ATT_EMPTY) break;  // first of its class to arrive here
        const uint4* q = rows + (size_t)9 * prev;   // any row of the slot's class: they all carry the same data
        bool eq = q[8].y == nb;
#pragma unroll
        for (int k = 0; k < 8; ++k) eq = eq && same4(q[k], r.q[k]);
        if (eq) {
            atomicMin(&tab[h], i);  // the slot keeps the row of first appearance
            break;
        }
        h = (h + 1) & mask;
    }
    slot_of[i] = h;
    atomicAdd(&cnt_tab[h], 1u);  // members of the class: k_att_plan reads the group's size here instead of counting
}
// with an arena to bring in: enough workgroups for the copy (one 16-byte word per lane and pass, at most 256 workgroups)
inline unsigned att_ingest_blocks(const IngestArgs& a)
{
    unsigned blocks = (a.n + 255) / 256;
    if (a.arena_src) blocks = std::max(blocks, (unsigned)std::min<uint64_t>(256, ((a.arena_len >> 4) + 255) / 256));
    return blocks;
}
}  // namespace

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_att_ingest(const IngestArgs a)
{
    __builtin_amdgcn_s_setprio(3);  // in front of a head, beside the previous step's G1 kernels (see fc_kernels.hip)
    att_ingest_body(blockIdx.x, gridDim.x, a);
}

void launch_att_ingest(hipStream_t s, const IngestArgs& a)
{
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_att_ingest, dim3(att_ingest_blocks(a)), dim3(256), 0, s, a);
}
void launch_att_ingest(hipStream_t s, const void* rows, uint32_t n, uint32_t* tab, uint32_t* cnt_tab, uint32_t tab_mask,
                       uint32_t* slot_of, uint64_t arena_len, AttPlan* plan, void* arena_pad32, const uint32_t* n_dev,
                       const void* arena_src, void* arena_dst)
{
    launch_att_ingest(s, IngestArgs{rows, n, tab, cnt_tab, tab_mask, slot_of, arena_len, plan, arena_pad32, n_dev, arena_src,
                                    arena_dst});
}
#endif

// ------------------------------------------------------------------ plan: one lane per input row, many workgroups
// Rounds 3-5 ran this as ONE workgroup of 1024 lanes with ~70 KB of LDS: 50 us of dependent L2 round trips beside a kernel
// that saturates the chip, and a paired launch made every LMD block reserve that LDS.  Now every input row has a lane of its
// own, 256 per workgroup, and what used to be three passes over the groups is one pass over the rows:
//   * a row is its class's representative iff the grouping table names it (k_att_ingest kept the row of first appearance);
//   * EVERY row resolves its committee from its own 144 bytes (the loads depend on the row alone, so they travel beside the
//     table look-up instead of behind it); only representatives count;
//   * group id, union word / byte offset and member-list start are exclusive prefix sums over the rows in batch order
//     (non-representatives contribute zero): wave shuffles -> LDS across the workgroup's four waves -> DECOUPLED LOOK-BACK
//     across workgroups: a workgroup publishes its own sums, adds up its predecessors' (64 per poll by its first wave,
//     stopping at the nearest one that already knows its inclusive prefix) and publishes its inclusive prefix.  Records are
//     8-byte words with a "written" bit, stored and loaded with agent-scope atomics (one sc1 store / load each: untorn, no
//     fence -- MI355X_MICROARCH.md, hand-off granules); a workgroup only ever waits for LOWER block indices, which the
//     dispatcher has started before it;
//   * maxima / totals the G1 plan needs go through device-scope atomics, committee row counts likewise;
//   * every workgroup then drains its memory operations and draws a ticket; the LAST one reads the totals (every atomic and
//     granule of the others is performed by then), picks k and the block size, turns the row counts of each table into
//     offsets + fill cursors, writes the AttPlan (device + pinned mirror) and clears the records for the next launch.
// k_att_members -- which already runs a lane per row -- writes the G1 descriptor and the committee's row-list entry of each
// group from the lane of its first row (it needs k / the block size: known only after this kernel).
// LDS: a few hundred bytes.  ~12 us at 8192 rows where the single workgroup took 50.
constexpr int PLAN_WAVES = PLAN_WG / 64;
constexpr uint32_t PLAN_STALL_LIMIT = 1u << 21;  // polls before a workgroup gives up waiting (seconds: never seen; a lost
                                                 // predecessor must not hang the device) -> ERR_STALL
constexpr uint32_t ERR_STALL = 12;

namespace {
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t word_load(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long plan_wave_incl_u64(unsigned long long v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ __forceinline__ unsigned long long plan_wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t plan_wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, (uint32_t)__shfl_xor(v, off, 64));
    return v;
}
// the three running sums of the scan: A = representatives (low 24 bits after look-back; 16 inside a workgroup) + class sizes,
// W = union words, B = union bytes
struct Sum3 { unsigned long long a, w, b; };

__device__ __forceinline__ void att_plan_body(const uint32_t bid, const uint32_t nb, const AttPlanArgs& a)
{
    __shared__ unsigned long long s_wave[3][PLAN_WAVES];
    __shared__ unsigned long long s_base[3];
    __shared__ uint32_t s_scan[PLAN_WAVES];
    __shared__ uint32_t s_ticket;
    const uint4* __restrict__ rows = static_cast<const uint4*>(a.rows);
    const uint32_t n = a.n_dev ? min(a.n, *a.n_dev) : a.n;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = bid * PLAN_WG + tid;
    const bool in = i < n;
    const bool dead = a.plan->error != 0;  // ingest refused a row: no groups are formed, nothing downstream may touch the bits

    // ---- 1. the row, its class, its committee
    uint32_t slot = 0;
    uint4 q0 = make_uint4(0, 0, 0, 0), q5 = q0, q8 = q0;
    if (in) {
        slot = a.slot_of[i];
        const uint4* p = rows + (size_t)9 * i;
        q0 = p[0]; q5 = p[5]; q8 = p[8];
    }
    uint32_t rep = NONE32, natts = 0;
    if (in) { rep = a.tab[slot]; natts = a.cnt_tab[slot]; }
    const unsigned long long spe = a.tables.slots_per_epoch;
    const unsigned long long slot_no = u64_of(q0.x, q0.y), index = u64_of(q0.z, q0.w), tep = u64_of(q5.z, q5.w);
    const uint32_t nbits = q8.y;
    uint32_t size = 0, table = NONE32, pos = 0, mbase = 0, index_over = 0;
    int32_t st = ST_OK;
    if (a.tables.t[0].valid && a.tables.t[0].epoch == tep) table = 0;
    else if (a.tables.t[1].valid && a.tables.t[1].epoch == tep) table = 1;
    if (table == NONE32) st = ST_NO_TABLE;
    else if (in) {
        const TableDev& t = a.tables.t[table];
        const unsigned long long cps = t.n_committees / spe;
        // compute_committee(index = (slot % SLOTS_PER_EPOCH) * cps + data.index, count = cps * SLOTS_PER_EPOCH):
        // on_attestation's get_beacon_committee asserts nothing about data.index itself -- only the flat id has
        // to exist; pe:727 (process_attestation) and pe_aggregate require data.index < cps
        const unsigned long long flat = index < 0xFFFFFFFFull ? (slot_no % spe) * cps + index : ~0ull;
        index_over = index >= cps ? 1u : 0u;
        if (flat >= t.n_committees) st = ST_INDEX_RANGE;
        else {
            pos = (uint32_t)flat;
            mbase = t.offsets[pos];
            size = t.offsets[pos + 1] - mbase;
            if (nbits != size) st = ST_BITS_LENGTH;  // len(aggregation_bits) == len(committee), pe:730
        }
    }
    if (in) a.rep_of[i] = rep;
    const bool is_rep = in && rep == i && !dead;
    const bool ok = is_rep && st == ST_OK;
    const uint32_t words = (nbits + 31) >> 5, bytes = (nbits + 7) >> 3;

    // ---- 2. exclusive prefix sums in batch order: inside the wave, across the workgroup, across the grid
    Sum3 v;
    v.a = is_rep ? (1ull | ((unsigned long long)natts << 16)) : 0ull;
    v.w = is_rep ? (unsigned long long)words : 0ull;
    v.b = is_rep ? (unsigned long long)bytes : 0ull;
    Sum3 inc;
    inc.a = plan_wave_incl_u64(v.a);
    inc.w = plan_wave_incl_u64(v.w);
    inc.b = plan_wave_incl_u64(v.b);
    if (lane == 63) { s_wave[0][wave] = inc.a; s_wave[1][wave] = inc.w; s_wave[2][wave] = inc.b; }
    __syncthreads();
    Sum3 wgb{0, 0, 0}, tot{0, 0, 0};  // sums of the waves in front of this one; of the whole workgroup
#pragma unroll
    for (int w = 0; w < PLAN_WAVES; ++w) {
        const unsigned long long ta = s_wave[0][w], tw = s_wave[1][w], tb = s_wave[2][w];
        if (w < (int)wave) { wgb.a += ta; wgb.w += tw; wgb.b += tb; }
        tot.a += ta; tot.w += tw; tot.b += tb;
    }
    // own sums in the grid's form: representatives in 24 bits, class sizes above
    const unsigned long long tot_a24 = (tot.a & 0xFFFFull) | ((tot.a >> 16) << 24);
    PlanRec* __restrict__ rec = a.rec;
    if (wave == 0) {
        Sum3 base{0, 0, 0};
        if (bid == 0) {
            if (lane == 0) {
                granule_store(&rec[0].incl[0], (tot_a24 << 1) | 1ull);
                granule_store(&rec[0].incl[1], (tot.w << 1) | 1ull);
                granule_store(&rec[0].incl[2], (tot.b << 1) | 1ull);
            }
        } else {
            if (lane == 0) {
                granule_store(&rec[bid].agg[0], (tot_a24 << 1) | 1ull);
                granule_store(&rec[bid].agg[1], (tot.w << 1) | 1ull);
                granule_store(&rec[bid].agg[2], (tot.b << 1) | 1ull);
            }
            int hi = (int)bid - 1;
            uint32_t polls = 0;
            bool stalled = false;
            for (;;) {  // one window of 64 predecessors per turn, nearest first (lane 0 = block hi)
                const int j = hi - (int)lane;
                const bool valid = j >= 0;
                unsigned long long x0 = 0, x1 = 0, x2 = 0;
                bool have_incl = false;
                for (;;) {
                    bool ready = true;
                    if (valid) {
                        const unsigned long long i0 = granule_load(&rec[j].incl[0]), i1 = granule_load(&rec[j].incl[1]),
                                                 i2 = granule_load(&rec[j].incl[2]);
                        if (i0 & i1 & i2 & 1ull) { x0 = i0; x1 = i1; x2 = i2; have_incl = true; }
                        else {
                            const unsigned long long a0 = granule_load(&rec[j].agg[0]), a1 = granule_load(&rec[j].agg[1]),
                                                     a2 = granule_load(&rec[j].agg[2]);
                            if (a0 & a1 & a2 & 1ull) { x0 = a0; x1 = a1; x2 = a2; }
                            else ready = false;
                        }
                    }
                    if (__all(ready)) break;
                    if (++polls > PLAN_STALL_LIMIT) { stalled = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (stalled) break;
                const unsigned long long mi = __ballot(valid && have_incl);
                const int first = mi ? __builtin_ctzll(mi) : 64;  // nearest predecessor that knows its inclusive prefix
                const bool take = valid && (int)lane <= first;
                base.a += plan_wave_sum_u64(take ? x0 >> 1 : 0ull);
                base.w += plan_wave_sum_u64(take ? x1 >> 1 : 0ull);
                base.b += plan_wave_sum_u64(take ? x2 >> 1 : 0ull);
                if (mi) break;
                hi -= 64;
                if (hi < 0) break;
            }
            if (stalled && lane == 0) atomicMax(&a.sync->err, ERR_STALL);
            if (lane == 0) {
                granule_store(&rec[bid].incl[0], ((base.a + tot_a24) << 1) | 1ull);
                granule_store(&rec[bid].incl[1], ((base.w + tot.w) << 1) | 1ull);
                granule_store(&rec[bid].incl[2], ((base.b + tot.b) << 1) | 1ull);
            }
        }
        if (lane == 0) { s_base[0] = base.a; s_base[1] = base.w; s_base[2] = base.b; }
    }
    __syncthreads();

    // ---- 3. the group's records, from the lane of its first row
    uint32_t g = 0;
    if (is_rep) {
        const unsigned long long ex_a = wgb.a + inc.a - v.a;  // in-workgroup form: count | sizes << 16
        g = (uint32_t)(s_base[0] & 0xFFFFFFull) + (uint32_t)(ex_a & 0xFFFFull);
        const uint32_t list_start = (uint32_t)((s_base[0] >> 24) + (ex_a >> 16));
        const uint32_t out_word = (uint32_t)(s_base[1] + wgb.w + inc.w - v.w);
        const uint32_t out_byte = (uint32_t)(s_base[2] + wgb.b + inc.b - v.b);
        AttGroup G;
        G.rep = i;
        G.n_atts = natts;
        G.list_start = list_start;
        G.cursor = 0;
        G.n_bits = nbits;
        G.out_word = out_word;
        G.out_byte = out_byte;
        G.table = st == ST_NO_TABLE ? NONE32 : table;
        G.pos = pos;
        G.size = size;
        G.member_base = mbase;
        G.sig_valid = FLAG_SIG_VALID;
        G.status_agg = (uint32_t)st;
        G.index_over = index_over;
        G.pad[0] = G.pad[1] = 0;
        a.grp[g] = G;
        UnionGroup u;
        u.list_start = list_start;
        u.n_atts = natts;
        u.n_bits = nbits;
        u.out_word = out_word;
        a.ug[g] = u;
        a.gid_of_row[i] = g;
        if (ok) atomicAdd(&a.crow_cnt[table][pos], 1u);  // rows per committee of each candidate table
    }
    // sums / maxima over all groups: one atomic per wave and value that has something to say
    {
        // the host path fails the whole aggregate on a group without a committee when pubkeys are wanted (engine_attest.cpp)
        const uint32_t e = (is_rep && (st != ST_OK || index_over) && a.want_pk) ? (st == ST_NO_TABLE ? ERR_NO_COMMITTEES : ERR_INVALID_ARG) : 0u;
        const uint32_t w_err = plan_wave_max_u32(e);
        const uint32_t w_size = plan_wave_max_u32(ok ? size : 0u);
        // word layout == byte layout as long as every union but the LAST one is a whole number of words: remember the first
        // group that is not (the last workgroup knows which group is the last)
        const uint32_t w_mis = plan_wave_max_u32((is_rep && bytes != 4 * words) ? ~g : 0u);
        const uint32_t w_r0 = (uint32_t)__builtin_popcountll(__ballot(ok && table == 0));
        const uint32_t w_r1 = (uint32_t)__builtin_popcountll(__ballot(ok && table == 1));
        const unsigned long long w_mem = plan_wave_sum_u64(ok ? (unsigned long long)size : 0ull);
        if (lane == 0) {
            PlanSync* S = a.sync;
            if (w_err) atomicMax(&S->err, w_err);
            if (w_size) atomicMax(&S->max_size, w_size);
            if (w_mis) atomicMax(&S->mis_key, w_mis);
            if (w_r0) atomicAdd(&S->rows_t[0], w_r0);
            if (w_r1) atomicAdd(&S->rows_t[1], w_r1);
            if (w_mem) atomicAdd(&S->total_members, w_mem);
        }
    }

    // ---- 4. arrive; the last workgroup writes the plan
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics and granules are performed
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(&a.sync->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != nb - 1) return;

    PlanSync* S = a.sync;
    const uint32_t rows_t0 = word_load(&S->rows_t[0]), rows_t1 = word_load(&S->rows_t[1]);
    if (tid == 0) {
        const unsigned long long la = granule_load(&rec[nb - 1].incl[0]) >> 1, lw = granule_load(&rec[nb - 1].incl[1]) >> 1,
                                 lb = granule_load(&rec[nb - 1].incl[2]) >> 1;
        const uint32_t ng = (uint32_t)(la & 0xFFFFFFull);
        const uint32_t max_size = word_load(&S->max_size), mis_key = word_load(&S->mis_key);
        const unsigned long long total_members = granule_load(&S->total_members);
        uint32_t err = max(a.plan->error, word_load(&S->err));
        if (!err && (lb > a.out_arena_cap || lw > 0xFFFFFFFFull)) err = ERR_CAPACITY;  // "output bit arena too small"
        // one block size for every group: k members per lane, blocks of BL = 2^L lanes, group g at slot g * BL.  k
        // follows from the largest committee so that its tasks fill a block exactly or nearly (sizes that differ by one --
        // 511 / 512 members -- would otherwise put half of the lanes of every block to sleep)
        uint32_t k = a.min_k, L = 0;
        {
            const unsigned long long k0 = max((unsigned long long)a.min_k, (total_members + a.target_slots - 1) / a.target_slots);
            uint32_t tasks = (uint32_t)((max_size + k0 - 1) / k0);
            if (tasks > (uint32_t)G1_WG) tasks = G1_WG;
            while ((1u << L) < tasks) ++L;
            while (L > 0 && ((unsigned long long)ng << L) > a.slot_cap) --L;  // bounded scratch: fewer, longer lanes
            k = max(a.min_k, (max_size + (1u << L) - 1) >> L);
            if (k == 0) k = 1;
        }
        AttPlan p;
        p.n_groups = err ? 0u : ng;  // a failing aggregate forms no groups: the handlers behind it apply nothing
        p.n_slots = p.n_groups << L;
        p.k = k;
        p.log2_block = L;
        p.out_words = (uint32_t)lw;
        p.out_bytes = (uint32_t)min(lb, 0xFFFFFFFFull);
        p.error = err;
        p.packed_same = (mis_key != 0 && (~mis_key) + 1 < ng) ? 0u : 1u;
        p.n_rows_table[0] = rows_t0;
        p.n_rows_table[1] = rows_t1;
        p.n_rows_in = n;
        p.last_error = err;
        p.total_members = total_members;
        *a.plan = p;
        *a.plan_host = p;
    }
    // rows per committee of each candidate table: counts -> offsets + fill cursors (the lists themselves are filled by
    // k_att_members; unordered, consumers order by group id).  A table without a row is skipped: its consumers
    // (k_lmd_vm_tables, k_participation_tables) return on plan->n_rows_table[t] == 0.
    for (int t = 0; t < 2; ++t) {
        if (!a.tables.t[t].valid || (t ? rows_t1 : rows_t0) == 0) continue;
        const uint32_t nc = a.tables.t[t].n_committees;
        const uint32_t per = (nc + 1 + PLAN_WG - 1) / PLAN_WG;  // consecutive entries per lane (entry nc: the end mark)
        const uint32_t b0 = min(tid * per, nc + 1), b1 = min(b0 + per, nc + 1);
        uint32_t sum = 0;
        for (uint32_t c = b0; c < b1; ++c) sum += c < nc ? word_load(&a.crow_cnt[t][c]) : 0u;
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += o;
        }
        __syncthreads();  // s_scan may still be read by the previous table's scan
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wave; ++w) run += s_scan[w];
        for (uint32_t c = b0; c < b1; ++c) {
            const uint32_t cnt = c < nc ? word_load(&a.crow_cnt[t][c]) : 0u;
            a.crow_start[t][c] = run;
            a.crow_cursor[t][c] = run;
            if (c < nc && cnt) __hip_atomic_store(&a.crow_cnt[t][c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
            run += cnt;
        }
    }
    // the records, clean for the next launch
    __syncthreads();
    // (stored the way they are read -- agent-scope, past the XCD's L2 -- so that no stale line of zeros waits there for the
    // next launch's polls)
    for (uint32_t j = tid; j < nb; j += PLAN_WG)
#pragma unroll
        for (int q = 0; q < 3; ++q) { granule_store(&rec[j].agg[q], 0ull); granule_store(&rec[j].incl[q], 0ull); }
    if (tid == 0) {
        unsigned long long* z = reinterpret_cast<unsigned long long*>(S);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(PlanSync) / 8); ++q) granule_store(z + q, 0ull);
    }
}
inline unsigned att_plan_blocks(const AttPlanArgs& a) { return std::max(1u, (a.n + PLAN_WG - 1) / PLAN_WG); }
}  // namespace

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(PLAN_WG) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_att_plan(const AttPlanArgs a)
{
    __builtin_amdgcn_s_setprio(3);
    att_plan_body(blockIdx.x, gridDim.x, a);
}

void launch_att_plan(hipStream_t s, const AttPlanArgs& a)
{
    hipLaunchKernelGGL(k_att_plan, dim3(att_plan_blocks(a)), dim3(PLAN_WG), 0, s, a);
}
#endif

// ------------------------------------------------------------------ members
namespace {
__device__ __forceinline__ void att_members_body(const uint32_t i /* input row of this lane */, const MembersArgs& a)
{
    const uint4* __restrict__ rows = static_cast<const uint4*>(a.rows);
    uint32_t* __restrict__ tab = a.tab;
    uint32_t* __restrict__ cnt_tab = a.cnt_tab;
    const uint32_t* __restrict__ slot_of = a.slot_of;
    const uint32_t* __restrict__ rep_of = a.rep_of;
    const uint32_t* __restrict__ gid_of_row = a.gid_of_row;
    AttGroup* __restrict__ grp = a.grp;
    AttPlan* __restrict__ plan = a.plan;
    uint32_t* __restrict__ ubytes = a.ubytes;
    uint32_t* __restrict__ member_row = a.member_row;
    uint32_t* __restrict__ host_group_of = a.host_group_of;
    uint4* __restrict__ host_out_rows = static_cast<uint4*>(a.host_out_rows);
    uint32_t n = a.n;
    if (a.n_dev) n = min(n, *a.n_dev);
    if (i == 0 && n == 0) plan->error = 0;
    if (i >= n) return;
    const uint32_t slot = slot_of[i];
    if (plan->n_groups) {
        const uint32_t rep = rep_of[i];
        const uint32_t g = gid_of_row[rep];
        const uint4 q8 = rows[(size_t)9 * i + 8];
        if (host_group_of) host_group_of[i] = g;
        AttGroup& G = grp[g];
        const uint32_t p = G.list_start + atomicAdd(&G.cursor, 1u);
        ubytes[p] = q8.x;       // byte offset of the member's bits in the arena (copied whole, offset 0)
        member_row[p] = i;
        if (!(q8.z & FLAG_SIG_VALID)) atomicAnd(&G.sig_valid, 0u);
        if (rep == i) {
            // the lane of the group's first row: the group's summation descriptor (k and the block size are the plan's: known
            // since k_att_plan's last workgroup) and its entry in its committee's row list
            const uint32_t k = plan->k, L = plan->log2_block;
            const bool ok = G.status_agg == (uint32_t)ST_OK;
            G1Group d;
            d.member_start = G.member_base;
            d.n_members = ok ? G.size : 0u;
            d.bits_word = G.out_word;
            d.slot_base = g << L;
            d.n_tasks = ok ? (G.size + k - 1) / k : 0u;
            d.k = k | (G.table == 1 ? 0x80000000u : 0u);
            d.log2_block = L;
            d.out_base = g;
            a.g1[g] = d;
            if (ok) a.crow_list[G.table][atomicAdd(&a.crow_cursor[G.table][G.pos], 1u)] = g;
            if (host_out_rows) {  // the group's output row: its data, bits_offset into the packed output arena
                uint4* o = host_out_rows + (size_t)9 * g;
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) o[k8] = rows[(size_t)9 * i + k8];
                o[8] = make_uint4(G.out_byte, q8.y, q8.z, 0u);  // flags: the host folds in the verdicts at completion
            }
        }
    }
    tab[slot] = ATT_EMPTY;  // every row of a class clears the class's slot: the table is empty again for the next call
    cnt_tab[slot] = 0;
    if (i == 0) plan->error = 0;  // consumed by k_att_plan (mirrored to the host): k_att_ingest of the next call starts clean
}
}  // namespace

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_att_members(const MembersArgs a)
{
    __builtin_amdgcn_s_setprio(3);
    att_members_body(blockIdx.x * 256 + threadIdx.x, a);
}

void launch_att_members(hipStream_t s, const MembersArgs& a)
{
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_att_members, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}
#endif

// ------------------------------------------------------------------ committee-sharded exchange (pe_aggregate_exchange)
// Every rank aggregates the rows of ITS committees; what the other ranks need of the result -- the aggregate attestation
// itself: AttestationData + OR-ed bits (pe:714-717), its attester count and its verdict flags -- is packed into fixed
// slots, all-gathered, and unpacked into one dense batch that the receiving rank ingests like any batch of rows.
//   send buffer: [0] groups packed, [1] error, [2..3] reserved | slot g: 36 words row, count, reserved, `wps` words of bits
#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_att_pack(const uint4* __restrict__ rows, const AttGroup* __restrict__ grp, const AttPlan* __restrict__ plan,
           const uint32_t* __restrict__ res_bits, const uint32_t* __restrict__ res_info, uint32_t slots, uint32_t wps,
           uint32_t* __restrict__ send)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ng_all = plan->n_groups;
    const uint32_t ng = min(ng_all, slots);
    if (g == 0) {
        send[0] = ng;
        send[1] = ng_all > slots ? ERR_CAPACITY : plan->last_error;  // not plan->error: k_att_members has cleared it
        send[2] = send[3] = 0;
    }
    if (g >= ng) return;
    const AttGroup G = grp[g];
    const uint32_t slot_words = 38 + wps;
    uint32_t* o = send + 4 + (size_t)g * slot_words;
    const uint4* r = rows + (size_t)9 * G.rep;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint4 q = r[k];
        o[4 * k] = q.x; o[4 * k + 1] = q.y; o[4 * k + 2] = q.z; o[4 * k + 3] = q.w;
    }
    const uint32_t count = res_info[2 * g], overlap = res_info[2 * g + 1];
    uint32_t flags = r[8].z & ~FLAG_SIG_VALID;
    if (G.sig_valid && !overlap) flags |= FLAG_SIG_VALID;   // the AND of the members' verdicts; overlapping members never verify (A.8)
    if (overlap) flags |= FLAG_OVERLAPPING;
    o[32] = 0;          // bits_offset: set by the receiver
    o[33] = G.n_bits;
    o[34] = flags;
    o[35] = 0;
    o[36] = count;
    o[37] = 0;
    const uint32_t nw = (G.n_bits + 31) >> 5;
    if (nw > wps) { atomicMax(&send[1], ERR_CAPACITY); return; }
    for (uint32_t j = 0; j < wps; ++j) o[38 + j] = j < nw ? res_bits[G.out_word + j] : 0u;
}

void launch_att_pack(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, const uint32_t* res_bits,
                     const uint32_t* res_info, uint32_t slots, uint32_t wps, uint32_t* send)
{
    hipLaunchKernelGGL(k_att_pack, dim3((std::max(slots, 1u) + 255) / 256), dim3(256), 0, s, static_cast<const uint4*>(rows), grp,
                       plan, res_bits, res_info, slots, wps, send);
}

// recv: `world` send buffers back to back.  Rank r's groups land behind those of ranks < r: out_rows[base_r + j] with
// bits_offset = (base_r + j) * 4 * wps into out_bits; *n_dev = the total; *err = the first non-zero error word.
__global__ void __launch_bounds__(256)
k_att_unpack(const uint32_t* __restrict__ recv, uint32_t world, uint32_t slots, uint32_t wps, uint32_t* __restrict__ out_rows,
             uint32_t* __restrict__ out_bits, uint32_t* __restrict__ n_dev, uint32_t* __restrict__ err_host)
{
    const uint32_t slot_words = 38 + wps;
    const size_t rank_words = 4 + (size_t)slots * slot_words;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = t / slots, j = t % slots;
    if (t == 0) {
        uint32_t total = 0, err = 0;
        for (uint32_t q = 0; q < world; ++q) {
            total += recv[q * rank_words];
            if (!err) err = recv[q * rank_words + 1];
        }
        *n_dev = err ? 0u : total;   // a rank whose aggregate failed: nothing is applied anywhere
        if (err_host) *err_host = err;
    }
    if (r >= world) return;
    uint32_t base = 0;
    for (uint32_t q = 0; q < r; ++q) base += recv[q * rank_words];
    if (j >= recv[r * rank_words]) return;
    const uint32_t* in = recv + r * rank_words + 4 + (size_t)j * slot_words;
    const uint32_t dst = base + j;
    uint32_t* o = out_rows + (size_t)36 * dst;
#pragma unroll
    for (int k = 0; k < 36; ++k) o[k] = in[k];
    o[32] = dst * 4 * wps;
    uint32_t* b = out_bits + (size_t)dst * wps;
    for (uint32_t k = 0; k < wps; ++k) b[k] = in[38 + k];
}

void launch_att_unpack(hipStream_t s, const uint32_t* recv, uint32_t world, uint32_t slots, uint32_t wps, void* out_rows,
                       uint32_t* out_bits, uint32_t* n_dev, uint32_t* err_host)
{
    const uint32_t n = std::max(world * slots, 1u);
    hipLaunchKernelGGL(k_att_unpack, dim3((n + 255) / 256), dim3(256), 0, s, recv, world, slots, wps,
                       static_cast<uint32_t*>(out_rows), out_bits, n_dev, err_host);
}
#endif

// ------------------------------------------------------------------ block lookups
namespace {
__device__ __forceinline__ uint32_t find_block_dev(const BlockTableDev& bt, const uint4& r0, const uint4& r1)
{
    uint32_t h = r0.x & bt.root_mask;  // roots are hash outputs: the leading word is uniform
    for (;;) {
        const uint32_t idx = bt.root_tab[h];
        if (idx == NONE32) return NONE32;
        const uint4* q = reinterpret_cast<const uint4*>(bt.roots + 32ull * idx);
        if (same4(q[0], r0) && same4(q[1], r1)) return idx;
        h = (h + 1) & bt.root_mask;
    }
}
__device__ __forceinline__ bool root_equals(const uint8_t* root32, const uint4& r0, const uint4& r1)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(root32);
    return w[0] == r0.x && w[1] == r0.y && w[2] == r0.z && w[3] == r0.w && w[4] == r1.x && w[5] == r1.y && w[6] == r1.z &&
           w[7] == r1.w;
}
}  // namespace

// ------------------------------------------------------------------ validate_on_attestation (A.4) per group
namespace {
__device__ __forceinline__ void att_validate_fc_body(const uint32_t g /* group of this lane */, const ValidateFcArgs& a)
{
    const uint4* __restrict__ rows = static_cast<const uint4*>(a.rows);
    const AttGroup* __restrict__ grp = a.grp;
    const AttPlan* __restrict__ plan = a.plan;
    const uint32_t cap = a.cap;
    const BlockTableDev& bt = a.bt;
    const FcCtx& fc = a.fc;
    const uint32_t* __restrict__ union_info = a.union_info;
    AttRow* __restrict__ out_rows = a.out_rows;
    int32_t* __restrict__ status_dev = a.status_dev;
    int32_t* __restrict__ status_host = a.status_host;
    uint32_t* __restrict__ count_host = a.count_host;
    uint32_t* __restrict__ err_host = a.err_host;
    const uint32_t ng = plan->n_groups;
    if (ng > cap) {  // the caller's status / count arrays hold fewer entries than groups were formed: nothing applies
        if (g < ng) status_dev[g] = -1;
        if (g == 0) *err_host = ERR_CAPACITY;
        return;
    }
    if (g >= ng) return;
    const AttGroup G = grp[g];
    Row9 r;
    load_row(r, rows, G.rep);
    const unsigned long long slot = row_slot(r), tep = row_target_epoch(r), spe = fc.slots_per_epoch;
    const uint32_t flags = r.q[8].z;
    int32_t st = ST_OK;
    uint32_t blk = NONE32;
    // validate_target_epoch_against_current_time (skipped for attestations from blocks, pe:1423)
    if (!(flags & FLAG_FROM_BLOCK) && tep != fc.cur_epoch && tep != fc.prev_epoch) st = ST_EPOCH_TIME;
    else if (tep != slot / spe) st = ST_EPOCH_SLOT;
    else {
        const uint32_t tgt = find_block_dev(bt, r.q[6], r.q[7]);
        if (tgt == NONE32) st = ST_UNKNOWN_TARGET;
        else {
            blk = find_block_dev(bt, r.q[1], r.q[2]);
            if (blk == NONE32) st = ST_UNKNOWN_BLOCK;
            else {
                uint32_t p = bt.pos_of_idx[blk];
                if (bt.slot_pos[p] > slot) st = ST_BLOCK_AFTER_SLOT;
                else {
                    // get_ancestor(store, beacon_block_root, compute_start_slot_at_epoch(target.epoch)) (A.2): the
                    // block's slot is <= data.slot and the epoch is data.slot's, so the walk is at most
                    // SLOTS_PER_EPOCH parent steps (slots strictly increase along parent links)
                    const unsigned long long start = tep * spe;
                    while (bt.slot_pos[p] > start && bt.parent_pos[p] != NONE32) p = bt.parent_pos[p];
                    if (p != bt.pos_of_idx[tgt]) st = ST_TARGET_NOT_ANCESTOR;
                    else if (fc.cur_slot < slot + 1) st = ST_SLOT_NOT_PAST;
                }
            }
        }
    }
    if (st == ST_OK && G.status_agg) st = (int32_t)G.status_agg;  // committee resolution: table, index, bits length
    // is_valid_indexed_attestation (A.7): the signature verdict, then what the OR-ed bits say (overlap, emptiness)
    const uint32_t cnt = union_info[2 * g], overlap = union_info[2 * g + 1];
    if (st == ST_OK) {
        if (!G.sig_valid) st = ST_BAD_SIGNATURE;
        else if (overlap) st = ST_BAD_SIGNATURE;
        else if (cnt == 0) st = ST_EMPTY;
    }
    AttRow o;
    o.member_base = G.member_base;
    o.n_bits = G.size;
    o.bits_word = G.out_word;
    o.block_idx = blk;
    o.epoch_p1 = (uint32_t)tep + 1;
    o.order = g;
    o.flag_mask = 0;
    o.which = 0;
    o.slot = (uint32_t)slot;
    o.gate = g;
    out_rows[g] = o;
    status_dev[g] = st;
    status_host[g] = st;
    if (count_host) count_host[g] = st == ST_OK ? cnt : 0u;
}
}  // namespace

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_att_validate_fc(const ValidateFcArgs a)
{
    __builtin_amdgcn_s_setprio(3);
    att_validate_fc_body(blockIdx.x * 256 + threadIdx.x, a);
}

void launch_att_validate_fc(hipStream_t s, const ValidateFcArgs& a)
{
    if (a.n_bound == 0) return;
    hipLaunchKernelGGL(k_att_validate_fc, dim3((a.n_bound + 255) / 256), dim3(256), 0, s, a);
}
void launch_att_validate_fc(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, uint32_t n_bound,
                            uint32_t cap, BlockTableDev bt, FcCtx fc, const uint32_t* union_info, AttRow* out_rows,
                            int32_t* status_dev, int32_t* status_host, uint32_t* count_host, uint32_t* err_host)
{
    launch_att_validate_fc(s, ValidateFcArgs{rows, grp, plan, n_bound, cap, bt, fc, union_info, out_rows, status_dev,
                                             status_host, count_host, err_host});
}
#endif

// ------------------------------------------------------------------ process_attestation's asserts + flag indices per group
#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_att_validate_state(const uint4* __restrict__ rows, const AttGroup* __restrict__ grp, const AttPlan* __restrict__ plan,
                     uint32_t cap, BlockTableDev bt, const StateCtxDev S,
                     const uint32_t* __restrict__ union_info, AttRow* __restrict__ out_rows,
                     int32_t* __restrict__ status_dev, int32_t* __restrict__ status_host, uint32_t* __restrict__ err_host)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ng = plan->n_groups;
    if (ng > cap) {
        if (g < ng) status_dev[g] = -1;
        if (g == 0) *err_host = ERR_CAPACITY;
        return;
    }
    if (g >= ng) return;
    const AttGroup G = grp[g];
    Row9 r;
    load_row(r, rows, G.rep);
    const unsigned long long slot = row_slot(r), tep = row_target_epoch(r), spe = S.slots_per_epoch;
    int32_t st = ST_OK;
    uint32_t flag_mask = 0;
    if (tep != S.prev_epoch && tep != S.cur_epoch) st = ST_EPOCH_TIME;                                   // pe:724
    else if (tep != slot / spe) st = ST_EPOCH_SLOT;                                                       // pe:725
    else if (!(slot + S.min_inclusion_delay <= S.slot && S.slot <= slot + spe)) st = ST_INCLUSION;        // pe:726
    else if (G.status_agg == (uint32_t)ST_NO_TABLE) st = ST_NO_TABLE;
    else if (G.index_over) st = ST_INDEX_RANGE;                                                           // pe:727
    else if (G.status_agg) st = (int32_t)G.status_agg;                                                    // pe:729-730
    else {
        // get_attestation_participation_flag_indices (A.9)
        const bool is_cur = tep == S.cur_epoch;
        const unsigned long long j_epoch = is_cur ? S.cj_epoch : S.pj_epoch;
        const uint8_t* j_root = is_cur ? S.cj_root : S.pj_root;
        const uint4 s0 = make_uint4(r.q[3].z, r.q[3].w, r.q[4].x, r.q[4].y), s1 = make_uint4(r.q[4].z, r.q[4].w, r.q[5].x, r.q[5].y);
        if (row_source_epoch(r) != j_epoch || !root_equals(j_root, s0, s1)) st = ST_SOURCE;  // assert is_matching_source
        else {
            const uint32_t tgt_blk = S.tgt_blk[is_cur ? 0 : 1];
            const bool matching_target = tgt_blk != NONE32 && root_equals(bt.roots + 32ull * tgt_blk, r.q[6], r.q[7]);
            const unsigned long long j = slot + spe - S.slot;  // pe:726 puts data.slot into [state.slot - spe, state.slot)
            const uint32_t head_blk = j < 64 ? S.head_blk[j] : NONE32;
            const bool matching_head = matching_target && head_blk != NONE32 &&
                                       root_equals(bt.roots + 32ull * head_blk, r.q[1], r.q[2]);
            const unsigned long long delay = S.slot - slot;
            if (delay <= S.sqrt_spe) flag_mask |= 1u;                               // TIMELY_SOURCE
            if (matching_target && delay <= spe) flag_mask |= 2u;                   // TIMELY_TARGET
            if (matching_head && delay == S.min_inclusion_delay) flag_mask |= 4u;   // TIMELY_HEAD
        }
    }
    const uint32_t cnt = union_info[2 * g], overlap = union_info[2 * g + 1];
    if (st == ST_OK) {
        if (!G.sig_valid) st = ST_BAD_SIGNATURE;   // pe:736
        else if (overlap) st = ST_BAD_SIGNATURE;
        else if (cnt == 0) st = ST_EMPTY;
    }
    AttRow o;
    o.member_base = G.member_base;
    o.n_bits = G.size;
    o.bits_word = G.out_word;
    o.block_idx = 0;
    o.epoch_p1 = 0;
    o.order = g;
    o.flag_mask = flag_mask;
    o.which = tep == S.cur_epoch ? 0u : 1u;  // pe:739-742
    o.slot = 0;
    o.gate = g;
    out_rows[g] = o;
    status_dev[g] = st;
    status_host[g] = st;
}

void launch_att_validate_state(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, uint32_t n_bound,
                               uint32_t cap, BlockTableDev bt, const StateCtxDev& st, const uint32_t* union_info,
                               AttRow* out_rows, int32_t* status_dev, int32_t* status_host, uint32_t* err_host)
{
    if (n_bound == 0) return;
    hipLaunchKernelGGL(k_att_validate_state, dim3((n_bound + 255) / 256), dim3(256), 0, s, static_cast<const uint4*>(rows),
                       grp, plan, cap, bt, st, union_info, out_rows, status_dev, status_host, err_host);
}
#endif

// ------------------------------------------------------------------ LMD update, validator-major, both tables
// One lane per validator walks the rows of ITS committee.  The lists are unordered, so the spec's sequential rule
// (pe:1435-1441: a later target epoch wins; among equal epochs the first in batch order, and only against a stored vote
// of a strictly earlier epoch) is applied by comparing (epoch, order) explicitly.
namespace {
__device__ __forceinline__ void lmd_vm_tables_body(const unsigned long long v /* validator of this lane */, const LmdVmArgs& a)
{
    const AttRow* __restrict__ rows = a.rows;
    const TablesDev& tables = a.tables;
    const uint32_t* __restrict__ cs0 = a.crow_start[0];
    const uint32_t* __restrict__ cs1 = a.crow_start[1];
    const uint32_t* __restrict__ cl0 = a.crow_list[0];
    const uint32_t* __restrict__ cl1 = a.crow_list[1];
    const AttPlan* __restrict__ plan = a.plan;
    const uint32_t* __restrict__ bit_arena = a.bit_arena;
    const uint8_t* __restrict__ flags = a.flags;
    unsigned long long* __restrict__ vote_key = reinterpret_cast<unsigned long long*>(a.vote_key);
    uint32_t* __restrict__ vote_block = a.vote_block;
    uint32_t* __restrict__ vote_slot = a.vote_slot;
    const uint32_t* __restrict__ gates = a.gates;
    if (v >= a.n_val) return;
    // a validator sits in one committee of EACH epoch: the same lane walks both tables, one after the other (two lanes
    // would race on its latest message)
    uint32_t best_e = 0, best_order = NONE32, new_block = NONE32, new_slot = 0;
    bool loaded = false;
    for (int t = 0; t < 2; ++t) {
        if (!tables.t[t].valid || plan->n_rows_table[t] == 0) continue;
        const uint32_t c = tables.t[t].inv_comm[v];
        if (c == NONE32) continue;
        const uint32_t* crow_start = t ? cs1 : cs0;
        const uint32_t* crow_list = t ? cl1 : cl0;
        const uint32_t kb = crow_start[c], ke = crow_start[c + 1];
        if (kb == ke) continue;
        if (!loaded) {
            if (flags[v] & VAL_EQUIVOCATING_BIT) return;  // pe:1438
            best_e = (uint32_t)(vote_key[v] >> 32);        // epoch + 1 of the stored message; 0 = none
            loaded = true;
        }
        const uint32_t i = tables.t[t].inv_pos[v];
        for (uint32_t k = kb; k < ke; ++k) {
            const AttRow r = rows[crow_list[k]];
            if (i >= r.n_bits) continue;
            if (gates[r.gate] != 0) continue;  // rejected by validation (or voided: overlapping members)
            if (!((bit_arena[r.bits_word + (i >> 5)] >> (i & 31)) & 1u)) continue;
            if (r.epoch_p1 > best_e || (r.epoch_p1 == best_e && new_block != NONE32 && r.order < best_order)) {
                best_e = r.epoch_p1;
                best_order = r.order;
                new_block = r.block_idx;
                new_slot = r.slot;
            }
        }
    }
    if (new_block != NONE32) {
        vote_key[v] = ((unsigned long long)best_e << 32) | 0xFFFFFFFFull;
        vote_block[v] = new_block;
        if (vote_slot) vote_slot[v] = new_slot;
    }
}
}  // namespace

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_lmd_vm_tables(const LmdVmArgs a)
{
    __builtin_amdgcn_s_setprio(3);  // in front of a head, beside the previous step's G1 kernels (see fc_kernels.hip)
    lmd_vm_tables_body((unsigned long long)blockIdx.x * 256 + threadIdx.x, a);
}

void launch_lmd_vm_tables(hipStream_t s, const LmdVmArgs& a)
{
    if (a.n_val == 0) return;
    hipLaunchKernelGGL(k_lmd_vm_tables, dim3((unsigned)((a.n_val + 255) / 256)), dim3(256), 0, s, a);
}
void launch_lmd_vm_tables(hipStream_t s, const AttRow* rows, TablesDev tables, uint32_t* const crow_start[2],
                          uint32_t* const crow_list[2], const AttPlan* plan, const uint32_t* bit_arena,
                          const uint8_t* flags, uint64_t n_val, uint64_t* vote_key, uint32_t* vote_block,
                          uint32_t* vote_slot, const uint32_t* gates)
{
    launch_lmd_vm_tables(s, LmdVmArgs{rows, tables, {crow_start[0], crow_start[1]}, {crow_list[0], crow_list[1]}, plan,
                                      bit_arena, flags, n_val, vote_key, vote_block, vote_slot, gates});
}
#endif

// ------------------------------------------------------------------ participation flags, one wave per committee
#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_participation_tables(const AttRow* __restrict__ rows, TablesDev tables, const uint32_t* __restrict__ cs0,
                       const uint32_t* __restrict__ cs1, const uint32_t* __restrict__ cl0,
                       const uint32_t* __restrict__ cl1, const AttPlan* __restrict__ plan,
                       const uint32_t* __restrict__ bit_arena, const uint16_t* __restrict__ eff_increments,
                       unsigned long long base_reward_per_increment, uint32_t* __restrict__ part_cur,
                       uint32_t* __restrict__ part_prev, unsigned long long* __restrict__ numerators,
                       const uint32_t* __restrict__ gates, uint32_t cap)
{
    const int t = blockIdx.y;
    // more groups than the caller's arrays hold: k_att_validate_state reported ERR_CAPACITY, nothing is applied and
    // numerators[] (sized by cap) is not written
    if (plan->n_groups > cap) return;
    if (!tables.t[t].valid || plan->n_rows_table[t] == 0) return;
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= tables.t[t].n_committees) return;
    const uint32_t* crow_start = t ? cs1 : cs0;
    const uint32_t* crow_list = t ? cl1 : cl0;
    const uint32_t kb = crow_start[c], ke = crow_start[c + 1];
    if (kb == ke) return;
    const int lane = threadIdx.x & 63;
    const uint32_t* members = tables.t[t].members;
    // the committee's rows in batch order (ascending group id): a selection loop, the lists hold one or a few rows
    uint32_t last = NONE32;  // NONE32 + 1 == 0
    for (uint32_t done = kb; done < ke; ++done) {
        uint32_t g = NONE32;
        for (uint32_t k = kb; k < ke; ++k) {
            const uint32_t x = crow_list[k];
            if ((last == NONE32 || x > last) && x < g) g = x;
        }
        last = g;
        const AttRow r = rows[g];
        unsigned long long num = 0;
        if (gates[r.gate] == 0) {
            uint32_t* part = r.which ? part_prev : part_cur;
            for (uint32_t i = lane; i < r.n_bits; i += 64) {
                const uint32_t word = bit_arena[r.bits_word + (i >> 5)];
                if (!((word >> (i & 31)) & 1u)) continue;
                const uint32_t v = members[r.member_base + i];
                uint8_t* pb = reinterpret_cast<uint8_t*>(part) + v;
                const uint32_t old = *pb;
                const uint32_t fresh = r.flag_mask & ~old & 0x7u;
                if (fresh) {
                    *pb = (uint8_t)(old | r.flag_mask);
                    // PARTICIPATION_FLAG_WEIGHTS = [14, 26, 14] (Appendix A.9)
                    const uint32_t wsum = ((fresh & 1u) ? 14u : 0u) + ((fresh & 2u) ? 26u : 0u) + ((fresh & 4u) ? 14u : 0u);
                    num += (unsigned long long)eff_increments[v] * base_reward_per_increment * wsum;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) num += __shfl_xor(num, off, 64);
        if (lane == 0) numerators[g] = num;
        // the next row of this committee may touch the same validators: this wave's byte stores must be visible to its
        // own later loads (same wave, program order: they are)
    }
}

void launch_participation_tables(hipStream_t s, const AttRow* rows, TablesDev tables, uint32_t* const crow_start[2],
                                 uint32_t* const crow_list[2], const AttPlan* plan, const uint32_t* bit_arena,
                                 const uint16_t* eff_increments, uint64_t base_reward_per_increment,
                                 uint32_t* part_cur_words, uint32_t* part_prev_words, uint64_t* numerators,
                                 const uint32_t* gates, uint32_t cap)
{
    uint32_t nc = tables.t[0].valid ? tables.t[0].n_committees : 0u;
    if (tables.t[1].valid) nc = nc > tables.t[1].n_committees ? nc : tables.t[1].n_committees;
    if (nc == 0) return;
    const unsigned ny = tables.t[1].valid ? 2u : 1u;
    hipLaunchKernelGGL(k_participation_tables, dim3((nc + 3) / 4, ny), dim3(256), 0, s, rows, tables, crow_start[0],
                       crow_start[1], crow_list[0], crow_list[1], plan, bit_arena, eff_increments,
                       (unsigned long long)base_reward_per_increment, part_cur_words, part_prev_words,
                       reinterpret_cast<unsigned long long*>(numerators), gates, cap);
}
#endif

}  // namespace posevo
