// fc_kernels.hip -- LMD-GHOST fork-choice and attestation bookkeeping kernels for gfx950.
//
//   k_votes          get_latest_attesting_balance's O(V) part (called from
//                    get_head pe:1116): stream vote/balance/flags (13 B per validator),
//                    LDS-privatised u64 histogram per workgroup, non-zero bins flushed
//                    with one global atomic each; also the active-balance totals the
//                    proposer boost needs.
//   k_tree           one workgroup, block tree resident in LDS in DFS pre-order:
//                    subtree weight = prefix-sum difference (pe:322: "B or descendants
//                    of B"), filter_block_tree viability (A.3) from a second scan,
//                    best child by (weight, root) (pe:1114-1116), then pointer jumping
//                    replaces the sequential descent of pe:1107-1116.
//   k_lmd_*          update_latest_messages (pe:1435-1441) for a whole batch with the
//                    sequential semantics kept by a 64-bit atomicMax on (epoch+1, ~order).
//   k_participation  the flag loop of process_attestation (pe:744-749).
//   k_bits_union     aggregation_bits = OR (validator guide; pe:715, pe:730), popcount by
//                    wave reduction.
//
// All of this is HBM/latency-bound integer work; no MFMA.
#include <cstdlib>
#include "kernels.h"

namespace posevo {

constexpr uint32_t VAL_ACTIVE = 0x01u, VAL_SLASHED = 0x02u, VAL_EQUIVOCATING = 0x04u;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The kernels in front of a head (union, LMD, votes, tree) are latency-critical guests beside the long G1 kernels of
// the previous step; the SIMD's issue arbiter serves the oldest wave first, i.e. the G1 kernel's.  Raising their wave
// priority puts them first (POSEVO_FC_PRIO=0 builds without it, for comparison).
#ifndef POSEVO_FC_PRIO_LEVEL
#define POSEVO_FC_PRIO_LEVEL 3
#endif
#if POSEVO_FC_PRIO_LEVEL > 0
#define POSEVO_FC_PRIO() __builtin_amdgcn_s_setprio(POSEVO_FC_PRIO_LEVEL)
#else
#define POSEVO_FC_PRIO() ((void)0)
#endif

// ------------------------------------------------------------------ votes
constexpr int VOTES_WG = 512;  // 256 lanes: 21 us, 512: 15.8, 1024: 15.4 at 1 M validators / 64 workgroups
constexpr int VOTES_PER_THREAD = 4;

// QUADS = quads of validators in flight per lane and iteration.  2: the stand-alone get_head (63 VGPRs, fastest).
// 1: the lean form for pipelined steps, where this kernel runs BESIDE k_g1_accumulate of the previous aggregate:
// that kernel's two waves per SIMD hold 464 of the 512 registers, and only a wave of <= 48 fits into the rest and can
// start before the accumulation's last workgroup retires.
// (body: block index and grid size are arguments, hist = the workgroup's n_blocks bins of LDS -- the same code runs as a
// kernel of its own and as one block range of a paired launch, pair_kernels.hip)
template <int QUADS>
__device__ __forceinline__ void votes_body(const uint32_t bid, const uint32_t nblk, const VotesArgs& a,
                                           unsigned long long* __restrict__ hist)
{
    const uint32_t* __restrict__ vote_block = a.vote_block;
    const uint64_t* __restrict__ eff_balance = a.eff_balance;
    const uint8_t* __restrict__ flags = a.flags;
    const uint64_t n_val = a.n_val;
    const uint32_t filter_slashed = a.filter_slashed;
    const uint32_t* __restrict__ pos_of_idx = a.pos_of_idx;
    const uint32_t n_blocks = a.n_blocks;
    unsigned long long* __restrict__ direct = reinterpret_cast<unsigned long long*>(a.direct);
    VoteTotals* __restrict__ totals = a.totals;
    const uint32_t* __restrict__ vote_slot = a.vote_slot;
    const uint32_t min_vote_slot = a.min_vote_slot;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += VOTES_WG) hist[b] = 0;
    __syncthreads();

    unsigned long long act_bal = 0;
    uint32_t act_num = 0;
    const uint64_t n_quads = (n_val + VOTES_PER_THREAD - 1) / VOTES_PER_THREAD;
    const uint64_t stride = (uint64_t)nblk * VOTES_WG;
    // QUADS quads (4 validators each: 16 + 32 + 4 bytes of vector loads) in flight per lane per iteration
    for (uint64_t q0 = (uint64_t)bid * VOTES_WG + threadIdx.x; q0 < n_quads; q0 += QUADS * stride) {
        uint32_t vb[4 * QUADS];
        unsigned long long bal[4 * QUADS];
        uint32_t fl[4 * QUADS];
#pragma unroll
        for (int u = 0; u < QUADS; ++u) {
            const uint64_t q = q0 + u * stride;
            const uint64_t v0 = q * VOTES_PER_THREAD;
            if (q < n_quads && v0 + 4 <= n_val) {  // arrays are 16-byte aligned and v0 % 4 == 0
                const uint4 t = *reinterpret_cast<const uint4*>(vote_block + v0);
                vb[4 * u] = t.x; vb[4 * u + 1] = t.y; vb[4 * u + 2] = t.z; vb[4 * u + 3] = t.w;
                const ulonglong2 b01 = *reinterpret_cast<const ulonglong2*>(eff_balance + v0);
                const ulonglong2 b23 = *reinterpret_cast<const ulonglong2*>(eff_balance + v0 + 2);
                bal[4 * u] = b01.x; bal[4 * u + 1] = b01.y; bal[4 * u + 2] = b23.x; bal[4 * u + 3] = b23.y;
                const uint32_t f = *reinterpret_cast<const uint32_t*>(flags + v0);
                fl[4 * u] = f & 0xff; fl[4 * u + 1] = (f >> 8) & 0xff; fl[4 * u + 2] = (f >> 16) & 0xff; fl[4 * u + 3] = f >> 24;
                if (vote_slot) {  // vote-expiry variant (RLMD-GHOST, pe:1585-1596): an expired message counts as none
                    const uint4 sl = *reinterpret_cast<const uint4*>(vote_slot + v0);
                    if (sl.x < min_vote_slot) vb[4 * u] = NONE32;
                    if (sl.y < min_vote_slot) vb[4 * u + 1] = NONE32;
                    if (sl.z < min_vote_slot) vb[4 * u + 2] = NONE32;
                    if (sl.w < min_vote_slot) vb[4 * u + 3] = NONE32;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = q < n_quads && v0 + k < n_val;
                    vb[4 * u + k] = ok ? vote_block[v0 + k] : NONE32;
                    bal[4 * u + k] = ok ? eff_balance[v0 + k] : 0ull;
                    fl[4 * u + k] = ok ? flags[v0 + k] : 0u;
                    if (ok && vote_slot && vote_slot[v0 + k] < min_vote_slot) vb[4 * u + k] = NONE32;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4 * QUADS; ++k) {
            if (!(fl[k] & VAL_ACTIVE)) continue;
            act_bal += bal[k];
            act_num += 1;
            if (fl[k] & VAL_EQUIVOCATING) continue;
            if (filter_slashed && (fl[k] & VAL_SLASHED)) continue;
            if (vb[k] >= n_blocks) continue;  // NONE32 = no latest message
            atomicAdd(&hist[vb[k]], bal[k]);  // ds_add_u64
        }
    }
    // active-balance totals: wave reduce -> LDS -> ONE plain store per workgroup into its own slot
    // (same-address device-scope atomics cost ~12 ns each and serialise: 4096 of them were 50 us)
    __shared__ unsigned long long wg_tot[2];
    if (threadIdx.x == 0) { wg_tot[0] = 0; wg_tot[1] = 0; }
    act_bal = wave_sum_u64(act_bal);
    act_num = wave_sum_u32(act_num);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && act_num) {
        atomicAdd(&wg_tot[0], act_bal);
        atomicAdd(&wg_tot[1], (unsigned long long)act_num);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[bid].total_active_balance = wg_tot[0];
        totals[bid].num_active = wg_tot[1];
    }
    // The slots no workgroup of THIS launch owns must read zero: after an in-place all-reduce they hold the other ranks'
    // sums of the previous round (ranks with unequal shards launch different grids), and a shrinking registry leaves
    // stale partials behind (the sharded head double-counted them).  Workgroup 0 clears them: no memset.
    if (bid == 0)
        for (uint32_t j = nblk + threadIdx.x; j < (uint32_t)VOTES_MAX_WG; j += VOTES_WG) {
            totals[j].total_active_balance = 0;
            totals[j].num_active = 0;
        }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += VOTES_WG) {
        const unsigned long long w = hist[b];
        if (w) atomicAdd(&direct[pos_of_idx[b]], w);
    }
}
// workgroups of a votes launch over n_val validators
inline unsigned votes_blocks(uint64_t n_val)
{
    const uint64_t n_quads = (n_val + VOTES_PER_THREAD - 1) / VOTES_PER_THREAD;
    uint64_t blocks = (n_quads + VOTES_WG - 1) / VOTES_WG;
    // Grid-stride over few, fat workgroups: every workgroup zeroes, scans and flushes a whole n_blocks histogram, so
    // at 1 M validators 64 workgroups (16 K validators each) beat 256 (k_votes 16 us vs 23 us, get_head p50 46 vs 53 us);
    // the count grows with the registry up to one workgroup per CU (launch_votes has the measured alternatives).
    uint64_t cap = n_val / 16384;
    cap = cap < 64 ? 64 : cap > (uint64_t)VOTES_MAX_WG ? (uint64_t)VOTES_MAX_WG : cap;
    return (unsigned)(blocks > cap ? cap : blocks);
}

#ifndef POSEVO_BODIES_ONLY
template <int QUADS>
__global__ void __launch_bounds__(VOTES_WG)
k_votes(const VotesArgs a)
{
    POSEVO_FC_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];  // n_blocks bins, by insertion index
    votes_body<QUADS>(blockIdx.x, gridDim.x, a, hist);
}

void launch_votes(hipStream_t s, const VotesArgs& a, int lean)
{
    launch_votes(s, a.vote_block, a.eff_balance, a.flags, a.n_val, a.filter_slashed, a.pos_of_idx, a.n_blocks, a.direct,
                 a.totals, 0, a.vote_slot, a.min_vote_slot, lean);
}
void launch_votes(hipStream_t s, const uint32_t* vote_block, const uint64_t* eff_balance, const uint8_t* flags,
                  uint64_t n_val, uint32_t filter_slashed, const uint32_t* pos_of_idx, uint32_t n_blocks,
                  uint64_t* direct, VoteTotals* totals, int zero_first, const uint32_t* vote_slot, uint32_t min_vote_slot,
                  int lean)
{
    if (zero_first) {  // caller-owned exchange buffer; the engine's own buffer is kept zeroed by k_tree
        (void)hipMemsetAsync(direct, 0, sizeof(uint64_t) * n_blocks, s);
        (void)hipMemsetAsync(totals, 0, sizeof(VoteTotals) * VOTES_MAX_WG, s);
    }
    if (n_val == 0) return;
    const uint64_t n_quads = (n_val + VOTES_PER_THREAD - 1) / VOTES_PER_THREAD;
    uint64_t blocks = (n_quads + VOTES_WG - 1) / VOTES_WG;
    // Grid-stride over few, fat workgroups: every workgroup zeroes, scans and flushes a whole n_blocks histogram, so
    // at 1 M validators 64 workgroups (16 K validators each) beat 256 (k_votes 16 us vs 23 us, get_head p50 46 vs 53 us).
    // Round 3 re-measured the alternatives against the 33.7 us p50 of this shape: eight flush rows (workgroup w adds into
    // row w % 8, k_tree sums the rows) with 256 workgroups 40.8 us; four quads in flight per lane 30.0 vs 30.2 us (nothing);
    // k_tree launched BESIDE k_votes on a second stream and released by a device-side ticket 70 us -- the cross-stream
    // event that keeps the next call ordered costs more than the launch gap it removes (profiles/README.md);
    // the count grows with the registry up to one workgroup per CU.
    uint64_t cap = n_val / 16384;
    cap = cap < 64 ? 64 : cap > (uint64_t)VOTES_MAX_WG ? (uint64_t)VOTES_MAX_WG : cap;
    if (blocks > cap) blocks = cap;
    if (first_use_on_this_device<0>()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_votes<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(uint64_t) * TREE_MAX_BLOCKS));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_votes<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(uint64_t) * TREE_MAX_BLOCKS));
    }
    const VotesArgs a{vote_block, eff_balance, flags, n_val, filter_slashed, pos_of_idx, n_blocks, direct, totals, vote_slot,
                      min_vote_slot};
    if (lean)
        hipLaunchKernelGGL(k_votes<1>, dim3((unsigned)blocks), dim3(VOTES_WG), sizeof(uint64_t) * n_blocks, s, a);
    else
        hipLaunchKernelGGL(k_votes<2>, dim3((unsigned)blocks), dim3(VOTES_WG), sizeof(uint64_t) * n_blocks, s, a);
}
#endif

// ------------------------------------------------------------------ tree
// (A fused get_head -- votes phase in every workgroup, tree phase in the last one to pass a device-scope ticket --
// was built and measured: p50 46 us against 33 us for the two launches below.  The fat 1024-lane / 147 KB-LDS
// workgroups slow the votes phase by more than the saved launch gap; dropped.)
// The kernel is instantiated for a few items-per-thread counts: 1024 lanes x PER >= n, picked by the host from the
// block count.  The per-item loops, not the ~25 barriers, set the time: a 4096-block tree measures 19.8 us at
// 1024 x 4, 26 us at 512 x 8 or 1024 x 8 (half the lanes idle) and 34 us at 256 x 16.
//
// LDS index skew: thread t owns items PER*t .. PER*t+PER-1, i.e. a lane stride of PER elements = a PER-way (u32) bank
// conflict on every own-item access.  i -> i + i/PER turns the stride into PER+1 (odd): conflict-free.
template <int PER>
__host__ __device__ __forceinline__ uint32_t SKT(uint32_t i) { return PER == 1 ? i : i + i / PER; }
// largest skewed index over the shapes in use: 8192 blocks at PER = 8 (PER = 4 only serves n <= 4096)
constexpr uint32_t TREE_LDS_ENTRIES = TREE_MAX_BLOCKS + 2 + ((TREE_MAX_BLOCKS + 2) >> 3) + 1;

// Exclusive prefix sum over n <= 8192 values held as 8 consecutive items per thread.
// out[SK(i)] (LDS, n+1 entries) = sum of in[0..i); wave shuffles + one LDS hop across the 16 waves.
template <typename T, int TREE_WG, int TREE_PER_THREAD>
__device__ __forceinline__ void block_exclusive_scan(const T (&item)[TREE_PER_THREAD], T* out, T* wave_tot, uint32_t n)
{
    auto SK = [](uint32_t i) { return SKT<TREE_PER_THREAD>(i); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T local = 0;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) local += item[k];
    T incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    T base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    T run = base + incl - local;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        if (i <= n) out[SK(i)] = run;
        run += item[k];
    }
    if (tid == TREE_WG - 1 && (uint32_t)(TREE_WG * TREE_PER_THREAD) <= n) out[SK(n)] = run;
    __syncthreads();
}

// Two exclusive prefix sums (u64 weights, u32 leaf counts) through ONE pair of barriers.
template <int TREE_WG, int TREE_PER_THREAD>
__device__ __forceinline__ void block_exclusive_scan2(const unsigned long long (&a)[TREE_PER_THREAD], unsigned long long* out_a,
                                                      unsigned long long* wave_a, const uint32_t (&b)[TREE_PER_THREAD],
                                                      uint32_t* out_b, uint32_t* wave_b, uint32_t n)
{
    auto SK = [](uint32_t i) { return SKT<TREE_PER_THREAD>(i); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long la = 0;
    uint32_t lb = 0;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) { la += a[k]; lb += b[k]; }
    unsigned long long ia = la;
    uint32_t ib = lb;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long oa = __shfl_up(ia, off, 64);
        const uint32_t ob = __shfl_up(ib, off, 64);
        if (lane >= off) { ia += oa; ib += ob; }
    }
    if (lane == 63) { wave_a[wave] = ia; wave_b[wave] = ib; }
    __syncthreads();
    unsigned long long base_a = 0;
    uint32_t base_b = 0;
    for (int w = 0; w < wave; ++w) { base_a += wave_a[w]; base_b += wave_b[w]; }
    unsigned long long ra = base_a + ia - la;
    uint32_t rb = base_b + ib - lb;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        if (i <= n) { out_a[SK(i)] = ra; out_b[SK(i)] = rb; }
        ra += a[k];
        rb += b[k];
    }
    if (tid == TREE_WG - 1 && (uint32_t)(TREE_WG * TREE_PER_THREAD) <= n) { out_a[SK(n)] = ra; out_b[SK(n)] = rb; }
    __syncthreads();
}

// LEAN: the form that fits beside a running k_g1_accumulate (one wave of 232 VGPRs per SIMD in streaming steps leaves
// 280 registers per SIMD lane -- rounds 2-3: two waves of 168, 176 left; a 512-lane workgroup = two waves per SIMD of <= 88
// here, where the 1024-lane shapes take 4 x 84).  Parent, rank and block index are loaded where they
// are used instead of up front -- three more round trips to L2, ~6 us -- so only pipelined calls use it.
// lds_entries: skewed index range of this launch (the host sizes the dynamic LDS for the block count at hand, not for
// the 8192-block maximum: 82 KB at 4096 blocks leaves room for another kernel's workgroup on the CU).
template <int TREE_WG, int TREE_PER_THREAD, bool LEAN = false>
__device__ __forceinline__ void tree_body(const TreeArgs& a, const uint32_t lds_entries, unsigned char* smem)  // ONE workgroup
{
    const TreeDev& tree = a.tree;
    unsigned long long* __restrict__ direct = reinterpret_cast<unsigned long long*>(a.direct);
    const VoteTotals* __restrict__ totals = a.totals;
    const unsigned long long ov_balance = a.ov_balance, ov_num = a.ov_num;
    const int use_override = a.use_override;
    const uint32_t justified_pos = a.justified_pos, boost_pos = a.boost_pos;
    const unsigned long long slots_per_epoch = a.slots_per_epoch, boost_percent = a.boost_percent,
                             balance_increment = a.balance_increment;
    unsigned long long* __restrict__ weights_by_idx = reinterpret_cast<unsigned long long*>(a.weights_by_idx);
    uint32_t* __restrict__ head_idx = a.head_idx;
    const int clear_direct = a.clear_direct;
    // LDS plan (n <= 8192, skewed indices):  S u64 (later bestW) | L u32 (later bestRank) | jump u32 | scratch
    // u64 regions first so that every 64-bit LDS access (ds_*_b64, ds_add_u64, ds_max_u64) is 8-byte aligned
    unsigned long long* S = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* wave_tot64 = S + lds_entries;         // 16 waves
    unsigned long long* tot = wave_tot64 + 16;                // 2 words
    uint32_t* L = reinterpret_cast<uint32_t*>(tot + 2);
    uint32_t* jump = L + lds_entries;
    uint32_t* wave_tot32 = jump + lds_entries;                // 16 waves

    const uint32_t n = tree.n;
    const int tid = threadIdx.x;
    auto SK = [](uint32_t i) { return SKT<TREE_PER_THREAD>(i); };

    // every global read up front: one round trip of memory latency for the whole kernel
    unsigned long long w_item[TREE_PER_THREAD];
    uint32_t l_item[TREE_PER_THREAD], sz[TREE_PER_THREAD], par_g[TREE_PER_THREAD], rk_g[TREE_PER_THREAD],
        idx_g[TREE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        const bool in = i < n;
        sz[k] = in ? tree.size[i] : 0u;
        w_item[k] = in ? direct[i] : 0ull;
        l_item[k] = (in && tree.leaf_ok[i]) ? 1u : 0u;
        if (!LEAN) {
            par_g[k] = in ? tree.parent[i] : NONE32;
            rk_g[k] = in ? tree.rank[i] + 1 : 0u;  // 0 = "no viable child yet"
            idx_g[k] = in ? tree.idx_of_pos[i] : 0u;
        }
    }
    unsigned long long t_bal = 0, t_num = 0;
    if (boost_pos != NONE32 && !use_override) {
        for (int j = tid; j < VOTES_MAX_WG; j += TREE_WG) {
            t_bal += totals[j].total_active_balance;
            t_num += totals[j].num_active;
        }
    }
    if (clear_direct) {  // leave the engine's weight buffer zeroed for the next get_head (no memset launch)
#pragma unroll
        for (int k = 0; k < TREE_PER_THREAD; ++k) {
            const uint32_t i = tid * TREE_PER_THREAD + k;
            if (i < n) direct[i] = 0ull;
        }
    }
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) l_item[k] = (l_item[k] && sz[k] == 1) ? 1u : 0u;

    // proposer boost (A.1): one extra "vote" of proposer_score at the boosted block -- it then
    // counts for that block and every ancestor, exactly the get_ancestor(...) == root test.
    if (boost_pos != NONE32) {
        // sum the votes kernel's per-workgroup partial totals (all ranks' partials after the all-reduce)
        if (tid == 0) { tot[0] = 0; tot[1] = 0; }
        __syncthreads();
        if (t_num) { atomicAdd(&tot[0], t_bal); atomicAdd(&tot[1], t_num); }
        __syncthreads();
        unsigned long long total = use_override ? ov_balance : tot[0];
        const unsigned long long num = use_override ? ov_num : tot[1];
        unsigned long long boost = 0;
        if (num > 0) {
            if (total < balance_increment) total = balance_increment;  // get_total_balance's max()
            const unsigned long long avg_balance = total / num;
            const unsigned long long committee_size = num / slots_per_epoch;
            const unsigned long long committee_weight = committee_size * avg_balance;
            // committee_weight * boost_percent can exceed 64 bits only beyond 1.8e17 Gwei * percent: split
            const unsigned long long q = committee_weight / 100, r = committee_weight % 100;
            boost = q * boost_percent + (r * boost_percent) / 100;
        }
#pragma unroll
        for (int k = 0; k < TREE_PER_THREAD; ++k)
            if ((uint32_t)(tid * TREE_PER_THREAD + k) == boost_pos) w_item[k] += boost;
    }

    block_exclusive_scan2<TREE_WG, TREE_PER_THREAD>(w_item, S, wave_tot64, l_item, L, wave_tot32, n);

    unsigned long long W[TREE_PER_THREAD];
    uint32_t viable = 0;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        W[k] = 0;
        if (i < n) {
            W[k] = S[SK(i + sz[k])] - S[SK(i)];
            if (L[SK(i + sz[k])] - L[SK(i)] > 0) viable |= 1u << k;
            weights_by_idx[LEAN ? tree.idx_of_pos[i] : idx_g[k]] = W[k];
        }
    }
    __syncthreads();
    // best child, pass 1: max weight among viable children
    unsigned long long* bestW = S;
    uint32_t* bestRank = L;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        if (i < n) { bestW[SK(i)] = 0; bestRank[SK(i)] = 0; }
    }
    __syncthreads();
    uint32_t par[TREE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        par[k] = ((viable >> k) & 1u) ? (LEAN ? tree.parent[i] : par_g[k]) : NONE32;  // viable => i < n
        if (par[k] != NONE32) atomicMax(&bestW[SK(par[k])], W[k]);
    }
    __syncthreads();
    // pass 2: among the heaviest, the lexicographically highest root
    if (LEAN) {
#pragma unroll
        for (int k = 0; k < TREE_PER_THREAD; ++k) {
            const uint32_t i = tid * TREE_PER_THREAD + k;
            rk_g[k] = par[k] != NONE32 ? tree.rank[i] + 1 : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k)
        if (par[k] != NONE32 && W[k] == bestW[SK(par[k])]) atomicMax(&bestRank[SK(par[k])], rk_g[k]);
    __syncthreads();
    // The descent of pe:1107-1116 without walking: the head is the deepest node of the chain justified root -> best child
    // -> best child ...  A node lies on that chain iff no ancestor-or-self below the justified root fails to be its
    // parent's best child.  In pre-order a subtree is the interval [pos, pos + size), so with g(u) = 1 for every node
    // strictly inside the justified subtree that is NOT the best child of its parent (0 elsewhere), the number of such
    // ancestors-or-self of position p is the prefix sum of  D[pos(u)] += g(u), D[pos(u) + size(u)] -= g(u)  at p --
    // one more block scan instead of up to log2(n) rounds of pointer jumping with two barriers each.  Chain nodes have
    // count 0, their pre-order positions grow downwards: the head is the largest such position inside the subtree.
    const uint32_t j_end = justified_pos + tree.size[justified_pos];
    uint32_t* D = jump;            // n + 1 signed counters
    uint32_t g_item[TREE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        const bool best = par[k] != NONE32 && W[k] == bestW[SK(par[k])] && rk_g[k] == bestRank[SK(par[k])];
        g_item[k] = (i < n && i > justified_pos && i < j_end && !best) ? 1u : 0u;
        if (i <= n) D[SK(i)] = g_item[k];   // entry n: only ever decremented
    }
    if (tid == TREE_WG - 1 && (uint32_t)(TREE_WG * TREE_PER_THREAD) <= n) D[SK(n)] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k)
        if (g_item[k]) atomicSub(&D[SK(tid * TREE_PER_THREAD + k + sz[k])], 1u);
    __syncthreads();
    uint32_t d_item[TREE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        d_item[k] = i < n ? D[SK(i)] : 0u;
    }
    uint32_t* C = L;  // bestRank's region: every lane has read it above (the scan's first barrier orders the reuse)
    block_exclusive_scan<uint32_t, TREE_WG, TREE_PER_THREAD>(d_item, C, wave_tot32, n);
    uint32_t best_pos = justified_pos;
#pragma unroll
    for (int k = 0; k < TREE_PER_THREAD; ++k) {
        const uint32_t i = tid * TREE_PER_THREAD + k;
        if (i > justified_pos && i < j_end && i < n && C[SK(i)] + d_item[k] == 0u && ((viable >> k) & 1u)) best_pos = max(best_pos, i);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best_pos = max(best_pos, (uint32_t)__shfl_xor((int)best_pos, off, 64));
    uint32_t* head_pos = wave_tot32;  // reused: the scan is through with it
    __syncthreads();
    if (tid == 0) head_pos[0] = justified_pos;
    __syncthreads();
    if ((tid & 63) == 0) atomicMax(&head_pos[0], best_pos);
    __syncthreads();
    if (tid == 0)  // host-coherent pinned word polled by the host: system-scope release
        __hip_atomic_store(head_idx, tree.idx_of_pos[head_pos[0]], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// dynamic LDS of a tree launch: skewed index of the last entry (n) of the shape, + 1, rounded up to keep the u32 regions
// 16-byte aligned; the bytes that go with it; the most any launch asks for (the opt-in beyond 64 KiB)
template <int PER>
inline uint32_t tree_lds_entries(uint32_t n) { return std::min<uint32_t>(TREE_LDS_ENTRIES, ((SKT<PER>(n + 1) + 2) + 3u) & ~3u); }
inline size_t tree_lds_bytes(uint32_t entries) { return sizeof(uint64_t) * ((size_t)entries + 18) + sizeof(uint32_t) * (2 * (size_t)entries + 16); }
constexpr size_t TREE_LDS_MAX = sizeof(uint64_t) * (TREE_LDS_ENTRIES + 18) + sizeof(uint32_t) * (2 * TREE_LDS_ENTRIES + 16);

#ifndef POSEVO_BODIES_ONLY
template <int TREE_WG, int TREE_PER_THREAD, bool LEAN = false>
__global__ void __launch_bounds__(TREE_WG)
k_tree(const TreeArgs a, const uint32_t lds_entries)
{
    POSEVO_FC_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tree_body<TREE_WG, TREE_PER_THREAD, LEAN>(a, lds_entries, smem);
}

template <int WG, int PER, bool LEAN = false>
static void launch_tree_shape(hipStream_t s, const TreeArgs& a)
{
    if (first_use_on_this_device<1000 + WG + PER + (LEAN ? 5000 : 0)>()) {  // > 64 KiB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree<WG, PER, LEAN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)TREE_LDS_MAX);
    }
    const uint32_t entries = tree_lds_entries<PER>(a.tree.n);
    hipLaunchKernelGGL((k_tree<WG, PER, LEAN>), dim3(1), dim3(WG), tree_lds_bytes(entries), s, a, entries);
}

void launch_tree(hipStream_t s, const TreeArgs& a, int lean)
{
    const uint32_t n = a.tree.n;
    if (n <= 1024) launch_tree_shape<1024, 1>(s, a);  // 40 VGPRs x 4 waves per SIMD: lean as it is
    else if (lean && n <= 2048) launch_tree_shape<512, 4, true>(s, a);
    else if (lean && n <= 4096) launch_tree_shape<512, 8, true>(s, a);
    else if (n <= 2048) launch_tree_shape<1024, 2>(s, a);
    else if (n <= 4096) launch_tree_shape<1024, 4>(s, a);
    else launch_tree_shape<1024, 8>(s, a);
}
void launch_tree(hipStream_t s, const TreeDev& tree, uint64_t* direct, const VoteTotals* totals,
                 uint64_t totals_override_balance, uint64_t totals_override_num, int use_override,
                 uint32_t justified_pos, uint32_t boost_pos, uint64_t slots_per_epoch, uint64_t boost_percent,
                 uint64_t balance_increment, uint64_t* weights_by_idx, uint32_t* head_idx, int clear_direct, int lean)
{
    launch_tree(s, TreeArgs{tree, direct, totals, totals_override_balance, totals_override_num, use_override, justified_pos,
                            boost_pos, slots_per_epoch, boost_percent, balance_increment, weights_by_idx, head_idx,
                            clear_direct}, lean);
}
#endif

#ifndef POSEVO_BODIES_ONLY
// ------------------------------------------------------------------ LMD update
// One wave per attestation; lane l walks bit words l, l+64, ...
__device__ __forceinline__ unsigned long long lmd_key(uint32_t epoch_p1, uint32_t order)
{
    return ((unsigned long long)epoch_p1 << 32) | (unsigned long long)(0xFFFFFFFEu - order);
}

template <int PHASE>
__global__ void __launch_bounds__(256)
k_lmd(const AttRow* __restrict__ rows, uint32_t n_rows, const uint32_t* __restrict__ members,
      const uint32_t* __restrict__ bit_arena, const uint8_t* __restrict__ flags,
      unsigned long long* __restrict__ vote_key, uint32_t* __restrict__ vote_block,
      uint32_t* __restrict__ vote_slot, const uint32_t* __restrict__ gates)
{
    const uint32_t a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const AttRow r = rows[a];
    if (gates && r.gate != NONE32 && gates[r.gate] != 0) return;  // voided on the device (overlapping members)
    const unsigned long long key = lmd_key(r.epoch_p1, r.order);
    // lane-parallel over bit positions: every lane tests its own bit (the 32 lanes sharing a word hit one line)
    for (uint32_t i = lane; i < r.n_bits; i += 64) {
        const uint32_t word = bit_arena[r.bits_word + (i >> 5)];
        if (!((word >> (i & 31)) & 1u)) continue;
        const uint32_t v = members[r.member_base + i];
        if (PHASE == 0) {
            if (flags[v] & VAL_EQUIVOCATING) continue;  // pe:1438
            atomicMax(&vote_key[v], key);                // strictly-later epoch wins; first in batch among equals
        } else {
            if (vote_key[v] == key) {                    // unique winner: (epoch, order) identifies one attestation
                vote_block[v] = r.block_idx;
                if (vote_slot) vote_slot[v] = r.slot;
                vote_key[v] = ((unsigned long long)r.epoch_p1 << 32) | 0xFFFFFFFFull;  // settled
            }
        }
    }
}

// Inverse of a committee table that partitions the validators: inv[v] = (committee id, index in committee).
__global__ void __launch_bounds__(256)
k_invert_committees(const uint32_t* __restrict__ members, const uint32_t* __restrict__ offsets, uint32_t n_committees,
                    uint32_t* __restrict__ inv_comm, uint32_t* __restrict__ inv_pos)
{
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_committees) return;
    const uint32_t b = offsets[c], e = offsets[c + 1];
    for (uint32_t i = b + (threadIdx.x & 63); i < e; i += 64) {
        const uint32_t v = members[i];
        inv_comm[v] = c;
        inv_pos[v] = i - b;
    }
}

void launch_invert_committees(hipStream_t s, const uint32_t* members, const uint32_t* offsets, uint32_t n_committees,
                              uint32_t* inv_comm, uint32_t* inv_pos, uint64_t n_val)
{
    (void)hipMemsetAsync(inv_comm, 0xFF, 4ull * n_val, s);
    if (n_committees == 0) return;
    hipLaunchKernelGGL(k_invert_committees, dim3((n_committees + 3) / 4), dim3(256), 0, s, members, offsets,
                       n_committees, inv_comm, inv_pos);
}

// update_latest_messages (pe:1435-1441), validator-major: one lane per validator walks the batch rows of ITS
// committee in batch order -- literally the spec's sequential loop, so no atomics and no tie-break tag are needed;
// vote/flag/key tables are streamed (coalesced) instead of hit by a million random 8-byte atomics.
// crow_start/crow_list: CSR of batch rows per committee id (rows in batch order).
__global__ void __launch_bounds__(256)
k_lmd_validator_major(const AttRow* __restrict__ rows, const uint32_t* __restrict__ crow_start,
                      const uint32_t* __restrict__ crow_list, const uint32_t* __restrict__ inv_comm,
                      const uint32_t* __restrict__ inv_pos, const uint32_t* __restrict__ bit_arena,
                      const uint8_t* __restrict__ flags, uint64_t n_val, unsigned long long* __restrict__ vote_key,
                      uint32_t* __restrict__ vote_block, uint32_t* __restrict__ vote_slot,
                      const uint32_t* __restrict__ gates)
{
    POSEVO_FC_PRIO();
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_val) return;
    const uint32_t c = inv_comm[v];
    if (c == NONE32) return;
    const uint32_t kb = crow_start[c], ke = crow_start[c + 1];
    if (kb == ke) return;
    if (flags[v] & VAL_EQUIVOCATING) return;  // pe:1438
    const uint32_t i = inv_pos[v];
    uint32_t epoch_p1 = (uint32_t)(vote_key[v] >> 32);  // 0 = no latest message
    uint32_t new_block = NONE32, new_slot = 0;
    for (uint32_t k = kb; k < ke; ++k) {
        const AttRow r = rows[crow_list[k]];
        if (i >= r.n_bits) continue;
        if (gates && r.gate != NONE32 && gates[r.gate] != 0) continue;  // voided on the device
        if (!((bit_arena[r.bits_word + (i >> 5)] >> (i & 31)) & 1u)) continue;
        if (r.epoch_p1 > epoch_p1) {  // "i not in latest_messages or target.epoch > latest_messages[i].epoch"
            epoch_p1 = r.epoch_p1;
            new_block = r.block_idx;
            new_slot = r.slot;
        }
    }
    if (new_block != NONE32) {
        vote_key[v] = ((unsigned long long)epoch_p1 << 32) | 0xFFFFFFFFull;
        vote_block[v] = new_block;
        if (vote_slot) vote_slot[v] = new_slot;
    }
}

void launch_lmd_validator_major(hipStream_t s, const AttRow* rows, const uint32_t* crow_start,
                                const uint32_t* crow_list, const uint32_t* inv_comm, const uint32_t* inv_pos,
                                const uint32_t* bit_arena, const uint8_t* flags, uint64_t n_val, uint64_t* vote_key,
                                uint32_t* vote_block, uint32_t* vote_slot, const uint32_t* gates)
{
    if (n_val == 0) return;
    hipLaunchKernelGGL(k_lmd_validator_major, dim3((unsigned)((n_val + 255) / 256)), dim3(256), 0, s, rows, crow_start,
                       crow_list, inv_comm, inv_pos, bit_arena, flags, n_val,
                       reinterpret_cast<unsigned long long*>(vote_key), vote_block, vote_slot, gates);
}

void launch_lmd_update(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                       const uint32_t* bit_arena, const uint8_t* flags, uint64_t* vote_key, uint32_t* vote_block,
                       uint32_t* vote_slot, const uint32_t* gates)
{
    if (n_rows == 0) return;
    const unsigned blocks = (n_rows + 3) / 4;
    hipLaunchKernelGGL(k_lmd<0>, dim3(blocks), dim3(256), 0, s, rows, n_rows, members, bit_arena, flags,
                       reinterpret_cast<unsigned long long*>(vote_key), vote_block, vote_slot, gates);
    hipLaunchKernelGGL(k_lmd<1>, dim3(blocks), dim3(256), 0, s, rows, n_rows, members, bit_arena, flags,
                       reinterpret_cast<unsigned long long*>(vote_key), vote_block, vote_slot, gates);
}

// ------------------------------------------------------------------ participation
__global__ void __launch_bounds__(256)
k_participation(const AttRow* __restrict__ rows, uint32_t n_rows, const uint32_t* __restrict__ members,
                const uint32_t* __restrict__ bit_arena, const uint16_t* __restrict__ eff_increments,
                unsigned long long base_reward_per_increment, uint32_t* __restrict__ part_cur,
                uint32_t* __restrict__ part_prev, unsigned long long* __restrict__ numerators,
                const uint32_t* __restrict__ numerator_slot, const uint32_t* __restrict__ gates)
{
    const uint32_t a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const AttRow r = rows[a];
    uint32_t* part = r.which ? part_prev : part_cur;
    unsigned long long num = 0;
    const uint32_t n_use = (gates && r.gate != NONE32 && gates[r.gate] != 0) ? 0u : r.n_bits;  // voided on the device
    for (uint32_t i = lane; i < n_use; i += 64) {
        const uint32_t word = bit_arena[r.bits_word + (i >> 5)];
        if (!((word >> (i & 31)) & 1u)) continue;
        const uint32_t v = members[r.member_base + i];
        // attestations of one round touch pairwise disjoint validators, and byte stores do not disturb the
        // neighbouring bytes: a plain byte read-modify-write is exact here (no atomics on the hot path)
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
    num = wave_sum_u64(num);
    if (lane == 0) numerators[numerator_slot[a]] = num;
}

void launch_participation(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                          const uint32_t* bit_arena, const uint16_t* eff_increments,
                          uint64_t base_reward_per_increment, uint32_t* part_cur_words, uint32_t* part_prev_words,
                          uint64_t* numerators, const uint32_t* numerator_slot, const uint32_t* gates)
{
    if (n_rows == 0) return;
    hipLaunchKernelGGL(k_participation, dim3((n_rows + 3) / 4), dim3(256), 0, s, rows, n_rows, members, bit_arena,
                       eff_increments, (unsigned long long)base_reward_per_increment, part_cur_words,
                       part_prev_words, reinterpret_cast<unsigned long long*>(numerators), numerator_slot, gates);
}

// ------------------------------------------------------------------ indexed attestations
// get_indexed_attestation (call sites pe:736, pe:975): attesting_indices =
// sorted(committee[i] for i with bits[i]).  One workgroup per attestation: compaction by a block prefix sum over the
// set bits (wave ballot + popcount), then a bitonic sort in LDS (committees hold <= 8192 members).
constexpr int IDX_WG = 256;
constexpr int IDX_MAX = 8192;

__global__ void __launch_bounds__(IDX_WG)
k_indexed_attestations(const AttRow* __restrict__ rows, uint32_t n_rows, const uint32_t* __restrict__ members,
                       const uint32_t* __restrict__ bit_arena, const uint32_t* __restrict__ out_offsets,
                       uint32_t* __restrict__ out_indices)
{
    __shared__ uint32_t keys[IDX_MAX];
    __shared__ uint32_t wave_cnt[IDX_WG / 64];
    __shared__ uint32_t base_s;
    const uint32_t a = blockIdx.x;
    if (a >= n_rows) return;
    const AttRow r = rows[a];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    // compaction, 256 bit positions per pass: ballot the set bits, rank = popcount of lower lanes + wave offsets
    for (uint32_t i0 = 0; i0 < r.n_bits; i0 += IDX_WG) {
        const uint32_t i = i0 + tid;
        bool set = false;
        if (i < r.n_bits) set = (bit_arena[r.bits_word + (i >> 5)] >> (i & 31)) & 1u;
        const unsigned long long ballot = __ballot(set);
        const uint32_t below = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __builtin_popcountll(ballot);
        __syncthreads();
        uint32_t off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (set) keys[off + below] = members[r.member_base + i];
        __syncthreads();
        if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const uint32_t count = base_s;
    uint32_t n2 = 1;
    while (n2 < count) n2 <<= 1;
    for (uint32_t i = count + tid; i < n2; i += IDX_WG) keys[i] = NONE32;  // pad: sorts to the end
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n2; i += IDX_WG) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    const uint32_t o = out_offsets[a];
    for (uint32_t i = tid; i < count; i += IDX_WG) out_indices[o + i] = keys[i];
}

void launch_indexed_attestations(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                                 const uint32_t* bit_arena, const uint32_t* out_offsets, uint32_t* out_indices)
{
    if (n_rows == 0) return;
    hipLaunchKernelGGL(k_indexed_attestations, dim3(n_rows), dim3(IDX_WG), 0, s, rows, n_rows, members, bit_arena,
                       out_offsets, out_indices);
}

// ------------------------------------------------------------------ FFG balance sums
// The three Gwei sums process_justification_and_finalization (pe:791-802) feeds to
// weigh_justification_and_finalization (pe:815-853):
//   total_active_balance     sum over validators active in the current epoch
//   previous_target_balance  sum over unslashed validators active in the previous epoch whose
//                            previous_epoch_participation has TIMELY_TARGET (get_unslashed_participating_indices)
//   current_target_balance   same for the current epoch
// One streaming pass (balance u64 + state flags u8 + two participation bytes), per-workgroup partials.
constexpr uint32_t SVAL_ACTIVE_CUR = 0x01u, SVAL_SLASHED = 0x02u, SVAL_ACTIVE_PREV = 0x08u, TIMELY_TARGET_BIT = 0x02u;

__global__ void __launch_bounds__(256)
k_ffg_balances(const uint64_t* __restrict__ balance, const uint8_t* __restrict__ sflags,
               const uint8_t* __restrict__ part_cur, const uint8_t* __restrict__ part_prev, uint64_t n_val,
               unsigned long long* __restrict__ partials /* [gridDim.x][3] */)
{
    unsigned long long tot = 0, prev = 0, cur = 0;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_val; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t f = sflags[v];
        const unsigned long long b = balance[v];
        if (f & SVAL_ACTIVE_CUR) tot += b;
        if (!(f & SVAL_SLASHED)) {
            if ((f & SVAL_ACTIVE_PREV) && (part_prev[v] & TIMELY_TARGET_BIT)) prev += b;
            if ((f & SVAL_ACTIVE_CUR) && (part_cur[v] & TIMELY_TARGET_BIT)) cur += b;
        }
    }
    __shared__ unsigned long long acc[3];
    if (threadIdx.x == 0) { acc[0] = 0; acc[1] = 0; acc[2] = 0; }
    tot = wave_sum_u64(tot);
    prev = wave_sum_u64(prev);
    cur = wave_sum_u64(cur);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], tot); atomicAdd(&acc[1], prev); atomicAdd(&acc[2], cur); }
    __syncthreads();
    if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = acc[threadIdx.x];
}

uint32_t launch_ffg_balances(hipStream_t s, const uint64_t* balance, const uint8_t* sflags, const uint8_t* part_cur,
                             const uint8_t* part_prev, uint64_t n_val, uint64_t* partials)
{
    uint64_t blocks = (n_val + 256 * 16 - 1) / (256 * 16);
    if (blocks > 256) blocks = 256;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_ffg_balances, dim3((unsigned)blocks), dim3(256), 0, s, balance, sflags, part_cur, part_prev,
                       n_val, reinterpret_cast<unsigned long long*>(partials));
    return (uint32_t)blocks;
}

// ------------------------------------------------------------------ working-state view = registry
__global__ void __launch_bounds__(256)
k_state_view_from_registry(const uint8_t* __restrict__ flags, const unsigned long long* __restrict__ balance,
                           unsigned long long increment, uint64_t n_val, uint8_t* __restrict__ sflags,
                           uint16_t* __restrict__ increments)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_val) return;
    const uint32_t f = flags[i];
    sflags[i] = (uint8_t)((f & (VAL_ACTIVE | VAL_SLASHED)) | ((f & VAL_ACTIVE) ? 0x08u : 0u));  // PE_VAL_ACTIVE_PREV
    increments[i] = (uint16_t)(balance[i] / increment);
}
void launch_state_view_from_registry(hipStream_t s, const uint8_t* flags, const uint64_t* balance, uint64_t increment,
                                     uint64_t n_val, uint8_t* sflags, uint16_t* increments)
{
    if (n_val == 0) return;
    hipLaunchKernelGGL(k_state_view_from_registry, dim3((unsigned)((n_val + 255) / 256)), dim3(256), 0, s, flags,
                       reinterpret_cast<const unsigned long long*>(balance), (unsigned long long)increment, n_val,
                       sflags, increments);
}

#endif  // POSEVO_BODIES_ONLY

// ------------------------------------------------------------------ bitfield union
// One wave per group.  The members' bitfields are read straight from the caller's arena as uploaded (byte offsets:
// an aligned dword pair + v_alignbyte_b32), so the host packs nothing.  Besides the OR and its popcount the wave sums
// the members' own popcounts: sum > popcount(OR) <=> two members share a bit (A.8: such aggregates are not merged --
// the aggregate signature would count that validator twice).
__device__ __forceinline__ void bits_union_body(const uint32_t g /* group of this WAVE */, const UnionArgs& a)
{
    const UnionGroup* __restrict__ groups = a.groups;
    const uint32_t* __restrict__ att_bytes = a.att_bytes;
    const uint8_t* __restrict__ bit_arena = a.bit_arena;
    uint32_t* __restrict__ out_arena = a.out_arena;
    uint32_t* __restrict__ out_info = a.out_info;
    uint32_t* __restrict__ host_arena = a.host_arena;
    uint32_t* __restrict__ host_info = a.host_info;
    uint32_t n_groups = a.n_groups;
    if (a.plan_dev) n_groups = a.plan_dev->n_groups;  // groups formed on the device: the grid covers an upper bound
    if (g >= n_groups) return;
    const int lane = threadIdx.x & 63;
    const UnionGroup d = groups[g];
    const uint32_t n_words = (d.n_bits + 31) >> 5;
    const uint32_t tail_mask = (d.n_bits & 31) ? ((1u << (d.n_bits & 31)) - 1u) : 0xFFFFFFFFu;
    const uint32_t* arena_w = reinterpret_cast<const uint32_t*>(bit_arena);  // staging block: 256-byte aligned
    uint32_t cnt = 0, member_sum = 0;
    for (uint32_t w = lane; w < n_words; w += 64) {
        uint32_t acc = 0;
        const uint32_t mask = (w == n_words - 1) ? tail_mask : 0xFFFFFFFFu;
        for (uint32_t k = 0; k < d.n_atts; ++k) {
            const uint32_t off = att_bytes[d.list_start + k] + 4u * w;
            const uint32_t lo = arena_w[off >> 2], hi = arena_w[(off >> 2) + 1];
            const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, off & 3u) & mask;
            member_sum += __builtin_popcount(v);
            acc |= v;
        }
        out_arena[d.out_word + w] = acc;
        if (host_arena) host_arena[d.out_word + w] = acc;
        cnt += __builtin_popcount(acc);
    }
    cnt = wave_sum_u32(cnt);
    member_sum = wave_sum_u32(member_sum);
    if (lane == 0) {
        if (out_info) {
            out_info[2 * g] = cnt;
            out_info[2 * g + 1] = member_sum - cnt;
        }
        if (host_info) {
            host_info[2 * g] = cnt;
            host_info[2 * g + 1] = member_sum - cnt;
        }
    }
}

#ifndef POSEVO_BODIES_ONLY
__global__ void __launch_bounds__(256)
k_bits_union(const UnionArgs a)
{
    POSEVO_FC_PRIO();
    bits_union_body(blockIdx.x * 4 + (threadIdx.x >> 6), a);
}

void launch_bits_union(hipStream_t s, const UnionArgs& a)
{
    if (a.n_groups == 0) return;
    hipLaunchKernelGGL(k_bits_union, dim3((a.n_groups + 3) / 4), dim3(256), 0, s, a);
}
void launch_bits_union(hipStream_t s, const UnionGroup* groups, uint32_t n_groups, const uint32_t* att_bytes,
                       const uint8_t* bit_arena, uint32_t* out_arena, uint32_t* out_info, uint32_t* host_arena,
                       uint32_t* host_info, const AttPlan* plan_dev)
{
    launch_bits_union(s, UnionArgs{groups, n_groups, att_bytes, bit_arena, out_arena, out_info, host_arena, host_info, plan_dev});
}
#endif

}  // namespace posevo
