// pair_kernels.hip -- paired launches: the fork-choice kernels of step N and the row kernels of step N + 1 of a streaming
// caller as block ranges of ONE grid each.
//
// The engine stream of a streaming step carries two chains that do not depend on each other: the handlers of the aggregate
// that has just been formed (validate_on_attestation A.4 -> update_latest_messages pe:1435-1441 -> the O(V) sum of
// get_latest_attesting_balance A.1 -> the tree pass of get_head pe:1102-1116) and the grouping of the NEXT batch of rows
// (ingest -> plan -> members -> union; pe:474, pe:659).  One behind the other they are ~250 us of latency-sized kernels and
// pace the step (the G1 sums, 160-220 us, hide under them); on two streams the command processor's queue interleaving makes
// every small kernel 3-5 x slower (five measurements).  As ONE launch per pair the two bodies run side by side
// on the CUs with no second queue: the chain becomes the sum of the pairwise maxima.
//
//   k_pair_ingest_validate   k_att_ingest(N+1)   |  k_att_validate_fc(N)      256-lane blocks
//   k_pair_plan_lmd          k_att_plan(N+1)     |  k_lmd_vm_tables(N)        256-lane blocks: the plan's first, then 256 validators each
//   k_pair_members_votes     k_att_members(N+1)  |  k_votes<lean>(N)          the votes' fat workgroups first, 512 lanes
//   k_pair_union_tree        k_bits_union(N+1)   |  k_tree<lean>(N)           block 0 = the tree's one workgroup
//
// The bodies are the stand-alone kernels' own code (att_kernels.hip / fc_kernels.hip compiled here with
// POSEVO_BODIES_ONLY: device functions that take their block index as an argument), so results cannot differ; which form
// runs is the engine's choice per step (engine_pair.cpp).  Integer / byte work, latency-bound: no MFMA.
#define POSEVO_BODIES_ONLY 1
#include "att_kernels.hip"
#include "fc_kernels.hip"
#undef POSEVO_BODIES_ONLY

namespace posevo {

// ------------------------------------------------------------------ ingest(N+1) | validate_fc(N)
__global__ void __launch_bounds__(256)
k_pair_ingest_validate(const IngestArgs ia, const ValidateFcArgs va, const uint32_t nb_ingest)
{
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x < nb_ingest) att_ingest_body(blockIdx.x, nb_ingest, ia);
    else att_validate_fc_body((blockIdx.x - nb_ingest) * 256 + threadIdx.x, va);
}

bool launch_pair_ingest_validate(hipStream_t s, const IngestArgs& ia, const ValidateFcArgs& va)
{
    if (ia.n == 0 || va.n_bound == 0) return false;
    const unsigned nb_ingest = att_ingest_blocks(ia), nb_val = (va.n_bound + 255) / 256;
    hipLaunchKernelGGL(k_pair_ingest_validate, dim3(nb_ingest + nb_val), dim3(256), 0, s, ia, va, (uint32_t)nb_ingest);
    return true;
}

// ------------------------------------------------------------------ plan(N+1) | LMD(N)
// 256-lane blocks both: the plan's workgroups first (a workgroup of the plan waits for the records of LOWER block indices
// only, att_kernels.hip), then one lane per validator.  No LDS to speak of: the LMD blocks queue for nothing (round 5's
// one-workgroup plan made each of them reserve its 70 KB).
__global__ void __launch_bounds__(PLAN_WG) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_pair_plan_lmd(const AttPlanArgs pa, const LmdVmArgs la, const uint32_t nb_plan)
{
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x < nb_plan) att_plan_body(blockIdx.x, nb_plan, pa);
    else lmd_vm_tables_body((unsigned long long)(blockIdx.x - nb_plan) * PLAN_WG + threadIdx.x, la);
}

bool launch_pair_plan_lmd(hipStream_t s, const AttPlanArgs& pa, const LmdVmArgs& la)
{
    if (la.n_val == 0) return false;
    const unsigned nb_plan = att_plan_blocks(pa), nb_lmd = (unsigned)((la.n_val + PLAN_WG - 1) / PLAN_WG);
    hipLaunchKernelGGL(k_pair_plan_lmd, dim3(nb_plan + nb_lmd), dim3(PLAN_WG), 0, s, pa, la, (uint32_t)nb_plan);
    return true;
}

// ------------------------------------------------------------------ members(N+1) | votes(N)
__global__ void __launch_bounds__(VOTES_WG)
k_pair_members_votes(const MembersArgs ma, const VotesArgs va, const uint32_t nb_votes)
{
    POSEVO_FC_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];  // the votes' n_blocks bins
    if (blockIdx.x < nb_votes) votes_body<1>(blockIdx.x, nb_votes, va, hist);
    else att_members_body((blockIdx.x - nb_votes) * VOTES_WG + threadIdx.x, ma);
}

bool launch_pair_members_votes(hipStream_t s, const MembersArgs& ma, const VotesArgs& va)
{
    if (ma.n == 0 || va.n_val == 0) return false;
    if (first_use_on_this_device<7001>())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_members_votes), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(uint64_t) * TREE_MAX_BLOCKS));
    const unsigned nb_votes = votes_blocks(va.n_val), nb_mem = (ma.n + VOTES_WG - 1) / VOTES_WG;
    hipLaunchKernelGGL(k_pair_members_votes, dim3(nb_votes + nb_mem), dim3(VOTES_WG), sizeof(uint64_t) * va.n_blocks, s, ma, va,
                       (uint32_t)nb_votes);
    return true;
}

// ------------------------------------------------------------------ union(N+1) | tree(N)
// Block 0 is the tree's workgroup (dispatched first); every union block reserves the tree's dynamic LDS as well (82 KB at
// 4096 blocks: one or two union blocks per CU, WG / 64 groups each -- 2048 groups are one round of the chip).
template <int WG, int PER, bool LEAN>
__global__ void __launch_bounds__(WG)
k_pair_union_tree(const UnionArgs ua, const TreeArgs ta, const uint32_t lds_entries)
{
    POSEVO_FC_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x == 0) tree_body<WG, PER, LEAN>(ta, lds_entries, smem);
    else bits_union_body((blockIdx.x - 1) * (WG / 64) + (threadIdx.x >> 6), ua);
}

template <int WG, int PER, bool LEAN>
static void launch_pair_union_tree_shape(hipStream_t s, const UnionArgs& ua, const TreeArgs& ta)
{
    if (first_use_on_this_device<7100 + WG + PER>())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_union_tree<WG, PER, LEAN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)TREE_LDS_MAX);
    const uint32_t entries = tree_lds_entries<PER>(ta.tree.n);
    const unsigned nb_union = (ua.n_groups + WG / 64 - 1) / (WG / 64);
    hipLaunchKernelGGL((k_pair_union_tree<WG, PER, LEAN>), dim3(1 + nb_union), dim3(WG), tree_lds_bytes(entries), s, ua, ta,
                       entries);
}

bool launch_pair_union_tree(hipStream_t s, const UnionArgs& ua, const TreeArgs& ta)
{
    if (ua.n_groups == 0) return false;
    const uint32_t n = ta.tree.n;
    // the shapes launch_tree picks for pipelined calls; a tree beyond 4096 blocks needs the 147 KB workgroup, beside which
    // no union block fits: those steps launch the two alone
    if (n <= 1024) launch_pair_union_tree_shape<1024, 1, false>(s, ua, ta);
    else if (n <= 2048) launch_pair_union_tree_shape<512, 4, true>(s, ua, ta);
    else if (n <= 4096) launch_pair_union_tree_shape<512, 8, true>(s, ua, ta);
    else return false;
    return true;
}

// The first launch of a kernel pays for loading its code object (~0.5 ms, measured inside a 20-step run: the first paired
// launch of a process took 530 us).  Resolve the pair kernels where the handle is created instead.
void pair_kernels_preload()
{
    if (!first_use_on_this_device<7999>()) return;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_ingest_validate));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_plan_lmd));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_members_votes));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_union_tree<1024, 1, false>));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_union_tree<512, 4, true>));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pair_union_tree<512, 8, true>));
    (void)hipGetLastError();
}

}  // namespace posevo
