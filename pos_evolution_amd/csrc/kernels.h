// kernels.h -- host-callable launch wrappers around the HIP kernels (internal interface
// between engine_*.cpp and the *.hip translation units; NOT the public ABI -- that is
// include/posevo.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace posevo {

// Function attributes (the opt-in to > 64 KiB of dynamic LDS) are per device: one flag per (kernel tag, device), so that
// a process driving several GPUs through several handles opts every device in.
template <int TAG>
static bool first_use_on_this_device()
{
    static bool done[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
// Registry / point-table rows: 24 Montgomery words (x | y) padded to 32 = 128 bytes, so that a gathered point is
// exactly one 128-byte memory line (96-byte rows straddle two lines half of the time: PMC showed 2x the
// algorithmic bytes fetched by k_g1_accumulate's random gather).
constexpr int G1_ROW_WORDS = 32;
constexpr int G2_WG_SLOTS = 128;        // point slots per 256-lane workgroup of the G2 kernels (a lane pair per point)
constexpr int G1_WG = 256;             // lanes (task slots) per workgroup of the accumulate kernel
constexpr int G1_LANE_PARTIAL_BYTES = 224;  // a lane's partial on its way to the tree: X | Y | ZZ | ZZZ as 14 limbs of 29 bits each
constexpr int TREE_MAX_BLOCKS = 8192;  // LDS-resident block tree capacity (K_tree)
constexpr int VOTES_MAX_WG = 256;      // workgroups of K_votes = slots of per-workgroup partial totals

// One summation group of the G1 accumulate kernel (device-resident array, built by the host).
//   sum over i in [0, n_members), bit i set (if bits_word != NONE32) of
//   points[ members ? members[member_start + i] : member_start + i ]
struct G1Group {
    uint32_t member_start;  // offset into the members array (or first point index when members == null)
    uint32_t n_members;
    uint32_t bits_word;     // u32-word offset of the group's bitfield in the device bit arena, or NONE32
    uint32_t slot_base;     // first lane slot of this group (aligned to its padded block)
    uint32_t n_tasks;       // ceil(n_members / k)
    uint32_t k;             // members per lane
    uint32_t log2_block;    // padded block = 1 << log2_block slots (<= 8); groups wider than 256 tasks use whole WGs
    uint32_t out_base;      // first slot in the wg-partials buffer; one partial per workgroup the group spans
};

// ---- G1 ----
// 96-byte big-endian uncompressed affine -> 24 u32 Montgomery limbs (x | y); infinity -> all zero.
void launch_g1_convert(hipStream_t s, const uint8_t* be96, uint32_t* mont24, uint64_t n);
// The registry in the accumulation's field form (fp381_s29.h: 14 + 14 limbs of 29 bits per 128-byte row, word 28 = the row
// holds a point), built from the 24-word Montgomery table.
void launch_g1_table_s29(hipStream_t s, const uint32_t* points_mont24, uint32_t* points_s29, uint64_t n);
// Per-lane XYZZ accumulation of k gathered points over that table: one partial per lane slot into lane_partials (limb-major
// per workgroup: ceil(n_slots / 256) * 256 * G1_LANE_PARTIAL_BYTES, the accumulation's own 14 x 29-bit limbs); single-task groups are written
// straight to wg_partials48.  plan_dev (nullable): n_groups / n_slots are read from this device-resident AttPlan instead (the
// arguments are then upper bounds that size the grid); members1: the member array of groups whose G1Group::k has bit 31 set.
struct AttPlan;
void launch_g1_accumulate(hipStream_t s, const uint32_t* points_s29, const uint32_t* members,
                          const uint32_t* bit_arena, const G1Group* groups, uint32_t n_groups, uint32_t n_slots,
                          uint32_t* lane_partials, uint32_t* wg_partials48, const AttPlan* plan_dev = nullptr,
                          const uint32_t* members1 = nullptr, int exclusive = 0, unsigned long long* clock_rec = nullptr);
// exclusive: the launch asks for this much LDS it never touches (more than half a CU's): at most one of its workgroups per CU
constexpr size_t G1_ACC_EXCLUSIVE_LDS = 82 * 1024;
// The compacting LDS tree over each workgroup's 256 lane partials: one 48-u32 XYZZ partial (192 bytes) per
// (group, workgroup) into wg_partials48.
void launch_g1_tree(hipStream_t s, const uint32_t* lane_partials, const G1Group* groups, uint32_t n_groups,
                    uint32_t n_slots, uint32_t* wg_partials48, int one_per_cu = 0, const AttPlan* plan_dev = nullptr,
                    int solo = 0);
// Per group: add its n_parts partials (stride = part_stride partials apart, starting at first[g] or
// g when first == null), then either write the XYZZ sum (48 u32) or normalise to 96-byte affine.
void launch_g1_finish(hipStream_t s, const uint32_t* partials48, const G1Group* groups, uint32_t n_groups,
                      uint32_t n_parts_fixed, uint32_t part_stride, uint8_t* out_be96, uint32_t* out_xyzz48,
                      const AttPlan* plan_dev = nullptr);

// ---- fork choice ----
struct TreeDev {               // block tree in DFS pre-order (device arrays of n entries)
    const uint32_t* size;      // subtree size of the block at pre-order position i
    const uint32_t* parent;    // pre-order position of the parent (NONE32 for the root of the order)
    const uint32_t* rank;      // rank of the block's root in lexicographic order (tie-break, pe:1114-1116)
    const uint8_t* leaf_ok;    // filter_block_tree's leaf test (Appendix A.3)
    const uint32_t* pos_of_idx;  // insertion index -> pre-order position
    const uint32_t* idx_of_pos;  // pre-order position -> insertion index
    uint32_t n;
};
struct VoteTotals {            // one per K_votes workgroup (VOTES_MAX_WG slots), summed by K_tree
    unsigned long long total_active_balance;
    unsigned long long num_active;
};
// direct[pos] += effective_balance of every counted validator voting for the block at pos.
void launch_votes(hipStream_t s, const uint32_t* vote_block, const uint64_t* eff_balance, const uint8_t* flags,
                  uint64_t n_val, uint32_t filter_slashed, const uint32_t* pos_of_idx,
                  uint32_t n_blocks, uint64_t* direct, VoteTotals* totals, int zero_first,
                  const uint32_t* vote_slot = nullptr, uint32_t min_vote_slot = 0, int lean = 0);
// Subtree sums (prefix scan over pre-order), viability, best child, pointer-jumping descent.
void launch_tree(hipStream_t s, const TreeDev& tree, uint64_t* direct, const VoteTotals* totals,
                 uint64_t totals_override_balance, uint64_t totals_override_num, int use_override,
                 uint32_t justified_pos, uint32_t boost_pos, uint64_t slots_per_epoch, uint64_t boost_percent,
                 uint64_t balance_increment, uint64_t* weights_by_idx, uint32_t* head_idx, int clear_direct,
                 int lean = 0);

// One resolved attestation (device row).
struct AttRow {
    uint32_t member_base;  // offset of the committee in the device members array
    uint32_t n_bits;
    uint32_t bits_word;    // u32-word offset of its (re-packed, zero-padded) bits in the device arena
    uint32_t block_idx;    // LMD vote: insertion index of beacon_block_root
    uint32_t epoch_p1;     // target.epoch + 1
    uint32_t order;        // position in the batch (first-seen wins among equal epochs, pe:1383/1440)
    uint32_t flag_mask;    // participation flags this attestation earns (process_attestation)
    uint32_t which;        // 0 current / 1 previous epoch participation
    uint32_t slot;         // attestation.data.slot (vote-expiry variant only: recorded with the latest message)
    uint32_t gate;         // index into the gate array (NONE32 = ungated): a non-zero gate word voids the row on the
                           // device (rows handed over resident by pe_aggregate whose members overlapped, A.8)
};
// update_latest_messages for a batch: phase 1 atomicMax of (epoch+1, ~order), phase 2 winners write.
void launch_lmd_update(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                       const uint32_t* bit_arena, const uint8_t* flags, uint64_t* vote_key, uint32_t* vote_block,
                       uint32_t* vote_slot = nullptr, const uint32_t* gates = nullptr);
// inverse committee map (partition tables only) and the validator-major form of update_latest_messages
void launch_invert_committees(hipStream_t s, const uint32_t* members, const uint32_t* offsets, uint32_t n_committees,
                              uint32_t* inv_comm, uint32_t* inv_pos, uint64_t n_val);
void launch_lmd_validator_major(hipStream_t s, const AttRow* rows, const uint32_t* crow_start,
                                const uint32_t* crow_list, const uint32_t* inv_comm, const uint32_t* inv_pos,
                                const uint32_t* bit_arena, const uint8_t* flags, uint64_t n_val, uint64_t* vote_key,
                                uint32_t* vote_block, uint32_t* vote_slot = nullptr, const uint32_t* gates = nullptr);
// process_attestation flag loop for one round of pairwise-disjoint attestations.
void launch_participation(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                          const uint32_t* bit_arena, const uint16_t* eff_increments,
                          uint64_t base_reward_per_increment, uint32_t* part_cur_words,
                          uint32_t* part_prev_words, uint64_t* numerators, const uint32_t* numerator_slot,
                          const uint32_t* gates = nullptr);
// aggregation_bits = OR over the group's member attestations; count = popcount (wave reduce).
struct UnionGroup {
    uint32_t list_start;   // into att_bytes[]: BYTE offsets of the member attestations' bits in the raw arena
    uint32_t n_atts;
    uint32_t n_bits;       // len(aggregation_bits): bits past it in the last word are masked off
    uint32_t out_word;     // word offset of the output bitfield
};
// bit_arena: the caller's packed bit arena as uploaded (any byte alignment per member; readable 8 bytes past the last
// member).  out_info[2g] = popcount of the union, out_info[2g + 1] = sum of the members' popcounts minus that: non-zero
// <=> members of the group overlap (their signatures would be counted twice, validator guide A.8).
// host_arena / host_info (nullable): the same two outputs written a second time straight into host-coherent pinned
// memory, so that no device-to-host copy command has to follow the kernel.
void launch_bits_union(hipStream_t s, const UnionGroup* groups, uint32_t n_groups, const uint32_t* att_bytes,
                       const uint8_t* bit_arena, uint32_t* out_arena, uint32_t* out_info,
                       uint32_t* host_arena = nullptr, uint32_t* host_info = nullptr, const AttPlan* plan_dev = nullptr);

// ---- attestation rows resident in device memory (att_kernels.hip; host side: engine_resident.cpp) ----------------
// pe_aggregate / pe_on_attestation_batch / pe_process_attestation_batch with the rows handed over in device (or
// pinned) memory: grouping by AttestationData (what aggregate_impl does with memcmp on the host), committee
// resolution, validate_on_attestation (A.4) and the asserts of pe:724-730 run on the device; the host enqueues a
// fixed sequence of launches sized by upper bounds and reads nothing of the rows.
constexpr uint32_t ATT_EMPTY = 0xFFFFFFFFu;
struct TableDev {                 // one candidate committee table (the store's current / previous epoch)
    uint64_t epoch;
    const uint32_t* members;
    const uint32_t* offsets;      // n_committees + 1
    const uint32_t* inv_comm;     // validator -> committee id (partition tables)
    const uint32_t* inv_pos;      // validator -> index in its committee
    uint32_t n_committees;
    uint32_t valid;
};
struct TablesDev {
    TableDev t[2];
    uint64_t slots_per_epoch;
};
struct AttPlan {                  // one per resident-rows aggregate: written by k_att_plan's last workgroup (device copy + pinned mirror)
    uint32_t n_groups;
    uint32_t n_slots;             // G1 plan: uniform blocks of 1 << log2_block lane slots, group g at slot g << log2_block
    uint32_t k, log2_block;
    uint32_t out_words, out_bytes;
    uint32_t error;               // 0 or -pe_status of the aggregate (reported when the call completes)
    uint32_t packed_same;         // every union but the last is a whole number of words: word layout == byte layout
    uint32_t n_rows_table[2];     // groups resolved against each candidate table
    uint32_t n_rows_in;
    uint32_t last_error;          // sticky copy of `error` (k_att_members clears `error` for the next call's ingest; the exchange
                                  // kernel of a committee-sharded step runs after it and still has to tell the other ranks)
    unsigned long long total_members;
};
struct AttGroup {                 // one per group, in order of first appearance
    uint32_t rep;                 // first input row of the group
    uint32_t n_atts, list_start, cursor;
    uint32_t n_bits, out_word, out_byte;
    uint32_t table;               // 0 / 1 = candidate table, NONE32 = none
    uint32_t pos, size, member_base;
    uint32_t sig_valid;           // AND of the members' PE_ATT_FLAG_SIGNATURE_VALID
    uint32_t status_agg;          // 0, or the pe_att_status that keeps the group from having a committee
    uint32_t index_over;          // data.index >= committees per slot (pe:727) although the flat committee id exists:
                                  // get_beacon_committee (A.6) does not assert it, process_attestation does
    uint32_t pad[2];
};
struct BlockTableDev {            // the store's blocks for device-side validation (built by refresh_tree)
    const uint32_t* root_tab;     // open addressing: slot -> insertion index (NONE32 = empty), keyed by the root's first 8 bytes
    uint32_t root_mask;
    const uint8_t* roots;         // 32 B per block, by insertion index
    const unsigned long long* slot_pos;  // block slot by pre-order position
    const uint32_t* parent_pos;   // parent's pre-order position
    const uint32_t* pos_of_idx;
    uint32_t n_blocks;
};
struct FcCtx {                    // store scalars validate_on_attestation reads (A.4)
    unsigned long long cur_slot, cur_epoch, prev_epoch, slots_per_epoch;
};
struct StateCtxDev {              // the slice of BeaconState process_attestation reads (pe:722-754), resolved by the host
    unsigned long long slot, cur_epoch, prev_epoch, slots_per_epoch, min_inclusion_delay, sqrt_spe;
    unsigned long long cj_epoch, pj_epoch;
    uint8_t cj_root[32], pj_root[32];
    uint32_t tgt_blk[2];          // get_block_root(state, epoch): [0] current, [1] previous epoch (insertion index)
    uint32_t head_blk[64];        // get_block_root_at_slot(state, slot - spe + j)
    unsigned long long base_reward_per_increment;
};
// tab / cnt_tab: the grouping table (slot -> first row of the class, ATT_EMPTY when free) and the class sizes; both are
// left clean by k_att_members.  arena_pad32: 32 bytes behind the copied bit arena, zeroed here.
void launch_att_ingest(hipStream_t s, const void* rows, uint32_t n, uint32_t* tab, uint32_t* cnt_tab, uint32_t tab_mask,
                       uint32_t* slot_of, uint64_t arena_len, AttPlan* plan, void* arena_pad32,
                       const uint32_t* n_dev = nullptr,  // n_dev: the row count lives on the device, n bounds it
                       const void* arena_src = nullptr, void* arena_dst = nullptr);  // arena_src (device memory, 16-byte
                                                          // aligned): the launch copies arena_len bytes to arena_dst itself
// k_att_plan runs one lane per input row over as many 256-lane workgroups as the batch needs.  What its workgroups tell each
// other travels through two small records in device memory, both all-zero between launches (the last workgroup to finish
// clears them):
struct PlanSync {                 // sums / maxima over all groups (device-scope atomics) + the arrival ticket
    uint32_t ticket;              // workgroups that have finished their part; the one that draws the last ticket writes the plan
    uint32_t max_size;            // largest committee among the resolved groups
    uint32_t rows_t[2];           // groups resolved against each candidate table
    uint32_t mis_key;             // max of ~g over groups whose union is not a whole number of words (0 = none)
    uint32_t err;                 // max of the groups' error words
    uint32_t pad[2];
    unsigned long long total_members, pad2[3];
};
struct PlanRec {                  // one per workgroup: its own sums (agg) and the sums up to and including it (incl), each value
    unsigned long long agg[3], incl[3], pad[2];  // an 8-byte word with bit 0 = "written" (decoupled look-back, att_kernels.hip)
};
constexpr uint32_t PLAN_WG = 256;
constexpr uint32_t PLAN_MAX_ROWS = 1u << 24;  // rows per resident aggregate (the look-back words hold 24-bit row counts)
struct AttPlanArgs {
    const void* rows; uint32_t n; const uint32_t* n_dev;
    const uint32_t* tab; const uint32_t* cnt_tab; const uint32_t* slot_of;
    uint32_t* rep_of; uint32_t* gid_of_row;
    AttGroup* grp; UnionGroup* ug;
    uint32_t* crow_start[2]; uint32_t* crow_cursor[2]; uint32_t* crow_cnt[2];
    PlanSync* sync; PlanRec* rec;
    AttPlan* plan; AttPlan* plan_host;
    uint64_t out_arena_cap; uint32_t target_slots, slot_cap, min_k, want_pk;
    TablesDev tables;
};
void launch_att_plan(hipStream_t s, const AttPlanArgs& a);
// committee-sharded exchange (pe_aggregate_exchange): the groups of the resident aggregate packed into fixed slots
// ([4 words header | slots x (36 words row, count, reserved, wps words of OR-ed bits)]), and `world` such buffers unpacked
// into one dense batch of rows (36 words each) + bits (wps words per row); *n_dev = rows unpacked
void launch_att_pack(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, const uint32_t* res_bits,
                     const uint32_t* res_info, uint32_t slots, uint32_t wps, uint32_t* send);
void launch_att_unpack(hipStream_t s, const uint32_t* recv, uint32_t world, uint32_t slots, uint32_t wps, void* out_rows,
                       uint32_t* out_bits, uint32_t* n_dev, uint32_t* err_host);
// n_bound: upper bound of the groups (sizes the grid); cap: entries of the caller's status / count arrays
void launch_att_validate_fc(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, uint32_t n_bound,
                            uint32_t cap, BlockTableDev bt, FcCtx fc, const uint32_t* union_info, AttRow* out_rows,
                            int32_t* status_dev, int32_t* status_host, uint32_t* count_host, uint32_t* err_host);
void launch_att_validate_state(hipStream_t s, const void* rows, const AttGroup* grp, const AttPlan* plan, uint32_t n_bound,
                               uint32_t cap, BlockTableDev bt, const StateCtxDev& st, const uint32_t* union_info,
                               AttRow* out_rows, int32_t* status_dev, int32_t* status_host, uint32_t* err_host);
// validator-major LMD update over both candidate tables in one launch (each lane walks its validator's committee of
// the current-epoch table, then of the previous-epoch one); the per-committee row
// lists are unordered (built with atomics), the batch-order rule is applied by comparing AttRow::order.
void launch_lmd_vm_tables(hipStream_t s, const AttRow* rows, TablesDev tables, uint32_t* const crow_start[2],
                          uint32_t* const crow_list[2], const AttPlan* plan, const uint32_t* bit_arena,
                          const uint8_t* flags, uint64_t n_val, uint64_t* vote_key, uint32_t* vote_block,
                          uint32_t* vote_slot, const uint32_t* gates);
// process_attestation's flag loop, one wave per committee: the committee's rows run in batch order inside the wave
// (committees of a partition table touch disjoint validators, so waves are independent).
void launch_participation_tables(hipStream_t s, const AttRow* rows, TablesDev tables, uint32_t* const crow_start[2],
                                 uint32_t* const crow_list[2], const AttPlan* plan, const uint32_t* bit_arena,
                                 const uint16_t* eff_increments, uint64_t base_reward_per_increment,
                                 uint32_t* part_cur_words, uint32_t* part_prev_words, uint64_t* numerators,
                                 const uint32_t* gates, uint32_t cap);  // cap: slots of numerators[]; more groups = no-op

// ---- paired launches (pair_kernels.hip) --------------------------------------------------------------------------
// A streaming caller's engine stream carries two independent chains: the fork-choice chain of step N (validate ->
// LMD -> votes -> tree) and the row chain of step N + 1 (ingest -> plan -> members -> union).  Side by side on two
// streams the command processor's queue interleaving costs more than the overlap gives; launched as ONE
// kernel per pair -- block ranges of one grid, each range running the body of its own kernel -- they overlap without a
// second queue.  The argument blocks below are the stand-alone kernels' parameters, so that a held-back launch can be
// issued either way (engine_pair.cpp).
struct IngestArgs {
    const void* rows; uint32_t n; uint32_t* tab; uint32_t* cnt_tab; uint32_t tab_mask; uint32_t* slot_of;
    uint64_t arena_len; AttPlan* plan; void* arena_pad32; const uint32_t* n_dev; const void* arena_src; void* arena_dst;
};
struct ValidateFcArgs {
    const void* rows; const AttGroup* grp; const AttPlan* plan; uint32_t n_bound, cap; BlockTableDev bt; FcCtx fc;
    const uint32_t* union_info; AttRow* out_rows; int32_t* status_dev; int32_t* status_host; uint32_t* count_host;
    uint32_t* err_host;
};
struct LmdVmArgs {
    const AttRow* rows; TablesDev tables; const uint32_t* crow_start[2]; const uint32_t* crow_list[2];
    const AttPlan* plan; const uint32_t* bit_arena; const uint8_t* flags; uint64_t n_val; uint64_t* vote_key;
    uint32_t* vote_block; uint32_t* vote_slot; const uint32_t* gates;
};
struct MembersArgs {
    const void* rows; uint32_t n; uint32_t* tab; uint32_t* cnt_tab; const uint32_t* slot_of; const uint32_t* rep_of;
    const uint32_t* gid_of_row; AttGroup* grp; AttPlan* plan; uint32_t* ubytes; uint32_t* member_row;
    uint32_t* host_group_of; void* host_out_rows; const uint32_t* n_dev;
    G1Group* g1; uint32_t* crow_cursor[2]; uint32_t* crow_list[2];  // written per group by the lane of its first row
};
struct VotesArgs {
    const uint32_t* vote_block; const uint64_t* eff_balance; const uint8_t* flags; uint64_t n_val;
    uint32_t filter_slashed; const uint32_t* pos_of_idx; uint32_t n_blocks; uint64_t* direct; VoteTotals* totals;
    const uint32_t* vote_slot; uint32_t min_vote_slot;
};
struct UnionArgs {
    const UnionGroup* groups; uint32_t n_groups; const uint32_t* att_bytes; const uint8_t* bit_arena;
    uint32_t* out_arena; uint32_t* out_info; uint32_t* host_arena; uint32_t* host_info; const AttPlan* plan_dev;
};
struct TreeArgs {
    TreeDev tree; uint64_t* direct; const VoteTotals* totals; uint64_t ov_balance, ov_num; int use_override;
    uint32_t justified_pos, boost_pos; uint64_t slots_per_epoch, boost_percent, balance_increment;
    uint64_t* weights_by_idx; uint32_t* head_idx; int clear_direct;
};
// the stand-alone launches over the same argument blocks (lean: the shapes that fit beside a running accumulation)
void launch_att_ingest(hipStream_t s, const IngestArgs& a);
void launch_att_validate_fc(hipStream_t s, const ValidateFcArgs& a);
void launch_lmd_vm_tables(hipStream_t s, const LmdVmArgs& a);
void launch_att_members(hipStream_t s, const MembersArgs& a);
void launch_votes(hipStream_t s, const VotesArgs& a, int lean);
void launch_bits_union(hipStream_t s, const UnionArgs& a);
void launch_tree(hipStream_t s, const TreeArgs& a, int lean);
// ... and pairwise.  Each returns false when the pair has no common shape (the caller then launches the two alone).
bool launch_pair_ingest_validate(hipStream_t s, const IngestArgs& rows_next, const ValidateFcArgs& fc_prev);
bool launch_pair_plan_lmd(hipStream_t s, const AttPlanArgs& rows_next, const LmdVmArgs& fc_prev);
bool launch_pair_members_votes(hipStream_t s, const MembersArgs& rows_next, const VotesArgs& fc_prev);
bool launch_pair_union_tree(hipStream_t s, const UnionArgs& rows_next, const TreeArgs& fc_prev);
void pair_kernels_preload();  // resolve the pair kernels' code object now rather than inside the first streaming step

// The working-state view mirrors the registry (pe_store_init): sflags = active/slashed (+ active-in-previous-epoch),
// increments = balance / effective_balance_increment.
void launch_state_view_from_registry(hipStream_t s, const uint8_t* flags, const uint64_t* balance, uint64_t increment,
                                     uint64_t n_val, uint8_t* sflags, uint16_t* increments);

// BLSPubkey decompression: 48-byte compressed -> Montgomery rows (nullable) and/or 96-byte uncompressed (nullable)
void launch_g1_decompress(hipStream_t s, const uint8_t* in48, uint64_t n, uint32_t* out_mont24, uint8_t* out_be96,
                          int32_t* status);

// KeyValidate's subgroup part over Montgomery rows (128-byte rows): status 0 ok, 3 r*P != infinity, 4 identity
void launch_g1_key_validate(hipStream_t s, const uint32_t* points_mont, uint64_t n, int32_t* status);

// G2 (g2_kernels.hip): same group descriptors, points as 48 Montgomery words [x0 x1 y0 y1], partials 96 words
void launch_g2_convert(hipStream_t s, const uint8_t* be192, uint32_t* mont48, uint64_t n);
void launch_g2_accumulate(hipStream_t s, const uint32_t* points_mont48, const uint32_t* members,
                          const G1Group* groups, uint32_t n_groups, uint32_t n_slots, uint32_t* wg_partials96);
void launch_g2_decompress(hipStream_t s, const uint8_t* in96, uint64_t n, uint32_t* out_mont48, uint8_t* out_be192,
                          int32_t* status);
// up to G2_BATCH_MAX arrays of compressed signatures decoded by one launch (first_block is filled by the launcher)
constexpr uint32_t G2_BATCH_MAX = 8;
struct G2DecompressBatch {
    const uint8_t* in96[G2_BATCH_MAX]; uint32_t* out_mont48[G2_BATCH_MAX]; int32_t* status[G2_BATCH_MAX];
    uint32_t n[G2_BATCH_MAX]; uint32_t first_block[G2_BATCH_MAX + 1]; uint32_t count;
};
void launch_g2_decompress_batch(hipStream_t s, G2DecompressBatch& b);
void launch_g2_finish(hipStream_t s, const uint32_t* partials96, const G1Group* groups, uint32_t n_groups,
                      uint8_t* out_be192);
// r * P == infinity per decoded point: status 0 -> 3 where it fails (non-zero entries are left alone)
void launch_g2_subgroup_check(hipStream_t s, const uint32_t* points_mont48, uint64_t n, int32_t* status);
// rows with a non-zero status become the (0, 0) row (infinity): a plain sum then leaves them out
void launch_g2_mask_bad(hipStream_t s, uint32_t* points_mont48, uint64_t n, const int32_t* status);
// the signature leg of pe_aggregate: per group the sum of its members' signature points (rows member_row[list_start ..
// + n_atts) of `ug`), compressed to the 96-byte BLSSignature wire form; out_bad[g] = members that did not decode
// (for the legs of up to G2_BATCH_MAX steps in one launch; n_groups bounds plan_dev's count; first_block is filled by the launcher)
struct UnionGroup;
struct AttPlan;
struct G2AggregateRowsBatch {
    const uint32_t* pts[G2_BATCH_MAX]; const int32_t* status[G2_BATCH_MAX]; const UnionGroup* ug[G2_BATCH_MAX];
    const uint32_t* member_row[G2_BATCH_MAX]; const AttPlan* plan_dev[G2_BATCH_MAX]; uint8_t* out96[G2_BATCH_MAX];
    uint32_t* out_bad[G2_BATCH_MAX]; uint32_t n_groups[G2_BATCH_MAX]; uint32_t first_block[G2_BATCH_MAX + 1]; uint32_t count;
};
void launch_g2_aggregate_rows(hipStream_t s, G2AggregateRowsBatch& b);
// the per-signature statuses of up to G2_BATCH_MAX legs into their pinned output blocks, one launch (a copy command per leg cost
// the stream ~15 us each)
struct G2StatusOutBatch { const int32_t* src[G2_BATCH_MAX]; int32_t* dst_host[G2_BATCH_MAX]; uint32_t n[G2_BATCH_MAX]; uint32_t count; };
void launch_g2_status_out(hipStream_t s, const G2StatusOutBatch& b);
// pe_aggregate_signatures: a device-resident index list copied with entries >= n replaced by 0 (*err |= 1 then); members per
// group whose status is non-zero (groups: member_start / n_members)
void launch_g2_index_check(hipStream_t s, const uint32_t* index, uint32_t total, uint32_t n, uint32_t* out_index, uint32_t* err);
void launch_g2_count_bad(hipStream_t s, const int32_t* status, const uint32_t* index, const G1Group* groups, uint32_t n_groups,
                         uint32_t* out_bad);

// get_indexed_attestation: sorted attesting indices per row, written at out_offsets[row] (committees <= 8192 members)
void launch_indexed_attestations(hipStream_t s, const AttRow* rows, uint32_t n_rows, const uint32_t* members,
                                 const uint32_t* bit_arena, const uint32_t* out_offsets, uint32_t* out_indices);

// FFG balance sums (pe:791-802): per-workgroup partials [blocks][3] = {total active, previous target, current target};
// returns the number of workgroups launched.
uint32_t launch_ffg_balances(hipStream_t s, const uint64_t* balance, const uint8_t* sflags, const uint8_t* part_cur,
                             const uint8_t* part_prev, uint64_t n_val, uint64_t* partials);

// compute_committee / compute_shuffled_index (pe:495-534) for a whole list: members[i] = indices[shuffled(i)].
// d_source: rounds * ceil(n/256) * 8 words scratch; d_pivots: rounds words; d_indices null = identity.
int launch_shuffle(hipStream_t s, const uint32_t* d_seed_be, uint32_t n, uint32_t rounds, uint32_t* d_source,
                   uint32_t* d_pivots, const uint32_t* d_indices, uint32_t* d_members);

}  // namespace posevo
