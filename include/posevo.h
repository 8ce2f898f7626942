/*
 * posevo.h -- C ABI of the MI355X attestation-aggregation + LMD-GHOST engine.
 *
 * The reference (ethereum/pos-evolution, one file: pos-evolution.md, cited as
 * pe:N) has no FFI; its "API" for this path is three pyspec signatures over
 * Python objects:
 *     get_head(store) -> Root                                   pe:1102
 *     on_attestation(store, attestation, is_from_block=False)   pe:963 / pe:1423
 *     process_attestation(state, attestation)                   pe:722
 * plus the handlers that feed them (get_forkchoice_store pe:1077, on_tick
 * pe:934, on_block pe:986, on_attester_slashing pe:1447).  Every entry point
 * below names the reference lines it replaces.  INTEGRATION.md shows the
 * ctypes / cgo stubs a client maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++ / torch types.  The caller owns every
 *    buffer it passes; the engine owns its handle and all device memory.
 *  - every function returns PE_OK (0) or a negative pe_status; nothing throws.
 *    A failing call leaves the store unmodified ("Invalid calls to handlers
 *    must not modify store", pe:1041).
 *  - one thread per handle at a time (the spec is sequential, pe:929-1039).
 *    Calls are synchronous: they return once their results are complete in the
 *    caller's buffers (pe_get_head polls the head word its kernel releases to host
 *    memory instead of waiting for the stream to drain; later calls are ordered
 *    behind it on the engine's stream).  The exception is opt-in: between pe_pipeline_begin
 *    and pe_pipeline_end the batch calls return once their work is enqueued (see there).
 *  - roots are 32 opaque bytes; the all-zero root is "unset" (Root(), pe:943).
 *  - G1 points cross the boundary in the 96-byte uncompressed form: big-endian
 *    x (48 B) || big-endian y (48 B); bit 6 of byte 0 set = point at infinity.
 *    Outputs are canonical (x, y fully reduced mod p), so equality with the
 *    oracle is integer equality (tolerance 0).
 *  - bitfields are packed LSB-first (bit i = byte i/8, bit i%8), the SSZ
 *    Bitlist order without the length delimiter; n_bits travels beside it.
 */
#ifndef POSEVO_H
#define POSEVO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_ABI_VERSION 6

typedef struct pe_engine pe_engine;

typedef enum pe_status {
    PE_OK = 0,
    PE_ERR_INVALID_ARG = -1,
    PE_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime failure (message via pe_last_error) */
    PE_ERR_OOM = -3,
    PE_ERR_UNKNOWN_PARENT = -4, /* on_block: assert block.parent_root in store.block_states (pe:990) */
    PE_ERR_FUTURE_BLOCK = -5,   /* on_block: assert get_current_slot(store) >= block.slot (pe:994) */
    PE_ERR_NOT_AFTER_FINALIZED = -6, /* on_block: assert block.slot > finalized_slot (pe:998) */
    PE_ERR_NOT_FINALIZED_DESCENDANT = -7, /* on_block: ancestor-at-finalized-slot check (pe:1000) */
    PE_ERR_DUPLICATE_BLOCK = -8,
    PE_ERR_UNKNOWN_ROOT = -9,
    PE_ERR_CAPACITY = -10,
    PE_ERR_NO_COMMITTEES = -11, /* no committee table loaded for the attestation's target epoch */
    PE_ERR_NOT_SLASHABLE = -12, /* on_attester_slashing: is_slashable_attestation_data false (pe:1453) */
    PE_ERR_INVALID_INDEXED = -13, /* is_valid_indexed_attestation false (pe:1455-1456) */
    PE_ERR_STATE = -14,         /* call sequence error (e.g. store not initialised) */
    PE_ERR_TIMEOUT = -15        /* multi-GPU: a collective did not complete within the engine's bounded wait */
} pe_status;

/* Per-attestation result codes written by the *_batch calls (0 = applied).  Each
 * value names the assert that failed; a rejected attestation changes nothing. */
typedef enum pe_att_status {
    PE_ATT_OK = 0,
    PE_ATT_TARGET_EPOCH_NOT_CURRENT_OR_PREVIOUS = 1, /* validate_target_epoch_against_current_time / pe:724 */
    PE_ATT_TARGET_EPOCH_SLOT_MISMATCH = 2,           /* target.epoch == compute_epoch_at_slot(data.slot), pe:725 */
    PE_ATT_UNKNOWN_TARGET_ROOT = 3,
    PE_ATT_UNKNOWN_BEACON_BLOCK_ROOT = 4,
    PE_ATT_BLOCK_AFTER_ATTESTATION_SLOT = 5,
    PE_ATT_TARGET_NOT_ANCESTOR = 6,                  /* LMD vote must be consistent with FFG vote target */
    PE_ATT_SLOT_NOT_IN_PAST = 7,                     /* get_current_slot(store) >= data.slot + 1 */
    PE_ATT_NO_COMMITTEE_TABLE = 8,
    PE_ATT_COMMITTEE_INDEX_OUT_OF_RANGE = 9,         /* pe:727 */
    PE_ATT_BITS_LENGTH_MISMATCH = 10,                /* len(aggregation_bits) == len(committee), pe:730 */
    PE_ATT_EMPTY_OR_INVALID_INDICES = 11,            /* is_valid_indexed_attestation structural part, pe:736/976 */
    PE_ATT_BAD_SIGNATURE = 12,                       /* injected pairing result false, pe:736/976 */
    PE_ATT_INCLUSION_WINDOW = 13,                    /* pe:726 */
    PE_ATT_SOURCE_MISMATCH = 14                      /* assert is_matching_source (Appendix A.9) */
} pe_att_status;

/* Constants the reference cites by name only (pe:467, 1021-1022, 1054, ...);
 * defaults = mainnet preset. */
typedef struct pe_config {
    uint64_t slots_per_epoch;                 /* 32, pe:472 */
    uint64_t seconds_per_slot;                /* 12, pe:1536 */
    uint64_t intervals_per_slot;              /* 3,  pe:1536 */
    uint64_t safe_slots_to_update_justified;  /* 8,  pe:1054 */
    uint64_t proposer_score_boost;            /* 40 (percent of one slot's committee weight), pe:1355 */
    uint64_t effective_balance_increment;     /* 1e9 Gwei, pe:126 */
    uint64_t min_attestation_inclusion_delay; /* 1, pe:726 */
    uint64_t max_validators_per_committee;    /* 2048, pe:715 */
    uint32_t filter_slashed;                  /* 0: this era's get_latest_attesting_balance ignores `slashed` */
    int32_t  device;                          /* HIP device ordinal; -1 = current device */
    uint64_t reserve_validators;              /* capacity hints (0 = grow on demand) */
    uint32_t reserve_blocks;
    uint32_t max_committee_tables;            /* epochs of committee tables kept resident (0 = 4) */
    uint64_t vote_expiry_slots;               /* 0 = LMD-GHOST (no expiry).  eta > 0: RLMD-GHOST's vote expiry period
                                                 (pe:1585-1596; eta = 1 is Goldfish's GHOST-Eph, pe:1549): get_head
                                                 counts a latest message only if message.slot + eta >= current slot */
} pe_config;

/* Validator flag bits (T1: the per-validator byte the vote kernel streams). */
#define PE_VAL_ACTIVE       0x01u  /* activation_epoch <= current_epoch(justified state) < exit_epoch */
#define PE_VAL_SLASHED      0x02u  /* Validator.slashed (pe:40) */
#define PE_VAL_EQUIVOCATING 0x04u  /* in store.equivocating_indices (pe:897); set by the engine */

/* Attestation row: AttestationData (pe:689-697) + where its Bitlist (pe:715) lives
 * in the caller's bit arena + the injected signature verdict (pe:717). */
#define PE_ATT_FLAG_SIGNATURE_VALID 0x1u  /* result of the out-of-scope pairing check */
#define PE_ATT_FLAG_FROM_BLOCK      0x2u  /* is_from_block (pe:1423) */
#define PE_ATT_FLAG_OVERLAPPING_BITS 0x4u /* set by pe_aggregate on an output row whose members share a bit: the summed
                                             signature counts that validator twice while bits and aggregate pubkey count
                                             it once, so the aggregate can never verify (validator guide, Appendix A.8:
                                             "aggregates with overlapping bits are not merged").  PE_ATT_FLAG_SIGNATURE_VALID
                                             is cleared on such a row; re-aggregate without the duplicate members. */
typedef struct pe_attestation {
    uint64_t slot;                  /* data.slot */
    uint64_t index;                 /* data.index: committee index within the slot */
    uint8_t  beacon_block_root[32]; /* data.beacon_block_root: the LMD GHOST vote */
    uint64_t source_epoch;          /* data.source */
    uint8_t  source_root[32];
    uint64_t target_epoch;          /* data.target */
    uint8_t  target_root[32];
    uint32_t bits_offset;           /* byte offset of aggregation_bits in the arena */
    uint32_t n_bits;                /* len(aggregation_bits) */
    uint32_t flags;                 /* PE_ATT_FLAG_* */
    uint32_t reserved0;
} pe_attestation;                   /* 144 bytes */

/* The slice of BeaconState that process_attestation (pe:722-754) reads. */
typedef struct pe_state_ctx {
    uint64_t slot;                        /* state.slot */
    uint8_t  chain_tip_root[32];          /* latest block of the state's chain: get_block_root* resolve against it */
    uint64_t current_justified_epoch;     /* state.current_justified_checkpoint */
    uint8_t  current_justified_root[32];
    uint64_t previous_justified_epoch;    /* state.previous_justified_checkpoint */
    uint8_t  previous_justified_root[32];
    uint64_t base_reward_per_increment;   /* get_base_reward_per_increment(state) (Appendix A.9) */
} pe_state_ctx;

/* ---- lifecycle --------------------------------------------------------- */
uint32_t    pe_abi_version(void);
void        pe_config_default(pe_config* cfg);
int         pe_engine_create(const pe_config* cfg, pe_engine** out);
void        pe_engine_destroy(pe_engine* h);
const char* pe_strerror(int status);
const char* pe_last_error(const pe_engine* h);  /* detail of the last failing call on this handle */
/* Run the engine's kernels on a caller-owned HIP stream (hipStream_t as void*).
 * NULL = the engine's own stream.  Lets a host framework order its collectives
 * (RCCL) with the engine's kernels without host synchronisation. */
int         pe_set_stream(pe_engine* h, void* hip_stream);

/* ---- store: get_forkchoice_store (pe:1077-1095) ------------------------- */
/* Anchor block + state: justified = finalized = best_justified = (anchor_epoch,
 * anchor_root); time = genesis_time + SECONDS_PER_SLOT * anchor_slot; no latest
 * messages; no equivocators; no boost.  Resets any previous store contents. */
int pe_store_init(pe_engine* h, uint64_t genesis_time, uint64_t anchor_slot,
                  const uint8_t anchor_root[32]);

/* checkpoint_states[justified_checkpoint].validators (Appendix A.1): balances and
 * activity of the JUSTIFIED-checkpoint state.  pubkeys96 may be NULL (no G1 work).
 * Copies the caller's buffers.  Keeps existing votes when n is unchanged/grows. */
int pe_set_validators(pe_engine* h, uint64_t n, const uint8_t* pubkeys96,
                      const uint64_t* effective_balance, const uint8_t* flags);
/* Cheap refresh when the justified checkpoint changes: balances + flags only. */
int pe_set_balances(pe_engine* h, uint64_t n, const uint64_t* effective_balance, const uint8_t* flags);

/* on_tick (pe:934-955): time, boost reset on a new slot, best_justified promotion. */
int pe_on_tick(pe_engine* h, uint64_t time);

/* on_block (pe:986-1036) minus state_transition: the caller supplies the block's
 * root and the post-state's current_justified / finalized checkpoints. */
int pe_on_block(pe_engine* h, const uint8_t root[32], const uint8_t parent_root[32], uint64_t slot,
                uint64_t post_justified_epoch, const uint8_t post_justified_root[32],
                uint64_t post_finalized_epoch, const uint8_t post_finalized_root[32]);
/* Raw insertion (store.blocks[root] = block, pe:1016) with only the structural
 * checks (known parent, slot > parent.slot); no time/finality/boost handling.
 * For bulk loading a tree (checkpoint sync, benchmarks). */
int pe_add_block(pe_engine* h, const uint8_t root[32], const uint8_t parent_root[32], uint64_t slot,
                 uint64_t post_justified_epoch, const uint8_t post_justified_root[32],
                 uint64_t post_finalized_epoch, const uint8_t post_finalized_root[32]);

/* Direct setters for store scalars (pe:891-896); root may be the zero root. */
int pe_set_checkpoints(pe_engine* h, uint64_t justified_epoch, const uint8_t justified_root[32],
                       uint64_t finalized_epoch, const uint8_t finalized_root[32]);
int pe_set_proposer_boost(pe_engine* h, const uint8_t root[32]);
/* on_attester_slashing's effect (pe:1459-1461): equivocating_indices.add(index). */
int pe_mark_equivocating(pe_engine* h, const uint64_t* indices, uint64_t n);
/* on_attester_slashing (pe:1447-1461) on two IndexedAttestations: checks
 * is_slashable_attestation_data (pe:1134-1143) and the structural part of
 * is_valid_indexed_attestation, then marks the intersection. */
int pe_on_attester_slashing(pe_engine* h,
                            const pe_attestation* data_1, const uint64_t* indices_1, uint64_t n_1,
                            const pe_attestation* data_2, const uint64_t* indices_2, uint64_t n_2);

/* Committee table of one epoch = get_beacon_committee(state, slot, index) for every
 * (slot, index) of that epoch (Appendix A.6 / compute_committee pe:495-504), as CSR:
 * committee id = (slot % SLOTS_PER_EPOCH) * committees_per_slot + index,
 * members[offsets[id] .. offsets[id+1]).  n_committees must be a multiple of
 * SLOTS_PER_EPOCH.  The engine keeps the cfg.max_committee_tables most recently used epochs. */
int pe_set_committees(pe_engine* h, uint64_t epoch, uint32_t n_committees,
                      const uint32_t* offsets, const uint32_t* members);

/* The same table computed ON THE GPU from the epoch's seed: compute_committee (pe:495-504) over
 * compute_shuffled_index (pe:513-534, swap-or-not, shuffle_round_count rounds of SHA-256) for every committee
 * of the epoch: committee c = [active_indices[shuffled(i)] for i in [n*c/count, n*(c+1)/count)], count =
 * n_committees.  seed = get_seed(state, epoch, DOMAIN_BEACON_ATTESTER) (pe:481-486) and the active set are the
 * caller's (state accessors); active_indices NULL = validators 0 .. n_active - 1 (every validator active).  Registers
 * the table for `epoch` like pe_set_committees and KEEPS IT ON THE DEVICE; out_offsets (n_committees + 1) and
 * out_members (n_active) are optional copies of the result (NULL: nothing is read back). */
int pe_compute_committees(pe_engine* h, uint64_t epoch, const uint8_t seed[32], const uint32_t* active_indices,
                          uint32_t n_active, uint32_t n_committees, uint32_t shuffle_round_count,
                          uint32_t* out_offsets, uint32_t* out_members);

/* The same shuffle without a wait and without a read-back, for a caller that streams epochs: the kernels go to a
 * stream of their own (beside the steps' G1 sums and fork-choice kernels; a shuffling is known one epoch ahead --
 * MIN_SEED_LOOKAHEAD, get_seed pe:481-486 -- so it is enqueued an epoch before the calls that read it), the table is
 * registered at once, and the first call that reads it makes the engine's stream wait for the shuffle.  A caller that streams with lag depth L keeps
 * max_committee_tables >= L + 3: the least recently used table is rewritten in place, and if work in flight may still
 * read it the call first completes everything in flight. */
int pe_compute_committees_async(pe_engine* h, uint64_t epoch, const uint8_t seed[32], const uint32_t* active_indices,
                                uint32_t n_active, uint32_t n_committees, uint32_t shuffle_round_count);

/* ---- pipelined calls: one wait per step --------------------------------- */
/* Between pe_pipeline_begin and pe_pipeline_end the batch calls (pe_aggregate, pe_on_attestation_batch,
 * pe_process_attestation_batch) validate on the host, enqueue their device work and RETURN WITHOUT WAITING:
 *   - what is host-derived is complete at return: out_atts rows, out_n_groups, group_of, every status[] entry that
 *     names a failed host-side assert;
 *   - device results (OR-ed bits, counts, aggregate pubkeys, numerators, and the status of rows whose verdict needs the
 *     bits: PE_ATT_EMPTY_OR_INVALID_INDICES / PE_ATT_BAD_SIGNATURE for overlapping members) are written into the
 *     caller's buffers when pe_pipeline_end returns -- the buffers must stay alive until then;
 *   - pe_get_head stays synchronous (it returns a root) and is ordered behind everything enqueued before it; the G1
 *     sums of a pipelined pe_aggregate run on a second stream beside the fork-choice kernels;
 *   - any other entry point first completes the pipeline's outstanding work.
 * Results are the same as with synchronous calls.  pe_pipeline_end returns the first deferred error, if any. */
int pe_pipeline_begin(pe_engine* h);
int pe_pipeline_end(pe_engine* h);
/* Lagged end, for a caller that streams step after step: returns once the pipeline L before this one is complete
 * (its buffers filled, its deferred error returned), L = the lag depth (default 2); this one completes at the L-th next
 * pe_pipeline_end_lagged, at pe_pipeline_end or at any other synchronous call.  The G1 sums of step N then run while
 * the host prepares and enqueues steps N+1 .. N+L-1.  Buffers handed to the calls of a lagged pipeline must stay alive
 * until that later completion. */
int pe_pipeline_end_lagged(pe_engine* h);
/* Lag depth of the lagged pipelines, 1 .. 7 (L + 1 sets of staging / output blocks rotate; they allocate on first use).
 * Depth 2 keeps a step's outputs two steps behind; depth 3 lets the host run far enough ahead that the device queue never
 * drains while a step's latency-bound tail (tree, normalisation) is still in flight -- what bench.py's throughput leg
 * uses.  Completes everything in flight first; not inside a pipeline (PE_ERR_STATE). */
int pe_pipeline_set_lag(pe_engine* h, uint32_t depth);
uint32_t pe_pipeline_get_lag(const pe_engine* h);
/* Which buffers may be touched: pipelines are numbered from 1 in the order of their pe_pipeline_begin(_streaming);
 * pe_pipeline_generation = the number of the one begun last, pe_pipeline_completed = the highest number whose outputs
 * are complete (they complete in order: after a pe_pipeline_end_lagged that is generation - lag or later, after
 * pe_pipeline_end or any synchronous call it is generation).  A caller that rotates its output buffers reads set g
 * when pe_pipeline_completed(h) >= g and hands it out again after that -- no need to count lagged ends itself. */
uint64_t pe_pipeline_generation(const pe_engine* h);
uint64_t pe_pipeline_completed(const pe_engine* h);
/* pe_pipeline_begin for a pipeline that will end lagged: the G1 sums of its pe_aggregate are not launched by that call
 * but by pe_pipeline_end_lagged, BEHIND the step's fork-choice kernels.  k_g1_accumulate fills every CU for its whole
 * run, so launched first it would hold pe_get_head (and with it the host's preparation of the next step) back until it
 * ends; launched last it overlaps exactly that preparation.  Results are unchanged. */
int pe_pipeline_begin_streaming(pe_engine* h);
/* Device-resident hand-over: pass PE_BITS_RESIDENT as bits_arena (arena_len ignored) to pe_on_attestation_batch /
 * pe_process_attestation_batch when `atts` are rows of out_atts of the LAST pe_aggregate on this handle (any subset,
 * any order, fields other than flags unchanged): their OR-ed bits are used where pe_aggregate left them in HBM --
 * nothing is re-packed or re-uploaded, and it works inside a pipeline where out_bits_arena is not filled yet.  Rows
 * whose members overlapped (PE_ATT_FLAG_OVERLAPPING_BITS) are rejected with PE_ATT_BAD_SIGNATURE; len(aggregation_bits)
 * must equal the committee length.  A row that is not a row of the last pe_aggregate -- e.g. one of an earlier
 * aggregate of the same pipeline, which a later one replaced -- fails the call with PE_ERR_INVALID_ARG (rows are
 * recognised by bits_offset, n_bits and a fold of their AttestationData), nothing applied. */
#define PE_BITS_RESIDENT ((const uint8_t*)(uintptr_t)1)
/* Rows resident on the device: the host off the step's critical path.
 * pe_aggregate accepts `atts` in DEVICE memory (hipMalloc of the engine's device, 16-byte aligned; the bits as above).
 * The host then reads nothing of the rows: grouping by AttestationData (pe:689-697), committee resolution
 * (get_beacon_committee's index arithmetic, Appendix A.6), validate_on_attestation (Appendix A.4, called at pe:970) and
 * the asserts of process_attestation (pe:724-730) all run on the device, and EVERY output of the call -- out_atts,
 * *out_n_groups, group_of, bits, counts, aggregate pubkeys -- is complete when a synchronous call returns / when the
 * pipeline's outputs are (nothing is host-derived any more).  Capacities are the caller's bounds: out_atts, group_of,
 * out_count, out_aggpk96 hold n entries (groups <= rows).  The handlers then take
 *     pe_on_attestation_batch(h, PE_ROWS_RESIDENT, cap, PE_BITS_RESIDENT, 0, status, NULL, out_count)
 *     pe_process_attestation_batch(h, state, PE_ROWS_RESIDENT, cap, PE_BITS_RESIDENT, 0, status, out_numerators)
 * = on_attestation / process_attestation over every group of that aggregate, in group order; cap = entries of the
 * status / count / numerator arrays (entries past the groups formed read 0; fewer entries than groups: PE_ERR_CAPACITY
 * at completion, nothing applied).  Statuses and results equal those of the host-row path.  Restrictions of this mode:
 *   - committees are resolved against the tables of the store's CURRENT and PREVIOUS epoch (the only targets
 *     validate_on_attestation admits for gossip attestations; a from-block row with an older target reads
 *     PE_ATT_NO_COMMITTEE_TABLE -- hand such rows over from host memory), and both tables must partition the registry
 *     (a real shuffling does; pe_compute_committees tables always do);
 *   - the store's clock and tables stay unchanged between pe_aggregate and its handlers (else PE_ERR_STATE);
 *   - a failing aggregate (bits outside the arena, no committee table / index out of range / len(aggregation_bits) !=
 *     len(committee) when aggregate pubkeys are asked for, output arena too small) forms NO groups: its error is returned
 *     where its outputs complete, and the handlers behind it apply nothing;
 *   - signature points (sig_points96) and the partial / sharded forms take host rows. */
#define PE_ROWS_RESIDENT ((const pe_attestation*)(uintptr_t)1)

/* ---- the hot path ------------------------------------------------------ */
/* get_head (pe:1102-1116): full recomputation from the V-entry vote table:
 * get_filtered_block_tree (Appendix A.3), get_latest_attesting_balance
 * (Appendix A.1, incl. proposer boost and equivocation mask), heaviest-child
 * descent with ties to the lexicographically higher root (pe:1114-1116). */
int pe_get_head(pe_engine* h, uint8_t out_root[32]);
/* The same computation with the root delivered like every other output of a pipeline: inside pe_pipeline_begin ...
 * _end(_lagged) the call enqueues its kernels and returns; out_root is written where the pipeline's outputs complete
 * (it must stay alive until then).  For a caller that streams steps and consumes results behind -- chain sync, replay,
 * the throughput leg of bench.py -- so that its loop never blocks on the device inside a step.  Outside a pipeline it is
 * pe_get_head. */
int pe_get_head_async(pe_engine* h, uint8_t out_root[32]);
/* get_latest_attesting_balance(store, root) for every block, in insertion order
 * (index 0 = anchor).  out must hold pe_num_blocks(h) entries. */
int pe_get_weights(pe_engine* h, uint64_t* out_weights, uint32_t n);
/* The per-block weights the LAST head computation left behind (pe_get_head, or pe_head_from_weights on a reduced
 * multi-GPU buffer), without recomputing them. */
int pe_get_last_weights(pe_engine* h, uint64_t* out_weights, uint32_t n);

/* on_attestation (pe:963-979, pe:1423-1428) x n, applied AS IF sequentially in
 * array order: validate_on_attestation (Appendix A.4), committee lookup
 * (get_indexed_attestation), is_valid_indexed_attestation (structural +
 * injected signature verdict), update_latest_messages (pe:1435-1441).
 * status[i] receives a pe_att_status.  out_aggpk96 (nullable, 96*n bytes)
 * receives sum of the attesters' pubkeys = the G1 sum FastAggregateVerify
 * consumes (Appendix A.7); out_count (nullable) the number of attesters. */
int pe_on_attestation_batch(pe_engine* h, const pe_attestation* atts, uint32_t n,
                            const uint8_t* bits_arena, uint64_t arena_len,
                            int32_t* status, uint8_t* out_aggpk96, uint32_t* out_count);

/* get_indexed_attestation (Appendix A.6; call sites pe:736, pe:975) x n: attesting_indices =
 * sorted(committee[i] for i with aggregation_bits[i]), resolved against the committee table of each row's target
 * epoch exactly as pe_on_attestation_batch resolves it (no store time/root validation: this is the state-only
 * helper).  out_offsets has n + 1 entries; indices of row i are out_indices[out_offsets[i] .. out_offsets[i+1]).
 * status[i] = PE_ATT_NO_COMMITTEE_TABLE / _COMMITTEE_INDEX_OUT_OF_RANGE / _BITS_LENGTH_MISMATCH or 0.
 * out_indices must have room for the total number of set bits (<= sum of n_bits). */
int pe_get_indexed_attestations(pe_engine* h, const pe_attestation* atts, uint32_t n,
                                const uint8_t* bits_arena, uint64_t arena_len, int32_t* status,
                                uint32_t* out_offsets, uint32_t* out_indices, uint64_t out_indices_cap);

/* Aggregation (validator guide, Appendix A.8; reference prose pe:474/659/715/1536):
 * attestations with identical AttestationData and n_bits form one group (groups
 * ordered by first appearance).  Per group: aggregation_bits = OR of the members'
 * bits; signature = sum of the members' signature points (sig_points96, 96 B per
 * input attestation, nullable); aggregate pubkey = sum of pubkey[committee[i]]
 * over the OR-ed bits (needs the committee table of every target epoch in the batch;
 * NULL out to skip).  A batch may span target epochs (an epoch boundary); only the
 * partial / sharded forms below want one target epoch per call.
 *   out_atts[g]   : the group's data, bits_offset into out_bits_arena
 *   group_of[i]   : group index of input attestation i (nullable)
 * Capacities: out_atts has room for n rows, out_bits_arena for out_arena_cap bytes.
 * bits_arena may lie in pageable host memory (copied during the call), in pinned host memory or in device memory
 * (hipMalloc of the engine's device): the latter two are picked up by the copy engine without a pass on the host, and
 * must stay unchanged until the call's outputs are complete (inside a pipeline: until the pipeline's are).  The same
 * holds for pe_aggregate_partial and pe_aggregate_sharded.  The attestation rows are host memory, or -- pe_aggregate
 * only -- device memory: see PE_ROWS_RESIDENT above. */
int pe_aggregate(pe_engine* h, const pe_attestation* atts, uint32_t n,
                 const uint8_t* bits_arena, uint64_t arena_len, const uint8_t* sig_points96,
                 pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                 uint8_t* out_bits_arena, uint64_t out_arena_cap,
                 uint8_t* out_sig96, uint8_t* out_aggpk96, uint32_t* out_count);

/* process_attestation (pe:722-754) x n, as if sequentially in array order, on the
 * engine's working participation arrays: asserts pe:724-730, participation flag
 * indices (Appendix A.9), flag RMW pe:745-749, and per attestation the
 * proposer_reward_numerator (pe:744-749).  The caller finishes pe:752-754
 * (numerator // denominator, increase_balance). */
int pe_process_attestation_batch(pe_engine* h, const pe_state_ctx* state,
                                 const pe_attestation* atts, uint32_t n,
                                 const uint8_t* bits_arena, uint64_t arena_len,
                                 int32_t* status, uint64_t* out_numerators);
/* state.{current,previous}_epoch_participation of the working state (pe:739-742).
 * which: 0 = current, 1 = previous. */
int pe_participation_set(pe_engine* h, int which, const uint8_t* flags, uint64_t n);
int pe_participation_get(pe_engine* h, int which, uint8_t* out_flags, uint64_t n);
/* process_participation_flag_updates at an epoch boundary: previous = current, current = 0. */
int pe_participation_rotate(pe_engine* h);

/* The working BeaconState's registry view (the state process_attestation / FFG run on), when it differs from the
 * justified-checkpoint state given to pe_set_validators: effective balances (get_base_reward, pe:749; FFG sums)
 * and flags PE_VAL_ACTIVE (active in get_current_epoch(state)), PE_VAL_SLASHED, PE_VAL_ACTIVE_PREV (active in
 * get_previous_epoch(state)).  Until this is called the engine uses the pe_set_validators data for both. */
#define PE_VAL_ACTIVE_PREV 0x08u
int pe_state_set_validators(pe_engine* h, uint64_t n, const uint64_t* effective_balance, const uint8_t* flags);
/* Read-backs for checkpoint / resume (pe_store_init drops committee tables, the working-state view and participation
 * with the store; a restart hands them back).  *out_is_set = 0: the view still mirrors pe_set_validators. */
int pe_state_get_validators(pe_engine* h, uint64_t n, uint64_t* out_effective_balance, uint8_t* out_flags, int* out_is_set);
/* Epochs of the committee tables held (out_epochs NULL: count only), and one table: out_offsets u32[n_committees + 1],
 * out_members u32[offsets[n_committees]] (either may be NULL; *out_n_committees is always written). */
int pe_get_committee_epochs(pe_engine* h, uint64_t* out_epochs, uint32_t cap, uint32_t* out_n);
int pe_get_committees(pe_engine* h, uint64_t epoch, uint32_t* out_n_committees, uint32_t* out_offsets,
                      uint32_t offsets_cap, uint32_t* out_members, uint64_t members_cap);
/* The balance sums of process_justification_and_finalization (pe:791-802) over the working state and the engine's
 * participation arrays: out[0] = get_total_active_balance(state), out[1] = previous_target_balance,
 * out[2] = current_target_balance (each max(EFFECTIVE_BALANCE_INCREMENT, sum), Appendix A.1).  The caller feeds
 * them to weigh_justification_and_finalization (pe:815-853), which is scalar logic on the state. */
int pe_ffg_balances(pe_engine* h, uint64_t out[3]);

/* Plain G1 sum over caller-chosen groups: out[g] = sum_{j in [offsets[g], offsets[g+1])}
 * points[index[j]] (index NULL = identity).  points96 NULL = the validators' pubkeys. */
int pe_g1_sum(pe_engine* h, const uint8_t* points96, uint64_t n_points,
              const uint32_t* index, const uint32_t* offsets, uint32_t n_groups, uint8_t* out96);

/* BLSPubkey wire format (Validator.pubkey: BLSPubkey, pe:37): 48 bytes, big-endian x, flag bits in the leading byte
 * (bit 7 compressed, bit 6 infinity, bit 5 y > (p-1)/2).  Decompression runs on the GPU (one square root per key);
 * status[i]: 0 ok, 1 malformed encoding (flag bits, x >= p), 2 x is not on the curve.  Subgroup: pe_g1_key_validate.
 *   pe_g1_decompress          48-byte keys -> 96-byte uncompressed affine (the format pe_set_validators takes)
 *   pe_set_pubkeys_compressed decompress straight into the registry of the n validators already loaded;
 *                             any status != 0 fails the call (PE_ERR_INVALID_ARG) and leaves no pubkeys loaded
 *   pe_g1_compress            96-byte affine -> 48-byte compressed (serialisation only: host, no handle) */
/* KeyValidate (FastAggregateVerify validates every pubkey before summing them, Appendix A.7; call sites pe:736, pe:976):
 * beyond "decodes to a curve point" (pe_g1_decompress / pe_set_pubkeys_compressed) a key must not be the identity and
 * must lie in the prime-order subgroup, r * P == infinity -- one 255-bit scalar multiplication per key on the GPU.
 * points96: n uncompressed points, or NULL = the n pubkeys of the registry as loaded.
 * status[i]: 0 valid, 3 not in the subgroup, 4 identity.  An optional once-per-registry-load pass (~65 ms per 1 M keys). */
int pe_g1_key_validate(pe_engine* h, const uint8_t* points96, uint64_t n, int32_t* status);
int pe_g1_decompress(pe_engine* h, const uint8_t* in48, uint64_t n, uint8_t* out96, int32_t* status);
int pe_set_pubkeys_compressed(pe_engine* h, uint64_t n, const uint8_t* pubkeys48, int32_t* status);
int pe_g1_compress(const uint8_t* in96, uint64_t n, uint8_t* out48);

/* bls.Aggregate over real BLSSignature points (type pe:37, Attestation.signature pe:717; aggregation prose pe:659,
 * pe:1536): plain G2 sum over caller-chosen groups, out[g] = sum_{j in [offsets[g], offsets[g+1])} points[index[j]]
 * (index NULL = identity).  Points are 192-byte uncompressed affine, ZCash order x.c1 | x.c0 | y.c1 | y.c0, each 48
 * bytes big-endian; bit 6 of byte 0 flags infinity.  Outputs are canonical affine in the same format (exact). */
#define PE_G2_POINT_BYTES 192
/* BLSSignature wire format (pe:37, pe:717): 96 bytes x.c1 | x.c0, flag bits as for BLSPubkey with the sign taken on
 * (y.c1, y.c0).  pe_g2_decompress runs on the GPU (an Fp2 square root per point: two Fp exponentiations, ~0.95 ms
 * per 8192 points); status[i]: 0 ok, 1 malformed, 2 not on the curve; no subgroup check.  pe_g2_compress is host-side
 * serialisation. */
int pe_g2_decompress(pe_engine* h, const uint8_t* in96, uint64_t n, uint8_t* out192, int32_t* status);
int pe_g2_compress(const uint8_t* in192, uint64_t n, uint8_t* out96);
int pe_g2_sum(pe_engine* h, const uint8_t* points192, uint64_t n_points,
              const uint32_t* index, const uint32_t* offsets, uint32_t n_groups, uint8_t* out192);
/* bls.Aggregate per committee over the UNAGGREGATED signatures of an epoch (pe:717: one BLSSignature per attester; pe:474,
 * pe:659, pe:1536: aggregated per committee): signatures96 = n compressed signatures (host or DEVICE memory; device memory
 * at a 16-byte boundary is read in place), aggregate g = sum of signatures96[index[j]] for j in [offsets[g], offsets[g+1])
 * (index: host or device memory, NULL = identity; offsets: host), decompressed on the whole chip (k_g2_decompress: an Fp2
 * square root each, curve membership checked; PE_SIG_CHECK_SUBGROUP adds the endomorphism test), summed (k_g2_accumulate /
 * k_g2_finish) and handed back compressed in out_signatures96 (n_groups x 96 bytes, host).  sig_status (nullable, n entries,
 * host): 0 ok, 1 malformed, 2 not on the curve, 3 outside G2; a signature that does not decode is left out of its sum and
 * counted in out_bad[g] (nullable, n_groups entries).  Synchronous.  1 048 576 signatures in 2048 committees: ~1 k
 * Fp products per signature against ~10 per addition -- the decompression is the cost of the call. */
int pe_aggregate_signatures(pe_engine* h, const uint8_t* signatures96, uint64_t n, const uint32_t* index,
                            const uint32_t* offsets, uint32_t n_groups, uint32_t sig_flags, uint8_t* out_signatures96,
                            int32_t* sig_status, uint32_t* out_bad);

/* ---- inspection (parity checks) ---------------------------------------- */
uint32_t pe_num_blocks(const pe_engine* h);
uint64_t pe_num_validators(const pe_engine* h);
int pe_block_root_at(const pe_engine* h, uint32_t block_index, uint8_t out_root[32]);
int pe_block_index_of(const pe_engine* h, const uint8_t root[32], uint32_t* out_index);
/* store.latest_messages: epoch and block index per validator; block index
 * 0xFFFFFFFF = no message. */
int pe_get_latest_messages(pe_engine* h, uint64_t* out_epoch, uint32_t* out_block_index, uint64_t n);
/* ---- checkpoint / resume: the store is a handful of flat arrays -------------------------------
 * Export: pe_get_store_scalars, pe_get_block x pe_num_blocks (insertion order: parents first), pe_get_validator_flags
 * (incl. PE_VAL_EQUIVOCATING), pe_get_latest_messages (+ _slots under the vote-expiry variant), pe_participation_get.
 * Import into a fresh handle: pe_store_init(genesis_time, block 0) -> pe_add_block in order -> pe_set_validators
 * (balances, flags without the equivocating bit, pubkeys) -> pe_mark_equivocating -> pe_set_latest_messages ->
 * pe_on_tick(time) -> pe_set_checkpoints / pe_set_best_justified / pe_set_proposer_boost -> pe_participation_set.
 * pos_evolution_amd.Engine.export_state / import_state do exactly this. */
int pe_get_block(const pe_engine* h, uint32_t block_index, uint8_t root[32], uint32_t* parent_index, uint64_t* slot,
                 uint64_t* post_justified_epoch, uint8_t post_justified_root[32],
                 uint64_t* post_finalized_epoch, uint8_t post_finalized_root[32]);
int pe_get_validator_flags(const pe_engine* h, uint8_t* out_flags, uint64_t n);
int pe_get_latest_message_slots(pe_engine* h, uint32_t* out_slot, uint64_t n);
/* block_index 0xFFFFFFFF = no message; slot may be NULL (only read under the vote-expiry variant). */
int pe_set_latest_messages(pe_engine* h, uint64_t n, const uint64_t* epoch, const uint32_t* block_index,
                           const uint32_t* slot);
int pe_set_best_justified(pe_engine* h, uint64_t epoch, const uint8_t root[32]);
int pe_get_store_scalars(const pe_engine* h, uint64_t* time, uint64_t* genesis_time,
                         uint64_t* justified_epoch, uint8_t justified_root[32],
                         uint64_t* finalized_epoch, uint8_t finalized_root[32],
                         uint64_t* best_justified_epoch, uint8_t best_justified_root[32],
                         uint8_t proposer_boost_root[32]);

/* ---- pe_aggregate with its signature leg (bls.Aggregate, pe:659, pe:714-717) -------------------------------------- */
/* Everything pe_aggregate does, plus Attestation.signature of every aggregate it forms: the sum in G2 of the member
 * attestations' BLSSignatures, returned in the 96-byte compressed wire form (pe:37, pe:717) -- what a validator client
 * publishes as "a well-packed aggregate attestation" (pe:659).
 *   signatures        n signatures, one per input row, in host or device memory (host memory is read before the call
 *                     returns -- a buffer refilled per step is fine; device memory is read where the leg runs, which in a
 *                     pipeline may be after the call: it stays unchanged until the pipeline has completed, like device rows):
 *                     PE_SIG_G2_COMPRESSED    96 bytes each (x.c1 | x.c0 big-endian, flag bits in the leading byte): decoded
 *                                             on the device (square root in Fp2, sign and canonicity checks, curve membership);
 *                     PE_SIG_G2_UNCOMPRESSED  192 bytes each (x.c1 | x.c0 | y.c1 | y.c0): converted, trusted to lie on the curve;
 *   | PE_SIG_CHECK_SUBGROUP  additionally r * P = infinity for every decoded point (a curve point outside G2 is not a
 *                     BLSSignature; is_valid_indexed_attestation, pe:736 / pe:976, presumes members of G2);
 *   out_signatures96  96 bytes per group formed, group order as out_atts (capacity: n entries);
 *   sig_status        n entries, one per input row: PE_SIG_OK, PE_SIG_MALFORMED (encoding), PE_SIG_NOT_ON_CURVE,
 *                     PE_SIG_NOT_IN_SUBGROUP.
 * A member whose signature does not decode is left out of its group's sum and the group's row loses
 * PE_ATT_FLAG_SIGNATURE_VALID (the handlers then reject it with PE_ATT_BAD_SIGNATURE): an aggregate over an invalid
 * signature can never verify.  Members with overlapping bits: the row carries PE_ATT_FLAG_OVERLAPPING_BITS as with
 * pe_aggregate, and its signature is the plain sum (which counts a validator twice -- such an aggregate does not verify
 * either, Appendix A.8).  Rows in host or device memory, synchronous or inside a pipeline, exactly as pe_aggregate; the
 * signature sums run on the engine's state-transition stream beside the aggregate pubkeys and the fork-choice kernels. */
#define PE_SIG_G2_COMPRESSED   1u
#define PE_SIG_G2_UNCOMPRESSED 2u
#define PE_SIG_CHECK_SUBGROUP  0x100u
#define PE_SIG_OK 0
#define PE_SIG_MALFORMED 1
#define PE_SIG_NOT_ON_CURVE 2
#define PE_SIG_NOT_IN_SUBGROUP 3
int pe_aggregate_signed(pe_engine* h, const pe_attestation* atts, uint32_t n,
                        const uint8_t* bits_arena, uint64_t arena_len,
                        const uint8_t* signatures, uint32_t sig_format_flags,
                        pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                        uint8_t* out_bits_arena, uint64_t out_arena_cap,
                        uint8_t* out_signatures96, int32_t* sig_status,
                        uint8_t* out_aggpk96, uint32_t* out_count);
/* The subgroup check on its own: n points in the 192-byte uncompressed form (host memory); status[i] = PE_SIG_OK or
 * PE_SIG_NOT_IN_SUBGROUP (infinity passes). */
int pe_g2_subgroup_check(pe_engine* h, const uint8_t* points192, uint64_t n, int32_t* status);

/* ---- multi-GPU exchange (validator-range shards) ----------- */
/* Each rank owns a contiguous validator range and the whole (small) block table.
 * get_head splits at the one exchange point:
 *   pe_votes_partial  : this shard's direct vote weight per block (tree order) into a
 *                       caller-owned DEVICE buffer of n_blocks + PE_EXCHANGE_EXTRA u64;
 *                       the extra entries carry the shard's active balance / active
 *                       validator count as per-workgroup partials (the proposer boost
 *                       needs the global values, Appendix A.1).  Asynchronous on the
 *                       engine's stream.
 *   <host framework: ONE all-reduce(sum, u64) of the buffer -- RCCL via torch.distributed>
 *   pe_head_from_weights : subtree sums + descent from the reduced DEVICE buffer.
 * Integer sums: bit-exact for any reduction order. */
#define PE_EXCHANGE_EXTRA 512
int pe_votes_partial(pe_engine* h, void* dev_buf_u64, uint32_t n_blocks);
int pe_head_from_weights(pe_engine* h, const void* dev_buf_u64, uint32_t n_blocks, uint8_t out_root[32]);
/* G1: per-group partial sums in XYZZ coordinates (X|Y|ZZ|ZZZ, 4 x 48 B, Montgomery form) of this shard's
 * points into a caller-owned DEVICE buffer; after an all-gather over ranks,
 * pe_g1_finish adds the n_ranks partials per group and normalises to affine. */
#define PE_G1_PARTIAL_BYTES 192
/* dev_partials_capacity: how many partials (groups) the caller's device buffer holds; a batch that forms more groups
 * fails with PE_ERR_CAPACITY before anything is written. */
int pe_g1_partial(pe_engine* h, const uint32_t* index, const uint32_t* offsets, uint32_t n_groups,
                  void* dev_partials, uint32_t dev_partials_capacity);
int pe_g1_finish(pe_engine* h, const void* dev_gathered, uint32_t n_ranks, uint32_t n_groups,
                 uint8_t* out96);
/* pe_aggregate whose aggregate pubkeys stay projective (XYZZ) partials of THIS shard's committee
 * members, written to a caller-owned DEVICE buffer (PE_G1_PARTIAL_BYTES per group, group
 * order as in out_atts); all-gather them and call pe_g1_finish.  Asynchronous w.r.t. the
 * partials; everything else as pe_aggregate. */
int pe_aggregate_partial(pe_engine* h, const pe_attestation* atts, uint32_t n,
                         const uint8_t* bits_arena, uint64_t arena_len,
                         pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                         uint8_t* out_bits_arena, uint64_t out_arena_cap, uint32_t* out_count,
                         void* dev_partials, uint32_t dev_partials_capacity);

/* ---- RCCL inside the engine: the exchange steps behind the C ABI ---------- */
/* For clients without a collective library of their own (the cgo / bindgen / Panama bindings of INTEGRATION.md): the
 * engine owns its RCCL communicators and issues the two collectives on its own streams, between its own kernels.
 *   rank 0: pe_dist_unique_id(id); ship the PE_DIST_ID_BYTES to the other ranks (any side channel)
 *   every rank: pe_dist_init(h, id, rank, world)      -- one process per GPU, one handle per process
 *   pe_get_head_sharded   = pe_votes_partial -> ncclAllReduce(u64, sum, B + PE_EXCHANGE_EXTRA) -> pe_head_from_weights
 *   pe_aggregate_sharded  = pe_aggregate_partial -> ncclAllGather(192 B x groups) -> pe_g1_finish
 * The id holds two ncclUniqueIds: the all-reduce and the all-gather have a communicator each, because inside a
 * pipeline the G1 chain of a step (partials -> all-gather -> finish) runs on the engine's finishing stream beside the
 * next step's fork-choice kernels and their all-reduce, and one communicator must not be driven from two streams.
 * pe_aggregate_sharded takes its rows from host memory or -- like pe_aggregate -- from DEVICE memory (grouping and
 * committee resolution on the device, handlers with PE_ROWS_RESIDENT; every rank passes the same rows in the same order,
 * each with the bits of ITS members).
 * Both calls may be made inside pe_pipeline_begin(_streaming) ... _end(_lagged): nothing then waits except the poll
 * for the head; pe_aggregate_sharded's unions are handed on with PE_BITS_RESIDENT like pe_aggregate's, its outputs are
 * complete where the pipeline's are.  Every rank must make the same sequence of calls (collectives pair up by order).
 * librccl is loaded with dlopen at the first pe_dist_* call (the copy already in the process, e.g. torch's, else the
 * ROCm installation's); single-GPU use never touches it. */
#define PE_DIST_ID_BYTES 256
int pe_dist_unique_id(uint8_t out_id[PE_DIST_ID_BYTES]);
int pe_dist_init(pe_engine* h, const uint8_t id[PE_DIST_ID_BYTES], int rank, int world);
/* Flags for pe_dist_init_ex (pe_dist_init = flags 0, or PE_DIST_SINGLE_COMM when POSEVO_DIST_SINGLE_COMM=1 is set):
 *   PE_DIST_SINGLE_COMM  both collectives on ONE communicator and ONE stream (the engine's): every rank then issues them
 *                        in program order on one queue, so two collective kernels can never become resident in different
 *                        orders on different ranks.  Costs overlap: a step's all-gather (behind its G1 partials) holds
 *                        the next step's fork-choice kernels back.  The form to fall back to when the two-communicator
 *                        form times out (PE_ERR_TIMEOUT). */
#define PE_DIST_SINGLE_COMM 1u
int pe_dist_init_ex(pe_engine* h, const uint8_t id[PE_DIST_ID_BYTES], int rank, int world, uint32_t flags);
/* The same exchange through the CALLER's collectives (MPI, a host-staged shim, a test double): two function pointers
 * replace RCCL.  Both operate on DEVICE memory and are ordered on the given HIP stream: an implementation either enqueues
 * on that stream or synchronises it, exchanges and returns -- work the engine enqueues on the stream afterwards must see
 * the result.  Return 0 on success; anything else fails the engine call with PE_ERR_NO_DEVICE.  The engine-owned
 * streaming sharded step runs unchanged on top (tests/test_gpu_dist_custom.py drives it with two ranks that share one
 * GPU). */
typedef struct pe_collectives {
    void* user;
    /* in-place sum over ranks of count uint64 at dev_buf */
    int (*all_reduce_u64)(void* user, void* dev_buf, uint64_t count, void* hip_stream);
    /* rank r's bytes_per_rank bytes at dev_send land at dev_recv + r * bytes_per_rank on every rank */
    int (*all_gather)(void* user, const void* dev_send, void* dev_recv, uint64_t bytes_per_rank, void* hip_stream);
} pe_collectives;
int pe_dist_init_custom(pe_engine* h, int rank, int world, const pe_collectives* fn);
/* Bounded waits (default 30 000 ms; 0 = wait for ever; POSEVO_DIST_TIMEOUT_MS presets it): once a handle has
 * pe_dist_init'ed, the waits for enqueued work poll with this limit; past it the call aborts the communicators
 * (ncclCommAbort), returns PE_ERR_TIMEOUT and the handle refuses further sharded calls until pe_dist_destroy +
 * pe_dist_init(_ex).  A hung exchange thus surfaces as an error on every rank instead of a stuck job.  The outputs of
 * the calls and pipelines in flight at that moment are lost (pe_pipeline_completed stays where it was); the store
 * keeps what the device had applied -- votes and flags are idempotent, the caller replays the lost steps. */
int pe_dist_set_timeout_ms(pe_engine* h, uint32_t ms);
/* Upper bound of the groups one pe_aggregate_sharded over rows in DEVICE memory may form (default: its row count n).  The
 * all-gather of such a call is sized before the device has formed the groups; a caller that knows its epoch has C
 * committees sets C and ships C x 192 B per rank instead of n x 192 B.  More groups than the bound: PE_ERR_CAPACITY where
 * the outputs complete. */
int pe_dist_set_max_groups(pe_engine* h, uint32_t max_groups);
int pe_dist_destroy(pe_engine* h);
int pe_get_head_sharded(pe_engine* h, uint8_t out_root[32]);
/* pe_get_head_async's counterpart: inside a pipeline nothing waits, out_root is written where the pipeline's outputs
 * complete (a streaming caller's loop then never waits for the all-reduce inside a step). */
int pe_get_head_sharded_async(pe_engine* h, uint8_t out_root[32]);
int pe_aggregate_sharded(pe_engine* h, const pe_attestation* atts, uint32_t n,
                         const uint8_t* bits_arena, uint64_t arena_len,
                         pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                         uint8_t* out_bits_arena, uint64_t out_arena_cap, uint8_t* out_aggpk96, uint32_t* out_count);

/* ---- committee-sharded steps (Option B; pe:474 "parallelising the aggregation of attestations") ---- */
/* The other way to split an epoch over N GPUs: by COMMITTEE instead of by validator range.  Every rank holds the whole
 * registry and the whole store; rank g is handed only the rows of the committees it serves (its attestation subnets) and
 * runs pe_aggregate over them as on one GPU -- its unions and its aggregate pubkeys are complete, there is no G1
 * collective, and a rank's grouping / union / G1 work is 1/N of the epoch's.  pe_aggregate_exchange then makes the
 * epoch whole again: the aggregates of the LAST pe_aggregate over rows in device memory (AttestationData + OR-ed bits +
 * attester count + verdict flags: the aggregate attestation a validator client publishes, pe:659, pe:714-717) are packed
 * into fixed slots, all-gathered over the handle's communicator (pe_dist_init / pe_dist_init_custom), and ingested as ONE
 * batch that becomes the handle's resident aggregate: the handlers that follow
 *     pe_on_attestation_batch(h, PE_ROWS_RESIDENT, cap, PE_BITS_RESIDENT, 0, status, NULL, out_count)
 *     pe_process_attestation_batch(h, state, PE_ROWS_RESIDENT, cap, PE_BITS_RESIDENT, 0, status, out_numerators)
 * apply the whole epoch's votes and flags to this rank's copy of the store, and pe_get_head (the plain one: no weight
 * exchange) returns the same head on every rank.  Outputs: the gathered aggregates, rank after rank (out_atts rows with
 * bits_offset into out_bits_arena, out_count), cap_groups entries each -- cap_groups >= world x the bound of
 * pe_dist_set_max_groups (required, and the same on every rank: PE_ERR_STATE without one -- the exchange is sized by it).  Every rank calls it once per step; inside a
 * pipeline nothing waits (with RCCL) and the local aggregate's G1 sums keep running beside the exchange.  A union may be
 * at most max_validators_per_committee bits (pe_config). */
int pe_aggregate_exchange(pe_engine* h, pe_attestation* out_atts, uint32_t* out_n_groups, uint8_t* out_bits_arena,
                          uint64_t out_arena_cap, uint32_t* out_count, uint32_t cap_groups);

/* Measurement hooks (per-kernel event brackets, the in-situ timeline) are not part of the boundary: include/posevo_profile.h. */

#ifdef __cplusplus
}
#endif
#endif /* POSEVO_H */
