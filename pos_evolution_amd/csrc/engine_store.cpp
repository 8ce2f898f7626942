// engine_store.cpp -- the fork-choice store and its handlers.
//
// The store mirrors the reference's `Store` (pe:889-901) as flat tables: a block table (root -> insertion index, parent
// index, slot, post-state checkpoints), the three checkpoints, time, proposer_boost_root, and per-validator device
// arrays (latest message, effective balance, flags, pubkey).  Handlers follow the reference line by line where it
// defines them (on_tick pe:934-955, on_block pe:986-1036, should_update_justified_checkpoint pe:1046-1061,
// on_attester_slashing pe:1447-1461) and SURVEY.md Appendix A where it only calls them (get_ancestor A.2).
#include "engine_internal.h"

using namespace posevo;

namespace posevo {

// ------------------------------------------------------------------ tree snapshot
// DFS pre-order of the block tree + subtree sizes, root ranks and filter_block_tree's leaf test.
int refresh_tree(pe_engine* h)
{
    if (!h->tree_dirty) return PE_OK;
    if (h->held.active) PE_TRY(held_issue(h));  // held-back launches read the tables this upload rewrites
    const uint32_t n = (uint32_t)h->blocks.size();
    if (n > (uint32_t)TREE_MAX_BLOCKS)
        return fail(h, PE_ERR_CAPACITY, "block table exceeds the LDS-resident tree capacity (8192)");
    std::vector<uint32_t> first_child(n, NONE32), next_sib(n, NONE32), last_child(n, NONE32);
    for (uint32_t i = 1; i < n; ++i) {  // children in insertion order
        const uint32_t p = h->blocks[i].parent;
        if (last_child[p] == NONE32) first_child[p] = i; else next_sib[last_child[p]] = i;
        last_child[p] = i;
    }
    std::vector<uint32_t> pos_of(n), idx_of(n), size(n, 1), parent_pos(n, NONE32), stack;
    stack.reserve(64);
    uint32_t pos = 0;
    stack.push_back(0);
    std::vector<uint32_t> order;
    order.reserve(n);
    while (!stack.empty()) {  // iterative pre-order
        const uint32_t b = stack.back();
        stack.pop_back();
        pos_of[b] = pos;
        idx_of[pos] = b;
        ++pos;
        order.push_back(b);
        // push children in reverse so the first child is visited first
        uint32_t cnt = 0;
        for (uint32_t c = first_child[b]; c != NONE32; c = next_sib[c]) ++cnt;
        const size_t base = stack.size();
        stack.resize(base + cnt);
        uint32_t k = 0;
        for (uint32_t c = first_child[b]; c != NONE32; c = next_sib[c]) stack[base + cnt - 1 - k++] = c;
    }
    for (uint32_t k = n; k-- > 1;) {  // children after parents in pre-order: accumulate sizes bottom-up
        const uint32_t b = order[k];
        size[h->blocks[b].parent] += size[b];
    }
    std::vector<uint32_t> sz_pos(n), rank_pos(n);
    std::vector<uint8_t> leaf_pos(n);
    std::vector<uint32_t> by_root(n);
    std::iota(by_root.begin(), by_root.end(), 0u);
    std::sort(by_root.begin(), by_root.end(),
              [&](uint32_t a, uint32_t b) { return h->blocks[a].root < h->blocks[b].root; });  // lexicographic
    std::vector<uint32_t> rank(n);
    for (uint32_t r = 0; r < n; ++r) rank[by_root[r]] = r;
    for (uint32_t b = 0; b < n; ++b) {
        const uint32_t p = pos_of[b];
        sz_pos[p] = size[b];
        rank_pos[p] = rank[b];
        parent_pos[p] = h->blocks[b].parent == NONE32 ? NONE32 : pos_of[h->blocks[b].parent];
        const Block& blk = h->blocks[b];
        const bool correct_justified = h->justified.epoch == 0 || blk.post_justified == h->justified;
        const bool correct_finalized = h->finalized.epoch == 0 || blk.post_finalized == h->finalized;
        leaf_pos[p] = (correct_justified && correct_finalized) ? 1 : 0;
    }
    const size_t cap = std::max<size_t>(n, 64);
    HIP_TRY(h, h->d_tsize.ensure(cap * 4));
    HIP_TRY(h, h->d_tparent.ensure(cap * 4));
    HIP_TRY(h, h->d_trank.ensure(cap * 4));
    HIP_TRY(h, h->d_tleaf.ensure(cap));
    HIP_TRY(h, h->d_tpos.ensure(cap * 4));
    HIP_TRY(h, h->d_tidx.ensure(cap * 4));
    HIP_TRY(h, h->d_weights.ensure(cap * 8));
    {
        const size_t before_d = h->d_direct.cap, before_t = h->d_totals.cap;
        HIP_TRY(h, h->d_direct.ensure(cap * 8));
        HIP_TRY(h, h->d_totals.ensure(sizeof(VoteTotals) * VOTES_MAX_WG));
        // the engine's own weight buffer is zero between get_head calls: k_votes adds, k_tree clears
        if (h->d_direct.cap != before_d) HIP_TRY(h, hipMemsetAsync(h->d_direct.p, 0, h->d_direct.cap, h->stream));
        if (h->d_totals.cap != before_t) HIP_TRY(h, hipMemsetAsync(h->d_totals.p, 0, h->d_totals.cap, h->stream));
    }
    HIP_TRY(h, h->d_head.ensure(64));
    HIP_TRY(h, h->h_head.ensure(64));
    // device-side block lookups for validation of rows resident on the device (att_kernels.hip): root -> insertion index
    // (open addressing on the leading word of the root), the roots themselves, the slot of the block at each position
    uint32_t tab_size = 64;
    while (tab_size < 2 * n) tab_size <<= 1;
    std::vector<uint32_t> root_tab(tab_size, NONE32);
    std::vector<uint8_t> roots_flat(32ull * n);
    std::vector<uint64_t> slot_pos(n);
    for (uint32_t b = 0; b < n; ++b) {
        memcpy(&roots_flat[32ull * b], h->blocks[b].root.data(), 32);
        slot_pos[pos_of[b]] = h->blocks[b].slot;
        uint32_t w;
        memcpy(&w, h->blocks[b].root.data(), 4);
        uint32_t slot = w & (tab_size - 1);
        while (root_tab[slot] != NONE32) slot = (slot + 1) & (tab_size - 1);
        root_tab[slot] = b;
    }
    HIP_TRY(h, h->d_broot_tab.ensure(4ull * tab_size));
    HIP_TRY(h, h->d_broots.ensure(32ull * cap));
    HIP_TRY(h, h->d_bslot_pos.ensure(8ull * cap));
    h->broot_mask = tab_size - 1;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_broot_tab.p, root_tab.data(), 4ull * tab_size, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_broots.p, roots_flat.data(), 32ull * n, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_bslot_pos.p, slot_pos.data(), 8ull * n, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_tsize.p, sz_pos.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_tparent.p, parent_pos.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_trank.p, rank_pos.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_tleaf.p, leaf_pos.data(), n, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_tpos.p, pos_of.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_tidx.p, idx_of.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipStreamSynchronize(s));  // the host vectors die at scope exit
    h->h_pos_of_idx = pos_of;
    h->tree_dirty = false;
    return PE_OK;
}

TreeDev tree_dev(const pe_engine* h)
{
    TreeDev t;
    t.size = h->d_tsize.as<uint32_t>();
    t.parent = h->d_tparent.as<uint32_t>();
    t.rank = h->d_trank.as<uint32_t>();
    t.leaf_ok = h->d_tleaf.as<uint8_t>();
    t.pos_of_idx = h->d_tpos.as<uint32_t>();
    t.idx_of_pos = h->d_tidx.as<uint32_t>();
    t.n = (uint32_t)h->blocks.size();
    return t;
}

int insert_block(pe_engine* h, const Root& root, uint32_t parent, uint64_t slot, const Checkpoint& pj,
                 const Checkpoint& pf)
{
    Block b;
    b.root = root;
    b.parent = parent;
    b.slot = slot;
    b.post_justified = pj;
    b.post_finalized = pf;
    h->index_of.emplace(root, (uint32_t)h->blocks.size());
    h->blocks.push_back(b);
    h->tree_dirty = true;
    return PE_OK;
}

CommitteeTable* find_table(pe_engine* h, uint64_t epoch)
{
    // the rows of a batch nearly always share one target epoch: try the table of the previous hit first
    CommitteeTable* hit = nullptr;
    if (h->last_table < h->tables.size()) {
        CommitteeTable& t = h->tables[h->last_table];
        if (t.epoch == epoch && t.n_committees) hit = &t;
    }
    for (size_t i = 0; !hit && i < h->tables.size(); ++i)
        if (h->tables[i].epoch == epoch && h->tables[i].n_committees) { h->last_table = i; hit = &h->tables[i]; }
    if (hit && hit->ready_pending) {
        // shuffled by pe_compute_committees_async on the state-transition stream: whoever is about to read the table
        // enqueues on (or forks from) the engine's stream, which waits for the shuffle once
        (void)hipStreamWaitEvent(h->stream, hit->ev_ready, 0);
        hit->ready_pending = false;
    }
    return hit;
}

// Re-pack one attestation's bits into 32-bit words (zero padded, masked to n_use bits); returns popcount.
uint32_t pack_bits(const uint8_t* src, uint32_t n_use, uint32_t* dst_words)
{
    const uint32_t n_words = (n_use + 31) / 32;
    const uint32_t n_bytes = (n_use + 7) / 8;
    if (n_words == 0) return 0;
    dst_words[n_words - 1] = 0;
    memcpy(dst_words, src, n_bytes);  // little-endian host: byte k of the arena is byte k of the word stream
    if (n_use & 31) dst_words[n_words - 1] &= (1u << (n_use & 31)) - 1u;
    uint32_t cnt = 0;
    for (uint32_t w = 0; w < n_words; ++w) cnt += (uint32_t)__builtin_popcount(dst_words[w]);
    return cnt;
}

int ensure_validator_arrays(pe_engine* h, uint64_t n)
{
    const size_t n4 = (n + 3) & ~size_t(3);
    HIP_TRY(h, h->d_balance.ensure(std::max<size_t>(64, n4 * 8)));
    HIP_TRY(h, h->d_flags.ensure(std::max<size_t>(64, n4)));
    HIP_TRY(h, h->d_incr.ensure(std::max<size_t>(64, n4 * 2)));
    return PE_OK;
}

int upload_balances(pe_engine* h, uint64_t n, const uint64_t* bal, const uint8_t* flags)
{
    std::vector<uint16_t> incr(n);
    const uint64_t inc = h->cfg.effective_balance_increment;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t q = bal[i] / inc;
        if (q > 0xFFFF) return fail(h, PE_ERR_INVALID_ARG, "effective_balance / increment exceeds 65535");
        incr[i] = (uint16_t)q;
    }
    int rc = ensure_validator_arrays(h, n);
    if (rc) return rc;
    // keep equivocation marks across balance refreshes (equivocating_indices only grows, pe:1459-1461)
    std::vector<uint8_t> f(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t v = flags[i] & (PE_VAL_ACTIVE | PE_VAL_SLASHED);
        if (i < h->h_flags.size() && (h->h_flags[i] & PE_VAL_EQUIVOCATING)) v |= PE_VAL_EQUIVOCATING;
        f[i] = v;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_balance.p, bal, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_flags.p, f.data(), n, hipMemcpyHostToDevice, h->stream));
    if (!h->state_view_set) {  // the working state mirrors the registry until pe_state_set_validators says otherwise
        const size_t n4 = (n + 3) & ~size_t(3);
        HIP_TRY(h, h->d_sbalance.ensure(std::max<size_t>(64, n4 * 8)));
        HIP_TRY(h, h->d_sflags.ensure(std::max<size_t>(64, n4)));
        std::vector<uint8_t> sf(n);
        for (uint64_t i = 0; i < n; ++i)  // active now => also counted as active in the previous epoch
            sf[i] = (uint8_t)((flags[i] & (PE_VAL_ACTIVE | PE_VAL_SLASHED)) | ((flags[i] & PE_VAL_ACTIVE) ? PE_VAL_ACTIVE_PREV : 0));
        HIP_TRY(h, hipMemcpyAsync(h->d_incr.p, incr.data(), n * 2, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_sbalance.p, bal, n * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_sflags.p, sf.data(), n, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->h_flags.swap(f);
    return PE_OK;
}

// get_head's tree launch as an argument block: the store's scalars of NOW (justified root, proposer boost) by value, the
// head index into head_word (host-coherent pinned memory: no D2H copy), which reads NONE32 until the kernel has run.
int tree_args(pe_engine* h, uint64_t* d_direct, const VoteTotals* d_totals, int clear_direct, uint32_t* head_word, TreeArgs* out)
{
    uint32_t just_idx;
    if (!find_block(h, h->justified.root, &just_idx))
        return fail(h, PE_ERR_UNKNOWN_ROOT, "justified checkpoint root is not in the store");
    uint32_t boost_pos = NONE32;
    if (!is_zero_root(h->boost_root)) {
        uint32_t bi;
        if (find_block(h, h->boost_root, &bi)) boost_pos = h->h_pos_of_idx[bi];
    }
    *const_cast<volatile uint32_t*>(head_word) = NONE32;
    *out = TreeArgs{tree_dev(h), d_direct, d_totals, 0, 0, 0, h->h_pos_of_idx[just_idx], boost_pos, h->cfg.slots_per_epoch,
                    h->cfg.proposer_score_boost, h->cfg.effective_balance_increment, h->d_weights.as<uint64_t>(), head_word,
                    clear_direct};
    return PE_OK;
}
// ... and its votes launch over the engine's own weight buffer
VotesArgs votes_args(const pe_engine* h)
{
    return VotesArgs{h->d_vote_block.as<uint32_t>(), h->d_balance.as<uint64_t>(), h->d_flags.as<uint8_t>(), h->n_val,
                     h->cfg.filter_slashed, h->d_tpos.as<uint32_t>(), (uint32_t)h->blocks.size(), h->d_direct.as<uint64_t>(),
                     h->d_totals.as<VoteTotals>(), expiry_slots_ptr(h), min_vote_slot(h)};
}

// get_head's device part on arbitrary weight buffer.
int run_tree(pe_engine* h, uint64_t* d_direct, const VoteTotals* d_totals, int clear_direct, uint32_t* head_out,
             uint32_t* async_word)
{
    // async_word: a slot of the pinned output block instead of the engine's head word -- pe_get_head_async: nobody polls,
    // the call's completion reads it when the pipeline's outputs are complete
    volatile uint32_t* head_word = async_word ? async_word : h->h_head.as<uint32_t>();
    TreeArgs ta;
    PE_TRY(tree_args(h, d_direct, d_totals, clear_direct, const_cast<uint32_t*>(head_word), &ta));
    {
        ProfScope ps(h, PE_KERNEL_TREE);
        launch_tree(h->stream, ta, /*lean=*/h->pipelining ? 1 : 0);  // inside a pipeline: the shape that fits beside an accumulation
    }
    HIP_TRY(h, hipGetLastError());
    // a streaming pipeline's G1 sums go out now, ordered behind k_tree on the device: they start the moment the head
    // is known, and their launch calls overlap the fork-choice kernels instead of following the poll below
    if (h->streaming) {
        PE_TRY(run_deferred(h));
        complete_oldest_if_ready(h);
    }
    if (async_word) return PE_OK;
    // k_tree's last act is a system-scope release store of the head index into this host-coherent word: polling it
    // sees the result a few microseconds before hipStreamSynchronize returns.  Bounded: after ~200 us (a hung or
    // faulted kernel) the stream sync takes over and reports the error.
    bool seen = false;
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 0;; ++it) {
            if (*head_word != NONE32) { seen = true; break; }
            if ((it & 63) == 63 &&
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 200.0)
                break;
        }
    }
    if (!seen) {
        const hipError_t e = bounded_stream_sync(h, h->stream);  // a sharded head waits behind an all-reduce
        if (e == hipErrorNotReady)
            return fail(h, PE_ERR_TIMEOUT, "get_head: the weights' all-reduce did not complete within the bounded wait; "
                                           "the communicators were aborted");
        HIP_TRY(h, e);
    }
    *head_out = *head_word;
    if (*head_out >= h->blocks.size()) return fail(h, PE_ERR_NO_DEVICE, "tree kernel returned an invalid head index");
    return PE_OK;
}

}  // namespace posevo

extern "C" {

// ---------------------------------------------------------------- store
int pe_store_init(pe_engine* h, uint64_t genesis_time, uint64_t anchor_slot, const uint8_t anchor_root[32])
{
    if (!h || !anchor_root) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    h->blocks.clear();
    h->index_of.clear();
    // nothing of the previous store may resolve against the new one: committee tables (get_beacon_committee of the old
    // chain's states), the working-state view and its participation arrays, the resident aggregate
    for (auto& t : h->tables) { t.n_committees = 0; t.offsets.clear(); t.is_partition = false; t.stamp = 0; }
    h->state_view_set = false;
    h->res_valid = false;
    h->rr.valid = false;
    if (h->n_val) {
        const size_t n4 = (h->n_val + 3) & ~size_t(3);
        HIP_TRY(h, hipMemsetAsync(h->d_part_cur.p, 0, n4, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_part_prev.p, 0, n4, h->stream));
        // the working state mirrors the registry again until pe_state_set_validators says otherwise
        HIP_TRY(h, hipMemcpyAsync(h->d_sbalance.p, h->d_balance.p, 8 * h->n_val, hipMemcpyDeviceToDevice, h->stream));
        launch_state_view_from_registry(h->stream, h->d_flags.as<uint8_t>(), h->d_balance.as<uint64_t>(),
                                        h->cfg.effective_balance_increment, h->n_val, h->d_sflags.as<uint8_t>(),
                                        h->d_incr.as<uint16_t>());
    }
    h->genesis_time = genesis_time;
    h->time = genesis_time + h->cfg.seconds_per_slot * anchor_slot;       // pe:1085
    const uint64_t anchor_epoch = anchor_slot / h->cfg.slots_per_epoch;   // get_current_epoch(anchor_state)
    Checkpoint cp;
    cp.epoch = anchor_epoch;
    cp.root = to_root(anchor_root);
    h->justified = h->finalized = h->best_justified = cp;                 // pe:1081-1082, 1089
    h->boost_root = Root{};                                               // pe:1083
    // the anchor's own post-state checkpoints are not given by get_forkchoice_store; the anchor is
    // only ever a leaf while nothing descends from it, and then get_head returns it regardless.
    insert_block(h, cp.root, NONE32, anchor_slot, cp, cp);
    for (auto& f : h->h_flags) f &= (uint8_t)~PE_VAL_EQUIVOCATING;        // equivocating_indices = set()
    if (h->n_val) {
        HIP_TRY(h, hipMemcpyAsync(h->d_flags.p, h->h_flags.data(), h->n_val, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_vote_key.p, 0, h->n_val * 8, h->stream));       // latest_messages = {}
        HIP_TRY(h, hipMemsetAsync(h->d_vote_block.p, 0xFF, h->n_val * 4, h->stream));
        if (h->cfg.vote_expiry_slots) HIP_TRY(h, hipMemsetAsync(h->d_vote_slot.p, 0, h->n_val * 4, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->initialised = true;
    return PE_OK;
}

int pe_set_validators(pe_engine* h, uint64_t n, const uint8_t* pubkeys96, const uint64_t* effective_balance,
                      const uint8_t* flags)
{
    if (!h || (n && (!effective_balance || !flags))) return PE_ERR_INVALID_ARG;
    if (n >= 0xFFFFFFFFull) return fail(h, PE_ERR_CAPACITY, "validator index must fit 32 bits");
    PE_TRY(enter(h));
    const uint64_t old_n = h->n_val;
    int rc = upload_balances(h, n, effective_balance, flags);
    if (rc) return rc;
    const size_t n4 = (n + 3) & ~size_t(3);
    HIP_TRY(h, h->d_vote_key.ensure(std::max<size_t>(64, n4 * 8), true, h->stream));
    HIP_TRY(h, h->d_vote_block.ensure(std::max<size_t>(64, n4 * 4), true, h->stream));
    if (h->cfg.vote_expiry_slots) HIP_TRY(h, h->d_vote_slot.ensure(std::max<size_t>(64, n4 * 4), true, h->stream));
    HIP_TRY(h, h->d_part_cur.ensure(std::max<size_t>(64, n4), true, h->stream));
    HIP_TRY(h, h->d_part_prev.ensure(std::max<size_t>(64, n4), true, h->stream));
    if (n > old_n) {  // new validators: no latest message, no participation
        HIP_TRY(h, hipMemsetAsync(h->d_vote_key.as<uint64_t>() + old_n, 0, (n4 - old_n) * 8, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_vote_block.as<uint32_t>() + old_n, 0xFF, (n4 - old_n) * 4, h->stream));
        if (h->cfg.vote_expiry_slots)
            HIP_TRY(h, hipMemsetAsync(h->d_vote_slot.as<uint32_t>() + old_n, 0, (n4 - old_n) * 4, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_part_cur.as<uint8_t>() + old_n, 0, n4 - old_n, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_part_prev.as<uint8_t>() + old_n, 0, n4 - old_n, h->stream));
    }
    if (pubkeys96 && n) {
        HIP_TRY(h, h->d_points.ensure(4ull * G1_ROW_WORDS * n));
        h->points29_valid = false;
        // convert in chunks through a bounded device staging buffer
        const uint64_t chunk = std::min<uint64_t>(n, 1u << 20);
        HIP_TRY(h, h->d_tmp_be.ensure(96ull * chunk));
        for (uint64_t base = 0; base < n; base += chunk) {
            const uint64_t m = std::min(chunk, n - base);
            HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, pubkeys96 + 96ull * base, 96ull * m, hipMemcpyHostToDevice,
                                      h->stream));
            launch_g1_convert(h->stream, h->d_tmp_be.as<uint8_t>(), h->d_points.as<uint32_t>() + (uint64_t)G1_ROW_WORDS * base, m);
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
        HIP_TRY(h, hipGetLastError());
        h->have_points = true;
        PE_TRY(build_points29(h, n));  // the registry in the accumulation's field form, now: never inside a G1 launch
    } else if (!pubkeys96) {
        h->have_points = h->have_points && n <= old_n;
    }
    if (h->d_totals.p) HIP_TRY(h, hipMemsetAsync(h->d_totals.p, 0, h->d_totals.cap, h->stream));  // grid may shrink
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_val = n;
    return PE_OK;
}

int pe_set_balances(pe_engine* h, uint64_t n, const uint64_t* effective_balance, const uint8_t* flags)
{
    if (!h || !effective_balance || !flags) return PE_ERR_INVALID_ARG;
    if (n != h->n_val) return fail(h, PE_ERR_INVALID_ARG, "pe_set_balances: n differs from the registry size");
    PE_TRY(enter(h));
    return upload_balances(h, n, effective_balance, flags);
}

int pe_on_tick(pe_engine* h, uint64_t time)
{
    int rc = need_init(h, /*flush=*/false, /*keep_held=*/true);  // host-side scalars only
    if (rc) return rc;
    if (time < h->genesis_time) return fail(h, PE_ERR_INVALID_ARG, "time before genesis");
    const uint64_t previous_slot = current_slot(h);
    h->time = time;                                                    // pe:938
    const uint64_t cur = current_slot(h);
    if (cur > previous_slot) h->boost_root = Root{};                   // pe:943-944
    if (!(cur > previous_slot && slots_since_epoch_start(h, cur) == 0)) return PE_OK;  // pe:947-948
    if (h->best_justified.epoch > h->justified.epoch) {                // pe:951-955
        const uint64_t finalized_slot = start_slot(h, h->finalized.epoch);
        uint32_t bj, fi;
        if (find_block(h, h->best_justified.root, &bj) && find_block(h, h->finalized.root, &fi) &&
            get_ancestor(h, bj, finalized_slot) == fi) {
            h->justified = h->best_justified;
            h->tree_dirty = true;
        }
    }
    return PE_OK;
}

static int add_block_common(pe_engine* h, const uint8_t* root, const uint8_t* parent_root, uint64_t slot,
                            uint64_t pj_epoch, const uint8_t* pj_root, uint64_t pf_epoch, const uint8_t* pf_root,
                            bool handler)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!root || !parent_root || !pj_root || !pf_root) return PE_ERR_INVALID_ARG;
    uint32_t parent;
    if (!find_block(h, to_root(parent_root), &parent)) return fail(h, PE_ERR_UNKNOWN_PARENT, "unknown parent");  // pe:990
    const Root r = to_root(root);
    if (h->index_of.count(r)) {
        // store.blocks[root] = block is idempotent in the reference; the table keeps the first insertion
        return handler ? PE_OK : fail(h, PE_ERR_DUPLICATE_BLOCK, "duplicate block");
    }
    if (h->blocks.size() >= (size_t)TREE_MAX_BLOCKS) return fail(h, PE_ERR_CAPACITY, "block table full (8192)");
    Checkpoint pj, pf;
    pj.epoch = pj_epoch; pj.root = to_root(pj_root);
    pf.epoch = pf_epoch; pf.root = to_root(pf_root);
    if (slot <= h->blocks[parent].slot) return fail(h, PE_ERR_INVALID_ARG, "block slot must exceed its parent's slot");
    if (handler) {
        if (current_slot(h) < slot) return fail(h, PE_ERR_FUTURE_BLOCK, "block from the future");          // pe:994
        const uint64_t finalized_slot = start_slot(h, h->finalized.epoch);
        if (!(slot > finalized_slot)) return fail(h, PE_ERR_NOT_AFTER_FINALIZED, "slot <= finalized slot");  // pe:998
        uint32_t fi;
        if (!find_block(h, h->finalized.root, &fi) || get_ancestor(h, parent, finalized_slot) != fi)
            return fail(h, PE_ERR_NOT_FINALIZED_DESCENDANT, "not a descendant of the finalized checkpoint");  // pe:1000
    }
    insert_block(h, r, parent, slot, pj, pf);                                                               // pe:1016-1018
    if (!handler) return PE_OK;
    // proposer boost (pe:1020-1024)
    const uint64_t time_into_slot = (h->time - h->genesis_time) % h->cfg.seconds_per_slot;
    const bool before_attesting = time_into_slot < h->cfg.seconds_per_slot / h->cfg.intervals_per_slot;
    if (current_slot(h) == slot && before_attesting) h->boost_root = r;
    // justified checkpoint (pe:1027-1031)
    if (pj.epoch > h->justified.epoch) {
        if (pj.epoch > h->best_justified.epoch) h->best_justified = pj;
        // should_update_justified_checkpoint (pe:1046-1061)
        bool update = false;
        if (slots_since_epoch_start(h, current_slot(h)) < h->cfg.safe_slots_to_update_justified) {
            update = true;
        } else {
            const uint64_t justified_slot = start_slot(h, h->justified.epoch);
            uint32_t nj, cj;
            update = find_block(h, pj.root, &nj) && find_block(h, h->justified.root, &cj) &&
                     get_ancestor(h, nj, justified_slot) == cj;
        }
        if (update) h->justified = pj;
    }
    // finalized checkpoint (pe:1034-1036)
    if (pf.epoch > h->finalized.epoch) {
        h->finalized = pf;
        h->justified = pj;
    }
    h->tree_dirty = true;
    return PE_OK;
}

int pe_on_block(pe_engine* h, const uint8_t root[32], const uint8_t parent_root[32], uint64_t slot,
                uint64_t pj_epoch, const uint8_t pj_root[32], uint64_t pf_epoch, const uint8_t pf_root[32])
{
    return add_block_common(h, root, parent_root, slot, pj_epoch, pj_root, pf_epoch, pf_root, true);
}
int pe_add_block(pe_engine* h, const uint8_t root[32], const uint8_t parent_root[32], uint64_t slot,
                 uint64_t pj_epoch, const uint8_t pj_root[32], uint64_t pf_epoch, const uint8_t pf_root[32])
{
    return add_block_common(h, root, parent_root, slot, pj_epoch, pj_root, pf_epoch, pf_root, false);
}

int pe_set_checkpoints(pe_engine* h, uint64_t je, const uint8_t jr[32], uint64_t fe, const uint8_t fr[32])
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!jr || !fr) return PE_ERR_INVALID_ARG;
    uint32_t tmp;
    if (!find_block(h, to_root(jr), &tmp)) return fail(h, PE_ERR_UNKNOWN_ROOT, "justified root unknown");
    h->justified.epoch = je; h->justified.root = to_root(jr);
    h->finalized.epoch = fe; h->finalized.root = to_root(fr);
    if (h->best_justified.epoch < je) h->best_justified = h->justified;
    h->tree_dirty = true;
    return PE_OK;
}

int pe_set_proposer_boost(pe_engine* h, const uint8_t root[32])
{
    int rc = need_init(h, /*flush=*/false, /*keep_held=*/true);  // a host-side scalar (read where get_head is enqueued), like on_tick's
    if (rc) return rc;
    if (!root) return PE_ERR_INVALID_ARG;
    const Root r = to_root(root);
    uint32_t tmp;
    if (!is_zero_root(r) && !find_block(h, r, &tmp)) return fail(h, PE_ERR_UNKNOWN_ROOT, "boost root unknown");
    h->boost_root = r;
    return PE_OK;
}

int pe_mark_equivocating(pe_engine* h, const uint64_t* indices, uint64_t n)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (n && !indices) return PE_ERR_INVALID_ARG;
    for (uint64_t i = 0; i < n; ++i)
        if (indices[i] >= h->n_val) return fail(h, PE_ERR_INVALID_ARG, "validator index out of range");
    for (uint64_t i = 0; i < n; ++i) h->h_flags[indices[i]] |= PE_VAL_EQUIVOCATING;
    if (n) {
        HIP_TRY(h, hipMemcpyAsync(h->d_flags.p, h->h_flags.data(), h->n_val, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return PE_OK;
}

static bool sorted_unique_nonempty(const uint64_t* idx, uint64_t n)
{
    if (n == 0) return false;
    for (uint64_t i = 1; i < n; ++i)
        if (!(idx[i - 1] < idx[i])) return false;
    return true;
}

int pe_on_attester_slashing(pe_engine* h, const pe_attestation* d1, const uint64_t* i1, uint64_t n1,
                            const pe_attestation* d2, const uint64_t* i2, uint64_t n2)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!d1 || !d2) return PE_ERR_INVALID_ARG;
    // is_slashable_attestation_data (pe:1134-1143)
    const bool double_vote = !att_data_equal(*d1, *d2) && d1->target_epoch == d2->target_epoch;
    const bool surround = d1->source_epoch < d2->source_epoch && d2->target_epoch < d1->target_epoch;
    if (!(double_vote || surround)) return fail(h, PE_ERR_NOT_SLASHABLE, "attestation data not slashable");
    // is_valid_indexed_attestation (A.7): structure + injected signature verdict
    if (!sorted_unique_nonempty(i1, n1) || !(d1->flags & PE_ATT_FLAG_SIGNATURE_VALID) ||
        !sorted_unique_nonempty(i2, n2) || !(d2->flags & PE_ATT_FLAG_SIGNATURE_VALID))
        return fail(h, PE_ERR_INVALID_INDEXED, "invalid indexed attestation");
    std::vector<uint64_t> inter;
    std::set_intersection(i1, i1 + n1, i2, i2 + n2, std::back_inserter(inter));
    for (uint64_t v : inter)
        if (v >= h->n_val) return fail(h, PE_ERR_INVALID_ARG, "validator index out of range");
    return pe_mark_equivocating(h, inter.data(), inter.size());  // pe:1459-1461
}

int pe_set_committees(pe_engine* h, uint64_t epoch, uint32_t n_committees, const uint32_t* offsets,
                      const uint32_t* members)
{
    if (!h || !offsets || (n_committees && offsets[n_committees] && !members)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n_committees == 0 || n_committees % h->cfg.slots_per_epoch != 0)
        return fail(h, PE_ERR_INVALID_ARG, "n_committees must be a positive multiple of SLOTS_PER_EPOCH");
    if (offsets[0] != 0) return fail(h, PE_ERR_INVALID_ARG, "offsets[0] must be 0");
    for (uint32_t c = 0; c < n_committees; ++c)
        if (offsets[c + 1] < offsets[c]) return fail(h, PE_ERR_INVALID_ARG, "offsets not monotone");
    const uint32_t total = offsets[n_committees];
    std::vector<uint8_t> seen(h->n_val, 0);
    bool partition = true;
    for (uint32_t i = 0; i < total; ++i) {
        if (members[i] >= h->n_val) return fail(h, PE_ERR_INVALID_ARG, "committee member index out of range");
        if (seen[members[i]]) partition = false;
        seen[members[i]] = 1;
    }
    // within one committee members must be distinct (a committee is a slice of a permutation)
    if (!partition) {
        std::vector<uint32_t> tmp;
        for (uint32_t c = 0; c < n_committees; ++c) {
            tmp.assign(members + offsets[c], members + offsets[c + 1]);
            std::sort(tmp.begin(), tmp.end());
            if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end())
                return fail(h, PE_ERR_INVALID_ARG, "duplicate member inside a committee");
        }
    }
    CommitteeTable* t = find_table(h, epoch);
    if (!t) {
        if (h->tables.size() < (h->cfg.max_committee_tables ? h->cfg.max_committee_tables : 4u)) {
            h->tables.emplace_back();
            t = &h->tables.back();
        } else {
            t = &*std::min_element(h->tables.begin(), h->tables.end(),
                                   [](const CommitteeTable& a, const CommitteeTable& b) { return a.stamp < b.stamp; });
        }
    }
    HIP_TRY(h, t->d_members.ensure(std::max<size_t>(64, 4ull * total)));
    HIP_TRY(h, t->d_offsets.ensure(4ull * (n_committees + 1)));
    HIP_TRY(h, hipMemcpyAsync(t->d_members.p, members, 4ull * total, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(t->d_offsets.p, offsets, 4ull * (n_committees + 1), hipMemcpyHostToDevice, h->stream));
    if (partition && h->n_val) {
        HIP_TRY(h, t->d_inv_comm.ensure(4ull * h->n_val));
        HIP_TRY(h, t->d_inv_pos.ensure(4ull * h->n_val));
        launch_invert_committees(h->stream, t->d_members.as<uint32_t>(), t->d_offsets.as<uint32_t>(), n_committees,
                                 t->d_inv_comm.as<uint32_t>(), t->d_inv_pos.as<uint32_t>(), h->n_val);
        HIP_TRY(h, hipGetLastError());
    }
    t->n_val_at_load = h->n_val;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    t->epoch = epoch;
    t->n_committees = n_committees;
    t->offsets.assign(offsets, offsets + n_committees + 1);
    t->is_partition = partition;
    t->stamp = ++h->table_stamp;
    return PE_OK;
}

// asynchronous = true: pe_compute_committees_async -- nothing is waited for and nothing read back; the kernels go to a
// stream of their own, and the engine's stream waits for them the first time the table is read (find_table)
static int compute_committees_impl(pe_engine* h, uint64_t epoch, const uint8_t seed[32], const uint32_t* active_indices,
                                   uint32_t n_active, uint32_t n_committees, uint32_t shuffle_round_count,
                                   uint32_t* out_offsets, uint32_t* out_members, bool asynchronous)
{
    if (!h || !seed) return PE_ERR_INVALID_ARG;
    if (asynchronous) (void)hipSetDevice(h->device);
    else PE_TRY(enter(h));
    HostLap lap(&h->trace);
    if (n_committees == 0 || n_committees % h->cfg.slots_per_epoch != 0)
        return fail(h, PE_ERR_INVALID_ARG, "n_committees must be a positive multiple of SLOTS_PER_EPOCH");
    if (shuffle_round_count > 255) return fail(h, PE_ERR_INVALID_ARG, "shuffle_round_count is a uint8 in the spec");
    // active_indices NULL = every validator 0 .. n_active - 1 is active (get_active_validator_indices of a registry
    // without pending or exited validators): nothing to validate, nothing to upload
    const bool identity = active_indices == nullptr;
    if (identity) {
        if (n_active > h->n_val) return fail(h, PE_ERR_INVALID_ARG, "n_active exceeds the registry");
    } else {   // the active set: distinct validator indices (get_active_validator_indices is increasing)
        bool increasing = true;
        for (uint32_t i = 0; i < n_active && increasing; ++i) {
            if (active_indices[i] >= h->n_val) return fail(h, PE_ERR_INVALID_ARG, "active index out of range");
            if (i && active_indices[i] <= active_indices[i - 1]) increasing = false;
        }
        if (!increasing) {  // not sorted: the general distinctness check
            std::vector<uint8_t> seen(h->n_val, 0);
            for (uint32_t i = 0; i < n_active; ++i) {
                if (active_indices[i] >= h->n_val) return fail(h, PE_ERR_INVALID_ARG, "active index out of range");
                if (seen[active_indices[i]]) return fail(h, PE_ERR_INVALID_ARG, "duplicate active index");
                seen[active_indices[i]] = 1;
            }
        }
    }
    lap.mark("comm.1_validate");
    std::vector<uint32_t> offsets(n_committees + 1);
    for (uint32_t c = 0; c <= n_committees; ++c)
        offsets[c] = (uint32_t)(((uint64_t)n_active * c) / n_committees);  // start/end of pe:502-503
    CommitteeTable* t = find_table(h, epoch);
    if (!t) {
        if (h->tables.size() < (h->cfg.max_committee_tables ? h->cfg.max_committee_tables : 4u)) {
            h->tables.emplace_back();
            t = &h->tables.back();
        } else {
            t = &*std::min_element(h->tables.begin(), h->tables.end(),
                                   [](const CommitteeTable& a, const CommitteeTable& b) { return a.stamp < b.stamp; });
        }
    }
    if (asynchronous && t->n_committees) {
        // The table (the epoch's own, shuffled again, or the least recently used one) is rewritten in place: work still
        // in flight must not be reading it.  A pipeline notes the stamp counter at its begin and every use re-stamps a
        // table, so a table stamped after the oldest pipeline in flight began may be one of its tables -- then everything
        // in flight completes first (a caller that streams with lag depth L keeps at least L + 3 tables and shuffles an
        // epoch before its first use: it never waits here)
        uint64_t oldest = h->table_stamp + 1;
        for (int k = 0; k < h->n_arenas; ++k) {
            const pe_engine::PipeArena& a = h->arena[k];
            if (a.fenced || !a.pending.empty()) oldest = std::min(oldest, a.table_stamp_at_begin);
        }
        if (t->stamp >= oldest) PE_TRY(flush_pending(h));
    }
    // an earlier asynchronous shuffle into this slot (its staging pair and the table's arrays are about to be rewritten):
    // let it finish; a completed or never recorded event returns at once
    if (t->ev_ready) HIP_TRY(h, hipEventSynchronize(t->ev_ready));
    t->ready_pending = false;
    const uint32_t nb = (n_active + 255) / 256;
    if (asynchronous) {
        // Nothing of this call touches the pipelines' arenas or the engine's stream: the table carries its own small
        // staging pair (seed | offsets | active indices), the shuffle has its own stream and scratch, and the table's
        // event is what a later reader waits for (find_table).
        if (!h->prep_stream) {
            // normal priority: neither the lowest nor the highest the device offers placed the shuffle any better beside a
            // streaming step (0.56-0.72 ms with-shuffle against 0.55, round 3)
            HIP_TRY(h, hipStreamCreateWithFlags(&h->prep_stream, hipStreamNonBlocking));
        }
        const size_t o_seed = 0, o_offs = 64, o_idx = (64 + 4ull * (n_committees + 1) + 63) & ~size_t(63);
        const size_t bytes = o_idx + (identity ? 0 : 4ull * n_active) + 64;
        HIP_TRY(h, t->h_stage.ensure(bytes));
        HIP_TRY(h, t->d_stage.ensure(bytes));
        uint8_t* hs = t->h_stage.as<uint8_t>();
        uint32_t* sw = reinterpret_cast<uint32_t*>(hs + o_seed);
        for (int i = 0; i < 8; ++i)
            sw[i] = ((uint32_t)seed[4 * i] << 24) | ((uint32_t)seed[4 * i + 1] << 16) | ((uint32_t)seed[4 * i + 2] << 8) | seed[4 * i + 3];
        memcpy(hs + o_offs, offsets.data(), 4ull * (n_committees + 1));
        if (!identity && n_active) memcpy(hs + o_idx, active_indices, 4ull * n_active);
        HIP_TRY(h, t->d_members.ensure(std::max<size_t>(64, 4ull * n_active)));
        HIP_TRY(h, t->d_offsets.ensure(4ull * (n_committees + 1)));
        HIP_TRY(h, h->d_shuffle_scratch.ensure(std::max<size_t>(64, 32ull * nb * shuffle_round_count + 4ull * shuffle_round_count + 64)));
        if (h->n_val) {
            HIP_TRY(h, t->d_inv_comm.ensure(4ull * h->n_val));
            HIP_TRY(h, t->d_inv_pos.ensure(4ull * h->n_val));
        }
        hipStream_t ps = h->prep_stream;
        uint8_t* ds = t->d_stage.as<uint8_t>();
        HIP_TRY(h, hipMemcpyAsync(ds, hs, bytes, hipMemcpyHostToDevice, ps));
        uint32_t* d_source = h->d_shuffle_scratch.as<uint32_t>();
        uint32_t* d_pivots = d_source + 8ull * nb * shuffle_round_count;
        launch_shuffle(ps, reinterpret_cast<uint32_t*>(ds + o_seed), n_active, shuffle_round_count, d_source, d_pivots,
                       identity ? nullptr : reinterpret_cast<uint32_t*>(ds + o_idx), t->d_members.as<uint32_t>());
        HIP_TRY(h, hipMemcpyAsync(t->d_offsets.p, ds + o_offs, 4ull * (n_committees + 1), hipMemcpyDeviceToDevice, ps));
        if (h->n_val)
            launch_invert_committees(ps, t->d_members.as<uint32_t>(), t->d_offsets.as<uint32_t>(), n_committees,
                                     t->d_inv_comm.as<uint32_t>(), t->d_inv_pos.as<uint32_t>(), h->n_val);
        HIP_TRY(h, hipGetLastError());
        if (!t->ev_ready) HIP_TRY(h, hipEventCreateWithFlags(&t->ev_ready, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(t->ev_ready, ps));
        t->ready_pending = true;  // find_table makes the engine's stream wait before the table is read
        t->epoch = epoch;
        t->n_committees = n_committees;
        t->offsets.swap(offsets);
        t->is_partition = true;
        t->n_val_at_load = h->n_val;
        t->stamp = ++h->table_stamp;
        lap.mark("comm.async_launch");
        return PE_OK;
    }
    Stage st(h);
    PE_TRY(st.reserve(64 + (identity ? 0 : 4ull * n_active) + 4ull * (n_committees + 1) + 1024));
    const size_t off_seed = st.alloc(32);
    const size_t off_idx = st.alloc(identity ? 4 : 4ull * n_active + 4);
    const size_t off_offs = st.alloc(4ull * (n_committees + 1));
    uint32_t* sw = st.host<uint32_t>(off_seed);
    for (int i = 0; i < 8; ++i)
        sw[i] = ((uint32_t)seed[4 * i] << 24) | ((uint32_t)seed[4 * i + 1] << 16) | ((uint32_t)seed[4 * i + 2] << 8) | seed[4 * i + 3];
    if (!identity && n_active) memcpy(st.host<uint32_t>(off_idx), active_indices, 4ull * n_active);
    memcpy(st.host<uint32_t>(off_offs), offsets.data(), 4ull * (n_committees + 1));
    HIP_TRY(h, t->d_members.ensure(std::max<size_t>(64, 4ull * n_active)));
    HIP_TRY(h, t->d_offsets.ensure(4ull * (n_committees + 1)));
    HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(64, 32ull * nb * shuffle_round_count + 4ull * shuffle_round_count + 64)));
    HIP_TRY(h, st.upload());
    hipStream_t cs = h->stream;
    uint32_t* d_source = h->d_tmp_be.as<uint32_t>();
    uint32_t* d_pivots = d_source + 8ull * nb * shuffle_round_count;
    launch_shuffle(cs, st.dev<uint32_t>(off_seed), n_active, shuffle_round_count, d_source, d_pivots,
                   identity ? nullptr : st.dev<uint32_t>(off_idx), t->d_members.as<uint32_t>());
    HIP_TRY(h, hipMemcpyAsync(t->d_offsets.p, st.dev<uint32_t>(off_offs), 4ull * (n_committees + 1),
                              hipMemcpyDeviceToDevice, cs));
    if (h->n_val) {
        HIP_TRY(h, t->d_inv_comm.ensure(4ull * h->n_val));
        HIP_TRY(h, t->d_inv_pos.ensure(4ull * h->n_val));
        launch_invert_committees(cs, t->d_members.as<uint32_t>(), t->d_offsets.as<uint32_t>(), n_committees,
                                 t->d_inv_comm.as<uint32_t>(), t->d_inv_pos.as<uint32_t>(), h->n_val);
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("comm.2_launch");
    // the table stays on the device; the members come back only on request, through the pinned block (a pageable
    // 4 MB device-to-host copy is staged by the runtime in small pieces: ~2 ms against 0.3)
    OutBlock ob(h);
    size_t off_mem = 0;
    if (out_members && n_active) {
        off_mem = ob.alloc(4ull * n_active);
        PE_TRY(ob.ensure());
        HIP_TRY(h, hipMemcpyAsync(ob.host<uint32_t>(off_mem), t->d_members.p, 4ull * n_active, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    lap.mark("comm.3_wait");
    if (out_members && n_active) memcpy(out_members, ob.host<uint32_t>(off_mem), 4ull * n_active);
    if (out_offsets) memcpy(out_offsets, offsets.data(), 4ull * (n_committees + 1));
    t->epoch = epoch;
    t->n_committees = n_committees;
    t->offsets.swap(offsets);
    t->is_partition = true;  // a permutation of distinct indices, sliced
    t->n_val_at_load = h->n_val;
    t->stamp = ++h->table_stamp;
    lap.mark("comm.4_outputs");
    return PE_OK;
}

int pe_compute_committees(pe_engine* h, uint64_t epoch, const uint8_t seed[32], const uint32_t* active_indices,
                          uint32_t n_active, uint32_t n_committees, uint32_t shuffle_round_count,
                          uint32_t* out_offsets, uint32_t* out_members)
{
    return compute_committees_impl(h, epoch, seed, active_indices, n_active, n_committees, shuffle_round_count, out_offsets,
                                   out_members, false);
}
int pe_compute_committees_async(pe_engine* h, uint64_t epoch, const uint8_t seed[32], const uint32_t* active_indices,
                                uint32_t n_active, uint32_t n_committees, uint32_t shuffle_round_count)
{
    return compute_committees_impl(h, epoch, seed, active_indices, n_active, n_committees, shuffle_round_count, nullptr,
                                   nullptr, true);
}

// ---------------------------------------------------------------- get_head
int pe_get_head(pe_engine* h, uint8_t out_root[32])
{
    int rc = need_init(h, /*flush=*/false);  // ordered behind the enqueued batch calls on the stream: no wait needed
    if (rc) return rc;
    if (!out_root) return PE_ERR_INVALID_ARG;
    HostLap lap(&h->trace);
    rc = refresh_tree(h);
    if (rc) return rc;
    {
        uint32_t tmp;  // fail before anything is launched: k_votes adds into a buffer only k_tree clears
        if (!find_block(h, h->justified.root, &tmp))
            return fail(h, PE_ERR_UNKNOWN_ROOT, "justified checkpoint root is not in the store");
    }
    {
        ProfScope ps(h, PE_KERNEL_VOTES);
        launch_votes(h->stream, h->d_vote_block.as<uint32_t>(), h->d_balance.as<uint64_t>(), h->d_flags.as<uint8_t>(),
                     h->n_val, h->cfg.filter_slashed, h->d_tpos.as<uint32_t>(), (uint32_t)h->blocks.size(),
                     h->d_direct.as<uint64_t>(), h->d_totals.as<VoteTotals>(), 0, expiry_slots_ptr(h),
                     min_vote_slot(h), /*lean=*/h->pipelining ? 1 : 0);
    }
    lap.mark("head.1_launch_votes");
    uint32_t head;
    rc = run_tree(h, h->d_direct.as<uint64_t>(), h->d_totals.as<VoteTotals>(), 1, &head);
    lap.mark("head.2_tree_wait");
    if (rc) return rc;
    memcpy(out_root, h->blocks[head].root.data(), 32);
    return PE_OK;
}

// get_head whose root arrives with the pipeline's other outputs: votes + tree (and, in a streaming pipeline, the step's
// G1 sums behind them) are enqueued, nothing is polled.  The caller's loop then never blocks on the GPU inside a step.
int pe_get_head_async(pe_engine* h, uint8_t out_root[32])
{
    int rc = need_init(h, /*flush=*/false, /*keep_held=*/true);
    if (rc) return rc;
    if (!out_root) return PE_ERR_INVALID_ARG;
    if (!h->pipelining) return pe_get_head(h, out_root);  // outside a pipeline every call is synchronous
    // launches held back by an EARLIER pipeline, or a head already held for this one: out first, in order
    if (h->held.active && (h->held.arena != h->cur || h->held.have_head)) PE_TRY(held_issue(h));
    HostLap lap(&h->trace);
    PE_TRY(refresh_tree(h));
    {
        uint32_t tmp;
        if (!find_block(h, h->justified.root, &tmp))
            return fail(h, PE_ERR_UNKNOWN_ROOT, "justified checkpoint root is not in the store");
    }
    Stage st(h);
    OutBlock ob(h);
    const size_t off = ob.alloc(64);
    PE_TRY(ob.ensure());
    if (hold_eligible(h) && h->n_val) {
        // a streaming step: votes + tree (and, behind them, the step's G1 sums) go out with the NEXT aggregate's row kernels
        // (engine_pair.cpp) -- the store's scalars are read now, the vote tables when the kernels run, behind this step's
        // LMD update either way
        PE_TRY(tree_args(h, h->d_direct.as<uint64_t>(), h->d_totals.as<VoteTotals>(), 1, ob.host<uint32_t>(off), &h->held.tree));
        h->held.votes = votes_args(h);
        h->held.have_head = true;
        h->held.active = true;
        h->held.arena = h->cur;
    } else {
        if (h->held.active) PE_TRY(held_issue(h));
        {
            ProfScope ps(h, PE_KERNEL_VOTES);
            launch_votes(h->stream, votes_args(h), /*lean=*/1);
        }
        uint32_t unused;
        PE_TRY(run_tree(h, h->d_direct.as<uint64_t>(), h->d_totals.as<VoteTotals>(), 1, &unused, ob.host<uint32_t>(off)));
    }
    lap.mark("head.async_launch");
    const size_t base = ob.base;
    const int ai = h->cur;
    // the block table may grow before the completion runs; indices are stable (blocks are only appended)
    auto complete = [h, ai, base, off, out_root]() -> int {
        const uint32_t idx = *reinterpret_cast<const uint32_t*>(h->arena[ai].h_pin.as<uint8_t>() + base + off);
        if (idx >= h->blocks.size()) return fail(h, PE_ERR_NO_DEVICE, "tree kernel returned an invalid head index");
        memcpy(out_root, h->blocks[idx].root.data(), 32);
        return PE_OK;
    };
    return finish_call(h, st, ob, complete);
}

int pe_get_weights(pe_engine* h, uint64_t* out_weights, uint32_t n)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!out_weights || n != h->blocks.size()) return fail(h, PE_ERR_INVALID_ARG, "n must equal pe_num_blocks");
    uint8_t root[32];
    rc = pe_get_head(h, root);  // recompute, then read the per-block weights it left behind
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // get_head returns on the polled head word, not on kernel completion
    HIP_TRY(h, hipMemcpy(out_weights, h->d_weights.p, 8ull * n, hipMemcpyDeviceToHost));
    return PE_OK;
}

int pe_get_last_weights(pe_engine* h, uint64_t* out_weights, uint32_t n)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!out_weights || n != h->blocks.size() || !h->d_weights.p) return fail(h, PE_ERR_INVALID_ARG, "n must equal pe_num_blocks");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out_weights, h->d_weights.p, 8ull * n, hipMemcpyDeviceToHost));
    return PE_OK;
}

static int votes_partial_impl(pe_engine* h, void* dev_buf_u64, uint32_t n_blocks)
{
    if (!dev_buf_u64 || n_blocks != h->blocks.size()) return fail(h, PE_ERR_INVALID_ARG, "n_blocks mismatch");
    int rc = refresh_tree(h);
    if (rc) return rc;
    uint64_t* buf = static_cast<uint64_t*>(dev_buf_u64);
    {
        ProfScope ps(h, PE_KERNEL_VOTES);
        launch_votes(h->stream, h->d_vote_block.as<uint32_t>(), h->d_balance.as<uint64_t>(), h->d_flags.as<uint8_t>(),
                     h->n_val, h->cfg.filter_slashed, h->d_tpos.as<uint32_t>(), n_blocks, buf,
                     reinterpret_cast<VoteTotals*>(buf + n_blocks), 1, expiry_slots_ptr(h), min_vote_slot(h));
    }
    HIP_TRY(h, hipGetLastError());
    return PE_OK;
}

static int head_from_weights_impl(pe_engine* h, const void* dev_buf_u64, uint32_t n_blocks, uint8_t out_root[32])
{
    if (!dev_buf_u64 || !out_root || n_blocks != h->blocks.size())
        return fail(h, PE_ERR_INVALID_ARG, "n_blocks mismatch");
    int rc = refresh_tree(h);
    if (rc) return rc;
    uint64_t* buf = const_cast<uint64_t*>(static_cast<const uint64_t*>(dev_buf_u64));
    uint32_t head;
    rc = run_tree(h, buf, reinterpret_cast<const VoteTotals*>(buf + n_blocks), 0, &head);
    if (rc) return rc;
    memcpy(out_root, h->blocks[head].root.data(), 32);
    return PE_OK;
}

int pe_votes_partial(pe_engine* h, void* dev_buf_u64, uint32_t n_blocks)
{
    int rc = need_init(h);
    if (rc) return rc;
    return votes_partial_impl(h, dev_buf_u64, n_blocks);
}

int pe_head_from_weights(pe_engine* h, const void* dev_buf_u64, uint32_t n_blocks, uint8_t out_root[32])
{
    int rc = need_init(h);
    if (rc) return rc;
    return head_from_weights_impl(h, dev_buf_u64, n_blocks, out_root);
}

// ---------------------------------------------------------------- inspection
uint32_t pe_num_blocks(const pe_engine* h) { return h ? (uint32_t)h->blocks.size() : 0; }
uint64_t pe_num_validators(const pe_engine* h) { return h ? h->n_val : 0; }
int pe_block_root_at(const pe_engine* h, uint32_t i, uint8_t out_root[32])
{
    if (!h || !out_root || i >= h->blocks.size()) return PE_ERR_INVALID_ARG;
    memcpy(out_root, h->blocks[i].root.data(), 32);
    return PE_OK;
}
int pe_block_index_of(const pe_engine* h, const uint8_t root[32], uint32_t* out_index)
{
    if (!h || !root || !out_index) return PE_ERR_INVALID_ARG;
    uint32_t i;
    if (!find_block(h, to_root(root), &i)) return PE_ERR_UNKNOWN_ROOT;
    *out_index = i;
    return PE_OK;
}
int pe_get_latest_messages(pe_engine* h, uint64_t* out_epoch, uint32_t* out_block_index, uint64_t n)
{
    if (!h || !out_epoch || !out_block_index || n != h->n_val) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n == 0) return PE_OK;
    std::vector<uint64_t> key(n);
    HIP_TRY(h, hipMemcpyAsync(key.data(), h->d_vote_key.p, 8 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_block_index, h->d_vote_block.p, 4 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (uint64_t i = 0; i < n; ++i) {
        if (key[i] == 0) { out_epoch[i] = 0; out_block_index[i] = NONE32; }
        else out_epoch[i] = (key[i] >> 32) - 1;
    }
    return PE_OK;
}
// ---- checkpoint / resume: the store is a handful of flat arrays; these export and re-import them ----
int pe_get_block(const pe_engine* h, uint32_t i, uint8_t root[32], uint32_t* parent_index, uint64_t* slot,
                 uint64_t* pj_epoch, uint8_t pj_root[32], uint64_t* pf_epoch, uint8_t pf_root[32])
{
    if (!h || i >= h->blocks.size()) return PE_ERR_INVALID_ARG;
    const Block& b = h->blocks[i];
    if (root) memcpy(root, b.root.data(), 32);
    if (parent_index) *parent_index = b.parent;
    if (slot) *slot = b.slot;
    if (pj_epoch) *pj_epoch = b.post_justified.epoch;
    if (pj_root) memcpy(pj_root, b.post_justified.root.data(), 32);
    if (pf_epoch) *pf_epoch = b.post_finalized.epoch;
    if (pf_root) memcpy(pf_root, b.post_finalized.root.data(), 32);
    return PE_OK;
}
int pe_get_validator_flags(const pe_engine* h, uint8_t* out_flags, uint64_t n)
{
    if (!h || !out_flags || n != h->n_val) return PE_ERR_INVALID_ARG;
    memcpy(out_flags, h->h_flags.data(), n);
    return PE_OK;
}
int pe_get_latest_message_slots(pe_engine* h, uint32_t* out_slot, uint64_t n)
{
    if (!h || !out_slot || n != h->n_val) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (!h->cfg.vote_expiry_slots) { memset(out_slot, 0, 4 * n); return PE_OK; }
    if (n) HIP_TRY(h, hipMemcpyAsync(out_slot, h->d_vote_slot.p, 4 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}
int pe_set_latest_messages(pe_engine* h, uint64_t n, const uint64_t* epoch, const uint32_t* block_index,
                           const uint32_t* slot)
{
    int rc = need_init(h);
    if (rc) return rc;
    if (n != h->n_val || (n && (!epoch || !block_index))) return fail(h, PE_ERR_INVALID_ARG, "n must equal the registry size");
    (void)hipSetDevice(h->device);
    if (n == 0) return PE_OK;
    std::vector<uint64_t> key(n);
    std::vector<uint32_t> blk(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (block_index[i] == NONE32) { key[i] = 0; blk[i] = NONE32; continue; }
        if (block_index[i] >= h->blocks.size()) return fail(h, PE_ERR_INVALID_ARG, "latest message names an unknown block");
        if (epoch[i] >= 0xFFFFFFFEull) return fail(h, PE_ERR_CAPACITY, "target epoch does not fit 32 bits");
        key[i] = ((epoch[i] + 1) << 32) | 0xFFFFFFFFull;  // settled vote (see k_lmd)
        blk[i] = block_index[i];
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_vote_key.p, key.data(), 8 * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_vote_block.p, blk.data(), 4 * n, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.vote_expiry_slots) {
        if (slot) HIP_TRY(h, hipMemcpyAsync(h->d_vote_slot.p, slot, 4 * n, hipMemcpyHostToDevice, h->stream));
        else HIP_TRY(h, hipMemsetAsync(h->d_vote_slot.p, 0, 4 * n, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // the host vectors die at scope exit
    return PE_OK;
}
int pe_set_best_justified(pe_engine* h, uint64_t epoch, const uint8_t root[32])
{
    int rc = need_init(h);
    if (rc) return rc;
    if (!root) return PE_ERR_INVALID_ARG;
    h->best_justified.epoch = epoch;
    h->best_justified.root = to_root(root);
    return PE_OK;
}
int pe_get_store_scalars(const pe_engine* h, uint64_t* time, uint64_t* genesis_time, uint64_t* je, uint8_t jr[32],
                         uint64_t* fe, uint8_t fr[32], uint64_t* be, uint8_t br[32], uint8_t boost[32])
{
    if (!h) return PE_ERR_INVALID_ARG;
    if (time) *time = h->time;
    if (genesis_time) *genesis_time = h->genesis_time;
    if (je) *je = h->justified.epoch;
    if (jr) memcpy(jr, h->justified.root.data(), 32);
    if (fe) *fe = h->finalized.epoch;
    if (fr) memcpy(fr, h->finalized.root.data(), 32);
    if (be) *be = h->best_justified.epoch;
    if (br) memcpy(br, h->best_justified.root.data(), 32);
    if (boost) memcpy(boost, h->boost_root.data(), 32);
    return PE_OK;
}

}  // extern "C"
