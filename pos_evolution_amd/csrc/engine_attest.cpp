// engine_attest.cpp -- on_attestation (pe:963-979, 1423-1428), process_attestation (pe:722-754) and aggregation
// (validator guide A.8; pe:474/659/715/1536) over attestation rows in HOST memory: validate_on_attestation (A.4) and
// the grouping run here, the bit / point / table work on the device.  engine_resident.cpp is the same path with the
// rows resident in device memory.
#include "engine_internal.h"

using namespace posevo;

namespace posevo {

// validate_on_attestation (A.4) + committee resolution for on_attestation (pe:970-976).
int32_t validate_for_fork_choice(pe_engine* h, const pe_attestation& a, Resolved* out, BatchMemo* memo)
{
    const bool from_block = (a.flags & PE_ATT_FLAG_FROM_BLOCK) != 0;
    const uint64_t cur_slot = current_slot(h);
    if (!from_block) {  // validate_target_epoch_against_current_time
        const uint64_t cur_epoch = epoch_at_slot(h, cur_slot);
        const uint64_t prev_epoch = cur_epoch > 0 ? cur_epoch - 1 : 0;
        if (a.target_epoch != cur_epoch && a.target_epoch != prev_epoch)
            return PE_ATT_TARGET_EPOCH_NOT_CURRENT_OR_PREVIOUS;
    }
    if (a.target_epoch != epoch_at_slot(h, a.slot)) return PE_ATT_TARGET_EPOCH_SLOT_MISMATCH;
    uint32_t tgt_idx, blk_idx;
    if (!memo->find(h, 1, a.target_root, &tgt_idx)) return PE_ATT_UNKNOWN_TARGET_ROOT;
    if (!memo->find(h, 0, a.beacon_block_root, &blk_idx)) return PE_ATT_UNKNOWN_BEACON_BLOCK_ROOT;
    if (h->blocks[blk_idx].slot > a.slot) return PE_ATT_BLOCK_AFTER_ATTESTATION_SLOT;
    if (memo->ancestor(h, blk_idx, start_slot(h, a.target_epoch)) != tgt_idx) return PE_ATT_TARGET_NOT_ANCESTOR;
    if (cur_slot < a.slot + 1) return PE_ATT_SLOT_NOT_IN_PAST;
    // get_indexed_attestation -> get_beacon_committee(target_state, slot, index) (A.6)
    CommitteeTable* t = find_table(h, a.target_epoch);
    if (!t) return PE_ATT_NO_COMMITTEE_TABLE;
    const uint64_t cps = t->n_committees / h->cfg.slots_per_epoch;
    const uint64_t pos = (a.slot % h->cfg.slots_per_epoch) * cps + a.index;
    if (pos >= t->n_committees) return PE_ATT_COMMITTEE_INDEX_OUT_OF_RANGE;
    const uint32_t size = t->offsets[pos + 1] - t->offsets[pos];
    if (a.n_bits < size) return PE_ATT_BITS_LENGTH_MISMATCH;  // bits[i] would raise for i >= len(bits)
    out->table = t;
    out->pos = (uint32_t)pos;
    out->size = size;
    out->block_idx = blk_idx;
    return PE_ATT_OK;
}

}  // namespace posevo

// The handlers and pe_get_indexed_attestations re-pack the caller's bits on the host: device memory would fault there.
static bool bits_on_device(const uint8_t* bits_arena)
{
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, bits_arena) == hipSuccess) return pa.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return false;
}

// ---------------------------------------------------------------- on_attestation
// Rows handed over resident (bits_arena == PE_BITS_RESIDENT): which group of the last pe_aggregate is this row?
static inline uint32_t att_data_tag(const pe_attestation& a)
{
    uint32_t r0, r1;
    memcpy(&r0, a.beacon_block_root, 4);
    memcpy(&r1, a.target_root, 4);
    return (uint32_t)a.slot * 0x9E3779B1u ^ (uint32_t)a.index * 0x85EBCA6Bu ^ (uint32_t)a.target_epoch * 0xC2B2AE35u ^ r0 ^
           (r1 << 1);
}
static bool find_resident(const pe_engine* h, const pe_attestation& a, uint32_t* g_out, uint32_t guess)
{
    if (!h->res_valid) return false;
    const auto& rg = h->res_groups;
    const uint32_t tag = att_data_tag(a);
    if (guess < rg.size() && rg[guess].byte_off == a.bits_offset && rg[guess].n_bits == a.n_bits && rg[guess].tag == tag) {
        *g_out = guess;  // rows in group order
        return true;
    }
    size_t lo = 0, hi = rg.size();
    while (lo < hi) {  // byte_off is strictly increasing over the groups with bits; empty bitfields share an offset
        const size_t mid = (lo + hi) / 2;
        if (rg[mid].byte_off < a.bits_offset) lo = mid + 1; else hi = mid;
    }
    for (; lo < rg.size() && rg[lo].byte_off == a.bits_offset; ++lo)
        if (rg[lo].n_bits == a.n_bits && rg[lo].tag == tag) { *g_out = (uint32_t)lo; return true; }
    return false;
}


extern "C" {

int pe_on_attestation_batch(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                            uint64_t arena_len, int32_t* status, uint8_t* out_aggpk96, uint32_t* out_count)
{
    int rc = need_init(h, /*flush=*/false, /*keep_held=*/atts == PE_ROWS_RESIDENT);  // (that path holds or issues itself)
    if (rc) return rc;
    if (n && (!atts || !bits_arena || !status)) return PE_ERR_INVALID_ARG;
    if (atts == PE_ROWS_RESIDENT) {  // the groups of the last pe_aggregate over rows in device memory, validated there
        if (bits_arena != PE_BITS_RESIDENT || out_aggpk96)
            return fail(h, PE_ERR_INVALID_ARG, "PE_ROWS_RESIDENT goes with PE_BITS_RESIDENT; the aggregate pubkeys are pe_aggregate's");
        return on_attestation_resident(h, n, status, out_count);
    }
    if (out_aggpk96 && !h->have_points) return fail(h, PE_ERR_STATE, "aggregate pubkeys requested but no pubkeys loaded");
    if (n == 0) return PE_OK;
    const bool resident = bits_arena == PE_BITS_RESIDENT;
    if (resident && !h->res_valid) return fail(h, PE_ERR_STATE, "PE_BITS_RESIDENT: no pe_aggregate result is resident");
    if (!resident && bits_on_device(bits_arena))
        return fail(h, PE_ERR_INVALID_ARG, "bits in device memory: hand them over through pe_aggregate + PE_BITS_RESIDENT");
    HostLap lap(&h->trace);
    // ---- sizes first: the staging block must not move once pointers into it exist ----
    uint64_t word_bound = 0;
    std::vector<uint32_t> res_group(resident ? n : 0);
    for (uint32_t i = 0; i < n; ++i) {
        const pe_attestation& a = atts[i];
        if (a.target_epoch >= 0xFFFFFFFEull) return fail(h, PE_ERR_INVALID_ARG, "target epoch must fit 32 bits");
        if (resident) {
            if (!find_resident(h, a, &res_group[i], i))
                return fail(h, PE_ERR_INVALID_ARG, "PE_BITS_RESIDENT: row is not a row of the last pe_aggregate");
            continue;
        }
        if (a.n_bits > 0x7FFFFFFFu || (uint64_t)a.bits_offset + ((uint64_t)a.n_bits + 7) / 8 > arena_len)
            return fail(h, PE_ERR_INVALID_ARG, "attestation bits exceed the arena");
        word_bound += ((uint64_t)a.n_bits + 31) / 32 + 1;
    }
    lap.mark("att.1a_sizes_resident");
    Stage st(h);
    size_t csr_bound = 4ull * n + 1024;
    for (auto& t : h->tables) csr_bound += 4ull * (t.n_committees + 1) + 512;
    PE_TRY(st.reserve(word_bound * 4 + (sizeof(AttRow) + sizeof(G1Group)) * (size_t)n + csr_bound + 4096));
    const size_t off_words = st.alloc(word_bound * 4 + 4);
    const size_t off_rows = st.alloc(sizeof(AttRow) * (size_t)n);
    const size_t off_groups = st.alloc(sizeof(G1Group) * (size_t)n);
    uint32_t* words = st.host<uint32_t>(off_words);
    AttRow* rows = st.host<AttRow>(off_rows);
    lap.mark("att.1b_reserve");
    // ---- validate everything (validation reads only time/blocks/tables, never latest_messages) ----
    std::vector<Resolved> res(n);
    auto row_src_p = std::make_shared<std::vector<uint32_t>>();  // accepted row -> attestation index
    std::vector<uint32_t>& row_src = *row_src_p;
    row_src.reserve(n);
    uint32_t n_words = 0, n_rows = 0;
    CommitteeTable* first_table = nullptr;
    bool multi_table = false;
    BatchMemo memo;
    for (uint32_t i = 0; i < n; ++i) {
        const pe_attestation& a = atts[i];
        int32_t stt = validate_for_fork_choice(h, a, &res[i], &memo);
        uint32_t cnt = 0;
        if (stt == PE_ATT_OK) {
            const uint32_t use = res[i].size;  // bits beyond the committee length are never read (A.6)
            uint32_t bits_word;
            if (resident) {
                // the OR-ed bits are on the device: emptiness (and member overlap) is settled there -- an empty or
                // gated row changes nothing -- and reported when the call completes
                if (a.n_bits != use) stt = PE_ATT_BITS_LENGTH_MISMATCH;
                bits_word = h->res_groups[res_group[i]].word;
                cnt = 1;
            } else {
                cnt = use ? pack_bits(bits_arena + a.bits_offset, use, words + n_words) : 0;
                bits_word = n_words;
            }
            // is_valid_indexed_attestation (A.7): non-empty sorted-unique indices, then the signature verdict
            if (stt == PE_ATT_OK) {
                if (cnt == 0) stt = PE_ATT_EMPTY_OR_INVALID_INDICES;
                else if (!(a.flags & PE_ATT_FLAG_SIGNATURE_VALID)) stt = PE_ATT_BAD_SIGNATURE;
            }
            if (stt == PE_ATT_OK) {
                AttRow& r = rows[n_rows];
                r.member_base = res[i].table->offsets[res[i].pos];
                r.n_bits = use;
                r.bits_word = bits_word;
                r.block_idx = res[i].block_idx;
                r.epoch_p1 = (uint32_t)a.target_epoch + 1;
                r.order = n_rows;
                r.flag_mask = 0;
                r.which = 0;
                r.slot = (uint32_t)a.slot;
                r.gate = resident ? 2 * res_group[i] + 1 : NONE32;
                if (!resident) n_words += (use + 31) / 32;
                row_src.push_back(i);
                ++n_rows;
                if (first_table && first_table != res[i].table) multi_table = true;
                if (!first_table) first_table = res[i].table;
            }
        }
        status[i] = stt;
        if (out_count) out_count[i] = (stt == PE_ATT_OK && !resident) ? cnt : 0;
    }
    if (out_aggpk96)
        for (uint32_t i = 0; i < n; ++i) { memset(out_aggpk96 + 96ull * i, 0, 96); out_aggpk96[96ull * i] = 0x40; }
    lap.mark("att.1c_validate");
    if (n_rows == 0) return PE_OK;
    // Rows of different target epochs index different member arrays: make each table's rows contiguous (stable, so
    // the batch order inside a table is kept; `order` stays global).  Different tables = different epochs, where
    // the later epoch wins regardless of order, so per-table passes equal the sequential result.
    std::vector<std::pair<CommitteeTable*, std::pair<uint32_t, uint32_t>>> segs;  // table, [begin, end)
    if (multi_table) {
        std::vector<uint32_t> perm(n_rows);
        std::iota(perm.begin(), perm.end(), 0u);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) {
            return res[row_src[x]].table < res[row_src[y]].table;
        });
        std::vector<AttRow> tmp(rows, rows + n_rows);
        std::vector<uint32_t> src2(n_rows);
        for (uint32_t k = 0; k < n_rows; ++k) { rows[k] = tmp[perm[k]]; src2[k] = row_src[perm[k]]; }
        row_src.swap(src2);
    }
    for (uint32_t k = 0; k < n_rows;) {
        uint32_t e = k + 1;
        while (e < n_rows && res[row_src[e]].table == res[row_src[k]].table) ++e;
        segs.push_back({res[row_src[k]].table, {k, e}});
        k = e;
    }
    // aggregate pubkeys: one G1 plan over all accepted rows (groups in row order); members differ per table, so
    // one launch per segment over its slice of the descriptors
    OutBlock ob(h);
    size_t off_out96 = 0;
    std::vector<G1Plan> plans(segs.size());
    if (out_aggpk96) {
        off_out96 = ob.alloc(96ull * n_rows);
        PE_TRY(ob.ensure());
        G1Group* groups = st.host<G1Group>(off_groups);
        for (size_t sg = 0; sg < segs.size(); ++sg) {
            const uint32_t b0 = segs[sg].second.first, e0 = segs[sg].second.second;
            plan_g1(e0 - b0, [&](uint32_t g) { return rows[b0 + g].n_bits; }, groups + b0, &plans[sg]);
            for (uint32_t k = b0; k < e0; ++k) {
                groups[k].member_start = rows[k].member_base;
                groups[k].bits_word = rows[k].bits_word;
            }
        }
    }
    // Large batches on a partition table: validator-major LMD pass (streams the V-sized tables once, no atomics).
    // Small ones: committee-major with atomics (touches only the attesting validators).
    std::vector<size_t> seg_vm(segs.size(), (size_t)-1), seg_vm_list(segs.size(), 0);
    for (size_t sg = 0; sg < segs.size(); ++sg) {
        CommitteeTable* t = segs[sg].first;
        const uint32_t b0 = segs[sg].second.first, e0 = segs[sg].second.second;
        uint64_t bits_total = 0;
        for (uint32_t k = b0; k < e0; ++k) bits_total += rows[k].n_bits;
        if (!t->is_partition || t->n_val_at_load != h->n_val || !t->d_inv_comm.p || bits_total * 8 < h->n_val) continue;
        const uint32_t nc = t->n_committees;
        const size_t off_cs = st.alloc(4ull * (nc + 1));
        const size_t off_cl = st.alloc(4ull * (e0 - b0));
        if (st.overflow()) return fail(h, PE_ERR_OOM, "staging block overflow");
        uint32_t* cs = st.host<uint32_t>(off_cs);
        uint32_t* cl = st.host<uint32_t>(off_cl);
        memset(cs, 0, 4ull * (nc + 1));
        for (uint32_t k = b0; k < e0; ++k) cs[res[row_src[k]].pos + 1] += 1;
        for (uint32_t c = 0; c < nc; ++c) cs[c + 1] += cs[c];
        std::vector<uint32_t> cur(cs, cs + nc);
        for (uint32_t k = b0; k < e0; ++k) cl[cur[res[row_src[k]].pos]++] = k - b0;  // batch order kept
        seg_vm[sg] = off_cs;
        seg_vm_list[sg] = off_cl;
    }
    lap.mark("att.1d_segments_csr");
    const uint32_t* d_bits = resident ? h->arena[h->res_arena].d_res_bits.as<uint32_t>() : st.dev<uint32_t>(off_words);
    const uint32_t* d_gates = resident ? h->arena[h->res_arena].d_res_info.as<uint32_t>() : nullptr;
    HIP_TRY(h, st.upload());
    for (size_t sg = 0; sg < segs.size(); ++sg) {
        CommitteeTable* t = segs[sg].first;
        const uint32_t b0 = segs[sg].second.first, e0 = segs[sg].second.second;
        {
            ProfScope ps(h, PE_KERNEL_LMD);
            if (seg_vm[sg] != (size_t)-1)
                launch_lmd_validator_major(h->stream, st.dev<AttRow>(off_rows) + b0, st.dev<uint32_t>(seg_vm[sg]),
                                           st.dev<uint32_t>(seg_vm_list[sg]), t->d_inv_comm.as<uint32_t>(),
                                           t->d_inv_pos.as<uint32_t>(), d_bits,
                                           h->d_flags.as<uint8_t>(), h->n_val, h->d_vote_key.as<uint64_t>(),
                                           h->d_vote_block.as<uint32_t>(),
                                           h->cfg.vote_expiry_slots ? h->d_vote_slot.as<uint32_t>() : nullptr, d_gates);
            else
                launch_lmd_update(h->stream, st.dev<AttRow>(off_rows) + b0, e0 - b0, t->d_members.as<uint32_t>(),
                                  d_bits, h->d_flags.as<uint8_t>(), h->d_vote_key.as<uint64_t>(),
                                  h->d_vote_block.as<uint32_t>(),
                                  h->cfg.vote_expiry_slots ? h->d_vote_slot.as<uint32_t>() : nullptr, d_gates);
        }
        if (out_aggpk96) {
            g1_stream_guard(h, h->stream);
            rc = launch_g1_planned(h, h->d_points.as<uint32_t>(), t->d_members.as<uint32_t>(), d_bits,
                                   st.dev<G1Group>(off_groups) + b0, plans[sg],
                                   ob.host<uint8_t>(off_out96) + 96ull * b0, nullptr);
            if (rc) return rc;
        }
        t->stamp = ++h->table_stamp;
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("att.2_launch");
    const size_t ob_base = ob.base;
    auto res_group_p = std::make_shared<std::vector<uint32_t>>(std::move(res_group));
    std::shared_ptr<std::vector<uint32_t>> info_p = resident ? h->res_info_host : nullptr;
    const int ai = h->cur;
    auto complete = [h, row_src_p, res_group_p, n_rows, resident, status, out_aggpk96, out_count, ob_base, off_out96,
                     info_p, ai]() -> int {
        const std::vector<uint32_t>& src = *row_src_p;
        if (out_aggpk96)
            for (uint32_t k = 0; k < n_rows; ++k)
                memcpy(out_aggpk96 + 96ull * src[k], h->arena[ai].h_pin.as<uint8_t>() + ob_base + off_out96 + 96ull * k, 96);
        if (resident) {  // emptiness / overlap of the resident unions, now that the aggregate's counts are here
            const std::vector<uint32_t>& info = *info_p;
            for (uint32_t k = 0; k < n_rows; ++k) {
                const uint32_t i = src[k], g = (*res_group_p)[i];
                if (2 * (size_t)g + 1 >= info.size()) return fail(h, PE_ERR_STATE, "resident aggregate did not complete");
                const uint32_t cnt = info[2 * g], overlap = info[2 * g + 1];
                if (overlap) status[i] = PE_ATT_BAD_SIGNATURE;
                else if (cnt == 0) status[i] = PE_ATT_EMPTY_OR_INVALID_INDICES;
                if (out_count) out_count[i] = status[i] == PE_ATT_OK ? cnt : 0;
            }
        }
        return PE_OK;
    };
    HostLap lap2(&h->trace);
    rc = finish_call(h, st, ob, complete);
    lap2.mark("att.3_wait_outputs");
    return rc;
}

int pe_get_indexed_attestations(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                                uint64_t arena_len, int32_t* status, uint32_t* out_offsets, uint32_t* out_indices,
                                uint64_t out_indices_cap)
{
    if (!h || !out_offsets || (n && (!atts || !bits_arena || !status))) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    out_offsets[0] = 0;
    if (n == 0) return PE_OK;
    if (bits_on_device(bits_arena)) return fail(h, PE_ERR_INVALID_ARG, "bits in device memory: only pe_aggregate reads them there");
    uint64_t word_bound = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (atts[i].n_bits > 0x7FFFFFFFu || (uint64_t)atts[i].bits_offset + ((uint64_t)atts[i].n_bits + 7) / 8 > arena_len)
            return fail(h, PE_ERR_INVALID_ARG, "attestation bits exceed the arena");
        word_bound += ((uint64_t)atts[i].n_bits + 31) / 32 + 1;
    }
    Stage st(h);
    PE_TRY(st.reserve(word_bound * 4 + (sizeof(AttRow) + 4) * (size_t)(n + 1) + 4096));
    const size_t off_words = st.alloc(word_bound * 4);
    const size_t off_rows = st.alloc(sizeof(AttRow) * (size_t)n);
    const size_t off_offs = st.alloc(4ull * (n + 1));
    uint32_t* words = st.host<uint32_t>(off_words);
    AttRow* rows = st.host<AttRow>(off_rows);
    uint32_t* offs = st.host<uint32_t>(off_offs);
    std::vector<CommitteeTable*> row_table;
    std::vector<uint32_t> row_src;
    uint32_t n_words = 0, n_rows = 0;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const pe_attestation& a = atts[i];
        int32_t s = PE_ATT_OK;
        CommitteeTable* t = find_table(h, a.target_epoch);
        uint32_t cnt = 0;
        if (!t) s = PE_ATT_NO_COMMITTEE_TABLE;
        else {
            const uint64_t cps = t->n_committees / h->cfg.slots_per_epoch;
            const uint64_t pos = (a.slot % h->cfg.slots_per_epoch) * cps + a.index;
            if (pos >= t->n_committees) s = PE_ATT_COMMITTEE_INDEX_OUT_OF_RANGE;
            else {
                const uint32_t size = t->offsets[pos + 1] - t->offsets[pos];
                if (a.n_bits < size) s = PE_ATT_BITS_LENGTH_MISMATCH;
                else if (size > 8192) return fail(h, PE_ERR_CAPACITY, "committee larger than 8192 members");
                else {
                    cnt = size ? pack_bits(bits_arena + a.bits_offset, size, words + n_words) : 0;
                    AttRow& r = rows[n_rows];
                    r.member_base = t->offsets[pos];
                    r.n_bits = size;
                    r.bits_word = n_words;
                    r.block_idx = r.epoch_p1 = r.order = r.flag_mask = r.which = r.slot = 0;
                    r.gate = NONE32;
                    offs[n_rows] = (uint32_t)total;
                    n_words += (size + 31) / 32;
                    row_table.push_back(t);
                    row_src.push_back(i);
                    ++n_rows;
                }
            }
        }
        status[i] = s;
        out_offsets[i] = (uint32_t)total;
        total += cnt;
        if (total > 0xFFFFFFFFull) return fail(h, PE_ERR_CAPACITY, "more than 2^32 attesting indices in one call");
    }
    out_offsets[n] = (uint32_t)total;
    if (total > out_indices_cap) return fail(h, PE_ERR_CAPACITY, "out_indices too small");
    if (n_rows == 0 || total == 0) return PE_OK;
    if (!out_indices) return PE_ERR_INVALID_ARG;
    OutBlock ob(h);
    const size_t off_idx = ob.alloc(4ull * total);
    PE_TRY(ob.ensure());
    HIP_TRY(h, st.upload());
    for (uint32_t k = 0; k < n_rows;) {  // one launch per run of rows sharing a table (members array)
        uint32_t e = k + 1;
        while (e < n_rows && row_table[e] == row_table[k]) ++e;
        launch_indexed_attestations(h->stream, st.dev<AttRow>(off_rows) + k, e - k,
                                    row_table[k]->d_members.as<uint32_t>(), st.dev<uint32_t>(off_words),
                                    st.dev<uint32_t>(off_offs) + k, ob.dev<uint32_t>(off_idx));
        k = e;
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, ob.download());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(out_indices, ob.host<uint32_t>(off_idx), 4ull * total);
    return PE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- aggregation
struct AggState {  // what the completion of one pe_aggregate needs after the wait
    std::vector<uint32_t> rep, gstart, order, gof, out_byte_off, out_word;
    std::vector<uint32_t> all_valid;
    const pe_attestation* atts = nullptr;
    pe_attestation* out_atts = nullptr;
    uint8_t *out_bits_arena = nullptr, *out_sig96 = nullptr, *out_aggpk96 = nullptr;
    uint32_t* out_count = nullptr;
    uint32_t ng = 0;
    size_t base = 0, off_obits = 0, off_oinfo = 0, off_opk = 0, off_osig = 0;
    size_t off_opkx = 0;      // aggregate pubkeys of the groups of further committee tables (compacted)
    size_t packed_bytes = 0;  // > 0: the word-aligned unions in the pinned block ARE the caller's byte-packed layout
    int tune_arm = -1;
};

namespace posevo {
int aggregate_impl(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                   uint64_t arena_len, const uint8_t* sig_points96, pe_attestation* out_atts,
                   uint32_t* out_n_groups, uint32_t* group_of, uint8_t* out_bits_arena, uint64_t out_arena_cap,
                   uint8_t* out_sig96, uint8_t* out_aggpk96, uint32_t* out_count, void* dev_partials,
                   uint32_t dev_partials_capacity, bool partials_may_defer)
{
    if (!h) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    if (!out_n_groups || (n && (!atts || !bits_arena || !out_atts || !out_bits_arena))) return PE_ERR_INVALID_ARG;
    if (out_sig96 && !sig_points96) return fail(h, PE_ERR_INVALID_ARG, "out_sig96 requires sig_points96");
    const bool want_pk = out_aggpk96 || dev_partials;
    if (want_pk && !h->have_points) return fail(h, PE_ERR_STATE, "aggregate pubkeys requested but no pubkeys loaded");
    *out_n_groups = 0;
    if (n == 0) return PE_OK;
    if (rows_on_device(atts)) {  // rows the host cannot read: grouped, resolved and validated on the device
        if (sig_points96 || out_sig96 || dev_partials)
            return fail(h, PE_ERR_INVALID_ARG, "rows in device memory: signature points and sharded partials take host rows");
        return aggregate_resident(h, atts, n, bits_arena, arena_len, out_atts, out_n_groups, group_of, out_bits_arena,
                                  out_arena_cap, out_aggpk96, out_count);
    }
    if (h->held.active) PE_TRY(held_issue(h));  // host rows: nothing to pair the held-back launches with
    HostLap lap(&h->trace);
    auto stp = std::make_shared<AggState>();
    AggState& A = *stp;
    // ---- group by identical AttestationData + n_bits, in order of first appearance (flat open addressing) ----
    auto hash_att = [](const pe_attestation& a) {
        uint64_t hsh = a.slot * 0x9E3779B97F4A7C15ull ^ (a.index + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        uint64_t t;
        memcpy(&t, a.beacon_block_root, 8); hsh ^= t * 0x94D049BB133111EBull;
        memcpy(&t, a.target_root, 8); hsh ^= (t + a.target_epoch) * 0xD6E8FEB86659FD93ull;
        memcpy(&t, a.source_root, 8); hsh ^= (t + a.source_epoch) * 0xA24BAED4963EE407ull;
        hsh ^= a.n_bits;
        return hsh ^ (hsh >> 29);
    };
    uint32_t tab_size = 16;
    while (tab_size < 2 * n) tab_size <<= 1;
    std::vector<uint32_t> table(tab_size, NONE32);  // slot -> group id
    std::vector<uint32_t>& gof = A.gof;
    std::vector<uint32_t>& rep = A.rep;             // rep[g] = first attestation of group g
    std::vector<uint32_t> gcount;
    std::vector<uint32_t> boff(n);                  // bits_offset per row, compact: the later passes never re-read the rows
    std::vector<uint32_t>& gvalid = A.all_valid;    // AND of the members' signature verdicts
    gof.resize(n);
    uint64_t lo = ~0ull, hi = 0;                    // byte span of the arena this call reads
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t b0 = atts[i].bits_offset, b1 = b0 + ((uint64_t)atts[i].n_bits + 7) / 8;
        if (b1 > arena_len || atts[i].n_bits > 0x7FFFFFFFu) return fail(h, PE_ERR_INVALID_ARG, "attestation bits exceed the arena");
        if (atts[i].n_bits) { lo = std::min(lo, b0); hi = std::max(hi, b1); }
        uint32_t slot = (uint32_t)hash_att(atts[i]) & (tab_size - 1);
        uint32_t g;
        for (;;) {
            g = table[slot];
            if (g == NONE32) {
                g = (uint32_t)rep.size();
                table[slot] = g;
                rep.push_back(i);
                gcount.push_back(0);
                gvalid.push_back(PE_ATT_FLAG_SIGNATURE_VALID);
                break;
            }
            const pe_attestation& r = atts[rep[g]];
            if (r.n_bits == atts[i].n_bits && att_data_equal(r, atts[i])) break;
            slot = (slot + 1) & (tab_size - 1);
        }
        gof[i] = g;
        gcount[g] += 1;
        gvalid[g] &= atts[i].flags;
        boff[i] = atts[i].bits_offset;
    }
    if (lo > hi) lo = hi = 0;
    lo &= ~uint64_t(3);                              // keep the members' word alignment relative to the upload
    if (hi - lo >= 0xFFFFFFF0ull) return fail(h, PE_ERR_CAPACITY, "bit arena span exceeds 4 GiB");
    const uint32_t ng = (uint32_t)rep.size();
    if (dev_partials && ng > dev_partials_capacity)  // before anything is launched: the buffer is the caller's
        return fail(h, PE_ERR_CAPACITY, "dev_partials holds fewer groups than the batch forms");
    A.ng = ng;
    std::vector<uint32_t>& gstart = A.gstart;        // counting sort: members of group g, in input order
    std::vector<uint32_t>& order = A.order;
    gstart.assign(ng + 1, 0);
    order.resize(n);
    for (uint32_t g = 0; g < ng; ++g) gstart[g + 1] = gstart[g] + gcount[g];
    {
        std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
        for (uint32_t i = 0; i < n; ++i) order[cur[gof[i]]++] = i;
    }
    lap.mark("agg.1_group");
    // ---- resolve committees (aggregate pubkey) ----
    std::vector<Resolved> gres(want_pk ? ng : 0);
    CommitteeTable* table_pk = nullptr;
    auto xgroups_p = std::make_shared<std::vector<uint32_t>>();  // groups whose committee table is not the first one's
    std::vector<uint32_t>& xgroups = *xgroups_p;
    if (want_pk) {
        for (uint32_t g = 0; g < ng; ++g) {
            const pe_attestation& a = atts[rep[g]];
            CommitteeTable* t = find_table(h, a.target_epoch);
            if (!t) return fail(h, PE_ERR_NO_COMMITTEES, "no committee table for a group's target epoch");
            if (table_pk && t != table_pk) {
                // a batch around an epoch boundary: the groups of the first table go through the main launch, the
                // others through one (synchronous-stream) launch per further table -- see "further tables" below
                if (dev_partials)
                    return fail(h, PE_ERR_INVALID_ARG, "pe_aggregate_partial / _sharded: one target epoch per call");
                xgroups.push_back(g);
            } else {
                table_pk = t;
            }
            const uint64_t cps = t->n_committees / h->cfg.slots_per_epoch;
            if (a.index >= cps) return fail(h, PE_ERR_INVALID_ARG, "committee index out of range");
            const uint64_t pos = (a.slot % h->cfg.slots_per_epoch) * cps + a.index;
            const uint32_t size = t->offsets[pos + 1] - t->offsets[pos];
            if (a.n_bits != size) return fail(h, PE_ERR_INVALID_ARG, "len(aggregation_bits) != len(committee)");  // pe:730
            gres[g].table = t;
            gres[g].pos = (uint32_t)pos;
            gres[g].size = size;
        }
    }
    // ---- lay everything out in the staging block ----
    uint64_t out_words = 0, out_bytes = 0;
    for (uint32_t g = 0; g < ng; ++g) {
        out_words += (atts[rep[g]].n_bits + 31) / 32;
        out_bytes += (atts[rep[g]].n_bits + 7) / 8;
    }
    if (out_bytes > out_arena_cap) return fail(h, PE_ERR_CAPACITY, "output bit arena too small");
    const size_t span = (size_t)(hi - lo);
    Stage st(h);
    PE_TRY(st.reserve(span + 64 + sizeof(UnionGroup) * (size_t)ng + 4ull * n + 3 * sizeof(G1Group) * (size_t)ng +
                      4ull * n + 8192));
    const size_t off_arena = st.alloc(span + 16);
    const size_t off_ug = st.alloc(sizeof(UnionGroup) * (size_t)ng);
    const size_t off_ub = st.alloc(4ull * n);
    const size_t off_g1 = st.alloc(sizeof(G1Group) * (size_t)ng);
    const size_t off_g1s = st.alloc(sizeof(G1Group) * (size_t)ng);
    const size_t off_idx = st.alloc(4ull * n);
    const size_t off_g1x = xgroups.empty() ? 0 : st.alloc(sizeof(G1Group) * xgroups.size());
    // the caller's bits travel as they are (one copy into the pinned block): k_bits_union reads the members at
    // their byte offsets, masks the tail of the last word and never needs re-packed words
    lap.mark("agg.2a_resolve_reserve");
    // Where do the caller's bits live?  Pageable host memory is copied into the pinned block here; pinned host memory
    // and device memory are copied by the copy engine straight into the device block (no pass over them on the host) --
    // the caller then keeps them unchanged until the call's outputs are complete.
    int arena_kind = 0;  // 0 pageable host, 1 pinned host, 2 device
    if (span) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, bits_arena) == hipSuccess) {
            if (pa.type == hipMemoryTypeDevice) arena_kind = 2;
            else if (pa.type == hipMemoryTypeHost) arena_kind = 1;
        } else {
            (void)hipGetLastError();  // plain malloc'ed memory is "invalid value" to older runtimes: not an error here
        }
    }
    if (arena_kind == 0) memcpy(st.host<uint8_t>(off_arena), bits_arena + lo, span);
    memset(st.host<uint8_t>(off_arena) + span, 0, 16);
    lap.mark("agg.2b_memcpy_arena");
    UnionGroup* ug = st.host<UnionGroup>(off_ug);
    uint32_t* ubytes = st.host<uint32_t>(off_ub);
    A.out_byte_off.resize(ng);
    A.out_word.resize(ng);
    {
        uint32_t ow = 0, obytes = 0;
        for (uint32_t g = 0; g < ng; ++g) {
            ug[g].list_start = gstart[g];
            ug[g].n_atts = gcount[g];
            ug[g].n_bits = atts[rep[g]].n_bits;
            ug[g].out_word = ow;
            A.out_word[g] = ow;
            ow += (atts[rep[g]].n_bits + 31) / 32;
            A.out_byte_off[g] = obytes;
            obytes += (atts[rep[g]].n_bits + 7) / 8;
        }
        const uint32_t lo32 = (uint32_t)lo;          // bits_offset is 32 bits wide, so is every offset at or below it
        for (uint32_t k = 0; k < n; ++k) ubytes[k] = boff[order[k]] - lo32;
    }
    lap.mark("agg.2c_union_groups");
    // resident outputs: the OR-ed bits and {popcount, overlap} stay on the device for the calls that follow
    PE_TRY(ensure_quiesced(h, h->A().d_res_bits, out_words * 4 + 64));
    PE_TRY(ensure_quiesced(h, h->A().d_res_info, 8ull * ng + 64));
    OutBlock ob(h);
    const size_t off_obits = ob.alloc(out_words * 4 + 4);
    const size_t off_oinfo = ob.alloc(8ull * ng);
    const size_t off_opk = out_aggpk96 ? ob.alloc(96ull * ng) : 0;
    const size_t off_osig = out_sig96 ? ob.alloc(96ull * ng) : 0;
    const size_t off_opkx = (out_aggpk96 && !xgroups.empty()) ? ob.alloc(96ull * xgroups.size()) : 0;
    PE_TRY(ob.ensure());
    G1Plan plan_pk, plan_sig;
    int tune_arm = -1;  // >= 0: this call is an autotune trial of shape `tune_arm`
    if (want_pk) {
        G1Group* gr = st.host<G1Group>(off_g1);
        uint64_t total_pk = 0;
        for (uint32_t g = 0; g < ng; ++g) total_pk += gres[g].size;
        uint32_t target = G1_TARGET_LANES;
        if (total_pk >= (1ull << 19)) {
            // Streaming pipelines: ONE wave per SIMD (65 536 lanes, k = 16 at 1 M validators): the S29 accumulation loses 3 %
            // to its two-wave shape alone (tools/accbench) and leaves every SIMD the registers the guests of a streaming
            // step need -- k_g1_tree, k_g1_finish, k_att_plan, the fork-choice chain all run beside it.  Synchronous calls:
            // the shape the first four large calls measured faster.
            // (also at configs[4], where the accumulation is 0.89 of the 0.95 ms step: two waves per SIMD measured 1.00-1.05 ms
            // there -- the accumulation 0.92-0.99 ms instead of 0.89, k_g1_tree 0.75 ms waiting for registers)
            if (h->streaming) target = G1_TARGET_LANES / 2;
            else if (h->g1_target_slots) target = h->g1_target_slots;
            else {
                tune_arm = h->g1_tune_calls & 1;
                target = tune_arm ? G1_TARGET_LANES / 2 : G1_TARGET_LANES;
            }
        }
        plan_g1(ng, [&](uint32_t g) { return gres[g].table == table_pk ? gres[g].size : 0u; }, gr, &plan_pk, G1_WG, target);
        for (uint32_t g = 0; g < ng; ++g) {
            gr[g].member_start = gres[g].table == table_pk ? table_pk->offsets[gres[g].pos] : 0u;
            gr[g].bits_word = A.out_word[g];  // the OR-ed bits, device resident: no round trip
        }
    }
    // further tables: their groups (kept in group order, contiguous per table) get descriptor arrays of their own
    struct XSeg { CommitteeTable* table; uint32_t begin, end; G1Plan plan; };
    std::vector<XSeg> xsegs;
    if (!xgroups.empty()) {
        std::stable_sort(xgroups.begin(), xgroups.end(), [&](uint32_t x, uint32_t y) { return gres[x].table < gres[y].table; });
        G1Group* grx = st.host<G1Group>(off_g1x);
        for (uint32_t b = 0; b < xgroups.size();) {
            uint32_t e = b + 1;
            while (e < xgroups.size() && gres[xgroups[e]].table == gres[xgroups[b]].table) ++e;
            XSeg sg{gres[xgroups[b]].table, b, e, G1Plan()};
            plan_g1(e - b, [&](uint32_t k) { return gres[xgroups[b + k]].size; }, grx + b, &sg.plan);
            for (uint32_t k = b; k < e; ++k) {
                grx[k].member_start = sg.table->offsets[gres[xgroups[k]].pos];
                grx[k].bits_word = A.out_word[xgroups[k]];
            }
            xsegs.push_back(sg);
            b = e;
        }
    }
    if (out_sig96) {
        G1Group* gr = st.host<G1Group>(off_g1s);
        plan_g1(ng, [&](uint32_t g) { return gcount[g]; }, gr, &plan_sig);
        for (uint32_t g = 0; g < ng; ++g) gr[g].member_start = gstart[g];
        memcpy(st.host<uint32_t>(off_idx), order.data(), 4ull * n);  // points indexed by input attestation
    }
    lap.mark("agg.2d_ensure_plan");
    if (st.overflow()) return fail(h, PE_ERR_OOM, "staging block overflow");
    // the rows of the result are host data: complete at return, also inside a pipeline (bits / counts / sums follow)
    for (uint32_t g = 0; g < ng; ++g) {
        out_atts[g] = atts[rep[g]];
        out_atts[g].bits_offset = A.out_byte_off[g];
        out_atts[g].flags = (atts[rep[g]].flags & ~(uint32_t)PE_ATT_FLAG_SIGNATURE_VALID) | A.all_valid[g];
    }
    if (group_of) memcpy(group_of, gof.data(), 4ull * n);
    *out_n_groups = ng;
    h->res_groups.resize(ng);
    for (uint32_t g = 0; g < ng; ++g)
        h->res_groups[g] = {A.out_byte_off[g], atts[rep[g]].n_bits, A.out_word[g], A.all_valid[g], att_data_tag(atts[rep[g]])};
    auto info_p = std::make_shared<std::vector<uint32_t>>();
    // the resident hand-over is published at the end, when every enqueue below has succeeded: a failing launch or copy
    // must not leave res_valid pointing at unions that were never computed, nor a deferred launch at a staging region
    // whose cursor never advanced
    h->res_valid = false;
    const size_t deferred_before = h->deferred.size();
    struct Unwind {
        pe_engine* h; size_t keep; bool armed = true;
        ~Unwind() { if (armed) { h->res_valid = false; if (h->deferred.size() > keep) h->deferred.resize(keep); } }
    } unwind{h, deferred_before};
    lap.mark("agg.2e_out_rows");
    // ---- device ----
    hipStream_t ms = h->stream;
    // pipelined + host outputs: the G1 sums run on the side stream, beside the fork-choice kernels of the calls that
    // follow (they only need the union).  Sharded partials stay on the main stream, where the caller's collective is.
    // (partials for the engine's own exchange follow the same route; partials for a caller's collective never do)
    const bool on_side = want_pk && (!dev_partials || partials_may_defer) && h->pipelining && h->side_stream &&
                         h->stream == h->own_stream && xgroups.empty();
    h->last_agg_on_side = on_side;
    hipStream_t gs = on_side ? h->side_stream : ms;
    // A previous aggregate of THIS pipeline reads the arena's d_res_bits / d_res_info from its G1 launch.  In a streaming
    // pipeline that launch may still sit in h->deferred (nothing has read the unions yet): issue it now, so that the
    // ev_join wait below orders this call's k_bits_union -- which rewrites the arena's resident words from word 0 --
    // behind the earlier chain (without this the first aggregate's pubkeys were summed over the second's unions).
    if (!h->deferred.empty()) PE_TRY(run_deferred(h));
    if (h->A().side_used) HIP_TRY(h, hipStreamWaitEvent(ms, h->ev_join, 0));
    PE_TRY(aux_join(h, ms));  // ... and behind an earlier process_attestation / signature leg of this pipeline (they read
                              // the resident words on the state-transition stream)
    if (arena_kind == 0) {
        HIP_TRY(h, st.upload());
    } else {  // the bits by the copy engine from where they lie, then the zero pad and everything behind it
        HIP_TRY(h, hipMemcpyAsync(st.dev<uint8_t>(off_arena), bits_arena + lo, span,
                                  arena_kind == 2 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ms));
        HIP_TRY(h, hipMemcpyAsync(st.dev<uint8_t>(off_arena) + span, st.host<uint8_t>(off_arena) + span,
                                  st.used - (off_arena + span), hipMemcpyHostToDevice, ms));
    }
    lap.mark("agg.3a_h2d");
    {
        ProfScope ps(h, PE_KERNEL_BITS_UNION);
        // the kernels write their host-bound outputs straight into the pinned block (host-coherent, like the head
        // word): no device-to-host copy commands in a step
        launch_bits_union(ms, st.dev<UnionGroup>(off_ug), ng, st.dev<uint32_t>(off_ub), st.dev<uint8_t>(off_arena),
                          h->A().d_res_bits.as<uint32_t>(), h->A().d_res_info.as<uint32_t>(), ob.host<uint32_t>(off_obits),
                          ob.host<uint32_t>(off_oinfo));
    }
    lap.mark("agg.3b_union");
    if (want_pk) {
        pe_engine::PipeArena* arena = &h->A();
        const uint32_t* d_points = h->d_points.as<uint32_t>();
        const uint32_t* d_members = table_pk->d_members.as<uint32_t>();
        const uint32_t* d_union = arena->d_res_bits.as<uint32_t>();
        const G1Group* d_groups = st.dev<G1Group>(off_g1);
        uint8_t* out_pk = out_aggpk96 ? ob.host<uint8_t>(off_opk) : nullptr;
        uint32_t* jac = static_cast<uint32_t*>(dev_partials);
        // streaming pipelines launch the sums behind the step's fork-choice kernels (run_tree / pe_pipeline_end_lagged)
            const bool defer = on_side && h->streaming && !g1_chain_idle(h);
        if (defer) tune_arm = -1;  // the autotune's event pair assumes launch and read-back in one call
        auto launch_g1 = [h, arena, d_points, d_members, d_union, d_groups, plan_pk, out_pk, jac, on_side, gs, tune_arm]() -> int {
            hipStream_t ms_ = h->stream;
            if (on_side) {
                // everything enqueued on the engine's stream so far comes first: the union this sum reads, and -- when
                // the launch was deferred to the end of a streaming pipeline -- the step's fork-choice kernels, which
                // would otherwise queue behind an accumulation that fills every CU
                HIP_TRY(h, hipEventRecord(h->ev_fork, ms_));
                HIP_TRY(h, hipStreamWaitEvent(gs, h->ev_fork, 0));
                // a second aggregate in the SAME pipeline shares this arena's d_partials with the first one's finish
                if (arena->side_used) HIP_TRY(h, hipStreamWaitEvent(gs, h->ev_join, 0));
            } else {
                g1_stream_guard(h, gs);
            }
            int arm = tune_arm;
            if (arm >= 0) {
                if (!h->g1_tune_ev[0] && (hipEventCreate(&h->g1_tune_ev[0]) != hipSuccess ||
                                          hipEventCreate(&h->g1_tune_ev[1]) != hipSuccess)) {
                    h->g1_tune_ev[0] = h->g1_tune_ev[1] = nullptr;
                    arm = -1;
                } else {
                    (void)hipEventRecord(h->g1_tune_ev[0], gs);
                }
            }
            int rc = launch_g1_planned(h, d_points, d_members, d_union, d_groups, plan_pk, out_pk, jac, gs,
                                       on_side ? h->fin_stream : gs, on_side ? &arena->d_partials : nullptr,
                                       on_side ? &arena->d_lane_partials : nullptr);
            if (rc) return rc;
            if (arm >= 0) (void)hipEventRecord(h->g1_tune_ev[1], on_side ? h->g1_tail() : gs);
            if (on_side) {
                HIP_TRY(h, hipEventRecord(h->ev_join, h->g1_tail()));
                h->side_busy = true;
                h->side_ever = true;
                arena->side_used = true;
            }
            return PE_OK;
        };
        if (defer) {
            // scratch sizes are settled now, while nothing of the launch is in flight
            PE_TRY(ensure_quiesced(h, arena->d_partials,
                                   std::max<size_t>(PE_G1_PARTIAL_BYTES, (size_t)PE_G1_PARTIAL_BYTES * plan_pk.n_partials)));
            PE_TRY(ensure_quiesced(h, arena->d_lane_partials,
                                   (size_t)G1_LANE_PARTIAL_BYTES * G1_WG * ((plan_pk.n_slots + G1_WG - 1) / G1_WG)));
            h->deferred.push_back(launch_g1);
        } else {
            int rc = launch_g1();
            if (rc) return rc;
        }
        for (const XSeg& sg : xsegs) {  // further tables: same stream, same scratch, one after the other
            g1_stream_guard(h, ms);
            int rc = launch_g1_planned(h, d_points, sg.table->d_members.as<uint32_t>(), d_union,
                                       st.dev<G1Group>(off_g1x) + sg.begin, sg.plan,
                                       out_aggpk96 ? ob.host<uint8_t>(off_opkx) + 96ull * sg.begin : nullptr, nullptr, ms);
            if (rc) return rc;
            sg.table->stamp = ++h->table_stamp;
        }
        lap.mark("agg.3d_g1_launch");
        table_pk->stamp = ++h->table_stamp;
    }
    if (out_sig96) {  // bls.Aggregate: sum of the members' signature points (engine-owned scratch: completes in-call)
        g1_stream_guard(h, ms);
        HIP_TRY(h, h->d_tmp_be.ensure(96ull * n));
        HIP_TRY(h, h->d_tmp_points.ensure(4ull * G1_ROW_WORDS * n));
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, sig_points96, 96ull * n, hipMemcpyHostToDevice, ms));
        launch_g1_convert(ms, h->d_tmp_be.as<uint8_t>(), h->d_tmp_points.as<uint32_t>(), n);
        int rc = launch_g1_planned(h, h->d_tmp_points.as<uint32_t>(), st.dev<uint32_t>(off_idx), nullptr,
                                   st.dev<G1Group>(off_g1s), plan_sig, ob.host<uint8_t>(off_osig), nullptr, ms, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, /*caller_rows=*/n);
        if (rc) return rc;
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("agg.3e_d2h_pk_join");
    A.atts = atts;
    A.out_atts = out_atts;
    A.out_bits_arena = out_bits_arena;
    A.out_sig96 = out_sig96;
    A.out_aggpk96 = out_aggpk96;
    A.out_count = out_count;
    A.base = ob.base;
    A.off_obits = off_obits;
    A.off_oinfo = off_oinfo;
    A.off_opk = off_opk;
    A.off_osig = off_osig;
    A.off_opkx = off_opkx;
    A.tune_arm = tune_arm;
    {   // every union a whole number of words (committee sizes that are multiples of 32), except possibly the last one?
        bool same = true;
        for (uint32_t g = 0; g < ng && same; ++g) same = A.out_byte_off[g] == 4ull * A.out_word[g];
        A.packed_bytes = same ? (size_t)out_bytes : 0;
    }
    const int ai = h->cur;
    auto complete = [h, stp, info_p, ai, xgroups_p]() -> int {
        AggState& S = *stp;
        const uint8_t* pin = h->arena[ai].h_pin.as<uint8_t>() + S.base;
        if (S.tune_arm >= 0) {
            float ms_ = 0;
            if (hipEventElapsedTime(&ms_, h->g1_tune_ev[0], h->g1_tune_ev[1]) == hipSuccess && ms_ > 0)
                h->g1_tune_best[S.tune_arm] = std::min(h->g1_tune_best[S.tune_arm], ms_);
            if (++h->g1_tune_calls >= 4)
                h->g1_target_slots = h->g1_tune_best[1] < h->g1_tune_best[0] ? G1_TARGET_LANES / 2 : G1_TARGET_LANES;
        }
        const uint8_t* obits = pin + S.off_obits;
        const uint32_t* oinfo = reinterpret_cast<const uint32_t*>(pin + S.off_oinfo);
        if (S.packed_bytes) memcpy(S.out_bits_arena, obits, S.packed_bytes);  // one copy instead of one per group
        for (uint32_t g = 0; g < S.ng; ++g) {
            const uint32_t nb = S.out_atts[g].n_bits;
            if (!S.packed_bytes) memcpy(S.out_bits_arena + S.out_byte_off[g], obits + 4ull * S.out_word[g], (nb + 7) / 8);
            if (S.out_count) S.out_count[g] = oinfo[2 * g];
            if (oinfo[2 * g + 1]) {
                // members overlap: the summed signature counts a validator twice while bits and pubkey count it
                // once -- such an aggregate can never verify (A.8).  Say so instead of returning it as valid.
                S.out_atts[g].flags = (S.out_atts[g].flags & ~(uint32_t)PE_ATT_FLAG_SIGNATURE_VALID) | PE_ATT_FLAG_OVERLAPPING_BITS;
            }
        }
        info_p->assign(oinfo, oinfo + 2 * (size_t)S.ng);
        if (S.out_aggpk96) memcpy(S.out_aggpk96, pin + S.off_opk, 96ull * S.ng);
        if (S.out_aggpk96)  // groups of further tables: their sums were computed compacted, per table
            for (size_t k = 0; k < xgroups_p->size(); ++k)
                memcpy(S.out_aggpk96 + 96ull * (*xgroups_p)[k], pin + S.off_opkx + 96ull * k, 96);
        if (S.out_sig96) memcpy(S.out_sig96, pin + S.off_osig, 96ull * S.ng);
        return PE_OK;
    };
    unwind.armed = false;
    h->res_info_host = info_p;
    h->res_valid = true;
    h->res_arena = h->cur;
    ++h->res_generation;
    HostLap lap2(&h->trace);
    const int rc = finish_call(h, st, ob, complete, /*force_sync=*/out_sig96 != nullptr);
    lap2.mark("agg.4_wait_outputs");
    return rc;
}
}  // namespace posevo

extern "C" {

int pe_aggregate(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena, uint64_t arena_len,
                 const uint8_t* sig_points96, pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                 uint8_t* out_bits_arena, uint64_t out_arena_cap, uint8_t* out_sig96, uint8_t* out_aggpk96,
                 uint32_t* out_count)
{
    return aggregate_impl(h, atts, n, bits_arena, arena_len, sig_points96, out_atts, out_n_groups, group_of,
                          out_bits_arena, out_arena_cap, out_sig96, out_aggpk96, out_count, nullptr);
}

int pe_aggregate_partial(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                         uint64_t arena_len, pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                         uint8_t* out_bits_arena, uint64_t out_arena_cap, uint32_t* out_count, void* dev_partials,
                         uint32_t dev_partials_capacity)
{
    if (!dev_partials) return PE_ERR_INVALID_ARG;
    return aggregate_impl(h, atts, n, bits_arena, arena_len, nullptr, out_atts, out_n_groups, group_of,
                          out_bits_arena, out_arena_cap, nullptr, nullptr, out_count, dev_partials,
                          dev_partials_capacity);
}

// ---------------------------------------------------------------- process_attestation
int pe_process_attestation_batch(pe_engine* h, const pe_state_ctx* st, const pe_attestation* atts, uint32_t n,
                                 const uint8_t* bits_arena, uint64_t arena_len, int32_t* status,
                                 uint64_t* out_numerators)
{
    // (over device rows the flag pass runs on the state-transition stream and shares nothing with held-back fork-choice launches)
    int rc = need_init(h, /*flush=*/false, /*keep_held=*/atts == PE_ROWS_RESIDENT);
    if (rc) return rc;
    if (!st || (n && (!atts || !bits_arena || !status || !out_numerators))) return PE_ERR_INVALID_ARG;
    if (atts == PE_ROWS_RESIDENT) {
        if (bits_arena != PE_BITS_RESIDENT) return fail(h, PE_ERR_INVALID_ARG, "PE_ROWS_RESIDENT goes with PE_BITS_RESIDENT");
        return process_attestation_resident(h, st, n, status, out_numerators);
    }
    if (n == 0) return PE_OK;
    const bool resident = bits_arena == PE_BITS_RESIDENT;
    if (resident && !h->res_valid) return fail(h, PE_ERR_STATE, "PE_BITS_RESIDENT: no pe_aggregate result is resident");
    if (!resident && bits_on_device(bits_arena))
        return fail(h, PE_ERR_INVALID_ARG, "bits in device memory: hand them over through pe_aggregate + PE_BITS_RESIDENT");
    uint32_t tip;
    if (!find_block(h, to_root(st->chain_tip_root), &tip)) return fail(h, PE_ERR_UNKNOWN_ROOT, "chain tip unknown");
    const uint64_t spe = h->cfg.slots_per_epoch;
    const uint64_t cur_epoch = st->slot / spe;
    const uint64_t prev_epoch = cur_epoch > 0 ? cur_epoch - 1 : 0;
    const uint64_t sqrt_spe = isqrt64(spe);
    Checkpoint cj, pj;
    cj.epoch = st->current_justified_epoch; cj.root = to_root(st->current_justified_root);
    pj.epoch = st->previous_justified_epoch; pj.root = to_root(st->previous_justified_root);

    HostLap lap(&h->trace);
    uint64_t word_bound = 0;
    auto res_group_p = std::make_shared<std::vector<uint32_t>>(resident ? n : 0);
    std::vector<uint32_t>& res_group = *res_group_p;
    for (uint32_t i = 0; i < n; ++i) {
        if (resident) {
            if (!find_resident(h, atts[i], &res_group[i], i))
                return fail(h, PE_ERR_INVALID_ARG, "PE_BITS_RESIDENT: row is not a row of the last pe_aggregate");
            continue;
        }
        if (atts[i].n_bits > 0x7FFFFFFFu || (uint64_t)atts[i].bits_offset + ((uint64_t)atts[i].n_bits + 7) / 8 > arena_len)
            return fail(h, PE_ERR_INVALID_ARG, "attestation bits exceed the arena");
        word_bound += ((uint64_t)atts[i].n_bits + 31) / 32 + 1;
    }
    // get_block_root* walks from the chain tip: one per distinct slot asked, not one per row
    uint64_t anc_slot[64];
    uint32_t anc_idx[64];
    for (int k = 0; k < 64; ++k) anc_slot[k] = ~0ull;
    auto tip_ancestor = [&](uint64_t slot) {
        const int k = (int)(slot & 63);
        if (anc_slot[k] != slot) { anc_slot[k] = slot; anc_idx[k] = get_ancestor(h, tip, slot); }
        return anc_idx[k];
    };
    Stage stg(h);
    PE_TRY(stg.reserve(word_bound * 4 + (sizeof(AttRow) + 4) * (size_t)n + 4096));
    const size_t off_words = stg.alloc(word_bound * 4 + 4);
    const size_t off_rows = stg.alloc(sizeof(AttRow) * (size_t)n);
    const size_t off_nslot = stg.alloc(4ull * n);
    uint32_t* words = stg.host<uint32_t>(off_words);
    struct Acc { AttRow row; uint32_t src; CommitteeTable* table; uint32_t pos; };
    std::vector<Acc> acc;
    acc.reserve(n);
    uint32_t n_words = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const pe_attestation& a = atts[i];
        out_numerators[i] = 0;
        int32_t s = PE_ATT_OK;
        CommitteeTable* t = nullptr;
        uint64_t pos = 0;
        uint32_t size = 0;
        if (a.target_epoch != prev_epoch && a.target_epoch != cur_epoch) s = PE_ATT_TARGET_EPOCH_NOT_CURRENT_OR_PREVIOUS;  // pe:724
        else if (a.target_epoch != a.slot / spe) s = PE_ATT_TARGET_EPOCH_SLOT_MISMATCH;                                  // pe:725
        else if (!(a.slot + h->cfg.min_attestation_inclusion_delay <= st->slot && st->slot <= a.slot + spe))
            s = PE_ATT_INCLUSION_WINDOW;                                                                                 // pe:726
        else if (!(t = find_table(h, a.target_epoch))) s = PE_ATT_NO_COMMITTEE_TABLE;
        else if (a.index >= t->n_committees / spe) s = PE_ATT_COMMITTEE_INDEX_OUT_OF_RANGE;                              // pe:727
        else {
            pos = (a.slot % spe) * (t->n_committees / spe) + a.index;
            size = t->offsets[pos + 1] - t->offsets[pos];
            if (a.n_bits != size) s = PE_ATT_BITS_LENGTH_MISMATCH;                                                       // pe:730
        }
        uint32_t flag_mask = 0;
        if (s == PE_ATT_OK) {
            // get_attestation_participation_flag_indices (A.9)
            const Checkpoint& justified = a.target_epoch == cur_epoch ? cj : pj;
            Checkpoint src;
            src.epoch = a.source_epoch; src.root = to_root(a.source_root);
            if (!(src == justified)) s = PE_ATT_SOURCE_MISMATCH;  // assert is_matching_source
            else {
                // get_block_root(state, epoch) / get_block_root_at_slot(state, slot): the state's chain is the
                // ancestry of chain_tip_root (both slots are < state.slot by pe:726)
                const uint32_t tgt_blk = tip_ancestor(a.target_epoch * spe);
                const bool matching_target = memcmp(h->blocks[tgt_blk].root.data(), a.target_root, 32) == 0;
                const uint32_t head_blk = tip_ancestor(a.slot);
                const bool matching_head = matching_target && memcmp(h->blocks[head_blk].root.data(), a.beacon_block_root, 32) == 0;
                const uint64_t delay = st->slot - a.slot;
                if (delay <= sqrt_spe) flag_mask |= 1u;                                        // TIMELY_SOURCE
                if (matching_target && delay <= spe) flag_mask |= 2u;                          // TIMELY_TARGET
                if (matching_head && delay == h->cfg.min_attestation_inclusion_delay) flag_mask |= 4u;  // TIMELY_HEAD
            }
        }
        if (s == PE_ATT_OK) {
            // resident rows: emptiness / member overlap are settled on the device and reported at completion
            const uint32_t cnt = resident ? 1u : size ? pack_bits(bits_arena + a.bits_offset, size, words + n_words) : 0;
            if (cnt == 0) s = PE_ATT_EMPTY_OR_INVALID_INDICES;                                 // pe:736
            else if (!(a.flags & PE_ATT_FLAG_SIGNATURE_VALID)) s = PE_ATT_BAD_SIGNATURE;
            if (s == PE_ATT_OK) {
                Acc e;
                e.row.member_base = t->offsets[pos];
                e.row.n_bits = size;
                e.row.bits_word = resident ? h->res_groups[res_group[i]].word : n_words;
                e.row.gate = resident ? 2 * res_group[i] + 1 : NONE32;
                e.row.block_idx = 0;
                e.row.epoch_p1 = 0;
                e.row.order = 0;
                e.row.slot = 0;
                e.row.flag_mask = flag_mask;
                e.row.which = a.target_epoch == cur_epoch ? 0u : 1u;                           // pe:739-742
                e.src = i;
                e.table = t;
                e.pos = (uint32_t)pos;
                acc.push_back(e);
                if (!resident) n_words += (size + 31) / 32;
            }
        }
        status[i] = s;
    }
    if (acc.empty()) return PE_OK;
    lap.mark("proc.1_validate_pack");
    // ---- rounds: attestations of one round touch pairwise disjoint validators, so the order inside a
    // round is irrelevant; rounds run in order, which keeps the sequential semantics of pe:745-749 ----
    std::vector<uint32_t> round_of(acc.size());
    bool single_round = true;
    {
        // round = how many earlier attestations of this batch hit the same (table, committee); a flat counter per
        // table replaces a hash map (the common case is one attestation per committee: everything in round 0)
        std::vector<std::vector<uint16_t>> cnt(h->tables.size());
        for (size_t k = 0; k < acc.size(); ++k) {
            uint32_t r;
            if (acc[k].table->is_partition) {
                const size_t ti = (size_t)(acc[k].table - h->tables.data());
                if (cnt[ti].empty()) cnt[ti].assign(acc[k].table->n_committees, 0);
                r = cnt[ti][acc[k].pos]++;
            } else {
                r = (uint32_t)k;  // committees may overlap: fully sequential
            }
            round_of[k] = r;
            if (r) single_round = false;
        }
    }
    // rows sorted by (round, table); one launch per (round, table)
    std::vector<size_t> ord(acc.size());
    std::iota(ord.begin(), ord.end(), size_t(0));
    bool one_table = true;
    for (size_t k = 1; k < acc.size() && one_table; ++k) one_table = acc[k].table == acc[0].table;
    if (!(single_round && one_table))
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
            if (round_of[x] != round_of[y]) return round_of[x] < round_of[y];
            return acc[x].table < acc[y].table;
        });
    AttRow* rows = stg.host<AttRow>(off_rows);
    uint32_t* nslot = stg.host<uint32_t>(off_nslot);
    for (size_t k = 0; k < ord.size(); ++k) { rows[k] = acc[ord[k]].row; nslot[k] = acc[ord[k]].src; }
    OutBlock ob(h);
    const size_t off_num = ob.alloc(8ull * n);
    PE_TRY(ob.ensure());
    const uint32_t* d_bits = resident ? h->arena[h->res_arena].d_res_bits.as<uint32_t>() : stg.dev<uint32_t>(off_words);
    const uint32_t* d_gates = resident ? h->arena[h->res_arena].d_res_info.as<uint32_t>() : nullptr;
    HIP_TRY(h, stg.upload());
    memset(ob.host<uint8_t>(off_num), 0, 8ull * n);  // the kernel writes the numerators straight into the pinned block
    hipStream_t ss = state_stream_begin(h, /*reads_scratch=*/true, (uint32_t)std::min<size_t>(n, UINT32_MAX));  // behind the upload and the unions; beside whatever follows on the engine's stream
    for (size_t k = 0; k < ord.size();) {
        size_t e = k + 1;
        while (e < ord.size() && round_of[ord[e]] == round_of[ord[k]] && acc[ord[e]].table == acc[ord[k]].table) ++e;
        ProfScope ps(h, PE_KERNEL_PARTICIPATION, ss);
        launch_participation(ss, stg.dev<AttRow>(off_rows) + k, (uint32_t)(e - k),
                             acc[ord[k]].table->d_members.as<uint32_t>(), d_bits,
                             h->d_incr.as<uint16_t>(), st->base_reward_per_increment, h->d_part_cur.as<uint32_t>(),
                             h->d_part_prev.as<uint32_t>(), ob.host<uint64_t>(off_num), stg.dev<uint32_t>(off_nslot) + k,
                             d_gates);
        k = e;
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("proc.2_rounds_h2d_launch");
    auto src_p = std::make_shared<std::vector<uint32_t>>();
    if (resident)
        for (size_t k = 0; k < acc.size(); ++k) src_p->push_back(acc[k].src);
    std::shared_ptr<std::vector<uint32_t>> info_p = resident ? h->res_info_host : nullptr;
    const size_t ob_base = ob.base;
    const int ai = h->cur;
    auto complete = [h, n, out_numerators, status, ob_base, off_num, resident, src_p, res_group_p, info_p, ai]() -> int {
        memcpy(out_numerators, h->arena[ai].h_pin.as<uint8_t>() + ob_base + off_num, 8ull * n);
        if (resident) {
            const std::vector<uint32_t>& info = *info_p;
            for (uint32_t i : *src_p) {
                const uint32_t g = (*res_group_p)[i];
                if (2 * (size_t)g + 1 >= info.size()) return fail(h, PE_ERR_STATE, "resident aggregate did not complete");
                if (info[2 * g + 1]) status[i] = PE_ATT_BAD_SIGNATURE;
                else if (info[2 * g] == 0) status[i] = PE_ATT_EMPTY_OR_INVALID_INDICES;
            }
        }
        return PE_OK;
    };
    HostLap lap2(&h->trace);
    rc = finish_call(h, stg, ob, complete);
    lap2.mark("proc.3_wait_d2h");
    return rc;
}



int pe_participation_set(pe_engine* h, int which, const uint8_t* flags, uint64_t n)
{
    if (!h || !flags || n != h->n_val || (which != 0 && which != 1)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    DevBuf& b = which ? h->d_part_prev : h->d_part_cur;
    HIP_TRY(h, hipMemcpyAsync(b.p, flags, n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}
int pe_participation_get(pe_engine* h, int which, uint8_t* out_flags, uint64_t n)
{
    if (!h || !out_flags || n != h->n_val || (which != 0 && which != 1)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    DevBuf& b = which ? h->d_part_prev : h->d_part_cur;
    HIP_TRY(h, hipMemcpyAsync(out_flags, b.p, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}
int pe_participation_rotate(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    std::swap(h->d_part_cur, h->d_part_prev);  // previous = current
    // on the state stream, behind the flag passes that still write the old arrays (and off the fork-choice stream)
    // The array that becomes "current" was "previous": its last readers and writers are the flag passes of earlier steps, on
    // this stream; synchronous readers on the engine's stream (pe_participation_get, pe_ffg_balances) complete before they
    // return.  So the memset needs no ordering against the engine's stream -- unless a flag pass had to be placed there
    // (state_stream_begin's fall-back when its fork event could not be recorded): then this rotation forks behind it.
    hipStream_t ss;
    if (h->state_work_on_main) {
        h->state_work_on_main = false;
        ss = state_stream_begin(h);
    } else {
        ss = state_stream_unordered(h);
    }
    if (h->n_val) HIP_TRY(h, hipMemsetAsync(h->d_part_cur.p, 0, (h->n_val + 3) & ~uint64_t(3), ss));  // current = 0
    return PE_OK;
}

int pe_state_set_validators(pe_engine* h, uint64_t n, const uint64_t* effective_balance, const uint8_t* flags)
{
    if (!h || (n && (!effective_balance || !flags))) return PE_ERR_INVALID_ARG;
    if (n != h->n_val) return fail(h, PE_ERR_INVALID_ARG, "pe_state_set_validators: n differs from the registry size");
    PE_TRY(enter(h));
    std::vector<uint16_t> incr(n);
    const uint64_t inc = h->cfg.effective_balance_increment;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t q = effective_balance[i] / inc;
        if (q > 0xFFFF) return fail(h, PE_ERR_INVALID_ARG, "effective_balance / increment exceeds 65535");
        incr[i] = (uint16_t)q;
    }
    const size_t n4 = (n + 3) & ~size_t(3);
    HIP_TRY(h, h->d_sbalance.ensure(std::max<size_t>(64, n4 * 8)));
    HIP_TRY(h, h->d_sflags.ensure(std::max<size_t>(64, n4)));
    HIP_TRY(h, hipMemcpyAsync(h->d_sbalance.p, effective_balance, n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_sflags.p, flags, n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_incr.p, incr.data(), n * 2, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->state_view_set = true;
    return PE_OK;
}

// Read-back of the working-state view (checkpoint / resume): *out_is_set = 0 while the view still mirrors the registry.
int pe_state_get_validators(pe_engine* h, uint64_t n, uint64_t* out_effective_balance, uint8_t* out_flags, int* out_is_set)
{
    if (!h || !out_is_set || (n && (!out_effective_balance || !out_flags))) return PE_ERR_INVALID_ARG;
    if (n != h->n_val) return fail(h, PE_ERR_INVALID_ARG, "pe_state_get_validators: n differs from the registry size");
    PE_TRY(enter(h));
    *out_is_set = h->state_view_set ? 1 : 0;
    if (n && h->d_sbalance.p && h->d_sflags.p) {
        HIP_TRY(h, hipMemcpyAsync(out_effective_balance, h->d_sbalance.p, n * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(out_flags, h->d_sflags.p, n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return PE_OK;
}

// The committee tables the handle holds: epochs first (out_epochs NULL: count only), then one table at a time.
int pe_get_committee_epochs(pe_engine* h, uint64_t* out_epochs, uint32_t cap, uint32_t* out_n)
{
    if (!h || !out_n) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    uint32_t k = 0;
    for (auto& t : h->tables) {
        if (!t.n_committees) continue;
        if (out_epochs) {
            if (k >= cap) return fail(h, PE_ERR_CAPACITY, "pe_get_committee_epochs: more tables than cap");
            out_epochs[k] = t.epoch;
        }
        ++k;
    }
    *out_n = k;
    return PE_OK;
}
int pe_get_committees(pe_engine* h, uint64_t epoch, uint32_t* out_n_committees, uint32_t* out_offsets,
                      uint32_t offsets_cap, uint32_t* out_members, uint64_t members_cap)
{
    if (!h || !out_n_committees) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    CommitteeTable* t = find_table(h, epoch);
    if (!t) return fail(h, PE_ERR_NO_COMMITTEES, "no committee table for the epoch");
    *out_n_committees = t->n_committees;
    if (out_offsets) {
        if (offsets_cap < t->n_committees + 1) return fail(h, PE_ERR_CAPACITY, "pe_get_committees: offsets_cap too small");
        memcpy(out_offsets, t->offsets.data(), 4ull * (t->n_committees + 1));
    }
    if (out_members) {  // the members live on the device (a table computed by pe_compute_committees never left it)
        const uint64_t total = t->offsets.empty() ? 0 : t->offsets.back();
        if (members_cap < total) return fail(h, PE_ERR_CAPACITY, "pe_get_committees: members_cap too small");
        if (total) {
            HIP_TRY(h, hipMemcpyAsync(out_members, t->d_members.p, 4ull * total, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
    }
    return PE_OK;
}

int pe_ffg_balances(pe_engine* h, uint64_t out[3])
{
    if (!h || !out) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    OutBlock ob(h);
    const size_t off = ob.alloc(8ull * 3 * 256);
    PE_TRY(ob.ensure());
    uint32_t blocks = 0;
    if (h->n_val) {
        blocks = launch_ffg_balances(h->stream, h->d_sbalance.as<uint64_t>(), h->d_sflags.as<uint8_t>(),
                                     h->d_part_cur.as<uint8_t>(), h->d_part_prev.as<uint8_t>(), h->n_val,
                                     ob.dev<uint64_t>(off));
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, ob.download());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    uint64_t s[3] = {0, 0, 0};
    const uint64_t* p = ob.host<uint64_t>(off);
    for (uint32_t b = 0; b < blocks; ++b)
        for (int k = 0; k < 3; ++k) s[k] += p[3 * b + k];
    for (int k = 0; k < 3; ++k) out[k] = std::max<uint64_t>(h->cfg.effective_balance_increment, s[k]);  // get_total_balance
    return PE_OK;
}

}  // extern "C"
