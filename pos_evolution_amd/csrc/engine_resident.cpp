// engine_resident.cpp -- pe_aggregate / pe_on_attestation_batch / pe_process_attestation_batch over attestation rows
// that lie in DEVICE memory (PE_ROWS_RESIDENT, include/posevo.h).
//
// With host rows (engine_attest.cpp) one host thread groups the rows by 128-byte memcmp, walks
// validate_on_attestation (A.4) per row and lays out every descriptor array -- ~300 us per 1 M-validator epoch, more
// than the GPU needs for the epoch's G1 sums, and the same on every rank of a sharded run.  Here the host reads nothing
// of the rows: it sizes grids and scratch by upper bounds (n input rows >= groups), enqueues a fixed sequence of
// kernels (att_kernels.hip) and copies results out of the pinned block when the call completes.  Reference functions
// replaced: the grouping of the validator guide's aggregation (A.8; pe:474/659), validate_on_attestation called at
// pe:970, the asserts pe:724-730, get_attestation_participation_flag_indices (A.9), update_latest_messages
// pe:1435-1441, the flag loop pe:745-749.  Results are identical to the host path (tests/test_gpu_resident_rows.py).
//
// Committees are resolved against the tables of the store's CURRENT and PREVIOUS epoch (the only target epochs
// validate_on_attestation admits for gossip attestations); both must be partition tables (a real shuffling is).
#include "engine_internal.h"


using namespace posevo;

namespace posevo {

namespace {

struct RrLayout {  // typed views into PipeArena::d_rr
    AttPlan* plan;
    uint32_t *slot_of, *rep_of, *gid_of_row, *ubytes, *member_row;
    PlanSync* sync;
    PlanRec* rec;
    AttGroup* grp;
    UnionGroup* ug;
    G1Group* g1;
    AttRow *rows_fc, *rows_st;
    int32_t *status_fc, *status_st;
    uint32_t *crow_start[2], *crow_cursor[2], *crow_cnt[2], *crow_list[2];
    size_t bytes;
};

RrLayout rr_layout(uint8_t* base, uint32_t cap_n, uint32_t cap_c)
{
    RrLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = base ? base + off : nullptr;
        off = (off + bytes + 255) & ~size_t(255);
        return p;
    };
    L.plan = reinterpret_cast<AttPlan*>(take(sizeof(AttPlan)));
    L.slot_of = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    L.rep_of = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    L.gid_of_row = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    L.sync = reinterpret_cast<PlanSync*>(take(sizeof(PlanSync)));
    L.rec = reinterpret_cast<PlanRec*>(take(sizeof(PlanRec) * ((size_t)cap_n / PLAN_WG + 1)));
    L.ubytes = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    L.member_row = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    L.grp = reinterpret_cast<AttGroup*>(take(sizeof(AttGroup) * (size_t)cap_n));
    L.ug = reinterpret_cast<UnionGroup*>(take(sizeof(UnionGroup) * (size_t)cap_n));
    L.g1 = reinterpret_cast<G1Group*>(take(sizeof(G1Group) * (size_t)cap_n));
    L.rows_fc = reinterpret_cast<AttRow*>(take(sizeof(AttRow) * (size_t)cap_n));
    L.rows_st = reinterpret_cast<AttRow*>(take(sizeof(AttRow) * (size_t)cap_n));
    L.status_fc = reinterpret_cast<int32_t*>(take(4ull * cap_n));
    L.status_st = reinterpret_cast<int32_t*>(take(4ull * cap_n));
    for (int t = 0; t < 2; ++t) {
        L.crow_start[t] = reinterpret_cast<uint32_t*>(take(4ull * (cap_c + 1)));
        L.crow_cursor[t] = reinterpret_cast<uint32_t*>(take(4ull * (cap_c + 1)));
        L.crow_cnt[t] = reinterpret_cast<uint32_t*>(take(4ull * (cap_c + 1)));
        L.crow_list[t] = reinterpret_cast<uint32_t*>(take(4ull * cap_n));
    }
    L.bytes = off;
    return L;
}

// One of the arena's two scratch sets (0: its own, 1: the exchanged aggregate's), by reference
struct RrSet {
    DevBuf &tab, &rr, &bits, &info;
    uint32_t &rows_cap, &comm_cap, &tab_size;
};
RrSet rr_set(pe_engine::PipeArena& a, int set)
{
    if (set == 0) return RrSet{a.d_rr_tab, a.d_rr, a.d_res_bits, a.d_res_info, a.rr_rows_cap, a.rr_comm_cap, a.rr_tab_size};
    return RrSet{a.x_rr_tab, a.x_rr, a.x_res_bits, a.x_res_info, a.x_rows_cap, a.x_comm_cap, a.x_tab_size};
}
RrLayout rr_of(pe_engine::PipeArena& a, int set = 0)
{
    const RrSet s = rr_set(a, set);
    return rr_layout(s.rr.as<uint8_t>(), s.rows_cap, s.comm_cap);
}

// The arena's scratch for n input rows and tables of up to n_comm committees.  Growing waits for everything enqueued --
// so when one arena has to grow, every arena of the rotation grows with it: the steps of a stream look alike, and an arena
// that grew at ITS first use would drain the pipeline once per arena (and with it the fork-choice launches held back for
// pairing, engine_pair.cpp: the first lag + 1 steps of a run went out unpaired).
int rr_ensure(pe_engine* h, pe_engine::PipeArena& arena, int set, uint32_t n, uint32_t n_comm)
{
    {
        const RrSet rs = rr_set(arena, set);
        if (n <= rs.rows_cap && n_comm <= rs.comm_cap && rs.rr.p && rs.tab.p) return PE_OK;
    }
    PE_TRY(flush_pending(h));
    HIP_TRY(h, hipDeviceSynchronize());
    uint32_t cap_n = 1024, cap_c = 64;
    for (int i = 0; i < h->n_arenas; ++i) {
        const RrSet o = rr_set(h->arena[i], set);
        cap_n = std::max(cap_n, o.rows_cap);
        cap_c = std::max(cap_c, o.comm_cap);
    }
    while (cap_n < n) cap_n *= 2;
    while (cap_c < n_comm) cap_c *= 2;
    uint32_t tab = 2048;
    while (tab < 2 * cap_n) tab <<= 1;
    const RrLayout L = rr_layout(nullptr, cap_n, cap_c);
    for (int i = 0; i < h->n_arenas; ++i) {
        pe_engine::PipeArena& other = h->arena[i];
        if (set != 0 && &other != &arena) continue;  // the exchanged aggregate's set: where it is used
        const RrSet a = rr_set(other, set);
        if (cap_n <= a.rows_cap && cap_c <= a.comm_cap && a.rr.p && a.tab.p) continue;
        a.rr.release();
        a.tab.release();
        HIP_TRY(h, a.rr.ensure(L.bytes));
        HIP_TRY(h, a.tab.ensure(8ull * tab));  // slot -> first row | slot -> class size
        // the grouping table is kept empty by its users (k_att_members clears what k_att_ingest filled); the plan's error
        // word starts at zero
        HIP_TRY(h, hipMemsetAsync(a.tab.p, 0xFF, 4ull * tab, h->stream));
        HIP_TRY(h, hipMemsetAsync(a.tab.as<uint8_t>() + 4ull * tab, 0, 4ull * tab, h->stream));
        HIP_TRY(h, hipMemsetAsync(a.rr.p, 0, L.bytes, h->stream));
        a.rows_cap = cap_n;
        a.comm_cap = cap_c;
        a.tab_size = tab;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}

// The tables a row's committee may come from: the store's current and previous epoch.
int candidate_tables(pe_engine* h, TablesDev* out, CommitteeTable* tabs[2])
{
    memset(out, 0, sizeof(*out));
    out->slots_per_epoch = h->cfg.slots_per_epoch;
    const uint64_t cur = epoch_at_slot(h, current_slot(h));
    const uint64_t prev = cur > 0 ? cur - 1 : cur;
    const uint64_t epochs[2] = {cur, prev};
    for (int k = 0; k < 2; ++k) {
        tabs[k] = nullptr;
        out->t[k].epoch = epochs[k];
        if (k == 1 && prev == cur) continue;
        CommitteeTable* t = find_table(h, epochs[k]);
        if (!t) continue;
        if (!t->is_partition || t->n_val_at_load != h->n_val || !t->d_inv_comm.p)
            return fail(h, PE_ERR_INVALID_ARG,
                        "rows in device memory: the committee tables of the current / previous epoch must partition the "
                        "registry they were loaded for (use host rows otherwise)");
        tabs[k] = t;
        out->t[k].members = t->d_members.as<uint32_t>();
        out->t[k].offsets = t->d_offsets.as<uint32_t>();
        out->t[k].inv_comm = t->d_inv_comm.as<uint32_t>();
        out->t[k].inv_pos = t->d_inv_pos.as<uint32_t>();
        out->t[k].n_committees = t->n_committees;
        out->t[k].valid = 1;
    }
    return PE_OK;
}

uint32_t g1_target_slots(const pe_engine* h)
{
    if (h->streaming) return G1_TARGET_LANES / 2;  // one wave per SIMD: see aggregate_impl
    return h->g1_target_slots ? h->g1_target_slots : G1_TARGET_LANES;
}

int plan_error_to_status(pe_engine* h, uint32_t err, const char* who)
{
    switch (err) {
        case 0: return PE_OK;
        case 10: return fail(h, PE_ERR_CAPACITY, std::string(who) + ": output capacity too small for the groups formed");
        case 11: return fail(h, PE_ERR_NO_COMMITTEES, std::string(who) + ": no committee table for a group's target epoch");
        case 12: return fail(h, PE_ERR_NO_DEVICE, std::string(who) + ": a workgroup of k_att_plan gave up waiting for its predecessors' records");
        default:
            return fail(h, PE_ERR_INVALID_ARG, std::string(who) + ": a row was refused on the device (bits exceed the arena, "
                        "target epoch beyond 32 bits, committee index out of range or len(aggregation_bits) != len(committee))");
    }
}

}  // namespace

bool rows_on_device(const void* p)
{
    if (!p || p == (const void*)PE_ROWS_RESIDENT) return false;
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, p) == hipSuccess) return pa.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // plain host memory is "invalid value" to older runtimes: not an error here
    return false;
}

BlockTableDev block_table_dev(const pe_engine* h)
{
    BlockTableDev bt;
    bt.root_tab = h->d_broot_tab.as<uint32_t>();
    bt.root_mask = h->broot_mask;
    bt.roots = h->d_broots.as<uint8_t>();
    bt.slot_pos = h->d_bslot_pos.as<unsigned long long>();
    bt.parent_pos = h->d_tparent.as<uint32_t>();
    bt.pos_of_idx = h->d_tpos.as<uint32_t>();
    bt.n_blocks = (uint32_t)h->blocks.size();
    return bt;
}

// ---------------------------------------------------------------- pe_aggregate, rows in device memory
int aggregate_resident(pe_engine* h, const pe_attestation* d_rows, uint32_t n, const uint8_t* bits_arena,
                       uint64_t arena_len, pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                       uint8_t* out_bits_arena, uint64_t out_arena_cap, uint8_t* out_aggpk96, uint32_t* out_count,
                       uint32_t* dev_partials, int set, const uint32_t* n_dev)
{
    if (!h->initialised) return fail(h, PE_ERR_STATE, "rows in device memory: the store's clock picks the committee tables; call pe_store_init first");
    if ((uintptr_t)d_rows & 15) return fail(h, PE_ERR_INVALID_ARG, "rows in device memory must be 16-byte aligned");
    if (arena_len >= 0xFFFFFFF0ull) return fail(h, PE_ERR_CAPACITY, "bit arena exceeds 4 GiB");
    if (n >= PLAN_MAX_ROWS) return fail(h, PE_ERR_CAPACITY, "rows in device memory: at most 2^24 - 1 rows per aggregate");
    const bool want_pk = out_aggpk96 != nullptr || dev_partials != nullptr;
    if (want_pk && !h->have_points) return fail(h, PE_ERR_STATE, "aggregate pubkeys requested but no pubkeys loaded");
    HostLap lap(&h->trace);
    TablesDev tables;
    CommitteeTable* tabs[2];
    PE_TRY(candidate_tables(h, &tables, tabs));
    const uint32_t n_comm = std::max(tables.t[0].n_committees, tables.t[1].n_committees);
    pe_engine::PipeArena& A = h->A();
    PE_TRY(rr_ensure(h, A, set, n, n_comm));
    const RrLayout L = rr_of(A, set);
    const RrSet RS = rr_set(A, set);
    // upper bounds: groups <= n; lane slots of the G1 plan <= slot_cap (k_att_plan lengthens the lanes instead)
    const uint32_t target = g1_target_slots(h);
    const uint32_t slot_cap = std::max<uint32_t>(2 * G1_TARGET_LANES, (n + G1_WG - 1) / G1_WG * G1_WG);
    const size_t words_cap = (size_t)out_arena_cap / 4 + n + 16;
    Stage st(h);
    PE_TRY(st.reserve((size_t)arena_len + 256));
    const size_t pad_at = ((size_t)arena_len + 15) & ~size_t(15);  // 32 zero bytes behind the bits, written by k_att_ingest
    const size_t off_arena = st.alloc(pad_at + 32);
    if (set == 0) {
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_res_bits, words_cap * 4 + 64));
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_res_info, 8ull * n + 64));
    } else {
        PE_TRY(ensure_quiesced(h, RS.bits, words_cap * 4 + 64));
        PE_TRY(ensure_quiesced(h, RS.info, 8ull * n + 64));
    }
    OutBlock ob(h);
    const size_t off_plan = ob.alloc(sizeof(AttPlan));
    const size_t off_rows = ob.alloc(sizeof(pe_attestation) * (size_t)n);
    const size_t off_gof = ob.alloc(4ull * n);
    const size_t off_obits = ob.alloc(words_cap * 4);
    const size_t off_oinfo = ob.alloc(8ull * n);
    const size_t off_opk = out_aggpk96 ? ob.alloc(96ull * n) : 0;
    PE_TRY(ob.ensure());
    if (want_pk) {  // scratch of the G1 chain, sized by the bounds before anything is in flight
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_partials, std::max<size_t>(PE_G1_PARTIAL_BYTES, (size_t)PE_G1_PARTIAL_BYTES * n)));
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_lane_partials, (size_t)G1_LANE_PARTIAL_BYTES * slot_cap));
        PE_TRY(ensure_quiesced(h, h->d_partials, std::max<size_t>(PE_G1_PARTIAL_BYTES, (size_t)PE_G1_PARTIAL_BYTES * n)));
        PE_TRY(ensure_quiesced(h, h->d_lane_partials, (size_t)G1_LANE_PARTIAL_BYTES * slot_cap));
    }
    lap.mark("ragg.1_reserve");
    *out_n_groups = 0;  // known when the call completes
    AttPlan* plan_host = ob.host<AttPlan>(off_plan);
    memset(plan_host, 0, sizeof(AttPlan));
    plan_host->error = 0xFFFFFFFFu;  // "k_att_plan has not run": a completion that finds it reports a failed launch
    // ---- device ----
    int arena_kind = 0;  // 0 pageable host, 1 pinned host, 2 device
    if (arena_len) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, bits_arena) == hipSuccess) {
            if (pa.type == hipMemoryTypeDevice) arena_kind = 2;
            else if (pa.type == hipMemoryTypeHost) arena_kind = 1;
        } else {
            (void)hipGetLastError();
        }
    }
    hipStream_t ms = h->stream;
    // Fork-choice launches held back by the PREVIOUS step of a streaming caller go out pairwise with this aggregate's row
    // kernels (engine_pair.cpp).  Anything else that is held -- launches of THIS pipeline (they read the scratch this
    // call rewrites), or an aggregate in a form that is not the plain streaming one -- goes out first, alone and in order.
    const bool pair_ok = h->held.active && h->held.arena != h->cur && set == 0 && !n_dev && hold_eligible(h);
    if (h->held.active && !pair_ok) PE_TRY(held_issue(h));
    // an earlier aggregate of THIS pipeline still has to read the arena's resident words / descriptors from its G1 launch
    // (deferred in a streaming pipeline): issue it, then order this call's kernels behind that chain
    // The exchanged aggregate of a committee-sharded step has a scratch set of its own (set 1) precisely so as NOT to wait.
    if (set == 0) {
        if (!h->deferred.empty()) PE_TRY(run_deferred(h));
        if (A.side_used) HIP_TRY(h, hipStreamWaitEvent(ms, h->ev_join, 0));
        if (sig_batch_holds(h, h->cur)) PE_TRY(sig_batch_flush(h));  // a collected signature leg of this pipeline reads them too
        if (A.leg_used) HIP_TRY(h, hipStreamWaitEvent(ms, A.ev_leg, 0));
        PE_TRY(aux_join(h, ms));  // ... or from a process_attestation / signature leg on the state-transition stream
    }
    h->res_valid = false;
    h->rr.valid = false;
    const size_t deferred_before = h->deferred.size();
    struct Unwind {
        pe_engine* h; size_t keep; bool armed = true;
        ~Unwind() { if (armed) { h->rr.valid = false; if (h->deferred.size() > keep) h->deferred.resize(keep); } }
    } unwind{h, deferred_before};
    bool ingest_copies = false;
    if (arena_kind == 0) {
        memcpy(st.host<uint8_t>(off_arena), bits_arena, arena_len);
        memset(st.host<uint8_t>(off_arena) + arena_len, 0, pad_at + 32 - arena_len);
        HIP_TRY(h, st.upload());
    } else if (arena_len) {  // bytes [arena_len, pad_at) may keep old bits: they lie inside the last dword pair only when
                             // arena_len is not a multiple of 4, and then belong to no member (masked by n_bits)
        // device memory at a 16-byte boundary: k_att_ingest brings the bits in itself (no copy command in the chain)
        if (arena_kind == 2 && (reinterpret_cast<uintptr_t>(bits_arena) & 15) == 0 && n > 0) ingest_copies = true;
        else HIP_TRY(h, hipMemcpyAsync(st.dev<uint8_t>(off_arena), bits_arena, arena_len,
                                       arena_kind == 2 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ms));
    }
    lap.mark("ragg.2_bits");
    uint32_t* cnt_tab = RS.tab.as<uint32_t>() + RS.tab_size;
    const IngestArgs ia{d_rows, n, RS.tab.as<uint32_t>(), cnt_tab, RS.tab_size - 1, L.slot_of, arena_len, L.plan,
                        st.dev<uint8_t>(off_arena) + pad_at, n_dev, ingest_copies ? bits_arena : nullptr,
                        ingest_copies ? st.dev<uint8_t>(off_arena) : nullptr};
    AttPlanArgs pa;
    pa.rows = d_rows;
    pa.n = n;
    pa.n_dev = n_dev;
    pa.tab = RS.tab.as<uint32_t>();
    pa.cnt_tab = cnt_tab;
    pa.slot_of = L.slot_of;
    pa.rep_of = L.rep_of;
    pa.gid_of_row = L.gid_of_row;
    pa.grp = L.grp;
    pa.ug = L.ug;
    for (int t = 0; t < 2; ++t) {
        pa.crow_start[t] = L.crow_start[t];
        pa.crow_cursor[t] = L.crow_cursor[t];
        pa.crow_cnt[t] = L.crow_cnt[t];
    }
    pa.sync = L.sync;
    pa.rec = L.rec;
    pa.plan = L.plan;
    pa.plan_host = plan_host;
    pa.out_arena_cap = out_arena_cap;
    pa.target_slots = target;
    pa.slot_cap = slot_cap;
    pa.min_k = G1_MIN_K;
    pa.want_pk = want_pk ? 1u : 0u;
    pa.tables = tables;
    const MembersArgs ma{d_rows, n, RS.tab.as<uint32_t>(), cnt_tab, L.slot_of, L.rep_of, L.gid_of_row, L.grp, L.plan, L.ubytes,
                         L.member_row, group_of ? ob.host<uint32_t>(off_gof) : nullptr, ob.host<uint8_t>(off_rows), n_dev,
                         L.g1, {L.crow_cursor[0], L.crow_cursor[1]}, {L.crow_list[0], L.crow_list[1]}};
    const UnionArgs ua{L.ug, n, L.ubytes, st.dev<uint8_t>(off_arena), RS.bits.as<uint32_t>(), RS.info.as<uint32_t>(),
                       ob.host<uint32_t>(off_obits), ob.host<uint32_t>(off_oinfo), L.plan};
    // ingest -> plan -> members -> union, each one beside the held-back fork-choice kernel of the previous step if there is one
    PE_TRY(launch_rows_paired(h, ia, pa, ma, ua));
    HIP_TRY(h, hipGetLastError());
    lap.mark("ragg.3_group_union");
    if (want_pk) {
        const bool on_side = h->pipelining && h->side_stream && h->stream == h->own_stream;
        h->last_agg_on_side = on_side;
        pe_engine::PipeArena* arena = &A;
        const uint32_t* d_points = h->d_points.as<uint32_t>();
        const uint32_t* d_union = RS.bits.as<uint32_t>();
        uint8_t* out_pk = out_aggpk96 ? ob.host<uint8_t>(off_opk) : nullptr;
        G1Plan bound;
        bound.n_groups = n;
        bound.n_slots = slot_cap;
        bound.n_partials = n;
        const G1Group* d_groups = L.g1;
        const AttPlan* d_plan = L.plan;
        auto launch_g1 = [h, arena, d_points, tables, d_union, d_groups, bound, out_pk, on_side, d_plan, dev_partials]() -> int {
            hipStream_t gs = on_side ? h->side_stream : h->stream;
            if (on_side) {  // behind everything enqueued on the engine's stream so far (see aggregate_impl)
                HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
                HIP_TRY(h, hipStreamWaitEvent(gs, h->ev_fork, 0));
            }
            if (on_side) {
                if (arena->side_used) HIP_TRY(h, hipStreamWaitEvent(gs, h->ev_join, 0));
            } else {
                g1_stream_guard(h, gs);
            }
            PE_TRY(launch_g1_planned(h, d_points, tables.t[0].members, d_union, d_groups, bound, out_pk, dev_partials, gs,
                                     on_side ? h->fin_stream : gs, on_side ? &arena->d_partials : nullptr,
                                     on_side ? &arena->d_lane_partials : nullptr, d_plan, tables.t[1].members));
            if (on_side) {
                HIP_TRY(h, hipEventRecord(h->ev_join, h->g1_tail()));
                h->side_busy = true;
                h->side_ever = true;
                arena->side_used = true;
            }
            return PE_OK;
        };
        // (Launched with its own aggregate instead of held until the next one -- possible since the accumulation keeps itself
        // one per CU -- the holes a late host step leaves on the device go away and the accumulations stretch by what the
        // holes cost: 0.278 / 0.281 against 0.275 / 0.276 ms per step on the driver's command, profiles/NOTES_r06.md 4.)
        if (on_side && h->streaming && !g1_chain_idle(h)) h->deferred.push_back(launch_g1);
        else PE_TRY(launch_g1());
        lap.mark("ragg.4_g1");
    }
    for (int t = 0; t < 2; ++t)
        if (tabs[t]) tabs[t]->stamp = ++h->table_stamp;
    unwind.armed = false;
    h->rr.valid = true;
    h->rr.arena = h->cur;
    h->rr.set = set;
    h->rr.n_in = n;
    h->rr.rows = d_rows;
    h->rr.tables = tables;
    ++h->rr.generation;
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_plan, off_rows, off_gof, off_obits, off_oinfo, off_opk, n, out_atts, out_n_groups,
                     group_of, out_bits_arena, out_aggpk96, out_count]() -> int {
        const uint8_t* pin = h->arena[ai].h_pin.as<uint8_t>() + base;
        const AttPlan P = *reinterpret_cast<const AttPlan*>(pin + off_plan);
        if (P.error) return plan_error_to_status(h, P.error == 0xFFFFFFFFu ? 1u : P.error, "pe_aggregate (rows in device memory)");
        const uint32_t ng = P.n_groups;
        if (ng > n) return fail(h, PE_ERR_NO_DEVICE, "k_att_plan returned more groups than rows");
        *out_n_groups = ng;
        memcpy(out_atts, pin + off_rows, sizeof(pe_attestation) * (size_t)ng);
        if (group_of) memcpy(group_of, pin + off_gof, 4ull * n);
        const uint32_t* obits = reinterpret_cast<const uint32_t*>(pin + off_obits);
        const uint32_t* oinfo = reinterpret_cast<const uint32_t*>(pin + off_oinfo);
        if (P.packed_same) memcpy(out_bits_arena, obits, P.out_bytes);
        uint32_t word = 0;
        for (uint32_t g = 0; g < ng; ++g) {
            pe_attestation& o = out_atts[g];
            const uint32_t nb = o.n_bits;
            if (!P.packed_same) memcpy(out_bits_arena + o.bits_offset, obits + word, (nb + 7) / 8);
            word += (nb + 31) / 32;
            if (out_count) out_count[g] = oinfo[2 * g];
            if (oinfo[2 * g + 1])  // members overlap (A.8): never verifiable, say so
                o.flags = (o.flags & ~(uint32_t)PE_ATT_FLAG_SIGNATURE_VALID) | PE_ATT_FLAG_OVERLAPPING_BITS;
        }
        if (out_aggpk96) memcpy(out_aggpk96, pin + off_opk, 96ull * ng);
        return PE_OK;
    };
    HostLap lap2(&h->trace);
    const int rc = finish_call(h, st, ob, complete);
    lap2.mark("ragg.5_wait_outputs");
    return rc;
}

int resident_plan_dev(pe_engine* h, const AttPlan** out)
{
    if (!h->rr.valid) return fail(h, PE_ERR_STATE, "no aggregate over rows in device memory on this handle");
    *out = rr_of(h->arena[h->rr.arena], h->rr.set).plan;
    return PE_OK;
}

int resident_parts(pe_engine* h, ResidentParts* out)
{
    if (!h->rr.valid) return fail(h, PE_ERR_STATE, "no aggregate over rows in device memory on this handle");
    pe_engine::PipeArena& RA = h->arena[h->rr.arena];
    const RrLayout L = rr_of(RA, h->rr.set);
    const RrSet RS = rr_set(RA, h->rr.set);
    out->rows = h->rr.rows;
    out->grp = L.grp;
    out->plan = L.plan;
    out->res_bits = RS.bits.as<uint32_t>();
    out->res_info = RS.info.as<uint32_t>();
    out->n_in = h->rr.n_in;
    return PE_OK;
}

int resident_lists(pe_engine* h, const UnionGroup** ug, const uint32_t** member_row)
{
    if (!h->rr.valid) return fail(h, PE_ERR_STATE, "no aggregate over rows in device memory on this handle");
    const RrLayout L = rr_of(h->arena[h->rr.arena], h->rr.set);
    *ug = L.ug;
    *member_row = L.member_row;
    return PE_OK;
}

// ---------------------------------------------------------------- on_attestation x groups of the resident aggregate
static int resident_precheck(pe_engine* h, const char* who)
{
    if (!h->rr.valid) return fail(h, PE_ERR_STATE, std::string(who) + ": PE_ROWS_RESIDENT without a pe_aggregate over rows in device memory on this handle");
    TablesDev now;
    CommitteeTable* tabs[2];
    PE_TRY(candidate_tables(h, &now, tabs));
    for (int t = 0; t < 2; ++t)
        if (now.t[t].epoch != h->rr.tables.t[t].epoch || now.t[t].valid != h->rr.tables.t[t].valid ||
            now.t[t].members != h->rr.tables.t[t].members)
            return fail(h, PE_ERR_STATE, std::string(who) + ": the store's clock or its committee tables changed since the resident pe_aggregate: aggregate again");
    return PE_OK;
}

int on_attestation_resident(pe_engine* h, uint32_t cap, int32_t* status, uint32_t* out_count)
{
    PE_TRY(resident_precheck(h, "pe_on_attestation_batch"));
    if (cap == 0) return PE_OK;
    // launches held back by an earlier pipeline, or already some of this one (a second batch, a head in front of it): out
    // first, in order
    if (h->held.active && (h->held.arena != h->cur || h->held.have_fc || h->held.have_head)) PE_TRY(held_issue(h));
    HostLap lap(&h->trace);
    PE_TRY(refresh_tree(h));
    pe_engine::PipeArena& RA = h->arena[h->rr.arena];
    const RrLayout L = rr_of(RA, h->rr.set);
    const RrSet RS = rr_set(RA, h->rr.set);
    Stage st(h);
    OutBlock ob(h);
    const size_t off_status = ob.alloc(4ull * cap);
    const size_t off_count = ob.alloc(4ull * cap);
    const size_t off_err = ob.alloc(16);
    PE_TRY(ob.ensure());
    *ob.host<uint32_t>(off_err) = 0;
    FcCtx fc;
    fc.cur_slot = current_slot(h);
    fc.cur_epoch = epoch_at_slot(h, fc.cur_slot);
    fc.prev_epoch = fc.cur_epoch > 0 ? fc.cur_epoch - 1 : 0;
    fc.slots_per_epoch = h->cfg.slots_per_epoch;
    const uint32_t n_launch = std::max<uint32_t>(h->rr.n_in, 1);  // one lane per possible group
    // entries past the groups formed read "nothing applied"
    memset(ob.host<int32_t>(off_status), 0, 4ull * cap);
    memset(ob.host<uint32_t>(off_count), 0, 4ull * cap);
    const ValidateFcArgs va{h->rr.rows, L.grp, L.plan, n_launch, cap, block_table_dev(h), fc, RS.info.as<uint32_t>(), L.rows_fc,
                            L.status_fc, ob.host<int32_t>(off_status), ob.host<uint32_t>(off_count), ob.host<uint32_t>(off_err)};
    const LmdVmArgs la{L.rows_fc, h->rr.tables, {L.crow_start[0], L.crow_start[1]}, {L.crow_list[0], L.crow_list[1]}, L.plan,
                       RS.bits.as<uint32_t>(), h->d_flags.as<uint8_t>(), h->n_val, h->d_vote_key.as<uint64_t>(),
                       h->d_vote_block.as<uint32_t>(), h->cfg.vote_expiry_slots ? h->d_vote_slot.as<uint32_t>() : nullptr,
                       reinterpret_cast<const uint32_t*>(L.status_fc)};
    if (hold_eligible(h) && h->rr.arena == h->cur && h->rr.set == 0 && h->n_val) {
        // a streaming step: validate + LMD go out with the NEXT aggregate's row kernels (engine_pair.cpp); they read this
        // arena's scratch and unions, which stay untouched until the arena comes round again (lag depth + 1 steps)
        h->held.validate = va;
        h->held.lmd = la;
        h->held.have_fc = true;
        h->held.active = true;
        h->held.arena = h->cur;
    } else {
        if (h->held.active) PE_TRY(held_issue(h));
        {
            ProfScope ps(h, PE_KERNEL_ATT_VALIDATE);  // timeline mode only
            launch_att_validate_fc(h->stream, va);
        }
        ProfScope ps(h, PE_KERNEL_LMD);
        launch_lmd_vm_tables(h->stream, la);
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("ratt.1_launch");
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_status, off_count, off_err, cap, status, out_count]() -> int {
        const uint8_t* pin = h->arena[ai].h_pin.as<uint8_t>() + base;
        const uint32_t err = *reinterpret_cast<const uint32_t*>(pin + off_err);
        if (err) return plan_error_to_status(h, err, "pe_on_attestation_batch (PE_ROWS_RESIDENT)");
        memcpy(status, pin + off_status, 4ull * cap);
        if (out_count) memcpy(out_count, pin + off_count, 4ull * cap);
        return PE_OK;
    };
    HostLap lap2(&h->trace);
    const int rc = finish_call(h, st, ob, complete);
    lap2.mark("ratt.2_wait_outputs");
    return rc;
}

// ---------------------------------------------------------------- process_attestation x groups of the resident aggregate
int process_attestation_resident(pe_engine* h, const pe_state_ctx* sc, uint32_t cap, int32_t* status, uint64_t* out_numerators)
{
    PE_TRY(resident_precheck(h, "pe_process_attestation_batch"));
    if (cap == 0) return PE_OK;
    const uint64_t spe = h->cfg.slots_per_epoch;
    if (spe > 64) return fail(h, PE_ERR_CAPACITY, "PE_ROWS_RESIDENT: SLOTS_PER_EPOCH above 64");
    uint32_t tip;
    if (!find_block(h, to_root(sc->chain_tip_root), &tip)) return fail(h, PE_ERR_UNKNOWN_ROOT, "chain tip unknown");
    HostLap lap(&h->trace);
    PE_TRY(refresh_tree(h));
    pe_engine::PipeArena& RA = h->arena[h->rr.arena];
    const RrLayout L = rr_of(RA, h->rr.set);
    const RrSet RS = rr_set(RA, h->rr.set);
    Stage st(h);
    StateCtxDev S;  // travels as a kernel argument: no copy command
    memset(&S, 0, sizeof(S));
    S.slot = sc->slot;
    S.cur_epoch = sc->slot / spe;
    S.prev_epoch = S.cur_epoch > 0 ? S.cur_epoch - 1 : 0;
    S.slots_per_epoch = spe;
    S.min_inclusion_delay = h->cfg.min_attestation_inclusion_delay;
    S.sqrt_spe = isqrt64(spe);
    S.cj_epoch = sc->current_justified_epoch;
    S.pj_epoch = sc->previous_justified_epoch;
    memcpy(S.cj_root, sc->current_justified_root, 32);
    memcpy(S.pj_root, sc->previous_justified_root, 32);
    S.base_reward_per_increment = sc->base_reward_per_increment;
    // get_block_root_at_slot(state, s) for the slots pe:726 admits, and get_block_root(state, epoch) for the two target
    // epochs: the state's chain is the ancestry of chain_tip_root; one walk down from the tip serves them all
    {
        uint32_t cur = tip;
        for (uint64_t j = spe; j-- > 0;) {
            if (S.slot + j < spe) { S.head_blk[j] = NONE32; continue; }  // before slot 0
            cur = get_ancestor(h, cur, S.slot + j - spe);
            S.head_blk[j] = cur;
        }
        S.tgt_blk[0] = get_ancestor(h, tip, S.cur_epoch * spe);
        S.tgt_blk[1] = get_ancestor(h, tip, S.prev_epoch * spe);
    }
    OutBlock ob(h);
    const size_t off_err = ob.alloc(16);  // in front of the cap-sized arrays: nothing sized by the caller can reach it
    const size_t off_status = ob.alloc(4ull * cap);
    const size_t off_num = ob.alloc(8ull * cap);
    PE_TRY(ob.ensure());
    *ob.host<uint32_t>(off_err) = 0;
    memset(ob.host<int32_t>(off_status), 0, 4ull * cap);
    memset(ob.host<uint64_t>(off_num), 0, 8ull * cap);  // k_participation_tables writes the rows it runs
    const uint32_t n_launch = std::max<uint32_t>(h->rr.n_in, 1);
    hipStream_t ss = state_stream_begin(h, /*reads_scratch=*/true, n_launch);  // behind the unions and the plan; beside the next step's fork-choice chain
    {
        ProfScope ps(h, PE_KERNEL_ATT_VALIDATE, ss);  // timeline mode only
        launch_att_validate_state(ss, h->rr.rows, L.grp, L.plan, n_launch, cap, block_table_dev(h), S,
                                  RS.info.as<uint32_t>(), L.rows_st, L.status_st, ob.host<int32_t>(off_status),
                                  ob.host<uint32_t>(off_err));
    }
    {
        ProfScope ps(h, PE_KERNEL_PARTICIPATION, ss);
        launch_participation_tables(ss, L.rows_st, h->rr.tables, L.crow_start, L.crow_list, L.plan,
                                    RS.bits.as<uint32_t>(), h->d_incr.as<uint16_t>(), sc->base_reward_per_increment,
                                    h->d_part_cur.as<uint32_t>(), h->d_part_prev.as<uint32_t>(), ob.host<uint64_t>(off_num),
                                    reinterpret_cast<const uint32_t*>(L.status_st), cap);
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("rproc.1_launch");
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_status, off_num, off_err, cap, status, out_numerators]() -> int {
        const uint8_t* pin = h->arena[ai].h_pin.as<uint8_t>() + base;
        const uint32_t err = *reinterpret_cast<const uint32_t*>(pin + off_err);
        if (err) return plan_error_to_status(h, err, "pe_process_attestation_batch (PE_ROWS_RESIDENT)");
        memcpy(status, pin + off_status, 4ull * cap);
        memcpy(out_numerators, pin + off_num, 8ull * cap);
        return PE_OK;
    };
    HostLap lap2(&h->trace);
    const int rc = finish_call(h, st, ob, complete);
    lap2.mark("rproc.2_wait_outputs");
    return rc;
}

}  // namespace posevo
