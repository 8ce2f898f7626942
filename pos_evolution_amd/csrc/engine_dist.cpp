// engine_dist.cpp -- multi-GPU exchange behind the C ABI: validator-range shards, one all-reduce of
// per-block weights, one all-gather of per-committee XYZZ partials.
#include "engine_internal.h"

using namespace posevo;

namespace posevo {
Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // POSEVO_RCCL_PATH: the library to load instead (a deployment's own build; tests/native/stub_rccl.cpp, with which two
        // processes sharing one GPU drive this file's RCCL path -- real RCCL refuses two ranks on one device)
        if (const char* p = getenv("POSEVO_RCCL_PATH"))
            if (*p) x.lib = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!x.lib)
        for (const char* n : names)
            if ((x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // one already in the process (torch's)
        if (!x.lib)
            for (const char* n : names)
                if ((x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!x.lib) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.lib, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(dlsym(x.lib, "ncclCommAbort"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.lib, "ncclAllReduce"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.lib, "ncclAllGather"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.lib, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.lib, "ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}
int rccl_fail(pe_engine* h, ncclResult_t r, const char* what)
{
    const char* msg = rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error";
    return fail(h, PE_ERR_NO_DEVICE, std::string(what) + ": " + msg);
}
#define RCCL_TRY(h, expr)                                          \
    do {                                                           \
        ncclResult_t _r = (expr);                                  \
        if (_r != ncclSuccess) return rccl_fail((h), _r, #expr);   \
    } while (0)
static_assert(2 * sizeof(ncclUniqueId) == PE_DIST_ID_BYTES, "PE_DIST_ID_BYTES must hold two ncclUniqueIds");

// ---------------------------------------------------------------- the two exchange steps
int dist_all_reduce_u64(pe_engine* h, void* dev_buf, size_t count, hipStream_t s)
{
    if (h->dist_wedged) return fail(h, PE_ERR_TIMEOUT, "an earlier exchange timed out on this handle: pe_dist_destroy, then initialise again");
    if (h->coll_custom) {
        if (h->coll.all_reduce_u64(h->coll.user, dev_buf, count, s) != 0)
            return fail(h, PE_ERR_NO_DEVICE, "the caller's all_reduce_u64 failed");
        return PE_OK;
    }
    RCCL_TRY(h, rccl().AllReduce(dev_buf, dev_buf, count, ncclUint64, ncclSum, h->comm, s));
    return PE_OK;
}
int dist_all_gather(pe_engine* h, const void* send, void* recv, size_t bytes_per_rank, hipStream_t s)
{
    if (h->dist_wedged) return fail(h, PE_ERR_TIMEOUT, "an earlier exchange timed out on this handle: pe_dist_destroy, then initialise again");
    if (h->coll_custom) {
        if (h->coll.all_gather(h->coll.user, send, recv, bytes_per_rank, s) != 0)
            return fail(h, PE_ERR_NO_DEVICE, "the caller's all_gather failed");
        return PE_OK;
    }
    // its own communicator when it travels on the G1 chain's stream beside the engine stream's all-reduce
    ncclComm_t c = (s == h->stream || h->dist_single_comm) ? h->comm : h->comm_g1;
    RCCL_TRY(h, rccl().AllGather(send, recv, bytes_per_rank / 4, ncclUint32, c, s));
    return PE_OK;
}

// ---------------------------------------------------------------- bounded waits
// A rank whose peer never arrives would wait for ever inside hipEventSynchronize with a collective kernel spinning on
// the device.  On a handle that exchanges with other ranks the waits poll instead and give up after dist_timeout_ms:
// the communicators are aborted (their kernels leave the device), the handle is marked, and the caller gets an error it
// can act on -- every rank of the job sees the same thing, because every rank waits for the same exchange.
static void dist_abort(pe_engine* h)
{
    h->dist_wedged = true;
    if (h->coll_custom || !rccl().ok || !rccl().CommAbort) return;
    if (h->comm) (void)rccl().CommAbort(h->comm);
    if (h->comm_g1 && h->comm_g1 != h->comm) (void)rccl().CommAbort(h->comm_g1);
    h->comm = h->comm_g1 = nullptr;
}
template <typename Query>
static hipError_t bounded_wait(pe_engine* h, Query query)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 0;; ++it) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();
        if ((it & 255) == 255) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > (double)h->dist_timeout_ms) {
                dist_abort(h);
                return hipErrorNotReady;
            }
            if (ms > 2.0) std::this_thread::yield();  // long waits (a drain) need not burn the core
        }
    }
}
hipError_t bounded_event_sync(pe_engine* h, hipEvent_t ev)
{
    if (!h->dist_ready() || h->dist_timeout_ms == 0 || h->dist_world < 2) return hipEventSynchronize(ev);
    return bounded_wait(h, [ev] { return hipEventQuery(ev); });
}
hipError_t bounded_stream_sync(pe_engine* h, hipStream_t s)
{
    if (!h->dist_ready() || h->dist_timeout_ms == 0 || h->dist_world < 2) return hipStreamSynchronize(s);
    return bounded_wait(h, [s] { return hipStreamQuery(s); });
}
}  // namespace posevo

extern "C" {

// ---------------------------------------------------------------- RCCL inside the C ABI
// librccl is resolved at run time (dlopen): a process that already carries one -- torch ships its own -- shares it, a
// plain C / Go / Rust client gets the ROCm installation's.  No link-time dependency: single-GPU users never load it.
int pe_dist_unique_id(uint8_t out_id[PE_DIST_ID_BYTES])
{
    if (!out_id) return PE_ERR_INVALID_ARG;
    if (!rccl().ok) return PE_ERR_NO_DEVICE;
    for (int k = 0; k < 2; ++k) {  // one communicator per stream that carries collectives (see pe_dist_init)
        ncclUniqueId id;
        if (rccl().GetUniqueId(&id) != ncclSuccess) return PE_ERR_NO_DEVICE;
        memcpy(out_id + k * sizeof(id), &id, sizeof(id));
    }
    return PE_OK;
}

int pe_dist_init_ex(pe_engine* h, const uint8_t id[PE_DIST_ID_BYTES], int rank, int world, uint32_t flags)
{
    if (!h || !id || world < 1 || rank < 0 || rank >= world || (flags & ~PE_DIST_SINGLE_COMM)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (!rccl().ok) return fail(h, PE_ERR_NO_DEVICE, "librccl could not be loaded (dlopen librccl.so.1)");
    if (h->dist_ready()) return fail(h, PE_ERR_STATE, "pe_dist_init: this handle already has a communicator");
    // Two communicators: the all-reduce of get_head travels on the engine's stream, the all-gather of the G1
    // partials on the finishing stream of the G1 chain (beside the next step's fork-choice kernels).  One communicator
    // must not be driven from two streams at once; two of them may.  PE_DIST_SINGLE_COMM: one communicator, and the
    // all-gather comes back to the engine's stream (see pe_aggregate_sharded).
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    RCCL_TRY(h, rccl().CommInitRank(&h->comm, world, uid, rank));
    h->dist_single_comm = (flags & PE_DIST_SINGLE_COMM) != 0;
    if (h->dist_single_comm) {
        h->comm_g1 = h->comm;
    } else {
        memcpy(&uid, id + sizeof(uid), sizeof(uid));
        ncclResult_t r2 = rccl().CommInitRank(&h->comm_g1, world, uid, rank);
        if (r2 != ncclSuccess) {
            (void)rccl().CommDestroy(h->comm);
            h->comm = nullptr;
            h->comm_g1 = nullptr;
            return rccl_fail(h, r2, "ncclCommInitRank (aggregation communicator)");
        }
    }
    if (!h->ev_xchg) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_xchg, hipEventDisableTiming));
    h->dist_rank = rank;
    h->dist_world = world;
    h->dist_wedged = false;
    if (const char* e = getenv("POSEVO_DIST_TIMEOUT_MS")) h->dist_timeout_ms = (uint32_t)atol(e);
    return PE_OK;
}
int pe_dist_init(pe_engine* h, const uint8_t id[PE_DIST_ID_BYTES], int rank, int world)
{
    const char* e = getenv("POSEVO_DIST_SINGLE_COMM");
    return pe_dist_init_ex(h, id, rank, world, e && atoi(e) != 0 ? PE_DIST_SINGLE_COMM : 0u);
}

int pe_dist_init_custom(pe_engine* h, int rank, int world, const pe_collectives* fn)
{
    if (!h || !fn || !fn->all_reduce_u64 || !fn->all_gather || world < 1 || rank < 0 || rank >= world) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (h->dist_ready()) return fail(h, PE_ERR_STATE, "pe_dist_init_custom: this handle already has a communicator");
    h->coll = *fn;
    h->coll_custom = true;
    h->dist_single_comm = false;
    h->dist_rank = rank;
    h->dist_world = world;
    h->dist_wedged = false;
    if (const char* e = getenv("POSEVO_DIST_TIMEOUT_MS")) h->dist_timeout_ms = (uint32_t)atol(e);
    return PE_OK;
}

int pe_dist_set_timeout_ms(pe_engine* h, uint32_t ms)
{
    if (!h) return PE_ERR_INVALID_ARG;
    h->dist_timeout_ms = ms;
    return PE_OK;
}
int pe_dist_set_max_groups(pe_engine* h, uint32_t max_groups)
{
    if (!h) return PE_ERR_INVALID_ARG;
    h->dist_max_groups = max_groups;
    return PE_OK;
}

int pe_dist_destroy(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    const int rc = h->dist_wedged ? PE_OK : enter(h);  // after a timeout the enqueued work is lost with the communicators
    if (h->dist_wedged) {
        (void)hipDeviceSynchronize();  // aborted collectives have left the device
        for (auto& a : h->arena) {
            a.pending.clear();
            a.stage_cursor = a.out_cursor = 0;
            a.fenced = a.side_used = a.aux_used = a.aux_reads_scratch = false;
            a.fence_pending = a.leg_used = false;
        }
        h->deferred.clear();
        // launches held back for the next aggregate (engine_pair.cpp) point into the regions reset above and carry the lost
        // communicator's collectives: they go with the rest
        h->held = pe_engine::HeldFc{};
        h->sig_batch.clear();
        h->pipelining = h->streaming = false;
        h->side_busy = h->aux_busy = false;
        h->res_valid = false;
        h->rr.valid = false;
        h->early_rc = PE_OK;
        h->xchg_blocks = 0;  // what an aborted exchange left in the self-cleaning buffer is unknown: zeroed at the next use
    }
    if (h->comm) {
        (void)hipStreamSynchronize(h->stream);
        for (hipStream_t s : {h->fin_stream, h->norm_stream})
            if (s) (void)hipStreamSynchronize(s);
        (void)rccl().CommDestroy(h->comm);
        if (h->comm_g1 && h->comm_g1 != h->comm) (void)rccl().CommDestroy(h->comm_g1);
    }
    h->comm = h->comm_g1 = nullptr;
    h->coll_custom = false;
    h->dist_wedged = false;
    h->dist_world = 1;
    h->dist_rank = 0;
    return rc;
}

// get_head over all shards: this shard's direct weights -> ONE all-reduce(sum, u64) of B + PE_EXCHANGE_EXTRA words on
// the engine's stream -> subtree sums + descent on every rank (same root everywhere; integer sums are order-free).
static int get_head_sharded_impl(pe_engine* h, uint8_t out_root[32], bool async)
{
    // inside a pipeline the call is ordered behind the enqueued batch calls on the stream, like pe_get_head
    int rc = need_init(h, /*flush=*/!(h && h->pipelining), /*keep_held=*/async && h && h->pipelining);
    if (rc) return rc;
    if (!out_root) return PE_ERR_INVALID_ARG;
    // launches held back by an EARLIER pipeline, or a head already held for this one: out first, in order
    if (h->held.active && (h->held.arena != h->cur || h->held.have_head)) PE_TRY(held_issue(h));
    if (!h->dist_ready()) return fail(h, PE_ERR_STATE, "pe_get_head_sharded: call pe_dist_init first");
    if (h->dist_wedged)  // before anything is launched: k_votes / the unions would otherwise run with no exchange to follow
        return fail(h, PE_ERR_TIMEOUT, "an earlier exchange timed out on this handle: pe_dist_destroy, then initialise again");
    if (!h->pipelining) async = false;  // outside a pipeline every call is synchronous
    const uint32_t nb = (uint32_t)h->blocks.size();
    const size_t words = (size_t)nb + PE_EXCHANGE_EXTRA;
    if (words * 8 > h->d_xchg.cap) {  // the engine's own exchange buffer is self-cleaning, like pe_get_head's: zero it once
        PE_TRY(ensure_quiesced(h, h->d_xchg, words * 8));
        HIP_TRY(h, hipMemsetAsync(h->d_xchg.p, 0, h->d_xchg.cap, h->stream));
        h->xchg_blocks = 0;
    }
    if (h->xchg_blocks != nb || h->xchg_nval != h->n_val) {
        // the block count changed (the totals sit at a different offset now) or the registry did (a different number
        // of k_votes workgroups store totals; slots none of them writes must read zero)
        HIP_TRY(h, hipMemsetAsync(h->d_xchg.p, 0, h->d_xchg.cap, h->stream));
        h->xchg_blocks = nb;
        h->xchg_nval = h->n_val;
    }
    HostLap lap(&h->trace);
    rc = refresh_tree(h);
    if (rc) return rc;
    {
        uint32_t tmp;  // fail before anything is launched: k_votes adds into a buffer only k_tree clears
        if (!find_block(h, h->justified.root, &tmp))
            return fail(h, PE_ERR_UNKNOWN_ROOT, "justified checkpoint root is not in the store");
    }
    Stage st(h);
    OutBlock ob(h);
    size_t off = 0;
    if (async) {
        off = ob.alloc(64);
        PE_TRY(ob.ensure());
    }
    uint64_t* buf = h->d_xchg.as<uint64_t>();
    if (async && hold_eligible(h) && h->n_val) {
        // a streaming step: votes -> all-reduce -> tree go out with the NEXT aggregate's row kernels (engine_pair.cpp); every
        // rank holds and issues alike, so the all-reduce keeps its place in the order of this communicator's collectives
        PE_TRY(tree_args(h, buf, reinterpret_cast<const VoteTotals*>(buf + nb), /*clear_direct=*/1, ob.host<uint32_t>(off),
                         &h->held.tree));
        h->held.votes = VotesArgs{h->d_vote_block.as<uint32_t>(), h->d_balance.as<uint64_t>(), h->d_flags.as<uint8_t>(),
                                  h->n_val, h->cfg.filter_slashed, h->d_tpos.as<uint32_t>(), nb, buf,
                                  reinterpret_cast<VoteTotals*>(buf + nb), expiry_slots_ptr(h), min_vote_slot(h)};
        h->held.between = [h, words]() -> int { return dist_all_reduce_u64(h, h->d_xchg.p, words, h->stream); };
        h->held.have_head = true;
        h->held.active = true;
        h->held.arena = h->cur;
        const size_t base = ob.base;
        const int ai = h->cur;
        auto complete = [h, ai, base, off, out_root]() -> int {
            const uint32_t idx = *reinterpret_cast<const uint32_t*>(h->arena[ai].h_pin.as<uint8_t>() + base + off);
            if (idx >= h->blocks.size()) return fail(h, PE_ERR_NO_DEVICE, "tree kernel returned an invalid head index");
            memcpy(out_root, h->blocks[idx].root.data(), 32);
            return PE_OK;
        };
        return finish_call(h, st, ob, complete);
    }
    if (h->held.active) PE_TRY(held_issue(h));
    {
        ProfScope ps(h, PE_KERNEL_VOTES);
        // no memsets (k_tree zeroes the weights it read; the totals are plain per-workgroup stores), and inside a
        // pipeline the lean form that fits beside a running k_g1_accumulate
        launch_votes(h->stream, h->d_vote_block.as<uint32_t>(), h->d_balance.as<uint64_t>(), h->d_flags.as<uint8_t>(),
                     h->n_val, h->cfg.filter_slashed, h->d_tpos.as<uint32_t>(), nb, buf,
                     reinterpret_cast<VoteTotals*>(buf + nb), 0, expiry_slots_ptr(h), min_vote_slot(h),
                     /*lean=*/h->pipelining ? 1 : 0);
    }
    HIP_TRY(h, hipGetLastError());
    lap.mark("dist.votes_launch");
    PE_TRY(dist_all_reduce_u64(h, h->d_xchg.p, words, h->stream));
    lap.mark("dist.all_reduce_enqueue");
    uint32_t head = 0;
    rc = run_tree(h, buf, reinterpret_cast<const VoteTotals*>(buf + nb), /*clear_direct=*/1, &head,
                  async ? ob.host<uint32_t>(off) : nullptr);
    lap.mark("dist.tree_wait");
    if (rc) return rc;
    if (!async) {
        memcpy(out_root, h->blocks[head].root.data(), 32);
        return PE_OK;
    }
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off, out_root]() -> int {
        const uint32_t idx = *reinterpret_cast<const uint32_t*>(h->arena[ai].h_pin.as<uint8_t>() + base + off);
        if (idx >= h->blocks.size()) return fail(h, PE_ERR_NO_DEVICE, "tree kernel returned an invalid head index");
        memcpy(out_root, h->blocks[idx].root.data(), 32);
        return PE_OK;
    };
    return finish_call(h, st, ob, complete);
}

int pe_get_head_sharded(pe_engine* h, uint8_t out_root[32]) { return get_head_sharded_impl(h, out_root, false); }
// the root delivered like every other output of the pipeline (pe_get_head_async's counterpart): a streaming caller's
// loop then never waits for the all-reduce inside a step
int pe_get_head_sharded_async(pe_engine* h, uint8_t out_root[32]) { return get_head_sharded_impl(h, out_root, true); }

// pe_aggregate over all shards: rank-local bitfield unions, global aggregate pubkeys.  Every rank passes attestations
// that form the SAME groups in the SAME order (group g of every rank = that rank's members of committee g); the
// XYZZ partials (192 B per group) are all-gathered and every rank runs the finishing add + normalisation.
// Inside a pipeline nothing here waits: kernels, the all-gather and the finish are enqueued on the engine's stream, the
// unions stay resident for PE_BITS_RESIDENT hand-over, and the outputs are complete at pe_pipeline_end.
int pe_aggregate_sharded(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                         uint64_t arena_len, pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                         uint8_t* out_bits_arena, uint64_t out_arena_cap, uint8_t* out_aggpk96, uint32_t* out_count)
{
    if (!h || !out_n_groups || !out_aggpk96) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    if (!h->pipelining) PE_TRY(flush_pending(h));
    if (!h->dist_ready()) return fail(h, PE_ERR_STATE, "pe_aggregate_sharded: call pe_dist_init first");
    if (h->dist_wedged)  // before anything is launched: k_votes / the unions would otherwise run with no exchange to follow
        return fail(h, PE_ERR_TIMEOUT, "an earlier exchange timed out on this handle: pe_dist_destroy, then initialise again");
    if (n == 0) { *out_n_groups = 0; return PE_OK; }
    const bool dev_rows = rows_on_device(atts);
    // slots of the exchange: host rows -- the groups the host formed; rows in device memory -- the caller's bound (the
    // collective is sized before the device has formed the groups)
    const uint32_t slots_bound = dev_rows && h->dist_max_groups ? std::min(n, h->dist_max_groups) : n;
    PE_TRY(ensure_quiesced(h, h->d_xpart, (size_t)PE_G1_PARTIAL_BYTES * n));
    PE_TRY(ensure_quiesced(h, h->d_xgather, (size_t)PE_G1_PARTIAL_BYTES * n * (size_t)h->dist_world));
    int rc;
    if (dev_rows)
        rc = aggregate_resident(h, atts, n, bits_arena, arena_len, out_atts, out_n_groups, group_of, out_bits_arena,
                                out_arena_cap, nullptr, out_count, h->d_xpart.as<uint32_t>());
    else
        rc = aggregate_impl(h, atts, n, bits_arena, arena_len, nullptr, out_atts, out_n_groups, group_of, out_bits_arena,
                            out_arena_cap, nullptr, nullptr, out_count, h->d_xpart.p, n, /*partials_may_defer=*/true);
    if (rc) return rc;
    const uint32_t slots = dev_rows ? slots_bound : *out_n_groups;
    if (slots == 0) return PE_OK;
    Stage st(h);
    OutBlock ob(h);
    const size_t off_pk = ob.alloc(96ull * slots);
    PE_TRY(ob.ensure());
    uint8_t* pin_pk = ob.host<uint8_t>(off_pk);
    // all-gather of the ranks' partials, then the finishing add + normalisation, written straight into the pinned
    // block.  In a streaming pipeline the partials' kernels were deferred behind the step's fork-choice kernels; the
    // exchange follows them (every rank runs the same calls, so the collectives are issued in the same order everywhere)
    const bool on_side = h->last_agg_on_side;
    const AttPlan* plan_dev = nullptr;
    if (dev_rows) {  // the finish reads the number of groups where k_att_plan left it
        const AttPlan* p = nullptr;
        PE_TRY(resident_plan_dev(h, &p));
        plan_dev = p;
    }
    pe_engine::PipeArena* arena = &h->A();  // the closure may run while a LATER pipeline is the current one (held launches)
    auto exchange = [h, arena, slots, pin_pk, on_side, plan_dev]() -> int {
        HostLap lap(&h->trace);
        hipStream_t cs = on_side ? h->g1_tail() : h->stream;  // where this aggregate's partials were produced
        hipStream_t xs = cs;
        if (on_side && h->dist_single_comm) {  // one communicator: the all-gather travels on the engine's stream
            HIP_TRY(h, hipEventRecord(h->ev_xchg, cs));
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_xchg, 0));
            xs = h->stream;
        }
        PE_TRY(dist_all_gather(h, h->d_xpart.p, h->d_xgather.p, (size_t)slots * PE_G1_PARTIAL_BYTES, xs));
        lap.mark("dist.all_gather_enqueue");
        if (xs != cs) {  // ... and the finish goes back to the chain's stream
            HIP_TRY(h, hipEventRecord(h->ev_xchg, xs));
            HIP_TRY(h, hipStreamWaitEvent(cs, h->ev_xchg, 0));
        }
        {
            ProfScope ps(h, PE_KERNEL_G1_NORMALISE, cs);
            launch_g1_finish(cs, h->d_xgather.as<uint32_t>(), nullptr, slots, (uint32_t)h->dist_world, slots, pin_pk, nullptr,
                             plan_dev);
        }
        HIP_TRY(h, hipGetLastError());
        if (on_side) {
            HIP_TRY(h, hipEventRecord(h->ev_join, cs));  // the end of this arena's G1 chain moved
            // ... and the arena's completion has to wait for it: a block that grew between the aggregate and this
            // exchange completed the arena (and cleared these marks) with the finish not yet enqueued
            h->side_busy = true;
            h->side_ever = true;
            arena->side_used = true;
        }
        return PE_OK;
    };
    if (!h->deferred.empty()) h->deferred.push_back(exchange);
    else PE_TRY(exchange());
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_pk, slots, out_aggpk96, out_n_groups, dev_rows]() -> int {
        // rows in device memory: the group count is an output, set by the aggregate's own completion (which ran just
        // before this one); host rows: it was known at the call (and *out_n_groups need not outlive it)
        const uint32_t ng = dev_rows ? *out_n_groups : slots;
        if (ng > slots)
            return fail(h, PE_ERR_CAPACITY, "pe_aggregate_sharded: the rows formed more groups than pe_dist_set_max_groups allows");
        memcpy(out_aggpk96, h->arena[ai].h_pin.as<uint8_t>() + base + off_pk, 96ull * ng);
        return PE_OK;
    };
    return finish_call(h, st, ob, complete);
}

// ---------------------------------------------------------------- committee-sharded steps (Option B)
// "Parallelising the aggregation of attestations" (pe:474) along its natural axis: the committees.  Rank g is handed only
// the rows of the committees it serves (its subnets) and sums THEIR aggregate pubkeys over a replicated registry -- no G1
// collective, a rank's grouping / union / G1 work is 1/N of the epoch's.  What the other ranks need of the result is the
// aggregate attestation itself (pe:714-717: data + OR-ed bits, the object a validator client publishes, pe:659); one
// all-gather carries every rank's aggregates to every rank, which ingests them as ONE dense batch: the handlers behind
// (PE_ROWS_RESIDENT) then apply the whole epoch's votes and flags to the rank's full copy of the store, and get_head is
// the plain pe_get_head -- no weight exchange, every rank reaches the same head from the same latest messages.
//   pe_aggregate(local device rows, ... out_aggpk96 ...)   -- this rank's committees: unions + aggregate pubkeys
//   pe_aggregate_exchange(...)                              -- pack -> all-gather -> the gathered aggregates become the
//                                                              resident aggregate the handlers run over
// The G1 chain of the local aggregate keeps running beside the exchange (the gathered batch has scratch of its own).
int pe_aggregate_exchange(pe_engine* h, pe_attestation* out_atts, uint32_t* out_n_groups, uint8_t* out_bits_arena,
                          uint64_t out_arena_cap, uint32_t* out_count, uint32_t cap_groups)
{
    if (!h || !out_atts || !out_n_groups || !out_bits_arena) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    if (!h->pipelining) PE_TRY(flush_pending(h));
    if (!h->dist_ready()) return fail(h, PE_ERR_STATE, "pe_aggregate_exchange: call pe_dist_init first");
    if (h->dist_wedged)  // before anything is launched: k_votes / the unions would otherwise run with no exchange to follow
        return fail(h, PE_ERR_TIMEOUT, "an earlier exchange timed out on this handle: pe_dist_destroy, then initialise again");
    if (!h->rr.valid || h->rr.set != 0 || h->rr.arena != h->cur)
        return fail(h, PE_ERR_STATE, "pe_aggregate_exchange follows a pe_aggregate over rows in device memory (same pipeline)");
    ResidentParts P;
    PE_TRY(resident_parts(h, &P));
    const uint32_t world = (uint32_t)h->dist_world;
    // the all-gather's size follows from the slot count, which must be the SAME on every rank: the local row count is not
    // (ranks serve different committees), so the caller's bound is required here (without one, ranks issued
    // all-gathers of different sizes -- a hang or a corrupt unpack)
    if (!h->dist_max_groups)
        return fail(h, PE_ERR_STATE, "pe_aggregate_exchange: call pe_dist_set_max_groups first (the same bound on every rank)");
    const uint32_t slots = h->dist_max_groups;
    const uint32_t wps = (h->cfg.max_validators_per_committee + 31) / 32;  // words of one union
    const uint32_t n_bound = world * slots;
    if (cap_groups < n_bound)
        return fail(h, PE_ERR_CAPACITY, "pe_aggregate_exchange: the output arrays hold fewer than world x max_groups entries");
    const size_t rank_words = 4 + (size_t)slots * (38 + wps);
    pe_engine::PipeArena& A = h->A();
    PE_TRY(ensure_quiesced(h, A.d_xsend, rank_words * 4));
    PE_TRY(ensure_quiesced(h, A.d_xrecv, rank_words * 4 * world));
    PE_TRY(ensure_quiesced(h, A.d_xrows, (size_t)144 * n_bound + 64));
    PE_TRY(ensure_quiesced(h, A.d_xbits, (size_t)4 * wps * n_bound + 64));
    PE_TRY(ensure_quiesced(h, A.d_xn, 64));
    Stage st(h);
    OutBlock ob(h);
    const size_t off_err = ob.alloc(16);
    PE_TRY(ob.ensure());
    *ob.host<uint32_t>(off_err) = 0;
    hipStream_t ms = h->stream;
    launch_att_pack(ms, P.rows, P.grp, P.plan, P.res_bits, P.res_info, slots, wps, A.d_xsend.as<uint32_t>());
    HIP_TRY(h, hipGetLastError());
    {
        HostLap lap(&h->trace);
        PE_TRY(dist_all_gather(h, A.d_xsend.p, A.d_xrecv.p, rank_words * 4, ms));  // on the engine's stream: `comm`
        lap.mark("dist.exchange_all_gather_enqueue");
    }
    launch_att_unpack(ms, A.d_xrecv.as<uint32_t>(), world, slots, wps, A.d_xrows.p, A.d_xbits.as<uint32_t>(),
                      A.d_xn.as<uint32_t>(), ob.host<uint32_t>(off_err));
    HIP_TRY(h, hipGetLastError());
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_err]() -> int {
        const uint32_t err = *reinterpret_cast<const uint32_t*>(h->arena[ai].h_pin.as<uint8_t>() + base + off_err);
        if (err == 10) return fail(h, PE_ERR_CAPACITY, "pe_aggregate_exchange: a rank formed more groups than pe_dist_set_max_groups "
                                                        "allows, or a union longer than max_validators_per_committee bits");
        if (err) return fail(h, PE_ERR_INVALID_ARG, "pe_aggregate_exchange: the aggregate of a rank failed; nothing was applied");
        return PE_OK;
    };
    PE_TRY(finish_call(h, st, ob, complete));
    // the gathered aggregates as this rank's resident aggregate (scratch set 1; the row count is a device result)
    return aggregate_resident(h, static_cast<const pe_attestation*>(A.d_xrows.p), n_bound, A.d_xbits.as<uint8_t>(),
                              (uint64_t)4 * wps * n_bound, out_atts, out_n_groups, nullptr, out_bits_arena, out_arena_cap,
                              nullptr, out_count, nullptr, /*set=*/1, A.d_xn.as<uint32_t>());
}

}  // extern "C"
