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
        for (const char* n : names)
            if ((x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // one already in the process (torch's)
        if (!x.lib)
            for (const char* n : names)
                if ((x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!x.lib) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.lib, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
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

int pe_dist_init(pe_engine* h, const uint8_t id[PE_DIST_ID_BYTES], int rank, int world)
{
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (!rccl().ok) return fail(h, PE_ERR_NO_DEVICE, "librccl could not be loaded (dlopen librccl.so.1)");
    if (h->comm) return fail(h, PE_ERR_STATE, "pe_dist_init: this handle already has a communicator");
    // Two communicators: the all-reduce of get_head travels on the engine's stream, the all-gather of the G1
    // partials on the finishing stream of the G1 chain (beside the next step's fork-choice kernels).  One communicator
    // must not be driven from two streams at once; two of them may.
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    RCCL_TRY(h, rccl().CommInitRank(&h->comm, world, uid, rank));
    memcpy(&uid, id + sizeof(uid), sizeof(uid));
    ncclResult_t r2 = rccl().CommInitRank(&h->comm_g1, world, uid, rank);
    if (r2 != ncclSuccess) {
        (void)rccl().CommDestroy(h->comm);
        h->comm = nullptr;
        h->comm_g1 = nullptr;
        return rccl_fail(h, r2, "ncclCommInitRank (aggregation communicator)");
    }
    h->dist_rank = rank;
    h->dist_world = world;
    return PE_OK;
}

int pe_dist_destroy(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (h->comm) {
        (void)hipStreamSynchronize(h->stream);
        if (h->fin_stream) (void)hipStreamSynchronize(h->fin_stream);
        if (h->norm_stream) (void)hipStreamSynchronize(h->norm_stream);
        (void)rccl().CommDestroy(h->comm);
        if (h->comm_g1) (void)rccl().CommDestroy(h->comm_g1);
        h->comm = h->comm_g1 = nullptr;
    }
    h->dist_world = 1;
    h->dist_rank = 0;
    return PE_OK;
}

// get_head over all shards: this shard's direct weights -> ONE all-reduce(sum, u64) of B + PE_EXCHANGE_EXTRA words on
// the engine's stream -> subtree sums + descent on every rank (same root everywhere; integer sums are order-free).
int pe_get_head_sharded(pe_engine* h, uint8_t out_root[32])
{
    // inside a pipeline the call is ordered behind the enqueued batch calls on the stream, like pe_get_head
    int rc = need_init(h, /*flush=*/!(h && h->pipelining));
    if (rc) return rc;
    if (!out_root) return PE_ERR_INVALID_ARG;
    if (!h->comm) return fail(h, PE_ERR_STATE, "pe_get_head_sharded: call pe_dist_init first");
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
    uint64_t* buf = h->d_xchg.as<uint64_t>();
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
    RCCL_TRY(h, rccl().AllReduce(h->d_xchg.p, h->d_xchg.p, words, ncclUint64, ncclSum, h->comm, h->stream));
    lap.mark("dist.all_reduce_enqueue");
    uint32_t head;
    rc = run_tree(h, buf, reinterpret_cast<const VoteTotals*>(buf + nb), /*clear_direct=*/1, &head);
    if (rc == PE_OK) memcpy(out_root, h->blocks[head].root.data(), 32);
    lap.mark("dist.tree_wait");
    return rc;
}

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
    if (!h->comm) return fail(h, PE_ERR_STATE, "pe_aggregate_sharded: call pe_dist_init first");
    if (n == 0) { *out_n_groups = 0; return PE_OK; }
    PE_TRY(ensure_quiesced(h, h->d_xpart, (size_t)PE_G1_PARTIAL_BYTES * n));
    PE_TRY(ensure_quiesced(h, h->d_xgather, (size_t)PE_G1_PARTIAL_BYTES * n * (size_t)h->dist_world));
    int rc = aggregate_impl(h, atts, n, bits_arena, arena_len, nullptr, out_atts, out_n_groups, group_of, out_bits_arena,
                            out_arena_cap, nullptr, nullptr, out_count, h->d_xpart.p, n, /*partials_may_defer=*/true);
    if (rc) return rc;
    const uint32_t ng = *out_n_groups;
    if (ng == 0) return PE_OK;
    Stage st(h);
    OutBlock ob(h);
    const size_t off_pk = ob.alloc(96ull * ng);
    PE_TRY(ob.ensure());
    uint8_t* pin_pk = ob.host<uint8_t>(off_pk);
    // all-gather of the ranks' partials, then the finishing add + normalisation, written straight into the pinned
    // block.  In a streaming pipeline the partials' kernels were deferred behind the step's fork-choice kernels; the
    // exchange follows them (every rank runs the same calls, so the collectives are issued in the same order everywhere)
    const bool on_side = h->last_agg_on_side;
    auto exchange = [h, ng, pin_pk, on_side]() -> int {
        HostLap lap(&h->trace);
        hipStream_t xs = on_side ? h->g1_tail() : h->stream;  // where this aggregate's partials were produced
        RCCL_TRY(h, rccl().AllGather(h->d_xpart.p, h->d_xgather.p, (size_t)ng * (PE_G1_PARTIAL_BYTES / 4), ncclUint32,
                                     h->comm_g1, xs));
        lap.mark("dist.all_gather_enqueue");
        {
            ProfScope ps(h, PE_KERNEL_G1_NORMALISE, xs);
            launch_g1_finish(xs, h->d_xgather.as<uint32_t>(), nullptr, ng, (uint32_t)h->dist_world, ng, pin_pk, nullptr);
        }
        HIP_TRY(h, hipGetLastError());
        if (on_side) HIP_TRY(h, hipEventRecord(h->ev_join, xs));  // the end of this arena's G1 chain moved
        return PE_OK;
    };
    if (!h->deferred.empty()) h->deferred.push_back(exchange);
    else PE_TRY(exchange());
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_pk, ng, out_aggpk96]() -> int {
        memcpy(out_aggpk96, h->arena[ai].h_pin.as<uint8_t>() + base + off_pk, 96ull * ng);
        return PE_OK;
    };
    return finish_call(h, st, ob, complete);
}

}  // extern "C"
