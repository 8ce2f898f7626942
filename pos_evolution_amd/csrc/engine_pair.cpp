// engine_pair.cpp -- the fork-choice launches of a streaming step held back and paired with the next step's row kernels.
//
// A streaming caller (pe_pipeline_begin_streaming / _end_lagged over rows in device memory) runs, per step, on the engine's
// stream:  ingest -> plan -> members -> union   (pe_aggregate: grouping of the step's rows, pe:474 / pe:659)
//          validate -> LMD                      (pe_on_attestation_batch: A.4, pe:1435-1441)
//          votes -> tree                        (pe_get_head_async: A.1, pe:1102-1116)
// The first chain of step N + 1 needs nothing the other two of step N produce, and each is ~120 us of latency-sized
// kernels.  So the handlers of step N do not launch when they are called: their argument blocks are kept on the handle
// (pe_engine::held) and go out when step N + 1's pe_aggregate arrives, each one as a block range of the same grid as a row
// kernel of that aggregate (pair_kernels.hip).  The step's G1 sums (deferred as before) go out in front of the pairs -- they
// depend on the step's row chain only -- and the fence of its lagged pipeline behind the pair that carries its k_tree.
//
// Nothing changes for the caller: the per-function ABI, the order of effects on the store (the held kernels run in the
// same order on the same stream, merely later), the lag contract (a step's outputs are complete when the lag-th next
// pipeline has ended: the launches are out by the NEXT aggregate at the latest, or at any call that needs the stream --
// held_issue runs them alone, in order -- pe_pipeline_end, pe_get_head, every synchronous entry point).
// POSEVO_PAIR=0 disables the holding (A/B).
#include "engine_internal.h"

using namespace posevo;

namespace posevo {

// The calls of a streaming pipeline on the engine's own stream: only there is a next aggregate to wait for.  (Sharded steps
// hold too: every rank makes the same calls, so the collectives the held launches carry -- the weights' all-reduce between
// votes and tree, the partials' all-gather behind the step's G1 launch -- are issued in the same order on every rank.)
bool hold_eligible(const pe_engine* h)
{
    return h->pairing && h->streaming && h->pipelining && h->stream == h->own_stream && h->side_stream != nullptr &&
           !h->dist_wedged;
}

int fence_arena(pe_engine* h, pe_engine::PipeArena& a)
{
    HIP_TRY(h, hipEventRecord(a.ev_main, h->stream));
    if (a.side_used) HIP_TRY(h, hipEventRecord(a.ev_side, h->g1_tail()));  // the last kernel of the G1 chain runs there
    if (a.aux_used) HIP_TRY(h, hipEventRecord(a.ev_aux, h->aux_stream));    // flag pass, signature leg
    a.fence_pending = false;
    return PE_OK;
}

// The step's G1 sums (and what follows them on their streams: a sharded step's all-gather, a signature leg).  They need
// nothing of the step's fork-choice kernels -- only its row chain, a step old by now -- so they go out FIRST: the accumulation
// is what paces a streaming run, and queued here it starts the moment its predecessor ends instead of waiting for
// the pair that carries the step's k_tree (the second step of a run started 110 us late that way).
static int launch_held_g1(pe_engine* h, pe_engine::HeldFc& L)
{
    int rc = PE_OK;
    for (auto& f : L.g1) {
        const int r = f();
        if (r && !rc) rc = r;
    }
    L.g1.clear();
    (void)h;
    return rc;
}
// ... and behind the step's last kernel on the engine's stream: the fence of its pipeline, if that has ended meanwhile.
static int after_tree(pe_engine* h, pe_engine::HeldFc& L)
{
    int rc = PE_OK;
    if (L.fence_pending && L.arena >= 0) {
        pe_engine::PipeArena& a = h->arena[L.arena];
        const int r = fence_arena(h, a);
        if (r && !rc) rc = r;
        // that fence accounts for the old pipeline's G1 launch -- unless the CURRENT pipeline has put work on the side stream
        // already (held_issue is reachable at any point of a pipeline): then somebody still has to wait for that
        if (L.arena != h->cur && !h->A().side_used) h->side_busy = false;
    }
    return rc;
}

int held_issue(pe_engine* h)
{
    if (!h->held.active) return PE_OK;
    pe_engine::HeldFc L = std::move(h->held);
    h->held = pe_engine::HeldFc{};
    hipStream_t s = h->stream;
    int rc0 = launch_held_g1(h, L);
    if (L.have_fc) {
        {
            ProfScope ps(h, PE_KERNEL_ATT_VALIDATE);  // timeline mode only
            launch_att_validate_fc(s, L.validate);
        }
        ProfScope ps(h, PE_KERNEL_LMD);
        launch_lmd_vm_tables(s, L.lmd);
    }
    if (L.have_head) {
        {
            ProfScope ps(h, PE_KERNEL_VOTES);
            launch_votes(s, L.votes, /*lean=*/1);
        }
        int rb = PE_OK;
        if (L.between) { rb = L.between(); if (rb && !rc0) rc0 = rb; }
        ProfScope ps(h, PE_KERNEL_TREE);
        if (!rb) launch_tree(s, L.tree, /*lean=*/1);  // no tree over weights whose exchange failed
    }
    const hipError_t e = hipGetLastError();
    const int rc = after_tree(h, L);
    if (e != hipSuccess) return hip_fail(h, e, "launching the held-back fork-choice kernels");
    return rc0 ? rc0 : rc;
}

int launch_rows_paired(pe_engine* h, const IngestArgs& ia, const AttPlanArgs& pa, const MembersArgs& ma, const UnionArgs& ua)
{
    hipStream_t s = h->stream;
    if (!h->held.active) {  // nothing held: the row chain alone
        {
            ProfScope ps(h, PE_KERNEL_ATT_GROUP, s);  // timeline mode only: ingest + plan + members
            launch_att_ingest(s, ia);
            launch_att_plan(s, pa);
            launch_att_members(s, ma);
        }
        ProfScope ps(h, PE_KERNEL_BITS_UNION, s);
        launch_bits_union(s, ua);
        return PE_OK;
    }
    pe_engine::HeldFc L = std::move(h->held);
    h->held = pe_engine::HeldFc{};
    HostLap lap(&h->trace);
    int rc0 = launch_held_g1(h, L);
    lap.mark("pair.0_held_g1");
    // a pair without a common shape goes out as two launches (the two are independent: any order)
    if (L.have_fc) {
        {
            ProfScope ps(h, PE_KERNEL_PAIR_INGEST_VALIDATE, s);
            if (!launch_pair_ingest_validate(s, ia, L.validate)) {
                launch_att_validate_fc(s, L.validate);
                launch_att_ingest(s, ia);
            }
        }
        ProfScope ps(h, PE_KERNEL_PAIR_PLAN_LMD, s);
        if (!launch_pair_plan_lmd(s, pa, L.lmd)) {
            launch_lmd_vm_tables(s, L.lmd);
            launch_att_plan(s, pa);
        }
    } else {
        ProfScope ps(h, PE_KERNEL_ATT_GROUP, s);
        launch_att_ingest(s, ia);
        launch_att_plan(s, pa);
    }
    if (L.have_head) {
        {
            ProfScope ps(h, PE_KERNEL_PAIR_MEMBERS_VOTES, s);
            if (!launch_pair_members_votes(s, ma, L.votes)) {
                launch_votes(s, L.votes, /*lean=*/1);
                launch_att_members(s, ma);
            }
        }
        lap.mark("pair.1_three_pairs");
        int rb = PE_OK;
        if (L.between) { rb = L.between(); if (rb && !rc0) rc0 = rb; lap.mark("pair.2_between"); }
        ProfScope ps(h, PE_KERNEL_PAIR_UNION_TREE, s);
        if (rb) launch_bits_union(s, ua);  // no tree over weights whose exchange failed
        else if (!launch_pair_union_tree(s, ua, L.tree)) {
            launch_tree(s, L.tree, /*lean=*/1);
            launch_bits_union(s, ua);
        }
    } else {
        {
            ProfScope ps(h, PE_KERNEL_ATT_GROUP, s);
            launch_att_members(s, ma);
        }
        ProfScope ps(h, PE_KERNEL_BITS_UNION, s);
        launch_bits_union(s, ua);
    }
    const hipError_t e = hipGetLastError();
    lap.mark("pair.3_union_tree");
    const int rc = after_tree(h, L);
    if (e != hipSuccess) return hip_fail(h, e, "launching the paired kernels");
    lap.mark("pair.4_fence");
    if (h->streaming) complete_oldest_if_ready(h);  // as pe_get_head_async did behind its k_tree: the copy-out of the oldest
                                                    // pipeline, if the device is through with it
    return rc0 ? rc0 : rc;
}

}  // namespace posevo
