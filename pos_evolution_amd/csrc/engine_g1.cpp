// engine_g1.cpp -- G1 sums: planning + launch of the three-kernel sum (accumulate / tree / finish); plain G1 and G2
// sums over caller-chosen groups; the BLSPubkey / BLSSignature wire formats (pe:37, pe:717).
#include "engine_internal.h"

using namespace posevo;

namespace posevo {

// G1 work of one pe_aggregate may run on the side stream (pipelined calls): anything else that is about to use the
// shared G1 scratch (d_partials) on another stream first waits for it.
void g1_stream_guard(pe_engine* h, hipStream_t s)
{
    // ev_join marks the end of the last G1 launch on the side stream (a lagged pipeline may still be running it);
    // waiting on a completed event costs nothing
    if (h->side_ever && s != h->side_stream && s != h->fin_stream && s != h->norm_stream) (void)hipStreamWaitEvent(s, h->ev_join, 0);
}

// The registry table in the accumulation's field form (k_g1_table_s29), built where the registry is loaded -- on the
// engine's stream, which every G1 launch is ordered behind.  (Built lazily inside the first G1 launch -- round 3's S29
// experiment -- the allocation ran ensure_quiesced INSIDE a streaming pipeline's deferred launch: it completed the
// current arena, copying the aggregate pubkeys out before k_g1_finish had written them.)
int build_points29(pe_engine* h, uint64_t n)
{
    HIP_TRY(h, h->d_points29.ensure(std::max<size_t>(128, 4ull * G1_ROW_WORDS * n)));
    launch_g1_table_s29(h->stream, h->d_points.as<uint32_t>(), h->d_points29.as<uint32_t>(), n);
    HIP_TRY(h, hipGetLastError());
    h->points29_valid = true;
    return PE_OK;
}

// Launch accumulate + finish for device-resident descriptors.  No copies, no synchronisation.
// fin != s: the tree and finish kernels go to their own stream behind an event (a pipelined aggregate: they then overlap
// the next aggregate's accumulation); partials / lane_partials: the scratch the kernels hand over through (per arena
// when pipelined).  plan_dev: the plan lives on the device (rows resident there, engine_resident.cpp): `plan` then holds
// upper bounds that size grids and scratch; d_members1: member array of the groups of the second candidate table.
int launch_g1_planned(pe_engine* h, const uint32_t* d_points, const uint32_t* d_members, const uint32_t* d_bits,
                      const G1Group* d_groups, const G1Plan& plan, uint8_t* d_out96, uint32_t* dev_jac,
                      hipStream_t s, hipStream_t fin, DevBuf* partials, DevBuf* lane_partials, const AttPlan* plan_dev,
                      const uint32_t* d_members1, uint64_t caller_rows)
{
    if (plan.n_groups == 0) return PE_OK;
    if (!s) s = h->stream;
    if (!fin) fin = s;
    if (!partials) partials = &h->d_partials;
    if (!lane_partials) lane_partials = &h->d_lane_partials;
    PE_TRY(ensure_quiesced(h, *partials,
                           std::max<size_t>(PE_G1_PARTIAL_BYTES, (size_t)PE_G1_PARTIAL_BYTES * plan.n_partials)));
    const size_t lane_bytes = (size_t)G1_LANE_PARTIAL_BYTES * G1_WG * ((plan.n_slots + G1_WG - 1) / G1_WG);
    PE_TRY(ensure_quiesced(h, *lane_partials, std::max<size_t>(PE_G1_PARTIAL_BYTES, lane_bytes)));
    // the table in the accumulation's field form (S29): the registry's is built at the first use after the registry
    // changed; caller-supplied points (pe_g1_sum, the G1-flavoured signature leg: d_tmp_points) are converted per call
    const uint32_t* d_points29;
    if (d_points == h->d_points.as<uint32_t>()) {
        if (!h->points29_valid) return fail(h, PE_ERR_STATE, "the registry's table of the accumulation's field form was not built");
        d_points29 = h->d_points29.as<uint32_t>();
    } else {
        // caller_rows: the rows of d_tmp_points this call's conversion filled, handed down by the caller (not a
        // side channel on the handle, which another user of that scratch buffer would leave stale)
        const uint64_t n_rows = caller_rows;
        if (d_points != h->d_tmp_points.as<uint32_t>() || 4ull * G1_ROW_WORDS * n_rows > h->d_tmp_points.cap)
            return fail(h, PE_ERR_STATE, "G1 sum over an unknown point table");
        // sized like d_tmp_points, by its callers' rule: engine-owned scratch of calls that complete before they return (a
        // flush from here could complete the current arena in the middle of the call that is filling it)
        HIP_TRY(h, h->d_tmp_points29.ensure(std::max<size_t>(128, 4ull * G1_ROW_WORDS * n_rows)));
        launch_g1_table_s29(s, d_points, h->d_tmp_points29.as<uint32_t>(), n_rows);
        d_points29 = h->d_tmp_points29.as<uint32_t>();
    }
    // A side-stream chain of a streaming run: the accumulation may be made the only one of its kind on a CU (Tune::exclusive:
    // the LDS it asks for leaves 78 KB, enough for the fork-choice tree up to 4096 blocks beside it -- an 8192-block tree's
    // 147 KB workgroup would find no CU while accumulations follow each other, so those stores keep the old signature).
    const bool chain = fin != s && fin == h->fin_stream;
    const int exclusive = chain && h->tune.exclusive && h->blocks.size() <= 4096 ? 1 : 0;
    {
        // Since the paired launches (round 5) the accumulations of a streaming run follow each other on their stream with
        // nothing but queue packets in between, and every packet there is step time: the two event records of a bracket cost
        // ~9 us of a ~19 us gap.  Totals mode therefore brackets one launch in four (the average duration is a sample mean;
        // pe_profile_get's launch count says how many were measured); timeline mode brackets all.
        const bool skip = !h->prof_timeline && s != h->stream && (h->acc_launches++ & 3) != 0;
        ProfScope ps(h, PE_KERNEL_G1_ACCUMULATE, s, skip);
        unsigned long long* clock_rec = nullptr;
        if (h->profiling) {
            if (!h->acc_clock && hipHostMalloc(reinterpret_cast<void**>(&h->acc_clock), 16ull * pe_engine::ACC_CLOCK_RING,
                                               hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h->acc_clock = nullptr;
            }
            if (h->acc_clock) {
                clock_rec = h->acc_clock + 2 * (h->acc_clock_n++ % pe_engine::ACC_CLOCK_RING);
                clock_rec[0] = clock_rec[1] = 0;
            }
        }
        launch_g1_accumulate(s, d_points29, d_members, d_bits, d_groups, plan.n_groups, plan.n_slots,
                             lane_partials->as<uint32_t>(), partials->as<uint32_t>(), plan_dev, d_members1, exclusive, clock_rec);
    }
    hipStream_t ts = fin;  // (on the accumulation's own stream the tree measured 0.433 vs 0.338 ms per step, round 3)
    if (ts != s) {
        HIP_TRY(h, hipEventRecord(h->ev_acc, s));
        HIP_TRY(h, hipStreamWaitEvent(ts, h->ev_acc, 0));
    }
    {
        ProfScope ps(h, PE_KERNEL_G1_TREE, ts);
        launch_g1_tree(ts, lane_partials->as<uint32_t>(), d_groups, plan.n_groups, plan.n_slots, partials->as<uint32_t>(),
                       /*one_per_cu=*/ts != s ? 1 : 0,   // on its own stream it meets the next step's k_tree: leave it room
                       plan_dev, /*solo=*/exclusive);
    }
    hipStream_t ns = fin;
    if (chain && h->norm_stream) {  // the finish gets a stream of its own
        ns = h->norm_stream;
        HIP_TRY(h, hipEventRecord(h->ev_tree, ts));
        HIP_TRY(h, hipStreamWaitEvent(ns, h->ev_tree, 0));
    } else if (ts != fin) {
        HIP_TRY(h, hipEventRecord(h->ev_tree, ts));
        HIP_TRY(h, hipStreamWaitEvent(fin, h->ev_tree, 0));
    }
    {
        ProfScope ps(h, PE_KERNEL_G1_NORMALISE, ns);
        launch_g1_finish(ns, partials->as<uint32_t>(), d_groups, plan.n_groups, 0, 0, d_out96, dev_jac, plan_dev);
    }
    HIP_TRY(h, hipGetLastError());
    return PE_OK;
}

// The collected signature legs (pe_aggregate_signed), launched: one decompression over all of them, then per leg the subgroup
// check it asked for, the per-group sums and the status copy into ITS arena's pinned block, marked by the arena's ev_leg.  The
// legs' stream is joined into no other stream: round 5 joined it into the state-transition stream -- which is the tree's
// (Tune::state_on) -- and every G1 chain then stood behind the decompression.
int sig_batch_flush(pe_engine* h)
{
    if (h->sig_batch.empty()) return PE_OK;
    std::vector<pe_engine::SigSeg> segs;
    segs.swap(h->sig_batch);
    const bool own = h->aux_stream && h->stream == h->own_stream;
    hipStream_t ss = own ? leg_stream(h) : h->stream;
    // Where the decompression runs.  Beside an accumulation it starves it: both want the multiplier, the SIMDs' arbiters serve
    // the older wave, and an accumulation that meets a decompression ends when that ends (0.8-1.2 ms instead of 0.2; round 4
    // measured it, round 6 again with whole batches: profiles/r06_engine_timeline_signed_beside.txt).  A streaming caller's
    // batch therefore goes onto the ACCUMULATIONS' stream, between two of them by stream order -- no event, no overlap: one
    // ~1 ms launch per batch of steps is what the signatures cost the G1 chain; the latency-sized rest (sums, status copies)
    // follows on the legs' own stream.
    bool between = own && h->side_stream != nullptr;
    for (auto& sg : segs) between = between && sg.streaming;
    hipStream_t ds = between ? h->side_stream : ss;
    // ... and so do the sums since round 6: on the legs' low-priority stream they ran beside the accumulations after all (a fifth
    // active queue, and one accumulation in eight at 360 us instead of 200): 0.62 / 0.55 ms per signed step against 0.53 / 0.52
    // with the whole leg between two accumulations (profiles/NOTES_r06.md 10)
    if (between) ss = ds;
    if (ss != h->stream) {  // behind the groupings (and the host signatures' copies) enqueued on the engine's stream so far
        HIP_TRY(h, hipEventRecord(h->ev_aux_fork, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(ss, h->ev_aux_fork, 0));
        if (ds != ss) HIP_TRY(h, hipStreamWaitEvent(ds, h->ev_aux_fork, 0));
    }
    G2DecompressBatch b{};
    for (auto& sg : segs) {
        if (sg.copy_from) HIP_TRY(h, hipMemcpyAsync(const_cast<uint8_t*>(sg.d_in), sg.copy_from, sg.bytes, hipMemcpyDeviceToDevice, ds));
        if (sg.compressed) {
            b.in96[b.count] = sg.d_in;
            b.out_mont48[b.count] = sg.d_pts;
            b.status[b.count] = sg.d_status;
            b.n[b.count] = sg.n;
            ++b.count;
        } else {
            HIP_TRY(h, hipMemsetAsync(sg.d_status, 0, 4ull * sg.n, ds));
            launch_g2_convert(ds, sg.d_in, sg.d_pts, sg.n);
        }
    }
    if (b.count) {
        ProfScope ps(h, PE_KERNEL_G2_DECOMPRESS, ds);
        launch_g2_decompress_batch(ds, b);
    }
    if (ds != ss) {
        HIP_TRY(h, hipEventRecord(h->ev_sig, ds));
        HIP_TRY(h, hipStreamWaitEvent(ss, h->ev_sig, 0));
    }
    G2AggregateRowsBatch r{};
    for (auto& sg : segs) {
        if (sg.check_subgroup) launch_g2_subgroup_check(ss, sg.d_pts, sg.n, sg.d_status);
        r.pts[r.count] = sg.d_pts;
        r.status[r.count] = sg.d_status;
        r.ug[r.count] = sg.d_ug;
        r.member_row[r.count] = sg.d_member_row;
        r.plan_dev[r.count] = sg.plan_dev;
        r.out96[r.count] = sg.o_sig;
        r.out_bad[r.count] = sg.o_bad;
        r.n_groups[r.count] = sg.ng_bound;
        ++r.count;
    }
    {
        ProfScope ps(h, PE_KERNEL_G2_ACCUMULATE, ss);
        launch_g2_aggregate_rows(ss, r);
    }
    G2StatusOutBatch so{};
    for (auto& sg : segs) {
        so.src[so.count] = sg.d_status;
        so.dst_host[so.count] = sg.o_st;
        so.n[so.count] = sg.n;
        ++so.count;
    }
    launch_g2_status_out(ss, so);  // (one copy command per leg: ~15 us each on a stream the accumulations wait behind)
    for (auto& sg : segs) {
        if (ss != h->stream) {  // the leg's own mark: whoever completes its arena (or rewrites the arena's scratch) waits for it
            pe_engine::PipeArena& a = h->arena[sg.arena];
            HIP_TRY(h, hipEventRecord(a.ev_leg, ss));
            a.leg_used = true;
        }
    }
    HIP_TRY(h, hipGetLastError());
    return PE_OK;
}
bool sig_batch_holds(const pe_engine* h, int arena)
{
    for (const auto& sg : h->sig_batch)
        if (sg.arena == arena) return true;
    return false;
}

}  // namespace posevo

extern "C" {

// ---------------------------------------------------------------- plain G1 sums
// Shared by pe_g1_sum / pe_g1_partial: groups over an optional index list, staged and launched.
static int g1_sum_common(pe_engine* h, const uint32_t* d_pts, uint64_t n_pts, const uint32_t* index,
                         const uint32_t* offsets, uint32_t n_groups, uint8_t* out96_host, uint32_t* dev_jac)
{
    const uint32_t total = offsets[n_groups];
    for (uint32_t g = 0; g < n_groups; ++g)
        if (offsets[g + 1] < offsets[g]) return fail(h, PE_ERR_INVALID_ARG, "offsets not monotone");
    if (index) {
        for (uint32_t j = 0; j < total; ++j)
            if (index[j] >= n_pts) return fail(h, PE_ERR_INVALID_ARG, "point index out of range");
    } else if (total > n_pts) {
        return fail(h, PE_ERR_INVALID_ARG, "offsets exceed the number of points");
    }
    Stage st(h);
    PE_TRY(st.reserve(sizeof(G1Group) * (size_t)n_groups + 4ull * total + 4096));
    const size_t off_g = st.alloc(sizeof(G1Group) * (size_t)n_groups);
    const size_t off_i = st.alloc(4ull * total + 4);
    G1Group* gr = st.host<G1Group>(off_g);
    G1Plan plan;
    plan_g1(n_groups, [&](uint32_t g) { return offsets[g + 1] - offsets[g]; }, gr, &plan);
    for (uint32_t g = 0; g < n_groups; ++g) gr[g].member_start = offsets[g];
    if (index) memcpy(st.host<uint32_t>(off_i), index, 4ull * total);
    OutBlock ob(h);
    const size_t off_o = out96_host ? ob.alloc(96ull * n_groups) : 0;
    PE_TRY(ob.ensure());
    HIP_TRY(h, st.upload());
    int rc = launch_g1_planned(h, d_pts, index ? st.dev<uint32_t>(off_i) : nullptr, nullptr, st.dev<G1Group>(off_g), plan,
                               out96_host ? ob.dev<uint8_t>(off_o) : nullptr, dev_jac, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, d_pts == h->d_tmp_points.as<uint32_t>() ? n_pts : 0);
    if (rc) return rc;
    if (out96_host) HIP_TRY(h, ob.download());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (out96_host) memcpy(out96_host, ob.host<uint8_t>(off_o), 96ull * n_groups);
    return PE_OK;
}

int pe_g1_sum(pe_engine* h, const uint8_t* points96, uint64_t n_points, const uint32_t* index, const uint32_t* offsets,
              uint32_t n_groups, uint8_t* out96)
{
    if (!h || !offsets || !out96) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n_groups == 0) return PE_OK;
    const uint32_t* d_pts;
    uint64_t np;
    if (points96) {
        if (n_points >= 0xFFFFFFFFull) return fail(h, PE_ERR_CAPACITY, "too many points");
        HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(96, 96ull * n_points)));
        HIP_TRY(h, h->d_tmp_points.ensure(std::max<size_t>(128, 4ull * G1_ROW_WORDS * n_points)));
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, points96, 96ull * n_points, hipMemcpyHostToDevice, h->stream));
        launch_g1_convert(h->stream, h->d_tmp_be.as<uint8_t>(), h->d_tmp_points.as<uint32_t>(), n_points);
        d_pts = h->d_tmp_points.as<uint32_t>();
        np = n_points;
    } else {
        if (!h->have_points) return fail(h, PE_ERR_STATE, "no pubkeys loaded");
        d_pts = h->d_points.as<uint32_t>();
        np = h->n_val;
    }
    return g1_sum_common(h, d_pts, np, index, offsets, n_groups, out96, nullptr);
}

int pe_g1_partial(pe_engine* h, const uint32_t* index, const uint32_t* offsets, uint32_t n_groups, void* dev_partials,
                  uint32_t dev_partials_capacity)
{
    if (!h || !offsets || !dev_partials) return PE_ERR_INVALID_ARG;
    if (n_groups > dev_partials_capacity) return fail(h, PE_ERR_CAPACITY, "dev_partials holds fewer than n_groups partials");
    PE_TRY(enter(h));
    if (!h->have_points) return fail(h, PE_ERR_STATE, "no pubkeys loaded");
    if (n_groups == 0) return PE_OK;
    return g1_sum_common(h, h->d_points.as<uint32_t>(), h->n_val, index, offsets, n_groups, nullptr,
                         static_cast<uint32_t*>(dev_partials));
}

int pe_g1_finish(pe_engine* h, const void* dev_gathered, uint32_t n_ranks, uint32_t n_groups, uint8_t* out96)
{
    if (!h || !dev_gathered || !out96 || n_ranks == 0) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n_groups == 0) return PE_OK;
    HIP_TRY(h, h->d_out96.ensure(96ull * n_groups));
    {
        ProfScope ps(h, PE_KERNEL_G1_NORMALISE);
        launch_g1_finish(h->stream, static_cast<const uint32_t*>(dev_gathered), nullptr, n_groups, n_ranks, n_groups,
                         h->d_out96.as<uint8_t>(), nullptr);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, h->A().h_pin.ensure(96ull * n_groups));
    HIP_TRY(h, hipMemcpyAsync(h->A().h_pin.p, h->d_out96.p, 96ull * n_groups, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(out96, h->A().h_pin.p, 96ull * n_groups);
    return PE_OK;
}

// ---------------------------------------------------------------- BLSPubkey wire format (8(f) rank 3)
static int g1_decompress_common(pe_engine* h, const uint8_t* in48, uint64_t n, uint32_t* d_mont24, uint8_t* out96,
                                int32_t* status, uint64_t* n_bad)
{
    const uint64_t chunk = 1ull << 20;
    HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(64, (48ull + 96ull + 4ull) * std::min(chunk, n))));
    *n_bad = 0;
    for (uint64_t base = 0; base < n; base += chunk) {
        const uint64_t m = std::min(chunk, n - base);
        uint8_t* d_in = h->d_tmp_be.as<uint8_t>();
        uint8_t* d_out = d_in + 48ull * m;
        int32_t* d_st = reinterpret_cast<int32_t*>(d_out + 96ull * m);
        HIP_TRY(h, hipMemcpyAsync(d_in, in48 + 48ull * base, 48ull * m, hipMemcpyHostToDevice, h->stream));
        launch_g1_decompress(h->stream, d_in, m, d_mont24 ? d_mont24 + (uint64_t)G1_ROW_WORDS * base : nullptr, out96 ? d_out : nullptr, d_st);
        HIP_TRY(h, hipGetLastError());
        if (out96) HIP_TRY(h, hipMemcpyAsync(out96 + 96ull * base, d_out, 96ull * m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(status + base, d_st, 4ull * m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (uint64_t i = 0; i < m; ++i) *n_bad += status[base + i] != 0;
    }
    return PE_OK;
}

int pe_g1_decompress(pe_engine* h, const uint8_t* in48, uint64_t n, uint8_t* out96, int32_t* status)
{
    if (!h || (n && (!in48 || !out96 || !status))) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n == 0) return PE_OK;
    uint64_t n_bad = 0;
    return g1_decompress_common(h, in48, n, nullptr, out96, status, &n_bad);
}

int pe_set_pubkeys_compressed(pe_engine* h, uint64_t n, const uint8_t* pubkeys48, int32_t* status)
{
    if (!h || (n && (!pubkeys48 || !status))) return PE_ERR_INVALID_ARG;
    if (n != h->n_val) return fail(h, PE_ERR_INVALID_ARG, "pe_set_pubkeys_compressed: n differs from the registry size");
    PE_TRY(enter(h));
    if (n == 0) return PE_OK;
    HIP_TRY(h, h->d_points.ensure(4ull * G1_ROW_WORDS * n));
    h->have_points = false;
    h->points29_valid = false;
    uint64_t n_bad = 0;
    int rc = g1_decompress_common(h, pubkeys48, n, h->d_points.as<uint32_t>(), nullptr, status, &n_bad);
    if (rc) return rc;
    if (n_bad) return fail(h, PE_ERR_INVALID_ARG, std::to_string(n_bad) + " pubkeys do not decode to curve points (see status[])");
    h->have_points = true;
    PE_TRY(build_points29(h, n));
    return PE_OK;
}

// KeyValidate (Appendix A.7: FastAggregateVerify validates every pubkey): points96 NULL = the registry as loaded.
int pe_g1_key_validate(pe_engine* h, const uint8_t* points96, uint64_t n, int32_t* status)
{
    if (!h || (n && !status)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n == 0) return PE_OK;
    const uint32_t* d_pts;
    if (points96) {
        if (n >= 0xFFFFFFFFull) return fail(h, PE_ERR_CAPACITY, "too many points");
        HIP_TRY(h, h->d_tmp_be.ensure(96ull * n));
        HIP_TRY(h, h->d_tmp_points.ensure(4ull * G1_ROW_WORDS * n + 4ull * n));
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, points96, 96ull * n, hipMemcpyHostToDevice, h->stream));
        launch_g1_convert(h->stream, h->d_tmp_be.as<uint8_t>(), h->d_tmp_points.as<uint32_t>(), n);
        d_pts = h->d_tmp_points.as<uint32_t>();
    } else {
        if (!h->have_points || n != h->n_val) return fail(h, PE_ERR_STATE, "pe_g1_key_validate: no pubkeys loaded / n differs from the registry");
        d_pts = h->d_points.as<uint32_t>();
    }
    HIP_TRY(h, h->d_out96.ensure(4ull * n));
    int32_t* d_st = h->d_out96.as<int32_t>();
    launch_g1_key_validate(h->stream, d_pts, n, d_st);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(status, d_st, 4ull * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}

int pe_g2_decompress(pe_engine* h, const uint8_t* in96, uint64_t n, uint8_t* out192, int32_t* status)
{
    if (!h || (n && (!in96 || !out192 || !status))) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    const uint64_t chunk = 1ull << 19;
    HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(64, (96ull + 192ull + 4ull) * std::min(chunk, std::max<uint64_t>(n, 1)))));
    for (uint64_t base = 0; base < n; base += chunk) {
        const uint64_t m = std::min(chunk, n - base);
        uint8_t* d_in = h->d_tmp_be.as<uint8_t>();
        uint8_t* d_out = d_in + 96ull * m;
        int32_t* d_st = reinterpret_cast<int32_t*>(d_out + 192ull * m);
        HIP_TRY(h, hipMemcpyAsync(d_in, in96 + 96ull * base, 96ull * m, hipMemcpyHostToDevice, h->stream));
        launch_g2_decompress(h->stream, d_in, m, nullptr, d_out, d_st);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(out192 + 192ull * base, d_out, 192ull * m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(status + base, d_st, 4ull * m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return PE_OK;
}

// Serialisation only (flag bits from a 48-byte comparison); no device work, no handle.
int pe_g1_compress(const uint8_t* in96, uint64_t n, uint8_t* out48)
{
    if (n && (!in96 || !out48)) return PE_ERR_INVALID_ARG;
    static const uint8_t HALF_BE[48] = {  // (p - 1) / 2, big-endian
        0x0d, 0x00, 0x88, 0xf5, 0x1c, 0xbf, 0xf3, 0x4d, 0x25, 0x8d, 0xd3, 0xdb, 0x21, 0xa5, 0xd6, 0x6b,
        0xb2, 0x3b, 0xa5, 0xc2, 0x79, 0xc2, 0x89, 0x5f, 0xb3, 0x98, 0x69, 0x50, 0x7b, 0x58, 0x7b, 0x12,
        0x0f, 0x55, 0xff, 0xff, 0x58, 0xa9, 0xff, 0xff, 0xdc, 0xff, 0x7f, 0xff, 0xff, 0xff, 0xd5, 0x55};
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* p = in96 + 96 * i;
        uint8_t* o = out48 + 48 * i;
        if (p[0] & 0x40) {
            memset(o, 0, 48);
            o[0] = 0xC0;
            continue;
        }
        memcpy(o, p, 48);
        o[0] = (uint8_t)((o[0] & 0x1f) | 0x80 | (memcmp(p + 48, HALF_BE, 48) > 0 ? 0x20 : 0));
    }
    return PE_OK;
}

int pe_g2_compress(const uint8_t* in192, uint64_t n, uint8_t* out96)
{
    if (n && (!in192 || !out96)) return PE_ERR_INVALID_ARG;
    static const uint8_t HALF_BE[48] = {
        0x0d, 0x00, 0x88, 0xf5, 0x1c, 0xbf, 0xf3, 0x4d, 0x25, 0x8d, 0xd3, 0xdb, 0x21, 0xa5, 0xd6, 0x6b,
        0xb2, 0x3b, 0xa5, 0xc2, 0x79, 0xc2, 0x89, 0x5f, 0xb3, 0x98, 0x69, 0x50, 0x7b, 0x58, 0x7b, 0x12,
        0x0f, 0x55, 0xff, 0xff, 0x58, 0xa9, 0xff, 0xff, 0xdc, 0xff, 0x7f, 0xff, 0xff, 0xff, 0xd5, 0x55};
    static const uint8_t ZERO48[48] = {0};
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* p = in192 + 192 * i;
        uint8_t* o = out96 + 96 * i;
        if (p[0] & 0x40) {
            memset(o, 0, 96);
            o[0] = 0xC0;
            continue;
        }
        memcpy(o, p, 96);  // x.c1 | x.c0
        const uint8_t* y1 = p + 96;
        const uint8_t* y0 = p + 144;
        const bool larger = memcmp(y1, ZERO48, 48) != 0 ? memcmp(y1, HALF_BE, 48) > 0 : memcmp(y0, HALF_BE, 48) > 0;
        o[0] = (uint8_t)((o[0] & 0x1f) | 0x80 | (larger ? 0x20 : 0));
    }
    return PE_OK;
}

// ---------------------------------------------------------------- plain G2 sums (8(f) rank 3)
int pe_g2_sum(pe_engine* h, const uint8_t* points192, uint64_t n_points, const uint32_t* index, const uint32_t* offsets,
              uint32_t n_groups, uint8_t* out192)
{
    if (!h || !offsets || !out192 || (n_points && !points192)) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n_groups == 0) return PE_OK;
    if (n_points >= 0xFFFFFFFFull / 48) return fail(h, PE_ERR_CAPACITY, "too many points");
    const uint32_t total = offsets[n_groups];
    for (uint32_t g = 0; g < n_groups; ++g)
        if (offsets[g + 1] < offsets[g]) return fail(h, PE_ERR_INVALID_ARG, "offsets not monotone");
    if (index) {
        for (uint32_t j = 0; j < total; ++j)
            if (index[j] >= n_points) return fail(h, PE_ERR_INVALID_ARG, "point index out of range");
    } else if (total > n_points) {
        return fail(h, PE_ERR_INVALID_ARG, "offsets exceed the number of points");
    }
    HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(192, 192ull * n_points)));
    HIP_TRY(h, h->d_tmp_points.ensure(std::max<size_t>(192, 192ull * n_points)));
    if (n_points) {
        HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, points192, 192ull * n_points, hipMemcpyHostToDevice, h->stream));
        launch_g2_convert(h->stream, h->d_tmp_be.as<uint8_t>(), h->d_tmp_points.as<uint32_t>(), n_points);
    }
    Stage st(h);
    PE_TRY(st.reserve(sizeof(G1Group) * (size_t)n_groups + 4ull * total + 4096));
    const size_t off_g = st.alloc(sizeof(G1Group) * (size_t)n_groups);
    const size_t off_i = st.alloc(4ull * total + 4);
    G1Group* gr = st.host<G1Group>(off_g);
    G1Plan plan;
    plan_g1(n_groups, [&](uint32_t g) { return offsets[g + 1] - offsets[g]; }, gr, &plan, G2_WG_SLOTS,
            G1_TARGET_LANES / 2);
    for (uint32_t g = 0; g < n_groups; ++g) gr[g].member_start = offsets[g];
    if (index) memcpy(st.host<uint32_t>(off_i), index, 4ull * total);
    OutBlock ob(h);
    const size_t off_o = ob.alloc(192ull * n_groups);
    PE_TRY(ob.ensure());
    HIP_TRY(h, st.upload());
    HIP_TRY(h, h->d_partials.ensure(std::max<size_t>(384, 384ull * plan.n_partials)));
    {
        ProfScope ps(h, PE_KERNEL_G2_ACCUMULATE);
        launch_g2_accumulate(h->stream, h->d_tmp_points.as<uint32_t>(), index ? st.dev<uint32_t>(off_i) : nullptr,
                             st.dev<G1Group>(off_g), plan.n_groups, plan.n_slots, h->d_partials.as<uint32_t>());
    }
    {
        ProfScope ps(h, PE_KERNEL_G2_NORMALISE);
        launch_g2_finish(h->stream, h->d_partials.as<uint32_t>(), st.dev<G1Group>(off_g), plan.n_groups,
                         ob.dev<uint8_t>(off_o));
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, ob.download());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(out192, ob.host<uint8_t>(off_o), 192ull * n_groups);
    return PE_OK;
}

// ---------------------------------------------------------------- bls.Aggregate over an epoch's unaggregated signatures
int pe_aggregate_signatures(pe_engine* h, const uint8_t* signatures96, uint64_t n, const uint32_t* index,
                            const uint32_t* offsets, uint32_t n_groups, uint32_t sig_flags, uint8_t* out_signatures96,
                            int32_t* sig_status, uint32_t* out_bad)
{
    if (!h || !offsets || !out_signatures96 || (n && !signatures96)) return PE_ERR_INVALID_ARG;
    if (sig_flags & ~(uint32_t)PE_SIG_CHECK_SUBGROUP) return fail(h, PE_ERR_INVALID_ARG, "pe_aggregate_signatures: unknown flags");
    PE_TRY(enter(h));
    if (n_groups == 0) return PE_OK;
    if (n >= 0xFFFFFFFFull / 48) return fail(h, PE_ERR_CAPACITY, "too many signatures");
    HostLap lap(&h->trace);
    const uint32_t total = offsets[n_groups];
    for (uint32_t g = 0; g < n_groups; ++g)
        if (offsets[g + 1] < offsets[g]) return fail(h, PE_ERR_INVALID_ARG, "offsets not monotone");
    auto on_device = [](const void* p) {
        hipPointerAttribute_t pa;
        if (p && hipPointerGetAttributes(&pa, p) == hipSuccess) return pa.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        return false;
    };
    const bool idx_dev = index && on_device(index);
    if (index && !idx_dev) {
        for (uint32_t j = 0; j < total; ++j)
            if (index[j] >= n) return fail(h, PE_ERR_INVALID_ARG, "signature index out of range");
    } else if (!index && total > n) {
        return fail(h, PE_ERR_INVALID_ARG, "offsets exceed the number of signatures");
    }
    hipStream_t s = h->stream;
    // wire bytes: in place when they lie in device memory at a 16-byte boundary, else through the engine's scratch
    const bool sig_dev = on_device(signatures96);
    const uint8_t* d_in = signatures96;
    if (!(sig_dev && (reinterpret_cast<uintptr_t>(signatures96) & 15) == 0)) {
        HIP_TRY(h, h->d_tmp_be.ensure(std::max<size_t>(96, 96ull * n)));
        if (n) HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, signatures96, 96ull * n, sig_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        d_in = h->d_tmp_be.as<uint8_t>();
    }
    HIP_TRY(h, h->d_tmp_points.ensure(std::max<size_t>(192, 192ull * n + 4ull * n + (idx_dev ? 4ull * total : 0) + 64)));
    uint32_t* d_pts = h->d_tmp_points.as<uint32_t>();
    int32_t* d_status = reinterpret_cast<int32_t*>(h->d_tmp_points.as<uint8_t>() + 192ull * n);
    uint32_t* d_index_checked = reinterpret_cast<uint32_t*>(h->d_tmp_points.as<uint8_t>() + 196ull * n);  // a device index, range-checked
    lap.mark("usig.1_checks_scratch");
    launch_g2_decompress(s, d_in, n, d_pts, nullptr, d_status);  // a signature that does not decode becomes the (0, 0) row: infinity
    lap.mark("usig.2_decompress_launch");
    if (sig_flags & PE_SIG_CHECK_SUBGROUP) {
        launch_g2_subgroup_check(s, d_pts, n, d_status);
        launch_g2_mask_bad(s, d_pts, n, d_status);  // a decoded point outside G2 is left out of its sum like an undecodable one
    }
    Stage st(h);
    PE_TRY(st.reserve(sizeof(G1Group) * (size_t)n_groups + (idx_dev ? 0 : 4ull * total) + 4096));
    const size_t off_g = st.alloc(sizeof(G1Group) * (size_t)n_groups);
    const size_t off_i = st.alloc((index && !idx_dev ? 4ull * total : 0) + 4);
    G1Group* gr = st.host<G1Group>(off_g);
    G1Plan plan;
    plan_g1(n_groups, [&](uint32_t g) { return offsets[g + 1] - offsets[g]; }, gr, &plan, G2_WG_SLOTS, G1_TARGET_LANES / 2);
    for (uint32_t g = 0; g < n_groups; ++g) gr[g].member_start = offsets[g];
    if (index && !idx_dev) memcpy(st.host<uint32_t>(off_i), index, 4ull * total);
    const uint32_t* d_index = !index ? nullptr : idx_dev ? d_index_checked : st.dev<uint32_t>(off_i);
    // Statuses come back through the PINNED output block (a copy into the caller's pageable memory is staged and pinned by the
    // runtime page by page: 2 x 4 MB cost the call 30 ms of its 50 in bench.py's process, profiles/r06_sig_host_phases.txt); the
    // bad-member counts are taken on the device (k_g2_count_bad), so neither statuses nor index travel for them; a device-resident
    // index is range-checked there too (k_g2_index_check: the host cannot read it).
    const bool want_status = sig_status != nullptr;
    OutBlock ob(h);
    const size_t off_o = ob.alloc(192ull * n_groups);
    const size_t off_st = want_status ? ob.alloc(4ull * n) : 0;
    const size_t off_bad = ob.alloc(4ull * n_groups);
    const size_t off_err = ob.alloc(4);
    PE_TRY(ob.ensure());
    *ob.host<uint32_t>(off_err) = 0;
    HIP_TRY(h, st.upload());
    if (idx_dev) {
        HIP_TRY(h, hipMemsetAsync(ob.dev<uint32_t>(off_err), 0, 4, s));
        launch_g2_index_check(s, index, total, (uint32_t)n, d_index_checked, ob.dev<uint32_t>(off_err));
        HIP_TRY(h, ob.download(off_err, 4));
    }
    HIP_TRY(h, h->d_partials.ensure(std::max<size_t>(384, 384ull * plan.n_partials)));
    {
        ProfScope ps(h, PE_KERNEL_G2_ACCUMULATE);
        launch_g2_accumulate(s, d_pts, d_index, st.dev<G1Group>(off_g), plan.n_groups, plan.n_slots, h->d_partials.as<uint32_t>());
    }
    {
        ProfScope ps(h, PE_KERNEL_G2_NORMALISE);
        launch_g2_finish(s, h->d_partials.as<uint32_t>(), st.dev<G1Group>(off_g), plan.n_groups, ob.dev<uint8_t>(off_o));
    }
    if (out_bad) launch_g2_count_bad(s, d_status, d_index, st.dev<G1Group>(off_g), n_groups, ob.dev<uint32_t>(off_bad));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, ob.download(off_o, 192ull * n_groups));
    if (out_bad) HIP_TRY(h, ob.download(off_bad, 4ull * n_groups));
    if (want_status && n) HIP_TRY(h, hipMemcpyAsync(ob.host<int32_t>(off_st), d_status, 4ull * n, hipMemcpyDeviceToHost, s));
    lap.mark("usig.3_plan_launch_copies");
    HIP_TRY(h, hipStreamSynchronize(s));
    lap.mark("usig.4_wait");
    if (*ob.host<uint32_t>(off_err)) return fail(h, PE_ERR_INVALID_ARG, "signature index out of range (device-resident index)");
    PE_TRY(pe_g2_compress(ob.host<uint8_t>(off_o), n_groups, out_signatures96));
    if (sig_status && n) memcpy(sig_status, ob.host<int32_t>(off_st), 4ull * n);
    if (out_bad) memcpy(out_bad, ob.host<uint32_t>(off_bad), 4ull * n_groups);
    lap.mark("usig.5_outputs");
    return PE_OK;
}

// ---------------------------------------------------------------- pe_aggregate with its signature leg
int pe_g2_subgroup_check(pe_engine* h, const uint8_t* points192, uint64_t n, int32_t* status)
{
    if (!h || (n && (!points192 || !status))) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (n == 0) return PE_OK;
    if (n >= 0xFFFFFFFFull / 48) return fail(h, PE_ERR_CAPACITY, "too many points");
    HIP_TRY(h, h->d_tmp_be.ensure(192ull * n + 4ull * n));
    HIP_TRY(h, h->d_tmp_points.ensure(192ull * n));
    int32_t* d_st = reinterpret_cast<int32_t*>(h->d_tmp_be.as<uint8_t>() + 192ull * n);
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_be.p, points192, 192ull * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(d_st, 0, 4ull * n, h->stream));
    launch_g2_convert(h->stream, h->d_tmp_be.as<uint8_t>(), h->d_tmp_points.as<uint32_t>(), n);
    launch_g2_subgroup_check(h->stream, h->d_tmp_points.as<uint32_t>(), n, d_st);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(status, d_st, 4ull * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PE_OK;
}

int pe_aggregate_signed(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena, uint64_t arena_len,
                        const uint8_t* signatures, uint32_t sig_format_flags, pe_attestation* out_atts,
                        uint32_t* out_n_groups, uint32_t* group_of, uint8_t* out_bits_arena, uint64_t out_arena_cap,
                        uint8_t* out_signatures96, int32_t* sig_status, uint8_t* out_aggpk96, uint32_t* out_count)
{
    if (!h || !out_n_groups || !out_atts || (n && (!atts || !signatures || !out_signatures96 || !sig_status)))
        return PE_ERR_INVALID_ARG;
    const uint32_t fmt = sig_format_flags & 0xFFu;
    if ((fmt != PE_SIG_G2_COMPRESSED && fmt != PE_SIG_G2_UNCOMPRESSED) || (sig_format_flags & ~(0xFFu | PE_SIG_CHECK_SUBGROUP)))
        return fail(h, PE_ERR_INVALID_ARG, "pe_aggregate_signed: unknown signature format / flags");
    (void)hipSetDevice(h->device);
    if (n == 0) return pe_aggregate(h, atts, n, bits_arena, arena_len, nullptr, out_atts, out_n_groups, group_of,
                                    out_bits_arena, out_arena_cap, nullptr, out_aggpk96, out_count);
    const bool dev_rows = rows_on_device(atts);
    const size_t sig_bytes = fmt == PE_SIG_G2_COMPRESSED ? 96 : 192;
    HostLap lap(&h->trace);
    {   // the arena's signature scratch, sized while nothing of this call is in flight
        if (!h->pipelining) PE_TRY(flush_pending(h));
        pe_engine::PipeArena& A = h->A();
        (void)A;
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_sig_in, sig_bytes * n));
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_sig_pts, 192ull * n));
        PE_TRY(ensure_quiesced_arenas(h, &pe_engine::PipeArena::d_sig_status, 4ull * n));
    }
    bool sig_on_device = false;
    {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, signatures) == hipSuccess) sig_on_device = pa.type == hipMemoryTypeDevice;
        else (void)hipGetLastError();
    }
    // Signatures in HOST memory are brought in during the call, in front of the grouping on the engine's stream: the leg below
    // may be deferred past this call's return (a streaming step holds it back until the next aggregate), and a caller is free
    // to refill a host buffer per step -- as with bits_arena.  Device-resident signatures are read where the leg
    // runs (include/posevo.h: they stay unchanged until the pipeline completes, like device rows).
    if (!sig_on_device)
        HIP_TRY(h, hipMemcpyAsync(h->A().d_sig_in.p, signatures, sig_bytes * n, hipMemcpyHostToDevice, h->stream));
    // host rows: the grouping comes back in group_of (host-derived, complete at return) -- keep one if the caller has none
    auto gof_p = std::make_shared<std::vector<uint32_t>>();
    if (!dev_rows && !group_of) {
        gof_p->resize(n);
        group_of = gof_p->data();
    }
    lap.mark("sagg.1_prepare");
    const int rc = pe_aggregate(h, atts, n, bits_arena, arena_len, nullptr, out_atts, out_n_groups, group_of, out_bits_arena,
                                out_arena_cap, nullptr, out_aggpk96, out_count);
    if (rc) return rc;
    lap.mark("sagg.2_aggregate");
    pe_engine::PipeArena& A = h->A();
    const UnionGroup* d_ug = nullptr;
    const uint32_t* d_member_row = nullptr;
    const AttPlan* plan_dev = nullptr;
    uint32_t ng_bound = n;
    Stage st(h);
    if (dev_rows) {
        PE_TRY(resident_lists(h, &d_ug, &d_member_row));
        PE_TRY(resident_plan_dev(h, &plan_dev));
    } else {
        const uint32_t ng = *out_n_groups;
        ng_bound = ng;
        PE_TRY(st.reserve(sizeof(UnionGroup) * (size_t)ng + 4ull * n + 1024));
        const size_t off_ug = st.alloc(sizeof(UnionGroup) * (size_t)std::max<uint32_t>(ng, 1));
        const size_t off_ord = st.alloc(4ull * n);
        UnionGroup* ug = st.host<UnionGroup>(off_ug);
        uint32_t* order = st.host<uint32_t>(off_ord);
        for (uint32_t g = 0; g < ng; ++g) ug[g] = UnionGroup{0, 0, 0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            if (group_of[i] >= ng) return fail(h, PE_ERR_NO_DEVICE, "pe_aggregate returned a group index out of range");
            ug[group_of[i]].n_atts += 1;
        }
        uint32_t run = 0;
        for (uint32_t g = 0; g < ng; ++g) { ug[g].list_start = run; run += ug[g].n_atts; ug[g].n_atts = 0; }
        for (uint32_t i = 0; i < n; ++i) { UnionGroup& u = ug[group_of[i]]; order[u.list_start + u.n_atts++] = i; }
        HIP_TRY(h, st.upload());
        d_ug = st.dev<UnionGroup>(off_ug);
        d_member_row = st.dev<uint32_t>(off_ord);
    }
    OutBlock ob(h);
    const size_t off_sig = ob.alloc(96ull * std::max<uint32_t>(ng_bound, 1));
    const size_t off_bad = ob.alloc(4ull * std::max<uint32_t>(ng_bound, 1));
    const size_t off_st = ob.alloc(4ull * n);
    PE_TRY(ob.ensure());
    memset(ob.host<uint32_t>(off_bad), 0, 4ull * std::max<uint32_t>(ng_bound, 1));
    // The leg: decode the signatures, sum them per group, compress the sums -- on the signature legs' stream, beside the
    // aggregate's pubkey sums and the fork-choice kernels.  k_g2_decompress is ONE chain of ~970 dependent Fp products per lane:
    // 8192 signatures are 128 waves that hold an eighth of the chip for ~0.95 ms, and so are 65 536.  A streaming caller's legs
    // are therefore COLLECTED (pe_engine::sig_batch) and decoded by one launch per Tune::sig_batch steps (sig_batch_flush); the
    // per-step sums follow it.  A step's signature outputs stay inside the lag contract as long as the lag depth exceeds the
    // batch by the ~4 steps a decompression lasts (bench.py runs the signed leg at lag 7); a pipeline that completes earlier
    // (a drain, a synchronous call) flushes what has been collected (complete_arena).
    pe_engine::SigSeg seg;
    seg.arena = h->cur;
    seg.n = n;
    seg.compressed = fmt == PE_SIG_G2_COMPRESSED;
    seg.check_subgroup = (sig_format_flags & PE_SIG_CHECK_SUBGROUP) != 0;
    seg.d_in = A.d_sig_in.as<uint8_t>();
    seg.copy_from = nullptr;
    if (sig_on_device) {  // in place where the kernel's 16-byte loads allow it, else through the arena's scratch (copied by the leg)
        if ((reinterpret_cast<uintptr_t>(signatures) & 15) == 0) seg.d_in = signatures;
        else seg.copy_from = signatures;
    }
    seg.bytes = sig_bytes * n;
    seg.d_pts = A.d_sig_pts.as<uint32_t>();
    seg.d_status = A.d_sig_status.as<int32_t>();
    seg.d_ug = d_ug;
    seg.d_member_row = d_member_row;
    seg.ng_bound = ng_bound;
    seg.plan_dev = plan_dev;
    seg.o_sig = ob.host<uint8_t>(off_sig);
    seg.o_bad = ob.host<uint32_t>(off_bad);
    seg.o_st = ob.host<int32_t>(off_st);
    const bool collect = seg.compressed && h->streaming && h->pipelining && h->stream == h->own_stream && h->aux_stream != nullptr;
    seg.streaming = collect && h->last_agg_on_side && h->side_stream != nullptr;
    h->sig_batch.push_back(seg);
    lap.mark("sagg.3_leg_collect");
    if (!collect || h->sig_batch.size() >= (size_t)std::min<int>(std::max(h->tune.sig_batch, 1), (int)G2_BATCH_MAX)) {
        PE_TRY(sig_batch_flush(h));
        lap.mark("sagg.4_leg_flush");
    }
    const size_t base = ob.base;
    const int ai = h->cur;
    auto complete = [h, ai, base, off_sig, off_bad, off_st, n, ng_bound, dev_rows, out_n_groups, out_atts, out_signatures96,
                     sig_status, gof_p]() -> int {
        const uint8_t* pin = h->arena[ai].h_pin.as<uint8_t>() + base;
        const uint32_t ng = dev_rows ? *out_n_groups : ng_bound;  // device rows: set by the aggregate's completion just before
        if (ng > ng_bound) return fail(h, PE_ERR_NO_DEVICE, "pe_aggregate_signed: more groups than rows");
        memcpy(out_signatures96, pin + off_sig, 96ull * ng);
        memcpy(sig_status, pin + off_st, 4ull * n);
        const uint32_t* bad = reinterpret_cast<const uint32_t*>(pin + off_bad);
        for (uint32_t g = 0; g < ng; ++g)
            if (bad[g]) out_atts[g].flags &= ~(uint32_t)PE_ATT_FLAG_SIGNATURE_VALID;  // an undecodable member: never verifiable
        return PE_OK;
    };
    return finish_call(h, st, ob, complete);
}

}  // extern "C"
