// engine_core.cpp -- lifecycle of the handle, completion of batch calls (arenas, pipelines), profiling hooks.
//
// There is NO CPU fallback: every hot-path entry point fails with PE_ERR_NO_DEVICE when the HIP device is unavailable.
#include "engine_internal.h"

using namespace posevo;

namespace posevo {

// ------------------------------------------------------------------ errors
int fail(pe_engine* h, int code, const std::string& msg)
{
    if (h) h->last_error = msg;
    return code;
}
int hip_fail(pe_engine* h, hipError_t e, const char* what)
{
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(h, e == hipErrorOutOfMemory ? PE_ERR_OOM : PE_ERR_NO_DEVICE, m);
}

// ------------------------------------------------------------------ completion of batch calls
// Launches a streaming pipeline held back (the G1 sums of its pe_aggregate): issue them now.
int run_deferred(pe_engine* h)
{
    if (h->deferred.empty()) return PE_OK;
    std::vector<std::function<int()>> todo;
    todo.swap(h->deferred);
    int rc = PE_OK;
    for (auto& f : todo) {
        const int r = f();
        if (r && !rc) rc = r;
    }
    return rc;
}
// Wait for what the batch calls put into one arena, then run their completions in call order.
int complete_arena(pe_engine* h, int ai)
{
    pe_engine::PipeArena& a = h->arena[ai];
    if (a.pending.empty() && !a.fenced && a.stage_cursor == 0 && a.out_cursor == 0) return PE_OK;
    // fork-choice launches of this arena's pipeline still held back for a next aggregate (engine_pair.cpp): nobody waits for
    // work that has not been launched
    if (h->held.active && h->held.arena == ai) PE_TRY(held_issue(h));
    if (ai == h->cur) PE_TRY(run_deferred(h));  // deferred launches belong to the arena the calls are going into
    if (sig_batch_holds(h, ai)) PE_TRY(sig_batch_flush(h));  // a signature leg still collected for a later launch: now
    if (a.fence_pending) PE_TRY(fence_arena(h, a));  // (cannot happen behind held_issue; kept as the invariant's last line)
    hipError_t e = hipSuccess;
    if (a.fenced) {  // a lagged pipeline: its end was marked on every stream it used
        e = bounded_event_sync(h, a.ev_main);
        if (a.side_used) {
            hipError_t e2 = bounded_event_sync(h, a.ev_side);
            if (e == hipSuccess) e = e2;
        }
        if (a.aux_used) {
            hipError_t e2 = bounded_event_sync(h, a.ev_aux);
            if (e == hipSuccess) e = e2;
        }
    } else {         // the arena the calls are still going into
        e = bounded_stream_sync(h, h->stream);
        if (h->aux_busy) {
            hipError_t e2 = bounded_stream_sync(h, h->aux_stream);
            if (e == hipSuccess) e = e2;
            h->aux_busy = false;
        }
        if (h->side_busy) {
            for (hipStream_t s : {h->side_stream, h->fin_stream, h->norm_stream}) {
                if (!s) continue;
                hipError_t e2 = bounded_stream_sync(h, s);
                if (e == hipSuccess) e = e2;
            }
            h->side_busy = false;
        }
    }
    if (a.leg_used) {  // the arena's signature leg runs on a stream of its own: joined into no other (a ~1 ms decompression in
                       // front of the tree's stream would put every G1 chain behind it)
        hipError_t e2 = bounded_event_sync(h, a.ev_leg);
        if (e == hipSuccess) e = e2;
        a.leg_used = false;
    }
    std::vector<std::function<int()>> todo;
    todo.swap(a.pending);
    // arenas complete oldest first; the CURRENT arena completed in the middle of its own pipeline (a block that had to grow)
    // is not a completed pipeline: the calls still to come put their outputs behind
    if (a.generation > h->pipes_completed && !(h->pipelining && ai == h->cur)) h->pipes_completed = a.generation;
    a.stage_cursor = a.out_cursor = 0;
    a.fenced = a.side_used = a.aux_used = a.aux_reads_scratch = a.fence_pending = false;
    if (e == hipErrorNotReady)
        return fail(h, PE_ERR_TIMEOUT, "a collective did not complete within " + std::to_string(h->dist_timeout_ms) +
                    " ms: the communicators were aborted (pe_dist_destroy, then pe_dist_init_ex with PE_DIST_SINGLE_COMM)");
    if (e != hipSuccess) return hip_fail(h, e, "waiting for the enqueued batch calls");
    int rc = PE_OK;
    for (auto& f : todo) {
        const int r = f();
        if (r && !rc) rc = r;
    }
    return rc;
}
// A streaming pipeline's pe_get_head has ~25 us to spare between launching k_tree and seeing the head: spend them on
// the completion (copy-out of ~330 KB) of the oldest lagged pipeline, if the device is already through with it --
// pe_pipeline_end_lagged would otherwise do that work after this step's calls.  A failing completion is reported by the
// next call that completes pipelines.
void complete_oldest_if_ready(pe_engine* h)
{
    const int ai = (h->cur + 1) % h->n_arenas;
    pe_engine::PipeArena& a = h->arena[ai];
    if (!a.fenced || a.fence_pending || a.pending.empty() || sig_batch_holds(h, ai)) return;
    if (hipEventQuery(a.ev_main) != hipSuccess || (a.side_used && hipEventQuery(a.ev_side) != hipSuccess) ||
        (a.aux_used && hipEventQuery(a.ev_aux) != hipSuccess) || (a.leg_used && hipEventQuery(a.ev_leg) != hipSuccess)) {
        (void)hipGetLastError();  // hipErrorNotReady is not an error here
        return;
    }
    const int rc = complete_arena(h, ai);
    if (rc && !h->early_rc) h->early_rc = rc;
}
// Everything: the lagged arena first (it is the older one), then the current one.
int flush_pending(pe_engine* h)
{
    int rc = h->early_rc;
    h->early_rc = PE_OK;
    if (h->held.active) {  // launches held back for a next aggregate that is not coming now
        const int r = held_issue(h);
        if (r && !rc) rc = r;
    }
    for (int k = 1; k <= h->n_arenas; ++k) {  // oldest first, the current one last
        const int r = complete_arena(h, (h->cur + k) % h->n_arenas);
        if (r && !rc) rc = r;
    }
    return rc;
}

// Register a batch call's completion.  Outside a pipeline: wait now and run it (the call is synchronous, as
// include/posevo.h promises).  Inside one: advance the cursors and return; pe_pipeline_end waits once.
int finish_call(pe_engine* h, const Stage& st, const OutBlock& ob, std::function<int()> complete, bool force_sync)
{
    h->A().pending.push_back(std::move(complete));
    h->A().stage_cursor = st.end();
    h->A().out_cursor = ob.end();
    if (!h->pipelining || force_sync) return flush_pending(h);
    return PE_OK;
}

// A device buffer other enqueued work may still read: wait for that work before re-allocating it.
int ensure_quiesced(pe_engine* h, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return PE_OK;
    int rc = flush_pending(h);
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, b.ensure(bytes));
    return PE_OK;
}

// The same for a buffer every arena of the rotation holds: when the current arena's copy has to grow, everything enqueued is
// waited for once -- and then EVERY arena's copy grows, so that the steps of a stream look alike: an arena that grew at ITS
// first use drained the pipeline once per arena (a lag depth of 15 means sixteen such steps; the signed leg of round 6 ran at
// 1.5 ms per step until its three buffers grew together).
int ensure_quiesced_arenas(pe_engine* h, DevBuf pe_engine::PipeArena::*m, size_t bytes)
{
    if (bytes <= (h->A().*m).cap) return PE_OK;
    int rc = flush_pending(h);
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    ++h->arena_growths;
    HIP_TRY(h, (h->A().*m).ensure(bytes));
    const size_t cap = (h->A().*m).cap;  // with its head-room: the others get exactly that
    for (int i = 0; i < h->n_arenas; ++i) HIP_TRY(h, (h->arena[i].*m).ensure(cap, false, nullptr, true));
    return PE_OK;
}
// ... and for the staging / output blocks, which grow without a device-wide wait: the arenas nothing is in flight on grow with
// the current one (at the start of a stream that is all of them).
void grow_idle_arenas(pe_engine* h, size_t stage_bytes, size_t out_bytes)
{
    for (int i = 0; i < h->n_arenas; ++i) {
        pe_engine::PipeArena& o = h->arena[i];
        if (i == h->cur || !o.pending.empty() || o.fenced || o.fence_pending || o.stage_cursor || o.out_cursor) continue;
        if (h->held.active && h->held.arena == i) continue;
        if (stage_bytes) {  // exactly the current arena's capacities
            (void)o.h_stage.ensure(h->A().h_stage.cap, true);
            (void)o.d_stage.ensure(h->A().d_stage.cap, false, nullptr, true);
        }
        if (out_bytes) {
            (void)o.d_outblk.ensure(h->A().d_outblk.cap, false, nullptr, true);
            (void)o.h_pin.ensure(h->A().h_pin.cap, true);
        }
    }
    (void)hipGetLastError();
}

// Entry of a call that is not part of the pipelined hot path: complete whatever the batch calls left enqueued.
static int aux_quiesce(pe_engine* h)
{
    if (!h->aux_busy) return PE_OK;
    h->aux_busy = false;
    HIP_TRY(h, hipStreamSynchronize(h->aux_stream));
    return PE_OK;
}
int enter(pe_engine* h)
{
    (void)hipSetDevice(h->device);
    const int rc = flush_pending(h);
    const int rc2 = aux_quiesce(h);  // e.g. a participation rotation outside any batch call
    return rc ? rc : rc2;
}
// The same stream for work that depends on NOTHING the engine's stream holds (the participation rotation of a new epoch:
// it follows the previous flag passes on this very stream): no fork event -- every event record on the engine's stream is
// one more packet in the chain of small kernels that paces a streaming step.
hipStream_t state_stream_unordered(pe_engine* h)
{
    if (h->stream != h->own_stream || !h->aux_stream) return h->stream;
    h->aux_busy = true;
    h->A().aux_used = true;
    return h->aux_stream;
}
// The stream of the signature legs: the handle's own state-transition stream where it has one, else created at the first leg --
// at the LEAST priority: the runtime keeps a set of hardware queues per priority level, so the legs' ~1 ms decompressions get a
// queue that none of the engine's four hot streams shares (as a fifth normal-priority stream it landed in the tree's queue, and
// every G1 tree, flag pass and finish of four steps stood behind each decompression: profiles/r06_engine_timeline_signed_before.txt).
hipStream_t leg_stream(pe_engine* h)
{
    if (!h->aux_owned) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&h->aux_owned, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            h->aux_owned = nullptr;
            return h->aux_stream ? h->aux_stream : h->stream;
        }
    }
    return h->aux_owned;
}
hipStream_t state_stream_begin(pe_engine* h, bool reads_scratch, uint32_t rows_hint)
{
    // a caller-owned stream (pe_set_stream) carries everything: the caller orders its own work against it
    if (h->stream != h->own_stream || !h->aux_stream) return h->stream;
    // Tune::state_on = 2: WHICH of the G1 chain's streams carries the flag passes follows the size of the step.  They ride the
    // shorter of the two kernels that own a stream: the tree's (~75 us of latency whatever the size) beside an epoch-sized
    // accumulation (~210 us), the accumulation's beside a slot-sized one (~40 us: four adds per lane) -- there the tree's
    // stream with the flag passes on it (75 + 35 us) set the per-slot step's period (profiles/NOTES_r06.md 9).  Consecutive flag
    // passes are ordered (participation flags decide the reward numerators of the next): a change of stream goes through an event.
    if (h->tune.state_on == 2 && rows_hint && h->side_stream && h->fin_stream && h->ev_aux_switch) {
        hipStream_t want = rows_hint <= pe_engine::STATE_ON_SIDE_MAX_ROWS ? h->side_stream : h->fin_stream;
        if (want != h->aux_stream) {
            if (hipEventRecord(h->ev_aux_switch, h->aux_stream) == hipSuccess &&
                hipStreamWaitEvent(want, h->ev_aux_switch, 0) == hipSuccess)
                h->aux_stream = want;   // what the old stream held is now in front of everything that follows on the new one
            else
                (void)hipGetLastError();
        }
    }
    if (hipEventRecord(h->ev_aux_fork, h->stream) != hipSuccess ||
        hipStreamWaitEvent(h->aux_stream, h->ev_aux_fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        h->state_work_on_main = true;  // the next participation rotation must be ordered behind THIS stream
        return h->stream;
    }
    h->aux_busy = true;
    h->A().aux_used = true;
    if (reads_scratch) h->A().aux_reads_scratch = true;
    return h->aux_stream;
}
// An aggregate that rewrites the current arena's scratch (group descriptors, plan, member lists, resident union words)
// while an EARLIER call of the same pipeline still reads them on the state-transition stream (process_attestation's flag
// pass, the signature leg): order the rewriting stream behind that work (ADVICE r3 -- the same hazard the deferred G1
// launch had on the side stream, ev_join).  ev_aux_fork is free again here: its wait was enqueued when it was recorded.
int aux_join(pe_engine* h, hipStream_t ms)
{
    // (only such readers: the participation rotation of every step also lives on that stream and reads no scratch --
    // joining behind IT put every step's row chain behind the previous step's flag pass: 0.33 -> 0.86 ms per step)
    if (!h->aux_stream || ms == h->aux_stream || !h->A().aux_reads_scratch) return PE_OK;
    h->A().aux_reads_scratch = false;
    HIP_TRY(h, hipEventRecord(h->ev_aux_fork, h->aux_stream));
    HIP_TRY(h, hipStreamWaitEvent(ms, h->ev_aux_fork, 0));
    return PE_OK;
}
int need_init(pe_engine* h, bool flush, bool keep_held)
{
    if (!h) return PE_ERR_INVALID_ARG;
    if (!h->initialised) return fail(h, PE_ERR_STATE, "store not initialised: call pe_store_init first");
    (void)hipSetDevice(h->device);
    if (!keep_held && h->held.active) PE_TRY(held_issue(h));
    if (!flush) return PE_OK;
    const int rc = flush_pending(h);
    const int rc2 = aux_quiesce(h);
    return rc ? rc : rc2;
}

// ---- which streams share a hardware queue ------------------------------------------------------------------------------
// The runtime maps a process's streams onto GPU_MAX_HW_QUEUES (four) hardware queues per priority and a queue runs its
// packets in submission order: two of the handle's hot streams on one queue run strictly behind each other (round 5: a handle
// created beside another one, or after the host had created streams of its own, ran its finish kernel in its row chain's
// queue -- 308 us per slot-step instead of 120).  HIP does not say which queue a stream got, so the handle ASKS the device:
// a kernel that spins for ~60 us on one stream, a kernel that stamps the clock on each of the others -- a stamp taken after
// the spin ended sat in the spinner's queue.  pe_engine_create keeps creating streams until four of them lie on four different
// queues (at most ten; the rest are destroyed), ~0.3 ms once per handle; POSEVO_QUEUE_PROBE=0 skips it.
namespace {
__global__ void k_probe_spin(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    out[0] = wall_clock64();
}
__global__ void k_probe_mark(unsigned long long* out) { out[0] = wall_clock64(); }
}  // namespace

// cls[i] = the lowest index among the streams that share stream i's queue.  false: the probe itself failed (nothing is known).
bool probe_queue_classes(const std::vector<hipStream_t>& c, std::vector<int>& cls)
{
    const size_t n = c.size();
    cls.assign(n, -1);
    unsigned long long* out = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&out), 8 * (n + 1), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    bool ok = true;
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(1), 0, c[0], out + n, 1ull);   // code objects in, clocks up
    hipLaunchKernelGGL(k_probe_mark, dim3(1), dim3(1), 0, c[0], out);
    ok = hipStreamSynchronize(c[0]) == hipSuccess;
    for (size_t i = 0; i < n && ok; ++i) {
        if (cls[i] != -1) continue;
        cls[i] = (int)i;
        bool any = false;
        for (size_t j = i + 1; j < n; ++j) any = any || cls[j] == -1;
        if (!any) break;
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(1), 0, c[i], out + n, 6000ull);  // 60 us
        for (size_t j = i + 1; j < n; ++j)
            if (cls[j] == -1) hipLaunchKernelGGL(k_probe_mark, dim3(1), dim3(1), 0, c[j], out + j);
        for (size_t j = i; j < n; ++j)
            if (j == i || cls[j] == -1) ok = ok && hipStreamSynchronize(c[j]) == hipSuccess;
        for (size_t j = i + 1; j < n && ok; ++j)
            if (cls[j] == -1 && out[j] >= out[n]) cls[j] = (int)i;  // stamped after the spin had ended: behind it in its queue
    }
    if (hipGetLastError() != hipSuccess) ok = false;
    (void)hipHostFree(out);
    return ok;
}

}  // namespace posevo

extern "C" {

uint32_t pe_abi_version(void) { return PE_ABI_VERSION; }

void pe_config_default(pe_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->slots_per_epoch = 32;
    c->seconds_per_slot = 12;
    c->intervals_per_slot = 3;
    c->safe_slots_to_update_justified = 8;
    c->proposer_score_boost = 40;
    c->effective_balance_increment = 1000000000ull;
    c->min_attestation_inclusion_delay = 1;
    c->max_validators_per_committee = 2048;
    c->filter_slashed = 0;
    c->device = -1;
}

const char* pe_strerror(int status)
{
    switch (status) {
        case PE_OK: return "ok";
        case PE_ERR_INVALID_ARG: return "invalid argument";
        case PE_ERR_NO_DEVICE: return "no HIP device / HIP runtime failure";
        case PE_ERR_OOM: return "out of memory";
        case PE_ERR_UNKNOWN_PARENT: return "on_block: parent block unknown";
        case PE_ERR_FUTURE_BLOCK: return "on_block: block is from the future";
        case PE_ERR_NOT_AFTER_FINALIZED: return "on_block: block slot not after the finalized slot";
        case PE_ERR_NOT_FINALIZED_DESCENDANT: return "on_block: block does not descend from the finalized checkpoint";
        case PE_ERR_DUPLICATE_BLOCK: return "block already in the store";
        case PE_ERR_UNKNOWN_ROOT: return "unknown root";
        case PE_ERR_CAPACITY: return "capacity exceeded";
        case PE_ERR_NO_COMMITTEES: return "no committee table for the epoch";
        case PE_ERR_NOT_SLASHABLE: return "attestation data not slashable";
        case PE_ERR_INVALID_INDEXED: return "invalid indexed attestation";
        case PE_ERR_STATE: return "call sequence error";
        case PE_ERR_TIMEOUT: return "multi-GPU exchange timed out";
        default: return "unknown status";
    }
}
const char* pe_last_error(const pe_engine* h) { return h ? h->last_error.c_str() : ""; }

int pe_engine_create(const pe_config* cfg, pe_engine** out)
{
    if (!out) return PE_ERR_INVALID_ARG;
    *out = nullptr;
    pe_config c;
    if (cfg) c = *cfg; else pe_config_default(&c);
    if (c.slots_per_epoch == 0 || c.seconds_per_slot == 0 || c.intervals_per_slot == 0 ||
        c.effective_balance_increment == 0)
        return PE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return PE_ERR_NO_DEVICE;  // no CPU fallback
    int dev = c.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return PE_ERR_NO_DEVICE;
    }
    if (dev >= ndev) return PE_ERR_INVALID_ARG;
    if (hipSetDevice(dev) != hipSuccess) return PE_ERR_NO_DEVICE;
    pe_engine* h = new (std::nothrow) pe_engine();
    if (!h) return PE_ERR_OOM;
    h->cfg = c;
    h->device = dev;
    // Stream priorities: all normal.  The runtime maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by
    // default) per priority level, and how the engine's five streams fare depends on what else the process runs there:
    // with torch's NCCL process group (high-priority streams) in the process the next step's fork-choice chain is scheduled
    // behind the running accumulation (0.59 vs 0.37 ms/step, one rank over RCCL) -- bench.py keeps torch on gloo for that
    // reason.  Priorities inside the engine were measured in rounds 3-4 and dropped: engine stream high + accumulation low
    // repairs that case and ruins the plain one (0.82-0.85 vs 0.34 ms: the accumulation's waves get preempted); a row-chain
    // stream of its own -- at normal priority it shares a hardware queue with a long kernel and waits for it (0.49-0.62 ms),
    // in the high- or low-priority pool, or with a queue per stream (GPU_MAX_HW_QUEUES=8), the small kernels of two chains
    // interleave and each takes 3-5x longer (0.36-0.44 ms vs 0.27): DESIGN.md 3.4.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    auto mk = [&](hipStream_t* s) { return hipStreamCreateWithPriority(s, hipStreamNonBlocking, (prio_least + prio_greatest) / 2); };
    if (mk(&h->own_stream) != hipSuccess) {
        delete h;
        return PE_ERR_NO_DEVICE;
    }
    h->stream = h->own_stream;
    // Pipelined steps run three things at once: the fork-choice kernels of step N+1, k_g1_accumulate of step N and
    // k_g1_finish of step N-1/N, each on its own stream.  (CU-masked streams -- a private CU partition for the
    // latency-sized fork-choice kernels -- were measured and dropped: hipExtStreamCreateWithCUMask with 16 / 32 / 48
    // CUs taken out made k_g1_accumulate 1.6x / 1.0x / 5.8x slower, profiles/r02_cu_mask_sweep.txt.)
    // A handle creates FOUR streams (engine, side, fin, norm): the runtime gives a process's first four streams a hardware
    // queue each and lets every further one share (tools/qmap.py) -- with a fifth created here, a second handle
    // in the process shared queues with ITSELF.  The stream of the signature legs (and, with Tune::state_on = 0, of the flag
    // passes) is created when it is first needed: leg_stream().
    const bool ok_streams = mk(&h->side_stream) == hipSuccess && mk(&h->fin_stream) == hipSuccess &&
                            (h->tune.state_on != 0 || mk(&h->aux_stream) == hipSuccess);
    h->aux_owned = h->aux_stream;  // what pe_engine_destroy destroys (aux_stream becomes an alias below unless state_on is 0)
    // k_g1_finish has a stream of its own behind k_g1_tree: on one finishing stream the two latency-bound guests of a step
    // ran one behind the other and THAT stream set the step's period (0.43-0.47 -> 0.34 ms, round 3)
    if (hipEventCreateWithFlags(&h->ev_tree, hipEventDisableTiming) != hipSuccess ||
        (ok_streams && mk(&h->norm_stream) != hipSuccess)) {
        pe_engine_destroy(h);
        return PE_ERR_NO_DEVICE;
    }
    // The four hot streams on four different hardware queues, whatever the process created before this handle
    if (ok_streams && h->tune.queue_probe) {
        std::vector<hipStream_t> cand = {h->own_stream, h->side_stream, h->fin_stream, h->norm_stream};
        std::vector<int> cls;
        for (;;) {
            if (!probe_queue_classes(cand, cls)) break;
            std::vector<size_t> reps;
            for (size_t i = 0; i < cand.size(); ++i)
                if (cls[i] == (int)i) reps.push_back(i);
            if (reps.size() >= 4 || cand.size() >= 10) {
                if (reps.size() >= 4) {  // the first four queue representatives serve; what shares a queue with one of them goes
                    hipStream_t pick[4] = {cand[reps[0]], cand[reps[1]], cand[reps[2]], cand[reps[3]]};
                    for (hipStream_t st : cand)
                        if (st != pick[0] && st != pick[1] && st != pick[2] && st != pick[3]) (void)hipStreamDestroy(st);
                    h->own_stream = h->stream = pick[0];
                    h->side_stream = pick[1];
                    h->fin_stream = pick[2];
                    h->norm_stream = pick[3];
                    h->queues_distinct = true;
                } else {                 // fewer than four queues to be had (GPU_MAX_HW_QUEUES < 4?): the first four streams stay
                    for (size_t i = 4; i < cand.size(); ++i) (void)hipStreamDestroy(cand[i]);
                }
                break;
            }
            hipStream_t extra = nullptr;
            if (mk(&extra) != hipSuccess) { (void)hipGetLastError(); for (size_t i = 4; i < cand.size(); ++i) (void)hipStreamDestroy(cand[i]); break; }
            cand.push_back(extra);
        }
    }
    // Tune::state_on: the state-transition work on the tree's stream instead of its own (the runtime maps the engine's
    // streams onto four hardware queues; two streams that share one run in submission order)
    if (ok_streams && h->tune.state_on != 0) h->aux_stream = h->fin_stream;  // (2: state_stream_begin moves it by the step's size)
    if (!ok_streams ||
        hipEventCreateWithFlags(&h->ev_aux_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_aux_switch, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_acc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_sig, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        false) {
        pe_engine_destroy(h);
        return PE_ERR_NO_DEVICE;
    }
    for (auto& a : h->arena)
        if (hipEventCreateWithFlags(&a.ev_main, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.ev_side, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.ev_aux, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.ev_leg, hipEventDisableTiming) != hipSuccess) {
            pe_engine_destroy(h);
            return PE_ERR_NO_DEVICE;
        }
    if (const char* e = getenv("POSEVO_PIPELINE_LAG")) {
        const int lag = atoi(e);
        if (lag >= 1 && lag < pe_engine::MAX_ARENAS) h->n_arenas = lag + 1;
    }
    h->tables.reserve(c.max_committee_tables ? c.max_committee_tables : 4u);
    if (h->pairing) pair_kernels_preload();
    *out = h;
    return PE_OK;
}

void pe_engine_destroy(pe_engine* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)flush_pending(h);
    (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    if (h->fin_stream) (void)hipStreamSynchronize(h->fin_stream);
    if (h->norm_stream) (void)hipStreamSynchronize(h->norm_stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (h->aux_owned) (void)hipStreamSynchronize(h->aux_owned);
    if (h->prep_stream) { (void)hipStreamSynchronize(h->prep_stream); (void)hipStreamDestroy(h->prep_stream); }
    if (h->prof_base) (void)hipEventDestroy(h->prof_base);
    h->d_shuffle_scratch.release();
    h->d_points29.release();
    if (h->comm && rccl().ok) (void)rccl().CommDestroy(h->comm);
    if (h->comm_g1 && rccl().ok) (void)rccl().CommDestroy(h->comm_g1);
    h->d_xchg.release();
    h->d_xpart.release();
    h->d_xgather.release();
    for (auto& a : h->arena) {
        a.d_res_bits.release();
        a.d_res_info.release();
        a.d_partials.release();
        a.d_lane_partials.release();
        a.d_rr_tab.release();
        a.d_rr.release();
        for (DevBuf* b : {&a.x_rr_tab, &a.x_rr, &a.x_res_bits, &a.x_res_info, &a.d_xsend, &a.d_xrecv, &a.d_xrows, &a.d_xbits, &a.d_xn})
            b->release();
        a.d_sig_in.release();
        a.d_sig_pts.release();
        a.d_sig_status.release();
        a.d_stage.release();
        a.d_outblk.release();
        a.h_stage.release();
        a.h_pin.release();
        if (a.ev_main) (void)hipEventDestroy(a.ev_main);
        if (a.ev_side) (void)hipEventDestroy(a.ev_side);
        if (a.ev_leg) (void)hipEventDestroy(a.ev_leg);
        if (a.ev_aux) (void)hipEventDestroy(a.ev_aux);
    }
    for (DevBuf* b : {&h->d_points, &h->d_balance, &h->d_flags, &h->d_incr, &h->d_sbalance, &h->d_sflags, &h->d_vote_key, &h->d_vote_block, &h->d_vote_slot,
                      &h->d_part_cur, &h->d_part_prev, &h->d_tsize, &h->d_tparent, &h->d_trank, &h->d_tleaf,
                      &h->d_tpos, &h->d_tidx, &h->d_direct, &h->d_weights, &h->d_totals, &h->d_head,
                      &h->d_broot_tab, &h->d_broots, &h->d_bslot_pos,
                      &h->d_partials, &h->d_lane_partials, &h->d_out96, &h->d_tmp_points, &h->d_tmp_be, &h->d_tmp_points29})
        b->release();
    for (auto& t : h->tables) {
        t.d_members.release(); t.d_offsets.release(); t.d_inv_comm.release(); t.d_inv_pos.release();
        t.d_stage.release(); t.h_stage.release();
        if (t.ev_ready) (void)hipEventDestroy(t.ev_ready);
    }
    h->h_head.release();
    for (auto& p : h->prof)
        for (auto& ev : p.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (hipEvent_t ev : h->g1_tune_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->event_pool) (void)hipEventDestroy(ev);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_acc) (void)hipEventDestroy(h->ev_acc);
    if (h->acc_clock) (void)hipHostFree(h->acc_clock);
    if (h->ev_sig) (void)hipEventDestroy(h->ev_sig);
    if (h->ev_aux_fork) (void)hipEventDestroy(h->ev_aux_fork);
    if (h->ev_aux_switch) (void)hipEventDestroy(h->ev_aux_switch);
    if (h->ev_tree) (void)hipEventDestroy(h->ev_tree);
    if (h->ev_xchg) (void)hipEventDestroy(h->ev_xchg);
    if (h->norm_stream) (void)hipStreamDestroy(h->norm_stream);
    if (h->aux_owned) (void)hipStreamDestroy(h->aux_owned);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->fin_stream) (void)hipStreamDestroy(h->fin_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->trace.on && h->g1_tune_calls)
        fprintf(stderr, "[posevo host] accumulate autotune: 131072 slots %.3f ms, 65536 slots %.3f ms -> %u\n",
                h->g1_tune_best[0], h->g1_tune_best[1], h->g1_target_slots);
    if (h->trace.on)
        for (auto& kv : h->trace.acc) {
            std::vector<float> v = kv.second.all;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[posevo host] %-30s calls %5llu  avg %8.1f  min %8.1f  p50 %8.1f  max %8.1f us\n",
                    kv.first.c_str(), (unsigned long long)kv.second.n, kv.second.sum / kv.second.n, kv.second.mn,
                    v.empty() ? 0.0 : (double)v[v.size() / 2], kv.second.mx);
        }
    delete h;
}

int pe_set_stream(pe_engine* h, void* hip_stream)
{
    if (!h) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    (void)hipStreamSynchronize(h->stream);
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return PE_OK;
}

// ---------------------------------------------------------------- pipelined calls
int pe_pipeline_begin(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    HostLap lap(&h->trace);
    PE_TRY(complete_arena(h, h->cur));  // a lagged pipeline in the other arena stays in flight
    lap.mark("pipe.begin_1_complete_current");
    // Steps of a stream look alike: size this (idle) arena like the largest one now, instead of growing it call by call
    // inside the pipeline -- a growth there waits for everything enqueued and re-allocates pinned memory (milliseconds).
    {
        pe_engine::PipeArena& a = h->A();
        size_t stage = 0, out = 0, bits = 0, info = 0, part = 0, lane = 0;
        for (auto& o : h->arena) {
            stage = std::max(stage, std::min(o.d_stage.cap, o.h_stage.cap));
            out = std::max(out, std::min(o.d_outblk.cap, o.h_pin.cap));
            bits = std::max(bits, o.d_res_bits.cap);
            info = std::max(info, o.d_res_info.cap);
            part = std::max(part, o.d_partials.cap);
            lane = std::max(lane, o.d_lane_partials.cap);
        }
        if (a.d_stage.cap < stage || a.h_stage.cap < stage || a.d_outblk.cap < out || a.h_pin.cap < out || a.d_res_bits.cap < bits ||
            a.d_res_info.cap < info || a.d_partials.cap < part || a.d_lane_partials.cap < lane)
            ++h->arena_growths;
        // exact sizes: head-room here would make every arena overshoot the one it copies (see DevBuf::ensure)
        if (stage) { HIP_TRY(h, a.d_stage.ensure(stage, false, nullptr, true)); HIP_TRY(h, a.h_stage.ensure(stage, true)); }
        if (out) { HIP_TRY(h, a.d_outblk.ensure(out, false, nullptr, true)); HIP_TRY(h, a.h_pin.ensure(out, true)); }
        if (bits) HIP_TRY(h, a.d_res_bits.ensure(bits, false, nullptr, true));
        if (info) HIP_TRY(h, a.d_res_info.ensure(info, false, nullptr, true));
        if (part) HIP_TRY(h, a.d_partials.ensure(part, false, nullptr, true));
        if (lane) HIP_TRY(h, a.d_lane_partials.ensure(lane, false, nullptr, true));
    }
    lap.mark("pipe.begin_2_size_arena");
    h->A().table_stamp_at_begin = h->table_stamp;
    h->A().generation = ++h->pipes_begun;
    h->pipelining = true;
    return PE_OK;
}
int pe_pipeline_end(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    h->pipelining = false;
    h->streaming = false;
    HostLap lap(&h->trace);
    const int rc = flush_pending(h);
    lap.mark("pipe.end_wait_outputs");
    return rc;
}

int pe_pipeline_begin_streaming(pe_engine* h)
{
    const int rc = pe_pipeline_begin(h);
    if (rc == PE_OK) h->streaming = true;
    return rc;
}

int pe_pipeline_set_lag(pe_engine* h, uint32_t depth)
{
    if (!h || depth < 1 || depth >= (uint32_t)pe_engine::MAX_ARENAS) return PE_ERR_INVALID_ARG;
    if (h->pipelining) return fail(h, PE_ERR_STATE, "pe_pipeline_set_lag inside a pipeline");
    PE_TRY(enter(h));  // nothing is in flight afterwards: the rotation may change
    h->n_arenas = (int)depth + 1;
    h->cur = 0;
    // The arenas that join the rotation get what the others have grown to (a registry load pushes 100 MB through the staging
    // and output blocks before any caller sets its lag): sized HERE, not at each one's first pipeline inside a stream of steps
    // (7 ms of pinned allocations per arena: a 20-step run at lag 8 read 1.5 ms per step).
    {
        using A = pe_engine::PipeArena;
        DevBuf A::*dev[] = {&A::d_stage, &A::d_outblk, &A::d_res_bits, &A::d_res_info, &A::d_partials, &A::d_lane_partials,
                            &A::d_sig_in, &A::d_sig_pts, &A::d_sig_status};
        for (auto m : dev) {
            size_t cap = 0;
            for (auto& o : h->arena) cap = std::max(cap, (o.*m).cap);
            if (cap)
                for (int i = 0; i < h->n_arenas; ++i) HIP_TRY(h, (h->arena[i].*m).ensure(cap, false, nullptr, true));
        }
        PinBuf A::*pin[] = {&A::h_stage, &A::h_pin};
        for (auto m : pin) {
            size_t cap = 0;
            for (auto& o : h->arena) cap = std::max(cap, (o.*m).cap);
            if (cap)
                for (int i = 0; i < h->n_arenas; ++i) HIP_TRY(h, (h->arena[i].*m).ensure(cap, true));
        }
    }
    return PE_OK;
}
uint32_t pe_pipeline_get_lag(const pe_engine* h) { return h ? (uint32_t)h->n_arenas - 1 : 0; }
uint64_t pe_pipeline_generation(const pe_engine* h) { return h ? h->pipes_begun : 0; }
uint64_t pe_pipeline_completed(const pe_engine* h) { return h ? h->pipes_completed : 0; }

int pe_pipeline_end_lagged(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    h->pipelining = false;
    h->streaming = false;
    HostLap lap(&h->trace);
    pe_engine::PipeArena& a = h->A();
    const bool held = h->held.active && h->held.arena == h->cur;
    if (held) {
        // the step's fork-choice launches wait for the next aggregate (engine_pair.cpp); what the step launches BEHIND its
        // k_tree -- its G1 sums and this pipeline's marks on the engine's and the G1 streams -- goes with them
        for (auto& f : h->deferred) h->held.g1.push_back(std::move(f));
        h->deferred.clear();
        h->held.fence_pending = true;
        a.fence_pending = true;
    } else {
        PE_TRY(run_deferred(h));  // the step's G1 sums start now, behind its fork-choice kernels
        lap.mark("pipe.end_lagged_launch_g1");
        // mark the end of this pipeline on both streams; its completions run when the NEXT lagged end (or any
        // synchronous call) has waited for the marks
        PE_TRY(fence_arena(h, a));
    }
    h->aux_busy = false;    // accounted for by the fence
    a.fenced = true;
    h->side_busy = false;   // accounted for by the fence from here on
    h->cur = (h->cur + 1) % h->n_arenas;
    int rc = complete_arena(h, h->cur);  // the oldest pipeline still in flight (lag depth back): its arena is reused next
    if (!rc) rc = h->early_rc;           // ... unless pe_get_head found it ready and completed it already
    h->early_rc = PE_OK;
    lap.mark("pipe.end_lagged_wait_previous");
    return rc;
}

// ---------------------------------------------------------------- profiling
int pe_profile_arena_growths(const pe_engine* h, uint64_t* out)
{
    if (!h || !out) return PE_ERR_INVALID_ARG;
    *out = h->arena_growths;
    return PE_OK;
}
int pe_profile_queue_classes(pe_engine* h, int32_t out_class[4])
{
    if (!h || !out_class) return PE_ERR_INVALID_ARG;
    PE_TRY(enter(h));
    if (!h->side_stream || !h->fin_stream || !h->norm_stream) return fail(h, PE_ERR_STATE, "the handle has no stream set of its own");
    for (hipStream_t s : {h->own_stream, h->side_stream, h->fin_stream, h->norm_stream}) HIP_TRY(h, hipStreamSynchronize(s));
    std::vector<int> cls;
    if (!probe_queue_classes({h->own_stream, h->side_stream, h->fin_stream, h->norm_stream}, cls))
        return fail(h, PE_ERR_NO_DEVICE, "the queue probe failed");
    for (int i = 0; i < 4; ++i) out_class[i] = cls[i];
    return PE_OK;
}
int pe_profile_enable(pe_engine* h, int on)
{
    if (!h) return PE_ERR_INVALID_ARG;
    h->profiling = on != 0;
    h->prof_timeline = on == 2;
    h->prof_dominant_only = on == 3;
    if (h->prof_timeline && !h->prof_base && hipEventCreate(&h->prof_base) != hipSuccess) {
        h->prof_base = nullptr;
        h->prof_timeline = false;
    }
    if (h->profiling)
        while (h->event_pool.size() < 4096) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) break;
            h->event_pool.push_back(e);
        }
    return PE_OK;
}
static void prof_drain(pe_engine* h)
{
    (void)flush_pending(h);
    (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    if (h->fin_stream) (void)hipStreamSynchronize(h->fin_stream);
    if (h->norm_stream) (void)hipStreamSynchronize(h->norm_stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (h->aux_owned) (void)hipStreamSynchronize(h->aux_owned);
    for (int k = 0; k < PE_KERNEL_COUNT; ++k) {
        auto& p = h->prof[k];
        for (auto& ev : p.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                p.total_ms += ms;
                p.launches += 1;
                if (h->prof_timeline && h->prof_base && h->prof_tl.size() < (1u << 20)) {
                    float t0 = 0;
                    if (hipEventElapsedTime(&t0, h->prof_base, ev.first) == hipSuccess)
                        h->prof_tl.push_back(pe_engine::TimelineEntry{k, t0, ms});
                    else
                        (void)hipGetLastError();  // e.g. a launch bracketed before the reset that marked time zero
                }
            }
            h->event_pool.push_back(ev.first);
            h->event_pool.push_back(ev.second);
        }
        p.pending.clear();
    }
}
int pe_profile_reset(pe_engine* h)
{
    if (!h) return PE_ERR_INVALID_ARG;
    prof_drain(h);
    for (auto& p : h->prof) { p.launches = 0; p.total_ms = 0; }
    h->acc_launches = 0;  // the first accumulation behind a reset is bracketed (one in four is, engine_g1.cpp)
    h->acc_clock_n = 0;
    h->prof_tl.clear();
    if (h->prof_timeline && h->prof_base) HIP_TRY(h, hipEventRecord(h->prof_base, h->stream));  // time zero
    return PE_OK;
}
int pe_profile_accumulate_mhz(pe_engine* h, double* out_mhz, uint32_t cap, uint32_t* out_n)
{
    if (!h || !out_n || (cap && !out_mhz)) return PE_ERR_INVALID_ARG;
    prof_drain(h);
    const uint64_t n = std::min<uint64_t>(h->acc_clock_n, pe_engine::ACC_CLOCK_RING);
    const uint64_t first = h->acc_clock_n - n;  // the ring keeps the last ACC_CLOCK_RING launches
    *out_n = (uint32_t)n;
    for (uint64_t i = 0; i < n && i < cap; ++i) {
        const unsigned long long* r = h->acc_clock + 2 * ((first + i) % pe_engine::ACC_CLOCK_RING);
        out_mhz[i] = r[0] ? 100.0 * (double)r[1] / (double)r[0] : 0.0;  // 0: the launch had no workgroup 0 to report (empty plan)
    }
    return PE_OK;
}
int pe_profile_timeline(pe_engine* h, int32_t* kernel, double* start_ms, double* duration_ms, uint32_t cap, uint32_t* out_n)
{
    if (!h || !out_n || (cap && (!kernel || !start_ms || !duration_ms))) return PE_ERR_INVALID_ARG;
    prof_drain(h);
    *out_n = (uint32_t)h->prof_tl.size();
    for (uint32_t i = 0; i < cap && i < h->prof_tl.size(); ++i) {
        kernel[i] = h->prof_tl[i].kernel;
        start_ms[i] = h->prof_tl[i].start_ms;
        duration_ms[i] = h->prof_tl[i].dur_ms;
    }
    return PE_OK;
}
int pe_profile_get(pe_engine* h, int kernel, uint64_t* launches, double* total_ms)
{
    if (!h || kernel < 0 || kernel >= PE_KERNEL_COUNT) return PE_ERR_INVALID_ARG;
    prof_drain(h);
    if (launches) *launches = h->prof[kernel].launches;
    if (total_ms) *total_ms = h->prof[kernel].total_ms;
    return PE_OK;
}


}  // extern "C"
