// engine_internal.h -- shared by the host-side translation units of libposevo (engine_*.cpp): the handle, its
// buffers, the per-call staging machinery and the helpers more than one unit uses.  NOT part of the ABI (that is
// include/posevo.h); the device-side interface is kernels.h.
//
//   engine_core.cpp    lifecycle, error strings, completion of batch calls, pipelines, profiling hooks
//   engine_store.cpp   the fork-choice store (pe:889-901) and its handlers, registry, committee tables, get_head
//   engine_attest.cpp  on_attestation / process_attestation / aggregation over host rows
//   engine_resident.cpp the same path over rows resident in device memory (grouping + validation on the device)
//   engine_g1.cpp      G1 / G2 sums over caller-chosen groups, BLSPubkey / BLSSignature wire formats
//   engine_dist.cpp    multi-GPU exchange: RCCL owned by the engine, function-table collectives
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <map>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/posevo.h"
#include "../../include/posevo_profile.h"
#include "kernels.h"

namespace posevo {

using Root = std::array<uint8_t, 32>;
struct RootHash {
    size_t operator()(const Root& r) const noexcept
    {
        uint64_t h;
        memcpy(&h, r.data(), 8);  // roots are hash outputs: the first 8 bytes are already uniform
        return (size_t)h;
    }
};
inline Root to_root(const uint8_t* p)
{
    Root r;
    memcpy(r.data(), p, 32);
    return r;
}
inline bool is_zero_root(const Root& r)
{
    for (uint8_t b : r)
        if (b) return false;
    return true;
}
struct Checkpoint {
    uint64_t epoch = 0;
    Root root{};
    bool operator==(const Checkpoint& o) const { return epoch == o.epoch && root == o.root; }
};
struct Block {
    Root root;
    uint32_t parent;  // insertion index; NONE32 for the anchor
    uint64_t slot;
    Checkpoint post_justified, post_finalized;  // block_states[root].{current_justified,finalized}_checkpoint
};

// Growable device / pinned-host buffers.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // exact: no head-room -- for buffers sized to MATCH others (the arenas of a rotation): with head-room each arena that
    // catches up overshoots the one it copies, and the next one catches up with that (a free + malloc per arena and turn)
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t s = nullptr, bool exact = false)
    {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = exact ? bytes : std::max(bytes, cap + cap / 2);
        ncap = (ncap + 255) & ~size_t(255);
        void* np = nullptr;
        hipError_t e = hipMalloc(&np, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool exact = false)
    {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = ((exact ? bytes : std::max(bytes, cap * 2)) + 4095) & ~size_t(4095);
        void* np = nullptr;
        hipError_t e = hipHostMalloc(&np, ncap, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (p) (void)hipHostFree(p);
        p = np;
        cap = ncap;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct CommitteeTable {
    uint64_t epoch = 0;
    uint32_t n_committees = 0;
    std::vector<uint32_t> offsets;  // n_committees + 1
    DevBuf d_members;               // u32[offsets.back()]
    DevBuf d_offsets;               // u32[n_committees + 1]
    DevBuf d_inv_comm, d_inv_pos;   // partition tables only: validator -> (committee id, index in committee)
    bool is_partition = false;      // every validator in at most one committee (true for a real shuffling)
    uint64_t n_val_at_load = 0;     // registry size the inverse map was built for
    uint64_t stamp = 0;
    PinBuf h_stage;                 // pe_compute_committees_async: the call's seed | offsets | active indices ...
    DevBuf d_stage;                 // ... and their device copy
    hipEvent_t ev_ready = nullptr;  // pe_compute_committees_async: recorded behind the shuffle on the stream it ran on
    bool ready_pending = false;     // ... and not yet waited for by the engine's stream
};

// Diagnostic only (POSEVO_HOST_TRACE=1): wall time of host phases, printed at pe_engine_destroy.
struct HostTrace {
    bool on = std::getenv("POSEVO_HOST_TRACE") != nullptr;
    struct Acc { double sum = 0, mn = 1e30, mx = 0; uint64_t n = 0; std::vector<float> all; };
    std::map<std::string, Acc> acc;
    void add(const char* name, double us)
    {
        Acc& a = acc[name];
        a.sum += us;
        a.mn = std::min(a.mn, us);
        a.mx = std::max(a.mx, us);
        a.n += 1;
        a.all.push_back((float)us);
    }
};
struct HostScope {
    HostTrace* t;
    const char* name;
    std::chrono::steady_clock::time_point t0;
    HostScope(HostTrace* t_, const char* n) : t(t_), name(n)
    {
        if (t->on) t0 = std::chrono::steady_clock::now();
    }
    ~HostScope()
    {
        if (!t->on) return;
        t->add(name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
};

struct HostLap {  // lap timer: mark(name) charges the time since the previous mark to `name`
    HostTrace* t;
    std::chrono::steady_clock::time_point last;
    explicit HostLap(HostTrace* t_) : t(t_)
    {
        if (t->on) last = std::chrono::steady_clock::now();
    }
    void mark(const char* name)
    {
        if (!t->on) return;
        auto now = std::chrono::steady_clock::now();
        t->add(name, std::chrono::duration<double, std::micro>(now - last).count());
        last = now;
    }
};

struct KernelProfile {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    uint64_t launches = 0;
    double total_ms = 0;
};

}  // namespace posevo

using namespace posevo;  // internal header: every unit that includes it is host code of this library

struct pe_engine {
    pe_config cfg{};
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string last_error;

    // ---- store scalars (pe:889-897) ----
    bool initialised = false;
    uint64_t time = 0, genesis_time = 0;
    Checkpoint justified, finalized, best_justified;
    Root boost_root{};
    std::vector<Block> blocks;
    std::unordered_map<Root, uint32_t, RootHash> index_of;

    // ---- validators (T1/T2) ----
    uint64_t n_val = 0;
    bool have_points = false;
    DevBuf d_points, d_balance, d_flags, d_incr, d_vote_key, d_vote_block, d_part_cur, d_part_prev;
    DevBuf d_vote_slot;  // vote-expiry variant only (cfg.vote_expiry_slots > 0): slot of each latest message
    DevBuf d_sbalance, d_sflags;  // working-state view (process_attestation rewards, FFG sums)
    bool state_view_set = false;  // false: the working state mirrors the pe_set_validators data
    std::vector<uint8_t> h_flags;  // host mirror (equivocating bit is OR-ed in here)

    // ---- tree snapshot (pre-order) ----
    bool tree_dirty = true;
    DevBuf d_tsize, d_tparent, d_trank, d_tleaf, d_tpos, d_tidx, d_direct, d_weights, d_totals, d_head;
    DevBuf d_broot_tab, d_broots, d_bslot_pos;  // device-side block lookups (root -> index hash table, roots, slot by position)
    uint32_t broot_mask = 0;
    std::vector<uint32_t> h_pos_of_idx;
    PinBuf h_head;  // 64 B of host-coherent pinned memory the tree kernel writes the head index into
    uint32_t votes_grid = 0;  // workgroups of the last k_votes launch on the engine's own buffers

    // ---- committees ----
    std::vector<CommitteeTable> tables;
    uint64_t table_stamp = 0;
    size_t last_table = 0;  // index of the table find_table returned last

    // ---- scratch ----
    DevBuf d_partials, d_lane_partials, d_out96, d_tmp_points, d_tmp_be;

    // ---- pipelined calls (pe_pipeline_begin / _end): one wait per step instead of one per call ----
    // A batch call lays out its inputs at stage_cursor / its outputs at out_cursor of the current arena, enqueues
    // copies + kernels and registers a completion (results pinned block -> caller buffers).  Outside a pipeline the
    // call then waits and runs it; inside one the cursors just advance and pe_pipeline_end (or any other synchronous
    // entry point) waits once for everything.  Two arenas: pe_pipeline_end_lagged fences the current one and hands
    // the next pipeline the other, so that a step's G1 sums may still run while the host prepares the next step.
    struct PipeArena {
        DevBuf d_stage, d_outblk;         // H2D staging block | device output block
        PinBuf h_stage, h_pin;            // their pinned host mirrors (h_pin is host-coherent: kernels write into it)
        DevBuf d_res_bits, d_res_info;    // resident hand-over: OR-ed bit words | {popcount, overlap} per group
        DevBuf d_partials, d_lane_partials;  // tree -> finish | accumulate -> tree hand-over of this arena's pipelined aggregate
        // rows resident on the device (engine_resident.cpp): grouping table (self-cleaning) and the per-call scratch
        DevBuf d_rr_tab, d_rr;
        DevBuf d_sig_in, d_sig_pts, d_sig_status;  // pe_aggregate_signed: wire bytes | Montgomery points | decode status
        uint32_t rr_rows_cap = 0, rr_comm_cap = 0, rr_tab_size = 0;
        // a second set of the same for the EXCHANGED aggregate of a committee-sharded step (pe_aggregate_exchange): the
        // local aggregate's descriptors and unions are still being read by its G1 chain when the gathered one is ingested
        DevBuf x_rr_tab, x_rr, x_res_bits, x_res_info;
        uint32_t x_rows_cap = 0, x_comm_cap = 0, x_tab_size = 0;
        DevBuf d_xsend, d_xrecv, d_xrows, d_xbits, d_xn;  // packed local groups | all ranks' | unpacked rows | bits | count
        size_t stage_cursor = 0, out_cursor = 0;
        uint64_t table_stamp_at_begin = 0;  // h->table_stamp when the pipeline that fills this arena began
        uint64_t generation = 0;            // ordinal of the pipeline that fills this arena (pe_pipeline_generation)
        std::vector<std::function<int()>> pending;
        hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_aux = nullptr;  // recorded by pe_pipeline_end_lagged
        hipEvent_t ev_leg = nullptr;   // the end of this arena's signature leg on the legs' stream (sig_batch_flush)
        bool leg_used = false;         // ... recorded and not yet waited for
        bool fenced = false, side_used = false, aux_used = false;
        bool fence_pending = false;  // fenced by pe_pipeline_end_lagged, but ev_main / ev_side are still to be recorded: behind the
                                     // pipeline's held-back fork-choice launches (engine_pair.cpp)
        bool aux_reads_scratch = false;  // work on the state-transition stream still reads this arena's grouping scratch / resident words
    };
    // lag depth L (pe_pipeline_set_lag, default 2) = L + 1 arenas in rotation: a lagged end waits for the pipeline L
    // back, never for the finish kernel of the one that has only just been fenced.  Arenas allocate on first use.
    static constexpr int MAX_ARENAS = 16;
    int n_arenas = 3;
    PipeArena arena[MAX_ARENAS];
    int cur = 0;
    PipeArena& A() { return arena[cur]; }
    bool pipelining = false;
    uint64_t pipes_begun = 0, pipes_completed = 0;  // pe_pipeline_generation / pe_pipeline_completed
    hipStream_t side_stream = nullptr;  // k_g1_accumulate of a pipelined pe_aggregate runs here, beside the fork-choice kernels
    uint64_t arena_growths = 0;         // (re)allocations of arena buffers since the handle was created (pe_profile_arena_growths):
                                        // a stream of like steps must not make any after its first one
    bool queues_distinct = false;       // the probe at creation found engine / side / fin / norm on four different hardware queues
    hipStream_t fin_stream = nullptr;   // ... its k_g1_tree here, beside the NEXT aggregate's accumulation
    // ... and its k_g1_finish here: on the tree's stream the two latency-bound guests of a step ran one behind the other
    // (tree 250-300 us beside an accumulation + finish 130 us), and that stream, not the accumulation, set the period
    // of a streaming run once the host was off the critical path (profiles/r03_timeline_fin_stream.txt)
    hipStream_t norm_stream = nullptr;
    hipEvent_t ev_tree = nullptr;       // tree done -> finish may start
    hipStream_t g1_tail() const { return norm_stream ? norm_stream : fin_stream; }  // where a side-stream G1 chain ends
    // State-transition work (process_attestation's flag pass, participation rotation) runs on a stream of its own: it
    // follows the head in a step, and on the engine's stream it stood between one step's head and the next step's
    // fork-choice chain (25 + 85 us per 1 M validators beside a running accumulation, profiles/r03_timeline_*.txt).
    hipStream_t aux_stream = nullptr;
    hipStream_t aux_owned = nullptr;    // the stream created for it (aux_stream may alias another one: Tune::state_on)
    hipStream_t prep_stream = nullptr;  // pe_compute_committees_async: next epochs' shuffles, beside everything else
    DevBuf d_shuffle_scratch;           // ... and their hash tables (the synchronous call uses d_tmp_be)
    hipEvent_t ev_aux_fork = nullptr;
    hipEvent_t ev_aux_switch = nullptr;  // state_stream_begin: the state-transition work changes its stream (Tune::state_on = 2)
    static constexpr uint32_t STATE_ON_SIDE_MAX_ROWS = 1024;  // steps over at most this many rows: flag passes on the accumulation's stream
    bool aux_busy = false;              // the aux stream holds work nobody has waited for yet
    bool state_work_on_main = false;    // state_stream_begin fell back to the engine's stream for a flag pass
    hipEvent_t ev_acc = nullptr;        // accumulate done -> finish may start
    // pe_aggregate_signed: the signature legs of a streaming caller's steps, collected until Tune::sig_batch of them go out
    // behind ONE decompression launch (engine_g1.cpp: sig_batch_flush)
    struct SigSeg {
        int arena;                      // the arena (pipeline) whose pinned block takes the leg's outputs
        uint32_t n;
        bool compressed, check_subgroup;
        bool streaming;                 // collected in a streaming step whose accumulations run on the side stream
        const uint8_t* d_in;            // wire bytes on the device (the caller's, or the arena's scratch)
        const uint8_t* copy_from;       // device-resident signatures at an address the kernel cannot read in place: copied first
        size_t bytes;
        uint32_t* d_pts; int32_t* d_status;
        const UnionGroup* d_ug; const uint32_t* d_member_row; uint32_t ng_bound; const AttPlan* plan_dev;
        uint8_t* o_sig; uint32_t* o_bad; int32_t* o_st;
    };
    std::vector<SigSeg> sig_batch;
    hipEvent_t ev_sig = nullptr;        // a batch's decompression is done (on the accumulations' stream) -> its sums may start
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_busy = false;             // the side stream holds work of the current arena nobody has waited for yet
    bool side_ever = false;             // ev_join has been recorded at least once
    bool streaming = false;             // pe_pipeline_begin_streaming: G1 launches are deferred to the pipeline's end
    std::vector<std::function<int()>> deferred;  // ... these
    // ---- the fork-choice launches of a streaming step, held back for the next aggregate (engine_pair.cpp) ----
    // In a streaming pipeline over rows in device memory pe_on_attestation_batch and pe_get_head_async do not launch: their
    // kernels' argument blocks wait here until the NEXT step's pe_aggregate arrives and are then launched pairwise with its
    // row kernels (pair_kernels.hip) -- or alone, in order, the moment anything else needs the stream (held_issue).
    struct HeldFc {
        bool active = false;
        int arena = -1;               // the arena (pipeline) the launches belong to
        bool have_fc = false;         // validate + lmd (pe_on_attestation_batch)
        bool have_head = false;       // votes + tree (pe_get_head_async)
        ValidateFcArgs validate{};
        LmdVmArgs lmd{};
        VotesArgs votes{};
        TreeArgs tree{};
        std::function<int()> between;         // a sharded head: the weights' all-reduce, between the votes and the tree
        std::vector<std::function<int()>> g1;  // the step's G1 launches: behind its k_tree, as without the holding
        bool fence_pending = false;   // pe_pipeline_end_lagged has closed the pipeline: its fence follows these launches
    } held;
    bool pairing = std::getenv("POSEVO_PAIR") == nullptr || std::atoi(std::getenv("POSEVO_PAIR")) != 0;
    // ---- scheduling knobs of the streaming G1 chain, read once per handle ----
    struct Tune {
        static int env(const char* name, int dflt)
        {
            const char* e = std::getenv(name);
            return e && *e ? std::atoi(e) : dflt;
        }
        // at most one accumulation workgroup per CU by an LDS request, the tree one per CU by registers (g1_kernels.hip)
        int exclusive = env("POSEVO_ACC_EXCLUSIVE", 1);
        // which stream carries the state-transition work: 0 = its own (aux), 1 = the tree's (fin), 2 = the tree's for steps over
        // more than 1024 rows and the accumulation's for smaller ones (state_stream_begin)
        int state_on = env("POSEVO_STATE_ON", 2);
        // pe_aggregate_signed in streaming steps: how many steps' signature legs share one decompression launch (1 = a launch
        // per step, round 5's shape; at most G2_BATCH_MAX)
        int sig_batch = env("POSEVO_SIG_BATCH", 8);
        // pe_engine_create asks the device which of its streams share a hardware queue and keeps four that do not (engine_core.cpp)
        int queue_probe = env("POSEVO_QUEUE_PROBE", 1);
    } tune;

    // ---- device-resident hand-over of the last pe_aggregate (PE_BITS_RESIDENT) ----
    // tag: a fold of the group's AttestationData -- a row handed over as resident must BE a row of the resident aggregate,
    // not merely sit at the same offset as one (two aggregates of equal shape lay their unions out alike)
    struct ResGroup { uint32_t byte_off, n_bits, word, sig_valid, tag; };
    std::vector<ResGroup> res_groups;     // sorted by byte_off (= group order)
    std::shared_ptr<std::vector<uint32_t>> res_info_host;  // copy of d_res_info, filled when that aggregate completes
    bool res_valid = false;
    int res_arena = 0;                    // which arena holds the resident bits
    uint64_t res_generation = 0;          // which pe_aggregate the resident data belongs to

    // ---- the last pe_aggregate over rows in DEVICE memory (PE_ROWS_RESIDENT hand-over to the handlers) ----
    struct ResidentRows {
        bool valid = false;
        int arena = 0;
        uint32_t n_in = 0;              // input rows = upper bound of the groups formed
        const void* rows = nullptr;     // the caller's device rows (unchanged until the pipeline completes)
        TablesDev tables{};             // the candidate committee tables the groups were resolved against
        uint64_t generation = 0;
        int set = 0;                    // 0: the arena's own scratch, 1: the exchanged aggregate's (x_*)
    } rr;

    // ---- accumulate-shape autotune (large pubkey aggregations) ----
    // 131072 task slots (two waves per SIMD, 6 tree levels at 512-member committees) or 65536 (one wave, 5 levels):
    // which one is faster depends on the box (fast boxes: one wave/SIMD by ~8 %, slow boxes: two by ~3 %;
    // profiles/r01_g1_phases_k8.txt / _k16.txt).  The first four large calls alternate A, B, A, B under a pair of
    // events, the better minimum is kept.
    int g1_tune_calls = 0;          // trials done so far (4 = decided)
    float g1_tune_best[2] = {1e30f, 1e30f};
    uint32_t g1_target_slots = 0;   // 0 = undecided
    hipEvent_t g1_tune_ev[2] = {nullptr, nullptr};
    // k_g1_accumulate reads the registry in the S29 field form: d_points29, built from d_points where the registry is loaded
    // (build_points29); d_tmp_points29: the same for caller-supplied points (d_tmp_points, `caller_rows` rows), per call
    bool points29_valid = false;
    DevBuf d_points29, d_tmp_points29;

    // ---- RCCL inside the engine (pe_dist_*): one communicator per handle, collectives on the engine's stream ----
    ncclComm_t comm = nullptr, comm_g1 = nullptr;  // get_head's all-reduce (engine stream) | the G1 partials' all-gather
    int early_rc = PE_OK;                           // status of a completion run ahead of its pipeline end (complete_oldest_if_ready)
    uint32_t xchg_blocks = 0;                       // block count / registry size d_xchg was last laid out for
    uint64_t xchg_nval = 0;
    bool last_agg_on_side = false;                  // the last aggregate's G1 chain went to the side / finishing streams
    int dist_rank = 0, dist_world = 1;
    DevBuf d_xchg, d_xpart, d_xgather;  // weights exchange | this rank's G1 partials | all ranks' partials
    pe_collectives coll{};              // pe_dist_init_custom: the caller's collectives instead of RCCL
    bool coll_custom = false;
    bool dist_single_comm = false;      // PE_DIST_SINGLE_COMM: both collectives on `comm`, on the engine's stream
    bool dist_wedged = false;           // a bounded wait expired: the communicators were aborted
    uint32_t dist_timeout_ms = 30000;   // bounded waits once the handle exchanges with other ranks (0 = unbounded)
    uint32_t dist_max_groups = 0;       // pe_dist_set_max_groups (0 = the row count of the call)
    hipEvent_t ev_xchg = nullptr;       // single-communicator mode: G1 chain <-> engine stream hand-over
    bool dist_ready() const { return comm != nullptr || coll_custom; }

    // ---- profiling ----
    bool profiling = false;
    uint64_t acc_launches = 0;  // k_g1_accumulate launches: totals mode brackets every 4th one (launch_g1_planned)
    // the shader clock of the accumulations launched while profiling is on (pe_profile_accumulate_mhz): a ring of
    // {100 MHz ticks, shader cycles} pairs in host memory, written by workgroup 0 of each launch
    static constexpr uint32_t ACC_CLOCK_RING = 4096;
    unsigned long long* acc_clock = nullptr;
    uint64_t acc_clock_n = 0;
    KernelProfile prof[PE_KERNEL_COUNT];
    // pe_profile_enable(h, 2): also a timeline of the bracketed launches (start relative to prof_base, duration)
    bool prof_timeline = false;
    bool prof_dominant_only = false;  // pe_profile_enable(h, 3): bracket k_g1_accumulate only (every bracket is two event
                                        // packets on a stream of latency-sized kernels, and host time)
    hipEvent_t prof_base = nullptr;
    struct TimelineEntry { int32_t kernel; float start_ms, dur_ms; };
    std::vector<TimelineEntry> prof_tl;
    std::vector<hipEvent_t> event_pool;
    HostTrace trace;
};

namespace posevo {

// ------------------------------------------------------------------ errors
int fail(pe_engine* h, int code, const std::string& msg);
int hip_fail(pe_engine* h, hipError_t e, const char* what);
#define HIP_TRY(h, expr)                                        \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) return hip_fail((h), _e, #expr);  \
    } while (0)
#define PE_TRY(expr)              \
    do {                          \
        const int _rc = (expr);   \
        if (_rc) return _rc;      \
    } while (0)

struct ProfScope {
    pe_engine* h;
    int k;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t take(pe_engine* h)  // hipEventCreate costs ~5 us: recycle (prof_drain returns them)
    {
        if (!h->event_pool.empty()) { hipEvent_t e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
    ProfScope(pe_engine* h_, int k_, hipStream_t s_ = nullptr, bool sampled_out = false) : h(h_), k(k_), s(s_ ? s_ : h_->stream)
    {
        if (!h->profiling || sampled_out) return;
        if (h->prof_dominant_only && k != PE_KERNEL_G1_ACCUMULATE) return;  // mode 3: the roofline's kernel alone
        if ((k == PE_KERNEL_ATT_GROUP || k == PE_KERNEL_ATT_VALIDATE) && !h->prof_timeline) return;  // timeline-only brackets
        a = take(h);
        b = take(h);
        if (!a || !b) { a = b = nullptr; return; }
        (void)hipEventRecord(a, s);
    }
    ~ProfScope()
    {
        if (!a) return;
        (void)hipEventRecord(b, s);
        h->prof[k].pending.emplace_back(a, b);
    }
};

// ------------------------------------------------------------------ completion of batch calls (engine_core.cpp)
int run_deferred(pe_engine* h);
int complete_arena(pe_engine* h, int ai);
void complete_oldest_if_ready(pe_engine* h);
int flush_pending(pe_engine* h);
int enter(pe_engine* h);
// keep_held: the call does not touch what the held-back fork-choice launches read or write and enqueues nothing that must
// follow them (host-side scalars; the calls that hold or pair themselves): everything else issues them first
int need_init(pe_engine* h, bool flush = true, bool keep_held = false);
// ---- held-back fork-choice launches (engine_pair.cpp)
bool hold_eligible(const pe_engine* h);
int held_issue(pe_engine* h);  // each kernel alone, in order, then what waits behind the step's k_tree
// the row kernels of an aggregate, pairwise with the held launches of the previous step where there are any
int launch_rows_paired(pe_engine* h, const IngestArgs& ia, const AttPlanArgs& pa, const MembersArgs& ma, const UnionArgs& ua);
int fence_arena(pe_engine* h, pe_engine::PipeArena& a);  // ev_main / ev_side of a lagged pipeline's end
// get_head's tree launch as an argument block (engine_store.cpp)
int tree_args(pe_engine* h, uint64_t* d_direct, const VoteTotals* d_totals, int clear_direct, uint32_t* head_word, TreeArgs* out);
VotesArgs votes_args(const pe_engine* h);
// The stream state-transition work goes to, ordered behind everything enqueued on the engine's stream so far.
hipStream_t state_stream_begin(pe_engine* h, bool reads_scratch = false, uint32_t rows_hint = 0);
hipStream_t leg_stream(pe_engine* h);              // where a signature leg runs (created at the first one)
bool probe_queue_classes(const std::vector<hipStream_t>& streams, std::vector<int>& cls);  // which share a hardware queue
int sig_batch_flush(pe_engine* h);                 // launch the signature legs collected so far (engine_g1.cpp)
bool sig_batch_holds(const pe_engine* h, int arena);  // ... one of which writes into this arena's output block
hipStream_t state_stream_unordered(pe_engine* h);  // the same stream, not ordered behind the engine's
int aux_join(pe_engine* h, hipStream_t ms);  // ms waits for what this pipeline put on the state-transition stream

// ------------------------------------------------------------------ spec helpers (A.10)
inline uint64_t current_slot(const pe_engine* h) { return (h->time - h->genesis_time) / h->cfg.seconds_per_slot; }
inline uint64_t epoch_at_slot(const pe_engine* h, uint64_t slot) { return slot / h->cfg.slots_per_epoch; }
// Vote-expiry variant (RLMD-GHOST, pe:1585-1596; eta = 1 is Goldfish's GHOST-Eph, pe:1549): only latest messages from
// the most recent eta slots count, i.e. message.slot + eta >= current_slot.  eta = 0 disables it (LMD-GHOST).
inline const uint32_t* expiry_slots_ptr(const pe_engine* h)
{
    return h->cfg.vote_expiry_slots ? h->d_vote_slot.as<uint32_t>() : nullptr;
}
inline uint32_t min_vote_slot(const pe_engine* h)
{
    const uint64_t cur = current_slot(h), eta = h->cfg.vote_expiry_slots;
    return (uint32_t)(cur > eta ? cur - eta : 0);
}
inline uint64_t start_slot(const pe_engine* h, uint64_t epoch) { return epoch * h->cfg.slots_per_epoch; }
inline uint64_t slots_since_epoch_start(const pe_engine* h, uint64_t slot) { return slot % h->cfg.slots_per_epoch; }

inline bool find_block(const pe_engine* h, const Root& r, uint32_t* idx)
{
    auto it = h->index_of.find(r);
    if (it == h->index_of.end()) return false;
    *idx = it->second;
    return true;
}
// get_ancestor (A.2): walk parent links while block.slot > slot.
inline uint32_t get_ancestor(const pe_engine* h, uint32_t idx, uint64_t slot)
{
    while (h->blocks[idx].slot > slot && h->blocks[idx].parent != NONE32) idx = h->blocks[idx].parent;
    return idx;
}
inline uint64_t isqrt64(uint64_t n)
{
    uint64_t x = n, y = (x + 1) / 2;
    while (y < x) { x = y; y = (x + n / x) / 2; }
    return x;
}

// ------------------------------------------------------------------ store (engine_store.cpp)
int refresh_tree(pe_engine* h);
TreeDev tree_dev(const pe_engine* h);
int insert_block(pe_engine* h, const Root& root, uint32_t parent, uint64_t slot, const Checkpoint& pj, const Checkpoint& pf);
CommitteeTable* find_table(pe_engine* h, uint64_t epoch);
int run_tree(pe_engine* h, uint64_t* d_direct, const VoteTotals* d_totals, int clear_direct, uint32_t* head_out,
             uint32_t* async_word = nullptr);
int ensure_validator_arrays(pe_engine* h, uint64_t n);
int upload_balances(pe_engine* h, uint64_t n, const uint64_t* bal, const uint8_t* flags);

// Re-pack one attestation's bits into 32-bit words (zero padded, masked to n_use bits); returns popcount.
uint32_t pack_bits(const uint8_t* src, uint32_t n_use, uint32_t* dst_words);

// ------------------------------------------------------------------ staging
// One pinned host block mirrored by one device block: a call lays out everything the kernels need (bit words,
// rows, group descriptors) in the pinned block, uploads it with ONE hipMemcpyAsync, and reads results back
// from one device output block with ONE copy.  (Separate pageable copies cost 30-50 us each on this box.)
// Offsets are relative to the block's cursor at the start of the call: inside a pipeline consecutive calls take
// consecutive regions (nothing an enqueued copy or kernel still needs is overwritten); outside one the cursor is 0.
// Growing a block re-allocates it, so a call reserves BEFORE it takes pointers, and a reservation that has to grow a
// block with enqueued work behind it first waits for that work (flush_pending).
void grow_idle_arenas(pe_engine* h, size_t stage_bytes, size_t out_bytes);  // engine_core.cpp
struct Stage {
    pe_engine* h;
    size_t base, used = 0;
    explicit Stage(pe_engine* h_) : h(h_), base((h_->A().stage_cursor + 255) & ~size_t(255)) {}
    int reserve(size_t bytes)
    {
        bytes += 4096;
        if (base + bytes > h->A().h_stage.cap || base + bytes > h->A().d_stage.cap) {
            // does not fit behind the calls already enqueued: wait for them once, then make room for two such steps
            // so that the next pipeline does not wait again
            const size_t want = base ? 2 * (base + bytes) : bytes;
            int rc = complete_arena(h, h->cur);
            if (rc) return rc;
            base = 0;
            hipError_t e = h->A().h_stage.ensure(want);
            if (e == hipSuccess) e = h->A().d_stage.ensure(want);
            if (e != hipSuccess) return hip_fail(h, e, "staging block");
            ++h->arena_growths;
            grow_idle_arenas(h, want, 0);
        }
        return PE_OK;
    }
    size_t alloc(size_t bytes)
    {
        const size_t off = (used + 255) & ~size_t(255);
        used = off + bytes;
        return off;
    }
    bool overflow() const { return base + used > h->A().h_stage.cap || base + used > h->A().d_stage.cap; }
    template <typename T> T* host(size_t off) const { return reinterpret_cast<T*>(h->A().h_stage.as<uint8_t>() + base + off); }
    template <typename T> T* dev(size_t off) const { return reinterpret_cast<T*>(h->A().d_stage.as<uint8_t>() + base + off); }
    hipError_t upload() const
    {
        if (used == 0) return hipSuccess;
        return hipMemcpyAsync(h->A().d_stage.as<uint8_t>() + base, h->A().h_stage.as<uint8_t>() + base, used,
                              hipMemcpyHostToDevice, h->stream);
    }
    size_t end() const { return base + used; }
};
struct OutBlock {  // device output block + pinned landing zone with the same layout
    pe_engine* h;
    size_t base, used = 0;
    explicit OutBlock(pe_engine* h_) : h(h_), base((h_->A().out_cursor + 255) & ~size_t(255)) {}
    size_t alloc(size_t bytes)
    {
        const size_t off = (used + 255) & ~size_t(255);
        used = off + bytes;
        return off;
    }
    int ensure()  // after the allocs, before anything of this call is enqueued
    {
        const size_t need = base + used + 256;
        if (need > h->A().d_outblk.cap || need > h->A().h_pin.cap) {
            const size_t want = base ? 2 * need : need;
            int rc = complete_arena(h, h->cur);
            if (rc) return rc;
            base = 0;
            hipError_t e = h->A().d_outblk.ensure(want);
            if (e == hipSuccess) e = h->A().h_pin.ensure(want);
            if (e != hipSuccess) return hip_fail(h, e, "output block");
            ++h->arena_growths;
            grow_idle_arenas(h, 0, want);
        }
        return PE_OK;
    }
    template <typename T> T* dev(size_t off) const { return reinterpret_cast<T*>(h->A().d_outblk.as<uint8_t>() + base + off); }
    template <typename T> T* host(size_t off) const { return reinterpret_cast<T*>(h->A().h_pin.as<uint8_t>() + base + off); }
    hipError_t download(hipStream_t s = nullptr) const  // whole region of this call
    {
        if (used == 0) return hipSuccess;
        return hipMemcpyAsync(h->A().h_pin.as<uint8_t>() + base, h->A().d_outblk.as<uint8_t>() + base, used,
                              hipMemcpyDeviceToHost, s ? s : h->stream);
    }
    hipError_t download(size_t off, size_t bytes, hipStream_t s = nullptr) const
    {
        if (bytes == 0) return hipSuccess;
        return hipMemcpyAsync(h->A().h_pin.as<uint8_t>() + base + off, h->A().d_outblk.as<uint8_t>() + base + off, bytes,
                              hipMemcpyDeviceToHost, s ? s : h->stream);
    }
    size_t end() const { return base + used; }
};

// A streaming pipeline holds its G1 sums back and launches them BEHIND the step's k_tree (pe_get_head / _async), not with
// the aggregate: queued behind its predecessor on the side stream an accumulation's workgroups are dispatched as CUs free
// up, the first CUs to finish take two of them (2 x 232 registers fit) and their waves run at half speed -- 335-355 us
// instead of 205-240 in five steps of twenty (tools/engine_timeline.py --cold 20; the same with the launch in front of
// k_tree or k_votes) -- and every kernel of the engine stream's chain then runs as the guest of an accumulation.
// No G1 chain of an earlier step can still be running: every arena has been completed (the engine was drained) and this
// pipeline has launched none.  The first step of a run then launches its accumulation with the aggregate instead of holding
// it back behind k_tree: there is no predecessor on the side stream whose retiring workgroups it could pile onto, and a run
// that starts from an idle device begins ~90 us earlier (tools/engine_timeline.py --cold 20).
inline bool g1_chain_idle(const pe_engine* h)
{
    for (int i = 0; i < h->n_arenas; ++i)
        if (h->arena[i].side_used) return false;
    return h->deferred.empty();
}

// Register a batch call's completion.  Outside a pipeline: wait now and run it (the call is synchronous, as
// include/posevo.h promises).  Inside one: advance the cursors and return; pe_pipeline_end waits once.
int finish_call(pe_engine* h, const Stage& st, const OutBlock& ob, std::function<int()> complete, bool force_sync = false);
// A device buffer other enqueued work may still read: wait for that work before re-allocating it.
int ensure_quiesced(pe_engine* h, DevBuf& b, size_t bytes);
int ensure_quiesced_arenas(pe_engine* h, DevBuf pe_engine::PipeArena::*m, size_t bytes);  // ... of every arena of the rotation
void grow_idle_arenas(pe_engine* h, size_t stage_bytes, size_t out_bytes);

// ------------------------------------------------------------------ G1 plan
constexpr uint32_t G1_TARGET_LANES = 131072;  // 2 waves per SIMD on 256 CUs
constexpr uint32_t G1_MIN_K = 4;              // fewest members per accumulation lane (fewer: more tree levels than adds)
struct G1Plan {
    uint32_t n_groups = 0, n_slots = 0, n_partials = 0;
};
// sizes[g] = members of group g; writes descriptors into `out` (member_start = 0, bits_word = NONE32: the caller
// fills them in afterwards).
// wg_slots = task slots per workgroup (G1: one lane per slot; G2: a lane pair per slot), target_slots = slots that
// fill the chip at two waves per SIMD.
template <typename SizeFn>
void plan_g1(uint32_t n_groups, SizeFn size_of, G1Group* out, G1Plan* plan, uint32_t wg_slots = G1_WG,
             uint32_t target_slots = G1_TARGET_LANES)
{
    uint64_t total = 0;
    for (uint32_t g = 0; g < n_groups; ++g) total += size_of(g);
    const uint32_t min_k = G1_MIN_K;
    const uint32_t k = (uint32_t)std::max<uint64_t>(min_k, (total + target_slots - 1) / target_slots);
    uint32_t cursor = 0, outp = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        G1Group& d = out[g];
        const uint32_t sz = size_of(g);
        d.member_start = 0;
        d.bits_word = NONE32;
        d.n_members = sz;
        d.k = k;
        d.n_tasks = (sz + k - 1) / k;
        if (d.n_tasks == 0) {
            d.log2_block = 0;
            d.slot_base = cursor;
            d.out_base = outp;
            continue;
        }
        if (d.n_tasks <= wg_slots) {
            uint32_t l2 = 0;
            while ((1u << l2) < d.n_tasks) ++l2;
            d.log2_block = l2;
            const uint32_t blk = 1u << l2;
            cursor = (cursor + blk - 1) & ~(blk - 1);
            d.slot_base = cursor;
            cursor += blk;
            d.out_base = outp;
            outp += 1;
        } else {
            d.log2_block = 9;  // wide: whole workgroups
            cursor = (cursor + wg_slots - 1) & ~(wg_slots - 1);
            d.slot_base = cursor;
            const uint32_t wgs = (d.n_tasks + wg_slots - 1) / wg_slots;
            cursor += wgs * wg_slots;
            d.out_base = outp;
            outp += wgs;
        }
    }
    plan->n_groups = n_groups;
    plan->n_slots = cursor;
    plan->n_partials = outp;
}

void g1_stream_guard(pe_engine* h, hipStream_t s);
int build_points29(pe_engine* h, uint64_t n);  // after the registry's points changed (n rows)
int launch_g1_planned(pe_engine* h, const uint32_t* d_points, const uint32_t* d_members, const uint32_t* d_bits,
                      const G1Group* d_groups, const G1Plan& plan, uint8_t* d_out96, uint32_t* dev_jac,
                      hipStream_t s = nullptr, hipStream_t fin = nullptr, DevBuf* partials = nullptr,
                      DevBuf* lane_partials = nullptr, const AttPlan* plan_dev = nullptr,
                      const uint32_t* d_members1 = nullptr, uint64_t caller_rows = 0);

// ------------------------------------------------------------------ attestation resolution
struct Resolved {
    CommitteeTable* table = nullptr;
    uint32_t pos = 0;        // committee id in the table
    uint32_t size = 0;       // committee length
    uint32_t block_idx = 0;  // beacon_block_root
};

// Per-call memo for the host-side walks of a batch: consecutive rows mostly name the same few roots.
struct BatchMemo {
    // direct-mapped root -> block index cache (a batch votes for a few dozen distinct blocks; the store's node-based
    // map costs two cache misses per lookup)
    struct Slot { uint8_t root[32]; uint32_t idx; uint32_t used; };
    Slot slots[256];
    std::vector<uint64_t> anc;  // per block index: (slot + 1) << 32 | ancestor index of the last get_ancestor asked
    BatchMemo() { for (auto& s : slots) s.used = 0; }
    bool find(const pe_engine* h, int /*which*/, const uint8_t* root, uint32_t* idx)
    {
        Slot& s = slots[root[0]];  // roots are hash outputs: any byte is uniform
        if (s.used && memcmp(s.root, root, 32) == 0) { *idx = s.idx; return true; }
        if (!find_block(h, to_root(root), idx)) return false;
        memcpy(s.root, root, 32);
        s.idx = *idx;
        s.used = 1;
        return true;
    }
    uint32_t ancestor(const pe_engine* h, uint32_t idx, uint64_t slot)
    {
        if (anc.size() != h->blocks.size()) anc.assign(h->blocks.size(), 0);
        const uint64_t tag = (slot + 1) << 32;
        if ((anc[idx] & 0xFFFFFFFF00000000ull) == tag && slot < 0xFFFFFFFEull) return (uint32_t)anc[idx];
        const uint32_t r = get_ancestor(h, idx, slot);
        if (slot < 0xFFFFFFFEull) anc[idx] = tag | r;
        return r;
    }
};

int32_t validate_for_fork_choice(pe_engine* h, const pe_attestation& a, Resolved* out, BatchMemo* memo);
// AttestationData (pe:689-697) is the first 128 bytes of the row, without padding
static_assert(offsetof(pe_attestation, bits_offset) == 128, "pe_attestation: AttestationData must be the leading 128 bytes");
inline bool att_data_equal(const pe_attestation& a, const pe_attestation& b) { return memcmp(&a, &b, 128) == 0; }

// librccl, resolved at run time (engine_dist.cpp)
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl();
// the two exchange steps, through RCCL or the caller's function table (engine_dist.cpp).  g1_chain: the all-gather of a
// G1 chain (its own communicator unless PE_DIST_SINGLE_COMM)
int dist_all_reduce_u64(pe_engine* h, void* dev_buf, size_t count, hipStream_t s);
int dist_all_gather(pe_engine* h, const void* send, void* recv, size_t bytes_per_rank, hipStream_t s);
// waits that give up after dist_timeout_ms on a handle that exchanges with other ranks (engine_dist.cpp)
hipError_t bounded_event_sync(pe_engine* h, hipEvent_t ev);
hipError_t bounded_stream_sync(pe_engine* h, hipStream_t s);

// rows resident in device memory (engine_resident.cpp)
bool rows_on_device(const void* p);
BlockTableDev block_table_dev(const pe_engine* h);
// dev_partials: the groups' sums stay XYZZ partials of this shard's members in that device buffer (n slots) instead of
// being normalised into out_aggpk96 (pe_aggregate_sharded)
int aggregate_resident(pe_engine* h, const pe_attestation* d_rows, uint32_t n, const uint8_t* bits_arena,
                       uint64_t arena_len, pe_attestation* out_atts, uint32_t* out_n_groups, uint32_t* group_of,
                       uint8_t* out_bits_arena, uint64_t out_arena_cap, uint8_t* out_aggpk96, uint32_t* out_count,
                       uint32_t* dev_partials = nullptr, int set = 0, const uint32_t* n_dev = nullptr);
// the pieces pe_aggregate_exchange packs: the resident aggregate's groups, unions and the caller's rows
struct ResidentParts { const void* rows; const AttGroup* grp; const AttPlan* plan; const uint32_t* res_bits; const uint32_t* res_info; uint32_t n_in; };
int resident_parts(pe_engine* h, ResidentParts* out);
// the device-side plan (group count, error word) of the last aggregate over rows in device memory
int resident_plan_dev(pe_engine* h, const AttPlan** out);
// ... and its grouping lists: group g's member rows are member_row[ug[g].list_start .. + ug[g].n_atts)
int resident_lists(pe_engine* h, const UnionGroup** ug, const uint32_t** member_row);
int on_attestation_resident(pe_engine* h, uint32_t cap, int32_t* status, uint32_t* out_count);
int process_attestation_resident(pe_engine* h, const pe_state_ctx* st, uint32_t cap, int32_t* status, uint64_t* out_numerators);

// pe_aggregate and its partial / sharded forms (engine_attest.cpp)
int aggregate_impl(pe_engine* h, const pe_attestation* atts, uint32_t n, const uint8_t* bits_arena,
                   uint64_t arena_len, const uint8_t* sig_points96, pe_attestation* out_atts,
                   uint32_t* out_n_groups, uint32_t* group_of, uint8_t* out_bits_arena, uint64_t out_arena_cap,
                   uint8_t* out_sig96, uint8_t* out_aggpk96, uint32_t* out_count, void* dev_partials,
                   uint32_t dev_partials_capacity = 0, bool partials_may_defer = false);

}  // namespace posevo
