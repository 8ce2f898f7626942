/* posevo_profile.h -- measurement hooks of libposevo.so: per-kernel launch counts / durations from HIP events the engine
 * records around its own launches, and an in-situ timeline of a streaming step.  NOT part of the drop-in boundary
 * (include/posevo.h): nothing a client of the reference's functions needs; bench.py's `roofline` leg and
 * tools/engine_timeline.py use them. */
#ifndef POSEVO_PROFILE_H
#define POSEVO_PROFILE_H
#include "posevo.h"
#ifdef __cplusplus
extern "C" {
#endif

/* When enabled, the engine brackets each launch of its kernels with HIP events on
 * the launch stream and accumulates per-kernel launch counts and durations. */
#define PE_KERNEL_G1_ACCUMULATE 0
#define PE_KERNEL_G1_NORMALISE  1
#define PE_KERNEL_VOTES         2
#define PE_KERNEL_TREE          3
#define PE_KERNEL_LMD           4
#define PE_KERNEL_PARTICIPATION 5
#define PE_KERNEL_BITS_UNION    6
#define PE_KERNEL_G2_ACCUMULATE 7
#define PE_KERNEL_G2_NORMALISE  8
#define PE_KERNEL_G1_TREE       9   /* the LDS tree over the lane partials: its own kernel since round 2 */
#define PE_KERNEL_ATT_GROUP     10  /* rows in device memory: ingest + plan + members (bracketed in timeline mode only) */
#define PE_KERNEL_ATT_VALIDATE  11  /* rows in device memory: the validate_on_attestation / process_attestation kernels (ditto) */
/* paired launches of a streaming caller (round 5): the fork-choice kernel of step N and the row kernel of step N + 1 as block
 * ranges of one grid (pair_kernels.hip); the stand-alone ids above count only the launches that ran alone */
#define PE_KERNEL_PAIR_INGEST_VALIDATE 12
#define PE_KERNEL_PAIR_PLAN_LMD        13
#define PE_KERNEL_PAIR_MEMBERS_VOTES   14
#define PE_KERNEL_PAIR_UNION_TREE      15
#define PE_KERNEL_G2_DECOMPRESS 16  /* the signature legs' decompression: one launch per POSEVO_SIG_BATCH streaming steps (round 6) */
#define PE_KERNEL_COUNT         17
/* on = 0 off, 1 per-kernel totals, 3 totals of PE_KERNEL_G1_ACCUMULATE only (what a timed region can afford: every bracket is
 * two event packets on its stream), 2 totals + a timeline: every bracketed launch's start (relative to the last
 * pe_profile_reset, which marks time zero on the engine's stream) and duration, read with pe_profile_timeline.  The
 * events are the engine's own, on the streams the kernels run on: an in-situ picture of a streaming step without a
 * profiler's serialisation (rocprofv3 stretches the 0.28 ms step to 0.4). */
int pe_profile_enable(pe_engine* h, int on);
int pe_profile_reset(pe_engine* h);
int pe_profile_get(pe_engine* h, int kernel, uint64_t* launches, double* total_ms);
/* The launches bracketed since the last pe_profile_reset, in the order they were drained (per kernel kind, launch order
 * inside a kind): kernel id, start and duration in ms.  At most cap entries are written; *out_n = how many exist. */
int pe_profile_timeline(pe_engine* h, int32_t* kernel, double* start_ms, double* duration_ms, uint32_t cap, uint32_t* out_n);

/* Which of the handle's four hot streams (engine, accumulation, tree, finish) share a hardware queue, asked of the device
 * again (a spin kernel on one stream, clock stamps on the others; everything enqueued completes first):
 * out_class[i] = the lowest i' whose stream shares stream i's queue -- {0, 1, 2, 3} when every stream has a queue of its own,
 * which is what pe_engine_create arranges whatever the process created before the handle (POSEVO_QUEUE_PROBE=0 skips it). */
int pe_profile_queue_classes(pe_engine* h, int32_t out_class[4]);
/* How often the handle has (re)allocated a buffer of its arenas (staging / output blocks, resident words, G1 and signature scratch)
 * since it was created.  Every such growth inside a stream of steps is milliseconds of allocation or a drained pipeline; after the
 * first step of a stream of like steps the count must stand still, whatever the lag depth. */
int pe_profile_arena_growths(const pe_engine* h, uint64_t* out);
/* The shader clock each k_g1_accumulate launch ran at, in MHz, for the launches since pe_profile_reset made while profiling was
 * on (the last 4096 of them), in launch order: workgroup 0 of every launch counts shader cycles against the fixed 100 MHz counter.
 * out_n = how many there are; at most cap are written.  The clock is the power management's and a multiplier-bound kernel follows
 * it one to one: ~2.05 GHz for the first milliseconds of heavy load after an idle moment, ~2.4 GHz after ~35 ms of it
 * (tools/clockramp.hip, profiles/r06_clockramp.txt) -- which is most of the distance between a 20-step and a 200-step run. */
int pe_profile_accumulate_mhz(pe_engine* h, double* out_mhz, uint32_t cap, uint32_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* POSEVO_PROFILE_H */
