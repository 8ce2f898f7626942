/* step_client.c -- a plain C99 client of libposevo.so: one epoch of attestations per step through the per-function C ABI
 * of include/posevo.h, as a consensus client written in C / Go (cgo) / Rust (bindgen) would drive it:
 *
 *   pe_aggregate                       validator guide, pos-evolution.md:1536 (Attestation pe:714-717)
 *   pe_on_attestation_batch            on_attestation x n, pe:963-979 / pe:1423-1441
 *   pe_get_head                        get_head, pe:1102-1116
 *   pe_process_attestation_batch       process_attestation x n, pe:722-754
 *
 * inside streaming pipelines (pe_pipeline_begin_streaming / pe_pipeline_end_lagged) with the aggregate's rows handed to
 * the handlers where they lie in HBM (PE_BITS_RESIDENT).  No Python, no torch: what this prints is the rate of the
 * engine behind its C boundary.
 *
 *   cc -O2 -std=c99 -Iinclude examples/step_client.c -Lpos_evolution_amd -lposevo -Wl,-rpath,$PWD/pos_evolution_amd -o step_client
 *   ./step_client workload.bin [sync|pipelined|streaming] [warmup steps, default 6] [nohash]
 * (nohash: a timing run -- the byte-wise fold of ~650 KB of outputs per step costs more than the step itself)
 *
 * workload.bin is written by examples/make_workload.py (synthetic registry, block tree, committee tables and one
 * epoch of attestations per step).  Output: ms per step, the head after every step folded into one hash together with
 * every output the calls produce -- tests/test_gpu_c_client.py compares that hash with the same steps driven from
 * Python through synchronous calls.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "posevo.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != PE_OK) {                                                                      \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, pe_strerror(rc_), pe_last_error(h)); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

typedef struct {
    uint64_t magic, n_val, n_comm, n_blocks, n_steps, slots_per_epoch, genesis_time, reserved;
} wl_header;

typedef struct {
    uint64_t epoch, n_atts, arena_len, tick_time;
    pe_state_ctx ctx;
    uint32_t* offsets;      /* n_comm + 1 */
    uint32_t* members;      /* n_val */
    pe_attestation* atts;   /* n_atts */
    uint8_t* arena;         /* arena_len */
    uint8_t* sigs;          /* n_atts x 96: the attestations' compressed BLSSignatures (files with header flag 1) */
} wl_step;

typedef struct {            /* the outputs of one step: a ring of these, consumed two steps behind when streaming */
    pe_attestation* out_atts;
    uint32_t n_groups;
    uint32_t *group_of, *count, *att_count;
    uint8_t *out_bits, *aggpk, *out_sigs;   /* out_sigs: 96-byte compressed aggregate signature per group */
    int32_t *status, *pstatus, *sig_status;
    uint64_t* numerators;
    uint8_t head[32];
    uint64_t arena_len;
} step_out;

static uint64_t fnv(uint64_t hsh, const void* p, size_t n)
{
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) { hsh ^= b[i]; hsh *= 0x100000001B3ull; }
    return hsh;
}

static void* xread(FILE* f, size_t bytes)
{
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* fold one completed step into the hash: everything the four calls handed back */
static int with_sigs;
static uint64_t fold(uint64_t hsh, const step_out* o)
{
    const uint32_t g = o->n_groups;
    hsh = fnv(hsh, &g, 4);
    for (uint32_t k = 0; k < g; ++k) {   /* rows without the (input dependent) reserved word */
        hsh = fnv(hsh, &o->out_atts[k], 128);
        hsh = fnv(hsh, &o->out_atts[k].bits_offset, 12);
    }
    if (g) hsh = fnv(hsh, o->out_bits, (size_t)o->out_atts[g - 1].bits_offset + (o->out_atts[g - 1].n_bits + 7) / 8);
    hsh = fnv(hsh, o->count, 4ull * g);
    hsh = fnv(hsh, o->aggpk, 96ull * g);
    hsh = fnv(hsh, o->status, 4ull * g);
    hsh = fnv(hsh, o->att_count, 4ull * g);
    hsh = fnv(hsh, o->pstatus, 4ull * g);
    hsh = fnv(hsh, o->numerators, 8ull * g);
    hsh = fnv(hsh, o->head, 32);
    if (with_sigs) hsh = fnv(hsh, o->out_sigs, 96ull * g);   /* Attestation.signature of every aggregate (pe:717) */
    return hsh;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s workload.bin [sync|pipelined|streaming] [warmup]\n", argv[0]); return 2; }
    const char* mode = argc > 2 ? argv[2] : "streaming";
    const uint64_t warmup_arg = argc > 3 ? (uint64_t)strtoull(argv[3], NULL, 10) : 6;
    const int hashing = !(argc > 4 && strcmp(argv[4], "nohash") == 0);
    const int streaming = strcmp(mode, "streaming") == 0, pipelined = streaming || strcmp(mode, "pipelined") == 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    wl_header hd;
    if (fread(&hd, sizeof hd, 1, f) != 1 || hd.magic != 0x30764F5645534F50ull) { fprintf(stderr, "bad workload file\n"); return 2; }
    with_sigs = (hd.reserved & 1) != 0;   /* the file carries one compressed BLSSignature per attestation */

    pe_engine* h = NULL;
    pe_config cfg;
    pe_config_default(&cfg);
    cfg.slots_per_epoch = hd.slots_per_epoch;
    cfg.max_committee_tables = (uint32_t)hd.n_steps + 1;
    {
        int rc = pe_engine_create(&cfg, &h);
        if (rc != PE_OK) { fprintf(stderr, "pe_engine_create -> %d (%s)\n", rc, pe_strerror(rc)); return 1; }
    }
    /* ---- the store: block tree, registry ---- */
    uint8_t* roots = (uint8_t*)xread(f, 32 * hd.n_blocks);
    uint32_t* parent = (uint32_t*)xread(f, 4 * hd.n_blocks);
    uint64_t* bslot = (uint64_t*)xread(f, 8 * hd.n_blocks);
    uint64_t* balance = (uint64_t*)xread(f, 8 * hd.n_val);
    uint8_t* vflags = (uint8_t*)xread(f, hd.n_val);
    uint8_t* pubkeys = (uint8_t*)xread(f, 96 * hd.n_val);
    CHECK(pe_store_init(h, hd.genesis_time, bslot[0], roots));
    for (uint64_t i = 1; i < hd.n_blocks; ++i)
        CHECK(pe_add_block(h, roots + 32 * i, roots + 32 * parent[i], bslot[i], 0, roots, 0, roots));
    CHECK(pe_set_validators(h, hd.n_val, pubkeys, balance, vflags));
    free(pubkeys);
    /* ---- the steps: committee table + attestations of one epoch each ---- */
    wl_step* steps = (wl_step*)calloc(hd.n_steps, sizeof(wl_step));
    uint64_t max_atts = 1, max_arena = 1;
    for (uint64_t s = 0; s < hd.n_steps; ++s) {
        wl_step* st = &steps[s];
        if (fread(st, 4 * 8 + sizeof(pe_state_ctx), 1, f) != 1) { fprintf(stderr, "short read\n"); return 2; }
        st->offsets = (uint32_t*)xread(f, 4 * (hd.n_comm + 1));
        st->members = (uint32_t*)xread(f, 4 * hd.n_val);
        st->atts = (pe_attestation*)xread(f, sizeof(pe_attestation) * st->n_atts);
        st->arena = (uint8_t*)xread(f, st->arena_len);
        st->sigs = with_sigs ? (uint8_t*)xread(f, 96 * st->n_atts) : NULL;
        CHECK(pe_set_committees(h, st->epoch, (uint32_t)hd.n_comm, st->offsets, st->members));
        if (st->n_atts > max_atts) max_atts = st->n_atts;
        if (st->arena_len > max_arena) max_arena = st->arena_len;
    }
    fclose(f);
    enum { RING = 4 };
    step_out ring[RING];
    for (int r = 0; r < RING; ++r) {
        step_out* o = &ring[r];
        o->out_atts = (pe_attestation*)calloc(max_atts, sizeof(pe_attestation));
        o->group_of = (uint32_t*)calloc(max_atts, 4);
        o->count = (uint32_t*)calloc(max_atts, 4);
        o->att_count = (uint32_t*)calloc(max_atts, 4);
        o->out_bits = (uint8_t*)calloc(max_arena, 1);
        o->aggpk = (uint8_t*)calloc(max_atts, 96);
        o->status = (int32_t*)calloc(max_atts, 4);
        o->pstatus = (int32_t*)calloc(max_atts, 4);
        o->numerators = (uint64_t*)calloc(max_atts, 8);
        o->out_sigs = (uint8_t*)calloc(max_atts, 96);
        o->sig_status = (int32_t*)calloc(max_atts, 4);
        o->arena_len = max_arena;
    }
    uint64_t hsh = 0xCBF29CE484222325ull, n_att = 0, n_att_warm = 0, consumed = 0;
    const uint64_t first_gen = pe_pipeline_generation(h);   /* pipelines begun before the steps (none, normally) */
    /* the first steps grow the engine's pinned staging blocks (milliseconds each): untimed, like bench.py's warm-up */
    const uint64_t warmup = warmup_arg + 2 < hd.n_steps ? warmup_arg : 0;
    double t0 = now_ms();
    for (uint64_t s = 0; s < hd.n_steps; ++s) {
        wl_step* st = &steps[s];
        step_out* o = &ring[s % RING];
        if (s == warmup && warmup) { t0 = now_ms(); n_att_warm = n_att; }
        CHECK(pe_on_tick(h, st->tick_time));
        CHECK(pe_participation_rotate(h));
        if (streaming) CHECK(pe_pipeline_begin_streaming(h));
        else if (pipelined) CHECK(pe_pipeline_begin(h));
        if (with_sigs)  /* pe_aggregate + bls.Aggregate over the members' signatures: the aggregate a validator publishes */
            CHECK(pe_aggregate_signed(h, st->atts, (uint32_t)st->n_atts, st->arena, st->arena_len, st->sigs,
                                      PE_SIG_G2_COMPRESSED, o->out_atts, &o->n_groups, o->group_of, o->out_bits,
                                      o->arena_len, o->out_sigs, o->sig_status, o->aggpk, o->count));
        else
            CHECK(pe_aggregate(h, st->atts, (uint32_t)st->n_atts, st->arena, st->arena_len, NULL, o->out_atts, &o->n_groups,
                               o->group_of, o->out_bits, o->arena_len, NULL, o->aggpk, o->count));
        /* the rows are complete at return; bits / counts / pubkeys when the pipeline is (two ends later when streaming) */
        const uint8_t* bits = pipelined ? PE_BITS_RESIDENT : o->out_bits;
        CHECK(pe_on_attestation_batch(h, o->out_atts, o->n_groups, bits, o->arena_len, o->status, NULL, o->att_count));
        CHECK(pe_get_head(h, o->head));
        CHECK(pe_process_attestation_batch(h, &st->ctx, o->out_atts, o->n_groups, bits, o->arena_len, o->pstatus,
                                           o->numerators));
        if (streaming) CHECK(pe_pipeline_end_lagged(h));
        else if (pipelined) CHECK(pe_pipeline_end(h));
        /* consume completed steps: this one, or -- streaming -- whatever pe_pipeline_completed says is complete (step s is
           pipeline s + 1; with lag depth 2 that is the step two back, but the client need not know the depth) */
        if (!streaming) {
            if (hashing) hsh = fold(hsh, o);
            for (uint32_t k = 0; k < o->n_groups; ++k) n_att += o->count[k];
        } else {
            while (consumed < pe_pipeline_completed(h) - first_gen) {
                const step_out* d = &ring[consumed % RING];
                if (hashing) hsh = fold(hsh, d);
                for (uint32_t k = 0; k < d->n_groups; ++k) n_att += d->count[k];
                ++consumed;
            }
            if (s + 1 - consumed >= RING) { fprintf(stderr, "output ring too shallow for the lag depth\n"); return 3; }
        }
    }
    if (streaming) {  /* drain: the steps still in flight complete here */
        CHECK(pe_pipeline_begin(h));
        CHECK(pe_pipeline_end(h));
        for (; consumed < hd.n_steps; ++consumed) {
            const step_out* d = &ring[consumed % RING];
            if (hashing) hsh = fold(hsh, d);
            for (uint32_t k = 0; k < d->n_groups; ++k) n_att += d->count[k];
        }
    }
    const double dt = now_ms() - t0;
    const uint64_t timed = hd.n_steps - warmup;
    /* streaming: the steps consumed inside the timed window lag two behind; the rate below charges the whole window
       with the attestations of the timed steps (the two-step skew cancels: two warm-up steps complete inside it) */
    char sig_hex[200] = "";
    if (with_sigs && hd.n_steps) {  /* the compressed aggregate signature of the last step's first aggregate */
        const step_out* d = &ring[(hd.n_steps - 1) % RING];
        for (int i = 0; i < 96 && d->n_groups; ++i) sprintf(sig_hex + 2 * i, "%02x", d->out_sigs[i]);
    }
    printf("{\"mode\": \"%s\", \"steps\": %llu, \"timed_steps\": %llu, \"ms_per_step\": %.4f, \"attestations\": %llu, "
           "\"attestations_per_s\": %.1f, \"hash\": \"%016llx\", \"aggregate_signature\": \"%s\"}\n",
           mode, (unsigned long long)hd.n_steps, (unsigned long long)timed, dt / (double)timed, (unsigned long long)n_att,
           (double)(n_att - n_att_warm) / (dt * 1e-3), (unsigned long long)hsh, sig_hex);
    pe_engine_destroy(h);
    return 0;
}
