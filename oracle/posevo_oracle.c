/*
 * posevo_oracle.c -- L1 CPU oracle ("port") for the pos-evolution hot path.
 *
 * ORACLE / TEST INFRASTRUCTURE ONLY.  Linked/loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker and
 * as the timed CPU baseline -- never by the product path (pos_evolution_amd/).
 *
 * What it restates (pe:N = /root/reference/pos-evolution.md line N):
 *   po_get_head                 get_head pe:1102-1116 over get_filtered_block_tree and
 *                               get_latest_attesting_balance [UPSTREAM-MEMORY, SURVEY.md A.1-A.3]
 *   po_update_latest_messages   update_latest_messages pe:1435-1441, applied in batch order
 *   po_process_attestation_flags  the flag loop of process_attestation pe:744-749
 *   po_bits_union               aggregation_bits = OR (validator guide; pe:715, pe:730)
 *   po_g1_*                     BLS12-381 G1 point sums (bls.Aggregate / the pubkey sum of
 *                               FastAggregateVerify; NOT in the reference: pe:165 is its only bls call)
 *
 * PARITY: L1 is accepted only after it is bit-identical to the L0 literal oracle
 * (oracle/spec.py, oracle/g1.py) on randomised small inputs
 * (tests/test_oracle_cport.py).  For the pieces L0 itself restates from memory
 * (Appendix A) and for all of G1, parity against the reference is UNPINNED: the
 * reference holds no vectors for them.
 *
 * The algorithms are deliberately different from the GPU engine's (reverse
 * topological subtree sum + explicit descent here; pre-order prefix scan +
 * pointer jumping there; 6x64-bit Montgomery limbs here; 12x32-bit there) so
 * that agreement is evidence.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ======================================================================= */
/* Fork choice                                                             */
/* ======================================================================= */

#define PO_VAL_ACTIVE 0x01u
#define PO_VAL_SLASHED 0x02u
#define PO_VAL_EQUIVOCATING 0x04u
#define PO_NONE 0xFFFFFFFFu

/*
 * Blocks are indexed in insertion order, parent before child (parent[i] < i,
 * parent[0] = PO_NONE or 0 for the anchor).  leaf_ok[i] = the checkpoint test
 * filter_block_tree applies when block i is a leaf (A.3).  roots = 32*n bytes.
 * vote_block[v] = block index of validator v's latest message or PO_NONE.
 *
 * weight(root) = sum of effective_balance over active, non-equivocating validators
 * whose latest-message block has `root` as ancestor-or-self (pe:322, A.1)
 * + proposer boost on every ancestor-or-self of boost_idx.
 * get_ancestor(vote, slot(root)) == root  <=>  root is ancestor-or-self of vote,
 * because slots strictly increase along parent links.
 */
int po_get_head(uint32_t n_blocks, const uint32_t* parent, const uint8_t* leaf_ok, const uint8_t* roots,
                uint64_t n_val, const uint32_t* vote_block, const uint64_t* eff_balance,
                const uint8_t* flags, int filter_slashed, uint32_t justified_idx, uint32_t boost_idx,
                uint64_t slots_per_epoch, uint64_t boost_percent, uint64_t balance_increment,
                uint64_t* out_weights, uint32_t* out_head)
{
    if (n_blocks == 0 || justified_idx >= n_blocks) return -1;
    uint64_t* w = out_weights;
    memset(w, 0, sizeof(uint64_t) * n_blocks);
    uint64_t total_active = 0, num_active = 0;
    for (uint64_t v = 0; v < n_val; ++v) {
        uint8_t f = flags[v];
        if (!(f & PO_VAL_ACTIVE)) continue;
        total_active += eff_balance[v];
        num_active += 1;
        if (f & PO_VAL_EQUIVOCATING) continue;
        if (filter_slashed && (f & PO_VAL_SLASHED)) continue;
        uint32_t b = vote_block[v];
        if (b == PO_NONE) continue;
        if (b >= n_blocks) return -2;
        w[b] += eff_balance[v];
    }
    if (boost_idx != PO_NONE && num_active > 0) {
        if (boost_idx >= n_blocks) return -3;
        if (total_active < balance_increment) total_active = balance_increment; /* get_total_balance max() */
        uint64_t avg_balance = total_active / num_active;
        uint64_t committee_size = num_active / slots_per_epoch;
        uint64_t committee_weight = committee_size * avg_balance;
        uint64_t proposer_score = (uint64_t)(((u128)committee_weight * boost_percent) / 100);
        w[boost_idx] += proposer_score;
    }
    /* subtree sums, reverse topological (= reverse insertion) order */
    for (uint32_t i = n_blocks - 1; i > 0; --i) {
        uint32_t p = parent[i];
        if (p != PO_NONE && p < i) w[p] += w[i];
    }
    /* filtered block tree: viable[i] = leaf ? leaf_ok[i] : any(viable[children]) */
    uint8_t* has_child = (uint8_t*)calloc(n_blocks, 1);
    uint8_t* viable = (uint8_t*)calloc(n_blocks, 1);
    if (!has_child || !viable) { free(has_child); free(viable); return -4; }
    for (uint32_t i = 1; i < n_blocks; ++i)
        if (parent[i] != PO_NONE && parent[i] < i) has_child[parent[i]] = 1;
    for (uint32_t i = n_blocks; i-- > 0;) {
        if (!has_child[i]) viable[i] = leaf_ok[i] ? 1 : 0;
        if (viable[i] && i > 0 && parent[i] != PO_NONE && parent[i] < i) viable[parent[i]] = 1;
    }
    /* descent (pe:1106-1116): children scan is O(B) per level as in the spec */
    uint32_t head = justified_idx;
    for (;;) {
        uint32_t best = PO_NONE;
        for (uint32_t c = head + 1; c < n_blocks; ++c) {
            if (parent[c] != head || !viable[c]) continue;
            if (best == PO_NONE) { best = c; continue; }
            if (w[c] > w[best] || (w[c] == w[best] && memcmp(roots + 32 * (size_t)c, roots + 32 * (size_t)best, 32) > 0))
                best = c;
        }
        if (best == PO_NONE) break;
        head = best;
    }
    free(has_child);
    free(viable);
    *out_head = head;
    return 0;
}

/*
 * update_latest_messages (pe:1435-1441) for a batch, literally in order.
 * Attestation a covers committee members[member_off[a] .. +n_bits[a]) with
 * bits at arena + bits_off[a] (LSB-first).  vote_epoch/vote_block are the
 * LatestMessage table (vote_block == PO_NONE <=> no message).
 */
void po_update_latest_messages(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits,
                               const uint32_t* bits_off, const uint64_t* target_epoch,
                               const uint32_t* block_idx, const uint8_t* arena, const uint32_t* members,
                               const uint8_t* val_flags, uint64_t* vote_epoch, uint32_t* vote_block)
{
    for (uint32_t a = 0; a < n_att; ++a) {
        const uint8_t* bits = arena + bits_off[a];
        for (uint32_t i = 0; i < n_bits[a]; ++i) {
            if (!((bits[i >> 3] >> (i & 7)) & 1)) continue;
            uint32_t v = members[member_off[a] + i];
            if (val_flags[v] & PO_VAL_EQUIVOCATING) continue;
            if (vote_block[v] == PO_NONE || target_epoch[a] > vote_epoch[v]) {
                vote_epoch[v] = target_epoch[a];
                vote_block[v] = block_idx[a];
            }
        }
    }
}

/*
 * The flag loop of process_attestation (pe:744-749), per attestation in order:
 * for each attesting index, for each flag in the attestation's flag mask that
 * is not yet set: set it, numerator += base_reward(index) * weight.
 * which[a] selects participation array 0 (current) / 1 (previous).
 * base_reward(i) = (eff_balance[i] / increment) * base_reward_per_increment (A.9).
 */
void po_process_attestation_flags(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits,
                                  const uint32_t* bits_off, const uint8_t* flag_mask, const uint8_t* which,
                                  const uint8_t* arena, const uint32_t* members, const uint64_t* eff_balance,
                                  uint64_t increment, uint64_t base_reward_per_increment,
                                  uint8_t* participation_current, uint8_t* participation_previous,
                                  uint64_t* out_numerators)
{
    static const uint64_t W[3] = {14, 26, 14}; /* PARTICIPATION_FLAG_WEIGHTS (A.9) */
    for (uint32_t a = 0; a < n_att; ++a) {
        const uint8_t* bits = arena + bits_off[a];
        uint8_t* part = which[a] ? participation_previous : participation_current;
        uint64_t num = 0;
        for (uint32_t i = 0; i < n_bits[a]; ++i) {
            if (!((bits[i >> 3] >> (i & 7)) & 1)) continue;
            uint32_t v = members[member_off[a] + i];
            uint64_t base_reward = (eff_balance[v] / increment) * base_reward_per_increment;
            for (int f = 0; f < 3; ++f) {
                uint8_t bit = (uint8_t)(1u << f);
                if ((flag_mask[a] & bit) && !(part[v] & bit)) {
                    part[v] |= bit;
                    num += base_reward * W[f];
                }
            }
        }
        out_numerators[a] = num;
    }
}

/* aggregation_bits of group g = OR over its member attestations (A.8).
 * group_start[g]..group_start[g+1] index att_list; every member has n_bits[g] bits. */
void po_bits_union(uint32_t n_groups, const uint32_t* group_start, const uint32_t* att_list,
                   const uint32_t* att_bits_off, const uint8_t* arena, const uint32_t* group_n_bits,
                   const uint32_t* out_bits_off, uint8_t* out_arena, uint32_t* out_count)
{
    for (uint32_t g = 0; g < n_groups; ++g) {
        uint32_t nb = (group_n_bits[g] + 7) / 8;
        uint8_t* o = out_arena + out_bits_off[g];
        memset(o, 0, nb);
        for (uint32_t k = group_start[g]; k < group_start[g + 1]; ++k) {
            const uint8_t* b = arena + att_bits_off[att_list[k]];
            for (uint32_t j = 0; j < nb; ++j) o[j] |= b[j];
        }
        if (group_n_bits[g] & 7) o[nb - 1] &= (uint8_t)((1u << (group_n_bits[g] & 7)) - 1);
        uint32_t c = 0;
        for (uint32_t j = 0; j < nb; ++j) c += (uint32_t)__builtin_popcount(o[j]);
        if (out_count) out_count[g] = c;
    }
}

/* ======================================================================= */
/* BLS12-381 base field: 6 x 64-bit limbs, Montgomery form, R = 2^384      */
/* ======================================================================= */

typedef struct { uint64_t l[6]; } fp;

static const fp FP_P = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                         0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const uint64_t FP_N0 = 0x89f3fffcfffcfffdULL; /* -p^-1 mod 2^64 */
static const fp FP_R1 = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                          0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}}; /* R mod p = mont(1) */
static const fp FP_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                          0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}}; /* R^2 mod p */

static int fp_is_zero(const fp* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }
static int fp_eq(const fp* a, const fp* b) { return memcmp(a, b, sizeof(fp)) == 0; }
static int fp_geq_p(const fp* a)
{
    for (int i = 5; i >= 0; --i) {
        if (a->l[i] > FP_P.l[i]) return 1;
        if (a->l[i] < FP_P.l[i]) return 0;
    }
    return 1;
}
static void fp_sub_p(fp* a)
{
    u128 borrow = 0;
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)a->l[i] - FP_P.l[i] - borrow;
        a->l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
static void fp_add(fp* r, const fp* a, const fp* b)
{
    u128 c = 0;
    for (int i = 0; i < 6; ++i) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    if (fp_geq_p(r)) fp_sub_p(r); /* a, b < p < 2^381 so no carry out of 384 bits */
}
static void fp_sub(fp* r, const fp* a, const fp* b)
{
    u128 borrow = 0;
    fp t;
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 6; ++i) { c += (u128)t.l[i] + FP_P.l[i]; t.l[i] = (uint64_t)c; c >>= 64; }
    }
    *r = t;
}
static void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }

/* Montgomery product a*b*R^-1 mod p (CIOS) */
static void fp_mul(fp* r, const fp* a, const fp* b)
{
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) {
        u128 c = 0;
        for (int j = 0; j < 6; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FP_N0;
        c = (u128)m * FP_P.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 6; ++j) {
            c += (u128)m * FP_P.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    fp out;
    memcpy(out.l, t, sizeof(out.l));
    if (t[6] || fp_geq_p(&out)) fp_sub_p(&out);
    *r = out;
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }

static void fp_from_be48(fp* r, const uint8_t* be)
{
    fp raw;
    for (int i = 0; i < 6; ++i) {
        uint64_t w = 0;
        for (int k = 0; k < 8; ++k) w = (w << 8) | be[(5 - i) * 8 + k];
        raw.l[i] = w;
    }
    fp_mul(r, &raw, &FP_R2); /* to Montgomery form */
}
static void fp_to_be48(uint8_t* be, const fp* a)
{
    fp one = {{1, 0, 0, 0, 0, 0}}, raw;
    fp_mul(&raw, a, &one); /* out of Montgomery form, fully reduced */
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 8; ++k) be[(5 - i) * 8 + k] = (uint8_t)(raw.l[i] >> (56 - 8 * k));
}

/* a^(p-2) by square-and-multiply (Fermat) */
static void fp_inv(fp* r, const fp* a)
{
    /* exponent p - 2, little-endian limbs */
    uint64_t e[6];
    memcpy(e, FP_P.l, sizeof(e));
    e[0] -= 2; /* p's low limb ends in ...aaab, no borrow */
    fp acc = FP_R1, base = *a;
    for (int i = 0; i < 6; ++i)
        for (int b = 0; b < 64; ++b) {
            if ((e[i] >> b) & 1) fp_mul(&acc, &acc, &base);
            fp_sqr(&base, &base);
        }
    *r = acc;
}

/* ======================================================================= */
/* G1: y^2 = x^3 + 4, Jacobian accumulators, affine inputs                 */
/* ======================================================================= */

typedef struct { fp x, y, z; } g1j; /* z == 0 <=> infinity */
typedef struct { fp x, y; int inf; } g1a;

static void g1j_set_inf(g1j* p) { memset(p, 0, sizeof(*p)); }
static int g1j_is_inf(const g1j* p) { return fp_is_zero(&p->z); }

static void g1j_double(g1j* r, const g1j* p)
{
    if (g1j_is_inf(p) || fp_is_zero(&p->y)) { g1j_set_inf(r); return; }
    /* a = 0: A = X^2, B = Y^2, C = B^2, D = 2((X+B)^2 - A - C), E = 3A, F = E^2 */
    fp A, B, C, D, E, F, t;
    fp_sqr(&A, &p->x);
    fp_sqr(&B, &p->y);
    fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B);
    fp_sqr(&t, &t);
    fp_sub(&t, &t, &A);
    fp_sub(&t, &t, &C);
    fp_dbl(&D, &t);
    fp_dbl(&E, &A);
    fp_add(&E, &E, &A);
    fp_sqr(&F, &E);
    fp X3, Y3, Z3;
    fp_dbl(&t, &D);
    fp_sub(&X3, &F, &t);
    fp_mul(&Z3, &p->y, &p->z);
    fp_dbl(&Z3, &Z3);
    fp_sub(&t, &D, &X3);
    fp_mul(&Y3, &E, &t);
    fp_dbl(&C, &C);
    fp_dbl(&C, &C);
    fp_dbl(&C, &C);
    fp_sub(&Y3, &Y3, &C);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void g1j_add_affine(g1j* r, const g1j* p, const g1a* q)
{
    if (q->inf) { *r = *p; return; }
    if (g1j_is_inf(p)) { r->x = q->x; r->y = q->y; r->z = FP_R1; return; }
    fp Z2, U2, S2, H, Rr, HH, HHH, V, t;
    fp_sqr(&Z2, &p->z);
    fp_mul(&U2, &q->x, &Z2);
    fp_mul(&S2, &q->y, &Z2);
    fp_mul(&S2, &S2, &p->z);
    fp_sub(&H, &U2, &p->x);
    fp_sub(&Rr, &S2, &p->y);
    if (fp_is_zero(&H)) {
        if (fp_is_zero(&Rr)) { g1j_double(r, p); return; }
        g1j_set_inf(r);
        return;
    }
    fp_sqr(&HH, &H);
    fp_mul(&HHH, &HH, &H);
    fp_mul(&V, &p->x, &HH);
    fp X3, Y3, Z3;
    fp_sqr(&X3, &Rr);
    fp_sub(&X3, &X3, &HHH);
    fp_dbl(&t, &V);
    fp_sub(&X3, &X3, &t);
    fp_sub(&t, &V, &X3);
    fp_mul(&Y3, &Rr, &t);
    fp_mul(&t, &p->y, &HHH);
    fp_sub(&Y3, &Y3, &t);
    fp_mul(&Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void g1j_add(g1j* r, const g1j* p, const g1j* q)
{
    if (g1j_is_inf(q)) { *r = *p; return; }
    if (g1j_is_inf(p)) { *r = *q; return; }
    fp Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t;
    fp_sqr(&Z1Z1, &p->z);
    fp_sqr(&Z2Z2, &q->z);
    fp_mul(&U1, &p->x, &Z2Z2);
    fp_mul(&U2, &q->x, &Z1Z1);
    fp_mul(&S1, &p->y, &Z2Z2);
    fp_mul(&S1, &S1, &q->z);
    fp_mul(&S2, &q->y, &Z1Z1);
    fp_mul(&S2, &S2, &p->z);
    fp_sub(&H, &U2, &U1);
    fp_sub(&Rr, &S2, &S1);
    if (fp_is_zero(&H)) {
        if (fp_is_zero(&Rr)) { g1j_double(r, p); return; }
        g1j_set_inf(r);
        return;
    }
    fp_sqr(&HH, &H);
    fp_mul(&HHH, &HH, &H);
    fp_mul(&V, &U1, &HH);
    fp X3, Y3, Z3;
    fp_sqr(&X3, &Rr);
    fp_sub(&X3, &X3, &HHH);
    fp_dbl(&t, &V);
    fp_sub(&X3, &X3, &t);
    fp_sub(&t, &V, &X3);
    fp_mul(&Y3, &Rr, &t);
    fp_mul(&t, &S1, &HHH);
    fp_sub(&Y3, &Y3, &t);
    fp_mul(&Z3, &p->z, &q->z);
    fp_mul(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void g1a_from_bytes96(g1a* r, const uint8_t* b)
{
    if (b[0] & 0x40) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }
    uint8_t xb[48];
    memcpy(xb, b, 48);
    xb[0] &= 0x1f; /* strip the three flag bits */
    fp_from_be48(&r->x, xb);
    fp_from_be48(&r->y, b + 48);
    r->inf = 0;
}

static void g1_write_affine96(uint8_t* out, const fp* x, const fp* y)
{
    fp_to_be48(out, x);
    fp_to_be48(out + 48, y);
}

static void g1j_to_bytes96(uint8_t* out, const g1j* p)
{
    if (g1j_is_inf(p)) { memset(out, 0, 96); out[0] = 0x40; return; }
    fp zi, zi2, zi3, x, y;
    fp_inv(&zi, &p->z);
    fp_sqr(&zi2, &zi);
    fp_mul(&zi3, &zi2, &zi);
    fp_mul(&x, &p->x, &zi2);
    fp_mul(&y, &p->y, &zi3);
    g1_write_affine96(out, &x, &y);
}

/* 1 if the 96-byte point is infinity or satisfies y^2 = x^3 + 4 with x, y < p. */
int po_g1_is_on_curve(const uint8_t* p96)
{
    if (p96[0] & 0x40) return 1;
    g1a a;
    g1a_from_bytes96(&a, p96);
    /* canonical check: re-encode and compare (rejects x, y >= p) */
    uint8_t chk[96];
    g1_write_affine96(chk, &a.x, &a.y);
    uint8_t in[96];
    memcpy(in, p96, 96);
    in[0] &= 0x1f;
    if (memcmp(chk, in, 96) != 0) return 0;
    fp y2, x3, four;
    fp_sqr(&y2, &a.y);
    fp_sqr(&x3, &a.x);
    fp_mul(&x3, &x3, &a.x);
    fp_dbl(&four, &FP_R1);
    fp_dbl(&four, &four);
    fp_add(&x3, &x3, &four);
    return fp_eq(&y2, &x3);
}

/*
 * out[g] = sum_{j in [offsets[g], offsets[g+1])} points[index ? index[j] : j]
 * points: 96-byte uncompressed affine; out: 96-byte canonical affine / infinity.
 */
int po_g1_sum_groups(const uint8_t* points96, uint64_t n_points, const uint32_t* index,
                     const uint32_t* offsets, uint32_t n_groups, uint8_t* out96)
{
    for (uint32_t g = 0; g < n_groups; ++g) {
        g1j acc;
        g1j_set_inf(&acc);
        for (uint32_t j = offsets[g]; j < offsets[g + 1]; ++j) {
            uint64_t k = index ? index[j] : j;
            if (k >= n_points) return -1;
            g1a q;
            g1a_from_bytes96(&q, points96 + 96 * k);
            g1j t;
            g1j_add_affine(&t, &acc, &q);
            acc = t;
        }
        g1j_to_bytes96(out96 + 96 * (size_t)g, &acc);
    }
    return 0;
}

/* out[i] = A + i*B for i in [0, n): one Jacobian running sum, batch-normalised
 * (Montgomery's trick) in chunks.  Used to build synthetic pubkey tables whose
 * subset sums have the closed form |S|*A + (sum i)*B. */
int po_g1_arith_progression(const uint8_t* a96, const uint8_t* b96, uint64_t n, uint8_t* out96)
{
    enum { CH = 1024 };
    g1a A, B;
    g1a_from_bytes96(&A, a96);
    g1a_from_bytes96(&B, b96);
    g1j cur;
    g1j_set_inf(&cur);
    { g1j t; g1j_add_affine(&t, &cur, &A); cur = t; }
    g1j* buf = (g1j*)malloc(sizeof(g1j) * CH);
    fp* pref = (fp*)malloc(sizeof(fp) * CH);
    if (!buf || !pref) { free(buf); free(pref); return -1; }
    for (uint64_t base = 0; base < n; base += CH) {
        uint32_t m = (uint32_t)((n - base < CH) ? (n - base) : CH);
        for (uint32_t i = 0; i < m; ++i) {
            buf[i] = cur;
            g1j t;
            g1j_add_affine(&t, &cur, &B);
            cur = t;
        }
        /* prefix products of the non-zero z's */
        fp run = FP_R1;
        for (uint32_t i = 0; i < m; ++i) {
            pref[i] = run;
            if (!g1j_is_inf(&buf[i])) fp_mul(&run, &run, &buf[i].z);
        }
        fp inv;
        fp_inv(&inv, &run);
        for (uint32_t i = m; i-- > 0;) {
            uint8_t* o = out96 + 96 * (size_t)(base + i);
            if (g1j_is_inf(&buf[i])) { memset(o, 0, 96); o[0] = 0x40; continue; }
            fp zi, zi2, zi3, x, y;
            fp_mul(&zi, &inv, &pref[i]);
            fp_mul(&inv, &inv, &buf[i].z);
            fp_sqr(&zi2, &zi);
            fp_mul(&zi3, &zi2, &zi);
            fp_mul(&x, &buf[i].x, &zi2);
            fp_mul(&y, &buf[i].y, &zi3);
            g1_write_affine96(o, &x, &y);
        }
    }
    free(buf);
    free(pref);
    return 0;
}

/* k*P by double-and-add, k = 32-byte big-endian scalar. */
int po_g1_scalar_mul(const uint8_t* k_be32, const uint8_t* p96, uint8_t* out96)
{
    g1a P;
    g1a_from_bytes96(&P, p96);
    g1j acc;
    g1j_set_inf(&acc);
    for (int i = 0; i < 256; ++i) {
        g1j t;
        g1j_double(&t, &acc);
        acc = t;
        if ((k_be32[i >> 3] >> (7 - (i & 7))) & 1) { g1j_add_affine(&t, &acc, &P); acc = t; }
    }
    g1j_to_bytes96(out96, &acc);
    return 0;
}

/* Jacobian partial of one shard, then a finishing sum: mirrors the multi-GPU
 * exchange so the shard arithmetic can be checked on CPU.
 * Partials are 144 bytes: X|Y|Z, each 6 little-endian u64 limbs in Montgomery
 * form -- byte-identical to 12 little-endian u32 limbs. */
int po_g1_partial_groups(const uint8_t* points96, uint64_t n_points, const uint32_t* index,
                         const uint32_t* offsets, uint32_t n_groups, uint8_t* out144)
{
    for (uint32_t g = 0; g < n_groups; ++g) {
        g1j acc;
        g1j_set_inf(&acc);
        for (uint32_t j = offsets[g]; j < offsets[g + 1]; ++j) {
            uint64_t k = index ? index[j] : j;
            if (k >= n_points) return -1;
            g1a q;
            g1a_from_bytes96(&q, points96 + 96 * k);
            g1j t;
            g1j_add_affine(&t, &acc, &q);
            acc = t;
        }
        memcpy(out144 + 144 * (size_t)g, &acc, 144);
    }
    return 0;
}

int po_g1_finish_partials(const uint8_t* gathered144, uint32_t n_ranks, uint32_t n_groups, uint8_t* out96)
{
    for (uint32_t g = 0; g < n_groups; ++g) {
        g1j acc;
        g1j_set_inf(&acc);
        for (uint32_t r = 0; r < n_ranks; ++r) {
            g1j q, t;
            memcpy(&q, gathered144 + 144 * ((size_t)r * n_groups + g), 144);
            g1j_add(&t, &acc, &q);
            acc = t;
        }
        g1j_to_bytes96(out96 + 96 * (size_t)g, &acc);
    }
    return 0;
}


/* ======================================================================= */
/* All-cores forms (OpenMP) for bench.py's cpu_baseline leg: the SAME loops  */
/* split over independent units.  tests/test_oracle_cport.py holds them equal */
/* to the single-thread functions above, which stay the checker of record.   */
/* ======================================================================= */
#ifdef _OPENMP
#include <omp.h>
int po_max_threads(void) { return omp_get_max_threads(); }
void po_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int po_max_threads(void) { return 1; }
void po_set_threads(int n) { (void)n; }
#endif

/* groups are independent */
void po_bits_union_mt(uint32_t n_groups, const uint32_t* group_start, const uint32_t* att_list,
                      const uint32_t* att_bits_off, const uint8_t* arena, const uint32_t* group_n_bits,
                      const uint32_t* out_bits_off, uint8_t* out_arena, uint32_t* out_count)
{
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < (int64_t)n_groups; ++g) {
        const uint32_t gs[2] = {group_start[g], group_start[g + 1]};
        po_bits_union(1, gs, att_list, att_bits_off, arena, group_n_bits + g, out_bits_off + g, out_arena,
                      out_count ? out_count + g : 0);
    }
}

int po_g1_sum_groups_mt(const uint8_t* points96, uint64_t n_points, const uint32_t* index,
                        const uint32_t* offsets, uint32_t n_groups, uint8_t* out96)
{
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t g = 0; g < (int64_t)n_groups; ++g) {
        const uint32_t off[2] = {offsets[g], offsets[g + 1]};
        const int r = po_g1_sum_groups(points96, n_points, index, off, 1, out96 + 96 * (size_t)g);
        if (r) {
#pragma omp atomic write
            rc = r;
        }
    }
    return rc;
}

/* Aggregate pubkey per attestation straight from its bits (get_attesting_indices + the sum of FastAggregateVerify,
 * A.6/A.7, without materialising the index list): out[a] = sum of points[members[member_off[a] + i]] over set bits i. */
int po_g1_sum_attesters(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits, const uint32_t* bits_off,
                        const uint8_t* arena, const uint32_t* members, const uint8_t* points96, uint64_t n_points,
                        uint8_t* out96)
{
    for (uint32_t a = 0; a < n_att; ++a) {
        const uint8_t* bits = arena + bits_off[a];
        g1j acc;
        g1j_set_inf(&acc);
        for (uint32_t i = 0; i < n_bits[a]; ++i) {
            if (!((bits[i >> 3] >> (i & 7)) & 1)) continue;
            const uint64_t k = members[member_off[a] + i];
            if (k >= n_points) return -1;
            g1a q;
            g1a_from_bytes96(&q, points96 + 96 * k);
            g1j t;
            g1j_add_affine(&t, &acc, &q);
            acc = t;
        }
        g1j_to_bytes96(out96 + 96 * (size_t)a, &acc);
    }
    return 0;
}
int po_g1_sum_attesters_mt(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits, const uint32_t* bits_off,
                           const uint8_t* arena, const uint32_t* members, const uint8_t* points96, uint64_t n_points,
                           uint8_t* out96)
{
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t a = 0; a < (int64_t)n_att; ++a) {
        const int r = po_g1_sum_attesters(1, member_off + a, n_bits + a, bits_off + a, arena, members, points96, n_points,
                                          out96 + 96 * (size_t)a);
        if (r) {
#pragma omp atomic write
            rc = r;
        }
    }
    return rc;
}

/* PRECONDITION: the committees of the batch are pairwise disjoint (one aggregate per committee of a table that
 * partitions the validators -- bench.py's step): attestations then touch disjoint validators and any order equals the
 * sequential one. */
void po_update_latest_messages_mt(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits,
                                  const uint32_t* bits_off, const uint64_t* target_epoch,
                                  const uint32_t* block_idx, const uint8_t* arena, const uint32_t* members,
                                  const uint8_t* val_flags, uint64_t* vote_epoch, uint32_t* vote_block)
{
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < (int64_t)n_att; ++a)
        po_update_latest_messages(1, member_off + a, n_bits + a, bits_off + a, target_epoch + a, block_idx + a, arena,
                                  members, val_flags, vote_epoch, vote_block);
}

/* same precondition */
void po_process_attestation_flags_mt(uint32_t n_att, const uint32_t* member_off, const uint32_t* n_bits,
                                     const uint32_t* bits_off, const uint8_t* flag_mask, const uint8_t* which,
                                     const uint8_t* arena, const uint32_t* members, const uint64_t* eff_balance,
                                     uint64_t increment, uint64_t base_reward_per_increment,
                                     uint8_t* participation_current, uint8_t* participation_previous,
                                     uint64_t* out_numerators)
{
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < (int64_t)n_att; ++a)
        po_process_attestation_flags(1, member_off + a, n_bits + a, bits_off + a, flag_mask + a, which + a, arena,
                                     members, eff_balance, increment, base_reward_per_increment,
                                     participation_current, participation_previous, out_numerators + a);
}

/* get_head with the O(V) vote scan split over validator ranges (one private weight array per thread, summed after);
 * the O(B) tree part is the single-thread code. */
int po_get_head_mt(uint32_t n_blocks, const uint32_t* parent, const uint8_t* leaf_ok, const uint8_t* roots,
                   uint64_t n_val, const uint32_t* vote_block, const uint64_t* eff_balance,
                   const uint8_t* flags, int filter_slashed, uint32_t justified_idx, uint32_t boost_idx,
                   uint64_t slots_per_epoch, uint64_t boost_percent, uint64_t balance_increment,
                   uint64_t* out_weights, uint32_t* out_head)
{
    if (n_blocks == 0 || justified_idx >= n_blocks) return -1;
    int nt = po_max_threads();
    if (nt < 1) nt = 1;
    uint64_t* priv = (uint64_t*)calloc((size_t)nt * (n_blocks + 2), sizeof(uint64_t));
    if (!priv) return -4;
    int bad = 0;
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        uint64_t* w = priv + (size_t)t * (n_blocks + 2);
        const uint64_t lo = n_val * (uint64_t)t / (uint64_t)T, hi = n_val * (uint64_t)(t + 1) / (uint64_t)T;
        for (uint64_t v = lo; v < hi; ++v) {
            const uint8_t f = flags[v];
            if (!(f & PO_VAL_ACTIVE)) continue;
            w[n_blocks] += eff_balance[v];
            w[n_blocks + 1] += 1;
            if (f & PO_VAL_EQUIVOCATING) continue;
            if (filter_slashed && (f & PO_VAL_SLASHED)) continue;
            const uint32_t b = vote_block[v];
            if (b == PO_NONE) continue;
            if (b >= n_blocks) { bad = 1; continue; }
            w[b] += eff_balance[v];
        }
    }
    if (bad) { free(priv); return -2; }
    /* a synthetic one-validator-per-block registry carries the summed direct weights into the single-thread code */
    uint64_t* bal = (uint64_t*)calloc((size_t)n_blocks + 1, sizeof(uint64_t));
    uint32_t* vb = (uint32_t*)calloc((size_t)n_blocks + 1, sizeof(uint32_t));
    uint8_t* fl = (uint8_t*)calloc((size_t)n_blocks + 1, 1);
    if (!bal || !vb || !fl) { free(priv); free(bal); free(vb); free(fl); return -4; }
    uint64_t total_active = 0, num_active = 0;
    for (int t = 0; t < nt; ++t) {
        const uint64_t* w = priv + (size_t)t * (n_blocks + 2);
        for (uint32_t b = 0; b < n_blocks; ++b) bal[b] += w[b];
        total_active += w[n_blocks];
        num_active += w[n_blocks + 1];
    }
    free(priv);
    for (uint32_t b = 0; b < n_blocks; ++b) { vb[b] = b; fl[b] = PO_VAL_ACTIVE; }
    /* boost needs the true totals: compute the proposer score here and hand it over as one more "validator" */
    uint64_t n_syn = n_blocks;
    if (boost_idx != PO_NONE && num_active > 0) {
        if (boost_idx >= n_blocks) { free(bal); free(vb); free(fl); return -3; }
        if (total_active < balance_increment) total_active = balance_increment;
        const uint64_t avg_balance = total_active / num_active;
        const uint64_t committee_weight = (num_active / slots_per_epoch) * avg_balance;
        bal[n_blocks] = (uint64_t)(((u128)committee_weight * boost_percent) / 100);
        vb[n_blocks] = boost_idx;
        fl[n_blocks] = PO_VAL_ACTIVE;
        n_syn = n_blocks + 1;
    }
    const int rc = po_get_head(n_blocks, parent, leaf_ok, roots, n_syn, vb, bal, fl, 0, justified_idx, PO_NONE,
                               slots_per_epoch, boost_percent, balance_increment, out_weights, out_head);
    free(bal); free(vb); free(fl);
    return rc;
}
