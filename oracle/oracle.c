/* oracle.c -- plain-C CPU restatement of the arkworks-rs/sumcheck prover path (BLS12-381 Fr).
 *
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/liboracle.so by oracle/Makefile.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker /
 * the timed CPU baseline ("kind": "port").  The product library (sumcheck_amd/csrc) never links,
 * loads or calls anything in this file.
 *
 * PARITY STATUS: "parity unpinned" for the ark-ff / ark-serialize / ark-poly semantics that are
 * not in /root/reference (F::rand, CanonicalSerialize, SparseMultilinearExtension::fix_variables);
 * see oracle/pyoracle.py's header and DESIGN.md section 3.  The reference is Rust; no cargo/rustc
 * in the image, so oracle/_ref cannot be built.  This file is cross-checked against the independent
 * big-integer oracle (oracle/pyoracle.py) and the committed fixtures in tests/golden/.
 *
 * Arithmetic: 4 x u64 little-endian limbs, Montgomery form R = 2^256, CIOS with unsigned __int128
 * (the portable ark-ff MontBackend shape).  All values canonical in [0, p).
 *
 * Each function cites the reference file:line it follows (relative to the reference repo root).
 */
#define _POSIX_C_SOURCE 200112L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;

static const fr_t FR_P = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const fr_t FR_ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}}; /* R mod p */
static const fr_t FR_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const uint64_t FR_INV = 0xfffffffeffffffffULL; /* -p^-1 mod 2^64 */

static inline int fr_geq_p(const fr_t *a) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > FR_P.l[i]) return 1;
        if (a->l[i] < FR_P.l[i]) return 0;
    }
    return 1;
}
static inline void fr_sub_p(fr_t *a) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - FR_P.l[i] - br;
        a->l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline fr_t fr_add(fr_t a, fr_t b) {
    fr_t r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (fr_geq_p(&r)) fr_sub_p(&r); /* top bit spare: no carry out of limb 3 */
    return r;
}
static inline fr_t fr_sub(fr_t a, fr_t b) {
    fr_t r; u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + FR_P.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static inline fr_t fr_mul(fr_t a, fr_t b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_INV;
        c = ((u128)m * FR_P.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * FR_P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr_t r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq_p(&r)) fr_sub_p(&r);
    return r;
}
static inline int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof(fr_t)) == 0; }
static inline fr_t fr_zero(void) { fr_t z = {{0, 0, 0, 0}}; return z; }
static inline fr_t fr_from_u64(uint64_t x) { fr_t a = {{x, 0, 0, 0}}; return fr_mul(a, FR_R2); }
static inline fr_t fr_neg(fr_t a) { return fr_sub(fr_zero(), a); }
static fr_t fr_pow(fr_t a, const uint64_t e[4]) {
    fr_t r = FR_ONE;
    for (int i = 255; i >= 0; --i) {
        r = fr_mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = fr_mul(r, a);
    }
    return r;
}
static fr_t fr_inv(fr_t a) { /* Fermat: a^(p-2) */
    uint64_t e[4] = {FR_P.l[0] - 2, FR_P.l[1], FR_P.l[2], FR_P.l[3]};
    return fr_pow(a, e);
}

/* ------------------------------------------------------------------------------------------ */
/* exported field helpers (for tests)                                                           */
/* ------------------------------------------------------------------------------------------ */
void orc_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { *(fr_t *)out = fr_mul(*(const fr_t *)a, *(const fr_t *)b); }
void orc_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out) { *(fr_t *)out = fr_add(*(const fr_t *)a, *(const fr_t *)b); }
void orc_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out) { *(fr_t *)out = fr_sub(*(const fr_t *)a, *(const fr_t *)b); }
void orc_fr_inv(const uint64_t *a, uint64_t *out) { *(fr_t *)out = fr_inv(*(const fr_t *)a); }
/* canonical integer limbs <-> Montgomery limbs */
void orc_fr_to_mont(const uint64_t *canon, uint64_t *out) { *(fr_t *)out = fr_mul(*(const fr_t *)canon, FR_R2); }
void orc_fr_from_mont(const uint64_t *mont, uint64_t *out) { fr_t one = {{1, 0, 0, 0}}; *(fr_t *)out = fr_mul(*(const fr_t *)mont, one); }

/* ------------------------------------------------------------------------------------------ */
/* DenseMultilinearExtension::fix_variables (ark-poly, external)            */
/* called at reference src/ml_sumcheck/protocol/prover.rs:88                                    */
/* ------------------------------------------------------------------------------------------ */
/* in: 2^nv elements; out: 2^(nv-k) elements (may alias a scratch).  LSB-first binding. */
void orc_fix_variables(const uint64_t *in, uint32_t nv, const uint64_t *point, uint32_t k, uint64_t *out) {
    size_t n = (size_t)1 << nv;
    fr_t *poly = (fr_t *)malloc(n * sizeof(fr_t));
    memcpy(poly, in, n * sizeof(fr_t));
    for (uint32_t i = 0; i < k; ++i) {
        fr_t r = ((const fr_t *)point)[i];
        size_t half = n >> 1;
        for (size_t b = 0; b < half; ++b) {
            fr_t lo = poly[2 * b], hi = poly[2 * b + 1];
            poly[b] = fr_add(lo, fr_mul(r, fr_sub(hi, lo)));
        }
        n = half;
    }
    memcpy(out, poly, n * sizeof(fr_t));
    free(poly);
}

/* ------------------------------------------------------------------------------------------ */
/* Prover: reference src/ml_sumcheck/protocol/prover.rs                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t num_vars, max_multiplicands, round, n_products, n_tables;
    fr_t *coeffs;            /* K */
    uint32_t *offsets;       /* K+1 */
    uint32_t *indices;       /* sum m_k */
    fr_t **tables;           /* U, each 2^(num_vars - bound) */
    fr_t *randomness;        /* up to num_vars */
    uint32_t n_rand;
    int threads;             /* 1 = the non-parallel build; >1 mirrors the rayon feature */
    int improved_fix;        /* 0 = fix parallel across tables only (prover.rs:87); 1 = inside tables */
    double t_init, t_bind, t_sum; /* bench.py's per-phase split: deep copy | fix_variables | the fold of prover.rs:110-148 (seconds) */
} orc_prover;

enum { ORC_OK = 0, ORC_ERR_CONSTANT = 1, ORC_ERR_FIRST_ROUND_HAS_MSG = 2, ORC_ERR_MISSING_MSG = 3, ORC_ERR_NOT_ACTIVE = 4 };

/* prover.rs:49-69 -- deep copy of every unique table */
int orc_prover_init(uint32_t num_vars, uint32_t max_multiplicands, uint32_t n_products, const uint64_t *coeffs,
                    const uint32_t *offsets, const uint32_t *indices, uint32_t n_tables, const uint64_t *const *tables,
                    int threads, orc_prover **out) {
    if (num_vars == 0) return ORC_ERR_CONSTANT; /* "Attempt to prove a constant." prover.rs:50-52 */
    orc_prover *p = (orc_prover *)calloc(1, sizeof(orc_prover));
    p->num_vars = num_vars; p->max_multiplicands = max_multiplicands; p->n_products = n_products; p->n_tables = n_tables;
    p->coeffs = (fr_t *)malloc(n_products * sizeof(fr_t)); memcpy(p->coeffs, coeffs, n_products * sizeof(fr_t));
    p->offsets = (uint32_t *)malloc((n_products + 1) * sizeof(uint32_t)); memcpy(p->offsets, offsets, (n_products + 1) * sizeof(uint32_t));
    uint32_t tot = offsets[n_products];
    p->indices = (uint32_t *)malloc(tot * sizeof(uint32_t)); memcpy(p->indices, indices, tot * sizeof(uint32_t));
    p->tables = (fr_t **)malloc(n_tables * sizeof(fr_t *));
    size_t n = (size_t)1 << num_vars;
    const double t0 = orc_now();
    /* the deep copy of prover.rs:55-59.  With threads > 1 every table is copied in slices by all threads, so that its pages are
     * first touched (and placed) by the cores that will read them; the reference's Vec::clone is one memcpy per table. */
    for (uint32_t u = 0; u < n_tables; ++u) {
        p->tables[u] = (fr_t *)malloc(n * sizeof(fr_t));
        if (threads > 1 && n >= ((size_t)1 << 16)) {
            const size_t slices = n >> 12; /* 128 KiB slices */
#pragma omp parallel for num_threads(threads) schedule(static)
            for (size_t s = 0; s < slices; ++s) memcpy(p->tables[u] + (s << 12), (const fr_t *)tables[u] + (s << 12), sizeof(fr_t) << 12);
        } else {
            memcpy(p->tables[u], tables[u], n * sizeof(fr_t));
        }
    }
    p->t_init = orc_now() - t0;
    p->randomness = (fr_t *)malloc(num_vars * sizeof(fr_t));
    p->threads = threads < 1 ? 1 : threads;
    *out = p;
    return ORC_OK;
}
void orc_prover_set_improved_fix(orc_prover *p, int on) { p->improved_fix = on; }

void orc_prover_free(orc_prover *p) {
    if (!p) return;
    for (uint32_t u = 0; u < p->n_tables; ++u) free(p->tables[u]);
    free(p->tables); free(p->coeffs); free(p->offsets); free(p->indices); free(p->randomness); free(p);
}

/* one table halving: ark-poly allocates a fresh vector per call; so do we */
static void fix_one_table(orc_prover *p, uint32_t u, size_t n_in, fr_t r, int inner_threads) {
    size_t half = n_in >> 1;
    fr_t *src = p->tables[u];
    fr_t *dst = (fr_t *)malloc((half ? half : 1) * sizeof(fr_t));
    (void)inner_threads;
#pragma omp parallel for num_threads(inner_threads) schedule(static) if (inner_threads > 1)
    for (size_t b = 0; b < half; ++b) {
        fr_t lo = src[2 * b], hi = src[2 * b + 1];
        dst[b] = fr_add(lo, fr_mul(r, fr_sub(hi, lo)));
    }
    free(src);
    p->tables[u] = dst;
}

/* prover.rs:74-153.  r_or_null: 4 limbs Montgomery or NULL.  out_evals: (max_multiplicands+1) x 4 limbs. */
int orc_prove_round(orc_prover *p, const uint64_t *r_or_null, uint64_t *out_evals) {
    /* validation first (same precedence as the reference's panics), then mutate */
    if (r_or_null && p->round == 0) return ORC_ERR_FIRST_ROUND_HAS_MSG; /* prover.rs:79-81 */
    if (!r_or_null && p->round > 0) return ORC_ERR_MISSING_MSG;         /* prover.rs:90-92 */
    if (p->round + 1 > p->num_vars) return ORC_ERR_NOT_ACTIVE;          /* prover.rs:96-98 */
    const double t_b0 = orc_now();
    if (r_or_null) {
        fr_t r = *(const fr_t *)r_or_null;
        p->randomness[p->n_rand++] = r;                                 /* prover.rs:82 */
        size_t n_in = (size_t)1 << (p->num_vars - (p->round - 1));
        if (p->improved_fix) {
            for (uint32_t u = 0; u < p->n_tables; ++u) fix_one_table(p, u, n_in, r, p->threads);
        } else {
            /* cfg_iter_mut!(flattened_ml_extensions): parallel ACROSS tables only, prover.rs:87-89 */
#pragma omp parallel for num_threads(p->threads) schedule(dynamic, 1) if (p->threads > 1)
            for (uint32_t u = 0; u < p->n_tables; ++u) fix_one_table(p, u, n_in, r, 1);
        }
    }
    p->round += 1;
    const double t_s0 = orc_now();
    p->t_bind += t_s0 - t_b0;

    const uint32_t i = p->round, nv = p->num_vars, degree = p->max_multiplicands;
    const size_t npts = (size_t)1 << (nv - i);
    const uint32_t D = degree + 1;
    fr_t *total = (fr_t *)calloc(D, sizeof(fr_t));
    /* cfg_into_iter!(0..1<<(nv-i), 1<<10).fold(...) + reduce: prover.rs:110-148.  Splits of at least
     * 1024 points; each split owns (products_sum, product) scratch; splits are summed at the end. */
    size_t nchunks = (npts + 1023) / 1024;
    int nthr = p->threads;
    if ((size_t)nthr > nchunks) nthr = (int)nchunks;
    if (nthr < 1) nthr = 1;
#pragma omp parallel num_threads(nthr) if (nthr > 1)
    {
        fr_t *products_sum = (fr_t *)calloc(D, sizeof(fr_t));
        fr_t *product = (fr_t *)calloc(D, sizeof(fr_t));
#pragma omp for schedule(dynamic, 1) nowait
        for (size_t ch = 0; ch < nchunks; ++ch) {
            size_t b0 = ch * 1024, b1 = b0 + 1024 < npts ? b0 + 1024 : npts;
            for (size_t b = b0; b < b1; ++b) {
                for (uint32_t k = 0; k < p->n_products; ++k) {
                    for (uint32_t t = 0; t < D; ++t) product[t] = p->coeffs[k];          /* prover.rs:116 */
                    for (uint32_t q = p->offsets[k]; q < p->offsets[k + 1]; ++q) {
                        const fr_t *table = p->tables[p->indices[q]];
                        fr_t start = table[b << 1];                                      /* prover.rs:119 */
                        fr_t step = fr_sub(table[(b << 1) + 1], start);                  /* prover.rs:120 */
                        for (uint32_t t = 0; t < D; ++t) {                               /* prover.rs:121-124 */
                            product[t] = fr_mul(product[t], start);
                            start = fr_add(start, step);
                        }
                    }
                    for (uint32_t t = 0; t < D; ++t) products_sum[t] = fr_add(products_sum[t], product[t]); /* 126-128 */
                }
            }
        }
#pragma omp critical
        { for (uint32_t t = 0; t < D; ++t) total[t] = fr_add(total[t], products_sum[t]); } /* prover.rs:139-148 */
        free(products_sum); free(product);
    }
    memcpy(out_evals, total, D * sizeof(fr_t));
    free(total);
    p->t_sum += orc_now() - t_s0;
    return ORC_OK;
}
/* seconds spent so far in: [0] prover_init's deep copy, [1] fix_variables (prover.rs:84-89), [2] the sums (prover.rs:100-153) */
void orc_prover_times(const orc_prover *p, double out[3]) { out[0] = p->t_init; out[1] = p->t_bind; out[2] = p->t_sum; }

/* ml_sumcheck/mod.rs:65-67: the last challenge is recorded but never bound */
void orc_prover_push_randomness(orc_prover *p, const uint64_t *r) {
    if (p->n_rand < p->num_vars) p->randomness[p->n_rand++] = *(const fr_t *)r;
}
/* copy out state: randomness (n_rand x 4), tables (U x 2^(nv-bound) x 4), round */
uint32_t orc_prover_state(orc_prover *p, uint64_t *randomness, uint64_t *tables_out, uint32_t *round) {
    uint32_t bound = p->round > 0 ? p->round - 1 : 0;
    size_t n = (size_t)1 << (p->num_vars - bound);
    if (randomness) memcpy(randomness, p->randomness, p->n_rand * sizeof(fr_t));
    if (tables_out) for (uint32_t u = 0; u < p->n_tables; ++u) memcpy(tables_out + 4 * n * u, p->tables[u], n * sizeof(fr_t));
    if (round) *round = p->round;
    return p->n_rand;
}

/* ------------------------------------------------------------------------------------------ */
/* Verifier: reference src/ml_sumcheck/protocol/verifier.rs (the parity checker)                */
/* ------------------------------------------------------------------------------------------ */
/* verifier.rs:139-251.  The u64/u128/BigInt tiers compute the same field value; restated as plain
 * Lagrange evaluation with field inversion (exact arithmetic => identical result). */
void orc_interpolate_uni_poly(const uint64_t *p_i, uint32_t len, const uint64_t *eval_at, uint64_t *out) {
    const fr_t *pv = (const fr_t *)p_i; fr_t x = *(const fr_t *)eval_at;
    fr_t check = fr_zero();
    for (uint32_t i = 0; i < len; ++i) { /* early return when eval_at is a node (verifier.rs:152-164) */
        if (fr_eq(&x, &check)) { *(fr_t *)out = pv[i]; return; }
        check = fr_add(check, FR_ONE);
    }
    fr_t res = fr_zero();
    for (uint32_t i = 0; i < len; ++i) {
        fr_t num = FR_ONE, den = FR_ONE;
        for (uint32_t j = 0; j < len; ++j) {
            if (j == i) continue;
            num = fr_mul(num, fr_sub(x, fr_from_u64(j)));
            fr_t d = i > j ? fr_from_u64(i - j) : fr_neg(fr_from_u64(j - i));
            den = fr_mul(den, d);
        }
        res = fr_add(res, fr_mul(fr_mul(pv[i], num), fr_inv(den)));
    }
    *(fr_t *)out = res;
}

/* verifier.rs:90-121.  polys: nv x D x 4 limbs; randomness: nv x 4.  returns 0 accept, 1 reject.
 * expected_out = subclaim.expected_evaluation (subclaim.point == randomness). */
int orc_check_and_generate_subclaim(uint32_t nv, uint32_t max_multiplicands, const uint64_t *polys, const uint64_t *randomness,
                                    const uint64_t *asserted_sum, uint64_t *expected_out) {
    uint32_t D = max_multiplicands + 1;
    fr_t expected = *(const fr_t *)asserted_sum;
    for (uint32_t i = 0; i < nv; ++i) {
        const fr_t *ev = (const fr_t *)polys + (size_t)i * D;
        fr_t s = fr_add(ev[0], ev[1]);
        if (!fr_eq(&s, &expected)) return 1; /* "Prover message is not consistent with the claim." */
        orc_interpolate_uni_poly((const uint64_t *)ev, D, randomness + 4 * i, (uint64_t *)&expected);
    }
    *(fr_t *)expected_out = expected;
    return 0;
}

/* ListOfProductsOfPolynomials::evaluate, data_structures.rs:99-109 */
void orc_poly_evaluate(uint32_t num_vars, uint32_t n_products, const uint64_t *coeffs, const uint32_t *offsets,
                       const uint32_t *indices, uint32_t n_tables, const uint64_t *const *tables, const uint64_t *point,
                       uint64_t *out) {
    fr_t *tv = (fr_t *)malloc(n_tables * sizeof(fr_t));
    for (uint32_t u = 0; u < n_tables; ++u) orc_fix_variables(tables[u], num_vars, point, num_vars, (uint64_t *)&tv[u]);
    fr_t acc = fr_zero();
    for (uint32_t k = 0; k < n_products; ++k) {
        fr_t pr = ((const fr_t *)coeffs)[k];
        for (uint32_t q = offsets[k]; q < offsets[k + 1]; ++q) pr = fr_mul(pr, tv[indices[q]]);
        acc = fr_add(acc, pr);
    }
    free(tv);
    *(fr_t *)out = acc;
}

/* ------------------------------------------------------------------------------------------ */
/* BLAKE2b-512 (RFC 7693), unkeyed -- the `blake2` crate's Blake2b512 used at src/rng.rs:5,22-25  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t h[8]; uint64_t t[2]; uint8_t buf[128]; size_t buflen; } b2b_t;
static const uint64_t B2B_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void b2b_compress(b2b_t *S, const uint8_t *block, int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8); /* little-endian host */
    for (int i = 0; i < 8; ++i) { v[i] = S->h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= S->t[0]; v[13] ^= S->t[1];
    if (last) v[14] = ~v[14];
#define B2B_G(a, b, c, d, x, y) do { \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr64(v[b] ^ v[c], 63); } while (0)
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = B2B_SIGMA[r];
        B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]]); B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]]); B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]]); B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]]); B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2B_G
    for (int i = 0; i < 8; ++i) S->h[i] ^= v[i] ^ v[i + 8];
}
static void b2b_init(b2b_t *S) {
    memset(S, 0, sizeof(*S));
    for (int i = 0; i < 8; ++i) S->h[i] = B2B_IV[i];
    S->h[0] ^= 0x01010000ULL ^ 64; /* digest length 64, no key, fanout 1, depth 1 */
}
static void b2b_update(b2b_t *S, const uint8_t *in, size_t len) {
    while (len > 0) {
        if (S->buflen == 128) { /* buffer full and more input follows: compress it */
            S->t[0] += 128; if (S->t[0] < 128) S->t[1]++;
            b2b_compress(S, S->buf, 0); S->buflen = 0;
        }
        size_t take = 128 - S->buflen; if (take > len) take = len;
        memcpy(S->buf + S->buflen, in, take); S->buflen += take; in += take; len -= take;
    }
}
static void b2b_final(const b2b_t *S0, uint8_t out[64]) { /* non-destructive: works on a copy (rng.rs:62-63 clones) */
    b2b_t S = *S0;
    S.t[0] += S.buflen; if (S.t[0] < S.buflen) S.t[1]++;
    memset(S.buf + S.buflen, 0, 128 - S.buflen);
    b2b_compress(&S, S.buf, 1);
    memcpy(out, S.h, 64);
}
void orc_blake2b512(const uint8_t *in, size_t len, uint8_t *out64) { b2b_t S; b2b_init(&S); b2b_update(&S, in, len); b2b_final(&S, out64); }

/* Blake2b512Rng, reference src/rng.rs:22-81 */
typedef struct { b2b_t d; } orc_rng;
orc_rng *orc_rng_setup(void) { orc_rng *r = (orc_rng *)malloc(sizeof(orc_rng)); b2b_init(&r->d); return r; } /* rng.rs:30-34 */
void orc_rng_free(orc_rng *r) { free(r); }
void orc_rng_feed_bytes(orc_rng *r, const uint8_t *buf, size_t len) { b2b_update(&r->d, buf, len); }     /* rng.rs:36-41 */
void orc_rng_fill_bytes(orc_rng *r, uint8_t *dest, size_t n) {                                          /* rng.rs:61-80 */
    uint8_t output[64]; b2b_final(&r->d, output);
    size_t ptr = 0, dptr = 0;
    while (ptr < n) {
        dest[ptr++] = output[dptr++];
        if (dptr == 64) { b2b_update(&r->d, output, 64); b2b_final(&r->d, output); dptr = 0; }
    }
    b2b_update(&r->d, output, 64); /* rng.rs:78 */
}
static uint64_t rng_next_u64(orc_rng *r) { uint8_t t[8]; orc_rng_fill_bytes(r, t, 8); uint64_t x; memcpy(&x, t, 8); return x; }
/* sample_round (verifier.rs:128-131) = F::rand: ark-ff Fp sampler (raw limbs are Montgomery form). */
void orc_rng_sample_fr(orc_rng *r, uint64_t *out) {
    for (;;) {
        fr_t a; for (int i = 0; i < 4; ++i) a.l[i] = rng_next_u64(r);
        a.l[3] &= 0xffffffffffffffffULL >> 1;
        if (!fr_geq_p(&a)) { *(fr_t *)out = a; return; }
    }
}
/* feed(&ProverMsg) : u64 LE len + canonical 32-byte LE elements (prover.rs:13-17 + ark-serialize) */
void orc_rng_feed_prover_msg(orc_rng *r, const uint64_t *evals, uint32_t D) {
    uint64_t len = D; b2b_update(&r->d, (const uint8_t *)&len, 8);
    fr_t one = {{1, 0, 0, 0}};
    for (uint32_t t = 0; t < D; ++t) { fr_t c = fr_mul(((const fr_t *)evals)[t], one); b2b_update(&r->d, (const uint8_t *)&c, 32); }
}
/* feed(&PolynomialInfo): data_structures.rs:47-55 */
void orc_rng_feed_poly_info(orc_rng *r, uint64_t max_multiplicands, uint64_t num_variables) {
    b2b_update(&r->d, (const uint8_t *)&max_multiplicands, 8); b2b_update(&r->d, (const uint8_t *)&num_variables, 8);
}

/* MLSumcheck::prove_as_subprotocol, reference src/ml_sumcheck/mod.rs:50-70.  out_proof: nv x D x 4 limbs.
 * If out_randomness != NULL receives nv x 4 limbs.  rng may be pre-fed by the caller. */
static double g_last_prove_times[3];
void orc_last_prove_times(double out[3]) { memcpy(out, g_last_prove_times, sizeof(g_last_prove_times)); } /* of the last orc_ml_prove on this process */
int orc_ml_prove(orc_rng *rng, uint32_t num_vars, uint32_t max_multiplicands, uint32_t n_products, const uint64_t *coeffs,
                 const uint32_t *offsets, const uint32_t *indices, uint32_t n_tables, const uint64_t *const *tables, int threads,
                 uint64_t *out_proof, uint64_t *out_randomness) {
    orc_prover *p; int rc = orc_prover_init(num_vars, max_multiplicands, n_products, coeffs, offsets, indices, n_tables, tables, threads, &p);
    if (rc) return rc;
    orc_rng_feed_poly_info(rng, max_multiplicands, num_vars);
    uint32_t D = max_multiplicands + 1; fr_t vm; int have = 0;
    for (uint32_t i = 0; i < num_vars; ++i) {
        rc = orc_prove_round(p, have ? (const uint64_t *)&vm : NULL, out_proof + (size_t)i * D * 4);
        if (rc) { orc_prover_free(p); return rc; }
        orc_rng_feed_prover_msg(rng, out_proof + (size_t)i * D * 4, D);
        orc_rng_sample_fr(rng, (uint64_t *)&vm); have = 1;
    }
    orc_prover_push_randomness(p, (const uint64_t *)&vm);
    if (out_randomness) memcpy(out_randomness, p->randomness, num_vars * sizeof(fr_t));
    orc_prover_times(p, g_last_prove_times);
    orc_prover_free(p);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* SparseMultilinearExtension (ark-poly, external) + GKR init                */
/* ------------------------------------------------------------------------------------------ */
static void precompute_eq(const fr_t *g, uint32_t dim, fr_t *dp) {
    dp[0] = fr_sub(FR_ONE, g[0]); dp[1] = g[0];
    for (uint32_t i = 1; i < dim; ++i)
        for (size_t b = 0; b < ((size_t)1 << i); ++b) {
            fr_t prev = dp[b];
            dp[b + ((size_t)1 << i)] = fr_mul(prev, g[i]);
            dp[b] = fr_sub(prev, dp[b + ((size_t)1 << i)]);
        }
}
typedef struct { uint64_t idx; fr_t v; } sp_ent;
static int sp_cmp(const void *a, const void *b) { uint64_t x = ((const sp_ent *)a)->idx, y = ((const sp_ent *)b)->idx; return x < y ? -1 : x > y; }
/* Bind the low `k` variables of a sparse MLE (entries idx[], vals[]) at `point`; the windowed fold of
 * ark-poly is algebraically out[idx >> k] += eq(point, idx & mask) * v; result sorted by index with
 * duplicates merged (BTreeMap order).  Returns the number of output entries. */
uint64_t orc_sparse_fix_variables(const uint64_t *idx, const uint64_t *vals, uint64_t nnz, const uint64_t *point, uint32_t k,
                                  uint64_t *out_idx, uint64_t *out_vals) {
    uint32_t window = 0; while (((uint64_t)1 << window) < nnz) window++; /* ark_std::log2(nnz) */
    sp_ent *cur = (sp_ent *)malloc((nnz ? nnz : 1) * sizeof(sp_ent));
    for (uint64_t i = 0; i < nnz; ++i) { cur[i].idx = idx[i]; cur[i].v = ((const fr_t *)vals)[i]; }
    uint64_t n = nnz; uint32_t done = 0;
    while (done < k) {
        uint32_t rem = k - done, fl = (window > 0 && rem > window) ? window : rem;
        fr_t *pre = (fr_t *)malloc(((size_t)1 << fl) * sizeof(fr_t));
        precompute_eq((const fr_t *)point + done, fl, pre);
        for (uint64_t i = 0; i < n; ++i) {
            cur[i].v = fr_mul(pre[cur[i].idx & (((uint64_t)1 << fl) - 1)], cur[i].v);
            cur[i].idx >>= fl;
        }
        free(pre);
        qsort(cur, n, sizeof(sp_ent), sp_cmp);
        uint64_t w = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (w > 0 && cur[w - 1].idx == cur[i].idx) cur[w - 1].v = fr_add(cur[w - 1].v, cur[i].v);
            else cur[w++] = cur[i];
        }
        n = w; done += fl;
    }
    if (k == 0) qsort(cur, n, sizeof(sp_ent), sp_cmp);
    for (uint64_t i = 0; i < n; ++i) { out_idx[i] = cur[i].idx; ((fr_t *)out_vals)[i] = cur[i].v; }
    free(cur);
    return n;
}

/* initialize_phase_one, reference src/gkr_round_sumcheck/mod.rs:22-42.
 * f1: nnz entries over 3*dim variables; h_g out: 2^dim; f1_g out: up to nnz entries (sorted). */
uint64_t orc_gkr_phase_one(const uint64_t *idx, const uint64_t *vals, uint64_t nnz, uint32_t dim, const uint64_t *f3, const uint64_t *g,
                           uint64_t *h_g, uint64_t *f1g_idx, uint64_t *f1g_vals) {
    uint64_t n = orc_sparse_fix_variables(idx, vals, nnz, g, dim, f1g_idx, f1g_vals);
    fr_t *a = (fr_t *)h_g; size_t N = (size_t)1 << dim;
    for (size_t i = 0; i < N; ++i) a[i] = fr_zero();
    for (uint64_t i = 0; i < n; ++i) {
        fr_t v = ((const fr_t *)f1g_vals)[i];
        if (!fr_is_zero(&v)) {
            uint64_t x = f1g_idx[i] & (N - 1), y = f1g_idx[i] >> dim;   /* mod.rs:34-35 */
            a[x] = fr_add(a[x], fr_mul(v, ((const fr_t *)f3)[y]));      /* mod.rs:36 */
        }
    }
    return n;
}
/* initialize_phase_two, reference src/gkr_round_sumcheck/mod.rs:57-63: dense 2^dim table */
void orc_gkr_phase_two(const uint64_t *f1g_idx, const uint64_t *f1g_vals, uint64_t nnz, uint32_t dim, const uint64_t *u, uint64_t *f1_gu) {
    uint64_t *oi = (uint64_t *)malloc((nnz ? nnz : 1) * 8); uint64_t *ov = (uint64_t *)malloc((nnz ? nnz : 1) * 32);
    uint64_t n = orc_sparse_fix_variables(f1g_idx, f1g_vals, nnz, u, dim, oi, ov);
    fr_t *d = (fr_t *)f1_gu; size_t N = (size_t)1 << dim;
    for (size_t i = 0; i < N; ++i) d[i] = fr_zero();
    for (uint64_t i = 0; i < n; ++i) d[oi[i]] = ((const fr_t *)ov)[i];
    free(oi); free(ov);
}

/* GKRRoundSumcheck::prove, reference src/gkr_round_sumcheck/mod.rs:93-139.
 * out_proof: 2 x dim x 3 x 4 limbs (phase1 msgs then phase2 msgs); out_uv: 2 x dim x 4 (u then v). */
int orc_gkr_prove(orc_rng *rng, const uint64_t *idx, const uint64_t *vals, uint64_t nnz, uint32_t dim, const uint64_t *f2,
                  const uint64_t *f3, const uint64_t *g, int threads, uint64_t *out_proof, uint64_t *out_uv) {
    size_t N = (size_t)1 << dim;
    uint64_t *h_g = (uint64_t *)malloc(N * 32), *f1g_idx = (uint64_t *)malloc((nnz ? nnz : 1) * 8), *f1g_vals = (uint64_t *)malloc((nnz ? nnz : 1) * 32);
    uint64_t n1 = orc_gkr_phase_one(idx, vals, nnz, dim, f3, g, h_g, f1g_idx, f1g_vals);
    uint32_t offsets[2] = {0, 2}, indices[2] = {0, 1};
    const uint64_t *tabs[2] = {h_g, f2};
    orc_prover *p; int rc = orc_prover_init(dim, 2, 1, (const uint64_t *)&FR_ONE, offsets, indices, 2, tabs, threads, &p); /* mod.rs:45-54 */
    if (rc) return rc;
    fr_t vm; int have = 0; fr_t *u = (fr_t *)out_uv, *v = (fr_t *)out_uv + dim;
    for (uint32_t i = 0; i < dim; ++i) {
        uint64_t *pm = out_proof + (size_t)i * 12;
        orc_prove_round(p, have ? (const uint64_t *)&vm : NULL, pm);
        orc_rng_feed_prover_msg(rng, pm, 3);
        orc_rng_sample_fr(rng, (uint64_t *)&vm); have = 1; u[i] = vm;
    }
    orc_prover_free(p);
    uint64_t *f1_gu = (uint64_t *)malloc(N * 32);
    orc_gkr_phase_two(f1g_idx, f1g_vals, n1, dim, (const uint64_t *)u, f1_gu);
    fr_t f2_u; orc_fix_variables(f2, dim, (const uint64_t *)u, dim, (uint64_t *)&f2_u);     /* mod.rs:122 */
    fr_t *f3_f2u = (fr_t *)malloc(N * 32);
    for (size_t i = 0; i < N; ++i) f3_f2u[i] = fr_mul(f2_u, ((const fr_t *)f3)[i]);          /* mod.rs:71-75 */
    const uint64_t *tabs2[2] = {f1_gu, (const uint64_t *)f3_f2u};
    rc = orc_prover_init(dim, 2, 1, (const uint64_t *)&FR_ONE, offsets, indices, 2, tabs2, threads, &p);
    have = 0;
    for (uint32_t i = 0; i < dim; ++i) {
        uint64_t *pm = out_proof + (size_t)(dim + i) * 12;
        orc_prove_round(p, have ? (const uint64_t *)&vm : NULL, pm);
        orc_rng_feed_prover_msg(rng, pm, 3);
        orc_rng_sample_fr(rng, (uint64_t *)&vm); have = 1; v[i] = vm;
    }
    orc_prover_free(p);
    free(h_g); free(f1g_idx); free(f1g_vals); free(f1_gu); free(f3_f2u);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic inputs: SplitMix64 keyed by (seed, stream, index)               */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL; uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_synth_table(uint64_t seed, uint64_t stream, uint64_t first, uint64_t n, uint64_t *out) {
    uint64_t key = splitmix64(seed ^ (stream * 0xD1342543DE82EF95ULL));
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t index = first + i;
        for (uint64_t attempt = 0;; ++attempt) {
            fr_t a;
            for (int k = 0; k < 4; ++k) a.l[k] = splitmix64(key ^ (splitmix64(index * 4 + k) + attempt * 0x9E3779B97F4A7C15ULL));
            a.l[3] &= 0xffffffffffffffffULL >> 1;
            if (!fr_geq_p(&a)) { ((fr_t *)out)[i] = a; break; }
        }
    }
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
