/*
 * hodor_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See hodor_oracle.h.
 * PARITY STATUS: parity unpinned against the Rust binary (reference unbuildable here; its tests
 * hold no known-answer vectors).  Pinned against Python big-int maths + hashlib.blake2s.
 *
 * Plain C11 + pthreads + unsigned __int128.  Paths cited are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "hodor_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * Field: ff_ce `#[derive(PrimeField)]` semantics for a 4-limb modulus (src/bn256.rs:4-7,
 * src/experiments/mod.rs:18-21).  Montgomery R = 2^256; values kept in [0, p).
 * ------------------------------------------------------------------------------------------ */

static int geq4(const uint64_t a[4], const uint64_t b[4])
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}

static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    return carry;
}

void ofr_add(const ofield *f, ofr *a, const ofr *b)
{
    uint64_t c = add4(a->l, a->l, b->l);
    if (c || geq4(a->l, f->p)) sub4(a->l, a->l, f->p);
}

void ofr_sub(const ofield *f, ofr *a, const ofr *b)
{
    if (sub4(a->l, a->l, b->l)) add4(a->l, a->l, f->p);
}

void ofr_neg(const ofield *f, ofr *a)
{
    if (!ofr_is_zero(a)) sub4(a->l, f->p, a->l);
}

void ofr_dbl(const ofield *f, ofr *a)
{
    ofr t = *a;
    ofr_add(f, a, &t);
}

/* Montgomery product (CIOS), result in [0, p). */
void ofr_mul(const ofield *f, ofr *a, const ofr *b)
{
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)a->l[j] * b->l[i] + t[j] + c;
            t[j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        u128 s = (u128)t[4] + c;
        t[4] = (uint64_t)s;
        t[5] = (uint64_t)(s >> 64);

        uint64_t m = t[0] * f->pinv;
        s = (u128)m * f->p[0] + t[0];
        c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) {
            s = (u128)m * f->p[j] + t[j] + c;
            t[j - 1] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        s = (u128)t[4] + c;
        t[3] = (uint64_t)s;
        t[4] = t[5] + (uint64_t)(s >> 64);
    }
    if (t[4] || geq4(t, f->p)) sub4(t, t, f->p);
    memcpy(a->l, t, 32);
}

void ofr_sqr(const ofield *f, ofr *a)
{
    ofr t = *a;
    ofr_mul(f, a, &t);
}

/* Field::pow with a single-limb exponent (all reference call sites pass &[u64; 1]). */
void ofr_pow(const ofield *f, ofr *out, const ofr *base, uint64_t e)
{
    ofr res = f->r, b = *base;
    int found = 0;
    for (int i = 63; i >= 0; i--) {
        if (found) ofr_sqr(f, &res);
        if ((e >> i) & 1) {
            found = 1;
            ofr_mul(f, &res, &b);
        }
    }
    *out = res;
}

static void pow_big(const ofield *f, ofr *out, const ofr *base, const uint64_t e[4])
{
    ofr res = f->r;
    for (int i = 255; i >= 0; i--) {
        ofr_sqr(f, &res);
        if ((e[i / 64] >> (i % 64)) & 1) ofr_mul(f, &res, base);
    }
    *out = res;
}

/* Field::inverse — canonical result, computed here as a^(p-2). */
int ofr_inverse(const ofield *f, ofr *out, const ofr *a)
{
    if (ofr_is_zero(a)) return -1;
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    sub4(e, f->p, two);
    pow_big(f, out, a, e);
    return 0;
}

int ofr_from_repr(const ofield *f, ofr *out, const uint64_t canon[4])
{
    if (geq4(canon, f->p)) return -1;
    memcpy(out->l, canon, 32);
    ofr_mul(f, out, &f->r2);
    return 0;
}

void ofr_into_repr(const ofield *f, uint64_t canon[4], const ofr *a)
{
    ofr one = {{1, 0, 0, 0}}, t = *a;
    ofr_mul(f, &t, &one);
    memcpy(canon, t.l, 32);
}

void ofr_from_u64(const ofield *f, ofr *out, uint64_t v)
{
    uint64_t c[4] = {v, 0, 0, 0};
    ofr_from_repr(f, out, c);   /* p > 2^64 for 4-limb fields */
}

int ofr_eq(const ofr *a, const ofr *b) { return memcmp(a, b, 32) == 0; }
int ofr_is_zero(const ofr *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

int ofield_init(ofield *f, const uint64_t modulus[4], uint64_t generator)
{
    memset(f, 0, sizeof(*f));
    memcpy(f->p, modulus, 32);
    if (!(modulus[0] & 1)) return -1;
    /* ff_derive picks limbs = min k with 2^(64k) >= 2p; this oracle handles the 4-limb case. */
    if (modulus[3] == 0 || (modulus[3] >> 63)) return -1;
    /* -p^{-1} mod 2^64 by Newton iteration */
    uint64_t inv = 1;
    for (int i = 0; i < 63; i++) {
        inv = inv * inv;
        inv = inv * modulus[0];
    }
    f->pinv = (uint64_t)(-(int64_t)inv);
    /* num_bits */
    int nb = 256;
    while (nb > 0 && !((modulus[(nb - 1) / 64] >> ((nb - 1) % 64)) & 1)) nb--;
    f->num_bits = (uint32_t)nb;
    f->capacity = (uint32_t)nb - 1;
    /* R = 2^256 mod p by 256 modular doublings of 1; R^2 by 256 more */
    uint64_t x[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; i++) {
        uint64_t c = add4(x, x, x);
        if (c || geq4(x, f->p)) sub4(x, x, f->p);
        if (i == 255) memcpy(f->r.l, x, 32);
    }
    memcpy(f->r2.l, x, 32);
    /* p - 1 = 2^S * t */
    uint64_t t[4], one[4] = {1, 0, 0, 0};
    sub4(t, f->p, one);
    uint32_t s = 0;
    while (!(t[0] & 1)) {
        for (int i = 0; i < 3; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 63);
        t[3] >>= 1;
        s++;
    }
    f->s = s;
    ofr_from_u64(f, &f->generator, generator);
    pow_big(f, &f->root_of_unity, &f->generator, t);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * threads: restates Worker::scope (src/fft/multicore.rs:60-85): chunk = elements / cpus
 * (1 if elements < cpus), one scoped thread per chunk.
 * ------------------------------------------------------------------------------------------ */

uint32_t o_num_cpus(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1u : (uint32_t)n;
}

static uint32_t log2_floor(size_t n)
{
    uint32_t r = 0;
    while (n > 1) { n >>= 1; r++; }
    return r;
}

typedef void (*chunk_fn)(void *ctx, size_t chunk_index, size_t start, size_t len);
typedef struct { chunk_fn fn; void *ctx; size_t idx, start, len; } chunk_job;

static void *chunk_tramp(void *p)
{
    chunk_job *j = (chunk_job *)p;
    j->fn(j->ctx, j->idx, j->start, j->len);
    return NULL;
}

static void worker_scope(uint32_t cpus, size_t elements, chunk_fn fn, void *ctx)
{
    if (cpus < 1) cpus = 1;
    if (elements == 0) return;
    size_t chunk = elements < cpus ? 1 : elements / cpus;
    size_t nchunks = (elements + chunk - 1) / chunk;
    if (nchunks == 1) { fn(ctx, 0, 0, elements); return; }
    pthread_t *th = (pthread_t *)malloc(nchunks * sizeof(pthread_t));
    chunk_job *jobs = (chunk_job *)malloc(nchunks * sizeof(chunk_job));
    for (size_t i = 0; i < nchunks; i++) {
        size_t start = i * chunk;
        size_t len = start + chunk > elements ? elements - start : chunk;
        jobs[i] = (chunk_job){fn, ctx, i, start, len};
        pthread_create(&th[i], NULL, chunk_tramp, &jobs[i]);
    }
    for (size_t i = 0; i < nchunks; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------------------------------
 * Domain::new_for_size — src/domains/mod.rs:21-44
 * ------------------------------------------------------------------------------------------ */
int odomain_new_for_size(const ofield *f, uint64_t size, odomain *out)
{
    uint64_t sz = 1;
    while (sz < size) sz <<= 1;          /* next_power_of_two (0 and 1 -> 1) */
    uint64_t power_of_two = 0, k = sz;
    while (k != 1) { k >>= 1; power_of_two++; }
    if (power_of_two > f->s) return -1;  /* SynthesisError::Error, :30-32 */
    ofr g = f->root_of_unity;
    for (uint64_t i = power_of_two; i < f->s; i++) ofr_sqr(f, &g);
    out->size = sz;
    out->power_of_two = power_of_two;
    out->generator = g;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * serial_fft — src/fft/fft.rs:21-66 (radix-2 DIT after bit reversal, running twiddle product)
 * ------------------------------------------------------------------------------------------ */
static uint32_t bitreverse(uint32_t n, uint32_t l)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

void o_serial_fft(const ofield *f, ofr *a, size_t n_, const ofr *omega, uint32_t log_n)
{
    uint32_t n = (uint32_t)n_;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rk = bitreverse(k, log_n);
        if (k < rk) { ofr t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        ofr w_m;
        ofr_pow(f, &w_m, omega, n / (2 * m));
        for (uint32_t k = 0; k < n; k += 2 * m) {
            ofr w = f->r;
            for (uint32_t j = 0; j < m; j++) {
                ofr t = a[k + j + m];
                ofr_mul(f, &t, &w);
                ofr tmp = a[k + j];
                ofr_sub(f, &tmp, &t);
                a[k + j + m] = tmp;
                ofr_add(f, &a[k + j], &t);
                ofr_mul(f, &w, &w_m);
            }
        }
        m *= 2;
    }
}

/* serial_fft_radix_4 — src/fft/radix4_fft/mod.rs:45-123 */
static uint64_t base4_digit_reverse(uint64_t n, uint64_t l)
{
    uint64_t r = 0;
    for (uint64_t i = 0; i < l; i++) { r = (r << 2) | (n & 3); n >>= 2; }
    return r;
}

void o_serial_fft_radix_4(const ofield *f, ofr *a, size_t n_, const ofr *omega, uint32_t log_n)
{
    uint64_t n = n_;
    uint64_t num_digits = log_n / 2;   /* reference asserts log_n % 2 == 0 (:50) */
    for (uint64_t k = 0; k < n; k++) {
        uint64_t rk = base4_digit_reverse(k, num_digits);
        if (k < rk) { ofr t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    ofr v;
    ofr_pow(f, &v, omega, n / 4);
    uint64_t m = 1;
    for (uint32_t s = 0; s < log_n / 2; s++) {
        ofr w_m;
        ofr_pow(f, &w_m, omega, n / (4 * m));
        for (uint64_t k = 0; k < n; k += 4 * m) {
            ofr w = f->r;
            for (uint64_t j = 0; j < m; j++) {
                ofr u = w;
                ofr x0 = a[k + j];
                ofr x1 = a[k + j + m];      ofr_mul(f, &x1, &w);
                ofr x2 = a[k + j + 2 * m];  ofr_mul(f, &u, &w); ofr_mul(f, &x2, &u);
                ofr x3 = a[k + j + 3 * m];  ofr_mul(f, &u, &w); ofr_mul(f, &x3, &u);

                ofr x0_plus_x2 = x0;  ofr_add(f, &x0_plus_x2, &x2);
                ofr x1_plus_x3 = x1;  ofr_add(f, &x1_plus_x3, &x3);
                a[k + j] = x0_plus_x2;          ofr_add(f, &a[k + j], &x1_plus_x3);
                a[k + j + 2 * m] = x0_plus_x2;  ofr_sub(f, &a[k + j + 2 * m], &x1_plus_x3);

                ofr x0_minus_x2 = x0; ofr_sub(f, &x0_minus_x2, &x2);
                ofr x1_minus_x3_by_w4 = x1;
                ofr_sub(f, &x1_minus_x3_by_w4, &x3);
                ofr_mul(f, &x1_minus_x3_by_w4, &v);
                a[k + j + m] = x0_minus_x2;      ofr_add(f, &a[k + j + m], &x1_minus_x3_by_w4);
                a[k + j + 3 * m] = x0_minus_x2;  ofr_sub(f, &a[k + j + 3 * m], &x1_minus_x3_by_w4);

                ofr_mul(f, &w, &w_m);
            }
        }
        m *= 4;
    }
}

/* parallel_fft — src/fft/fft.rs:68-124 */
typedef struct {
    const ofield *f; const ofr *a; ofr **tmp; const ofr *omega; ofr new_omega;
    uint32_t log_n, log_cpus, log_new_n;
} pfft_ctx;

static void pfft_shuffle(void *vctx, size_t j, size_t start, size_t len)
{
    (void)start; (void)len;
    pfft_ctx *c = (pfft_ctx *)vctx;
    const ofield *f = c->f;
    size_t num_cpus = (size_t)1 << c->log_cpus, new_n = (size_t)1 << c->log_new_n;
    ofr *tmp = c->tmp[j];
    ofr omega_j, omega_step;
    ofr_pow(f, &omega_j, c->omega, j);
    ofr_pow(f, &omega_step, c->omega, (uint64_t)j << c->log_new_n);
    ofr elt = f->r;
    for (size_t i = 0; i < new_n; i++) {
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = (i + (s << c->log_new_n)) % ((size_t)1 << c->log_n);
            ofr t = c->a[idx];
            ofr_mul(f, &t, &elt);
            ofr_add(f, &tmp[i], &t);
            ofr_mul(f, &elt, &omega_step);
        }
        ofr_mul(f, &elt, &omega_j);
    }
    o_serial_fft(f, tmp, new_n, &c->new_omega, c->log_new_n);
}

typedef struct { ofr *a; ofr **tmp; uint32_t log_cpus; } punshuf_ctx;

static void pfft_unshuffle(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    punshuf_ctx *c = (punshuf_ctx *)vctx;
    size_t mask = ((size_t)1 << c->log_cpus) - 1;
    for (size_t idx = start; idx < start + len; idx++)
        c->a[idx] = c->tmp[idx & mask][idx >> c->log_cpus];
}

void o_parallel_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                    uint32_t log_cpus)
{
    size_t num_cpus = (size_t)1 << log_cpus;
    uint32_t log_new_n = log_n - log_cpus;
    ofr **tmp = (ofr **)malloc(num_cpus * sizeof(ofr *));
    for (size_t j = 0; j < num_cpus; j++) tmp[j] = (ofr *)calloc((size_t)1 << log_new_n, sizeof(ofr));
    pfft_ctx c = {f, a, tmp, omega, {{0}}, log_n, log_cpus, log_new_n};
    ofr_pow(f, &c.new_omega, omega, num_cpus);
    /* worker.scope(0, ...) spawns one thread per sub-FFT (:87-108) */
    worker_scope((uint32_t)num_cpus, num_cpus, pfft_shuffle, &c);
    punshuf_ctx u = {a, tmp, log_cpus};
    worker_scope((uint32_t)num_cpus, n, pfft_unshuffle, &u);
    for (size_t j = 0; j < num_cpus; j++) free(tmp[j]);
    free(tmp);
}

/* parallel_fft_radix_4 — src/fft/radix4_fft/mod.rs:125-184: parallel_fft's split (the naive length-P DFT shuffle, :146-159,
 * the un-shuffle, :168-183) with serial_fft_radix_4 on the sub-sequences (:162); log_n and log_cpus even (:136-137). */
static void pfft4_shuffle(void *vctx, size_t j, size_t start, size_t len)
{
    (void)start; (void)len;
    pfft_ctx *c = (pfft_ctx *)vctx;
    const ofield *f = c->f;
    size_t num_cpus = (size_t)1 << c->log_cpus, new_n = (size_t)1 << c->log_new_n;
    ofr *tmp = c->tmp[j];
    ofr omega_j, omega_step;
    ofr_pow(f, &omega_j, c->omega, j);
    ofr_pow(f, &omega_step, c->omega, (uint64_t)j << c->log_new_n);
    ofr elt = f->r;
    for (size_t i = 0; i < new_n; i++) {
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = i + (s << c->log_new_n);                       /* :152 (no wrap: idx < n) */
            ofr t = c->a[idx];
            ofr_mul(f, &t, &elt);
            ofr_add(f, &tmp[i], &t);
            ofr_mul(f, &elt, &omega_step);
        }
        ofr_mul(f, &elt, &omega_j);
    }
    o_serial_fft_radix_4(f, tmp, new_n, &c->new_omega, c->log_new_n);   /* :162 */
}

int o_parallel_fft_radix_4(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t log_cpus)
{
    if (log_n < log_cpus || (log_n & 1) || (log_cpus & 1)) return -1;   /* asserts :133-137 */
    size_t num_cpus = (size_t)1 << log_cpus;
    uint32_t log_new_n = log_n - log_cpus;
    ofr **tmp = (ofr **)malloc(num_cpus * sizeof(ofr *));
    for (size_t j = 0; j < num_cpus; j++) tmp[j] = (ofr *)calloc((size_t)1 << log_new_n, sizeof(ofr));
    pfft_ctx c = {f, a, tmp, omega, {{0}}, log_n, log_cpus, log_new_n};
    ofr_pow(f, &c.new_omega, omega, num_cpus);                          /* :142 */
    worker_scope((uint32_t)num_cpus, num_cpus, pfft4_shuffle, &c);
    punshuf_ctx u = {a, tmp, log_cpus};
    worker_scope((uint32_t)num_cpus, n, pfft_unshuffle, &u);
    for (size_t j = 0; j < num_cpus; j++) free(tmp[j]);
    free(tmp);
    return 0;
}

/* best_fft of the radix-4 module — src/fft/radix4_fft/mod.rs:5-20: log_cpus rounded down to even (:8-12) */
int o_best_fft_radix_4(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t cpus)
{
    if (log_n & 1) return -1;                                           /* assert :7 */
    uint32_t log_cpus = log2_floor(cpus < 1 ? 1 : cpus);
    if (log_cpus & 1) log_cpus -= 1;
    if (log_n <= log_cpus) { o_serial_fft_radix_4(f, a, n, omega, log_n); return 0; }
    return o_parallel_fft_radix_4(f, a, n, omega, log_n, log_cpus);
}

/* best_fft — src/fft/fft.rs:5-19 with use_cpus_hint = None */
void o_best_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t cpus)
{
    uint32_t log_cpus = log2_floor(cpus < 1 ? 1 : cpus);
    if (log_cpus == 0 || log_n <= log_cpus) o_serial_fft(f, a, n, omega, log_n);
    else o_parallel_fft(f, a, n, omega, log_n, log_cpus);
}

/* serial_DIT_fft — src/fft/dit_fft/mod.rs:4-53.  (A decimation-in-frequency schedule despite its
 * name: butterflies on natural order, bit reversal at the end.)  `non_zero_entries_count` prunes every
 * block to its first min(block_len/2, count) butterflies (:29), which is exact when only the first
 * `count` inputs are non-zero — the reference's own zero-aware transform beside lde.rs. */
void o_serial_dit_fft(const ofield *f, ofr *a, size_t n_, const ofr *omega, uint32_t log_n,
                      size_t non_zero_entries_count)
{
    uint64_t n = (uint64_t)n_, m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        ofr w_m;
        ofr_pow(f, &w_m, omega, m);                                   /* :23 */
        uint64_t block_len = n / m;
        for (uint64_t block = 0; block < m; block++) {
            ofr w = f->r;
            uint64_t lim = block_len / 2 < (uint64_t)non_zero_entries_count ? block_len / 2
                                                                             : (uint64_t)non_zero_entries_count;
            for (uint64_t k = block * block_len; k < block * block_len + lim; k++) {   /* :29 */
                ofr t = a[k + block_len / 2];
                ofr tmp = a[k];
                ofr_sub(f, &tmp, &t);
                a[k + block_len / 2] = tmp;
                ofr_mul(f, &a[k + block_len / 2], &w);
                ofr_add(f, &a[k], &t);
                ofr_mul(f, &w, &w_m);
            }
        }
        m *= 2;
    }
    for (uint64_t k = 0; k < n; k++) {                                /* :44-51 */
        uint64_t rk = 0, x = k;
        for (uint32_t i = 0; i < log_n; i++) { rk = (rk << 1) | (x & 1); x >>= 1; }
        if (k < rk) { ofr t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
}

/* parallel_DIT_fft — src/fft/dit_fft/mod.rs:55-113: the shuffle of parallel_fft, pruned sub-FFTs */
typedef struct { pfft_ctx base; size_t nz; } pdit_ctx;
static void pdit_shuffle(void *vctx, size_t j, size_t start, size_t len)
{
    (void)start; (void)len;
    pdit_ctx *d = (pdit_ctx *)vctx;
    pfft_ctx *c = &d->base;
    const ofield *f = c->f;
    size_t num_cpus = (size_t)1 << c->log_cpus, new_n = (size_t)1 << c->log_new_n;
    ofr *tmp = c->tmp[j];
    ofr omega_j, omega_step;
    ofr_pow(f, &omega_j, c->omega, j);
    ofr_pow(f, &omega_step, c->omega, (uint64_t)j << c->log_new_n);
    ofr elt = f->r;
    for (size_t i = 0; i < new_n; i++) {
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = i + (s << c->log_new_n);                      /* :82 */
            ofr t = c->a[idx];
            ofr_mul(f, &t, &elt);
            ofr_add(f, &tmp[i], &t);
            ofr_mul(f, &elt, &omega_step);
        }
        ofr_mul(f, &elt, &omega_j);
    }
    o_serial_dit_fft(f, tmp, new_n, &c->new_omega, c->log_new_n, d->nz < new_n ? d->nz : new_n);   /* :92 */
}

void o_parallel_dit_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                        uint32_t log_cpus, size_t non_zero_entries_count)
{
    size_t num_cpus = (size_t)1 << log_cpus;
    uint32_t log_new_n = log_n - log_cpus;
    ofr **tmp = (ofr **)malloc(num_cpus * sizeof(ofr *));
    for (size_t j = 0; j < num_cpus; j++) tmp[j] = (ofr *)calloc((size_t)1 << log_new_n, sizeof(ofr));
    pdit_ctx d = {{f, a, tmp, omega, {{0}}, log_n, log_cpus, log_new_n}, non_zero_entries_count};
    ofr_pow(f, &d.base.new_omega, omega, num_cpus);
    worker_scope((uint32_t)num_cpus, num_cpus, pdit_shuffle, &d);
    punshuf_ctx u = {a, tmp, log_cpus};
    worker_scope((uint32_t)num_cpus, n, pfft_unshuffle, &u);
    for (size_t j = 0; j < num_cpus; j++) free(tmp[j]);
    free(tmp);
}

/* best_DIT_fft — src/fft/dit_fft/mod.rs:114-123 */
void o_best_dit_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t cpus,
                    size_t non_zero_entries_count)
{
    uint32_t log_cpus = log2_floor(cpus < 1 ? 1 : cpus);
    if (log_n <= log_cpus) o_serial_dit_fft(f, a, n, omega, log_n, non_zero_entries_count);
    else o_parallel_dit_fft(f, a, n, omega, log_n, log_cpus, non_zero_entries_count);
}

/* serial_lde — src/fft/lde.rs:15-126 (zero-aware FFT of a vector whose only non-zeros are the
 * first n/lde_factor coefficients) */
void o_serial_lde(const ofield *f, ofr *a, size_t n_, const ofr *omega, uint32_t log_n,
                  size_t lde_factor)
{
    uint32_t n = (uint32_t)n_;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rk = bitreverse(k, log_n);
        if (k < rk) { ofr t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    uint32_t m = 1, step = 0;
    for (uint32_t s = 0; s < log_n; s++) {
        ofr w_m;
        ofr_pow(f, &w_m, omega, n / (2 * m));
        int dense = (lde_factor >> step) <= 1;
        for (uint32_t k = 0; k < n; k += 2 * m) {
            ofr w = f->r;
            for (uint32_t j = 0; j < m; j++) {
                size_t odd = k + j + m, even = k + j;
                int odd_nz = 1, even_nz = 1;
                if (!dense) {
                    odd_nz = (odd & (lde_factor - 1)) < ((size_t)1 << step);
                    even_nz = (even & (lde_factor - 1)) < ((size_t)1 << step);
                }
                if (odd_nz && even_nz) {
                    ofr t = a[odd];
                    ofr_mul(f, &t, &w);
                    ofr tmp = a[even];
                    ofr_sub(f, &tmp, &t);
                    a[odd] = tmp;
                    ofr_add(f, &a[even], &t);
                } else if (!odd_nz && even_nz) {
                    a[odd] = a[even];
                } else if (odd_nz && !even_nz) {
                    ofr t = a[odd];
                    ofr_mul(f, &t, &w);
                    ofr tmp = t;
                    ofr_neg(f, &tmp);
                    a[odd] = tmp;
                    a[even] = t;
                }
                ofr_mul(f, &w, &w_m);
            }
        }
        step++;
        m *= 2;
    }
}

/* parallel_lde — src/fft/lde.rs:128-193: parallel_fft's split with the zero tail skipped in the shuffle (only
 * idx < a.len() / lde_factor contributes, :151-157) and, on the sub-sequences, serial_lde with the factor that is left
 * (lde_factor >> log_cpus, :163-168) or plain serial_fft when none is. */
typedef struct { pfft_ctx p; size_t non_trivial_len, lde_factor; } plde_ctx;

static void plde_shuffle(void *vctx, size_t j, size_t start, size_t len)
{
    (void)start; (void)len;
    plde_ctx *lc = (plde_ctx *)vctx;
    pfft_ctx *c = &lc->p;
    const ofield *f = c->f;
    size_t num_cpus = (size_t)1 << c->log_cpus, new_n = (size_t)1 << c->log_new_n;
    ofr *tmp = c->tmp[j];
    ofr omega_j, omega_step;
    ofr_pow(f, &omega_j, c->omega, j);
    ofr_pow(f, &omega_step, c->omega, (uint64_t)j << c->log_new_n);
    ofr elt = f->r;
    for (size_t i = 0; i < new_n; i++) {
        for (size_t s = 0; s < num_cpus; s++) {
            size_t idx = (i + (s << c->log_new_n)) % ((size_t)1 << c->log_n);   /* :150 */
            if (idx < lc->non_trivial_len) {                                    /* :151 */
                ofr t = c->a[idx];
                ofr_mul(f, &t, &elt);
                ofr_add(f, &tmp[i], &t);
            }
            ofr_mul(f, &elt, &omega_step);
        }
        ofr_mul(f, &elt, &omega_j);
    }
    size_t new_lde_factor = lc->lde_factor >> c->log_cpus;                      /* :163 */
    if (new_lde_factor <= 1) o_serial_fft(f, tmp, new_n, &c->new_omega, c->log_new_n);
    else o_serial_lde(f, tmp, new_n, &c->new_omega, c->log_new_n, new_lde_factor);
}

int o_parallel_lde(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t log_cpus, size_t lde_factor)
{
    if (log_n < log_cpus || lde_factor == 0) return -1;                         /* assert :137 */
    size_t num_cpus = (size_t)1 << log_cpus;
    uint32_t log_new_n = log_n - log_cpus;
    ofr **tmp = (ofr **)malloc(num_cpus * sizeof(ofr *));
    for (size_t j = 0; j < num_cpus; j++) tmp[j] = (ofr *)calloc((size_t)1 << log_new_n, sizeof(ofr));
    plde_ctx c = {{f, a, tmp, omega, {{0}}, log_n, log_cpus, log_new_n}, n / lde_factor, lde_factor};   /* :144 */
    ofr_pow(f, &c.p.new_omega, omega, num_cpus);
    worker_scope((uint32_t)num_cpus, num_cpus, plde_shuffle, &c);
    punshuf_ctx u = {a, tmp, log_cpus};
    worker_scope((uint32_t)num_cpus, n, pfft_unshuffle, &u);
    for (size_t j = 0; j < num_cpus; j++) free(tmp[j]);
    free(tmp);
    return 0;
}

/* best_lde — src/fft/lde.rs:4-13 */
int o_best_lde(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, size_t lde_factor, uint32_t cpus)
{
    uint32_t log_cpus = log2_floor(cpus < 1 ? 1 : cpus);
    if (log_n <= log_cpus) { o_serial_lde(f, a, n, omega, log_n, lde_factor); return 0; }
    return o_parallel_lde(f, a, n, omega, log_n, log_cpus, lde_factor);
}

/* distribute_powers — src/fft/mod.rs:110-123 */
typedef struct { const ofield *f; ofr *a; ofr g; size_t chunk; } dp_ctx;

static void dp_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    dp_ctx *c = (dp_ctx *)vctx;
    ofr u;
    ofr_pow(c->f, &u, &c->g, start);
    for (size_t i = start; i < start + len; i++) {
        ofr_mul(c->f, &c->a[i], &u);
        ofr_mul(c->f, &u, &c->g);
    }
}

void o_distribute_powers(const ofield *f, ofr *a, size_t n, const ofr *g, uint32_t cpus)
{
    dp_ctx c = {f, a, *g, 0};
    worker_scope(cpus, n, dp_chunk, &c);
}

/* Polynomial<F, Coefficients>::evaluate_at_domain_for_degree_one / coset_evaluate_at_domain_for_degree_one
 * (src/polynomials/mod.rs:229-258, :260-290): for q(x) = c + alpha x,  out[i] = alpha * u_i + c  with
 * u_i = g^i (g = generator of the size-n domain), resp. u_i = multiplicative_generator * g^i.  The reference
 * splits the range into Worker chunks that each restart u at g^(i * chunk); the values do not depend on it. */
int o_poly_degree_one_on_domain(const ofield *f, ofr *out, size_t n, const ofr *alpha, const ofr *c, int coset)
{
    odomain d;
    if (odomain_new_for_size(f, n, &d) || d.size != n) return -1;
    ofr u = f->r;                                   /* :245 g.pow(0) */
    if (coset) ofr_mul(f, &u, &f->generator);       /* :277 */
    for (size_t i = 0; i < n; i++) {
        ofr tmp = *alpha;                           /* :247-250 */
        ofr_mul(f, &tmp, &u);
        ofr_add(f, &tmp, c);
        out[i] = tmp;
        ofr_mul(f, &u, &d.generator);               /* :251 */
    }
    return 0;
}

void o_naive_dft(const ofield *f, const ofr *in, ofr *out, size_t n, const ofr *omega)
{
    for (size_t k = 0; k < n; k++) {
        ofr wk, w = f->r, acc = {{0, 0, 0, 0}};
        ofr_pow(f, &wk, omega, k);
        for (size_t i = 0; i < n; i++) {
            ofr t = in[i];
            ofr_mul(f, &t, &w);
            ofr_add(f, &acc, &t);
            ofr_mul(f, &w, &wk);
        }
        out[k] = acc;
    }
}

/* ------------------------------------------------------------------------------------------
 * Polynomial<F, Coefficients|Values> — src/polynomials/mod.rs
 * ------------------------------------------------------------------------------------------ */
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

int o_poly_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus)
{
    odomain d;
    if (!is_pow2(n) || odomain_new_for_size(f, n, &d)) return -1;
    o_best_fft(f, a, n, &d.generator, (uint32_t)d.power_of_two, cpus);
    return 0;
}

int o_poly_coset_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus)
{
    o_distribute_powers(f, a, n, &f->generator, cpus);   /* :626-631 */
    return o_poly_fft(f, a, n, cpus);
}

typedef struct { const ofield *f; ofr *a; ofr s; } scale_ctx;
static void scale_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    scale_ctx *c = (scale_ctx *)vctx;
    for (size_t i = start; i < start + len; i++) ofr_mul(c->f, &c->a[i], &c->s);
}

/* coset_fft_for_generator (src/polynomials/mod.rs:633-638) */
int o_poly_coset_fft_for_generator(const ofield *f, ofr *a, size_t n, const ofr *gen, uint32_t cpus)
{
    if (!is_pow2(n)) return -1;
    o_distribute_powers(f, a, n, gen, cpus);
    return o_poly_fft(f, a, n, cpus);
}

int o_poly_ifft(const ofield *f, ofr *a, size_t n, uint32_t cpus)
{
    odomain d;
    if (!is_pow2(n) || odomain_new_for_size(f, n, &d)) return -1;
    ofr omegainv, m, minv;
    ofr_inverse(f, &omegainv, &d.generator);
    ofr_from_u64(f, &m, n);
    ofr_inverse(f, &minv, &m);
    o_best_fft(f, a, n, &omegainv, (uint32_t)d.power_of_two, cpus);   /* :775 */
    scale_ctx c = {f, a, minv};
    worker_scope(cpus, n, scale_chunk, &c);                            /* :777-787 */
    return 0;
}

int o_poly_icoset_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus)
{
    ofr geninv;
    ofr_inverse(f, &geninv, &f->generator);
    if (o_poly_ifft(f, a, n, cpus)) return -1;
    o_distribute_powers(f, a, n, &geninv, cpus);                       /* :800-807 */
    return 0;
}

/* icoset_fft_for_generator (src/polynomials/mod.rs:809-815): the caller passes the inverse generator */
int o_poly_icoset_fft_for_generator(const ofield *f, ofr *a, size_t n, const ofr *geninv, uint32_t cpus)
{
    if (o_poly_ifft(f, a, n, cpus)) return -1;
    o_distribute_powers(f, a, n, geninv, cpus);
    return 0;
}

typedef struct {
    const ofield *f; const ofr *coeffs; size_t n; size_t factor; int coset;
    ofr coset_omega, this_omega; uint32_t log_n; ofr **results; size_t chunk;
} lde_ctx;

static void lde_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    lde_ctx *c = (lde_ctx *)vctx;
    const ofield *f = c->f;
    ofr coset_generator;
    ofr_pow(f, &coset_generator, &c->coset_omega, start);       /* :451 (i == start for chunk 1) */
    if (c->coset) ofr_mul(f, &coset_generator, &f->generator);  /* :577 */
    for (size_t r = start; r < start + len; r++) {
        ofr *cc = (ofr *)malloc(c->n * sizeof(ofr));
        memcpy(cc, c->coeffs, c->n * sizeof(ofr));
        o_distribute_powers(f, cc, c->n, &coset_generator, 1);
        o_serial_fft(f, cc, c->n, &c->this_omega, c->log_n);    /* num_cpus_hint = Some(1) */
        c->results[r] = cc;
        ofr_mul(f, &coset_generator, &c->coset_omega);
    }
}

typedef struct { ofr *out; ofr **results; size_t factor; } gather_ctx;
static void gather_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    gather_ctx *c = (gather_ctx *)vctx;
    for (size_t idx = start; idx < start + len; idx++)
        c->out[idx] = c->results[idx % c->factor][idx / c->factor];    /* :466-479 */
}

int o_poly_lde(const ofield *f, const ofr *coeffs, size_t n, size_t factor, int coset, ofr *out,
               uint32_t cpus)
{
    if (!is_pow2(n) || !is_pow2(factor)) return -1;
    odomain dn, dbig;
    if (odomain_new_for_size(f, n, &dn)) return -1;
    if (factor == 1) {   /* :419-421 / :545-547 */
        memcpy(out, coeffs, n * sizeof(ofr));
        return coset ? o_poly_coset_fft(f, out, n, cpus) : o_poly_fft(f, out, n, cpus);
    }
    if (odomain_new_for_size(f, n * factor, &dbig)) return -1;
    lde_ctx c;
    memset(&c, 0, sizeof(c));
    c.f = f; c.coeffs = coeffs; c.n = n; c.factor = factor; c.coset = coset;
    c.coset_omega = dbig.generator; c.this_omega = dn.generator;
    c.log_n = (uint32_t)dn.power_of_two;
    c.results = (ofr **)calloc(factor, sizeof(ofr *));
    /* NOTE the reference computes coset_omega^i with i = chunk index, which equals the first coset
     * index of the chunk only when chunk == 1 (factor <= cpus) — the only regime in which its
     * output is the LDE.  The oracle uses the chunk's first coset index (mathematically intended). */
    worker_scope(cpus, factor, lde_chunk, &c);
    gather_ctx g = {out, c.results, factor};
    worker_scope(cpus, n * factor, gather_chunk, &g);
    for (size_t i = 0; i < factor; i++) free(c.results[i]);
    free(c.results);
    return 0;
}

/* add_assign / sub_assign / mul_assign — src/polynomials/mod.rs:817-887 */
void o_poly_binary(const ofield *f, ofr *a, const ofr *b, size_t n, int op)
{
    for (size_t i = 0; i < n; i++) {
        if (op == 0) ofr_add(f, &a[i], &b[i]);
        else if (op == 1) ofr_sub(f, &a[i], &b[i]);
        else ofr_mul(f, &a[i], &b[i]);
    }
}

/* add_assign_scaled — :657-671 */
void o_poly_add_scaled(const ofield *f, ofr *a, const ofr *b, size_t n, const ofr *scaling)
{
    for (size_t i = 0; i < n; i++) {
        ofr t = b[i];
        ofr_mul(f, &t, scaling);
        ofr_add(f, &a[i], &t);
    }
}

/* negate :73-83, square :758-771, pow :744-756, scale :60-72, add_constant / sub_constant */
void o_poly_unary(const ofield *f, ofr *a, size_t n, int op, const ofr *c, uint64_t e)
{
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case 0: ofr_neg(f, &a[i]); break;
        case 1: ofr_sqr(f, &a[i]); break;
        case 2: { ofr t; ofr_pow(f, &t, &a[i], e); a[i] = t; break; }
        case 3: ofr_mul(f, &a[i], c); break;
        case 4: ofr_add(f, &a[i], c); break;
        default: ofr_sub(f, &a[i], c); break;
        }
    }
}

/* batch_inversion — :889-954 (Montgomery's trick over one chunk) */
int o_poly_batch_inversion(const ofield *f, ofr *a, size_t n)
{
    if (n == 0) return 0;
    ofr *grand = (ofr *)malloc(n * sizeof(ofr));
    ofr s = f->r;
    for (size_t i = 0; i < n; i++) { ofr_mul(f, &s, &a[i]); grand[i] = s; }
    ofr inv;
    if (ofr_inverse(f, &inv, &s)) { free(grand); return -1; }   /* SynthesisError::Error */
    for (size_t i = n; i-- > 0;) {
        ofr tmp = a[i];
        a[i] = i ? grand[i - 1] : f->r;
        ofr_mul(f, &a[i], &inv);
        ofr_mul(f, &inv, &tmp);
    }
    free(grand);
    return 0;
}

void o_poly_evaluate_at(const ofield *f, const ofr *coeffs, size_t n, const ofr *g, ofr *out)
{
    ofr x = f->r, acc = {{0, 0, 0, 0}};
    for (size_t i = 0; i < n; i++) {
        ofr v = x;
        ofr_mul(f, &v, &coeffs[i]);
        ofr_add(f, &acc, &v);
        ofr_mul(f, &x, g);
    }
    *out = acc;
}

/* The same sum as the reference schedules it (src/polynomials/mod.rs:685-711): one partial sum per
 * Worker chunk, each starting from g^(i*chunk), added up in chunk order.  Field addition is exact,
 * so the result equals o_poly_evaluate_at's for every `cpus`. */
typedef struct { const ofield *f; const ofr *coeffs; const ofr *g; ofr *sub; size_t chunk; } eval_ctx;
static void eval_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    eval_ctx *c = (eval_ctx *)vctx;
    const ofield *f = c->f;
    ofr x, acc = {{0, 0, 0, 0}};
    ofr_pow(f, &x, c->g, (uint64_t)start);      /* g.pow([(i*chunk) as u64]) */
    for (size_t i = start; i < start + len; i++) {
        ofr v = x;
        ofr_mul(f, &v, &c->coeffs[i]);
        ofr_add(f, &acc, &v);
        ofr_mul(f, &x, c->g);
    }
    c->sub[ci] = acc;
}

void o_poly_evaluate_at_mt(const ofield *f, const ofr *coeffs, size_t n, const ofr *g, ofr *out, uint32_t cpus)
{
    if (cpus < 1) cpus = 1;
    size_t chunk = n < cpus ? 1 : n / cpus;
    size_t nchunks = n ? (n + chunk - 1) / chunk : 0;
    ofr *sub = (ofr *)calloc(nchunks ? nchunks : 1, sizeof(ofr));
    eval_ctx c = {f, coeffs, g, sub, chunk};
    worker_scope(cpus, n, eval_chunk, &c);
    ofr acc = {{0, 0, 0, 0}};
    for (size_t i = 0; i < nchunks; i++) ofr_add(f, &acc, &sub[i]);
    free(sub);
    *out = acc;
}

/* ------------------------------------------------------------------------------------------
 * BLAKE2s (RFC 7693) with key + personalisation, as blake2s_simd::Params builds it
 * (src/iop/blake2s_trivial_iop.rs:8-16: hash_length 32, key "Squeamish Ossifrage",
 * personal "Shaftoe").
 * ------------------------------------------------------------------------------------------ */
static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                   0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last)
{
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) |
               ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y)                                  \
    do {                                                     \
        v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 12); \
        v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);  \
        v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 7);  \
    } while (0)
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void o_blake2s(uint8_t out[32], const uint8_t *key, size_t keylen, const uint8_t *personal,
               size_t personal_len, const uint8_t *data, size_t len)
{
    uint8_t param[32];
    memset(param, 0, 32);
    param[0] = 32;               /* digest length */
    param[1] = (uint8_t)keylen;  /* key length */
    param[2] = 1;                /* fanout */
    param[3] = 1;                /* depth */
    if (personal_len) memcpy(param + 24, personal, personal_len > 8 ? 8 : personal_len);   /* personal may be NULL with length 0 */
    uint32_t h[8];
    for (int i = 0; i < 8; i++) {
        uint32_t pw = (uint32_t)param[4 * i] | ((uint32_t)param[4 * i + 1] << 8) |
                      ((uint32_t)param[4 * i + 2] << 16) | ((uint32_t)param[4 * i + 3] << 24);
        h[i] = B2S_IV[i] ^ pw;
    }
    uint8_t block[64];
    uint64_t t = 0;
    if (keylen) {   /* the key is a full zero-padded first block */
        memset(block, 0, 64);
        memcpy(block, key, keylen);
        t = 64;
        if (len == 0) { b2s_compress(h, block, t, 1); goto done; }
        b2s_compress(h, block, t, 0);
    }
    while (len > 64) {
        t += 64;
        b2s_compress(h, data, t, 0);
        data += 64;
        len -= 64;
    }
    memset(block, 0, 64);
    if (len) memcpy(block, data, len);
    t += len;
    b2s_compress(h, block, t, 1);
done:
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)h[i];
        out[4 * i + 1] = (uint8_t)(h[i] >> 8);
        out[4 * i + 2] = (uint8_t)(h[i] >> 16);
        out[4 * i + 3] = (uint8_t)(h[i] >> 24);
    }
}

static const uint8_t IOP_KEY[] = "Squeamish Ossifrage";   /* 19 bytes */
static const uint8_t IOP_PERSONAL[] = "Shaftoe";          /* 7 bytes  */

/* encode_leaf :36-42 = raw Montgomery limbs, little-endian; hash_leaf :81-91 */
void o_hash_leaf(uint8_t out[32], const ofr *leaf)
{
    uint8_t enc[32];
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) enc[8 * i + b] = (uint8_t)(leaf->l[i] >> (8 * b));
    o_blake2s(out, IOP_KEY, 19, IOP_PERSONAL, 7, enc, 32);
}

/* hash_node :93-104 */
void o_hash_node(uint8_t out[32], const uint8_t l[32], const uint8_t r[32])
{
    uint8_t buf[64];
    memcpy(buf, l, 32);
    memcpy(buf + 32, r, 32);
    o_blake2s(out, IOP_KEY, 19, IOP_PERSONAL, 7, buf, 64);
}

/* Blake2sIopTree::create :131-219 — heap layout: root nodes[1], level l at [2^l, 2^(l+1)) */
typedef struct { const ofr *leafs; uint8_t *lh; } leafh_ctx;
static void leafh_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    leafh_ctx *c = (leafh_ctx *)vctx;
    for (size_t i = start; i < start + len; i++) o_hash_leaf(c->lh + 32 * i, &c->leafs[i]);
}
typedef struct { const uint8_t *in; uint8_t *out; } nodeh_ctx;
static void nodeh_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    nodeh_ctx *c = (nodeh_ctx *)vctx;
    for (size_t i = start; i < start + len; i++)
        o_hash_node(c->out + 32 * i, c->in + 64 * i, c->in + 64 * i + 32);
}

int o_iop_create(const ofr *leafs, size_t n, uint8_t *nodes, uint32_t cpus)
{
    if (!is_pow2(n) || n < 2) return -1;   /* :137; n == 1 underflows num_levels-1 in the reference */
    uint8_t *lh = (uint8_t *)malloc(n * 32);
    leafh_ctx lc = {leafs, lh};
    worker_scope(cpus, n, leafh_chunk, &lc);
    memset(nodes, 0, 32);   /* nodes[0] unused, stays zero (:145) */
    nodeh_ctx nc = {lh, nodes + 32 * (n / 2)};
    worker_scope(cpus, n / 2, nodeh_chunk, &nc);
    for (size_t width = n / 4; width >= 1; width /= 2) {
        nodeh_ctx c = {nodes + 32 * (2 * width), nodes + 32 * width};
        worker_scope(cpus, width, nodeh_chunk, &c);
    }
    free(lh);
    return 0;
}

/* interpret_hash :48-60 */
void o_interpret_hash(const ofield *f, const uint8_t h[32], ofr *out)
{
    uint64_t repr[4];
    for (int i = 0; i < 4; i++) {   /* read_be: first 8 bytes are the most significant limb */
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w = (w << 8) | h[8 * i + b];
        repr[3 - i] = w;
    }
    uint32_t shave = 256 - f->capacity;
    repr[3] &= 0xffffffffffffffffull >> (shave % 64);
    ofr_from_repr(f, out, repr);
}

/* get_path :251-279 */
size_t o_iop_path(const uint8_t *nodes, const ofr *leafs, size_t n, size_t tree_index, uint8_t *path)
{
    size_t cnt = 0;
    o_hash_leaf(path, &leafs[tree_index ^ 1]);
    cnt++;
    size_t idx = tree_index >> 1;
    for (size_t width = n / 2; width >= 2; width /= 2) {   /* log2_floor(n/2) iterations */
        memcpy(path + 32 * cnt, nodes + 32 * (width + (idx ^ 1)), 32);
        cnt++;
        idx >>= 1;
    }
    return cnt;
}

/* verify :236-249 */
int o_iop_verify(const uint8_t root[32], const ofr *leaf, const uint8_t *path, size_t path_len,
                 size_t tree_index)
{
    uint8_t h[32], t[32];
    o_hash_leaf(h, leaf);
    size_t idx = tree_index;
    for (size_t i = 0; i < path_len; i++) {
        if ((idx & 1) == 0) o_hash_node(t, h, path + 32 * i);
        else o_hash_node(t, path + 32 * i, h);
        memcpy(h, t, 32);
        idx >>= 1;
    }
    return memcmp(h, root, 32) == 0;
}

/* ------------------------------------------------------------------------------------------
 * COSET2 combiner (see hodor_oracle.h): leaf k = value[k] || value[k + n/2], one hash per coset
 * ------------------------------------------------------------------------------------------ */
void o_hash_leaf_pair(uint8_t out[32], const ofr *lo, const ofr *hi)
{
    uint8_t enc[64];
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) {
            enc[8 * i + b] = (uint8_t)(lo->l[i] >> (8 * b));        /* encode_leaf :36-42, twice */
            enc[32 + 8 * i + b] = (uint8_t)(hi->l[i] >> (8 * b));
        }
    /* a personalisation of its own ("Shaftoe2"): a COSET2 leaf is 64 bytes like the input of a node hash, and the two
     * must never collide (an interior node's children opened as a "value pair") */
    o_blake2s(out, IOP_KEY, 19, (const uint8_t *)"Shaftoe2", 8, enc, 64);
}

typedef struct { const ofr *values; size_t half; uint8_t *lh; } pairh_ctx;
static void pairh_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    pairh_ctx *c = (pairh_ctx *)vctx;
    for (size_t k = start; k < start + len; k++) o_hash_leaf_pair(c->lh + 32 * k, &c->values[k], &c->values[k + c->half]);
}

/* Blake2sIopTree::create :131-219 over the n/2 combined leaves */
int o_iop_create_coset2(const ofr *values, size_t n, uint8_t *nodes, uint32_t cpus)
{
    if (!is_pow2(n) || n < 4) return -1;
    size_t leaves = n / 2;
    uint8_t *lh = (uint8_t *)malloc(leaves * 32);
    pairh_ctx pc = {values, leaves, lh};
    worker_scope(cpus, leaves, pairh_chunk, &pc);
    memset(nodes, 0, 32);
    nodeh_ctx nc = {lh, nodes + 32 * (leaves / 2)};
    worker_scope(cpus, leaves / 2, nodeh_chunk, &nc);
    for (size_t width = leaves / 4; width >= 1; width /= 2) {
        nodeh_ctx c = {nodes + 32 * (2 * width), nodes + 32 * width};
        worker_scope(cpus, width, nodeh_chunk, &c);
    }
    free(lh);
    return 0;
}

/* get_path :251-279 for the leaf that holds natural_index (k = natural_index mod n/2) */
size_t o_iop_path_coset2(const uint8_t *nodes, const ofr *values, size_t n, size_t natural_index, uint8_t *path)
{
    size_t leaves = n / 2, k = natural_index % leaves, cnt = 0;
    o_hash_leaf_pair(path, &values[k ^ 1], &values[(k ^ 1) + leaves]);
    cnt++;
    size_t idx = k >> 1;
    for (size_t width = leaves / 2; width >= 2; width /= 2) {
        memcpy(path + 32 * cnt, nodes + 32 * (width + (idx ^ 1)), 32);
        cnt++;
        idx >>= 1;
    }
    return cnt;
}

/* verify :236-249 with the combined leaf */
int o_iop_verify_coset2(const uint8_t root[32], const ofr *lo, const ofr *hi, const uint8_t *path, size_t path_len,
                        size_t leaf_index)
{
    uint8_t h[32], t[32];
    o_hash_leaf_pair(h, lo, hi);
    size_t idx = leaf_index;
    for (size_t i = 0; i < path_len; i++) {
        if ((idx & 1) == 0) o_hash_node(t, h, path + 32 * i);
        else o_hash_node(t, path + 32 * i, h);
        memcpy(h, t, 32);
        idx >>= 1;
    }
    return memcmp(h, root, 32) == 0;
}

/* ------------------------------------------------------------------------------------------
 * NaiveFriIop::proof_from_lde_by_values — src/fri/fri_on_values.rs:11-159
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const ofield *f; const ofr *values; ofr *next; const ofr *omegas_inv; size_t next_size, stride;
    ofr challenge, two_inv;
} fold_ctx;

static void fold_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    fold_ctx *c = (fold_ctx *)vctx;
    const ofield *f = c->f;
    for (size_t idx = start; idx < start + len; idx++) {
        ofr a = c->values[idx], b = c->values[idx + c->next_size];
        ofr even = a;  ofr_add(f, &even, &b);
        ofr odd = a;   ofr_sub(f, &odd, &b);
        ofr_mul(f, &odd, &c->omegas_inv[idx * c->stride]);
        ofr tmp = odd;
        ofr_mul(f, &tmp, &c->challenge);
        ofr_add(f, &tmp, &even);
        ofr_mul(f, &tmp, &c->two_inv);
        c->next[idx] = tmp;
    }
}

typedef struct { const ofield *f; ofr *t; ofr w; } powtab_ctx;
static void powtab_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    powtab_ctx *c = (powtab_ctx *)vctx;
    ofr u;
    ofr_pow(c->f, &u, &c->w, start);
    for (size_t i = start; i < start + len; i++) { c->t[i] = u; ofr_mul(c->f, &u, &c->w); }
}

int o_fri_commit(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                 size_t out_deg_plus_one, uint32_t cpus, ofri_proto **outp)
{
    return o_fri_commit_combined(f, lde_values, n, lde_factor, out_deg_plus_one, O_COMBINER_TRIVIAL, cpus, outp);
}

static int tree_create(int combiner, const ofr *values, size_t n, uint8_t *nodes, uint32_t cpus)
{
    return combiner == O_COMBINER_COSET2 ? o_iop_create_coset2(values, n, nodes, cpus)
                                         : o_iop_create(values, n, nodes, cpus);
}

int o_fri_commit_combined(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                          size_t out_deg_plus_one, int combiner, uint32_t cpus, ofri_proto **outp)
{
    if (combiner != O_COMBINER_TRIVIAL && combiner != O_COMBINER_COSET2) return -1;
    const size_t min_tree = combiner == O_COMBINER_COSET2 ? 4 : 2;
    if (!is_pow2(n) || !is_pow2(lde_factor) || !is_pow2(out_deg_plus_one)) return -1;
    odomain d;
    if (odomain_new_for_size(f, n, &d)) return -1;
    size_t initial_degree_plus_one = n / lde_factor;
    if (initial_degree_plus_one < 2 * out_deg_plus_one) return -1;   /* num_steps >= 1 else roots.pop() panics (:124) */
    size_t num_steps = log2_floor(initial_degree_plus_one / out_deg_plus_one);

    ofri_proto *p = (ofri_proto *)calloc(1, sizeof(ofri_proto));
    p->num_steps = num_steps;
    p->initial_degree_plus_one = initial_degree_plus_one;
    p->output_coeffs_at_degree_plus_one = out_deg_plus_one;
    p->lde_factor = lde_factor;
    p->l0_nodes = (uint8_t *)malloc(n * 32);
    if (tree_create(combiner, lde_values, n, p->l0_nodes, cpus)) { free(p->l0_nodes); free(p); return -1; }   /* :17 */

    ofr two, two_inv, omega_inv;
    ofr_from_u64(f, &two, 2);
    ofr_inverse(f, &two_inv, &two);
    ofr_inverse(f, &omega_inv, &d.generator);
    ofr *omegas_inv = (ofr *)malloc((n / 2) * sizeof(ofr));              /* :24-40 */
    powtab_ctx pc = {f, omegas_inv, omega_inv};
    worker_scope(cpus, n / 2, powtab_chunk, &pc);

    p->inter_nodes = (uint8_t **)calloc(num_steps, sizeof(uint8_t *));
    p->inter_values = (ofr **)calloc(num_steps, sizeof(ofr *));
    p->inter_sizes = (size_t *)calloc(num_steps, sizeof(size_t));
    p->challenges = (ofr *)calloc(num_steps + 1, sizeof(ofr));
    uint8_t (*roots)[32] = (uint8_t (*)[32])calloc(num_steps, 32);

    ofr challenge;
    o_interpret_hash(f, p->l0_nodes + 32, &challenge);                   /* :51 */
    p->challenges[0] = challenge;
    size_t next_size = n / 2;
    const ofr *values = lde_values;
    for (size_t i = 0; i < num_steps; i++) {                             /* :61 */
        ofr *next = (ofr *)malloc(next_size * sizeof(ofr));
        fold_ctx fc = {f, values, next, omegas_inv, next_size, (size_t)1 << i, challenge, two_inv};
        worker_scope(cpus, next_size, fold_chunk, &fc);                  /* :70-104 */
        uint8_t *nodes = (uint8_t *)malloc(next_size * 32);
        if (next_size >= min_tree) tree_create(combiner, next, next_size, nodes, cpus);  /* :106 */
        else { free(nodes); free(next); o_fri_free(p); free(roots); free(omegas_inv); return -1; }
        memcpy(roots[i], nodes + 32, 32);
        o_interpret_hash(f, nodes + 32, &challenge);
        p->challenges[i + 1] = challenge;
        p->inter_nodes[i] = nodes;
        p->inter_values[i] = next;
        p->inter_sizes[i] = next_size;
        values = next;
        next_size >>= 1;
    }
    /* challenges.pop() :121 — the extra one is simply not reported */
    memcpy(p->final_root, roots[num_steps - 1], 32);                     /* :124 */
    size_t fin_n = p->inter_sizes[num_steps - 1];
    ofr *fin = (ofr *)malloc(fin_n * sizeof(ofr));
    memcpy(fin, values, fin_n * sizeof(ofr));
    o_poly_ifft(f, fin, fin_n, cpus);                                    /* :130-131 */
    p->final_coeffs = (ofr *)malloc(out_deg_plus_one * sizeof(ofr));
    memcpy(p->final_coeffs, fin, out_deg_plus_one * sizeof(ofr));        /* truncate :145 */
    free(fin);
    free(roots);
    free(omegas_inv);
    *outp = p;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * NaiveFriIop::proof_from_lde_through_coefficients — src/fri/mod.rs:156-248.  The reference's cross-check of
 * the commit phase: interpolate once (ifft :171, truncate to the degree bound :173), fold the COEFFICIENTS
 * a_2i + beta a_(2i+1) (:194-203) and re-extend every round with Polynomial::lde (:210-211).  Its own test asserts
 * the prototype equal to proof_from_lde_by_values field for field (:338-343).  `combiner` as for
 * o_fri_commit_combined (the reference has TRIVIAL only).
 * ------------------------------------------------------------------------------------------ */
typedef struct { const ofield *f; const ofr *old; ofr *next; ofr challenge; } cfold_ctx;
static void cfold_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    cfold_ctx *c = (cfold_ctx *)vctx;
    for (size_t i = start; i < start + len; i++) {
        ofr tmp = c->old[2 * i + 1];                 /* :196-201: tmp = old[1] * challenge + old[0] */
        ofr_mul(c->f, &tmp, &c->challenge);
        ofr_add(c->f, &tmp, &c->old[2 * i]);
        c->next[i] = tmp;
    }
}

int o_fri_commit_through_coefficients(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                                      size_t out_deg_plus_one, int combiner, uint32_t cpus, ofri_proto **outp)
{
    if (combiner != O_COMBINER_TRIVIAL && combiner != O_COMBINER_COSET2) return -1;
    const size_t min_tree = combiner == O_COMBINER_COSET2 ? 4 : 2;
    if (!is_pow2(n) || !is_pow2(lde_factor) || !is_pow2(out_deg_plus_one)) return -1;   /* asserts :165-166 */
    odomain d;
    if (odomain_new_for_size(f, n, &d)) return -1;
    size_t initial_degree_plus_one = n / lde_factor;                                  /* :168 */
    if (initial_degree_plus_one < 2 * out_deg_plus_one) return -1;   /* num_steps >= 1, else roots.pop() panics (:226) */
    size_t num_steps = log2_floor(initial_degree_plus_one / out_deg_plus_one);        /* :169 */
    if ((n >> num_steps) < min_tree) return -1;

    ofri_proto *p = (ofri_proto *)calloc(1, sizeof(ofri_proto));
    p->num_steps = num_steps;
    p->initial_degree_plus_one = initial_degree_plus_one;
    p->output_coeffs_at_degree_plus_one = out_deg_plus_one;
    p->lde_factor = lde_factor;
    p->l0_nodes = (uint8_t *)malloc(n * 32);
    tree_create(combiner, lde_values, n, p->l0_nodes, cpus);                          /* :162 */

    ofr *all = (ofr *)malloc(n * sizeof(ofr));
    memcpy(all, lde_values, n * sizeof(ofr));
    o_poly_ifft(f, all, n, cpus);                                                     /* :171 */
    size_t len = initial_degree_plus_one;                                             /* truncate :173 */
    ofr *coeffs = (ofr *)malloc(len * sizeof(ofr));
    memcpy(coeffs, all, len * sizeof(ofr));
    free(all);

    p->inter_nodes = (uint8_t **)calloc(num_steps, sizeof(uint8_t *));
    p->inter_values = (ofr **)calloc(num_steps, sizeof(ofr *));
    p->inter_sizes = (size_t *)calloc(num_steps, sizeof(size_t));
    p->challenges = (ofr *)calloc(num_steps + 1, sizeof(ofr));
    ofr challenge;
    o_interpret_hash(f, p->l0_nodes + 32, &challenge);                                /* :178-179 */
    p->challenges[0] = challenge;
    size_t next_len = len / 2;                                                        /* :180 */
    for (size_t i = 0; i < num_steps; i++) {                                          /* :185 */
        ofr *next = (ofr *)malloc(next_len * sizeof(ofr));
        cfold_ctx cc = {f, coeffs, next, challenge};
        worker_scope(cpus, next_len, cfold_chunk, &cc);                               /* :190-205 */
        size_t vsize = next_len * lde_factor;
        ofr *values = (ofr *)malloc(vsize * sizeof(ofr));
        o_poly_lde(f, next, next_len, lde_factor, 0, values, cpus);                   /* :208-209 from_coeffs + lde */
        uint8_t *nodes = (uint8_t *)malloc(vsize * 32);
        tree_create(combiner, values, vsize, nodes, cpus);                            /* :210 */
        o_interpret_hash(f, nodes + 32, &challenge);                                  /* :213 */
        p->challenges[i + 1] = challenge;
        p->inter_nodes[i] = nodes;
        p->inter_values[i] = values;
        p->inter_sizes[i] = vsize;
        free(coeffs);
        coeffs = next;                                                                /* :221 */
        next_len >>= 1;
    }
    /* challenges.pop() :224; final_root = roots.pop() :226 */
    memcpy(p->final_root, p->inter_nodes[num_steps - 1] + 32, 32);
    p->final_coeffs = coeffs;                                                         /* :232-234: len == out_deg_plus_one */
    *outp = p;
    return 0;
}

void o_fri_free(ofri_proto *p)
{
    if (!p) return;
    free(p->l0_nodes);
    for (size_t i = 0; i < p->num_steps; i++) {
        if (p->inter_nodes) free(p->inter_nodes[i]);
        if (p->inter_values) free(p->inter_values[i]);
    }
    free(p->inter_nodes);
    free(p->inter_values);
    free(p->inter_sizes);
    free(p->challenges);
    free(p->final_coeffs);
    free(p);
}

static void put_u64le(uint8_t *b, uint64_t v)
{
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
}

size_t o_fri_serialize(const ofri_proto *p, uint8_t *buf, size_t cap)
{
    size_t need = 8 + 32 * (p->num_steps + 1) + 32 * p->num_steps + 32 + 8 +
                  32 * p->output_coeffs_at_degree_plus_one;
    if (!buf || cap < need) return need;
    size_t o = 0;
    put_u64le(buf + o, p->num_steps); o += 8;
    memcpy(buf + o, p->l0_nodes + 32, 32); o += 32;
    for (size_t i = 0; i < p->num_steps; i++) { memcpy(buf + o, p->inter_nodes[i] + 32, 32); o += 32; }
    for (size_t i = 0; i < p->num_steps; i++) {
        for (int l = 0; l < 4; l++) put_u64le(buf + o + 8 * l, p->challenges[i].l[l]);
        o += 32;
    }
    memcpy(buf + o, p->final_root, 32); o += 32;
    put_u64le(buf + o, p->output_coeffs_at_degree_plus_one); o += 8;
    for (size_t i = 0; i < p->output_coeffs_at_degree_plus_one; i++) {
        for (int l = 0; l < 4; l++) put_u64le(buf + o + 8 * l, p->final_coeffs[i].l[l]);
        o += 32;
    }
    return o;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic input: index-addressable, so the GPU (hodor_gen_elements_dev) and
 * this oracle produce the same buffer from (seed, index) with no host staging.
 *   w(m)      = output number m of the SplitMix64 stream seeded `seed`
 *             = mix(seed + (m + 1) * 0x9E3779B97F4A7C15)
 *   cand(i,t) = limbs w(4*(16 i + t) + k), k = 0..3, top limb masked to NUM_BITS - 192 bits
 *   a[i]      = the first cand(i, t), t = 0..15, that is < p (canonical residue; after 16
 *               rejections — probability < 2^-16 per element even at the worst mask — the top limb is cleared),
 *               converted to Montgomery form as Fr::from_repr does.
 * The reference draws test inputs from rand 0.4's XorShiftRng (src/fft/mod.rs:71-77), which is
 * sequential; any uniform canonical residues exercise the same code.
 * ------------------------------------------------------------------------------------------ */
static uint64_t splitmix64_out(uint64_t seed, uint64_t m)
{
    uint64_t z = seed + (m + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct { const ofield *f; ofr *out; uint64_t first, seed; } gen_ctx;
static void gen_chunk(void *vctx, size_t ci, size_t start, size_t len)
{
    (void)ci;
    gen_ctx *c = (gen_ctx *)vctx;
    const ofield *f = c->f;
    const uint64_t mask = f->num_bits >= 256 ? ~0ULL : ((1ULL << (f->num_bits - 192)) - 1);
    for (size_t r = start; r < start + len; r++) {
        uint64_t i = c->first + r, cand[4];
        int ok = 0;
        for (uint64_t t = 0; t < 16 && !ok; t++) {
            for (uint64_t k = 0; k < 4; k++) cand[k] = splitmix64_out(c->seed, 4 * (16 * i + t) + k);
            cand[3] &= mask;
            ok = !geq4(cand, f->p);
        }
        if (!ok) cand[3] = 0;
        ofr_from_repr(f, &c->out[r], cand);
    }
}

void o_gen_elements(const ofield *f, ofr *out, uint64_t first_index, size_t count, uint64_t seed,
                    uint32_t cpus)
{
    gen_ctx c = {f, out, first_index, seed};
    worker_scope(cpus, count, gen_chunk, &c);
}

