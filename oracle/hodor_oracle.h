/*
 * hodor_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the NTT / LDE / Merkle-commit / FRI-commit hot path of
 * matter-labs/hodor.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may link or call this; the product path (hodor_amd/csrc,
 * libhodor_gpu.so) never does and fails loudly without a GPU.
 *
 * PARITY STATUS: **parity unpinned against the Rust binary.**  The reference is a
 * Rust crate (no rustc/cargo in this image -> unbuildable here, no oracle/_ref) and
 * none of its own tests holds a known-answer vector for this path (all are
 * differential).  The arithmetic lives in un-vendored crates:
 * ff_ce "0.7" (Montgomery Fr, R = 2^256 for 4-limb fields) and blake2s_simd "0.5"
 * (RFC 7693).  This restatement is pinned instead against
 *   (i)  Python big-int mathematics (oracle/pyref.py) and SURVEY.md Appendix A/B values,
 *   (ii) hashlib.blake2s(key=..., person=...) as an independent RFC 7693 implementation,
 *   (iii) the reference's own differential identities restated as tests (tests/).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef HODOR_ORACLE_H
#define HODOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Fr(FrRepr([u64;4])) — Montgomery form, little-endian limbs (src/bn256.rs:4-7, ff_ce derive). */
typedef struct { uint64_t l[4]; } ofr;

typedef struct {
    uint64_t p[4];        /* modulus */
    uint64_t pinv;        /* -p^{-1} mod 2^64 */
    ofr r;                /* R mod p   (== Fr::one()) */
    ofr r2;               /* R^2 mod p */
    ofr generator;        /* multiplicative_generator(), Montgomery */
    ofr root_of_unity;    /* generator^t, p-1 = 2^S * t, Montgomery */
    uint32_t s;           /* F::S */
    uint32_t num_bits;    /* F::NUM_BITS */
    uint32_t capacity;    /* F::CAPACITY = NUM_BITS - 1 */
} ofield;

/* ---- field (ff_ce PrimeField semantics) ---- */
int  ofield_init(ofield *f, const uint64_t modulus[4], uint64_t generator);
void ofr_add(const ofield *f, ofr *a, const ofr *b);              /* a += b  */
void ofr_sub(const ofield *f, ofr *a, const ofr *b);              /* a -= b  */
void ofr_neg(const ofield *f, ofr *a);
void ofr_dbl(const ofield *f, ofr *a);
void ofr_mul(const ofield *f, ofr *a, const ofr *b);              /* a *= b  */
void ofr_sqr(const ofield *f, ofr *a);
void ofr_pow(const ofield *f, ofr *out, const ofr *base, uint64_t e);
int  ofr_inverse(const ofield *f, ofr *out, const ofr *a);        /* 0 ok, -1 if a == 0 */
int  ofr_from_repr(const ofield *f, ofr *out, const uint64_t canon[4]);  /* -1 if >= p */
void ofr_into_repr(const ofield *f, uint64_t canon[4], const ofr *a);
void ofr_from_u64(const ofield *f, ofr *out, uint64_t v);
int  ofr_eq(const ofr *a, const ofr *b);
int  ofr_is_zero(const ofr *a);

/* ---- Domain::new_for_size (src/domains/mod.rs:21-44) ---- */
typedef struct { uint64_t size; uint64_t power_of_two; ofr generator; } odomain;
int odomain_new_for_size(const ofield *f, uint64_t size, odomain *out);  /* -1: SynthesisError */

/* ---- transforms (in place, natural -> natural) ---- */
void o_serial_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n);          /* src/fft/fft.rs:21-66 */
void o_serial_fft_radix_4(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n);  /* src/fft/radix4_fft/mod.rs:45-123 */
void o_parallel_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                    uint32_t log_cpus);                                                          /* src/fft/fft.rs:68-124 */
void o_best_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                uint32_t cpus);                                                                  /* src/fft/fft.rs:5-19 */
void o_serial_dit_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                      size_t non_zero_entries_count);                                            /* src/fft/dit_fft/mod.rs:4-53 */
void o_parallel_dit_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                        uint32_t log_cpus, size_t non_zero_entries_count);                       /* :55-113 */
void o_best_dit_fft(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                    uint32_t cpus, size_t non_zero_entries_count);                               /* :114-123 */
void o_serial_lde(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                  size_t lde_factor);                                                            /* src/fft/lde.rs:15-126 */
/* round 6: the remaining parallel forms (-1: the reference's asserts on log_n / log_cpus fail) */
int  o_parallel_fft_radix_4(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                            uint32_t log_cpus);                                                  /* src/fft/radix4_fft/mod.rs:125-184 */
int  o_best_fft_radix_4(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n,
                        uint32_t cpus);                                                          /* src/fft/radix4_fft/mod.rs:5-20 */
int  o_parallel_lde(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, uint32_t log_cpus,
                    size_t lde_factor);                                                          /* src/fft/lde.rs:128-193 */
int  o_best_lde(const ofield *f, ofr *a, size_t n, const ofr *omega, uint32_t log_n, size_t lde_factor,
                uint32_t cpus);                                                                  /* src/fft/lde.rs:4-13 */
void o_distribute_powers(const ofield *f, ofr *a, size_t n, const ofr *g, uint32_t cpus);       /* src/fft/mod.rs:110-123 */
void o_naive_dft(const ofield *f, const ofr *in, ofr *out, size_t n, const ofr *omega);         /* definition, O(n^2) */

/* ---- Polynomial<F, _> (src/polynomials/mod.rs) ---- */
int o_poly_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus);          /* :611-624 */
int o_poly_coset_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus);    /* :626-631 */
int o_poly_ifft(const ofield *f, ofr *a, size_t n, uint32_t cpus);         /* :773-798 */
int o_poly_icoset_fft(const ofield *f, ofr *a, size_t n, uint32_t cpus);   /* :800-807 */
/* lde_using_multiple_cosets :418-482 (coset=0) / coset_lde_using_multiple_cosets :544-609 (coset=1).
 * out has n*factor elements. */
int o_poly_lde(const ofield *f, const ofr *coeffs, size_t n, size_t factor, int coset,
               ofr *out, uint32_t cpus);
/* value-form ops (src/polynomials/mod.rs:60-83, 657-671, 744-771, 817-954); op codes as in
 * include/hodor_gpu.h (binary: 0 add 1 sub 2 mul; unary: 0 negate 1 square 2 pow 3 scale 4 add_constant 5 sub_constant) */
void o_poly_binary(const ofield *f, ofr *a, const ofr *b, size_t n, int op);
void o_poly_add_scaled(const ofield *f, ofr *a, const ofr *b, size_t n, const ofr *scaling);
int o_poly_coset_fft_for_generator(const ofield *f, ofr *a, size_t n, const ofr *gen, uint32_t cpus);      /* src/polynomials/mod.rs:633-638 */
int o_poly_icoset_fft_for_generator(const ofield *f, ofr *a, size_t n, const ofr *geninv, uint32_t cpus);  /* :809-815 */
int o_poly_degree_one_on_domain(const ofield *f, ofr *out, size_t n, const ofr *alpha, const ofr *c, int coset);   /* src/polynomials/mod.rs:229-290 */
void o_poly_unary(const ofield *f, ofr *a, size_t n, int op, const ofr *c, uint64_t e);
int  o_poly_batch_inversion(const ofield *f, ofr *a, size_t n);   /* -1 (a untouched) if any zero, :909 */
void o_poly_evaluate_at(const ofield *f, const ofr *coeffs, size_t n, const ofr *g, ofr *out);  /* :685-711, one chunk */
void o_poly_evaluate_at_mt(const ofield *f, const ofr *coeffs, size_t n, const ofr *g, ofr *out,
                           uint32_t cpus);                       /* :685-711 with its Worker chunks */

/* ---- BLAKE2s IOP (src/iop/blake2s_trivial_iop.rs) ---- */
void o_blake2s(uint8_t out[32], const uint8_t *key, size_t keylen, const uint8_t *personal,
               size_t personal_len, const uint8_t *data, size_t len);   /* RFC 7693, blake2s_simd Params */
void o_hash_leaf(uint8_t out[32], const ofr *leaf);                     /* :36-42, :81-91 */
void o_hash_node(uint8_t out[32], const uint8_t l[32], const uint8_t r[32]);   /* :93-104 */
int  o_iop_create(const ofr *leafs, size_t n, uint8_t *nodes /* n*32 */, uint32_t cpus);  /* :131-219 */
void o_interpret_hash(const ofield *f, const uint8_t h[32], ofr *out);  /* :48-60 */
size_t o_iop_path(const uint8_t *nodes, const ofr *leafs, size_t n, size_t tree_index,
                  uint8_t *path /* log2(n)*32 */);                      /* :251-279 */
int  o_iop_verify(const uint8_t root[32], const ofr *leaf, const uint8_t *path, size_t path_len,
                  size_t tree_index);                                   /* :236-249 */

/* ---- coset combining (README.md:46 "Proof size optimization with coset combining", unchecked there; the seam
 * is the CosetCombiner trait, src/iop/mod.rs:22-34, whose only instance is TrivialCombiner,
 * src/iop/trivial_coset_combiner.rs:17-53).  COSET2 = a size-2 combiner as an opt-in tree format DEFINED BY THIS
 * BUILD (the reference has none to be compared with):
 *   natural index i of n values  <->  tree element index t = 2 (i mod n/2) + (i div n/2)   (natural_index_into_tree_index)
 *   leaf k (k < n/2) = the 64 bytes  value[k] || value[k + n/2]  — the coset FRI opens together
 *   (get_coset_for_natural_index :29-35, src/fri/query_producer.rs:27-34), hashed with ONE keyed BLAKE2s call under
 *   the personalisation "Shaftoe2" (never equal to a node hash of the same 64 bytes);
 *   nodes: the heap array of a tree over n/2 leaves, (n/2) x 32 bytes, root nodes[1]; path: log2(n) - 1 digests.
 * n >= 4 (at least two leaves). */
enum { O_COMBINER_TRIVIAL = 0, O_COMBINER_COSET2 = 1 };
void o_hash_leaf_pair(uint8_t out[32], const ofr *lo, const ofr *hi);
int  o_iop_create_coset2(const ofr *values, size_t n, uint8_t *nodes /* (n/2)*32 */, uint32_t cpus);
size_t o_iop_path_coset2(const uint8_t *nodes, const ofr *values, size_t n, size_t natural_index,
                         uint8_t *path /* (log2(n)-1)*32 */);
int  o_iop_verify_coset2(const uint8_t root[32], const ofr *lo, const ofr *hi, const uint8_t *path,
                         size_t path_len, size_t leaf_index);

/* ---- FRI commit phase (src/fri/fri_on_values.rs:11-159) ---- */
typedef struct {
    size_t num_steps;
    size_t initial_degree_plus_one, output_coeffs_at_degree_plus_one, lde_factor;
    uint8_t *l0_nodes;            /* N*32: l0_commitment tree */
    uint8_t **inter_nodes;        /* num_steps trees */
    ofr **inter_values;           /* num_steps vectors, sizes N/2, N/4, ... */
    size_t *inter_sizes;
    ofr *challenges;              /* num_steps */
    uint8_t final_root[32];
    ofr *final_coeffs;            /* output_coeffs_at_degree_plus_one */
} ofri_proto;
int  o_fri_commit(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                  size_t out_deg_plus_one, uint32_t cpus, ofri_proto **out);
/* the same commit phase with every oracle (l0 and intermediates) built by `combiner`; the trees of a COSET2
 * prototype hold (size/2)*32 bytes.  Every committed vector needs >= 4 values (lde_factor * out_deg >= 4). */
int  o_fri_commit_combined(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                           size_t out_deg_plus_one, int combiner, uint32_t cpus, ofri_proto **out);
/* NaiveFriIop::proof_from_lde_through_coefficients — src/fri/mod.rs:156-248: the same prototype via ifft ->
 * coefficient folds a_2i + beta a_(2i+1) -> lde per round (the reference asserts it equal to the by-values
 * prototype, :338-343). */
int  o_fri_commit_through_coefficients(const ofield *f, const ofr *lde_values, size_t n, size_t lde_factor,
                                       size_t out_deg_plus_one, int combiner, uint32_t cpus, ofri_proto **out);
void o_fri_free(ofri_proto *p);
/* canonical prototype encoding (defined by this build, the reference has none — SURVEY F10):
 * u64le num_steps | roots[num_steps+1] (l0 + intermediates, 32 B each) | challenges[num_steps]
 * (32 B Montgomery LE) | final_root (32 B) | u64le n_final | final_coeffs (32 B each). */
size_t o_fri_serialize(const ofri_proto *p, uint8_t *buf, size_t cap);

/* ---- synthetic input, index-addressable; twin of hodor_gen_elements_dev ---- */
void o_gen_elements(const ofield *f, ofr *out, uint64_t first_index, size_t count, uint64_t seed,
                    uint32_t cpus);

/* threads helper */
uint32_t o_num_cpus(void);

#ifdef __cplusplus
}
#endif
#endif
