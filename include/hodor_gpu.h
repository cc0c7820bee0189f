/*
 * hodor_gpu.h — C ABI of the MI355X-native NTT / LDE / Merkle-commit / FRI-commit path.
 *
 * This is the drop-in boundary for matter-labs/hodor's prover: the reference has no FFI seam of its
 * own (one Rust crate, monomorphised over F: PrimeField), so the exports below are exactly the L2/L3
 * functions its upper layers (src/arp, src/ali, src/prover, src/fri) call, re-expressed as
 * `extern "C"` over plain pointers and sizes.  Each export cites the Rust item it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the Rust-side binding.
 *
 * Element type: `hodor_fr` is the memory image of Rust `Fr(FrRepr([u64; 4]))` — Montgomery form,
 * R = 2^256, limb 0 least significant, value in [0, p) (src/bn256.rs:4-7, ff_ce derive).  A Rust
 * `&mut [Fr]` maps to `(ptr as *mut hodor_fr, len)`.
 *
 * Two families:
 *   - slice API (host pointers, synchronous): copies in, runs the HIP kernels, copies out.
 *   - `_dev` API (device pointers + a hipStream_t passed as void*): stream-ordered, no host copies;
 *     this is what the benchmarks and a device-resident prover use.
 * All entry points return an int status and never throw or abort across the boundary.  There is no
 * CPU fallback: without a usable HIP device every compute entry point returns HODOR_ERR_DEVICE.
 */
#ifndef HODOR_GPU_H
#define HODOR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } hodor_fr;
typedef struct hodor_ctx hodor_ctx;
typedef struct hodor_fri_proto hodor_fri_proto;

enum {
    HODOR_OK = 0,
    HODOR_ERR_SIZE = 1,     /* non power of two, n != 1<<log_n, log_n > S ...   (SynthesisError::Error, src/domains/mod.rs:30-32; asserts at src/fft/fft.rs:34, src/iop/blake2s_trivial_iop.rs:137) */
    HODOR_ERR_INVALID = 2,  /* null pointer / unsupported modulus / non-invertible element */
    HODOR_ERR_DEVICE = 3    /* HIP runtime error or no device */
};

typedef struct {
    uint64_t modulus[4];
    uint32_t s;               /* F::S  (2-adicity)           */
    uint32_t num_bits;        /* F::NUM_BITS                 */
    uint32_t capacity;        /* F::CAPACITY                 */
    hodor_fr one;             /* F::one()  = R mod p         */
    hodor_fr generator;       /* F::multiplicative_generator() */
    hodor_fr root_of_unity;   /* F::root_of_unity()          */
} hodor_field_info;

/* ---- context: the constants `#[derive(PrimeField)]` generates (src/bn256.rs:4-7,
 * src/experiments/mod.rs:18-21) + device state (streams, twiddle cache, scratch).
 * `modulus` must be an odd prime with 2^239 < p < 2^255 (4-limb ff_ce field, R = 2^256; the reference's
 * two 4-limb fields have 255 and 252 bits). */
/* ABI revision of this header: bumped whenever an existing entry point changes its C signature or meaning (adding entry
 * points does not bump it).  A caller built against another revision must not call into the library: check once.
 *   4  hodor_fri_verify_proof_strict(_combined) take expected_lde_factor and expected_output_coeffs_at_degree_plus_one
 *      BEFORE natural_element_index (round 4; the round-3 form had five value arguments)
 *   5  COSET2 leaves hash under the personalisation "Shaftoe2" (trees, paths and proofs of rounds 3-4 in that opt-in
 *      format do not verify any more; the reference's format is untouched); hodor_iop_verify_combined (COSET2) refuses
 *      paths of any length but log2(n) - 1 and non-canonical values;
 *      receive buffers of the direct transports are the library's (hodor_exchange_direct_alloc_recv) */
#define HODOR_ABI_VERSION 5
int  hodor_abi_version(void);
/* With a device (device >= 0) the new context first runs a START-UP SELF-TEST (csrc/abi_selftest.hip, about a millisecond
 * and a half): one 2^10-point transform per kernel instantiation it will use, a 2^16-point fft -> ifft round trip and
 * one small FRI commit (tree, challenge, two folds, final coefficient), every result compared with the library's HOST
 * implementations of the same arithmetic.  A mismatch — a compiler, code-object or driver change that broke an
 * assumption of the device code — returns HODOR_ERR_DEVICE and no context; hodor_last_error(NULL) says which check
 * failed.  HODOR_SELFTEST=0 in the environment skips the test.  device < 0: a host-only context (field helpers,
 * verifiers, transcript), no test, no HIP call. */
int  hodor_ctx_create(const uint64_t modulus[4], uint64_t generator, int device, hodor_ctx **out);
/* Destroy after every prototype and handle obtained from this context has been freed (they hand their device memory
 * back to the context's pool), every hodor_exchange created on it has been destroyed (the call
 * is refused otherwise: the context stays alive) and no call on it is in flight. */
void hodor_ctx_destroy(hodor_ctx *ctx);
/* The same with a verdict: HODOR_OK = destroyed; HODOR_ERR_INVALID = REFUSED, the context is still alive (exchanges,
 * handles or prototypes of it exist — hodor_last_error says which): free them and call again.  hodor_ctx_destroy is this
 * call with the verdict dropped. */
int  hodor_ctx_try_destroy(hodor_ctx *ctx);
int  hodor_ctx_field_info(const hodor_ctx *ctx, hodor_field_info *out);
const char *hodor_last_error(const hodor_ctx *ctx);   /* ctx == NULL: why this thread's last hodor_ctx_create failed */
int  hodor_ctx_synchronize(hodor_ctx *ctx);
/* Tuning variables found in the environment when the library first read them ("NAME=value ...", empty
 * when none): HODOR_MAX_LOG_R, HODOR_TILE_LOG, HODOR_MIN_LOG_C, HODOR_TW_HI_MAX_LOG, HODOR_NTT_THREADS, HODOR_NTT_TW_SUB, HODOR_NTT_W9, HODOR_NTT_P1,
 * HODOR_MERKLE_TAIL_LOG, HODOR_MERKLE_LAT_LOG, HODOR_FRI_TAIL, HODOR_FRI_FUSE_FOLD, HODOR_BATCHINV_SEQ, HODOR_TABLE_CACHE,
 * HODOR_POOL_CACHE_GIB, HODOR_SLICE_SERIAL.
 * They are read once per process, change schedules only (never results), and a benchmark must echo
 * them (bench.py does, and refuses to run with any of them set unless told otherwise). */
const char *hodor_knobs_set(void);
/* Debug aid (fault injection): HODOR_DEBUG_FAIL_ALLOC=<k> makes the k-th device / pinned allocation the library asks the
 * runtime for in this process fail as out of memory ("<k>+": every one from the k-th on); this returns how many it has
 * asked for so far.  Every entry point must answer such a failure with an error code and leave every object usable
 * (tests/test_gpu_alloc_faults.py walks k over a whole proof-shaped sequence). */
long long hodor_debug_alloc_calls(void);
void hodor_debug_fail_alloc(long long k, int from_on);   /* the same, armed at run time: the k-th allocation FROM NOW (0 = disarm) */

/* ---- scalar field helpers on the host (ff_ce Field/PrimeField methods the callers use to derive
 * omegainv / minv / geninv, src/polynomials/mod.rs:146-166) */
int hodor_fr_mul(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out);
int hodor_fr_add(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out);
int hodor_fr_sub(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out);
int hodor_fr_pow(const hodor_ctx *ctx, const hodor_fr *a, uint64_t e, hodor_fr *out);
int hodor_fr_inverse(const hodor_ctx *ctx, const hodor_fr *a, hodor_fr *out);
int hodor_fr_from_repr(const hodor_ctx *ctx, const uint64_t canonical[4], hodor_fr *out);
int hodor_fr_into_repr(const hodor_ctx *ctx, const hodor_fr *a, uint64_t canonical[4]);

/* ---- Domain::new_for_size (src/domains/mod.rs:21-44) */
int hodor_domain_new_for_size(const hodor_ctx *ctx, uint64_t size, uint64_t *out_size,
                              uint32_t *out_log_n, hodor_fr *out_generator);

/* ================================ slice API (host memory) ================================ */

/* best_fft(a: &mut [F], worker, omega: &F, log_n, hint)  — src/fft/mod.rs:50, src/fft/fft.rs:5-19.
 * In place, natural -> natural, A[k] = sum_i a[i] omega^(ik); omega is any element of order n. */
int hodor_fft(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *omega, uint32_t log_n);
/* best_lde(a, worker, omega, log_n, lde_factor) — src/fft/mod.rs:46, src/fft/lde.rs:4-13.
 * Same output as hodor_fft when a[n/lde_factor..] is zero (which the callee assumes). */
int hodor_lde(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *omega, uint32_t log_n,
              size_t lde_factor);
/* distribute_powers(coeffs, worker, g): a[i] *= g^i — src/fft/mod.rs:110-123 */
int hodor_distribute_powers(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *g);

/* Polynomial<F, Coefficients>::{fft, coset_fft} — src/polynomials/mod.rs:611-631 (n power of two) */
int hodor_poly_fft(hodor_ctx *ctx, hodor_fr *a, size_t n);
int hodor_poly_coset_fft(hodor_ctx *ctx, hodor_fr *a, size_t n);
/* Polynomial<F, Values>::{ifft, icoset_fft} — src/polynomials/mod.rs:773-807 */
int hodor_poly_ifft(hodor_ctx *ctx, hodor_fr *a, size_t n);
int hodor_poly_icoset_fft(hodor_ctx *ctx, hodor_fr *a, size_t n);
/* coset_fft_for_generator (:633-638: distribute_powers(gen) then fft) and icoset_fft_for_generator (:809-815:
 * ifft then distribute_powers(geninv) — the caller passes the INVERSE, as the Rust caller does) */
int hodor_poly_coset_fft_for_generator(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *gen);
int hodor_poly_icoset_fft_for_generator(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *geninv);
/* Polynomial::lde / coset_lde (-> lde_using_multiple_cosets / coset_lde_using_multiple_cosets)
 * — src/polynomials/mod.rs:343, 349, 418-482, 544-609.  out has n*factor elements:
 * out[idx] = P(Omega^idx)  resp.  P(g * Omega^idx), natural order on the size n*factor domain. */
int hodor_poly_lde(hodor_ctx *ctx, const hodor_fr *coeffs, size_t n, size_t factor, hodor_fr *out);
int hodor_poly_coset_lde(hodor_ctx *ctx, const hodor_fr *coeffs, size_t n, size_t factor,
                         hodor_fr *out);

/* IopTree::create + get_root — src/iop/blake2s_trivial_iop.rs:131-224.  nodes: n x 32 bytes, heap
 * layout (root = nodes[32..64]).  n power of two, n >= 2. */
int hodor_iop_create(hodor_ctx *ctx, const hodor_fr *leafs, size_t n, uint8_t *nodes);
/* encode_root_into_challenge / interpret_hash — :48-60, :226-234 (host) */
int hodor_iop_challenge(const hodor_ctx *ctx, const uint8_t root[32], hodor_fr *out);
/* IopTreeHasher::hash_leaf / hash_node — :81-104 (host, single digests) */
int hodor_hash_leaf(const hodor_ctx *ctx, const hodor_fr *leaf, uint8_t out[32]);
int hodor_hash_node(const hodor_ctx *ctx, const uint8_t left[32], const uint8_t right[32], uint8_t out[32]);
/* get_path — :251-279 (host; path holds log2(n) digests, returns the count in *path_len) */
int hodor_iop_path(const hodor_ctx *ctx, const uint8_t *nodes, const hodor_fr *leafs, size_t n,
                   size_t tree_index, uint8_t *path, size_t *path_len);
/* verify — :236-249 (host); *ok = 1 when the path leads to root */
int hodor_iop_verify(const hodor_ctx *ctx, const uint8_t root[32], const hodor_fr *leaf,
                     const uint8_t *path, size_t path_len, size_t tree_index, int *ok);

/* ---- coset combining: the tree format as a parameter (CosetCombiner, src/iop/mod.rs:22-34) --------------------
 * The reference's IOP is generic over a CosetCombiner whose only instance is TrivialCombiner
 * (src/iop/trivial_coset_combiner.rs:17-53: identity index maps, coset {i, i + n/2}); its README lists "Proof size
 * optimization with coset combining" as not done (README.md:46).  FRI always opens the two members of a coset
 * together (src/fri/query_producer.rs:27-34), so HODOR_COMBINER_COSET2 — an opt-in format DEFINED BY THIS BUILD —
 * commits them as ONE leaf:
 *     natural index i of n values  <->  tree element t = 2 (i mod n/2) + (i div n/2)   (natural_index_into_tree_index)
 *     leaf k (k < n/2) = the 64 bytes value[k] || value[k + n/2], hashed with ONE keyed BLAKE2s call under a
 *                        personalisation of its own ("Shaftoe2"; nodes and the reference's leaves: "Shaftoe") — the
 *                        children of an interior node can therefore never be opened as a "value pair" (round 5)
 *     nodes = heap array of the tree over those n/2 leaves: (n/2) x 32 bytes, root = nodes[32..64]
 *     path  = log2(n) - 1 digests; a query returns both values of the coset
 * -> n compressions per tree instead of 2n, one path per FRI round instead of two.  n >= 4.  Every entry point
 * without a `combiner` argument is the TRIVIAL format, byte for byte the reference's. */
enum { HODOR_COMBINER_TRIVIAL = 0, HODOR_COMBINER_COSET2 = 1 };
/* IopTree::create (:131-219) in the chosen format; nodes: n x 32 bytes (TRIVIAL) / (n/2) x 32 bytes (COSET2) */
int hodor_iop_create_combined(hodor_ctx *ctx, const hodor_fr *leafs, size_t n, int combiner, uint8_t *nodes);
/* hash of one leaf: `values` holds 1 element (TRIVIAL, = hodor_hash_leaf) or the 2 of a coset (COSET2) */
int hodor_hash_leaf_combined(const hodor_ctx *ctx, const hodor_fr *values, int combiner, uint8_t out[32]);
/* get_path (:251-279) of the leaf that holds `natural_index` (host) */
int hodor_iop_path_combined(const hodor_ctx *ctx, const uint8_t *nodes, const hodor_fr *leafs, size_t n,
                            int combiner, size_t natural_index, uint8_t *path, size_t *path_len);
/* verify (:236-249): `values` = the queried element (TRIVIAL) or {value[k], value[k + n/2]}, k = natural_index mod n/2
 * (COSET2); n = the size of the committed vector */
int hodor_iop_verify_combined(const hodor_ctx *ctx, const uint8_t root[32], const hodor_fr *values,
                              const uint8_t *path, size_t path_len, size_t natural_index, size_t n, int combiner,
                              int *ok);

/* FriIop::proof_from_lde (NaiveFriIop::proof_from_lde_by_values) — src/fri/fri_on_values.rs:11-159.
 * The result mirrors FRIProofPrototype field for field (src/fri/mod.rs:106-117). */
int  hodor_fri_commit(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                      size_t output_coeffs_at_degree_plus_one, hodor_fri_proto **out);
/* the same with every oracle (l0 and intermediates) in the chosen format; COSET2 needs lde_factor *
 * output_coeffs_at_degree_plus_one >= 4 (two combined leaves in the last tree) */
int  hodor_fri_commit_combined(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                               size_t output_coeffs_at_degree_plus_one, int combiner, hodor_fri_proto **out);
/* NaiveFriIop::proof_from_lde_through_coefficients — src/fri/mod.rs:156-248: the same prototype by another route
 * (ifft of the codeword, coefficient folds a_2i + beta a_(2i+1), Polynomial::lde + commit per round); the
 * reference's test_one_fri_step asserts it equal to proof_from_lde_by_values field for field (:338-343), and so do
 * this library's tests.  `combiner` as above (the reference: HODOR_COMBINER_TRIVIAL). */
int  hodor_fri_commit_through_coefficients(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                                           size_t output_coeffs_at_degree_plus_one, int combiner,
                                           hodor_fri_proto **out);
int  hodor_fri_combiner(const hodor_fri_proto *p);   /* HODOR_COMBINER_* of the prototype's trees */
void hodor_fri_free(hodor_fri_proto *p);   /* before hodor_ctx_destroy of the context it came from */
size_t hodor_fri_num_steps(const hodor_fri_proto *p);
/* roots: l0 root followed by the intermediate roots -> (num_steps + 1) x 32 bytes (get_roots, src/fri/mod.rs:120-128) */
int hodor_fri_roots(const hodor_fri_proto *p, uint8_t *roots);
int hodor_fri_final_root(const hodor_fri_proto *p, uint8_t root[32]);
int hodor_fri_challenges(const hodor_fri_proto *p, hodor_fr *challenges /* num_steps */);
int hodor_fri_final_coefficients(const hodor_fri_proto *p, hodor_fr *coeffs /* output_coeffs_at_degree_plus_one */);
/* intermediate_values[step] (size n >> (step+1)) and the tree of step (-1 = l0 tree, size n; a COSET2 tree has
 * half as many entries as its vector) */
int hodor_fri_intermediate_values(hodor_fri_proto *p, size_t step, hodor_fr *values);
int hodor_fri_tree_nodes(hodor_fri_proto *p, int step, uint8_t *nodes);
/* canonical prototype encoding (the reference defines none, README.md:44):
 * u64le num_steps | roots | challenges | final_root | u64le n_final | final_coefficients.
 * Returns the byte count; writes only when buf != NULL and cap is large enough. */
size_t hodor_fri_serialize(const hodor_fri_proto *p, uint8_t *buf, size_t cap);

/* Blake2sTranscript (src/transcript/mod.rs:10-80) and Verifier::bytes_to_challenge_index
 * (src/verifier/mod.rs:246-263): host-side Fiat-Shamir glue, one running keyed BLAKE2s stream whose
 * finalize is non-destructive and whose digest is re-absorbed after every challenge. */
typedef struct hodor_transcript hodor_transcript;
int  hodor_transcript_new(const hodor_ctx *ctx, hodor_transcript **out);
void hodor_transcript_free(hodor_transcript *t);
int  hodor_transcript_commit_bytes(hodor_transcript *t, const uint8_t *bytes, size_t len);
int  hodor_transcript_commit_field_element(hodor_transcript *t, const hodor_fr *element);
int  hodor_transcript_get_challenge_bytes(hodor_transcript *t, uint8_t out[32]);
int  hodor_transcript_get_challenge(hodor_transcript *t, hodor_fr *out);
size_t hodor_bytes_to_challenge_index(const uint8_t *bytes, size_t len, size_t lde_size, size_t lde_factor);

/* ================================ device API (device memory) ============================== */
/* `stream` is a hipStream_t (NULL = the HIP legacy default stream, which is also PyTorch-ROCm's default).
 * All work is enqueued in stream order; nothing synchronises with the host unless stated.
 * Ordering rules of one context:
 *   - it owns ONE scratch pool (the ping-pong buffers of multi-pass transforms; batch_inversion,
 *     evaluate_at and fri_commit use it too), one twiddle-table cache and one device-memory pool (FRI prototypes, handles);
 *   - the slice API runs on streams the context creates itself (one non-blocking compute stream + three
 *     copy lanes, invisible to the caller) and is internally serialised on them;
 *   - `_dev` calls of one context on DIFFERENT streams are ordered on the scratch pool by the library (the
 *     stream of a call that needs the pool first waits for everything the previous user's stream had been
 *     given), so they are safe — also next to slice-API calls of other threads — but they do not overlap
 *     there.  Buffers the caller passes in are the caller's to order.  Use one context per stream when the
 *     streams are meant to run concurrently. */
int hodor_buf_alloc(hodor_ctx *ctx, size_t bytes, void **dev_ptr);
int hodor_buf_free(hodor_ctx *ctx, void *dev_ptr);
int hodor_buf_upload(hodor_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int hodor_buf_download(hodor_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
/* Slice-API callers that reuse their buffers (the prover's Vec<F> per register) can pin them once: copies
 * from/to a registered range run as DMA at the PCIe rate (hipHostRegister).  Unregister before freeing. */
int hodor_host_register(hodor_ctx *ctx, void *host_ptr, size_t bytes);
int hodor_host_unregister(hodor_ctx *ctx, void *host_ptr);

/* out-of-place natural->natural NTT of size 1<<log_n with an arbitrary omega (src == dst allowed) */
int hodor_fft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n,
                  const hodor_fr *omega);
/* `batch` independent transforms of size 1<<log_n stored back to back (the row/column transforms of
 * the 6-step decomposition, cf. the P sub-FFTs of parallel_fft, src/fft/fft.rs:83-108) */
int hodor_fft_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n,
                        size_t batch, const hodor_fr *omega);
/* 6-step twiddle step: a[r][c] *= omega^((row0 + r) * c) (* scale if non-NULL) for a rows x cols
 * row-major block; omega has order 2^log_order (cf. the omega^(j*idx) factors at src/fft/fft.rs:92-103) */
int hodor_twiddle_mul_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t rows, size_t cols,
                          uint64_t row0, const hodor_fr *omega, uint32_t log_order, const hodor_fr *scale);
/* Polynomial-level transforms on the canonical domain of size 1<<log_n */
int hodor_poly_fft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n);
int hodor_poly_ifft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n);
int hodor_poly_coset_fft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n);
int hodor_poly_icoset_fft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n);
int hodor_poly_coset_fft_for_generator_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                           uint32_t log_n, const hodor_fr *gen);
int hodor_poly_icoset_fft_for_generator_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                            uint32_t log_n, const hodor_fr *geninv);
/* LDE: src has 1<<log_n coefficients, dst has (1<<log_n)*factor values; coset != 0 -> coset_lde */
int hodor_poly_lde_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                       uint32_t log_n, size_t factor, int coset);
/* `batch` polynomials at once (all registers of the trace, src/prover/mod.rs:73-80): src holds batch
 * arrays of 1<<log_n coefficients back to back, dst batch arrays of (1<<log_n)*factor values; the
 * commit twin builds batch trees over n leaves each into batch node arrays of n*32 bytes. */
int hodor_poly_lde_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                             uint32_t log_n, size_t factor, int coset, size_t batch);
int hodor_iop_create_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n, size_t batch,
                               uint8_t *nodes);
/* the same in the chosen tree format (HODOR_COMBINER_COSET2: batch node arrays of (n/2)*32 bytes back to back) */
int hodor_iop_create_batch_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n, size_t batch,
                                        int combiner, uint8_t *nodes);
int hodor_distribute_powers_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t n, const hodor_fr *g);
/* Polynomial<F, Coefficients>::evaluate_at_domain_for_degree_one (coset != 0: coset_evaluate_at_...) —
 * src/polynomials/mod.rs:229-258, :260-290: out[i] = alpha * u_i + c for q(x) = c + alpha x, u_i = w^i over the
 * size-n domain's generator w (coset: multiplicative_generator * w^i).  n must be a power of two (the reference
 * rounds the requested size up to one); HODOR_ERR_SIZE otherwise.  The divisor polynomials of the DEEP step
 * (src/ali/per_register/deep.rs:59-66, :128-135) before their batch inversion. */
int hodor_poly_degree_one_on_domain_dev(hodor_ctx *ctx, void *stream, hodor_fr *out, size_t n,
                                        const hodor_fr *alpha, const hodor_fr *c, int coset);
/* The divisor precompute of ALIInstance::from_arp — inverse_divisor_for_dense_constraint_in_coset,
 * src/ali/per_register/mod.rs:60-160 — as ONE launch: out[i] = prod_j (x_i - roots[j]) / (x_i^T - 1) for x_i = g w^i on the
 * coset of the `evaluation_size`-point domain (g the multiplicative generator, T = column_size; both powers of two,
 * evaluation_size / column_size <= 2^16).  The reference fills x_i^T - 1 on the host, batch-inverts and multiplies the
 * root factors in on the host (two as_mut() passes — hodor_poly_as_mut_h replays them as written); this is the
 * device-resident form of the same vector: x^T - 1 takes only evaluation_size / column_size distinct values on the
 * coset, which are inverted on the host.  HODOR_ERR_INVALID when one of them is zero (the reference's batch_inversion
 * returns Err(SynthesisError::Error)).  `roots` is host memory (n_roots elements, may be 0). */
int hodor_poly_dense_divisor_on_coset_dev(hodor_ctx *ctx, void *stream, hodor_fr *out, size_t evaluation_size,
                                          size_t column_size, const hodor_fr *roots, size_t n_roots);
/* PrecomputedOmegas::new_for_domain — src/precomputations/mod.rs:14-66: for the domain of size
 * n = 1<<log_n with generator w: omegas[i] = w^i (n entries), coset[i] = g*w^i (n entries, g the
 * multiplicative generator), omegas_inv[i] = w^-i (n/2 entries).  A NULL output is skipped. */
int hodor_precomputed_omegas_dev(hodor_ctx *ctx, void *stream, uint32_t log_n, hodor_fr *omegas,
                                 hodor_fr *coset, hodor_fr *omegas_inv);
/* ---- value-form polynomial arithmetic on device-resident buffers (the pointwise steps either side of
 * every LDE in ALI: src/polynomials/mod.rs:60-83, 640-683, 744-771, 817-954) ---- */
enum { HODOR_OP_ADD = 0, HODOR_OP_SUB = 1, HODOR_OP_MUL = 2 };
enum { HODOR_UN_NEGATE = 0, HODOR_UN_SQUARE = 1, HODOR_UN_POW = 2, HODOR_UN_SCALE = 3,
       HODOR_UN_ADD_CONSTANT = 4, HODOR_UN_SUB_CONSTANT = 5 };
/* add_assign / sub_assign / mul_assign: a[i] op= b[i] */
int hodor_poly_binary_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, const hodor_fr *b, size_t n, int op);
/* add_assign_scaled: a[i] += scaling * b[i] */
int hodor_poly_add_scaled_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, const hodor_fr *b, size_t n,
                              const hodor_fr *scaling);
/* negate / square / pow(e) / scale(c) / add_constant(c) / sub_constant(c) */
int hodor_poly_unary_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t n, int op, const hodor_fr *c,
                         uint64_t e);
/* One DEEP quotient term in a single pass over the data: acc[i] = (accumulate ? acc[i] : 0) + alpha * (f[i] - value) *
 * divisor_inv[i] (alpha NULL = 1) — the sequence clone / add_constant(-value) / scale(alpha) / mul_assign(divisor^-1) /
 * add_assign of calculate_deep (src/ali/per_register/deep.rs:74-84 for every h1 term, :139-144 for h2), same canonical
 * result, a fifth of the memory traffic.  acc may alias neither f nor divisor_inv. */
int hodor_poly_quotient_term_dev(hodor_ctx *ctx, void *stream, hodor_fr *acc, const hodor_fr *f,
                                 const hodor_fr *divisor_inv, size_t n, const hodor_fr *value, const hodor_fr *alpha,
                                 int accumulate);
/* batch_inversion: a[i] = a[i]^-1; HODOR_ERR_INVALID (data untouched) if any element is zero
 * (SynthesisError::Error, src/polynomials/mod.rs:909).  Synchronises the stream. */
int hodor_poly_batch_inversion_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t n);
/* evaluate_at: *out (host) = sum coeffs[i] g^i.  Synchronises the stream. */
int hodor_poly_evaluate_at_dev(hodor_ctx *ctx, void *stream, const hodor_fr *coeffs, size_t n, const hodor_fr *g,
                               hodor_fr *out);
/* ---- one transform split over the P = 2^log_p GPUs of a node: 4-step / 6-step building blocks --------
 * n = N1 * N2 (N1 = 2^log_n1, N2 = 2^log_n2, P | N1, P | N2), x[n1*N2 + n2], r1 = N1/P, c2 = N2/P.  The
 * library does each rank's local arithmetic with the layout changes fused into the transform kernels'
 * addressing; the CALLER owns the communicator and runs ONE all-to-all of P equal contiguous slabs
 * (r1*c2 elements each; slab t of the send buffer goes to rank t, slab s of the receive buffer came from
 * rank s) between the two calls.  Generalises the reference's Cooley-Tukey split in parallel_fft
 * (src/fft/fft.rs:68-124) to distributed memory.  Per-rank layouts (row-major), rank q:
 *     A: a[n1][j] = x[n1*N2 + q*c2 + j]        N1 x c2   (column block q of the N1 x N2 input matrix)
 *     B: b[i][k2] = X[(q*r1 + i) + N1*k2]      r1 x N2   (row block q of the N1 x N2 output matrix)
 *   forward (omega = primitive n-th root):   A -> hodor_sixstep_columns_dev(inverse = 0) -> all-to-all
 *                                              -> hodor_sixstep_rows_dev(inverse = 0) -> B
 *   inverse (same omega; the library inverts it and folds n^-1 in):
 *                                            B -> hodor_sixstep_rows_dev(inverse = 1) -> all-to-all
 *                                              -> hodor_sixstep_columns_dev(inverse = 1) -> A
 * src != dst; every buffer holds n/P elements.  With P = 1 the pair is a complete transform whose output
 * is the N1 x N2 matrix X[k1 + N1*k2] (hodor_transpose_dev gives natural order).
 * Overlap: with K = 2^log_chunks > 1 the exchange is cut into K all-to-alls of n/(P*K) elements each, so that
 * the caller can put chunk k on the wire while the library works on chunk k+1:
 *   forward  columns(chunk = k): column group k (c2/K columns) of A -> dst = chunk buffer k, [P][r1][c2/K] slabs;
 *            rows(chunk ignored): src = the K received chunk buffers back to back -> B;
 *   inverse  rows(chunk = b): row group b (r1/K rows) of B -> dst = chunk buffer b, [P][r1/K][c2] slabs;
 *            columns(chunk ignored): src = the K received chunk buffers back to back -> A. */
int hodor_sixstep_columns_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n1,
                              uint32_t log_n2, uint32_t log_p, uint32_t rank, const hodor_fr *omega, int inverse,
                              uint32_t log_chunks, uint32_t chunk);
int hodor_sixstep_rows_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n1,
                           uint32_t log_n2, uint32_t log_p, uint32_t rank, const hodor_fr *omega, int inverse,
                           uint32_t log_chunks, uint32_t chunk);
/* Natural block order on either side costs one more all-to-all each, fed by these two copies:
 * pack: a 2^log_rows x 2^log_cols row-major block cut into the P slabs of an all-to-all,
 *       dst[(t*rows + i)*c + j] = src[i*cols + t*c + j], c = cols / P
 *       (natural block [r1][N2] -> pack -> all-to-all = layout A;  layout B -> pack -> all-to-all ->
 *        hodor_transpose_dev(rows = N1, cols = c2) = natural block of the output);
 * transpose: dst[c*rows + r] = src[r*cols + c]. */
int hodor_sixstep_pack_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_rows,
                           uint32_t log_cols, uint32_t log_p);
int hodor_transpose_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, size_t rows, size_t cols);
/* ---- the exchange between the two calls, for a caller without a collective library of its own (the Rust prover):
 * ONE all-to-all of P equal contiguous slabs per transform = the un-shuffle of parallel_fft, src/fft/fft.rs:111-123,
 * across devices.  RCCL is bound at run time (dlopen); nothing but this library has to be linked.
 *   hodor_exchange_available   1 when librccl could be bound in this process
 *   hodor_exchange_unique_id   ncclGetUniqueId: ONE rank calls it and hands the 128 opaque bytes to its peers by any
 *                              channel it has (a file, a socket, MPI)
 *   hodor_exchange_create      ncclCommInitRank on the context's device — collective: every rank calls it with the same
 *                              id; n_ranks a power of two.  The handle owns the communicator, a highest-priority
 *                              communication stream and 1 + 64 events (one "produced" event and a ring of completion
 *                              events, one per ticket).  hodor_exchange_adopt wraps a communicator (ncclComm_t) the
 *                              caller owns; it must live on the context's device and have n_ranks ranks of which
 *                              this is `rank` (checked where the library exposes ncclCommCuDevice / Count / UserRank).
 *   hodor_sixstep_exchange_dev chunk `chunk` of 2^log_chunks (elements [chunk*n_local/K, (chunk+1)*n_local/K) of both
 *                              buffers, P slabs each: slab t of the send piece -> rank t, slab s of the receive piece
 *                              <- rank s) as RCCL's all-to-all (ncclAllToAll; grouped ncclSend/ncclRecv where the
 *                              library lacks it) on the handle's communication stream, ordered
 *                              AFTER everything enqueued on `stream` so far; `stream` does not wait, so the next chunk's
 *                              arithmetic overlaps the wire time; *ticket (may be NULL) numbers the exchange
 *   hodor_sixstep_exchange_wait_dev  `stream` waits for the exchange with that ticket and every earlier one (0: all
 *                              issued so far) — call it with the LAST chunk's ticket before the consuming
 *                              hodor_sixstep_rows_dev / _columns_dev; two transforms in flight on one handle wait for
 *                              their own exchanges only.  Send and receive buffers must stay alive and untouched
 *                              between the two calls.
 * Destroy the handle before its context: the handle keeps a pointer to it, and hodor_ctx_destroy REFUSES (leaves the
 * context alive, sets hodor_last_error) while exchanges created on it exist. */
typedef struct hodor_exchange hodor_exchange;
#define HODOR_EXCHANGE_ID_BYTES 128
#define HODOR_EXCHANGE_MAX_RANKS 8     /* direct transport: the GPUs of one node */
#define HODOR_IPC_HANDLE_BYTES 64
int  hodor_exchange_available(void);
int  hodor_exchange_unique_id(uint8_t id[HODOR_EXCHANGE_ID_BYTES]);
int  hodor_exchange_create(hodor_ctx *ctx, const uint8_t id[HODOR_EXCHANGE_ID_BYTES], uint32_t n_ranks, uint32_t rank,
                           hodor_exchange **out);
int  hodor_exchange_adopt(hodor_ctx *ctx, void *nccl_comm, uint32_t n_ranks, uint32_t rank, hodor_exchange **out);
void hodor_exchange_destroy(hodor_exchange *x);
int  hodor_sixstep_exchange_dev(hodor_exchange *x, void *stream, const hodor_fr *send, hodor_fr *recv, size_t n_local,
                                uint32_t log_chunks, uint32_t chunk, uint64_t *ticket);
int  hodor_sixstep_exchange_wait_dev(hodor_exchange *x, void *stream, uint64_t ticket);
/* ---- direct transport: the exchange without an exchange (csrc/abi_exchange.hip) ------------------------------------
 * Every rank maps every rank's receive buffer into its own address space — hodor_ipc_export / _import (hipIpc*) between
 * processes, plain pointers when the ranks share a process — and the LAST pass of the producing transform stores each
 * output slab straight into the buffer of the rank it is for: no communicator, no copy kernel taking CUs from the
 * VALU-bound transform, no chunks, no send buffer.  Ordering is two generation counters per (slot, peer) in
 * fine-grained device memory, written by one tiny kernel and polled by a one-wave kernel (bounded: ~10 s, then the
 * handle reports HODOR_ERR_DEVICE).  A slot is one receive buffer of n/P elements on every rank; use as many slots as
 * transforms are in flight.
 *   hodor_exchange_create_direct       handle with n_slots slots (no RCCL needed); hodor_exchange_direct_flags: this
 *                                      rank's flag block, to be exported to the peers like a receive buffer
 *   hodor_exchange_direct_set_peers    slot's receive buffers recv[t] and (once) the flag blocks flags[t] of all ranks
 *                                      as mapped in THIS process (recv[rank] / flags[rank]: this rank's own)
 *   producer:  hodor_exchange_direct_begin_dev (the slot may be overwritten: every peer released it)
 *              hodor_sixstep_columns_direct_dev (forward) / hodor_sixstep_rows_direct_dev (inverse), chunked or not
 *              hodor_exchange_direct_signal_dev
 *   consumer:  hodor_exchange_direct_wait_dev (every peer's slab has arrived), then hodor_sixstep_rows_dev (forward) /
 *              hodor_sixstep_columns_dev(inverse = 1) on this rank's own receive buffer, then
 *              hodor_exchange_direct_release_dev
 * All stream-ordered on `stream`.  Unmeasured between real devices (single-GPU boxes): exercised at world 1, with
 * played ranks, and between processes that share the one GPU. */
int  hodor_ipc_export(hodor_ctx *ctx, void *dev_ptr, uint8_t handle[HODOR_IPC_HANDLE_BYTES]);
int  hodor_ipc_import(hodor_ctx *ctx, const uint8_t handle[HODOR_IPC_HANDLE_BYTES], void **dev_ptr);
int  hodor_ipc_close(hodor_ctx *ctx, void *dev_ptr);
int  hodor_exchange_create_direct(hodor_ctx *ctx, uint32_t n_ranks, uint32_t rank, uint32_t n_slots, hodor_exchange **out);
int  hodor_exchange_direct_flags(hodor_exchange *x, void **flags_dev_ptr, size_t *bytes);
int  hodor_exchange_direct_set_peers(hodor_exchange *x, uint32_t slot, void *const *recv, void *const *flags);
int  hodor_exchange_direct_begin_dev(hodor_exchange *x, void *stream, uint32_t slot);
int  hodor_exchange_direct_signal_dev(hodor_exchange *x, void *stream, uint32_t slot);
int  hodor_exchange_direct_wait_dev(hodor_exchange *x, void *stream, uint32_t slot);
int  hodor_exchange_direct_release_dev(hodor_exchange *x, void *stream, uint32_t slot);
/* A flag wait that gave up (~10 s without its peers) lets the stream run on: the results of that generation are UNDEFINED.
 * Call after synchronising the stream a generation ran on, before using its output: HODOR_ERR_DEVICE when any wait on
 * this handle has timed out (the handle is dead from then on: every later call returns the same). */
int  hodor_exchange_direct_status(hodor_exchange *x);
/* Copy-engine variant on the same handle and flags: the CHUNKED schedule's local send piece (written by
 * hodor_sixstep_columns_dev / _rows_dev as for hodor_sixstep_exchange_dev) is copied into the peers' mapped receive buffers
 * by n_ranks device-to-device copies, each on the stream of its destination so that they run side by side, gated and
 * collected by the handle's own stream (SDMA between devices: no CU taken from the transforms, wire time spread over
 * whatever is enqueued next).  Chunk 0 waits for the slot's release, the last chunk writes the
 * `arrived` flags; the consumer is hodor_exchange_direct_wait_dev + the plain call on the slot's receive buffer +
 * hodor_exchange_direct_release_dev.  `send` must stay valid until that wait has been enqueued. */
int  hodor_exchange_direct_copy_dev(hodor_exchange *x, void *stream, uint32_t slot, const hodor_fr *send, size_t n_local,
                                    uint32_t log_chunks, uint32_t chunk);
int  hodor_sixstep_columns_direct_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_exchange *x, uint32_t slot,
                                      uint32_t log_n1, uint32_t log_n2, const hodor_fr *omega, uint32_t log_chunks,
                                      uint32_t chunk);
int  hodor_sixstep_rows_direct_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_exchange *x, uint32_t slot,
                                   uint32_t log_n1, uint32_t log_n2, const hodor_fr *omega, uint32_t log_chunks,
                                   uint32_t chunk);
/* The receive buffers of the direct transports as allocations of the library's own (one per slot, n_local elements):
 * coarse = 0 -> FINE-GRAINED device memory (never held in this device's L2s, written through by the peers: the form
 * the memory model of DESIGN.md §6 is argued for), coarse = 1 -> plain hipMalloc (an A/B aid).  Export each with
 * hodor_ipc_export and hand every rank's pointers to hodor_exchange_direct_set_peers. */
int  hodor_exchange_direct_alloc_recv(hodor_exchange *x, size_t n_local, int coarse, void **recv /* n_slots */);

/* ---- the multi-GPU SCHEDULES inside the library (csrc/abi_dist.hip): one call per distributed transform / commit, over
 * whichever transport the handle carries.  HODOR_TRANSPORT_RCCL — RCCL's all-to-all on the library's communication
 * stream, chunked and overlapped — is THE DEFAULT of a handle with a communicator (hodor_exchange_create / _adopt) and
 * what north_star names; the two transports that map peer memory (hodor_exchange_create_direct; DIRECT is that handle's
 * default) are opt-in until a multi-GPU node has ranked the three.
 *   hodor_dist_split            the N1 x N2 split every function below uses (N1 = 2^9 from 2^19 to 2^27 points, else balanced)
 *   hodor_dist_set_transport    -1 = the handle's default; force_collectives: issue the exchange at world 1 too (testing)
 *   hodor_dist_ntt_forward_dev  layout A (src) -> layout B (dst), ONE exchange cut into 2^log_chunks overlapped pieces;
 *   hodor_dist_ntt_inverse_dev  layout B -> layout A, same omega (the library inverts it and folds n^-1 in);
 *                               n_local = 2^log_n / n_ranks elements in every buffer, src != dst
 *   hodor_dist_ntt_begin_dev /  the same in two halves (producer + exchange issued / wait + consumer), so that two
 *   hodor_dist_ntt_end_dev      INDEPENDENT transforms interleave on one stream and each exchange hides behind the
 *                               other's arithmetic; at most two in flight per handle; src must stay valid until end
 *   hodor_dist_ntt_natural_dev  natural block in, natural block out (three exchanges); inverse != 0: ifft's omega^-1, n^-1
 *   hodor_dist_lde_by_cosets_dev  Polynomial::lde / coset_lde by the reference's own coset schedule
 *                               (src/polynomials/mod.rs:418-482, :544-609): `coeffs` = all 2^log_n coefficients on every
 *                               rank, lde_block = this rank's natural block of the 2^log_n * factor values (paired != 0:
 *                               its PAIRED block — natural values [d B/2, (d+1) B/2) then N/2 + the same — for COSET2)
 *   hodor_dist_commit_dev       Blake2sIopTree::create over the ranks: subtree per rank (local_nodes: n_block x 32 bytes,
 *                               COSET2 n_block / 2 x 32), the P subtree roots exchanged (32 bytes per rank), the top
 *                               levels hashed by every rank: top = 2 P x 32 bytes, top[i] = global node i (top[1] = root)
 *   hodor_dist_lde_commit_dev   both: BASELINE config[2] over the node
 * All dist calls of one handle are serialised by the caller (one thread); buffers are the handle's own.  Like any
 * collective, every rank issues the SAME sequence of dist calls on its handle (the peer-mapped transports claim their
 * slots in call order).  Those transports need one slot per transform in flight and two slots for
 * hodor_dist_ntt_natural_dev (HODOR_ERR_INVALID otherwise — the schedule would wait for itself), and a block that fits
 * the slot's receive buffers (HODOR_ERR_SIZE).  Slots are claimed lowest-free-first and given back by the call that
 * enqueues their release (hodor_dist_ntt_end_dev for a split-phase transform), so begin / end pairs of different
 * transforms may be closed in any order.
 * On the peer-mapped transports every dist call but hodor_dist_commit_dev is stream-ordered and CANNOT see a flag wait
 * that gave up on a slow or dead peer: after synchronising the stream and BEFORE using a result, the caller MUST call
 * hodor_exchange_direct_status (HODOR_ERR_DEVICE = the result is undefined, the handle dead); hodor_dist_commit_dev /
 * hodor_dist_lde_commit_dev synchronise themselves and make that check before they hash the gathered subtree roots.
 * A dist call that fails after it has opened a generation on a slot marks the handle dead as well (its peers' waits
 * time out into the same state): destroy the handles and build new ones. */
enum { HODOR_TRANSPORT_RCCL = 0, HODOR_TRANSPORT_DIRECT = 1, HODOR_TRANSPORT_COPY = 2 };
typedef struct hodor_dist_op hodor_dist_op;
void hodor_dist_split(uint32_t log_n, uint32_t *log_n1, uint32_t *log_n2);
int  hodor_dist_set_transport(hodor_exchange *x, int transport, int force_collectives);
int  hodor_dist_ntt_forward_dev(hodor_exchange *x, void *stream, const hodor_fr *a, hodor_fr *b, size_t n_local,
                                uint32_t log_n, const hodor_fr *omega, uint32_t log_chunks);
int  hodor_dist_ntt_inverse_dev(hodor_exchange *x, void *stream, const hodor_fr *b, hodor_fr *a, size_t n_local,
                                uint32_t log_n, const hodor_fr *omega, uint32_t log_chunks);
int  hodor_dist_ntt_begin_dev(hodor_exchange *x, void *stream, const hodor_fr *src, size_t n_local, uint32_t log_n,
                              const hodor_fr *omega, int inverse, uint32_t log_chunks, hodor_dist_op **op);
int  hodor_dist_ntt_end_dev(hodor_dist_op *op, hodor_fr *dst);
int  hodor_dist_ntt_natural_dev(hodor_exchange *x, void *stream, const hodor_fr *src, hodor_fr *dst, size_t n_local,
                                uint32_t log_n, const hodor_fr *omega, int inverse);
int  hodor_dist_lde_by_cosets_dev(hodor_exchange *x, void *stream, const hodor_fr *coeffs, uint32_t log_n, size_t factor,
                                  int coset, int paired, hodor_fr *lde_block);
int  hodor_dist_commit_dev(hodor_exchange *x, void *stream, const hodor_fr *leafs_block, size_t n_block, int combiner,
                           uint8_t *local_nodes, uint8_t *top, uint8_t *root);
int  hodor_dist_lde_commit_dev(hodor_exchange *x, void *stream, const hodor_fr *coeffs, uint32_t log_n, size_t factor,
                               int coset, int combiner, hodor_fr *lde_block, uint8_t *local_nodes, uint8_t *top,
                               uint8_t *root);

/* Synthetic input for tests and benchmarks: dst[r] = element first_index + r of the
 * index-addressable SplitMix64 stream `seed` — uniform canonical residues (rejection-sampled < p)
 * in Montgomery form, i.e. what the reference's tests draw with Fr::rand (src/fft/mod.rs:71-77), but
 * reproducible on the CPU (oracle/hodor_oracle.c:o_gen_elements) and shardable across GPUs. */
int hodor_gen_elements_dev(hodor_ctx *ctx, void *stream, hodor_fr *dst, uint64_t first_index, size_t count,
                           uint64_t seed);
/* Merkle tree over n device-resident leaves into n*32 device bytes */
int hodor_iop_create_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n, uint8_t *nodes);
int hodor_iop_create_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n, int combiner,
                                  uint8_t *nodes);
/* IOP::query in the chosen format: COSET2 writes BOTH values of the coset {k, k + n/2}, k = natural_index mod n/2,
 * to values[0..2) and the path of the combined leaf (log2(n) - 1 digests); TRIVIAL = hodor_iop_query_dev. */
int hodor_iop_query_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, const uint8_t *nodes, size_t n,
                                 int combiner, size_t natural_index, hodor_fr *values, uint8_t *path, size_t *path_len);
/* IOP::query (src/iop/blake2s_trivial_iop.rs:324-338) on device-resident leaves and tree: the leaf
 * value and its authentication path (log2 n digests) are copied to the host buffers. */
int hodor_iop_query_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, const uint8_t *nodes, size_t n,
                        size_t natural_index, hodor_fr *value, uint8_t *path, size_t *path_len);
/* FRIProofPrototype::produce_proof (src/fri/query_producer.rs:10-53) against the device-resident
 * prototype: the serialised FRIProof (src/fri/mod.rs:139-147; layout documented in csrc/abi_fri.hip).
 * A COSET2 prototype writes ONE query per round — u64 index (the smaller member of the coset), the two values
 * (64 bytes), u64 path_len, path — where a TRIVIAL one writes two queries of one value each.
 * `lde_values_dev` is the codeword the prototype was committed from (device pointer).  Returns the
 * byte count; writes only when buf != NULL and cap is large enough; 0 on error. */
size_t hodor_fri_produce_proof(hodor_fri_proto *p, const hodor_fr *lde_values_dev,
                               size_t natural_first_element_index, uint8_t *buf, size_t cap);
/* NaiveFriIop::verify_proof_queries — src/fri/verifier.rs:131-289 (FriIop::verify_proof,
 * src/fri/mod.rs:96-102) over the bytes hodor_fri_produce_proof wrote.  Host-only.  *valid = 1/0 for
 * Ok(true)/Ok(false); the reference's Err(..) cases and a malformed buffer give HODOR_ERR_INVALID.
 * CAVEATS inherited from the reference, which this function mirrors step for step: it walks
 * zip(roots, queries.chunks_exact(2)) and binds neither the number of rounds, nor n_final, nor the path
 * lengths, nor output_coeffs_at_degree_plus_one to the claimed domain — a CALLER that accepts proofs from
 * an untrusted prover must itself check  n_roots == log2(initial_degree_plus_one / out_deg) + 1,
 * n_queries == 2 * n_roots, n_final == out_deg and path_len == log2(domain size of that round)
 * before calling; and, like the reference (it folds num_steps + 1 times), only proofs with
 * output_coeffs_at_degree_plus_one == 1 round-trip through produce_proof -> verify_proof. */
int hodor_fri_verify_proof(const hodor_ctx *ctx, const uint8_t *proof, size_t len, size_t natural_element_index,
                           const hodor_fr *expected_value_from_oracle, int *valid);
/* The same verifier with the proof bound to the parameters the CALLER chose before any hash is checked — what
 * the caveats above ask an integrator to do by hand.  *valid = 0 (HODOR_OK) unless the proof's lde_factor and
 * output_coeffs_at_degree_plus_one EQUAL the expected ones (a proof re-encoded with lde_factor = 1 — rate 1, every
 * function "low degree" — passes the reference's walk), initial_degree_plus_one == expected_domain_size /
 * expected_lde_factor, n_roots == log2(initial_degree_plus_one / out_deg) + 1, n_queries == 2 * n_roots,
 * n_final == out_deg and every path has the length of its round's tree; a truncated proof, which the reference's
 * walk (and hodor_fri_verify_proof) accepts, is refused here. */
int hodor_fri_verify_proof_strict(const hodor_ctx *ctx, const uint8_t *proof, size_t len, size_t expected_domain_size,
                                  size_t expected_lde_factor, size_t expected_output_coeffs_at_degree_plus_one,
                                  size_t natural_element_index, const hodor_fr *expected_value_from_oracle,
                                  int *valid);
/* Both verifiers for a proof whose trees were built by `combiner` (COSET2: one query per round, checked with one
 * path; n_queries == n_roots, path lengths log2(size) - 1). */
int hodor_fri_verify_proof_combined(const hodor_ctx *ctx, const uint8_t *proof, size_t len, int combiner,
                                    size_t natural_element_index, const hodor_fr *expected_value_from_oracle,
                                    int *valid);
int hodor_fri_verify_proof_strict_combined(const hodor_ctx *ctx, const uint8_t *proof, size_t len, int combiner,
                                           size_t expected_domain_size, size_t expected_lde_factor,
                                           size_t expected_output_coeffs_at_degree_plus_one,
                                           size_t natural_element_index,
                                           const hodor_fr *expected_value_from_oracle, int *valid);
/* NaiveFriIop::verify_prototype — src/fri/verifier.rs:10-129: the folding walk against the prover's own
 * device-resident vectors (two elements fetched per round).  *valid as above. */
int hodor_fri_verify_prototype(hodor_fri_proto *p, const hodor_fr *lde_values_dev, size_t natural_element_index,
                               int *valid);
/* FRI commit over a device-resident codeword; the prototype keeps its vectors/trees on the device */
int hodor_fri_commit_dev(hodor_ctx *ctx, void *stream, const hodor_fr *lde_values, size_t n,
                         size_t lde_factor, size_t output_coeffs_at_degree_plus_one,
                         hodor_fri_proto **out);
int hodor_fri_commit_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *lde_values, size_t n,
                                  size_t lde_factor, size_t output_coeffs_at_degree_plus_one, int combiner,
                                  hodor_fri_proto **out);
int hodor_fri_commit_through_coefficients_dev(hodor_ctx *ctx, void *stream, const hodor_fr *lde_values, size_t n,
                                              size_t lde_factor, size_t output_coeffs_at_degree_plus_one,
                                              int combiner, hodor_fri_proto **out);

/* ===================== handle API: device-resident Polynomial / IOP behind the reference's surface =====================
 * `hodor_poly` IS the reference's `Polynomial<F, P>` (src/polynomials/mod.rs:26-34: coeffs, exp, omega, omegainv, geninv,
 * minv) with `coeffs: Vec<F>` living in HBM; `hodor_iop` is `TrivialBlake2sIOP` / `Blake2sIopTree`
 * (src/iop/blake2s_trivial_iop.rs:106-339) with its `nodes` in HBM.  Every method src/arp, src/ali and src/prover call on
 * those types (:37-137 generic, :139-712 Coefficients, :715-955 Values; IOP::create / get_root / query,
 * src/iop/mod.rs:79-92) is one entry point below, so a Rust `struct Polynomial` that wraps the handle keeps the
 * callers' source unchanged and nothing but roots, evaluations, query answers and proofs crosses PCIe.
 *
 * Rules of the handles:
 *   - every operation is ENQUEUED on the context's own compute stream (hodor_ctx_stream) and returns at once; the
 *     calls that hand a result to the host (as_ref / read, evaluate_at, batch_inversion's zero check, roots, queries,
 *     prototypes) wait for it and are counted (hodor_ctx_host_round_trips);
 *   - device memory comes from a per-context pool: creating, cloning and freeing handles never calls hipMalloc /
 *     hipFree once the pool is warm (hodor_ctx_trim gives the cached blocks back to HIP);
 *   - a handle is not thread-safe (it is a `&mut self` object); different handles of one context may be used from
 *     different threads — their work is serialised on the one stream;
 *   - free every handle before hodor_ctx_destroy of its context;
 *   - the form (HODOR_FORM_*) is the Rust type parameter P: calling a Values method on a Coefficients handle is the
 *     type error the Rust compiler refuses — here HODOR_ERR_INVALID. */
typedef struct hodor_poly hodor_poly;
typedef struct hodor_iop hodor_iop;
enum { HODOR_FORM_COEFFICIENTS = 0, HODOR_FORM_VALUES = 1 };
typedef struct { uint32_t exp; hodor_fr omega, omegainv, geninv, minv; } hodor_poly_info;   /* :28-33 */

void    *hodor_ctx_stream(hodor_ctx *ctx);                    /* hipStream_t of the handle API, for `_dev` calls beside it */
uint64_t hodor_ctx_host_round_trips(const hodor_ctx *ctx);    /* device -> host results handed out since creation / reset */
void     hodor_ctx_reset_host_round_trips(hodor_ctx *ctx);       /* ... and the traffic counters below */
/* bytes the library has moved over PCIe for this context, host -> device and device -> host, since creation / reset */
void     hodor_ctx_host_traffic(const hodor_ctx *ctx, uint64_t *h2d_bytes, uint64_t *d2h_bytes);
int      hodor_ctx_trim(hodor_ctx *ctx);                      /* cached pool blocks back to HIP (synchronises the device) */
int      hodor_ctx_pool_stats(const hodor_ctx *ctx, size_t *cached_bytes, size_t *live_bytes);
size_t   hodor_ctx_pool_peak(hodor_ctx *ctx, int reset);       /* most pool bytes live at one time since creation / the last reset */

/* Polynomial::from_coeffs / from_values (:146-166, :722-742): `len` host elements, zero-padded to the next power of
 * two (Domain::new_for_size; HODOR_ERR_SIZE beyond the field's two-adicity).  new_for_size (:140-144, :716-720): zeros. */
int hodor_poly_from_host_h(hodor_ctx *ctx, int form, const hodor_fr *host, size_t len, hodor_poly **out);
int hodor_poly_new_for_size_h(hodor_ctx *ctx, int form, size_t size, hodor_poly **out);
/* the same from a device buffer produced on `producer_stream` (a hipStream_t; the copy is ordered behind it) */
int hodor_poly_from_dev_h(hodor_ctx *ctx, int form, const hodor_fr *dev_src, size_t len, void *producer_stream,
                          hodor_poly **out);
/* elements [first_index, first_index + count) of the synthetic stream `seed` (hodor_gen_elements_dev) as a polynomial */
int hodor_poly_gen_h(hodor_ctx *ctx, int form, uint64_t first_index, size_t count, uint64_t seed, hodor_poly **out);
int hodor_poly_clone_h(const hodor_poly *p, hodor_poly **out);                       /* #[derive(Clone)] :25 */
void hodor_poly_free_h(hodor_poly *p);
size_t hodor_poly_size_h(const hodor_poly *p);                                      /* size() :38 */
int hodor_poly_form_h(const hodor_poly *p);
int hodor_poly_info_h(const hodor_poly *p, hodor_poly_info *out);
void *hodor_poly_dev_ptr_h(hodor_poly *p);   /* the device buffer (size * 32 bytes); order your own work on hodor_ctx_stream */
/* as_ref() :42 — a host copy materialised on first use and kept until the polynomial is next modified; *host stays
 * valid until then (or until the handle is freed).  read: as_ref()[first .. first + count] without materialising the
 * rest; write: as_mut()[first ..] = in (:46) without materialising anything; elem_op: as_mut()[index].op(c) on the
 * device (HODOR_UN_*; e for POW). */
int hodor_poly_as_ref_h(hodor_poly *p, const hodor_fr **host);
/* as_mut() :46 for the WHOLE vector — what src/ali/per_register/mod.rs:118,139 do (`as_mut().chunks_mut(chunk)` inside a
 * worker.scope, batch_inversion between the two): *host is the polynomial's host image as `&mut [F]` (size * 32 bytes,
 * pinned from 1 MiB up), materialised like as_ref()'s — one download; none when the polynomial is still new_for_size's
 * zeros — and from this call on THE vector: the device copy is stale.  The image goes back in ONE upload when
 * hodor_poly_commit_mut_h is called (the end of the Rust borrow: a guard's Drop) or, failing that,
 * before the next operation on the handle that needs the device copy; as_ref / read / write work on the image meanwhile.
 * *host stays valid until the handle is freed or resized; writes through it after the write-back and before the next
 * hodor_poly_as_mut_h are lost (in Rust the borrow has ended and the compiler refuses them). */
int hodor_poly_as_mut_h(hodor_poly *p, hodor_fr **host);
int hodor_poly_commit_mut_h(hodor_poly *p);
int hodor_poly_read_h(hodor_poly *p, size_t first, size_t count, hodor_fr *out);
int hodor_poly_write_h(hodor_poly *p, size_t first, size_t count, const hodor_fr *in);
int hodor_poly_elem_op_h(hodor_poly *p, size_t index, int op, const hodor_fr *c, uint64_t e);
/* derive(PartialEq) :24 without moving either vector: *equal = 1 when form, size and every element agree */
int hodor_poly_equal_h(const hodor_poly *a, const hodor_poly *b, int *equal);
/* generic methods (:54-137); pad_* return HODOR_ERR_SIZE where the reference returns Err(SynthesisError::Error) */
int hodor_poly_distribute_powers_h(hodor_poly *p, const hodor_fr *g);
int hodor_poly_scale_h(hodor_poly *p, const hodor_fr *g);
int hodor_poly_negate_h(hodor_poly *p);
int hodor_poly_pad_by_factor_h(hodor_poly *p, size_t factor);
int hodor_poly_pad_to_size_h(hodor_poly *p, size_t new_size);
int hodor_poly_trim_to_degree_h(hodor_poly *p, size_t degree);
/* Coefficients -> Values in place (:611-638); the handle changes its form like the Rust value changes its type */
int hodor_poly_fft_h(hodor_poly *p);
int hodor_poly_coset_fft_h(hodor_poly *p);
int hodor_poly_coset_fft_for_generator_h(hodor_poly *p, const hodor_fr *gen);
/* Values -> Coefficients in place (:773-815) */
int hodor_poly_ifft_h(hodor_poly *p);
int hodor_poly_icoset_fft_h(hodor_poly *p);
int hodor_poly_icoset_fft_for_generator_h(hodor_poly *p, const hodor_fr *geninv);
/* lde / coset_lde (:343-349, also what filtering_lde / coset_filtering_lde :355-368, :484-499 compute): a NEW Values
 * handle of size * factor; `p` is left as it is (the Rust method consumes self: free it if you mean that).  batch:
 * all registers at once (src/prover/mod.rs:73-80) — `count` polynomials of one size; the outputs share one slab. */
int hodor_poly_lde_h(const hodor_poly *p, size_t factor, int coset, hodor_poly **out);
int hodor_poly_lde_batch_h(const hodor_poly *const *ps, size_t count, size_t factor, int coset, hodor_poly **outs);
/* add_assign / sub_assign / mul_assign / add_assign_scaled (:640-683, :817-887).  Coefficients: other may be shorter
 * (assert self.len >= other.len, the first other.len entries change); Values: equal sizes (assert_eq); mul_assign is
 * Values only.  op = HODOR_OP_*. */
int hodor_poly_binary_h(hodor_poly *a, const hodor_poly *b, int op);
int hodor_poly_add_assign_scaled_h(hodor_poly *a, const hodor_poly *b, const hodor_fr *scaling);
int hodor_poly_evaluate_at_h(hodor_poly *p, const hodor_fr *g, hodor_fr *out);     /* :685-711 (Coefficients) */
/* (coset_)evaluate_at_domain_for_degree_one (:229-290) of q(x) = c + alpha x on the size-n domain: a Values handle */
int hodor_poly_degree_one_on_domain_h(hodor_ctx *ctx, size_t n, const hodor_fr *alpha, const hodor_fr *c, int coset,
                                      hodor_poly **out);
/* hodor_poly_dense_divisor_on_coset_dev as a new Values handle of evaluation_size elements */
int hodor_poly_dense_divisor_on_coset_h(hodor_ctx *ctx, size_t evaluation_size, size_t column_size, const hodor_fr *roots,
                                        size_t n_roots, hodor_poly **out);
/* Values: pow / square / add_constant / batch_inversion (:744-771, :831-841, :889-954) */
int hodor_poly_pow_h(hodor_poly *p, uint64_t e);
int hodor_poly_square_h(hodor_poly *p);
int hodor_poly_add_constant_h(hodor_poly *p, const hodor_fr *c);
int hodor_poly_batch_inversion_h(hodor_poly *p);
/* one DEEP quotient term in one pass (hodor_poly_quotient_term_dev): acc = (accumulate ? acc : 0) + alpha (f - value) dinv */
int hodor_poly_quotient_term_h(hodor_poly *acc, const hodor_poly *f, const hodor_poly *divisor_inv, const hodor_fr *value,
                               const hodor_fr *alpha, int accumulate);

/* IOP::create(values.as_ref()) (src/iop/blake2s_trivial_iop.rs:282-300 -> Blake2sIopTree::create :131-219) over a
 * device-resident polynomial; batch: one launch sequence for all registers' oracles (src/prover/mod.rs:77-79). */
int hodor_iop_create_h(const hodor_poly *values, int combiner, hodor_iop **out);
int hodor_iop_create_batch_h(const hodor_poly *const *values, size_t count, int combiner, hodor_iop **outs);
void hodor_iop_free_h(hodor_iop *t);
size_t hodor_iop_size_h(const hodor_iop *t);                     /* number of committed values */
int hodor_iop_root_h(hodor_iop *t, uint8_t root[32]);            /* get_root :221 (32 bytes, fetched once) */
int hodor_iop_roots_h(hodor_iop *const *ts, size_t count, uint8_t *roots /* count x 32 */);   /* many roots, one wait */
int hodor_iop_nodes_h(hodor_iop *t, uint8_t *nodes);             /* the whole heap array (tests; n or n/2 entries) */
/* IOP::query(natural_index, values.as_ref()) :324-338: value(s) + path; COSET2 returns both members of the coset */
int hodor_iop_query_h(hodor_iop *t, const hodor_poly *values, size_t natural_index, hodor_fr *values_out, uint8_t *path,
                      size_t *path_len);

/* FriIop::proof_from_lde(&lde_values, ..) (src/fri/mod.rs:43-54) on a handle; through_coefficients != 0 selects
 * proof_from_lde_through_coefficients (:156-248).  prototype_into_proof / produce_proof (src/fri/query_producer.rs:10-53)
 * and verify_prototype (src/fri/verifier.rs:10-129) against the handle the prototype was committed from. */
int hodor_fri_commit_h(const hodor_poly *lde_values, size_t lde_factor, size_t output_coeffs_at_degree_plus_one,
                       int combiner, int through_coefficients, hodor_fri_proto **out);
/* proof_from_lde of several polynomials at once — h1 and h2 of Prover::prove (src/prover/mod.rs:112-113): commit 0 on the
 * context's stream, the others on auxiliary streams of the context, so that the latency-bound last rounds of one hide
 * behind the hashing of another; one wait hands all prototypes over (counted as one round trip) and the context's
 * stream continues behind all of them.  The prototypes are, byte for byte, those of `count` separate calls (by-values
 * route).  count <= 8.  The `_dev` form takes device pointers the caller has made ready on hodor_ctx_stream. */
int hodor_fri_commit_batch_h(const hodor_poly *const *lde_values, size_t count, size_t lde_factor,
                             size_t output_coeffs_at_degree_plus_one, int combiner, hodor_fri_proto **outs);
int hodor_fri_commit_batch_dev(hodor_ctx *ctx, const hodor_fr *const *lde_values, const size_t *ns, size_t count,
                               size_t lde_factor, size_t output_coeffs_at_degree_plus_one, int combiner,
                               hodor_fri_proto **outs);
size_t hodor_fri_produce_proof_h(hodor_fri_proto *p, const hodor_poly *lde_values, size_t natural_first_element_index,
                                 uint8_t *buf, size_t cap);
int hodor_fri_verify_prototype_h(hodor_fri_proto *p, const hodor_poly *lde_values, size_t natural_element_index,
                                 int *valid);
/* l0_commitment (step = -1) / intermediate_commitments[step] (src/fri/mod.rs:107-109) as an IOP object: a VIEW of the
 * prototype's tree (valid while the prototype lives, freed with hodor_iop_free_h; the root needs no round trip) */
int hodor_fri_commitment_h(hodor_fri_proto *p, int step, hodor_iop **out);
/* intermediate_values[step] (src/fri/mod.rs:110) as a Values handle of its own (a device copy) */
int hodor_fri_intermediate_values_h(hodor_fri_proto *p, size_t step, hodor_poly **out);

#ifdef __cplusplus
}
#endif
#endif /* HODOR_GPU_H */
