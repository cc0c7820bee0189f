// abi_fri.hip — the FRI entry points of include/hodor_gpu.h: commit (device and slice), query phase,
// proof bytes, the two verifiers and the prototype accessors.
#include "ctx.hpp"

// The prototype's device buffers (l0 tree, every intermediate vector and tree, the small result block) are carved
// from ONE slab out of the context's device pool (abi_poly.hip): a prover committing polynomial after polynomial
// meets hipMalloc once per size.  (Rounds 1-4 parked one freed slab on the context and paid hipMalloc + hipFree for
// every other: 6 GB allocated while the queue is busy stalled the free-running proof run of round 5 by ~190 ms.)
// A commit may run on ANY stream (the `_dev` entry points take the caller's): the pool orders the slab's new user behind
// its old one with an event (pool_alloc's `consumer`), and every entry point that enqueues work on the slab on a
// caller's stream synchronises before it returns, so that at hodor_fri_free the only work that can still be pending on
// the slab is the handle API's (ctx->stream) — the stream pool_release records its event on.
namespace {
// a staging block from the context's device pool (no hipMalloc / hipFree — which drains the device — per query); taken
// for `stream` (ordered behind the block's previous user) and given back once `stream` has run dry
struct PoolBuf {
    hodor_ctx *ctx;
    hipStream_t stream;
    void *p = nullptr;
    size_t got = 0;
    bool drained = false;
    PoolBuf(hodor_ctx *c, hipStream_t s) : ctx(c), stream(s) {}
    int alloc(size_t bytes) { return pool_alloc(ctx, bytes, &p, &got, stream); }
    ~PoolBuf()
    {
        if (!p) return;
        if (!drained) (void)hipStreamSynchronize(stream);
        pool_release(ctx, p, got);
    }
};
}  // namespace

extern "C" void hodor_fri_free(hodor_fri_proto *p)
{
    if (!p) return;
    hodor_ctx *ctx = p->ctx;
    if (ctx && p->slab) pool_release(ctx, p->slab, p->slab_bytes);
    if (ctx) ctx->live_handles.fetch_sub(1);
    delete p;
}

extern "C" int hodor_fri_commit_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *lde_values, size_t n,
                                    size_t lde_factor, size_t out_deg, hodor_fri_proto **out)
{
    return hodor_fri_commit_combined_dev(ctx, stream_, lde_values, n, lde_factor, out_deg, HODOR_COMBINER_TRIVIAL, out);
}

// One by-values commit in two halves, so that several commits can be in the queue at once (hodor_fri_commit_batch_h):
//   fri_commit_enqueue   everything the commit launches, on `stream`, nothing waited for; the prototype it returns has its
//                        device side complete in stream order and its host fields (roots, challenges, final
//                        coefficients) still empty
//   fri_commit_collect   those fields copied out of the slab's small block (the caller has made `xfer`'s stream wait for
//                        the commit's) — after xfer.finish() the prototype is what hodor_fri_commit_combined_dev returns
// Caller holds ctx->mu.
namespace {
struct FriPending {
    hodor_fri_proto *p = nullptr;
    uint8_t *d_small = nullptr;
    std::vector<uint8_t> small;
};

int fri_commit_enqueue(hodor_ctx *ctx, hipStream_t stream, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                       size_t out_deg, int combiner, FriPending *pend)
{
    if (combiner != HODOR_COMBINER_TRIVIAL && combiner != HODOR_COMBINER_COSET2) return HODOR_ERR_INVALID;
    const bool comb = combiner == HODOR_COMBINER_COSET2;
    if (!is_pow2(n) || !is_pow2(lde_factor) || !is_pow2(out_deg) || n < 2) return HODOR_ERR_SIZE;
    size_t initial_degree_plus_one = n / lde_factor;
    if (initial_degree_plus_one < 2 * out_deg) {   // num_steps == 0: the reference panics at roots.pop() (:124)
        set_err(ctx, "fri_commit: needs at least one folding step");
        return HODOR_ERR_SIZE;
    }
    size_t num_steps = log2u(initial_degree_plus_one / out_deg);
    if ((n >> num_steps) < (comb ? 4u : 2u)) {   // the smallest committed vector: a tree needs two leaves
        set_err(ctx, comb ? "fri_commit: COSET2 needs lde_factor * out_deg >= 4 (two combined leaves in the last tree)"
                          : "fri_commit: the last tree needs two leaves");
        return HODOR_ERR_SIZE;
    }
    uint32_t log_n = log2u(n);
    HFr omega, omega_inv;
    int rc = poly_domain(ctx, log_n, &omega);
    if (rc) return rc;
    ctx->F.inverse(omega, &omega_inv);

    hodor_fri_proto *p = new (std::nothrow) hodor_fri_proto();
    if (!p) return HODOR_ERR_INVALID;
    p->ctx = ctx;
    p->n = n;
    p->num_steps = num_steps;
    p->lde_factor = lde_factor;
    p->out_deg = out_deg;
    p->initial_degree_plus_one = initial_degree_plus_one;
    p->combiner = combiner;

#define FRICHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                   \
            if (e__ != hipErrorAssert) set_err(ctx, std::string(#expr) + ": " + hipGetErrorString(e__));   \
            fri_release(p);                                                            \
            return HODOR_ERR_DEVICE;                                                   \
        }                                                                              \
    } while (0)

    // slab layout: l0 tree | per step: values, tree | small block (challenges, roots, final coeffs)
    size_t fin_n = n >> num_steps;
    size_t small_bytes = 32 * (num_steps + 1) * 2 + 32 * fin_n * 2;
    const uint32_t winv_lo_bits = (log_n + 1) / 2;
    const size_t hi_cnt = (size_t)1 << (log_n - winv_lo_bits);
    small_bytes += 48 * hi_cnt;   // per-round copy of the w^-1 `hi` table scaled by beta/2
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t need = up(n * 32) + up(small_bytes);
    for (size_t i = 0, sz = n / 2; i < num_steps; i++, sz >>= 1) need += 2 * up(sz * 32);
    auto fri_release = [&](hodor_fri_proto *q) {   // error path: the ctx mutex is already held
        if (q->slab) {
            (void)hipStreamSynchronize(stream);    // whatever was enqueued on the slab before the error
            pool_release(ctx, q->slab, q->slab_bytes);
        }
        delete q;
    };
    if ((rc = pool_alloc(ctx, need, &p->slab, &p->slab_bytes, stream))) { delete p; return rc; }
    uint8_t *cursor = (uint8_t *)p->slab;
    auto carve = [&](size_t b) { uint8_t *r = cursor; cursor += up(b); return (void *)r; };
    p->l0_nodes = carve(n * 32);
    for (size_t i = 0, sz = n / 2; i < num_steps; i++, sz >>= 1) {
        p->inter_values.push_back(carve(sz * 32));
        p->inter_nodes.push_back(carve(sz * 32));
        p->inter_sizes.push_back(sz);
    }
    uint8_t *d_small = (uint8_t *)carve(small_bytes);
    uint4 *d_chal = (uint4 *)d_small;
    uint4 *d_roots = (uint4 *)(d_small + 32 * (num_steps + 1));
    uint4 *d_fin = (uint4 *)(d_small + 64 * (num_steps + 1));
    uint4 *d_hi_beta = (uint4 *)(d_small + 64 * (num_steps + 1) + 64 * fin_n);
    const Fr9 c16 = to_dev9(ctx->F, ctx->F.from_u64(16));

    TwoLevel winv;
    if ((rc = get_pow_table(ctx, omega_inv, log_n, &winv, 1, winv_lo_bits))) { fri_release(p); return rc; }
    uint32_t shave = 256 - ctx->F.capacity;
    Fr r2 = to_dev(ctx->F.r2);

    FRICHK(merkle_build_launch(stream, (const uint4 *)lde_values, (uint4 *)p->l0_nodes, n, ctx->mid, 1, nullptr,
                               nullptr, comb));                                                         // :17

    const uint4 *values = (const uint4 *)lde_values;
    size_t next_size = n / 2;
    const int tail_on = knobs().fri_tail;   // HODOR_FRI_TAIL=0: every round through the multi-launch path (A/B, debugging)
    // the challenge of round i (:51, :109) is derived from tree i-1 at the start of round i
    const uint4 *prev_nodes = (const uint4 *)p->l0_nodes;
    bool tail_done = false;
    for (size_t i = 0; i < num_steps; i++) {                                                             // :61
        if (tail_on && next_size <= (size_t)FRI_TAIL_THREADS && num_steps - i <= (size_t)FRI_TAIL_MAX_ROUNDS) {
            FriTailArgs T = {};   // the remaining rounds fit one workgroup: one launch for all of them
            FRICHK(challenge_launch(stream, prev_nodes, d_chal + 2 * i, d_roots + 2 * i, r2, shave, ctx->P));
            T.src = values;
            T.rounds = (uint32_t)(num_steps - i);
            for (uint32_t k = 0; k < T.rounds; k++) {
                T.values[k] = (uint4 *)p->inter_values[i + k];
                T.nodes[k] = (uint4 *)p->inter_nodes[i + k];
            }
            T.chal = d_chal;
            T.roots = d_roots;
            T.lo = winv.lo;
            T.hi = winv.hi;
            T.lo_bits = winv.lo_bits;
            T.first_round = (uint32_t)i;
            T.half0 = (uint32_t)next_size;
            T.shave = shave;
#ifdef HODOR_BOUNDS
            T.lo_bytes = winv.lo_bytes;
            T.hi_bytes = winv.hi_bytes;
#endif
            FRICHK(fri_tail_launch(stream, T, c16, r2, ctx->mid, ctx->Q, ctx->P, comb));
            values = (const uint4 *)p->inter_values[num_steps - 1];
            tail_done = true;
            break;
        }
        void *next = p->inter_values[i], *nodes = p->inter_nodes[i];
        FRICHK(fri_round_table_launch(stream, prev_nodes, d_chal + 2 * i, d_roots + 2 * i, winv.hi, d_hi_beta, hi_cnt,
                                      c16, r2, shave, ctx->Q, ctx->P));
        FoldArgs fold = {values, (uint4 *)next, next_size, winv.lo, d_hi_beta, winv.lo_bits, (uint32_t)i};
#ifdef HODOR_BOUNDS
        fold.lo_bytes = winv.lo_bytes;
        fold.hi_bytes = 48 * hi_cnt;
#endif
        const bool fused = merkle_fuses_fold(next_size);   // small rounds: fold inside the tree's leaf launch
        if (!fused) FRICHK(fri_fold_launch(stream, fold, ctx->Q));                                        // :70-104
        FRICHK(merkle_build_launch(stream, (const uint4 *)next, (uint4 *)nodes, next_size, ctx->mid, 1,
                                   fused ? &fold : nullptr, &ctx->Q, comb));                             // :106
        prev_nodes = (const uint4 *)nodes;
        values = (const uint4 *)next;
        next_size >>= 1;
    }
    if (!tail_done)   // the last tree's root (and the challenge the reference computes and pops, :120)
        FRICHK(challenge_launch(stream, prev_nodes, d_chal + 2 * num_steps, d_roots + 2 * num_steps, r2, shave, ctx->P));
    // final: values -> ifft -> truncate (:130-145)
    rc = poly_transform(ctx, stream, values, d_fin, log2u(fin_n), OP_IFFT);
    if (rc) { fri_release(p); return rc; }
#undef FRICHK
    pend->p = p;
    pend->d_small = d_small;
    pend->small.resize(64 * (num_steps + 1) + 32 * out_deg);   // challenges | roots | final coefficients: back to back in the small block
    return HODOR_OK;
}

// the copy of the small block into the caller's xfer (whose stream is ordered behind the commit's) ...
hipError_t fri_commit_fetch(FriPending *pend, HostXfer &xfer) { return xfer.d2h(pend->small.data(), pend->d_small, pend->small.size()); }

// ... and, after xfer.finish(), the host fields
void fri_commit_finish(hodor_ctx *ctx, FriPending *pend)
{
    hodor_fri_proto *p = pend->p;
    const size_t num_steps = p->num_steps;
    note_round_trip(ctx);
    p->roots.assign(pend->small.begin() + 32 * (num_steps + 1), pend->small.begin() + 64 * (num_steps + 1));
    p->challenges.resize(num_steps);
    memcpy(p->challenges.data(), pend->small.data(), 32 * num_steps);
    p->final_coeffs.resize(p->out_deg);
    memcpy(p->final_coeffs.data(), pend->small.data() + 64 * (num_steps + 1), 32 * p->out_deg);
    memcpy(p->final_root, p->roots.data() + 32 * num_steps, 32);   // roots.pop() :124
    ctx->live_handles.fetch_add(1);
}

void fri_commit_abandon(hodor_ctx *ctx, FriPending *pend, hipStream_t stream)
{
    if (!pend->p) return;
    (void)hipStreamSynchronize(stream);
    if (pend->p->slab) pool_release(ctx, pend->p->slab, pend->p->slab_bytes);
    delete pend->p;
    pend->p = nullptr;
}
}  // namespace

extern "C" int hodor_fri_commit_combined_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *lde_values, size_t n,
                                             size_t lde_factor, size_t out_deg, int combiner, hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t stream = pick_stream(ctx, stream_);
    FriPending pend;
    int rc = fri_commit_enqueue(ctx, stream, lde_values, n, lde_factor, out_deg, combiner, &pend);
    if (rc) return rc;
    hipError_t e;
    {
        HostXfer xfer(ctx, stream);      // through the context's pinned buffer (ctx.hpp): no heap page is pinned behind our back
        e = fri_commit_fetch(&pend, xfer);
        if (e == hipSuccess) e = xfer.finish();
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (e != hipErrorAssert) set_err(ctx, std::string("fri_commit: ") + hipGetErrorString(e));
        fri_commit_abandon(ctx, &pend, stream);
        return HODOR_ERR_DEVICE;
    }
    fri_commit_finish(ctx, &pend);
    *out = pend.p;
    return HODOR_OK;
}

// FriIop::proof_from_lde of SEVERAL polynomials at once — h1 and h2 of Prover::prove (src/prover/mod.rs:112-113: two
// independent commits back to back).  A commit's last rounds are latency-bound (a dozen launches of a workgroup or
// two each, the chip idle around them); issued one after the other on one stream the two tails add up.  Here commit 0
// runs on the context's stream and commit i > 0 on an auxiliary stream of the context (ordered behind everything the
// handles have enqueued so far), so the tail of one hides behind the hashing of the other; ONE wait and one copy hand
// all the prototypes' roots / challenges / final coefficients to the host, and the context's stream continues behind
// all of them.  Same prototypes, byte for byte, as `count` calls of hodor_fri_commit_h.
extern "C" int hodor_fri_commit_batch_dev(hodor_ctx *ctx, const hodor_fr *const *lde_values, const size_t *ns, size_t count,
                                          size_t lde_factor, size_t out_deg, int combiner, hodor_fri_proto **outs)
{
    NEED_DEVICE();
    if (!lde_values || !ns || !outs || count == 0 || count > 8) return HODOR_ERR_INVALID;
    for (size_t i = 0; i < count; i++)
        if (!lde_values[i]) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::vector<FriPending> pend(count);
    std::vector<hipStream_t> streams(count, ctx->stream);
    int rc = HODOR_OK;
    hipError_t e = hipSuccess;
    hipEvent_t fork = nullptr;
    if (count > 1) {
        e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(fork, ctx->stream);
        for (size_t i = 1; i < count && e == hipSuccess; i++) {
            if (!ctx->aux_streams[i - 1]) e = hipStreamCreateWithFlags(&ctx->aux_streams[i - 1], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->aux_streams[i - 1], fork, 0);   // the inputs are the handles' (ctx->stream)
            streams[i] = ctx->aux_streams[i - 1];
        }
        if (fork) (void)hipEventDestroy(fork);
    }
    size_t enq = 0;
    // the larger commits first: their hashing is what the others' tails hide behind
    std::vector<size_t> order(count);
    for (size_t i = 0; i < count; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ns[a] > ns[b]; });
    for (; e == hipSuccess && rc == HODOR_OK && enq < count; enq++) {
        const size_t i = order[enq];
        rc = fri_commit_enqueue(ctx, streams[i], lde_values[i], ns[i], lde_factor, out_deg, combiner, &pend[i]);
    }
    // join: the context's stream behind every auxiliary stream, then one copy of all the small blocks
    for (size_t i = 1; i < count && e == hipSuccess; i++) {
        hipEvent_t join = nullptr;
        e = hipEventCreateWithFlags(&join, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(join, streams[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, join, 0);
        if (join) (void)hipEventDestroy(join);
    }
    if (e == hipSuccess && rc == HODOR_OK) {
        HostXfer xfer(ctx, ctx->stream);
        for (size_t i = 0; i < count && e == hipSuccess; i++) e = fri_commit_fetch(&pend[i], xfer);
        if (e == hipSuccess) e = xfer.finish();
    }
    if (e != hipSuccess || rc != HODOR_OK) {
        (void)hipGetLastError();
        if (e != hipSuccess && e != hipErrorAssert) set_err(ctx, std::string("fri_commit (batch): ") + hipGetErrorString(e));
        for (size_t i = 0; i < count; i++) fri_commit_abandon(ctx, &pend[i], streams[i]);
        return rc ? rc : HODOR_ERR_DEVICE;
    }
    for (size_t i = 0; i < count; i++) {
        fri_commit_finish(ctx, &pend[i]);
        outs[i] = pend[i].p;
    }
    ctx->host_round_trips.fetch_sub(count - 1);   // one wait handed all of them over
    return HODOR_OK;
}

// NaiveFriIop::proof_from_lde_through_coefficients (src/fri/mod.rs:156-248) on the device: l0 commit (:162), ONE
// inverse transform of the codeword (:171; truncation :173 = the first n / lde_factor entries of its output), then
// per round the challenge from the previous tree's root (:178, :213 — k_challenge, never on the host), the
// coefficient fold a_2i + beta a_(2i+1) (:190-205, k_fri_fold_coeffs), Polynomial::lde of the folded coefficients
// (:208-209: one zero-padded transform, = the reference's per-coset schedule) and the tree over it (:210).  The
// prototype has the layout of the by-values one (same accessors, same bytes — the reference asserts the two equal,
// :338-343); the coefficient ping-pong buffers live in the same slab, behind the small block.
extern "C" int hodor_fri_commit_through_coefficients_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *lde_values,
                                                         size_t n, size_t lde_factor, size_t out_deg, int combiner,
                                                         hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    if (combiner != HODOR_COMBINER_TRIVIAL && combiner != HODOR_COMBINER_COSET2) return HODOR_ERR_INVALID;
    const bool comb = combiner == HODOR_COMBINER_COSET2;
    if (!is_pow2(n) || !is_pow2(lde_factor) || !is_pow2(out_deg) || n < 2) return HODOR_ERR_SIZE;   // asserts :165-166
    const size_t initial_degree_plus_one = n / lde_factor;                                        // :168
    if (initial_degree_plus_one < 2 * out_deg) {   // num_steps == 0: the reference panics at roots.pop() (:226)
        set_err(ctx, "fri_commit_through_coefficients: needs at least one folding step");
        return HODOR_ERR_SIZE;
    }
    const size_t num_steps = log2u(initial_degree_plus_one / out_deg);                            // :169
    if ((n >> num_steps) < (comb ? 4u : 2u)) {
        set_err(ctx, comb ? "fri_commit_through_coefficients: COSET2 needs lde_factor * out_deg >= 4"
                          : "fri_commit_through_coefficients: the last tree needs two leaves");
        return HODOR_ERR_SIZE;
    }
    const uint32_t log_n = log2u(n);
    HFr omega;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = poly_domain(ctx, log_n, &omega);
    if (rc) return rc;
    hipStream_t stream = pick_stream(ctx, stream_);

    hodor_fri_proto *p = new (std::nothrow) hodor_fri_proto();
    if (!p) return HODOR_ERR_INVALID;
    p->ctx = ctx;
    p->n = n;
    p->num_steps = num_steps;
    p->lde_factor = lde_factor;
    p->out_deg = out_deg;
    p->initial_degree_plus_one = initial_degree_plus_one;
    p->combiner = combiner;
    auto release = [&](hodor_fri_proto *q) {   // error path: the ctx mutex is already held
        if (q->slab) {
            (void)hipStreamSynchronize(stream);
            pool_release(ctx, q->slab, q->slab_bytes);
        }
        delete q;
    };
#define FRICHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                   \
            if (e__ != hipErrorAssert) set_err(ctx, std::string(#expr) + ": " + hipGetErrorString(e__));   \
            release(p);                                                                \
            return HODOR_ERR_DEVICE;                                                   \
        }                                                                              \
    } while (0)

    // slab: l0 tree | per step: values, tree | small block (challenges, roots) | coefficients A (n) | B (deg / 2)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t small_bytes = 64 * (num_steps + 1);
    size_t need = up(n * 32) + up(small_bytes) + up(n * 32) + up(initial_degree_plus_one * 16);
    for (size_t i = 0, sz = n / 2; i < num_steps; i++, sz >>= 1) need += 2 * up(sz * 32);
    if ((rc = pool_alloc(ctx, need, &p->slab, &p->slab_bytes, stream))) { delete p; return rc; }
    uint8_t *cursor = (uint8_t *)p->slab;
    auto carve = [&](size_t b) { uint8_t *r = cursor; cursor += up(b); return (void *)r; };
    p->l0_nodes = carve(n * 32);
    for (size_t i = 0, sz = n / 2; i < num_steps; i++, sz >>= 1) {
        p->inter_values.push_back(carve(sz * 32));
        p->inter_nodes.push_back(carve(sz * 32));
        p->inter_sizes.push_back(sz);
    }
    uint8_t *d_small = (uint8_t *)carve(small_bytes);
    uint4 *d_chal = (uint4 *)d_small;
    uint4 *d_roots = (uint4 *)(d_small + 32 * (num_steps + 1));
    uint4 *coef_a = (uint4 *)carve(n * 32);
    uint4 *coef_b = (uint4 *)carve(initial_degree_plus_one * 16);
    const uint32_t shave = 256 - ctx->F.capacity;
    const Fr r2 = to_dev(ctx->F.r2);

    FRICHK(merkle_build_launch(stream, (const uint4 *)lde_values, (uint4 *)p->l0_nodes, n, ctx->mid, 1, nullptr, nullptr,
                               comb));                                                                       // :162
    if ((rc = poly_transform(ctx, stream, (const uint4 *)lde_values, coef_a, log_n, OP_IFFT))) { release(p); return rc; }   // :171
    const uint4 *coeffs = coef_a;                      // its first initial_degree_plus_one entries (:173)
    uint4 *spare = coef_b;
    const uint4 *prev_nodes = (const uint4 *)p->l0_nodes;
    size_t next_len = initial_degree_plus_one / 2;                                                           // :180
    for (size_t i = 0; i < num_steps; i++, next_len >>= 1) {                                                 // :185
        FRICHK(challenge_launch(stream, prev_nodes, d_chal + 2 * i, d_roots + 2 * i, r2, shave, ctx->P));    // :178 / :213
        FRICHK(fri_fold_coeffs_launch(stream, coeffs, spare, next_len, d_chal + 2 * i, ctx->P));             // :190-205
        if ((rc = poly_lde_exec(ctx, stream, spare, (uint4 *)p->inter_values[i], log2u(next_len), lde_factor, 0))) {   // :208-209
            release(p);
            return rc;
        }
        FRICHK(merkle_build_launch(stream, (const uint4 *)p->inter_values[i], (uint4 *)p->inter_nodes[i],
                                   next_len * lde_factor, ctx->mid, 1, nullptr, nullptr, comb));             // :210
        prev_nodes = (const uint4 *)p->inter_nodes[i];
        uint4 *t = (uint4 *)coeffs;                    // the buffer just consumed takes the next round's output
        coeffs = spare;
        spare = t;
    }
    FRICHK(challenge_launch(stream, prev_nodes, d_chal + 2 * num_steps, d_roots + 2 * num_steps, r2, shave, ctx->P));   // the root :212; its challenge is popped :224

    std::vector<uint8_t> small(small_bytes);
    p->final_coeffs.resize(out_deg);                                                                         // :232-234
    {
        HostXfer xfer(ctx, stream);
        FRICHK(xfer.d2h(small.data(), d_small, small_bytes));
        FRICHK(xfer.d2h(p->final_coeffs.data(), coeffs, 32 * out_deg));
        FRICHK(xfer.finish());
    }
    note_round_trip(ctx);
    p->roots.assign(small.begin() + 32 * (num_steps + 1), small.end());
    p->challenges.resize(num_steps);
    memcpy(p->challenges.data(), small.data(), 32 * num_steps);
    memcpy(p->final_root, p->roots.data() + 32 * num_steps, 32);                                             // roots.pop() :226
#undef FRICHK
    ctx->live_handles.fetch_add(1);
    *out = p;
    return HODOR_OK;
}

extern "C" int hodor_fri_commit_through_coefficients(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n,
                                                     size_t lde_factor, size_t out_deg, int combiner,
                                                     hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) return HODOR_ERR_SIZE;
    DevBuf dv;
    HIPCHK(dev_malloc(&dv.p, n * 32));
    BOUNDS_NOTE(dv.p, n * 32);
    if (int rc_up = hodor_buf_upload(ctx, dv.p, lde_values, n * 32)) return rc_up;   // small codewords through the pinned buffer
    return hodor_fri_commit_through_coefficients_dev(ctx, (void *)ctx->stream, (const hodor_fr *)dv.p, n, lde_factor,
                                                     out_deg, combiner, out);
}

extern "C" int hodor_fri_commit(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                                size_t out_deg, hodor_fri_proto **out)
{
    return hodor_fri_commit_combined(ctx, lde_values, n, lde_factor, out_deg, HODOR_COMBINER_TRIVIAL, out);
}

extern "C" int hodor_fri_commit_combined(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                                         size_t out_deg, int combiner, hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) return HODOR_ERR_SIZE;
    DevBuf dv;
    HIPCHK(dev_malloc(&dv.p, n * 32));
    BOUNDS_NOTE(dv.p, n * 32);
    if (int rc_up = hodor_buf_upload(ctx, dv.p, lde_values, n * 32)) return rc_up;   // small codewords through the pinned buffer
    // the context's own compute stream, like every other slice entry point: in-order with the other
    // callers' transforms that share ctx->scratch (hodor_fri_commit_dev holds ctx->mu until it has
    // synchronised), never the legacy NULL stream, which has no ordering with a non-blocking stream
    return hodor_fri_commit_combined_dev(ctx, (void *)ctx->stream, (const hodor_fr *)dv.p, n, lde_factor, out_deg,
                                         combiner, out);
}

// IOP::query on device-resident leaves and tree (src/iop/blake2s_trivial_iop.rs:324-338)
extern "C" int hodor_iop_query_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *leafs, const uint8_t *nodes,
                                   size_t n, size_t natural_index, hodor_fr *value, uint8_t *path,
                                   size_t *path_len)
{
    NEED_DEVICE();
    if (!leafs || !nodes || !value || !path || !path_len) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2 || natural_index >= n) return HODOR_ERR_SIZE;   // asserts at :325-326
    hipStream_t stream = pick_stream(ctx, stream_);
    size_t entries = log2u(n) + 1;
    PoolBuf stage(ctx, stream);
    if (int rc_stage = stage.alloc(entries * 32)) return rc_stage;
    HIPCHK(iop_query_launch(stream, (const uint4 *)(leafs + (natural_index & ~(size_t)1)), (const uint4 *)nodes, n,
                            natural_index, (uint4 *)stage.p, ctx->mid));
    std::vector<uint8_t> host(entries * 32);
    {
        HostXfer xfer(ctx, stream);
        HIPCHK(xfer.d2h(host.data(), stage.p, entries * 32));
        HIPCHK(xfer.finish());
        stage.drained = true;
    }
    note_round_trip(ctx);
    memcpy(value, host.data(), 32);
    memcpy(path, host.data() + 32, (entries - 1) * 32);
    *path_len = entries - 1;
    return HODOR_OK;
}

// IOP::query on a tree built by `combiner`: COSET2 returns both values of the coset {k, k + n/2}, k = index mod n/2
// (values[0] at k, values[1] at k + n/2) and the path of the combined leaf (log2(n) - 1 digests); TRIVIAL returns
// values[0] only and the reference's path.
extern "C" int hodor_iop_query_combined_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *leafs, const uint8_t *nodes,
                                            size_t n, int combiner, size_t natural_index, hodor_fr *values,
                                            uint8_t *path, size_t *path_len)
{
    if (combiner == HODOR_COMBINER_TRIVIAL)
        return hodor_iop_query_dev(ctx, stream_, leafs, nodes, n, natural_index, values, path, path_len);
    NEED_DEVICE();
    if (combiner != HODOR_COMBINER_COSET2 || !leafs || !nodes || !values || !path || !path_len) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4 || natural_index >= n) return HODOR_ERR_SIZE;
    hipStream_t stream = pick_stream(ctx, stream_);
    size_t entries = log2u(n) + 1;                 // 2 values + (log2(n) - 1) digests
    PoolBuf stage(ctx, stream);
    if (int rc_stage = stage.alloc(entries * 32)) return rc_stage;
    HIPCHK(iop_query_coset2_launch(stream, (const uint4 *)leafs, (const uint4 *)nodes, n, natural_index,
                                   (uint4 *)stage.p, ctx->mid));
    std::vector<uint8_t> host(entries * 32);
    {
        HostXfer xfer(ctx, stream);
        HIPCHK(xfer.d2h(host.data(), stage.p, entries * 32));
        HIPCHK(xfer.finish());
        stage.drained = true;
    }
    note_round_trip(ctx);
    memcpy(values, host.data(), 64);
    memcpy(path, host.data() + 64, (entries - 2) * 32);
    *path_len = entries - 2;
    return HODOR_OK;
}

// FRIProofPrototype::produce_proof (src/fri/query_producer.rs:10-53): for the l0 oracle and every
// intermediate oracle, the two queries of the coset {idx, idx + size/2} (sorted), idx halving as the
// domain does (Domain::index_and_size_for_next_domain).  Serialised FRIProof (src/fri/mod.rs:139-147):
//   u64 num_queries | per query: u64 natural_index, value (32 B), u64 path_len, path |
//   u64 num_roots | roots | u64 n_final | final_coefficients |
//   u64 initial_degree_plus_one | u64 output_coeffs_at_degree_plus_one | u64 lde_factor
extern "C" size_t hodor_fri_produce_proof(hodor_fri_proto *p, const hodor_fr *lde_values_dev,
                                          size_t natural_first_element_index, uint8_t *buf, size_t cap)
{
    if (!p || !lde_values_dev) return 0;
    hodor_ctx *ctx = p->ctx;
    if (!ctx || ctx->device < 0 || natural_first_element_index >= p->n) return 0;
    const size_t rounds = p->num_steps + 1;
    const bool comb = p->combiner == HODOR_COMBINER_COSET2;
    size_t need = 8, stage_bytes = 0;
    for (size_t r = 0, sz = p->n; r < rounds; r++, sz >>= 1) {
        if (comb) {   // ONE query per round: index, both values of the coset, a path of log2(sz) - 1 digests
            size_t plen = log2u(sz) - 1;
            need += 8 + 64 + 8 + plen * 32;
            stage_bytes += (2 + plen) * 32;
        } else {
            size_t entries = log2u(sz) + 1;
            need += 2 * (8 + 32 + 8 + (entries - 1) * 32);
            stage_bytes += 2 * entries * 32;
        }
    }
    need += 8 + rounds * 32 + 8 + p->out_deg * 32 + 24;
    if (!buf || cap < need) return need;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    PoolBuf stage(ctx, ctx->stream);
    if (stage.alloc(stage_bytes)) return 0;
    std::vector<size_t> q_index, q_entries;
    size_t domain_size = p->n, domain_idx = natural_first_element_index, off = 0;
    for (size_t r = 0; r < rounds; r++) {
        const hodor_fr *leafs = r == 0 ? lde_values_dev : (const hodor_fr *)p->inter_values[r - 1];
        const uint4 *nodes = (const uint4 *)(r == 0 ? p->l0_nodes : p->inter_nodes[r - 1]);
        size_t pair = (domain_idx + domain_size / 2) % domain_size;
        size_t coset[2] = {domain_idx < pair ? domain_idx : pair, domain_idx < pair ? pair : domain_idx};
        size_t entries = log2u(domain_size) + 1;
        if (comb) {
            if (iop_query_coset2_launch(ctx->stream, (const uint4 *)leafs, nodes, domain_size, coset[0],
                                        (uint4 *)((uint8_t *)stage.p + off), ctx->mid) != hipSuccess)
                return 0;
            q_index.push_back(coset[0]);
            q_entries.push_back(entries);            // 2 values + (log2 - 1) digests = log2 + 1 entries
            off += entries * 32;
        }
        for (int k = 0; k < 2 && !comb; k++) {
            if (iop_query_launch(ctx->stream, (const uint4 *)(leafs + (coset[k] & ~(size_t)1)), nodes, domain_size,
                                 coset[k], (uint4 *)((uint8_t *)stage.p + off), ctx->mid) != hipSuccess)
                return 0;
            q_index.push_back(coset[k]);
            q_entries.push_back(entries);
            off += entries * 32;
        }
        size_t next = domain_size / 2;                       // index_and_size_for_next_domain
        domain_idx = domain_idx < next ? domain_idx : domain_idx - next;
        domain_size = next;
    }
    std::vector<uint8_t> host(stage_bytes);
    {
        HostXfer xfer(ctx, ctx->stream);
        if (xfer.d2h(host.data(), stage.p, stage_bytes) != hipSuccess || xfer.finish() != hipSuccess) return 0;
        stage.drained = true;
    }
    note_round_trip(ctx);
    size_t o = 0, h = 0;
    auto put64 = [&](uint64_t v) { memcpy(buf + o, &v, 8); o += 8; };
    put64(q_index.size());
    const size_t vals = comb ? 2 : 1;   // values carried by one query
    for (size_t q = 0; q < q_index.size(); q++) {
        put64(q_index[q]);
        memcpy(buf + o, host.data() + h, 32 * vals); o += 32 * vals;
        put64(q_entries[q] - vals);
        memcpy(buf + o, host.data() + h + 32 * vals, (q_entries[q] - vals) * 32); o += (q_entries[q] - vals) * 32;
        h += q_entries[q] * 32;
    }
    put64(rounds);
    memcpy(buf + o, p->roots.data(), rounds * 32); o += rounds * 32;
    put64(p->out_deg);
    memcpy(buf + o, p->final_coeffs.data(), p->out_deg * 32); o += p->out_deg * 32;
    put64(p->initial_degree_plus_one);
    put64(p->out_deg);
    put64(p->lde_factor);
    return o;
}

// NaiveFriIop::verify_proof_queries (src/fri/verifier.rs:131-289) over the serialised FRIProof that
// hodor_fri_produce_proof writes.  Host-only (log n hashes and a handful of field operations per
// round): works on a ctx without a device.  *valid = Ok(true/false); the reference's Err(..) cases
// (point outside the LDE domain, query count not a multiple of DEGREE = 2, wrong tree index) and a
// malformed buffer return HODOR_ERR_INVALID.
extern "C" int hodor_fri_verify_proof(const hodor_ctx *ctx, const uint8_t *proof, size_t len,
                                      size_t natural_element_index, const hodor_fr *expected_value_from_oracle,
                                      int *valid)
{
    return hodor_fri_verify_proof_combined(ctx, proof, len, HODOR_COMBINER_TRIVIAL, natural_element_index,
                                           expected_value_from_oracle, valid);
}

// COSET2 proofs (one query per round carrying both values of the coset): the same walk as below with the pair
// checked against the round's root by ONE path — see verify_coset2_walk.
static int verify_coset2_walk(const hodor_ctx *ctx, const uint8_t *proof, size_t len, size_t natural_element_index,
                              const hodor_fr *expected_value_from_oracle, int *valid);

extern "C" int hodor_fri_verify_proof_combined(const hodor_ctx *ctx, const uint8_t *proof, size_t len, int combiner,
                                               size_t natural_element_index,
                                               const hodor_fr *expected_value_from_oracle, int *valid)
{
    if (!ctx || !proof || !expected_value_from_oracle || !valid) return HODOR_ERR_INVALID;
    *valid = 0;
    if (combiner == HODOR_COMBINER_COSET2)
        return verify_coset2_walk(ctx, proof, len, natural_element_index, expected_value_from_oracle, valid);
    if (combiner != HODOR_COMBINER_TRIVIAL) return HODOR_ERR_INVALID;
    size_t o = 0;
    bool bad = false;
    auto get64 = [&]() -> uint64_t {
        uint64_t v = 0;
        if (o > len || len - o < 8) { bad = true; return 0; }
        memcpy(&v, proof + o, 8);
        o += 8;
        return v;
    };
    auto take = [&](uint64_t count) -> const uint8_t * {   // count 32-byte entries
        if (bad || count > (len - o) / 32) { bad = true; return nullptr; }
        const uint8_t *r = proof + o;
        o += (size_t)count * 32;
        return r;
    };
    struct Query { uint64_t index; const uint8_t *value; uint64_t path_len; const uint8_t *path; };
    uint64_t nq = get64();
    if (bad || nq > len / 48) return HODOR_ERR_INVALID;
    std::vector<Query> queries((size_t)nq);
    for (auto &q : queries) {
        q.index = get64();
        q.value = take(1);
        q.path_len = get64();
        q.path = take(q.path_len);
        if (bad) return HODOR_ERR_INVALID;
    }
    uint64_t n_roots = get64();
    const uint8_t *roots = take(n_roots);
    uint64_t n_final = get64();
    const uint8_t *final_coeffs = take(n_final);
    uint64_t initial_degree_plus_one = get64();
    (void)get64();   // output_coeffs_at_degree_plus_one: carried by the proof, unused by the verifier
    uint64_t lde_factor = get64();
    if (bad || o != len) return HODOR_ERR_INVALID;

    const HostField &F = ctx->F;
    HFr two_inv, omega, omega_inv;
    if (!F.inverse(F.add(F.one, F.one), &two_inv)) return HODOR_ERR_INVALID;
    uint64_t size;
    uint32_t log_size;
    if (lde_factor && initial_degree_plus_one > ~0ull / lde_factor) return HODOR_ERR_SIZE;
    if (!F.domain(initial_degree_plus_one * lde_factor, &size, &log_size, &omega)) return HODOR_ERR_SIZE;
    HFr x = F.pow(omega, natural_element_index);
    if (!(F.pow(x, size) == F.one) || F.pow(x, size / 2) == F.one) return HODOR_ERR_INVALID;
    if (!F.inverse(omega, &omega_inv)) return HODOR_ERR_INVALID;
    if (queries.size() % 2 != 0) return HODOR_ERR_INVALID;

    auto value_of = [&](const Query &q) { hodor_fr v; memcpy(v.l, q.value, 32); return to_h(&v); };
    bool have_expected = false;
    HFr expected = F.one;
    uint64_t domain_size = size, domain_idx = natural_element_index;
    const HFr oracle_value = to_h(expected_value_from_oracle);
    const size_t rounds = std::min<size_t>((size_t)n_roots, queries.size() / 2);   // zip(roots, chunks_exact)
    for (size_t rnd = 0; rnd < rounds; rnd++) {
        if (domain_size < 2) return HODOR_ERR_INVALID;
        const Query *qs = &queries[2 * rnd];
        const uint8_t *root = roots + 32 * rnd;
        uint64_t pair = (domain_idx + domain_size / 2) % domain_size;
        uint64_t coset[2] = {std::min(domain_idx, pair), std::max(domain_idx, pair)};
        for (int k = 0; k < 2; k++)
            if (qs[k].index != coset[0] && qs[k].index != coset[1]) return HODOR_OK;          // Ok(false)
        if (rnd == 0)
            for (int k = 0; k < 2; k++)
                if (qs[k].index == natural_element_index && !(value_of(qs[k]) == oracle_value)) return HODOR_OK;
        for (int k = 0; k < 2; k++)
            if (qs[k].index != coset[k]) return HODOR_ERR_INVALID;                            // "invalid tree index"
        for (int k = 0; k < 2; k++) {
            int ok = 0;
            hodor_fr leaf;
            memcpy(leaf.l, qs[k].value, 32);
            hodor_iop_verify(ctx, root, &leaf, qs[k].path, (size_t)qs[k].path_len, (size_t)qs[k].index, &ok);
            if (!ok) return HODOR_OK;
        }
        hodor_fr ch;
        if (hodor_iop_challenge(ctx, root, &ch)) return HODOR_ERR_INVALID;
        const HFr challenge = to_h(&ch);
        const HFr f_at_omega = value_of(qs[0]), f_at_minus_omega = value_of(qs[1]);
        if (have_expected) {
            int hits = 0;
            HFr supplied = F.one;
            for (int k = 0; k < 2; k++)
                if (qs[k].index == domain_idx) { hits++; supplied = value_of(qs[k]); }
            if (hits != 1 || !(supplied == expected)) return HODOR_OK;
        }
        HFr divisor = F.pow(omega_inv, coset[0]);
        HFr even = F.add(f_at_omega, f_at_minus_omega);
        HFr odd = F.mul(F.sub(f_at_omega, f_at_minus_omega), divisor);
        expected = F.mul(F.add(F.mul(odd, challenge), even), two_inv);
        have_expected = true;
        uint64_t next = domain_size / 2;                      // index_and_size_for_next_domain
        domain_idx = domain_idx < next ? domain_idx : domain_idx - next;
        domain_size = next;
        omega = F.sqr(omega);
        omega_inv = F.sqr(omega_inv);
    }
    if (!have_expected) return HODOR_ERR_INVALID;             // expect("is some")
    HFr point = F.pow(omega, domain_idx), acc = F.sub(F.one, F.one), power = F.one;
    for (uint64_t i = 0; i < n_final; i++) {
        hodor_fr c;
        memcpy(c.l, final_coeffs + 32 * i, 32);
        acc = F.add(acc, F.mul(power, to_h(&c)));
        power = F.mul(power, point);
    }
    *valid = (acc == expected) ? 1 : 0;
    return HODOR_OK;
}

// hodor_fri_verify_proof with the proof bound to the parameters the CALLER chose FIRST.  The reference's
// verify_proof_queries (src/fri/verifier.rs:131-289) walks zip(roots, queries.chunks_exact(2)) and therefore
// accepts a proof whose tail of rounds has been cut off, or whose final polynomial has any length, and it takes
// lde_factor / initial_degree_plus_one / output_coeffs_at_degree_plus_one from the proof itself — a prover that
// re-encodes lde_factor = 1 (rate 1: every function on the domain is "low degree") passes it.  This variant
// refuses (HODOR_OK with *valid = 0) every buffer whose fields differ from what
// FRIProofPrototype::produce_proof (src/fri/query_producer.rs:10-53) writes for the caller's parameters:
//     lde_factor == expected_lde_factor, output_coeffs_at_degree_plus_one == expected_out_deg,
//     initial_degree_plus_one == expected_domain_size / expected_lde_factor,
//     n_roots == log2(initial_degree_plus_one / out_deg) + 1, n_queries == 2 * n_roots, n_final == out_deg,
//     path_len of round k == log2(expected_domain_size >> k), both queries of a round.
// A well-formed proof is then handed to hodor_fri_verify_proof unchanged.
extern "C" int hodor_fri_verify_proof_strict(const hodor_ctx *ctx, const uint8_t *proof, size_t len,
                                             size_t expected_domain_size, size_t expected_lde_factor,
                                             size_t expected_out_deg, size_t natural_element_index,
                                             const hodor_fr *expected_value_from_oracle, int *valid)
{
    return hodor_fri_verify_proof_strict_combined(ctx, proof, len, HODOR_COMBINER_TRIVIAL, expected_domain_size,
                                                  expected_lde_factor, expected_out_deg, natural_element_index,
                                                  expected_value_from_oracle, valid);
}

static int verify_coset2_walk(const hodor_ctx *ctx, const uint8_t *proof, size_t len, size_t natural_element_index,
                              const hodor_fr *expected_value_from_oracle, int *valid)
{
    size_t o = 0;
    bool bad = false;
    auto get64 = [&]() -> uint64_t {
        uint64_t v = 0;
        if (o > len || len - o < 8) { bad = true; return 0; }
        memcpy(&v, proof + o, 8);
        o += 8;
        return v;
    };
    auto take = [&](uint64_t count) -> const uint8_t * {   // count 32-byte entries
        if (bad || count > (len - o) / 32) { bad = true; return nullptr; }
        const uint8_t *r = proof + o;
        o += (size_t)count * 32;
        return r;
    };
    struct Query { uint64_t index; const uint8_t *values; uint64_t path_len; const uint8_t *path; };
    uint64_t nq = get64();
    if (bad || nq > len / 80) return HODOR_ERR_INVALID;
    std::vector<Query> queries((size_t)nq);
    for (auto &q : queries) {
        q.index = get64();
        q.values = take(2);
        q.path_len = get64();
        q.path = take(q.path_len);
        if (bad) return HODOR_ERR_INVALID;
    }
    uint64_t n_roots = get64();
    const uint8_t *roots = take(n_roots);
    uint64_t n_final = get64();
    const uint8_t *final_coeffs = take(n_final);
    uint64_t initial_degree_plus_one = get64();
    (void)get64();
    uint64_t lde_factor = get64();
    if (bad || o != len) return HODOR_ERR_INVALID;

    const HostField &F = ctx->F;
    HFr two_inv, omega, omega_inv;
    if (!F.inverse(F.add(F.one, F.one), &two_inv)) return HODOR_ERR_INVALID;
    uint64_t size;
    uint32_t log_size;
    if (lde_factor && initial_degree_plus_one > ~0ull / lde_factor) return HODOR_ERR_SIZE;
    if (!F.domain(initial_degree_plus_one * lde_factor, &size, &log_size, &omega)) return HODOR_ERR_SIZE;
    HFr x = F.pow(omega, natural_element_index);
    if (!(F.pow(x, size) == F.one) || F.pow(x, size / 2) == F.one) return HODOR_ERR_INVALID;
    if (!F.inverse(omega, &omega_inv)) return HODOR_ERR_INVALID;

    auto value_of = [&](const uint8_t *b) { hodor_fr v; memcpy(v.l, b, 32); return to_h(&v); };
    bool have_expected = false;
    HFr expected = F.one;
    uint64_t domain_size = size, domain_idx = natural_element_index;
    const HFr oracle_value = to_h(expected_value_from_oracle);
    const size_t rounds = std::min<size_t>((size_t)n_roots, queries.size());   // zip(roots, queries)
    for (size_t rnd = 0; rnd < rounds; rnd++) {
        if (domain_size < 4) return HODOR_ERR_INVALID;        // no combined tree over fewer than two leaves
        const Query &q = queries[rnd];
        const uint8_t *root = roots + 32 * rnd;
        uint64_t pair = (domain_idx + domain_size / 2) % domain_size;
        uint64_t coset[2] = {std::min(domain_idx, pair), std::max(domain_idx, pair)};
        if (q.index != coset[0] && q.index != coset[1]) return HODOR_OK;                      // Ok(false)
        if (q.index != coset[0]) return HODOR_ERR_INVALID;                                    // "invalid tree index"
        const HFr f_at_omega = value_of(q.values), f_at_minus_omega = value_of(q.values + 32);
        const HFr &supplied = domain_idx == coset[0] ? f_at_omega : f_at_minus_omega;
        if (rnd == 0 && !(supplied == oracle_value)) return HODOR_OK;
        int ok = 0;
        hodor_fr pair_values[2];
        memcpy(pair_values, q.values, 64);
        if (hodor_iop_verify_combined(ctx, root, pair_values, q.path, (size_t)q.path_len, (size_t)coset[0],
                                      (size_t)domain_size, HODOR_COMBINER_COSET2, &ok))
            return HODOR_ERR_INVALID;
        if (!ok) return HODOR_OK;
        hodor_fr ch;
        if (hodor_iop_challenge(ctx, root, &ch)) return HODOR_ERR_INVALID;
        if (have_expected && !(supplied == expected)) return HODOR_OK;
        HFr divisor = F.pow(omega_inv, coset[0]);
        HFr even = F.add(f_at_omega, f_at_minus_omega);
        HFr odd = F.mul(F.sub(f_at_omega, f_at_minus_omega), divisor);
        expected = F.mul(F.add(F.mul(odd, to_h(&ch)), even), two_inv);
        have_expected = true;
        uint64_t next = domain_size / 2;
        domain_idx = domain_idx < next ? domain_idx : domain_idx - next;
        domain_size = next;
        omega = F.sqr(omega);
        omega_inv = F.sqr(omega_inv);
    }
    if (!have_expected) return HODOR_ERR_INVALID;
    HFr point = F.pow(omega, domain_idx), acc = F.sub(F.one, F.one), power = F.one;
    for (uint64_t i = 0; i < n_final; i++) {
        hodor_fr c;
        memcpy(c.l, final_coeffs + 32 * i, 32);
        acc = F.add(acc, F.mul(power, to_h(&c)));
        power = F.mul(power, point);
    }
    *valid = (acc == expected) ? 1 : 0;
    return HODOR_OK;
}

extern "C" int hodor_fri_verify_proof_strict_combined(const hodor_ctx *ctx, const uint8_t *proof, size_t len,
                                                      int combiner, size_t expected_domain_size,
                                                      size_t expected_lde_factor, size_t expected_out_deg,
                                                      size_t natural_element_index,
                                                      const hodor_fr *expected_value_from_oracle, int *valid)
{
    if (!ctx || !proof || !expected_value_from_oracle || !valid) return HODOR_ERR_INVALID;
    *valid = 0;
    if (combiner != HODOR_COMBINER_TRIVIAL && combiner != HODOR_COMBINER_COSET2) return HODOR_ERR_INVALID;
    const bool comb = combiner == HODOR_COMBINER_COSET2;
    if (!is_pow2(expected_domain_size) || natural_element_index >= expected_domain_size) return HODOR_ERR_SIZE;
    if (!is_pow2(expected_lde_factor) || !is_pow2(expected_out_deg) || expected_lde_factor > expected_domain_size ||
        expected_out_deg > expected_domain_size / expected_lde_factor)
        return HODOR_ERR_SIZE;
    size_t o = 0;
    auto get64 = [&](uint64_t *v) -> bool {
        if (o > len || len - o < 8) return false;
        memcpy(v, proof + o, 8);
        o += 8;
        return true;
    };
    auto skip = [&](uint64_t count) -> bool {   // count 32-byte entries
        if (count > (len - o) / 32) return false;
        o += (size_t)count * 32;
        return true;
    };
    uint64_t nq = 0;
    if (!get64(&nq) || nq > len / 48) return HODOR_ERR_INVALID;
    std::vector<uint64_t> path_lens((size_t)nq);
    for (auto &pl : path_lens) {
        uint64_t index;
        if (!get64(&index) || !skip(comb ? 2 : 1) || !get64(&pl) || !skip(pl)) return HODOR_ERR_INVALID;
    }
    uint64_t n_roots = 0, n_final = 0, deg = 0, out_deg = 0, factor = 0;
    if (!get64(&n_roots) || !skip(n_roots) || !get64(&n_final) || !skip(n_final) || !get64(&deg) ||
        !get64(&out_deg) || !get64(&factor) || o != len)
        return HODOR_ERR_INVALID;
    // the low-degree claim is the CALLER's: rate and degree bound as the verifier chose them, not as the proof says
    if (factor != expected_lde_factor || out_deg != expected_out_deg ||
        deg != expected_domain_size / expected_lde_factor)
        return HODOR_OK;
    const uint64_t rounds = log2u((size_t)deg) - log2u((size_t)out_deg) + 1;
    const uint64_t per_round = comb ? 1 : 2;
    if (n_roots != rounds || nq != per_round * rounds || n_final != out_deg) return HODOR_OK;
    if (comb && (expected_domain_size >> (rounds - 1)) < 4) return HODOR_OK;
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t want = log2u(expected_domain_size >> r) - (comb ? 1 : 0);
        for (uint64_t k = 0; k < per_round; k++)
            if (path_lens[per_round * r + k] != want) return HODOR_OK;
    }
    return hodor_fri_verify_proof_combined(ctx, proof, len, combiner, natural_element_index,
                                           expected_value_from_oracle, valid);
}

// NaiveFriIop::verify_prototype (src/fri/verifier.rs:10-129): the same folding walk against the
// prover's own (device-resident) vectors instead of Merkle queries — two elements per round are
// fetched from the device.  `lde_values_dev` is the codeword the prototype was committed from.
extern "C" int hodor_fri_verify_prototype(hodor_fri_proto *p, const hodor_fr *lde_values_dev,
                                          size_t natural_element_index, int *valid)
{
    if (!p || !lde_values_dev || !valid) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    *valid = 0;
    const HostField &F = ctx->F;
    HFr two_inv, omega, omega_inv;
    if (!F.inverse(F.add(F.one, F.one), &two_inv)) return HODOR_ERR_INVALID;
    uint64_t size;
    uint32_t log_size;
    if (!F.domain((uint64_t)p->initial_degree_plus_one * p->lde_factor, &size, &log_size, &omega))
        return HODOR_ERR_SIZE;
    HFr x = F.pow(omega, natural_element_index);
    if (!(F.pow(x, size) == F.one) || F.pow(x, size / 2) == F.one) {
        set_err(ctx, "initial challenge value is not in the LDE domain");
        return HODOR_ERR_INVALID;
    }
    if (!F.inverse(omega, &omega_inv)) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    bool have_expected = false;
    HFr expected = F.one;
    uint64_t domain_size = size, domain_idx = natural_element_index;
    auto fetch = [&](const hodor_fr *base, uint64_t i, HFr *out) -> bool {
        hodor_fr v;
        HostXfer xfer(ctx, nullptr);     // a blocking copy of one element, through the pinned buffer like every small one
        if (xfer.d2h(&v, base + i, 32) != hipSuccess || xfer.finish() != hipSuccess) return false;
        *out = to_h(&v);
        return true;
    };
    // zip(leaf_values ++ intermediate_values, challenges): num_steps rounds
    for (size_t rnd = 0; rnd < p->num_steps; rnd++) {
        const hodor_fr *values = rnd == 0 ? lde_values_dev : (const hodor_fr *)p->inter_values[rnd - 1];
        uint64_t pair = (domain_idx + domain_size / 2) % domain_size;
        uint64_t coset[2] = {std::min(domain_idx, pair), std::max(domain_idx, pair)};
        HFr f_at_omega, f_at_minus_omega;
        if (!fetch(values, coset[0], &f_at_omega) || !fetch(values, coset[1], &f_at_minus_omega)) {
            set_err(ctx, "verify_prototype: device read failed");
            return HODOR_ERR_DEVICE;
        }
        if (have_expected) {
            const HFr &supplied = domain_idx == coset[0] ? f_at_omega : f_at_minus_omega;
            if (!(supplied == expected)) return HODOR_OK;     // Ok(false)
        }
        hodor_fr ch = p->challenges[rnd];
        HFr divisor = F.pow(omega_inv, coset[0]);
        HFr even = F.add(f_at_omega, f_at_minus_omega);
        HFr odd = F.mul(F.sub(f_at_omega, f_at_minus_omega), divisor);
        expected = F.mul(F.add(F.mul(odd, to_h(&ch)), even), two_inv);
        have_expected = true;
        uint64_t next = domain_size / 2;
        domain_idx = domain_idx < next ? domain_idx : domain_idx - next;
        domain_size = next;
        omega = F.sqr(omega);
        omega_inv = F.sqr(omega_inv);
    }
    if (!have_expected) return HODOR_ERR_INVALID;
    HFr point = F.pow(omega, domain_idx), acc = F.sub(F.one, F.one), power = F.one;
    for (size_t i = 0; i < p->final_coeffs.size(); i++) {
        hodor_fr c = p->final_coeffs[i];
        acc = F.add(acc, F.mul(power, to_h(&c)));
        power = F.mul(power, point);
    }
    *valid = (acc == expected) ? 1 : 0;
    return HODOR_OK;
}

extern "C" size_t hodor_fri_num_steps(const hodor_fri_proto *p) { return p ? p->num_steps : 0; }
extern "C" int hodor_fri_combiner(const hodor_fri_proto *p) { return p ? p->combiner : -1; }

extern "C" int hodor_fri_roots(const hodor_fri_proto *p, uint8_t *roots)
{
    if (!p || !roots) return HODOR_ERR_INVALID;
    memcpy(roots, p->roots.data(), p->roots.size());
    return HODOR_OK;
}
extern "C" int hodor_fri_final_root(const hodor_fri_proto *p, uint8_t root[32])
{
    if (!p || !root) return HODOR_ERR_INVALID;
    memcpy(root, p->final_root, 32);
    return HODOR_OK;
}
extern "C" int hodor_fri_challenges(const hodor_fri_proto *p, hodor_fr *c)
{
    if (!p || !c) return HODOR_ERR_INVALID;
    memcpy(c, p->challenges.data(), 32 * p->num_steps);
    return HODOR_OK;
}
extern "C" int hodor_fri_final_coefficients(const hodor_fri_proto *p, hodor_fr *c)
{
    if (!p || !c) return HODOR_ERR_INVALID;
    memcpy(c, p->final_coeffs.data(), 32 * p->out_deg);
    return HODOR_OK;
}
extern "C" int hodor_fri_intermediate_values(hodor_fri_proto *p, size_t step, hodor_fr *values)
{
    if (!p || !values || step >= p->num_steps) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    return hodor_buf_download(ctx, values, p->inter_values[step], p->inter_sizes[step] * 32);
}
extern "C" int hodor_fri_tree_nodes(hodor_fri_proto *p, int step, uint8_t *nodes)
{
    if (!p || !nodes || step < -1 || step >= (int)p->num_steps) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    const size_t shift = p->combiner == HODOR_COMBINER_COSET2 ? 1 : 0;   // a COSET2 tree has half as many entries
    if (step < 0) return hodor_buf_download(ctx, nodes, p->l0_nodes, (p->n >> shift) * 32);
    return hodor_buf_download(ctx, nodes, p->inter_nodes[step], (p->inter_sizes[step] >> shift) * 32);
}

extern "C" size_t hodor_fri_serialize(const hodor_fri_proto *p, uint8_t *buf, size_t cap)
{
    if (!p) return 0;
    size_t need = 8 + 32 * (p->num_steps + 1) + 32 * p->num_steps + 32 + 8 + 32 * p->out_deg;
    if (!buf || cap < need) return need;
    size_t o = 0;
    uint64_t ns = p->num_steps, nf = p->out_deg;
    memcpy(buf + o, &ns, 8); o += 8;                                   // little-endian host
    memcpy(buf + o, p->roots.data(), p->roots.size()); o += p->roots.size();
    memcpy(buf + o, p->challenges.data(), 32 * p->num_steps); o += 32 * p->num_steps;
    memcpy(buf + o, p->final_root, 32); o += 32;
    memcpy(buf + o, &nf, 8); o += 8;
    memcpy(buf + o, p->final_coeffs.data(), 32 * p->out_deg); o += 32 * p->out_deg;
    return o;
}
