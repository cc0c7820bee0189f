// abi_host.hip — entry points that never touch the device: field helpers, single BLAKE2s hashes, path
// and root helpers of the IOP, the transcript.  They work on a context created with device = -1.
#include "ctx.hpp"

// ------------------------------------------------------------------------------------------------
// host scalar helpers
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_fr_mul(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out)
{
    if (!ctx || !a || !b || !out) return HODOR_ERR_INVALID;
    from_h(ctx->F.mul(to_h(a), to_h(b)), out);
    return HODOR_OK;
}
extern "C" int hodor_fr_add(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out)
{
    if (!ctx || !a || !b || !out) return HODOR_ERR_INVALID;
    from_h(ctx->F.add(to_h(a), to_h(b)), out);
    return HODOR_OK;
}
extern "C" int hodor_fr_sub(const hodor_ctx *ctx, const hodor_fr *a, const hodor_fr *b, hodor_fr *out)
{
    if (!ctx || !a || !b || !out) return HODOR_ERR_INVALID;
    from_h(ctx->F.sub(to_h(a), to_h(b)), out);
    return HODOR_OK;
}
extern "C" int hodor_fr_pow(const hodor_ctx *ctx, const hodor_fr *a, uint64_t e, hodor_fr *out)
{
    if (!ctx || !a || !out) return HODOR_ERR_INVALID;
    from_h(ctx->F.pow(to_h(a), e), out);
    return HODOR_OK;
}
extern "C" int hodor_fr_inverse(const hodor_ctx *ctx, const hodor_fr *a, hodor_fr *out)
{
    if (!ctx || !a || !out) return HODOR_ERR_INVALID;
    HFr r;
    if (!ctx->F.inverse(to_h(a), &r)) return HODOR_ERR_INVALID;
    from_h(r, out);
    return HODOR_OK;
}
extern "C" int hodor_fr_from_repr(const hodor_ctx *ctx, const uint64_t c[4], hodor_fr *out)
{
    if (!ctx || !c || !out) return HODOR_ERR_INVALID;
    HFr r;
    if (!ctx->F.from_repr(c, &r)) return HODOR_ERR_INVALID;
    from_h(r, out);
    return HODOR_OK;
}
extern "C" int hodor_fr_into_repr(const hodor_ctx *ctx, const hodor_fr *a, uint64_t c[4])
{
    if (!ctx || !a || !c) return HODOR_ERR_INVALID;
    ctx->F.into_repr(to_h(a), c);
    return HODOR_OK;
}

extern "C" int hodor_domain_new_for_size(const hodor_ctx *ctx, uint64_t size, uint64_t *out_size,
                                         uint32_t *out_log_n, hodor_fr *out_generator)
{
    if (!ctx || !out_size || !out_log_n || !out_generator) return HODOR_ERR_INVALID;
    HFr g;
    if (!ctx->F.domain(size, out_size, out_log_n, &g)) return HODOR_ERR_SIZE;
    from_h(g, out_generator);
    return HODOR_OK;
}

extern "C" int hodor_iop_challenge(const hodor_ctx *ctx, const uint8_t root[32], hodor_fr *out)
{
    if (!ctx || !root || !out) return HODOR_ERR_INVALID;
    uint64_t repr[4];
    for (int i = 0; i < 4; i++) {   // read_be
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w = (w << 8) | root[8 * i + b];
        repr[3 - i] = w;
    }
    uint32_t shave = 256 - ctx->F.capacity;
    repr[3] &= 0xffffffffffffffffull >> (shave % 64);
    HFr r;
    if (!ctx->F.from_repr(repr, &r)) return HODOR_ERR_INVALID;   // "in a field" expect
    from_h(r, out);
    return HODOR_OK;
}

static void host_hash_leaf(const hodor_ctx *ctx, const hodor_fr *leaf, uint8_t out[32])
{
    HostBlake2s::finish(ctx->mid.h, (const uint8_t *)leaf->l, 32, out);   // LE limbs == memory image
}
static void host_hash_node(const hodor_ctx *ctx, const uint8_t *l, const uint8_t *r, uint8_t out[32])
{
    uint8_t buf[64];
    memcpy(buf, l, 32);
    memcpy(buf + 32, r, 32);
    HostBlake2s::finish(ctx->mid.h, buf, 64, out);
}

// IopTreeHasher::{hash_leaf, hash_node} (src/iop/blake2s_trivial_iop.rs:81-104) for single digests on
// the host: the top log2(P) levels of a tree whose subtrees live on P GPUs, path checks, ...
extern "C" int hodor_hash_leaf(const hodor_ctx *ctx, const hodor_fr *leaf, uint8_t out[32])
{
    if (!ctx || !leaf || !out) return HODOR_ERR_INVALID;
    host_hash_leaf(ctx, leaf, out);
    return HODOR_OK;
}
extern "C" int hodor_hash_node(const hodor_ctx *ctx, const uint8_t left[32], const uint8_t right[32], uint8_t out[32])
{
    if (!ctx || !left || !right || !out) return HODOR_ERR_INVALID;
    host_hash_node(ctx, left, right, out);
    return HODOR_OK;
}

extern "C" int hodor_iop_path(const hodor_ctx *ctx, const uint8_t *nodes, const hodor_fr *leafs, size_t n,
                              size_t tree_index, uint8_t *path, size_t *path_len)
{
    if (!ctx || !nodes || !leafs || !path || !path_len) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2 || tree_index >= n) return HODOR_ERR_SIZE;
    size_t cnt = 0;
    host_hash_leaf(ctx, &leafs[tree_index ^ 1], path);
    cnt++;
    size_t idx = tree_index >> 1;
    for (size_t w = n / 2; w >= 2; w /= 2) {
        memcpy(path + 32 * cnt, nodes + 32 * (w + (idx ^ 1)), 32);
        cnt++;
        idx >>= 1;
    }
    *path_len = cnt;
    return HODOR_OK;
}

extern "C" int hodor_iop_verify(const hodor_ctx *ctx, const uint8_t root[32], const hodor_fr *leaf,
                                const uint8_t *path, size_t path_len, size_t tree_index, int *ok)
{
    if (!ctx || !root || !leaf || (!path && path_len) || !ok) return HODOR_ERR_INVALID;
    uint8_t h[32], t[32];
    host_hash_leaf(ctx, leaf, h);
    size_t idx = tree_index;
    for (size_t i = 0; i < path_len; i++) {
        if ((idx & 1) == 0) host_hash_node(ctx, h, path + 32 * i, t);
        else host_hash_node(ctx, path + 32 * i, h, t);
        memcpy(h, t, 32);
        idx >>= 1;
    }
    *ok = memcmp(h, root, 32) == 0;
    return HODOR_OK;
}

// ---- the same three for a tree built by `combiner` (COSET2: leaf k = value[k] || value[k + n/2], see the header) ----
static void host_hash_pair(const hodor_ctx *ctx, const hodor_fr *lo, const hodor_fr *hi, uint8_t out[32])
{
    uint8_t buf[64];
    memcpy(buf, lo->l, 32);          // encode_leaf: the raw Montgomery limbs, little-endian (host is LE)
    memcpy(buf + 32, hi->l, 32);
    HostBlake2s::finish(ctx->mid.hp, buf, 64, out);   // the COSET2 leaf midstate (personal "Shaftoe2"): not a node hash
}

extern "C" int hodor_hash_leaf_combined(const hodor_ctx *ctx, const hodor_fr *values, int combiner, uint8_t out[32])
{
    if (!ctx || !values || !out) return HODOR_ERR_INVALID;
    if (combiner == HODOR_COMBINER_TRIVIAL) host_hash_leaf(ctx, values, out);
    else if (combiner == HODOR_COMBINER_COSET2) host_hash_pair(ctx, values, values + 1, out);
    else return HODOR_ERR_INVALID;
    return HODOR_OK;
}

extern "C" int hodor_iop_path_combined(const hodor_ctx *ctx, const uint8_t *nodes, const hodor_fr *leafs, size_t n,
                                       int combiner, size_t natural_index, uint8_t *path, size_t *path_len)
{
    if (combiner == HODOR_COMBINER_TRIVIAL) return hodor_iop_path(ctx, nodes, leafs, n, natural_index, path, path_len);
    if (combiner != HODOR_COMBINER_COSET2 || !ctx || !nodes || !leafs || !path || !path_len) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4 || natural_index >= n) return HODOR_ERR_SIZE;
    const size_t leaves = n / 2, k = natural_index % leaves;
    size_t cnt = 0;
    host_hash_pair(ctx, &leafs[k ^ 1], &leafs[(k ^ 1) + leaves], path);
    cnt++;
    size_t idx = k >> 1;
    for (size_t w = leaves / 2; w >= 2; w /= 2) {
        memcpy(path + 32 * cnt, nodes + 32 * (w + (idx ^ 1)), 32);
        cnt++;
        idx >>= 1;
    }
    *path_len = cnt;
    return HODOR_OK;
}

extern "C" int hodor_iop_verify_combined(const hodor_ctx *ctx, const uint8_t root[32], const hodor_fr *values,
                                         const uint8_t *path, size_t path_len, size_t natural_index, size_t n,
                                         int combiner, int *ok)
{
    if (combiner == HODOR_COMBINER_TRIVIAL) return hodor_iop_verify(ctx, root, values, path, path_len, natural_index, ok);
    if (combiner != HODOR_COMBINER_COSET2 || !ctx || !root || !values || (!path && path_len) || !ok)
        return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4 || natural_index >= n) return HODOR_ERR_SIZE;
    *ok = 0;
    // COSET2 is this build's own format and its leaf compresses like a node (64 bytes, t = 128): what tells a leaf
    // from an interior node is its DEPTH.  A path of any other length than log2(n) - 1 would let a prover open the two
    // child digests of an interior node as a "coset value pair" — refused here, for every caller (the FRI walks
    // included).  The same for values that are not canonical residues: two byte strings must not open as one element.
    if (path_len != (size_t)log2u(n) - 1) return HODOR_OK;
    for (int k = 0; k < 2; k++)
        if (HostField::geq(values[k].l, ctx->F.p)) return HODOR_OK;
    uint8_t h[32], t[32];
    host_hash_pair(ctx, values, values + 1, h);
    size_t idx = natural_index % (n / 2);
    for (size_t i = 0; i < path_len; i++) {
        if ((idx & 1) == 0) host_hash_node(ctx, h, path + 32 * i, t);
        else host_hash_node(ctx, path + 32 * i, h, t);
        memcpy(h, t, 32);
        idx >>= 1;
    }
    *ok = memcmp(h, root, 32) == 0;
    return HODOR_OK;
}

// ---- Blake2sTranscript (src/transcript/mod.rs:10-80): host-side, sequential, O(#roots) ----
struct hodor_transcript {
    const hodor_ctx *ctx;
    HostBlake2sStream state;
    hodor_transcript(const hodor_ctx *c)
        : ctx(c), state((const uint8_t *)"Squeamish Ossifrage", 19, (const uint8_t *)"Shaftoe", 7) {}
};

extern "C" int hodor_transcript_new(const hodor_ctx *ctx, hodor_transcript **out)
{
    if (!ctx || !out) return HODOR_ERR_INVALID;
    if (ctx->F.num_bits >= 256) return HODOR_ERR_INVALID;   // assert!(F::NUM_BITS < 256), :41
    *out = new (std::nothrow) hodor_transcript(ctx);
    return *out ? HODOR_OK : HODOR_ERR_INVALID;
}
extern "C" void hodor_transcript_free(hodor_transcript *t) { delete t; }
extern "C" int hodor_transcript_commit_bytes(hodor_transcript *t, const uint8_t *bytes, size_t len)
{
    if (!t || (!bytes && len)) return HODOR_ERR_INVALID;
    t->state.update(bytes, len);
    return HODOR_OK;
}
extern "C" int hodor_transcript_commit_field_element(hodor_transcript *t, const hodor_fr *e)
{
    if (!t || !e) return HODOR_ERR_INVALID;
    uint64_t repr[4];
    t->ctx->F.into_repr(to_h(e), repr);                    // into_repr(): canonical, then write_be (:52-57)
    uint8_t be[32];
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) be[8 * i + b] = (uint8_t)(repr[3 - i] >> (56 - 8 * b));
    t->state.update(be, 32);
    return HODOR_OK;
}
extern "C" int hodor_transcript_get_challenge_bytes(hodor_transcript *t, uint8_t out[32])
{
    if (!t || !out) return HODOR_ERR_INVALID;
    t->state.finalize(out);
    t->state.update(out, 32);                               // the digest is re-absorbed (:61-62)
    return HODOR_OK;
}
extern "C" int hodor_transcript_get_challenge(hodor_transcript *t, hodor_fr *out)
{
    if (!t || !out) return HODOR_ERR_INVALID;
    uint8_t v[32];
    t->state.finalize(v);
    t->state.update(v, 32);
    return hodor_iop_challenge(t->ctx, v, out);             // same read_be + shave + from_repr as interpret_hash
}
// Verifier::bytes_to_challenge_index (src/verifier/mod.rs:246-263)
extern "C" size_t hodor_bytes_to_challenge_index(const uint8_t *bytes, size_t len, size_t lde_size, size_t lde_factor)
{
    if (!bytes || len < 8 || !lde_size || !lde_factor) return 0;
    uint64_t x = 0;
    for (size_t i = len - 8; i < len; i++) x = (x << 8) | bytes[i];   // BigEndian::read_u64 of the last 8 bytes
    size_t idx = (size_t)x % lde_size;
    if (idx % lde_factor == 0) idx = (idx + 1) % lde_size;
    if (idx % 2 == 0) idx = (idx + 1) % lde_size;
    return idx;
}

// ------------------------------------------------------------------------------------------------
// tuning knobs: one thread-safe read per process (C++11 static initialisation)
// ------------------------------------------------------------------------------------------------
namespace hodor {
static Knobs read_knobs()
{
    Knobs k;
    k.set[0] = 0;
    size_t used = 0;
    auto get = [&](const char *name, int dflt, int lo, int hi) {
        const char *e = getenv(name);
        if (!e) return dflt;
        int n = snprintf(k.set + used, sizeof(k.set) - used, "%s%s=%s", used ? " " : "", name, e);
        if (n > 0 && used + (size_t)n < sizeof(k.set)) used += (size_t)n;
        int v = atoi(e);
        return (v < lo || v > hi) ? dflt : v;
    };
#ifdef HODOR_TWOPASS
    k.max_log_r = get("HODOR_MAX_LOG_R", 9, 2, 12);
#else
    k.max_log_r = get("HODOR_MAX_LOG_R", 9, 2, 11);
#endif
#ifdef HODOR_TWOPASS
    k.tile_log = get("HODOR_TILE_LOG", 10, 6, 12);
#else
    k.tile_log = get("HODOR_TILE_LOG", 10, 6, 11);   // 2048 elements x 36 B + a quarter twiddle table is the largest tile that fits 160 KB
#endif
    k.min_log_c = get("HODOR_MIN_LOG_C", 2, 0, 4);
    k.tw_hi_max_log = get("HODOR_TW_HI_MAX_LOG", 17, 0, 20);
    k.ntt_threads = get("HODOR_NTT_THREADS", 0, 0, 1024);
    k.ntt_tw_sub = get("HODOR_NTT_TW_SUB", 1, 0, 1);
    k.ntt_w9 = get("HODOR_NTT_W9", 2, 0, 2);
    k.ntt_p1 = get("HODOR_NTT_P1", 1, 0, 1);
    k.merkle_tail_log = get("HODOR_MERKLE_TAIL_LOG", 6, 0, 30);
    k.merkle_lat_log = get("HODOR_MERKLE_LAT_LOG", 19, 0, 40);
    k.fri_tail = get("HODOR_FRI_TAIL", 1, 0, 1);
    k.fri_fuse_fold = get("HODOR_FRI_FUSE_FOLD", 1, 0, 2);
    k.batchinv_seq = get("HODOR_BATCHINV_SEQ", 8, 2, 64);
    k.table_cache = get("HODOR_TABLE_CACHE", 40, 1, 1000);
    k.slice_serial = get("HODOR_SLICE_SERIAL", 1, 0, 1);
    k.pool_cache_gib = get("HODOR_POOL_CACHE_GIB", 64, 0, 4096);
    return k;
}
const Knobs &knobs()
{
    static const Knobs k = read_knobs();
    return k;
}
}  // namespace hodor

extern "C" const char *hodor_knobs_set(void) { return hodor::knobs().set; }

// ---- allocation wrappers + fault injection (ctx.hpp)
namespace hodor {
namespace {
struct FailAlloc {
    long long k = 0;        // 0: off
    bool from_on = false;   // "<k>+"
    FailAlloc()
    {
        const char *e = getenv("HODOR_DEBUG_FAIL_ALLOC");
        if (!e || !*e) return;
        char *end = nullptr;
        k = strtoll(e, &end, 10);
        if (k < 0) k = 0;
        from_on = end && *end == '+';
    }
};
std::atomic<long long> g_alloc_calls{0};
std::atomic<long long> g_fail_at{-1};     // -1: not armed at run time (the environment decides)
std::atomic<int> g_fail_from_on{0};
bool alloc_fails()
{
    static const FailAlloc f;
    const long long c = g_alloc_calls.fetch_add(1) + 1;
    const long long at = g_fail_at.load();
    if (at >= 0) return at && (g_fail_from_on.load() ? c >= at : c == at);
    return f.k && (f.from_on ? c >= f.k : c == f.k);
}
}  // namespace
hipError_t dev_malloc(void **p, size_t bytes)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
hipError_t pinned_malloc(void **p, size_t bytes, unsigned flags)
{
    if (alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, flags);
}
}  // namespace hodor

// allocations the library has asked for so far in this process (what HODOR_DEBUG_FAIL_ALLOC counts)
extern "C" long long hodor_debug_alloc_calls(void) { return hodor::g_alloc_calls.load(); }
// arm at run time: the k-th allocation FROM NOW fails (k = 0: none does); from_on != 0: so does every later one
extern "C" void hodor_debug_fail_alloc(long long k, int from_on)
{
    hodor::g_fail_from_on.store(from_on ? 1 : 0);
    hodor::g_fail_at.store(k > 0 ? hodor::g_alloc_calls.load() + k : 0);
}

