// abi_poly.hip — the handle API of include/hodor_gpu.h: device-resident Polynomial<F, P> and IOP objects behind the
// reference's method surface (src/polynomials/mod.rs:26-955, src/iop/blake2s_trivial_iop.rs:106-339), so that the
// layers above (src/arp, src/ali, src/prover) keep calling the same methods on the same types while the vectors stay
// in HBM.  Every method is the `_dev` entry point it names, enqueued on the context's own stream, over buffers from
// the context's pool; nothing here computes on the host.
#include "ctx.hpp"

// ------------------------------------------------------------------------------------------------
// the pool
// ------------------------------------------------------------------------------------------------
// Blocks are recycled by size.  Polynomial sizes are powers of two and a FRI slab's size is a function of its shape, so
// from 1 MiB up a request is served by a cached block of EXACTLY its (MiB-rounded) size or by hipMalloc — round 5's
// "smallest block that fits and is not more than twice as large" handed 2^k-element requests 2^(k+1)-element blocks and
// cost a third of the pool; small blocks (flags, staging, 2-element polynomials) still take anything up to twice their
// size.  Ordering (round 6): a block goes back with an event recorded on the stream of its LAST user (ctx->stream for
// the handles; a `_dev` caller's own stream for the prototypes and staging blocks it made), and a new user on any other
// stream first waits for that event — the same stream needs nothing, stream order does it.
static size_t pool_round(size_t bytes)
{
    if (bytes < 256) return 256;
    if (bytes < ((size_t)1 << 20)) return (bytes + 255) & ~(size_t)255;
    return (bytes + (((size_t)1 << 20) - 1)) & ~(((size_t)1 << 20) - 1);     // MiB granules: fewer distinct sizes
}

static void pool_event_put(hodor_ctx *ctx, hipEvent_t ev)   // caller holds pool_mu
{
    if (!ev) return;
    if (ctx->pool_events.size() < 256) ctx->pool_events.push_back(ev);
    else (void)hipEventDestroy(ev);
}

// blocks evicted from the cache wait here until the library is about to enter the allocator anyway (hipFree waits for
// the whole device: a handle `free` in the middle of a proof must never do that)
static void pool_reap(hodor_ctx *ctx)
{
    std::vector<hodor_ctx::PoolBlock> dead;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        dead.swap(ctx->pool_zombies);
        ctx->pool_zombie_bytes = 0;
    }
    if (dead.empty()) return;
    for (auto &b : dead) {
        if (b.ev) (void)hipEventSynchronize(b.ev);
        (void)hipFree(b.p);
    }
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    for (auto &b : dead) pool_event_put(ctx, b.ev);
}

void pool_drain(hodor_ctx *ctx)
{
    std::multimap<size_t, hodor_ctx::PoolBlock> blocks;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        blocks.swap(ctx->pool_free);
        ctx->pool_cached = 0;
    }
    pool_reap(ctx);
    if (blocks.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto &b : blocks) (void)hipFree(b.second.p);
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    for (auto &b : blocks) pool_event_put(ctx, b.second.ev);
}

void pool_destroy_events(hodor_ctx *ctx)   // hodor_ctx_destroy, after pool_drain
{
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    for (hipEvent_t ev : ctx->pool_events) (void)hipEventDestroy(ev);
    ctx->pool_events.clear();
}

int pool_alloc(hodor_ctx *ctx, size_t bytes, void **out, size_t *got, hipStream_t consumer)
{
    const size_t want = pool_round(bytes);
    bool reap = false;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        auto it = ctx->pool_free.lower_bound(want);
        if (it != ctx->pool_free.end() && (it->first == want || (want < ((size_t)1 << 20) && it->first <= 2 * want))) {
            hodor_ctx::PoolBlock b = it->second;
            *out = b.p;
            *got = it->first;
            ctx->pool_cached -= it->first;
            ctx->pool_live += it->first;
            if (ctx->pool_live > ctx->pool_peak_live) ctx->pool_peak_live = ctx->pool_live;
            ctx->pool_free.erase(it);
            hipError_t e = hipSuccess;
            if (b.ev && b.last != consumer) e = hipStreamWaitEvent(consumer, b.ev, 0);
            pool_event_put(ctx, b.ev);
            if (e != hipSuccess) {   // cannot order the new user behind the old one: wait for the old one here
                (void)hipGetLastError();
                (void)hipStreamSynchronize(b.last);
            }
            BOUNDS_NOTE(*out, bytes);   // the bounds build knows a block by what was ASKED for, not by its size class
            return HODOR_OK;
        }
        reap = !ctx->pool_zombies.empty();
    }
    if (reap) pool_reap(ctx);   // entering the allocator anyway: the evicted blocks go back to HIP first
    hipError_t e = dev_malloc(out, want);
    if (e != hipSuccess) {   // give the cache back and try once more
        (void)hipGetLastError();
        pool_drain(ctx);
        e = dev_malloc(out, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err(ctx, std::string("device pool: hipMalloc: ") + hipGetErrorString(e));
        return HODOR_ERR_DEVICE;
    }
    *got = want;
    BOUNDS_NOTE(*out, bytes);
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    ctx->pool_live += want;
    if (ctx->pool_live > ctx->pool_peak_live) ctx->pool_peak_live = ctx->pool_live;
    return HODOR_OK;
}
int pool_alloc(hodor_ctx *ctx, size_t bytes, void **out, size_t *got) { return pool_alloc(ctx, bytes, out, got, ctx->stream); }

void pool_release(hodor_ctx *ctx, void *p, size_t bytes, hipStream_t last_user)
{
    if (!p) return;
    BOUNDS_FORGET(p);
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    hipEvent_t ev = nullptr;
    if (!ctx->pool_events.empty()) { ev = ctx->pool_events.back(); ctx->pool_events.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
    if (ev && hipEventRecord(ev, last_user) != hipSuccess) { (void)hipGetLastError(); pool_event_put(ctx, ev); ev = nullptr; }
    if (!ev) (void)hipStreamSynchronize(last_user);   // no event: the block is idle by the time anybody can take it
    ctx->pool_live -= bytes;
    ctx->pool_free.emplace(bytes, hodor_ctx::PoolBlock{p, ++ctx->pool_seq, ev, last_user});
    ctx->pool_cached += bytes;
    // over the cap: the blocks that have been idle for the longest time leave the cache (a process that has walked
    // through many sizes — a test suite, a size sweep — must not sit on all of them; a prover repeating one shape
    // never gets here).  The block that was just released always stays, whatever its size: it is the one the next
    // call of the same shape asks for (a 2^30 FRI prototype is one block of ~100 GiB — evicting it meant a hipMalloc
    // and a hipFree of that size per commit, 3 s instead of 135 ms).  Evicted blocks are NOT freed here (hipFree drains
    // the device): they wait in pool_zombies for the next call that enters the allocator (pool_alloc's slow path,
    // hodor_ctx_trim, hodor_ctx_synchronize, hodor_ctx_destroy).
    const size_t cap = std::max(ctx->pool_cache_cap, bytes);
    while (ctx->pool_cached > cap) {
        auto victim = ctx->pool_free.end();
        for (auto it = ctx->pool_free.begin(); it != ctx->pool_free.end(); ++it)
            if (it->second.p != p && (victim == ctx->pool_free.end() || it->second.seq < victim->second.seq)) victim = it;
        if (victim == ctx->pool_free.end()) break;
        ctx->pool_cached -= victim->first;
        ctx->pool_zombie_bytes += victim->first;
        ctx->pool_zombies.push_back(victim->second);
        ctx->pool_free.erase(victim);
    }
}
void pool_release(hodor_ctx *ctx, void *p, size_t bytes) { pool_release(ctx, p, bytes, ctx->stream); }
void pool_collect(hodor_ctx *ctx) { pool_reap(ctx); }

// ------------------------------------------------------------------------------------------------
// the objects
// ------------------------------------------------------------------------------------------------
namespace {
// one pooled allocation shared by the handles that are views into it (the outputs of a batched LDE / commit)
struct Slab {
    hodor_ctx *ctx;
    void *p;
    size_t bytes;
    std::atomic<int> refs;
};
Slab *slab_new(hodor_ctx *ctx, size_t bytes, int *rc)
{
    void *p = nullptr;
    size_t got = 0;
    *rc = pool_alloc(ctx, bytes, &p, &got);
    if (*rc) return nullptr;
    Slab *s = new (std::nothrow) Slab{ctx, p, got, {1}};
    if (!s) { pool_release(ctx, p, got); *rc = HODOR_ERR_INVALID; }
    return s;
}
void slab_unref(Slab *s)
{
    if (s && s->refs.fetch_sub(1) == 1) {
        pool_release(s->ctx, s->p, s->bytes);
        delete s;
    }
}
}  // namespace

// The host image of a polynomial (as_ref() / as_mut()): heap memory below 1 MiB, a PINNED allocation recycled through the
// context (ctx.hpp: host_free) from there up.  Contents are undefined after reserve().
struct HostImage {
    hodor_ctx *ctx = nullptr;
    hodor_fr *p = nullptr;
    size_t n = 0;              // elements the image holds
    size_t bytes = 0;          // allocation size
    bool pinned = false;
    hodor_fr *data() const { return p; }
    int reserve(hodor_ctx *c, size_t elems)
    {
        const size_t want = elems * sizeof(hodor_fr);
        if (p && want <= bytes) { n = elems; return HODOR_OK; }
        release();
        ctx = c;
        if (want < ((size_t)1 << 20)) {
            p = (hodor_fr *)malloc(want ? want : 32);
            if (!p) return HODOR_ERR_INVALID;
            pinned = false;
        } else {
            void *q = nullptr;
            {
                std::lock_guard<std::mutex> lk(c->host_mu);
                auto it = c->host_free.find(want);
                if (it != c->host_free.end()) { q = it->second; c->host_cached -= want; c->host_free.erase(it); }
            }
            if (!q && pinned_malloc(&q, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return HODOR_ERR_DEVICE;    // the runtime has no pinned memory for a copy of the polynomial
            }
            p = (hodor_fr *)q;
            pinned = true;
        }
        bytes = want;
        n = elems;
        return HODOR_OK;
    }
    void release()
    {
        if (!p) return;
        if (!pinned) free(p);
        else {
            bool keep = false;
            {
                std::lock_guard<std::mutex> lk(ctx->host_mu);
                if (ctx->host_cached + bytes <= hodor_ctx::HOST_CACHE_CAP) {
                    ctx->host_free.emplace(bytes, (void *)p);
                    ctx->host_cached += bytes;
                    keep = true;
                }
            }
            if (!keep) (void)hipHostFree(p);
        }
        p = nullptr;
        n = bytes = 0;
    }
    ~HostImage() { release(); }
    HostImage() = default;
    HostImage(const HostImage &) = delete;
    HostImage &operator=(const HostImage &) = delete;
};

void host_images_drain(hodor_ctx *ctx)   // hodor_ctx_destroy / hodor_ctx_trim
{
    std::multimap<size_t, void *> blocks;
    {
        std::lock_guard<std::mutex> lk(ctx->host_mu);
        blocks.swap(ctx->host_free);
        ctx->host_cached = 0;
    }
    for (auto &b : blocks) (void)hipHostFree(b.second);
}

struct hodor_poly {
    hodor_ctx *ctx = nullptr;
    int form = HODOR_FORM_COEFFICIENTS;
    Slab *slab = nullptr;
    size_t off = 0;            // byte offset of element 0 in the slab
    size_t n = 0;              // elements, a power of two
    uint32_t exp = 0;          // :28-33
    HFr omega, omegainv, geninv, minv;
    HostImage host;            // as_ref() / as_mut(): materialised on demand
    bool host_valid = false;   // the image equals the vector
    // as_mut() (:46) handed the image out as `&mut [F]`: from then on the IMAGE is the vector and the device copy is
    // stale, until the next operation that needs the device copy — or hodor_poly_commit_mut_h — writes it back
    bool host_dirty = false;
    bool all_zero = false;     // new_for_size and nothing since: the image of a large zero polynomial needs no download
    uint4 *d() const { return (uint4 *)((uint8_t *)slab->p + off); }
    hodor_fr *dfr() const { return (hodor_fr *)d(); }
    void *stream() const { return (void *)ctx->stream; }
    void touch() { host_valid = false; all_zero = false; }
};

struct hodor_iop {
    hodor_ctx *ctx = nullptr;
    int combiner = HODOR_COMBINER_TRIVIAL;
    size_t n = 0;              // committed values
    Slab *slab = nullptr;
    size_t off = 0;
    size_t entries = 0;        // heap entries: n (TRIVIAL) or n / 2 (COSET2)
    uint8_t root[32];
    bool root_valid = false;
    uint8_t *raw = nullptr;    // a VIEW of a tree some other object owns (a FRI prototype's commitments): slab == nullptr
    uint8_t *d() const { return slab ? (uint8_t *)slab->p + off : raw; }
};

// write the host image back after as_mut() (one upload of the whole vector; synchronous: the caller may take the next
// as_mut() and scribble on the image as soon as this returns)
static int poly_flush(hodor_poly *p)
{
    if (!p->host_dirty) return HODOR_OK;
    hodor_ctx *ctx = p->ctx;
    hipError_t e = hipSuccess;
    if (p->n <= 4) {   // q_poly.as_mut()[1] = F::one() (src/ali/per_register/mod.rs:199-202, deep.rs:61-62): from kernel arguments
        Fr v[4];
        for (size_t i = 0; i < p->n; i++) v[i] = to_dev(to_h(&p->host.data()[i]));
        e = store_elems_launch(ctx->stream, p->d(), v, (uint32_t)p->n);
    } else {
        HostXfer xfer(ctx, ctx->stream);
        e = xfer.h2d(p->d(), p->host.data(), p->n * 32);
        if (e == hipSuccess) e = xfer.finish();
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (e != hipErrorAssert) set_err(ctx, std::string("as_mut write-back: ") + hipGetErrorString(e));
        return HODOR_ERR_DEVICE;
    }
    p->host_dirty = false;
    p->host_valid = true;      // image and device copy agree again
    p->all_zero = false;
    return HODOR_OK;
}
// an operand whose device copy is about to be read or written: pending as_mut() writes go in first
#define POLY_SYNC(q)                                                                  \
    do {                                                                              \
        if ((q)->host_dirty) {                                                        \
            int rc_sync__ = poly_flush(const_cast<hodor_poly *>(q));                  \
            if (rc_sync__) return rc_sync__;                                          \
        }                                                                             \
    } while (0)
#define POLY_ENTRY_HOST(p)   /* entry points that work on the host image only */       \
    if (!(p)) return HODOR_ERR_INVALID;                                               \
    hodor_ctx *ctx = (p)->ctx;                                                        \
    NEED_DEVICE()
#define POLY_ENTRY(p)                                                                 \
    POLY_ENTRY_HOST(p);                                                               \
    POLY_SYNC(p)
#define NEED_FORM(p, f)                                                               \
    do {                                                                              \
        if ((p)->form != (f)) {                                                       \
            set_err(ctx, (f) == HODOR_FORM_VALUES ? "this method exists on Polynomial<F, Values> only"      \
                                                   : "this method exists on Polynomial<F, Coefficients> only"); \
            return HODOR_ERR_INVALID;                                                 \
        }                                                                             \
    } while (0)

// exp / omega / omegainv / geninv / minv of a polynomial of n elements (from_coeffs :150-163)
static int poly_set_domain(hodor_ctx *ctx, hodor_poly *p)
{
    uint64_t size;
    uint32_t k;
    HFr w;
    if (!ctx->F.domain(p->n, &size, &k, &w) || size != p->n) {
        set_err(ctx, "polynomial size exceeds the field's two-adicity");   // SynthesisError::Error, src/domains/mod.rs:30-32
        return HODOR_ERR_SIZE;
    }
    p->exp = k;
    p->omega = w;
    ctx->F.inverse(w, &p->omegainv);
    ctx->F.inverse(ctx->F.generator, &p->geninv);
    ctx->F.inverse(ctx->F.from_u64(p->n), &p->minv);
    return HODOR_OK;
}

// a polynomial of `padded` elements (a power of two, checked against the two-adicity) over a fresh slab
static int poly_make(hodor_ctx *ctx, int form, size_t padded, hodor_poly **out)
{
    if (form != HODOR_FORM_COEFFICIENTS && form != HODOR_FORM_VALUES) return HODOR_ERR_INVALID;
    hodor_poly *p = new (std::nothrow) hodor_poly();
    if (!p) return HODOR_ERR_INVALID;
    p->ctx = ctx;
    p->form = form;
    p->n = padded;
    int rc = poly_set_domain(ctx, p);
    if (!rc) p->slab = slab_new(ctx, padded * 32, &rc);
    if (rc) { delete p; return rc; }
    ctx->live_handles.fetch_add(1);
    *out = p;
    return HODOR_OK;
}

static size_t next_pow2(size_t len)   // Domain::new_for_size rounds up; 0 and 1 give the size-1 domain
{
    size_t m = 1;
    while (m < len) m <<= 1;
    return m;
}

// give `p` a fresh slab of new_n elements (contents undefined) and hand back the old one to be released by the caller
// AFTER it has enqueued whatever still reads it
static int poly_swap_storage(hodor_poly *p, size_t new_n, Slab **old, size_t *old_off)
{
    int rc = HODOR_OK;
    Slab *s = slab_new(p->ctx, new_n * 32, &rc);
    if (rc) return rc;
    *old = p->slab;
    *old_off = p->off;
    p->slab = s;
    p->off = 0;
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// context-level
// ------------------------------------------------------------------------------------------------
extern "C" void *hodor_ctx_stream(hodor_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" uint64_t hodor_ctx_host_round_trips(const hodor_ctx *ctx) { return ctx ? ctx->host_round_trips.load() : 0; }
extern "C" void hodor_ctx_reset_host_round_trips(hodor_ctx *ctx)
{
    if (!ctx) return;
    ctx->host_round_trips.store(0);
    ctx->h2d_bytes.store(0);
    ctx->d2h_bytes.store(0);
}
extern "C" int hodor_ctx_trim(hodor_ctx *ctx)
{
    NEED_DEVICE();
    pool_drain(ctx);
    host_images_drain(ctx);
    return HODOR_OK;
}
// the largest number of pool bytes that were live (handed out) at one time since creation / the last reset: what a run NEEDS,
// as opposed to what the pool has kept cached since
extern "C" size_t hodor_ctx_pool_peak(hodor_ctx *ctx, int reset)
{
    if (!ctx) return 0;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    const size_t peak = ctx->pool_peak_live;
    if (reset) ctx->pool_peak_live = ctx->pool_live;
    return peak;
}

extern "C" int hodor_ctx_pool_stats(const hodor_ctx *ctx_, size_t *cached, size_t *live)
{
    hodor_ctx *ctx = const_cast<hodor_ctx *>(ctx_);
    if (!ctx) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    if (cached) *cached = ctx->pool_cached + ctx->pool_zombie_bytes;   // evicted blocks are still ours until collected
    if (live) *live = ctx->pool_live;
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// constructors, accessors
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_poly_new_for_size_h(hodor_ctx *ctx, int form, size_t size, hodor_poly **out)
{
    NEED_DEVICE();
    if (!out) return HODOR_ERR_INVALID;
    hodor_poly *p = nullptr;
    int rc = poly_make(ctx, form, next_pow2(size), &p);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(p->d(), 0, p->n * 32, ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err(ctx, std::string("new_for_size: ") + hipGetErrorString(e));
        hodor_poly_free_h(p);
        return HODOR_ERR_DEVICE;
    }
    if (p->n <= 64 && p->host.reserve(ctx, p->n) == HODOR_OK) {   // a small zero polynomial is known on the host too
        memset(p->host.data(), 0, p->n * 32);
        p->host_valid = true;
    }
    p->all_zero = true;
    *out = p;
    return HODOR_OK;
}

extern "C" int hodor_poly_from_host_h(hodor_ctx *ctx, int form, const hodor_fr *host, size_t len, hodor_poly **out)
{
    NEED_DEVICE();
    if (!out || (!host && len)) return HODOR_ERR_INVALID;
    hodor_poly *p = nullptr;
    int rc = poly_make(ctx, form, next_pow2(len), &p);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (p->n > len) e = hipMemsetAsync((uint8_t *)p->d() + len * 32, 0, (p->n - len) * 32, ctx->stream);
    if (e == hipSuccess && len) {
        if (len <= 4) {   // from kernel arguments: no host buffer has to outlive the call
            Fr v[4];
            for (size_t i = 0; i < len; i++) v[i] = to_dev(to_h(&host[i]));
            e = store_elems_launch(ctx->stream, p->d(), v, (uint32_t)len);
        } else {
            HostXfer xfer(ctx, ctx->stream);   // small vectors through the pinned buffer, whole ones straight from the caller's
            e = xfer.h2d(p->d(), host, len * 32);
            if (e == hipSuccess) e = xfer.finish();   // `host` is the caller's: done with it on return
        }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (e != hipErrorAssert) set_err(ctx, std::string("from_host: ") + hipGetErrorString(e));
        hodor_poly_free_h(p);
        return HODOR_ERR_DEVICE;
    }
    if (p->n <= 64 && p->host.reserve(ctx, p->n) == HODOR_OK) {   // small polynomials (the degree-one q(x) of calculate_deep) stay readable without a round trip
        memset(p->host.data(), 0, p->n * 32);
        if (len) memcpy(p->host.data(), host, len * 32);
        p->host_valid = true;
    }
    *out = p;
    return HODOR_OK;
}

extern "C" int hodor_poly_from_dev_h(hodor_ctx *ctx, int form, const hodor_fr *dev_src, size_t len, void *producer_stream,
                                     hodor_poly **out)
{
    NEED_DEVICE();
    if (!out || (!dev_src && len)) return HODOR_ERR_INVALID;
    hodor_poly *p = nullptr;
    int rc = poly_make(ctx, form, next_pow2(len), &p);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if ((hipStream_t)producer_stream != ctx->stream) {   // order the copy behind the producer
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev, (hipStream_t)producer_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ev, 0);
        if (ev) (void)hipEventDestroy(ev);   // a recorded event may be destroyed: the wait keeps what it needs
    }
    if (e == hipSuccess && p->n > len) e = hipMemsetAsync((uint8_t *)p->d() + len * 32, 0, (p->n - len) * 32, ctx->stream);
    if (e == hipSuccess && len) e = hipMemcpyAsync(p->d(), dev_src, len * 32, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err(ctx, std::string("from_dev: ") + hipGetErrorString(e));
        hodor_poly_free_h(p);
        return HODOR_ERR_DEVICE;
    }
    *out = p;
    return HODOR_OK;
}

extern "C" int hodor_poly_gen_h(hodor_ctx *ctx, int form, uint64_t first_index, size_t count, uint64_t seed,
                                hodor_poly **out)
{
    NEED_DEVICE();
    if (!out) return HODOR_ERR_INVALID;
    hodor_poly *p = nullptr;
    int rc = poly_make(ctx, form, next_pow2(count), &p);
    if (rc) return rc;
    if (p->n > count && hipMemsetAsync((uint8_t *)p->d() + count * 32, 0, (p->n - count) * 32, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        hodor_poly_free_h(p);
        return HODOR_ERR_DEVICE;
    }
    rc = hodor_gen_elements_dev(ctx, p->stream(), p->dfr(), first_index, count, seed);
    if (rc) { hodor_poly_free_h(p); return rc; }
    *out = p;
    return HODOR_OK;
}

extern "C" int hodor_poly_clone_h(const hodor_poly *src, hodor_poly **out)
{
    POLY_ENTRY(src);
    if (!out) return HODOR_ERR_INVALID;
    hodor_poly *p = nullptr;
    int rc = poly_make(ctx, src->form, src->n, &p);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(p->d(), src->d(), src->n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    if (src->host_valid && src->n <= 64 && p->host.reserve(ctx, src->n) == HODOR_OK) {   // small ones stay known on the host
        memcpy(p->host.data(), src->host.data(), src->n * 32);
        p->host_valid = true;
    }
    p->all_zero = src->all_zero;
    *out = p;
    return HODOR_OK;
}

extern "C" void hodor_poly_free_h(hodor_poly *p)
{
    if (!p) return;
    slab_unref(p->slab);   // back to the pool: whatever is still enqueued on it runs before the next user (stream order)
    p->ctx->live_handles.fetch_sub(1);
    delete p;
}

extern "C" size_t hodor_poly_size_h(const hodor_poly *p) { return p ? p->n : 0; }
extern "C" int hodor_poly_form_h(const hodor_poly *p) { return p ? p->form : -1; }
extern "C" void *hodor_poly_dev_ptr_h(hodor_poly *p)
{
    if (!p) return nullptr;
    if (p->host_dirty && poly_flush(p)) return nullptr;   // pending as_mut() writes first
    p->touch();            // the caller may write through it
    return p->d();
}
extern "C" int hodor_poly_info_h(const hodor_poly *p, hodor_poly_info *out)
{
    if (!p || !out) return HODOR_ERR_INVALID;
    out->exp = p->exp;
    from_h(p->omega, &out->omega);
    from_h(p->omegainv, &out->omegainv);
    from_h(p->geninv, &out->geninv);
    from_h(p->minv, &out->minv);
    return HODOR_OK;
}

// the host image brought up to date (as_ref, as_mut): nothing to do while it is valid or dirty (then it IS the vector)
static int poly_materialise(hodor_poly *p)
{
    hodor_ctx *ctx = p->ctx;
    if (p->host_valid || p->host_dirty) return HODOR_OK;
    if (int rc = p->host.reserve(ctx, p->n)) {   // no exception crosses the ABI: a host copy of a vector that large is simply refused
        set_err(ctx, "as_ref / as_mut: no host memory for a copy of the polynomial");
        return rc;                               // HODOR_ERR_DEVICE: hipHostMalloc said no; HODOR_ERR_INVALID: malloc did
    }
    if (p->all_zero) {   // Polynomial::new_for_size(..) followed by as_mut() (src/ali/per_register/mod.rs:112-118): zeros need no PCIe
        memset(p->host.data(), 0, p->n * 32);
    } else {
        HostXfer xfer(ctx, ctx->stream);
        HIPCHK(xfer.d2h(p->host.data(), p->d(), p->n * 32));
        HIPCHK(xfer.finish());
        note_round_trip(ctx);
    }
    p->host_valid = true;
    return HODOR_OK;
}

extern "C" int hodor_poly_as_ref_h(hodor_poly *p, const hodor_fr **host)
{
    POLY_ENTRY_HOST(p);
    if (!host) return HODOR_ERR_INVALID;
    int rc = poly_materialise(p);
    if (rc) return rc;
    *host = p->host.data();
    return HODOR_OK;
}

// as_mut() :46 — the whole vector as `&mut [F]` on the host.  The image is materialised like as_ref()'s (one download,
// none for a polynomial that is still new_for_size's zeros) and is THE vector from now on; it goes back to the device in
// one upload when hodor_poly_commit_mut_h is called or, failing that, before the next operation on the handle that
// needs the device copy.  *host stays valid until the handle is freed or resized.
extern "C" int hodor_poly_as_mut_h(hodor_poly *p, hodor_fr **host)
{
    POLY_ENTRY_HOST(p);
    if (!host) return HODOR_ERR_INVALID;
    int rc = poly_materialise(p);
    if (rc) return rc;
    p->host_dirty = true;
    p->host_valid = false;
    *host = p->host.data();
    return HODOR_OK;
}

// the end of the `&mut` borrow: the image is uploaded now (HODOR_OK and nothing to do when no as_mut() is outstanding)
extern "C" int hodor_poly_commit_mut_h(hodor_poly *p)
{
    POLY_ENTRY_HOST(p);
    return poly_flush(p);
}

extern "C" void hodor_ctx_host_traffic(const hodor_ctx *ctx, uint64_t *h2d_bytes, uint64_t *d2h_bytes)
{
    if (h2d_bytes) *h2d_bytes = ctx ? ctx->h2d_bytes.load() : 0;
    if (d2h_bytes) *d2h_bytes = ctx ? ctx->d2h_bytes.load() : 0;
}

extern "C" int hodor_poly_read_h(hodor_poly *p, size_t first, size_t count, hodor_fr *out)
{
    POLY_ENTRY_HOST(p);
    if (!out && count) return HODOR_ERR_INVALID;
    if (first > p->n || count > p->n - first) return HODOR_ERR_SIZE;   // the slice index panics
    if (count == 0) return HODOR_OK;
    if (p->host_valid || p->host_dirty) {
        memcpy(out, p->host.data() + first, count * 32);
        return HODOR_OK;
    }
    HostXfer xfer(ctx, ctx->stream);
    HIPCHK(xfer.d2h(out, p->dfr() + first, count * 32));
    HIPCHK(xfer.finish());
    note_round_trip(ctx);
    return HODOR_OK;
}

extern "C" int hodor_poly_write_h(hodor_poly *p, size_t first, size_t count, const hodor_fr *in)
{
    POLY_ENTRY_HOST(p);
    if (!in && count) return HODOR_ERR_INVALID;
    if (first > p->n || count > p->n - first) return HODOR_ERR_SIZE;
    if (count == 0) return HODOR_OK;
    if (p->host_dirty) {   // an as_mut() is outstanding: the image is the vector
        memcpy(p->host.data() + first, in, count * 32);
        return HODOR_OK;
    }
    p->all_zero = false;
    if (count <= 4) {
        Fr v[4];
        for (size_t i = 0; i < count; i++) v[i] = to_dev(to_h(&in[i]));
        HIPCHK(store_elems_launch(ctx->stream, p->d() + 2 * first, v, (uint32_t)count));
    } else {
        HostXfer xfer(ctx, ctx->stream);
        HIPCHK(xfer.h2d(p->dfr() + first, in, count * 32));
        HIPCHK(xfer.finish());
    }
    if (p->host_valid) memcpy(p->host.data() + first, in, count * 32);   // the copy follows the write
    return HODOR_OK;
}

extern "C" int hodor_poly_elem_op_h(hodor_poly *p, size_t index, int op, const hodor_fr *c, uint64_t e)
{
    POLY_ENTRY(p);
    if (index >= p->n) return HODOR_ERR_SIZE;
    int rc = hodor_poly_unary_dev(ctx, p->stream(), p->dfr() + index, 1, op, c, e);
    if (rc) return rc;
    p->all_zero = false;
    if (p->host_valid) {   // the same operation on the host copy (src/ali/per_register/deep.rs:62 works on a 2-element q)
        const HostField &F = ctx->F;
        HFr v = to_h(&p->host.data()[index]), k = c ? to_h(c) : F.one;
        switch (op) {
        case HODOR_UN_NEGATE: v = F.sub(HFr{{0, 0, 0, 0}}, v); break;
        case HODOR_UN_SQUARE: v = F.mul(v, v); break;
        case HODOR_UN_POW: v = F.pow(v, e); break;
        case HODOR_UN_SCALE: v = F.mul(v, k); break;
        case HODOR_UN_ADD_CONSTANT: v = F.add(v, k); break;
        case HODOR_UN_SUB_CONSTANT: v = F.sub(v, k); break;
        default: p->touch(); return HODOR_OK;
        }
        from_h(v, &p->host.data()[index]);
    }
    return HODOR_OK;
}

extern "C" int hodor_poly_equal_h(const hodor_poly *a, const hodor_poly *b, int *equal)
{
    POLY_ENTRY(a);
    if (!b || !equal || b->ctx != ctx) return HODOR_ERR_INVALID;
    POLY_SYNC(b);
    *equal = 0;
    if (a->form != b->form || a->n != b->n) return HODOR_OK;
    void *flag = nullptr;
    size_t got = 0;
    int rc = pool_alloc(ctx, 256, &flag, &got);
    if (rc) return rc;
    uint32_t host_flag = 1;
    hipError_t e = hipMemsetAsync(flag, 0, 4, ctx->stream);
    if (e == hipSuccess) e = count_diff_launch(ctx->stream, a->d(), b->d(), a->n, (uint32_t *)flag);
    {
        HostXfer xfer(ctx, ctx->stream);
        if (e == hipSuccess) e = xfer.d2h(&host_flag, flag, 4);
        if (e == hipSuccess) e = xfer.finish();
    }
    pool_release(ctx, flag, got);
    HIPCHK(e);
    note_round_trip(ctx);
    *equal = host_flag == 0;
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// generic methods (:54-137)
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_poly_distribute_powers_h(hodor_poly *p, const hodor_fr *g)
{
    POLY_ENTRY(p);
    p->touch();
    return hodor_distribute_powers_dev(ctx, p->stream(), p->dfr(), p->n, g);
}
extern "C" int hodor_poly_scale_h(hodor_poly *p, const hodor_fr *g)
{
    POLY_ENTRY(p);
    p->touch();
    return hodor_poly_unary_dev(ctx, p->stream(), p->dfr(), p->n, HODOR_UN_SCALE, g, 0);
}
extern "C" int hodor_poly_negate_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    p->touch();
    return hodor_poly_unary_dev(ctx, p->stream(), p->dfr(), p->n, HODOR_UN_NEGATE, nullptr, 0);
}

// coeffs.resize(new_n, F::zero()) + the domain constants (:95-104, :115-124)
static int poly_resize(hodor_poly *p, size_t new_n)
{
    hodor_ctx *ctx = p->ctx;
    if (new_n == p->n) return HODOR_OK;
    const size_t old_n = p->n;
    hodor_poly probe;            // check the domain before touching anything: an Err must not leave a half-resized
    probe.n = new_n;             // polynomial behind
    int rc = poly_set_domain(ctx, &probe);
    if (rc) return rc;
    Slab *old = nullptr;
    size_t old_off = 0;
    if ((rc = poly_swap_storage(p, new_n, &old, &old_off))) return rc;
    const size_t keep = old_n < new_n ? old_n : new_n;
    hipError_t e = hipMemcpyAsync(p->d(), (uint8_t *)old->p + old_off, keep * 32, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && new_n > keep) e = hipMemsetAsync((uint8_t *)p->d() + keep * 32, 0, (new_n - keep) * 32, ctx->stream);
    slab_unref(old);
    p->n = new_n;
    p->exp = probe.exp;
    p->omega = probe.omega;
    p->omegainv = probe.omegainv;
    p->minv = probe.minv;
    p->touch();
    HIPCHK(e);
    return HODOR_OK;
}

extern "C" int hodor_poly_pad_by_factor_h(hodor_poly *p, size_t factor)
{
    POLY_ENTRY(p);
    if (factor == 1) return HODOR_OK;                                      // :86-88
    if (!is_pow2(factor)) return HODOR_ERR_SIZE;                           // Err(SynthesisError::Error) :89-92
    return poly_resize(p, p->n * factor);
}
extern "C" int hodor_poly_pad_to_size_h(hodor_poly *p, size_t new_size)
{
    POLY_ENTRY(p);
    if (new_size < p->n || !is_pow2(new_size)) return HODOR_ERR_SIZE;      // :108-114
    return poly_resize(p, new_size);
}
extern "C" int hodor_poly_trim_to_degree_h(hodor_poly *p, size_t degree)
{
    POLY_ENTRY(p);
    if (degree >= p->n - 1) return HODOR_OK;                               // size <= degree + 1 :129-131 (without the overflow)
    p->touch();
    HIPCHK(hipMemsetAsync(p->dfr() + degree + 1, 0, (p->n - degree - 1) * 32, ctx->stream));   // truncate + resize :132-133
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------------------------------
// out of place into a fresh pooled buffer, then the handle takes the new storage: the multi-pass transform ping-pongs
// between its destination and ONE scratch buffer (an in-place call needs two)
template <class Call>
static int poly_retransform(hodor_poly *p, int from, int to, Call call)
{
    hodor_ctx *ctx = p->ctx;
    NEED_FORM(p, from);
    Slab *old = nullptr;
    size_t old_off = 0;
    int rc = poly_swap_storage(p, p->n, &old, &old_off);
    if (rc) return rc;
    rc = call((const hodor_fr *)((uint8_t *)old->p + old_off), p->dfr());
    if (rc) {   // nothing was transformed: the handle keeps its old storage
        slab_unref(p->slab);
        p->slab = old;
        p->off = old_off;
        return rc;
    }
    slab_unref(old);
    p->form = to;
    p->touch();
    return HODOR_OK;
}

extern "C" int hodor_poly_fft_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    return poly_retransform(p, HODOR_FORM_COEFFICIENTS, HODOR_FORM_VALUES, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_fft_dev(ctx, p->stream(), s, d, p->exp);
    });
}
extern "C" int hodor_poly_coset_fft_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    return poly_retransform(p, HODOR_FORM_COEFFICIENTS, HODOR_FORM_VALUES, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_coset_fft_dev(ctx, p->stream(), s, d, p->exp);
    });
}
extern "C" int hodor_poly_coset_fft_for_generator_h(hodor_poly *p, const hodor_fr *gen)
{
    POLY_ENTRY(p);
    if (!gen) return HODOR_ERR_INVALID;
    return poly_retransform(p, HODOR_FORM_COEFFICIENTS, HODOR_FORM_VALUES, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_coset_fft_for_generator_dev(ctx, p->stream(), s, d, p->exp, gen);
    });
}
extern "C" int hodor_poly_ifft_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    return poly_retransform(p, HODOR_FORM_VALUES, HODOR_FORM_COEFFICIENTS, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_ifft_dev(ctx, p->stream(), s, d, p->exp);
    });
}
extern "C" int hodor_poly_icoset_fft_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    return poly_retransform(p, HODOR_FORM_VALUES, HODOR_FORM_COEFFICIENTS, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_icoset_fft_dev(ctx, p->stream(), s, d, p->exp);
    });
}
extern "C" int hodor_poly_icoset_fft_for_generator_h(hodor_poly *p, const hodor_fr *geninv)
{
    POLY_ENTRY(p);
    if (!geninv) return HODOR_ERR_INVALID;
    return poly_retransform(p, HODOR_FORM_VALUES, HODOR_FORM_COEFFICIENTS, [&](const hodor_fr *s, hodor_fr *d) {
        return hodor_poly_icoset_fft_for_generator_dev(ctx, p->stream(), s, d, p->exp, geninv);
    });
}

extern "C" int hodor_poly_lde_h(const hodor_poly *p, size_t factor, int coset, hodor_poly **out)
{
    POLY_ENTRY(p);
    if (!out) return HODOR_ERR_INVALID;
    NEED_FORM(p, HODOR_FORM_COEFFICIENTS);
    if (!is_pow2(factor)) { set_err(ctx, "lde factor must be a power of two"); return HODOR_ERR_SIZE; }   // assert :434
    hodor_poly *q = nullptr;
    int rc = poly_make(ctx, HODOR_FORM_VALUES, p->n * factor, &q);
    if (rc) return rc;
    rc = hodor_poly_lde_dev(ctx, p->stream(), p->dfr(), q->dfr(), p->exp, factor, coset);
    if (rc) { hodor_poly_free_h(q); return rc; }
    *out = q;
    return HODOR_OK;
}

extern "C" int hodor_poly_lde_batch_h(const hodor_poly *const *ps, size_t count, size_t factor, int coset,
                                      hodor_poly **outs)
{
    if (!ps || !outs || count == 0 || !ps[0]) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = ps[0]->ctx;
    NEED_DEVICE();
    if (!is_pow2(factor)) { set_err(ctx, "lde factor must be a power of two"); return HODOR_ERR_SIZE; }
    if (count > 65535) return HODOR_ERR_SIZE;
    const size_t n = ps[0]->n, big = n * factor;
    bool contiguous = true;
    for (size_t i = 0; i < count; i++) {
        if (!ps[i] || ps[i]->ctx != ctx || ps[i]->n != n) return HODOR_ERR_INVALID;
        POLY_SYNC(ps[i]);
        NEED_FORM(ps[i], HODOR_FORM_COEFFICIENTS);
        // back to back INSIDE ONE allocation (views of a batched result): two pool blocks that merely happen to be
        // neighbours in the address space are two buffers — a launch must not run across the seam (found by the handle
        // fuzzer on the bounds build, round 6)
        if (ps[i]->slab != ps[0]->slab || ps[i]->off != ps[0]->off + i * n * 32) contiguous = false;
    }
    {   // the size-n*factor domain must exist before anything is allocated
        HFr w;
        int rc = poly_domain(ctx, log2u(big), &w);
        if (rc) return rc;
    }
    int rc = HODOR_OK;
    Slab *dst = slab_new(ctx, count * big * 32, &rc);
    if (rc) return rc;
    const hodor_fr *src = ps[0]->dfr();
    Slab *gather = nullptr;
    if (!contiguous) {   // the batched transform wants its inputs back to back
        gather = slab_new(ctx, count * n * 32, &rc);
        if (rc) { slab_unref(dst); return rc; }
        for (size_t i = 0; i < count && rc == HODOR_OK; i++)
            if (hipMemcpyAsync((uint8_t *)gather->p + i * n * 32, ps[i]->d(), n * 32, hipMemcpyDeviceToDevice, ctx->stream) !=
                hipSuccess) {
                (void)hipGetLastError();
                rc = HODOR_ERR_DEVICE;
            }
        src = (const hodor_fr *)gather->p;
    }
    if (!rc) rc = hodor_poly_lde_batch_dev(ctx, (void *)ctx->stream, src, (hodor_fr *)dst->p, ps[0]->exp, factor, coset, count);
    if (gather) slab_unref(gather);
    if (rc) { slab_unref(dst); return rc; }
    for (size_t i = 0; i < count; i++) {
        hodor_poly *q = new (std::nothrow) hodor_poly();
        if (!q) {
            for (size_t k = 0; k < i; k++) hodor_poly_free_h(outs[k]);
            slab_unref(dst);
            return HODOR_ERR_INVALID;
        }
        q->ctx = ctx;
        q->form = HODOR_FORM_VALUES;
        q->n = big;
        (void)poly_set_domain(ctx, q);
        q->slab = dst;
        q->off = i * big * 32;
        dst->refs.fetch_add(1);
        ctx->live_handles.fetch_add(1);
        outs[i] = q;
    }
    slab_unref(dst);   // the views hold it now
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// arithmetic
// ------------------------------------------------------------------------------------------------
static int binary_len(hodor_ctx *ctx, const hodor_poly *a, const hodor_poly *b, int op, size_t *len)
{
    if (a->form != b->form) { set_err(ctx, "operands of different polynomial forms"); return HODOR_ERR_INVALID; }
    if (a->form == HODOR_FORM_VALUES) {
        if (a->n != b->n) { set_err(ctx, "value-form operands of different sizes"); return HODOR_ERR_SIZE; }   // assert_eq! :818
    } else {
        if (op == HODOR_OP_MUL) { set_err(ctx, "mul_assign exists on Polynomial<F, Values> only"); return HODOR_ERR_INVALID; }
        if (a->n < b->n) { set_err(ctx, "coefficient-form operand longer than self"); return HODOR_ERR_SIZE; }   // assert! :641
    }
    *len = b->n;
    return HODOR_OK;
}

extern "C" int hodor_poly_binary_h(hodor_poly *a, const hodor_poly *b, int op)
{
    POLY_ENTRY(a);
    if (!b || b->ctx != ctx) return HODOR_ERR_INVALID;
    POLY_SYNC(b);
    size_t len;
    int rc = binary_len(ctx, a, b, op, &len);
    if (rc) return rc;
    a->touch();
    return hodor_poly_binary_dev(ctx, a->stream(), a->dfr(), b->dfr(), len, op);
}

extern "C" int hodor_poly_add_assign_scaled_h(hodor_poly *a, const hodor_poly *b, const hodor_fr *scaling)
{
    POLY_ENTRY(a);
    if (!b || b->ctx != ctx || !scaling) return HODOR_ERR_INVALID;
    POLY_SYNC(b);
    size_t len;
    int rc = binary_len(ctx, a, b, HODOR_OP_ADD, &len);
    if (rc) return rc;
    a->touch();
    return hodor_poly_add_scaled_dev(ctx, a->stream(), a->dfr(), b->dfr(), len, scaling);
}

extern "C" int hodor_poly_evaluate_at_h(hodor_poly *p, const hodor_fr *g, hodor_fr *out)
{
    POLY_ENTRY(p);
    NEED_FORM(p, HODOR_FORM_COEFFICIENTS);
    return hodor_poly_evaluate_at_dev(ctx, p->stream(), p->dfr(), p->n, g, out);
}

extern "C" int hodor_poly_degree_one_on_domain_h(hodor_ctx *ctx, size_t n, const hodor_fr *alpha, const hodor_fr *c,
                                                 int coset, hodor_poly **out)
{
    NEED_DEVICE();
    if (!out || !alpha || !c) return HODOR_ERR_INVALID;
    if (!is_pow2(n)) { set_err(ctx, "degree_one_on_domain: n must be a power of two"); return HODOR_ERR_SIZE; }
    hodor_poly *q = nullptr;
    int rc = poly_make(ctx, HODOR_FORM_VALUES, n, &q);
    if (rc) return rc;
    rc = hodor_poly_degree_one_on_domain_dev(ctx, (void *)ctx->stream, q->dfr(), n, alpha, c, coset);
    if (rc) { hodor_poly_free_h(q); return rc; }
    *out = q;
    return HODOR_OK;
}

// ALIInstance::from_arp's divisor precompute on the device (src/ali/per_register/mod.rs:60-160; kernel: pointwise.hip
// k_dense_divisor): out[i] = prod_j (x_i - roots[j]) / (x_i^T - 1), x_i = g w^i, on the coset of the domain of
// `evaluation_size` points; T = column_size.  The n / T distinct values of x^T - 1 on the coset are inverted here on
// the host (HODOR_ERR_INVALID when one of them is zero — the reference's batch_inversion would return
// Err(SynthesisError::Error), :136) and travel with the roots in one small upload.
extern "C" int hodor_poly_dense_divisor_on_coset_dev(hodor_ctx *ctx, void *stream_, hodor_fr *out, size_t evaluation_size,
                                                     size_t column_size, const hodor_fr *roots, size_t n_roots)
{
    NEED_DEVICE();
    if (!out || (!roots && n_roots)) return HODOR_ERR_INVALID;
    uint64_t size;
    uint32_t log_n;
    HFr w;
    if (!is_pow2(evaluation_size) || !is_pow2(column_size) || column_size > evaluation_size ||
        !ctx->F.domain(evaluation_size, &size, &log_n, &w) || size != evaluation_size) {
        set_err(ctx, "dense_divisor_on_coset: sizes must be powers of two within the field's two-adicity, column_size <= evaluation_size");
        return HODOR_ERR_SIZE;
    }
    const size_t period = evaluation_size / column_size;
    if (period > ((size_t)1 << 16) || n_roots > ((size_t)1 << 20)) {
        set_err(ctx, "dense_divisor_on_coset: evaluation_size / column_size <= 2^16 and at most 2^20 roots");
        return HODOR_ERR_SIZE;
    }
    const HostField &F = ctx->F;
    std::vector<hodor_fr> small;
    try {
        small.resize(period + n_roots);
    } catch (...) {
        return HODOR_ERR_INVALID;
    }
    const HFr gT = F.pow(F.generator, column_size), wT = F.pow(w, column_size);   // x_i^T = g^T (w^T)^(i mod period)
    HFr v = gT;
    for (size_t k = 0; k < period; k++) {
        HFr inv;
        if (!F.inverse(F.sub(v, F.one), &inv)) {
            set_err(ctx, "dense_divisor_on_coset: x^T - 1 vanishes on the coset");
            return HODOR_ERR_INVALID;
        }
        from_h(inv, &small[k]);
        v = F.mul(v, wT);
    }
    if (n_roots) memcpy(small.data() + period, roots, n_roots * 32);
    hipStream_t stream = pick_stream(ctx, stream_);
    void *stage = nullptr;
    size_t got = 0;
    int rc = pool_alloc(ctx, small.size() * 32, &stage, &got, stream);
    if (rc) return rc;
    hipError_t e;
    {
        HostXfer xfer(ctx, stream);
        e = xfer.h2d(stage, small.data(), small.size() * 32);
        if (e == hipSuccess)
            e = dense_divisor_launch(stream, (uint4 *)out, evaluation_size, to_dev(w), to_dev(F.generator), (const uint4 *)stage,
                                     (uint32_t)period, (const uint4 *)stage + 2 * period, (uint32_t)n_roots, ctx->P);
        hipError_t e2 = xfer.finish();   // the staging bytes have left the pinned buffer (and `small`): nothing of the caller's is referenced after return
        if (e == hipSuccess) e = e2;
    }
    pool_release(ctx, stage, got, stream);
    HIPCHK(e);
    return HODOR_OK;
}

extern "C" int hodor_poly_dense_divisor_on_coset_h(hodor_ctx *ctx, size_t evaluation_size, size_t column_size,
                                                   const hodor_fr *roots, size_t n_roots, hodor_poly **out)
{
    NEED_DEVICE();
    if (!out) return HODOR_ERR_INVALID;
    if (!is_pow2(evaluation_size)) { set_err(ctx, "dense_divisor_on_coset: evaluation_size must be a power of two"); return HODOR_ERR_SIZE; }
    hodor_poly *q = nullptr;
    int rc = poly_make(ctx, HODOR_FORM_VALUES, evaluation_size, &q);
    if (rc) return rc;
    rc = hodor_poly_dense_divisor_on_coset_dev(ctx, (void *)ctx->stream, q->dfr(), evaluation_size, column_size, roots, n_roots);
    if (rc) { hodor_poly_free_h(q); return rc; }
    *out = q;
    return HODOR_OK;
}

extern "C" int hodor_poly_pow_h(hodor_poly *p, uint64_t e)
{
    POLY_ENTRY(p);
    NEED_FORM(p, HODOR_FORM_VALUES);
    p->touch();
    return hodor_poly_unary_dev(ctx, p->stream(), p->dfr(), p->n, HODOR_UN_POW, nullptr, e);
}
extern "C" int hodor_poly_square_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    NEED_FORM(p, HODOR_FORM_VALUES);
    p->touch();
    return hodor_poly_unary_dev(ctx, p->stream(), p->dfr(), p->n, HODOR_UN_SQUARE, nullptr, 0);
}
extern "C" int hodor_poly_add_constant_h(hodor_poly *p, const hodor_fr *c)
{
    POLY_ENTRY(p);
    NEED_FORM(p, HODOR_FORM_VALUES);
    if (!c) return HODOR_ERR_INVALID;
    p->touch();
    return hodor_poly_unary_dev(ctx, p->stream(), p->dfr(), p->n, HODOR_UN_ADD_CONSTANT, c, 0);
}
extern "C" int hodor_poly_batch_inversion_h(hodor_poly *p)
{
    POLY_ENTRY(p);
    NEED_FORM(p, HODOR_FORM_VALUES);
    p->touch();
    return hodor_poly_batch_inversion_dev(ctx, p->stream(), p->dfr(), p->n);
}
extern "C" int hodor_poly_quotient_term_h(hodor_poly *acc, const hodor_poly *f, const hodor_poly *dinv, const hodor_fr *value,
                                          const hodor_fr *alpha, int accumulate)
{
    POLY_ENTRY(acc);
    if (!f || !dinv || f->ctx != ctx || dinv->ctx != ctx || !value) return HODOR_ERR_INVALID;
    POLY_SYNC(f);
    POLY_SYNC(dinv);
    NEED_FORM(acc, HODOR_FORM_VALUES);
    NEED_FORM(f, HODOR_FORM_VALUES);
    NEED_FORM(dinv, HODOR_FORM_VALUES);
    if (f->n != acc->n || dinv->n != acc->n) return HODOR_ERR_SIZE;
    if (acc->d() == f->d() || acc->d() == dinv->d()) return HODOR_ERR_INVALID;
    acc->touch();
    return hodor_poly_quotient_term_dev(ctx, acc->stream(), acc->dfr(), f->dfr(), dinv->dfr(), acc->n, value, alpha, accumulate);
}

// ------------------------------------------------------------------------------------------------
// IOP
// ------------------------------------------------------------------------------------------------
static size_t iop_entries(size_t n, int combiner) { return combiner == HODOR_COMBINER_COSET2 ? n / 2 : n; }

extern "C" int hodor_iop_create_batch_h(const hodor_poly *const *vs, size_t count, int combiner, hodor_iop **outs)
{
    if (!vs || !outs || count == 0 || !vs[0]) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = vs[0]->ctx;
    NEED_DEVICE();
    if (combiner != HODOR_COMBINER_TRIVIAL && combiner != HODOR_COMBINER_COSET2) return HODOR_ERR_INVALID;
    if (count > 65535) return HODOR_ERR_SIZE;
    const size_t n = vs[0]->n;
    if (n < (combiner == HODOR_COMBINER_COSET2 ? 4u : 2u)) { set_err(ctx, "iop_create: too few leaves"); return HODOR_ERR_SIZE; }
    bool contiguous = true;
    for (size_t i = 0; i < count; i++) {
        if (!vs[i] || vs[i]->ctx != ctx || vs[i]->n != n) return HODOR_ERR_INVALID;
        POLY_SYNC(vs[i]);
        if (vs[i]->slab != vs[0]->slab || vs[i]->off != vs[0]->off + i * n * 32) contiguous = false;   // one allocation, as in lde_batch
    }
    const size_t entries = iop_entries(n, combiner);
    int rc = HODOR_OK;
    Slab *nodes = slab_new(ctx, count * entries * 32, &rc);
    if (rc) return rc;
    if (contiguous) {
        rc = hodor_iop_create_batch_combined_dev(ctx, (void *)ctx->stream, vs[0]->dfr(), n, count, combiner, (uint8_t *)nodes->p);
    } else {
        for (size_t i = 0; i < count && !rc; i++)
            rc = hodor_iop_create_combined_dev(ctx, (void *)ctx->stream, vs[i]->dfr(), n, combiner,
                                               (uint8_t *)nodes->p + i * entries * 32);
    }
    if (rc) { slab_unref(nodes); return rc; }
    for (size_t i = 0; i < count; i++) {
        hodor_iop *t = new (std::nothrow) hodor_iop();
        if (!t) {
            for (size_t k = 0; k < i; k++) hodor_iop_free_h(outs[k]);
            slab_unref(nodes);
            return HODOR_ERR_INVALID;
        }
        t->ctx = ctx;
        t->combiner = combiner;
        t->n = n;
        t->slab = nodes;
        t->off = i * entries * 32;
        t->entries = entries;
        nodes->refs.fetch_add(1);
        ctx->live_handles.fetch_add(1);
        outs[i] = t;
    }
    slab_unref(nodes);
    return HODOR_OK;
}

extern "C" int hodor_iop_create_h(const hodor_poly *values, int combiner, hodor_iop **out)
{
    return hodor_iop_create_batch_h(&values, 1, combiner, out);
}

extern "C" void hodor_iop_free_h(hodor_iop *t)
{
    if (!t) return;
    slab_unref(t->slab);
    t->ctx->live_handles.fetch_sub(1);
    delete t;
}
extern "C" size_t hodor_iop_size_h(const hodor_iop *t) { return t ? t->n : 0; }

extern "C" int hodor_iop_roots_h(hodor_iop *const *ts, size_t count, uint8_t *roots)
{
    if (!ts || !roots || count == 0 || !ts[0]) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = ts[0]->ctx;
    NEED_DEVICE();
    bool pending = false;
    {
        HostXfer xfer(ctx, ctx->stream);
        for (size_t i = 0; i < count; i++) {
            if (!ts[i] || ts[i]->ctx != ctx) return HODOR_ERR_INVALID;
            if (!ts[i]->root_valid) {
                HIPCHK(xfer.d2h(ts[i]->root, ts[i]->d() + 32, 32));   // nodes[1]
                pending = true;
            }
        }
        HIPCHK(xfer.finish());
    }
    if (pending) note_round_trip(ctx);
    for (size_t i = 0; i < count; i++) {
        ts[i]->root_valid = true;
        memcpy(roots + 32 * i, ts[i]->root, 32);
    }
    return HODOR_OK;
}
extern "C" int hodor_iop_root_h(hodor_iop *t, uint8_t root[32]) { return hodor_iop_roots_h(&t, 1, root); }

extern "C" int hodor_iop_nodes_h(hodor_iop *t, uint8_t *nodes)
{
    if (!t || !nodes) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = t->ctx;
    NEED_DEVICE();
    HostXfer xfer(ctx, ctx->stream);
    HIPCHK(xfer.d2h(nodes, t->d(), t->entries * 32));
    HIPCHK(xfer.finish());
    note_round_trip(ctx);
    return HODOR_OK;
}

extern "C" int hodor_iop_query_h(hodor_iop *t, const hodor_poly *values, size_t natural_index, hodor_fr *values_out,
                                 uint8_t *path, size_t *path_len)
{
    if (!t || !values) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = t->ctx;
    if (values->ctx != ctx) return HODOR_ERR_INVALID;
    POLY_SYNC(values);
    if (values->n != t->n) { set_err(ctx, "query: the values are not the vector this oracle commits to"); return HODOR_ERR_SIZE; }
    return hodor_iop_query_combined_dev(ctx, (void *)ctx->stream, values->dfr(), t->d(), t->n, t->combiner, natural_index,
                                        values_out, path, path_len);
}

// ------------------------------------------------------------------------------------------------
// FRI on handles
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_fri_commit_h(const hodor_poly *lde_values, size_t lde_factor, size_t out_deg, int combiner,
                                  int through_coefficients, hodor_fri_proto **out)
{
    POLY_ENTRY(lde_values);
    NEED_FORM(lde_values, HODOR_FORM_VALUES);
    if (through_coefficients)
        return hodor_fri_commit_through_coefficients_dev(ctx, lde_values->stream(), lde_values->dfr(), lde_values->n,
                                                         lde_factor, out_deg, combiner, out);
    return hodor_fri_commit_combined_dev(ctx, lde_values->stream(), lde_values->dfr(), lde_values->n, lde_factor, out_deg,
                                         combiner, out);
}

extern "C" int hodor_fri_commit_batch_h(const hodor_poly *const *lde_values, size_t count, size_t lde_factor, size_t out_deg,
                                        int combiner, hodor_fri_proto **outs)
{
    if (!lde_values || !outs || count == 0 || count > 8 || !lde_values[0]) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = lde_values[0]->ctx;
    NEED_DEVICE();
    const hodor_fr *ptrs[8];
    size_t ns[8];
    for (size_t i = 0; i < count; i++) {
        if (!lde_values[i] || lde_values[i]->ctx != ctx) return HODOR_ERR_INVALID;
        NEED_FORM(lde_values[i], HODOR_FORM_VALUES);
        POLY_SYNC(lde_values[i]);
        ptrs[i] = lde_values[i]->dfr();
        ns[i] = lde_values[i]->n;
    }
    return hodor_fri_commit_batch_dev(ctx, ptrs, ns, count, lde_factor, out_deg, combiner, outs);
}

extern "C" size_t hodor_fri_produce_proof_h(hodor_fri_proto *p, const hodor_poly *lde_values,
                                            size_t natural_first_element_index, uint8_t *buf, size_t cap)
{
    if (!p || !lde_values || lde_values->ctx != p->ctx || lde_values->n != p->n) return 0;
    if (lde_values->host_dirty && poly_flush(const_cast<hodor_poly *>(lde_values))) return 0;
    return hodor_fri_produce_proof(p, lde_values->dfr(), natural_first_element_index, buf, cap);
}

extern "C" int hodor_fri_verify_prototype_h(hodor_fri_proto *p, const hodor_poly *lde_values, size_t natural_element_index,
                                            int *valid)
{
    if (!p || !lde_values || lde_values->ctx != p->ctx || lde_values->n != p->n) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    POLY_SYNC(lde_values);
    HIPCHK(hipStreamSynchronize(ctx->stream));   // the walk fetches elements with blocking copies
    return hodor_fri_verify_prototype(p, lde_values->dfr(), natural_element_index, valid);
}

// l0_commitment (step = -1) / intermediate_commitments[step] (src/fri/mod.rs:107-109) as an IOP object: a VIEW of the
// prototype's own tree (no copy; valid while the prototype lives; its root is already on the host)
extern "C" int hodor_fri_commitment_h(hodor_fri_proto *p, int step, hodor_iop **out)
{
    if (!p || !out || step < -1 || step >= (int)p->num_steps) return HODOR_ERR_INVALID;
    hodor_iop *t = new (std::nothrow) hodor_iop();
    if (!t) return HODOR_ERR_INVALID;
    t->ctx = p->ctx;
    t->combiner = p->combiner;
    t->n = step < 0 ? p->n : p->inter_sizes[step];
    t->entries = iop_entries(t->n, p->combiner);
    t->raw = (uint8_t *)(step < 0 ? p->l0_nodes : p->inter_nodes[step]);
    memcpy(t->root, p->roots.data() + 32 * (size_t)(step + 1), 32);
    t->root_valid = true;
    t->ctx->live_handles.fetch_add(1);
    *out = t;
    return HODOR_OK;
}

extern "C" int hodor_fri_intermediate_values_h(hodor_fri_proto *p, size_t step, hodor_poly **out)
{
    if (!p || !out || step >= p->num_steps) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    return hodor_poly_from_dev_h(ctx, HODOR_FORM_VALUES, (const hodor_fr *)p->inter_values[step], p->inter_sizes[step],
                                 (void *)ctx->stream, out);
}
