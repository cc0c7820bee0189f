// ctx.hpp — internal to csrc/: the context and prototype behind the opaque handles of
// include/hodor_gpu.h, the kernels' host launchers, and the helpers the abi_*.hip files share.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/hodor_gpu.h"
#include "host_blake2s.hpp"
#include "host_field.hpp"
#include "knobs.hpp"
#include "ntt.cuh"

namespace hodor {

// kernels' host launchers (ntt.hip, pointwise.hip, merkle.hip, fri.hip)
hipError_t ntt_launch_pass(hipStream_t, const PassArgs &, const Fr9 *scale, const Fr9Params &);
hipError_t pow_table_launch(hipStream_t, uint4 *out, const Fr &base, const Fr &mult,
                            uint32_t log_stride, uint64_t count, uint32_t fmt, const FrParams &);
hipError_t degree_one_small_launch(hipStream_t s, uint4 *out, uint64_t n, const Fr &u, const Fr &alpha, const Fr &c,
                                   const FrParams &P);
hipError_t degree_one_launch(hipStream_t s, uint4 *out, uint64_t n, const TwoLevel &t, const Fr9 &alpha, const Fr9 &c,
                             const Fr9Params &Q);
hipError_t pow_table_w3_launch(hipStream_t, uint4 *out, const Fr &base, const Fr &mult,
                               uint32_t log_stride, uint64_t count, const W3Consts &, const FrParams &);
hipError_t pow_table_w9_launch(hipStream_t, uint32_t *out, const Fr &base, uint32_t log_stride, uint32_t count,
                               const W9Consts &, const FrParams &);
hipError_t distribute_powers_launch(hipStream_t, uint4 *a, uint64_t n, const TwoLevel &t, const Fr9Params &);
hipError_t distribute_powers_small_launch(hipStream_t, uint4 *a, uint64_t n, const Fr &g, const FrParams &);
hipError_t gen_elements_launch(hipStream_t, uint4 *out, uint64_t first, uint64_t count, uint64_t seed,
                               uint64_t top_mask, const Fr &r2, const FrParams &);
hipError_t dense_divisor_launch(hipStream_t, uint4 *out, uint64_t n, const Fr &w, const Fr &g, const uint4 *inv,
                                uint32_t period, const uint4 *roots, uint32_t n_roots, const FrParams &);
hipError_t scale_launch(hipStream_t, uint4 *a, uint64_t n, const Fr &f, const FrParams &);
hipError_t binary_launch(hipStream_t, uint4 *a, const uint4 *b, uint64_t n, int op, const FrParams &);
hipError_t add_scaled_launch(hipStream_t, uint4 *a, const uint4 *b, uint64_t n, const Fr &f, const FrParams &);
hipError_t unary_launch(hipStream_t, uint4 *a, uint64_t n, int op, const Fr &c, uint64_t e, const FrParams &);
hipError_t store_elems_launch(hipStream_t, uint4 *dst, const Fr *v, uint32_t count);   // count <= 4
hipError_t count_diff_launch(hipStream_t, const uint4 *a, const uint4 *b, uint64_t n, uint32_t *flag);
hipError_t quotient_term_launch(hipStream_t, uint4 *acc, const uint4 *f, const uint4 *dinv, uint64_t n, const Fr &value,
                                const Fr *alpha, bool accumulate, const FrParams &);
hipError_t batchinv_forward_launch(hipStream_t, const uint4 *a, uint64_t n, uint64_t T, uint4 *prefix, uint4 *prod,
                                   uint32_t *zero_flag, const FrParams &);
hipError_t batchinv_backward_launch(hipStream_t, uint4 *a, uint64_t n, uint64_t T, const uint4 *prefix,
                                    const uint4 *prod_inv, const FrParams &);
unsigned evaluate_at_table_blocks(uint32_t log_n);
hipError_t evaluate_at_table_launch(hipStream_t, const uint4 *a, uint64_t n, uint32_t log_n, const TwoLevel &t,
                                    uint4 *partials, uint32_t *ticket, uint4 *out, const Fr9Params &, const FrParams &);
hipError_t evaluate_at_launch(hipStream_t, const uint4 *a, uint64_t n, const Fr &g, uint4 *partials,
                              uint32_t *ticket, uint4 *out, const FrParams &);
hipError_t twiddle_mul_launch(hipStream_t, uint4 *a, uint64_t rows, uint64_t cols, uint64_t row0,
                              const TwoLevel &t, uint32_t log_order, const Fr *scale, const FrParams &);
hipError_t merkle_build_launch(hipStream_t, const uint4 *leafs, uint4 *nodes, uint64_t n, const B2Mid &,
                               uint32_t batch = 1, const FoldArgs *fold = nullptr, const Fr9Params *Q = nullptr,
                               bool comb = false);
hipError_t iop_query_coset2_launch(hipStream_t, const uint4 *values, const uint4 *nodes, uint64_t n, uint64_t index,
                                   uint4 *out, const B2Mid &);
bool merkle_fuses_fold(uint64_t n);
hipError_t iop_query_launch(hipStream_t, const uint4 *leaf_pair, const uint4 *nodes, uint64_t n,
                            uint64_t index, uint4 *out, const B2Mid &);
hipError_t challenge_launch(hipStream_t, const uint4 *nodes, uint4 *out, uint4 *root_out, const Fr &r2,
                            uint32_t shave_bits, const FrParams &);
hipError_t fri_round_table_launch(hipStream_t, const uint4 *nodes, uint4 *chal_out, uint4 *root_out,
                                  const uint4 *hi, uint4 *hi_out, uint64_t count, const Fr9 &c16, const Fr &r2,
                                  uint32_t shave, const Fr9Params &, const FrParams &);
hipError_t fri_tail_launch(hipStream_t, const FriTailArgs &, const Fr9 &c16, const Fr &r2, const B2Mid &,
                           const Fr9Params &, const FrParams &, bool comb = false);
hipError_t fri_fold_launch(hipStream_t, const FoldArgs &, const Fr9Params &);
hipError_t fri_fold_coeffs_launch(hipStream_t, const uint4 *src, uint4 *dst, uint64_t half, const uint4 *chal,
                                  const FrParams &);

static inline Fr to_dev(const HFr &a)
{
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)a.l[i];
        r.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    }
    return r;
}

// split a 256-bit integer (4 x u64) into 9 limbs of 29 bits
static inline void split29(const uint64_t l[4], uint32_t out[9])
{
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, w = bit >> 6, s = bit & 63;
        uint64_t v = l[w] >> s;
        if (s > 35 && w < 3) v |= l[w + 1] << (64 - s);
        out[i] = (uint32_t)(v & 0x1fffffffu);
    }
}

// R-form host element (x * 2^256) -> R'-form 9 x 29-bit multiplier operand (x * 2^261 mod p)
static inline Fr9 to_dev9(const HostField &F, const HFr &a)
{
    HFr t = a;
    for (int i = 0; i < 5; i++) t = F.add(t, t);
    Fr9 r;
    split29(t.l, r.v);
    return r;
}

// R-form host element as it is, in 9 limbs of 29 bits (a data operand of fr9_mul / fr9_add)
static inline Fr9 fr9_from_host(const HFr &a)
{
    Fr9 r;
    split29(a.l, r.v);
    return r;
}

struct PowTable {
    HFr base;
    uint32_t log_n;
    uint32_t lo_bits;
    uint32_t fmt;          // 0: 32-byte R-form entries, 1: 48-byte 9 x 29-bit R'-form entries, 2: 112-byte W3 entries
    HFr hi_mult;           // every `hi` entry is multiplied by this (one, or n^-1 for the last iNTT pass)
    uint4 *lo, *hi;
};

struct RadixTable {
    HFr omega;
    uint32_t log_n, log_r;
    uint4 *rtw;
    uint32_t *rtw9;        // omega_R^(e R/32), e < 16, W9 entries (null for R < 64)
};

// Every device / pinned allocation of the library goes through these two (abi_host.hip): hipMalloc / hipHostMalloc, except
// that HODOR_DEBUG_FAIL_ALLOC=<k> makes the k-th allocation of the process (counted from 1, both kinds together) fail with
// hipErrorOutOfMemory, and "<k>+" every allocation from the k-th on — the fault injection of tests/test_gpu_alloc_faults.py.
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t pinned_malloc(void **p, size_t bytes, unsigned flags);
}  // namespace hodor

using namespace hodor;

struct hodor_ctx {
    int device = -1;
    HostField F;
    FrParams P;
    Fr9Params Q;
    W3Consts K3;               // 2^87, 2^174, 2^261 mod p (plain integers) for the W3 table generator
    W9Consts K9;               // 2^(29 (c + 1)) mod p, c = 0 .. 8, for the W9 table generator
    B2Mid mid;
    hipStream_t stream = nullptr;
    hipStream_t aux_streams[7] = {};   // hodor_fri_commit_batch_*: commits beside the one on `stream` (created on first use)
    std::mutex mu;
    std::vector<PowTable> pow_tables;
    std::vector<RadixTable> radix_tables;
    void *scratch[2] = {nullptr, nullptr};
    size_t scratch_bytes[2] = {0, 0};
    // the scratch pool is handed from stream to stream: a new user first waits for everything the previous
    // user's stream has been given so far (ensure_scratch)
    hipStream_t scratch_owner = nullptr;
    bool scratch_owned = false;
    hipEvent_t scratch_ev = nullptr;
    bool scratch_ev_recorded = false;   // set by ScratchUse at the end of the first call that used the pool
    // slice API staging: IO_LANES independent (copy stream, in/out device buffers) sets, so that
    // concurrent callers (src/arp/per_register/mod.rs:43-49 calls best_fft from several scoped threads)
    // overlap one caller's upload with another's kernels and a third's download; see with_device_copy
    struct IoLane {
        hipStream_t stream = nullptr;
        hipEvent_t uploaded = nullptr, computed = nullptr;
        void *buf[2] = {nullptr, nullptr};   // grow-only (in / out)
        size_t bytes[2] = {0, 0};
        bool busy = false;
    };
    static constexpr int IO_LANES = 4;
    IoLane lanes[IO_LANES];
    std::mutex lane_mu;
    std::condition_variable lane_cv;
    // one upload and one download at a time (round 5): copies in the SAME direction share the link and — from pageable
    // memory — the runtime's page pinning, so two of them side by side finish later than one after the other; copies in
    // OPPOSITE directions overlap.  With these two mutexes N concurrent callers form a clean three-stage pipeline
    // (upload | kernels | download) whose throughput is the slower direction's.
    std::mutex up_mu, down_mu;
    // ...and each direction has its OWN stream (abi.hip: dir_streams_prepare).  The HIP runtime binds a copy engine to a
    // (stream, direction) pair the first time the pair is used — the lowest-numbered engine idle at that moment — and keeps
    // the binding.  Lanes that copy both ways on their own stream therefore tend to end up all on engine 0 (the first
    // upload and the first download of a lane both find it idle), and then an upload and a download "in parallel" share one
    // engine: measured, 17-19 ms each instead of 9.6 (profiles/r05/slice_trace.txt).  One stream that only ever uploads and
    // one that only ever downloads, bound while the other is busy, get two engines.
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    std::mutex dir_mu;           // the first slice call prepares them; a failed preparation is retried by the next call
    bool dir_ready = false;
    std::atomic<int> live_exchanges{0};   // hodor_exchange handles that point at this context (abi_exchange.hip)
    std::atomic<int> live_handles{0};     // hodor_poly / hodor_iop / hodor_fri_proto objects whose memory is this context's pool
    uint32_t max_log_r = 9;    // largest per-pass radix (2^max_log_r points)      } measured best on MI355X
    uint32_t tile_log = 10;    // elements per workgroup tile = 2^tile_log         } (bench/size_sweep.sh)
    uint32_t min_log_c = 2;    // fewest tile columns per pass (2^2 x 32 B = 128-byte runs)
    uint32_t tw_hi_max_log = 17;   // largest `hi` half (log2 entries) for which the second pass gets a hi-only
                                   // twiddle split (one product instead of two, for a table that outgrows L2)
    // device memory pool of the handle API (abi_poly.hip).  Everything a handle does is enqueued on ctx->stream, so a
    // block one handle gives back may be handed to the next at once (stream order makes the reuse safe) and a chain of
    // Polynomial operations with temporaries never meets hipMalloc / hipFree (which synchronise the device).
    // seq: when the block became idle (eviction is oldest-first); ev: recorded on `last`, the stream of the block's last
    // user, when it was released (null: the block was idle already) — a new user on another stream waits for it
    struct PoolBlock { void *p; uint64_t seq; hipEvent_t ev; hipStream_t last; };
    std::multimap<size_t, PoolBlock> pool_free;
    std::vector<PoolBlock> pool_zombies;        // evicted from the cache, not yet handed back to HIP (pool_collect)
    size_t pool_zombie_bytes = 0;
    std::vector<hipEvent_t> pool_events;        // idle events, reused
    uint64_t pool_seq = 0;
    size_t pool_cached = 0, pool_live = 0, pool_peak_live = 0;
    size_t pool_cache_cap = (size_t)64 << 30;   // idle bytes kept before blocks go back to HIP (HODOR_POOL_CACHE_GIB)
    std::mutex pool_mu;
    // device -> host results handed out so far (roots, evaluations, query answers, prototypes, as_ref() copies): every one
    // of them stalls the queue, so a device-resident prover counts them (hodor_ctx_host_round_trips) — and the bytes that
    // crossed PCIe in either direction on the library's behalf (hodor_ctx_host_traffic)
    std::atomic<uint64_t> host_round_trips{0};
    std::atomic<uint64_t> h2d_bytes{0}, d2h_bytes{0};
    // host images of whole polynomials (as_ref / as_mut of the handle API, abi_poly.hip): pinned allocations from 1 MiB
    // up — a copy of 128 MiB from pageable memory runs at a fraction of the link rate and has the runtime pin and unpin
    // the caller's pages around it — recycled by exact size (hipHostMalloc of that size costs tens of milliseconds)
    std::multimap<size_t, void *> host_free;
    size_t host_cached = 0;
    static constexpr size_t HOST_CACHE_CAP = (size_t)2 << 30;
    std::mutex host_mu;
    // One pinned host buffer for every SMALL device <-> host transfer of the library (HostXfer below): roots, challenges,
    // flags, query answers, prototypes' result blocks.  Why (round 5, the root cause of the suite's intermittent SIGABRT,
    // DESIGN.md §8): an asynchronous copy to or from PAGEABLE host memory makes the runtime pin the pages it touches and
    // map them into the GPU's address space at the same virtual address; the library used to hand it stack variables and
    // short-lived std::vectors, i.e. pages of the process heap that the NEXT small buffer of anybody (a torch CPU tensor,
    // a numpy array) shares.  When the runtime dropped the stale pin of such a page while another copy into the same
    // page was in flight, the GPU lost the mapping under it: "Memory access fault by GPU ... on address <a heap page>",
    // abort() on the ROCr event thread.  Memory the library pinned itself, once, shares a page with nobody.
    void *pinned = nullptr;
    static constexpr size_t PINNED_BYTES = (size_t)1 << 20;
    std::mutex pinned_mu;
    // ...and since round 6 the same holds for LARGE transfers: with glibc's dynamic mmap threshold a multi-megabyte vector
    // lives in the brk heap like any small buffer, its first and last page shared with its neighbours — the abort came back
    // (same test, same 6 KB torch copy, a heap page) once the suite uploaded many 1-4 MiB numpy arrays through the direct
    // path.  Pageable caller memory of any size now crosses in chunks through these rings of the library's own pinned
    // memory (staged_h2d / staged_d2h, abi.hip); only memory the runtime already knows as pinned (hipHostMalloc,
    // hodor_host_register / hipHostRegister) is handed to a copy directly.
    struct StageRing {
        static constexpr int K = 4;
        static constexpr size_t CHUNK = (size_t)4 << 20;
        void *buf[K] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev[K] = {nullptr, nullptr, nullptr, nullptr};
        bool recorded[K] = {false, false, false, false};
        std::mutex mu;
    };
    StageRing stage_up, stage_down;
    std::string err;           // written through set_err() only (entry points run concurrently)
    mutable std::mutex err_mu;
};

static inline void set_err(hodor_ctx *ctx, const std::string &msg)
{
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    ctx->err = msg;
}
#ifdef HODOR_BOUNDS
namespace hodor { int bounds_poll(hodor_ctx *ctx); }   // abi_bounds.hip: 1 = new violations (the message is in ctx->err)
#endif

// abi.hip: is `p` host memory the runtime has pinned (hipHostMalloc / hipHostRegister)?  Only such memory is ever handed
// to a copy; everything else goes through the context's staging rings, in chunks, copied in / out with memcpy.
// staged_h2d returns when the caller's bytes have left the caller's memory (the last chunks may still be on their way from
// the ring to the device, ordered on `stream`); staged_d2h returns when the caller's memory holds the data.
bool host_is_pinned(const void *p);
hipError_t staged_h2d(hodor_ctx *ctx, hipStream_t stream, void *dev, const void *host, size_t n);
hipError_t staged_d2h(hodor_ctx *ctx, hipStream_t stream, void *host, const void *dev, size_t n);

// Small transfers between device and host through the context's own pinned buffer (see hodor_ctx::pinned).  Holds the
// buffer's mutex for its lifetime; d2h() results are in the caller's memory after finish() (which synchronises `stream`);
// h2d() copies the caller's bytes into the buffer at once, so stack variables may go out of scope — the buffer itself is
// not reused before finish().  Transfers that do not fit (> 1 MiB: whole vectors) go through the staging rings
// (staged_h2d / staged_d2h) unless the caller's memory is pinned already, in which case it is copied directly and must stay
// alive until finish().
class HostXfer {
  public:
    HostXfer(hodor_ctx *c, hipStream_t s) : ctx_(c), stream_(s), lk_(c->pinned_mu) {}
    ~HostXfer() { (void)finish(); }
    hipError_t d2h(void *host, const void *dev, size_t n)
    {
        if (n == 0) return hipSuccess;
        hipError_t e = room(n);
        if (e != hipSuccess) return e;
        ctx_->d2h_bytes.fetch_add(n, std::memory_order_relaxed);
        if (n > hodor_ctx::PINNED_BYTES) {
            pending_ = true;
            return host_is_pinned(host) ? hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, stream_) : staged_d2h(ctx_, stream_, host, dev, n);
        }
        items_.push_back(Item{host, used_, n});
        e = hipMemcpyAsync((uint8_t *)ctx_->pinned + used_, dev, n, hipMemcpyDeviceToHost, stream_);
        used_ += (n + 63) & ~(size_t)63;
        pending_ = true;
        return e;
    }
    hipError_t h2d(void *dev, const void *host, size_t n)
    {
        if (n == 0) return hipSuccess;
        hipError_t e = room(n);
        if (e != hipSuccess) return e;
        ctx_->h2d_bytes.fetch_add(n, std::memory_order_relaxed);
        if (n > hodor_ctx::PINNED_BYTES) {
            pending_ = true;
            return host_is_pinned(host) ? hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, stream_) : staged_h2d(ctx_, stream_, dev, host, n);
        }
        memcpy((uint8_t *)ctx_->pinned + used_, host, n);
        e = hipMemcpyAsync(dev, (uint8_t *)ctx_->pinned + used_, n, hipMemcpyHostToDevice, stream_);
        used_ += (n + 63) & ~(size_t)63;
        pending_ = true;
        return e;
    }
    hipError_t finish()
    {
        if (!pending_) return hipSuccess;
        pending_ = false;
        hipError_t e = hipStreamSynchronize(stream_);
#ifdef HODOR_BOUNDS
        if (e == hipSuccess && hodor::bounds_poll(ctx_)) e = hipErrorAssert;   // a result is about to reach the host: was every access in range?
#endif
        if (e == hipSuccess)
            for (auto &it : items_) memcpy(it.host, (uint8_t *)ctx_->pinned + it.off, it.n);
        items_.clear();
        used_ = 0;
        return e;
    }

  private:
    struct Item { void *host; size_t off, n; };
    hipError_t room(size_t n)
    {
        if (!ctx_->pinned) {
            hipError_t e = pinned_malloc(&ctx_->pinned, hodor_ctx::PINNED_BYTES, hipHostMallocDefault);
            if (e != hipSuccess) { ctx_->pinned = nullptr; return e; }
        }
        if (n <= hodor_ctx::PINNED_BYTES && used_ + n > hodor_ctx::PINNED_BYTES) return finish();   // drain, start over at 0
        return hipSuccess;
    }
    hodor_ctx *ctx_;
    hipStream_t stream_;
    std::unique_lock<std::mutex> lk_;
    std::vector<Item> items_;
    size_t used_ = 0;
    bool pending_ = false;
};

struct hodor_fri_proto {
    hodor_ctx *ctx;
    size_t n, num_steps, lde_factor, out_deg, initial_degree_plus_one;
    int combiner = 0;                         // HODOR_COMBINER_*: the format of every tree of this prototype
    void *slab = nullptr;                     // one device allocation holding everything below
    size_t slab_bytes = 0;
    void *l0_nodes = nullptr;                 // device, n*32
    std::vector<void *> inter_values;         // device
    std::vector<void *> inter_nodes;          // device
    std::vector<size_t> inter_sizes;
    std::vector<uint8_t> roots;               // host: (num_steps+1)*32
    std::vector<hodor_fr> challenges;         // host: num_steps
    std::vector<hodor_fr> final_coeffs;       // host
    uint8_t final_root[32];
};

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            (void)hipGetLastError();   /* reported through the ABI: do not leave it for the next HIP user */ \
            if (e__ != hipErrorAssert) set_err(ctx, std::string(#expr) + ": " + hipGetErrorString(e__));   /* hipErrorAssert: the bounds build's poll has written the message */ \
            return HODOR_ERR_DEVICE;                                                  \
        }                                                                             \
    } while (0)

#define NEED_DEVICE()                                                                 \
    do {                                                                              \
        if (!ctx) return HODOR_ERR_INVALID;                                           \
        if (ctx->device < 0) { set_err(ctx, "context has no HIP device"); return HODOR_ERR_DEVICE; } \
        HIPCHK(hipSetDevice(ctx->device));                                            \
    } while (0)

static inline HFr to_h(const hodor_fr *a)
{
    HFr r;
    memcpy(r.l, a->l, 32);
    return r;
}
static inline void from_h(const HFr &a, hodor_fr *out) { memcpy(out->l, a.l, 32); }
static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static inline uint32_t log2u(size_t n)
{
    uint32_t r = 0;
    while (n > 1) { n >>= 1; r++; }
    return r;
}

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) { BOUNDS_FORGET(p); (void)hipFree(p); } }
};
}  // namespace

static inline hipStream_t pick_stream(hodor_ctx *ctx, void *stream)
{
    (void)ctx;
    return (hipStream_t)stream;   // NULL selects the HIP default (null) stream, as for any HIP API
}

// abi_poly.hip: the handle API's device pool (pool_drain: hodor_ctx_destroy / hodor_ctx_trim; it synchronises the device)
// `consumer`: the stream the block's new user enqueues on; `last_user`: the stream its old user enqueued on (default for
// both: ctx->stream, the handles' stream).  pool_collect: hand evicted blocks back to HIP (waits for the device).
int pool_alloc(hodor_ctx *ctx, size_t bytes, void **out, size_t *got);
int pool_alloc(hodor_ctx *ctx, size_t bytes, void **out, size_t *got, hipStream_t consumer);
void pool_release(hodor_ctx *ctx, void *p, size_t bytes);
void pool_release(hodor_ctx *ctx, void *p, size_t bytes, hipStream_t last_user);
void pool_drain(hodor_ctx *ctx);
void pool_collect(hodor_ctx *ctx);
void pool_destroy_events(hodor_ctx *ctx);
void host_images_drain(hodor_ctx *ctx);   // cached pinned host images of the handle API back to HIP
static inline void note_round_trip(hodor_ctx *ctx) { ctx->host_round_trips.fetch_add(1, std::memory_order_relaxed); }

// abi_selftest.hip: the start-up self-test of hodor_ctx_create and the message of its failure (per thread)
int ctx_self_test(hodor_ctx *ctx);
const char *ctx_create_error();

// defined in abi.hip (caller holds ctx->mu)
int trim_table_cache(hodor_ctx *ctx);
int get_pow_table(hodor_ctx *ctx, const HFr &base, uint32_t log_n, TwoLevel *out, uint32_t fmt,
                  uint32_t lo_bits = 0xffffffffu, const HFr *hi_mult_p = nullptr);
int ensure_scratch(hodor_ctx *ctx, int which, size_t bytes, hipStream_t user);
// Armed right after a successful ensure_scratch (caller holds ctx->mu): when the call leaves — normally or on an
// error path — everything it enqueued on `stream` that touches the pool is behind ctx->scratch_ev.
struct ScratchUse {
    hodor_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    void arm(hodor_ctx *c, hipStream_t s) { ctx = c; stream = s; }
    ~ScratchUse()
    {
        if (!ctx) return;
        if (!ctx->scratch_ev && hipEventCreateWithFlags(&ctx->scratch_ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ctx->scratch_ev = nullptr;
            ctx->scratch_ev_recorded = false;
            return;
        }
        if (hipEventRecord(ctx->scratch_ev, stream) == hipSuccess) {
            ctx->scratch_ev_recorded = true;
        } else {   // could not mark the end of this call: fall back to a full wait so that nobody races it
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
        }
    }
};
struct NttLayout {
    bool col_mode = false;           // transform along the slow axis of a [2^log_n][2^log_width] array
    uint32_t log_width = 0;          // columns transformed (and the width of the intermediate arrays)
    uint32_t src_log_width = 0, dst_log_width = 0;   // 0: same as log_width; else the caller's wider array ...
    uint64_t src_col_off = 0, dst_col_off = 0;       // ... and the array column of tile column 0 in it
    uint64_t col0 = 0;               // global index of array column 0
    const HFr *tw2d_root = nullptr;  // 2D twiddle root^(index * (col0 + col)) ...
    uint32_t tw2d_log_order = 0;     // ... of order 2^tw2d_log_order, on the first pass's inputs (true) or the
    bool tw2d_on_load = false;       //     last pass's outputs (false)
    SplitAddr src_split = {}, dst_split = {};
    // direct exchange: the last pass writes slab t into peer_tab[t] (see PassArgs)
    const uint64_t *peer_tab = nullptr;
    uint64_t peer_off = 0;
    uint32_t peer_log = 0, peer_self = 0;
#ifdef HODOR_BOUNDS
    uint64_t bx_peer_host[8] = {};      // bounds build: the receive buffers behind peer_tab and their size
    uint64_t bx_peer_bytes = 0;
    uint32_t bx_peers = 0;
#endif
};
int ntt_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n, const HFr &omega,
             uint64_t nnz, const HFr *scale, const HFr *pre, const HFr *post, uint32_t batch = 1,
             const NttLayout *lay = nullptr);
int poly_domain(hodor_ctx *ctx, uint32_t log_n, HFr *omega);
enum PolyOp { OP_FFT, OP_COSET_FFT, OP_IFFT, OP_ICOSET_FFT };
int poly_transform(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n, PolyOp op,
                   const HFr *gen = nullptr);   // gen: the coset generator (OP_COSET_FFT) / its inverse (OP_ICOSET_FFT) instead of the field's
// lde / coset_lde of 2^log_n coefficients into 2^log_n * factor values (caller holds ctx->mu)
int poly_lde_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n, size_t factor,
                  int coset, uint32_t batch = 1);
