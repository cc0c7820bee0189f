// abi_exchange.hip — the exchange step of the 4-step / 6-step transform behind the C ABI.
//
// hodor_sixstep_columns_dev / hodor_sixstep_rows_dev (abi_sixstep.hip) do a rank's local arithmetic; between
// them sits ONE all-to-all of P equal contiguous slabs — the distributed-memory form of the un-shuffle at
// /root/reference/src/fft/fft.rs:111-123.  A Python caller can run it through torch.distributed
// (hodor_amd/sixstep.py); a caller without a collective library of its own — the Rust prover — uses the entry
// points below and needs nothing but this repository's .so: RCCL is bound at run time (dlopen of librccl.so.1,
// the copy already in the process if there is one), there is no link-time dependency and a box without RCCL
// still loads the library (the exchange entry points then return HODOR_ERR_DEVICE).
//
// Ordering.  The exchange of a chunk runs on a communication stream the hodor_exchange owns:
//     hodor_sixstep_exchange_dev(x, stream, ...)   records an event on `stream` (everything enqueued there so far —
//                                                  the kernels that wrote the chunk — must finish first), makes the
//                                                  communication stream wait for it and enqueues RCCL's
//                                                  all-to-all of the P slabs (ncclAllToAll; grouped ncclSend / ncclRecv where
//                                                  the library lacks it) there: `stream` itself does NOT
//                                                  wait, so the arithmetic of the next chunk overlaps the wire time;
//     hodor_sixstep_exchange_wait_dev(x, stream, t) makes `stream` wait for the exchange with ticket t and all earlier ones
//                                                  (0: every exchange issued so far) before the consuming rows / columns
//                                                  call reads the receive buffer.
// The caller keeps send and receive buffers alive and untouched from the exchange call to the wait.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: every function is resolved with dlsym

#include "ctx.hpp"

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;   // RCCL's own entry point (absent from NCCL proper): optional
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;   // optional: the device an adopted communicator lives on
    decltype(&ncclCommCount) CommCount = nullptr;         // optional
    decltype(&ncclCommUserRank) CommUserRank = nullptr;   // optional
    std::string why;
    bool ok = false;
};

const Rccl &rccl()
{
    static const Rccl R = [] {
        Rccl r;
        // the copy that is already mapped (PyTorch-ROCm brings its own) before any other: one RCCL per process
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            const char *e = dlerror();
            r.why = std::string("librccl.so.1 not found: ") + (e ? e : "?");
            return r;
        }
#define BIND(field, sym)                                                          \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym));            \
        if (!r.field) { r.why = "librccl lacks " #sym; return r; }
        BIND(GetUniqueId, ncclGetUniqueId)
        BIND(CommInitRank, ncclCommInitRank)
        BIND(CommDestroy, ncclCommDestroy)
        BIND(GetErrorString, ncclGetErrorString)
        BIND(GroupStart, ncclGroupStart)
        BIND(GroupEnd, ncclGroupEnd)
        BIND(Send, ncclSend)
        BIND(Recv, ncclRecv)
#undef BIND
        r.AllToAll = reinterpret_cast<decltype(r.AllToAll)>(dlsym(r.lib, "ncclAllToAll"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.lib, "ncclCommCuDevice"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
        r.ok = true;
        return r;
    }();
    return R;
}

}  // namespace

struct hodor_exchange {
    hodor_ctx *ctx = nullptr;
    int device = -1;                // copied at creation: destroy must not depend on the context still being alive
    ncclComm_t comm = nullptr;
    bool owns_comm = false;
    uint32_t n_ranks = 1, rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ready = nullptr;     // recorded on the caller's stream: the chunk has been produced
    static constexpr uint64_t RING = 64;
    hipEvent_t done[RING] = {};     // done[t % RING]: recorded on comm_stream after exchange number t (tickets start at 1)
    uint64_t issued = 0;            // number of exchanges enqueued so far = the latest ticket
    bool counted = false;           // registered in ctx->live_exchanges (hodor_ctx_destroy refuses while any is alive)
    std::mutex mu;
};

static_assert(HODOR_EXCHANGE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the unique id travels as opaque bytes");

#define NCCLCHK(expr)                                                                         \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            set_err(ctx, std::string(#expr) + ": " + rccl().GetErrorString(r__));             \
            return HODOR_ERR_DEVICE;                                                          \
        }                                                                                     \
    } while (0)

extern "C" int hodor_exchange_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int hodor_exchange_unique_id(uint8_t id[HODOR_EXCHANGE_ID_BYTES])
{
    if (!id) return HODOR_ERR_INVALID;
    if (!rccl().ok) return HODOR_ERR_DEVICE;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return HODOR_ERR_DEVICE;
    memcpy(id, u.internal, HODOR_EXCHANGE_ID_BYTES);
    return HODOR_OK;
}

static int exchange_finish(hodor_ctx *ctx, hodor_exchange *x, hodor_exchange **out)
{
    hipError_t e;
    // highest priority: a chunk's exchange must not queue behind the workgroups of the next chunk's arithmetic, which
    // is what it is supposed to overlap (and what the consuming call finally waits for)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if ((e = hipStreamCreateWithPriority(&x->comm_stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&x->ready, hipEventDisableTiming)) != hipSuccess) {
        set_err(ctx, std::string("exchange: ") + hipGetErrorString(e));
        hodor_exchange_destroy(x);
        return HODOR_ERR_DEVICE;
    }
    for (uint64_t i = 0; i < hodor_exchange::RING; i++)
        if ((e = hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming)) != hipSuccess) {
            set_err(ctx, std::string("exchange: ") + hipGetErrorString(e));
            hodor_exchange_destroy(x);
            return HODOR_ERR_DEVICE;
        }
    ctx->live_exchanges.fetch_add(1);
    x->counted = true;
    *out = x;
    return HODOR_OK;
}

extern "C" int hodor_exchange_create(hodor_ctx *ctx, const uint8_t id[HODOR_EXCHANGE_ID_BYTES], uint32_t n_ranks,
                                     uint32_t rank, hodor_exchange **out)
{
    NEED_DEVICE();
    if (!id || !out) return HODOR_ERR_INVALID;
    if (n_ranks == 0 || (n_ranks & (n_ranks - 1)) || rank >= n_ranks) {
        set_err(ctx, "exchange: the number of ranks must be a power of two and rank < n_ranks");
        return HODOR_ERR_SIZE;
    }
    if (!rccl().ok) { set_err(ctx, rccl().why); return HODOR_ERR_DEVICE; }
    hodor_exchange *x = new (std::nothrow) hodor_exchange();
    if (!x) return HODOR_ERR_INVALID;
    x->ctx = ctx;
    x->device = ctx->device;
    x->n_ranks = n_ranks;
    x->rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, HODOR_EXCHANGE_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&x->comm, (int)n_ranks, u, (int)rank);   // collective over the ranks
    if (r != ncclSuccess) {
        set_err(ctx, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
        delete x;
        return HODOR_ERR_DEVICE;
    }
    x->owns_comm = true;
    return exchange_finish(ctx, x, out);
}

extern "C" int hodor_exchange_adopt(hodor_ctx *ctx, void *nccl_comm, uint32_t n_ranks, uint32_t rank,
                                    hodor_exchange **out)
{
    NEED_DEVICE();
    if (!nccl_comm || !out) return HODOR_ERR_INVALID;
    if (n_ranks == 0 || (n_ranks & (n_ranks - 1)) || rank >= n_ranks) return HODOR_ERR_SIZE;
    if (!rccl().ok) { set_err(ctx, rccl().why); return HODOR_ERR_DEVICE; }
    // the adopted communicator must be the one the caller describes: on the context's device, n_ranks wide, this rank
    {
        const Rccl &R = rccl();
        int dev = -1, cnt = -1, me = -1;
        if (R.CommCuDevice && R.CommCuDevice((ncclComm_t)nccl_comm, &dev) == ncclSuccess && dev != ctx->device) {
            set_err(ctx, "exchange_adopt: the communicator lives on device " + std::to_string(dev) +
                             ", the context on device " + std::to_string(ctx->device));
            return HODOR_ERR_INVALID;
        }
        if (R.CommCount && R.CommCount((ncclComm_t)nccl_comm, &cnt) == ncclSuccess && cnt != (int)n_ranks) {
            set_err(ctx, "exchange_adopt: the communicator has " + std::to_string(cnt) + " ranks, not " + std::to_string(n_ranks));
            return HODOR_ERR_SIZE;
        }
        if (R.CommUserRank && R.CommUserRank((ncclComm_t)nccl_comm, &me) == ncclSuccess && me != (int)rank) {
            set_err(ctx, "exchange_adopt: this is rank " + std::to_string(me) + " of the communicator, not " + std::to_string(rank));
            return HODOR_ERR_SIZE;
        }
    }
    hodor_exchange *x = new (std::nothrow) hodor_exchange();
    if (!x) return HODOR_ERR_INVALID;
    x->ctx = ctx;
    x->device = ctx->device;
    x->comm = (ncclComm_t)nccl_comm;
    x->n_ranks = n_ranks;
    x->rank = rank;
    return exchange_finish(ctx, x, out);
}

extern "C" void hodor_exchange_destroy(hodor_exchange *x)
{
    if (!x) return;
    if (x->device >= 0) (void)hipSetDevice(x->device);
    if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
    if (x->owns_comm && x->comm && rccl().ok) (void)rccl().CommDestroy(x->comm);
    if (x->ready) (void)hipEventDestroy(x->ready);
    for (uint64_t i = 0; i < hodor_exchange::RING; i++)
        if (x->done[i]) (void)hipEventDestroy(x->done[i]);
    if (x->comm_stream) (void)hipStreamDestroy(x->comm_stream);
    if (x->counted && x->ctx) x->ctx->live_exchanges.fetch_sub(1);
    delete x;
}

// Chunk `chunk` of 2^log_chunks: elements [chunk * n_local / K, (chunk + 1) * n_local / K) of both buffers, seen as
// n_ranks equal slabs; slab t of the send piece goes to rank t, slab s of the receive piece comes from rank s —
// exactly the buffers hodor_sixstep_columns_dev / _rows_dev write and gather from.
extern "C" int hodor_sixstep_exchange_dev(hodor_exchange *x, void *stream, const hodor_fr *send, hodor_fr *recv,
                                          size_t n_local, uint32_t log_chunks, uint32_t chunk, uint64_t *ticket)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!send || !recv || send == recv) return HODOR_ERR_INVALID;
    if (log_chunks > 20 || chunk >= (1u << log_chunks) || n_local == 0 ||
        n_local % ((size_t)x->n_ranks << log_chunks) != 0) {
        set_err(ctx, "exchange: n_local must be a multiple of n_ranks * chunks, chunk < chunks");
        return HODOR_ERR_SIZE;
    }
    const size_t piece = n_local >> log_chunks, slab = piece / x->n_ranks;
    const uint8_t *s = (const uint8_t *)(send + (size_t)chunk * piece);
    uint8_t *r = (uint8_t *)(recv + (size_t)chunk * piece);
    std::lock_guard<std::mutex> lk(x->mu);
    HIPCHK(hipEventRecord(x->ready, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent(x->comm_stream, x->ready, 0));
    const Rccl &R = rccl();
    if (R.AllToAll) {
        // RCCL's all-to-all of equal pieces: the same P sends and P receives, issued by the library that knows the
        // topology (and a plain device copy on a one-rank communicator)
        NCCLCHK(R.AllToAll(s, r, slab * 32, ncclInt8, x->comm, x->comm_stream));
    } else {
        NCCLCHK(R.GroupStart());
        for (uint32_t peer = 0; peer < x->n_ranks; peer++) {
            ncclResult_t a = R.Send(s + (size_t)peer * slab * 32, slab * 32, ncclInt8, (int)peer, x->comm, x->comm_stream);
            ncclResult_t b = R.Recv(r + (size_t)peer * slab * 32, slab * 32, ncclInt8, (int)peer, x->comm, x->comm_stream);
            if (a != ncclSuccess || b != ncclSuccess) {
                (void)R.GroupEnd();
                set_err(ctx, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(a != ncclSuccess ? a : b));
                return HODOR_ERR_DEVICE;
            }
        }
        NCCLCHK(R.GroupEnd());
    }
    x->issued += 1;
    HIPCHK(hipEventRecord(x->done[x->issued % hodor_exchange::RING], x->comm_stream));
    if (ticket) *ticket = x->issued;
    return HODOR_OK;
}

// `ticket`: what hodor_sixstep_exchange_dev handed out for the LAST chunk the consumer needs (the communication stream
// runs in order, so every earlier exchange is then complete too); 0 = everything issued so far.  Two transforms in
// flight on one handle (the forward of one step interleaved with the inverse of another) therefore wait for their own
// exchanges only.  A ticket more than 64 exchanges old has been overtaken: the wait then covers everything.
extern "C" int hodor_sixstep_exchange_wait_dev(hodor_exchange *x, void *stream, uint64_t ticket)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->issued == 0) return HODOR_OK;
    if (ticket > x->issued) { set_err(ctx, "exchange: ticket from the future"); return HODOR_ERR_INVALID; }
    if (ticket == 0 || x->issued - ticket >= hodor_exchange::RING) ticket = x->issued;
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, x->done[ticket % hodor_exchange::RING], 0));
    return HODOR_OK;
}
