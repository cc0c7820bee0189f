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

#include "exchange.hpp"

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
    if (x->slots) {
        (void)hipDeviceSynchronize();
        for (uint32_t i = 0; i < x->n_slots; i++)
            if (x->slots[i].d_tab) (void)hipFree(x->slots[i].d_tab);
        delete[] x->slots;
    }
    if (x->my_flags) (void)hipFree(x->my_flags);
    if (x->d_err) (void)hipHostFree(x->d_err);
    (void)hipDeviceSynchronize();           // the work buffers and the library's own receive buffers may still be read
    for (int i = 0; i < hodor_exchange::WORK; i++) {
        if (x->work[i]) { BOUNDS_FORGET(x->work[i]); (void)hipFree(x->work[i]); }
        if (x->work_free[i]) (void)hipEventDestroy(x->work_free[i]);
    }
    for (auto &r : x->own_recv)
        if (r) { BOUNDS_FORGET(r); (void)hipFree(r); }
    if (x->comm_stream) (void)hipStreamSynchronize(x->comm_stream);
    for (uint32_t t = 0; t < HODOR_EXCHANGE_MAX_RANKS; t++) {
        if (x->peer_done[t]) (void)hipEventDestroy(x->peer_done[t]);
        if (x->peer_stream[t]) (void)hipStreamDestroy(x->peer_stream[t]);
    }
    if (x->gate) (void)hipEventDestroy(x->gate);
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


// =====================================================================================================================
// Direct transport: the exchange without an exchange.  Every rank maps every rank's receive buffer (hipIpc* between
// processes — hodor_ipc_export / _import below — or plain pointers when the ranks share a process) and the LAST pass
// of the producing transform stores each output slab straight into the buffer of the rank it is for
// (k_ntt_pass<1>, PassArgs::peer_tab: the kernel's streaming stores go out over xGMI while its other tiles compute).
// No communicator, no copy kernel competing with the VALU-bound transform for CUs, no chunking (the overlap of wire
// and arithmetic happens inside the one launch), no staging send buffer.  What remains of the collective is ordering,
// done with two generation counters per (slot, peer) in fine-grained device memory:
//     arrived[s]   written by rank s into MY block after its producer kernels: its slab of generation g is in my buffer
//     released[t]  written by rank t into MY block after its consumer kernels: it has finished reading what I sent
//   producer:  begin  (wait released[t] >= g - 1 for all t: the slot may be overwritten)
//              hodor_sixstep_columns_direct_dev / _rows_direct_dev ...
//              signal (arrived[me] := g in every peer's block)
//   consumer:  wait   (arrived[s] >= g for all s) ; hodor_sixstep_rows_dev / _columns_dev on the local buffer ;
//              release (released[me] := g in every peer's block)
// Flag writes are one tiny kernel (system-scope stores behind a system fence, after the producer in stream order); flag
// waits are a one-wave kernel that polls with system-scope loads, sleeps between polls and gives up after ~10 s (the
// handle then reports HODOR_ERR_DEVICE at the next call instead of hanging the queue).  Unmeasured between real
// devices: the pool's boxes have one GPU; exercised at world 1, with ranks played in one process, and
// with processes sharing the one GPU over IPC handles.
// =====================================================================================================================
namespace hodor {

struct FlagTargets {
    uint32_t *p[HODOR_EXCHANGE_MAX_RANKS];
};

__global__ void k_flags_write(FlagTargets T, uint32_t count, uint32_t value)
{
    __threadfence_system();
    if (threadIdx.x < count) __hip_atomic_store(T.p[threadIdx.x], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one wave: lane t polls flags[t] until it has reached `value` (wrap-safe), ~10 s at most
__global__ void k_flags_wait(const uint32_t *flags, uint32_t count, uint32_t value, uint32_t *err)
{
    const uint32_t t = threadIdx.x;
    if (t >= count) return;
    const uint64_t t0 = wall_clock64();           // 100 MHz
    for (;;) {
        const uint32_t v = __hip_atomic_load(flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - value) >= 0) break;
        if (wall_clock64() - t0 > 1000000000ull) { atomicExch(err, 1u); break; }
        __builtin_amdgcn_s_sleep(32);
    }
    __threadfence_system();
}

}  // namespace hodor

extern "C" int hodor_ipc_export(hodor_ctx *ctx, void *dev_ptr, uint8_t handle[HODOR_IPC_HANDLE_BYTES])
{
    NEED_DEVICE();
    if (!dev_ptr || !handle) return HODOR_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) <= HODOR_IPC_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, dev_ptr));
    memset(handle, 0, HODOR_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return HODOR_OK;
}

extern "C" int hodor_ipc_import(hodor_ctx *ctx, const uint8_t handle[HODOR_IPC_HANDLE_BYTES], void **dev_ptr)
{
    NEED_DEVICE();
    if (!dev_ptr || !handle) return HODOR_ERR_INVALID;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    HIPCHK(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return HODOR_OK;
}

extern "C" int hodor_ipc_close(hodor_ctx *ctx, void *dev_ptr)
{
    NEED_DEVICE();
    if (!dev_ptr) return HODOR_ERR_INVALID;
    HIPCHK(hipIpcCloseMemHandle(dev_ptr));
    return HODOR_OK;
}

extern "C" int hodor_exchange_create_direct(hodor_ctx *ctx, uint32_t n_ranks, uint32_t rank, uint32_t n_slots,
                                            hodor_exchange **out)
{
    NEED_DEVICE();
    if (!out) return HODOR_ERR_INVALID;
    if (n_ranks == 0 || (n_ranks & (n_ranks - 1)) || n_ranks > HODOR_EXCHANGE_MAX_RANKS || rank >= n_ranks ||
        n_slots == 0 || n_slots > 16) {
        set_err(ctx, "exchange (direct): n_ranks a power of two <= 8, rank < n_ranks, 1 <= n_slots <= 16");
        return HODOR_ERR_SIZE;
    }
    hodor_exchange *x = new (std::nothrow) hodor_exchange();
    if (!x) return HODOR_ERR_INVALID;
    x->ctx = ctx;
    x->device = ctx->device;
    x->n_ranks = n_ranks;
    x->rank = rank;
    x->n_slots = n_slots;
    x->slots = new (std::nothrow) hodor_exchange::Slot[n_slots];
    const size_t flag_bytes = (size_t)n_slots * 2 * n_ranks * sizeof(uint32_t);
    hipError_t e = x->slots ? hipSuccess : hipErrorOutOfMemory;
    // fine-grained: the flags are written by other devices while kernels of this one poll them
    if (e == hipSuccess) e = hipExtMallocWithFlags((void **)&x->my_flags, flag_bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(x->my_flags, 0, flag_bytes);
    if (e == hipSuccess) e = pinned_malloc((void **)&x->d_err, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) *x->d_err = 0;
    for (uint32_t i = 0; e == hipSuccess && i < n_slots; i++) e = dev_malloc((void **)&x->slots[i].d_tab, n_ranks * sizeof(uint64_t));
    // copy-engine transport (hodor_exchange_direct_copy_dev): a stream of its own for the peer copies and their flags
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err(ctx, std::string("exchange (direct): ") + hipGetErrorString(e));
        hodor_exchange_destroy(x);
        return HODOR_ERR_DEVICE;
    }
    x->peer_flags[rank] = x->my_flags;
    ctx->live_exchanges.fetch_add(1);
    x->counted = true;
    *out = x;
    return HODOR_OK;
}

// The receive buffers of the direct transports as allocations of the library's own, one per slot, n_local elements
// each.  `coarse` = 0 (the default of every caller in this repository): FINE-GRAINED device memory, like the flags —
// never cached in this device's L2s and written through by the peers, so that neither a stale line of the previous
// generation on the consumer's side nor a dirty line on the producer's can exist ("memory model of the
// direct transport"); `coarse` = 1: plain hipMalloc, for the A/B on a node that shows the fine-grained form to cost
// anything.  Export each pointer with hodor_ipc_export and pass every rank's to hodor_exchange_direct_set_peers.
extern "C" int hodor_exchange_direct_alloc_recv(hodor_exchange *x, size_t n_local, int coarse, void **recv /* n_slots */)
{
    if (!x || !x->ctx || !x->slots || !recv) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (n_local == 0 || x->n_slots > 16) return HODOR_ERR_SIZE;
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->own_recv[0]) { set_err(ctx, "exchange (direct): the receive buffers exist already"); return HODOR_ERR_INVALID; }
    for (uint32_t i = 0; i < x->n_slots; i++) {
        hipError_t e = coarse ? dev_malloc(&x->own_recv[i], n_local * 32)
                              : hipExtMallocWithFlags(&x->own_recv[i], n_local * 32, hipDeviceMallocFinegrained);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (uint32_t k = 0; k < i; k++) { (void)hipFree(x->own_recv[k]); x->own_recv[k] = nullptr; }
            x->own_recv[i] = nullptr;
            set_err(ctx, std::string("exchange (direct): receive buffers: ") + hipGetErrorString(e));
            return HODOR_ERR_DEVICE;
        }
        BOUNDS_NOTE(x->own_recv[i], n_local * 32);
        recv[i] = x->own_recv[i];
    }
    x->own_recv_bytes = n_local * 32;
    return HODOR_OK;
}

extern "C" int hodor_exchange_direct_flags(hodor_exchange *x, void **flags_dev_ptr, size_t *bytes)
{
    if (!x || !x->my_flags || !flags_dev_ptr) return HODOR_ERR_INVALID;
    *flags_dev_ptr = x->my_flags;
    if (bytes) *bytes = (size_t)x->n_slots * 2 * x->n_ranks * sizeof(uint32_t);
    return HODOR_OK;
}

extern "C" int hodor_exchange_direct_set_peers(hodor_exchange *x, uint32_t slot, void *const *recv, void *const *flags)
{
    if (!x || !x->ctx || !x->slots) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!recv || slot >= x->n_slots) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(x->mu);
    uint64_t tab[HODOR_EXCHANGE_MAX_RANKS];
    for (uint32_t t = 0; t < x->n_ranks; t++) {
        if (!recv[t]) { set_err(ctx, "exchange (direct): null receive buffer"); return HODOR_ERR_INVALID; }
        tab[t] = (uint64_t)(uintptr_t)recv[t];
        if (flags) {
            if (!flags[t]) { set_err(ctx, "exchange (direct): null flag block"); return HODOR_ERR_INVALID; }
            if (t != x->rank) x->peer_flags[t] = (uint32_t *)flags[t];
        }
    }
    {
        HostXfer xfer(ctx, nullptr);
        HIPCHK(xfer.h2d(x->slots[slot].d_tab, tab, x->n_ranks * sizeof(uint64_t)));
        HIPCHK(xfer.finish());
    }
    for (uint32_t t = 0; t < x->n_ranks; t++) x->slots[slot].h_tab[t] = tab[t];
    x->slots[slot].set = true;
    return HODOR_OK;
}

// flag word of (slot, kind, index) inside a rank's block: kind 0 = arrived[], 1 = released[]
static inline uint32_t *flag_at(const hodor_exchange *x, uint32_t *block, uint32_t slot, uint32_t kind, uint32_t idx)
{
    return block + ((size_t)slot * 2 + kind) * x->n_ranks + idx;
}

static int direct_ready(hodor_exchange *x, uint32_t slot)
{
    hodor_ctx *ctx = x->ctx;
    if (!x->slots || slot >= x->n_slots || !x->slots[slot].set) {
        set_err(ctx, "exchange (direct): slot has no peers (hodor_exchange_direct_set_peers)");
        return HODOR_ERR_INVALID;
    }
    for (uint32_t t = 0; t < x->n_ranks; t++)
        if (!x->peer_flags[t]) { set_err(ctx, "exchange (direct): a peer's flag block is missing"); return HODOR_ERR_INVALID; }
    if (*(volatile uint32_t *)x->d_err) {   // pinned host memory: no synchronisation with the device
        set_err(ctx, "exchange (direct): this handle is dead — a schedule failed after it had opened a generation on a slot, or a "
                     "wait for a peer timed out; destroy it and create a new one on every rank");
        return HODOR_ERR_DEVICE;
    }
    return HODOR_OK;
}

static int direct_write(hodor_exchange *x, hipStream_t stream, uint32_t slot, uint32_t kind, uint32_t value)
{
    hodor_ctx *ctx = x->ctx;
    hodor::FlagTargets T = {};
    for (uint32_t t = 0; t < x->n_ranks; t++) T.p[t] = flag_at(x, x->peer_flags[t], slot, kind, x->rank);
    hipLaunchKernelGGL(hodor::k_flags_write, dim3(1), dim3(64), 0, stream, T, x->n_ranks, value);
    HIPCHK(hipGetLastError());
    return HODOR_OK;
}

static int direct_wait(hodor_exchange *x, hipStream_t stream, uint32_t slot, uint32_t kind, uint32_t value)
{
    hodor_ctx *ctx = x->ctx;
    hipLaunchKernelGGL(hodor::k_flags_wait, dim3(1), dim3(64), 0, stream, flag_at(x, x->my_flags, slot, kind, 0),
                       x->n_ranks, value, x->d_err);
    HIPCHK(hipGetLastError());
    return HODOR_OK;
}

extern "C" int hodor_exchange_direct_begin_dev(hodor_exchange *x, void *stream, uint32_t slot)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = direct_ready(x, slot);
    if (rc) return rc;
    const uint32_t g = x->slots[slot].produced + 1;
    // every peer must have finished reading what this rank put into the slot last time
    if (g > 1 && (rc = direct_wait(x, (hipStream_t)stream, slot, 1, g - 1))) return rc;
    x->slots[slot].produced = g;   // the generation opens only once its wait is in the queue
    return HODOR_OK;
}

extern "C" int hodor_exchange_direct_signal_dev(hodor_exchange *x, void *stream, uint32_t slot)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = direct_ready(x, slot);
    if (rc) return rc;
    if (x->slots[slot].produced == 0) { set_err(ctx, "exchange (direct): signal without begin"); return HODOR_ERR_INVALID; }
    return direct_write(x, (hipStream_t)stream, slot, 0, x->slots[slot].produced);
}

extern "C" int hodor_exchange_direct_wait_dev(hodor_exchange *x, void *stream, uint32_t slot)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = direct_ready(x, slot);
    if (rc) return rc;
    const uint32_t g = x->slots[slot].consumed + 1;
    if ((rc = direct_wait(x, (hipStream_t)stream, slot, 0, g))) return rc;
    x->slots[slot].consumed = g;
    return HODOR_OK;
}

// A wait that gave up (~10 s without its peers) lets the stream run on: what the consumer transform then reads — or the
// producer overwrites — is UNDEFINED for that generation.  The flag it leaves behind is host-visible; call this after
// synchronising the stream the generation ran on, before using the result: MANDATORY for a caller of the stream-ordered
// hodor_dist_* / hodor_exchange_direct_* calls (nothing in them can look at the flag before the stream has run);
// hodor_dist_commit_dev, which synchronises itself, checks it before it hashes the gathered roots; the Python binding
// offers it as DirectExchange.status() / .synchronize_and_check() and this repository's tests and benchmarks call it
// wherever they read a result of a peer-mapped transport.
// HODOR_ERR_DEVICE: some wait on this handle has timed out; every later call on the handle fails the same way.
extern "C" int hodor_exchange_direct_status(hodor_exchange *x)
{
    if (!x || !x->ctx || !x->d_err) return HODOR_ERR_INVALID;
    if (*(volatile uint32_t *)x->d_err) {
        set_err(x->ctx, "exchange (direct): a wait for a peer timed out; the results of that generation are undefined");
        return HODOR_ERR_DEVICE;
    }
    return HODOR_OK;
}

extern "C" int hodor_exchange_direct_release_dev(hodor_exchange *x, void *stream, uint32_t slot)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = direct_ready(x, slot);
    if (rc) return rc;
    if (x->slots[slot].consumed == 0) { set_err(ctx, "exchange (direct): release without wait"); return HODOR_ERR_INVALID; }
    return direct_write(x, (hipStream_t)stream, slot, 1, x->slots[slot].consumed);
}

// Copy-engine transport: the chunked schedule's send pieces (hodor_sixstep_columns_dev / _rows_dev into a LOCAL send
// buffer, exactly as for hodor_sixstep_exchange_dev) moved into the peers' mapped receive buffers by P device-to-device
// copies per chunk, each on the stream of its destination (one copy engine / link per peer, all P at once; the handle's
// own stream gates them and collects them) — between devices the runtime runs those on the SDMA engines, so the
// exchange takes no CU from the VALU-bound transforms and, unlike the direct stores, is spread over whatever the caller
// enqueues next.  Slab t of the piece lands in rank t's buffer where an all-to-all would put it (piece `chunk`, slab
// `rank`).  Chunk 0 first waits until every peer has released the slot; the last chunk is followed by the `arrived`
// flags.  The consumer is the direct transport's: hodor_exchange_direct_wait_dev, the plain rows / columns call on the
// slot's own receive buffer, hodor_exchange_direct_release_dev.  `send` must stay valid until that wait has been
// enqueued (this rank's own `arrived` flag is written after all of its copies).  Unmeasured between real devices.
extern "C" int hodor_exchange_direct_copy_dev(hodor_exchange *x, void *stream, uint32_t slot, const hodor_fr *send,
                                              size_t n_local, uint32_t log_chunks, uint32_t chunk)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!send || !x->comm_stream) return HODOR_ERR_INVALID;
    if (log_chunks > 20 || chunk >= (1u << log_chunks) || n_local == 0 ||
        n_local % ((size_t)x->n_ranks << log_chunks) != 0) {
        set_err(ctx, "exchange (copy): n_local must be a multiple of n_ranks * chunks, chunk < chunks");
        return HODOR_ERR_SIZE;
    }
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = direct_ready(x, slot);
    if (rc) return rc;
    const size_t piece = n_local >> log_chunks, slab = piece / x->n_ranks;
    HIPCHK(hipEventRecord(x->ready, (hipStream_t)stream));
    HIPCHK(hipStreamWaitEvent(x->comm_stream, x->ready, 0));
    if (chunk == 0) {
        const uint32_t g = x->slots[slot].produced + 1;
        if (g > 1 && (rc = direct_wait(x, x->comm_stream, slot, 1, g - 1))) return rc;
        x->slots[slot].produced = g;
    } else if (x->slots[slot].produced == 0) {
        set_err(ctx, "exchange (copy): chunk 0 opens a generation");
        return HODOR_ERR_INVALID;
    }
    const uint8_t *src = (const uint8_t *)(send + (size_t)chunk * piece);
    if (!x->gate) HIPCHK(hipEventCreateWithFlags(&x->gate, hipEventDisableTiming));
    HIPCHK(hipEventRecord(x->gate, x->comm_stream));                // the chunk exists and the receivers have let go of the slot
    for (uint32_t i = 0; i < x->n_ranks; i++) {
        const uint32_t t = (x->rank + 1 + i) % x->n_ranks;          // start with the neighbour: the ranks' copies fan out over the links
        if (!x->peer_stream[t]) {
            HIPCHK(hipStreamCreateWithFlags(&x->peer_stream[t], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&x->peer_done[t], hipEventDisableTiming));
        }
        uint8_t *dst = (uint8_t *)(uintptr_t)x->slots[slot].h_tab[t] + ((size_t)chunk * piece + (size_t)x->rank * slab) * 32;
        HIPCHK(hipStreamWaitEvent(x->peer_stream[t], x->gate, 0));
        HIPCHK(hipMemcpyAsync(dst, src + (size_t)t * slab * 32, slab * 32, hipMemcpyDeviceToDevice, x->peer_stream[t]));
        HIPCHK(hipEventRecord(x->peer_done[t], x->peer_stream[t]));
    }
    for (uint32_t t = 0; t < x->n_ranks; t++)                       // comm_stream: after every copy of this chunk
        HIPCHK(hipStreamWaitEvent(x->comm_stream, x->peer_done[t], 0));
    if (chunk + 1 == (1u << log_chunks)) return direct_write(x, x->comm_stream, slot, 0, x->slots[slot].produced);
    return HODOR_OK;
}

#ifdef HODOR_BOUNDS
extern "C" uint64_t hodor_exchange_direct_host_table(hodor_exchange *x, uint32_t slot, uint64_t out[8])
{
    if (!x || !x->slots || slot >= x->n_slots) return 0;
    for (uint32_t t = 0; t < 8; t++) out[t] = t < x->n_ranks ? x->slots[slot].h_tab[t] : 0;
    return x->own_recv_bytes;
}
#endif

// the device table of a slot and this rank's index, for abi_sixstep.hip
extern "C" int hodor_exchange_direct_table(hodor_exchange *x, uint32_t slot, const uint64_t **tab, uint32_t *n_ranks,
                                           uint32_t *rank)
{
    if (!x || !x->slots || slot >= x->n_slots || !x->slots[slot].set || !tab) return HODOR_ERR_INVALID;
    *tab = x->slots[slot].d_tab;
    if (n_ranks) *n_ranks = x->n_ranks;
    if (rank) *rank = x->rank;
    return HODOR_OK;
}
