// exchange.hpp — internal to csrc/: the exchange handle behind `hodor_exchange` (abi_exchange.hip owns its life cycle and
// the three transports; abi_sixstep.hip the producers of the direct transport; abi_dist.hip the schedules on top).
#pragma once
#include <rccl/rccl.h>   // types and prototypes only: every function is resolved with dlsym (abi_exchange.hip)

#include "ctx.hpp"

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
    // ---- direct transport (no communicator, no copy: the producing pass stores into the peers' receive buffers)
    struct Slot {
        uint64_t *d_tab = nullptr;          // device array of n_ranks receive-buffer addresses (as mapped HERE)
        uint64_t h_tab[HODOR_EXCHANGE_MAX_RANKS] = {};   // the same addresses on the host (copy-engine transport)
        bool set = false;
        uint32_t produced = 0, consumed = 0;   // generations this rank has started producing into / consuming from the slot
    };
    uint32_t n_slots = 0;
    Slot *slots = nullptr;
    uint32_t *my_flags = nullptr;           // this rank's flag block: per slot { arrived[n_ranks], released[n_ranks] }
    uint32_t *peer_flags[HODOR_EXCHANGE_MAX_RANKS] = {};   // every rank's flag block as mapped HERE (own one included)
    uint32_t *d_err = nullptr;              // pinned host word (device-visible): set by a flag wait that timed out
    // copy-engine transport: one stream per destination, so that the copies to the seven peers run on seven engines /
    // links at once instead of one after the other on comm_stream (created at the first copy; `gate` is recorded on
    // comm_stream once the generation may be written, peer_done[t] after the copy to rank t)
    hipStream_t peer_stream[HODOR_EXCHANGE_MAX_RANKS] = {};
    hipEvent_t peer_done[HODOR_EXCHANGE_MAX_RANKS] = {};
    hipEvent_t gate = nullptr;
    // abi_dist.hip: a schedule claims the LOWEST slot no open operation holds and gives it back when its release has been
    // enqueued — a function of the sequence of dist calls only, hence the same slot on every rank
    bool slot_busy[16] = {};
    void *own_recv[16] = {};                // receive buffers the library allocated itself (hodor_exchange_direct_alloc_recv)
    size_t own_recv_bytes = 0;
    // ---- the schedule inside the library (abi_dist.hip): grow-only work buffers (send / receive pieces of the RCCL and
    // copy-engine transports, the staging buffers of the LDE by cosets), each with the event that marks its last use
    static constexpr int WORK = 8;
    void *work[WORK] = {};
    size_t work_bytes[WORK] = {};
    hipEvent_t work_free[WORK] = {};        // recorded on the stream that last read / wrote the buffer
    bool work_used[WORK] = {};
    int transport = -1;                     // HODOR_TRANSPORT_*; -1: RCCL when there is a communicator, DIRECT otherwise
    int force_collectives = 0;              // world 1: issue the (one-rank) exchange anyway (a testing aid)
    int ops_in_flight = 0;
    bool pair_busy[2] = {false, false};     // work buffers {0, 1} / {2, 3}: the send / receive pair of a split-phase transform
};

