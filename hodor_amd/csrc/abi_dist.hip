// abi_dist.hip — the multi-GPU SCHEDULES behind the C ABI: what rounds 1-4 sequenced in Python
// (hodor_amd/sixstep.py, hodor_amd/distributed.py) as library entry points, so that a Rust process per GPU calls ONE
// function per distributed transform / commit and owns no schedule of its own:
//
//   hodor_dist_ntt_forward_dev / _inverse_dev     one transform split over the ranks, layout A <-> B, ONE exchange cut
//                                                 into 2^log_chunks overlapped pieces (the distributed form of
//                                                 parallel_fft, /root/reference/src/fft/fft.rs:68-124)
//   hodor_dist_ntt_begin_dev / _end_dev           the same split in two, so that two independent transforms interleave on
//                                                 one stream and each exchange runs behind the other's arithmetic
//   hodor_dist_ntt_natural_dev                    natural blocks in and out (three exchanges)
//   hodor_dist_lde_by_cosets_dev                  Polynomial::lde's own coset schedule (src/polynomials/mod.rs:418-482)
//                                                 with the cosets dealt to the ranks and ONE exchange for the interleave
//   hodor_dist_commit_dev                         subtree per rank, the P subtree roots exchanged (32 bytes per rank),
//                                                 the top log2(P) levels hashed by every rank
//
// All of them run over whichever transport the hodor_exchange handle carries (RCCL all-to-all on the library's
// communication stream — the default; the direct stores; the copy engine), through ONE helper for the plain all-to-alls
// (dist_all_to_all) and the split-phase pair for the transform's own exchange.  Send / receive / staging buffers are
// the handle's own (grow-only), guarded by events so that calls on different streams do not race for them.
// Unmeasured between real devices like everything in §8(e): exercised at world 1 over each transport and between two
// processes that share the one GPU.
#include "exchange.hpp"

extern "C" int hodor_exchange_direct_table(hodor_exchange *x, uint32_t slot, const uint64_t **tab, uint32_t *n_ranks,
                                           uint32_t *rank);

namespace {

int effective_transport(const hodor_exchange *x)
{
    if (x->transport >= 0) return x->transport;
    return x->comm ? HODOR_TRANSPORT_RCCL : HODOR_TRANSPORT_DIRECT;
}

// grow-only work buffer i of the handle, at least `bytes`; `stream` is made to wait for the buffer's last user
int work_acquire(hodor_exchange *x, int i, size_t bytes, hipStream_t stream, void **out)
{
    hodor_ctx *ctx = x->ctx;
    if (x->work_bytes[i] < bytes) {
        if (x->work[i]) {
            HIPCHK(hipDeviceSynchronize());
            BOUNDS_FORGET(x->work[i]);
            HIPCHK(hipFree(x->work[i]));
            x->work[i] = nullptr;
            x->work_bytes[i] = 0;
        }
        HIPCHK(dev_malloc(&x->work[i], bytes));
        BOUNDS_NOTE(x->work[i], bytes);
        x->work_bytes[i] = bytes;
    } else if (x->work_used[i] && x->work_free[i]) {
        HIPCHK(hipStreamWaitEvent(stream, x->work_free[i], 0));
    }
    *out = x->work[i];
    return HODOR_OK;
}

// everything enqueued on `stream` so far is the last use of work buffer i
int work_done(hodor_exchange *x, int i, hipStream_t stream)
{
    hodor_ctx *ctx = x->ctx;
    if (!x->work_free[i]) HIPCHK(hipEventCreateWithFlags(&x->work_free[i], hipEventDisableTiming));
    HIPCHK(hipEventRecord(x->work_free[i], stream));
    x->work_used[i] = true;
    return HODOR_OK;
}

// Slots of the peer-mapped transports.  Round 5 claimed them round robin, which hands a slot that an OPEN split-phase
// transform still holds to the next one as soon as the others have cycled (begin A, begin B, end B, begin C: C gets A's
// slot, and its begin waits for a release only A's end will enqueue later on the same stream).  Now: the lowest slot no
// open operation holds — every rank issues the same sequence of dist calls, so every rank picks the same one — and -1
// when all are held (the caller refuses with HODOR_ERR_INVALID).  Caller holds x->mu.
int claim_slot(hodor_exchange *x)
{
    for (uint32_t s = 0; s < x->n_slots; s++)
        if (!x->slot_busy[s]) { x->slot_busy[s] = true; return (int)s; }
    return -1;
}
void unclaim_slot(hodor_exchange *x, uint32_t s) { if (s < 16) x->slot_busy[s] = false; }

// A schedule failed after it had opened a generation on a peer-mapped slot (a flag wait, a producer or a copy is in the
// queue, the matching signal / release is not): the peers will wait for flags that never come.  The handle is marked
// dead — every later call on it returns HODOR_ERR_DEVICE at once (direct_ready) instead of meeting a half-open
// protocol, and the peers' waits time out into the same state on their side.
int dist_dead(hodor_exchange *x, int rc)
{
    if (x->d_err) *(volatile uint32_t *)x->d_err = 1;
    return rc;
}

void *own_recv_of(const hodor_exchange *x, uint32_t slot) { return (void *)(uintptr_t)x->slots[slot].h_tab[x->rank]; }

// One plain all-to-all of P equal slabs of `send` (n_local elements; slab t -> rank t) over the handle's transport.
// *recv: where the slabs arrive (slab s from rank s) — a work buffer of the handle (RCCL) or the claimed slot's own
// receive buffer (direct transports); valid, with `stream` ordered behind the arrival, when the call returns.  The caller
// enqueues its consumer on `stream` and then calls dist_a2a_release (the peers may overwrite the slot again).
struct A2A { int transport; uint32_t slot; int work_recv; };
int dist_all_to_all(hodor_exchange *x, hipStream_t stream, const hodor_fr *send, size_t n_local, int work_recv, hodor_fr **recv,
                    A2A *h)
{
    hodor_ctx *ctx = x->ctx;
    h->transport = effective_transport(x);
    h->slot = 0;
    h->work_recv = -1;
    if (x->n_ranks == 1 && !x->force_collectives) {   // nothing to exchange
        h->transport = -1;
        *recv = const_cast<hodor_fr *>(send);
        return HODOR_OK;
    }
    int rc;
    if (h->transport == HODOR_TRANSPORT_RCCL) {
        void *r = nullptr;
        if ((rc = work_acquire(x, work_recv, n_local * 32, stream, &r))) return rc;
        h->work_recv = work_recv;
        uint64_t ticket = 0;
        if ((rc = hodor_sixstep_exchange_dev(x, stream, send, (hodor_fr *)r, n_local, 0, 0, &ticket))) return rc;
        if ((rc = hodor_sixstep_exchange_wait_dev(x, stream, ticket))) return rc;
        *recv = (hodor_fr *)r;
        return HODOR_OK;
    }
    // direct transports: the copy engine moves `send` into the peers' receive buffers of a slot
    if (!x->slots) { set_err(ctx, "dist: the exchange handle carries no transport"); h->transport = -1; return HODOR_ERR_INVALID; }
    if (n_local * 32 > x->own_recv_bytes && x->own_recv_bytes) { set_err(ctx, "dist: the slot's receive buffers are too small"); h->transport = -1; return HODOR_ERR_SIZE; }
    int slot;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        slot = claim_slot(x);
    }
    if (slot < 0) { set_err(ctx, "dist: every slot of the handle is held by an open operation"); h->transport = -1; return HODOR_ERR_INVALID; }
    h->slot = (uint32_t)slot;
    if ((rc = hodor_exchange_direct_copy_dev(x, stream, h->slot, send, n_local, 0, 0)) ||
        (rc = hodor_exchange_direct_wait_dev(x, stream, h->slot))) {
        std::lock_guard<std::mutex> lk(x->mu);
        unclaim_slot(x, h->slot);
        h->transport = -1;
        return dist_dead(x, rc);
    }
    *recv = (hodor_fr *)own_recv_of(x, h->slot);
    return HODOR_OK;
}

int dist_a2a_release(hodor_exchange *x, hipStream_t stream, const A2A &h)
{
    if (h.transport < 0) return HODOR_OK;
    if (h.transport == HODOR_TRANSPORT_RCCL) return work_done(x, h.work_recv, stream);
    int rc = hodor_exchange_direct_release_dev(x, stream, h.slot);
    std::lock_guard<std::mutex> lk(x->mu);
    unclaim_slot(x, h.slot);
    return rc ? dist_dead(x, rc) : rc;
}

// after the stream a schedule ran on has been synchronised: did one of its flag waits give up (peer-mapped transports)?
int dist_check_waits(hodor_exchange *x)
{
    if (x->d_err && *(volatile uint32_t *)x->d_err) {
        set_err(x->ctx, "dist: a wait for a peer timed out; the results of this call are undefined");
        return HODOR_ERR_DEVICE;
    }
    return HODOR_OK;
}

}  // namespace

// n = N1 * N2 with N1 <= N2.  From 2^19 to 2^27 points N1 = 2^9: the column transforms are then ONE pass of the transform
// kernel and the rows two — three passes over the data like the single-device plan, where the balanced split costs four
// (profiles/r02/sixstep_rank_shape.txt); outside that range the split is balanced.
extern "C" void hodor_dist_split(uint32_t log_n, uint32_t *log_n1, uint32_t *log_n2)
{
    const uint32_t a = (log_n >= 19 && log_n <= 27) ? 9 : log_n / 2;
    if (log_n1) *log_n1 = a;
    if (log_n2) *log_n2 = log_n - a;
}

extern "C" int hodor_dist_set_transport(hodor_exchange *x, int transport, int force_collectives)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    std::lock_guard<std::mutex> lk(x->mu);
    if (transport == HODOR_TRANSPORT_RCCL && !x->comm) { set_err(ctx, "dist: this handle has no communicator"); return HODOR_ERR_INVALID; }
    if ((transport == HODOR_TRANSPORT_DIRECT || transport == HODOR_TRANSPORT_COPY) && !x->slots) {
        set_err(ctx, "dist: this handle has no direct transport (hodor_exchange_create_direct)");
        return HODOR_ERR_INVALID;
    }
    if (transport < -1 || transport > HODOR_TRANSPORT_COPY) return HODOR_ERR_INVALID;
    x->transport = transport;
    x->force_collectives = force_collectives ? 1 : 0;
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// one transform, split in two
// ------------------------------------------------------------------------------------------------
struct hodor_dist_op {
    hodor_exchange *x;
    hipStream_t stream;
    int inverse, transport;
    uint32_t log_n1, log_n2, log_p, log_chunks, slot, pair;
    hodor_fr omega;
    const hodor_fr *recv;       // what the consumer reads
    int work_send, work_recv;   // work buffers held (-1: none)
    uint64_t ticket;
    bool collective;
    bool holds_slot = false;    // a slot of a peer-mapped transport is claimed (given back by end)
};

extern "C" int hodor_dist_ntt_begin_dev(hodor_exchange *x, void *stream_, const hodor_fr *src, size_t n_local, uint32_t log_n,
                                        const hodor_fr *omega, int inverse, uint32_t log_chunks, hodor_dist_op **out)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!src || !omega || !out) return HODOR_ERR_INVALID;
    const uint32_t log_p = log2u(x->n_ranks);
    if (log_n > 40 || log_n < 2 * log_p || n_local != ((size_t)1 << (log_n - log_p))) {
        set_err(ctx, "dist_ntt: n_local must be 2^log_n / n_ranks");
        return HODOR_ERR_SIZE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    uint32_t log_n1, log_n2;
    hodor_dist_split(log_n, &log_n1, &log_n2);
    const uint32_t K = 1u << log_chunks;
    if (n_local % ((size_t)K * x->n_ranks) != 0) { set_err(ctx, "dist_ntt: too many chunks for this transform"); return HODOR_ERR_SIZE; }
    hodor_dist_op *op = new (std::nothrow) hodor_dist_op();
    if (!op) return HODOR_ERR_INVALID;
    op->x = x;
    op->stream = stream;
    op->inverse = inverse ? 1 : 0;
    op->log_n1 = log_n1;
    op->log_n2 = log_n2;
    op->log_p = log_p;
    op->log_chunks = log_chunks;
    op->omega = *omega;
    op->work_send = op->work_recv = -1;
    op->ticket = 0;
    int rc = HODOR_OK;
    uint32_t pair = 0;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        op->transport = effective_transport(x);
        op->collective = x->n_ranks > 1 || x->force_collectives;
        if (x->pair_busy[0] && x->pair_busy[1]) { set_err(ctx, "dist_ntt: at most two transforms in flight on one handle"); rc = HODOR_ERR_INVALID; }
        else { pair = x->pair_busy[0] ? 1 : 0; x->pair_busy[pair] = true; x->ops_in_flight++; }
        if (!rc && op->transport != HODOR_TRANSPORT_RCCL) {
            if (!x->slots) { set_err(ctx, "dist_ntt: the exchange handle carries no transport"); rc = HODOR_ERR_INVALID; }
            else if (x->ops_in_flight > (int)x->n_slots) {   // a second begin on the only slot would wait for a release
                set_err(ctx, "dist_ntt: a handle with N slots carries at most N transforms in flight");   // its own end enqueues
                rc = HODOR_ERR_INVALID;
            } else if (x->own_recv_bytes && n_local * 32 > x->own_recv_bytes) {   // the peers would store past the end of the slot
                set_err(ctx, "dist_ntt: the slot's receive buffers are too small for this transform");
                rc = HODOR_ERR_SIZE;
            } else {
                const int sl = claim_slot(x);
                if (sl < 0) { set_err(ctx, "dist_ntt: every slot of the handle is held by an open operation"); rc = HODOR_ERR_INVALID; }
                else { op->slot = (uint32_t)sl; op->holds_slot = true; }
            }
            if (rc) { x->ops_in_flight--; x->pair_busy[pair] = false; }
        }
    }
    if (rc) { delete op; return rc; }
    op->pair = pair;
    // the producing half: forward = the column transforms, inverse = the inverse row transforms, chunk by chunk
    auto produce = [&](uint32_t k, hodor_fr *dst) {
        return inverse ? hodor_sixstep_rows_dev(ctx, stream, src, dst, log_n1, log_n2, log_p, x->rank, omega, 1, log_chunks, k)
                       : hodor_sixstep_columns_dev(ctx, stream, src, dst, log_n1, log_n2, log_p, x->rank, omega, 0, log_chunks, k);
    };
    const size_t step = n_local / K;
    if (op->transport == HODOR_TRANSPORT_DIRECT) {
        rc = hodor_exchange_direct_begin_dev(x, stream, op->slot);
        for (uint32_t k = 0; k < K && !rc; k++)
            rc = inverse ? hodor_sixstep_rows_direct_dev(ctx, stream, src, x, op->slot, log_n1, log_n2, omega, log_chunks, k)
                         : hodor_sixstep_columns_direct_dev(ctx, stream, src, x, op->slot, log_n1, log_n2, omega, log_chunks, k);
        if (!rc) rc = hodor_exchange_direct_signal_dev(x, stream, op->slot);
        op->recv = (const hodor_fr *)own_recv_of(x, op->slot);
    } else {
        void *send = nullptr, *recv = nullptr;
        op->work_send = 2 * (int)pair;
        rc = work_acquire(x, op->work_send, n_local * 32, stream, &send);
        if (!rc && op->transport == HODOR_TRANSPORT_RCCL && op->collective) {
            op->work_recv = 2 * (int)pair + 1;
            rc = work_acquire(x, op->work_recv, n_local * 32, stream, &recv);
        }
        for (uint32_t k = 0; k < K && !rc; k++) {
            rc = produce(k, (hodor_fr *)send + k * step);
            if (rc) break;
            if (op->transport == HODOR_TRANSPORT_COPY) rc = hodor_exchange_direct_copy_dev(x, stream, op->slot, (const hodor_fr *)send, n_local, log_chunks, k);
            else if (op->collective) rc = hodor_sixstep_exchange_dev(x, stream, (const hodor_fr *)send, (hodor_fr *)recv, n_local, log_chunks, k, &op->ticket);
        }
        if (op->transport == HODOR_TRANSPORT_COPY) op->recv = (const hodor_fr *)own_recv_of(x, op->slot);
        else op->recv = op->collective ? (const hodor_fr *)recv : (const hodor_fr *)send;
    }
    if (rc) {
        std::lock_guard<std::mutex> lk(x->mu);
        x->ops_in_flight--;
        x->pair_busy[pair] = false;
        if (op->work_send >= 0) (void)work_done(x, op->work_send, stream);
        if (op->work_recv >= 0) (void)work_done(x, op->work_recv, stream);
        if (op->holds_slot) {   // a generation may be open on the slot (begin / chunk 0 went in, signal did not): the handle is dead
            unclaim_slot(x, op->slot);
            (void)dist_dead(x, rc);
        }
        delete op;
        return rc;
    }
    *out = op;
    return HODOR_OK;
}

extern "C" int hodor_dist_ntt_end_dev(hodor_dist_op *op, hodor_fr *dst)
{
    if (!op || !op->x) return HODOR_ERR_INVALID;
    hodor_exchange *x = op->x;
    hodor_ctx *ctx = x->ctx;
    int rc = HODOR_OK;
    hipStream_t stream = op->stream;
    // the exchange has arrived (waited for even when the call is about to fail on a null dst: a slot is released only
    // after its wait, or the peers' next generation would overwrite what has not been read) ...
    if (op->transport == HODOR_TRANSPORT_RCCL) { if (op->collective) rc = hodor_sixstep_exchange_wait_dev(x, stream, op->ticket); }
    else rc = hodor_exchange_direct_wait_dev(x, stream, op->slot);
    const bool waited = rc == HODOR_OK;
    if (!rc && !dst) rc = HODOR_ERR_INVALID;
    // ... the consuming half: forward = the row transforms, inverse = the inverse column transforms
    if (!rc)
        rc = op->inverse ? hodor_sixstep_columns_dev(ctx, stream, op->recv, dst, op->log_n1, op->log_n2, op->log_p, x->rank, &op->omega, 1, op->log_chunks, 0)
                         : hodor_sixstep_rows_dev(ctx, stream, op->recv, dst, op->log_n1, op->log_n2, op->log_p, x->rank, &op->omega, 0, op->log_chunks, 0);
    if (op->transport != HODOR_TRANSPORT_RCCL) {
        int r2 = waited ? hodor_exchange_direct_release_dev(x, stream, op->slot) : HODOR_ERR_DEVICE;   // the peers may overwrite the slot again
        if (r2) (void)dist_dead(x, r2);   // no release went out: the peers' next begin on this slot can only time out
        if (!rc) rc = r2;
    }
    {
        std::lock_guard<std::mutex> lk(x->mu);
        if (op->work_send >= 0) (void)work_done(x, op->work_send, stream);
        if (op->work_recv >= 0) (void)work_done(x, op->work_recv, stream);
        if (op->holds_slot) unclaim_slot(x, op->slot);
        x->ops_in_flight--;
        x->pair_busy[op->pair] = false;
    }
    delete op;
    return rc;
}

extern "C" int hodor_dist_ntt_forward_dev(hodor_exchange *x, void *stream, const hodor_fr *a, hodor_fr *b, size_t n_local,
                                          uint32_t log_n, const hodor_fr *omega, uint32_t log_chunks)
{
    hodor_dist_op *op = nullptr;
    int rc = hodor_dist_ntt_begin_dev(x, stream, a, n_local, log_n, omega, 0, log_chunks, &op);
    return rc ? rc : hodor_dist_ntt_end_dev(op, b);
}
extern "C" int hodor_dist_ntt_inverse_dev(hodor_exchange *x, void *stream, const hodor_fr *b, hodor_fr *a, size_t n_local,
                                          uint32_t log_n, const hodor_fr *omega, uint32_t log_chunks)
{
    hodor_dist_op *op = nullptr;
    int rc = hodor_dist_ntt_begin_dev(x, stream, b, n_local, log_n, omega, 1, log_chunks, &op);
    return rc ? rc : hodor_dist_ntt_end_dev(op, a);
}

// natural block in, natural block out: pack + exchange -> layout A, the transform, pack + exchange + transpose (three
// exchanges; a prover that keeps A / B between its transforms never pays the outer two).  inverse: Polynomial::ifft's
// omega^-1 and n^-1 (src/polynomials/mod.rs:773-798).  src is left untouched; dst may not alias it.
extern "C" int hodor_dist_ntt_natural_dev(hodor_exchange *x, void *stream_, const hodor_fr *src, hodor_fr *dst, size_t n_local,
                                          uint32_t log_n, const hodor_fr *omega, int inverse)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!src || !dst || !omega || src == dst) return HODOR_ERR_INVALID;
    const uint32_t log_p = log2u(x->n_ranks);
    if (log_n > 40 || log_n < 2 * log_p || n_local != ((size_t)1 << (log_n - log_p))) return HODOR_ERR_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    uint32_t log_n1, log_n2;
    hodor_dist_split(log_n, &log_n1, &log_n2);
    hodor_fr w = *omega, scale;
    if (inverse) {
        HFr wi, ni;
        if (!ctx->F.inverse(to_h(omega), &wi) || !ctx->F.inverse(ctx->F.from_u64(1ull << log_n), &ni)) return HODOR_ERR_INVALID;
        from_h(wi, &w);
        from_h(ni, &scale);
    }
    int rc;
    void *t0 = nullptr, *t1 = nullptr;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        if (x->ops_in_flight) { set_err(ctx, "dist_ntt_natural: a split-phase transform is in flight on this handle"); return HODOR_ERR_INVALID; }
        if (effective_transport(x) != HODOR_TRANSPORT_RCCL && x->slots && x->n_slots < 2) {
            // the slot of the first exchange is released only after the transform that reads it, and that transform's own
            // exchange needs a slot: on one slot its begin would wait for that release (tests/test_flag_protocol_model_cpu.py)
            set_err(ctx, "dist_ntt_natural: needs a handle with at least two slots");
            return HODOR_ERR_INVALID;
        }
        if ((rc = work_acquire(x, 4, n_local * 32, stream, &t0)) || (rc = work_acquire(x, 5, n_local * 32, stream, &t1))) return rc;
    }
    // From here on the call holds work buffers 4 and 5 and (between an exchange and its release) a slot: every exit runs
    // through `leave`, which marks the buffers' last use — and a failure between an exchange and its release still
    // enqueues that release (dist_a2a_release), so that the peers are not left waiting for it.
    auto leave = [&](int r) {
        std::lock_guard<std::mutex> lk(x->mu);
        (void)work_done(x, 4, stream);
        (void)work_done(x, 5, stream);
        return r;
    };
    // natural block -> layout A
    const hodor_fr *packed = src;
    if (log_p) {
        if ((rc = hodor_sixstep_pack_dev(ctx, stream, src, (hodor_fr *)t0, log_n1 - log_p, log_n2, log_p))) return leave(rc);
        packed = (const hodor_fr *)t0;
    }
    A2A h;
    hodor_fr *a = nullptr;
    if ((rc = dist_all_to_all(x, stream, packed, n_local, 6, &a, &h))) return leave(rc);
    // A -> B (into t1)
    rc = hodor_dist_ntt_forward_dev(x, stream, a, (hodor_fr *)t1, n_local, log_n, &w, 0);
    int r2 = dist_a2a_release(x, stream, h);
    if (rc || (rc = r2)) return leave(rc);
    if (inverse && (rc = hodor_poly_unary_dev(ctx, stream, (hodor_fr *)t1, n_local, HODOR_UN_SCALE, &scale, 0))) return leave(rc);
    // layout B -> natural block of the output
    const hodor_fr *pb = (const hodor_fr *)t1;
    if (log_p) {
        if ((rc = hodor_sixstep_pack_dev(ctx, stream, (const hodor_fr *)t1, (hodor_fr *)t0, log_n1 - log_p, log_n2, log_p))) return leave(rc);
        pb = (const hodor_fr *)t0;
    }
    hodor_fr *y = nullptr;
    if ((rc = dist_all_to_all(x, stream, pb, n_local, 6, &y, &h))) return leave(rc);
    rc = hodor_transpose_dev(ctx, stream, y, dst, (size_t)1 << log_n1, (size_t)1 << (log_n2 - log_p));   // [k1][k2_local] -> [k2_local][k1]
    r2 = dist_a2a_release(x, stream, h);
    return leave(rc ? rc : r2);
}

// ------------------------------------------------------------------------------------------------
// LDE by cosets, commit by subtrees
// ------------------------------------------------------------------------------------------------
// lde_using_multiple_cosets / coset_lde_using_multiple_cosets (src/polynomials/mod.rs:418-482, :544-609) with the cosets
// dealt in contiguous blocks to the ranks: rank r transforms the f / P cosets i = r f/P + t (independent work items of
// the reference's own schedule, :446-460) and ONE exchange interleaves them, out[idx] = res[idx % f][idx / f]
// (:466-479).  coeffs: all n = 2^log_n coefficients, replicated on every rank.  lde_block: this rank's natural block of
// the n f values (n f / P elements) — or, with `paired`, its PAIRED block for a COSET2 commit: natural values
// [d B/2, (d+1) B/2) followed by N/2 + the same range (B = n f / P), so that both members of every COSET2 leaf of chunk d
// are local.  Needs P | f and P | n (2 P | n when paired).
extern "C" int hodor_dist_lde_by_cosets_dev(hodor_exchange *x, void *stream_, const hodor_fr *coeffs, uint32_t log_n, size_t factor,
                                            int coset, int paired, hodor_fr *lde_block)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!coeffs || !lde_block) return HODOR_ERR_INVALID;
    const size_t n = (size_t)1 << log_n, f = factor, P = x->n_ranks;
    if (!is_pow2(f) || f % P || n % P || (paired && n % (2 * P))) {
        set_err(ctx, "dist_lde: the number of ranks must divide the LDE factor and the polynomial size");
        return HODOR_ERR_SIZE;
    }
    const uint32_t log_big = log_n + log2u(f);
    uint64_t sz;
    uint32_t lg;
    HFr W;
    if (log_big > 40 || !ctx->F.domain(1ull << log_big, &sz, &lg, &W)) return HODOR_ERR_SIZE;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t fp = f / P;
    const uint32_t log_fp = log2u(fp), log_p = log2u(P);
    int rc;
    void *res = nullptr, *send = nullptr, *half = nullptr;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        if (x->ops_in_flight) { set_err(ctx, "dist_lde: a split-phase transform is in flight on this handle"); return HODOR_ERR_INVALID; }
        if ((rc = work_acquire(x, 4, fp * n * 32, stream, &res)) || (rc = work_acquire(x, 5, fp * n * 32, stream, &send))) return rc;
        if (paired && (rc = work_acquire(x, 7, fp * n * 16, stream, &half))) return rc;
    }
    auto leave = [&](int r) {   // every exit marks the last use of the work buffers the call holds
        std::lock_guard<std::mutex> lk(x->mu);
        (void)work_done(x, 4, stream);
        (void)work_done(x, 5, stream);
        if (paired) (void)work_done(x, 7, stream);
        return r;
    };
    // my cosets: res[t][k] = P((g) W^(i) w^k), i = rank fp + t — the coset scale runs inside the first pass of the transform
    for (size_t t = 0; t < fp; t++) {
        const size_t i = (size_t)x->rank * fp + t;
        HFr gen = ctx->F.pow(W, i);
        if (coset) gen = ctx->F.mul(gen, ctx->F.generator);
        hodor_fr g;
        from_h(gen, &g);
        hodor_fr *out = (hodor_fr *)res + t * n;
        rc = (i != 0 || coset) ? hodor_poly_coset_fft_for_generator_dev(ctx, stream, coeffs, out, log_n, &g)
                               : hodor_poly_fft_dev(ctx, stream, coeffs, out, log_n);
        if (rc) return leave(rc);
    }
    // part: my cosets on a k range of 2^log_len values, [fp][2^log_len] -> my k block of that range, [klen][f]
    auto interleave = [&](const hodor_fr *part, uint32_t log_len, hodor_fr *out) -> int {
        const size_t klen = ((size_t)1 << log_len) / P, m = fp << log_len;
        const hodor_fr *s = part;
        int r;
        if (fp > 1 && P > 1) {   // slab d = my cosets on rank d's k range: [fp][P][klen] -> [P][fp][klen]
            if ((r = hodor_sixstep_pack_dev(ctx, stream, part, (hodor_fr *)send, log_fp, log_len, log_p))) return r;
            s = (const hodor_fr *)send;
        }
        A2A h;
        hodor_fr *recv = nullptr;
        if ((r = dist_all_to_all(x, stream, s, m, 6, &recv, &h))) return r;
        // slab s came from rank s and holds cosets i = s fp + t: [f][klen] in coset order -> [klen][f]
        r = hodor_transpose_dev(ctx, stream, recv, out, f, klen);
        int r2 = dist_a2a_release(x, stream, h);
        return r ? r : r2;
    };
    if (!paired) {
        rc = interleave((const hodor_fr *)res, log_n, lde_block);
    } else {
        for (int hh = 0; hh < 2 && !rc; hh++) {   // the two halves of every coset's k range, each dealt to the ranks
            hipError_t e = hipMemcpy2DAsync(half, (n / 2) * 32, (const uint8_t *)res + (size_t)hh * (n / 2) * 32, n * 32,
                                            (n / 2) * 32, fp, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { (void)hipGetLastError(); set_err(ctx, std::string("dist_lde: ") + hipGetErrorString(e)); return leave(HODOR_ERR_DEVICE); }
            rc = interleave((const hodor_fr *)half, log_n - 1, lde_block + (size_t)hh * (n * f / P / 2));
        }
    }
    return leave(rc);
}

// Blake2sIopTree::create (src/iop/blake2s_trivial_iop.rs:131-219) over the ranks: each rank builds the complete subtree
// over its block (`leafs_block`, n_block values: its natural block — or, combiner = COSET2, its PAIRED block), the P
// subtree roots travel as ONE exchange of 32 bytes per rank, and every rank hashes the replicated top log2(P) levels
// itself.  local_nodes: the subtree's heap array (n_block x 32 bytes, COSET2: n_block / 2 x 32), local node w + j of a
// local level of width w = global node w P + rank w + j.  top: 2 P x 32 bytes, top[i] = global node i for 1 <= i < 2 P
// (top[P + r] = rank r's subtree root, top[1] = the root, top[0] zero); root: 32 bytes (may be NULL).  Synchronises `stream`.
extern "C" int hodor_dist_commit_dev(hodor_exchange *x, void *stream_, const hodor_fr *leafs_block, size_t n_block, int combiner,
                                     uint8_t *local_nodes, uint8_t *top, uint8_t *root)
{
    if (!x || !x->ctx) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = x->ctx;
    NEED_DEVICE();
    if (!leafs_block || !local_nodes || !top) return HODOR_ERR_INVALID;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t P = x->n_ranks;
    int rc = hodor_iop_create_combined_dev(ctx, stream, leafs_block, n_block, combiner, local_nodes);
    if (rc) return rc;
    void *mine = nullptr;
    {
        std::lock_guard<std::mutex> lk(x->mu);
        if (x->ops_in_flight) { set_err(ctx, "dist_commit: a split-phase transform is in flight on this handle"); return HODOR_ERR_INVALID; }
        if ((rc = work_acquire(x, 5, P * 32, stream, &mine))) return rc;
    }
    auto leave = [&](int r) {
        std::lock_guard<std::mutex> lk(x->mu);
        (void)work_done(x, 5, stream);
        return r;
    };
    for (size_t t = 0; t < P; t++)   // the same 32 bytes for every rank: an all-gather spelled as the all-to-all the transports have
        if (hipMemcpyAsync((uint8_t *)mine + 32 * t, local_nodes + 32, 32, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
            (void)hipGetLastError();
            set_err(ctx, "dist_commit: copying the subtree root failed");
            return leave(HODOR_ERR_DEVICE);
        }
    A2A h;
    hodor_fr *gathered = nullptr;
    if ((rc = dist_all_to_all(x, stream, (const hodor_fr *)mine, P, 6, &gathered, &h))) return leave(rc);
    memset(top, 0, 2 * P * 32);
    // lock order (advisor, round 5): the pinned staging buffer (HostXfer) is released BEFORE x->mu is taken — set_peers
    // takes them the other way round
    hipError_t e;
    {
        HostXfer xfer(ctx, stream);
        e = xfer.d2h(top + 32 * P, gathered, 32 * P);
        if (e == hipSuccess) e = xfer.finish();
    }
    int r2 = dist_a2a_release(x, stream, h);   // (the copy above has completed: the peers may overwrite the slot)
    {
        std::lock_guard<std::mutex> lk(x->mu);
        (void)work_done(x, 5, stream);
    }
    HIPCHK(e);
    if (r2) return r2;
    // the stream has been synchronised: a flag wait of the 32-byte exchange that gave up (slow or dead peer) left the
    // slot holding whatever it held — the root hashed from it would be silently wrong (advisor, round 5)
    if (h.transport == HODOR_TRANSPORT_DIRECT || h.transport == HODOR_TRANSPORT_COPY)
        if ((rc = dist_check_waits(x))) return rc;
    note_round_trip(ctx);
    for (size_t w = P / 2; w >= 1; w /= 2)
        for (size_t i = 0; i < w; i++)
            if ((rc = hodor_hash_node(ctx, top + 32 * (2 * (w + i)), top + 32 * (2 * (w + i) + 1), top + 32 * (w + i)))) return rc;
    if (root) memcpy(root, top + 32, 32);
    return HODOR_OK;
}

// LDE by cosets (ONE exchange) + commit by subtrees (ONE 32-byte exchange): BASELINE config[2] over the node.
extern "C" int hodor_dist_lde_commit_dev(hodor_exchange *x, void *stream, const hodor_fr *coeffs, uint32_t log_n, size_t factor,
                                         int coset, int combiner, hodor_fr *lde_block, uint8_t *local_nodes, uint8_t *top,
                                         uint8_t *root)
{
    if (!x) return HODOR_ERR_INVALID;
    int rc = hodor_dist_lde_by_cosets_dev(x, stream, coeffs, log_n, factor, coset, combiner == HODOR_COMBINER_COSET2, lde_block);
    if (rc) return rc;
    return hodor_dist_commit_dev(x, stream, lde_block, ((size_t)factor << log_n) / x->n_ranks, combiner, local_nodes, top, root);
}
