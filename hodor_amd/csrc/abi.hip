// abi.hip — implementation of include/hodor_gpu.h: context, twiddle cache, NTT planning, and the
// C entry points that stand where the reference's L2/L3 Rust functions stand.  No CPU fallback: a
// context without a device refuses every compute call with HODOR_ERR_DEVICE.
#include "ctx.hpp"
#include <chrono>
#include <cstdio>

namespace hodor {

// constants of fr9.cuh for this modulus
static void make_params9(const HostField &F, Fr9Params *Q)
{
    split29(F.p, Q->p);
    uint32_t p0 = Q->p[0], inv = 1;               // p0 odd: Newton iteration mod 2^32, then mask
    for (int i = 0; i < 5; i++) inv *= 2 - p0 * inv;
    Q->pinv = (0u - inv) & 0x1fffffffu;
    // 4p spread: limb i (< 8) borrows 2^29 from limb i+1 so that every low limb is >= 2^29 - 1
    uint64_t p4[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        p4[i] |= F.p[i] << 2;
        p4[i + 1] |= F.p[i] >> 62;
    }
    uint32_t q[9];
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, w = bit >> 6, s = bit & 63;
        uint64_t v = p4[w] >> s;
        if (s > 35) v |= p4[w + 1] << (64 - s);
        q[i] = (uint32_t)(i < 8 ? (v & 0x1fffffffu) : v);
    }
    for (int i = 0; i < 9; i++) {
        uint32_t c = q[i];
        if (i < 8) c += 1u << 29;
        if (i > 0) c -= 1;
        Q->c4p[i] = c;
    }
    // 5p = 4p + p and 11p = 5p + 5p + p, spread the same way
    auto spread_sum = [&](const uint32_t *a, const uint32_t *b, uint32_t *plain, uint32_t *out) {
        uint32_t carry = 0;
        for (int i = 0; i < 9; i++) {
            uint32_t v = a[i] + b[i] + carry;
            carry = i < 8 ? v >> 29 : 0;
            if (i < 8) v &= 0x1fffffffu;
            plain[i] = v;
            if (i < 8) v += 1u << 29;
            if (i > 0) v -= 1;
            out[i] = v;
        }
    };
    uint32_t p5[9], p10[9], p11[9], scratch9[9];
    spread_sum(q, Q->p, p5, Q->c5p);
    spread_sum(p5, p5, p10, scratch9);
    spread_sum(p10, Q->p, p11, Q->c11p);
    // mu = floor(2^(red_bit + 16) / p), red_bit = NUM_BITS - 5, by binary long division (~12-bit quotient)
    const int red_bit = (int)F.num_bits - 5;
    Q->red_shift = (uint32_t)(red_bit - 232);
    const int top = red_bit + 16;
    uint64_t rem[5] = {0, 0, 0, 0, 0};
    uint32_t mu = 0;
    for (int bit = top; bit >= 0; bit--) {
        for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);   // rem <<= 1
        rem[0] <<= 1;
        if (bit == top) rem[0] |= 1;
        bool ge = rem[4] != 0;
        if (!ge) {
            ge = true;
            for (int i = 3; i >= 0; i--) {
                if (rem[i] > F.p[i]) break;
                if (rem[i] < F.p[i]) { ge = false; break; }
            }
        }
        mu <<= 1;
        if (ge) {
            uint64_t bw = 0;
            for (int i = 0; i < 4; i++) {
                u128_t d = (u128_t)rem[i] - F.p[i] - bw;
                rem[i] = (uint64_t)d;
                bw = (uint64_t)(d >> 64) & 1;
            }
            rem[4] -= bw;
            mu |= 1;
        }
    }
    Q->mu = mu;
}

}  // namespace hodor

// ------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------
static int free_tables(hodor_ctx *ctx)
{
    HIPCHK(hipDeviceSynchronize());
    for (auto &t : ctx->pow_tables) { BOUNDS_FORGET(t.lo); BOUNDS_FORGET(t.hi); (void)hipFree(t.lo); (void)hipFree(t.hi); }
    for (auto &t : ctx->radix_tables) { BOUNDS_FORGET(t.rtw); BOUNDS_FORGET(t.rtw9); (void)hipFree(t.rtw); if (t.rtw9) (void)hipFree(t.rtw9); }
    ctx->pow_tables.clear();
    ctx->radix_tables.clear();
    return HODOR_OK;
}

// Cache eviction happens only here, at the start of an operation and before it has looked up any
// table: evicting later would free tables the same operation already holds pointers to.
int trim_table_cache(hodor_ctx *ctx)
{
    const size_t cap = (size_t)knobs().table_cache;   // HODOR_TABLE_CACHE (40): a debugging aid — small values evict often
    if (ctx->pow_tables.size() > cap || ctx->radix_tables.size() > 2 * cap) return free_tables(ctx);
    return HODOR_OK;
}

// base^e = lo[e & mask] * hi[e >> lo_bits] for e < 2^log_n
// fmt 2 = 112-byte W3 entries for k_ntt_pass (fr9w3.cuh), fmt 1 = 48-byte 9 x 29-bit R'-form entries
// (fr9_mul operand: fold, distribute_powers, evaluate_at), fmt 0 = 32-byte R-form entries (twiddle_mul)
int get_pow_table(hodor_ctx *ctx, const HFr &base, uint32_t log_n, TwoLevel *out, uint32_t fmt, uint32_t lo_bits,
                  const HFr *hi_mult_p)
{
    if (lo_bits > log_n) lo_bits = (log_n + 1) / 2;
    const HFr hi_mult = hi_mult_p ? *hi_mult_p : ctx->F.one;
    for (auto &t : ctx->pow_tables)
        if (t.log_n == log_n && t.lo_bits == lo_bits && t.fmt == fmt && t.base == base && t.hi_mult == hi_mult) {
            *out = TwoLevel{t.lo, t.hi, t.lo_bits};
#ifdef HODOR_BOUNDS
            out->lo_bytes = ((size_t)1 << t.lo_bits) * (fmt == 2 ? 112 : (fmt ? 48 : 32));
            out->hi_bytes = ((size_t)1 << (log_n - t.lo_bits)) * (fmt == 2 ? 112 : (fmt ? 48 : 32));
#endif
            return HODOR_OK;
        }
    PowTable t;
    t.base = base;
    t.log_n = log_n;
    t.lo_bits = lo_bits;
    t.fmt = fmt;
    t.hi_mult = hi_mult;
    const size_t esz = fmt == 2 ? 112 : (fmt ? 48 : 32);
    uint64_t lo_cnt = 1ull << t.lo_bits, hi_cnt = 1ull << (log_n - t.lo_bits);
    t.lo = t.hi = nullptr;
    auto fail = [&](hipError_t e, const char *what) {   // a half-built table is not cached and not leaked
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        BOUNDS_FORGET(t.lo);
        BOUNDS_FORGET(t.hi);
        if (t.lo) (void)hipFree(t.lo);
        if (t.hi) (void)hipFree(t.hi);
        set_err(ctx, std::string(what) + ": " + hipGetErrorString(e));
        return HODOR_ERR_DEVICE;
    };
    hipError_t e;
    if ((e = dev_malloc((void **)&t.lo, lo_cnt * esz)) != hipSuccess) return fail(e, "power table (hipMalloc)");
    BOUNDS_NOTE(t.lo, lo_cnt * esz);
    if ((e = dev_malloc((void **)&t.hi, hi_cnt * esz)) != hipSuccess) return fail(e, "power table (hipMalloc)");
    BOUNDS_NOTE(t.hi, hi_cnt * esz);
    Fr b = to_dev(base), one = to_dev(ctx->F.one);
    if (fmt == 2) {
        if ((e = pow_table_w3_launch(ctx->stream, t.lo, b, one, 0, lo_cnt, ctx->K3, ctx->P)) != hipSuccess ||
            (e = pow_table_w3_launch(ctx->stream, t.hi, b, to_dev(hi_mult), t.lo_bits, hi_cnt, ctx->K3, ctx->P)) != hipSuccess)
            return fail(e, "power table (k_pow_table_w3)");
    } else {
        if ((e = pow_table_launch(ctx->stream, t.lo, b, one, 0, lo_cnt, fmt, ctx->P)) != hipSuccess ||
            (e = pow_table_launch(ctx->stream, t.hi, b, to_dev(hi_mult), t.lo_bits, hi_cnt, fmt, ctx->P)) != hipSuccess)
            return fail(e, "power table (k_pow_table)");
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "power table");   // tables are shared across streams afterwards
    ctx->pow_tables.push_back(t);
    *out = TwoLevel{t.lo, t.hi, t.lo_bits};
#ifdef HODOR_BOUNDS
    out->lo_bytes = lo_cnt * esz;
    out->hi_bytes = hi_cnt * esz;
#endif
    return HODOR_OK;
}

// omega_R^e = omega^(e << (log_n - log_r)), e < R/2
static int get_radix_table(hodor_ctx *ctx, const HFr &omega, uint32_t log_n, uint32_t log_r,
                           const uint4 **out, const uint32_t **out9)
{
    for (auto &t : ctx->radix_tables)
        if (t.log_n == log_n && t.log_r == log_r && t.omega == omega) {
            *out = t.rtw;
            *out9 = t.rtw9;
            return HODOR_OK;
        }
    RadixTable t;
    t.omega = omega;
    t.log_n = log_n;
    t.log_r = log_r;
    uint64_t cnt = log_r ? (1ull << (log_r - 1)) : 1;
    t.rtw = nullptr;
    t.rtw9 = nullptr;
    // an error below must not leak what has been allocated so far: the entry is not in the cache yet, so nothing
    // else (trim_table_cache, hodor_ctx_destroy) would ever free it
    auto fail = [&](hipError_t e, const char *what) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        BOUNDS_FORGET(t.rtw);
        BOUNDS_FORGET(t.rtw9);
        if (t.rtw) (void)hipFree(t.rtw);
        if (t.rtw9) (void)hipFree(t.rtw9);
        set_err(ctx, std::string(what) + ": " + hipGetErrorString(e));
        return HODOR_ERR_DEVICE;
    };
    hipError_t e;
    if ((e = dev_malloc((void **)&t.rtw, cnt * 112)) != hipSuccess) return fail(e, "radix table (hipMalloc)");
    BOUNDS_NOTE(t.rtw, cnt * 112);
    if ((e = pow_table_w3_launch(ctx->stream, t.rtw, to_dev(omega), to_dev(ctx->F.one), log_n - log_r, cnt, ctx->K3,
                                 ctx->P)) != hipSuccess)
        return fail(e, "radix table (k_pow_table_w3)");
    if (log_r >= 6) {   // the 16 powers omega_R^(e R/32) = omega^(e << (log_n - 5)) the wave-uniform steps use
        if ((e = dev_malloc((void **)&t.rtw9, 16 * W9_WORDS * sizeof(uint32_t))) != hipSuccess ||
            (BOUNDS_NOTE(t.rtw9, 16 * W9_WORDS * sizeof(uint32_t)), false) ||
            (e = pow_table_w9_launch(ctx->stream, t.rtw9, to_dev(omega), log_n - 5, 16, ctx->K9, ctx->P)) != hipSuccess) {
            // the W9 table is an optimisation: without it the pass takes the W3 path for every step
            (void)hipGetLastError();
            BOUNDS_FORGET(t.rtw9);
            if (t.rtw9) (void)hipFree(t.rtw9);
            t.rtw9 = nullptr;
        }
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "radix table (synchronize)");
    ctx->radix_tables.push_back(t);
    *out = t.rtw;
    *out9 = t.rtw9;
    return HODOR_OK;
}

// staging buffers of the slice API: kept across calls so a caller looping over registers
// (src/arp/per_register/mod.rs:43-49) pays hipMalloc once, not per call
static hodor_ctx::IoLane *lane_acquire(hodor_ctx *ctx)
{
    std::unique_lock<std::mutex> lk(ctx->lane_mu);
    for (;;) {
        for (auto &L : ctx->lanes)
            if (!L.busy) { L.busy = true; return &L; }
        ctx->lane_cv.wait(lk);
    }
}
static void lane_release(hodor_ctx *ctx, hodor_ctx::IoLane *L)
{
    { std::lock_guard<std::mutex> lk(ctx->lane_mu); L->busy = false; }
    ctx->lane_cv.notify_one();
}
static hipError_t lane_prepare(hodor_ctx::IoLane *L, int which, size_t bytes)
{
    hipError_t e;
    if (!L->stream) {
        if ((e = hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&L->uploaded, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&L->computed, hipEventDisableTiming)) != hipSuccess) return e;
    }
    if (L->bytes[which] >= bytes) return hipSuccess;
    if (L->buf[which]) {   // the lane is idle (its last call synchronised its stream) — safe to free
        BOUNDS_FORGET(L->buf[which]);
        if ((e = hipFree(L->buf[which])) != hipSuccess) return e;
        L->buf[which] = nullptr;
        L->bytes[which] = 0;
    }
    if ((e = dev_malloc(&L->buf[which], bytes)) != hipSuccess) return e;
    BOUNDS_NOTE(L->buf[which], bytes);
    L->bytes[which] = bytes;
    return hipSuccess;
}

// ---- pageable caller memory never meets the runtime (ctx.hpp: hodor_ctx::StageRing)
bool host_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime: pageable
    return at.type == hipMemoryTypeHost;
}

static hipError_t stage_slot(hodor_ctx::StageRing &R, int slot)   // caller holds R.mu: the slot exists and nobody is using it
{
    hipError_t e;
    if (!R.buf[slot]) {
        if ((e = pinned_malloc(&R.buf[slot], hodor_ctx::StageRing::CHUNK, hipHostMallocDefault)) != hipSuccess) { R.buf[slot] = nullptr; return e; }
    }
    if (!R.ev[slot] && (e = hipEventCreateWithFlags(&R.ev[slot], hipEventDisableTiming)) != hipSuccess) { R.ev[slot] = nullptr; return e; }
    if (R.recorded[slot]) {
        if ((e = hipEventSynchronize(R.ev[slot])) != hipSuccess) return e;
        R.recorded[slot] = false;
    }
    return hipSuccess;
}

hipError_t staged_h2d(hodor_ctx *ctx, hipStream_t stream, void *dev, const void *host, size_t n)
{
    hodor_ctx::StageRing &R = ctx->stage_up;
    constexpr size_t C = hodor_ctx::StageRing::CHUNK;
    std::lock_guard<std::mutex> lk(R.mu);
    hipError_t e;
    size_t i = 0;
    for (size_t off = 0; off < n; off += C, i++) {
        const int slot = (int)(i % hodor_ctx::StageRing::K);
        const size_t len = n - off < C ? n - off : C;
        if ((e = stage_slot(R, slot)) != hipSuccess) return e;
        memcpy(R.buf[slot], (const uint8_t *)host + off, len);
        if ((e = hipMemcpyAsync((uint8_t *)dev + off, R.buf[slot], len, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        if ((e = hipEventRecord(R.ev[slot], stream)) != hipSuccess) return e;
        R.recorded[slot] = true;
    }
    return hipSuccess;
}

hipError_t staged_d2h(hodor_ctx *ctx, hipStream_t stream, void *host, const void *dev, size_t n)
{
    hodor_ctx::StageRing &R = ctx->stage_down;
    constexpr size_t C = hodor_ctx::StageRing::CHUNK;
    constexpr size_t K = hodor_ctx::StageRing::K;
    std::lock_guard<std::mutex> lk(R.mu);
    hipError_t e;
    const size_t chunks = (n + C - 1) / C;
    auto collect = [&](size_t j) -> hipError_t {     // chunk j has arrived in its slot: hand it to the caller
        const int slot = (int)(j % K);
        hipError_t e2 = hipEventSynchronize(R.ev[slot]);
        R.recorded[slot] = false;
        if (e2 != hipSuccess) return e2;
        const size_t off = j * C, len = n - off < C ? n - off : C;
        memcpy((uint8_t *)host + off, R.buf[slot], len);
        return hipSuccess;
    };
    for (size_t i = 0; i < chunks; i++) {
        const int slot = (int)(i % K);
        if (i >= K && (e = collect(i - K)) != hipSuccess) return e;
        if ((e = stage_slot(R, slot)) != hipSuccess) return e;
        const size_t off = i * C, len = n - off < C ? n - off : C;
        if ((e = hipMemcpyAsync(R.buf[slot], (const uint8_t *)dev + off, len, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipEventRecord(R.ev[slot], stream)) != hipSuccess) return e;
        R.recorded[slot] = true;
    }
    for (size_t j = chunks > K ? chunks - K : 0; j < chunks; j++)
        if ((e = collect(j)) != hipSuccess) return e;
    return hipSuccess;
}

static void stage_destroy(hodor_ctx::StageRing &R)
{
    for (int i = 0; i < hodor_ctx::StageRing::K; i++) {
        if (R.ev[i]) (void)hipEventDestroy(R.ev[i]);
        if (R.buf[i]) (void)hipHostFree(R.buf[i]);
    }
}

// The two direction streams of the slice API (ctx.hpp).  Bound to their copy engines here, once: a train of small uploads is
// put in flight on the upload stream and the download stream's first copy is submitted while they run, so that the
// runtime finds the upload's engine busy and gives the download another one.
static hipError_t dir_streams_prepare(hodor_ctx *ctx)
{
    std::lock_guard<std::mutex> once(ctx->dir_mu);
    if (ctx->dir_ready) return hipSuccess;
    // Nothing is latched on failure (round 6: a std::call_once here turned ONE refused allocation into a slice API that
    // answered "out of memory" for the rest of the context's life): what was created is destroyed and the next call tries again.
    hipError_t e;
    hipStream_t up = nullptr, down = nullptr;
    void *scratch = nullptr;
    const size_t half = hodor_ctx::PINNED_BYTES / 2;
    auto undo = [&](hipError_t err) {
        (void)hipGetLastError();
        if (scratch) (void)hipFree(scratch);
        if (up) (void)hipStreamDestroy(up);
        if (down) (void)hipStreamDestroy(down);
        return err;
    };
    if ((e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking)) != hipSuccess) return undo(e);
    if ((e = hipStreamCreateWithFlags(&down, hipStreamNonBlocking)) != hipSuccess) return undo(e);
    if ((e = dev_malloc(&scratch, hodor_ctx::PINNED_BYTES)) != hipSuccess) return undo(e);
    {
        std::lock_guard<std::mutex> lk(ctx->pinned_mu);
        if (!ctx->pinned && (e = pinned_malloc(&ctx->pinned, hodor_ctx::PINNED_BYTES, hipHostMallocDefault)) != hipSuccess) {
            ctx->pinned = nullptr;
            return undo(e);
        }
        for (int i = 0; i < 256 && e == hipSuccess; i++)
            e = hipMemcpyAsync(scratch, ctx->pinned, half, hipMemcpyHostToDevice, up);
        if (e == hipSuccess)
            e = hipMemcpyAsync((char *)ctx->pinned + half, (char *)scratch + half, half, hipMemcpyDeviceToHost, down);
        hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(down);
        if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    }
    if (e != hipSuccess) return undo(e);
    (void)hipFree(scratch);
    ctx->up_stream = up;
    ctx->down_stream = down;
    ctx->dir_ready = true;
    return hipSuccess;
}

// Caller holds ctx->mu.  `user`: the stream whose work is about to use the pool.  Every call that uses the pool
// ends by recording ctx->scratch_ev on its own stream (ScratchUse below) — while that stream is certainly alive —
// and the next user, if it runs on a different stream, waits for that event: calls on different streams are
// ordered on the pool instead of racing for it, and no stream handle is ever touched after its call returned.
int ensure_scratch(hodor_ctx *ctx, int which, size_t bytes, hipStream_t user)
{
    if (ctx->scratch_owned && ctx->scratch_owner != user && ctx->scratch_ev_recorded)
        HIPCHK(hipStreamWaitEvent(user, ctx->scratch_ev, 0));
    ctx->scratch_owner = user;
    ctx->scratch_owned = true;
    if (ctx->scratch_bytes[which] >= bytes) return HODOR_OK;
    if (ctx->scratch[which]) {
        HIPCHK(hipDeviceSynchronize());
        BOUNDS_FORGET(ctx->scratch[which]);
        HIPCHK(hipFree(ctx->scratch[which]));
        ctx->scratch[which] = nullptr;
        ctx->scratch_bytes[which] = 0;
    }
    HIPCHK(dev_malloc(&ctx->scratch[which], bytes));
    BOUNDS_NOTE(ctx->scratch[which], bytes);
    ctx->scratch_bytes[which] = bytes;
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// NTT planning + execution
// ------------------------------------------------------------------------------------------------
static void plan_radices(const hodor_ctx *ctx, uint32_t log_n, std::vector<uint32_t> *out)
{
    out->clear();
    if (log_n <= ctx->tile_log) {   // the whole transform fits one workgroup tile
        out->push_back(log_n);
        return;
    }
    uint32_t passes = (log_n + ctx->max_log_r - 1) / ctx->max_log_r;
    uint32_t base = log_n / passes, rem = log_n % passes;
    for (uint32_t i = 0; i < passes; i++) out->push_back(base + (i < rem ? 1 : 0));
}

// dst[k] = post^k * scale * sum_i (pre^i * src[i]) omega^(ik),  src[i] = 0 for i >= nnz.
// src may equal dst.  Caller holds ctx->mu.
// `lay` (6-step building blocks): column mode over a [2^log_n][width] array (batch must then be the
// width: it sizes the scratch), the 2D twiddle, split addressing of the first pass's input / the last
// pass's output.
int ntt_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n,
             const HFr &omega, uint64_t nnz, const HFr *scale, const HFr *pre, const HFr *post,
             uint32_t batch, const NttLayout *lay)
{
    std::vector<uint32_t> radices;
    plan_radices(ctx, log_n, &radices);
    const size_t passes = radices.size();
    const size_t bytes = ((size_t)32 << log_n) * batch;
    int rc = trim_table_cache(ctx);
    if (rc) return rc;

    TwoLevel tw = {nullptr, nullptr, 0}, pre_t = {nullptr, nullptr, 0}, post_t = {nullptr, nullptr, 0};
    // Split the exponent so that the second pass (exponents are multiples of n/(R1*R2)) needs only the
    // `hi` table — one multiplication per element instead of two — as long as `hi` stays L2-sized.
    uint32_t tw_lo_bits = 0xffffffffu;
    if (passes > 1) {
        uint32_t shift2 = log_n - radices[0] - radices[1];
        if (log_n - shift2 <= ctx->tw_hi_max_log && shift2 <= 16) tw_lo_bits = shift2;
    }
    if (passes > 1 && (rc = get_pow_table(ctx, omega, log_n, &tw, 2, tw_lo_bits))) return rc;
    // iNTT: fold the n^-1 scale into the `hi` half of the LAST pass's twiddle table — every element of
    // that pass is multiplied by a twiddle anyway (tw_always covers the exponent-0 ones), so the scale
    // costs no product of its own (it is a separate streaming pass on the CPU, src/polynomials/mod.rs:777-787)
    TwoLevel tw_last = tw;
    const bool fold_scale = scale && passes > 1;
    if (fold_scale && (rc = get_pow_table(ctx, omega, log_n, &tw_last, 2, tw_lo_bits, scale))) return rc;
    if (pre && (rc = get_pow_table(ctx, *pre, log_n, &pre_t, 2))) return rc;
    if (post && (rc = get_pow_table(ctx, *post, log_n, &post_t, 2))) return rc;
    TwoLevel tw2d_t = {nullptr, nullptr, 0};
    if (lay && lay->tw2d_root && (rc = get_pow_table(ctx, *lay->tw2d_root, lay->tw2d_log_order, &tw2d_t, 2))) return rc;

    // ping-pong buffers: the last pass writes dst; a pass never runs in place unless it is the only
    // one (a single tile is fully staged in LDS before anything is written back).
    std::vector<uint4 *> outs(passes);
    ScratchUse pool;
    if (passes == 1) {
        outs[0] = dst;
    } else {
        bool in_place = (src == dst);
        if ((rc = ensure_scratch(ctx, 0, bytes, stream))) return rc;
        pool.arm(ctx, stream);
        if (!dst && passes > 2) {
            // direct exchange: the last pass writes into the peers' buffers and there is no local destination to
            // ping-pong through — the second scratch buffer takes its place for the passes in between
            if ((rc = ensure_scratch(ctx, 1, bytes, stream))) return rc;
            dst = (uint4 *)ctx->scratch[1];
        }
        uint4 *s0 = (uint4 *)ctx->scratch[0], *s1 = nullptr;
        // walk backwards: pass P-1 -> dst, P-2 -> s0, P-3 -> dst (or s1 if that would clobber src)...
        for (size_t i = 0; i < passes; i++) {
            size_t from_end = passes - 1 - i;
            outs[i] = (from_end % 2 == 0) ? dst : s0;
        }
        if (in_place && outs[0] == dst) {
            // pass 0 would overwrite its own (strided) input: route passes 0 and 1 through s1/s0
            if ((rc = ensure_scratch(ctx, 1, bytes, stream))) return rc;
            s1 = (uint4 *)ctx->scratch[1];
            outs[0] = s1;   // then pass 1 -> s0, pass 2 -> dst, ... parity preserved
        }
    }

    uint32_t log_l = 0;
    const uint4 *cur = src;
    Fr9 scale_d = {};
    if (scale) scale_d = to_dev9(ctx->F, *scale);
    for (size_t i = 0; i < passes; i++) {
        uint32_t log_r = radices[i];
        PassArgs A = {};
        A.src = cur;
        A.dst = outs[i];
        if ((rc = get_radix_table(ctx, omega, log_n, log_r, &A.rtw, &A.rtw9))) return rc;
        A.tw = (i + 1 == passes) ? tw_last : tw;
        A.tw_always = (fold_scale && i + 1 == passes) ? 1 : 0;
        A.pre = (i == 0) ? pre_t : TwoLevel{nullptr, nullptr, 0};
        A.post = (i + 1 == passes) ? post_t : TwoLevel{nullptr, nullptr, 0};
        A.nnz = (i == 0) ? nnz : (1ull << log_n);
        A.log_n = log_n;
        A.log_r = log_r;
        // tile = 2^tile_log elements, but never fewer than 2^min_log_c columns: a radix-512 pass gets a
        // 2048-element tile (128-byte runs in HBM) rather than 1024 elements in 64-byte runs
        uint32_t log_c = ctx->tile_log > log_r ? ctx->tile_log - log_r : 0;
        if (log_c < ctx->min_log_c && log_r + ctx->min_log_c <= 11) log_c = ctx->min_log_c;   // <= 2048 elements: fits LDS with any twiddle table
        if (lay && lay->col_mode) {
            if (log_c > lay->log_width) log_c = lay->log_width;     // the tile's columns are array columns
            A.col_mode = 1;
            A.log_width = lay->log_width;
            A.col0 = lay->col0;
            // a pass reads what the previous one wrote (compact, log_width); only the first pass's source
            // and the last pass's destination may be the caller's wider arrays
            A.src_log_width = (i == 0 && lay->src_log_width) ? lay->src_log_width : lay->log_width;
            A.src_col_off = (i == 0) ? lay->src_col_off : 0;
            A.dst_log_width = (i + 1 == passes && lay->dst_log_width) ? lay->dst_log_width : lay->log_width;
            A.dst_col_off = (i + 1 == passes) ? lay->dst_col_off : 0;
            if (tw2d_t.lo && ((lay->tw2d_on_load && i == 0) || (!lay->tw2d_on_load && i + 1 == passes))) {
                A.tw2d = tw2d_t;
                A.tw2d_on_load = lay->tw2d_on_load ? 1 : 0;
            }
        } else if (log_c > log_n - log_r) {
            log_c = log_n - log_r;
        }
        if (lay && i == 0) A.src_split = lay->src_split;
        if (lay && i + 1 == passes) A.dst_split = lay->dst_split;
        if (lay && i + 1 == passes && lay->peer_tab) {
            A.peer_tab = lay->peer_tab;
            A.peer_off = lay->peer_off;
            A.peer_log = lay->peer_log;
            A.peer_self = lay->peer_self;
        }
        A.log_c = log_c;
        A.log_l = log_l;
        A.apply_tw = (i == 0) ? 0 : 1;
        A.batch = batch;
        A.src_batch_stride = (i == 0) ? nnz : (1ull << log_n);
        A.dbg = 0;
        A.log_skip = 0;
        if (i == 0 && nnz && nnz < (1ull << log_n) && !(nnz & (nnz - 1))) {
            uint32_t s = log_n - log2u((size_t)nnz);
            A.log_skip = s < log_r ? s : log_r;
        }
#ifdef HODOR_BOUNDS
        {   // what this pass may touch (bounds.cuh): the caller's arrays at the sizes the call promises, the ping-pong buffers
            const uint64_t n_el = 1ull << log_n;
            if (i == 0) {
                if (A.col_mode) A.bx_src_bytes = (32ull << log_n) << A.src_log_width;
                else if (A.src_split.on) A.bx_src_bytes = 32ull * n_el * batch;
                else A.bx_src_bytes = 32ull * nnz * batch;
            } else A.bx_src_bytes = bytes;
            if (i + 1 == passes) A.bx_dst_bytes = A.col_mode ? (32ull << log_n) << A.dst_log_width : 32ull * n_el * batch;
            else A.bx_dst_bytes = bytes;
            if (passes == 1 && src == dst && A.bx_src_bytes > A.bx_dst_bytes) A.bx_dst_bytes = A.bx_src_bytes;
            A.bx_rtw_bytes = (log_r ? (1ull << (log_r - 1)) : 1) * 112;
            A.bx_rtw9_bytes = A.rtw9 ? 16 * W9_WORDS * sizeof(uint32_t) : 0;
            A.bx_peers = 0;
            if (A.peer_tab && lay) {
                A.bx_peers = lay->bx_peers;
                A.bx_peer_bytes = lay->bx_peer_bytes;
                for (uint32_t t = 0; t < lay->bx_peers && t < 8; t++) A.bx_peer_host[t] = lay->bx_peer_host[t];
            }
        }
#endif
        HIPCHK(ntt_launch_pass(stream, A, (scale && !fold_scale && i + 1 == passes) ? &scale_d : nullptr, ctx->Q));
        cur = outs[i];
        log_l += log_r;
    }
    return HODOR_OK;
}

int poly_domain(hodor_ctx *ctx, uint32_t log_n, HFr *omega)
{
    uint64_t sz;
    uint32_t k;
    if (log_n > 63 || !ctx->F.domain(1ull << log_n, &sz, &k, omega)) {
        set_err(ctx, "domain too large for the field's 2-adicity");
        return HODOR_ERR_SIZE;
    }
    return HODOR_OK;
}

int poly_transform(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n, PolyOp op,
                   const HFr *gen)
{
    HFr omega;
    int rc = poly_domain(ctx, log_n, &omega);
    if (rc) return rc;
    uint64_t n = 1ull << log_n;
    switch (op) {
    case OP_FFT:
        return ntt_exec(ctx, stream, src, dst, log_n, omega, n, nullptr, nullptr, nullptr, 1, nullptr);
    case OP_COSET_FFT:   // distribute_powers(g) then fft — src/polynomials/mod.rs:626-631; `gen`: coset_fft_for_generator :633-638
        return ntt_exec(ctx, stream, src, dst, log_n, omega, n, nullptr, gen ? gen : &ctx->F.generator, nullptr);
    case OP_IFFT:
    case OP_ICOSET_FFT: {   // best_fft(omegainv) then * minv (then * geninv^i) — :773-807
        HFr oinv, minv, ginv;
        ctx->F.inverse(omega, &oinv);
        ctx->F.inverse(ctx->F.from_u64(n), &minv);
        ctx->F.inverse(ctx->F.generator, &ginv);
        if (gen) ginv = *gen;      // icoset_fft_for_generator (:809-815) is handed the INVERSE generator
        return ntt_exec(ctx, stream, src, dst, log_n, oinv, n, &minv, nullptr,
                        op == OP_ICOSET_FFT ? &ginv : nullptr);
    }
    }
    return HODOR_ERR_INVALID;
}

// lde / coset_lde: one zero-padded transform of size n*factor — identical output to the
// reference's per-coset schedule (asserted by its own tests, src/polynomials/mod.rs:1026-1031)
int poly_lde_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst,
                  uint32_t log_n, size_t factor, int coset, uint32_t batch)
{
    if (!is_pow2(factor)) { set_err(ctx, "lde factor must be a power of two"); return HODOR_ERR_SIZE; }
    uint32_t log_big = log_n + log2u(factor);
    HFr Omega;
    int rc = poly_domain(ctx, log_big, &Omega);
    if (rc) return rc;
    return ntt_exec(ctx, stream, src, dst, log_big, Omega, 1ull << log_n, nullptr,
                    coset ? &ctx->F.generator : nullptr, nullptr, batch);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_ctx_create(const uint64_t modulus[4], uint64_t generator, int device,
                                hodor_ctx **out)
{
    if (!modulus || !out) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = new (std::nothrow) hodor_ctx();
    if (!ctx) return HODOR_ERR_INVALID;
    if (!ctx->F.init(modulus, generator)) { delete ctx; return HODOR_ERR_INVALID; }
    // the 9 x 29-bit kernels keep the top bits of a value in limb 8 (bits 232..): 240 <= NUM_BITS <= 255
    // covers both 4-limb fields of the reference (255 and 252 bits)
    if (ctx->F.num_bits < 240) { delete ctx; return HODOR_ERR_INVALID; }
    for (int i = 0; i < 4; i++) {
        ctx->P.p[2 * i] = (uint32_t)ctx->F.p[i];
        ctx->P.p[2 * i + 1] = (uint32_t)(ctx->F.p[i] >> 32);
    }
    ctx->P.pinv = (uint32_t)ctx->F.pinv;
    Fr one = to_dev(ctx->F.one);
    for (int i = 0; i < 8; i++) ctx->P.one[i] = one.v[i];
    make_params9(ctx->F, &ctx->Q);
    {   // plain-integer 2^(87 c), c = 1..3: into_repr strips the R of the Montgomery power
        HFr two = ctx->F.from_u64(2);
        for (int c = 0; c < 3; c++) {
            HFr plain;
            ctx->F.into_repr(ctx->F.pow(two, 87 * (uint64_t)(c + 1)), plain.l);
            ctx->K3.k[c] = to_dev(plain);
        }
        for (int c = 0; c < 9; c++) {
            HFr plain;
            ctx->F.into_repr(ctx->F.pow(two, 29 * (uint64_t)(c + 1)), plain.l);
            ctx->K9.k[c] = to_dev(plain);
        }
    }
    // BASE_BLAKE2S_PARAMS, src/iop/blake2s_trivial_iop.rs:8-16
    HostBlake2s::keyed_midstate(ctx->mid.h, (const uint8_t *)"Squeamish Ossifrage", 19,
                                (const uint8_t *)"Shaftoe", 7);
    // COSET2 leaves (this build's opt-in format) hash under a personalisation of their own, so that the 64 bytes of two
    // child digests can never open as a "coset value pair" (round-4 advisor finding; the depth check stays as well)
    HostBlake2s::keyed_midstate(ctx->mid.hp, (const uint8_t *)"Squeamish Ossifrage", 19,
                                (const uint8_t *)"Shaftoe2", 8);
    ctx->device = device;
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device >= count ||
            hipSetDevice(device) != hipSuccess ||
            hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return HODOR_ERR_DEVICE;
        }
        const Knobs &k = knobs();
        ctx->max_log_r = (uint32_t)k.max_log_r;
        ctx->tw_hi_max_log = (uint32_t)k.tw_hi_max_log;
        ctx->tile_log = (uint32_t)k.tile_log;
        ctx->min_log_c = (uint32_t)k.min_log_c;
        ctx->pool_cache_cap = (size_t)k.pool_cache_gib << 30;
        if (ctx->max_log_r > ctx->tile_log) ctx->max_log_r = ctx->tile_log;
        // the kernels this context is about to use against the host implementations of the same arithmetic
        // (abi_selftest.hip); a context that fails is never handed out
        if (int rc = ctx_self_test(ctx)) {
            (void)hodor_ctx_try_destroy(ctx);
            return rc;
        }
    }
    *out = ctx;
    return HODOR_OK;
}

extern "C" int hodor_abi_version(void) { return HODOR_ABI_VERSION; }

extern "C" void hodor_ctx_destroy(hodor_ctx *ctx) { (void)hodor_ctx_try_destroy(ctx); }

extern "C" int hodor_ctx_try_destroy(hodor_ctx *ctx)
{
    if (!ctx) return HODOR_OK;
    if (ctx->live_exchanges.load() != 0) {
        // a hodor_exchange keeps a pointer to its context (error reporting, device): destroying the context under it
        // would leave that pointer dangling.  Refuse — the handle stays valid, the caller destroys the exchanges
        // first (header: "Destroy the handle before its context") and calls again.
        set_err(ctx, "hodor_ctx_destroy: exchanges created on this context are still alive; destroy them first");
        return HODOR_ERR_INVALID;
    }
    if (ctx->live_handles.load() != 0) {
        // polynomial / IOP handles and FRI prototypes give their device memory back to THIS context's pool when freed
        set_err(ctx, "hodor_ctx_destroy: " + std::to_string(ctx->live_handles.load()) +
                         " polynomial / IOP handles or FRI prototypes of this context are still alive; free them first");
        return HODOR_ERR_INVALID;
    }
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        for (auto &t : ctx->pow_tables) { BOUNDS_FORGET(t.lo); BOUNDS_FORGET(t.hi); (void)hipFree(t.lo); (void)hipFree(t.hi); }
        for (auto &t : ctx->radix_tables) { BOUNDS_FORGET(t.rtw); BOUNDS_FORGET(t.rtw9); (void)hipFree(t.rtw); if (t.rtw9) (void)hipFree(t.rtw9); }
        for (int i = 0; i < 2; i++)
            if (ctx->scratch[i]) { BOUNDS_FORGET(ctx->scratch[i]); (void)hipFree(ctx->scratch[i]); }
        if (ctx->scratch_ev) (void)hipEventDestroy(ctx->scratch_ev);
        pool_drain(ctx);
        pool_destroy_events(ctx);
        host_images_drain(ctx);
        for (auto &L : ctx->lanes) {
            for (int i = 0; i < 2; i++)
                if (L.buf[i]) { BOUNDS_FORGET(L.buf[i]); (void)hipFree(L.buf[i]); }
            if (L.uploaded) (void)hipEventDestroy(L.uploaded);
            if (L.computed) (void)hipEventDestroy(L.computed);
            if (L.stream) (void)hipStreamDestroy(L.stream);
        }
        if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
        if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        stage_destroy(ctx->stage_up);
        stage_destroy(ctx->stage_down);
        for (auto &a : ctx->aux_streams)
            if (a) (void)hipStreamDestroy(a);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return HODOR_OK;
}

extern "C" int hodor_ctx_field_info(const hodor_ctx *ctx, hodor_field_info *out)
{
    if (!ctx || !out) return HODOR_ERR_INVALID;
    memcpy(out->modulus, ctx->F.p, 32);
    out->s = ctx->F.s;
    out->num_bits = ctx->F.num_bits;
    out->capacity = ctx->F.capacity;
    from_h(ctx->F.one, &out->one);
    from_h(ctx->F.generator, &out->generator);
    from_h(ctx->F.root_of_unity, &out->root_of_unity);
    return HODOR_OK;
}

// The message is copied under the error mutex into a per-thread buffer: another thread's failing call may
// reassign ctx->err at any time, so a pointer into it could dangle.  Valid until this thread's next call here.
extern "C" const char *hodor_last_error(const hodor_ctx *ctx)
{
    if (!ctx) return ctx_create_error();   // why this thread's last hodor_ctx_create failed ("" when it did not)
    static thread_local std::string snapshot;
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    snapshot = ctx->err;
    return snapshot.c_str();
}

extern "C" int hodor_ctx_synchronize(hodor_ctx *ctx)
{
    NEED_DEVICE();
    HIPCHK(hipDeviceSynchronize());
#ifdef HODOR_BOUNDS
    if (hodor::bounds_poll(ctx)) return HODOR_ERR_DEVICE;
#endif
    pool_collect(ctx);   // the device is idle: blocks evicted from the pool's cache go back to HIP now
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// device API
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_buf_alloc(hodor_ctx *ctx, size_t bytes, void **dev_ptr)
{
    NEED_DEVICE();
    if (!dev_ptr) return HODOR_ERR_INVALID;
    HIPCHK(dev_malloc(dev_ptr, bytes ? bytes : 1));
    BOUNDS_NOTE(*dev_ptr, bytes);
    return HODOR_OK;
}
extern "C" int hodor_buf_free(hodor_ctx *ctx, void *dev_ptr)
{
    NEED_DEVICE();
    BOUNDS_FORGET(dev_ptr);
    HIPCHK(hipFree(dev_ptr));
    return HODOR_OK;
}
extern "C" int hodor_buf_upload(hodor_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    NEED_DEVICE();
    if (bytes <= hodor_ctx::PINNED_BYTES) {   // small: through the context's pinned buffer (ctx.hpp), never a pin of the caller's page
        HostXfer xfer(ctx, nullptr);
        HIPCHK(xfer.h2d(dev_dst, host_src, bytes));
        HIPCHK(xfer.finish());
        return HODOR_OK;
    }
    HostXfer xfer(ctx, nullptr);   // (counts the bytes; pageable memory goes through the staging ring)
    HIPCHK(xfer.h2d(dev_dst, host_src, bytes));
    HIPCHK(xfer.finish());
    return HODOR_OK;
}
extern "C" int hodor_buf_download(hodor_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    NEED_DEVICE();
    if (bytes <= hodor_ctx::PINNED_BYTES) {
        HostXfer xfer(ctx, nullptr);
        HIPCHK(xfer.d2h(host_dst, dev_src, bytes));
        HIPCHK(xfer.finish());
    } else {
        HostXfer xfer(ctx, nullptr);
        HIPCHK(xfer.d2h(host_dst, dev_src, bytes));
        HIPCHK(xfer.finish());
    }
    note_round_trip(ctx);
    return HODOR_OK;
}

// Pinning of caller-owned host memory: the slice API's copies from/to a registered range are true DMA
// at the link rate instead of staged copies through the driver's bounce buffers.
extern "C" int hodor_host_register(hodor_ctx *ctx, void *host_ptr, size_t bytes)
{
    NEED_DEVICE();
    if (!host_ptr || !bytes) return HODOR_ERR_INVALID;
    HIPCHK(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return HODOR_OK;
}
extern "C" int hodor_host_unregister(hodor_ctx *ctx, void *host_ptr)
{
    NEED_DEVICE();
    if (!host_ptr) return HODOR_ERR_INVALID;
    HIPCHK(hipHostUnregister(host_ptr));
    return HODOR_OK;
}

extern "C" int hodor_fft_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                             uint32_t log_n, const hodor_fr *omega)
{
    NEED_DEVICE();
    if (!src || !dst || !omega) return HODOR_ERR_INVALID;
    if (log_n > ctx->F.s || log_n > 40) return HODOR_ERR_SIZE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, to_h(omega),
                    1ull << log_n, nullptr, nullptr, nullptr);
}

static int poly_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst, uint32_t log_n,
                    PolyOp op, const hodor_fr *gen = nullptr)
{
    NEED_DEVICE();
    if (!src || !dst) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HFr g;
    if (gen) g = to_h(gen);
    return poly_transform(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, op, gen ? &g : nullptr);
}
extern "C" int hodor_poly_coset_fft_for_generator_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst,
                                                      uint32_t log_n, const hodor_fr *gen)
{ return gen ? poly_dev(ctx, s, src, dst, log_n, OP_COSET_FFT, gen) : HODOR_ERR_INVALID; }
extern "C" int hodor_poly_icoset_fft_for_generator_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst,
                                                       uint32_t log_n, const hodor_fr *geninv)
{ return geninv ? poly_dev(ctx, s, src, dst, log_n, OP_ICOSET_FFT, geninv) : HODOR_ERR_INVALID; }
extern "C" int hodor_poly_fft_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst, uint32_t log_n)
{ return poly_dev(ctx, s, src, dst, log_n, OP_FFT); }
extern "C" int hodor_poly_ifft_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst, uint32_t log_n)
{ return poly_dev(ctx, s, src, dst, log_n, OP_IFFT); }
extern "C" int hodor_poly_coset_fft_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst, uint32_t log_n)
{ return poly_dev(ctx, s, src, dst, log_n, OP_COSET_FFT); }
extern "C" int hodor_poly_icoset_fft_dev(hodor_ctx *ctx, void *s, const hodor_fr *src, hodor_fr *dst, uint32_t log_n)
{ return poly_dev(ctx, s, src, dst, log_n, OP_ICOSET_FFT); }

extern "C" int hodor_poly_lde_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                  uint32_t log_n, size_t factor, int coset)
{
    NEED_DEVICE();
    if (!src || !dst) return HODOR_ERR_INVALID;
    if (src == dst && factor != 1) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_lde_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, factor,
                         coset);
}

// Batched multi-column LDE and commit: all registers' f_ldes and their oracles in
// one call each (src/prover/mod.rs:73-80 loops over the witness polynomials).
extern "C" int hodor_poly_lde_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                        uint32_t log_n, size_t factor, int coset, size_t batch)
{
    NEED_DEVICE();
    if (!src || !dst || src == dst) return HODOR_ERR_INVALID;
    if (batch == 0 || batch > 65535) return HODOR_ERR_SIZE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_lde_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, factor, coset,
                         (uint32_t)batch);
}

extern "C" int hodor_iop_create_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n,
                                          size_t batch, uint8_t *nodes)
{
    NEED_DEVICE();
    if (!leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2 || batch == 0 || batch > 65535) return HODOR_ERR_SIZE;
    HIPCHK(merkle_build_launch(pick_stream(ctx, stream), (const uint4 *)leafs, (uint4 *)nodes, n, ctx->mid,
                               (uint32_t)batch));
    return HODOR_OK;
}

// the batched commit in the chosen tree format (COSET2: batch node arrays of (n/2)*32 bytes back to back)
extern "C" int hodor_iop_create_batch_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n,
                                                   size_t batch, int combiner, uint8_t *nodes)
{
    if (combiner == HODOR_COMBINER_TRIVIAL) return hodor_iop_create_batch_dev(ctx, stream, leafs, n, batch, nodes);
    NEED_DEVICE();
    if (combiner != HODOR_COMBINER_COSET2 || !leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4 || batch == 0 || batch > 65535) return HODOR_ERR_SIZE;
    HIPCHK(merkle_build_launch(pick_stream(ctx, stream), (const uint4 *)leafs, (uint4 *)nodes, n, ctx->mid,
                               (uint32_t)batch, nullptr, nullptr, true));
    return HODOR_OK;
}

// a[i] *= g^i through the cached two-level table of g (caller holds ctx->mu)
static int distribute_powers_exec(hodor_ctx *ctx, hipStream_t stream, uint4 *a, size_t n, const HFr &g)
{
    if (n == 0) return HODOR_OK;
    if (n < ((size_t)1 << 16)) {   // not worth a table (two allocations and a generation kernel)
        HIPCHK(distribute_powers_small_launch(stream, a, n, to_dev(g), ctx->P));
        return HODOR_OK;
    }
    uint32_t log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    int rc = trim_table_cache(ctx);
    if (rc) return rc;
    TwoLevel t;
    if ((rc = get_pow_table(ctx, g, log_n, &t, 1))) return rc;
    HIPCHK(distribute_powers_launch(stream, a, n, t, ctx->Q));
    return HODOR_OK;
}

extern "C" int hodor_distribute_powers_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t n,
                                           const hodor_fr *g)
{
    NEED_DEVICE();
    if (!a || !g) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return distribute_powers_exec(ctx, pick_stream(ctx, stream), (uint4 *)a, n, to_h(g));
}

extern "C" int hodor_poly_degree_one_on_domain_dev(hodor_ctx *ctx, void *stream_, hodor_fr *out, size_t n,
                                                  const hodor_fr *alpha, const hodor_fr *c, int coset)
{
    NEED_DEVICE();
    if (!out || !alpha || !c) return HODOR_ERR_INVALID;
    uint64_t size;
    uint32_t log_n;
    HFr w;
    if (n == 0 || (n & (n - 1)) || !ctx->F.domain(n, &size, &log_n, &w) || size != n) {
        set_err(ctx, "degree_one_on_domain: the size must be a power of two within the field's two-adicity");
        return HODOR_ERR_SIZE;
    }
    hipStream_t stream = pick_stream(ctx, stream_);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const HFr a = coset ? ctx->F.mul(to_h(alpha), ctx->F.generator) : to_h(alpha);   // :277 u starts at the coset factor
    if (n < ((size_t)1 << 16)) {
        HIPCHK(degree_one_small_launch(stream, (uint4 *)out, n, to_dev(w), to_dev(a), to_dev(to_h(c)), ctx->P));
        return HODOR_OK;
    }
    int rc = trim_table_cache(ctx);
    if (rc) return rc;
    TwoLevel t;
    if ((rc = get_pow_table(ctx, w, log_n, &t, 1))) return rc;
    HIPCHK(degree_one_launch(stream, (uint4 *)out, n, t, fr9_from_host(a), fr9_from_host(to_h(c)), ctx->Q));
    return HODOR_OK;
}

extern "C" int hodor_precomputed_omegas_dev(hodor_ctx *ctx, void *stream, uint32_t log_n, hodor_fr *omegas,
                                            hodor_fr *coset, hodor_fr *omegas_inv)
{
    NEED_DEVICE();
    uint64_t size;
    uint32_t lg;
    HFr w, winv;
    if (log_n > 40 || !ctx->F.domain(1ull << log_n, &size, &lg, &w)) {
        set_err(ctx, "domain larger than the field's two-adicity");
        return HODOR_ERR_SIZE;
    }
    if (!ctx->F.inverse(w, &winv)) return HODOR_ERR_INVALID;
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t n = 1ull << log_n;
    if (omegas) HIPCHK(pow_table_launch(s, (uint4 *)omegas, to_dev(w), to_dev(ctx->F.one), 0, n, 0, ctx->P));
    if (coset) HIPCHK(pow_table_launch(s, (uint4 *)coset, to_dev(w), to_dev(ctx->F.generator), 0, n, 0, ctx->P));
    if (omegas_inv && n >= 2)
        HIPCHK(pow_table_launch(s, (uint4 *)omegas_inv, to_dev(winv), to_dev(ctx->F.one), 0, n / 2, 0, ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_fft_batch_dev(hodor_ctx *ctx, void *stream, const hodor_fr *src, hodor_fr *dst,
                                   uint32_t log_n, size_t batch, const hodor_fr *omega)
{
    NEED_DEVICE();
    if (!src || !dst || !omega) return HODOR_ERR_INVALID;
    if (log_n > ctx->F.s || log_n > 40 || batch == 0 || batch > 65535) return HODOR_ERR_SIZE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_exec(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, to_h(omega),
                    1ull << log_n, nullptr, nullptr, nullptr, (uint32_t)batch);
}

extern "C" int hodor_twiddle_mul_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t rows, size_t cols,
                                     uint64_t row0, const hodor_fr *omega, uint32_t log_order,
                                     const hodor_fr *scale)
{
    NEED_DEVICE();
    if (!a || !omega) return HODOR_ERR_INVALID;
    if (log_order > 62) return HODOR_ERR_SIZE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    TwoLevel t;
    int rc = trim_table_cache(ctx);
    if (rc) return rc;
    rc = get_pow_table(ctx, to_h(omega), log_order, &t, 0);
    if (rc) return rc;
    Fr sc = {};
    if (scale) sc = to_dev(to_h(scale));
    HIPCHK(twiddle_mul_launch(pick_stream(ctx, stream), (uint4 *)a, rows, cols, row0, t, log_order,
                              scale ? &sc : nullptr, ctx->P));
    return HODOR_OK;
}

// ---- value-form polynomial arithmetic on device ----
extern "C" int hodor_poly_binary_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, const hodor_fr *b, size_t n, int op)
{
    NEED_DEVICE();
    if (!a || !b) return HODOR_ERR_INVALID;
    if (op < 0 || op > 2) return HODOR_ERR_INVALID;
    HIPCHK(binary_launch(pick_stream(ctx, stream), (uint4 *)a, (const uint4 *)b, n, op, ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_poly_add_scaled_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, const hodor_fr *b, size_t n,
                                         const hodor_fr *scaling)
{
    NEED_DEVICE();
    if (!a || !b || !scaling) return HODOR_ERR_INVALID;
    HIPCHK(add_scaled_launch(pick_stream(ctx, stream), (uint4 *)a, (const uint4 *)b, n, to_dev(to_h(scaling)), ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_poly_unary_dev(hodor_ctx *ctx, void *stream, hodor_fr *a, size_t n, int op, const hodor_fr *c,
                                    uint64_t e)
{
    NEED_DEVICE();
    if (!a || op < 0 || op > 5 || (op >= 3 && !c)) return HODOR_ERR_INVALID;
    Fr cd = {};
    if (c) cd = to_dev(to_h(c));
    if (op == 2 && e == 2) op = 1;   // pow(2) is square (src/polynomials/mod.rs:746-748)
    HIPCHK(unary_launch(pick_stream(ctx, stream), (uint4 *)a, n, op, cd, e, ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_poly_quotient_term_dev(hodor_ctx *ctx, void *stream, hodor_fr *acc, const hodor_fr *f,
                                            const hodor_fr *divisor_inv, size_t n, const hodor_fr *value,
                                            const hodor_fr *alpha, int accumulate)
{
    NEED_DEVICE();
    if (!acc || !f || !divisor_inv || !value) return HODOR_ERR_INVALID;
    Fr a = {};
    if (alpha) a = to_dev(to_h(alpha));
    HIPCHK(quotient_term_launch(pick_stream(ctx, stream), (uint4 *)acc, (const uint4 *)f, (const uint4 *)divisor_inv, n,
                                to_dev(to_h(value)), alpha ? &a : nullptr, accumulate != 0, ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_gen_elements_dev(hodor_ctx *ctx, void *stream, hodor_fr *dst, uint64_t first_index,
                                      size_t count, uint64_t seed)
{
    NEED_DEVICE();
    if (!dst && count) return HODOR_ERR_INVALID;
    // the top limb's mask; contexts are only created for 240 <= NUM_BITS <= 255 (hodor_ctx_create), guarded anyway
    if (ctx->F.num_bits <= 192) return HODOR_ERR_INVALID;
    const uint64_t top_mask = ctx->F.num_bits >= 256 ? ~0ull : ((1ull << (ctx->F.num_bits - 192)) - 1);
    HIPCHK(gen_elements_launch(pick_stream(ctx, stream), (uint4 *)dst, first_index, count, seed, top_mask,
                               to_dev(ctx->F.r2), ctx->P));
    return HODOR_OK;
}

extern "C" int hodor_poly_batch_inversion_dev(hodor_ctx *ctx, void *stream_, hodor_fr *a, size_t n)
{
    NEED_DEVICE();
    if (!a) return HODOR_ERR_INVALID;
    if (n == 0) return HODOR_OK;
    hipStream_t stream = pick_stream(ctx, stream_);
    std::lock_guard<std::mutex> lk(ctx->mu);
    // levels: n -> T0 = n/8 subsequence products -> T1 = T0/8 -> ... until at most TOP are left.  Those
    // come to the host together with the zero flag (one synchronisation, needed anyway to leave the data
    // untouched on error) and are inverted there: a Fermat inversion is ~380 dependent products, 20 us on
    // a CPU core but 0.4 ms of latency for a lone GPU lane.
    constexpr uint64_t TOP = 16;
    struct Level { uint64_t n, T; size_t prefix_off, prod_off; };
    std::vector<Level> levels;
    size_t off = 0;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int seq = knobs().batchinv_seq;   // elements per thread and level
    for (uint64_t m = n; m > TOP || levels.empty();) {
        uint64_t T = (m + seq - 1) / seq;
        Level L = {m, T, off, 0};
        off += up(m * 32);
        L.prod_off = off;
        off += up(T * 32);
        levels.push_back(L);
        m = T;
    }
    int rc = ensure_scratch(ctx, 0, off + 256, stream);
    if (rc) return rc;
    ScratchUse pool;
    pool.arm(ctx, stream);
    uint8_t *base = (uint8_t *)ctx->scratch[0];
    uint32_t *flag = (uint32_t *)(base + off);
    HIPCHK(hipMemsetAsync(flag, 0, 4, stream));
    const uint4 *cur = (const uint4 *)a;
    for (size_t l = 0; l < levels.size(); l++) {   // forward: nothing of `a` is modified yet
        const Level &L = levels[l];
        HIPCHK(batchinv_forward_launch(stream, cur, L.n, L.T, (uint4 *)(base + L.prefix_off), (uint4 *)(base + L.prod_off),
                                       l == 0 ? flag : nullptr, ctx->P));
        cur = (const uint4 *)(base + L.prod_off);
    }
    const Level &top = levels.back();
    hodor_fr top_prod[TOP];
    uint32_t host_flag = 0;
    HostXfer xfer(ctx, stream);      // through the context's pinned buffer: no stack page is ever pinned (ctx.hpp)
    HIPCHK(xfer.d2h(top_prod, base + top.prod_off, top.T * 32));
    HIPCHK(xfer.d2h(&host_flag, flag, 4));
    HIPCHK(xfer.finish());
    note_round_trip(ctx);
    if (host_flag) {   // full_grand_product.inverse() is None -> SynthesisError::Error, data untouched (:909)
        set_err(ctx, "batch_inversion: zero element");
        return HODOR_ERR_INVALID;
    }
    {   // the <= TOP remaining products: Montgomery's trick once more, on the host — ONE Fermat inversion (~20 us) and
        // 3 (T - 1) products instead of T inversions
        HFr v[TOP], prefix[TOP];
        HFr acc = ctx->F.one;
        for (uint64_t i = 0; i < top.T; i++) {
            v[i] = to_h(&top_prod[i]);
            prefix[i] = acc;                     // v[0] * ... * v[i-1]
            acc = ctx->F.mul(acc, v[i]);
        }
        HFr inv;
        if (!ctx->F.inverse(acc, &inv)) { set_err(ctx, "batch_inversion: zero product"); return HODOR_ERR_INVALID; }
        for (uint64_t i = top.T; i-- > 0;) {
            from_h(ctx->F.mul(inv, prefix[i]), &top_prod[i]);   // (v[0..i])^-1 * v[0..i-1] = v[i]^-1
            inv = ctx->F.mul(inv, v[i]);
        }
    }
    HIPCHK(xfer.h2d(base + top.prod_off, top_prod, top.T * 32));
    for (size_t l = levels.size(); l-- > 0;) {
        const Level &L = levels[l];
        uint4 *target = l == 0 ? (uint4 *)a : (uint4 *)(base + levels[l - 1].prod_off);
        HIPCHK(batchinv_backward_launch(stream, target, L.n, L.T, (const uint4 *)(base + L.prefix_off),
                                        (const uint4 *)(base + L.prod_off), ctx->P));
    }
    HIPCHK(xfer.finish());   // the call is synchronous: a[] holds the inverses on return
    return HODOR_OK;
}

extern "C" int hodor_poly_evaluate_at_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *coeffs, size_t n,
                                          const hodor_fr *g, hodor_fr *out)
{
    NEED_DEVICE();
    if (!coeffs || !g || !out) return HODOR_ERR_INVALID;
    hipStream_t stream = pick_stream(ctx, stream_);
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint32_t log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    const bool table = n >= ((size_t)1 << 16);   // below that a table is not worth its two allocations
    const size_t blocks = table ? evaluate_at_table_blocks(log_n) : 256;
    int rc = ensure_scratch(ctx, 0, 32 * (blocks + 2) + 64, stream);
    if (rc) return rc;
    ScratchUse pool;
    pool.arm(ctx, stream);
    uint4 *partials = (uint4 *)ctx->scratch[0];
    uint4 *res = partials + 2 * blocks;
    uint32_t *ticket = (uint32_t *)(res + 2);
    HIPCHK(hipMemsetAsync(ticket, 0, 4, stream));
    if (table) {
        TwoLevel t;
        if ((rc = trim_table_cache(ctx)) || (rc = get_pow_table(ctx, to_h(g), log_n, &t, 1))) return rc;
        HIPCHK(evaluate_at_table_launch(stream, (const uint4 *)coeffs, n, log_n, t, partials, ticket, res, ctx->Q, ctx->P));
    } else {
        HIPCHK(evaluate_at_launch(stream, (const uint4 *)coeffs, n, to_dev(to_h(g)), partials, ticket, res, ctx->P));
    }
    HostXfer xfer(ctx, stream);
    HIPCHK(xfer.d2h(out, res, 32));
    HIPCHK(xfer.finish());
    note_round_trip(ctx);
    return HODOR_OK;
}

// IopTree::create with the tree format chosen by `combiner` (HODOR_COMBINER_*): COSET2 hashes the coset
// {k, k + n/2} as ONE 64-byte leaf and writes the (n/2)-entry heap array of the tree over those n/2 leaves.
extern "C" int hodor_iop_create_combined_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n, int combiner,
                                             uint8_t *nodes)
{
    if (combiner == HODOR_COMBINER_TRIVIAL) return hodor_iop_create_dev(ctx, stream, leafs, n, nodes);
    NEED_DEVICE();
    if (combiner != HODOR_COMBINER_COSET2 || !leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4) { set_err(ctx, "iop_create (COSET2): n must be a power of two >= 4"); return HODOR_ERR_SIZE; }
    HIPCHK(merkle_build_launch(pick_stream(ctx, stream), (const uint4 *)leafs, (uint4 *)nodes, n, ctx->mid, 1, nullptr,
                               nullptr, true));
    return HODOR_OK;
}

extern "C" int hodor_iop_create_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n,
                                    uint8_t *nodes)
{
    NEED_DEVICE();
    if (!leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) { set_err(ctx, "iop_create: n must be a power of two >= 2"); return HODOR_ERR_SIZE; }
    HIPCHK(merkle_build_launch(pick_stream(ctx, stream), (const uint4 *)leafs, (uint4 *)nodes, n, ctx->mid));
    return HODOR_OK;
}

// ------------------------------------------------------------------------------------------------
// slice API: host memory in, host memory out
// ------------------------------------------------------------------------------------------------

// run `op` on a device copy of a[0..n_in) producing n_out elements back into `out`
// Host slice in -> device -> `op` -> device -> host slice out.  The copies run on the caller's lane
// (its own stream and buffers) WITHOUT the context mutex; only the enqueue of `op` on the context's
// compute stream is serialised (twiddle caches and pass scratch are shared).  Events chain
// upload -> compute -> download, so with several threads inside the library the PCIe link is busy in
// both directions while the kernels of a third caller run.  `separate_out`: `op` may not work in place.
// HODOR_SLICE_TRACE=1 (debugging aid of bench/slice_threads.cpp, never set in production): one line per call on stderr with
// the call's phase boundaries in microseconds since the first traced call.
static bool slice_trace_on()
{
    static const bool on = [] { const char *e = getenv("HODOR_SLICE_TRACE"); return e && *e && *e != '0'; }();
    return on;
}
static double slice_trace_us()
{
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

template <class Op>
static int with_device_copy(hodor_ctx *ctx, const void *in, size_t n_in, void *out, size_t n_out, Op op,
                            bool separate_out = false)
{
    const bool trace = slice_trace_on();
    double tr[8] = {0};
    if (trace) tr[0] = slice_trace_us();
    hodor_ctx::IoLane *L = lane_acquire(ctx);
    const bool serial = knobs().slice_serial != 0;
    hipStream_t us = nullptr, ds = nullptr;      // the streams this call uploads and downloads on
    auto fail = [&](hipError_t e, const char *what) {
        if (us) (void)hipStreamSynchronize(us);
        if (ds && ds != us) (void)hipStreamSynchronize(ds);
        // set_err takes err_mu only: this runs while a HostXfer (pinned_mu) may still be in scope, and the commit /
        // batch_inversion / evaluate_at paths take ctx->mu BEFORE pinned_mu — taking ctx->mu here was an inversion
        if (e != hipErrorAssert) set_err(ctx, std::string(what) + ": " + hipGetErrorString(e));   // hipErrorAssert: the bounds build's poll wrote the message
        lane_release(ctx, L);
        return HODOR_ERR_DEVICE;
    };
    hipError_t e;
    if ((e = lane_prepare(L, 0, (n_in ? n_in : 1) * 32)) != hipSuccess) return fail(e, "slice staging (in)");
    void *din = L->buf[0], *dptr_out = din;
    if (n_out != n_in || separate_out) {
        if ((e = lane_prepare(L, 1, (n_out ? n_out : 1) * 32)) != hipSuccess) return fail(e, "slice staging (out)");
        dptr_out = L->buf[1];
    }
    if (serial && (e = dir_streams_prepare(ctx)) != hipSuccess) return fail(e, "slice streams");
    us = serial ? ctx->up_stream : L->stream;
    ds = serial ? ctx->down_stream : L->stream;
    {
        std::unique_lock<std::mutex> up(ctx->up_mu, std::defer_lock);
        if (serial) up.lock();
        if (trace) tr[1] = slice_trace_us();
        const bool small_in = n_in * 32 <= hodor_ctx::PINNED_BYTES;   // small slices go through the context's pinned buffer
        if (small_in) {
            HostXfer xfer(ctx, us);
            if ((e = xfer.h2d(din, in, n_in * 32)) != hipSuccess || (e = xfer.finish()) != hipSuccess) return fail(e, "slice upload");
        }
        if (!small_in) ctx->h2d_bytes.fetch_add(n_in * 32, std::memory_order_relaxed);
        if ((!small_in && (e = host_is_pinned(in) ? hipMemcpyAsync(din, in, n_in * 32, hipMemcpyHostToDevice, us)
                                                  : staged_h2d(ctx, us, din, in, n_in * 32)) != hipSuccess) ||
            (e = hipEventRecord(L->uploaded, us)) != hipSuccess ||
            (serial && (e = hipStreamSynchronize(us)) != hipSuccess))   // the link is free for the next upload
            return fail(e, "slice upload");
        if (trace) tr[2] = slice_trace_us();
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (trace) tr[3] = slice_trace_us();
        e = hipStreamWaitEvent(ctx->stream, L->uploaded, 0);
        rc = e == hipSuccess ? op((const uint4 *)din, (uint4 *)dptr_out) : HODOR_ERR_DEVICE;
        if (e == hipSuccess) e = hipEventRecord(L->computed, ctx->stream);
        if (rc || e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);
    }
    if (rc) { (void)hipStreamSynchronize(us); lane_release(ctx, L); return rc; }
    if (e != hipSuccess) return fail(e, "slice compute");
    {
        std::unique_lock<std::mutex> down(ctx->down_mu, std::defer_lock);
        if (serial) {
            if ((e = hipEventSynchronize(L->computed)) != hipSuccess) return fail(e, "slice compute");   // wait OUTSIDE the lock
            if (trace) tr[4] = slice_trace_us();
            down.lock();
        }
        if (trace) tr[5] = slice_trace_us();
        if ((e = hipStreamWaitEvent(ds, L->computed, 0)) != hipSuccess) return fail(e, "slice download");
        if (n_out * 32 <= hodor_ctx::PINNED_BYTES) {
            HostXfer xfer(ctx, ds);
            if ((e = xfer.d2h(out, dptr_out, n_out * 32)) != hipSuccess || (e = xfer.finish()) != hipSuccess)
                return fail(e, "slice download");
        } else if ((ctx->d2h_bytes.fetch_add(n_out * 32, std::memory_order_relaxed), false) ||
                   (e = host_is_pinned(out) ? hipMemcpyAsync(out, dptr_out, n_out * 32, hipMemcpyDeviceToHost, ds)
                                            : staged_d2h(ctx, ds, out, dptr_out, n_out * 32)) != hipSuccess ||
                   (e = hipStreamSynchronize(ds)) != hipSuccess)
            return fail(e, "slice download");
        if (trace) tr[6] = slice_trace_us();
    }
    if (trace)
        fprintf(stderr, "slice lane %d n=%zu: enter %.0f  up [%.0f %.0f]  op enq %.0f  computed %.0f  down [%.0f %.0f] us\n",
                (int)(L - &ctx->lanes[0]), n_in, tr[0], tr[1], tr[2], tr[3], tr[4], tr[5], tr[6]);
    lane_release(ctx, L);
    return HODOR_OK;
}

extern "C" int hodor_fft(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *omega, uint32_t log_n)
{
    NEED_DEVICE();
    if (!a || !omega) return HODOR_ERR_INVALID;
    if (log_n > 40 || n != ((size_t)1 << log_n)) { set_err(ctx, "fft: n != 1 << log_n"); return HODOR_ERR_SIZE; }   // assert_eq at src/fft/fft.rs:34
    HFr w = to_h(omega);
    return with_device_copy(ctx, a, n, a, n, [&](const uint4 *s, uint4 *d) {
        return ntt_exec(ctx, ctx->stream, s, d, log_n, w, n, nullptr, nullptr, nullptr);
    });
}

extern "C" int hodor_lde(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *omega, uint32_t log_n,
                         size_t lde_factor)
{
    NEED_DEVICE();
    if (!a || !omega) return HODOR_ERR_INVALID;
    if (log_n > 40 || n != ((size_t)1 << log_n) || !is_pow2(lde_factor) || lde_factor > n) return HODOR_ERR_SIZE;
    HFr w = to_h(omega);
    return with_device_copy(ctx, a, n, a, n, [&](const uint4 *s, uint4 *d) {
        return ntt_exec(ctx, ctx->stream, s, d, log_n, w, n / lde_factor, nullptr, nullptr, nullptr);
    });
}

extern "C" int hodor_distribute_powers(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *g)
{
    NEED_DEVICE();
    if (!a || !g) return HODOR_ERR_INVALID;
    HFr gh = to_h(g);
    return with_device_copy(ctx, a, n, a, n, [&](const uint4 *s, uint4 *d) -> int {
        (void)s;
        return distribute_powers_exec(ctx, ctx->stream, d, n, gh);
    });
}

static int poly_slice(hodor_ctx *ctx, hodor_fr *a, size_t n, PolyOp op, const hodor_fr *gen = nullptr)
{
    NEED_DEVICE();
    if (!a) return HODOR_ERR_INVALID;
    if (!is_pow2(n)) { set_err(ctx, "polynomial size must be a power of two"); return HODOR_ERR_SIZE; }
    uint32_t log_n = log2u(n);
    HFr g;
    if (gen) g = to_h(gen);
    return with_device_copy(ctx, a, n, a, n, [&](const uint4 *s, uint4 *d) {
        return poly_transform(ctx, ctx->stream, s, d, log_n, op, gen ? &g : nullptr);
    });
}
extern "C" int hodor_poly_coset_fft_for_generator(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *gen)
{ return gen ? poly_slice(ctx, a, n, OP_COSET_FFT, gen) : HODOR_ERR_INVALID; }
extern "C" int hodor_poly_icoset_fft_for_generator(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *geninv)
{ return geninv ? poly_slice(ctx, a, n, OP_ICOSET_FFT, geninv) : HODOR_ERR_INVALID; }
extern "C" int hodor_poly_fft(hodor_ctx *ctx, hodor_fr *a, size_t n) { return poly_slice(ctx, a, n, OP_FFT); }
extern "C" int hodor_poly_coset_fft(hodor_ctx *ctx, hodor_fr *a, size_t n) { return poly_slice(ctx, a, n, OP_COSET_FFT); }
extern "C" int hodor_poly_ifft(hodor_ctx *ctx, hodor_fr *a, size_t n) { return poly_slice(ctx, a, n, OP_IFFT); }
extern "C" int hodor_poly_icoset_fft(hodor_ctx *ctx, hodor_fr *a, size_t n) { return poly_slice(ctx, a, n, OP_ICOSET_FFT); }

static int poly_lde_slice(hodor_ctx *ctx, const hodor_fr *coeffs, size_t n, size_t factor, hodor_fr *out,
                          int coset)
{
    NEED_DEVICE();
    if (!coeffs || !out) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || !is_pow2(factor)) return HODOR_ERR_SIZE;
    uint32_t log_n = log2u(n);
    return with_device_copy(ctx, coeffs, n, out, n * factor, [&](const uint4 *s, uint4 *d) {
        return poly_lde_exec(ctx, ctx->stream, s, d, log_n, factor, coset);
    });
}
extern "C" int hodor_poly_lde(hodor_ctx *ctx, const hodor_fr *c, size_t n, size_t f, hodor_fr *out)
{ return poly_lde_slice(ctx, c, n, f, out, 0); }
extern "C" int hodor_poly_coset_lde(hodor_ctx *ctx, const hodor_fr *c, size_t n, size_t f, hodor_fr *out)
{ return poly_lde_slice(ctx, c, n, f, out, 1); }

extern "C" int hodor_iop_create_combined(hodor_ctx *ctx, const hodor_fr *leafs, size_t n, int combiner, uint8_t *nodes)
{
    if (combiner == HODOR_COMBINER_TRIVIAL) return hodor_iop_create(ctx, leafs, n, nodes);
    NEED_DEVICE();
    if (combiner != HODOR_COMBINER_COSET2 || !leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 4) { set_err(ctx, "iop_create (COSET2): n must be a power of two >= 4"); return HODOR_ERR_SIZE; }
    return with_device_copy(ctx, leafs, n, nodes, n / 2, [&](const uint4 *s, uint4 *d) -> int {
        HIPCHK(merkle_build_launch(ctx->stream, s, d, n, ctx->mid, 1, nullptr, nullptr, true));
        return HODOR_OK;
    }, true);
}

extern "C" int hodor_iop_create(hodor_ctx *ctx, const hodor_fr *leafs, size_t n, uint8_t *nodes)
{
    NEED_DEVICE();
    if (!leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) { set_err(ctx, "iop_create: n must be a power of two >= 2"); return HODOR_ERR_SIZE; }
    return with_device_copy(ctx, leafs, n, nodes, n, [&](const uint4 *s, uint4 *d) -> int {
        HIPCHK(merkle_build_launch(ctx->stream, s, d, n, ctx->mid));
        return HODOR_OK;
    }, true);
}
