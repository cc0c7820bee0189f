// abi.hip — implementation of include/hodor_gpu.h: context, twiddle cache, NTT planning, and the
// C entry points that stand where the reference's L2/L3 Rust functions stand.  No CPU fallback: a
// context without a device refuses every compute call with HODOR_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/hodor_gpu.h"
#include "host_blake2s.hpp"
#include "host_field.hpp"
#include "ntt.cuh"

namespace hodor {

// kernels' host launchers (ntt.hip, pointwise.hip, merkle.hip, fri.hip)
hipError_t ntt_launch_pass(hipStream_t, const PassArgs &, const Fr9 *scale, const Fr9Params &);
hipError_t pow_table_launch(hipStream_t, uint4 *out, const Fr &base, const Fr &mult,
                            uint32_t log_stride, uint64_t count, uint32_t fmt, const FrParams &);
hipError_t distribute_powers_launch(hipStream_t, uint4 *a, uint64_t n, const TwoLevel &t, const Fr9Params &);
hipError_t distribute_powers_small_launch(hipStream_t, uint4 *a, uint64_t n, const Fr &g, const FrParams &);
hipError_t scale_launch(hipStream_t, uint4 *a, uint64_t n, const Fr &f, const FrParams &);
hipError_t binary_launch(hipStream_t, uint4 *a, const uint4 *b, uint64_t n, int op, const FrParams &);
hipError_t add_scaled_launch(hipStream_t, uint4 *a, const uint4 *b, uint64_t n, const Fr &f, const FrParams &);
hipError_t unary_launch(hipStream_t, uint4 *a, uint64_t n, int op, const Fr &c, uint64_t e, const FrParams &);
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
                               uint32_t batch = 1, const FoldArgs *fold = nullptr, const Fr9Params *Q = nullptr);
bool merkle_fuses_fold(uint64_t n);
hipError_t iop_query_launch(hipStream_t, const uint4 *leaf_pair, const uint4 *nodes, uint64_t n,
                            uint64_t index, uint4 *out, const B2Mid &);
hipError_t challenge_launch(hipStream_t, const uint4 *nodes, uint4 *out, uint4 *root_out, const Fr &r2,
                            uint32_t shave_bits, const FrParams &);
hipError_t fri_round_table_launch(hipStream_t, const uint4 *nodes, uint4 *chal_out, uint4 *root_out,
                                  const uint4 *hi, uint4 *hi_out, uint64_t count, const Fr9 &c16, const Fr &r2,
                                  uint32_t shave, const Fr9Params &, const FrParams &);
hipError_t fri_tail_launch(hipStream_t, const FriTailArgs &, const Fr9 &c16, const Fr &r2, const B2Mid &,
                           const Fr9Params &, const FrParams &);
hipError_t fri_fold_launch(hipStream_t, const FoldArgs &, const Fr9Params &);

static Fr to_dev(const HFr &a)
{
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)a.l[i];
        r.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    }
    return r;
}

// split a 256-bit integer (4 x u64) into 9 limbs of 29 bits
static void split29(const uint64_t l[4], uint32_t out[9])
{
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, w = bit >> 6, s = bit & 63;
        uint64_t v = l[w] >> s;
        if (s > 35 && w < 3) v |= l[w + 1] << (64 - s);
        out[i] = (uint32_t)(v & 0x1fffffffu);
    }
}

// R-form host element (x * 2^256) -> R'-form 9 x 29-bit multiplier operand (x * 2^261 mod p)
static Fr9 to_dev9(const HostField &F, const HFr &a)
{
    HFr t = a;
    for (int i = 0; i < 5; i++) t = F.add(t, t);
    Fr9 r;
    split29(t.l, r.v);
    return r;
}

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

struct PowTable {
    HFr base;
    uint32_t log_n;
    uint32_t lo_bits;
    uint32_t fmt;          // 0: 32-byte R-form entries, 1: 48-byte 9 x 29-bit R'-form entries
    HFr hi_mult;           // every `hi` entry is multiplied by this (one, or n^-1 for the last iNTT pass)
    uint4 *lo, *hi;
};

struct RadixTable {
    HFr omega;
    uint32_t log_n, log_r;
    uint4 *rtw;
};

}  // namespace hodor

using namespace hodor;

struct hodor_ctx {
    int device = -1;
    HostField F;
    FrParams P;
    Fr9Params Q;
    B2Mid mid;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::vector<PowTable> pow_tables;
    std::vector<RadixTable> radix_tables;
    void *scratch[2] = {nullptr, nullptr};
    size_t scratch_bytes[2] = {0, 0};
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
    static constexpr int IO_LANES = 3;
    IoLane lanes[IO_LANES];
    std::mutex lane_mu;
    std::condition_variable lane_cv;
    void *fri_slab = nullptr;  // parked FRI prototype slab (see hodor_fri_free)
    size_t fri_slab_bytes = 0;
    uint32_t max_log_r = 9;    // largest per-pass radix (2^max_log_r points)      } measured best on MI355X
    uint32_t tile_log = 10;    // elements per workgroup tile = 2^tile_log         } (bench/size_sweep.sh)
    std::string err;
};

struct hodor_fri_proto {
    hodor_ctx *ctx;
    size_t n, num_steps, lde_factor, out_deg, initial_degree_plus_one;
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
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e__);            \
            return HODOR_ERR_DEVICE;                                                  \
        }                                                                             \
    } while (0)

#define NEED_DEVICE()                                                                 \
    do {                                                                              \
        if (!ctx) return HODOR_ERR_INVALID;                                           \
        if (ctx->device < 0) { ctx->err = "context has no HIP device"; return HODOR_ERR_DEVICE; } \
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
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------------
static int free_tables(hodor_ctx *ctx)
{
    HIPCHK(hipDeviceSynchronize());
    for (auto &t : ctx->pow_tables) { (void)hipFree(t.lo); (void)hipFree(t.hi); }
    for (auto &t : ctx->radix_tables) (void)hipFree(t.rtw);
    ctx->pow_tables.clear();
    ctx->radix_tables.clear();
    return HODOR_OK;
}

// Cache eviction happens only here, at the start of an operation and before it has looked up any
// table: evicting later would free tables the same operation already holds pointers to.
static int trim_table_cache(hodor_ctx *ctx)
{
    if (ctx->pow_tables.size() > 40 || ctx->radix_tables.size() > 80) return free_tables(ctx);
    return HODOR_OK;
}

// base^e = lo[e & mask] * hi[e >> lo_bits] for e < 2^log_n
// fmt 1 = 9 x 29-bit R'-form entries for k_ntt_pass, fmt 0 = 32-byte R-form entries (fold, twiddle_mul)
static int get_pow_table(hodor_ctx *ctx, const HFr &base, uint32_t log_n, TwoLevel *out, uint32_t fmt,
                         uint32_t lo_bits = 0xffffffffu, const HFr *hi_mult_p = nullptr)
{
    if (lo_bits > log_n) lo_bits = (log_n + 1) / 2;
    const HFr hi_mult = hi_mult_p ? *hi_mult_p : ctx->F.one;
    for (auto &t : ctx->pow_tables)
        if (t.log_n == log_n && t.lo_bits == lo_bits && t.fmt == fmt && t.base == base && t.hi_mult == hi_mult) {
            *out = TwoLevel{t.lo, t.hi, t.lo_bits};
            return HODOR_OK;
        }
    PowTable t;
    t.base = base;
    t.log_n = log_n;
    t.lo_bits = lo_bits;
    t.fmt = fmt;
    t.hi_mult = hi_mult;
    const size_t esz = fmt ? 48 : 32;
    uint64_t lo_cnt = 1ull << t.lo_bits, hi_cnt = 1ull << (log_n - t.lo_bits);
    HIPCHK(hipMalloc((void **)&t.lo, lo_cnt * esz));
    HIPCHK(hipMalloc((void **)&t.hi, hi_cnt * esz));
    Fr b = to_dev(base), one = to_dev(ctx->F.one);
    HIPCHK(pow_table_launch(ctx->stream, t.lo, b, one, 0, lo_cnt, fmt, ctx->P));
    HIPCHK(pow_table_launch(ctx->stream, t.hi, b, to_dev(hi_mult), t.lo_bits, hi_cnt, fmt, ctx->P));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // tables are shared across streams afterwards
    ctx->pow_tables.push_back(t);
    *out = TwoLevel{t.lo, t.hi, t.lo_bits};
    return HODOR_OK;
}

// omega_R^e = omega^(e << (log_n - log_r)), e < R/2
static int get_radix_table(hodor_ctx *ctx, const HFr &omega, uint32_t log_n, uint32_t log_r,
                           const uint4 **out)
{
    for (auto &t : ctx->radix_tables)
        if (t.log_n == log_n && t.log_r == log_r && t.omega == omega) {
            *out = t.rtw;
            return HODOR_OK;
        }
    RadixTable t;
    t.omega = omega;
    t.log_n = log_n;
    t.log_r = log_r;
    uint64_t cnt = log_r ? (1ull << (log_r - 1)) : 1;
    HIPCHK(hipMalloc((void **)&t.rtw, cnt * 48));
    HIPCHK(pow_table_launch(ctx->stream, t.rtw, to_dev(omega), to_dev(ctx->F.one), log_n - log_r, cnt, 1,
                            ctx->P));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->radix_tables.push_back(t);
    *out = t.rtw;
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
        if ((e = hipFree(L->buf[which])) != hipSuccess) return e;
        L->buf[which] = nullptr;
        L->bytes[which] = 0;
    }
    if ((e = hipMalloc(&L->buf[which], bytes)) != hipSuccess) return e;
    L->bytes[which] = bytes;
    return hipSuccess;
}

static int ensure_scratch(hodor_ctx *ctx, int which, size_t bytes)
{
    if (ctx->scratch_bytes[which] >= bytes) return HODOR_OK;
    if (ctx->scratch[which]) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(ctx->scratch[which]));
        ctx->scratch[which] = nullptr;
        ctx->scratch_bytes[which] = 0;
    }
    HIPCHK(hipMalloc(&ctx->scratch[which], bytes));
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
static int ntt_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst, uint32_t log_n,
                    const HFr &omega, uint64_t nnz, const HFr *scale, const HFr *pre, const HFr *post,
                    uint32_t batch = 1)
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
        if (log_n - shift2 <= 17 && shift2 <= 16) tw_lo_bits = shift2;
    }
    if (passes > 1 && (rc = get_pow_table(ctx, omega, log_n, &tw, 1, tw_lo_bits))) return rc;
    // iNTT: fold the n^-1 scale into the `hi` half of the LAST pass's twiddle table — every element of
    // that pass is multiplied by a twiddle anyway (tw_always covers the exponent-0 ones), so the scale
    // costs no product of its own (it is a separate streaming pass on the CPU, src/polynomials/mod.rs:777-787)
    TwoLevel tw_last = tw;
    const bool fold_scale = scale && passes > 1;
    if (fold_scale && (rc = get_pow_table(ctx, omega, log_n, &tw_last, 1, tw_lo_bits, scale))) return rc;
    if (pre && (rc = get_pow_table(ctx, *pre, log_n, &pre_t, 1))) return rc;
    if (post && (rc = get_pow_table(ctx, *post, log_n, &post_t, 1))) return rc;

    // ping-pong buffers: the last pass writes dst; a pass never runs in place unless it is the only
    // one (a single tile is fully staged in LDS before anything is written back).
    std::vector<uint4 *> outs(passes);
    if (passes == 1) {
        outs[0] = dst;
    } else {
        bool in_place = (src == dst);
        if ((rc = ensure_scratch(ctx, 0, bytes))) return rc;
        uint4 *s0 = (uint4 *)ctx->scratch[0], *s1 = nullptr;
        // walk backwards: pass P-1 -> dst, P-2 -> s0, P-3 -> dst (or s1 if that would clobber src)...
        for (size_t i = 0; i < passes; i++) {
            size_t from_end = passes - 1 - i;
            outs[i] = (from_end % 2 == 0) ? dst : s0;
        }
        if (in_place && outs[0] == dst) {
            // pass 0 would overwrite its own (strided) input: route passes 0 and 1 through s1/s0
            if ((rc = ensure_scratch(ctx, 1, bytes))) return rc;
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
        PassArgs A;
        A.src = cur;
        A.dst = outs[i];
        if ((rc = get_radix_table(ctx, omega, log_n, log_r, &A.rtw))) return rc;
        A.tw = (i + 1 == passes) ? tw_last : tw;
        A.tw_always = (fold_scale && i + 1 == passes) ? 1 : 0;
        A.pre = (i == 0) ? pre_t : TwoLevel{nullptr, nullptr, 0};
        A.post = (i + 1 == passes) ? post_t : TwoLevel{nullptr, nullptr, 0};
        A.nnz = (i == 0) ? nnz : (1ull << log_n);
        A.log_n = log_n;
        A.log_r = log_r;
        uint32_t log_c = ctx->tile_log > log_r ? ctx->tile_log - log_r : 0;
        if (log_c > log_n - log_r) log_c = log_n - log_r;
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
        HIPCHK(ntt_launch_pass(stream, A, (scale && !fold_scale && i + 1 == passes) ? &scale_d : nullptr, ctx->Q));
        cur = outs[i];
        log_l += log_r;
    }
    return HODOR_OK;
}

static int poly_domain(hodor_ctx *ctx, uint32_t log_n, HFr *omega)
{
    uint64_t sz;
    uint32_t k;
    if (log_n > 63 || !ctx->F.domain(1ull << log_n, &sz, &k, omega)) {
        ctx->err = "domain too large for the field's 2-adicity";
        return HODOR_ERR_SIZE;
    }
    return HODOR_OK;
}

enum PolyOp { OP_FFT, OP_COSET_FFT, OP_IFFT, OP_ICOSET_FFT };

static int poly_transform(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst,
                          uint32_t log_n, PolyOp op)
{
    HFr omega;
    int rc = poly_domain(ctx, log_n, &omega);
    if (rc) return rc;
    uint64_t n = 1ull << log_n;
    switch (op) {
    case OP_FFT:
        return ntt_exec(ctx, stream, src, dst, log_n, omega, n, nullptr, nullptr, nullptr);
    case OP_COSET_FFT:   // distribute_powers(g) then fft — src/polynomials/mod.rs:626-631
        return ntt_exec(ctx, stream, src, dst, log_n, omega, n, nullptr, &ctx->F.generator, nullptr);
    case OP_IFFT:
    case OP_ICOSET_FFT: {   // best_fft(omegainv) then * minv (then * geninv^i) — :773-807
        HFr oinv, minv, ginv;
        ctx->F.inverse(omega, &oinv);
        ctx->F.inverse(ctx->F.from_u64(n), &minv);
        ctx->F.inverse(ctx->F.generator, &ginv);
        return ntt_exec(ctx, stream, src, dst, log_n, oinv, n, &minv, nullptr,
                        op == OP_ICOSET_FFT ? &ginv : nullptr);
    }
    }
    return HODOR_ERR_INVALID;
}

// lde / coset_lde: one zero-padded transform of size n*factor — identical output to the
// reference's per-coset schedule (asserted by its own tests, src/polynomials/mod.rs:1026-1031)
static int poly_lde_exec(hodor_ctx *ctx, hipStream_t stream, const uint4 *src, uint4 *dst,
                         uint32_t log_n, size_t factor, int coset, uint32_t batch = 1)
{
    if (!is_pow2(factor)) { ctx->err = "lde factor must be a power of two"; return HODOR_ERR_SIZE; }
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
    // BASE_BLAKE2S_PARAMS, src/iop/blake2s_trivial_iop.rs:8-16
    HostBlake2s::keyed_midstate(ctx->mid.h, (const uint8_t *)"Squeamish Ossifrage", 19,
                                (const uint8_t *)"Shaftoe", 7);
    ctx->device = device;
    if (device >= 0) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device >= count ||
            hipSetDevice(device) != hipSuccess ||
            hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return HODOR_ERR_DEVICE;
        }
        const char *e = getenv("HODOR_MAX_LOG_R");
        if (e && atoi(e) >= 2 && atoi(e) <= 11) ctx->max_log_r = (uint32_t)atoi(e);
        e = getenv("HODOR_TILE_LOG");
        if (e && atoi(e) >= 6 && atoi(e) <= 12) ctx->tile_log = (uint32_t)atoi(e);
        if (ctx->max_log_r > ctx->tile_log) ctx->max_log_r = ctx->tile_log;
    }
    *out = ctx;
    return HODOR_OK;
}

extern "C" void hodor_ctx_destroy(hodor_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        for (auto &t : ctx->pow_tables) { (void)hipFree(t.lo); (void)hipFree(t.hi); }
        for (auto &t : ctx->radix_tables) (void)hipFree(t.rtw);
        for (int i = 0; i < 2; i++)
            if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
        if (ctx->fri_slab) (void)hipFree(ctx->fri_slab);
        for (auto &L : ctx->lanes) {
            for (int i = 0; i < 2; i++)
                if (L.buf[i]) (void)hipFree(L.buf[i]);
            if (L.uploaded) (void)hipEventDestroy(L.uploaded);
            if (L.computed) (void)hipEventDestroy(L.computed);
            if (L.stream) (void)hipStreamDestroy(L.stream);
        }
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
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

extern "C" const char *hodor_last_error(const hodor_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

extern "C" int hodor_ctx_synchronize(hodor_ctx *ctx)
{
    NEED_DEVICE();
    HIPCHK(hipDeviceSynchronize());
    return HODOR_OK;
}

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

// ------------------------------------------------------------------------------------------------
// device API
// ------------------------------------------------------------------------------------------------
extern "C" int hodor_buf_alloc(hodor_ctx *ctx, size_t bytes, void **dev_ptr)
{
    NEED_DEVICE();
    if (!dev_ptr) return HODOR_ERR_INVALID;
    HIPCHK(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return HODOR_OK;
}
extern "C" int hodor_buf_free(hodor_ctx *ctx, void *dev_ptr)
{
    NEED_DEVICE();
    HIPCHK(hipFree(dev_ptr));
    return HODOR_OK;
}
extern "C" int hodor_buf_upload(hodor_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    NEED_DEVICE();
    HIPCHK(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return HODOR_OK;
}
extern "C" int hodor_buf_download(hodor_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    NEED_DEVICE();
    HIPCHK(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return HODOR_OK;
}

static inline hipStream_t pick_stream(hodor_ctx *ctx, void *stream)
{
    (void)ctx;
    return (hipStream_t)stream;   // NULL selects the HIP default (null) stream, as for any HIP API
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
                    PolyOp op)
{
    NEED_DEVICE();
    if (!src || !dst) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_transform(ctx, pick_stream(ctx, stream), (const uint4 *)src, (uint4 *)dst, log_n, op);
}
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

extern "C" int hodor_precomputed_omegas_dev(hodor_ctx *ctx, void *stream, uint32_t log_n, hodor_fr *omegas,
                                            hodor_fr *coset, hodor_fr *omegas_inv)
{
    NEED_DEVICE();
    uint64_t size;
    uint32_t lg;
    HFr w, winv;
    if (log_n > 40 || !ctx->F.domain(1ull << log_n, &size, &lg, &w)) {
        ctx->err = "domain larger than the field's two-adicity";
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
    static int seq = -1;   // elements per thread and level
    if (seq < 0) {
        const char *e = getenv("HODOR_BATCHINV_SEQ");
        seq = e ? atoi(e) : 8;
        if (seq < 2) seq = 2;
    }
    for (uint64_t m = n; m > TOP || levels.empty();) {
        uint64_t T = (m + seq - 1) / seq;
        Level L = {m, T, off, 0};
        off += up(m * 32);
        L.prod_off = off;
        off += up(T * 32);
        levels.push_back(L);
        m = T;
    }
    int rc = ensure_scratch(ctx, 0, off + 256);
    if (rc) return rc;
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
    HIPCHK(hipMemcpyAsync(top_prod, base + top.prod_off, top.T * 32, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&host_flag, flag, 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (host_flag) {   // full_grand_product.inverse() is None -> SynthesisError::Error, data untouched (:909)
        ctx->err = "batch_inversion: zero element";
        return HODOR_ERR_INVALID;
    }
    for (uint64_t i = 0; i < top.T; i++) {
        HFr inv;
        if (!ctx->F.inverse(to_h(&top_prod[i]), &inv)) { ctx->err = "batch_inversion: zero product"; return HODOR_ERR_INVALID; }
        from_h(inv, &top_prod[i]);
    }
    HIPCHK(hipMemcpyAsync(base + top.prod_off, top_prod, top.T * 32, hipMemcpyHostToDevice, stream));
    for (size_t l = levels.size(); l-- > 0;) {
        const Level &L = levels[l];
        uint4 *target = l == 0 ? (uint4 *)a : (uint4 *)(base + levels[l - 1].prod_off);
        HIPCHK(batchinv_backward_launch(stream, target, L.n, L.T, (const uint4 *)(base + L.prefix_off),
                                        (const uint4 *)(base + L.prod_off), ctx->P));
    }
    HIPCHK(hipStreamSynchronize(stream));   // top_prod is a stack buffer the upload reads from
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
    int rc = ensure_scratch(ctx, 0, 32 * (blocks + 2) + 64);
    if (rc) return rc;
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
    HIPCHK(hipMemcpyAsync(out, res, 32, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return HODOR_OK;
}

extern "C" int hodor_iop_create_dev(hodor_ctx *ctx, void *stream, const hodor_fr *leafs, size_t n,
                                    uint8_t *nodes)
{
    NEED_DEVICE();
    if (!leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) { ctx->err = "iop_create: n must be a power of two >= 2"; return HODOR_ERR_SIZE; }
    HIPCHK(merkle_build_launch(pick_stream(ctx, stream), (const uint4 *)leafs, (uint4 *)nodes, n, ctx->mid));
    return HODOR_OK;
}

// The prototype's device buffers (l0 tree, every intermediate vector and tree, the small result block)
// are carved from ONE slab; a freed slab is parked on the context and reused by the next commit of a
// size that fits, so a prover committing polynomial after polynomial pays hipMalloc once.
extern "C" void hodor_fri_free(hodor_fri_proto *p)
{
    if (!p) return;
    hodor_ctx *ctx = p->ctx;
    if (ctx && ctx->device >= 0 && p->slab) {
        (void)hipSetDevice(ctx->device);
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->fri_slab) {
            ctx->fri_slab = p->slab;
            ctx->fri_slab_bytes = p->slab_bytes;
        } else if (p->slab_bytes > ctx->fri_slab_bytes) {
            (void)hipFree(ctx->fri_slab);
            ctx->fri_slab = p->slab;
            ctx->fri_slab_bytes = p->slab_bytes;
        } else {
            (void)hipFree(p->slab);
        }
    }
    delete p;
}

extern "C" int hodor_fri_commit_dev(hodor_ctx *ctx, void *stream_, const hodor_fr *lde_values, size_t n,
                                    size_t lde_factor, size_t out_deg, hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || !is_pow2(lde_factor) || !is_pow2(out_deg) || n < 2) return HODOR_ERR_SIZE;
    size_t initial_degree_plus_one = n / lde_factor;
    if (initial_degree_plus_one < 2 * out_deg) {   // num_steps == 0: the reference panics at roots.pop() (:124)
        ctx->err = "fri_commit: needs at least one folding step";
        return HODOR_ERR_SIZE;
    }
    size_t num_steps = log2u(initial_degree_plus_one / out_deg);
    if ((n >> num_steps) < 2) return HODOR_ERR_SIZE;
    uint32_t log_n = log2u(n);
    HFr omega, omega_inv;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = poly_domain(ctx, log_n, &omega);
    if (rc) return rc;
    ctx->F.inverse(omega, &omega_inv);
    hipStream_t stream = pick_stream(ctx, stream_);

    hodor_fri_proto *p = new (std::nothrow) hodor_fri_proto();
    if (!p) return HODOR_ERR_INVALID;
    p->ctx = ctx;
    p->n = n;
    p->num_steps = num_steps;
    p->lde_factor = lde_factor;
    p->out_deg = out_deg;
    p->initial_degree_plus_one = initial_degree_plus_one;

#define FRICHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e__);             \
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
        if (q->slab) (void)hipFree(q->slab);
        delete q;
    };
    if (ctx->fri_slab && ctx->fri_slab_bytes >= need) {
        p->slab = ctx->fri_slab;
        p->slab_bytes = ctx->fri_slab_bytes;
        ctx->fri_slab = nullptr;
        ctx->fri_slab_bytes = 0;
    } else {
        FRICHK(hipMalloc(&p->slab, need));
        p->slab_bytes = need;
    }
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

    FRICHK(merkle_build_launch(stream, (const uint4 *)lde_values, (uint4 *)p->l0_nodes, n, ctx->mid));   // :17

    const uint4 *values = (const uint4 *)lde_values;
    size_t next_size = n / 2;
    static int tail_on = -1;   // HODOR_FRI_TAIL=0: every round through the multi-launch path (A/B, debugging)
    if (tail_on < 0) {
        const char *e = getenv("HODOR_FRI_TAIL");
        tail_on = e ? atoi(e) : 1;
    }
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
            FRICHK(fri_tail_launch(stream, T, c16, r2, ctx->mid, ctx->Q, ctx->P));
            values = (const uint4 *)p->inter_values[num_steps - 1];
            tail_done = true;
            break;
        }
        void *next = p->inter_values[i], *nodes = p->inter_nodes[i];
        FRICHK(fri_round_table_launch(stream, prev_nodes, d_chal + 2 * i, d_roots + 2 * i, winv.hi, d_hi_beta, hi_cnt,
                                      c16, r2, shave, ctx->Q, ctx->P));
        FoldArgs fold = {values, (uint4 *)next, next_size, winv.lo, d_hi_beta, winv.lo_bits, (uint32_t)i};
        const bool fused = merkle_fuses_fold(next_size);   // small rounds: fold inside the tree's leaf launch
        if (!fused) FRICHK(fri_fold_launch(stream, fold, ctx->Q));                                        // :70-104
        FRICHK(merkle_build_launch(stream, (const uint4 *)next, (uint4 *)nodes, next_size, ctx->mid, 1,
                                   fused ? &fold : nullptr, &ctx->Q));                                   // :106
        prev_nodes = (const uint4 *)nodes;
        values = (const uint4 *)next;
        next_size >>= 1;
    }
    if (!tail_done)   // the last tree's root (and the challenge the reference computes and pops, :120)
        FRICHK(challenge_launch(stream, prev_nodes, d_chal + 2 * num_steps, d_roots + 2 * num_steps, r2, shave, ctx->P));
    // final: values -> ifft -> truncate (:130-145)
    rc = poly_transform(ctx, stream, values, d_fin, log2u(fin_n), OP_IFFT);
    if (rc) { fri_release(p); return rc; }

    // challenges | roots | final coefficients sit back to back in the slab's small block: one copy
    std::vector<uint8_t> small(64 * (num_steps + 1) + 32 * out_deg);
    FRICHK(hipMemcpyAsync(small.data(), d_small, small.size(), hipMemcpyDeviceToHost, stream));
    FRICHK(hipStreamSynchronize(stream));
    p->roots.assign(small.begin() + 32 * (num_steps + 1), small.begin() + 64 * (num_steps + 1));
    p->challenges.resize(num_steps);
    memcpy(p->challenges.data(), small.data(), 32 * num_steps);
    p->final_coeffs.resize(out_deg);
    memcpy(p->final_coeffs.data(), small.data() + 64 * (num_steps + 1), 32 * out_deg);
    memcpy(p->final_root, p->roots.data() + 32 * num_steps, 32);   // roots.pop() :124
#undef FRICHK
    *out = p;
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
template <class Op>
static int with_device_copy(hodor_ctx *ctx, const void *in, size_t n_in, void *out, size_t n_out, Op op,
                            bool separate_out = false)
{
    hodor_ctx::IoLane *L = lane_acquire(ctx);
    auto fail = [&](hipError_t e, const char *what) {
        (void)hipStreamSynchronize(L->stream);
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        }
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
    if ((e = hipMemcpyAsync(din, in, n_in * 32, hipMemcpyHostToDevice, L->stream)) != hipSuccess ||
        (e = hipEventRecord(L->uploaded, L->stream)) != hipSuccess)
        return fail(e, "slice upload");
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        e = hipStreamWaitEvent(ctx->stream, L->uploaded, 0);
        rc = e == hipSuccess ? op((const uint4 *)din, (uint4 *)dptr_out) : HODOR_ERR_DEVICE;
        if (e == hipSuccess) e = hipEventRecord(L->computed, ctx->stream);
        if (rc || e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);
    }
    if (rc) { (void)hipStreamSynchronize(L->stream); lane_release(ctx, L); return rc; }
    if (e != hipSuccess) return fail(e, "slice compute");
    if ((e = hipStreamWaitEvent(L->stream, L->computed, 0)) != hipSuccess ||
        (e = hipMemcpyAsync(out, dptr_out, n_out * 32, hipMemcpyDeviceToHost, L->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(L->stream)) != hipSuccess)
        return fail(e, "slice download");
    lane_release(ctx, L);
    return HODOR_OK;
}

extern "C" int hodor_fft(hodor_ctx *ctx, hodor_fr *a, size_t n, const hodor_fr *omega, uint32_t log_n)
{
    NEED_DEVICE();
    if (!a || !omega) return HODOR_ERR_INVALID;
    if (log_n > 40 || n != ((size_t)1 << log_n)) { ctx->err = "fft: n != 1 << log_n"; return HODOR_ERR_SIZE; }   // assert_eq at src/fft/fft.rs:34
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

static int poly_slice(hodor_ctx *ctx, hodor_fr *a, size_t n, PolyOp op)
{
    NEED_DEVICE();
    if (!a) return HODOR_ERR_INVALID;
    if (!is_pow2(n)) { ctx->err = "polynomial size must be a power of two"; return HODOR_ERR_SIZE; }
    uint32_t log_n = log2u(n);
    return with_device_copy(ctx, a, n, a, n, [&](const uint4 *s, uint4 *d) {
        return poly_transform(ctx, ctx->stream, s, d, log_n, op);
    });
}
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

extern "C" int hodor_iop_create(hodor_ctx *ctx, const hodor_fr *leafs, size_t n, uint8_t *nodes)
{
    NEED_DEVICE();
    if (!leafs || !nodes) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) { ctx->err = "iop_create: n must be a power of two >= 2"; return HODOR_ERR_SIZE; }
    return with_device_copy(ctx, leafs, n, nodes, n, [&](const uint4 *s, uint4 *d) -> int {
        HIPCHK(merkle_build_launch(ctx->stream, s, d, n, ctx->mid));
        return HODOR_OK;
    }, true);
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

extern "C" int hodor_fri_commit(hodor_ctx *ctx, const hodor_fr *lde_values, size_t n, size_t lde_factor,
                                size_t out_deg, hodor_fri_proto **out)
{
    NEED_DEVICE();
    if (!lde_values || !out) return HODOR_ERR_INVALID;
    if (!is_pow2(n) || n < 2) return HODOR_ERR_SIZE;
    DevBuf dv;
    HIPCHK(hipMalloc(&dv.p, n * 32));
    HIPCHK(hipMemcpy(dv.p, lde_values, n * 32, hipMemcpyHostToDevice));
    return hodor_fri_commit_dev(ctx, nullptr, (const hodor_fr *)dv.p, n, lde_factor, out_deg, out);
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
    DevBuf stage;
    HIPCHK(hipMalloc(&stage.p, entries * 32));
    HIPCHK(iop_query_launch(stream, (const uint4 *)(leafs + (natural_index & ~(size_t)1)), (const uint4 *)nodes, n,
                            natural_index, (uint4 *)stage.p, ctx->mid));
    std::vector<uint8_t> host(entries * 32);
    HIPCHK(hipMemcpyAsync(host.data(), stage.p, entries * 32, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    memcpy(value, host.data(), 32);
    memcpy(path, host.data() + 32, (entries - 1) * 32);
    *path_len = entries - 1;
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
    size_t need = 8, stage_bytes = 0;
    for (size_t r = 0, sz = p->n; r < rounds; r++, sz >>= 1) {
        size_t entries = log2u(sz) + 1;
        need += 2 * (8 + 32 + 8 + (entries - 1) * 32);
        stage_bytes += 2 * entries * 32;
    }
    need += 8 + rounds * 32 + 8 + p->out_deg * 32 + 24;
    if (!buf || cap < need) return need;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DevBuf stage;
    if (hipMalloc(&stage.p, stage_bytes) != hipSuccess) return 0;
    std::vector<size_t> q_index, q_entries;
    size_t domain_size = p->n, domain_idx = natural_first_element_index, off = 0;
    for (size_t r = 0; r < rounds; r++) {
        const hodor_fr *leafs = r == 0 ? lde_values_dev : (const hodor_fr *)p->inter_values[r - 1];
        const uint4 *nodes = (const uint4 *)(r == 0 ? p->l0_nodes : p->inter_nodes[r - 1]);
        size_t pair = (domain_idx + domain_size / 2) % domain_size;
        size_t coset[2] = {domain_idx < pair ? domain_idx : pair, domain_idx < pair ? pair : domain_idx};
        size_t entries = log2u(domain_size) + 1;
        for (int k = 0; k < 2; k++) {
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
    if (hipMemcpyAsync(host.data(), stage.p, stage_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return 0;
    size_t o = 0, h = 0;
    auto put64 = [&](uint64_t v) { memcpy(buf + o, &v, 8); o += 8; };
    put64(q_index.size());
    for (size_t q = 0; q < q_index.size(); q++) {
        put64(q_index[q]);
        memcpy(buf + o, host.data() + h, 32); o += 32;
        put64(q_entries[q] - 1);
        memcpy(buf + o, host.data() + h + 32, (q_entries[q] - 1) * 32); o += (q_entries[q] - 1) * 32;
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
    if (!ctx || !proof || !expected_value_from_oracle || !valid) return HODOR_ERR_INVALID;
    *valid = 0;
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
        ctx->err = "initial challenge value is not in the LDE domain";
        return HODOR_ERR_INVALID;
    }
    if (!F.inverse(omega, &omega_inv)) return HODOR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    bool have_expected = false;
    HFr expected = F.one;
    uint64_t domain_size = size, domain_idx = natural_element_index;
    auto fetch = [&](const hodor_fr *base, uint64_t i, HFr *out) -> bool {
        hodor_fr v;
        if (hipMemcpy(&v, base + i, 32, hipMemcpyDeviceToHost) != hipSuccess) return false;
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
            ctx->err = "verify_prototype: device read failed";
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

extern "C" size_t hodor_fri_num_steps(const hodor_fri_proto *p) { return p ? p->num_steps : 0; }

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
    HIPCHK(hipMemcpy(values, p->inter_values[step], p->inter_sizes[step] * 32, hipMemcpyDeviceToHost));
    return HODOR_OK;
}
extern "C" int hodor_fri_tree_nodes(hodor_fri_proto *p, int step, uint8_t *nodes)
{
    if (!p || !nodes || step < -1 || step >= (int)p->num_steps) return HODOR_ERR_INVALID;
    hodor_ctx *ctx = p->ctx;
    NEED_DEVICE();
    if (step < 0) HIPCHK(hipMemcpy(nodes, p->l0_nodes, p->n * 32, hipMemcpyDeviceToHost));
    else HIPCHK(hipMemcpy(nodes, p->inter_nodes[step], p->inter_sizes[step] * 32, hipMemcpyDeviceToHost));
    return HODOR_OK;
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
