// abi_selftest.hip — the start-up self-test of a context (hodor_ctx_create, round 6).
//
// Why: parts of the device code lean on properties no type system checks — the store phase of k_ntt_pass re-reads its
// arguments through __builtin_amdgcn_kernarg_segment_ptr() behind an opaque asm (ntt.hip, "late arguments"), the W9
// constants arrive by scalar loads, the BLAKE2s rounds run on DPP lane permutations.  A compiler, code-object or driver
// change that breaks one of them would so far have been noticed by the test suite only.  Every context therefore checks,
// before it is handed out, the kernels it is about to use against the HOST implementations of the same arithmetic
// (host_field.hpp, host_blake2s.hpp — plain C++, nothing shared with the device code but the field constants):
//
//   1. one 2^10-point transform per k_ntt_pass instantiation the context will use — the plain one (MODE 0) through
//      hodor_fft_dev, the general one (MODE 1: column mode, 2D twiddles) through the 4-step pair at P = 1; P1 / generic
//      follows the field — every output element against a host radix-2 transform;
//   2. a 2^16-point fft -> ifft round trip (multi-pass plan, W3 / W9 tables, n^-1 folded into the last pass), compared
//      with its input on the device;
//   3. one FRI commit of a 2^11-value vector: the l0 tree (leaf and node hashes sampled on every level against
//      HostBlake2s with a midstate derived afresh), the challenge from its root, the first fold (k_fri_fold or its fused
//      form, every element) and the second (inside k_fri_tail, every element), the final coefficient.
//
// A mismatch makes hodor_ctx_create return HODOR_ERR_DEVICE; hodor_last_error(NULL) names the check.  Cost: measured
// on MI355X in profiles/r06/selftest.txt.  HODOR_SELFTEST=0 skips it (a process that creates contexts in a loop);
// HODOR_SELFTEST_CORRUPT=k is a debugging aid for the test of the test (tests/test_gpu_selftest.py): 1 flips a bit of the
// 9 x 29 field parameters, 2 of the BLAKE2s midstate, 3 of the W9 table constants, 4 of the 8 x 32 field parameters —
// each before anything runs — and 5 flips one word of a cached twiddle table between the first transform and its repeat.
#include "ctx.hpp"

extern "C" int hodor_fft_dev(hodor_ctx *, void *, const hodor_fr *, hodor_fr *, uint32_t, const hodor_fr *);

namespace {

struct SelfTestFail { std::string what; };

uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// n values below 2^(NUM_BITS - 1) <= p: valid Montgomery images, no rejection loop
std::vector<hodor_fr> pattern(const HostField &F, size_t n, uint64_t seed)
{
    std::vector<hodor_fr> v(n);
    const uint64_t top = (1ull << (F.num_bits - 1 - 192)) - 1;
    for (auto &e : v) {
        for (int i = 0; i < 4; i++) e.l[i] = splitmix(seed);
        e.l[3] &= top;
    }
    return v;
}

// radix-2 DIT after bit reversal on the host (the shape of src/fft/fft.rs:21-66), HostField arithmetic
void host_ntt(const HostField &F, std::vector<HFr> &a, const HFr &omega)
{
    const size_t n = a.size();
    const uint32_t log_n = log2u(n);
    for (size_t k = 0; k < n; k++) {
        size_t r = 0;
        for (uint32_t b = 0; b < log_n; b++) r |= ((k >> b) & 1) << (log_n - 1 - b);
        if (k < r) std::swap(a[k], a[r]);
    }
    for (size_t m = 1; m < n; m <<= 1) {
        const HFr w_m = F.pow(omega, n / (2 * m));
        for (size_t k = 0; k < n; k += 2 * m) {
            HFr w = F.one;
            for (size_t j = 0; j < m; j++) {
                const HFr t = F.mul(a[k + j + m], w), u = a[k + j];
                a[k + j] = F.add(u, t);
                a[k + j + m] = F.sub(u, t);
                w = F.mul(w, w_m);
            }
        }
    }
}

#define ST_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) { (void)hipGetLastError(); throw SelfTestFail{std::string(#expr) + ": " + hipGetErrorString(e__)}; } \
    } while (0)
#define ST_RC(expr)                                                                               \
    do {                                                                                          \
        int rc__ = (expr);                                                                        \
        if (rc__) throw SelfTestFail{std::string(#expr) + " failed: " + hodor_last_error(ctx)};   \
    } while (0)

struct PoolTmp {   // a pooled device buffer for the duration of the test
    hodor_ctx *ctx;
    void *p = nullptr;
    size_t got = 0;
    PoolTmp(hodor_ctx *c, size_t bytes) : ctx(c)
    {
        if (pool_alloc(c, bytes, &p, &got)) throw SelfTestFail{"device pool allocation failed"};
    }
    ~PoolTmp() { pool_release(ctx, p, got); }
    hodor_fr *fr() const { return (hodor_fr *)p; }
};

void flip_param_bit(void *p) { *(uint32_t *)p ^= 4u; }

void check_transforms(hodor_ctx *ctx, int corrupt)
{
    const HostField &F = ctx->F;
    const uint32_t LOG = 10;
    const size_t n = (size_t)1 << LOG;
    uint64_t sz;
    uint32_t lg;
    HFr w;
    if (!F.domain(n, &sz, &lg, &w)) return;   // a field without a 2^10 domain uses no transform of that size
    const std::vector<hodor_fr> in = pattern(F, n, 0x53454C46);
    std::vector<HFr> exp(n);
    for (size_t i = 0; i < n; i++) exp[i] = to_h(&in[i]);
    host_ntt(F, exp, w);
    hodor_fr omega;
    from_h(w, &omega);
    PoolTmp src(ctx, n * 32), d0(ctx, n * 32), d1(ctx, n * 32), tmp(ctx, n * 32);
    std::vector<hodor_fr> plain(n), general(n);
    void *s = (void *)ctx->stream;
    auto run = [&] {
        HostXfer xfer(ctx, ctx->stream);
        ST_HIP(xfer.h2d(src.p, in.data(), n * 32));
        ST_RC(hodor_fft_dev(ctx, s, src.fr(), d0.fr(), LOG, &omega));                                        // k_ntt_pass<0, .>
        ST_RC(hodor_sixstep_columns_dev(ctx, s, src.fr(), tmp.fr(), LOG / 2, LOG - LOG / 2, 0, 0, &omega, 0, 0, 0));   // k_ntt_pass<1, .>
        ST_RC(hodor_sixstep_rows_dev(ctx, s, tmp.fr(), d1.fr(), LOG / 2, LOG - LOG / 2, 0, 0, &omega, 0, 0, 0));
        ST_HIP(xfer.d2h(plain.data(), d0.p, n * 32));
        ST_HIP(xfer.d2h(general.data(), d1.p, n * 32));
        ST_HIP(xfer.finish());
    };
    run();
    if (corrupt == 5) {   // one word of a twiddle table the transforms above built and cached
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            if (ctx->radix_tables.empty()) throw SelfTestFail{"HODOR_SELFTEST_CORRUPT=5: no cached table to corrupt"};
            uint32_t word = 0;
            uint8_t *victim = (uint8_t *)ctx->radix_tables[0].rtw + 112 + 8;   // entry 1 (omega_R^1) of the first radix table, word 2
            ST_HIP(hipMemcpy(&word, victim, 4, hipMemcpyDeviceToHost));
            word ^= 0x10u;
            ST_HIP(hipMemcpy(victim, &word, 4, hipMemcpyHostToDevice));
        }
        run();
    }
    const size_t N1 = (size_t)1 << (LOG / 2), N2 = n / N1;
    for (size_t k = 0; k < n; k++)
        if (memcmp(plain[k].l, exp[k].l, 32) != 0)
            throw SelfTestFail{"2^10-point transform (k_ntt_pass, plain) differs from the host transform at output " + std::to_string(k)};
    for (size_t k1 = 0; k1 < N1; k1++)
        for (size_t k2 = 0; k2 < N2; k2++)   // layout B at P = 1: b[k1][k2] = X[k1 + N1 k2]
            if (memcmp(general[k1 * N2 + k2].l, exp[k1 + N1 * k2].l, 32) != 0)
                throw SelfTestFail{"2^10-point 4-step transform (k_ntt_pass, general mode) differs from the host transform at output " +
                                   std::to_string(k1 + N1 * k2)};
}

void check_round_trip(hodor_ctx *ctx)
{
    const HostField &F = ctx->F;
    const uint32_t LOG = 16;
    const size_t n = (size_t)1 << LOG;
    uint64_t sz;
    uint32_t lg;
    HFr w;
    if (!F.domain(n, &sz, &lg, &w)) return;
    PoolTmp a(ctx, n * 32), b(ctx, n * 32), flag(ctx, 256);
    ST_RC(hodor_gen_elements_dev(ctx, (void *)ctx->stream, a.fr(), 0, n, 0x53454C46));
    ST_RC(hodor_poly_fft_dev(ctx, (void *)ctx->stream, a.fr(), b.fr(), LOG));
    ST_RC(hodor_poly_ifft_dev(ctx, (void *)ctx->stream, b.fr(), b.fr(), LOG));
    ST_HIP(hipMemsetAsync(flag.p, 0, 4, ctx->stream));
    ST_HIP(count_diff_launch(ctx->stream, (const uint4 *)a.p, (const uint4 *)b.p, n, (uint32_t *)flag.p));
    uint32_t diff = 1;
    {
        HostXfer xfer(ctx, ctx->stream);
        ST_HIP(xfer.d2h(&diff, flag.p, 4));
        ST_HIP(xfer.finish());
    }
    if (diff) throw SelfTestFail{"ifft(fft(x)) != x on 2^16 points (multi-pass plan)"};
}

void check_commit(hodor_ctx *ctx)
{
    const HostField &F = ctx->F;
    const uint32_t LOG = 11;
    const size_t n = (size_t)1 << LOG;
    uint64_t sz;
    uint32_t lg;
    HFr w;
    if (!F.domain(n, &sz, &lg, &w)) return;
    const std::vector<hodor_fr> v = pattern(F, n, 0x465249);
    PoolTmp dv(ctx, n * 32);
    {
        HostXfer xfer(ctx, ctx->stream);
        ST_HIP(xfer.h2d(dv.p, v.data(), n * 32));
        ST_HIP(xfer.finish());
    }
    hodor_fri_proto *p = nullptr;
    ST_RC(hodor_fri_commit_combined_dev(ctx, (void *)ctx->stream, dv.fr(), n, 2, 1, HODOR_COMBINER_TRIVIAL, &p));
    struct Free { hodor_fri_proto *p; ~Free() { hodor_fri_free(p); } } guard{p};
    if (p->num_steps != LOG - 1) throw SelfTestFail{"FRI commit: unexpected number of rounds"};
    std::vector<uint8_t> nodes(n * 32);
    std::vector<hodor_fr> r0(n / 2), r1(n / 4), last(2);
    {
        HostXfer xfer(ctx, ctx->stream);
        ST_HIP(xfer.d2h(nodes.data(), p->l0_nodes, n * 32));
        ST_HIP(xfer.d2h(r0.data(), p->inter_values[0], (n / 2) * 32));
        ST_HIP(xfer.d2h(r1.data(), p->inter_values[1], (n / 4) * 32));
        ST_HIP(xfer.d2h(last.data(), p->inter_values[p->num_steps - 1], 64));
        ST_HIP(xfer.finish());
    }
    // ---- the l0 tree against HostBlake2s, the midstate derived afresh (src/iop/blake2s_trivial_iop.rs:8-16, :81-104)
    uint32_t mid[8];
    HostBlake2s::keyed_midstate(mid, (const uint8_t *)"Squeamish Ossifrage", 19, (const uint8_t *)"Shaftoe", 7);
    auto leaf_hash = [&](const hodor_fr &l, uint8_t out[32]) { HostBlake2s::finish(mid, (const uint8_t *)l.l, 32, out); };
    auto node_hash = [&](const uint8_t *l, const uint8_t *r, uint8_t out[32]) {
        uint8_t buf[64];
        memcpy(buf, l, 32);
        memcpy(buf + 32, r, 32);
        HostBlake2s::finish(mid, buf, 64, out);
    };
    uint8_t h[32], hl[32], hr[32];
    for (size_t i = n / 2; i < n; i += n / 16 + 1) {   // the level above the leaves: node i = H(H(leaf 2j), H(leaf 2j + 1)), j = i - n/2
        const size_t j = i - n / 2;
        leaf_hash(v[2 * j], hl);
        leaf_hash(v[2 * j + 1], hr);
        node_hash(hl, hr, h);
        if (memcmp(h, nodes.data() + 32 * i, 32) != 0) throw SelfTestFail{"Merkle tree: leaf level differs from the host's BLAKE2s at node " + std::to_string(i)};
    }
    for (size_t width = n / 4; width >= 1; width /= 2)   // inner levels: a few nodes of each
        for (size_t i = width; i < 2 * width; i += width / 4 + 1) {
            node_hash(nodes.data() + 32 * (2 * i), nodes.data() + 32 * (2 * i + 1), h);
            if (memcmp(h, nodes.data() + 32 * i, 32) != 0) throw SelfTestFail{"Merkle tree: node " + std::to_string(i) + " differs from the host's BLAKE2s"};
        }
    if (memcmp(nodes.data() + 32, p->roots.data(), 32) != 0) throw SelfTestFail{"FRI commit: the l0 root is not the tree's"};
    // ---- challenges and folds (src/fri/fri_on_values.rs:51, :70-104)
    HFr winv, two_inv;
    if (!F.inverse(w, &winv) || !F.inverse(F.from_u64(2), &two_inv)) throw SelfTestFail{"field inversion failed"};
    auto check_fold = [&](const hodor_fr *src, size_t half, const HFr &step, const hodor_fr *got, size_t round) {
        hodor_fr c;
        if (hodor_iop_challenge(ctx, p->roots.data() + 32 * round, &c)) throw SelfTestFail{"interpret_hash failed"};
        if (memcmp(c.l, p->challenges[round].l, 32) != 0) throw SelfTestFail{"FRI commit: challenge " + std::to_string(round) + " differs from interpret_hash of its root"};
        const HFr beta = to_h(&c);
        HFr u = F.one;
        for (size_t i = 0; i < half; i++) {
            const HFr a = to_h(&src[i]), b = to_h(&src[i + half]);
            const HFr odd = F.mul(F.sub(a, b), u);
            const HFr e = F.mul(F.add(F.mul(odd, beta), F.add(a, b)), two_inv);
            if (memcmp(e.l, got[i].l, 32) != 0)
                throw SelfTestFail{"FRI commit: fold of round " + std::to_string(round) + " differs from the host's at element " + std::to_string(i)};
            u = F.mul(u, step);
        }
    };
    check_fold(v.data(), n / 2, winv, r0.data(), 0);                       // k_fri_fold (or fused into the tree's leaf launch)
    check_fold(r0.data(), n / 4, F.mul(winv, winv), r1.data(), 1);         // inside k_fri_tail
    const HFr fin = F.mul(F.add(to_h(&last[0]), to_h(&last[1])), two_inv);   // ifft of the last 2 values, truncated to 1 coefficient
    if (memcmp(fin.l, p->final_coeffs[0].l, 32) != 0) throw SelfTestFail{"FRI commit: final coefficient differs from the host's"};
}

thread_local std::string g_create_err;

}  // namespace

const char *ctx_create_error() { return g_create_err.c_str(); }

// called by hodor_ctx_create on a fully initialised context with a device; HODOR_OK or HODOR_ERR_DEVICE
int ctx_self_test(hodor_ctx *ctx)
{
    static const int enabled = [] { const char *e = getenv("HODOR_SELFTEST"); return (e && *e == '0') ? 0 : 1; }();
    static const int corrupt = [] { const char *e = getenv("HODOR_SELFTEST_CORRUPT"); return e ? atoi(e) : 0; }();
    g_create_err.clear();
    if (!enabled) return HODOR_OK;
    switch (corrupt) {
    case 1: flip_param_bit(&ctx->Q); break;
    case 2: flip_param_bit(&ctx->mid.h[3]); break;
    case 3: flip_param_bit(&ctx->K9.k[4]); break;
    case 4: flip_param_bit(&ctx->P.p[2]); break;
    default: break;
    }
    try {
        check_transforms(ctx, corrupt);
        check_round_trip(ctx);
        check_commit(ctx);
    } catch (const SelfTestFail &f) {
        g_create_err = "hodor_ctx_create: start-up self-test failed: " + f.what;
        (void)hipStreamSynchronize(ctx->stream);
        return HODOR_ERR_DEVICE;
    } catch (...) {
        g_create_err = "hodor_ctx_create: start-up self-test failed (out of host memory)";
        (void)hipStreamSynchronize(ctx->stream);
        return HODOR_ERR_DEVICE;
    }
    ctx->host_round_trips.store(0);   // the test's own transfers are not the caller's
    ctx->h2d_bytes.store(0);
    ctx->d2h_bytes.store(0);
    return HODOR_OK;
}
