// hodor.hpp — C++ host-side mirror of the reference's L3 interface for the hot path, written on top
// of the C ABI only (include/hodor_gpu.h).  Same names, argument meaning and error behaviour as the
// Rust items they stand for, so host code (and tests) read like the reference's:
//
//   hodor::Field                    the `F: PrimeField` type parameter + the `Worker` (one hodor_ctx)
//   hodor::Domain                   src/domains/mod.rs:14-71
//   hodor::Polynomial<Form>         src/polynomials/mod.rs:26-34, :37-137, :139-712 (Coefficients), :715-955 (Values)
//   hodor::Blake2sIopTree           src/iop/blake2s_trivial_iop.rs:106-280
//   hodor::TrivialBlake2sIOP        src/iop/blake2s_trivial_iop.rs:282-339 (+ Query :341-375)
//   hodor::NaiveFriIop              src/fri/mod.rs:63-117, :156-248, src/fri/fri_on_values.rs:11-159
//
// Since round 5 every object is DEVICE-RESIDENT: a Polynomial owns a `hodor_poly` handle (its `coeffs: Vec<F>`
// lives in HBM), an IOP a `hodor_iop`, a prototype a `hodor_fri_proto`.  The methods are the reference's and each is
// one call of the handle API; what reaches the host are roots, evaluations, query answers, proofs — and `as_ref()`,
// which materialises a host copy on demand exactly where Rust code would look at the slice.  This is the stand-in
// for the Rust `struct Polynomial` of INTEGRATION.md §3: src/arp, src/ali and src/prover compile against it unchanged.
//
// Errors: SynthesisError::Error and the reference's asserts become hodor::SynthesisError exceptions
// (thrown on this side of the ABI; the ABI itself returns status codes).
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/hodor_gpu.h"

// equality on the C element type (global scope so that std::vector<hodor_fr> comparisons find it)
inline bool operator==(const hodor_fr &a, const hodor_fr &b)
{
    return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
inline bool operator!=(const hodor_fr &a, const hodor_fr &b) { return !(a == b); }

namespace hodor {

struct SynthesisError : std::runtime_error {
    int code;
    SynthesisError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

typedef hodor_fr Fr;

// One prime field on one device: what `F: PrimeField` + `Worker` are to the reference.
class Field {
  public:
    Field(const uint64_t modulus[4], uint64_t generator, int device = 0)
    {
        int rc = hodor_ctx_create(modulus, generator, device, &ctx_);
        if (rc) throw SynthesisError(rc, "hodor_ctx_create failed");
        hodor_ctx_field_info(ctx_, &info_);
    }
    ~Field() { hodor_ctx_destroy(ctx_); }
    Field(const Field &) = delete;
    Field &operator=(const Field &) = delete;

    hodor_ctx *ctx() const { return ctx_; }
    uint32_t S() const { return info_.s; }
    uint32_t capacity() const { return info_.capacity; }
    Fr one() const { return info_.one; }
    Fr zero() const { return Fr{{0, 0, 0, 0}}; }
    Fr multiplicative_generator() const { return info_.generator; }
    Fr root_of_unity() const { return info_.root_of_unity; }

    Fr mul(const Fr &a, const Fr &b) const { Fr r; hodor_fr_mul(ctx_, &a, &b, &r); return r; }
    Fr add(const Fr &a, const Fr &b) const { Fr r; hodor_fr_add(ctx_, &a, &b, &r); return r; }
    Fr sub(const Fr &a, const Fr &b) const { Fr r; hodor_fr_sub(ctx_, &a, &b, &r); return r; }
    Fr negate(const Fr &a) const { return sub(zero(), a); }
    Fr pow(const Fr &a, uint64_t e) const { Fr r; hodor_fr_pow(ctx_, &a, e, &r); return r; }
    Fr inverse(const Fr &a) const
    {
        Fr r;
        if (hodor_fr_inverse(ctx_, &a, &r)) throw SynthesisError(HODOR_ERR_INVALID, "inverse of zero");
        return r;
    }
    Fr from_u64(uint64_t v) const
    {
        uint64_t c[4] = {v, 0, 0, 0};
        Fr r;
        hodor_fr_from_repr(ctx_, c, &r);
        return r;
    }
    void check(int rc, const char *what) const
    {
        if (rc) throw SynthesisError(rc, std::string(what) + ": " + hodor_last_error(ctx_));
    }
    // device -> host results handed out so far (the prover's stalls) and the bytes that crossed PCIe either way;
    // synchronize = wait for everything enqueued
    uint64_t host_round_trips() const { return hodor_ctx_host_round_trips(ctx_); }
    std::pair<uint64_t, uint64_t> host_traffic() const   // (host -> device, device -> host) bytes
    {
        uint64_t up = 0, down = 0;
        hodor_ctx_host_traffic(ctx_, &up, &down);
        return {up, down};
    }
    void reset_host_round_trips() const { hodor_ctx_reset_host_round_trips(ctx_); }
    void synchronize() const { check(hodor_ctx_synchronize(ctx_), "synchronize"); }

  private:
    hodor_ctx *ctx_ = nullptr;
    hodor_field_info info_;
};

// src/fft/multicore.rs:16-107 — Worker: `scope(elements, |scope, chunk| ...)` hands the closure a scope to spawn threads
// on and the chunk size elements / cpus (1 when there are fewer elements than cpus); every spawned thread is joined
// before scope returns.  The host-side loops of the layers above the boundary (ALI's divisor precompute,
// src/ali/per_register/mod.rs:116-157) run on it exactly as written.
class Worker {
  public:
    size_t cpus;
    Worker() : cpus(std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1) {}
    explicit Worker(size_t c) : cpus(c ? c : 1) {}                                        // new_with_cpus :26
    uint32_t log_num_cpus() const { uint32_t r = 0; while (((size_t)1 << (r + 1)) <= cpus) r++; return r; }   // :38
    size_t get_chunk_size(size_t elements) const { return elements < cpus ? 1 : elements / cpus; }           // :76-87
    struct Scope {
        std::vector<std::thread> threads;
        template <class G> void spawn(G g) { threads.emplace_back(std::move(g)); }
        ~Scope() { for (auto &t : threads) t.join(); }
    };
    template <class Fn> void scope(size_t elements, Fn f) const                                              // :61-74
    {
        Scope s;
        f(s, get_chunk_size(elements));
    }
};

// src/domains/mod.rs:14-71
struct Domain {
    uint64_t size;
    uint64_t power_of_two;
    Fr generator;

    static Domain new_for_size(const Field &F, uint64_t size)
    {
        Domain d;
        uint32_t k;
        int rc = hodor_domain_new_for_size(F.ctx(), size, &d.size, &k, &d.generator);
        if (rc) throw SynthesisError(rc, "Domain::new_for_size: size exceeds the field's 2-adicity");
        d.power_of_two = k;
        return d;
    }
    static std::vector<size_t> coset_for_natural_index_and_size(size_t natural_index, size_t domain_size)
    {
        size_t pair = (natural_index + domain_size / 2) % domain_size;
        if (natural_index < pair) return {natural_index, pair};
        return {pair, natural_index};
    }
    static std::pair<size_t, size_t> index_and_size_for_next_domain(size_t natural_index, size_t domain_size)
    {
        size_t next = domain_size / 2;
        return {natural_index < next ? natural_index : natural_index - next, next};
    }
};

// src/precomputations/mod.rs:7-66 — omegas[i] = w^i, coset[i] = g*w^i (domain size), omegas_inv[i] = w^-i
// (half the domain); the tables are produced on the device and downloaded.
struct PrecomputedOmegas {
    std::vector<Fr> omegas, coset, omegas_inv;

    static PrecomputedOmegas new_for_domain(const Field &F, const Domain &domain)
    {
        PrecomputedOmegas t;
        const size_t n = (size_t)domain.size;
        t.omegas.resize(n);
        t.coset.resize(n);
        t.omegas_inv.resize(n / 2);
        void *dev = nullptr;
        F.check(hodor_buf_alloc(F.ctx(), (2 * n + n / 2 + 1) * sizeof(Fr), &dev), "PrecomputedOmegas: alloc");
        Fr *d = static_cast<Fr *>(dev);
        int rc = hodor_precomputed_omegas_dev(F.ctx(), nullptr, (uint32_t)domain.power_of_two, d, d + n, d + 2 * n);
        if (!rc) rc = hodor_buf_download(F.ctx(), t.omegas.data(), d, n * sizeof(Fr));
        if (!rc) rc = hodor_buf_download(F.ctx(), t.coset.data(), d + n, n * sizeof(Fr));
        if (!rc && n >= 2) rc = hodor_buf_download(F.ctx(), t.omegas_inv.data(), d + 2 * n, (n / 2) * sizeof(Fr));
        hodor_buf_free(F.ctx(), dev);
        F.check(rc, "PrecomputedOmegas::new_for_domain");
        return t;
    }
};

struct Coefficients { static constexpr int form = HODOR_FORM_COEFFICIENTS; };
struct Values { static constexpr int form = HODOR_FORM_VALUES; };

// what `&[F]` is to the reference's callers: a read-only view of the host copy `as_ref()` materialised
struct Slice {
    const Fr *p;
    size_t n;
    size_t size() const { return n; }
    const Fr &operator[](size_t i) const { return p[i]; }
    const Fr *begin() const { return p; }
    const Fr *end() const { return p + n; }
    std::vector<Fr> to_vec() const { return std::vector<Fr>(p, p + n); }
    bool operator==(const Slice &o) const { return n == o.n && memcmp(p, o.p, n * sizeof(Fr)) == 0; }
    bool operator!=(const Slice &o) const { return !(*this == o); }
    bool operator==(const std::vector<Fr> &o) const { return n == o.size() && memcmp(p, o.data(), n * sizeof(Fr)) == 0; }
    bool operator!=(const std::vector<Fr> &o) const { return !(*this == o); }
};

// what `&mut [F]` is: the polynomial's host image, writable, for as long as this guard lives — the Rust borrow.  While
// it lives the image IS the vector; its destructor writes the image back to the device in one upload
// (hodor_poly_commit_mut_h).  chunks_mut(chunk) = slice::chunks_mut.
struct MutSlice {
    Fr *p = nullptr;
    size_t n = 0;
    hodor_poly *h = nullptr;
    MutSlice(Fr *ptr, size_t len, hodor_poly *handle) : p(ptr), n(len), h(handle) {}
    MutSlice(const MutSlice &) = delete;
    MutSlice &operator=(const MutSlice &) = delete;
    MutSlice(MutSlice &&o) noexcept : p(o.p), n(o.n), h(o.h) { o.h = nullptr; }
    ~MutSlice() { if (h) (void)hodor_poly_commit_mut_h(h); }   // a failed upload leaves the image pending: the next device operation retries and reports
    size_t size() const { return n; }
    Fr &operator[](size_t i) const { return p[i]; }
    Fr *begin() const { return p; }
    Fr *end() const { return p + n; }
    struct Chunk {
        Fr *p;
        size_t n;
        Fr *begin() const { return p; }
        Fr *end() const { return p + n; }
        size_t size() const { return n; }
        Fr &operator[](size_t i) const { return p[i]; }
    };
    std::vector<Chunk> chunks_mut(size_t chunk) const
    {
        std::vector<Chunk> out;
        for (size_t o = 0; o < n; o += chunk) out.push_back(Chunk{p + o, n - o < chunk ? n - o : chunk});
        return out;
    }
};

// src/polynomials/mod.rs:26-34.  Move-only like the Rust value; clone() is #[derive(Clone)].
template <class Form>
class Polynomial {
  public:
    const Field *F = nullptr;
    hodor_poly *h = nullptr;          // coeffs: Vec<F>, in HBM
    uint32_t exp = 0;
    Fr omega, omegainv, geninv, minv;

    Polynomial() = default;
    Polynomial(const Field &field, hodor_poly *handle) : F(&field), h(handle) { refresh(); }
    ~Polynomial() { if (h) hodor_poly_free_h(h); }
    Polynomial(const Polynomial &) = delete;
    Polynomial &operator=(const Polynomial &) = delete;
    Polynomial(Polynomial &&o) noexcept { take(o); }
    Polynomial &operator=(Polynomial &&o) noexcept
    {
        if (this != &o) { if (h) hodor_poly_free_h(h); take(o); }
        return *this;
    }
    Polynomial clone() const
    {
        hodor_poly *c = nullptr;
        F->check(hodor_poly_clone_h(h, &c), "clone");
        return Polynomial(*F, c);
    }

    // from_coeffs / from_values (:146-166, :722-742) and new_for_size (:140-144, :716-720)
    static Polynomial from_vec(const Field &F, const std::vector<Fr> &v)
    {
        hodor_poly *p = nullptr;
        F.check(hodor_poly_from_host_h(F.ctx(), Form::form, v.data(), v.size(), &p), "from_coeffs / from_values");
        return Polynomial(F, p);
    }
    static Polynomial new_for_size(const Field &F, size_t size)
    {
        hodor_poly *p = nullptr;
        F.check(hodor_poly_new_for_size_h(F.ctx(), Form::form, size, &p), "new_for_size");
        return Polynomial(F, p);
    }
    // elements [first, first + count) of the synthetic SplitMix64 stream, generated in place
    static Polynomial generated(const Field &F, uint64_t first, size_t count, uint64_t seed)
    {
        hodor_poly *p = nullptr;
        F.check(hodor_poly_gen_h(F.ctx(), Form::form, first, count, seed, &p), "gen_elements");
        return Polynomial(F, p);
    }

    size_t size() const { return hodor_poly_size_h(h); }                                  // :38
    Slice as_ref() const                                                                  // :42
    {
        const Fr *p = nullptr;
        F->check(hodor_poly_as_ref_h(h, &p), "as_ref");
        return Slice{p, size()};
    }
    // as_mut() :46 — the whole vector as `&mut [F]`: one download (none for new_for_size's zeros), the write-back
    // when the guard goes out of scope.  `poly.as_mut()[1] = F.one();` works on the temporary guard like the Rust line.
    MutSlice as_mut()
    {
        Fr *p = nullptr;
        F->check(hodor_poly_as_mut_h(h, &p), "as_mut");
        return MutSlice(p, size(), h);
    }
    std::vector<Fr> into_coeffs() && { return as_ref().to_vec(); }                        // :50
    // as_ref()[i] / as_mut()[i] = v / as_mut()[i].sub_assign(&v) ... without moving the rest of the vector
    Fr at(size_t i) const { Fr v; F->check(hodor_poly_read_h(h, i, 1, &v), "as_ref()[i]"); return v; }
    void set(size_t i, const Fr &v) { F->check(hodor_poly_write_h(h, i, 1, &v), "as_mut()[i] = v"); }
    void add_assign_at(size_t i, const Fr &v) { F->check(hodor_poly_elem_op_h(h, i, HODOR_UN_ADD_CONSTANT, &v, 0), "as_mut()[i] += v"); }
    void sub_assign_at(size_t i, const Fr &v) { F->check(hodor_poly_elem_op_h(h, i, HODOR_UN_SUB_CONSTANT, &v, 0), "as_mut()[i] -= v"); }

    void distribute_powers(const Fr &g) { F->check(hodor_poly_distribute_powers_h(h, &g), "distribute_powers"); }   // :54
    void scale(const Fr &g) { F->check(hodor_poly_scale_h(h, &g), "scale"); }                                       // :59
    void negate() { F->check(hodor_poly_negate_h(h), "negate"); }                                                   // :72
    // pad_by_factor / pad_to_size (:85-125): `false` stands for Err(SynthesisError::Error)
    bool pad_by_factor(size_t factor) { return resized(hodor_poly_pad_by_factor_h(h, factor), "pad_by_factor"); }
    bool pad_to_size(size_t new_size) { return resized(hodor_poly_pad_to_size_h(h, new_size), "pad_to_size"); }
    void trim_to_degree(size_t degree) { F->check(hodor_poly_trim_to_degree_h(h, degree), "trim_to_degree"); }     // :127

    // add_assign / sub_assign / add_assign_scaled (:640-683 Coefficients, :817-873 Values)
    void add_assign(const Polynomial &o) { F->check(hodor_poly_binary_h(h, o.h, HODOR_OP_ADD), "add_assign"); }
    void sub_assign(const Polynomial &o) { F->check(hodor_poly_binary_h(h, o.h, HODOR_OP_SUB), "sub_assign"); }
    void add_assign_scaled(const Polynomial &o, const Fr &s) { F->check(hodor_poly_add_assign_scaled_h(h, o.h, &s), "add_assign_scaled"); }

    // Coefficients only
    Fr evaluate_at(const Fr &g) const                                                     // :685-711
    {
        static_assert(std::is_same<Form, Coefficients>::value, "evaluate_at exists on Polynomial<F, Coefficients>");
        Fr r;
        F->check(hodor_poly_evaluate_at_h(h, &g, &r), "evaluate_at");
        return r;
    }
    // Values only (:744-771, :831-841, :875-954)
    void pow(uint64_t e) { values_only(); F->check(hodor_poly_pow_h(h, e), "pow"); }
    void square() { values_only(); F->check(hodor_poly_square_h(h), "square"); }
    void add_constant(const Fr &c) { values_only(); F->check(hodor_poly_add_constant_h(h, &c), "add_constant"); }
    void mul_assign(const Polynomial &o) { values_only(); F->check(hodor_poly_binary_h(h, o.h, HODOR_OP_MUL), "mul_assign"); }
    bool batch_inversion()   // Err(SynthesisError::Error) when an element is zero (:909) -> false, data untouched
    {
        values_only();
        int rc = hodor_poly_batch_inversion_h(h);
        if (rc == HODOR_ERR_INVALID) return false;
        F->check(rc, "batch_inversion");
        return true;
    }

    bool operator==(const Polynomial &o) const   // derive(PartialEq): decided on the device, 4 bytes come back
    {
        int eq = 0;
        F->check(hodor_poly_equal_h(h, o.h, &eq), "eq");
        return eq != 0;
    }
    bool operator!=(const Polynomial &o) const { return !(*this == o); }

    void refresh()   // the cached domain constants follow the handle
    {
        hodor_poly_info i;
        hodor_poly_info_h(h, &i);
        exp = i.exp; omega = i.omega; omegainv = i.omegainv; geninv = i.geninv; minv = i.minv;
    }
    // the handle changed its form in place (fft / ifft ...): the C++ value changes its type
    template <class To>
    Polynomial<To> retype() &&
    {
        Polynomial<To> q(*F, h);
        h = nullptr;
        return q;
    }

  private:
    static void values_only() { static_assert(std::is_same<Form, Values>::value, "this method exists on Polynomial<F, Values>"); }
    void take(Polynomial &o)
    {
        F = o.F; h = o.h; exp = o.exp; omega = o.omega; omegainv = o.omegainv; geninv = o.geninv; minv = o.minv;
        o.h = nullptr;
    }
    bool resized(int rc, const char *what)
    {
        if (rc == HODOR_ERR_SIZE) return false;
        F->check(rc, what);
        refresh();
        return true;
    }
};

inline Polynomial<Coefficients> from_coeffs(const Field &F, const std::vector<Fr> &c)
{
    return Polynomial<Coefficients>::from_vec(F, c);
}
inline Polynomial<Values> from_values(const Field &F, const std::vector<Fr> &v)
{
    return Polynomial<Values>::from_vec(F, v);
}

// Polynomial<F, Coefficients>::fft / coset_fft / coset_fft_for_generator (:611-638): consume self
inline Polynomial<Values> fft(Polynomial<Coefficients> p)
{
    p.F->check(hodor_poly_fft_h(p.h), "fft");
    return std::move(p).retype<Values>();
}
inline Polynomial<Values> coset_fft(Polynomial<Coefficients> p)
{
    p.F->check(hodor_poly_coset_fft_h(p.h), "coset_fft");
    return std::move(p).retype<Values>();
}
inline Polynomial<Values> coset_fft_for_generator(Polynomial<Coefficients> p, const Fr &gen)
{
    p.F->check(hodor_poly_coset_fft_for_generator_h(p.h, &gen), "coset_fft_for_generator");
    return std::move(p).retype<Values>();
}
// Polynomial<F, Values>::ifft / icoset_fft / icoset_fft_for_generator (:773-815)
inline Polynomial<Coefficients> ifft(Polynomial<Values> p)
{
    p.F->check(hodor_poly_ifft_h(p.h), "ifft");
    return std::move(p).retype<Coefficients>();
}
inline Polynomial<Coefficients> icoset_fft(Polynomial<Values> p)
{
    p.F->check(hodor_poly_icoset_fft_h(p.h), "icoset_fft");
    return std::move(p).retype<Coefficients>();
}
// `geninv` is the inverse of the coset generator, as in the reference
inline Polynomial<Coefficients> icoset_fft_for_generator(Polynomial<Values> p, const Fr &geninv)
{
    p.F->check(hodor_poly_icoset_fft_for_generator_h(p.h, &geninv), "icoset_fft_for_generator");
    return std::move(p).retype<Coefficients>();
}
// lde / coset_lde (:343-349 -> :418-482, :544-609).  The Rust methods consume self and callers clone first
// (src/prover/mod.rs:74: w.clone().lde(..)); taking a reference here saves exactly that clone.
inline Polynomial<Values> lde_impl(const Polynomial<Coefficients> &p, size_t factor, bool coset)
{
    hodor_poly *q = nullptr;
    p.F->check(hodor_poly_lde_h(p.h, factor, coset ? 1 : 0, &q), "lde");   // HODOR_ERR_SIZE: assert!(factor.is_power_of_two())
    return Polynomial<Values>(*p.F, q);
}
inline Polynomial<Values> lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, false); }
inline Polynomial<Values> coset_lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, true); }
// filtering_lde / coset_filtering_lde (:355-368, :484-499): zero-pad then best_lde — the same values as lde / coset_lde
// (asserted by the reference, :1026-1031); on the device both ARE the one zero-padded transform
inline Polynomial<Values> filtering_lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, false); }
inline Polynomial<Values> coset_filtering_lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, true); }
// every register at once (src/prover/mod.rs:73-76): one batched launch sequence, the outputs share one allocation
inline std::vector<Polynomial<Values>> lde_all(const std::vector<Polynomial<Coefficients>> &polys, size_t factor, bool coset = false)
{
    std::vector<Polynomial<Values>> out;
    if (polys.empty()) return out;
    std::vector<const hodor_poly *> in;
    for (auto &p : polys) in.push_back(p.h);
    std::vector<hodor_poly *> hs(polys.size(), nullptr);
    const Field &F = *polys[0].F;
    F.check(hodor_poly_lde_batch_h(in.data(), in.size(), factor, coset ? 1 : 0, hs.data()), "lde (batch)");
    for (auto *q : hs) out.emplace_back(F, q);
    return out;
}
// (coset_)evaluate_at_domain_for_degree_one (:229-290) of a 2-coefficient polynomial q(x) = c0 + c1 x
inline Polynomial<Values> evaluate_at_domain_for_degree_one(const Polynomial<Coefficients> &q, size_t domain_size, bool coset = false);
inline Polynomial<Values> coset_evaluate_at_domain_for_degree_one(const Polynomial<Coefficients> &q, size_t domain_size)   // :260-290
{
    return evaluate_at_domain_for_degree_one(q, domain_size, true);
}
inline Polynomial<Values> evaluate_at_domain_for_degree_one(const Polynomial<Coefficients> &q, size_t domain_size, bool coset)
{
    if (q.size() != 2) throw SynthesisError(HODOR_ERR_SIZE, "evaluate_at_domain_for_degree_one: assert_eq!(self.coeffs.len(), 2)");
    const Fr c0 = q.at(0), c1 = q.at(1);   // a small polynomial built on the host is still known there: no round trip
    hodor_poly *v = nullptr;
    size_t n = 1;
    while (n < domain_size) n <<= 1;       // Domain::new_for_size(domain_size) :235
    q.F->check(hodor_poly_degree_one_on_domain_h(q.F->ctx(), n, &c1, &c0, coset ? 1 : 0, &v), "evaluate_at_domain_for_degree_one");
    return Polynomial<Values>(*q.F, v);
}
// ALIInstance::from_arp's inverse_divisor_for_dense_constraint_in_coset (src/ali/per_register/mod.rs:60-160) without
// the two host passes: prod_j (x - roots[j]) / (x^T - 1) on the coset of the evaluation domain, generated on the device
// (include/hodor_gpu.h: hodor_poly_dense_divisor_on_coset_h).  T = column_size.
inline Polynomial<Values> dense_divisor_on_coset(const Field &F, size_t evaluation_size, size_t column_size,
                                                 const std::vector<Fr> &roots)
{
    hodor_poly *v = nullptr;
    F.check(hodor_poly_dense_divisor_on_coset_h(F.ctx(), evaluation_size, column_size, roots.data(), roots.size(), &v),
            "dense_divisor_on_coset");
    return Polynomial<Values>(F, v);
}
// one DEEP quotient term in one pass (include/hodor_gpu.h: hodor_poly_quotient_term_h) — the fused form of
// clone / add_constant / scale / mul_assign / add_assign (src/ali/per_register/deep.rs:74-84)
inline void quotient_term(Polynomial<Values> &acc, const Polynomial<Values> &f, const Polynomial<Values> &divisor_inv,
                          const Fr &value, const Fr *alpha, bool accumulate)
{
    acc.F->check(hodor_poly_quotient_term_h(acc.h, f.h, divisor_inv.h, &value, alpha, accumulate ? 1 : 0), "quotient_term");
}
// Polynomial::from_roots (:168-227): prod (x - r_i) — per-chunk products on the host, then lde / mul_assign / ifft
// exactly as the reference composes them
inline Polynomial<Coefficients> from_roots(const Field &F, const std::vector<Fr> &roots, size_t chunks = 4)
{
    if (roots.empty()) throw SynthesisError(HODOR_ERR_INVALID, "from_roots: result.expect(\"is some\")");
    Domain domain = Domain::new_for_size(F, roots.size() + 1);
    const size_t chunk = (roots.size() + chunks - 1) / chunks;
    std::unique_ptr<Polynomial<Values>> result;
    for (size_t start = 0; start < roots.size(); start += chunk) {
        std::vector<Fr> s;
        for (size_t k = start; k < roots.size() && k < start + chunk; k++) {
            const Fr &r = roots[k];
            if (s.empty()) { s = {F.negate(r), F.one()}; continue; }                                   // :187-190
            std::vector<Fr> tmp(s.size() + 1, F.zero());
            for (size_t i = 0; i < s.size(); i++) tmp[i + 1] = s[i];                                    // x * s
            for (size_t i = 0; i < s.size(); i++) tmp[i] = F.sub(tmp[i], F.mul(s[i], r));               // - r * s  :195-199
            s = tmp;
        }
        auto t = from_coeffs(F, s);
        auto tv = lde(t, (size_t)domain.size / t.size());                                               // :214-216
        if (result) result->mul_assign(tv); else result.reset(new Polynomial<Values>(std::move(tv)));   // :217-221
    }
    return ifft(std::move(*result));                                                                    // :225
}

typedef std::vector<uint8_t> Hash32;   // [u8; 32]

// ---- CosetCombiner (src/iop/mod.rs:22-34) -------------------------------------------------------------------
// TrivialCombiner is the reference's only instance (src/iop/trivial_coset_combiner.rs:17-53); Coset2Combiner is the
// opt-in format of this build (HODOR_COMBINER_COSET2: the coset {i, i + n/2} is ONE 64-byte leaf — the README's
// unchecked "coset combining", README.md:46).  The reference's trait maps indices without knowing the domain size;
// a non-trivial combiner needs it, so the maps take `domain_size` here (INTEGRATION.md shows the Rust side).
struct TrivialCombiner {
    static constexpr int id = HODOR_COMBINER_TRIVIAL;
    static constexpr size_t COSET_SIZE = 2, EXPECTED_DEGREE = 2;
    static size_t tree_index_into_natural_index(size_t t, size_t) { return t; }
    static size_t natural_index_into_tree_index(size_t i, size_t) { return i; }
    static std::vector<size_t> get_coset_for_natural_index(size_t i, size_t n)
    {
        return Domain::coset_for_natural_index_and_size(i, n);
    }
};
struct Coset2Combiner {
    static constexpr int id = HODOR_COMBINER_COSET2;
    static constexpr size_t COSET_SIZE = 2, EXPECTED_DEGREE = 2;
    static size_t tree_index_into_natural_index(size_t t, size_t n) { return (t >> 1) + (t & 1) * (n / 2); }
    static size_t natural_index_into_tree_index(size_t i, size_t n) { return 2 * (i % (n / 2)) + i / (n / 2); }
    static std::vector<size_t> get_coset_for_natural_index(size_t i, size_t n)
    {
        return Domain::coset_for_natural_index_and_size(i, n);
    }
};

// src/iop/blake2s_trivial_iop.rs:341-375
struct TrivialBlake2sIopQuery {
    size_t index;
    Fr value_;
    std::vector<Hash32> path_;
    size_t tree_index() const { return index; }
    size_t natural_index() const { return index; }
    const Fr &value() const { return value_; }
    const std::vector<Hash32> &path() const { return path_; }
};
// the IOP over a COSET2 tree: a query answers for the whole coset (both values, ONE path of log2(n) - 1 digests)
struct Coset2Blake2sIopQuery {
    size_t index;                 // the smaller member of the coset = the leaf index
    Fr values_[2];                // value[index], value[index + n/2]
    std::vector<Hash32> path_;
    size_t natural_index() const { return index; }
    const std::vector<Hash32> &path() const { return path_; }
};

// src/iop/blake2s_trivial_iop.rs:106-339: Blake2sIopTree and the IOP over it, `nodes` in HBM.  `create` takes the
// device-resident vector (IOP::create(lde.as_ref()), src/prover/mod.rs:77-79) or a host vector (which it uploads and keeps).
template <class Combiner>
class Blake2sIOP {
  public:
    const Field *F = nullptr;
    hodor_iop *h = nullptr;
    std::shared_ptr<Polynomial<Values>> own;    // leaves uploaded by create(F, vector)

    Blake2sIOP() = default;
    Blake2sIOP(const Field &field, hodor_iop *handle) : F(&field), h(handle) {}
    ~Blake2sIOP() { if (h) hodor_iop_free_h(h); }
    Blake2sIOP(const Blake2sIOP &) = delete;
    Blake2sIOP &operator=(const Blake2sIOP &) = delete;
    Blake2sIOP(Blake2sIOP &&o) noexcept : F(o.F), h(o.h), own(std::move(o.own)) { o.h = nullptr; }
    Blake2sIOP &operator=(Blake2sIOP &&o) noexcept
    {
        if (this != &o) { if (h) hodor_iop_free_h(h); F = o.F; h = o.h; own = std::move(o.own); o.h = nullptr; }
        return *this;
    }

    static Blake2sIOP create(const Field &F, const Polynomial<Values> &leafs)
    {
        hodor_iop *t = nullptr;
        F.check(hodor_iop_create_h(leafs.h, Combiner::id, &t), "IopTree::create");   // assert!(size.is_power_of_two()) :137
        return Blake2sIOP(F, t);
    }
    static Blake2sIOP create(const Field &F, const std::vector<Fr> &leafs)
    {
        if (leafs.empty() || (leafs.size() & (leafs.size() - 1)))
            throw SynthesisError(HODOR_ERR_SIZE, "IopTree::create: the number of leaves must be a power of two");
        auto v = std::make_shared<Polynomial<Values>>(from_values(F, leafs));
        Blake2sIOP t = create(F, *v);
        t.own = v;
        return t;
    }
    // all registers' oracles in one batched commit (src/prover/mod.rs:77-79)
    static std::vector<Blake2sIOP> create_all(const Field &F, const std::vector<Polynomial<Values>> &ldes)
    {
        std::vector<const hodor_poly *> in;
        for (auto &l : ldes) in.push_back(l.h);
        std::vector<hodor_iop *> ts(ldes.size(), nullptr);
        if (!ldes.empty()) F.check(hodor_iop_create_batch_h(in.data(), in.size(), Combiner::id, ts.data()), "IopTree::create (batch)");
        std::vector<Blake2sIOP> out;
        for (auto *t : ts) out.emplace_back(F, t);
        return out;
    }
    uint64_t size() const { return hodor_iop_size_h(h); }
    Hash32 get_root() const                                                               // :221
    {
        Hash32 r(32);
        F->check(hodor_iop_root_h(h, r.data()), "get_root");
        return r;
    }
    template <class AnyIOP>
    static std::vector<Hash32> get_roots(const Field &F, const std::vector<AnyIOP> &iops)   // many roots, one wait
    {
        std::vector<hodor_iop *> ts;
        for (auto &t : iops) ts.push_back(t.h);
        std::vector<uint8_t> flat(32 * ts.size());
        if (!ts.empty()) F.check(hodor_iop_roots_h(ts.data(), ts.size(), flat.data()), "get_root (batch)");
        std::vector<Hash32> out;
        for (size_t i = 0; i < ts.size(); i++) out.emplace_back(flat.begin() + 32 * i, flat.begin() + 32 * (i + 1));
        return out;
    }
    std::vector<uint8_t> nodes() const   // the whole heap array (tests)
    {
        std::vector<uint8_t> n((Combiner::id == HODOR_COMBINER_COSET2 ? size() / 2 : size()) * 32);
        F->check(hodor_iop_nodes_h(h, n.data()), "nodes");
        return n;
    }
    static Fr encode_root_into_challenge(const Field &F, const Hash32 &root)              // :226-234
    {
        Fr r;
        F.check(hodor_iop_challenge(F.ctx(), root.data(), &r), "interpret_hash");
        return r;
    }
    Fr get_challenge_scalar_from_root() const { return encode_root_into_challenge(*F, get_root()); }
    bool operator==(const Blake2sIOP &o) const { return get_root() == o.get_root(); }

  protected:
    // IOP::query(natural_index, leafs) :324-338 — raw form: up to two values + the path
    size_t raw_query(size_t natural_index, const Polynomial<Values> &leafs, Fr values[2], std::vector<Hash32> *path) const
    {
        if (natural_index >= size() || leafs.size() != size())
            throw SynthesisError(HODOR_ERR_SIZE, "query index out of range");             // asserts :325-326
        std::vector<uint8_t> buf(32 * 64);
        size_t cnt = 0;
        F->check(hodor_iop_query_h(h, leafs.h, natural_index, values, buf.data(), &cnt), "query");
        for (size_t i = 0; i < cnt; i++) path->emplace_back(buf.begin() + 32 * i, buf.begin() + 32 * (i + 1));
        return cnt;
    }
    const Polynomial<Values> &owned_leafs(const std::vector<Fr> &leafs) const
    {
        if (!own || leafs.size() != own->size())
            throw SynthesisError(HODOR_ERR_SIZE, "query: pass the vector the oracle was created from");
        return *own;
    }
};

class TrivialBlake2sIOP : public Blake2sIOP<TrivialCombiner> {
  public:
    typedef Blake2sIOP<TrivialCombiner> Base;
    TrivialBlake2sIOP() = default;
    TrivialBlake2sIOP(Base &&b) : Base(std::move(b)) {}
    static TrivialBlake2sIOP create(const Field &F, const Polynomial<Values> &leafs) { return TrivialBlake2sIOP(Base::create(F, leafs)); }
    static TrivialBlake2sIOP create(const Field &F, const std::vector<Fr> &leafs) { return TrivialBlake2sIOP(Base::create(F, leafs)); }
    static std::vector<TrivialBlake2sIOP> create_all(const Field &F, const std::vector<Polynomial<Values>> &ldes)
    {
        std::vector<TrivialBlake2sIOP> out;
        for (auto &b : Base::create_all(F, ldes)) out.emplace_back(std::move(b));
        return out;
    }
    TrivialBlake2sIopQuery query(size_t natural_index, const Polynomial<Values> &leafs) const
    {
        TrivialBlake2sIopQuery q{natural_index, {}, {}};
        Fr v[2];
        raw_query(natural_index, leafs, v, &q.path_);
        q.value_ = v[0];
        return q;
    }
    TrivialBlake2sIopQuery query(size_t natural_index, const std::vector<Fr> &leafs) const { return query(natural_index, owned_leafs(leafs)); }
    static bool verify_query(const Field &F, const TrivialBlake2sIopQuery &q, const Hash32 &root)
    {
        std::vector<uint8_t> flat;
        for (auto &h : q.path_) flat.insert(flat.end(), h.begin(), h.end());
        int ok = 0;
        F.check(hodor_iop_verify(F.ctx(), root.data(), &q.value_, flat.data(), q.path_.size(), q.tree_index(), &ok), "verify");
        return ok != 0;
    }
};
typedef TrivialBlake2sIOP Blake2sIopTree;   // the reference separates the tree from the IOP over it; here they are one object

class Coset2Blake2sIOP : public Blake2sIOP<Coset2Combiner> {
  public:
    typedef Blake2sIOP<Coset2Combiner> Base;
    Coset2Blake2sIOP() = default;
    Coset2Blake2sIOP(Base &&b) : Base(std::move(b)) {}
    static Coset2Blake2sIOP create(const Field &F, const Polynomial<Values> &values) { return Coset2Blake2sIOP(Base::create(F, values)); }
    static Coset2Blake2sIOP create(const Field &F, const std::vector<Fr> &values) { return Coset2Blake2sIOP(Base::create(F, values)); }
    static std::vector<Coset2Blake2sIOP> create_all(const Field &F, const std::vector<Polynomial<Values>> &ldes)
    {
        std::vector<Coset2Blake2sIOP> out;
        for (auto &b : Base::create_all(F, ldes)) out.emplace_back(std::move(b));
        return out;
    }
    Coset2Blake2sIopQuery query(size_t natural_index, const Polynomial<Values> &values) const
    {
        Coset2Blake2sIopQuery q{natural_index % (size() / 2), {}, {}};
        raw_query(natural_index, values, q.values_, &q.path_);
        return q;
    }
    Coset2Blake2sIopQuery query(size_t natural_index, const std::vector<Fr> &values) const { return query(natural_index, owned_leafs(values)); }
    static bool verify_query(const Field &F, const Coset2Blake2sIopQuery &q, const Hash32 &root, size_t n)
    {
        std::vector<uint8_t> flat;
        for (auto &h : q.path_) flat.insert(flat.end(), h.begin(), h.end());
        int ok = 0;
        F.check(hodor_iop_verify_combined(F.ctx(), root.data(), q.values_, flat.data(), q.path_.size(), q.index, n,
                                          HODOR_COMBINER_COSET2, &ok), "verify (COSET2)");
        return ok != 0;
    }
};

// src/fri/mod.rs:139-147
struct FRIProof {
    std::vector<TrivialBlake2sIopQuery> queries;
    std::vector<Hash32> roots;
    std::vector<Fr> final_coefficients;
    size_t initial_degree_plus_one, output_coeffs_at_degree_plus_one, lde_factor;

    // the wire format of hodor_fri_produce_proof / hodor_fri_verify_proof (documented in csrc/abi_fri.hip)
    std::vector<uint8_t> to_bytes() const
    {
        std::vector<uint8_t> out;
        auto put64 = [&](uint64_t v) { for (int b = 0; b < 8; b++) out.push_back((uint8_t)(v >> (8 * b))); };
        auto put = [&](const void *p, size_t n) { out.insert(out.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
        put64(queries.size());
        for (auto &q : queries) {
            put64(q.index);
            put(q.value_.l, 32);
            put64(q.path_.size());
            for (auto &h : q.path_) put(h.data(), 32);
        }
        put64(roots.size());
        for (auto &r : roots) put(r.data(), 32);
        put64(final_coefficients.size());
        for (auto &c : final_coefficients) put(c.l, 32);
        put64(initial_degree_plus_one);
        put64(output_coeffs_at_degree_plus_one);
        put64(lde_factor);
        return out;
    }
    static FRIProof from_bytes(const std::vector<uint8_t> &raw)   // TRIVIAL format (one value per query)
    {
        size_t o = 0;
        auto need = [&](size_t k) { if (o + k > raw.size()) throw SynthesisError(HODOR_ERR_INVALID, "FRIProof: truncated"); };
        auto get64 = [&]() { need(8); uint64_t v = 0; for (int b = 0; b < 8; b++) v |= (uint64_t)raw[o + b] << (8 * b); o += 8; return v; };
        FRIProof p;
        const uint64_t nq = get64();
        for (uint64_t i = 0; i < nq; i++) {
            TrivialBlake2sIopQuery q;
            q.index = (size_t)get64();
            need(32); memcpy(q.value_.l, raw.data() + o, 32); o += 32;
            const uint64_t pl = get64();
            for (uint64_t k = 0; k < pl; k++) { need(32); q.path_.emplace_back(raw.begin() + o, raw.begin() + o + 32); o += 32; }
            p.queries.push_back(std::move(q));
        }
        const uint64_t nr = get64();
        for (uint64_t i = 0; i < nr; i++) { need(32); p.roots.emplace_back(raw.begin() + o, raw.begin() + o + 32); o += 32; }
        const uint64_t nf = get64();
        for (uint64_t i = 0; i < nf; i++) { need(32); Fr c; memcpy(c.l, raw.data() + o, 32); o += 32; p.final_coefficients.push_back(c); }
        p.initial_degree_plus_one = (size_t)get64();
        p.output_coeffs_at_degree_plus_one = (size_t)get64();
        p.lde_factor = (size_t)get64();
        return p;
    }
};

// src/fri/mod.rs:106-117 — the prover's prototype, device-resident: the library keeps every intermediate vector and
// tree in HBM; the fields the transcript and the proof need (roots, challenges, final coefficients) are on the host
class FRIProofPrototype {
  public:
    const Field *F = nullptr;
    hodor_fri_proto *h = nullptr;
    std::vector<Fr> challenges;
    Hash32 final_root;
    std::vector<Fr> final_coefficients;
    size_t initial_degree_plus_one = 0, output_coeffs_at_degree_plus_one = 0, lde_factor = 0;

    FRIProofPrototype() = default;
    ~FRIProofPrototype() { if (h) hodor_fri_free(h); }
    FRIProofPrototype(const FRIProofPrototype &) = delete;
    FRIProofPrototype &operator=(const FRIProofPrototype &) = delete;
    FRIProofPrototype(FRIProofPrototype &&o) noexcept { *this = std::move(o); }
    FRIProofPrototype &operator=(FRIProofPrototype &&o) noexcept
    {
        if (this != &o) {
            if (h) hodor_fri_free(h);
            F = o.F; h = o.h; o.h = nullptr;
            challenges = std::move(o.challenges); final_root = std::move(o.final_root);
            final_coefficients = std::move(o.final_coefficients); roots_ = std::move(o.roots_);
            initial_degree_plus_one = o.initial_degree_plus_one;
            output_coeffs_at_degree_plus_one = o.output_coeffs_at_degree_plus_one;
            lde_factor = o.lde_factor;
        }
        return *this;
    }
    size_t num_steps() const { return challenges.size(); }
    std::vector<Hash32> get_roots() const { return roots_; }                              // :120-128
    Hash32 get_final_root() const { return final_root; }
    std::vector<Fr> get_final_coefficients() const { return final_coefficients; }
    // l0_commitment / intermediate_commitments[i] (:107-108) as IOP objects over the prototype's own trees
    TrivialBlake2sIOP l0_commitment() const { return commitment(-1); }
    TrivialBlake2sIOP intermediate_commitment(size_t i) const { return commitment((int)i); }
    // intermediate_values[i] (:109): a device copy, made when asked for
    Polynomial<Values> intermediate_values(size_t i) const
    {
        hodor_poly *v = nullptr;
        F->check(hodor_fri_intermediate_values_h(h, i, &v), "intermediate_values");
        return Polynomial<Values>(*F, v);
    }
    std::vector<uint8_t> serialized() const   // the canonical prototype bytes (hodor_fri_serialize)
    {
        std::vector<uint8_t> b(hodor_fri_serialize(h, nullptr, 0));
        hodor_fri_serialize(h, b.data(), b.size());
        return b;
    }
    void load(const Field &field, hodor_fri_proto *proto, size_t n, size_t factor, size_t out_deg)
    {
        F = &field; h = proto;
        const size_t steps = hodor_fri_num_steps(h);
        challenges.resize(steps);
        hodor_fri_challenges(h, challenges.data());
        final_root.resize(32);
        hodor_fri_final_root(h, final_root.data());
        final_coefficients.resize(out_deg);
        hodor_fri_final_coefficients(h, final_coefficients.data());
        std::vector<uint8_t> flat(32 * (steps + 1));
        hodor_fri_roots(h, flat.data());
        roots_.clear();
        for (size_t i = 0; i <= steps; i++) roots_.emplace_back(flat.begin() + 32 * i, flat.begin() + 32 * (i + 1));
        initial_degree_plus_one = n / factor;
        output_coeffs_at_degree_plus_one = out_deg;
        lde_factor = factor;
    }

  private:
    std::vector<Hash32> roots_;
    TrivialBlake2sIOP commitment(int step) const
    {
        hodor_iop *t = nullptr;
        F->check(hodor_fri_commitment_h(h, step, &t), "commitment");
        return TrivialBlake2sIOP(Blake2sIOP<TrivialCombiner>(*F, t));
    }
};

// FRIProofPrototype::produce_proof / FriIop::prototype_into_proof, src/fri/query_producer.rs:10-53, src/fri/mod.rs:49-54
inline std::vector<uint8_t> produce_proof_bytes(const FRIProofPrototype &p, const Polynomial<Values> &iop_values,
                                                size_t natural_first_element_index)
{
    const size_t need = hodor_fri_produce_proof_h(p.h, iop_values.h, natural_first_element_index, nullptr, 0);
    if (!need) throw SynthesisError(HODOR_ERR_INVALID, "produce_proof");
    std::vector<uint8_t> raw(need);
    if (hodor_fri_produce_proof_h(p.h, iop_values.h, natural_first_element_index, raw.data(), raw.size()) != need)
        throw SynthesisError(HODOR_ERR_DEVICE, std::string("produce_proof: ") + hodor_last_error(p.F->ctx()));
    return raw;
}
inline FRIProof produce_proof(const FRIProofPrototype &p, const Polynomial<Values> &iop_values,
                              size_t natural_first_element_index)
{
    return FRIProof::from_bytes(produce_proof_bytes(p, iop_values, natural_first_element_index));
}

// src/fri/mod.rs:63-104, :156-248 + src/fri/fri_on_values.rs:11-159
struct NaiveFriIop {
    // FriIop::verify_proof -> verify_proof_queries (src/fri/mod.rs:96-102, src/fri/verifier.rs:131-289);
    // Err(..) surfaces as SynthesisError
    static bool verify_proof(const Field &F, const FRIProof &proof, size_t natural_element_index,
                             const Fr &expected_value)
    {
        std::vector<uint8_t> raw = proof.to_bytes();
        int ok = 0;
        F.check(hodor_fri_verify_proof(F.ctx(), raw.data(), raw.size(), natural_element_index, &expected_value, &ok),
                "verify_proof_queries");
        return ok != 0;
    }

    // the same verdict with the proof bound to the parameters the VERIFIER chose first (hodor_fri_verify_proof_strict):
    // what a verifier of UNTRUSTED proofs calls — the reference's walk accepts a proof whose last rounds were cut off,
    // and reads lde_factor / the degree bound from the proof itself (a prover may lower the rate)
    static bool verify_proof_strict(const Field &F, const FRIProof &proof, size_t expected_domain_size,
                                    size_t expected_lde_factor, size_t expected_output_coeffs_at_degree_plus_one,
                                    size_t natural_element_index, const Fr &expected_value)
    {
        std::vector<uint8_t> raw = proof.to_bytes();
        int ok = 0;
        F.check(hodor_fri_verify_proof_strict(F.ctx(), raw.data(), raw.size(), expected_domain_size, expected_lde_factor,
                                              expected_output_coeffs_at_degree_plus_one, natural_element_index,
                                              &expected_value, &ok), "verify_proof_queries (strict)");
        return ok != 0;
    }

    // verify_prototype (src/fri/verifier.rs:10-129) against the prover's own device-resident vectors
    static bool verify_prototype(const FRIProofPrototype &p, const Polynomial<Values> &lde_values, size_t natural_element_index)
    {
        int ok = 0;
        p.F->check(hodor_fri_verify_prototype_h(p.h, lde_values.h, natural_element_index, &ok), "verify_prototype");
        return ok != 0;
    }

    static FRIProofPrototype proof_from_lde(const Polynomial<Values> &lde_values, size_t lde_factor,
                                            size_t output_coeffs_at_degree_plus_one, int combiner = HODOR_COMBINER_TRIVIAL)
    {
        return commit(lde_values, lde_factor, output_coeffs_at_degree_plus_one, combiner, 0, "proof_from_lde_by_values");
    }
    // several commits at once — h1 and h2 of Prover::prove (src/prover/mod.rs:112-113): their latency-bound tails overlap
    // on streams of the context, one wait hands all prototypes over (hodor_fri_commit_batch_h); the prototypes are those
    // of one proof_from_lde each
    static std::vector<FRIProofPrototype> proof_from_lde_all(const std::vector<const Polynomial<Values> *> &ldes, size_t lde_factor,
                                                             size_t output_coeffs_at_degree_plus_one, int combiner = HODOR_COMBINER_TRIVIAL)
    {
        std::vector<FRIProofPrototype> out;
        if (ldes.empty()) return out;
        const Field &F = *ldes[0]->F;
        std::vector<const hodor_poly *> in;
        for (auto *l : ldes) in.push_back(l->h);
        std::vector<hodor_fri_proto *> hs(ldes.size(), nullptr);
        F.check(hodor_fri_commit_batch_h(in.data(), in.size(), lde_factor, output_coeffs_at_degree_plus_one, combiner, hs.data()),
                "proof_from_lde (batch)");
        out.resize(ldes.size());
        for (size_t i = 0; i < ldes.size(); i++) out[i].load(F, hs[i], ldes[i]->size(), lde_factor, output_coeffs_at_degree_plus_one);
        return out;
    }
    static FRIProofPrototype proof_from_lde_by_values(const Polynomial<Values> &lde_values, size_t lde_factor,
                                                      size_t output_coeffs_at_degree_plus_one)
    {
        return proof_from_lde(lde_values, lde_factor, output_coeffs_at_degree_plus_one);
    }
    // src/fri/mod.rs:156-248
    static FRIProofPrototype proof_from_lde_through_coefficients(const Polynomial<Values> &lde_values, size_t lde_factor,
                                                                 size_t output_coeffs_at_degree_plus_one)
    {
        return commit(lde_values, lde_factor, output_coeffs_at_degree_plus_one, HODOR_COMBINER_TRIVIAL, 1,
                      "proof_from_lde_through_coefficients");
    }

  private:
    static FRIProofPrototype commit(const Polynomial<Values> &lde_values, size_t lde_factor, size_t out_deg, int combiner,
                                    int through_coefficients, const char *what)
    {
        const Field &F = *lde_values.F;
        hodor_fri_proto *h = nullptr;
        F.check(hodor_fri_commit_h(lde_values.h, lde_factor, out_deg, combiner, through_coefficients, &h), what);
        FRIProofPrototype p;
        p.load(F, h, lde_values.size(), lde_factor, out_deg);
        return p;
    }
};

// Blake2sTranscript (src/transcript/mod.rs:10-80) + Verifier::bytes_to_challenge_index (src/verifier/mod.rs:246-263)
class Transcript {
  public:
    explicit Transcript(const Field &F) : F_(&F) { F.check(hodor_transcript_new(F.ctx(), &t_), "Transcript::new"); }
    ~Transcript() { hodor_transcript_free(t_); }
    Transcript(const Transcript &) = delete;
    Transcript &operator=(const Transcript &) = delete;
    void commit_bytes(const uint8_t *b, size_t n) { F_->check(hodor_transcript_commit_bytes(t_, b, n), "commit_bytes"); }
    void commit_bytes(const Hash32 &h) { commit_bytes(h.data(), h.size()); }
    void commit_field_element(const Fr &e) { F_->check(hodor_transcript_commit_field_element(t_, &e), "commit_field_element"); }
    Hash32 get_challenge_bytes()
    {
        Hash32 b(32);
        F_->check(hodor_transcript_get_challenge_bytes(t_, b.data()), "get_challenge_bytes");
        return b;
    }
    Fr get_challenge()
    {
        Fr c;
        F_->check(hodor_transcript_get_challenge(t_, &c), "get_challenge");
        return c;
    }
    static size_t bytes_to_challenge_index(const Hash32 &bytes, size_t lde_size, size_t lde_factor)
    {
        return hodor_bytes_to_challenge_index(bytes.data(), bytes.size(), lde_size, lde_factor);
    }

  private:
    const Field *F_;
    hodor_transcript *t_ = nullptr;
};

}  // namespace hodor
