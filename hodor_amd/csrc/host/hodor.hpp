// hodor.hpp — C++ host-side mirror of the reference's L3 interface for the hot path, written on top
// of the C ABI only (include/hodor_gpu.h).  Same names, argument meaning and error behaviour as the
// Rust items they stand for, so host code (and tests) read like the reference's:
//
//   hodor::Field                    the `F: PrimeField` type parameter (one hodor_ctx)
//   hodor::Domain                   src/domains/mod.rs:14-71
//   hodor::Polynomial<Form>         src/polynomials/mod.rs:26-34, :139-712 (Coefficients), :715-955 (Values)
//   hodor::Blake2sIopTree           src/iop/blake2s_trivial_iop.rs:106-280
//   hodor::TrivialBlake2sIOP        src/iop/blake2s_trivial_iop.rs:282-339 (+ Query :341-375)
//   hodor::NaiveFriIop              src/fri/mod.rs:63-117, src/fri/fri_on_values.rs:11-159
//
// Errors: SynthesisError::Error and the reference's asserts become hodor::SynthesisError exceptions
// (thrown on this side of the ABI; the ABI itself returns status codes).
#pragma once
#include <stdexcept>
#include <string>
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

  private:
    hodor_ctx *ctx_ = nullptr;
    hodor_field_info info_;
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

struct Coefficients {};
struct Values {};

// src/polynomials/mod.rs:26-34
template <class Form>
class Polynomial {
  public:
    const Field *F;
    std::vector<Fr> coeffs;
    uint32_t exp;
    Fr omega, omegainv, geninv, minv;

    size_t size() const { return coeffs.size(); }
    const std::vector<Fr> &as_ref() const { return coeffs; }
    std::vector<Fr> into_coeffs() && { return std::move(coeffs); }

    // from_coeffs / from_values: pad to a power of two and cache domain constants (:146-166, :722-742)
    static Polynomial from_vec(const Field &F, std::vector<Fr> v)
    {
        Polynomial p;
        p.F = &F;
        Domain d = Domain::new_for_size(F, v.size());
        v.resize(d.size, F.zero());
        p.coeffs = std::move(v);
        p.exp = (uint32_t)d.power_of_two;
        p.omega = d.generator;
        p.omegainv = F.inverse(d.generator);
        p.geninv = F.inverse(F.multiplicative_generator());
        p.minv = F.inverse(F.from_u64(d.size));
        return p;
    }

    // pad_by_factor / pad_to_size / trim_to_degree (:85-138): host-side bookkeeping, the vector stays a
    // plain `Vec<F>`; `false` stands for Err(SynthesisError::Error)
    void refresh_domain()
    {
        Domain d = Domain::new_for_size(*F, coeffs.size());
        exp = (uint32_t)d.power_of_two;
        omega = d.generator;
        omegainv = F->inverse(d.generator);
        minv = F->inverse(F->from_u64(d.size));
    }
    bool pad_by_factor(size_t factor)
    {
        if (factor == 1) return true;
        if (factor == 0 || (factor & (factor - 1))) return false;
        coeffs.resize(coeffs.size() * factor, F->zero());
        refresh_domain();
        return true;
    }
    bool pad_to_size(size_t new_size)
    {
        if (new_size < coeffs.size() || new_size == 0 || (new_size & (new_size - 1))) return false;
        coeffs.resize(new_size, F->zero());
        refresh_domain();
        return true;
    }
    void trim_to_degree(size_t degree)
    {
        const size_t size = coeffs.size();
        if (size <= degree + 1) return;
        coeffs.resize(degree + 1);
        coeffs.resize(size, F->zero());
    }

    void distribute_powers(const Fr &g)   // :55-58 -> src/fft/mod.rs:110
    {
        F->check(hodor_distribute_powers(F->ctx(), coeffs.data(), coeffs.size(), &g), "distribute_powers");
    }

    template <class To>
    Polynomial<To> retype() &&
    {
        Polynomial<To> q;
        q.F = F; q.coeffs = std::move(coeffs); q.exp = exp; q.omega = omega; q.omegainv = omegainv;
        q.geninv = geninv; q.minv = minv;
        return q;
    }
};

inline Polynomial<Coefficients> from_coeffs(const Field &F, std::vector<Fr> c)
{
    return Polynomial<Coefficients>::from_vec(F, std::move(c));
}
inline Polynomial<Values> from_values(const Field &F, std::vector<Fr> v)
{
    return Polynomial<Values>::from_vec(F, std::move(v));
}

// Polynomial<F, Coefficients>::fft / coset_fft (:611-631)
inline Polynomial<Values> fft(Polynomial<Coefficients> p)
{
    p.F->check(hodor_fft(p.F->ctx(), p.coeffs.data(), p.coeffs.size(), &p.omega, p.exp), "fft");
    return std::move(p).retype<Values>();
}
inline Polynomial<Values> coset_fft(Polynomial<Coefficients> p)
{
    p.F->check(hodor_poly_coset_fft(p.F->ctx(), p.coeffs.data(), p.coeffs.size()), "coset_fft");
    return std::move(p).retype<Values>();
}
// coset_fft_for_generator (:633-638)
inline Polynomial<Values> coset_fft_for_generator(Polynomial<Coefficients> p, const Fr &gen)
{
    p.F->check(hodor_poly_coset_fft_for_generator(p.F->ctx(), p.coeffs.data(), p.coeffs.size(), &gen),
               "coset_fft_for_generator");
    return std::move(p).retype<Values>();
}
// Polynomial<F, Values>::ifft / icoset_fft (:773-807)
inline Polynomial<Coefficients> ifft(Polynomial<Values> p)
{
    p.F->check(hodor_poly_ifft(p.F->ctx(), p.coeffs.data(), p.coeffs.size()), "ifft");
    return std::move(p).retype<Coefficients>();
}
inline Polynomial<Coefficients> icoset_fft(Polynomial<Values> p)
{
    p.F->check(hodor_poly_icoset_fft(p.F->ctx(), p.coeffs.data(), p.coeffs.size()), "icoset_fft");
    return std::move(p).retype<Coefficients>();
}
// icoset_fft_for_generator (:809-815): `geninv` is the inverse of the coset generator, as in the reference
inline Polynomial<Coefficients> icoset_fft_for_generator(Polynomial<Values> p, const Fr &geninv)
{
    p.F->check(hodor_poly_icoset_fft_for_generator(p.F->ctx(), p.coeffs.data(), p.coeffs.size(), &geninv),
               "icoset_fft_for_generator");
    return std::move(p).retype<Coefficients>();
}
// lde / coset_lde (:343-349 -> :418-482, :544-609)
inline Polynomial<Values> lde_impl(const Polynomial<Coefficients> &p, size_t factor, bool coset)
{
    if (factor == 0 || (factor & (factor - 1)))
        throw SynthesisError(HODOR_ERR_SIZE, "lde factor must be a power of two");   // assert!(factor.is_power_of_two())
    std::vector<Fr> out(p.coeffs.size() * factor);
    int rc = coset ? hodor_poly_coset_lde(p.F->ctx(), p.coeffs.data(), p.coeffs.size(), factor, out.data())
                   : hodor_poly_lde(p.F->ctx(), p.coeffs.data(), p.coeffs.size(), factor, out.data());
    p.F->check(rc, "lde");
    return from_values(*p.F, std::move(out));
}
inline Polynomial<Values> lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, false); }
inline Polynomial<Values> coset_lde(const Polynomial<Coefficients> &p, size_t factor) { return lde_impl(p, factor, true); }
// filtering_lde (:355-368): zero-pad then best_lde
inline Polynomial<Values> filtering_lde(const Polynomial<Coefficients> &p, size_t factor)
{
    std::vector<Fr> v = p.coeffs;
    v.resize(p.coeffs.size() * factor, p.F->zero());
    Domain d = Domain::new_for_size(*p.F, v.size());
    p.F->check(hodor_lde(p.F->ctx(), v.data(), v.size(), &d.generator, (uint32_t)d.power_of_two, factor),
               "best_lde");
    return from_values(*p.F, std::move(v));
}

// coset_filtering_lde (:484-499): distribute_powers(multiplicative_generator), zero-pad, best_lde
inline Polynomial<Values> coset_filtering_lde(Polynomial<Coefficients> p, size_t factor)
{
    if (factor == 1) return coset_fft(std::move(p));
    p.distribute_powers(p.F->multiplicative_generator());
    return filtering_lde(p, factor);
}

typedef std::vector<uint8_t> Hash32;   // [u8; 32]

// src/iop/blake2s_trivial_iop.rs:106-280
class Blake2sIopTree {
  public:
    const Field *F;
    uint64_t size_;
    std::vector<uint8_t> nodes;   // size * 32, heap layout, root at [32, 64)

    static Blake2sIopTree create(const Field &F, const std::vector<Fr> &leafs)
    {
        Blake2sIopTree t;
        t.F = &F;
        t.size_ = leafs.size();
        t.nodes.assign(leafs.size() * 32, 0);
        F.check(hodor_iop_create(F.ctx(), leafs.data(), leafs.size(), t.nodes.data()), "IopTree::create");
        return t;
    }
    uint64_t size() const { return size_; }
    Hash32 get_root() const { return Hash32(nodes.begin() + 32, nodes.begin() + 64); }
    static Fr encode_root_into_challenge(const Field &F, const Hash32 &root)
    {
        Fr r;
        F.check(hodor_iop_challenge(F.ctx(), root.data(), &r), "interpret_hash");
        return r;
    }
    Fr get_challenge_scalar_from_root() const { return encode_root_into_challenge(*F, get_root()); }
    std::vector<Hash32> get_path(size_t tree_index, const std::vector<Fr> &leafs_values) const
    {
        std::vector<uint8_t> buf(32 * 64);
        size_t cnt = 0;
        F->check(hodor_iop_path(F->ctx(), nodes.data(), leafs_values.data(), leafs_values.size(), tree_index,
                                buf.data(), &cnt), "get_path");
        std::vector<Hash32> path;
        for (size_t i = 0; i < cnt; i++) path.emplace_back(buf.begin() + 32 * i, buf.begin() + 32 * (i + 1));
        return path;
    }
    static bool verify(const Field &F, const Hash32 &root, const Fr &leaf_value, const std::vector<Hash32> &path,
                       size_t tree_index)
    {
        std::vector<uint8_t> flat;
        for (auto &h : path) flat.insert(flat.end(), h.begin(), h.end());
        int ok = 0;
        F.check(hodor_iop_verify(F.ctx(), root.data(), &leaf_value, flat.data(), path.size(), tree_index, &ok),
                "verify");
        return ok != 0;
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

// src/iop/blake2s_trivial_iop.rs:282-339
class TrivialBlake2sIOP {
  public:
    Blake2sIopTree tree;
    static TrivialBlake2sIOP create(const Field &F, const std::vector<Fr> &leafs)
    {
        return TrivialBlake2sIOP{Blake2sIopTree::create(F, leafs)};
    }
    Hash32 get_root() const { return tree.get_root(); }
    Fr get_challenge_scalar_from_root() const { return tree.get_challenge_scalar_from_root(); }
    TrivialBlake2sIopQuery query(size_t natural_index, const std::vector<Fr> &leafs) const
    {
        if (natural_index >= tree.size() || natural_index >= leafs.size())
            throw SynthesisError(HODOR_ERR_SIZE, "query index out of range");   // asserts :325-326
        return TrivialBlake2sIopQuery{natural_index, leafs[natural_index], tree.get_path(natural_index, leafs)};
    }
    static bool verify_query(const Field &F, const TrivialBlake2sIopQuery &q, const Hash32 &root)
    {
        return Blake2sIopTree::verify(F, root, q.value(), q.path(), q.tree_index());
    }
    bool operator==(const TrivialBlake2sIOP &o) const { return get_root() == o.get_root(); }
};

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

// the IOP over a COSET2 tree: a query answers for the whole coset (both values, ONE path of log2(n) - 1 digests)
struct Coset2Blake2sIopQuery {
    size_t index;                 // the smaller member of the coset = the leaf index
    Fr values_[2];                // value[index], value[index + n/2]
    std::vector<Hash32> path_;
    size_t natural_index() const { return index; }
    const std::vector<Hash32> &path() const { return path_; }
};
class Coset2Blake2sIOP {
  public:
    const Field *F;
    uint64_t size_;               // number of committed VALUES (the tree has size_/2 leaves)
    std::vector<uint8_t> nodes;   // (size_/2) * 32, heap layout, root at [32, 64)
    static Coset2Blake2sIOP create(const Field &F, const std::vector<Fr> &values)
    {
        Coset2Blake2sIOP t{&F, values.size(), std::vector<uint8_t>(values.size() * 16, 0)};
        F.check(hodor_iop_create_combined(F.ctx(), values.data(), values.size(), HODOR_COMBINER_COSET2, t.nodes.data()),
                "IopTree::create (COSET2)");
        return t;
    }
    Hash32 get_root() const { return Hash32(nodes.begin() + 32, nodes.begin() + 64); }
    Coset2Blake2sIopQuery query(size_t natural_index, const std::vector<Fr> &values) const
    {
        if (natural_index >= size_ || values.size() != size_) throw SynthesisError(HODOR_ERR_SIZE, "query index out of range");
        const size_t half = size_ / 2, k = natural_index % half;
        Coset2Blake2sIopQuery q{k, {values[k], values[k + half]}, {}};
        std::vector<uint8_t> buf(32 * 64);
        size_t cnt = 0;
        F->check(hodor_iop_path_combined(F->ctx(), nodes.data(), values.data(), values.size(), HODOR_COMBINER_COSET2,
                                         natural_index, buf.data(), &cnt), "get_path (COSET2)");
        for (size_t i = 0; i < cnt; i++) q.path_.emplace_back(buf.begin() + 32 * i, buf.begin() + 32 * (i + 1));
        return q;
    }
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

// src/fri/mod.rs:106-117 — field for field
struct FRIProofPrototype {
    TrivialBlake2sIOP l0_commitment;
    std::vector<TrivialBlake2sIOP> intermediate_commitments;
    std::vector<Polynomial<Values>> intermediate_values;
    std::vector<Fr> challenges;
    Hash32 final_root;
    std::vector<Fr> final_coefficients;
    size_t initial_degree_plus_one, output_coeffs_at_degree_plus_one, lde_factor;

    std::vector<Hash32> get_roots() const   // :120-128
    {
        std::vector<Hash32> r{l0_commitment.get_root()};
        for (auto &c : intermediate_commitments) r.push_back(c.get_root());
        return r;
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
};

// FRIProofPrototype::produce_proof, src/fri/query_producer.rs:10-53 (host-resident prototype)
inline FRIProof produce_proof(const FRIProofPrototype &p, const Polynomial<Values> &iop_values,
                              size_t natural_first_element_index)
{
    FRIProof proof{{}, {}, p.final_coefficients, p.initial_degree_plus_one, p.output_coeffs_at_degree_plus_one,
                   p.lde_factor};
    size_t domain_size = p.initial_degree_plus_one * p.lde_factor, domain_idx = natural_first_element_index;
    for (size_t r = 0; r <= p.intermediate_commitments.size(); r++) {
        const TrivialBlake2sIOP &iop = r == 0 ? p.l0_commitment : p.intermediate_commitments[r - 1];
        const std::vector<Fr> &leafs = r == 0 ? iop_values.coeffs : p.intermediate_values[r - 1].coeffs;
        for (size_t idx : Domain::coset_for_natural_index_and_size(domain_idx, domain_size))
            proof.queries.push_back(iop.query(idx, leafs));
        proof.roots.push_back(iop.get_root());
        auto nx = Domain::index_and_size_for_next_domain(domain_idx, domain_size);
        domain_idx = nx.first;
        domain_size = nx.second;
    }
    return proof;
}

// src/fri/mod.rs:63-104 + src/fri/fri_on_values.rs:11-159
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

    static FRIProofPrototype proof_from_lde(const Polynomial<Values> &lde_values, size_t lde_factor,
                                            size_t output_coeffs_at_degree_plus_one)
    {
        const Field &F = *lde_values.F;
        hodor_fri_proto *h = nullptr;
        F.check(hodor_fri_commit(F.ctx(), lde_values.coeffs.data(), lde_values.size(), lde_factor,
                                 output_coeffs_at_degree_plus_one, &h), "proof_from_lde_by_values");
        size_t steps = hodor_fri_num_steps(h), n = lde_values.size();
        auto tree_of = [&](int step, size_t sz) {
            Blake2sIopTree t;
            t.F = &F;
            t.size_ = sz;
            t.nodes.resize(sz * 32);
            F.check(hodor_fri_tree_nodes(h, step, t.nodes.data()), "fri tree");
            return TrivialBlake2sIOP{t};
        };
        FRIProofPrototype p{tree_of(-1, n), {}, {}, std::vector<Fr>(steps), Hash32(32),
                            std::vector<Fr>(output_coeffs_at_degree_plus_one), n / lde_factor,
                            output_coeffs_at_degree_plus_one, lde_factor};
        for (size_t i = 0; i < steps; i++) {
            size_t sz = n >> (i + 1);
            p.intermediate_commitments.push_back(tree_of((int)i, sz));
            std::vector<Fr> v(sz);
            F.check(hodor_fri_intermediate_values(h, i, v.data()), "fri values");
            p.intermediate_values.push_back(from_values(F, std::move(v)));
        }
        hodor_fri_challenges(h, p.challenges.data());
        hodor_fri_final_root(h, p.final_root.data());
        hodor_fri_final_coefficients(h, p.final_coefficients.data());
        hodor_fri_free(h);
        return p;
    }
};

}  // namespace hodor
