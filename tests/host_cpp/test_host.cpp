// C++ restatement of the reference's own inline tests for the hot path, written against the host
// mirror (hodor_amd/csrc/host/hodor.hpp -> C ABI -> HIP kernels).  Each test names the Rust test it
// follows.  Build: g++ -O2 -std=c++17 test_host.cpp -L<repo>/hodor_amd -lhodor_gpu
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ali_instance.hpp"

using namespace hodor;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } \
    } while (0)

// XorShiftRng::from_seed([0x3dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654]) — src/fft/mod.rs:71 —
// and the Fr::rand rejection recipe of ff_derive (top limb shaved, value used as the Montgomery image).
struct XorShiftRng {
    uint32_t x = 0x3dbe6259, y = 0x8d313d76, z = 0x3237db17, w = 0xe5bc0654;
    uint32_t next_u32() { uint32_t t = x ^ (x << 11); x = y; y = z; z = w; w = w ^ (w >> 19) ^ (t ^ (t >> 8)); return w; }
    uint64_t next_u64() { uint64_t hi = next_u32(); return (hi << 32) | next_u32(); }
};

static Fr rand_fr(XorShiftRng &rng, const uint64_t p[4], int shave)
{
    for (;;) {
        Fr r;
        for (int i = 0; i < 4; i++) r.l[i] = rng.next_u64();
        r.l[3] &= ~0ull >> shave;
        bool lt = false;
        for (int i = 3; i >= 0; i--) { if (r.l[i] < p[i]) { lt = true; break; } if (r.l[i] > p[i]) break; }
        if (lt) return r;
    }
}

static const uint64_t BN256_FR[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};

// test_worker_size (src/fft/mod.rs:281-328): forward then inverse * n^-1 is the identity; omega^n == 1
static void test_fft_inverse_identity(const Field &F)
{
    XorShiftRng rng;
    const uint32_t LOG_N = 13;
    std::vector<Fr> a(1u << LOG_N);
    for (auto &v : a) v = rand_fr(rng, BN256_FR, 1);
    Domain domain = Domain::new_for_size(F, a.size());
    CHECK(F.pow(domain.generator, a.size()) == F.one());
    auto values = fft(from_coeffs(F, a));
    CHECK(values.as_ref() != a);
    auto back = ifft(std::move(values));
    CHECK(back.as_ref() == a);
    auto cv = coset_fft(from_coeffs(F, a));
    CHECK(icoset_fft(std::move(cv)).as_ref() == a);
    // any coset generator (coset_fft_for_generator / icoset_fft_for_generator, :633-638, :809-815); with the
    // field's own generator it is coset_fft
    const Fr gen = rand_fr(rng, BN256_FR, 1);
    auto gv = coset_fft_for_generator(from_coeffs(F, a), gen);
    CHECK(icoset_fft_for_generator(std::move(gv), F.inverse(gen)).as_ref() == a);
    CHECK(coset_fft_for_generator(from_coeffs(F, a), F.multiplicative_generator()) == coset_fft(from_coeffs(F, a)));
}

// test_lde_correctness / test_various_ldes (src/polynomials/mod.rs:988-1130):
// multi-coset LDE == filtering LDE == FFT of the zero-padded coefficients
static void test_lde_correctness(const Field &F)
{
    XorShiftRng rng;
    const size_t N = 1u << 10, LDE_FACTOR = 16;
    std::vector<Fr> coeffs(N);
    for (auto &v : coeffs) v = rand_fr(rng, BN256_FR, 1);
    auto poly = from_coeffs(F, coeffs);
    auto multi = lde(poly, LDE_FACTOR);
    auto filtering = filtering_lde(poly, LDE_FACTOR);
    std::vector<Fr> padded = coeffs;
    padded.resize(N * LDE_FACTOR, F.zero());
    auto naive = fft(from_coeffs(F, padded));
    CHECK(multi == filtering);
    CHECK(multi == naive);
    // coset variant evaluates on g * <Omega>: index 0 is P(g)
    auto cl = coset_lde(poly, LDE_FACTOR);
    Fr g = F.multiplicative_generator(), x = F.one(), acc = F.zero();
    for (size_t i = 0; i < N; i++) { acc = F.add(acc, F.mul(coeffs[i], x)); x = F.mul(x, g); }
    CHECK(cl.at(0) == acc);
    CHECK(coset_filtering_lde(from_coeffs(F, coeffs), LDE_FACTOR) == cl);   // :484-499 == :349
    bool threw = false;
    try { lde(poly, 3); } catch (const SynthesisError &) { threw = true; }
    CHECK(threw);
}

// make_small_iop (src/iop/blake2s_trivial_iop.rs:390-409): 64 leaves 2^i, every query verifies
static void test_make_small_iop(const Field &F)
{
    const size_t SIZE = 64;
    std::vector<Fr> inputs;
    Fr f = F.one();
    for (size_t i = 0; i < SIZE; i++) { inputs.push_back(f); f = F.add(f, f); }
    auto iop = TrivialBlake2sIOP::create(F, inputs);
    auto root = iop.get_root();
    for (size_t i = 0; i < SIZE; i++) {
        auto query = iop.query(i, inputs);
        CHECK(query.path().size() == 6);
        CHECK(TrivialBlake2sIOP::verify_query(F, query, root));
        auto bad = query;
        bad.value_ = F.add(bad.value_, F.one());
        CHECK(!TrivialBlake2sIOP::verify_query(F, bad, root));
    }
}

// make_small_iop again, over the COSET2 tree format (Coset2Combiner, hodor.hpp): every query answers for its whole
// coset with ONE path of log2(64) - 1 digests, the index maps are inverse bijections that put a coset in one leaf
static void test_make_small_iop_coset2(const Field &F)
{
    const size_t SIZE = 64;
    std::vector<Fr> inputs;
    Fr f = F.one();
    for (size_t i = 0; i < SIZE; i++) { inputs.push_back(f); f = F.add(f, f); }
    auto iop = Coset2Blake2sIOP::create(F, inputs);
    auto root = iop.get_root();
    CHECK(iop.nodes().size() == SIZE / 2 * 32);
    CHECK(!(root == TrivialBlake2sIOP::create(F, inputs).get_root()));
    for (size_t i = 0; i < SIZE; i++) {
        size_t t = Coset2Combiner::natural_index_into_tree_index(i, SIZE);
        CHECK(Coset2Combiner::tree_index_into_natural_index(t, SIZE) == i);
        auto coset = Coset2Combiner::get_coset_for_natural_index(i, SIZE);
        CHECK(coset.size() == Coset2Combiner::COSET_SIZE && (t >> 1) == coset[0]);
        auto query = iop.query(i, inputs);
        CHECK(query.index == coset[0] && query.path().size() == 5);
        CHECK(query.values_[0] == inputs[coset[0]] && query.values_[1] == inputs[coset[1]]);
        CHECK(Coset2Blake2sIOP::verify_query(F, query, root, SIZE));
        auto bad = query;
        bad.values_[1] = F.add(bad.values_[1], F.one());
        CHECK(!Coset2Blake2sIOP::verify_query(F, bad, root, SIZE));
    }
}

// test_one_fri_step (src/fri/mod.rs:252-361): coefficients 1,2,4,8, lde 4, output degree+1 = 2
static void test_one_fri_step(const Field &F)
{
    std::vector<Fr> lde_coeffs;
    Fr f = F.one();
    for (int i = 0; i < 4; i++) { lde_coeffs.push_back(f); f = F.add(f, f); }
    const size_t lde_factor = 4, output_at_degree_plus_one = 2;
    auto coeffs_poly = from_coeffs(F, lde_coeffs);
    auto lde_values = lde(coeffs_poly, lde_factor);
    auto proto = NaiveFriIop::proof_from_lde(lde_values, lde_factor, output_at_degree_plus_one);

    const size_t coset_index = 3, coset_pair_index = coset_index + lde_factor * 2;
    Fr divisor = F.pow(lde_values.omegainv, coset_index);
    Fr two_inv = F.inverse(F.from_u64(2));
    Fr challenge = proto.challenges[0];
    Fr value_at_omega = lde_values.at(coset_index), value_at_minus_omega = lde_values.at(coset_pair_index);
    Fr t0 = F.add(value_at_omega, value_at_minus_omega);
    Fr t1 = F.mul(F.mul(F.sub(value_at_omega, value_at_minus_omega), divisor), challenge);
    t0 = F.mul(F.add(t0, t1), two_inv);

    std::vector<Fr> new_coeffs;
    for (size_t i = 0; i < lde_coeffs.size(); i += 2)
        new_coeffs.push_back(F.add(F.mul(lde_coeffs[i + 1], challenge), lde_coeffs[i]));
    CHECK(proto.final_coefficients == new_coeffs);
    auto next_lde = lde(from_coeffs(F, new_coeffs), lde_factor);
    CHECK(next_lde.at(coset_index) == t0);
    CHECK(proto.num_steps() == 1);
    CHECK(proto.intermediate_values(0) == next_lde);
    // commitments are the trees of the vectors they commit to; final_root is the last root
    CHECK(proto.l0_commitment() == TrivialBlake2sIOP::create(F, lde_values));
    CHECK(proto.l0_commitment().nodes() == TrivialBlake2sIOP::create(F, lde_values).nodes());
    CHECK(proto.intermediate_commitment(0) == TrivialBlake2sIOP::create(F, next_lde));
    CHECK(proto.final_root == proto.get_roots().back());
    CHECK(proto.challenges[0] == proto.l0_commitment().get_challenge_scalar_from_root());
    // proof_from_lde_through_coefficients (:156-248): the reference's own assertion, every field equal (:338-343)
    auto by_coeffs = NaiveFriIop::proof_from_lde_through_coefficients(lde_values, lde_factor, output_at_degree_plus_one);
    CHECK(by_coeffs.final_coefficients == proto.final_coefficients);
    CHECK(by_coeffs.final_root == proto.final_root);
    CHECK(by_coeffs.intermediate_values(0) == proto.intermediate_values(0));
    CHECK(by_coeffs.challenges == proto.challenges);
    CHECK(by_coeffs.l0_commitment() == proto.l0_commitment());
    CHECK(by_coeffs.intermediate_commitment(0) == proto.intermediate_commitment(0));
    CHECK(by_coeffs.serialized() == proto.serialized());
    for (size_t i = 1; i < lde_values.size(); i += 2) CHECK(NaiveFriIop::verify_prototype(by_coeffs, lde_values, i));   // :345-349
}

// from_coeffs pads ragged input to the next power of two with zeros (src/polynomials/mod.rs:146-166);
// an empty vector becomes the size-1 zero polynomial (Domain::new_for_size(0) -> 1)
static void test_ragged_and_empty_inputs(const Field &F)
{
    XorShiftRng rng;
    std::vector<Fr> five(5);
    for (auto &v : five) v = rand_fr(rng, BN256_FR, 1);
    auto p = from_coeffs(F, five);
    CHECK(p.size() == 8 && p.exp == 3);
    CHECK(p.as_ref()[5] == F.zero() && p.at(7) == F.zero());
    std::vector<Fr> padded = five;
    padded.resize(8, F.zero());
    CHECK(fft(std::move(p)) == fft(from_coeffs(F, padded)));
    auto e = from_coeffs(F, {});
    CHECK(e.size() == 1 && e.at(0) == F.zero());
    CHECK(fft(std::move(e)).at(0) == F.zero());                // a 1-point transform is the identity
    auto one = from_coeffs(F, std::vector<Fr>{five[0]});
    CHECK(lde(one, 4).as_ref() == std::vector<Fr>(4, five[0]));  // constant polynomial
}

static void test_domain_errors(const Field &F)
{
    bool threw = false;
    try { Domain::new_for_size(F, 1ull << 33); } catch (const SynthesisError &e) { threw = (e.code == HODOR_ERR_SIZE); }
    CHECK(threw);   // S = 32: SynthesisError::Error (src/domains/mod.rs:30-32)
    auto c = Domain::coset_for_natural_index_and_size(5, 16);
    CHECK(c[0] == 5 && c[1] == 13);
    auto nx = Domain::index_and_size_for_next_domain(13, 16);
    CHECK(nx.first == 5 && nx.second == 8);
}

// test_fib_fri_iop_verifier (src/fri/mod.rs:364-507): commit down to a constant, produce_proof, then
// verify_proof with the right and a wrong expected value and with a corrupted query
static void test_fri_proof_and_verifier(const Field &F)
{
    XorShiftRng rng;
    std::vector<Fr> coeffs(64);
    for (auto &v : coeffs) v = rand_fr(rng, BN256_FR, 1);
    const size_t lde_factor = 8;
    auto lde_values = lde(from_coeffs(F, coeffs), lde_factor);
    auto proto = NaiveFriIop::proof_from_lde(lde_values, lde_factor, 1);
    CHECK(proto.num_steps() == 6 && proto.final_coefficients.size() == 1);
    for (size_t index : {size_t(1), size_t(77), lde_values.size() - 1}) {
        FRIProof proof = produce_proof(proto, lde_values, index);
        CHECK(proof.queries.size() == 2 * proof.roots.size() && proof.roots.size() == 7);
        CHECK(proof.to_bytes() == produce_proof_bytes(proto, lde_values, index));      // parse / re-encode round trip
        CHECK(NaiveFriIop::verify_proof(F, proof, index, lde_values.at(index)));
        CHECK(!NaiveFriIop::verify_proof(F, proof, index, F.add(lde_values.at(index), F.one())));
        FRIProof bad = proof;
        bad.queries[3].value_ = F.add(bad.queries[3].value_, F.one());
        CHECK(!NaiveFriIop::verify_proof(F, bad, index, lde_values.at(index)));
    }
    bool threw = false;   // Err: a point of the half-size sub-domain
    try { NaiveFriIop::verify_proof(F, produce_proof(proto, lde_values, 2), 2, lde_values.at(2)); }
    catch (const SynthesisError &) { threw = true; }
    CHECK(threw);
}

// PrecomputedOmegas::new_for_domain (src/precomputations/mod.rs:14-66) against running products
static void test_precomputed_omegas(const Field &F)
{
    Domain d = Domain::new_for_size(F, 64);
    auto t = PrecomputedOmegas::new_for_domain(F, d);
    CHECK(t.omegas.size() == 64 && t.coset.size() == 64 && t.omegas_inv.size() == 32);
    Fr w = F.one(), wi = F.one(), inv = F.inverse(d.generator);
    for (size_t i = 0; i < 64; i++) {
        CHECK(t.omegas[i] == w);
        CHECK(t.coset[i] == F.mul(w, F.multiplicative_generator()));
        if (i < 32) CHECK(t.omegas_inv[i] == wi);
        w = F.mul(w, d.generator);
        wi = F.mul(wi, inv);
    }
    CHECK(w == F.one());
}

// pad_by_factor / pad_to_size / trim_to_degree (src/polynomials/mod.rs:85-138) and the identity the
// reference's LDE tests rest on: lde(f) == fft of the coefficients padded by f (:1026-1031)
static void test_padding_helpers(const Field &F)
{
    XorShiftRng rng;
    std::vector<Fr> c(64);
    for (auto &v : c) v = rand_fr(rng, BN256_FR, 1);
    auto p = from_coeffs(F, c);
    CHECK(!p.pad_by_factor(3) && !p.pad_to_size(100) && !p.pad_to_size(32));
    CHECK(p.pad_by_factor(1) && p.size() == 64);
    CHECK(p.pad_by_factor(4) && p.size() == 256 && p.exp == 8);
    { auto v = p.as_ref(); for (size_t i = 64; i < 256; i++) CHECK(v[i] == F.zero()); }
    auto padded = fft(std::move(p));
    auto direct = lde(from_coeffs(F, c), 4);
    CHECK(padded.as_ref() == direct.as_ref());
    auto q = from_coeffs(F, c);
    CHECK(q.pad_to_size(128) && q.size() == 128 && q.exp == 7);
    q.trim_to_degree(9);
    { auto v = q.as_ref(); for (size_t i = 0; i < 128; i++) CHECK(i < 10 ? v[i] == c[i] : v[i] == F.zero()); }
}


// The value-form surface ALI calls (src/polynomials/mod.rs:54-83, :640-711, :744-771, :817-954) on device-resident
// polynomials, each against the same operation done element by element on the host field
static void test_value_form_methods(const Field &F)
{
    XorShiftRng rng;
    const size_t N = 1u << 9;
    std::vector<Fr> a(N), b(N);
    for (auto &v : a) v = rand_fr(rng, BN256_FR, 1);
    for (auto &v : b) v = rand_fr(rng, BN256_FR, 1);
    const Fr s = rand_fr(rng, BN256_FR, 1);
    auto expect = [&](const Polynomial<Values> &p, auto fn) {
        auto v = p.as_ref();
        for (size_t i = 0; i < N; i++) CHECK(v[i] == fn(i));
    };
    auto pa = from_values(F, a), pb = from_values(F, b);
    { auto t = pa.clone(); t.add_assign(pb); expect(t, [&](size_t i) { return F.add(a[i], b[i]); }); }
    { auto t = pa.clone(); t.sub_assign(pb); expect(t, [&](size_t i) { return F.sub(a[i], b[i]); }); }
    { auto t = pa.clone(); t.mul_assign(pb); expect(t, [&](size_t i) { return F.mul(a[i], b[i]); }); }
    { auto t = pa.clone(); t.add_assign_scaled(pb, s); expect(t, [&](size_t i) { return F.add(a[i], F.mul(b[i], s)); }); }
    { auto t = pa.clone(); t.scale(s); expect(t, [&](size_t i) { return F.mul(a[i], s); }); }
    { auto t = pa.clone(); t.negate(); expect(t, [&](size_t i) { return F.negate(a[i]); }); }
    { auto t = pa.clone(); t.square(); expect(t, [&](size_t i) { return F.mul(a[i], a[i]); }); }
    { auto t = pa.clone(); t.pow(5); expect(t, [&](size_t i) { return F.pow(a[i], 5); }); }
    { auto t = pa.clone(); t.pow(2); expect(t, [&](size_t i) { return F.mul(a[i], a[i]); }); }    // :746-748
    { auto t = pa.clone(); t.add_constant(s); expect(t, [&](size_t i) { return F.add(a[i], s); }); }
    { auto t = pa.clone(); t.distribute_powers(s); Fr x = F.one(); auto v = t.as_ref();
      for (size_t i = 0; i < N; i++) { CHECK(v[i] == F.mul(a[i], x)); x = F.mul(x, s); } }
    {   // test_batch_inversion (:959-985)
        auto t = pa.clone();
        CHECK(t.batch_inversion());
        expect(t, [&](size_t i) { return F.inverse(a[i]); });
        auto z = pa.clone();
        z.set(17, F.zero());
        CHECK(!z.batch_inversion());                  // Err(SynthesisError::Error), data untouched (:909)
        CHECK(z.at(16) == a[16] && z.at(17) == F.zero());
    }
    {   // evaluate_at (:685-711) and the degree-one divisor of calculate_deep (deep.rs:58-72)
        auto c = from_coeffs(F, a);
        Fr x = F.one(), acc = F.zero();
        for (size_t i = 0; i < N; i++) { acc = F.add(acc, F.mul(a[i], x)); x = F.mul(x, s); }
        CHECK(c.evaluate_at(s) == acc);
        auto q = Polynomial<Coefficients>::new_for_size(F, 2);
        q.set(1, F.one());                            // q_poly.as_mut()[1] = F::one()        deep.rs:61
        q.sub_assign_at(0, s);                        // q_poly.as_mut()[0].sub_assign(&root) deep.rs:62
        const uint64_t before = F.host_round_trips();
        auto d = evaluate_at_domain_for_degree_one(q, N);
        CHECK(F.host_round_trips() == before);        // built on the host, evaluated on the device: nothing came back
        Domain dom = Domain::new_for_size(F, N);
        Fr u = F.one();
        auto v = d.as_ref();
        for (size_t i = 0; i < N; i++) { CHECK(v[i] == F.sub(u, s)); u = F.mul(u, dom.generator); }
        auto dc = evaluate_at_domain_for_degree_one(q, N, true);
        CHECK(dc.at(0) == F.sub(F.multiplicative_generator(), s));
    }
    {   // Coefficients: the other operand may be shorter (:641), Values: sizes must agree (:818)
        auto big = from_coeffs(F, a);
        std::vector<Fr> small_v(b.begin(), b.begin() + 64);
        auto small_p = from_coeffs(F, small_v);
        big.add_assign(small_p);
        auto v = big.as_ref();
        for (size_t i = 0; i < N; i++) CHECK(v[i] == (i < 64 ? F.add(a[i], b[i]) : a[i]));
        bool threw = false;
        try { small_p.add_assign(big); } catch (const SynthesisError &) { threw = true; }
        CHECK(threw);
        auto vs = from_values(F, small_v);
        threw = false;
        try { pa.add_assign(vs); } catch (const SynthesisError &) { threw = true; }
        CHECK(threw);
    }
    {   // from_roots (:168-227) against the product of the linear factors evaluated at a point
        std::vector<Fr> roots(b.begin(), b.begin() + 13);
        auto z = from_roots(F, roots);
        CHECK(z.size() == 16);
        Fr prod = F.one();
        for (auto &r : roots) prod = F.mul(prod, F.sub(s, r));
        CHECK(z.evaluate_at(s) == prod);
        CHECK(z.evaluate_at(roots[5]) == F.zero());
    }
    {   // the fused quotient term == the five passes it replaces (deep.rs:74-84)
        auto inv = pb.clone();
        CHECK(inv.batch_inversion());
        auto t = pa.clone();
        t.add_constant(F.negate(a[3]));
        t.scale(s);
        t.mul_assign(inv);
        auto acc = Polynomial<Values>::new_for_size(F, N);
        auto sum = acc.clone();
        sum.add_assign(t);
        quotient_term(acc, pa, inv, a[3], &s, true);
        CHECK(acc == sum);
    }
}

// Prover::prove's use of the types (src/prover/mod.rs:73-95, :142-151): every register's LDE and oracle in one call
// each, roots in one wait, queries from the device-resident oracles; equal to the one-by-one calls
static void test_batched_ldes_and_oracles(const Field &F)
{
    XorShiftRng rng;
    const size_t N = 1u << 8, LDE_FACTOR = 8, REGS = 3;
    std::vector<Polynomial<Coefficients>> ws;
    for (size_t r = 0; r < REGS; r++) {
        std::vector<Fr> c(N);
        for (auto &v : c) v = rand_fr(rng, BN256_FR, 1);
        ws.push_back(from_coeffs(F, c));
    }
    auto f_ldes = lde_all(ws, LDE_FACTOR);
    auto f_oracles = TrivialBlake2sIOP::create_all(F, f_ldes);
    auto roots = TrivialBlake2sIOP::get_roots(F, f_oracles);
    for (size_t r = 0; r < REGS; r++) {
        auto single = lde(ws[r], LDE_FACTOR);
        CHECK(f_ldes[r] == single);
        auto tree = TrivialBlake2sIOP::create(F, single);
        CHECK(tree.get_root() == roots[r] && tree.nodes() == f_oracles[r].nodes());
        auto q = f_oracles[r].query(77, f_ldes[r]);
        CHECK(q.value() == single.at(77) && TrivialBlake2sIOP::verify_query(F, q, roots[r]));
    }
    // an in-place method on one of the views leaves its siblings alone
    auto keep = f_ldes[1].clone();
    f_ldes[0].scale(F.from_u64(3));
    auto back = ifft(std::move(f_ldes[2]));
    CHECK(back.size() == N * LDE_FACTOR && f_ldes[1] == keep);
}

// The multi-GPU building blocks from compiled host code (what a Rust process per GPU binds): P = 2 ranks
// played one after the other on the one device, the all-to-all done with plain device copies.
// forward A -> columns -> exchange -> rows -> B, B transposed back == the single-device transform
// (parallel_fft, src/fft/fft.rs:68-124, is the same split on one machine); inverse returns A.
static void test_sixstep_two_ranks(const Field &F)
{
    const uint32_t log_n = 10, log_n1 = 5, log_n2 = 5, log_p = 1;
    const size_t n = (size_t)1 << log_n, P = 2, m = n / P, N1 = 32, N2 = 32, r1 = N1 / P, c2 = N2 / P;
    XorShiftRng rng;
    std::vector<Fr> x(n);
    for (auto &v : x) v = rand_fr(rng, BN256_FR, 1);
    Domain d = Domain::new_for_size(F, n);
    std::vector<Fr> spec = x;
    F.check(hodor_fft(F.ctx(), spec.data(), n, &d.generator, log_n), "fft");
    Fr *dev[2][4];
    for (size_t q = 0; q < P; q++)
        for (int b = 0; b < 4; b++) F.check(hodor_buf_alloc(F.ctx(), m * sizeof(Fr), (void **)&dev[q][b]), "alloc");
    for (size_t q = 0; q < P; q++) {   // layout A: column block q of the N1 x N2 matrix
        std::vector<Fr> a(m);
        for (size_t i = 0; i < N1; i++)
            for (size_t j = 0; j < c2; j++) a[i * c2 + j] = x[i * N2 + q * c2 + j];
        F.check(hodor_buf_upload(F.ctx(), dev[q][0], a.data(), m * sizeof(Fr)), "upload");
        F.check(hodor_sixstep_columns_dev(F.ctx(), nullptr, dev[q][0], dev[q][1], log_n1, log_n2, log_p, (uint32_t)q,
                                          &d.generator, 0, 0, 0), "columns");
    }
    F.check(hodor_ctx_synchronize(F.ctx()), "sync");
    auto exchange = [&](int from, int to) {   // slab s of rank t's receive buffer = slab t of rank s's send buffer
        std::vector<Fr> h(m);
        const size_t slab = m / P;
        for (size_t t = 0; t < P; t++)
            for (size_t s_ = 0; s_ < P; s_++) {
                F.check(hodor_buf_download(F.ctx(), h.data(), dev[s_][from] + t * slab, slab * sizeof(Fr)), "dl");
                F.check(hodor_buf_upload(F.ctx(), dev[t][to] + s_ * slab, h.data(), slab * sizeof(Fr)), "ul");
            }
    };
    exchange(1, 2);
    for (size_t q = 0; q < P; q++) {
        F.check(hodor_sixstep_rows_dev(F.ctx(), nullptr, dev[q][2], dev[q][3], log_n1, log_n2, log_p, (uint32_t)q,
                                       &d.generator, 0, 0, 0), "rows");
        std::vector<Fr> b(m);   // layout B: b[i][k2] = X[(q*r1 + i) + N1*k2]
        F.check(hodor_ctx_synchronize(F.ctx()), "sync");
        F.check(hodor_buf_download(F.ctx(), b.data(), dev[q][3], m * sizeof(Fr)), "dl");
        for (size_t i = 0; i < r1; i++)
            for (size_t k2 = 0; k2 < N2; k2++) CHECK(b[i * N2 + k2] == spec[(q * r1 + i) + N1 * k2]);
    }
    for (size_t q = 0; q < P; q++)
        F.check(hodor_sixstep_rows_dev(F.ctx(), nullptr, dev[q][3], dev[q][1], log_n1, log_n2, log_p, (uint32_t)q,
                                       &d.generator, 1, 0, 0), "rows^-1");
    F.check(hodor_ctx_synchronize(F.ctx()), "sync");
    exchange(1, 2);
    for (size_t q = 0; q < P; q++) {
        F.check(hodor_sixstep_columns_dev(F.ctx(), nullptr, dev[q][2], dev[q][3], log_n1, log_n2, log_p, (uint32_t)q,
                                          &d.generator, 1, 0, 0), "columns^-1");
        std::vector<Fr> a(m);
        F.check(hodor_ctx_synchronize(F.ctx()), "sync");
        F.check(hodor_buf_download(F.ctx(), a.data(), dev[q][3], m * sizeof(Fr)), "dl");
        for (size_t i = 0; i < N1; i++)
            for (size_t j = 0; j < c2; j++) CHECK(a[i * c2 + j] == x[i * N2 + q * c2 + j]);
    }
    for (size_t q = 0; q < P; q++)
        for (int b = 0; b < 4; b++) hodor_buf_free(F.ctx(), dev[q][b]);
}

// Polynomial::as_mut() (src/polynomials/mod.rs:46) for the whole slice and ALIInstance::from_arp's divisor precompute
// (src/ali/per_register/mod.rs:36-244) AS WRITTEN on top of it — worker.scope + as_mut().chunks_mut(), batch_inversion,
// as_mut().chunks_mut() again — against (a) the closed form evaluated on the host element by element and (b) the
// device-resident form of the same vectors (hodor::dense_divisor_on_coset); an instance whose trace fills only part of
// its column domain (num_rows < column size: several roots) and constraints that start late (start_at > 0) included.
static void test_as_mut_and_ali_divisors(const Field &F)
{
    Worker worker(7);   // an odd worker count: chunks that do not divide the vector evenly (get_chunk_size, multicore.rs:76-87)
    CHECK(worker.get_chunk_size(3) == 1 && worker.get_chunk_size(64) == 9 && worker.log_num_cpus() == 2);
    {   // the guard: the image is the vector, the write-back happens when it goes out of scope
        auto p = Polynomial<Values>::new_for_size(F, 64);
        F.reset_host_round_trips();
        {
            MutSlice s = p.as_mut();
            CHECK(s.size() == 64);
            for (size_t i = 0; i < 64; i++) s[i] = F.from_u64(i + 1);
            CHECK(p.at(5) == F.from_u64(6));              // as_ref()[5] during the borrow reads the image
        }
        CHECK(F.host_round_trips() == 0);                 // new_for_size's zeros are not downloaded
        p.square();
        for (size_t i = 0; i < 64; i += 9) CHECK(p.at(i) == F.from_u64((i + 1) * (i + 1)));
        p.as_mut()[3] = F.one();                          // the temporary guard: `p.as_mut()[3] = F::one();`
        auto q = p.clone();
        CHECK(q.at(3) == F.one() && q.at(4) == F.from_u64(25) && p == q);
    }
    struct Case { uint64_t num_rows; ali::DenseConstraint dc; };
    const Case cases[] = {{64, {0, 1}}, {64, {2, 3}}, {50, {1, 2}}, {1024, {0, 1}}};
    for (const Case &c : cases) {
        Domain column = Domain::new_for_size(F, c.num_rows);
        Domain evaluation = Domain::new_for_size(F, column.size * ali::MAX_CONSTRAINT_POWER);
        auto written = ali::inverse_divisor_for_dense_constraint_in_coset(F, column, evaluation, c.dc, c.num_rows, worker);
        CHECK(written.second == column.size - c.dc.start_at - (column.size - c.num_rows) - c.dc.span);
        const std::vector<Fr> roots = ali::dense_constraint_roots(F, column, c.dc, c.num_rows);
        CHECK(roots.size() == c.dc.start_at + (column.size - (c.num_rows - c.dc.span)));
        auto resident = dense_divisor_on_coset(F, (size_t)evaluation.size, (size_t)column.size, roots);
        CHECK(written.first == resident);
        Slice got = written.first.as_ref();
        Fr x = F.multiplicative_generator();
        for (size_t i = 0; i < got.size(); i++) {         // prod (x - root) / (x^T - 1), one element at a time
            Fr d = F.inverse(F.sub(F.pow(x, column.size), F.one()));
            for (const Fr &r : roots) d = F.mul(d, F.sub(x, r));
            CHECK(got[i] == d);
            x = F.mul(x, evaluation.generator);
        }
    }
    for (bool resident : {false, true}) {                 // from_arp, both forms: the same instance
        auto I = ali::ALIInstance::from_arp(F, 256, worker, resident);
        auto J = ali::ALIInstance::from_arp(F, 256, Worker(3), !resident);
        CHECK(I.constraint_divisors == J.constraint_divisors);
        CHECK(I.boundary_constraint_divisors.at(0) == J.boundary_constraint_divisors.at(0));
        const Fr alpha = F.from_u64(77), beta = F.from_u64(5);
        CHECK(I.calculate_adjustment_polynomial_in_coset(F, 3, alpha, beta) == J.calculate_adjustment_polynomial_in_coset(F, 3, alpha, beta));
        Fr x = F.multiplicative_generator();              // boundary divisor of row 0: 1 / (x - 1)
        Slice b = I.boundary_constraint_divisors.at(0).as_ref();
        for (size_t i = 0; i < b.size(); i += 37) {
            CHECK(F.mul(b[i], F.sub(F.mul(F.pow(I.constraints_domain.generator, i), x), F.one())) == F.one());
        }
    }
    // no vanishing value on the coset is ever met with the field's generator; a zero in the vector is the reference's Err
    auto z = Polynomial<Values>::new_for_size(F, 8);
    CHECK(!z.batch_inversion());
}

int main()
{
    Field F(BN256_FR, 7, 0);
    CHECK(F.S() == 32 && F.capacity() == 254);
    test_domain_errors(F);
    test_ragged_and_empty_inputs(F);
    test_precomputed_omegas(F);
    test_fft_inverse_identity(F);
    test_lde_correctness(F);
    test_make_small_iop(F);
    test_make_small_iop_coset2(F);
    test_one_fri_step(F);
    test_fri_proof_and_verifier(F);
    test_sixstep_two_ranks(F);
    test_padding_helpers(F);
    test_value_form_methods(F);
    test_batched_ldes_and_oracles(F);
    test_as_mut_and_ali_divisors(F);
    printf("host_cpp: all tests passed\n");
    return 0;
}
