// prove_shape.cpp — the phases of the reference's cubic-VDF proof run
// (/root/reference/src/experiments/cubic_vdf.rs:288-354, the order of Prover::prove, src/prover/mod.rs:66-174) written
// in C++ against hodor_amd/csrc/host/hodor.hpp ONLY: Polynomial / TrivialBlake2sIOP / NaiveFriIop / Transcript objects
// and their methods, as the Rust layers above the boundary call them — no `_dev` entry point, no device pointer, no
// stream in this file.  Every polynomial, LDE, oracle and FRI vector lives in HBM behind those objects; what comes
// back to the host are the roots, the evaluations at z, the prototypes' roots / final coefficients and the query
// answers, and the library counts those round trips.
//
// The synthetic trace and constraint system are those of tests/prove_shape_ref.py (same SplitMix64 streams, same
// operation order: tests/ali_replay_ref.py = from_arp + calculate_g, src/ali/per_register/mod.rs:36-526;
// tests/deep_replay_ref.py = calculate_deep, src/ali/per_register/deep.rs:14-146), so the proof bytes written here must
// equal the bytes the CPU oracle assembles for the same shape (tests/test_gpu_prove_shape.py, bench/prove_shape.py).
// Since round 6 the value-form inputs of calculate_g are the REAL ones: ALIInstance::from_arp's inverse divisors,
// boundary divisors and adjustment polynomials (tests/host_cpp/ali_instance.hpp), AS WRITTEN in the reference through
// Polynomial::as_mut() (ali_mode 0) or device-resident (ali_mode 1) — same proof bytes either way.
//
//   prove_shape <log_rows> <registers> <lde_factor> <combiner 0|1> <out.bin> [reps=1] [sync_phases=0] [ali_mode=0] [fri_batch=1]
// prints one JSON line: total / per-phase milliseconds (median run), host round trips and PCIe bytes of one proof and of
// the from_arp precompute before it, proof size.
// Build: g++ -O2 -std=c++17 prove_shape.cpp -L<repo>/hodor_amd -lhodor_gpu
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "ali_instance.hpp"

using namespace hodor;

static const uint64_t BN256_FR[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const char *PHASES[] = {"Witness polys", "F LDEs", "F oracles", "G poly", "G LDE", "G oracle", "H1 and H2", "FRI", "queries"};
static const size_t G_FACTOR = ali::MAX_CONSTRAINT_POWER;   // constraint domain = 4 x trace domain
static const uint64_t SEED = 0x50524F56;     // tests/prove_shape_ref.py:make_trace

typedef Polynomial<Coefficients> PolyC;
typedef Polynomial<Values> PolyV;

static bool g_fri_batch = true;   // argv[9]: 0 = one proof_from_lde after the other (the A/B of hodor_fri_commit_batch_h)

struct Prep {   // ALIInstance::from_arp's vectors + fixed scalars (prove_shape_ref.make_trace)
    Fr coeff, constant[2], boundary_value, masks[2];
    ali::ALIInstance instance;
};

static void put64(std::vector<uint8_t> &o, uint64_t v) { for (int b = 0; b < 8; b++) o.push_back((uint8_t)(v >> (8 * b))); }
static void put(std::vector<uint8_t> &o, const void *p, size_t n) { o.insert(o.end(), (const uint8_t *)p, (const uint8_t *)p + n); }

// ALI's calculate_g (src/ali/per_register/mod.rs:246-526) on the synthetic constraint system of tests/ali_replay_ref.py
// (calculate_g_for_instance): two challenges per constraint drawn from the transcript as the reference draws them
struct Term { int reg; uint64_t power; int kind; };   // kind: 0 one, 1 minus_one, 2 scale
static PolyC calculate_g(const Field &F, Transcript &transcript, const std::vector<PolyC> &witness, const Prep &P)
{
    const Term c0[] = {{0, 2, 2}, {1, 1, 1}}, c1[] = {{1, 3, 0}, {0, 1, 2}};
    const Term *constraints[2] = {c0, c1};
    const uint64_t degree[2] = {2, 4};
    const ali::ALIInstance &I = P.instance;
    const size_t big = (size_t)I.constraints_domain.size;
    PolyV g = PolyV::new_for_size(F, big), batch = PolyV::new_for_size(F, big);              // :249, :427
    for (int ci = 0; ci < 2; ci++) {
        const uint64_t adjustment = I.max_constraint_power - degree[ci];                   // :431
        const Fr alpha = transcript.get_challenge(), beta = transcript.get_challenge();    // :432-433
        PolyV cv = PolyV::new_for_size(F, big);                                            // :451
        for (int t = 0; t < 2; t++) {
            const Term &term = constraints[ci][t];
            PolyV base = coset_lde(witness[term.reg], G_FACTOR);      // :402-417
            if (term.power != 1) base.pow(term.power);
            if (term.kind == 1) base.negate();
            else if (term.kind == 2) base.scale(P.coeff);
            cv.add_assign(base);                                      // :455-463
        }
        cv.add_constant(P.constant[ci]);                              // :465
        if (adjustment) cv.mul_assign(I.calculate_adjustment_polynomial_in_coset(F, adjustment, alpha, beta));   // :435-449, :467
        else cv.scale(alpha);                                         // :470
        batch.add_assign(cv);                                         // :473
    }
    batch.mul_assign(I.constraint_divisors);                          // :476-478
    g.add_assign(batch);                                              // :480
    for (const ali::BoundaryConstraint &b_c : ali::BOUNDARY) {        // :486-521
        const Fr alpha = transcript.get_challenge(), beta = transcript.get_challenge();
        const uint64_t adjustment = I.max_constraint_power - 1;
        PolyC w = witness[b_c.reg].clone();
        w.sub_assign_at(0, P.boundary_value);                         // witness_poly.as_mut()[0].sub_assign(&value) :511 (one element: on the device)
        PolyV cv = coset_lde(w, G_FACTOR);
        if (adjustment) cv.mul_assign(I.calculate_adjustment_polynomial_in_coset(F, adjustment, alpha, beta));
        else cv.scale(alpha);
        cv.mul_assign(I.boundary_constraint_divisors.at(b_c.at_row));
        g.add_assign(cv);
    }
    return icoset_fft(std::move(g));                                  // :523
}

// ALI's calculate_deep (src/ali/per_register/deep.rs:14-146) with the instance of tests/deep_replay_ref.py
struct Deep { PolyV h1, h2; std::vector<Fr> f_at_z_m; Fr g_at_z; };
static Deep calculate_deep(const Field &F, const std::vector<PolyC> &f_polys, const std::vector<PolyV> &f_ldes, const PolyC &g_poly,
                           const PolyV &g_lde, const Fr &z, const Fr masks[2], const Fr alphas[3])
{
    const int MASKS[3][2] = {{0, 0}, {1, 0}, {0, 1}};   // (register, mask index)
    const size_t f_size = f_ldes[0].size(), g_size = g_lde.size();
    std::map<int, PolyV> divisors;
    Deep d;
    d.h1 = PolyV::new_for_size(F, f_size);
    auto inverse_divisor = [&](const Fr &root, size_t size) {         // :58-72, :126-135
        PolyC q_poly = PolyC::new_for_size(F, 2);
        q_poly.set(1, F.one());                                       // q_poly.as_mut()[1] = F::one()        :61
        q_poly.sub_assign_at(0, root);                                // q_poly.as_mut()[0].sub_assign(&root) :62
        PolyV inv = evaluate_at_domain_for_degree_one(q_poly, size);
        if (!inv.batch_inversion()) throw SynthesisError(HODOR_ERR_INVALID, "divisor vanishes on the domain");
        return inv;
    };
    for (int k = 0; k < 3; k++) {
        const int reg = MASKS[k][0], mi = MASKS[k][1];
        const Fr root = F.mul(masks[mi], z);                          // :34-43
        const Fr value = f_polys[reg].evaluate_at(root);              // :54
        d.f_at_z_m.push_back(value);
        if (!divisors.count(mi)) divisors.emplace(mi, inverse_divisor(root, f_size));
        quotient_term(d.h1, f_ldes[reg], divisors.at(mi), value, &alphas[k], true);   // :74-84 in one pass
    }
    PolyV inv = inverse_divisor(z, g_size);
    d.g_at_z = g_poly.evaluate_at(z);                                 // :137
    d.h2 = PolyV::new_for_size(F, g_size);
    quotient_term(d.h2, g_lde, inv, d.g_at_z, nullptr, false);        // :139-144
    return d;
}

template <class IOP> struct QueryBytes;
template <> struct QueryBytes<TrivialBlake2sIOP> {
    static void put_query(std::vector<uint8_t> &o, const TrivialBlake2sIopQuery &q)
    {
        put(o, q.value_.l, 32); put64(o, q.path_.size());
        for (auto &h : q.path_) put(o, h.data(), 32);
    }
};
template <> struct QueryBytes<Coset2Blake2sIOP> {
    static void put_query(std::vector<uint8_t> &o, const Coset2Blake2sIopQuery &q)
    {
        put(o, q.values_[0].l, 32); put(o, q.values_[1].l, 32); put64(o, q.path_.size());
        for (auto &h : q.path_) put(o, h.data(), 32);
    }
};

struct Clock {
    const Field &F;
    bool sync;
    std::chrono::steady_clock::time_point t0;
    std::map<std::string, double> ms;
    double now_lap()
    {
        if (sync) F.synchronize();
        auto t1 = std::chrono::steady_clock::now();
        double d = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
        return d;
    }
    void start() { if (sync) F.synchronize(); t0 = std::chrono::steady_clock::now(); }
    void lap(const char *name) { ms[name] += now_lap(); }
};

// Prover::prove (src/prover/mod.rs:66-174)
template <class IOP>
static std::vector<uint8_t> prove(const Field &F, const std::vector<PolyV> &trace, const Prep &P, size_t lde_factor, Clock &clk)
{
    const int combiner = std::is_same<IOP, Coset2Blake2sIOP>::value ? HODOR_COMBINER_COSET2 : HODOR_COMBINER_TRIVIAL;
    Transcript T(F);
    clk.start();
    // Witness polys: calculate_witness_polys, one ifft per register (src/arp/per_register/mod.rs:13-68)
    std::vector<PolyC> w_polys;
    for (auto &v : trace) w_polys.push_back(ifft(v.clone()));
    clk.lap("Witness polys");
    // F LDEs (:73-76) and F oracles (:77-85)
    std::vector<PolyV> f_ldes = lde_all(w_polys, lde_factor);
    clk.lap("F LDEs");
    std::vector<IOP> f_oracles = IOP::create_all(F, f_ldes);
    std::vector<Hash32> f_iop_roots = IOP::get_roots(F, f_oracles);
    for (auto &r : f_iop_roots) T.commit_bytes(r);
    clk.lap("F oracles");
    // G poly (:87), G LDE (:89), G oracle (:91-93)
    std::vector<PolyC> two;
    two.push_back(w_polys[0].clone());
    two.push_back(w_polys[1].clone());
    PolyC g_poly = calculate_g(F, T, two, P);
    clk.lap("G poly");
    PolyV g_lde = lde(g_poly, lde_factor);
    clk.lap("G LDE");
    IOP g_oracle = IOP::create(F, g_lde);
    Hash32 g_iop_root = g_oracle.get_root();
    T.commit_bytes(g_iop_root);
    clk.lap("G oracle");
    // DEEP (:97-104)
    const Fr z = T.get_challenge();
    Fr alphas[3];
    for (auto &a : alphas) a = T.get_challenge();
    Deep d = calculate_deep(F, w_polys, f_ldes, g_poly, g_lde, z, P.masks, alphas);
    clk.lap("H1 and H2");
    // FRI (:110-111): the two commits are independent — issued together their latency-bound tails overlap
    FRIProofPrototype p1, p2;
    if (g_fri_batch) {
        auto both = NaiveFriIop::proof_from_lde_all({&d.h1, &d.h2}, lde_factor, 1, combiner);
        p1 = std::move(both[0]);
        p2 = std::move(both[1]);
    } else {
        p1 = NaiveFriIop::proof_from_lde(d.h1, lde_factor, 1, combiner);
        p2 = NaiveFriIop::proof_from_lde(d.h2, lde_factor, 1, combiner);
    }
    clk.lap("FRI");
    // query phase (:113-151)
    for (FRIProofPrototype *p : {&p1, &p2}) {
        T.commit_bytes(p->get_final_root());
        for (auto &c : p->get_final_coefficients()) T.commit_field_element(c);
    }
    const size_t x1 = Transcript::bytes_to_challenge_index(T.get_challenge_bytes(), d.h1.size(), lde_factor);
    const size_t x2 = Transcript::bytes_to_challenge_index(T.get_challenge_bytes(), d.h2.size(), lde_factor);
    std::vector<uint8_t> proof1 = produce_proof_bytes(p1, d.h1, x1), proof2 = produce_proof_bytes(p2, d.h2, x2);
    std::vector<uint8_t> out;
    put64(out, d.f_at_z_m.size());
    for (auto &v : d.f_at_z_m) put(out, v.l, 32);
    put(out, d.g_at_z.l, 32);
    for (auto &r : f_iop_roots) put(out, r.data(), 32);
    put(out, g_iop_root.data(), 32);
    for (size_t r = 0; r < f_oracles.size(); r++) QueryBytes<IOP>::put_query(out, f_oracles[r].query(x1, f_ldes[r]));
    QueryBytes<IOP>::put_query(out, g_oracle.query(x2, g_lde));
    clk.lap("queries");
    for (FRIProofPrototype *p : {&p1, &p2}) {
        auto roots = p->get_roots();
        put64(out, roots.size());
        for (auto &r : roots) put(out, r.data(), 32);
    }
    put64(out, x1); put64(out, proof1.size()); put(out, proof1.data(), proof1.size());
    put64(out, x2); put64(out, proof2.size()); put(out, proof2.data(), proof2.size());
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s log_rows registers lde_factor combiner out.bin [reps] [sync_phases] [ali_mode]\n", argv[0]); return 2; }
    const unsigned log_rows = (unsigned)atoi(argv[1]);
    const size_t registers = (size_t)atoi(argv[2]), lde_factor = (size_t)atoi(argv[3]);
    const int combiner = atoi(argv[4]);
    const int reps = argc > 6 ? atoi(argv[6]) : 1;
    const bool sync_phases = argc > 7 && atoi(argv[7]) != 0;
    const int ali_mode = argc > 8 ? atoi(argv[8]) : 0;   // 0: from_arp as written (as_mut), 1: device-resident precompute
    g_fri_batch = argc > 9 ? atoi(argv[9]) != 0 : true;
    try {
        Field F(BN256_FR, 7, 0);
        const size_t n = (size_t)1 << log_rows;
        std::vector<PolyV> trace;
        for (size_t r = 0; r < registers; r++) trace.push_back(PolyV::generated(F, 0, n, SEED + r));
        Prep P;
        {
            auto sc = PolyV::generated(F, 0, 6, SEED + 100).as_ref().to_vec();   // 8 entries, the last two are padding
            P.coeff = sc[0]; P.constant[0] = sc[1]; P.constant[1] = sc[2]; P.boundary_value = sc[3]; P.masks[0] = sc[4]; P.masks[1] = sc[5];
        }
        // ALIInstance::from_arp (instance setup, before the proof): timed and metered on its own
        Worker worker;
        double from_arp_ms[2] = {0, 0};
        uint64_t from_arp_trips = 0;
        std::pair<uint64_t, uint64_t> from_arp_bytes;
        for (int rep = 0; rep < 2; rep++) {   // run 0 warms the pool, the pinned host images and the tables
            F.synchronize();
            F.reset_host_round_trips();
            auto t0 = std::chrono::steady_clock::now();
            P.instance = ali::ALIInstance::from_arp(F, n, worker, ali_mode != 0);
            F.synchronize();
            from_arp_ms[rep] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            from_arp_trips = F.host_round_trips();
            from_arp_bytes = F.host_traffic();
        }

        std::vector<uint8_t> proof;
        std::vector<std::pair<double, std::map<std::string, double>>> runs;
        uint64_t trips = 0;
        size_t peak_live = 0;
        std::pair<uint64_t, uint64_t> bytes;
        for (int rep = 0; rep < reps + 1; rep++) {   // run 0 warms the twiddle tables, the pool and the FRI slab
            Clock clk{F, sync_phases, {}, {}};
            F.synchronize();
            F.reset_host_round_trips();
            hodor_ctx_pool_peak(F.ctx(), 1);
            auto t0 = std::chrono::steady_clock::now();
            std::vector<uint8_t> got = combiner ? prove<Coset2Blake2sIOP>(F, trace, P, lde_factor, clk)
                                                : prove<TrivialBlake2sIOP>(F, trace, P, lde_factor, clk);
            F.synchronize();
            double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            trips = F.host_round_trips();
            bytes = F.host_traffic();
            peak_live = hodor_ctx_pool_peak(F.ctx(), 0);
            if (rep == 0) proof = got;
            else if (got != proof) { fprintf(stderr, "the run is not deterministic\n"); return 1; }
            if (rep > 0 || reps == 0) runs.emplace_back(total, clk.ms);
        }
        FILE *f = fopen(argv[5], "wb");
        if (!f || fwrite(proof.data(), 1, proof.size(), f) != proof.size()) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
        fclose(f);
        std::sort(runs.begin(), runs.end(), [](auto &a, auto &b) { return a.first < b.first; });
        if (runs.empty()) runs.emplace_back(0.0, std::map<std::string, double>());
        auto &med = runs[runs.size() / 2];
        size_t cached = 0, live = 0;
        hodor_ctx_pool_stats(F.ctx(), &cached, &live);
        printf("{\"log_rows\": %u, \"registers\": %zu, \"lde_factor\": %zu, \"combiner\": %d, \"ali_mode\": \"%s\", \"fri_batch\": %s, \"proof_bytes\": %zu, \"reps\": %d, "
               "\"sync_phases\": %s, \"total_ms\": %.3f, \"best_ms\": %.3f, \"host_round_trips\": %llu, \"h2d_bytes\": %llu, \"d2h_bytes\": %llu, "
               "\"from_arp\": {\"ms_cold\": %.3f, \"ms\": %.3f, \"host_threads\": %zu, \"host_round_trips\": %llu, \"h2d_bytes\": %llu, \"d2h_bytes\": %llu}, "
               "\"pool_gib\": %.2f, \"pool_peak_live_gib\": %.2f, \"phases_ms\": {",
               log_rows, registers, lde_factor, combiner, ali_mode ? "device-resident" : "as written (as_mut)", g_fri_batch ? "true" : "false", proof.size(), reps,
               sync_phases ? "true" : "false", med.first, runs[0].first, (unsigned long long)trips, (unsigned long long)bytes.first,
               (unsigned long long)bytes.second, from_arp_ms[0], from_arp_ms[1], worker.cpus, (unsigned long long)from_arp_trips,
               (unsigned long long)from_arp_bytes.first, (unsigned long long)from_arp_bytes.second,
               (double)(cached + live) / (1ull << 30), (double)peak_live / (1ull << 30));
        for (size_t i = 0; i < 9; i++) printf("%s\"%s\": %.3f", i ? ", " : "", PHASES[i], med.second[PHASES[i]]);
        printf("}, \"runs_ms\": [");
        for (size_t i = 0; i < runs.size(); i++) printf("%s%.3f", i ? ", " : "", runs[i].first);
        printf("]}\n");
    } catch (const SynthesisError &e) {
        fprintf(stderr, "SynthesisError(%d): %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
