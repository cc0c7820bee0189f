// ali_instance.hpp — ALIInstance::from_arp's precompute (/root/reference/src/ali/per_register/mod.rs:36-244) and
// calculate_adjustment_polynomial_in_coset (:291-306) in C++ against hodor_amd/csrc/host/hodor.hpp ONLY, for the
// synthetic ARP instance of tests/ali_replay_ref.py (same constants: MAX_CONSTRAINT_POWER, DENSE, BOUNDARY).
//
// Two forms, selected by `device_resident`:
//   false  AS WRITTEN — inverse_divisor_for_dense_constraint_in_coset line for line: Polynomial::new_for_size,
//          worker.scope + as_mut().chunks_mut() filling x^T - 1 with one `pow` per element, batch_inversion,
//          worker.scope + as_mut().chunks_mut() again for the root factors; boundary divisors by q_poly.as_mut()[..]
//          + coset_evaluate_at_domain_for_degree_one + batch_inversion; the adjustment polynomial from
//          Polynomial::from_values(precomputations.coset.clone()).  This is what "src/ali untouched" means on the
//          handle surface: the vector crosses PCIe where the Rust code looks at the slice (round 6:
//          hodor_poly_as_mut_h, write-back when the MutSlice guard goes out of scope).
//   true   the same vectors without the crossings: hodor::dense_divisor_on_coset (one launch; a one-function change in
//          src/ali) and the coset table kept as a device-resident Polynomial that is cloned instead of re-uploaded.
// Both give bit-identical vectors (tests/host_cpp/test_host.cpp checks it; tests/test_host_cpp.py compares the proofs
// built on either with the CPU oracle's).
#pragma once
#include <map>

#include "../../hodor_amd/csrc/host/hodor.hpp"

namespace ali {
using namespace hodor;
typedef Polynomial<Coefficients> PolyC;
typedef Polynomial<Values> PolyV;

static const uint64_t MAX_CONSTRAINT_POWER = 4;     // tests/ali_replay_ref.py
struct DenseConstraint { size_t start_at, span; };   // src/air/mod.rs:30-33
static const DenseConstraint DENSE = {0, 1};
struct BoundaryConstraint { size_t reg, at_row; };
static const BoundaryConstraint BOUNDARY[] = {{0, 0}};

// such calls most likely will have start at 0 and num_steps = domain_size - 1                        (:59)
inline std::pair<PolyV, size_t> inverse_divisor_for_dense_constraint_in_coset(const Field &F, const Domain &column_domain,
                                                                             const Domain &evaluation_domain,
                                                                             DenseConstraint dense_constraint, uint64_t num_rows,
                                                                             const Worker &worker)
{
    const size_t start_at = dense_constraint.start_at;
    const uint64_t span = dense_constraint.span;
    size_t divisor_degree = (size_t)column_domain.size;
    const uint64_t divisor_domain_size = column_domain.size;
    divisor_degree -= start_at;
    divisor_degree -= (size_t)(divisor_domain_size - num_rows);
    divisor_degree -= (size_t)span;

    std::vector<Fr> roots;                                                                  // :73-93
    {
        const Fr roots_generator = column_domain.generator;
        Fr root = F.one();
        for (size_t k = 0; k < start_at; k++) {
            roots.push_back(root);
            root = F.mul(root, roots_generator);
        }
        const uint64_t last_step = num_rows - span;
        root = F.pow(roots_generator, last_step);
        for (uint64_t k = last_step; k < divisor_domain_size; k++) {
            roots.push_back(root);
            root = F.mul(root, roots_generator);
        }
    }
    const Fr evaluation_domain_generator = evaluation_domain.generator;
    const Fr multiplicative_generator = F.multiplicative_generator();
    const size_t evaluation_domain_size = (size_t)evaluation_domain.size;

    // these are values at the coset
    PolyV inverse_divisors = PolyV::new_for_size(F, evaluation_domain_size);                // :112

    // prepare for batch inversion
    {
        MutSlice slice = inverse_divisors.as_mut();                                         // :118
        worker.scope(inverse_divisors.size(), [&](Worker::Scope &scope, size_t chunk) {
            size_t i = 0;
            for (auto inv_divis : slice.chunks_mut(chunk)) {
                scope.spawn([&F, &evaluation_domain_generator, &multiplicative_generator, divisor_domain_size, inv_divis, i, chunk] {
                    Fr x = F.pow(evaluation_domain_generator, (uint64_t)(i * chunk));
                    x = F.mul(x, multiplicative_generator);
                    for (Fr &v : inv_divis) {
                        v = F.pow(x, divisor_domain_size);
                        v = F.sub(v, F.one());
                        x = F.mul(x, evaluation_domain_generator);
                    }
                });
                i++;
            }
        });
    }   // the borrow ends: one upload

    // now polynomial is filled with X^T - 1, and need to be inversed
    if (!inverse_divisors.batch_inversion()) throw SynthesisError(HODOR_ERR_INVALID, "batch_inversion: X^T - 1 vanishes on the coset");   // :136

    // now do the evaluation
    {
        MutSlice slice = inverse_divisors.as_mut();                                         // :139 — one download
        worker.scope(inverse_divisors.size(), [&](Worker::Scope &scope, size_t chunk) {
            size_t i = 0;
            for (auto inv_divis : slice.chunks_mut(chunk)) {
                scope.spawn([&F, &roots, &evaluation_domain_generator, &multiplicative_generator, inv_divis, i, chunk] {
                    Fr x = F.pow(evaluation_domain_generator, (uint64_t)(i * chunk));
                    x = F.mul(x, multiplicative_generator);
                    for (Fr &v : inv_divis) {
                        Fr d = v;
                        for (const Fr &root : roots) {
                            Fr tmp = F.sub(x, root);                                        // (X - root)
                            d = F.mul(d, tmp);
                        }
                        // 1 / ( (X^T-1) / (X - 1)(X - omega)(...) ) =  (X - 1)(X - omega)(...) / (X^T-1)
                        v = d;
                        x = F.mul(x, evaluation_domain_generator);
                    }
                });
                i++;
            }
        });
    }   // one upload
    return {std::move(inverse_divisors), divisor_degree};
}

// the roots alone (:73-93), for the device-resident form
inline std::vector<Fr> dense_constraint_roots(const Field &F, const Domain &column_domain, DenseConstraint dc, uint64_t num_rows)
{
    std::vector<Fr> roots;
    Fr root = F.one();
    for (size_t k = 0; k < dc.start_at; k++) { roots.push_back(root); root = F.mul(root, column_domain.generator); }
    const uint64_t last_step = num_rows - dc.span;
    root = F.pow(column_domain.generator, last_step);
    for (uint64_t k = last_step; k < column_domain.size; k++) { roots.push_back(root); root = F.mul(root, column_domain.generator); }
    return roots;
}

struct ALIInstance {                                                                        // :21-34
    uint64_t num_rows = 0, max_constraint_power = 0;
    Domain column_domain, constraints_domain;
    PolyV constraint_divisors;                                  // constraint_divisors[Dense(DENSE)] — the one density of the instance
    std::map<uint64_t, PolyV> boundary_constraint_divisors;
    PrecomputedOmegas precomputations;                          // host tables, as the reference holds them (AS WRITTEN form)
    PolyV coset;                                                // device-resident form: precomputations.coset as a polynomial
    bool device_resident = false;

    static ALIInstance from_arp(const Field &F, uint64_t num_rows, const Worker &worker, bool device_resident)
    {
        ALIInstance r;
        r.device_resident = device_resident;
        r.num_rows = num_rows;
        r.max_constraint_power = MAX_CONSTRAINT_POWER;                                                       // :40-45
        r.column_domain = Domain::new_for_size(F, num_rows);                                                  // :47
        r.constraints_domain = Domain::new_for_size(F, r.column_domain.size * r.max_constraint_power);       // :48
        if (device_resident) {
            PolyC x = PolyC::new_for_size(F, 2);
            x.as_mut()[1] = F.one();
            r.coset = coset_evaluate_at_domain_for_degree_one(x, (size_t)r.constraints_domain.size);          // g w^i, generated in HBM
            r.constraint_divisors = dense_divisor_on_coset(F, (size_t)r.constraints_domain.size, (size_t)r.column_domain.size,
                                                           dense_constraint_roots(F, r.column_domain, DENSE, num_rows));
        } else {
            r.precomputations = PrecomputedOmegas::new_for_domain(F, r.constraints_domain);                   // :49
            r.constraint_divisors = inverse_divisor_for_dense_constraint_in_coset(F, r.column_domain, r.constraints_domain, DENSE,
                                                                                  num_rows, worker).first;   // :173-181
        }
        for (const BoundaryConstraint &b_c : BOUNDARY) {                                                      // :196-210
            if (r.boundary_constraint_divisors.count(b_c.at_row)) continue;
            // precompute divisors
            PolyC q_poly = PolyC::new_for_size(F, 2);
            q_poly.as_mut()[1] = F.one();
            const Fr root = F.pow(r.column_domain.generator, (uint64_t)b_c.at_row);
            { MutSlice s = q_poly.as_mut(); s[0] = F.sub(s[0], root); }                                       // q_poly.as_mut()[0].sub_assign(&root)
            PolyV inverse_q_poly_coset_values = coset_evaluate_at_domain_for_degree_one(q_poly, (size_t)r.constraints_domain.size);
            if (!inverse_q_poly_coset_values.batch_inversion()) throw SynthesisError(HODOR_ERR_INVALID, "boundary divisor vanishes on the coset");
            r.boundary_constraint_divisors.emplace((uint64_t)b_c.at_row, std::move(inverse_q_poly_coset_values));
        }
        return r;
    }

    // calculate_adjustment_polynomial_in_coset (:291-306)
    PolyV calculate_adjustment_polynomial_in_coset(const Field &F, uint64_t adjustment, const Fr &alpha, const Fr &beta) const
    {
        if (adjustment < 1) throw SynthesisError(HODOR_ERR_INVALID, "assert!(adjustment >= 1)");
        PolyV poly = device_resident ? coset.clone() : from_values(F, precomputations.coset);   // from_values(precomputations.coset.clone())
        poly.pow(adjustment);
        poly.scale(alpha);
        poly.add_constant(beta);
        return poly;
    }
};

}  // namespace ali
