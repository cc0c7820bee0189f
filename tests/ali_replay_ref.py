"""The operation sequence of ALI's `calculate_g` (/root/reference/src/ali/per_register/mod.rs:402-526) on a
small synthetic constraint system, written once against an abstract set of polynomial operations so that
the same schedule runs (a) on the CPU oracle, (b) device-resident through the `_dev` ABI (SURVEY.md
§8(f).1: no PCIe between the LDE and the iFFT), (c) "transform-only offload": the slice API for
coset_lde / icoset_fft and the value-form operations on the host, which is what a prover sees that
binds only the §8(a) functions.  Test infrastructure (tests/test_gpu_ali_replay.py, bench/ali_replay.py).

    per constraint:  for each term:  base = coset_lde(witness[reg], f); base.pow(power); base.scale(coeff)
                                     constraint_values.add_assign(base)                        :402-417, :455-463
                     constraint_values.add_constant(c);  .mul_assign(adj) | .scale(alpha)      :465-471
                     batch_values.add_assign(constraint_values)                                :473
    batch_values.mul_assign(divisors);  g_values.add_assign(batch_values)                      :476-480
    boundary:  w = witness[reg]; w[0] -= value;  cv = coset_lde(w, f); cv.scale(alpha);
               cv.mul_assign(boundary_divisors);  g_values.add_assign(cv)                      :486-521
    g_poly = g_values.icoset_fft()                                                             :523
"""

# (register, power, coeff_kind) per term; coeff_kind: "one" | "minus_one" | "scale"
CONSTRAINTS = [
    {"terms": [(0, 2, "scale"), (1, 1, "minus_one")], "adjust": True},
    {"terms": [(1, 3, "one"), (0, 1, "scale")], "adjust": False},
]


def calculate_g(ops, witness, factor, consts):
    """`ops`: the polynomial operations (see OracleOps / DeviceOps / OffloadOps); `witness`: two coefficient
    vectors in the ops' own representation; `consts`: dict of Montgomery scalars and value-form inputs
    (adj, divisors, boundary_divisors in the ops' representation).  Returns g_poly's coefficients."""
    g = ops.zeros_like(consts["divisors"])
    batch = ops.zeros_like(consts["divisors"])
    for ci, c in enumerate(CONSTRAINTS):
        cv = ops.zeros_like(consts["divisors"])
        for reg, power, kind in c["terms"]:
            base = ops.coset_lde(witness[reg], factor)
            if power != 1:
                ops.pow(base, power)
            if kind == "minus_one":
                ops.negate(base)
            elif kind == "scale":
                ops.scale(base, consts["coeff"])
            ops.add_assign(cv, base)
        ops.add_constant(cv, consts["constant"][ci])
        if c["adjust"]:
            ops.mul_assign(cv, consts["adj"])
        else:
            ops.scale(cv, consts["alpha"])
        ops.add_assign(batch, cv)
    ops.mul_assign(batch, consts["divisors"])
    ops.add_assign(g, batch)
    w = ops.sub_from_first(witness[0], consts["boundary_value"])
    cv = ops.coset_lde(w, factor)
    ops.scale(cv, consts["alpha"])
    ops.mul_assign(cv, consts["boundary_divisors"])
    ops.add_assign(g, cv)
    return ops.icoset_fft(g)


class OracleOps:
    """Host arrays (n, 4) uint64, every operation by the CPU oracle."""

    def __init__(self, O):
        self.O = O

    def zeros_like(self, a):
        import numpy as np
        return np.zeros_like(a)

    def coset_lde(self, coeffs, factor):
        return self.O.poly_lde(coeffs, factor, coset=True)

    def icoset_fft(self, a):
        b = a.copy()
        self.O.poly_icoset_fft(b)
        return b

    def pow(self, a, e):
        self.O.poly_unary(a, "pow", e=e)

    def negate(self, a):
        self.O.poly_unary(a, "negate")

    def scale(self, a, s):
        self.O.poly_unary(a, "scale", c=s)

    def add_constant(self, a, c):
        self.O.poly_unary(a, "add_constant", c=c)

    def add_assign(self, a, b):
        self.O.poly_binary(a, b, "add")

    def mul_assign(self, a, b):
        self.O.poly_binary(a, b, "mul")

    def sub_from_first(self, coeffs, value):
        from oracle.oracle import array_to_ints, ints_to_array
        w = coeffs.copy()
        w[0] = ints_to_array([self.O.sub(array_to_ints(w[0:1])[0], value)])[0]
        return w


class OffloadOps(OracleOps):
    """Transform-only offload: coset_lde / icoset_fft through the slice API (host pointers in and out,
    PCIe both ways per call), everything else on the host as in OracleOps."""

    def __init__(self, O, ctx):
        super().__init__(O)
        self.ctx = ctx

    def coset_lde(self, coeffs, factor):
        return self.ctx.poly_lde(coeffs, factor, coset=True)

    def icoset_fft(self, a):
        b = a.copy()
        self.ctx.poly_icoset_fft(b)
        return b


class DeviceOps:
    """Device tensors (n, 4) int64; every operation a `_dev` call on one stream, nothing crosses PCIe."""

    def __init__(self, ctx, stream=None):
        self.ctx, self.stream = ctx, stream

    def zeros_like(self, a):
        import torch
        return torch.zeros_like(a)

    def coset_lde(self, coeffs, factor):
        import torch
        n = coeffs.shape[0]
        out = torch.empty((n * factor, 4), dtype=torch.int64, device=coeffs.device)
        self.ctx.poly_lde_dev(coeffs, out, n.bit_length() - 1, factor, coset=True, stream=self.stream)
        return out

    def icoset_fft(self, a):
        import torch
        out = torch.empty_like(a)
        self.ctx.poly_icoset_fft_dev(a, out, a.shape[0].bit_length() - 1, stream=self.stream)
        return out

    def pow(self, a, e):
        self.ctx.poly_unary_dev(a, a.shape[0], "pow", e=e, stream=self.stream)

    def negate(self, a):
        self.ctx.poly_unary_dev(a, a.shape[0], "negate", stream=self.stream)

    def scale(self, a, s):
        self.ctx.poly_unary_dev(a, a.shape[0], "scale", c=s, stream=self.stream)

    def add_constant(self, a, c):
        self.ctx.poly_unary_dev(a, a.shape[0], "add_constant", c=c, stream=self.stream)

    def add_assign(self, a, b):
        self.ctx.poly_binary_dev(a, b, a.shape[0], "add", stream=self.stream)

    def mul_assign(self, a, b):
        self.ctx.poly_binary_dev(a, b, a.shape[0], "mul", stream=self.stream)

    def sub_from_first(self, coeffs, value):
        w = coeffs.clone()
        self.ctx.poly_unary_dev(w[0:1], 1, "sub_constant", c=value, stream=self.stream)
        return w


def make_inputs(O, log_n, factor, seed=0x414C49):
    """Two witness polynomials (coefficients) + the value-form inputs, all SplitMix64-generated."""
    from oracle.oracle import array_to_ints
    n = 1 << log_n
    big = n * factor
    witness = [O.gen_elements(0, n, seed), O.gen_elements(0, n, seed + 1)]
    sc = array_to_ints(O.gen_elements(0, 5, seed + 2))
    consts = {"coeff": sc[0], "alpha": sc[1], "constant": [sc[2], sc[3]], "boundary_value": sc[4],
              "adj": O.gen_elements(0, big, seed + 3), "divisors": O.gen_elements(0, big, seed + 4),
              "boundary_divisors": O.gen_elements(0, big, seed + 5)}
    return witness, consts


# =====================================================================================================================
# Round 6: ALIInstance::from_arp's precompute (/root/reference/src/ali/per_register/mod.rs:36-244) and calculate_g with the
# inputs it REALLY works on — the inverse divisors of the dense constraints, the boundary-constraint divisors and the
# adjustment polynomials — instead of the SplitMix64 stand-ins above.  Restated against the CPU oracle (numpy arrays of
# Montgomery elements, every operation one of the oracle's C functions); tests/host_cpp/ali_instance.hpp replays the same
# functions AS WRITTEN through hodor.hpp (Polynomial::as_mut() + Worker::scope + batch_inversion) and, beside it, through
# the device fast path (hodor_poly_dense_divisor_on_coset_h) — all three must give the same vectors, and the proofs
# built on them the same bytes (tests/test_gpu_prove_shape.py, tests/test_host_cpp.py).
#
# The synthetic ARP instance (shared with ali_instance.hpp):
#     num_rows = column domain size n, max_constraint_power = 4 (constraints domain 4 n)
#     two constraints of ONE density Dense{start_at: 0, span: 1} (they hold on every row but the last):
#         c0  degree 2 -> adjustment 2 (adjustment polynomial alpha x^2 + beta on the coset)
#         c1  degree 4 -> adjustment 0 (scaled by alpha)
#     one boundary constraint: register 0 at row 0 -> adjustment 3
MAX_CONSTRAINT_POWER = 4
DENSE = {"start_at": 0, "span": 1}
INSTANCE_CONSTRAINTS = [
    {"terms": [(0, 2, "scale"), (1, 1, "minus_one")], "degree": 2},
    {"terms": [(1, 3, "one"), (0, 1, "scale")], "degree": 4},
]
BOUNDARY = [{"register": 0, "at_row": 0}]


def dense_constraint_roots(O, column_size, num_rows, start_at, span):
    """:73-93 — the rows where the constraint does NOT hold: the first start_at, and from num_rows - span to the end of
    the column domain."""
    _, _, gen = O.domain(column_size)
    roots, root = [], O.one()
    for _ in range(start_at):
        roots.append(root)
        root = O.mul(root, gen)
    last_step = num_rows - span
    root = O.pow(gen, last_step)
    for _ in range(last_step, column_size):
        roots.append(root)
        root = O.mul(root, gen)
    return roots


def inverse_divisor_for_dense_constraint_in_coset(O, column_size, evaluation_size, start_at, span, num_rows):
    """:60-160 -> (inverse divisors on the coset of the evaluation domain, divisor degree): X^T - 1 evaluated at
    x_i = g w^i (:116-131), batch inversion (:136), times (x_i - root) for every root (:140-157)."""
    divisor_degree = column_size - start_at - (column_size - num_rows) - span                     # :69-72
    roots = dense_constraint_roots(O, column_size, num_rows, start_at, span)
    x = O.poly_degree_one_on_domain(evaluation_size, O.one(), 0, coset=True)                       # x_i = g w^i
    inv = x.copy()
    O.poly_unary(inv, "pow", e=column_size)                                                        # x^T            :123
    O.poly_unary(inv, "sub_constant", c=O.one())                                                   # - 1            :124
    O.poly_batch_inversion(inv)                                                                    #                :136
    for root in roots:                                                                             # d *= x - root  :146-151
        t = x.copy()
        O.poly_unary(t, "sub_constant", c=root)
        O.poly_binary(inv, t, "mul")
    return inv, divisor_degree


def boundary_constraint_divisor(O, column_size, evaluation_size, row):
    """:196-210 — q(x) = x - w_col^row on the coset of the constraints domain, inverted"""
    _, _, gen = O.domain(column_size)
    root = O.pow(gen, row)
    q = O.poly_degree_one_on_domain(evaluation_size, O.one(), O.sub(0, root), coset=True)
    O.poly_batch_inversion(q)
    return q


def from_arp(O, num_rows):
    """ALIInstance::from_arp (:36-244) for the instance above: the value-form inputs calculate_g multiplies by."""
    size, _, _ = O.domain(num_rows)                                    # column_domain                         :47
    big, _, _ = O.domain(size * MAX_CONSTRAINT_POWER)                  # constraints_domain                    :48
    divisors, degree = inverse_divisor_for_dense_constraint_in_coset(O, size, big, DENSE["start_at"], DENSE["span"], num_rows)
    return {"column_size": size, "constraints_size": big, "divisor_degree": degree,
            "coset": O.poly_degree_one_on_domain(big, O.one(), 0, coset=True),                    # precomputations.coset :49
            "constraint_divisors": divisors,
            "boundary_constraint_divisors": {b["at_row"]: boundary_constraint_divisor(O, size, big, b["at_row"]) for b in BOUNDARY}}


def calculate_adjustment_polynomial_in_coset(ops, coset, adjustment, alpha, beta):
    """:291-306 — from_values(precomputations.coset.clone()), pow(adjustment), scale(alpha), add_constant(beta)"""
    poly = ops.clone(coset)
    ops.pow(poly, adjustment)
    ops.scale(poly, alpha)
    ops.add_constant(poly, beta)
    return poly


def calculate_g_for_instance(ops, witness, inst, consts, draw):
    """calculate_g (:246-526) on the instance above.  `inst`: from_arp's vectors in the ops' representation; `draw()`: the
    transcript's next challenge (two per constraint — alpha, beta — as the reference draws them, :432-433, :483-484);
    `consts`: coeff, constant[2], boundary_value."""
    g = ops.zeros_like(inst["constraint_divisors"])
    batch = ops.clone(g)                                                                           # :427
    for ci, c in enumerate(INSTANCE_CONSTRAINTS):
        adjustment = MAX_CONSTRAINT_POWER - c["degree"]                                            # :431
        alpha, beta = draw(), draw()
        adj = calculate_adjustment_polynomial_in_coset(ops, inst["coset"], adjustment, alpha, beta) if adjustment else None
        cv = ops.clone(g)                                                                          # :451
        for reg, power, kind in c["terms"]:
            base = ops.coset_lde(witness[reg], MAX_CONSTRAINT_POWER)
            if power != 1:
                ops.pow(base, power)
            if kind == "minus_one":
                ops.negate(base)
            elif kind == "scale":
                ops.scale(base, consts["coeff"])
            ops.add_assign(cv, base)
        ops.add_constant(cv, consts["constant"][ci])                                               # :465
        if adj is not None:
            ops.mul_assign(cv, adj)                                                                # :467
        else:
            ops.scale(cv, alpha)                                                                   # :470
        ops.add_assign(batch, cv)                                                                  # :473
    ops.mul_assign(batch, inst["constraint_divisors"])                                             # :476-478
    ops.add_assign(g, batch)                                                                       # :480
    for b in BOUNDARY:                                                                             # :486-521
        alpha, beta = draw(), draw()
        adjustment = MAX_CONSTRAINT_POWER - 1
        adj = calculate_adjustment_polynomial_in_coset(ops, inst["coset"], adjustment, alpha, beta) if adjustment else None
        w = ops.sub_from_first(witness[b["register"]], consts["boundary_value"])                   # :510-511
        cv = ops.coset_lde(w, MAX_CONSTRAINT_POWER)                                                # :512
        if adj is not None:
            ops.mul_assign(cv, adj)
        else:
            ops.scale(cv, alpha)
        ops.mul_assign(cv, inst["boundary_constraint_divisors"][b["at_row"]])                      # :520-521
        ops.add_assign(g, cv)
    return ops.icoset_fft(g)                                                                       # :523


def _oracle_clone(self, a):
    return a.copy()


def _device_clone(self, a):
    return a.clone()


OracleOps.clone = _oracle_clone
DeviceOps.clone = _device_clone
