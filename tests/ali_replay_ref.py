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
