"""The operation sequence of ALI's `calculate_deep` (/root/reference/src/ali/per_register/deep.rs:14-146) — the
DEEP quotients h1, h2 of the prover (src/prover/mod.rs:112-113) — on a small synthetic instance, written once
against an abstract set of polynomial operations (the same split as tests/ali_replay_ref.py): (a) the CPU
oracle, (b) device-resident through the `_dev` ABI.  Test infrastructure.

    z = challenge                                                                                   :23
    per mask m (register r):   root = mask * z                                                      :34-43
        f_at_z = f_polys[r].evaluate_at(root)                                                       :54
        divisor[mask] = (x - root) on the LDE domain, evaluate_at_domain_for_degree_one, inverted   :58-72
        t = f_ldes[r].clone(); t.add_constant(-f_at_z); alpha = challenge; t.scale(alpha)           :74-81
        t.mul_assign(divisor); h1.add_assign(t)                                                     :82-84
    inverse (x - z) on g's LDE domain; g_at_z = g_poly.evaluate_at(z)                               :126-137
    h2 = g_lde.clone(); h2.add_constant(-g_at_z); h2.mul_assign(inverse)                            :139-144
"""

# (register, mask index): two registers, two distinct masks, one of them used twice (the divisor cache of :58)
MASKS = [(0, 0), (1, 0), (0, 1)]


def calculate_deep(ops, f_polys, f_ldes, g_poly, g_lde, scalars):
    """`scalars`: dict z, masks [2], alphas [len(MASKS)] (Montgomery integers; the transcript's challenges in
    the order the reference draws them).  Returns (h1_lde, h2_lde, f_at_z_m, g_at_z)."""
    z = scalars["z"]
    f_size, g_size = ops.size(f_ldes[0]), ops.size(g_lde)
    divisors = {}
    h1 = ops.zeros(f_size, f_ldes[0])
    f_at_z_m = []
    for k, (reg, mi) in enumerate(MASKS):
        root = ops.F.mul(scalars["masks"][mi], z)
        value = ops.evaluate_at(f_polys[reg], root)
        f_at_z_m.append(value)
        if mi not in divisors:
            q = ops.degree_one_on_domain(f_size, ops.F.one(), ops.F.neg(root), f_ldes[0])   # q(x) = x - root
            ops.batch_inversion(q)
            divisors[mi] = q
        if hasattr(ops, "quotient_term"):                # the five passes below as one (hodor_poly_quotient_term_dev)
            ops.quotient_term(h1, f_ldes[reg], divisors[mi], value, scalars["alphas"][k], True)
        else:
            t = ops.clone(f_ldes[reg])
            ops.add_constant(t, ops.F.neg(value))
            ops.scale(t, scalars["alphas"][k])
            ops.mul_assign(t, divisors[mi])
            ops.add_assign(h1, t)
    inv = ops.degree_one_on_domain(g_size, ops.F.one(), ops.F.neg(z), g_lde)
    ops.batch_inversion(inv)
    g_at_z = ops.evaluate_at(g_poly, z)
    if hasattr(ops, "quotient_term"):
        h2 = ops.empty(g_size, g_lde)
        ops.quotient_term(h2, g_lde, inv, g_at_z, None, False)
    else:
        h2 = ops.clone(g_lde)
        ops.add_constant(h2, ops.F.neg(g_at_z))
        ops.mul_assign(h2, inv)
    return h1, h2, f_at_z_m, g_at_z


class _Scalars:
    """Montgomery-integer scalar arithmetic through the oracle (the prover does these on the host)."""

    def __init__(self, O):
        self.O = O

    def mul(self, a, b):
        return self.O.mul(a, b)

    def neg(self, a):
        return self.O.sub(0, a)

    def one(self):
        return self.O.one()


class OracleOps:
    def __init__(self, O):
        self.O, self.F = O, _Scalars(O)

    def size(self, a):
        return a.shape[0]

    def zeros(self, n, like):
        import numpy as np
        return np.zeros((n, 4), dtype=np.uint64)

    def clone(self, a):
        return a.copy()

    def evaluate_at(self, coeffs, x):
        return self.O.evaluate_at(coeffs, x)

    def degree_one_on_domain(self, n, alpha, c, like):
        return self.O.poly_degree_one_on_domain(n, alpha, c)

    def batch_inversion(self, a):
        self.O.poly_batch_inversion(a)

    def add_constant(self, a, c):
        self.O.poly_unary(a, "add_constant", c=c)

    def scale(self, a, s):
        self.O.poly_unary(a, "scale", c=s)

    def mul_assign(self, a, b):
        self.O.poly_binary(a, b, "mul")

    def add_assign(self, a, b):
        self.O.poly_binary(a, b, "add")


class DeviceOps:
    """Device tensors (n, 4) int64; every polynomial operation a `_dev` call (the two evaluations return their
    scalar to the host, as the reference's do: they go into the proof)."""

    def __init__(self, O, ctx, stream=None):
        self.ctx, self.stream, self.F = ctx, stream, _Scalars(O)

    def size(self, a):
        return a.shape[0]

    def zeros(self, n, like):
        import torch
        return torch.zeros((n, 4), dtype=torch.int64, device=like.device)

    def clone(self, a):
        return a.clone()

    def evaluate_at(self, coeffs, x):
        return self.ctx.poly_evaluate_at_dev(coeffs, coeffs.shape[0], x, stream=self.stream)

    def degree_one_on_domain(self, n, alpha, c, like):
        import torch
        out = torch.empty((n, 4), dtype=torch.int64, device=like.device)
        self.ctx.poly_degree_one_on_domain_dev(out, n, alpha, c, stream=self.stream)
        return out

    def batch_inversion(self, a):
        self.ctx.poly_batch_inversion_dev(a, a.shape[0], stream=self.stream)

    def add_constant(self, a, c):
        self.ctx.poly_unary_dev(a, a.shape[0], "add_constant", c=c, stream=self.stream)

    def scale(self, a, s):
        self.ctx.poly_unary_dev(a, a.shape[0], "scale", c=s, stream=self.stream)

    def mul_assign(self, a, b):
        self.ctx.poly_binary_dev(a, b, a.shape[0], "mul", stream=self.stream)

    def add_assign(self, a, b):
        self.ctx.poly_binary_dev(a, b, a.shape[0], "add", stream=self.stream)


class FusedDeviceOps(DeviceOps):
    """DeviceOps with every quotient term — clone / add_constant / scale / mul_assign / add_assign — as ONE pass
    (hodor_poly_quotient_term_dev): what a device-resident prover calls; same canonical results."""

    def empty(self, n, like):
        import torch
        return torch.empty((n, 4), dtype=torch.int64, device=like.device)

    def quotient_term(self, acc, f, divisor_inv, value, alpha, accumulate):
        self.ctx.poly_quotient_term_dev(acc, f, divisor_inv, acc.shape[0], value, alpha, accumulate, stream=self.stream)


def make_inputs(O, log_n, factor, g_factor, seed=0x44454550):
    """Two witness polynomials and g (coefficients), their LDEs (plain lde: src/prover/mod.rs:73-80, :91-93), the
    challenges.  All SplitMix64-generated; g has g_factor / factor times the witness's length like ALI's g."""
    from oracle.oracle import array_to_ints
    n = 1 << log_n
    f_polys = [O.gen_elements(0, n, seed), O.gen_elements(0, n, seed + 1)]
    g_poly = O.gen_elements(0, n * (g_factor // factor if g_factor > factor else 1), seed + 2)
    f_ldes = [O.poly_lde(p, factor) for p in f_polys]
    g_lde = O.poly_lde(g_poly, factor)
    sc = array_to_ints(O.gen_elements(0, 3 + len(MASKS), seed + 3))
    scalars = {"z": sc[0], "masks": [sc[1], sc[2]], "alphas": sc[3:]}
    return f_polys, f_ldes, g_poly, g_lde, scalars
