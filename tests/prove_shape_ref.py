"""The phases of the reference's cubic-VDF proof run (/root/reference/src/experiments/cubic_vdf.rs:288-354, the same
order as Prover::prove, src/prover/mod.rs:66-174) on a synthetic trace of the same SHAPE — `registers` columns of
2^log_rows values, LDE factor `lde_factor`, a degree-4 constraint domain — written once against abstract operations so
that the same schedule runs (a) on the CPU oracle (the "CPU port" timed beside the device) and (b) device-resident
through the `_dev` ABI, the transcript driving every challenge.  Test infrastructure (tests/test_gpu_prove_shape.py,
bench/prove_shape.py): SURVEY.md §8(f).1/3/4 in use, not new scope — the AIR/ARP/ALI layers that produce the real
constraint system stay in Rust; their polynomial work is replayed by tests/ali_replay_ref.py (calculate_g) and
tests/deep_replay_ref.py (calculate_deep).

    phase (the reference's own names, cubic_vdf.rs)     what runs
    "Witness polys"     :300-302   calculate_witness_polys: one ifft per register (trace values -> coefficients)
    "F LDEs"            :304-308   lde(lde_factor) of every witness polynomial          (one batched launch sequence)
    "F oracles"         :310-318   Blake2sIopTree::create per LDE, roots into the transcript   (one batched commit)
    "G poly"            :324-326   calculate_g: coset LDEs on the constraint domain, value-form ops, icoset_fft
    "G LDE"             :328-330   lde(lde_factor) of g
    "G oracle"          :332-335   tree over g's LDE, root into the transcript
    "H1 and H2"         :337-346   calculate_deep: evaluations at z, divisor polynomials, batch inversions, quotients
    "FRI"               :350-351   proof_from_lde_by_values of h1 and of h2
    "queries"           prover/mod.rs:124-151   query indices from the transcript, FRI proofs, f / g oracle queries
"""
import hashlib
import time

import numpy as np

import ali_replay_ref as ali
import deep_replay_ref as deep

PHASES = ["Witness polys", "F LDEs", "F oracles", "G poly", "G LDE", "G oracle", "H1 and H2", "FRI", "queries"]
G_FACTOR = ali.MAX_CONSTRAINT_POWER   # constraint domain = 4 x trace domain (cubic constraints, padded to a power of two)


def u64(v):
    return int(v).to_bytes(8, "little")


def fr_bytes(mont):
    return int(mont).to_bytes(32, "little")


def make_trace(O, log_rows, registers, seed=0x50524F56):
    """`registers` columns of 2^log_rows trace values from the SplitMix64 stream, fixed scalars, and — since round 6 —
    the vectors ALIInstance::from_arp REALLY precomputes for calculate_g (ali_replay_ref.from_arp: the inverse divisors
    of the dense constraints, the boundary-constraint divisors, the coset table of the adjustment polynomials), where
    rounds 3-5 fed synthetic stand-ins."""
    from oracle.oracle import array_to_ints
    n = 1 << log_rows
    trace = [O.gen_elements(0, n, seed + r) for r in range(registers)]
    sc = array_to_ints(O.gen_elements(0, 6, seed + 100))
    prep = {"coeff": sc[0], "constant": [sc[1], sc[2]], "boundary_value": sc[3], "masks": [sc[4], sc[5]],
            "instance": ali.from_arp(O, n)}
    return trace, prep


def prove(ops, trace, prep, lde_factor, clock=None):
    """Runs the phases; returns (proof bytes, dict phase -> seconds, dict of phase outputs for digests).
    `ops`: OracleProver / DeviceProver below.  `clock`: callable that waits for the device and returns a time."""
    clock = clock or time.perf_counter
    T = ops.transcript()
    times, marks = {}, {}
    t0 = clock()

    def lap(name):
        nonlocal t0
        t1 = clock()
        times[name] = times.get(name, 0.0) + (t1 - t0)
        t0 = t1

    # ---- Witness polys
    w_polys = [ops.ifft(v) for v in trace]
    lap("Witness polys")
    # ---- F LDEs, F oracles
    f_ldes = ops.lde_all(w_polys, lde_factor)
    lap("F LDEs")
    f_trees = ops.commit_all(f_ldes)
    f_roots = [ops.root(t) for t in f_trees]
    for r in f_roots:
        T.commit_bytes(r)
    lap("F oracles")
    # ---- G poly (calculate_g draws its combination challenge from the transcript)
    g_poly = ali.calculate_g_for_instance(ops.ali, w_polys[:2], prep["instance"], prep, lambda: ops.challenge(T))
    lap("G poly")
    g_lde = ops.lde_all([g_poly], lde_factor)[0]
    lap("G LDE")
    g_tree = ops.commit_all([g_lde])[0]
    g_root = ops.root(g_tree)
    T.commit_bytes(g_root)
    lap("G oracle")
    # ---- DEEP
    scalars = {"z": ops.challenge(T), "masks": prep["masks"], "alphas": [ops.challenge(T) for _ in deep.MASKS]}
    h1, h2, f_at_z_m, g_at_z = deep.calculate_deep(ops.deep, w_polys[:2], f_ldes[:2], g_poly, g_lde, scalars)
    lap("H1 and H2")
    # ---- FRI commits
    p1 = ops.fri_commit(h1, lde_factor)
    p2 = ops.fri_commit(h2, lde_factor)
    lap("FRI")
    # ---- query phase (prover/mod.rs:124-151)
    for p in (p1, p2):
        T.commit_bytes(ops.fri_final_root(p))
        for c in ops.fri_final_coeffs(p):
            ops.commit_field_element(T, c)
    x1 = ops.challenge_index(T, ops.size(h1), lde_factor)
    x2 = ops.challenge_index(T, ops.size(h2), lde_factor)
    proof1 = ops.fri_proof(p1, h1, x1, lde_factor)
    proof2 = ops.fri_proof(p2, h2, x2, lde_factor)
    f_queries = [ops.query(t, l, x1) for t, l in zip(f_trees, f_ldes)]
    g_query = ops.query(g_tree, g_lde, x2)
    lap("queries")

    out = [u64(len(f_at_z_m))] + [fr_bytes(v) for v in f_at_z_m] + [fr_bytes(g_at_z)]
    out += f_roots + [g_root]
    for value, path in f_queries + [g_query]:
        vb = b"".join(fr_bytes(v) for v in value) if isinstance(value, tuple) else fr_bytes(value)
        out += [vb, u64(len(path))] + [bytes(x) for x in path]
    out += [u64(len(ops.fri_roots(p1)))] + ops.fri_roots(p1) + [u64(len(ops.fri_roots(p2)))] + ops.fri_roots(p2)
    out += [u64(x1), u64(len(proof1)), proof1, u64(x2), u64(len(proof2)), proof2]
    # what a verifier needs besides the proof bytes: the two FRI proofs and the values the h oracles hold at the queried
    # points (the verifier recomputes them from the f / g queries and the DEEP equations; here they come from the prover)
    prove.last = {"fri": [(proof1, ops.size(h1), x1, ops.value_at(h1, x1)), (proof2, ops.size(h2), x2, ops.value_at(h2, x2))],
                  "queries": [(f_roots[k], x1, f_queries[k]) for k in range(len(f_queries))] + [(g_root, x2, g_query)]}
    marks = {"f_roots": b"".join(f_roots).hex(), "g_root": g_root.hex(),
             "h1_fri": hashlib.blake2s(ops.fri_serialized(p1), digest_size=32).hexdigest(),
             "h2_fri": hashlib.blake2s(ops.fri_serialized(p2), digest_size=32).hexdigest(),
             "x": [x1, x2]}
    ops.release(p1, p2)
    return b"".join(out), times, marks


# ------------------------------------------------------------------------------------------- CPU oracle ("CPU port")
class OracleProver:
    def __init__(self, O, F, combiner=0):
        # combiner: 0 = the reference's tree format, 1 = COSET2 (every oracle — f, g, FRI — commits the coset
        # {i, i + n/2} as one leaf; include/hodor_gpu.h HODOR_COMBINER_COSET2)
        self.O, self.F, self.combiner = O, F, combiner
        self.ali, self.deep = ali.OracleOps(O), deep.OracleOps(O)

    def transcript(self):
        from oracle import pyref as P
        return P.Transcript(self.F)

    def challenge(self, T):
        return self.F.to_mont(T.get_challenge())

    def commit_field_element(self, T, mont):
        T.commit_field_element(self.F.from_mont(mont))

    def challenge_index(self, T, size, factor):
        from oracle import pyref as P
        return P.bytes_to_challenge_index(T.get_challenge_bytes(), size, factor)

    def size(self, a):
        return a.shape[0]

    def ifft(self, values):
        a = values.copy()
        self.O.poly_ifft(a)
        return a

    def lde_all(self, polys, factor):
        return [self.O.poly_lde(p, factor) for p in polys]

    def commit_all(self, ldes):
        if self.combiner:
            return [self.O.iop_create_coset2(l) for l in ldes]
        return [self.O.iop_create(l) for l in ldes]

    def root(self, nodes):
        return bytes(nodes[1])

    def fri_commit(self, lde, factor):
        return self.O.fri_commit(lde, factor, 1, combiner=self.combiner)

    def fri_roots(self, p):
        return list(p["roots"])

    def fri_final_root(self, p):
        return p["final_root"]

    def fri_final_coeffs(self, p):
        from oracle.oracle import array_to_ints
        return array_to_ints(p["final_coeffs"])

    def fri_serialized(self, p):
        return p["serialized"]

    def fri_proof(self, p, lde, index, factor):
        """FRIProofPrototype::produce_proof (src/fri/query_producer.rs:10-53) from the oracle's vectors: the trees
        are rebuilt by the C oracle (the prototype dict keeps values, not nodes); wire format of
        hodor_fri_produce_proof."""
        from oracle.oracle import array_to_ints
        size, idx = len(lde), index
        queries, roots = [], []
        for vec in [lde] + p["inter_values"]:
            pair = (idx + size // 2) % size
            if self.combiner:
                nodes = self.O.iop_create_coset2(vec)
                lo, hi = sorted([idx, pair])
                both = array_to_ints(vec[lo:lo + 1])[0], array_to_ints(vec[hi:hi + 1])[0]
                queries.append((lo, both, self.O.iop_path_coset2(nodes, vec, lo)))
            else:
                nodes = self.O.iop_create(vec)
                for i in sorted([idx, pair]):
                    queries.append((i, array_to_ints(vec[i:i + 1])[0], self.O.iop_path(nodes, vec, i)))
            roots.append(bytes(nodes[1]))
            nxt = size // 2
            idx = idx if idx < nxt else idx - nxt
            size = nxt
        out = u64(len(queries))
        for i, value, path in queries:
            vb = b"".join(fr_bytes(v) for v in value) if isinstance(value, tuple) else fr_bytes(value)
            out += u64(i) + vb + u64(len(path)) + b"".join(bytes(x) for x in path)
        out += u64(len(roots)) + b"".join(roots)
        fc = self.fri_final_coeffs(p)
        out += u64(len(fc)) + b"".join(fr_bytes(c) for c in fc)
        return out + u64(len(lde) // factor) + u64(1) + u64(factor)

    def query(self, nodes, lde, index):
        from oracle.oracle import array_to_ints
        if self.combiner:
            half = len(lde) // 2
            k = index % half
            return (array_to_ints(lde[k:k + 1])[0], array_to_ints(lde[k + half:k + half + 1])[0]), \
                self.O.iop_path_coset2(nodes, lde, index)
        return array_to_ints(lde[index:index + 1])[0], self.O.iop_path(nodes, lde, index)

    def value_at(self, a, index):
        from oracle.oracle import array_to_ints
        return array_to_ints(a[index:index + 1])[0]

    def release(self, *protos):
        pass


# ------------------------------------------------------------------------------------------- device-resident
class DeviceProver:
    """Every polynomial, LDE, tree and FRI vector stays in HBM; what crosses to the host are the 32-byte roots,
    the evaluations at z, the FRI prototypes' roots / final coefficients and the query answers."""

    def __init__(self, O, ctx, stream=None, combiner=0):
        self.O, self.ctx, self.stream, self.combiner = O, ctx, stream, combiner
        self.host_round_trips = 0
        prover = self

        class CountingDeep(deep.FusedDeviceOps):     # the two calls of calculate_deep that return to the host
            def evaluate_at(self, coeffs, x):
                prover.host_round_trips += 1
                return super().evaluate_at(coeffs, x)

            def batch_inversion(self, a):
                prover.host_round_trips += 1
                return super().batch_inversion(a)
        self.ali, self.deep = ali.DeviceOps(ctx, stream), CountingDeep(O, ctx, stream)

    def transcript(self):
        from hodor_amd import _lib
        return _lib.Transcript(self.ctx)

    def challenge(self, T):
        return T.get_challenge()

    def commit_field_element(self, T, mont):
        T.commit_field_element(mont)

    def challenge_index(self, T, size, factor):
        return self.ctx.bytes_to_challenge_index(T.get_challenge_bytes(), size, factor)

    def size(self, a):
        return a.shape[0]

    def ifft(self, values):
        import torch
        out = torch.empty_like(values)
        self.ctx.poly_ifft_dev(values, out, values.shape[0].bit_length() - 1, stream=self.stream)
        return out

    def lde_all(self, polys, factor):
        """All registers in one batched call (hodor_poly_lde_batch_dev)."""
        import torch
        n = polys[0].shape[0]
        src = polys[0] if len(polys) == 1 else torch.cat(polys)
        dst = torch.empty((len(polys) * n * factor, 4), dtype=torch.int64, device=src.device)
        self.ctx.poly_lde_batch_dev(src, dst, n.bit_length() - 1, factor, len(polys), stream=self.stream)
        return [dst[i * n * factor:(i + 1) * n * factor] for i in range(len(polys))]

    def commit_all(self, ldes):
        import torch
        n = ldes[0].shape[0]
        same = all(l.data_ptr() == ldes[0].data_ptr() + i * n * 32 for i, l in enumerate(ldes))
        leafs = ldes[0] if len(ldes) == 1 else (torch.cat(ldes) if not same else None)
        per = n // 2 if self.combiner else n          # a COSET2 tree has n/2 entries
        nodes = torch.empty((len(ldes) * per, 32), dtype=torch.uint8, device=ldes[0].device)
        # (when the batch LDE left the columns back to back they are committed where they are)
        self.ctx.iop_create_batch_combined_dev(ldes[0] if leafs is None else leafs, n, len(ldes), self.combiner, nodes,
                                               stream=self.stream)
        return [nodes[i * per:(i + 1) * per] for i in range(len(ldes))]

    def root(self, nodes):
        self.host_round_trips += 1
        return bytes(nodes[1].cpu().numpy())

    def fri_commit(self, lde, factor):
        self.host_round_trips += 1
        return self.ctx.fri_commit_dev(lde, lde.shape[0], factor, 1, stream=self.stream, combiner=self.combiner)

    def fri_roots(self, p):
        return list(p.roots)

    def fri_final_root(self, p):
        return p.final_root

    def fri_final_coeffs(self, p):
        from oracle.oracle import array_to_ints
        return array_to_ints(p.final_coeffs)

    def fri_serialized(self, p):
        return p.serialized

    def fri_proof(self, p, lde, index, factor):
        self.host_round_trips += 1
        return p.produce_proof(lde, index)["raw"]

    def query(self, nodes, lde, index):
        self.host_round_trips += 1
        if self.combiner:
            values, path = self.ctx.iop_query_combined_dev(lde, nodes, lde.shape[0], self.combiner, index, stream=self.stream)
            return tuple(values), path
        return self.ctx.iop_query_dev(lde, nodes, lde.shape[0], index, stream=self.stream)

    def value_at(self, a, index):
        from oracle.oracle import array_to_ints     # (a verification aid after the run: not one of the prover's round trips)
        return array_to_ints(a[index:index + 1].cpu().numpy().view(np.uint64))[0]

    def release(self, *protos):
        for p in protos:
            p.free()


def to_device(trace, prep):
    import torch

    def dev(x):
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x.view(np.int64).copy()).cuda()
        if isinstance(x, dict):
            return {k: dev(v) for k, v in x.items()}
        return x
    return [dev(t) for t in trace], dev(prep)


def copy_prep(prep):
    """a deep copy of make_trace's prep (the provers modify nothing in it, but a test must not have to trust that)"""
    if isinstance(prep, np.ndarray):
        return prep.copy()
    if isinstance(prep, dict):
        return {k: copy_prep(v) for k, v in prep.items()}
    if isinstance(prep, list):
        return [copy_prep(v) for v in prep]
    return prep
