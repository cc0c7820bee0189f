"""pyref.py — Python big-int restatement of the hodor hot path (TEST INFRASTRUCTURE ONLY).

Independent of oracle/hodor_oracle.c: plain Python integers + hashlib.blake2s.  Used to
(a) pin the C oracle and (b) generate the small golden vectors under tests/golden/
(tests/golden/gen_golden.py).  PARITY STATUS: parity unpinned against the Rust binary — the
reference cannot be built in this image and holds no known-answer vectors.

Reference citations are relative to /root/reference.
"""
import hashlib

# Fields defined by the reference
BN256_FR_MODULUS = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # src/bn256.rs:5
BN256_FR_GENERATOR = 7                                                                             # src/bn256.rs:6
EXPERIMENTS_FR_MODULUS = 3618502788666131213697322783095070105623107215331596699973092056135872020481  # src/experiments/mod.rs:19
EXPERIMENTS_FR_GENERATOR = 3

IOP_KEY = b"Squeamish Ossifrage"    # src/iop/blake2s_trivial_iop.rs:12
IOP_PERSONAL = b"Shaftoe"           # src/iop/blake2s_trivial_iop.rs:13


class Field:
    """ff_ce #[derive(PrimeField)] constants for a 4-limb field (R = 2^256)."""

    def __init__(self, p, g):
        self.p, self.g = p, g
        self.R = (1 << 256) % p
        self.Rinv = pow(self.R, -1, p)
        self.num_bits = p.bit_length()
        self.capacity = self.num_bits - 1
        t, s = p - 1, 0
        while t % 2 == 0:
            t //= 2
            s += 1
        self.S, self.t = s, t
        self.root_of_unity = pow(g, t, p)

    # Montgomery <-> canonical
    def to_mont(self, x):
        return (x * self.R) % self.p

    def from_mont(self, m):
        return (m * self.Rinv) % self.p

    def domain_generator(self, size):
        """Domain::new_for_size, src/domains/mod.rs:21-44 (canonical value)."""
        sz = 1
        while sz < size:
            sz <<= 1
        k = sz.bit_length() - 1
        if k > self.S:
            raise ValueError("SynthesisError::Error")
        return pow(self.root_of_unity, 1 << (self.S - k), self.p), k, sz


BN256 = Field(BN256_FR_MODULUS, BN256_FR_GENERATOR)
EXPERIMENTS = Field(EXPERIMENTS_FR_MODULUS, EXPERIMENTS_FR_GENERATOR)
# not a field of the reference: the BN254 scalar field (2-adicity 28, p mod 2^29 != 1), a third modulus
# for the tests so that "the modulus is a parameter" is exercised beyond the two fields with S >= 32
BN254 = Field(21888242871839275222246405745257275088548364400416034343698204186575808495617, 5)


def mont_to_bytes(m):
    """Fr(FrRepr([u64;4])) memory image == encode_leaf (blake2s_trivial_iop.rs:36-42)."""
    return int(m).to_bytes(32, "little")


def bytes_to_mont(b):
    return int.from_bytes(b, "little")


# ------------------------------------------------------------------ transforms (canonical ints)
def naive_dft(F, a, omega):
    n, p = len(a), F.p
    return [sum(a[i] * pow(omega, (i * k) % n, p) for i in range(n)) % p for k in range(n)]


def ntt(F, a, omega):
    """Radix-2 DIT after bit reversal, src/fft/fft.rs:21-66 (canonical ints, table twiddles)."""
    n, p = len(a), F.p
    log_n = n.bit_length() - 1
    a = list(a)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    while m < n:
        w_m = pow(omega, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % p
                u = a[k + j]
                a[k + j] = (u + t) % p
                a[k + j + m] = (u - t) % p
                w = w * w_m % p
        m *= 2
    return a


def _digit_reverse(k, digits, bits):
    r = 0
    for _ in range(digits):
        r = (r << bits) | (k & ((1 << bits) - 1))
        k >>= bits
    return r


def serial_fft_radix_4(F, a, omega):
    """src/fft/radix4_fft/mod.rs:45-123, loop for loop (canonical ints)"""
    n, p = len(a), F.p
    log_n = n.bit_length() - 1
    assert log_n % 2 == 0
    a = list(a)
    for k in range(n):
        rk = _digit_reverse(k, log_n // 2, 2)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    v = pow(omega, n // 4, p)
    m = 1
    for _ in range(log_n // 2):
        w_m = pow(omega, n // (4 * m), p)
        for k in range(0, n, 4 * m):
            w = 1
            for j in range(m):
                u = w
                x0 = a[k + j]
                x1 = a[k + j + m] * w % p
                u = u * w % p
                x2 = a[k + j + 2 * m] * u % p
                u = u * w % p
                x3 = a[k + j + 3 * m] * u % p
                a[k + j] = (x0 + x2 + x1 + x3) % p
                a[k + j + 2 * m] = (x0 + x2 - x1 - x3) % p
                t = (x1 - x3) * v % p
                a[k + j + m] = (x0 - x2 + t) % p
                a[k + j + 3 * m] = (x0 - x2 - t) % p
                w = w * w_m % p
        m *= 4
    return a


def serial_lde(F, a, omega, lde_factor):
    """src/fft/lde.rs:15-126: the radix-2 transform whose early rounds skip the operands known to be zero (is_non_zero
    :28-31, the four-way match :90-116)"""
    n, p = len(a), F.p
    log_n = n.bit_length() - 1
    a = list(a)
    for k in range(n):
        rk = _digit_reverse(k, log_n, 1)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m, step = 1, 0
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), p)
        dense = (lde_factor >> step) <= 1
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                odd, even = k + j + m, k + j
                if dense:
                    odd_nz = even_nz = True
                else:
                    odd_nz = (odd & (lde_factor - 1)) < (1 << step)
                    even_nz = (even & (lde_factor - 1)) < (1 << step)
                if odd_nz and even_nz:
                    t = a[odd] * w % p
                    a[odd] = (a[even] - t) % p
                    a[even] = (a[even] + t) % p
                elif even_nz:
                    a[odd] = a[even]
                elif odd_nz:
                    t = a[odd] * w % p
                    a[odd] = (-t) % p
                    a[even] = t
                w = w * w_m % p
        step += 1
        m *= 2
    return a


def _parallel_split(F, a, omega, log_cpus, sub, non_trivial_len=None):
    """The Cooley-Tukey split shared by parallel_fft (src/fft/fft.rs:68-124), parallel_fft_radix_4
    (src/fft/radix4_fft/mod.rs:125-184) and parallel_lde (src/fft/lde.rs:128-193): sub-sequence j is the naive length-P DFT
    shuffle of the inputs (running `elt`), transformed by `sub` with omega^P, and the outputs are un-shuffled
    a[idx] = tmp[idx & (P - 1)][idx >> log_cpus]."""
    n, p = len(a), F.p
    log_n = n.bit_length() - 1
    assert log_n >= log_cpus
    num_cpus, log_new_n = 1 << log_cpus, log_n - log_cpus
    new_omega = pow(omega, num_cpus, p)
    tmp = []
    for j in range(num_cpus):
        t = [0] * (1 << log_new_n)
        omega_j, omega_step, elt = pow(omega, j, p), pow(omega, j << log_new_n, p), 1
        for i in range(1 << log_new_n):
            for s_ in range(num_cpus):
                idx = (i + (s_ << log_new_n)) % n
                if non_trivial_len is None or idx < non_trivial_len:
                    t[i] = (t[i] + a[idx] * elt) % p
                elt = elt * omega_step % p
            elt = elt * omega_j % p
        tmp.append(sub(t, new_omega))
    return [tmp[idx & (num_cpus - 1)][idx >> log_cpus] for idx in range(n)]


def parallel_fft_radix_4(F, a, omega, log_cpus):
    """src/fft/radix4_fft/mod.rs:125-184"""
    log_n = len(a).bit_length() - 1
    assert log_n % 2 == 0 and log_cpus % 2 == 0
    return _parallel_split(F, a, omega, log_cpus, lambda t, w: serial_fft_radix_4(F, t, w))


def parallel_lde(F, a, omega, log_cpus, lde_factor):
    """src/fft/lde.rs:128-193"""
    new_factor = lde_factor >> log_cpus
    sub = (lambda t, w: ntt(F, t, w)) if new_factor <= 1 else (lambda t, w: serial_lde(F, t, w, new_factor))
    return _parallel_split(F, a, omega, log_cpus, sub, non_trivial_len=len(a) // lde_factor)


def best_lde(F, a, omega, lde_factor, cpus):
    """src/fft/lde.rs:4-13"""
    log_n, log_cpus = len(a).bit_length() - 1, cpus.bit_length() - 1
    return serial_lde(F, a, omega, lde_factor) if log_n <= log_cpus else parallel_lde(F, a, omega, log_cpus, lde_factor)


def distribute_powers(F, a, g):
    """src/fft/mod.rs:110-123"""
    p, out, u = F.p, [], 1
    for v in a:
        out.append(v * u % p)
        u = u * g % p
    return out


def poly_fft(F, a):
    omega, _, _ = F.domain_generator(len(a))
    return ntt(F, a, omega)


def poly_ifft(F, a):
    """src/polynomials/mod.rs:773-798"""
    omega, _, _ = F.domain_generator(len(a))
    minv = pow(len(a), -1, F.p)
    return [v * minv % F.p for v in ntt(F, a, pow(omega, -1, F.p))]


def poly_coset_fft(F, a):
    return poly_fft(F, distribute_powers(F, a, F.g))


def poly_icoset_fft(F, a):
    return distribute_powers(F, poly_ifft(F, a), pow(F.g, -1, F.p))


def poly_lde(F, coeffs, factor, coset=False):
    """lde_using_multiple_cosets / coset_lde_using_multiple_cosets,
    src/polynomials/mod.rs:418-482, :544-609: out[idx] = res[idx % f][idx // f]."""
    n = len(coeffs)
    if factor == 1:
        return poly_coset_fft(F, coeffs) if coset else poly_fft(F, coeffs)
    Omega, _, _ = F.domain_generator(n * factor)
    omega, _, _ = F.domain_generator(n)
    results = []
    for i in range(factor):
        gen = pow(Omega, i, F.p)
        if coset:
            gen = gen * F.g % F.p
        results.append(ntt(F, distribute_powers(F, coeffs, gen), omega))
    return [results[idx % factor][idx // factor] for idx in range(n * factor)]


# ------------------------------------------------------------------ BLAKE2s IOP
def b2s(data):
    return hashlib.blake2s(data, digest_size=32, key=IOP_KEY, person=IOP_PERSONAL).digest()


def hash_leaf(mont):
    return b2s(mont_to_bytes(mont))


def hash_node(l, r):
    return b2s(l + r)


def iop_create(leafs_mont):
    """Blake2sIopTree::create, blake2s_trivial_iop.rs:131-219. Returns nodes (list of 32-B, heap)."""
    n = len(leafs_mont)
    assert n >= 2 and n & (n - 1) == 0
    lh = [hash_leaf(v) for v in leafs_mont]
    nodes = [b"\x00" * 32] * n
    for i in range(n // 2):
        nodes[n // 2 + i] = hash_node(lh[2 * i], lh[2 * i + 1])
    w = n // 4
    while w >= 1:
        for i in range(w):
            nodes[w + i] = hash_node(nodes[2 * (w + i)], nodes[2 * (w + i) + 1])
        w //= 2
    return nodes


def interpret_hash(F, h):
    """blake2s_trivial_iop.rs:48-60 -> canonical int."""
    v = int.from_bytes(h, "big")
    shave = 256 - F.capacity
    top_mask = (0xFFFFFFFFFFFFFFFF >> (shave % 64))
    v &= (top_mask << 192) | ((1 << 192) - 1)
    assert v < F.p
    return v


def iop_path(nodes, leafs_mont, tree_index):
    """get_path, blake2s_trivial_iop.rs:251-279"""
    n = len(nodes)
    path = [hash_leaf(leafs_mont[tree_index ^ 1])]
    idx = tree_index >> 1
    w = n // 2
    while w >= 2:
        path.append(nodes[w + (idx ^ 1)])
        idx >>= 1
        w //= 2
    return path


def iop_verify(root, leaf_mont, path, tree_index):
    h = hash_leaf(leaf_mont)
    idx = tree_index
    for el in path:
        h = hash_node(h, el) if idx & 1 == 0 else hash_node(el, h)
        idx >>= 1
    return h == root


# ------------------------------------------------------------------ COSET2 combiner (opt-in tree format of this build)
# README.md:46 "Proof size optimization with coset combining" (unchecked in the reference); the seam is the
# CosetCombiner trait (src/iop/mod.rs:22-34, only instance src/iop/trivial_coset_combiner.rs:17-53).
TRIVIAL, COSET2 = 0, 1


def coset2_natural_to_tree(i, n):
    """natural_index_into_tree_index: the two members of a coset {k, k + n/2} become neighbours 2k, 2k + 1."""
    return 2 * (i % (n // 2)) + i // (n // 2)


def coset2_tree_to_natural(t, n):
    return (t >> 1) + (t & 1) * (n // 2)


COSET2_LEAF_PERSONAL = b"Shaftoe2"     # COSET2 leaves hash under their own personalisation: never equal to a node hash


def hash_leaf_pair(lo_mont, hi_mont):
    return hashlib.blake2s(mont_to_bytes(lo_mont) + mont_to_bytes(hi_mont), digest_size=32, key=IOP_KEY,
                           person=COSET2_LEAF_PERSONAL).digest()


def iop_create_coset2(values_mont):
    """Tree over the n/2 combined leaves (value[k] || value[k + n/2]); heap array of n/2 digests."""
    n = len(values_mont)
    assert n >= 4 and n & (n - 1) == 0
    L = n // 2
    lh = [hash_leaf_pair(values_mont[k], values_mont[k + L]) for k in range(L)]
    nodes = [b"\x00" * 32] * L
    for i in range(L // 2):
        nodes[L // 2 + i] = hash_node(lh[2 * i], lh[2 * i + 1])
    w = L // 4
    while w >= 1:
        for i in range(w):
            nodes[w + i] = hash_node(nodes[2 * (w + i)], nodes[2 * (w + i) + 1])
        w //= 2
    return nodes


def iop_path_coset2(nodes, values_mont, natural_index):
    n = len(values_mont)
    L = n // 2
    k = natural_index % L
    path = [hash_leaf_pair(values_mont[k ^ 1], values_mont[(k ^ 1) + L])]
    idx, w = k >> 1, L // 2
    while w >= 2:
        path.append(nodes[w + (idx ^ 1)])
        idx >>= 1
        w //= 2
    return path


def iop_verify_coset2(root, lo_mont, hi_mont, path, leaf_index):
    h = hash_leaf_pair(lo_mont, hi_mont)
    idx = leaf_index
    for el in path:
        h = hash_node(h, el) if idx & 1 == 0 else hash_node(el, h)
        idx >>= 1
    return h == root


def _tree(combiner, mont):
    return iop_create_coset2(mont) if combiner == COSET2 else iop_create(mont)


# ------------------------------------------------------------------ FRI commit (by values)
def fri_commit(F, lde_values, lde_factor, out_deg_plus_one, combiner=TRIVIAL):
    """src/fri/fri_on_values.rs:11-159. lde_values canonical ints.
    Returns dict(roots=[l0 + intermediates], challenges, final_root, final_coeffs, inter_values)."""
    p = F.p
    n = len(lde_values)
    omega, _, _ = F.domain_generator(n)
    omega_inv = pow(omega, -1, p)
    two_inv = pow(2, -1, p)
    num_steps = ((n // lde_factor) // out_deg_plus_one).bit_length() - 1
    assert num_steps >= 1
    nodes = _tree(combiner, [F.to_mont(v) for v in lde_values])
    roots = [nodes[1]]
    challenge = interpret_hash(F, nodes[1])
    challenges = [challenge]
    values = list(lde_values)
    inter = []
    for i in range(num_steps):
        half = len(values) // 2
        stride = 1 << i
        nxt = []
        for idx in range(half):
            a, b = values[idx], values[idx + half]
            even = (a + b) % p
            odd = (a - b) * pow(omega_inv, idx * stride, p) % p
            nxt.append((odd * challenge + even) * two_inv % p)
        nodes = _tree(combiner, [F.to_mont(v) for v in nxt])
        roots.append(nodes[1])
        challenge = interpret_hash(F, nodes[1])
        challenges.append(challenge)
        inter.append(nxt)
        values = nxt
    challenges.pop()
    final_coeffs = poly_ifft(F, values)[:out_deg_plus_one]
    return dict(roots=roots, challenges=challenges, final_root=roots[-1],
                final_coeffs=final_coeffs, inter_values=inter)


def fri_fold_coeffs(F, coeffs, beta):
    """src/fri/mod.rs:194-203: a_{2i} + beta * a_{2i+1}"""
    return [(coeffs[2 * i] + beta * coeffs[2 * i + 1]) % F.p for i in range(len(coeffs) // 2)]


def fri_commit_through_coefficients(F, lde_values, lde_factor, out_deg_plus_one, combiner=TRIVIAL):
    """src/fri/mod.rs:156-248: l0 commit (:162), ifft + truncate (:171-173), then per round fold the COEFFICIENTS
    (:190-205), from_coeffs + lde (:208-209), commit (:210), challenge (:213).  Same dict as fri_commit; the
    reference's test asserts the two prototypes equal (:338-343)."""
    n = len(lde_values)
    initial_degree_plus_one = n // lde_factor
    num_steps = (initial_degree_plus_one // out_deg_plus_one).bit_length() - 1
    assert num_steps >= 1
    nodes = _tree(combiner, [F.to_mont(v) for v in lde_values])
    roots = [nodes[1]]
    challenge = interpret_hash(F, nodes[1])
    challenges = [challenge]
    coeffs = poly_ifft(F, list(lde_values))[:initial_degree_plus_one]
    inter = []
    for _ in range(num_steps):
        coeffs = fri_fold_coeffs(F, coeffs, challenge)
        values = poly_lde(F, coeffs, lde_factor)
        nodes = _tree(combiner, [F.to_mont(v) for v in values])
        roots.append(nodes[1])
        challenge = interpret_hash(F, nodes[1])
        challenges.append(challenge)
        inter.append(values)
    challenges.pop()
    assert len(coeffs) == out_deg_plus_one
    return dict(roots=roots, challenges=challenges, final_root=roots[-1],
                final_coeffs=coeffs, inter_values=inter)


def fri_serialize(F, proto):
    """Canonical prototype encoding defined by this build (same layout as o_fri_serialize)."""
    out = len(proto["challenges"]).to_bytes(8, "little")
    for r in proto["roots"]:
        out += r
    for c in proto["challenges"]:
        out += mont_to_bytes(F.to_mont(c))
    out += proto["final_root"]
    out += len(proto["final_coeffs"]).to_bytes(8, "little")
    for c in proto["final_coeffs"]:
        out += mont_to_bytes(F.to_mont(c))
    return out


# ------------------------------------------------------------------ transcript (src/transcript/mod.rs:26-80)
class Transcript:
    """Blake2sTranscript restated on hashlib: one running keyed stream; finalize is non-destructive
    (hashlib's .digest() is too) and every challenge digest is re-absorbed."""

    def __init__(self, F):
        self.F = F
        self.state = hashlib.blake2s(digest_size=32, key=IOP_KEY, person=IOP_PERSONAL)

    def commit_bytes(self, b):
        self.state.update(bytes(b))

    def commit_field_element(self, canonical):
        self.state.update(int(canonical).to_bytes(32, "big"))      # into_repr().write_be, :52-57

    def get_challenge_bytes(self):
        v = self.state.digest()
        self.state.update(v)
        return v

    def get_challenge(self):
        return interpret_hash(self.F, self.get_challenge_bytes())


def bytes_to_challenge_index(b, lde_size, lde_factor):
    """Verifier::bytes_to_challenge_index, src/verifier/mod.rs:246-263"""
    idx = int.from_bytes(b[-8:], "big") % lde_size
    if idx % lde_factor == 0:
        idx = (idx + 1) % lde_size
    if idx % 2 == 0:
        idx = (idx + 1) % lde_size
    return idx


# ------------------------------------------------------------------ FRI query phase
def fri_produce_proof(F, proto, lde_values, natural_first_element_index, lde_factor, out_deg_plus_one,
                      combiner=TRIVIAL):
    """FRIProofPrototype::produce_proof, src/fri/query_producer.rs:10-53: for the l0 oracle and each
    intermediate one, the two queries of the (sorted) coset of domain_idx; the index halves with the
    domain.  `proto` = fri_commit(...) output, `lde_values` canonical ints.  Values are returned in
    Montgomery form (the leaf bytes), like the device path does.
    COSET2: ONE query per round — (coset[0], (value[coset[0]], value[coset[1]]), path of log2(size) - 1 digests)."""
    domain_size, domain_idx = len(lde_values), natural_first_element_index
    queries, roots = [], []
    for r, vec in enumerate([lde_values] + proto["inter_values"]):
        leafs = [F.to_mont(v) for v in vec]
        nodes = _tree(combiner, leafs)
        pair = (domain_idx + domain_size // 2) % domain_size
        if combiner == COSET2:
            lo, hi = sorted([domain_idx, pair])
            queries.append((lo, (leafs[lo], leafs[hi]), iop_path_coset2(nodes, leafs, lo)))
        for idx in sorted([domain_idx, pair]) if combiner == TRIVIAL else []:
            queries.append((idx, leafs[idx], iop_path(nodes, leafs, idx)))
        roots.append(nodes[1])
        nxt = domain_size // 2
        domain_idx = domain_idx if domain_idx < nxt else domain_idx - nxt
        domain_size = nxt
    return dict(queries=queries, roots=roots, final_coeffs=[F.to_mont(c) for c in proto["final_coeffs"]],
                initial_degree_plus_one=len(lde_values) // lde_factor,
                output_coeffs_at_degree_plus_one=out_deg_plus_one, lde_factor=lde_factor)


def fri_proof_to_bytes(proof):
    """The FRIProof wire format this build defines (hodor_amd/csrc/abi_fri.hip, hodor_fri_produce_proof); a COSET2
    query carries both values of its coset (64 bytes) where a TRIVIAL one carries one."""
    u64 = lambda v: int(v).to_bytes(8, "little")
    out = u64(len(proof["queries"]))
    for idx, value, path in proof["queries"]:
        vb = b"".join(mont_to_bytes(v) for v in value) if isinstance(value, tuple) else mont_to_bytes(value)
        out += u64(idx) + vb + u64(len(path)) + b"".join(bytes(x) for x in path)
    out += u64(len(proof["roots"])) + b"".join(bytes(r) for r in proof["roots"])
    out += u64(len(proof["final_coeffs"])) + b"".join(mont_to_bytes(c) for c in proof["final_coeffs"])
    out += u64(proof["initial_degree_plus_one"]) + u64(proof["output_coeffs_at_degree_plus_one"])
    out += u64(proof["lde_factor"])
    return out


# ------------------------------------------------------------------ FRI verifier (acceptance oracle)
def fri_verify_proof_queries_coset2(F, proof, natural_element_index, expected_value_from_oracle):
    """verify_proof_queries (src/fri/verifier.rs:131-289) for COSET2 proofs: the same walk, with the two values of a
    round arriving in ONE query that is checked against the root with ONE path."""
    p = F.p
    two_inv = pow(2, -1, p)
    size = proof["initial_degree_plus_one"] * proof["lde_factor"]
    omega, _, size = F.domain_generator(size)
    x = pow(omega, natural_element_index, p)
    if pow(x, size, p) != 1 or pow(x, size // 2, p) == 1:
        raise ValueError("initial challenge value is not in the LDE domain")
    omega_inv = pow(omega, -1, p)
    expected = None
    domain_size, domain_idx = size, natural_element_index
    for rnd, (root, q) in enumerate(zip(proof["roots"], proof["queries"])):
        if domain_size < 4:
            raise ValueError("domain too small for a combined leaf")
        pair = (domain_idx + domain_size // 2) % domain_size
        coset = sorted([domain_idx, pair])
        if q[0] not in coset:
            return False
        if q[0] != coset[0]:
            raise ValueError("invalid tree index")
        lo, hi = q[1]
        supplied = lo if domain_idx == coset[0] else hi
        if rnd == 0 and supplied != expected_value_from_oracle:
            return False
        if not iop_verify_coset2(root, lo, hi, q[2], coset[0]):
            return False
        challenge = interpret_hash(F, root)
        if expected is not None and F.from_mont(supplied) != expected:
            return False
        f_at_omega, f_at_minus_omega = F.from_mont(lo), F.from_mont(hi)
        divisor = pow(omega_inv, coset[0], p)
        even = (f_at_omega + f_at_minus_omega) % p
        odd = (f_at_omega - f_at_minus_omega) * divisor % p
        expected = (odd * challenge + even) * two_inv % p
        nxt = domain_size // 2
        domain_idx = domain_idx if domain_idx < nxt else domain_idx - nxt
        domain_size = nxt
        omega = omega * omega % p
        omega_inv = omega_inv * omega_inv % p
    if expected is None:
        raise ValueError("is some")
    point = pow(omega, domain_idx, p)
    acc, power = 0, 1
    for c in proof["final_coeffs"]:
        acc = (acc + power * F.from_mont(c)) % p
        power = power * point % p
    return acc == expected


def fri_verify_proof_queries(F, proof, natural_element_index, expected_value_from_oracle, degree=2):
    """NaiveFriIop::verify_proof_queries, src/fri/verifier.rs:131-289, restated line for line.
    `proof`: dict(queries=[(natural_index, value_mont, [path digests])], roots=[bytes], final_coeffs=[mont],
    initial_degree_plus_one, lde_factor).  Returns True / False (Err cases raise ValueError).
    NOTE (reference behaviour): after the last oracle the verifier folds once more with the challenge of
    the final root and compares with the final coefficients evaluated in the next domain — consistent
    with the prover only when output_coeffs_at_degree_plus_one == 1."""
    p = F.p
    two_inv = pow(2, -1, p)
    size = proof["initial_degree_plus_one"] * proof["lde_factor"]
    omega, _, size = F.domain_generator(size)
    x = pow(omega, natural_element_index, p)
    if pow(x, size, p) != 1 or pow(x, size // 2, p) == 1:
        raise ValueError("initial challenge value is not in the LDE domain")
    omega_inv = pow(omega, -1, p)
    expected = None
    domain_size, domain_idx = size, natural_element_index
    queries = proof["queries"]
    if len(queries) % degree != 0:
        raise ValueError("invalid number of queries")
    for rnd, root in enumerate(proof["roots"]):
        qs = queries[degree * rnd:degree * (rnd + 1)]
        pair = (domain_idx + domain_size // 2) % domain_size
        coset = sorted([domain_idx, pair])
        if any(q[0] not in coset for q in qs):
            return False
        if rnd == 0:
            for q in qs:
                if q[0] == natural_element_index and q[1] != expected_value_from_oracle:
                    return False
        for c, q in zip(coset, qs):
            if q[0] != c:
                raise ValueError("invalid tree index")
        for q in qs:
            if not iop_verify(root, q[1], q[2], q[0]):
                return False
        challenge = interpret_hash(F, root)
        f_at_omega = F.from_mont(qs[0][1])
        if expected is not None:
            hit = [q for q in qs if q[0] == domain_idx]
            if len(hit) != 1 or F.from_mont(hit[0][1]) != expected:
                return False
        f_at_minus_omega = F.from_mont(qs[1][1])
        divisor = pow(omega_inv, coset[0], p)
        even = (f_at_omega + f_at_minus_omega) % p
        odd = (f_at_omega - f_at_minus_omega) * divisor % p
        expected = (odd * challenge + even) * two_inv % p
        nxt = domain_size // 2
        domain_idx = domain_idx if domain_idx < nxt else domain_idx - nxt
        domain_size = nxt
        omega = omega * omega % p
        omega_inv = omega_inv * omega_inv % p
    point = pow(omega, domain_idx, p)
    acc, power = 0, 1
    for c in proof["final_coeffs"]:
        acc = (acc + power * F.from_mont(c)) % p
        power = power * point % p
    return acc == expected
