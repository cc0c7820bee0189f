"""oracle.py — ctypes binding of oracle/libhodor_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays are numpy uint64 of shape (n, 4): the memory image of Rust `&[Fr]` (Montgomery limbs).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhodor_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("hodor_oracle.c", "hodor_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhodor_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OFr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class OField(C.Structure):
    _fields_ = [("p", C.c_uint64 * 4), ("pinv", C.c_uint64), ("r", OFr), ("r2", OFr),
                ("generator", OFr), ("root_of_unity", OFr), ("s", C.c_uint32),
                ("num_bits", C.c_uint32), ("capacity", C.c_uint32)]


class ODomain(C.Structure):
    _fields_ = [("size", C.c_uint64), ("power_of_two", C.c_uint64), ("generator", OFr)]


class OFriProto(C.Structure):
    _fields_ = [("num_steps", C.c_size_t), ("initial_degree_plus_one", C.c_size_t),
                ("output_coeffs_at_degree_plus_one", C.c_size_t), ("lde_factor", C.c_size_t),
                ("l0_nodes", C.POINTER(C.c_uint8)),
                ("inter_nodes", C.POINTER(C.POINTER(C.c_uint8))),
                ("inter_values", C.POINTER(C.POINTER(OFr))),
                ("inter_sizes", C.POINTER(C.c_size_t)),
                ("challenges", C.POINTER(OFr)), ("final_root", C.c_uint8 * 32),
                ("final_coeffs", C.POINTER(OFr))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.o_fri_commit.restype = C.c_int
        _lib.o_fri_serialize.restype = C.c_size_t
        _lib.o_iop_path.restype = C.c_size_t
        _lib.o_num_cpus.restype = C.c_uint32
    return _lib


def _int_to_limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def _limbs_to_int(l):
    return sum(int(l[i]) << (64 * i) for i in range(4))


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def ints_to_array(vals):
    """list of Montgomery ints -> (n,4) uint64"""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = _int_to_limbs(v)
    return out


def array_to_ints(a):
    return [_limbs_to_int(row) for row in a.reshape(-1, 4)]


class Oracle:
    """One field instance of the C oracle."""

    def __init__(self, modulus, generator):
        self.L = lib()
        self.f = OField()
        mod = (C.c_uint64 * 4)(*_int_to_limbs(modulus))
        if self.L.ofield_init(C.byref(self.f), mod, C.c_uint64(generator)) != 0:
            raise ValueError("unsupported modulus")
        self.modulus = modulus
        self.cpus = int(self.L.o_num_cpus())

    # ---- constants
    def one(self):
        return _limbs_to_int(self.f.r.l)

    def const(self, name):
        return _limbs_to_int(getattr(self.f, name).l)

    def fr(self, mont_int):
        return OFr((C.c_uint64 * 4)(*_int_to_limbs(mont_int)))

    def from_canonical(self, x):
        out = OFr()
        c = (C.c_uint64 * 4)(*_int_to_limbs(x))
        assert self.L.ofr_from_repr(C.byref(self.f), C.byref(out), c) == 0
        return _limbs_to_int(out.l)

    def to_canonical(self, m):
        c = (C.c_uint64 * 4)()
        a = self.fr(m)
        self.L.ofr_into_repr(C.byref(self.f), c, C.byref(a))
        return _limbs_to_int(c)

    def mul(self, a, b):
        x, y = self.fr(a), self.fr(b)
        self.L.ofr_mul(C.byref(self.f), C.byref(x), C.byref(y))
        return _limbs_to_int(x.l)

    def add(self, a, b):
        x, y = self.fr(a), self.fr(b)
        self.L.ofr_add(C.byref(self.f), C.byref(x), C.byref(y))
        return _limbs_to_int(x.l)

    def sub(self, a, b):
        x, y = self.fr(a), self.fr(b)
        self.L.ofr_sub(C.byref(self.f), C.byref(x), C.byref(y))
        return _limbs_to_int(x.l)

    def pow(self, a, e):
        x, out = self.fr(a), OFr()
        self.L.ofr_pow(C.byref(self.f), C.byref(out), C.byref(x), C.c_uint64(e))
        return _limbs_to_int(out.l)

    def inverse(self, a):
        x, out = self.fr(a), OFr()
        assert self.L.ofr_inverse(C.byref(self.f), C.byref(out), C.byref(x)) == 0
        return _limbs_to_int(out.l)

    def domain(self, size):
        d = ODomain()
        if self.L.odomain_new_for_size(C.byref(self.f), C.c_uint64(size), C.byref(d)) != 0:
            raise ValueError("SynthesisError::Error")
        return int(d.size), int(d.power_of_two), _limbs_to_int(d.generator.l)

    # ---- random Montgomery elements (uniform canonical residues), numpy only
    def random_elements(self, n, seed):
        rng = np.random.default_rng(seed)
        out = np.zeros((n, 4), dtype=np.uint64)
        top_bits = self.modulus.bit_length() - 192
        p_limbs = np.array(_int_to_limbs(self.modulus), dtype=np.uint64)
        need = np.arange(n)
        while len(need):
            cand = rng.integers(0, 1 << 64, size=(len(need), 4), dtype=np.uint64)
            cand[:, 3] &= np.uint64((1 << top_bits) - 1)
            # lexicographic compare < p from the top limb down
            lt = np.zeros(len(need), dtype=bool)
            eq = np.ones(len(need), dtype=bool)
            for i in (3, 2, 1, 0):
                lt |= eq & (cand[:, i] < p_limbs[i])
                eq &= cand[:, i] == p_limbs[i]
            out[need[lt]] = cand[lt]
            need = need[~lt]
        return out   # a uniform residue is also a uniform Montgomery representation

    def gen_elements(self, first_index, count, seed, cpus=None):
        """SplitMix64 index-addressable input; same buffer as Context.gen_elements_dev."""
        out = np.zeros((count, 4), dtype=np.uint64)
        self.L.o_gen_elements(C.byref(self.f), _ptr(out), C.c_uint64(first_index), C.c_size_t(count),
                              C.c_uint64(seed), C.c_uint32(cpus or self.cpus))
        return out

    # ---- transforms (in place on (n,4) uint64 arrays)
    def serial_fft(self, a, omega, log_n):
        w = self.fr(omega)
        self.L.o_serial_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n))

    def serial_fft_radix_4(self, a, omega, log_n):
        w = self.fr(omega)
        self.L.o_serial_fft_radix_4(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n))

    def parallel_fft(self, a, omega, log_n, log_cpus):
        w = self.fr(omega)
        self.L.o_parallel_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w),
                              C.c_uint32(log_n), C.c_uint32(log_cpus))

    def best_fft(self, a, omega, log_n, cpus=None):
        w = self.fr(omega)
        self.L.o_best_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w),
                          C.c_uint32(log_n), C.c_uint32(cpus or self.cpus))

    def serial_dit_fft(self, a, omega, log_n, non_zero_entries_count):
        w = self.fr(omega)
        self.L.o_serial_dit_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                C.c_size_t(non_zero_entries_count))

    def parallel_dit_fft(self, a, omega, log_n, log_cpus, non_zero_entries_count):
        w = self.fr(omega)
        self.L.o_parallel_dit_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                  C.c_uint32(log_cpus), C.c_size_t(non_zero_entries_count))

    def best_dit_fft(self, a, omega, log_n, non_zero_entries_count, cpus=None):
        w = self.fr(omega)
        self.L.o_best_dit_fft(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                              C.c_uint32(cpus or self.cpus), C.c_size_t(non_zero_entries_count))

    def serial_lde(self, a, omega, log_n, lde_factor):
        w = self.fr(omega)
        self.L.o_serial_lde(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w),
                            C.c_uint32(log_n), C.c_size_t(lde_factor))

    def parallel_fft_radix_4(self, a, omega, log_n, log_cpus):
        w = self.fr(omega)
        if self.L.o_parallel_fft_radix_4(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                         C.c_uint32(log_cpus)) != 0:
            raise ValueError("assert!(log_n >= log_cpus && log_n % 2 == 0 && log_cpus % 2 == 0)")

    def best_fft_radix_4(self, a, omega, log_n, cpus=None):
        w = self.fr(omega)
        if self.L.o_best_fft_radix_4(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                     C.c_uint32(cpus or self.cpus)) != 0:
            raise ValueError("assert!(log_n % 2 == 0)")

    def parallel_lde(self, a, omega, log_n, log_cpus, lde_factor):
        w = self.fr(omega)
        if self.L.o_parallel_lde(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                 C.c_uint32(log_cpus), C.c_size_t(lde_factor)) != 0:
            raise ValueError("assert!(log_n >= log_cpus)")

    def best_lde(self, a, omega, log_n, lde_factor, cpus=None):
        w = self.fr(omega)
        if self.L.o_best_lde(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                             C.c_size_t(lde_factor), C.c_uint32(cpus or self.cpus)) != 0:
            raise ValueError("assert!(log_n >= log_cpus)")

    def naive_dft(self, a, omega):
        out = np.zeros_like(a)
        w = self.fr(omega)
        self.L.o_naive_dft(C.byref(self.f), _ptr(a), _ptr(out), C.c_size_t(len(a)), C.byref(w))
        return out

    def distribute_powers(self, a, g, cpus=None):
        gg = self.fr(g)
        self.L.o_distribute_powers(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(gg),
                                   C.c_uint32(cpus or self.cpus))

    def _poly(self, name, a, cpus):
        rc = getattr(self.L, name)(C.byref(self.f), _ptr(a), C.c_size_t(len(a)),
                                   C.c_uint32(cpus or self.cpus))
        if rc != 0:
            raise ValueError(name + " failed")

    def poly_fft(self, a, cpus=None):
        self._poly("o_poly_fft", a, cpus)

    def poly_coset_fft(self, a, cpus=None):
        self._poly("o_poly_coset_fft", a, cpus)

    def poly_ifft(self, a, cpus=None):
        self._poly("o_poly_ifft", a, cpus)

    def poly_icoset_fft(self, a, cpus=None):
        self._poly("o_poly_icoset_fft", a, cpus)

    def poly_lde(self, coeffs, factor, coset=False, cpus=None):
        out = np.zeros((len(coeffs) * factor, 4), dtype=np.uint64)
        rc = self.L.o_poly_lde(C.byref(self.f), _ptr(coeffs), C.c_size_t(len(coeffs)),
                               C.c_size_t(factor), C.c_int(1 if coset else 0), _ptr(out),
                               C.c_uint32(cpus or self.cpus))
        if rc != 0:
            raise ValueError("o_poly_lde failed")
        return out

    def poly_binary(self, a, b, op):
        self.L.o_poly_binary(C.byref(self.f), _ptr(a), _ptr(b), C.c_size_t(len(a)),
                             C.c_int({"add": 0, "sub": 1, "mul": 2}[op]))

    def poly_add_scaled(self, a, b, scaling):
        sc = self.fr(scaling)
        self.L.o_poly_add_scaled(C.byref(self.f), _ptr(a), _ptr(b), C.c_size_t(len(a)), C.byref(sc))

    def poly_unary(self, a, op, c=0, e=0):
        code = {"negate": 0, "square": 1, "pow": 2, "scale": 3, "add_constant": 4, "sub_constant": 5}[op]
        cc = self.fr(c)
        self.L.o_poly_unary(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.c_int(code), C.byref(cc), C.c_uint64(e))

    def poly_coset_fft_for_generator(self, a, gen, cpus=None):
        g = self.fr(gen)
        if self.L.o_poly_coset_fft_for_generator(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(g),
                                                 C.c_uint32(cpus or self.cpus)) != 0:
            raise ValueError("SynthesisError::Error")

    def poly_icoset_fft_for_generator(self, a, geninv, cpus=None):
        g = self.fr(geninv)
        if self.L.o_poly_icoset_fft_for_generator(C.byref(self.f), _ptr(a), C.c_size_t(len(a)), C.byref(g),
                                                  C.c_uint32(cpus or self.cpus)) != 0:
            raise ValueError("SynthesisError::Error")

    def poly_degree_one_on_domain(self, n, alpha, c, coset=False):
        """values of q(x) = c + alpha x on the size-n domain (or its coset): src/polynomials/mod.rs:229-290"""
        out = np.zeros((n, 4), dtype=np.uint64)
        aa, cc = self.fr(alpha), self.fr(c)
        if self.L.o_poly_degree_one_on_domain(C.byref(self.f), _ptr(out), C.c_size_t(n), C.byref(aa), C.byref(cc),
                                              C.c_int(1 if coset else 0)) != 0:
            raise ValueError("SynthesisError::Error")
        return out

    def poly_batch_inversion(self, a):
        if self.L.o_poly_batch_inversion(C.byref(self.f), _ptr(a), C.c_size_t(len(a))) != 0:
            raise ValueError("SynthesisError::Error")

    def evaluate_at(self, coeffs, g, cpus=1):
        """sum coeffs[i] g^i; cpus > 1 follows the reference's Worker chunking (same value)."""
        gg, out = self.fr(g), OFr()
        if cpus == 1:
            self.L.o_poly_evaluate_at(C.byref(self.f), _ptr(coeffs), C.c_size_t(len(coeffs)),
                                      C.byref(gg), C.byref(out))
        else:
            self.L.o_poly_evaluate_at_mt(C.byref(self.f), _ptr(coeffs), C.c_size_t(len(coeffs)),
                                         C.byref(gg), C.byref(out), C.c_uint32(cpus or self.cpus))
        return _limbs_to_int(out.l)

    # ---- IOP
    def hash_leaf(self, mont):
        out = (C.c_uint8 * 32)()
        x = self.fr(mont)
        self.L.o_hash_leaf(out, C.byref(x))
        return bytes(out)

    def hash_node(self, l, r):
        out = (C.c_uint8 * 32)()
        self.L.o_hash_node(out, l, r)
        return bytes(out)

    def iop_create(self, leafs, cpus=None):
        n = len(leafs)
        nodes = np.zeros((n, 32), dtype=np.uint8)
        rc = self.L.o_iop_create(_ptr(leafs), C.c_size_t(n), nodes.ctypes.data_as(C.c_void_p),
                                 C.c_uint32(cpus or self.cpus))
        if rc != 0:
            raise ValueError("o_iop_create failed")
        return nodes

    def interpret_hash(self, h):
        out = OFr()
        self.L.o_interpret_hash(C.byref(self.f), bytes(h), C.byref(out))
        return _limbs_to_int(out.l)

    def iop_path(self, nodes, leafs, tree_index):
        n = len(leafs)
        path = np.zeros((max(1, n.bit_length() - 1), 32), dtype=np.uint8)
        cnt = self.L.o_iop_path(nodes.ctypes.data_as(C.c_void_p), _ptr(leafs), C.c_size_t(n),
                                C.c_size_t(tree_index), path.ctypes.data_as(C.c_void_p))
        return path[:cnt]

    def iop_verify(self, root, leaf_mont, path, tree_index):
        x = self.fr(leaf_mont)
        path = np.ascontiguousarray(path)
        return bool(self.L.o_iop_verify(bytes(root), C.byref(x), path.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(len(path)), C.c_size_t(tree_index)))

    # ---- COSET2 combiner (opt-in tree format, hodor_oracle.h)
    def hash_leaf_pair(self, lo_mont, hi_mont):
        out = (C.c_uint8 * 32)()
        a, b = self.fr(lo_mont), self.fr(hi_mont)
        self.L.o_hash_leaf_pair(out, C.byref(a), C.byref(b))
        return bytes(out)

    def iop_create_coset2(self, values, cpus=None):
        n = len(values)
        nodes = np.zeros((n // 2, 32), dtype=np.uint8)
        rc = self.L.o_iop_create_coset2(_ptr(values), C.c_size_t(n), nodes.ctypes.data_as(C.c_void_p),
                                        C.c_uint32(cpus or self.cpus))
        if rc != 0:
            raise ValueError("o_iop_create_coset2 failed")
        return nodes

    def iop_path_coset2(self, nodes, values, natural_index):
        n = len(values)
        path = np.zeros((max(1, n.bit_length() - 2), 32), dtype=np.uint8)
        self.L.o_iop_path_coset2.restype = C.c_size_t
        cnt = self.L.o_iop_path_coset2(nodes.ctypes.data_as(C.c_void_p), _ptr(values), C.c_size_t(n),
                                       C.c_size_t(natural_index), path.ctypes.data_as(C.c_void_p))
        return path[:cnt]

    def iop_verify_coset2(self, root, lo_mont, hi_mont, path, leaf_index):
        a, b = self.fr(lo_mont), self.fr(hi_mont)
        path = np.ascontiguousarray(path)
        return bool(self.L.o_iop_verify_coset2(bytes(root), C.byref(a), C.byref(b), path.ctypes.data_as(C.c_void_p),
                                               C.c_size_t(len(path)), C.c_size_t(leaf_index)))

    # ---- FRI
    def fri_commit(self, lde_values, lde_factor, out_deg_plus_one, cpus=None, combiner=0, through_coefficients=False):
        """Returns dict(serialized=bytes, roots, challenges (Montgomery ints), final_root,
        final_coeffs (n,4), inter_values [arrays]).  combiner: 0 TRIVIAL (the reference's), 1 COSET2.
        through_coefficients: proof_from_lde_through_coefficients (src/fri/mod.rs:156-248) instead of
        proof_from_lde_by_values (src/fri/fri_on_values.rs:11-159)."""
        pp = C.POINTER(OFriProto)()
        fn = self.L.o_fri_commit_through_coefficients if through_coefficients else self.L.o_fri_commit_combined
        fn.restype = C.c_int
        rc = fn(C.byref(self.f), _ptr(lde_values), C.c_size_t(len(lde_values)),
                C.c_size_t(lde_factor), C.c_size_t(out_deg_plus_one), C.c_int(combiner),
                C.c_uint32(cpus or self.cpus), C.byref(pp))
        if rc != 0:
            raise ValueError("o_fri_commit failed")
        p = pp.contents
        need = self.L.o_fri_serialize(pp, None, C.c_size_t(0))
        buf = (C.c_uint8 * need)()
        self.L.o_fri_serialize(pp, buf, C.c_size_t(need))
        ns = p.num_steps
        roots = [bytes(p.l0_nodes[32:64])] + [bytes(p.inter_nodes[i][32:64]) for i in range(ns)]
        challenges = [_limbs_to_int(p.challenges[i].l) for i in range(ns)]
        inter = []
        for i in range(ns):
            sz = p.inter_sizes[i]
            arr = np.ctypeslib.as_array(C.cast(p.inter_values[i], C.POINTER(C.c_uint64)), shape=(sz, 4)).copy()
            inter.append(arr)
        nf = p.output_coeffs_at_degree_plus_one
        fc = np.ctypeslib.as_array(C.cast(p.final_coeffs, C.POINTER(C.c_uint64)), shape=(nf, 4)).copy()
        res = dict(serialized=bytes(buf), roots=roots, challenges=challenges,
                   final_root=bytes(p.final_root), final_coeffs=fc, inter_values=inter,
                   num_steps=ns)
        self.L.o_fri_free(pp)
        return res
