"""Python mirror of the reference's `Polynomial<F, P>` / `TrivialBlake2sIOP` / `FRIProofPrototype` over the library's
HANDLE API (include/hodor_gpu.h, "handle API"): the coefficient / value vector, the tree nodes and the prototype live in
HBM behind an opaque handle; only what the reference's callers read on the host (roots, evaluations, query answers,
proofs, `as_ref()` when asked) crosses PCIe.  Method names and argument meaning follow src/polynomials/mod.rs,
src/iop/mod.rs:79-92 and src/fri/mod.rs:43-54; a reference `Err(SynthesisError::Error)` / `assert!` is a HodorError here.

Elements are Montgomery-form integers (what `Fr`'s limbs hold) on the Python side, (n, 4) uint64 arrays in bulk.
There is no CPU path: every call goes to libhodor_gpu.so, which refuses to compute without a device."""
import ctypes as C

import numpy as np

from ._lib import (COSET2, ERR_INVALID, OK, TRIVIAL, HodorError, _Fr, _fr, _hptr, _to_int)  # noqa: F401

COEFFICIENTS, VALUES = 0, 1          # HODOR_FORM_*: the Rust type parameter P
OP = {"add": 0, "sub": 1, "mul": 2}
UN = {"negate": 0, "square": 1, "pow": 2, "scale": 3, "add_constant": 4, "sub_constant": 5}


class _PolyInfo(C.Structure):
    _fields_ = [("exp", C.c_uint32), ("omega", _Fr), ("omegainv", _Fr), ("geninv", _Fr), ("minv", _Fr)]


def _declare(L):
    if getattr(L, "_hodor_handles_declared", False):
        return
    L.hodor_poly_size_h.restype = C.c_size_t
    L.hodor_iop_size_h.restype = C.c_size_t
    L.hodor_fri_produce_proof_h.restype = C.c_size_t
    L.hodor_poly_dev_ptr_h.restype = C.c_void_p
    L.hodor_ctx_stream.restype = C.c_void_p
    L.hodor_ctx_host_round_trips.restype = C.c_uint64
    L.hodor_poly_free_h.restype = None
    L.hodor_iop_free_h.restype = None
    L.hodor_ctx_reset_host_round_trips.restype = None
    L.hodor_ctx_host_traffic.restype = None
    L._hodor_handles_declared = True


class Polynomial:
    """`Polynomial<F, Coefficients>` / `Polynomial<F, Values>` (src/polynomials/mod.rs:24-34) with `coeffs` in HBM."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        _declare(ctx.L)

    # ---- constructors (:140-166, :716-742)
    @classmethod
    def _from_host(cls, ctx, form, arr):
        _declare(ctx.L)
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        ctx._chk(ctx.L.hodor_poly_from_host_h(ctx.h, C.c_int(form), _hptr(arr), C.c_size_t(len(arr)), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_coeffs(cls, ctx, coeffs):
        return cls._from_host(ctx, COEFFICIENTS, coeffs)

    @classmethod
    def from_values(cls, ctx, values):
        return cls._from_host(ctx, VALUES, values)

    @classmethod
    def new_for_size(cls, ctx, form, size):
        _declare(ctx.L)
        h = C.c_void_p()
        ctx._chk(ctx.L.hodor_poly_new_for_size_h(ctx.h, C.c_int(form), C.c_size_t(size), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def generated(cls, ctx, form, first_index, count, seed):
        """elements [first_index, first_index + count) of the synthetic stream `seed` (hodor_gen_elements_dev)"""
        _declare(ctx.L)
        h = C.c_void_p()
        ctx._chk(ctx.L.hodor_poly_gen_h(ctx.h, C.c_int(form), C.c_uint64(first_index), C.c_size_t(count),
                                        C.c_uint64(seed), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx, form, tensor_or_ptr, length, producer_stream=0):
        _declare(ctx.L)
        ptr = tensor_or_ptr if isinstance(tensor_or_ptr, int) else tensor_or_ptr.data_ptr()
        h = C.c_void_p()
        ctx._chk(ctx.L.hodor_poly_from_dev_h(ctx.h, C.c_int(form), C.c_void_p(ptr), C.c_size_t(length),
                                             C.c_void_p(producer_stream), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def degree_one_on_domain(cls, ctx, n, alpha, c, coset=False):
        """(coset_)evaluate_at_domain_for_degree_one (:229-290) of q(x) = c + alpha x"""
        _declare(ctx.L)
        h = C.c_void_p()
        ctx._chk(ctx.L.hodor_poly_degree_one_on_domain_h(ctx.h, C.c_size_t(n), C.byref(_fr(alpha)), C.byref(_fr(c)),
                                                         C.c_int(1 if coset else 0), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_roots(cls, ctx, roots, chunks=4):
        """Polynomial::from_roots (:168-227): prod (x - r_i).  Per-chunk products in coefficient form on the host (the
        library's scalar helpers), then from_coeffs / lde / mul_assign / ifft exactly as the reference composes them."""
        roots = [int(r) for r in roots]
        if not roots:
            raise HodorError(ERR_INVALID, 'from_roots: result.expect("is some")')
        size = 1
        while size < len(roots) + 1:
            size <<= 1
        chunk = (len(roots) + chunks - 1) // chunks
        result = None
        for start in range(0, len(roots), chunk):
            s = []
            for r in roots[start:start + chunk]:
                if not s:
                    s = [ctx.sub(0, r), ctx.one]                                   # :187-190
                    continue
                tmp = [0] + s                                                      # x * s
                for i, c in enumerate(s):
                    tmp[i] = ctx.sub(tmp[i], ctx.mul(c, r))                        # - r * s  :195-199
                s = tmp
            arr = np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in s], dtype=np.uint64)
            t = cls.from_coeffs(ctx, arr)
            tv = t.lde(size // t.size())                                           # :214-216
            t.free()
            if result is None:
                result = tv
            else:
                result.mul_assign(tv)                                              # :217-221
                tv.free()
        return result.ifft()                                                       # :225

    def clone(self):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_poly_clone_h(self.h, C.byref(h)))
        return Polynomial(self.ctx, h)

    def free(self):
        if self.h:
            self.ctx.L.hodor_poly_free_h(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- accessors
    def size(self):
        return int(self.ctx.L.hodor_poly_size_h(self.h))

    @property
    def form(self):
        return int(self.ctx.L.hodor_poly_form_h(self.h))

    def info(self):
        """exp, omega, omegainv, geninv, minv (:28-33)"""
        i = _PolyInfo()
        self.ctx._chk(self.ctx.L.hodor_poly_info_h(self.h, C.byref(i)))
        return {"exp": int(i.exp), "omega": _to_int(i.omega.l), "omegainv": _to_int(i.omegainv.l),
                "geninv": _to_int(i.geninv.l), "minv": _to_int(i.minv.l)}

    def dev_ptr(self):
        return int(self.ctx.L.hodor_poly_dev_ptr_h(self.h))

    def as_ref(self):
        """as_ref() :42 — the whole vector on the host (a copy of the library's host mirror)"""
        p = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_poly_as_ref_h(self.h, C.byref(p)))
        n = self.size()
        buf = (C.c_uint64 * (4 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).copy()

    def as_mut(self):
        """as_mut() :46 — the WHOLE vector as a writable (n, 4) uint64 view of the library's host image (no copy).  The
        image is the vector from now on; it is uploaded by commit_mut() or before the handle's next device operation.
        `with p.mutable() as a:` is the Rust borrow: the write-back happens when the block ends."""
        p = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_poly_as_mut_h(self.h, C.byref(p)))
        n = self.size()
        buf = (C.c_uint64 * (4 * n)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)

    def commit_mut(self):
        self.ctx._chk(self.ctx.L.hodor_poly_commit_mut_h(self.h))

    def mutable(self):
        poly = self

        class _Borrow:
            def __enter__(self):
                return poly.as_mut()

            def __exit__(self, *exc):
                poly.commit_mut()
                return False
        return _Borrow()

    def read(self, first, count):
        out = np.zeros((count, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.hodor_poly_read_h(self.h, C.c_size_t(first), C.c_size_t(count), _hptr(out)))
        return out

    def write(self, first, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
        self.ctx._chk(self.ctx.L.hodor_poly_write_h(self.h, C.c_size_t(first), C.c_size_t(len(arr)), _hptr(arr)))

    def elem_op(self, index, op, c=0, e=0):
        """as_mut()[index].op(c) on the device"""
        self.ctx._chk(self.ctx.L.hodor_poly_elem_op_h(self.h, C.c_size_t(index), C.c_int(UN[op]), C.byref(_fr(c)),
                                                      C.c_uint64(e)))

    def __eq__(self, other):
        eq = C.c_int()
        self.ctx._chk(self.ctx.L.hodor_poly_equal_h(self.h, other.h, C.byref(eq)))
        return bool(eq.value)

    __hash__ = None

    # ---- generic methods (:54-137)
    def distribute_powers(self, g):
        self.ctx._chk(self.ctx.L.hodor_poly_distribute_powers_h(self.h, C.byref(_fr(g))))

    def scale(self, g):
        self.ctx._chk(self.ctx.L.hodor_poly_scale_h(self.h, C.byref(_fr(g))))

    def negate(self):
        self.ctx._chk(self.ctx.L.hodor_poly_negate_h(self.h))

    def pad_by_factor(self, factor):
        self.ctx._chk(self.ctx.L.hodor_poly_pad_by_factor_h(self.h, C.c_size_t(factor)))

    def pad_to_size(self, new_size):
        self.ctx._chk(self.ctx.L.hodor_poly_pad_to_size_h(self.h, C.c_size_t(new_size)))

    def trim_to_degree(self, degree):
        self.ctx._chk(self.ctx.L.hodor_poly_trim_to_degree_h(self.h, C.c_size_t(degree)))

    # ---- transforms (:611-638, :773-815): in place, the handle changes its form as the Rust value changes its type
    def fft(self):
        self.ctx._chk(self.ctx.L.hodor_poly_fft_h(self.h))
        return self

    def coset_fft(self):
        self.ctx._chk(self.ctx.L.hodor_poly_coset_fft_h(self.h))
        return self

    def coset_fft_for_generator(self, gen):
        self.ctx._chk(self.ctx.L.hodor_poly_coset_fft_for_generator_h(self.h, C.byref(_fr(gen))))
        return self

    def ifft(self):
        self.ctx._chk(self.ctx.L.hodor_poly_ifft_h(self.h))
        return self

    def icoset_fft(self):
        self.ctx._chk(self.ctx.L.hodor_poly_icoset_fft_h(self.h))
        return self

    def icoset_fft_for_generator(self, geninv):
        self.ctx._chk(self.ctx.L.hodor_poly_icoset_fft_for_generator_h(self.h, C.byref(_fr(geninv))))
        return self

    def lde(self, factor, coset=False):
        """lde / coset_lde (:343-349): a NEW Values polynomial of size * factor"""
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_poly_lde_h(self.h, C.c_size_t(factor), C.c_int(1 if coset else 0), C.byref(h)))
        return Polynomial(self.ctx, h)

    def coset_lde(self, factor):
        return self.lde(factor, coset=True)

    # The reference's other spellings of the same two functions — filtering_lde / coset_filtering_lde (:355-368,
    # :484-499: best_lde on the zero-padded vector), lde_using_multiple_cosets(_naive) and the coset twins (:370-609) —
    # all return the values `lde` / `coset_lde` return (asserted by its own tests, :1030); one device schedule serves them.
    filtering_lde = lde_using_multiple_cosets = lde_using_multiple_cosets_naive = lambda self, factor: self.lde(factor)
    coset_filtering_lde = coset_lde_using_multiple_cosets = coset_lde_using_multiple_cosets_naive = coset_lde

    def into_coeffs(self):
        """into_coeffs(self) :50 — the vector on the host; the handle is consumed"""
        out = self.as_ref()
        self.free()
        return out

    @staticmethod
    def lde_all(polys, factor, coset=False):
        """every register's LDE in one call (src/prover/mod.rs:73-80)"""
        ctx = polys[0].ctx
        ins = (C.c_void_p * len(polys))(*[p.h for p in polys])
        outs = (C.c_void_p * len(polys))()
        ctx._chk(ctx.L.hodor_poly_lde_batch_h(ins, C.c_size_t(len(polys)), C.c_size_t(factor),
                                              C.c_int(1 if coset else 0), outs))
        return [Polynomial(ctx, C.c_void_p(o)) for o in outs]

    # ---- arithmetic (:640-711, :744-771, :817-954)
    def _binary(self, other, op):
        self.ctx._chk(self.ctx.L.hodor_poly_binary_h(self.h, other.h, C.c_int(OP[op])))

    def add_assign(self, other):
        self._binary(other, "add")

    def sub_assign(self, other):
        self._binary(other, "sub")

    def mul_assign(self, other):
        self._binary(other, "mul")

    def add_assign_scaled(self, other, scaling):
        self.ctx._chk(self.ctx.L.hodor_poly_add_assign_scaled_h(self.h, other.h, C.byref(_fr(scaling))))

    def evaluate_at(self, g):
        out = _Fr()
        self.ctx._chk(self.ctx.L.hodor_poly_evaluate_at_h(self.h, C.byref(_fr(g)), C.byref(out)))
        return _to_int(out.l)

    def pow(self, e):
        self.ctx._chk(self.ctx.L.hodor_poly_pow_h(self.h, C.c_uint64(e)))

    def square(self):
        self.ctx._chk(self.ctx.L.hodor_poly_square_h(self.h))

    def add_constant(self, c):
        self.ctx._chk(self.ctx.L.hodor_poly_add_constant_h(self.h, C.byref(_fr(c))))

    def batch_inversion(self):
        self.ctx._chk(self.ctx.L.hodor_poly_batch_inversion_h(self.h))

    def quotient_term(self, f, divisor_inv, value, alpha, accumulate):
        """self = (accumulate ? self : 0) + alpha (f - value) divisor_inv — one DEEP term (src/ali/per_register/deep.rs)"""
        self.ctx._chk(self.ctx.L.hodor_poly_quotient_term_h(self.h, f.h, divisor_inv.h, C.byref(_fr(value)),
                                                            C.byref(_fr(alpha)), C.c_int(1 if accumulate else 0)))


class IopTree:
    """`TrivialBlake2sIOP` (src/iop/blake2s_trivial_iop.rs:282-339) — or the COSET2 format — with `nodes` in HBM."""

    def __init__(self, ctx, handle, combiner):
        self.ctx, self.h, self.combiner = ctx, handle, combiner
        _declare(ctx.L)

    @classmethod
    def create(cls, values, combiner=TRIVIAL):
        """IOP::create(values.as_ref()) :289-300"""
        h = C.c_void_p()
        values.ctx._chk(values.ctx.L.hodor_iop_create_h(values.h, C.c_int(combiner), C.byref(h)))
        return cls(values.ctx, h, combiner)

    @classmethod
    def create_all(cls, polys, combiner=TRIVIAL):
        """all registers' oracles in one launch sequence (src/prover/mod.rs:77-79)"""
        ctx = polys[0].ctx
        ins = (C.c_void_p * len(polys))(*[p.h for p in polys])
        outs = (C.c_void_p * len(polys))()
        ctx._chk(ctx.L.hodor_iop_create_batch_h(ins, C.c_size_t(len(polys)), C.c_int(combiner), outs))
        return [cls(ctx, C.c_void_p(o), combiner) for o in outs]

    def free(self):
        if self.h:
            self.ctx.L.hodor_iop_free_h(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def size(self):
        return int(self.ctx.L.hodor_iop_size_h(self.h))

    def get_root(self):
        r = (C.c_uint8 * 32)()
        self.ctx._chk(self.ctx.L.hodor_iop_root_h(self.h, r))
        return bytes(r)

    @staticmethod
    def get_roots(trees):
        """many roots behind one wait"""
        ctx = trees[0].ctx
        hs = (C.c_void_p * len(trees))(*[t.h for t in trees])
        out = np.zeros((len(trees), 32), dtype=np.uint8)
        ctx._chk(ctx.L.hodor_iop_roots_h(hs, C.c_size_t(len(trees)), out.ctypes.data_as(C.c_void_p)))
        return [bytes(r) for r in out]

    def nodes(self):
        n = self.size() // (2 if self.combiner == COSET2 else 1)
        out = np.zeros((n, 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.hodor_iop_nodes_h(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def query(self, natural_index, values):
        """IOP::query :324-338 -> (values [1 or 2 Montgomery ints], path [32-byte entries])"""
        vals = np.zeros((2, 4), dtype=np.uint64)
        path = np.zeros((64, 32), dtype=np.uint8)
        plen = C.c_size_t()
        self.ctx._chk(self.ctx.L.hodor_iop_query_h(self.h, values.h, C.c_size_t(natural_index), _hptr(vals),
                                                   path.ctypes.data_as(C.c_void_p), C.byref(plen)))
        k = 2 if self.combiner == COSET2 else 1
        return [_to_int(vals[i]) for i in range(k)], [bytes(path[i]) for i in range(plen.value)]


class FriPrototypeHandle:
    """`FRIProofPrototype` (src/fri/mod.rs:106-117) committed from a device-resident polynomial."""

    def __init__(self, lde_values, lde_factor, output_coeffs_at_degree_plus_one, combiner=TRIVIAL,
                 through_coefficients=False):
        """NaiveFriIop::proof_from_lde (src/fri/mod.rs:43-54) / proof_from_lde_through_coefficients (:156-248)"""
        from ._lib import FriPrototype
        self.ctx, self.lde_values = lde_values.ctx, lde_values
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_fri_commit_h(lde_values.h, C.c_size_t(lde_factor),
                                                    C.c_size_t(output_coeffs_at_degree_plus_one), C.c_int(combiner),
                                                    C.c_int(1 if through_coefficients else 0), C.byref(h)))
        self.proto = FriPrototype(self.ctx, h)      # roots, challenges, final coefficients, canonical serialization

    @classmethod
    def commit_all(cls, ldes, lde_factor, output_coeffs_at_degree_plus_one, combiner=TRIVIAL):
        """proof_from_lde of several polynomials at once (h1 and h2, src/prover/mod.rs:112-113): hodor_fri_commit_batch_h —
        the commits overlap on streams of the context, one wait hands all prototypes over; same prototypes as one
        FriPrototypeHandle(...) each."""
        from ._lib import FriPrototype
        ctx = ldes[0].ctx
        ins = (C.c_void_p * len(ldes))(*[l.h for l in ldes])
        outs = (C.c_void_p * len(ldes))()
        ctx._chk(ctx.L.hodor_fri_commit_batch_h(ins, C.c_size_t(len(ldes)), C.c_size_t(lde_factor),
                                                C.c_size_t(output_coeffs_at_degree_plus_one), C.c_int(combiner), outs))
        res = []
        for l, h in zip(ldes, outs):
            obj = cls.__new__(cls)
            obj.ctx, obj.lde_values = ctx, l
            obj.proto = FriPrototype(ctx, C.c_void_p(h))
            res.append(obj)
        return res

    def produce_proof_bytes(self, natural_first_element_index):
        """prototype_into_proof / produce_proof (src/fri/query_producer.rs:10-53), in this build's wire format"""
        L = self.ctx.L
        need = L.hodor_fri_produce_proof_h(self.proto.h, self.lde_values.h, C.c_size_t(natural_first_element_index),
                                           None, C.c_size_t(0))
        if need == 0:      # "0 on error": the size-returning entry point has no code to give; the reason is the context's last error
            raise HodorError(ERR_INVALID, "hodor_fri_produce_proof_h: " + (L.hodor_last_error(self.ctx.h) or b"").decode())
        buf = (C.c_uint8 * need)()
        got = L.hodor_fri_produce_proof_h(self.proto.h, self.lde_values.h, C.c_size_t(natural_first_element_index),
                                          buf, C.c_size_t(need))
        if got != need:
            raise HodorError(ERR_INVALID, "hodor_fri_produce_proof_h: " + (L.hodor_last_error(self.ctx.h) or b"").decode())
        return bytes(buf)

    def verify_prototype(self, natural_element_index):
        """verify_prototype (src/fri/verifier.rs:10-129)"""
        ok = C.c_int()
        self.ctx._chk(self.ctx.L.hodor_fri_verify_prototype_h(self.proto.h, self.lde_values.h,
                                                              C.c_size_t(natural_element_index), C.byref(ok)))
        return bool(ok.value)

    def commitment(self, step):
        """l0_commitment (step = -1) / intermediate_commitments[step] as an IopTree view"""
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_fri_commitment_h(self.proto.h, C.c_int(step), C.byref(h)))
        return IopTree(self.ctx, h, self.proto.combiner)

    def intermediate_values(self, step):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.hodor_fri_intermediate_values_h(self.proto.h, C.c_size_t(step), C.byref(h)))
        return Polynomial(self.ctx, h)

    def free(self):
        self.proto.free()


def host_round_trips(ctx):
    _declare(ctx.L)
    return int(ctx.L.hodor_ctx_host_round_trips(ctx.h))


def reset_host_round_trips(ctx):
    """... and the PCIe byte counters"""
    _declare(ctx.L)
    ctx.L.hodor_ctx_reset_host_round_trips(ctx.h)


def host_traffic(ctx):
    """(host -> device bytes, device -> host bytes) the library has moved for this context since creation / reset"""
    _declare(ctx.L)
    up, down = C.c_uint64(), C.c_uint64()
    ctx.L.hodor_ctx_host_traffic(ctx.h, C.byref(up), C.byref(down))
    return int(up.value), int(down.value)
