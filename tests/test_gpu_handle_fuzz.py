"""Stateful fuzz of the HANDLE API (-m gpu): seeded random PROGRAMS over a set of live `Polynomial` handles — constructors,
clone / free, every in-place method, the transforms that change a handle's form, LDEs (single and batched), host access
(read / write / elem_op / as_ref / as_mut with and without the explicit write-back), oracles and FRI commits — executed on
the device and, step by step, on a model that holds each polynomial as a host vector driven by the CPU oracle.  Every
program ends (and is interleaved) with whole-vector comparisons.

What this is for: the single-method tests of test_gpu_handles.py pin each method; they do not pin the STATE the methods
share — the pool (blocks handed from one handle to the next behind events, freed while work is in flight, `trim`), pinned
host images recycled by size, the "host image is the vector" state of `as_mut()` meeting every other entry point (clone,
free, binary operand, tree, FRI, batched LDE, equality), the known-zero state of `new_for_size`, size changes in place.
A wrong ordering or a stale copy there shows up as a differing vector a few steps later, which is what is compared.
Nothing here reads /root/reference."""
import os
import random

import numpy as np
import pytest

import hodor_amd
from hodor_amd.handles import COEFFICIENTS, VALUES, FriPrototypeHandle, IopTree, Polynomial

pytestmark = pytest.mark.gpu

SIZE_LOGS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17]
SIZE_WEIGHTS = [2, 2, 3, 3, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 1, 1]     # 2^15 elements = 1 MiB: pinned images, pool classes
MAX_LIVE = 10
MAX_LOG = 18


def _int(row):
    return sum(int(row[i]) << (64 * i) for i in range(4))


class _P:
    """one live polynomial: the handle and what the reference's Vec<F> would hold"""

    def __init__(self, h, m, form):
        self.h, self.m, self.form = h, m, form

    @property
    def n(self):
        return len(self.m)


class _Program:
    def __init__(self, ctx, O, seed):
        self.ctx, self.O, self.rng = ctx, O, random.Random(seed)
        self.live = []
        self.trace = []
        self.counter = seed * 100003

    # ---- helpers
    def rand_elements(self, n):
        self.counter += 1
        return self.O.random_elements(n, self.counter)

    def rand_scalar(self, nonzero=False):
        v = _int(self.rand_elements(1)[0])
        return v if (v or not nonzero) else self.O.one()

    def pick(self, pred=lambda p: True):
        c = [p for p in self.live if pred(p)]
        return self.rng.choice(c) if c else None

    def check(self, p):
        assert p.h.size() == p.n and p.h.form == p.form, self.trace[-12:]
        got = p.h.as_ref()
        if not np.array_equal(got, p.m):
            bad = np.nonzero((got != p.m).any(axis=1))[0]
            raise AssertionError("vector differs at %d of %d positions (first %d); last steps: %s"
                                 % (len(bad), p.n, bad[0], self.trace[-12:]))

    def drop(self, p):
        p.h.free()
        self.live.remove(p)

    def log(self, *what):
        self.trace.append(" ".join(str(w) for w in what))

    # ---- steps (each returns False when it had nothing to act on)
    def s_new(self):
        if len(self.live) >= MAX_LIVE:
            return False
        k = self.rng.choices(SIZE_LOGS, SIZE_WEIGHTS)[0]
        form = self.rng.choice((COEFFICIENTS, VALUES))
        how = self.rng.choice(("host", "host_ragged", "zeros", "generated", "degree_one"))
        self.log("new", how, k, form)
        if how == "host":
            a = self.rand_elements(1 << k)
            h = Polynomial._from_host(self.ctx, form, a)
            m = a.copy()
        elif how == "host_ragged":                                   # zero-padded to the next power of two
            length = self.rng.randint((1 << k) // 2 + 1, 1 << k)
            a = self.rand_elements(length)
            h = Polynomial._from_host(self.ctx, form, a)
            m = np.zeros((1 << k, 4), dtype=np.uint64)
            m[:length] = a
        elif how == "zeros":
            h = Polynomial.new_for_size(self.ctx, form, 1 << k)
            m = np.zeros((1 << k, 4), dtype=np.uint64)
        elif how == "generated":
            first = self.rng.randrange(1 << 20)
            h = Polynomial.generated(self.ctx, form, first, 1 << k, 77)
            m = self.O.gen_elements(first, 1 << k, 77)
        else:
            alpha, c, coset = self.rand_scalar(), self.rand_scalar(), self.rng.random() < 0.5
            h = Polynomial.degree_one_on_domain(self.ctx, 1 << k, alpha, c, coset)
            m = self.O.poly_degree_one_on_domain(1 << k, alpha, c, coset)
            form = VALUES
        self.live.append(_P(h, m, form))
        return True

    def s_clone(self):
        p = self.pick()
        if p is None or len(self.live) >= MAX_LIVE:
            return False
        self.log("clone", p.n)
        self.live.append(_P(p.h.clone(), p.m.copy(), p.form))
        return True

    def s_free(self):
        p = self.pick()
        if p is None:
            return False
        self.log("free", p.n)
        self.drop(p)
        return True

    def s_unary(self):
        p = self.pick()
        if p is None:
            return False
        op = self.rng.choice(("scale", "negate", "distribute_powers") + (("square", "pow", "add_constant") if p.form == VALUES else ()))
        self.log(op, p.n, p.form)
        if op == "scale":
            c = self.rand_scalar()
            p.h.scale(c)
            self.O.poly_unary(p.m, "scale", c)
        elif op == "negate":
            p.h.negate()
            self.O.poly_unary(p.m, "negate")
        elif op == "distribute_powers":
            g = self.rand_scalar()
            p.h.distribute_powers(g)
            self.O.distribute_powers(p.m, g)
        elif op == "square":
            p.h.square()
            self.O.poly_unary(p.m, "square")
        elif op == "pow":
            e = self.rng.choice((0, 1, 2, 3, 5, 77, (1 << 64) - 1))
            p.h.pow(e)
            self.O.poly_unary(p.m, "pow", e=e)
        else:
            c = self.rand_scalar()
            p.h.add_constant(c)
            self.O.poly_unary(p.m, "add_constant", c)
        return True

    def s_binary(self):
        a = self.pick()
        if a is None:
            return False
        b = self.pick(lambda q: q is not a and q.form == a.form and (q.n == a.n if a.form == VALUES else q.n <= a.n))
        if b is None:
            if len(self.live) >= MAX_LIVE:
                return False
            b = _P(a.h.clone(), a.m.copy(), a.form)                   # x op= x.clone() is as good an operand as any
            self.live.append(b)
        op = self.rng.choice(("add", "sub", "scaled") + (("mul",) if a.form == VALUES else ()))
        self.log("binary", op, a.n, b.n, a.form)
        if op == "scaled":
            s = self.rand_scalar()
            a.h.add_assign_scaled(b.h, s)
            head = a.m[:b.n]                                         # a view: coefficient form adds into the low part (:641-660)
            self.O.poly_add_scaled(head, b.m, s)
        else:
            a.h._binary(b.h, op)
            head = a.m[:b.n]
            self.O.poly_binary(head, b.m, op)
        return True

    def s_transform(self):
        p = self.pick(lambda q: q.n >= 1)
        if p is None:
            return False
        if p.form == COEFFICIENTS:
            which = self.rng.choice(("fft", "coset_fft", "coset_fft_for_generator"))
            self.log(which, p.n)
            if which == "fft":
                p.h.fft()
                self.O.poly_fft(p.m)
            elif which == "coset_fft":
                p.h.coset_fft()
                self.O.poly_coset_fft(p.m)
            else:
                g = self.rand_scalar(nonzero=True)
                p.h.coset_fft_for_generator(g)
                self.O.poly_coset_fft_for_generator(p.m, g)
            p.form = VALUES
        else:
            which = self.rng.choice(("ifft", "icoset_fft", "icoset_fft_for_generator"))
            self.log(which, p.n)
            if which == "ifft":
                p.h.ifft()
                self.O.poly_ifft(p.m)
            elif which == "icoset_fft":
                p.h.icoset_fft()
                self.O.poly_icoset_fft(p.m)
            else:
                g = self.rand_scalar(nonzero=True)
                p.h.icoset_fft_for_generator(g)
                self.O.poly_icoset_fft_for_generator(p.m, g)
            p.form = COEFFICIENTS
        return True

    def s_lde(self):
        p = self.pick(lambda q: q.form == COEFFICIENTS and q.n <= 1 << (MAX_LOG - 3))
        if p is None or len(self.live) >= MAX_LIVE:
            return False
        factor, coset = self.rng.choice((1, 2, 4, 8)), self.rng.random() < 0.5
        self.log("lde", p.n, factor, coset)
        q = p.h.lde(factor, coset)
        self.live.append(_P(q, self.O.poly_lde(p.m, factor, coset), VALUES))
        return True

    def s_lde_all(self):
        p = self.pick(lambda q: q.form == COEFFICIENTS and q.n <= 1 << (MAX_LOG - 4))
        if p is None:
            return False
        same = [q for q in self.live if q.form == COEFFICIENTS and q.n == p.n][:3]
        if len(self.live) + len(same) > MAX_LIVE + 2:
            return False
        factor, coset = self.rng.choice((2, 4)), self.rng.random() < 0.5
        self.log("lde_all", p.n, len(same), factor, coset)
        outs = Polynomial.lde_all([q.h for q in same], factor, coset)
        for q, o in zip(same, outs):
            self.live.append(_P(o, self.O.poly_lde(q.m, factor, coset), VALUES))
        return True

    def s_resize(self):
        p = self.pick()                                                # (generic over the form, :84-137)
        if p is None:
            return False
        which = self.rng.choice(("pad_by_factor", "pad_to_size", "trim_to_degree"))
        if which == "trim_to_degree":
            degree = self.rng.randrange(p.n + 2)
            self.log(which, p.n, degree)
            p.h.trim_to_degree(degree)                                # zeroes [degree + 1 .. n) (:127-136); the size stays
            if degree + 1 < p.n:
                p.m[degree + 1:] = 0
            return True
        factor = self.rng.choice((1, 2, 4))
        if p.n * factor > 1 << MAX_LOG:
            return False
        self.log(which, p.n, factor)
        if which == "pad_by_factor":
            p.h.pad_by_factor(factor)
        else:
            p.h.pad_to_size(p.n * factor)
        m = np.zeros((p.n * factor, 4), dtype=np.uint64)
        m[:p.n] = p.m
        p.m = m
        return True

    def s_batch_inversion(self):
        p = self.pick(lambda q: q.form == VALUES)
        if p is None:
            return False
        has_zero = not p.m.any(axis=1).all()
        self.log("batch_inversion", p.n, "with a zero" if has_zero else "")
        if has_zero:                                                  # Err(SynthesisError::Error) before anything is written (:919)
            with pytest.raises(hodor_amd.HodorError) as e:
                p.h.batch_inversion()
            assert e.value.code == hodor_amd.ERR_INVALID
            self.check(p)
        else:
            p.h.batch_inversion()
            self.O.poly_batch_inversion(p.m)
        return True

    def s_evaluate_at(self):
        p = self.pick(lambda q: q.form == COEFFICIENTS)
        if p is None:
            return False
        g = self.rand_scalar()
        self.log("evaluate_at", p.n)
        assert p.h.evaluate_at(g) == self.O.evaluate_at(p.m, g), self.trace[-12:]
        return True

    def s_host_patch(self):
        p = self.pick()
        if p is None:
            return False
        which = self.rng.choice(("write", "elem_op", "read"))
        first = self.rng.randrange(p.n)
        count = self.rng.randint(1, min(p.n - first, 9))
        self.log(which, p.n, first, count)
        if which == "write":
            patch = self.rand_elements(count)
            p.h.write(first, patch)
            p.m[first:first + count] = patch
        elif which == "read":
            assert np.array_equal(p.h.read(first, count), p.m[first:first + count]), self.trace[-12:]
        else:
            c = self.rand_scalar()
            op = self.rng.choice(("add_constant", "sub_constant", "scale", "negate", "square"))
            p.h.elem_op(first, op, c)
            one = p.m[first:first + 1]
            self.O.poly_unary(one, op, c)
        return True

    def s_as_mut(self):
        """the borrow: writes through the host image, then either the guard's write-back or nothing at all — in which case
        the NEXT thing that happens to the handle (any step of this program) has to find the writes"""
        p = self.pick()
        if p is None:
            return False
        style = self.rng.choice(("patches", "whole", "chunks", "read_only"))
        commit = self.rng.random() < 0.5
        again = self.rng.random() < 0.3
        self.log("as_mut", p.n, style, "commit" if commit else "left open", "twice" if again else "")
        for _ in range(2 if again else 1):
            v = p.h.as_mut()
            assert v.shape == p.m.shape and np.array_equal(v, p.m), self.trace[-12:]   # the image IS the vector
            if style == "whole":
                a = self.rand_elements(p.n)
                v[:] = a
                p.m[:] = a
            elif style == "patches":
                for _ in range(self.rng.randint(1, 4)):
                    i = self.rng.randrange(p.n)
                    e = self.rand_elements(1)[0]
                    v[i] = e
                    p.m[i] = e
            elif style == "chunks":                                   # chunks_mut(period): one value per chunk (ALI's fill)
                period = 1 << self.rng.randrange(0, max(1, p.n.bit_length()))
                vals = self.rand_elements((p.n + period - 1) // period)
                for c in range(0, p.n, period):
                    v[c:c + period] = vals[c // period]
                    p.m[c:c + period] = vals[c // period]
            del v
        if commit:
            p.h.commit_mut()
        return True

    def s_tree(self):
        p = self.pick(lambda q: 2 <= q.n <= 1 << 14)                  # IOP::create takes `&[F]`: either form's as_ref()
        if p is None:
            return False
        combiner = hodor_amd.COSET2 if (p.n >= 4 and self.rng.random() < 0.3) else hodor_amd.TRIVIAL
        self.log("tree", p.n, combiner)
        t = IopTree.create(p.h, combiner)
        nodes = self.O.iop_create(p.m) if combiner == hodor_amd.TRIVIAL else self.O.iop_create_coset2(p.m)
        assert t.get_root() == bytes(nodes[1]), self.trace[-12:]
        idx = self.rng.randrange(p.n)
        vals, path = t.query(idx, p.h)
        if combiner == hodor_amd.TRIVIAL:
            assert vals == [_int(p.m[idx])] and path == [bytes(x) for x in self.O.iop_path(nodes, p.m, idx)], self.trace[-12:]
        else:
            lo = idx % (p.n // 2)
            assert vals == [_int(p.m[lo]), _int(p.m[lo + p.n // 2])], self.trace[-12:]
            assert path == [bytes(x) for x in self.O.iop_path_coset2(nodes, p.m, idx)], self.trace[-12:]
        t.free()
        return True

    def s_fri(self):
        p = self.pick(lambda q: q.form == VALUES and 32 <= q.n <= 1 << 13)
        if p is None:
            return False
        lde_factor = self.rng.choice((2, 4, 8))
        out_deg = self.rng.choice((1, 2))
        through = self.rng.random() < 0.3
        if p.n // lde_factor < 2 * out_deg:
            return False
        batch = [p] + [q for q in self.live if q is not p and q.form == VALUES and q.n == p.n][:1]
        self.log("fri", p.n, lde_factor, out_deg, through, len(batch))
        if len(batch) == 2 and not through:
            got = FriPrototypeHandle.commit_all([q.h for q in batch], lde_factor, out_deg)
        else:
            batch = batch[:1]
            got = [FriPrototypeHandle(p.h, lde_factor, out_deg, through_coefficients=through)]
        for q, g in zip(batch, got):
            exp = self.O.fri_commit(q.m, lde_factor, out_deg, through_coefficients=through)
            assert g.proto.serialized == exp["serialized"], self.trace[-12:]
            g.free()
        return True

    def s_equal(self):
        a = self.pick()
        b = self.pick()
        if a is None:
            return False
        self.log("equal", a.n, b.n)
        exp = a.form == b.form and a.n == b.n and np.array_equal(a.m, b.m)
        assert (a.h == b.h) == exp, self.trace[-12:]
        return True

    def s_refused(self):
        """calls the reference would not compile or would panic on: an error code, and nothing changes"""
        p = self.pick()
        if p is None:
            return False
        self.log("refused", p.n, p.form)
        calls = []
        if p.form == COEFFICIENTS:
            calls += [lambda: p.h.square(), lambda: p.h.batch_inversion(), lambda: p.h.ifft(), lambda: p.h._binary(p.h, "mul"),
                      lambda: p.h.pad_by_factor(3)]
        else:
            calls += [lambda: p.h.fft(), lambda: p.h.lde(2), lambda: p.h.evaluate_at(self.O.one()), lambda: p.h.pad_by_factor(6),
                      lambda: p.h.pad_to_size(p.n // 2) if p.n > 1 else p.h.pad_to_size(3)]
            other = self.pick(lambda q: q.form == VALUES and q.n != p.n)
            if other is not None:
                calls.append(lambda: p.h._binary(other.h, "add"))
        call = self.rng.choice(calls)
        with pytest.raises(hodor_amd.HodorError):
            call()
        return True

    def s_housekeeping(self):
        which = self.rng.choice(("trim", "synchronize", "check_one"))
        self.log(which)
        if which == "trim":
            self.ctx.trim()
        elif which == "synchronize":
            self.ctx.synchronize()
        else:
            p = self.pick()
            if p is not None:
                self.check(p)
        return True

    STEPS = (("s_new", 10), ("s_clone", 4), ("s_free", 6), ("s_unary", 10), ("s_binary", 10), ("s_transform", 10),
             ("s_lde", 5), ("s_lde_all", 3), ("s_resize", 4), ("s_batch_inversion", 4), ("s_evaluate_at", 3),
             ("s_host_patch", 8), ("s_as_mut", 12), ("s_tree", 4), ("s_fri", 3), ("s_equal", 3), ("s_refused", 3),
             ("s_housekeeping", 6))

    def run(self, steps):
        names = [s for s, _ in self.STEPS]
        weights = [w for _, w in self.STEPS]
        done = 0
        while done < steps:
            try:
                if getattr(self, self.rng.choices(names, weights)[0])():
                    done += 1
            except hodor_amd.HodorError as e:          # an error code nobody asked for: say where in the program
                raise AssertionError("%s; last steps: %s" % (e, self.trace[-12:])) from e
        for p in list(self.live):
            self.check(p)
            self.drop(p)


# HODOR_FUZZ_PROGRAMS / HODOR_FUZZ_STEPS / HODOR_FUZZ_SEED0: a longer hunt than the suite's (bench/handle_fuzz_hunt.sh)
PROGRAMS = int(os.environ.get("HODOR_FUZZ_PROGRAMS", "8"))
STEPS = int(os.environ.get("HODOR_FUZZ_STEPS", "150"))
SEED0 = int(os.environ.get("HODOR_FUZZ_SEED0", "0"))


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + PROGRAMS))
def test_random_programs_over_live_handles(gpu_ctxs, oracles, field_name, seed):
    import gc
    ctx, O = gpu_ctxs[field_name], oracles[field_name]
    gc.collect()                                                      # (handles an earlier test left to the collector)
    live0 = ctx.pool_stats()[1]
    prog = _Program(ctx, O, 1000 * seed + sorted(gpu_ctxs).index(field_name))
    try:
        prog.run(STEPS)
    finally:
        for p in prog.live:                                           # a failed program must not poison the session's context
            p.h.free()
    assert ctx.pool_stats()[1] == live0                               # every block went back to the pool


def test_concurrent_programs_share_one_context(gpu_ctxs, oracles):
    """"different handles of one context may be used from different threads — their work is serialised on the one stream"
    (include/hodor_gpu.h, handle API rules): three programs at once on ONE context — one pool, one cache of pinned host
    images, one stream — each over its own handles, `trim` and `synchronize` included.  (ctypes drops the GIL for the
    length of every library call, so the calls do interleave.)"""
    import threading
    import gc
    ctx, O = gpu_ctxs["bn256"], oracles["bn256"]
    gc.collect()
    live0 = ctx.pool_stats()[1]
    progs = [_Program(ctx, O, 777000 + 10 * SEED0 + t) for t in range(3)]
    errors = []

    def work(prog):
        try:
            prog.run(min(STEPS, 400))
        except BaseException as e:          # noqa: BLE001 — reported below, on the main thread
            errors.append((prog.trace[-12:], e))

    ts = [threading.Thread(target=work, args=(p,)) for p in progs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for prog in progs:
        for p in prog.live:
            p.h.free()
    assert not errors, errors[0]
    assert ctx.pool_stats()[1] == live0
