"""ctypes binding of libhodor_gpu.so (include/hodor_gpu.h).

Host arrays are numpy uint64 of shape (n, 4) — the memory image of Rust `&[Fr]`.  Device arrays are
anything exposing `.data_ptr()` (torch int64/uint64 CUDA tensors of shape (n, 4)) or a raw int
address.  No compute happens in Python and there is no CPU fallback.
"""
import ctypes as C
import weakref
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("HODOR_LIB") or os.path.join(_HERE, "libhodor_gpu.so")   # HODOR_LIB: A/B builds (bench/ab.sh)
_CSRC = os.path.join(_HERE, "csrc")

# Fields the reference defines (src/bn256.rs:5-6, src/experiments/mod.rs:19-20)
BN256_FR_MODULUS = 52435875175126190479447740508185965837690552500527637822603658699938581184513
BN256_FR_GENERATOR = 7
EXPERIMENTS_FR_MODULUS = 3618502788666131213697322783095070105623107215331596699973092056135872020481
EXPERIMENTS_FR_GENERATOR = 3

OK, ERR_SIZE, ERR_INVALID, ERR_DEVICE = 0, 1, 2, 3
ABI_VERSION = 5          # HODOR_ABI_VERSION of include/hodor_gpu.h this binding was written against
TRIVIAL, COSET2 = 0, 1   # HODOR_COMBINER_*: the tree format (CosetCombiner, src/iop/mod.rs:22-34)

# every symbol include/hodor_gpu.h declares
EXPORTS = [
    "hodor_abi_version", "hodor_ctx_try_destroy", "hodor_ctx_create", "hodor_ctx_destroy", "hodor_ctx_field_info", "hodor_last_error",
    "hodor_ctx_synchronize", "hodor_knobs_set", "hodor_debug_alloc_calls", "hodor_debug_fail_alloc",
    "hodor_fr_mul", "hodor_fr_add", "hodor_fr_sub", "hodor_fr_pow", "hodor_fr_inverse",
    "hodor_fr_from_repr", "hodor_fr_into_repr", "hodor_domain_new_for_size",
    "hodor_fft", "hodor_lde", "hodor_distribute_powers",
    "hodor_poly_fft", "hodor_poly_coset_fft", "hodor_poly_ifft", "hodor_poly_icoset_fft",
    "hodor_poly_lde", "hodor_poly_coset_lde",
    "hodor_iop_create", "hodor_hash_leaf", "hodor_hash_node", "hodor_iop_challenge", "hodor_iop_path", "hodor_iop_verify",
    "hodor_fri_commit", "hodor_fri_free", "hodor_fri_num_steps", "hodor_fri_roots",
    "hodor_fri_final_root", "hodor_fri_challenges", "hodor_fri_final_coefficients",
    "hodor_fri_intermediate_values", "hodor_fri_tree_nodes", "hodor_fri_serialize",
    "hodor_transcript_new", "hodor_transcript_free", "hodor_transcript_commit_bytes",
    "hodor_transcript_commit_field_element", "hodor_transcript_get_challenge_bytes",
    "hodor_transcript_get_challenge", "hodor_bytes_to_challenge_index",
    "hodor_buf_alloc", "hodor_buf_free", "hodor_buf_upload", "hodor_buf_download",
    "hodor_host_register", "hodor_host_unregister",
    "hodor_fft_dev", "hodor_fft_batch_dev", "hodor_twiddle_mul_dev", "hodor_poly_fft_dev", "hodor_poly_ifft_dev", "hodor_poly_coset_fft_dev",
    "hodor_poly_icoset_fft_dev", "hodor_poly_coset_fft_for_generator_dev", "hodor_poly_icoset_fft_for_generator_dev",
    "hodor_poly_coset_fft_for_generator", "hodor_poly_icoset_fft_for_generator", "hodor_poly_lde_dev", "hodor_poly_lde_batch_dev", "hodor_iop_create_batch_dev", "hodor_distribute_powers_dev", "hodor_poly_degree_one_on_domain_dev", "hodor_precomputed_omegas_dev",
    "hodor_poly_binary_dev", "hodor_poly_add_scaled_dev", "hodor_poly_unary_dev", "hodor_poly_quotient_term_dev",
    "hodor_poly_batch_inversion_dev", "hodor_poly_evaluate_at_dev", "hodor_gen_elements_dev",
    "hodor_sixstep_columns_dev", "hodor_sixstep_rows_dev", "hodor_sixstep_pack_dev", "hodor_transpose_dev",
    "hodor_iop_create_dev", "hodor_iop_query_dev", "hodor_fri_produce_proof", "hodor_fri_commit_dev", "hodor_fri_verify_proof", "hodor_fri_verify_proof_strict", "hodor_fri_verify_prototype",
    "hodor_exchange_direct_copy_dev",
    "hodor_iop_create_combined", "hodor_hash_leaf_combined", "hodor_iop_path_combined", "hodor_iop_verify_combined",
    "hodor_iop_create_combined_dev", "hodor_iop_create_batch_combined_dev", "hodor_iop_query_combined_dev", "hodor_fri_commit_combined",
    "hodor_fri_commit_combined_dev", "hodor_fri_combiner", "hodor_fri_verify_proof_combined",
    "hodor_fri_verify_proof_strict_combined",
    "hodor_ipc_export", "hodor_ipc_import", "hodor_ipc_close", "hodor_exchange_create_direct", "hodor_exchange_direct_flags",
    "hodor_exchange_direct_set_peers", "hodor_exchange_direct_begin_dev", "hodor_exchange_direct_signal_dev",
    "hodor_exchange_direct_wait_dev", "hodor_exchange_direct_release_dev", "hodor_exchange_direct_status", "hodor_sixstep_columns_direct_dev",
    "hodor_sixstep_rows_direct_dev",
    "hodor_exchange_available", "hodor_exchange_unique_id", "hodor_exchange_create", "hodor_exchange_adopt",
    "hodor_exchange_destroy", "hodor_sixstep_exchange_dev", "hodor_sixstep_exchange_wait_dev",
    # round 5: the multi-GPU schedules inside the library (csrc/abi_dist.hip)
    "hodor_dist_split", "hodor_dist_set_transport", "hodor_dist_ntt_forward_dev", "hodor_dist_ntt_inverse_dev",
    "hodor_dist_ntt_begin_dev", "hodor_dist_ntt_end_dev", "hodor_dist_ntt_natural_dev", "hodor_dist_lde_by_cosets_dev",
    "hodor_dist_commit_dev", "hodor_dist_lde_commit_dev", "hodor_exchange_direct_alloc_recv",
    # round 5: proof_from_lde_through_coefficients and the handle API (device-resident Polynomial / IOP)
    "hodor_ctx_host_round_trips", "hodor_ctx_pool_stats", "hodor_ctx_reset_host_round_trips", "hodor_ctx_stream",
    "hodor_ctx_trim", "hodor_fri_commit_h", "hodor_fri_commit_through_coefficients",
    "hodor_fri_commit_through_coefficients_dev", "hodor_fri_commitment_h", "hodor_fri_intermediate_values_h",
    "hodor_fri_produce_proof_h", "hodor_fri_verify_prototype_h", "hodor_iop_create_batch_h", "hodor_iop_create_h",
    "hodor_iop_free_h", "hodor_iop_nodes_h", "hodor_iop_query_h", "hodor_iop_root_h", "hodor_iop_roots_h",
    "hodor_iop_size_h", "hodor_poly_add_assign_scaled_h", "hodor_poly_add_constant_h", "hodor_poly_as_ref_h",
    "hodor_poly_batch_inversion_h", "hodor_poly_binary_h", "hodor_poly_clone_h",
    "hodor_poly_coset_fft_for_generator_h", "hodor_poly_coset_fft_h", "hodor_poly_degree_one_on_domain_h",
    "hodor_poly_dev_ptr_h", "hodor_poly_distribute_powers_h", "hodor_poly_elem_op_h", "hodor_poly_equal_h",
    "hodor_poly_evaluate_at_h", "hodor_poly_fft_h", "hodor_poly_form_h", "hodor_poly_free_h",
    "hodor_poly_from_dev_h", "hodor_poly_from_host_h", "hodor_poly_gen_h", "hodor_poly_icoset_fft_for_generator_h",
    "hodor_poly_icoset_fft_h", "hodor_poly_ifft_h", "hodor_poly_info_h", "hodor_poly_lde_batch_h",
    "hodor_poly_lde_h", "hodor_poly_negate_h", "hodor_poly_new_for_size_h", "hodor_poly_pad_by_factor_h",
    "hodor_poly_pad_to_size_h", "hodor_poly_pow_h", "hodor_poly_quotient_term_h", "hodor_poly_read_h",
    "hodor_poly_scale_h", "hodor_poly_size_h", "hodor_poly_square_h", "hodor_poly_trim_to_degree_h",
    "hodor_poly_write_h",
    # round 6: whole-slice as_mut() with write-back, PCIe byte counters
    "hodor_poly_as_mut_h", "hodor_poly_commit_mut_h", "hodor_ctx_host_traffic",
    "hodor_poly_dense_divisor_on_coset_dev", "hodor_poly_dense_divisor_on_coset_h",
    "hodor_fri_commit_batch_h", "hodor_fri_commit_batch_dev", "hodor_ctx_pool_peak",
]


class HodorError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        names = {ERR_SIZE: "HODOR_ERR_SIZE", ERR_INVALID: "HODOR_ERR_INVALID", ERR_DEVICE: "HODOR_ERR_DEVICE"}
        super().__init__("%s %s" % (names.get(code, code), msg))


class _Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class _FieldInfo(C.Structure):
    _fields_ = [("modulus", C.c_uint64 * 4), ("s", C.c_uint32), ("num_bits", C.c_uint32),
                ("capacity", C.c_uint32), ("one", _Fr), ("generator", _Fr), ("root_of_unity", _Fr)]


def lib_path():
    return _LIB


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)
            if f.endswith((".hip", ".cuh", ".hpp"))] + [os.path.join(_HERE, "..", "include", "hodor_gpu.h")]
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    """Load libhodor_gpu.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError("libhodor_gpu.so is missing: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (there is no CPU fallback)")
        try:
            # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  Whichever is
            # loaded first serves the whole process, and torch cannot initialise on top of the system
            # copy — so when torch is installed, let it load its runtime before libhodor_gpu.so binds.
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(_LIB)
        if _lib.hodor_abi_version() != ABI_VERSION:
            raise ImportError("libhodor_gpu.so speaks ABI revision %d, this binding %d: rebuild (hodor_amd.build())"
                              % (_lib.hodor_abi_version(), ABI_VERSION))
        _lib.hodor_last_error.restype = C.c_char_p
        _lib.hodor_knobs_set.restype = C.c_char_p
        _lib.hodor_debug_alloc_calls.restype = C.c_longlong
        _lib.hodor_debug_fail_alloc.restype = None
        _lib.hodor_debug_fail_alloc.argtypes = [C.c_longlong, C.c_int]
        _lib.hodor_fri_num_steps.restype = C.c_size_t
        _lib.hodor_fri_serialize.restype = C.c_size_t
        _lib.hodor_fri_produce_proof.restype = C.c_size_t
        _lib.hodor_bytes_to_challenge_index.restype = C.c_size_t
        _lib.hodor_transcript_free.restype = None
        _lib.hodor_ctx_destroy.restype = None
        _lib.hodor_fri_free.restype = None
        _lib.hodor_exchange_destroy.restype = None
    return _lib


def knobs_set():
    """Tuning variables the library found in the environment ("NAME=value ...", "" when none)."""
    return lib().hodor_knobs_set().decode()


def _limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def _to_int(l):
    return sum(int(l[i]) << (64 * i) for i in range(4))


def _fr(x):
    return _Fr((C.c_uint64 * 4)(*_limbs(x)))


def _hptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _dptr(t):
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class FriPrototype:
    """Mirror of FRIProofPrototype (src/fri/mod.rs:106-117) held by the library."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        ctx._protos.add(self)            # a prototype must not outlive its context (hodor_fri_free uses it)
        L = ctx.L
        self.num_steps = int(L.hodor_fri_num_steps(handle))
        self.combiner = int(L.hodor_fri_combiner(handle))
        roots = np.zeros((self.num_steps + 1, 32), dtype=np.uint8)
        ctx._chk(L.hodor_fri_roots(handle, roots.ctypes.data_as(C.c_void_p)))
        self.roots = [bytes(r) for r in roots]
        fr = (C.c_uint8 * 32)()
        ctx._chk(L.hodor_fri_final_root(handle, fr))
        self.final_root = bytes(fr)
        ch = np.zeros((self.num_steps, 4), dtype=np.uint64)
        ctx._chk(L.hodor_fri_challenges(handle, _hptr(ch)))
        self.challenges = [_to_int(r) for r in ch]
        need = L.hodor_fri_serialize(handle, None, C.c_size_t(0))
        buf = (C.c_uint8 * need)()
        L.hodor_fri_serialize(handle, buf, C.c_size_t(need))
        self.serialized = bytes(buf)
        # final coefficients: count is encoded in the serialization tail
        off = 8 + 32 * (self.num_steps + 1) + 32 * self.num_steps + 32
        nf = int.from_bytes(self.serialized[off:off + 8], "little")
        fc = np.zeros((nf, 4), dtype=np.uint64)
        ctx._chk(L.hodor_fri_final_coefficients(handle, _hptr(fc)))
        self.final_coeffs = fc

    def intermediate_values(self, step, size):
        out = np.zeros((size, 4), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.hodor_fri_intermediate_values(self.h, C.c_size_t(step), _hptr(out)))
        return out

    def tree_nodes(self, step, size):
        out = np.zeros((size, 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.hodor_fri_tree_nodes(self.h, C.c_int(step), out.ctypes.data_as(C.c_void_p)))
        return out

    def produce_proof(self, lde_values_dev, natural_index):
        """FRIProofPrototype::produce_proof -> dict(queries=[(index, value_int, [path])], roots, final_coeffs)."""
        L = self.ctx.L
        need = L.hodor_fri_produce_proof(self.h, _dptr(lde_values_dev), C.c_size_t(natural_index), None, C.c_size_t(0))
        if need == 0:
            raise HodorError(ERR_INVALID, "hodor_fri_produce_proof")
        buf = (C.c_uint8 * need)()
        got = L.hodor_fri_produce_proof(self.h, _dptr(lde_values_dev), C.c_size_t(natural_index), buf, C.c_size_t(need))
        if got != need:
            raise HodorError(ERR_DEVICE, "hodor_fri_produce_proof")
        raw = bytes(buf)
        o = 0

        def u64():
            nonlocal o
            v = int.from_bytes(raw[o:o + 8], "little")
            o += 8
            return v
        queries = []
        for _ in range(u64()):
            idx = u64()
            value = int.from_bytes(raw[o:o + 32], "little")
            o += 32
            if self.combiner == COSET2:      # both values of the coset in one query
                value = (value, int.from_bytes(raw[o:o + 32], "little"))
                o += 32
            plen = u64()
            path = [raw[o + 32 * k:o + 32 * (k + 1)] for k in range(plen)]
            o += 32 * plen
            queries.append((idx, value, path))
        roots = []
        for _ in range(u64()):
            roots.append(raw[o:o + 32])
            o += 32
        nf = u64()
        final = [int.from_bytes(raw[o + 32 * k:o + 32 * (k + 1)], "little") for k in range(nf)]
        o += 32 * nf
        meta = (u64(), u64(), u64())
        return dict(queries=queries, roots=roots, final_coeffs=final, initial_degree_plus_one=meta[0],
                    output_coeffs_at_degree_plus_one=meta[1], lde_factor=meta[2], raw=raw)

    def verify_prototype(self, lde_values_dev, natural_index):
        """NaiveFriIop::verify_prototype (src/fri/verifier.rs:10-129) -> bool."""
        ok = C.c_int(0)
        self.ctx._chk(self.ctx.L.hodor_fri_verify_prototype(self.h, _dptr(lde_values_dev), C.c_size_t(natural_index),
                                                            C.byref(ok)))
        return bool(ok.value)

    def free(self):
        if self.h:
            if self.ctx.h:               # the context frees the prototypes it still holds when it closes
                self.ctx.L.hodor_fri_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Transcript:
    """Blake2sTranscript (src/transcript/mod.rs:26-80)."""

    def __init__(self, ctx):
        self.ctx, self.h = ctx, C.c_void_p()
        ctx._chk(ctx.L.hodor_transcript_new(ctx.h, C.byref(self.h)))

    def commit_bytes(self, b):
        self.ctx._chk(self.ctx.L.hodor_transcript_commit_bytes(self.h, bytes(b), C.c_size_t(len(b))))

    def commit_field_element(self, mont):
        x = _fr(mont)
        self.ctx._chk(self.ctx.L.hodor_transcript_commit_field_element(self.h, C.byref(x)))

    def get_challenge_bytes(self):
        out = (C.c_uint8 * 32)()
        self.ctx._chk(self.ctx.L.hodor_transcript_get_challenge_bytes(self.h, out))
        return bytes(out)

    def get_challenge(self):
        out = _Fr()
        self.ctx._chk(self.ctx.L.hodor_transcript_get_challenge(self.h, C.byref(out)))
        return _to_int(out.l)

    def __del__(self):
        try:
            if self.h:
                self.ctx.L.hodor_transcript_free(self.h)
                self.h = None
        except Exception:
            pass


RCCL, DIRECT, COPY = 0, 1, 2   # HODOR_TRANSPORT_*


class _DistCalls:
    """The multi-GPU schedules inside the library (csrc/abi_dist.hip) on an exchange handle: one call per distributed
    transform / commit, over whichever transport the handle carries.  Tensors are (m, 4) int64 on the handle's device."""

    def set_transport(self, transport=-1, force_collectives=False):
        self.ctx._chk(self.ctx.L.hodor_dist_set_transport(self.h, C.c_int(transport), C.c_int(1 if force_collectives else 0)))

    def dist_begin(self, src, log_n, omega, inverse=False, log_chunks=0, stream=None):
        op = C.c_void_p()
        w = _fr(omega)
        self.ctx._chk(self.ctx.L.hodor_dist_ntt_begin_dev(self.h, C.c_void_p(stream), _dptr(src), C.c_size_t(src.shape[0]),
                                                          C.c_uint32(log_n), C.byref(w), C.c_int(1 if inverse else 0),
                                                          C.c_uint32(log_chunks), C.byref(op)))
        return {"op": op, "src": src}     # src must outlive the exchange

    def dist_end(self, h, dst):
        self.ctx._chk(self.ctx.L.hodor_dist_ntt_end_dev(h["op"], _dptr(dst)))
        h["op"], h["src"] = None, None
        return dst

    def dist_forward(self, a, b, log_n, omega, log_chunks=0, stream=None):
        return self.dist_end(self.dist_begin(a, log_n, omega, False, log_chunks, stream), b)

    def dist_inverse(self, b, a, log_n, omega, log_chunks=0, stream=None):
        return self.dist_end(self.dist_begin(b, log_n, omega, True, log_chunks, stream), a)

    def dist_natural(self, src, dst, log_n, omega, inverse=False, stream=None):
        w = _fr(omega)
        self.ctx._chk(self.ctx.L.hodor_dist_ntt_natural_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                            C.c_size_t(src.shape[0]), C.c_uint32(log_n), C.byref(w),
                                                            C.c_int(1 if inverse else 0)))
        return dst

    def dist_lde_by_cosets(self, coeffs, log_n, factor, lde_block, coset=False, paired=False, stream=None):
        self.ctx._chk(self.ctx.L.hodor_dist_lde_by_cosets_dev(self.h, C.c_void_p(stream), _dptr(coeffs), C.c_uint32(log_n),
                                                              C.c_size_t(factor), C.c_int(1 if coset else 0),
                                                              C.c_int(1 if paired else 0), _dptr(lde_block)))
        return lde_block

    def dist_commit(self, leafs_block, local_nodes, combiner=TRIVIAL, stream=None):
        """-> (root bytes, dict global node index -> bytes for 1 <= i < 2 P)"""
        P = self.n_ranks
        top = np.zeros((2 * P, 32), dtype=np.uint8)
        root = (C.c_uint8 * 32)()
        self.ctx._chk(self.ctx.L.hodor_dist_commit_dev(self.h, C.c_void_p(stream), _dptr(leafs_block),
                                                       C.c_size_t(leafs_block.shape[0]), C.c_int(combiner),
                                                       _dptr(local_nodes), top.ctypes.data_as(C.c_void_p), root))
        return bytes(root), {i: bytes(top[i]) for i in range(1, 2 * P)}


class Exchange(_DistCalls):
    """hodor_exchange: the all-to-all of the 4-step transform on a communicator and a communication stream the
    library owns (csrc/abi_exchange.hip: grouped ncclSend/ncclRecv, RCCL bound at run time).  The unique id has to
    reach every rank by a channel of the caller's: `Exchange.over_process_group` uses torch.distributed for that one
    broadcast and nothing else."""

    def __init__(self, ctx, unique_id, n_ranks, rank):
        self.ctx, self.n_ranks, self.rank = ctx, n_ranks, rank
        self.h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        ctx._chk(ctx.L.hodor_exchange_create(ctx.h, buf, C.c_uint32(n_ranks), C.c_uint32(rank), C.byref(self.h)))
        ctx._exchanges.add(self)         # closed with (before) the context: the handle's calls report through it

    @staticmethod
    def available():
        return bool(lib().hodor_exchange_available())

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        rc = lib().hodor_exchange_unique_id(buf)
        if rc != OK:
            raise HodorError(rc, "hodor_exchange_unique_id (librccl not available?)")
        return bytes(buf)

    @classmethod
    def over_process_group(cls, ctx, rank, world, group=None):
        """Rank 0 draws the id, torch.distributed carries the 128 bytes to the others, every rank joins."""
        import torch.distributed as dist
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(ctx, box[0], world, rank)

    def exchange(self, send, recv, log_chunks=0, chunk=0, stream=None):
        """Chunk `chunk` of 2^log_chunks of the (n_local, 4) buffers goes on the wire behind everything enqueued on
        `stream` so far; `stream` does not wait (see wait).  Returns the exchange's ticket."""
        ticket = C.c_uint64(0)
        self.ctx._chk(self.ctx.L.hodor_sixstep_exchange_dev(self.h, C.c_void_p(stream), _dptr(send), _dptr(recv),
                                                            C.c_size_t(send.shape[0]), C.c_uint32(log_chunks),
                                                            C.c_uint32(chunk), C.byref(ticket)))
        return int(ticket.value)

    def wait(self, stream=None, ticket=0):
        """`stream` waits for the exchange `ticket` (as returned by exchange) and all earlier ones; 0 = all so far."""
        self.ctx._chk(self.ctx.L.hodor_sixstep_exchange_wait_dev(self.h, C.c_void_p(stream), C.c_uint64(ticket)))

    def close(self):
        if self.h:
            self.ctx.L.hodor_exchange_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RawDeviceArray:
    """A hipMalloc allocation seen through __cuda_array_interface__, so that torch can wrap it without copying."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}


class DirectExchange(_DistCalls):
    """The direct transport of the 4-step exchange (csrc/abi_exchange.hip): every rank maps every rank's receive
    buffers and the producing transform's last pass stores each slab straight into the buffer of the rank it is for —
    no communicator, no copy, no chunks.  One instance per rank; `n_slots` receive buffers of `n_local` elements
    (as many as transforms are in flight).  Wiring the peers:
      * DirectExchange.connect_local(list of instances)   ranks that live in ONE process (played ranks, world 1)
      * x.connect_processes(group)                        one process per rank: hipIpc handles travel once over
                                                          torch.distributed (all_gather_object on `group`)"""

    def __init__(self, ctx, n_ranks, rank, n_local, n_slots=4, coarse=False):
        import torch
        self.ctx, self.n_ranks, self.rank, self.n_local, self.n_slots = ctx, n_ranks, rank, n_local, n_slots
        self.h = C.c_void_p()
        ctx._chk(ctx.L.hodor_exchange_create_direct(ctx.h, C.c_uint32(n_ranks), C.c_uint32(rank), C.c_uint32(n_slots),
                                                    C.byref(self.h)))
        ctx._exchanges.add(self)
        # the receive buffers are allocations of the library's own (hodor_exchange_direct_alloc_recv: FINE-GRAINED device
        # memory unless `coarse`, freed with the handle): an IPC handle names a whole allocation, the peers must find the
        # buffer at its start, and the memory model of the transport is argued for uncached receive buffers
        ptrs = (C.c_void_p * n_slots)()
        ctx._chk(ctx.L.hodor_exchange_direct_alloc_recv(self.h, C.c_size_t(n_local), C.c_int(1 if coarse else 0), ptrs))
        self._raw = [ptrs[i] for i in range(n_slots)]
        self.recv = [torch.as_tensor(_RawDeviceArray(p, (n_local, 4)), device="cuda") for p in self._raw]
        fp, fb = C.c_void_p(), C.c_size_t()
        ctx._chk(ctx.L.hodor_exchange_direct_flags(self.h, C.byref(fp), C.byref(fb)))
        self.flags_ptr, self.flags_bytes = fp.value, fb.value
        self._next = 0
        self._imported = []

    def _set_peers(self, slot, recv_ptrs, flag_ptrs):
        r = (C.c_void_p * self.n_ranks)(*recv_ptrs)
        f = (C.c_void_p * self.n_ranks)(*flag_ptrs)
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_set_peers(self.h, C.c_uint32(slot), r, f))

    @staticmethod
    def connect_local(instances):
        for x in instances:
            for slot in range(x.n_slots):
                x._set_peers(slot, [p.recv[slot].data_ptr() for p in instances], [p.flags_ptr for p in instances])

    def connect_processes(self, group=None):
        import torch.distributed as dist
        L, ctx = self.ctx.L, self.ctx

        def export(ptr):
            h = (C.c_uint8 * 64)()
            ctx._chk(L.hodor_ipc_export(ctx.h, C.c_void_p(ptr), h))
            return bytes(h)

        mine = {"flags": export(self.flags_ptr), "recv": [export(p) for p in self._raw]}
        everyone = [None] * self.n_ranks
        dist.all_gather_object(everyone, mine, group=group)

        def imp(handle):
            p = C.c_void_p()
            ctx._chk(L.hodor_ipc_import(ctx.h, (C.c_uint8 * 64).from_buffer_copy(handle), C.byref(p)))
            self._imported.append(p.value)
            return p.value
        flag_ptrs = [self.flags_ptr if r == self.rank else imp(everyone[r]["flags"]) for r in range(self.n_ranks)]
        for slot in range(self.n_slots):
            recv_ptrs = [self.recv[slot].data_ptr() if r == self.rank else imp(everyone[r]["recv"][slot])
                         for r in range(self.n_ranks)]
            self._set_peers(slot, recv_ptrs, flag_ptrs)
        dist.barrier(group=group)

    def next_slot(self):
        s = self._next
        self._next = (self._next + 1) % self.n_slots
        return s

    def begin(self, slot, stream=None):
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_begin_dev(self.h, C.c_void_p(stream), C.c_uint32(slot)))

    def signal(self, slot, stream=None):
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_signal_dev(self.h, C.c_void_p(stream), C.c_uint32(slot)))

    def wait(self, slot, stream=None):
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_wait_dev(self.h, C.c_void_p(stream), C.c_uint32(slot)))

    def release(self, slot, stream=None):
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_release_dev(self.h, C.c_void_p(stream), C.c_uint32(slot)))

    def status(self):
        """hodor_exchange_direct_status: call after synchronising the stream a generation ran on and BEFORE using its
        output — raises HODOR_ERR_DEVICE when a flag wait of this handle gave up on its peers (that generation's results
        are undefined and the handle is dead)."""
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_status(self.h))

    def synchronize_and_check(self, stream=None):
        """Wait for `stream` (None: the whole device) and then status(): the one safe way to read a result of the
        stream-ordered dist_* / direct calls of this handle (a timed-out flag wait cannot be seen any earlier)."""
        import torch
        if stream is None:
            torch.cuda.synchronize()
        else:
            torch.cuda.ExternalStream(stream).synchronize()
        self.status()

    def copy(self, slot, send, log_chunks=0, chunk=0, stream=None):
        """copy-engine variant: chunk `chunk` of the local send buffer -> the peers' receive buffers of `slot`"""
        self.ctx._chk(self.ctx.L.hodor_exchange_direct_copy_dev(self.h, C.c_void_p(stream), C.c_uint32(slot), _dptr(send),
                                                                C.c_size_t(send.shape[0]), C.c_uint32(log_chunks),
                                                                C.c_uint32(chunk)))

    def columns(self, src, slot, log_n1, log_n2, omega, log_chunks=0, chunk=0, stream=None):
        w = _fr(omega)
        self.ctx._chk(self.ctx.L.hodor_sixstep_columns_direct_dev(self.ctx.h, C.c_void_p(stream), _dptr(src), self.h,
                                                                  C.c_uint32(slot), C.c_uint32(log_n1), C.c_uint32(log_n2),
                                                                  C.byref(w), C.c_uint32(log_chunks), C.c_uint32(chunk)))

    def rows(self, src, slot, log_n1, log_n2, omega, log_chunks=0, chunk=0, stream=None):
        w = _fr(omega)
        self.ctx._chk(self.ctx.L.hodor_sixstep_rows_direct_dev(self.ctx.h, C.c_void_p(stream), _dptr(src), self.h,
                                                               C.c_uint32(slot), C.c_uint32(log_n1), C.c_uint32(log_n2),
                                                               C.byref(w), C.c_uint32(log_chunks), C.c_uint32(chunk)))

    def close(self):
        if self.h:
            import torch
            torch.cuda.synchronize()
            self.ctx.L.hodor_exchange_destroy(self.h)
            self.h = None
            for p in self._imported:
                self.ctx.L.hodor_ipc_close(self.ctx.h, C.c_void_p(p))
            self._imported = []
            self.recv = []
            self._raw, self.recv = [], []      # (the receive buffers went with the handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_LIVE_CONTEXTS = weakref.WeakSet()


def trim_all():
    """Every live context gives its idle pool blocks back to HIP (tests that need most of the HBM for themselves)."""
    for c in list(_LIVE_CONTEXTS):
        try:
            c.trim()
        except HodorError:
            pass


class Context:
    """hodor_ctx: one prime field + one device (device=-1: host-only helpers, no compute)."""

    def __init__(self, modulus=BN256_FR_MODULUS, generator=BN256_FR_GENERATOR, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self._protos = weakref.WeakSet()
        self._exchanges = weakref.WeakSet()
        mod = (C.c_uint64 * 4)(*_limbs(modulus))
        rc = self.L.hodor_ctx_create(mod, C.c_uint64(generator), C.c_int(device), C.byref(self.h))
        if rc != OK:
            self.L.hodor_last_error.restype = C.c_char_p
            why = self.L.hodor_last_error(None)      # NULL: why this thread's last hodor_ctx_create failed (the self-test's verdict)
            raise HodorError(rc, "hodor_ctx_create" + (": " + why.decode() if why else ""))
        self.modulus, self.device = modulus, device
        _LIVE_CONTEXTS.add(self)
        info = _FieldInfo()
        self._chk(self.L.hodor_ctx_field_info(self.h, C.byref(info)))
        self.S, self.num_bits, self.capacity = int(info.s), int(info.num_bits), int(info.capacity)
        self.one = _to_int(info.one.l)
        self.generator = _to_int(info.generator.l)
        self.root_of_unity = _to_int(info.root_of_unity.l)

    def trim(self):
        """Idle blocks of the context's device pool (FRI prototypes, handles) back to HIP."""
        if self.h and self.device >= 0:
            self._chk(self.L.hodor_ctx_trim(self.h))

    def pool_stats(self):
        cached, live = C.c_size_t(), C.c_size_t()
        self._chk(self.L.hodor_ctx_pool_stats(self.h, C.byref(cached), C.byref(live)))
        return cached.value, live.value

    def host_round_trips(self):
        self.L.hodor_ctx_host_round_trips.restype = C.c_uint64
        return int(self.L.hodor_ctx_host_round_trips(self.h))

    def close(self):
        if self.h:
            _LIVE_CONTEXTS.discard(self)
            for proto in list(self._protos):
                proto.free()
            for x in list(self._exchanges):
                x.close()
            rc = self.L.hodor_ctx_try_destroy(self.h)
            if rc != OK:       # refused: something of this context is still alive — say so instead of leaking silently
                msg = self.L.hodor_last_error(self.h).decode()
                self.h = None
                raise HodorError(rc, "hodor_ctx_try_destroy: " + msg)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise HodorError(rc, (self.L.hodor_last_error(self.h) or b"").decode())

    def synchronize(self):
        self._chk(self.L.hodor_ctx_synchronize(self.h))

    # ---- host scalar helpers
    def _bin(self, name, a, b):
        out = _Fr()
        x, y = _fr(a), _fr(b)
        self._chk(getattr(self.L, name)(self.h, C.byref(x), C.byref(y), C.byref(out)))
        return _to_int(out.l)

    def mul(self, a, b):
        return self._bin("hodor_fr_mul", a, b)

    def add(self, a, b):
        return self._bin("hodor_fr_add", a, b)

    def sub(self, a, b):
        return self._bin("hodor_fr_sub", a, b)

    def pow(self, a, e):
        out, x = _Fr(), _fr(a)
        self._chk(self.L.hodor_fr_pow(self.h, C.byref(x), C.c_uint64(e), C.byref(out)))
        return _to_int(out.l)

    def inverse(self, a):
        out, x = _Fr(), _fr(a)
        self._chk(self.L.hodor_fr_inverse(self.h, C.byref(x), C.byref(out)))
        return _to_int(out.l)

    def from_repr(self, canonical):
        out = _Fr()
        c = (C.c_uint64 * 4)(*_limbs(canonical))
        self._chk(self.L.hodor_fr_from_repr(self.h, c, C.byref(out)))
        return _to_int(out.l)

    def into_repr(self, mont):
        c, x = (C.c_uint64 * 4)(), _fr(mont)
        self._chk(self.L.hodor_fr_into_repr(self.h, C.byref(x), c))
        return _to_int(c)

    def bytes_to_challenge_index(self, b, lde_size, lde_factor):
        return int(self.L.hodor_bytes_to_challenge_index(bytes(b), C.c_size_t(len(b)), C.c_size_t(lde_size),
                                                         C.c_size_t(lde_factor)))

    def domain(self, size):
        """Domain::new_for_size -> (size, log_n, generator)"""
        sz, k, g = C.c_uint64(), C.c_uint32(), _Fr()
        self._chk(self.L.hodor_domain_new_for_size(self.h, C.c_uint64(size), C.byref(sz), C.byref(k), C.byref(g)))
        return int(sz.value), int(k.value), _to_int(g.l)

    # ---- slice API
    def fft(self, a, omega, log_n):
        w = _fr(omega)
        self._chk(self.L.hodor_fft(self.h, _hptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n)))

    def lde(self, a, omega, log_n, lde_factor):
        w = _fr(omega)
        self._chk(self.L.hodor_lde(self.h, _hptr(a), C.c_size_t(len(a)), C.byref(w), C.c_uint32(log_n),
                                   C.c_size_t(lde_factor)))

    def distribute_powers(self, a, g):
        gg = _fr(g)
        self._chk(self.L.hodor_distribute_powers(self.h, _hptr(a), C.c_size_t(len(a)), C.byref(gg)))

    def poly_fft(self, a):
        self._chk(self.L.hodor_poly_fft(self.h, _hptr(a), C.c_size_t(len(a))))

    def poly_coset_fft(self, a):
        self._chk(self.L.hodor_poly_coset_fft(self.h, _hptr(a), C.c_size_t(len(a))))

    def poly_ifft(self, a):
        self._chk(self.L.hodor_poly_ifft(self.h, _hptr(a), C.c_size_t(len(a))))

    def poly_icoset_fft(self, a):
        self._chk(self.L.hodor_poly_icoset_fft(self.h, _hptr(a), C.c_size_t(len(a))))

    def poly_coset_fft_for_generator(self, a, gen):
        g = _fr(gen)
        self._chk(self.L.hodor_poly_coset_fft_for_generator(self.h, _hptr(a), C.c_size_t(len(a)), C.byref(g)))

    def poly_icoset_fft_for_generator(self, a, geninv):
        g = _fr(geninv)
        self._chk(self.L.hodor_poly_icoset_fft_for_generator(self.h, _hptr(a), C.c_size_t(len(a)), C.byref(g)))

    def poly_lde(self, coeffs, factor, coset=False):
        out = np.zeros((len(coeffs) * factor, 4), dtype=np.uint64)
        fn = self.L.hodor_poly_coset_lde if coset else self.L.hodor_poly_lde
        self._chk(fn(self.h, _hptr(coeffs), C.c_size_t(len(coeffs)), C.c_size_t(factor), _hptr(out)))
        return out

    def iop_create(self, leafs):
        nodes = np.zeros((len(leafs), 32), dtype=np.uint8)
        self._chk(self.L.hodor_iop_create(self.h, _hptr(leafs), C.c_size_t(len(leafs)),
                                          nodes.ctypes.data_as(C.c_void_p)))
        return nodes

    def hash_leaf(self, mont):
        out, x = (C.c_uint8 * 32)(), _fr(mont)
        self._chk(self.L.hodor_hash_leaf(self.h, C.byref(x), out))
        return bytes(out)

    def hash_node(self, left, right):
        out = (C.c_uint8 * 32)()
        self._chk(self.L.hodor_hash_node(self.h, bytes(left), bytes(right), out))
        return bytes(out)

    def iop_challenge(self, root):
        out = _Fr()
        self._chk(self.L.hodor_iop_challenge(self.h, bytes(root), C.byref(out)))
        return _to_int(out.l)

    def iop_path(self, nodes, leafs, tree_index):
        n = len(leafs)
        path = np.zeros((max(1, n.bit_length() - 1), 32), dtype=np.uint8)
        cnt = C.c_size_t()
        self._chk(self.L.hodor_iop_path(self.h, nodes.ctypes.data_as(C.c_void_p), _hptr(leafs),
                                        C.c_size_t(n), C.c_size_t(tree_index),
                                        path.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return path[:cnt.value]

    def iop_verify(self, root, leaf_mont, path, tree_index):
        ok, x = C.c_int(), _fr(leaf_mont)
        path = np.ascontiguousarray(path)
        self._chk(self.L.hodor_iop_verify(self.h, bytes(root), C.byref(x), path.ctypes.data_as(C.c_void_p),
                                          C.c_size_t(len(path)), C.c_size_t(tree_index), C.byref(ok)))
        return bool(ok.value)

    def fri_commit(self, lde_values, lde_factor, out_deg_plus_one, combiner=TRIVIAL, through_coefficients=False):
        """proof_from_lde_by_values (src/fri/fri_on_values.rs:11-159), or — through_coefficients —
        proof_from_lde_through_coefficients (src/fri/mod.rs:156-248)."""
        h = C.c_void_p()
        fn = self.L.hodor_fri_commit_through_coefficients if through_coefficients else self.L.hodor_fri_commit_combined
        self._chk(fn(self.h, _hptr(lde_values), C.c_size_t(len(lde_values)), C.c_size_t(lde_factor),
                     C.c_size_t(out_deg_plus_one), C.c_int(combiner), C.byref(h)))
        return FriPrototype(self, h)

    # ---- the tree format as a parameter (HODOR_COMBINER_COSET2: leaf k = value[k] || value[k + n/2])
    def iop_create_combined(self, leafs, combiner):
        n = len(leafs)
        nodes = np.zeros((n // 2 if combiner == COSET2 else n, 32), dtype=np.uint8)
        self._chk(self.L.hodor_iop_create_combined(self.h, _hptr(leafs), C.c_size_t(n), C.c_int(combiner),
                                                   nodes.ctypes.data_as(C.c_void_p)))
        return nodes

    def hash_leaf_combined(self, values_mont, combiner):
        out = (C.c_uint8 * 32)()
        arr = (_Fr * len(values_mont))(*[_fr(v) for v in values_mont])
        self._chk(self.L.hodor_hash_leaf_combined(self.h, arr, C.c_int(combiner), out))
        return bytes(out)

    def iop_path_combined(self, nodes, leafs, combiner, natural_index):
        n = len(leafs)
        path = np.zeros((max(1, n.bit_length() - 1), 32), dtype=np.uint8)
        cnt = C.c_size_t()
        self._chk(self.L.hodor_iop_path_combined(self.h, nodes.ctypes.data_as(C.c_void_p), _hptr(leafs), C.c_size_t(n),
                                                 C.c_int(combiner), C.c_size_t(natural_index),
                                                 path.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return path[:cnt.value]

    def iop_verify_combined(self, root, values_mont, path, natural_index, n, combiner):
        ok = C.c_int()
        arr = (_Fr * len(values_mont))(*[_fr(v) for v in values_mont])
        path = np.ascontiguousarray(path)
        self._chk(self.L.hodor_iop_verify_combined(self.h, bytes(root), arr, path.ctypes.data_as(C.c_void_p),
                                                   C.c_size_t(len(path)), C.c_size_t(natural_index), C.c_size_t(n),
                                                   C.c_int(combiner), C.byref(ok)))
        return bool(ok.value)

    # ---- device API (tensors / raw device pointers)
    def fft_dev(self, src, dst, log_n, omega, stream=None):
        w = _fr(omega)
        self._chk(self.L.hodor_fft_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst), C.c_uint32(log_n),
                                       C.byref(w)))

    def fft_batch_dev(self, src, dst, log_n, batch, omega, stream=None):
        w = _fr(omega)
        self._chk(self.L.hodor_fft_batch_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                             C.c_uint32(log_n), C.c_size_t(batch), C.byref(w)))

    def twiddle_mul_dev(self, a, rows, cols, row0, omega, log_order, scale=None, stream=None):
        w = _fr(omega)
        sc = _fr(scale) if scale is not None else None
        self._chk(self.L.hodor_twiddle_mul_dev(self.h, C.c_void_p(stream), _dptr(a), C.c_size_t(rows),
                                               C.c_size_t(cols), C.c_uint64(row0), C.byref(w),
                                               C.c_uint32(log_order), C.byref(sc) if sc is not None else None))

    def _poly_dev(self, name, src, dst, log_n, stream):
        self._chk(getattr(self.L, name)(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst), C.c_uint32(log_n)))

    def poly_fft_dev(self, src, dst, log_n, stream=None):
        self._poly_dev("hodor_poly_fft_dev", src, dst, log_n, stream)

    def poly_ifft_dev(self, src, dst, log_n, stream=None):
        self._poly_dev("hodor_poly_ifft_dev", src, dst, log_n, stream)

    def poly_coset_fft_dev(self, src, dst, log_n, stream=None):
        self._poly_dev("hodor_poly_coset_fft_dev", src, dst, log_n, stream)

    def poly_icoset_fft_dev(self, src, dst, log_n, stream=None):
        self._poly_dev("hodor_poly_icoset_fft_dev", src, dst, log_n, stream)

    def poly_coset_fft_for_generator_dev(self, src, dst, log_n, gen, stream=None):
        g = _fr(gen)
        self._chk(self.L.hodor_poly_coset_fft_for_generator_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                                C.c_uint32(log_n), C.byref(g)))

    def poly_icoset_fft_for_generator_dev(self, src, dst, log_n, geninv, stream=None):
        g = _fr(geninv)
        self._chk(self.L.hodor_poly_icoset_fft_for_generator_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                                 C.c_uint32(log_n), C.byref(g)))

    def poly_lde_dev(self, src, dst, log_n, factor, coset=False, stream=None):
        self._chk(self.L.hodor_poly_lde_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                            C.c_uint32(log_n), C.c_size_t(factor), C.c_int(1 if coset else 0)))

    def poly_lde_batch_dev(self, src, dst, log_n, factor, batch, coset=False, stream=None):
        self._chk(self.L.hodor_poly_lde_batch_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                  C.c_uint32(log_n), C.c_size_t(factor),
                                                  C.c_int(1 if coset else 0), C.c_size_t(batch)))

    def iop_create_batch_dev(self, leafs, n, batch, nodes, stream=None):
        self._chk(self.L.hodor_iop_create_batch_dev(self.h, C.c_void_p(stream), _dptr(leafs), C.c_size_t(n),
                                                    C.c_size_t(batch), _dptr(nodes)))

    def poly_degree_one_on_domain_dev(self, out, n, alpha, c, coset=False, stream=None):
        """out[i] = alpha * u_i + c over the size-n domain (coset: g * w^i): src/polynomials/mod.rs:229-290"""
        aa, cc = _fr(alpha), _fr(c)
        self._chk(self.L.hodor_poly_degree_one_on_domain_dev(self.h, C.c_void_p(stream), _dptr(out), C.c_size_t(n),
                                                             C.byref(aa), C.byref(cc), C.c_int(1 if coset else 0)))

    def distribute_powers_dev(self, a, n, g, stream=None):
        gg = _fr(g)
        self._chk(self.L.hodor_distribute_powers_dev(self.h, C.c_void_p(stream), _dptr(a), C.c_size_t(n), C.byref(gg)))

    def precomputed_omegas_dev(self, log_n, omegas=None, coset=None, omegas_inv=None, stream=None):
        """PrecomputedOmegas::new_for_domain (src/precomputations/mod.rs:14-66) into device buffers."""
        p = lambda t: _dptr(t) if t is not None else C.c_void_p(None)
        self._chk(self.L.hodor_precomputed_omegas_dev(self.h, C.c_void_p(stream), C.c_uint32(log_n), p(omegas),
                                                      p(coset), p(omegas_inv)))

    def fri_verify_proof(self, raw, natural_index, expected_value):
        """NaiveFriIop::verify_proof_queries (src/fri/verifier.rs:131-289) over serialised proof bytes -> bool;
        the reference's Err(..) cases raise HodorError.  Host-only."""
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        ev, ok = _fr(expected_value), C.c_int(0)
        self._chk(self.L.hodor_fri_verify_proof(self.h, buf, C.c_size_t(len(raw)), C.c_size_t(natural_index),
                                                C.byref(ev), C.byref(ok)))
        return bool(ok.value)

    def fri_verify_proof_strict(self, raw, domain_size, lde_factor, out_deg_plus_one, natural_index, expected_value,
                                combiner=TRIVIAL):
        """hodor_fri_verify_proof_strict(_combined): the verifier above, refusing (False) every proof whose
        lde_factor / output_coeffs_at_degree_plus_one are not the ones the CALLER expects, or whose round / query /
        final-coefficient counts or path lengths are not those of a proof over `domain_size` points."""
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        ev, ok = _fr(expected_value), C.c_int(0)
        self._chk(self.L.hodor_fri_verify_proof_strict_combined(
            self.h, buf, C.c_size_t(len(raw)), C.c_int(combiner), C.c_size_t(domain_size), C.c_size_t(lde_factor),
            C.c_size_t(out_deg_plus_one), C.c_size_t(natural_index), C.byref(ev), C.byref(ok)))
        return bool(ok.value)

    def fri_verify_proof_combined(self, raw, combiner, natural_index, expected_value):
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        ev, ok = _fr(expected_value), C.c_int(0)
        self._chk(self.L.hodor_fri_verify_proof_combined(self.h, buf, C.c_size_t(len(raw)), C.c_int(combiner),
                                                         C.c_size_t(natural_index), C.byref(ev), C.byref(ok)))
        return bool(ok.value)

    def poly_binary_dev(self, a, b, n, op, stream=None):
        self._chk(self.L.hodor_poly_binary_dev(self.h, C.c_void_p(stream), _dptr(a), _dptr(b), C.c_size_t(n),
                                               C.c_int({"add": 0, "sub": 1, "mul": 2}[op])))

    def poly_add_scaled_dev(self, a, b, n, scaling, stream=None):
        sc = _fr(scaling)
        self._chk(self.L.hodor_poly_add_scaled_dev(self.h, C.c_void_p(stream), _dptr(a), _dptr(b), C.c_size_t(n),
                                                   C.byref(sc)))

    def poly_unary_dev(self, a, n, op, c=None, e=0, stream=None):
        code = {"negate": 0, "square": 1, "pow": 2, "scale": 3, "add_constant": 4, "sub_constant": 5}[op]
        cc = _fr(c) if c is not None else None
        self._chk(self.L.hodor_poly_unary_dev(self.h, C.c_void_p(stream), _dptr(a), C.c_size_t(n), C.c_int(code),
                                              C.byref(cc) if cc is not None else None, C.c_uint64(e)))

    def sixstep_columns_dev(self, src, dst, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0,
                            chunk=0, stream=None):
        w = _fr(omega)
        self._chk(self.L.hodor_sixstep_columns_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                   C.c_uint32(log_n1), C.c_uint32(log_n2), C.c_uint32(log_p),
                                                   C.c_uint32(rank), C.byref(w), C.c_int(1 if inverse else 0),
                                                   C.c_uint32(log_chunks), C.c_uint32(chunk)))

    def sixstep_rows_dev(self, src, dst, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0,
                         chunk=0, stream=None):
        w = _fr(omega)
        self._chk(self.L.hodor_sixstep_rows_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                C.c_uint32(log_n1), C.c_uint32(log_n2), C.c_uint32(log_p),
                                                C.c_uint32(rank), C.byref(w), C.c_int(1 if inverse else 0),
                                                C.c_uint32(log_chunks), C.c_uint32(chunk)))

    def sixstep_pack_dev(self, src, dst, log_rows, log_cols, log_p, stream=None):
        self._chk(self.L.hodor_sixstep_pack_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                                C.c_uint32(log_rows), C.c_uint32(log_cols), C.c_uint32(log_p)))

    def transpose_dev(self, src, dst, rows, cols, stream=None):
        self._chk(self.L.hodor_transpose_dev(self.h, C.c_void_p(stream), _dptr(src), _dptr(dst),
                                             C.c_size_t(rows), C.c_size_t(cols)))

    def host_register(self, a):
        """Pin a host array the slice API will be called on repeatedly (hipHostRegister)."""
        self._chk(self.L.hodor_host_register(self.h, _hptr(a), C.c_size_t(a.nbytes)))

    def host_unregister(self, a):
        self._chk(self.L.hodor_host_unregister(self.h, _hptr(a)))

    def gen_elements_dev(self, dst, first_index, count, seed, stream=None):
        """dst[r] = element first_index + r of the SplitMix64 input stream `seed`."""
        self._chk(self.L.hodor_gen_elements_dev(self.h, C.c_void_p(stream), _dptr(dst), C.c_uint64(first_index),
                                                C.c_size_t(count), C.c_uint64(seed)))

    def poly_quotient_term_dev(self, acc, f, divisor_inv, n, value, alpha=None, accumulate=True, stream=None):
        """acc[i] (+)= alpha * (f[i] - value) * divisor_inv[i] in one pass (calculate_deep's quotient term)."""
        v = _fr(value)
        al = _fr(alpha) if alpha is not None else None
        self._chk(self.L.hodor_poly_quotient_term_dev(self.h, C.c_void_p(stream), _dptr(acc), _dptr(f), _dptr(divisor_inv),
                                                      C.c_size_t(n), C.byref(v), C.byref(al) if al is not None else None,
                                                      C.c_int(1 if accumulate else 0)))

    def poly_batch_inversion_dev(self, a, n, stream=None):
        self._chk(self.L.hodor_poly_batch_inversion_dev(self.h, C.c_void_p(stream), _dptr(a), C.c_size_t(n)))

    def poly_evaluate_at_dev(self, coeffs, n, g, stream=None):
        out, gg = _Fr(), _fr(g)
        self._chk(self.L.hodor_poly_evaluate_at_dev(self.h, C.c_void_p(stream), _dptr(coeffs), C.c_size_t(n),
                                                    C.byref(gg), C.byref(out)))
        return _to_int(out.l)

    def iop_create_dev(self, leafs, n, nodes, stream=None):
        self._chk(self.L.hodor_iop_create_dev(self.h, C.c_void_p(stream), _dptr(leafs), C.c_size_t(n), _dptr(nodes)))

    def iop_query_dev(self, leafs, nodes, n, natural_index, stream=None):
        value, cnt = _Fr(), C.c_size_t()
        path = np.zeros((max(1, n.bit_length() - 1), 32), dtype=np.uint8)
        self._chk(self.L.hodor_iop_query_dev(self.h, C.c_void_p(stream), _dptr(leafs), _dptr(nodes), C.c_size_t(n),
                                             C.c_size_t(natural_index), C.byref(value),
                                             path.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return _to_int(value.l), path[:cnt.value]

    def fri_commit_dev(self, lde_values, n, lde_factor, out_deg_plus_one, stream=None, combiner=TRIVIAL,
                       through_coefficients=False):
        h = C.c_void_p()
        fn = (self.L.hodor_fri_commit_through_coefficients_dev if through_coefficients
              else self.L.hodor_fri_commit_combined_dev)
        self._chk(fn(self.h, C.c_void_p(stream), _dptr(lde_values), C.c_size_t(n), C.c_size_t(lde_factor),
                     C.c_size_t(out_deg_plus_one), C.c_int(combiner), C.byref(h)))
        return FriPrototype(self, h)

    def iop_create_batch_combined_dev(self, leafs, n, batch, combiner, nodes, stream=None):
        self._chk(self.L.hodor_iop_create_batch_combined_dev(self.h, C.c_void_p(stream), _dptr(leafs), C.c_size_t(n),
                                                             C.c_size_t(batch), C.c_int(combiner), _dptr(nodes)))

    def iop_create_combined_dev(self, leafs, n, combiner, nodes, stream=None):
        self._chk(self.L.hodor_iop_create_combined_dev(self.h, C.c_void_p(stream), _dptr(leafs), C.c_size_t(n),
                                                       C.c_int(combiner), _dptr(nodes)))

    def iop_query_combined_dev(self, leafs, nodes, n, combiner, natural_index, stream=None):
        """-> ([values], path): both values of the coset for COSET2, the queried one for TRIVIAL."""
        values, cnt = (_Fr * 2)(), C.c_size_t()
        path = np.zeros((max(1, n.bit_length() - 1), 32), dtype=np.uint8)
        self._chk(self.L.hodor_iop_query_combined_dev(self.h, C.c_void_p(stream), _dptr(leafs), _dptr(nodes),
                                                      C.c_size_t(n), C.c_int(combiner), C.c_size_t(natural_index), values,
                                                      path.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return [_to_int(values[i].l) for i in range(2 if combiner == COSET2 else 1)], path[:cnt.value]
