"""hodor_amd — MI355X-native NTT / LDE / Merkle-commit / FRI-commit path for the hodor STARK prover.

The product is `libhodor_gpu.so` (hand-written HIP kernels for gfx950 behind the C ABI declared in
include/hodor_gpu.h).  This package is only the thin ctypes binding the tests and bench.py use; it
has no CPU fallback and fails loudly when the library is missing.
"""
from ._lib import (BN256_FR_GENERATOR, BN256_FR_MODULUS, COSET2, ERR_DEVICE, ERR_INVALID, ERR_SIZE, EXPERIMENTS_FR_GENERATOR, OK, TRIVIAL,
                   EXPERIMENTS_FR_MODULUS, Context, DirectExchange, Exchange, FriPrototype, HodorError, build, lib, lib_path)

from .handles import FriPrototypeHandle, IopTree, Polynomial  # noqa: E402  (device-resident objects over the handle API)

__all__ = ["Polynomial", "IopTree", "FriPrototypeHandle", "Context", "DirectExchange", "Exchange", "FriPrototype", "HodorError", "build", "lib", "lib_path",
           "BN256_FR_MODULUS", "BN256_FR_GENERATOR", "EXPERIMENTS_FR_MODULUS",
           "EXPERIMENTS_FR_GENERATOR", "TRIVIAL", "COSET2", "OK", "ERR_SIZE", "ERR_INVALID", "ERR_DEVICE"]
