#!/usr/bin/env python3
"""Full-size known answers for BASELINE.json configs 1-3, computed by the CPU oracle
(oracle/hodor_oracle.c) in the build container and committed as tests/golden/fullsize_digests.json.

Inputs are the index-addressable SplitMix64 stream of SURVEY.md §8(d) (o_gen_elements ==
hodor_gen_elements_dev), so the GPU regenerates them on the device from (seed, n) alone; outputs
are compared through BLAKE2s-256 digests of the raw buffers (Montgomery limbs, little endian —
the memory image of `&[Fr]`) plus the Merkle root / the serialized FRI prototype in full.

  config 1   2^24 coefficients -> Polynomial::fft -> digest; ifft of that must give the input back
             (the reference compares every element at 2^22: src/fft/mod.rs:128-184)
  config 2   2^22 coefficients -> lde(8) (+ coset_lde(8)) -> digest -> Blake2sIopTree::create -> root,
             digest of all nodes (src/polynomials/mod.rs:1084-1130, src/iop/blake2s_trivial_iop.rs:131-219)
  config 3   2^23 coefficients -> lde(8) = 2^26 codeword -> proof_from_lde_by_values(lde 8, out 1):
             the canonical prototype bytes (src/fri/fri_on_values.rs:11-159)

Provenance: this repository's C restatement of the reference (the Rust crate cannot be built in this
image), NOT the Rust binary.  Takes ~10 minutes on 8 cores and ~10 GB of RAM:
    python tests/golden/gen_fullsize.py [--small]   (--small: 2^16-scale smoke of the script itself)
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import pyref as P  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

SEED_NTT = 0x484F444F52
SEED_LDE = 0x484F444F52 + 1
SEED_FRI = 0x484F444F52 + 2
OUT = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")


def digest(a):
    return hashlib.blake2s(memoryview(np.ascontiguousarray(a)).cast("B"), digest_size=32).hexdigest()


def multi_gpu_shapes(O, logs):
    res = {}
    t0 = time.time()
    for log_n in logs:
        a = O.gen_elements(0, 1 << log_n, SEED_NTT)
        e = {"seed": SEED_NTT, "input": digest(a)}
        O.poly_fft(a)
        e["fft"] = digest(a)
        res[str(log_n)] = e
        print("ntt 2^%d done %.0f s" % (log_n, time.time() - t0), flush=True)
        del a
    return res


def coset2_shapes(O, res, small):
    """The COSET2 tree format (include/hodor_gpu.h, HODOR_COMBINER_COSET2) on the same inputs: the tree over
    lde(8) of 2^22 coefficients and the FRI prototypes of the 2^20 / 2^26 codewords."""
    t0 = time.time()
    for log_n in ([10] if small else [18, 22]):
        a = O.gen_elements(0, 1 << log_n, SEED_LDE)
        l = O.poly_lde(a, 8)
        nodes = O.iop_create_coset2(l)
        e = res["lde"].setdefault(str(log_n), {})
        e["coset2_root"] = bytes(nodes[1]).hex()
        e["coset2_nodes"] = digest(nodes)
        print("coset2 tree 2^%d x8 done %.0f s" % (log_n, time.time() - t0), flush=True)
        del a, l, nodes
    res.setdefault("fri_coset2", {})
    for log_deg in ([9] if small else [17, 23]):
        a = O.gen_elements(0, 1 << log_deg, SEED_FRI)
        code = O.poly_lde(a, 8)
        r = O.fri_commit(code, 8, 1, combiner=1)
        e = {"seed": SEED_FRI, "factor": 8, "out_deg_plus_one": 1, "codeword": digest(code),
             "num_steps": int(r["num_steps"]), "serialized": r["serialized"].hex(),
             "serialized_digest": hashlib.blake2s(r["serialized"], digest_size=32).hexdigest(),
             "final_root": r["final_root"].hex()}
        res["fri_coset2"][str(log_deg + 3)] = e
        print("fri coset2 2^%d done %.0f s" % (log_deg + 3, time.time() - t0), flush=True)
        del a, code, r


def reference_test_sizes(small):
    """The reference's own differential tests at THEIR sizes and over THEIR field (experiments::Fr), restated on the oracle:
      test_parallel_radix4_fft (src/fft/mod.rs:128-184)        2^22 points: parallel_fft == parallel_fft_radix_4 ==
                                                               parallel_DIT_fft, element for element
      test_various_ldes (src/polynomials/mod.rs:1084-1130)     2^22 coefficients x 16: lde_using_multiple_cosets ==
                                                               filtering_lde (zero-pad + best_lde) == fft of the padded vector
    The identities are asserted HERE (and again by tests/test_oracle_cpu.py, the LDE one at full size only on request);
    the digests of the common results are what the GPU is held to (tests/test_gpu_fullsize.py)."""
    O = Oracle(P.EXPERIMENTS.p, P.EXPERIMENTS.g)
    log_n, log_lde = (12, 4) if small else (22, 4)
    n, factor = 1 << log_n, 1 << log_lde
    res = {"field": "src/experiments/mod.rs Fr", "modulus": hex(P.EXPERIMENTS.p)}
    t0 = time.time()
    log_cpus = O.cpus.bit_length() - 1
    r4 = log_cpus - (log_cpus & 1)
    a = O.gen_elements(0, n, SEED_NTT)
    _, k, w = O.domain(n)
    b, c = a.copy(), a.copy()
    e = {"seed": SEED_NTT, "log_n": log_n, "input": digest(a)}
    O.parallel_fft(a, w, k, log_cpus)
    O.parallel_fft_radix_4(b, w, k, r4)
    O.parallel_dit_fft(c, w, k, log_cpus, n)
    assert np.array_equal(a, b) and np.array_equal(a, c), "test_parallel_radix4_fft fails on the oracle"
    e["fft"] = digest(a)
    res["parallel_radix4_fft"] = e
    print("test_parallel_radix4_fft 2^%d done %.0f s" % (log_n, time.time() - t0), flush=True)
    del a, b, c
    coeffs = O.gen_elements(0, n, SEED_LDE)
    e = {"seed": SEED_LDE, "log_n": log_n, "factor": factor, "input": digest(coeffs)}
    coset = O.poly_lde(coeffs, factor)                       # lde_using_multiple_cosets
    _, K, W = O.domain(n * factor)
    filt = np.zeros((n * factor, 4), dtype=np.uint64)
    filt[:n] = coeffs
    O.best_lde(filt, W, K, factor)                           # filtering_lde :355-368
    assert np.array_equal(filt, coset), "test_various_ldes: filtering_lde != lde_using_multiple_cosets on the oracle"
    del coset
    naive = np.zeros((n * factor, 4), dtype=np.uint64)
    naive[:n] = coeffs
    O.best_fft(naive, W, K)                                  # Polynomial::fft of the padded vector
    assert np.array_equal(filt, naive), "test_various_ldes: filtering_lde != naive fft on the oracle"
    e["lde"] = digest(naive)
    res["various_ldes"] = e
    print("test_various_ldes 2^%d x %d done %.0f s" % (log_n, factor, time.time() - t0), flush=True)
    return res


def main():
    if "--ref-sizes" in sys.argv:
        small = "--small" in sys.argv
        res = json.load(open(OUT)) if not small else {}
        res["reference_tests"] = reference_test_sizes(small)
        out = OUT if not small else "/tmp/fullsize_small_ref.json"
        with open(out, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("updated", out)
        return
    if "--coset2" in sys.argv:
        O = Oracle(P.BN256.p, P.BN256.g)
        small = "--small" in sys.argv
        res = json.load(open(OUT)) if not small else {"lde": {}}
        coset2_shapes(O, res, small)
        out = OUT if not small else "/tmp/fullsize_small_coset2.json"
        with open(out, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("updated", out)
        return
    if "--multi" in sys.argv:
        O = Oracle(P.BN256.p, P.BN256.g)
        res = json.load(open(OUT))
        logs = [int(x) for x in sys.argv[sys.argv.index("--multi") + 1:]] or [25, 26, 27]
        res["ntt"].update(multi_gpu_shapes(O, logs))
        with open(OUT, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("updated", OUT)
        return
    small = "--small" in sys.argv
    O = Oracle(P.BN256.p, P.BN256.g)
    res = {"field": "src/bn256.rs Fr", "modulus": hex(P.BN256.p), "generator": "SplitMix64 index-addressable, o_gen_elements",
           "digest": "BLAKE2s-256 of the raw little-endian Montgomery limbs",
           "provenance": "oracle/hodor_oracle.c (C restatement), generated by tests/golden/gen_fullsize.py",
           "ntt": {}, "lde": {}, "fri": {}}
    t0 = time.time()
    for log_n in ([12, 14] if small else [20, 22, 24]):
        n = 1 << log_n
        a = O.gen_elements(0, n, SEED_NTT)
        e = {"seed": SEED_NTT, "input": digest(a)}
        f = a.copy()
        O.poly_fft(f)
        e["fft"] = digest(f)
        g = f.copy()
        O.poly_ifft(g)
        assert np.array_equal(g, a), "oracle: ifft(fft(x)) != x"
        c = a.copy()
        O.poly_coset_fft(c)
        e["coset_fft"] = digest(c)
        i = a.copy()
        O.poly_ifft(i)
        e["ifft"] = digest(i)
        res["ntt"][str(log_n)] = e
        print("ntt 2^%d done %.0f s" % (log_n, time.time() - t0), flush=True)
        del a, f, g, c, i

    for log_n in ([10] if small else [18, 22]):
        n, factor = 1 << log_n, 8
        a = O.gen_elements(0, n, SEED_LDE)
        e = {"seed": SEED_LDE, "factor": factor, "input": digest(a)}
        l = O.poly_lde(a, factor)
        e["lde"] = digest(l)
        nodes = O.iop_create(l)
        e["root"] = bytes(nodes[1]).hex()
        e["nodes"] = digest(nodes)
        del nodes
        c = O.poly_lde(a, factor, coset=True)
        e["coset_lde"] = digest(c)
        nodes = O.iop_create(c)
        e["coset_root"] = bytes(nodes[1]).hex()
        res["lde"][str(log_n)] = e
        print("lde 2^%d x8 done %.0f s" % (log_n, time.time() - t0), flush=True)
        del a, l, c, nodes

    for log_deg in ([9] if small else [17, 23]):
        n, factor = 1 << log_deg, 8
        a = O.gen_elements(0, n, SEED_FRI)
        code = O.poly_lde(a, factor)
        e = {"seed": SEED_FRI, "factor": factor, "out_deg_plus_one": 1, "codeword": digest(code)}
        r = O.fri_commit(code, factor, 1)
        e["num_steps"] = int(r["num_steps"])
        e["serialized"] = r["serialized"].hex()
        e["serialized_digest"] = hashlib.blake2s(r["serialized"], digest_size=32).hexdigest()
        e["final_root"] = r["final_root"].hex()
        res["fri"][str(log_deg + 3)] = e
        print("fri 2^%d done %.0f s" % (log_deg + 3, time.time() - t0), flush=True)
        del a, code, r

    # the multi-GPU bench shapes (bench.py --gpus 2 / 4 / 8 at its default 2^24 points per GPU): input and forward
    # transform only.  `--multi` adds them to the committed file without recomputing the rest (~25 min on 8 cores).
    out = OUT if not small else "/tmp/fullsize_small.json"
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
