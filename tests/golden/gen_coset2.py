"""Generates tests/golden/coset2_golden.json from oracle/pyref.py (Python big-int arithmetic + hashlib.blake2s)
for the COSET2 tree format (include/hodor_gpu.h, HODOR_COMBINER_COSET2) — an implementation independent of both
the C oracle and the HIP kernels.

Provenance: the COSET2 format is DEFINED BY THIS BUILD (the reference lists coset combining as not done,
README.md:46; its CosetCombiner trait, src/iop/mod.rs:22-34, has one instance, the trivial one): these are
known answers of the format's definition, not outputs of the Rust crate.  Run from the repo root:
    python tests/golden/gen_coset2.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402


def hx(v):
    return "%064x" % v


def main():
    out = {"_provenance": "oracle/pyref.py (Python big-int + hashlib.blake2s); format defined by this build, "
                          "NOT produced by the Rust reference"}
    for name, F in (("bn256", P.BN256), ("experiments", P.EXPERIMENTS), ("bn254", P.BN254)):
        rng = random.Random(0xC05E72 + len(name))
        fld = {"cases": {}}
        for n in (4, 16, 64):
            vals = [F.to_mont(rng.randrange(F.p)) for _ in range(n)]
            nodes = P.iop_create_coset2(vals)
            fld["cases"]["merkle_%d" % n] = {
                "values_mont": [hx(v) for v in vals],
                "nodes": [x.hex() for x in nodes],
                "path_3": [x.hex() for x in P.iop_path_coset2(nodes, vals, 3 % n)],
                "path_last": [x.hex() for x in P.iop_path_coset2(nodes, vals, n - 1)],
            }
        for log_deg, f, od, idx in ((3, 4, 1, 5), (4, 8, 2, 77), (5, 4, 1, 127)):
            coeffs = [rng.randrange(F.p) for _ in range(1 << log_deg)]
            lde = P.poly_lde(F, coeffs, f)
            proto = P.fri_commit(F, lde, f, od, combiner=P.COSET2)
            proof = P.fri_produce_proof(F, proto, lde, idx, f, od, combiner=P.COSET2)
            fld["cases"]["fri_%d_x%d_o%d" % (log_deg, f, od)] = {
                "coeffs": [hx(v) for v in coeffs], "factor": f, "out_deg": od, "index": idx,
                "serialized": P.fri_serialize(F, proto).hex(),
                "proof": P.fri_proof_to_bytes(proof).hex(),
                "expected_value_mont": hx(F.to_mont(lde[idx])),
                "verifies": bool(od == 1 and P.fri_verify_proof_queries_coset2(F, proof, idx, F.to_mont(lde[idx]))),
            }
        out[name] = fld
    path = os.path.join(ROOT, "tests", "golden", "coset2_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
