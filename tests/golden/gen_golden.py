"""Generates tests/golden/hodor_golden.json from oracle/pyref.py (Python big-int arithmetic +
hashlib.blake2s) — an implementation independent of both the C oracle and the HIP kernels.

Provenance: RESTATEMENT of the reference's algorithms (the Rust crate cannot be built in this image
and its tests hold no known-answer vectors — SURVEY.md §8c), plus the constants in SURVEY.md
Appendix A/B.  Run from the repo root:  python tests/golden/gen_golden.py
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
    out = {"_provenance": "oracle/pyref.py (Python big-int + hashlib.blake2s restatement); "
                          "NOT produced by the Rust reference"}
    for name, F in (("bn256", P.BN256), ("experiments", P.EXPERIMENTS), ("bn254", P.BN254)):
        rng = random.Random(0x484F444F52 + len(name))
        fld = {"modulus": hx(F.p), "generator": F.g, "S": F.S, "R": hx(F.R),
               "root_of_unity": hx(F.root_of_unity), "cases": {}}
        # NTT / iNTT / coset vectors (canonical residues, hex)
        for n in (1, 2, 4, 8, 32, 64):
            a = [rng.randrange(F.p) for _ in range(n)]
            fld["cases"]["ntt_%d" % n] = {
                "input": [hx(v) for v in a],
                "fft": [hx(v) for v in P.poly_fft(F, a)],
                "ifft": [hx(v) for v in P.poly_ifft(F, a)],
                "coset_fft": [hx(v) for v in P.poly_coset_fft(F, a)],
                "icoset_fft": [hx(v) for v in P.poly_icoset_fft(F, a)],
            }
        # LDE
        for n, f in ((4, 4), (8, 8), (16, 2)):
            a = [rng.randrange(F.p) for _ in range(n)]
            fld["cases"]["lde_%d_x%d" % (n, f)] = {
                "input": [hx(v) for v in a], "factor": f,
                "lde": [hx(v) for v in P.poly_lde(F, a, f)],
                "coset_lde": [hx(v) for v in P.poly_lde(F, a, f, coset=True)],
            }
        # Merkle: leaves given as Montgomery memory images
        for n in (2, 16, 64):
            leafs = [F.to_mont(rng.randrange(F.p)) for _ in range(n)]
            nodes = P.iop_create(leafs)
            fld["cases"]["merkle_%d" % n] = {
                "leafs_mont": [hx(v) for v in leafs],
                "nodes": [x.hex() for x in nodes],
                "challenge": hx(P.interpret_hash(F, nodes[1])),
                "path_3": [x.hex() for x in P.iop_path(nodes, leafs, 3 % n)],
            }
        ones = [F.R] * 16   # make_small_tree, src/iop/blake2s_trivial_iop.rs:377-387
        nodes = P.iop_create(ones)
        fld["cases"]["make_small_tree"] = {"root": nodes[1].hex(),
                                           "challenge": hx(P.interpret_hash(F, nodes[1]))}
        # FRI commit
        for deg, f, outd in ((4, 4, 2), (8, 8, 1), (32, 4, 2)):
            coeffs = [rng.randrange(F.p) for _ in range(deg)]
            lde = P.poly_lde(F, coeffs, f)
            proto = P.fri_commit(F, lde, f, outd)
            fld["cases"]["fri_%d_x%d_o%d" % (deg, f, outd)] = {
                "coeffs": [hx(v) for v in coeffs], "lde_factor": f, "out_deg_plus_one": outd,
                "roots": [r.hex() for r in proto["roots"]],
                "challenges": [hx(c) for c in proto["challenges"]],
                "final_coeffs": [hx(c) for c in proto["final_coeffs"]],
                "serialized": P.fri_serialize(F, proto).hex(),
            }
        out[name] = fld
    out["blake2s"] = {"h_empty": P.b2s(b"").hex(),
                      "h_abc": P.b2s(b"abc").hex(),
                      "h_64_zero": P.b2s(b"\x00" * 64).hex()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hodor_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
